#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: kernel-trace stats + two PMC passes for the bench.
# Writes raw output under gpurun_out/profN; scripts/summarise_profiles.py turns it into profiles/*.
set -o pipefail
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT/trace $OUT/pmc_fetch $OUT/pmc_write
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --ramp 0 --no-cpu-baseline --no-layout-compare > $OUT/trace.log 2>&1 && \
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --ramp 0 --no-cpu-baseline --no-layout-compare > $OUT/pmc_fetch.log 2>&1 && \
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --ramp 0 --no-cpu-baseline --no-layout-compare > $OUT/pmc_write.log 2>&1
echo "exit=$?"; grep -h '"metric"' $OUT/trace.log | cut -c1-160
ls $OUT/*/*/ | head -30
