#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: HBM read bytes (FETCH_SIZE) per kernel of the config-5 iteration.
set -o pipefail
TAG=${1:-c5pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT/pmc_fetch
cd /tmp && export TMPDIR=/tmp
export ONLY=1,1,0 NIT=20
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/scripts/time_config5.py > $OUT/pmc_fetch.log 2>&1
echo "exit=$?"
