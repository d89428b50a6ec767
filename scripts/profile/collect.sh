#!/bin/bash
# rocprofv3 collection, run ON THE GPU BOX (via gpurun) from the repo root.  Raw output goes under gpurun_out/<tag>;
# scripts/profile/summarise_profiles.py turns it into the committed summaries under profiles/.
#
#   bash scripts/profile/collect.sh bench   <tag>   kernel trace + stats, then FETCH_SIZE and WRITE_SIZE passes of bench.py
#   bash scripts/profile/collect.sh sq      <tag>   SQ counters (VALU / LDS / wait cycles) of the bench's kernels, three passes
#   bash scripts/profile/collect.sh config5 <tag>   kernel trace + stats of the BASELINE configs[4] iteration (L=512, PxMALA)
#   bash scripts/profile/collect.sh any     <tag> script.py [args]   kernel trace + stats of an arbitrary python script
#
# Counter passes are separate runs with --pmc only (never combined with a trace domain); the program itself follows `--`.
set -o pipefail
WHAT=${1:?what: bench | sq | config5 | any}
TAG=${2:?tag}
shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
[ -f "$ROOT/bench.py" ] || { echo "collect.sh: $ROOT/bench.py not found (run from the repo root or set GRAFT_REPO_ROOT)" >&2; exit 2; }
OUT=$ROOT/gpurun_out/$TAG
BENCH_FLAGS="--no-cpu-baseline --no-layout-compare --no-config-legs --no-noise-leg"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
case $WHAT in
bench)
  mkdir -p $OUT/trace $OUT/pmc_fetch $OUT/pmc_write
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 100 --warmup 10 --ramp 0 $BENCH_FLAGS > $OUT/trace.log 2>&1 && \
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 10 --warmup 2 --ramp 0 $BENCH_FLAGS > $OUT/pmc_fetch.log 2>&1 && \
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 10 --warmup 2 --ramp 0 $BENCH_FLAGS > $OUT/pmc_write.log 2>&1
  echo "exit=$?"; grep -h '"metric"' $OUT/trace.log | cut -c1-160 ;;
sq)
  mkdir -p $OUT/a $OUT/b $OUT/c
  timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --steps 10 --warmup 2 --ramp 0 $BENCH_FLAGS > $OUT/a.log 2>&1 && \
  timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $OUT/b -- python3 $ROOT/bench.py --steps 10 --warmup 2 --ramp 0 $BENCH_FLAGS > $OUT/b.log 2>&1 && \
  timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $OUT/c -- python3 $ROOT/bench.py --steps 10 --warmup 2 --ramp 0 $BENCH_FLAGS > $OUT/c.log 2>&1
  echo "exit=$?" ;;
config5)
  # eager launches (graph=0) so that every kernel of the iteration is its own record; one chain = BASELINE configs[4] per GPU
  export ONLY=${ONLY:-1,1,0} NIT=${NIT:-300}
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/scripts/timing/time_config5.py > $OUT.log 2>&1
  echo "exit=$?"; grep "ms/iter" $OUT.log ;;
any)
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 "$@" > $OUT.log 2>&1
  echo "exit=$?"; tail -3 $OUT.log ;;
*) echo "unknown sub-command $WHAT"; exit 2 ;;
esac
