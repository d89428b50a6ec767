#!/bin/bash
# kernel-trace of the bench with every DFT launch on one stream (isolated kernel durations)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-serial}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PXM_NO_SIDE_STREAMS=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-layout-compare > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$OUT/*/*_kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:8.1f} us", f"{float(r['Percentage']):5.1f}%")
PY
grep -h '"metric"' $OUT/log.txt | cut -c1-200
