#!/bin/bash
# usage: prof_any.sh TAG script.py [args]   -- kernel-trace stats of an arbitrary python script
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/$@ > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$OUT/*/*_kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:24]:
    print(r['Name'][:72].ljust(72), r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:9.1f} us", f"{float(r['Percentage']):5.1f}%")
PY
tail -4 $OUT/log.txt
