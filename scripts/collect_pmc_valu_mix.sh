#!/bin/bash
# VALU instruction mix of the bench's kernels (fp64 add / mul / fma, int32 / int64, conversions, transcendentals)
TAG=${1:-valumix}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT/a $OUT/b
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --ramp 0 --no-cpu-baseline --no-layout-compare > $OUT/a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --ramp 0 --no-cpu-baseline --no-layout-compare > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in ("a", "b"):
    f = sorted(glob.glob("$OUT/%s/*/*_counter_collection.csv" % sub))
    if not f: print("no csv for", sub); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[-1])):
        k = r["Kernel_Name"].split("(")[0]
        if "k_ring2px_group5<true>" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k[:50], {c: round(sum(v) / len(v) / 9984) for c, v in d.items()}, "(per wave; 9984 waves per launch)")
PY
tail -3 $OUT/a.log | cut -c1-200
