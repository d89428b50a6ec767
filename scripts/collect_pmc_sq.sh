#!/bin/bash
# SQ counters of the bench's kernels (VALU / LDS / wait cycles), two --pmc passes; raw output under gpurun_out/<tag>
TAG=${1:-sq1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT/a $OUT/b
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --ramp 0 --no-cpu-baseline --no-layout-compare > $OUT/a.log 2>&1 && \
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --ramp 0 --no-cpu-baseline --no-layout-compare > $OUT/b.log 2>&1
RC=$?
mkdir -p $OUT/c
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $OUT/c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --ramp 0 --no-cpu-baseline --no-layout-compare > $OUT/c.log 2>&1
echo "exit=$RC $?"
python3 - <<PY
import csv, glob, collections
for sub in ("a", "b", "c"):
    f = sorted(glob.glob("$OUT/%s/*/*_counter_collection.csv" % sub))
    if not f: print("no csv for", sub); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[-1])):
        k = r["Kernel_Name"].split("(")[0]
        if "k_ring2px_group" in k or "k_sht_gemm<1, 2" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k[:50], {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
