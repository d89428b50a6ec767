#!/bin/bash
# HBM bytes per kernel of the BASELINE configs[4] iteration (run ON THE GPU BOX from the repo root): two --pmc passes
# (FETCH_SIZE, WRITE_SIZE; never combined with a trace domain) of scripts/timing/time_config5.py, eager launches, one chain.
#   bash scripts/profile/pmc_config5.sh <tag>     -> gpurun_out/<tag>/{fetch,write}; summary printed
set -o pipefail
TAG=${1:?tag}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT/fetch $OUT/write
export ONLY=1,1,0 NIT=${NIT:-12}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/scripts/timing/time_config5.py > $OUT/fetch.log 2>&1 && \
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/scripts/timing/time_config5.py > $OUT/write.log 2>&1
echo "exit=$?"
python3 - $OUT <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
tot = {}
for kind, col in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = glob.glob(f"{out}/{kind}/*/*counter_collection.csv")
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != col:
            continue
        a = acc[r["Kernel_Name"].split("(")[0]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    tot[kind] = acc
names = sorted(tot["fetch"], key=lambda k: -tot["fetch"][k][1])
print("kernel | launches | read MB per launch (FETCH_SIZE KiB x 2: gfx950) | written MB per launch")
for k in names:
    n, v = tot["fetch"][k]
    w = tot["write"].get(k, [1, 0.0])
    if n < 8:
        continue
    print(f"{k[-60:]:60s} {n:5d} {2 * v * 1024 / n / 1e6:9.2f} {w[1] * 1024 / max(w[0], 1) / 1e6:9.2f}")
PY
