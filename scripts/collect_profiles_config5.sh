#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: kernel trace + stats of the config-5 iteration
# (L=512 weak-lensing PxMALA, 2 chains, fused operator, eager launches so every kernel is its own record).
# scripts/summarise_profiles.py <bench tag> <name> <sq tag> <this tag> turns it into profiles/<name>_L512_*.
set -o pipefail
TAG=${1:-c5}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ONLY=2,1,0 NIT=300
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/time_config5.py > $OUT.log 2>&1
echo "exit=$?"; grep "ms/iter" $OUT.log
