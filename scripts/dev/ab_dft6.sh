# A/B of -DPXM_D5_ABLATE variants of the 256 < L <= 512 phi-DFT kernels (run on the GPU box from the repo root): variant
# libraries under /tmp (only dft5.hip is recompiled), kernel averages of one operator loop under rocprofv3.
#   bash scripts/dev/ab_dft6.sh 0 16 32 2 4 50 54
set -o pipefail
ROOT=$(pwd)
mkdir -p /tmp/pxm_ab gpurun_out/ab_dft6
for v in "$@"; do
  rm -rf /tmp/pxm_ab/build_$v && cp -r pxmcmc_amd/csrc/build /tmp/pxm_ab/build_$v && rm -f /tmp/pxm_ab/build_$v/dft5*.o
  make -C pxmcmc_amd/csrc -j16 BUILD=/tmp/pxm_ab/build_$v OUT=/tmp/pxm_ab/lib$v.so EXTRA="-DPXM_D5_ABLATE=$v" > /tmp/pxm_ab/build_$v.log 2>&1 || { echo "build $v failed"; tail -5 /tmp/pxm_ab/build_$v.log; continue; }
  export PXM_LIB_PATH=/tmp/pxm_ab/lib$v.so
  (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/ab_dft6/v$v -- python3 $ROOT/scripts/timing/time_wl_operator.py > $ROOT/gpurun_out/ab_dft6/v$v.log 2>&1)
  python3 - "$v" <<'PY'
import csv, glob, sys
v = sys.argv[1]
f = glob.glob(f"gpurun_out/ab_dft6/v{v}/*/*_kernel_stats.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "px6" in r["Name"] or "ring6" in r["Name"]]
print(f"ABLATE={v}:", "  ".join(f"{r['Name'].split('(')[0].split('::')[-1]} x{r['Calls']} {float(r['AverageNs']) / 1e3:.1f} us" for r in rows), flush=True)
PY
done
