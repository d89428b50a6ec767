# A/B of build variants of the recursion kernels (run on the GPU box from the repo root): variant libraries under /tmp (only
# sht_rec.hip is recompiled), kernel averages of scripts/timing/time_rec.py under rocprofv3 + its check against the table GEMM.
#   bash scripts/dev/ab_rec.sh "" "-DPXM_REC_LDS_REDUCE=1"
set -o pipefail
ROOT=$(pwd)
mkdir -p /tmp/pxm_ab gpurun_out/ab_rec
i=0
for extra in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pxm_ab/build_r$i && cp -r pxmcmc_amd/csrc/build /tmp/pxm_ab/build_r$i && rm -f /tmp/pxm_ab/build_r$i/sht_rec*.o
  make -C pxmcmc_amd/csrc -j16 BUILD=/tmp/pxm_ab/build_r$i OUT=/tmp/pxm_ab/libr$i.so EXTRA="$extra" > /tmp/pxm_ab/build_r$i.log 2>&1 || { echo "build '$extra' failed"; tail -5 /tmp/pxm_ab/build_r$i.log; continue; }
  export PXM_LIB_PATH=/tmp/pxm_ab/libr$i.so
  (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/ab_rec/v$i -- python3 $ROOT/scripts/timing/time_rec.py 512 2:1 > $ROOT/gpurun_out/ab_rec/v$i.log 2>&1)
  grep -h "max\|diff\|err" gpurun_out/ab_rec/v$i.log | tail -4
  python3 - "$i" "$extra" <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/ab_rec/v{sys.argv[1]}/*/*_kernel_stats.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_rec_" in r["Name"]]
print(f"[{sys.argv[2] or 'default'}]", "  ".join(f"{r['Name'].split('(')[0].split('::')[-1]} x{r['Calls']} {float(r['AverageNs']) / 1e3:.1f} us" for r in rows), flush=True)
PY
done
