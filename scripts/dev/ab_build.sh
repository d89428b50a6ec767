# A/B of library build variants (run on the GPU box from the repo root): variant libraries are built under
# /tmp and loaded with PXM_LIB_PATH; prints ms per step and the DFT launch time with the f32-unit and the fp64 noise stream.
#   bash scripts/dev/ab_build.sh "-DPXM_D5_N64_GROUP=2" "-DPXM_D5_PHILOX_FIRST=0" ...
set -o pipefail
mkdir -p /tmp/pxm_ab gpurun_out/ab
i=0
LIBS=(default)
for extra in "$@"; do
  i=$((i+1))
  make -C pxmcmc_amd/csrc -j16 BUILD=/tmp/pxm_ab/build_$i OUT=/tmp/pxm_ab/lib$i.so EXTRA="$extra" > /tmp/pxm_ab/build_$i.log 2>&1 || { echo "build $extra failed"; tail -5 /tmp/pxm_ab/build_$i.log; continue; }
  LIBS+=("/tmp/pxm_ab/lib$i.so|$extra")
done
for ent in "${LIBS[@]}"; do
  lib=${ent%%|*}
  if [ "$lib" = default ]; then unset PXM_LIB_PATH; else export PXM_LIB_PATH=$lib; fi
  python bench.py --no-config-legs --no-cpu-baseline --no-layout-compare --steps 1000 > gpurun_out/ab/b.json 2>/dev/null
  python - "$ent" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab/b.json"))
print(sys.argv[1], "| f64 (headline) ms", round(d["ms_per_step"], 4), "dft", round(d["dft_kernel"]["avg_launch_us"], 1), "| f32 ms",
      round(d["noise_leg"]["ms_per_step"], 4), "dft32", round(d["noise_leg"]["dft_kernel_avg_launch_us"], 1))
PY
done
