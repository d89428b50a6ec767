# A/B of environment switches (run on the GPU box from the repo root): one bench line per setting, e.g.
#   bash scripts/dev/ab_env.sh "PXM_X=0" "PXM_GEMM_VALU=1"
mkdir -p gpurun_out/ab
for kv in "$@"; do
  env "$kv" python bench.py --no-config-legs --no-cpu-baseline --no-layout-compare --no-noise-leg --steps 1000 > gpurun_out/ab/e.json 2> gpurun_out/ab/e.err || { echo "$kv: bench failed"; tail -3 gpurun_out/ab/e.err; continue; }
  python - "$kv" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab/e.json"))
print(sys.argv[1], "| ms", round(d["ms_per_step"], 4), "samples/s", round(d["value"]), "| gemm", [(c["workgroups"], round(c["avg_us"], 1)) for c in d["roofline"]["launch_classes"]],
      "| dft", round(d["dft_kernel"]["avg_launch_us"], 1), "| mfma TF", round(d["roofline"]["mfma_tflops"], 1))
PY
done
