for rep in 1 2; do
for v in "PXM_X=0" "PXM_GEMM_GEOM=41" "PXM_GEMM_MERGE=1"; do
  echo "== $v"; env $v python bench.py --no-cpu-baseline --no-layout-compare 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],3), round(d['roofline']['mfma_tflops'],1))"
done
done
