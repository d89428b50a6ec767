#!/bin/bash
# A/B helper (run on the GPU box via gpurun): one bench line per environment setting, e.g.
#   bash scripts/bench_variants.sh "PXM_X=0" "PXM_NO_DFT_GROUP=1" "PXM_GEMM_ORDER=plain"
# prints samples/s, ms per step, GEMM us per launch, HBM fraction, MFMA TFLOP/s.
[ $# -eq 0 ] && set -- "PXM_X=0"
for v in "$@"; do
  echo "== $v"
  env $v python bench.py --no-cpu-baseline --no-layout-compare 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],3), round(d['roofline']['mfma_tflops'],1))"
done
