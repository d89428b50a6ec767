O3=$GRAFT_REPO_ROOT/pxmcmc_amd/lib/libpxmcmc_amd_o3.so
for rep in 1 2; do
for v in "PXM_X=0" "PXM_NO_DFT_GROUP=1 PXM_DFT_TOP_FIRST=1" "PXM_NO_DFT_GROUP=1 PXM_DFT_TOP_FIRST=1 PXM_LIB_PATH=$O3" "PXM_NO_DFT_GROUP=1 PXM_DFT_TOP_FIRST=1 PXM_TOP_SPLIT=1 PXM_NSIDE=3 PXM_LIB_PATH=$O3" "PXM_NO_DFT_GROUP=1 PXM_TOP_SPLIT=1 PXM_NSIDE=3 PXM_LIB_PATH=$O3" "PXM_NO_DFT_GROUP=1 PXM_DFT_TOP_FIRST=1 PXM_TOP_SPLIT=1 PXM_NSIDE=3"; do
  echo "== $v" | sed "s|$O3|O3|"; env $v python bench.py --no-cpu-baseline --steps 400 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],3))"
done
done
