export PSEG_PLAN_FROM_ENV=1   # PSEG_* variables set below become the plan switches of the engines the Python tools create
for d in 0 8 1 4 9 13; do
  echo "DBG=$d"; PSEG_DBG=$d PSEG_LIB=page-segmentation_amd/csrc/libpseg_diag.so timeout -k 10 120 python bench.py --no-cpu-baseline --steps 10 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k: round(v*1000) for k,v in d['roofline']['per_kernel_ms'].items()})" || exit 1
done
