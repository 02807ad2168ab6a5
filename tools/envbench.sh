# usage: bash tools/envbench.sh "VAR=1 VAR2=x" ...   (one bench line per environment set)
for envs in "$@"; do
  echo "== $envs"
  env $envs timeout -k 10 120 python bench.py --no-cpu-baseline --steps 10 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k: round(v*1000,1) for k,v in d['roofline']['per_kernel_ms'].items()})" || exit 1
done
