cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2
B="python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline"
for v in "SRX_GRAM_LAG=-1" "SRX_GRAM_LAG=-2" "SRX_GRAM_LAG=-8" "SRX_GRAM_LAG=-64" "SRX_GRAM_LAG=1"; do
  env $v $B 2>&1 | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); print('$v', 'ms/step', round(d['ms_per_step'],3), 'gram', round(d['kernels']['gram_sparse']['avg_ms'],3), 'compact', round(d['kernels']['hvg_compact']['avg_ms'],3))
except Exception as e: print('$v', 'failed', t[-2:])"
done
