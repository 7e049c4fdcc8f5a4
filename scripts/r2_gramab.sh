# A/B env settings, Gram-related classes only
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="python bench.py --gpus 1 --steps 8 --warmup 2 --lean $BENCH_ARGS"
for v in "$@"; do
  env $v $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('$v', 'ms/step', round(d['ms_per_step'],3), 'gram', round(k['gram_sparse'],3), 'bucket', round(k['gram_bucket'],3), 'sum', round(k['gram_sparse']+k['gram_bucket'],3))"
done
