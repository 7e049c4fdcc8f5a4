# A/B a list of env settings on the lean bench line: scripts/r2_ab.sh "A=1" "B=2 C=3" ...   (BENCH_ARGS adds bench flags)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="python bench.py --gpus 1 --steps 8 --warmup 2 --lean $BENCH_ARGS"
for v in "$@"; do
  env $v $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'ms/step', round(d['ms_per_step'],3), {k:round(x,3) for k,x in d['kernel_ms_per_step'].items()})"
done
