cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|rror|assert" | tail -8
B="python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline"
for v in "SRX_X=0" "SRX_NO_LAZY=1"; do
env $v $B 2>&1 | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]); print('$v','ms/step', round(d['ms_per_step'],3), d['stage_ms_per_step'], {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
done
