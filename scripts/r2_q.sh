cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline"
for v in "SRX_X=0" "SRX_WB_FREE_CUS=64" "SRX_WB_FREE_CUS=16" "SRX_NO_OVERLAP=1"; do
env $v $B 2>&1 | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]); print('$v','ms/step', round(d['ms_per_step'],3), d['stage_ms_per_step'], {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
done
bash scripts/r2_tl.sh
