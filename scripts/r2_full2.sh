cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 2200 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|rror|assert|^FAILED" | tail -8
B="python bench.py --gpus 1 --steps 10 --warmup 3 --lean"
$B 2>&1 | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]); print('ms/step', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()}, 'spmm frac', d['roofline_spmm']['frac'], 'unattr', d['unattributed_ms_per_step'])"
