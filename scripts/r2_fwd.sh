cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_pca_gpu.py tests/test_configs_gpu.py tests/test_backed_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
B="python bench.py --gpus 1 --steps 5 --warmup 2 --lean"
for v in "SRX_X=0" "SRX_FWD_TILED=1"; do
env $v $B 2>&1 | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]); print('$v','ms/step', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()}, d['roofline_spmm']['frac'])"
done
python bench.py --gpus 1 --steps 3 --warmup 1 --lean --storage f64 2>&1 | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]); print('f64','ms/step', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
