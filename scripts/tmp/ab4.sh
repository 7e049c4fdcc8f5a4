cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_pca_gpu.py tests/test_configs_gpu.py tests/test_backed_gpu.py -m gpu -x -q 2>&1 | tail -3
run() { python bench.py --gpus 1 --steps 8 --warmup 2 --lean $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(d['ms_per_step'],3), 'spmm', round(d['kernel_ms_per_step']['spmm_fwd'],3), 'res', d['config']['pca_residual'])"; }
for rep in 1 2; do
run wide
SRX_FWD_NARROW=1 run narrow
run wide_f64 "--storage f64"
done
run solver2 "--solver 2 --steps 2"
