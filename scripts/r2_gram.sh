cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_pca_gpu.py -x -q -m gpu 2>&1 | tail -8
python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2/gram1.json 2> gpurun_out/r2/gram1.err; tail -3 gpurun_out/r2/gram1.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2/gram1.json').read().strip().splitlines()[-1])
print('ms/step', round(d['ms_per_step'],3), d['stage_ms_per_step'], {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()}, 'resid', d['config']['pca_residual'], d['config']['subspace_iterations'])
PY
