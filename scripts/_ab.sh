cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5ab
timeout 900 python -m pytest tests/test_pca_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | tail -5
python bench.py --gpus 1 --steps 10 --warmup 3 --lean > gpurun_out/r5ab/new.json 2> gpurun_out/r5ab/new.err
SRX_COUNT_V1=1 python bench.py --gpus 1 --steps 10 --warmup 3 --lean > gpurun_out/r5ab/old.json 2> gpurun_out/r5ab/old.err
python - <<'PY'
import json
for f in ('new','old'):
    try:
        d=json.loads(open(f'gpurun_out/r5ab/{f}.json').read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], {k:v for k,v in d.get('kernel_classes_ms',{}).items()} if 'kernel_classes_ms' in d else [ (k,v) for k,v in d['config'].items() if 'ms' in k][:12])
    except Exception as e: print(f, 'ERR', e)
PY
