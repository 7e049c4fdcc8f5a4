# tests for the PCA path + a short bench with per-kernel lines (development helper)
python -m pytest tests/test_pca_gpu.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps ${1:-3} --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms/step', round(d['ms_per_step'],3), 'cells/s', round(d['value']))
for k,v in d['kernels'].items(): print(' ', k, v['launches'], round(v['avg_ms'],3), 'ms', round(v['GBps']), 'GB/s')
print(d['stage_ms_per_step'])
"
