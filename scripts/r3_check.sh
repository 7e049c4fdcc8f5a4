# round 3: full GPU test-suite + the driver's bench command; outputs under gpurun_out/r3check
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r3check
( time python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -60 ) > gpurun_out/r3check/pytest.log 2>&1
tail -5 gpurun_out/r3check/pytest.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r3check/bench.json 2> gpurun_out/r3check/bench.err
tail -c 600 gpurun_out/r3check/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3check/bench.json').read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'value', d['value'])
print(d['kernel_ms_per_step'], d['unattributed_ms_per_step'])
print('cold', d.get('cold_step'))
print('step_roofline', d.get('step_roofline'))
print('c5', {k:v for k,v in d.get('c5_backed',{}).items() if k in ('value','ms_per_step','failed','runs')})
for k in ('cpu_baseline','cpu_baseline_reference_faithful','cpu_baseline_c1_serial'):
    print(k, d.get(k))
print('roofline', d['roofline'])
PY
