cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_bench_launch_gpu.py -x -q -m gpu 2>&1 | tail -5
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r2/bench_full.json 2> gpurun_out/r2/bench_full.err
tail -5 gpurun_out/r2/bench_full.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2/bench_full.json').read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'value', d['value'], 'unattributed', d['unattributed_ms_per_step'])
print(d['kernel_ms_per_step'])
for k in ('f64_storage','approximate_selection','roofline_spmm_iter','skewed_genes','hard_spectrum','incl_h2d','cpu_baseline','cpu_baseline_threaded','roofline','roofline_spmm'):
    v=d.get(k); 
    if isinstance(v,dict): v={a:(b if not isinstance(b,str) else b[:60]) for a,b in v.items() if a not in('note','sample')}
    print(k, v)
PY
