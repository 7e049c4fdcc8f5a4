cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_pca_gpu.py tests/test_configs_gpu.py tests/test_backed_gpu.py -x -q -m gpu 2>&1 | tail -3
B="python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline"
$B 2>&1 | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]); print('ms/step', round(d['ms_per_step'],3), d['stage_ms_per_step'], {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o g -- $B > /dev/null 2>&1
python - <<'PY'
import csv,sys
rows=list(csv.DictReader(open('/tmp/kt/g_kernel_stats.csv')))
for r in rows[:12]: print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
PY
