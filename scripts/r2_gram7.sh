cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_pca_gpu.py -x -q -m gpu 2>&1 | tail -3
B="python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline"
for v in "SRX_X=0" "SRX_GRAM_RBLK=128" "SRX_GRAM_RBLK=256 SRX_GRAM_CHUNK=64" "SRX_GRAM_CHUNK=64" "SRX_GRAM_CHUNK=1024" "SRX_GRAM_RBLK=32 SRX_GRAM_CHUNK=512"; do
  env $v $B 2>&1 | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); print('$v', 'ms/step', round(d['ms_per_step'],3), 'gram', round(d['kernels']['gram_sparse']['avg_ms'],3), 'compact', round(d['kernels']['hvg_compact']['avg_ms'],3))
except Exception as e: print('$v', 'failed', t[-2:])"
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/kt4 -o g -- $B > /dev/null 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r2/kt4/g_kernel_stats.csv')))
for r in rows[:8]: print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
PY
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d gpurun_out/r2/pm4 -o g -- $B > /dev/null 2>&1
python profiles/summarize_pmc.py gpurun_out/r2/pm4/g_results.db k_gram_stripes | tail -1
