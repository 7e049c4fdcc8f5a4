cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_pca_gpu.py -x -q -m gpu 2>&1 | tail -3
B="python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline"
for v in "SRX_GRAM_LAG=1" "SRX_GRAM_LAG=0" "SRX_GRAM_LAG=2" "SRX_GRAM_LAG=1 SRX_GRAM_Z=1" "SRX_GRAM_LAG=2 SRX_GRAM_Z=1" "SRX_GRAM_LAG=1 SRX_GRAM_RBLK=512" "SRX_GRAM_LAG=2 SRX_GRAM_RBLK=512" "SRX_GRAM_LAG=1 SRX_GRAM_RBLK=2048" "SRX_GRAM_COOP=0"; do
  env $v $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'ms/step', round(d['ms_per_step'],3), 'gram', round(d['kernels']['gram_sparse']['avg_ms'],3), 'compact', round(d['kernels']['hvg_compact']['avg_ms'],3))"
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/kt2 -o g -- $B > /dev/null 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r2/kt2/g_kernel_stats.csv')))
for r in rows[:12]: print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
PY
