cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for d in 1 2 3 5; do
rm -rf gpurun_out/prof_lean
SRX_GRAM_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lean -o q -- python bench.py --steps 2 --warmup 1 --lean > gpurun_out/prof_lean.log 2>&1
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_lean/**/q_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_gram_stripes' in r['Name'] or 'k_bucket' in r['Name']: print($d, r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
PY
done
