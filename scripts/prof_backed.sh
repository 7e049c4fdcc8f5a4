# rocprofv3 kernel stats of a reduced out-of-core run (development helper)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/prof_bk
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bk -o q -- python bench.py --config c5 --backed --cells 2000000 --steps 1 --warmup 0 > gpurun_out/prof_bk.log 2>&1
python - <<'PY'
import csv,glob,json
f=glob.glob('gpurun_out/prof_bk/**/q_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:18]: print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), round(float(r['AverageNs'])/1e3,1),'us', round(float(r['TotalDurationNs'])/1e6,1),'ms')
d=json.loads(open('gpurun_out/prof_bk.log').read().strip().splitlines()[-1]); print(d['h2d'], d['runs'][0])
PY
