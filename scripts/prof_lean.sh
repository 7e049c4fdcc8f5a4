# rocprofv3 kernel-trace of a short lean bench run: top kernels by average duration (development helper)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/prof_lean
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lean -o q -- python bench.py --steps 4 --warmup 1 --lean > gpurun_out/prof_lean.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_lean/**/q_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:16]: print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), round(float(r['AverageNs'])/1e3,1),'us')
PY
