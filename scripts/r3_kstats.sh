# rocprofv3 kernel stats of the lean bench: scripts/r3_kstats.sh [grep pattern]   (env in front as usual)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/fp; rocprofv3 --kernel-trace --stats -d /tmp/fp -o t --output-format csv -- python bench.py --gpus 1 --steps ${STEPS:-3} --warmup 1 --lean $BENCH_ARGS > /dev/null 2>&1
python - "$1" <<'PY'
import csv,glob,sys
pat=sys.argv[1] if len(sys.argv)>1 else ''
rows=list(csv.DictReader(open(glob.glob('/tmp/fp/*kernel_stats.csv')[0])))
for r in rows:
    n=r['Name']
    if pat and pat not in n: continue
    print(f"{n[:70]:70s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.3f} ms")
PY
