# A/B of the Gram workgroup placement on one box (development helper)
for m in 16 8 0 8 16 0; do
SRX_GRAM_MAP=$m python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('map $m', 'gram', round(d['kernels']['gram_sparse']['avg_ms'],3), 'step', round(d['ms_per_step'],3))
"
done
cd /tmp; export TMPDIR=/tmp
for m in 0 8; do
SRX_GRAM_MAP=$m rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf$m -o q -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import sqlite3
c=sqlite3.connect('/tmp/pf$m/q_results.db')
t=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
T=lambda s:[x for x in t if s in x][0]
rows=c.execute(f"select s.display_name, sum(e.value), count(distinct d.id) from {T('pmc_event')} e join {T('info_pmc')} p on e.pmc_id=p.id join {T('kernel_dispatch')} d on d.event_id=e.event_id join {T('kernel_symbol')} s on d.kernel_id=s.id where p.name='FETCH_SIZE' and s.display_name like '%gram_sparse%' group by s.display_name").fetchall()
for n,v,k in rows: print('map $m FETCH per launch GB', v/k*1024*2/1e9)
PY
done
