cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2
export SRX_GRAM_LAG=0
for N in 6000 40000 300000 1300000; do
  B="python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --cells $N"
  rm -rf gpurun_out/r2/s$N
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d gpurun_out/r2/s$N -o g -- $B > gpurun_out/r2/s$N.log 2>&1
  echo "N=$N"; python profiles/summarize_pmc.py gpurun_out/r2/s$N/g_results.db k_gram_stripes | tail -1
  python - $N <<'PY'
import sqlite3,sys
c=sqlite3.connect(f'gpurun_out/r2/s{sys.argv[1]}/g_results.db')
t=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
T=lambda s:[x for x in t if s in x][0]
kd,ks=T('kernel_dispatch'),T('kernel_symbol')
for r in c.execute(f"select s.display_name, count(*), avg(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id where s.display_name like '%gram_stripes%' or s.display_name like '%k_bucket%' group by s.display_name"): print('   ', r[0][:40], r[1], round(r[2]/1e3,1),'us')
PY
done
