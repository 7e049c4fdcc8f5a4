cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline"
rm -rf /tmp/kt; rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o g -- $B > /dev/null 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/kt/g_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step: find last k_row_sum
idx=[i for i,r in enumerate(rows) if 'k_row_sum' in r['Kernel_Name']]
rows=rows[idx[-1]:]
t0=int(rows[0]['Start_Timestamp'])
for r in rows:
    n=r['Kernel_Name'].split('(')[0].replace('void ','').replace('srx::','')[:34]
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    if e-s>60 or 'row_pass' in n: print(f"{n:36s} q{r.get('Queue_Id','?')} start {s:9.1f} end {e:9.1f} dur {e-s:8.1f}")
print('last end', max(int(r['End_Timestamp']) for r in rows)/1e3 - t0/1e3)
PY
