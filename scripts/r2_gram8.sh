cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2
B="python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline"
for v in "SRX_GRAM_RBLK=256 SRX_GRAM_CHUNK=32" "SRX_GRAM_RBLK=256 SRX_GRAM_CHUNK=64" "SRX_GRAM_RBLK=256 SRX_GRAM_CHUNK=128" "SRX_GRAM_RBLK=512 SRX_GRAM_CHUNK=16" "SRX_GRAM_RBLK=512 SRX_GRAM_CHUNK=32" "SRX_GRAM_RBLK=512 SRX_GRAM_CHUNK=64" "SRX_GRAM_RBLK=1024 SRX_GRAM_CHUNK=16" "SRX_GRAM_RBLK=1024 SRX_GRAM_CHUNK=32" "SRX_GRAM_RBLK=192 SRX_GRAM_CHUNK=64"; do
  rm -rf /tmp/kt; env $v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o g -- $B > /dev/null 2>&1
  python - "$v" <<'PY'
import csv,sys
rows=list(csv.DictReader(open('/tmp/kt/g_kernel_stats.csv')))
d={r['Name'].split('(')[0][-30:]:round(float(r['AverageNs'])/1e3,1) for r in rows}
print(sys.argv[1], {k:v for k,v in d.items() if 'gram_str' in k or 'bucket' in k})
PY
done
