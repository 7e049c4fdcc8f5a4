# per-kernel times of the pre-PCA passes at several shard sizes (does a pass speed up when the shard fits the
# 256 MB Infinity Cache?) — development helper
for n in 20000 40000 80000 160000 1300000; do
python bench.py --cells $n --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']; nnz=d['config']['nnz_per_gpu']
print($n, 'nnz', nnz, ' '.join('%s %.3f ms (%.2f ns/knnz)' % (a, k[a]['avg_ms'], k[a]['avg_ms']*1e6/(nnz/1e3)) for a in ('normalize_log1p','gene_moments','hvg_compact','gram_sparse')), 'step', round(d['ms_per_step'],3))
"
done
