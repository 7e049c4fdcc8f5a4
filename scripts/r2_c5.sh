cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2
free -g | head -2
( time timeout 900 python bench.py --config c5 --backed --steps 2 ) > gpurun_out/r2/c5_backed.json 2> gpurun_out/r2/c5_backed.err; tail -4 gpurun_out/r2/c5_backed.err
cut -c1-1500 gpurun_out/r2/c5_backed.json
( time timeout 600 python bench.py --config c5 --steps 3 --warmup 1 --lean ) > gpurun_out/r2/c5_resident.json 2> gpurun_out/r2/c5_resident.err; tail -4 gpurun_out/r2/c5_resident.err
python -c "
import json; d=json.loads(open('gpurun_out/r2/c5_resident.json').read().strip().splitlines()[-1]); print('c5 resident', d['ms_per_step'], d['value'], d['kernel_ms_per_step'], d['config']['pca_residual'])"
( timeout 600 python bench.py --config c2 --steps 10 --warmup 2 --lean ) 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', d['ms_per_step'], d['value'], d['kernel_ms_per_step'])"
