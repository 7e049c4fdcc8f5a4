cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2
export SRX_BENCH_TRACE=1
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 >gpurun_out/r2/d3.json | grep "\[bench\]" | awk '{print $7}' | tr '\n' ' '; echo
python -c "
import json; d=json.loads(open('gpurun_out/r2/d3.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stage_ms_per_step'])"
