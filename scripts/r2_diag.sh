# round 2: reproduce the driver's bench command and attribute the step time (kernel + HIP API traces)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2/diag_20_5.json 2> gpurun_out/r2/diag_20_5.err
python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2/diag_5_2.json 2> gpurun_out/r2/diag_5_2.err
rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d gpurun_out/r2/hiptrace -o t -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2/hiptrace.json 2> gpurun_out/r2/hiptrace.err
ls -la gpurun_out/r2/hiptrace/*
for f in gpurun_out/r2/diag_20_5.json gpurun_out/r2/diag_5_2.json gpurun_out/r2/hiptrace.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],3), d['stage_ms_per_step'], {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()})
PY
done
head -40 gpurun_out/r2/hiptrace/*hip_api_stats.csv
# drop the big traces, keep stats
find gpurun_out/r2/hiptrace -name '*trace.csv' -size +20M -delete
