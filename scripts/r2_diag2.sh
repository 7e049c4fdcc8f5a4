cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2
export SRX_BENCH_TRACE=1
for v in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --max-copies-gb 80" "--steps 10 --warmup 2"; do
  echo "=== $v"; python bench.py --gpus 1 $v --no-cpu-baseline 2>&1 >/dev/null | grep "\[bench\]"
done
echo "=== nograph"; SRX_NO_GRAPH=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep "\[bench\]"
