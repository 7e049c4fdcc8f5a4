set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.log
tail -3 gpurun_out/gpu_tests.log
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01e -o c3 -- $B > gpurun_out/prof_r01e_bench.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f2 -o c3 -- $B > gpurun_out/pmc_f2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w2 -o c3 -- $B > gpurun_out/pmc_w2.log 2>&1
ls -la gpurun_out/prof_r01e gpurun_out/pmc_f2 gpurun_out/pmc_w2
