# rocprofv3 kernel-trace of a short bench run -> gpurun_out/prof_quick (development helper)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/prof_quick
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_quick -o q -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_quick.log 2>&1
python profiles/summarize_rocpd.py gpurun_out/prof_quick/q_results.db > gpurun_out/prof_quick.md
head -40 gpurun_out/prof_quick.md
