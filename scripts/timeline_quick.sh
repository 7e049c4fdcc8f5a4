# dispatch timeline of the last step of a short lean run (development helper): timeline_quick.sh [ENV=..] [-- bench args]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/tlq
env "$@" rocprofv3 --kernel-trace -d gpurun_out/tlq -o t -- python bench.py --gpus 1 --steps 2 --warmup 1 --lean $BENCH_ARGS > /dev/null 2> gpurun_out/tlq.log
python profiles/timeline_rocpd.py gpurun_out/tlq/t_results.db k_row_sum > gpurun_out/timeline_quick.md
rm -rf gpurun_out/tlq
