cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4p2
B="python bench.py --gpus 1 --steps 3 --warmup 1 --lean"
rm -rf gpurun_out/r4p2/kt gpurun_out/r4p2/pf gpurun_out/r4p2/pw
rocprofv3 --kernel-trace --stats -d gpurun_out/r4p2/kt -o c3 -- $B > gpurun_out/r4p2/kt_bench.json 2> gpurun_out/r4p2/kt.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r4p2/pf -o c3 -- $B > /dev/null 2> gpurun_out/r4p2/pf.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r4p2/pw -o c3 -- $B > /dev/null 2> gpurun_out/r4p2/pw.log
python profiles/summarize_rocpd.py gpurun_out/r4p2/kt/c3_results.db > gpurun_out/r4p2/rocprof_c3_table.md
python profiles/make_traffic.py gpurun_out/r4p2/pf/c3_results.db gpurun_out/r4p2/pw/c3_results.db c3 4 > gpurun_out/r4p2/traffic_c3.json
python profiles/timeline_rocpd.py gpurun_out/r4p2/kt/c3_results.db k_row_sum > gpurun_out/r4p2/timeline_c3.md
tail -3 gpurun_out/r4p2/kt_bench.json | cut -c1-300
head -16 gpurun_out/r4p2/rocprof_c3_table.md
rm -rf gpurun_out/r4p2/kt gpurun_out/r4p2/pf gpurun_out/r4p2/pw
