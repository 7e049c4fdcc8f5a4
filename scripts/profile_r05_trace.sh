cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=${1:-gpurun_out/r5p}; mkdir -p $O
B="python bench.py --gpus 1 --steps 3 --warmup 1 --lean"
rm -rf $O/kt
rocprofv3 --kernel-trace --stats -d $O/kt -o c3 -- $B > $O/kt_bench.json 2> $O/kt.log
python profiles/summarize_rocpd.py $O/kt/c3_results.db > $O/rocprof_c3_table.md
python profiles/timeline_rocpd.py $O/kt/c3_results.db k_row_sum > $O/timeline_c3.md
tail -1 $O/kt_bench.json | cut -c1-200
head -60 $O/rocprof_c3_table.md
rm -rf $O/kt
