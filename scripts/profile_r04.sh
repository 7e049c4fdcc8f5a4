# round 4: rocprofv3 kernel trace (+stats), PMC traffic passes, Gram PMC table, timeline of the last step — of the bench command
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4p
B="python bench.py --gpus 1 --steps 3 --warmup 1 --lean"
rm -rf gpurun_out/r4p/kt gpurun_out/r4p/pf gpurun_out/r4p/pw
rocprofv3 --kernel-trace --stats -d gpurun_out/r4p/kt -o c3 -- $B > gpurun_out/r4p/kt_bench.json 2> gpurun_out/r4p/kt.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r4p/pf -o c3 -- $B > /dev/null 2> gpurun_out/r4p/pf.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r4p/pw -o c3 -- $B > /dev/null 2> gpurun_out/r4p/pw.log
python profiles/summarize_rocpd.py gpurun_out/r4p/kt/c3_results.db > gpurun_out/r4p/rocprof_c3_table.md
python profiles/make_traffic.py gpurun_out/r4p/pf/c3_results.db gpurun_out/r4p/pw/c3_results.db c3 4 > gpurun_out/r4p/traffic_c3.json
python profiles/timeline_rocpd.py gpurun_out/r4p/kt/c3_results.db k_row_sum > gpurun_out/r4p/timeline_c3.md
bash scripts/pmc_gram.sh k_gram_stripes > gpurun_out/r4p/pmc_gram.txt 2>&1
bash scripts/pmc_gram.sh k_gene_moments > gpurun_out/r4p/pmc_moments.txt 2>&1
bash scripts/pmc_kernel.sh k_spmm_rows > gpurun_out/r4p/pmc_spmm_rows.txt 2>&1
tail -3 gpurun_out/r4p/kt_bench.json | cut -c1-400
head -30 gpurun_out/r4p/rocprof_c3_table.md
rm -rf gpurun_out/r4p/kt gpurun_out/r4p/pf gpurun_out/r4p/pw gpurun_out/pmcg*
