# round 5: rocprofv3 kernel trace (+stats), PMC traffic passes, PMC tables of the three largest kernels, timeline of the last step — of the bench command
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5p; mkdir -p $O
B="python bench.py --gpus 1 --steps 3 --warmup 1 --lean"
rm -rf $O/kt $O/pf $O/pw
rocprofv3 --kernel-trace --stats -d $O/kt -o c3 -- $B > $O/kt_bench.json 2> $O/kt.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o c3 -- $B > /dev/null 2> $O/pf.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o c3 -- $B > /dev/null 2> $O/pw.log
python profiles/summarize_rocpd.py $O/kt/c3_results.db > $O/rocprof_c3_table.md
python profiles/make_traffic.py $O/pf/c3_results.db $O/pw/c3_results.db c3 4 > $O/traffic_c3.json
python profiles/timeline_rocpd.py $O/kt/c3_results.db k_row_sum > $O/timeline_c3.md
bash scripts/pmc_gram.sh k_gram_stripes > $O/pmc_gram.txt 2>&1
bash scripts/pmc_gram.sh k_gene_moments > $O/pmc_moments.txt 2>&1
bash scripts/pmc_kernel.sh k_spmm_rows > $O/pmc_spmm_rows.txt 2>&1
bash scripts/pmc_kernel.sh k_jacobi_eig2 > $O/pmc_jacobi.txt 2>&1
tail -1 $O/kt_bench.json | cut -c1-300
head -30 $O/rocprof_c3_table.md
rm -rf $O/kt $O/pf $O/pw gpurun_out/pmcg*
