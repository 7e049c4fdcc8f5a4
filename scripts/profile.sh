# The round's profile of the bench command on a GPU box (run through gpurun; everything lands in $O, copy what is to be judged into
# profiles/): rocprofv3 kernel trace + stats, the timeline of the last traced step, FETCH_SIZE / WRITE_SIZE traffic (separate PMC
# passes — never combined with another trace domain), and the PMC tables of the largest kernels (scripts/pmc_kernel.sh).
#   usage: bash scripts/profile.sh [out-dir]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=${1:-gpurun_out/prof}; mkdir -p $O
B="python bench.py --gpus 1 --steps 3 --warmup 1 --lean"
rm -rf $O/kt $O/pf $O/pw
rocprofv3 --kernel-trace --stats -d $O/kt -o c3 -- $B > $O/kt_bench.json 2> $O/kt.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o c3 -- $B > /dev/null 2> $O/pf.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o c3 -- $B > /dev/null 2> $O/pw.log
python profiles/summarize_rocpd.py $O/kt/c3_results.db > $O/rocprof_c3.md
python profiles/make_traffic.py $O/pf/c3_results.db $O/pw/c3_results.db c3 auto > $O/traffic_c3.json
python profiles/timeline_rocpd.py $O/kt/c3_results.db k_row_sum > $O/timeline_c3.md
for k in k_gram_stripes k_gene_moments k_spmm_rows k_jacobi_eig2 k_rowcount_list k_bucket; do
  bash scripts/pmc_kernel.sh $k > $O/pmc_$k.md 2>&1
done
tail -1 $O/kt_bench.json | cut -c1-300
head -40 $O/rocprof_c3.md
rm -rf $O/kt $O/pf $O/pw
