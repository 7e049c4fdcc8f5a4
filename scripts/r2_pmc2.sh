cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2p
bash scripts/pmc_gram.sh k_gram_stripes > gpurun_out/r2p/pmc_gram.txt 2>&1
bash scripts/pmc_gram.sh k_gene_moments > gpurun_out/r2p/pmc_moments.txt 2>&1
rm -rf gpurun_out/pmcg*
grep -v "^\[" gpurun_out/r2p/pmc_gram.txt | head -30
