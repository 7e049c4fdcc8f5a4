# PMC passes over a short bench run, Gram kernel only (development helper).  usage: pmc_gram.sh [kernel-substring]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
K=${1:-k_gram_stripes}
B="python bench.py --steps 1 --warmup 1 --lean"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcg$i
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmcg$i -o g -- $B > gpurun_out/pmcg$i.log 2>&1
  python profiles/summarize_pmc.py gpurun_out/pmcg$i/g_results.db $K || tail -5 gpurun_out/pmcg$i.log
done
