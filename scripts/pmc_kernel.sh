# PMC passes (one rocprofv3 run per counter group, --kernel-trace only) over a short bench run, for one kernel:
#   scripts/pmc_kernel.sh <kernel-substring> [bench flags ...]        e.g.  scripts/pmc_kernel.sh k_spmm_t --solver 2
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
K=$1; shift
B="python bench.py --steps 1 --warmup 1 --lean $*"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmck$i
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmck$i -o g -- $B > /tmp/pmck$i.log 2>&1
  python profiles/summarize_pmc.py /tmp/pmck$i/g_results.db $K || tail -5 /tmp/pmck$i.log
done
