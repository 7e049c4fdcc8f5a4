# L2 hit / fabric traffic of one kernel (development helper).  usage: pmc_l2.sh [kernel-substring] [ENV=..]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
K=${1:-k_gram_stripes}; shift
B="python bench.py --steps 1 --warmup 1 --lean"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcl$i
  env "$@" rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmcl$i -o g -- $B > gpurun_out/pmcl$i.log 2>&1
  python profiles/summarize_pmc.py gpurun_out/pmcl$i/g_results.db $K || tail -5 gpurun_out/pmcl$i.log
done
