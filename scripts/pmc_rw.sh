# fabric read / write bytes of one kernel (development helper).  usage: pmc_rw.sh kernel-substring [bench args...]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
K=$1; shift
B="python bench.py --steps 1 --warmup 1 --lean $@"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcr$i
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmcr$i -o g -- $B > gpurun_out/pmcr$i.log 2>&1
  python profiles/summarize_pmc.py gpurun_out/pmcr$i/g_results.db $K || tail -5 gpurun_out/pmcr$i.log
  rm -rf gpurun_out/pmcr$i
done
