# knock-out builds: scripts/r3_knock.sh "-DFOO" ...   rocprofv3 kernel stats of one lean bench run per build (the results are
# wrong by construction and the solve may fail: only the kernel durations matter).  PAT = kernel name pattern to print.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for f in "$@"; do
  if [ "$f" = "-" ]; then f=""; fi
  touch singlerust_amd/csrc/pca.hip singlerust_amd/csrc/genes.hip
  SRX_EXTRA_FLAGS="$f" python -m singlerust_amd.build > /dev/null 2>&1 || echo BUILD FAILED
  echo "[$f]"; STEPS=2 bash scripts/r3_kstats.sh "${PAT:-k_}" | sort -t' ' -k1,1 | grep -v "calls     1 avg       [0-9]\." | head -${LINES_MAX:-6}
done
touch singlerust_amd/csrc/pca.hip singlerust_amd/csrc/genes.hip; python -m singlerust_amd.build > /dev/null 2>&1
