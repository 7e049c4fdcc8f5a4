# per-kernel durations under compile-time switches (rocprofv3 kernel trace): scripts/r3_flags_prof.sh KERNEL_SUBSTR "-DFOO" ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
K=$1; shift
for f in "$@"; do
  if [ "$f" = "-" ]; then f=""; fi
  touch singlerust_amd/csrc/pca.hip
  SRX_EXTRA_FLAGS="$f" python -m singlerust_amd.build > /dev/null 2>&1 || echo BUILD FAILED
  rm -rf /tmp/fp; rocprofv3 --kernel-trace --stats -d /tmp/fp -o t --output-format csv -- python bench.py --gpus 1 --steps ${STEPS:-1} --warmup 0 --lean $BENCH_ARGS > /dev/null 2>&1
  echo "[$f]"; grep -h "$K" /tmp/fp/*kernel_stats.csv | cut -c1-200 | head -5
done
