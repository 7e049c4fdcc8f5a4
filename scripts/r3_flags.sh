# A/B of compile-time switches: scripts/r3_flags.sh "-DFOO" "-DBAR -DBAZ" ...  ("-" = no flags); rebuilds pca.o each time on the GPU box
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for f in "$@"; do
  if [ "$f" = "-" ]; then f=""; fi
  touch singlerust_amd/csrc/pca.hip
  SRX_EXTRA_FLAGS="$f" python -m singlerust_amd.build > /dev/null 2>&1 || echo BUILD FAILED
  python bench.py --gpus 1 --steps ${STEPS:-5} --warmup 2 --lean $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$f]', 'ms/step', round(d['ms_per_step'],3), 'res', d['config']['pca_residual'], 'it', d['config']['subspace_iterations'][:2], {k:round(x,3) for k,x in d['kernel_ms_per_step'].items()})"
done
