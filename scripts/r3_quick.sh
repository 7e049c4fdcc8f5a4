# round 3 helper: PCA parity tests, then the lean bench line under a list of env settings
#   scripts/r3_quick.sh "A=1" "B=2 C=3" ...     (BENCH_ARGS adds bench flags; TESTS overrides the test selection; "-" = no env)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=${TESTS:-"tests/test_pca_gpu.py tests/test_configs_gpu.py"}
if [ "$T" != "none" ]; then python -m pytest $T -m gpu -x -q 2>&1 | tail -4; fi
B="python bench.py --gpus 1 --steps ${STEPS:-10} --warmup 3 --lean $BENCH_ARGS"
for v in "$@"; do
  if [ "$v" = "-" ]; then v="SRX_NOP=1"; fi
  env $v $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'ms/step', round(d['ms_per_step'],3), 'res', d['config']['pca_residual'], 'it', d['config']['subspace_iterations'][:2], {k:round(x,3) for k,x in d['kernel_ms_per_step'].items()})"
done
