# A/B of an experimental build of the library (singlerust_amd/lib/libsrx_hip_exp.so, built by hand with -D switches)
# against the regular one on the GPU box: bench line summary for each, then the PCA parity tests on the experiment.
cd $GRAFT_REPO_ROOT
L=singlerust_amd/lib
summ() { python - "$1" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d=json.loads(line); r=d["roofline"]
        print(f'{sys.argv[1]}: {d["ms_per_step"]:.2f} ms/step  gram {r["avg_ms"]:.3f} ms  resid {d["config"]["pca_residual"]:.2e} iters {d["config"]["subspace_iterations"]}')
PY
}
python bench.py $AB_ARGS --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/ab_base.log 2>&1; summ gpurun_out/ab_base.log
cp $L/libsrx_hip.so /tmp/base.so; cp $L/libsrx_hip_exp.so $L/libsrx_hip.so
python bench.py $AB_ARGS --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/ab_exp.log 2>&1; summ gpurun_out/ab_exp.log
if [ -n "$AB_TESTS" ]; then timeout 900 python -m pytest tests/test_pca_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -8; fi
cp /tmp/base.so $L/libsrx_hip.so
