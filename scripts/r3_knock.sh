# knock-out builds: scripts/r3_knock.sh "-DFOO" ...   prints the per-class kernel times of one lean bench step per build
# (results are wrong by construction: only the class times matter; a failing solve still prints what the profiler saw)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for f in "$@"; do
  if [ "$f" = "-" ]; then f=""; fi
  touch singlerust_amd/csrc/pca.hip singlerust_amd/csrc/genes.hip
  SRX_EXTRA_FLAGS="$f" python -m singlerust_amd.build > /dev/null 2>&1 || echo BUILD FAILED
  python - <<PY 2>&1 | tail -3
import json, subprocess, sys
p = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "4", "--warmup", "1", "--lean"] + "$BENCH_ARGS".split(), capture_output=True, text=True)
try:
    d = json.loads(p.stdout.strip().splitlines()[-1])
    print("[$f]", "ms/step", round(d["ms_per_step"], 3), {k: round(x, 3) for k, x in d["kernel_ms_per_step"].items()})
except Exception as e:
    print("[$f] bench failed:", p.stderr.strip().splitlines()[-3:])
PY
done
