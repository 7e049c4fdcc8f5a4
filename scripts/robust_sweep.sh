# parameter sweep of the pipeline looking for failures (NOCONV, errors) — development helper
cd $GRAFT_REPO_ROOT
run() {
  out=$(python bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1)
  echo "$* :: $(echo "$out" | python -c "
import json,sys
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('ok', round(d['ms_per_step'],2), 'ms', d['config']['pca_solver'], d['config']['subspace_iterations'], '%.2e' % d['config']['pca_residual'])
except Exception: print('FAIL', l[:200])
")"
}
for hvg in 300 1000 4000 8000; do for npc in 5 30 56; do run --config c2 --hvg $hvg --npc $npc; done; done
run --config c2 --hvg 2000 --npc 100
run --config c2 --hvg 2000 --npc 200
run --config c2 --hvg 2000 --npc 50 --storage f64
run --config c2 --hvg 6000 --npc 50 --storage f64
run --config c2 --hvg 10000 --npc 20
run --config c2 --hvg 10000 --npc 20 --solver 1
run --config c3 --hvg 500 --npc 10
run --config c3 --hvg 4000 --npc 50
run --config c3 --cells 50000 --hvg 2000 --npc 50
run --config c3 --cells 5000 --hvg 2000 --npc 50
run --config c3 --cells 500 --hvg 2000 --npc 50
