cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py --gpus 1 --steps 8 --warmup 2 --lean $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(d['ms_per_step'],3), 'spmm', round(d['kernel_ms_per_step']['spmm_fwd'],3), 'res', d['config']['pca_residual'])"; }
for f in "-DSPMM_OUTQ=1 -DSPMM_KSUB=2" "-DSPMM_OUTQ=2 -DSPMM_KSUB=2" "-DSPMM_OUTQ=4 -DSPMM_KSUB=2" "-DSPMM_OUTQ=2 -DSPMM_KSUB=1" "-DSPMM_OUTQ=4 -DSPMM_KSUB=1" "-DSPMM_OUTQ=1 -DSPMM_KSUB=2"; do
touch singlerust_amd/csrc/pca.hip; SRX_EXTRA_FLAGS="$f" python -m singlerust_amd.build > /dev/null 2>&1 || echo BUILD FAILED
SRX_FWD_NARROW=1 run "narrow[$f]"
run "wide[$f]"
done
