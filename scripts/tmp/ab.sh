cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py --gpus 1 --steps 8 --warmup 2 --lean 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(d['ms_per_step'],3), 'spmm', round(d['kernel_ms_per_step']['spmm_fwd'],3))"; }
for rep in 1 2; do
cp scripts/tmp/pca_head.hip singlerust_amd/csrc/pca.hip; python -m singlerust_amd.build > /dev/null 2>&1 || echo BUILD FAILED
run head
cp scripts/tmp/pca_new.hip singlerust_amd/csrc/pca.hip; SRX_EXTRA_FLAGS="-DSPMM_KSUB=2" python -m singlerust_amd.build > /dev/null 2>&1 || echo BUILD FAILED
run new_k2
SRX_FWD_NOPERM=1 run new_k2_noperm
touch singlerust_amd/csrc/pca.hip; SRX_EXTRA_FLAGS="-DSPMM_KSUB=1" python -m singlerust_amd.build > /dev/null 2>&1 || echo BUILD FAILED
run new_k1
SRX_FWD_NOPERM=1 run new_k1_noperm
done
