# one-off: build with -DSPMM_TIMING and run one bench step (prints per-wave phase times of k_spmm_rows)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
touch singlerust_amd/csrc/pca.hip
SRX_EXTRA_FLAGS="-DSPMM_TIMING $XF" python -m singlerust_amd.build > /dev/null 2>&1 || echo BUILD FAILED
python bench.py --gpus 1 --steps 1 --warmup 0 --lean 2>&1 | grep "spmm timing" | head -40
