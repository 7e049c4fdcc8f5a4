cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for f in tests/test_pca_gpu.py tests/test_configs_gpu.py tests/test_backed_gpu.py; do
timeout 900 python -X faulthandler -m pytest $f -x -v -m gpu 2>&1 | grep -E "PASSED|FAILED|ERROR|Fatal|File .*tests|passed|failed" | tail -6
done
