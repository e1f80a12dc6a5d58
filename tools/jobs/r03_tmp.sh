cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "groups_above_64" 2>&1 | grep -E "AssertionError|assert |Error" | head -10
