cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_edit_loop.py tests/test_gpu_round3.py -m gpu -x -q -s 2>&1 | tail -30


