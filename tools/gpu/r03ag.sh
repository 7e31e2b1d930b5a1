cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fused_step.py -m gpu -x -q 2>&1 | tail -5
