cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "more_than_32" 2>&1 | tail -30
