cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gru_wide.py -m gpu -x -q -k "64_actions or refuse" 2>&1 | tail -15
