R=$GRAFT_REPO_ROOT
cd $R
timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -6
