cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "dw0_batch or split_schedule" 2>&1 | tail -15
