set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
python tools/probes/rollout_tile_ab.py > $O/rollout_tile_ab.txt 2>&1
cat $O/rollout_tile_ab.txt
python -m pytest tests/test_hip_parity.py -x -q -k "fused_rollout or rollout_edge" 2>&1 | tail -5
