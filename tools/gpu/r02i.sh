R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02i
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "gru" 2>&1 | tail -15 > $O/pytest_gru.txt
cat $O/pytest_gru.txt
python bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/cfg5_v2.json 2>> $O/err.txt
CM_PROF_WARMUP=50 python tools/phase_prof.py gru > $O/phase_gru.txt 2>&1; cat $O/phase_gru.txt
