# the whole GPU suite, as the driver runs it at round end
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-tests}
mkdir -p $O
cd $GRAFT_REPO_ROOT
(time timeout 3000 python -m pytest tests -q -m gpu) > $O/gpu_tests.txt 2>&1; tail -15 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
