# shape-env generator with the index walker + HAND threshold at 2^21: parity (shape env vs numpy twin, cfg4 tests) and A/B
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-r04o}
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_eval.py tests/test_cli_gpu.py -x -q -m gpu -k "shape or full_size or eval" > $O/parity.txt 2>&1; tail -3 $O/parity.txt
ab() { label=$1; shift
  for lib in "" $GRAFT_REPO_ROOT/cleanmarl_amd/libcleanmarl_hip_ab.so "" $GRAFT_REPO_ROOT/cleanmarl_amd/libcleanmarl_hip_ab.so; do
    CM_LIB_PATH=$lib timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); p=o['phase_ms']
print('$label', 'old' if '$lib' else 'new', 'ms/step %.4f rollout %.4f value %.3f actor_k %.4f frac %.4f' % (o['ms_per_step'], p['rollout'], p['value_pass_scan'], o['kernel_ms']['actor_fwd_bwd'], o['roofline']['frac']))"
  done; }
( ab cfg4 --workload cfg4; ab cfg3_2048 --envs 2048; ab cfg4_256 --workload cfg4 --envs 256 ) 2>/dev/null | tee $O/ab.txt
