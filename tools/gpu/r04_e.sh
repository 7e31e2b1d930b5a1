# A/B on one box: default build vs cleanmarl_amd/libcleanmarl_hip_ab.so (the previous variant of the change under test); parity first
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-r04e}
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_fused_step.py tests/test_coma_gpu.py -x -q -m gpu > $O/parity.txt 2>&1; tail -4 $O/parity.txt
ab() { # label, args...
  label=$1; shift
  for lib in "" $GRAFT_REPO_ROOT/cleanmarl_amd/libcleanmarl_hip_ab.so "" $GRAFT_REPO_ROOT/cleanmarl_amd/libcleanmarl_hip_ab.so; do
    CM_LIB_PATH=$lib timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); p=o['phase_ms']
print('$label', 'old' if '$lib' else 'new', 'ms/step %.4f rollout %.3f value %.3f actor_stream %.3f actor_k %.4f critic_k %.4f frac %.4f' % (o['ms_per_step'], p['rollout'], p['value_pass_scan'], p['update_actor_stream'], o['kernel_ms']['actor_fwd_bwd'], o['kernel_ms']['critic_fwd_bwd'], o['roofline']['frac']))"
  done
}
(
ab cfg3
ab cfg3_512 --envs 512
ab cfg2 --workload cfg2
ab cfg4 --workload cfg4
) 2>/dev/null | tee $O/ab.txt
if [ -n "$PROF" ]; then python cleanmarl_amd/build.py --prof > /dev/null 2>&1; python tools/phase_prof.py actor > $O/phase_actor.txt 2>&1; cat $O/phase_actor.txt; fi
