# rollout A/B (default build vs libcleanmarl_hip_ab.so) + rollout parity tests + fresh phase profiles of both rollout tilings
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-r04g}
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_eval.py tests/test_coma_gpu.py -x -q -m gpu -k "rollout or eval or coma" > $O/parity.txt 2>&1; tail -4 $O/parity.txt
ab() { # label, args...
  label=$1; shift
  for lib in "" $GRAFT_REPO_ROOT/cleanmarl_amd/libcleanmarl_hip_ab.so "" $GRAFT_REPO_ROOT/cleanmarl_amd/libcleanmarl_hip_ab.so; do
    CM_LIB_PATH=$lib timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); p=o['phase_ms']
print('$label', 'old' if '$lib' else 'new', 'ms/step %.4f rollout %.4f value %.3f actor_stream %.3f actor_k %.4f critic_k %.4f frac %.4f' % (o['ms_per_step'], p['rollout'], p['value_pass_scan'], p['update_actor_stream'], o['kernel_ms']['actor_fwd_bwd'], o['kernel_ms']['critic_fwd_bwd'], o['roofline']['frac']))"
  done
}
(
ab cfg3
ab cfg3_2048 --envs 2048
ab cfg3_512 --envs 512
) 2>/dev/null | tee $O/ab.txt
python cleanmarl_amd/build.py --prof > /dev/null 2>&1
python tools/phase_prof.py rollout > $O/phase_rollout64s.txt 2>&1; cat $O/phase_rollout64s.txt
python tools/phase_prof.py rollout 512 8 > $O/phase_rollout16s.txt 2>&1; cat $O/phase_rollout16s.txt
