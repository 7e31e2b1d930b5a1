# share experiments: grids of the split critic's kernels, forced one-pass critic
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r04c
mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { # label, env...
  label=$1; shift
  for i in 1 2; do
    env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --envs 512 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); p=o['phase_ms']
print('$label', 'ms/step %.4f rollout %.3f value %.3f actor_stream %.3f critic_stream %.3f actor_k %.3f critic_k %.3f' % (o['ms_per_step'], p['rollout'], p['value_pass_scan'], p['update_actor_stream'], p['update_critic_stream'], o['kernel_ms']['actor_fwd_bwd'], o['kernel_ms']['critic_fwd_bwd']))"
  done
}
(
run base A=1
run dw0_256 CM_DW0_GRID=256
run dw0_128 CM_DW0_GRID=128
run dw0_64 CM_DW0_GRID=64
run tg_256 CM_TRAIN_GRID=256
run tg_128 CM_TRAIN_GRID=128
run both_128 CM_DW0_GRID=128 CM_TRAIN_GRID=128
run both_256 CM_DW0_GRID=256 CM_TRAIN_GRID=256
run dw0_128_b8 CM_DW0_GRID=128 CM_DW0_BATCH=8
run fusedcritic CM_CRITIC_SCHEDULE=fused
run sched1 CM_CRITIC_OVERLAP=1
run sched1_both128 CM_CRITIC_OVERLAP=1 CM_DW0_GRID=128 CM_TRAIN_GRID=128
) | tee $O/share_grids.txt
