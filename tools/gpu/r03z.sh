R=$GRAFT_REPO_ROOT
cd /tmp
for ov in 0 1 2; do for cs in auto split; do for db in auto 4; do
  CM_CRITIC_OVERLAP=$ov CM_CRITIC_SCHEDULE=$cs CM_DW0_BATCH=$db python $R/bench.py --workload cfg3 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; b=json.loads(sys.stdin.read()); print('cfg3 overlap=$ov critic=$cs dw0=$db', round(b['ms_per_step'],4), {k:round(v,3) for k,v in b['phase_ms'].items()}, b['kernel_ms'])"
done; done; done
for ov in 0 1 2; do
  CM_CRITIC_OVERLAP=$ov python $R/bench.py --workload cfg3 --envs 2048 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; b=json.loads(sys.stdin.read()); print('cfg3envs2048 overlap=$ov', round(b['ms_per_step'],4), {k:round(v,3) for k,v in b['phase_ms'].items()})"
  CM_CRITIC_OVERLAP=$ov python $R/bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; b=json.loads(sys.stdin.read()); print('cfg4 overlap=$ov', round(b['ms_per_step'],4), {k:round(v,3) for k,v in b['phase_ms'].items()})"
done
