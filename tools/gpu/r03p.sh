R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03p
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_cli_gpu.py -q -x -s -k "bf16 or learning_signal_on" 2>&1 | grep -v amdgpu | tail -12
for m in "" bf16x3 bf16; do
CM_MFMA=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg3 mfma=$m', 'ms_per_step %.4f' % d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items()}, 'actor', round(d['kernel_ms']['actor_fwd_bwd'],4), 'critic', round(d['kernel_ms']['critic_fwd_bwd'],4), d['dtype'][:12])" | tee -a $O/bench_mfma.txt
done
