# Regenerates the measurement artefacts of a round on a gpurun box:  bash tools/refresh_profiles.sh r02   (outputs: gpurun_out/<tag>/, copy
# the summaries to profiles/<tag>_*).  Needs libcleanmarl_hip.so and, for the phase profiles, libcleanmarl_hip_prof.so (python -m cleanmarl_amd.build --prof).
set -x
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# ---- the driver's command: headline line with every extra leg (other workloads, shares, fair CPU baseline)
python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
# ---- per-kernel durations of the same timed region (extras off: they would mix other workloads into the averages)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
# ---- PMC passes, one counter set per run (MI355X_MICROARCH.md)
mkdir -p $O/pmc
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc/pmc_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc/pmc_write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc/pmc_mfma -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc --emit $O/pmc_dominant_kernel.json "profiles/${TAG}_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES, separate passes; FETCH_SIZE x2 per MI355X_MICROARCH.md)" > $O/pmc_summary.txt 2>&1
find $O/pmc -name "*.csv" -size +2M -delete
# ---- per-kernel durations of the per-GPU share (512 envs) and of the other BASELINE configs
for w in "cfg3 --envs 512" "cfg2" "cfg4" "cfg5"; do
  n=$(echo $w | tr -d ' -')
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k_$n -- python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_$n.json 2>/dev/null
  cp $(find /tmp/k_$n -name "*kernel_stats.csv" | head -1) $O/${n}_kernel_stats.csv
done
# ---- phase profiles (s_memtime per phase, -DCM_PHASE_PROF build)
python $R/tools/phase_prof.py actor > $O/phase_actor.txt 2>&1
python $R/tools/phase_prof.py critic > $O/phase_critic.txt 2>&1
python $R/tools/phase_prof.py rollout > $O/phase_rollout.txt 2>&1
python $R/tools/phase_prof.py rollout 512 8 > $O/phase_rollout16s.txt 2>&1
python $R/tools/phase_prof.py act > $O/phase_act.txt 2>&1
CM_PROF_WARMUP=50 python $R/tools/phase_prof.py gru > $O/phase_gru.txt 2>&1
python $R/tools/phase_prof.py grurollout > $O/phase_grurollout.txt 2>&1
python $R/tools/bench_configs.py > $O/configs_learner.txt 2>&1
# ---- widened rows: COMA, host-env plumbing, layered schedule
python $R/tools/bench_coma.py > $O/coma_bench.json 2> $O/coma.err
python $R/tools/bench_host_env.py 256 8 128 > $O/host_env.txt 2>&1
python $R/tools/bench_wide.py 2>&1 | grep -v amdgpu.ids > $O/wide_schedule.txt
# ---- opt-in compensated-bf16 arithmetic (NOT the default)
CM_MFMA=bf16x3 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_bf16x3.json 2>/dev/null
CM_MFMA=bf16 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_bf16.json 2>/dev/null
# ---- steady-state rate of the product surface (driver.run through the CLI), not only of bench.py's inner loop
python $R/tools/cli_steady_state.py > $O/cli_steady_state.txt 2>&1
ls -la $O
# ---- kernel timelines of one steady-state iteration (two queues): the 512-env share of config 3, config 2
for w in "cfg3 --envs 512" "cfg2"; do
  n=$(echo $w | tr -d ' -')
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$n -- python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python $R/tools/trace_timeline.py $(find /tmp/kt_$n -name "*kernel_trace.csv" | head -1) k_ro 3 > $O/timeline_$n.txt 2>&1
done
# ---- round 5: per-workgroup span of the actor pass (cm_clock_probe: shader clock, ramp and tail, who finishes when) at both sizes, equal and shipped split
python $R/tools/probes/wg_span.py 4096 2>&1 | grep -v amdgpu.ids > $O/wg_span.txt
CM_TILE_SPLIT=50 python $R/tools/probes/wg_span.py 4096 2>&1 | grep -v amdgpu.ids > $O/wg_span_equal_split.txt
python $R/tools/probes/wg_span.py 512 2>&1 | grep -v amdgpu.ids > $O/wg_span_envs512.txt
# ---- the whole GPU suite at these sources
cd $R && (time timeout 3000 python -m pytest tests -q -m gpu) > $O/gputests.txt 2>&1; tail -5 $O/gputests.txt
