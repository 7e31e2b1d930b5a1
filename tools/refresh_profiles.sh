# Regenerates EVERY measurement artefact of a round at the sources of ONE commit, on a gpurun box:
#     gpurun --timeout 3000 -- 'HEAD=<git short hash> bash tools/refresh_profiles.sh r06'
# Outputs: gpurun_out/<tag>/ ; tools/stamp_profiles.py then copies them to profiles/<tag>_* with the source hash of the library they were
# taken on (cleanmarl_amd/build.py::source_hash) and the commit written into each (json: a key; text: a comment line; csv: MANIFEST only).
# Needs libcleanmarl_hip.so and, for the phase profiles, libcleanmarl_hip_prof.so (python -m cleanmarl_amd.build --prof) built from the same sources.
set -x
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
export TAG
cd /tmp && export TMPDIR=/tmp
RUN="bash $R/tools/gpu/run.sh"
# ---- the driver's command: headline line with every extra leg (solo leg, other workloads, shares, CPU baseline)
python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
# ---- per-kernel durations of the same timed region, shipped schedule (two streams: in-iteration durations are co-residency figures) ...
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
# ---- ... and with every launch alone on the device (CM_CRITIC_OVERLAP=0: the one-stream schedule of rounds 2 - 5): the kernels' own durations,
# what roofline.launch_ms of the solo leg must agree with
CM_CRITIC_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks1 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_under_rocprof_one_stream.json 2>/dev/null
cp $(find /tmp/ks1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_one_stream.csv
# ---- PMC passes, one counter set per run (MI355X_MICROARCH.md; counter collection serialises the launches: per-kernel figures are solo figures)
$RUN pmc
# ---- per-kernel durations of the per-GPU share (512 envs) and of the other BASELINE configs
for w in "cfg3 --envs 512" "cfg2" "cfg4" "cfg5"; do
  n=$(echo $w | tr -d ' -')
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k_$n -- python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_$n.json 2>/dev/null
  cp $(find /tmp/k_$n -name "*kernel_stats.csv" | head -1) $O/${n}_kernel_stats.csv
done
# ---- phase profiles (s_memtime per phase, -DCM_PHASE_PROF build)
python $R/tools/phase_prof.py actor > $O/phase_actor.txt 2>&1
python $R/tools/phase_prof.py critic > $O/phase_critic.txt 2>&1
python $R/tools/phase_prof.py rollout > $O/phase_rollout64s.txt 2>&1
python $R/tools/phase_prof.py rollout 512 8 > $O/phase_rollout16s.txt 2>&1
python $R/tools/phase_prof.py act > $O/phase_act.txt 2>&1
CM_PROF_WARMUP=50 python $R/tools/phase_prof.py gru > $O/phase_gru.txt 2>&1
python $R/tools/phase_prof.py grurollout > $O/phase_grurollout.txt 2>&1
python $R/tools/bench_configs.py > $O/configs_learner.txt 2>&1
# ---- widened rows: COMA (64-wide and the reference's default 128-wide critic), host-env plumbing, layered schedule
python $R/tools/bench_coma.py > $O/coma_bench.json 2> $O/coma.err
python $R/tools/bench_coma.py --critic-hidden 128 > $O/coma128_bench.json 2> $O/coma128.err
python $R/tools/bench_host_env.py 256 8 128 > $O/host_env.txt 2>&1
python $R/tools/bench_wide.py 2>&1 | grep -v amdgpu.ids > $O/wide_schedule.txt
# ---- opt-in compensated-bf16 arithmetic (NOT the default)
CM_MFMA=bf16x3 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_bf16x3.json 2>/dev/null
CM_MFMA=bf16 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_bf16.json 2>/dev/null
# ---- steady-state rate of the product surface (driver.run through the CLI), not only of bench.py's inner loop
python $R/tools/cli_steady_state.py > $O/cli_steady_state.txt 2>&1
# ---- kernel timelines of one steady-state iteration (two queues): config 3 at full size, its 512-env share, config 2
for w in "cfg3" "cfg3 --envs 512" "cfg2"; do
  n=$(echo $w | tr -d ' -')
  rm -rf /tmp/kt_$n
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$n -- python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extras --solo-launches 0 > /dev/null 2>&1
  python $R/tools/trace_timeline.py $(find /tmp/kt_$n -name "*kernel_trace.csv" | head -1) k_ro 3 > $O/timeline_$n.txt 2>&1
  # launches of an iteration that are not this library's kernels (torch copies / fills): counted per name
  python - $(find /tmp/kt_$n -name "*kernel_trace.csv" | head -1) > $O/foreign_launches_$n.txt <<'PYEOF'
import collections, csv, sys
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "at::native" in k or "rocclr" in k or "Fill" in k:
        n[k[:110]] += 1
print("launches that are not libcleanmarl_hip kernels, whole run (1 set-up pass that allocates and fills the buffers + 13 iterations; a count that does not")
print("grow with the iterations is set-up; __amd_rocclr_copyBuffer = the asynchronous copy of an update's record buffer to pinned host memory):")
for k, v in n.most_common():
    print(f"  {v:5d}  {k}")
PYEOF
done
# ---- per-workgroup span of the actor pass (cm_clock_probe: shader clock, ramp and tail, who finishes when) at both sizes, equal and shipped split
python $R/tools/probes/wg_span.py 4096 2>&1 | grep -v amdgpu.ids > $O/wg_span.txt
CM_TILE_SPLIT=50 python $R/tools/probes/wg_span.py 4096 2>&1 | grep -v amdgpu.ids > $O/wg_span_equal_split.txt
python $R/tools/probes/wg_span.py 512 2>&1 | grep -v amdgpu.ids > $O/wg_span_envs512.txt
# ---- the whole GPU suite at these sources, and the observed maxima of its parity metrics (tests/parity.py)
cd $R && (time timeout 3000 python -m pytest tests -q -m gpu) > $O/gputests.txt 2>&1; tail -5 $O/gputests.txt
cp $R/gpurun_out/parity_observed.txt $O/parity_observed.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python - > $O/MANIFEST.txt <<PYEOF
import os, sys
sys.path.insert(0, "$R")
from cleanmarl_amd.build import source_hash
print("tag $TAG  commit ${HEAD:-unknown}  source_hash", source_hash())
for f in sorted(os.listdir("$O")):
    p = os.path.join("$O", f)
    if os.path.isfile(p):
        print(f"{os.path.getsize(p):10d}  {f}")
PYEOF
cat $O/MANIFEST.txt
