#!/bin/bash
# tools/gpu/run.sh -- the ONE runner for GPU-box recipes (replaces the per-experiment r0N*.sh scripts of rounds 1 - 4; those are in the
# git history).  Called through gpurun from the repo root, e.g.
#     gpurun --timeout 1200 -- 'TAG=r05b bash tools/gpu/run.sh tests bench'
#     gpurun --timeout 900  -- 'TAG=r05c bash tools/gpu/run.sh "ab CM_CRITIC_OVERLAP 1 2 -- --workload cfg3 --envs 512" "timeline --workload cfg3 --envs 512"'
# Every argument is one recipe (quote recipes that take arguments); outputs go to gpurun_out/$TAG/.  Recipes:
#   tests [pytest args]          the GPU suite as the driver runs it (+ smoke); default: tests -m gpu
#   bench [bench.py args]        the driver's command: python bench.py --steps 20 --warmup 5 [args]            -> bench[_<args>].json
#   quick [bench.py args]        bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 [args]            -> quick_<args>.json
#   stats [bench.py args]        rocprofv3 --kernel-trace --stats around `quick`                                -> <args>_kernel_stats.csv
#   timeline [bench.py args]     rocprofv3 --kernel-trace around `quick` + tools/trace_timeline.py              -> timeline_<args>.txt
#   ab VAR v1 v2 .. -- [args]    `quick [args]` once per value of the environment variable VAR                  -> ab_<VAR>.txt (ms per step, actor ms)
#   phase <kernel> [E A]         tools/phase_prof.py on the -DCM_PHASE_PROF build                               -> phase_<kernel>.txt
#   pmc [bench.py args]          the three separate --pmc passes + tools/pmc_summary.py (MI355X_MICROARCH.md)   -> pmc_summary.txt, pmc_dominant_kernel.json
#   py <script> [args]           python <script> [args]                                                         -> py_<script>.txt
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${TAG:-run}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
slug() { echo "$*" | tr -cd 'A-Za-z0-9_=.' | cut -c1-60; }
quick() { python "$R/bench.py" --steps "${STEPS:-20}" --warmup "${WARMUP:-5}" --no-cpu-baseline --no-extras "$@"; }

for recipe in "$@"; do
  set -- $recipe
  cmd=$1; shift
  echo "=== $cmd $*"
  case $cmd in
    tests)
      cd "$R"
      if [ $# -eq 0 ]; then set -- tests -m gpu; fi
      (time timeout "${TEST_TIMEOUT:-3000}" python -m pytest -q "$@") > "$O/gpu_tests$(slug "$*" | sed 's/^tests-mgpu$//').txt" 2>&1
      tail -12 "$O"/gpu_tests*.txt
      python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
      cd /tmp ;;
    bench)
      python "$R/bench.py" --steps "${STEPS:-20}" --warmup "${WARMUP:-5}" "$@" > "$O/bench$(slug "$*").json" 2> "$O/bench$(slug "$*").err"
      tail -c 2500 "$O/bench$(slug "$*").json" ;;
    quick)
      quick "$@" > "$O/quick_$(slug "$*").json" 2> "$O/quick_$(slug "$*").err"
      python - "$O/quick_$(slug "$*").json" <<'EOF'
import json, sys
o = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("ms_per_step %.4f  actor %.4f critic %.4f  frac %.4f  phases %s" % (o["ms_per_step"], o["kernel_ms"]["actor_fwd_bwd"], o["kernel_ms"]["critic_fwd_bwd"], o["roofline"]["frac"], {k: round(v, 4) for k, v in o["phase_ms"].items()}))
EOF
      ;;
    stats)
      d=/tmp/ks_$(slug "$*"); rm -rf "$d"
      rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python "$R/bench.py" --steps "${STEPS:-20}" --warmup "${WARMUP:-5}" --no-cpu-baseline --no-extras "$@" > "$O/bench_under_rocprof_$(slug "$*").json" 2>/dev/null
      cp "$(find "$d" -name '*kernel_stats.csv' | head -1)" "$O/$(slug "$*")_kernel_stats.csv"
      head -12 "$O/$(slug "$*")_kernel_stats.csv" | cut -c1-220 ;;
    timeline)
      d=/tmp/kt_$(slug "$*"); rm -rf "$d"
      rocprofv3 --kernel-trace --output-format csv -d "$d" -- python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" > /dev/null 2>&1
      python "$R/tools/trace_timeline.py" "$(find "$d" -name '*kernel_trace.csv' | head -1)" "${ANCHOR:-k_ro}" 3 > "$O/timeline_$(slug "$*").txt" 2>&1
      cat "$O/timeline_$(slug "$*").txt" ;;
    ab)
      var=$1; shift
      vals=()
      while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
      [ $# -gt 0 ] && shift
      for v in "${vals[@]}"; do
        for rep in $(seq 1 "${REPS:-1}"); do
          line=$(env "$var=$v" python "$R/bench.py" --steps "${STEPS:-20}" --warmup "${WARMUP:-5}" --no-cpu-baseline --no-extras "$@" 2>/dev/null | grep '^{' | python -c "
import json, sys
o = json.loads(sys.stdin.readline())
print('ms_per_step %.4f actor %.4f critic %.4f frac %.4f phases %s' % (o['ms_per_step'], o['kernel_ms']['actor_fwd_bwd'], o['kernel_ms']['critic_fwd_bwd'], o['roofline']['frac'], {k: round(v, 3) for k, v in o['phase_ms'].items()}))")
          echo "$var=$v [$*] $line" | tee -a "$O/ab_$var.txt"
        done
      done ;;
    phase)
      python "$R/tools/phase_prof.py" "$@" > "$O/phase_$(slug "$*").txt" 2>&1
      grep -v amdgpu.ids "$O/phase_$(slug "$*").txt" ;;
    pmc)
      mkdir -p "$O/pmc"
      for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
        n=$(echo $c | cut -d' ' -f1 | tr 'A-Z' 'a-z' | sed 's/_size//; s/sq_valu_mfma_busy_cycles/mfma/')
        rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc/pmc_$n" -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > /dev/null 2>&1
      done
      python "$R/tools/pmc_summary.py" "$O/pmc" --emit "$O/pmc_dominant_kernel.json" "profiles/${TAG}_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, separate passes; FETCH_SIZE x2 per MI355X_MICROARCH.md)" > "$O/pmc_summary.txt" 2>&1
      find "$O/pmc" -name "*.csv" -size +2M -delete
      cat "$O/pmc_summary.txt" ;;
    py)
      s=$1; shift
      python "$R/$s" "$@" > "$O/py_$(slug "$(basename "$s")$*").txt" 2>&1
      tail -40 "$O/py_$(slug "$(basename "$s")$*").txt" ;;
    *) echo "unknown recipe: $cmd" ;;
  esac
done
