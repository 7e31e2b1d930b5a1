#!/bin/bash
# tools/debug/solo_ab.sh VAR v1 v2 .. -- [bench.py args]: bench.py once per value of VAR, printing the iteration time and the SOLO phase times
# (phase_solo_ms: each phase alone on the device) -- kernel A/Bs whose in-iteration numbers are co-residency figures under the two-stream schedule
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
var=$1; shift
vals=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ $# -gt 0 ] && shift
for v in "${vals[@]}"; do
  for rep in $(seq 1 "${REPS:-2}"); do
    env "$var=$v" python "$R/bench.py" --steps "${STEPS:-20}" --warmup "${WARMUP:-5}" --no-cpu-baseline --no-extras "$@" 2>/dev/null | grep '^{' | python -c "
import json, sys
o = json.loads(sys.stdin.readline())
print('$var=$v [$*] ms_per_step %.4f  solo %s' % (o['ms_per_step'], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in o['phase_solo_ms'].items() if k != 'note'}))"
  done
done
