# round 2, step q: fused wide-input critic kernel -- parity then A/B against the split schedule
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02q
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "critic or golden or learner or update or additiv or gru" 2>&1 | tail -15 > $O/parity.txt
cat $O/parity.txt
CM_PROF_WARMUP=100 python tools/phase_prof.py critic > $O/phase_critic_fused.txt 2>&1
cat $O/phase_critic_fused.txt
for s in fused split; do
  CM_CRITIC_SCHEDULE=$s python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_$s.json 2> $O/bench_$s.err
  python - <<P
import json
d = json.loads(open("$O/bench_$s.json").read().strip().splitlines()[-1])
print("$s", d["ms_per_step"], {k: v for k, v in d.get("phase_roofline", {}).items() if "critic" in k})
P
done
