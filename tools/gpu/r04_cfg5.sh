R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -q -m gpu -k "gru or lstm or eval or resume or checkpoint" 2>&1 | tail -3
for k in 1 2 3; do python bench.py --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5', d['ms_per_step'], d.get('phase_ms'))"; done
