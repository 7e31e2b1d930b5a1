cd $GRAFT_REPO_ROOT
for s in 52 54 55 56 57 58 60; do
  CM_TILE_SPLIT=$s python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "
import json,sys
o=json.loads(sys.stdin.readline()); print('split $s', round(o['ms_per_step'],4), round(o['phase_solo_ms']['actor_fwd_bwd'],4))"
done
