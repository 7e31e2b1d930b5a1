R=$GRAFT_REPO_ROOT
cd /tmp
for m in beside defer1 defer2 defer3 beside defer1 defer2; do
  CM_GRU_CRITIC=$m python $R/bench.py --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; b=json.loads(sys.stdin.read()); print('cfg5 critic=$m', round(b['ms_per_step'],4), {k:round(v,3) for k,v in b['phase_ms'].items()}, b['kernel_ms'])"
done
cd $R
for m in defer1 defer2; do
  CM_GRU_CRITIC=$m timeout 600 python -m pytest tests -m gpu -x -q -k "gru or lstm" 2>&1 | tail -2
done
