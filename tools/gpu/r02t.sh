# round 2, step t: critic overlap schedule (0 = one stream, 2 = critic epochs on the low-priority stream) per share size
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02t
mkdir -p $O
cd $R
for e in 2048 1024 512 256; do for o in 0 1 2; do
  CM_CRITIC_OVERLAP=$o python bench.py --envs $e --steps 30 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e overlap=$o', round(d['ms_per_step'],3))" | tee -a $O/ab.txt
done; done
