# config 4 (IPPO 2048 x 10 x 256, 125-column observations, 17 actions): parity tests + bench legs with the hand-ordered / compiler-scheduled actor forms
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04cfg4
mkdir -p $O
cd $R
python -m pytest tests/test_hip_parity.py -q -m gpu -k "ippo or full_size or two_chunk or forms" 2>&1 | tail -3
for f in auto loop; do
  CM_MLP_FORMS=$f python bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d.get('phase_ms'), {k:v for k,v in d.get('kernel_ms',{}).items()})"
done
