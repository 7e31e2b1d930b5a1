# two ranks on ONE GPU (gloo control plane, test hook): kernel timeline of one steady-state iteration per rank with the RCCL-stand-in (gloo)
# exchange and with the one-shot peer exchange (hipIpc mailboxes) -- where each message / step sits relative to the next actor pass
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04peer
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in rccl peer; do
  rm -rf /tmp/kt_$mode
  CM_BENCH_BACKEND=gloo rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$mode -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 $R/bench.py --gpus 2 --steps 10 --warmup 3 --envs 1024 --allreduce $mode --no-extras > $O/bench_$mode.json 2> $O/bench_$mode.err
  cat $O/bench_$mode.json | head -c 600; echo
  n=0
  for f in $(find /tmp/kt_$mode -name "*kernel_trace.csv"); do
    if grep -q k_rollout $f; then n=$((n+1)); python $R/tools/trace_timeline.py $f k_ro 3 > $O/timeline_${mode}_proc$n.txt 2>&1; fi
  done
  head -40 $O/timeline_${mode}_proc1.txt
done
