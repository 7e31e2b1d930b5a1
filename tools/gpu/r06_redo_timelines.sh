set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for w in "cfg3" "cfg3 --envs 512" "cfg2"; do
  n=$(echo $w | tr -d ' -')
  rm -rf /tmp/kt_$n
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$n -- python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extras --solo-launches 0 > /dev/null 2>&1
  python $R/tools/trace_timeline.py $(find /tmp/kt_$n -name "*kernel_trace.csv" | head -1) k_ro 3 > $O/timeline_$n.txt 2>&1
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
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k5 -- python $R/bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_cfg5.json 2>/dev/null
cp $(find /tmp/k5 -name "*kernel_stats.csv" | head -1) $O/cfg5_kernel_stats.csv
cd $R
$R/tools/probes/_bin/cosimd_overlap > $R/gpurun_out/r06/cosimd_overlap.txt 2>&1; cat $R/gpurun_out/r06/cosimd_overlap.txt
for f in cfg3 cfg5; do cat $O/foreign_launches_$f.txt; head -30 $O/timeline_$f.txt; done
