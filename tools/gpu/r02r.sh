# round 2, step r: per-GPU shares with the fused critic + kernel breakdown of the 512-env share; PMC traffic of the fused critic
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02r
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for e in 4096 2048 1024 512; do python $R/bench.py --workload cfg3 --envs $e --steps 30 --warmup 8 --no-cpu-baseline --no-extras >> $O/cfg3_shares.txt 2>/dev/null; done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k512 -- python $R/bench.py --envs 512 --steps 30 --warmup 8 --no-cpu-baseline --no-extras > /dev/null 2>&1
cp $(find /tmp/k512 -name "*kernel_stats.csv" | head -1) $O/cfg3_512_kernel_stats.csv
cp $(find /tmp/k512 -name "*kernel_trace.csv" | head -1) $O/cfg3_512_kernel_trace.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  python - "$f" $c >> $O/pmc_critic.txt <<'P'
import csv, sys, collections
f, c = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == c:
        agg[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:6]:
    print(c, k, "calls", len(v), "mean", sum(v) / len(v))
P
done
cat $O/pmc_critic.txt
python - <<P
import json
for l in open("$O/cfg3_shares.txt"):
    d = json.loads(l); print(d["config"].get("envs_per_gpu", d["config"]), d["ms_per_step"], {k: round(v["ms"], 3) for k, v in d.get("phase_roofline", {}).items()})
P
head -25 $O/cfg3_512_kernel_stats.csv | cut -c1-200
