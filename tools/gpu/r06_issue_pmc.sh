# round 6: instruction-issue counters of the kernels of one iteration (is the fp32 actor pass bound by MFMA + VALU issue?): two --pmc passes
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O/pmc_issue; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_issue/a -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --solo-launches 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_issue/b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --solo-launches 0 > /dev/null 2>&1
python - $O/pmc_issue > $O/issue_counters.txt <<'PYEOF'
import collections, csv, glob, os, sys
root = sys.argv[1]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_COEXEC_CYCLES",
         "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"]
print("per launch (mean over launches), raw counter values summed over the device; rocprofv3 --pmc, two passes")
import json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from cleanmarl_amd.build import source_hash
rec = {"source_hash": source_hash(), "workload": "cfg3", "kernels": {}}
for k in d:
    if "at::native" in k or "rocclr" in k:
        continue
    rec["kernels"][k] = {n: sum(v) / len(v) for n, v in d[k].items()}
json.dump(rec, open(os.path.join(os.path.dirname(root), "issue_counters.json"), "w"), indent=1)
for k in sorted(d, key=lambda k: -sum(d[k].get("GRBM_GUI_ACTIVE", [0]))):
    if "at::native" in k or "rocclr" in k:
        continue
    print(k[:100])
    for n in names:
        if n in d[k]:
            v = d[k][n]
            print(f"    {n:30s} {sum(v) / len(v):16.0f}   ({len(v)} launches)")
PYEOF
cat $O/issue_counters.txt | head -60
find $O/pmc_issue -name "*.csv" -size +1M -delete
