# COMA at the default 128-wide critic: MFMA-busy and LDS bank-conflict counters per kernel (separate passes)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04comapmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p1 -- python $R/tools/bench_coma.py --critic-hidden 128 --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/p2 -- python $R/tools/bench_coma.py --critic-hidden 128 --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
python - <<'PY' | tee $O/coma128_pmc.txt
import csv, glob, collections
def agg(d):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    out = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for fn in f:
        for r in csv.DictReader(open(fn)):
            k = r['Kernel_Name'][:60]
            out[k][r['Counter_Name']] += float(r['Counter_Value'])
            cnt[(k, r['Counter_Name'])] += 1
    return out, cnt
o1, c1 = agg('/tmp/p1'); o2, c2 = agg('/tmp/p2')
for k in sorted(o1, key=lambda k: -o1[k].get('GRBM_GUI_ACTIVE', 0))[:10]:
    g = o1[k].get('GRBM_GUI_ACTIVE', 0); m = o1[k].get('SQ_VALU_MFMA_BUSY_CYCLES', 0)
    n = c1[(k, 'GRBM_GUI_ACTIVE')]
    bc = o2.get(k, {}).get('SQ_LDS_BANK_CONFLICT', 0); la = o2.get(k, {}).get('SQ_LDS_IDX_ACTIVE', 0)
    print("%-62s launches %3d  gui_active/launch %10.0f  mfma_busy %% (of 4 SIMD x 256 CU x active... raw ratio) %6.3f  lds conflict/active %6.3f" % (k, n, g / max(n, 1), m / max(g, 1), bc / max(la, 1)))
PY
