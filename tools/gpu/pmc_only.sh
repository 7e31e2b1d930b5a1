# the three PMC passes + summary + stamped record only (after a late kernel change: profiles/pmc_dominant_kernel.json must carry the hash of the sources)
set -x
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}_pmc
mkdir -p $O/pmc
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_noextras.json 2>/dev/null; cat $O/bench_noextras.json | head -c 1500
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc/pmc_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc/pmc_write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc/pmc_mfma -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc --emit $O/pmc_dominant_kernel.json "profiles/${TAG}_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES, separate passes; FETCH_SIZE x2 per MI355X_MICROARCH.md)" > $O/pmc_summary.txt 2>&1
find $O/pmc -name "*.csv" -size +2M -delete
cat $O/pmc_summary.txt | head -8
