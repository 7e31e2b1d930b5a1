set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01l
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
mkdir -p $O/pmc
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc/pmc_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc/pmc_write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc/pmc_mfma -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1
find $O/pmc -name "*.csv" -size +2M -delete
python $R/tools/phase_prof.py actor > $O/phase_actor.txt 2>&1
python $R/tools/phase_prof.py critic > $O/phase_critic.txt 2>&1
python $R/tools/phase_prof.py rollout > $O/phase_rollout.txt 2>&1
python $R/tools/bench_configs.py > $O/configs_learner.txt 2>&1
for w in cfg2 cfg4 cfg5; do python $R/bench.py --workload $w --no-cpu-baseline >> $O/bench_other_workloads.txt 2>/dev/null; done
python $R/tools/bench_coma.py > $O/coma_bench.json 2> $O/coma.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kc -- python $R/tools/bench_coma.py --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/kc -name "*kernel_stats.csv" | head -1) $O/coma_kernel_stats.csv
ls -la $O
# opt-in compensated-bf16 arithmetic (NOT the default): same bench / kernel stats / phases under the flag
CM_MFMA=bf16x3 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_bf16x3.json 2>/dev/null
CM_MFMA=bf16x3 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kb -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/kb -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bf16x3.csv
CM_MFMA=bf16x3 python $R/tools/phase_prof.py actor > $O/phase_actor_bf16x3.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k5 -- python $R/bench.py --workload cfg5 --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/k5 -name "*kernel_stats.csv" | head -1) $O/cfg5_kernel_stats.csv
ls -la $O
# layered schedule (hidden 65..256 / deeper than 2 hidden layers)
python $R/tools/bench_wide.py 2>&1 | grep -v amdgpu.ids > $O/wide_schedule.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kw -- python $R/tools/probes/wide_prof.py > /dev/null 2>&1
cp $(find /tmp/kw -name "*kernel_stats.csv" | head -1) $O/wide_kernel_stats.csv
