# kernel timeline of one steady-state iteration of the PRODUCT surface (driver.run through the CLI, no bench instrumentation) at the 512-env share
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r03cli
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_cli -- python -m cleanmarl_amd.mappo_multienvs --env_type=synthetic --batch_size=512 --synthetic_agents=8 --synthetic_steps=128 --total_timesteps=$((512*128*300)) --eval_steps=1000000000 --log_every=1000000000 --actor_hidden_dim=64 --critic_hidden_dim=64 > /dev/null 2>$R/gpurun_out/r03cli/err.txt
python $R/tools/trace_timeline.py $(find /tmp/kt_cli -name "*kernel_trace.csv" | head -1) k_ro 20 > $R/gpurun_out/r03cli/timeline_cli_envs512.txt 2>&1
cat $R/gpurun_out/r03cli/timeline_cli_envs512.txt
