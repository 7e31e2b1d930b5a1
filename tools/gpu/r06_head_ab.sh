#!/bin/bash
# round 6: the actor pass's head -- (a) layout / issue cost of v_mfma_f32_4x4x1_16b_f32, (b) CM_HEAD_FAST A/B (hardware exp / log / rcp in the PPO head)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
O=gpurun_out/r06_head_ab.txt
{
  echo "# tools/probes/mfma4x4_layout.hip"
  tools/probes/_bin/mfma4x4_layout
  echo "# tools/debug/solo_ab.sh CM_LIB_PATH ..."
  LIBS=""
  for s in "" $AB_SUFFIXES; do LIBS="$LIBS $R/cleanmarl_amd/libcleanmarl_hip${s:+_$s}.so"; done
  REPS=2 tools/debug/solo_ab.sh CM_LIB_PATH $LIBS $R/cleanmarl_amd/libcleanmarl_hip.so --
} > $O 2>&1
cat $O
