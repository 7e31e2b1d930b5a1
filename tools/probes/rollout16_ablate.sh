# Builds probe variants of the library that differ only in cm_rollout.hip's -DRO16_ABL=<mask> (phases of the 16-row rollout step
# compiled out; results are then wrong on purpose) into tools/probes/_variants/ro16_<mask>.so (git-ignored).
# usage (build box):  bash tools/probes/rollout16_ablate.sh 1 2 4 8 16 32 64    then on the GPU:  bash tools/probes/rollout16_ablate.sh run 1 2 ...
cd "$(dirname "$0")/../.."
V=tools/probes/_variants; O=cleanmarl_amd/build/libcleanmarl_hip.so.obj
if [ "$1" = run ]; then
  shift
  for m in 0 "$@"; do
    lib=$PWD/cleanmarl_amd/libcleanmarl_hip.so; [ $m != 0 ] && lib=$PWD/$V/ro16_$m.so
    echo "RO16_ABL=$m: $(CM_LIB_PATH=$lib RO16_ONLY_TIME=1 python tools/probes/rollout_tile_ab.py 2>/dev/null | grep 'E=  512 A=8')"
  done
  exit 0
fi
mkdir -p $V
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -DRO16_ABL=$m -c cleanmarl_amd/csrc/cm_rollout.hip -o /tmp/ro16_$m.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/ro16_$m.so $(ls $O/*.o | grep -v cm_rollout) /tmp/ro16_$m.o && echo built $V/ro16_$m.so
done
