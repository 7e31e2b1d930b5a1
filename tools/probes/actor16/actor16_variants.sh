# Builds probe variants of the library that differ only in cm_mlp_actor16.hip's -D flags (tools/probes/_variants/<name>.so, git-ignored),
# for tools/probes/actor16_ab.sh.  usage: actor16_variants.sh <name> [-DA16_BPERMUTE | -DA16_ABL=1 | -DA16_ABL=4 ...]
# Needs the objects of a normal build (python -m cleanmarl_amd.build) under cleanmarl_amd/build/.
set -e
name=$1; shift
cd "$(dirname "$0")/../../cleanmarl_amd"
O=build/libcleanmarl_hip.so.obj
V=../tools/probes/_variants
mkdir -p $V build/a16_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -mllvm -amdgpu-use-amdgpu-trackers=1 \
  -Rpass-analysis=kernel-resource-usage "$@" -c csrc/cm_mlp_actor16.hip -o build/a16_$name/a16.o 2>&1 | grep -A12 "k_actor16" | grep -E "VGPRs Spill|error" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/$name.so $(ls $O/*.o | grep -v cm_mlp_actor16) build/a16_$name/a16.o
