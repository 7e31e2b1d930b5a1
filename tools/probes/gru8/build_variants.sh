# experiment builds of the eight-wave GRU forward sweep: only cm_gru.hip is recompiled (-DCM_PHASE_PROF + one switch), the rest of the
# objects come from the last `python cleanmarl_amd/build.py --prof`.  Usage: bash tools/probes/gru8/build_variants.sh NOST4 NOST2 NODH ...
cd "$(dirname "$0")/../../.."
OBJ=cleanmarl_amd/build/libcleanmarl_hip_prof.so.obj
for v in "$@"; do
  flags=""
  for f in $(echo $v | tr '+' ' '); do flags="$flags -DCM_X_$f"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -DCM_PHASE_PROF $flags -c cleanmarl_amd/csrc/cm_gru.hip -o /tmp/cm_gru_$v.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/gru8/lib_$v.so /tmp/cm_gru_$v.o $(ls $OBJ/*.hip.o | grep -v cm_gru.hip.o) && echo built $v
done
