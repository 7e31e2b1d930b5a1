R=$GRAFT_REPO_ROOT
cd $R
for v in NOST4 NOST2 NODH NOST4+NOST2+NODH; do echo "== $v"; CM_PROF_LIB=$R/tools/probes/gru8/lib_$v.so timeout 300 python tools/phase_prof.py gru 2>&1 | grep -v amdgpu.ids | head -18 | grep -v " 0.0  "; done
