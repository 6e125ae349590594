export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/gpu_e2e_ab.sh r06e2q "q0p2:DIFACTO_PREP_PRIORITY=0,q0p3:DIFACTO_PREP_PRIORITY=0+DIFACTO_PREP_STREAMS=3,q0p2b:DIFACTO_PREP_PRIORITY=0" rec,criteo
bash tools/gpu_modes.sh r06e2q_bench "|--ctx-option prep_priority=0" 2>&1 | grep -v "^W2026"
