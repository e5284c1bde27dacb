# per-kernel durations of the captured latent-UNet forward (graph replay) under rocprofv3; prints the table via prof_ldm_table.py
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_fwd
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_fwd -o p -- python3 $GRAFT_REPO_ROOT/tools/experiments/ab_ldm_forward.py $GRAFT_REPO_ROOT/jointimagegeneration_amd/csrc/libguidegen_hip.so > $GRAFT_REPO_ROOT/gpurun_out/prof_fwd.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/experiments/prof_ldm_table.py $GRAFT_REPO_ROOT/gpurun_out/prof_fwd/p_results.db
