#!/bin/bash
# round 4, session 16: build-kernel phase cycles of the final kernels (FeedingJaco), same-box A/B of the solve kernel's LDS row window
# (640 / 960 = default / 1280 pairs; parity-neutral: which rows come from LDS and which from L2)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04p; mkdir -p $O; cd $R
timeout 200 python3 tools/gpu_build_phases.py FeedingJacoVecEnv > $O/build_phases_feeding.txt 2>&1; grep -v amdgpu $O/build_phases_feeding.txt
STEPS=300 bash tools/ab_run.sh > $O/ab_lds_window.txt 2>&1; grep -v amdgpu $O/ab_lds_window.txt
