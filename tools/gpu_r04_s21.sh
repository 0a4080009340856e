#!/bin/bash
# round 4, session 21: TIMING experiment -- the narrowphase of every pass executed twice against the default: what one narrowphase costs the STEP
# under the chunk overlap (the ceiling of any further gain in it)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04u; mkdir -p $O; cd $R
STEPS=200 bash tools/ab_run.sh > $O/ab_narrowphase_twice.txt 2>&1; grep -v amdgpu $O/ab_narrowphase_twice.txt
