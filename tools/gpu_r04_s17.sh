#!/bin/bash
# round 4, session 17: same-box A/B of the one-sided sphere cull in the broadphase sweep (exact: the second bounding test cannot reject what the first passed when A is a sphere)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04q; mkdir -p $O; cd $R
STEPS=300 bash tools/ab_run.sh > $O/ab_one_sided.txt 2>&1; grep -v amdgpu $O/ab_one_sided.txt
