#!/bin/bash
# round 4, session 20: TIMING experiment -- every wave_sync() as an LDS-only wait (unsafe build, results not checked) against the default: an upper
# bound on what the workgroup fences of __syncthreads() cost the wave-per-environment kernels
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04t; mkdir -p $O; cd $R
STEPS=200 bash tools/ab_run.sh > $O/ab_sync.txt 2>&1; grep -v amdgpu $O/ab_sync.txt
