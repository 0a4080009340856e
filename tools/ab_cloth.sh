#!/bin/bash
# on the GPU box: tools/gpu_cloth_bench.py with every variant library
cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/gpu_cloth_bench.py ${1:-256} ${2:-5}
for v in assistive_gym_amd/lib/variants/*.so; do AGX_LIB=$PWD/$v python tools/gpu_cloth_bench.py ${1:-256} ${2:-5}; done
