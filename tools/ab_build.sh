#!/bin/bash
# build variants of libagx.so for same-box A/B runs:  tools/ab_build.sh name "-DFLAG ..." [name2 "-D..."]
cd $(dirname $0)/..
mkdir -p assistive_gym_amd/lib/variants
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value $2 -o assistive_gym_amd/lib/variants/$1.so assistive_gym_amd/csrc/agx_api.hip &
  shift 2
done
wait
ls -la assistive_gym_amd/lib/variants
