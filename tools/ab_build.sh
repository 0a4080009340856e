#!/bin/bash
# build variants of libagx.so for same-box A/B runs:  tools/ab_build.sh name "-DFLAG ..." [name2 "-D..."]
cd $(dirname $0)/..
mkdir -p assistive_gym_amd/lib/variants
while [ $# -ge 2 ]; do
  python -m assistive_gym_amd.build --out assistive_gym_amd/lib/variants/$1.so --extra "$2" &
  shift 2
done
wait
ls -la assistive_gym_amd/lib/variants
