#!/bin/bash
# on the GPU box: per-kernel durations (rocprofv3 kernel trace) of tools/gpu_cloth_bench.py for every variant library
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in default $R/assistive_gym_amd/lib/variants/*.so; do
  rm -rf /tmp/cp && mkdir -p /tmp/cp
  if [ $v = default ]; then unset AGX_LIB; else export AGX_LIB=$v; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp -- python $R/tools/gpu_cloth_bench.py ${1:-256} ${2:-5} > /dev/null 2>&1
  f=$(find /tmp/cp -name "*kernel_stats.csv" | head -1)
  echo "== $(basename $v)"; grep -E "cloth|build|solve" $f | cut -d, -f1-4
done
