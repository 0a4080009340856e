#!/bin/bash
# round 3: soak runs with fresh device-side resets every episode -- nothing non-finite may reach the outputs, the non-finite guard's count is reported
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r03q}
rm -rf $O && mkdir -p $O
cd $R
timeout 400 python tools/gpu_soak.py 2000 device FeedingJacoVecEnv 4096 > $O/soak_feeding_device.log 2>&1; tail -1 $O/soak_feeding_device.log
timeout 400 python tools/gpu_soak.py 2000 device FeedingPandaVecEnv 4096 > $O/soak_feeding_panda_device.log 2>&1; tail -1 $O/soak_feeding_panda_device.log
timeout 400 python tools/gpu_soak.py 2000 device ScratchItchPR2HumanVecEnv 4096 > $O/soak_scratchitch_device.log 2>&1; tail -1 $O/soak_scratchitch_device.log
timeout 400 python tools/gpu_soak.py 2000 pool BedBathingSawyerVecEnv 4096 > $O/soak_bedbathing_pool.log 2>&1; tail -1 $O/soak_bedbathing_pool.log
timeout 500 python tools/gpu_soak.py 620 device DressingBaxterVecEnv 1024 > $O/soak_dressing_device.log 2>&1; tail -1 $O/soak_dressing_device.log
