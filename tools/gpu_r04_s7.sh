#!/bin/bash
# session 7: the three tests that were red in session 6, with the failing cases written to gpurun_out/ for an emulator replay
O=gpurun_out/r04g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scratch_itch_robots.py -m gpu -q -k "warm_start or sawyer" -s 2>&1 | tail -40 > $O/pytest_three.log
echo "pytest rc=$?"; cat $O/pytest_three.log | cut -c1-300
mv gpurun_out/*.npz $O/ 2>/dev/null
