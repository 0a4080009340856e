cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests -m gpu -x -q -s 2>&1 | grep "worst\|passed\|failed"
STEPS=200 bash tools/ab_run.sh
AGX_LIB=$PWD/assistive_gym_amd/lib/variants/b_greg.so timeout 300 python tools/gpu_diag.py 2>&1 | grep "narrowphase\|selection\|sweep\|collide  \|cull\|aabbs"
