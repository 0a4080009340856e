cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in assistive_gym_amd/lib/variants/*.so; do AGX_LIB=$PWD/$v timeout 300 python tools/micro_solve.py 2>&1 | tail -1; done
