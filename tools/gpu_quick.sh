cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests -m gpu -x -q -s 2>&1 | grep "worst\|passed\|failed"
STEPS=200 bash tools/ab_run.sh
