cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/gpu_diag.py 2>&1 | tail -22
timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(round(j['value']), j['roofline']['kernels_ms_per_step'])"
