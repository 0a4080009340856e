cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests -m gpu -x -q -s 2>&1 | grep "worst\|passed\|failed\|Error\|error" | head
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(round(j['value']), j['roofline']['kernels_ms_per_step'])"
