cd ${GRAFT_REPO_ROOT:-/root/repo}
for c in 3 4 5 6; do
  GPU_MAX_HW_QUEUES=8 AGX_CHUNKS=$c timeout 200 python bench.py --task feeding --steps 300 --warmup 20 --no-cpu-baseline --no-configs > /tmp/c$c.json 2> /tmp/c$c.err
  python - <<PY
import json
try:
    j = json.load(open('/tmp/c$c.json')); print('hwq8 chunks $c', round(j['value']))
except Exception as e: print('chunks $c failed', e)
PY
done
