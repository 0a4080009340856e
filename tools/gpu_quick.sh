cd ${GRAFT_REPO_ROOT:-/root/repo}
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null
python -c "
import sys; sys.path.insert(0,'.')
import bench; print('usable cores', bench._usable_cores())"
timeout 600 python bench.py --steps 100 --warmup 20 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(round(j['value']), j['cpu_baseline'])"
