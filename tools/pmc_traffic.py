"""Reduce the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of
tools/pmc_workload.py, as /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes) to per-launch
HBM bytes per kernel -> profiles/r02_traffic_<task>.json (read by bench.py for roofline.traffic and
roofline.valu_issue_frac).

  python tools/pmc_traffic.py <task> <fetch_counter_collection.csv> <write_counter_collection.csv> [<valu_counter_collection.csv>] [envs]
      [--more <counter_collection.csv> ...] [--stats <kernel_stats.csv>] [--out profiles/r03_traffic_<task>.json]
  --more: further PMC passes; every counter found is reported per environment and launch (`counters_per_env_launch`)
  --stats: a rocprofv3 --kernel-trace --stats summary of the same workload -> `ms_per_launch`

Units / corrections: both counters are in KiB.  On gfx950 FETCH_SIZE tallies 128-B requests at
64 B, so it is doubled (guide); WRITE_SIZE is used as reported.  The observation-only launches of
the workload have a known byte count (one state record read, one observation written per env)
and are reported beside the corrected counters as a calibration check."""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, counter):
    """kernel -> [sum of bytes, sum of environments (= workgroups), launches]; launches of a step come in
    chunks of different sizes, so traffic is normalised per environment"""
    acc = defaultdict(lambda: [0.0, 0, 0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter and r['Kernel_Name'].startswith('agx_'):
            a = acc[r['Kernel_Name'].split('(')[0]]
            a[0] += float(r['Counter_Value']) * 1024.0; a[1] += int(r['Grid_Size']) // int(r['Workgroup_Size']); a[2] += 1
    return acc


def per_kernel_count(path, counter):
    """kernel -> [sum of counter values, sum of environments, launches]"""
    acc = defaultdict(lambda: [0.0, 0, 0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter and r['Kernel_Name'].startswith('agx_'):
            a = acc[r['Kernel_Name'].split('(')[0]]
            a[0] += float(r['Counter_Value']); a[1] += int(r['Grid_Size']) // int(r['Workgroup_Size']); a[2] += 1
    return acc


def main():
    argv = sys.argv
    more, stats, outp = [], None, None
    if '--out' in argv:
        k = argv.index('--out'); outp = argv[k + 1]; del argv[k:k + 2]
    if '--stats' in argv:
        k = argv.index('--stats'); stats = argv[k + 1]; del argv[k:k + 2]
    if '--more' in argv:
        k = argv.index('--more'); more = argv[k + 1:]; del argv[k:]
    task = sys.argv[1]
    fetch, write = per_kernel(sys.argv[2], 'FETCH_SIZE'), per_kernel(sys.argv[3], 'WRITE_SIZE')
    valu = per_kernel_count(sys.argv[4], 'SQ_INSTS_VALU') if len(sys.argv) > 4 and sys.argv[4].endswith('.csv') else {}
    envs = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 4096
    sys.path.insert(0, ROOT)
    from assistive_gym_amd.blob import ModelBlob
    blob = ModelBlob.load({'feeding': 'feeding_jaco', 'bedbathing': 'bed_bathing_sawyer', 'scratchitch': 'scratch_itch_pr2', 'armmanipulation': 'arm_manipulation_sawyer', 'dressing': 'dressing_baxter', 'drinking': 'drinking_jaco'}[task])
    if task == 'scratchitch':
        blob = blob.coop()
    out = {'envs': envs, 'note': 'hbm_bytes_per_launch is for a launch over all `envs` environments; a step issues chunks of them', 'correction': 'FETCH_SIZE x2 (gfx950, guide), WRITE_SIZE as reported', 'kernels': {}}
    for k in sorted(set(fetch) | set(write)):
        f = fetch[k][0] / max(1, fetch[k][1])           # raw bytes per environment of a launch
        w = write[k][0] / max(1, write[k][1])
        out['kernels'][k] = {'launches_sampled': fetch[k][2], 'fetch_raw_bytes_per_env': f, 'write_raw_bytes_per_env': w,
                             'hbm_bytes_per_env_launch': 2.0 * f + w, 'hbm_bytes_per_launch': (2.0 * f + w) * envs}
        if k in valu and valu[k][1]:
            out['kernels'][k]['valu_insts_per_env_launch'] = valu[k][0] / valu[k][1]     # wave-level VALU instructions per environment (= per wave) and launch
    for path in more:                        # every counter of the extra passes, per environment (= workgroup) and launch
        names = {r['Counter_Name'] for r in csv.DictReader(open(path))}
        for cname in sorted(names):
            for k, a in per_kernel_count(path, cname).items():
                if k in out['kernels'] and a[1]:
                    out['kernels'][k].setdefault('counters_per_env_launch', {})[cname] = a[0] / a[1]
    if stats:
        for r in csv.DictReader(open(stats)):
            k = r['Name'].split('(')[0]
            if k in out['kernels']:
                out['kernels'][k]['ms_per_launch'] = float(r['AverageNs']) * 1e-6; out['kernels'][k]['launches_in_trace'] = int(r['Calls'])
    obs_k = [k for k in out['kernels'] if k.startswith('agx_observe_kernel')]
    if obs_k:
        o = out['kernels'][obs_k[0]]
        known_r, known_w = blob.state_words * 4, blob.obs_dim * 4
        out['calibration'] = {'kernel': obs_k[0], 'known_read_bytes_per_env': known_r, 'known_write_bytes_per_env': known_w,
                              'fetch_x2_over_known': 2.0 * o['fetch_raw_bytes_per_env'] / known_r, 'write_over_known': o['write_raw_bytes_per_env'] / known_w}
    json.dump(out, open(outp or os.path.join(ROOT, 'profiles', 'r03_traffic_%s.json' % task), 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
