"""How much the wheel model of the Stretch depends on the [BULLET-UNVERIFIED] friction conventions, measured in the CPU oracle: the device
solves ONE friction row per contact along the slip direction; Bullet may use two fixed directions, warm starting and a residual early-out
(the oracle's switches AGX_P_ORACLE_*, which the device does not have).  Three drives of two seconds each -- straight, turning on the spot,
an arc -- from one settled FeedingStretch state per ground friction, with the device's conventions and with each switch on: distance driven,
yaw turned, and both relative to pure rolling (wheel radius x wheel angle; radius x angle difference / track).

    python tests/diag/stretch_friction_model.py -> profiles/r03/stretch_friction_model.json (+ a markdown table on stdout)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from assistive_gym_amd.blob import ModelBlob       # noqa: E402
from assistive_gym_amd.host.reset import make_states   # noqa: E402
from oracle_lib import Oracle                       # noqa: E402

SWITCHES = {'device conventions': {}, 'two friction directions': dict(FRICTION_DIRS=2), 'warm start 0.85': dict(WARMSTART=0.85),
            'residual early-out 1e-7': dict(ORACLE_RESIDUAL_EPS=1e-7), 'all three': dict(ORACLE_RESIDUAL_EPS=1e-7, FRICTION_DIRS=2, WARMSTART=0.85)}
DRIVES = {'straight': (1.0, 1.0), 'spin': (1.0, -1.0), 'arc': (1.0, 0.5)}
R_WHEEL, TRACK = 0.0508, 2 * 0.15765


def main():
    blob = ModelBlob.load('feeding_stretch')
    st, _ = make_states(blob, 1, seed=4005, impairment='none')
    out = {'wheel_radius': R_WHEEL, 'track': TRACK, 'steps': 20, 'rows': []}
    for fr in (0.5, 0.1, 0.025):                    # the range build_assistive_env draws the ground's friction from (env.py:120); x 0.5 for the wheel's own
        for sw, kw in SWITCHES.items():
            b = blob
            for k, v in kw.items():
                b = b.set_param(k, v)
            o = Oracle(b)
            for name, aw in DRIVES.items():
                s = st[0].copy()
                b.view(s[None])['plane_friction'][0] = fr
                if hasattr(o.L, 'agxo_warm_clear'):
                    o.L.agxo_warm_clear()
                o.settle(s, 25)
                q0 = b.view(s[None])['q'][0].copy()
                a = np.zeros(5, np.float32); a[0], a[1] = aw
                for _ in range(20):
                    o.step(s, a)
                q1 = b.view(s[None])['q'][0].copy()
                dth = q1[6:8] - q0[6:8]
                dist, yaw = float(np.linalg.norm(q1[:2] - q0[:2])), float(q1[3] - q0[3])
                roll_dist, roll_yaw = float(R_WHEEL * dth.mean()), float(R_WHEEL * (dth[0] - dth[1]) / TRACK)
                out['rows'].append(dict(ground_friction=fr, conventions=sw, drive=name, distance=dist, yaw=yaw, rolling_distance=roll_dist, rolling_yaw=roll_yaw))
    dst = os.path.join(ROOT, 'profiles', 'r03', 'stretch_friction_model.json')
    json.dump(out, open(dst, 'w'), indent=1)
    print('| ground friction | conventions | straight: distance / rolling | spin: yaw / rolling | arc: distance, yaw / rolling |')
    print('|---|---|---|---|---|')
    for fr in (0.5, 0.1, 0.025):
        for sw in SWITCHES:
            r = {x['drive']: x for x in out['rows'] if x['ground_friction'] == fr and x['conventions'] == sw}
            f = lambda a, b: '%.3f (%.0f %%)' % (a, 100 * a / b) if abs(b) > 1e-9 else '%.3f' % a
            print('| %.3f | %s | %s | %s | %s, %s |' % (fr, sw, f(r['straight']['distance'], r['straight']['rolling_distance']), f(r['spin']['yaw'], r['spin']['rolling_yaw']),
                                                 f(r['arc']['distance'], r['arc']['rolling_distance']), f(r['arc']['yaw'], r['arc']['rolling_yaw'])))
    print('->', dst)


if __name__ == '__main__':
    main()
