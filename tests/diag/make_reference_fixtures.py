"""Generates tests/golden/ref_steps.npz and tests/golden/ref_units.npz by EXECUTING THE REFERENCE'S OWN PYTHON
(/root/reference/assistive_gym, imported unmodified through tests/refbridge: a `pybullet` facade over the CPU oracle's physics
plus inert gym / ray / keras stubs).  Run here (the reference does not exist on the GPU box); the fixtures are committed.

    python tests/diag/make_reference_fixtures.py

ref_steps.npz -- per case of tests/refcases.py: the inputs (model, co-op flag, state record, garment, action) and what the
  reference's <Task><Robot>Env.step() returned on them: observation, reward, done, info['total_force_on_human'],
  info['task_success'], task-specific attributes (forces, new contact points, sleeve verdicts ...) and the state record
  after the step (the bookkeeping the reference keeps in Python attributes written back in the record's layout).
ref_units.npz -- direct calls: Util.sleeve_on_arm_reward / line_intersects_triangle on random sleeve / arm configurations
  (util.py:125-202), Util.capsule_points and point_on_capsule (util.py:58-113), AssistiveEnv.human_preferences for the six
  task strings (env.py:237-274), HumanCreation.create_human under a recording createMultiBody (human_creation.py:58-316,
  both genders, with and without the cloth spheres, two limit scales), config.ini through the reference's configparser use.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import refbridge as rb      # noqa: E402
import refcases             # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def extras_of(env, task):
    g = lambda *names: np.array([float(getattr(env, n)) for n in names])
    if task == 'feeding': return g('robot_force_on_human', 'spoon_force_on_human')
    if task == 'bed_bathing': return g('tool_force', 'tool_force_on_human', 'total_force_on_human', 'new_contact_points')
    if task == 'scratch_itch': return g('total_force_on_human', 'tool_force', 'tool_force_at_target')
    if task == 'drinking': return g('robot_force_on_human', 'cup_force_on_human')
    if task == 'arm_manipulation': return g('tool_right_force', 'tool_left_force', 'tool_right_force_on_human', 'tool_left_force_on_human', 'total_force_on_human')
    return g('cloth_force_sum', 'robot_force_on_human', 'forearm_in_sleeve', 'upperarm_in_sleeve')


def make_steps():
    out = {}
    cases = refcases.build_cases()
    names = []
    for c in cases:
        b = refcases.variant_blob(c['model'], c['coop'], c['variant'])
        r = rb.ref_step(b, c['state'], c['action'], c['cloth'])
        w = r['world']
        assert not w.gain_mismatches, (c['name'], w.gain_mismatches[:3])      # Agent.control's gains / forces == the blob's
        assert not w.ignored, (c['name'], w.ignored[:3])
        n = c['name']; names.append(n)
        out[n + '/state'] = c['state']; out[n + '/action'] = c['action']
        if c['cloth'] is not None: out[n + '/cloth'] = c['cloth']
        out[n + '/obs'] = r['obs']; out[n + '/reward'] = np.float64(r['reward']); out[n + '/done'] = np.bool_(r['done'])
        out[n + '/total_force'] = np.float64(r['info']['total_force_on_human']); out[n + '/task_success'] = np.int32(r['info']['task_success'])
        out[n + '/lens'] = np.array([r['info'][k] for k in ('action_robot_len', 'action_human_len', 'obs_robot_len', 'obs_human_len')], dtype=np.int32)
        out[n + '/extras'] = extras_of(r['env'], rb.TASK_OF_KIND[b.task_kind])
        out[n + '/state_out'] = r['state']
        if b.task_kind == 5: out[n + '/cloth_out'] = r['cloth']         # the water after the step (the garment's nodes are not kept: 2 x 3268 x 3 per case)
        out[n + '/meta'] = np.array([c['model'], '1' if c['coop'] else '0', c['variant']])
        w.close()
        print('%-46s reward %10.4f  force %9.3f  success %d' % (n, r['reward'], r['info']['total_force_on_human'], r['info']['task_success']))
    out['names'] = np.array(names)
    np.savez_compressed(os.path.join(GOLDEN, 'ref_steps.npz'), **out)
    print(len(names), 'step cases ->', os.path.join(GOLDEN, 'ref_steps.npz'))


def make_units():
    rb.install()
    envs = rb.reference_envs()
    from _agx_reference.envs.util import Util
    from _agx_reference.envs.human_creation import HumanCreation
    from _agx_reference.envs.env import AssistiveEnv
    rb._CURRENT[0] = None
    out = {}
    rng = np.random.RandomState(20260926)
    util = Util(0, np.random.RandomState(7))
    # ---- sleeve_on_arm_reward: random arms and sleeves; a third of the sleeves are rings around the forearm / the upper arm
    ins, outs = [], []
    for k in range(240):
        sh = rng.uniform(-0.3, 0.3, 3) + [0, 0, 1.0]
        el = sh + rng.uniform(-1, 1, 3) * [0.2, 0.2, 0.1] + [0, -0.1, -0.25]
        wr = el + rng.uniform(-1, 1, 3) * 0.15 + [0, -0.2, 0.05]
        rad = [0.043, 0.0355][k % 2]
        kind = k % 3
        if kind == 0:
            pts = rng.uniform(-0.4, 0.4, (6, 3)) + el
        else:
            p0, p1 = (el, wr) if kind == 1 else (sh, el)
            ax = (p1 - p0) / np.linalg.norm(p1 - p0)
            u = np.cross(ax, rng.uniform(-1, 1, 3)); u /= np.linalg.norm(u); v = np.cross(ax, u)
            mid = p0 + rng.uniform(0.1, 0.9) * (p1 - p0)
            r = rng.uniform(0.06, 0.2); ph = rng.uniform(0, 2 * np.pi)
            pts = np.array([mid + rng.uniform(-0.03, 0.03) * ax + r * (np.cos(ph + a) * u + np.sin(ph + a) * v) for a in np.deg2rad([0, 120, 240, 60, 180, 300])])
            pts += rng.normal(0, 0.01, pts.shape)
        res = util.sleeve_on_arm_reward(pts[:3], pts[3:], sh, el, wr, rad, rad, rad)
        ins.append(np.concatenate([pts.ravel(), sh, el, wr, [rad]])); outs.append([float(x) for x in res])
    out['sleeve_in'], out['sleeve_out'] = np.array(ins), np.array(outs)
    print('sleeve_on_arm_reward: forearm_in %d, upperarm_in %d of %d' % (out['sleeve_out'][:, 0].sum(), out['sleeve_out'][:, 1].sum(), len(outs)))
    # ---- line_intersects_triangle
    tri_in = rng.uniform(-1, 1, (300, 15)); tri_in[:150, 9:12] = tri_in[:150, :9].reshape(150, 3, 3).mean(axis=1) + rng.uniform(-1, 1, (150, 3)) * [0.05, 0.05, 1.0]
    tri_in[:150, 12:15] = 2 * tri_in[:150, :9].reshape(150, 3, 3).mean(axis=1) - tri_in[:150, 9:12]
    out['tri_in'] = tri_in
    out['tri_out'] = np.array([bool(util.line_intersects_triangle(t[0:3], t[3:6], t[6:9], t[9:12], t[12:15])) for t in tri_in])
    # ---- capsule_points as generate_targets calls it (bed_bathing.py:173-188) and point_on_capsule as generate_target does (scratch_itch.py:134-141)
    for g, (ul, ur, fl, fr) in (('male', (0.279, 0.043, 0.257, 0.033)), ('female', (0.264, 0.0355, 0.234, 0.027))):
        out['capsule_upper_' + g] = np.array(util.capsule_points(p1=np.array([0, 0, 0]), p2=np.array([0, 0, -ul]), radius=ur, distance_between_points=0.03))
        out['capsule_fore_' + g] = np.array(util.capsule_points(p1=np.array([0, 0, 0]), p2=np.array([0, 0, -fl]), radius=fr, distance_between_points=0.03))
        pts = []
        for seed in range(16):
            util.np_random = np.random.RandomState(seed)
            pts.append(util.point_on_capsule(p1=np.array([0, 0, 0]), p2=np.array([0, 0, -ul]), radius=ur, theta_range=(0, np.pi * 2)))
        out['point_on_capsule_' + g] = np.array(pts)
    # ---- human_preferences for the six task strings
    class _E(AssistiveEnv):
        def __init__(self, task):
            import configparser
            self.task = task
            self.configp = configparser.ConfigParser()
            self.configp.read(os.path.join(rb.REF_ROOT, 'assistive_gym', 'config.ini'))
            for a, t in (('C_v', 'velocity_weight'), ('C_f', 'force_nontarget_weight'), ('C_hf', 'high_forces_weight'), ('C_fd', 'food_hit_weight'),
                         ('C_fdv', 'food_velocities_weight'), ('C_d', 'dressing_force_weight'), ('C_p', 'high_pressures_weight')):
                setattr(self, a, self.config(t, 'human_preferences'))          # env.py:60-66
    class _T:
        def __init__(self, n): self.n = n
        def get_closest_points(self, human, distance): return [], [], [], [], [0.0] * self.n
    hp_in, hp_out = [], []
    tasks = ['feeding', 'drinking', 'bed_bathing', 'scratch_itch', 'dressing', 'arm_manipulation']
    for ti, task in enumerate(tasks):
        e = _E(task)
        for k in range(12):
            v, tf, tt = rng.uniform(0, 2), rng.uniform(0, 30), rng.uniform(0, 25)
            fh = -float(rng.randint(0, 3)); fv = rng.uniform(0, 1, rng.randint(0, 4))
            df = rng.uniform(-5, 5, (rng.randint(1, 6), 3))
            am = rng.uniform(0, 20, 2); amt = am.sum() + rng.uniform(0, 10); n0, n1 = int(rng.randint(0, 5)), int(rng.randint(0, 5))
            e.tool_right, e.tool_left, e.human = _T(n0), _T(n1), None
            r = e.human_preferences(end_effector_velocity=v, total_force_on_human=tf + tt, tool_force_at_target=tt, food_hit_human_reward=fh, food_mouth_velocities=list(fv),
                                    dressing_forces=df, arm_manipulation_tool_forces_on_human=list(am), arm_manipulation_total_force_on_human=amt)
            hp_in.append([ti, v, tf + tt, tt, fh, fv.sum(), np.linalg.norm(df, axis=-1).sum(), am[0], am[1], amt, n0, n1]); hp_out.append(float(r))
    out['pref_in'], out['pref_out'], out['pref_tasks'] = np.array(hp_in), np.array(hp_out), np.array(tasks)
    # ---- config.ini as AssistiveEnv.config reads it
    e = _E('feeding')
    for sec in e.configp.sections():
        for key in e.configp[sec]:
            out['config/%s/%s' % (sec, key)] = np.float64(e.config(key, sec))
    # ---- create_human under the recording createMultiBody
    for gender in ('male', 'female'):
        for cloth in (False, True):
            for ls in (1.0, 0.7):
                rb.RECORD['shapes'].clear(); rb.RECORD['bodies'].clear(); rb.RECORD['filter'].clear()
                hc = HumanCreation(0, np_random=np.random.RandomState(1), cloth=cloth)
                hc.create_human(static=True, limit_scale=ls, specular_color=[0.1, 0.1, 0.1], gender=gender, config=e.config)
                b = rb.RECORD['bodies'][-1]; sh = rb.RECORD['shapes']
                key = 'human/%s/%d/%.1f/' % (gender, int(cloth), ls)
                out[key + 'mass'] = np.array(b['linkMasses'], dtype=np.float64)
                out[key + 'pos'] = np.array(b['linkPositions'], dtype=np.float64)
                out[key + 'parent'] = np.array(b['linkParentIndices'], dtype=np.int32)
                out[key + 'jtype'] = np.array(b['linkJointTypes'], dtype=np.int32)
                out[key + 'axis'] = np.array(b['linkJointAxis'], dtype=np.float64)
                out[key + 'lower'] = np.array(b['linkLowerLimits'], dtype=np.float64); out[key + 'upper'] = np.array(b['linkUpperLimits'], dtype=np.float64)
                out[key + 'base_pos'] = np.array(b['basePosition'], dtype=np.float64)
                shape_of = [b['baseCollisionShapeIndex']] + list(b['linkCollisionShapeIndices'])
                rows = []
                for si in shape_of:               # [type, radius, height, frame pos(3), frame quat(4), mesh scale]
                    if si < 0: rows.append([-1] + [0.0] * 10); continue
                    s_ = sh[si]
                    rows.append([s_['shapeType'], s_.get('radius', 0.0), s_.get('height', 0.0)] + list(s_.get('collisionFramePosition', [0, 0, 0])) +
                                list(s_.get('collisionFrameOrientation', [0, 0, 0, 1])) + [np.mean(s_.get('meshScale', [0.0]))])
                out[key + 'shapes'] = np.array(rows, dtype=np.float64)
                out[key + 'radii'] = np.array([hc.hand_radius, hc.elbow_radius, hc.shoulder_radius])
                on = sorted({(la, lb) for (_, _, la, lb, en) in rb.RECORD['filter'] if en})
                off_after = set()
                state = {}
                for (_, _, la, lb, en) in rb.RECORD['filter']: state[(la, lb)] = en
                out[key + 'self_collision'] = np.array(sorted(k for k, v in state.items() if v), dtype=np.int32)
    np.savez_compressed(os.path.join(GOLDEN, 'ref_units.npz'), **out)
    print('unit fixtures ->', os.path.join(GOLDEN, 'ref_units.npz'), len(out), 'arrays')


if __name__ == '__main__':
    assert rb.available(), 'needs /root/reference'
    which = sys.argv[1:] or ['steps', 'units']
    if 'units' in which: make_units()
    if 'steps' in which: make_steps()
