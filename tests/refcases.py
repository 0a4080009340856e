"""The input side of the reference-pinned fixtures (tests/golden/ref_steps.npz): (model, state record, action) cases that exercise every
branch of the Python half of the step path -- plain / clipped / limit-clamped actions, tremor, co-op with per-env limit scale, the
arm-limit classifier incl. a roll-back, food eaten / spilled / hitting the person, robot and tool forces on the person, wiped targets,
scratches, the pressure term, the sleeve reward branches, the last step of an episode.  Built on the CPU from the host samplers and
the oracle (no reference needed); tests/diag/make_reference_fixtures.py feeds them to the reference's own step() (tests/refbridge).
"""
import numpy as np

from assistive_gym_amd.blob import ModelBlob
from assistive_gym_amd.model import compiler as L
from assistive_gym_amd.model import xform as X


def variant_blob(model, coop, variant):
    b = ModelBlob.load(model)
    if variant == 'piter0':          # a frozen garment: the six sleeve vertices stay where the case puts them
        w = b.words.copy()
        oc = b.h['OFF_CLOTH']
        w.view(np.float32)[oc + int(b.i[oc + L.CL['OFF_PARAM']]) + L.CP['PITER']] = 0.0
        b = ModelBlob(w, b.meta)
    return b.coop() if coop else b


def _oracle(b):
    from oracle_lib import Oracle
    return Oracle(b)


def _shift_robot(b, s, d):
    """translate the robot base, the tool(s) and the food with it"""
    v = b.view(s.reshape(1, -1))
    d = np.asarray(d, dtype=np.float32)
    v['base'][0, :3] += d
    for f in range(b.nfree):
        kind = int(b.i[b.h['OFF_FREE'] + f * L.F['STRIDE'] + L.F['KIND']])
        if kind != 2:                # not the bowl
            v['free'][0, f, :3] += d
    return s


def feeding_cases():
    from assistive_gym_amd.host.reset import make_states
    out = []
    for model in ('feeding_jaco', 'feeding_sawyer'):
        b = ModelBlob.load(model)
        o = _oracle(b)
        rng = np.random.RandomState(11)
        for imp, seed in (('none', 4101), ('limits', 4102), ('tremor', 4103), ('tremor', 4104)):
            if model != 'feeding_jaco' and imp != 'tremor':
                continue
            st, _ = make_states(b, 1, seed=seed, impairment=imp)
            s = st[0].copy(); o.settle(s, 25)
            for k in range(3):
                a = rng.uniform(-1, 1, b.act_dim).astype(np.float32)
                out.append(dict(name='%s_%s_%d_step%d' % (model, imp, seed, k), model=model, coop=False, variant='', state=s.copy(), cloth=None, action=a))
                o.step(s, a)
    b = ModelBlob.load('feeding_jaco'); o = _oracle(b)
    st, _ = make_states(b, 1, seed=4201, impairment='none')
    s = st[0].copy(); o.settle(s, 25)
    # actions beyond the box are clipped (env.py:187); a joint sitting next to its limit takes the clamp branch (env.py:203-210)
    a = np.array([3.0, -2.5, 0.4, 1.0, -1.0, 7.0, -0.2], dtype=np.float32)
    out.append(dict(name='feeding_clipped_action', model='feeding_jaco', coop=False, variant='', state=s.copy(), cloth=None, action=a))
    s2 = s.copy(); v = b.view(s2.reshape(1, -1))
    for d, side in ((1, 'UPPER'), (3, 'LOWER')):
        lim = b.robot_f(d, side)
        v['q'][0, d] = lim - 0.01 if side == 'UPPER' else lim + 0.01
        v['qt'][0, d] = v['q'][0, d]
    o.settle(s2, 2)
    a = np.array([0.2, 1.0, 0.1, -1.0, 0.3, 0.0, 0.0], dtype=np.float32)
    out.append(dict(name='feeding_limit_clamped_targets', model='feeding_jaco', coop=False, variant='', state=s2.copy(), cloth=None, action=a))
    # the last step of an episode (feeding.py:36)
    s3 = s.copy(); b.view(s3.reshape(1, -1))['iteration'][0] = 199
    out.append(dict(name='feeding_episode_end', model='feeding_jaco', coop=False, variant='', state=s3, cloth=None, action=np.zeros(7, dtype=np.float32)))
    # food: one particle spilled, one eaten (feeding.py:57-72), one resting against the person's head (:74-78)
    s4 = s.copy(); v = b.view(s4.reshape(1, -1)); f0 = b.h['FOOD0']
    v['free'][0, f0 + 0, :3] += np.array([-0.095, -0.095, 0.095], dtype=np.float32); v['free'][0, f0 + 0, 7:13] = 0
    v['free'][0, f0 + 1, :3] = v['target'][0] + np.array([0, 0, 0.045], dtype=np.float32); v['free'][0, f0 + 1, 7:13] = 0
    pos, rot = o.fk(s4)
    head = pos[b.task_i('HEAD_LINK')]
    v['free'][0, f0 + 2, :3] = (head + np.array([0.0, -0.02, 0.14])).astype(np.float32); v['free'][0, f0 + 2, 7:13] = 0
    out.append(dict(name='feeding_food_events', model='feeding_jaco', coop=False, variant='', state=s4, cloth=None, action=np.zeros(7, dtype=np.float32)))
    # the spoon pushed against the person's face: tool and robot forces on the human (feeding.py:45-48, env.py:241-250)
    for k, depth in enumerate((0.0, 0.01, 0.03)):
        s5 = s.copy(); v = b.view(s5.reshape(1, -1))
        tool = v['free'][0, 0, :3].astype(np.float64)
        want = v['target'][0].astype(np.float64) + np.array([0.0, -0.02 + depth, 0.0])
        _shift_robot(b, s5, want - tool)
        v['free'][0, :, 7:13] = 0
        a = np.array([0.1, -0.2, 0.3, 0.0, 0.1, -0.1, 0.2], dtype=np.float32) * k
        out.append(dict(name='feeding_spoon_on_face_%d' % k, model='feeding_jaco', coop=False, variant='', state=s5, cloth=None, action=a))
    # co-op: the head joints driven hard into their (scaled) limits; tremor in co-op (env.py:201-215)
    co = b.coop(); oc = _oracle(co)
    for imp, seed in (('limits', 4301), ('tremor', 4302), ('none', 4303)):
        st, _ = make_states(co, 1, seed=seed, impairment=imp)
        s = st[0].copy(); oc.settle(s, 25)
        rng = np.random.RandomState(seed)
        for k in range(3):
            a = rng.uniform(-1, 1, co.act_dim).astype(np.float32)
            a[7:] = np.sign(a[7:])
            out.append(dict(name='feeding_coop_%s_step%d' % (imp, k), model='feeding_jaco', coop=True, variant='', state=s.copy(), cloth=None, action=a))
            oc.step(s, a)
    return out


def bed_cases():
    from assistive_gym_amd.host.reset_bed import make_states
    from bed_util import arm_points, move_pad_to
    out = []
    b = ModelBlob.load('bed_bathing_sawyer'); o = _oracle(b)
    rng = np.random.RandomState(21)
    for imp, seed in (('none', 5101), ('tremor', 5102), ('limits', 5103)):
        st, _ = make_states(b, 1, seed=seed, impairment=imp)
        s = st[0].copy()
        for k in range(2):
            a = rng.uniform(-1, 1, b.act_dim).astype(np.float32)
            out.append(dict(name='bed_%s_step%d' % (imp, k), model='bed_bathing_sawyer', coop=False, variant='', state=s.copy(), cloth=None, action=a))
            o.step(s, a)
    # the pad pressed onto the forearm / the upper arm: wiped targets, tool force at the target (bed_bathing.py:41-78)
    for arm, along, depth in (('fore', 0.5, 0.004), ('fore', 0.3, 0.008), ('upper', 0.6, 0.004)):
        st, infos = make_states(b, 1, seed=5201, human_q_override={3: np.deg2rad(70)})
        s = st[0].copy()
        sh, el, wr, _ = arm_points(b, o, s)
        p0, p1 = (el, wr) if arm == 'fore' else (sh, el)
        c = b.collider([k for k in range(*b.meta['ranges']['human_male' if infos[0]['gender'] == 'male' else 'human_female'])
                        if b.collider(k)['link'] == (7 if arm == 'fore' else 5)][0])
        move_pad_to(b, s, p0 + along * (p1 - p0) + np.array([0, 0, c['radius'] + 0.0025 - depth]))
        for k in range(2):
            a = (rng.uniform(-1, 1, b.act_dim) * 0.3).astype(np.float32)
            out.append(dict(name='bed_wiping_%s_%.1f_step%d' % (arm, along, k), model='bed_bathing_sawyer', coop=False, variant='', state=s.copy(), cloth=None, action=a))
            o.step(s, a)
    # co-op on the PR2 (17 actions): the arm-limit classifier remembers a valid pose, then rolls an invalid one back (human.py:134-152)
    co = ModelBlob.load('bed_bathing_pr2').coop(); oc = _oracle(co)
    st, _ = make_states(co, 1, seed=5301, impairment='none')
    s = st[0].copy()
    a = rng.uniform(-1, 1, co.act_dim).astype(np.float32)
    out.append(dict(name='bed_coop_valid_pose', model='bed_bathing_pr2', coop=True, variant='', state=s.copy(), cloth=None, action=a))
    oc.step(s, a)
    v = co.view(s.reshape(1, -1)); nr = co.nrobot
    lo = np.array([co.robot_f(nr + j, 'LOWER') for j in (3, 4, 5, 6)]); hi = np.array([co.robot_f(nr + j, 'UPPER') for j in (3, 4, 5, 6)])
    import ctypes as C
    r2 = np.random.RandomState(3)
    for _ in range(10000):
        bad = r2.uniform(lo, hi)
        x = np.array([(-bad[0]) % (2 * np.pi), bad[1] % (2 * np.pi), -bad[2], (-bad[3]) % (2 * np.pi)])
        oc.L.agxo_arm_limit_logit.restype = C.c_double
        if oc.L.agxo_arm_limit_logit(C.c_void_p(oc.h), x.ctypes.data_as(C.c_void_p)) < -2.0:
            break
    v['q'][0, nr + 3:nr + 7] = bad; v['qt'][0, nr + 3:nr + 7] = bad; v['qd'][0, nr:] = 0; v['tremor_target'][0][3:7] = bad
    out.append(dict(name='bed_coop_rollback', model='bed_bathing_pr2', coop=True, variant='', state=s.copy(), cloth=None, action=np.zeros(co.act_dim, dtype=np.float32)))
    return out


def scratch_cases():
    from assistive_gym_amd.host.reset_scratch import make_states
    out = []
    rng = np.random.RandomState(31)
    for model, coop in (('scratch_itch_pr2', True), ('scratch_itch_jaco', False), ('scratch_itch_pr2', False)):
        b = ModelBlob.load(model); b = b.coop() if coop else b
        o = _oracle(b)
        for imp, seed in (('none', 6101), ('tremor', 6102)):
            st, _ = make_states(b, 1, seed=seed, impairment=imp)
            s = st[0].copy()
            for k in range(2):
                a = rng.uniform(-1, 1, b.act_dim).astype(np.float32)
                out.append(dict(name='%s%s_%s_step%d' % (model, '_coop' if coop else '', imp, k), model=model, coop=coop, variant='', state=s.copy(), cloth=None, action=a))
                o.step(s, a)
        if b.meta.get('mount') == 'wheelchair':
            continue         # the crafted contact below moves the robot base, which a wheelchair-mounted arm cannot do without hitting the chair
        # the tip pressed into the skin at the target: scratches count when the tip moved > 1 cm (scratch_itch.py:28-32,46-57)
        # (seed 6201: the tip leaves the skin within the step -- a scratch counts with zero force at the target, scratch_itch.py:28-32; seed 6213:
        # the tip stays pressed on the target through all three steps: tool_force_at_target > 0 in the observation and the preferences)
        for tag, cseed in (('scratching', 6201), ('scratchingb', 6213)):
          st, _ = make_states(b, 1, seed=cseed, impairment='none')
          s = st[0].copy()
          v = b.view(s.reshape(1, -1))
          pos, rot = o.fk(s)
          limb = b.task_i_n('ARM_LINK', 2)[int(v['task'][0][L.SI['LIMB']])]
          lp, lR = pos[limb], rot[limb]
          tgt = lR @ s[b.h['S_TASK']:b.h['S_TASK'] + 3].astype(np.float64) + lp
          axis = lR @ np.array([0, 0, -1.0])
          radial = (tgt - lp) - np.dot(tgt - lp, axis) * axis; radial /= np.linalg.norm(radial)
          fp, fq = v['free'][0, 0, :3].astype(np.float64), v['free'][0, 0, 3:7].astype(np.float64)
          bp, bq = X.compose(fp, fq, b.free_f(0, 'REFPOS', 3), b.free_f(0, 'REFQUAT', 4))
          tip, _ = X.compose(bp, bq, b.task_f('TOOL_OBS_POS', 3), b.task_f('TOOL_OBS_QUAT', 4))
          _shift_robot(b, s, tgt + radial * (0.01 - 0.003) - tip)
          v['free'][0, 0, 7:] = 0
          for k in range(3):
              a = (rng.uniform(-1, 1, b.act_dim) * 0.4).astype(np.float32)
              out.append(dict(name='%s%s_%s_step%d' % (model, '_coop' if coop else '', tag, k), model=model, coop=coop, variant='', state=s.copy(), cloth=None, action=a))
              o.step(s, a)
    return out


def arm_cases():
    from assistive_gym_amd.host.reset_arm import make_states
    out = []
    rng = np.random.RandomState(41)
    for model, coop in (('arm_manipulation_sawyer', False), ('arm_manipulation_pr2', False), ('arm_manipulation_sawyer', True)):
        b = ModelBlob.load(model); b = b.coop() if coop else b
        o = _oracle(b)
        fo = _oracle(b.set_param('HUMAN_GRAVITY_Z', -1.0))

        def fall(st, n):
            st = st.copy()
            for i in range(len(st)):
                fo.settle(st[i], n)
            return st
        st, _ = make_states(b, 1, seed=7101, arm_settler=fall, fall_steps=30)
        s = st[0].copy()
        for k in range(2):
            a = rng.uniform(-1, 1, b.act_dim).astype(np.float32)
            out.append(dict(name='%s%s_step%d' % (model, '_coop' if coop else '', k), model=model, coop=coop, variant='', state=s.copy(), cloth=None, action=a))
            o.step(s, a)
        if model != 'arm_manipulation_sawyer':
            continue
        # the scooper under the stretched-out forearm: tool forces on the person, contact points for the pressure term (env.py:259-272)
        st, infos = make_states(b, 1, seed=7201)
        s = st[0].copy(); v = b.view(s.reshape(1, -1)); nr = b.nrobot
        v['q'][0, nr + 3:nr + 7] = [np.deg2rad(60), 0.0, np.deg2rad(-90), 0.0]
        v['qt'][0, nr:] = v['q'][0, nr:]; v['tremor_target'][0] = v['q'][0, nr:]
        pos, rot = o.fk(s)
        el, wr = pos[nr + 7], pos[nr + 9]
        g = 'human_male' if infos[0]['gender'] == 'male' else 'human_female'
        rad = [b.collider(k) for k in range(*b.meta['ranges'][g]) if b.collider(k)['link'] == 7][0]['radius']
        fp, fq = v['free'][0, 0, :3].astype(np.float64), v['free'][0, 0, 3:7].astype(np.float64)
        hv = np.concatenate([X.apply(fp, fq, b.collider(c)['verts']) for c in range(*b.meta['ranges']['tool'])])
        top = hv[np.argmax(hv[:, 2])]
        _shift_robot(b, s, 0.5 * (el + wr) - np.array([0, 0, rad + 0.0025 - 0.003]) - top)
        for k in range(2):
            a = (rng.uniform(-1, 1, b.act_dim) * 0.3).astype(np.float32)
            out.append(dict(name='%s%s_lifting_step%d' % (model, '_coop' if coop else '', k), model=model, coop=coop, variant='', state=s.copy(), cloth=None, action=a))
            o.step(s, a)
    return out


def dressing_cases():
    from assistive_gym_amd.host.reset_dressing import make_states
    out = []
    rng = np.random.RandomState(51)
    for coop in (False, True):
        # the sleeve branches of the reward on a frozen garment (util.py:134-202, dressing.py:48-56)
        b = variant_blob('dressing_baxter', coop, 'piter0'); o = _oracle(b)
        st, cloth, infos = make_states(b, 1, 53)
        oc = b.h['OFF_CLOTH']
        tri = [int(x) for x in b.i[oc + L.CL['TRI']:oc + L.CL['TRI'] + 6]]
        links = b.task_i_n('OBS_LINK', 3)
        for where in ('away', 'forearm', 'upperarm'):
            s, c = st[0].copy(), cloth[0].copy()
            b.view(s.reshape(1, -1))['task'][0, L.DR['CLOTH_GRAVITY']] = 0
            c[0] += np.array([0, 0, 5.0], dtype=np.float32)
            if where != 'away':
                pos, _ = o.fk(s)
                sh, el, wr = pos[links[0]], pos[links[1]], pos[links[2]]
                p0, p1 = (el, wr) if where == 'forearm' else (sh, el)
                axis = (p1 - p0) / np.linalg.norm(p1 - p0)
                u = np.cross(axis, [0, 0, 1.0]); u /= np.linalg.norm(u)
                w = np.cross(axis, u)
                mid = p0 + 0.5 * (p1 - p0)
                ring = [mid + 0.15 * (np.cos(t) * u + np.sin(t) * w) for t in np.deg2rad([0, 120, 240])]
                ring2 = [mid + 0.02 * axis + 0.15 * (np.cos(t) * u + np.sin(t) * w) for t in np.deg2rad([60, 180, 300])]
                for n, p in zip(tri, ring + ring2):
                    c[0, n] = p; c[1, n] = 0
            a = rng.uniform(-1, 1, b.act_dim).astype(np.float32)
            out.append(dict(name='dressing%s_sleeve_%s' % ('_coop' if coop else '', where), model='dressing_baxter', coop=coop, variant='piter0', state=s, cloth=c, action=a))
    # the live garment: hanging from the gripper, then dragged over the forearm (cloth forces, dressing.py:34-46)
    b = variant_blob('dressing_baxter', False, ''); o = _oracle(b)
    st, cloth, infos = make_states(b, 1, 57)
    s, c = st[0].copy(), cloth[0].copy()
    o.settle_cloth(s, c, 6)
    b.view(s.reshape(1, -1))['task'][0, L.DR['CLOTH_GRAVITY']] = np.float32(-9.81).view(np.int32)
    a = rng.uniform(-1, 1, b.act_dim).astype(np.float32)
    out.append(dict(name='dressing_hanging', model='dressing_baxter', coop=False, variant='', state=s.copy(), cloth=c.copy(), action=a))
    pos, _ = o.fk(s)
    links = b.task_i_n('OBS_LINK', 3)
    el, wr = pos[links[1]], pos[links[2]]
    oc = b.h['OFF_CLOTH']
    tri = [int(x) for x in b.i[oc + L.CL['TRI']:oc + L.CL['TRI'] + 6]]
    c2 = c.copy()
    c2[0] += (0.5 * (el + wr) + np.array([0, 0, 0.02]) - c2[0, tri].mean(axis=0)).astype(np.float32)
    c2[1] = 0
    out.append(dict(name='dressing_on_forearm', model='dressing_baxter', coop=False, variant='', state=s.copy(), cloth=c2, action=(0.2 * a).astype(np.float32)))
    return out


def stretch_cases():
    """the Stretch (agents/stretch.py): take_step's branches for a mobile robot -- action_multiplier (env.py:196-197), action_duplication
    (env.py:218-220: one target for the four telescoping joints, clamped at joint 5's limit), per-joint gains and forces (stretch.py:49-50),
    the observation without the wheel angles and in the MOVING base frame (feeding.py:90-92, agent.py:60-64) -- for the four tasks it runs in"""
    from assistive_gym_amd.host import reset, reset_bed, reset_dressing, reset_scratch
    out = []
    rng = np.random.RandomState(61)
    b = ModelBlob.load('feeding_stretch'); o = _oracle(b)
    st, _ = reset.make_states(b, 1, seed=6101, impairment='none')
    s = st[0].copy(); o.settle(s, 25)
    for k in range(2):
        a = rng.uniform(-1, 1, b.act_dim).astype(np.float32)
        out.append(dict(name='feeding_stretch_step%d' % k, model='feeding_stretch', coop=False, variant='', state=s.copy(), cloth=None, action=a))
        o.step(s, a)
    out.append(dict(name='feeding_stretch_clipped_action', model='feeding_stretch', coop=False, variant='', state=s.copy(), cloth=None,
                    action=np.array([4.0, -3.0, 1.0, 2.5, -1.7], dtype=np.float32)))
    s2 = s.copy(); v = b.view(s2.reshape(1, -1))
    v['q'][0, 9:13] = 0.125; v['qt'][0, 9:13] = 0.125           # the telescoping joints next to their 0.13 m limit: the duplicated target takes the clamp branch
    out.append(dict(name='feeding_stretch_arm_at_limit', model='feeding_stretch', coop=False, variant='', state=s2, cloth=None,
                    action=np.array([0.3, -0.3, -0.5, 1.0, 0.2], dtype=np.float32)))
    bc = b.coop(); oc_ = _oracle(bc)
    st, _ = reset.make_states(bc, 1, seed=6102, impairment='tremor')
    s = st[0].copy(); oc_.settle(s, 25)
    out.append(dict(name='feeding_stretch_coop_tremor', model='feeding_stretch', coop=True, variant='', state=s.copy(), cloth=None,
                    action=rng.uniform(-1, 1, bc.act_dim).astype(np.float32)))
    for model, mod, seed in (('scratch_itch_stretch', reset_scratch, 6103), ('bed_bathing_stretch', reset_bed, 6104)):
        b = ModelBlob.load(model); o = _oracle(b)
        st = mod.make_states(b, 1, seed=seed, impairment='none')[0]
        s = st[0].copy(); o.settle(s, 10)                       # onto the wheels
        for k in range(2):
            a = rng.uniform(-1, 1, b.act_dim).astype(np.float32)
            out.append(dict(name='%s_step%d' % (model, k), model=model, coop=False, variant='', state=s.copy(), cloth=None, action=a))
            o.step(s, a)
    b = ModelBlob.load('scratch_itch_stretch').coop(); o = _oracle(b)
    st = reset_scratch.make_states(b, 1, seed=6105, impairment='limits')[0]
    s = st[0].copy(); o.settle(s, 10)
    out.append(dict(name='scratch_itch_stretch_coop', model='scratch_itch_stretch', coop=True, variant='', state=s.copy(), cloth=None,
                    action=rng.uniform(-1, 1, b.act_dim).astype(np.float32)))
    b = ModelBlob.load('dressing_stretch'); o = _oracle(b)
    st, cloth, _ = reset_dressing.make_states(b, 1, 6106)
    s, c = st[0].copy(), cloth[0].copy()
    o.settle_cloth(s, c, 6)
    b.view(s.reshape(1, -1))['task'][0, L.DR['CLOTH_GRAVITY']] = np.float32(-9.81).view(np.int32)
    out.append(dict(name='dressing_stretch_hanging', model='dressing_stretch', coop=False, variant='', state=s.copy(), cloth=c.copy(),
                    action=rng.uniform(-1, 1, b.act_dim).astype(np.float32)))
    return out


def drinking_cases():
    """DrinkingJaco (drinking.py): the water at rest in the cup under random actions, with tremor, the person co-acting; the cup tipping over
    (spills: -1 each, drinking.py:77-80); particles put at the mouth (+10 each, teleported, :66-75) and onto the person's chest (the
    water-hit count of the preferences, :84-88); the step that ends the episode."""
    from assistive_gym_amd.host.reset_drinking import make_states
    from assistive_gym_amd.model import compiler as L
    out = []
    b = variant_blob('drinking_jaco', False, '')
    o = _oracle(b)
    settled = {}
    for imp, seed in (('none', 3), ('tremor', 5)):
        st, water, _ = make_states(b, 1, seed=seed, impairment=imp)
        s, w = st[0].copy(), water[0].copy()
        o.settle_cloth(s, w, 50)                                                                    # drinking.py:176-177
        settled[imp] = (s.copy(), w.copy())
        rng = np.random.RandomState(seed)
        for k in range(2):
            a = rng.uniform(-1, 1, b.act_dim).astype(np.float32)
            out.append(dict(name='drinking_%s_step%d' % (imp, k), model='drinking_jaco', coop=False, variant='', state=s.copy(), cloth=w.copy(), action=a))
            o.step_cloth(s, w, a)
    # tipping the cup over: the wrist joints driven for 40 steps, then the steps in which the first particles leave the 0.1 m shell
    s, w = settled['none'][0].copy(), settled['none'][1].copy()
    a = np.zeros(7, np.float32); a[4], a[5], a[6] = 0.5, 1.0, 1.0
    taken = 0
    for k in range(90):
        s0, w0 = s.copy(), w.copy()
        obs, rew, done, info = o.step_cloth(s, w, a)
        if info[4] < 0 and taken < 3:
            out.append(dict(name='drinking_spilling_%d' % taken, model='drinking_jaco', coop=False, variant='', state=s0, cloth=w0, action=a.copy()))
            taken += 1
    assert taken == 3
    # at the mouth: three particles lifted out of the cup to within 0.03 m of the target, one of them moving
    s, w = settled['none'][0].copy(), settled['none'][1].copy()
    v = b.view(s[None])
    target = v['target'][0].astype(np.float64)
    top = np.argsort(-w[0][:, 2])[:4]
    fall = np.array([0.0, -0.015, 0.049])                                                          # one step is 0.1 s of free fall; clear of the face
    w[0][top[0]] = target + fall + [0.0, 0.0, 0.008]; w[0][top[1]] = target + fall + [0.013, 0.0, -0.004]; w[0][top[2]] = target + fall + [-0.012, 0.03, 0.0]
    w[1][top[0]] = w[1][top[1]] = 0.0; w[1][top[2]] = [0.0, -0.3, 0.0]
    b.view(s[None])['task_success'][0] = 46                                                         # 46 + 3 >= 0.75 x 64: info['task_success'] turns 1 (drinking.py:38)
    w[0][top[3]] = target + [0.0, -0.07, 0.06]; w[1][top[3]] = 0.0                                  # near the mouth but outside the 0.03 m ball: not drunk
    out.append(dict(name='drinking_at_the_mouth', model='drinking_jaco', coop=False, variant='', state=s.copy(), cloth=w.copy(), action=np.zeros(7, np.float32)))
    _, _, _, info = o.step_cloth(s.copy(), w.copy(), np.zeros(7, np.float32))
    assert info[4] == 29.0, info[4]                                                                 # three drunk, the fourth counted as spilled (0.1 m from the cup)
    # onto the person: three particles that left the cup earlier (no longer in `waters`, still in `waters_active`) land on the lap in this step
    s, w = settled['none'][0].copy(), settled['none'][1].copy()
    u = s.view(np.uint32); st0 = b.h['S_TASK']
    for j, i in enumerate(top[:3]):
        w[0][i] = target + [0.1 + 0.02 * (j - 1), -0.2, -0.402]; w[1][i] = [0.0, 0.0, -0.98]
        u[st0 + L.DK['ALIVE'] + (int(i) >> 5)] &= ~np.uint32(1 << (int(i) & 31))
    out.append(dict(name='drinking_water_on_the_person', model='drinking_jaco', coop=False, variant='', state=s.copy(), cloth=w.copy(), action=np.zeros(7, np.float32)))
    s1 = s.copy(); _, _, _, info = o.step_cloth(s1, w.copy(), np.zeros(7, np.float32))
    u1 = s1.view(np.uint32)
    assert info[4] == 0.0 and sum(bin(int(x)).count('1') for x in u1[st0 + L.DK['ACTIVE']:st0 + L.DK['ACTIVE'] + 2]) == 61
    # the last step of the episode
    s, w = settled['tremor'][0].copy(), settled['tremor'][1].copy()
    b.view(s[None])['iteration'][0] = 199
    # (not a zero action: the cup then stays at the IK target whose roll is exactly pi, the branch cut of getEulerFromQuaternion in the tilt term
    # -|roll - pi/2| (drinking.py:30-31) -- the reference's reward jumps by 0.1 pi between +pi and -pi there, and float32 lands on either side)
    out.append(dict(name='drinking_episode_end', model='drinking_jaco', coop=False, variant='', state=s, cloth=w, action=np.full(7, -0.5, np.float32)))
    # co-optimisation: the person's head joints act too
    co = variant_blob('drinking_jaco', True, '')
    st, water, _ = make_states(co, 1, seed=11, impairment='none')
    s, w = st[0].copy(), water[0].copy()
    oc = _oracle(co)
    oc.settle_cloth(s, w, 50)
    rng = np.random.RandomState(11)
    for k in range(2):
        a = rng.uniform(-1, 1, co.act_dim).astype(np.float32)
        out.append(dict(name='drinking_coop_step%d' % k, model='drinking_jaco', coop=True, variant='', state=s.copy(), cloth=w.copy(), action=a))
        oc.step_cloth(s, w, a)
    return out


def build_cases(tasks=('feeding', 'bed', 'scratch', 'arm', 'dressing', 'stretch', 'drinking')):
    fns = dict(feeding=feeding_cases, bed=bed_cases, scratch=scratch_cases, arm=arm_cases, dressing=dressing_cases, stretch=stretch_cases, drinking=drinking_cases)
    out = []
    for t in tasks:
        out += fns[t]()
    assert len({c['name'] for c in out}) == len(out)
    return out
