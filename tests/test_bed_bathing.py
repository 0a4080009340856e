"""BedBathingSawyer-v1 (BASELINE config 3) without a GPU: the model blob against the reference's numbers, the oracle's task
layer against an independent numpy restatement of bed_bathing.py, and the device code (bed_bathing kernel variant compiled
for the CPU wave emulator) against the oracle.  PARITY UNPINNED vs PyBullet as everywhere (oracle/agx_oracle.h)."""
import numpy as np
import pytest

from bed_util import arm_points, move_pad_to, pad_pose, target_world_positions


@pytest.fixture(scope='module')
def bed():
    from assistive_gym_amd.blob import ModelBlob
    return ModelBlob.load('bed_bathing_sawyer')


@pytest.fixture(scope='module')
def bed_oracle(bed):
    from oracle_lib import Oracle
    return Oracle(bed)


@pytest.fixture(scope='module')
def bed_emu(bed):
    from emu_lib import Emu
    return Emu(bed)


def _states(bed, n, seed, **kw):
    from assistive_gym_amd.host.reset_bed import make_states
    return make_states(bed, n, seed=seed, **kw)


def wiping_state(bed, bed_oracle, seed=1001, depth=0.004, along=0.5, arm='fore'):
    """a post-reset state with the right arm abducted (so that the pad does not reach the trunk) and the wiping pad pressed
    `depth` into the top of the forearm / upper arm"""
    st, infos = _states(bed, 1, seed, human_q_override={3: np.deg2rad(70)})
    s = st[0].copy()
    sh, el, wr, _ = arm_points(bed, bed_oracle, s)
    a, b = (el, wr) if arm == 'fore' else (sh, el)
    c = bed.collider([k for k in range(*bed.meta['ranges']['human_male' if infos[0]['gender'] == 'male' else 'human_female'])
                      if bed.collider(k)['link'] == (7 if arm == 'fore' else 5)][0])
    move_pad_to(bed, s, a + along * (b - a) + np.array([0, 0, c['radius'] + 0.0025 - depth]))
    return s


# ---- model data ------------------------------------------------------------------------------------------------------
def test_model_header_and_tables(bed):
    from assistive_gym_amd.model import compiler as L
    assert bed.task_kind == L.TASK_BED_BATHING
    assert (bed.ndof, bed.nrobot, bed.nhdof, bed.nfree, bed.act_dim, bed.obs_dim) == (20, 10, 10, 1, 7, 24)    # bed_bathing.py:10: 17 + 7
    # Sawyer (agents/sawyer.py:8-17): arm joints, gripper, PyBullet numbering of the movable joints
    assert [bed.robot_i(d, 'PB_INDEX') for d in range(10)] == [3, 4, 8, 9, 10, 11, 13, 16, 20, 22]
    assert [bed.robot_i(d, 'ACT') for d in range(10)] == [0, -1, 1, 2, 3, 4, 5, 6, -1, -1]
    assert [bed.robot_i(d, 'JTYPE') for d in range(10)] == [0] * 8 + [1, 1]                                   # prismatic fingers
    assert np.isclose(bed.robot_f(8, 'QT0'), 0.0125) and np.isclose(bed.robot_f(9, 'QT0'), -0.0125)            # sawyer.py:21
    assert np.isclose(bed.robot_f(0, 'KP'), 0.05) and np.isclose(bed.robot_f(0, 'MAXF'), 1.0)                  # robot.py:36-37
    assert np.isclose(sum(bed.robot_f(d, 'MASS') for d in range(10)), 81.8 - 60.864 - 2.0687 - 0.0001, atol=0.05)   # 81.8 kg incl. the fixed base links
    # the human's right arm chain (human.right_arm_joints), masses of pecs / upper arm / forearm / hand (human_creation.py:189,203)
    assert [bed.robot_i(d, 'PB_INDEX') for d in range(10, 20)] == list(range(10))
    m = 78.4
    assert np.allclose([bed.robot_f(d, 'MASS') for d in (12, 15, 17, 19)], [0.05 * m, 0.033 * m, 0.019 * m, 0.0065 * m], rtol=1e-6)
    assert np.isclose(bed.param('HUMAN_GRAVITY_Z'), -1.0) and bed.param('ROBOT_GRAVITY_Z') == 0.0             # bed_bathing.py:162-164
    # wiper: three 0.1 kg links welded together, link 1 = the wiping pad, 3.9 cm below the handle frame
    assert np.isclose(bed.free_f(0, 'MASS'), 0.3) and np.allclose(bed.task_f('TOOL_OBS_POS', 3), [0, 0, -0.039])
    r = bed.meta['ranges']
    assert [bed.collider(c)['link'] for c in range(*r['tool'])] == [-1, 0, 1] and bed.task_i('PAD_LINK') == 0b100    # bitmask over link + 1: link 1
    assert all(bed.collider(c)['friction'] == 5.0 for c in range(*r['bed']))                                   # bed_bathing.py:116
    assert bed.task_i_n('NT', 4) == [81, 48, 56, 35]                                                         # 129 / 91 targets
    assert np.isclose(bed.task_f('W_WIPE'), 5.0) and np.isclose(bed.task_f('SUCCESS_FRAC'), 0.3)              # config.ini:9-13


def test_targets_lie_on_the_arm_capsules(bed):
    """generate_targets (bed_bathing.py:173-188): every target sits on the surface of its capsule's cylinder"""
    from assistive_gym_amd.model.human import HumanModel
    ntmax = bed.task_i('NT_MAX')
    for g, gender in enumerate(('male', 'female')):
        hm = HumanModel(gender)
        nts = bed.task_i_n('NT', 4)[2 * g:2 * g + 2]
        o = bed.h['OFF_TARGETS'] + 4 * g * ntmax
        tab = bed.f[o:o + 4 * sum(nts)].reshape(-1, 4)
        arm = bed.i[o:o + 4 * sum(nts)].reshape(-1, 4)[:, 3]
        assert list(arm) == [0] * nts[0] + [1] * nts[1]
        for a, key in ((0, 'upperarm'), (1, 'forearm')):
            rad, length = hm.dims[key]
            p = tab[arm == a, :3]
            assert np.allclose(np.hypot(p[:, 0], p[:, 1]), rad, atol=1e-6) and (p[:, 2] < 0).all() and (p[:, 2] > -length).all()


def test_reset_sampler_properties(bed, bed_oracle):
    st, infos = _states(bed, 6, 3001)
    v = bed.view(st)
    assert (v['total_food'] == np.where(v['gender'] == 0, 129, 91)).all()
    for i in range(6):
        nt = int(v['total_food'][i])
        alive = v['task'][i].view(np.uint32)
        assert sum(bin(int(w)).count('1') for w in alive) == nt
        # the TOC search found a base from which the start pose is reached: the pad starts near the target pose, over the bed
        p, _ = pad_pose(bed, st[i])
        assert np.linalg.norm(p - infos[i]['target_ee_pos']) < 0.25
        assert (v['frozen'][i] != 0) == (infos[i]['impairment'] != 'tremor')
        sh, el, wr, _ = arm_points(bed, bed_oracle, st[i])
        assert 0.75 < sh[2] < 1.05 and 0.75 < wr[2] < 1.05                     # lying on the mattress
    o = bed_oracle.observe(st[0])
    assert o.shape == (24,) and np.isfinite(o).all() and o[-1] == 0


# ---- oracle task layer vs an independent numpy restatement of bed_bathing.py ---------------------------------------------
def test_oracle_reward_decomposition_and_wiping(bed, bed_oracle):
    s = wiping_state(bed, bed_oracle)
    tw = target_world_positions(bed, bed_oracle, s)
    wiped_total = 0
    for k in range(3):
        a = np.random.RandomState(k).uniform(-1, 1, 7).astype(np.float32) * 0.2
        pre = s.copy()
        before = bed.view(pre.reshape(1, -1))['task'][0].view(np.uint32).copy()
        obs, rew, done, info = bed_oracle.step(s, a)
        v = bed.view(s.reshape(1, -1))
        after = v['task'][0].view(np.uint32)
        newly = [t for t in range(len(tw)) if (before[t >> 5] >> (t & 31) & 1) and not (after[t >> 5] >> (t & 31) & 1)]
        assert len(newly) == int(info[4])                                     # new_contact_points
        wiped_total += len(newly)
        assert int(v['task_success'][0]) == wiped_total
        # the wiped targets are those near the pad (within the pad's half diagonal + 2.5 cm of its centre at some time of the step)
        pp, _ = pad_pose(bed, s)
        tw = target_world_positions(bed, bed_oracle, s)
        for t in newly:
            assert np.linalg.norm(tw[t] - pp) < 0.05 + 0.025 + 0.02
        # reward = distance + action + wiping + preferences (bed_bathing.py:20-27, env.py:237-274 non-feeding branch)
        total_f, robot_f, pad_f, pref = info[0], info[2], info[3], info[5]
        assert pad_f <= total_f + 1e-6 and obs[-1] >= pad_f - 1e-6             # tool_force (all tool contacts) >= tool_force_on_human
        dist_term = rew - (0.01 * -np.linalg.norm(a) + 5.0 * info[4] + pref)
        assert -0.02 < -dist_term < 0.02                                      # the pad touches the arm: |closest distance| is millimetres
        # preferences: -0.25 v_ee - 0.01 (total - pad) - 0.05 [pad >= 10] pad, so pref + 0.01 (total - pad) + ... = -0.25 v <= 0
        hf = 0.0 if pad_f < 10 else -pad_f
        v_ee = -(pref + 0.01 * (total_f - pad_f) - 0.05 * hf) / 0.25
        assert v_ee >= -1e-6
        assert not done
    assert wiped_total >= 2


def test_oracle_distance_reward_is_the_closest_tool_human_distance(bed, bed_oracle):
    """away from the human the reward's distance term is -(smallest gap between any wiper box and any human collider)"""
    st, _ = _states(bed, 2, 1001)
    s = st[0].copy()
    a = np.zeros(7, dtype=np.float32)
    obs, rew, done, info = bed_oracle.step(s, a)
    dist = -(rew - info[5])                                                   # no action, no wiping
    # brute force from sampled surface points of the boxes vs capsule axes: a coarse upper bound that must be close
    from assistive_gym_amd.model import xform as X
    v = bed.view(s.reshape(1, -1))
    fp, fq = v['free'][0, 0, :3].astype(float), v['free'][0, 0, 3:7].astype(float)
    pos, rot = bed_oracle.fk(s)
    g = int(v['gender'][0])
    best = 1e9
    for c in range(*bed.meta['ranges']['tool']):
        tv = X.apply(fp, fq, bed.collider(c)['verts'])
        grid = np.array([tv[0] + u * (tv[4] - tv[0]) + w * (tv[2] - tv[0]) + t * (tv[1] - tv[0]) for u in np.linspace(0, 1, 5) for w in np.linspace(0, 1, 5) for t in (0, 1)])
        for h in range(*bed.meta['ranges']['human_male' if g == 0 else 'human_female']):
            hc = bed.collider(h)
            if len(hc['verts']) > 2:
                continue
            body = hc['body']
            if body >= 300:
                hp, hq = v['human'][0, body - 300, :3].astype(float), v['human'][0, body - 300, 3:].astype(float)
                hv = X.apply(hp, hq, hc['verts'])
            else:
                hv = hc['verts'] @ rot[body].T + pos[body]
            p0, p1 = hv[0], hv[-1]
            for q in grid:
                t = 0.0 if len(hv) == 1 else np.clip(np.dot(q - p0, p1 - p0) / max(np.dot(p1 - p0, p1 - p0), 1e-12), 0, 1)
                best = min(best, np.linalg.norm(q - (p0 + t * (p1 - p0))) - hc['radius'])
    assert dist <= best + 1e-6 and best - dist < 0.02


# ---- device code on the wave emulator vs the oracle ----------------------------------------------------------------------
def _compare(bed, o, e, s, actions, tol=2e-5):
    so, se = s.copy(), s.copy()
    for a in actions:
        oo, orr, od, oi = o.step(so, a)
        eo, er, ed, ei, _ = e.step(se, a)
        assert oi[6] == ei[6] and oi[7] == ei[7], 'same contacts, same rows'
        assert np.abs(oo - eo).max() < tol and abs(orr - er) < tol * max(1.0, abs(orr)) and od == ed
        assert oi[4] == ei[4] and oi[1] == ei[1]                              # targets wiped, task success flag
        for k in (0, 2, 3):
            assert abs(oi[k] - ei[k]) <= 1e-3 * max(1.0, abs(oi[k]))          # forces: 1e-3 relative (north star)
        vo, ve = bed.view(so.reshape(1, -1)), bed.view(se.reshape(1, -1))
        assert np.abs(vo['q'] - ve['q']).max() < tol and np.array_equal(vo['task'], ve['task']) and vo['task_success'][0] == ve['task_success'][0]
    return so, se


def test_emulator_free_space(bed, bed_oracle, bed_emu):
    st, _ = _states(bed, 2, 1001)
    for i in range(2):
        _compare(bed, bed_oracle, bed_emu, st[i], [np.random.RandomState(10 * i + k).uniform(-1, 1, 7).astype(np.float32) for k in range(3)])


def test_emulator_wiping_contact(bed, bed_oracle, bed_emu):
    s = wiping_state(bed, bed_oracle)
    so, se = _compare(bed, bed_oracle, bed_emu, s, [np.random.RandomState(k).uniform(-1, 1, 7).astype(np.float32) * 0.2 for k in range(3)], tol=1e-4)
    assert bed.view(so.reshape(1, -1))['task_success'][0] >= 2


def test_emulator_tremor_arm_is_dynamic(bed, bed_oracle, bed_emu):
    st, infos = _states(bed, 1, 2001, impairment='tremor')
    q0 = bed.view(st[0].reshape(1, -1))['q'][0, 10:].copy()
    so, se = _compare(bed, bed_oracle, bed_emu, st[0], [np.random.RandomState(k).uniform(-1, 1, 7).astype(np.float32) for k in range(4)])
    assert np.abs(bed.view(so.reshape(1, -1))['q'][0, 10:] - q0).max() > 1e-3     # tremor targets move the arm (env.py:212-215)


def test_emulator_observe(bed, bed_oracle, bed_emu):
    st, _ = _states(bed, 1, 1005)
    assert np.abs(bed_oracle.observe(st[0]) - bed_emu.observe(st[0])).max() < 1e-5


# ---- pose-dependent arm limits (human.py:134-152) and the co-op flavour ----------------------------------------------------
def _mlp_numpy(bed, x):
    """the Keras model restated in numpy float64 from the blob's weights"""
    W = bed.f[bed.h['OFF_MLP']:bed.h['OFF_MLP'] + 8705].astype(np.float64)
    W1, b1, W2, b2 = W[:256].reshape(4, 64), W[256:320], W[320:4416].reshape(64, 64), W[4416:4480]
    W3, b3, W4, b4 = W[4480:8576].reshape(64, 64), W[8576:8640], W[8640:8704], W[8704]
    return np.tanh(np.tanh(np.tanh(x @ W1 + b1) @ W2 + b2) @ W3 + b3) @ W4 + b4


def _remap(a):
    """human.py:142-145 for the right arm"""
    tz, tx, ty, qe = a
    return np.array([(-tz + 2 * np.pi) % (2 * np.pi), (tx + 2 * np.pi) % (2 * np.pi), -ty, (-qe + 2 * np.pi) % (2 * np.pi)])


def test_h5_reader_and_blob_weights(bed):
    """assets/realistic_arm_limits_model.h5 read without h5py: 8,705 parameters, tanh x3 + sigmoid; the blob carries them"""
    import os
    from assistive_gym_amd.model.h5lite import H5File, load_keras_dense_stack
    path = '/root/reference/assistive_gym/envs/assets/realistic_arm_limits_model.h5'
    if not os.path.exists(path):
        pytest.skip('reference assets not on this box')
    ds = H5File(path).datasets()
    assert ds['/model_weights/dense_1/dense_1/kernel:0'].shape == (4, 64) and ds['/optimizer_weights/Adam/iterations:0'] == 425200
    stack = load_keras_dense_stack(path)
    assert [k.shape for k, _ in stack] == [(4, 64), (64, 64), (64, 64), (64, 1)] and sum(k.size + b.size for k, b in stack) == 8705
    flat = np.concatenate([np.concatenate([k.ravel(), b.ravel()]) for k, b in stack])
    assert np.array_equal(flat, bed.f[bed.h['OFF_MLP']:bed.h['OFF_MLP'] + 8705])
    raw = open(path, 'rb').read()
    assert raw.count(b'"activation": "tanh"') == 3 and raw.count(b'"activation": "sigmoid"') == 1


def test_oracle_classifier_matches_numpy(bed, bed_oracle):
    import ctypes as C
    from oracle_lib import lib
    L = lib(); L.agxo_arm_limit_logit.restype = C.c_double
    rng = np.random.RandomState(0)
    cls = []
    for k in range(500):
        x = np.array([rng.uniform(0, 2 * np.pi), rng.uniform(0, 2 * np.pi), rng.uniform(-1.6, 1.6), rng.uniform(0, 2.3)])
        z = L.agxo_arm_limit_logit(C.c_void_p(bed_oracle.h), x.ctypes.data_as(C.c_void_p))
        assert abs(z - _mlp_numpy(bed, x)) < 1e-9
        cls.append(z > 0)
    assert 0.1 < np.mean(cls) < 0.9                                           # both verdict classes occur


def _invalid_arm_pose(bed, rng):
    """an arm pose inside the joint limits that the classifier rejects, with a clear margin"""
    lo = np.array([bed.robot_f(bed.nrobot + j, 'LOWER') for j in (3, 4, 5, 6)]); hi = np.array([bed.robot_f(bed.nrobot + j, 'UPPER') for j in (3, 4, 5, 6)])
    for _ in range(10000):
        a = rng.uniform(lo, hi)
        if _mlp_numpy(bed, _remap(a)) < -2.0:
            return a
    raise AssertionError('no invalid pose found')


def test_coop_arm_limit_rollback(bed, bed_oracle, bed_emu):
    """co-op env (human arm controllable): an invalid arm pose is rolled back to the last valid one with zero velocity, on the
    oracle and on the device code alike; a valid pose is remembered"""
    from emu_lib import Emu
    from oracle_lib import Oracle
    coop = bed.coop()
    assert coop.is_coop and coop.task_i('ARM_LIMIT_ON') == 1 and (coop.act_dim, coop.obs_dim) == (17, 24 + 28)    # bed_bathing.py:10
    assert bed.task_i('ARM_LIMIT_ON') == 0
    o1, e1 = Oracle(coop.set_param('FRAME_SKIP', 1)), Emu(coop.set_param('FRAME_SKIP', 1))
    st, _ = _states(coop, 1, 7001, impairment='none')
    s = st[0].copy()
    v = coop.view(s.reshape(1, -1))
    nr = coop.nrobot
    assert v['frozen'][0] == 0                                                 # a controllable human stays dynamic (human.py:108)
    valid = v['q'][0, nr + 3:nr + 7].copy()
    assert _mlp_numpy(coop, _remap(valid.astype(np.float64))) > 0
    # first substep: the sampled pose is valid -> remembered
    a0 = np.zeros(17, dtype=np.float32)
    so, se = s.copy(), s.copy()
    o1.step(so, a0); e1.step(se, a0)
    for x in (so, se):
        vx = coop.view(x.reshape(1, -1))
        assert vx['task'][0, 10] == 1 and np.abs(vx['task'][0, 6:10].view(np.float32) - vx['q'][0, nr + 3:nr + 7]).max() == 0
    # teleport the arm into a rejected pose: the next substep rolls the four joints back
    bad = _invalid_arm_pose(coop, np.random.RandomState(3))
    for x in (so, se):
        vx = coop.view(x.reshape(1, -1))
        prev = vx['task'][0, 6:10].view(np.float32).copy()
        vx['q'][0, nr + 3:nr + 7] = bad; vx['qt'][0, nr + 3:nr + 7] = bad; vx['qd'][0, nr:] = 0
    oo, _, _, _ = o1.step(so, a0); eo, _, _, _, _ = e1.step(se, a0)
    for x in (so, se):
        vx = coop.view(x.reshape(1, -1))
        assert np.abs(vx['q'][0, nr + 3:nr + 7] - prev).max() < 1e-6 and np.abs(vx['qd'][0, nr + 3:nr + 7]).max() == 0
    assert np.abs(oo - eo).max() < 1e-4


def test_emulator_coop_episode(bed, bed_oracle, bed_emu):
    """17 actions (7 robot + 10 human arm joints), 24 + 28 observations; device code vs oracle with the classifier in the loop"""
    from emu_lib import Emu
    from oracle_lib import Oracle
    coop = bed.coop()
    o, e = Oracle(coop), Emu(coop)
    st, _ = _states(coop, 1, 7101)
    so, se = st[0].copy(), st[0].copy()
    for k in range(4):
        a = np.random.RandomState(50 + k).uniform(-1, 1, 17).astype(np.float32)
        oo, orr, od, oi = o.step(so, a)
        eo, er, ed, ei, _ = e.step(se, a)
        assert oo.shape == (52,) and np.abs(oo - eo).max() < 2e-5 and abs(orr - er) < 2e-5
        vo, ve = coop.view(so.reshape(1, -1)), coop.view(se.reshape(1, -1))
        assert np.abs(vo['q'] - ve['q']).max() < 2e-5 and np.array_equal(vo['task'][0, 10], ve['task'][0, 10])
    # the human part of the observation: joint angles of the 10 arm joints, then shoulder / elbow / wrist in the human's frame
    assert np.abs(oo[24 + 7:24 + 17] - vo['q'][0, coop.nrobot:]).max() < 1e-6


def test_noop_rule_is_off_where_the_robot_rests_on_the_person():
    """tests/golden/wiping_resting_arm_case.npz (written by the GPU test of the no-op re-test rule, session r04f): a Sawyer link RESTS on the
    person -- two contacts at dist = +1 um carrying 5.1 N.  The rule applied regardless (AGX_P_NOOP_PEN = 0) moves total_force_on_human by
    1.5e-3 relative against the plain 50-sweep solve in float64; with the switch (the default) every substep that has a robot / tool contact
    with the person is solved with plain sweeps and the step is IDENTICAL to the plain solve; the kernel sources (wave emulator) follow."""
    import os
    from assistive_gym_amd.blob import ModelBlob
    from emu_lib import Emu
    from oracle_lib import Oracle
    import conditioning as C
    d = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'wiping_resting_arm_case.npz'))
    b = ModelBlob.load('bed_bathing_sawyer')
    plain = Oracle(b.set_param('NOOP_RETEST', 0.0)).step(d['start'].copy(), d['action'])
    rule = Oracle(b).step(d['start'].copy(), d['action'])
    regardless = Oracle(b.set_param('NOOP_PEN', 0.0)).step(d['start'].copy(), d['action'])
    assert plain[3][0] > 5.0
    assert abs(regardless[3][0] - plain[3][0]) > 1e-3 * plain[3][0]
    assert rule[3][0] == plain[3][0] and np.array_equal(rule[0], plain[0]) and rule[1] == plain[1]
    emu = Emu(b).step(d['start'].copy(), d['action'])
    assert abs(emu[3][0] - plain[3][0]) <= max(1e-3 * plain[3][0], C.force_floor(b))
    old = Emu(b.set_param('NOOP_PEN', 0.0)).step(d['start'].copy(), d['action'])
    assert abs(float(d['dev_info'][0]) - old[3][0]) < 2e-3      # the device of that session still applied the rule here (5.1385 N): what the emulator gives without the switch
