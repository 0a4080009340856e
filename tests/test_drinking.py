"""Drinking<Robot>-v1 (assistive_gym/envs/drinking.py, drinking_envs.py) -- model, CPU oracle, and the kernel sources on the CPU wave emulator (the GPU
tests are tests/test_zz_gpu_drinking.py): the
model blob against the reference's tables, the host reset, the water particles in the oracle (they come to rest in the cup, stay in it while
it tilts a little, pour out when it tips over), and the task layer's terms against a numpy restatement.  The reference's own step() runs
on this oracle through tests/refbridge (test_reference_pinned.py).  PARITY UNPINNED vs PyBullet (the physics half)."""
import numpy as np
import pytest

from assistive_gym_amd.model import compiler as L
from assistive_gym_amd.model import xform as X


@pytest.fixture(scope='module')
def dk():
    from assistive_gym_amd.blob import ModelBlob
    from oracle_lib import Oracle
    b = ModelBlob.load('drinking_jaco')
    return b, Oracle(b)


@pytest.fixture(scope='module')
def settled(dk):
    from assistive_gym_amd.host.reset_drinking import make_states
    b, o = dk
    st, water, infos = make_states(b, 1, seed=3, impairment='none')
    s, w = st[0].copy(), water[0].copy()
    o.settle_cloth(s, w, 50)                                                                        # drinking.py:176-177
    return s, w, infos[0]


def _in_cup(b, s, w):
    v = b.view(s[None])
    tp, tq = v['free'][0, 0, :3].astype(np.float64), v['free'][0, 0, 3:7].astype(np.float64)
    rel = np.array([X.quat_rotate(X.quat_conj(tq), p - tp) for p in w[0].astype(np.float64)])       # the cup's axis is its mesh y axis
    return (np.hypot(rel[:, 0], rel[:, 2]) < 0.04) & (rel[:, 1] > 0) & (rel[:, 1] < 0.13), rel


def test_model_tables(dk):
    b, o = dk
    assert b.task_kind == L.TASK_DRINKING and (b.act_dim, b.obs_dim, b.nhdof, b.nfree) == (7, 25, 4, 1) and b.h['SIM_SUBSTEPS'] == 4      # drinking.py:8,157
    assert b.param('NITER') == 10                                                                                                      # numSolverIterations (:157)
    arm = [d for d in range(b.nrobot) if b.robot_i(d, 'ACT') >= 0]
    assert np.isclose(b.robot_f(arm[0], 'KP'), 0.005)                                                                                  # :130
    assert np.allclose([b.robot_f(d, 'QT0') for d in range(b.nrobot) if b.robot_i(d, 'ACT') < 0], 0.63)                                # jaco.py:21
    assert np.allclose(b.task_f('TOOL_POS', 3), [0.05, -0.005, 0]) and np.allclose(b.task_f('TOOL_QUAT', 4), X.quat_from_rpy([0, -np.pi / 2, np.pi / 2]))   # jaco.py:27,32
    assert np.isclose(b.task_f('W_WIPE'), 0.1) and np.isclose(b.task_f('SUCCESS_FRAC'), 0.75) and np.isclose(b.task_f('TARGET_RADIUS'), 0.05)               # config.ini:24,26; drinking.py:64
    oc = b.h['OFF_CLOTH']
    assert b.i[oc + L.CL['NN']] == 64 and b.i[oc + L.CL['PARTICLES']] == 1 and b.i[oc + L.CL['NL']] == 0
    r = b.meta['ranges']
    assert r['tool'][1] - r['tool'][0] == 68                                                                                           # the cup's convex pieces
    x0 = b.f[oc + int(b.i[oc + L.CL['OFF_X0']]):oc + int(b.i[oc + L.CL['OFF_X0']]) + 192].reshape(64, 3)
    assert np.allclose(np.unique(np.round(x0[:, 0], 6)), [-0.02, -0.01, 0.0, 0.01]) and np.allclose(np.unique(np.round(x0[:, 2], 6)), [0.075, 0.085, 0.095, 0.105])   # :163-167


def test_env_ids_and_robot_tables():
    """the 12 ids of drinking_envs.py:15-67 and the per-robot 'drinking' entries of agents/<robot>.py"""
    from assistive_gym_amd.envs import ENV_IDS
    from assistive_gym_amd.blob import ModelBlob
    ids = sorted(k for k in ENV_IDS if k.startswith('Drinking'))
    assert ids == sorted('Drinking%s%s-v1' % (r, h) for r in ('PR2', 'Baxter', 'Sawyer', 'Jaco', 'Stretch', 'Panda') for h in ('', 'Human'))
    want = dict(jaco=([0.05, -0.005, 0], [0, -np.pi / 2, np.pi / 2], [0, np.pi / 2, 0], 7, 25), panda=([0.05, 0, 0.01], [0, -np.pi / 2, np.pi / 2], [0, np.pi / 2, 0], 7, 25),
                sawyer=([0.05, 0.125, 0], [0, 0, np.pi / 2], [0, -np.pi / 2, np.pi], 7, 25), baxter=([0.05, 0.125, 0], [0, 0, np.pi / 2], [0, -np.pi / 2, np.pi], 7, 25),
                pr2=([-0.01, 0, -0.05], [np.pi / 2, 0, 0], [0, 0, 0], 7, 25), stretch=([0, 0, -0.05], [np.pi / 2, 0, 0], [0, 0, np.pi / 2], 5, 21))
    for robot, (tool_pos, tool_rpy, ee_rpy, act, obs) in want.items():
        b = ModelBlob.load('drinking_' + robot)
        assert b.task_kind == L.TASK_DRINKING and (b.act_dim, b.obs_dim) == (act, obs) and b.h['SIM_SUBSTEPS'] == 4 and b.param('NITER') == 10, robot
        assert np.allclose(b.meta['ee_rpy'], ee_rpy) and b.has_reset_generator, robot
        if robot in ('jaco', 'panda', 'pr2', 'stretch'):             # the tool hangs off the end-effector link itself (TOOL_POS is then the table's entry)
            assert np.allclose(b.task_f('TOOL_POS', 3), tool_pos, atol=1e-6) and np.allclose(b.task_f('TOOL_QUAT', 4), X.quat_from_rpy(tool_rpy), atol=1e-6), robot
        d0 = next(d for d in range(b.nrobot) if b.robot_i(d, 'ACT') >= 0 and b.robot_i(d, 'PB_INDEX') not in (0, 1))
        assert np.isclose(b.robot_f(d0, 'KP'), 0.005) or robot == 'stretch'                                                            # drinking.py:130 (the Stretch keeps its own gains, stretch.py:49)
        cc = b.coop()
        assert (cc.act_dim, cc.obs_dim) == (act + 4, obs + 23)                                                                          # drinking.py:8: 19 + the 4 head joints


@pytest.mark.parametrize('robot', ['jaco', 'sawyer', pytest.param('pr2', marks=__import__('conftest').full), 'stretch'])
def test_device_reset_sampler_matches_its_restatement(robot):
    """DrinkingEnv.reset's sampling on the device (csrc/agx_reset.h on the wave emulator) against the numpy restatement (oracle/reset_oracle.py):
    the wheelchair-mounted arm by IK restarts, the free-standing robots by the base pose search with the mouth as a second START goal and the
    mouth with the start orientation as the further goal (drinking.py:143: goal kind 2), the Stretch by its placement draws; every water
    particle alive"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
    import reset_oracle as ro
    from assistive_gym_amd.blob import ModelBlob
    from emu_lib import Emu
    b = ModelBlob.load('drinking_' + robot)
    o, e = ro.with_collision_check(b.words), Emu(b)
    for seed in (1005, 77):
        st, info = o.sample(seed)
        se, ie = e.sample(seed)
        assert np.array_equal(st.view(np.uint32), se.view(np.uint32)) or np.abs(st - se).max() <= 1e-6, (robot, seed)
        assert info['ik_ok'] and ie[0] == 1.0
        if robot in ('sawyer', 'pr2'):
            assert info['toc']['goals_reached'] >= 2                                                     # the start pose AND the mouth (robot.py:196-200)
        v = b.view(st.reshape(1, -1))
        assert v['total_food'][0] == 64 and v['task'][0][L.DK['ALIVE']] == -1 and v['task'][0][L.DK['ACTIVE'] + 1] == -1
        assert np.linalg.norm(v['target'][0]) > 0.5 and v['free'][0, 0, 2] > (0.5 if robot == 'stretch' else 0.8)        # the mouth target; the cup in the hand (the Stretch's lift starts at 0.75 +- 0.1)


def test_water_rests_in_the_cup(dk, settled):
    b, o = dk
    s, w, info = settled
    assert info['ik_ok']
    inside, rel = _in_cup(b, s, w)
    assert inside.all() and np.linalg.norm(w[1], axis=1).max() < 0.02
    assert 0.010 < rel[:, 1].min() < 0.014 and rel[:, 1].max() < 0.045                      # on the bottom (6.5 mm + the radius), three to four layers
    d = np.linalg.norm(w[0][:, None] - w[0][None], axis=2) + np.eye(64)
    assert d.min() > 0.0085                                                                 # spheres of 5 mm radius: the projection leaves them at most 15 % compressed
    v = b.view(s[None])
    assert v['total_food'][0] == 64 and v['task'][0][L.DK['ALIVE']] == -1 and v['task'][0][L.DK['ALIVE'] + 1] == -1


def test_water_stays_when_tilted_a_little_and_pours_when_tipped_over(dk, settled):
    b, o = dk
    s, w, _ = settled
    s, w = s.copy(), w.copy()
    a = np.zeros(7, np.float32); a[4], a[5], a[6] = 0.5, 1.0, 1.0
    water_reward = 0.0
    for k in range(25):
        obs, rew, done, info = o.step_cloth(s, w, a)
        water_reward += info[4]
    assert _in_cup(b, s, w)[0].all() and water_reward == 0                                   # tilted by about 25 degrees: nothing spilled
    for k in range(60):
        obs, rew, done, info = o.step_cloth(s, w, a)
        water_reward += info[4]
    inside, _ = _in_cup(b, s, w)
    v = b.view(s[None])
    alive = bin(int(v['task'][0][L.DK['ALIVE']]) & 0xffffffff).count('1') + bin(int(v['task'][0][L.DK['ALIVE'] + 1]) & 0xffffffff).count('1')
    assert inside.sum() < 10 and alive < 10 and water_reward == -(64 - alive)               # upside down: each particle further than 0.1 m from the cup costs 1 (drinking.py:77-80)
    assert np.isfinite(obs).all() and np.isfinite(rew)


def test_reward_terms(dk, settled):
    """distance of the cup's top centre to the mouth, the action norm, the tilt term and the preferences (drinking.py:19-33, env.py:249-256)"""
    b, o = dk
    s, w, _ = settled
    s, w = s.copy(), w.copy()
    a = np.array([0.3, -0.2, 0.1, 0.5, -0.4, 0.2, 0.1], np.float32)
    obs, rew, done, info = o.step_cloth(s, w, a)
    v = b.view(s[None])
    tp, tq = v['free'][0, 0, :3].astype(np.float64), v['free'][0, 0, 3:7].astype(np.float64)
    cp, cq = X.compose(tp, tq, np.array([0, 0.06, 0]), X.quat_from_rpy([np.pi / 2, 0, 0]))                # drinking.py:24
    top, _ = X.compose(cp, cq, np.array([0, 0, -0.055]), np.array([0, 0, 0, 1.0]))                          # :25, 138
    x, y, z, ww = cq
    roll = np.arctan2(2 * (y * z + ww * x), ww * ww - x * x - y * y + z * z)                                # p.getEulerFromQuaternion
    ee_v = np.linalg.norm(o.world_frame_velocity(s, b.task_i('EE_LINK'))) if hasattr(o, 'world_frame_velocity') else None
    want_wo_pref = -np.linalg.norm(v['target'][0].astype(np.float64) - top) - 0.01 * np.linalg.norm(a) - 0.1 * abs(roll - np.pi / 2)
    assert abs((rew - info[5]) - want_wo_pref) < 2e-5                                                       # info[5] = the preferences score
    assert info[5] <= 0 and obs.shape == (25,) and not done
    # the observation is the feeding one with the cup (drinking.py:93-106): cup pose in the robot's base frame, cup - mouth, joint angles, head pose, force
    bp, bq = v['base'][0, :3].astype(np.float64), v['base'][0, 3:].astype(np.float64)
    ip, iq = X.invert(bp, bq)
    cup_real, _ = X.compose(ip, iq, tp, tq)
    assert np.allclose(obs[:3], cup_real, atol=1e-5)
    mouth_real, _ = X.compose(ip, iq, v['target'][0].astype(np.float64), np.array([0, 0, 0, 1.0]))
    assert np.allclose(obs[7:10], cup_real - mouth_real, atol=1e-5)


# ------------------------------------------------------------------------------------------------ the water kernel source (csrc/agx_water.h) on the wave emulator
def _water_on_emulator(b, o, s, w, action):
    """one env step: the oracle steps the scene and records the frames of every internal substep; the kernel source replays the water over that
    trace from the same start.  Returns (oracle water, kernel water, kernel report, oracle state after, info)"""
    import ctypes as C
    from emu_lib import lib, _p
    L = lib(0)
    nsub = int(b.param('FRAME_SKIP')) * b.h['SIM_SUBSTEPS']
    trace = np.zeros((nsub, b.ndof + b.nfree, 12), np.float32)
    s0, w_o = s.copy(), w.copy()
    o.L.agxo_trace_into(_p(trace))
    try:
        obs, rew, done, info = o.step_cloth(s, w_o, action)
    finally:
        o.L.agxo_trace_into(None)
    w_k = w.copy(); report = np.zeros(64, np.int32)
    words = np.ascontiguousarray(b.words)
    rc = L.agx_emu_water(_p(words), _p(s0), _p(trace), _p(w_k), _p(report), C.c_int(nsub))
    assert rc == 0
    return w_o, w_k, report, info


def test_water_kernel_source_matches_the_oracle(dk, settled):
    """at rest in the cup, while the cup is driven to tip over (particles sliding, leaving, falling), and landing on the person's lap"""
    b, o = dk
    s, w, _ = settled
    s, w = s.copy(), w.copy()
    a = np.zeros(7, np.float32); a[4], a[5], a[6] = 0.5, 1.0, 1.0
    worst = 0.0
    for k in range(60):
        w_o, w_k, report, info = _water_on_emulator(b, o, s, w, a if k else np.array([0.3, -0.2, 0.1, 0.5, -0.4, 0.2, 0.1], np.float32))
        # the particles that count: within SPILL_DIST + 5 cm of the cup (beyond 0.1 m a particle is spilled and leaves self.waters, drinking.py:77-80;
        # what it bounces off on its way down -- the arm, the wheelchair's 44 pieces -- is a chaotic sequence of single contacts)
        cup = b.view(s[None])['free'][0, 0, :3].astype(np.float64)
        near = (np.abs(w_o[0]).max(axis=1) < 500) & (np.linalg.norm(w_o[0] - cup, axis=1) < 0.15)
        if not near.any():
            break
        dp = np.abs(w_k[0, near].astype(np.float64) - w_o[0, near]).max(axis=1); dv = np.abs(w_k[1, near].astype(np.float64) - w_o[1, near]).max(axis=1)
        worst = max(worst, dp.max())
        # float32 against float64 over 20 substeps x 10 iterations of a jittering, contact-rich pile (the oracle's own particles move at up to
        # 0.1 m/s "at rest").  The typical particle agrees to a few micrometres; a particle on the threshold of one more neighbour overlap
        # (the contact set is discontinuous) ends up to half a millimetre away in single steps (measured: max 0.55 mm, median 3.6 um over
        # the 70 steps of this trajectory).  Velocities are position differences over 5 ms: 200 x the position noise.
        assert dp.max() < 2e-3 and np.median(dp) < 5e-5 and np.median(dv) < 2e-3, (k, dp.max(), np.median(dp), np.median(dv))
        w = w_o                                           # follow the oracle: every step starts both from the same water
    assert worst > 0                                      # (not the same arithmetic: a real comparison)
    # onto the lap: the report is the person-hit flag of the last internal substep
    v = b.view(s[None]); target = v['target'][0].astype(np.float64)
    s2, w2 = settled[0].copy(), settled[1].copy()
    top = np.argsort(-w2[0][:, 2])[:3]
    for j, i in enumerate(top):
        w2[0][i] = target + [0.1 + 0.02 * (j - 1), -0.2, -0.402]; w2[1][i] = [0.0, 0.0, -0.98]
    w_o, w_k, report, info = _water_on_emulator(b, o, s2, w2, np.zeros(7, np.float32))
    assert sorted(np.nonzero(report)[0]) == sorted(int(i) for i in top) and np.abs(w_k[0] - w_o[0]).max() < 2e-3


def test_water_kernel_source_compiles_for_gfx950(tmp_path):
    """hipcc cross-compiles without a GPU: registers, scratch and LDS of the kernel body as the device would run it"""
    import os, shutil, subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc on this box')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-c', '-I', os.path.join(root, 'assistive_gym_amd', 'csrc'), os.path.join(root, 'tests', 'diag', 'water_kernel_check.hip'),
                        '-o', str(tmp_path / 'wk.o'), '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    usage = {ln.split('remark:')[1].split(':')[0].strip(): ln.split(':')[-1].split('[')[0].strip() for ln in r.stderr.splitlines() if 'remark:' in ln and ':' in ln.split('remark:')[1]}
    assert int(usage['ScratchSize [bytes/lane]']) == 0 and int(usage['VGPRs']) <= 160 and int(usage['LDS Size [bytes/block]']) == 4 * (12 * 64 + 6 * 192 + 3 * 64 + 14 * 192 + 192)      # frames, boxes, positions, shape table, shape list (the candidates live in registers: 3 waves per SIMD)
