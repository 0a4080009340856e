"""Oracle-dump tool of the parity protocol: run the REFERENCE (assistive_gym on the Zackory/bullet3 PyBullet fork it
pins, setup.py:21) and record, for every step of a seeded random-action episode of ANY built single-agent environment (Feeding, BedBathing,
ScratchItch, Dressing, ArmManipulation, Drinking on every robot of assistive_gym_amd.envs.ENV_IDS; FeedingJaco-v1 by default), the complete
physics state in THIS repo's state-record layout together with the reference's observation, reward, done and `total_force_on_human`.

    python tools/pybullet_dump.py --seed 1001 --steps 200 --out tests/golden/pybullet_dump_seed1001.npz
    python tools/pybullet_dump.py --env BedBathingSawyer-v1 --out tests/golden/pybullet_dump_bed_bathing_sawyer_seed1001.npz
    python tools/pybullet_dump.py --env DressingBaxter-v1 ...          (also records the garment's node positions per step)
    python tools/pybullet_dump.py --env DrinkingJaco-v1 ...            (also records the 64 water particles per step)

The result is consumed by tests/test_reference_dump.py: each recorded state is injected (`agx_set_state` / the oracle), the recorded
action is applied, and observation, reward and force are compared with what the reference produced (1e-3 relative).  Committing such a
file turns "PARITY UNPINNED" into a pinned parity statement for the physics half as well.

THIS SCRIPT CANNOT RUN IN THE BUILD CONTAINER (no pybullet, no gym, no network) and has never been executed against the real PyBullet.
What has been executed is its state capture (tests/refbridge/capture.py): on the fake `pybullet` of tests/refbridge, whose bodies are the
CPU oracle's, capture(adopt(state)) gives `state` back for all six tasks (tests/test_reference_dump.py::test_capture_inverts_adopt_on_the_bridge).
Requirements where it is run: the reference checkout importable as `assistive_gym`, its PyBullet fork (per-body gravity,
agent.py:196-197; the cloth API for Dressing), numpy; this repository (incl. tests/) on PYTHONPATH.  The conventions the capture relies on
are listed at the top of tests/refbridge/capture.py; the one most likely to need a flip on the real engine is which frame
getBasePositionAndOrientation reports for bodies whose centre of mass is not the URDF origin (bowl, wiper, scratcher).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

TASK_MODULE = {0: 'feeding_envs', 1: 'bed_bathing_envs', 2: 'scratch_itch_envs', 3: 'dressing_envs', 4: 'arm_manipulation_envs', 5: 'drinking_envs'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seed', type=int, default=1001)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden', 'pybullet_dump_seed1001.npz'))
    ap.add_argument('--env', default='FeedingJaco-v1', help='an env id that assistive_gym_amd builds (assistive_gym_amd.envs.ENV_IDS), single agent')
    ap.add_argument('--bridge', action='store_true', help="END-TO-END REHEARSAL WITHOUT PYBULLET: run on the fake `pybullet` of tests/refbridge (bodies, joints, contacts "
                    "and cloth are the CPU oracle's) with the reference's own env class and step() on top; the start state comes from this repository's reset.  The file "
                    "it writes is NOT a PyBullet dump (name it bridge_dump_*.npz): it exercises this tool's capture, the file format and the consumers")
    ap.add_argument('--action-scale', type=float, default=1.0)
    args = ap.parse_args()
    if args.bridge:
        return bridge_main(args)

    import importlib
    import pybullet as p                                    # the fork pinned by the reference's setup.py:21
    from assistive_gym_amd.blob import ModelBlob
    from assistive_gym_amd.envs import ENV_IDS
    from refbridge import capture as cap
    assert args.env in ENV_IDS and not args.env.endswith('Human-v1'), 'a single-agent env id of assistive_gym_amd.envs.ENV_IDS'
    model = ENV_IDS[args.env].model
    blob = ModelBlob.load(model)

    env = getattr(importlib.import_module('assistive_gym.envs.' + TASK_MODULE[blob.task_kind]), args.env.split('-')[0] + 'Env')()     # e.g. feeding_envs.py:17-39
    env.seed(args.seed)                                     # env.py:78-80
    obs0 = env.reset()
    initial = cap.remember(env, blob)                       # food particles / wiping targets in creation order
    rng = np.random.RandomState(args.seed)
    actions = rng.uniform(-1, 1, (args.steps, blob.act_dim)).astype(np.float32)
    has_cloth = blob.h['OFF_CLOTH'] > 0
    nn = int(blob.i[blob.h['OFF_CLOTH']]) if has_cloth else 0

    def snap():
        cl = np.zeros((2, nn, 3), dtype=np.float32) if has_cloth else None
        s = cap.capture(env, blob, p, initial, cloth_out=cl)
        if hasattr(env, 'plane'):
            blob.view(s.reshape(1, -1))['plane_friction'][0] = p.getDynamicsInfo(env.plane.body, -1, physicsClientId=env.id)[1]      # env.py:120
        return s, cl

    states, cloths, obs, rew, done, force, success = [], [], [], [], [], [], []
    for k in range(args.steps):
        s, cl = snap(); states.append(s); cloths.append(cl)
        o, r, d, info = env.step(actions[k])                # e.g. feeding.py:12-43
        obs.append(np.asarray(o, dtype=np.float64)); rew.append(float(r)); done.append(bool(d))
        force.append(float(info['total_force_on_human'])); success.append(int(info['task_success']))
    s, cl = snap(); states.append(s); cloths.append(cl)
    extra = dict(cloth=np.asarray(cloths, dtype=np.float32)) if has_cloth else {}
    np.savez_compressed(args.out, blob_version=blob.h['VERSION'], model=model, seed=args.seed, obs0=np.asarray(obs0, dtype=np.float64),
                        states=np.asarray(states, dtype=np.float32), actions=actions, obs=np.asarray(obs), reward=np.asarray(rew),
                        done=np.asarray(done), total_force_on_human=np.asarray(force), task_success=np.asarray(success),
                        gender=env.human.gender, impairment=env.human.impairment, pybullet_api=p.getAPIVersion(), **extra)
    print('wrote', args.out, '(%d steps, return %.3f)' % (args.steps, sum(rew)))
    env.disconnect()


def bridge_main(args):
    """the same loop as main() -- capture, the reference's step(), record -- on tests/refbridge (needs /root/reference: the reference's Python runs)"""
    import refbridge
    import refcases
    from refbridge import capture as cap
    from assistive_gym_amd.blob import ModelBlob
    from assistive_gym_amd.envs import ENV_IDS
    assert refbridge.available(), 'the bridge imports the reference from /root/reference'
    assert args.env in ENV_IDS and not args.env.endswith('Human-v1')
    model = ENV_IDS[args.env].model
    blob = ModelBlob.load(model)
    task = cap.TASK_OF_KIND[blob.task_kind]
    # the episode starts from one of the start states of the reference-pinned cases (tests/refcases.py: this repository's reset, settled)
    start = next(c for c in refcases.build_cases(tasks=({'feeding': 'feeding', 'bed_bathing': 'bed', 'scratch_itch': 'scratch', 'dressing': 'dressing', 'arm_manipulation': 'arm', 'drinking': 'drinking'}[task],))
                 if c['model'] == model and not c['coop'] and not c['variant'])
    refbridge.install()
    p = sys.modules['pybullet']
    env, w = refbridge.adopt(blob, start['state'].copy(), None if start['cloth'] is None else start['cloth'].copy())
    initial = {}
    if task == 'feeding':                               # (what cap.remember() reads right after a real reset(): the particles in creation order)
        class _F:
            def __init__(self, body): self.body = body
            def __eq__(self, o): return getattr(o, 'body', None) == self.body
            def __hash__(self): return hash(self.body)
        initial['foods'] = [_F(refbridge.FOOD0 + k) for k in range(blob.nfood)]
        env.bowl = _F(refbridge.BOWL)                   # (step() never touches the bowl; the capture reads its body id)
    if task == 'drinking':
        class _W:
            def __init__(self, body): self.body = body
            def __eq__(self, o): return getattr(o, 'body', None) == self.body
            def __hash__(self): return hash(self.body)
        initial['waters'] = [_W(refbridge.WATER0 + k) for k in range(w.nwater)]
    if task == 'bed_bathing':
        ids = sorted(m for m in w.markers if m >= w.first_target_marker)
        nt = sum(int(x) for x in blob.task_i_n('NT', 4)[2 * w.gender:2 * w.gender + 2])
        initial['targets'] = ids[:nt]
    rng = np.random.RandomState(args.seed)
    actions = (rng.uniform(-1, 1, (args.steps, blob.act_dim)) * args.action_scale).astype(np.float32)
    has_cloth = blob.h['OFF_CLOTH'] > 0 and start['cloth'] is not None
    nn = start['cloth'].shape[1] if has_cloth else 0

    def snap():
        cl = np.zeros((2, nn, 3), dtype=np.float32) if has_cloth else None
        s = cap.capture(env, blob, p, initial, cloth_out=cl)
        if has_cloth and task != 'drinking':
            cl[1] = w.store()[1][1]                     # node velocities: not in the fork's API (a real dump has zeros there); the bridge knows them
        return s, cl
    states, cloths, obs, rew, done, force, success = [], [], [], [], [], [], []
    for k in range(args.steps):
        s, cl = snap(); states.append(s); cloths.append(cl)
        o, r, d, info = env.step(refbridge.split_action(env, actions[k]))
        obs.append(refbridge.flat_obs(o).astype(np.float64)); rew.append(float(r)); done.append(bool(d))
        force.append(float(info['total_force_on_human'])); success.append(int(info['task_success']))
    s, cl = snap(); states.append(s); cloths.append(cl)
    extra = dict(cloth=np.asarray(cloths, dtype=np.float32)) if has_cloth else {}
    np.savez_compressed(args.out, blob_version=blob.h['VERSION'], model=model, seed=args.seed, states=np.asarray(states, dtype=np.float32), actions=actions,
                        obs=np.asarray(obs), reward=np.asarray(rew), done=np.asarray(done), total_force_on_human=np.asarray(force), task_success=np.asarray(success),
                        source='tests/refbridge: the reference\'s Python on the CPU oracle\'s physics -- NOT PyBullet', start_case=start['name'], **extra)
    print('wrote', args.out, '(%d steps from %s, return %.3f, max force %.3f) -- a REHEARSAL file, not a PyBullet dump' % (args.steps, start['name'], sum(rew), max(force)))
    w.close()


if __name__ == '__main__':
    main()
