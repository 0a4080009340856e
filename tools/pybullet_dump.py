#!/usr/bin/env python3
"""Oracle-dump tool of the parity protocol: run the REFERENCE (assistive_gym on the
Zackory/bullet3 PyBullet fork it pins, setup.py:21) and record, for every step of a seeded random-action
episode of a Feeding<Robot>-v1 environment (FeedingJaco-v1 by default; --env FeedingPanda-v1 / FeedingSawyer-v1 / FeedingBaxter-v1 /
FeedingPR2-v1: the same state-record layout, driven by the blob's metadata), the complete physics state in THIS repo's state-record
layout together with the reference's observation, reward, done and `total_force_on_human`.

    python tools/pybullet_dump.py --seed 1001 --steps 200 --out tests/golden/pybullet_dump_seed1001.npz
    python tools/pybullet_dump.py --env FeedingSawyer-v1 --out tests/golden/pybullet_dump_feeding_sawyer_seed1001.npz

The other tasks (bed bathing, scratch itch, dressing, arm manipulation) need a capture_state of their own (task words, tool link
frames, the garment) and are not covered yet.

The result is consumed by tests/test_reference_dump.py: each recorded state is injected
(`agx_set_state` / the oracle), the recorded action is applied, and observation, reward and force are
compared with what the reference produced (1e-3 relative).  Committing such a file turns "PARITY
UNPINNED" into a pinned parity statement.

THIS SCRIPT CANNOT RUN IN THE BUILD CONTAINER (no pybullet, no gym, no network) and has therefore
never been executed; it is written against the reference sources (file:line cited below) and the blob
metadata.  Requirements where it is run: the reference checkout importable as `assistive_gym`, its
PyBullet fork (per-body gravity, agent.py:196-197), numpy; this repository on PYTHONPATH for the blob.

Conventions that matter (and are asserted where they can be):
  * robot DoF order = the reference's joint indices `meta['dof_links']` (Jaco arm 1..7, fingers 9/11/13),
    head DoFs = human joints `meta['human_dynamic_joints']` (agents/human.py:9);
  * base / free-body poses are what p.getBasePositionAndOrientation reports (the convention of
    Agent.get_base_pos_orient / set_base_pos_orient, agents/agent.py:142-150, which the reset code of
    both the reference and assistive_gym_amd/host/reset.py uses);
  * the 17 static human collision bodies are the links `meta['human_bodies']` (-1 = base) with their
    URDF link frames (getLinkState(...)[4:6], computeForwardKinematics=True);
  * particles keep their creation index (feeding.py:154-159) even after the reference drops them from
    its lists (feeding.py:80-83).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def capture_state(env, blob, meta, foods, p):
    """the reference's current physics state -> one state record ([state_words] float32)"""
    from assistive_gym_amd.model import compiler as L
    s = blob.new_state(1)
    v = blob.view(s)
    cid = env.id
    robot, human = env.robot.body, env.human.body
    nrobot = blob.nrobot
    # --- articulated DoFs: robot arm + fingers, then the head chain
    js = p.getJointStates(robot, meta['dof_links'], physicsClientId=cid)
    v['q'][0, :nrobot] = [j[0] for j in js]
    v['qd'][0, :nrobot] = [j[1] for j in js]
    hs = p.getJointStates(human, meta['human_dynamic_joints'], physicsClientId=cid)
    v['q'][0, nrobot:] = [j[0] for j in hs]
    v['qd'][0, nrobot:] = [j[1] for j in hs]
    # motor targets: the arm targets are recomputed by take_step from q and the action (env.py:201-215);
    # the fingers keep the target set at reset (robot.py set_gripper_open_position); the head keeps the
    # targets of setup_joints (human.py:123)
    v['qt'][0, :] = v['q'][0, :]
    grip = list(env.robot.right_gripper_indices)             # feeding.py:143: opened to gripper_pos[task] at reset
    for k, j in enumerate(meta['dof_links']):
        if j in grip:
            v['qt'][0, k] = env.robot.gripper_pos[env.task][grip.index(j)]
    # --- free bodies: tool, bowl, particles
    def base_state(body):
        pos, orn = p.getBasePositionAndOrientation(body, physicsClientId=cid)
        lin, ang = p.getBaseVelocity(body, physicsClientId=cid)
        return list(pos) + list(orn) + list(lin) + list(ang)
    tb, fb0 = blob.h['TOOL_BODY'], blob.h['FOOD0']
    v['free'][0, tb] = base_state(env.tool.body)
    bowl_index = [b for b in range(blob.nfree) if b != tb and not (fb0 <= b < fb0 + blob.nfood)]
    assert len(bowl_index) == 1
    v['free'][0, bowl_index[0]] = base_state(env.bowl.body)
    for k, f in enumerate(foods):
        v['free'][0, fb0 + k] = base_state(f.body)
    # --- static frames
    pos, orn = p.getBasePositionAndOrientation(robot, physicsClientId=cid)
    v['base'][0] = list(pos) + list(orn)
    for k, link in enumerate(meta['human_bodies']):
        if link < 0:
            pos, orn = p.getBasePositionAndOrientation(human, physicsClientId=cid)
        else:
            ls = p.getLinkState(human, link, computeForwardKinematics=True, physicsClientId=cid)
            pos, orn = ls[4], ls[5]
        v['human'][0, k] = list(pos) + list(orn)
    # --- per-environment words
    v['plane_friction'][0] = p.getDynamicsInfo(env.plane.body, -1, physicsClientId=cid)[1]      # env.py:120
    v['gender'][0] = 1 if env.human.gender == 'female' else 0
    v['target'][0] = env.target_pos                                                               # feeding.py:184-196
    alive = sum(1 << k for k, f in enumerate(foods) if f in env.foods)
    active = sum(1 << k for k, f in enumerate(foods) if f in env.foods_active)
    v['food_alive'][0], v['food_active'][0] = alive, active
    v['iteration'][0], v['task_success'][0], v['total_food'][0] = env.iteration, env.task_success, env.total_food_count
    v['rng'][0] = [12345, 6789]                 # device RNG of the teleport positions; not compared
    tremor = env.human.impairment == 'tremor'
    v['frozen'][0] = 0 if tremor else (((1 << blob.nhdof) - 1) << nrobot)                         # human.py:108-112
    v['limit_scale'][0] = env.human.limit_scale                                                    # human.py:85 (scales the head joint limits)
    if tremor:                                  # env.py:212-215: target + tremors * (+1 / -1 by iteration parity)
        ctrl = list(env.human.controllable_joint_indices)
        for k, j in enumerate(meta['human_dynamic_joints']):
            v['tremor'][0, k] = env.human.tremors[ctrl.index(j)]
            v['tremor_target'][0, k] = env.human.target_joint_angles[ctrl.index(j)]
    return s[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seed', type=int, default=1001)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden', 'pybullet_dump_seed1001.npz'))
    ap.add_argument('--env', default='FeedingJaco-v1', help='a Feeding<Robot>-v1 id that assistive_gym_amd builds')
    args = ap.parse_args()

    import importlib
    import json
    import pybullet as p                                    # the fork pinned by the reference's setup.py:21
    from assistive_gym_amd.blob import ModelBlob, DATA_DIR
    from assistive_gym_amd.envs import ENV_IDS
    assert args.env.startswith('Feeding') and not args.env.endswith('Human-v1') and args.env in ENV_IDS, 'a single-agent Feeding<Robot>-v1 id'
    model = ENV_IDS[args.env].model
    blob = ModelBlob.load(model)
    meta = json.load(open(os.path.join(DATA_DIR, model + '.meta.json')))

    env = getattr(importlib.import_module('assistive_gym.envs.feeding_envs'), args.env.split('-')[0] + 'Env')()     # feeding_envs.py:17-39
    env.seed(args.seed)                                     # env.py:78-80
    obs0 = env.reset()
    foods = list(env.foods)                                 # creation order, feeding.py:154-159
    assert len(foods) == blob.nfood and len(meta['dof_links']) == blob.nrobot
    rng = np.random.RandomState(args.seed)
    actions = rng.uniform(-1, 1, (args.steps, blob.act_dim)).astype(np.float32)

    states, obs, rew, done, force, success = [], [], [], [], [], []
    for k in range(args.steps):
        states.append(capture_state(env, blob, meta, foods, p))
        o, r, d, info = env.step(actions[k])                # feeding.py:12-43
        obs.append(np.asarray(o, dtype=np.float64)); rew.append(float(r)); done.append(bool(d))
        force.append(float(info['total_force_on_human'])); success.append(int(info['task_success']))
    states.append(capture_state(env, blob, meta, foods, p))
    np.savez_compressed(args.out, blob_version=blob.h['VERSION'], model=model, seed=args.seed, obs0=np.asarray(obs0, dtype=np.float64),
                        states=np.asarray(states, dtype=np.float32), actions=actions, obs=np.asarray(obs), reward=np.asarray(rew),
                        done=np.asarray(done), total_force_on_human=np.asarray(force), task_success=np.asarray(success),
                        gender=env.human.gender, impairment=env.human.impairment,
                        pybullet_api=p.getAPIVersion())
    print('wrote', args.out, '(%d steps, return %.3f)' % (args.steps, sum(rew)))
    env.disconnect()


if __name__ == '__main__':
    main()
