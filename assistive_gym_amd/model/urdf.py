"""Minimal URDF reader for the model compiler.

Restates what PyBullet's ``loadURDF`` extracts for the dynamics of the hot path
(reference call sites: assistive_gym/envs/agents/jaco.py:52-54, agents/furniture.py:10-40,
assistive_gym/envs/env.py:117): link inertial data, collision geometry, joint tree, joint
axes/limits, ``<contact><lateral_friction>``.  Visuals, transmissions and gazebo tags are ignored.

Link numbering follows PyBullet: joint/link index = order of a depth-first walk from the root
link, children visited in the order their joints appear in the file, so the indices used by the
reference (e.g. Jaco arm joints 1..7, end effector 8, fingers 9/11/13 -- agents/jaco.py:8-17)
can be used unchanged.
"""
import os
import xml.etree.ElementTree as ET

import numpy as np

from . import xform as X


def _vec(s, n=3, default=0.0):
    if s is None:
        return np.full(n, default, dtype=np.float64)
    return np.array([float(x) for x in s.split()], dtype=np.float64)


class Collision:
    def __init__(self, kind, pos, quat, **kw):
        self.kind = kind          # 'box' | 'sphere' | 'cylinder' | 'capsule' | 'mesh'
        self.pos = pos
        self.quat = quat
        self.__dict__.update(kw)  # size / radius / length / filename / scale


class Link:
    def __init__(self, name):
        self.name = name
        self.mass = 0.0
        self.com_pos = np.zeros(3)
        self.com_quat = np.array([0, 0, 0, 1.0])
        self.inertia = np.zeros((3, 3))   # as written in the file, in the inertial frame
        self.collisions = []
        self.lateral_friction = 0.5       # PyBullet default when no <contact> tag [BULLET-UNVERIFIED]
        self.index = None                 # PyBullet link index (-1 = base)
        self.parent_joint = None
        self.child_joints = []


class Joint:
    def __init__(self, name):
        self.name = name
        self.type = 'fixed'
        self.parent = None
        self.child = None
        self.pos = np.zeros(3)
        self.quat = np.array([0, 0, 0, 1.0])
        self.axis = np.array([1.0, 0, 0])
        self.lower = 0.0
        self.upper = -1.0      # PyBullet reports (0, -1) for "no limit"
        self.effort = 0.0
        self.damping = 0.0
        self.index = None


class Urdf:
    def __init__(self, path):
        self.path = path
        self.dir = os.path.dirname(path)
        root = ET.parse(path).getroot()
        self.links, self.joints = {}, {}
        self.link_order, self.joint_order = [], []
        for le in root.findall('link'):
            lk = Link(le.get('name'))
            ine = le.find('inertial')
            if ine is not None:
                m = ine.find('mass')
                lk.mass = float(m.get('value')) if m is not None else 0.0
                o = ine.find('origin')
                if o is not None:
                    lk.com_pos = _vec(o.get('xyz'))
                    lk.com_quat = X.quat_from_rpy(_vec(o.get('rpy')))
                it = ine.find('inertia')
                if it is not None:
                    g = lambda k: float(it.get(k, 0.0))
                    lk.inertia = np.array([[g('ixx'), g('ixy'), g('ixz')],
                                           [g('ixy'), g('iyy'), g('iyz')],
                                           [g('ixz'), g('iyz'), g('izz')]])
            ct = le.find('contact')
            if ct is not None and ct.find('lateral_friction') is not None:
                lk.lateral_friction = float(ct.find('lateral_friction').get('value'))
            for ce in le.findall('collision'):
                o = ce.find('origin')
                pos = _vec(o.get('xyz')) if o is not None else np.zeros(3)
                quat = X.quat_from_rpy(_vec(o.get('rpy'))) if o is not None else np.array([0, 0, 0, 1.0])
                ge = ce.find('geometry')
                if ge is None:
                    continue
                if ge.find('box') is not None:
                    lk.collisions.append(Collision('box', pos, quat, size=_vec(ge.find('box').get('size'))))
                elif ge.find('sphere') is not None:
                    lk.collisions.append(Collision('sphere', pos, quat, radius=float(ge.find('sphere').get('radius'))))
                elif ge.find('cylinder') is not None:
                    c = ge.find('cylinder')
                    lk.collisions.append(Collision('cylinder', pos, quat, radius=float(c.get('radius')), length=float(c.get('length'))))
                elif ge.find('capsule') is not None:
                    c = ge.find('capsule')
                    lk.collisions.append(Collision('capsule', pos, quat, radius=float(c.get('radius')), length=float(c.get('length'))))
                elif ge.find('mesh') is not None:
                    me = ge.find('mesh')
                    sc = _vec(me.get('scale'), default=1.0) if me.get('scale') else np.ones(3)
                    lk.collisions.append(Collision('mesh', pos, quat, filename=os.path.join(self.dir, me.get('filename')), scale=sc))
            self.links[lk.name] = lk
            self.link_order.append(lk.name)
        for je in root.findall('joint'):
            j = Joint(je.get('name'))
            j.type = je.get('type')
            j.parent = je.find('parent').get('link')
            j.child = je.find('child').get('link')
            o = je.find('origin')
            if o is not None:
                j.pos = _vec(o.get('xyz'))
                j.quat = X.quat_from_rpy(_vec(o.get('rpy')))
            a = je.find('axis')
            if a is not None:
                j.axis = _vec(a.get('xyz'))
            lim = je.find('limit')
            if lim is not None and j.type in ('revolute', 'prismatic'):
                j.lower = float(lim.get('lower', 0.0))
                j.upper = float(lim.get('upper', 0.0))
            if lim is not None:
                j.effort = float(lim.get('effort', 0.0))
            d = je.find('dynamics')
            if d is not None:
                j.damping = float(d.get('damping', 0.0))
            self.joints[j.name] = j
            self.joint_order.append(j.name)
            self.links[j.child].parent_joint = j
            self.links[j.parent].child_joints.append(j)
        roots = [n for n in self.link_order if self.links[n].parent_joint is None]
        assert len(roots) == 1, roots
        self.root = self.links[roots[0]]
        self.root.index = -1
        self.indexed_joints = []
        self._number(self.root)

    def _number(self, link):
        for j in link.child_joints:
            j.index = len(self.indexed_joints)
            self.indexed_joints.append(j)
            self.links[j.child].index = j.index
            self._number(self.links[j.child])

    def link_by_index(self, idx):
        if idx == -1:
            return self.root
        return self.links[self.indexed_joints[idx].child]
