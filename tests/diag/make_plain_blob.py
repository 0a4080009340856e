"""Writes tests/golden/feeding_jaco_plain.agxblob: the PLAIN FeedingJaco model of tests/diag/approximation_budget.py -- full collision hulls (the model compiler run with
robot_hull_max_verts=None), KEEP = 0 in every pair group, a row for every contact inside the break distance, contact / row / pair budgets of 1024 / 4096 / 10^6, plain 50
sweeps (NOOP_RETEST = 0) -- for tests/test_gpu_bench_size.py::test_plain_oracle_parity_config2, which runs where the reference's assets are not (the GPU box).
Needs /root/reference (the compiler reads the reference's URDF / mesh assets).  usage: python tests/diag/make_plain_blob.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'diag'))
import approximation_budget as AB
b = AB.plain_blob('feeding_jaco')
out = os.path.join(ROOT, 'tests', 'golden', 'feeding_jaco_plain.agxblob')
b.words.tofile(out)
print('wrote', out, b.words.nbytes, 'bytes; hull vertices', b.h['NVERT'], 'colliders', b.h['NCOLL'])
