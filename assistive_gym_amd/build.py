"""Build libagx.so (HIP, gfx950) in-tree.  `python -m assistive_gym_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'agx_api.hip')
DEPS = [os.path.join(HERE, 'csrc', f) for f in sorted(f for f in os.listdir(os.path.join(HERE, 'csrc')) if f.endswith(('.h', '.hip')))] + \
       [os.path.join(os.path.dirname(HERE), 'include', f) for f in ('agx.h', 'agx_blob.h')]
OUT = os.path.join(HERE, 'lib', 'libagx.so')


def build(force=False, verbose=False):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wno-unused-value', '-o', OUT, SRC]
    if verbose:
        cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='-v' in sys.argv))
