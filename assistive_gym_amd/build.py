"""Build libagx.so (HIP, gfx950) in-tree.  `python -m assistive_gym_amd.build`.

The kernels are compiled once per variant (limits + task layer, csrc/agx_kernels.hip) and linked with the handle /
C-ABI code (csrc/agx_api.hip)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
DEPS = [os.path.join(CSRC, f) for f in sorted(f for f in os.listdir(CSRC) if f.endswith(('.h', '.hip')))] + \
       [os.path.join(os.path.dirname(HERE), 'include', f) for f in ('agx.h', 'agx_blob.h')]
OUT = os.path.join(HERE, 'lib', 'libagx.so')
VARIANTS = ['FEEDING', 'FEEDING_L', 'FEEDING_M', 'BED_BATHING', 'BED_BATHING_L', 'BED_BATHING_M', 'SCRATCH_ITCH', 'SCRATCH_ITCH_M', 'BED_SETTLE', 'DRESSING', 'DRESSING_L', 'DRESSING_M', 'ARM_MANIPULATION', 'ARM_MANIPULATION_L', 'DRINKING', 'DRINKING_L', 'DRINKING_M']


def build(force=False, verbose=False, extra=(), out=None, only=None):
    """out / extra: an A/B build of the same library with extra compiler flags (same-box comparisons: AGX_LIB=<out> python bench.py);
    only: kernel variants the extra flags apply to (e.g. ['FEEDING']) -- the other objects are taken from the main build's lib/obj"""
    OUT = out or globals()['OUT']
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and not out and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    base = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value'] + list(extra)
    if verbose:
        base.append('-Rpass-analysis=kernel-resource-usage')
    objdir = os.path.join(HERE, 'lib', 'obj' if not out else 'obj_' + os.path.splitext(os.path.basename(out))[0])
    os.makedirs(objdir, exist_ok=True)
    jobs = [(os.path.join(objdir, 'agx_api.o'), base + ['-c', os.path.join(CSRC, 'agx_api.hip')])]
    reuse = []
    for v in VARIANTS:
        if only and v not in only:
            reuse.append(os.path.join(HERE, 'lib', 'obj', 'agx_kernels_%s.o' % v.lower()))
            continue
        jobs.append((os.path.join(objdir, 'agx_kernels_%s.o' % v.lower()), base + ['-DAGX_VARIANT_' + v, '-c', os.path.join(CSRC, 'agx_kernels.hip')]))
    if only:
        jobs[0] = (os.path.join(HERE, 'lib', 'obj', 'agx_api.o'), None)
    with ThreadPoolExecutor(len(jobs)) as ex:
        list(ex.map(lambda j: j[1] and subprocess.check_call(j[1] + ['-o', j[0]]), jobs))
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + [j[0] for j in jobs] + reuse)
    return OUT


if __name__ == '__main__':
    a = sys.argv
    out = a[a.index('--out') + 1] if '--out' in a else None
    extra = a[a.index('--extra') + 1].split() if '--extra' in a else ()
    only = a[a.index('--only') + 1].split(',') if '--only' in a else None
    print(build(force='--force' in a, verbose='-v' in a, extra=extra, out=out, only=only))
