"""Compile the HIP engine for gfx950:  python -m dig_amd.build  ->  dig_amd/lib/libdig3d.so

hipcc cross-compiles without a GPU.  The .so is built IN-TREE (git-ignored, but shipped to the GPU
box by gpurun).  -ffp-contract=off: the geometry kernels reproduce the reference's float32 operation
order bit for bit (csrc/common.h); fused multiply-adds are spelled explicitly where wanted.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libdig3d.so')
SOURCES = ['graph.hip', 'geometry.hip', 'basis.hip', 'segment.hip', 'triplet.hip', 'dense.hip', 'chain.hip', 'diffgeom.hip', 'norm.hip',
           'readout.hip', 'radial.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC',
         '-fhip-fp32-correctly-rounded-divide-sqrt', '-Wno-unused-result']


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]


def _obj_stale(src, obj):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [src] + _headers())


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """compile what changed (one hipcc process per translation unit, in parallel) and relink."""
    if not force and not _stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(LIBDIR, exist_ok=True)
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        o = os.path.join(LIBDIR, s.replace('.hip', '.o'))
        objs.append(o)
        if force or _obj_stale(src, o):
            jobs.append([hipcc] + FLAGS + ['-c', src, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB])
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
