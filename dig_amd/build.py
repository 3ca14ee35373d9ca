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
SOURCES = ['abi.hip', 'graph.hip', 'geometry.hip', 'basis.hip', 'segment.hip', 'triplet.hip', 'triplet_wave.hip', 'basis_mfma.hip', 'dense.hip', 'chain.hip', 'wide.hip', 'diffgeom.hip', 'norm.hip',
           'readout.hip', 'radial.hip', 'sbf2.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC',
         '-fhip-fp32-correctly-rounded-divide-sqrt', '-Wno-unused-result']


HEADER = os.path.join(os.path.dirname(HERE), 'include', 'dig3d.h')


def _headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h'))


def _digest(paths, extra=''):
    import hashlib
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode() + b'\0')
        with open(p, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:24]


def source_hash():
    """digest of everything the binary depends on: csrc/*, include/dig3d.h and the compile flags.  Baked into the library
    (csrc/abi.hip) and re-derived by dig_amd/_hip.load(): a library built from other sources refuses to load."""
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.h')))
    return _digest(files + [HEADER], ' '.join(FLAGS))


def _stamp(path):
    try:
        with open(path + '.stamp') as f:
            return f.read().strip()
    except OSError:
        return ''


def _write_stamp(path, value):
    with open(path + '.stamp', 'w') as f:
        f.write(value)


def build(force=False, verbose=True):
    """compile what changed (one hipcc process per translation unit, in parallel) and relink.  "Changed" is decided by
    CONTENT (a digest of the source, the headers and the flags kept next to every object), never by mtimes: a tree
    copied with preserved or scrambled timestamps rebuilds exactly what differs."""
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(LIBDIR, exist_ok=True)
    abi = source_hash()
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        o = os.path.join(LIBDIR, s.replace('.hip', '.o'))
        objs.append(o)
        extra = ['-DDIG3D_ABI_HASH="%s"' % abi] if s == 'abi.hip' else []
        want = _digest([src] + _headers(), ' '.join(FLAGS + extra))
        if force or not os.path.exists(o) or _stamp(o) != want:
            jobs.append(([hipcc] + FLAGS + extra + ['-c', src, '-o', o], o, want))
    if not jobs and os.path.exists(LIB) and _stamp(LIB) == abi:
        return LIB

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)

    def compile_one(job):
        cmd, o, want = job
        run(cmd)
        _write_stamp(o, want)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB])
    _write_stamp(LIB, abi)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
