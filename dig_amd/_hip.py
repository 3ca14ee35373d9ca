"""ctypes binding of libdig3d.so (the C ABI declared in include/dig3d.h).

The prototypes are PARSED from include/dig3d.h, so the header is the single source of truth and a
symbol that is declared but not exported fails at load time.  There is no CPU fallback: if the shared
library is missing every op raises (tests/test_boundary.py checks that).
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), 'include', 'dig3d.h')
LIB_PATH = os.path.join(HERE, 'lib', 'libdig3d.so')

_CT = {
    'int': ctypes.c_int, 'int64_t': ctypes.c_int64, 'float': ctypes.c_float, 'double': ctypes.c_double,
    'uint32_t': ctypes.c_uint32,
}


class Dig3dError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """-> {name: [(ctype, argname), ...]} for every `int dig3d_*(...)` declaration."""
    txt = open(path).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    protos = {}
    for m in re.finditer(r'\bint\s+(dig3d_\w+)\s*\(([^)]*)\)\s*;', txt):
        name, args = m.group(1), m.group(2)
        sig = []
        for a in args.split(','):
            a = ' '.join(a.split())
            if not a or a == 'void':
                continue
            if '*' in a:
                sig.append((ctypes.c_void_p, a.split('*')[-1].strip()))
            else:
                typ, nm = a.replace('const ', '').rsplit(' ', 1)
                sig.append((_CT[typ], nm))
        protos[name] = sig
    return protos


_lib = None
_fns = {}


def load():
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64 (same SONAME as /opt/rocm's): it must be the copy already mapped
    # when libdig3d.so is loaded, or the two sides would talk to different HIP runtimes.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise Dig3dError(f'{LIB_PATH} not built — run `python -m dig_amd.build` (needs hipcc). '
                         'dig_amd has no CPU fallback.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, sig in parse_header().items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise Dig3dError(f'libdig3d.so does not export {name} declared in include/dig3d.h') from e
        fn.restype = ctypes.c_int
        fn.argtypes = [t for t, _ in sig]
        _fns[name] = fn
    # the binary must come from THESE sources: a library from another checkout (or a stale object) would take the
    # header's argument lists and write through mismatched pointers — a GPU memory fault instead of an error
    from .build import source_hash
    buf = ctypes.create_string_buffer(64)
    n = _fns['dig3d_abi_hash'](ctypes.cast(buf, ctypes.c_void_p), 64)
    try:
        want = source_hash()
    except OSError as e:                 # a package shipped without dig_amd/csrc or include/: nothing to compare against
        _fns.clear()
        raise Dig3dError(f'cannot hash the sources {LIB_PATH} should have been built from ({e}): '
                         'run `python -m dig_amd.build` from a source checkout') from e
    built = buf.value.decode() if n > 0 else ''
    if built != want:
        _fns.clear()
        raise Dig3dError(f'{LIB_PATH} was built from other sources (library {built or "?"}, tree {want}): '
                         'run `python -m dig_amd.build`')
    _lib = lib
    return lib


def device_info():
    """what the library sees of the current device: dict(cus, wave, xcds, lds_bytes, device, clock_khz, hbm_bytes)"""
    if _lib is None:
        load()
    arr = (ctypes.c_int * 8)()
    rc = _fns['dig3d_device_info'](ctypes.cast(arr, ctypes.c_void_p))
    if rc != 0:
        raise Dig3dError(f'dig3d_device_info failed with code {rc}')
    v = list(arr)
    return dict(cus=v[0], wave=v[1], xcds=v[2], lds_bytes=v[3], device=v[4], clock_khz=v[5],
                hbm_bytes=(v[6] & 0xffffffff) | (v[7] << 32))


def call(name, *args):
    """Invoke a C-ABI entry point; raises Dig3dError on a non-zero return (mirrors ATen's RuntimeError)."""
    if _lib is None:
        load()
    rc = _fns[name](*args)
    if rc != 0:
        raise Dig3dError(f'{name} failed with code {rc} '
                         f'({"bad argument" if rc == -1 else "HIP launch error" if rc == -2 else "?"})')


def query(name, *args):
    """Entry points that return a size (``dig3d_*_blocks``) rather than an error code."""
    if _lib is None:
        load()
    return int(_fns[name](*args))


def query_str(name, *args, cap=128):
    """Entry points that write a name into a caller buffer (``dig3d_*_kernel``): -> str"""
    if _lib is None:
        load()
    buf = ctypes.create_string_buffer(cap)
    n = int(_fns[name](*args, ctypes.cast(buf, ctypes.c_void_p), cap))
    if n < 0:
        raise Dig3dError(f'{name} failed with code {n}')
    return buf.value.decode()


def ptr(t):
    """device pointer of a tensor (None -> NULL)"""
    return None if t is None else t.data_ptr()
