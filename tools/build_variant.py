"""A second library that differs from the tree's in compile-time switches of ONE translation unit (same sources, so the
same ABI hash: dig_amd/_hip.py loads it) — for same-box A/B of a kernel parameter:

    python tools/build_variant.py u4 segment.hip -DFC_WAVE_U=4      ->  dig_amd/lib/libdig3d_u4.so   (the unit must wrap the
    macro in #ifndef for the time of the experiment)
    DIG3D_ABL_LIB=dig_amd/lib/libdig3d_u4.so python tools/time_conv.py        (bench.py honours DIG3D_ABL_LIB too)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dig_amd.build import FLAGS, CSRC, LIBDIR, SOURCES, build  # noqa: E402

tag, unit, defs = sys.argv[1], sys.argv[2], sys.argv[3:]
build(verbose=False)
hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
alt = os.path.join(LIBDIR, f'{unit.replace(".hip", "")}_{tag}.o')
subprocess.check_call([hipcc] + FLAGS + defs + ['-c', os.path.join(CSRC, unit), '-o', alt])
objs = [alt if s == unit else os.path.join(LIBDIR, s.replace('.hip', '.o')) for s in SOURCES]
out = os.path.join(LIBDIR, f'libdig3d_{tag}.so')
subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', out])
print(out)
