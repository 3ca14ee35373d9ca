"""bench.py on an alternative build of the library (tools/build_variant.py):  python tools/bench_variant.py <lib.so> <bench args>.
The line it prints is a diagnostic of that build, never evidence for the tree's."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dig_amd import _hip  # noqa: E402

_hip.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
