"""TEST INFRASTRUCTURE — import the reference's ``dig.threedgraph`` sources VERBATIM
from /root/reference (read-only, build container only) on top of oracle/pyg_shim.py.

The reference tree is loaded under the alias package name ``digref`` (its own relative
imports keep working) so it never collides with this repository's ``dig`` drop-in alias.
Nothing is copied: the files are executed where they lie.  The GPU box has no
/root/reference; ``available()`` is False there and every caller must skip.
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get('DIG_REFERENCE_ROOT', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'dig', 'threedgraph', 'method'))


def load():
    """Returns the module ``digref.threedgraph`` with .method/.utils/.evaluation imported."""
    if not available():
        raise RuntimeError('reference tree not present (expected on the GPU box)')
    from . import pyg_shim
    pyg_shim.install()
    if 'digref' not in sys.modules:
        pkg = types.ModuleType('digref')
        pkg.__path__ = [os.path.join(REFERENCE_ROOT, 'dig')]
        sys.modules['digref'] = pkg
    # dataset/ needs h5py + sklearn + network; stub the package so method/ imports cleanly
    if 'digref.threedgraph.dataset' not in sys.modules:
        ds = types.ModuleType('digref.threedgraph.dataset')
        sys.modules['digref.threedgraph.dataset'] = ds
    importlib.import_module('digref.threedgraph.method')
    importlib.import_module('digref.threedgraph.utils')
    importlib.import_module('digref.threedgraph.evaluation')
    return sys.modules['digref.threedgraph']


def load_gspherenet():
    """-> the module of dig/ggraph3D/method/G_SphereNet/model/spherenet.py (G-SphereNet's private SphereNet, SURVEY.md
    §8f-4), executed verbatim where it lies; only that file and its two relative imports (features,
    geometric_computing) are loaded — the rest of ggraph3D needs rdkit."""
    if not available():
        raise RuntimeError('reference tree not present (expected on the GPU box)')
    from . import pyg_shim
    pyg_shim.install()
    name = 'digref_gsphere'
    if name not in sys.modules:
        pkg = types.ModuleType(name)
        pkg.__path__ = [os.path.join(REFERENCE_ROOT, 'dig', 'ggraph3D', 'method', 'G_SphereNet', 'model')]
        sys.modules[name] = pkg
    return importlib.import_module(name + '.spherenet')
