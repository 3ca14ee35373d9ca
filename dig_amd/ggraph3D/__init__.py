"""G-SphereNet's private SphereNet (dig/ggraph3D/method/G_SphereNet/model/spherenet.py) on the HIP engine — the one
module of ggraph3D that is the threedgraph hot path under another name.  The generative model around
it (sphgen.py, rdkit evaluation, ...) is out of scope."""
from .spherenet import SphereNet

__all__ = ['SphereNet']
