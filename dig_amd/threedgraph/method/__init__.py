from .run import run
from .schnet import SchNet
from .dime_family import DimeNetPP, SphereNet
from .comenet import ComENet
from .pronet import ProNet

__all__ = ['run', 'SchNet', 'DimeNetPP', 'SphereNet', 'ComENet', 'ProNet']
