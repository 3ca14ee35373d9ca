from .run import run
from .schnet import SchNet
from .dime_family import DimeNetPP, SphereNet
from .comenet import ComENet

__all__ = ['run', 'SchNet', 'DimeNetPP', 'SphereNet', 'ComENet']
