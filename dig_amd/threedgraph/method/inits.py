"""Parameter initialisers with the semantics of torch_geometric.nn.inits, used at
spherenet.py:44-48,126-148,200-207 and comenet.py:50-80.  Init only — never on the step path."""
import math

import torch


def glorot_orthogonal_(w, scale=2.0):
    torch.nn.init.orthogonal_(w.data)
    with torch.no_grad():
        w.data.mul_(math.sqrt(scale / ((w.size(-2) + w.size(-1)) * w.data.var().item())))
    return w


def glorot_(w):
    a = math.sqrt(6.0 / (w.size(-2) + w.size(-1)))
    w.data.uniform_(-a, a)
    return w


def kaiming_uniform_(w, fan, a):
    bound = math.sqrt(6.0 / ((1 + a * a) * fan))
    w.data.uniform_(-bound, bound)
    return w
