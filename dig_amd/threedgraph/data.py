"""Minimal molecule-batch plumbing (what run.py:53-55 gets from torch_geometric's DataLoader/Batch,
SURVEY.md A.7): concatenate per-molecule ``z, pos[, force, node_feature]`` along dim 0, stack ``y``, add the
sorted ``batch`` vector and ``ptr``.  Any object with attributes z/pos/y works as a sample (PyG ``Data``
included), so reference datasets plug in unchanged.
"""
from types import SimpleNamespace

import torch

_NODE_KEYS = ('z', 'pos', 'force', 'node_feature')


class MolBatch(SimpleNamespace):
    def to(self, device, non_blocking=False):
        out = MolBatch(**vars(self))
        for k, v in vars(self).items():
            if torch.is_tensor(v):
                setattr(out, k, v.to(device, non_blocking=non_blocking))
        return out

    def pin_memory(self):
        out = MolBatch(**vars(self))
        for k, v in vars(self).items():
            if torch.is_tensor(v):
                setattr(out, k, v.pin_memory())
        return out


def collate(samples):
    out = MolBatch()
    n = torch.tensor([int(s.z.size(0)) for s in samples], dtype=torch.int64)
    for k in _NODE_KEYS:
        vals = [getattr(s, k, None) for s in samples]
        if all(torch.is_tensor(v) for v in vals):
            setattr(out, k, torch.cat(vals, 0))
        elif k == 'node_feature':
            out.node_feature = None
    ys = [getattr(s, 'y', None) for s in samples]
    if all(v is not None for v in ys):
        out.y = torch.cat([torch.as_tensor(v).reshape(-1) for v in ys], 0)
    out.batch = torch.arange(len(samples), dtype=torch.int64).repeat_interleave(n)
    out.ptr = torch.cat([torch.zeros(1, dtype=torch.int64), n.cumsum(0)])
    out.ptr_list = out.ptr.tolist()          # host copy (dig_amd/graphed.py splits batches without a device read)
    out.num_graphs = len(samples)
    return out


class _Positions(torch.utils.data.Dataset):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return i


class DataLoader(torch.utils.data.DataLoader):
    """torch_geometric.data.DataLoader(dataset, batch_size, shuffle) equivalent.  Datasets that store their
    molecules flat (dig_amd.threedgraph.dataset) are batched by ONE vectorised gather per batch
    (``dataset.collate_indices``) instead of per-sample Python objects."""

    def __init__(self, dataset, batch_size=1, shuffle=False, **kwargs):
        kwargs.pop('collate_fn', None)
        self.molecules = dataset
        flat = hasattr(dataset, 'collate_indices')
        src = _Positions(len(dataset)) if flat else dataset
        fn = dataset.collate_indices if flat else collate
        if kwargs.get('batch_sampler') is not None:            # data-parallel plans (dig_amd/dp.py) bring their batches
            super().__init__(src, collate_fn=fn, **kwargs)
        else:
            super().__init__(src, batch_size, shuffle, collate_fn=fn, **kwargs)
