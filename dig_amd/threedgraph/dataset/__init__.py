"""``dig.threedgraph.dataset`` drop-ins: QM93D and MD17 from LOCAL ``.npz`` files.

The reference classes (dataset/PygQM93D.py:11-117, dataset/PygMD17.py:10-107) are PyG ``InMemoryDataset``s that
download the raw file and cache a collated ``(data, slices)`` ``.pt``.  There is no network here and no PyG, so
these read the same raw files (or, when only the reference's processed cache ``<root>/<folder>/processed/*_pyg.pt`` is
present, that file — un-pickled with stub classes, no PyG needed: ``read_processed``) — ``<root>/qm9/raw/qm9_eV.npz`` (keys R, Z, N + 12 targets, PygQM93D.py:81-99) and
``<root>/<name>/raw/<name>_dft.npz`` (keys E, F, R, z, PygMD17.py:80-91) — and keep the molecules as FLAT arrays
(one ``z`` / ``pos`` array + a pointer vector): indexing a sample is two slices, collating a batch is one
vectorised gather (``collate_indices``), which at GPU step times (5 ms) is what keeps the loader off the critical
path.  Same public surface as the reference: ``len``, integer / tensor / slice indexing, ``dataset.data.y =
dataset.data['U0']`` target selection, ``get_idx_split(data_size, train_size, valid_size, seed)`` (identical
``sklearn.utils.shuffle`` call => identical split indices, test/threedgraph/dataset/test_QM93D.py:31-34).
"""
import os.path as osp

import numpy as np
import torch

from ..data import MolBatch

QM9_TARGETS = ['mu', 'alpha', 'homo', 'lumo', 'gap', 'r2', 'zpve', 'U0', 'U', 'H', 'G', 'Cv']


class _Store:
    """attribute + item access to the per-dataset tensors (``dataset.data.y``, ``dataset.data['U0']``)."""

    def __getitem__(self, k):
        return getattr(self, k)

    def __setitem__(self, k, v):
        setattr(self, k, v)

    def keys(self):
        return [k for k in vars(self)]


def get_idx_split(data_size, train_size, valid_size, seed):
    """PygQM93D.py:113-117 / PygMD17.py:102-106, verbatim semantics."""
    from sklearn.utils import shuffle
    ids = shuffle(range(data_size), random_state=seed)
    return {'train': torch.tensor(ids[:train_size]), 'valid': torch.tensor(ids[train_size:train_size + valid_size]),
            'test': torch.tensor(ids[train_size + valid_size:])}


class FlatMoleculeDataset(torch.utils.data.Dataset):
    """Molecules stored flat: ``data.z [sumN]``, ``data.pos [sumN,3]`` (+ ``data.force``), graph-level ``data.y``
    and named targets ``[G]``; ``ptr [G+1]``.  ``index`` (None = all) selects a subset without copying."""

    def __init__(self, data, ptr, index=None):
        self.data, self.ptr = data, ptr
        self.index = index

    def __len__(self):
        return int(self.index.numel()) if self.index is not None else int(self.ptr.numel()) - 1

    def _graph_keys(self):
        G = self.ptr.numel() - 1
        return [k for k, v in vars(self.data).items() if torch.is_tensor(v) and v.size(0) == G and k not in ('z', 'pos', 'force')]

    def __getitem__(self, i):
        if isinstance(i, (int, np.integer)):
            g = int(self.index[i]) if self.index is not None else int(i)
            if g < 0:
                g += self.ptr.numel() - 1
            s, e = int(self.ptr[g]), int(self.ptr[g + 1])
            out = MolBatch(z=self.data.z[s:e], pos=self.data.pos[s:e])
            if hasattr(self.data, 'force'):
                out.force = self.data.force[s:e]
            for k in self._graph_keys():
                setattr(out, k, self.data[k][g:g + 1])
            return out
        idx = torch.as_tensor(np.arange(len(self))[i] if isinstance(i, slice) else i, dtype=torch.int64).reshape(-1)
        base = self.index[idx] if self.index is not None else idx
        return FlatMoleculeDataset(self.data, self.ptr, base)

    def collate_indices(self, idx):
        """One batch from sample positions ``idx`` (vectorised; what the DataLoader calls instead of per-sample
        collation): z, pos[, force], y, batch, ptr, num_graphs."""
        idx = torch.as_tensor(idx, dtype=torch.int64)
        g = self.index[idx] if self.index is not None else idx
        s, e = self.ptr[g], self.ptr[g + 1]
        n = e - s
        bvec = torch.arange(g.numel(), dtype=torch.int64).repeat_interleave(n)
        ptr = torch.cat([torch.zeros(1, dtype=torch.int64), n.cumsum(0)])
        rows = torch.arange(int(ptr[-1]), dtype=torch.int64) - ptr[:-1][bvec] + s[bvec]
        out = MolBatch(z=self.data.z[rows], pos=self.data.pos[rows], batch=bvec, ptr=ptr, num_graphs=int(g.numel()),
                       node_feature=None, ptr_list=ptr.tolist())
        if hasattr(self.data, 'force'):
            out.force = self.data.force[rows]
        for k in (self.collate_keys if self.collate_keys is not None else self._graph_keys()):
            setattr(out, k, self.data[k][g])
        return out

    # graph-level tensors a BATCH carries: the trainer reads only ``y`` (run.py:127), so the other eleven QM9 targets
    # are not gathered per batch; set to None to collate every graph-level key, or list more names
    collate_keys = ('y',)

    def get_idx_split(self, data_size, train_size, valid_size, seed):
        return get_idx_split(data_size, train_size, valid_size, seed)


# ---------------------------------------------------------------------------------------------------------------
# processed ``(data, slices)`` files of the reference (PygQM93D.py:108-111, PygMD17.py:96-99) — read WITHOUT PyG
# ---------------------------------------------------------------------------------------------------------------
class _PygStub:
    """stands in for any torch_geometric class while un-pickling: keeps the pickled state, runs no PyG code."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__['_state'] = state


class _StubPickle:
    """``pickle_module`` for torch.load: classes from torch_geometric.* resolve to _PygStub, everything else (torch
    tensors, storages, builtins, collections) resolves normally."""
    import pickle as _p
    __name__ = 'dig_amd_stub_pickle'
    Pickler = _p.Pickler
    load, loads, dump, dumps = _p.load, _p.loads, _p.dump, _p.dumps

    class Unpickler(_p.Unpickler):
        def find_class(self, module, name):
            if module.split('.')[0] == 'torch_geometric':
                return _PygStub
            return super().find_class(module, name)


def _attr_mapping(obj):
    """{name: tensor} of a pickled PyG Data object: PyG >= 2.0 keeps them in ``_store._mapping``, PyG 1.x in
    ``__dict__``."""
    d = getattr(obj, '__dict__', {})
    store = d.get('_store')
    if store is not None:
        m = getattr(store, '__dict__', {}).get('_mapping')
        if isinstance(m, dict):
            return m
    return {k: v for k, v in d.items() if torch.is_tensor(v)}


def read_processed(path):
    """-> (data: _Store, ptr) from a ``(data, slices)`` ``.pt`` written by the reference's ``process()``: node-level
    tensors (sliced by the atom pointer) stay flat, graph-level targets (one row per molecule) become [G] vectors."""
    obj = torch.load(path, map_location='cpu', weights_only=False, pickle_module=_StubPickle)
    if not (isinstance(obj, (tuple, list)) and len(obj) >= 2 and isinstance(obj[1], dict)):
        raise RuntimeError(f'{path}: not a (data, slices) file of torch_geometric.data.InMemoryDataset')
    attrs, slices = _attr_mapping(obj[0]), obj[1]
    if 'z' not in attrs or 'pos' not in attrs:
        raise RuntimeError(f'{path}: the collated Data object has no z / pos')
    ptr = slices['z'].to(torch.int64)
    G = ptr.numel() - 1
    data = _Store()
    for k, v in attrs.items():
        if not torch.is_tensor(v) or k not in slices:
            continue
        sl = slices[k].to(torch.int64)
        if torch.equal(sl, ptr):                                   # node level: z, pos, force
            data[k] = v.to(torch.int64) if k == 'z' else v
        elif sl.numel() == G + 1 and torch.equal(sl, torch.arange(G + 1)):      # graph level: one entry per molecule
            data[k] = v.reshape(G, -1).squeeze(1) if v.numel() == G else v
    return data, ptr


def _raw(root, folder, fname, url):
    path = osp.join(root, folder, 'raw', fname)
    if not osp.exists(path):
        raise FileNotFoundError(
            f'{path} not found.  dig_amd does not download datasets (no network on the build/bench nodes): fetch '
            f'{url} and place it there — it is the same raw file the reference downloads.')
    return np.load(path)


class QM93D(FlatMoleculeDataset):
    r"""QM9 with 3D positions (dataset/PygQM93D.py).  ``root/qm9/raw/qm9_eV.npz`` must exist."""
    url = 'https://github.com/klicperajo/dimenet/raw/master/data/qm9_eV.npz'

    def __init__(self, root='dataset/', transform=None, pre_transform=None, pre_filter=None):
        if transform is not None or pre_transform is not None or pre_filter is not None:
            raise NotImplementedError('transforms/filters are PyG hooks the threedgraph examples never use')
        processed = osp.join(root, 'qm9', 'processed', 'qm9_pyg.pt')       # PygQM93D.py:75-77
        if osp.exists(processed) and not osp.exists(osp.join(root, 'qm9', 'raw', 'qm9_eV.npz')):
            data, ptr = read_processed(processed)                            # a cache written by the reference itself
            super().__init__(data, ptr)
            return
        raw = _raw(root, 'qm9', 'qm9_eV.npz', self.url)
        N = torch.from_numpy(raw['N'].astype(np.int64))
        data = _Store()
        data.z = torch.from_numpy(raw['Z'].astype(np.int64))
        data.pos = torch.from_numpy(raw['R'].astype(np.float32))
        for name in QM9_TARGETS:
            data[name] = torch.from_numpy(raw[name].astype(np.float32))
        data.y = data.mu                       # PygQM93D.py:99: y defaults to the first target
        super().__init__(data, torch.cat([torch.zeros(1, dtype=torch.int64), N.cumsum(0)]))


class MD17(FlatMoleculeDataset):
    r"""MD17 trajectories with energies and forces (dataset/PygMD17.py).  ``root/<name>/raw/<name>_dft.npz``."""

    def __init__(self, root='dataset/', name='benzene_old', transform=None, pre_transform=None, pre_filter=None):
        if transform is not None or pre_transform is not None or pre_filter is not None:
            raise NotImplementedError('transforms/filters are PyG hooks the threedgraph examples never use')
        self.name = name
        processed = osp.join(root, name, 'processed', name + '_pyg.pt')     # PygMD17.py:69-71
        if osp.exists(processed) and not osp.exists(osp.join(root, name, 'raw', name + '_dft.npz')):
            data, ptr = read_processed(processed)
            super().__init__(data, ptr)
            return
        raw = _raw(root, name, name + '_dft.npz', 'http://quantum-machine.org/gdml/data/npz/' + name + '_dft.npz')
        E, F, R, z = raw['E'], raw['F'], raw['R'], raw['z']
        G, n = R.shape[0], R.shape[1]
        data = _Store()
        data.z = torch.from_numpy(np.tile(z.astype(np.int64), G))
        data.pos = torch.from_numpy(R.reshape(G * n, 3).astype(np.float32))
        data.force = torch.from_numpy(F.reshape(G * n, 3).astype(np.float32))
        data.y = torch.from_numpy(E.reshape(G).astype(np.float32))
        super().__init__(data, torch.arange(G + 1, dtype=torch.int64) * n)
