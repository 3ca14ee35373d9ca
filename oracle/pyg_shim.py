"""TEST INFRASTRUCTURE — pure-torch CPU stand-ins for the third-party wheels the
reference's ``dig.threedgraph`` imports but which are absent from /root/reference
and cannot be installed offline:

    torch_scatter  2.0.9   (docs/environment.yaml:15)
    torch_sparse   0.6.13  (docs/environment.yaml:16)
    torch_cluster  1.6.0   (docs/environment.yaml:17)
    torch_geometric 2.1.0  (docs/environment.yaml:19)

Each stand-in restates the published semantics of the upstream op (SURVEY.md
Appendix A.1-A.7) at exactly the call sites the reference uses:

    scatter / scatter_min     geometric_computing.py:75, spherenet.py:171,211,224,313,
                              dimenetpp.py:150,190,203,286, schnet.py:55,81,
                              comenet.py:304,311,316,325,398
    SparseTensor              geometric_computing.py:28-30,35,40-41,54
    radius_graph              spherenet.py:304, dimenetpp.py:277, schnet.py:156, comenet.py:294
    GraphConv / GraphNorm     comenet.py:130-133,150-152,160
    inits                     spherenet.py:44-48 ..., comenet.py:50-80
    DataLoader / Batch / Data run.py:53-55

``install()`` registers them in ``sys.modules`` under the upstream names so the
reference sources can be imported verbatim (oracle/ref_loader.py).  Nothing in the
product package ``dig_amd`` imports this file.
"""
import importlib
import math
import sys
import types

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- torch_scatter
def _expand_index(index, src, dim):
    if index.dim() == src.dim():
        return index
    shape = [1] * src.dim()
    shape[dim] = -1
    return index.view(shape).expand_as(src)


def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    dim = dim if dim >= 0 else src.dim() + dim
    idx = _expand_index(index, src, dim)
    if out is None:
        if dim_size is None:
            dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
        size = list(src.shape)
        size[dim] = dim_size
        out = src.new_zeros(size)
        return out.scatter_add(dim, idx, src)
    return out.scatter_add_(dim, idx, src)


scatter_add = scatter_sum


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    dim = dim if dim >= 0 else src.dim() + dim
    s = scatter_sum(src, index, dim, out, dim_size)
    n = s.size(dim)
    cnt = torch.bincount(index, minlength=n).clamp(min=1).to(src.dtype)
    shape = [1] * s.dim()
    shape[dim] = -1
    return s / cnt.view(shape)


def scatter_min(src, index, dim=-1, out=None, dim_size=None):
    """1-D restatement: returns (min value per segment, position of the FIRST minimum);
    empty segment -> value 0, arg = src.numel() (sentinel, comenet.py:305).  Gradient flows
    to the arg-min element only (torch_scatter backward of scatter_min)."""
    assert src.dim() == 1 and index.dim() == 1 and out is None
    E = src.numel()
    if dim_size is None:
        dim_size = int(index.max()) + 1 if E > 0 else 0
    sd = src.detach()
    best = torch.full((dim_size,), float('inf'), dtype=src.dtype, device=src.device)
    best = best.scatter_reduce(0, index, sd, 'amin', include_self=True)
    pos = torch.arange(E, device=src.device)
    cand = torch.where(sd == best[index], pos, torch.full_like(pos, E))
    arg = torch.full((dim_size,), E, dtype=torch.long, device=src.device)
    arg = arg.scatter_reduce(0, index, cand, 'amin', include_self=True)
    valid = arg < E
    val = src.new_zeros(dim_size)
    if bool(valid.any()):
        val = val.index_put((valid.nonzero().view(-1),), src[arg[valid]])
    return val, arg


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce='sum'):
    if reduce in ('sum', 'add'):
        return scatter_sum(src, index, dim, out, dim_size)
    if reduce == 'mean':
        return scatter_mean(src, index, dim, out, dim_size)
    if reduce == 'min':
        return scatter_min(src, index, dim, out, dim_size)[0]
    raise ValueError(reduce)


# --------------------------------------------------------------------------- torch_sparse
class _Storage:
    def __init__(self, row, col, value):
        self._row, self._col, self._value = row, col, value

    def row(self):
        return self._row

    def col(self):
        return self._col

    def value(self):
        return self._value


class SparseTensor:
    """COO sorted by (row, col) + CSR pointer; only the members
    geometric_computing.py:28-30,35,40-41,54 touches."""

    def __init__(self, row=None, col=None, value=None, sparse_sizes=None, _sorted=False):
        M, N = sparse_sizes
        if not _sorted:
            perm = torch.argsort(row * N + col, stable=True)
            row, col = row[perm], col[perm]
            value = value[perm] if value is not None else None
        self.storage = _Storage(row, col, value)
        self._sizes = (M, N)
        cnt = torch.bincount(row, minlength=M)
        self._rowptr = torch.cat([cnt.new_zeros(1), cnt.cumsum(0)])

    def sparse_sizes(self):
        return self._sizes

    def __getitem__(self, idx):
        rowptr = self._rowptr
        cnt = rowptr[idx + 1] - rowptr[idx]
        n = idx.numel()
        new_row = torch.arange(n, device=idx.device).repeat_interleave(cnt)
        tot = int(cnt.sum())
        excl = cnt.cumsum(0) - cnt
        within = torch.arange(tot, device=idx.device) - excl.repeat_interleave(cnt)
        src = rowptr[idx].repeat_interleave(cnt) + within
        val = self.storage._value
        return SparseTensor(row=new_row, col=self.storage._col[src],
                            value=val[src] if val is not None else None,
                            sparse_sizes=(n, self._sizes[1]), _sorted=True)

    def set_value(self, value, layout=None):
        return SparseTensor(row=self.storage._row, col=self.storage._col, value=value,
                            sparse_sizes=self._sizes, _sorted=True)

    def sum(self, dim):
        assert dim == 1
        if self.storage._value is None:
            return self._rowptr[1:] - self._rowptr[:-1]
        return scatter_sum(self.storage._value, self.storage._row, 0, dim_size=self._sizes[0])

    # dig/ggraph3D/method/G_SphereNet/model/geometric_computing.py:13-19 only
    def to_dense(self):
        M, N = self._sizes
        val = self.storage._value
        out = torch.zeros(M, N, dtype=val.dtype if val is not None else torch.float32, device=self.storage._row.device)
        out[self.storage._row, self.storage._col] = val if val is not None else 1
        return out

    @staticmethod
    def from_dense(mat):
        row, col = mat.nonzero(as_tuple=True)
        return SparseTensor(row=row, col=col, value=mat[row, col], sparse_sizes=tuple(mat.shape), _sorted=True)


def _sparse_matmul(*a, **k):  # pronet only; never on the hot path
    raise NotImplementedError


# --------------------------------------------------------------------------- torch_cluster
def radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32,
                 flow='source_to_target', num_workers=1):
    """torch_cluster.radius_graph, CUDA ordering rule: per target (ascending),
    in-radius sources of the same graph in ascending index, strict d^2 < r^2 accumulated in the
    input dtype; collect up to max_num_neighbors(+1 when not loop) points INCLUDING the
    target itself, then drop the self pair."""
    assert flow == 'source_to_target'
    N = x.size(0)
    if batch is None:
        batch = x.new_zeros(N, dtype=torch.long)
    cap = max_num_neighbors if loop else max_num_neighbors + 1
    r2 = x.new_tensor(r) * x.new_tensor(r)
    src_l, tgt_l = [], []
    # per-graph dense blocks (batch is sorted)
    counts = torch.bincount(batch, minlength=int(batch.max()) + 1 if N else 0)
    start = 0
    for c in counts.tolist():
        if c == 0:
            continue
        p = x[start:start + c]
        d = p[:, None, :] - p[None, :, :]
        d2 = d[..., 0] * d[..., 0]
        for k in range(1, x.size(1)):
            d2 = d2 + d[..., k] * d[..., k]
        adj = d2 < r2                                    # [target, source], includes self
        rank = adj.cumsum(1)
        adj = adj & (rank <= cap)
        if not loop:
            adj = adj & ~torch.eye(c, dtype=torch.bool, device=x.device)
        t, s = adj.nonzero(as_tuple=True)
        src_l.append(s + start)
        tgt_l.append(t + start)
        start += c
    if not src_l:
        return torch.zeros(2, 0, dtype=torch.long, device=x.device)
    return torch.stack([torch.cat(src_l), torch.cat(tgt_l)])


# --------------------------------------------------------------------------- torch_geometric.nn.inits
def knn_graph(x, k, batch=None, loop=False, flow='source_to_target', cosine=False, num_workers=1):
    """torch_cluster.knn_graph restated: for every node i the k nearest OTHER nodes j of its graph (squared L2 in the
    input precision), edge_index = [j, i], grouped by i, nearest first."""
    assert flow == 'source_to_target' and not cosine and not loop
    N = x.size(0)
    if batch is None:
        batch = torch.zeros(N, dtype=torch.long, device=x.device)
    d2 = (x.unsqueeze(1) - x.unsqueeze(0)).pow(2).sum(-1)
    bad = (batch.unsqueeze(1) != batch.unsqueeze(0)) | torch.eye(N, dtype=torch.bool, device=x.device)
    d2 = d2.masked_fill(bad, float('inf'))
    kk = min(k, N)
    val, idx = torch.topk(d2, kk, dim=1, largest=False, sorted=True)
    keep = torch.isfinite(val)
    i = torch.arange(N, device=x.device).unsqueeze(1).expand_as(idx)[keep]
    return torch.stack([idx[keep], i], 0)


def glorot_orthogonal(tensor, scale):
    if tensor is not None:
        torch.nn.init.orthogonal_(tensor.data)
        scale /= ((tensor.size(-2) + tensor.size(-1)) * tensor.var())
        tensor.data *= scale.sqrt()


def glorot(tensor):
    if tensor is not None:
        stdv = math.sqrt(6.0 / (tensor.size(-2) + tensor.size(-1)))
        tensor.data.uniform_(-stdv, stdv)


def zeros(tensor):
    if tensor is not None:
        tensor.data.fill_(0)


def ones(tensor):
    if tensor is not None:
        tensor.data.fill_(1)


def uniform(size, tensor):
    if tensor is not None:
        bound = 1.0 / math.sqrt(size)
        tensor.data.uniform_(-bound, bound)


def kaiming_uniform(tensor, fan, a):
    if tensor is not None:
        bound = math.sqrt(6 / ((1 + a ** 2) * fan))
        tensor.data.uniform_(-bound, bound)


# --------------------------------------------------------------------------- torch_geometric.nn
class MessagePassing(torch.nn.Module):
    """the slice of PyG's MessagePassing that pronet.py:111-147 uses: ``propagate(edge_index, x=(x, x), edge_weight=...)``
    with flow source_to_target, aggr='add': message(x_j = x[0][j], edge_weight) summed into the targets i."""

    def __init__(self, aggr='add', *a, **k):
        super().__init__()
        self.aggr = aggr

    def propagate(self, edge_index, size=None, **kwargs):
        x = kwargs['x']
        x_src, x_dst = x if isinstance(x, (tuple, list)) else (x, x)
        j, i = edge_index[0], edge_index[1]
        msg = self.message(x_j=x_src[j], edge_weight=kwargs.get('edge_weight'))
        return scatter_sum(msg, i, 0, dim_size=x_dst.size(0))


class GraphConv(torch.nn.Module):
    """PyG 2.x GraphConv(aggr='add'): lin_rel(sum_{j->i} message(x_j, w)) + lin_root(x_i)."""

    def __init__(self, in_channels, out_channels, aggr='add', bias=True):
        super().__init__()
        self.lin_rel = torch.nn.Linear(in_channels, out_channels, bias=bias)
        self.lin_root = torch.nn.Linear(in_channels, out_channels, bias=False)

    def reset_parameters(self):
        self.lin_rel.reset_parameters()
        self.lin_root.reset_parameters()

    def message(self, x_j, edge_weight):
        return x_j if edge_weight is None else edge_weight.view(-1, 1) * x_j

    def forward(self, x, edge_index, edge_weight=None):
        msg = self.message(x[edge_index[0]], edge_weight)
        out = scatter_sum(msg, edge_index[1], 0, dim_size=x.size(0))
        return self.lin_rel(out) + self.lin_root(x)


class GraphNorm(torch.nn.Module):
    def __init__(self, in_channels, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = torch.nn.Parameter(torch.ones(in_channels))
        self.bias = torch.nn.Parameter(torch.zeros(in_channels))
        self.mean_scale = torch.nn.Parameter(torch.ones(in_channels))

    def reset_parameters(self):
        ones(self.weight)
        zeros(self.bias)
        ones(self.mean_scale)

    def forward(self, x, batch=None):
        if batch is None:
            batch = x.new_zeros(x.size(0), dtype=torch.long)
        B = int(batch.max()) + 1
        mean = scatter_mean(x, batch, 0, dim_size=B)
        out = x - mean.index_select(0, batch) * self.mean_scale
        var = scatter_mean(out.pow(2), batch, 0, dim_size=B)
        std = (var + self.eps).sqrt().index_select(0, batch)
        return self.weight * out / std + self.bias


class GaussianSmearing(torch.nn.Module):  # comenet/features.py:13 import only
    def __init__(self, start=0.0, stop=5.0, num_gaussians=50):
        super().__init__()
        offset = torch.linspace(start, stop, num_gaussians)
        self.coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
        self.register_buffer('offset', offset)

    def forward(self, dist):
        dist = dist.view(-1, 1) - self.offset.view(1, -1)
        return torch.exp(self.coeff * torch.pow(dist, 2))


# --------------------------------------------------------------------------- torch_geometric.data
class Data:
    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def keys(self):
        return [k for k in self.__dict__ if not k.startswith('_')]

    def to(self, device):
        out = Data()
        for k in self.keys:
            v = getattr(self, k)
            setattr(out, k, v.to(device) if torch.is_tensor(v) else v)
        return out

    def __getattr__(self, name):  # missing attributes read as None (PyG behaviour)
        if name.startswith('__'):
            raise AttributeError(name)
        return None


class Batch(Data):
    @staticmethod
    def from_data_list(data_list):
        out = Batch()
        keys = data_list[0].keys
        for k in keys:
            vals = [getattr(d, k) for d in data_list]
            if torch.is_tensor(vals[0]):
                setattr(out, k, torch.cat([v if v.dim() > 0 else v.view(1) for v in vals], 0))
            else:
                setattr(out, k, vals)
        n = [int(d.z.size(0)) for d in data_list]
        out.batch = torch.arange(len(n)).repeat_interleave(torch.tensor(n))
        out.ptr = torch.cat([torch.zeros(1, dtype=torch.long), torch.tensor(n).cumsum(0)])
        out.num_graphs = len(n)
        return out


class DataLoader(torch.utils.data.DataLoader):
    def __init__(self, dataset, batch_size=1, shuffle=False, **kw):
        super().__init__(dataset, batch_size, shuffle, collate_fn=Batch.from_data_list, **kw)


class InMemoryDataset:
    def __init__(self, *a, **k):
        pass


def download_url(*a, **k):
    raise RuntimeError('no network')


class _SummaryWriter:
    def __init__(self, *a, **k):
        self.scalars = []

    def add_scalar(self, tag, value, step):
        self.scalars.append((tag, float(value), int(step)))

    def close(self):
        pass


# --------------------------------------------------------------------------- install
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    """Register the stand-ins under the upstream module names (idempotent)."""
    import numpy
    if not hasattr(numpy, 'math'):
        numpy.math = math            # features.py:70-71 uses np.math.factorial (numpy<2)
    if 'torch_scatter' in sys.modules and getattr(sys.modules['torch_scatter'], '_dig_shim', False):
        return
    _mod('torch_scatter', scatter=scatter, scatter_min=scatter_min, scatter_add=scatter_sum,
         scatter_sum=scatter_sum, scatter_mean=scatter_mean, _dig_shim=True)
    _mod('torch_sparse', SparseTensor=SparseTensor, matmul=_sparse_matmul)
    _mod('torch_cluster', radius_graph=radius_graph)
    inits = _mod('torch_geometric.nn.inits', glorot_orthogonal=glorot_orthogonal, glorot=glorot,
                 zeros=zeros, ones=ones, uniform=uniform, kaiming_uniform=kaiming_uniform)
    schnet = _mod('torch_geometric.nn.models.schnet', GaussianSmearing=GaussianSmearing)
    models = _mod('torch_geometric.nn.models', schnet=schnet)
    nn = _mod('torch_geometric.nn', radius_graph=radius_graph, knn_graph=knn_graph, GraphConv=GraphConv,
              GraphNorm=GraphNorm, MessagePassing=MessagePassing, inits=inits, models=models)
    data = _mod('torch_geometric.data', Data=Data, Batch=Batch, DataLoader=DataLoader,
                InMemoryDataset=InMemoryDataset, download_url=download_url)
    _mod('torch_geometric', nn=nn, data=data)
    try:
        importlib.import_module('torch.utils.tensorboard')
    except Exception:
        tb = _mod('torch.utils.tensorboard', SummaryWriter=_SummaryWriter)
        torch.utils.tensorboard = tb
    for name in ('h5py',):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _mod(name)
