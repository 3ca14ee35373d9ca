"""Adam on flat buffers — the optimizer of ``run`` (method/run.py:50: ``Adam(model.parameters(), lr, weight_decay)``).

``torch.optim.Adam`` walks 135 small tensors per SphereNet step (multi-tensor launches: 255 us of GPU time, 5 % of
the step).  ``FlatAdam`` re-points every parameter at a slice of ONE flat buffer, keeps ``exp_avg`` / ``exp_avg_sq``
flat as well, and updates everything with one HIP kernel (csrc/dense.hip:k_adam_flat).  When the gradients already
are views of one flat buffer (dig_amd/graphed.py produces them that way) nothing is packed.  Same hyper-parameters,
same arithmetic and the same ``state_dict`` layout (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``) as
``torch.optim.Adam``, so ``valid_checkpoint.pt`` (run.py:87-93) round-trips with the reference's optimizer.
One documented difference: a parameter whose ``.grad`` is None is stepped with a zero gradient (its moments decay, weight
decay applies), where torch.optim.Adam skips it — every parameter of the threedgraph models receives a gradient.
"""
import torch

from ._hip import call, ptr


def flat_layout(params):
    """offsets (in floats) of every parameter inside a flat buffer, each aligned to 4 floats = 16 bytes (the HIP
    kernels read weights with 16-byte loads), and the padded total.  Shared by FlatAdam and dig_amd/graphed.py so
    that the graph's flat gradient buffer can be consumed in place."""
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off += (p.numel() + 3) // 4 * 4
    return offs, off


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        for group in self.param_groups:
            ps = [p for p in group['params'] if p.requires_grad]
            if not ps:
                continue
            if not all(p.dtype == torch.float32 for p in ps):
                raise ValueError('FlatAdam: float32 parameters only')
            offs, npad = flat_layout(ps)
            flat = torch.zeros(npad, dtype=torch.float32, device=ps[0].device)
            m, v = torch.zeros_like(flat), torch.zeros_like(flat)
            step_t = torch.tensor(0.0)
            for p, off in zip(ps, offs):
                k = p.numel()
                flat[off:off + k].copy_(p.data.reshape(-1))
                p.data = flat[off:off + k].view_as(p)                 # the parameter now lives in the flat buffer
                self.state[p] = dict(step=step_t, exp_avg=m[off:off + k].view_as(p),
                                     exp_avg_sq=v[off:off + k].view_as(p))
            group['_flat'] = dict(param=flat, m=m, v=v, offs=offs, npad=npad, ps=ps, step=0, step_t=step_t, gbuf=None)

    def _flat_grad(self, fl):
        """the gradients as one flat buffer: zero-copy when they already are consecutive views of one."""
        ps, offs = fl['ps'], fl['offs']
        g0 = ps[0].grad
        if g0 is not None:
            base = g0.data_ptr()
            ok = all(p.grad is not None and p.grad.is_contiguous() and p.grad.data_ptr() == base + 4 * off
                     for p, off in zip(ps, offs))
            if ok and base % 16 == 0 and g0.untyped_storage().nbytes() - (base - g0.untyped_storage().data_ptr()) >= 4 * fl['npad']:
                return base, None
        if fl['gbuf'] is None:
            fl['gbuf'] = torch.zeros(fl['npad'], dtype=torch.float32, device=fl['param'].device)
        # generic route (eager backward): pack with ONE multi-tensor copy instead of a small copy per parameter
        if fl.get('gviews') is None:
            fl['gviews'] = [fl['gbuf'][off:off + p.numel()].view_as(p) for p, off in zip(ps, offs)]
        dst = [v for v, p in zip(fl['gviews'], ps) if p.grad is not None]
        src = [p.grad for p in ps if p.grad is not None]
        for v, p in zip(fl['gviews'], ps):
            if p.grad is None:
                v.zero_()
        if dst:
            torch._foreach_copy_(dst, src)
        return fl['gbuf'].data_ptr(), fl['gbuf']

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            fl = group.get('_flat')
            if fl is not None and not fl['param'].is_cuda:
                raise RuntimeError('FlatAdam.step: parameters must live on the GPU (the update is one HIP kernel; '
                                   'there is no CPU path)')
        st = torch.cuda.current_stream().cuda_stream
        for group in self.param_groups:
            fl = group.get('_flat')
            if fl is None:
                continue
            fl['step'] += 1
            fl['step_t'].fill_(float(fl['step']))
            b1, b2 = group['betas']
            gptr, keep = self._flat_grad(fl)
            call('dig3d_adam_flat', ptr(fl['param']), gptr, ptr(fl['m']), ptr(fl['v']), fl['npad'], float(group['lr']),
                 float(b1), float(b2), float(group['eps']), float(group['weight_decay']),
                 float(1.0 - b1 ** fl['step']), float(1.0 - b2 ** fl['step']), st)
            del keep
        return loss

    def state_dict(self):
        """torch.optim.Adam layout.  Internally every parameter's ``step`` is ONE shared tensor (a single fill per
        step instead of 135); torch.save would preserve that sharing and torch.optim.Adam, after loading it, would
        advance the shared counter once per parameter — so the saved state gets an independent ``step`` tensor per
        parameter (tests/test_host_logic.py::test_flat_adam_state_dict_loads_into_torch_adam)."""
        sd = super().state_dict()
        for g in sd['param_groups']:
            g.pop('_flat', None)
        sd['state'] = {k: dict(v, step=v['step'].clone()) if 'step' in v else v for k, v in sd['state'].items()}
        return sd

    def load_state_dict(self, state_dict):
        """values are copied INTO the flat buffers (the views must keep pointing there)."""
        groups = state_dict['param_groups']
        for group, saved in zip(self.param_groups, groups):
            for k in ('lr', 'betas', 'eps', 'weight_decay'):
                if k in saved:
                    group[k] = saved[k]
            fl = group.get('_flat')
            for p, pid in zip(group['params'], saved['params']):
                st = state_dict['state'].get(pid)
                if st is None or fl is None:
                    continue
                self.state[p]['exp_avg'].copy_(st['exp_avg'])
                self.state[p]['exp_avg_sq'].copy_(st['exp_avg_sq'])
                fl['step'] = int(float(st['step']))
            if fl is not None:
                fl['step_t'].fill_(float(fl['step']))
