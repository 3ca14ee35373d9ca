"""Trainer with the API of dig/threedgraph/method/run.py (``run().run(...)``, ``train``, ``val``), driving the
HIP engine.  Differences from the reference, all additive:
  * if ``torch.distributed`` is initialised (one process per GPU, RCCL) the training set is sharded by graph
    and the flat gradient bucket is all-reduced between ``backward`` and ``step`` (dig_amd/dp.py); validation
    sums are all-reduced so every rank reports the same MAE;
  * SphereNet (energy) and DimeNet++ (energy, or energy_and_force with its double backward) training replays
    forward+loss+backward as ONE HIP graph per batch size (dig_amd/graphed.py; ``run.use_hip_graph = False``
    restores kernel-by-kernel launches);
  * TensorBoard logging is optional (the package is absent from this image).
Loss, optimiser, scheduler, checkpoint keys and printed lines follow run.py:47-101.
"""
import os

import torch
from torch.autograd import grad
from torch.optim import Adam
from torch.optim.lr_scheduler import StepLR

from ... import dp, ops
from ..data import DataLoader, DeviceLoader

try:                                   # run.py:8 — optional here
    from torch.utils.tensorboard import SummaryWriter
except Exception:                      # pragma: no cover
    SummaryWriter = None


def _num_atoms(dataset):
    """atoms per molecule (host tensor) for the data-parallel cost balance; None if the dataset cannot tell cheaply."""
    ptr = getattr(dataset, 'ptr', None)
    if torch.is_tensor(ptr):                                      # flat storage (dig_amd.threedgraph.dataset)
        n = ptr[1:] - ptr[:-1]
        index = getattr(dataset, 'index', None)
        return n[index] if index is not None else n
    if len(dataset) <= 200000:
        try:
            return torch.tensor([int(dataset[i].z.size(0)) for i in range(len(dataset))])
        except Exception:
            return None
    return None


def _is_mean_reduced(loss_func):
    """True when zero-padding rescales the loss by n / (n + pad): the property the graphed force loss relies on
    (dig_amd/graphed.py rescales the padded mean to the live atom count).  torch.nn.L1Loss() / MSELoss() pass;
    reduction='sum' or any non-elementwise loss does not and takes the kernel-by-kernel step."""
    try:
        g = torch.Generator().manual_seed(0)
        x, t = torch.randn(5, 3, generator=g), torch.randn(5, 3, generator=g)
        z = torch.zeros(3, 3)
        a = float(loss_func(x, t))
        b = float(loss_func(torch.cat([x, z]), torch.cat([t, z])))
        return abs(b * 8.0 / 5.0 - a) <= 1e-5 * max(1.0, abs(a))
    except Exception:
        return False


class run():
    r"""The base script for running different 3DGN methods (same call signature as the reference)."""

    use_hip_graph = True
    seed = 0              # data-parallel shuffling plan (shared by all ranks)

    def __init__(self):
        self._bucket = None
        self._stepper = None

    def run(self, device, train_dataset, valid_dataset, test_dataset, model, loss_func, evaluation, epochs=500,
            batch_size=32, vt_batch_size=32, lr=0.0005, lr_decay_factor=0.5, lr_decay_step_size=50,
            weight_decay=0, energy_and_force=False, p=100, save_dir='', log_dir=''):
        model = model.to(device)
        num_params = sum(q.numel() for q in model.parameters())
        if dp.rank() == 0:
            print(f'#Params: {num_params}')
        if device.type == 'cuda':
            from ...optim import FlatAdam          # Adam's arithmetic and state_dict layout, one kernel per step
            optimizer = FlatAdam(model.parameters(), lr=lr, weight_decay=weight_decay)
        else:
            optimizer = Adam(model.parameters(), lr=lr, weight_decay=weight_decay)
        scheduler = StepLR(optimizer, step_size=lr_decay_step_size, gamma=lr_decay_factor)
        world, rk = dp.world_size(), dp.rank()
        self._bucket = None
        if world > 1:
            # every rank starts from rank 0's weights (replicas built under different RNG state would otherwise train
            # apart silently: the averaged gradient is identical, the weights it is applied to are not)
            dp.broadcast_parameters(model, optimizer)
            self._bucket = dp.GradBucket(model)
        self._stepper = None
        name = type(model).__name__
        graphable = name in ('DimeNetPP', 'SphereNet', 'SchNet', 'ComENet')      # see graphed.py
        if (self.use_hip_graph and device.type == 'cuda' and graphable and model._fused_ok()
                and bool(getattr(model, 'energy_and_force', False)) == bool(energy_and_force)
                and _is_mean_reduced(loss_func)):
            from ...graphed import GraphedStep
            # torch.nn.L1Loss() (the loss of every reference example, run.py:127): one kernel forward, one backward
            l1 = isinstance(loss_func, torch.nn.L1Loss) and loss_func.reduction == 'mean'
            energy_loss = ((lambda out, y: ops.l1_mean(out, y.unsqueeze(1))) if l1
                           else (lambda out, y: loss_func(out, y.unsqueeze(1))))
            energy_loss.is_l1_mean = bool(l1)        # GraphedStep: L1 energies + L1 forces -> the whole loss as one kernel
            self._stepper = GraphedStep(model, energy_loss, grad_scale=1.0 / world, force_loss=loss_func, p=p)
        if world > 1:
            # training: one deterministic plan on every rank — global batches of batch_size * world graphs, dealt to
            # the ranks balanced by estimated cost (n * deg^2), reshuffled every epoch, ragged last batch weighted by
            # B_local / B_global; validation / test: ragged shards that cover the set (exact global MAE)
            n_at = _num_atoms(train_dataset)
            costs = dp.molecule_cost(n_at) if n_at is not None else None
            sampler = dp.BalancedBatchSampler(len(train_dataset), batch_size, rk, world, costs, shuffle=True, seed=self.seed)
            train_loader = DataLoader(train_dataset, batch_sampler=sampler)
            valid_loader = DataLoader(valid_dataset, batch_sampler=dp.ListBatchSampler(
                dp.shard_indices(len(valid_dataset), rk, world, drop_tail=False), vt_batch_size))
            test_loader = DataLoader(test_dataset, batch_sampler=dp.ListBatchSampler(
                dp.shard_indices(len(test_dataset), rk, world, drop_tail=False), vt_batch_size))
            if self._stepper is not None:
                self._stepper.set_scale(1.0 / world)
                self._precapture_union(train_dataset, sampler, device)
        else:
            train_loader = DataLoader(train_dataset, batch_size, shuffle=True)
            valid_loader = DataLoader(valid_dataset, vt_batch_size, shuffle=False)
            test_loader = DataLoader(test_dataset, vt_batch_size, shuffle=False)
        best_valid = float('inf')
        best_test = float('inf')
        if save_dir != '' and not os.path.exists(save_dir):
            os.makedirs(save_dir, exist_ok=True)
        writer = None
        if log_dir != '':
            os.makedirs(log_dir, exist_ok=True)
            if SummaryWriter is not None and rk == 0:
                writer = SummaryWriter(log_dir=log_dir)
        log = print if rk == 0 else (lambda *a, **k: None)
        for epoch in range(1, epochs + 1):
            log('\n=====Epoch {}'.format(epoch), flush=True)
            log('\nTraining...', flush=True)
            train_mae = self.train(model, optimizer, train_loader, energy_and_force, p, loss_func, device)
            log('\n\nEvaluating...', flush=True)
            valid_mae = self.val(model, valid_loader, energy_and_force, p, evaluation, device)
            log('\n\nTesting...', flush=True)
            test_mae = self.val(model, test_loader, energy_and_force, p, evaluation, device)
            log()
            log({'Train': train_mae, 'Validation': valid_mae, 'Test': test_mae})
            if writer is not None:
                writer.add_scalar('train_mae', train_mae, epoch)
                writer.add_scalar('valid_mae', valid_mae, epoch)
                writer.add_scalar('test_mae', test_mae, epoch)
            if valid_mae < best_valid:
                best_valid, best_test = valid_mae, test_mae
                if save_dir != '' and rk == 0:
                    log('Saving checkpoint...')
                    checkpoint = {'epoch': epoch, 'model_state_dict': model.state_dict(),
                                  'optimizer_state_dict': optimizer.state_dict(),
                                  'scheduler_state_dict': scheduler.state_dict(), 'best_valid_mae': best_valid,
                                  'num_params': num_params}
                    torch.save(checkpoint, os.path.join(save_dir, 'valid_checkpoint.pt'))
            scheduler.step()
        log(f'Best validation MAE so far: {best_valid}')
        log(f'Test MAE when got best validation result: {best_test}')
        if writer is not None:
            writer.close()
        self.best_valid, self.best_test = best_valid, best_test

    def _precapture_union(self, train_dataset, sampler, device):
        """Data parallel: every rank captures the size classes of the WHOLE job's first epoch before step 0.  The
        sampler's plan is deterministic, so each rank walks its own first-epoch batches once (radius-graph builds
        only), the ranks exchange {class key: count} and capture the union, most frequent first, all at the same
        time — instead of each rank stalling the other seven at the all-reduce for ~1 s whenever IT meets a new class
        (graphed.py ``scan_classes`` / ``precapture``).  Classes that only appear in later epochs' shuffles are still
        captured on first sight."""
        import torch.distributed as dist
        plan = [b for b in sampler.plan()[0] if len(b)]
        # a SAMPLE of the plan: evenly spaced batches up to ``precapture_scan`` (default 256) — the scan costs a collate, a
        # host->device copy and a radius-graph build with one host sync per batch, a whole extra data pass on a large set
        # would be the price of the first epoch's stalls it avoids.  Classes the sample misses are captured on first sight.
        cap = int(getattr(self, 'precapture_scan', 256))
        if len(plan) > cap > 0:
            plan = [plan[(i * len(plan)) // cap] for i in range(cap)]
        seen, err = {}, None
        try:
            loader = DeviceLoader(DataLoader(train_dataset, batch_sampler=plan), device)
            seen = self._stepper.scan_classes(loader)
        except Exception as ex:                       # e.g. check_z_bounds' IndexError on one rank's shard
            err = f'{type(ex).__name__}: {ex}'
        mine = {k: v[0] for k, v in seen.items()}
        every = [None] * dp.world_size()
        dist.all_gather_object(every, (mine, err))    # the error travels WITH the keys: all ranks abort together
        errs = [(r, e) for r, (_, e) in enumerate(every) if e is not None]
        if errs:
            raise RuntimeError('size-class scan before step 0 failed on rank(s) '
                               + '; '.join(f'{r}: {e}' for r, e in errs))
        union = {}
        for d, _ in every:
            for k, c in d.items():
                union[k] = union.get(k, 0) + c
        made = self._stepper.precapture(seen, union)
        self.precapture_report = dict(local_classes=len(mine), local_counts=sorted(mine.values()), union_classes=len(union),
                                      captured=made)
        return self.precapture_report

    def _loss(self, model, batch_data, energy_and_force, p, loss_func):
        out = model(batch_data)
        if energy_and_force:
            from ... import diffops
            with diffops.force_gradient_scope():        # run.py:126 — the create_graph backward that needs positions only
                force = -grad(outputs=out, inputs=batch_data.pos, grad_outputs=torch.ones_like(out),
                              create_graph=True, retain_graph=True)[0]
            e_loss = loss_func(out, batch_data.y.unsqueeze(1))
            f_loss = loss_func(force, batch_data.force)
            return e_loss + p * f_loss, out, force
        return loss_func(out, batch_data.y.unsqueeze(1)), out, None

    def train(self, model, optimizer, train_loader, energy_and_force, p, loss_func, device):
        model.train()
        loss_accum = torch.zeros((), device=device)
        steps = 0
        # pinned staging buffers, one async copy per batch, collate on a worker thread (dig_amd/threedgraph/data.py)
        loader = iter(DeviceLoader(train_loader, device))
        # data parallel: this rank's share B_local / B_global of every step's global batch (ragged last batch)
        # — taken from the sampler's deterministic plan BEFORE iterating: the loader thread runs the sampler's __iter__
        # later (and rebinds its ``weights`` list), so reading that attribute here would see the previous epoch's list
        weights = None
        sampler = getattr(train_loader, 'batch_sampler', None)
        if self._bucket is not None and hasattr(sampler, 'plan'):
            weights = list(sampler.plan()[1])
        nxt = next(loader, None)
        while nxt is not None:
            batch_data = nxt
            nxt = next(loader, None)                  # one batch of look-ahead: its radius graph is queued
                                                      # behind this step's replay (GraphedStep.prefetch)
            w = weights[steps] if weights else None
            if self._stepper is not None:
                # the graphed step overwrites its static gradient buffer: no zero_grad
                if w is not None:
                    self._stepper.set_scale(w)
                # ONE HIP-graph replay: forward + loss + backward; the gradient all-reduce starts behind it, beside the
                # rest of the next batch's graph build
                start = self._bucket.allreduce_flat_start if self._bucket is not None else None
                loss = self._stepper(batch_data, prefetch=nxt, after_replay=start)
                if self._bucket is not None:
                    self._bucket.allreduce_flat_finish()
            else:
                if self._bucket is not None:
                    self._bucket.zero()
                else:
                    optimizer.zero_grad()
                loss, _, _ = self._loss(model, batch_data, energy_and_force, p, loss_func)
                if loss.is_cuda:
                    # = loss.backward(), with the weight-gradient reductions of all layers in one launch
                    ops.backward(loss, [q for q in model.parameters() if q.requires_grad])
                else:
                    loss.backward()
                if self._bucket is not None:
                    self._bucket.allreduce(scale=w)
            optimizer.step()
            # no per-step host sync (the reference calls .item() every step); DP: weighted like the gradient, so the
            # all-reduced sum below is the loss of the global batch
            loss_accum += loss.detach() * (w * dp.world_size() if w is not None else 1.0)
            steps += 1
        total = loss_accum.item() / max(steps, 1)
        return dp.allreduce_scalar_sum(total, device) / dp.world_size()

    def val(self, model, data_loader, energy_and_force, p, evaluation, device):
        model.eval()
        preds, targets, preds_force, targets_force = [], [], [], []
        for batch_data in DeviceLoader(data_loader, device):
            if energy_and_force:
                out = model(batch_data)
                force = -grad(outputs=out, inputs=batch_data.pos, grad_outputs=torch.ones_like(out),
                              create_graph=False, retain_graph=False)[0]
                preds_force.append(force.detach())
                targets_force.append(batch_data.force.clone())   # a view of a recycled DeviceLoader slot: keep a copy
            else:
                with torch.no_grad():
                    out = model(batch_data)
            preds.append(out.detach())
            # batch_data.* are views into DeviceLoader's recycled device slots: what outlives the iteration is cloned
            # (otherwise the targets of batch k are overwritten by the bytes of batch k + depth)
            targets.append(batch_data.y.unsqueeze(1).clone())
        if not preds:                                  # a rank whose ragged shard is empty (set smaller than world)
            preds, targets = [torch.zeros(0, 1, device=device)], [torch.zeros(0, 1, device=device)]
            preds_force, targets_force = [torch.zeros(0, 3, device=device)], [torch.zeros(0, 3, device=device)]
        preds, targets = torch.cat(preds, 0), torch.cat(targets, 0)

        def mae(pred, true):
            if dp.world_size() == 1:
                return evaluation.eval({'y_true': true, 'y_pred': pred})['mae']
            s = dp.allreduce_scalar_sum((pred - true).abs().sum().item(), device)
            n = dp.allreduce_scalar_sum(float(true.numel()), device)
            return s / n

        energy_mae = mae(preds, targets)
        if energy_and_force:
            force_mae = mae(torch.cat(preds_force, 0), torch.cat(targets_force, 0))
            if dp.rank() == 0:
                print({'Energy MAE': energy_mae, 'Force MAE': force_mae})
            return energy_mae + p * force_mae
        return energy_mae
