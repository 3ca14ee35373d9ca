"""Is this GPU lease usable at all?  Framework-only GPU work in a SUBPROCESS — no pytest, nothing of this package's
kernels.  About one lease in eight of the pool this was developed on faults inside torch's own first host->device copies
('Memory access fault by GPU' before any kernel of this repository has run: profiles/r04_leases/); on such a box no GPU
result says anything about the code, and the run must say so instead of dying without a word.

Shared by tests/conftest.py, __graft_entry__.smoke() and bench.py: probe, report ``FAULTY GPU LEASE``, stop.  No runtime
switch is tried and nothing is re-executed (r04 tried nine switch sets on two faulty leases; none helped)."""
import os
import subprocess
import sys

BOX_PROBE = ("import torch; m = torch.nn.Linear(64, 64).to('cuda'); x = torch.ones(8, 64).to('cuda'); "
             "print('BOX_OK', float(m(x).sum().cpu()))")

FAULTY = ('FAULTY GPU LEASE — `torch.nn.Linear(64, 64).to("cuda")` crashes in a fresh subprocess on this box, with '
          'nothing of this repository imported ({detail}).  Nothing was run; this is not a failure of the code '
          '(docs/history/DESIGN_rounds_1_to_5.md §0c).')


def box_probe(timeout=180, env=None):
    """-> (ok, detail).  ``python -I`` (isolated: no PYTHONPATH, no user site) so that the probe is the framework and
    nothing else — no sitecustomize of a harness, no module of this repository can be imported by accident."""
    try:
        r = subprocess.run([sys.executable, '-I', '-c', BOX_PROBE], capture_output=True, text=True, timeout=timeout,
                           env=dict(os.environ, **(env or {})))
    except subprocess.TimeoutExpired:
        return False, 'framework-only probe timed out'
    if r.returncode == 0 and 'BOX_OK' in r.stdout:
        return True, ''
    tail = ' | '.join((r.stdout + r.stderr).strip().splitlines()[-3:])
    return False, f'framework-only probe exited {r.returncode}: {tail[:400]}'
