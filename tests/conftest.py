import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

# ---- memory-fault hunting modes (DESIGN.md "r03 driver fault"): off by default -------------------------------------------
#   DIG3D_EFENCE=hi|lo   every tensor in its own mapping with an unmapped guard granule right above (hi) / below (lo) it
#                        and a 0x7f payload (tools/efence): any out-of-bounds access and any index read from an
#                        unwritten slot faults deterministically.  HIP-graph capture is unavailable under it.
#   DIG3D_POISON=1       torch.empty() returns NaN / INT_MAX instead of whatever the block held before.
EFENCE = os.environ.get('DIG3D_EFENCE', '')
POISON = os.environ.get('DIG3D_POISON', '') not in ('', '0')


def _activate_hunting_modes():
    import torch
    if EFENCE and torch.cuda.is_available():
        so = os.path.join(ROOT, 'tools', 'efence', 'libefence.so')
        if not os.path.exists(so):
            raise RuntimeError(f'{so} not built: tools/efence/build.sh')
        alloc = torch.cuda.memory.CUDAPluggableAllocator(so, 'efence_malloc', 'efence_free')
        torch.cuda.memory.change_current_allocator(alloc)
    if POISON:
        torch.use_deterministic_algorithms(True, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = True


def pytest_configure(config):
    # the CPU oracle's ops are tiny (E ~ 1e4 rows): on a 256-thread host torch's intra-op pool costs far more in
    # barriers than it gains (253 s vs 0.4 s for one SphereNet step) — cap it
    try:
        import torch
        torch.set_num_threads(min(torch.get_num_threads(), 16))
    except Exception:
        pass
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'hipgraph: captures HIP graphs (skipped under DIG3D_EFENCE: the fence '
                                       'allocator has no capture support)')
    _activate_hunting_modes()
    # DIG3D_ROUTES="name=value,name=value": flip kernel-route selectors of dig_amd.ops for this session (what bench.py --route
    # does for the step time, here for the PARITY numbers: bisecting which route moved an error in the parity report)
    routes = os.environ.get('DIG3D_ROUTES', '')
    if routes:
        from dig_amd import ops
        for kv in routes.split(','):
            name, val = kv.split('=')
            cur = getattr(ops, name)
            setattr(ops, name, bool(int(val)) if isinstance(cur, bool) else type(cur)(float(val)))


from dig_amd.boxprobe import FAULTY, box_probe  # noqa: E402  (framework-only subprocess probe; shared with smoke() / bench.py)


def pytest_sessionstart(session):
    import torch
    if not torch.cuda.is_available() or os.environ.get('DIG3D_SKIP_BOX_PROBE'):
        return
    if os.environ.get('PYTEST_XDIST_WORKER'):
        return                                       # the controller process probed
    ok, detail = box_probe()
    if not ok:
        pytest.exit(FAULTY.format(detail=detail), returncode=3)


def pytest_collection_modifyitems(config, items):
    import torch
    # the canary sorts first whatever the file order: it names the device and walks the eager path stage by stage
    items.sort(key=lambda it: 0 if 'test_gpu_00_canary' in it.nodeid else 1)
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
