import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def pytest_configure(config):
    # the CPU oracle's ops are tiny (E ~ 1e4 rows): on a 256-thread host torch's intra-op pool costs far more in
    # barriers than it gains (253 s vs 0.4 s for one SphereNet step) — cap it
    try:
        import torch
        torch.set_num_threads(min(torch.get_num_threads(), 16))
    except Exception:
        pass
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
