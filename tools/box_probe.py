"""Is this GPU box usable AT ALL?  (r04: one lease in ~8 of this pool faults inside torch's own `model.to('cuda')` — before any
kernel of this repository runs; the driver's r03 GPU test run died the same way.)

Runs framework-only operations (no dig_amd import), each in its own subprocess so a GPU memory-access fault (SIGABRT)
is an observation, not the end of the probe.  If the plain environment fails, the same operations are retried under
candidate workarounds.  Prints one JSON record; exit code 0 = box healthy, 3 = box faulty."""
import json
import os
import subprocess
import sys
import time

OPS = {
    'alloc_fill': "x = torch.zeros(1 << 20, device='cuda'); torch.cuda.synchronize(); print(float(x.sum()))",
    'kernel': "x = torch.ones(1 << 20, device='cuda'); y = (x * 2 + 1).sum(); torch.cuda.synchronize(); print(float(y))",
    'h2d_pageable_small': "x = torch.arange(1000, dtype=torch.float32).to('cuda'); torch.cuda.synchronize(); print(float(x.sum()))",
    'h2d_pageable_large': "x = torch.ones(1 << 22).to('cuda'); torch.cuda.synchronize(); print(float(x.sum()))",
    'h2d_pinned': "x = torch.ones(1 << 20).pin_memory().to('cuda', non_blocking=True); torch.cuda.synchronize(); print(float(x.sum()))",
    'd2h': "x = torch.ones(1 << 20, device='cuda'); print(float(x.cpu().sum()))",
    'module_to': "m = torch.nn.Sequential(*[torch.nn.Linear(64, 64) for _ in range(8)]).to('cuda'); torch.cuda.synchronize(); print(sum(float(p.sum()) for p in m.parameters()) is not None)",
}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# runtime switches under which the faulting operations are repeated (characterisation only: nothing in the suite, smoke()
# or bench.py applies them — on the two faulty leases of r04 none of them changed anything)
BOX_WORKAROUNDS = (
    ('sdma_off', {'HSA_ENABLE_SDMA': '0'}),                                  # copies by shader blits instead of the SDMA engines
    ('no_direct_dispatch', {'AMD_DIRECT_DISPATCH': '0'}),
    ('fine_grain_pcie', {'HSA_FORCE_FINE_GRAIN_PCIE': '1'}),
    ('no_caching_allocator', {'PYTORCH_NO_HIP_MEMORY_CACHING': '1'}),
    ('serialized', {'AMD_SERIALIZE_KERNEL': '3', 'AMD_SERIALIZE_COPY': '3', 'HSA_ENABLE_SDMA': '0'}),
    ('dev_kernarg', {'HIP_FORCE_DEV_KERNARG': '1'}),
    ('no_fragment_allocator', {'HSA_DISABLE_FRAGMENT_ALLOCATOR': '1'}),
)
ENVS = dict([('plain', {})] + list(BOX_WORKAROUNDS))


def run(op, env):
    e = dict(os.environ, **env)
    t0 = time.time()
    try:
        r = subprocess.run([sys.executable, '-I', '-c', 'import torch\n' + OPS[op]], env=e, capture_output=True, text=True, timeout=120)
        rc, tail = r.returncode, (r.stdout + r.stderr).strip().splitlines()[-1:] if (r.stdout + r.stderr).strip() else []
    except subprocess.TimeoutExpired:
        rc, tail = 'timeout', []
    return dict(rc=rc, s=round(time.time() - t0, 1), tail=[t[:160] for t in tail])


def main():
    uuid = ''
    try:
        out = subprocess.run(['rocminfo'], capture_output=True, text=True, timeout=60).stdout
        uuid = next((l.split()[-1] for l in out.splitlines() if 'Uuid' in l and 'GPU-' in l), '')
    except Exception:
        pass
    rec = dict(utc=time.strftime('%Y%m%dT%H%M%SZ', time.gmtime()), gpu_uuid=uuid, kernel=open('/proc/sys/kernel/osrelease').read().strip(),
               plain={})
    order = ['module_to'] + [o for o in OPS if o != 'module_to']
    first = run('module_to', {})
    rec['plain']['module_to'] = first
    healthy = first['rc'] == 0
    if not healthy:                       # characterise the fault: which operation, which workaround
        for op in order[1:]:
            rec['plain'][op] = run(op, {})
        bad = [op for op in order if rec['plain'][op]['rc'] != 0]
        rec['workarounds'] = {}
        for name, env in ENVS.items():
            if name == 'plain':
                continue
            rec['workarounds'][name] = {op: run(op, env) for op in bad[:3]}
        try:
            d = subprocess.run('dmesg 2>/dev/null | tail -40', shell=True, capture_output=True, text=True, timeout=20).stdout
            rec['dmesg_tail'] = d.splitlines()[-25:]
        except Exception:
            pass
    rec['healthy'] = healthy
    os.makedirs('gpurun_out/boxes', exist_ok=True)
    json.dump(rec, open(f'gpurun_out/boxes/box_{rec["utc"]}.json', 'w'), indent=1)
    print('[box_probe]', json.dumps(rec)[:1800], flush=True)
    sys.exit(0 if healthy else 3)


if __name__ == '__main__':
    main()
