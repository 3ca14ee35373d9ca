"""Per-kernel register / scratch / LDS budget of a csrc/*.hip translation unit, from the compiler's own report
(-Rpass-analysis=kernel-resource-usage, device-only compile; no GPU needed):

    python tools/kernel_resources.py dense.hip [name-filter]      ->  table: VGPRs, AGPRs, spilled VGPRs, scratch bytes/lane,
                                                                       waves/SIMD, LDS bytes
Used to keep `scratch_` out of the hot kernels (k_linear_pw<1,3,...> spilled 32 VGPRs)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def resources(src):
    from dig_amd.build import FLAGS, CSRC
    path = src if os.path.exists(src) else os.path.join(CSRC, src)
    cmd = ['/opt/rocm/bin/hipcc'] + [f for f in FLAGS if f != '-fPIC'] + ['--cuda-device-only', '-c', path, '-o', '/dev/null',
                                                                           '-Rpass-analysis=kernel-resource-usage']
    r = subprocess.run(cmd, capture_output=True, text=True)
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r'remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass-analysis', line) or re.search(r'remark:\s+(.*?) \[-Rpass-analysis', line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith('Function Name:'):
            cur = dict(name=t.split(':', 1)[1].strip())
            rows.append(cur)
        elif cur is not None and ':' in t:
            k, v = t.split(':', 1)
            cur[k.strip()] = v.strip()
    return rows


def demangle(names):
    p = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True)
    return p.stdout.splitlines()


if __name__ == '__main__':
    rows = resources(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    for r, n in zip(rows, demangle([r['name'] for r in rows])):
        short = re.sub(r'\(.*', '', n).replace('void ', '')
        if flt and flt not in short:
            continue
        print(f"{short[:70]:70s} vgpr {r.get('VGPRs', '?'):>4s} agpr {r.get('AGPRs', '?'):>4s} spill {r.get('VGPRs Spill', '?'):>3s} "
              f"scratch {r.get('ScratchSize [bytes/lane]', '?'):>4s} waves/SIMD {r.get('Occupancy [waves/SIMD]', '?'):>2s} lds {r.get('LDS Size [bytes/block]', '?'):>6s}")
