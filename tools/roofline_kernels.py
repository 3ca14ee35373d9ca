"""Roofline workloads for the aggregation kernels the MODELS run (not only the public scatter API), shared by
bench.py (HIP-event timing) and tools/pmc_kernels.py (the same launches under ``rocprofv3 --pmc``).

Each workload returns ``dict(name, kernel, launch, bytes, detail)``: ``launch()`` enqueues exactly one launch of the
kernel on the current stream; ``bytes`` is the ALGORITHMIC traffic of that launch, spelled out in
``detail``.

  scatter_add        k_segsum_sorted<32,3>  public ``scatter(src, sorted int64 index)``       4MC + 8M + 4SC
  edge_to_node       k_seg_fused<32,true>   v = scatter(e2, i) of update_v (spherenet.py:211), CSR driven, C = 128
                                            4MC (rows) + 4(S+1) (row pointer) + 4SC (out)
  comenet_conv       k_seg_fused<64,false>  EdgeGraphConv sum_j w_e * x_j (comenet.py:130-133), C = 256, 32 in-edges
                                            per atom, 128-atom molecules: 4EC (weights) + 4E (source ids) + 4NC (x,
                                            every row needed at least once) + 4(N+1) + 4NC (out)
  comenet_featconv   k_featconv<64>         the same with w_e = Wc f_e evaluated in the kernel (12 features per edge):
                                            4EK + 4E + 4NC + 4(N+1) + 4NC — the [E,C] weight stream is gone
  triplet_fwd        k_trip_fwd_w<1,true>   x_kj[idx_kj] * (W2s Ps) * (W2t Pt) -> scatter over idx_ji
                                            (spherenet.py:165-171), C = 64: 4EC (x_kj) + 4T(8+8) (projected bases) +
                                            4T (idx_kj, int32) + 4(E+1) (triplet row pointer) + 4EC (out)
                                            = SURVEY's "fused triplet op" figure with int32 indices and a CSR pointer
                                            instead of the int64 idx_ji list
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402


def sorted_index(M, seglen, seed=7):
    g = torch.Generator(device='cpu').manual_seed(seed)
    lens = torch.randint(1, 2 * seglen, (M // seglen + M // (4 * seglen) + 64,), generator=g)
    idx = torch.arange(lens.numel()).repeat_interleave(lens)[:M]
    assert idx.numel() == M
    return idx


def wl_scatter_add(M=1 << 22, C=128, seglen=17):
    from dig_amd import ops
    idx = sorted_index(M, seglen).cuda()
    S = int(idx[-1]) + 1
    src = torch.randn(M, C, device='cuda')
    state = {}

    def launch():
        state['out'] = ops.scatter(src, idx, dim=0, dim_size=S, assume_sorted=True)

    def check():
        ref = torch.zeros(S, C, dtype=torch.float64, device='cuda').index_add_(0, idx, src.double())
        return (state['out'].double() - ref).abs().max().item()

    return dict(name='scatter_add', kernel='k_segsum_sorted<32, 3>', launch=launch, check=check,
                bytes=4 * M * C + 8 * M + 4 * S * C, rows=M, channels=C, segments=S,
                detail='4*M*C + 8*M + 4*S*C')


def wl_edge_to_node(M=1 << 22, C=128, seglen=17):
    from dig_amd import ops
    from dig_amd.graph import Seg
    idx = sorted_index(M, seglen)
    S = int(idx[-1]) + 1
    kptr = torch.zeros(S + 1, dtype=torch.int64)
    kptr[1:] = torch.bincount(idx, minlength=S).cumsum(0)
    seg = Seg(idx.to(torch.int32).cuda(), kptr.to(torch.int32).cuda(), None, S)
    src = torch.randn(M, C, device='cuda')
    idx_d = idx.cuda()
    state = {}

    def launch():
        state['out'] = ops.segment_fused_raw(None, None, src, None, seg, C)

    def check():
        ref = torch.zeros(S, C, dtype=torch.float64, device='cuda').index_add_(0, idx_d, src.double())
        return (state['out'].double() - ref).abs().max().item()

    return dict(name='edge_to_node', kernel=f'k_seg_fused<{C // 4}, true>', launch=launch, check=check,      # (no gather: the pipelined loop, csrc/segment.hip)
                bytes=4 * M * C + 4 * (S + 1) + 4 * S * C, rows=M, channels=C, segments=S,
                detail='4*M*C + 4*(S+1) + 4*S*C')


def wl_comenet_conv(molecules=1024, atoms=128, deg=32, C=256):
    """config 5 at its stress size: N = 131 072 atoms, E = 4.19e6 edges (32 in-edges per atom, sources inside the
    atom's own 128-atom molecule, ascending — the layout radius_graph produces)."""
    from dig_amd import ops
    from dig_amd.graph import Seg
    N, E = molecules * atoms, molecules * atoms * deg
    g = torch.Generator(device='cpu').manual_seed(3)
    # per atom: deg distinct sources of its molecule, ascending
    pick = torch.rand(N, atoms, generator=g).argsort(1)[:, :deg].sort(1).values          # [N, deg]
    src_id = (pick + (torch.arange(N) // atoms * atoms).unsqueeze(1)).reshape(-1).to(torch.int32).cuda()
    kptr = (torch.arange(N + 1, dtype=torch.int64) * deg).to(torch.int32).cuda()
    dst = torch.arange(N, dtype=torch.int32).repeat_interleave(deg).cuda()
    seg = Seg(dst, kptr, None, N)
    X = torch.randn(N, C, device='cuda')
    W = torch.randn(E, C, device='cuda')
    state = {}

    def launch():
        state['out'] = ops.segment_fused_raw(X, src_id, W, None, seg, C)

    def check():          # one molecule's worth of rows in float64
        n = 4 * atoms
        ref = (X.double()[src_id[:n * deg].long()] * W[:n * deg].double()).view(n, deg, C).sum(1)
        return (state['out'][:n].double() - ref).abs().max().item()

    return dict(name='comenet_conv', kernel=f'k_seg_fused<{C // 4}, false>', launch=launch, check=check,
                bytes=4 * E * C + 4 * E + 4 * N * C + 4 * (N + 1) + 4 * N * C, rows=E, channels=C, segments=N,
                detail='4*E*C + 4*E + 4*N*C + 4*(N+1) + 4*N*C')


def wl_comenet_featconv(molecules=1024, atoms=128, deg=32, C=256, K=12):
    """the same convolution with the edge weight w_e = Wc f_e (K = num_radial * num_spherical^2 = 12 features per edge)
    evaluated inside the kernel (csrc/segment.hip:k_featconv): the [E, C] weight tensor of comenet_conv is never formed,
    so the algorithmic traffic drops from 4EC to 4EK on the edge side — the kernel is bound by the dependent
    index -> row-gather chain and 2K FMAs per element, not by HBM (its fraction of the HBM roofline is low BY DESIGN;
    compare its time with comenet_conv's)."""
    from dig_amd._hip import call, ptr
    from dig_amd.graph import _stream
    N, E = molecules * atoms, molecules * atoms * deg
    g = torch.Generator(device='cpu').manual_seed(3)
    pick = torch.rand(N, atoms, generator=g).argsort(1)[:, :deg].sort(1).values
    src_id = (pick + (torch.arange(N) // atoms * atoms).unsqueeze(1)).reshape(-1).to(torch.int32).cuda()
    kptr = (torch.arange(N + 1, dtype=torch.int64) * deg).to(torch.int32).cuda()
    X = torch.randn(N, C, device='cuda')
    F = torch.randn(E, K, device='cuda')
    Wc = torch.randn(C, K, device='cuda') / K ** 0.5
    out = torch.empty(N, C, device='cuda')

    def launch():
        call('dig3d_featconv', ptr(X), ptr(src_id), ptr(F), K, ptr(Wc), ptr(kptr), None, N, C, ptr(out), None, _stream())

    def check():
        n = 4 * atoms
        w = F[:n * deg].double() @ Wc.double().t()
        ref = (X.double()[src_id[:n * deg].long()] * w).view(n, deg, C).sum(1)
        return (out[:n].double() - ref).abs().max().item()

    # arithmetic of one launch (csrc/segment.hip:k_featconv): per (edge, channel) the weight w = sum_k Wc[c,k] f[e,k] is K
    # multiply-adds, then one multiply-add of w * x_j into the running sum: 2 (K + 1) flops — this kernel trades the [E, C]
    # weight stream for that arithmetic, so the ceiling it is read against is the float32 VALU rate, not HBM
    return dict(name='comenet_featconv', kernel=f'k_featconv<{C // 4}, {K if K in (6, 12) else 0}>', launch=launch, check=check,
                bytes=4 * E * K + 4 * E + 4 * N * C + 4 * (N + 1) + 4 * N * C, rows=E, channels=C, segments=N,
                detail='4*E*K + 4*E + 4*N*C + 4*(N+1) + 4*N*C',
                flops=2 * (K + 1) * E * C, flops_detail='2*(K+1)*E*C  (K FMAs for w_e[c] = Wc[c,:] . f_e, one FMA w_e[c] * x_j[c])')


def wl_triplet_fwd(batch=512, C=64):
    """the fused triplet interaction on a real radius graph of ``batch`` QM9-like molecules (T ~ 1.8e6)."""
    from dig_amd._hip import call, ptr
    from dig_amd.graph import build_graph, _stream
    from dig_amd.synthetic import make_batch, batch_to
    b = batch_to(make_batch(batch, 9, 29, 0.08, 5.0, seed=1), 'cuda')
    g = build_graph(b.pos, b.batch, 5.0, triplets=True)
    E, T = g.E, g.T
    X = torch.randn(E, C, device='cuda')
    Ps = torch.randn(T, 8, device='cuda')
    Pt = torch.randn(T, 8, device='cuda')
    w2s = torch.randn(C, 8, device='cuda')
    w2t = torch.randn(C, 8, device='cuda')
    out = torch.empty(E, C, device='cuda')

    def launch():
        call('dig3d_triplet_fwd', ptr(X), ptr(g.kj), ptr(Ps), ptr(Pt), ptr(w2s), ptr(w2t), ptr(g.tptr), None, E, C,
             ptr(out), 0, _stream())

    def check():
        n = min(E, 2000)
        t1 = int(g.tptr[n])
        m = X.double()[g.kj[:t1].long()] * (Ps[:t1].double() @ w2s.double().t()) * (Pt[:t1].double() @ w2t.double().t())
        ref = torch.zeros(n, C, dtype=torch.float64, device='cuda').index_add_(0, g.ji[:t1].long(), m)
        return ((out[:n].double() - ref).abs().max() / ref.abs().max()).item()

    # which kernel dig3d_triplet_fwd launches is the LIBRARY's decision (route, width, direction, segment count:
    # csrc/triplet.hip) — ask it instead of naming one here (r05: this table still said k_trip_fwd<16, true> after route 0 had
    # moved to the wave-per-segment kernel, and the PMC pass found no kernel of that name)
    from dig_amd import _hip
    kernel = _hip.query_str('dig3d_triplet_fwd_kernel', E, C, 1, 0, 0)
    # per (triplet, channel): two 8-term dot products (W2s[c,:] . Ps[t,:], W2t[c,:] . Pt[t,:]) = 16 FMAs, their product, the
    # product with the gathered row and the add into the running sum (one multiply + one FMA): 2*16 + 1 + 2 = 35 flops
    return dict(name='triplet_fwd', kernel=kernel, launch=launch, check=check,
                bytes=4 * E * C + 4 * T * 16 + 4 * T + 4 * (E + 1) + 4 * E * C, rows=T, channels=C, segments=E,
                detail='4*E*C + 4*T*(8+8) + 4*T + 4*(E+1) + 4*E*C',
                flops=35 * T * C, flops_detail='35*T*C  (two 8-term dots, their product, times the gathered row, accumulate)')


WORKLOADS = dict(scatter_add=wl_scatter_add, edge_to_node=wl_edge_to_node, comenet_conv=wl_comenet_conv,
                 comenet_featconv=wl_comenet_featconv, triplet_fwd=wl_triplet_fwd)


def calibration_copy(M=1 << 22, C=128):
    """float4 streaming copy of KNOWN size through k_gather_mul with an identity index (reads 4MC + 4M, writes 4MC):
    the PMC passes scale FETCH_SIZE / WRITE_SIZE by known / measured of this launch (MI355X_MICROARCH.md §HBM)."""
    from dig_amd._hip import call, ptr
    from dig_amd.graph import _stream
    src = torch.randn(M, C, device='cuda')
    ident = torch.arange(M, dtype=torch.int32, device='cuda')
    out = torch.empty(M, C, device='cuda')

    def launch():
        call('dig3d_gather_mul', ptr(src), ptr(ident), None, None, M, C, ptr(out), None, _stream())

    return dict(name='calib', kernel='k_gather_mul', launch=launch, read_bytes=4 * M * C + 4 * M, write_bytes=4 * M * C)


def time_workload(wl, iters=30, warmup=10):
    """mean / min launch duration (ms) from HIP events recorded on the stream the kernel is launched on."""
    for _ in range(warmup):
        wl['launch']()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record()
        wl['launch']()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return sum(ms) / len(ms), ms[0]


def kernel_name_matches(kname, printed):
    """does the kernel ``kname`` ("k_trip_fwd_w<1, true>", "k_gather_mul") name the dispatch rocprofv3 printed as ``printed``
    ("void (anonymous namespace)::k_trip_fwd_w<1, true>(float const*, ...)", "k_gather_mul(HIP_vector_type<float, 4u> const*,
    ...)")?  The base name must END at the key ('k_trip_fwd' is not 'k_trip_fwd_w'), template arguments must agree when given,
    and the argument list — which has '<' and '(' of its own — is not part of the name."""
    key = kname.split('<')[0]
    tmpl = kname[len(key):].replace(' ', '')
    head = printed.replace('(anonymous namespace)::', '').split('(')[0]
    base = head.split('<')[0].split('::')[-1].split(' ')[-1]
    return base == key and (not tmpl or tmpl in head.replace(' ', ''))


def collect_pmc(names, timeout=240, keep_dir=None):
    """HBM bytes per launch of the named workloads from the PMC counters, as MI355X_MICROARCH.md §HBM prescribes:
    FETCH_SIZE and WRITE_SIZE in SEPARATE ``rocprofv3 --kernel-trace --pmc`` passes (they do not fit one pass, and
    ``--pmc`` is never combined with other trace domains), each scaled by known/measured bytes of a float4 streaming
    copy in the same process (FETCH_SIZE under-reports wide coalesced reads 2x on gfx950; WRITE_SIZE uncalibrated).
    Returns {name: dict(read_bytes, write_bytes, traffic_bytes)} + {'_calibration': ...}; {} if rocprofv3 is absent
    or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(prof):
        return {}
    script = os.path.join(ROOT, 'tools', 'pmc_kernels.py')
    kernels = {n: None for n in names}
    raw = {}
    base = keep_dir or tempfile.mkdtemp(prefix='dig3d_pmc_')
    env = dict(os.environ, TMPDIR='/tmp')
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = os.path.join(base, counter)
        shutil.rmtree(d, ignore_errors=True)
        cmd = [prof, '--kernel-trace', '--pmc', counter, '-d', d, '-o', 'p', '--output-format', 'csv', '--',
               sys.executable, script] + list(names)
        try:
            r = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True, timeout=timeout)
        except (subprocess.TimeoutExpired, OSError) as ex:
            return {'_error': f'{counter} pass: {type(ex).__name__}: {str(ex)[:300]}'}
        for line in r.stdout.splitlines():
            if line.startswith('PMCWL '):
                _, n, rest = line.split(' ', 2)
                kernels[n] = rest.rsplit(' ', 1)[0]
        files = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
        if r.returncode != 0 or not files:
            tail = ' | '.join(l for l in (r.stderr or '').splitlines()[-6:])
            return {'_error': f'{counter} pass: rc {r.returncode}, {len(files)} counter csv; stderr tail: {tail[:600]}'}
        acc = {}
        for row in csv.DictReader(open(files[0])):
            if row['Counter_Name'] == counter:
                acc.setdefault(row['Kernel_Name'], []).append(float(row['Counter_Value']))
        raw[counter] = acc
    if not keep_dir:
        shutil.rmtree(base, ignore_errors=True)

    def mean_for(counter, kname, skip=2):
        for k, v in raw[counter].items():
            if kernel_name_matches(kname, k):
                v = v[skip:] if len(v) > skip else v
                return sum(v) / len(v)
        return None

    M, C = 1 << 22, 128
    known_r, known_w = 4 * M * C + 4 * M, 4 * M * C
    cf, cw = mean_for('FETCH_SIZE', 'k_gather_mul'), mean_for('WRITE_SIZE', 'k_gather_mul')
    if not cf or not cw:
        return {'_error': 'no dispatch of the calibration copy k_gather_mul in the counter CSVs'}
    fs, ws = known_r / (cf * 1024.0), known_w / (cw * 1024.0)      # counters are in KB
    out = {'_calibration': dict(kernel='k_gather_mul identity float4 copy', known_read_bytes=known_r,
                                known_write_bytes=known_w, fetch_scale=fs, write_scale=ws,
                                raw_KB=dict(FETCH_SIZE=cf, WRITE_SIZE=cw))}
    for n in names:
        if kernels.get(n) is None:
            continue
        f, w = mean_for('FETCH_SIZE', kernels[n]), mean_for('WRITE_SIZE', kernels[n])
        if f is None or w is None:
            # say so: a silent null is how a mislabelled kernel went unnoticed for a round
            out[n] = dict(error=f'no dispatch of {kernels[n]!r} in the counter CSVs; kernels seen: '
                                + ', '.join(sorted({k.split('(')[0][:60] for k in raw['FETCH_SIZE']})[:40]))
            continue
        rb, wb = f * 1024.0 * fs, w * 1024.0 * ws
        out[n] = dict(read_bytes=rb, write_bytes=wb, traffic_bytes=rb + wb, raw_KB=dict(FETCH_SIZE=f, WRITE_SIZE=w))
    return out
