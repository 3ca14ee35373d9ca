// dig3d segment / gather kernels — the aggregation backbone of every interaction block.
// Replaces torch_scatter.scatter(..., reduce='sum') and ATen row gathers at:
//   spherenet.py:165-171,211,224   dimenetpp.py:148-150,190,203   schnet.py:34,55,81
//   comenet.py:130-133 (EdgeGraphConv message+aggregate), :398
// Design (MI355X): every forward reduction index is sorted (edges grouped by target, triplets by
// j->i edge, nodes by graph), so reductions are CONTIGUOUS SEGMENT SUMS: a "worker" of C/4 lanes owns a
// run of rows, each lane carries one float4 column slice in registers, rows stream through 16-byte
// coalesced loads, every source byte is read once and every output row written once — no atomics, no
// memset, deterministic summation order (ascending row).  Backward of a gather (scatter_add by an
// UNSORTED index) uses the transposed CSR built once per batch (graph.hip), so it is the same kernel.
#include "common.h"

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

__device__ __forceinline__ void f4_acc(float4& a, const float4 v) {
  a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
}
__device__ __forceinline__ float4 f4_mul(const float4 a, const float4 b) {
  return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}

// ================================================================================================
// (a) scatter_add with a SORTED int64 index, API form  out = scatter(src, index, dim=0, dim_size=S).
//     Each worker (LPR lanes, C = 4*LPR) gets L consecutive rows and OWNS every segment that STARTS
//     inside its run: it skips the leading rows that continue a segment begun earlier (binary search
//     on the sorted index, no data read) and runs past its end to finish its last segment.  Empty
//     segments (gaps in the index, leading and trailing) are zero-filled by the owner of the next
//     segment start, so the launch needs no prior memset.
// ================================================================================================
//     MODE bit 0: the index of a U-row batch is fetched by ONE coalesced load (lane u of the worker reads
//     idx[r+u]) and broadcast with in-register shuffles, instead of U loads that every lane repeats;
//     MODE bit 1: the source stream is read with non-temporal loads (each byte is used once).
__device__ __forceinline__ float4 ld_stream(const float4* p, bool nt) {
  if (!nt) return *p;
  float4 v;
  v.x = __builtin_nontemporal_load(&p->x);
  v.y = __builtin_nontemporal_load(&p->y);
  v.z = __builtin_nontemporal_load(&p->z);
  v.w = __builtin_nontemporal_load(&p->w);
  return v;
}

template <int LPR, int MODE>
__global__ void __launch_bounds__(256) k_segsum_sorted(const float4* __restrict__ src,
                                                        const int64_t* __restrict__ idx, int64_t M,
                                                        int64_t S, int L, float4* __restrict__ out, int mean) {
  constexpr bool SHF = (MODE & 1) != 0, NT = (MODE & 2) != 0;
  constexpr int U = SHF ? (LPR < 16 ? LPR : 16) : 8;
  const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPR;
  const int c = threadIdx.x % LPR;
  const int64_t r0 = w * (int64_t)L;
  if (r0 >= M) return;
  const int64_t r1 = (r0 + L < M) ? r0 + L : M;
  int64_t r = r0;
  int64_t prev = -1;
  if (r0 > 0) {
    prev = idx[r0 - 1];
    // first row in [r0, r1) whose id differs from prev (upper bound of prev)
    int64_t lo = r0, hi = r1;
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      if (idx[mid] == prev) lo = mid + 1; else hi = mid;
    }
    r = lo;
    if (r >= r1) return;  // this run lies entirely inside a segment owned by an earlier worker
  }
  int64_t cur = idx[r];
  for (int64_t s = prev + 1; s < cur; ++s) out[s * LPR + c] = f4_zero();
  float4 acc = f4_zero();
  int nrows = 0;          // rows accumulated into acc (reduce = 'mean' divides by it at the store)
  auto fin = [&](float4 a) {
    if (mean && nrows > 1) {
      const float q = 1.0f / (float)nrows;
      a.x *= q; a.y *= q; a.z *= q; a.w *= q;
    }
    return a;
  };
  // main run: rows [r, r1) are all owned
  for (; r < r1; r += U) {
    int64_t id[U];
    float4 v[U];
    const int nv = (r1 - r < U) ? (int)(r1 - r) : U;   // valid rows of this batch
    int64_t mine = 0;
    if (SHF && c < nv) mine = idx[r + c];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = u < nv;
      if (SHF) {
        const int lo = __shfl((int)(mine & 0xffffffffll), u, LPR);
        const int hi = __shfl((int)(mine >> 32), u, LPR);
        id[u] = ((int64_t)hi << 32) | (uint32_t)lo;
      } else {
        id[u] = ok ? idx[r + u] : 0;
      }
      v[u] = ok ? ld_stream(&src[(r + u) * LPR + c], NT) : f4_zero();
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (u < nv && id[u] != cur) {
        out[cur * LPR + c] = fin(acc);
        for (int64_t s = cur + 1; s < id[u]; ++s) out[s * LPR + c] = f4_zero();
        cur = id[u];
        acc = f4_zero();
        nrows = 0;
      }
      f4_acc(acc, v[u]);
      if (u < nv) ++nrows;
    }
  }
  // tail: rows after r1 that continue the last segment
  r = r1;
  bool open = true;
  while (open && r < M) {
    int64_t id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) id[u] = (r + u < M) ? idx[r + u] : -2;
    int n = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (open && id[u] == cur) ++n; else open = false;
    }
    // the n rows are loaded as ONE batch of predicated loads (a `for (u < n)` loop issued them one dependent round trip
    // at a time: ~8 serial HBM latencies per worker at 17-row segments), no row beyond the segment is touched
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (u < n) ? ld_stream(&src[(r + u) * LPR + c], NT) : f4_zero();
#pragma unroll
    for (int u = 0; u < U; ++u) f4_acc(acc, v[u]);
    r += n;
    nrows += n;
  }
  out[cur * LPR + c] = fin(acc);
  if (r >= M) {  // owner of the globally last segment also clears the trailing empty segments
    for (int64_t s = cur + 1; s < S; ++s) out[s * LPR + c] = f4_zero();
  }
}

// generic (any C): one thread per (segment-run, channel) — used for C = 1 readouts and odd widths.
__global__ void k_segsum_sorted_generic(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                        int64_t M, int64_t S, int C, float* __restrict__ out, int mean) {
  // one thread per output element (s, c): binary-search the row range of s.
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= S * C) return;
  int64_t s = q / C;
  int c = (int)(q - s * C);
  int64_t lo = 0, hi = M;
  while (lo < hi) {  // lower bound of s
    int64_t mid = (lo + hi) >> 1;
    if (idx[mid] < s) lo = mid + 1; else hi = mid;
  }
  float acc = 0.f;
  int64_t r = lo;
  for (; r < M && idx[r] == s; ++r) acc += src[r * C + c];
  out[q] = (mean && r - lo > 1) ? acc / (float)(r - lo) : acc;
}

// ================================================================================================
// (b) CSR segment sum with optional row map:  out[s,:] = sum_{p in [kptr[s],kptr[s+1])} src[map(p),:]
//     map == nullptr -> identity (forward aggregation); map = perm -> backward of a row gather.
// (d) fused gather * mul (* mul) + segment sum:
//        out[s,:] = sum_p X[ix[t],:] * A[t,:] (* B[t,:]),   t = map ? map[p] : p
//     forward of the triplet interaction (spherenet.py:165-171), of SchNet's cfconv (schnet.py:34,55)
//     and of ComENet's EdgeGraphConv (comenet.py:130-133); with map = transposed CSR it is also their
//     backward w.r.t. X.  X/ix may be null (no gather factor), B may be null.
// ================================================================================================
// PIPE (launches that do not fill the chip: a few hundred segments, the reference's batch size): every load unconditional
// (positions past the end repeat the segment's last edge and are multiplied by 0) and the index chain of batch i + 1
// (position -> (map ->) row id) requested before batch i's rows — in the plain loop a batch is two or three DEPENDENT trips and
// its predicated loads run one after the other: a wave of SchNet's cfconv walked its ~18 edges in ~15 trips, 21 us per launch at
// 608 segments; same sums in the same order.  At scale (the roofline launches) the other waves of the SIMD hide the chain and
// the plain loop's fewer instructions win when rows are gathered (ComENet's convolution at 4.2e6 edges: 800 us plain, 845
// pipelined), while a pure segment sum (no X: edge -> node at 4.2e6 rows) gains from the unconditional loads (427 -> 405 us): the
// host picks PIPE for grids that do not fill the chip and for launches without a gather.
template <int LPR, bool PIPE>
__global__ void __launch_bounds__(256) k_seg_fused(const float4* __restrict__ X, const int* __restrict__ ix,
                                                    const float4* __restrict__ A, const float4* __restrict__ B,
                                                    const int* __restrict__ kptr, const int* __restrict__ map,
                                                    int S, float4* __restrict__ out, int mean, int swz) {
  const int64_t w = ((int64_t)dig3d_xcd_block(swz) * blockDim.x + threadIdx.x) / LPR;
  const int c = threadIdx.x % LPR;
  if (w >= S) return;
  const int b = kptr[w], e = kptr[w + 1];
  float4 acc = f4_zero();
  constexpr int U = 4;                    // (PIPE with 8: 15-18 us against 14.8 at 608 segments)
  if (PIPE) {
    if (b < e) {
      int tn[U], rn[U];
      auto req_idx = [&](int p) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int pp = p + u < e ? p + u : e - 1;
          tn[u] = map ? map[pp] : pp;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) rn[u] = ix ? ix[tn[u]] : tn[u];
      };
      req_idx(b);
      for (int p = b; p < e; p += U) {
        float4 va[U], vx[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (A) va[u] = A[(int64_t)tn[u] * LPR + c];
          if (X) vx[u] = X[(int64_t)rn[u] * LPR + c];
          if (B) vb[u] = B[(int64_t)tn[u] * LPR + c];
        }
        req_idx(p + U < e ? p + U : p);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          float4 x = A ? va[u] : make_float4(1.f, 1.f, 1.f, 1.f);
          if (X) x = f4_mul(x, vx[u]);
          if (B) x = f4_mul(x, vb[u]);
          const float m = p + u < e ? 1.f : 0.f;
          acc.x += x.x * m; acc.y += x.y * m; acc.z += x.z * m; acc.w += x.w * m;
        }
      }
    }
  } else
  for (int p = b; p < e; p += U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (p + u < e) {
        const int t = map ? map[p + u] : p + u;
        float4 x = A ? A[(int64_t)t * LPR + c] : make_float4(1.f, 1.f, 1.f, 1.f);
        if (X) x = f4_mul(x, X[(int64_t)(ix ? ix[t] : t) * LPR + c]);
        if (B) x = f4_mul(x, B[(int64_t)t * LPR + c]);
        v[u] = x;
      } else {
        v[u] = f4_zero();
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) f4_acc(acc, v[u]);
  }
  if (mean && e - b > 1) {
    const float q = 1.0f / (float)(e - b);
    acc.x *= q; acc.y *= q; acc.z *= q; acc.w *= q;
  }
  out[(int64_t)w * LPR + c] = acc;
}

// generic-C version of the same contraction (one thread per output element).
__global__ void k_seg_fused_generic(const float* __restrict__ X, const int* __restrict__ ix,
                                    const float* __restrict__ A, const float* __restrict__ B,
                                    const int* __restrict__ kptr, const int* __restrict__ map, int S, int C,
                                    float* __restrict__ out, int mean) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (int64_t)S * C) return;
  int s = (int)(q / C);
  int c = (int)(q - (int64_t)s * C);
  float acc = 0.f;
  for (int p = kptr[s], e = kptr[s + 1]; p < e; ++p) {
    const int t = map ? map[p] : p;
    float x = A ? A[(int64_t)t * C + c] : 1.f;
    if (X) x = x * X[(int64_t)(ix ? ix[t] : t) * C + c];
    if (B) x = x * B[(int64_t)t * C + c];
    acc += x;
  }
  const int n = kptr[s + 1] - kptr[s];
  out[q] = (mean && n > 1) ? acc / (float)n : acc;
}

// ================================================================================================
// (c) row gather (* optional per-row factors):  out[m,:] = X[ix[m],:] (* A[m,:]) (* B[m,:])
//     forward of x[i], x[j], x_kj[idx_kj]; backward of a segment sum; and the per-triplet factor
//     gradients of (d):  gA[t] = G[ji[t]] * X[kj[t]] * B[t]  ->  k_gather_mul2.
// ================================================================================================
__global__ void k_gather_mul(const float4* __restrict__ X, const int* __restrict__ ix,
                             const float4* __restrict__ A, const float4* __restrict__ B, int64_t M, int C4,
                             float4* __restrict__ out, const int* __restrict__ cnt) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= M * C4) return;
  int64_t m = q / C4;
  int c = (int)(q - m * C4);
  if (cnt && m >= *cnt) {  // padded row of a static-shape (HIP-graph) batch: exact zero
    out[q] = f4_zero();
    return;
  }
  float4 v = X[(int64_t)ix[m] * C4 + c];
  if (A) v = f4_mul(v, A[q]);
  if (B) v = f4_mul(v, B[q]);
  out[q] = v;
}
__global__ void k_gather_mul_generic(const float* __restrict__ X, const int* __restrict__ ix,
                                     const float* __restrict__ A, const float* __restrict__ B, int64_t M, int C,
                                     float* __restrict__ out, const int* __restrict__ cnt) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= M * C) return;
  int64_t m = q / C;
  int c = (int)(q - m * C);
  if (cnt && m >= *cnt) {
    out[q] = 0.f;
    return;
  }
  float v = X[(int64_t)ix[m] * C + c];
  if (A) v = v * A[q];
  if (B) v = v * B[q];
  out[q] = v;
}

// P[t] = G[ig[t]] * X[ix[t]];  outA[t] = P * B[t] (if outA),  outB[t] = P * A[t] (if outB)
__global__ void k_gather_mul2(const float4* __restrict__ G, const int* __restrict__ ig,
                              const float4* __restrict__ X, const int* __restrict__ ix,
                              const float4* __restrict__ A, const float4* __restrict__ B, int64_t M, int C4,
                              float4* __restrict__ outA, float4* __restrict__ outB, const int* __restrict__ cnt) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= M * C4) return;
  int64_t m = q / C4;
  int c = (int)(q - m * C4);
  if (cnt && m >= *cnt) {  // padded row of a static-shape (HIP-graph) batch: exact zero
    if (outA) outA[q] = f4_zero();
    if (outB) outB[q] = f4_zero();
    return;
  }
  float4 p = f4_mul(G[(int64_t)ig[m] * C4 + c], X[(int64_t)ix[m] * C4 + c]);
  if (outA) outA[q] = B ? f4_mul(p, B[q]) : p;
  if (outB) outB[q] = f4_mul(p, A[q]);
}

// ================================================================================================
// C ABI
// ================================================================================================
// No mutable state in the library: the rows-per-worker override and the kernel variant are ARGUMENTS of
// dig3d_segment_sum_sorted_tuned (sweeps and tests); the plain entry point passes (0 = heuristic, mode 3).
// MI355X, M = 2^22, C = 128, L = 64: mode 0/1/2/3 -> 4.94/4.87/4.95/5.08 TB/s on the same box.
template <int LPR>
static void launch_sorted(const float* src, const int64_t* idx, int64_t M, int64_t S, float* out, int L, int mode,
                          int mean, hipStream_t st) {
  if (L <= 0) {
    // aim for >= 8 waves per CU worth of workers, runs between 16 and 48 rows (sweep on MI355X, M = 2^22, C = 128, with
    // the batched tail loads of round 3: L = 16/32/48/64/96/128/256 -> 4.83/5.87/5.88/5.69/5.67/5.39/5.47 TB/s;
    // round 1, serial tail: 8/16/32/64/128/256 -> 2.95/4.05/5.15/5.36/5.24/5.20)
    int64_t target_workers = (int64_t)dig3d_num_cus() * 32 * (64 / LPR);
    int64_t l = (M + target_workers - 1) / target_workers;
    L = (int)(l < 16 ? 16 : (l > 48 ? 48 : l));
  }
  int64_t workers = (M + L - 1) / L;
  int64_t threads = workers * LPR;
#define SEG_LAUNCH(MODE)                                                                                    \
  hipLaunchKernelGGL((k_segsum_sorted<LPR, MODE>), dim3(dig3d_blocks(threads, 256)), dim3(256), 0, st,      \
                     (const float4*)src, idx, M, S, L, (float4*)out, mean)
  switch (mode) {
    case 1: SEG_LAUNCH(1); break;
    case 2: SEG_LAUNCH(2); break;
    case 3: SEG_LAUNCH(3); break;
    default: SEG_LAUNCH(0); break;
  }
#undef SEG_LAUNCH
}

// ================================================================================================
// (e) feature-weighted graph convolution of ComENet (comenet.py:130-133,160-175): the edge weight is itself a linear
//     map of a few edge features, w_e = Wc f_e (Wc = lin2.weight lin1.weight of the bias-free TwoLayerLinear, [C, K],
//     K = num_radial * num_spherical^{1,2} <= 16), so it is evaluated on the fly instead of being written and re-read
//     as an [E, C] tensor (537 MB per convolution at E = 5.2e5, C = 256):
//        out[s,:] = sum_{t in seg(s)} X[ix[t],:] * (Wc f_t)
//     With the transposed CSR and ix = the other end of the edge the same kernel is the gradient w.r.t. X.
//     k_featconv_wgrad: gWc[c,k] = sum_t f_t[k] * G[ig[t],c] * X[ix[t],c]   (per-block partials, then one reduction).
// ================================================================================================
#define FC_KMAX 16
#define FC_WGRAD_U 4        // edges of one target per batch in the weight gradient's wave form
#define FC_WAVE_U 4         // row gathers in flight per wave (8: 136 VGPRs, three waves per SIMD, 6.39 vs 6.34 ms per config-5 step)
// KT = the feature count at compile time (ComENet: 12 = num_radial * num_spherical^2 and 6 = num_radial * num_spherical)
// or 0 = run-time K <= 16: with a compile-time K the per-feature loop has no branches and the feature row arrives in a
// few wide scalar loads.
// ---- C = 256 with a compile-time K: a wave per segment, the segment's index chain walked ONCE ----------------------
// The loop below (k_featconv's general body) asks for a batch of four edges at a time, and every batch is a chain of
// dependent trips: edge position -> (map ->) source row -> 1-KB row, then K scalar loads of the feature row whose
// s_waitcnt also waits for everything requested behind them (scalar loads return out of order).  Counters at 5.2e5 edges
// (profiles/r05_stall_counters_comenet_128.json): a wave lives 16.8 us for 1.9 us of VALU work, VALU busy 34 %.
// Here lane l of the wave reads the position, source row and feature row of edge l of the segment (chunks of 64 edges):
// two dependent VECTOR trips for the whole chunk.  The feature rows go to the wave's slice of LDS and come back as
// broadcast reads (3 ds_read_b128 per edge at K = 12); the source rows reach the gathers through v_readlane, so a batch
// of U row gathers is ONE trip and the next batch is in flight while this one is consumed.  Same products, same order of
// additions as the general body: bit-identical results.
// What bounds it now is the VALU itself: the time does not move when every gather hits L1 (tools/diag_featconv_bound.py: 62 us
// with the real rows, 61 with all sources = row 0 at 5.2e5 edges) and grows 3.5 us per feature where the packed-FMA rate says
// 1.7.  Measured and not kept: the feature row read one edge ahead (no difference), the weights on the matrix cores
// (v_mfma_f32_16x16x4_f32, three forms: 137, 92 and 113 us against 62 — the operand layouts need four narrow gathers per edge
// where this form issues one 16-byte gather; v_mfma_f32_4x4x1_16b_f32, whose layout does fit 16-byte gathers: 88 us, and 59 when
// every gather hits L1 — a quad of lanes is four edges, so a quarter wave touches four rows; docs/history/r06_featconv_mfma.hip.txt).
template <int K, int U>
__device__ __forceinline__ void featconv_wave(const float4* __restrict__ X, const int* __restrict__ ix,
                                              const float* __restrict__ F, const float* __restrict__ Wc,
                                              const int* __restrict__ kptr, const int* __restrict__ map, int S,
                                              float4* __restrict__ out, int swz, const float4* __restrict__ add) {
  static_assert(K % 2 == 0, "feature rows are moved as 8- or 16-byte pieces");
  constexpr int VW = (K % 4 == 0) ? 4 : 2;            // floats per piece
  constexpr int NV = K / VW;
  __shared__ __attribute__((aligned(16))) float sF[4][64 * K];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(dig3d_xcd_block(swz) * 4 + wv);
  float wr[4][K];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < K; ++k) wr[j][k] = Wc[(4 * lane + j) * K + k];
  if (w >= S) return;
  const int b = kptr[w], e = kptr[w + 1];
  float* sf = sF[wv];
  float4 acc = f4_zero();
  for (int cb = b; cb < e; cb += 64) {
    const int n = e - cb < 64 ? e - cb : 64;          // wave-uniform
    const int pl = cb + (lane < n ? lane : n - 1);
    const int tl = map ? map[pl] : pl;
    const int rl = ix[tl];
    {
      const float* f = F + (int64_t)tl * K;
      if (VW == 4) {
        float4 v[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = reinterpret_cast<const float4*>(f)[q];
#pragma unroll
        for (int q = 0; q < NV; ++q) reinterpret_cast<float4*>(sf + lane * K)[q] = v[q];
      } else {
        float2 v[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = reinterpret_cast<const float2*>(f)[q];
#pragma unroll
        for (int q = 0; q < NV; ++q) reinterpret_cast<float2*>(sf + lane * K)[q] = v[q];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // U row gathers in flight, each register refilled as soon as its edge is consumed
    float4 xn[U];
    auto request = [&](int u, int jpos) {
      const int jj = jpos < n ? jpos : n - 1;
      const int row = __builtin_amdgcn_readlane(rl, jj);
      xn[u] = X[(int64_t)row * 64 + lane];
    };
#pragma unroll
    for (int u = 0; u < U; ++u) request(u, u);
    for (int j = 0; j < n; j += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float live = j + u < n ? 1.0f : 0.0f;
        const int jj = j + u < n ? j + u : n - 1;
        const float* __restrict__ fr = sf + jj * K;
        float fk[K];
        if (VW == 4) {
#pragma unroll
          for (int q = 0; q < NV; ++q) {
            const float4 t = reinterpret_cast<const float4*>(fr)[q];
            fk[4 * q] = t.x; fk[4 * q + 1] = t.y; fk[4 * q + 2] = t.z; fk[4 * q + 3] = t.w;
          }
        } else {
#pragma unroll
          for (int q = 0; q < NV; ++q) {
            const float2 t = reinterpret_cast<const float2*>(fr)[q];
            fk[2 * q] = t.x; fk[2 * q + 1] = t.y;
          }
        }
        float4 we = f4_zero();
#pragma unroll
        for (int k = 0; k < K; ++k) {
          we.x = fmaf(fk[k], wr[0][k], we.x); we.y = fmaf(fk[k], wr[1][k], we.y);
          we.z = fmaf(fk[k], wr[2][k], we.z); we.w = fmaf(fk[k], wr[3][k], we.w);
        }
        const float4 v = f4_mul(xn[u], we);
        request(u, j + U + u);            // unconditional (past the end: the last row again, a cache hit): a branch here makes the
                                          // compiler wait for EVERY gather in flight at each edge
        acc.x = fmaf(v.x, live, acc.x); acc.y = fmaf(v.y, live, acc.y);
        acc.z = fmaf(v.z, live, acc.z); acc.w = fmaf(v.w, live, acc.w);
        __builtin_amdgcn_sched_barrier(0);      // (left alone the scheduler hoists the feature reads of all U edges: 198 VGPRs)
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the next chunk rewrites the feature rows
    __builtin_amdgcn_wave_barrier();
  }
  if (add) f4_acc(acc, add[(int64_t)w * 64 + lane]);
  out[(int64_t)w * 64 + lane] = acc;
}

template <int LPR, int KT>
__global__ void __launch_bounds__(256) k_featconv(const float4* __restrict__ X, const int* __restrict__ ix,
                                                   const float* __restrict__ F, int Krt, const float* __restrict__ Wc,
                                                   const int* __restrict__ kptr, const int* __restrict__ map, int S,
                                                   float4* __restrict__ out, int swz, const float4* __restrict__ add) {
  if constexpr (LPR == 64 && (KT == 12 || KT == 6)) {
    featconv_wave<KT, FC_WAVE_U>(X, ix, F, Wc, kptr, map, S, out, swz, add);
    return;
  }
  int64_t w = ((int64_t)dig3d_xcd_block(swz) * blockDim.x + threadIdx.x) / LPR;
  // LPR == 64 (C = 256): one wave per segment, so the segment, its edges and their feature rows are wave-uniform —
  // told to the compiler (readfirstlane), the CSR / index / feature reads become scalar loads instead of 64-lane
  // vector loads of one address (12 of them per edge for the features alone)
  if (LPR == 64) w = __builtin_amdgcn_readfirstlane((int)w);
  const int c = threadIdx.x % LPR;
  const int K = KT ? KT : Krt;
  constexpr int KL = KT ? KT : FC_KMAX;   // loop bound
  float wr[4][KL];                        // this lane's four rows of Wc
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < KL; ++k) wr[j][k] = k < K ? Wc[(4 * c + j) * K + k] : 0.f;
  if (w >= S) return;
  const int b = kptr[w], e = kptr[w + 1];
  // (Measured and not kept: a graph-local variant — the workgroups of a molecule stage its 128 rows in LDS once and gather
  // from there — 118 us against 85 us per launch in the ComENet step, 7.66 vs 7.17 ms per step.  The 1-KB row gathers are
  // not what bounds this kernel: per edge it issues 24 packed FMAs for the weight plus ~15 scalar / address instructions.)
  float4 acc = f4_zero();
  constexpr int U = 4;                    // edges per batch
  // Software pipeline: the edge ids and gathered rows of batch i + 1 are requested BEFORE batch i is consumed.  One batch
  // at a time, a wave walked its ~32 edges as 8 dependent round trips (edge id -> source row -> 1-KB row, ~1.5 us each,
  // 6 waves per SIMD to hide them: 62 us per launch at 5.2e5 edges); with the next batch in flight only the first trip is
  // exposed.
  // Everything in a request is UNCONDITIONAL (positions past the end are clamped to the segment's last edge, their
  // contribution is multiplied by 0): a predicate at a load puts it in its own branch, and the four index -> row chains
  // of a batch then run one after the other instead of side by side.
  int tn[U];
  float4 xn[U];
  auto request = [&](int p) {
    int pp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) pp[u] = p + u < e ? p + u : e - 1;
    if (map) {
#pragma unroll
      for (int u = 0; u < U; ++u) tn[u] = map[pp[u]];
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) tn[u] = pp[u];
    }
    int row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (LPR == 64) tn[u] = __builtin_amdgcn_readfirstlane(tn[u]);
      row[u] = ix[tn[u]];
      if (LPR == 64) row[u] = __builtin_amdgcn_readfirstlane(row[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) xn[u] = X[(int64_t)row[u] * LPR + c];
  };
  if (b < e) request(b);
  for (int p = b; p < e; p += U) {
    int t[U];
    float4 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      t[u] = tn[u];
      x[u] = xn[u];
    }
    if (p + U < e) request(p + U);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float live = p + u < e ? 1.0f : 0.0f;
      const float* __restrict__ f = F + (int64_t)t[u] * K;
      float4 we = f4_zero();
#pragma unroll
      for (int k = 0; k < KL; ++k) {
        if (KT || k < K) {
          const float fk = f[k];
          we.x = fmaf(fk, wr[0][k], we.x); we.y = fmaf(fk, wr[1][k], we.y);
          we.z = fmaf(fk, wr[2][k], we.z); we.w = fmaf(fk, wr[3][k], we.w);
        }
      }
      const float4 v = f4_mul(x[u], we);
      acc.x = fmaf(v.x, live, acc.x); acc.y = fmaf(v.y, live, acc.y);
      acc.z = fmaf(v.z, live, acc.z); acc.w = fmaf(v.w, live, acc.w);
    }
  }
  if (add) f4_acc(acc, add[(int64_t)w * LPR + c]);       // a gradient already accumulated on these rows (see dig3d_featconv)
  out[(int64_t)w * LPR + c] = acc;
}

// ---- the weight gradient at C = 256 with a compile-time K: the same idea as featconv_wave ----------------------------
// The general body below reads ig[t], ix[t] -> X row (-> G row when the target changes) -> K scalar feature loads inside ONE
// iteration of four edges: two to three dependent trips per iteration with nothing in flight behind them (a wave lives
// 66 iterations x ~1.9 us at 5.2e5 edges over 1 984 waves; VALU busy 14 %, profiles/r05_stall_counters_comenet_128.json).
// Here lane l reads both indices and the feature row of edge l of a 64-edge chunk (one vector trip, the rows to LDS), a
// ballot marks where the target row changes, and the chunk is walked in batches of at most U edges OF ONE TARGET: U row
// gathers + that target's G row per batch, the next batch requested before this one is consumed.  Every load in the loop is
// unconditional (a load under a branch makes the compiler drain all gathers in flight at each use).  Same products in the
// same order per wave as the general body: bit-identical partials.
template <int K, int U>
__device__ __forceinline__ void featconv_wgrad_wave(const float4* __restrict__ G, const int* __restrict__ ig,
                                                    const float4* __restrict__ X, const int* __restrict__ ix,
                                                    const float* __restrict__ F, int64_t t0, int64_t t1, float* sf,
                                                    float (&gw)[4][K]) {
  constexpr int VW = (K % 4 == 0) ? 4 : 2;
  constexpr int NV = K / VW;
  const int lane = threadIdx.x & 63;
  for (int64_t tb = t0; tb < t1; tb += 64) {
    const int n = t1 - tb < 64 ? (int)(t1 - tb) : 64;            // wave-uniform
    const int64_t pl = tb + (lane < n ? lane : n - 1);
    const int rgl = ig[pl], rxl = ix[pl];
    {
      const float* f = F + pl * K;
      if (VW == 4) {
        float4 v[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = reinterpret_cast<const float4*>(f)[q];
#pragma unroll
        for (int q = 0; q < NV; ++q) reinterpret_cast<float4*>(sf + lane * K)[q] = v[q];
      } else {
        float2 v[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = reinterpret_cast<const float2*>(f)[q];
#pragma unroll
        for (int q = 0; q < NV; ++q) reinterpret_cast<float2*>(sf + lane * K)[q] = v[q];
      }
    }
    const int prev = __shfl_up(rgl, 1);
    const unsigned long long starts = __ballot(lane < n && (lane == 0 || rgl != prev));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    auto batch_len = [&](int j) {                                 // edges of j's target from j on, at most U
      const unsigned long long m = j < 63 ? starts >> (j + 1) : 0ull;
      const int end = m ? j + 1 + __builtin_ctzll(m) : n;
      return end - j < U ? end - j : U;
    };
    float4 xn[U], gn;
    int lenn;
    auto request = [&](int j0) {
      lenn = batch_len(j0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int jj = j0 + (u < lenn ? u : lenn - 1);
        xn[u] = X[(int64_t)__builtin_amdgcn_readlane(rxl, jj) * 64 + lane];
      }
      gn = G[(int64_t)__builtin_amdgcn_readlane(rgl, j0) * 64 + lane];
    };
    request(0);
    for (int j = 0; j < n;) {
      float4 x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = xn[u];
      const float4 g = gn;
      const int len = lenn;
      const int jn = j + len;
      request(jn < n ? jn : j);          // (past the end: this batch again — cache hits, never consumed)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float4 p = u < len ? f4_mul(g, x[u]) : f4_zero();
        const float* __restrict__ fr = sf + (j + (u < len ? u : len - 1)) * K;
        float fk[K];
        if (VW == 4) {
#pragma unroll
          for (int q = 0; q < NV; ++q) {
            const float4 t = reinterpret_cast<const float4*>(fr)[q];
            fk[4 * q] = t.x; fk[4 * q + 1] = t.y; fk[4 * q + 2] = t.z; fk[4 * q + 3] = t.w;
          }
        } else {
#pragma unroll
          for (int q = 0; q < NV; ++q) {
            const float2 t = reinterpret_cast<const float2*>(fr)[q];
            fk[2 * q] = t.x; fk[2 * q + 1] = t.y;
          }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
          gw[0][k] = fmaf(fk[k], p.x, gw[0][k]); gw[1][k] = fmaf(fk[k], p.y, gw[1][k]);
          gw[2][k] = fmaf(fk[k], p.z, gw[2][k]); gw[3][k] = fmaf(fk[k], p.w, gw[3][k]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      j = jn;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the next chunk rewrites the feature rows
    __builtin_amdgcn_wave_barrier();
  }
}

template <int LPR, int KT>
__global__ void __launch_bounds__(256) k_featconv_wgrad(const float4* __restrict__ G, const int* __restrict__ ig,
                                                         const float4* __restrict__ X, const int* __restrict__ ix,
                                                         const float* __restrict__ F, int Krt, int64_t M,
                                                         float* __restrict__ part, int swz, const int* __restrict__ cnt) {
  constexpr int NG = 256 / LPR;           // lane groups per block; each walks its own slice of the edges
  if (cnt && *cnt < M) M = *cnt;          // static-shape batch: the live edges (padded ones point at row 0 on both ends)
  constexpr int KL = KT ? KT : FC_KMAX;
  const int K = KT ? KT : Krt;
  __shared__ float sm[LPR * 4 * KL];
  const int grp = threadIdx.x / LPR, c = threadIdx.x % LPR;
  float gw[4][KL];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < KL; ++k) gw[j][k] = 0.f;
  const int64_t ngroups = (int64_t)gridDim.x * NG;
  int64_t gid = (int64_t)dig3d_xcd_block(swz) * NG + grp;
  if (LPR == 64) gid = __builtin_amdgcn_readfirstlane((int)gid);     // wave-uniform edge range: scalar index / feature loads
  const int64_t per = (M + ngroups - 1) / ngroups;
  const int64_t t0 = gid * per, t1 = t0 + per < M ? t0 + per : M;
  if constexpr (LPR == 64 && (KT == 12 || KT == 6)) {
    if (t0 < t1) featconv_wgrad_wave<KT, FC_WGRAD_U>(G, ig, X, ix, F, t0, t1, sm + grp * (64 * KT), gw);
    __syncthreads();                      // the feature rows shared sm with the block's sum below
  } else {
  constexpr int U = 4;                    // edges in flight (dependent index -> row gathers)
  int cur_rg = -1;                        // the G row is re-read only when ig[t] changes (edge lists sorted by one end
  float4 gcur = f4_zero();                // keep it for ~32 consecutive edges)
  for (int64_t tb = t0; tb < t1; tb += U) {
    float4 pr[U];
    int rgs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t t = tb + u;
      int rg = t < t1 ? ig[t] : -1, rx = t < t1 ? ix[t] : 0;
      if (LPR == 64) rg = __builtin_amdgcn_readfirstlane(rg), rx = __builtin_amdgcn_readfirstlane(rx);
      rgs[u] = rg;
      pr[u] = t < t1 ? X[(int64_t)rx * LPR + c] : f4_zero();
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (rgs[u] >= 0 && rgs[u] != cur_rg) {
        cur_rg = rgs[u];
        gcur = G[(int64_t)cur_rg * LPR + c];
      }
      pr[u] = rgs[u] >= 0 ? f4_mul(gcur, pr[u]) : f4_zero();
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t t = tb + u < t1 ? tb + u : t0;     // (pr is zero past the end)
      const float* __restrict__ f = F + t * K;
      const float4 p = pr[u];
#pragma unroll
      for (int k = 0; k < KL; ++k) {
        if (KT || k < K) {
          const float fk = f[k];
          gw[0][k] = fmaf(fk, p.x, gw[0][k]); gw[1][k] = fmaf(fk, p.y, gw[1][k]);
          gw[2][k] = fmaf(fk, p.z, gw[2][k]); gw[3][k] = fmaf(fk, p.w, gw[3][k]);
        }
      }
    }
  }
  }
  // the block's groups add up through LDS, one after the other
  for (int g = 0; g < NG; ++g) {
    if (grp == g) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < KL; ++k) {
          float* q = sm + (c * 4 + j) * KL + k;
          *q = g == 0 ? gw[j][k] : *q + gw[j][k];
        }
    }
    __syncthreads();
  }
  float* outp = part + (int64_t)blockIdx.x * (LPR * 4 * K);
  for (int q = threadIdx.x; q < LPR * 4 * K; q += 256) {
    const int row = q / K, k = q - row * K;
    outp[q] = sm[row * KL + k];
  }
}

// out[j] = sum_b part[b, j]: 64 columns x 16 slices of the partials per workgroup (512 partials of ComENet's feature-
// weight gradient: 64 dependent additions per thread with 4 slices, 18.8 us per launch; 8 with 16)
__global__ void __launch_bounds__(1024) k_part_reduce(const float* __restrict__ part, int nparts, int n,
                                                       float* __restrict__ out) {
  __shared__ float red[16][64];
  const int jj = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + jj;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (j < n) {
    int b = pl;
    for (; b + 48 < nparts; b += 64) {
      s0 += part[(int64_t)b * n + j];
      s1 += part[(int64_t)(b + 16) * n + j];
      s2 += part[(int64_t)(b + 32) * n + j];
      s3 += part[(int64_t)(b + 48) * n + j];
    }
    for (; b < nparts; b += 16) s0 += part[(int64_t)b * n + j];
  }
  red[pl][jj] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (pl == 0 && j < n) {
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) v += red[q][jj];
    out[j] = v;
  }
}

// ================================================================================================
// (f) backward of an embedding lookup x = weight[idx] (spherenet.py:85 `self.emb(z)`, V <= 128 atom types):
//     gW[v,c] = sum_{m: idx[m] = v} g[m,c], two phases, deterministic.  Phase 1: a block takes `rows` (64 ... M/64) rows x 64 channels and
//     accumulates into FOUR [V][64] tables in LDS (one per row lane: lane l adds rows l, l+4, ... in order; the row's type is
//     wave-uniform, so it is a scalar load and the table address is a shared-row add), sums the four tables in a fixed
//     order and writes one partial table per block.  Phase 2: k_part_reduce over the row chunks.
//     (A one-thread-per-output scan of the index was measured first: 50 us vs the framework's 39 us at 608 atoms.)
// ================================================================================================
__global__ void __launch_bounds__(256) k_embedding_bwd_part(const int64_t* __restrict__ idx, const float* __restrict__ g,
                                                             int M, int V, int C, float* __restrict__ part, int rows) {
  extern __shared__ float etab[];                    // [4][V][64]
  const int c = threadIdx.x & 63;
  int rl = threadIdx.x >> 6;
  rl = __builtin_amdgcn_readfirstlane(rl);           // one wave per row lane
  const int cg = blockIdx.y * 64 + c;
  for (int q = threadIdx.x; q < 4 * V * 64; q += 256) etab[q] = 0.f;
  __syncthreads();
  float* __restrict__ tab = etab + rl * V * 64 + c;
  const int r0 = blockIdx.x * rows;
  const int r1 = r0 + rows < M ? r0 + rows : M;
  constexpr int U = 4;                               // rows in flight per lane
  for (int r = r0 + rl; r < r1; r += 4 * U) {
    int v[U];
    float x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rr = r + 4 * u;
      v[u] = rr < r1 ? (int)idx[rr] : -1;
      x[u] = (rr < r1 && cg < C) ? g[(int64_t)rr * C + cg] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (v[u] >= 0 && v[u] < V) tab[v[u] * 64] += x[u];
  }
  __syncthreads();
  float* __restrict__ outp = part + (int64_t)blockIdx.x * V * C;
  for (int q = threadIdx.x; q < V * 64; q += 256) {
    const int vv = q >> 6, cc = q & 63;
    const int col = blockIdx.y * 64 + cc;
    if (col < C)
      outp[(int64_t)vv * C + col] = (etab[q] + etab[V * 64 + q]) + (etab[2 * V * 64 + q] + etab[3 * V * 64 + q]);
  }
}

__global__ void __launch_bounds__(256) k_embedding_fwd(const int64_t* __restrict__ idx, const float4* __restrict__ w, int M,
                                                        int V, int c4, float4* __restrict__ out) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= (int64_t)M * c4) return;
  const int m = (int)(q / c4);
  int64_t v = idx[m];
  v = v < 0 ? 0 : (v >= V ? V - 1 : v);
  out[q] = w[v * c4 + (q - (int64_t)m * c4)];
}

// (g) torch.cat([x[i], x[j], r], -1) for the edge initialisation (see dig3d_edge_cat) ---------------------------------------
// z != NULL: x is the EMBEDDING TABLE and node n's row is x[z[n]] (the lookup of method/spherenet/spherenet.py:84 folded in:
// no [N, Cx] node-feature tensor, no launch for it)
__global__ void __launch_bounds__(256) k_edge_cat(const float4* __restrict__ x, const int* __restrict__ ei,
                                                   const int* __restrict__ ej, const float4* __restrict__ r, int64_t E,
                                                   int cx4, int cr4, float4* __restrict__ out,
                                                   const int* __restrict__ cnt, const int64_t* __restrict__ z) {
  const int row4 = 2 * cx4 + cr4;
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= E * row4) return;
  const int64_t e = q / row4;
  const int c = (int)(q - e * row4);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c >= 2 * cx4) {
    v = r[e * cr4 + (c - 2 * cx4)];
  } else if (!cnt || e < *cnt) {           // gathered rows past the live count of a padded batch are zero (as dig3d_gather_mul)
    int64_t row = c < cx4 ? ei[e] : ej[e];
    if (z) row = z[row];
    v = x[row * cx4 + (c < cx4 ? c : c - cx4)];
  }
  out[q] = v;
}

// blocks [0, nbA): 4 x cx4 lanes per node — sub-group q takes the node's CSR positions p = q (mod 4) of the incoming (by i)
// and then of the outgoing (by j) gradient rows, the four partial sums are added in order through LDS (15 + 15 dependent
// index -> row loads per thread with one sub-group: 23 us at 600 atoms);  blocks [nbA, ...): the last Cr columns copied out
__global__ void __launch_bounds__(256) k_edge_cat_bwd(const float4* __restrict__ G, const int* __restrict__ kptr_i,
                                                       const int* __restrict__ perm_i, const int* __restrict__ kptr_j,
                                                       const int* __restrict__ perm_j, int N, int64_t E, int cx4, int cr4,
                                                       int nbA, float4* __restrict__ gx, float4* __restrict__ gr) {
  __shared__ float4 sh[256];
  const int row4 = 2 * cx4 + cr4;
  if ((int)blockIdx.x < nbA) {
    const int npb = 64 / cx4;                        // nodes per block (256 threads = npb nodes x 4 sub-groups x cx4 lanes)
    const int ln = threadIdx.x / (4 * cx4), q = (threadIdx.x / cx4) & 3, c = threadIdx.x % cx4;
    const int n = blockIdx.x * npb + ln;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N) {
      for (int p = kptr_i[n] + q, en = kptr_i[n + 1]; p < en; p += 4) {
        const int64_t e = perm_i ? perm_i[p] : p;
        f4_acc(a, G[e * row4 + c]);
      }
      for (int p = kptr_j[n] + q, en = kptr_j[n + 1]; p < en; p += 4) {
        const int64_t e = perm_j ? perm_j[p] : p;
        f4_acc(a, G[e * row4 + cx4 + c]);
      }
    }
    sh[threadIdx.x] = a;
    __syncthreads();
    if (q == 0 && n < N) {
      const float4* s4 = sh + ln * 4 * cx4 + c;
      float4 v = s4[0];
      f4_acc(v, s4[cx4]);
      f4_acc(v, s4[2 * cx4]);
      f4_acc(v, s4[3 * cx4]);
      gx[(int64_t)n * cx4 + c] = v;
    }
    return;
  }
  const int64_t qq = (int64_t)(blockIdx.x - nbA) * 256 + threadIdx.x;
  if (qq >= E * cr4) return;
  const int64_t e = qq / cr4;
  gr[qq] = G[e * row4 + 2 * cx4 + (qq - e * cr4)];
}

extern "C" {

// out[S,C] = scatter_add(src[M,C], index[M]) for a sorted int64 index in [0,S).  torch_scatter.scatter
// (reduce='sum', dim=0) semantics: rows of `out` with no source row are zero.
static int segment_sorted_impl(const float* src, const int64_t* index, int64_t M, int C, int64_t S, float* out,
                               int rows_per_worker, int mode, int mean, void* stream) {
  DIG3D_ENTER();
  hipStream_t st = (hipStream_t)stream;
  if (M < 0 || S < 0 || C <= 0 || rows_per_worker < 0 || mode < 0 || mode > 3) return DIG3D_ERR_ARG;
  const int L = rows_per_worker;
  if (S == 0) return DIG3D_OK;
  if (M == 0) {
    if (dig3d_zero_async(out, sizeof(float) * (size_t)S * C, st) != hipSuccess) return DIG3D_ERR_LAUNCH;
    return DIG3D_OK;
  }
  const bool aligned = (((uintptr_t)src | (uintptr_t)out) & 15) == 0;
  if (aligned && C == 32) launch_sorted<8>(src, index, M, S, out, L, mode, mean, st);
  else if (aligned && C == 64) launch_sorted<16>(src, index, M, S, out, L, mode, mean, st);
  else if (aligned && C == 128) launch_sorted<32>(src, index, M, S, out, L, mode, mean, st);
  else if (aligned && C == 256) launch_sorted<64>(src, index, M, S, out, L, mode, mean, st);
  else
    hipLaunchKernelGGL(k_segsum_sorted_generic, dim3(dig3d_blocks(S * C, 256)), dim3(256), 0, st, src, index, M, S,
                       C, out, mean);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_segment_sum_sorted_tuned(const float* src, const int64_t* index, int64_t M, int C, int64_t S, float* out,
                                   int rows_per_worker, int mode, void* stream) {
  return segment_sorted_impl(src, index, M, C, S, out, rows_per_worker, mode, 0, stream);
}

// torch_scatter.scatter(..., reduce='mean') for a sorted int64 index: the same pass, every output row divided by its
// row count (empty rows stay 0) — dig/ggraph3D/method/G_SphereNet/model/spherenet.py:171-172,205,297.
int dig3d_segment_mean_sorted(const float* src, const int64_t* index, int64_t M, int C, int64_t S, float* out,
                              void* stream) {
  return segment_sorted_impl(src, index, M, C, S, out, 0, 3, 1, stream);
}

int dig3d_segment_sum_sorted(const float* src, const int64_t* index, int64_t M, int C, int64_t S, float* out,
                             void* stream) {
  return dig3d_segment_sum_sorted_tuned(src, index, M, C, S, out, 0, 3, stream);
}

// out[S,C] = sum over CSR segments of  A[t,:] * X[ix[t],:] * B[t,:]   (any of X/ix, A, B, map may be null,
// at least one of X, A non-null).  kptr[S+1]; t = map ? map[p] : p.
#define kSegPipeMaxBlocks (8 * dig3d_num_cus())   // below ~2 waves per SIMD of groups: the pipelined loop (k_seg_fused<LPR, true>)
static const bool kXcdSwizzle = true;      // XCD-contiguous block order of the gather kernels (common.h: dig3d_xcd_block)

static int segment_fused_impl(const float* X, const int* ix, const float* A, const float* B, const int* kptr,
                              const int* map, int S, int C, float* out, int mean, void* stream) {
  DIG3D_ENTER();
  hipStream_t st = (hipStream_t)stream;
  if (S < 0 || C <= 0 || (!X && !A)) return DIG3D_ERR_ARG;
  if (S == 0) return DIG3D_OK;
  const bool aligned = (((uintptr_t)X | (uintptr_t)A | (uintptr_t)B | (uintptr_t)out) & 15) == 0;
  // gathers: XCD-contiguous block order (see xcd_block); pure streams keep the natural order
#define LAUNCH_FUSED(LPR)                                                                                   \
  do {                                                                                                      \
    const int nblk = dig3d_blocks((int64_t)S * LPR, 256);                                                   \
    const int swz = (kXcdSwizzle && X && ix && nblk >= 64) ? 1 : 0;                                         \
    if (nblk < kSegPipeMaxBlocks || !X)                                                                     \
      hipLaunchKernelGGL((k_seg_fused<LPR, true>), dim3(swz ? dig3d_xcd_grid(nblk) : nblk), dim3(256), 0, st,     \
                         (const float4*)X, ix, (const float4*)A, (const float4*)B, kptr, map, S, (float4*)out, \
                         mean, swz);                                                                        \
    else                                                                                                    \
      hipLaunchKernelGGL((k_seg_fused<LPR, false>), dim3(swz ? dig3d_xcd_grid(nblk) : nblk), dim3(256), 0, st,    \
                         (const float4*)X, ix, (const float4*)A, (const float4*)B, kptr, map, S, (float4*)out, \
                         mean, swz);                                                                        \
  } while (0)
  if (aligned && C == 32) LAUNCH_FUSED(8);
  else if (aligned && C == 64) LAUNCH_FUSED(16);
  else if (aligned && C == 128) LAUNCH_FUSED(32);
  else if (aligned && C == 256) LAUNCH_FUSED(64);
  else
    hipLaunchKernelGGL(k_seg_fused_generic, dim3(dig3d_blocks((int64_t)S * C, 256)), dim3(256), 0, st, X, ix, A, B,
                       kptr, map, S, C, out, mean);
#undef LAUNCH_FUSED
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_segment_fused(const float* X, const int* ix, const float* A, const float* B, const int* kptr,
                        const int* map, int S, int C, float* out, void* stream) {
  return segment_fused_impl(X, ix, A, B, kptr, map, S, C, out, 0, stream);
}

// the 'mean' flavour: every output row divided by its segment length (CSR driven; unsorted keys through `map`).
int dig3d_segment_fused_mean(const float* X, const int* ix, const float* A, const float* B, const int* kptr,
                             const int* map, int S, int C, float* out, void* stream) {
  return segment_fused_impl(X, ix, A, B, kptr, map, S, C, out, 1, stream);
}

// out[M,C] = X[ix[m],:] * A[m,:] * B[m,:]   (A, B optional).  cnt (device, optional): rows m >= *cnt are
// padding of a static-shape batch and are written as zeros.
int dig3d_gather_mul(const float* X, const int* ix, const float* A, const float* B, int64_t M, int C, float* out,
                     const int* cnt, void* stream) {
  DIG3D_ENTER();
  hipStream_t st = (hipStream_t)stream;
  if (M < 0 || C <= 0 || !X || !ix) return DIG3D_ERR_ARG;
  if (M == 0) return DIG3D_OK;
  const bool aligned = (((uintptr_t)X | (uintptr_t)A | (uintptr_t)B | (uintptr_t)out) & 15) == 0;
  if (aligned && (C & 3) == 0)
    hipLaunchKernelGGL(k_gather_mul, dim3(dig3d_blocks(M * (C / 4), 256)), dim3(256), 0, st, (const float4*)X, ix,
                       (const float4*)A, (const float4*)B, M, C / 4, (float4*)out, cnt);
  else
    hipLaunchKernelGGL(k_gather_mul_generic, dim3(dig3d_blocks(M * C, 256)), dim3(256), 0, st, X, ix, A, B, M, C,
                       out, cnt);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// P = G[ig[m]] * X[ix[m]];  outA = P * B (B optional -> P),  outB = P * A.   C % 4 == 0 required.
int dig3d_gather_mul2(const float* G, const int* ig, const float* X, const int* ix, const float* A, const float* B,
                      int64_t M, int C, float* outA, float* outB, const int* cnt, void* stream) {
  DIG3D_ENTER();
  hipStream_t st = (hipStream_t)stream;
  if (M < 0 || C <= 0 || (C & 3) || !G || !X || (outB && !A)) return DIG3D_ERR_ARG;
  if ((((uintptr_t)G | (uintptr_t)X | (uintptr_t)A | (uintptr_t)B | (uintptr_t)outA | (uintptr_t)outB) & 15) != 0)
    return DIG3D_ERR_ARG;
  if (M == 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_gather_mul2, dim3(dig3d_blocks(M * (C / 4), 256)), dim3(256), 0, st, (const float4*)G, ig,
                     (const float4*)X, ix, (const float4*)A, (const float4*)B, M, C / 4, (float4*)outA,
                     (float4*)outB, cnt);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// out[S,C] = sum_{t in seg(s)} X[ix[t],:] * (Wc f_t): the EdgeGraphConv aggregation of ComENet with the edge weight
// evaluated on the fly (see k_featconv).  F [M,K] row-major, Wc [C,K], K <= 16, C in {64, 128, 256}; kptr/map as in
// dig3d_segment_fused.
int dig3d_featconv_supported(int K, int C) { return (K >= 1 && K <= FC_KMAX && (C == 64 || C == 128 || C == 256)) ? 1 : 0; }

int dig3d_featconv(const float* X, const int* ix, const float* F, int K, const float* Wc, const int* kptr,
                   const int* map, int S, int C, float* out, const float* add, void* stream) {
  DIG3D_ENTER();
  if (S < 0 || !dig3d_featconv_supported(K, C) || !X || !ix || !F || !Wc || !kptr || !out) return DIG3D_ERR_ARG;
  if ((((uintptr_t)X | (uintptr_t)out | (uintptr_t)add) & 15) != 0) return DIG3D_ERR_ARG;
  if (S == 0) return DIG3D_OK;
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH_FC(LPR)                                                                                          \
  do {                                                                                                             \
    const int nblk = dig3d_blocks((int64_t)S * LPR, 256);                                                          \
    const int swz = (kXcdSwizzle && nblk >= 64) ? 1 : 0;                                                           \
    if (K == 12)                                                                                                   \
      hipLaunchKernelGGL((k_featconv<LPR, 12>), dim3(swz ? dig3d_xcd_grid(nblk) : nblk), dim3(256), 0, st, (const float4*)X, \
                         ix, F, K, Wc, kptr, map, S, (float4*)out, swz, (const float4*)add);                       \
    else if (K == 6)                                                                                               \
      hipLaunchKernelGGL((k_featconv<LPR, 6>), dim3(swz ? dig3d_xcd_grid(nblk) : nblk), dim3(256), 0, st, (const float4*)X, \
                         ix, F, K, Wc, kptr, map, S, (float4*)out, swz, (const float4*)add);                       \
    else                                                                                                           \
      hipLaunchKernelGGL((k_featconv<LPR, 0>), dim3(swz ? dig3d_xcd_grid(nblk) : nblk), dim3(256), 0, st, (const float4*)X, \
                         ix, F, K, Wc, kptr, map, S, (float4*)out, swz, (const float4*)add);                       \
  } while (0)
  if (C == 256) LAUNCH_FC(64);
  else if (C == 128) LAUNCH_FC(32);
  else LAUNCH_FC(16);
#undef LAUNCH_FC
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_featconv_wgrad_blocks(int64_t M) {
  int64_t nb = (M + 1023) / 1024;         // >= 256 edges per lane group
  if (nb > 512) nb = 512;             // (768: 6.390 vs 6.394 ms per config-5 step, same box)
  if (nb >= 64) nb &= ~(int64_t)7;        // multiple of 8: XCD-contiguous edge ranges
  return nb < 1 ? 1 : (int)nb;
}

// gWc[C,K] = sum_t f_t[k] * G[ig[t],c] * X[ix[t],c] over the M edges; part: float[blocks * C*K].
int dig3d_featconv_wgrad(const float* G, const int* ig, const float* X, const int* ix, const float* F, int K, int64_t M,
                         int C, float* part, float* gWc, int reduce_now, const int* cnt, void* stream) {
  DIG3D_ENTER();
  if (M < 0 || !dig3d_featconv_supported(K, C) || !G || !ig || !X || !ix || !F || !part || !gWc) return DIG3D_ERR_ARG;
  if ((((uintptr_t)G | (uintptr_t)X) & 15) != 0) return DIG3D_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    if (dig3d_zero_async(gWc, sizeof(float) * (size_t)C * K, st) != hipSuccess) return DIG3D_ERR_LAUNCH;
    return DIG3D_OK;
  }
  const int nb = dig3d_featconv_wgrad_blocks(M);
#define LAUNCH_FW1(LPR, KT)                                                                                      \
  hipLaunchKernelGGL((k_featconv_wgrad<LPR, KT>), dim3(nb), dim3(256), 0, st, (const float4*)G, ig, (const float4*)X, \
                     ix, F, K, M, part, (kXcdSwizzle && (nb & 7) == 0 && nb >= 64) ? 1 : 0, cnt)
#define LAUNCH_FW(LPR)                                                                                           \
  do {                                                                                                           \
    if (K == 12) LAUNCH_FW1(LPR, 12);                                                                            \
    else if (K == 6) LAUNCH_FW1(LPR, 6);                                                                         \
    else LAUNCH_FW1(LPR, 0);                                                                                     \
  } while (0)
  if (C == 256) LAUNCH_FW(64);
  else if (C == 128) LAUNCH_FW(32);
  else LAUNCH_FW(16);
#undef LAUNCH_FW
#undef LAUNCH_FW1
  DIG3D_CHECK_LAUNCH();
  if (reduce_now) {
    hipLaunchKernelGGL(k_part_reduce, dim3((C * K + 63) / 64), dim3(1024), 0, st, part, nb, C * K, gWc);
    DIG3D_CHECK_LAUNCH();
  }
  return DIG3D_OK;
}

// out[m,:] = weight[idx[m],:]
int dig3d_embedding_fwd(const int64_t* idx, const float* weight, int M, int V, int C, float* out, void* stream) {
  DIG3D_ENTER();
  if (M < 0 || V < 1 || C < 4 || (C & 3) || !weight || !out || (M > 0 && !idx)) return DIG3D_ERR_ARG;
  if ((((uintptr_t)weight | (uintptr_t)out) & 15) != 0) return DIG3D_ERR_ARG;
  if (M == 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_embedding_fwd, dim3(dig3d_blocks((int64_t)M * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, idx,
                     (const float4*)weight, M, V, C / 4, (float4*)out);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// gW[V,C] = sum over the M rows of g grouped by idx (int64, values in [0, V), V <= 128): backward of weight[idx].
// part: float[dig3d_embedding_bwd_chunks(M) * V * C].
// rows per block: 64 at a few hundred atoms (a wave's rows are one batch of loads: 600 atoms 21 -> ~6 us), growing so that
// at most 64 partial tables are written
static int embedding_bwd_rows(int M) {
  int r = (M + 63) / 64;
  r = (r + 15) & ~15;
  return r < 64 ? 64 : r;
}
int dig3d_embedding_bwd_chunks(int M) {
  const int r = embedding_bwd_rows(M);
  return M <= 0 ? 1 : (M + r - 1) / r;
}

int dig3d_embedding_bwd(const int64_t* idx, const float* g, int M, int V, int C, float* part, float* gW, int reduce_now,
                        void* stream) {
  DIG3D_ENTER();
  if (M < 0 || V < 1 || V > 128 || C < 1 || !gW || !part || (M > 0 && (!idx || !g))) return DIG3D_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    if (dig3d_zero_async(gW, sizeof(float) * (size_t)V * C, st) != hipSuccess) return DIG3D_ERR_LAUNCH;
    return DIG3D_OK;
  }
  static const bool attr_ok = hipFuncSetAttribute((const void*)k_embedding_bwd_part,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   4 * 128 * 64 * 4) == hipSuccess;   // set once
  if (!attr_ok) return DIG3D_ERR_LAUNCH;
  const int nch = dig3d_embedding_bwd_chunks(M);
  hipLaunchKernelGGL(k_embedding_bwd_part, dim3(nch, (C + 63) / 64), dim3(256), sizeof(float) * 4 * V * 64, st, idx, g, M,
                     V, C, part, embedding_bwd_rows(M));
  DIG3D_CHECK_LAUNCH();
  if (reduce_now) {          // else the caller reduces the nch partial tables (dig3d_reduce_many, stride V*C)
    hipLaunchKernelGGL(k_part_reduce, dim3((V * C + 63) / 64), dim3(1024), 0, st, part, nch, V * C, gW);
    DIG3D_CHECK_LAUNCH();
  }
  return DIG3D_OK;
}


// ---- the input of the edge initialisation: torch.cat([x[i], x[j], rbf0], dim=-1) (method/spherenet/spherenet.py:88-89,
// dimenetpp.py:74-75) as one kernel, and its backward as one more: gx[n] = sum over the edges with i = n of the first Cx
// gradient columns + sum over the edges with j = n of the next Cx, gr[e] = the last Cr columns.  The framework ran two
// gathers + a cat forward and three slicing copies + two segment sums + an add backward.
int dig3d_edge_cat_supported(int Cx, int Cr) {
  return ((Cx == 64 || Cx == 128 || Cx == 256) && Cr > 0 && (Cr & 3) == 0) ? 1 : 0;
}

int dig3d_edge_cat(const float* x, const int* i, const int* j, const float* r, int64_t E, int Cx, int Cr, float* out,
                   const int* cnt, void* stream) {
  DIG3D_ENTER();
  if (E < 0 || !dig3d_edge_cat_supported(Cx, Cr) || !x || !i || !j || !r || !out) return DIG3D_ERR_ARG;
  if ((((uintptr_t)x | (uintptr_t)r | (uintptr_t)out) & 15) != 0) return DIG3D_ERR_ARG;
  if (E == 0) return DIG3D_OK;
  const int row4 = (2 * Cx + Cr) / 4;
  hipLaunchKernelGGL(k_edge_cat, dim3(dig3d_blocks(E * row4, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)x, i,
                     j, (const float4*)r, E, Cx / 4, Cr / 4, (float4*)out, cnt, (const int64_t*)nullptr);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// The same with the node rows looked up in an embedding table: out[e] = cat(table[z[i[e]]], table[z[j[e]]], r[e]); z int64 [N]
// with values in [0, rows of table) (the caller checks them: dig_amd check_z_bounds).
int dig3d_edge_cat_emb(const float* table, const int64_t* z, const int* i, const int* j, const float* r, int64_t E, int Cx,
                       int Cr, float* out, const int* cnt, void* stream) {
  DIG3D_ENTER();
  if (E < 0 || !dig3d_edge_cat_supported(Cx, Cr) || !table || !z || !i || !j || !r || !out) return DIG3D_ERR_ARG;
  if ((((uintptr_t)table | (uintptr_t)r | (uintptr_t)out) & 15) != 0) return DIG3D_ERR_ARG;
  if (E == 0) return DIG3D_OK;
  const int row4 = (2 * Cx + Cr) / 4;
  hipLaunchKernelGGL(k_edge_cat, dim3(dig3d_blocks(E * row4, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)table,
                     i, j, (const float4*)r, E, Cx / 4, Cr / 4, (float4*)out, cnt, z);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// G [E, 2 Cx + Cr]; (kptr_i, perm_i) / (kptr_j, perm_j): the edges grouped by i / by j (perm NULL: already in that order);
// gx [N, Cx], gr [E, Cr].
int dig3d_edge_cat_bwd(const float* G, const int* kptr_i, const int* perm_i, const int* kptr_j, const int* perm_j, int N,
                       int64_t E, int Cx, int Cr, float* gx, float* gr, void* stream) {
  DIG3D_ENTER();
  if (E < 0 || N < 0 || !dig3d_edge_cat_supported(Cx, Cr) || !G || !kptr_i || !kptr_j || !gx || !gr) return DIG3D_ERR_ARG;
  if ((((uintptr_t)G | (uintptr_t)gx | (uintptr_t)gr) & 15) != 0) return DIG3D_ERR_ARG;
  const int lpr = Cx / 4, npb = 64 / lpr;              // lanes per node, nodes per block (4 sub-groups of lanes per node)
  const int nbA = (N + npb - 1) / npb, nbB = dig3d_blocks(E * (Cr / 4), 256);
  if (N == 0 && E == 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_edge_cat_bwd, dim3(nbA + (E > 0 ? nbB : 0)), dim3(256), 0, (hipStream_t)stream, (const float4*)G,
                     kptr_i, perm_i, kptr_j, perm_j, N, E, Cx / 4, Cr / 4, nbA, (float4*)gx, (float4*)gr);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

}  // extern "C"
