// dig3d triplet interaction — the T-sized part of SphereNet / DimeNet++ update_e, fused end to end.
//
// Reference (method/spherenet/spherenet.py:163-171, dimenetpp.py:146-150), per interaction layer:
//     sbf = lin_sbf2(lin_sbf1(sbf))            [T, ns*nr]   -> 8 -> int_emb
//     t   = lin_t2(lin_t1(t))                  [T, ns^2*nr] -> 8 -> int_emb        (SphereNet only)
//     x_kj = x_kj[idx_kj] * sbf * t ;  x_kj = scatter(x_kj, idx_ji, dim_size=E)
// with sbf / t the Bessel x harmonic tables of features.py:213-222,256-263 ([T,42] and [T,294] floats at
// ns=7: 150 MB per batch of 32 molecules, re-read by every layer and again in backward).
//
// Here none of the T-wide tensors exists:
//   k_basis_project   basis(t) is evaluated in registers and immediately contracted with the stacked first
//                     Linear of ALL layers -> P[l][t][8]   (the only T-sized tensors kept: 32 B/triplet/layer)
//   k_trip_fwd        out[e] = sum_{t in seg(e)} X[kj[t]] * (W2s P_s[t]) * (W2t P_t[t])   (second Linear in
//                     registers, gather, products and the segment sum in one pass; through the transposed
//                     CSR the same kernel is the backward w.r.t. X)
//   k_trip_bwd        gP_s, gP_t (per triplet) and gW2s, gW2t (two-stage deterministic reduction)
//   k_basis_wgrad     gW1 = sum_t gP[t] (x) basis(t), basis recomputed, two-stage deterministic reduction
// All float32, contraction order fixed (no atomics) => run-to-run identical results.
#include <stdio.h>
#include "sph.h"
#include "basis_mfma.h"

// wave-per-segment forms (triplet_wave.hip): 0 = launched, 1 = channel count not covered
int trip_fwd_wave(const float* X, const int* ix, const float* Ps, const float* Pt, const float* W2s, const float* W2t,
                  const int* kptr, const int* map, int S, int C, float* out, const float* add, hipStream_t st);
int trip_fwd_lds(const float* X, const int* ix, const float* Ps, const float* Pt, const float* W2s, const float* W2t,
                  const int* kptr, const int* map, int S, int C, float* out, const float* add, hipStream_t st);
int trip_bwd_wave_blocks(int E, int C);
int trip_bwd_wave(const float* G, const float* X, const int* kj, const float* Ps, const float* Pt, const float* W2s,
                  const float* W2t, const int* tptr, int E, int C, float* gPs, float* gPt, float* part, int nb,
                  const float* gPs_add, const float* gPt_add, hipStream_t st);
int trip_bwd_lds(const float* G, const float* X, const int* kj, const float* Ps, const float* Pt, const float* W2s,
                  const float* W2t, const int* tptr, int E, int C, float* gPs, float* gPt, float* part, int nb,
                  const float* gPs_add, const float* gPt_add, hipStream_t st);

#define PB 8          // projected basis width per layer (basis_emb_size <= 8, zero padded)
#define PO 32         // stacked outputs handled per launch (4 layers x 8)

// ------------------------------------------------------------------------------------------------
// k_basis_project: one thread per triplet.
//   bes[E, NS*nr]   radial table (k_bessel);   g = kj[t]
//   Ws[(l*nr+n)*PO + o]            stacked+transposed lin_sbf1 weights, o = layer*8 + b   (zero padded)
//   Wt[((h)*nr+n)*PO + o]          same for lin_t1, h in [0, NS*NS), radial order = h % NS
//                                  (the reference's broadcast pairing, spherenet/features.py:262)
//   Ps/Pt[l][t][8]
// The weight index is wave-uniform => scalar loads; every FMA takes its weight from an SGPR.
// ------------------------------------------------------------------------------------------------
template <int NS, bool TOR>
__global__ void __launch_bounds__(128) k_basis_project(const float* __restrict__ bes, const int* __restrict__ kj,
                                                        const float* __restrict__ angle,
                                                        const float* __restrict__ torsion, int T, int nr,
                                                        const float* __restrict__ pref,
                                                        const float* __restrict__ Ws,
                                                        const float* __restrict__ Wt, int L,
                                                        float* __restrict__ Ps, float* __restrict__ Pt,
                                                        const int* __restrict__ cnt) {
  constexpr int H2 = NS * NS;
  __shared__ float sPref[NS_MAX * NS_MAX];
  extern __shared__ float sBes[];                    // [blockDim][KB|1]: this thread's radial row
  for (int q = threadIdx.x; q < NS_MAX * NS_MAX; q += blockDim.x) sPref[q] = pref[q];
  __syncthreads();
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T || (cnt && t >= *cnt)) return;   // padded triplets: P rows are only ever read through the CSR
  const int KB = NS * nr;
  float* __restrict__ brow = sBes + threadIdx.x * (KB | 1);
  {  // stage the gathered radial row first: independent loads, issued 8 at a time (the per-(l,n) dependent
     // global load was the latency floor of this kernel)
    const float* __restrict__ g = bes + (int64_t)kj[t] * KB;
    int k = 0;
    for (; k + 8 <= KB; k += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = g[k + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) brow[k + u] = v[u];
    }
    for (; k < KB; ++k) brow[k] = g[k];
  }
  float Y[TOR ? H2 : NS];
  real_sph_harm<NS>(angle[t], TOR ? torsion[t] : 0.f, sPref, !TOR, Y);
  float as[PO], at[PO];
#pragma unroll
  for (int o = 0; o < PO; ++o) { as[o] = 0.f; at[o] = 0.f; }
  // sbf part: Y_l0 * bes[l, n]
#pragma unroll
  for (int l = 0; l < NS; ++l) {
    const float y = TOR ? Y[l * l] : Y[l];
    for (int n = 0; n < nr; ++n) {
      const float b = y * brow[l * nr + n];
      const float* __restrict__ w = Ws + (int64_t)(l * nr + n) * PO;
#pragma unroll
      for (int o = 0; o < PO; ++o) as[o] = fmaf(b, w[o], as[o]);
    }
  }
  if (TOR) {
#pragma unroll
    for (int h = 0; h < H2; ++h) {
      const float y = Y[TOR ? h : 0];
      for (int n = 0; n < nr; ++n) {
        const float b = y * brow[(h % NS) * nr + n];
        const float* __restrict__ w = Wt + (int64_t)(h * nr + n) * PO;
#pragma unroll
        for (int o = 0; o < PO; ++o) at[o] = fmaf(b, w[o], at[o]);
      }
    }
  }
  for (int l = 0; l < L; ++l) {
    float4* ps = (float4*)(Ps + ((int64_t)l * T + t) * PB);
#pragma unroll
    for (int q = 0; q < PB / 4; ++q) {
      // compile-time register indices: select by layer with a uniform switch
      float4 v;
      switch (l) {
        case 0: v = make_float4(as[0 + 4 * q], as[1 + 4 * q], as[2 + 4 * q], as[3 + 4 * q]); break;
        case 1: v = make_float4(as[8 + 4 * q], as[9 + 4 * q], as[10 + 4 * q], as[11 + 4 * q]); break;
        case 2: v = make_float4(as[16 + 4 * q], as[17 + 4 * q], as[18 + 4 * q], as[19 + 4 * q]); break;
        default: v = make_float4(as[24 + 4 * q], as[25 + 4 * q], as[26 + 4 * q], as[27 + 4 * q]); break;
      }
      ps[q] = v;
    }
    if (TOR) {
      float4* pt = (float4*)(Pt + ((int64_t)l * T + t) * PB);
#pragma unroll
      for (int q = 0; q < PB / 4; ++q) {
        float4 v;
        switch (l) {
          case 0: v = make_float4(at[0 + 4 * q], at[1 + 4 * q], at[2 + 4 * q], at[3 + 4 * q]); break;
          case 1: v = make_float4(at[8 + 4 * q], at[9 + 4 * q], at[10 + 4 * q], at[11 + 4 * q]); break;
          case 2: v = make_float4(at[16 + 4 * q], at[17 + 4 * q], at[18 + 4 * q], at[19 + 4 * q]); break;
          default: v = make_float4(at[24 + 4 * q], at[25 + 4 * q], at[26 + 4 * q], at[27 + 4 * q]); break;
        }
        pt[q] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_trip_fwd: worker = LPR lanes per output segment (C = 4*LPR channels, one float4 per lane).
//   out[s, c] = sum_{p in [kptr[s], kptr[s+1])}  X[ix[t], c] * ws(t, c) * wt(t, c),   t = map ? map[p] : p
//   ws(t, c) = sum_b W2s[c, b] * Ps[t, b]            (lin_sbf2 applied on the fly; W2 rows in registers)
// forward: (X = x_kj, ix = idx_kj, kptr = tptr, map = null); backward w.r.t. x_kj: (X = G, ix = idx_ji,
// kptr/map = transposed CSR of idx_kj).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dot8(const float* __restrict__ w, const float4 a, const float4 b) {
  float s = w[0] * a.x;
  s = fmaf(w[1], a.y, s); s = fmaf(w[2], a.z, s); s = fmaf(w[3], a.w, s);
  s = fmaf(w[4], b.x, s); s = fmaf(w[5], b.y, s); s = fmaf(w[6], b.z, s); s = fmaf(w[7], b.w, s);
  return s;
}

template <int LPR, bool TOR>
__global__ void __launch_bounds__(256) k_trip_fwd(const float4* __restrict__ X, const int* __restrict__ ix,
                                                   const float4* __restrict__ Ps, const float4* __restrict__ Pt,
                                                   const float* __restrict__ W2s, const float* __restrict__ W2t,
                                                   const int* __restrict__ kptr, const int* __restrict__ map,
                                                   int S, float4* __restrict__ out) {
  const int w = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPR);
  const int c = threadIdx.x % LPR;
  if (w >= S) return;
  float ws_w[4][PB], wt_w[4][PB];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      ws_w[q][b] = W2s[(4 * c + q) * PB + b];
      wt_w[q][b] = TOR ? W2t[(4 * c + q) * PB + b] : 0.f;
    }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int p0 = kptr[w], p1 = kptr[w + 1];
  // 4 triplets per trip: all basis / row loads of a batch are issued before the first use (a segment has ~13 triplets; one
  // at a time the loop was a chain of dependent L2 round trips).  r04: the INDICES of the next batch (map -> triplet ->
  // ix -> row id: two or three dependent loads) are fetched while the current batch is consumed, so a batch starts with
  // its row loads instead of its index chain.  Every load is unconditional on a clamped position (slots past the end
  // repeat the segment's last triplet and are not accumulated): same sums in the same order as before.
  constexpr int UT = 4;
  if (p0 < p1) {
    int ntt[UT], nrow[UT];
    auto req_idx = [&](int p) {
#pragma unroll
      for (int u = 0; u < UT; ++u) {
        const int pos = p + u < p1 ? p + u : p1 - 1;
        ntt[u] = map ? map[pos] : pos;
      }
#pragma unroll
      for (int u = 0; u < UT; ++u) nrow[u] = ix[ntt[u]];
    };
    req_idx(p0);
    for (int p = p0; p < p1; p += UT) {
      float4 a0[UT], a1[UT], b0[UT], b1[UT], x[UT];
#pragma unroll
      for (int u = 0; u < UT; ++u) {
        const int64_t tt = ntt[u];
        a0[u] = Ps[2 * tt];
        a1[u] = Ps[2 * tt + 1];
        if (TOR) {
          b0[u] = Pt[2 * tt];
          b1[u] = Pt[2 * tt + 1];
        }
        x[u] = X[(int64_t)nrow[u] * LPR + c];
      }
      if (p + UT < p1) req_idx(p + UT);
#pragma unroll
      for (int u = 0; u < UT; ++u) {
        float4 v = x[u];
        v.x *= dot8(ws_w[0], a0[u], a1[u]);
        v.y *= dot8(ws_w[1], a0[u], a1[u]);
        v.z *= dot8(ws_w[2], a0[u], a1[u]);
        v.w *= dot8(ws_w[3], a0[u], a1[u]);
        if (TOR) {
          v.x *= dot8(wt_w[0], b0[u], b1[u]);
          v.y *= dot8(wt_w[1], b0[u], b1[u]);
          v.z *= dot8(wt_w[2], b0[u], b1[u]);
          v.w *= dot8(wt_w[3], b0[u], b1[u]);
        }
        if (p + u < p1) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
      }
    }
  }
  out[(int64_t)w * LPR + c] = acc;
}

// ------------------------------------------------------------------------------------------------
// k_trip_bwd: gradients w.r.t. the projected bases and the second Linears, edge-segment mapping
// (every triplet of segment e shares G[e]):
//   gws[t,c] = G[e,c] X[kj[t],c] wt(t,c)      gwt[t,c] = G[e,c] X[kj[t],c] ws(t,c)
//   gPs[t,b] = sum_c gws[t,c] W2s[c,b]        (reduced over the LPR lanes of the worker by xor-shuffles)
//   gW2s[c,b] = sum_t gws[t,c] Ps[t,b]        (registers -> block partial -> k_reduce_partials)
// Persistent grid: each worker strides over segments so the gW2 accumulators live in registers.
// part[blockIdx][2][C][PB]
// ------------------------------------------------------------------------------------------------
template <int LPR>
__device__ __forceinline__ float worker_sum(float v) {
  if (LPR == 16) {
    // a 16-lane worker is exactly one DPP row: all-reduce by row rotations 8, 4, 2, 1 — four VALU adds with a DPP
    // operand instead of four ds_bpermute round trips through the LDS pipe (64 of them per triplet before)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xF, 0xF, false));
    return v;
  }
#pragma unroll
  for (int o = LPR >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int LPR, bool TOR>
__global__ void __launch_bounds__(256) k_trip_bwd(const float4* __restrict__ G, const float4* __restrict__ X,
                                                   const int* __restrict__ kj, const float4* __restrict__ Ps,
                                                   const float4* __restrict__ Pt, const float* __restrict__ W2s,
                                                   const float* __restrict__ W2t, const int* __restrict__ tptr,
                                                   int E, float* __restrict__ gPs, float* __restrict__ gPt,
                                                   float* __restrict__ part) {
  constexpr int WPB = 256 / LPR;                  // workers per block
  __shared__ float sred[256 * 8];                 // cross-worker reduction of the gW2 accumulators
  const int wib = threadIdx.x / LPR;
  const int c = threadIdx.x % LPR;
  float ws_w[4][PB], wt_w[4][PB];
  float gs[4][PB], gt[4][PB];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      ws_w[q][b] = W2s[(4 * c + q) * PB + b];
      wt_w[q][b] = TOR ? W2t[(4 * c + q) * PB + b] : 0.f;
      gs[q][b] = 0.f;
      gt[q][b] = 0.f;
    }
  for (int e = blockIdx.x * WPB + wib; e < E; e += gridDim.x * WPB) {
    const float4 g4 = G[(int64_t)e * LPR + c];
    const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
    // (r04, measured and not kept: requesting the operands of triplet t + 1 — and the row index of t + 2 — before the
    // arithmetic of t: 42.3 vs 39.5 us at 1.0e5 triplets.  The loop is not a bare latency chain: ~270 VALU / DPP
    // instructions per triplet step keep the two resident waves of a SIMD busy while the other one waits.)
    for (int t = tptr[e], t1 = tptr[e + 1]; t < t1; ++t) {
      const float4 a0 = Ps[2 * (int64_t)t], a1 = Ps[2 * (int64_t)t + 1];
      const float pa[PB] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
      if (TOR) { b0 = Pt[2 * (int64_t)t]; b1 = Pt[2 * (int64_t)t + 1]; }
      const float pb[PB] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      const float4 x4 = X[(int64_t)kj[t] * LPR + c];
      const float xx[4] = {x4.x, x4.y, x4.z, x4.w};
      float gws[4], gwt[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float ws = dot8(ws_w[q], a0, a1);
        const float gx = gg[q] * xx[q];
        if (TOR) {
          const float wt = dot8(wt_w[q], b0, b1);
          gws[q] = gx * wt;
          gwt[q] = gx * ws;
        } else {
          gws[q] = gx;
          gwt[q] = 0.f;
        }
      }
      float ps[PB], pt[PB];
#pragma unroll
      for (int b = 0; b < PB; ++b) {
        float s = gws[0] * ws_w[0][b];
        s = fmaf(gws[1], ws_w[1][b], s); s = fmaf(gws[2], ws_w[2][b], s); s = fmaf(gws[3], ws_w[3][b], s);
        ps[b] = worker_sum<LPR>(s);
        if (TOR) {
          float u = gwt[0] * wt_w[0][b];
          u = fmaf(gwt[1], wt_w[1][b], u); u = fmaf(gwt[2], wt_w[2][b], u); u = fmaf(gwt[3], wt_w[3][b], u);
          pt[b] = worker_sum<LPR>(u);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          gs[q][b] = fmaf(gws[q], pa[b], gs[q][b]);
          if (TOR) gt[q][b] = fmaf(gwt[q], pb[b], gt[q][b]);
        }
      }
      if (c == 0) {
        float4* o = (float4*)(gPs + (int64_t)t * PB);
        o[0] = make_float4(ps[0], ps[1], ps[2], ps[3]);
        o[1] = make_float4(ps[4], ps[5], ps[6], ps[7]);
      }
      if (TOR && c == (LPR > 1 ? 1 : 0)) {
        float4* o = (float4*)(gPt + (int64_t)t * PB);
        o[0] = make_float4(pt[0], pt[1], pt[2], pt[3]);
        o[1] = make_float4(pt[4], pt[5], pt[6], pt[7]);
      }
    }
  }
  // block reduction over the WPB workers: lanes with equal c hold partial sums of the same (channel, b)
  float* outp = part + (int64_t)blockIdx.x * (2 * 4 * LPR * PB);
#pragma unroll
  for (int br = 0; br < (TOR ? 2 : 1); ++br) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __syncthreads();
#pragma unroll
      for (int b = 0; b < PB; ++b) sred[threadIdx.x * PB + b] = br == 0 ? gs[q][b] : gt[q][b];
      __syncthreads();
      // LPR*PB outputs for this (br, q); thread j sums column j over the WPB workers (ascending)
      for (int j = threadIdx.x; j < LPR * PB; j += 256) {
        const int cc = j / PB, b = j - cc * PB;
        float s = 0.f;
        for (int wk = 0; wk < WPB; ++wk) s += sred[(wk * LPR + cc) * PB + b];
        outp[(br * 4 * LPR + (4 * cc + q)) * PB + b] = s;
      }
    }
  }
}

// (r04, measured and not kept: the same interaction grouped by the SOURCE NODE of the output edges — one workgroup per node
// stages the node's in-edge rows of X in LDS once and serves all its out-edges, the backward as ONE pass producing gX, gPs,
// gPt and the W2 partials.  HBM traffic fell to the algorithmic figure (X, Ps, Pt read once), the time did not: forward
// 23.4 vs 18.8 us at 32 QM9-like molecules, 74.9 vs 70.7 us at 32 OC20-like systems, 195 vs 177 us at 512 molecules; the
// fused backward 96 vs 18.9 + 39.6 us (256 VGPRs, one workgroup per CU).  Both forms are latency-, not traffic-bound at two
// waves per SIMD; the node form adds a staging barrier and a longer pointer chain (sptr -> sperm -> tptr -> P) per
// workgroup.  Numbers: profiles/r04_triplet_and_basis_routes_timing.jsonl; code: commit 05fde4b.)

// out[j] = sum_{k < nparts} part[k*stride + j], j < n.  Block = 32 outputs x 8 partial-lanes; fixed
// summation tree => deterministic.
__global__ void __launch_bounds__(256) k_reduce_partials(const float* __restrict__ part, int nparts, int stride,
                                                         int n, float* __restrict__ out) {
  __shared__ float red[8][33];
  const int jj = threadIdx.x & 31, kg = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + jj;
  float s = 0.f;
  if (j < n)
    for (int k = kg; k < nparts; k += 8) s += part[(int64_t)k * stride + j];
  red[kg][jj] = s;
  __syncthreads();
  if (kg == 0 && j < n)
    out[j] = ((red[0][jj] + red[1][jj]) + (red[2][jj] + red[3][jj])) +
             ((red[4][jj] + red[5][jj]) + (red[6][jj] + red[7][jj]));
}

// ------------------------------------------------------------------------------------------------
// k_basis_wgrad: gWs[o*KS + (l*nr+n)] = sum_t gPs[lyr(o)][t][b(o)] * Y_l0(t) bes[kj[t], l, n]     (and gWt)
// Block = 384 threads; thread j < KS + KT owns one basis column k and its PO accumulators.  Per chunk of
// TC triplets: threads 0..TC-1 evaluate the harmonics and stage the radial row in LDS, everybody stages
// gP, then every column thread runs TC x PO FMAs.  part[blockIdx][(KS+KT)*PO] -> k_reduce_partials.
// ------------------------------------------------------------------------------------------------
#define WG_TPB 384
#define WG_TC 64
template <int NS, bool TOR>
__global__ void __launch_bounds__(WG_TPB) k_basis_wgrad(const float* __restrict__ bes, const int* __restrict__ kj,
                                                         const float* __restrict__ angle,
                                                         const float* __restrict__ torsion, int T, int nr,
                                                         const float* __restrict__ pref,
                                                         const float* __restrict__ gPs,
                                                         const float* __restrict__ gPt, int L,
                                                         float* __restrict__ part, const int* __restrict__ cnt) {
  const int Tl = (cnt && *cnt < T) ? *cnt : T;     // static-shape batch: rows in [Tl, T) are padding
  constexpr int H2 = TOR ? NS * NS : NS;
  constexpr int YS = H2 | 1;                       // odd row strides: conflict-free column access
  extern __shared__ float smem[];
  const int KB = NS * nr;                          // radial row width
  const int BS = KB | 1;
  float* sY = smem;                                // [TC][YS]
  float* sB = sY + WG_TC * YS;                     // [TC][BS]
  float* sGs = sB + WG_TC * BS;                    // [TC][PO]
  float* sGt = sGs + WG_TC * PO;                   // [TC][PO]
  __shared__ float sPref[NS_MAX * NS_MAX];
  for (int q = threadIdx.x; q < NS_MAX * NS_MAX; q += WG_TPB) sPref[q] = pref[q];
  const int KS = NS * nr, KT = TOR ? NS * NS * nr : 0;
  const int j = threadIdx.x;
  const bool is_s = j < KS, is_t = !is_s && j < KS + KT;
  // column decode
  int yh = 0, bo = 0;                              // harmonic index into sY row, radial index into sB row
  if (is_s) {
    const int l = j / nr, n = j - l * nr;
    yh = TOR ? l * l : l;
    bo = l * nr + n;
  } else if (is_t) {
    const int k = j - KS;
    const int h = k / nr, n = k - h * nr;
    yh = h;
    bo = (h % NS) * nr + n;
  }
  float acc[PO];
#pragma unroll
  for (int o = 0; o < PO; ++o) acc[o] = 0.f;
  const int nchunks = (Tl + WG_TC - 1) / WG_TC;
  for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int t0 = ch * WG_TC;
    const int nt = (Tl - t0 < WG_TC) ? Tl - t0 : WG_TC;
    __syncthreads();
    if (j < nt) {
      float Y[H2];
      real_sph_harm<NS>(angle[t0 + j], TOR ? torsion[t0 + j] : 0.f, sPref, !TOR, Y);
#pragma unroll
      for (int h = 0; h < H2; ++h) sY[j * YS + h] = Y[h];
    }
    // radial rows: TC x KB gathered floats, spread over the block — EIGHT index -> value chains in flight per thread
    // (unconditional on clamped positions, masked on the store): one element per loop trip was 7 serial pairs of dependent
    // round trips per chunk, more than the chunk's arithmetic (r04: 100 -> 77 us at T = 1.0e5 together with two blocks per CU)
    for (int q0 = j; q0 < nt * KB; q0 += 8 * WG_TPB) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = q0 + u * WG_TPB, qc = q < nt * KB ? q : 0;
        const int r = qc / KB, k = qc - r * KB;
        v[u] = bes[(int64_t)kj[t0 + r] * KB + k];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = q0 + u * WG_TPB;
        if (q < nt * KB) {
          const int r = q / KB, k = q - r * KB;
          sB[r * BS + k] = v[u];
        }
      }
    }
    for (int q0 = j; q0 < nt * PO; q0 += 6 * WG_TPB) {
      float vs[6], vt[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int q = q0 + u * WG_TPB, qc = q < nt * PO ? q : 0;
        const int r = qc / PO, o = qc - r * PO;
        const int l = o / PB, b = o - l * PB;
        const bool live = l < L;
        const int64_t src = ((int64_t)(live ? l : 0) * T + t0 + r) * PB + b;
        vs[u] = gPs[src];
        vt[u] = TOR ? gPt[src] : 0.f;
        if (!live) vs[u] = vt[u] = 0.f;
      }
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int q = q0 + u * WG_TPB;
        if (q < nt * PO) {
          sGs[q] = vs[u];
          if (TOR) sGt[q] = vt[u];
        }
      }
    }
    __syncthreads();
    if (is_s || is_t) {
      const float* __restrict__ sG = is_s ? sGs : sGt;
#pragma unroll 4
      for (int r = 0; r < nt; ++r) {
        const float bv = sY[r * YS + yh] * sB[r * BS + bo];
        const float4* g4 = (const float4*)(sG + r * PO);
#pragma unroll
        for (int o4 = 0; o4 < PO / 4; ++o4) {
          const float4 g = g4[o4];
          acc[4 * o4 + 0] = fmaf(bv, g.x, acc[4 * o4 + 0]);
          acc[4 * o4 + 1] = fmaf(bv, g.y, acc[4 * o4 + 1]);
          acc[4 * o4 + 2] = fmaf(bv, g.z, acc[4 * o4 + 2]);
          acc[4 * o4 + 3] = fmaf(bv, g.w, acc[4 * o4 + 3]);
        }
      }
    }
  }
  if (is_s || is_t) {
    // OUTPUT-major partials, [32][KS] then [32][KT]: row l*8 + b is the gradient of weight row b of layer l's first
    // basis Linear ([basis_emb, K] in the reference's layout) — contiguous, so no transposing copy per layer and pass
    // (8 framework copies per step before); consecutive threads write consecutive floats
    float* o = part + (int64_t)blockIdx.x * (KS + KT) * PO + (is_s ? j : (int64_t)KS * PO + (j - KS));
    const int K = is_s ? KS : KT;
#pragma unroll
    for (int q = 0; q < PO; ++q) o[(int64_t)q * K] = acc[q];
  }
}

// ================================================================================================
// C ABI
// ================================================================================================
// Ws[k][l*8 + b] = lin_sbf1_l.weight[b][k] (zero for b >= basis size, l >= L): the stacked, transposed, zero-padded weight
// layout k_basis_project reads with scalar loads — one launch for both tables instead of the framework's cat +
// transposing copy per table and forward (4 launches per step)
struct BasisStackDesc {
  const float* Ws[PO / PB];
  const float* Wt[PO / PB];
  int bs_s[PO / PB], bs_t[PO / PB];
};
__global__ void __launch_bounds__(256) k_basis_stack(BasisStackDesc d, int L, int KS, int KT, float* __restrict__ outS,
                                                      float* __restrict__ outT) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int nS = KS * PO;
  if (i < nS) {
    const int k = i / PO, o = i - k * PO, l = o / PB, b = o - l * PB;
    outS[i] = (l < L && b < d.bs_s[l]) ? d.Ws[l][(int64_t)b * KS + k] : 0.f;
  } else if (i < nS + KT * PO) {
    const int q = i - nS;
    const int k = q / PO, o = q - k * PO, l = o / PB, b = o - l * PB;
    outT[q] = (l < L && b < d.bs_t[l]) ? d.Wt[l][(int64_t)b * KT + k] : 0.f;
  }
}

extern "C" {

// outS float[KS * 32], outT float[KT * 32] (outT / Wt NULL: no torsion) from Ws[l] [bs_s[l], KS], Wt[l] [bs_t[l], KT], l < L <= 4
int dig3d_basis_stack(int L, const void* const* Ws, const void* const* Wt, const int* bs_s, const int* bs_t, int KS, int KT,
                      float* outS, float* outT, void* stream) {
  DIG3D_ENTER();
  if (L < 1 || L > PO / PB || !Ws || !bs_s || KS < 1 || !outS) return DIG3D_ERR_ARG;
  const bool tor = Wt != nullptr && outT != nullptr;
  if (tor && (!bs_t || KT < 1)) return DIG3D_ERR_ARG;
  BasisStackDesc d;
  for (int l = 0; l < PO / PB; ++l) {
    d.Ws[l] = l < L ? (const float*)Ws[l] : nullptr;
    d.Wt[l] = (tor && l < L) ? (const float*)Wt[l] : nullptr;
    d.bs_s[l] = l < L ? bs_s[l] : 0;
    d.bs_t[l] = (tor && l < L) ? bs_t[l] : 0;
    if (l < L && (!d.Ws[l] || d.bs_s[l] < 1 || d.bs_s[l] > PB || (tor && (!d.Wt[l] || d.bs_t[l] < 1 || d.bs_t[l] > PB))))
      return DIG3D_ERR_ARG;
  }
  const int n = (KS + (tor ? KT : 0)) * PO;
  hipLaunchKernelGGL(k_basis_stack, dim3(dig3d_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, d, L, KS, tor ? KT : 0,
                     outS, outT);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}


// P[l][t][0..7] for l < L (L <= 4) from the stacked, transposed, zero-padded first basis Linears:
//   Ws[ns*nr][32], Wt[ns*ns*nr][32] (Wt/torsion/Pt NULL => DimeNet++: no torsion branch).
int dig3d_basis_project(const float* bes, const int* kj, const float* angle, const float* torsion, int T,
                        int ns, int nr, const float* pref, const float* Ws, const float* Wt, int L, float* Ps,
                        float* Pt, const int* cnt, int route, void* stream) {
  DIG3D_ENTER();
  if (T <= 0) return DIG3D_OK;
  if (L < 1 || L > PO / PB || ns < 1 || ns > NS_MAX || nr < 1 || !bes || !kj || !angle || !Ws || !Ps)
    return DIG3D_ERR_ARG;
  const bool tor = torsion != nullptr;
  if (tor && (!Wt || !Pt)) return DIG3D_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  // matrix-core route (basis_mfma.hip) for the shapes it covers (route 0: eight waves x 32 triplets, route 2: the r04 form,
  // four waves x 64); route 1 keeps the VALU kernel (tests compare them)
  if (route != 1 && basis_project_mfma(bes, kj, angle, torsion, T, ns, nr, pref, Ws, Wt, L, Ps, Pt, cnt, route == 2, st) == 0) {
    DIG3D_CHECK_LAUNCH();
    return DIG3D_OK;
  }
  dim3 grid(dig3d_blocks(T, 128)), block(128);
  const size_t shm = sizeof(float) * 128 * (size_t)((ns * nr) | 1);
  if (shm > 60000) return DIG3D_ERR_ARG;
#define BP_CASE(NS)                                                                                          \
  case NS:                                                                                                   \
    if (tor)                                                                                                 \
      hipLaunchKernelGGL((k_basis_project<NS, true>), grid, block, shm, st, bes, kj, angle, torsion, T, nr,  \
                         pref, Ws, Wt, L, Ps, Pt, cnt);                                                      \
    else                                                                                                     \
      hipLaunchKernelGGL((k_basis_project<NS, false>), grid, block, shm, st, bes, kj, angle, torsion, T, nr, \
                         pref, Ws, Wt, L, Ps, Pt, cnt);                                                      \
    break;
  switch (ns) {
    BP_CASE(1) BP_CASE(2) BP_CASE(3) BP_CASE(4) BP_CASE(5) BP_CASE(6) BP_CASE(7) BP_CASE(8)
    default: return DIG3D_ERR_ARG;
  }
#undef BP_CASE
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// gWs[32][ns*nr], gWt[32][ns*ns*nr] (row l*8 + b = weight row b of layer l) from gPs/gPt[L][T][8].  part: float[nblocks * (KS+KT) * 32] scratch,
// nblocks = dig3d_basis_wgrad_blocks(T).
// worker blocks of the weight-gradient kernels.  The matrix-core kernel keeps 40 KB of LDS and <= 168 registers per block:
// three blocks (12 waves) fit a CU, one generates operands while the others multiply.  Every block costs a fixed ~43 KB
// partial and a prologue, so three per CU only pay with >= 6 tiles per block (r06, same box, 256 / 512 / 768 / 1024 blocks:
// T = 1.1e5 83.4 / 66.3 / 68.4 / 78.9 us; T = 5.9e5 394 / 253 / 217 / 275 us; T = 1.6e6 1058 / 656 / 534 / 656 us).  The VALU
// kernel was insensitive (512 / 1024 blocks -> 8.19-8.36 ms per config-4 step).
int dig3d_basis_wgrad_blocks(int T) {
  const int nchunks = (T + WG_TC - 1) / WG_TC, cus = dig3d_num_cus();
  const int cap = nchunks >= 18 * cus ? 3 * cus : 2 * cus;
  const int nb = nchunks < cap ? nchunks : cap;
  return nb < 1 ? 1 : nb;
}

int dig3d_basis_wgrad(const float* bes, const int* kj, const float* angle, const float* torsion, int T, int ns,
                      int nr, const float* pref, const float* gPs, const float* gPt, int L, float* part,
                      float* gWs, float* gWt, const int* cnt, int reduce_now, int route, void* stream) {
  DIG3D_ENTER();
  if (L < 1 || L > PO / PB || ns < 1 || ns > NS_MAX || nr < 1 || !gWs || !part) return DIG3D_ERR_ARG;
  const bool tor = torsion != nullptr;
  if (tor && (!gPt || !gWt)) return DIG3D_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int KS = ns * nr, KT = tor ? ns * ns * nr : 0;
  if (KS + KT > WG_TPB) return DIG3D_ERR_ARG;
  if (T <= 0) {
    // (an EMPTY torsion array has a null data pointer: with no triplets the torsion branch is recognised by its gradient
    // buffer — found by the poisoned-allocation sweep of r05: lin_t1.weight.grad of a batch of isolated atoms was never written)
    if (dig3d_zero_async(gWs, sizeof(float) * KS * PO, st) != hipSuccess) return DIG3D_ERR_LAUNCH;
    if (gWt && dig3d_zero_async(gWt, sizeof(float) * (size_t)ns * ns * nr * PO, st) != hipSuccess) return DIG3D_ERR_LAUNCH;
    return DIG3D_OK;
  }
  const int nb = dig3d_basis_wgrad_blocks(T);
  const int H2 = tor ? ns * ns : ns;
  const size_t shm = sizeof(float) * ((size_t)WG_TC * (H2 | 1) + (size_t)WG_TC * ((ns * nr) | 1) + 2 * WG_TC * PO);
  const bool mfma = route != 1 &&
                    basis_wgrad_mfma(bes, kj, angle, torsion, T, ns, nr, pref, gPs, gPt, L, part, cnt, nb, st) == 0;
  if (!mfma) {
#define WG_CASE(NS)                                                                                          \
  case NS:                                                                                                   \
    if (tor)                                                                                                 \
      hipLaunchKernelGGL((k_basis_wgrad<NS, true>), dim3(nb), dim3(WG_TPB), shm, st, bes, kj, angle, torsion, \
                         T, nr, pref, gPs, gPt, L, part, cnt);                                               \
    else                                                                                                     \
      hipLaunchKernelGGL((k_basis_wgrad<NS, false>), dim3(nb), dim3(WG_TPB), shm, st, bes, kj, angle,        \
                         torsion, T, nr, pref, gPs, gPt, L, part, cnt);                                      \
    break;
  switch (ns) {
    WG_CASE(1) WG_CASE(2) WG_CASE(3) WG_CASE(4) WG_CASE(5) WG_CASE(6) WG_CASE(7) WG_CASE(8)
    default: return DIG3D_ERR_ARG;
  }
#undef WG_CASE
  }
  DIG3D_CHECK_LAUNCH();
  const int n = (KS + KT) * PO;
  // columns [0, KS) -> gWs, [KS, KS+KT) -> gWt: two reductions over the same partial buffer (row stride n); with
  // reduce_now == 0 the caller reduces later, together with every other layer's partials (dig3d_reduce_many)
  if (!reduce_now) return DIG3D_OK;
  hipLaunchKernelGGL(k_reduce_partials, dim3(dig3d_blocks(KS * PO, 32)), dim3(256), 0, st, part, nb, n, KS * PO, gWs);
  if (tor)
    hipLaunchKernelGGL(k_reduce_partials, dim3(dig3d_blocks(KT * PO, 32)), dim3(256), 0, st, part + KS * PO, nb, n,
                       KT * PO, gWt);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// the ONE place that decides which kernel dig3d_triplet_fwd launches (the entry point and dig3d_triplet_fwd_kernel, the
// name the measurement tools ask for, both read it)
// route: 0 = the measured best form for the size, 1 = lane groups (k_trip_fwd), 2 = a wave per segment with scalar-loaded
// operands (k_trip_fwd_w), 3 = a wave per segment that walks the index chain once (k_trip_fwd_l).  -> 0 / 1 / 2: lane
// groups / k_trip_fwd_w / k_trip_fwd_l.
// Same box, C = 64, torsion (profiles/r06_triplet_lds_form_timing.jsonl): 7.8k segments / 1.0e5 triplets: 12.3 (w) vs 12.7 (l)
// us forward stand-alone, 13.9 vs 13.1 through the transposed CSR; 36.7k / 5.9e5: 51.5 vs 37.6 and 66.3 vs 38.0 (lane groups
// 65.9); 1.2e5 / 1.6e6: 126.7 vs 96.5.  In the replayed step the l form wins at the small size too (config 2: 1.374 vs 1.390
// ms with it in every launch, config 3: 3.662 vs 3.682), so it is the form for C = 64 / 128; C = 256 (not measured with it:
// 140 VGPRs) keeps the r05 rule.  The three forms are bit-identical, so the switch does not show in the results.
#define kTripBwdLds 1
static int trip_fwd_form(int S, int C, bool transposed, int route) {
  if (route == 1 || !(C == 64 || C == 128 || C == 256)) return 0;
  if (route == 2) return 1;
  if (route == 3 || C <= 128) return 2;
  return (transposed && S >= 24576) ? 0 : 1;
}

// name of the kernel dig3d_triplet_fwd launches for these arguments, as rocprofv3 prints it ("k_trip_fwd_w<1, true, false>"):
// written to name[cap], returns its length (or -1).  tools/roofline_kernels.py labels its roofline line and finds the
// kernel's PMC rows with it.
int dig3d_triplet_fwd_kernel(int S, int C, int torsion, int transposed, int route, char* name, int cap) {
  if (!name || cap < 32) return DIG3D_ERR_ARG;
  if (C != 16 && C != 32 && C != 64 && C != 128 && C != 256) return DIG3D_ERR_ARG;
  const char* tf = torsion ? "true" : "false";
  const int form = trip_fwd_form(S, C, transposed != 0, route);
  if (form) return snprintf(name, cap, "k_trip_fwd_%c<%d, %s, false>", form == 2 ? 'l' : 'w', C / 64, tf);   // <CPL, TOR, ADD>
  return snprintf(name, cap, "k_trip_fwd<%d, %s>", C / 4, tf);
}

// out[S,C] = sum over (kptr,map) segments of X[ix[t]] * (W2s Ps[t]) * (W2t Pt[t]);  C in {16,32,64,128,256}.
// Ps/Pt: [T,8]; W2s/W2t: [C,8] (lin_sbf2 / lin_t2 weights, zero padded to 8 columns); Pt/W2t NULL => no
// torsion factor.
// the lane-group kernels have no `add` operand: o[q] += a[q] for q < n (n read on the device when n_dev != NULL)
static __global__ void __launch_bounds__(256) k_trip_add_rows(float* __restrict__ o, const float* __restrict__ a, int64_t n,
                                                              const int* __restrict__ n_dev, int per) {
  if (n_dev) n = (int64_t)*n_dev * per;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (int64_t)gridDim.x * 256) o[q] += a[q];
}

int dig3d_triplet_fwd_add(const float* X, const int* ix, const float* Ps, const float* Pt, const float* W2s,
                          const float* W2t, const int* kptr, const int* map, int S, int C, float* out, const float* add,
                          int route, void* stream);
int dig3d_triplet_fwd(const float* X, const int* ix, const float* Ps, const float* Pt, const float* W2s,
                      const float* W2t, const int* kptr, const int* map, int S, int C, float* out, int route,
                      void* stream) {
  return dig3d_triplet_fwd_add(X, ix, Ps, Pt, W2s, W2t, kptr, map, S, C, out, nullptr, route, stream);
}

// the same with out += add [S, C] (NULL: none) inside the launch: the final pass of energy_and_force, where a second
// gradient reaches the same tensor (dig_amd/diffops.py: _TripT / _TripBwd2)
int dig3d_triplet_fwd_add(const float* X, const int* ix, const float* Ps, const float* Pt, const float* W2s,
                          const float* W2t, const int* kptr, const int* map, int S, int C, float* out, const float* add,
                          int route, void* stream) {
  DIG3D_ENTER();
  if (S < 0 || !X || !ix || !Ps || !W2s || !kptr || !out || ((uintptr_t)add & 15)) return DIG3D_ERR_ARG;
  if (S == 0) return DIG3D_OK;
  if ((((uintptr_t)X | (uintptr_t)Ps | (uintptr_t)Pt | (uintptr_t)out | (uintptr_t)W2s | (uintptr_t)W2t) & 15) != 0)
    return DIG3D_ERR_ARG;
  const bool tor = Pt != nullptr;
  if (tor && !W2t) return DIG3D_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  // C = 64 / 128 / 256: a wave per segment, a lane per channel (triplet_wave.hip; which of its two forms: trip_fwd_form);
  // route 1 and the narrow widths: 16 ... 64 lanes per segment, four channels per lane (below)
  const int form = trip_fwd_form(S, C, map != nullptr, route);
  if ((form == 2 && trip_fwd_lds(X, ix, Ps, Pt, W2s, W2t, kptr, map, S, C, out, add, st) == 0) ||
      (form == 1 && trip_fwd_wave(X, ix, Ps, Pt, W2s, W2t, kptr, map, S, C, out, add, st) == 0)) {
    DIG3D_CHECK_LAUNCH();
    return DIG3D_OK;
  }
#define TF(LPR)                                                                                               \
  do {                                                                                                        \
    dim3 grid(dig3d_blocks((int64_t)S * LPR, 256));                                                           \
    if (tor)                                                                                                  \
      hipLaunchKernelGGL((k_trip_fwd<LPR, true>), grid, dim3(256), 0, st, (const float4*)X, ix,               \
                         (const float4*)Ps, (const float4*)Pt, W2s, W2t, kptr, map, S, (float4*)out);         \
    else                                                                                                      \
      hipLaunchKernelGGL((k_trip_fwd<LPR, false>), grid, dim3(256), 0, st, (const float4*)X, ix,              \
                         (const float4*)Ps, (const float4*)Pt, W2s, W2t, kptr, map, S, (float4*)out);         \
  } while (0)
  switch (C) {
    case 16: TF(4); break;
    case 32: TF(8); break;
    case 64: TF(16); break;
    case 128: TF(32); break;
    case 256: TF(64); break;
    default: return DIG3D_ERR_ARG;
  }
#undef TF
  if (add) {
    const int64_t n = (int64_t)S * C;
    int nb = dig3d_blocks(n, 1024);
    hipLaunchKernelGGL(k_trip_add_rows, dim3(nb > 2048 ? 2048 : nb), dim3(256), 0, st, out, add, n, (const int*)nullptr, 0);
  }
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// The kernel is a chain of dependent loads per triplet (kj[t] -> X row), one triplet at a time per worker: its time is
// (triplets per worker) x latency, so more, shorter workers win until the chip's wave slots are full (register-heavy:
// ~3 blocks per CU).  Same-box A/B on config 4: cap 256 / 768 / 1536 -> 8.26 / 8.13 / 8.07 ms per step.
// DIG3D_TRIP_BWD_BLOCKS overrides the cap (read once).
#define kTripBwdCap (8 * dig3d_num_cus())       // worker blocks of k_trip_bwd (each writes one partial of the W2 gradients)
int dig3d_triplet_bwd_blocks(int E, int C, int route) {
  if (route != 1) {
    const int nbw = trip_bwd_wave_blocks(E, C);
    if (nbw > 0) return nbw;
  }
  int wpb = 256 / (C / 4);
  int nb = (E + wpb - 1) / wpb;
  if (nb > kTripBwdCap) nb = kTripBwdCap;
  return nb < 1 ? 1 : nb;
}

// gPs/gPt [T,8], gW2s/gW2t [C,8].  part: float[nblocks * 2*C*8], nblocks = dig3d_triplet_bwd_blocks(E, C).
int dig3d_triplet_bwd_add(const float* G, const float* X, const int* kj, const float* Ps, const float* Pt,
                          const float* W2s, const float* W2t, const int* tptr, int E, int C, float* gPs, float* gPt,
                          float* part, float* gW2s, float* gW2t, int reduce_now, int route, const float* gPs_add,
                          const float* gPt_add, void* stream);
int dig3d_triplet_bwd(const float* G, const float* X, const int* kj, const float* Ps, const float* Pt,
                      const float* W2s, const float* W2t, const int* tptr, int E, int C, float* gPs, float* gPt,
                      float* part, float* gW2s, float* gW2t, int reduce_now, int route, void* stream) {
  return dig3d_triplet_bwd_add(G, X, kj, Ps, Pt, W2s, W2t, tptr, E, C, gPs, gPt, part, gW2s, gW2t, reduce_now, route, nullptr,
                               nullptr, stream);
}

// the same with gPs += gPs_add, gPt += gPt_add ([T, 8]; NULL: none) inside the launch (see dig3d_triplet_fwd_add)
int dig3d_triplet_bwd_add(const float* G, const float* X, const int* kj, const float* Ps, const float* Pt,
                          const float* W2s, const float* W2t, const int* tptr, int E, int C, float* gPs, float* gPt,
                          float* part, float* gW2s, float* gW2t, int reduce_now, int route, const float* gPs_add,
                          const float* gPt_add, void* stream) {
  DIG3D_ENTER();
  if (E < 0 || !G || !X || !kj || !Ps || !W2s || !tptr || !gPs || !part || !gW2s) return DIG3D_ERR_ARG;
  if (gPt_add && (!gPs_add || !Pt)) return DIG3D_ERR_ARG;
  if ((((uintptr_t)G | (uintptr_t)X | (uintptr_t)Ps | (uintptr_t)Pt | (uintptr_t)W2s | (uintptr_t)W2t) & 15) != 0)
    return DIG3D_ERR_ARG;
  const bool tor = Pt != nullptr;
  if (tor && (!W2t || !gPt || !gW2t)) return DIG3D_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (E == 0) {
    if (dig3d_zero_async(gW2s, sizeof(float) * C * PB, st) != hipSuccess) return DIG3D_ERR_LAUNCH;
    if (tor && dig3d_zero_async(gW2t, sizeof(float) * C * PB, st) != hipSuccess) return DIG3D_ERR_LAUNCH;
    return DIG3D_OK;
  }
  const int nb = dig3d_triplet_bwd_blocks(E, C, route);
  // the wave-per-segment backward in its two forms (triplet_wave.hip): k_trip_bwd_l (the index chain walked once) for C = 64 /
  // 128 — at C = 256 its projected rows in vector registers do not fit next to 64 weights and 64 gradient sums per lane —
  // and k_trip_bwd_w otherwise; route 2 / 3 force one (tests compare them: bit-identical)
  const bool lds = route == 3 || (route == 0 && kTripBwdLds && C <= 128);
  const bool wave = route != 1 && trip_bwd_wave_blocks(E, C) > 0 &&
                    (lds ? trip_bwd_lds(G, X, kj, Ps, Pt, W2s, W2t, tptr, E, C, gPs, gPt, part, nb, gPs_add,
                                        (Pt != nullptr) ? gPt_add : nullptr, st)
                         : trip_bwd_wave(G, X, kj, Ps, Pt, W2s, W2t, tptr, E, C, gPs, gPt, part, nb, gPs_add,
                                         (Pt != nullptr) ? gPt_add : nullptr, st)) == 0;
#define TB(LPR)                                                                                               \
  do {                                                                                                        \
    if (tor)                                                                                                  \
      hipLaunchKernelGGL((k_trip_bwd<LPR, true>), dim3(nb), dim3(256), 0, st, (const float4*)G,               \
                         (const float4*)X, kj, (const float4*)Ps, (const float4*)Pt, W2s, W2t, tptr, E, gPs,  \
                         gPt, part);                                                                          \
    else                                                                                                      \
      hipLaunchKernelGGL((k_trip_bwd<LPR, false>), dim3(nb), dim3(256), 0, st, (const float4*)G,              \
                         (const float4*)X, kj, (const float4*)Ps, (const float4*)Pt, W2s, W2t, tptr, E, gPs,  \
                         gPt, part);                                                                          \
  } while (0)
  if (!wave) switch (C) {
    case 16: TB(4); break;
    case 32: TB(8); break;
    case 64: TB(16); break;
    case 128: TB(32); break;
    case 256: TB(64); break;
    default: return DIG3D_ERR_ARG;
  }
#undef TB
  if (!wave && gPs_add) {           // T = tptr[E] lives on the device
    hipLaunchKernelGGL(k_trip_add_rows, dim3(1024), dim3(256), 0, st, gPs, gPs_add, (int64_t)0, tptr + E, PB);
    if (tor && gPt_add) hipLaunchKernelGGL(k_trip_add_rows, dim3(1024), dim3(256), 0, st, gPt, gPt_add, (int64_t)0, tptr + E, PB);
  }
  DIG3D_CHECK_LAUNCH();
  const int n = 2 * C * PB;
  if (!reduce_now) return DIG3D_OK;      // partial rows of stride n: [0, C*PB) -> gW2s, [C*PB, 2*C*PB) -> gW2t
  hipLaunchKernelGGL(k_reduce_partials, dim3(dig3d_blocks(C * PB, 32)), dim3(256), 0, st, part, nb, n, C * PB, gW2s);
  if (tor)
    hipLaunchKernelGGL(k_reduce_partials, dim3(dig3d_blocks(C * PB, 32)), dim3(256), 0, st, part + C * PB, nb, n,
                       C * PB, gW2t);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

}  // extern "C"
