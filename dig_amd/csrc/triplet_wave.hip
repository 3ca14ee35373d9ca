// dig3d triplet interaction, WAVE-PER-SEGMENT form (round 5) — same contracts as k_trip_fwd / k_trip_bwd of triplet.hip
// (method/spherenet/spherenet.py:164-171, dimenetpp.py:147-150), for C in {64, 128, 256} channels.
//
// What the counters said about the 16-lanes-per-segment kernels at the reference's batch size (32 QM9-like molecules:
// 8.7k segments, 1.0e5 triplets; profiles/r05_stall_counters.json): k_trip_fwd<16> — 1 978 waves on 1 024 SIMDs, 0.9 waves
// resident per SIMD, 50 % of the wave cycles parked on memory, 33 % issue stalls of the dependent 8-term dot chains,
// 16 % issuing; k_trip_bwd<16> — 1.06 waves per SIMD, 44 % parked.  184 / 216 VGPRs (64 second-Linear weights per lane
// plus four triplets of operands in flight) cap them at two waves per SIMD, and a wave walks its four segments alone:
// the kernel's time IS one wave's latency chain.
//
// Here a wave owns ONE segment and a lane owns C / 64 channels:
//   * everything that is per triplet — its position in the CSR, the triplet id, the gathered row id, the two projected
//     basis rows P_s[t], P_t[t] (2 x 8 floats) — is WAVE-UNIFORM: scalar loads into SGPRs (s_load_dwordx8), the dot
//     products take them as the SGPR operand of v_fma_f32;
//   * per lane: 8 + 8 weights per channel (16 VGPRs at C = 64 instead of 64), ~48 VGPRs in all -> 8 waves per SIMD, and
//     4x the waves (8.7k): the latency of one wave's chain is hidden by seven others instead of by nothing;
//   * the only vector memory traffic is the gathered row X[row] (256 B per wave-instruction at C = 64) and the output row.
// Arithmetic per (triplet, channel) and the order of the sums over a segment's triplets are those of k_trip_fwd: results are
// bit-identical to it (tests/test_gpu_ops.py compares the two).
// (r05, measured and reverted: the forward proper with the gathered rows requested ONE BATCH AHEAD — straight-line loop body,
// slots past the end blended out arithmetically, unrolled by two so the row registers ping-pong; 46 VGPRs, bit-identical —
// 11.96 vs 12.27 us at 7.8k segments / 1.0e5 triplets, but 55.3 vs 51.1 us at 36.7k / 5.9e5 and 138.6 vs 129.5 at 1.2e5 /
// 1.6e6; in the step 1.705 vs 1.694 ms (config 2), 5.589 vs 5.545 (config 4), one box.  With eight waves per SIMD the other
// waves already cover a wave's row latency; the deeper pipeline only adds instructions and registers.)
// (also measured and not kept: a capped grid of 4 096 blocks whose waves stride over several segments, so the 4 KB of
// second-Linear weights per wave are loaded once per ~7 segments instead of once per segment: 52.7 vs 51.6 us at 36.7k
// segments, 130.8 vs 125.8 at 1.2e5 — the weights come from L1 / L2 anyway and one wave per segment balances the uneven
// segment lengths better.)
// (and: the two projected-basis rows of a triplet through VECTOR loads of a wave-uniform address instead of scalar loads — no
// SGPR limit, 4 or 8 triplets in flight at 96 / 167 VGPRs: 20.6 / 23.1 vs 12.3 us at 7.8k segments, 85 / 96 vs 51 at 36.7k,
// 222 / 260 vs 127 at 1.2e5.  The scalar path is the right one.  Counters at the config-4 size, `profiles/
// r05_stall_counters_spherenet_oc20.json`: 3.7 waves resident per SIMD, 62 % of the wave cycles parked on memory, 22 % issue
// stalls, 16 % issuing — a wave's life is its chain of dependent scalar + row loads, whatever is done around it.)
// (r06, a different DECOMPOSITION, measured and not kept: node-centric — one workgroup per node j stages the <= 33 contiguous
// rows X[rowptr[j] ..) of its incoming edges in LDS once and serves all outgoing edges e = (j -> i) of j, a wave per edge; no
// index is read per triplet (triplet n of e is t = tptr[e] + n, its row the n-th incoming edge of j not coming from i), the
// projected rows stay scalar loads of consecutive addresses; transposed direction likewise with the G rows of the outgoing
// edges staged.  Bit-identical to this kernel in both directions at all three sizes, L2 -> CU traffic 5x lower (weights once
// per ~3 edges, rows from LDS) — and slower everywhere: forward 29.9 vs 11.9 us at 7.8k edges / 1.0e5 triplets, 94.7 vs 51.8 at
// 36.7k / 5.9e5, 149.5 vs 127.6 at 1.2e5 / 1.6e6; transposed 47.8 vs 13.7, 135.6 vs 66.6, 193.5 vs 169.6
// (profiles/r06_triplet_node_centric_timing.jsonl; source kept as docs/history/r06_triplet_node.hip.txt).  A wave that walks ~3
// edges pays three serial prologues (edge id -> target, triplet pointer -> first basis rows) where three one-edge waves pay
// them side by side: the kernel is bound by the per-wave chain of scalar loads, not by the gathered rows the staging removes.)
//
// Backward (k_trip_bwd_w): per triplet the lane forms gws = g x wt, gwt = g x ws for its channels; the 16 channel sums
// gP_s[t][0..7], gP_t[t][0..7] are reduced over the wave by a 4-step halving butterfly inside each 16-lane row (DPP
// partners l^8, l^7, l^3, l^1: row_ror:8, row_half_mirror, quad_perm[3,2,1,0], quad_perm[1,0,3,2] — 16 -> 8 -> 4 -> 2 -> 1
// live values per lane, lane l ends with sum number l % 16 of its row) and two cross-row exchanges; lanes 0..15 store the
// 64 bytes of the two gradient rows.  The second-Linear weight gradients accumulate in 16 registers per channel and leave
// through a block partial (dig3d_reduce_many sums them), exactly like k_trip_bwd.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "triplet_wave.hip uses v_permlane16_swap / v_permlane32_swap: build with --offload-arch=gfx950 (dig_amd/build.py)"
#endif
#include "common.h"

#define PB 8

namespace {

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// dot of a lane's 8 weights with a wave-uniform basis row, in k_trip_fwd's order (w0 a0, then fma chain)
__device__ __forceinline__ float dot8u(const float* __restrict__ w, const float* __restrict__ a) {
  float s = w[0] * a[0];
#pragma unroll
  for (int b = 1; b < PB; ++b) s = fmaf(w[b], a[b], s);
  return s;
}

template <int CPL>
struct Row {
  float v[CPL];
};
template <int CPL>
__device__ __forceinline__ Row<CPL> load_row(const float* __restrict__ X, int64_t off) {
  Row<CPL> r;
  if (CPL == 4) {
    const float4 q = *(const float4*)(X + off);
    r.v[0] = q.x; r.v[1] = q.y; r.v[2 % CPL] = q.z; r.v[3 % CPL] = q.w;
  } else if (CPL == 2) {
    const float2 q = *(const float2*)(X + off);
    r.v[0] = q.x; r.v[1 % CPL] = q.y;
  } else {
    r.v[0] = X[off];
  }
  return r;
}
template <int CPL>
__device__ __forceinline__ void store_row(float* __restrict__ X, int64_t off, const Row<CPL>& r) {
  if (CPL == 4) *(float4*)(X + off) = make_float4(r.v[0], r.v[1 % CPL], r.v[2 % CPL], r.v[3 % CPL]);
  else if (CPL == 2) *(float2*)(X + off) = make_float2(r.v[0], r.v[1 % CPL]);
  else X[off] = r.v[0];
}

// ------------------------------------------------------------------------------------------------------------------
// forward: out[s, c] = sum_{p in [kptr[s], kptr[s+1])} X[ix[t], c] * (W2s[c,:] . Ps[t,:]) * (W2t[c,:] . Pt[t,:]),  t = map ? map[p] : p
// ------------------------------------------------------------------------------------------------------------------
// ADD: out += add[s, c] — the OTHER gradient that reaches the same tensor in the final pass of energy_and_force (the sum then
// happens here, as one rounded addition like the framework's, instead of in an addition launch per layer and operand)
template <int CPL, bool TOR, bool ADD = false>
__global__ void __launch_bounds__(256) k_trip_fwd_w(const float* __restrict__ X, const int* __restrict__ ix,
                                                     const float* __restrict__ Ps, const float* __restrict__ Pt,
                                                     const float* __restrict__ W2s, const float* __restrict__ W2t,
                                                     const int* __restrict__ kptr, const int* __restrict__ map, int S,
                                                     float* __restrict__ out, const float* __restrict__ add = nullptr) {
  constexpr int C = 64 * CPL;
  constexpr int UT = 4;                                   // triplets whose loads are in flight together
  const int lane = threadIdx.x & 63;
  const int s = uni(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (s >= S) return;
  float ws_w[CPL][PB], wt_w[CPL][PB];
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    const float4* a = (const float4*)(W2s + (lane * CPL + q) * PB);
    const float4 a0 = a[0], a1 = a[1];
    ws_w[q][0] = a0.x; ws_w[q][1] = a0.y; ws_w[q][2] = a0.z; ws_w[q][3] = a0.w;
    ws_w[q][4] = a1.x; ws_w[q][5] = a1.y; ws_w[q][6] = a1.z; ws_w[q][7] = a1.w;
    if (TOR) {
      const float4* b = (const float4*)(W2t + (lane * CPL + q) * PB);
      const float4 b0 = b[0], b1 = b[1];
      wt_w[q][0] = b0.x; wt_w[q][1] = b0.y; wt_w[q][2] = b0.z; wt_w[q][3] = b0.w;
      wt_w[q][4] = b1.x; wt_w[q][5] = b1.y; wt_w[q][6] = b1.z; wt_w[q][7] = b1.w;
    }
  }
  Row<CPL> acc, arow;
#pragma unroll
  for (int q = 0; q < CPL; ++q) acc.v[q] = 0.f;
  if (ADD) arow = load_row<CPL>(add, (int64_t)s * C + lane * CPL);
  const int p0 = kptr[s], p1 = kptr[s + 1];
  if (p0 < p1) {
    // the INDEX chain of batch i + 1 (position -> triplet id -> gathered row id: two or three dependent scalar loads) is
    // requested before batch i is consumed, so a batch starts with its operand loads; slots past the end repeat the
    // segment's last triplet and are not accumulated
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
      float a[UT][PB], b[UT][PB];
      Row<CPL> x[UT];
#pragma unroll
      for (int u = 0; u < UT; ++u) {
        const float* pa = Ps + (int64_t)ntt[u] * PB;
#pragma unroll
        for (int k = 0; k < PB; ++k) a[u][k] = pa[k];
        if (TOR) {
          const float* pb = Pt + (int64_t)ntt[u] * PB;
#pragma unroll
          for (int k = 0; k < PB; ++k) b[u][k] = pb[k];
        }
        x[u] = load_row<CPL>(X, (int64_t)nrow[u] * C + lane * CPL);
      }
      if (p + UT < p1) req_idx(p + UT);
#pragma unroll
      for (int u = 0; u < UT; ++u) {
        if (p + u < p1) {
#pragma unroll
          for (int q = 0; q < CPL; ++q) {
            float v = x[u].v[q];
            v *= dot8u(ws_w[q], a[u]);
            if (TOR) v *= dot8u(wt_w[q], b[u]);
            acc.v[q] += v;
          }
        }
      }
    }
  }
  if (ADD) {
#pragma unroll
    for (int q = 0; q < CPL; ++q) acc.v[q] += arow.v[q];
  }
  store_row<CPL>(out, (int64_t)s * C + lane * CPL, acc);
}

// ------------------------------------------------------------------------------------------------------------------
// forward, the segment's index chain walked ONCE (route 3; the idea of segment.hip:featconv_wave).  k_trip_fwd_w asks for the
// projected rows of a batch and for the indices of the next one with scalar loads, and scalar loads return out of order: the
// wait before a batch's first product is lgkmcnt(0), i.e. it also waits for the index loads just issued — every batch of
// four triplets is a full memory trip of its wave.  Here lane l of the wave reads the position, triplet id, gathered row id
// and the two projected rows (8 + 8 floats, two or four 16-byte loads) of triplet l of the segment (chunks of 64): one or two
// dependent VECTOR trips for the whole chunk.  The projected rows go to the wave's slice of LDS and come back as broadcast
// reads (wave-uniform address), the row ids reach the gathers through v_readlane; UX row gathers are in flight, each
// register refilled as soon as its triplet is consumed, every load unconditional.  Same products in the same order:
// bit-identical to k_trip_fwd_w / k_trip_fwd.
// ------------------------------------------------------------------------------------------------------------------
template <int CPL, bool TOR, bool ADD = false>
__global__ void __launch_bounds__(256) k_trip_fwd_l(const float* __restrict__ X, const int* __restrict__ ix,
                                                     const float* __restrict__ Ps, const float* __restrict__ Pt,
                                                     const float* __restrict__ W2s, const float* __restrict__ W2t,
                                                     const int* __restrict__ kptr, const int* __restrict__ map, int S,
                                                     float* __restrict__ out, const float* __restrict__ add = nullptr,
                                                     int swz = 0) {
  constexpr int C = 64 * CPL;
  constexpr int PW = TOR ? 2 * PB : PB;                   // floats per triplet in LDS
  constexpr int UX = 4;                                   // row gathers in flight
  __shared__ __attribute__((aligned(16))) float sP[4][64 * PW];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // swz: XCD-contiguous block order (common.h: dig3d_xcd_block) — the rows a molecule's triplets gather then live in ONE L2
  const int s = uni(dig3d_xcd_block(swz) * 4 + wv);
  if (s >= S) return;
  float ws_w[CPL][PB], wt_w[CPL][PB];
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    const float4* a = (const float4*)(W2s + (lane * CPL + q) * PB);
    const float4 a0 = a[0], a1 = a[1];
    ws_w[q][0] = a0.x; ws_w[q][1] = a0.y; ws_w[q][2] = a0.z; ws_w[q][3] = a0.w;
    ws_w[q][4] = a1.x; ws_w[q][5] = a1.y; ws_w[q][6] = a1.z; ws_w[q][7] = a1.w;
    if (TOR) {
      const float4* b = (const float4*)(W2t + (lane * CPL + q) * PB);
      const float4 b0 = b[0], b1 = b[1];
      wt_w[q][0] = b0.x; wt_w[q][1] = b0.y; wt_w[q][2] = b0.z; wt_w[q][3] = b0.w;
      wt_w[q][4] = b1.x; wt_w[q][5] = b1.y; wt_w[q][6] = b1.z; wt_w[q][7] = b1.w;
    }
  }
  Row<CPL> acc, arow;
#pragma unroll
  for (int q = 0; q < CPL; ++q) acc.v[q] = 0.f;
  if (ADD) arow = load_row<CPL>(add, (int64_t)s * C + lane * CPL);
  const int p0 = kptr[s], p1 = kptr[s + 1];
  float* sp = sP[wv];
  for (int cb = p0; cb < p1; cb += 64) {
    const int n = p1 - cb < 64 ? p1 - cb : 64;            // wave-uniform
    const int pl = cb + (lane < n ? lane : n - 1);
    const int tl = map ? map[pl] : pl;
    const int rl = ix[tl];
    {
      const float4* pa = (const float4*)(Ps + (int64_t)tl * PB);
      const float4 a0 = pa[0], a1 = pa[1];
      float4 b0, b1;
      if (TOR) {
        const float4* pb = (const float4*)(Pt + (int64_t)tl * PB);
        b0 = pb[0]; b1 = pb[1];
      }
      float4* d = (float4*)(sp + lane * PW);
      d[0] = a0; d[1] = a1;
      if (TOR) { d[2] = b0; d[3] = b1; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    Row<CPL> xn[UX];
    auto request = [&](int u, int jpos) {
      const int jj = jpos < n ? jpos : n - 1;
      const int row = __builtin_amdgcn_readlane(rl, jj);
      xn[u] = load_row<CPL>(X, (int64_t)row * C + lane * CPL);
    };
#pragma unroll
    for (int u = 0; u < UX; ++u) request(u, u);
    for (int j = 0; j < n; j += UX) {
#pragma unroll
      for (int u = 0; u < UX; ++u) {
        const bool live = j + u < n;
        const float4* fr = (const float4*)(sp + (live ? j + u : n - 1) * PW);
        float a[PB], b[PB];
        {
          const float4 a0 = fr[0], a1 = fr[1];
          a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w; a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
          if (TOR) {
            const float4 b0 = fr[2], b1 = fr[3];
            b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
          }
        }
        const Row<CPL> x = xn[u];
        request(u, j + UX + u);             // unconditional (past the end: the last row again)
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          float v = x.v[q];
          v *= dot8u(ws_w[q], a);
          if (TOR) v *= dot8u(wt_w[q], b);
          const float sum = acc.v[q] + v;
          acc.v[q] = live ? sum : acc.v[q];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the next chunk rewrites the projected rows
    __builtin_amdgcn_wave_barrier();
  }
  if (ADD) {
#pragma unroll
    for (int q = 0; q < CPL; ++q) acc.v[q] += arow.v[q];
  }
  store_row<CPL>(out, (int64_t)s * C + lane * CPL, acc);
}

// ------------------------------------------------------------------------------------------------------------------
// backward: gPs/gPt [T,8] and the partials of gW2s/gW2t [C,8]; segments = edges e (tptr), every triplet of e shares G[e]
// ------------------------------------------------------------------------------------------------------------------
// v[0..15] per lane -> lane l of every 16-lane row holds sum over the row of v[l % 16]
__device__ __forceinline__ float dpp_ror8(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_half_mirror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_quad_rev(float v) {      // quad_perm [3,2,1,0]
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x1B, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_quad_swap(float v) {     // quad_perm [1,0,3,2]
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
}

// keep[i] / send[i], i < 8: the lane's halves for the first butterfly step (partner l ^ 8), already selected by bit 3
__device__ __forceinline__ float row16_reduce16(const float (&keep)[8], const float (&send)[8], int lane) {
  const bool b2 = (lane & 4) != 0, b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
  float w8[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) w8[i] = keep[i] + dpp_ror8(send[i]);
  // step 2: partner l ^ 7 (bit 3 equal, bit 2 differs); lanes with bit 2 keep the upper half
  float w4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float k = b2 ? w8[i + 4] : w8[i];
    const float sd = b2 ? w8[i] : w8[i + 4];
    w4[i] = k + dpp_half_mirror(sd);
  }
  // step 3: partner l ^ 3 (bits 3, 2 equal, bit 1 differs)
  float w2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float k = b1 ? w4[i + 2] : w4[i];
    const float sd = b1 ? w4[i] : w4[i + 2];
    w2[i] = k + dpp_quad_rev(sd);
  }
  // step 4: partner l ^ 1
  const float k = b0 ? w2[1] : w2[0];
  const float sd = b0 ? w2[0] : w2[1];
  return k + dpp_quad_swap(sd);
}

// r[l] + r[l ^ 16] + r[l ^ 32] + r[l ^ 48] on the VALU (v_permlane16_swap / v_permlane32_swap, gfx950): no LDS-pipe
// instruction in the triplet loop, so the only lgkmcnt traffic there is the scalar prefetch of the next triplet
__device__ __forceinline__ float cross_row_sum(float r) {
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  unsigned u = __float_as_uint(r);
  const v2u a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float s = __uint_as_float(a.x) + __uint_as_float(a.y);
  u = __float_as_uint(s);
  const v2u b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(b.x) + __uint_as_float(b.y);
}

// waves per block of the backward: every BLOCK leaves a partial of the second-Linear gradients (2 C x 8 floats) for the
// deferred reduction, so eight waves per block halve that traffic against four at the same number of resident waves
#define BW_WPB 8

// ADD: gPs[t, :] += gPs_add[t, :] (and gPt likewise when TOR) — see k_trip_fwd_w
template <int CPL, bool TOR, bool ADD = false>
__global__ void __launch_bounds__(64 * BW_WPB) k_trip_bwd_w(const float* __restrict__ G, const float* __restrict__ X,
                                                     const int* __restrict__ kj, const float* __restrict__ Ps,
                                                     const float* __restrict__ Pt, const float* __restrict__ W2s,
                                                     const float* __restrict__ W2t, const int* __restrict__ tptr, int E,
                                                     float* __restrict__ gPs, float* __restrict__ gPt,
                                                     float* __restrict__ part, const float* __restrict__ gPs_add = nullptr,
                                                     const float* __restrict__ gPt_add = nullptr) {
  constexpr int C = 64 * CPL;
  __shared__ float sred[BW_WPB * 64 * PB];          // cross-wave reduction of one (table, channel slot) at a time
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  float ws_w[CPL][PB], wt_w[CPL][PB], gs[CPL][PB], gt[CPL][PB];
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    const float4* a = (const float4*)(W2s + (lane * CPL + q) * PB);
    const float4 a0 = a[0], a1 = a[1];
    ws_w[q][0] = a0.x; ws_w[q][1] = a0.y; ws_w[q][2] = a0.z; ws_w[q][3] = a0.w;
    ws_w[q][4] = a1.x; ws_w[q][5] = a1.y; ws_w[q][6] = a1.z; ws_w[q][7] = a1.w;
    if (TOR) {
      const float4* b = (const float4*)(W2t + (lane * CPL + q) * PB);
      const float4 b0 = b[0], b1 = b[1];
      wt_w[q][0] = b0.x; wt_w[q][1] = b0.y; wt_w[q][2] = b0.z; wt_w[q][3] = b0.w;
      wt_w[q][4] = b1.x; wt_w[q][5] = b1.y; wt_w[q][6] = b1.z; wt_w[q][7] = b1.w;
    }
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      gs[q][b] = 0.f;
      gt[q][b] = 0.f;
      if (!TOR) wt_w[q][b] = 0.f;
    }
  }
  // first butterfly step folded into the products: a lane with bit 3 clear keeps the eight gP_s sums and sends the gP_t ones,
  // a lane with bit 3 set the other way round — its weights for "keep" and "send" are selected once, here
  const bool b3 = (lane & 8) != 0;
  float wk[CPL][PB], wsd[CPL][PB];
#pragma unroll
  for (int q = 0; q < CPL; ++q)
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      wk[q][b] = b3 ? wt_w[q][b] : ws_w[q][b];
      wsd[q][b] = b3 ? ws_w[q][b] : wt_w[q][b];
    }
  for (int e = uni(blockIdx.x * BW_WPB + wv); e < E; e += gridDim.x * BW_WPB) {
    const Row<CPL> g = load_row<CPL>(G, (int64_t)e * C + lane * CPL);
    const int t0 = tptr[e], t1 = tptr[e + 1];
    if (t0 >= t1) continue;
    // software pipeline: the operands of triplet t + 1 (row id -> gathered row, the two basis rows) are requested before
    // the arithmetic of triplet t — a wave alone on its SIMD (the long segments at the end of the launch) runs at the
    // arithmetic's pace instead of one memory round trip per triplet
    float pa[PB], pb[PB], na[PB], nb_[PB], nadd = 0.f;
    Row<CPL> x, nx;
    auto request = [&](int t) {
      const int row = kj[t];
      if (ADD) {
        if (lane < 8) nadd = gPs_add[(int64_t)t * PB + lane];
        else if (TOR && lane < 16) nadd = gPt_add[(int64_t)t * PB + (lane - 8)];
      }
      const float* qa = Ps + (int64_t)t * PB;
#pragma unroll
      for (int k = 0; k < PB; ++k) na[k] = qa[k];
      if (TOR) {
        const float* qb = Pt + (int64_t)t * PB;
#pragma unroll
        for (int k = 0; k < PB; ++k) nb_[k] = qb[k];
      }
      nx = load_row<CPL>(X, (int64_t)row * C + lane * CPL);
    };
    request(t0);
    for (int t = t0; t < t1; ++t) {
#pragma unroll
      for (int k = 0; k < PB; ++k) {
        pa[k] = na[k];
        pb[k] = TOR ? nb_[k] : 0.f;
      }
      x = nx;
      const float padd = nadd;
      if (t + 1 < t1) request(t + 1);
      float keep[8], send[8];
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        const float ws = dot8u(ws_w[q], pa);
        const float gx = g.v[q] * x.v[q];
        float gws, gwt;
        if (TOR) {
          const float wt = dot8u(wt_w[q], pb);
          gws = gx * wt;
          gwt = gx * ws;
        } else {
          gws = gx;
          gwt = 0.f;
        }
        const float gk = b3 ? gwt : gws, gsd = b3 ? gws : gwt;
#pragma unroll
        for (int b = 0; b < PB; ++b) {
          // (the lane's CPL channels are summed in channel order before the cross-lane reduction)
          keep[b] = q == 0 ? gk * wk[q][b] : fmaf(gk, wk[q][b], keep[b]);
          send[b] = q == 0 ? gsd * wsd[q][b] : fmaf(gsd, wsd[q][b], send[b]);
          gs[q][b] = fmaf(gws, pa[b], gs[q][b]);
          if (TOR) gt[q][b] = fmaf(gwt, pb[b], gt[q][b]);
        }
      }
      float r = row16_reduce16(keep, send, lane);    // lane l: sum number l % 16 over its 16-lane row
      r = cross_row_sum(r);
      if (ADD) r += padd;
      if (lane < 8) gPs[(int64_t)t * PB + lane] = r;
      else if (TOR && lane < 16) gPt[(int64_t)t * PB + (lane - 8)] = r;
    }
  }
  // block partial of the second-Linear weight gradients, layout of k_trip_bwd: part[block][table][C][PB]; the waves are
  // summed in wave order
  float* outp = part + (int64_t)blockIdx.x * (2 * C * PB);
#pragma unroll
  for (int br = 0; br < (TOR ? 2 : 1); ++br) {
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      __syncthreads();
#pragma unroll
      for (int b = 0; b < PB; ++b) sred[(wv * 64 + lane) * PB + b] = br == 0 ? gs[q][b] : gt[q][b];
      __syncthreads();
      for (int j = threadIdx.x; j < 64 * PB; j += 64 * BW_WPB) {
        const int ln = j / PB, b = j - ln * PB;
        float sum = sred[(0 * 64 + ln) * PB + b];
#pragma unroll
        for (int w2 = 1; w2 < BW_WPB; ++w2) sum += sred[(w2 * 64 + ln) * PB + b];      // the waves in wave order
        outp[(br * C + (ln * CPL + q)) * PB + b] = sum;
      }
    }
  }
}

// The backward with the segment's index chain walked ONCE (see k_trip_fwd_l): chunks of 32 triplets; lane (j, h) = (lane / 2,
// lane % 2) reads the row id and the projected row of table h of triplet j (one vector trip for the chunk) into the wave's
// slice of LDS; the rows come back as broadcast reads, the row ids through v_readlane; four unconditional gathers in flight.
// The 16 sums of a triplet are parked in LDS too and leave as two 16-byte stores per lane at the end of the chunk (were
// sixteen 4-byte stores per triplet).  The next edge's gradient row and triplet range are requested while this edge is
// worked on.  Same arithmetic, same order: results and partials are bit-identical to k_trip_bwd_w.
template <int CPL, bool TOR, bool ADD = false>
__global__ void __launch_bounds__(64 * BW_WPB) k_trip_bwd_l(const float* __restrict__ G, const float* __restrict__ X,
                                                     const int* __restrict__ kj, const float* __restrict__ Ps,
                                                     const float* __restrict__ Pt, const float* __restrict__ W2s,
                                                     const float* __restrict__ W2t, const int* __restrict__ tptr, int E,
                                                     float* __restrict__ gPs, float* __restrict__ gPt,
                                                     float* __restrict__ part, const float* __restrict__ gPs_add = nullptr,
                                                     const float* __restrict__ gPt_add = nullptr) {
  constexpr int C = 64 * CPL;
  constexpr int CH = 32, UX = 4;                      // triplets per chunk, row gathers in flight
  // per wave: the projected rows (CH x 16 floats) and the parked results (CH x 16 floats); the cross-wave reduction of the
  // weight gradients at the end reuses the space (BW_WPB * 64 * PB floats = the first half)
  __shared__ __attribute__((aligned(16))) float sbuf[BW_WPB * 2 * CH * 16];
  float* sred = sbuf;
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  float ws_w[CPL][PB], wt_w[CPL][PB], gs[CPL][PB], gt[CPL][PB];
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    const float4* a = (const float4*)(W2s + (lane * CPL + q) * PB);
    const float4 a0 = a[0], a1 = a[1];
    ws_w[q][0] = a0.x; ws_w[q][1] = a0.y; ws_w[q][2] = a0.z; ws_w[q][3] = a0.w;
    ws_w[q][4] = a1.x; ws_w[q][5] = a1.y; ws_w[q][6] = a1.z; ws_w[q][7] = a1.w;
    if (TOR) {
      const float4* b = (const float4*)(W2t + (lane * CPL + q) * PB);
      const float4 b0 = b[0], b1 = b[1];
      wt_w[q][0] = b0.x; wt_w[q][1] = b0.y; wt_w[q][2] = b0.z; wt_w[q][3] = b0.w;
      wt_w[q][4] = b1.x; wt_w[q][5] = b1.y; wt_w[q][6] = b1.z; wt_w[q][7] = b1.w;
    }
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      gs[q][b] = 0.f;
      gt[q][b] = 0.f;
      if (!TOR) wt_w[q][b] = 0.f;
    }
  }
  // first butterfly step folded into the products: a lane with bit 3 clear keeps the eight gP_s sums and sends the gP_t ones,
  // a lane with bit 3 set the other way round — its weights for "keep" and "send" are selected once, here
  const bool b3 = (lane & 8) != 0;
  float wk[CPL][PB], wsd[CPL][PB];
#pragma unroll
  for (int q = 0; q < CPL; ++q)
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      wk[q][b] = b3 ? wt_w[q][b] : ws_w[q][b];
      wsd[q][b] = b3 ? ws_w[q][b] : wt_w[q][b];
    }
  float* sp = sbuf + wv * (2 * CH * 16);
  float* so = sp + CH * 16;
  const int jl = lane >> 1, h = lane & 1;
  const int estride = gridDim.x * BW_WPB;
  int e = uni(blockIdx.x * BW_WPB + wv);
  Row<CPL> gnext;
  int t0n = 0, t1n = 0;
  if (e < E) {
    gnext = load_row<CPL>(G, (int64_t)e * C + lane * CPL);
    t0n = tptr[e];
    t1n = tptr[e + 1];
  }
  for (; e < E; e += estride) {
    const Row<CPL> g = gnext;
    const int t0 = t0n, t1 = t1n;
    {
      const int e2 = e + estride < E ? e + estride : E - 1;      // unconditional (past the end: the last edge, never used)
      gnext = load_row<CPL>(G, (int64_t)e2 * C + lane * CPL);
      t0n = tptr[e2];
      t1n = tptr[e2 + 1];
    }
    for (int cb = t0; cb < t1; cb += CH) {
      const int n = t1 - cb < CH ? t1 - cb : CH;                 // wave-uniform
      const int tl = cb + (jl < n ? jl : n - 1);
      const int rl = kj[tl];
      {
        const float* tab = (TOR && h) ? Pt : Ps;
        const float4* pr = (const float4*)(tab + (int64_t)tl * PB);
        const float4 r0 = pr[0], r1 = pr[1];
        float4* d = (float4*)(sp + jl * 16 + h * 8);
        d[0] = r0;
        d[1] = r1;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      Row<CPL> xn[UX];
      auto request = [&](int u, int jpos) {
        const int jj = jpos < n ? jpos : n - 1;
        const int row = __builtin_amdgcn_readlane(rl, 2 * jj);
        xn[u] = load_row<CPL>(X, (int64_t)row * C + lane * CPL);
      };
#pragma unroll
      for (int u = 0; u < UX; ++u) request(u, u);
      for (int j = 0; j < n; j += UX) {
#pragma unroll
        for (int u = 0; u < UX; ++u) {
          const Row<CPL> x = xn[u];
          request(u, j + UX + u);               // unconditional (past the end: the last row again)
          if (j + u < n) {                      // wave-uniform; no load inside
            float pa[PB], pb[PB];
            {
              const float4* fr = (const float4*)(sp + (j + u) * 16);
              const float4 a0 = fr[0], a1 = fr[1];
              pa[0] = a0.x; pa[1] = a0.y; pa[2] = a0.z; pa[3] = a0.w; pa[4] = a1.x; pa[5] = a1.y; pa[6] = a1.z; pa[7] = a1.w;
              if (TOR) {
                const float4 b0 = fr[2], b1 = fr[3];
                pb[0] = b0.x; pb[1] = b0.y; pb[2] = b0.z; pb[3] = b0.w; pb[4] = b1.x; pb[5] = b1.y; pb[6] = b1.z; pb[7] = b1.w;
              } else {
#pragma unroll
                for (int k = 0; k < PB; ++k) pb[k] = 0.f;
              }
            }
            float keep[8], send[8];
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
              const float ws = dot8u(ws_w[q], pa);
              const float gx = g.v[q] * x.v[q];
              float gws, gwt;
              if (TOR) {
                const float wt = dot8u(wt_w[q], pb);
                gws = gx * wt;
                gwt = gx * ws;
              } else {
                gws = gx;
                gwt = 0.f;
              }
              const float gk = b3 ? gwt : gws, gsd = b3 ? gws : gwt;
#pragma unroll
              for (int b = 0; b < PB; ++b) {
                keep[b] = q == 0 ? gk * wk[q][b] : fmaf(gk, wk[q][b], keep[b]);
                send[b] = q == 0 ? gsd * wsd[q][b] : fmaf(gsd, wsd[q][b], send[b]);
                gs[q][b] = fmaf(gws, pa[b], gs[q][b]);
                if (TOR) gt[q][b] = fmaf(gwt, pb[b], gt[q][b]);
              }
            }
            float r = row16_reduce16(keep, send, lane);    // lane l: sum number l % 16 over its 16-lane row
            r = cross_row_sum(r);
            if (lane < 16) so[(j + u) * 16 + lane] = r;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      {
        // lane (j, h): the eight sums of table h of triplet j
        const float4* sr = (const float4*)(so + jl * 16 + h * 8);
        float4 o0 = sr[0], o1 = sr[1];
        if (ADD) {
          const float* at = (TOR && h) ? gPt_add : gPs_add;
          const float4* ad = (const float4*)(at + (int64_t)tl * PB);
          const float4 d0 = ad[0], d1 = ad[1];
          o0.x += d0.x; o0.y += d0.y; o0.z += d0.z; o0.w += d0.w;
          o1.x += d1.x; o1.y += d1.y; o1.z += d1.z; o1.w += d1.w;
        }
        if (jl < n && (TOR || h == 0)) {
          float4* dst = (float4*)(((TOR && h) ? gPt : gPs) + (int64_t)tl * PB);
          dst[0] = o0;
          dst[1] = o1;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the next chunk rewrites both slices
      __builtin_amdgcn_wave_barrier();
    }
  }
  // block partial of the second-Linear weight gradients, layout of k_trip_bwd: part[block][table][C][PB]; the waves are
  // summed in wave order
  float* outp = part + (int64_t)blockIdx.x * (2 * C * PB);
#pragma unroll
  for (int br = 0; br < (TOR ? 2 : 1); ++br) {
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      __syncthreads();
#pragma unroll
      for (int b = 0; b < PB; ++b) sred[(wv * 64 + lane) * PB + b] = br == 0 ? gs[q][b] : gt[q][b];
      __syncthreads();
      for (int j = threadIdx.x; j < 64 * PB; j += 64 * BW_WPB) {
        const int ln = j / PB, b = j - ln * PB;
        float sum = sred[(0 * 64 + ln) * PB + b];
#pragma unroll
        for (int w2 = 1; w2 < BW_WPB; ++w2) sum += sred[(w2 * 64 + ln) * PB + b];      // the waves in wave order
        outp[(br * C + (ln * CPL + q)) * PB + b] = sum;
      }
    }
  }
}

}  // namespace

// 0: launched; 1: this channel count keeps the 16-lane kernels (C = 16, 32)
int trip_fwd_wave(const float* X, const int* ix, const float* Ps, const float* Pt, const float* W2s, const float* W2t,
                  const int* kptr, const int* map, int S, int C, float* out, const float* add, hipStream_t st) {
  const bool tor = Pt != nullptr;
  const dim3 grid((S + 3) / 4), block(256);
#define TFW(CPL)                                                                                                       \
  do {                                                                                                                 \
    if (tor && add) hipLaunchKernelGGL((k_trip_fwd_w<CPL, true, true>), grid, block, 0, st, X, ix, Ps, Pt, W2s, W2t, kptr, map, S, out, add); \
    else if (tor) hipLaunchKernelGGL((k_trip_fwd_w<CPL, true>), grid, block, 0, st, X, ix, Ps, Pt, W2s, W2t, kptr, map, S, out, nullptr); \
    else if (add) hipLaunchKernelGGL((k_trip_fwd_w<CPL, false, true>), grid, block, 0, st, X, ix, Ps, Pt, W2s, W2t, kptr, map, S, out, add); \
    else hipLaunchKernelGGL((k_trip_fwd_w<CPL, false>), grid, block, 0, st, X, ix, Ps, Pt, W2s, W2t, kptr, map, S, out, nullptr);   \
  } while (0)
  switch (C) {
    case 64: TFW(1); return 0;
    case 128: TFW(2); return 0;
    case 256: TFW(4); return 0;
    default: return 1;
  }
#undef TFW
}

int trip_fwd_lds(const float* X, const int* ix, const float* Ps, const float* Pt, const float* W2s, const float* W2t,
                 const int* kptr, const int* map, int S, int C, float* out, const float* add, hipStream_t st) {
  const bool tor = Pt != nullptr;
  // from 65 536 segments on the gathered table no longer fits the eight 4-MB L2s side by side unless every XCD works on its own
  // range of molecules: 92.9 -> 88.7 us at 1.2e5 segments / 1.6e6 triplets; nothing at the step sizes (3.7e4: 4.687 vs 4.700 ms
  // per config-4 step), so the natural order stays there
  const int nblk = (S + 3) / 4, swz = S >= 65536 ? 1 : 0;
  const dim3 grid(swz ? dig3d_xcd_grid(nblk) : nblk), block(256);
#define TFL(CPL)                                                                                                       \
  do {                                                                                                                 \
    if (tor && add) hipLaunchKernelGGL((k_trip_fwd_l<CPL, true, true>), grid, block, 0, st, X, ix, Ps, Pt, W2s, W2t, kptr, map, S, out, add, swz); \
    else if (tor) hipLaunchKernelGGL((k_trip_fwd_l<CPL, true>), grid, block, 0, st, X, ix, Ps, Pt, W2s, W2t, kptr, map, S, out, nullptr, swz); \
    else if (add) hipLaunchKernelGGL((k_trip_fwd_l<CPL, false, true>), grid, block, 0, st, X, ix, Ps, Pt, W2s, W2t, kptr, map, S, out, add, swz); \
    else hipLaunchKernelGGL((k_trip_fwd_l<CPL, false>), grid, block, 0, st, X, ix, Ps, Pt, W2s, W2t, kptr, map, S, out, nullptr, swz);   \
  } while (0)
  switch (C) {
    case 64: TFL(1); return 0;
    case 128: TFL(2); return 0;
    case 256: TFL(4); return 0;
    default: return 1;
  }
#undef TFL
}

int trip_bwd_wave_blocks(int E, int C) {
  if (C != 64 && C != 128 && C != 256) return 0;
  // one wave per segment up to eight waves per SIMD (4 blocks of 8 waves per CU), then strided.  (Four-wave blocks, 8 per
  // CU: the kernel 30 us instead of 34 at the config-2 size, but 1 946 partials per layer for the deferred reduction to read
  // back — k_reduce_many 91 us instead of ~40 in the step, a net loss.)
  int nb = (E + BW_WPB - 1) / BW_WPB;
  const int cap = 4 * dig3d_num_cus();
  if (nb > cap) nb = cap;
  return nb < 1 ? 1 : nb;
}

int trip_bwd_wave(const float* G, const float* X, const int* kj, const float* Ps, const float* Pt, const float* W2s,
                  const float* W2t, const int* tptr, int E, int C, float* gPs, float* gPt, float* part, int nb,
                  const float* gPs_add, const float* gPt_add, hipStream_t st) {
  const bool tor = Pt != nullptr;
  const bool add = gPs_add != nullptr;
#define TBW(CPL)                                                                                                          \
  do {                                                                                                                    \
    if (tor && add) hipLaunchKernelGGL((k_trip_bwd_w<CPL, true, true>), dim3(nb), dim3(64 * BW_WPB), 0, st, G, X, kj, Ps, Pt, W2s, W2t, tptr, E, gPs, gPt, part, gPs_add, gPt_add); \
    else if (tor) hipLaunchKernelGGL((k_trip_bwd_w<CPL, true>), dim3(nb), dim3(64 * BW_WPB), 0, st, G, X, kj, Ps, Pt, W2s, W2t, tptr, E, gPs, gPt, part, nullptr, nullptr); \
    else if (add) hipLaunchKernelGGL((k_trip_bwd_w<CPL, false, true>), dim3(nb), dim3(64 * BW_WPB), 0, st, G, X, kj, Ps, Pt, W2s, W2t, tptr, E, gPs, gPt, part, gPs_add, nullptr); \
    else hipLaunchKernelGGL((k_trip_bwd_w<CPL, false>), dim3(nb), dim3(64 * BW_WPB), 0, st, G, X, kj, Ps, Pt, W2s, W2t, tptr, E, gPs, gPt, part, nullptr, nullptr);   \
  } while (0)
  switch (C) {
    case 64: TBW(1); return 0;
    case 128: TBW(2); return 0;
    case 256: TBW(4); return 0;
    default: return 1;
  }
#undef TBW
}

int trip_bwd_lds(const float* G, const float* X, const int* kj, const float* Ps, const float* Pt, const float* W2s,
                  const float* W2t, const int* tptr, int E, int C, float* gPs, float* gPt, float* part, int nb,
                  const float* gPs_add, const float* gPt_add, hipStream_t st) {
  const bool tor = Pt != nullptr;
  const bool add = gPs_add != nullptr;
#define TBL(CPL)                                                                                                          \
  do {                                                                                                                    \
    if (tor && add) hipLaunchKernelGGL((k_trip_bwd_l<CPL, true, true>), dim3(nb), dim3(64 * BW_WPB), 0, st, G, X, kj, Ps, Pt, W2s, W2t, tptr, E, gPs, gPt, part, gPs_add, gPt_add); \
    else if (tor) hipLaunchKernelGGL((k_trip_bwd_l<CPL, true>), dim3(nb), dim3(64 * BW_WPB), 0, st, G, X, kj, Ps, Pt, W2s, W2t, tptr, E, gPs, gPt, part, nullptr, nullptr); \
    else if (add) hipLaunchKernelGGL((k_trip_bwd_l<CPL, false, true>), dim3(nb), dim3(64 * BW_WPB), 0, st, G, X, kj, Ps, Pt, W2s, W2t, tptr, E, gPs, gPt, part, gPs_add, nullptr); \
    else hipLaunchKernelGGL((k_trip_bwd_l<CPL, false>), dim3(nb), dim3(64 * BW_WPB), 0, st, G, X, kj, Ps, Pt, W2s, W2t, tptr, E, gPs, gPt, part, nullptr, nullptr);   \
  } while (0)
  switch (C) {
    case 64: TBL(1); return 0;
    case 128: TBL(2); return 0;
    case 256: TBL(4); return 0;
    default: return 1;
  }
#undef TBL
}
