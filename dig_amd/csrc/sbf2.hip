// dig3d — the angular basis of DimeNet++ contracted with the first basis Linears of ALL interaction blocks, as a family of
// two kernels CLOSED UNDER DIFFERENTIATION: the energy_and_force route's twin of k_basis_project (triplet.hip).
//
//   reference:  sbf[t, l nr + n] = bes[idx_kj[t], l nr + n] * Y_l0(angle[t])          method/dimenetpp/features.py:183-220
//               P_b[t, :]        = lin_sbf1_b(sbf[t, :])   for every block b           method/dimenetpp/dimenetpp.py:146
//   run.py:126-131 differentiates this TWICE (force = -dE/dpos with create_graph, then loss.backward()).
//
// Round 5 formed the [T, ns nr] table with a framework broadcasting multiply (force_path.py) and ran a stacked T-row dense
// layer over it in every pass: ~30 launches and ~0.6 ms of a 5.3-ms config-3 step.  Here the table is never formed:
//
//   k_sbf_t   a thread per triplet.  With  u0[ln] = A[kj,ln] h0_l + s B[kj,ln] h1_l,   u1[ln] = A[kj,ln] h1_l + s B[kj,ln] h2_l
//             (h0, h1, h2 = Y_l0, dY_l0/dtheta, d2Y_l0/dtheta2 at angle[t];  B, s optional):
//                 outP[t, j]  = sum_ln W[j, ln] u0[ln]                      the projection            (forward; d/dgP in pass 3)
//                 outA[t]     = sum_j gP[t, j] sum_ln W[j, ln] u1[ln]      gradient w.r.t. the angle  (passes 2, 3, 4)
//                               (evaluated as sum_ln u1[ln] v[ln], v = gP W, the sum over the degrees l in float64)
//                 partW[blk]  = sum_{t in block} gP[t, j] u0[ln]           weight-gradient partials   (passes 3, 4)
//                 Hout[t, l]  = B ? s h1_l : h0_l                           what k_sbf_e multiplies by
//   k_sbf_e   a wave per edge over the transposed CSR of idx_kj:
//                 out[e, l nr + n] = sum_j W[j, l nr + n] sum_{t: kj[t] = e} Hout[t, l] gP[t, j]      gradient w.r.t. bes
//
// pass 1 (forward):                 k_sbf_t(A = bes)                          -> P
// pass 2 (create_graph backward):   k_sbf_t(A = bes; gP) -> g_angle, Hout;    k_sbf_e -> g_bes
// pass 3 (its backward; c_bes, c_a the incoming gradients of g_bes, g_angle):
//                                   k_sbf_t(A = c_bes, B = bes, s = c_a; gP) -> d/dgP, d/dangle, partials of d/dW, Hout
//                                   k_sbf_e -> d/dbes
// pass 4 (final backward):          as pass 2, plus the weight partials
// — seven launches per step.  J = 8 L outputs live in L separate [T, 8] matrices (the operands of the fused triplet kernels),
// read and written through pointer tables: no column split / merge kernels.  Rows t >= *cnt (padding of a static-shape
// batch) are never read and written as zeros; the partials are reduced in block order (dig3d_reduce_many): deterministic.
#include "common.h"
#include "sph.h"

#define SB_TT 128          // triplets per block of k_sbf_t (= threads)
#define SB_LMAX 8          // blocks (groups of 8 outputs)

struct SbfPtrs {
  const float* gP[SB_LMAX];      // [T, 8] each, or null table
  float* outP[SB_LMAX];
};

// (Y_l0, dY_l0/dtheta, d2Y_l0/dtheta2), l < NS: values from the float32 recurrences the energy route uses (sph.h), the
// derivatives from the Legendre recurrences in float64 (as every derivative kernel of csrc/diffgeom.hip)
template <int NS>
__device__ __forceinline__ void sbf_harm(float theta, const float* __restrict__ pref, float (&h0)[NS], double (&h1)[NS],
                                         double (&h2)[NS]) {
  real_sph_harm<NS>(theta, 0.f, pref, true, h0);
  const double th = (double)theta, x = cos(th), sn = sin(th);
  double p[NS], d1[NS], d2[NS];
  p[0] = 1.0; d1[0] = 0.0; d2[0] = 0.0;
  if (NS > 1) { p[1] = x; d1[1] = 1.0; d2[1] = 0.0; }
#pragma unroll
  for (int l = 2; l < NS; ++l) {
    const double a = (double)(2 * l - 1), b = (double)(l - 1), c = 1.0 / (double)l;
    p[l] = (a * x * p[l - 1] - b * p[l - 2]) * c;
    d1[l] = (a * (p[l - 1] + x * d1[l - 1]) - b * d1[l - 2]) * c;
    d2[l] = (a * (2.0 * d1[l - 1] + x * d2[l - 1]) - b * d2[l - 2]) * c;
  }
#pragma unroll
  for (int l = 0; l < NS; ++l) {
    const double k = (double)pref[l * NS_MAX];
    h1[l] = k * (-sn * d1[l]);
    h2[l] = k * (sn * sn * d2[l] - x * d1[l]);
  }
}

// J is a template parameter (8, 16, 32, 64): the 32 running sums of a triplet are compile-time registers, and for every input
// (l, n) the 32 weights are one contiguous LDS row read with ds_read_b128 (all lanes the same address: a broadcast) — 32
// INDEPENDENT multiply-adds per row instead of one dependent chain per output.
template <int NS, int NR, int J, bool HASGP>
__global__ void __launch_bounds__(SB_TT) k_sbf_t(const float* __restrict__ A, const float* __restrict__ B,
                                                  const float* __restrict__ s, const float* __restrict__ angle,
                                                  const int* __restrict__ kj, const float* __restrict__ W,
                                                  const float* __restrict__ pref, SbfPtrs ptrs, int has_outp,
                                                  float* __restrict__ outA, float* __restrict__ partW,
                                                  float* __restrict__ Hout, int T, const int* __restrict__ cnt) {
  constexpr int K = NS * NR, L = J / 8, KP = K + 1;
  extern __shared__ float sm[];
  float* sWt = sm;                                  // [K][J]: the weights transposed
  float* sA = sm + 64 * K;                          // [SB_TT][K + 1]: this thread's gathered row of A (then u0, for the partials)
  float* sB = sA + SB_TT * KP;                      // [SB_TT][K + 1]: s * its row of B              (only when B)
  float* sG = sB + (B ? SB_TT * KP : 0);            // [SB_TT][J + 1]                                 (weight partials only)
  __shared__ float sPref[NS_MAX * NS_MAX];
  const int tid = threadIdx.x;
  for (int q = tid; q < J * K; q += SB_TT) {
    const int j = q / K, ln = q - j * K;
    sWt[ln * J + j] = W[q];
  }
  for (int q = tid; q < NS_MAX * NS_MAX; q += SB_TT) sPref[q] = pref[q];
  __syncthreads();
  const int t = blockIdx.x * SB_TT + tid;
  const int tl = cnt ? min(T, *cnt) : T;
  const bool live = t < tl;
  float h0[NS];
  double h1[NS], h2[NS];
  float sv = 0.f;
#pragma unroll
  for (int l = 0; l < NS; ++l) { h0[l] = 0.f; h1[l] = h2[l] = 0.0; }
  float* __restrict__ arow = sA + tid * KP;
  float* __restrict__ brow = B ? sB + tid * KP : arow;
  {  // the gathered rows, staged through this thread's LDS slots (independent loads, issued 8 at a time)
    const int e = live ? kj[t] : 0;
    const float keep = live ? 1.0f : 0.f;
    const float* __restrict__ ra = A + (int64_t)e * K;
    int k = 0;
    for (; k + 8 <= K; k += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ra[k + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) arow[k + u] = v[u] * keep;
    }
    for (; k < K; ++k) arow[k] = ra[k] * keep;
    if (B) {
      sv = (live && s) ? s[t] : (live ? 1.0f : 0.f);
      const float* __restrict__ rbp = B + (int64_t)e * K;
      k = 0;
      for (; k + 8 <= K; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = rbp[k + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) brow[k + u] = v[u] * sv;
      }
      for (; k < K; ++k) brow[k] = rbp[k] * sv;
    }
  }
  if (live) sbf_harm<NS>(angle[t], sPref, h0, h1, h2);
  if (Hout && t < T) {
    float hv[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) hv[l] = l < NS ? (B ? sv * (float)h1[l < NS ? l : 0] : h0[l < NS ? l : 0]) : 0.f;
    float4* ho = (float4*)(Hout + (int64_t)t * 8);
    ho[0] = make_float4(hv[0], hv[1], hv[2], hv[3]);
    ho[1] = make_float4(hv[4], hv[5], hv[6], hv[7]);
  }
  float gp[J], acc[J];
#pragma unroll
  for (int j = 0; j < J; ++j) gp[j] = acc[j] = 0.f;
  if (HASGP && live) {
#pragma unroll
    for (int b = 0; b < L; ++b) {
      const float4* g = (const float4*)(ptrs.gP[b] + (int64_t)t * 8);
      const float4 g0 = g[0], g1 = g[1];
      gp[8 * b + 0] = g0.x; gp[8 * b + 1] = g0.y; gp[8 * b + 2] = g0.z; gp[8 * b + 3] = g0.w;
      gp[8 * b + 4] = g1.x; gp[8 * b + 5] = g1.y; gp[8 * b + 6] = g1.z; gp[8 * b + 7] = g1.w;
    }
  }
  // ---- one pass over the inputs (l, n): P[j] += W[j, ln] u0[ln];  v[ln] = sum_j gP[j] W[j, ln] feeds the angle gradient
  //      sum_l [h1_l sum_n A[ln] v[ln] + s h2_l sum_n B[ln] v[ln]] — the sum over the degrees l (harmonics of alternating
  //      sign: heavy cancellation) in float64, as the derivative kernels of csrc/diffgeom.hip do
  double ga = 0.0;
#pragma unroll
  for (int l = 0; l < NS; ++l) {
    float sa = 0.f, sb = 0.f;
    const float y0 = h0[l], y1 = B ? (float)h1[l] : 0.f;
#pragma unroll 1
    for (int n = 0; n < NR; ++n) {
      const int ln = l * NR + n;
      const float av = arow[ln], bv = B ? brow[ln] : 0.f;
      const float u0 = av * y0 + bv * y1;
      if (partW) arow[ln] = u0;                      // (this thread's own slot: the row is not needed again)
      const float4* __restrict__ w4 = (const float4*)(sWt + ln * J);
      float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
#pragma unroll
      for (int j4 = 0; j4 < J / 4; ++j4) {
        const float4 w = w4[j4];
        acc[4 * j4 + 0] = fmaf(w.x, u0, acc[4 * j4 + 0]);
        acc[4 * j4 + 1] = fmaf(w.y, u0, acc[4 * j4 + 1]);
        acc[4 * j4 + 2] = fmaf(w.z, u0, acc[4 * j4 + 2]);
        acc[4 * j4 + 3] = fmaf(w.w, u0, acc[4 * j4 + 3]);
        if (HASGP) {
          v0 = fmaf(w.x, gp[4 * j4 + 0], v0);
          v1 = fmaf(w.y, gp[4 * j4 + 1], v1);
          v2 = fmaf(w.z, gp[4 * j4 + 2], v2);
          v3 = fmaf(w.w, gp[4 * j4 + 3], v3);
        }
      }
      if (HASGP) {
        const float v = (v0 + v1) + (v2 + v3);
        sa = fmaf(av, v, sa);
        sb = fmaf(bv, v, sb);
      }
    }
    if (HASGP) ga += h1[l] * (double)sa + h2[l] * (double)sb;      // (the staged B row already carries s)
  }
  if (has_outp && t < T) {
#pragma unroll
    for (int b = 0; b < L; ++b) {
      float4* po = (float4*)(ptrs.outP[b] + (int64_t)t * 8);
      po[0] = make_float4(acc[8 * b + 0], acc[8 * b + 1], acc[8 * b + 2], acc[8 * b + 3]);
      po[1] = make_float4(acc[8 * b + 4], acc[8 * b + 5], acc[8 * b + 6], acc[8 * b + 7]);
    }
  }
  if (HASGP && outA && t < T) outA[t] = (float)ga;
  // ---- weight-gradient partials of this block: part[blk][j K + ln] = sum_t gP[t, j] u0[t, ln]
  if (HASGP && partW) {
#pragma unroll
    for (int j = 0; j < J; ++j) sG[tid * (J + 1) + j] = gp[j];
    __syncthreads();
    // part[j][ln] = sum_r gP[r][j] u0[r][ln] over the block's 128 triplets on the matrix cores (v_mfma_f32_16x16x4_f32, exact
    // float32 multiply-adds): wave w takes the 16-row tiles jt = w, w + 2, ... of j; three 16-column tiles of ln.  A[m = j][k = r]
    // = sG[r][j], B[k = r][n = ln] = sA[r][ln]; D: lane (x, q) holds rows 4q .. 4q + 3 of column x.  (The thread-per-output
    // loop this replaces read two LDS words per multiply-add: 50-70 us of a 100-us launch.)
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    float* __restrict__ po = partW + (int64_t)blockIdx.x * J * K;
    const int wave = tid >> 6, lane = tid & 63, x = lane & 15, q = lane >> 4;
    constexpr int NT = (K + 15) / 16;
    for (int jt = wave; jt * 16 < J; jt += SB_TT / 64) {
      f32x4 c[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) c[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int jrow = 16 * jt + x;
      const bool jok = jrow < J;
#pragma unroll 4
      for (int st = 0; st < SB_TT / 4; ++st) {
        const int r = 4 * st + q;
        const float av = jok ? sG[r * (J + 1) + jrow] : 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int ln = 16 * nt + x;
          const float bv = ln < K ? sA[r * KP + ln] : 0.f;
          c[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c[nt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int ln = 16 * nt + x;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const int j = 16 * jt + 4 * q + cc;
          if (ln < K && j < J) po[j * K + ln] = c[nt][cc];
        }
      }
    }
  }
}

// a wave per edge over the transposed CSR (kptr, perm) of idx_kj; out [E, NS NR]
template <int NS, int NR>
__global__ void __launch_bounds__(256) k_sbf_e(const float* __restrict__ H, SbfPtrs ptrs, const float* __restrict__ W, int J,
                                                const int* __restrict__ kptr, const int* __restrict__ perm, int E,
                                                float* __restrict__ out) {
  constexpr int K = NS * NR;
  extern __shared__ float sm[];
  float* sW = sm;                                   // [J][K]
  float* sQ = sm + 64 * K;                          // [4 waves][64 * NS_MAX]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int q = tid; q < J * K; q += 256) sW[q] = W[q];
  __syncthreads();
  const int e = blockIdx.x * 4 + wave;
  if (e >= E) return;
  const int p0 = kptr[e], p1 = kptr[e + 1];
  // lane owns output j = lane (J <= 64): Q[j][l] = sum_t H[t, l] gP[t, j]
  float q[NS];
#pragma unroll
  for (int l = 0; l < NS; ++l) q[l] = 0.f;
  const bool own = lane < J;
  const float* __restrict__ gbase = own ? ptrs.gP[lane >> 3] : ptrs.gP[0];
  const int gc = lane & 7;
  // the edge's triplet ids are read ONCE, a lane per triplet (chunks of 64), and reach the loads through v_readlane; four
  // triplets' rows are in flight, every load unconditional.  (One triplet at a time, perm -> rows: two dependent trips per
  // triplet with nothing behind them — 26 us per launch at 1.0e5 triplets.)  Same products in the same order.
  const float mown = own ? 1.f : 0.f;
  constexpr int U = 4;
  for (int cb = p0; cb < p1; cb += 64) {
    const int n = p1 - cb < 64 ? p1 - cb : 64;              // wave-uniform
    const int pl = cb + (lane < n ? lane : n - 1);
    const int tl = perm ? perm[pl] : pl;
    float4 han[U], hbn[U];
    float gvn[U];
    auto request = [&](int u, int jpos) {
      const int t = __builtin_amdgcn_readlane(tl, jpos < n ? jpos : n - 1);
      han[u] = *(const float4*)(H + (int64_t)t * 8);
      hbn[u] = *(const float4*)(H + (int64_t)t * 8 + 4);
      gvn[u] = gbase[(int64_t)t * 8 + gc];
    };
#pragma unroll
    for (int u = 0; u < U; ++u) request(u, u);
    for (int j = 0; j < n; j += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float4 ha = han[u], hb = hbn[u];
        const float gv = gvn[u] * mown;
        request(u, j + U + u);
        if (j + u < n) {                                     // wave-uniform; no load inside
          const float hv[8] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
#pragma unroll
          for (int l = 0; l < NS; ++l) q[l] = fmaf(hv[l], gv, q[l]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float* __restrict__ mq = sQ + wave * (64 * NS_MAX);
#pragma unroll
  for (int l = 0; l < NS; ++l) mq[lane * NS_MAX + l] = q[l];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane < K) {
    const int l = lane / NR;
    float acc = 0.f;
    for (int j = 0; j < J; ++j) acc = fmaf(sW[j * K + lane], mq[j * NS_MAX + l], acc);
    out[(int64_t)e * K + lane] = acc;
  }
}

extern "C" {

int dig3d_sbf2_supported(int ns, int nr) { return (ns == 7 && nr == 6) || (ns == 3 && nr == 6) || (ns == 3 && nr == 4); }
int dig3d_sbf2_blocks(int T) { return T <= 0 ? 1 : (T + SB_TT - 1) / SB_TT; }

// see the file header.  A, B [E, ns nr] (B, s NULL: single table); angle [T]; kj [T] int32; W [J, ns nr], J = 8 L <= 64;
// gP / outP: host arrays of L device pointers to [T, 8] matrices (NULL array: absent); outA [T], partW
// float[dig3d_sbf2_blocks(T) * J * ns nr], Hout [T, 8] — each optional.  cnt: device live triplet count or NULL.
int dig3d_sbf2_t(const float* A, const float* B, const float* s, const float* angle, const int* kj, const float* W, int J,
                 int ns, int nr, const float* pref, const void* const* gP, void* const* outP, float* outA, float* partW,
                 float* Hout, int T, const int* cnt, void* stream) {
  DIG3D_ENTER();
  if (T < 0 || !A || !angle || !kj || !W || !pref || J < 8 || J > 64 || (J & 7) || !dig3d_sbf2_supported(ns, nr)) return DIG3D_ERR_ARG;
  if ((partW || outA) && !gP) return DIG3D_ERR_ARG;
  if (T == 0) return DIG3D_OK;
  SbfPtrs p;
  for (int b = 0; b < SB_LMAX; ++b) {
    p.gP[b] = (gP && b < J / 8) ? (const float*)gP[b] : nullptr;
    p.outP[b] = (outP && b < J / 8) ? (float*)outP[b] : nullptr;
    if (b < J / 8 && ((gP && (!p.gP[b] || ((uintptr_t)p.gP[b] & 15))) || (outP && (!p.outP[b] || ((uintptr_t)p.outP[b] & 15)))))
      return DIG3D_ERR_ARG;
  }
  if (Hout && ((uintptr_t)Hout & 15)) return DIG3D_ERR_ARG;
  const int K = ns * nr;
  if (J != 8 && J != 16 && J != 32 && J != 64) return DIG3D_ERR_ARG;
  const size_t shm = sizeof(float) * (64 * K + SB_TT * (K + 1) * (B ? 2 : 1) + (partW ? SB_TT * (J + 1) : 0));
  const dim3 grid(dig3d_sbf2_blocks(T));
  hipStream_t st = (hipStream_t)stream;
#define SBT3(NS, NR, JJ, G)                                                                                                  \
  {                                                                                                                          \
    static const bool ok = hipFuncSetAttribute((const void*)k_sbf_t<NS, NR, JJ, G>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                               (int)(sizeof(float) * (64 * NS * NR + 2 * SB_TT * (NS * NR + 1) + SB_TT * 65))) == hipSuccess; \
    if (!ok) return DIG3D_ERR_LAUNCH;                                                                                        \
    hipLaunchKernelGGL((k_sbf_t<NS, NR, JJ, G>), grid, dim3(SB_TT), shm, st, A, B, s, angle, kj, W, pref, p, outP ? 1 : 0,    \
                       outA, partW, Hout, T, cnt);                                                                           \
  }
#define SBT2(NS, NR, JJ)          \
  {                               \
    if (gP) SBT3(NS, NR, JJ, true) \
    else SBT3(NS, NR, JJ, false)   \
  }
#define SBT(NS, NR)                  \
  {                                  \
    if (J == 32) SBT2(NS, NR, 32)    \
    else if (J == 16) SBT2(NS, NR, 16) \
    else if (J == 8) SBT2(NS, NR, 8) \
    else SBT2(NS, NR, 64)            \
  }
  if (ns == 7 && nr == 6) SBT(7, 6)
  else if (ns == 3 && nr == 6) SBT(3, 6)
  else SBT(3, 4)
#undef SBT
#undef SBT2
#undef SBT3
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// out[e, l nr + n] = sum_j W[j, l nr + n] sum_{p in [kptr[e], kptr[e+1])} H[t, l] gP[t, j],  t = perm ? perm[p] : p
int dig3d_sbf2_e(const float* H, const void* const* gP, const float* W, int J, int ns, int nr, const int* kptr, const int* perm,
                 int E, float* out, void* stream) {
  DIG3D_ENTER();
  if (E < 0 || !H || !gP || !W || !kptr || !out || J < 8 || J > 64 || (J & 7) || !dig3d_sbf2_supported(ns, nr)) return DIG3D_ERR_ARG;
  if (E == 0) return DIG3D_OK;
  SbfPtrs p;
  for (int b = 0; b < SB_LMAX; ++b) {
    p.gP[b] = b < J / 8 ? (const float*)gP[b] : nullptr;
    p.outP[b] = nullptr;
    if (b < J / 8 && !p.gP[b]) return DIG3D_ERR_ARG;
  }
  const int K = ns * nr;
  const size_t shm = sizeof(float) * (64 * K + 4 * 64 * NS_MAX);
  const dim3 grid((E + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
#define SBE(NS, NR)                                                                                                      \
  {                                                                                                                      \
    static const bool ok = hipFuncSetAttribute((const void*)k_sbf_e<NS, NR>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                               (int)(sizeof(float) * (64 * NS * NR + 4 * 64 * NS_MAX))) == hipSuccess;   \
    if (!ok) return DIG3D_ERR_LAUNCH;                                                                                    \
    hipLaunchKernelGGL((k_sbf_e<NS, NR>), grid, dim3(256), shm, st, H, p, W, J, kptr, perm, E, out);                     \
  }
  if (ns == 7 && nr == 6) SBE(7, 6)
  else if (ns == 3 && nr == 6) SBE(3, 6)
  else SBE(3, 4)
#undef SBE
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

}  // extern "C"
