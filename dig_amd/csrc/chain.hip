// dig3d layer chain, first-order passes — the 8-layer residual block after the triplet aggregation
//     Y_l = res_l + act_l(Y_{l-1} W_l^T + b_l)                       (spherenet.py:172-182, :34-50; dimenetpp.py:152-161)
// and its input-gradient recursion
//     gZ_l = g_l * act'(Z_l),   g_{l-1} = gZ_l W_l  (+ skip)          (autograd of the same lines)
// on row tiles that never leave the CU.  Replaces the round-2 kernels (dense.hip: 64-row tiles, W_l staged through LDS
// for every layer, four barriers per layer, half of the CUs idle at E ~ 8.4k rows: 82.8 / 77.0 us per chain,
// profiles/r03a_spherenet_qm9_kernel_stats.csv).
//
// Layout of one workgroup (8 waves, R = 16 * RB rows, RB in {1, 2, 3, 4}):
//   * wave w owns the 16 output channels [16w, 16w + 16) of EVERY row of the tile, for every layer;
//   * its slice of W_l — forward W_l[16w + x][:], backward W_l[:][16w + x] — lives in 32 VGPRs and is the A operand of
//     v_mfma_f32_16x16x4_f32; the slice of layer l+1 is fetched (L2 hits: the weights are shared by all workgroups) into
//     a second register set under the MFMAs of layer l.  No weight ever passes through LDS, no barrier guards one;
//   * at E ~ 8.7k rows every CU holds ~34 rows and still has to stream all 8 x 64 KB of weights: the weight stream, not
//     the MFMA time, is the floor.  Read straight from the row-major W, a wave's operand load touches 16 cache lines per
//     16-lane pass (lane x = row 16w + x of W) — an ablation put that stream at 10 of 49.5 us.  k_chain_pack therefore
//     re-lays every weight once per step in operand order, Wf[w][j][lane][c] = W[16w + x][16j + 4q + c] (forward) and
//     Wb[w][j][lane][c] = W[16j + 4q + c][16w + x] (backward; lane = x + 16q), so each of a wave's 8 loads per layer is one
//     contiguous kilobyte;
//   * the product is formed TRANSPOSED, D[channel][row] = sum_k W[channel][k] X[row][k]: the D layout of the 16x16 MFMA
//     (lane (x, q): column x, rows 4q .. 4q + 3) then gives lane (x, q) four CONSECUTIVE channels 16w + 4q .. + 3 of row
//     x — one 16-byte store per output, and the SAME lane owns the same (row, channels) slot in every layer, so bias,
//     activation, external residuals, the skip tile (registers) and in the backward the running gradient are all
//     lane-local: no output scratch, no transpose through LDS;
//   * only the activation tile (the B operand all waves share) goes through LDS, double buffered: ONE barrier per layer;
//   * the reduction index is permuted identically for both operands (lane group q supplies k = 16j + 4q + c for
//     instruction c of group j) so each operand fragment of 4 MFMAs is one 16-byte read; LDS pitch 136 floats keeps the
//     four 16-lane groups of a ds_read_b128 on distinct banks ((2x + q) mod 16 is a bijection on every group).
// MFMA time per layer and 16-row block: 32 instructions x 32 cycles per wave, two waves per SIMD = 0.85 us; the row
// tile is chosen per launch so that the grid fills the CUs once (chainr_row_blocks).
#include "dense_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CRP 136          // LDS row pitch of the activation tile (floats)
#define CRT 512          // threads per workgroup (8 waves)

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
// One layer on the tile.  wc: this layer's weight slice (registers); the next layer's is loaded first and only waited
// for after the MFMAs, where it replaces wc — so no MFMA ever waits on a load issued in the same layer.  FULLK: K == 128
// (every layer but possibly the first).  The body is BRANCH-FREE apart from the row mask of the stores: optional
// operands (bias, external residual, skip tile, saved pre-activation) are loaded unconditionally from a valid address
// and selected — a load under a branch would make the outstanding-load count unknowable to the compiler and turn every
// later s_waitcnt into a wait for everything.  Activations: swish or none (dig3d_chain_* rejects others).
__device__ __forceinline__ float4 f4sel(bool c, float4 a, float4 b) {
  return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}
// swf = 1 (swish) or 0 (identity), blended arithmetically — exact in both cases (1*s + 0 = s, 0*s + 1 = 1) — because a
// select on a wave-uniform condition is compiled to a branch around every element
__device__ __forceinline__ float swish_or_id(float z, float swf) { return z * (swf * fast_sigmoid(z) + (1.0f - swf)); }
__device__ __forceinline__ float dswish_or_one(float z, float swf) {
  const float s = fast_sigmoid(z);
  return swf * (s * (1.0f + z * (1.0f - s))) + (1.0f - swf);
}
// (act', act'') of swish (swf = 1) or of the identity (swf = 0: 1, 0) — the formulas of dense_common.h:act_d12
__device__ __forceinline__ void d12_swish_or_id(float z, float swf, float& d1, float& d2) {
  const float s = fast_sigmoid(z);
  d1 = swf * (s * (1.0f + z * (1.0f - s))) + (1.0f - swf);
  d2 = swf * (s * (1.0f - s) * (2.0f + z * (1.0f - 2.0f * s)));
}

// DD: the second-order pass of the energy_and_force route (dig3d_chainp_dd, the backward of the input-gradient recursion):
// same products in the same layer order, but the epilogue is  y = t act'(Z0_l),  z = t G0_l act''(Z0_l)  with the saved
// pre-activation Z0_l and the saved total gradient G0_l of the first backward pass (no bias); residual handling unchanged.
// FK (with DD): the lin_kj layer of the FRONT's second-order pass (k_front_dd): with t = u W_kj^T, gm / v = d.fk_gm / d.fk_v,
// rb = d.mul[l], z = Z0_l (a = swish(z)):   c_ga = t a'(z),   HZ = t (gm rb) a''(z) + v gm a'(z),   d_rb = c_ga gm,
// y = c_ga rb + v a  (the gradient w.r.t. gm: input tile of the lin_down layer and X operand of its weight gradient).
template <int RB, bool FULLK, bool DD, int FK = 0>
__device__ __forceinline__ void chainr_fwd_layer(const ChainDesc& d, int l, int nl, int M, int m0, int wave, int x, int q,
                                                 float4 (&wc)[8], float4 (&skip)[RB], const float* __restrict__ sIn,
                                                 float* __restrict__ sOut) {
  const int cq = 16 * wave + 4 * q;
  const int K = d.K[l], res = d.res[l], N = d.N[l];
  const float sw = d.act[l] == ACT_SWISH ? 1.0f : 0.f;
  const bool save = d.save[l] != 0;
  const bool live = 16 * wave < N;                 // a layer with fewer than 128 outputs: the other waves only keep step
  float4 wn[8];
  {                                                // next layer's packed slice (the last layer re-reads its own: unused)
    const float* __restrict__ p = d.W[l + 1 < nl ? l + 1 : l] + (wave * 8 * 64 + (x + 16 * q)) * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) wn[j] = *(const float4*)(p + j * 256);
  }
  const float* __restrict__ rx = d.resext[l];
  const bool ext = res == 1 && rx != nullptr;
  float4 rv[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int m = min(m0 + 16 * rb + x, M - 1);
    const float* rp = (ext && live) ? rx + (int64_t)m * 128 + cq : d.W[l];
    rv[rb] = *(const float4*)rp;
  }
  const float* __restrict__ mx = d.mul[l];
  const bool hasm = mx != nullptr;
  float4 mv[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int m = min(m0 + 16 * rb + x, M - 1);
    mv[rb] = *(const float4*)((hasm && live) ? mx + (int64_t)m * 128 + cq : d.W[l]);
  }
  const bool hasb = d.bias[l] != nullptr;
  float4 bv = *(const float4*)((hasb && live) ? d.bias[l] + cq : d.W[l]);
  bv = f4sel(hasb, bv, make_float4(0.f, 0.f, 0.f, 0.f));
  float4 z0v[RB], g0v[RB], vv[RB];
  if (DD) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      // (row pitch N: the front's lin_down layer has N < 128 outputs; waves without output columns read row 0)
      const int64_t o = live ? (int64_t)min(m0 + 16 * rb + x, M - 1) * N + cq : 0;
      z0v[rb] = *(const float4*)(d.Z0[l] + o);
      g0v[rb] = *(const float4*)((FK ? d.fk_gm : d.G0[l]) + o);
      if (FK) vv[rb] = *(const float4*)((d.fk_v ? d.fk_v : d.fk_gm) + o);
    }
  }
  const float hasv = (FK && d.fk_v) ? 1.0f : 0.f;
  f32x4 acc[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* pa = sIn + x * CRP + 4 * q;
  if (!live) {
  } else if (FULLK) {
    float4 xb[2][RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) xb[0][rb] = *(const float4*)(pa + (16 * rb) * CRP);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j + 1 < 8) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) xb[(j + 1) & 1][rb] = *(const float4*)(pa + (16 * rb) * CRP + 16 * (j + 1));
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma4(wc[j].x, xb[j & 1][rb].x, acc[rb]);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma4(wc[j].y, xb[j & 1][rb].y, acc[rb]);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma4(wc[j].z, xb[j & 1][rb].z, acc[rb]);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma4(wc[j].w, xb[j & 1][rb].w, acc[rb]);
    }
  } else {
    const int nj = (K + 15) >> 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < nj) {
        float4 xb[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) xb[rb] = *(const float4*)(pa + (16 * rb) * CRP + 16 * j);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma4(wc[j].x, xb[rb].x, acc[rb]);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma4(wc[j].y, xb[rb].y, acc[rb]);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma4(wc[j].z, xb[rb].z, acc[rb]);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma4(wc[j].w, xb[rb].w, acc[rb]);
      }
    }
  }
  // the weight registers change hands BEFORE the stores below: the loads they wait for are a whole MFMA section old,
  // whereas a wait placed after the stores would also wait for the stores (one in-order counter on gfx9)
#pragma unroll
  for (int j = 0; j < 8; ++j) wc[j] = wn[j];
  float* __restrict__ Yo = d.Y[l];
  float* __restrict__ Zo = d.Z[l] ? d.Z[l] : Yo;   // no Z wanted: the value is overwritten by Y right behind it
  const bool skp = res == 2;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int r = 16 * rb + x;
    float4 z = make_float4(acc[rb][0] + bv.x, acc[rb][1] + bv.y, acc[rb][2] + bv.z, acc[rb][3] + bv.w);
    float4 y;
    if (DD && FK) {
      const float4 t = z, gm = g0v[rb], rbv = mv[rb];
      const float4 v = make_float4(vv[rb].x * hasv, vv[rb].y * hasv, vv[rb].z * hasv, vv[rb].w * hasv);
      float4 drb;
#define FK_ONE(c)                                                                  \
      {                                                                            \
        const float zz = z0v[rb].c, sg = fast_sigmoid(zz), a = zz * sg;            \
        const float d1 = sg * (1.0f + zz * (1.0f - sg));                           \
        const float d2 = sg * (1.0f - sg) * (2.0f + zz * (1.0f - 2.0f * sg));      \
        const float cga = t.c * d1;                                                \
        z.c = t.c * (gm.c * rbv.c) * d2 + v.c * gm.c * d1;                         \
        drb.c = cga * gm.c;                                                        \
        y.c = cga * rbv.c + v.c * a;                                               \
      }
      FK_ONE(x) FK_ONE(y) FK_ONE(z) FK_ONE(w)
#undef FK_ONE
      if (live) *(float4*)(d.fk_drb + (int64_t)min(m0 + r, M - 1) * N + cq) = drb;
    } else if (DD) {
      const float4 t = z;
      float d1, d2;
      d12_swish_or_id(z0v[rb].x, sw, d1, d2); y.x = t.x * d1; z.x = t.x * g0v[rb].x * d2;
      d12_swish_or_id(z0v[rb].y, sw, d1, d2); y.y = t.y * d1; z.y = t.y * g0v[rb].y * d2;
      d12_swish_or_id(z0v[rb].z, sw, d1, d2); y.z = t.z * d1; z.z = t.z * g0v[rb].z * d2;
      d12_swish_or_id(z0v[rb].w, sw, d1, d2); y.w = t.w * d1; z.w = t.w * g0v[rb].w * d2;
    } else {
      y = make_float4(swish_or_id(z.x, sw), swish_or_id(z.y, sw), swish_or_id(z.z, sw), swish_or_id(z.w, sw));
    }
    if (!FK) y = f4sel(hasm, make_float4(y.x * mv[rb].x, y.y * mv[rb].y, y.z * mv[rb].z, y.w * mv[rb].w), y);
    const float4 add = f4sel(ext, rv[rb], skip[rb]);
    y = f4sel(ext || skp, f4add(add, y), y);
    // rows beyond M are copies of row M - 1 (clamped loads everywhere), so their results are too: the stores are
    // unconditional to the clamped row — no lane mask, every wait count known to the compiler (`live` is wave-uniform
    // and false only for the upper waves of a layer with fewer than 128 outputs)
    const int64_t o = (int64_t)min(m0 + r, M - 1) * N + cq;
    if (live) {
      *(float4*)(Zo + o) = z;
      *(float4*)(Yo + o) = y;
    }
    if (live && sOut) *(float4*)(sOut + r * CRP + cq) = y;       // input tile of a later layer
    skip[rb] = f4sel(save, y, skip[rb]);
  }
  __syncthreads();        // sOut complete; every wave is done reading sIn (a later layer overwrites it)
}

template <int RB, bool DD = false>
__global__ void __launch_bounds__(CRT) k_chainr_fwd(const float* __restrict__ X0, int M, ChainDesc d) {
  extern __shared__ float csm[];
  constexpr int R = 16 * RB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, x = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.x * R;
  float* sA = csm;
  float* sB = csm + R * CRP;
  const int nl = d.nl;
  const int K0 = d.K[0];
  {                                                // stage the input tile (zero beyond K_0 up to the next multiple of 16)
    const int K0p = (K0 + 15) & ~15;
    const int tc = (tid & 31) * 4, tr = tid >> 5;
#pragma unroll
    for (int it = 0; it < RB; ++it) {
      const int r = tr + 16 * it, m = m0 + r;
      if (tc < K0p) {                              // rows beyond M: copies of row M - 1 (see the stores of a layer)
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tc < K0) v = *(const float4*)(X0 + (int64_t)min(m, M - 1) * K0 + tc);
        *(float4*)(sA + r * CRP + tc) = v;
      }
    }
  }
  float4 wc[8];
  {                                                // packed W_0 slice (zero beyond K_0: k_chain_pack)
    const float* __restrict__ p = d.W[0] + (wave * 8 * 64 + lane) * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) wc[j] = *(const float4*)(p + j * 256);
  }
  float4 skip[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) skip[rb] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();        // input tile staged
  auto buf = [&](int b) -> float* { return b < 0 ? nullptr : (b ? sB : sA); };
  if (K0 == 128) chainr_fwd_layer<RB, true, DD>(d, 0, nl, M, m0, wave, x, q, wc, skip, sA, buf(d.outbuf[0]));
  else chainr_fwd_layer<RB, false, DD>(d, 0, nl, M, m0, wave, x, q, wc, skip, sA, buf(d.outbuf[0]));
  for (int l = 1; l < nl; ++l)
    chainr_fwd_layer<RB, true, DD>(d, l, nl, M, m0, wave, x, q, wc, skip, buf(d.inbuf[l]), buf(d.outbuf[l]));
}

// The second-order pass of the FRONT (energy_and_force: the backward of k_front_bwd w.r.t. its differentiable inputs), the
// same three products in the same order as the forward on the tile u = (gradient w.r.t. gx1):
//   lin_ji:   t = u W_ji^T   ->  d g_xji = t a'(z_ji),  HZ_ji = t g_xji a''(z_ji)                      (the chain's DD epilogue)
//   lin_kj:   t = u W_kj^T   ->  HZ_kj, d rb, c_gm                                                      (FK epilogue above)
//   lin_down: t = c_gm W_d^T ->  d g_xd = t a'(z_d),   HZ_d = t g_xd a''(z_d)
// d: a 3-layer ChainDesc (W packed forward slices; Z0 = saved pre-activations; G0 = (g_xji, -, g_xd); mul[1] = rb; Y = (d g_xji,
// c_gm, d g_xd); Z = (HZ_ji, HZ_kj, HZ_d); N = (128, 128, ND)).  The weight gradients of this pass are dW_ji = GZ_ji^T u,
// dW_kj = GZ_kj^T u, dW_d = GZ_d^T c_gm with the GZ written by k_front_bwd (dig3d_chain_wgrad_n).
template <int RB>
__global__ void __launch_bounds__(CRT) k_front_dd(const float* __restrict__ U0, int M, ChainDesc d) {
  extern __shared__ float csm[];
  constexpr int R = 16 * RB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, x = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.x * R;
  float* sA = csm;
  float* sB = csm + R * CRP;
  {
    const int tc = (tid & 31) * 4, tr = tid >> 5;
#pragma unroll
    for (int it = 0; it < RB; ++it) {
      const int r = tr + 16 * it;
      *(float4*)(sA + r * CRP + tc) = *(const float4*)(U0 + (int64_t)min(m0 + r, M - 1) * 128 + tc);
    }
  }
  float4 wc[8];
  {
    const float* __restrict__ p = d.W[0] + (wave * 8 * 64 + lane) * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) wc[j] = *(const float4*)(p + j * 256);
  }
  float4 skip[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) skip[rb] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  chainr_fwd_layer<RB, true, true, 0>(d, 0, 3, M, m0, wave, x, q, wc, skip, sA, nullptr);
  chainr_fwd_layer<RB, true, true, 1>(d, 1, 3, M, m0, wave, x, q, wc, skip, sA, sB);
  chainr_fwd_layer<RB, true, true, 0>(d, 2, 3, M, m0, wave, x, q, wc, skip, sB, nullptr);
}

// ------------------------------------------------------------------------------------------------------------------
// backward (input-gradient recursion; the weight gradients are dense.hip:k_chain_wgrad over the GZ_l written here)
// ------------------------------------------------------------------------------------------------------------------
// packed backward slice Wb[w][j][lane][c] = W_l[16j + 4q + c][16w + x] (zero for columns >= K_l): the A operand of MFMA
// (j, c), 8 contiguous kilobyte loads per wave and layer
__device__ __forceinline__ void chainr_bwd_fetch_w(const ChainBwdDesc& d, int l, int wave, int lane, float (&w)[32]) {
  const float* __restrict__ p = d.W[l] + (wave * 8 * 64 + lane) * 4;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float4 v = *(const float4*)(p + j * 256);
    w[4 * j + 0] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
  }
}

template <int RB>
__global__ void __launch_bounds__(CRT) k_chainr_bwd(const float* __restrict__ gout, int M, ChainBwdDesc d,
                                                     float* __restrict__ gx0) {
  extern __shared__ float csm[];
  constexpr int R = 16 * RB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, x = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.x * R;
  const int k0 = 16 * wave, cq = k0 + 4 * q;       // this lane's four channels (of g_l and of g_{l-1} alike)
  const int nl = d.nl;
  int64_t orow[RB];                                // clamped row offsets of this lane's slots (loads are unconditional)
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) orow[rb] = (int64_t)min(m0 + 16 * rb + x, M - 1) * 128 + cq;
  float wc[32];
  float4 g[RB], skip[RB], rz[RB], ra[RB];
  // optional per-row operands of layer l, from a valid address either way (see chainr_fwd_layer): the saved
  // pre-activation (absent when the layer has no activation) and the gradient that reached Z_l directly (force path)
  auto fetch_rows = [&](int l, float4 (&z)[RB], float4 (&a)[RB]) {
    const float* __restrict__ Z = d.Z[l] ? d.Z[l] : gout;
    const float* __restrict__ A = d.gzadd[l] ? d.gzadd[l] : gout;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) z[rb] = *(const float4*)(Z + orow[rb]);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) a[rb] = *(const float4*)(A + (d.gzadd[l] ? orow[rb] : (int64_t)cq));
  };
  chainr_bwd_fetch_w(d, nl - 1, wave, lane, wc);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    g[rb] = *(const float4*)(gout + orow[rb]);
    skip[rb] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  fetch_rows(nl - 1, rz, ra);
  bool pending = false;                            // `skip` holds a gradient for the most recent saved tile (uniform)
  for (int l = nl - 1; l >= 0; --l) {
    float* __restrict__ sG = csm + ((l & 1) ? R * CRP : 0);
    const int res = d.res[l], K = d.K[l];
    const float sw = (d.act[l] == ACT_SWISH && d.Z[l] != nullptr) ? 1.0f : 0.f;
    const bool save = d.save[l] != 0;
    float* __restrict__ GZ = d.GZ[l];
    float* __restrict__ gr = (res == 1 && d.gres[l]) ? d.gres[l] : GZ;     // not wanted: overwritten by GZ right behind
    float* __restrict__ Gt = d.G[l] ? d.G[l] : GZ;
    const bool zadd = d.gzadd[l] != nullptr;
    const bool take = save && pending, put = res == 2, acc_put = pending && !save;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int r = 16 * rb + x;
      float4 gg = g[rb];
      gg = f4sel(take, f4add(skip[rb], gg), gg);                 // this layer's output was the skip tile
      skip[rb] = f4sel(put, f4sel(acc_put, f4add(skip[rb], gg), gg), skip[rb]);
      float4 gz = make_float4(gg.x * dswish_or_one(rz[rb].x, sw), gg.y * dswish_or_one(rz[rb].y, sw),
                              gg.z * dswish_or_one(rz[rb].z, sw), gg.w * dswish_or_one(rz[rb].w, sw));
      gz = f4sel(zadd, f4add(gz, ra[rb]), gz);
      // rows beyond M are copies of row M - 1 (clamped loads): unconditional stores to the clamped row, no lane mask
      *(float4*)(gr + orow[rb]) = gg;
      *(float4*)(Gt + orow[rb]) = gg;
      *(float4*)(GZ + orow[rb]) = gz;
      *(float4*)(sG + r * CRP + cq) = gz;
    }
    pending = put ? true : (save ? false : pending);
    __syncthreads();      // gZ_l tile complete (the other buffer is free for layer l - 1: everyone is past its MFMAs)
    // operands of layer l - 1 (layer 0 re-reads its own): issued now, consumed after this layer's MFMAs
    float wn[32];
    float4 rzn[RB], ran[RB];
    const int ln = l > 0 ? l - 1 : 0;
    chainr_bwd_fetch_w(d, ln, wave, lane, wn);
    fetch_rows(ln, rzn, ran);
    const bool live = k0 < K;                      // K_0 may be < 128: the other waves have no output columns
    f32x4 acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (live) {
      const float* pa = sG + x * CRP + 4 * q;
      float4 xb[2][RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) xb[0][rb] = *(const float4*)(pa + (16 * rb) * CRP);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j + 1 < 8) {
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) xb[(j + 1) & 1][rb] = *(const float4*)(pa + (16 * rb) * CRP + 16 * (j + 1));
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma4(wc[4 * j + 0], xb[j & 1][rb].x, acc[rb]);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma4(wc[4 * j + 1], xb[j & 1][rb].y, acc[rb]);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma4(wc[4 * j + 2], xb[j & 1][rb].z, acc[rb]);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma4(wc[4 * j + 3], xb[j & 1][rb].w, acc[rb]);
      }
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) wc[i] = wn[i];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) rz[rb] = rzn[rb], ra[rb] = ran[rb];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) g[rb] = make_float4(acc[rb][0], acc[rb][1], acc[rb][2], acc[rb][3]);
    if (l == 0 && live) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const int m = min(m0 + 16 * rb + x, M - 1);
        if (cq < K) *(float4*)(gx0 + (int64_t)m * K + cq) = g[rb];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The FRONT of an interaction block (spherenet.py:150-163, dimenetpp.py:130-145):
//     x_ji = swish(lin_ji(x1)),   t = swish(lin_kj(x1)) * rb,   xd = swish(lin_down(t))          (rb = lin_rbf2(lin_rbf1(rbf)))
// Forward = a 3-layer program of k_chainr_fwd (both branches read the x1 tile, the product with the radial projection is
// the multiplicative operand of the epilogue, lin_down has ND < 128 outputs).  Backward = this kernel: the three
// input-gradient products on one row tile, the gradient of t, of rb and of x1 never leaving the CU in between, and the
// other gradients that reach x1 (the chain's skip connection, the readout) added in its last epilogue.  Replaces three
// merged dgrad+wgrad launches, three framework multiplies and the framework additions on x1 (r03a profile: 13 x 19.4 us +
// 14 + 13 elementwise launches per step); the weight gradients of the three layers are one k_chain_wgrad launch.
// ------------------------------------------------------------------------------------------------------------------
struct FrontBwdDesc {
  const float* Wd;  const float* Wkj;  const float* Wji;   // packed backward slices (k_chain_pack Wb format)
  const float* Zd;  const float* Zkj;  const float* Zji;   // saved pre-activations [M,ND], [M,128], [M,128]
  const float* rb;                                          // [M,128]
  const float* gxd; const float* gxji;                      // incoming gradients [M,ND], [M,128]
  const float* gadd0; const float* gadd1;                   // further gradients of x1 [M,128], or null
  float* GZd; float* GZkj; float* GZji;                     // out: pre-activation gradients (operands of the weight gradients)
  float* grb; float* gx1;                                   // out [M,128]
  int ND;
  // energy_and_force (optional, null otherwise): Gm receives the gradient that reached t = swish(z_kj) * rb (the second-order
  // pass needs it); gzaddD / gzaddKj / gzaddJi are the act'' terms of that pass, added to the pre-activation gradients
  float* Gm;
  const float* gzaddD; const float* gzaddKj; const float* gzaddJi;
  const float* grbadd;                                      // a further gradient of rb [M,128] (added to grb), or null
};

template <int RB>
__device__ __forceinline__ void front_mma(const float (&w)[32], const float* __restrict__ sG, int x, int q, int nj,
                                          f32x4 (&acc)[RB]) {
  const float* pa = sG + x * CRP + 4 * q;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j < nj) {
      float4 xb[RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) xb[rb] = *(const float4*)(pa + (16 * rb) * CRP + 16 * j);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma4(w[4 * j + 0], xb[rb].x, acc[rb]);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma4(w[4 * j + 1], xb[rb].y, acc[rb]);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma4(w[4 * j + 2], xb[rb].z, acc[rb]);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma4(w[4 * j + 3], xb[rb].w, acc[rb]);
    }
  }
}

__device__ __forceinline__ void front_fetch_w(const float* __restrict__ Wb, int wave, int lane, float (&w)[32]) {
  const float* __restrict__ p = Wb + (wave * 8 * 64 + lane) * 4;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float4 v = *(const float4*)(p + j * 256);
    w[4 * j + 0] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
  }
}

template <int RB>
__global__ void __launch_bounds__(CRT) k_front_bwd(int M, FrontBwdDesc d) {
  extern __shared__ float csm[];
  constexpr int R = 16 * RB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, x = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.x * R;
  const int cq = 16 * wave + 4 * q;
  const int ND = d.ND;
  const bool liveD = 16 * wave < ND;               // this wave owns columns of the lin_down output
  float* sA = csm;
  float* sB = csm + R * CRP;
  int64_t o128[RB], oD[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int m = min(m0 + 16 * rb + x, M - 1);    // rows beyond M: copies of row M - 1 (unconditional loads and stores)
    o128[rb] = (int64_t)m * 128 + cq;
    oD[rb] = liveD ? (int64_t)m * ND + cq : 0;
  }
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float wa[32], wb[32];
  front_fetch_w(d.Wd, wave, lane, wa);
  // ---- step 1: lin_down.  gz_d = g_xd * swish'(z_d);  g_t = gz_d W_down
  f32x4 acc[RB];
  {
    float4 gd[RB], zd[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      gd[rb] = *(const float4*)(d.gxd + oD[rb]);
      zd[rb] = *(const float4*)(d.Zd + oD[rb]);
    }
    front_fetch_w(d.Wkj, wave, lane, wb);          // consumed in step 2
    const bool hzd = d.gzaddD != nullptr;
    float4 ad[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) ad[rb] = *(const float4*)(hzd ? d.gzaddD + oD[rb] : d.Wd);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      float4 gz = make_float4(gd[rb].x * dswish_or_one(zd[rb].x, 1.f), gd[rb].y * dswish_or_one(zd[rb].y, 1.f),
                              gd[rb].z * dswish_or_one(zd[rb].z, 1.f), gd[rb].w * dswish_or_one(zd[rb].w, 1.f));
      gz = f4sel(hzd, f4add(gz, ad[rb]), gz);
      if (liveD) {
        *(float4*)(d.GZd + oD[rb]) = gz;
        *(float4*)(sA + (16 * rb + x) * CRP + cq) = gz;
      }
    }
  }
  __syncthreads();
  float4 zk[RB], rv[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    zk[rb] = *(const float4*)(d.Zkj + o128[rb]);
    rv[rb] = *(const float4*)(d.rb + o128[rb]);
  }
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
  front_mma<RB>(wa, sA, x, q, ND >> 4, acc);       // reduction over the ND outputs of lin_down
  // ---- step 2: the product t = swish(z_kj) * rb and lin_kj.  g_rb = g_t swish(z_kj);  gz_kj = g_t rb swish'(z_kj)
  front_fetch_w(d.Wji, wave, lane, wa);            // consumed in step 3 (wa is free: its MFMAs are issued)
  float4 zj[RB], gj[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    zj[rb] = *(const float4*)(d.Zji + o128[rb]);
    gj[rb] = *(const float4*)(d.gxji + o128[rb]);
  }
  const bool hzk = d.gzaddKj != nullptr, hzj = d.gzaddJi != nullptr, hrb = d.grbadd != nullptr;
  float4 ak[RB], aj[RB], ar[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    ak[rb] = *(const float4*)(hzk ? d.gzaddKj + o128[rb] : d.Wji);
    aj[rb] = *(const float4*)(hzj ? d.gzaddJi + o128[rb] : d.Wji);
    ar[rb] = *(const float4*)(hrb ? d.grbadd + o128[rb] : d.Wji);
  }
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const float4 gt = make_float4(acc[rb][0], acc[rb][1], acc[rb][2], acc[rb][3]);
    if (d.Gm) *(float4*)(d.Gm + o128[rb]) = gt;
    float4 grb, gz;
    {
      const float s0 = fast_sigmoid(zk[rb].x), s1 = fast_sigmoid(zk[rb].y), s2 = fast_sigmoid(zk[rb].z), s3 = fast_sigmoid(zk[rb].w);
      grb = make_float4(gt.x * (zk[rb].x * s0), gt.y * (zk[rb].y * s1), gt.z * (zk[rb].z * s2), gt.w * (zk[rb].w * s3));
      gz = make_float4((gt.x * rv[rb].x) * (s0 * (1.0f + zk[rb].x * (1.0f - s0))), (gt.y * rv[rb].y) * (s1 * (1.0f + zk[rb].y * (1.0f - s1))),
                       (gt.z * rv[rb].z) * (s2 * (1.0f + zk[rb].z * (1.0f - s2))), (gt.w * rv[rb].w) * (s3 * (1.0f + zk[rb].w * (1.0f - s3))));
    }
    gz = f4sel(hzk, f4add(gz, ak[rb]), gz);
    grb = f4sel(hrb, f4add(grb, ar[rb]), grb);
    *(float4*)(d.grb + o128[rb]) = grb;
    *(float4*)(d.GZkj + o128[rb]) = gz;
    *(float4*)(sB + (16 * rb + x) * CRP + cq) = gz;
  }
  __syncthreads();
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
  front_mma<RB>(wb, sB, x, q, 8, acc);             // g_x1 (first part) = gz_kj W_kj
  // ---- step 3: lin_ji.  gz_ji = g_xji * swish'(z_ji);  g_x1 += gz_ji W_ji  (same accumulators)
  const bool has0 = d.gadd0 != nullptr, has1 = d.gadd1 != nullptr;
  float4 a0[RB], a1[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    a0[rb] = *(const float4*)(has0 ? d.gadd0 + o128[rb] : d.Wji);
    a1[rb] = *(const float4*)(has1 ? d.gadd1 + o128[rb] : d.Wji);
  }
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    float4 gz = make_float4(gj[rb].x * dswish_or_one(zj[rb].x, 1.f), gj[rb].y * dswish_or_one(zj[rb].y, 1.f),
                            gj[rb].z * dswish_or_one(zj[rb].z, 1.f), gj[rb].w * dswish_or_one(zj[rb].w, 1.f));
    gz = f4sel(hzj, f4add(gz, aj[rb]), gz);
    *(float4*)(d.GZji + o128[rb]) = gz;
    *(float4*)(sA + (16 * rb + x) * CRP + cq) = gz;              // sA: every wave is past the step-1 products (barrier of step 2)
  }
  __syncthreads();
  front_mma<RB>(wa, sA, x, q, 8, acc);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    float4 g = make_float4(acc[rb][0], acc[rb][1], acc[rb][2], acc[rb][3]);
    g = f4sel(has0, f4add(g, a0[rb]), g);
    g = f4sel(has1, f4add(g, a1[rb]), g);
    *(float4*)(d.gx1 + o128[rb]) = g;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// weights in operand order (once per step and chain; 2 x 64 KB per layer)
// ------------------------------------------------------------------------------------------------------------------
#define PACK_MAX 64       // layers per pack launch: all chains and fronts of a model forward (4 x (3 + 8) for the default)
struct ChainPackDesc {
  const float* W[PACK_MAX];
  int K[PACK_MAX];
  int N[PACK_MAX];
};

// grid (16, nl): block (b, l) writes rows of 4 KB: thread t of 256 -> float4 slot s = b * 256 + t of 4096 per format;
// slot s = ((w * 8 + j) * 64 + lane): reads are 16-byte (forward format) / 4 strided dwords (backward format) — 128 KB
// per layer in total, L2-resident, once per step
__global__ void __launch_bounds__(256) k_chain_pack(ChainPackDesc d, float* __restrict__ Wf, float* __restrict__ Wb) {
  const int l = blockIdx.y;
  const int K = d.K[l], N = d.N[l];
  const float* __restrict__ W = d.W[l];
  const int s = blockIdx.x * 256 + threadIdx.x;
  const int lane = s & 63, j = (s >> 6) & 7, w = s >> 9, x = lane & 15, q = lane >> 4;
  const int kk = 16 * j + 4 * q;
  float4 f = make_float4(0.f, 0.f, 0.f, 0.f), b = f;
  if (kk < K && 16 * w + x < N) f = *(const float4*)(W + (int64_t)(16 * w + x) * K + kk);      // K % 4 == 0
  if (16 * w + x < K && kk < N) {                                                  // N % 4 == 0: rows kk .. kk + 3 < N
    const float* p = W + (int64_t)kk * K + 16 * w + x;
    b = make_float4(p[0], p[K], p[2 * (int64_t)K], p[3 * (int64_t)K]);
  }
  *(float4*)(Wf + (int64_t)l * 16384 + 4 * s) = f;
  *(float4*)(Wb + (int64_t)l * 16384 + 4 * s) = b;
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
static int chainr_cus() { return dig3d_num_cus(); }

// 16-row blocks per workgroup: the smallest tile whose grid passes over the CUs the fewest times (cost = passes x rows per
// tile); from ~4 passes on, the widest tile (each weight register load then serves 64 rows).
static int chainr_row_blocks(int M, int rbmax = 4) {
  const int cus = chainr_cus();
  const int t16 = (M + 15) / 16;
  if (t16 >= 16 * cus) return rbmax;
  int best = 1, best_cost = 1 << 30;
  for (int rb = 1; rb <= rbmax; ++rb) {
    const int blocks = (t16 + rb - 1) / rb;
    const int cost = ((blocks + cus - 1) / cus) * rb;
    if (cost < best_cost || (cost == best_cost && rb > best)) best = rb, best_cost = cost;
  }
  return best;
}

template <int RB, bool DD = false>
static int chainr_fwd_go(const float* X0, int M, const ChainDesc& d, hipStream_t st) {
  constexpr int R = 16 * RB;
  const size_t shm = sizeof(float) * 2 * R * CRP;
  static const bool attr_ok = hipFuncSetAttribute((const void*)k_chainr_fwd<RB, DD>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)shm) == hipSuccess;      // set once
  if (!attr_ok) return DIG3D_ERR_LAUNCH;
  hipLaunchKernelGGL((k_chainr_fwd<RB, DD>), dim3((M + R - 1) / R), dim3(CRT), shm, st, X0, M, d);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

static int chainr_dd_launch(const float* H0, int M, const ChainDesc& d, hipStream_t st) {
  switch (chainr_row_blocks(M, 3)) {                 // two more row operands per layer (Z0, G0): 3 row blocks
    case 1: return chainr_fwd_go<1, true>(H0, M, d, st);
    case 2: return chainr_fwd_go<2, true>(H0, M, d, st);
    default: return chainr_fwd_go<3, true>(H0, M, d, st);
  }
}

template <int RB>
static int chainr_bwd_go(const float* gout, int M, const ChainBwdDesc& d, float* gx0, hipStream_t st) {
  constexpr int R = 16 * RB;
  const size_t shm = sizeof(float) * 2 * R * CRP;
  static const bool attr_ok = hipFuncSetAttribute((const void*)k_chainr_bwd<RB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)shm) == hipSuccess;      // set once
  if (!attr_ok) return DIG3D_ERR_LAUNCH;
  hipLaunchKernelGGL(k_chainr_bwd<RB>, dim3((M + R - 1) / R), dim3(CRT), shm, st, gout, M, d, gx0);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

static int chainr_fwd_launch(const float* X0, int M, const ChainDesc& d, hipStream_t st) {
  switch (chainr_row_blocks(M, 4)) {
    case 1: return chainr_fwd_go<1>(X0, M, d, st);
    case 2: return chainr_fwd_go<2>(X0, M, d, st);
    case 3: return chainr_fwd_go<3>(X0, M, d, st);
    default: return chainr_fwd_go<4>(X0, M, d, st);
  }
}

static int chainr_bwd_launch(const float* gout, int M, const ChainBwdDesc& d, float* gx0, hipStream_t st) {
  switch (chainr_row_blocks(M, 3)) {                 // the backward keeps two more row operands: 3 row blocks fit the registers
    case 1: return chainr_bwd_go<1>(gout, M, d, gx0, st);
    case 2: return chainr_bwd_go<2>(gout, M, d, gx0, st);
    default: return chainr_bwd_go<3>(gout, M, d, gx0, st);
  }
}

extern "C" {

// Weights in MFMA operand order (see the file header): Wf, Wb float[nl * 16384] each.  W[l] [N[l], K[l]] row-major,
// K[l] <= 128 (multiple of 8), N[l] <= 128 (multiple of 16; N == NULL: 128 everywhere); missing rows / columns are zero.
int dig3d_chain_pack(int nl, const void* const* W, const int* K, const int* N, float* Wf, float* Wb, void* stream) {
  DIG3D_ENTER();
  if (nl < 1 || nl > PACK_MAX || !W || !K || !Wf || !Wb || !al16(Wf) || !al16(Wb)) return DIG3D_ERR_ARG;
  ChainPackDesc d;
  for (int l = 0; l < nl; ++l) {
    const int n = N ? N[l] : 128;
    if (!W[l] || !al16(W[l]) || K[l] <= 0 || K[l] > 128 || (K[l] & 7) || n <= 0 || n > 128 || (n & 15)) return DIG3D_ERR_ARG;
    d.W[l] = (const float*)W[l];
    d.K[l] = K[l];
    d.N[l] = n;
  }
  hipLaunchKernelGGL(k_chain_pack, dim3(16, nl), dim3(256), 0, (hipStream_t)stream, d, Wf, Wb);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// dig3d_chain_fwd on packed weights (Wf from dig3d_chain_pack); activations: none or swish.
int dig3d_chainp_fwd(const float* X0, int M, int nl, const float* Wf, const void* const* bias, const void* const* resext,
                     void* const* Z, void* const* Y, const int* K, const int* res, const int* save, const int* act,
                     void* stream) {
  DIG3D_ENTER();
  if (M < 0 || nl < 1 || nl > CH_MAX || !X0 || !Wf || !Y || !K || !res || !save || !act || !Z || !al16(X0) || !al16(Wf))
    return DIG3D_ERR_ARG;
  if (M == 0) return DIG3D_OK;
  ChainDesc d;
  for (int l = 0; l < nl; ++l) {
    if (!Y[l] || K[l] <= 0 || K[l] > 128 || (K[l] & 7) || (l > 0 && K[l] != 128)) return DIG3D_ERR_ARG;
    if (act[l] != ACT_NONE && act[l] != ACT_SWISH) return DIG3D_ERR_ARG;
    if (res[l] == 1 && !(resext && resext[l])) return DIG3D_ERR_ARG;
    if (res[l] == 2 && l == 0) return DIG3D_ERR_ARG;
    d.W[l] = Wf + (size_t)l * 16384;
    d.bias[l] = bias ? (const float*)bias[l] : nullptr;
    d.resext[l] = resext ? (const float*)resext[l] : nullptr;
    d.Z[l] = (float*)Z[l];
    d.Y[l] = (float*)Y[l];
    d.K[l] = K[l];
    d.res[l] = res[l];
    d.save[l] = save[l];
    d.act[l] = act[l];
    d.Z0[l] = d.G0[l] = nullptr;
    d.N[l] = 128;
    d.mul[l] = nullptr;
    d.inbuf[l] = l & 1;
    d.outbuf[l] = (l + 1) & 1;
    if (!al16(d.bias[l]) || !al16(d.resext[l]) || !al16(d.Z[l]) || !al16(d.Y[l])) return DIG3D_ERR_ARG;
  }
  d.nl = nl;
  return chainr_fwd_launch(X0, M, d, (hipStream_t)stream);
}

// dig3d_chain_dd on packed weights (Wf from dig3d_chain_pack): the second-order pass of the energy_and_force route — the
// backward of dig3d_chainp_bwd w.r.t. (gout, Z) — on the register-resident kernel (r05: it was the last pass of the chain
// still on the round-2 LDS kernel, k_chain_fwd<true>, 106 us per chain at E ~ 9.4k against 53 for the forward).
// Arguments as dig3d_chain_dd with Wf in place of the row-major weights.
int dig3d_chainp_dd(const float* H0, int M, int nl, const float* Wf, const void* const* Z0, const void* const* G0,
                    const void* const* ggres, void* const* HZ, void* const* U, const int* K, const int* res, const int* save,
                    const int* act, void* stream) {
  DIG3D_ENTER();
  if (M < 0 || nl < 1 || nl > CH_MAX || !H0 || !Wf || !Z0 || !G0 || !HZ || !U || !K || !res || !save || !act || !al16(H0) ||
      !al16(Wf))
    return DIG3D_ERR_ARG;
  if (M == 0) return DIG3D_OK;
  ChainDesc d;
  for (int l = 0; l < nl; ++l) {
    if (!U[l] || !HZ[l] || !Z0[l] || !G0[l] || K[l] <= 0 || K[l] > 128 || (K[l] & 7) || (l > 0 && K[l] != 128)) return DIG3D_ERR_ARG;
    if (act[l] != ACT_NONE && act[l] != ACT_SWISH) return DIG3D_ERR_ARG;
    if (res[l] == 2 && l == 0) return DIG3D_ERR_ARG;
    d.W[l] = Wf + (size_t)l * 16384;
    d.bias[l] = nullptr;
    d.resext[l] = ggres ? (const float*)ggres[l] : nullptr;     // gradient w.r.t. the gres output of layer l, may be absent
    d.Z[l] = (float*)HZ[l];
    d.Y[l] = (float*)U[l];
    d.K[l] = K[l];
    d.res[l] = res[l];
    d.save[l] = save[l];
    d.act[l] = act[l];
    d.Z0[l] = (const float*)Z0[l];
    d.G0[l] = (const float*)G0[l];
    d.N[l] = 128;
    d.mul[l] = nullptr;
    d.inbuf[l] = l & 1;
    d.outbuf[l] = (l + 1) & 1;
    if (!al16(d.resext[l]) || !al16(d.Z[l]) || !al16(d.Y[l]) || !al16(d.Z0[l]) || !al16(d.G0[l])) return DIG3D_ERR_ARG;
  }
  d.nl = nl;
  return chainr_dd_launch(H0, M, d, (hipStream_t)stream);
}

// dig3d_chain_bwd on packed weights (Wb from dig3d_chain_pack).
int dig3d_chainp_bwd(const float* gout, int M, int nl, const float* Wb, const void* const* Z, void* const* GZ,
                     void* const* gres, const int* K, const int* res, const int* save, const int* act, float* gx0,
                     void* const* G, const void* const* gz_add, void* stream) {
  DIG3D_ENTER();
  if (M < 0 || nl < 1 || nl > CH_MAX || !gout || !Wb || !Z || !GZ || !gres || !K || !res || !save || !act || !gx0)
    return DIG3D_ERR_ARG;
  if (M == 0) return DIG3D_OK;
  if (!al16(gout) || !al16(gx0) || !al16(Wb)) return DIG3D_ERR_ARG;
  ChainBwdDesc d;
  for (int l = 0; l < nl; ++l) {
    if (!GZ[l] || K[l] <= 0 || K[l] > 128 || (K[l] & 7) || (l > 0 && K[l] != 128)) return DIG3D_ERR_ARG;
    if (act[l] != ACT_NONE && (!Z[l] || act[l] != ACT_SWISH)) return DIG3D_ERR_ARG;
    if (res[l] == 1 && !gres[l]) return DIG3D_ERR_ARG;
    if (res[l] == 2 && l == 0) return DIG3D_ERR_ARG;
    if (!al16(Z[l]) || !al16(GZ[l]) || !al16(gres[l])) return DIG3D_ERR_ARG;
    d.W[l] = Wb + (size_t)l * 16384;
    d.Z[l] = (const float*)Z[l];
    d.GZ[l] = (float*)GZ[l];
    d.gres[l] = (float*)gres[l];
    d.G[l] = G ? (float*)G[l] : nullptr;
    d.gzadd[l] = gz_add ? (const float*)gz_add[l] : nullptr;
    if (!al16(d.G[l]) || !al16(d.gzadd[l])) return DIG3D_ERR_ARG;
    d.K[l] = K[l];
    d.res[l] = res[l];
    d.save[l] = save[l];
    d.act[l] = act[l];
  }
  d.nl = nl;
  return chainr_bwd_launch(gout, M, d, gx0, (hipStream_t)stream);
}


// Front of an interaction block, forward (see k_front_bwd's header).  Wf float[3 * 16384]: dig3d_chain_pack of
// (lin_ji.weight [128,128], lin_kj.weight [128,128], lin_down.weight [ND,128]) with N = (128, 128, ND).  Out: Zji, Xji,
// Zkj, T [M,128]; Zd, Xd [M,ND].  ND % 16 == 0, ND <= 128.
int dig3d_front_fwd(const float* x1, int M, const float* Wf, const float* b_ji, const float* b_kj, const float* rb,
                    float* Zji, float* Xji, float* Zkj, float* T, float* Zd, float* Xd, int ND, void* stream) {
  DIG3D_ENTER();
  if (M < 0 || !x1 || !Wf || !rb || !Zji || !Xji || !Zkj || !T || !Zd || !Xd || ND <= 0 || ND > 128 || (ND & 15))
    return DIG3D_ERR_ARG;
  if (!al16(x1) || !al16(Wf) || !al16(b_ji) || !al16(b_kj) || !al16(rb) || !al16(Zji) || !al16(Xji) || !al16(Zkj) ||
      !al16(T) || !al16(Zd) || !al16(Xd))
    return DIG3D_ERR_ARG;
  if (M == 0) return DIG3D_OK;
  ChainDesc d;
  const float* bias[3] = {b_ji, b_kj, nullptr};
  float* Z[3] = {Zji, Zkj, Zd};
  float* Y[3] = {Xji, T, Xd};
  const int N[3] = {128, 128, ND}, inb[3] = {0, 0, 1}, outb[3] = {-1, 1, -1};
  for (int l = 0; l < 3; ++l) {
    d.W[l] = Wf + (size_t)l * 16384;
    d.bias[l] = bias[l];
    d.resext[l] = nullptr;
    d.Z[l] = Z[l];
    d.Y[l] = Y[l];
    d.K[l] = 128;
    d.res[l] = 0;
    d.save[l] = 0;
    d.act[l] = ACT_SWISH;
    d.N[l] = N[l];
    d.mul[l] = l == 1 ? rb : nullptr;
    d.inbuf[l] = inb[l];
    d.outbuf[l] = outb[l];
    d.Z0[l] = d.G0[l] = nullptr;
  }
  d.nl = 3;
  return chainr_fwd_launch(x1, M, d, (hipStream_t)stream);
}

// Front of an interaction block, backward.  Wb float[3 * 16384] (same pack call).  In: the saved Zd [M,ND], Zkj, Zji,
// rb [M,128]; gxd [M,ND], gxji [M,128]; gadd0 / gadd1 [M,128] or NULL (other gradients reaching x1).  Out: GZd [M,ND],
// GZkj, GZji [M,128] (operands of dig3d_chain_wgrad_n with X = (x1, x1, T)), grb, gx1 [M,128].
int dig3d_front_bwd_add(int M, const float* Wb, const float* Zd, const float* Zkj, const float* Zji, const float* rb,
                        const float* gxd, const float* gxji, const float* gadd0, const float* gadd1, float* GZd, float* GZkj,
                        float* GZji, float* grb, float* gx1, int ND, float* Gm, const float* gzaddD, const float* gzaddKj,
                        const float* gzaddJi, const float* grb_add, void* stream);
int dig3d_front_bwd(int M, const float* Wb, const float* Zd, const float* Zkj, const float* Zji, const float* rb,
                    const float* gxd, const float* gxji, const float* gadd0, const float* gadd1, float* GZd, float* GZkj,
                    float* GZji, float* grb, float* gx1, int ND, float* Gm, const float* gzaddD, const float* gzaddKj,
                    const float* gzaddJi, void* stream) {
  return dig3d_front_bwd_add(M, Wb, Zd, Zkj, Zji, rb, gxd, gxji, gadd0, gadd1, GZd, GZkj, GZji, grb, gx1, ND, Gm, gzaddD, gzaddKj,
                             gzaddJi, nullptr, stream);
}
// the same with grb += grb_add [M,128] (NULL: none): the final pass of energy_and_force, where the front's own create_graph
// backward sends a second gradient to rb (dig_amd/diffops.py:_Front2)
int dig3d_front_bwd_add(int M, const float* Wb, const float* Zd, const float* Zkj, const float* Zji, const float* rb,
                        const float* gxd, const float* gxji, const float* gadd0, const float* gadd1, float* GZd, float* GZkj,
                        float* GZji, float* grb, float* gx1, int ND, float* Gm, const float* gzaddD, const float* gzaddKj,
                        const float* gzaddJi, const float* grb_add, void* stream) {
  DIG3D_ENTER();
  if (!al16(Gm) || !al16(gzaddD) || !al16(gzaddKj) || !al16(gzaddJi) || !al16(grb_add)) return DIG3D_ERR_ARG;
  if (M < 0 || !Wb || !Zd || !Zkj || !Zji || !rb || !gxd || !gxji || !GZd || !GZkj || !GZji || !grb || !gx1 || ND <= 0 ||
      ND > 128 || (ND & 15))
    return DIG3D_ERR_ARG;
  if (!al16(Wb) || !al16(Zd) || !al16(Zkj) || !al16(Zji) || !al16(rb) || !al16(gxd) || !al16(gxji) || !al16(gadd0) ||
      !al16(gadd1) || !al16(GZd) || !al16(GZkj) || !al16(GZji) || !al16(grb) || !al16(gx1))
    return DIG3D_ERR_ARG;
  if (M == 0) return DIG3D_OK;
  FrontBwdDesc d;
  d.Wji = Wb; d.Wkj = Wb + 16384; d.Wd = Wb + 2 * 16384;
  d.Zd = Zd; d.Zkj = Zkj; d.Zji = Zji; d.rb = rb; d.gxd = gxd; d.gxji = gxji; d.gadd0 = gadd0; d.gadd1 = gadd1;
  d.GZd = GZd; d.GZkj = GZkj; d.GZji = GZji; d.grb = grb; d.gx1 = gx1; d.ND = ND;
  d.Gm = Gm; d.gzaddD = gzaddD; d.gzaddKj = gzaddKj; d.gzaddJi = gzaddJi; d.grbadd = grb_add;
  hipStream_t st = (hipStream_t)stream;
#define FRONT_GO(RB_)                                                                                                   \
  {                                                                                                                     \
    const size_t shm = sizeof(float) * 2 * 16 * RB_ * CRP;                                                              \
    static const bool ok = hipFuncSetAttribute((const void*)k_front_bwd<RB_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                               (int)shm) == hipSuccess;                                                \
    if (!ok) return DIG3D_ERR_LAUNCH;                                                                                   \
    hipLaunchKernelGGL(k_front_bwd<RB_>, dim3((M + 16 * RB_ - 1) / (16 * RB_)), dim3(CRT), shm, st, M, d);              \
  }
  switch (chainr_row_blocks(M, 3)) {
    case 1: FRONT_GO(1) break;
    case 2: FRONT_GO(2) break;
    default: FRONT_GO(3) break;
  }
#undef FRONT_GO
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// Front of an interaction block, SECOND-order pass (energy_and_force: the backward of dig3d_front_bwd w.r.t. gxji, gxd, rb
// and the pre-activations; see k_front_dd).  In: U [M,128] = gradient w.r.t. gx1, V [M,128] (or NULL) = gradient w.r.t. grb;
// Wf (forward slices of the same pack call); the saved Zji, Zkj, Zd; rb; gxji, gxd (what dig3d_front_bwd received) and Gm
// (what it wrote).  Out: dgxji, HZji, HZkj, drb, Cgm [M,128]; dgxd, HZd [M,ND].  Weight gradients of the pass:
// dig3d_chain_wgrad_n over GZ = (GZji, GZkj, GZd) of dig3d_front_bwd with X = (U, U, Cgm).
int dig3d_front_dd(const float* U, const float* V, int M, const float* Wf, const float* Zji, const float* Zkj, const float* Zd,
                   const float* rb, const float* gxji, const float* gxd, const float* Gm, float* dgxji, float* HZji,
                   float* HZkj, float* drb, float* Cgm, float* dgxd, float* HZd, int ND, void* stream) {
  DIG3D_ENTER();
  if (M < 0 || !U || !Wf || !Zji || !Zkj || !Zd || !rb || !gxji || !gxd || !Gm || !dgxji || !HZji || !HZkj || !drb || !Cgm ||
      !dgxd || !HZd || ND <= 0 || ND > 128 || (ND & 15))
    return DIG3D_ERR_ARG;
  if (!al16(U) || !al16(V) || !al16(Wf) || !al16(Zji) || !al16(Zkj) || !al16(Zd) || !al16(rb) || !al16(gxji) || !al16(gxd) ||
      !al16(Gm) || !al16(dgxji) || !al16(HZji) || !al16(HZkj) || !al16(drb) || !al16(Cgm) || !al16(dgxd) || !al16(HZd))
    return DIG3D_ERR_ARG;
  if (M == 0) return DIG3D_OK;
  ChainDesc d;
  const float* Z0[3] = {Zji, Zkj, Zd};
  const float* G0[3] = {gxji, Gm, gxd};
  float* HZ[3] = {HZji, HZkj, HZd};
  float* Y[3] = {dgxji, Cgm, dgxd};
  const int N[3] = {128, 128, ND}, inb[3] = {0, 0, 1}, outb[3] = {-1, 1, -1};
  for (int l = 0; l < 3; ++l) {
    d.W[l] = Wf + (size_t)l * 16384;
    d.bias[l] = nullptr;
    d.resext[l] = nullptr;
    d.Z[l] = HZ[l];
    d.Y[l] = Y[l];
    d.K[l] = 128;
    d.res[l] = 0;
    d.save[l] = 0;
    d.act[l] = ACT_SWISH;
    d.N[l] = N[l];
    d.mul[l] = l == 1 ? rb : nullptr;
    d.inbuf[l] = inb[l];
    d.outbuf[l] = outb[l];
    d.Z0[l] = Z0[l];
    d.G0[l] = G0[l];
  }
  d.nl = 3;
  d.fk_gm = Gm; d.fk_v = V; d.fk_drb = drb;
  hipStream_t st = (hipStream_t)stream;
#define FDD_GO(RB_)                                                                                                      \
  {                                                                                                                     \
    const size_t shm = sizeof(float) * 2 * 16 * RB_ * CRP;                                                              \
    static const bool ok = hipFuncSetAttribute((const void*)k_front_dd<RB_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                               (int)shm) == hipSuccess;                                                \
    if (!ok) return DIG3D_ERR_LAUNCH;                                                                                   \
    hipLaunchKernelGGL(k_front_dd<RB_>, dim3((M + 16 * RB_ - 1) / (16 * RB_)), dim3(CRT), shm, st, U, M, d);            \
  }
  switch (chainr_row_blocks(M, 3)) {
    case 1: FDD_GO(1) break;
    case 2: FDD_GO(2) break;
    default: FDD_GO(3) break;
  }
#undef FDD_GO
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

}  // extern "C"
