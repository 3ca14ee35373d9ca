// dig3d dense layers — the hidden-channel Linears inside the interaction MLPs, on the f32 matrix cores.
//
// Reference call sites (all `y = act(F.linear(x, W, b))`, float32):
//   method/spherenet/spherenet.py:34-50 (ResidualLayer), :150-182 (lin_ji, lin_kj, lin_down, lin_up, lin),
//   :79-91 (init_e.lin), :209-216 (output block);  dimenetpp.py same lines;  schnet.py:29-59;  comenet.py:87-215.
// M = E or N rows (10^3..10^7), K, N in {64, 128, 256, 384}: at the reference's batch sizes these GEMMs are
// short and wide-K-less, and a general GEMM library runs them at a few TF (one 256x256 macro tile for a
// 128x128 weight gradient).  Here:
//   k_linear_fwd        Y = act(X W^T + b) (+ res), optionally also Z = X W^T + b for the backward
//   k_linear_bwd_input  gX = (gY * act'(Z)) W          (act' applied while staging, gZ never stored)
//   k_linear_bwd_weight gW = (gY * act'(Z))^T X, gb = column sums; split over row chunks, two-stage
//                       deterministic reduction (no atomics)
//   k_linear_bwd_both   the two gradients in one launch; _s variants: 32-row tiles for E ~ 10^4 rows
//   k_linear_pw         M >= ~5*10^4 rows: persistent wave-independent blocks, weight slice resident in LDS
//   k_chain_fwd / k_chain_bwd / k_chain_wgrad   the 8-layer residual block on LDS-resident row tiles (forward, input-
//                       gradient recursion, all weight gradients); k_chain_fwd<true> = the second-order pass
//   k_linear_dd, ACT_D2 epilogue, grouped variants, small-K kernels, k_reduce_many, k_adam_flat
// v_mfma_f32_32x32x2_f32: exact f32 (bitwise an fmaf chain), 157 TF peak.
#include "common.h"
#include <stdlib.h>

#include "dense_common.h"

__device__ __forceinline__ f32x16 zero16() {
  f32x16 v;
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = 0.f;
  return v;
}

// 4 consecutive floats of row `row` starting at column `col` of a row-major [nrows, ncols] matrix with leading
// dimension ld; zeros outside.  vec: ld % 4 == 0 and 16-byte aligned base (then col % 4 == 0 makes the
// address aligned); otherwise element-wise loads.
__device__ __forceinline__ float4 ld4(const float* __restrict__ src, int64_t ld, int row, int nrows, int col,
                                      int ncols, bool vec) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row < nrows && col < ncols) {
    const float* p = src + (int64_t)row * ld + col;
    if (vec && col + 3 < ncols) {
      v = *(const float4*)p;
    } else {
      v.x = p[0];
      if (col + 1 < ncols) v.y = p[1];
      if (col + 2 < ncols) v.z = p[2];
      if (col + 3 < ncols) v.w = p[3];
    }
  }
  return v;
}

__device__ __forceinline__ float4 gz4(float4 g, float4 z, int act) {
  if (act != ACT_NONE) {
    g.x *= act_bwd(z.x, act); g.y *= act_bwd(z.y, act); g.z *= act_bwd(z.z, act); g.w *= act_bwd(z.w, act);
  }
  return g;
}

// ------------------------------------------------------------------------------------------------
// Tiling (all three kernels): 512 threads = 8 waves, one v_mfma_f32_32x32x2_f32 accumulator tile (or two) per
// wave; operands staged through LDS in reduction chunks of DBK = 128 (row pitch 132 floats: (i*132) mod 64 =
// 4i, so the 16 lanes of a ds_read_b128 group hit 16 distinct 16-byte slots).  The global->register fetch of
// chunk c+1 is issued before the MFMAs of chunk c.  Results leave through an LDS transpose so that every
// global store is a 16-byte row segment (the row-per-lane dword stores of the MFMA layout were 40 % of the
// kernel at E ~ 10^4 rows).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 f4sum(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
#define NTH 512
#define DBK 128          // reduction chunk staged per iteration
#define DBKP 132         // LDS row pitch (floats)

// forward:  WN waves along N, WM = 8/WN along M;  BM = 32*WM rows x BN = 32*WN columns per block.
// The k pairing inside an MFMA is permuted (lane half h supplies k = 8q+4h+j for instruction j of group q)
// identically for A and B, so both operands are read with one ds_read_b128 per 4 MFMAs.
template <int WN>
__device__ __forceinline__ void linear_fwd_body(const float* __restrict__ X, const float* __restrict__ W,
                                                const float* __restrict__ bias, const float* __restrict__ res,
                                                int M, int K, int N, int act, float* __restrict__ Y,
                                                float* __restrict__ Z, float* __restrict__ smem, int bx, int by) {
  constexpr int WM = 8 / WN, BM = 32 * WM, BN = 32 * WN;
  constexpr int NA = BM / 16, NW = BN / 16;          // float4 per thread per chunk
  const bool rowscale = act == ACT_ROWSCALE;         // Y = res[m] * z, Z = res[m] (see ACT_ROWSCALE)
  const bool keepd = !rowscale && act < ACT_D2 && (act & ACT_KEEP_DERIV);   // Z receives act'(z) (ACT_KEEP_DERIV)
  if (rowscale) act = ACT_NONE;
  else if (act < ACT_D2) act &= 3;
  float* sA = smem;
  float* sW = smem + BM * DBKP;
  const int m0 = bx * BM, n0 = by * BN;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave / WN, wn = wave % WN, i = lane & 31, h = lane >> 5;
  const bool vec = (K & 3) == 0;
  const int tr = threadIdx.x >> 5, tc = (threadIdx.x & 31) * 4;   // 16 rows x 128 cols per pass
  float4 ra[NA], rw[NW];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int it = 0; it < NA; ++it) ra[it] = ld4(X, K, m0 + tr + 16 * it, M, k0 + tc, K, vec);
#pragma unroll
    for (int it = 0; it < NW; ++it) rw[it] = ld4(W, K, n0 + tr + 16 * it, N, k0 + tc, K, vec);
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < NA; ++it) *(float4*)(sA + (tr + 16 * it) * DBKP + tc) = ra[it];
#pragma unroll
    for (int it = 0; it < NW; ++it) *(float4*)(sW + (tr + 16 * it) * DBKP + tc) = rw[it];
  };
  f32x16 acc = zero16();
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += DBK) {
    commit();
    __syncthreads();
    if (k0 + DBK < K) fetch(k0 + DBK);
    const int kq = ((K - k0 < DBK ? K - k0 : DBK) + 7) >> 3;   // live 8-wide groups (tail zero padded by ld4)
    const float* pa = sA + (wm * 32 + i) * DBKP + 4 * h;
    const float* pb = sW + (wn * 32 + i) * DBKP + 4 * h;
    for (int q = 0; q < kq; ++q) {
      const float4 a = *(const float4*)(pa + 8 * q);
      const float4 b = *(const float4*)(pb + 8 * q);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // epilogue: accumulators -> LDS [BM][BN+4] -> 16-byte row segments
  constexpr int OP = BN + 4;
  float* sO = smem;
#pragma unroll
  for (int r = 0; r < 16; ++r)
    sO[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * OP + wn * 32 + i] = acc[r];
  __syncthreads();
  constexpr int C4 = BN / 4;                         // float4 per tile row
  for (int q = threadIdx.x; q < BM * C4; q += NTH) {
    const int r = q / C4, c = (q - r * C4) * 4;
    const int m = m0 + r, n = n0 + c;
    if (m >= M || n >= N) continue;
    float4 z = *(const float4*)(sO + r * OP + c);
    if (bias && act < ACT_D2) {
      const float4 bv = *(const float4*)(bias + n);
      z.x += bv.x; z.y += bv.y; z.z += bv.z; z.w += bv.w;
    }
    const int64_t o = (int64_t)m * N + n;
    if (rowscale) {
      const float sc = res[m];
      if (Z) *(float4*)(Z + o) = make_float4(sc, sc, sc, sc);
      *(float4*)(Y + o) = make_float4(z.x * sc, z.y * sc, z.z * sc, z.w * sc);
      continue;
    }
    if (act >= ACT_D2) {
      // double backward of a dense layer (dig_amd/diffops.py:_DgradAct): the GEMM result t = ggx W^T leaves as
      //   Y = t * act'(z0)      (gradient w.r.t. gy)      and      Z = t * gy0 * act''(z0)   (gradient w.r.t. z)
      // with z0 = bias-slot pointer (the layer's saved pre-activation), gy0 = res-slot pointer; both [M,N].
      const float4 z0 = *(const float4*)(bias + o), g0 = *(const float4*)(res + o);
      float d1, d2;
      float4 y, w;
      act_d12(z0.x, act - ACT_D2, d1, d2); y.x = z.x * d1; w.x = z.x * g0.x * d2;
      act_d12(z0.y, act - ACT_D2, d1, d2); y.y = z.y * d1; w.y = z.y * g0.y * d2;
      act_d12(z0.z, act - ACT_D2, d1, d2); y.z = z.z * d1; w.z = z.z * g0.z * d2;
      act_d12(z0.w, act - ACT_D2, d1, d2); y.w = z.w * d1; w.w = z.w * g0.w * d2;
      *(float4*)(Y + o) = y;
      *(float4*)(Z + o) = w;
      continue;
    }
    if (Z)
      *(float4*)(Z + o) = keepd ? make_float4(act_bwd(z.x, act), act_bwd(z.y, act), act_bwd(z.z, act), act_bwd(z.w, act)) : z;
    float4 y = make_float4(act_fwd(z.x, act), act_fwd(z.y, act), act_fwd(z.z, act), act_fwd(z.w, act));
    if (res) {
      const float4 rv = *(const float4*)(res + o);
      y.x = rv.x + y.x; y.y = rv.y + y.y; y.z = rv.z + y.z; y.w = rv.w + y.w;
    }
    *(float4*)(Y + o) = y;
  }
}

template <int WN>
__global__ void __launch_bounds__(NTH) k_linear_fwd(const float* __restrict__ X, const float* __restrict__ W,
                                                     const float* __restrict__ bias, const float* __restrict__ res,
                                                     int M, int K, int N, int act, float* __restrict__ Y,
                                                     float* __restrict__ Z) {
  constexpr int WM = 8 / WN;
  __shared__ float smem[(32 * WM + 32 * WN) * DBKP];
  linear_fwd_body<WN>(X, W, bias, res, M, K, N, act, Y, Z, smem, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// Small-M variants (E ~ 10^4 rows, the reference's batch size): a 64-row tile grid has only ~137 blocks for 256 CUs
// and each block needs 3.4 us of f32 MFMA for a [64,128]x[128,128] product (two waves per SIMD), so the layer is
// bound by HALF the chip working serially.  Here: 32-row tiles, 4 waves (one per SIMD: 1.7 us of MFMA per block),
// reduction staged in chunks of 64 so that a block needs 44 KB of LDS and three blocks share a CU — 272 blocks, all
// resident at once, the loads of one overlapping the MFMAs of its neighbours.
// ------------------------------------------------------------------------------------------------
#define SNTH 256
#define SKC 64           // reduction chunk
#define SKP 68           // LDS pitch of a chunk row (68 mod 64 = 4: conflict-free ds_read_b128, as DBKP)

// (the 32-row FORWARD variant was measured and removed: the forward never gained from it — tools/bench_dense.py, r02)

// ------------------------------------------------------------------------------------------------
// Large-M layers (M >= 32k rows, reduction length <= 128): PERSISTENT, WAVE-INDEPENDENT blocks.
//   MODE 0  forward          out = act(X W^T + b) (+res), pre-activation to Z      reduction over k, columns = N
//   MODE 1  input gradient   out = (gY * act'(Z)) W (+ gAdd)                        reduction over n, columns = K
// One block per CU keeps a 128-column slice of the weight operand in LDS for its whole life ([col][red], so both MFMA
// operands are read with ds_read_b128).  After that single staging barrier the eight waves never synchronise again:
// each walks its own sequence of 32-row tiles, stages the tile's rows (64 reduction columns at a time) in a PRIVATE
// 8.5 KB LDS slab, multiplies it against all 128 resident columns (4 accumulator tiles: one A fragment feeds 16 MFMAs)
// and stores the result straight from the accumulators (each store instruction writes two full 128-byte row segments).
// The loads of the next half tile are issued before the MFMAs of the current one; the second wave of the SIMD runs its
// MFMAs while this one stores.  At M = 262 144, K = N = 128 the block-synchronous version of this kernel (one barrier
// triple per tile, stores drained before the next tile) ran 144 us; the traffic bound is 402 MB -> ~65 us.
// ------------------------------------------------------------------------------------------------
#define PW_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
#define PW_SMEM_BYTES ((128 * DBKP + 8 * 32 * SKP) * 4)      // 67.6 KB weights + 8 x 8.5 KB private slabs = 134 KB

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// unconditional 16-byte load from a full 32-row tile starting at the uniform pointer `tb` (leading dimension ld):
// columns >= ncols read the row's last 4 values and the consumer zeroes them with a 0/1 multiplier when it commits the
// registers to LDS (a select at the load would be turned into a branch around it, and then s_waitcnt has to drain the
// queue instead of counting).  32-bit lane offsets on a scalar base keep the address in one VGPR.
__device__ __forceinline__ float4 ld4c(const float* __restrict__ tb, int ld, int lrow, int col, int ncols) {
  return *(const float4*)(tb + (lrow * ld + (col < ncols ? col : ncols - 4)));
}
__device__ __forceinline__ float4 f4scale(float4 v, float s) { return make_float4(v.x * s, v.y * s, v.z * s, v.w * s); }

// softplus - log 2 out of line: the exact log1pf(expf(z)) is ~150 instructions, 64 inlined copies per epilogue variant
__device__ __noinline__ float act_ssp_call(float z) { return act_fwd(z, ACT_SSP); }
template <int ACT>
__device__ __forceinline__ float act_fwd_c(float z) {
  if (ACT == ACT_SSP) return act_ssp_call(z);
  return act_fwd(z, ACT);
}

// NCH = 64-wide reduction chunks per tile: 2 (reduction <= 128: 128 resident columns, four accumulator tiles per wave) or
// 4 (reduction <= 256: 64 resident columns [64][260], two accumulator tiles; the column slices of one row range run on
// the same XCD — blockIdx.x + gridDim.x * blockIdx.y keeps x mod 8 — so their re-reads of the rows hit its L2).
template <int MODE, int ACT, bool HASZ, bool HASRES, int NCH>
__global__ void __launch_bounds__(NTH) k_linear_pw(const float* __restrict__ A, const float* __restrict__ Zp,
                                                    const float* __restrict__ W, const float* __restrict__ bias,
                                                    const float* __restrict__ res, int M, int K, int N,
                                                    float* __restrict__ Y, float* __restrict__ Z, int keepd, int ldw) {
  constexpr int NT = NCH == 2 ? 4 : 2, NCOL = 32 * NT, BP = NCH * 64 + 4;   // acc tiles, resident columns, sB pitch
  extern __shared__ float psm[];
  float* sB = psm;                                   // [NCOL out columns][BP reduction]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  float* sA = psm + 128 * DBKP + wave * (32 * SKP);  // this wave's rows [32][SKP]
  const int KR = MODE == 0 ? K : N;                  // reduction length (64 (NCH - 1) < KR <= 64 NCH, % 4 == 0)
  const int NO = MODE == 0 ? N : K;                  // output columns (% 4 == 0)
  const int c0 = blockIdx.y * NCOL;
  const int ntiles = M >> 5, nwaves = gridDim.x * 8;   // full tiles only (host: tail rows; ntiles >= 1)
  const int lr = lane >> 4, lc = (lane & 15) * 4;    // 4 rows x 64 columns per load instruction
  float4 ra[8], rz[8];
  auto fetch = [&](int tile, int ch) {
    const int64_t t0 = (int64_t)tile * 32 * KR;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      ra[it] = ld4c(A + t0, KR, lr + 4 * it, ch * 64 + lc, KR);
      if (MODE == 1 && ACT != ACT_NONE) rz[it] = ld4c(Zp + t0, KR, lr + 4 * it, ch * 64 + lc, KR);
    }
  };
  int tile = blockIdx.x * 8 + wave;
  {
    // Weight slice -> LDS.  Unconditional loads of clamped addresses, zeroed by a 0/1 factor on the way into LDS, with the
    // first row tile's loads issued behind them: with a predicate at the load every one of the eight sat in its own
    // branch followed by `s_waitcnt vmcnt(0)` + its LDS write — eight SERIAL round trips to L2, and the rows after them,
    // before the first MFMA of the block (M = 16 384: each wave has one tile and nothing hides the staging; 30.9 -> 29.4 us).
    constexpr int CPR = MODE == 0 ? NCH * 16 : NCOL / 4, RPP = NTH / CPR;   // float4 per staged row, rows per pass
    constexpr int NIT = (MODE == 0 ? NCOL : NCH * 64) / RPP;
    const int tr = threadIdx.x / CPR, tc = (threadIdx.x % CPR) * 4;
    const int col = MODE == 0 ? tc : c0 + tc;        // column of W
    const int colc = col < K ? col : K - 4;
    float4 w[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = (MODE == 0 ? c0 : 0) + tr + RPP * it;   // row of W
      w[it] = *(const float4*)(W + ((int64_t)(row < N ? row : N - 1) * ldw + colc));   // ldw >= K: a column slice of W
    }
    fetch(tile < ntiles ? tile : ntiles - 1, 0);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = (MODE == 0 ? c0 : 0) + tr + RPP * it;
      const float4 v = f4scale(w[it], (row < N && col < K) ? 1.0f : 0.0f);
      if (MODE == 0) {                               // sB[n][k] = W[c0 + n][k]: NCOL rows of 64 NCH floats
        *(float4*)(sB + (tr + RPP * it) * BP + tc) = v;
      } else {                                       // transposed: sB[k][n] = W[n][c0 + k], 64 NCH rows n of NCOL floats
        const int n = tr + RPP * it;
        sB[(tc + 0) * BP + n] = v.x;
        sB[(tc + 1) * BP + n] = v.y;
        sB[(tc + 2) * BP + n] = v.z;
        sB[(tc + 3) * BP + n] = v.w;
      }
    }
  }
  // (the rows are read-only and restrict-qualified, so the compiler is free to re-issue their loads after the barrier —
  // and does, for the input-gradient variants; an empty asm that consumes them pins the loads where they were written)
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    asm volatile("" ::"v"(ra[it].x), "v"(ra[it].y), "v"(ra[it].z), "v"(ra[it].w));
    if (MODE == 1 && ACT != ACT_NONE) asm volatile("" ::"v"(rz[it].x), "v"(rz[it].y), "v"(rz[it].z), "v"(rz[it].w));
  }
  __syncthreads();                                   // sB complete; the only block-wide barrier
  // every tile is full (the host sends M % 32 tail rows and ragged column slices elsewhere) and the optional outputs are
  // compile-time: loads and stores are straight-line code, so s_waitcnt counts the stores issued after the next tile's
  // loads instead of draining them
  constexpr bool hasz = HASZ, hasres = HASRES;
  float4 bvh[NT / 2];                                // this lane's bias values of the 64-column halves
#pragma unroll
  for (int half = 0; half < NT / 2; ++half)
    bvh[half] = (MODE == 0 && bias) ? *(const float4*)(bias + c0 + 64 * half + lc) : make_float4(0.f, 0.f, 0.f, 0.f);
  // Written so that s_waitcnt can COUNT instead of drain: the first tile is peeled off the loop, the (one or two)
  // reduction chunks are separate calls and the prefetch is unconditional (the last tile re-fetches itself), so on every
  // path into a commit the loads it needs are followed by a known number of stores.  Otherwise each wave waits for the
  // stores of tile i before it touches tile i+1; with 2048 waves in phase the chip alternates between an MFMA phase
  // with an idle memory system and a 64 MB store burst with idle matrix pipes (measured: times ADD, 55 us + 40-70 us).
  f32x16 acc[NT];
  auto chunk = [&](int tile, int ch, bool last) __attribute__((always_inline)) {
    {
      const float ckeep = ch * 64 + lc < KR ? 1.0f : 0.0f;
#pragma unroll
      for (int it = 0; it < 8; ++it)
        *(float4*)(sA + (lr + 4 * it) * SKP + lc) = f4scale(MODE == 1 ? gz4(ra[it], rz[it], ACT) : ra[it], ckeep);
      // ACT_DERIV (gZ = gY * Z, two multiplies per value): with nothing to compute the scheduler hoists the next chunk's 16
      // loads above these LDS writes and both register sets are live at once — 256 VGPRs + 4 ... 32 spilled to scratch
      // (r05: 98 scratch_ instructions in this instantiation).  The fence keeps the fetch behind the commit.
      if (MODE == 1 && ACT == ACT_DERIV) __builtin_amdgcn_sched_barrier(0);
      const int nxt = tile + nwaves < ntiles ? tile + nwaves : tile;
      fetch(last ? nxt : tile, last ? 0 : ch + 1);
      wave_lds_fence();
      const int left = KR - ch * 64;
      const int kq = ((left < 64 ? left : 64) + 7) >> 3;
      const float* pa = sA + i * SKP + 4 * h;
      const float* pb = sB + i * BP + ch * 64 + 4 * h;
      if (MODE == 0) {
        // fragment reads of step q+1 are issued before the 16 MFMAs of step q
        float4 a = *(const float4*)pa, b[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = *(const float4*)(pb + t * 32 * BP);
        for (int q = 0; q < kq; ++q) {
          const int qn = q + 1 < kq ? q + 1 : q;
          const float4 an = *(const float4*)(pa + 8 * qn);
          float4 bn[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) bn[t] = *(const float4*)(pb + t * 32 * BP + 8 * qn);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = PW_MFMA(a.x, b[t].x, acc[t]);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = PW_MFMA(a.y, b[t].y, acc[t]);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = PW_MFMA(a.z, b[t].z, acc[t]);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = PW_MFMA(a.w, b[t].w, acc[t]);
          a = an;
#pragma unroll
          for (int t = 0; t < NT; ++t) b[t] = bn[t];
        }
      } else {                                       // (the input gradient has no registers left for the look-ahead)
        for (int q = 0; q < kq; ++q) {
          const float4 a = *(const float4*)(pa + 8 * q);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const float4 b = *(const float4*)(pb + t * 32 * BP + 8 * q);
            acc[t] = PW_MFMA(a.x, b.x, acc[t]);
            acc[t] = PW_MFMA(a.y, b.y, acc[t]);
            acc[t] = PW_MFMA(a.z, b.z, acc[t]);
            acc[t] = PW_MFMA(a.w, b.w, acc[t]);
          }
        }
      }
      wave_lds_fence();                              // the slab is rewritten by the next half tile
    }
  };
  auto tile_body = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = zero16();
#pragma unroll
    for (int c = 0; c < NCH; ++c) chunk(tile, c, c == NCH - 1);   // always NCH chunks (no path-dependent registers)
    // epilogue through the private slab, 64 output columns at a time: accumulators -> [32][64] -> 16-byte row segments
    // (one store instruction = four 256-byte row pieces)
#pragma unroll
    for (int half = 0; half < NT / 2; ++half) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) sA[((r & 3) + 8 * (r >> 2) + 4 * h) * SKP + 32 * tt + i] = acc[2 * half + tt][r];
      wave_lds_fence();
      const int c = c0 + 64 * half + lc;
      const int64_t t0 = (int64_t)tile * 32 * NO;    // uniform tile base, 32-bit lane offsets
      const float4 bv = bvh[half];
#pragma unroll
      for (int g = 0; g < 2; ++g) {                // residual / gAdd rows fetched four at a time (register budget)
        float4 rv[4];
        if (hasres) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            rv[j] = *(const float4*)(res + t0 + ((lr + 4 * (4 * g + j)) * NO + c));
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int it = 4 * g + j;
          float4 z = *(const float4*)(sA + (lr + 4 * it) * SKP + lc);
          const int o = (lr + 4 * it) * NO + c;
          if (MODE == 0) {
            z = f4sum(z, bv);
            if (hasz)
              *(float4*)(Z + t0 + o) =
                  keepd ? make_float4(act_bwd(z.x, ACT), act_bwd(z.y, ACT), act_bwd(z.z, ACT), act_bwd(z.w, ACT)) : z;
            float4 y = make_float4(act_fwd_c<ACT>(z.x), act_fwd_c<ACT>(z.y), act_fwd_c<ACT>(z.z), act_fwd_c<ACT>(z.w));
            if (hasres) y = f4sum(rv[j], y);
            *(float4*)(Y + t0 + o) = y;
          } else {
            if (hasres) z = f4sum(rv[j], z);       // gAdd
            *(float4*)(Y + t0 + o) = z;
          }
        }
      }
      wave_lds_fence();
    }
  };
  if (tile < ntiles) {
    tile_body(tile);
    for (tile += nwaves; tile < ntiles; tile += nwaves) tile_body(tile);
  }
}

template <int MODE>
static int launch_pw(const float* A, const float* Zp, const float* W, const float* bias, const float* res, int M, int K,
                     int N, int act, float* Y, float* Z, hipStream_t st, int ldw = 0) {
  if (ldw == 0) ldw = K;               // W [N, ldw] row-major, the K columns used start at W
  const int keepd = (MODE == 0 && (act & ACT_KEEP_DERIV)) ? 1 : 0;
  if (MODE == 0) act &= 3;
  const bool wide = (MODE == 0 ? K : N) > 128;          // reduction 129..256: 64-column slices, four chunks
  const int ny = ((MODE == 0 ? N : K) + (wide ? 63 : 127)) / (wide ? 64 : 128);
  const dim3 grid(256 / ny > 0 ? 256 / ny : 1, ny);
#define PW_LAUNCH1(ACT, HZ, HR, NCH)                                                                                  \
  {                                                                                                                   \
    static const bool attr_ok = hipFuncSetAttribute((const void*)k_linear_pw<MODE, ACT, HZ, HR, NCH>,                 \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize,                      \
                                                     PW_SMEM_BYTES) == hipSuccess; /* set once */                     \
    if (!attr_ok) return 1;                                                                                           \
    hipLaunchKernelGGL((k_linear_pw<MODE, ACT, HZ, HR, NCH>), grid, dim3(NTH), PW_SMEM_BYTES, st, A, Zp, W, bias, res, \
                       M, K, N, Y, Z, keepd, ldw);                                                                    \
    return 0;                                                                                                         \
  }
#define PW_LAUNCH(ACT, HZ, HR)                                                                                        \
  {                                                                                                                   \
    if (wide) PW_LAUNCH1(ACT, HZ, HR, 4)                                                                              \
    PW_LAUNCH1(ACT, HZ, HR, 2)                                                                                        \
  }
#define PW_CASE(ACT)                                                                                                  \
  {                                                                                                                   \
    if (MODE == 0 && Z) {                                                                                             \
      if (res) PW_LAUNCH(ACT, (MODE == 0), true)                                                                      \
      PW_LAUNCH(ACT, (MODE == 0), false)                                                                              \
    }                                                                                                                 \
    if (res) PW_LAUNCH(ACT, false, true)                                                                              \
    PW_LAUNCH(ACT, false, false)                                                                                      \
  }
  if (act == ACT_SWISH) PW_CASE(ACT_SWISH)
  if (act == ACT_SSP) PW_CASE(ACT_SSP)
  if constexpr (MODE == 1) {
    if (act == ACT_DERIV) PW_CASE(ACT_DERIV)
  }
  PW_CASE(ACT_NONE)
#undef PW_LAUNCH
#undef PW_LAUNCH1
#undef PW_CASE
}

// G <= 8 independent layers of the SAME shape (the five output blocks update_v of a SphereNet / DimeNet++ forward:
// spherenet.py:185-216 — N_atoms x 256 GEMMs, 20 blocks each, latency bound) in ONE launch: blockIdx.z = layer.
#define GRP_MAX 8
struct GroupFwd {
  const float* X[GRP_MAX];
  const float* W[GRP_MAX];
  const float* bias[GRP_MAX];
  const float* res[GRP_MAX];
  float* Y[GRP_MAX];
  float* Z[GRP_MAX];
};
template <int WN>
__global__ void __launch_bounds__(NTH) k_linear_fwd_grouped(GroupFwd d, int M, int K, int N, int act) {
  constexpr int WM = 8 / WN;
  __shared__ float smem[(32 * WM + 32 * WN) * DBKP];
  const int g = blockIdx.z;
  linear_fwd_body<WN>(d.X[g], d.W[g], d.bias[g], d.res[g], M, K, N, act, d.Y[g], d.Z[g], smem, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// backward w.r.t. the input:  gX[m,k] = sum_n gZ[m,n] W[n,k],  gZ = gY * act'(Z).
// Block: 64 rows x 128 output columns (2 x 4 waves), reduction over n in chunks of DBK.
// A = gZ (16-byte reads, k-permutation as above); B[n][k] read with 4 ds_read_b32 per group from the
// row-major W chunk (lanes along k: conflict free).
// ------------------------------------------------------------------------------------------------
#define BWD_SMEM ((64 + DBK) * DBKP)          // floats: the dgrad staging (101 KB) also covers the wgrad tile

// gZa (optional, [M,N]): a gradient that reached the pre-activation directly (second-order term of the force path):
// the staged operand is gY * act'(Z) + gZa.

__device__ __forceinline__ void dgrad_body(const float* __restrict__ gY, const float* __restrict__ Zp,
                                           const float* __restrict__ W, int M, int K, int N, int act,
                                           float* __restrict__ gX, const float* __restrict__ gAdd,
                                           float* __restrict__ smem, int bx, int by,
                                           const float* __restrict__ gZa = nullptr) {
  float* sG = smem;                         // gZ chunk  [64 rows][DBK n]
  float* sW = smem + 64 * DBKP;             // W chunk   [DBK n][128 k]
  const int m0 = bx * 64, kb = by * 128;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  const int wm = wave >> 2, wk = wave & 3;
  const bool vecn = (N & 3) == 0, veck = (K & 3) == 0;
  const int tr = threadIdx.x >> 5, tc = (threadIdx.x & 31) * 4;   // 16 rows x 128 cols per pass
  float4 rg[4], rz[4], rw[8], ra[4];
  auto fetch = [&](int n0) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      rg[it] = ld4(gY, N, m0 + tr + 16 * it, M, n0 + tc, N, vecn);
      if (act != ACT_NONE) rz[it] = ld4(Zp, N, m0 + tr + 16 * it, M, n0 + tc, N, vecn);
      if (gZa) ra[it] = ld4(gZa, N, m0 + tr + 16 * it, M, n0 + tc, N, vecn);
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) rw[it] = ld4(W, K, n0 + tr + 16 * it, N, kb + tc, K, veck);
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      float4 v = gz4(rg[it], rz[it], act);
      if (gZa) v = f4sum(v, ra[it]);
      *(float4*)(sG + (tr + 16 * it) * DBKP + tc) = v;
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) *(float4*)(sW + (tr + 16 * it) * DBKP + tc) = rw[it];
  };
  f32x16 acc = zero16();
  fetch(0);
  for (int n0 = 0; n0 < N; n0 += DBK) {
    commit();
    __syncthreads();
    if (n0 + DBK < N) fetch(n0 + DBK);
    const int nq = ((N - n0 < DBK ? N - n0 : DBK) + 7) >> 3;
    const float* pa = sG + (wm * 32 + i) * DBKP + 4 * h;
    const float* pb = sW + (4 * h) * DBKP + wk * 32 + i;
    for (int q = 0; q < nq; ++q) {
      const float4 a = *(const float4*)(pa + 8 * q);
      const float* b = pb + (8 * q) * DBKP;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[DBKP], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[2 * DBKP], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[3 * DBKP], acc, 0, 0, 0);
    }
    __syncthreads();
  }
  float* sO = smem;                          // [64][132]
#pragma unroll
  for (int r = 0; r < 16; ++r) sO[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * DBKP + wk * 32 + i] = acc[r];
  __syncthreads();
  for (int q = threadIdx.x; q < 64 * 32; q += NTH) {
    const int r = q >> 5, c = (q & 31) * 4;
    const int m = m0 + r, k = kb + c;
    if (m >= M || k >= K) continue;
    float4 v = *(const float4*)(sO + r * DBKP + c);
    float* o = gX + (int64_t)m * K + k;
    if (veck) {
      if (gAdd) {                           // gX = gAdd + gZ W: folds the gradient accumulation of a residual stream
        const float4 a = *(const float4*)(gAdd + (int64_t)m * K + k);
        v.x = a.x + v.x; v.y = a.y + v.y; v.z = a.z + v.z; v.w = a.w + v.w;
      }
      *(float4*)o = v;
    } else {
      const float* a = gAdd ? gAdd + (int64_t)m * K + k : nullptr;
      o[0] = (a ? a[0] : 0.f) + v.x;
      if (k + 1 < K) o[1] = (a ? a[1] : 0.f) + v.y;
      if (k + 2 < K) o[2] = (a ? a[2] : 0.f) + v.z;
      if (k + 3 < K) o[3] = (a ? a[3] : 0.f) + v.w;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward w.r.t. weight and bias:  gW[n,k] = sum_m gZ[m,n] X[m,k],  gb[n] = sum_m gZ[m,n].
// grid = (row-chunk workers, n tiles of 128, k tiles of 128); each block strides over 32-row chunks and
// keeps a 128x128 partial in registers (wave w: n rows 32(w&3).., two 32-wide k tiles 64(w>>2)..).
// part[(blockIdx.x)][N*K + N]  ->  k_dense_reduce.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wgrad_body(const float* __restrict__ gY, const float* __restrict__ Zp,
                                           const float* __restrict__ X, int M, int K, int N, int act,
                                           float* __restrict__ part, float* __restrict__ smem, int wx, int wy, int wz,
                                           int nworkers, const float* __restrict__ gZa = nullptr, int db = 0) {
  // smem: staging gZ chunk [32 m][128 n] + X chunk [32 m][128 k]; then the [128][132] out tile
  // db: the staging pair alternates between the two halves of smem (the out tile needs all of it anyway), ONE barrier per
  // chunk: a wave that finishes its MFMAs commits the next chunk at once instead of waiting for the slowest wave
  float* sG = smem;
  float* sX = smem + 32 * DBKP;
  const int nb0 = wy * 128, kb0 = wz * 128;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  const int wn = wave & 3, wk = wave >> 2;
  const bool vecn = (N & 3) == 0, veck = (K & 3) == 0;
  const int tr = threadIdx.x >> 5, tc = (threadIdx.x & 31) * 4;   // 16 rows x 128 cols per pass
  f32x16 acc[2];
  acc[0] = zero16();
  acc[1] = zero16();
  float bsum = 0.f;                         // thread n < 128: column sum of gZ
  float4 rg[2], rz[2], rx[2], ra[2];
  auto fetch = [&](int m0) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      rg[it] = ld4(gY, N, m0 + tr + 16 * it, M, nb0 + tc, N, vecn);
      if (act != ACT_NONE) rz[it] = ld4(Zp, N, m0 + tr + 16 * it, M, nb0 + tc, N, vecn);
      if (gZa) ra[it] = ld4(gZa, N, m0 + tr + 16 * it, M, nb0 + tc, N, vecn);
      rx[it] = ld4(X, K, m0 + tr + 16 * it, M, kb0 + tc, K, veck);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      float4 v = gz4(rg[it], rz[it], act);
      if (gZa) v = f4sum(v, ra[it]);
      *(float4*)(sG + (tr + 16 * it) * DBKP + tc) = v;
      *(float4*)(sX + (tr + 16 * it) * DBKP + tc) = rx[it];
    }
  };
  const int nchunks = (M + 31) / 32;
  if (wx < nchunks) fetch(wx * 32);
  int flip = 0;
  for (int ch = wx; ch < nchunks; ch += nworkers) {
    if (db) {                               // (the other half: every wave's MFMAs of the chunk before the last are done —
      sG = smem + flip * (64 * DBKP);       //  each passed the barrier below once since)
      sX = sG + 32 * DBKP;
      flip ^= 1;
    } else {
      __syncthreads();                      // previous chunk's MFMA reads are done
    }
    commit();
    __syncthreads();
    if (ch + nworkers < nchunks) fetch((ch + nworkers) * 32);
    if (wz == 0 && threadIdx.x < 128) {
      float s = 0.f;
#pragma unroll 8
      for (int r = 0; r < 32; ++r) s += sG[r * DBKP + threadIdx.x];
      bsum += s;
    }
    const float* pa = sG + h * DBKP + wn * 32 + i;        // A[i = n][kk = m]: row m = 2*s + h
    const float* pb = sX + h * DBKP + wk * 64 + i;
#pragma unroll 4
    for (int s = 0; s < 16; ++s) {
      const float a = pa[(2 * s) * DBKP];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pb[(2 * s) * DBKP], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pb[(2 * s) * DBKP + 32], acc[1], 0, 0, 0);
    }
  }
  __syncthreads();
  // out tile [128 n][128 k] through LDS, then 16-byte row segments into this block's partial
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      smem[(wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * DBKP + wk * 64 + 32 * t + i] = acc[t][r];
  __syncthreads();
  float* outp = part + (int64_t)wx * ((int64_t)N * K + N);
  for (int q = threadIdx.x; q < 128 * 32; q += NTH) {
    const int r = q >> 5, c = (q & 31) * 4;
    const int n = nb0 + r, k = kb0 + c;
    if (n >= N || k >= K) continue;
    const float4 v = *(const float4*)(smem + r * DBKP + c);
    float* o = outp + (int64_t)n * K + k;
    if (veck) {
      *(float4*)o = v;
    } else {
      o[0] = v.x;
      if (k + 1 < K) o[1] = v.y;
      if (k + 2 < K) o[2] = v.z;
      if (k + 3 < K) o[3] = v.w;
    }
  }
  if (wz == 0 && threadIdx.x < 128 && nb0 + threadIdx.x < N) outp[(int64_t)N * K + nb0 + threadIdx.x] = bsum;
}

__global__ void __launch_bounds__(NTH) k_linear_bwd_input(const float* __restrict__ gY, const float* __restrict__ Zp,
                                                           const float* __restrict__ W, int M, int K, int N, int act,
                                                           float* __restrict__ gX, const float* __restrict__ gAdd) {
  __shared__ float smem[BWD_SMEM];
  dgrad_body(gY, Zp, W, M, K, N, act, gX, gAdd, smem, blockIdx.x, blockIdx.y);
}

__global__ void __launch_bounds__(NTH) k_linear_bwd_weight(const float* __restrict__ gY, const float* __restrict__ Zp,
                                                            const float* __restrict__ X, int M, int K, int N, int act,
                                                            float* __restrict__ part) {
  __shared__ float smem[128 * DBKP];
  wgrad_body(gY, Zp, X, M, K, N, act, part, smem, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x);
}

// Both gradients of one layer in ONE launch: the weight-gradient workers (few, long) are scheduled first, the
// input-gradient row tiles fill the remaining CUs.  Each of the two alone occupies about half of the chip at
// E ~ 10^4 rows (one 100-KB-LDS block per CU, 130-160 blocks), so back to back they cost their sum; merged, their max.
__global__ void __launch_bounds__(NTH) k_linear_bwd_both(const float* __restrict__ gY, const float* __restrict__ Zp,
                                                          const float* __restrict__ W, const float* __restrict__ X,
                                                          int M, int K, int N, int act, float* __restrict__ gX,
                                                          const float* __restrict__ gAdd, float* __restrict__ part,
                                                          int nworkers, int wg_blocks, const float* __restrict__ gZa) {
  __shared__ float smem[BWD_SMEM];
  int b = blockIdx.x;
  if (b < wg_blocks) {
    const int nt = (N + 127) / 128;
    const int wx = b % nworkers, wy = (b / nworkers) % nt, wz = b / (nworkers * nt);
    wgrad_body(gY, Zp, X, M, K, N, act, part, smem, wx, wy, wz, nworkers, gZa);
  } else {
    b -= wg_blocks;
    const int mt = (M + 63) / 64;
    dgrad_body(gY, Zp, W, M, K, N, act, gX, gAdd, smem, b % mt, b / mt, gZa);
  }
}

// ---- small-M backward: 256-thread blocks, two per CU (68 KB of LDS), dgrad on 32-row tiles, wgrad workers with four
// accumulator tiles per wave; see the small-M note above ------------------------------------------------
__device__ __forceinline__ void dgrad_body_s(const float* __restrict__ gY, const float* __restrict__ Zp,
                                             const float* __restrict__ W, int M, int K, int N, int act,
                                             float* __restrict__ gX, const float* __restrict__ gAdd,
                                             float* __restrict__ smem, int bx, int by,
                                             const float* __restrict__ gZa) {
  float* sG = smem;                    // gZ chunk [32 rows][SKC n]   pitch SKP
  float* sW = smem + 32 * SKP;         // W chunk  [SKC n][128 k]     pitch DBKP
  const int m0 = bx * 32, kb = by * 128;
  const int wk = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  const bool vecn = (N & 3) == 0, veck = (K & 3) == 0;
  const int gr = threadIdx.x >> 4, gc = (threadIdx.x & 15) * 4;   // 16 rows x 64 cols per pass (gZ chunk)
  const int wr = threadIdx.x >> 5, wc = (threadIdx.x & 31) * 4;   // 8 rows x 128 cols per pass (W chunk)
  float4 rg[2], rz[2], ra[2], rw[8];
  auto fetch = [&](int n0) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      rg[it] = ld4(gY, N, m0 + gr + 16 * it, M, n0 + gc, N, vecn);
      if (act != ACT_NONE) rz[it] = ld4(Zp, N, m0 + gr + 16 * it, M, n0 + gc, N, vecn);
      if (gZa) ra[it] = ld4(gZa, N, m0 + gr + 16 * it, M, n0 + gc, N, vecn);
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) rw[it] = ld4(W, K, n0 + wr + 8 * it, N, kb + wc, K, veck);
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      float4 v = gz4(rg[it], rz[it], act);
      if (gZa) v = f4sum(v, ra[it]);
      *(float4*)(sG + (gr + 16 * it) * SKP + gc) = v;
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) *(float4*)(sW + (wr + 8 * it) * DBKP + wc) = rw[it];
  };
  f32x16 acc = zero16();
  fetch(0);
  for (int n0 = 0; n0 < N; n0 += SKC) {
    commit();
    __syncthreads();
    if (n0 + SKC < N) fetch(n0 + SKC);
    const int nq = ((N - n0 < SKC ? N - n0 : SKC) + 7) >> 3;
    const float* pa = sG + i * SKP + 4 * h;
    const float* pb = sW + (4 * h) * DBKP + wk * 32 + i;
    for (int q = 0; q < nq; ++q) {
      const float4 a = *(const float4*)(pa + 8 * q);
      const float* b = pb + (8 * q) * DBKP;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[DBKP], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[2 * DBKP], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[3 * DBKP], acc, 0, 0, 0);
    }
    __syncthreads();
  }
  float* sO = smem;                    // [32][132]
#pragma unroll
  for (int r = 0; r < 16; ++r) sO[((r & 3) + 8 * (r >> 2) + 4 * h) * DBKP + wk * 32 + i] = acc[r];
  __syncthreads();
  for (int q = threadIdx.x; q < 32 * 32; q += SNTH) {
    const int r = q >> 5, c = (q & 31) * 4;
    const int m = m0 + r, k = kb + c;
    if (m >= M || k >= K) continue;
    float4 v = *(const float4*)(sO + r * DBKP + c);
    float* o = gX + (int64_t)m * K + k;
    if (veck) {
      if (gAdd) v = f4sum(*(const float4*)(gAdd + (int64_t)m * K + k), v);
      *(float4*)o = v;
    } else {
      const float* a = gAdd ? gAdd + (int64_t)m * K + k : nullptr;
      o[0] = (a ? a[0] : 0.f) + v.x;
      if (k + 1 < K) o[1] = (a ? a[1] : 0.f) + v.y;
      if (k + 2 < K) o[2] = (a ? a[2] : 0.f) + v.z;
      if (k + 3 < K) o[3] = (a ? a[3] : 0.f) + v.w;
    }
  }
}

// weight-gradient worker, 4 waves: wave w owns n rows [32w, 32w+32) x all 128 k columns (four accumulator tiles)
__device__ __forceinline__ void wgrad_body_s(const float* __restrict__ gY, const float* __restrict__ Zp,
                                             const float* __restrict__ X, int M, int K, int N, int act,
                                             float* __restrict__ part, float* __restrict__ smem, int wx, int wy, int wz,
                                             int nworkers, const float* __restrict__ gZa) {
  float* sG = smem;                    // [32 m][DBKP]
  float* sX = smem + 32 * DBKP;        // [32 m][DBKP]
  const int nb0 = wy * 128, kb0 = wz * 128;
  const int wn = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  const bool vecn = (N & 3) == 0, veck = (K & 3) == 0;
  const int tr = threadIdx.x >> 5, tc = (threadIdx.x & 31) * 4;   // 8 rows x 128 cols per pass
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = zero16();
  float bsum = 0.f;
  float4 rg[4], rz[4], rx[4], ra[4];
  auto fetch = [&](int m0) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      rg[it] = ld4(gY, N, m0 + tr + 8 * it, M, nb0 + tc, N, vecn);
      if (act != ACT_NONE) rz[it] = ld4(Zp, N, m0 + tr + 8 * it, M, nb0 + tc, N, vecn);
      if (gZa) ra[it] = ld4(gZa, N, m0 + tr + 8 * it, M, nb0 + tc, N, vecn);
      rx[it] = ld4(X, K, m0 + tr + 8 * it, M, kb0 + tc, K, veck);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      float4 v = gz4(rg[it], rz[it], act);
      if (gZa) v = f4sum(v, ra[it]);
      *(float4*)(sG + (tr + 8 * it) * DBKP + tc) = v;
      *(float4*)(sX + (tr + 8 * it) * DBKP + tc) = rx[it];
    }
  };
  const int nchunks = (M + 31) / 32;
  if (wx < nchunks) fetch(wx * 32);
  for (int ch = wx; ch < nchunks; ch += nworkers) {
    __syncthreads();
    commit();
    __syncthreads();
    if (ch + nworkers < nchunks) fetch((ch + nworkers) * 32);
    if (wz == 0 && threadIdx.x < 128) {
      float sm = 0.f;
#pragma unroll 8
      for (int r = 0; r < 32; ++r) sm += sG[r * DBKP + threadIdx.x];
      bsum += sm;
    }
    const float* pa = sG + h * DBKP + wn * 32 + i;
    const float* pb = sX + h * DBKP + i;
#pragma unroll 4
    for (int st = 0; st < 16; ++st) {
      const float a = pa[(2 * st) * DBKP];
      const float* b = pb + (2 * st) * DBKP;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[32], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[64], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[96], acc[3], 0, 0, 0);
    }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      smem[(wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * DBKP + 32 * t + i] = acc[t][r];
  __syncthreads();
  float* outp = part + (int64_t)wx * ((int64_t)N * K + N);
  for (int q = threadIdx.x; q < 128 * 32; q += SNTH) {
    const int r = q >> 5, c = (q & 31) * 4;
    const int n = nb0 + r, k = kb0 + c;
    if (n >= N || k >= K) continue;
    const float4 v = *(const float4*)(smem + r * DBKP + c);
    float* o = outp + (int64_t)n * K + k;
    if (veck) {
      *(float4*)o = v;
    } else {
      o[0] = v.x;
      if (k + 1 < K) o[1] = v.y;
      if (k + 2 < K) o[2] = v.z;
      if (k + 3 < K) o[3] = v.w;
    }
  }
  if (wz == 0 && threadIdx.x < 128 && nb0 + threadIdx.x < N) outp[(int64_t)N * K + nb0 + threadIdx.x] = bsum;
}

#define SBWD_SMEM (128 * DBKP)         // the wgrad out tile [128][132] (67.6 KB) covers every staging layout
__global__ void __launch_bounds__(SNTH) k_linear_bwd_both_s(const float* __restrict__ gY, const float* __restrict__ Zp,
                                                             const float* __restrict__ W, const float* __restrict__ X,
                                                             int M, int K, int N, int act, float* __restrict__ gX,
                                                             const float* __restrict__ gAdd, float* __restrict__ part,
                                                             int nworkers, int wg_blocks,
                                                             const float* __restrict__ gZa) {
  __shared__ float smem[SBWD_SMEM];
  int b = blockIdx.x;
  if (b < wg_blocks) {
    const int nt = (N + 127) / 128;
    const int wx = b % nworkers, wy = (b / nworkers) % nt, wz = b / (nworkers * nt);
    wgrad_body_s(gY, Zp, X, M, K, N, act, part, smem, wx, wy, wz, nworkers, gZa);
  } else {
    b -= wg_blocks;
    const int mt = (M + 31) / 32;
    dgrad_body_s(gY, Zp, W, M, K, N, act, gX, gAdd, smem, b % mt, b / mt, gZa);
  }
}
__global__ void __launch_bounds__(SNTH) k_linear_bwd_input_s(const float* __restrict__ gY, const float* __restrict__ Zp,
                                                              const float* __restrict__ W, int M, int K, int N, int act,
                                                              float* __restrict__ gX, const float* __restrict__ gAdd) {
  __shared__ float smem[32 * SKP + SKC * DBKP];
  dgrad_body_s(gY, Zp, W, M, K, N, act, gX, gAdd, smem, blockIdx.x, blockIdx.y, nullptr);
}

// Double backward of a dense layer in ONE launch (dig_amd/diffops.py:_DgradAct.backward): the weight-gradient workers of
//     gW = (gy * act'(z))^T ggx
// share the grid with the row tiles of   t = ggx W^T,  o_gy = t act'(z),  o_z = t gy act''(z)   (second-order epilogue).
__global__ void __launch_bounds__(NTH) k_linear_dd(const float* __restrict__ ggx, const float* __restrict__ W,
                                                    const float* __restrict__ Zp, const float* __restrict__ gy, int M,
                                                    int K, int N, int act, float* __restrict__ o_gy,
                                                    float* __restrict__ o_z, float* __restrict__ part, int nworkers,
                                                    int wg_blocks) {
  __shared__ float smem[BWD_SMEM];
  int b = blockIdx.x;
  if (b < wg_blocks) {
    const int nt = (N + 127) / 128;
    const int wx = b % nworkers, wy = (b / nworkers) % nt, wz = b / (nworkers * nt);
    wgrad_body(gy, Zp, ggx, M, K, N, act, part, smem, wx, wy, wz, nworkers);
  } else {
    b -= wg_blocks;
    const int mt = (M + 63) / 64;
    if (act == ACT_NONE)      // plain t = ggx W^T
      linear_fwd_body<4>(ggx, W, nullptr, nullptr, M, K, N, ACT_NONE, o_gy, nullptr, smem, b % mt, b / mt);
    else
      linear_fwd_body<4>(ggx, W, Zp, gy, M, K, N, ACT_D2 + act, o_gy, o_z, smem, b % mt, b / mt);
  }
}

struct GroupBwd {
  const float* gY[GRP_MAX];
  const float* Z[GRP_MAX];
  const float* W[GRP_MAX];
  const float* X[GRP_MAX];
  float* gX[GRP_MAX];
  const float* gAdd[GRP_MAX];
  float* part[GRP_MAX];
  const float* gZa[GRP_MAX];
};
// the backward of G same-shape layers in one launch (blockIdx.y = layer), each as in k_linear_bwd_both
__global__ void __launch_bounds__(NTH) k_linear_bwd_both_grouped(GroupBwd d, int M, int K, int N, int act, int nworkers,
                                                                  int wg_blocks) {
  __shared__ float smem[BWD_SMEM];
  const int g = blockIdx.y;
  int b = blockIdx.x;
  if (b < wg_blocks) {
    const int nt = (N + 127) / 128;
    const int wx = b % nworkers, wy = (b / nworkers) % nt, wz = b / (nworkers * nt);
    wgrad_body(d.gY[g], d.Z[g], d.X[g], M, K, N, act, d.part[g], smem, wx, wy, wz, nworkers, d.gZa[g]);
  } else {
    b -= wg_blocks;
    const int mt = (M + 63) / 64;
    dgrad_body(d.gY[g], d.Z[g], d.W[g], M, K, N, act, d.gX[g], d.gAdd[g], smem, b % mt, b / mt, d.gZa[g]);
  }
}

// input gradients of G same-shape layers (the create_graph backward of the grouped output blocks)
__global__ void __launch_bounds__(NTH) k_linear_bwd_input_grouped(GroupBwd d, int M, int K, int N, int act) {
  __shared__ float smem[BWD_SMEM];
  const int g = blockIdx.z;
  dgrad_body(d.gY[g], d.Z[g], d.W[g], M, K, N, act, d.gX[g], nullptr, smem, blockIdx.x, blockIdx.y);
}

// k_linear_dd for G same-shape layers: here X[g] = ggx, gY[g] = gy, gX[g] = o_gy, gAdd slot unused, gZa slot = o_z (out)
struct GroupDD {
  const float* ggx[GRP_MAX];
  const float* W[GRP_MAX];
  const float* Z[GRP_MAX];
  const float* gy[GRP_MAX];
  float* o_gy[GRP_MAX];
  float* o_z[GRP_MAX];
  float* part[GRP_MAX];
};
__global__ void __launch_bounds__(NTH) k_linear_dd_grouped(GroupDD d, int M, int K, int N, int act, int nworkers,
                                                            int wg_blocks) {
  __shared__ float smem[BWD_SMEM];
  const int g = blockIdx.y;
  int b = blockIdx.x;
  if (b < wg_blocks) {
    const int nt = (N + 127) / 128;
    const int wx = b % nworkers, wy = (b / nworkers) % nt, wz = b / (nworkers * nt);
    wgrad_body(d.gy[g], d.Z[g], d.ggx[g], M, K, N, act, d.part[g], smem, wx, wy, wz, nworkers);
  } else {
    b -= wg_blocks;
    const int mt = (M + 63) / 64;
    if (act == ACT_NONE)
      linear_fwd_body<4>(d.ggx[g], d.W[g], nullptr, nullptr, M, K, N, ACT_NONE, d.o_gy[g], nullptr, smem, b % mt, b / mt);
    else
      linear_fwd_body<4>(d.ggx[g], d.W[g], d.Z[g], d.gy[g], M, K, N, ACT_D2 + act, d.o_gy[g], d.o_z[g], smem, b % mt,
                         b / mt);
  }
}

// out[j] = sum_k part[k*stride + j]: 16 outputs x 16 partial lanes per block, fixed tree (deterministic)
__global__ void __launch_bounds__(256) k_dense_reduce(const float* __restrict__ part, int nparts, int64_t stride,
                                                      int n, float* __restrict__ out) {
  __shared__ float red[16][17];
  const int jj = threadIdx.x & 15, kg = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + jj;
  float s0 = 0.f, s1 = 0.f;
  if (j < n) {
    int k = kg;
    for (; k + 16 < nparts; k += 32) {       // two independent accumulation chains per thread
      s0 += part[(int64_t)k * stride + j];
      s1 += part[(int64_t)(k + 16) * stride + j];
    }
    if (k < nparts) s0 += part[(int64_t)k * stride + j];
  }
  red[kg][jj] = s0 + s1;
  __syncthreads();
  if (kg == 0 && j < n) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][jj];
    out[j] = t;
  }
}

// ================================================================================================
// C ABI.  Supported shapes: K % 8 == 0, N % 4 == 0 (forward needs only K % 8), all pointers 16-byte aligned.
// ================================================================================================
extern "C" {

// any K >= 1; N a multiple of 8 (output tiles and the dgrad k-pairing); K % 4 != 0 takes element-wise staging
int dig3d_linear_supported(int K, int N) { return (K > 0 && N > 0 && (N & 7) == 0) ? 1 : 0; }

// row-chunk workers (= partial gradients) of the weight-gradient kernels: a constant, the library keeps no mutable
// state.  Sweep on MI355X (SphereNet B=32 step): 32/48/64/96/128/192 -> 5.60/5.21/4.84/4.72/4.60/5.00 ms
#define kWgradWorkers (dig3d_num_cus() / 2)      // row-chunk workers of a single weight-gradient launch: half a block per CU

// When the 64-row tile grid cannot fill the chip (E ~ 10^4 rows) the 32-row / 256-thread kernels are an option.
// Measured on MI355X (same box, A/B, tools/bench_dense.py): the stand-alone input gradient gains at every size
// (17.2 -> 13.1 us at M = 9.4k, K = N = 128; 326 -> 274 us at M = 131k, K = N = 256; 9.97 -> 8.31 ms at M = 4.2M), the
// forward is a wash (13.5 vs 13.1 us; 265 vs 263 us) and the merged dgrad+wgrad launch loses at K = N = 128 (19.7 -> 24
// us at M = 8.7k; 304 -> 394 us at M = 262k: its weight-gradient workers get one wave per SIMD instead of two) but
// gains at K = N = 256 (94 -> 88 us at M = 16k, 18.0 -> 16.9 ms at M = 4.2M).  Defaults follow those measurements.
// The switches are read once (process-lifetime constants, for A/B measurements), the library keeps no mutable state.
// k_linear_pw pays off once (almost) every one of its 2048 waves has a 32-row tile: wave tiles = (M / 32) x column
// slices (128 wide for reductions <= 128, 64 wide above).  Measured crossover: K = N = 128 between M = 32k (tiled 24 us
// vs 30) and 64k (35.7 vs 46.2) -> 1536 wave tiles.  DIG3D_PW_MIN_TILES overrides (A/B, read once).
#define kPersistMinTiles (6 * dig3d_num_cus())   // = 1536 wave tiles on 256 CUs (the measured crossover)
static bool persist_rows(int M, int red, int cols) {
  const int slice = red > 128 ? 64 : 128;
  return (int64_t)(M >> 5) * ((cols + slice - 1) / slice) >= kPersistMinTiles;
}
static bool linear_small_m(int M, int K, int N) {
  return (int64_t)((M + 63) / 64) * ((K + 127) / 128) < 384 && M >= 64;
}

// Y[M,N] = act(X[M,K] W[N,K]^T + bias[N]) (+ res[M,N]);  Z (optional) receives the pre-activation — or, with act |
// ACT_KEEP_DERIV (5, 6), act'(pre-activation): the backward entries then take act = ACT_DERIV (3) and only multiply.
int dig3d_linear_fwd(const float* X, const float* W, const float* bias, const float* res, int M, int K, int N,
                     int act, float* Y, float* Z, void* stream) {
  DIG3D_ENTER();
  const bool d2 = act >= ACT_D2;       // second-order epilogue: bias slot = z0 [M,N], res slot = gy0 [M,N], Z = 2nd output
  const bool keepd = act == (ACT_SWISH | ACT_KEEP_DERIV) || act == (ACT_SSP | ACT_KEEP_DERIV);
  if (M < 0 || !dig3d_linear_supported(K, N) || !X || !W || !Y || act < 0 || (act > 2 && !d2 && !keepd) || act > ACT_D2 + 2)
    return DIG3D_ERR_ARG;
  if ((d2 && (!bias || !res || !Z)) || (keepd && !Z)) return DIG3D_ERR_ARG;
  if (!al16(X) || !al16(W)) return DIG3D_ERR_ARG;
  if (M == 0) return DIG3D_OK;
  hipStream_t st = (hipStream_t)stream;
  if (((uintptr_t)Y | (uintptr_t)Z | (uintptr_t)res | (uintptr_t)bias) & 15) return DIG3D_ERR_ARG;
  if (act < ACT_D2 && (K & 3) == 0 && persist_rows(M, K, N) &&
      ((K > 64 && K <= 128 && (N & 127) == 0) || (K > 128 && K <= 256 && (N & 63) == 0))) {
    // large M, K <= 128: persistent wave-independent blocks over the full 32-row tiles (k_linear_pw<0>), W slice
    // resident in LDS; the M % 32 tail rows go through the tiled kernel
    const int Mf = M & ~31, Mt = M - Mf;
    if (launch_pw<0>(X, nullptr, W, bias, res, Mf, K, N, act, Y, Z, st)) return DIG3D_ERR_LAUNCH;
    if (Mt) {
      const int64_t ox = (int64_t)Mf * K, oy = (int64_t)Mf * N;
      hipLaunchKernelGGL((k_linear_fwd<4>), dim3(1, (N + 127) / 128), dim3(NTH), 0, st, X + ox, W, bias, res ? res + oy : res,
                         Mt, K, N, act, Y + oy, Z ? Z + oy : Z);
    }
  } else if (N > 64) {
    dim3 grid((M + 63) / 64, (N + 127) / 128);
    hipLaunchKernelGGL((k_linear_fwd<4>), grid, dim3(NTH), 0, st, X, W, bias, res, M, K, N, act, Y, Z);
  } else if (N > 32) {
    dim3 grid((M + 127) / 128, 1);
    hipLaunchKernelGGL((k_linear_fwd<2>), grid, dim3(NTH), 0, st, X, W, bias, res, M, K, N, act, Y, Z);
  } else {
    dim3 grid((M + 255) / 256, 1);
    hipLaunchKernelGGL((k_linear_fwd<1>), grid, dim3(NTH), 0, st, X, W, bias, res, M, K, N, act, Y, Z);
  }
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// Y[M,N] = rs[m] * (X W^T + bias)[m,:] — SchNet's filter `mlp(gauss(d)) * cosine_cutoff(d)` (method/schnet/schnet.py:31-33)
// without the elementwise multiply; D[M,N] (or NULL) receives rs[m] in every column: the backward of the layer is the
// plain one with Z = D and act = ACT_DERIV (gZ = gY * D).  rs itself gets no gradient here.
int dig3d_linear_fwd_rowscale(const float* X, const float* W, const float* bias, const float* rs, int M, int K, int N,
                              float* Y, float* D, void* stream) {
  DIG3D_ENTER();
  if (M < 0 || !dig3d_linear_supported(K, N) || !X || !W || !Y || !rs) return DIG3D_ERR_ARG;
  if (!al16(X) || !al16(W) || !al16(Y) || !al16(D) || !al16(bias)) return DIG3D_ERR_ARG;
  if (M == 0) return DIG3D_OK;
  hipStream_t st = (hipStream_t)stream;
  if (N > 64)
    hipLaunchKernelGGL((k_linear_fwd<4>), dim3((M + 63) / 64, (N + 127) / 128), dim3(NTH), 0, st, X, W, bias, rs, M, K, N,
                       ACT_ROWSCALE, Y, D);
  else if (N > 32)
    hipLaunchKernelGGL((k_linear_fwd<2>), dim3((M + 127) / 128, 1), dim3(NTH), 0, st, X, W, bias, rs, M, K, N, ACT_ROWSCALE, Y,
                       D);
  else
    hipLaunchKernelGGL((k_linear_fwd<1>), dim3((M + 255) / 256, 1), dim3(NTH), 0, st, X, W, bias, rs, M, K, N, ACT_ROWSCALE, Y,
                       D);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// gX[M,K] = (gY * act'(Z)) W (+ gx_add[M,K] when non-NULL)      (Z may be NULL when act == 0)
int dig3d_linear_bwd_input(const float* gY, const float* Z, const float* W, int M, int K, int N, int act,
                           float* gX, const float* gx_add, void* stream) {
  DIG3D_ENTER();
  if (M < 0 || !dig3d_linear_supported(K, N) || !gY || !W || !gX || (act != 0 && !Z)) return DIG3D_ERR_ARG;
  if (!al16(gY) || !al16(Z) || !al16(W)) return DIG3D_ERR_ARG;
  if (M == 0) return DIG3D_OK;
  if ((N & 3) == 0 && al16(gX) && al16(gx_add) && persist_rows(M, N, K) &&
      ((N > 64 && N <= 128 && (K & 127) == 0) || (N > 128 && N <= 256 && (K & 63) == 0))) {
    // large M, N <= 128: k_linear_pw<1> over the full 32-row tiles, the M % 32 tail rows through the tiled kernel
    const int Mf = M & ~31, Mt = M - Mf;
    hipStream_t st = (hipStream_t)stream;
    if (launch_pw<1>(gY, Z, W, nullptr, gx_add, Mf, K, N, act, gX, nullptr, st)) return DIG3D_ERR_LAUNCH;
    if (Mt) {
      const int64_t oy = (int64_t)Mf * N, ox = (int64_t)Mf * K;
      hipLaunchKernelGGL(k_linear_bwd_input_s, dim3(1, (K + 127) / 128), dim3(SNTH), 0, st, gY + oy, Z ? Z + oy : Z, W, Mt, K, N,
                         act, gX + ox, gx_add ? gx_add + ox : gx_add);
    }
  } else if (M >= 64) {      // better at every M measured (8.4k ... 4.2M rows: +0 ... +20 %)
    dim3 grid((M + 31) / 32, (K + 127) / 128);
    hipLaunchKernelGGL(k_linear_bwd_input_s, grid, dim3(SNTH), 0, (hipStream_t)stream, gY, Z, W, M, K, N, act, gX,
                       gx_add);
  } else {
    dim3 grid((M + 63) / 64, (K + 127) / 128);
    hipLaunchKernelGGL(k_linear_bwd_input, grid, dim3(NTH), 0, (hipStream_t)stream, gY, Z, W, M, K, N, act, gX, gx_add);
  }
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// ---- a column slice of the weight: Y = act(X W[:, c0 : c0 + K]^T + bias) (+ res) and gX = (gY act'(Z)) W[:, c0 : c0 + K] ----
// lin_cat(cat([h1, h2], 1)) = h1 W[:, :H]^T + (h2 W[:, H:]^T + b) (method/comenet/comenet.py:199-200): with the slice taken
// by the kernel (W points at column c0, rows are ldw floats apart) the concatenation is never formed, both products run
// on the persistent kernel (reduction <= 256) and the input gradients come out as two contiguous tensors.  Only the
// shapes of the persistent kernel with M % 32 == 0 (dig3d_linear_wslice_supported); other shapes: concatenate.
int dig3d_linear_wslice_supported(int M, int K, int N) {
  return (M > 0 && (M & 31) == 0 && (K & 3) == 0 && (N & 3) == 0 && persist_rows(M, K, N) && persist_rows(M, N, K) &&
          ((K > 64 && K <= 128 && (N & 127) == 0) || (K > 128 && K <= 256 && (N & 63) == 0)) &&
          ((N > 64 && N <= 128 && (K & 127) == 0) || (N > 128 && N <= 256 && (K & 63) == 0)))
             ? 1
             : 0;
}

int dig3d_linear_fwd_wslice(const float* X, const float* W, int ldw, const float* bias, const float* res, int M, int K,
                            int N, int act, float* Y, float* Z, void* stream) {
  DIG3D_ENTER();
  const bool keepd = act == (ACT_SWISH | ACT_KEEP_DERIV) || act == (ACT_SSP | ACT_KEEP_DERIV);
  if (!dig3d_linear_wslice_supported(M, K, N) || !X || !W || !Y || ldw < K || (ldw & 3) || act < 0 || (act > 2 && !keepd) ||
      (keepd && !Z))
    return DIG3D_ERR_ARG;
  if (!al16(X) || !al16(W) || !al16(Y) || !al16(Z) || !al16(res) || !al16(bias)) return DIG3D_ERR_ARG;
  if (launch_pw<0>(X, nullptr, W, bias, res, M, K, N, act, Y, Z, (hipStream_t)stream, ldw)) return DIG3D_ERR_LAUNCH;
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_linear_bwd_input_wslice(const float* gY, const float* Z, const float* W, int ldw, int M, int K, int N, int act,
                                  float* gX, const float* gx_add, void* stream) {
  DIG3D_ENTER();
  if (!dig3d_linear_wslice_supported(M, K, N) || !gY || !W || !gX || ldw < K || (ldw & 3) || act < 0 || act > ACT_DERIV ||
      (act != 0 && !Z))
    return DIG3D_ERR_ARG;
  if (!al16(gY) || !al16(Z) || !al16(W) || !al16(gX) || !al16(gx_add)) return DIG3D_ERR_ARG;
  if (launch_pw<1>(gY, Z, W, nullptr, gx_add, M, K, N, act, gX, nullptr, (hipStream_t)stream, ldw)) return DIG3D_ERR_LAUNCH;
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// Both gradients of a layer in one launch: gX[M,K] and gWb[N*K+N] (see dig3d_linear_bwd_weight); part as there.
int dig3d_linear_wgrad_blocks(int M);

// number of weight-gradient workers (= partials written) dig3d_linear_bwd uses for this shape
int dig3d_linear_bwd_workers(int M, int K, int N) {
  int nb = dig3d_linear_wgrad_blocks(M);
  const int dg = ((M + 63) / 64) * ((K + 127) / 128);
  const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
  // every block needs a CU to itself (100 KB of LDS): when both gradients fit into ONE wave of 256 blocks, size the
  // weight-gradient workers to exactly the CUs the row tiles leave free instead of spilling a few blocks into a
  // second wave (E = 8.7k rows: 136 row tiles + 128 workers = 264 blocks -> 120 workers)
  if (dg < 256 && tiles == 1 && 256 - dg >= 32 && nb > 256 - dg) nb = 256 - dg;
  return nb;
}

static int linear_bwd_impl(const float* gY, const float* Z, const float* W, const float* X, int M, int K, int N, int act,
                           float* gX, const float* gx_add, float* part, float* gWb, int reduce_now,
                           const float* gz_add, void* stream) {
  DIG3D_ENTER();
  if (M < 0 || !dig3d_linear_supported(K, N) || !gY || !W || !X || !gX || !part || !gWb || (act != 0 && !Z))
    return DIG3D_ERR_ARG;
  if (!al16(gY) || !al16(Z) || !al16(W) || !al16(X) || !al16(gz_add)) return DIG3D_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    if (dig3d_zero_async(gWb, sizeof(float) * ((size_t)N * K + N), st) != hipSuccess) return DIG3D_ERR_LAUNCH;
    return DIG3D_OK;
  }
  const int nb = dig3d_linear_bwd_workers(M, K, N);
  const int dg = ((M + 63) / 64) * ((K + 127) / 128);
  const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
  const int wg = nb * tiles;
  if (!gz_add && (N & 3) == 0 && al16(gX) && al16(gx_add) && M >= 49152 && persist_rows(M, N, K) &&
      ((N > 64 && N <= 128 && (K & 127) == 0) || (N > 128 && N <= 256 && (K & 63) == 0))) {
    // large M: every CU is busy with either gradient on its own, so the merged launch buys nothing; the input gradient
    // goes through the persistent kernel (k_linear_pw<1>, 127 us at M = 262 144, K = N = 128), the weight gradient
    // through its workers (157 us) — merged they took 318 us
    const int Mf = M & ~31, Mt = M - Mf;
    if (launch_pw<1>(gY, Z, W, nullptr, gx_add, Mf, K, N, act, gX, nullptr, st)) return DIG3D_ERR_LAUNCH;
    if (Mt) {
      const int64_t oy = (int64_t)Mf * N, ox = (int64_t)Mf * K;
      hipLaunchKernelGGL(k_linear_bwd_input_s, dim3(1, (K + 127) / 128), dim3(SNTH), 0, st, gY + oy, Z ? Z + oy : Z, W, Mt, K, N,
                         act, gX + ox, gx_add ? gx_add + ox : gx_add);
    }
    hipLaunchKernelGGL(k_linear_bwd_weight, dim3(nb, (N + 127) / 128, (K + 127) / 128), dim3(NTH), 0, st, gY, Z, X, M, K,
                       N, act, part);
  } else if (K >= 256 && N >= 256 && M >= 64) {
    // E ~ 10^4 rows: 256-thread blocks, two per CU, 32-row dgrad tiles (k_linear_bwd_both_s)
    const int dgs = ((M + 31) / 32) * ((K + 127) / 128);
    hipLaunchKernelGGL(k_linear_bwd_both_s, dim3(wg + dgs), dim3(SNTH), 0, st, gY, Z, W, X, M, K, N, act, gX, gx_add,
                       part, nb, wg, gz_add);
  } else {
    hipLaunchKernelGGL(k_linear_bwd_both, dim3(wg + dg), dim3(NTH), 0, st, gY, Z, W, X, M, K, N, act, gX, gx_add, part,
                       nb, wg, gz_add);
  }
  DIG3D_CHECK_LAUNCH();
  const int64_t stride = (int64_t)N * K + N;
  if (reduce_now) {
    hipLaunchKernelGGL(k_dense_reduce, dim3(dig3d_blocks(stride, 16)), dim3(256), 0, st, part, nb, stride,
                       (int)stride, gWb);
    DIG3D_CHECK_LAUNCH();
  }
  return DIG3D_OK;
}

int dig3d_linear_bwd(const float* gY, const float* Z, const float* W, const float* X, int M, int K, int N, int act,
                     float* gX, const float* gx_add, float* part, float* gWb, int reduce_now, void* stream) {
  return linear_bwd_impl(gY, Z, W, X, M, K, N, act, gX, gx_add, part, gWb, reduce_now, nullptr, stream);
}

// the same with gz_add [M,N] added to the pre-activation gradient: gZ = gY * act'(Z) + gz_add — the final backward of a
// layer whose pre-activation also received the act'' term of the force double backward (dig_amd/diffops.py).
int dig3d_linear_bwd_zadd(const float* gY, const float* Z, const float* W, const float* X, int M, int K, int N, int act,
                          float* gX, const float* gx_add, float* part, float* gWb, int reduce_now, const float* gz_add,
                          void* stream) {
  return linear_bwd_impl(gY, Z, W, X, M, K, N, act, gX, gx_add, part, gWb, reduce_now, gz_add, stream);
}

// o_gy[M,N] = (ggx W^T) act'(Z),  o_z[M,N] = (ggx W^T) gy act''(Z)  (o_z unused when act == 0), and
// gWb[0 : N*K] = (gy act'(Z))^T ggx  — the three results of the double backward of Y = act(X W^T + b) w.r.t. an incoming
// ggx [M,K], in one launch.  N > 64.  part: float[dig3d_linear_bwd_workers(M, N, K) ... see dig3d_linear_dd_workers].
int dig3d_linear_dd_workers(int M, int K, int N) {
  int nb = dig3d_linear_wgrad_blocks(M);
  const int fw = ((M + 63) / 64) * ((N + 127) / 128);
  const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
  if (fw < 256 && tiles == 1 && 256 - fw >= 32 && nb > 256 - fw) nb = 256 - fw;
  return nb;
}
int dig3d_linear_dd(const float* ggx, const float* W, const float* Z, const float* gy, int M, int K, int N, int act,
                    float* o_gy, float* o_z, float* part, float* gWb, int reduce_now, void* stream) {
  DIG3D_ENTER();
  if (M < 0 || !dig3d_linear_supported(K, N) || N <= 64 || !ggx || !W || !gy || !o_gy || !part || !gWb || act < 0 ||
      act > 2 || (act != 0 && (!Z || !o_z)))
    return DIG3D_ERR_ARG;
  if (!al16(ggx) || !al16(W) || !al16(Z) || !al16(gy) || !al16(o_gy) || !al16(o_z)) return DIG3D_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t stride = (int64_t)N * K + N;
  if (M == 0) {
    if (dig3d_zero_async(gWb, sizeof(float) * (size_t)stride, st) != hipSuccess) return DIG3D_ERR_LAUNCH;
    return DIG3D_OK;
  }
  const int nb = dig3d_linear_dd_workers(M, K, N);
  const int fw = ((M + 63) / 64) * ((N + 127) / 128);
  const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
  const int wg = nb * tiles;
  hipLaunchKernelGGL(k_linear_dd, dim3(wg + fw), dim3(NTH), 0, st, ggx, W, Z, gy, M, K, N, act, o_gy, o_z, part, nb, wg);
  DIG3D_CHECK_LAUNCH();
  if (reduce_now) {
    hipLaunchKernelGGL(k_dense_reduce, dim3(dig3d_blocks(stride, 16)), dim3(256), 0, st, part, nb, stride, (int)stride,
                       gWb);
    DIG3D_CHECK_LAUNCH();
  }
  return DIG3D_OK;
}

int dig3d_linear_wgrad_blocks(int M) {
  // partial traffic (nb x (N*K+N) floats written, then read) against MFMA time per worker
  int nch = (M + 31) / 32;
  // one worker per CU once every worker has >= 4 chunks of its own (M >= 32k): 128 workers leave half the chip idle
  const int cap = M >= 32768 ? 2 * kWgradWorkers : kWgradWorkers;
  if (nch > cap) nch = cap;
  // a few hundred rows (the atom-level output blocks: M ~ 600, five groups x four 128x128 tiles per launch): one worker
  // per 32-row chunk means 380 blocks that each write a 64-KB partial for 32 rows of work; kSmallMDiv chunks per worker
  constexpr int kSmallMDiv = 3;
  if ((M + 31) / 32 <= 32) nch = (nch + kSmallMDiv - 1) / kSmallMDiv;
  return nch < 1 ? 1 : nch;
}

// gWb[N*K + N]: the weight gradient [N,K] followed by the bias gradient [N] (one buffer, one reduction).
// part: float[dig3d_linear_wgrad_blocks(M) * (N*K + N)].
int dig3d_linear_bwd_weight(const float* gY, const float* Z, const float* X, int M, int K, int N, int act,
                            float* part, float* gWb, int reduce_now, void* stream) {
  DIG3D_ENTER();
  if (M < 0 || !dig3d_linear_supported(K, N) || !gY || !X || !gWb || !part || (act != 0 && !Z)) return DIG3D_ERR_ARG;
  if (!al16(gY) || !al16(Z) || !al16(X)) return DIG3D_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    if (dig3d_zero_async(gWb, sizeof(float) * ((size_t)N * K + N), st) != hipSuccess) return DIG3D_ERR_LAUNCH;
    return DIG3D_OK;
  }
  const int nb = dig3d_linear_wgrad_blocks(M);
  dim3 grid(nb, (N + 127) / 128, (K + 127) / 128);
  hipLaunchKernelGGL(k_linear_bwd_weight, grid, dim3(NTH), 0, st, gY, Z, X, M, K, N, act, part);
  DIG3D_CHECK_LAUNCH();
  const int64_t stride = (int64_t)N * K + N;
  if (reduce_now) {
    hipLaunchKernelGGL(k_dense_reduce, dim3(dig3d_blocks(stride, 16)), dim3(256), 0, st, part, nb, stride,
                       (int)stride, gWb);
    DIG3D_CHECK_LAUNCH();
  }
  return DIG3D_OK;
}

// out_d[j] = sum_k part_d[k*stride_d + j] for `count` <= RED_MAX independent reductions in ONE launch (the weight
// gradients of many layers whose per-layer reduction was deferred: dig3d_linear_bwd(..., reduce_now = 0)).
#define RED_MAX 120         // 120 x 32 bytes of table + the flag: inside the 4-KB kernel-argument segment (a SphereNet step has ~70)
struct ReduceTable {
  const float* part[RED_MAX];
  float* out[RED_MAX];
  int64_t stride[RED_MAX];
  int nparts[RED_MAX];
  int n[RED_MAX];
  int accumulate;          // 1: out += sum (a second set of partials of gradients already reduced by an earlier launch)
};
static_assert(sizeof(ReduceTable) <= 4096, "kernel arguments");
#define RED_KG 4           // partial-index groups per block (x 64 consecutive outputs = 256 threads)
__global__ void __launch_bounds__(64 * RED_KG) k_reduce_many(ReduceTable t) {
  // 64 consecutive outputs per block row (256-byte segments of every partial: the 16-wide version read 64-byte pieces,
  // half of each 128-byte line), RED_KG lanes per output over the partials, 8 independent accumulators each
  // (16-byte lanes — four outputs per thread, a quarter of the workgroups — measured inside the step: +17 us; the partials
  // come from HBM / MALL, not L2, and the reduction lives on the number of requests in flight).
  // r05: 8 loads in flight per thread instead of 4.  The launch's time is that of its LONGEST reduction (the wave-per-
  // segment triplet backward leaves ~1 000 partials per layer instead of ~500: 1 000 / 4 groups / 8 in flight = 30 dependent
  // round trips, what 500 / 4 / 4 cost before).  Measured and not kept: 16 groups x 8 in flight on 1 024-thread blocks —
  // 18.8 -> 48.8 us per launch in the config-2 step (most of the ~60 reductions of a launch have 30 - 120 partials: the
  // extra groups idle, two blocks per CU, a 16-way LDS sum per output).
  __shared__ float red[RED_KG * 64];
  const int d = blockIdx.y;
  const int n = t.n[d], nparts = t.nparts[d];
  const int64_t stride = t.stride[d];
  const float* __restrict__ part = t.part[d];
  if (nparts >= 256 && n <= 4096) {
    // TALL reduction (few outputs, many partials: the second-Linear gradients of the triplet backward, 1 024 outputs x ~1 000
    // partials): 64 outputs per block would leave 16 blocks walking 250 partials per thread.  Here 16 outputs x 16 partial
    // groups per block: 64 blocks, 61 partials per thread, 8 in flight.  (64-byte pieces of every partial row — half a cache
    // line, the reason the wide form uses 64 outputs — do not matter at 4 MB.)
    const int jj = threadIdx.x & 15, kg = threadIdx.x >> 4;
    for (int j0 = blockIdx.x * 16; j0 < n; j0 += gridDim.x * 16) {      // uniform per block
      const int j = j0 + jj;
      float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (j < n) {
        int k = kg;
        for (; k + 7 * 16 < nparts; k += 8 * 16) {
#pragma unroll
          for (int u = 0; u < 8; ++u) s[u] += part[(int64_t)(k + u * 16) * stride + j];
        }
        for (; k < nparts; k += 16) s[0] += part[(int64_t)k * stride + j];
      }
      __syncthreads();
      red[kg * 16 + jj] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
      __syncthreads();
      if (kg == 0 && j < n) {
        float v = red[jj];
#pragma unroll
        for (int g = 1; g < 16; ++g) v += red[g * 16 + jj];
        t.out[d][j] = t.accumulate ? t.out[d][j] + v : v;
      }
    }
    return;
  }
  const int jj = threadIdx.x & 63, kg = threadIdx.x >> 6;
  for (int j0 = blockIdx.x * 64; j0 < n; j0 += gridDim.x * 64) {        // uniform per block
    const int j = j0 + jj;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (j < n) {
      int k = kg;
      for (; k + 7 * RED_KG < nparts; k += 8 * RED_KG) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] += part[(int64_t)(k + u * RED_KG) * stride + j];
      }
      for (; k < nparts; k += RED_KG) s[0] += part[(int64_t)k * stride + j];
    }
    __syncthreads();
    red[kg * 64 + jj] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (kg == 0 && j < n) {
      float v = red[jj];
#pragma unroll
      for (int g = 1; g < RED_KG; ++g) v += red[g * 64 + jj];
      t.out[d][j] = t.accumulate ? t.out[d][j] + v : v;
    }
  }
}

// host arrays of length count (any count: chunked by RED_MAX internally)
static int reduce_many_impl(const void* const* parts, const int* nparts, const int64_t* strides, const int* ns,
                            void* const* outs, int count, int accumulate, void* stream) {
  DIG3D_ENTER();
  if (count < 0 || (count > 0 && (!parts || !nparts || !strides || !ns || !outs))) return DIG3D_ERR_ARG;
  for (int c0 = 0; c0 < count; c0 += RED_MAX) {
    ReduceTable t;
    t.accumulate = accumulate;
    const int c = count - c0 < RED_MAX ? count - c0 : RED_MAX;
    int maxn = 1, tallb = 0;
    for (int d = 0; d < c; ++d) {
      if (!parts[c0 + d] || !outs[c0 + d] || nparts[c0 + d] < 1 || ns[c0 + d] < 0) return DIG3D_ERR_ARG;
      t.part[d] = (const float*)parts[c0 + d];
      t.out[d] = (float*)outs[c0 + d];
      t.stride[d] = strides[c0 + d];
      t.nparts[d] = nparts[c0 + d];
      t.n[d] = ns[c0 + d];
      if (ns[c0 + d] > maxn) maxn = ns[c0 + d];
      if (nparts[c0 + d] >= 256 && ns[c0 + d] <= 4096 && (ns[c0 + d] + 15) / 16 > tallb) tallb = (ns[c0 + d] + 15) / 16;
    }
    int bx = (maxn + 63) / 64;
    if (bx < tallb) bx = tallb;             // tall reductions take 16 outputs per block (k_reduce_many)
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(k_reduce_many, dim3(bx, c), dim3(64 * RED_KG), 0, (hipStream_t)stream, t);
    DIG3D_CHECK_LAUNCH();
  }
  return DIG3D_OK;
}

int dig3d_reduce_many(const void* const* parts, const int* nparts, const int64_t* strides, const int* ns,
                      void* const* outs, int count, void* stream) {
  return reduce_many_impl(parts, nparts, strides, ns, outs, count, 0, stream);
}

// outs[d] += sum of the partials: further contributions to gradients an earlier dig3d_reduce_many already wrote (a weight
// that enters the autograd graph twice — forward node and double-backward node of the force path).
int dig3d_reduce_many_acc(const void* const* parts, const int* nparts, const int64_t* strides, const int* ns,
                          void* const* outs, int count, void* stream) {
  return reduce_many_impl(parts, nparts, strides, ns, outs, count, 1, stream);
}

}  // extern "C"

extern "C" {

// G <= 8 layers of identical shape in one launch; X/W/bias/res/Y/Z are HOST arrays of G device pointers (bias, res, Z
// entries may be NULL).  Same semantics per layer as dig3d_linear_fwd.
int dig3d_linear_fwd_grouped(int G, const void* const* X, const void* const* W, const void* const* bias,
                             const void* const* res, int M, int K, int N, int act, void* const* Y, void* const* Z,
                             void* stream) {
  DIG3D_ENTER();
  const bool d2 = act >= ACT_D2;       // second-order epilogue: bias slot = z0 [M,N], res slot = gy0 [M,N], Z = 2nd output
  const bool rsc = act == ACT_ROWSCALE;  // Y_g = rs_g[m] * (x_g W_g^T + b_g), rs_g [M] in the res slot, Z_g receives rs (as dig3d_linear_fwd_rowscale)
  if (G < 1 || G > GRP_MAX || M < 0 || !dig3d_linear_supported(K, N) || !X || !W || !Y || act < 0 || (act > 2 && !d2 && !rsc) ||
      act > ACT_D2 + 2 || ((d2 || rsc) && (!res || !Z)) || (d2 && !bias))
    return DIG3D_ERR_ARG;
  if (M == 0) return DIG3D_OK;
  GroupFwd d;
  for (int g = 0; g < G; ++g) {
    d.X[g] = (const float*)X[g];
    d.W[g] = (const float*)W[g];
    d.bias[g] = bias ? (const float*)bias[g] : nullptr;
    d.res[g] = res ? (const float*)res[g] : nullptr;
    d.Y[g] = (float*)Y[g];
    d.Z[g] = Z ? (float*)Z[g] : nullptr;
    if (!d.X[g] || !d.W[g] || !d.Y[g] || !al16(d.X[g]) || !al16(d.W[g])) return DIG3D_ERR_ARG;
    if (rsc && (!d.res[g] || !d.Z[g])) return DIG3D_ERR_ARG;
    if (((uintptr_t)d.Y[g] | (uintptr_t)d.Z[g] | (uintptr_t)d.res[g] | (uintptr_t)d.bias[g]) & 15) return DIG3D_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  if (N > 64) {
    dim3 grid((M + 63) / 64, (N + 127) / 128, G);
    hipLaunchKernelGGL((k_linear_fwd_grouped<4>), grid, dim3(NTH), 0, st, d, M, K, N, act);
  } else if (N > 32) {
    dim3 grid((M + 127) / 128, 1, G);
    hipLaunchKernelGGL((k_linear_fwd_grouped<2>), grid, dim3(NTH), 0, st, d, M, K, N, act);
  } else {
    dim3 grid((M + 255) / 256, 1, G);
    hipLaunchKernelGGL((k_linear_fwd_grouped<1>), grid, dim3(NTH), 0, st, d, M, K, N, act);
  }
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// Both gradients of G same-shape layers in one launch (+ ONE reduction launch when reduce_now).  Host arrays of G
// device pointers; part[g]: float[dig3d_linear_bwd_workers(M,K,N) * (N*K+N)], gWb[g]: float[N*K+N].
int dig3d_linear_bwd_grouped(int G, const void* const* gY, const void* const* Z, const void* const* W,
                             const void* const* X, int M, int K, int N, int act, void* const* gX,
                             const void* const* gx_add, void* const* part, void* const* gWb, int reduce_now,
                             const void* const* gz_add, void* stream) {
  DIG3D_ENTER();
  if (G < 1 || G > GRP_MAX || M < 0 || !dig3d_linear_supported(K, N) || !gY || !W || !X || !part || !gWb)
    return DIG3D_ERR_ARG;                  // gX == NULL: no layer's input needs a gradient (the input-gradient tiles are not launched)
  hipStream_t st = (hipStream_t)stream;
  const int64_t stride = (int64_t)N * K + N;
  if (M == 0) {
    for (int g = 0; g < G; ++g)
      if (dig3d_zero_async(gWb[g], sizeof(float) * (size_t)stride, st) != hipSuccess) return DIG3D_ERR_LAUNCH;
    return DIG3D_OK;
  }
  GroupBwd d;
  for (int g = 0; g < G; ++g) {
    d.gY[g] = (const float*)gY[g];
    d.Z[g] = Z ? (const float*)Z[g] : nullptr;
    d.W[g] = (const float*)W[g];
    d.X[g] = (const float*)X[g];
    d.gX[g] = gX ? (float*)gX[g] : nullptr;
    d.gAdd[g] = gx_add ? (const float*)gx_add[g] : nullptr;
    d.part[g] = (float*)part[g];
    d.gZa[g] = gz_add ? (const float*)gz_add[g] : nullptr;
    if (!d.gY[g] || !d.W[g] || !d.X[g] || (gX && !d.gX[g]) || !d.part[g] || !gWb[g] || (act != 0 && !d.Z[g])) return DIG3D_ERR_ARG;
    if (!al16(d.gY[g]) || !al16(d.Z[g]) || !al16(d.W[g]) || !al16(d.X[g]) || !al16(d.gZa[g])) return DIG3D_ERR_ARG;
  }
  // one wave of blocks per layer would leave CUs idle at N_atoms rows: workers sized as for a single layer
  int nb = dig3d_linear_wgrad_blocks(M);
  const int dg = gX ? ((M + 63) / 64) * ((K + 127) / 128) : 0;
  const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
  const int wg = nb * tiles;
  hipLaunchKernelGGL(k_linear_bwd_both_grouped, dim3(wg + dg, G), dim3(NTH), 0, st, d, M, K, N, act, nb, wg);
  DIG3D_CHECK_LAUNCH();
  if (reduce_now) {
    ReduceTable t;
    t.accumulate = 0;
    for (int g = 0; g < G; ++g) {
      t.part[g] = d.part[g];
      t.out[g] = (float*)gWb[g];
      t.stride[g] = stride;
      t.nparts[g] = nb;
      t.n[g] = (int)stride;
    }
    int bx = (int)((stride + 15) / 16);
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(k_reduce_many, dim3(bx, G), dim3(64 * RED_KG), 0, st, t);
    DIG3D_CHECK_LAUNCH();
  }
  return DIG3D_OK;
}

// gX_g = (gY_g * act'(Z_g)) W_g for G same-shape layers in one launch.
int dig3d_linear_bwd_input_grouped(int G, const void* const* gY, const void* const* Z, const void* const* W, int M, int K,
                                   int N, int act, void* const* gX, void* stream) {
  DIG3D_ENTER();
  if (G < 1 || G > GRP_MAX || M < 0 || !dig3d_linear_supported(K, N) || !gY || !W || !gX) return DIG3D_ERR_ARG;
  if (M == 0) return DIG3D_OK;
  GroupBwd d;
  for (int g = 0; g < G; ++g) {
    d.gY[g] = (const float*)gY[g];
    d.Z[g] = Z ? (const float*)Z[g] : nullptr;
    d.W[g] = (const float*)W[g];
    d.X[g] = nullptr;
    d.gX[g] = (float*)gX[g];
    d.gAdd[g] = nullptr;
    d.part[g] = nullptr;
    d.gZa[g] = nullptr;
    if (!d.gY[g] || !d.W[g] || !d.gX[g] || (act != 0 && !d.Z[g])) return DIG3D_ERR_ARG;
    if (!al16(d.gY[g]) || !al16(d.Z[g]) || !al16(d.W[g])) return DIG3D_ERR_ARG;
  }
  dim3 grid((M + 63) / 64, (K + 127) / 128, G);
  hipLaunchKernelGGL(k_linear_bwd_input_grouped, grid, dim3(NTH), 0, (hipStream_t)stream, d, M, K, N, act);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// dig3d_linear_dd for G same-shape layers (N > 64); part[g]: float[dig3d_linear_wgrad_blocks(M) * (N*K+N)].
int dig3d_linear_dd_grouped(int G, const void* const* ggx, const void* const* W, const void* const* Z,
                            const void* const* gy, int M, int K, int N, int act, void* const* o_gy, void* const* o_z,
                            void* const* part, void* const* gWb, int reduce_now, void* stream) {
  DIG3D_ENTER();
  if (G < 1 || G > GRP_MAX || M < 0 || !dig3d_linear_supported(K, N) || N <= 64 || !ggx || !W || !gy || !o_gy || !part ||
      !gWb || act < 0 || act > 2 || (act != 0 && (!Z || !o_z)))
    return DIG3D_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t stride = (int64_t)N * K + N;
  if (M == 0) {
    for (int g = 0; g < G; ++g)
      if (dig3d_zero_async(gWb[g], sizeof(float) * (size_t)stride, st) != hipSuccess) return DIG3D_ERR_LAUNCH;
    return DIG3D_OK;
  }
  GroupDD d;
  for (int g = 0; g < G; ++g) {
    d.ggx[g] = (const float*)ggx[g];
    d.W[g] = (const float*)W[g];
    d.Z[g] = Z ? (const float*)Z[g] : nullptr;
    d.gy[g] = (const float*)gy[g];
    d.o_gy[g] = (float*)o_gy[g];
    d.o_z[g] = o_z ? (float*)o_z[g] : nullptr;
    d.part[g] = (float*)part[g];
    if (!d.ggx[g] || !d.W[g] || !d.gy[g] || !d.o_gy[g] || !d.part[g] || !gWb[g] || (act != 0 && (!d.Z[g] || !d.o_z[g])))
      return DIG3D_ERR_ARG;
    if (!al16(d.ggx[g]) || !al16(d.W[g]) || !al16(d.Z[g]) || !al16(d.gy[g]) || !al16(d.o_gy[g]) || !al16(d.o_z[g]))
      return DIG3D_ERR_ARG;
  }
  const int nb = dig3d_linear_wgrad_blocks(M);
  const int fw = ((M + 63) / 64) * ((N + 127) / 128);
  const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
  const int wg = nb * tiles;
  hipLaunchKernelGGL(k_linear_dd_grouped, dim3(wg + fw, G), dim3(NTH), 0, st, d, M, K, N, act, nb, wg);
  DIG3D_CHECK_LAUNCH();
  if (reduce_now) {
    ReduceTable t;
    t.accumulate = 0;
    for (int g = 0; g < G; ++g) {
      t.part[g] = d.part[g];
      t.out[g] = (float*)gWb[g];
      t.stride[g] = stride;
      t.nparts[g] = nb;
      t.n[g] = N * K;
    }
    int bx = (int)((stride + 15) / 16);
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(k_reduce_many, dim3(bx, G), dim3(64 * RED_KG), 0, st, t);
    DIG3D_CHECK_LAUNCH();
  }
  return DIG3D_OK;
}

}  // extern "C"

// ================================================================================================
// Chain of up to 8 hidden-width layers on one 64-row tile that never leaves the CU:
//     Y_l = res_l + act_l(Y_{l-1} W_l^T + b_l),   res_l in { none, external tensor, the tile saved by an earlier layer }
// (spherenet.py:172-182: lin_up + skip, ResidualLayer before the skip, lin + skip, two ResidualLayers after).
// Per layer only W_l (64 KB, L2-resident) is read and Z_l / Y_l (kept for the backward) are written; the activation
// tile and the skip tile stay in LDS, the next layer's weights are fetched into registers under the current MFMAs.
// All layers have N = 128 outputs; K_0 <= 128 (multiple of 8), K_l = 128 afterwards.
// ================================================================================================
// DD = false: the forward.  DD = true: the BACKWARD OF THE FIRST BACKWARD pass (energy_and_force: the force is a
// gradient, the loss differentiates through it).  With h = the incoming gradient w.r.t. the chain-input gradient, the
// adjoint of   gZ_l = gtot_l * act'(Z_l),  g_{l-1} = gZ_l W_l  (+ skip)   runs in FORWARD layer order with the forward's
// GEMM and skip pattern:   t_l = h_{l-1} W_l^T,   u_l = t_l act'(Z_l) (+ residual terms),   dL/dZ_l = t_l gtot_l act''(Z_l),
// h_l = u_l.  Y[l] receives u_l (operand of the weight gradient gZ_l^T h_{l-1}), Z[l] receives dL/dZ_l.
template <bool DD>
__global__ void __launch_bounds__(NTH) k_chain_fwd(const float* __restrict__ X0, int M, ChainDesc d) {
  extern __shared__ float csm[];
  float* sA = csm;                       // [64][132] current input tile
  float* sS = csm + 64 * DBKP;           // [64][132] saved skip tile
  float* sW = csm + 128 * DBKP;          // [128][132] weights of the current layer; reused as the output scratch
  const int m0 = blockIdx.x * 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 2, wn = wave & 3, i = lane & 31, h = lane >> 5;
  const int tr = threadIdx.x >> 5, tc = (threadIdx.x & 31) * 4;
  float4 rw[8], rz0[4], rg0[4];
  auto fetchW = [&](int l) {
    const float* __restrict__ W = d.W[l];
    const int K = d.K[l];
#pragma unroll
    for (int it = 0; it < 8; ++it) rw[it] = ld4(W, K, tr + 16 * it, 128, tc, K, true);
    if (DD) {                             // the layer's saved tiles travel with its weights
      const float* __restrict__ Z0 = d.Z0[l];
      const float* __restrict__ G0 = d.G0[l];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        rz0[it] = ld4(Z0, 128, m0 + tr + 16 * it, M, tc, 128, true);
        rg0[it] = ld4(G0, 128, m0 + tr + 16 * it, M, tc, 128, true);
      }
    }
  };
  // stage the input tile and the first weights
  {
    const int K0 = d.K[0];
#pragma unroll
    for (int it = 0; it < 4; ++it)
      *(float4*)(sA + (tr + 16 * it) * DBKP + tc) = ld4(X0, K0, m0 + tr + 16 * it, M, tc, K0, true);
  }
  fetchW(0);
  for (int l = 0; l < d.nl; ++l) {
#pragma unroll
    for (int it = 0; it < 8; ++it) *(float4*)(sW + (tr + 16 * it) * DBKP + tc) = rw[it];
    __syncthreads();
    float4 cz0[4], cg0[4];                // this layer's saved tiles (the registers are refilled for layer l + 1)
    if (DD) {
#pragma unroll
      for (int it = 0; it < 4; ++it) cz0[it] = rz0[it], cg0[it] = rg0[it];
    }
    if (l + 1 < d.nl) fetchW(l + 1);
    f32x16 acc = zero16();
    {
      const int kq = d.K[l] >> 3;
      const float* pa = sA + (wm * 32 + i) * DBKP + 4 * h;
      const float* pb = sW + (wn * 32 + i) * DBKP + 4 * h;
      for (int q = 0; q < kq; ++q) {
        const float4 a = *(const float4*)(pa + 8 * q);
        const float4 b = *(const float4*)(pb + 8 * q);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
      }
    }
    __syncthreads();                      // every wave is done with sA and sW
    float* sO = sW;
#pragma unroll
    for (int r = 0; r < 16; ++r) sO[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * DBKP + wn * 32 + i] = acc[r];
    __syncthreads();
    const int act = d.act[l], res = d.res[l], save = d.save[l];
    const float* __restrict__ bias = d.bias[l];
    const float* __restrict__ rx = d.resext[l];
    float* __restrict__ Zo = d.Z[l];
    float* __restrict__ Yo = d.Y[l];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = tr + 16 * it;
      const int m = m0 + r;
      float4 z = *(const float4*)(sO + r * DBKP + tc);
      if (bias) {
        const float4 bv = *(const float4*)(bias + tc);
        z.x += bv.x; z.y += bv.y; z.z += bv.z; z.w += bv.w;
      }
      float4 y;
      if (DD) {
        float d1, d2;
        const float4 t = z;
        act_d12(cz0[it].x, act, d1, d2); y.x = t.x * d1; z.x = t.x * cg0[it].x * d2;
        act_d12(cz0[it].y, act, d1, d2); y.y = t.y * d1; z.y = t.y * cg0[it].y * d2;
        act_d12(cz0[it].z, act, d1, d2); y.z = t.z * d1; z.z = t.z * cg0[it].z * d2;
        act_d12(cz0[it].w, act, d1, d2); y.w = t.w * d1; z.w = t.w * cg0[it].w * d2;
      } else {
        y = make_float4(act_fwd(z.x, act), act_fwd(z.y, act), act_fwd(z.z, act), act_fwd(z.w, act));
      }
      const int64_t o = (int64_t)m * 128 + tc;
      if (res == 1) {
        if (m < M && rx) {
          const float4 rv = *(const float4*)(rx + o);
          y.x = rv.x + y.x; y.y = rv.y + y.y; y.z = rv.z + y.z; y.w = rv.w + y.w;
        }
      } else if (res == 2) {
        const float4 rv = *(const float4*)(sS + r * DBKP + tc);
        y.x = rv.x + y.x; y.y = rv.y + y.y; y.z = rv.z + y.z; y.w = rv.w + y.w;
      }
      if (m < M) {
        if (Zo) *(float4*)(Zo + o) = z;
        *(float4*)(Yo + o) = y;
      }
      *(float4*)(sA + r * DBKP + tc) = y;            // input of the next layer (same thread owns this slot)
      if (save) *(float4*)(sS + r * DBKP + tc) = y;
    }
    __syncthreads();                      // sO (= sW) consumed, sA / sS complete
  }
}

extern "C" {

// Forward of a chain of nl <= 8 layers with 128 outputs each (see k_chain_fwd).  Host arrays of length nl:
// W[l] [128,K_l], bias[l] (or NULL), resext[l] (external residual [M,128] or NULL), Z[l] (or NULL), Y[l], K[l],
// res[l] (0 none / 1 external / 2 saved tile), save[l], act[l].  K_0 % 8 == 0, K_0 <= 128, K_l = 128 for l > 0.
static int chain_fwd_impl(const float* X0, int M, int nl, const void* const* W, const void* const* bias,
                          const void* const* resext, void* const* Z, void* const* Y, const int* K, const int* res,
                          const int* save, const int* act, const void* const* Z0, const void* const* G0, void* stream) {
  DIG3D_ENTER();
  const bool dd = Z0 != nullptr;
  if (M < 0 || nl < 1 || nl > CH_MAX || !X0 || !W || !Y || !K || !res || !save || !act || !Z || (dd && !G0))
    return DIG3D_ERR_ARG;
  if (M == 0) return DIG3D_OK;
  ChainDesc d;
  for (int l = 0; l < nl; ++l) {
    if (!W[l] || !Y[l] || K[l] <= 0 || K[l] > 128 || (K[l] & 7) || (l > 0 && K[l] != 128)) return DIG3D_ERR_ARG;
    if (act[l] != ACT_NONE && act[l] != ACT_SWISH) return DIG3D_ERR_ARG;
    if (res[l] == 1 && !dd && !(resext && resext[l])) return DIG3D_ERR_ARG;
    if (res[l] == 2 && l == 0) return DIG3D_ERR_ARG;
    if (dd && (!Z0[l] || !G0[l] || !Z[l] || !al16(Z0[l]) || !al16(G0[l]))) return DIG3D_ERR_ARG;
    d.W[l] = (const float*)W[l];
    d.bias[l] = bias ? (const float*)bias[l] : nullptr;
    d.resext[l] = resext ? (const float*)resext[l] : nullptr;
    d.Z[l] = (float*)Z[l];
    d.Y[l] = (float*)Y[l];
    d.K[l] = K[l];
    d.res[l] = res[l];
    d.save[l] = save[l];
    d.act[l] = act[l];
    d.Z0[l] = dd ? (const float*)Z0[l] : nullptr;
    d.G0[l] = dd ? (const float*)G0[l] : nullptr;
    if (!al16(d.W[l]) || !al16(d.bias[l]) || !al16(d.resext[l]) || !al16(d.Z[l]) || !al16(d.Y[l])) return DIG3D_ERR_ARG;
  }
  d.nl = nl;
  const size_t shm = sizeof(float) * 256 * DBKP;        // 135 KB
  if (dd) {
    static const bool attr_ok = hipFuncSetAttribute((const void*)k_chain_fwd<true>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int)(sizeof(float) * 256 * DBKP)) == hipSuccess;   // set once
    if (!attr_ok) return DIG3D_ERR_LAUNCH;
    hipLaunchKernelGGL(k_chain_fwd<true>, dim3((M + 63) / 64), dim3(NTH), shm, (hipStream_t)stream, X0, M, d);
  } else {
    static const bool attr_ok = hipFuncSetAttribute((const void*)k_chain_fwd<false>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int)(sizeof(float) * 256 * DBKP)) == hipSuccess;   // set once
    if (!attr_ok) return DIG3D_ERR_LAUNCH;
    hipLaunchKernelGGL(k_chain_fwd<false>, dim3((M + 63) / 64), dim3(NTH), shm, (hipStream_t)stream, X0, M, d);
  }
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_chain_fwd(const float* X0, int M, int nl, const void* const* W, const void* const* bias,
                    const void* const* resext, void* const* Z, void* const* Y, const int* K, const int* res,
                    const int* save, const int* act, void* stream) {
  return chain_fwd_impl(X0, M, nl, W, bias, resext, Z, Y, K, res, save, act, nullptr, nullptr, stream);
}

// Backward of dig3d_chain_bwd (the second-order pass of energy_and_force; see k_chain_fwd<true>).  H0 [M,K[0]]: gradient
// w.r.t. the gx0 output of dig3d_chain_bwd; ggres[l] [M,128] (or NULL): gradient w.r.t. its gres[l] output (res[l] == 1);
// Z0[l], G0[l]: the saved pre-activation and the total gradient G[l] stored by dig3d_chain_bwd.  Out: U[l] [M,128] = the
// gradient w.r.t. the total gradient of layer l (U[nl-1] = gradient w.r.t. gout; U[l-1] = the operand of layer l's
// weight gradient GZ[l]^T U[l-1], with H0 for l = 0) and HZ[l] [M,128] = gradient w.r.t. Z[l].
int dig3d_chain_dd(const float* H0, int M, int nl, const void* const* W, const void* const* Z0, const void* const* G0,
                   const void* const* ggres, void* const* HZ, void* const* U, const int* K, const int* res,
                   const int* save, const int* act, void* stream) {
  if (!Z0 || !G0 || !HZ) return DIG3D_ERR_ARG;
  return chain_fwd_impl(H0, M, nl, W, nullptr, ggres, HZ, U, K, res, save, act, Z0, G0, stream);
}

}  // extern "C"

// ================================================================================================
// Backward of that chain, two launches instead of one merged dgrad+wgrad launch per layer:
//  (1) k_chain_bwd: the INPUT-gradient recursion on the 64-row tile, layers in reverse order, the gradient tile and the
//      skip-connection accumulator never leaving LDS:   gZ_l = g_l * act'(Z_l),   g_{l-1} = gZ_l W_l  (+ skip tile).
//      Per layer it reads Z_l and W_l (prefetched into registers under the previous layer's MFMAs) and writes gZ_l —
//      the operand of the weight gradient — plus the gradient of an external residual where the layer has one.
//  (2) k_chain_wgrad: gW_l = gZ_l^T Y_{l-1}, gb_l = column sums, for ALL layers in one launch (blockIdx.z = layer),
//      256 / nl row-chunk workers per layer: 8 x 32 partials per chain instead of 8 x ~120.
// ================================================================================================
__global__ void __launch_bounds__(NTH) k_chain_bwd(const float* __restrict__ gout, int M, ChainBwdDesc d,
                                                    float* __restrict__ gx0) {
  extern __shared__ float csm[];
  float* sG = csm;                       // [64][132] gZ of the current layer (MFMA A operand)
  float* sS = csm + 64 * DBKP;           // [64][132] gradient waiting for the layer whose output was the skip tile
  float* sW = csm + 128 * DBKP;          // [128 n][132 k] weights of the current layer; reused as the output scratch
  const int m0 = blockIdx.x * 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 2, wk = wave & 3, i = lane & 31, h = lane >> 5;
  const int tr = threadIdx.x >> 5, tc = (threadIdx.x & 31) * 4;
  float4 rw[8], rz[4], rg[4], ra[4];
  auto fetch = [&](int l) {
    const float* __restrict__ W = d.W[l];
    const int K = d.K[l];
#pragma unroll
    for (int it = 0; it < 8; ++it) rw[it] = ld4(W, K, tr + 16 * it, 128, tc, K, true);
    if (d.act[l] != ACT_NONE) {
      const float* __restrict__ Z = d.Z[l];
#pragma unroll
      for (int it = 0; it < 4; ++it) rz[it] = ld4(Z, 128, m0 + tr + 16 * it, M, tc, 128, true);
    }
    if (d.gzadd[l]) {
      const float* __restrict__ A = d.gzadd[l];
#pragma unroll
      for (int it = 0; it < 4; ++it) ra[it] = ld4(A, 128, m0 + tr + 16 * it, M, tc, 128, true);
    }
  };
  const int nl = d.nl;
#pragma unroll
  for (int it = 0; it < 4; ++it) rg[it] = ld4(gout, 128, m0 + tr + 16 * it, M, tc, 128, true);
  fetch(nl - 1);
  bool pending = false;                  // sS holds a gradient for the most recent saved tile (uniform)
  for (int l = nl - 1; l >= 0; --l) {
    const int act = d.act[l], res = d.res[l], K = d.K[l];
    float* __restrict__ GZ = d.GZ[l];
    float* __restrict__ gr = d.gres[l];
    float* __restrict__ Gt = d.G[l];
    const bool zadd = d.gzadd[l] != nullptr;
    // elementwise stage: every thread owns the same (row, 4 columns) slots of sG / sS in every layer
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = tr + 16 * it, m = m0 + r;
      float4 g = rg[it];
      float* ps = sS + r * DBKP + tc;
      if (d.save[l] && pending) g = f4sum(*(const float4*)ps, g);      // this layer's output was the skip tile
      if (res == 2) *(float4*)ps = (pending && !d.save[l]) ? f4sum(*(const float4*)ps, g) : g;
      const int64_t o = (int64_t)m * 128 + tc;
      if (res == 1 && m < M) *(float4*)(gr + o) = g;
      if (Gt && m < M) *(float4*)(Gt + o) = g;
      float4 gz = gz4(g, rz[it], act);
      if (zadd) gz = f4sum(gz, ra[it]);
      if (m < M) *(float4*)(GZ + o) = gz;
      *(float4*)(sG + r * DBKP + tc) = gz;
    }
    if (d.save[l]) pending = false;
    if (res == 2) pending = true;
    __syncthreads();                      // the previous layer's output scratch (= sW) has been read by everyone
#pragma unroll
    for (int it = 0; it < 8; ++it) *(float4*)(sW + (tr + 16 * it) * DBKP + tc) = rw[it];
    __syncthreads();
    if (l > 0) fetch(l - 1);
    f32x16 acc = zero16();
    const bool live = wk * 32 < K;        // K_0 may be < 128: the other waves have no output columns
    if (live) {
      const float* pa = sG + (wm * 32 + i) * DBKP + 4 * h;
      const float* pb = sW + (4 * h) * DBKP + wk * 32 + i;
      for (int q = 0; q < 16; ++q) {
        const float4 a = *(const float4*)(pa + 8 * q);
        const float* b = pb + (8 * q) * DBKP;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[DBKP], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[2 * DBKP], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[3 * DBKP], acc, 0, 0, 0);
      }
    }
    __syncthreads();                      // every wave is done with sW
    float* sO = sW;
    if (live) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sO[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * DBKP + wk * 32 + i] = acc[r];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) rg[it] = *(const float4*)(sO + (tr + 16 * it) * DBKP + tc);
    if (l == 0) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int m = m0 + tr + 16 * it;
        if (m < M && tc < K) *(float4*)(gx0 + (int64_t)m * K + tc) = rg[it];
      }
    }
  }
}

#define WG_MAX 64                // layers per weight-gradient launch (dig3d_wgrad_many: a whole backward pass)
struct ChainWgradDesc {
  const float* GZ[WG_MAX];
  const float* X[WG_MAX];        // input of layer l: X0 for l = 0, Y_{l-1} afterwards
  float* part[WG_MAX];           // [nworkers][N_l*K_l + N_l]
  int K[WG_MAX];
  int N[WG_MAX];                 // outputs of layer l (<= 128)
  int M[WG_MAX];                 // rows of layer l
};

__global__ void __launch_bounds__(NTH) k_chain_wgrad(ChainWgradDesc d, int nworkers) {
  __shared__ float smem[128 * DBKP];
  const int l = blockIdx.z;
  wgrad_body(d.GZ[l], nullptr, d.X[l], d.M[l], d.K[l], d.N[l], ACT_NONE, d.part[l], smem, blockIdx.x, 0, 0, nworkers);
}

// one 128 x 128 tile of one layer's weight gradient per blockIdx.z (dig3d_wgrad_many)
struct WgradManyDesc {
  const float* GY[WG_MAX];
  const float* Z[WG_MAX];        // pre-activation (act'(Z) applied while staging) or null
  const float* X[WG_MAX];
  float* part[WG_MAX];
  int M[WG_MAX], K[WG_MAX], N[WG_MAX], act[WG_MAX], wy[WG_MAX], wz[WG_MAX], nw[WG_MAX];
  int db;                        // alternate the staging buffers (wgrad_body)
};
static_assert(sizeof(WgradManyDesc) <= 4096, "kernel arguments");
__global__ void __launch_bounds__(NTH) k_wgrad_many(WgradManyDesc d) {
  __shared__ float smem[128 * DBKP];
  const int t = blockIdx.z;
  if ((int)blockIdx.x >= d.nw[t]) return;          // layers of few rows have fewer workers than the grid is wide
  wgrad_body(d.GY[t], d.Z[t], d.X[t], d.M[t], d.K[t], d.N[t], d.act[t], d.part[t], smem, blockIdx.x, d.wy[t], d.wz[t],
             d.nw[t], nullptr, d.db);
}

extern "C" {

// Input-gradient pass of the chain (see k_chain_bwd).  Host arrays of length nl as in dig3d_chain_fwd; GZ[l] [M,128]
// receives the pre-activation gradient of every layer, gres[l] [M,128] the gradient of layer l's external residual
// (res[l] == 1; NULL elsewhere), gx0 [M,K[0]] the gradient of the chain input.  Optional (NULL or arrays of length nl
// with NULL entries): G[l] [M,128] receives the total gradient w.r.t. layer l's output (dig3d_chain_dd needs it), gz_add[l]
// [M,128] is added to the pre-activation gradient (gZ_l = g_l act'(Z_l) + gz_add[l]).
int dig3d_chain_bwd(const float* gout, int M, int nl, const void* const* W, const void* const* Z, void* const* GZ,
                    void* const* gres, const int* K, const int* res, const int* save, const int* act, float* gx0,
                    void* const* G, const void* const* gz_add, void* stream) {
  DIG3D_ENTER();
  if (M < 0 || nl < 1 || nl > CH_MAX || !gout || !W || !Z || !GZ || !gres || !K || !res || !save || !act || !gx0)
    return DIG3D_ERR_ARG;
  if (M == 0) return DIG3D_OK;
  if (!al16(gout) || !al16(gx0)) return DIG3D_ERR_ARG;
  ChainBwdDesc d;
  for (int l = 0; l < nl; ++l) {
    if (!W[l] || !GZ[l] || K[l] <= 0 || K[l] > 128 || (K[l] & 7) || (l > 0 && K[l] != 128)) return DIG3D_ERR_ARG;
    if (act[l] != ACT_NONE && (!Z[l] || act[l] != ACT_SWISH)) return DIG3D_ERR_ARG;
    if (res[l] == 1 && !gres[l]) return DIG3D_ERR_ARG;
    if (res[l] == 2 && l == 0) return DIG3D_ERR_ARG;
    if (!al16(W[l]) || !al16(Z[l]) || !al16(GZ[l]) || !al16(gres[l])) return DIG3D_ERR_ARG;
    d.W[l] = (const float*)W[l];
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
  const size_t shm = sizeof(float) * 256 * DBKP;        // 135 KB
  static const bool attr_ok = hipFuncSetAttribute((const void*)k_chain_bwd, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)(sizeof(float) * 256 * DBKP)) == hipSuccess;   // set once
  if (!attr_ok) return DIG3D_ERR_LAUNCH;
  hipLaunchKernelGGL(k_chain_bwd, dim3((M + 63) / 64), dim3(NTH), shm, (hipStream_t)stream, gout, M, d, gx0);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// row-chunk workers per layer of dig3d_chain_wgrad (= partials it writes per layer)
// blocks of the launch (all layers): 256 at E ~ 8k rows, 512 (two per CU) from 32k rows — same-box A/B: config 4 (77k
// rows) 8.21 -> 8.15 ms with 512, config 2 (7.8k rows) 2.954 -> 2.967.  DIG3D_CHAIN_WGRAD_BLOCKS overrides (read once).
int dig3d_chain_wgrad_workers(int M, int nl) {
  if (nl < 1) nl = 1;
  const int chunks = (M + 31) / 32;
  int nb = (M >= 32768 ? 2 * dig3d_num_cus() : dig3d_num_cus()) / nl;
  if (nb < 8) nb = 8;
  if (nb > chunks) nb = chunks;
  return nb < 1 ? 1 : nb;
}

int dig3d_chain_wgrad_n(int nl, const void* const* GZ, const void* const* X, const int* K, const int* N, int M,
                        void* const* part, void* const* gWb, int reduce_now, void* stream);

// Weight / bias gradients of all nl layers in ONE launch: part[l] float[workers * (128*K[l] + 128)] receives the
// partials, gWb[l] float[128*K[l] + 128] their sum when reduce_now (else the caller reduces: dig3d_reduce_many).
int dig3d_chain_wgrad(int nl, const void* const* GZ, const void* const* X, const int* K, int M, void* const* part,
                      void* const* gWb, int reduce_now, void* stream) {
  return dig3d_chain_wgrad_n(nl, GZ, X, K, nullptr, M, part, gWb, reduce_now, stream);
}

// The same for layers with N[l] <= 128 outputs (N % 4 == 0; NULL: 128 everywhere): GZ[l] [M,N[l]], partial / gradient
// size N[l]*K[l] + N[l].  The three layers of an interaction block's front (dig3d_front_bwd) are one such launch.
int dig3d_chain_wgrad_n(int nl, const void* const* GZ, const void* const* X, const int* K, const int* N, int M,
                        void* const* part, void* const* gWb, int reduce_now, void* stream) {
  DIG3D_ENTER();
  if (M < 0 || nl < 1 || nl > CH_MAX || !GZ || !X || !K || !part || !gWb) return DIG3D_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  ChainWgradDesc d;
  for (int l = 0; l < nl; ++l) {
    const int n = N ? N[l] : 128;
    if (!GZ[l] || !X[l] || !part[l] || !gWb[l] || K[l] <= 0 || K[l] > 128 || (K[l] & 3) || n <= 0 || n > 128 || (n & 3))
      return DIG3D_ERR_ARG;
    if (!al16(GZ[l]) || !al16(X[l])) return DIG3D_ERR_ARG;
    d.GZ[l] = (const float*)GZ[l];
    d.X[l] = (const float*)X[l];
    d.part[l] = (float*)part[l];
    d.K[l] = K[l];
    d.N[l] = n;
    d.M[l] = M;
  }
  if (M == 0) {
    for (int l = 0; l < nl; ++l)
      if (dig3d_zero_async(gWb[l], sizeof(float) * (d.N[l] * (size_t)K[l] + d.N[l]), st) != hipSuccess) return DIG3D_ERR_LAUNCH;
    return DIG3D_OK;
  }
  const int nb = dig3d_chain_wgrad_workers(M, nl);
  hipLaunchKernelGGL(k_chain_wgrad, dim3(nb, 1, nl), dim3(NTH), 0, st, d, nb);
  DIG3D_CHECK_LAUNCH();
  if (reduce_now) {
    ReduceTable t;
    t.accumulate = 0;
    for (int l = 0; l < nl; ++l) {
      const int64_t stride = d.N[l] * (int64_t)K[l] + d.N[l];
      t.part[l] = d.part[l];
      t.out[l] = (float*)gWb[l];
      t.stride[l] = stride;
      t.nparts[l] = nb;
      t.n[l] = (int)stride;
    }
    hipLaunchKernelGGL(k_reduce_many, dim3(1024, nl), dim3(64 * RED_KG), 0, st, t);
    DIG3D_CHECK_LAUNCH();
  }
  return DIG3D_OK;
}

// Weight-gradient PARTIALS of MANY dense layers in one launch (a few for more than 64 tiles) — every dense layer of a
// whole backward pass (dig_amd/ops.py: deferred_reductions collects them; 45 layers for the default SphereNet, 51 of
// 256 x 256 for ComENet): layer l has GY[l] [M[l], N[l]] — the gradient w.r.t. the layer OUTPUT when Z[l] / act[l] are
// given (gZ = gY * act'(Z) is formed while staging), else already the pre-activation gradient — and X[l] [M[l], K[l]];
// N, K multiples of 4 of any size (tiles of 128 x 128).  part[l] float[nworkers[l] * (N[l]*K[l] + N[l])] (weights, then
// the bias column sums).  nworkers[l] row-chunk workers per tile of layer l (the caller sizes them by rows so that a
// launch holds ~2 blocks per CU of equal work): a layer writes that many partials instead of the 32 - 128 of a launch of
// its own — the reduction that follows (dig3d_reduce_many) reads a fraction of the bytes — and no layer's launch runs
// on a half-empty chip.
// (r05, measured and reverted: the launch reducing its OWN partials — the last worker of a tile to arrive, found through a
// per-tile arrival counter, sums the tile's partials in worker order; correct and deterministic (343 GPU tests green) but
// config 2 1.511 -> 1.820 ms, config 4 5.38 -> 5.77, config 5 6.92 -> 7.80 on one box: the partials come from other XCDs,
// so every worker needs a device-scope release (L2 write-back) before it bumps the counter and the last one an acquire (L2
// invalidate) — far dearer on an 8-L2 part than the 34-us reduction launch it removes.)
int dig3d_wgrad_many(int nl, const void* const* GY, const void* const* Z, const int* act, const void* const* X,
                     const int* K, const int* N, const int* M, const int* nworkers, void* const* part, int route,
                     void* stream) {
  DIG3D_ENTER();
  if (nl < 1 || !GY || !X || !K || !N || !M || !part || !nworkers || (route != 0 && route != 1)) return DIG3D_ERR_ARG;
  WgradManyDesc d;
  d.db = route;                // 1: alternating staging buffers, one barrier per chunk (same sums in the same order)
  int nt = 0, gx = 1;
  auto flush = [&]() {
    if (nt) hipLaunchKernelGGL(k_wgrad_many, dim3(gx, 1, nt), dim3(NTH), 0, (hipStream_t)stream, d);
    nt = 0;
    gx = 1;
  };
  for (int l = 0; l < nl; ++l) {
    const int a = (Z && act && Z[l]) ? act[l] : ACT_NONE;
    if (nworkers[l] < 1 || nworkers[l] > 65535) return DIG3D_ERR_ARG;
    if (!GY[l] || !X[l] || !part[l] || K[l] <= 0 || (K[l] & 3) || N[l] <= 0 || (N[l] & 3) || M[l] < 1 || !al16(GY[l]) ||
        !al16(X[l]) || (a != ACT_NONE && !al16(Z[l])) || (a != ACT_NONE && a != ACT_SWISH && a != ACT_SSP && a != ACT_DERIV))
      return DIG3D_ERR_ARG;
    for (int wy = 0; wy * 128 < N[l]; ++wy)
      for (int wz = 0; wz * 128 < K[l]; ++wz) {
        d.GY[nt] = (const float*)GY[l];
        d.Z[nt] = a != ACT_NONE ? (const float*)Z[l] : nullptr;
        d.X[nt] = (const float*)X[l];
        d.part[nt] = (float*)part[l];
        d.M[nt] = M[l]; d.K[nt] = K[l]; d.N[nt] = N[l]; d.act[nt] = a; d.wy[nt] = wy; d.wz[nt] = wz;
        d.nw[nt] = nworkers[l];
        if (nworkers[l] > gx) gx = nworkers[l];
        if (++nt == WG_MAX) flush();
      }
  }
  flush();
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

}  // extern "C"

// ================================================================================================
// Adam on FLAT buffers (run.py:50 `Adam(model.parameters(), lr, weight_decay)`): one elementwise pass over all
// parameters instead of a multi-tensor launch over 135 small tensors (255 us -> ~10 us per step for SphereNet).
// torch.optim.Adam arithmetic (single-tensor path): m = lerp(m, g, 1-b1); v = b2 v + (1-b2) g^2;
// p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps);  weight decay is added to the gradient first.
// ================================================================================================
__global__ void k_adam_flat(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                            float4* __restrict__ v, int64_t n4, float one_m_b1, float b2, float one_m_b2,
                            float step_size, float sqrt_bc2, float eps, float wd) {
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
    float4 pp = p[q], gg = g[q], mm = m[q], vv = v[q];
    float* P = &pp.x; float* G = &gg.x; float* Mm = &mm.x; float* V = &vv.x;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // the operation sequence of torch.optim.Adam (_multi_tensor_adam): lerp, mul + addcmul, sqrt / sqrt(bc2) + eps, addcdiv
      float gr = G[c];
      if (wd != 0.f) gr = gr + wd * P[c];
      Mm[c] = Mm[c] + (gr - Mm[c]) * one_m_b1;
      V[c] = V[c] * b2 + one_m_b2 * (gr * gr);
      const float denom = sqrtf(V[c]) / sqrt_bc2 + eps;
      P[c] = P[c] - step_size * (Mm[c] / denom);
    }
    p[q] = pp; m[q] = mm; v[q] = vv;
  }
}

extern "C" {

// n must be a multiple of 4 (the host pads the flat buffers); all four buffers 16-byte aligned.
int dig3d_adam_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                    double beta1, double beta2, double eps, double weight_decay, double bias_correction1,
                    double bias_correction2, void* stream) {
  DIG3D_ENTER();
  if (n < 0 || (n & 3) || !param || !grad || !exp_avg || !exp_avg_sq) return DIG3D_ERR_ARG;
  if (!al16(param) || !al16(grad) || !al16(exp_avg) || !al16(exp_avg_sq)) return DIG3D_ERR_ARG;
  if (n == 0) return DIG3D_OK;
  const int64_t n4 = n >> 2;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  // every derived scalar in DOUBLE, one rounding each (as torch's Python floats): 1.0f - 0.999f would be 0.000999987
  hipLaunchKernelGGL(k_adam_flat, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float4*)param,
                     (const float4*)grad, (float4*)exp_avg, (float4*)exp_avg_sq, n4, (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), (float)(lr / bias_correction1), (float)sqrt(bias_correction2), (float)eps,
                     (float)weight_decay);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

}  // extern "C"

// ================================================================================================
// Small-K layers (K <= 8): the radial-basis projections lin_rbf*(rbf) with K = num_radial = 6 or basis_emb_size = 8
// (spherenet.py:86-90,153-155,182).  2*K MACs per output: no tiles, no LDS for the forward, one row tile in LDS
// for the backward.  The 100-KB-LDS MFMA kernels spent 12 + 16 + 5 us per layer on them.
// ================================================================================================
#define SK_MAX 16         // widest small-K layer; the kernels are instantiated for 8 (the common case) and 16
// W is staged k-major in LDS (sW[k][n]): lanes that own consecutive output columns then read consecutive words
// (the row-major W[n][k] read straight from global cost one L1 transaction per lane: 12 us per layer).
template <int SKM>
__global__ void __launch_bounds__(256) k_smallk_fwd(const float* __restrict__ X, const float* __restrict__ W,
                                                     const float* __restrict__ bias, const float* __restrict__ res,
                                                     int M, int K, int N, int act, float* __restrict__ Y,
                                                     float* __restrict__ Z) {
  __shared__ float sW[SKM * 256];
  const bool keepd = act & ACT_KEEP_DERIV;           // Z receives act'(z)
  act &= 3;
  for (int q = threadIdx.x; q < N * K; q += 256) {
    const int n = q / K, k = q - n * K;
    sW[k * N + n] = W[q];
  }
  __syncthreads();
  const int n4 = N >> 2;
  const int64_t total = (int64_t)M * n4;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
    const int m = (int)(q / n4), n = (int)(q - (int64_t)m * n4) * 4;
    float4 z = bias ? *(const float4*)(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < SKM; ++k) {
      if (k < K) {
        const float x = X[(int64_t)m * K + k];
        const float4 w = *(const float4*)(sW + k * N + n);
        z.x = fmaf(x, w.x, z.x); z.y = fmaf(x, w.y, z.y); z.z = fmaf(x, w.z, z.z); z.w = fmaf(x, w.w, z.w);
      }
    }
    const int64_t o = (int64_t)m * N + n;
    if (Z)
      *(float4*)(Z + o) = keepd ? make_float4(act_bwd(z.x, act), act_bwd(z.y, act), act_bwd(z.z, act), act_bwd(z.w, act)) : z;
    float4 y = make_float4(act_fwd(z.x, act), act_fwd(z.y, act), act_fwd(z.z, act), act_fwd(z.w, act));
    if (res) {
      const float4 r = *(const float4*)(res + o);
      y.x = r.x + y.x; y.y = r.y + y.y; y.z = r.z + y.z; y.w = r.w + y.w;
    }
    *(float4*)(Y + o) = y;
  }
}

// backward: 32-row tiles; gZ tile [32][N<=256] in LDS.
//   gX[m,k] = sum_n gZ[m,n] W[n,k]     8 threads per row, each over N/8 columns, xor-shuffle reduction
//   gW[n,k], gb[n]                      thread n (+256 per pass) accumulates over the tile rows in registers;
//                                       block partial -> part[blockIdx][N*K + N] -> k_dense_reduce
template <int SKM>
__global__ void __launch_bounds__(256) k_smallk_bwd(const float* __restrict__ gY, const float* __restrict__ Zp,
                                                     const float* __restrict__ W, const float* __restrict__ X,
                                                     int M, int K, int N, int act, float* __restrict__ gX,
                                                     const float* __restrict__ gAdd, float* __restrict__ part) {
  __shared__ float sG[32 * 260];
  __shared__ float sX[32 * SKM];
  __shared__ float sW[SKM * 256];          // k-major copy of W: conflict-free, coalesced reads along n
  for (int q = threadIdx.x; q < N * K; q += 256) {
    const int n = q / K, k = q - n * K;
    sW[k * N + n] = W[q];
  }
  const int NP = N + 4;
  const int ntiles = (M + 31) / 32;
  float gw[SKM], gb = 0.f;                 // this thread's weight-gradient row (n = threadIdx.x; N <= 256)
#pragma unroll
  for (int k = 0; k < SKM; ++k) gw[k] = 0.f;
  const int n4 = N >> 2;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m0 = tile * 32;
    __syncthreads();
    for (int q = threadIdx.x; q < 32 * n4; q += 256) {
      const int r = q / n4, c = (q - r * n4) * 4;
      const int m = m0 + r;
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M) {
        g = *(const float4*)(gY + (int64_t)m * N + c);
        if (act != ACT_NONE) g = gz4(g, *(const float4*)(Zp + (int64_t)m * N + c), act);
      }
      *(float4*)(sG + r * NP + c) = g;
    }
    for (int q = threadIdx.x; q < 32 * SKM; q += 256) {
      const int r = q / SKM, k = q - r * SKM;
      const int m = m0 + r;
      sX[q] = (m < M && k < K && X) ? X[(int64_t)m * K + k] : 0.f;
    }
    __syncthreads();
    if (gX) {                                 // 8 threads per row
      const int r = threadIdx.x >> 3, l8 = threadIdx.x & 7;
      float acc[SKM];
#pragma unroll
      for (int k = 0; k < SKM; ++k) acc[k] = 0.f;
      for (int n = l8; n < N; n += 8) {
        const float g = sG[r * NP + n];
#pragma unroll
        for (int k = 0; k < SKM; ++k)
          if (k < K) acc[k] = fmaf(g, sW[k * N + n], acc[k]);
      }
#pragma unroll
      for (int k = 0; k < SKM; ++k) {
        float v = acc[k];
        v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
        acc[k] = v;
      }
      const int m = m0 + r;
#pragma unroll
      for (int kb = 0; kb < SKM; kb += 8) {     // lane l8 of the row's 8 threads writes columns l8, l8 + 8
        const int kk = kb + l8;
        if (m < M && kk < K) {
          float v = 0.f;
#pragma unroll
          for (int k = 0; k < SKM; ++k)
            if (k == kk) v = acc[k];
          if (gAdd) v = gAdd[(int64_t)m * K + kk] + v;
          gX[(int64_t)m * K + kk] = v;
        }
      }
    }
    if (part && threadIdx.x < N) {
      const int n = threadIdx.x;
      for (int r = 0; r < 32; ++r) {
        const float g = sG[r * NP + n];
        gb += g;
#pragma unroll
        for (int k = 0; k < SKM; ++k) gw[k] = fmaf(g, sX[r * SKM + k], gw[k]);
      }
    }
  }
  if (part && threadIdx.x < N) {
    float* outp = part + (int64_t)blockIdx.x * ((int64_t)N * K + N);
    const int n = threadIdx.x;
#pragma unroll
    for (int k = 0; k < SKM; ++k)
      if (k < K) outp[(int64_t)n * K + k] = gw[k];
    outp[(int64_t)N * K + n] = gb;
  }
}

extern "C" {

int dig3d_smallk_supported(int K, int N) { return (K >= 1 && K <= SK_MAX && N >= 8 && N <= 256 && (N & 7) == 0) ? 1 : 0; }

int dig3d_smallk_blocks(int M) {
  int nt = (M + 31) / 32;
  if (nt > 256) nt = 256;
  return nt < 1 ? 1 : nt;
}

// Y = act(X W^T + b) (+ res) for K <= 16, N <= 256 (N % 8 == 0); same argument meaning as dig3d_linear_fwd.
int dig3d_smallk_fwd(const float* X, const float* W, const float* bias, const float* res, int M, int K, int N, int act,
                     float* Y, float* Z, void* stream) {
  DIG3D_ENTER();
  const bool keepd = act == (ACT_SWISH | ACT_KEEP_DERIV) || act == (ACT_SSP | ACT_KEEP_DERIV);
  if (M < 0 || !dig3d_smallk_supported(K, N) || !X || !W || !Y || act < 0 || (act > 2 && !keepd) || (keepd && !Z))
    return DIG3D_ERR_ARG;
  if (((uintptr_t)Y | (uintptr_t)Z | (uintptr_t)res | (uintptr_t)bias) & 15) return DIG3D_ERR_ARG;
  if (M == 0) return DIG3D_OK;
  int blocks = dig3d_blocks((int64_t)M * (N / 4), 256);
  if (blocks > 1024) blocks = 1024;          // grid-stride: the W staging is amortised over several row groups
  if (K <= 8)
    hipLaunchKernelGGL(k_smallk_fwd<8>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, X, W, bias, res, M, K, N, act, Y,
                       Z);
  else
    hipLaunchKernelGGL(k_smallk_fwd<16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, X, W, bias, res, M, K, N, act,
                       Y, Z);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// gX (or NULL) and gWb = {gW[N,K], gb[N]} (or NULL, then part may be NULL) in one launch + one reduction.
// part: float[dig3d_smallk_blocks(M) * (N*K + N)].
int dig3d_smallk_bwd(const float* gY, const float* Z, const float* W, const float* X, int M, int K, int N, int act,
                     float* gX, const float* gx_add, float* part, float* gWb, int reduce_now, void* stream) {
  DIG3D_ENTER();
  if (M < 0 || !dig3d_smallk_supported(K, N) || !gY || !W || (act != 0 && !Z) || (gWb && (!part || !X)))
    return DIG3D_ERR_ARG;
  if (!al16(gY) || !al16(Z)) return DIG3D_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t stride = (int64_t)N * K + N;
  if (M == 0) {
    if (gWb && dig3d_zero_async(gWb, sizeof(float) * (size_t)stride, st) != hipSuccess) return DIG3D_ERR_LAUNCH;
    return DIG3D_OK;
  }
  const int nb = dig3d_smallk_blocks(M);
  if (K <= 8)
    hipLaunchKernelGGL(k_smallk_bwd<8>, dim3(nb), dim3(256), 0, st, gY, Z, W, X, M, K, N, act, gX, gx_add,
                       gWb ? part : nullptr);
  else
    hipLaunchKernelGGL(k_smallk_bwd<16>, dim3(nb), dim3(256), 0, st, gY, Z, W, X, M, K, N, act, gX, gx_add,
                       gWb ? part : nullptr);
  DIG3D_CHECK_LAUNCH();
  if (gWb && reduce_now) {
    hipLaunchKernelGGL(k_dense_reduce, dim3(dig3d_blocks(stride, 16)), dim3(256), 0, st, part, nb, stride, (int)stride,
                       gWb);
    DIG3D_CHECK_LAUNCH();
  }
  return DIG3D_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Composed feature weights.  ComENet's TwoLayerLinear (method/comenet/comenet.py:87-105) without bias and activation is
// x -> lin2(lin1(x)) = x (W2 W1)^T; the engine applies the [hidden, K <= 16] product Wc = W2 W1 inside the convolution
// (segment.hip:k_featconv).  A step needs 2 x num_layers of them, all known before the first layer runs: ONE launch
// forms them all, ONE launch turns their gradients into those of the factors (the framework matmuls were 24 library
// launches of 8 - 11 us per step for 200 kFLOP each).
//   forward   Wc[n,k]  = sum_m W2[n,m] W1[m,k]
//   backward  gW2[n,m] = sum_k gWc[n,k] W1[m,k],   gW1[m,k] = sum_n W2[n,m] gWc[n,k]
// ------------------------------------------------------------------------------------------------
#define CMP_MAX 16
struct ComposeDesc {
  const float* W2[CMP_MAX];      // [N, Mid]
  const float* W1[CMP_MAX];      // [Mid, K]
  const float* G[CMP_MAX];       // backward: gWc [N, K]
  float* out[CMP_MAX];           // forward: Wc [N, K]
  float* gW2[CMP_MAX];
  float* gW1[CMP_MAX];
  int N[CMP_MAX], Mid[CMP_MAX], K[CMP_MAX];
};

__global__ void __launch_bounds__(256) k_compose_fwd(ComposeDesc d) {
  const int p = blockIdx.y, N = d.N[p], Mid = d.Mid[p], K = d.K[p];
  const float* __restrict__ W2 = d.W2[p];
  const float* __restrict__ W1 = d.W1[p];
  for (int q = blockIdx.x * 256 + threadIdx.x; q < N * K; q += gridDim.x * 256) {
    const int n = q / K, k = q - n * K;
    float s = 0.f;
    for (int m = 0; m < Mid; ++m) s = fmaf(W2[n * Mid + m], W1[m * K + k], s);
    d.out[p][q] = s;
  }
}

// gW2: one thread per element (K <= 16 terms).  gW1[m,k] sums N = 256 terms: the 64 lanes of a wave take n, n + 64, ...
// and reduce by shuffles (a thread per element ran 256 dependent strided loads: 99 us for 5 MFLOP).
__global__ void __launch_bounds__(256) k_compose_bwd(ComposeDesc d) {
  const int p = blockIdx.y, N = d.N[p], Mid = d.Mid[p], K = d.K[p];
  const float* __restrict__ W2 = d.W2[p];
  const float* __restrict__ W1 = d.W1[p];
  const float* __restrict__ G = d.G[p];
  for (int q = blockIdx.x * 256 + threadIdx.x; q < N * Mid; q += gridDim.x * 256) {
    const int n = q / Mid, m = q - n * Mid;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(G[n * K + k], W1[m * K + k], s);
    d.gW2[p][q] = s;
  }
  const int lane = threadIdx.x & 63;
  for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < Mid * K; r += gridDim.x * 4) {   // wave-uniform
    const int m = r / K, k = r - m * K;
    float s = 0.f;
    for (int n = lane; n < N; n += 64) s = fmaf(W2[n * Mid + m], G[n * K + k], s);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) d.gW1[p][r] = s;
  }
}

extern "C" {

// np <= 16 products out[p] [N[p], K[p]] = W2[p] [N[p], Mid[p]] · W1[p] [Mid[p], K[p]] in one launch (row-major, dense).
int dig3d_compose_fwd(int np, const void* const* W2, const void* const* W1, const int* N, const int* Mid, const int* K,
                      void* const* out, void* stream) {
  DIG3D_ENTER();
  if (np < 1 || np > CMP_MAX || !W2 || !W1 || !N || !Mid || !K || !out) return DIG3D_ERR_ARG;
  ComposeDesc d;
  int big = 1;
  for (int p = 0; p < np; ++p) {
    if (!W2[p] || !W1[p] || !out[p] || N[p] < 1 || Mid[p] < 1 || K[p] < 1 || (int64_t)N[p] * Mid[p] > (1 << 24)) return DIG3D_ERR_ARG;
    d.W2[p] = (const float*)W2[p];
    d.W1[p] = (const float*)W1[p];
    d.out[p] = (float*)out[p];
    d.N[p] = N[p];
    d.Mid[p] = Mid[p];
    d.K[p] = K[p];
    if (N[p] * K[p] > big) big = N[p] * K[p];
  }
  hipLaunchKernelGGL(k_compose_fwd, dim3(dig3d_blocks(big, 256), np), dim3(256), 0, (hipStream_t)stream, d);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// gradients of the factors from the gradients gWc[p] [N[p], K[p]] of the products: gW2[p] [N, Mid], gW1[p] [Mid, K].
int dig3d_compose_bwd(int np, const void* const* gWc, const void* const* W2, const void* const* W1, const int* N,
                      const int* Mid, const int* K, void* const* gW2, void* const* gW1, void* stream) {
  DIG3D_ENTER();
  if (np < 1 || np > CMP_MAX || !gWc || !W2 || !W1 || !N || !Mid || !K || !gW2 || !gW1) return DIG3D_ERR_ARG;
  ComposeDesc d;
  int big = 1;
  for (int p = 0; p < np; ++p) {
    if (!gWc[p] || !W2[p] || !W1[p] || !gW2[p] || !gW1[p] || N[p] < 1 || Mid[p] < 1 || K[p] < 1 ||
        (int64_t)N[p] * Mid[p] > (1 << 24))
      return DIG3D_ERR_ARG;
    d.G[p] = (const float*)gWc[p];
    d.W2[p] = (const float*)W2[p];
    d.W1[p] = (const float*)W1[p];
    d.gW2[p] = (float*)gW2[p];
    d.gW1[p] = (float*)gW1[p];
    d.N[p] = N[p];
    d.Mid[p] = Mid[p];
    d.K[p] = K[p];
    if (N[p] * Mid[p] > big) big = N[p] * Mid[p];
  }
  hipLaunchKernelGGL(k_compose_bwd, dim3(dig3d_blocks(big, 256), np), dim3(256), 0, (hipStream_t)stream, d);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

}  // extern "C"
