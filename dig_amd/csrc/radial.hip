// dig3d — every radial-basis projection of a SphereNet / DimeNet++ forward in ONE launch, and all of their backward
// passes in one more.
//
// rbf [E, nr] (nr = 6) is projected 2 + 3 L times per forward (method/spherenet/spherenet.py:86-90 init: lin_rbf_0 with
// bias + swish, lin_rbf_1; :153-155,182 per layer: lin_rbf2(lin_rbf1(rbf)) and lin_rbf(rbf)): 14 launches of a few
// microseconds forward, 14 backward, plus the autograd adds that sum the 10 gradients arriving at rbf.  Every "head"
// reads the same [E, nr] rows, so one kernel computes all outputs and one kernel all gradients (the rbf gradient
// summed in registers, weight gradients as per-tile partials for dig3d_reduce_many).
//   head kinds:  single   Y = act(X Wa^T + b)            Wa [N, K]
//                two-layer Y = (X Wa^T) Wb^T              Wa [J, K], Wb [N, J]   (no bias / activation between, :153-155)
#include "common.h"

#define RB_HEADS 16
#define RB_KMAX 8        // K (num_radial) and J (basis_emb_size) <= 8
#define RB_NMAX 256

struct RadialHeads {
  const float* Wa[RB_HEADS];
  const float* Wb[RB_HEADS];      // null: single layer
  const float* bias[RB_HEADS];    // single layer only
  float* Y[RB_HEADS];             // forward outputs
  const float* gY[RB_HEADS];      // backward inputs
  int N[RB_HEADS];
  int J[RB_HEADS];                // rows of Wa (= N for a single layer)
  int act[RB_HEADS];              // 0 none, 1 swish (single layer only)
  int poff[RB_HEADS];             // offset of this head's partial gradients inside a block's partial row
  int nheads;
};

__device__ __forceinline__ float rb_sigmoid(float z) { return 1.0f / (1.0f + expf(-z)); }

// forward: grid (row chunks, heads); a wave walks 16-row tiles with the head's weights as MFMA A operands in registers
// (v_mfma_f32_16x16x4_f32, D[channel][row] so that a lane ends with FOUR CONSECUTIVE CHANNELS of one row: 16-byte stores):
//   single layer   D = bias + Wa X^T            A: lane (g, i) = Wa[16 cb + i][g + 4 s], s < 2;  B: X[row i][g + 4 s]
//   two-layer      D1 = Wa X^T  (t^T, [J][row]), then D = Wb t^T with the contraction slot (g, v) LABELLED j = 4 g + v — the
//                  registers D1 leaves in lane (g, i) ARE the B operand; A: Wb[16 cb + i][4 g + v]
// The kernel this replaces (weights in LDS, a thread per (row, 4 outputs), 48 + 8 LDS reads per 16-byte store): 26.6 us at
// 8 704 rows x 10 heads, docs/history/r06_radial_fwd_lds.hip.txt.  N > 128: blockIdx.z walks the 128-channel halves.
typedef float rb_f32x4 __attribute__((ext_vector_type(4)));
#define RB_TILE 16

__global__ void __launch_bounds__(256) k_radial_fwd_mfma(const float* __restrict__ X, int M, int K, RadialHeads d,
                                                          int tiles_per_block) {
  const int h = blockIdx.y;
  const int N = d.N[h], J = d.J[h], act = d.act[h];
  const float* __restrict__ Wa = d.Wa[h];
  const float* __restrict__ Wb = d.Wb[h];
  const float* __restrict__ bias = d.bias[h];
  float* __restrict__ Y = d.Y[h];
  const int cbase = blockIdx.z * 128;
  if (cbase >= N) return;
  const int nc = N - cbase < 128 ? N - cbase : 128;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, i = l & 15, g = l >> 4;
  const int ntiles = (M + RB_TILE - 1) / RB_TILE;
  const int t0 = blockIdx.x * tiles_per_block;
  const int t1 = t0 + tiles_per_block < ntiles ? t0 + tiles_per_block : ntiles;
  // A operands.  single: wa[cb][s] = Wa[ch][g + 4 s];  two-layer: wa[0][s] = Wa[j = i][g + 4 s], wb[cb][v] = Wb[ch][4 g + v]
  float w[8][4], wa0[2];                              // w: single layer [cb][s] (two used), two-layer [cb][v]
  rb_f32x4 b4[8];
#pragma unroll
  for (int cb = 0; cb < 8; ++cb) {
    const int ch = 16 * cb + i, chc = cbase + (ch < nc ? ch : nc - 1);
    const float okc = ch < nc ? 1.0f : 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int k = g + 4 * s;
      if (!Wb) w[cb][s] = Wa[(int64_t)chc * K + (k < K ? k : K - 1)] * (k < K ? okc : 0.f);
      else if (cb == 0) wa0[s] = Wa[(i < J ? i : J - 1) * K + (k < K ? k : K - 1)] * ((k < K && i < J) ? 1.0f : 0.f);
    }
    if (Wb) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int j = 4 * g + v;
        w[cb][v] = Wb[(int64_t)chc * J + (j < J ? j : J - 1)] * (j < J ? okc : 0.f);
      }
    }
    // D rows of lane (g, .): channels 16 cb + 4 g + v
    const int c4 = 16 * cb + 4 * g;
    b4[cb] = rb_f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias && !Wb && c4 < nc) {
      const float4 bv = *(const float4*)(bias + cbase + c4);       // N % 4 == 0
      b4[cb] = rb_f32x4{bv.x, bv.y, bv.z, bv.w};
    }
  }
  float xn[2];
  auto fetch = [&](int t) {
    const int mc = t * RB_TILE + i < M ? t * RB_TILE + i : M - 1;
#pragma unroll
    for (int s = 0; s < 2; ++s) xn[s] = X[(int64_t)mc * K + (g + 4 * s < K ? g + 4 * s : K - 1)];
  };
  if (t0 + wave < t1) fetch(t0 + wave);
  for (int t = t0 + wave; t < t1; t += 4) {
    const int m = t * RB_TILE + i;
    float xb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) xb[s] = xn[s] * (g + 4 * s < K ? 1.0f : 0.f);
    if (t + 4 < t1) fetch(t + 4);
    rb_f32x4 t4 = rb_f32x4{0.f, 0.f, 0.f, 0.f};
    if (Wb) {
      t4 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa0[0], xb[0], t4, 0, 0, 0);
      t4 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa0[1], xb[1], t4, 0, 0, 0);
    }
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {
      if (16 * cb >= nc) break;
      rb_f32x4 acc = b4[cb];
      if (!Wb) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[cb][0], xb[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[cb][1], xb[1], acc, 0, 0, 0);
        if (act == 1) {
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[v] *= rb_sigmoid(acc[v]);
        }
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[cb][v], t4[v], acc, 0, 0, 0);
      }
      const int c4 = 16 * cb + 4 * g;
      if (m < M && c4 < nc) *(float4*)(Y + (int64_t)m * N + cbase + c4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
  }
}

// backward: grid (row chunks, heads); a block's four waves walk the 16-row tiles of its chunk with the head's weight
// gradient PERSISTENT in MFMA accumulators, so a launch leaves ~2 CUs / H partial rows instead of one per 32 rows (272 at
// config 2: their reduction was a third of k_reduce_many).  Every head, single or two-layer, is the same two products on
// the same operand registers (lane (g, i) = (lane / 16, lane % 16) holds gZ[row 4g + v][channel 16 cb + i], v < 4, cb < 8):
//     S  = gZ^T [X | 1]    v_mfma_f32_16x16x4_f32, A = gZ^T (channel i, row label g), B = [X | 1] (row label g, column i):
//                          S[:, :K] the weight gradient, S[:, K] the bias gradient; 32 accumulator registers
//     gX = gZ W_eff        192 FMAs per lane against W_eff[channel][k] in registers + a 16-lane DPP butterfly
// with W_eff = Wa (single) or Wb Wa (two-layer: Y = X (Wb Wa)^T), and at the end of the block, for a two-layer head,
//     gWb = S Wa^T,  gWa = Wb^T S      (since t = X Wa^T: gWb = gY^T t, gWa = (gY Wb)^T X).
// (The contraction index of an MFMA is a label: A and B only have to agree on which row sits in slice (g, v), so the tile
// is loaded ONCE, in the layout a coalesced dword read gives.)  gX of head h goes to slice h of gx_work, summed in a fixed
// order by k_radial_gx_sum.  N > 128: the chunk is walked once per 128 channels.
// The kernel this replaces (one block per 32-row tile, VALU, 52 us at 8 704 rows): docs/history/r06_radial_bwd_tile32.hip.txt.
#define RB_SK 9            // LDS pitch of an S / Wb row: K <= 8 columns + the bias column, odd

__device__ __forceinline__ float rb_dpp_quad_swap(float v) {      // quad_perm [1,0,3,2]
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ float rb_dpp_quad_rev2(float v) {      // quad_perm [2,3,0,1]
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
}
__device__ __forceinline__ float rb_dpp_half_mirror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));
}
__device__ __forceinline__ float rb_dpp_mirror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));
}
// sum over the 16 lanes of a DPP row, in every lane of the row
__device__ __forceinline__ float rb_row_sum(float v) {
  v += rb_dpp_quad_swap(v);
  v += rb_dpp_quad_rev2(v);
  v += rb_dpp_half_mirror(v);
  v += rb_dpp_mirror(v);
  return v;
}

template <int KP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KP <= 6 ? 2 : 1, KP <= 6 ? 2 : 1))) k_radial_bwd_mfma(const float* __restrict__ X, int M, int K, RadialHeads d,
                                                          float* __restrict__ gXw, float* __restrict__ part, int pstride,
                                                          int tiles_per_block) {
  __shared__ float sS[4 * 128 * RB_SK];
  __shared__ float sA[4 * 64];
  __shared__ float sWa[RB_KMAX * RB_KMAX];
  __shared__ float sWb[128 * RB_SK];
  const int h = blockIdx.y;
  const int N = d.N[h], J = d.J[h], act = d.act[h];
  const float* __restrict__ Wa = d.Wa[h];
  const float* __restrict__ Wb = d.Wb[h];
  const float* __restrict__ gY = d.gY[h];
  const float* __restrict__ bias = d.bias[h];
  float* __restrict__ ph = part + (int64_t)blockIdx.x * pstride + d.poff[h];
  float* __restrict__ gXh = gXw ? gXw + (int64_t)h * M * K : nullptr;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, i = l & 15, g = l >> 4;
  const int ntiles = (M + RB_TILE - 1) / RB_TILE;
  const int t0 = blockIdx.x * tiles_per_block;
  const int t1 = t0 + tiles_per_block < ntiles ? t0 + tiles_per_block : ntiles;
  float gwa = 0.f;                                   // two-layer: this thread's share of gWa[j][k], summed over the halves
  if (Wb)
    for (int q = threadIdx.x; q < J * K; q += 256) sWa[q] = Wa[q];
  for (int cbase = 0; cbase < N; cbase += 128) {
    const int nc = N - cbase < 128 ? N - cbase : 128;    // channels of this half
    if (Wb) {
      __syncthreads();                               // the previous half's epilogue reads are done
      for (int q = threadIdx.x; q < nc * J; q += 256) {
        const int n = q / J, j = q - n * J;
        sWb[n * RB_SK + j] = Wb[(int64_t)cbase * J + q];
      }
      __syncthreads();
    }
    // W_eff[channel 16 cb + i][k] and the bias of the activation head: every load issued before the first use
    float we[8][KP];
    int chc[8];
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) chc[cb] = cbase + (16 * cb + i < nc ? 16 * cb + i : nc - 1);
#define RB_OKC(cb) (16 * (cb) + i < nc ? 1.0f : 0.f)
    if (!Wb) {
#pragma unroll
      for (int cb = 0; cb < 8; ++cb)
#pragma unroll
        for (int k = 0; k < KP; ++k) we[cb][k] = Wa[(int64_t)chc[cb] * K + (k < K ? k : K - 1)];
#pragma unroll
      for (int cb = 0; cb < 8; ++cb)
#pragma unroll
        for (int k = 0; k < KP; ++k) we[cb][k] = k < K ? we[cb][k] * RB_OKC(cb) : 0.f;
    } else {
#pragma unroll
      for (int cb = 0; cb < 8; ++cb)
#pragma unroll
        for (int k = 0; k < KP; ++k) we[cb][k] = 0.f;
#pragma unroll
      for (int j = 0; j < RB_KMAX; ++j)
        if (j < J) {
          float wa[KP];
#pragma unroll
          for (int k = 0; k < KP; ++k) wa[k] = k < K ? sWa[j * K + k] : 0.f;
#pragma unroll
          for (int cb = 0; cb < 8; ++cb) {
            const float wb = sWb[(chc[cb] - cbase) * RB_SK + j] * RB_OKC(cb);
#pragma unroll
            for (int k = 0; k < KP; ++k) we[cb][k] = fmaf(wb, wa[k], we[cb][k]);
          }
        }
    }
    rb_f32x4 acc[8];
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) acc[cb] = rb_f32x4{0.f, 0.f, 0.f, 0.f};
    // the next tile's rows are in flight while this one is consumed (two register sets); buffer loads: rows past the
    // end of the array (the over-read of a partial last half) return 0, a NULL gradient is a zero-sized buffer
    const __amdgpu_buffer_rsrc_t gsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)gY, 0, gY ? (unsigned)((int64_t)M * N * 4) : 0u, 0x00020000);
    float gn[4][8], xn[4];
    auto fetch = [&](int t) {
      const int r0 = t * RB_TILE + 4 * g;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int m = r0 + v < M ? r0 + v : M - 1;
#pragma unroll
        for (int cb = 0; cb < 8; ++cb)            // one offset register per row; out-of-range channels are masked on use
          gn[v][cb] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(gsrc, (m * N + cbase + i) * 4 + 64 * cb, 0, 0));
        xn[v] = X[(int64_t)m * K + (i < K ? i : 0)];
      }
    };
    if (t0 + wave < t1) fetch(t0 + wave);
    for (int t = t0 + wave; t < t1; t += 4) {
      const int r0 = t * RB_TILE + 4 * g;
      float gz[4][8], xb[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float okr = r0 + v < M ? 1.0f : 0.f;
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) gz[v][cb] = gn[v][cb] * (okr * RB_OKC(cb));
        xb[v] = i < K ? xn[v] * okr : (i == K ? okr : 0.f);
      }
      if (t + 4 < t1) fetch(t + 4);
      if (act == 1) {
        // gZ = gY act'(z), z recomputed from the K inputs (wave-uniform branch: one head of a SphereNet bundle)
        float bz[8];
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) bz[cb] = bias ? bias[chc[cb]] : 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int m = r0 + v < M ? r0 + v : M - 1;
          float xr[KP];
#pragma unroll
          for (int k = 0; k < KP; ++k) xr[k] = X[(int64_t)m * K + (k < K ? k : 0)];
#pragma unroll
          for (int cb = 0; cb < 8; ++cb) {
            float z = bz[cb];
#pragma unroll
            for (int k = 0; k < KP; ++k) z = fmaf(xr[k], we[cb][k], z);     // we[.][k >= K] = 0
            const float s = rb_sigmoid(z);
            gz[v][cb] *= s * (1.0f + z * (1.0f - s));
          }
        }
      }
#pragma unroll
      for (int cb = 0; cb < 8; ++cb)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(gz[v][cb], xb[v], acc[cb], 0, 0, 0);
      if (gXh) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          float out = 0.f;
#pragma unroll
          for (int k = 0; k < KP; ++k) {
            float p = 0.f;
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) p = fmaf(gz[v][cb], we[cb][k], p);
            p = rb_row_sum(p);
            if (i == k) out = p;
          }
          const int m = r0 + v;
          if (m < M && i < K) {
            float* q = gXh + (int64_t)m * K + i;
            *q = cbase ? *q + out : out;
          }
        }
      }
    }
    // S of the block: the four waves' accumulators through LDS.  acc[cb][v] = S[16 cb + 4 g + v][i]
    if (i <= K) {
#pragma unroll
      for (int cb = 0; cb < 8; ++cb)
#pragma unroll
        for (int v = 0; v < 4; ++v) sS[(wave * 128 + 16 * cb + 4 * g + v) * RB_SK + i] = acc[cb][v];
    }
    __syncthreads();
    for (int q = threadIdx.x; q < 128 * (K + 1); q += 256) {
      const int n = q / (K + 1), k = q - n * (K + 1);
      const int o = n * RB_SK + k;
      sS[o] = ((sS[o] + sS[128 * RB_SK + o]) + sS[2 * 128 * RB_SK + o]) + sS[3 * 128 * RB_SK + o];
    }
    __syncthreads();
    if (!Wb) {
      for (int q = threadIdx.x; q < nc * K; q += 256) {
        const int n = q / K, k = q - n * K;
        ph[(int64_t)(cbase + n) * K + k] = sS[n * RB_SK + k];
      }
      for (int n = threadIdx.x; n < nc; n += 256) ph[(int64_t)N * K + cbase + n] = sS[n * RB_SK + K];
      __syncthreads();                               // sS is rewritten by the next half
    } else {
      // gWb[n][j] = sum_k S[n][k] Wa[j][k]
      for (int q = threadIdx.x; q < nc * J; q += 256) {
        const int n = q / J, j = q - n * J;
        float vsum = 0.f;
        for (int k = 0; k < K; ++k) vsum = fmaf(sS[n * RB_SK + k], sWa[j * K + k], vsum);
        ph[J * K + (int64_t)(cbase + n) * J + j] = vsum;
      }
      // gWa[j][k] = sum_n Wb[n][j] S[n][k]: output o = (j, k) by four threads over a quarter of the channels each
      const int o = threadIdx.x & 63, qr = threadIdx.x >> 6;
      float vsum = 0.f;
      if (o < J * K) {
        const int j = o / K, k = o - j * K;
        for (int n = qr * 32; n < qr * 32 + 32 && n < nc; ++n) vsum = fmaf(sWb[n * RB_SK + j], sS[n * RB_SK + k], vsum);
      }
      sA[threadIdx.x] = vsum;
      __syncthreads();
      if (threadIdx.x < 64) gwa += (sA[threadIdx.x] + sA[64 + threadIdx.x]) + (sA[128 + threadIdx.x] + sA[192 + threadIdx.x]);
    }
  }
  if (Wb && threadIdx.x < J * K) ph[threadIdx.x] = gwa;
}

__global__ void __launch_bounds__(256) k_radial_gx_sum(const float* __restrict__ work, int G, int64_t n,
                                                        float* __restrict__ out) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= n) return;
  float s = work[q];
  for (int g = 1; g < G; ++g) s += work[(int64_t)g * n + q];     // fixed order: deterministic
  out[q] = s;
}

extern "C" {

static int fill_heads(RadialHeads& d, int H, const void* const* Wa, const void* const* Wb, const void* const* bias,
                      const int* N, const int* J, const int* act, int K) {
  if (H < 1 || H > RB_HEADS || K < 1 || K > RB_KMAX || !Wa || !N || !J || !act) return 0;
  int off = 0;
  for (int h = 0; h < H; ++h) {
    d.Wa[h] = (const float*)Wa[h];
    d.Wb[h] = Wb ? (const float*)Wb[h] : nullptr;
    d.bias[h] = bias ? (const float*)bias[h] : nullptr;
    d.N[h] = N[h];
    d.J[h] = d.Wb[h] ? J[h] : N[h];
    d.act[h] = act[h];
    d.Y[h] = nullptr;
    d.gY[h] = nullptr;
    if (!d.Wa[h] || N[h] < 8 || N[h] > RB_NMAX || (N[h] & 3)) return 0;
    if (d.Wb[h] && (J[h] < 1 || J[h] > RB_KMAX || act[h] != 0 || d.bias[h])) return 0;
    if (act[h] != 0 && act[h] != 1) return 0;
    d.poff[h] = off;
    off += d.Wb[h] ? (J[h] * K + N[h] * J[h]) : (N[h] * K + N[h]);
  }
  d.nheads = H;
  return off;
}

// floats of one block's partial row for this head configuration (= total number of weight-gradient entries)
int dig3d_radial_partial_stride(int H, const int* N, const int* J, const int* two_layer, int K) {
  int off = 0;
  for (int h = 0; h < H; ++h) off += two_layer[h] ? (J[h] * K + N[h] * J[h]) : (N[h] * K + N[h]);
  return off;
}
// 16-row tiles per block of the backward launch: ~2 blocks per CU over all heads, whole tiles per wave (a multiple of 4)
static int radial_tiles_per_block(int M, int H) {
  const int ntiles = (M + RB_TILE - 1) / RB_TILE;
  int c0 = 2 * dig3d_num_cus() / (H < 1 ? 1 : H);
  if (c0 < 1) c0 = 1;
  int tpb = (ntiles + c0 - 1) / c0;
  tpb = (tpb + 3) & ~3;
  return tpb < 4 ? 4 : tpb;
}
// partial rows the backward launch writes (= its row chunks)
int dig3d_radial_blocks(int M, int H) {
  if (M <= 0) return 1;
  const int ntiles = (M + RB_TILE - 1) / RB_TILE, tpb = radial_tiles_per_block(M, H);
  return (ntiles + tpb - 1) / tpb;
}

// Y_h = head_h(X) for H <= 16 heads over the same X [M, K <= 8].  Host arrays of H entries; Wb[h] NULL = single layer
// (Wa [N,K], optional bias, act 0/1 = none/swish), else two-layer (Wa [J,K], Wb [N,J]).
int dig3d_radial_fwd(const float* X, int M, int K, int H, const void* const* Wa, const void* const* Wb,
                     const void* const* bias, const int* N, const int* J, const int* act, void* const* Y,
                     void* stream) {
  DIG3D_ENTER();
  RadialHeads d;
  if (M < 0 || !X || !Y || !fill_heads(d, H, Wa, Wb, bias, N, J, act, K)) return DIG3D_ERR_ARG;
  if (M == 0) return DIG3D_OK;
  int maxn = 8;
  for (int h = 0; h < H; ++h) {
    d.Y[h] = (float*)Y[h];
    if (!d.Y[h] || ((uintptr_t)d.Y[h] & 15) || ((uintptr_t)d.bias[h] & 15)) return DIG3D_ERR_ARG;
    if (N[h] > maxn) maxn = N[h];
  }
  const int tpb = radial_tiles_per_block(M, H);
  const int chunks = ((M + RB_TILE - 1) / RB_TILE + tpb - 1) / tpb;
  hipLaunchKernelGGL(k_radial_fwd_mfma, dim3(chunks, H, (maxn + 127) / 128), dim3(256), 0, (hipStream_t)stream, X, M, K, d, tpb);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// gX [M,K] (may be NULL) and part[dig3d_radial_blocks(M, H)][stride]: per-chunk partials of every head's weight gradients
// at poff_h (single: [N*K gWa | N gb]; two-layer: [J*K gWa | N*J gWb]); gY[h] may be NULL (head unused: zero gradient).
// gx_work: float[dig3d_radial_bwd_groups(H) * M * K] — one slice per head, summed into gX by k_radial_gx_sum (not needed
// for H == 1 or gX == NULL).
int dig3d_radial_bwd_groups(int H) { return H < 1 ? 1 : H; }

int dig3d_radial_bwd(const float* X, int M, int K, int H, const void* const* Wa, const void* const* Wb,
                     const void* const* bias, const int* N, const int* J, const int* act, const void* const* gY,
                     float* gX, float* part, float* gx_work, void* stream) {
  DIG3D_ENTER();
  RadialHeads d;
  int stride;
  if (M < 0 || !X || !gY || !part || !(stride = fill_heads(d, H, Wa, Wb, bias, N, J, act, K))) return DIG3D_ERR_ARG;
  if (gX && H > 1 && !gx_work) return DIG3D_ERR_ARG;
  if (M == 0) return DIG3D_OK;
  for (int h = 0; h < H; ++h) {
    d.gY[h] = (const float*)gY[h];
    if ((int64_t)M * N[h] * 4 >= (1LL << 32) - 4096) return DIG3D_ERR_ARG;      // 32-bit buffer offsets
  }
  float* gxw = !gX ? nullptr : (H > 1 ? gx_work : gX);
  const int tpb = radial_tiles_per_block(M, H);
  const dim3 grid(dig3d_radial_blocks(M, H), H);
  if (K <= 6)
    hipLaunchKernelGGL(k_radial_bwd_mfma<6>, grid, dim3(256), 0, (hipStream_t)stream, X, M, K, d, gxw, part, stride, tpb);
  else
    hipLaunchKernelGGL(k_radial_bwd_mfma<8>, grid, dim3(256), 0, (hipStream_t)stream, X, M, K, d, gxw, part, stride, tpb);
  DIG3D_CHECK_LAUNCH();
  if (gX && H > 1) {
    const int64_t n = (int64_t)M * K;
    hipLaunchKernelGGL(k_radial_gx_sum, dim3(dig3d_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, gx_work, H, n, gX);
    DIG3D_CHECK_LAUNCH();
  }
  return DIG3D_OK;
}

}  // extern "C"
