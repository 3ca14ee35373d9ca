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
#define RB_ROWS 32

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

// forward: thread per (row, 4 outputs) of one head, heads along blockIdx.y
__global__ void __launch_bounds__(256) k_radial_fwd(const float* __restrict__ X, int M, int K, RadialHeads d) {
  __shared__ float sWa[RB_KMAX * RB_NMAX];      // k-major Wa: [K][J]
  __shared__ float sWb[RB_KMAX * RB_NMAX];      // j-major Wb: [J][N]
  const int h = blockIdx.y;
  const int N = d.N[h], J = d.J[h];
  const float* __restrict__ Wa = d.Wa[h];
  const float* __restrict__ Wb = d.Wb[h];
  for (int q = threadIdx.x; q < J * K; q += 256) {
    const int j = q / K, k = q - j * K;
    sWa[k * J + j] = Wa[q];
  }
  if (Wb)
    for (int q = threadIdx.x; q < N * J; q += 256) {
      const int n = q / J, j = q - n * J;
      sWb[j * N + n] = Wb[q];
    }
  __syncthreads();
  const float* __restrict__ bias = d.bias[h];
  const int act = d.act[h];
  float* __restrict__ Y = d.Y[h];
  const int n4 = N >> 2;
  const int64_t total = (int64_t)M * n4;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
    const int m = (int)(q / n4), n = (int)(q - (int64_t)m * n4) * 4;
    float x[RB_KMAX];
#pragma unroll
    for (int k = 0; k < RB_KMAX; ++k) x[k] = k < K ? X[(int64_t)m * K + k] : 0.f;
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!Wb) {
      if (bias) z = *(const float4*)(bias + n);
#pragma unroll
      for (int k = 0; k < RB_KMAX; ++k)
        if (k < K) {
          const float4 w = *(const float4*)(sWa + k * N + n);
          z.x = fmaf(x[k], w.x, z.x); z.y = fmaf(x[k], w.y, z.y); z.z = fmaf(x[k], w.z, z.z); z.w = fmaf(x[k], w.w, z.w);
        }
      if (act == 1) {
        z.x *= rb_sigmoid(z.x); z.y *= rb_sigmoid(z.y); z.z *= rb_sigmoid(z.z); z.w *= rb_sigmoid(z.w);
      }
    } else {
#pragma unroll
      for (int j = 0; j < RB_KMAX; ++j)
        if (j < J) {
          float t = 0.f;
#pragma unroll
          for (int k = 0; k < RB_KMAX; ++k)
            if (k < K) t = fmaf(x[k], sWa[k * J + j], t);
          const float4 w = *(const float4*)(sWb + j * N + n);
          z.x = fmaf(t, w.x, z.x); z.y = fmaf(t, w.y, z.y); z.z = fmaf(t, w.z, z.z); z.w = fmaf(t, w.w, z.w);
        }
    }
    *(float4*)(Y + (int64_t)m * N + n) = z;
  }
}

// backward: one block per 32-row tile, every head in turn.
//   gX[m,k] (summed over heads, registers)        part[block][poff_h + ...] = this tile's weight-gradient contribution:
//   single:    [N*K gWa | N gb]                    two-layer: [J*K gWa | N*J gWb]
__global__ void __launch_bounds__(256) k_radial_bwd(const float* __restrict__ X, int M, int K, RadialHeads d,
                                                     float* __restrict__ gX, float* __restrict__ part, int pstride) {
  __shared__ float sG[RB_ROWS * (RB_NMAX + 4)];
  __shared__ float sX[RB_ROWS * RB_KMAX];
  __shared__ float sT[RB_ROWS * RB_KMAX];       // two-layer: t = X Wa^T, then gT
  __shared__ float sWa[RB_KMAX * RB_NMAX];      // k-major [K][J]
  __shared__ float sWb[RB_KMAX * RB_NMAX];      // j-major [J][N]
  const int m0 = blockIdx.x * RB_ROWS;
  for (int q = threadIdx.x; q < RB_ROWS * RB_KMAX; q += 256) {
    const int r = q / RB_KMAX, k = q - r * RB_KMAX;
    const int m = m0 + r;
    sX[q] = (m < M && k < K) ? X[(int64_t)m * K + k] : 0.f;
  }
  const int r8 = threadIdx.x >> 3, l8 = threadIdx.x & 7;     // 8 threads per row for the row reductions
  float accx[RB_KMAX];
#pragma unroll
  for (int k = 0; k < RB_KMAX; ++k) accx[k] = 0.f;
  float* __restrict__ prow = part + (int64_t)blockIdx.x * pstride;
  // gridDim.y head groups (heads y, y + G, ...): one tile's heads are independent except for the sum gX, which every
  // group writes to its own slice (summed by k_radial_gx_sum) — with one block walking all 2 + 2L heads in turn the
  // launch was a 108-us dependent chain of ~10 us per head on a quarter-occupied chip.  (r05, measured and not kept: all
  // global loads of a head — weights and the gradient tile — issued before the barrier that frees the LDS, and 1 or 3 heads
  // per block instead of 2: config 2 1.5150 / 1.5138 / 1.5170 ms, config 4 5.392 / 5.407 / 5.399 — inside the box-to-box
  // spread, so the simpler loop stays.)
  if (gridDim.y > 1) gX += (int64_t)blockIdx.y * M * K;
  for (int h = blockIdx.y; h < d.nheads; h += gridDim.y) {
    const int N = d.N[h], J = d.J[h], act = d.act[h];
    const float* __restrict__ Wa = d.Wa[h];
    const float* __restrict__ Wb = d.Wb[h];
    const float* __restrict__ gY = d.gY[h];
    const float* __restrict__ bias = d.bias[h];
    float* __restrict__ ph = prow + d.poff[h];
    const int NP = N + 4;
    __syncthreads();                               // previous head's LDS reads are done (and sX is complete)
    for (int q = threadIdx.x; q < J * K; q += 256) {
      const int j = q / K, k = q - j * K;
      sWa[k * J + j] = Wa[q];
    }
    if (Wb)
      for (int q = threadIdx.x; q < N * J; q += 256) {
        const int n = q / J, j = q - n * J;
        sWb[j * N + n] = Wb[q];
      }
    __syncthreads();
    // stage gZ = gY (* act'(z), z recomputed from the K inputs)
    const int n4 = N >> 2;
    for (int q = threadIdx.x; q < RB_ROWS * n4; q += 256) {
      const int r = q / n4, c = (q - r * n4) * 4;
      const int m = m0 + r;
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M && gY) {
        g = *(const float4*)(gY + (int64_t)m * N + c);
        if (act == 1) {
          float4 z = bias ? *(const float4*)(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int k = 0; k < RB_KMAX; ++k)
            if (k < K) {
              const float x = sX[r * RB_KMAX + k];
              const float4 w = *(const float4*)(sWa + k * N + c);
              z.x = fmaf(x, w.x, z.x); z.y = fmaf(x, w.y, z.y); z.z = fmaf(x, w.z, z.z); z.w = fmaf(x, w.w, z.w);
            }
          float s;
          s = rb_sigmoid(z.x); g.x *= s * (1.0f + z.x * (1.0f - s));
          s = rb_sigmoid(z.y); g.y *= s * (1.0f + z.y * (1.0f - s));
          s = rb_sigmoid(z.z); g.z *= s * (1.0f + z.z * (1.0f - s));
          s = rb_sigmoid(z.w); g.w *= s * (1.0f + z.w * (1.0f - s));
        }
      }
      *(float4*)(sG + r * NP + c) = g;
    }
    if (Wb) {                                      // t[r][j] = sum_k x[r][k] Wa[j][k]
      for (int q = threadIdx.x; q < RB_ROWS * RB_KMAX; q += 256) {
        const int r = q / RB_KMAX, j = q - r * RB_KMAX;
        float t = 0.f;
        if (j < J)
#pragma unroll
          for (int k = 0; k < RB_KMAX; ++k)
            if (k < K) t = fmaf(sX[r * RB_KMAX + k], sWa[k * J + j], t);
        sT[q] = t;
      }
    }
    __syncthreads();
    if (!Wb) {
      // gX += gZ Wa          (8 threads per row over N/8 columns each)
      for (int n = l8; n < N; n += 8) {
        const float g = sG[r8 * NP + n];
#pragma unroll
        for (int k = 0; k < RB_KMAX; ++k)
          if (k < K) accx[k] = fmaf(g, sWa[k * N + n], accx[k]);
      }
      // gWa[n][k], gb[n]     (thread n over the tile rows)
      for (int n = threadIdx.x; n < N; n += 256) {
        float gw[RB_KMAX], gb = 0.f;
#pragma unroll
        for (int k = 0; k < RB_KMAX; ++k) gw[k] = 0.f;
        for (int r = 0; r < RB_ROWS; ++r) {
          const float g = sG[r * NP + n];
          gb += g;
#pragma unroll
          for (int k = 0; k < RB_KMAX; ++k) gw[k] = fmaf(g, sX[r * RB_KMAX + k], gw[k]);
        }
#pragma unroll
        for (int k = 0; k < RB_KMAX; ++k)
          if (k < K) ph[n * K + k] = gw[k];
        ph[N * K + n] = gb;
      }
    } else {
      // gWb[n][j] = sum_r gY[r][n] t[r][j]
      for (int n = threadIdx.x; n < N; n += 256) {
        float gw[RB_KMAX];
#pragma unroll
        for (int j = 0; j < RB_KMAX; ++j) gw[j] = 0.f;
        for (int r = 0; r < RB_ROWS; ++r) {
          const float g = sG[r * NP + n];
#pragma unroll
          for (int j = 0; j < RB_KMAX; ++j) gw[j] = fmaf(g, sT[r * RB_KMAX + j], gw[j]);
        }
#pragma unroll
        for (int j = 0; j < RB_KMAX; ++j)
          if (j < J) ph[J * K + n * J + j] = gw[j];
      }
      // gT[r][j] = sum_n gY[r][n] Wb[n][j]   (8 threads per row, xor-shuffle sum), then into sT
      float gt[RB_KMAX];
#pragma unroll
      for (int j = 0; j < RB_KMAX; ++j) gt[j] = 0.f;
      for (int n = l8; n < N; n += 8) {
        const float g = sG[r8 * NP + n];
#pragma unroll
        for (int j = 0; j < RB_KMAX; ++j)
          if (j < J) gt[j] = fmaf(g, sWb[j * N + n], gt[j]);
      }
#pragma unroll
      for (int j = 0; j < RB_KMAX; ++j) {
        float v = gt[j];
        v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
        gt[j] = v;
      }
      __syncthreads();                             // every read of sT (= t) is done
      if (l8 == 0) {
#pragma unroll
        for (int j = 0; j < RB_KMAX; ++j) sT[r8 * RB_KMAX + j] = gt[j];
      }
      __syncthreads();
      // gX += gT Wa  — thread (row, l8 = k) adds its own k directly (no further reduction needed for this part):
      // fold it into accx through lane l8 == 0's slot after the final shuffle; simpler: keep a second accumulator
      if (l8 == 0) {
#pragma unroll
        for (int k = 0; k < RB_KMAX; ++k)
          if (k < K) {
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < RB_KMAX; ++j)
              if (j < J) v = fmaf(sT[r8 * RB_KMAX + j], sWa[k * J + j], v);
            accx[k] += v;
          }
      }
      // gWa[j][k] = sum_r gT[r][j] x[r][k]
      if (threadIdx.x < J * K) {
        const int j = threadIdx.x / K, k = threadIdx.x - j * K;
        float v = 0.f;
        for (int r = 0; r < RB_ROWS; ++r) v = fmaf(sT[r * RB_KMAX + j], sX[r * RB_KMAX + k], v);
        ph[j * K + k] = v;
      }
    }
  }
  // reduce the 8 partial row sums and write gX
#pragma unroll
  for (int k = 0; k < RB_KMAX; ++k) {
    float v = accx[k];
    v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    accx[k] = v;
  }
  const int m = m0 + r8;
  if (gX && m < M && l8 < K) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < RB_KMAX; ++k)
      if (k == l8) v = accx[k];
    gX[(int64_t)m * K + l8] = v;
  }
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
int dig3d_radial_blocks(int M) { return M <= 0 ? 1 : (M + RB_ROWS - 1) / RB_ROWS; }

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
  int bx = dig3d_blocks((int64_t)M * (maxn / 4), 256);
  if (bx > 256) bx = 256;                     // grid-stride: the weight staging is amortised
  hipLaunchKernelGGL(k_radial_fwd, dim3(bx, H), dim3(256), 0, (hipStream_t)stream, X, M, K, d);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// gX [M,K] (may be NULL) and part[dig3d_radial_blocks(M)][stride]: per-tile partials of every head's weight gradients
// at poff_h (single: [N*K gWa | N gb]; two-layer: [J*K gWa | N*J gWb]); gY[h] may be NULL (head unused: zero gradient).
// head groups of dig3d_radial_bwd (= slices of its gx_work buffer)
int dig3d_radial_bwd_groups(int H) {
  int g = (H + 1) / 2;                      // two heads per block
  if (g > 8) g = 8;
  return g < 1 ? 1 : g;
}

int dig3d_radial_bwd(const float* X, int M, int K, int H, const void* const* Wa, const void* const* Wb,
                     const void* const* bias, const int* N, const int* J, const int* act, const void* const* gY,
                     float* gX, float* part, float* gx_work, void* stream) {
  DIG3D_ENTER();
  RadialHeads d;
  int stride;
  if (M < 0 || !X || !gY || !part || !(stride = fill_heads(d, H, Wa, Wb, bias, N, J, act, K))) return DIG3D_ERR_ARG;
  if (M == 0) return DIG3D_OK;
  for (int h = 0; h < H; ++h) {
    d.gY[h] = (const float*)gY[h];
    if ((uintptr_t)d.gY[h] & 15) return DIG3D_ERR_ARG;
  }
  const int G = (gx_work && gX) ? dig3d_radial_bwd_groups(H) : 1;
  hipLaunchKernelGGL(k_radial_bwd, dim3(dig3d_radial_blocks(M), G), dim3(256), 0, (hipStream_t)stream, X, M, K, d,
                     G > 1 ? gx_work : gX, part, stride);
  DIG3D_CHECK_LAUNCH();
  if (G > 1) {
    const int64_t n = (int64_t)M * K;
    hipLaunchKernelGGL(k_radial_gx_sum, dim3(dig3d_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, gx_work, G, n, gX);
    DIG3D_CHECK_LAUNCH();
  }
  return DIG3D_OK;
}

}  // extern "C"
