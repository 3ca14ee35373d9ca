// dig3d — GraphNorm (torch_geometric.nn.GraphNorm as used by method/comenet/comenet.py:160,213) as a fused
// per-graph kernel pair, and the mean flavour of the segment reductions (scatter(..., reduce='mean'):
// dig/ggraph3D/method/G_SphereNet/model/spherenet.py:171-172,205,297).
//
//   mean_g = sum_{n in g} x_n / |g|;  o = x - mean_g * mean_scale;  var_g = sum o^2 / |g|;
//   y = weight * o / sqrt(var_g + eps) + bias
// The reference composes this from two scatter_mean calls, two gathers and ~8 elementwise kernels (forward) and
// their autograd (backward).  Nodes of a graph are contiguous (batch vector sorted), so one workgroup owns one graph:
// three passes over its rows (they stay in L2: <= 128 atoms x 256 channels x 4 B), statistics in registers / LDS, no
// atomics.  Backward: two passes + per-graph partials of the three parameter gradients, reduced in ascending graph
// order by a second tiny kernel (deterministic).
#include "common.h"

#define GN_TPB 256

// threads are laid out (rg, c): c = tid % CW owns channel c (+ k*CW), rg = tid / CW strides the rows.
struct GnLayout {
  int CW, RG;
};
__device__ __forceinline__ GnLayout gn_layout(int C) {
  GnLayout L;
  if (C <= GN_TPB && (GN_TPB % C) == 0) {
    L.CW = C;
    L.RG = GN_TPB / C;
  } else {
    L.CW = GN_TPB;
    L.RG = 1;
  }
  return L;
}

// block-wide sum over the row groups of per-thread value v for channel slot (tid % CW); result valid in all threads
__device__ __forceinline__ float gn_reduce(float v, float* sh, int CW, int RG) {
  if (RG == 1) return v;
  __syncthreads();
  sh[threadIdx.x] = v;
  __syncthreads();
  float s = 0.f;
  const int c = threadIdx.x % CW;
  for (int r = 0; r < RG; ++r) s += sh[r * CW + c];
  return s;
}

__global__ void __launch_bounds__(GN_TPB) k_graphnorm_fwd(const float* __restrict__ x, const int* __restrict__ ptr,
                                                           int B, int C, const float* __restrict__ weight,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ mean_scale, float eps,
                                                           float* __restrict__ y, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out) {
  __shared__ float sh[GN_TPB];
  const int g = blockIdx.x;
  if (g >= B) return;
  const int r0 = ptr[g], r1 = ptr[g + 1];
  const int n = r1 - r0;
  const float inv_n = 1.0f / (float)(n > 0 ? n : 1);
  const GnLayout L = gn_layout(C);
  const int c0 = threadIdx.x % L.CW, rg = threadIdx.x / L.CW;
  for (int c = c0; c < C; c += L.CW) {
    float s = 0.f;
    for (int r = r0 + rg; r < r1; r += L.RG) s += x[(int64_t)r * C + c];
    s = gn_reduce(s, sh, L.CW, L.RG);
    const float m = s * inv_n;
    const float ms = m * mean_scale[c];
    float q = 0.f;
    for (int r = r0 + rg; r < r1; r += L.RG) {
      const float o = x[(int64_t)r * C + c] - ms;
      q += o * o;
    }
    q = gn_reduce(q, sh, L.CW, L.RG);
    const float rstd = 1.0f / sqrtf(q * inv_n + eps);
    const float w = weight[c], b = bias[c];
    for (int r = r0 + rg; r < r1; r += L.RG) {
      const float o = x[(int64_t)r * C + c] - ms;
      y[(int64_t)r * C + c] = w * o * rstd + b;
    }
    if (rg == 0) {
      mean_out[(int64_t)g * C + c] = m;
      rstd_out[(int64_t)g * C + c] = rstd;
    }
  }
}

// gx, and per-graph partials part[g, 0..2, c] = (gw, gb, g mean_scale)
__global__ void __launch_bounds__(GN_TPB) k_graphnorm_bwd(const float* __restrict__ gy, const float* __restrict__ x,
                                                           const int* __restrict__ ptr, int B, int C,
                                                           const float* __restrict__ weight,
                                                           const float* __restrict__ mean_scale,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, float* __restrict__ gx,
                                                           float* __restrict__ part) {
  __shared__ float sh[GN_TPB];
  const int g = blockIdx.x;
  if (g >= B) return;
  const int r0 = ptr[g], r1 = ptr[g + 1];
  const int n = r1 - r0;
  const float inv_n = 1.0f / (float)(n > 0 ? n : 1);
  const GnLayout L = gn_layout(C);
  const int c0 = threadIdx.x % L.CW, rg = threadIdx.x / L.CW;
  for (int c = c0; c < C; c += L.CW) {
    const float m = mean[(int64_t)g * C + c], r = rstd[(int64_t)g * C + c];
    const float s = mean_scale[c], w = weight[c];
    const float ms = m * s;
    float sg = 0.f, sgo = 0.f;
    for (int q = r0 + rg; q < r1; q += L.RG) {
      const float gv = gy[(int64_t)q * C + c];
      sg += gv;
      sgo += gv * (x[(int64_t)q * C + c] - ms);
    }
    sg = gn_reduce(sg, sh, L.CW, L.RG);
    sgo = gn_reduce(sgo, sh, L.CW, L.RG);
    // go = w r (gy - (r^2/n) o S_go);  sum(go) = w r (S_g - (r^2/n) sum(o) S_go), sum(o) = n m (1 - s)
    const float k = r * r * inv_n * sgo;
    const float sum_o = (float)n * m * (1.0f - s);
    const float sum_go = w * r * (sg - k * sum_o);
    const float shift = s * inv_n * sum_go;
    for (int q = r0 + rg; q < r1; q += L.RG) {
      const float o = x[(int64_t)q * C + c] - ms;
      gx[(int64_t)q * C + c] = w * r * (gy[(int64_t)q * C + c] - k * o) - shift;
    }
    if (rg == 0) {
      float* p = part + (int64_t)g * 3 * C;
      p[c] = r * sgo;            // d/d weight
      p[C + c] = sg;             // d/d bias
      p[2 * C + c] = -m * sum_go;  // d/d mean_scale
    }
  }
}

// ---- C % 4 == 0, C / 4 a power of two <= 256: the kernels that run --------------------------------------------------
// The per-channel layout above walks a graph's rows one scalar at a time in three passes (128 atoms x 256 channels: 60 us
// per launch for 33 MB of traffic, 128 four-wave workgroups on 128 CUs).  Here a workgroup of 1024 threads owns a graph:
// C/4 lanes across a row (one 16-byte load each), 4096/C rows per pass, and every thread KEEPS its first GN4_RC rows in
// registers — a graph of up to GN4_RC * 4096 / C rows (128 at C = 256) is read from memory once in the forward pass and
// once (gy and x) in the backward pass; longer graphs re-read the rest from L2.
#define GN4_TPB 1024
#define GN4_RC 8

__device__ __forceinline__ float4 gn4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 gn4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 gn4_mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 gn4_scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 gn4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// sum over the row groups of the workgroup for this thread's channel quad; result in every thread
__device__ __forceinline__ float4 gn4_reduce(float4 v, float4* sh, int CW, int RP) {
  __syncthreads();
  sh[threadIdx.x] = v;
  __syncthreads();
  float4 s = gn4_zero();
  const int c = threadIdx.x % CW;
  for (int r = 0; r < RP; ++r) s = gn4_add(s, sh[r * CW + c]);
  return s;
}

__global__ void __launch_bounds__(GN4_TPB) k_graphnorm_fwd4(const float4* __restrict__ x, const int* __restrict__ ptr,
                                                             int B, int CW, const float4* __restrict__ weight,
                                                             const float4* __restrict__ bias,
                                                             const float4* __restrict__ mean_scale, float eps,
                                                             float4* __restrict__ y, float4* __restrict__ mean_out,
                                                             float4* __restrict__ rstd_out) {
  __shared__ float4 sh[GN4_TPB];
  const int g = blockIdx.x;
  const int r0 = ptr[g], r1 = ptr[g + 1];
  const int n = r1 - r0;
  const float inv_n = 1.0f / (float)(n > 0 ? n : 1);
  const int RP = GN4_TPB / CW, c = threadIdx.x % CW, rg = threadIdx.x / CW;
  const int rx = r0 + rg + GN4_RC * RP;              // first row of this thread that is not kept in registers
  float4 xv[GN4_RC];
  float4 s = gn4_zero();
#pragma unroll
  for (int j = 0; j < GN4_RC; ++j) {
    const int r = r0 + rg + j * RP;
    xv[j] = r < r1 ? x[(int64_t)r * CW + c] : gn4_zero();
    s = gn4_add(s, xv[j]);
  }
  for (int r = rx; r < r1; r += RP) s = gn4_add(s, x[(int64_t)r * CW + c]);
  s = gn4_reduce(s, sh, CW, RP);
  const float4 m = gn4_scale(s, inv_n);
  const float4 ms = gn4_mul(m, mean_scale[c]);
  float4 q = gn4_zero();
#pragma unroll
  for (int j = 0; j < GN4_RC; ++j) {
    const float4 o = gn4_sub(xv[j], ms);
    if (r0 + rg + j * RP < r1) q = gn4_add(q, gn4_mul(o, o));
  }
  for (int r = rx; r < r1; r += RP) {
    const float4 o = gn4_sub(x[(int64_t)r * CW + c], ms);
    q = gn4_add(q, gn4_mul(o, o));
  }
  q = gn4_reduce(q, sh, CW, RP);
  float4 rstd;
  rstd.x = 1.0f / sqrtf(q.x * inv_n + eps);
  rstd.y = 1.0f / sqrtf(q.y * inv_n + eps);
  rstd.z = 1.0f / sqrtf(q.z * inv_n + eps);
  rstd.w = 1.0f / sqrtf(q.w * inv_n + eps);
  const float4 w = weight[c], b = bias[c];
  auto out = [&](float4 xr) {                        // w * o * rstd + b, the operation order of the per-channel kernel
    const float4 o = gn4_sub(xr, ms);
    return gn4_add(gn4_mul(gn4_mul(w, o), rstd), b);
  };
#pragma unroll
  for (int j = 0; j < GN4_RC; ++j) {
    const int r = r0 + rg + j * RP;
    if (r < r1) y[(int64_t)r * CW + c] = out(xv[j]);
  }
  for (int r = rx; r < r1; r += RP) y[(int64_t)r * CW + c] = out(x[(int64_t)r * CW + c]);
  if (rg == 0) {
    mean_out[(int64_t)g * CW + c] = m;
    rstd_out[(int64_t)g * CW + c] = rstd;
  }
}

__global__ void __launch_bounds__(GN4_TPB) k_graphnorm_bwd4(const float4* __restrict__ gy, const float4* __restrict__ x,
                                                             const int* __restrict__ ptr, int B, int CW,
                                                             const float4* __restrict__ weight,
                                                             const float4* __restrict__ mean_scale,
                                                             const float4* __restrict__ mean,
                                                             const float4* __restrict__ rstd, float4* __restrict__ gx,
                                                             float4* __restrict__ part) {
  __shared__ float4 sh[GN4_TPB];
  const int g = blockIdx.x;
  const int r0 = ptr[g], r1 = ptr[g + 1];
  const int n = r1 - r0;
  const float inv_n = 1.0f / (float)(n > 0 ? n : 1);
  const int RP = GN4_TPB / CW, c = threadIdx.x % CW, rg = threadIdx.x / CW;
  const int rx = r0 + rg + GN4_RC * RP;
  const float4 m = mean[(int64_t)g * CW + c], r = rstd[(int64_t)g * CW + c];
  const float4 s = mean_scale[c], w = weight[c];
  const float4 ms = gn4_mul(m, s);
  float4 gv[GN4_RC], ov[GN4_RC];                     // gy and o = x - mean * mean_scale of the rows kept
  float4 sg = gn4_zero(), sgo = gn4_zero();
#pragma unroll
  for (int j = 0; j < GN4_RC; ++j) {
    const int q = r0 + rg + j * RP;
    const bool in = q < r1;
    gv[j] = in ? gy[(int64_t)q * CW + c] : gn4_zero();
    ov[j] = in ? gn4_sub(x[(int64_t)q * CW + c], ms) : gn4_zero();
    sg = gn4_add(sg, gv[j]);
    sgo = gn4_add(sgo, gn4_mul(gv[j], ov[j]));
  }
  for (int q = rx; q < r1; q += RP) {
    const float4 gq = gy[(int64_t)q * CW + c];
    sg = gn4_add(sg, gq);
    sgo = gn4_add(sgo, gn4_mul(gq, gn4_sub(x[(int64_t)q * CW + c], ms)));
  }
  sg = gn4_reduce(sg, sh, CW, RP);
  sgo = gn4_reduce(sgo, sh, CW, RP);
  // go = w r (gy - (r^2/n) o S_go);  sum(go) = w r (S_g - (r^2/n) sum(o) S_go), sum(o) = n m (1 - s)
  float4 k, sum_go, shift, wr;
#define GN4_LANE(f)                                                  \
  k.f = r.f * r.f * inv_n * sgo.f;                                   \
  sum_go.f = w.f * r.f * (sg.f - k.f * ((float)n * m.f * (1.0f - s.f))); \
  shift.f = s.f * inv_n * sum_go.f;                                  \
  wr.f = w.f * r.f;
  GN4_LANE(x) GN4_LANE(y) GN4_LANE(z) GN4_LANE(w)
#undef GN4_LANE
  auto out = [&](float4 gq, float4 o) { return gn4_sub(gn4_mul(wr, gn4_sub(gq, gn4_mul(k, o))), shift); };
#pragma unroll
  for (int j = 0; j < GN4_RC; ++j) {
    const int q = r0 + rg + j * RP;
    if (q < r1) gx[(int64_t)q * CW + c] = out(gv[j], ov[j]);
  }
  for (int q = rx; q < r1; q += RP)
    gx[(int64_t)q * CW + c] = out(gy[(int64_t)q * CW + c], gn4_sub(x[(int64_t)q * CW + c], ms));
  if (rg == 0) {
    float4* p = part + (int64_t)g * 3 * CW;
    p[c] = gn4_mul(r, sgo);                          // d/d weight
    p[CW + c] = sg;                                  // d/d bias
    p[2 * CW + c] = make_float4(-m.x * sum_go.x, -m.y * sum_go.y, -m.z * sum_go.z, -m.w * sum_go.w);   // d/d mean_scale
  }
}

static inline bool gn4_ok(int C, const void* a, const void* b, const void* c, const void* d) {
  const int cw = C >> 2;
  return (C & 3) == 0 && cw <= 256 && (cw & (cw - 1)) == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15) == 0;
}

// out[c] = sum_b part[b, c]: 64 columns x 4 slices of b per workgroup (ascending b inside a slice, slices added in order)
__global__ void __launch_bounds__(256) k_colsum(const float* __restrict__ part, int nb, int n, float* __restrict__ out) {
  __shared__ float sh[256];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
  const int per = (nb + 3) >> 2, b0 = sl * per, b1 = b0 + per < nb ? b0 + per : nb;
  float v = 0.f;
  if (c < n)
    for (int b = b0; b < b1; ++b) v += part[(int64_t)b * n + c];
  sh[threadIdx.x] = v;
  __syncthreads();
  if (sl == 0 && c < n) out[c] = (sh[threadIdx.x] + sh[threadIdx.x + 64]) + (sh[threadIdx.x + 128] + sh[threadIdx.x + 192]);
}

// out[s, :] = g[s, :] / max(kptr[s+1] - kptr[s], 1)   (backward of a segment mean, before the row gather)
__global__ void k_rows_div_count(const float* __restrict__ g, const int* __restrict__ kptr, int S, int C,
                                 float* __restrict__ out) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (int64_t)S * C) return;
  int s = (int)(q / C);
  int n = kptr[s + 1] - kptr[s];
  out[q] = g[q] / (float)(n > 0 ? n : 1);
}

// rows [*start, N) of y[N, C] = 0: the padded node rows of a static-shape batch behind the last graph (the per-graph
// kernels write the rows of their graphs only; dense layers downstream read every row, 0 x garbage must not be NaN)
__global__ void __launch_bounds__(256) k_zero_rows_from(float* __restrict__ y, const int* __restrict__ start, int N, int C) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x + (int64_t)(*start) * C;
  if (i < (int64_t)N * C) y[i] = 0.f;
}

extern "C" {

// y[r, :] = 0 for *start <= r < N (start: device scalar, e.g. ptr + B).  The grid covers all N rows' worth of threads; the
// live rows' threads fall off the end.
int dig3d_zero_rows_from(float* y, const int* start, int N, int C, void* stream) {
  DIG3D_ENTER();
  if (N < 0 || C <= 0 || !y || !start) return DIG3D_ERR_ARG;
  if (N == 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_zero_rows_from, dim3(dig3d_blocks((int64_t)N * C, 256)), dim3(256), 0, (hipStream_t)stream, y, start, N, C);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// y = GraphNorm(x) over the graphs ptr[B+1] (nodes of a graph contiguous); mean / rstd [B, C] are kept for the backward.
int dig3d_graphnorm_fwd(const float* x, const int* ptr, int B, int C, const float* weight, const float* bias,
                        const float* mean_scale, float eps, float* y, float* mean, float* rstd, void* stream) {
  DIG3D_ENTER();
  if (B < 0 || C <= 0 || !x || !ptr || !y || !mean || !rstd) return DIG3D_ERR_ARG;
  if (B == 0) return DIG3D_OK;
  if (gn4_ok(C, x, y, mean, rstd) && gn4_ok(C, weight, bias, mean_scale, nullptr))
    hipLaunchKernelGGL(k_graphnorm_fwd4, dim3(B), dim3(GN4_TPB), 0, (hipStream_t)stream, (const float4*)x, ptr, B, C >> 2,
                       (const float4*)weight, (const float4*)bias, (const float4*)mean_scale, eps, (float4*)y,
                       (float4*)mean, (float4*)rstd);
  else
    hipLaunchKernelGGL(k_graphnorm_fwd, dim3(B), dim3(GN_TPB), 0, (hipStream_t)stream, x, ptr, B, C, weight, bias,
                       mean_scale, eps, y, mean, rstd);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// gx [N, C]; part: float[B * 3 * C] receives the per-graph partials of (g weight, g bias, g mean_scale); gparams[3 * C] their
// sum over the graphs, or NULL when the caller reduces `part` itself (B rows of stride 3 C: dig3d_reduce_many).
int dig3d_graphnorm_bwd(const float* gy, const float* x, const int* ptr, int B, int C, const float* weight,
                        const float* mean_scale, const float* mean, const float* rstd, float* gx, float* part,
                        float* gparams, void* stream) {
  DIG3D_ENTER();
  hipStream_t st = (hipStream_t)stream;
  if (B < 0 || C <= 0 || !gy || !x || !ptr || !gx || !part || (B == 0 && !gparams)) return DIG3D_ERR_ARG;
  if (B == 0) {
    if (dig3d_zero_async(gparams, sizeof(float) * 3 * C, st) != hipSuccess) return DIG3D_ERR_LAUNCH;
    return DIG3D_OK;
  }
  if (gn4_ok(C, gy, x, gx, part) && gn4_ok(C, weight, mean_scale, mean, rstd))
    hipLaunchKernelGGL(k_graphnorm_bwd4, dim3(B), dim3(GN4_TPB), 0, st, (const float4*)gy, (const float4*)x, ptr, B, C >> 2,
                       (const float4*)weight, (const float4*)mean_scale, (const float4*)mean, (const float4*)rstd,
                       (float4*)gx, (float4*)part);
  else
    hipLaunchKernelGGL(k_graphnorm_bwd, dim3(B), dim3(GN_TPB), 0, st, gy, x, ptr, B, C, weight, mean_scale, mean, rstd, gx,
                       part);
  if (gparams) hipLaunchKernelGGL(k_colsum, dim3(dig3d_blocks(3 * C, 64)), dim3(256), 0, st, part, B, 3 * C, gparams);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_rows_div_count(const float* g, const int* kptr, int S, int C, float* out, void* stream) {
  DIG3D_ENTER();
  if (S <= 0) return DIG3D_OK;
  if (C <= 0 || !g || !kptr || !out) return DIG3D_ERR_ARG;
  hipLaunchKernelGGL(k_rows_div_count, dim3(dig3d_blocks((int64_t)S * C, 256)), dim3(256), 0, (hipStream_t)stream, g,
                     kptr, S, C, out);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

}  // extern "C"
