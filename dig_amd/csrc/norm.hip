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

// out[c] = sum_b part[b, c], ascending b
__global__ void k_colsum(const float* __restrict__ part, int nb, int n, float* __restrict__ out) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  float v = 0.f;
  for (int b = 0; b < nb; ++b) v += part[(int64_t)b * n + c];
  out[c] = v;
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

extern "C" {

// y = GraphNorm(x) over the graphs ptr[B+1] (nodes of a graph contiguous); mean / rstd [B, C] are kept for the backward.
int dig3d_graphnorm_fwd(const float* x, const int* ptr, int B, int C, const float* weight, const float* bias,
                        const float* mean_scale, float eps, float* y, float* mean, float* rstd, void* stream) {
  DIG3D_ENTER();
  if (B < 0 || C <= 0 || !x || !ptr || !y || !mean || !rstd) return DIG3D_ERR_ARG;
  if (B == 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_graphnorm_fwd, dim3(B), dim3(GN_TPB), 0, (hipStream_t)stream, x, ptr, B, C, weight, bias,
                     mean_scale, eps, y, mean, rstd);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// gx [N, C]; part: float[B * 3 * C] scratch; gparams[3 * C] = (g weight, g bias, g mean_scale).
int dig3d_graphnorm_bwd(const float* gy, const float* x, const int* ptr, int B, int C, const float* weight,
                        const float* mean_scale, const float* mean, const float* rstd, float* gx, float* part,
                        float* gparams, void* stream) {
  DIG3D_ENTER();
  hipStream_t st = (hipStream_t)stream;
  if (B < 0 || C <= 0 || !gy || !x || !ptr || !gx || !part || !gparams) return DIG3D_ERR_ARG;
  if (B == 0) {
    if (hipMemsetAsync(gparams, 0, sizeof(float) * 3 * C, st) != hipSuccess) return DIG3D_ERR_LAUNCH;
    return DIG3D_OK;
  }
  hipLaunchKernelGGL(k_graphnorm_bwd, dim3(B), dim3(GN_TPB), 0, st, gy, x, ptr, B, C, weight, mean_scale, mean, rstd, gx,
                     part);
  hipLaunchKernelGGL(k_colsum, dim3(dig3d_blocks(3 * C, 256)), dim3(256), 0, st, part, B, 3 * C, gparams);
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
