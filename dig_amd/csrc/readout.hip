// dig3d — the output blocks ("update_v" + "update_u", method/spherenet/spherenet.py:185-225, dimenetpp.py:164-204) of all
// interaction layers, batched.  Every block reads only its layer's e2 and adds a per-graph scalar into u, so the
// L + 1 blocks of a forward are independent of each other and of the edge chain that follows them.  At the
// reference's batch size each block is ~7 launches of 5-15 us on N_atoms ~ 600 rows (latency bound, 20 workgroups);
// running the G = L + 1 blocks as ONE grouped launch per stage turns 7 G launches into 7 and fills 5x more CUs:
//     v_g  = scatter(e2_g, i)               k_segsum_grouped       (edges sorted by target: CSR segment sum)
//     ...  = lin_up / lins                  csrc/dense.hip:k_linear_fwd_grouped / k_linear_bwd_both_grouped
//     y_g  = lin(v_g)   (256 -> out <= 8)   k_smalln_fwd_grouped   (row dot products, no GEMM library)
//     u    = ((0 + scatter(y_0, batch)) + scatter(y_1, batch)) + ...   k_graphsum_grouped (the reference's order)
#include "common.h"

#define RG_MAX 8
struct PtrTable {
  const float* in[RG_MAX];
  float* out[RG_MAX];
  const float* aux[RG_MAX];
  const float* aux2[RG_MAX];
  float* out2[RG_MAX];
  // energy_and_force (null otherwise): a second product of the segment sum (in2 * aux2), addends of the gather's two outputs
  const float* in2[RG_MAX];
  const float* add[RG_MAX];
  const float* add2[RG_MAX];
};

__device__ __forceinline__ void f4add(float4& a, const float4 v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }

// out_g[s,:] = sum_{p in [kptr[s], kptr[s+1])} in_g[p,:] (* aux_g[p,:])     (C = 4 * LPR, worker = LPR lanes per segment)
// with aux: e2 = lin_rbf(rbf) * e1 (spherenet.py:90,182) is never materialised — the product is formed while summing.
template <int LPR>
__global__ void __launch_bounds__(256) k_segsum_grouped(PtrTable t, const int* __restrict__ kptr, int S) {
  const int g = blockIdx.y;
  const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPR;
  const int c = threadIdx.x % LPR;
  if (w >= S) return;
  const float4* __restrict__ A = (const float4*)t.in[g];
  const float4* __restrict__ Bm = (const float4*)t.aux[g];
  const float4* __restrict__ A2 = (const float4*)t.in2[g];      // second product (in2 * aux2) added row by row, or null
  const float4* __restrict__ B2 = (const float4*)t.aux2[g];
  const int b = kptr[w], e = kptr[w + 1];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int U = 4;
  for (int p = b; p < e; p += U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p + u < e) {
        v[u] = A[(int64_t)(p + u) * LPR + c];
        if (Bm) {
          const float4 m = Bm[(int64_t)(p + u) * LPR + c];
          v[u].x *= m.x; v[u].y *= m.y; v[u].z *= m.z; v[u].w *= m.w;
        }
        if (A2) {
          const float4 a2 = A2[(int64_t)(p + u) * LPR + c], m2 = B2[(int64_t)(p + u) * LPR + c];
          v[u].x += a2.x * m2.x; v[u].y += a2.y * m2.y; v[u].z += a2.z * m2.z; v[u].w += a2.w * m2.w;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) f4add(acc, v[u]);
  }
  ((float4*)t.out[g])[(int64_t)w * LPR + c] = acc;
}

// out_g[m,:] = in_g[ix[m],:] (* aux_g[m,:]),  out2_g[m,:] = in_g[ix[m],:] * aux2_g[m,:]   (rows m >= *cnt: zeros)
// — backward of k_segsum_grouped (with the product form: the two factor gradients in one pass)
__global__ void k_gather_grouped(PtrTable t, const int* __restrict__ ix, int64_t M, int C4,
                                 const int* __restrict__ cnt) {
  const int g = blockIdx.y;
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= M * C4) return;
  int64_t m = q / C4;
  int c = (int)(q - m * C4);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f), w = v;
  if (!(cnt && m >= *cnt)) {
    v = ((const float4*)t.in[g])[(int64_t)ix[m] * C4 + c];
    w = v;
    if (t.aux[g]) {
      const float4 a = ((const float4*)t.aux[g])[q];
      v.x *= a.x; v.y *= a.y; v.z *= a.z; v.w *= a.w;
    }
    if (t.out2[g]) {
      const float4 a = ((const float4*)t.aux2[g])[q];
      w.x *= a.x; w.y *= a.y; w.z *= a.z; w.w *= a.w;
    }
  }
  if (t.add[g]) {
    const float4 a = ((const float4*)t.add[g])[q];
    v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
  }
  ((float4*)t.out[g])[q] = v;
  if (t.out2[g]) {
    if (t.add2[g]) {
      const float4 a = ((const float4*)t.add2[g])[q];
      w.x += a.x; w.y += a.y; w.z += a.z; w.w += a.w;
    }
    ((float4*)t.out2[g])[q] = w;
  }
}

// y_g[m,n] = sum_k x_g[m,k] W_g[n,k] (+ bias_g[n]),  N <= 8: one wavefront per row, lanes stride k, xor-shuffle sum
__global__ void __launch_bounds__(256) k_smalln_fwd_grouped(PtrTable t, int M, int K, int N) {
  const int g = blockIdx.y;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (m >= M) return;
  const float* __restrict__ x = t.in[g] + (int64_t)m * K;
  const float* __restrict__ W = t.aux[g];
  float acc[RG_MAX];
#pragma unroll
  for (int n = 0; n < RG_MAX; ++n) acc[n] = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float xv = x[k];
#pragma unroll
    for (int n = 0; n < RG_MAX; ++n)
      if (n < N) acc[n] = fmaf(xv, W[(int64_t)n * K + k], acc[n]);
  }
#pragma unroll
  for (int n = 0; n < RG_MAX; ++n) {
    float v = acc[n];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    acc[n] = v;
  }
  if (lane < N) {
    float v = 0.f;
#pragma unroll
    for (int n = 0; n < RG_MAX; ++n)
      if (n == lane) v = acc[n];
    if (t.aux2[g]) v += t.aux2[g][lane];
    t.out[g][(int64_t)m * N + lane] = v;
  }
}

// backward of the small-N layer.  in = gy_g [M,N], aux = W_g [N,K], aux2 = x_g [M,K], out = gx_g [M,K];
// part_g[blockIdx.x][N*K + N] = this block's SN_ROWS-row partial of (gW, gb)  (reduced by dig3d_reduce_many).
struct PartTable {
  float* part[RG_MAX];
};
#define SN_ROWS 8        // rows per block: a thread walks them one after the other (64 rows: 32 us per launch at 600 atoms; 16: 10.4; 8: 8.0)
__global__ void __launch_bounds__(256) k_smalln_bwd_grouped(PtrTable t, PartTable pt, int M, int K, int N) {
  __shared__ float sgy[SN_ROWS * RG_MAX];
  const int g = blockIdx.y;
  const int m0 = blockIdx.x * SN_ROWS;
  const int rows = (M - m0 < SN_ROWS) ? M - m0 : SN_ROWS;
  const float* __restrict__ gy = t.in[g];
  const float* __restrict__ W = t.aux[g];
  const float* __restrict__ x = t.aux2[g];
  float* __restrict__ gx = t.out[g];
  for (int q = threadIdx.x; q < SN_ROWS * N; q += 256) {
    const int r = q / N, n = q - r * N;
    sgy[r * RG_MAX + n] = (r < rows) ? gy[(int64_t)(m0 + r) * N + n] : 0.f;
  }
  __syncthreads();
  float* outp = pt.part[g] ? pt.part[g] + (int64_t)blockIdx.x * ((int64_t)N * K + N) : nullptr;
  for (int k = threadIdx.x; k < K; k += 256) {
    float w[RG_MAX], gw[RG_MAX];
#pragma unroll
    for (int n = 0; n < RG_MAX; ++n) {
      w[n] = (n < N) ? W[(int64_t)n * K + k] : 0.f;
      gw[n] = 0.f;
    }
    for (int r = 0; r < rows; ++r) {
      const float xv = x ? x[(int64_t)(m0 + r) * K + k] : 0.f;
      float s = 0.f;
#pragma unroll
      for (int n = 0; n < RG_MAX; ++n)
        if (n < N) {
          const float gv = sgy[r * RG_MAX + n];
          s = fmaf(gv, w[n], s);
          gw[n] = fmaf(gv, xv, gw[n]);
        }
      if (gx) gx[(int64_t)(m0 + r) * K + k] = s;
    }
    if (outp) {
#pragma unroll
      for (int n = 0; n < RG_MAX; ++n)
        if (n < N) outp[(int64_t)n * K + k] = gw[n];
    }
  }
  if (outp && threadIdx.x < N) {
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += sgy[r * RG_MAX + threadIdx.x];
    outp[(int64_t)N * K + threadIdx.x] = s;
  }
}

// u[b,c] = ((0 + sum_{n in graph b} y_0[n,c]) + sum y_1[n,c]) + ...   one thread per (b, c); rows ascending.
// C == 1 (an energy head): one wave per graph; the rows of a group are loaded 64 at a time, one per lane, and added in
// ROW ORDER by broadcasts — the same sums as the thread-per-output kernel below without its chain of 90 dependent loads
__global__ void __launch_bounds__(64) k_graphsum_grouped_c1(PtrTable t, int G, const int* __restrict__ ptr, int B,
                                                             float* __restrict__ u) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int r0 = ptr[b], r1 = ptr[b + 1];
  float tot = 0.f;
  for (int g = 0; g < G; ++g) {
    const float* __restrict__ y = t.in[g];
    float s = 0.f;
    for (int c0 = r0; c0 < r1; c0 += 64) {
      const int n = r1 - c0 < 64 ? r1 - c0 : 64;
      const float v = lane < n ? y[c0 + lane] : 0.f;
      for (int l = 0; l < n; ++l) s += __shfl(v, l, 64);
    }
    tot = tot + s;
  }
  if (lane == 0) u[b] = tot;
}

__global__ void k_graphsum_grouped(PtrTable t, int G, const int* __restrict__ ptr, int B, int C,
                                   float* __restrict__ u) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= B * C) return;
  const int b = q / C, c = q - b * C;
  const int r0 = ptr[b], r1 = ptr[b + 1];
  float tot = 0.f;
  for (int g = 0; g < G; ++g) {
    const float* __restrict__ y = t.in[g];
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += y[(int64_t)r * C + c];
    tot = tot + s;
  }
  u[q] = tot;
}

// mean |out - y| in ONE block: fixed summation order (thread-strided partial sums, then a 256-leaf tree) -> deterministic
// (seed / g given: also g[i] = sgn[i] * seed[0], the gradient under a backward seed known at forward time — a captured
// training step's device scalar — so the backward pass needs no launch of its own)
__global__ void __launch_bounds__(256) k_l1_loss_fwd(const float* __restrict__ out, const float* __restrict__ y, int n,
                                                      float* __restrict__ loss, float* __restrict__ sgn,
                                                      const float* __restrict__ seed, float* __restrict__ g) {
  __shared__ float red[256];
  const float inv = 1.0f / (float)n;
  const float sd = seed ? seed[0] : 0.f;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float dlt = out[i] - y[i];
    s += fabsf(dlt);
    const float sg = dlt > 0.f ? inv : (dlt < 0.f ? -inv : 0.f);
    sgn[i] = sg;
    if (g) g[i] = sg * sd;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = red[0] * inv;
}

// The energy_and_force loss of run.py:126-131 with torch.nn.L1Loss():  loss = mean |out - y| + p * mean |force - f|,
// force = -gpos (gpos = d sum(out) / d pos), in ONE block and ONE launch — the framework's chain was neg / sub / abs / mean /
// mul / add forward and sign / neg / div / mul ... backward: ~24 launches of ~5 us per step.  Rows of a padded static-shape
// batch behind the live atom count carry zero force and zero target, the mean is over the 3 * (*cntN) live entries.  With the
// backward seed known (a captured step's device scalar) the two gradients are written here as well:
//   g_out[i] = sign(out_i - y_i) / nE * seed,   g_gpos[j] = -p * sign(-gpos_j - f_j) / (3 * cntN) * seed.
// Fixed summation order (thread-strided partial sums, then a 256-leaf tree): deterministic.
__global__ void __launch_bounds__(256) k_ef_l1_loss(const float* __restrict__ out, const float* __restrict__ y, int nE,
                                                     const float* __restrict__ gpos, const float* __restrict__ f, int n3,
                                                     const int* __restrict__ cntN, float p, const float* __restrict__ seed,
                                                     float* __restrict__ loss, float* __restrict__ sgn_e,
                                                     float* __restrict__ sgn_f, float* __restrict__ g_out,
                                                     float* __restrict__ g_gpos) {
  __shared__ float red[256];
  const int live3 = cntN ? 3 * cntN[0] : n3;
  const float invE = 1.0f / (float)nE, invF = 1.0f / (float)(live3 > 0 ? live3 : 1);
  const float sd = seed ? seed[0] : 0.f;
  float se = 0.f, sf = 0.f;
  for (int i = threadIdx.x; i < nE; i += 256) {
    const float dlt = out[i] - y[i];
    se += fabsf(dlt);
    const float sg = dlt > 0.f ? invE : (dlt < 0.f ? -invE : 0.f);
    sgn_e[i] = sg;
    if (g_out) g_out[i] = sg * sd;
  }
  for (int j = threadIdx.x; j < n3; j += 256) {
    const float dlt = j < live3 ? (-gpos[j]) - f[j] : 0.f;
    sf += fabsf(dlt);
    const float sg = dlt > 0.f ? -p * invF : (dlt < 0.f ? p * invF : 0.f);     // d |-g - f| / d g = -sign(-g - f)
    sgn_f[j] = sg;
    if (g_gpos) g_gpos[j] = sg * sd;
  }
  red[threadIdx.x] = se;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  const float le = red[0] * invE;
  __syncthreads();
  red[threadIdx.x] = sf;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = le + p * (red[0] * invF);
}

__global__ void __launch_bounds__(256) k_scale_by_scalar(const float* __restrict__ v, const float* __restrict__ scalar, int n,
                                                          float* __restrict__ g) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) g[i] = v[i] * scalar[0];
}

// Column groups of a row-major matrix <-> one contiguous matrix per group (W = 8 floats = two float4 per row and group):
//   split: outs[l][t][0..7] = in[t][8 l .. 8 l + 7]            merge: out[t][8 l ..] = ins[l][t][0..7]  (ins[l] NULL: zeros)
// The L first basis Linears of the energy_and_force route applied as ONE T-row layer with stacked weights
// (dimenetpp.py:146: rbf/sbf -> lin_sbf1 of every block reads the same [T, ns*nr] table) hand each block its own
// contiguous [T, 8] operand of the fused triplet kernels; the two kernels are each other's adjoint.
struct ColGroups {
  float* p[RG_MAX];
};
__global__ void __launch_bounds__(256) k_cols_split8(const float4* __restrict__ in, int64_t T, int L, ColGroups outs) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;            // float4 index of the stacked matrix
  if (q >= T * 2 * L) return;
  const int64_t t = q / (2 * L);
  const int c = (int)(q - t * 2 * L), l = c >> 1, h = c & 1;
  ((float4*)outs.p[l])[t * 2 + h] = in[q];
}
__global__ void __launch_bounds__(256) k_cols_merge8(ColGroups ins, int64_t T, int L, float4* __restrict__ out) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= T * 2 * L) return;
  const int64_t t = q / (2 * L);
  const int c = (int)(q - t * 2 * L), l = c >> 1, h = c & 1;
  const float4* src = (const float4*)ins.p[l];
  out[q] = src ? src[t * 2 + h] : make_float4(0.f, 0.f, 0.f, 0.f);
}

// dst[r][c] = src[r][c] inside src's [rs, cs], 0 outside: zero padding AND slicing (rd, cd may be larger or smaller) — the
// copy around the MFMA kernels for layer widths that are not multiples of 8 (spherenet.py:253-259 accepts any
// hidden_channels / int_emb_size): the weight's rows, the bias and the residual are padded to the next multiple of 8, the
// result is sliced back.  Linear and its own adjoint with the shapes exchanged, so closed under differentiation.
__global__ void __launch_bounds__(256) k_pad2d(const float* __restrict__ src, int rs, int cs, float* __restrict__ dst, int rd,
                                                int cd) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)rd * cd) return;
  const int r = (int)(i / cd), c = (int)(i - (int64_t)r * cd);
  dst[i] = (r < rs && c < cs) ? src[(int64_t)r * cs + c] : 0.f;
}

// out[arg[s]] = g[s] for arg[s] < n (arg unique among the valid entries; sentinel arg == n: nothing), 0 elsewhere: the
// gradient of torch_scatter.scatter_min w.r.t. its source (each segment's minimum came from exactly one entry).  Two passes
// in one launch-ordered pair: zero fill, then the unique scatter — no atomics.
__global__ void __launch_bounds__(256) k_scatter_unique(const float* __restrict__ g, const int64_t* __restrict__ arg, int S, int n,
                                                         float* __restrict__ out) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  const int64_t a = arg[s];
  if (a >= 0 && a < n) out[a] = g[s];
}

extern "C" {

static int fill_table(PtrTable& t, int G, const void* const* in, void* const* out, const void* const* aux,
                      const void* const* aux2) {
  if (G < 1 || G > RG_MAX) return 0;
  for (int g = 0; g < G; ++g) {
    t.in[g] = in ? (const float*)in[g] : nullptr;
    t.out[g] = out ? (float*)out[g] : nullptr;
    t.aux[g] = aux ? (const float*)aux[g] : nullptr;
    t.aux2[g] = aux2 ? (const float*)aux2[g] : nullptr;
    t.out2[g] = nullptr;
    t.in2[g] = t.add[g] = t.add2[g] = nullptr;
  }
  for (int g = G; g < RG_MAX; ++g) t.in2[g] = t.add[g] = t.add2[g] = nullptr;
  return 1;
}

// out_g[S,C] = CSR segment sums of in_g[M,C] (* mul_g[M,C] when mul != NULL) for G <= 8 tensors sharing one row pointer
// (C in {32,64,128,256}).
int dig3d_segment_sum_grouped2(int G, const void* const* in, const void* const* mul, const void* const* in2,
                               const void* const* mul2, const int* kptr, int S, int C, void* const* out, void* stream);
int dig3d_segment_sum_grouped(int G, const void* const* in, const void* const* mul, const int* kptr, int S, int C,
                              void* const* out, void* stream) {
  return dig3d_segment_sum_grouped2(G, in, mul, nullptr, nullptr, kptr, S, C, out, stream);
}

// the same with a second product per group: out_g = segment sums of in_g * mul_g + in2_g * mul2_g (in2 == NULL: none; entries
// of in2 may be NULL per group) — the double backward of the product form on the energy_and_force route
int dig3d_segment_sum_grouped2(int G, const void* const* in, const void* const* mul, const void* const* in2,
                               const void* const* mul2, const int* kptr, int S, int C, void* const* out, void* stream) {
  DIG3D_ENTER();
  PtrTable t;
  if (!fill_table(t, G, in, out, mul, in2 ? mul2 : nullptr) || S < 0 || !kptr || (in2 && !mul2)) return DIG3D_ERR_ARG;
  if (C != 32 && C != 64 && C != 128 && C != 256) return DIG3D_ERR_ARG;
  for (int g = 0; g < G; ++g) {
    t.in2[g] = in2 ? (const float*)in2[g] : nullptr;
    if (t.in2[g] && (!t.aux2[g] || (((uintptr_t)t.in2[g] | (uintptr_t)t.aux2[g]) & 15))) return DIG3D_ERR_ARG;
  }
  for (int g = 0; g < G; ++g)
    if (!t.in[g] || !t.out[g] || (((uintptr_t)t.in[g] | (uintptr_t)t.out[g]) & 15)) return DIG3D_ERR_ARG;
  if (S == 0) return DIG3D_OK;
  hipStream_t st = (hipStream_t)stream;
  const int lpr = C / 4;
  dim3 grid(dig3d_blocks((int64_t)S * lpr, 256), G);
  if (lpr == 8) hipLaunchKernelGGL((k_segsum_grouped<8>), grid, dim3(256), 0, st, t, kptr, S);
  else if (lpr == 16) hipLaunchKernelGGL((k_segsum_grouped<16>), grid, dim3(256), 0, st, t, kptr, S);
  else if (lpr == 32) hipLaunchKernelGGL((k_segsum_grouped<32>), grid, dim3(256), 0, st, t, kptr, S);
  else hipLaunchKernelGGL((k_segsum_grouped<64>), grid, dim3(256), 0, st, t, kptr, S);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// out_g[M,C] = in_g[ix[m],:] (* mul_g[m,:]) and, when out2 != NULL, out2_g[M,C] = in_g[ix[m],:] * mul2_g[m,:], for G
// tensors sharing one index (C % 4 == 0); rows >= *cnt zero.
int dig3d_gather_grouped_add(int G, const void* const* in, const int* ix, int64_t M, int C, void* const* out,
                             const void* const* mul, void* const* out2, const void* const* mul2, const void* const* add,
                             const void* const* add2, const int* cnt, void* stream);
int dig3d_gather_grouped(int G, const void* const* in, const int* ix, int64_t M, int C, void* const* out,
                         const void* const* mul, void* const* out2, const void* const* mul2, const int* cnt,
                         void* stream) {
  return dig3d_gather_grouped_add(G, in, ix, M, C, out, mul, out2, mul2, nullptr, nullptr, cnt, stream);
}

// the same with addends: out_g += add_g, out2_g += add2_g ([M, C]; NULL arrays or NULL entries: none) inside the launch — the
// second gradient that reaches a factor in the final pass of energy_and_force (see dig3d_triplet_fwd_add)
int dig3d_gather_grouped_add(int G, const void* const* in, const int* ix, int64_t M, int C, void* const* out,
                             const void* const* mul, void* const* out2, const void* const* mul2, const void* const* add,
                             const void* const* add2, const int* cnt, void* stream) {
  DIG3D_ENTER();
  PtrTable t;
  if (!fill_table(t, G, in, out, mul, mul2) || M < 0 || C <= 0 || (C & 3) || !ix || (out2 && !mul2)) return DIG3D_ERR_ARG;
  for (int g = 0; g < G; ++g) {
    t.add[g] = add ? (const float*)add[g] : nullptr;
    t.add2[g] = add2 ? (const float*)add2[g] : nullptr;
    if ((((uintptr_t)t.add[g] | (uintptr_t)t.add2[g]) & 15)) return DIG3D_ERR_ARG;
  }
  for (int g = 0; g < G; ++g) {
    if (!t.in[g] || !t.out[g] || (((uintptr_t)t.in[g] | (uintptr_t)t.out[g]) & 15)) return DIG3D_ERR_ARG;
    if (out2) t.out2[g] = (float*)out2[g];
    if (out2 && (!t.out2[g] || !t.aux2[g])) return DIG3D_ERR_ARG;
  }
  if (M == 0) return DIG3D_OK;
  dim3 grid(dig3d_blocks(M * (C / 4), 256), G);
  hipLaunchKernelGGL(k_gather_grouped, grid, dim3(256), 0, (hipStream_t)stream, t, ix, M, C / 4, cnt);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// y_g[M,N] = x_g[M,K] W_g[N,K]^T (+ bias_g), N <= 8 — the `lin` heads of the output blocks (spherenet.py:216).
int dig3d_smalln_fwd_grouped(int G, const void* const* X, const void* const* W, const void* const* bias, int M, int K,
                             int N, void* const* Y, void* stream) {
  DIG3D_ENTER();
  PtrTable t;
  if (!fill_table(t, G, X, Y, W, bias) || M < 0 || K < 1 || N < 1 || N > RG_MAX) return DIG3D_ERR_ARG;
  for (int g = 0; g < G; ++g)
    if (!t.in[g] || !t.out[g] || !t.aux[g]) return DIG3D_ERR_ARG;
  if (M == 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_smalln_fwd_grouped, dim3((M + 3) / 4, G), dim3(256), 0, (hipStream_t)stream, t, M, K, N);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_smalln_blocks(int M) { return M <= 0 ? 1 : (M + SN_ROWS - 1) / SN_ROWS; }

// gX_g[M,K] = gY_g W_g and the partials of (gW_g [N,K], gb_g [N]) per block of rows (dig3d_smalln_blocks(M) of them):
// part[g]: float[dig3d_smalln_blocks(M) * (N*K + N)]  ->  reduce with dig3d_reduce_many.  gX entries may be NULL.
int dig3d_smalln_bwd_grouped(int G, const void* const* gY, const void* const* W, const void* const* X, int M, int K,
                             int N, void* const* gX, void* const* part, void* stream) {
  DIG3D_ENTER();
  PtrTable t;
  if (!fill_table(t, G, gY, gX, W, X) || M < 0 || K < 1 || N < 1 || N > RG_MAX || !part) return DIG3D_ERR_ARG;
  PartTable pt;
  for (int g = 0; g < G; ++g) {
    pt.part[g] = (float*)part[g];
    // X_g may be NULL when no weight partial is wanted (part[g] NULL): the input-gradient half alone
    if (!t.in[g] || !t.aux[g] || (!t.aux2[g] && pt.part[g])) return DIG3D_ERR_ARG;
  }
  if (M == 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_smalln_bwd_grouped, dim3(dig3d_smalln_blocks(M), G), dim3(256), 0, (hipStream_t)stream, t, pt, M, K,
                     N);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// u[B,C] = sum over the G tensors (in the given order) of their per-graph row sums — update_u of every block
// (spherenet.py:219-225) in the reference's accumulation order.
int dig3d_graph_sum_grouped(int G, const void* const* Y, const int* ptr, int B, int C, float* u, void* stream) {
  DIG3D_ENTER();
  PtrTable t;
  if (!fill_table(t, G, Y, nullptr, nullptr, nullptr) || B < 0 || C < 1 || !ptr || !u) return DIG3D_ERR_ARG;
  for (int g = 0; g < G; ++g)
    if (!t.in[g]) return DIG3D_ERR_ARG;
  if (B == 0) return DIG3D_OK;
  if (C == 1)
    hipLaunchKernelGGL(k_graphsum_grouped_c1, dim3(B), dim3(64), 0, (hipStream_t)stream, t, G, ptr, B, u);
  else
    hipLaunchKernelGGL(k_graphsum_grouped, dim3(dig3d_blocks((int64_t)B * C, 64)), dim3(64), 0, (hipStream_t)stream, t, G,
                       ptr, B, C, u);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// L1 loss of run.py:127 (`loss_func(out, batch_data.y.unsqueeze(1))` with torch.nn.L1Loss(): mean |out - y|) and its
// gradient in two launches instead of the framework's sub / abs / mean / sgn / div / mul / fill chain (8 launches of ~4.6 us
// each in a 2.2 ms step): forward writes the loss and sgn[i] = sign(out_i - y_i) / n (torch.sgn: 0 at 0), backward scales
// it by the incoming scalar gradient read from DEVICE memory (the data-parallel scale of a captured step lives there).
int dig3d_l1_loss_fwd(const float* out, const float* y, int n, float* loss, float* sgn, const float* seed, float* g,
                      void* stream) {
  DIG3D_ENTER();
  if (n < 1 || !out || !y || !loss || !sgn || ((seed == nullptr) != (g == nullptr))) return DIG3D_ERR_ARG;
  hipLaunchKernelGGL(k_l1_loss_fwd, dim3(1), dim3(256), 0, (hipStream_t)stream, out, y, n, loss, sgn, seed, g);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_ef_l1_loss(const float* out, const float* y, int nE, const float* gpos, const float* f, int n3, const int* cntN,
                     float p, const float* seed, float* loss, float* sgn_e, float* sgn_f, float* g_out, float* g_gpos,
                     void* stream) {
  DIG3D_ENTER();
  if (nE < 1 || n3 < 0 || !out || !y || !loss || !sgn_e || (n3 > 0 && (!gpos || !f || !sgn_f)) ||
      ((seed == nullptr) != (g_out == nullptr)) || ((seed == nullptr) != (g_gpos == nullptr)))
    return DIG3D_ERR_ARG;
  hipLaunchKernelGGL(k_ef_l1_loss, dim3(1), dim3(256), 0, (hipStream_t)stream, out, y, nE, gpos, f, n3, cntN, p, seed, loss,
                     sgn_e, sgn_f, g_out, g_gpos);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// g[i] = v[i] * scalar[0]
int dig3d_scale_by_scalar(const float* v, const float* scalar, int n, float* g, void* stream) {
  DIG3D_ENTER();
  if (n < 0 || !v || !scalar || !g) return DIG3D_ERR_ARG;
  if (n == 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_scale_by_scalar, dim3(dig3d_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, v, scalar, n, g);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_pad2d(const float* src, int rs, int cs, float* dst, int rd, int cd, void* stream) {
  DIG3D_ENTER();
  if (rs < 0 || cs < 0 || rd < 0 || cd < 0 || !dst || (!src && (int64_t)rs * cs > 0)) return DIG3D_ERR_ARG;
  if ((int64_t)rd * cd == 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_pad2d, dim3(dig3d_blocks((int64_t)rd * cd, 256)), dim3(256), 0, (hipStream_t)stream, src, rs, cs, dst, rd, cd);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_scatter_unique(const float* g, const int64_t* arg, int S, int n, float* out, void* stream) {
  DIG3D_ENTER();
  if (S < 0 || n < 0 || !out || (S > 0 && (!g || !arg))) return DIG3D_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (n > 0 && dig3d_zero_async(out, sizeof(float) * (size_t)n, st) != hipSuccess) return DIG3D_ERR_LAUNCH;
  if (S == 0 || n == 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_scatter_unique, dim3(dig3d_blocks(S, 256)), dim3(256), 0, st, g, arg, S, n, out);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// in [T, 8 L] row-major -> outs[l] [T, 8] (l < L <= 8); and the adjoint (ins[l] may be NULL: a zero block)
int dig3d_cols_split8(const float* in, int64_t T, int L, void* const* outs, void* stream) {
  DIG3D_ENTER();
  if (T < 0 || L < 1 || L > RG_MAX || !outs || (T > 0 && !in) || (((uintptr_t)in) & 15)) return DIG3D_ERR_ARG;
  ColGroups g;
  for (int l = 0; l < RG_MAX; ++l) {
    g.p[l] = l < L ? (float*)outs[l] : nullptr;
    if (l < L && (!g.p[l] || (((uintptr_t)g.p[l]) & 15))) return DIG3D_ERR_ARG;
  }
  if (T == 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_cols_split8, dim3(dig3d_blocks(T * 2 * L, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)in, T,
                     L, g);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_cols_merge8(const void* const* ins, int64_t T, int L, float* out, void* stream) {
  DIG3D_ENTER();
  if (T < 0 || L < 1 || L > RG_MAX || !ins || (T > 0 && !out) || (((uintptr_t)out) & 15)) return DIG3D_ERR_ARG;
  ColGroups g;
  for (int l = 0; l < RG_MAX; ++l) {
    g.p[l] = l < L ? (float*)ins[l] : nullptr;
    if (g.p[l] && (((uintptr_t)g.p[l]) & 15)) return DIG3D_ERR_ARG;
  }
  if (T == 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_cols_merge8, dim3(dig3d_blocks(T * 2 * L, 256)), dim3(256), 0, (hipStream_t)stream, g, T, L, (float4*)out);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

}  // extern "C"
