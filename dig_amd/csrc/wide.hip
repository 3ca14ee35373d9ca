// dig3d — 256-wide layer chains on row tiles that stay on the CU (r04):
//   * the OUTPUT BLOCKS of SphereNet / DimeNet++ (method/spherenet/spherenet.py:185-216, dimenetpp.py:164-204):
//         h = lin_up(v)  (128 -> 256),   h = swish(lins_j(h))  (256 -> 256, three layers)
//     for the G = L + 1 blocks of a forward as ONE launch per pass (blockIdx.y = block): round 3 ran them as four grouped
//     launches forward (k_linear_fwd_grouped, MFMA busy 9 %) and four backward (k_linear_bwd_both_grouped, 15 %);
//   * ComENet's residual layers (method/comenet/comenet.py:209-210):   h = h + swish(lins_j(h))   (256 -> 256, four layers)
//     — one launch per pass instead of one per layer (M = 16 384 rows, 26-32 us per launch before).
//   Y_l = res_l * Y_{l-1} + act_l(Y_{l-1} W_l^T + b_l),   gZ_l = g_l * act_l'(Z_l),   g_{l-1} = gZ_l W_l + res_l * g_l
//
// The design of chain.hip (128-wide) carried over to 256 outputs:
//   * workgroup = 8 waves on R = 16 * RB rows; wave w owns the output channels [32 w, 32 w + 32) of every row — TWO
//     16-channel MFMA tiles, processed one after the other so that one tile's weight slice (64 VGPRs at K = 256) is in
//     use while the next slice (the layer's second tile, then the next layer's first) is in flight from L2: no weight
//     passes through LDS, no MFMA waits on a load issued in its own section;
//   * the product is formed transposed, D[channel][row] (v_mfma_f32_16x16x4_f32): lane (x, q) ends up with four
//     consecutive channels of row x — bias, activation, residual and the 16-byte stores are lane-local;
//   * weights are re-laid once per step in operand order (k_wide_pack), Wf[w][t][j][lane] = W[32w + 16t + x][16j + 4q + c]
//     for the forward and Wb[w][t][j][lane] = W[16j + 4q + c][32w + 16t + x] for the input gradient: every operand load is a
//     contiguous kilobyte per wave;
//   * only the activation tile goes through LDS (pitch 264 floats, double buffered): one barrier per layer.
// Weight gradients: the GZ_l written by the backward and the layer inputs feed dense.hip's dig3d_wgrad_many with every
// other layer of the step.
#include "dense_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WD_N 256            // output channels of every layer
#define WD_P 264            // LDS pitch of the activation tile (floats): 66 float4 per row, (2x + q) mod 16 distinct per 16 lanes
#define WD_T 512            // threads per workgroup
#define WD_LMAX 4           // layers per launch
#define WD_GMAX 8           // groups per launch (blockIdx.y)

struct WideFwdDesc {
  const float* X0[WD_GMAX];             // [M, K0]
  const float* W[WD_GMAX][WD_LMAX];     // packed forward slices (k_wide_pack)
  const float* bias[WD_GMAX][WD_LMAX];  // [256] or null
  float* Z[WD_GMAX][WD_LMAX];           // pre-activations [M,256] (null: layer without activation)
  float* Y[WD_GMAX][WD_LMAX];           // layer outputs [M,256]
  int K0, nl, M;
  int act[WD_LMAX], res[WD_LMAX];       // act: 0 none / 1 swish; res: 1 = add the layer's input (K must be 256)
  // second-order mode (k_wide_fwd<RB, true>, dig3d_wide_dd): the saved pre-activation and the saved total gradient of the
  // first backward pass per layer; Z then receives t G0 act''(Z0), Y receives t act'(Z0) (+ res), no bias
  const float* Z0[WD_GMAX][WD_LMAX];
  const float* G0[WD_GMAX][WD_LMAX];
};

struct WideBwdDesc {
  const float* gout[WD_GMAX];           // gradient of the last layer's output [M,256]
  const float* W[WD_GMAX][WD_LMAX];     // packed backward slices
  const float* Z[WD_GMAX][WD_LMAX];     // saved pre-activations (null: no activation)
  float* GZ[WD_GMAX][WD_LMAX];          // out: pre-activation gradients [M,256] (operands of the weight gradients)
  float* gx0[WD_GMAX];                  // out: gradient of the chain input [M,K0]
  const float* gadd[WD_GMAX];           // further gradient of the chain input [M,K0] added in the last epilogue, or null
  int K0, nl, M;
  int act[WD_LMAX], res[WD_LMAX];
  // energy_and_force (null otherwise): G receives the total gradient w.r.t. every layer's output (the second-order pass
  // needs it), gzadd is added to the pre-activation gradient (the act'' terms of that pass)
  float* G[WD_GMAX][WD_LMAX];
  const float* gzadd[WD_GMAX][WD_LMAX];
};

__device__ __forceinline__ f32x4 wd_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float wd_swish(float z) { return z * fast_sigmoid(z); }
__device__ __forceinline__ float wd_dswish(float z) {
  const float s = fast_sigmoid(z);
  return s * (1.0f + z * (1.0f - s));
}

__device__ __forceinline__ void wd_d12(float z, float& d1, float& d2) {
  const float s = fast_sigmoid(z);
  d1 = s * (1.0f + z * (1.0f - s));
  d2 = s * (1.0f - s) * (2.0f + z * (1.0f - 2.0f * s));
}

// one 16-channel tile of a layer on the R-row tile: acc[rb] = sum_k W[ch][k] X[row][k]
template <int RB>
__device__ __forceinline__ void wd_tile_mma(const float4 (&w)[16], const float* __restrict__ sIn, int x, int q, int nj,
                                            f32x4 (&acc)[RB]) {
  const float* pa = sIn + x * WD_P + 4 * q;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j < nj) {
      float4 xb[RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) xb[rb] = *(const float4*)(pa + (16 * rb) * WD_P + 16 * j);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[rb] = wd_mfma(w[j].x, xb[rb].x, acc[rb]);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[rb] = wd_mfma(w[j].y, xb[rb].y, acc[rb]);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[rb] = wd_mfma(w[j].z, xb[rb].z, acc[rb]);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[rb] = wd_mfma(w[j].w, xb[rb].w, acc[rb]);
    }
  }
}

// packed slice (w, t) of one weight: 16 float4 per lane (the upper eight are not loaded when nj == 8)
__device__ __forceinline__ void wd_fetch(const float* __restrict__ Wp, int wave, int t, int lane, int nj, float4 (&w)[16]) {
  const float* __restrict__ p = Wp + ((int64_t)((wave * 2 + t) * nj) * 64 + lane) * 4;
#pragma unroll
  for (int j = 0; j < 16; ++j)
    if (j < nj) w[j] = *(const float4*)(p + j * 256);
}

template <int RB, bool DD = false>
__global__ void __launch_bounds__(WD_T) k_wide_fwd(WideFwdDesc d) {
  extern __shared__ float wsm[];
  constexpr int R = 16 * RB;
  const int g = blockIdx.y;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, x = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.x * R, M = d.M, nl = d.nl, K0 = d.K0;
  float* sA = wsm;
  float* sB = wsm + R * WD_P;
  {                                                  // input tile (rows beyond M: zeros)
    const float* __restrict__ X0 = d.X0[g];
    const int c4 = (tid & 63) * 4, r0 = tid >> 6;
    if (c4 < K0) {
#pragma unroll
      for (int it = 0; it < 2 * RB; ++it) {
        const int r = r0 + 8 * it, m = m0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < M) v = *(const float4*)(X0 + (int64_t)m * K0 + c4);
        *(float4*)(sA + r * WD_P + c4) = v;
      }
    }
  }
  float4 wa[16], wb[16];
  wd_fetch(d.W[g][0], wave, 0, lane, K0 >> 4, wa);
  __syncthreads();
  float* sIn = sA;
  float* sOut = sB;
  for (int l = 0; l < nl; ++l) {
    const int nj = (l == 0 ? K0 : WD_N) >> 4;
    const bool act = d.act[l] != 0, res = d.res[l] != 0;
    const float* __restrict__ bias = d.bias[g][l];
    float* __restrict__ Zo = d.Z[g][l];
    float* __restrict__ Yo = d.Y[g][l];
    f32x4 acc[RB];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      // the slice needed NEXT is requested before this tile's MFMAs: (l, 1) during tile 0, (l + 1, 0) during tile 1
      if (t == 0) wd_fetch(d.W[g][l], wave, 1, lane, nj, wb);
      else if (l + 1 < nl) wd_fetch(d.W[g][l + 1], wave, 0, lane, WD_N >> 4, wa);
      if (t == 0) wd_tile_mma<RB>(wa, sIn, x, q, nj, acc);
      else wd_tile_mma<RB>(wb, sIn, x, q, nj, acc);
      const int ch = 32 * wave + 16 * t + 4 * q;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!DD && bias) bv = *(const float4*)(bias + ch);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const int r = 16 * rb + x, m = m0 + r;
        float4 z = make_float4(acc[rb][0] + bv.x, acc[rb][1] + bv.y, acc[rb][2] + bv.z, acc[rb][3] + bv.w);
        float4 y;
        if (DD) {
          y = z;
          if (act) {                                 // y = t act'(Z0),  z = t G0 act''(Z0)
            const int64_t o = (int64_t)(m < M ? m : M - 1) * WD_N + ch;
            const float4 z0 = *(const float4*)(d.Z0[g][l] + o), g0 = *(const float4*)(d.G0[g][l] + o);
            const float4 tt = z;
            float d1, d2;
            wd_d12(z0.x, d1, d2); y.x = tt.x * d1; z.x = tt.x * g0.x * d2;
            wd_d12(z0.y, d1, d2); y.y = tt.y * d1; z.y = tt.y * g0.y * d2;
            wd_d12(z0.z, d1, d2); y.z = tt.z * d1; z.z = tt.z * g0.z * d2;
            wd_d12(z0.w, d1, d2); y.w = tt.w * d1; z.w = tt.w * g0.w * d2;
          }
        } else {
          y = act ? make_float4(wd_swish(z.x), wd_swish(z.y), wd_swish(z.z), wd_swish(z.w)) : z;
        }
        if (res) {
          const float4 in = *(const float4*)(sIn + r * WD_P + ch);
          y = make_float4(in.x + y.x, in.y + y.y, in.z + y.z, in.w + y.w);
        }
        if (m < M) {
          if (Zo) *(float4*)(Zo + (int64_t)m * WD_N + ch) = z;
          *(float4*)(Yo + (int64_t)m * WD_N + ch) = y;
        }
        *(float4*)(sOut + r * WD_P + ch) = y;
      }
    }
    __syncthreads();                                 // sOut complete; everyone is done reading sIn
    float* tmp = sIn;
    sIn = sOut;
    sOut = tmp;
  }
}

template <int RB>
__global__ void __launch_bounds__(WD_T) k_wide_bwd(WideBwdDesc d) {
  extern __shared__ float wsm[];
  constexpr int R = 16 * RB;
  const int g = blockIdx.y;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, x = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.x * R, M = d.M, nl = d.nl, K0 = d.K0;
  float* sA = wsm;
  float* sB = wsm + R * WD_P;
  // this lane's slots: channels 32 wave + 16 t + 4 q .. + 3 of rows 16 rb + x, for t = 0, 1
  float4 gr[2][RB];
  {
    const float* __restrict__ go = d.gout[g];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const int m = m0 + 16 * rb + x;
        gr[t][rb] = m < M ? *(const float4*)(go + (int64_t)m * WD_N + 32 * wave + 16 * t + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
  }
  float4 wa[16], wb[16];
  wd_fetch(d.W[g][nl - 1], wave, 0, lane, 16, wa);
  float* sG = sA;
  for (int l = nl - 1; l >= 0; --l) {
    const int K = l == 0 ? K0 : WD_N;
    const bool act = d.act[l] != 0, res = d.res[l] != 0;
    const float* __restrict__ Zl = d.Z[g][l];
    float* __restrict__ GZ = d.GZ[g][l];
    // gZ_l = g_l * act'(Z_l) -> global (weight-gradient operand) and the LDS tile (B operand of the product below)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ch = 32 * wave + 16 * t + 4 * q;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const int r = 16 * rb + x, m = m0 + r;
        float4 gz = gr[t][rb];
        if (d.G[g][l] && m < M) *(float4*)(d.G[g][l] + (int64_t)m * WD_N + ch) = gz;
        if (act && Zl) {
          const float4 z = m < M ? *(const float4*)(Zl + (int64_t)m * WD_N + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
          gz = make_float4(gz.x * wd_dswish(z.x), gz.y * wd_dswish(z.y), gz.z * wd_dswish(z.z), gz.w * wd_dswish(z.w));
        }
        if (d.gzadd[g][l] && m < M) {
          const float4 a = *(const float4*)(d.gzadd[g][l] + (int64_t)m * WD_N + ch);
          gz = make_float4(gz.x + a.x, gz.y + a.y, gz.z + a.z, gz.w + a.w);
        }
        if (m < M) *(float4*)(GZ + (int64_t)m * WD_N + ch) = gz;
        *(float4*)(sG + r * WD_P + ch) = gz;
      }
    }
    __syncthreads();                                 // gZ tile complete (the other buffer is free: everyone is past its MFMAs)
    const bool live = 32 * wave < K;                 // lin_up (K = 128): the upper four waves have no input channel
    f32x4 acc[RB];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t == 0) wd_fetch(d.W[g][l], wave, 1, lane, 16, wb);
      else if (l > 0) wd_fetch(d.W[g][l - 1], wave, 0, lane, 16, wa);
      if (live) {
        if (t == 0) wd_tile_mma<RB>(wa, sG, x, q, 16, acc);
        else wd_tile_mma<RB>(wb, sG, x, q, 16, acc);
      } else {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        float4 gp = make_float4(acc[rb][0], acc[rb][1], acc[rb][2], acc[rb][3]);
        if (res) gp = make_float4(gp.x + gr[t][rb].x, gp.y + gr[t][rb].y, gp.z + gr[t][rb].z, gp.w + gr[t][rb].w);
        gr[t][rb] = gp;
      }
    }
    sG = sG == sA ? sB : sA;
  }
  if (32 * wave < K0) {
    float* __restrict__ gx0 = d.gx0[g];
    const float* __restrict__ ga = d.gadd[g];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const int m = m0 + 16 * rb + x;
        const int64_t o = (int64_t)m * K0 + 32 * wave + 16 * t + 4 * q;
        if (m < M) {
          float4 v = gr[t][rb];
          if (ga) {
            const float4 a = *(const float4*)(ga + o);
            v = make_float4(v.x + a.x, v.y + a.y, v.z + a.z, v.w + a.w);
          }
          *(float4*)(gx0 + o) = v;
        }
      }
  }
}

// operand-order copies of up to 32 weights W [256, K] (K = 128 or 256): fwd[i] and bwd[i], 256 * K floats each
#define WD_PACK_MAX 32
struct WidePackDesc {
  const float* W[WD_PACK_MAX];
  float* fwd[WD_PACK_MAX];
  float* bwd[WD_PACK_MAX];
  int K[WD_PACK_MAX];
};
__global__ void __launch_bounds__(256) k_wide_pack(WidePackDesc d) {
  const int i = blockIdx.y;
  const int K = d.K[i], nj = K >> 4;
  const float* __restrict__ W = d.W[i];
  // forward: element e = (((w*2 + t)*nj + j)*64 + lane)*4 + c  <-  W[32w + 16t + x][16j + 4q + c]
  for (int e = blockIdx.x * 256 + threadIdx.x; e < WD_N * K; e += gridDim.x * 256) {
    const int c = e & 3, lane = (e >> 2) & 63, rest = e >> 8, j = rest % nj, wt = rest / nj;
    const int x = lane & 15, q = lane >> 4;
    d.fwd[i][e] = W[(int64_t)(16 * wt + x) * K + 16 * j + 4 * q + c];
  }
  // backward: element e = (((w*2 + t)*16 + j)*64 + lane)*4 + c  <-  W[16j + 4q + c][32w + 16t + x]   (zero beyond K columns)
  for (int e = blockIdx.x * 256 + threadIdx.x; e < WD_N * WD_N; e += gridDim.x * 256) {
    const int c = e & 3, lane = (e >> 2) & 63, rest = e >> 8, j = rest & 15, wt = rest >> 4;
    const int x = lane & 15, q = lane >> 4;
    const int col = 16 * wt + x;
    if (col < K) d.bwd[i][(int64_t)(wt * 16 + j) * 256 + lane * 4 + c] = W[(int64_t)(16 * j + 4 * q + c) * K + col];
  }
}

static int wd_rb(int M, int G) {
  // rows per workgroup: the grid should fill the CUs once; more rows per workgroup = fewer copies of the weight stream
  const int cus = dig3d_num_cus();
  for (int rb = 1; rb < 4; ++rb)
    if ((int64_t)((M + 16 * rb - 1) / (16 * rb)) * G <= cus) return rb;
  return 4;
}

extern "C" {

// n <= 32 weights W[i] [256, K[i]] (K = 128 or 256) -> fwd[i] (256*K floats) and bwd[i] (65536 floats; only the first
// 256*K are meaningful when K = 128).  Host arrays of device pointers.
int dig3d_wide_pack(int n, const void* const* W, const int* K, void* const* fwd, void* const* bwd, void* stream) {
  DIG3D_ENTER();
  if (n < 1 || n > WD_PACK_MAX || !W || !K || !fwd || !bwd) return DIG3D_ERR_ARG;
  WidePackDesc d;
  for (int i = 0; i < n; ++i) {
    if (!W[i] || !fwd[i] || !bwd[i] || (K[i] != 128 && K[i] != 256)) return DIG3D_ERR_ARG;
    d.W[i] = (const float*)W[i];
    d.fwd[i] = (float*)fwd[i];
    d.bwd[i] = (float*)bwd[i];
    d.K[i] = K[i];
  }
  hipLaunchKernelGGL(k_wide_pack, dim3(256, n), dim3(256), 0, (hipStream_t)stream, d);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_wide_supported(int M, int K0, int nl, int G) {
  return M > 0 && (K0 == 128 || K0 == 256) && nl >= 1 && nl <= WD_LMAX && G >= 1 && G <= WD_GMAX;
}

// G groups x nl layers.  Host arrays, group-major: X0[G]; Wp / bias / Z / Y [G * nl]; act / res [nl].
// Y_l = res_l * Y_{l-1} + act_l(Y_{l-1} W_l^T + b_l); Z (pre-activation) is written where non-null.
static int wide_fwd_impl(int G, int nl, int M, int K0, const void* const* X0, const void* const* Wp, const void* const* bias,
                         void* const* Z, void* const* Y, const int* act, const int* res, const void* const* Z0,
                         const void* const* G0, void* stream) {
  DIG3D_ENTER();
  const bool dd = Z0 != nullptr;
  if (!dig3d_wide_supported(M, K0, nl, G) || !X0 || !Wp || (!dd && !bias) || !Z || !Y || !act || !res || (dd && !G0))
    return DIG3D_ERR_ARG;
  WideFwdDesc d;
  d.K0 = K0; d.nl = nl; d.M = M;
  for (int l = 0; l < nl; ++l) {
    d.act[l] = act[l];
    d.res[l] = res[l];
    if (res[l] && l == 0 && K0 != WD_N) return DIG3D_ERR_ARG;
  }
  for (int g = 0; g < G; ++g) {
    if (!X0[g] || !al16(X0[g])) return DIG3D_ERR_ARG;
    d.X0[g] = (const float*)X0[g];
    for (int l = 0; l < nl; ++l) {
      const int i = g * nl + l;
      if (!Wp[i] || !Y[i] || !al16(Wp[i]) || !al16(Y[i]) || !al16(Z[i]) || (!dd && !al16(bias[i]))) return DIG3D_ERR_ARG;
      d.W[g][l] = (const float*)Wp[i];
      d.bias[g][l] = dd ? nullptr : (const float*)bias[i];
      d.Z[g][l] = (float*)Z[i];
      d.Y[g][l] = (float*)Y[i];
      d.Z0[g][l] = dd ? (const float*)Z0[i] : nullptr;
      d.G0[g][l] = dd ? (const float*)G0[i] : nullptr;
      if (dd && act[l] && (!d.Z0[g][l] || !d.G0[g][l] || !al16(d.Z0[g][l]) || !al16(d.G0[g][l]))) return DIG3D_ERR_ARG;
    }
  }
  const int rb = wd_rb(M, G);
  const dim3 grid((M + 16 * rb - 1) / (16 * rb), G);
  const size_t shm = sizeof(float) * 2 * 16 * rb * WD_P;
#define WD_F(RB, DD_)                                                                                                       \
  {                                                                                                                         \
    static const bool ok = hipFuncSetAttribute((const void*)k_wide_fwd<RB, DD_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                               (int)(sizeof(float) * 2 * 16 * RB * WD_P)) == hipSuccess;                    \
    if (!ok) return DIG3D_ERR_LAUNCH;                                                                                       \
    hipLaunchKernelGGL((k_wide_fwd<RB, DD_>), grid, dim3(WD_T), shm, (hipStream_t)stream, d);                               \
  }
  if (dd) {
    switch (rb) {
      case 1: WD_F(1, true) break;
      case 2: WD_F(2, true) break;
      case 3: WD_F(3, true) break;
      default: WD_F(4, true) break;
    }
  } else {
    switch (rb) {
      case 1: WD_F(1, false) break;
      case 2: WD_F(2, false) break;
      case 3: WD_F(3, false) break;
      default: WD_F(4, false) break;
    }
  }
#undef WD_F
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_wide_fwd(int G, int nl, int M, int K0, const void* const* X0, const void* const* Wp, const void* const* bias,
                   void* const* Z, void* const* Y, const int* act, const int* res, void* stream) {
  return wide_fwd_impl(G, nl, M, K0, X0, Wp, bias, Z, Y, act, res, nullptr, nullptr, stream);
}

// The SECOND-order pass of the chain (energy_and_force: the backward of dig3d_wide_bwd w.r.t. gout and the pre-activations),
// the forward's products in the forward's layer order on H0[g] [M,K0] = gradient w.r.t. gx0[g]:
//   t_l = U_{l-1} W_l^T (U_{-1} = H0),   U_l = t_l act'(Z0_l) + res_l U_{l-1},   HZ_l = t_l G0_l act''(Z0_l)
// Z0 / G0 [G * nl]: the saved pre-activations and the G written by dig3d_wide_bwd (NULL entries for layers without
// activation: U_l = t_l, no HZ_l).  Out: U [G * nl] (U[.., nl-1] = gradient w.r.t. gout; U_{l-1} is the X operand of layer
// l's weight gradient in this pass, GZ_l of dig3d_wide_bwd the other), HZ [G * nl] (NULL where there is no activation).
int dig3d_wide_dd(int G, int nl, int M, int K0, const void* const* H0, const void* const* Wp, const void* const* Z0,
                  const void* const* G0, void* const* HZ, void* const* U, const int* act, const int* res, void* stream) {
  if (!Z0 || !G0) return DIG3D_ERR_ARG;
  return wide_fwd_impl(G, nl, M, K0, H0, Wp, nullptr, HZ, U, act, res, Z0, G0, stream);
}

// Input-gradient recursion of the same chain.  gout[G] [M,256]; Wp: packed BACKWARD slices [G * nl]; Z [G * nl] (null: no
// activation); out: GZ [G * nl] [M,256], gx0[G] [M,K0] (+ gadd[g] when non-null).
int dig3d_wide_bwd(int G, int nl, int M, int K0, const void* const* gout, const void* const* Wp, const void* const* Z,
                   void* const* GZ, void* const* gx0, const void* const* gadd, const int* act, const int* res,
                   void* const* Gout, const void* const* gzadd, void* stream) {
  DIG3D_ENTER();
  if (!dig3d_wide_supported(M, K0, nl, G) || !gout || !Wp || !Z || !GZ || !gx0 || !act || !res) return DIG3D_ERR_ARG;
  WideBwdDesc d;
  d.K0 = K0; d.nl = nl; d.M = M;
  for (int l = 0; l < nl; ++l) {
    d.act[l] = act[l];
    d.res[l] = res[l];
  }
  for (int g = 0; g < G; ++g) {
    if (!gout[g] || !gx0[g] || !al16(gout[g]) || !al16(gx0[g])) return DIG3D_ERR_ARG;
    d.gout[g] = (const float*)gout[g];
    d.gx0[g] = (float*)gx0[g];
    d.gadd[g] = gadd ? (const float*)gadd[g] : nullptr;
    if (d.gadd[g] && !al16(d.gadd[g])) return DIG3D_ERR_ARG;
    for (int l = 0; l < nl; ++l) {
      const int i = g * nl + l;
      if (!Wp[i] || !GZ[i] || !al16(Wp[i]) || !al16(GZ[i]) || !al16(Z[i])) return DIG3D_ERR_ARG;
      d.W[g][l] = (const float*)Wp[i];
      d.Z[g][l] = (const float*)Z[i];
      d.GZ[g][l] = (float*)GZ[i];
      d.G[g][l] = Gout ? (float*)Gout[i] : nullptr;
      d.gzadd[g][l] = gzadd ? (const float*)gzadd[i] : nullptr;
      if (!al16(d.G[g][l]) || !al16(d.gzadd[g][l])) return DIG3D_ERR_ARG;
    }
  }
  const int rb = wd_rb(M, G);
  const dim3 grid((M + 16 * rb - 1) / (16 * rb), G);
  const size_t shm = sizeof(float) * 2 * 16 * rb * WD_P;
#define WD_B(RB)                                                                                                       \
  {                                                                                                                    \
    static const bool ok = hipFuncSetAttribute((const void*)k_wide_bwd<RB>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                               (int)(sizeof(float) * 2 * 16 * RB * WD_P)) == hipSuccess;               \
    if (!ok) return DIG3D_ERR_LAUNCH;                                                                                  \
    hipLaunchKernelGGL(k_wide_bwd<RB>, grid, dim3(WD_T), shm, (hipStream_t)stream, d);                                 \
  }
  switch (rb) {
    case 1: WD_B(1) break;
    case 2: WD_B(2) break;
    case 3: WD_B(3) break;
    default: WD_B(4) break;
  }
#undef WD_B
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

}  // extern "C"
