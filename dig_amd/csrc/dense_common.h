// dig3d dense layers: pieces shared by dense.hip and chain.hip (activations, the chain descriptors).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ACT_NONE 0
#define ACT_SWISH 1      // x * sigmoid(x)                      (spherenet.py:14-15, comenet.py swish)
#define ACT_SSP 2        // softplus(x) - log(2)                (schnet.py:97-103)
#define ACT_DERIV 3      // BACKWARD kernels: the Z operand already holds act'(z) (written by a forward launched with
                         // act | ACT_KEEP_DERIV), gZ = gY * Z — the exp / reciprocal are not redone by the input-gradient
                         // kernel (once per column slice) and the weight-gradient kernel (once per k tile)
#define ACT_KEEP_DERIV 4 // FORWARD flag (ACT_SWISH | 4, ACT_SSP | 4): the Z output receives act'(z) instead of z
#define ACT_ROWSCALE 7   // FORWARD (dig3d_linear_fwd_rowscale only): Y = rs[m] * (x W^T + b), rs [M] in the res slot, and Z
                         // receives rs[m] — the layer's "derivative tensor" for a backward with ACT_DERIV
#define ACT_D2 8         // ACT_D2 + act: second-order epilogue of k_linear_fwd (see linear_fwd_body)


// swish through v_exp_f32 / v_rcp_f32 (each ~1 ulp): the IEEE-exact expf + correctly rounded division this file is
// otherwise compiled with cost 1.4 us per layer on an 8.7k-row tile set (ablation of k_chain_fwd).
// exp(-z) = 2^t, t = -z log2(e).  Rounding t to float32 alone costs |t| 2^-24 RELATIVE error in the result (r03's
// `__expf`: 6e-7 at |z| = 10 — several ulp, in every activation of every layer, forward and backward).  The product's
// rounding error is recovered exactly with one fma (plus the low word of log2 e) and applied as the first-order factor
// 2^r = 1 + r ln 2: three extra instructions, result within ~1.5 ulp for every z.
__device__ __forceinline__ float exp_neg(float z) {
  const float L2E_HI = 1.44269502162933349609375f;          // float32(log2 e)
  const float L2E_LO = 1.92596299112661746e-8f;             // log2 e - L2E_HI
  const float t = -z * L2E_HI;
  float r = __fmaf_rn(-z, L2E_HI, -t);                      // exact residual of the rounded product
  r = __fmaf_rn(-z, L2E_LO, r);
  r = (fabsf(t) <= 3.0e38f) ? r : (t != t ? t : 0.0f);     // z = +-inf: the residual above is inf - inf (NaN stays NaN)
  const float e = __builtin_amdgcn_exp2f(fminf(t, 126.0f));   // finite for any finite z: inf * r below would be NaN
  return __fmaf_rn(e, r * 0.693147180559945309417f, e);
}
__device__ __forceinline__ float fast_sigmoid(float z) { return __frcp_rn(1.0f + exp_neg(z)); }

__device__ __forceinline__ float act_fwd(float z, int act) {
  if (act == ACT_SWISH) return z * fast_sigmoid(z);
  if (act == ACT_SSP) return (z > 20.0f ? z : log1pf(expf(z))) - 0.69314718055994530942f;
  return z;
}
// first and second derivative of the activation (IEEE expf: these feed the double backward of the force path)
__device__ __forceinline__ void act_d12(float z, int act, float& d1, float& d2) {
  if (act == ACT_SWISH) {
    const float s = 1.0f / (1.0f + expf(-z));
    d1 = s * (1.0f + z * (1.0f - s));
    d2 = s * (1.0f - s) * (2.0f + z * (1.0f - 2.0f * s));
  } else if (act == ACT_SSP) {
    const float s = 1.0f / (1.0f + expf(-z));
    d1 = s;
    d2 = s * (1.0f - s);
  } else {
    d1 = 1.0f;
    d2 = 0.0f;
  }
}
__device__ __forceinline__ float act_bwd(float z, int act) {
  if (act == ACT_DERIV) return z;
  if (act == ACT_SWISH) {
    const float s = fast_sigmoid(z);
    return s * (1.0f + z * (1.0f - s));
  }
  if (act == ACT_SSP) return z > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-z));
  return 1.0f;
}


static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---- descriptors of the layer-chain kernels (dense.hip: k_chain_fwd<true>, chain.hip: k_chainr_fwd / k_chainr_bwd) ----
#define CH_MAX 8
struct ChainDesc {
  const float* W[CH_MAX];
  const float* bias[CH_MAX];
  const float* resext[CH_MAX];   // external residual [M,128] or null
  float* Z[CH_MAX];              // pre-activation out (or null when act == none)
  float* Y[CH_MAX];              // layer output
  int K[CH_MAX];
  int res[CH_MAX];               // 0 none, 1 external, 2 saved tile
  int save[CH_MAX];              // keep Y_l as the saved (skip) tile
  int act[CH_MAX];
  int nl;
  // chain.hip only (first-order kernels on packed weights; dense.hip leaves them unset):
  int N[CH_MAX];                 // outputs of layer l: 128, or fewer (multiple of 16: lin_down has int_emb_size = 64)
  const float* mul[CH_MAX];      // external multiplicative operand [M,128] applied after the activation, or null
  int inbuf[CH_MAX];             // LDS buffer (0 / 1) holding the layer's input tile
  int outbuf[CH_MAX];            // LDS buffer receiving its output tile, -1: the output only goes to memory
  // second-order mode (k_chain_fwd<true>, see dig3d_chain_dd): the saved pre-activation and the saved total gradient of
  // the first backward pass, per layer
  const float* Z0[CH_MAX];
  const float* G0[CH_MAX];
  // chain.hip:k_front_dd only — the lin_kj layer of the front's second-order pass (see there): the gradient gm that reached
  // the product t = swish(z_kj) * rb in the first backward pass, the incoming gradient w.r.t. grb (or null), and the output
  // that receives the gradient w.r.t. rb
  const float* fk_gm;
  const float* fk_v;
  float* fk_drb;
};


struct ChainBwdDesc {
  const float* W[CH_MAX];
  const float* Z[CH_MAX];        // pre-activation saved by the forward (null when act == none)
  float* GZ[CH_MAX];             // out: gradient w.r.t. the pre-activation [M,128]
  float* gres[CH_MAX];           // out: gradient of the external residual of layer l [M,128] (res == 1), else null
  float* G[CH_MAX];              // out (optional): total gradient w.r.t. the layer output (second-order pass needs it)
  const float* gzadd[CH_MAX];    // in (optional): gradient that reached Z_l directly (act'' term of the force path)
  int K[CH_MAX];
  int res[CH_MAX];
  int save[CH_MAX];
  int act[CH_MAX];
  int nl;
};


