// Per-edge values of the model front (one implementation for the per-stage kernels AND the fused launch dig3d_edge_front):
//   edge_dist_value     geometric_computing.py:25 / schnet.py:158 / comenet.py:297-298
//   distemb_value       spherenet/features.py:151-182 (the reference's float32 formula, term by term)
//   bessel_value        spherenet/features.py:185-216 / dimenetpp/features.py:195-216 (double, as sympy's lambdified table)
#pragma once
#include "common.h"

// mode 0: sqrt(sum((pos[i]-pos[j])^2)); mode 1: (pos[j]-pos[i]).norm()
__device__ __forceinline__ float edge_dist_value(const float* __restrict__ pos, int s, int d, int mode) {
  f3 pj = load3(pos, s), pi = load3(pos, d);
  return mode == 0 ? ref_len(f3_sub(pi, pj)) : ref_norm(f3_sub(pj, pi));
}

// Envelope(d / cutoff) * sin(freq * d / cutoff), p_ = exponent + 1
__device__ __forceinline__ float distemb_value(float dist, float freq, float cutoff, int p_) {
  // x.pow(p-1), two more multiplies, 1/x + a x0 + ...
  const float p = (float)p_;
  const float a = -(p + 1) * (p + 2) / 2, b = p * (p + 2), c = -p * (p + 1) / 2;
  const float x = dist / cutoff;
  float x0 = 1.f;
  for (int k = 0; k < p_ - 1; ++k) x0 *= x;
  const float x1 = x0 * x, x2 = x1 * x;
  const float env = 1.0f / x + a * x0 + b * x1 + c * x2;
  return env * sinf(freq * x);
}

// norm[l,n] * j_l(z[l,n] * x) * (envelope(x) if env_p > 0), x = dist / cutoff, column ln = l * nr + n
__device__ __forceinline__ float bessel_value(float dist, float cutoff, int ln, int l, const double* __restrict__ zeros,
                                              const double* __restrict__ norms, int env_p) {
  double x = (double)(dist / cutoff);
  double u = zeros[ln] * x;
  double s, c;
  sincos(u, &s, &c);
  double jm = s / u;  // j_0
  double j = jm;
  if (l >= 1) {
    j = s / (u * u) - c / u;  // j_1
    for (int a = 1; a < l; ++a) {
      double jn = (2 * a + 1) / u * j - jm;
      jm = j;
      j = jn;
    }
  }
  double v = norms[ln] * j;
  if (env_p > 0) {
    // Envelope (features.py:151-164): p = exponent+1; 1/x + a x^(p-1) + b x^p + c x^(p+1)
    double p = (double)env_p;
    double a = -(p + 1) * (p + 2) / 2, b = p * (p + 2), cc = -p * (p + 1) / 2;
    double x0 = 1.0;
    for (int k = 0; k < env_p - 1; ++k) x0 *= x;
    double x1 = x0 * x, x2 = x1 * x;
    v *= 1.0 / x + a * x0 + b * x1 + cc * x2;
  }
  return (float)v;
}
