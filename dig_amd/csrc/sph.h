// Real spherical harmonics shared by basis.hip (table kernels) and triplet.hip (fused projection).
#pragma once
#include "common.h"

#define NS_MAX 8

// Real spherical harmonics up to degree NS-1 for one (theta, phi):
//   Y[l*l + 0] = K_l0 P~_l^0,  Y[l*l + m] = K_lm C_m P~_l^m,  Y[l*l + 2l+1-m] = K_lm S_m P~_l^m
// with P~ the associated Legendre table WITHOUT the sin^m factor (features.py:75-96) and
// C_m + i S_m = (sin(theta) e^{i phi})^m (features.py:104-115).  pref[l*NS_MAX+m] carries
// sqrt(2) (m>0) and, for ComENet's GemNet convention, the extra (-1)^m.
template <int NS>
__device__ __forceinline__ void real_sph_harm(float theta, float phi, const float* __restrict__ pref,
                                              bool zero_m_only, float* __restrict__ Y) {
  float ct, st;
  sincosf(theta, &st, &ct);     // one argument reduction for both (sinf + cosf separately: ~290 instructions, sincosf ~160)
  float P[NS][NS];
#pragma unroll
  for (int m = 0; m < NS; ++m) {
    if (m == 0) P[0][0] = 1.f; else P[m][m] = (float)(1 - 2 * m) * P[m - 1][m - 1];
    if (m + 1 < NS) P[m + 1][m] = (float)(2 * m + 1) * ct * P[m][m];
#pragma unroll
    for (int l = m + 2; l < NS; ++l)
      P[l][m] = ((float)(2 * l - 1) * ct * P[l - 1][m] - (float)(l + m - 1) * P[l - 2][m]) / (float)(l - m);
    if (zero_m_only) break;
  }
  if (zero_m_only) {
#pragma unroll
    for (int l = 0; l < NS; ++l) Y[l] = pref[l * NS_MAX] * P[l][0];
    return;
  }
  float cp, sp;
  sincosf(phi, &sp, &cp);
  float x = st * cp, y = st * sp;
  float Cm[NS], Sm[NS];
  Cm[0] = 1.f;
  Sm[0] = 0.f;
#pragma unroll
  for (int m = 1; m < NS; ++m) {
    Sm[m] = x * Sm[m - 1] + y * Cm[m - 1];
    Cm[m] = x * Cm[m - 1] - y * Sm[m - 1];
  }
#pragma unroll
  for (int l = 0; l < NS; ++l) {
    Y[l * l] = pref[l * NS_MAX] * P[l][0];
#pragma unroll
    for (int m = 1; m <= l; ++m) {
      float k = pref[l * NS_MAX + m] * P[l][m];
      Y[l * l + m] = k * Cm[m];
      Y[l * l + 2 * l + 1 - m] = k * Sm[m];
    }
  }
}

