// matrix-core route of the first basis Linears (basis_mfma.hip); 0 = launched, 1 = shape not covered (VALU kernels run)
#pragma once
#include "common.h"

int basis_project_mfma(const float* bes, const int* kj, const float* angle, const float* torsion, int T, int ns, int nr,
                       const float* pref, const float* Ws, const float* Wt, int L, float* Ps, float* Pt, const int* cnt,
                       int form, hipStream_t st);
int basis_wgrad_mfma(const float* bes, const int* kj, const float* angle, const float* torsion, int T, int ns, int nr,
                     const float* pref, const float* gPs, const float* gPt, int L, float* part, const int* cnt, int nb,
                     hipStream_t st);
