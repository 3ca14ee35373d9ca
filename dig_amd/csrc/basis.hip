// dig3d basis functions: radial Bessel basis, real spherical harmonics, and their products.
//   method/spherenet/features.py:149-263   (dist_emb, angle_emb, torsion_emb)
//   method/dimenetpp/features.py:149-220   (angle_emb with envelope)
//   method/comenet/features.py:257-348     (angle_emb, torsion_emb; GemNet harmonics)
//   method/schnet/schnet.py:85-94,29-32    (gaussian smearing, cosine cutoff)
// The reference builds one sympy-lambdified torch call per basis function (ns*nr + ns^2 separate
// launches, 22-78 s of symbolic work per constructor).  Here the closed forms are evaluated directly:
// spherical Bessel j_l by upward recurrence from sin/cos IN FLOAT64 (the float32 closed forms the
// reference evaluates lose all digits for small z*d/c at high l — DESIGN.md "basis accuracy"), real
// harmonics by the Legendre / (x+iy)^m recurrences the reference's formulas are generated from.
// Constants (zeros of j_l rounded to float32, normalisers, harmonic prefactors) come from the host,
// computed exactly as the reference does (scipy brentq) — see dig_amd/threedgraph/method/basis.py.
#include "common.h"
#include "edge_values.h"

#include "sph.h"


// bes[e, l*nr+n] = norm[l,n] * j_l(z[l,n] * x) * (envelope(x) if env_p > 0),  x = dist/cutoff.
__global__ void k_bessel(const float* __restrict__ dist, int E, float cutoff, int ns, int nr,
                         const double* __restrict__ zeros, const double* __restrict__ norms, int env_p,
                         float* __restrict__ out) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int K = ns * nr;
  if (q >= (int64_t)E * K) return;
  int e = (int)(q / K);
  int ln = (int)(q - (int64_t)e * K);
  int l = ln / nr;
  out[q] = bessel_value(dist[e], cutoff, ln, l, zeros, norms, env_p);
}

// out[m, h*nr + n] = Y_h(theta[m], phi[m]) * bes[g(m), order(h)*nr + n]
//   zero_m_only (phi == nullptr): h = l in [0,NS), order(h) = l               -> [M, NS*nr]
//   pair_mode 0 (SphereNet features.py:262 broadcast): order(h) = h % NS      -> [M, NS*NS*nr]
//   pair_mode 1 (ComENet features.py:346-348):          order(h) = degree l(h)
// g(m) = gidx ? gidx[m] : m.  One block handles TPB rows: harmonics are computed one row per
// thread into LDS, then the block writes the product coalesced.
#define SPH_TPB 128
template <int NS>
__global__ void __launch_bounds__(SPH_TPB) k_sph_basis(const float* __restrict__ bes, const int* __restrict__ gidx,
                                                        const float* __restrict__ theta,
                                                        const float* __restrict__ phi, int M, int nr,
                                                        const float* __restrict__ pref, int pair_mode,
                                                        float* __restrict__ out) {
  constexpr int H2 = NS * NS;
  __shared__ float sY[SPH_TPB][H2 + 1];
  __shared__ int sG[SPH_TPB];
  __shared__ float sPref[NS_MAX * NS_MAX];
  const bool zero_m = (phi == nullptr);
  const int H = zero_m ? NS : H2;
  for (int q = threadIdx.x; q < NS_MAX * NS_MAX; q += SPH_TPB) sPref[q] = pref[q];
  __syncthreads();
  const int m0 = blockIdx.x * SPH_TPB;
  const int m = m0 + threadIdx.x;
  if (m < M) {
    float Y[H2];
    real_sph_harm<NS>(theta[m], zero_m ? 0.f : phi[m], sPref, zero_m, Y);
    for (int h = 0; h < H; ++h) sY[threadIdx.x][h] = Y[h];
    sG[threadIdx.x] = gidx ? gidx[m] : m;
  }
  __syncthreads();
  const int rows = (M - m0 < SPH_TPB) ? M - m0 : SPH_TPB;
  const int K = H * nr;            // output row width
  const int KB = NS * nr;          // bessel row width
  const int64_t total = (int64_t)rows * K;
  for (int64_t q = threadIdx.x; q < total; q += SPH_TPB) {
    int row = (int)(q / K);
    int hn = (int)(q - (int64_t)row * K);
    int h = hn / nr, n = hn - h * nr;
    int order;
    if (zero_m) order = h;
    else if (pair_mode == 0) order = h % NS;
    else {  // degree of harmonic h: l = floor(sqrt(h))
      order = 0;
      while ((order + 1) * (order + 1) <= h) ++order;
    }
    out[(int64_t)m0 * K + q] = sY[row][h] * bes[(int64_t)sG[row] * KB + order * nr + n];
  }
}

// SchNet: gaussian smearing exp(coeff * (d - offset_k)^2)  (schnet.py:92-94)  and the cosine
// cutoff C = 0.5 * (cos(d * pi / cutoff) + 1)  (schnet.py:31).
__global__ void k_gauss_smear(const float* __restrict__ dist, int E, const float* __restrict__ offset, int G,
                              float coeff, float* __restrict__ out) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (int64_t)E * G) return;
  int e = (int)(q / G), k = (int)(q - (int64_t)e * G);
  float d = dist[e] - offset[k];
  out[q] = expf(coeff * (d * d));
}
__global__ void k_cos_cutoff(const float* __restrict__ dist, int E, float cutoff, float* __restrict__ out) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  out[e] = 0.5f * (cosf(dist[e] * DIG3D_PI_F / cutoff) + 1.0f);
}

extern "C" {

int dig3d_bessel_basis(const float* dist, int E, float cutoff, int ns, int nr, const double* zeros,
                       const double* norms, int envelope_p, float* out, void* stream) {
  DIG3D_ENTER();
  if (E <= 0) return DIG3D_OK;
  if (ns < 1 || ns > NS_MAX || nr < 1) return DIG3D_ERR_ARG;
  hipLaunchKernelGGL(k_bessel, dim3(dig3d_blocks((int64_t)E * ns * nr, 256)), dim3(256), 0, (hipStream_t)stream,
                     dist, E, cutoff, ns, nr, zeros, norms, envelope_p, out);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_sph_basis(const float* bes, const int* gidx, const float* theta, const float* phi, int M, int ns,
                    int nr, const float* pref, int pair_mode, float* out, void* stream) {
  DIG3D_ENTER();
  if (M <= 0) return DIG3D_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(dig3d_blocks(M, SPH_TPB)), block(SPH_TPB);
#define SPH_CASE(NS)                                                                                       \
  case NS:                                                                                                 \
    hipLaunchKernelGGL((k_sph_basis<NS>), grid, block, 0, st, bes, gidx, theta, phi, M, nr, pref, pair_mode, \
                       out);                                                                               \
    break;
  switch (ns) {
    SPH_CASE(1) SPH_CASE(2) SPH_CASE(3) SPH_CASE(4) SPH_CASE(5) SPH_CASE(6) SPH_CASE(7) SPH_CASE(8)
    default: return DIG3D_ERR_ARG;
  }
#undef SPH_CASE
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_gauss_smear(const float* dist, int E, const float* offset, int G, float coeff, float* out,
                      void* stream) {
  DIG3D_ENTER();
  if (E <= 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_gauss_smear, dim3(dig3d_blocks((int64_t)E * G, 256)), dim3(256), 0, (hipStream_t)stream,
                     dist, E, offset, G, coeff, out);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_cos_cutoff(const float* dist, int E, float cutoff, float* out, void* stream) {
  DIG3D_ENTER();
  if (E <= 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_cos_cutoff, dim3(dig3d_blocks(E, 256)), dim3(256), 0, (hipStream_t)stream, dist, E,
                     cutoff, out);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

}  // extern "C"
