// dig3d — first- and second-order derivative kernels of the geometry / basis pipeline.
//
// energy_and_force training (method/run.py:126-131) computes  force = -d out / d pos  with create_graph=True and then
// differentiates the loss THROUGH that gradient: everything between pos and the embeddings is differentiated twice
// (utils/geometric_computing.py:25-75 dist / angle / torsion; method/*/features.py dist_emb, Bessel x harmonics).
// The reference leaves that to autograd over ~10^2 elementwise ATen ops per quantity.  Here every pointwise function
// is ONE kernel per derivative order, generated from a single templated definition by forward-mode dual numbers
// (dual.h): order 1 = vector-Jacobian product, order 2 = (J w, g^T H w) for the incoming direction w.  Reductions
// from triplets onto edges are CSR segment sums (ascending, no atomics), as everywhere in the engine.
//
//   vec[e] = pos[i] - pos[j] for edge e = (j -> i)           (linear: HIP row gathers, ops.gather_rows)
//   dist[e] = |vec[e]|                                        k_vec_len / k_edge_combine (base term)
//   v1 = vec[ji[t]], v2 = -vec[kj[t]], v3 = -vec[targ[t]]     (pos_i - pos_j, pos_k - pos_j, pos_kn - pos_j)
//   angle[t] = atan2(|v1 x v2|, v1 . v2)                      k_tripgeom_d
//   torsion[t] = atan2(((v1 x v2) x (v1 x v3)) . v1 / |v1|, (v1 x v2) . (v1 x v3))   at the arg-min neighbour
//   bes[e, l, n], Y_h(theta[, phi]), rbf[e, n]                k_bessel_d, k_harm_d, k_distemb_*
#include "common.h"
#include "edge_values.h"
#include "dual.h"
#include "sph.h"

// ---------------------------------------------------------------------------------------------------------------
// pointwise functions, templated on the scalar type
// ---------------------------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ T fn_angle(const V3<T>& v1, const V3<T>& v2) {
  return dl_atan2(v3_len(v3_cross(v1, v2)), v3_dot(v1, v2));
}
template <class T>
__device__ __forceinline__ T fn_torsion(const V3<T>& v1, const V3<T>& v2, const V3<T>& v3) {
  V3<T> p1 = v3_cross(v1, v2), p2 = v3_cross(v1, v3);
  T a = v3_dot(p1, p2);
  T b = v3_dot(v3_cross(p1, p2), v1) / v3_len(v1);
  return dl_atan2(b, a);      // the reference's "+ 2 pi where <= 0" shift has zero derivative
}

// norm[l,n] * j_l(z[l,n] * d / cutoff) (* envelope(d / cutoff) when env_p > 0) — csrc/basis.hip:k_bessel
template <class T>
__device__ __forceinline__ T fn_bessel(const T& d, double cutoff, int l, double z, double norm, int env_p) {
  T x = d / cutoff;
  T u = z * x;
  T s = dl_sin(u), c = dl_cos(u);
  T jm = s / u;
  T j = jm;
  if (l >= 1) {
    j = s / (u * u) - c / u;
    for (int a = 1; a < l; ++a) {
      T jn = (double)(2 * a + 1) / u * j - jm;
      jm = j;
      j = jn;
    }
  }
  T v = norm * j;
  if (env_p > 0) {
    double p = (double)env_p;
    double ea = -(p + 1) * (p + 2) / 2, eb = p * (p + 2), ec = -p * (p + 1) / 2;
    T x0 = Lift<T>::of(1.0);
    for (int k = 0; k < env_p - 1; ++k) x0 = x0 * x;
    T x1 = x0 * x, x2 = x1 * x;
    v = v * (1.0 / x + ea * x0 + eb * x1 + ec * x2);
  }
  return v;
}

// Envelope(d/c) * sin(freq * d/c)  (spherenet/features.py:151-182), p = exponent + 1
template <class T>
__device__ __forceinline__ T fn_distemb(const T& d, const T& freq, double cutoff, int p_) {
  T x = d / cutoff;
  double p = (double)p_;
  double ea = -(p + 1) * (p + 2) / 2, eb = p * (p + 2), ec = -p * (p + 1) / 2;
  T x0 = Lift<T>::of(1.0);
  for (int k = 0; k < p_ - 1; ++k) x0 = x0 * x;
  T x1 = x0 * x, x2 = x1 * x;
  T env = 1.0 / x + ea * x0 + eb * x1 + ec * x2;
  return env * dl_sin(freq * x);
}

// real spherical harmonics in the order of sph.h:real_sph_harm, emitted one at a time (no [NS*NS] array of duals
// in registers): emit(h, Y_h).
template <class T, int NS, class Emit>
__device__ __forceinline__ void fn_harmonics(const T& theta, const T& phi, const float* __restrict__ pref,
                                             bool zero_m_only, Emit emit) {
  T ct = dl_cos(theta), st = dl_sin(theta);
  T x = st, y = st;
  if (!zero_m_only) {
    x = st * dl_cos(phi);
    y = st * dl_sin(phi);
  }
  T Cm = Lift<T>::of(1.0), Sm = Lift<T>::of(0.0);
  T Pmm = Lift<T>::of(1.0);
#pragma unroll
  for (int m = 0; m < NS; ++m) {
    if (m > 0) {
      Pmm = (double)(1 - 2 * m) * Pmm;
      T Sn = x * Sm + y * Cm;
      T Cn = x * Cm - y * Sm;
      Sm = Sn;
      Cm = Cn;
    }
    T P2 = Pmm, P1 = Pmm;
#pragma unroll
    for (int l = m; l < NS; ++l) {
      T P = Pmm;
      if (l == m + 1) P = (double)(2 * m + 1) * ct * Pmm;
      if (l > m + 1) P = ((double)(2 * l - 1) * ct * P1 - (double)(l + m - 1) * P2) / (double)(l - m);
      P2 = P1;
      P1 = P;
      T k = (double)pref[l * NS_MAX + m] * P;
      if (zero_m_only) {
        emit(l, k);
      } else if (m == 0) {
        emit(l * l, k);
      } else {
        emit(l * l + m, k * Cm);
        emit(l * l + 2 * l + 1 - m, k * Sm);
      }
    }
    if (zero_m_only) break;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// |vec| with the reference's float32 operation order (identical bits to geometry.hip:k_edge_dist on vec = pos_i-pos_j)
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_vec_len(const float* __restrict__ vec, int E, int mode, float* __restrict__ dist,
                          const int* __restrict__ cnt, float pad) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  if (cnt && e >= *cnt) {
    dist[e] = pad;
    return;
  }
  f3 v = load3(vec, e);
  dist[e] = mode == 0 ? ref_len(v) : ref_norm(v);
}

// key[t] = edge whose vector enters torsion[t] as v3 (arg-min reference neighbour); a dummy segment >= E when the
// triplet has none or when the arg-min is the triplet's own k (the value is then a float32 rounding residue of the
// reference's arithmetic, analytically constant: DESIGN.md §4).  The dummies are spread over TKEY_DUMMIES segments
// (E + t % TKEY_DUMMIES) so that no segment of the CSR built from the keys is long.
// val: CSR position -> edge id map (public API graphs).
#define TKEY_DUMMIES 1024
__global__ void k_targ_key(const int* __restrict__ targ, const int* __restrict__ kj, const int* __restrict__ val,
                           int T, int E, int* __restrict__ key, const int* __restrict__ cnt) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  int a = targ[t];
  const int dummy = E + (t & (TKEY_DUMMIES - 1));
  if ((cnt && t >= *cnt) || a < 0) {
    key[t] = dummy;
    return;
  }
  if (val) a = val[a];
  key[t] = (a == kj[t]) ? dummy : a;
}

// ---------------------------------------------------------------------------------------------------------------
// per-triplet derivative kernel.
//   ORD 1:  gv{1,2,3}[t] = g_angle[t] * d angle / d v{1,2,3} + g_tor[t] * d torsion / d v{1,2,3}
//   ORD 2:  w = (ggvec[ji], -ggvec[kj], -ggvec[key]);  o_ga[t] = grad(angle) . w,  o_gt[t] = grad(torsion) . w,
//           gv{1,2,3}[t] = (g_angle H_angle + g_tor H_torsion) w
// key[t] == E: no third vector (see k_targ_key).  Rows t >= *cnt (padding of a static-shape batch) write zeros.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ V3<double> ld3d(const float* __restrict__ p, int i, double sgn) {
  const float* q = p + 3ll * i;
  return {sgn * (double)q[0], sgn * (double)q[1], sgn * (double)q[2]};
}

template <int ORD, bool TOR>
__global__ void __launch_bounds__(256) k_tripgeom_d(const float* __restrict__ vec, const float* __restrict__ ggvec,
                                                     const int* __restrict__ ji, const int* __restrict__ kj,
                                                     const int* __restrict__ key, int T, int E,
                                                     const float* __restrict__ g_angle,
                                                     const float* __restrict__ g_tor, float* __restrict__ gv1,
                                                     float* __restrict__ gv2, float* __restrict__ gv3,
                                                     float* __restrict__ o_ga, float* __restrict__ o_gt,
                                                     const int* __restrict__ cnt) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const bool live = !(cnt && t >= *cnt);
  int e1 = 0, e2 = 0, e3 = E;
  if (live) {
    e1 = ji[t];
    e2 = kj[t];
    if (TOR) e3 = key[t];
  }
  const bool has3 = TOR && e3 < E;
  double o1[3] = {0, 0, 0}, o2[3] = {0, 0, 0}, o3[3] = {0, 0, 0};
  double ja = 0, jt = 0;
  if (live) {
    const double ga = g_angle ? (double)g_angle[t] : 0.0;
    const double gt = (has3 && g_tor) ? (double)g_tor[t] : 0.0;
    V3<double> a1 = ld3d(vec, e1, 1.0), a2 = ld3d(vec, e2, -1.0);
    V3<double> a3 = has3 ? ld3d(vec, e3, -1.0) : V3<double>{0, 0, 0};
    if (ORD == 1) {
      typedef D1<double> S;
      for (int i = 0; i < (TOR ? 9 : 6); ++i) {
        if (i >= 6 && !has3) break;
        auto mk = [&](const V3<double>& a, int base) {
          return V3<S>{S{a.x, i == base ? 1.0 : 0.0}, S{a.y, i == base + 1 ? 1.0 : 0.0}, S{a.z, i == base + 2 ? 1.0 : 0.0}};
        };
        V3<S> v1 = mk(a1, 0), v2 = mk(a2, 3), v3 = mk(a3, 6);
        double r = 0;
        if (i < 6) r += ga * fn_angle(v1, v2).d;
        if (has3) r += gt * fn_torsion(v1, v2, v3).d;
        if (i < 3) o1[i] = r; else if (i < 6) o2[i - 3] = r; else o3[i - 6] = r;
      }
    } else {
      typedef D1<double> S;
      typedef D1<S> Q;
      V3<double> w1 = ld3d(ggvec, e1, 1.0), w2 = ld3d(ggvec, e2, -1.0);
      V3<double> w3 = has3 ? ld3d(ggvec, e3, -1.0) : V3<double>{0, 0, 0};
      for (int i = 0; i < (TOR ? 9 : 6); ++i) {
        if (i >= 6 && !has3) break;
        auto mk = [&](const V3<double>& a, const V3<double>& w, int base) {
          return V3<Q>{Q{S{a.x, w.x}, S{i == base ? 1.0 : 0.0, 0.0}}, Q{S{a.y, w.y}, S{i == base + 1 ? 1.0 : 0.0, 0.0}},
                       Q{S{a.z, w.z}, S{i == base + 2 ? 1.0 : 0.0, 0.0}}};
        };
        V3<Q> v1 = mk(a1, w1, 0), v2 = mk(a2, w2, 3), v3 = mk(a3, w3, 6);
        double r = 0;
        if (i < 6) {
          Q an = fn_angle(v1, v2);
          r += ga * an.d.d;
          if (i == 0) ja = an.v.d;
        }
        if (has3) {
          Q to = fn_torsion(v1, v2, v3);
          r += gt * to.d.d;
          if (i == 0) jt = to.v.d;
        }
        if (i < 3) o1[i] = r; else if (i < 6) o2[i - 3] = r; else o3[i - 6] = r;
      }
    }
  }
  float* q = gv1 + 3ll * t;
  q[0] = (float)o1[0]; q[1] = (float)o1[1]; q[2] = (float)o1[2];
  q = gv2 + 3ll * t;
  q[0] = (float)o2[0]; q[1] = (float)o2[1]; q[2] = (float)o2[2];
  if (TOR) {
    q = gv3 + 3ll * t;
    q[0] = (float)o3[0]; q[1] = (float)o3[1]; q[2] = (float)o3[2];
  }
  if (ORD == 2) {
    if (o_ga) o_ga[t] = (float)ja;
    if (TOR && o_gt) o_gt[t] = (float)jt;
  }
}

// per-edge combine:  out[e] = base(e) + sum_{t in seg_ji(e)} gv1[t] - sum_{t: kj[t]=e} gv2[t] - sum_{t: key[t]=e} gv3[t]
//   ORD 1: base = g_dist[e] * u,                 u = vec[e] / |vec[e]|
//   ORD 2: base = g_dist[e] * (w - u (u.w)) / |vec[e]|,  w = ggvec[e];   o_gd[e] = u . w
// any of the triplet arrays may be null (SchNet: distances only).
// EC_LPE = 16 lanes per edge: the edge's triplets (in each of the three groupings) dealt to the lanes, float64 sums joined by a
// butterfly — a thread per edge walked ~14 + ~14 rows through a chain of dependent loads (17 us at 9.4k edges).
#define EC_LPE 16
template <int ORD>
__global__ void __launch_bounds__(256) k_edge_combine(const float* __restrict__ vec, const float* __restrict__ ggvec,
                                                       const float* __restrict__ g_dist, int E,
                                                       const int* __restrict__ tptr, const float* __restrict__ gv1,
                                                       const int* __restrict__ kptr2, const int* __restrict__ perm2,
                                                       const float* __restrict__ gv2, const int* __restrict__ kptr3,
                                                       const int* __restrict__ perm3, const float* __restrict__ gv3,
                                                       float* __restrict__ out, float* __restrict__ o_gd,
                                                       const int* __restrict__ cnt) {
  const int e = (blockIdx.x * blockDim.x + threadIdx.x) / EC_LPE, sub = threadIdx.x & (EC_LPE - 1);
  const bool inb = e < E;
  const bool live = inb && !(cnt && e >= *cnt);
  double ax = 0, ay = 0, az = 0;
  if (live) {
    if (gv1 && tptr)
      for (int t = tptr[e] + sub, t1 = tptr[e + 1]; t < t1; t += EC_LPE) {
        const float* q = gv1 + 3ll * t;
        ax += q[0]; ay += q[1]; az += q[2];
      }
    if (gv2 && kptr2)
      for (int p = kptr2[e] + sub, p1 = kptr2[e + 1]; p < p1; p += EC_LPE) {
        const float* q = gv2 + 3ll * perm2[p];
        ax -= q[0]; ay -= q[1]; az -= q[2];
      }
    if (gv3 && kptr3)
      for (int p = kptr3[e] + sub, p1 = kptr3[e + 1]; p < p1; p += EC_LPE) {
        const float* q = gv3 + 3ll * perm3[p];
        ax -= q[0]; ay -= q[1]; az -= q[2];
      }
  }
#pragma unroll
  for (int off = EC_LPE / 2; off > 0; off >>= 1) {
    ax += __shfl_xor(ax, off, 64);
    ay += __shfl_xor(ay, off, 64);
    az += __shfl_xor(az, off, 64);
  }
  if (!inb || sub != 0) return;
  float* o = out + 3ll * e;
  if (!live) {
    o[0] = o[1] = o[2] = 0.f;
    if (ORD == 2 && o_gd) o_gd[e] = 0.f;
    return;
  }
  const float* v = vec + 3ll * e;
  const double vx = v[0], vy = v[1], vz = v[2];
  const double d = sqrt(vx * vx + vy * vy + vz * vz);
  const double ux = vx / d, uy = vy / d, uz = vz / d;
  const double gd = g_dist ? (double)g_dist[e] : 0.0;
  if (ORD == 1) {
    ax += gd * ux; ay += gd * uy; az += gd * uz;
  } else {
    const float* w = ggvec + 3ll * e;
    const double uw = ux * w[0] + uy * w[1] + uz * w[2];
    ax += gd * (w[0] - ux * uw) / d;
    ay += gd * (w[1] - uy * uw) / d;
    az += gd * (w[2] - uz * uw) / d;
    if (o_gd) o_gd[e] = (float)uw;
  }
  o[0] = (float)ax; o[1] = (float)ay; o[2] = (float)az;
}

// ---------------------------------------------------------------------------------------------------------------
// Bessel basis derivatives, a WAVE per edge, a lane per function k < K = ns*nr (float64 dual numbers: one sin / cos pair and
// the l-step recurrence per function — with a thread per edge walking all 42 functions the launch was 105 half-empty
// workgroups of 42-deep serial float64 chains: 53 us (first order) and 94 us (second order) at 13.4k edges).
//   ORD 1:  o_d[e] = sum_k g[e,k] f_k'(d)
//   ORD 2:  o_g[e,k] = gg_d[e] f_k'(d),   o_d[e] = gg_d[e] * sum_k g[e,k] f_k''(d)
// The sum over k is a float64 butterfly over the wave (rounded to float32 once).
// ---------------------------------------------------------------------------------------------------------------
// (r06, second form: a lane per (edge, k) ITEM of a block of floor(256 / K) edges — with a wave per edge 42 of the 64 lanes
// worked; the per-edge sums are taken from LDS in k order by one thread per edge.)
#define BD_TERMS 1024            // K = ns * nr <= 1024
template <int ORD>
__global__ void __launch_bounds__(256) k_bessel_d(const float* __restrict__ dist, int E, float cutoff, int ns, int nr,
                                                   const double* __restrict__ zeros, const double* __restrict__ norms,
                                                   int env_p, const float* __restrict__ g,
                                                   const float* __restrict__ gg_d, float* __restrict__ o_d,
                                                   float* __restrict__ o_g, const int* __restrict__ cnt, int epb) {
  __shared__ double term[BD_TERMS];
  const int K = ns * nr;
  const int e0 = blockIdx.x * epb;
  const int live_n = cnt ? *cnt : E;
  for (int it = threadIdx.x; it < epb * K; it += 256) {
    const int el = it / K, k = it - el * K, e = e0 + el;
    double t = 0;
    if (e < E) {
      if (e < live_n) {
        const double d = (double)dist[e];
        if (ORD == 1) {
          D1<double> x{d, 1.0};
          t = (double)g[(int64_t)e * K + k] * fn_bessel(x, (double)cutoff, k / nr, zeros[k], norms[k], env_p).d;
        } else {
          D1<D1<double>> x{{d, 1.0}, {1.0, 0.0}};
          D1<D1<double>> f = fn_bessel(x, (double)cutoff, k / nr, zeros[k], norms[k], env_p);
          o_g[(int64_t)e * K + k] = (float)((double)gg_d[e] * f.v.d);
          t = (double)g[(int64_t)e * K + k] * f.d.d;
        }
      } else if (ORD == 2) {
        o_g[(int64_t)e * K + k] = 0.f;
      }
    }
    term[it] = t;
  }
  __syncthreads();
  if ((int)threadIdx.x < epb) {
    const int e = e0 + threadIdx.x;
    if (e < E) {
      double acc = 0;
      for (int k = 0; k < K; ++k) acc += term[threadIdx.x * K + k];
      o_d[e] = (float)(ORD == 2 && e < live_n ? (double)gg_d[e] * acc : acc);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// dist_emb:  rbf[e,n] = Envelope(d/c) sin(freq_n d/c)   (float32 forward, freq learnable)
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_distemb_fwd(const float* __restrict__ dist, const float* __restrict__ freq, int E, int nr,
                              float cutoff, int p_, float* __restrict__ out, const int* __restrict__ cnt) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (int64_t)E * nr) return;
  int e = (int)(q / nr), n = (int)(q - (int64_t)e * nr);
  if (cnt && e >= *cnt) {
    out[q] = 0.f;
    return;
  }
  out[q] = distemb_value(dist[e], freq[n], cutoff, p_);   // the reference's float32 formula (features.py:158-164,181-182)
}

// The three per-edge launches of the model front as ONE (energy route; 3 x ~4.7 us of launch floor at 8 000 edges):
// dist[e], rbf[e, 0..nrd) = dist_emb, bes[e, 0..K) = Bessel table.  One thread per (edge, column) of the nrd + K columns;
// every thread forms the edge's distance itself (two L2-resident position rows) with the arithmetic of k_edge_dist, so
// the three outputs are bit-identical to those of the per-stage kernels.
__global__ void __launch_bounds__(256) k_edge_front(const float* __restrict__ pos, const int* __restrict__ src,
                                                     const int* __restrict__ dst, int E, int mode,
                                                     const int* __restrict__ cnt, float pad, float* __restrict__ dist,
                                                     const float* __restrict__ freq, int nrd, float cutoff_d, int p_,
                                                     float* __restrict__ rbf, float cutoff_b, int ns, int nr,
                                                     const double* __restrict__ zeros, const double* __restrict__ norms,
                                                     int env_p, float* __restrict__ bes) {
  const int K = ns * nr, W = nrd + K;
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= (int64_t)E * W) return;
  const int e = (int)(q / W), c = (int)(q - (int64_t)e * W);
  const bool padded = cnt && e >= *cnt;
  const float d = padded ? pad : edge_dist_value(pos, src[e], dst[e], mode);
  if (c == 0) dist[e] = d;
  if (c < nrd) {
    rbf[(int64_t)e * nrd + c] = padded ? 0.f : distemb_value(d, freq[c], cutoff_d, p_);
  } else {
    const int ln = c - nrd;
    bes[(int64_t)e * K + ln] = bessel_value(d, cutoff_b, ln, ln / nr, zeros, norms, env_p);
  }
}

#define DE_TPB 256
#define DE_NRMAX 16
// ORD 1:  o_d[e] = sum_n g[e,n] f_d;   part[block, n] = sum_{e in block} g[e,n] f_freq
// ORD 2:  w = (gg_d[e], gg_f[n]):  o_g[e,n] = f_d w_d + f_f w_f;  o_d[e] = sum_n g (f_dd w_d + f_df w_f);
//         part[block, n] = sum_e g (f_fd w_d + f_ff w_f)
// A lane per (edge, n): DE_LPE = 16 lanes per edge (n >= nr idle), 16 edges per block — with a thread per edge walking all nr
// functions (two float64 dual evaluations each) the launch was 34 workgroups of 12-deep serial float64 chains: 12.3 us at
// 8.7k edges (first order), 20 us (second).  The sums over n (per edge) and over the block's edges (per n) are float64
// butterflies; one partial row per block.
#define DE_LPE 16
template <int ORD>
__global__ void __launch_bounds__(DE_TPB) k_distemb_d(const float* __restrict__ dist, const float* __restrict__ freq,
                                                       int E, int nr, float cutoff, int p_,
                                                       const float* __restrict__ g, const float* __restrict__ gg_d,
                                                       const float* __restrict__ gg_f, float* __restrict__ o_d,
                                                       float* __restrict__ o_g, float* __restrict__ part,
                                                       const int* __restrict__ cnt) {
  __shared__ double red[DE_TPB / 64][DE_LPE];
  const int n = threadIdx.x & (DE_LPE - 1);
  const int e = blockIdx.x * (DE_TPB / DE_LPE) + (threadIdx.x / DE_LPE);
  const bool live = e < E && !(cnt && e >= *cnt);
  double acc = 0, pf = 0;
  if (live && n < nr) {
    const double d = (double)dist[e];
    const double gn = (double)g[(int64_t)e * nr + n];
    const double f = (double)freq[n];
    if (ORD == 1) {
      typedef D1<double> S;
      acc = gn * fn_distemb(S{d, 1.0}, S{f, 0.0}, (double)cutoff, p_).d;
      pf = gn * fn_distemb(S{d, 0.0}, S{f, 1.0}, (double)cutoff, p_).d;
    } else {
      typedef D1<double> S;
      typedef D1<S> Q;
      const double wd = gg_d ? (double)gg_d[e] : 0.0, wf = gg_f ? (double)gg_f[n] : 0.0;
      Q a = fn_distemb(Q{S{d, wd}, S{1.0, 0.0}}, Q{S{f, wf}, S{0.0, 0.0}}, (double)cutoff, p_);
      Q b = fn_distemb(Q{S{d, wd}, S{0.0, 0.0}}, Q{S{f, wf}, S{1.0, 0.0}}, (double)cutoff, p_);
      o_g[(int64_t)e * nr + n] = (float)a.v.d;       // J . w
      acc = gn * a.d.d;
      pf = gn * b.d.d;
    }
  } else if (ORD == 2 && e < E && n < nr) {
    o_g[(int64_t)e * nr + n] = 0.f;
  }
  // o_d[e]: the sum over the edge's DE_LPE lanes
#pragma unroll
  for (int off = DE_LPE / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (n == 0 && e < E) o_d[e] = (float)acc;
  // part[block, n]: the sum over the block's edges — lanes with the same n inside the wave, then the waves through LDS
#pragma unroll
  for (int off = DE_LPE; off < 64; off <<= 1) pf += __shfl_xor(pf, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < DE_LPE) red[wave][lane] = pf;
  __syncthreads();
  if (threadIdx.x < nr) {
    double v = 0;
    for (int w = 0; w < DE_TPB / 64; ++w) v += red[w][threadIdx.x];
    part[(int64_t)blockIdx.x * nr + threadIdx.x] = (float)v;
  }
}

// out[c] = sum_b part[b, c]: one wave per column (blockIdx.x = c), lanes stride the nb partial rows, float64 butterfly — a fixed
// order, so deterministic (a thread per column walking all rows was a serial chain of nb loads: fine at 34 rows, not at 544)
__global__ void __launch_bounds__(64) k_colsum_small(const float* __restrict__ part, int nb, int n, float* __restrict__ out) {
  const int c = blockIdx.x, lane = threadIdx.x;
  double v = 0;
  for (int r = lane; r < nb; r += 64) v += (double)part[(int64_t)r * n + c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if (lane == 0) out[c] = (float)v;
}

// ---------------------------------------------------------------------------------------------------------------
// spherical-harmonic derivatives, one thread per row (theta[m], phi[m]);  H = NS (phi == null) or NS*NS columns.
//   ORD 1:  o_th[m] = sum_h g[m,h] dY_h/dtheta,  o_ph[m] = sum_h g[m,h] dY_h/dphi
//   ORD 2:  w = (gg_th[m], gg_ph[m]):  o_g[m,h] = grad(Y_h) . w;   (o_th, o_ph)[m] = sum_h g[m,h] (Hess(Y_h) w)
// ---------------------------------------------------------------------------------------------------------------
template <int NS, int ORD>
__global__ void __launch_bounds__(128) k_harm_d(const float* __restrict__ theta, const float* __restrict__ phi, int M,
                                                 const float* __restrict__ pref, const float* __restrict__ g,
                                                 const float* __restrict__ gg_th, const float* __restrict__ gg_ph,
                                                 float* __restrict__ o_th, float* __restrict__ o_ph,
                                                 float* __restrict__ o_g, const int* __restrict__ cnt) {
  __shared__ float sPref[NS_MAX * NS_MAX];
  for (int q = threadIdx.x; q < NS_MAX * NS_MAX; q += blockDim.x) sPref[q] = pref[q];
  __syncthreads();
  int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const bool zero_m = (phi == nullptr);
  const int H = zero_m ? NS : NS * NS;
  const float* gr = g + (int64_t)m * H;
  if (cnt && m >= *cnt) {
    o_th[m] = 0.f;
    if (!zero_m) o_ph[m] = 0.f;
    if (ORD == 2)
      for (int h = 0; h < H; ++h) o_g[(int64_t)m * H + h] = 0.f;
    return;
  }
  const double th = (double)theta[m], ph = zero_m ? 0.0 : (double)phi[m];
  if (ORD == 1) {
    typedef D1<double> S;
    double a = 0, b = 0;
    fn_harmonics<S, NS>(S{th, 1.0}, S{ph, 0.0}, sPref, zero_m, [&](int h, const S& y) { a += (double)gr[h] * y.d; });
    if (!zero_m)
      fn_harmonics<S, NS>(S{th, 0.0}, S{ph, 1.0}, sPref, zero_m, [&](int h, const S& y) { b += (double)gr[h] * y.d; });
    o_th[m] = (float)a;
    if (!zero_m) o_ph[m] = (float)b;
  } else {
    typedef D1<double> S;
    typedef D1<S> Q;
    const double wt = gg_th ? (double)gg_th[m] : 0.0, wp = (!zero_m && gg_ph) ? (double)gg_ph[m] : 0.0;
    double a = 0, b = 0;
    float* og = o_g + (int64_t)m * H;
    fn_harmonics<Q, NS>(Q{S{th, wt}, S{1.0, 0.0}}, Q{S{ph, wp}, S{0.0, 0.0}}, sPref, zero_m, [&](int h, const Q& y) {
      og[h] = (float)y.v.d;
      a += (double)gr[h] * y.d.d;
    });
    if (!zero_m)
      fn_harmonics<Q, NS>(Q{S{th, wt}, S{0.0, 0.0}}, Q{S{ph, wp}, S{1.0, 0.0}}, sPref, zero_m,
                          [&](int h, const Q& y) { b += (double)gr[h] * y.d.d; });
    o_th[m] = (float)a;
    if (!zero_m) o_ph[m] = (float)b;
  }
}

// Y[m, h] forward table (float32 recurrences of sph.h, the values the fused kernels use), rows >= *cnt zero.
template <int NS>
__global__ void __launch_bounds__(128) k_harm_fwd(const float* __restrict__ theta, const float* __restrict__ phi, int M,
                                                   const float* __restrict__ pref, float* __restrict__ out,
                                                   const int* __restrict__ cnt) {
  __shared__ float sPref[NS_MAX * NS_MAX];
  for (int q = threadIdx.x; q < NS_MAX * NS_MAX; q += blockDim.x) sPref[q] = pref[q];
  __syncthreads();
  int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const bool zero_m = (phi == nullptr);
  const int H = zero_m ? NS : NS * NS;
  float Y[NS * NS];
  const bool live = !(cnt && m >= *cnt);
  if (live) real_sph_harm<NS>(theta[m], zero_m ? 0.f : phi[m], sPref, zero_m, Y);
  for (int h = 0; h < H; ++h) out[(int64_t)m * H + h] = live ? Y[h] : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------
// activation pieces of the twice-differentiable dense layer (ops.py:_DgradAct):
//   k_act_bwd2:     o_gy = t * act'(z),  o_z = t * gy * act''(z)            (t = ggx W^T)
//   k_preact_merge: out  = gy * act'(z) + gz                                 (gz: gradient that reached z directly)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void act_d12(float z, int act, float& d1, float& d2) {
  if (act == 1) {              // swish
    const float s = 1.0f / (1.0f + expf(-z));
    d1 = s * (1.0f + z * (1.0f - s));
    d2 = s * (1.0f - s) * (2.0f + z * (1.0f - 2.0f * s));
  } else if (act == 2) {       // shifted softplus
    const float s = 1.0f / (1.0f + expf(-z));
    d1 = s;
    d2 = s * (1.0f - s);
  } else {
    d1 = 1.0f;
    d2 = 0.0f;
  }
}
__global__ void k_act_bwd2(const float* __restrict__ t, const float* __restrict__ gy, const float* __restrict__ z,
                           int64_t n, int act, float* __restrict__ o_gy, float* __restrict__ o_z) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  float d1, d2;
  act_d12(z[q], act, d1, d2);
  const float tv = t[q];
  o_gy[q] = tv * d1;
  o_z[q] = tv * gy[q] * d2;
}
__global__ void k_preact_merge(const float* __restrict__ gy, const float* __restrict__ z, const float* __restrict__ gz,
                               int64_t n, int act, float* __restrict__ out) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  float d1, d2;
  act_d12(z[q], act, d1, d2);
  out[q] = (gy ? gy[q] * d1 : 0.f) + (gz ? gz[q] : 0.f);
}

// ================================================================================================================
// C ABI
// ================================================================================================================
// ------------------------------------------------------------------------------------------------
// y = a * b, twice differentiable (r04): the elementwise products of the energy_and_force route — x_kj * (radial
// projection), e2 = lin_rbf(rbf) * e1 (spherenet.py:90,155,182; dimenetpp.py:77,137,160) — as three kernels per product
// and step (forward / backward / double backward) instead of the ~9 framework multiplies and additions the autograd of
// `a * b` issues under create_graph.  NULL incoming gradients are zeros.
// ------------------------------------------------------------------------------------------------
__global__ void k_ew_mul(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) y[q] = a[q] * b[q];
}
// out = ((in_0 + in_1) + in_2) + ... : the gradients that reach ONE tensor from its n consumers, summed by one launch in a
// fixed order instead of n - 1 framework additions (dig_amd/diffops.py:fan_out: rbf [E, 6] has 2 + 2 L consumers in each of
// the two backward passes of an energy_and_force step)
#define SUM_MANY_MAX 16
struct SumManyDesc {
  const float* in[SUM_MANY_MAX];
  int n;
};
__global__ void k_sum_many(SumManyDesc d, float* __restrict__ out, int64_t numel) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= numel) return;
  float s = d.in[0][q];
#pragma unroll 4
  for (int i = 1; i < d.n; ++i) s += d.in[i][q];
  out[q] = s;
}
// ga = g * b (+ adda), gb = g * a (+ addb).  adda / addb (optional): the gradients that reached a and b through the OTHER graph
// of an energy_and_force step (the product's own double backward), added here instead of by a framework addition each
__global__ void k_ew_mul_bwd(const float* __restrict__ g, const float* __restrict__ a, const float* __restrict__ b,
                             float* __restrict__ ga, float* __restrict__ gb, int64_t n, const float* __restrict__ adda,
                             const float* __restrict__ addb) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const float gv = g ? g[q] : 0.f;
  float va = gv * b[q], vb = gv * a[q];
  if (adda) va += adda[q];
  if (addb) vb += addb[q];
  ga[q] = va;
  gb[q] = vb;
}
// backward of (ga, gb) = (g b, g a) w.r.t. (g, a, b) for incoming (gga, ggb):  og = gga b + ggb a,  oa = ggb g,  ob = gga g
__global__ void k_ew_mul_bwd2(const float* __restrict__ gga, const float* __restrict__ ggb, const float* __restrict__ g,
                              const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ og,
                              float* __restrict__ oa, float* __restrict__ ob, int64_t n) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const float x = gga ? gga[q] : 0.f, y = ggb ? ggb[q] : 0.f, gv = g[q];
  og[q] = x * b[q] + y * a[q];
  oa[q] = y * gv;
  ob[q] = x * gv;
}

extern "C" {

int dig3d_vec_len(const float* vec, int E, int mode, float* dist, const int* cnt, float pad, void* stream) {
  DIG3D_ENTER();
  if (E <= 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_vec_len, dim3(dig3d_blocks(E, 256)), dim3(256), 0, (hipStream_t)stream, vec, E, mode, dist, cnt,
                     pad);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_torsion_key_segments(int E) { return E + TKEY_DUMMIES; }

int dig3d_torsion_key(const int* targ, const int* kj, const int* val, int T, int E, int* key, const int* cnt,
                      void* stream) {
  DIG3D_ENTER();
  if (T <= 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_targ_key, dim3(dig3d_blocks(T, 256)), dim3(256), 0, (hipStream_t)stream, targ, kj, val, T, E, key,
                     cnt);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// order 1 (ggvec == NULL) or order 2 per-triplet derivative pass; key == NULL: no torsion.
int dig3d_tripgeom_grad(const float* vec, const float* ggvec, const int* ji, const int* kj, const int* key, int T,
                        int E, const float* g_angle, const float* g_tor, float* gv1, float* gv2, float* gv3,
                        float* o_ga, float* o_gt, const int* cnt, void* stream) {
  DIG3D_ENTER();
  if (T <= 0) return DIG3D_OK;
  if (!vec || !ji || !kj || !gv1 || !gv2 || (key && !gv3)) return DIG3D_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(dig3d_blocks(T, 256)), block(256);
#define TG(ORD, TOR)                                                                                              \
  hipLaunchKernelGGL((k_tripgeom_d<ORD, TOR>), grid, block, 0, st, vec, ggvec, ji, kj, key, T, E, g_angle, g_tor, \
                     gv1, gv2, gv3, o_ga, o_gt, cnt)
  if (!ggvec) {
    if (key) TG(1, true); else TG(1, false);
  } else {
    if (key) TG(2, true); else TG(2, false);
  }
#undef TG
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_edge_combine(const float* vec, const float* ggvec, const float* g_dist, int E, const int* tptr,
                       const float* gv1, const int* kptr2, const int* perm2, const float* gv2, const int* kptr3,
                       const int* perm3, const float* gv3, float* out, float* o_gd, const int* cnt, void* stream) {
  DIG3D_ENTER();
  if (E <= 0) return DIG3D_OK;
  if (!vec || !out) return DIG3D_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(dig3d_blocks((int64_t)E * EC_LPE, 256)), block(256);
  if (!ggvec)
    hipLaunchKernelGGL((k_edge_combine<1>), grid, block, 0, st, vec, ggvec, g_dist, E, tptr, gv1, kptr2, perm2, gv2, kptr3,
                       perm3, gv3, out, o_gd, cnt);
  else
    hipLaunchKernelGGL((k_edge_combine<2>), grid, block, 0, st, vec, ggvec, g_dist, E, tptr, gv1, kptr2, perm2, gv2, kptr3,
                       perm3, gv3, out, o_gd, cnt);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_bessel_grad(const float* dist, int E, float cutoff, int ns, int nr, const double* zeros, const double* norms,
                      int envelope_p, const float* g, const float* gg_d, float* o_d, float* o_g, const int* cnt,
                      void* stream) {
  DIG3D_ENTER();
  if (E <= 0) return DIG3D_OK;
  if (ns < 1 || ns > NS_MAX || nr < 1 || !g || !o_d || (gg_d && !o_g)) return DIG3D_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int K = ns * nr;
  if (K > BD_TERMS) return DIG3D_ERR_ARG;
  const int epb = K >= 256 ? 1 : 256 / K;             // edges per block: floor(256 / K) whole edges of K items
  dim3 grid(dig3d_blocks(E, epb)), block(256);
  if (!gg_d)
    hipLaunchKernelGGL((k_bessel_d<1>), grid, block, 0, st, dist, E, cutoff, ns, nr, zeros, norms, envelope_p, g, gg_d, o_d,
                       o_g, cnt, epb);
  else
    hipLaunchKernelGGL((k_bessel_d<2>), grid, block, 0, st, dist, E, cutoff, ns, nr, zeros, norms, envelope_p, g, gg_d, o_d,
                       o_g, cnt, epb);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_distemb_fwd(const float* dist, const float* freq, int E, int nr, float cutoff, int p, float* out,
                      const int* cnt, void* stream) {
  DIG3D_ENTER();
  if (E <= 0) return DIG3D_OK;
  if (nr < 1 || p < 1) return DIG3D_ERR_ARG;
  hipLaunchKernelGGL(k_distemb_fwd, dim3(dig3d_blocks((int64_t)E * nr, 256)), dim3(256), 0, (hipStream_t)stream, dist,
                     freq, E, nr, cutoff, p, out, cnt);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_edge_front(const float* pos, const int* src, const int* dst, int E, int mode, const int* cnt, float pad,
                     float* dist, const float* freq, int nrd, float cutoff_d, int p, float* rbf, float cutoff_b, int ns,
                     int nr, const double* zeros, const double* norms, int envelope_p, float* bes, void* stream) {
  DIG3D_ENTER();
  if (E <= 0) return DIG3D_OK;
  if (!pos || !src || !dst || !dist || !freq || !rbf || !zeros || !norms || !bes || nrd < 1 || p < 1 || ns < 1 ||
      ns > NS_MAX || nr < 1)
    return DIG3D_ERR_ARG;
  hipLaunchKernelGGL(k_edge_front, dim3(dig3d_blocks((int64_t)E * (nrd + ns * nr), 256)), dim3(256), 0, (hipStream_t)stream,
                     pos, src, dst, E, mode, cnt, pad, dist, freq, nrd, cutoff_d, p, rbf, cutoff_b, ns, nr, zeros, norms,
                     envelope_p, bes);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_distemb_blocks(int E) { return E <= 0 ? 1 : (E + DE_TPB / DE_LPE - 1) / (DE_TPB / DE_LPE); }

// order 1 (gg_d == gg_f == NULL, o_g unused) or order 2.  part: float[dig3d_distemb_blocks(E) * nr]; o_f[nr] (written when
// reduce_now, or E <= 0).
int dig3d_distemb_grad(const float* dist, const float* freq, int E, int nr, float cutoff, int p, const float* g,
                       const float* gg_d, const float* gg_f, int order, float* o_d, float* o_g, float* part,
                       float* o_f, const int* cnt, int reduce_now, void* stream) {
  DIG3D_ENTER();
  hipStream_t st = (hipStream_t)stream;
  if (nr < 1 || nr > DE_NRMAX || p < 1 || !o_f || (order != 1 && order != 2)) return DIG3D_ERR_ARG;
  if (E <= 0) {
    if (dig3d_zero_async(o_f, sizeof(float) * nr, st) != hipSuccess) return DIG3D_ERR_LAUNCH;
    return DIG3D_OK;
  }
  if (!g || !o_d || !part || (order == 2 && !o_g)) return DIG3D_ERR_ARG;
  const int nb = dig3d_distemb_blocks(E);
  if (order == 1)
    hipLaunchKernelGGL((k_distemb_d<1>), dim3(nb), dim3(DE_TPB), 0, st, dist, freq, E, nr, cutoff, p, g, gg_d, gg_f, o_d,
                       o_g, part, cnt);
  else
    hipLaunchKernelGGL((k_distemb_d<2>), dim3(nb), dim3(DE_TPB), 0, st, dist, freq, E, nr, cutoff, p, g, gg_d, gg_f, o_d,
                       o_g, part, cnt);
  DIG3D_CHECK_LAUNCH();
  if (reduce_now) {           // else the caller sums the nb partial rows (dig3d_reduce_many, stride nr)
    hipLaunchKernelGGL(k_colsum_small, dim3(nr), dim3(64), 0, st, part, nb, nr, o_f);
    DIG3D_CHECK_LAUNCH();
  }
  return DIG3D_OK;
}

int dig3d_harmonics_fwd(const float* theta, const float* phi, int M, int ns, const float* pref, float* out,
                        const int* cnt, void* stream) {
  DIG3D_ENTER();
  if (M <= 0) return DIG3D_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(dig3d_blocks(M, 128)), block(128);
#define HF(NS)                                                                            \
  case NS:                                                                                \
    hipLaunchKernelGGL((k_harm_fwd<NS>), grid, block, 0, st, theta, phi, M, pref, out, cnt); \
    break;
  switch (ns) {
    HF(1) HF(2) HF(3) HF(4) HF(5) HF(6) HF(7) HF(8)
    default: return DIG3D_ERR_ARG;
  }
#undef HF
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// order 1 (gg_th == gg_ph == NULL) or order 2 (order argument decides; a NULL direction counts as zero).
int dig3d_harmonics_grad(const float* theta, const float* phi, int M, int ns, const float* pref, const float* g,
                         const float* gg_th, const float* gg_ph, int order, float* o_th, float* o_ph, float* o_g,
                         const int* cnt, void* stream) {
  DIG3D_ENTER();
  if (M <= 0) return DIG3D_OK;
  if (!g || !o_th || (phi && !o_ph) || (order == 2 && !o_g) || (order != 1 && order != 2)) return DIG3D_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(dig3d_blocks(M, 128)), block(128);
#define HG(NS)                                                                                                   \
  case NS:                                                                                                       \
    if (order == 1)                                                                                              \
      hipLaunchKernelGGL((k_harm_d<NS, 1>), grid, block, 0, st, theta, phi, M, pref, g, gg_th, gg_ph, o_th, o_ph, \
                         o_g, cnt);                                                                              \
    else                                                                                                         \
      hipLaunchKernelGGL((k_harm_d<NS, 2>), grid, block, 0, st, theta, phi, M, pref, g, gg_th, gg_ph, o_th, o_ph, \
                         o_g, cnt);                                                                              \
    break;
  switch (ns) {
    HG(1) HG(2) HG(3) HG(4) HG(5) HG(6) HG(7) HG(8)
    default: return DIG3D_ERR_ARG;
  }
#undef HG
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_act_bwd2(const float* t, const float* gy, const float* z, int64_t n, int act, float* o_gy, float* o_z,
                   void* stream) {
  DIG3D_ENTER();
  if (n <= 0) return DIG3D_OK;
  if (!t || !gy || !z || !o_gy || !o_z) return DIG3D_ERR_ARG;
  hipLaunchKernelGGL(k_act_bwd2, dim3(dig3d_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, t, gy, z, n, act, o_gy,
                     o_z);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_preact_merge(const float* gy, const float* z, const float* gz, int64_t n, int act, float* out,
                       void* stream) {
  DIG3D_ENTER();
  if (n <= 0) return DIG3D_OK;
  if (!z || !out || (!gy && !gz)) return DIG3D_ERR_ARG;
  hipLaunchKernelGGL(k_preact_merge, dim3(dig3d_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, gy, z, gz, n, act,
                     out);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_sum_many(const void* const* in, int n, int64_t numel, float* out, void* stream) {
  DIG3D_ENTER();
  if (numel <= 0) return DIG3D_OK;
  if (!in || !out || n < 1 || n > SUM_MANY_MAX) return DIG3D_ERR_ARG;
  SumManyDesc d;
  d.n = n;
  for (int i = 0; i < SUM_MANY_MAX; ++i) {
    d.in[i] = i < n ? (const float*)in[i] : nullptr;
    if (i < n && !d.in[i]) return DIG3D_ERR_ARG;
  }
  hipLaunchKernelGGL(k_sum_many, dim3(dig3d_blocks(numel, 256)), dim3(256), 0, (hipStream_t)stream, d, out, numel);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_ew_mul(const float* a, const float* b, float* y, int64_t n, void* stream) {
  DIG3D_ENTER();
  if (n <= 0) return DIG3D_OK;
  if (!a || !b || !y) return DIG3D_ERR_ARG;
  hipLaunchKernelGGL(k_ew_mul, dim3(dig3d_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, y, n);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_ew_mul_bwd(const float* g, const float* a, const float* b, float* ga, float* gb, int64_t n, const float* adda,
                     const float* addb, void* stream) {
  DIG3D_ENTER();
  if (n <= 0) return DIG3D_OK;
  if (!a || !b || !ga || !gb) return DIG3D_ERR_ARG;
  hipLaunchKernelGGL(k_ew_mul_bwd, dim3(dig3d_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, g, a, b, ga, gb, n, adda,
                     addb);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_ew_mul_bwd2(const float* gga, const float* ggb, const float* g, const float* a, const float* b, float* og,
                      float* oa, float* ob, int64_t n, void* stream) {
  DIG3D_ENTER();
  if (n <= 0) return DIG3D_OK;
  if (!g || !a || !b || !og || !oa || !ob) return DIG3D_ERR_ARG;
  hipLaunchKernelGGL(k_ew_mul_bwd2, dim3(dig3d_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, gga, ggb, g, a, b, og,
                     oa, ob, n);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

}  // extern "C"
