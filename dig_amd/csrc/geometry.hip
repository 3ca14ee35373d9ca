// dig3d geometry: xyz -> (dist, angle, torsion) and ComENet's (dist, theta, phi, tau).
// Float32 with the reference's exact IEEE operation order (common.h "reference float" helpers):
//   utils/geometric_computing.py:25,44-48,65-75      (dist, angle, torsion = min over quadruplets)
//   method/schnet/schnet.py:158                       (dist = norm)
//   method/comenet/comenet.py:297-385                 (nearest / 2nd-nearest references, theta, phi, tau)
// The quadruplet list (Q ~ T*deg entries, five int64 + six float[Q,3] temporaries in the reference) is
// never materialised: each triplet walks the CSR row of j in registers and keeps a running min + argmin.
#include "common.h"

// mode 0: sqrt(sum((pos[i]-pos[j])^2))  (geometric_computing.py:25)
// mode 1: (pos[j]-pos[i]).norm()        (schnet.py:158, comenet.py:297-298)
__global__ void k_edge_dist(const float* __restrict__ pos, const int* __restrict__ src,
                            const int* __restrict__ dst, int E, int mode, float* __restrict__ dist,
                            const int* __restrict__ cnt, float pad) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  if (cnt && e >= *cnt) {  // padded edge of a static-shape batch: a harmless in-range distance
    dist[e] = pad;
    return;
  }
  f3 pj = load3(pos, src[e]), pi = load3(pos, dst[e]);
  dist[e] = mode == 0 ? ref_len(f3_sub(pi, pj)) : ref_norm(f3_sub(pj, pi));
}

// one thread per triplet t = (k -> j -> i):  angle[t], torsion[t], targ[t] (CSR position of the
// arg-min reference neighbour; -1 when only the self term exists).
__global__ void k_triplet_geom(const float* __restrict__ pos, const int* __restrict__ rowptr,
                               const int* __restrict__ col, const int* __restrict__ esrc,
                               const int* __restrict__ edst, const int* __restrict__ kj,
                               const int* __restrict__ ji, int T, int use_torsion,
                               float* __restrict__ angle, float* __restrict__ torsion,
                               int* __restrict__ targ, const int* __restrict__ cnt) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  if (cnt && t >= *cnt) {
    angle[t] = 0.f;
    if (use_torsion) torsion[t] = 0.f;
    if (use_torsion && targ) targ[t] = -1;
    return;
  }
  int e = ji[t];
  int i = edst[e], j = esrc[e], k = esrc[kj[t]];
  f3 pj = load3(pos, j);
  f3 v_ji = f3_sub(load3(pos, i), pj);
  f3 v_jk = f3_sub(load3(pos, k), pj);
  float a = ref_dot(v_ji, v_jk);
  float b = ref_norm(ref_cross(v_ji, v_jk));
  angle[t] = atan2f(b, a);
  if (!use_torsion) return;
  float d_ji = ref_len(v_ji);
  f3 plane1 = ref_cross(v_ji, v_jk);
  float best = INFINITY;
  int arg = -1;
  for (int p = rowptr[j], en = rowptr[j + 1]; p < en; ++p) {
    int kn = col[p];
    if (kn == i) continue;
    f3 v_jn = f3_sub(load3(pos, kn), pj);
    f3 plane2 = ref_cross(v_ji, v_jn);
    float ta = ref_dot(plane1, plane2);
    float tb = ref_dot(ref_cross(plane1, plane2), v_ji) / d_ji;
    float tor = atan2f(tb, ta);
    if (tor <= 0.0f) tor += DIG3D_2PI_F;
    if (tor < best) {
      best = tor;
      arg = p;
    }
  }
  torsion[t] = arg >= 0 ? best : 0.0f;
  if (targ) targ[t] = arg;
}

// per-segment (min, first arg-min) of val[map[p]] (+ add[map[p]]) over p in [kptr[s], kptr[s+1]);
// map == nullptr => identity.  Empty segment: value 0, arg = sentinel (torch_scatter.scatter_min).
__global__ void k_segment_argmin(const float* __restrict__ val, const float* __restrict__ add,
                                 const int* __restrict__ kptr, const int* __restrict__ map, int S,
                                 int sentinel, float* __restrict__ out_val, int* __restrict__ out_arg) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  float best = INFINITY;
  int arg = sentinel;
  for (int p = kptr[s], en = kptr[s + 1]; p < en; ++p) {
    int m = map ? map[p] : p;
    float v = val[m];
    if (add) v = v + add[m];
    if (arg == sentinel || v < best) {  // first element always taken; later ones only on strict '<'
      best = v;
      arg = m;
    }
  }
  if (out_val) out_val[s] = arg == sentinel ? 0.0f : best;
  out_arg[s] = arg;
}

// add[clamp(arg[n])] = cutoff  (comenet.py:305-308: sentinel >= E is clamped to edge 0 first)
__global__ void k_bump(const int* __restrict__ arg, int N, int E, float cutoff, float* __restrict__ add) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  int a = arg[n];
  if (a >= E) a = 0;
  add[a] = cutoff;
}

// ComENet per-edge angles.  a0/a1: nearest / 2nd nearest incoming edge of every node (keyed by
// target i); b0/b1: the same keyed by source j (comenet.py:304-327), sentinels clamped to 0 here.
__global__ void k_comenet_geom(const float* __restrict__ pos, const int* __restrict__ src,
                               const int* __restrict__ dst, int E, const int* __restrict__ a0,
                               const int* __restrict__ a1, const int* __restrict__ b0,
                               const int* __restrict__ b1, float* __restrict__ theta,
                               float* __restrict__ phi, float* __restrict__ tau) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int j = src[e], i = dst[e];
  auto clampE = [E](int a) { return a >= E ? 0 : a; };
  auto vec = [&](int q) { return f3_sub(load3(pos, src[q]), load3(pos, dst[q])); };
  int a0i = clampE(a0[i]), a1i = clampE(a1[i]);
  int b0j = clampE(b0[j]), b1j = clampE(b1[j]);
  int n0 = src[a0i];
  int n0_j = dst[b0j];
  int idx_iref = (n0 == j) ? a1i : a0i;
  int idx_jref = (n0_j == i) ? b1j : b0j;
  f3 v_ji = vec(e), v_in0 = vec(a0i), v_in1 = vec(a1i), v_iref = vec(idx_iref), v_jref = vec(idx_jref);
  f3 nv = f3_neg(v_ji);
  // theta
  float a = ref_dot(nv, v_in0);
  float b = ref_norm(ref_cross(nv, v_in0));
  float th = atan2f(b, a);
  if (th < 0.0f) th += DIG3D_PI_F;
  theta[e] = th;
  // phi
  float d = ref_len(v_ji);
  f3 p1 = ref_cross(nv, v_in0), p2 = ref_cross(nv, v_in1);
  a = ref_dot(p1, p2);
  b = ref_dot(ref_cross(p1, p2), v_ji) / d;
  float ph = atan2f(b, a);
  if (ph < 0.0f) ph += DIG3D_PI_F;
  phi[e] = ph;
  // tau
  p1 = ref_cross(v_ji, v_jref);
  p2 = ref_cross(v_ji, v_iref);
  a = ref_dot(p1, p2);
  b = ref_dot(ref_cross(p1, p2), v_ji) / d;
  float ta = atan2f(b, a);
  if (ta < 0.0f) ta += DIG3D_PI_F;
  tau[e] = ta;
}

extern "C" {

int dig3d_edge_dist(const float* pos, const int* src, const int* dst, int E, int mode, float* dist,
                    const int* cnt, float pad, void* stream) {
  DIG3D_ENTER();
  if (E <= 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_edge_dist, dim3(dig3d_blocks(E, 256)), dim3(256), 0, (hipStream_t)stream, pos, src, dst, E,
                     mode, dist, cnt, pad);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_triplet_geom(const float* pos, const int* rowptr, const int* col, const int* esrc, const int* edst,
                       const int* kj, const int* ji, int T, int use_torsion, float* angle, float* torsion,
                       int* targ, const int* cnt, void* stream) {
  DIG3D_ENTER();
  if (T <= 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_triplet_geom, dim3(dig3d_blocks(T, 256)), dim3(256), 0, (hipStream_t)stream, pos, rowptr,
                     col, esrc, edst, kj, ji, T, use_torsion, angle, torsion, targ, cnt);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_segment_argmin(const float* val, const float* add, const int* kptr, const int* map, int S,
                         int sentinel, float* out_val, int* out_arg, void* stream) {
  DIG3D_ENTER();
  if (S <= 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_segment_argmin, dim3(dig3d_blocks(S, 256)), dim3(256), 0, (hipStream_t)stream, val, add,
                     kptr, map, S, sentinel, out_val, out_arg);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_comenet_bump(const int* arg, int N, int E, float cutoff, float* add, void* stream) {
  DIG3D_ENTER();
  hipStream_t st = (hipStream_t)stream;
  if (E <= 0) return DIG3D_OK;
  if (hipMemsetAsync(add, 0, sizeof(float) * (size_t)E, st) != hipSuccess) return DIG3D_ERR_LAUNCH;
  if (N > 0) hipLaunchKernelGGL(k_bump, dim3(dig3d_blocks(N, 256)), dim3(256), 0, st, arg, N, E, cutoff, add);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_comenet_geom(const float* pos, const int* src, const int* dst, int E, const int* a0, const int* a1,
                       const int* b0, const int* b1, float* theta, float* phi, float* tau, void* stream) {
  DIG3D_ENTER();
  if (E <= 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_comenet_geom, dim3(dig3d_blocks(E, 256)), dim3(256), 0, (hipStream_t)stream, pos, src, dst,
                     E, a0, a1, b0, b1, theta, phi, tau);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

}  // extern "C"
