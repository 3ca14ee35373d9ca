// dig3d geometry: xyz -> (dist, angle, torsion) and ComENet's (dist, theta, phi, tau).
// Float32 with the reference's exact IEEE operation order (common.h "reference float" helpers):
//   utils/geometric_computing.py:25,44-48,65-75      (dist, angle, torsion = min over quadruplets)
//   method/schnet/schnet.py:158                       (dist = norm)
//   method/comenet/comenet.py:297-385                 (nearest / 2nd-nearest references, theta, phi, tau)
// The quadruplet list (Q ~ T*deg entries, five int64 + six float[Q,3] temporaries in the reference) is
// never materialised: each triplet walks the CSR row of j in registers and keeps a running min + argmin.
#include "common.h"
#include "edge_values.h"

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
  dist[e] = edge_dist_value(pos, src[e], dst[e], mode);
}

// TG_LPT = 4 lanes per triplet t = (k -> j -> i):  angle[t], torsion[t], targ[t] (CSR position of the arg-min reference
// neighbour; -1 when only the self term exists).  The candidates n of the torsion minimum (the incoming edges of j, ~14 at
// QM9 sizes, one atan2 each) are dealt to the four lanes; the group keeps (value, position) and prefers the SMALLER position
// on equal values — the first minimum in ascending position order, what the one-thread loop (and torch_scatter's
// scatter_min over the reference's ascending quadruplets) selects: bit-identical outputs, 13.1 -> ~7 us at 1.0e5 triplets.
#define TG_LPT 4
__global__ void __launch_bounds__(256) k_triplet_geom(const float* __restrict__ pos, const int* __restrict__ rowptr,
                               const int* __restrict__ col, const int* __restrict__ esrc,
                               const int* __restrict__ edst, const int* __restrict__ kj,
                               const int* __restrict__ ji, int T, int use_torsion,
                               float* __restrict__ angle, float* __restrict__ torsion,
                               int* __restrict__ targ, const int* __restrict__ cnt) {
  const int t = (blockIdx.x * blockDim.x + threadIdx.x) / TG_LPT, sub = threadIdx.x & (TG_LPT - 1);
  if (t >= T) return;                      // the lanes of a triplet leave together
  if (cnt && t >= *cnt) {
    if (sub == 0) {
      angle[t] = 0.f;
      if (use_torsion) torsion[t] = 0.f;
      if (use_torsion && targ) targ[t] = -1;
    }
    return;
  }
  int e = ji[t];
  int i = edst[e], j = esrc[e], k = esrc[kj[t]];
  f3 pj = load3(pos, j);
  f3 v_ji = f3_sub(load3(pos, i), pj);
  f3 v_jk = f3_sub(load3(pos, k), pj);
  float a = ref_dot(v_ji, v_jk);
  float b = ref_norm(ref_cross(v_ji, v_jk));
  if (sub == 0) angle[t] = atan2f(b, a);
  if (!use_torsion) return;                // uniform
  float d_ji = ref_len(v_ji);
  f3 plane1 = ref_cross(v_ji, v_jk);
  float best = INFINITY;
  int arg = 0x7fffffff;
  for (int p = rowptr[j] + sub, en = rowptr[j + 1]; p < en; p += TG_LPT) {
    int kn = col[p];
    if (kn == i) continue;
    f3 v_jn = f3_sub(load3(pos, kn), pj);
    f3 plane2 = ref_cross(v_ji, v_jn);
    float ta = ref_dot(plane1, plane2);
    float tb = ref_dot(ref_cross(plane1, plane2), v_ji) / d_ji;
    float tor = atan2f(tb, ta);
    if (tor <= 0.0f) tor += DIG3D_2PI_F;
    if (tor < best) {                      // ascending p inside a lane: the lane's first minimum
      best = tor;
      arg = p;
    }
  }
#pragma unroll
  for (int off = TG_LPT / 2; off > 0; off >>= 1) {
    const float ob = __shfl_xor(best, off, 64);
    const int oa = __shfl_xor(arg, off, 64);
    if (ob < best || (ob == best && oa < arg)) {
      best = ob;
      arg = oa;
    }
  }
  if (sub == 0) {
    const bool found = arg != 0x7fffffff;
    torsion[t] = found ? best : 0.0f;
    if (targ) targ[t] = found ? arg : -1;
  }
}

// per-segment (min, first arg-min) of val[map[p]] (+ add[map[p]]) over p in [kptr[s], kptr[s+1]);
// map == nullptr => identity.  Empty segment: value 0, arg = sentinel (torch_scatter.scatter_min).
// 16 lanes per segment (32 neighbours per atom at ComENet's 128-atom molecules: a thread per segment was a chain of 32
// dependent index -> value loads on a quarter of the chip, 21 us per launch): lane l takes positions p0 + l, p0 + l + 16,
// ..., the group keeps (value, position) and prefers the smaller position on ties — the FIRST occurrence of the minimum,
// as torch_scatter's scatter_min (comenet.py:304-327).
__global__ void __launch_bounds__(256) k_segment_argmin16(const float* __restrict__ val, const float* __restrict__ add,
                                                           const int* __restrict__ kptr, const int* __restrict__ map,
                                                           int S, int sentinel, float* __restrict__ out_val,
                                                           int* __restrict__ out_arg) {
  const int s = (blockIdx.x * 256 + threadIdx.x) >> 4, l = threadIdx.x & 15;
  if (s >= S) return;                      // the 16 lanes of a segment leave together
  float best = INFINITY;
  int bp = 0x7fffffff, arg = sentinel;
  for (int p = kptr[s] + l, en = kptr[s + 1]; p < en; p += 16) {
    const int m = map ? map[p] : p;
    float v = val[m];
    if (add) v = v + add[m];
    if (bp == 0x7fffffff || v < best) {
      best = v;
      bp = p;
      arg = m;
    }
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, 16);
    const int op = __shfl_xor(bp, off, 16), om = __shfl_xor(arg, off, 16);
    // an empty lane (op == INT_MAX) never wins; a non-empty one beats an empty one whatever its value (the serial loop
    // takes its first element unconditionally, NaN included)
    const bool take = op != 0x7fffffff && (bp == 0x7fffffff || ov < best || (ov == best && op < bp));
    if (take) {
      best = ov;
      bp = op;
      arg = om;
    }
  }
  if (l == 0) {
    if (out_val) out_val[s] = arg == sentinel ? 0.0f : best;
    out_arg[s] = arg;
  }
}
__global__ void k_bump(const int* __restrict__ arg, int N, int E, float cutoff, float* __restrict__ add,
                       const int* __restrict__ cnt_n) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N || (cnt_n && n >= *cnt_n)) return;   // padded nodes of a static-shape batch are not atoms: an atom WITHOUT
                                                  // neighbours bumps edge 0 (the reference clamps its sentinel to 0), they must not
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
                               float* __restrict__ phi, float* __restrict__ tau, const int* __restrict__ cnt) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  if (cnt && e >= *cnt) {          // padded edges of a static-shape batch (dig_amd/graphed.py): finite, never read through a CSR
    theta[e] = 0.f;
    phi[e] = 0.f;
    tau[e] = 0.f;
    return;
  }
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

// G-SphereNet's private geometry (dig/ggraph3D/method/G_SphereNet/model/geometric_computing.py:13-19,83-103): the torsion
// reference of a triplet (k -> j -> i) is j's NEAREST node of the same molecule (knn_graph k = 1, no cutoff), or the
// second nearest when the nearest is i.  n1/n2[v] = nearest / second-nearest node of v inside its graph (squared
// float32 distances, strict '<': the lower index wins a tie), -1 when the graph is too small.
__global__ void k_nearest_two(const float* __restrict__ pos, const int* __restrict__ batch, const int* __restrict__ gptr,
                              int N, int* __restrict__ n1, int* __restrict__ n2) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= N) return;
  const int b = batch[v];
  const f3 p = load3(pos, v);
  float d1 = INFINITY, d2 = INFINITY;
  int a1 = -1, a2 = -1;
  for (int u = gptr[b], ue = gptr[b + 1]; u < ue; ++u) {
    if (u == v) continue;
    const f3 q = f3_sub(p, load3(pos, u));
    const float d = (q.x * q.x + q.y * q.y) + q.z * q.z;
    if (d < d1) {
      d2 = d1; a2 = a1;
      d1 = d; a1 = u;
    } else if (d < d2) {
      d2 = d; a2 = u;
    }
  }
  n1[v] = a1;
  n2[v] = a2;
}

// angle[t] as k_triplet_geom; torsion[t] in (0, 2 pi] against the knn reference of j (no minimum over neighbours).
__global__ void k_triplet_geom_knn(const float* __restrict__ pos, const int* __restrict__ esrc,
                                   const int* __restrict__ edst, const int* __restrict__ kj,
                                   const int* __restrict__ ji, int T, const int* __restrict__ n1,
                                   const int* __restrict__ n2, float* __restrict__ angle, float* __restrict__ torsion) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const int e = ji[t];
  const int i = edst[e], j = esrc[e], k = esrc[kj[t]];
  const f3 pj = load3(pos, j);
  const f3 v_ji = f3_sub(load3(pos, i), pj);
  const f3 v_jk = f3_sub(load3(pos, k), pj);
  angle[t] = atan2f(ref_norm(ref_cross(v_ji, v_jk)), ref_dot(v_ji, v_jk));
  int kn = n1[j];
  if (kn == i) kn = n2[j];
  if (kn < 0) {
    torsion[t] = 0.f;
    return;
  }
  const f3 v_jn = f3_sub(load3(pos, kn), pj);
  const float d_ji = ref_len(v_ji);
  const f3 plane1 = ref_cross(v_ji, v_jk), plane2 = ref_cross(v_ji, v_jn);
  const float a = ref_dot(plane1, plane2);
  const float b = ref_dot(ref_cross(plane1, plane2), v_ji) / d_ji;
  float tor = atan2f(b, a);
  if (tor <= 0.0f) tor += DIG3D_2PI_F;
  torsion[t] = tor;
}

// ProNet per-edge geometry (method/pronet/pronet.py:385-446), float32 in the reference's operation order.
// Reference atoms are SEQUENCE neighbours (i-1, i+1) modulo the total node count of the batch (the reference's own
// wrap-around, :396-397).  level 0 (aminoacid): theta, phi, tau;  level 1 (backbone / allatom): theta, phi and the three
// Euler angles between the local frames built from (N, CA, C) of residues i and j.
__global__ void k_pronet_geom(const float* __restrict__ pos, const float* __restrict__ pos_n,
                              const float* __restrict__ pos_c, const int* __restrict__ src,
                              const int* __restrict__ dst, int E, int N, int level, float* __restrict__ dist,
                              float* __restrict__ theta, float* __restrict__ phi, float* __restrict__ a1,
                              float* __restrict__ a2, float* __restrict__ a3) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int j = src[e], i = dst[e];
  const int refi0 = (i - 1 + N) % N, refi1 = (i + 1) % N;
  const f3 pi = load3(pos, i), pj = load3(pos, j);
  const float d = ref_norm(f3_sub(pi, pj));                       // (pos[i] - pos[j]).norm(dim=1)
  dist[e] = d;
  const f3 vij = f3_sub(pj, pi);                                  // pos[j] - pos[i]
  const f3 v0 = f3_sub(load3(pos, refi0), pi), v1 = f3_sub(load3(pos, refi1), pi);
  theta[e] = atan2f(ref_norm(ref_cross(vij, v0)), ref_dot(vij, v0));
  {
    const f3 p1 = ref_cross(v0, v1), p2 = ref_cross(v0, vij);
    const float a = ref_dot(p1, p2);
    const float b = ref_dot(ref_cross(p1, p2), v0) / ref_norm(v0);
    phi[e] = atan2f(b, a);
  }
  if (level == 0) {
    const int refj0 = (j - 1 + N) % N, refj1 = (j + 1) % N;
    const int refi = (refi0 == j) ? refi1 : refi0;
    const int refj = (refj0 == i) ? refj1 : refj0;
    const f3 p1 = ref_cross(vij, f3_sub(load3(pos, refi), pi));
    const f3 p2 = ref_cross(vij, f3_sub(load3(pos, refj), pj));
    const float a = ref_dot(p1, p2);
    const float b = ref_dot(ref_cross(p1, p2), vij) / d;
    a1[e] = atan2f(b, a);                                          // tau
    return;
  }
  const f3 o1x = f3_sub(load3(pos_n, i), pi);
  const f3 o1z = ref_cross(o1x, ref_cross(o1x, f3_sub(load3(pos_c, i), pi)));
  const float l1 = ref_norm(o1z) + 1e-7f;
  const f3 o2x = f3_sub(load3(pos_n, j), pj);
  const f3 o2z = ref_cross(o2x, ref_cross(o2x, f3_sub(load3(pos_c, j), pj)));
  const float l2 = ref_norm(o2z) + 1e-7f;
  const f3 nn = ref_cross(o1z, o2z);
  a1[e] = atan2f(ref_dot(ref_cross(o1x, nn), o1z) / l1, ref_dot(o1x, nn));
  a2[e] = atan2f(ref_norm(ref_cross(o1z, o2z)), ref_dot(o1z, o2z));
  a3[e] = atan2f(ref_dot(ref_cross(nn, o2x), o2z) / l2, ref_dot(nn, o2x));
}

// pos_emb (pronet.py:362-372): E[e, 0:h] = cos((j-i) f_k), E[e, h:2h] = sin((j-i) f_k),
// f_k = exp(2k * -(ln 10000 / num_pos_emb)) computed by the host exactly as torch does.
__global__ void k_pos_emb(const int* __restrict__ src, const int* __restrict__ dst, int E,
                          const float* __restrict__ freq, int half, float* __restrict__ out) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (int64_t)E * half) return;
  const int e = (int)(q / half), k = (int)(q - (int64_t)e * half);
  const float ang = (float)(src[e] - dst[e]) * freq[k];
  out[(int64_t)e * 2 * half + k] = cosf(ang);
  out[(int64_t)e * 2 * half + half + k] = sinf(ang);
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
  hipLaunchKernelGGL(k_triplet_geom, dim3(dig3d_blocks((int64_t)T * TG_LPT, 256)), dim3(256), 0, (hipStream_t)stream, pos, rowptr,
                     col, esrc, edst, kj, ji, T, use_torsion, angle, torsion, targ, cnt);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_segment_argmin(const float* val, const float* add, const int* kptr, const int* map, int S,
                         int sentinel, float* out_val, int* out_arg, void* stream) {
  DIG3D_ENTER();
  if (S <= 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_segment_argmin16, dim3(dig3d_blocks((int64_t)S * 16, 256)), dim3(256), 0, (hipStream_t)stream, val,
                     add, kptr, map, S, sentinel, out_val, out_arg);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_comenet_bump(const int* arg, int N, int E, float cutoff, float* add, const int* cnt_n, void* stream) {
  DIG3D_ENTER();
  hipStream_t st = (hipStream_t)stream;
  if (E <= 0) return DIG3D_OK;
  if (dig3d_zero_async(add, sizeof(float) * (size_t)E, st) != hipSuccess) return DIG3D_ERR_LAUNCH;
  if (N > 0) hipLaunchKernelGGL(k_bump, dim3(dig3d_blocks(N, 256)), dim3(256), 0, st, arg, N, E, cutoff, add, cnt_n);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_comenet_geom(const float* pos, const int* src, const int* dst, int E, const int* a0, const int* a1,
                       const int* b0, const int* b1, float* theta, float* phi, float* tau, const int* cnt, void* stream) {
  DIG3D_ENTER();
  if (E <= 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_comenet_geom, dim3(dig3d_blocks(E, 256)), dim3(256), 0, (hipStream_t)stream, pos, src, dst,
                     E, a0, a1, b0, b1, theta, phi, tau, cnt);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_pronet_geom(const float* pos, const float* pos_n, const float* pos_c, const int* src, const int* dst, int E,
                      int N, int level, float* dist, float* theta, float* phi, float* a1, float* a2, float* a3,
                      void* stream) {
  DIG3D_ENTER();
  if (E <= 0) return DIG3D_OK;
  if (!pos || !src || !dst || N <= 0 || !dist || !theta || !phi || !a1 || (level != 0 && (!pos_n || !pos_c || !a2 || !a3)))
    return DIG3D_ERR_ARG;
  hipLaunchKernelGGL(k_pronet_geom, dim3(dig3d_blocks(E, 256)), dim3(256), 0, (hipStream_t)stream, pos, pos_n, pos_c, src,
                     dst, E, N, level, dist, theta, phi, a1, a2, a3);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_pos_emb(const int* src, const int* dst, int E, const float* freq, int half, float* out, void* stream) {
  DIG3D_ENTER();
  if (E <= 0) return DIG3D_OK;
  if (!src || !dst || !freq || half < 1 || !out) return DIG3D_ERR_ARG;
  hipLaunchKernelGGL(k_pos_emb, dim3(dig3d_blocks((int64_t)E * half, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, E,
                     freq, half, out);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_nearest_two(const float* pos, const int* batch, const int* gptr, int N, int* n1, int* n2, void* stream) {
  DIG3D_ENTER();
  if (N <= 0) return DIG3D_OK;
  if (!pos || !batch || !gptr || !n1 || !n2) return DIG3D_ERR_ARG;
  hipLaunchKernelGGL(k_nearest_two, dim3(dig3d_blocks(N, 128)), dim3(128), 0, (hipStream_t)stream, pos, batch, gptr, N, n1,
                     n2);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_triplet_geom_knn(const float* pos, const int* esrc, const int* edst, const int* kj, const int* ji, int T,
                           const int* n1, const int* n2, float* angle, float* torsion, void* stream) {
  DIG3D_ENTER();
  if (T <= 0) return DIG3D_OK;
  if (!pos || !esrc || !edst || !kj || !ji || !n1 || !n2 || !angle || !torsion) return DIG3D_ERR_ARG;
  hipLaunchKernelGGL(k_triplet_geom_knn, dim3(dig3d_blocks(T, 256)), dim3(256), 0, (hipStream_t)stream, pos, esrc, edst, kj,
                     ji, T, n1, n2, angle, torsion);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

}  // extern "C"
