/* dig3d.h — C ABI of libdig3d.so, the MI355X (gfx950) engine behind DIG's dig.threedgraph hot path.
 *
 * The reference (divelab/DIG) has NO native/FFI interface on this path: it calls Python functions of
 * four third-party wheels.  Each entry point below names the reference call site(s) whose arithmetic it
 * replaces (paths relative to /root/reference/dig/threedgraph/).  The Python host in dig_amd/ binds these
 * symbols with ctypes (dig_amd/_hip.py); INTEGRATION.md shows the stub a DIG maintainer would add.
 *
 * Conventions
 *   - every pointer is DEVICE memory (HBM) unless named host_*; plain pointers + sizes, no torch types;
 *   - float data is float32 row-major [rows, C]; internal indices are int32; int64 only where the
 *     reference API hands int64 (batch vector, scatter index);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it, nothing
 *     inside synchronises or allocates: outputs and workspaces are caller-provided;
 *   - return 0 on success, DIG3D_ERR_ARG (-1) for bad arguments, DIG3D_ERR_LAUNCH (-2) if HIP rejected
 *     a launch; no exceptions cross the ABI; thread-safe for distinct streams;
 *   - index contents are trusted (as in torch_scatter): out-of-range indices are undefined behaviour;
 *   - `cnt` parameters (device int32, may be NULL): static-shape batches for HIP-graph replay allocate every
 *     array at a bucket capacity; *cnt is the live row count, rows beyond it are padding (written as exact
 *     zeros / a harmless constant and never reduced).  CSR-driven kernels need no cnt: padded segments are empty.
 */
#ifndef DIG3D_H
#define DIG3D_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DIG3D_OK 0
#define DIG3D_ERR_ARG (-1)
#define DIG3D_ERR_LAUNCH (-2)

/* ---------------------------------------------------------------------------------------------------
 * Graph construction (graph.hip) — integer work, bit-exact.
 * ------------------------------------------------------------------------------------------------- */

/* torch_cluster.radius_graph(pos, r, batch, loop, max_num_neighbors) as called at
 * method/spherenet/spherenet.py:304, method/dimenetpp/dimenetpp.py:277, method/schnet/schnet.py:156,
 * method/comenet/comenet.py:294 — plus the CSR and the triplet COUNT of
 * utils/geometric_computing.py:27-30, all without a host round trip.
 *   in : pos[N,3] f32, batch[N] i64 sorted ascending
 *   out: ptr[N+2]   graph pointer (ptr[g] = first node of graph g)
 *        nbr[N*W], deg[N]   padded neighbour table, W = max_num_neighbors + (loop ? 0 : 1)
 *        rowptr[N+1] CSR over targets; src[N*W], dst[N*W]: first E entries = edge list grouped by target
 *        (ascending), sources ascending inside a target  == edge_index[0], edge_index[1]
 *        cnt[N*W], tptr[N*W+1]: first E(+1) entries = triplets per edge / their exclusive scan
 *        meta[8] i64: [0]=B graphs, [1]=E edges, [2]=T triplets, [7] bit0 = batch not sorted
 *        ws[N*W/4096 + 2] scratch;  batch32[N] (or NULL): the batch vector as int32
 *        meta_host (or NULL): PINNED host memory, int64[8] — the last kernel of the build writes meta there itself (the
 *        device stores to mapped host memory), so the sizes reach the host without a copy command
 * The caller waits for an event recorded behind this call (or, meta_host = NULL, copies meta to the host once), then sizes
 * idx_kj/idx_ji and calls dig3d_graph_triplets_fill. */
int dig3d_graph_build(const float* pos, const int64_t* batch, int N, float r, int max_num_neighbors, int loop,
                      int* ptr, int* nbr, int* deg, int* rowptr, int* src, int* dst, int* cnt, int* tptr,
                      int64_t* meta, int* ws, int want_triplets, int* batch32, int64_t* meta_host, void* stream);

/* Gradient pieces -> ONE flat buffer (the optimizer's and the data-parallel all-reduce's layout; run.py:121-134 hands the
 * per-parameter .grad tensors to torch.optim.Adam): piece p copies n[p] floats from src[p] (NULL: zeros) to
 * flat[off[p] ...] and zero-fills up to span[p].  Host arrays of np entries. */
int dig3d_pack_flat(int np, const void* const* src, const int* n, const int* span, const int* off, float* flat,
                    void* stream);

/* idx_kj / idx_ji of utils/geometric_computing.py:33-41 (SparseTensor row-select + mask), int32.
 * CSR (rowptr, col[, val]) over targets; val = original edge id per CSR entry or NULL (identity);
 * esrc/edst = edge endpoints in original edge order; tptr[E+1] from the count stage. */
int dig3d_graph_triplets_fill(const int* rowptr, const int* col, const int* val, const int* esrc,
                              const int* edst, const int* tptr, int E, int* kj, int* ji, void* stream);

/* count stage for a caller-supplied edge list (public xyz_to_dat path): tptr[E+1], *total = T. */
int dig3d_graph_triplets_count(const int* rowptr, const int* col, const int* esrc, const int* edst, int E,
                               int* cnt, int* tptr, int64_t* total, int* ws, void* stream);

/* Transposed CSR of key[M] in [0,S): kptr[S+1], perm[M] = positions grouped by key, ascending inside a
 * key (stable counting sort).  Turns the backward of the row gathers x[i], x[j], x_kj[idx_kj]
 * (ATen index backward = unsorted scatter_add; spherenet.py:88,165) into a contiguous segment sum.
 * hist[S], cursor[S], tmp[M], ws[S/4096+3] scratch. */
int dig3d_csr_by_key(const int* key, int M, int S, int* kptr, int* perm, int* hist, int* cursor, int* tmp, int* ws,
                     void* stream);

/* The same for n <= 4 keys in one set of launches (host arrays of device pointers / sizes): a training step needs the
 * grouping of the edges by source AND of the triplets by their k->j edge.  hc[i]: 2*S[i] ints (histogram + cursors; adjacent
 * buffers are zeroed together), S[i] <= 32768 (larger: one dig3d_csr_by_key per key). */
int dig3d_csr_by_keys(int n, const void* const* key, const int* M, const int* S, void* const* kptr, void* const* perm,
                      void* const* hc, void* const* tmp, void* stream);
/* The same on a workspace the caller keeps between calls: hc_clean = 1 promises that every hc[i] is all zero on entry (no
 * zero-fill launch in front of the histogram), 0 zeroes it first; both leave hc zero again (the last kernel of the set clears
 * it), so a caller that reuses one workspace passes 0 once and 1 from then on. */
int dig3d_csr_by_keys_ws(int n, const void* const* key, const int* M, const int* S, void* const* kptr, void* const* perm,
                         void* const* hc, void* const* tmp, int hc_clean, void* stream);

int dig3d_scan_i32(const int* in, int* out /* n+1 */, int n, int64_t* total, int* ws, void* stream);

/* Refill of the static-shape (bucket-capacity) buffers of a HIP-graph batch in ONE launch: for each of n <= 16
 * arrays of 4-byte words dst[a][0..live) = src[a], dst[a][live..cap) = fill[a]; cnt_out[0..3] = cnt[0..3].
 * src/dst/live_words/cap_words/fill/cnt are HOST arrays (descriptors travel as kernel arguments). */
int dig3d_pack_static(const void* const* src, void* const* dst, const int* live_words, const int* cap_words,
                      const uint32_t* fill, int n, const int* cnt, int* cnt_out, void* stream);
int dig3d_cast_i32_i64(const int* in, int64_t* out, int64_t n, void* stream);
int dig3d_cast_i64_i32(const int64_t* in, int* out, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Geometry (geometry.hip) — float32 in the reference's exact IEEE operation order.
 * ------------------------------------------------------------------------------------------------- */

/* mode 0: dist of utils/geometric_computing.py:25; mode 1: norm of method/schnet/schnet.py:158 and
 * method/comenet/comenet.py:297-298. */
int dig3d_edge_dist(const float* pos, const int* src, const int* dst, int E, int mode, float* dist,
                    const int* cnt, float pad, void* stream);

/* angle (geometric_computing.py:44-48) and torsion = min over quadruplets (:51-75) per triplet, without
 * materialising the quadruplet list; targ[t] = CSR position of the arg-min neighbour (may be NULL). */
int dig3d_triplet_geom(const float* pos, const int* rowptr, const int* col, const int* esrc, const int* edst,
                       const int* kj, const int* ji, int T, int use_torsion, float* angle, float* torsion,
                       int* targ, const int* cnt, void* stream);

/* torch_scatter.scatter_min(val (+add), key) over CSR segments (comenet.py:304,311,316,325): first
 * arg-min wins, empty segment -> value 0 and arg = sentinel.  map NULL = identity. */
int dig3d_segment_argmin(const float* val, const float* add, const int* kptr, const int* map, int S,
                         int sentinel, float* out_val, int* out_arg, void* stream);

/* add = zeros(E); add[clamp(arg[n])] = cutoff   (comenet.py:305-308, 317-321); cnt_n (or NULL): live atoms of a padded batch */
int dig3d_comenet_bump(const int* arg, int N, int E, float cutoff, float* add, const int* cnt_n, void* stream);

/* theta, phi, tau per edge (comenet.py:329-385) from the four arg-min tables. */
int dig3d_comenet_geom(const float* pos, const int* src, const int* dst, int E, const int* a0, const int* a1,
                       const int* b0, const int* b1, float* theta, float* phi, float* tau, const int* cnt, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Basis functions (basis.hip).
 * ------------------------------------------------------------------------------------------------- */

/* out[E, ns*nr] = norm[l,n] * j_l(zero[l,n] * dist/cutoff) (* Envelope(dist/cutoff) when envelope_p > 0,
 * envelope_p = exponent + 1): the radial factor of angle_emb / torsion_emb
 * (method/spherenet/features.py:213, method/dimenetpp/features.py:212-214, method/comenet/features.py:289,340).
 * zeros/norms: [ns*nr] float64 device arrays. */
int dig3d_bessel_basis(const float* dist, int E, float cutoff, int ns, int nr, const double* zeros,
                       const double* norms, int envelope_p, float* out, void* stream);

/* out[M, H*nr] = Y_h(theta, phi) * bes[g(m), order(h), n]; phi NULL -> m = 0 harmonics only (H = ns),
 * else H = ns*ns.  pair_mode 0: order(h) = h % ns (spherenet/features.py:262), 1: order = degree of h
 * (comenet/features.py:346-348).  pref[8*8]: harmonic prefactors, row stride 8. */
int dig3d_sph_basis(const float* bes, const int* gidx, const float* theta, const float* phi, int M, int ns,
                    int nr, const float* pref, int pair_mode, float* out, void* stream);

/* method/schnet/schnet.py:92-94 and :31 */
int dig3d_gauss_smear(const float* dist, int E, const float* offset, int G, float coeff, float* out,
                      void* stream);
int dig3d_cos_cutoff(const float* dist, int E, float cutoff, float* out, void* stream);

/* G-SphereNet's private geometry (dig/ggraph3D/method/G_SphereNet/model/geometric_computing.py:13-19,57-103): nearest and
 * second-nearest node of every node inside its molecule (knn_graph k = 1, 2), and angle / torsion of every triplet with
 * the torsion taken against that reference (the second nearest when the nearest is the triplet's own i). */
int dig3d_nearest_two(const float* pos, const int* batch, const int* gptr, int N, int* n1, int* n2, void* stream);
int dig3d_triplet_geom_knn(const float* pos, const int* esrc, const int* edst, const int* kj, const int* ji, int T,
                           const int* n1, const int* n2, float* angle, float* torsion, void* stream);

/* ProNet (method/pronet/pronet.py:385-446): per-edge dist, theta, phi and tau (level 0, aminoacid) or the three
 * Euler angles a1..a3 between the (N, CA, C) frames of residues i and j (level 1, backbone / allatom); reference
 * atoms are the sequence neighbours (i-1, i+1) modulo the batch's node count, as in the reference.
 * pos_emb (:362-372): cos / sin of (j - i) * freq[k], freq = the reference's exp(arange(0, P, 2) * -(ln 1e4 / P)). */
int dig3d_pronet_geom(const float* pos, const float* pos_n, const float* pos_c, const int* src, const int* dst, int E,
                      int N, int level, float* dist, float* theta, float* phi, float* a1, float* a2, float* a3,
                      void* stream);
int dig3d_pos_emb(const int* src, const int* dst, int E, const float* freq, int half, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Aggregation (segment.hip).
 * ------------------------------------------------------------------------------------------------- */

/* torch_scatter.scatter(src, index, dim=0, dim_size=S, reduce='sum') for a SORTED int64 index —
 * method/spherenet/spherenet.py:171,211,224,313; method/dimenetpp/dimenetpp.py:150,190,203,286;
 * method/schnet/schnet.py:55,81; method/comenet/comenet.py:398.  Rows of out without a source are 0.
 * Algorithmic bytes: 4*M*C + 8*M + 4*S*C. */
int dig3d_segment_sum_sorted(const float* src, const int64_t* index, int64_t M, int C, int64_t S, float* out,
                             void* stream);
/* the same with the work split passed explicitly (bench sweeps / tiling-independence tests; the library keeps no
 * mutable tuning state): rows_per_worker 0 = heuristic; mode bit 0 = index batch by one coalesced load + shuffles,
 * bit 1 = non-temporal source loads (the plain entry point uses 0, 3).  Results are bit-identical for every value. */
int dig3d_segment_sum_sorted_tuned(const float* src, const int64_t* index, int64_t M, int C, int64_t S, float* out,
                                   int rows_per_worker, int mode, void* stream);

/* out[s,:] = sum_{p in [kptr[s],kptr[s+1])} A[t,:] * X[ix[t],:] * B[t,:],  t = map ? map[p] : p.
 * Fuses gather * multiply * scatter_add: x_kj[idx_kj] * sbf * t -> scatter (spherenet.py:165-171),
 * v[j] * W -> scatter (schnet.py:34,55), edge_weight * x_j -> aggregate (comenet.py:130-133); with the
 * transposed CSR as (kptr, map) it is their backward w.r.t. X and the backward of any row gather. */
int dig3d_segment_fused(const float* X, const int* ix, const float* A, const float* B, const int* kptr,
                        const int* map, int S, int C, float* out, void* stream);

/* reduce='mean' flavours of the two reductions above — scatter(..., reduce='mean') of
 * dig/ggraph3D/method/G_SphereNet/model/spherenet.py:171-172,205,297 and PyG GraphNorm's scatter_mean
 * (method/comenet/comenet.py:160): rows divided by their segment length, empty rows 0. */
int dig3d_segment_mean_sorted(const float* src, const int64_t* index, int64_t M, int C, int64_t S, float* out,
                              void* stream);
int dig3d_segment_fused_mean(const float* X, const int* ix, const float* A, const float* B, const int* kptr,
                             const int* map, int S, int C, float* out, void* stream);

/* the atom-type embedding lookup out[m,:] = weight[idx[m],:] (spherenet.py:85, schnet.py:124, comenet.py:98); idx int64
 * in [0, V) (clamped for memory safety), C % 4 == 0. */
int dig3d_embedding_fwd(const int64_t* idx, const float* weight, int M, int V, int C, float* out, void* stream);
/* backward of the atom-type embedding lookup x = weight[idx] (spherenet.py:85, schnet.py:124, comenet.py:98):
 * gW[V,C] = sum over the M rows of g grouped by idx (int64 in [0, V), V <= 128); deterministic.
 * part: float[dig3d_embedding_bwd_chunks(M) * V * C]; reduce_now = 0 leaves the sum of the chunk tables (stride V*C) to the
 * caller's dig3d_reduce_many (M == 0 always writes gW = 0). */
int dig3d_embedding_bwd_chunks(int M);
int dig3d_embedding_bwd(const int64_t* idx, const float* g, int M, int V, int C, float* part, float* gW, int reduce_now,
                        void* stream);

/* ComENet's EdgeGraphConv with a feature-defined edge weight (method/comenet/comenet.py:130-133 propagate with
 * message = edge_weight * x_j, :160-175 edge_weight = lin_feature(feature)): out[S,C] = sum_{t in seg(s)} X[ix[t],:] *
 * (Wc f_t), the weight Wc f_t evaluated on the fly (never an [E,C] tensor).  F [M,K] row-major, Wc [C,K], K <= 16,
 * C in {64,128,256}; kptr / map as in dig3d_segment_fused (with the transposed CSR and ix = the other end of the edge
 * the same call is the gradient w.r.t. X; `add` [S,C] or NULL is added to the result: the gradient already accumulated on
 * x by its other consumers, so autograd has nothing to sum).  dig3d_featconv_wgrad: gWc[C,K] = sum_t f_t[k] G[ig[t],c]
 * X[ix[t],c]; part float[dig3d_featconv_wgrad_blocks(M) * C*K]. */
int dig3d_featconv_supported(int K, int C);
int dig3d_featconv(const float* X, const int* ix, const float* F, int K, const float* Wc, const int* kptr,
                   const int* map, int S, int C, float* out, const float* add, void* stream);
int dig3d_featconv_wgrad_blocks(int64_t M);
int dig3d_featconv_wgrad(const float* G, const int* ig, const float* X, const int* ix, const float* F, int K, int64_t M,
                         int C, float* part, float* gWc, int reduce_now, const int* cnt, void* stream);
/* out[s,:] = g[s,:] / max(kptr[s+1]-kptr[s], 1): the gradient of a segment mean before its row gather. */
int dig3d_rows_div_count(const float* g, const int* kptr, int S, int C, float* out, void* stream);

/* GraphNorm (torch_geometric.nn.GraphNorm, method/comenet/comenet.py:160,213) over graphs ptr[B+1] whose nodes are
 * contiguous: y = weight * (x - mean_g * mean_scale) / sqrt(var_g + eps) + bias, one workgroup per graph; mean / rstd
 * [B,C] are outputs kept for the backward.  Backward: gx and gparams[3C] = (g weight, g bias, g mean_scale);
 * part = float[B*3*C] receives the per-graph partials; gparams == NULL (B > 0): the caller sums them (B rows of stride
 * 3C, e.g. with dig3d_reduce_many together with the weight-gradient partials of the same backward pass). */
int dig3d_graphnorm_fwd(const float* x, const int* ptr, int B, int C, const float* weight, const float* bias,
                        const float* mean_scale, float eps, float* y, float* mean, float* rstd, void* stream);
/* y[r,:] = 0 for *start <= r < N (start: a device scalar, e.g. ptr + B): the padded node rows of a static-shape batch
 * (dig_amd/graphed.py) behind the last graph, which the per-graph kernels do not write. */
int dig3d_zero_rows_from(float* y, const int* start, int N, int C, void* stream);
int dig3d_graphnorm_bwd(const float* gy, const float* x, const int* ptr, int B, int C, const float* weight,
                        const float* mean_scale, const float* mean, const float* rstd, float* gx, float* part,
                        float* gparams, void* stream);

/* Composed weights of bias-free, activation-free two-layer projections (TwoLayerLinear, method/comenet/comenet.py:87-105:
 * lin2(lin1(x)) = x (W2 W1)^T).  np <= 16 products out[p] [N[p],K[p]] = W2[p] [N[p],Mid[p]] . W1[p] [Mid[p],K[p]] in one
 * launch; the backward turns the gradients gWc[p] of the products into gW2[p] = gWc W1^T and gW1[p] = W2^T gWc. */
int dig3d_compose_fwd(int np, const void* const* W2, const void* const* W1, const int* N, const int* Mid, const int* K,
                      void* const* out, void* stream);
int dig3d_compose_bwd(int np, const void* const* gWc, const void* const* W2, const void* const* W1, const int* N,
                      const int* Mid, const int* K, void* const* gW2, void* const* gW1, void* stream);

/* The input of the edge initialisation, torch.cat([x[i], x[j], rbf0], dim=-1) (method/spherenet/spherenet.py:88-89,
 * dimenetpp.py:74-75): out[e] = [ x[i[e], 0:Cx] | x[j[e], 0:Cx] | r[e, 0:Cr] ], the gathered parts of rows e >= *cnt (padding of
 * a static-shape batch) zero.  Backward: gx[n] = sum_{e: i[e] = n} G[e, 0:Cx] + sum_{e: j[e] = n} G[e, Cx:2Cx] through the two CSR
 * groupings of the edges (perm NULL: edges already in that order), gr[e] = G[e, 2Cx:].  Cx in {64, 128, 256}, Cr % 4 == 0. */
int dig3d_edge_cat_supported(int Cx, int Cr);
int dig3d_edge_cat(const float* x, const int* i, const int* j, const float* r, int64_t E, int Cx, int Cr, float* out,
                   const int* cnt, void* stream);
/* The same with x[n] = table[z[n]] looked up in the kernel (the nn.Embedding of method/spherenet/spherenet.py:84 folded in;
 * z int64 [N], values in [0, rows of table) checked by the caller): no [N, Cx] node-feature tensor.  Its backward is
 * dig3d_edge_cat_bwd followed by dig3d_embedding_bwd. */
int dig3d_edge_cat_emb(const float* table, const int64_t* z, const int* i, const int* j, const float* r, int64_t E, int Cx,
                       int Cr, float* out, const int* cnt, void* stream);
int dig3d_edge_cat_bwd(const float* G, const int* kptr_i, const int* perm_i, const int* kptr_j, const int* perm_j, int N,
                       int64_t E, int Cx, int Cr, float* gx, float* gr, void* stream);

/* out[m,:] = X[ix[m],:] * A[m,:] * B[m,:]  (ATen index at spherenet.py:88,165; schnet.py:34). */
int dig3d_gather_mul(const float* X, const int* ix, const float* A, const float* B, int64_t M, int C, float* out,
                     const int* cnt, void* stream);

/* P = G[ig[m]] * X[ix[m]]; outA = P * B; outB = P * A  — per-row factor gradients of dig3d_segment_fused.
 * cnt (device, optional): rows m >= *cnt are padding of a static-shape batch and are written as zeros. */
int dig3d_gather_mul2(const float* G, const int* ig, const float* X, const int* ix, const float* A,
                      const float* B, int64_t M, int C, float* outA, float* outB, const int* cnt, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Fused triplet interaction (triplet.hip) — method/spherenet/spherenet.py:163-171, dimenetpp.py:146-150:
 *     sbf = lin_sbf2(lin_sbf1(sbf)); t = lin_t2(lin_t1(t)); x_kj = scatter(x_kj[idx_kj] * sbf * t, idx_ji)
 * without ever writing the [T, ns*nr] / [T, ns*ns*nr] basis tables or the [T, int_emb] factors.
 * ------------------------------------------------------------------------------------------------- */

/* The stacked, transposed, zero-padded weight tables dig3d_basis_project reads: outS[k][l*8 + b] = Ws[l][b][k] (lin_sbf1 of
 * layer l, [bs_s[l], KS] as in the reference, spherenet.py:110-117), outT likewise from lin_t1 (NULL: no torsion); L <= 4. */
int dig3d_basis_stack(int L, const void* const* Ws, const void* const* Wt, const int* bs_s, const int* bs_t, int KS, int KT,
                      float* outS, float* outT, void* stream);

/* Ps[l][t][0..7] = lin_sbf1_l(angle_emb(t)),  Pt[l][t][0..7] = lin_t1_l(torsion_emb(t)) for l < L <= 4:
 * the basis row (features.py:213-222,256-263) is evaluated in registers from bes[E, ns*nr] (dig3d_bessel_basis),
 * angle[T], torsion[T] and immediately contracted with the stacked, transposed, zero-padded weights
 * Ws[ns*nr][32], Wt[ns*ns*nr][32] (column o = layer*8 + b).  torsion/Wt/Pt NULL => DimeNet++ (no torsion). */
int dig3d_basis_project(const float* bes, const int* kj, const float* angle, const float* torsion, int T, int ns,
                        int nr, const float* pref, const float* Ws, const float* Wt, int L, float* Ps, float* Pt,
                        const int* cnt, int route, void* stream);

/* Backward of dig3d_basis_project w.r.t. the weights: gWs[32][ns*nr], gWt[32][ns*ns*nr] (row l*8 + b = row b of layer l's
 * lin_sbf1 / lin_t1 weight) from gPs/gPt[L][T][8].
 * part: float[dig3d_basis_wgrad_blocks(T) * (ns*nr + ns*ns*nr) * 32] scratch (two-stage, deterministic). */
/* route (both entry points; an argument, the library holds no mutable state): 0 = matrix cores (basis_mfma.hip:
 * v_mfma_f32_16x16x4_f32, IEEE float32) for num_spherical 3 / 7 and T >= 2048, VALU kernels otherwise; 1 = VALU kernels
 * always (parity tests compare the two). */
int dig3d_basis_wgrad_blocks(int T);   /* reduce_now = 0 below: partials only, see dig3d_reduce_many */
int dig3d_basis_wgrad(const float* bes, const int* kj, const float* angle, const float* torsion, int T, int ns,
                      int nr, const float* pref, const float* gPs, const float* gPt, int L, float* part,
                      float* gWs, float* gWt, const int* cnt, int reduce_now, int route, void* stream);

/* The angular basis of DimeNet++ contracted with the first basis Linears of ALL interaction blocks, closed under
 * differentiation on two kernels (csrc/sbf2.hip) — the energy_and_force twin of dig3d_basis_project:
 *   sbf[t, l nr + n] = bes[idx_kj[t], l nr + n] * Y_l0(angle[t])     (method/dimenetpp/features.py:183-220)
 *   P_b = lin_sbf1_b(sbf) for every block b                           (method/dimenetpp/dimenetpp.py:146)
 * differentiated twice by method/run.py:126-131.  The [T, ns nr] table is never formed.
 * dig3d_sbf2_t (a thread per triplet): with h0/h1/h2 = (Y_l0, dY_l0/dtheta, d2Y_l0/dtheta2)(angle[t]),
 *   u0[ln] = A[kj[t],ln] h0_l + s[t] B[kj[t],ln] h1_l,   u1[ln] = A[kj[t],ln] h1_l + s[t] B[kj[t],ln] h2_l   (B, s may be NULL)
 *   outP[b][t,c] = sum_ln W[8b+c,ln] u0[ln];  outA[t] = sum_j gP[j/8][t,j%8] sum_ln W[j,ln] u1[ln];
 *   partW[blk][j (ns nr) + ln] = sum_{t in block blk} gP[..][t,j] u0[ln]   (dig3d_sbf2_blocks(T) blocks; dig3d_reduce_many);
 *   Hout[t,l] = B ? s[t] h1_l : h0_l   ([T,8], l >= ns zero).
 * A, B [E, ns nr]; W [J, ns nr], J = 8 L <= 64; gP / outP: host arrays of L device pointers to [T,8] matrices; every output
 * optional; L in {1, 2, 4, 8}.  Rows t >= *cnt (cnt NULL: T) are not read and written as zeros.
 * dig3d_sbf2_e (a wave per edge over the transposed CSR of idx_kj):
 *   out[e, l nr + n] = sum_j W[j, l nr + n] sum_{p in [kptr[e],kptr[e+1])} H[t,l] gP[j/8][t,j%8],  t = perm ? perm[p] : p.
 * The four passes of a step: forward sbf2_t(A = bes) -> P;  create_graph backward sbf2_t(bes; gP) -> g_angle, H and
 * sbf2_e -> g_bes;  its backward sbf2_t(A = c_bes, B = bes, s = c_angle; gP) -> d/dgP, d/dangle, d/dW partials, H and
 * sbf2_e -> d/dbes;  final backward as the second plus the weight partials.  (ns, nr) in {(7,6), (3,6), (3,4)}. */
int dig3d_sbf2_supported(int ns, int nr);
int dig3d_sbf2_blocks(int T);
int dig3d_sbf2_t(const float* A, const float* B, const float* s, const float* angle, const int* kj, const float* W, int J,
                 int ns, int nr, const float* pref, const void* const* gP, void* const* outP, float* outA, float* partW,
                 float* Hout, int T, const int* cnt, void* stream);
int dig3d_sbf2_e(const float* H, const void* const* gP, const float* W, int J, int ns, int nr, const int* kptr, const int* perm,
                 int E, float* out, void* stream);

/* out[s,:] = sum_{p in [kptr[s],kptr[s+1])} X[ix[t],:] * (W2s Ps[t]) * (W2t Pt[t]),  t = map ? map[p] : p.
 * Ps/Pt [T,8]; W2s/W2t [C,8] = lin_sbf2 / lin_t2 weights (zero padded to 8 columns); C in {16,32,64,128,256}.
 * Forward: X = x_kj, ix = idx_kj, (kptr,map) = (tptr, NULL).  Backward w.r.t. x_kj: X = grad_out, ix = idx_ji,
 * (kptr,map) = transposed CSR of idx_kj. */
int dig3d_triplet_fwd(const float* X, const int* ix, const float* Ps, const float* Pt, const float* W2s,
                      const float* W2t, const int* kptr, const int* map, int S, int C, float* out, int route,
                      void* stream);
/* The same with `out += add` ([S, C], NULL: none) inside the launch — the final backward of method/run.py:126-133
 * (energy_and_force), where a second gradient reaches the same tensor: one rounded addition, as the framework's. */
int dig3d_triplet_fwd_add(const float* X, const int* ix, const float* Ps, const float* Pt, const float* W2s,
                          const float* W2t, const int* kptr, const int* map, int S, int C, float* out, const float* add,
                          int route, void* stream);
/* name of the kernel dig3d_triplet_fwd launches for (S segments, C channels, torsion factor present, transposed CSR, route),
 * as rocprofv3 prints it — the measurement tools label their roofline line and look up PMC rows with the library's own
 * answer instead of a table of their own (tools/roofline_kernels.py).  Writes name[cap] (cap >= 32), returns the length. */
int dig3d_triplet_fwd_kernel(int S, int C, int torsion, int transposed, int route, char* name, int cap);

/* gPs/gPt [T,8] and gW2s/gW2t [C,8] of the same op.  part: float[dig3d_triplet_bwd_blocks(E,C,route) * 2*C*8].
 * route (an argument of all three; the library holds no mutable state): 0 = a wave per segment, a lane per channel
 * (triplet_wave.hip, 8 waves per SIMD) for C = 64 / 128 / 256, the lane-group kernels for C = 16 / 32.  At C = 64 / 128
 * the wave kernels run in the form that walks a segment's index chain ONCE — a lane per triplet reads position, triplet
 * id, row id and the projected rows, which then reach the products through LDS broadcasts (k_trip_fwd_l, k_trip_bwd_l);
 * at C = 256 in the form with the per-triplet operands as wave-uniform scalar loads (k_trip_fwd_w, k_trip_bwd_w).
 * 1 = the lane-group kernels always (16 ... 64 lanes per segment, four channels per lane); 2 / 3 = the scalar-operand /
 * index-chain-once forms always.  Results of all routes are bit-identical (the gradients w.r.t. the weights between
 * routes 0, 2, 3); parity tests compare them. */
int dig3d_triplet_bwd_blocks(int E, int C, int route);
int dig3d_triplet_bwd(const float* G, const float* X, const int* kj, const float* Ps, const float* Pt,
                      const float* W2s, const float* W2t, const int* tptr, int E, int C, float* gPs, float* gPt,
                      float* part, float* gW2s, float* gW2t, int reduce_now, int route, void* stream);
/* The same with gPs += gPs_add, gPt += gPt_add ([T, 8], NULL: none) inside the launch (see dig3d_triplet_fwd_add). */
int dig3d_triplet_bwd_add(const float* G, const float* X, const int* kj, const float* Ps, const float* Pt,
                          const float* W2s, const float* W2t, const int* tptr, int E, int C, float* gPs, float* gPt,
                          float* part, float* gW2s, float* gW2t, int reduce_now, int route, const float* gPs_add,
                          const float* gPt_add, void* stream);


/* ---------------------------------------------------------------------------------------------------
 * Dense hidden-channel layers (dense.hip) on the f32 matrix cores (v_mfma_f32_32x32x2_f32, exact f32):
 * `act(F.linear(x, W, b))` of method/spherenet/spherenet.py:34-50,79-91,150-182,209-216 (same in dimenetpp.py),
 * method/schnet/schnet.py:29-59, method/comenet/comenet.py:87-215.   act: 0 none, 1 swish, 2 shifted softplus.
 * A layer that is differentiated once may keep act'(z) instead of z: forward act | 4 (5, 6) writes the derivative to Z,
 * the backward entries (dig3d_linear_bwd*, dig3d_smallk_bwd, dig3d_wgrad_many) then take act = 3 ("Z is the
 * derivative": gZ = gY * Z) — the exponential is evaluated once per element instead of once per kernel that needs gZ.
 * Shapes: N % 8 == 0, any K (dig3d_linear_supported); row-major, base pointers 16-byte aligned.
 * ------------------------------------------------------------------------------------------------- */
int dig3d_linear_supported(int K, int N);

/* Y[M,N] = act(X[M,K] W[N,K]^T + bias[N]) (+ res[M,N]); Z (or NULL) receives X W^T + bias for the backward. */
int dig3d_linear_fwd(const float* X, const float* W, const float* bias, const float* res, int M, int K, int N,
                     int act, float* Y, float* Z, void* stream);

/* Y[M,N] = rs[m] * (X W^T + bias)[m,:]: the filter-generating layer of SchNet with its cosine cutoff,
 * `self.mlp(dist_emb) * C.view(-1, 1)` (method/schnet/schnet.py:31-33), in one launch.  D [M,N] (or NULL) receives rs[m]
 * in every column: the layer's backward is the plain one with Z = D, act = 3.  No gradient for rs. */
int dig3d_linear_fwd_rowscale(const float* X, const float* W, const float* bias, const float* rs, int M, int K, int N,
                              float* Y, float* D, void* stream);

/* gX[M,K] = (gY * act'(Z)) W (+ gx_add when non-NULL: the gradient already accumulated on the layer's input, e.g.
 * from a skip connection);  Z may be NULL when act == 0. */
int dig3d_linear_bwd_input(const float* gY, const float* Z, const float* W, int M, int K, int N, int act,
                           float* gX, const float* gx_add, void* stream);

/* The same two products with a COLUMN SLICE of the weight: W points at column c0 of a row-major [N, ldw] matrix and the K
 * columns from there are used — Y = act(X[M,K] W[:, c0:c0+K]^T + bias) (+ res), gX[M,K] = (gY act'(Z)) W[:, c0:c0+K].
 * `lin_cat(torch.cat([h1, h2], 1))` (method/comenet/comenet.py:199) = h1 W[:, :H]^T + (h2 W[:, H:]^T + b) without the
 * concatenation; only for dig3d_linear_wslice_supported(M, K, N) (large M % 32 == 0, 64 < K, N <= 256). */
int dig3d_linear_wslice_supported(int M, int K, int N);
int dig3d_linear_fwd_wslice(const float* X, const float* W, int ldw, const float* bias, const float* res, int M, int K,
                            int N, int act, float* Y, float* Z, void* stream);
int dig3d_linear_bwd_input_wslice(const float* gY, const float* Z, const float* W, int ldw, int M, int K, int N, int act,
                                  float* gX, const float* gx_add, void* stream);

/* gWb[N*K + N] = { gW[N,K] = (gY * act'(Z))^T X,  gb[N] = column sums }.  Two-stage deterministic reduction:
 * part = float[dig3d_linear_wgrad_blocks(M) * (N*K + N)] scratch. */
int dig3d_linear_wgrad_blocks(int M);
/* both gradients of one layer in ONE launch (weight-gradient workers + input-gradient row tiles share the grid) */
int dig3d_linear_bwd_workers(int M, int K, int N);   /* partials dig3d_linear_bwd writes for this shape */
int dig3d_linear_bwd(const float* gY, const float* Z, const float* W, const float* X, int M, int K, int N, int act,
                     float* gX, const float* gx_add, float* part, float* gWb, int reduce_now, void* stream);
/* the same with gz_add [M,N] added to the pre-activation gradient (gZ = gY act'(Z) + gz_add): the final backward of a
 * layer whose pre-activation also received the act'' term of the energy_and_force double backward (run.py:126-133). */
int dig3d_linear_bwd_zadd(const float* gY, const float* Z, const float* W, const float* X, int M, int K, int N, int act,
                          float* gX, const float* gx_add, float* part, float* gWb, int reduce_now, const float* gz_add,
                          void* stream);
/* double backward of the layer w.r.t. an incoming ggx [M,K] in one launch (N > 64): o_gy = (ggx W^T) act'(Z),
 * o_z = (ggx W^T) gy act''(Z), gWb[0:N*K] = (gy act'(Z))^T ggx; part: float[dig3d_linear_dd_workers(M,K,N) * (N*K+N)]. */
int dig3d_linear_dd_workers(int M, int K, int N);
int dig3d_linear_dd(const float* ggx, const float* W, const float* Z, const float* gy, int M, int K, int N, int act,
                    float* o_gy, float* o_z, float* part, float* gWb, int reduce_now, void* stream);
/* reduce_now = 0 (here and in dig3d_linear_bwd_weight / dig3d_smallk_bwd): only the partials are written; the caller
 * reduces the weight gradients of many layers later in ONE launch: */
int dig3d_reduce_many(const void* const* parts, const int* nparts, const int64_t* strides, const int* ns,
                      void* const* outs, int count, void* stream);
int dig3d_reduce_many_acc(const void* const* parts, const int* nparts, const int64_t* strides, const int* ns,
                          void* const* outs, int count, void* stream);   /* outs[d] += ... (second contributions) */
int dig3d_linear_bwd_weight(const float* gY, const float* Z, const float* X, int M, int K, int N, int act,
                            float* part, float* gWb, int reduce_now, void* stream);

/* Forward of a chain of nl <= 8 hidden-width layers on row tiles that stay in LDS:
 *     Y_l = res_l + act_l(Y_{l-1} W_l^T + b_l),  res_l: 0 none, 1 external tensor resext[l], 2 the tile saved by an
 *     earlier layer (save[l'] != 0)            — method/spherenet/spherenet.py:172-182, dimenetpp.py:152-161.
 * All layers have 128 outputs; K[0] <= 128 (multiple of 8), K[l] = 128 afterwards.  W/bias/resext/Z/Y/K/res/save/act
 * are HOST arrays of length nl (device pointers inside); Z[l] / Y[l] receive every layer's pre-activation / output
 * (the backward needs them). */
int dig3d_chain_fwd(const float* X0, int M, int nl, const void* const* W, const void* const* bias,
                    const void* const* resext, void* const* Z, void* const* Y, const int* K, const int* res,
                    const int* save, const int* act, void* stream);

/* The same chain (forward and input-gradient recursion) on weights re-laid in MFMA operand order — csrc/chain.hip: weight
 * slices in registers (v_mfma_f32_16x16x4_f32, transposed product), one barrier per layer, 16-row blocks chosen per launch
 * from the CU count.  dig3d_chain_pack writes Wf / Wb float[nl * 16384] (once per step: the weights change with every
 * optimizer step); dig3d_chainp_fwd / dig3d_chainp_bwd take them in place of W and otherwise have the arguments of
 * dig3d_chain_fwd / dig3d_chain_bwd.  Activations: none or swish (spherenet.py:34-50,172-182 use swish only).
 * dig3d_chain_pack: W[l] [N[l], K[l]] row-major, N[l] <= 128 (multiple of 16; N == NULL: 128), missing rows / columns zero;
 * nl <= 64, so one launch packs every chain and front of a model forward.
 * Replaces: the same reference lines as dig3d_chain_fwd (method/spherenet/spherenet.py:172-182, dimenetpp.py:152-161). */
int dig3d_chain_pack(int nl, const void* const* W, const int* K, const int* N, float* Wf, float* Wb, void* stream);
int dig3d_chainp_fwd(const float* X0, int M, int nl, const float* Wf, const void* const* bias, const void* const* resext,
                     void* const* Z, void* const* Y, const int* K, const int* res, const int* save, const int* act,
                     void* stream);
/* dig3d_chain_dd (the second-order pass of energy_and_force, see there) on packed weights: Wf in place of W. */
int dig3d_chainp_dd(const float* H0, int M, int nl, const float* Wf, const void* const* Z0, const void* const* G0,
                    const void* const* ggres, void* const* HZ, void* const* U, const int* K, const int* res, const int* save,
                    const int* act, void* stream);
int dig3d_chainp_bwd(const float* gout, int M, int nl, const float* Wb, const void* const* Z, void* const* GZ,
                     void* const* gres, const int* K, const int* res, const int* save, const int* act, float* gx0,
                     void* const* G, const void* const* gz_add, void* stream);

/* The FRONT of an interaction block — x_ji = swish(lin_ji(x1)), t = swish(lin_kj(x1)) * rb, xd = swish(lin_down(t)) with
 * rb = lin_rbf2(lin_rbf1(rbf)) — as ONE launch per pass (csrc/chain.hip): forward a 3-layer program of the chain kernel
 * (both branches read the x1 tile; the product is an epilogue operand; lin_down has ND = int_emb_size outputs), backward
 * k_front_bwd (three input-gradient products on one row tile; gadd0 / gadd1: the other gradients that reach x1 — skip
 * connection, readout — or NULL).  Wf / Wb float[3 * 16384] = dig3d_chain_pack of (lin_ji.weight, lin_kj.weight,
 * lin_down.weight) with N = (128, 128, ND).  Weight gradients: dig3d_chain_wgrad_n over GZ = (GZji, GZkj, GZd),
 * X = (x1, x1, T), N = (128, 128, ND).  Replaces method/spherenet/spherenet.py:150-163, dimenetpp.py:130-145 and their autograd. */
int dig3d_front_fwd(const float* x1, int M, const float* Wf, const float* b_ji, const float* b_kj, const float* rb,
                    float* Zji, float* Xji, float* Zkj, float* T, float* Zd, float* Xd, int ND, void* stream);
int dig3d_front_bwd(int M, const float* Wb, const float* Zd, const float* Zkj, const float* Zji, const float* rb,
                    const float* gxd, const float* gxji, const float* gadd0, const float* gadd1, float* GZd, float* GZkj,
                    float* GZji, float* grb, float* gx1, int ND, float* Gm, const float* gzaddD, const float* gzaddKj,
                    const float* gzaddJi, void* stream);
/* The same with grb += grb_add [M,128] (NULL: none) inside the launch — the second gradient that reaches rb in the final pass of
 * method/run.py:126-133 (it comes from the front's own create_graph backward). */
int dig3d_front_bwd_add(int M, const float* Wb, const float* Zd, const float* Zkj, const float* Zji, const float* rb,
                        const float* gxd, const float* gxji, const float* gadd0, const float* gadd1, float* GZd, float* GZkj,
                        float* GZji, float* grb, float* gx1, int ND, float* Gm, const float* gzaddD, const float* gzaddKj,
                        const float* gzaddJi, const float* grb_add, void* stream);
/* energy_and_force (method/run.py:126-131: the force is a gradient and the loss differentiates through it) — the front closed
 * under differentiation on three launches.  dig3d_front_bwd's optional arguments (NULL on the energy route): Gm [M,128]
 * receives the gradient that reached the product t = swish(z_kj) * rb; gzaddD [M,ND] / gzaddKj / gzaddJi [M,128] are added to
 * the pre-activation gradients (the act'' terms the second-order pass sent to the pre-activations).
 * dig3d_front_dd = the backward of dig3d_front_bwd w.r.t. (gxji, gxd, rb, Zji, Zkj, Zd), the forward's three products on the
 * tile U = gradient w.r.t. gx1 (V = gradient w.r.t. grb, or NULL):
 *   dgxji = (U Wji^T) a'(Zji),  HZji = (U Wji^T) gxji a''(Zji);   with t = U Wkj^T, gm = Gm:
 *   HZkj = t gm rb a''(Zkj) + V gm a'(Zkj),  drb = t a'(Zkj) gm,  Cgm = t a'(Zkj) rb + V swish(Zkj);
 *   dgxd = (Cgm Wd^T) a'(Zd),  HZd = (Cgm Wd^T) gxd a''(Zd).
 * Weight gradients of that pass: dig3d_chain_wgrad_n over the GZ of dig3d_front_bwd with X = (U, U, Cgm). */
int dig3d_front_dd(const float* U, const float* V, int M, const float* Wf, const float* Zji, const float* Zkj, const float* Zd,
                   const float* rb, const float* gxji, const float* gxd, const float* Gm, float* dgxji, float* HZji,
                   float* HZkj, float* drb, float* Cgm, float* dgxd, float* HZd, int ND, void* stream);

/* Backward of that chain in two launches.  dig3d_chain_bwd: the input-gradient recursion (layers in reverse order, the
 * gradient tile and the skip accumulator stay in LDS): GZ[l] [M,128] receives g_l * act'(Z[l]) for every layer, gres[l]
 * [M,128] the gradient of layer l's external residual (res[l] == 1; NULL elsewhere), gx0 [M,K[0]] the gradient of the
 * chain input.  dig3d_chain_wgrad: gW_l = GZ[l]^T X[l] (X[0] = the chain input, X[l] = Y[l-1]) and the bias column sums
 * of all layers in one launch; part[l] float[workers * (128*K[l] + 128)] with workers =
 * dig3d_chain_wgrad_workers(M, nl); gWb[l] float[128*K[l] + 128] is written when reduce_now, else the caller reduces the
 * partials later (dig3d_reduce_many).  Replaces the per-layer dig3d_linear_bwd sweep of _Chain.backward
 * (reference: the autograd backward of spherenet.py:172-182). */
int dig3d_chain_bwd(const float* gout, int M, int nl, const void* const* W, const void* const* Z, void* const* GZ,
                    void* const* gres, const int* K, const int* res, const int* save, const int* act, float* gx0,
                    void* const* G, const void* const* gz_add, void* stream);
/* optional (NULL, or length-nl arrays with NULL entries): G[l] [M,128] receives the total gradient w.r.t. layer l's
 * output, gz_add[l] [M,128] is added to the pre-activation gradient.
 * dig3d_chain_dd: the backward of dig3d_chain_bwd (energy_and_force, method/run.py:126-131: the force is a gradient and
 * the loss differentiates through it), one launch, forward layer order: H0 [M,K[0]] = gradient w.r.t. gx0, ggres[l]
 * (or NULL) = gradient w.r.t. gres[l]; Z0[l] / G0[l] = the saved pre-activation / the G[l] written by dig3d_chain_bwd.
 * U[l] [M,128] receives the gradient w.r.t. layer l's total gradient (U[nl-1]: w.r.t. gout; U[l-1], H0 for l = 0: the
 * X operand of dig3d_chain_wgrad for this pass), HZ[l] [M,128] the gradient w.r.t. Z[l]. */
int dig3d_chain_dd(const float* H0, int M, int nl, const void* const* W, const void* const* Z0, const void* const* G0,
                   const void* const* ggres, void* const* HZ, void* const* U, const int* K, const int* res,
                   const int* save, const int* act, void* stream);
int dig3d_chain_wgrad_workers(int M, int nl);
int dig3d_chain_wgrad(int nl, const void* const* GZ, const void* const* X, const int* K, int M, void* const* part,
                      void* const* gWb, int reduce_now, void* stream);
int dig3d_chain_wgrad_n(int nl, const void* const* GZ, const void* const* X, const int* K, const int* N, int M,
                        void* const* part, void* const* gWb, int reduce_now, void* stream);
/* Weight-gradient PARTIALS of many dense layers in one launch (a few beyond 64 tiles of 128 x 128): the dense layers of a
 * whole backward pass.  GY[l] [M[l], N[l]] is the gradient w.r.t. the layer output when Z[l] (pre-activation) and act[l]
 * are given, else already the pre-activation gradient (Z or act NULL); X[l] [M[l], K[l]]; N, K multiples of 4.  part[l]
 * float[nworkers[l] * (N[l]*K[l] + N[l])] (nworkers[l] row-chunk workers per 128 x 128 tile of layer l); reduce with
 * dig3d_reduce_many.  Same arithmetic per layer as dig3d_linear_bwd_weight
 * (autograd of F.linear: spherenet.py:150-216, comenet.py:87-215, schnet.py:29-59).  route 0 / 1: one or two staging
 * buffers per block (two: one barrier per 32-row chunk instead of two); the partials are bit-identical. */
int dig3d_wgrad_many(int nl, const void* const* GY, const void* const* Z, const int* act, const void* const* X,
                     const int* K, const int* N, const int* M, const int* nworkers, void* const* part, int route,
                     void* stream);

/* torch.optim.Adam step (method/run.py:50,133) on FLAT buffers: one elementwise pass over all parameters.
 * n % 4 == 0.  The scalar hyper-parameters travel as DOUBLE and are combined in double exactly as torch.optim.Adam
 * combines its Python floats — 1 - beta2, lr / bias_correction1, sqrt(bias_correction2) — before the single rounding to
 * float32: `1.0f - 0.999f` is 1.3e-5 off 0.001, and that factor scales the second moment of every step (r04: it was the
 * whole excess of the 30-step trajectory distance over the reference's own float32 noise).
 * bias_correction{1,2} = 1 - beta{1,2}^step computed by the host. */
int dig3d_adam_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                    double beta1, double beta2, double eps, double weight_decay, double bias_correction1,
                    double bias_correction2, void* stream);

/* Small-K layers (K <= 16, N <= 256, N % 8 == 0): the radial-basis projections lin_rbf*(rbf) of
 * method/spherenet/spherenet.py:86-90,153-155,182 (K = num_radial or basis_emb_size).  Same semantics as
 * dig3d_linear_fwd / dig3d_linear_bwd; gX or gWb may be NULL.  part: float[dig3d_smallk_blocks(M) * (N*K + N)]. */
int dig3d_smallk_supported(int K, int N);
int dig3d_smallk_blocks(int M);
int dig3d_smallk_fwd(const float* X, const float* W, const float* bias, const float* res, int M, int K, int N, int act,
                     float* Y, float* Z, void* stream);
int dig3d_smallk_bwd(const float* gY, const float* Z, const float* W, const float* X, int M, int K, int N, int act,
                     float* gX, const float* gx_add, float* part, float* gWb, int reduce_now, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * First- and second-order derivatives of the geometry / basis pipeline (diffgeom.hip): what
 * `force = -grad(out, pos, create_graph=True)` followed by `loss.backward()` (method/run.py:126-133) needs from
 * utils/geometric_computing.py:25,44-75 (dist, angle, torsion) and method/spherenet/features.py:151-263
 * (dist_emb, Bessel x harmonics).  "order 1" = vector-Jacobian product, "order 2" = (J w, g^T H w) for an incoming
 * direction w.  vec[e] = pos[i] - pos[j] (edge j->i) is produced by the (linear) row gathers.
 * ------------------------------------------------------------------------------------------------- */
/* |vec[e]| in the reference's float32 operation order (mode as dig3d_edge_dist); rows >= *cnt get `pad`. */
int dig3d_vec_len(const float* vec, int E, int mode, float* dist, const int* cnt, float pad, void* stream);
/* key[t] = edge whose vector is the third argument of torsion[t] (the scatter-min winner of
 * geometric_computing.py:75), a dummy value in [E, dig3d_torsion_key_segments(E)) when there is none or it is the
 * triplet's own k; val maps CSR positions to edge ids (NULL = identity). */
int dig3d_torsion_key(const int* targ, const int* kj, const int* val, int T, int E, int* key, const int* cnt,
                      void* stream);
int dig3d_torsion_key_segments(int E);   /* number of key values (E real edges + the spread dummy segments) */
/* per-triplet pass: gv1/gv2/gv3 [T,3] = (order 1: ggvec NULL) g_angle grad(angle) + g_tor grad(torsion) w.r.t.
 * (v1, v2, v3) = (vec[ji], -vec[kj], -vec[key]); (order 2) the Hessian-vector products for w gathered from ggvec, plus
 * o_ga / o_gt [T] = grad . w.  key NULL: no torsion (gv3 unused). */
int dig3d_tripgeom_grad(const float* vec, const float* ggvec, const int* ji, const int* kj, const int* key, int T,
                        int E, const float* g_angle, const float* g_tor, float* gv1, float* gv2, float* gv3,
                        float* o_ga, float* o_gt, const int* cnt, void* stream);
/* out[e] = (dist term) + sum_{seg_ji(e)} gv1 - sum_{kj = e} gv2 - sum_{key = e} gv3 over CSR segments (tptr; kptr2/perm2;
 * kptr3/perm3), any triplet array may be NULL.  order 2 (ggvec != NULL) also writes o_gd[e] = u . ggvec[e]. */
int dig3d_edge_combine(const float* vec, const float* ggvec, const float* g_dist, int E, const int* tptr,
                       const float* gv1, const int* kptr2, const int* perm2, const float* gv2, const int* kptr3,
                       const int* perm3, const float* gv3, float* out, float* o_gd, const int* cnt, void* stream);
/* Bessel basis (dig3d_bessel_basis): order 1 (gg_d NULL) o_d[e] = sum_k g[e,k] f_k'(d); order 2 o_g[e,k] = gg_d f_k',
 * o_d[e] = gg_d sum_k g[e,k] f_k''. */
int dig3d_bessel_grad(const float* dist, int E, float cutoff, int ns, int nr, const double* zeros, const double* norms,
                      int envelope_p, const float* g, const float* gg_d, float* o_d, float* o_g, const int* cnt,
                      void* stream);
/* dist_emb (method/spherenet/features.py:151-182): rbf[e,n] = Envelope(d/c) sin(freq[n] d/c), p = exponent + 1;
 * gradients w.r.t. d and the learnable freq (block partials + deterministic column sum; reduce_now = 0 leaves the sum of
 * the dig3d_distemb_blocks(E) partial rows of nr to the caller's dig3d_reduce_many). */
int dig3d_distemb_fwd(const float* dist, const float* freq, int E, int nr, float cutoff, int p, float* out,
                      const int* cnt, void* stream);
/* dig3d_edge_dist + dig3d_distemb_fwd + dig3d_bessel_basis as ONE launch (the model front of the energy route:
 * spherenet.py:309-311 xyz_to_dat's dist, then emb's dist_emb and the Bessel table of the angle / torsion embeddings):
 * dist[E], rbf[E, nrd] (rows >= *cnt zero), bes[E, ns*nr]; the same arithmetic as the three entry points, bit for bit. */
int dig3d_edge_front(const float* pos, const int* src, const int* dst, int E, int mode, const int* cnt, float pad,
                     float* dist, const float* freq, int nrd, float cutoff_d, int p, float* rbf, float cutoff_b, int ns,
                     int nr, const double* zeros, const double* norms, int envelope_p, float* bes, void* stream);
int dig3d_distemb_blocks(int E);
int dig3d_distemb_grad(const float* dist, const float* freq, int E, int nr, float cutoff, int p, const float* g,
                       const float* gg_d, const float* gg_f, int order, float* o_d, float* o_g, float* part,
                       float* o_f, const int* cnt, int reduce_now, void* stream);
/* real spherical harmonics table Y[M, ns] (phi NULL) or Y[M, ns*ns] in the order of dig3d_sph_basis, and its
 * derivatives w.r.t. (theta, phi). */
int dig3d_harmonics_fwd(const float* theta, const float* phi, int M, int ns, const float* pref, float* out,
                        const int* cnt, void* stream);
int dig3d_harmonics_grad(const float* theta, const float* phi, int M, int ns, const float* pref, const float* g,
                         const float* gg_th, const float* gg_ph, int order, float* o_th, float* o_ph, float* o_g,
                         const int* cnt, void* stream);
/* activation pieces of the twice-differentiable dense layer: o_gy = t act'(z), o_z = t gy act''(z);
 * out = gy act'(z) + gz. */
int dig3d_act_bwd2(const float* t, const float* gy, const float* z, int64_t n, int act, float* o_gy, float* o_z,
                   void* stream);
int dig3d_preact_merge(const float* gy, const float* z, const float* gz, int64_t n, int act, float* out,
                       void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Output blocks of all layers, batched (readout.hip, dense.hip) — method/spherenet/spherenet.py:185-225,
 * dimenetpp.py:164-204: the L+1 `update_v` / `update_u` blocks of a forward are independent; each stage of all
 * of them is ONE launch.  Every array argument is a HOST array of G <= 8 device pointers.
 * ------------------------------------------------------------------------------------------------- 
 * act = 7 (row scale, as dig3d_linear_fwd_rowscale): Y_g = rs_g[m] * (x_g W_g^T + b_g) with rs_g [M] in the res slot, Z_g [M,N]
 * receives rs_g broadcast — the backward entries then take act = 3 with that Z. */
int dig3d_linear_fwd_grouped(int G, const void* const* X, const void* const* W, const void* const* bias,
                             const void* const* res, int M, int K, int N, int act, void* const* Y, void* const* Z,
                             void* stream);
/* part[g]: float[dig3d_linear_wgrad_blocks(M) * (N*K+N)]; gX == NULL: no input gradient is wanted (its tiles are not launched) */
int dig3d_linear_bwd_grouped(int G, const void* const* gY, const void* const* Z, const void* const* W,
                             const void* const* X, int M, int K, int N, int act, void* const* gX,
                             const void* const* gx_add, void* const* part, void* const* gWb, int reduce_now,
                             const void* const* gz_add, void* stream);
/* grouped forms of dig3d_linear_bwd_input and dig3d_linear_dd (the energy_and_force route of the output blocks) */
int dig3d_linear_bwd_input_grouped(int G, const void* const* gY, const void* const* Z, const void* const* W, int M, int K,
                                   int N, int act, void* const* gX, void* stream);
int dig3d_linear_dd_grouped(int G, const void* const* ggx, const void* const* W, const void* const* Z,
                            const void* const* gy, int M, int K, int N, int act, void* const* o_gy, void* const* o_z,
                            void* const* part, void* const* gWb, int reduce_now, void* stream);
/* out_g = segment sums of in_g (* mul_g): with mul, e2 = lin_rbf(rbf) * e1 (spherenet.py:90,182) is never written */
int dig3d_segment_sum_grouped(int G, const void* const* in, const void* const* mul, const int* kptr, int S, int C,
                              void* const* out, void* stream);
/* The same with a second product per group, out_g = segment sums of in_g * mul_g + in2_g * mul2_g (NULL: none) — the double
 * backward of the product form on the energy_and_force route (method/run.py:126-133 over dimenetpp.py:160,176). */
int dig3d_segment_sum_grouped2(int G, const void* const* in, const void* const* mul, const void* const* in2,
                               const void* const* mul2, const int* kptr, int S, int C, void* const* out, void* stream);
/* its backward: out_g = in_g[ix] (* mul_g), out2_g = in_g[ix] * mul2_g (both factor gradients in one pass) */
int dig3d_gather_grouped(int G, const void* const* in, const int* ix, int64_t M, int C, void* const* out,
                         const void* const* mul, void* const* out2, const void* const* mul2, const int* cnt,
                         void* stream);
/* The same with addends, out_g += add_g and out2_g += add2_g ([M, C]; NULL: none), inside the launch (see
 * dig3d_triplet_fwd_add). */
int dig3d_gather_grouped_add(int G, const void* const* in, const int* ix, int64_t M, int C, void* const* out,
                             const void* const* mul, void* const* out2, const void* const* mul2, const void* const* add,
                             const void* const* add2, const int* cnt, void* stream);
int dig3d_smalln_fwd_grouped(int G, const void* const* X, const void* const* W, const void* const* bias, int M, int K,
                             int N, void* const* Y, void* stream);
int dig3d_smalln_blocks(int M);
int dig3d_smalln_bwd_grouped(int G, const void* const* gY, const void* const* W, const void* const* X, int M, int K,
                             int N, void* const* gX, void* const* part, void* stream);
int dig3d_graph_sum_grouped(int G, const void* const* Y, const int* ptr, int B, int C, float* u, void* stream);

/* L1 loss (method/run.py:127 with torch.nn.L1Loss(): mean |out - y|, out [B,C] and y broadcast to it by the caller) and its
 * gradient: dig3d_l1_loss_fwd writes loss[1] and sgn[n] = sign(out - y) / n (0 at 0, as torch.sgn); the gradient w.r.t.
 * out is sgn scaled by the incoming scalar gradient, read from device memory (dig3d_scale_by_scalar).  Replaces 8 framework
 * launches per step.  seed / g (both or neither): a backward seed known at forward time — a device scalar, e.g. 1 / world of
 * a captured data-parallel step — g[n] = sgn * seed[0] is then written by the same launch. */
int dig3d_l1_loss_fwd(const float* out, const float* y, int n, float* loss, float* sgn, const float* seed, float* g,
                      void* stream);
int dig3d_scale_by_scalar(const float* v, const float* scalar, int n, float* g, void* stream);

/* The energy_and_force loss of method/run.py:126-131 with torch.nn.L1Loss() in one launch:
 *   loss = mean_i |out_i - y_i| + p * sum_j |-gpos_j - f_j| / (3 * *cntN),   gpos = d sum(out) / d pos  [n3 = 3 N floats]
 * (force = -gpos, run.py:126; cntN NULL: all n3 entries live; a padded static-shape batch: only the first 3 * *cntN).
 * sgn_e [nE] / sgn_f [n3] receive d loss / d out and d loss / d gpos (the backward multiplies them by its incoming scalar:
 * dig3d_scale_by_scalar); seed / g_out / g_gpos (all or none): the backward seed is already known — a captured step's device
 * scalar — and the two gradients are written by this launch. */
int dig3d_ef_l1_loss(const float* out, const float* y, int nE, const float* gpos, const float* f, int n3, const int* cntN,
                     float p, const float* seed, float* loss, float* sgn_e, float* sgn_f, float* g_out, float* g_gpos,
                     void* stream);

/* dst[rd, cd] = src[rs, cs] zero-padded / sliced (dst[r][c] = src[r][c] where both exist, 0 elsewhere): the copy around the
 * MFMA kernels for layer widths that are not multiples of 8 — method/spherenet/spherenet.py:253-259 accepts any
 * hidden_channels / int_emb_size, the kernels want output widths that are multiples of 8 (dig_amd/ops.py:linear). */
int dig3d_pad2d(const float* src, int rs, int cs, float* dst, int rd, int cd, void* stream);

/* Column groups of 8 of a row-major matrix <-> one contiguous [T, 8] matrix per group: outs[l][t][b] = in[t][8 l + b],
 * l < L <= 8, and the adjoint (ins[l] NULL: a zero block).  The first basis Linears lin_sbf1 of ALL interaction blocks
 * (method/dimenetpp/dimenetpp.py:146) applied as ONE layer with stacked weights in the energy_and_force route; each block's
 * slice is the [T, 8] operand of dig3d_triplet_fwd / _bwd. */
int dig3d_cols_split8(const float* in, int64_t T, int L, void* const* outs, void* stream);
int dig3d_cols_merge8(const void* const* ins, int64_t T, int L, float* out, void* stream);

/* out[n] = 0; out[arg[s]] = g[s] for s < S with 0 <= arg[s] < n (arg unique among those; arg == n is torch_scatter's
 * "empty segment" sentinel) — gradient of scatter_min w.r.t. its source (method/comenet/comenet.py:304-327). */
int dig3d_scatter_unique(const float* g, const int64_t* arg, int S, int n, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * All radial-basis projections of a forward in one launch (radial.hip) — method/spherenet/spherenet.py:86-90
 * (lin_rbf_0 + swish, lin_rbf_1), :153-155 (lin_rbf2(lin_rbf1(rbf))), :182 (lin_rbf): H <= 16 "heads" over the same
 * rbf [M, K <= 8].  Host arrays of H entries: Wb[h] NULL = single layer (Wa [N,K], bias or NULL, act 0/1 = none/swish),
 * else two-layer (Wa [J,K], Wb [N,J], J <= 8).  Backward: gX (sum over heads) and dig3d_radial_blocks(M, H) partial rows of
 * every weight gradient (reduce with dig3d_reduce_many; layout in radial.hip).
 * ------------------------------------------------------------------------------------------------- */
int dig3d_radial_partial_stride(int H, const int* N, const int* J, const int* two_layer, int K);
int dig3d_radial_blocks(int M, int H);
int dig3d_radial_fwd(const float* X, int M, int K, int H, const void* const* Wa, const void* const* Wb,
                     const void* const* bias, const int* N, const int* J, const int* act, void* const* Y,
                     void* stream);
/* gx_work (float[dig3d_radial_bwd_groups(H) * M * K], required when H > 1 and gX != NULL): every head writes its gX share
 * to its own slice, summed in a fixed order by a second tiny launch. */
int dig3d_radial_bwd_groups(int H);
int dig3d_radial_bwd(const float* X, int M, int K, int H, const void* const* Wa, const void* const* Wb,
                     const void* const* bias, const int* N, const int* J, const int* act, const void* const* gY,
                     float* gX, float* part, float* gx_work, void* stream);

/* out = ((in[0] + in[1]) + in[2]) + ... over n <= 16 equally shaped float arrays (host array of device pointers), one launch,
 * fixed order: the accumulation of the gradients that reach one tensor from its n consumers — autograd's n - 1 additions
 * per backward pass of method/run.py:126-133 (rbf of method/dimenetpp/dimenetpp.py:55-78,130-160 has 2 + 2 L consumers). */
int dig3d_sum_many(const void* const* in, int n, int64_t numel, float* out, void* stream);

/* y = a * b, twice differentiable (diffgeom.hip): the elementwise products of the energy_and_force route — x_kj * radial
 * projection and e2 = lin_rbf(rbf) * e1 (method/spherenet/spherenet.py:90,155,182; dimenetpp.py:77,137,160) under
 * run.py:126-133's double backward.  _bwd: ga = g b (+ adda), gb = g a (+ addb); g NULL = zeros; adda / addb (optional): the
 * gradients that reached a and b through the step's other graph, folded in instead of one framework addition each.  _bwd2: the backward of that pair for incoming (gga, ggb)
 * (either may be NULL = zeros): og = gga b + ggb a, oa = ggb g, ob = gga g.  n elements, any shape. */
int dig3d_ew_mul(const float* a, const float* b, float* y, int64_t n, void* stream);
int dig3d_ew_mul_bwd(const float* g, const float* a, const float* b, float* ga, float* gb, int64_t n, const float* adda,
                     const float* addb, void* stream);
int dig3d_ew_mul_bwd2(const float* gga, const float* ggb, const float* g, const float* a, const float* b, float* og,
                      float* oa, float* ob, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * 256-wide layer chains (wide.hip): the output blocks of all interaction layers of SphereNet / DimeNet++
 * (method/spherenet/spherenet.py:185-216, dimenetpp.py:164-204: lin_up 128 -> 256, then lins 256 -> 256 with swish) as G groups
 * of ONE launch per pass, and ComENet's residual layers h = h + swish(lin(h)) (method/comenet/comenet.py:209-210).
 *   Y_l = res_l * Y_{l-1} + act_l(Y_{l-1} W_l^T + b_l),  l < nl <= 4, 256 outputs, K_0 in {128, 256}, K_l = 256 afterwards.
 * dig3d_wide_pack re-lays n <= 32 weights [256, K] in MFMA operand order (fwd[i]: 256*K floats; bwd[i]: 65536 floats).
 * Host arrays, group-major: X0[G], gout[G], gx0[G], gadd[G] (or NULL); Wp / bias / Z / Y / GZ [G * nl]; act (0 none, 1 swish)
 * and res (1: add the layer's input) [nl].  The backward writes GZ_l = g_l * act'(Z_l) (operands of the weight gradients,
 * dig3d_wgrad_many) and the input gradient gx0 (+ gadd).
 * ------------------------------------------------------------------------------------------------- */
int dig3d_wide_supported(int M, int K0, int nl, int G);
int dig3d_wide_pack(int n, const void* const* W, const int* K, void* const* fwd, void* const* bwd, void* stream);
int dig3d_wide_fwd(int G, int nl, int M, int K0, const void* const* X0, const void* const* Wp, const void* const* bias,
                   void* const* Z, void* const* Y, const int* act, const int* res, void* stream);
int dig3d_wide_bwd(int G, int nl, int M, int K0, const void* const* gout, const void* const* Wp, const void* const* Z,
                   void* const* GZ, void* const* gx0, const void* const* gadd, const int* act, const int* res,
                   void* const* Gout, const void* const* gzadd, void* stream);
/* energy_and_force (method/run.py:126-131) — the chain closed under differentiation on the same kernels.  dig3d_wide_bwd's
 * optional arrays [G * nl] (NULL on the energy route): Gout[i] receives the total gradient w.r.t. layer i's output, gzadd[i] is
 * added to its pre-activation gradient.  dig3d_wide_dd = the backward of dig3d_wide_bwd w.r.t. (gout, Z): the forward's
 * products in the forward's layer order on H0[g] [M,K0] = gradient w.r.t. gx0[g] (Wp: packed FORWARD slices):
 *   t_l = U_{l-1} W_l^T (U_{-1} = H0),   U_l = t_l act'(Z0_l) + res_l U_{l-1},   HZ_l = t_l G0_l act''(Z0_l)
 * Z0 / G0 [G * nl] = saved pre-activations / the Gout of dig3d_wide_bwd (NULL entries where act = 0: U_l = t_l, HZ[i] NULL).
 * U[.., nl-1] is the gradient w.r.t. gout; the weight gradients of the pass are GZ_l^T U_{l-1} (dig3d_wgrad_many). */
int dig3d_wide_dd(int G, int nl, int M, int K0, const void* const* H0, const void* const* Wp, const void* const* Z0,
                  const void* const* G0, void* const* HZ, void* const* U, const int* act, const int* res, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Library identity (abi.hip).  No reference counterpart: the reference reaches its kernels through Python wheels
 * (torch_scatter / torch_cluster / torch_sparse, imported at method/spherenet/spherenet.py:1-20) whose version check is
 * pip's; a ctypes binding has none, so the host compares dig3d_abi_hash with the hash of the header it parsed.
 *   dig3d_abi_hash   out[cap] <- NUL-terminated hex digest of csrc/ + include/dig3d.h at build time; returns its length
 *   dig3d_device_info  host int info[8]: CUs used by the worker heuristics, wavefront size, XCDs, LDS bytes per
 *                    workgroup, device ordinal, clock kHz, total HBM bytes (low / high 32 bits)
 * ------------------------------------------------------------------------------------------------- */
int dig3d_abi_hash(char* out, int cap);
int dig3d_device_info(int* info);

#ifdef __cplusplus
}
#endif
#endif /* DIG3D_H */
