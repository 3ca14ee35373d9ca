// dig3d graph construction: radius graph, CSR, triplet lists, transposed (sort-by-key) CSRs.
// All integer work, bit-exact against the reference:
//   radius_graph      torch_cluster semantics at spherenet.py:304 / dimenetpp.py:277 / schnet.py:156 /
//                     comenet.py:294  (CUDA ordering rule)
//   triplets          utils/geometric_computing.py:27-41
// Layout: edges are produced grouped by target (ascending), sources ascending inside a target, so
// the edge list IS the CSR of the (target <- source) adjacency and every forward reduction in the
// models is a contiguous segment reduction.  Internal indices are int32; int64 only at the API.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// graph pointer from the sorted batch vector:  ptr[g] = first node of graph g, ptr[B] = N.
// meta[0] = B, meta[7] |= 1 if batch is not sorted.
// ONE: the whole batch vector by a single workgroup, which zeroes meta[0..8) itself before the first error bit can be set (no
// separate zero-fill launch in front of the graph build: ~4.7 us of launch floor per step at the reference's batch sizes).
template <bool ONE>
__global__ void __launch_bounds__(1024) k_graph_ptr(const int64_t* __restrict__ batch, int N, int* __restrict__ ptr,
                                                     int64_t* __restrict__ meta, int* __restrict__ batch32) {
  if (ONE) {
    if (threadIdx.x < 8) meta[threadIdx.x] = 0;
    __syncthreads();
  }
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += ONE ? (int)blockDim.x : N) {
    int64_t b = batch[n];
    if (batch32) batch32[n] = (int)b;       // the int32 copy every segment kernel reads (was a framework cast kernel)
    int64_t prev = n > 0 ? batch[n - 1] : -1;
    if (b < prev) atomicOr((unsigned long long*)&meta[7], 1ull);
    if (b < 0 || b >= N || prev >= N) {  // graph ids must lie in [0, N): ptr has N+2 slots
      atomicOr((unsigned long long*)&meta[7], 2ull);
      continue;
    }
    for (int64_t q = prev + 1; q <= b; ++q) ptr[q] = n;
    if (n == N - 1) {
      ptr[b + 1] = N;
      meta[0] = b + 1;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// radius graph: one wavefront per target node.  Sources of the same graph are visited in ascending
// index, 64 per step; __ballot gives the in-radius mask, popcount of the lower lanes the rank.
// Rule (torch_cluster.radius + radius_graph): collect up to `cap` in-radius points INCLUDING the
// target itself (cap = max_num_neighbors+1 when loop==0), then drop the self pair.
// d2 is accumulated in float32 x,y,z order without FMA contraction, strict `<` against r*r.
__global__ void k_radius(const float* __restrict__ pos, const int64_t* __restrict__ batch,
                         const int* __restrict__ ptr, int N, float r, int cap, int loop, int width,
                         int* __restrict__ nbr, int* __restrict__ deg) {
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (wave >= N) return;
  int i = wave;
  int g = (int)batch[i];
  int s = ptr[g], e = ptr[g + 1];
  float r2 = r * r;
  f3 pi = load3(pos, i);
  int taken = 0, outc = 0;
  for (int base = s; base < e && taken < cap; base += 64) {
    int src = base + lane;
    bool hit = false;
    if (src < e) {
      f3 ps = load3(pos, src);
      float dx = ps.x - pi.x, dy = ps.y - pi.y, dz = ps.z - pi.z;
      float d2 = (dx * dx + dy * dy) + dz * dz;
      hit = d2 < r2;
    }
    uint64_t m = __ballot(hit);
    int rank = taken + __popcll(m & lanemask_lt());
    bool take = hit && rank < cap;
    bool keep = take && (loop || src != i);
    uint64_t km = __ballot(keep);
    if (keep) nbr[(int64_t)i * width + outc + __popcll(km & lanemask_lt())] = src;
    taken += __popcll(m);
    outc += __popcll(km);
  }
  if (lane == 0) deg[i] = outc;
}

// ------------------------------------------------------------------------------------------------
// exclusive scan, int32.  out has n+1 entries (out[n] = total); total is also stored to *total_out
// (int64) when non-null.  n may live on the device (n_dev != nullptr => n = min(n_max, *n_dev)).
// Single block for n <= 32768 (the common case: N ~ 600 nodes, E ~ 10^4 edges), else 3 kernels.
#define SCAN_T 1024
__device__ __forceinline__ int block_excl_scan(int v, int* sh, int* total) {
  // sh: SCAN_T/64 ints.  returns exclusive prefix of v across the block.
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int y = __shfl_up(x, o);
    if (lane >= o) x += y;
  }
  if (lane == 63) sh[w] = x;
  __syncthreads();
  if (w == 0) {
    int t = lane < (int)(blockDim.x >> 6) ? sh[lane] : 0;
    int xs = t;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      int y = __shfl_up(xs, o);
      if (lane >= o) xs += y;
    }
    if (lane < (int)(blockDim.x >> 6)) sh[lane] = xs - t;  // exclusive wave offsets
    if (lane == (int)(blockDim.x >> 6) - 1) *total = xs;
  }
  __syncthreads();
  return sh[w] + x - v;
}

// (meta, mirror): the LAST scan of a graph build also hands the batch's sizes to the host — thread 0 copies meta[0..8) into
// ``mirror``, pinned host memory the device writes directly (was a framework device->host copy, one blit kernel per step)
// Called by EVERY thread of the (single) block that wrote the last size, after those writes: a block barrier, then lane q < 8
// stores word q — eight posted writes in flight at once instead of eight system-scope stores in program order by one thread
// (the scan that carries the mirror ran 12.8 us against 4.8 for the one that does not).
__device__ __forceinline__ void mirror_meta(const int64_t* meta, int64_t* mirror) {
  if (!mirror) return;                   // uniform
  __threadfence();
  __syncthreads();
  if (threadIdx.x < 8) {
    __hip_atomic_store(&mirror[threadIdx.x], __hip_atomic_load(&meta[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // (no system-scope fence: the host reads the mirror after an event recorded BEHIND this kernel — the end of the kernel is
    // the release)
  }
}

__global__ void __launch_bounds__(SCAN_T) k_scan_single(const int* __restrict__ in, int* __restrict__ out,
                                                        int n_max, const int64_t* __restrict__ n_dev,
                                                        int64_t* __restrict__ total_out, const int64_t* meta,
                                                        int64_t* mirror) {
  __shared__ int sh[SCAN_T / 64];
  __shared__ int tot;
  int n = n_max;
  if (n_dev) {
    int64_t nd = *n_dev;
    n = nd < n_max ? (int)nd : n_max;
  }
  // tiles of 4 x SCAN_T consecutive entries, four per thread, the running total carried from tile to tile (a thread walking
  // its own run of n / SCAN_T entries read them one dependent, uncoalesced load at a time: 13 us at n = 8 700 against 4.8 us of
  // launch floor at n = 600)
  int carry = 0;
  for (int base = 0; base < n; base += 4 * SCAN_T) {
    const int i0 = base + 4 * threadIdx.x;
    int v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = i0 + k < n ? in[i0 + k] : 0;
    int off = carry + block_excl_scan(v[0] + v[1] + v[2] + v[3], sh, &tot);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i0 + k < n) out[i0 + k] = off;
      off += v[k];
    }
    carry += tot;
    __syncthreads();                     // sh / tot are rewritten by the next tile
  }
  if (threadIdx.x == 0) {
    out[n] = carry;
    if (total_out) *total_out = carry;
  }
  mirror_meta(meta, mirror);
}

#define SCAN_CHUNK 4096  // elements per block in the multi-block path (1024 threads x 4)
__global__ void __launch_bounds__(SCAN_T) k_scan_local(const int* __restrict__ in, int* __restrict__ out,
                                                       int n_max, const int64_t* __restrict__ n_dev,
                                                       int* __restrict__ bsum) {
  __shared__ int sh[SCAN_T / 64];
  __shared__ int tot;
  int n = n_max;
  if (n_dev) {
    int64_t nd = *n_dev;
    n = nd < n_max ? (int)nd : n_max;
  }
  int base = blockIdx.x * SCAN_CHUNK + threadIdx.x * 4;
  int v[4];
  int s = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[q] = base + q < n ? in[base + q] : 0;
    s += v[q];
  }
  int off = block_excl_scan(s, sh, &tot);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (base + q < n) out[base + q] = off;
    off += v[q];
  }
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SCAN_T) k_scan_sums(int* __restrict__ bsum, int nb, int* __restrict__ out,
                                                      int n_max, const int64_t* __restrict__ n_dev,
                                                      int64_t* __restrict__ total_out, const int64_t* meta,
                                                      int64_t* mirror) {
  __shared__ int sh[SCAN_T / 64];
  __shared__ int tot;
  int n = n_max;
  if (n_dev) {
    int64_t nd = *n_dev;
    n = nd < n_max ? (int)nd : n_max;
  }
  int per = (nb + SCAN_T - 1) / SCAN_T;
  int b = threadIdx.x * per;
  int e = b + per < nb ? b + per : nb;
  int s = 0;
  for (int q = b; q < e; ++q) s += bsum[q];
  int off = block_excl_scan(s, sh, &tot);
  for (int q = b; q < e; ++q) {
    int v = bsum[q];
    bsum[q] = off;
    off += v;
  }
  if (threadIdx.x == 0) {
    out[n] = tot;
    if (total_out) *total_out = tot;
  }
  mirror_meta(meta, mirror);
}

__global__ void k_scan_add(int* __restrict__ out, int n_max, const int64_t* __restrict__ n_dev,
                           const int* __restrict__ bsum) {
  int n = n_max;
  if (n_dev) {
    int64_t nd = *n_dev;
    n = nd < n_max ? (int)nd : n_max;
  }
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) out[q] += bsum[q / SCAN_CHUNK];
}

static int scan_i32(const int* in, int* out, int n_max, const int64_t* n_dev, int64_t* total_out,
                    int* ws, hipStream_t st, const int64_t* meta = nullptr, int64_t* mirror = nullptr) {
  if (n_max <= 32768) {
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(SCAN_T), 0, st, in, out, n_max, n_dev, total_out, meta, mirror);
  } else {
    int nb = (n_max + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (!ws) return DIG3D_ERR_ARG;
    hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(SCAN_T), 0, st, in, out, n_max, n_dev, ws);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_T), 0, st, ws, nb, out, n_max, n_dev, total_out, meta, mirror);
    hipLaunchKernelGGL(k_scan_add, dim3(dig3d_blocks(n_max, 256)), dim3(256), 0, st, out, n_max, n_dev, ws);
  }
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// ------------------------------------------------------------------------------------------------
// compact the padded neighbour table into the edge list (CSR by target).
__global__ void k_edges_fill(const int* __restrict__ nbr, const int* __restrict__ deg,
                             const int* __restrict__ rowptr, int N, int width, int* __restrict__ src,
                             int* __restrict__ dst) {
  int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= (int64_t)N * width) return;
  int i = (int)(slot / width), q = (int)(slot - (int64_t)i * width);
  if (q >= deg[i]) return;
  int e = rowptr[i] + q;
  src[e] = nbr[slot];
  dst[e] = i;
}

// ------------------------------------------------------------------------------------------------
// triplets k->j->i (k != i) for every edge e = (j->i): count, then fill.
//   CSR: rowptr[N+1], col[E] (sources of the edges sorted by (target, source)), val[E] = original
//   edge id of each CSR entry (nullptr => identity, i.e. the edge list is already in CSR order).
//   esrc/edst: endpoints of the edges in ORIGINAL order (what idx_ji indexes).
__global__ void k_trip_count(const int* __restrict__ rowptr, const int* __restrict__ col,
                             const int* __restrict__ esrc, const int* __restrict__ edst, int E_max,
                             const int64_t* __restrict__ E_dev, int* __restrict__ cnt) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  int E = E_max;
  if (E_dev) {
    int64_t ed = *E_dev;
    E = ed < E_max ? (int)ed : E_max;
  }
  if (e >= E) return;
  int j = esrc[e], i = edst[e];
  int b = rowptr[j], en = rowptr[j + 1];
  int c = en - b;
  for (int p = b; p < en; ++p)
    if (col[p] == i) --c;
  cnt[e] = c;
}

// TF_LPE = 16 lanes per edge e = (j -> i): lane l looks at the l-th (l + 16-th, ...) incoming edge of j; the triplets of e keep
// the ascending order of those edges (rank = number of kept positions before mine, from the group's ballot bits) — the
// same lists as a thread per edge writing them one after the other (10.3 us at 8.7k edges / 1.0e5 triplets).
#define TF_LPE 16
__global__ void __launch_bounds__(256) k_trip_fill(const int* __restrict__ rowptr, const int* __restrict__ col,
                            const int* __restrict__ val, const int* __restrict__ esrc,
                            const int* __restrict__ edst, const int* __restrict__ tptr, int E,
                            int* __restrict__ kj, int* __restrict__ ji) {
  const int e = (blockIdx.x * blockDim.x + threadIdx.x) / TF_LPE, sub = threadIdx.x & (TF_LPE - 1);
  const bool inb = e < E;                  // (no early return: the ballots below are wave-wide)
  const int j = inb ? esrc[e] : 0, i = inb ? edst[e] : 0;
  int w = inb ? tptr[e] : 0;
  const int b = inb ? rowptr[j] : 0, en = inb ? rowptr[j + 1] : 0;
  const int shift = (threadIdx.x & 63) & ~(TF_LPE - 1);         // first lane of my group inside the wave
  // the longest row of the wave decides the trip count (uniform loop: every lane takes part in every ballot)
  int len = en - b;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int o = __shfl_xor(len, off, 64);
    len = o > len ? o : len;
  }
  for (int base = 0; base < len; base += TF_LPE) {
    const int p = b + base + sub;
    const bool keep = inb && p < en && col[p] != i;
    const uint64_t m = __ballot(keep);
    const unsigned grp = (unsigned)((m >> shift) & ((1u << TF_LPE) - 1u));
    if (keep) {
      const int r = __popc(grp & ((1u << sub) - 1u));
      kj[w + r] = val ? val[p] : p;
      ji[w + r] = e;
    }
    w += __popc(grp);
  }
}

// ------------------------------------------------------------------------------------------------
// transposed CSR of an arbitrary key array (for the backward of row gathers: scatter_add by an
// UNSORTED index becomes a segment sum over perm).  Deterministic: atomics only decide slots inside a
// segment, then every segment is sorted ascending, so perm lists the positions of each key in
// increasing order (== a stable counting sort).
__global__ void k_key_hist(const int* __restrict__ key, int M, int* __restrict__ hist) {
  int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m < M) atomicAdd(&hist[key[m]], 1);
}
__global__ void k_key_fill(const int* __restrict__ key, int M, const int* __restrict__ kptr,
                           int* __restrict__ cursor, int* __restrict__ perm) {
  int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  int k = key[m];
  int slot = kptr[k] + atomicAdd(&cursor[k], 1);
  perm[slot] = m;
}
// Ordering the entries of every segment (their slots came from atomics): rank sort — the rank of an entry is the
// number of smaller entries of its segment (entries are distinct positions), every read independent.  The choice is
// made PER SEGMENT on the device (no host round trip, and one long segment does not slow the short ones down):
//  * k_seg_rank_sort       segments of <= RANK_MAX entries (molecular graphs: 10-33): one thread per ELEMENT;
//  * k_seg_rank_sort_long  longer segments (a scatter onto a handful of keys): one WORKGROUP per segment, the segment
//                          streamed through LDS in tiles, O(len^2 / 256) per thread.
#define RANK_MAX 1024
#define LONG_TILE 2048
__device__ __forceinline__ void seg_rank_sort_long_body(const int* __restrict__ kptr, int S, const int* __restrict__ tmp,
                                                        int* __restrict__ perm, int* tile, int blk, int nblk) {
  for (int s = blk; s < S; s += nblk) {                      // uniform per block
    const int b = kptr[s], e = kptr[s + 1];
    if (e - b <= RANK_MAX) continue;
    for (int a0 = b; a0 < e; a0 += 256 * 4) {                // four entries per thread per sweep
      int v[4], rank[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int a = a0 + threadIdx.x + 256 * u;
        v[u] = a < e ? tmp[a] : 0x7fffffff;
        rank[u] = 0;
      }
      for (int q0 = b; q0 < e; q0 += LONG_TILE) {
        __syncthreads();
        for (int q = threadIdx.x; q < LONG_TILE; q += 256) tile[q] = (q0 + q < e) ? tmp[q0 + q] : 0x7fffffff;
        __syncthreads();
        const int n = (e - q0 < LONG_TILE) ? e - q0 : LONG_TILE;
        for (int q = 0; q < n; ++q) {
          const int w = tile[q];
#pragma unroll
          for (int u = 0; u < 4; ++u) rank[u] += w < v[u];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int a = a0 + threadIdx.x + 256 * u;
        if (a < e) perm[b + rank[u]] = v[u];
      }
    }
  }
}
__global__ void __launch_bounds__(256) k_seg_rank_sort_long(const int* __restrict__ kptr, int S,
                                                             const int* __restrict__ tmp, int* __restrict__ perm) {
  __shared__ int tile[LONG_TILE];
  seg_rank_sort_long_body(kptr, S, tmp, perm, tile, blockIdx.x, gridDim.x);
}

__device__ __forceinline__ void seg_rank_sort_body(const int* __restrict__ key, const int* __restrict__ kptr,
                                                   const int* __restrict__ tmp, int M, int* __restrict__ perm, int p) {
  if (p >= M) return;
  const int v = tmp[p];
  const int s = key[v];
  const int b = kptr[s], e = kptr[s + 1];
  if (e - b > RANK_MAX) return;        // k_seg_rank_sort_long owns this segment
  int rank = 0;
  for (int q = b; q < e; ++q) rank += tmp[q] < v;
  perm[b + rank] = v;
}
__global__ void k_seg_rank_sort(const int* __restrict__ key, const int* __restrict__ kptr,
                                const int* __restrict__ tmp, int M, int* __restrict__ perm) {
  seg_rank_sort_body(key, kptr, tmp, M, perm, blockIdx.x * blockDim.x + threadIdx.x);
}

// ---- the same build for up to CSRS_MAX keys at once (blockIdx.y = which): a replayed step needs the transposed CSRs of the
// next batch's edge sources AND of its triplets' k->j edges, 2 x (memset, histogram, scan, fill, two rank sorts) = 12
// launches of ~5 us each behind the replay; batched they are one memset + four launches.
#define CSRS_MAX 4
struct CsrSet {
  const int* key[CSRS_MAX];
  int* kptr[CSRS_MAX];
  int* perm[CSRS_MAX];
  int* hist[CSRS_MAX];
  int* cursor[CSRS_MAX];
  int* tmp[CSRS_MAX];
  int M[CSRS_MAX], S[CSRS_MAX];
};
__global__ void k_keys_hist(CsrSet t) {
  const int w = blockIdx.y, m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m < t.M[w]) atomicAdd(&t.hist[w][t.key[w][m]], 1);
}
__global__ void __launch_bounds__(SCAN_T) k_keys_scan(CsrSet t) {      // one block per key, S <= 32768 each
  __shared__ int sh[SCAN_T / 64];
  __shared__ int tot;
  const int w = blockIdx.x, n = t.S[w];
  const int* __restrict__ in = t.hist[w];
  int* __restrict__ out = t.kptr[w];
  int carry = 0;                         // tiles of 4 x SCAN_T entries, as k_scan_single
  for (int base = 0; base < n; base += 4 * SCAN_T) {
    const int i0 = base + 4 * threadIdx.x;
    int v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = i0 + k < n ? in[i0 + k] : 0;
    int off = carry + block_excl_scan(v[0] + v[1] + v[2] + v[3], sh, &tot);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i0 + k < n) out[i0 + k] = off;
      off += v[k];
    }
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) out[n] = carry;
}
__global__ void k_keys_fill(CsrSet t) {
  const int w = blockIdx.y, m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= t.M[w]) return;
  const int k = t.key[w][m];
  t.tmp[w][t.kptr[w][k] + atomicAdd(&t.cursor[w][k], 1)] = m;
}

// blocks [0, nshort): one thread per entry (short segments); blocks [nshort, ...): the long segments, one workgroup each
// rezero: the histogram / cursor words of this key (nobody reads them after k_keys_fill) are left ZERO for the next call on the
// same workspace, which then needs no zero-fill launch in front of its histogram (dig3d_csr_by_keys_ws)
__global__ void __launch_bounds__(256) k_keys_rank_sort(CsrSet t, int nshort, int rezero) {
  __shared__ int tile[LONG_TILE];
  const int w = blockIdx.y;
  if (rezero) {
    int* __restrict__ hc = t.hist[w];                      // cursor = hist + S (checked by the host)
    for (int q = blockIdx.x * 256 + threadIdx.x; q < 2 * t.S[w]; q += gridDim.x * 256) hc[q] = 0;
  }
  if ((int)blockIdx.x < nshort)
    seg_rank_sort_body(t.key[w], t.kptr[w], t.tmp[w], t.M[w], t.perm[w], blockIdx.x * 256 + threadIdx.x);
  else if (t.M[w] > RANK_MAX)
    seg_rank_sort_long_body(t.kptr[w], t.S[w], t.tmp[w], t.perm[w], tile, blockIdx.x - nshort, gridDim.x - nshort);
}

__global__ void k_i32_to_i64(const int* __restrict__ in, int64_t* __restrict__ out, int64_t n) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) out[q] = in[q];
}
__global__ void k_i64_to_i32(const int64_t* __restrict__ in, int* __restrict__ out, int64_t n) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) out[q] = (int)in[q];
}

// ------------------------------------------------------------------------------------------------
// refill of the static-shape buffers of a HIP-graph batch (dig_amd/graphed.py): up to 16 arrays of 4-byte words,
// dst[a][0..live) = src[a][..], dst[a][live..cap) = fill[a]; plus the live counts.  One launch instead of ~35
// small copies and fills per step.
#define PACK_MAX 16
struct PackTable {
  const uint32_t* src[PACK_MAX];
  uint32_t* dst[PACK_MAX];
  int live[PACK_MAX];
  int cap[PACK_MAX];
  uint32_t fill[PACK_MAX];
  int cnt[4];
  int* cnt_out;
};
__global__ void k_pack_static(PackTable t) {
  const int a = blockIdx.y;
  const uint32_t* __restrict__ src = t.src[a];
  uint32_t* __restrict__ dst = t.dst[a];
  const int live = t.live[a], cap = t.cap[a];
  const uint32_t fill = t.fill[a];
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < cap; q += gridDim.x * blockDim.x)
    dst[q] = q < live ? src[q] : fill;
  if (a == 0 && blockIdx.x == 0 && threadIdx.x < 4 && t.cnt_out) t.cnt_out[threadIdx.x] = t.cnt[threadIdx.x];
}

// ------------------------------------------------------------------------------------------------
// Gradient pieces -> the flat gradient buffer of a replayed step (dig_amd/graphed.py): entry e copies n[e] floats from
// src[e] (NULL: zeros) to flat[off[e] ...] and zero-fills up to span[e] (the 16-byte padding of dig_amd.optim.flat_layout).
// One launch per 192 pieces instead of the framework's concatenation (two copy kernels + a fill).
// ------------------------------------------------------------------------------------------------
#define PF_MAX 192
struct FlatPack {
  const float* src[PF_MAX];
  int n[PF_MAX], span[PF_MAX], off[PF_MAX];
};
__global__ void __launch_bounds__(256) k_pack_flat(FlatPack t, float* __restrict__ flat) {
  const int e = blockIdx.y;
  const float* __restrict__ src = t.src[e];
  float* __restrict__ dst = flat + t.off[e];
  const int n = src ? t.n[e] : 0, span = t.span[e];
  for (int q = blockIdx.x * 256 + threadIdx.x; q < span; q += gridDim.x * 256) dst[q] = q < n ? src[q] : 0.f;
}

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

// np pieces (host arrays: device pointers or NULL, lengths, padded lengths, offsets in floats) -> flat
int dig3d_pack_flat(int np, const void* const* src, const int* n, const int* span, const int* off, float* flat,
                    void* stream) {
  DIG3D_ENTER();
  if (np < 0 || (np > 0 && (!n || !span || !off || !src || !flat))) return DIG3D_ERR_ARG;
  for (int p0 = 0; p0 < np; p0 += PF_MAX) {
    FlatPack t;
    const int m = np - p0 < PF_MAX ? np - p0 : PF_MAX;
    int big = 1;
    for (int e = 0; e < m; ++e) {
      if (n[p0 + e] < 0 || span[p0 + e] < n[p0 + e] || off[p0 + e] < 0) return DIG3D_ERR_ARG;
      t.src[e] = (const float*)src[p0 + e];
      t.n[e] = n[p0 + e];
      t.span[e] = span[p0 + e];
      t.off[e] = off[p0 + e];
      if (span[p0 + e] > big) big = span[p0 + e];
    }
    int bx = (big + 1023) / 1024;             // up to 4 elements per thread for the largest piece
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(k_pack_flat, dim3(bx, m), dim3(256), 0, (hipStream_t)stream, t, flat);
    DIG3D_CHECK_LAUNCH();
  }
  return DIG3D_OK;
}

// host arrays of n <= 16 descriptors (device pointers, word counts); cnt_out (device int[4]) receives cnt[0..3].
int dig3d_pack_static(const void* const* src, void* const* dst, const int* live_words, const int* cap_words,
                      const uint32_t* fill, int n, const int* cnt, int* cnt_out, void* stream) {
  DIG3D_ENTER();
  if (n < 1 || n > PACK_MAX || !src || !dst || !live_words || !cap_words || !fill) return DIG3D_ERR_ARG;
  PackTable t;
  int maxcap = 1;
  for (int a = 0; a < n; ++a) {
    if (live_words[a] < 0 || cap_words[a] < live_words[a] || !dst[a] || (live_words[a] > 0 && !src[a]))
      return DIG3D_ERR_ARG;
    t.src[a] = (const uint32_t*)src[a];
    t.dst[a] = (uint32_t*)dst[a];
    t.live[a] = live_words[a];
    t.cap[a] = cap_words[a];
    t.fill[a] = fill[a];
    if (cap_words[a] > maxcap) maxcap = cap_words[a];
  }
  for (int q = 0; q < 4; ++q) t.cnt[q] = cnt ? cnt[q] : 0;
  t.cnt_out = cnt ? cnt_out : nullptr;
  int bx = (maxcap + 1023) / 1024;
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(k_pack_static, dim3(bx, n), dim3(256), 0, (hipStream_t)stream, t);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// Stage 1 of the per-batch graph build (no host sync inside):
//   ptr[N+2], nbr[N*width], deg[N], rowptr[N+1], src/dst[N*width] (worst case), cnt[N*width],
//   tptr[N*width+1], meta[8] (int64: [0]=B, [1]=E, [2]=T, [7]=error bits), ws[>= N*width/4096+2].
// meta_host (optional): PINNED host memory, int64[8]; the last kernel of the build writes meta there itself — the caller
// waits for an event recorded behind this call and reads the sizes, no copy.  NULL: the caller copies meta to the host.
// Then dig3d_graph_triplets_fill.
int dig3d_graph_build(const float* pos, const int64_t* batch, int N, float r, int max_num_neighbors,
                      int loop, int* ptr, int* nbr, int* deg, int* rowptr, int* src, int* dst, int* cnt,
                      int* tptr, int64_t* meta, int* ws, int want_triplets, int* batch32, int64_t* meta_host,
                      void* stream) {
  DIG3D_ENTER();
  if (N < 0 || !pos || !batch || !meta) return DIG3D_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (N == 0 || N > 16384) {
    if (dig3d_zero_async(meta, 8 * sizeof(int64_t), st) != hipSuccess) return DIG3D_ERR_LAUNCH;
  }
  if (N == 0) return DIG3D_OK;
  int width = max_num_neighbors + (loop ? 0 : 1);
  int cap = width;
  if (N <= 16384)      // one workgroup walks the batch vector and zeroes meta itself
    hipLaunchKernelGGL(k_graph_ptr<true>, dim3(1), dim3(1024), 0, st, batch, N, ptr, meta, batch32);
  else
    hipLaunchKernelGGL(k_graph_ptr<false>, dim3(dig3d_blocks(N, 256)), dim3(256), 0, st, batch, N, ptr, meta, batch32);
  hipLaunchKernelGGL(k_radius, dim3(dig3d_blocks((int64_t)N * 64, 256)), dim3(256), 0, st, pos, batch, ptr, N,
                     r, cap, loop, width, nbr, deg);
  DIG3D_CHECK_LAUNCH();
  int rc = scan_i32(deg, rowptr, N, nullptr, &meta[1], ws, st, meta, want_triplets ? nullptr : meta_host);
  if (rc) return rc;
  int64_t slots = (int64_t)N * width;
  hipLaunchKernelGGL(k_edges_fill, dim3(dig3d_blocks(slots, 256)), dim3(256), 0, st, nbr, deg, rowptr, N, width,
                     src, dst);
  DIG3D_CHECK_LAUNCH();
  if (want_triplets) {
    if (slots > 2147483647LL) return DIG3D_ERR_ARG;
    hipLaunchKernelGGL(k_trip_count, dim3(dig3d_blocks(slots, 256)), dim3(256), 0, st, rowptr, src, src, dst,
                       (int)slots, &meta[1], cnt);
    DIG3D_CHECK_LAUNCH();
    rc = scan_i32(cnt, tptr, (int)slots, &meta[1], &meta[2], ws, st, meta, meta_host);
    if (rc) return rc;
  }
  return DIG3D_OK;
}

// Stage 2: fill idx_kj / idx_ji (int32) once T is known on the host.
int dig3d_graph_triplets_fill(const int* rowptr, const int* col, const int* val, const int* esrc,
                              const int* edst, const int* tptr, int E, int* kj, int* ji, void* stream) {
  DIG3D_ENTER();
  if (E <= 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_trip_fill, dim3(dig3d_blocks((int64_t)E * TF_LPE, 256)), dim3(256), 0, (hipStream_t)stream, rowptr, col, val,
                     esrc, edst, tptr, E, kj, ji);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// Triplet count + scan for a caller-supplied CSR (generic xyz_to_dat path).  tptr[E+1]; *total (int64).
int dig3d_graph_triplets_count(const int* rowptr, const int* col, const int* esrc, const int* edst, int E,
                               int* cnt, int* tptr, int64_t* total, int* ws, void* stream) {
  DIG3D_ENTER();
  hipStream_t st = (hipStream_t)stream;
  if (E <= 0) {
    if (dig3d_zero_async(total, sizeof(int64_t), st) != hipSuccess) return DIG3D_ERR_LAUNCH;
    if (dig3d_zero_async(tptr, sizeof(int), st) != hipSuccess) return DIG3D_ERR_LAUNCH;
    return DIG3D_OK;
  }
  hipLaunchKernelGGL(k_trip_count, dim3(dig3d_blocks(E, 256)), dim3(256), 0, st, rowptr, col, esrc, edst, E,
                     (const int64_t*)nullptr, cnt);
  DIG3D_CHECK_LAUNCH();
  return scan_i32(cnt, tptr, E, nullptr, total, ws, st);
}

// Transposed CSR: key[M] in [0,S) -> kptr[S+1], perm[M] (positions grouped by key, ascending inside).
// hist/cursor: int[S] scratch each; tmp: int[M] scratch; ws: int[S/4096+3].
// n <= 4 transposed CSRs in one set of launches (host arrays of device pointers / sizes; hc[i]: 2 S[i] ints = histogram +
// cursors of key i; adjacent hc buffers are zeroed by one memset).  S[i] <= 32768 (single-block scans); larger: one
// dig3d_csr_by_key per key.
int dig3d_csr_by_keys_ws(int n, const void* const* key, const int* M, const int* S, void* const* kptr, void* const* perm,
                         void* const* hc, void* const* tmp, int hc_clean, void* stream);
int dig3d_csr_by_keys(int n, const void* const* key, const int* M, const int* S, void* const* kptr, void* const* perm,
                      void* const* hc, void* const* tmp, void* stream) {
  return dig3d_csr_by_keys_ws(n, key, M, S, kptr, perm, hc, tmp, -1, stream);
}

// hc_clean: 1 = the caller's workspace hc is all zero on entry (left so by the previous call of this function on it): no
// zero-fill launch; 0 = zero it first.  Either way the launch set leaves hc zero again.  (-1: dig3d_csr_by_keys — zero
// first, leave as is.)
int dig3d_csr_by_keys_ws(int n, const void* const* key, const int* M, const int* S, void* const* kptr, void* const* perm,
                         void* const* hc, void* const* tmp, int hc_clean, void* stream) {
  DIG3D_ENTER();
  hipStream_t st = (hipStream_t)stream;
  if (n < 1 || n > CSRS_MAX || !key || !M || !S || !kptr || !perm || !hc || !tmp) return DIG3D_ERR_ARG;
  CsrSet t;
  int maxM = 0, lb = 0;
  for (int i = 0; i < n; ++i) {
    if (M[i] < 0 || S[i] < 1 || S[i] > 32768 || !kptr[i] || !hc[i] || (M[i] > 0 && (!key[i] || !perm[i] || !tmp[i]))) return DIG3D_ERR_ARG;
    t.key[i] = (const int*)key[i];
    t.kptr[i] = (int*)kptr[i];
    t.perm[i] = (int*)perm[i];
    t.hist[i] = (int*)hc[i];
    t.cursor[i] = (int*)hc[i] + S[i];
    t.tmp[i] = (int*)tmp[i];
    t.M[i] = M[i];
    t.S[i] = S[i];
    if (M[i] > maxM) maxM = M[i];
    if (M[i] > RANK_MAX) {
      const int l = S[i] < 256 ? S[i] : 256;
      if (l > lb) lb = l;
    }
  }
  for (int i = 0; i < n && hc_clean != 1;) {             // one memset per run of adjacent histogram / cursor buffers
    int j = i;
    size_t words = 2 * (size_t)S[i];
    while (j + 1 < n && (int*)hc[j + 1] == (int*)hc[j] + 2 * S[j]) words += 2 * (size_t)S[++j];
    if (dig3d_zero_async(hc[i], sizeof(int) * words, st) != hipSuccess) return DIG3D_ERR_LAUNCH;
    i = j + 1;
  }
  if (maxM > 0) hipLaunchKernelGGL(k_keys_hist, dim3(dig3d_blocks(maxM, 256), n), dim3(256), 0, st, t);
  hipLaunchKernelGGL(k_keys_scan, dim3(n), dim3(SCAN_T), 0, st, t);
  if (maxM > 0) {
    const int nshort = dig3d_blocks(maxM, 256);
    hipLaunchKernelGGL(k_keys_fill, dim3(nshort, n), dim3(256), 0, st, t);
    hipLaunchKernelGGL(k_keys_rank_sort, dim3(nshort + lb, n), dim3(256), 0, st, t, nshort, hc_clean >= 0 ? 1 : 0);
  }
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

int dig3d_csr_by_key(const int* key, int M, int S, int* kptr, int* perm, int* hist, int* cursor, int* tmp, int* ws,
                     void* stream) {
  DIG3D_ENTER();
  hipStream_t st = (hipStream_t)stream;
  if (S < 0 || M < 0) return DIG3D_ERR_ARG;
  if (S == 0) return DIG3D_OK;
  if (cursor == hist + S) {             // adjacent workspaces (dig_amd/graph.py allocates them as one): one memset
    if (dig3d_zero_async(hist, sizeof(int) * 2 * (size_t)S, st) != hipSuccess) return DIG3D_ERR_LAUNCH;
  } else {
    if (dig3d_zero_async(hist, sizeof(int) * (size_t)S, st) != hipSuccess) return DIG3D_ERR_LAUNCH;
    if (dig3d_zero_async(cursor, sizeof(int) * (size_t)S, st) != hipSuccess) return DIG3D_ERR_LAUNCH;
  }
  if (M > 0) hipLaunchKernelGGL(k_key_hist, dim3(dig3d_blocks(M, 256)), dim3(256), 0, st, key, M, hist);
  int rc = scan_i32(hist, kptr, S, nullptr, nullptr, ws + 1, st);
  if (rc) return rc;
  if (M > 0) {
    // slots inside a segment come from atomics (arbitrary order) -> tmp; the rank sort makes perm deterministic
    hipLaunchKernelGGL(k_key_fill, dim3(dig3d_blocks(M, 256)), dim3(256), 0, st, key, M, kptr, cursor, tmp);
    hipLaunchKernelGGL(k_seg_rank_sort, dim3(dig3d_blocks(M, 256)), dim3(256), 0, st, key, kptr, tmp, M, perm);
    if (M > RANK_MAX) {                 // a longer segment can only exist then; blocks skip the short segments
      int lb = S < 1024 ? S : 1024;
      hipLaunchKernelGGL(k_seg_rank_sort_long, dim3(lb), dim3(256), 0, st, kptr, S, tmp, perm);
    }
  }
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

// Exclusive scan exposed for the host (rowptr from degree counts).
int dig3d_scan_i32(const int* in, int* out, int n, int64_t* total, int* ws, void* stream) {
  DIG3D_ENTER();
  if (n < 0) return DIG3D_ERR_ARG;
  return scan_i32(in, out, n, nullptr, total, ws, (hipStream_t)stream);
}

int dig3d_cast_i32_i64(const int* in, int64_t* out, int64_t n, void* stream) {
  DIG3D_ENTER();
  if (n <= 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_i32_to_i64, dim3(dig3d_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, n);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}
int dig3d_cast_i64_i32(const int64_t* in, int* out, int64_t n, void* stream) {
  DIG3D_ENTER();
  if (n <= 0) return DIG3D_OK;
  hipLaunchKernelGGL(k_i64_to_i32, dim3(dig3d_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, n);
  DIG3D_CHECK_LAUNCH();
  return DIG3D_OK;
}

}  // extern "C"
