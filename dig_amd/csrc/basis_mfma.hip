// dig3d — the first basis Linears of the triplet interaction on the matrix cores (r04).
//
// Reference (method/spherenet/spherenet.py:163,166 with features.py:213-222,256-263): per interaction layer
//     sbf1 = lin_sbf1(sbf)   sbf [T, ns*nr]      -> [T, basis_emb <= 8]
//     t1   = lin_t1(t)       t   [T, ns*ns*nr]   -> [T, basis_emb <= 8]
// — two GEMMs per layer with K = 42 / 294 at the defaults, M = T (1e5 triplets per 32 QM9 molecules, 1e6+ on OC20-like
// systems).  triplet.hip:k_basis_project evaluates the basis in registers and runs the contraction as scalar-weight
// FMAs: 10.7k FMAs per triplet, 119 us per step at VALU busy 16 % (r03 counters) — the weight fetches (s_load) and the
// one-triplet-per-lane dependency chains, not the arithmetic, set its time.  Here the same contraction is
// v_mfma_f32_16x16x4_f32 (IEEE float32 multiply-add, no reduced precision):
//   * a wave generates the harmonics Y_h(t) and the gathered radial row bes[kj[t]] of 64 triplets into its PRIVATE LDS
//     slab (one lane per triplet, no block barrier), then forms the A operand basis(t, k) = Y_h * bes[h mod ns, n] on the
//     fly — two LDS reads and one multiply per MFMA pair — so the [T, 294] table still never exists;
//   * the reduction index is laid out as k' = n * H2P + h (H2P = harmonics padded to a multiple of 4) so that the four
//     k of one MFMA step share n and differ in h = 4 s + kq: no per-lane division in the loop;
//   * the stacked first Linears of up to four layers are the B operand, 32 output columns (two accumulator tiles), staged
//     once per workgroup in LDS with the column-tile bit XOR-swizzled by (k' >> 1) & 1: the 4 x 16 lanes of an operand
//     read hit 64 distinct banks.
// The weight gradient is the transposed product gW[k'][o] = sum_t basis(t, k') gP[t][o] on the same operands (its row
// tiles dealt to the four waves of a workgroup, one partial per workgroup in the layout of k_basis_wgrad).  Shapes outside the instantiated set keep the VALU kernels (basis_project_mfma returns 1).
#include "basis_mfma.h"
#include "sph.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define BM_PO 32            // stacked outputs (4 layers x 8)
#define BM_PB 8
#define BM_NRMAX 6          // radial functions covered by the weight-gradient accumulator set

__device__ __forceinline__ f32x4 bm_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ void bm_wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int NS, bool TOR>
struct BasisDims {
  static constexpr int H2 = TOR ? NS * NS : NS;     // harmonics per triplet
  static constexpr int H2P = (H2 + 3) & ~3;         // padded to whole MFMA steps
  static constexpr int YS = H2P + 1;                // odd LDS pitch; column H2P is a zero slot
  static constexpr int QH = H2P / 4;                // MFMA steps per radial index
};

// one thread's share of a gathered radial row -> LDS: elements [part * ch, part * ch + ch) of the KB, ch = ceil(KB / nparts),
// BATCH independent loads in flight (unconditional, clamped; masked on the store; the accumulators are not live here, the
// registers are free: 24 per trip for a whole row per lane, 16 for a third of a row).  One element per loop trip behind its
// own index load was the time of the first r04 kernels: 14 serial round trips of ~1.5 us per 64-triplet tile.
template <int BATCH>
__device__ __forceinline__ void bm_stage_row(const float* __restrict__ g, float* __restrict__ dst, int KB, int part, int nparts,
                                             bool live) {
  const int ch = (KB + nparts - 1) / nparts;
  const int k0 = part * ch, k1 = k0 + ch < KB ? k0 + ch : KB;
  for (int k = k0; k < k1; k += BATCH) {
    float v[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) v[u] = g[k + u < k1 ? k + u : k1 - 1];
#pragma unroll
    for (int u = 0; u < BATCH; ++u)
      if (k + u < k1) dst[k + u] = live ? v[u] : 0.f;
  }
}

// harmonics + gathered radial rows of `ntr` triplets starting at t0 into the wave's slab (rows beyond Tl: zeros)
template <int NS, bool TOR, int NTR>
__device__ __forceinline__ void bm_generate(const float* __restrict__ bes, const int* __restrict__ kj,
                                            const float* __restrict__ angle, const float* __restrict__ torsion, int t0,
                                            int Tl, int nr, const float* sPref, float* sY, float* sB, int lane) {
  using D = BasisDims<NS, TOR>;
  const int KB = NS * nr, BS = KB | 1;
  if (lane < NTR) {
    const int t = t0 + lane;
    float Y[D::H2];
    if (t < Tl) {
      real_sph_harm<NS>(angle[t], TOR ? torsion[t] : 0.f, sPref, !TOR, Y);
    } else {
#pragma unroll
      for (int h = 0; h < D::H2; ++h) Y[h] = 0.f;
    }
#pragma unroll
    for (int h = 0; h < D::H2; ++h) sY[lane * D::YS + h] = Y[h];
#pragma unroll
    for (int h = D::H2; h <= D::H2P; ++h) sY[lane * D::YS + h] = 0.f;
  }
  // radial rows: 64 / NTR lanes share a triplet's row
  constexpr int LPT = 64 / NTR;
  const int tl = lane % NTR, part = lane / NTR;
  const int t = t0 + tl;
  const bool live = t < Tl;
  bm_stage_row<24>(bes + (int64_t)(live ? kj[t] : 0) * KB, sB + tl * BS, KB, part, LPT, live);
}

template <int NS, bool TOR, int NTR>
__host__ __device__ inline int bm_fwd_slab(int nr) {
  using D = BasisDims<NS, TOR>;
  const int ops = NTR * D::YS + NTR * ((NS * nr) | 1), outs = NTR * 36 * (TOR ? 2 : 1);
  return ops > outs ? ops : outs;
}

// ------------------------------------------------------------------------------------------------------------------
// forward: Ps[l][t][8] (and Pt) for l < L from the stacked, transposed, zero-padded weights Ws[ns*nr][32], Wt[ns*ns*nr][32]
// ------------------------------------------------------------------------------------------------------------------
// NTR triplets per wave tile, NW = 256 / NTR waves per workgroup.  NTR = 64 / four waves was the r04 form: its slabs and the
// weight tables fill the LDS of a CU with ONE workgroup and the kernel keeps 282 registers — one wave per SIMD, so a wave's
// operand generation (harmonics, gathered radial rows: global loads) and its MFMA loop never overlap with anything
// (57 us at T = 1.1e5 for 19 us of MFMA issue).  NTR = 32 / eight waves shares the same weight tables between twice the
// waves (32 accumulator registers instead of 64, <= 256 registers by the launch bound): two waves per SIMD, one generates
// while the other multiplies, and the tail of the tile deal is half as long.  Per-row arithmetic (the k order of every
// accumulator) is unchanged: results are bit-identical between the two forms.
template <int NS, bool TOR, int NTR>
__global__ void __launch_bounds__(16384 / NTR) k_basis_project_mfma(const float* __restrict__ bes, const int* __restrict__ kj,
                                                             const float* __restrict__ angle,
                                                             const float* __restrict__ torsion, int T, int nr,
                                                             const float* __restrict__ pref, const float* __restrict__ Ws,
                                                             const float* __restrict__ Wt, int L, float* __restrict__ Ps,
                                                             float* __restrict__ Pt, const int* __restrict__ cnt) {
  using D = BasisDims<NS, TOR>;
  constexpr int NW = 256 / NTR, NT = 64 * NW, MB = NTR / 16;     // waves, threads, 16-row blocks per wave tile
  extern __shared__ float bsm[];
  __shared__ float sPref[NS_MAX * NS_MAX];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
  const int KB = NS * nr, BS = KB | 1;
  const int KTP = TOR ? nr * D::H2P : 0, KSP = nr * 8;
  float* sWt = bsm;                                       // [KTP][32], column tile swizzled
  float* sWs = sWt + KTP * BM_PO;                         // [KSP][32]
  // per-wave slab: operands [NTR][YS] + [NTR][BS] while multiplying, then the transposed results [NTR][36] (x 2 with torsion)
  const int slabf = bm_fwd_slab<NS, TOR, NTR>(nr);
  float* sY = sWs + KSP * BM_PO + wave * slabf;
  float* sB = sY + NTR * D::YS;
  const int Tl = (cnt && *cnt < T) ? *cnt : T;            // static-shape batch: rows in [Tl, T) are padding, never written
  for (int q = threadIdx.x; q < NS_MAX * NS_MAX; q += NT) sPref[q] = pref[q];
  // weight tables -> LDS, eight UNCONDITIONAL loads in flight per thread (padded slots read a clamped address and are
  // zeroed on the way in): one load per loop trip under a predicate was 45 serial L2 round trips per workgroup — ~27 us
  // of the 68 us this kernel took at T = 1e5 (r04 first version)
  if (TOR) {
    for (int q0 = threadIdx.x; q0 < KTP * BM_PO; q0 += NT * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = q0 + NT * u, qc = q < KTP * BM_PO ? q : 0;
        const int kp = qc >> 5, o = qc & 31, n = kp / D::H2P, h = kp - n * D::H2P;
        const float w = Wt[(int64_t)((h < D::H2 ? h : 0) * nr + n) * BM_PO + o];
        v[u] = h < D::H2 ? w : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = q0 + NT * u;
        if (q < KTP * BM_PO) {
          const int kp = q >> 5, o = q & 31;
          sWt[kp * BM_PO + (o ^ (((kp >> 1) & 1) << 4))] = v[u];
        }
      }
    }
  }
  for (int q0 = threadIdx.x; q0 < KSP * BM_PO; q0 += NT * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = q0 + NT * u, qc = q < KSP * BM_PO ? q : 0;
      const int kp = qc >> 5, o = qc & 31, n = kp >> 3, l = kp & 7;
      const float w = Ws[(int64_t)((l < NS ? l : 0) * nr + n) * BM_PO + o];
      v[u] = l < NS ? w : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = q0 + NT * u;
      if (q < KSP * BM_PO) {
        const int kp = q >> 5, o = q & 31;
        sWs[kp * BM_PO + (o ^ (((kp >> 1) & 1) << 4))] = v[u];
      }
    }
  }
  __syncthreads();
  const int ntiles = (Tl + NTR - 1) / NTR;
  const int swz = ((kq >> 1) & 1) << 4;
  for (int tile = blockIdx.x * NW + wave; tile < ntiles; tile += gridDim.x * NW) {
    const int t0 = tile * NTR;
    bm_wave_fence();                                      // the previous tile's operand reads are done
    bm_generate<NS, TOR, NTR>(bes, kj, angle, torsion, t0, Tl, nr, sPref, sY, sB, lane);
    bm_wave_fence();
    f32x4 accS[MB][2], accT[MB][2];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        accS[m][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        accT[m][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    const float* yr = sY + i * D::YS;
    const float* br = sB + i * BS;
    if (TOR) {
      for (int n = 0; n < nr; ++n) {
#pragma unroll
        for (int sq = 0; sq < D::QH; ++sq) {
          const int h = 4 * sq + kq;
          const int bo = (h % NS) * nr + n;               // padded h >= H2: Y is 0 there, any valid radial slot will do
          const int kp = n * D::H2P + h;
          const float b0 = sWt[kp * BM_PO + (i ^ swz)], b1 = sWt[kp * BM_PO + ((16 + i) ^ swz)];
#pragma unroll
          for (int m = 0; m < MB; ++m) {
            const float a = yr[(16 * m) * D::YS + h] * br[(16 * m) * BS + bo];
            accT[m][0] = bm_mfma(a, b0, accT[m][0]);
            accT[m][1] = bm_mfma(a, b1, accT[m][1]);
          }
        }
      }
    }
    for (int n = 0; n < nr; ++n) {
#pragma unroll
      for (int sq = 0; sq < 2; ++sq) {
        const int l = 4 * sq + kq;
        const bool ok = l < NS;
        const int yo = ok ? (TOR ? l * l : l) : D::H2P;   // zero slot for the padded degrees
        const int bo = ok ? l * nr + n : 0;
        const int kp = n * 8 + l;
        const float b0 = sWs[kp * BM_PO + (i ^ swz)], b1 = sWs[kp * BM_PO + ((16 + i) ^ swz)];
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          const float a = yr[(16 * m) * D::YS + yo] * br[(16 * m) * BS + bo];
          accS[m][0] = bm_mfma(a, b0, accS[m][0]);
          accS[m][1] = bm_mfma(a, b1, accS[m][1]);
        }
      }
    }
    // D layout: lane (i, kq) holds rows 4 kq + r (triplets), column i of the tile: output o = 16 ct + i = layer * 8 + b.
    // Stored straight from there a lane writes 64 single floats per tile (8 segments of 32 bytes per store instruction);
    // the tile is transposed through the wave's slab (free now: pitch 36 floats) instead, and every lane writes ITS
    // triplet's eight floats per layer and table as two 16-byte stores — consecutive lanes, consecutive 32-byte rows.
    bm_wave_fence();                                      // every operand read of this tile is done
    float* sOs = sY;                                      // [NTR][36]
    float* sOt = sY + NTR * 36;                           // [NTR][36]  (bm_fwd_slab sizes the slab for both uses)
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = (16 * m + 4 * kq + r) * 36 + 16 * ct + i;
          sOs[o] = accS[m][ct][r];
          if (TOR) sOt[o] = accT[m][ct][r];
        }
    bm_wave_fence();
    if (NTR == 64) {
      const int t = t0 + lane;
      if (t < Tl) {
        const float* rs = sOs + lane * 36;
        const float* rt = sOt + lane * 36;
        for (int l = 0; l < L; ++l) {
          float4* ps = (float4*)(Ps + ((int64_t)l * T + t) * BM_PB);
          ps[0] = *(const float4*)(rs + 8 * l);
          ps[1] = *(const float4*)(rs + 8 * l + 4);
          if (TOR) {
            float4* pt = (float4*)(Pt + ((int64_t)l * T + t) * BM_PB);
            pt[0] = *(const float4*)(rt + 8 * l);
            pt[1] = *(const float4*)(rt + 8 * l + 4);
          }
        }
      }
    } else {
      // 32 triplets, 64 lanes: the upper half-wave stores the torsion table (or, without one, the odd layers)
      const int tl = lane & (NTR - 1), hf = lane / NTR, t = t0 + tl;
      if (t < Tl) {
        const float* rr = (TOR && hf ? sOt : sOs) + tl * 36;
        float* base = TOR && hf ? Pt : Ps;
        for (int l = TOR ? 0 : hf; l < L; l += TOR ? 1 : 2) {
          float4* pp = (float4*)(base + ((int64_t)l * T + t) * BM_PB);
          pp[0] = *(const float4*)(rr + 8 * l);
          pp[1] = *(const float4*)(rr + 8 * l + 4);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// weight gradient: part[blockIdx][ [32][KS] then [32][KT] ] = sum over this workgroup's triplets of gP (x) basis
//   gW'[k'][o] = sum_t basis(t, k') gP[t][o]:  A = basis^T (rows = k', 16 per accumulator tile), B = gP (32 columns).
// The 16-row tiles of k' (20 for the torsion table + 3 for the sbf table at ns = 7, nr = 6) are DEALT TO THE FOUR WAVES of
// the workgroup (tile m -> wave m mod 4): a wave keeps <= 6 x 2 accumulator tiles (48 registers) for the whole kernel and
// owns its output rows outright — no cross-wave reduction, no accumulator file of 184 registers (the first r04 version:
// every wave all 23 tiles of its own 32 triplets, 255 VGPRs + 256 AGPRs, 119.7 us at T = 1.0e5 against 99.7 for the
// VALU kernel).  The 64 triplets of a block tile are generated cooperatively into ONE shared slab (wave 0: harmonics;
// waves 1-3: radial rows and incoming gradients), 40 KB per workgroup: three workgroups per CU, so one generates while
// the others multiply (amdgpu_waves_per_eu(3, 3) holds the kernel to 168 registers for that, no scratch — r06 same box:
// 253 -> 217 us at T = 5.9e5 with three blocks per CU).
// ------------------------------------------------------------------------------------------------------------------
#define BM_WTB 64           // triplets per block tile of the weight-gradient kernel
template <int NS, bool TOR>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) k_basis_wgrad_mfma(const float* __restrict__ bes, const int* __restrict__ kj,
                                                           const float* __restrict__ angle,
                                                           const float* __restrict__ torsion, int T, int nr,
                                                           const float* __restrict__ pref, const float* __restrict__ gPs,
                                                           const float* __restrict__ gPt, int L, float* __restrict__ part,
                                                           const int* __restrict__ cnt) {
  using D = BasisDims<NS, TOR>;
  constexpr int MTT = TOR ? (BM_NRMAX * D::H2P + 15) / 16 : 0;   // row tiles of the torsion table at the largest nr
  constexpr int MTS = (BM_NRMAX * 8 + 15) / 16;                  // sbf table
  constexpr int MPW = (MTT + MTS + 3) / 4;                       // tiles per wave
  extern __shared__ float bsm[];
  __shared__ float sPref[NS_MAX * NS_MAX];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
  const int KB = NS * nr, BS = KB | 1;
  const int KTP = TOR ? nr * D::H2P : 0, KSP = nr * 8;
  const int mtt = (KTP + 15) >> 4, mts = (KSP + 15) >> 4, MT = mtt + mts;
  const int KS = NS * nr, KT = TOR ? NS * NS * nr : 0;
  float* sY = bsm;                                         // [64][YS]
  float* sB = sY + BM_WTB * D::YS;                         // [64][BS]
  float* sGs = sB + BM_WTB * BS;                           // [64][32], column tile swizzled by (t >> 1) & 1
  float* sGt = sGs + BM_WTB * BM_PO;
  const int Tl = (cnt && *cnt < T) ? *cnt : T;
  for (int q = threadIdx.x; q < NS_MAX * NS_MAX; q += 256) sPref[q] = pref[q];
  // this lane's operand rows of the wave's tiles: k' = 16 m + i -> (harmonic slot | radial slot << 8)
  int ix[MPW];
#pragma unroll
  for (int j = 0; j < MPW; ++j) {
    const int m = wave + 4 * j;
    int v = D::H2P;                                        // zero slot, radial slot 0
    if (m < mtt) {
      const int kp = 16 * m + i;
      const int n = kp / D::H2P, h = kp - n * D::H2P;
      if (kp < KTP) v = h | (((h % NS) * nr + n) << 8);
    } else if (m < MT) {
      const int kp = 16 * (m - mtt) + i;
      const int n = kp >> 3, l = kp & 7;
      if (n < nr && l < NS) v = (TOR ? l * l : l) | ((l * nr + n) << 8);
    }
    ix[j] = v;
  }
  f32x4 acc[MPW][2];
#pragma unroll
  for (int j = 0; j < MPW; ++j) acc[j][0] = acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ntiles = (Tl + BM_WTB - 1) / BM_WTB;
  const int swz = ((kq >> 1) & 1) << 4;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int t0 = tile * BM_WTB;
    __syncthreads();                                       // sPref staged / the previous tile's operand reads are done
    if (wave == 0) {                                       // harmonics of the 64 triplets, one per lane
      const int t = t0 + lane;
      float Y[D::H2];
      if (t < Tl) {
        real_sph_harm<NS>(angle[t], TOR ? torsion[t] : 0.f, sPref, !TOR, Y);
      } else {
#pragma unroll
        for (int h = 0; h < D::H2; ++h) Y[h] = 0.f;
      }
#pragma unroll
      for (int h = 0; h < D::H2; ++h) sY[lane * D::YS + h] = Y[h];
#pragma unroll
      for (int h = D::H2; h <= D::H2P; ++h) sY[lane * D::YS + h] = 0.f;
    } else {                                               // 192 threads: gathered radial rows, then the incoming gradients
      const int w3 = threadIdx.x - 64;
      {                                                    // three threads per triplet row, eight loads in flight each
        const int tl = w3 & (BM_WTB - 1), prt = w3 >> 6, t = t0 + tl;
        const bool live = t < Tl;
        bm_stage_row<16>(bes + (int64_t)(live ? kj[t] : 0) * KB, sB + tl * BS, KB, prt, 3, live);
      }
      // incoming gradients: item = (table, layer, triplet) = two float4; 512 items on 192 threads, all loads of a thread's
      // three items issued before the first LDS store (unconditional on a clamped row, zeroed on the way in)
      float4 v0[3], v1[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int q = w3 + 192 * r;
        const int tl = q & (BM_WTB - 1), l = (q >> 6) & 3, isT = (q >> 8) & 1, t = t0 + tl;
        const bool ok = q < BM_WTB * 8 && l < L && t < Tl && (TOR || !isT);
        const float4* p = (const float4*)(((isT && TOR) ? gPt : gPs) + ((int64_t)(ok ? l : 0) * T + (ok ? t : 0)) * BM_PB);
        v0[r] = p[0];
        v1[r] = p[1];
        if (!ok) v0[r] = v1[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int q = w3 + 192 * r;
        const int tl = q & (BM_WTB - 1), l = (q >> 6) & 3, isT = (q >> 8) & 1;
        if (q < BM_WTB * 8 && (TOR || !isT)) {
          float* dstp = (isT ? sGt : sGs) + tl * BM_PO;
          const int sw = ((tl >> 1) & 1) << 4;
          *(float4*)(dstp + ((l * 8) ^ sw)) = v0[r];
          *(float4*)(dstp + ((l * 8 + 4) ^ sw)) = v1[r];
        }
      }
    }
    __syncthreads();
#pragma unroll 2
    for (int st = 0; st < BM_WTB / 4; ++st) {
      const int tl = 4 * st + kq;
      const float* yr = sY + tl * D::YS;
      const float* br = sB + tl * BS;
      const float gs0 = sGs[tl * BM_PO + (i ^ swz)], gs1 = sGs[tl * BM_PO + ((16 + i) ^ swz)];
      float gt0 = 0.f, gt1 = 0.f;
      if (TOR) {
        gt0 = sGt[tl * BM_PO + (i ^ swz)];
        gt1 = sGt[tl * BM_PO + ((16 + i) ^ swz)];
      }
#pragma unroll
      for (int j = 0; j < MPW; ++j) {
        const int m = wave + 4 * j;                        // wave-uniform: which table this tile belongs to
        if (m < MT) {
          const float a = yr[ix[j] & 255] * br[ix[j] >> 8];
          const bool tt = m < mtt;
          acc[j][0] = bm_mfma(a, tt ? gt0 : gs0, acc[j][0]);
          acc[j][1] = bm_mfma(a, tt ? gt1 : gs1, acc[j][1]);
        }
      }
    }
  }
  // every wave owns its rows of the partial (k_basis_wgrad's layout: [32][KS] then [32][KT]).  Stored straight from the
  // accumulators a lane's 48 values go to 48 different cache lines, 4 bytes each (lane i = output column o, stride KT
  // floats): 5.5 M four-byte write requests per launch at 512 blocks — the fixed ~20 us of this kernel at T = 1.1e5.  The
  // block's partial is assembled in LDS (the operand slab is free now) and written out as whole rows, 16 bytes per lane.
  __syncthreads();                                         // the last tile's operand reads are done
  float* sOut = bsm;                                       // [(KS + KT) * 32], final layout
#pragma unroll
  for (int j = 0; j < MPW; ++j) {
    const int m = wave + 4 * j;
    if (m >= MT) continue;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int o = 16 * ct + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (m < mtt) {
          const int kp = 16 * m + 4 * kq + r;
          const int n = kp / D::H2P, h = kp - n * D::H2P;
          if (kp < KTP && h < D::H2) sOut[KS * BM_PO + o * KT + h * nr + n] = acc[j][ct][r];
        } else {
          const int kp = 16 * (m - mtt) + 4 * kq + r;
          const int n = kp >> 3, l = kp & 7;
          if (n < nr && l < NS) sOut[o * KS + l * nr + n] = acc[j][ct][r];
        }
      }
    }
  }
  __syncthreads();
  float4* outp = (float4*)(part + (int64_t)blockIdx.x * (KS + KT) * BM_PO);     // (KS + KT) * 32 floats: a multiple of 4
  const int n4 = (KS + KT) * BM_PO / 4;
  for (int q = threadIdx.x; q < n4; q += 256) outp[q] = ((const float4*)sOut)[q];
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
template <int NS, bool TOR, int NTR>
static size_t bm_fwd_smem(int nr) {
  using D = BasisDims<NS, TOR>;
  return sizeof(float) * ((size_t)(TOR ? nr * D::H2P : 0) * BM_PO + (size_t)nr * 8 * BM_PO +
                          (256 / NTR) * (size_t)bm_fwd_slab<NS, TOR, NTR>(nr));
}
template <int NS, bool TOR>
static size_t bm_wg_smem(int nr) {
  using D = BasisDims<NS, TOR>;
  const int KB = NS * nr, BS = KB | 1;
  const size_t ops = (size_t)(BM_WTB * D::YS + BM_WTB * BS + 2 * BM_WTB * BM_PO);
  const size_t outs = (size_t)(KB + (TOR ? NS * KB : 0)) * BM_PO;      // the block's partial, assembled in LDS by the epilogue
  return sizeof(float) * (ops > outs ? ops : outs);
}

#define BM_LDS_LIMIT (160 * 1024 - 2048)

template <int NS, bool TOR, int NTR>
static int bm_launch_fwd_n(const float* bes, const int* kj, const float* angle, const float* torsion, int T, int nr,
                           const float* pref, const float* Ws, const float* Wt, int L, float* Ps, float* Pt, const int* cnt,
                           hipStream_t st) {
  const size_t shm = bm_fwd_smem<NS, TOR, NTR>(nr);
  if (shm > BM_LDS_LIMIT) return 1;
  static const bool attr_ok = hipFuncSetAttribute((const void*)k_basis_project_mfma<NS, TOR, NTR>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, BM_LDS_LIMIT) == hipSuccess;
  if (!attr_ok) return 1;
  constexpr int NW = 256 / NTR;
  const int ntiles = (T + NTR - 1) / NTR;
  int nb = (ntiles + NW - 1) / NW;
  if (nb > dig3d_num_cus()) nb = dig3d_num_cus();
  hipLaunchKernelGGL((k_basis_project_mfma<NS, TOR, NTR>), dim3(nb), dim3(64 * NW), shm, st, bes, kj, angle, torsion, T, nr,
                     pref, Ws, Wt, L, Ps, Pt, cnt);
  return 0;
}
// form 0: eight waves x 32 triplets (two waves per SIMD); form 1: the r04 kernel, four waves x 64 triplets
template <int NS, bool TOR>
static int bm_launch_fwd(const float* bes, const int* kj, const float* angle, const float* torsion, int T, int nr,
                         const float* pref, const float* Ws, const float* Wt, int L, float* Ps, float* Pt, const int* cnt,
                         int form, hipStream_t st) {
  if (form == 0 && bm_launch_fwd_n<NS, TOR, 32>(bes, kj, angle, torsion, T, nr, pref, Ws, Wt, L, Ps, Pt, cnt, st) == 0) return 0;
  return bm_launch_fwd_n<NS, TOR, 64>(bes, kj, angle, torsion, T, nr, pref, Ws, Wt, L, Ps, Pt, cnt, st);
}

template <int NS, bool TOR>
static int bm_launch_wg(const float* bes, const int* kj, const float* angle, const float* torsion, int T, int nr,
                        const float* pref, const float* gPs, const float* gPt, int L, float* part, const int* cnt, int nb,
                        hipStream_t st) {
  const size_t shm = bm_wg_smem<NS, TOR>(nr);
  if (shm > BM_LDS_LIMIT || nr > BM_NRMAX) return 1;
  static const bool attr_ok = hipFuncSetAttribute((const void*)k_basis_wgrad_mfma<NS, TOR>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, BM_LDS_LIMIT) == hipSuccess;
  if (!attr_ok) return 1;
  hipLaunchKernelGGL((k_basis_wgrad_mfma<NS, TOR>), dim3(nb), dim3(256), shm, st, bes, kj, angle, torsion, T, nr, pref, gPs,
                     gPt, L, part, cnt);
  return 0;
}

// 0: launched; 1: shape not covered (the caller runs the VALU kernel)
int basis_project_mfma(const float* bes, const int* kj, const float* angle, const float* torsion, int T, int ns, int nr,
                       const float* pref, const float* Ws, const float* Wt, int L, float* Ps, float* Pt, const int* cnt,
                       int form, hipStream_t st) {
  const bool tor = torsion != nullptr;
  if (T < 2048 || nr < 1 || nr > 8) return 1;            // small batches: the launch is latency, not arithmetic
  if ((((uintptr_t)Ps | (uintptr_t)Pt) & 15) != 0) return 1;   // float4 stores of the projected rows
  if (ns == 7) return tor ? bm_launch_fwd<7, true>(bes, kj, angle, torsion, T, nr, pref, Ws, Wt, L, Ps, Pt, cnt, form, st)
                          : bm_launch_fwd<7, false>(bes, kj, angle, torsion, T, nr, pref, Ws, Wt, L, Ps, Pt, cnt, form, st);
  if (ns == 3) return tor ? bm_launch_fwd<3, true>(bes, kj, angle, torsion, T, nr, pref, Ws, Wt, L, Ps, Pt, cnt, form, st)
                          : bm_launch_fwd<3, false>(bes, kj, angle, torsion, T, nr, pref, Ws, Wt, L, Ps, Pt, cnt, form, st);
  return 1;
}

int basis_wgrad_mfma(const float* bes, const int* kj, const float* angle, const float* torsion, int T, int ns, int nr,
                     const float* pref, const float* gPs, const float* gPt, int L, float* part, const int* cnt, int nb,
                     hipStream_t st) {
  const bool tor = torsion != nullptr;
  if (T < 2048 || nr < 1) return 1;
  if ((((uintptr_t)gPs | (uintptr_t)gPt | (uintptr_t)part) & 15) != 0) return 1;  // float4 loads of the gradient rows, float4 stores of the partials
  if (ns == 7) return tor ? bm_launch_wg<7, true>(bes, kj, angle, torsion, T, nr, pref, gPs, gPt, L, part, cnt, nb, st)
                          : bm_launch_wg<7, false>(bes, kj, angle, torsion, T, nr, pref, gPs, gPt, L, part, cnt, nb, st);
  if (ns == 3) return tor ? bm_launch_wg<3, true>(bes, kj, angle, torsion, T, nr, pref, gPs, gPt, L, part, cnt, nb, st)
                          : bm_launch_wg<3, false>(bes, kj, angle, torsion, T, nr, pref, gPs, gPt, L, part, cnt, nb, st);
  return 1;
}
