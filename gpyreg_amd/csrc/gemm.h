// gemm.h -- batched, tiled MFMA GEMM with per-tile k-ranges (the N^3 engine).
//
//   C[b] (M x N) = beta * C[b] + alpha * Aop[b] (M x K) * Bop[b] (K x N),   b = blockIdx.y
//
// Every O(N^3) stage of the GP path is one or more launches of this kernel
// (see plan.h): the trsm of the blocked Cholesky (as a product with the inverse of
// the diagonal block), its syrk trailing update, the triangular inverse, the
// W^T W product that yields (K + sigma^2 I)^-1, and the K* solves of predict.
//
// Design (CDNA4, 64-wide waves):
//   * block tile 128 x 128, 256 threads = 4 waves, each wave a 64 x 64 sub-tile held
//     as 4 x 4 accumulators of v_mfma_f64_16x16x4_f64 (or the f32 form): 128 VGPRs.
//     (A 64 x 64 variant serves launches too small to fill the chip.)
//   * operands are staged global -> registers -> LDS in k-slabs of 16, double
//     buffered, one barrier per slab; the next slab's global loads are issued before
//     the current slab's MFMAs so HBM/L2 latency hides under 64 MFMAs per wave.
//   * an operand may be stored "m-major" ([row][k], k contiguous) or "k-major"
//     ([k][row], row contiguous).  Both are copied straight (16-byte vectors,
//     coalesced) into an LDS image of the same orientation; only the fragment
//     address differs.  LDS strides (17 and 144 elements) make every ds_read_b64
//     fragment read bank-conflict free.
//   * triangular structure is expressed as a per-tile k-range [k0, k1) so that tiles
//     never multiply by the zero half of a triangular operand, and `lower_only`
//     launches only the tiles on or below the diagonal.
#pragma once
#include "common.h"

#include <type_traits>

namespace gpc {

// k-slab per LDS stage: 16 for fp64, 32 for fp32 -- the same 64 KB of LDS per block and the same
// MFMA time per slab (64 x 64 cycles, 128 x 32 cycles), so the per-slab barrier and staging are
// amortised equally in both precisions
template <typename T>
constexpr int BKT_v = sizeof(T) == 4 ? 32 : 16;
// m-major LDS image [BT][k].  fp64: rows padded to 17 elements, filled with 8-byte stores; the fragment reads are
// conflict free, the stores are not (~5e8 conflict cycles per cfg3 step), without costing MFMA issue (measured: the
// swizzled form below is +-0 on cfg3 / cfg5).  fp32 (MM_SWZ): rows of exactly one k-slab (32 floats = 8 chunks of 16
// bytes = 32 banks), chunk j of row r stored at chunk j ^ ((r >> 1) & 7): the copy writes whole 16-byte vectors
// (ds_write_b128: 16 lanes = two rows x 8 chunks = all 64 banks once) and the fragment reads of 16 rows x 4 k hit 64
// distinct banks as well -- cfg4 42.0 -> 40.4 ms (+3.9 %).
template <typename T>
constexpr bool MM_SWZ = sizeof(T) == 4;
template <typename T>
constexpr int LDQ_v = MM_SWZ<T> ? BKT_v<T> : BKT_v<T> + 1;  // stride of an m-major LDS image
// stride of a k-major LDS image [16][BT+16]; elements reserved per operand per stage
constexpr int ldp_of(int BT) { return BT + 16; }
template <typename T>
constexpr int opsz_of(int BT) {
  return BKT_v<T> * (BT + 16) > BT * LDQ_v<T> ? BKT_v<T> * (BT + 16) : BT * LDQ_v<T>;
}

enum { KLO_ZERO = 0, KLO_ROW = 1, KLO_COL = 2 };  // k0 = 0 | ti*128 | tj*128
enum { KHI_FULL = 0, KHI_ROW = 1, KHI_COL = 2 };  // k1 = K | (ti+1)*128 | (tj+1)*128

struct GemmArgs {
  const void* A;
  const void* B;
  void* C;
  long long sA, sB, sC;  // batch strides in elements
  int lda, ldb, ldc;
  int M, N, K;  // multiples of 128
  double alpha;
  int beta;  // 0 or 1
  int klo, khi, lower_only;
  int tiles_m, tiles_n;
  int flags = 0;       // GPC_GEMM_FLAGS: 8 = XCD-affine tile queues in persistent launches
  const unsigned short* rsv = nullptr;  // persistent launches: table of the CUs this launch stays off (cu_reserve below)
  int* ctr = nullptr;  // persistent launches: zeroed device counters the blocks draw tiles from
  int ntiles = 0, batch = 0;
  // EPI = 1 launches (predict): C is not stored; colsq[(b * tiles_m + ti) * N + col] receives the sum over the 128 rows
  // of tile row ti of (alpha * C[row][col])^2 -- the column sums of squares of V = W Ks (gaussian_process.py:1756-1760)
  // per tile row, in a fixed order; a small reduction over the tile rows follows
  double* colsq = nullptr;
};
inline int g_gemm_flags = 8 | 16;  // bit 3: XCD-affine tile queues in persistent launches; bit 4: XCD-aware order of plain launches

// Staging addresses are split into a block-uniform pointer `u` (tile origin, advanced by the
// caller one k-slab at a time: scalar adds only) and a per-thread element offset fixed for the
// whole k-loop, so the loop carries no per-thread 64-bit address arithmetic and the loads use
// the scalar-base + 32-bit-offset form of global_load.
template <typename T, bool KM, int BT, int NT>
__device__ __forceinline__ unsigned stage_toff(int ld, int t) {
  constexpr int VEC = MM<T>::VEC;
  if constexpr (!KM) {  // stored [row][k]
    constexpr int TPR = BKT_v<T> / VEC;  // threads per row
    return (unsigned)(t / TPR) * (unsigned)ld + (unsigned)((t % TPR) * VEC);
  } else {  // stored [k][row]
    constexpr int VPR = BT / VEC;  // vectors per k-row
    return (unsigned)(t / VPR) * (unsigned)ld + (unsigned)((t % VPR) * VEC);
  }
}
// elements between consecutive passes p of one thread (block-uniform)
template <typename T, bool KM, int BT, int NT>
__device__ __forceinline__ size_t stage_pstride(int ld) {
  constexpr int VEC = MM<T>::VEC;
  if constexpr (!KM)
    return (size_t)(NT / (BKT_v<T> / VEC)) * ld;
  else
    return (size_t)(NT / (BT / VEC)) * ld;
}
// One staging pass of an operand: NV 16-byte vectors per thread through raw buffer loads
// (buffer_load_dwordx4 v, voffset, s[rsrc], soffset offen): the descriptor's base is the tile
// origin, the per-thread byte offset `voff` is fixed for the whole k-loop and the slab / pass
// offsets are scalars, so the loop carries no vector address arithmetic at all
// (241 instead of 256 VGPRs in the fp64 128-tile kernel).
// AUX: cache policy of the loads (gfx940+: bit 0 sc0, bit 1 nt, bit 4 sc1); 16 = sc1, L1-bypassing loads for bytes another
// workgroup of the SAME launch has just stored write-through (tools/seam_probe.hip; the product's kernels use 0)
template <typename T, bool KM, int BT, int NT, int AUX = 0>
__device__ __forceinline__ void g2r(typename MM<T>::vec_t (&r)[(BT * BKT_v<T>) / (NT * MM<T>::VEC)],
                                    __amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned pstride_bytes,
                                    unsigned voff) {
  using vec_t = typename MM<T>::vec_t;
  constexpr int NV = (BT * BKT_v<T>) / (NT * MM<T>::VEC);
  static_assert(sizeof(vec_t) == 16, "16-byte staging vectors");
  static_assert(KM ? (NT % (BT / MM<T>::VEC) == 0) : (NT % (BKT_v<T> / MM<T>::VEC) == 0), "pass layout");
#pragma unroll
  for (int p = 0; p < NV; ++p) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + p * pstride_bytes, AUX);
    r[p] = *reinterpret_cast<const vec_t*>(&v);
  }
}

template <typename T, bool KM, int BT, int NT>
__device__ __forceinline__ void r2s(T* __restrict__ s,
                                    const typename MM<T>::vec_t (&r)[(BT * BKT_v<T>) / (NT * MM<T>::VEC)],
                                    int t) {
  using vec_t = typename MM<T>::vec_t;
  constexpr int VEC = MM<T>::VEC;
  constexpr int NV = (BT * BKT_v<T>) / (NT * VEC);
  constexpr int LDP = ldp_of(BT);
  if constexpr (!KM) {
    constexpr int TPR = BKT_v<T> / VEC;
    constexpr int RPP = NT / TPR;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int row = t / TPR + p * RPP;
      const int kc = (t % TPR) * VEC;
      if constexpr (MM_SWZ<T>) {
        *reinterpret_cast<vec_t*>(s + row * LDQ_v<T> + (((t % TPR) ^ ((row >> 1) & 7)) * VEC)) = r[p];
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[row * LDQ_v<T> + kc + e] = r[p][e];
      }
    }
  } else {
    constexpr int VPR = BT / VEC;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int v = t + NT * p;
      const int k = v / VPR;
      const int mc = (v % VPR) * VEC;
      *reinterpret_cast<vec_t*>(s + k * LDP + mc) = r[p];
    }
  }
}

template <typename T, bool KM, int BT>
__device__ __forceinline__ T frag(const T* __restrict__ s, int r0, int kk, int lane) {
  if constexpr (!KM) {
    if constexpr (MM_SWZ<T>) {
      constexpr int VEC = MM<T>::VEC;
      const int k = kk + (lane >> 4), l15 = lane & 15;  // r0 is a multiple of 16: the row's key is (l15 >> 1) & 7
      return s[(r0 + l15) * LDQ_v<T> + (((k / VEC) ^ ((l15 >> 1) & 7)) * VEC) + (k % VEC)];
    } else {
      return s[(r0 + (lane & 15)) * LDQ_v<T> + kk + (lane >> 4)];
    }
  }
  else
    return s[(kk + (lane >> 4)) * ldp_of(BT) + r0 + (lane & 15)];
}

__device__ __forceinline__ void tri_tile(int tile, int& ti, int& tj) {
  int i = (int)((sqrtf(8.f * (float)tile + 1.f) - 1.f) * 0.5f);
  while (i * (i + 1) / 2 > tile) --i;
  while ((i + 1) * (i + 2) / 2 <= tile) ++i;
  ti = i;
  tj = tile - i * (i + 1) / 2;
}

// BT = block tile (128: 4 waves x 64x64; 64: 4 waves x 32x32, for launches too small to
// fill 256 CUs with 128-tiles).  Triangular k-ranges stay 128-granular in both.
// NW = waves per block: 4 (2 x 2 waves of BT/2 x BT/2) or 8 (2 x 4 waves of BT/2 x BT/4:
// half the accumulators per wave, so twice the waves per SIMD to cover staging and barriers).
// HO ("hand-off"): the result is stored with agent-scope (write-through, sc1) stores, the form MI355X_MICROARCH.md
// prescribes for bytes that cross workgroups INSIDE one launch.  HO = 1 (tools/seam_probe.hip): operands through sc1
// (L1-bypassing) loads as well, no fence needed.  HO = 2 (dag.h): plain operand loads -- the consumer has run ONE
// agent-scope acquire after it learnt that its predecessors are done.  (sc1 loads of lines a write-through store has
// just dropped from every L2 cost a trip to memory per k-slab: a 64-tile task of k = 128 took 10-40 us, a 128-tile
// task of k = 1152 580 us.)
#ifdef GPC_TILE_TRACE
// one-off diagnosis (a library built with -DGPC_TILE_TRACE): wall-clock ticks summed per phase of gemm_tile for the
// hand-off tiles -- [BT == 64][prologue, k-loop, epilogue stores issued, count]
__device__ unsigned long long g_tile_trace[2][4];
#define TILE_TS(k) do { if (HO && threadIdx.x == 0) { const long long _n = wall_clock64(); atomicAdd(&g_tile_trace[BT == 64][k], (unsigned long long)(_n - _ts)); _ts = _n; } } while (0)
#else
#define TILE_TS(k) do { } while (0)
#endif
// BTN (round 5): tile COLUMNS when they differ from the tile rows BT -- the rectangular 128 x 64 tile of launches that are
// too small for 128-tiles (one wave of tiles, as long as its longest) and run L2-bound as 64-tiles: half the tile-count
// granularity of the one, 25 % fewer operand bytes per flop than the other.  Triangular k-ranges stay 128-granular, every
// element keeps its k-order: the bits do not depend on the tile shape.
template <typename T, bool AKM, bool BKM, int BT, int NW, int HO = 0, int EPI = 0, int BTN = BT>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const int bx, const int by, T* __restrict__ smem) {
#ifdef GPC_TILE_TRACE
  long long _ts = wall_clock64();
  if (HO && threadIdx.x == 0) atomicAdd(&g_tile_trace[BT == 64][3], 1ull);
#endif
  using acc_t = typename MM<T>::acc_t;
  using vec_t = typename MM<T>::vec_t;
  constexpr int NT = 64 * NW;
  constexpr int NVA = (BT * BKT_v<T>) / (NT * MM<T>::VEC), NVB = (BTN * BKT_v<T>) / (NT * MM<T>::VEC);
  constexpr int OPSZA = opsz_of<T>(BT), OPSZB = opsz_of<T>(BTN), STG = OPSZA + OPSZB;  // one LDS stage: [A | B]
  constexpr int WCOLS = NW / 2;        // waves along n
  constexpr int WTM = BT / 2;          // wave tile rows
  constexpr int WTN = BTN / WCOLS;     // wave tile cols
  constexpr int MRM = WTM / 16, MRN = WTN / 16;
  static_assert(NVA >= 1 && NVB >= 1 && MRM >= 1 && MRN >= 1, "tile shape");
  const int t = threadIdx.x;
  const int lane = t & 63, w = t >> 6, wr = w / WCOLS, wc = w % WCOLS;

  // Tile order = dispatch order.  Tiles differ in k-length when the k-range is
  // triangular, so the longest tiles go first (LPT) and the short ones fill the tail;
  // the major index is the one the k-range depends on, which also keeps consecutive
  // blocks on one operand panel.
  int ti, tj;
  if (g.lower_only) {
    if constexpr (BTN == BT) {
      tri_tile(bx, ti, tj);  // ti ascending: longest first for KLO_ROW (lauum), uniform for syrk
    } else {
      static_assert(BT == 2 * BTN, "lower tiles of a rectangular launch: two tile columns per tile row");
      // tile row ti holds the columns 0 .. 2 ti + 1: ti (ti + 1) tiles come before it
      int i = (int)((sqrtf(4.f * (float)bx + 1.f) - 1.f) * 0.5f);
      while (i * (i + 1) > bx) --i;
      while ((i + 1) * (i + 2) <= bx) ++i;
      ti = i;
      tj = bx - i * (i + 1);
    }
  } else if (g.khi == KHI_COL) {
    tj = g.tiles_n - 1 - bx / g.tiles_m;
    ti = bx % g.tiles_m;
  } else if (g.klo == KLO_COL) {
    tj = bx / g.tiles_m;
    ti = bx % g.tiles_m;
  } else if (g.khi == KHI_ROW) {
    ti = g.tiles_m - 1 - bx / g.tiles_n;
    tj = bx % g.tiles_n;
  } else {
    ti = bx / g.tiles_n;
    tj = bx % g.tiles_n;
  }
  const int m0 = ti * BT, n0 = tj * BTN;
  const int m128 = (m0 / TILE) * TILE, n128 = (n0 / TILE) * TILE;
  int k0 = g.klo == KLO_ROW ? m128 : (g.klo == KLO_COL ? n128 : 0);
  int k1 = g.khi == KHI_ROW ? m128 + TILE : (g.khi == KHI_COL ? n128 + TILE : g.K);
  if (k1 > g.K) k1 = g.K;
  const int nk = (k1 - k0) / BKT_v<T>;

  const T* __restrict__ A = reinterpret_cast<const T*>(g.A) + (size_t)by * g.sA;
  const T* __restrict__ B = reinterpret_cast<const T*>(g.B) + (size_t)by * g.sB;
  T* __restrict__ C = reinterpret_cast<T*>(g.C) + (size_t)by * g.sC;

  acc_t acc[MRM][MRN];
#pragma unroll
  for (int i = 0; i < MRM; ++i)
#pragma unroll
    for (int j = 0; j < MRN; ++j) acc[i][j] = acc_t{0, 0, 0, 0};
  // fp32: two-level accumulation.  The MFMA adds its products to the running sum one k-step after
  // the other, so a K-long fp32 sum carries an error of ~sqrt(K) eps |sum| (measured: the fp32
  // W^T W at N = 1408 was 3.7e-6 |A^-1| off and the gradient contraction, a difference of large
  // terms, 1.5e-3 instead of 3e-5).  Every two k-slabs (64 k) the accumulators are folded into a second
  // set and restart from zero: the first level only ever holds a 64-term partial sum, the second
  // takes K/64 additions.
  constexpr bool TWO_LEVEL = sizeof(T) == 4;
  acc_t acc2[TWO_LEVEL ? MRM : 1][TWO_LEVEL ? MRN : 1];
  if constexpr (TWO_LEVEL) {
#pragma unroll
    for (int i = 0; i < MRM; ++i)
#pragma unroll
      for (int j = 0; j < MRN; ++j) acc2[i][j] = acc_t{0, 0, 0, 0};
  }

  if (nk > 0) {
    vec_t ra[NVA], rb[NVB];
    // byte offsets: per thread (fixed) and per pass (block-uniform), both inside ONE k-slab of the operand (at most 128
    // rows of an m-major image or 16 / 32 k-rows of a k-major one: megabytes).  From slab to slab an m-major operand
    // moves by BKT elements -- a scalar offset that stays below K * sizeof(T) -- while a k-major operand moves by BKT
    // whole rows: its 64-bit base pointer is advanced (two scalar adds) and the descriptor rebuilt, so no 32-bit byte
    // offset ever spans a panel and the operand may be of any size (round 6; rounds 1-5 carried a 32-bit slab offset
    // over the whole k-major panel, which capped N at 16384 in fp64).
    const unsigned toa = stage_toff<T, AKM, BT, NT>(g.lda, t) * (unsigned)sizeof(T),
                   tob = stage_toff<T, BKM, BTN, NT>(g.ldb, t) * (unsigned)sizeof(T);
    const unsigned psa = (unsigned)(stage_pstride<T, AKM, BT, NT>(g.lda) * sizeof(T)),
                   psb = (unsigned)(stage_pstride<T, BKM, BTN, NT>(g.ldb) * sizeof(T));
    const T* pa = AKM ? A + (size_t)k0 * g.lda + m0 : A + (size_t)m0 * g.lda + k0;
    const T* pb = BKM ? B + (size_t)k0 * g.ldb + n0 : B + (size_t)n0 * g.ldb + k0;
    const size_t rowsa = (size_t)BKT_v<T> * g.lda, rowsb = (size_t)BKT_v<T> * g.ldb;  // elements per slab of a k-major operand
    __amdgpu_buffer_rsrc_t rsa = make_rsrc(pa), rsb = make_rsrc(pb);
    unsigned ua = 0, ub = 0;  // slab offsets of m-major operands (bytes)
    auto next_slab = [&]() {
      if constexpr (AKM) {
        pa += rowsa;
        rsa = make_rsrc(pa);
      } else {
        ua += (unsigned)(BKT_v<T> * sizeof(T));
      }
      if constexpr (BKM) {
        pb += rowsb;
        rsb = make_rsrc(pb);
      } else {
        ub += (unsigned)(BKT_v<T> * sizeof(T));
      }
    };
    constexpr int AUX = HO == 1 ? 16 : 0;  // HO = 2 (dag.h): plain loads behind the consumer's agent-scope acquire
    g2r<T, AKM, BT, NT, AUX>(ra, rsa, ua, psa, toa);
    g2r<T, BKM, BTN, NT, AUX>(rb, rsb, ub, psb, tob);
    r2s<T, AKM, BT, NT>(smem, ra, t);
    r2s<T, BKM, BTN, NT>(smem + OPSZA, rb, t);
    // slab 1 is in flight while slab 0 is multiplied (k-ranges are 128-granular: nk is a multiple of 128 / BKT >= 4)
    next_slab();
    g2r<T, AKM, BT, NT, AUX>(ra, rsa, ua, psa, toa);
    g2r<T, BKM, BTN, NT, AUX>(rb, rsb, ub, psb, tob);
    __syncthreads();
    TILE_TS(0);

    // Software pipeline of one k-slab (KS k-steps of MRM x MRN MFMAs) out of LDS stage CUR.
    // Measured on MI355X (tools/mfma_ladder.hip, profiles/r01g_mfma_ladder.txt): the MFMA pipe
    // stays ~98% busy only if no two memory instructions are issued back to back, so
    //   * fragments are double buffered in registers, the LDS reads of k-step s+1 go out ahead
    //     of the MFMAs of k-step s;
    //   * the LDS writes of the next slab (stage CUR^1) are spread between the MFMAs of k-step KS-2,
    //   * the block barrier sits between the last two k-steps, so the first fragments of the next
    //     slab are read during the last k-step and no k-step waits on LDS latency;
    //   * the global loads of the slab after next are spread between the MFMAs of the last k-step
    //     (in flight during the first k-steps of the next slab; consumed by its k-step KS-2).
    // WR: a next slab exists (ra/rb hold it); LD: a slab after next exists.
    T af[2][MRM], bf[2][MRN];
    auto load_frags = [&](int set, const T* a_s, const T* b_s, int kk) {
#pragma unroll
      for (int i = 0; i < MRM; ++i) af[set][i] = frag<T, AKM, BT>(a_s, wr * WTM + i * 16, kk, lane);
#pragma unroll
      for (int j = 0; j < MRN; ++j) bf[set][j] = frag<T, BKM, BTN>(b_s, wc * WTN + j * 16, kk, lane);
    };
    auto mfmas = [&](int set) {
#pragma unroll
      for (int i = 0; i < MRM; ++i)
#pragma unroll
        for (int j = 0; j < MRN; ++j) acc[i][j] = MM<T>::mma(af[set][i], bf[set][j], acc[i][j]);
    };
    constexpr int NMEM = NVA + NVB;                 // memory instructions per slab and direction
    constexpr int PER = (MRM * MRN) / NMEM;         // MFMAs between two of them
    constexpr int REM = MRM * MRN - PER * NMEM;     // (rectangular tile: 8 MFMAs, 6 memory instructions -- the last two MFMAs follow)
    static_assert(PER >= 1, "interleave pattern");
    constexpr int KS = BKT_v<T> / 4;  // k-steps per slab: 4 (fp64) or 8 (fp32)
    static_assert(KS >= 4 && KS % 2 == 0, "fragment sets alternate per k-step and wrap per slab");
    auto slab = [&](auto cur_c, auto wr_c, auto ld_c) {
      constexpr int CUR = decltype(cur_c)::value;
      constexpr bool WR = decltype(wr_c)::value, LD = decltype(ld_c)::value;
      const T* a_s = smem + CUR * STG;
      const T* b_s = a_s + OPSZA;
      T* a_n = smem + (CUR ^ 1) * STG;
#pragma unroll
      for (int ks = 0; ks < KS - 2; ++ks) {
        load_frags((ks + 1) & 1, a_s, b_s, 4 * (ks + 1));
        mfmas(ks & 1);
        __builtin_amdgcn_sched_group_barrier(0x100, MRM + MRN, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, MRM * MRN, 0);
      }
      // k-step KS-2: the LDS writes of the next slab between its MFMAs
      load_frags(1, a_s, b_s, 4 * (KS - 1));
      mfmas(0);
      if constexpr (WR) {
        r2s<T, AKM, BT, NT>(a_n, ra, t);
        r2s<T, BKM, BTN, NT>(a_n + OPSZA, rb, t);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, MRM + MRN, 0);
#pragma unroll
      for (int q = 0; q < NMEM; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
        if constexpr (WR) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      if constexpr (REM > 0) __builtin_amdgcn_sched_group_barrier(0x008, REM, 0);
      __syncthreads();
      // k-step KS-1: the global loads of the slab after next between its MFMAs
      if constexpr (LD) {
        next_slab();
        g2r<T, AKM, BT, NT, AUX>(ra, rsa, ua, psa, toa);
        g2r<T, BKM, BTN, NT, AUX>(rb, rsb, ub, psb, tob);
      }
      if constexpr (WR) load_frags(0, a_n, a_n + OPSZA, 0);
      mfmas(1);
      if constexpr (WR) __builtin_amdgcn_sched_group_barrier(0x100, MRM + MRN, 0);
#pragma unroll
      for (int q = 0; q < NMEM; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
        if constexpr (LD) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      if constexpr (REM > 0) __builtin_amdgcn_sched_group_barrier(0x008, REM, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    load_frags(0, smem, smem + OPSZA, 0);
    int it = 0;
    for (; it + 3 < nk; it += 2) {
      slab(I0{}, std::true_type{}, std::true_type{});
      slab(I1{}, std::true_type{}, std::true_type{});
      if constexpr (TWO_LEVEL) {
#pragma unroll
        for (int i = 0; i < MRM; ++i)
#pragma unroll
          for (int j = 0; j < MRN; ++j) {
            acc2[i][j] += acc[i][j];
            acc[i][j] = acc_t{0, 0, 0, 0};
          }
      }
    }
    // nk is even and >= 4, so exactly two slabs are left: the last one that
    // still stages a successor, and the last one
    slab(I0{}, std::true_type{}, std::false_type{});
    slab(I1{}, std::false_type{}, std::false_type{});
  }

  TILE_TS(1);
  const T alpha = (T)g.alpha;
  if constexpr (EPI == 1) {
    // column sums of squares of this tile instead of the tile itself.  A lane holds 4 rows x MRN columns of each of
    // its MRM fragments: rows first in the lane (fragment, then register), then the four lane groups, then the wave rows
    static_assert(NW == 4 && BT == 128 && BTN == BT, "the reduction is laid out for 2 x 2 waves of 64 x 64");
    double cs[MRN];
#pragma unroll
    for (int j = 0; j < MRN; ++j) {
      double sq = 0.0;
#pragma unroll
      for (int i = 0; i < MRM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double v;
          if constexpr (TWO_LEVEL)
            v = (double)(alpha * (acc[i][j][r] + acc2[i][j][r]));
          else
            v = (double)(alpha * acc[i][j][r]);
          sq = fma(v, v, sq);
        }
      sq += __shfl_xor(sq, 16, 64);
      sq += __shfl_xor(sq, 32, 64);
      cs[j] = sq;
    }
    double* red = reinterpret_cast<double*>(smem);  // [2 wave rows][BT columns]; the k-loop is done with the stages
    __syncthreads();
    if ((lane >> 4) == 0) {
#pragma unroll
      for (int j = 0; j < MRN; ++j) red[wr * BT + wc * WTN + j * 16 + (lane & 15)] = cs[j];
    }
    __syncthreads();
    if (t < BT) g.colsq[((size_t)by * g.tiles_m + ti) * (size_t)g.N + n0 + t] = red[t] + red[BT + t];
    __syncthreads();  // (a persistent block restages LDS for its next tile)
    return;
  }
  // beta = 1: the old values of one accumulator row (MRN x 4 per lane) are loaded as a batch before
  // any store of that row -- interleaved load/add/store through the same pointer serialises into
  // MRM x MRN x 4 dependent memory round trips (the 64-tile syrk of a 256-node took 12 us
  // instead of 7.5 us for that reason)
#pragma unroll
  for (int i = 0; i < MRM; ++i) {
    T old[MRN][4];
    if (g.beta) {
#pragma unroll
      for (int j = 0; j < MRN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wr * WTM + i * 16 + MM<T>::row_of(lane, r);
          const int col = n0 + wc * WTN + j * 16 + (lane & 15);
          if constexpr (HO == 1)
            old[j][r] = __hip_atomic_load(&C[(size_t)row * g.ldc + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else
            old[j][r] = C[(size_t)row * g.ldc + col];
        }
    }
#pragma unroll
    for (int j = 0; j < MRN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * WTM + i * 16 + MM<T>::row_of(lane, r);
        const int col = n0 + wc * WTN + j * 16 + (lane & 15);
        T v;
        if constexpr (TWO_LEVEL)
          v = alpha * (acc[i][j][r] + acc2[i][j][r]);
        else
          v = alpha * acc[i][j][r];
        if (g.beta) v += old[j][r];
        if constexpr (HO)
          __hip_atomic_store(&C[(size_t)row * g.ldc + col], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
          C[(size_t)row * g.ldc + col] = v;
      }
  }
  TILE_TS(2);
}

template <typename T, bool AKM, bool BKM, int BT, int NW, int EPI = 0, int BTN = BT>
__global__ __launch_bounds__(64 * NW, NW / 2) void gemm_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) T smem[2 * (opsz_of<T>(BT) + opsz_of<T>(BTN))];
  int bx = blockIdx.x, by = blockIdx.y;
  if ((g.flags & 16) && gridDim.y >= 8) {
    // (with fewer than 8 samples a contiguous range is a piece of ONE sample's longest-first tile list: the
    // XCDs would get unequal work -- cfg4, one sample: -6.5 %)
    // XCD-aware work order: workgroups are dealt round-robin over the 8 XCDs in dispatch order (x fastest), so
    // workgroup L runs on XCD L % 8.  Give XCD x the contiguous range [x total/8, (x+1) total/8) of the
    // (sample, tile) items: its blocks then work on the same one or two samples and share panels in its L2.
    const int total = gridDim.x * gridDim.y, per = total >> 3;
    const int L = blockIdx.y * gridDim.x + blockIdx.x;
    if (L < (per << 3)) {
      const int wk = (L & 7) * per + (L >> 3);
      bx = wk % (int)gridDim.x;
      by = wk / (int)gridDim.x;
    }
  }
  gemm_tile<T, AKM, BKM, BT, NW, 0, EPI, BTN>(g, bx, by, smem);
}

// Two independent products in ONE launch (plan.h: the syrk A22 -= T21 T21^T and the inverse product U = T21 W11 of
// a node both wait for T21 only): workgroups [0, n1) of a sample are tiles of g1, the rest tiles of g2.  The deep
// levels of the recursion are bound by the ~5 us every dependent launch costs, not by their flops.
template <typename T, bool A1, bool B1, bool A2, bool B2, int BT, int NW>
__global__ __launch_bounds__(64 * NW, NW / 2) void gemm_dual_kernel(GemmArgs g1, GemmArgs g2, int n1) {
  __shared__ __attribute__((aligned(16))) T smem[4 * opsz_of<T>(BT)];
  int bx = blockIdx.x, by = blockIdx.y;
  if ((g1.flags & 16) && gridDim.y >= 8) {  // XCD-aware work order, as in gemm_kernel
    const int total = gridDim.x * gridDim.y, per = total >> 3;
    const int L = blockIdx.y * gridDim.x + blockIdx.x;
    if (L < (per << 3)) {
      const int wk = (L & 7) * per + (L >> 3);
      bx = wk % (int)gridDim.x;
      by = wk / (int)gridDim.x;
    }
  }
  if (bx < n1)
    gemm_tile<T, A1, B1, BT, NW>(g1, bx, by, smem);
  else
    gemm_tile<T, A2, B2, BT, NW>(g2, bx - n1, by, smem);
}

// Persistent form for launches with more tiles than block slots: a fixed grid of blocks pulls
// (tile, sample) pairs off a device counter (longest tiles of every sample first).  The grid is
// `g_persist_spare` blocks short of two per CU, which leaves that many CUs with one resident
// GEMM block instead of two: room (registers, LDS) for the leaf / deep-level launches of the
// other sample group, which otherwise wait for this whole launch to drain
// (profiles/r01g_coresidency_probe.txt).
// XCD affinity (default; GPC_XCD_AFFINE=0 clears g.flags & 8): the (tile, sample) pairs are split into 8 queues by sample index
// and a block serves the queue of the XCD it runs on first (HW_REG_XCC_ID), stealing from the
// others when its own is empty.  Blocks that share an L2 then work on tiles of the same
// sample, consecutive tiles of a sample share an operand panel, and the panel is read
// from HBM / Infinity Cache once per XCD instead of once per tile.
template <typename T, bool AKM, bool BKM, int BT, int NW, int EPI = 0>
__global__ __launch_bounds__(64 * NW, NW / 2) void gemm_persist_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) T smem[4 * opsz_of<T>(BT)];
  __shared__ int next_tile;
  if (g.rsv && cu_reserve_bail(g.rsv, g.ctr)) return;
  if (!(g.flags & 8)) {
    const int total = g.ntiles * g.batch;
    for (;;) {
      if (threadIdx.x == 0) next_tile = atomicAdd(g.ctr, 1);
      __syncthreads();
      const int idx = __builtin_amdgcn_readfirstlane(next_tile);
      __syncthreads();  // everyone holds idx before thread 0 may overwrite it; also fences the LDS stages
      if (idx >= total) break;
      gemm_tile<T, AKM, BKM, BT, NW, 0, EPI>(g, idx / g.batch, idx % g.batch, smem);
    }
    return;
  }
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int q0 = (int)(xcc & (NQ - 1));
  // queue q: samples q, q + 8, ... when there are at least 8 samples; with fewer, sample q % batch
  // and every nclass-th tile of it starting at tile q / batch (interleaved, so that each queue
  // keeps the longest-first order), nclass = 8 / batch
  const int nclass = g.batch >= NQ ? 1 : NQ / g.batch;
  for (int a = 0; a < NQ; ++a) {
    const int q = (q0 + a) & (NQ - 1);
    int nsq, cls = 0, smp = q;
    if (g.batch >= NQ) {
      nsq = (g.batch - q + NQ - 1) / NQ;
    } else {
      if (q >= g.batch * nclass) continue;
      nsq = 1;
      smp = q % g.batch;
      cls = q / g.batch;
    }
    if (nsq <= 0) continue;
    const int total = ((g.ntiles - cls + nclass - 1) / nclass) * nsq;
    for (;;) {
      if (threadIdx.x == 0) next_tile = atomicAdd(g.ctr + q, 1);
      __syncthreads();
      const int idx = __builtin_amdgcn_readfirstlane(next_tile);
      __syncthreads();
      if (idx >= total) break;
      // sample-major inside a queue: all tiles of the queue's first sample (longest first), then its second,
      // ... so the 64 blocks of an XCD work on neighbouring tiles of ONE sample and share its operand panels
      // through their L2.  (Interleaving the samples tile by tile, as round 1 did, put 8 samples' panels
      // into every XCD's L2 at once: cfg3 W^T W 5.58 -> 5.36 ms, step 20.9 -> 20.4 ms; cfg5 552 -> 531 ms.)
      const int ntq = total / nsq;
      const int sq = idx / ntq, tq = idx - sq * ntq;
      gemm_tile<T, AKM, BKM, BT, NW, 0, EPI>(g, cls + tq * nclass, smp + NQ * sq, smem);
    }
  }
}

inline int g_persist_spare = 0;     // tunable: GPC_PERSIST_SPARE (block slots a persistent launch leaves free)
inline int g_block_slots = 512;     // two 128-tile blocks per CU (set from the device's CU count)
template <typename T, int BT, int NW>
inline hipError_t launch_gemm_bt(hipStream_t st, GemmArgs g, bool akm, bool bkm, int batch, int* ctr = nullptr,
                                 const unsigned short* reserve = nullptr) {
  const int tm = g.M / BT, tn = g.N / BT;
  g.tiles_m = tm;
  g.tiles_n = tn;
  g.flags = g_gemm_flags;
  const int ntiles = g.lower_only ? tm * (tm + 1) / 2 : tm * tn;
  if (ntiles <= 0 || batch <= 0) return hipSuccess;
  g.ntiles = ntiles;
  g.batch = batch;
  g.ctr = ctr;
  const unsigned dyn = 0u;
  // block slots of the chip for this tile size: two 128-tile workgroups per CU, four 64-tile ones
  const int cap = (BT == 128 ? g_block_slots : 2 * g_block_slots) - g_persist_spare;
  g.rsv = reserve;
  if (ctr && (BT == 128 || reserve) && cap > 0 && ((long long)ntiles * batch > cap || reserve)) {
    // reserved launches: 128-tiles keep round 2's oversized grid (its surplus drains through the reserved CUs, of which
    // every shader engine has some); 64-tiles (round 5, independent pipelines) take the exact-fit grid -- the workgroups
    // that land on reserved CUs return, nothing stays pending in the dispatcher
    dim3 grid(reserve ? (BT == 128 ? cap + 96 : cap) : cap), block(64 * NW);
    if (!akm && !bkm)
      hipLaunchKernelGGL((gemm_persist_kernel<T, false, false, BT, NW>), grid, block, dyn, st, g);
    else if (!akm && bkm)
      hipLaunchKernelGGL((gemm_persist_kernel<T, false, true, BT, NW>), grid, block, dyn, st, g);
    else if (akm && bkm)
      hipLaunchKernelGGL((gemm_persist_kernel<T, true, true, BT, NW>), grid, block, dyn, st, g);
    else
      hipLaunchKernelGGL((gemm_persist_kernel<T, true, false, BT, NW>), grid, block, dyn, st, g);
    return hipGetLastError();
  }
  dim3 grid(ntiles, batch), block(64 * NW);
  if (!akm && !bkm)
    hipLaunchKernelGGL((gemm_kernel<T, false, false, BT, NW>), grid, block, dyn, st, g);
  else if (!akm && bkm)
    hipLaunchKernelGGL((gemm_kernel<T, false, true, BT, NW>), grid, block, dyn, st, g);
  else if (akm && bkm)
    hipLaunchKernelGGL((gemm_kernel<T, true, true, BT, NW>), grid, block, dyn, st, g);
  else
    hipLaunchKernelGGL((gemm_kernel<T, true, false, BT, NW>), grid, block, dyn, st, g);
  return hipGetLastError();
}

#ifdef GPC_EXPERIMENTS
// 128 x 64 tiles (plain launches): see gemm_tile's BTN
template <typename T>
inline hipError_t launch_gemm_rect(hipStream_t st, GemmArgs g, bool akm, bool bkm, int batch) {
  constexpr int BT = 128, BTN = 64;
  g.tiles_m = g.M / BT;
  g.tiles_n = g.N / BTN;
  g.flags = g_gemm_flags;
  const int ntiles = g.lower_only ? g.tiles_m * (g.tiles_m + 1) : g.tiles_m * g.tiles_n;
  if (ntiles <= 0 || batch <= 0) return hipSuccess;
  g.ntiles = ntiles;
  g.batch = batch;
  g.ctr = nullptr;
  g.rsv = nullptr;
  dim3 grid(ntiles, batch), block(256);
  if (!akm && !bkm)
    hipLaunchKernelGGL((gemm_kernel<T, false, false, BT, 4, 0, BTN>), grid, block, 0, st, g);
  else if (!akm && bkm)
    hipLaunchKernelGGL((gemm_kernel<T, false, true, BT, 4, 0, BTN>), grid, block, 0, st, g);
  else if (akm && bkm)
    hipLaunchKernelGGL((gemm_kernel<T, true, true, BT, 4, 0, BTN>), grid, block, 0, st, g);
  else
    hipLaunchKernelGGL((gemm_kernel<T, true, false, BT, 4, 0, BTN>), grid, block, 0, st, g);
  return hipGetLastError();
}
// 128 x 128 tiles with EIGHT waves (2 x 4 waves of 64 x 32; plain launches): the 128-tile's operand traffic per flop at the
// 64-tile kernel's four waves per SIMD -- tried for the same launches as the rectangular tile (g_rect_mode = 1)
template <typename T>
inline hipError_t launch_gemm_w8(hipStream_t st, GemmArgs g, bool akm, bool bkm, int batch) {
  constexpr int BT = 128;
  g.tiles_m = g.M / BT;
  g.tiles_n = g.N / BT;
  g.flags = g_gemm_flags;
  const int ntiles = g.lower_only ? g.tiles_m * (g.tiles_m + 1) / 2 : g.tiles_m * g.tiles_n;
  if (ntiles <= 0 || batch <= 0) return hipSuccess;
  g.ntiles = ntiles;
  g.batch = batch;
  g.ctr = nullptr;
  g.rsv = nullptr;
  dim3 grid(ntiles, batch), block(512);
  if (!akm && !bkm)
    hipLaunchKernelGGL((gemm_kernel<T, false, false, BT, 8>), grid, block, 0, st, g);
  else if (!akm && bkm)
    hipLaunchKernelGGL((gemm_kernel<T, false, true, BT, 8>), grid, block, 0, st, g);
  else if (akm && bkm)
    hipLaunchKernelGGL((gemm_kernel<T, true, true, BT, 8>), grid, block, 0, st, g);
  else
    hipLaunchKernelGGL((gemm_kernel<T, true, false, BT, 8>), grid, block, 0, st, g);
  return hipGetLastError();
}
// launches of at least this many 128-tiles (times samples) that are still "small" take the rectangular tile; 0: never
inline int g_rect_min_blocks = 0;  // tunable: gpc_set_option("rect_min", n) / GPC_RECT_MIN
inline int g_rect_mode = 0;        // 0: 128 x 64 tiles of four waves; 1: 128 x 128 tiles of eight waves ("rect_mode")

#else
constexpr int g_rect_min_blocks = 0;
#endif  // GPC_EXPERIMENTS

// Launches that cannot put ~2 blocks of 128-tiles on every CU use 64-tiles (4x the blocks,
// a quarter of the work each): the deep levels of the recursion are latency-, not
// throughput-bound.  force_bt: 0 = choose, 64 / 128 = as given (tests).
inline int g_small_launch_blocks = 1100;  // tunable: GPC_SMALL_BLOCKS
inline bool g_dual_launch = true;         // tunable: GPC_DUAL (plan.h: syrk + U of a node in one launch)
// reserved_small_bt: tile of a RESERVED launch below that threshold -- 128 (the deferred products) or, experiments build
// only, 64 (independent pipelines); carried per launch (it used to be a process-wide global that every pipeline of every
// context wrote: contexts on different threads raced on it)
template <typename T>
inline hipError_t launch_gemm(hipStream_t st, GemmArgs g, bool akm, bool bkm, int batch, int force_bt = 0,
                             int* ctr = nullptr, const unsigned short* reserve = nullptr, int reserved_small_bt = 128) {
  const int tm = g.M / TILE, tn = g.N / TILE;
  const long long blocks128 = (long long)(g.lower_only ? tm * (tm + 1) / 2 : tm * tn) * batch;
  const bool small = force_bt ? (force_bt == 64) : (blocks128 < g_small_launch_blocks);
#ifdef GPC_EXPERIMENTS
  if ((force_bt == 12864 || (!force_bt && small && g_rect_min_blocks > 0 && blocks128 >= g_rect_min_blocks)) && !(reserve && ctr))
    return (g_rect_mode == 1 && force_bt != 12864) ? launch_gemm_w8<T>(st, g, akm, bkm, batch) : launch_gemm_rect<T>(st, g, akm, bkm, batch);
#endif
  if (small && !(reserve && ctr)) return launch_gemm_bt<T, 64, 4>(st, g, akm, bkm, batch);
#ifdef GPC_EXPERIMENTS
  if (small && reserved_small_bt == 64) return launch_gemm_bt<T, 64, 4>(st, g, akm, bkm, batch, ctr, reserve);
#endif
  return launch_gemm_bt<T, 128, 4>(st, g, akm, bkm, batch, ctr, ctr ? reserve : nullptr);
}

// true when launch_gemm would run g as a plain 64-tile launch (the precondition of the dual launch)
inline bool gemm_is_small(const GemmArgs& g, int batch) {
  const int tm = g.M / TILE, tn = g.N / TILE;
  const long long b = (long long)(g.lower_only ? tm * (tm + 1) / 2 : tm * tn) * batch;
  return b < g_small_launch_blocks && !(g_rect_min_blocks > 0 && b >= g_rect_min_blocks);
}
// g1: m-major x m-major (syrk), g2: m-major x k-major (U = T21 W11), both as 64-tile launches in one grid
template <typename T>
inline hipError_t launch_gemm_dual_small(hipStream_t st, GemmArgs g1, GemmArgs g2, int batch) {
  constexpr int BT = 64;
  int n[2];
  GemmArgs* gs[2] = {&g1, &g2};
  for (int i = 0; i < 2; ++i) {
    GemmArgs& g = *gs[i];
    g.tiles_m = g.M / BT;
    g.tiles_n = g.N / BT;
    g.flags = g_gemm_flags;
    g.ntiles = n[i] = g.lower_only ? g.tiles_m * (g.tiles_m + 1) / 2 : g.tiles_m * g.tiles_n;
    g.batch = batch;
    g.ctr = nullptr;
    g.rsv = nullptr;
  }
  if (n[0] + n[1] <= 0 || batch <= 0) return hipSuccess;
  hipLaunchKernelGGL((gemm_dual_kernel<T, false, false, false, true, BT, 4>), dim3(n[0] + n[1], batch), dim3(256), 0, st,
                     g1, g2, n[0]);
  return hipGetLastError();
}

// V = A B with A m-major and B k-major (predict: W Ks), 128-tiles, NOT stored: g.colsq receives the column sums of
// squares per tile row (gemm_tile, EPI = 1).  Persistent XCD-affine queues when the launch has more tiles than block
// slots and `ctr` (CTR_STRIDE zeroed counters) is given.
template <typename T>
inline hipError_t launch_gemm_colsq(hipStream_t st, GemmArgs g, int batch, int* ctr) {
  constexpr int BT = 128;
  g.tiles_m = g.M / BT;
  g.tiles_n = g.N / BT;
  g.flags = g_gemm_flags;
  g.lower_only = 0;
  g.ntiles = g.tiles_m * g.tiles_n;
  if (g.ntiles <= 0 || batch <= 0) return hipSuccess;
  g.batch = batch;
  g.ctr = ctr;
  g.rsv = nullptr;
  const int cap = g_block_slots - g_persist_spare;
  if (ctr && cap > 0 && (long long)g.ntiles * batch > cap)
    hipLaunchKernelGGL((gemm_persist_kernel<T, false, true, BT, 4, 1>), dim3(cap), dim3(256), 0, st, g);
  else
    hipLaunchKernelGGL((gemm_kernel<T, false, true, BT, 4, 1>), dim3(g.ntiles, batch), dim3(256), 0, st, g);
  return hipGetLastError();
}

// algorithmic flops of one launch (for the roofline bookkeeping)
inline double gemm_flops(const GemmArgs& g, int batch) {
  const int tm = g.M / TILE, tn = g.N / TILE;
  double f = 0;
  for (int ti = 0; ti < tm; ++ti)
    for (int tj = 0; tj < tn; ++tj) {
      if (g.lower_only && tj > ti) continue;
      int k0 = g.klo == KLO_ROW ? ti * TILE : (g.klo == KLO_COL ? tj * TILE : 0);
      int k1 = g.khi == KHI_ROW ? (ti + 1) * TILE : (g.khi == KHI_COL ? (tj + 1) * TILE : g.K);
      if (k1 > g.K) k1 = g.K;
      if (k1 > k0) f += 2.0 * TILE * TILE * (k1 - k0);
    }
  return f * batch;
}

}  // namespace gpc
