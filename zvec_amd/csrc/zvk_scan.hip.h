// zvk_scan.hip.h — the distance-scan kernels: 4-wave tile (flat small batches, coarse pass, IVF list scan) and the 8-wave wide flat tile.
// Part of the device code of libzvec_hip (included through scan_kernels.hip.h).
#pragma once
#include "zvk_common.hip.h"

namespace zvk {

// register staging of one k-step: 4 x 16 B of the base slab and NG x 16 B of the query rows per thread
template <int NG>
struct StageRegs {
  f32x4 b0, b1, b2, b3;
  f32x4 q[NG];
};

// NT: the base slab is fetched with non-temporal loads.  The IVF list scan reads every list row ONCE per launch
// while each work-group re-reads its few query rows at every tile; with default-policy loads the 30 GB base stream
// evicts those query lines from the 4 MiB L2 between two tiles (5.7 MB pass through an XCD's L2 per tile time), so
// every query re-read went out to the fabric: +8 % traffic.  Streaming the base around the L2's retention keeps
// the query rows resident: list scan 5.33 -> 4.97 ms at 10M x 768.  (Not for flat scans whose query tiles share
// the base stream THROUGH the L2.)
template <int NG, bool NT>
__device__ __forceinline__ void stage_load(StageRegs<NG> &sr, const float *base, const float *queries,
                                           const uint32_t (&qoff)[NG], uint32_t tile, uint32_t ks, uint32_t dpad,
                                           int tid, uint32_t vrows = TILE_N) {
  const f32x4 *bsrc = reinterpret_cast<const f32x4 *>(base + (size_t)tile * TILE_N * dpad + (size_t)ks * SLAB) + tid;
  if constexpr (NT) {
    // a list's last tile is padded to 128 rows; each of the four loads covers 32 rows of the slab, and a load whose
    // rows are all padding re-reads the first quarter instead (same lines: no new traffic, same number of loads in
    // flight) — the padded columns are masked at admission anyway.  Saves ~2 % of the list bytes at ~2400 rows/list.
    sr.b0 = __builtin_nontemporal_load(bsrc);
    sr.b1 = __builtin_nontemporal_load(bsrc + (vrows > 32 ? 256 : 0));
    sr.b2 = __builtin_nontemporal_load(bsrc + (vrows > 64 ? 512 : 0));
    sr.b3 = __builtin_nontemporal_load(bsrc + (vrows > 96 ? 768 : 0));
  } else {
    sr.b0 = bsrc[0];
    sr.b1 = bsrc[256];
    sr.b2 = bsrc[512];
    sr.b3 = bsrc[768];
  }
#pragma unroll
  for (int i = 0; i < NG; ++i)
    sr.q[i] = *reinterpret_cast<const f32x4 *>(queries + (size_t)(qoff[i] + ks * TILE_K));
}

template <int NG>
__device__ __forceinline__ void stage_store(const StageRegs<NG> &sr, float *Bb, float *Qb, int srow, int sswz,
                                            int tid) {
  f32x4 *bdst = reinterpret_cast<f32x4 *>(Bb) + tid;
  bdst[0] = sr.b0;
  bdst[256] = sr.b1;
  bdst[512] = sr.b2;
  bdst[768] = sr.b3;
#pragma unroll
  for (int i = 0; i < NG; ++i)
    *reinterpret_cast<f32x4 *>(Qb + ((srow + 32 * i) * 8 + sswz) * 4) = sr.q[i];
}

// ---------------------------------------------------------------------------------------------
// The scan kernel.  256 threads = 4 waves; wave w owns tile columns [32w, 32w+32); all waves share the
// ROWS query rows of the work item.  Two matrix-core shapes:
//   M16 = false : ROWS = NG*32, v_mfma_f32_32x32x2_f32   (flat scans, coarse assign: many queries per tile)
//   M16 = true  : ROWS = 32 as two 16-row halves, v_mfma_f32_16x16x4_f32   (IVF list scan: a list is probed
//                 by ~10 queries of the batch; the second half is skipped — wave-uniformly — when the item
//                 has <= 16 query rows, so a 16-row item costs half the matrix-core cycles of a 32x32 tile,
//                 while a list probed by 17..32 queries is still streamed from HBM only once)
// EXCL selects the bitmap-gated variant (the filter word is fetched with the tile, unconditionally, so
// the no-filter variant carries no extra load).
// Persistent loop over work items: static grid-stride for flat, a device work queue for IVF (every
// wave reaches the loop exit: `item` is uniform in the work-group and bounded by a value read once).
// ---------------------------------------------------------------------------------------------
typedef float floatx4 __attribute__((ext_vector_type(4)));

#ifdef ZVK_CLOCK_STAMP
// Diagnostic build only (tools/build_variant.sh clk -DZVK_CLOCK_STAMP): every work-group of the wide flat kernel, and of the IVF list scan, stamps
// the shader clock (s_memtime) and the constant 100 MHz clock (s_memrealtime) when it starts and when it ends; the
// in-kernel clock is d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md "DVFS give-back" item 6).  The stamps go
// to a buffer of their own that nothing else reads.
__device__ unsigned long long zvk_clock_stamps[1024][4];
#endif

template <int NG, bool M16>
struct ScanShape {
  static constexpr int ROWS = M16 ? 32 : NG * QGROUP;
  static constexpr int QLOADS = M16 ? 1 : NG;           // 16-byte query loads per thread per k-step
};

// F16: rows and queries are IEEE half (DT_FP16); products are exact in fp32 and accumulated in fp32 by
// v_mfma_f32_16x16x32_f16 / v_mfma_f32_32x32x16_f16 — the reference converts to fp32 and accumulates in fp32
// too (distance_matrix_accum_fp16.i:554-594).  The staging is byte-identical: a row segment per k-step is
// 128 B either way (32 floats or 64 halves); `dpad` counts 4-byte words per row.
template <int NG, bool M16, bool EXCL, bool F16>
__global__ void __launch_bounds__(256, (NG >= 4 ? 2 : (NG == 2 && !M16 ? 3 : 1))) scan_kernel(const ScanArgs a) {
  constexpr int ROWS = ScanShape<NG, M16>::ROWS;
  constexpr int QL = ScanShape<NG, M16>::QLOADS;
  constexpr int QROWMASK = 31;
  extern __shared__ f32x4 zvk_smem4[];
  float *smem = reinterpret_cast<float *>(zvk_smem4);
  float *Qs = smem;                      // [2][ROWS*32]
  float *Bs = Qs + 2 * ROWS * TILE_K;    // [2][SLAB]
  float *qn_s = Bs + 2 * SLAB;           // [ROWS]
  RowState st;
  st.tau = qn_s + ROWS;
  st.cnt = reinterpret_cast<uint32_t *>(st.tau + ROWS);
  uint32_t *qrow_s = st.cnt + ROWS;
  uint32_t *slot_s = qrow_s + ROWS;
  uint32_t *item_s = slot_s + ROWS;          // [4] work-queue hand-off word
  st.k = a.k;
  st.gt = reinterpret_cast<float *>(item_s + 4);
  st.tq = st.gt + ROWS;
  st.gtau = a.gtau;
  st.qrow = qrow_s;
  st.Ls = st.tq + ROWS;
  st.Li = reinterpret_cast<uint32_t *>(st.Ls + (size_t)ROWS * a.k);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;                 // 32x32x2 operand coordinates
  const int r16 = lane & 15, kq = lane >> 4;              // 16x16x4 operand coordinates
  const int srow = (tid >> 3) & QROWMASK, schunk = tid & 7;   // staging coordinates
  const int sswz = schunk ^ ((srow >> 1) & 7);            // swizzled chunk for the Q image
  const uint32_t dpad = a.dpad, nks = a.nks, k = a.k;

  uint32_t total;
  if (a.mode == 0) total = a.nchunks * a.nqtiles;
  else total = *a.total_items;
#ifdef ZVK_CLOCK_STAMP
  if (tid == 0 && blockIdx.x < 1024 && a.mode == 1) {
    zvk_clock_stamps[blockIdx.x][0] = __builtin_amdgcn_s_memtime();
    zvk_clock_stamps[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
  }
#endif

  for (uint32_t iter = 0;; ++iter) {
    uint32_t item;
    if (a.mode == 0) {
      item = blockIdx.x + iter * gridDim.x;
    } else {
      // dynamic dealing: one returning atomic per item (largest lists first => balanced tail)
      if (tid == 0) item_s[0] = atomicAdd(a.queue, 1u);
      __syncthreads();
      item = item_s[0];
    }
    if (item >= total) break;   // uniform: every wave of the work-group leaves together
    // ---- decode the work item (uniform) ----
    uint32_t tile_begin, tile_end, nrows, rows_valid_total;
    uint64_t dense0 = 0;  // dense position of the first row of the list / store
    uint32_t tile0 = 0;   // first tile of the row range the dense mapping refers to
    uint32_t li = 0, r0 = 0, chunk = 0;
    if (a.mode == 0) {
      uint32_t qtile = item % a.nqtiles;
      chunk = item / a.nqtiles;
      uint32_t ntiles_total = (uint32_t)((a.n + TILE_N - 1) / TILE_N);
      tile_begin = chunk * a.tiles_per_chunk;
      tile_end = min(tile_begin + a.tiles_per_chunk, ntiles_total);
      r0 = qtile * ROWS;
      nrows = min((uint32_t)ROWS, a.nq - r0);
      rows_valid_total = (uint32_t)min((uint64_t)0xffffffffu, a.n);  // rows valid from tile 0
    } else {
      // binary search: item_off[pos] <= item < item_off[pos+1]
      uint32_t lo = 0, hi = a.nlist;
      while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (a.item_off[mid] <= item) lo = mid; else hi = mid;
      }
      uint32_t within = item - a.item_off[lo];
      li = a.list_order[lo];
      uint32_t qcnt = a.list_qoff[li + 1] - a.list_qoff[li];
      uint32_t ngroups = (qcnt + ROWS - 1) / ROWS;
      chunk = within / ngroups;
      uint32_t group = within % ngroups;
      uint32_t lsize = a.list_size[li];
      uint32_t ltiles = (lsize + TILE_N - 1) / TILE_N;
      tile0 = a.list_tile0[li];
      const uint32_t tpc = a.list_tpc[li];
      tile_begin = tile0 + chunk * tpc;
      tile_end = tile0 + min((chunk + 1) * tpc, ltiles);
      r0 = group * ROWS;
      nrows = min((uint32_t)ROWS, qcnt - r0);
      rows_valid_total = lsize;
      dense0 = a.list_dense0[li];
    }

    // ---- per-item LDS state ----
    for (int j = tid; j < ROWS; j += 256) {
      uint32_t qrow, slot;
      if ((uint32_t)j < nrows) {
        if (a.mode == 0) {
          qrow = r0 + j;
          slot = qrow * a.nchunks + chunk;
        } else {
          uint32_t e = a.list_qoff[li] + r0 + j;
          qrow = a.csr_q[e];
          slot = a.csr_slot[e] + chunk;
        }
      } else {
        qrow = (a.mode == 0) ? r0 : a.csr_q[a.list_qoff[li] + r0];  // any valid row; results unused
        slot = IDX_NONE;
      }
      qrow_s[j] = qrow;
      slot_s[j] = slot;
      qn_s[j] = (a.metric == METRIC_L2) ? a.qnorm[qrow] : 0.f;
      st.tau[j] = a.threshold;
      st.gt[j] = a.threshold;
      st.tq[j] = a.threshold;
      st.cnt[j] = 0;
    }
    __syncthreads();

    // staging sources
    uint32_t qoff[QL];   // float offsets into the padded query matrix (host guarantees nq*dpad < 2^32)
#pragma unroll
    for (int i = 0; i < QL; ++i) qoff[i] = qrow_s[srow + 32 * i] * dpad + (uint32_t)schunk * 4u;

    const uint32_t ntiles = tile_end - tile_begin;
    const uint32_t nsteps = ntiles * nks;

    // Software pipeline: the operands of step t travel HBM -> registers sr[t & 1] -> LDS buffer t & 1.
    // PF steps are kept in flight in registers (2 for the HBM-bound small shapes: ~40 KB per
    // work-group on the wire while the matrix cores chew the current step).
    constexpr int PF = (NG <= 2) ? 2 : 1;
    StageRegs<QL> sr[2];
    floatx16 acc[M16 ? 1 : NG];
    floatx4 acc16[4];                 // [row half][column block]
    const bool two = nrows > 16;      // uniform: second 16-row half in use
    if constexpr (M16) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc16[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;
    }

    uint32_t tile = tile_begin, ks = 0;          // coordinates of the step being computed
    uint32_t ptile = tile_begin, pks = 0;        // coordinates of the next step to fetch
    uint32_t fetched = 0;
    auto advance = [&](uint32_t &t_, uint32_t &k_) { if (++k_ == nks) { k_ = 0; ++t_; } };
    if (nsteps > 0) {
      stage_load<QL, M16>(sr[0], a.base, a.queries, qoff, ptile, pks, dpad, tid, rows_valid_total - (ptile - tile0) * TILE_N);
      if (nsteps > 1) advance(ptile, pks);
      fetched = 1;
      stage_store<QL>(sr[0], Bs, Qs, srow, sswz, tid);
      if (PF == 2) {
        stage_load<QL, M16>(sr[1], a.base, a.queries, qoff, ptile, pks, dpad, tid, rows_valid_total - (ptile - tile0) * TILE_N);
        if (nsteps > 2) advance(ptile, pks);
        fetched = 2;
      }
    }
    __syncthreads();

    // per-tile column constants, fetched with every step (same address within a tile: L1/L2 hits) so that
    // the loads in flight per step are the same on every path and the epilogue never drains the pipeline
    float bn0 = 0.f, bn1 = 0.f;
    uint32_t ex0 = 0, ex1 = 0;

    for (uint32_t s0 = 0; s0 < nsteps; s0 += 2) {
#pragma unroll
     for (int u = 0; u < 2; ++u) {
      const uint32_t s = s0 + u;
      if (s >= nsteps) break;                      // uniform
      const int buf = u;                            // == s & 1
      // fetch step s + PF into sr[(s + PF) & 1].  Unconditional on purpose: a conditional load would
      // give the compiler two paths with different numbers of loads in flight and it would then wait
      // for the YOUNGEST set before the LDS store below (vmcnt merges conservatively); past the end
      // the last step is simply fetched again and never used.
      stage_load<QL, M16>(sr[(u + PF) & 1], a.base, a.queries, qoff, ptile, pks, dpad, tid, rows_valid_total - (ptile - tile0) * TILE_N);
      if (fetched + 1 < nsteps) advance(ptile, pks);
      ++fetched;
      {
        const uint32_t pos0 = tile * TILE_N + wave * 32;
        if constexpr (M16) {
          bn0 = a.bnorm[(size_t)pos0 + r16];
          bn1 = a.bnorm[(size_t)pos0 + 16 + r16];
        } else {
          bn0 = a.bnorm[(size_t)pos0 + r];
        }
        if constexpr (EXCL) {
          // (positions in a list's tail padding are clamped: they are masked by rows_valid_total anyway)
          const uint64_t dlast = a.ndense - 1;
          const uint64_t d0 = min(dense0 + (uint64_t)(tile - tile0) * TILE_N + wave * 32 + (M16 ? r16 : r), dlast);
          ex0 = (a.exclude[d0 >> 5] >> (d0 & 31)) & 1u;
          if constexpr (M16) {
            const uint64_t d1 = min(d0 + 16, dlast);
            ex1 = (a.exclude[d1 >> 5] >> (d1 & 31)) & 1u;
          }
        }
      }
      const bool has_next = (s + 1 < nsteps);

      // ---- MFMA over this 32-float k-step ----
      {
        const float *Qb = Qs + buf * ROWS * TILE_K;
        const float *Bb = Bs + buf * SLAB;
        if constexpr (M16) {
          const int swz = (r16 >> 1) & 7;
#pragma unroll
          for (int kk2 = 0; kk2 < 2; ++kk2) {
            const int c = (kq + 4 * kk2) ^ swz;
            const f32x4 af0 = *reinterpret_cast<const f32x4 *>(Qb + (r16 * 8 + c) * 4);
            f32x4 af1 = af0;
            if (two) af1 = *reinterpret_cast<const f32x4 *>(Qb + ((16 + r16) * 8 + c) * 4);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
              const int brow = wave * 32 + cb * 16 + r16;
              const f32x4 bf = *reinterpret_cast<const f32x4 *>(Bb + (brow * 8 + c) * 4);
              if constexpr (F16) {
                acc16[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af0), __builtin_bit_cast(f16x8, bf), acc16[cb], 0, 0, 0);
                if (two) acc16[2 + cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af1), __builtin_bit_cast(f16x8, bf), acc16[2 + cb], 0, 0, 0);
              } else {
                acc16[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af0.x, bf.x, acc16[cb], 0, 0, 0);
                acc16[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af0.y, bf.y, acc16[cb], 0, 0, 0);
                acc16[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af0.z, bf.z, acc16[cb], 0, 0, 0);
                acc16[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af0.w, bf.w, acc16[cb], 0, 0, 0);
                if (two) {
                  acc16[2 + cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af1.x, bf.x, acc16[2 + cb], 0, 0, 0);
                  acc16[2 + cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af1.y, bf.y, acc16[2 + cb], 0, 0, 0);
                  acc16[2 + cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af1.z, bf.z, acc16[2 + cb], 0, 0, 0);
                  acc16[2 + cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af1.w, bf.w, acc16[2 + cb], 0, 0, 0);
                }
              }
            }
          }
        } else {
          const int brow = wave * 32 + r;
          const int swz = (r >> 1) & 7;
          constexpr int KK_UNROLL = (NG >= 4) ? 2 : 4;   // keep the A-fragment live range short when NG is large
#pragma unroll KK_UNROLL
          for (int kk = 0; kk < 4; ++kk) {
            const int c = (2 * kk + h) ^ swz;
            const f32x4 bf = *reinterpret_cast<const f32x4 *>(Bb + (brow * 8 + c) * 4);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
              const f32x4 af = *reinterpret_cast<const f32x4 *>(Qb + ((g * 32 + r) * 8 + c) * 4);
              if constexpr (F16) {
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af), __builtin_bit_cast(f16x8, bf), acc[g], 0, 0, 0);
              } else {
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc[g], 0, 0, 0);
              }
            }
          }
        }
      }

      // ---- tile epilogue: metric fix-up, then bounded top-k admission ----
      // The MFMA C layout spreads one query row over the lanes of a wave and the 4 waves hold different
      // columns of it, so the scores of one row group x 128 columns are transposed through the staging
      // buffer that is idle during this step (16 KiB) and every row is then admitted by ONE owner wave
      // (row i of the group belongs to wave i % 4): no locks, no atomics.
      if (ks == nks - 1) {
        if (a.dump == nullptr)
          for (int j = tid; j < ROWS; j += 256) {  // refresh the query-wide bounds (visible after the barrier below)
            const float g_ = fkey_inv(__hip_atomic_load(&a.gtau[qrow_s[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            st.gt[j] = g_;
            st.tq[j] = fminf(st.tau[j], g_);       // (tau[j] is only written by the owner wave, in earlier epilogues)
          }
        float *Sc = Bs + (buf ^ 1) * SLAB;                                  // [<=32 rows][128 cols]
        const uint32_t local0 = (tile - tile0) * TILE_N + wave * 32;        // row index inside list/store, lane 0
        const uint32_t pos0 = tile * TILE_N;                                // padded position of column 0
        if constexpr (M16) {
          const bool v0 = (local0 + r16 < rows_valid_total) && (ex0 == 0);
          const bool v1 = (local0 + 16 + r16 < rows_valid_total) && (ex1 == 0);
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            if (g == 1 && !two) break;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
              const float bn = cb ? bn1 : bn0;
              const bool cv = cb ? v1 : v0;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int row_l = g * 16 + kq * 4 + e;
                const float dot = acc16[g * 2 + cb][e];
                float sc;
                if (a.metric == METRIC_L2) sc = fmaxf(fmaf(-2.f, dot, qn_s[row_l] + bn), 0.f);
                else if (a.metric == METRIC_IP) sc = -dot;
                else sc = 1.f - dot;
                Sc[row_l * TILE_N + wave * 32 + cb * 16 + r16] = cv ? sc : __builtin_inff();
                acc16[g * 2 + cb][e] = 0.f;
              }
            }
          }
          __syncthreads();
          {
            // rows dealt round-robin to the 4 waves; the next row's scores and bound are fetched from LDS while
            // the current one is examined (the fast path is otherwise one exposed LDS latency per row)
            f32x2 v = *reinterpret_cast<const f32x2 *>(Sc + wave * TILE_N + 2 * lane);
            float t0 = st.tq[wave];
#pragma unroll 1
            for (int i = 0; i < 8; ++i) {
              const int row = i * 4 + wave;
              const int nrow = min(row + 4, ROWS - 1);
              const f32x2 vn = *reinterpret_cast<const f32x2 *>(Sc + nrow * TILE_N + 2 * lane);
              const float tn = st.tq[nrow];
              if ((uint32_t)row < nrows) {
                if (a.dump) *reinterpret_cast<f32x2 *>(a.dump + (size_t)qrow_s[row] * a.dump_stride + pos0 + 2 * lane) = v;
                else owner_row(st, row, v.x, v.y, t0, pos0, lane);
              }
              v = vn;
              t0 = tn;
            }
          }
          __syncthreads();
        } else {
          const bool colvalid = (local0 + r < rows_valid_total) && (ex0 == 0);
#pragma unroll
          for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int row_l = (e & 3) + 8 * (e >> 2) + 4 * h;
              const float dot = acc[g][e];
              float sc;
              if (a.metric == METRIC_L2) sc = fmaxf(fmaf(-2.f, dot, qn_s[g * 32 + row_l] + bn0), 0.f);
              else if (a.metric == METRIC_IP) sc = -dot;
              else sc = 1.f - dot;
              Sc[row_l * TILE_N + wave * 32 + r] = colvalid ? sc : __builtin_inff();
              acc[g][e] = 0.f;
            }
            __syncthreads();
            {
              f32x2 v = *reinterpret_cast<const f32x2 *>(Sc + wave * TILE_N + 2 * lane);
              float t0 = st.tq[g * 32 + wave];
#pragma unroll 1
              for (int i = 0; i < 8; ++i) {
                const int row_l = i * 4 + wave;                  // rows dealt round-robin to the 4 waves
                const int row = g * 32 + row_l;
                const int nrow_l = min(row_l + 4, 31);
                const f32x2 vn = *reinterpret_cast<const f32x2 *>(Sc + nrow_l * TILE_N + 2 * lane);
                const float tn = st.tq[g * 32 + nrow_l];
                if ((uint32_t)row < nrows) {
                  if (a.dump) *reinterpret_cast<f32x2 *>(a.dump + (size_t)qrow_s[row] * a.dump_stride + pos0 + 2 * lane) = v;
                  else owner_row(st, row, v.x, v.y, t0, pos0, lane);
                }
                v = vn;
                t0 = tn;
              }
            }
            __syncthreads();
          }
        }
      }

      if (has_next) stage_store<QL>(sr[u ^ 1], Bs + (buf ^ 1) * SLAB, Qs + (buf ^ 1) * ROWS * TILE_K, srow, sswz, tid);
      __syncthreads();
      advance(tile, ks);
     }
    }

    // ---- write the partial lists ----
    for (uint32_t j = tid; a.dump == nullptr && j < nrows * k; j += 256) {
      uint32_t row = j / k, t = j - row * k;
      uint32_t c = st.cnt[row];
      size_t o = (size_t)slot_s[row] * k + t;
      a.part_s[o] = (t < c) ? st.Ls[(size_t)row * k + t] : __builtin_inff();
      a.part_i[o] = (t < c) ? st.Li[(size_t)row * k + t] : IDX_NONE;
    }
    __syncthreads();
  }
#ifdef ZVK_CLOCK_STAMP
  if (tid == 0 && blockIdx.x < 1024 && a.mode == 1) {
    zvk_clock_stamps[blockIdx.x][2] = __builtin_amdgcn_s_memtime();
    zvk_clock_stamps[blockIdx.x][3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// The wide flat-scan kernel: 512 threads = 8 waves as 2 (query-row halves) x 4 (column blocks) over a
// 128-query x 128-row tile.  Same LDS images, K-step, epilogue and list logic as scan_kernel<4,..>, but
// the staging of one K-step (16 KiB base slab + 16 KiB query rows) is shared by 8 waves instead of 4 and
// each wave carries only 2 accumulator groups (32 registers), so the kernel fits 128 VGPRs and runs at
// 4 waves per SIMD (two work-groups per CU): twice the resident waves of the 4-wave NG=4 shape at the
// same base-row reuse, which is what the matrix cores need to stay busy across barriers and epilogues.
// Flat mode only (mode 0); the IVF list scan keeps the 16-row shape above.
// ---------------------------------------------------------------------------------------------
constexpr int W8_ROWS = 128;
__host__ __device__ inline size_t scan8_lds_bytes(uint32_t k) {
  return (2 * (size_t)W8_ROWS * TILE_K + 2 * (size_t)SLAB + 7 * (size_t)W8_ROWS + 4 + 2 * (size_t)W8_ROWS * k) * 4;
}

template <bool EXCL, bool F16, bool GATHER>
__global__ void __launch_bounds__(512, 4) scan8_kernel(const ScanArgs a) {
  constexpr int ROWS = W8_ROWS;
  extern __shared__ f32x4 zvk_smem4[];
  float *smem = reinterpret_cast<float *>(zvk_smem4);
  float *Qs = smem;                      // [2][ROWS*32]
  float *Bs = Qs + 2 * ROWS * TILE_K;    // [2][SLAB]
  float *qn_s = Bs + 2 * SLAB;           // [ROWS]
  RowState st;
  st.tau = qn_s + ROWS;
  st.cnt = reinterpret_cast<uint32_t *>(st.tau + ROWS);
  uint32_t *qrow_s = st.cnt + ROWS;
  uint32_t *slot_s = qrow_s + ROWS;
  st.k = a.k;
  st.gt = reinterpret_cast<float *>(slot_s + ROWS + 4);
  st.tq = st.gt + ROWS;
  st.gtau = a.gtau;
  st.qrow = qrow_s;
  st.Ls = st.tq + ROWS;
  st.Li = reinterpret_cast<uint32_t *>(st.Ls + (size_t)ROWS * a.k);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wn = wave & 3, wm = wave >> 2;               // column block, query-row half
  const int r = lane & 31, h = lane >> 5;                 // 32x32x2 operand coordinates
  const int srow = tid >> 3, schunk = tid & 7;            // staging coordinates: rows srow and srow + 64
  const int sswz = schunk ^ ((srow >> 1) & 7);
  const uint32_t dpad = a.dpad, nks = a.nks, k = a.k;
  const uint32_t ntiles_total = (uint32_t)((a.n + TILE_N - 1) / TILE_N);
  const uint32_t rows_valid_total = (uint32_t)min((uint64_t)0xffffffffu, a.n);

  // XCD-aware item order: work-groups whose ids agree modulo 8 share an XCD (and its L2), so the query tiles that
  // stream the SAME chunk of the base are given ids of one residue class: the chunk is then fetched from HBM once
  // and the other query tiles read it from that XCD's L2.  (A speed choice only; any placement is correct.)
  const uint32_t vtotal = ((a.nchunks + 7) / 8) * 8 * a.nqtiles;
  for (uint32_t v = blockIdx.x; v < vtotal; v += gridDim.x) {   // uniform exit
    const uint32_t qtile = (v >> 3) % a.nqtiles;
    const uint32_t chunk = ((v >> 3) / a.nqtiles) * 8 + (v & 7);
    if (chunk >= a.nchunks) continue;
    const uint32_t tile_begin = chunk * a.tiles_per_chunk;
    const uint32_t tile_end = min(tile_begin + a.tiles_per_chunk, ntiles_total);
    const uint32_t r0 = qtile * ROWS;
    const uint32_t nrows = min((uint32_t)ROWS, a.nq - r0);

    for (int j = tid; j < ROWS; j += 512) {
      const bool live = (uint32_t)j < nrows;
      const uint32_t qrow = live ? r0 + j : r0;
      qrow_s[j] = qrow;
      slot_s[j] = live ? qrow * a.nchunks + chunk : IDX_NONE;
      qn_s[j] = (a.metric == METRIC_L2) ? a.qnorm[qrow] : 0.f;
      st.tau[j] = a.threshold;
      st.gt[j] = a.threshold;
      st.tq[j] = a.threshold;
      st.cnt[j] = 0;
    }
    __syncthreads();

    const uint32_t nsteps = (tile_end - tile_begin) * nks;

    // Staging by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write pass).  One wave-instruction writes
    // 1 KiB of LDS linearly in lane order, which is exactly a piece of the base slab (stored in HBM as its LDS
    // image); the query image's XOR swizzle is applied on the SOURCE side instead: the lane whose LDS slot is
    // chunk p of row `srow` fetches chunk p ^ ((srow >> 1) & 7) of that query row.
    const uint32_t gq0 = qrow_s[srow] * dpad + (uint32_t)sswz * 4u;
    const uint32_t gq1 = qrow_s[srow + 64] * dpad + (uint32_t)sswz * 4u;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    // GATHER: this lane's two slab pieces are chunk p of tile rows srow and srow+64; the rows come from stored
    // positions gp0 / gp1, whose own in-tile row decides the swizzle they were stored with.  The positions of the
    // tile being fetched are held in registers and the next tile's are loaded one tile ahead.
    uint32_t gp0 = 0, gp1 = 0, gpn0 = 0, gpn1 = 0, gp_tile = ~0u;
    auto gather_src = [&](uint32_t gp, uint32_t k_) {
      const uint32_t rs = gp & (TILE_N - 1);
      const uint32_t chunk = (uint32_t)schunk ^ (uint32_t)((srow >> 1) & 7) ^ ((rs >> 1) & 7u);
      return reinterpret_cast<const f32x4 *>(a.base + (size_t)(gp >> 7) * TILE_N * dpad + (size_t)k_ * SLAB) + (rs * 8 + chunk);
    };
    auto stage_glds = [&](uint32_t t_, uint32_t k_, float *Bb, float *Qb) {
      char *bl = reinterpret_cast<char *>(Bb) + wave * 1024;      // wave-uniform destinations
      char *ql = reinterpret_cast<char *>(Qb) + wave * 1024;
      if constexpr (GATHER) {
        if (t_ != gp_tile) {                       // uniform: first step of a new tile
          if (gp_tile == ~0u) {
            gp0 = a.gather_pos[(size_t)t_ * TILE_N + srow];
            gp1 = a.gather_pos[(size_t)t_ * TILE_N + srow + 64];
          } else {
            gp0 = gpn0;
            gp1 = gpn1;
          }
          gp_tile = t_;
          const uint32_t tn = min(t_ + 1, tile_end - 1);
          gpn0 = a.gather_pos[(size_t)tn * TILE_N + srow];
          gpn1 = a.gather_pos[(size_t)tn * TILE_N + srow + 64];
        }
        __builtin_amdgcn_global_load_lds((glb_void *)gather_src(gp0, k_), (lds_void *)bl, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void *)gather_src(gp1, k_), (lds_void *)(bl + 8192), 16, 0, 0);
      } else {
        const f32x4 *bsrc = reinterpret_cast<const f32x4 *>(a.base + (size_t)t_ * TILE_N * dpad + (size_t)k_ * SLAB) + tid;
        __builtin_amdgcn_global_load_lds((glb_void *)bsrc, (lds_void *)bl, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void *)(bsrc + 512), (lds_void *)(bl + 8192), 16, 0, 0);
      }
      __builtin_amdgcn_global_load_lds((glb_void *)(a.queries + (size_t)(gq0 + k_ * TILE_K)), (lds_void *)ql, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void *)(a.queries + (size_t)(gq1 + k_ * TILE_K)), (lds_void *)(ql + 8192), 16, 0, 0);
    };
    floatx16 acc[2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;

    uint32_t tile = tile_begin, ks = 0;          // step being computed
    uint32_t ptile = tile_begin, pks = 0;        // next step to fetch
    uint32_t fetched = 0;
    auto advance = [&](uint32_t &t_, uint32_t &k_) { if (++k_ == nks) { k_ = 0; ++t_; } };
    if (nsteps > 0) {
      stage_glds(ptile, pks, Bs, Qs);
      if (nsteps > 1) advance(ptile, pks);
      fetched = 1;
    }
    // metric fix-up as one fused multiply-add + clamp: L2 -2*dot + (|q|^2 + |b|^2) clamped at 0; IP -dot; cosine 1 - dot
    const float m_alpha = (a.metric == METRIC_L2) ? -2.f : -1.f;
    const float m_beta = (a.metric == METRIC_COSINE) ? 1.f : 0.f;
    const float m_nrm = (a.metric == METRIC_L2) ? 1.f : 0.f;
    const float m_lo = (a.metric == METRIC_L2) ? 0.f : -__builtin_inff();

    float bn0 = 0.f;
    uint32_t ex0 = 0;
    for (uint32_t s0 = 0; s0 < nsteps; s0 += 2) {
#pragma unroll
     for (int u = 0; u < 2; ++u) {
      const uint32_t s = s0 + u;
      if (s >= nsteps) break;                      // uniform
      const int buf = u;
      const bool has_next = (s + 1 < nsteps);
      // Step s: the barrier (preceded by each wave's vmcnt(0), which retires its own DMA pieces) publishes buffer
      // `buf`, filled during step s-1, and retires every read of the other buffer, which is then refilled with step
      // s+1 under this step's matrix work.
      __syncthreads();
      if (has_next) stage_glds(ptile, pks, Bs + (buf ^ 1) * SLAB, Qs + (buf ^ 1) * ROWS * TILE_K);
      if (fetched + 1 < nsteps) advance(ptile, pks);
      ++fetched;
      if constexpr (GATHER) {
        // column norm through the position list: two dependent loads, issued at the start of the tile's last step
        // so that they land under its matrix work
        if (ks == nks - 1) bn0 = a.bnorm[a.gather_pos[(size_t)tile * TILE_N + wn * 32 + r]];
      } else {
        const uint32_t pos0 = tile * TILE_N + wn * 32;
        bn0 = a.bnorm[(size_t)pos0 + r];
        if constexpr (EXCL) {
          const uint64_t d0 = min((uint64_t)pos0 + r, a.ndense - 1);
          ex0 = (a.exclude[d0 >> 5] >> (d0 & 31)) & 1u;
        }
      }
      {
        const float *Qb = Qs + buf * ROWS * TILE_K + wm * 64 * TILE_K;
        const float *Bb = Bs + buf * SLAB;
        const int brow = wn * 32 + r;
        const int swz = (r >> 1) & 7;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int c = (2 * kk + h) ^ swz;
          const f32x4 bf = *reinterpret_cast<const f32x4 *>(Bb + (brow * 8 + c) * 4);
          const f32x4 af0 = *reinterpret_cast<const f32x4 *>(Qb + (r * 8 + c) * 4);
          const f32x4 af1 = *reinterpret_cast<const f32x4 *>(Qb + ((32 + r) * 8 + c) * 4);
          if constexpr (F16) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af0), __builtin_bit_cast(f16x8, bf), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af1), __builtin_bit_cast(f16x8, bf), acc[1], 0, 0, 0);
          } else {
            // (back-to-back accumulation into one group measured 3 % faster than alternating the two groups)
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af0.x, bf.x, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af0.y, bf.y, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af0.z, bf.z, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af0.w, bf.w, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af1.x, bf.x, acc[1], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af1.y, bf.y, acc[1], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af1.z, bf.z, acc[1], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af1.w, bf.w, acc[1], 0, 0, 0);
          }
        }
      }

      // ---- tile epilogue: two rounds; in round gi the row half wm transposes its group 2*wm+gi through 16 KiB of
      // the operand buffer this step has just finished with (half 0: its base slab, half 1: its query rows) and the
      // 4 waves of the half admit its 32 rows (row i of the group belongs to wave i % 4 of the half)
#ifdef ZVK_SCAN8_NOEPI      // (diagnostic builds: the matrix loop of the wide kernel alone — results are then wrong)
      if (ks == nks - 1 && a.n == 0xffffffffffffffffull) {
#else
      if (ks == nks - 1) {
#endif
        if (a.dump == nullptr && tid < ROWS) {
          const float g_ = fkey_inv(__hip_atomic_load(&a.gtau[qrow_s[tid]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          st.gt[tid] = g_;
          st.tq[tid] = fminf(st.tau[tid], g_);
        }
        __syncthreads();                                                          // every wave is done reading `buf`
        float *Sc = wm ? (Qs + buf * ROWS * TILE_K) : (Bs + buf * SLAB);          // [32 rows][128 cols]
        const uint32_t pos0 = tile * TILE_N;
        const bool colvalid = (pos0 + wn * 32 + r < rows_valid_total) && (ex0 == 0);
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
          const int gbase = (wm * 2 + gi) * 32;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row_l = (e & 3) + 8 * (e >> 2) + 4 * h;
            const float dot = acc[gi][e];
            const float sc = fmaxf(fmaf(m_alpha, dot, fmaf(m_nrm, qn_s[gbase + row_l] + bn0, m_beta)), m_lo);
            Sc[row_l * TILE_N + wn * 32 + r] = colvalid ? sc : __builtin_inff();
            acc[gi][e] = 0.f;
          }
          __syncthreads();
          {
            f32x2 v = *reinterpret_cast<const f32x2 *>(Sc + wn * TILE_N + 2 * lane);
            float t0 = st.tq[gbase + wn];
#pragma unroll 1
            for (int i = 0; i < 8; ++i) {
              const int row_l = i * 4 + wn;
              const int row = gbase + row_l;
              const int nrow_l = min(row_l + 4, 31);
              const f32x2 vn = *reinterpret_cast<const f32x2 *>(Sc + nrow_l * TILE_N + 2 * lane);
              const float tn = st.tq[gbase + nrow_l];
              if ((uint32_t)row < nrows) {
                if (a.dump) *reinterpret_cast<f32x2 *>(a.dump + (size_t)qrow_s[row] * a.dump_stride + pos0 + 2 * lane) = v;
                else owner_row(st, row, v.x, v.y, t0, pos0, lane);
              }
              v = vn;
              t0 = tn;
            }
          }
          __syncthreads();
        }
      }

      advance(tile, ks);
     }
    }

    for (uint32_t j = tid; a.dump == nullptr && j < nrows * k; j += 512) {
      uint32_t row = j / k, t = j - row * k;
      uint32_t c = st.cnt[row];
      size_t o = (size_t)slot_s[row] * k + t;
      a.part_s[o] = (t < c) ? st.Ls[(size_t)row * k + t] : __builtin_inff();
      uint32_t pi = (t < c) ? st.Li[(size_t)row * k + t] : IDX_NONE;
      if constexpr (GATHER) { if (pi != IDX_NONE) pi = a.gather_pos[pi]; }      // logical row -> stored position
      a.part_i[o] = pi;
    }
    __syncthreads();
  }
#ifdef ZVK_CLOCK_STAMP
  if (tid == 0 && blockIdx.x < 1024 && a.dump == nullptr) {
    zvk_clock_stamps[blockIdx.x][2] = __builtin_amdgcn_s_memtime();
    zvk_clock_stamps[blockIdx.x][3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

}  // namespace zvk
