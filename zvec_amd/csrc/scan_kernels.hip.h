// scan_kernels.hip.h — gfx950 (CDNA4, wave64) device code of the zvec flat / IVF-Flat scan core.
//
// One kernel does the whole hot path of SURVEY §8(a) rows 1-4, 6-11: the batched query x base
// distance matrix on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains, so the
// scores are fp32-faithful like the reference's AVX kernels, euclidean_distance_matrix_fp32.cc:229-320,
// inner_product_matrix_fp32.cc:509-556) fused with the per-query bounded top-k
// (ailego::Heap semantics, heap.h:103-114: keep the k smallest, first-seen wins ties).
//
// HBM layout ("blocked"): base rows are stored in tiles of 128 rows; inside a tile the K dimension
// is cut in steps of 32 floats, and each (tile, k-step) slab is 128 rows x 128 B = 16 KiB contiguous,
// already XOR-swizzled the way the LDS image wants it.  A work-group therefore streams a list as
// a sequence of contiguous 16 KiB slabs (perfectly coalesced, DRAM-page friendly) and the slab is
// copied to LDS verbatim.
//
//   float offset of (pos, kcol), row stride dpad (multiple of 32):
//     tile = pos / 128, r = pos % 128, ks = kcol / 32, c = (kcol % 32) / 4, e = kcol % 4
//     off  = tile*128*dpad + ks*4096 + (r*8 + (c ^ ((r >> 1) & 7)))*4 + e
//
// The swizzle makes the ds_read_b128 operand fetch conflict free: lanes 0..31 of a wave read 32
// different rows at the same 16-byte chunk; (r&1)*8 + (c ^ ((r>>1)&7)) is a bijection onto the 16
// 16-byte slots of a 256 B bank row for each of the instruction's 16-lane groups.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zvk {

constexpr int TILE_N = 128;   // base rows per tile (4 waves x 32 MFMA columns)
constexpr int TILE_K = 32;    // floats per k-step (one 128 B line per row)
constexpr int QGROUP = 32;    // query rows per MFMA row block
constexpr int SLAB = TILE_N * TILE_K;  // floats per (tile, k-step) slab
constexpr uint32_t IDX_NONE = 0xffffffffu;

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

enum { METRIC_L2 = 0, METRIC_IP = 1, METRIC_COSINE = 2 };

__host__ __device__ inline size_t blocked_offset(uint64_t pos, uint32_t kcol, uint32_t dpad) {
  uint64_t tile = pos >> 7;
  uint32_t r = (uint32_t)(pos & 127);
  uint32_t ks = kcol >> 5, c = (kcol & 31) >> 2, e = kcol & 3;
  return (size_t)tile * TILE_N * dpad + (size_t)ks * SLAB + (size_t)((r * 8 + (c ^ ((r >> 1) & 7))) * 4 + e);
}

// ---------------------------------------------------------------------------------------------
// scan kernel arguments
// ---------------------------------------------------------------------------------------------
struct ScanArgs {
  const float *base;        // blocked rows
  const float *bnorm;       // [padded positions] squared norms (L2 only)
  const uint32_t *exclude;  // nullable bitset over DENSE positions (32-bit words), set = skip
  const float *queries;     // [nq][dpad] row-major, zero padded
  const float *qnorm;       // [nq] squared norms (L2 only)
  uint32_t dpad;
  uint32_t nks;             // dpad / 32
  int metric;
  uint32_t k;
  float threshold;
  int mode;                 // 0 flat, 1 ivf
  // flat decomposition: item = chunk * nqtiles + qtile
  uint32_t nq;
  uint64_t n;               // rows in the flat store
  uint32_t tiles_per_chunk;
  const uint32_t *list_tpc;       // IVF: tiles per chunk of each list (shorter chunks for the lists dealt last)
  // wide flat kernel, GATHER variant: logical row i of the scan is stored position gather_pos[i] (ascending kept
  // positions of a sparse filter, padded to whole tiles with any valid position); n counts logical rows
  const uint32_t *gather_pos;
  uint32_t nchunks;
  uint32_t nqtiles;
  // ivf decomposition (built on device by the plan kernels)
  const uint32_t *total_items;  // [1]
  uint32_t *queue;              // [1] work-queue head (zeroed per search): items are dealt dynamically
  const uint32_t *list_order;   // [nlist] lists sorted by stored size, largest first (LPT dealing)
  const uint32_t *item_off;     // [nlist+1] exclusive prefix of work items over list_order positions
  const uint32_t *list_tile0;   // [nlist] first tile of the list in the blocked store
  const uint32_t *list_size;    // [nlist] rows stored in the list (this shard)
  const uint64_t *list_dense0;  // [nlist] dense (unpadded) position of the list's first row
  const uint32_t *list_qoff;    // [nlist+1] CSR offsets: queries probing the list
  const uint32_t *csr_q;        // query row
  const uint32_t *csr_slot;     // output slot of (query, probe rank), chunk 0
  uint32_t nlist;
  uint64_t ndense;              // number of dense positions (bits of `exclude`)
  uint32_t *gtau;               // [nq] query-wide admission bounds (keys, see fkey), initialised to fkey(threshold)
  // dense-score mode (small cache-resident bases, e.g. the IVF coarse step): instead of admitting
  // into top-k lists the kernel writes every score to dump[query][padded position]; selection is then
  // done by merge_kernel over whole rows (one wave per query)
  float *dump;                  // nullable
  uint32_t dump_stride;         // floats per query row (= tiles * 128)
  // outputs: per-(slot) partial lists
  float *part_s;                // [slots][k]
  uint32_t *part_i;             // [slots][k] padded position, IDX_NONE = empty
};

// LDS footprint in bytes for a given NG / k (host mirrors this)
__host__ __device__ inline size_t scan_lds_bytes(int ng, uint32_t k, bool m16 = false) {
  size_t rows = m16 ? 32 : (size_t)ng * QGROUP;
  return (2 * rows * TILE_K + 2 * (size_t)SLAB + 7 * rows + 4 + 2 * rows * k) * 4;
}

// broadcast of lane `l` (wave-uniform index) without touching the LDS crossbar: v_readlane_b32
__device__ __forceinline__ float bcast_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ uint32_t bcast_u(uint32_t v, int l) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}

// order-preserving float <-> uint32 map (so that atomicMin on the key is a float min, negative IP scores included)
__device__ __forceinline__ uint32_t fkey(float f) {
  const uint32_t b = __builtin_bit_cast(uint32_t, f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
  const uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __builtin_bit_cast(float, b);
}

struct RowState {
  float *tau;        // admission bound per row: threshold until the list is full, then its k-th score
  uint32_t *cnt;
  float *Ls;         // [rows][k] scores, ascending (score, position)
  uint32_t *Li;      // [rows][k] positions
  uint32_t k;
  float *gt;         // [rows] query-wide bound fetched from gtau at the start of the tile epilogue
  float *tq;         // [rows] min(tau, gt): the one value the fast path reads
  uint32_t *gtau;    // global [nq] keys: min over all work-groups of a FULL local list's k-th score
  const uint32_t *qrow;  // [rows] global query row of each local row
};

// Whole-wave insertion of candidate (s, o, i) into a bounded list kept SORTED ascending by
// (score, order, index) in LDS: count the entries that precede it (one ballot per 64 entries), shift
// the tail up by one, drop it in.  No reduction, no atomics.  Returns false when the candidate does
// not make the list.  The kept set is the k smallest under (score, scan order): exactly what the
// reference's sequential `if (score < heap.top) replace` (heap.h:103-114) keeps whenever no two
// scores tie at the k-th place.  `c` (entries in the list) and `tau` (admission bound) are wave-uniform
// values the caller keeps in registers.
template <bool HAS_ORD>
__device__ __forceinline__ bool sorted_insert(float *L, uint32_t *O, uint32_t *I, uint32_t k, uint32_t &c, float s,
                                              uint32_t o, uint32_t i, int lane, float &tau) {
  if (k <= 64) {
    // fast path: one entry per lane
    float es = 0.f;
    uint32_t eo = 0, ei = 0;
    bool less = false;
    const uint32_t j = (uint32_t)lane;
    if (j < c) {
      es = L[j];
      ei = I[j];
      if (HAS_ORD) eo = O[j];
      less = es < s || (es == s && (eo < o || (eo == o && ei < i)));
    }
    const uint32_t p = (uint32_t)__popcll(__ballot(less));
    if (p >= k) return false;
    const uint32_t hi = min(c, k - 1);                 // entries [p, hi) move up by one
    if (j >= p && j < hi) {
      L[j + 1] = es;
      I[j + 1] = ei;
      if (HAS_ORD) O[j + 1] = eo;
    }
    if (lane == 0) {
      L[p] = s;
      I[p] = i;
      if (HAS_ORD) O[p] = o;
    }
    c = min(c + 1, k);
    if (c == k) tau = (p == k - 1) ? s : bcast_f(es, (int)k - 2);   // new k-th = candidate or the old (k-1)-th
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    return true;
  }
  uint32_t p = 0;
  for (uint32_t j0 = 0; j0 < c; j0 += 64) {
    const uint32_t j = j0 + lane;
    bool less = false;
    if (j < c) {
      const float es = L[j];
      const uint32_t ei = I[j];
      const uint32_t eo = HAS_ORD ? O[j] : 0u;
      less = es < s || (es == s && (eo < o || (eo == o && ei < i)));
    }
    p += (uint32_t)__popcll(__ballot(less));
  }
  if (p >= k) return false;
  const uint32_t hi = min(c, k - 1);
  if (hi > p) {
    for (int m = (int)((hi - 1) >> 6); m >= (int)(p >> 6); --m) {   // top chunk first: never overwrites unread data
      const uint32_t j = (uint32_t)m * 64u + lane;
      const bool mv = j >= p && j < hi;
      float es = 0.f;
      uint32_t eo = 0, ei = 0;
      if (mv) { es = L[j]; ei = I[j]; if (HAS_ORD) eo = O[j]; }
      __builtin_amdgcn_wave_barrier();
      if (mv) { L[j + 1] = es; I[j + 1] = ei; if (HAS_ORD) O[j + 1] = eo; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (lane == 0) {
    L[p] = s;
    I[p] = i;
    if (HAS_ORD) O[p] = o;
  }
  c = min(c + 1, k);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  if (c == k) tau = L[k - 1];
  return true;
}

// Owner-wave admission of one row of the score tile: lane holds the scores of columns 2*lane and
// 2*lane+1 (s0, s1); pos0 = padded position of column 0 of the tile.
__device__ __forceinline__ void owner_row(const RowState &st, int row, float s0, float s1, float t0, uint32_t pos0, int lane) {
  // admission bound t0 = min(this list's k-th score, the query-wide bound shared by every work-group that
  // scans for the same query), pre-read by the caller together with the scores.  A score above the shared
  // bound cannot be in the final top-k: some work-group already holds k candidates at or below it.  Ties (==)
  // are kept; the merge orders them.
  uint64_t m0 = __ballot(s0 <= t0);
  uint64_t m1 = __ballot(s1 <= t0);
  if ((m0 | m1) == 0) return;
  float tl = st.tau[row];
  const float tg = st.gt[row];
  float t = fminf(tl, tg);
  const uint32_t k = st.k;
  uint32_t c = st.cnt[row];
  float *L = st.Ls + (size_t)row * k;
  uint32_t *I = st.Li + (size_t)row * k;
  bool improved = false;
  while ((m0 | m1) != 0) {
    int l;
    float cs;
    uint32_t ci;
    if (m0 != 0) {
      l = __builtin_ctzll(m0);
      cs = bcast_f(s0, l);
      ci = pos0 + 2u * (uint32_t)l;
      m0 &= m0 - 1;
    } else {
      l = __builtin_ctzll(m1);
      cs = bcast_f(s1, l);
      ci = pos0 + 2u * (uint32_t)l + 1u;
      m1 &= m1 - 1;
    }
    if (sorted_insert<false>(L, nullptr, I, k, c, cs, 0u, ci, lane, tl)) {
      improved = true;
      t = fminf(tl, tg);
      m0 &= __ballot(s0 <= t);
      m1 &= __ballot(s1 <= t);
    }
  }
  if (lane == 0) {
    st.cnt[row] = c;
    st.tau[row] = tl;
    st.tq[row] = fminf(tl, tg);
    if (improved && c == k && tl < tg) atomicMin(&st.gtau[st.qrow[row]], fkey(tl));
  }
}

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
#ifndef ZVK_M16_NOEPI
                else owner_row(st, row, v.x, v.y, t0, pos0, lane);
#endif
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
      if (ks == nks - 1) {
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
}

// ---------------------------------------------------------------------------------------------
// merge kernel: one wave per query; merges the query's slots (scan order = slot order, then
// position) into the final sorted top-k.  Also used for the shard merge after the all-gather.
// ---------------------------------------------------------------------------------------------
struct MergeArgs {
  const float *part_s;
  const uint32_t *part_i;        // positions (nullptr when part_keys is used)
  const uint64_t *part_keys;     // alternative candidate keys (shard merge); nullptr otherwise
  const uint32_t *slot_begin;    // [nq+1] or nullptr => q*slots_per_q
  uint32_t slots_per_q;
  // candidate e of slot j lives at ((slot_base + j*slot_stride) * k + e)
  uint32_t slot_stride;          // 1 for scan partials; nq for [part][q][k] shard layout
  const uint32_t *part_counts;   // optional [slots] valid entries per slot (shard merge)
  uint32_t k;
  uint32_t slot_len;             // candidates per slot (k for partial lists; the row length for dense scores)
  // packed shard exchange: part j's arrays start packed_stride BYTES after part j-1's (one all-gather buffer:
  // per rank [count*k keys u64][count*k scores f32][count counts u32], padded to 16 B); 0 = separate arrays
  uint64_t packed_stride;
  float threshold;
  const uint32_t *bound_keys;    // optional [nq]: order-preserving key of an upper bound of each query's final k-th score
  const uint64_t *keymap;        // position -> key (nullable => key = position)
  uint64_t *out_keys;            // [nq][k]
  float *out_scores;             // [nq][k]
  uint32_t *out_idx;             // optional [nq][k] positions
  uint32_t *out_counts;          // [nq]
};

// Launched with 64 threads (one wave per query) or, for small batches of partial-list merges, 256: the extra waves
// only help gathering the survivors (the one phase that streams every candidate); wave 0 finishes alone.
__global__ void __launch_bounds__(256) merge_kernel(const MergeArgs a) {
  extern __shared__ f32x4 zvk_smem4[];
  const uint32_t k = a.k;
  float *Ls = reinterpret_cast<float *>(zvk_smem4);          // [k]
  uint32_t *Lo = reinterpret_cast<uint32_t *>(Ls + k);       // [k] order (slot)
  uint32_t *Li = Lo + k;                                      // [k] idx / candidate ordinal
  const int lane = threadIdx.x & 63;
  const uint32_t wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const uint32_t q = blockIdx.x;
  uint32_t sb, nslots;
  if (a.slot_begin) { sb = a.slot_begin[q]; nslots = a.slot_begin[q + 1] - sb; }
  else if (a.slot_stride == 1) { sb = q * a.slots_per_q; nslots = a.slots_per_q; }
  else { sb = q; nslots = a.slots_per_q; }

  uint32_t cnt = 0;            // uniform
  float tau = a.threshold;     // uniform admission bound: threshold until the list is full, then its k-th score
  const uint32_t sl = a.slot_len;
  const uint64_t total = (uint64_t)nslots * sl;
  constexpr int U = 16;          // candidate batches fetched together: one wave per query is latency-bound on this stream

  // Dense rows (coarse step): a cheap, exact upper bound of the k-th score before any insertion — every
  // lane takes the minimum of its own strided elements; those are 64 distinct candidates, so the k-th
  // smallest of them is >= the k-th smallest of the whole row.  Cuts the insertions to the few elements
  // at or below that bound.
  if (a.part_i == nullptr && a.part_keys == nullptr && a.part_counts == nullptr && k <= 64 && total >= 64 && nwaves == 1) {
    float mn = __builtin_inff();
    for (uint64_t base = 0; base < total; base += 64 * U) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t e = base + (uint64_t)u * 64 + lane;
        v[u] = (e < total) ? a.part_s[(size_t)sb * sl + e] : __builtin_inff();
      }
#pragma unroll
      for (int u = 0; u < U; ++u) mn = fminf(mn, v[u]);
    }
    uint32_t rank = 0;
    for (int m = 0; m < 64; ++m) {
      const float o = bcast_f(mn, m);
      rank += (o < mn || (o == mn && m < lane)) ? 1u : 0u;
    }
    const uint64_t hit = __ballot(rank == k - 1);
    const float bound = bcast_f(mn, __builtin_ctzll(hit));
    tau = fminf(tau, bound);
  }
  else if (a.bound_keys != nullptr) {
    // fused scans: the shared admission bound of the query (min over work-groups of a full list's k-th score) is an
    // upper bound of the final k-th score; nothing above it can be in the result
    tau = fminf(tau, fkey_inv(a.bound_keys[q]));
  }

  // Survivors: usually only a few dozen candidates are at or below the bound.  Gather them (ballot compaction, no
  // ordering yet), sort the <= 128 survivors once by (score, slot, index) with a bitonic network in LDS and emit the
  // first k — instead of one dependent sorted insertion per survivor.  More survivors than that (heavy ties, no
  // bound yet, fewer than k admissible candidates in a long row): the general insertion path below.
  constexpr uint32_t SURV = 128;                 // sorted at once
  constexpr uint32_t GATHER = 512;               // gathered at most; between the two, one k-select trims them first
  __shared__ unsigned long long surv_hi[GATHER]; // order-preserving score key << 32 | slot
  __shared__ uint32_t surv_lo[GATHER];           // index / candidate ordinal
  __shared__ uint32_t sh_ns;
  if (total <= 0xffffffffull && k <= SURV) {
    const uint32_t tot = (uint32_t)total;
    if (wave == 0) {
      surv_hi[lane] = ~0ull; surv_hi[lane + 64] = ~0ull;
      surv_lo[lane] = IDX_NONE; surv_lo[lane + 64] = IDX_NONE;
      if (lane == 0) sh_ns = 0;
    }
    __syncthreads();
    uint32_t ns = 0;       // uniform per wave: survivors seen so far (single wave) / at the last append (several waves)
    const bool dense_row = a.part_i == nullptr && a.part_keys == nullptr && a.part_counts == nullptr && nslots == 1;
    auto gather = [&](auto dense_tag) {
      constexpr bool DENSE = decltype(dense_tag)::value;
      for (uint32_t base = wave * 64 * U; base < tot && ns <= GATHER; base += nwaves * 64 * U) {
        float sv[U];
        uint32_t iv[U], jv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t e = base + (uint32_t)u * 64 + lane;
          bool valid = e < tot;
          if constexpr (DENSE) {                  // one row of scores: element e is candidate e
            sv[u] = valid ? a.part_s[(size_t)sb * sl + e] : __builtin_inff();
            iv[u] = e;
            jv[u] = 0;
          } else {
            const uint32_t j = valid ? e / sl : 0, t = valid ? e - j * sl : 0;
            const size_t o = a.packed_stride ? ((size_t)q * sl + t) : (((size_t)sb + (size_t)j * a.slot_stride) * sl + t);
            const size_t pbytes = (size_t)j * a.packed_stride;
            float sc = __builtin_inff();
            uint32_t idx = IDX_NONE;
            if (valid && a.part_counts)
              valid = t < (a.packed_stride ? *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(a.part_counts + q) + pbytes)
                                           : a.part_counts[sb + (size_t)j * a.slot_stride]);
            if (valid) {
              sc = a.packed_stride ? *reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.part_s + o) + pbytes) : a.part_s[o];
              idx = a.part_i ? a.part_i[o] : t;
              if (a.part_i && idx == IDX_NONE) valid = false;
            }
            sv[u] = valid ? sc : __builtin_inff();      // (+inf never passes: tau <= FLT_MAX)
            iv[u] = idx;
            jv[u] = j;
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool in = sv[u] <= tau;
          const uint64_t m = __ballot(in);
          if (m) {
            const uint32_t add = (uint32_t)__popcll(m);
            uint32_t first = ns;
            if (nwaves > 1) {                      // the waves append through one LDS counter
              uint32_t o = 0;
              if (lane == 0) o = atomicAdd(&sh_ns, add);
              first = bcast_u(o, 0);
            }
            const uint32_t pos = first + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (in && pos < GATHER) {
              surv_hi[pos] = ((unsigned long long)fkey(sv[u] + 0.f) << 32) | jv[u];
              surv_lo[pos] = iv[u];
            }
            ns = first + add;
          }
        }
      }
    };
    if (dense_row) gather(std::true_type{}); else gather(std::false_type{});
    if (nwaves > 1) {
      __syncthreads();
      if (wave != 0) return;                       // (no work-group barrier below this point)
      ns = sh_ns;
    }
    if (ns > SURV && ns <= GATHER) {
      // too many for one sort: find the k-th smallest score key among the survivors (bisection on the 32-bit key,
      // counts by ballot) and keep only the candidates at or below it (k plus ties)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      constexpr int PER = GATHER / 64;
      unsigned long long rh[PER];
      uint32_t rl[PER];
#pragma unroll
      for (int e = 0; e < PER; ++e) {
        const uint32_t i = (uint32_t)e * 64 + lane;
        rh[e] = (i < ns) ? surv_hi[i] : ~0ull;
        rl[e] = (i < ns) ? surv_lo[i] : IDX_NONE;
      }
      uint32_t lo = 0, hi = 0xffffffffu;         // smallest key T with count(key <= T) >= k
      while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        uint32_t c = 0;
#pragma unroll
        for (int e = 0; e < PER; ++e) c += (uint32_t)__popcll(__ballot((uint32_t)(rh[e] >> 32) <= mid && rh[e] != ~0ull));
        if (c >= k) hi = mid; else lo = mid + 1;
      }
      __builtin_amdgcn_wave_barrier();
      uint32_t n2 = 0;
#pragma unroll
      for (int e = 0; e < PER; ++e) {
        const bool keep = rh[e] != ~0ull && (uint32_t)(rh[e] >> 32) <= lo;
        const uint64_t m = __ballot(keep);
        const uint32_t pos = n2 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (keep && pos < SURV) { surv_hi[pos] = rh[e]; surv_lo[pos] = rl[e]; }
        n2 += (uint32_t)__popcll(m);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      if (n2 <= SURV) {
        for (uint32_t i = n2 + lane; i < SURV; i += 64) { surv_hi[i] = ~0ull; surv_lo[i] = IDX_NONE; }
        ns = n2;
      } else {
        ns = GATHER + 1;    // (more than 128 candidates tie at the k-th score) -> general path
      }
    }
    if (ns <= SURV) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      for (uint32_t size = 2; size <= SURV; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
          // 64 compare-exchanges per step: lane -> the lower index of its pair
          const uint32_t i = ((uint32_t)lane / stride) * (stride * 2) + ((uint32_t)lane % stride);
          const uint32_t j = i + stride;
          const bool up = ((i & size) == 0);
          const unsigned long long xh = surv_hi[i], yh = surv_hi[j];
          const uint32_t xl = surv_lo[i], yl = surv_lo[j];
          const bool gt = xh > yh || (xh == yh && xl > yl);
          if (gt == up) { surv_hi[i] = yh; surv_hi[j] = xh; surv_lo[i] = yl; surv_lo[j] = xl; }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_wave_barrier();
        }
      }
      const uint32_t c = min(ns, k);
      for (uint32_t j = lane; j < k; j += 64) {
        const size_t o = (size_t)q * k + j;
        if (j < c) {
          const unsigned long long w = surv_hi[j];
          const uint32_t vo = (uint32_t)w, vi = surv_lo[j];
          uint64_t key;
          if (a.part_keys)
            key = a.packed_stride ? *reinterpret_cast<const uint64_t *>(reinterpret_cast<const char *>(a.part_keys + (size_t)q * sl + vi) + (size_t)vo * a.packed_stride)
                                  : a.part_keys[((size_t)sb + (size_t)vo * a.slot_stride) * sl + vi];
          else key = a.keymap ? a.keymap[vi] : (uint64_t)vi;
          a.out_keys[o] = key;
          a.out_scores[o] = fkey_inv((uint32_t)(w >> 32));
          if (a.out_idx) a.out_idx[o] = vi;
        } else {
          a.out_keys[o] = ~0ull;
          a.out_scores[o] = __builtin_inff();
          if (a.out_idx) a.out_idx[o] = IDX_NONE;
        }
      }
      if (lane == 0) a.out_counts[q] = c;
      return;
    }
  }

  if (wave != 0) return;       // the general path is one wave's work
  for (uint64_t base = 0; base < total; base += 64 * U) {
    float sv[U];
    uint32_t iv[U], jv[U];
    bool vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t e = base + (uint64_t)u * 64 + lane;
      bool valid = e < total;
      const uint32_t j = valid ? (uint32_t)(e / sl) : 0, t = valid ? (uint32_t)(e - (uint64_t)j * sl) : 0;
      const size_t o = a.packed_stride ? ((size_t)q * sl + t) : (((size_t)sb + (size_t)j * a.slot_stride) * sl + t);
      const size_t pbytes = (size_t)j * a.packed_stride;
      float s = __builtin_inff();
      uint32_t idx = IDX_NONE;
      if (valid && a.part_counts)
        valid = t < (a.packed_stride ? *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(a.part_counts + q) + pbytes)
                                     : a.part_counts[sb + (size_t)j * a.slot_stride]);
      if (valid) {
        s = a.packed_stride ? *reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.part_s + o) + pbytes) : a.part_s[o];
        idx = a.part_i ? a.part_i[o] : t;
        if (a.part_i && idx == IDX_NONE) valid = false;
      }
      sv[u] = valid ? s : __builtin_inff();
      iv[u] = idx;
      jv[u] = j;
      vv[u] = valid;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float s = sv[u];
      const bool valid = vv[u];
      uint64_t m = __ballot(valid && s <= tau);
      while (m) {
        const int l = __builtin_ctzll(m);
        const float cs = bcast_f(s, l);
        const uint32_t co = bcast_u(jv[u], l), ci = bcast_u(iv[u], l);
        m &= m - 1;
        if (sorted_insert<true>(Ls, Lo, Li, k, cnt, cs, co, ci, lane, tau)) m &= __ballot(valid && s <= tau);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();

  // the list is already sorted ascending by (score, slot, index) = the reference's result order
  for (uint32_t j = lane; j < k; j += 64) {
    const size_t o = (size_t)q * k + j;
    if (j < cnt) {
      const float v = Ls[j];
      const uint32_t vo = Lo[j], vi = Li[j];
      uint64_t key;
      if (a.part_keys)
        key = a.packed_stride ? *reinterpret_cast<const uint64_t *>(reinterpret_cast<const char *>(a.part_keys + (size_t)q * sl + vi) + (size_t)vo * a.packed_stride)
                              : a.part_keys[((size_t)sb + (size_t)vo * a.slot_stride) * sl + vi];
      else key = a.keymap ? a.keymap[vi] : (uint64_t)vi;
      a.out_keys[o] = key;
      a.out_scores[o] = v;
      if (a.out_idx) a.out_idx[o] = vi;
    } else {
      a.out_keys[o] = ~0ull;
      a.out_scores[o] = __builtin_inff();
      if (a.out_idx) a.out_idx[o] = IDX_NONE;
    }
  }
  if (lane == 0) a.out_counts[q] = cnt;
}

// ---------------------------------------------------------------------------------------------
// data-movement kernels
// ---------------------------------------------------------------------------------------------
// element (pos, e) of a blocked store whose rows hold fp32 (F16=false) or halves (F16=true); dpadw = words/row
template <bool F16>
__device__ __forceinline__ float load_elem(const float *base, uint64_t pos, uint32_t e, uint32_t dpadw) {
  if constexpr (F16) {
    const _Float16 *h = reinterpret_cast<const _Float16 *>(base);
    return (float)h[blocked_offset(pos, e >> 1, dpadw) * 2 + (e & 1)];
  } else {
    return base[blocked_offset(pos, e, dpadw)];
  }
}
template <bool F16>
__device__ __forceinline__ void store_elem(float *base, uint64_t pos, uint32_t e, uint32_t dpadw, float v) {
  if constexpr (F16) {
    _Float16 *h = reinterpret_cast<_Float16 *>(base);
    h[blocked_offset(pos, e >> 1, dpadw) * 2 + (e & 1)] = (_Float16)v;    // exact: v came from a half
  } else {
    base[blocked_offset(pos, e, dpadw)] = v;
  }
}
template <bool F16>
__device__ __forceinline__ float load_row_elem(const void *rows, size_t row, uint32_t dim_in, uint32_t c) {
  if constexpr (F16) return (float)reinterpret_cast<const _Float16 *>(rows)[row * dim_in + c];
  else return reinterpret_cast<const float *>(rows)[row * dim_in + c];
}

// one wave per row: rows [n][dim_in] (row-major, fp32 or fp16) -> blocked store at positions pos0 + i
// (or dst_pos[i]); writes the squared norm of the scanned dims (fp32); zero-fills the k padding.
template <bool F16>
__global__ void __launch_bounds__(256) pack_rows_kernel(const void *src, uint64_t n, uint32_t dim_in,
                                                        uint32_t dscan, uint32_t dpadw,
                                                        const uint64_t *src_row,   // nullable gather
                                                        uint64_t pos0, const uint64_t *dst_pos,
                                                        float *base, float *bnorm, float *extra /*cosine norm*/) {
  const int lane = threadIdx.x & 63;
  uint64_t i = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  uint64_t sr = src_row ? src_row[i] : i;
  uint64_t pos = dst_pos ? dst_pos[i] : pos0 + i;
  const uint32_t nelem = dpadw * (F16 ? 2u : 1u);
  float acc = 0.f;
  for (uint32_t c = lane; c < nelem; c += 64) {
    float v = (c < dscan) ? load_row_elem<F16>(src, sr, dim_in, c) : 0.f;
    store_elem<F16>(base, pos, c, dpadw, v);
    acc = fmaf(v, v, acc);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) {
    if (bnorm) bnorm[pos] = acc;
    if (extra) {
      // the stored norm column: one float, or (fp16 rows) the two half slots that hold its bits
      if constexpr (F16) {
        const uint16_t *h = reinterpret_cast<const uint16_t *>(src) + sr * dim_in + dscan;
        extra[pos] = (dim_in >= dscan + 2) ? __builtin_bit_cast(float, (uint32_t)h[0] | ((uint32_t)h[1] << 16)) : 0.f;
      } else {
        extra[pos] = (dim_in > dscan) ? load_row_elem<F16>(src, sr, dim_in, dscan) : 0.f;
      }
    }
  }
}

// queries [nq][dim_in] (fp32 or fp16) -> padded row-major [nq][dpadw words] of the same element type + squared norms
template <bool F16>
__global__ void __launch_bounds__(256) prep_queries_kernel(const void *src, uint32_t nq, uint32_t dim_in,
                                                           uint32_t dscan, uint32_t dpadw, float *dst,
                                                           float *qnorm, uint32_t *gtau, float threshold) {
  const int lane = threadIdx.x & 63;
  uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= nq) return;
  const uint32_t nelem = dpadw * (F16 ? 2u : 1u);
  float acc = 0.f;
  for (uint32_t c = lane; c < nelem; c += 64) {
    float v = (c < dscan) ? load_row_elem<F16>(src, i, dim_in, c) : 0.f;
    if constexpr (F16) reinterpret_cast<_Float16 *>(dst)[(size_t)i * nelem + c] = (_Float16)v;
    else dst[(size_t)i * dpadw + c] = v;
    acc = fmaf(v, v, acc);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) {
    qnorm[i] = acc;
    gtau[i] = fkey(threshold);
  }
}

// blocked row -> plain row (get_vector_by_id), in the store's element type
template <bool F16>
__global__ void unpack_row_kernel(const float *base, const float *extra, uint64_t pos, uint32_t dscan,
                                  uint32_t dim_out, uint32_t dpadw, void *out) {
  for (uint32_t c = threadIdx.x; c < dim_out; c += blockDim.x) {
    if constexpr (F16) {
      if (c < dscan) {
        reinterpret_cast<_Float16 *>(out)[c] = (_Float16)load_elem<F16>(base, pos, c, dpadw);
      } else {                                        // the norm's bits, low half first
        const uint32_t bits = extra ? __builtin_bit_cast(uint32_t, extra[pos]) : 0u;
        reinterpret_cast<uint16_t *>(out)[c] = (uint16_t)(c == dscan ? bits : bits >> 16);
      }
    } else {
      reinterpret_cast<float *>(out)[c] = (c < dscan) ? load_elem<F16>(base, pos, c, dpadw) : (extra ? extra[pos] : 0.f);
    }
  }
}

// one launch instead of two memsets + fill_gtau before the IVF plan: zero `nzero` plan words (list_count, list_fill),
// zero the 4 work-queue words, reset the shared bounds of `nq` queries to the threshold
// ... and set this search's chunk length of every list: `tpc` tiles, a quarter of that for the lists flagged as the tail
__global__ void ivf_reset_kernel(uint32_t *zero0, uint32_t nzero, uint32_t *queue, uint32_t *gtau, uint32_t nq, float threshold,
                                 uint32_t *list_tpc, const uint32_t *list_tail, uint32_t nlist, uint32_t tpc) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nzero) zero0[i] = 0;
  if (i < 4) queue[i] = 0;
  if (i < nq) gtau[i] = fkey(threshold);
  if (i < nlist) list_tpc[i] = list_tail[i] ? max(1u, tpc >> 2) : tpc;
}

// gtau[q] = min(gtau[q], k-th score of a sample scan, nudged up by ~1e-6 relative) — only for full sample lists.
// The k-th best score of ANY subset of the rows bounds the final k-th score from above, so starting every
// work-group of the main scan at that bound drops nothing it could keep; it only spares the list warm-up.
__global__ void seed_gtau_kernel(uint32_t *gtau, const float *scores, const uint32_t *counts, uint32_t n, uint32_t k) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || counts[i] < k) return;
  const float s = scores[(size_t)i * k + (k - 1)];
  const float b = s + fabsf(s) * 1e-6f + 1e-30f;
  if (b == b) atomicMin(&gtau[i], fkey(b));
}

__global__ void fill_keys_kernel(uint64_t *keys, uint64_t pos0, uint64_t n, const uint64_t *src) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keys[pos0 + i] = src ? src[i] : pos0 + i;
}

// bitset widening is not needed: the API bitset is uint64 words, bit i of word i/64 == bit (i&31) of
// 32-bit word i/32 on a little-endian host/device, so the kernel reads it as uint32 words directly.

// ---------------------------------------------------------------------------------------------
// L2 refinement of the final lists.  The scan forms squared distances as |q|^2 + |b|^2 - 2 q.b on the
// matrix cores, whose rounding error scales with the NORMS; the reference sums (q-b)^2 directly
// (euclidean_distance_matrix_fp32.cc:229-283), whose error scales with the DISTANCE (an identical vector
// scores exactly 0).  The k winners of every query are therefore re-scored directly (one wave per
// (query, result): a 3 KiB gather each, ~30 MB per 1024x10 batch) and the list is re-sorted by the
// refined score, previous rank breaking ties.
// ---------------------------------------------------------------------------------------------
template <bool F16>
__global__ void __launch_bounds__(256) rescore_l2_kernel(const float *base, const float *queries, uint32_t dpadw,
                                                         const uint32_t *idx, const uint32_t *counts, uint32_t nq,
                                                         uint32_t k, float *scores) {
  const int lane = threadIdx.x & 63;
  const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (uint64_t)nq * k) return;
  const uint32_t q = (uint32_t)(w / k), j = (uint32_t)(w - (uint64_t)q * k);
  if (j >= counts[q]) return;
  const uint32_t id = idx[w];
  const uint32_t nelem = dpadw * (F16 ? 2u : 1u);
  float acc = 0.f;
  for (uint32_t c = lane; c < nelem; c += 64) {
    float x;
    if constexpr (F16) x = (float)reinterpret_cast<const _Float16 *>(queries)[(size_t)q * nelem + c];
    else x = queries[(size_t)q * dpadw + c];
    const float d = x - load_elem<F16>(base, id, c, dpadw);
    acc = fmaf(d, d, acc);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) scores[w] = acc;
}

// one wave per query: stable re-sort of (score, key, idx) by score through LDS
__global__ void __launch_bounds__(64) resort_kernel(uint64_t *keys, float *scores, uint32_t *idx, uint32_t *counts,
                                                    uint32_t k, float threshold) {
  extern __shared__ f32x4 zvk_smem4[];
  float *S = reinterpret_cast<float *>(zvk_smem4);            // [k]
  uint32_t *I = reinterpret_cast<uint32_t *>(S + k);          // [k]
  uint64_t *K = reinterpret_cast<uint64_t *>(I + k + (k & 1)); // [k], 8-byte aligned
  const int lane = threadIdx.x;
  const uint32_t q = blockIdx.x;
  const uint32_t c = counts[q];
  for (uint32_t j = lane; j < c; j += 64) {
    S[j] = scores[(size_t)q * k + j];
    I[j] = idx[(size_t)q * k + j];
    K[j] = keys[(size_t)q * k + j];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  for (uint32_t j = lane; j < c; j += 64) {
    const float v = S[j];
    uint32_t rank = 0;
    for (uint32_t u = 0; u < c; ++u) {
      const float w = S[u];
      rank += (w < v || (w == v && u < j)) ? 1u : 0u;
    }
    const size_t o = (size_t)q * k + rank;
    scores[o] = v;
    idx[o] = I[j];
    keys[o] = K[j];
  }
  // RNN radius on the refined score: results past the threshold are cut (topk_to_result,
  // ivf_searcher_context.h:184-208 / flat_streamer_context.h)
  uint32_t keep = 0;
  for (uint32_t j0 = 0; j0 < c; j0 += 64) {
    const uint32_t j = j0 + lane;
    keep += (uint32_t)__popcll(__ballot(j < c && S[j] <= threshold));
  }
  if (lane == 0 && keep != c) counts[q] = keep;
}

// ---------------------------------------------------------------------------------------------
// Predicate materialisation: the reference evaluates its composite document
// filter — deleted(id) || !invert_result.contains(id) || !forward_bool[id] (doc_filter.cc:74-87, delete_store.h:
// 61-72, inverted_search_result.h:34-50) — once per CANDIDATE through a std::function.  Here the same predicate
// is evaluated once per STORAGE POSITION into the 1-bit-per-position exclude set the scan kernels gate on: one
// thread per position, one 64-bit output word per wave (a ballot).  The roaring bitmaps stay in their portable
// serialised form (CRoaring 2.0.4 `roaring_bitmap_portable_serialize`, RoaringFormatSpec) in HBM; the host only
// parses the container directory.  HBM-bound integer work: 8 B of key in, 1 bit out per position, plus the
// (cache-resident) container probes.
// ---------------------------------------------------------------------------------------------
struct RoaringView {
  const uint64_t *ckey;     // [nc] ascending: (high 32 bits of the id << 16) | container key
  const uint32_t *cinfo;    // [nc] type (bits 0-1: 0 array, 1 bitmap, 2 run) | element / run count << 2
  const uint64_t *coff;     // [nc] byte offset of the container payload inside `bytes`
  const uint8_t *bytes;     // the serialised stream
  uint32_t nc;
  uint32_t present;         // 0 = this term of the predicate is absent
  uint32_t trunc32;         // ids are cast to uint32 before the probe (32-bit bitmap behind a 64-bit id API)
};

struct DocFilterArgs {
  const uint64_t *keys;     // [n] document id of each storage position (nullptr => id = position)
  uint64_t n;
  // IVF: positions are list-order (dense) positions while `keys` is laid out by padded position
  const uint64_t *list_dense0;   // [nlist + 1] or nullptr
  const uint32_t *list_tile0;    // [nlist]
  uint32_t nlist;
  RoaringView del;          // set => excluded
  RoaringView inv;          // clear => excluded
  const uint8_t *forward;   // Arrow boolean bitmap (LSB first), clear => excluded; nullptr = absent
  uint64_t forward_len;     // ids >= forward_len are not excluded by this term (doc_filter.cc:104-107)
  uint64_t *out;            // [(n + 63) / 64]
};

__device__ __forceinline__ uint32_t ld_u16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

__device__ inline bool roaring_contains(const RoaringView &v, uint64_t id) {
  if (v.trunc32) id &= 0xffffffffull;
  const uint64_t ck = id >> 16;
  const uint32_t low = (uint32_t)(id & 0xffffu);
  uint32_t lo = 0, hi = v.nc;            // first container with key >= ck
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (v.ckey[mid] < ck) lo = mid + 1; else hi = mid;
  }
  if (lo >= v.nc || v.ckey[lo] != ck) return false;
  const uint32_t info = v.cinfo[lo];
  const uint32_t type = info & 3u, cnt = info >> 2;
  const uint8_t *pl = v.bytes + v.coff[lo];
  if (type == 1u) return (pl[low >> 3] >> (low & 7u)) & 1u;
  if (type == 0u) {                      // sorted u16 values
    uint32_t a = 0, b = cnt;
    while (a < b) {
      const uint32_t m = (a + b) >> 1;
      if (ld_u16(pl + 2 * m) < low) a = m + 1; else b = m;
    }
    return a < cnt && ld_u16(pl + 2 * a) == low;
  }
  // runs (start, length - 1), ascending: last run with start <= low
  uint32_t a = 0, b = cnt;
  while (a < b) {
    const uint32_t m = (a + b) >> 1;
    if (ld_u16(pl + 4 * m) <= low) a = m + 1; else b = m;
  }
  if (a == 0) return false;
  const uint32_t start = ld_u16(pl + 4 * (a - 1)), len1 = ld_u16(pl + 4 * (a - 1) + 2);
  return low - start <= len1;
}

__global__ void __launch_bounds__(256) doc_filter_kernel(const DocFilterArgs a) {
  const uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  bool excl = false;
  if (p < a.n) {
    uint64_t kpos = p;
    if (a.list_dense0 != nullptr) {
      uint32_t lo = 0, hi = a.nlist;       // last list with dense0 <= p
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.list_dense0[mid] <= p) lo = mid; else hi = mid;
      }
      kpos = (uint64_t)a.list_tile0[lo] * TILE_N + (p - a.list_dense0[lo]);
    }
    const uint64_t id = a.keys ? a.keys[kpos] : kpos;
    if (a.del.present) excl = roaring_contains(a.del, id);
    if (!excl && a.inv.present) excl = !roaring_contains(a.inv, id);
    if (!excl && a.forward != nullptr && id < a.forward_len) excl = !((a.forward[id >> 3] >> (id & 7u)) & 1u);
  }
  const uint64_t word = __ballot(excl);
  if ((threadIdx.x & 63) == 0 && p < a.n) a.out[p >> 6] = word;
}

// ---------------------------------------------------------------------------------------------
// Sparse keep-sets (bitmap-gated scan, BASELINE configs[4]): when the predicate keeps a minority of the rows
// the kept rows are first compacted into a temporary blocked store (stream compaction of the bitset, then a
// row copy between two blocked layouts) and the dense scan runs over that — work proportional to the KEPT rows,
// as on the CPU where filtered rows are skipped before the distance (flat_searcher_context.h:949-963).
//   1. keep_count_kernel : kept rows per 2048-bit chunk      2. (host-launched) exclusive scan of the counts
//   3. keep_fill_kernel  : kept positions, ascending         4. compact_rows_kernel: row copy + norms + keys
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) keep_count_kernel(const uint32_t *excl, uint64_t n, uint32_t *chunk_cnt) {
  // one work-group per 2048 rows = 64 words; 64 lanes of wave 0 suffice
  const uint64_t w0 = (uint64_t)blockIdx.x * 64;
  const int lane = threadIdx.x;
  if (lane >= 64) return;
  const uint64_t w = w0 + lane;
  const uint64_t nwords = (n + 31) / 32;
  uint32_t keep = 0;
  if (w < nwords) {
    uint32_t bits = ~excl[w];
    const uint64_t rem = n - w * 32;
    if (rem < 32) bits &= (1u << rem) - 1u;
    keep = (uint32_t)__popc(bits);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) keep += __shfl_xor(keep, off);
  if (lane == 0) chunk_cnt[blockIdx.x] = keep;
}

__global__ void __launch_bounds__(1024) u32_exclusive_scan_kernel(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *total) {
  __shared__ uint32_t sh[1024];
  __shared__ uint32_t carry;
  const int tid = threadIdx.x;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n; base += 1024) {
    const uint32_t i = base + tid;
    const uint32_t v = (i < n) ? in[i] : 0;
    sh[tid] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      uint32_t t = (tid >= off) ? sh[tid - off] : 0;
      __syncthreads();
      sh[tid] += t;
      __syncthreads();
    }
    const uint32_t incl = sh[tid], c = carry;
    if (i < n) out[i] = c + incl - v;
    __syncthreads();
    if (tid == 1023) carry = c + incl;
    __syncthreads();
  }
  if (tid == 0) *total = carry;
}

__global__ void __launch_bounds__(64) keep_fill_kernel(const uint32_t *excl, uint64_t n, const uint32_t *chunk_off, uint32_t *pos) {
  const int lane = threadIdx.x;
  const uint64_t w = (uint64_t)blockIdx.x * 64 + lane;
  const uint64_t nwords = (n + 31) / 32;
  uint32_t bits = 0;
  if (w < nwords) {
    bits = ~excl[w];
    const uint64_t rem = n - w * 32;
    if (rem < 32) bits &= (1u << rem) - 1u;
  }
  const uint32_t cnt = (uint32_t)__popc(bits);
  uint32_t incl = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  uint32_t o = chunk_off[blockIdx.x] + incl - cnt;
  while (bits) {
    const int b = __builtin_ctz(bits);
    bits &= bits - 1;
    pos[o++] = (uint32_t)(w * 32 + b);
  }
}

// one wave per kept row: copy the row between two blocked stores (same dpadw), with its norm, key and extra
__global__ void __launch_bounds__(256) compact_rows_kernel(const float *src, const float *src_norm, const float *src_extra,
                                                           const uint64_t *src_keys, const uint32_t *pos, uint32_t kept,
                                                           uint32_t dpadw, float *dst, float *dst_norm, float *dst_extra,
                                                           uint64_t *dst_keys) {
  const int lane = threadIdx.x & 63;
  const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= kept) return;
  const uint32_t p = pos[i];
  // 16-byte chunks: chunk c of row r lives at word offset tile*128*dpadw + (c/8)*4096 + (r*8 + ((c%8) ^ swz(r)))*4
  const uint32_t nchunks = dpadw / 4;
  const f32x4 *s4 = reinterpret_cast<const f32x4 *>(src);
  f32x4 *d4 = reinterpret_cast<f32x4 *>(dst);
  const uint32_t sr = p & 127, dr = i & 127;
  const size_t sbase = (size_t)(p >> 7) * TILE_N * dpadw / 4, dbase = (size_t)(i >> 7) * TILE_N * dpadw / 4;
  for (uint32_t c = lane; c < nchunks; c += 64) {
    const uint32_t ks = c >> 3, cc = c & 7;
    const size_t so = sbase + (size_t)ks * (SLAB / 4) + sr * 8 + (cc ^ ((sr >> 1) & 7));
    const size_t dofs = dbase + (size_t)ks * (SLAB / 4) + dr * 8 + (cc ^ ((dr >> 1) & 7));
    d4[dofs] = s4[so];
  }
  if (lane == 0) {
    dst_norm[i] = src_norm[p];
    dst_keys[i] = src_keys[p];
    if (dst_extra && src_extra) dst_extra[i] = src_extra[p];
  }
}

// ---------------------------------------------------------------------------------------------
// brute force by primary keys (FlatStreamer::search_bf_by_p_keys_impl, flat_streamer.cc:346-389): every
// query comes with its own short list of storage positions; one wave scores one (query, position) pair
// DIRECTLY (sum of (q-b)^2 / q.b over the row, no norm expansion) — the path is taken when a filter is so
// selective that gathering beats scanning.  Scores land in a padded [nq][maxlen] matrix for merge_kernel.
// ---------------------------------------------------------------------------------------------
template <bool F16>
__global__ void __launch_bounds__(256) pkeys_score_kernel(const float *base, const float *queries, uint32_t dpadw,
                                                          int metric, const uint32_t *pos, const uint32_t *off,
                                                          uint32_t nq, uint32_t maxlen, float *out_s, uint32_t *out_i) {
  const int lane = threadIdx.x & 63;
  const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (uint64_t)nq * maxlen) return;
  const uint32_t q = (uint32_t)(w / maxlen), j = (uint32_t)(w - (uint64_t)q * maxlen);
  const uint32_t len = off[q + 1] - off[q];
  const uint32_t nelem = dpadw * (F16 ? 2u : 1u);
  float sc = __builtin_inff();
  uint32_t id = IDX_NONE;
  if (j < len) {
    id = pos[off[q] + j];
    if (id != IDX_NONE) {
      float acc = 0.f;
      for (uint32_t c = lane; c < nelem; c += 64) {
        const float b = load_elem<F16>(base, id, c, dpadw);
        float x;
        if constexpr (F16) x = (float)reinterpret_cast<const _Float16 *>(queries)[(size_t)q * nelem + c];
        else x = queries[(size_t)q * dpadw + c];
        if (metric == METRIC_L2) { const float d = x - b; acc = fmaf(d, d, acc); }
        else acc = fmaf(x, b, acc);
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
      sc = (metric == METRIC_L2) ? acc : (metric == METRIC_IP ? -acc : 1.f - acc);
    }
  }
  if (lane == 0) {
    out_s[w] = sc;
    out_i[w] = id;
  }
}

// ---------------------------------------------------------------------------------------------
// IVF plan kernels: turn the per-query probe lists into list-major work.
// ---------------------------------------------------------------------------------------------
struct PlanArgs {
  const uint32_t *coarse_idx;     // [nq][nprobe] list ids in probe order (IDX_NONE = none)
  const uint32_t *coarse_cnt;     // [nq]
  uint32_t nq, nprobe, nlist;
  uint32_t max_scan_count;
  int brute_force;                // probe every list in id order
  const uint32_t *list_size;      // stored rows (this shard)
  const uint32_t *list_size_global;  // rows of the whole index (scan-count rule)
  const uint32_t *list_order;        // [nlist] lists by stored size, largest first
  const uint32_t *list_tpc;       // [nlist] tiles per chunk of each list
  uint32_t rows_per_group;        // NG*32 of the scan kernel
  // outputs
  uint32_t *q_nprobe;             // [nq] lists actually probed (IndexContext::Stats)
  uint32_t *q_scanned;            // [nq] total_scan_count
  uint32_t *q_nslots;             // [nq]
  uint32_t *slot_begin;           // [nq+1]
  uint32_t *list_count;           // [nlist] queries probing the list (zeroed before)
  uint32_t *list_fill;            // [nlist] fill cursors (zeroed before)
  uint32_t *list_qoff;            // [nlist+1]
  uint32_t *item_off;             // [nlist+1]
  uint32_t *total_items;          // [1]
  uint32_t *csr_q, *csr_slot;
};

__device__ __forceinline__ uint32_t list_chunks(uint32_t size, uint32_t tiles_per_chunk) {
  uint32_t tiles = (size + TILE_N - 1) / TILE_N;
  return (tiles + tiles_per_chunk - 1) / tiles_per_chunk;
}

// probe rule of IVFSearcher::search_impl (ivf_searcher.cc:223-237): walk the coarse result in
// order while total_scan_count < max_scan_count; every probed list adds its full vector_count.
__device__ __forceinline__ uint32_t probe_list(const PlanArgs &p, uint32_t q, uint32_t rank) {
  return p.brute_force ? rank : p.coarse_idx[(size_t)q * p.nprobe + rank];
}

// wave-wide inclusive prefix sum (6 shuffle steps)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t t = __shfl_up(v, off);
    if (lane >= off) v += t;
  }
  return v;
}

// One wave per query, one lane per probe rank (64 ranks per pass): evaluates the probe rule with a
// prefix sum of the global list sizes instead of a serial walk.
//   probed(rank)  <=>  sum of vector_count of the lists before it  <  max_scan_count
// FILL = false: counts (q_nprobe, q_scanned, q_nslots, list_count); FILL = true: writes the CSR.
template <bool FILL>
__global__ void __launch_bounds__(256) plan_wave_kernel(const PlanArgs p) {
  const int lane = threadIdx.x & 63;
  const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= p.nq) return;
  const uint32_t np = p.brute_force ? p.nlist : min(p.coarse_cnt[q], p.nprobe);
  uint32_t scanned_before = 0;   // uniform carries across 64-rank passes
  uint32_t slot_carry = FILL ? p.slot_begin[q] : 0;
  uint32_t probes = 0, scanned = 0;
  for (uint32_t r0 = 0; r0 < np; r0 += 64) {
    const uint32_t rnk = r0 + lane;
    const bool in = rnk < np;
    const uint32_t l = in ? probe_list(p, q, rnk) : 0;
    const uint32_t szg = in ? p.list_size_global[l] : 0;
    const uint32_t incl = wave_incl_scan(szg, lane);
    const uint32_t before = scanned_before + incl - szg;
    const bool probed = in && (p.brute_force || before < p.max_scan_count);
    const uint32_t szl = probed ? p.list_size[l] : 0;
    const uint32_t ch = szl ? list_chunks(szl, p.list_tpc[l]) : 0;
    const uint32_t chincl = wave_incl_scan(ch, lane);
    if (FILL) {
      if (szl) {
        const uint32_t e = p.list_qoff[l] + atomicAdd(&p.list_fill[l], 1u);
        p.csr_q[e] = q;
        p.csr_slot[e] = slot_carry + chincl - ch;
      }
    } else {
      if (szl) atomicAdd(&p.list_count[l], 1u);
      probes += (uint32_t)__popcll(__ballot(probed));
      const uint32_t probed_sz = wave_incl_scan(probed ? szg : 0, lane);
      scanned += __shfl(probed_sz, 63);
    }
    slot_carry += __shfl(chincl, 63);
    scanned_before += __shfl(incl, 63);
    if (!p.brute_force && scanned_before >= p.max_scan_count) break;   // uniform
  }
  if (!FILL && lane == 0) {
    p.q_nprobe[q] = probes;
    p.q_scanned[q] = scanned;
    p.q_nslots[q] = slot_carry;
  }
}

// Large-k fallback of the IVF search (k beyond the LDS-resident lists of the scan kernel): one wave per query walks
// its probe ranks with the same probe rule as plan_wave_kernel and either counts the rows it will scan on this shard
// (FILL = false) or writes their padded positions — excluded rows as holes — for pkeys_score_kernel (FILL = true).
template <bool FILL>
__global__ void __launch_bounds__(256) ivf_expand_kernel(const PlanArgs p, const uint32_t *list_tile0, const uint64_t *list_dense0,
                                                         const uint32_t *exclude, uint32_t *q_rows, const uint32_t *q_off,
                                                         uint32_t *pos) {
  const int lane = threadIdx.x & 63;
  const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= p.nq) return;
  const uint32_t np = p.brute_force ? p.nlist : min(p.coarse_cnt[q], p.nprobe);
  uint32_t scanned_before = 0, rows = 0;     // uniform
  uint32_t o = FILL ? q_off[q] : 0;
  for (uint32_t rnk = 0; rnk < np; ++rnk) {
    if (!p.brute_force && scanned_before >= p.max_scan_count) break;
    const uint32_t l = probe_list(p, q, rnk);
    scanned_before += p.list_size_global[l];
    const uint32_t sz = p.list_size[l];
    if (FILL) {
      const uint32_t p0 = list_tile0[l] * TILE_N;
      const uint64_t d0 = list_dense0[l];
      for (uint32_t j = lane; j < sz; j += 64) {
        bool ex = false;
        if (exclude) { const uint64_t d = d0 + j; ex = (exclude[d >> 5] >> (d & 31)) & 1u; }
        pos[o + j] = ex ? IDX_NONE : p0 + j;
      }
      o += sz;
    }
    rows += sz;
  }
  if (!FILL && lane == 0) q_rows[q] = rows;
}

// single work-group exclusive scans: slot_begin over queries, list_qoff / item_off over lists
__global__ void __launch_bounds__(1024) plan_scan_kernel(const PlanArgs p) {
  __shared__ uint32_t wtot[16];
  __shared__ uint32_t carry;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  // 1024 elements per round: wave-level shuffle scan, 16 wave totals through LDS (two barriers per round)
  auto block_scan = [&](auto getv, auto putv, uint32_t n, uint32_t *total_out) {
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
      const uint32_t i = base + tid;
      const uint32_t v = (i < n) ? getv(i) : 0;
      const uint32_t incl = wave_incl_scan(v, lane);
      if (lane == 63) wtot[wave] = incl;
      __syncthreads();
      uint32_t before = carry;
      for (int w = 0; w < wave; ++w) before += wtot[w];
      if (i < n) putv(i, before + incl - v);
      __syncthreads();
      if (tid == 1023) carry = before + incl;
    }
    __syncthreads();
    if (tid == 0) *total_out = carry;
    __syncthreads();
  };
  block_scan([&](uint32_t i) { return p.q_nslots[i]; }, [&](uint32_t i, uint32_t v) { p.slot_begin[i] = v; },
             p.nq, &p.slot_begin[p.nq]);
  block_scan([&](uint32_t i) { return p.list_count[i]; }, [&](uint32_t i, uint32_t v) { p.list_qoff[i] = v; },
             p.nlist, &p.list_qoff[p.nlist]);
  block_scan(
      [&](uint32_t i) {
        const uint32_t l = p.list_order[i];
        uint32_t c = p.list_count[l];
        uint32_t groups = (c + p.rows_per_group - 1) / p.rows_per_group;
        return groups * list_chunks(p.list_size[l], p.list_tpc[l]);
      },
      [&](uint32_t i, uint32_t v) { p.item_off[i] = v; }, p.nlist, &p.item_off[p.nlist]);
  if (tid == 0) *p.total_items = p.item_off[p.nlist];
}

// ---------------------------------------------------------------------------------------------
// "ivf.inverted_body" of a dumped reference index -> plain rows in list order (SURVEY next-2).  Layout written by
// IVFDumper (src/core/algorithm/ivf/ivf_dumper.cc:19-81,388-406; ivf_dumper.h:33-160): per inverted list, at
// InvertedListMeta::offset, blocks of `bvc` (32) vectors, each block padded to 32 bytes; a FULL block of a
// column-major index is transposed in units of the element's alignment (unit u of vector i at (u*bvc + i)*unit),
// every other block is row-major.  One wave per row; pure byte movement.
// ---------------------------------------------------------------------------------------------
struct IvfBodyArgs {
  const uint8_t *body;
  const uint64_t *list_off;     // [nlist] byte offset of each list in the body
  const uint64_t *list_row0;    // [nlist + 1] first global row (InvertedListMeta::id_offset), last = total
  uint32_t nlist;
  uint32_t bvc;                 // block_vector_count
  uint32_t block_size;          // bytes of a full block
  uint32_t elem_size;           // bytes per vector
  uint32_t unit;                // alignment unit of the element type (2 = fp16, 4 = fp32)
  uint32_t column_major;
  uint8_t *rows;                // out: [total][elem_size]
  uint64_t total;
};

__global__ void __launch_bounds__(256) ivf_body_rows_kernel(const IvfBodyArgs a) {
  const int lane = threadIdx.x & 63;
  const uint64_t g = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= a.total) return;
  uint32_t lo = 0, hi = a.nlist;          // last list with row0 <= g
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a.list_row0[mid] <= g) lo = mid; else hi = mid;
  }
  while (lo + 1 < a.nlist && a.list_row0[lo + 1] <= g) ++lo;    // (empty lists share a row0)
  const uint64_t i = g - a.list_row0[lo];
  const uint64_t cnt = a.list_row0[lo + 1] - a.list_row0[lo];
  const uint64_t blk = i / a.bvc, r = i % a.bvc;
  const bool full = (blk + 1) * a.bvc <= cnt;
  const uint8_t *b0 = a.body + a.list_off[lo] + blk * a.block_size;
  uint8_t *dst = a.rows + g * a.elem_size;
  const uint32_t units = a.elem_size / a.unit;
  if (a.column_major && full) {
    if (a.unit == 4) {
      for (uint32_t u = lane; u < units; u += 64)
        reinterpret_cast<uint32_t *>(dst)[u] = reinterpret_cast<const uint32_t *>(b0)[(size_t)u * a.bvc + r];
    } else {
      for (uint32_t u = lane; u < units; u += 64)
        reinterpret_cast<uint16_t *>(dst)[u] = reinterpret_cast<const uint16_t *>(b0)[(size_t)u * a.bvc + r];
    }
  } else {
    const uint8_t *src = b0 + r * a.elem_size;
    if (a.unit == 4) {
      for (uint32_t u = lane; u < units; u += 64) reinterpret_cast<uint32_t *>(dst)[u] = reinterpret_cast<const uint32_t *>(src)[u];
    } else {
      for (uint32_t u = lane; u < units; u += 64) reinterpret_cast<uint16_t *>(dst)[u] = reinterpret_cast<const uint16_t *>(src)[u];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k-means helpers (IVF build): mean of member rows per cluster, members given as CSR of row ids.
// ---------------------------------------------------------------------------------------------
template <bool F16>
__global__ void __launch_bounds__(256) centroid_mean_kernel(const void *rows, uint32_t dim,
                                                            const uint64_t *member_off,
                                                            const uint64_t *members, void *centroids) {
  const uint32_t c = blockIdx.x;
  const uint64_t b = member_off[c], e = member_off[c + 1];
  if (e == b) return;  // empty cluster keeps its previous centroid
  const float inv = 1.0f / (float)(e - b);
  for (uint32_t col = threadIdx.x; col < dim; col += blockDim.x) {
    float acc = 0.f;
    for (uint64_t m = b; m < e; ++m) acc += load_row_elem<F16>(rows, members[m], dim, col);
    if constexpr (F16) reinterpret_cast<_Float16 *>(centroids)[(size_t)c * dim + col] = (_Float16)(acc * inv);   // RNE
    else reinterpret_cast<float *>(centroids)[(size_t)c * dim + col] = acc * inv;
  }
}

// row gather in bytes (element-type agnostic)
__global__ void gather_rows_kernel(const void *rows, uint32_t row_bytes, const uint64_t *ids, uint64_t n, void *out) {
  uint64_t i = blockIdx.x;
  if (i >= n) return;
  const uint16_t *src = reinterpret_cast<const uint16_t *>(rows) + (size_t)ids[i] * (row_bytes / 2);
  uint16_t *dst = reinterpret_cast<uint16_t *>(out) + (size_t)i * (row_bytes / 2);
  for (uint32_t c = threadIdx.x; c < row_bytes / 2; c += blockDim.x) dst[c] = src[c];
}

}  // namespace zvk
