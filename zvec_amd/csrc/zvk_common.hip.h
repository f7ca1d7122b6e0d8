// zvk_common.hip.h — constants, launch arguments, wave helpers, the bounded sorted list and the owner-wave admission.
// Part of the device code of libzvec_hip (included through scan_kernels.hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

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

// LDS stores the compiler does not see (no `vmcnt(0)` in front of them while LDS-DMA is in flight); the caller waits on lgkmcnt
// before the barrier that publishes them.
__device__ __forceinline__ uint32_t lds_off(const void *p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
__device__ __forceinline__ void lds_store_u32(uint32_t off, uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(off), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_store_u64(uint32_t off, uint64_t v) { asm volatile("ds_write_b64 %0, %1" ::"v"(off), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_store_f32(float *p, float v) { lds_store_u32(lds_off(p), __builtin_bit_cast(uint32_t, v)); }

// One lane inserts candidate (s, pos) into the sorted list of `row` (the lanes of a wave work on different rows at once).  The
// same admission rule and the same kept set as owner_row / sorted_insert (zvk_common.hip.h): s <= min(tau, gt), ascending
// (score, position), the k smallest kept.
constexpr uint32_t LANE_INSERT_MAX_K = 11;                  // (the list is held in registers: 2 x 11; scan256's LDS holds no longer lists either)
__device__ __forceinline__ void lane_insert(const RowState &st, int row, float s, uint32_t pos) {
  constexpr int KMAX = (int)LANE_INSERT_MAX_K;
  const uint32_t k = st.k;
  float tl = st.tau[row];
  const float tg = st.gt[row];
  uint32_t c = st.cnt[row];
  float *L = st.Ls + (size_t)row * k;
  uint32_t *I = st.Li + (size_t)row * k;
  // the whole list in one round of loads: the number of entries in front of the candidate, then the tail moves up by stores only
  float es[KMAX];
  uint32_t ei[KMAX];
#pragma unroll
  for (int t = 0; t < KMAX; ++t) {
    const bool in = (uint32_t)t < c;
    es[t] = in ? L[t] : 0.f;
    ei[t] = in ? I[t] : 0u;
  }
  if (!(s <= fminf(tl, tg))) return;
  uint32_t p = 0;
#pragma unroll
  for (int t = 0; t < KMAX; ++t) p += ((uint32_t)t < c && (es[t] < s || (es[t] == s && ei[t] < pos))) ? 1u : 0u;
  if (p >= k) return;                                       // k entries precede it
  const uint32_t hi = min(c, k - 1);                        // entries [p, hi) move up by one
#pragma unroll
  for (int t = 0; t < KMAX - 1; ++t) {
    if ((uint32_t)t >= p && (uint32_t)t < hi) {
      L[t + 1] = es[t];
      I[t + 1] = ei[t];
    }
  }
  L[p] = s;
  I[p] = pos;
  c = min(c + 1, k);
  st.cnt[row] = c;
  if (c == k) {
    tl = s;                                                 // the new k-th: the candidate, or the old (k-1)-th it pushed up
#pragma unroll
    for (int t = 0; t < KMAX - 1; ++t)
      if (p != k - 1 && (uint32_t)t + 2 == k) tl = es[t];
    st.tau[row] = tl;
    if (tl < tg) atomicMin(&st.gtau[st.qrow[row]], fkey(tl));
  }
  st.tq[row] = fminf(tl, tg);
}

}  // namespace zvk
