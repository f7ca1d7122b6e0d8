// zvk_filter.hip.h — predicate materialisation (roaring -> bitset), kept-position lists, compaction, scoring by position lists.
// Part of the device code of libzvec_hip (included through scan_kernels.hip.h).
#pragma once
#include "zvk_common.hip.h"
#include "zvk_rows.hip.h"

namespace zvk {

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
// a wave takes PKEYS_ROWS consecutive entries of one query's list: their loads are issued together (12 independent
// 16-byte loads per lane at d = 768 instead of 3), and four times fewer waves have to be launched
constexpr uint32_t PKEYS_ROWS = 4;
// the wave's PKEYS_ROWS rows (positions id[], IDX_NONE = hole) against one prepared query row: every lane returns the scores
template <bool F16>
__device__ __forceinline__ void pkeys_rows_distance(const float *base, const float *qrow, uint32_t dpadw, int metric,
                                                    const uint32_t (&id)[PKEYS_ROWS], int lane, float (&sc)[PKEYS_ROWS]) {
  const float *trow[PKEYS_ROWS];
  uint32_t swz[PKEYS_ROWS];
#pragma unroll
  for (uint32_t r = 0; r < PKEYS_ROWS; ++r) {
    const uint32_t p = (id[r] != IDX_NONE) ? id[r] : 0u;          // (holes read row 0 and are discarded)
    const uint32_t row = p & 127;
    trow[r] = base + (size_t)(p >> 7) * TILE_N * dpadw + (size_t)(row * 8) * 4;
    swz[r] = (row >> 1) & 7;
  }
  const uint32_t nchunks = dpadw >> 2;
  float acc[PKEYS_ROWS];
#pragma unroll
  for (uint32_t r = 0; r < PKEYS_ROWS; ++r) acc[r] = 0.f;
  for (uint32_t c0 = lane; c0 < nchunks; c0 += 64) {
    const uint32_t ks = c0 >> 3, c = c0 & 7;
    const f32x4 qv = *reinterpret_cast<const f32x4 *>(qrow + (size_t)ks * TILE_K + c * 4);
    f32x4 bv[PKEYS_ROWS];
#pragma unroll
    for (uint32_t r = 0; r < PKEYS_ROWS; ++r)
      bv[r] = *reinterpret_cast<const f32x4 *>(trow[r] + (size_t)ks * SLAB + (size_t)((c ^ swz[r]) * 4));
#pragma unroll
    for (uint32_t r = 0; r < PKEYS_ROWS; ++r) {
      if constexpr (F16) {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        const h8 bh = __builtin_bit_cast(h8, bv[r]), qh = __builtin_bit_cast(h8, qv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = (float)qh[e], b = (float)bh[e];
          if (metric == METRIC_L2) { const float d = x - b; acc[r] = fmaf(d, d, acc[r]); }
          else acc[r] = fmaf(x, b, acc[r]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (metric == METRIC_L2) { const float d = qv[e] - bv[r][e]; acc[r] = fmaf(d, d, acc[r]); }
          else acc[r] = fmaf(qv[e], bv[r][e], acc[r]);
        }
      }
    }
  }
#pragma unroll
  for (uint32_t r = 0; r < PKEYS_ROWS; ++r) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc[r] += __shfl_xor(acc[r], o);
    sc[r] = (metric == METRIC_L2) ? acc[r] : (metric == METRIC_IP ? -acc[r] : 1.f - acc[r]);
  }
}

template <bool F16>
__global__ void __launch_bounds__(256) pkeys_score_kernel(const float *base, const float *queries, uint32_t dpadw,
                                                          int metric, const uint32_t *pos, const uint32_t *off,
                                                          uint32_t nq, uint32_t maxlen, float *out_s, uint32_t *out_i) {
  const int lane = threadIdx.x & 63;
  const uint32_t groups = (maxlen + PKEYS_ROWS - 1) / PKEYS_ROWS;
  const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (uint64_t)nq * groups) return;
  const uint32_t q = (uint32_t)(w / groups), j0 = (uint32_t)(w - (uint64_t)q * groups) * PKEYS_ROWS;
  const uint32_t len = off[q + 1] - off[q];
  uint32_t id[PKEYS_ROWS];
#pragma unroll
  for (uint32_t r = 0; r < PKEYS_ROWS; ++r) {
    const uint32_t j = j0 + r;
    id[r] = (j < len) ? pos[off[q] + j] : IDX_NONE;
  }
  float sc[PKEYS_ROWS];
  pkeys_rows_distance<F16>(base, queries + (size_t)q * dpadw, dpadw, metric, id, lane, sc);
  if (lane < (int)PKEYS_ROWS && j0 + lane < maxlen) {
    float a = sc[0];
    uint32_t i = id[0];
#pragma unroll
    for (uint32_t r = 1; r < PKEYS_ROWS; ++r)
      if ((uint32_t)lane == r) { a = sc[r]; i = id[r]; }
    out_s[(size_t)q * maxlen + j0 + lane] = (i != IDX_NONE) ? a : __builtin_inff();
    out_i[(size_t)q * maxlen + j0 + lane] = i;
  }
}

// The same scoring with the first selection step folded in (the single-query IVF route): a block takes PKEYS_BLOCK consecutive
// candidates of one query's stream and writes only their k best — sorted by (score, place in the stream) — as list
// (query, block) of a [nq][blocks][k] matrix.  The stream's 4 bytes of score + 4 of position per candidate are never written,
// and ONE merge of the lists (ordered by list, then entry = by place in the stream again) finishes the selection where the
// score matrix needed two.  Unused entries: +inf / IDX_NONE.  k <= PKEYS_TOPK_MAX.
constexpr uint32_t PKEYS_BLOCK = 128;
constexpr uint32_t PKEYS_TOPK_MAX = 64;
template <bool F16>
__global__ void __launch_bounds__(256) pkeys_topk_kernel(const float *base, const float *queries, uint32_t dpadw, int metric,
                                                         const uint32_t *pos, const uint32_t *off, uint32_t nq, uint32_t maxlen,
                                                         uint32_t k, float *out_s, uint32_t *out_i) {
  __shared__ unsigned long long s_key[PKEYS_BLOCK];      // order-preserving score key << 32 | place in the block; ~0 = hole
  __shared__ float s_sc[PKEYS_BLOCK];
  __shared__ uint32_t s_pos[PKEYS_BLOCK];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t bpq = (maxlen + PKEYS_BLOCK - 1) / PKEYS_BLOCK;
  const uint32_t q = blockIdx.x / bpq, b = blockIdx.x - q * bpq;
  const uint32_t len = min(off[q + 1] - off[q], maxlen);
  const float *qrow = queries + (size_t)q * dpadw;
  constexpr uint32_t PER_WAVE = PKEYS_BLOCK / 4;
#pragma unroll 2
  for (uint32_t g = 0; g < PER_WAVE / PKEYS_ROWS; ++g) {
    const uint32_t t0 = wave * PER_WAVE + g * PKEYS_ROWS, j0 = b * PKEYS_BLOCK + t0;
    if (j0 >= len) {                                       // uniform: past the stream
      if (lane < (int)PKEYS_ROWS) s_key[t0 + lane] = ~0ull;
      continue;
    }
    uint32_t id[PKEYS_ROWS];
#pragma unroll
    for (uint32_t r = 0; r < PKEYS_ROWS; ++r) id[r] = (j0 + r < len) ? pos[off[q] + j0 + r] : IDX_NONE;
    float sc[PKEYS_ROWS];
    pkeys_rows_distance<F16>(base, qrow, dpadw, metric, id, lane, sc);
    if (lane < (int)PKEYS_ROWS) {
      float a = sc[0];
      uint32_t i = id[0];
#pragma unroll
      for (uint32_t r = 1; r < PKEYS_ROWS; ++r)
        if ((uint32_t)lane == r) { a = sc[r]; i = id[r]; }
      s_key[t0 + lane] = (i != IDX_NONE) ? (((unsigned long long)fkey(a + 0.f) << 32) | (t0 + lane)) : ~0ull;
      s_sc[t0 + lane] = a;
      s_pos[t0 + lane] = i;
    }
  }
  __syncthreads();
  if (wave != 0) return;
  constexpr int KPL = PKEYS_BLOCK / 64;                   // keys per lane
  unsigned long long kk[KPL];
#pragma unroll
  for (int e = 0; e < KPL; ++e) kk[e] = s_key[e * 64 + lane];
  float rs = __builtin_inff();
  uint32_t ri = IDX_NONE;
  for (uint32_t r = 0; r < k; ++r) {                       // k rounds of a wave-wide minimum (keys are distinct)
    unsigned long long m = kk[0];
#pragma unroll
    for (int e = 1; e < KPL; ++e) m = kk[e] < m ? kk[e] : m;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const uint32_t hi = __shfl_xor((uint32_t)(m >> 32), o), lo = __shfl_xor((uint32_t)m, o);
      const unsigned long long other = ((unsigned long long)hi << 32) | lo;
      m = other < m ? other : m;
    }
    if (m == ~0ull) break;                                 // uniform: nothing left
    if ((uint32_t)lane == r) { const uint32_t t = (uint32_t)m & (PKEYS_BLOCK - 1); rs = s_sc[t]; ri = s_pos[t]; }
#pragma unroll
    for (int e = 0; e < KPL; ++e)
      if (kk[e] == m) kk[e] = ~0ull;
  }
  if ((uint32_t)lane < k) {
    const size_t o = ((size_t)q * bpq + b) * k + lane;
    out_s[o] = rs;
    out_i[o] = ri;
  }
}

// host side: number of 256-thread blocks pkeys_score_kernel needs for nq lists of up to maxlen entries
inline unsigned pkeys_score_blocks(uint64_t nq, uint64_t maxlen) {
  return (unsigned)((nq * ((maxlen + PKEYS_ROWS - 1) / PKEYS_ROWS) + 3) / 4);
}

// every stored row against every query of a SMALL batch, one wave per (query, row), direct distance: the coarse step of a
// handful of queries (the MFMA tile kernel would run one work-group per 128 centroids, each a chain of dependent loads)
template <bool F16>
__global__ void __launch_bounds__(256) rows_score_kernel(const float *base, const float *queries, uint32_t dpadw, int metric,
                                                         uint32_t n, uint32_t nq, uint32_t stride, float *out_s) {
  const int lane = threadIdx.x & 63;
  const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (uint64_t)nq * stride) return;
  const uint32_t q = (uint32_t)(w / stride), id = (uint32_t)(w - (uint64_t)q * stride);
  float sc = __builtin_inff();
  if (id < n) sc = wave_row_distance<F16>(base, id, queries + (size_t)q * dpadw, dpadw, metric, lane);
  if (lane == 0) out_s[w] = sc;
}

}  // namespace zvk
