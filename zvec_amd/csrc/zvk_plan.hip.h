// zvk_plan.hip.h — IVF search plan: probe rule, list-major CSR, work items; large-k expansion.
// Part of the device code of libzvec_hip (included through scan_kernels.hip.h).
#pragma once
#include "zvk_common.hip.h"

namespace zvk {

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
  // measurement hook (nullable): [0] rows of the DISTINCT probed lists, [1] sum over lists of count x rows — the
  // algorithmic bytes / flops of the list scan this plan feeds; written (not added) by plan_scan_kernel
  unsigned long long *work_stats;
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

// Small-batch route of the IVF search (a handful of queries: the list-major tile scan would run a few hundred work items
// whose k-steps are a chain of dependent HBM round trips): block (rank, query) evaluates the probe rule of plan_wave_kernel
// for its query (one lane per probe rank, prefix sums of the list sizes) and, if its rank is probed, writes the padded
// positions of that list's rows into the query's fixed-stride slice pos[q * stride ...] (excluded rows as IDX_NONE; the
// unused tail of the slice is set to IDX_NONE by the same blocks).  Block rank 0 also writes the slice bounds for pkeys_score_kernel (off[q] = q * stride)
// and the per-query statistics the tile path reports (lists probed, rows scanned).
__global__ void __launch_bounds__(256) ivf_expand_direct_kernel(const PlanArgs p, const uint32_t *list_tile0, const uint64_t *list_dense0,
                                                                const uint32_t *exclude, uint32_t stride, uint32_t *off, uint32_t *pos) {
  __shared__ uint32_t sh[4];                     // my list, my offset, probed?, rows of the whole probe set
  const int tid = threadIdx.x, lane = tid & 63;
  const uint32_t q = blockIdx.y, mine = blockIdx.x;
  const uint32_t np = min(p.coarse_cnt[q], p.nprobe);
  if (tid < 64) {
    uint32_t before_g = 0, before_l = 0, probes = 0, scanned = 0;    // uniform carries across 64-rank passes
    uint32_t my_l = 0, my_o = 0, my_p = 0;
    for (uint32_t r0 = 0; r0 < np; r0 += 64) {
      const uint32_t rnk = r0 + lane;
      const bool in = rnk < np;
      const uint32_t l = in ? probe_list(p, q, rnk) : 0;
      const uint32_t szg = in ? p.list_size_global[l] : 0;
      const uint32_t incl_g = wave_incl_scan(szg, lane);
      const bool probed = in && (before_g + incl_g - szg) < p.max_scan_count;
      const uint32_t szl = probed ? p.list_size[l] : 0;
      const uint32_t incl_l = wave_incl_scan(szl, lane);
      if (mine >= r0 && mine < r0 + 64) {
        const int ml = (int)(mine - r0);
        my_l = __shfl(l, ml);
        my_o = before_l + __shfl(incl_l - szl, ml);
        my_p = __shfl((uint32_t)probed, ml);
      }
      probes += (uint32_t)__popcll(__ballot(probed));
      scanned += __shfl(wave_incl_scan(probed ? szg : 0, lane), 63);
      before_g += __shfl(incl_g, 63);
      before_l += __shfl(incl_l, 63);
      if (before_g >= p.max_scan_count) break;   // uniform: no later rank is probed
    }
    if (lane == 0) {
      sh[0] = my_l; sh[1] = my_o; sh[2] = (mine < np) ? my_p : 0u; sh[3] = before_l;
      if (mine == 0) {
        p.q_nprobe[q] = probes;
        p.q_scanned[q] = scanned;
        off[q] = q * stride;
        if (q == p.nq - 1) off[p.nq] = p.nq * stride;
        if (p.work_stats) {                      // measurement hook: rows this launch scores (no list is shared here)
          atomicAdd(&p.work_stats[0], (unsigned long long)before_l);
          atomicAdd(&p.work_stats[1], (unsigned long long)before_l);
        }
      }
    }
  }
  __syncthreads();
  {
    // the unused tail of the query's slice holds IDX_NONE (no separate fill launch): every block of the query writes its share
    const uint32_t total = min(sh[3], stride), tail = stride - total;
    const uint32_t share = (tail + gridDim.x - 1) / gridDim.x;
    const uint32_t b = total + min(mine * share, tail), e = total + min((mine + 1) * share, tail);
    uint32_t *dst = pos + (size_t)q * stride;
    for (uint32_t j = b + tid; j < e; j += 256) dst[j] = IDX_NONE;
  }
  if (!sh[2]) return;
  const uint32_t l = sh[0], o = sh[1];
  const uint32_t sz = min(p.list_size[l], stride > o ? stride - o : 0u);   // (stride bounds the rows of any probe set)
  const uint32_t p0 = list_tile0[l] * TILE_N;
  const uint64_t d0 = list_dense0[l];
  uint32_t *dst = pos + (size_t)q * stride + o;
  for (uint32_t j = tid; j < sz; j += 256) {
    bool ex = false;
    if (exclude) { const uint64_t d = d0 + j; ex = (exclude[d >> 5] >> (d & 31)) & 1u; }
    dst[j] = ex ? IDX_NONE : p0 + j;
  }
}

// single work-group exclusive scans: slot_begin over queries, list_qoff / item_off over lists
__global__ void __launch_bounds__(1024) plan_scan_kernel(const PlanArgs p) {
  __shared__ uint32_t wtot[16];
  __shared__ uint32_t carry;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const uint32_t NT = blockDim.x, NW = NT >> 6;      // 1024 threads, or 256 when another lane's scan may be resident
  // NT elements per round: wave-level shuffle scan, the wave totals through LDS (two barriers per round)
  auto block_scan = [&](auto getv, auto putv, uint32_t n, uint32_t *total_out) {
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += NT) {
      const uint32_t i = base + tid;
      const uint32_t v = (i < n) ? getv(i) : 0;
      const uint32_t incl = wave_incl_scan(v, lane);
      if (lane == 63) wtot[wave] = incl;
      __syncthreads();
      uint32_t before = carry;
      for (int w = 0; w < wave; ++w) before += wtot[w];
      if (i < n) putv(i, before + incl - v);
      __syncthreads();
      if ((uint32_t)tid == NT - 1) carry = before + incl;
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
  if (p.work_stats) {                       // (profiling only) folded in here: no extra launches inside a timed step
    __shared__ unsigned long long red[2][16];
    unsigned long long rows = 0, pairs = 0;
    for (uint32_t l = tid; l < p.nlist; l += NT) {
      const uint32_t c = p.list_count[l];
      if (c) { rows += p.list_size[l]; pairs += (unsigned long long)c * p.list_size[l]; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { rows += __shfl_xor(rows, off); pairs += __shfl_xor(pairs, off); }
    if (lane == 0) { red[0][wave] = rows; red[1][wave] = pairs; }
    __syncthreads();
    if (tid == 0) {
      unsigned long long r = 0, q = 0;
      for (uint32_t w = 0; w < NW; ++w) { r += red[0][w]; q += red[1][w]; }
      p.work_stats[0] = r;
      p.work_stats[1] = q;
    }
  }
}

}  // namespace zvk
