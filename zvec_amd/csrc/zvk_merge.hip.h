// zvk_merge.hip.h — per-query merge / select of partial lists, dense score rows and shard candidates.
// Part of the device code of libzvec_hip (included through scan_kernels.hip.h).
#pragma once
#include "zvk_common.hip.h"

namespace zvk {

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
  // 1: equal scores are ordered by the candidate's ORDINAL in the query's candidate stream (slot * slot_len + entry) instead
  // of (slot, index): for streams that are already in scan order while their indices are not (the small-batch IVF route:
  // positions of probed rows in probe order).  Only for part_keys == nullptr.
  uint32_t order_by_ordinal;
  const uint64_t *keymap;        // position -> key (nullable => key = position)
  uint64_t *out_keys;            // [nq][k]
  float *out_scores;             // [nq][k]
  uint32_t *out_idx;             // optional [nq][k] positions
  uint32_t *out_counts;          // [nq]
};

// Launched with 64 threads (one wave per query) or, for small batches of partial-list merges, 256: the extra waves
// only help gathering the survivors (the one phase that streams every candidate); wave 0 finishes alone.
__global__ void __launch_bounds__(256) merge_kernel(const MergeArgs a) {
  const uint32_t q = blockIdx.x;
  extern __shared__ f32x4 zvk_smem4[];
  const uint32_t k = a.k;
  float *Ls = reinterpret_cast<float *>(zvk_smem4);          // [k]
  uint32_t *Lo = reinterpret_cast<uint32_t *>(Ls + k);       // [k] order (slot)
  uint32_t *Li = Lo + k;                                      // [k] idx / candidate ordinal
  const int lane = threadIdx.x & 63;
  const uint32_t wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  uint32_t sb, nslots;
  if (a.slot_begin) { sb = a.slot_begin[q]; nslots = a.slot_begin[q + 1] - sb; }
  else if (a.slot_stride == 1) { sb = q * a.slots_per_q; nslots = a.slots_per_q; }
  else { sb = q; nslots = a.slots_per_q; }

  uint32_t cnt = 0;            // uniform
  float tau = a.threshold;     // uniform admission bound: threshold until the list is full, then its k-th score
  const uint32_t sl = a.slot_len;
  const uint64_t total = (uint64_t)nslots * sl;
  constexpr int U = 16;          // candidate batches fetched together: one wave per query is latency-bound on this stream
  __shared__ float wave_min[4][64];   // the bound passes of several waves meet here

  // Dense rows (coarse step): a cheap, exact upper bound of the k-th score before any insertion — every
  // lane takes the minimum of its own strided elements; those are 64 distinct candidates, so the k-th
  // smallest of them is >= the k-th smallest of the whole row.  Cuts the insertions to the few elements
  // at or below that bound.
  if ((a.part_i == nullptr || (nslots == 1 && a.slot_stride == 1 && a.packed_stride == 0)) && a.part_keys == nullptr &&
      a.part_counts == nullptr && k <= 64 && total >= 64) {
    float mn = __builtin_inff();
    for (uint64_t base = (uint64_t)wave * 64 * U; base < total; base += (uint64_t)nwaves * 64 * U) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t e = base + (uint64_t)u * 64 + lane;
        v[u] = (e < total) ? a.part_s[(size_t)sb * sl + e] : __builtin_inff();
      }
#pragma unroll
      for (int u = 0; u < U; ++u) mn = fminf(mn, v[u]);
    }
    if (nwaves > 1) {      // several waves (small batches): lane l's minimum over every wave's share — still 64 disjoint subsets
      wave_min[wave & 3][lane] = mn;
      __syncthreads();
      for (uint32_t w = 0; w < nwaves && w < 4; ++w) mn = fminf(mn, wave_min[w][lane]);
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
  // Contiguous partial lists ([slots][k], each ascending): the HEADS of the slots are distinct candidates, so the k-th
  // smallest of 64 lane-wise minima over the heads bounds the final k-th score from above — with hundreds of slots
  // this is far tighter than the scan's shared bound (the k-th of ONE slot), which lets through a few candidates
  // of every slot (768 slots x ~3 survivors overflowed the gather and sent single-query merges down the slow path).
  if (a.part_i != nullptr && a.part_keys == nullptr && a.part_counts == nullptr && a.packed_stride == 0 &&
      a.slot_stride == 1 && nslots >= 64 && k <= 64) {
    // (every wave takes a share, four heads in flight per lane: a plain loop was a chain of nslots / 64 dependent round trips —
    // 12 of them, most of the merge's time, for the 734 lists of a single-query IVF search)
    float mn = __builtin_inff();
    const uint32_t tstride = nwaves * 64;
    for (uint32_t j0 = wave * 64 + lane; j0 < nslots; j0 += tstride * 4) {
      float v[4];
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t j = j0 + u * tstride;
        v[u] = (j < nslots) ? a.part_s[((size_t)sb + j) * sl] : __builtin_inff();
      }
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) mn = fminf(mn, v[u]);
    }
    if (nwaves > 1) {      // lane l's minimum over every wave's share: still 64 disjoint sets of heads
      wave_min[wave & 3][lane] = mn;
      __syncthreads();
      for (uint32_t w = 0; w < nwaves && w < 4; ++w) mn = fminf(mn, wave_min[w][lane]);
    }
    uint32_t rank = 0;
    for (int m = 0; m < 64; ++m) {
      const float o = bcast_f(mn, m);
      rank += (o < mn || (o == mn && m < lane)) ? 1u : 0u;
    }
    const uint64_t hit = __ballot(rank == k - 1);
    tau = fminf(tau, bcast_f(mn, __builtin_ctzll(hit)));
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
    // three shapes of the candidate stream: 1 = one dense row of scores (element e is candidate e); 2 = contiguous
    // partial lists of one query ([slots][k], the scans' output: scores are read linearly, slot number and index only
    // for the survivors); 0 = anything else (strided / packed shard lists with per-slot counts)
    const bool linear_lists = !dense_row && a.part_i != nullptr && a.part_keys == nullptr && a.part_counts == nullptr &&
                              a.packed_stride == 0 && a.slot_stride == 1;
    auto gather = [&](auto mode_tag) {
      constexpr int MODE = decltype(mode_tag)::value;
      constexpr bool DENSE = MODE == 1;
      constexpr bool LINEAR = MODE == 2;
      for (uint32_t base = wave * 64 * U; base < tot && ns <= GATHER; base += nwaves * 64 * U) {
        float sv[U];
        uint32_t iv[U], jv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t e = base + (uint32_t)u * 64 + lane;
          bool valid = e < tot;
          if constexpr (DENSE || LINEAR) {
            sv[u] = valid ? a.part_s[(size_t)sb * sl + e] : __builtin_inff();    // (unused list entries hold +inf)
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
            jv[u] = a.order_by_ordinal ? e : j;
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
              uint32_t sj = jv[u], si = iv[u];
              if constexpr (LINEAR) {                // survivors only: which slot, which stored position
                sj = a.order_by_ordinal ? iv[u] : iv[u] / sl;
                si = a.part_i[(size_t)sb * sl + iv[u]];
              }
              surv_hi[pos] = ((unsigned long long)fkey(sv[u] + 0.f) << 32) | sj;
              surv_lo[pos] = si;
            }
            ns = first + add;
          }
        }
      }
    };
    if (dense_row) gather(std::integral_constant<int, 1>{});
    else if (linear_lists) gather(std::integral_constant<int, 2>{});
    else gather(std::integral_constant<int, 0>{});
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
      jv[u] = a.order_by_ordinal ? (uint32_t)e : j;
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

}  // namespace zvk
