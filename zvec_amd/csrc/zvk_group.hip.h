// zvk_group.hip.h — group-by search: the selection side of FlatSearcherContext::group_by_search_impl /
// FlatStreamer::group_by_search_impl / group_by_search_p_keys_impl (flat_searcher_context.h:1005-1043,
// flat_streamer.cc:391-483) and topk_to_group_result (flat_streamer_context.h:135-180).
//
// The reference keeps one bounded heap of `group_topk` documents per group id while it walks the rows, then orders the
// groups by their best score and keeps the first `group_num`.  Here the group of every storage position is DATA
// (uint32 per position; the host maps the caller's std::string ids to dense numbers once) and the distances of a batch
// are already in HBM as a candidate matrix — the dense-score rows of the flat scan, or the (query, listed position)
// scores of the p_keys path.  Four small passes over that matrix:
//   group_best   best score of every group per query (atomicMin on the order-preserving key, filtered by a plain read)
//   merge_kernel (existing, dense-row mode) the `group_num` best groups per query
//   group_fill   one wave per (query, selected group): the group's `group_topk` best candidates, sorted insertion
//   group_emit   groups in ascending order of their (refined) best score -> caller layout
// Candidate e of query q: score cs[q * stride + e], storage position ci ? ci[q * stride + e] : e (IDX_NONE = hole).
#pragma once
#include "zvk_common.hip.h"

namespace zvk {

constexpr uint32_t GROUP_EMPTY_KEY = 0xffffffffu;   // no admissible member seen (fkey of a real score is never this: NaN only)

__global__ void __launch_bounds__(256) group_best_kernel(const float *cs, const uint32_t *ci, uint32_t stride, uint32_t len,
                                                         const uint32_t *group_of, uint32_t ngroups, uint32_t *gbest) {
  const uint32_t q = blockIdx.y;
  const float *row = cs + (size_t)q * stride;
  const uint32_t *irow = ci ? ci + (size_t)q * stride : nullptr;
  uint32_t *best = gbest + (size_t)q * ngroups;
  for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < len; e += gridDim.x * 256u) {
    const float s = row[e];
    if (!(s < __builtin_inff())) continue;              // excluded / padding / hole
    const uint32_t pos = irow ? irow[e] : e;
    if (pos == IDX_NONE) continue;
    const uint32_t g = group_of[pos];
    if (g >= ngroups) continue;
    const uint32_t key = fkey(s + 0.f);
    // most candidates are not their group's best: a plain (possibly stale, hence larger) read filters them out
    if (key < best[g]) atomicMin(best + g, key);
  }
}

// keys -> scores in place (the group pick is merge_kernel's dense-row mode, which reads floats; +inf never passes)
__global__ void __launch_bounds__(256) group_keys_to_scores_kernel(uint32_t *keys, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t k = keys[i];
  reinterpret_cast<float *>(keys)[i] = (k == GROUP_EMPTY_KEY) ? __builtin_inff() : fkey_inv(k);
}

// One wave per (query, slot): slot s of query q is group sel[q][s] (s < nsel[q]).  Rows of the outputs are
// [q * gnum + s][gk]; scores ascending by (score, candidate ordinal) = the reference's heap order (first seen wins ties).
// The RNN radius is applied at the end (the reference cuts the sorted list, it does not gate the heap).
template <bool HAS_CI>
__global__ void __launch_bounds__(64) group_fill_kernel(const float *cs, const uint32_t *ci, uint32_t stride, uint32_t len,
                                                        const uint32_t *group_of, const uint32_t *sel, const uint32_t *nsel,
                                                        uint32_t gnum, uint32_t gk, float threshold, bool cut, const uint64_t *keymap,
                                                        uint64_t *out_keys, float *out_scores, uint32_t *out_idx, uint32_t *out_counts) {
  extern __shared__ f32x4 zvk_smem4[];
  float *L = reinterpret_cast<float *>(zvk_smem4);     // [gk]
  uint32_t *I = reinterpret_cast<uint32_t *>(L + gk);  // [gk] candidate ordinals
  const int lane = threadIdx.x;
  const uint32_t rowid = blockIdx.x;
  const uint32_t q = rowid / gnum, s = rowid - q * gnum;
  const float *row = cs + (size_t)q * stride;
  const uint32_t *irow = HAS_CI ? ci + (size_t)q * stride : nullptr;
  uint32_t c = 0;
  float tau = 3.402823466e+38f;                        // FLT_MAX: +inf candidates never pass
  if (s < nsel[q]) {
    const uint32_t g = sel[(size_t)q * gnum + s];
    constexpr int U = 8;
    for (uint32_t base = 0; base < len; base += 64 * U) {
      float sv[U];
      bool in[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t e = base + (uint32_t)u * 64 + lane;
        sv[u] = (e < len) ? row[e] : __builtin_inff();
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t e = base + (uint32_t)u * 64 + lane;
        in[u] = false;
        if (sv[u] <= tau) {                              // membership only for candidates that could enter
          const uint32_t pos = HAS_CI ? irow[e] : e;
          in[u] = pos != IDX_NONE && group_of[pos] == g;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint64_t m = __ballot(in[u] && sv[u] <= tau);
        while (m) {
          const int l = __builtin_ctzll(m);
          const float cand = bcast_f(sv[u], l);
          const uint32_t ce = base + (uint32_t)u * 64 + (uint32_t)l;
          m &= m - 1;
          if (sorted_insert<false>(L, nullptr, I, gk, c, cand, 0u, ce, lane, tau)) m &= __ballot(in[u] && sv[u] <= tau);
        }
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  uint32_t keep = 0;
  for (uint32_t j0 = 0; j0 < gk; j0 += 64) {
    const uint32_t j = j0 + lane;
    const size_t o = (size_t)rowid * gk + j;
    float v = __builtin_inff();
    if (j < gk) {
      if (j < c) {
        v = L[j];
        const uint32_t e = I[j];
        const uint32_t pos = HAS_CI ? irow[e] : e;
        out_scores[o] = v;
        out_idx[o] = pos;
        out_keys[o] = keymap ? keymap[pos] : (uint64_t)pos;
      } else {
        out_scores[o] = __builtin_inff();
        out_idx[o] = IDX_NONE;
        out_keys[o] = ~0ull;
      }
    }
    keep += (uint32_t)__popcll(__ballot(j < c && (!cut || v <= threshold)));
  }
  if (lane == 0) out_counts[rowid] = keep;
}

// Groups of a query in ascending order of their best score (rows hold the sorted, possibly re-scored documents; row[0]
// is the best one even when the radius cut left the group empty — the reference lists such a group with no documents).
__global__ void __launch_bounds__(64) group_emit_kernel(const uint32_t *sel, const uint32_t *nsel, uint32_t gnum, uint32_t gk,
                                                        const uint64_t *keys, const float *scores, const uint32_t *counts,
                                                        uint32_t *out_groups, uint32_t *out_ngroups, uint64_t *out_keys,
                                                        float *out_scores, uint32_t *out_counts) {
  const int lane = threadIdx.x;
  const uint32_t rowid = blockIdx.x;
  const uint32_t q = rowid / gnum, s = rowid - q * gnum;
  const uint32_t ns = nsel[q];
  if (s == 0 && lane == 0) out_ngroups[q] = ns;
  if (s >= ns) {
    // unused tail slots of the caller arrays
    const uint32_t r = s;                                // slots >= ns keep their place
    for (uint32_t j = lane; j < gk; j += 64) {
      out_keys[((size_t)q * gnum + r) * gk + j] = ~0ull;
      out_scores[((size_t)q * gnum + r) * gk + j] = __builtin_inff();
    }
    if (lane == 0) { out_groups[(size_t)q * gnum + r] = IDX_NONE; out_counts[(size_t)q * gnum + r] = 0; }
    return;
  }
  const float mine = scores[(size_t)rowid * gk];
  uint32_t rank = 0;
  for (uint32_t t = 0; t < ns; ++t) {
    const float o = scores[((size_t)q * gnum + t) * gk];
    rank += (o < mine || (o == mine && t < s)) ? 1u : 0u;
  }
  const size_t dst = ((size_t)q * gnum + rank) * gk;
  for (uint32_t j = lane; j < gk; j += 64) {
    out_keys[dst + j] = keys[(size_t)rowid * gk + j];
    out_scores[dst + j] = scores[(size_t)rowid * gk + j];
  }
  if (lane == 0) {
    out_groups[(size_t)q * gnum + rank] = sel[(size_t)q * gnum + s];
    out_counts[(size_t)q * gnum + rank] = counts[rowid];
  }
}

}  // namespace zvk
