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
//   group_fill   the `group_topk` best candidates of every picked group by sorted insertion: one pass per query with a
//                group -> slot table (group_fill_query), or one wave per (query, picked group) when the lists of a query
//                do not fit the LDS (group_fill)
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

// slot_tab[q][g] = slot of group g among the picked groups of query q, GROUP_NO_SLOT otherwise (pre-set by a memset 0xff)
constexpr uint16_t GROUP_NO_SLOT = 0xffffu;
__global__ void __launch_bounds__(64) group_slot_kernel(const uint32_t *sel, const uint32_t *nsel, uint32_t gnum, uint32_t ngroups,
                                                        uint16_t *slot_tab) {
  const uint32_t q = blockIdx.x;
  for (uint32_t s = threadIdx.x; s < nsel[q]; s += 64) slot_tab[(size_t)q * ngroups + sel[(size_t)q * gnum + s]] = (uint16_t)s;
}

// The same selection as group_fill_kernel in ONE pass over a query's candidates: a work-group of W waves per query, wave w
// takes every W-th run of 64 * U candidates and keeps its own sorted list per slot in LDS (candidate -> group -> slot
// through the two tables, looked up only for scores that can still enter some list); at the end wave 0 folds the
// other waves' lists into its own by sorted insertion — (score, ordinal) order, so the result is the one a single
// sequential pass gives — and writes the rows.  Used when W * gnum * gk entries fit the LDS.
template <bool HAS_CI, int W>
__global__ void __launch_bounds__(64 * W) group_fill_query_kernel(const float *cs, const uint32_t *ci, uint32_t stride, uint32_t len,
                                                                  const uint32_t *group_of, uint32_t ngroups, const uint16_t *slot_tab,
                                                                  const uint32_t *nsel, uint32_t gnum, uint32_t gk, float threshold,
                                                                  bool cut, const uint64_t *keymap, uint64_t *out_keys,
                                                                  float *out_scores, uint32_t *out_idx, uint32_t *out_counts) {
  extern __shared__ f32x4 zvk_smem4[];
  const size_t per = (size_t)gnum * gk;
  float *Lall = reinterpret_cast<float *>(zvk_smem4);                  // [W][gnum][gk]
  uint32_t *Iall = reinterpret_cast<uint32_t *>(Lall + (size_t)W * per);  // [W][gnum][gk]
  float *Tall = reinterpret_cast<float *>(Iall + (size_t)W * per);     // [W][gnum] admission bound of each list
  uint32_t *Call = reinterpret_cast<uint32_t *>(Tall + (size_t)W * gnum);  // [W][gnum] entries
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const uint32_t q = blockIdx.x;
  const uint32_t ns = nsel[q];
  const float *row = cs + (size_t)q * stride;
  const uint32_t *irow = HAS_CI ? ci + (size_t)q * stride : nullptr;
  const uint16_t *tab = slot_tab + (size_t)q * ngroups;
  float *L = Lall + (size_t)w * per, *T = Tall + (size_t)w * gnum;
  uint32_t *I = Iall + (size_t)w * per, *C = Call + (size_t)w * gnum;
  for (uint32_t s = lane; s < gnum; s += 64) { T[s] = 3.402823466e+38f; C[s] = 0; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  float tmax = ns ? 3.402823466e+38f : -__builtin_inff();   // max over the slots' bounds: nothing above it can enter any list
  auto refresh_tmax = [&]() {
    float m = -__builtin_inff();
    for (uint32_t s = lane; s < ns; s += 64) m = fmaxf(m, T[s]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    tmax = m;
  };
  constexpr int U = 8;
  for (uint32_t base = (uint32_t)w * 64 * U; base < len && ns; base += (uint32_t)W * 64 * U) {
    float sv[U];
    uint32_t sl[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t e = base + (uint32_t)u * 64 + lane;
      sv[u] = (e < len) ? row[e] : __builtin_inff();
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t e = base + (uint32_t)u * 64 + lane;
      sl[u] = GROUP_NO_SLOT;
      if (sv[u] <= tmax) {
        const uint32_t pos = HAS_CI ? irow[e] : e;
        if (pos != IDX_NONE) {
          const uint32_t g = group_of[pos];
          if (g < ngroups) sl[u] = tab[g];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint64_t m = __ballot(sl[u] != GROUP_NO_SLOT && sv[u] <= tmax);
      while (m) {
        const int l = __builtin_ctzll(m);
        m &= m - 1;
        const float cand = bcast_f(sv[u], l);
        const uint32_t s = bcast_u(sl[u], l);
        float t = T[s];
        if (!(cand <= t)) continue;
        uint32_t c = C[s];
        const float t_before = t;
        if (sorted_insert<false>(L + (size_t)s * gk, nullptr, I + (size_t)s * gk, gk, c, cand, 0u, base + (uint32_t)u * 64 + (uint32_t)l, lane, t)) {
          if (lane == 0) { T[s] = t; C[s] = c; }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_wave_barrier();
          if (t < t_before && t_before >= tmax) { refresh_tmax(); m &= __ballot(sv[u] <= tmax); }
        }
      }
    }
  }
  __syncthreads();
  if (w != 0) return;
  // fold the other waves' lists into wave 0's
  for (uint32_t s = 0; s < ns; ++s) {
    float t = Tall[s];
    uint32_t c = Call[s];
    for (int ow = 1; ow < W; ++ow) {
      const uint32_t oc = Call[(size_t)ow * gnum + s];
      const float *OL = Lall + (size_t)ow * per + (size_t)s * gk;
      const uint32_t *OI = Iall + (size_t)ow * per + (size_t)s * gk;
      for (uint32_t j = 0; j < oc; ++j) {
        const float cand = OL[j];
        if (!(cand <= t)) break;                          // ascending: the rest cannot enter either
        sorted_insert<false>(Lall + (size_t)s * gk, nullptr, Iall + (size_t)s * gk, gk, c, cand, 0u, OI[j], lane, t);
      }
    }
    if (lane == 0) Call[s] = c;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  for (uint32_t s = 0; s < gnum; ++s) {
    const uint32_t rowid = q * gnum + s;
    const uint32_t c = s < ns ? Call[s] : 0u;
    uint32_t keep = 0;
    for (uint32_t j0 = 0; j0 < gk; j0 += 64) {
      const uint32_t j = j0 + lane;
      const size_t o = (size_t)rowid * gk + j;
      float v = __builtin_inff();
      if (j < gk) {
        if (j < c) {
          v = Lall[(size_t)s * gk + j];
          const uint32_t e = Iall[(size_t)s * gk + j];
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
