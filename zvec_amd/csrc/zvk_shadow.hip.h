// zvk_shadow.hip.h — half-width pre-selection of an fp32 IVF index ("shadow lists"): conversion, query preparation, certificate.
// Part of the device code of libzvec_hip (included through scan_kernels.hip.h).
//
// The IVF list scan of an fp32 index is bound by the bytes of the probed lists (0.98 of what the HBM gives a streaming
// reader), so the only way to a faster search with the SAME answer is to read fewer bytes.  The shadow lists hold every stored row a
// second time, rounded to fp16, at the same positions of the same blocked layout.  A search then
//   1. scans the SHADOW lists (half the bytes) for the k' > k best rows of every query by their fp16 scores,
//   2. re-scores those k' rows against the fp32 rows with the very kernel that refines the final lists of the fp32 route
//      (wave_row_distance: for L2 the scores of both routes are the same bits), keeps the k best,
//   3. CERTIFIES the result per query: every row that was not kept has a shadow score >= t (the k'-th kept one), and a row's true
//      score cannot be better than its shadow score by more than the rounding the two conversions can cause —
//        L2:  | ‖q − b‖ − ‖q16 − b16‖ |  <=  ‖q − q16‖ + ‖b − b16‖              (triangle inequality; both terms are MEASURED:
//                                                                                 per query, and as the maximum over the stored rows)
//        IP:  | q·b − q16·b16 |          <=  ‖q‖ ‖b − b16‖ + ‖q − q16‖ ‖b16‖     (Cauchy-Schwarz)
//      plus the accumulation error of the fp32 sums (gamma_d = (d + 8) 2^-23 of the operand magnitudes: twice the textbook bound).
//      If the best possible true score of an unseen row is still worse than the k-th re-scored one, the k rows ARE the k best of
//      the probed lists — what IVFSearcher::search_impl's fp32 scan (ivf_searcher.cc:217-247) keeps.  Otherwise the query is
//      flagged and the host re-runs it on the fp32 lists (zvec_hip_ivf_shadow_certify).
// Equal true scores inside the k-th place: both routes order by (score, probe rank, position); a tie straddling the k-th place between
// a kept and an unseen row makes the certificate fail (strict inequality), so it is re-run, never guessed.
#pragma once
#include "zvk_rows.hip.h"

namespace zvk {

// device-side facts about the shadow lists, filled by shadow_rows_kernel (bit patterns of non-negative floats: atomicMax on
// the uint is the float max)
struct ShadowFacts {
  uint32_t max_err;     // max over rows of ‖b − b16‖
  uint32_t max_norm;    // max over rows of ‖b16‖
  uint32_t max_abs;     // max |element| (>= 65504 cannot be held in a half: the shadow is refused)
  uint32_t pad;
};

// one wave per stored position: fp32 row -> its fp16 twin at the same position of a blocked store with half the row words;
// squared norm of the ROUNDED row for the scan's |b|^2 term
// (the padding rows behind a list's last row are never written in the fp32 lists: they become zero rows here and feed no fact)
__global__ void __launch_bounds__(256) shadow_rows_kernel(const float *base32, uint32_t dpad32, uint32_t dscan, float *base16,
                                                          uint32_t dpad16, float *bnorm16, uint64_t n, const uint32_t *list_tile0,
                                                          const uint32_t *list_size, uint32_t nlist, uint64_t nvalid, ShadowFacts *facts) {
  const int lane = threadIdx.x & 63;
  const uint64_t pos = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pos >= n) return;
  bool valid = pos < nvalid;                         // flat store (list_tile0 == nullptr): the first nvalid positions
  if (list_tile0) {
    // IVF: the list that owns this tile — the last one whose first tile is <= it (empty lists share their successor's first tile)
    const uint32_t tile = (uint32_t)(pos >> 7);
    uint32_t lo = 0, hi = nlist;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (list_tile0[mid] <= tile) lo = mid; else hi = mid;
    }
    valid = pos - (uint64_t)list_tile0[lo] * TILE_N < (uint64_t)list_size[lo];
  }
  const uint32_t nelem = dpad16 * 2u;
  float acc = 0.f, err = 0.f, mx = 0.f;
  for (uint32_t c = lane; c < nelem; c += 64) {
    const float v = (valid && c < dscan) ? load_elem<false>(base32, pos, c, dpad32) : 0.f;
    const _Float16 h = (_Float16)v;                  // round to nearest even (HalfFloatConverter's rounding, half_float_converter.cc)
    const float hf = (float)h;
    store_elem<true>(base16, pos, c, dpad16, hf);
    acc = fmaf(hf, hf, acc);
    const float d = v - hf;                          // exact (Sterbenz) unless v overflowed the half range (caught by max_abs)
    err = fmaf(d, d, err);
    mx = fmaxf(mx, fabsf(v));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    acc += __shfl_xor(acc, off);
    err += __shfl_xor(err, off);
    mx = fmaxf(mx, __shfl_xor(mx, off));
  }
  if (lane == 0) {
    bnorm16[pos] = acc;
    // rounded UP a little: the facts feed a bound (sums of <= 2^16 squares in fp32: relative error far below 1e-3)
    atomicMax(&facts->max_err, __builtin_bit_cast(uint32_t, sqrtf(err) * 1.001f));
    atomicMax(&facts->max_norm, __builtin_bit_cast(uint32_t, sqrtf(acc) * 1.001f));
    atomicMax(&facts->max_abs, __builtin_bit_cast(uint32_t, mx));
  }
}

// fp32 queries [nq][dim_in] -> halves [nq][dpad16 words], squared norms of the ROUNDED rows, and per query
// qinfo[q] = { ‖q − q16‖, ‖q‖ } (both rounded up a little)
__global__ void __launch_bounds__(256) shadow_prep_queries_kernel(const float *src, uint32_t nq, uint32_t dim_in, uint32_t dscan,
                                                                  uint32_t dpad16, float *dst, float *qnorm16, f32x2 *qinfo) {
  const int lane = threadIdx.x & 63;
  const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= nq) return;
  const uint32_t nelem = dpad16 * 2u;
  float acc = 0.f, err = 0.f, full = 0.f;
  for (uint32_t c = lane; c < nelem; c += 64) {
    const float v = (c < dscan) ? src[(size_t)i * dim_in + c] : 0.f;
    const _Float16 h = (_Float16)v;
    const float hf = (float)h;
    reinterpret_cast<_Float16 *>(dst)[(size_t)i * nelem + c] = h;
    acc = fmaf(hf, hf, acc);
    const float d = v - hf;
    err = fmaf(d, d, err);
    full = fmaf(v, v, full);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    acc += __shfl_xor(acc, off);
    err += __shfl_xor(err, off);
    full += __shfl_xor(full, off);
  }
  if (lane == 0) {
    qnorm16[i] = acc;
    f32x2 o;
    o.x = sqrtf(err) * 1.001f;       // (inf / nan — a query element beyond the half range — makes every certificate fail: re-run in fp32)
    o.y = sqrtf(full) * 1.001f;
    qinfo[i] = o;
  }
}

// true score of every pre-selected row: one wave per (query, candidate), the summation of rescore_l2_kernel / the small-batch route
template <bool F16>
__global__ void __launch_bounds__(256) shadow_rescore_kernel(const float *base, const float *queries, uint32_t dpadw, int metric,
                                                             const uint32_t *idx, const uint32_t *counts, uint32_t nq, uint32_t kp,
                                                             float *scores) {
  const int lane = threadIdx.x & 63;
  const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (uint64_t)nq * kp) return;
  const uint32_t q = (uint32_t)(w / kp), j = (uint32_t)(w - (uint64_t)q * kp);
  if (j >= counts[q]) return;
  const float acc = wave_row_distance<F16>(base, idx[w], queries + (size_t)q * dpadw, dpadw, metric, lane);
  if (lane == 0) scores[w] = acc;
}

struct ShadowSelectArgs {
  const uint64_t *c_keys;     // [nq][kp] pre-selected rows in shadow-score order (merge_kernel's output)
  const float *c_shadow;      // [nq][kp] their shadow scores (ascending)
  const float *c_true;        // [nq][kp] their re-scored (fp32) scores
  const uint32_t *c_idx;      // [nq][kp] positions
  const uint32_t *c_counts;   // [nq]
  const f32x2 *qinfo;         // [nq] { ‖q − q16‖, ‖q‖ }
  const ShadowFacts *facts;
  uint32_t kp, k, dscan;
  int metric;
  uint64_t *out_keys;         // [nq][k]
  float *out_scores;
  uint32_t *out_idx;          // nullable
  uint32_t *out_counts;
  uint32_t *flags;            // [nq] 1 = not certified
  uint32_t *nflag;            // [1]  number of flagged queries of this search
};

// one wave per query (kp <= 64: one candidate per lane): stable rank by the true score, the k best out, certificate
__global__ void __launch_bounds__(64) shadow_select_kernel(const ShadowSelectArgs a) {
  const uint32_t q = blockIdx.x;
  const int lane = threadIdx.x;
  const uint32_t c = min(a.c_counts[q], a.kp);
  const size_t o = (size_t)q * a.kp;
  const bool have = (uint32_t)lane < c;
  const float s = have ? a.c_true[o + lane] : __builtin_inff();
  // rank = candidates that precede this one under (true score, shadow order)
  uint32_t rank = 0;
  for (uint32_t u = 0; u < c; ++u) {
    const float w = bcast_f(s, (int)u);
    rank += (w < s || (w == s && u < (uint32_t)lane)) ? 1u : 0u;
  }
  const uint32_t keep = min(c, a.k);
  if (have && rank < keep) {
    const size_t d = (size_t)q * a.k + rank;
    a.out_keys[d] = a.c_keys[o + lane];
    a.out_scores[d] = s;
    if (a.out_idx) a.out_idx[d] = a.c_idx[o + lane];
  }
  // the k-th true score (lane with rank k - 1), wave-uniform
  const unsigned long long kth = __ballot(have && rank + 1 == a.k);
  const f32x2 qi = a.qinfo[q];
  // (a query with an element beyond the half range has infinite / undefined shadow scores: never certified, not even trivially)
  bool certified = qi.x < 3.0e38f && qi.y < 3.0e38f;
  if (certified && c >= a.kp) {                       // a full pre-selection: rows may have been left out
    certified = false;
    if (kth) {
      const float sk = bcast_f(s, __builtin_ctzll(kth));
      const float t = a.c_shadow[o + a.kp - 1];       // every row left out has a shadow score >= t
      const float eb = __builtin_bit_cast(float, a.facts->max_err), bn = __builtin_bit_cast(float, a.facts->max_norm);
      // (d + 8) 2^-23: twice the textbook bound of a length-d fp32 sum — the matrix cores' internal accumulation of a k-step is not
      // documented to round to nearest at every addition
      const float gamma = (float)(a.dscan + 8) * 1.1920929e-7f;
      if (a.metric == METRIC_L2) {
        // shadow score = |q16|^2 + |b16|^2 - 2 q16.b16 in fp32: off the exact ‖q16 − b16‖^2 by <= gamma (|q16|^2 + |b16|^2)
        const float tl = t - gamma * 2.f * (qi.y * qi.y + bn * bn);
        const float lo = sqrtf(fmaxf(tl, 0.f)) * (1.f - 2e-7f) - (qi.x + eb);      // best possible true distance of an unseen row
        const float hi = sqrtf(fmaxf(sk, 0.f)) * (1.f + 2e-7f) * (1.f + gamma);    // the k-th kept distance, rounding of its sum included
        certified = lo > hi;
      } else {
        // score = -q.b (smaller is better)
        const float eps = qi.y * eb + qi.x * bn + gamma * 2.f * qi.y * bn;
        certified = (t - eps) > (sk + fabsf(sk) * 2e-7f + gamma * qi.y * bn);
      }
    }
  }
  if (lane == 0) {
    a.out_counts[q] = keep;
    a.flags[q] = certified ? 0u : 1u;
    if (!certified) atomicAdd(a.nflag, 1u);
  }
}

}  // namespace zvk
