// zvk_rows.hip.h — row movement between plain and blocked layouts, query preparation, resets, L2 refinement.
// Part of the device code of libzvec_hip (included through scan_kernels.hip.h).
#pragma once
#include "zvk_common.hip.h"

namespace zvk {

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
                                                        float *base, float *bnorm, float *extra /*cosine norm*/,
                                                        uint64_t *keys_out /*nullable*/, const uint64_t *key_src /*nullable: key = position*/) {
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
    if (keys_out) keys_out[pos] = key_src ? key_src[i] : pos;     // (appends: one launch instead of pack + fill_keys)
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

// blocked rows -> plain rows for a list of positions (fetch_vector: the vectors of a result list in one launch);
// one work-group per position
template <bool F16>
__global__ void unpack_rows_kernel(const float *base, const float *extra, const uint64_t *pos, uint32_t dscan,
                                   uint32_t dim_out, uint32_t dpadw, void *out) {
  const uint64_t p = pos[blockIdx.x];
  const size_t row_bytes = (size_t)dim_out * (F16 ? 2u : 4u);
  char *o = reinterpret_cast<char *>(out) + (size_t)blockIdx.x * row_bytes;
  for (uint32_t c = threadIdx.x; c < dim_out; c += blockDim.x) {
    if constexpr (F16) {
      if (c < dscan) {
        reinterpret_cast<_Float16 *>(o)[c] = (_Float16)load_elem<F16>(base, p, c, dpadw);
      } else {
        const uint32_t bits = extra ? __builtin_bit_cast(uint32_t, extra[p]) : 0u;
        reinterpret_cast<uint16_t *>(o)[c] = (uint16_t)(c == dscan ? bits : bits >> 16);
      }
    } else {
      reinterpret_cast<float *>(o)[c] = (c < dscan) ? load_elem<F16>(base, p, c, dpadw) : (extra ? extra[p] : 0.f);
    }
  }
}

// chunk length of a list by its level in the deal order (guided self-scheduling: the queue deals the lists largest first,
// so level 0 = the head of every search gets long chunks, level 2 = its tail short ones)
__host__ __device__ inline uint32_t level_tpc(uint32_t level, uint32_t tpc) {
  return level == 0 ? min(2u * tpc, 32u) : (level == 1 ? tpc : max(1u, tpc >> 2));
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
  if (i < nlist) list_tpc[i] = level_tpc(list_tail[i], tpc);
}

// gtau[q] = min(gtau[q], k-th score of a sample scan + slack) — only for full sample lists.
// The k-th best score of ANY subset of the rows bounds the final k-th score from above, so starting every
// work-group of the main scan at that bound drops nothing it could keep; it only spares the list warm-up.
// Slack: the sample scan and the main scan may sum the same row's dot product in different orders (different tile
// shapes), and the rounding error of a selection score scales with the operands' NORMS, not with the score
// (L2: |q|^2 + |b|^2 - 2 q.b; IP / cosine: |q||b|) — so the bound is lifted by 8e-6 of that magnitude, taken over the k
// sample rows (one of which is the row the bound must still admit), plus 1e-6 relative.  `idx`: positions of the
// sample's winners (through `pos_map` when the sample was a gathered view).
__global__ void seed_gtau_kernel(uint32_t *gtau, const float *scores, const uint32_t *idx, const uint32_t *pos_map,
                                 const uint32_t *counts, const float *qnorm, const float *bnorm, int metric, uint32_t n,
                                 uint32_t k) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || counts[i] < k) return;
  const float s = scores[(size_t)i * k + (k - 1)];
  float bmax = 0.f;
  for (uint32_t j = 0; j < k; ++j) {
    uint32_t p = idx[(size_t)i * k + j];
    if (pos_map) p = pos_map[p];
    bmax = fmaxf(bmax, bnorm[p]);
  }
  const float mag = (metric == 0) ? qnorm[i] + bmax : sqrtf(qnorm[i] * bmax);
  const float b = s + 8e-6f * mag + fabsf(s) * 1e-6f + 1e-30f;
  if (b == b) atomicMin(&gtau[i], fkey(b));
}


// Bound seeding without a selection: the scores of a prefix of the base ([query][stride], `len` valid columns) -> an upper bound
// of every query's final k-th score.  Each of the block's 256 threads keeps the minimum of its own strided share of the row; those
// are 256 disjoint sets, so the k-th smallest of the 256 minima is >= the k-th smallest of the row (and within a rank or two of
// it: the row's best k land in different shares but for the odd collision).  Same slack as seed_gtau_kernel, from the norms of the
// k candidates that carry the bound.  One block per query; k <= 256.
__global__ void __launch_bounds__(256) seed_bound_kernel(const float *dump, uint32_t stride, uint32_t len, uint32_t *gtau, const float *qnorm,
                                                         const float *bnorm, int metric, uint32_t k) {
  const uint32_t q = blockIdx.x, tid = threadIdx.x;
  const float *row = dump + (size_t)q * stride;
  float mn = __builtin_inff();
  uint32_t mi = 0;
  for (uint32_t e0 = tid; e0 < len; e0 += 256 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t e = e0 + (uint32_t)u * 256;
      v[u] = e < len ? row[e] : __builtin_inff();
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (v[u] < mn) { mn = v[u]; mi = e0 + (uint32_t)u * 256; }
  }
  __shared__ float sm[256];
  __shared__ uint32_t bmax_bits;
  sm[tid] = mn;
  if (tid == 0) bmax_bits = 0u;
  __syncthreads();
  uint32_t rank = 0;
  for (uint32_t m = 0; m < 256; ++m) {
    const float o = sm[m];
    rank += (o < mn || (o == mn && m < tid)) ? 1u : 0u;
  }
  if (rank < k && mn < __builtin_inff()) atomicMax(&bmax_bits, __builtin_bit_cast(uint32_t, fmaxf(bnorm[mi], 0.f)));      // (non-negative floats order as their bits)
  __syncthreads();
  if (rank == k - 1) {
    const float bmax = __builtin_bit_cast(float, bmax_bits);
    const float mag = (metric == 0) ? qnorm[q] + bmax : sqrtf(qnorm[q] * bmax);
    const float b = mn + 8e-6f * mag + fabsf(mn) * 1e-6f + 1e-30f;
    if (b == b) atomicMin(&gtau[q], fkey(b));
  }
}

// exclude set of a search over a store with holes: the caller's bits OR the store's hole bits
__global__ void or_bits_kernel(uint64_t *out, const uint64_t *a, const uint64_t *b, uint64_t words) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < words) out[i] = a[i] | b[i];
}

// keys of rows packed at scattered positions (streamed IVF build: a chunk's kept rows land in their lists)
__global__ void scatter_keys_kernel(uint64_t *keys, const uint64_t *dst_pos, const uint64_t *src, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keys[dst_pos[i]] = src[i];
}

// bitset widening is not needed: the API bitset is uint64 words, bit i of word i/64 == bit (i&31) of
// 32-bit word i/32 on a little-endian host/device, so the kernel reads it as uint32 words directly.

// direct distance of one stored row against one prepared query row, a whole wave per pair: lane-strided over the row's
// 16-byte chunks (4 floats / 8 halves of consecutive dimensions: one chunk of the blocked layout, see blocked_offset), so a
// 768-d fp32 row is three independent 16-byte loads per lane; per-lane partial sums, then a butterfly
template <bool F16>
__device__ __forceinline__ float wave_row_distance(const float *base, uint32_t pos, const float *qrow, uint32_t dpadw, int metric, int lane) {
  const uint32_t tile = pos >> 7, r = pos & 127, swz = (r >> 1) & 7;
  const float *trow = base + (size_t)tile * TILE_N * dpadw + (size_t)(r * 8) * 4;
  const uint32_t nchunks = dpadw >> 2;
  float acc = 0.f;
  for (uint32_t id = lane; id < nchunks; id += 64) {
    const uint32_t ks = id >> 3, c = id & 7;
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(trow + (size_t)ks * SLAB + (size_t)((c ^ swz) * 4));
    const f32x4 qv = *reinterpret_cast<const f32x4 *>(qrow + (size_t)ks * TILE_K + c * 4);
    if constexpr (F16) {
      typedef _Float16 h8 __attribute__((ext_vector_type(8)));
      const h8 bh = __builtin_bit_cast(h8, bv), qh = __builtin_bit_cast(h8, qv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = (float)qh[e], b = (float)bh[e];
        if (metric == METRIC_L2) { const float d = x - b; acc = fmaf(d, d, acc); }
        else acc = fmaf(x, b, acc);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (metric == METRIC_L2) { const float d = qv[e] - bv[e]; acc = fmaf(d, d, acc); }
        else acc = fmaf(qv[e], bv[e], acc);
      }
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
  return (metric == METRIC_L2) ? acc : (metric == METRIC_IP ? -acc : 1.f - acc);
}

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
                                                         uint32_t k, float *scores, uint32_t rows_per_query) {
  // rows_per_query > 1: several result rows belong to one query (the per-group lists of a group-by search)
  const int lane = threadIdx.x & 63;
  const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (uint64_t)nq * k) return;
  const uint32_t r = (uint32_t)(w / k), j = (uint32_t)(w - (uint64_t)r * k);
  if (j >= counts[r]) return;
  const uint32_t q = r / rows_per_query;
  // the same summation as the small-batch route's direct scores (wave_row_distance): a document's L2 score does not depend
  // on the route that found it
  const float acc = wave_row_distance<F16>(base, idx[w], queries + (size_t)q * dpadw, dpadw, METRIC_L2, lane);
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
// Query reformers on the device, bit for bit what the host side of the reference produces:
//   CosineReformer::transform (src/core/quantizer/cosine_reformer.cc:66-112): q / ||q|| followed by ||q||, with
//   ||q|| = Norm2Matrix<float,1> in its AVX-512 order (norm_matrix_fp32.i:120-157: two 16-lane accumulators over
//   32-element strides, a trailing 16-chunk and the masked tail into accumulator 0, acc0+acc1, low+high halves,
//   pairwise tree, sqrt) and Normalizer<float>::L2's element-wise division;
//   HalfFloatReformer (half_float_reformer.cc): round-to-nearest-even fp32 -> fp16.
// 16 lanes play the 16 SIMD lanes of one query (4 queries per wave); every lane runs the same sequence of
// correctly rounded fma / add / sqrt / div operations as its CPU lane, so the results are identical.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) reform_queries_kernel(const float *in, uint32_t nq, uint32_t dim, int cosine,
                                                             int out_f16, void *out) {
  const int l16 = threadIdx.x & 15;
  const uint32_t q = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (q >= nq) return;                                   // (whole 16-lane groups leave together)
  const float *m = in + (size_t)q * dim;
  float n = 0.f;
  if (cosine) {
    float s0 = 0.f, s1 = 0.f;
    const uint32_t aligned = (dim >> 5) << 5;
    uint32_t p = 0;
    for (; p != aligned; p += 32) {
      const float a = m[p + l16], b = m[p + 16 + l16];
      s0 = fmaf(a, a, s0);
      s1 = fmaf(b, b, s1);
    }
    if (dim >= aligned + 16) {
      const float a = m[p + l16];
      s0 = fmaf(a, a, s0);
      p += 16;
    }
    if (p + l16 < dim) {
      const float a = m[p + l16];
      s0 = fmaf(a, a, s0);
    }
    float v = s0 + s1;
    v = v + __shfl_xor(v, 8, 16);        // low256 + high256 (lanes 0..7 hold t[i] = v[i] + v[i+8])
    v = v + __shfl_xor(v, 1, 16);        // (t0+t1), (t2+t3), ...
    v = v + __shfl_xor(v, 2, 16);        // (t0+t1)+(t2+t3), (t4+t5)+(t6+t7)
    v = v + __shfl_xor(v, 4, 16);
    n = sqrtf(__shfl(v, 0, 16));
  }
  const uint32_t extra = cosine ? (out_f16 ? 2u : 1u) : 0u;
  for (uint32_t c = l16; c < dim; c += 16) {
    float x = m[c];
    if (cosine && n > 0.f) x = x / n;
    if (out_f16) reinterpret_cast<_Float16 *>(out)[(size_t)q * (dim + extra) + c] = (_Float16)x;
    else reinterpret_cast<float *>(out)[(size_t)q * (dim + extra) + c] = x;
  }
  if (cosine && l16 == 0) {
    if (out_f16) {
      const uint32_t bits = __builtin_bit_cast(uint32_t, n);
      uint16_t *o = reinterpret_cast<uint16_t *>(out) + (size_t)q * (dim + 2) + dim;
      o[0] = (uint16_t)bits;
      o[1] = (uint16_t)(bits >> 16);
    } else {
      reinterpret_cast<float *>(out)[(size_t)q * (dim + 1) + dim] = n;
    }
  }
}

}  // namespace zvk
