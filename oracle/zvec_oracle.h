/*
 * zvec_oracle.h — CPU restatement of the zvec (Proxima) flat / IVF-Flat distance-scan hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the timed CPU baseline.  The product path (zvec_amd/, include/zvec_hip.h) never
 * links, imports or falls back to it.
 *
 * Parity status: PINNED.  The arithmetic kernels and the bounded heap are validated bit-for-bit
 * against the reference's own sources compiled by oracle/Makefile into oracle/_ref/
 * (src/ailego/math/{euclidean_distance,inner_product}_matrix_fp32.cc, norm2_matrix_fp32.cc,
 * normalizer.cc, src/include/zvec/ailego/container/heap.h) and against the known-answer values
 * of the reference's unit tests (the .json files under tests/golden/).  The scan loops (flat, IVF) are pinned by
 * the reference's structured-data tests (flat_streamer_test.cc:104-178,731-801,
 * ivf_searcher_test.cc:200-321).  The full reference FlatSearcher/IVFSearcher classes are
 * unbuildable here without stand-ins (IndexStorage::MemoryBlock needs ailego::BufferHandle, whose
 * only implementation needs Arrow/Parquet, absent) — see DESIGN.md §Oracle.
 *
 * Score conventions are the reference's "boundary B" ones (what IndexMetric kernels emit):
 *   ZO_METRIC_L2      squared Euclidean            euclidean_distance_matrix_fp32.cc:287-320
 *   ZO_METRIC_IP      MINUS inner product          inner_product_matrix_fp32.cc:870
 *   ZO_METRIC_COSINE  1 - ip over the first d dims cosine_distance_matrix.h:32-50
 *                     (rows carry d+1 floats: L2-normalised vector + its original norm)
 */
#ifndef ZVEC_ORACLE_H_
#define ZVEC_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ZO_METRIC_L2 = 0, ZO_METRIC_IP = 1, ZO_METRIC_COSINE = 2 };

/* Optional plug for the timed CPU baseline: use the reference's own SIMD 1x1 kernels
 * (from oracle/_ref) inside the restated scan loops.  NULL => restated C kernels. */
typedef float (*zo_dist_fn)(const float *m, const float *q, size_t dim);
void zo_set_distance_override(int metric, zo_dist_fn fn);

/* ---- arithmetic kernels (1 x 1) ------------------------------------------------------- */
float zo_sqeuclid_f32(const float *m, const float *q, size_t dim);
float zo_ip_f32(const float *m, const float *q, size_t dim);        /* +ip */
float zo_minus_ip_f32(const float *m, const float *q, size_t dim);  /* -ip */
float zo_cosine_f32(const float *m, const float *q, size_t dim_with_norm);
/* one-to-many cosine (IndexMetric::batch_distance of the Cosine metric): 1 - InnerProductDistanceBatch's ip */
float zo_ip_batch_f32(const float *m, const float *q, size_t dim);
float zo_ip_batch_f16(const uint16_t *m, const uint16_t *q, size_t dim);
float zo_cosine_batch_f32(const float *m, const float *q, size_t dim_with_norm);
float zo_cosine_batch_f16(const uint16_t *m, const uint16_t *q, size_t dim_with_norm);
float zo_norm2_f32(const float *m, size_t dim);
/* Normalizer<float>::L2 : arr /= ||arr|| (if > 0); *norm receives the norm. */
void zo_normalize_l2_f32(float *arr, size_t dim, float *norm);
/* CosineConverter / CosineReformer::transform for fp32: out has dim+1 floats. */
void zo_cosine_transform_f32(const float *in, size_t dim, float *out);
/* generic dispatch by metric, `dim` is the element dimension (d+1 for cosine) */
float zo_distance(int metric, const float *m, const float *q, size_t dim);

/* M x N block kernels (euclidean_distance_matrix_fp32.cc:323-929, inner_product_matrix_fp32.cc:588-1179; layouts
 * euclidean_distance_matrix.h:56-95): m = column-major block m[k*M + i] of M vectors, q = N interleaved queries
 * q[k*N + j], out[j*M + i].  For M in {8, 16, 32} — the block heights the scan paths use (FlatSearcher<32>,
 * flat_searcher_context.h) — every (i, j) is one sequential fused-multiply-add chain over k in the reference's SIMD
 * bodies (the vector lanes run across i, never across k): restated and pinned bit for bit.  The M = 2 / 4 bodies pack
 * several k into one vector and are NOT restated. */
void zo_sqeuclid_block_f32(int M, int N, const float *m, const float *q, size_t dim, float *out);
void zo_minus_ip_block_f32(int M, int N, const float *m, const float *q, size_t dim, float *out);
/* fp16 blocks (euclidean_distance_matrix_fp16.cc, inner_product_matrix_fp16.cc, distance_matrix_accum_fp16.i): halves are
 * widened to fp32 and accumulated by the same per-pair sequential FMA chain (M in {8, 16, 32}; pinned bit for bit). */
void zo_sqeuclid_block_f16(int M, int N, const uint16_t *m, const uint16_t *q, size_t dim, float *out);
void zo_minus_ip_block_f16(int M, int N, const uint16_t *m, const uint16_t *q, size_t dim, float *out);

/* fp16 rows (DT_FP16): halves as uint16_t; restated AVX-512 (no FP16 ISA) order, fp32 accumulation */
float zo_sqeuclid_f16(const uint16_t *m, const uint16_t *q, size_t dim);
float zo_ip_f16(const uint16_t *m, const uint16_t *q, size_t dim);
float zo_minus_ip_f16(const uint16_t *m, const uint16_t *q, size_t dim);
uint16_t zo_float_to_half(float f);   /* round to nearest even (HalfFloatConverter / Reformer) */
float zo_half_to_float(uint16_t h);
void zo_set_distance_override_f16(int metric, zo_dist_fn fn);

/* ---- bounded heap (ailego::Heap<IndexDocument> + IndexDocumentHeap) ------------------- */
typedef struct {
  uint64_t key;
  float score;
  uint32_t index;
} zo_doc;

typedef struct {
  zo_doc *a;
  size_t n;
  size_t limit;
  float threshold;
} zo_heap;

void zo_heap_init(zo_heap *h, zo_doc *storage, size_t limit, float threshold);
void zo_heap_emplace(zo_heap *h, uint64_t key, float score, uint32_t index);
/* ascending by score; equal scores ordered by (index) — the reference's std::sort leaves the
 * order of equal scores unspecified (heap.h:173-175). */
void zo_heap_sort(zo_heap *h);
/* test hook: run a sequence of emplace() calls and return the heap array as laid out in memory */
size_t zo_heap_replay(const float *scores, size_t n, size_t limit, float threshold,
                      uint32_t *out_index, float *out_score);

/* ---- flat scan (FlatSearcherContext::batch_search_row_{no,}filter) -------------------- */
/* exclude_bits: nullable, 1 bit per storage position, set = excluded (filter(key)==true). */
int zo_flat_search(const float *base, const uint64_t *keys, uint64_t n, uint32_t dim, int metric,
                   const float *queries, uint32_t nq, uint32_t topk, float threshold,
                   const uint64_t *exclude_bits, uint64_t *out_keys, float *out_scores,
                   uint32_t *out_index, uint32_t *out_counts);

/* ---- IVF-Flat (IVFSearcher::search_impl / search_bf_impl, IVFEntity::search) ---------- */
/* vecs/keys are in inverted-list order; list l owns positions [list_offsets[l], list_offsets[l+1]).
 * nprobe   = max(round(nlist*scan_ratio),1)            ivf_searcher_context.h:70-74
 * max_scan = max(bf_threshold, ceil(N*scan_ratio))     ivf_searcher_context.h:75-78
 * brute_force != 0 (or N <= bf_threshold) scans every list in id order (ivf_searcher.cc:188). */
int zo_ivf_search(const float *centroids, uint32_t nlist, const uint64_t *list_offsets,
                  const float *vecs, const uint64_t *keys, uint32_t dim, int metric,
                  const float *queries, uint32_t nq, uint32_t topk, float threshold,
                  uint32_t nprobe, uint32_t max_scan_count, int brute_force,
                  const uint64_t *exclude_bits, uint64_t *out_keys, float *out_scores,
                  uint32_t *out_index, uint32_t *out_counts, uint32_t *out_scanned,
                  uint32_t *out_probes /* nullable: [nq][nprobe] probed list ids, ~0u = unused */);

/* multi-threaded across queries (the reference bench's parallelism: tools/core/bench.cc:145-245) */
int zo_flat_search_mt(const float *base, const uint64_t *keys, uint64_t n, uint32_t dim,
                      int metric, const float *queries, uint32_t nq, uint32_t topk,
                      float threshold, const uint64_t *exclude_bits, uint64_t *out_keys,
                      float *out_scores, uint32_t *out_index, uint32_t *out_counts, int threads);
int zo_ivf_search_mt(const float *centroids, uint32_t nlist, const uint64_t *list_offsets,
                     const float *vecs, const uint64_t *keys, uint32_t dim, int metric,
                     const float *queries, uint32_t nq, uint32_t topk, float threshold,
                     uint32_t nprobe, uint32_t max_scan_count, int brute_force,
                     const uint64_t *exclude_bits, uint64_t *out_keys, float *out_scores,
                     uint32_t *out_index, uint32_t *out_counts, uint32_t *out_scanned,
                     int threads);

/* dtype-generic forms: dtype 0 = fp32 rows, 1 = fp16 rows (centroids, vectors and queries all of that type) */
int zo_flat_search_mt_t(int dtype, const void *base, const uint64_t *keys, uint64_t n, uint32_t dim, int metric,
                        const void *queries, uint32_t nq, uint32_t topk, float threshold,
                        const uint64_t *exclude_bits, uint64_t *out_keys, float *out_scores, uint32_t *out_index,
                        uint32_t *out_counts, int threads);
int zo_ivf_search_mt_t(int dtype, const void *centroids, uint32_t nlist, const uint64_t *list_offsets,
                       const void *vecs, const uint64_t *keys, uint32_t dim, int metric, const void *queries,
                       uint32_t nq, uint32_t topk, float threshold, uint32_t nprobe, uint32_t max_scan_count,
                       int brute_force, const uint64_t *exclude_bits, uint64_t *out_keys, float *out_scores,
                       uint32_t *out_index, uint32_t *out_counts, uint32_t *out_scanned, int threads);

/* FlatSearcherContext<32>::batch_search_column_{nofilter,filter} (flat_searcher_context.h:682-845): the column-major
 * dense path — 32-row transposed blocks x query groups of 32/16/8/4/2/1 through the M x N block kernels; `base` is
 * row-major here (transposed inside, as FlatBuilder::write_column_index does).  dtype: 0 fp32, 1 fp16.  L2 / IP. */
int zo_flat_search_column_t(int dtype, const void *base, const uint64_t *keys, uint64_t n, uint32_t dim, int metric,
                            const void *queries, uint32_t nq, uint32_t topk, float threshold,
                            const uint64_t *exclude_bits, uint64_t *out_keys, float *out_scores,
                            uint32_t *out_index, uint32_t *out_counts);

/* ---- shard merge (CombinedVectorColumnIndexer::Search, combined_vector_column_indexer.cc:172-232)
 * concat partial lists, sort by score, truncate to topk. */
int zo_merge_topk(const uint64_t *keys, const float *scores, const uint32_t *counts,
                  uint32_t nparts, uint32_t nq, uint32_t topk, uint64_t *out_keys,
                  float *out_scores, uint32_t *out_counts);

#ifdef __cplusplus
}
#endif
#endif /* ZVEC_ORACLE_H_ */
