/*
 * zvec_hip.h — C ABI of the MI355X (gfx950) flat / IVF-Flat distance-scan core for zvec.
 *
 * This is the drop-in boundary: plain C, opaque handles, plain pointers and sizes, `int` status
 * (0 = success, negative = the reference's IndexError value, src/include/zvec/core/framework/
 * index_error.h:39-62 / src/core/framework/index_error.cc:20-71).  The reference has no FFI for this
 * path (it is in-process C++); every entry point below states the reference C++ operator it stands
 * behind, so that the C++ subclasses a maintainer registers with INDEX_FACTORY_REGISTER_STREAMER /
 * _SEARCHER are one-line forwards.
 *
 * Scores are the reference's metric-kernel ("boundary B") scores, smaller = better:
 *   L2 -> squared Euclidean, IP -> MINUS inner product, COSINE -> 1 - ip on normalised rows.
 * `core_interface::Index::_dense_search` (src/core/interface/index.cc:624-649) applies
 * metric->normalize() above this boundary (IP: negate) and keeps doing so unchanged.
 *
 * Pointers named d_* are DEVICE pointers on the handle's GPU; all others are host pointers.
 * *_dev entry points are asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 * context's own stream); the host-pointer forms copy in/out and return after completion.
 */
#ifndef ZVEC_HIP_H_
#define ZVEC_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZVEC_HIP_ABI_VERSION 1

/* IndexMeta::DataType subset (src/include/zvec/core/framework/index_meta.h:27-50) */
enum { ZVEC_HIP_DT_FP32 = 0, ZVEC_HIP_DT_FP16 = 1 };

/* IndexMetric names served: "SquaredEuclidean" (src/core/metric/euclidean_metric.cc:743),
 * "InnerProduct" (inner_product_metric.cc:256), "Cosine" (cosine_metric.cc:141). */
enum { ZVEC_HIP_METRIC_L2 = 0, ZVEC_HIP_METRIC_IP = 1, ZVEC_HIP_METRIC_COSINE = 2 };

/* IndexError values used (index_error.cc:20-71) */
enum {
  ZVEC_HIP_OK = 0,
  ZVEC_HIP_ERR_RUNTIME = -1,           /* IndexError_Runtime (a HIP call failed) */
  ZVEC_HIP_ERR_UNSUPPORTED = -12,      /* IndexError_Unsupported */
  ZVEC_HIP_ERR_OUT_OF_RANGE = -17,     /* IndexError_OutOfRange */
  ZVEC_HIP_ERR_NO_MEMORY = -19,        /* IndexError_NoMemory (device or host allocation failed) */
  ZVEC_HIP_ERR_NO_READY = -21,         /* IndexError_NoReady */
  ZVEC_HIP_ERR_NO_EXIST = -22,         /* IndexError_NoExist (position / key not found) */
  ZVEC_HIP_ERR_MISMATCH = -24,         /* IndexError_Mismatch (dim / dtype / metric mismatch) */
  ZVEC_HIP_ERR_INVALID_ARGUMENT = -31, /* IndexError_InvalidArgument */
  ZVEC_HIP_ERR_NO_INDEX_LOADED = -204, /* IndexError_NoIndexLoaded */
  ZVEC_HIP_ERR_NO_TRAINED = -205       /* IndexError_NoTrained */
};

typedef struct zvec_hip_flat_s *zvec_hip_flat_t;
typedef struct zvec_hip_ivf_s *zvec_hip_ivf_t;
typedef struct zvec_hip_ctx_s *zvec_hip_ctx_t;
typedef struct zvec_hip_shards_s *zvec_hip_shards_t;
typedef struct zvec_hip_gate_s *zvec_hip_gate_t;

/* library / device ------------------------------------------------------------------------- */
int zvec_hip_abi_version(void);
int zvec_hip_device_count(int *count);
const char *zvec_hip_error_string(int code); /* IndexError::What analogue */

/* Process-wide options of the host-pointer entry points (also read from the environment at first use: ZVEC_HIP_WAIT,
 * ZVEC_HIP_ZEROCOPY).  zvec calls boundary B with ONE query per call from many threads, each with its own context
 * (src/core/interface/index.cc:24-45,605-619; tools/core/bench.cc:145-245), so how a call waits and how 3 KB travel matter:
 *   "wait"      0 = spin in hipStreamSynchronize; 1 (default) = poll a completion word in pinned memory — spinning while the
 *               answer is ~0.1 ms away, SLEEPING when the previous wait on the context was long (64 spinning callers exhaust a
 *               16-CPU quota and are frozen by the scheduler for most of every period: 19.5 k searches/s at 16 threads fell to
 *               5.2 k at 64; sleeping: 18.4 k); 2 = block on a hipEventBlockingSync event
 *   "zerocopy"  transfers up to 256 KiB may skip the copy engine through host-mapped pinned slots of the context: bit 1 (value 2,
 *               the default) = the last kernels write keys | scores | counts straight into host memory; bit 0 = the first kernel
 *               reads the queries in place (measured slower than the staged copy: off by default); 0 = staged copies both ways
 *   "assign256" 1 (default) = fp16 labelling / k-means assignment on the 256 x 256 multi-phase tile; 0 = the 128 x 128 tile
 *   "scan256"   1 (default) = flat scans of fp16 rows by at least 256 queries with k <= 11 and no filter take the 256 x 256
 *               multi-phase tile when the base is streamed (past the Infinity Cache); 2 = on small bases too (tests); 0 = the
 *               128 x 128 tile always.  Same results either way (ZVEC_HIP_ASSIGN256 / ZVEC_HIP_SCAN256 in the environment).
 * Unsupported (-12) for an unknown name, invalid argument (-1) for a value outside the option's range. */
int zvec_hip_set_option(const char *name, int value);
int zvec_hip_get_option(const char *name, int *value);

/* Page-locked host memory for callers that assemble query batches themselves (the micro-batcher of include/zvec_hip_operator.hpp
 * gathers the single queries of zvec's caller threads into such a block): a host-pointer search whose `queries` lie in it uploads
 * them in one DMA instead of the runtime's staged copy of pageable memory.  NoMemory (-19) when the allocation fails. */
int zvec_hip_host_alloc(uint64_t bytes, void **out);
int zvec_hip_host_free(void *p);

/* search context: IndexRunner::create_context() (index_runner.h:400-470).  One per caller thread;
 * owns a HIP stream and the scan workspace.  Passing NULL to a search uses the handle's built-in
 * context under a mutex. */
int zvec_hip_ctx_create(int device, zvec_hip_ctx_t *out);
int zvec_hip_ctx_destroy(zvec_hip_ctx_t ctx);
int zvec_hip_ctx_synchronize(zvec_hip_ctx_t ctx);
/* bind the context to a caller-owned hipStream_t (e.g. torch's current stream); NULL restores its own */
int zvec_hip_ctx_set_stream(zvec_hip_ctx_t ctx, void *stream);

/* Pipelining consecutive batches.  The reference overlaps queries by giving every caller thread its own context
 * (index.cc:24-45); on the GPU one search is a short sequence of small kernels around ONE bandwidth- or MFMA-bound scan.
 * Contexts that share a gate run that dominant scan kernel one after the other, in call order, while everything else of
 * their searches (query preparation, coarse pass, plan, merges, L2 refinement, the caller's candidate exchange) overlaps
 * the other context's scan: with two contexts on two streams the device runs scans back to back.  A gate is one HIP
 * event, re-recorded behind every gated scan and waited for in front of the next; contexts and gate on one device.
 * Destroy the gate after detaching (set_gate(ctx, NULL)) or destroying its contexts. */
int zvec_hip_gate_create(int device, zvec_hip_gate_t *out);
int zvec_hip_gate_destroy(zvec_hip_gate_t gate);
int zvec_hip_ctx_set_gate(zvec_hip_ctx_t ctx, zvec_hip_gate_t gate);

/* ---- flat (brute force) -------------------------------------------------------------------
 * stands behind FlatStreamer<32> / FlatSearcher<32>
 *   (src/core/algorithm/flat/flat_streamer.cc:304-389, flat_searcher.cc:162-211).
 * `dim` is the element dimension of IndexMeta (for COSINE: d+1 floats, or d+2 halves for DT_FP16 — the
 * trailing fp32 norm written by CosineConverter, cosine_converter.cc:112-134,205-212; it is not scanned). */
int zvec_hip_flat_create(uint32_t dim, int dtype, int metric, int device, zvec_hip_flat_t *out);
int zvec_hip_flat_destroy(zvec_hip_flat_t h);
int zvec_hip_flat_reserve(zvec_hip_flat_t h, uint64_t capacity);
/* IndexStreamer::add_impl / add_with_id_impl (index_runner.h:476-487), FlatBuilder::build
 * (flat_builder.cc:188-276): append n rows (dim elements each) in storage order.
 * keys == NULL -> key = storage position. */
int zvec_hip_flat_append(zvec_hip_flat_t h, const void *vecs, uint64_t n, const uint64_t *keys);
int zvec_hip_flat_append_dev(zvec_hip_flat_t h, const void *d_vecs, uint64_t n,
                             const uint64_t *d_keys, void *stream);
/* IndexStreamer::add_with_id_impl(id, vec, qmeta, ctx) in bulk (index_runner.h:483-487) — what core_interface::Index::_dense_add
 * calls for every document (src/core/interface/index.cc:505-537).  FlatStreamerEntity::add_vector_with_id semantics
 * (flat_streamer_entity.cc:900-990), row by row: the row of ids[i] lives at storage position ids[i] under key ids[i];
 * id == count appends, id > count first pads positions [count, id) with holes (kInvalidKey rows no search returns),
 * id < count overwrites in place.  keys: NULL (key = id, the reference's rule) or the key to store with each row, for
 * a caller that keeps its own id -> position map.  zvec_hip_flat_holes: hole positions currently in the store. */
int zvec_hip_flat_put(zvec_hip_flat_t h, const uint32_t *ids, uint64_t n, const void *vecs, const uint64_t *keys);
int zvec_hip_flat_holes(zvec_hip_flat_t h, uint64_t *count);
int zvec_hip_flat_count(zvec_hip_flat_t h, uint64_t *count);
/* IndexRunner::get_vector_by_id (index_runner.h:450-453): copy row `pos` (dim elements) to out. */
int zvec_hip_flat_get_vector(zvec_hip_flat_t h, uint64_t pos, void *out);
/* the same for a list of positions in one launch + one copy (IndexContext::set_fetch_vector, index_context.h:139:
 * the vectors of a result list, index.cc:635-647); out = n rows of the element type, row-major */
int zvec_hip_flat_get_vectors(zvec_hip_flat_t h, const uint64_t *positions, uint64_t n, void *out);
/* search_impl / search_bf_impl(query, qmeta, count, ctx) (index_runner.h:490-531).
 * exclude_bitset: nullable, 1 bit per storage position (bit i of word i/64), set = filter(key)
 * returned true = excluded (IndexFilter, index_filter.h:48-50).
 * threshold: IndexContext::threshold() RNN radius, FLT_MAX = none (index_document.h:250-261).
 * outputs: [count][topk] keys / scores ascending by score, out_counts[q] valid entries. */
int zvec_hip_flat_search(zvec_hip_flat_t h, zvec_hip_ctx_t ctx, const void *queries,
                         uint32_t count, uint32_t topk, float threshold,
                         const uint64_t *exclude_bitset, uint64_t *out_keys, float *out_scores,
                         uint32_t *out_counts);
int zvec_hip_flat_search_dev(zvec_hip_flat_t h, zvec_hip_ctx_t ctx, const void *d_queries,
                             uint32_t count, uint32_t topk, float threshold,
                             const uint64_t *d_exclude_bitset, uint64_t *d_out_keys,
                             float *d_out_scores, uint32_t *d_out_counts, void *stream);

/* FlatStreamer::search_bf_by_p_keys_impl (flat_streamer.cc:346-389; index_runner.h:579-585): query q is
 * compared only with the rows ids[offsets[q] .. offsets[q+1]) (storage positions; the host maps primary
 * keys to positions and drops unknown keys, as get_vector_by_key != 0 -> continue does).  Same outputs and
 * filter / threshold meaning as zvec_hip_flat_search. */
int zvec_hip_flat_search_by_ids(zvec_hip_flat_t h, zvec_hip_ctx_t ctx, const void *queries,
                                uint32_t count, const uint32_t *ids, const uint32_t *offsets,
                                uint32_t topk, float threshold, const uint64_t *exclude_bitset,
                                uint64_t *out_keys, float *out_scores, uint32_t *out_counts);

/* Group-by search: search_impl / search_bf_impl of a context with set_group_params(group_num, group_topk) and
 * set_group_by(fn) (index_context.h:129,215; FlatSearcherContext::group_by_search_impl, flat_searcher_context.h:1005-1043;
 * FlatStreamer::group_by_search_impl, flat_streamer.cc:391-437; result assembly topk_to_group_result,
 * flat_streamer_context.h:135-180).  Every group keeps its `group_topk` closest documents; the `group_num` groups whose
 * best document is closest are returned, best group first.  group_of_position[pos] (host, one entry per stored row, values
 * < ngroups; larger values = "no group", skipped) is the caller's group_by(key) mapped to dense numbers — the plugin sweeps
 * the std::function once per key, like the filter.  Outputs: out_groups[count][group_num] group numbers (out_ngroups[q]
 * valid), out_keys / out_scores [count][group_num][group_topk] ascending, out_counts[count][group_num] documents at or
 * below `threshold` (a group whose best document is beyond the radius is listed with 0 documents, as the reference does).
 * Filter bits as in zvec_hip_flat_search. */
int zvec_hip_flat_search_grouped(zvec_hip_flat_t h, zvec_hip_ctx_t ctx, const void *queries, uint32_t count,
                                 const uint32_t *group_of_position, uint32_t ngroups, uint32_t group_num, uint32_t group_topk,
                                 float threshold, const uint64_t *exclude_bitset, uint32_t *out_groups, uint32_t *out_ngroups,
                                 uint64_t *out_keys, float *out_scores, uint32_t *out_counts);
/* FlatStreamer::group_by_search_p_keys_impl (flat_streamer.cc:439-483): the grouped form of zvec_hip_flat_search_by_ids */
int zvec_hip_flat_search_grouped_by_ids(zvec_hip_flat_t h, zvec_hip_ctx_t ctx, const void *queries, uint32_t count,
                                        const uint32_t *ids, const uint32_t *offsets, const uint32_t *group_of_position,
                                        uint32_t ngroups, uint32_t group_num, uint32_t group_topk, float threshold,
                                        const uint64_t *exclude_bitset, uint32_t *out_groups, uint32_t *out_ngroups,
                                        uint64_t *out_keys, float *out_scores, uint32_t *out_counts);

/* IndexMetric::batch_distance (src/include/zvec/core/framework/index_metric.h:85-87; ailego BaseDistance::ComputeBatch,
 * src/ailego/math_batch/distance_batch.h:29-49): ONE query against n scattered stored rows (storage positions), scores
 * only, in the listed order — the one-to-many form the graph indexes drive; for L2 / IP the reference's batch form is a
 * loop of the 1x1 kernel.  Positions beyond the row count score +inf. */
int zvec_hip_flat_batch_distance(zvec_hip_flat_t h, zvec_hip_ctx_t ctx, const void *query, const uint32_t *positions,
                                 uint32_t n, float *out_scores);

/* ---- IVF-Flat -----------------------------------------------------------------------------
 * stands behind IVFStreamer / IVFSearcher (src/core/algorithm/ivf/ivf_streamer.cc:183-250,
 * ivf_searcher.cc:183-250) with IVFCentroidIndex (ivf_centroid_index.cc:273-297) and
 * IVFEntity::search (ivf_entity.cc:587-716). */
int zvec_hip_ivf_create(uint32_t dim, int dtype, int metric, int device, zvec_hip_ivf_t *out);
int zvec_hip_ivf_destroy(zvec_hip_ivf_t h);
/* load a trained index (what IVFSearcher::load reads from the ivf.* segments,
 * ivf_index_format.h:26-60,152-164): centroids [nlist][dim]; rows in inverted-list order, list l
 * owning rows [list_offsets[l], list_offsets[l+1]); keys per row (NULL -> row number). */
int zvec_hip_ivf_load(zvec_hip_ivf_t h, const void *centroids, uint32_t nlist,
                      const uint64_t *list_offsets, const void *vecs, const uint64_t *keys);
/* IVFBuilder::train + build on the GPU (ivf_builder.cc:212-403): k-means over a sample, nearest-
 * centroid labelling, list packing.  d_vecs: [n][dim] row-major DEVICE rows; keys: host, nullable.
 * The trained structure can be read back with zvec_hip_ivf_export so that a CPU reference can
 * search the very same index. */
int zvec_hip_ivf_build_dev(zvec_hip_ivf_t h, const void *d_vecs, uint64_t n, const uint64_t *keys,
                           uint32_t nlist, uint32_t kmeans_iters, uint32_t sample_per_list,
                           uint64_t seed, void *stream);
int zvec_hip_ivf_build(zvec_hip_ivf_t h, const void *vecs, uint64_t n, const uint64_t *keys,
                       uint32_t nlist, uint32_t kmeans_iters, uint32_t sample_per_list,
                       uint64_t seed);
int zvec_hip_ivf_info(zvec_hip_ivf_t h, uint64_t *count, uint32_t *nlist);
/* read back: centroids [nlist][dim], list_offsets [nlist+1], row_ids [count] = original row number
 * (build) / load-order row (load) of each list-order position; any pointer may be NULL. */
int zvec_hip_ivf_export(zvec_hip_ivf_t h, void *centroids, uint64_t *list_offsets,
                        uint64_t *row_ids);
int zvec_hip_ivf_get_vector(zvec_hip_ivf_t h, uint64_t list_pos, void *out);
int zvec_hip_ivf_get_vectors(zvec_hip_ivf_t h, const uint64_t *list_positions, uint64_t n, void *out);
/* IVFSearcher::search_impl(query, qmeta, count, ctx):
 *   nprobe         = max(round(nlist*scan_ratio),1)          ivf_searcher_context.h:70-74
 *   max_scan_count = max(bf_threshold, ceil(N*scan_ratio))    ivf_searcher_context.h:75-78
 * lists are probed in coarse-score order while the running scanned count < max_scan_count
 * (ivf_searcher.cc:217-237).  exclude_bitset is over list-order positions. */
int zvec_hip_ivf_search(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, const void *queries, uint32_t count,
                        uint32_t topk, float threshold, uint32_t nprobe, uint32_t max_scan_count,
                        const uint64_t *exclude_bitset, uint64_t *out_keys, float *out_scores,
                        uint32_t *out_counts);
int zvec_hip_ivf_search_dev(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, const void *d_queries,
                            uint32_t count, uint32_t topk, float threshold, uint32_t nprobe,
                            uint32_t max_scan_count, const uint64_t *d_exclude_bitset,
                            uint64_t *d_out_keys, float *d_out_scores, uint32_t *d_out_counts,
                            void *stream);
/* The coarse pass apart from the rest, for a sharded index.  Every shard must plan from the SAME probe sets, and the
 * pass (2 x Q x nlist x d flop against the replicated centroids, IVFCentroidIndex::search, ivf_centroid_index.cc:273-297) is the one
 * part of a shard's step that does not shrink with the number of shards.  Dealt over the shards instead: shard r runs
 * zvec_hip_ivf_coarse_dev on its slice of the batch, the probe lists — d_probe_idx [count][min(nprobe, nlist)] centroid ids in
 * coarse-score order, d_probe_cnt [count] valid entries — are exchanged (Q x nprobe x 4 bytes), and every shard runs
 * zvec_hip_ivf_search_probes_dev, which is zvec_hip_ivf_search_dev without its coarse pass (ivf_searcher.cc:217-247 from the
 * given lists; the max_scan_count rule still uses the global list sizes).  Same results as zvec_hip_ivf_search_dev. */
int zvec_hip_ivf_coarse_dev(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, const void *d_queries, uint32_t count, uint32_t nprobe,
                            uint32_t *d_probe_idx, uint32_t *d_probe_cnt, void *stream);
int zvec_hip_ivf_search_probes_dev(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, const void *d_queries, uint32_t count, uint32_t topk,
                                   float threshold, uint32_t nprobe, uint32_t max_scan_count, const uint32_t *d_probe_idx,
                                   const uint32_t *d_probe_cnt, const uint64_t *d_exclude_bitset, uint64_t *d_out_keys,
                                   float *d_out_scores, uint32_t *d_out_counts, void *stream);
/* Half-width pre-selection for an fp32 L2 / inner-product index ("shadow lists"): same results, about half the bytes per search.
 * IVFSearcher::search_impl scores every row of the probed lists in fp32 (ivf_searcher.cc:217-247, ivf_entity.cc:664-717); on the
 * GPU that scan is bound by the bytes of the lists.  zvec_hip_ivf_set_shadow(h, 1, preselect) stores every row a second time rounded
 * to fp16 (HalfFloatConverter's rounding) at the same positions; searches of more than a handful of queries (no radius, k <= 32) then
 * scan the fp16 rows for `preselect` rows per query (0: chosen by the index, from max(32, 3k): zvec_hip_ivf_shadow_width; <= 64), re-score those on the fp32 rows with the kernel that
 * refines the fp32 route's final lists, keep the k best and CERTIFY them: a row that was left out cannot beat the k-th kept one once
 * the measured rounding of the two conversions (max over the stored rows of |b - b16|, per query |q - q16|; triangle inequality for
 * L2, Cauchy-Schwarz for IP) and the accumulation error are allowed for.  Queries that fail the certificate get a second pass over the
 * fp16 rows at twice the width, and what fails that too is re-run on the fp32
 * lists: by zvec_hip_ivf_search itself (host pointers), or — device pointers, where the search call only enqueues — by
 * zvec_hip_ivf_shadow_certify, which the caller runs on the same context with the same arguments before it reads the results
 * (*rerun = queries that ended on the fp32 lists; a call with no shadow search pending returns 0 at once).  Unsupported: fp16 / cosine indexes, rows
 * beyond the half range.  enable = 0 frees the copy.  zvec_hip_ivf_shadow_info: state, bytes held, max |b - b16|, max |b16|.
 * Like zvec_hip_ivf_load, zvec_hip_ivf_set_shadow is an index-level operation: not while searches of the index are in flight.
 * Data the twin cannot serve (rows within the fp16 rounding of each other, a few rows of huge norm under inner product) would pay the
 * fp16 scan AND the fp32 re-run on every search: after four certify steps in a row that re-ran more than half of their queries the
 * next 64 searches of the index read the fp32 rows directly, then the twin is tried again (same results either way). */
/* The same for a flat store (FlatSearcher::search_impl / search_bf_impl, flat_searcher.cc:162-211: every row scored in fp32): an fp16
 * twin of the rows present at the call; searches (no radius, k <= 32, any batch size) pre-select on it, re-score in fp32, certify.  A
 * single query's scan is bound by the bytes of the rows (halved), a wide batch's by the fp32 matrix rate (fp16 instead).  ANY later
 * mutation of the store (append / put / load / reserve) drops the twin; set it again when the rows have settled.  Host-pointer
 * searches certify inside the call, zvec_hip_flat_search_dev is followed by zvec_hip_flat_shadow_certify. */
int zvec_hip_flat_set_shadow(zvec_hip_flat_t h, int enable, uint32_t preselect);
int zvec_hip_flat_shadow_info(zvec_hip_flat_t h, int *enabled, uint64_t *bytes, float *max_row_error, float *max_row_norm);
int zvec_hip_flat_shadow_width(zvec_hip_flat_t h, uint32_t topk, uint32_t *rows);
int zvec_hip_flat_shadow_certify(zvec_hip_flat_t h, zvec_hip_ctx_t ctx, const void *d_queries, uint32_t count, uint32_t topk,
                                 const uint64_t *d_exclude_bitset, uint64_t *d_out_keys, float *d_out_scores, uint32_t *d_out_counts,
                                 void *stream, uint32_t *rerun);
int zvec_hip_ivf_set_shadow(zvec_hip_ivf_t h, int enable, uint32_t preselect);
int zvec_hip_ivf_shadow_info(zvec_hip_ivf_t h, int *enabled, uint64_t *bytes, float *max_row_error, float *max_row_norm);
/* preselect = 0 leaves the width to the index: it starts at max(32, 3k); six certify steps in a row without a re-run narrow it by 8
 * (a narrower list is cheaper to keep), a step that re-ran more than 1/32 of its queries widens it by 8 and makes the width it failed
 * at the floor from then on; range max(16, 1.5k rounded up to 8) .. 64.  *rows = the width the next search with this k will use
 * (0: no twin). */
int zvec_hip_ivf_shadow_width(zvec_hip_ivf_t h, uint32_t topk, uint32_t *rows);
int zvec_hip_ivf_shadow_certify(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, const void *d_queries, uint32_t count, uint32_t topk,
                                uint32_t nprobe, uint32_t max_scan_count, const uint64_t *d_exclude_bitset, uint64_t *d_out_keys,
                                float *d_out_scores, uint32_t *d_out_counts, void *stream, uint32_t *rerun);
/* IVFSearcher::search_bf_impl: every list in list-id order (ivf_entity.cc:719-745). */
int zvec_hip_ivf_search_bf(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, const void *queries,
                           uint32_t count, uint32_t topk, float threshold,
                           const uint64_t *exclude_bitset, uint64_t *out_keys, float *out_scores,
                           uint32_t *out_counts);
/* shard: keep only the lists this shard owns (centroids stay replicated), the multi-GPU partition of
 * SURVEY §8(e) ("whole inverted lists assigned to GPUs, balanced by bytes").  The list -> shard map is the greedy
 * largest-first assignment of zvec_hip_ivf_shard_map over the GLOBAL list sizes, so every rank derives the same map
 * without communicating.  Call before load / build / begin_lists. */
int zvec_hip_ivf_keep_shard(zvec_hip_ivf_t h, uint32_t shard, uint32_t nshards);
/* the map itself (pure host arithmetic, no GPU needed): lists ordered by (128-row tiles desc, id asc), each given to the
 * shard holding the fewest tiles so far (lowest shard on ties).  owner_out[nlist]; shard_rows_out[nshards] nullable. */
int zvec_hip_ivf_shard_map(const uint32_t *list_sizes, uint32_t nlist, uint32_t nshards, uint32_t *owner_out,
                           uint64_t *shard_rows_out);
/* owner of every list of a loaded / filling index (owner_out[nlist]) */
int zvec_hip_ivf_list_owners(zvec_hip_ivf_t h, uint32_t *owner_out);

/* ---- streamed IVF build ---------------------------------------------------------------------
 * IVFBuilder's three phases as separate calls, so that a (sharded) index far larger than one chunk of raw rows can be
 * built without ever holding the corpus: train (ivf_builder.cc:212-267) on a sample; label chunks of rows with their
 * nearest centroid (ivf_builder.h:253-274, ivf_builder.cc:607-650); announce the global list sizes, then hand the chunks
 * over again with their labels — rows of lists this shard owns are appended to those lists in arrival order (the
 * dump step, ivf_builder.cc:652-729 / ivf_dumper.h:33-160).  zvec_hip_ivf_build[_dev] is exactly
 * train(strided sample) + label(all) + begin + add(all) + end. */
int zvec_hip_ivf_train_dev(zvec_hip_ivf_t h, const void *d_sample, uint64_t n_sample, uint32_t nlist,
                           uint32_t kmeans_iters, uint64_t seed, void *stream);
/* centroids trained elsewhere (e.g. broadcast from rank 0): [nlist][dim] host rows of the index element type */
int zvec_hip_ivf_set_centroids(zvec_hip_ivf_t h, const void *centroids, uint32_t nlist);
int zvec_hip_ivf_get_centroids(zvec_hip_ivf_t h, void *centroids, uint32_t *nlist);   /* either pointer nullable */
/* d_labels[i] = id of the centroid nearest to d_rows[i] under the index metric (DEVICE array, n entries);
 * returns after the labels are complete.  NoTrained (-205) before train / set_centroids / load. */
int zvec_hip_ivf_label_dev(zvec_hip_ivf_t h, const void *d_rows, uint64_t n, uint32_t *d_labels, void *stream);
/* list_sizes[nlist]: GLOBAL row count of every list (all shards).  Allocates the lists this shard owns. */
int zvec_hip_ivf_begin_lists(zvec_hip_ivf_t h, const uint32_t *list_sizes);
/* labels / keys: HOST arrays for the n rows of this call (keys NULL -> key = first_row + i); first_row = global row
 * number of d_rows[0] (what zvec_hip_ivf_export reports as row_ids).  InvalidArgument if a list overflows its
 * announced size. */
int zvec_hip_ivf_add_dev(zvec_hip_ivf_t h, const void *d_rows, uint64_t n, const uint32_t *labels,
                         const uint64_t *keys, uint64_t first_row, void *stream);
/* NoReady (-21) while any owned list is short of its announced size; the index is searchable afterwards */
int zvec_hip_ivf_end_lists(zvec_hip_ivf_t h);
/* per-query statistics of the last search on ctx (IndexContext::Stats, index_context.h:67-111):
 * scanned[q] = total_scan_count, probes[q] = lists actually probed.  Host arrays, nullable. */
int zvec_hip_ivf_last_stats(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, uint32_t count,
                            uint32_t *scanned, uint32_t *probes);

/* ---- partial top-k merge ------------------------------------------------------------------
 * CombinedVectorColumnIndexer::Search concat + sort + truncate
 * (src/db/index/column/vector_column/combined_vector_column_indexer.cc:91-232); also the merge
 * after the RCCL all-gather of per-GPU candidate lists.
 * inputs [nparts][count][topk] keys/scores + [nparts][count] counts; ties keep part order. */
int zvec_hip_merge_topk(zvec_hip_ctx_t ctx, const uint64_t *keys, const float *scores,
                        const uint32_t *counts, uint32_t nparts, uint32_t count, uint32_t topk,
                        uint64_t *out_keys, float *out_scores, uint32_t *out_counts);
int zvec_hip_merge_topk_dev(zvec_hip_ctx_t ctx, const uint64_t *d_keys, const float *d_scores,
                            const uint32_t *d_counts, uint32_t nparts, uint32_t count,
                            uint32_t topk, uint64_t *d_out_keys, float *d_out_scores,
                            uint32_t *d_out_counts, void *stream);

/* Packed form for the shard exchange: one buffer per part laid out
 *   [count*topk keys u64][count*topk scores f32][count counts u32]  padded to a multiple of 16 bytes
 * (zvec_hip_packed_bytes), parts `part_stride` bytes apart — exactly what one all-gather of the per-rank
 * buffers produces, so the candidate lists go search -> RCCL -> merge without any repacking kernel.
 * zvec_hip_packed_layout returns the three offsets inside one part. */
uint64_t zvec_hip_packed_bytes(uint32_t count, uint32_t topk);
int zvec_hip_merge_topk_packed_dev(zvec_hip_ctx_t ctx, const void *d_packed, uint64_t part_stride,
                                   uint32_t nparts, uint32_t count, uint32_t topk, uint64_t *d_out_keys,
                                   float *d_out_scores, uint32_t *d_out_counts, void *stream);

/* ---- one index over several GPUs, inside one process ---------------------------------------
 * zvec is a single-process embedded library: the plugin cannot count on a launcher to use the 8 GPUs of a node.  One
 * zvec_hip_shards_t owns G device shards of ONE index and does what CombinedVectorColumnIndexer::Search does over
 * blocks (combined_vector_column_indexer.cc:91-232): fan the batch out, rebase, concatenate in part order, keep the
 * top-k.  Partition: FLAT = contiguous row ranges (every append is cut into G pieces; key = the caller's
 * key or the global storage position), IVF = whole inverted lists dealt by zvec_hip_ivf_shard_map, centroids
 * replicated, probe sets global.  A search runs one worker thread per shard (own device, context and stream); each
 * writes its candidate lists in the packed layout and peer-copies them (xGMI) to devices[0], which merges them with
 * the kernel of zvec_hip_merge_topk_packed_dev.  `devices` may repeat a device (several shards on one GPU).
 * The one-process-per-GPU form of the same partition (RCCL all-gather of the same packed lists) is zvec_amd/dist.py. */
enum { ZVEC_HIP_SHARDS_FLAT = 0, ZVEC_HIP_SHARDS_IVF = 1 };
int zvec_hip_shards_create(uint32_t dim, int dtype, int metric, int kind, const int *devices, uint32_t ndev,
                           zvec_hip_shards_t *out);
int zvec_hip_shards_destroy(zvec_hip_shards_t h);
/* rows held in total / per shard (per_shard[ndev] nullable) */
int zvec_hip_shards_count(zvec_hip_shards_t h, uint64_t *total, uint64_t *per_shard);
/* FLAT: IndexStreamer::add_impl in bulk (index_runner.h:476-487); keys NULL -> key = global storage position */
int zvec_hip_shards_flat_append(zvec_hip_shards_t h, const void *vecs, uint64_t n, const uint64_t *keys);
/* IVF: IVFBuilder::train + build over the shards (ivf_builder.cc:212-403): k-means on devices[0], labels computed in
 * G pieces, every shard fills the lists it owns.  Same sample / seed rule as zvec_hip_ivf_build, hence the same
 * centroids and lists as an unsharded build. */
int zvec_hip_shards_ivf_build(zvec_hip_shards_t h, const void *vecs, uint64_t n, const uint64_t *keys, uint32_t nlist,
                              uint32_t kmeans_iters, uint32_t sample_per_list, uint64_t seed);
/* IVF: IVFSearcher::load of host arrays (see zvec_hip_ivf_load); every shard keeps its lists */
int zvec_hip_shards_ivf_load(zvec_hip_shards_t h, const void *centroids, uint32_t nlist, const uint64_t *list_offsets,
                             const void *vecs, const uint64_t *keys);
/* the segment-payload loaders over the shards (same arguments as zvec_hip_ivf_load_segments /
 * zvec_hip_flat_load_features): what the plugin's load() / open() call when it is configured with several devices */
int zvec_hip_shards_ivf_load_segments(zvec_hip_shards_t h, const void *inverted_header, uint64_t header_bytes,
                                      const void *inverted_meta, uint64_t meta_bytes, const void *inverted_body,
                                      uint64_t body_bytes, const void *keys, uint64_t keys_bytes, const void *centroids);
int zvec_hip_shards_flat_load_features(zvec_hip_shards_t h, const void *features, uint64_t bytes, uint64_t count,
                                       int column_major, uint32_t batch_size, const uint64_t *keys);
/* FLAT: search_bf_by_p_keys_impl and the fetch_vector gather over the shards; ids / positions are GLOBAL storage
 * positions (append order), unknown ids are skipped / NO_EXIST like the single-device entries */
int zvec_hip_shards_flat_search_by_ids(zvec_hip_shards_t h, const void *queries, uint32_t count, const uint64_t *ids,
                                       const uint32_t *offsets, uint32_t topk, float threshold, const uint64_t *exclude_bitset,
                                       uint64_t *out_keys, float *out_scores, uint32_t *out_counts);
int zvec_hip_shards_flat_get_vectors(zvec_hip_shards_t h, const uint64_t *positions, uint64_t n, void *out);
/* IVF: deal the coarse pass over the shards (off by default): shard g scores its 1/G of the batch against its replica of the
 * centroids and peer-copies that slice of the probe lists into every shard's table (xGMI, Q x (nprobe + 1) x 4 bytes in all),
 * every shard then plans from the table (zvec_hip_ivf_coarse_dev / zvec_hip_ivf_search_probes_dev).  Same results. */
int zvec_hip_shards_deal_coarse(zvec_hip_shards_t h, int enable);
/* search_impl(query, qmeta, count, ctx) over the shards; host pointers; FLAT ignores nprobe / max_scan_count.
 * exclude_bitset: 1 bit per GLOBAL storage position (FLAT: append order; IVF: list-order positions of the whole
 * index), sliced per shard on the host.  Results: the global top-k, as from one unsharded index. */
int zvec_hip_shards_search(zvec_hip_shards_t h, const void *queries, uint32_t count, uint32_t topk, float threshold,
                           uint32_t nprobe, uint32_t max_scan_count, const uint64_t *exclude_bitset,
                           uint64_t *out_keys, float *out_scores, uint32_t *out_counts);

/* ---- measurement hook ---------------------------------------------------------------------
 * Records HIP events around the dominant scan kernel of each search on ctx (on the stream the
 * kernel is launched on) and returns the accumulated launch count / milliseconds since the last
 * reset.  Used by bench.py for the roofline line. */
int zvec_hip_ctx_profile(zvec_hip_ctx_t ctx, int enable);
int zvec_hip_ctx_profile_read(zvec_hip_ctx_t ctx, uint64_t *launches, double *scan_ms,
                              double *algorithmic_bytes, double *algorithmic_flops, int reset);

/* Per-box calibration for the measurement line (bench.py `roofline.box_clock_mhz` / `box_stream_gbs`): what this box's HBM delivers
 * to a pure streaming reader (16-byte non-temporal loads over `bytes` of device memory: d_buf, or a scratch allocation when NULL),
 * best of `reps` launches, and the shader clock held under that load (s_memtime against the constant 100 MHz s_memrealtime).  So a
 * reader can tell a slow box from slow code: the list scan of SURVEY §8(d) streams the index the same way. */
int zvec_hip_calibrate(int device, const void *d_buf, uint64_t bytes, uint32_t reps, double *clock_mhz, double *stream_gbs);

/* FlatSearcher::load of the features segment of a dumped flat index (FlatBuilder<32>::write_row_index /
 * write_column_index, src/core/algorithm/flat/flat_builder.cc:186-276): `count` rows of the handle's element type —
 * row-major, or (column_major != 0) full `batch_size`-row blocks transposed in units of the element type followed by
 * a row-major remainder.  keys: the "flat.keys" payload (uint64[count]) or NULL (key = position).  Appends. */
int zvec_hip_flat_load_features(zvec_hip_flat_t h, const void *features, uint64_t bytes, uint64_t count, int column_major,
                                uint32_t batch_size, const uint64_t *keys);

/* The centroid index in a space of its own.  IVFBuilder trains INNER-PRODUCT indexes through a MipsConverter
 * (src/core/algorithm/ivf/ivf_builder.cc:552-555): the nested "ivf.centroid" index of such a file holds converted centroids — more
 * dimensions, squared-Euclidean metric, a MipsReformer named in its meta — and IVFCentroidIndex::search reforms every query before
 * the coarse scan (ivf_centroid_index.cc:273-297), while the inverted lists keep the original rows and metric.
 * zvec_hip_ivf_load_segments may then be given centroids = NULL; zvec_hip_ivf_set_coarse_space installs the converted centroid
 * rows (nlist x coarse_dim elements of the index's element type, centroid-id order; coarse_metric L2 or IP) and searches go
 * through zvec_hip_ivf_search_coarse, which takes the reformed queries ([count][coarse_dim]) beside the original ones
 * (zvec_hip_ivf_search_bf needs neither).  IVFSearcher::search_impl otherwise, ivf_searcher.cc:183-250. */
int zvec_hip_ivf_set_coarse_space(zvec_hip_ivf_t h, uint32_t coarse_dim, int coarse_metric, const void *centroids, uint32_t nlist);
int zvec_hip_ivf_search_coarse(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, const void *queries, const void *coarse_queries, uint32_t count,
                               uint32_t topk, float threshold, uint32_t nprobe, uint32_t max_scan_count,
                               const uint64_t *exclude_bitset, uint64_t *out_keys, float *out_scores, uint32_t *out_counts);

/* HipFlatStreamer::open over a storage the reference's FlatStreamer has written (FlatStreamerEntity, flat_streamer_entity.cc:43-47,
 * flat_streamer_entity.h:287-311, BlockHeader / DeletionMap flat_index_format.h:91-126): `blocks` = a run of `nblocks` blocks of
 * `block_size` bytes as they lie in a "flat.features<i>" segment — [block_vector_count x element][block_vector_count x u64 key]
 * ... [DeletionMap][BlockHeader] — and keep[b] = the live rows of block b (bit r: r < vector_count, not deleted, key valid).
 * One strided copy + one launch per run instead of a provider walk row by row.  Appends, in block / row order.
 * `bytes` = the extent of `blocks` the caller vouches for: InvalidArgument when nblocks x block_size does not fit in it or a block
 * is too small for its rows, keys and 16 tail bytes. */
int zvec_hip_flat_load_blocks(zvec_hip_flat_t h, const void *blocks, uint64_t bytes, uint64_t nblocks, uint32_t block_size,
                              uint32_t block_vector_count, const uint32_t *keep);

/* IVFSearcher::load from the raw payloads of the segments a dumped reference index holds (SURVEY next-2):
 *   inverted_header  "ivf.inverted_header": InvertedIndexHeader (ivf_index_format.h:26-37) + the serialised IndexMeta
 *                    (IndexMetaFormatHeader, src/core/framework/index_meta.cc:23-34: major order, data type, dimension)
 *   inverted_meta    "ivf.inverted_meta":   InvertedListMeta[inverted_list_count] (ivf_index_format.h:41-47)
 *   inverted_body    "ivf.inverted_body":   32-vector blocks per list as IVFDumper writes them (ivf_dumper.cc:19-81,
 *                    388-406): full blocks of a column-major index transposed, everything else row-major
 *   keys             "hc.keys":             uint64[total_vector_count] in dump (= list) order
 *   centroids        [inverted_list_count][dim] rows of the index element type (the features of the nested centroid
 *                    index "ivf.centroid", which the caller's storage layer opens)
 * The plugin obtains these blobs from zvec's own IndexStorage (segment->read), exactly as IVFSearcher::load does; the
 * body is uploaded as it is and re-laid out on the GPU.  Mismatch (-24) when data type / dimension differ from the
 * handle's, InvalidArgument for inconsistent sizes or offsets.  Parity: checked against index files dumped by
 * the reference's own IVFDumper / FlatBuilder / MemoryDumper compiled in place (tests/golden/ref_index_files.npz). */
int zvec_hip_ivf_load_segments(zvec_hip_ivf_t h, const void *inverted_header, uint64_t header_bytes,
                               const void *inverted_meta, uint64_t meta_bytes, const void *inverted_body,
                               uint64_t body_bytes, const void *keys, uint64_t keys_bytes, const void *centroids);

/* The container framing of a dumped index FILE (IndexFormat, src/include/zvec/core/framework/index_format.h:26-95;
 * IndexUnpacker::unpack, index_unpacker.h:103-330): [MetaHeader][segments' data][segment metas + ids][MetaFooter],
 * possibly chained.  Lists the segments of a whole file image — id, byte offset in the image, size — so that a caller
 * holding only the file can hand "flat.features" / "flat.keys" / "ivf.*" / "hc.keys" to the loaders above (the plugin
 * gets them from zvec's IndexStorage instead).  Header / footer / meta CRCs (crc32c) are always checked, the content CRC
 * when checksum != 0.  Host-only code.  out may be NULL with cap 0 to count; OutOfRange (-17) when cap is too small
 * (count is still set), Mismatch (-24) for a bad size field or checksum, InvalidArgument for inconsistent offsets. */
typedef struct {
  char id[64];
  uint64_t offset, size, padding;
  uint32_t crc, reserved_;
} zvec_hip_segment_t;
int zvec_hip_container_segments(const void *image, uint64_t bytes, int checksum, zvec_hip_segment_t *out, uint32_t cap,
                                uint32_t *count);

/* Query reformers on the device, for callers that keep raw fp32 query batches in HBM:
 *   cosine != 0: CosineReformer::transform (src/core/quantizer/cosine_reformer.cc:66-112) — q/||q|| followed by ||q||
 *                (out rows: dim+1 floats, or dim+2 halves with the norm's bytes in the last two slots);
 *   out_dtype == ZVEC_HIP_DT_FP16: HalfFloatReformer (round to nearest even).
 * d_in: [count][dim] fp32, d_out: device buffer of count rows of the output width.  Bit-identical to the reference's
 * AVX-512 host code (norm summation order, correctly rounded sqrt / divide). */
int zvec_hip_reform_queries_dev(zvec_hip_ctx_t ctx, const float *d_in, uint32_t count, uint32_t dim, int cosine,
                                int out_dtype, void *d_out, void *stream);

/* ---- predicate materialisation --------------------------------------------------------
 * Replaces the per-candidate IndexFilter callback (index_filter.h:48-50) whose producers are
 * DocFilter::is_filtered (src/db/sqlengine/planner/doc_filter.cc:74-87), DeleteStore::Filter
 * (src/db/index/common/delete_store.h:61-72) and InvertedSearchResult::Filter
 * (src/db/index/column/inverted_column/inverted_search_result.h:34-50):
 *     excluded(id) = deleted.contains(id) || !invert_result.contains((uint32_t)id) || !forward_bool[id]
 * evaluated once per storage position of the index (id = the key stored at that position) into the
 * 1-bit-per-position exclude bitset the search entry points take.  Each term is optional.
 *   delete_kind  ZVEC_HIP_ROARING_NONE / _32 (roaring_bitmap_portable_serialize, CRoaring 2.0.4) / _64MAP
 *                (roaring::Roaring64Map::write) / _FILE (a delete-store file image: the 64-byte BitmapMetaHeader
 *                of concurrent_roaring_bitmap.h:186-192 — magic, is_32bit, crc32c — followed by the payload)
 *   invert       32-bit portable roaring (the inverted-index result set: ids that MATCH), or NULL
 *   forward_bits Arrow boolean buffer, LSB first, forward_len bits (ids beyond it are not excluded, doc_filter.cc:104)
 * A 32-bit delete bitmap is probed with (uint32_t)id, like ConcurrentRoaringBitmap64::contains (:196-203).
 * Pointers in the descriptor are HOST memory; out_words is (count+63)/64 uint64 on the device (out_on_device != 0)
 * or on the host.  Returns InvalidArgument for a malformed stream, Mismatch (-24) for a bad magic / checksum. */
enum { ZVEC_HIP_ROARING_NONE = 0, ZVEC_HIP_ROARING_32 = 1, ZVEC_HIP_ROARING_64MAP = 2, ZVEC_HIP_ROARING_FILE = 3 };
typedef struct {
  const void *delete_bitmap;
  uint64_t delete_bytes;
  int32_t delete_kind;
  int32_t reserved_;
  const void *invert_bitmap;
  uint64_t invert_bytes;
  const uint8_t *forward_bits;
  uint64_t forward_len;
} zvec_hip_doc_filter_t;
int zvec_hip_flat_build_filter(zvec_hip_flat_t h, zvec_hip_ctx_t ctx, const zvec_hip_doc_filter_t *filter,
                               uint64_t *out_words, int out_on_device, void *stream);
/* positions are list-order positions, as for zvec_hip_ivf_search's exclude_bitset */
int zvec_hip_ivf_build_filter(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, const zvec_hip_doc_filter_t *filter,
                              uint64_t *out_words, int out_on_device, void *stream);
/* ailego::Crc32c::Hash (src/ailego/hash/crc32c.cc:626-634): raw CRC-32C update, no pre/post inversion */
uint32_t zvec_hip_crc32c(const void *data, uint64_t len, uint32_t crc);

#ifdef __cplusplus
}
#endif
#endif /* ZVEC_HIP_H_ */
