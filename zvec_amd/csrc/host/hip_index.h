// hip_index.h — C++ host side above the C ABI (include/zvec_hip.h): the reference's index-operator
// surface for the flat / IVF scan path, same names, argument meaning and error behaviour, so that
// (a) tests read like the reference's own (tests/cpp/test_host_mirror.cc), and
// (b) the real drop-in subclasses of core::IndexStreamer / core::IndexSearcher shown in INTEGRATION.md
//     are line-for-line forwards of these bodies (they only swap these mirror types for the framework's).
//
// Mirrors (reference file:line):
//   IndexMeta / IndexQueryMeta   src/include/zvec/core/framework/index_meta.h:27-50,525-580
//   IndexDocument(+List)         src/include/zvec/core/framework/index_document.h:69-218,313
//   IndexFilter                  src/include/zvec/core/framework/index_filter.h:22-71   (true = EXCLUDE)
//   IndexContext                 src/include/zvec/core/framework/index_context.h:123-262
//   IndexRunner operators        src/include/zvec/core/framework/index_runner.h:440-531
//   FlatStreamer / FlatSearcher  src/core/algorithm/flat/flat_streamer.cc:304-389, flat_searcher.cc:162-211
//   IVFSearcher (+Context)       src/core/algorithm/ivf/ivf_searcher.cc:183-250, ivf_searcher_context.h:61-79
// Header-only, no framework dependency; link with -lzvec_hip.  The operator logic itself — sweeps, probe parameters, p_keys
// mapping, result assembly, fetch_vector, add_with_id, the search sequences, the micro-batcher — is NOT here: it is
// include/zvec_hip_operator.hpp, the one copy this mirror and the real plugin (plugin/hip_plugin.cc) both instantiate.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <map>
#include <unordered_map>
#include <memory>
#include <atomic>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

#include "../../../include/zvec_hip.h"
#include "../../../include/zvec_hip_operator.hpp"

namespace zvec_hip_host {

// IndexError values (src/core/framework/index_error.cc:20-71)
enum : int {
  IndexError_Success = 0,
  IndexError_Runtime = -1,
  IndexError_Unsupported = -12,
  IndexError_NoExist = -22,
  IndexError_Mismatch = -24,
  IndexError_InvalidArgument = -31,
  IndexError_NoIndexLoaded = -204,
};

struct IndexMeta {
  enum DataType { DT_UNDEFINED = 0, DT_FP16 = 1, DT_FP32 = 2 };   // values of index_meta.h:31-41
  static uint32_t unit_size(DataType t) { return t == DT_FP16 ? 2u : 4u; }
  IndexMeta() {}
  IndexMeta(DataType t, uint32_t dim) : type_(t), dimension_(dim) {}
  void set_metric(const std::string &name, uint32_t /*revision*/ = 0) { metric_ = name; }
  const std::string &metric_name() const { return metric_; }
  DataType data_type() const { return type_; }
  uint32_t dimension() const { return dimension_; }
  uint32_t element_size() const { return dimension_ * unit_size(type_); }
  DataType type_{DT_FP32};
  uint32_t dimension_{0};
  std::string metric_{"SquaredEuclidean"};
};

struct IndexQueryMeta {
  IndexQueryMeta() {}
  IndexQueryMeta(IndexMeta::DataType t, uint32_t dim) : type_(t), dimension_(dim) {}
  uint32_t dimension() const { return dimension_; }
  uint32_t element_size() const { return dimension_ * IndexMeta::unit_size(type_); }
  IndexMeta::DataType data_type() const { return type_; }
  IndexMeta::DataType type_{IndexMeta::DT_FP32};
  uint32_t dimension_{0};
};

// string-keyed parameter bag (ailego::Params subset); keys as in ivf_params.h:25-78
class Params {
 public:
  void set(const std::string &k, double v) { kv_[k] = v; }
  bool get(const std::string &k, double *v) const {
    auto it = kv_.find(k);
    if (it == kv_.end()) return false;
    *v = it->second;
    return true;
  }
 private:
  std::map<std::string, double> kv_;
};
static const char *const PARAM_IVF_SEARCHER_SCAN_RATIO = "proxima.ivf.searcher.scan_ratio";
static const char *const PARAM_IVF_SEARCHER_BRUTE_FORCE_THRESHOLD = "proxima.ivf.searcher.brute_force_threshold";
// not a reference parameter: > 0 turns on the micro-batcher for single-query searches (see MicroBatcher below)
static const char *const PARAM_HIP_SEARCHER_BATCH_WINDOW_US = "proxima.hip.searcher.batch_window_us";
static const char *const PARAM_HIP_SEARCHER_MAX_BATCH = "proxima.hip.searcher.max_batch";
static const char *const PARAM_HIP_SEARCHER_BATCH_LINGER_US = "proxima.hip.searcher.batch_linger_us";
static const char *const PARAM_HIP_SEARCHER_HALF_WIDTH_PRESELECT = "proxima.hip.searcher.half_width_preselect";
static const char *const PARAM_HIP_SEARCHER_PRESELECT_ROWS = "proxima.hip.searcher.preselect_rows";

class IndexDocument {
 public:
  IndexDocument() {}
  IndexDocument(uint64_t k, float s) : key_(k), score_(s) {}
  uint64_t key() const { return key_; }
  float score() const { return score_; }
  // the stored row, present when the context has fetch_vector on (index_document.h:207-216 carries a MemoryBlock;
  // here the bytes are owned by the document and stay valid as long as the result list does)
  const std::vector<char> &vector() const { return vector_; }
  void set_vector(const char *p, size_t bytes) { vector_.assign(p, p + bytes); }
  bool operator<(const IndexDocument &rhs) const { return score_ < rhs.score_; }   // index_document.h:143
 private:
  uint64_t key_{0};
  float score_{0.f};
  std::vector<char> vector_;
};
using IndexDocumentList = std::vector<IndexDocument>;

// index_document.h:278-314
class GroupIndexDocument {
 public:
  const std::string &group_id() const { return group_id_; }
  const std::vector<IndexDocument> &docs() const { return docs_; }
  std::vector<IndexDocument> *mutable_docs() { return &docs_; }
  void set_group_id(const std::string &id) { group_id_ = id; }
 private:
  std::string group_id_;
  std::vector<IndexDocument> docs_;
};
using IndexGroupDocumentList = std::vector<GroupIndexDocument>;

class IndexFilter {
 public:
  template <typename T> void set(T &&fn) { fn_ = std::forward<T>(fn); }
  void reset() { fn_ = nullptr; }
  bool is_valid() const { return (bool)fn_; }
  bool operator()(uint64_t key) const { return fn_ ? fn_(key) : false; }   // true => filtered OUT
 private:
  std::function<bool(uint64_t)> fn_;
};

using zvec_hip_op::FairSharedMutex;

inline int metric_from_name(const std::string &name) {
  if (name == "SquaredEuclidean") return ZVEC_HIP_METRIC_L2;
  if (name == "InnerProduct") return ZVEC_HIP_METRIC_IP;
  if (name == "Cosine") return ZVEC_HIP_METRIC_COSINE;
  return -1;
}

// IndexContext for this path: topk, filter, RNN threshold, result lists, one HIP stream + workspace.
class Context {
 public:
  using Pointer = std::unique_ptr<Context>;
  explicit Context(int device, uint32_t magic) : magic_(magic) { rc_ = zvec_hip_ctx_create(device, &h_); }
  ~Context() { if (h_) zvec_hip_ctx_destroy(h_); }
  bool ok() const { return rc_ == 0; }
  void set_topk(uint32_t k) { topk_ = k; }
  uint32_t topk() const { return topk_; }
  void set_threshold(float v) { threshold_ = v; }
  void set_fetch_vector(bool v) { fetch_vector_ = v; }                       // index_context.h:139
  bool fetch_vector() const { return fetch_vector_; }
  float threshold() const { return threshold_; }
  template <typename T> void set_filter(T &&fn) { filter_.set(std::forward<T>(fn)); }
  void reset_filter() { filter_.reset(); has_doc_ = false; has_bits_ = false; }
  const IndexFilter &filter() const { return filter_; }
  // group-by search (index_context.h:129,209-222; flat_streamer_context.h:183-191)
  void set_group_params(uint32_t group_num, uint32_t group_topk) { group_num_ = group_num; group_topk_ = group_topk; }
  template <typename T> void set_group_by(T &&fn) { group_by_ = std::forward<T>(fn); }
  void reset_group_by() { group_by_ = nullptr; }
  bool group_by_search() const { return group_num_ > 0; }
  const IndexGroupDocumentList &group_result() const { return group_results_.at(0); }
  const IndexGroupDocumentList &group_result(size_t i) const { return group_results_.at(i); }
  // side channel of SURVEY H4: an already materialised predicate (1 bit per storage position)
  void set_exclude_bitset(std::vector<uint64_t> words) { bits_ = std::move(words); has_bits_ = true; has_doc_ = false; }
  // the composite document filter as data (doc_filter.cc:74-87): serialised roaring bitmaps + the forward bool
  // buffer; the index materialises it on the GPU (zvec_hip_*_build_filter) instead of sweeping a callback
  void set_doc_filter(const zvec_hip_doc_filter_t &f) { doc_ = f; has_doc_ = true; has_bits_ = false; }
  const IndexDocumentList &result() const { return results_.at(0); }
  const IndexDocumentList &result(size_t i) const { return results_.at(i); }
  IndexDocumentList *mutable_result(size_t i) { return &results_.at(i); }
  uint32_t magic() const { return magic_; }
  void set_magic(uint32_t m) { magic_ = m; }
  zvec_hip_ctx_t handle() const { return h_; }

  // ---- what zvec_hip_op::FlatOperator / IVFOperator read (zvec_hip_operator.hpp, "Requirements on the template arguments")
  zvec_hip_ctx_t hip() const { return h_; }
  bool op_has_filter() const { return filter_.is_valid(); }
  bool op_filtered(uint64_t key) const { return filter_(key); }
  bool op_has_group_by() const { return (bool)group_by_; }
  std::string op_group_of(uint64_t key) const { return group_by_(key); }
  uint32_t op_group_num() const { return group_num_; }
  uint32_t op_group_topk() const { return group_topk_; }
  const uint64_t *op_preset_bits() const { return has_bits_ ? bits_.data() : nullptr; }
  const zvec_hip_doc_filter_t *op_doc_filter() const { return has_doc_ ? &doc_ : nullptr; }
  zvec_hip_op::Scratch &op_scratch() { return scratch_; }
  std::vector<IndexDocumentList> &op_results() { return results_; }
  std::vector<IndexGroupDocumentList> &op_group_results() { return group_results_; }

 private:
  zvec_hip_ctx_t h_{nullptr};
  int rc_{0};
  uint32_t magic_{0};
  uint32_t topk_{0};
  float threshold_{FLT_MAX};
  IndexFilter filter_;
  std::vector<uint64_t> bits_;
  bool has_bits_{false};
  zvec_hip_doc_filter_t doc_{};
  bool has_doc_{false};
  bool fetch_vector_{false};
  std::vector<IndexDocumentList> results_{1};
  uint32_t group_num_{0}, group_topk_{0};
  std::function<std::string(uint64_t)> group_by_;
  std::vector<IndexGroupDocumentList> group_results_{1};
  zvec_hip_op::Scratch scratch_;
};

// the mirror's document types, as the shared operator logic needs them
struct MirrorDocs {
  using Document = IndexDocument;
  using DocumentList = IndexDocumentList;
  using GroupDocument = GroupIndexDocument;
  using GroupDocumentList = IndexGroupDocumentList;
  static Document make(uint64_t key, float score) { return IndexDocument(key, score); }
  static void attach(Document &d, uint32_t /*index*/, const char *row, size_t bytes) { d.set_vector(row, bytes); }   // (owned copy)
};

inline zvec_hip_op::BatcherOptions batcher_options(const Params &params) {
  zvec_hip_op::BatcherOptions bo;
  double v;
  if (params.get(PARAM_HIP_SEARCHER_BATCH_WINDOW_US, &v)) bo.window_us = (uint32_t)v;
  if (params.get(PARAM_HIP_SEARCHER_MAX_BATCH, &v)) bo.max_batch = (uint32_t)v;
  if (params.get(PARAM_HIP_SEARCHER_BATCH_LINGER_US, &v)) bo.linger_us = (uint32_t)v;
  if (params.get(PARAM_HIP_SEARCHER_HALF_WIDTH_PRESELECT, &v)) bo.shadow = (uint32_t)v;
  if (params.get(PARAM_HIP_SEARCHER_PRESELECT_ROWS, &v)) bo.shadow_preselect = (uint32_t)v;
  return bo;
}

// ---- flat: one class body serves the "FlatStreamer" and "FlatSearcher" registrations -------------
class HipFlatStreamer {
 public:
  ~HipFlatStreamer() { close(); }
  int init(const IndexMeta &meta, const Params &params) {
    meta_ = meta;
    metric_ = metric_from_name(meta.metric_name());
    bo_ = batcher_options(params);
    if (metric_ < 0 || (meta.data_type() != IndexMeta::DT_FP32 && meta.data_type() != IndexMeta::DT_FP16)) return IndexError_Unsupported;
    return 0;
  }
  int open(int device = 0) {
    device_ = device;
    static std::atomic<uint32_t> next_magic{0x48495031u};    // IndexContext::GenerateMagic (index_context.cc:22-25)
    magic_ = next_magic.fetch_add(1);
    // add_with_id keeps the reference's "storage position == id" rule here (FlatStreamerEntity::add_vector_with_id)
    return op_.create(meta_.dimension(), meta_.data_type() == IndexMeta::DT_FP16 ? ZVEC_HIP_DT_FP16 : ZVEC_HIP_DT_FP32, metric_,
                      meta_.element_size(), device, 1, /*position_is_id=*/true, bo_);
  }
  int close() { op_.destroy(); return 0; }
  const IndexMeta &meta() const { return meta_; }
  uint32_t magic() const { return magic_; }
  Context::Pointer create_context() const {
    if (!op_.ready()) return nullptr;
    Context::Pointer c(new Context(device_, magic_));
    return c->ok() ? std::move(c) : nullptr;
  }
  //! Add a vector into index (index_runner.h:476-480)
  int add_impl(uint64_t key, const void *query, const IndexQueryMeta &qmeta, Context::Pointer & /*context*/) {
    if (!op_.ready() || !query || qmeta.element_size() != meta_.element_size()) return IndexError_InvalidArgument;
    return op_.append(query, 1, &key);
  }
  //! Add a vector with id into index (index_runner.h:483-487) — the call core_interface::Index::_dense_add makes
  int add_with_id_impl(uint32_t id, const void *query, const IndexQueryMeta &qmeta, Context::Pointer & /*context*/) {
    if (!op_.ready() || !query || qmeta.element_size() != meta_.element_size()) return IndexError_InvalidArgument;
    return op_.put(id, query);
  }
  static constexpr uint64_t kInvalidKey = zvec_hip_op::kInvalidKey;              // flat_index_format.h:29
  //! bulk form used by FlatBuilder::build / FlatSearcher::load (flat_builder.cc:188-276)
  int add_batch(const void *vecs, uint64_t n, const uint64_t *keys) {
    if (!op_.ready()) return IndexError_InvalidArgument;
    return op_.append(vecs, n, keys);
  }
  //! Similarity search (index_runner.h:490-500)
  int search_impl(const void *query, const IndexQueryMeta &qmeta, Context::Pointer &context) const {
    return search_impl(query, qmeta, 1, context);
  }
  int search_impl(const void *query, const IndexQueryMeta &qmeta, uint32_t count, Context::Pointer &context) const {
    Context *ctx = bind(query, qmeta, context);
    return ctx ? op_.search(query, count, ctx) : (int)IndexError_InvalidArgument;
  }
  //! Similarity brute force search (index_runner.h:520-531): the flat scan is the brute force
  int search_bf_impl(const void *query, const IndexQueryMeta &qmeta, Context::Pointer &context) const {
    return search_impl(query, qmeta, 1, context);
  }
  int search_bf_impl(const void *query, const IndexQueryMeta &qmeta, uint32_t count, Context::Pointer &context) const {
    return search_impl(query, qmeta, count, context);
  }
  //! Linear search by primary keys (index_runner.h:579-585; flat_streamer.cc:346-389): unknown keys are skipped
  int search_bf_by_p_keys_impl(const void *query, const std::vector<std::vector<uint64_t>> &p_keys, const IndexQueryMeta &qmeta,
                               uint32_t count, Context::Pointer &context) const {
    Context *ctx = bind(query, qmeta, context);
    return ctx ? op_.search_by_keys(query, p_keys, count, ctx) : (int)IndexError_InvalidArgument;
  }
  //! Fetch vector by id (index_runner.h:445-453)
  int get_vector_by_id(uint32_t id, std::vector<float> *out) const {
    if (!op_.ready()) return IndexError_InvalidArgument;
    out->resize(meta_.dimension());
    return op_.vector_of_pos(id, out->data());
  }
  uint64_t count() const { uint64_t n = 0; if (op_.handle()) zvec_hip_flat_count(op_.handle(), &n); return n; }

 private:
  //! flat_searcher.cc:194-198 "Invalid context or topk not set yet"; a context made by another index is re-bound
  Context *bind(const void *query, const IndexQueryMeta &qmeta, Context::Pointer &context) const {
    if (!op_.ready() || !query || qmeta.element_size() != meta_.element_size()) return nullptr;
    Context *ctx = context.get();
    if (!ctx || (ctx->topk() == 0 && !ctx->group_by_search())) return nullptr;
    if (ctx->magic() != magic_) ctx->set_magic(magic_);
    return ctx;
  }
  IndexMeta meta_;
  int metric_{0};
  int device_{0};
  uint32_t magic_{0};
  zvec_hip_op::BatcherOptions bo_;
  zvec_hip_op::FlatOperator<Context, MirrorDocs> op_;
};
using HipFlatSearcher = HipFlatStreamer;

// ---- IVF -----------------------------------------------------------------------------------------
class HipIVFSearcher {
 public:
  ~HipIVFSearcher() { unload(); }
  int init(const Params &params) {
    double v;
    if (params.get(PARAM_IVF_SEARCHER_SCAN_RATIO, &v)) scan_ratio_ = (float)v;
    if (params.get(PARAM_IVF_SEARCHER_BRUTE_FORCE_THRESHOLD, &v)) bruteforce_threshold_ = (uint32_t)v;
    bo_ = batcher_options(params);
    if (scan_ratio_ <= 0.0f) return IndexError_InvalidArgument;   // ivf_searcher_context.h:65-69
    return 0;
  }
  //! what IVFSearcher::load reads from the ivf.* segments (ivf_index_format.h:26-60,152-164)
  // centroids / vecs: rows of meta.data_type() elements (fp32 or fp16)
  int load(const IndexMeta &meta, const void *centroids, uint32_t nlist, const uint64_t *list_offsets,
           const void *vecs, const uint64_t *keys, int device = 0) {
    meta_ = meta;
    int metric = metric_from_name(meta.metric_name());
    if (metric < 0 || (meta.data_type() != IndexMeta::DT_FP32 && meta.data_type() != IndexMeta::DT_FP16)) return IndexError_Unsupported;
    device_ = device;
    static std::atomic<uint32_t> next_magic{0x49564631u};
    magic_ = next_magic.fetch_add(1);
    int rc = op_.create(meta.dimension(), meta.data_type() == IndexMeta::DT_FP16 ? ZVEC_HIP_DT_FP16 : ZVEC_HIP_DT_FP32, metric,
                        meta.element_size(), device, 1, bo_);
    if (rc != 0) return rc;
    return op_.load_arrays(centroids, nlist, list_offsets, vecs, keys);
  }
  int unload() { op_.destroy(); return 0; }
  Context::Pointer create_context() const {
    if (!op_.ready()) return nullptr;                          // "Load the index first" ivf_searcher.cc:257-260
    Context::Pointer c(new Context(device_, magic_));
    return c->ok() ? std::move(c) : nullptr;
  }
  // IVFSearcherContext::update (ivf_searcher_context.h:61-79)
  uint32_t nprobe() const { return zvec_hip_op::probe_params(op_.nlist(), op_.count(), scan_ratio_, bruteforce_threshold_).nprobe; }
  uint32_t max_scan_count() const { return zvec_hip_op::probe_params(op_.nlist(), op_.count(), scan_ratio_, bruteforce_threshold_).max_scan; }
  int search_impl(const void *query, const IndexQueryMeta &qmeta, Context::Pointer &context) const {
    return search_impl(query, qmeta, 1, context);
  }
  int search_impl(const void *query, const IndexQueryMeta &qmeta, uint32_t count, Context::Pointer &context) const {
    return run(query, qmeta, count, context, false);
  }
  int search_bf_impl(const void *query, const IndexQueryMeta &qmeta, Context::Pointer &context) const {
    return run(query, qmeta, 1, context, true);
  }
  int search_bf_impl(const void *query, const IndexQueryMeta &qmeta, uint32_t count, Context::Pointer &context) const {
    return run(query, qmeta, count, context, true);
  }
  zvec_hip_op::MicroBatcher<MirrorDocs>::Stats batcher_stats() const { return op_.batcher_stats(); }

 private:
  int run(const void *query, const IndexQueryMeta &qmeta, uint32_t count, Context::Pointer &context, bool bf) const {
    if (!op_.ready()) return IndexError_NoIndexLoaded;
    if (!query || qmeta.element_size() != meta_.element_size()) return IndexError_InvalidArgument;   // ivf_searcher.cc:191-194
    Context *ctx = context.get();
    if (!ctx || ctx->topk() == 0) return IndexError_InvalidArgument;                                 // ivf_searcher.cc:197-200
    if (ctx->magic() != magic_) ctx->set_magic(magic_);
    return op_.search(query, count, ctx, bf, scan_ratio_, bruteforce_threshold_);
  }
  zvec_hip_op::BatcherOptions bo_;
  IndexMeta meta_;
  int device_{0};
  uint32_t magic_{0};
  float scan_ratio_{0.1f};                 // kDefaultScanRatio  ivf_searcher_context.h:211
  uint32_t bruteforce_threshold_{1000u};   // kDefaultBfThreshold ivf_searcher_context.h:212
  zvec_hip_op::IVFOperator<Context, MirrorDocs> op_;
};

// "IVFStreamer" is what the product instantiates (indexes/ivf_index.cc:38-39); in the reference it is the same read-only
// operator over a dumped index as the searcher (ivf_streamer.h:28-85: open / search / unload, no add_impl)
using HipIVFStreamer = HipIVFSearcher;

}  // namespace zvec_hip_host
