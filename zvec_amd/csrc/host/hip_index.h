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
// Header-only, no framework dependency; link with -lzvec_hip.
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

// reader/writer lock that cannot starve the writer: std::shared_mutex on glibc prefers readers, and searches that
// overlap continuously (several threads in a loop) would keep an add_impl waiting forever.  Everybody passes a gate
// first; a writer keeps the gate while the readers in flight drain, so new readers queue behind it.
class FairSharedMutex {
 public:
  void lock() { gate_.lock(); rw_.lock(); gate_.unlock(); }
  void unlock() { rw_.unlock(); }
  void lock_shared() { gate_.lock(); rw_.lock_shared(); gate_.unlock(); }
  void unlock_shared() { rw_.unlock_shared(); }
 private:
  std::mutex gate_;
  std::shared_mutex rw_;
};

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
  bool group_by_valid() const { return (bool)group_by_; }
  uint32_t group_num() const { return group_num_; }
  uint32_t group_topk() const { return group_topk_; }
  const IndexGroupDocumentList &group_result() const { return group_results_.at(0); }
  const IndexGroupDocumentList &group_result(size_t i) const { return group_results_.at(i); }
  std::vector<IndexGroupDocumentList> &mutable_group_results() { return group_results_; }
  // host sweep of the group-by callback: dense group number of every storage position + the ids they stand for
  void sweep_groups(const std::vector<uint64_t> &keys_by_position, std::vector<uint32_t> *group_of, std::vector<std::string> *ids) const {
    std::unordered_map<std::string, uint32_t> number_of;
    ids->clear();
    group_of->resize(keys_by_position.size());
    for (size_t i = 0; i < keys_by_position.size(); ++i) {
      std::string id = group_by_(keys_by_position[i]);
      auto it = number_of.find(id);
      if (it == number_of.end()) {
        it = number_of.emplace(id, (uint32_t)ids->size()).first;
        ids->push_back(std::move(id));
      }
      (*group_of)[i] = it->second;
    }
  }
  // side channel of SURVEY H4: an already materialised predicate (1 bit per storage position)
  void set_exclude_bitset(std::vector<uint64_t> words) { bits_ = std::move(words); has_bits_ = true; has_doc_ = false; }
  // the composite document filter as data (doc_filter.cc:74-87): serialised roaring bitmaps + the forward bool
  // buffer; the index materialises it on the GPU (zvec_hip_*_build_filter) instead of sweeping a callback
  void set_doc_filter(const zvec_hip_doc_filter_t &f) { doc_ = f; has_doc_ = true; has_bits_ = false; }
  bool has_doc_filter() const { return has_doc_; }
  bool has_any_filter() const { return has_doc_ || has_bits_ || filter_.is_valid(); }
  void take_single(IndexDocumentList &&list) { results_.assign(1, IndexDocumentList()); results_[0] = std::move(list); }
  const zvec_hip_doc_filter_t &doc_filter() const { return doc_; }
  std::vector<uint64_t> &bits() { return bits_; }
  const IndexDocumentList &result() const { return results_.at(0); }
  const IndexDocumentList &result(size_t i) const { return results_.at(i); }
  IndexDocumentList *mutable_result(size_t i) { return &results_.at(i); }
  uint32_t magic() const { return magic_; }
  void set_magic(uint32_t m) { magic_ = m; }
  zvec_hip_ctx_t handle() const { return h_; }

  // host sweep of the callback over the keys of the storage positions -> bitset
  const uint64_t *materialise(const std::vector<uint64_t> &keys_by_position) {
    if (has_bits_) return bits_.data();
    if (!filter_.is_valid()) return nullptr;
    bits_.assign((keys_by_position.size() + 63) / 64, 0);
    for (size_t i = 0; i < keys_by_position.size(); ++i)
      if (keys_by_position[i] != ~0ull && filter_(keys_by_position[i])) bits_[i >> 6] |= (1ull << (i & 63));   // (holes carry kInvalidKey)
    return bits_.data();
  }
  void take(uint32_t count, uint32_t topk, const std::vector<uint64_t> &keys, const std::vector<float> &scores,
            const std::vector<uint32_t> &counts) {
    results_.assign(count, IndexDocumentList());
    for (uint32_t q = 0; q < count; ++q) {
      results_[q].reserve(counts[q]);
      for (uint32_t j = 0; j < counts[q]; ++j) results_[q].emplace_back(keys[(size_t)q * topk + j], scores[(size_t)q * topk + j]);
    }
  }
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
};

// fetch_vector: one gather of the stored rows of every result document (key -> position through `pos_of_key`,
// built lazily from the keys the index holds), then the rows are attached to the documents
template <typename GetRows>
inline int attach_result_vectors(Context *ctx, uint32_t count, size_t row_bytes, const std::vector<uint64_t> &keys_by_pos,
                                 std::unordered_map<uint64_t, uint64_t> *pos_of_key, std::mutex *map_mu, GetRows &&get_rows) {
  std::vector<uint64_t> pos;
  {
    // searches are const and run concurrently (index_runner.h:490-531): the lazily built map is shared mutable state
    std::lock_guard<std::mutex> g(*map_mu);
    if (pos_of_key->size() != keys_by_pos.size()) {
      pos_of_key->clear();
      pos_of_key->reserve(keys_by_pos.size());
      for (uint64_t i = 0; i < keys_by_pos.size(); ++i) pos_of_key->emplace(keys_by_pos[i], i);
    }
    for (uint32_t q = 0; q < count; ++q)
      for (const auto &d : ctx->result(q)) {
        auto it = pos_of_key->find(d.key());
        if (it == pos_of_key->end()) return IndexError_NoExist;
        pos.push_back(it->second);
      }
  }
  if (pos.empty()) return 0;
  std::vector<char> rows(pos.size() * row_bytes);
  int rc = get_rows(pos.data(), pos.size(), rows.data());
  if (rc != 0) return rc;
  size_t o = 0;
  for (uint32_t q = 0; q < count; ++q)
    for (auto &d : *ctx->mutable_result(q)) d.set_vector(rows.data() + (o++) * row_bytes, row_bytes);
  return 0;
}

// ---- micro-batcher -------------------------------------------------------------------------------
// The product drives boundary B with ONE query per call from many threads (index.cc:617), which leaves the GPU at a
// few thousand searches per second, while one batched call answers 1024 queries in 5 ms.  The batcher turns the
// former into the latter behind the same single-query entry point: concurrent callers with the same topk join an open
// batch; the first one in becomes its leader, keeps the batch open while an earlier batch is still searching (at most
// until it is full or `window_us` has passed), runs ONE batched search and hands every caller its own result list.
// A lone caller on an idle index is not delayed at all; under load the batches grow by themselves.  Callers with a filter / threshold / fetch_vector bypass it (their searches are not interchangeable).
class MicroBatcher {
 public:
  // runs a batched search of `count` queries (row-major, row_bytes each) and fills keys/scores [count][topk] + counts
  using RunFn = std::function<int(const void *queries, uint32_t count, uint32_t topk, std::vector<uint64_t> *keys,
                                  std::vector<float> *scores, std::vector<uint32_t> *counts)>;
  // linger_us: how long a leader keeps its batch open even when nothing else is searching (0 = a lone caller is never
  // delayed; a few tens of microseconds let callers that arrive in a burst share the first batch too)
  MicroBatcher(size_t row_bytes, uint32_t max_batch, uint32_t window_us, uint32_t linger_us, RunFn fn)
      : row_bytes_(row_bytes), max_batch_(std::max<uint32_t>(1, max_batch)), window_us_(window_us),
        linger_us_(std::min(linger_us, window_us)), fn_(std::move(fn)) {}

  int search(const void *query, uint32_t topk, IndexDocumentList *out) {
    std::shared_ptr<Batch> b;
    bool leader = false;
    uint32_t slot = 0;
    {
      std::unique_lock<std::mutex> lk(mu_);
      // join the open batch if it takes this topk and has room; otherwise wait for it to close and open a new one
      while (open_ && (open_->topk != topk || open_->n >= max_batch_)) cv_.wait(lk);
      b = open_;
      if (!b) {
        b = std::make_shared<Batch>();
        b->topk = topk;
        const auto now = std::chrono::steady_clock::now();
        b->deadline = now + std::chrono::microseconds(window_us_);
        b->linger = now + std::chrono::microseconds(linger_us_);
        open_ = b;
        leader = true;
      }
      slot = b->n++;
      b->queries.insert(b->queries.end(), static_cast<const char *>(query), static_cast<const char *>(query) + row_bytes_);
      if (b->n >= max_batch_) cv_.notify_all();
      if (leader) {
        // collect while an earlier batch is still searching (no added latency on an idle index: a lone caller goes
        // at once unless a linger is configured), at most until the batch is full or the window has passed
        while (b->n < max_batch_) {
          const auto now = std::chrono::steady_clock::now();
          const auto until = inflight_ > 0 ? b->deadline : b->linger;
          if (now >= until) break;
          cv_.wait_until(lk, until);
        }
        open_.reset();                    // the next arrival opens (and leads) the next batch
        ++inflight_;
        cv_.notify_all();
      }
    }
    if (leader) {
      const int rc = fn_(b->queries.data(), b->n, topk, &b->keys, &b->scores, &b->counts);
      {
        std::lock_guard<std::mutex> g(mu_);
        --inflight_;
        cv_.notify_all();                 // a leader that was collecting behind this batch may go now
      }
      {
        std::lock_guard<std::mutex> g(b->mu); // the members of THIS batch wait on its own lock: no stampede on mu_
        b->rc = rc;
        b->done = true;
      }
      b->cv.notify_all();
    } else {
      std::unique_lock<std::mutex> bl(b->mu);
      while (!b->done) b->cv.wait(bl);
    }
    if (b->rc != 0) return b->rc;
    out->clear();
    for (uint32_t j = 0; j < b->counts[slot]; ++j)
      out->emplace_back(b->keys[(size_t)slot * topk + j], b->scores[(size_t)slot * topk + j]);
    return 0;
  }

 private:
  struct Batch {
    uint32_t topk = 0, n = 0;
    bool done = false;
    int rc = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::chrono::steady_clock::time_point deadline, linger;
    std::vector<char> queries;
    std::vector<uint64_t> keys;
    std::vector<float> scores;
    std::vector<uint32_t> counts;
  };
  size_t row_bytes_;
  uint32_t max_batch_, window_us_, linger_us_;
  RunFn fn_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::shared_ptr<Batch> open_;
  uint32_t inflight_ = 0;      // batches currently searching
};

// ---- flat: one class body serves the "FlatStreamer" and "FlatSearcher" registrations -------------
class HipFlatStreamer {
 public:
  ~HipFlatStreamer() { close(); }
  int init(const IndexMeta &meta, const Params & /*params*/) {
    meta_ = meta;
    metric_ = metric_from_name(meta.metric_name());
    if (metric_ < 0 || (meta.data_type() != IndexMeta::DT_FP32 && meta.data_type() != IndexMeta::DT_FP16)) return IndexError_Unsupported;
    return 0;
  }
  int open(int device = 0) {
    device_ = device;
    static std::atomic<uint32_t> next_magic{0x48495031u};    // IndexContext::GenerateMagic (index_context.cc:22-25)
    magic_ = next_magic.fetch_add(1);
    return zvec_hip_flat_create(meta_.dimension(), meta_.data_type() == IndexMeta::DT_FP16 ? ZVEC_HIP_DT_FP16 : ZVEC_HIP_DT_FP32,
                                metric_, device, &h_);
  }
  int close() { int rc = h_ ? zvec_hip_flat_destroy(h_) : 0; h_ = nullptr; return rc; }
  const IndexMeta &meta() const { return meta_; }
  uint32_t magic() const { return magic_; }
  Context::Pointer create_context() const {
    if (!h_) return nullptr;
    Context::Pointer c(new Context(device_, magic_));
    return c->ok() ? std::move(c) : nullptr;
  }
  //! Add a vector into index (index_runner.h:476-480)
  int add_impl(uint64_t key, const void *query, const IndexQueryMeta &qmeta, Context::Pointer & /*context*/) {
    if (!h_ || !query || qmeta.element_size() != meta_.element_size()) return IndexError_InvalidArgument;
    std::unique_lock<FairSharedMutex> w(keys_mu_);    // add vs search: flat_streamer.cc:236-242, flat_streamer_entity.cc:150
    int rc = zvec_hip_flat_append(h_, query, 1, &key);
    if (rc == 0) keys_.push_back(key);
    return rc;
  }
  //! Add a vector with id into index (index_runner.h:483-487) — the call core_interface::Index::_dense_add makes.
  //! FlatStreamerEntity::add_vector_with_id (flat_streamer_entity.cc:900-990): position == id == key; ids beyond the count
  //! leave holes (kInvalidKey rows no search returns), an id below the count overwrites in place
  int add_with_id_impl(uint32_t id, const void *query, const IndexQueryMeta &qmeta, Context::Pointer & /*context*/) {
    if (!h_ || !query || qmeta.element_size() != meta_.element_size()) return IndexError_InvalidArgument;
    std::unique_lock<FairSharedMutex> w(keys_mu_);
    int rc = zvec_hip_flat_put(h_, &id, 1, query, nullptr);
    if (rc != 0) return rc;
    if (keys_.size() <= id) keys_.resize((size_t)id + 1, kInvalidKey);
    keys_[id] = id;
    return 0;
  }
  static constexpr uint64_t kInvalidKey = ~0ull;              // flat_index_format.h:29
  //! bulk form used by FlatBuilder::build / FlatSearcher::load (flat_builder.cc:188-276)
  int add_batch(const void *vecs, uint64_t n, const uint64_t *keys) {
    if (!h_) return IndexError_InvalidArgument;
    std::unique_lock<FairSharedMutex> w(keys_mu_);
    int rc = zvec_hip_flat_append(h_, vecs, n, keys);
    if (rc == 0) for (uint64_t i = 0; i < n; ++i) keys_.push_back(keys ? keys[i] : keys_.size());
    return rc;
  }
  //! Similarity search (index_runner.h:490-500)
  int search_impl(const void *query, const IndexQueryMeta &qmeta, Context::Pointer &context) const {
    return search_impl(query, qmeta, 1, context);
  }
  int search_impl(const void *query, const IndexQueryMeta &qmeta, uint32_t count, Context::Pointer &context) const {
    if (!h_ || !query || qmeta.element_size() != meta_.element_size()) return IndexError_InvalidArgument;
    Context *ctx = context.get();
    if (!ctx || (ctx->topk() == 0 && !ctx->group_by_search())) return IndexError_InvalidArgument;    // flat_searcher.cc:194-198
    if (ctx->magic() != magic_) ctx->set_magic(magic_);                 // context made by another index: re-bind
    std::shared_lock<FairSharedMutex> r(keys_mu_);    // keys_ (filter sweep, bitset size, fetch_vector) vs add
    if (ctx->group_by_search()) return group_search(query, count, ctx, nullptr, nullptr);   // flat_streamer.cc:323-324
    const uint32_t k = ctx->topk();
    std::vector<uint64_t> keys((size_t)count * k);
    std::vector<float> scores((size_t)count * k);
    std::vector<uint32_t> counts(count);
    const uint64_t *bits = nullptr;
    if (ctx->has_doc_filter()) {            // composite filter as data: materialised on the GPU
      ctx->bits().assign((keys_.size() + 63) / 64, 0);
      int frc = zvec_hip_flat_build_filter(h_, ctx->handle(), &ctx->doc_filter(), ctx->bits().data(), 0, nullptr);
      if (frc != 0) return frc;
      bits = ctx->bits().data();
    } else {
      bits = ctx->materialise(keys_);
    }
    int rc = zvec_hip_flat_search(h_, ctx->handle(), query, count, k, ctx->threshold(), bits,
                                  keys.data(), scores.data(), counts.data());
    if (rc != 0) return rc;
    ctx->take(count, k, keys, scores, counts);
    return ctx->fetch_vector() ? attach_vectors(ctx, count) : 0;
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
    if (!h_ || !query || qmeta.element_size() != meta_.element_size() || p_keys.size() != count) return IndexError_InvalidArgument;
    Context *ctx = context.get();
    if (!ctx || (ctx->topk() == 0 && !ctx->group_by_search())) return IndexError_InvalidArgument;
    if (ctx->magic() != magic_) ctx->set_magic(magic_);
    std::shared_lock<FairSharedMutex> r(keys_mu_);
    std::vector<uint32_t> ids, offs(count + 1, 0);
    {
      std::lock_guard<std::mutex> g(map_mu_);
      if (pos_of_key_.size() != keys_.size()) {
        pos_of_key_.clear();
        for (uint64_t i = 0; i < keys_.size(); ++i) pos_of_key_.emplace(keys_[i], i);
      }
      for (uint32_t q = 0; q < count; ++q) {
        for (uint64_t key : p_keys[q]) {
          auto it = pos_of_key_.find(key);
          if (it != pos_of_key_.end()) ids.push_back((uint32_t)it->second);
        }
        offs[q + 1] = (uint32_t)ids.size();
      }
    }
    if (ids.empty()) ids.push_back(0);
    if (ctx->group_by_search()) return group_search(query, count, ctx, ids.data(), offs.data());   // flat_streamer.cc:365-366
    const uint32_t k = ctx->topk();
    std::vector<uint64_t> keys((size_t)count * k);
    std::vector<float> scores((size_t)count * k);
    std::vector<uint32_t> counts(count);
    int rc = zvec_hip_flat_search_by_ids(h_, ctx->handle(), query, count, ids.data(), offs.data(), k, ctx->threshold(),
                                         ctx->materialise(keys_), keys.data(), scores.data(), counts.data());
    if (rc != 0) return rc;
    ctx->take(count, k, keys, scores, counts);
    return ctx->fetch_vector() ? attach_vectors(ctx, count) : 0;
  }
  //! Fetch vector by id (index_runner.h:445-453)
  int get_vector_by_id(uint32_t id, std::vector<float> *out) const {
    if (!h_) return IndexError_InvalidArgument;
    out->resize(meta_.dimension());
    return zvec_hip_flat_get_vector(h_, id, out->data());
  }
  uint64_t count() const { uint64_t n = 0; if (h_) zvec_hip_flat_count(h_, &n); return n; }
 protected:
  //! group_by_search_impl / group_by_search_p_keys_impl (flat_streamer.cc:391-483) + topk_to_group_result
  //! (flat_streamer_context.h:135-180); the caller holds keys_mu_ shared.  ids == nullptr: every row competes
  int group_search(const void *query, uint32_t count, Context *ctx, const uint32_t *ids, const uint32_t *offs) const {
    if (!ctx->group_by_valid()) return IndexError_InvalidArgument;     // "Invalid group-by function"
    const uint32_t gnum = ctx->group_num(), gk = ctx->group_topk();
    if (gk == 0) return IndexError_InvalidArgument;
    std::vector<uint32_t> group_of;
    std::vector<std::string> group_ids;
    ctx->sweep_groups(keys_, &group_of, &group_ids);
    const size_t rows = (size_t)count * gnum;
    std::vector<uint64_t> keys(rows * gk);
    std::vector<float> scores(rows * gk);
    std::vector<uint32_t> counts(rows), groups(rows), ngroups(count);
    const uint32_t none = 0;
    const uint32_t *gof = group_of.empty() ? &none : group_of.data();
    const uint32_t ng = std::max<uint32_t>(1u, (uint32_t)group_ids.size());
    const uint64_t *bits = ctx->materialise(keys_);
    int rc = ids ? zvec_hip_flat_search_grouped_by_ids(h_, ctx->handle(), query, count, ids, offs, gof, ng, gnum, gk, ctx->threshold(),
                                                       bits, groups.data(), ngroups.data(), keys.data(), scores.data(), counts.data())
                 : zvec_hip_flat_search_grouped(h_, ctx->handle(), query, count, gof, ng, gnum, gk, ctx->threshold(), bits,
                                                groups.data(), ngroups.data(), keys.data(), scores.data(), counts.data());
    if (rc != 0) return rc;
    auto &res = ctx->mutable_group_results();
    res.assign(count, IndexGroupDocumentList());
    for (uint32_t q = 0; q < count; ++q) {
      res[q].resize(ngroups[q]);
      for (uint32_t s = 0; s < ngroups[q]; ++s) {
        const size_t row = (size_t)q * gnum + s;
        res[q][s].set_group_id(group_ids[groups[row]]);
        for (uint32_t j = 0; j < counts[row]; ++j) res[q][s].mutable_docs()->emplace_back(keys[row * gk + j], scores[row * gk + j]);
      }
    }
    return 0;
  }
  IndexMeta meta_;
  int metric_{0};
  int device_{0};
  uint32_t magic_{0};
  zvec_hip_flat_t h_{nullptr};
  std::vector<uint64_t> keys_;
  mutable FairSharedMutex keys_mu_;
  mutable std::mutex map_mu_;
  mutable std::unordered_map<uint64_t, uint64_t> pos_of_key_;
  int attach_vectors(Context *ctx, uint32_t count) const {
    zvec_hip_flat_t h = h_;
    return attach_result_vectors(ctx, count, meta_.element_size(), keys_, &pos_of_key_, &map_mu_,
                                 [h](const uint64_t *p, size_t n, void *out) { return zvec_hip_flat_get_vectors(h, p, n, out); });
  }
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
    if (params.get(PARAM_HIP_SEARCHER_BATCH_WINDOW_US, &v)) batch_window_us_ = (uint32_t)v;
    if (params.get(PARAM_HIP_SEARCHER_MAX_BATCH, &v)) max_batch_ = (uint32_t)v;
    if (params.get(PARAM_HIP_SEARCHER_BATCH_LINGER_US, &v)) batch_linger_us_ = (uint32_t)v;
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
    int rc = zvec_hip_ivf_create(meta.dimension(), meta.data_type() == IndexMeta::DT_FP16 ? ZVEC_HIP_DT_FP16 : ZVEC_HIP_DT_FP32,
                                 metric, device, &h_);
    if (rc != 0) return rc;
    rc = zvec_hip_ivf_load(h_, centroids, nlist, list_offsets, vecs, keys);
    if (rc != 0) return rc;
    nlist_ = nlist;
    count_ = list_offsets[nlist];
    keys_.resize(count_);
    for (uint64_t i = 0; i < count_; ++i) keys_[i] = keys ? keys[i] : i;
    if (batch_window_us_ > 0) {
      batcher_.reset(new MicroBatcher(meta_.element_size(), max_batch_, batch_window_us_, batch_linger_us_,
          [this](const void *q, uint32_t n, uint32_t k, std::vector<uint64_t> *ks, std::vector<float> *sc, std::vector<uint32_t> *cn) {
            // a small pool of workspaces shared by the leaders (any thread may lead a batch; creating a context —
            // a stream plus its buffers — per thread would cost more than the searches)
            std::unique_ptr<Context> c;
            {
              std::lock_guard<std::mutex> g(pool_mu_);
              if (!pool_.empty()) { c = std::move(pool_.back()); pool_.pop_back(); }
            }
            if (!c) c.reset(new Context(device_, magic_));
            if (!c->ok()) return (int)IndexError_Runtime;
            ks->assign((size_t)n * k, 0);
            sc->assign((size_t)n * k, 0.f);
            cn->assign(n, 0);
            int rc = zvec_hip_ivf_search(h_, c->handle(), q, n, k, FLT_MAX, nprobe(), max_scan_count(), nullptr, ks->data(),
                                         sc->data(), cn->data());
            std::lock_guard<std::mutex> g(pool_mu_);
            pool_.push_back(std::move(c));
            return rc;
          }));
    }
    return 0;
  }
  int unload() { batcher_.reset(); pool_.clear(); int rc = h_ ? zvec_hip_ivf_destroy(h_) : 0; h_ = nullptr; return rc; }
  Context::Pointer create_context() const {
    if (!h_) return nullptr;                                   // "Load the index first" ivf_searcher.cc:257-260
    Context::Pointer c(new Context(device_, magic_));
    return c->ok() ? std::move(c) : nullptr;
  }
  // IVFSearcherContext::update (ivf_searcher_context.h:61-79)
  uint32_t nprobe() const { return std::max<uint32_t>((uint32_t)std::round((float)nlist_ * scan_ratio_), 1u); }
  uint32_t max_scan_count() const {
    uint32_t m = (uint32_t)std::ceil((float)count_ * scan_ratio_);
    return std::max(bruteforce_threshold_, m);
  }
  int search_impl(const void *query, const IndexQueryMeta &qmeta, Context::Pointer &context) const {
    return search_impl(query, qmeta, 1, context);
  }
  int search_impl(const void *query, const IndexQueryMeta &qmeta, uint32_t count, Context::Pointer &context) const {
    if (h_ && count_ <= bruteforce_threshold_) return search_bf_impl(query, qmeta, count, context);   // ivf_searcher.cc:188-190
    Context *ctx = context.get();
    if (batcher_ && count == 1 && h_ && query && ctx && ctx->topk() != 0 && qmeta.element_size() == meta_.element_size() &&
        !ctx->has_any_filter() && !ctx->fetch_vector() && ctx->threshold() == FLT_MAX) {
      IndexDocumentList r;                       // single plain query: ride a shared batch (MicroBatcher)
      int rc = batcher_->search(query, ctx->topk(), &r);
      if (rc == 0) ctx->take_single(std::move(r));
      return rc;
    }
    return run(query, qmeta, count, context, false);
  }
  int search_bf_impl(const void *query, const IndexQueryMeta &qmeta, Context::Pointer &context) const {
    return run(query, qmeta, 1, context, true);
  }
  int search_bf_impl(const void *query, const IndexQueryMeta &qmeta, uint32_t count, Context::Pointer &context) const {
    return run(query, qmeta, count, context, true);
  }
 private:
  int run(const void *query, const IndexQueryMeta &qmeta, uint32_t count, Context::Pointer &context, bool bf) const {
    if (!h_) return IndexError_NoIndexLoaded;
    if (!query || qmeta.element_size() != meta_.element_size()) return IndexError_InvalidArgument;   // ivf_searcher.cc:191-194
    Context *ctx = context.get();
    if (!ctx || ctx->topk() == 0) return IndexError_InvalidArgument;                                 // ivf_searcher.cc:197-200
    if (ctx->magic() != magic_) ctx->set_magic(magic_);
    const uint32_t k = ctx->topk();
    std::vector<uint64_t> keys((size_t)count * k);
    std::vector<float> scores((size_t)count * k);
    std::vector<uint32_t> counts(count);
    const uint64_t *bits = nullptr;
    if (ctx->has_doc_filter()) {
      ctx->bits().assign((keys_.size() + 63) / 64, 0);
      int frc = zvec_hip_ivf_build_filter(h_, ctx->handle(), &ctx->doc_filter(), ctx->bits().data(), 0, nullptr);
      if (frc != 0) return frc;
      bits = ctx->bits().data();
    } else {
      bits = ctx->materialise(keys_);
    }
    int rc = bf ? zvec_hip_ivf_search_bf(h_, ctx->handle(), query, count, k, ctx->threshold(), bits, keys.data(),
                                         scores.data(), counts.data())
                : zvec_hip_ivf_search(h_, ctx->handle(), query, count, k, ctx->threshold(), nprobe(), max_scan_count(),
                                      bits, keys.data(), scores.data(), counts.data());
    if (rc != 0) return rc;
    ctx->take(count, k, keys, scores, counts);
    if (!ctx->fetch_vector()) return 0;
    zvec_hip_ivf_t h = h_;
    return attach_result_vectors(ctx, count, meta_.element_size(), keys_, &pos_of_key_, &map_mu_,
                                 [h](const uint64_t *p, size_t n, void *out) { return zvec_hip_ivf_get_vectors(h, p, n, out); });
  }
  mutable std::mutex map_mu_;
  mutable std::unordered_map<uint64_t, uint64_t> pos_of_key_;
  std::unique_ptr<MicroBatcher> batcher_;
  mutable std::mutex pool_mu_;
  mutable std::vector<std::unique_ptr<Context>> pool_;
  uint32_t batch_window_us_{0}, max_batch_{1024}, batch_linger_us_{0};
  IndexMeta meta_;
  int device_{0};
  uint32_t magic_{0};
  zvec_hip_ivf_t h_{nullptr};
  uint32_t nlist_{0};
  uint64_t count_{0};
  float scan_ratio_{0.1f};                 // kDefaultScanRatio  ivf_searcher_context.h:211
  uint32_t bruteforce_threshold_{1000u};   // kDefaultBfThreshold ivf_searcher_context.h:212
  std::vector<uint64_t> keys_;
};

// "IVFStreamer" is what the product instantiates (indexes/ivf_index.cc:38-39); in the reference it is the same read-only
// operator over a dumped index as the searcher (ivf_streamer.h:28-85: open / search / unload, no add_impl)
using HipIVFStreamer = HipIVFSearcher;

}  // namespace zvec_hip_host
