// hip_plugin.cc — the zvec-side binding of the MI355X scan core: IndexStreamer / IndexSearcher subclasses that forward
// to the C ABI of include/zvec_hip.h and register themselves with the reference's factory.
//
// This file is meant to be compiled INSIDE a zvec checkout (e.g. as src/core/algorithm/hip/hip_plugin.cc, or into
// libzvec_hip_plugin.so loaded with IndexPluginBroker::emplace, index_plugin.h:73-101) against zvec's own headers:
//     g++ -std=c++17 -fsyntax-only -I<zvec>/src/include -I<zvec>/src -I<this repo>/include plugin/hip_plugin.cc
// (__graft_entry__.build() runs exactly this when /root/reference is present; nothing of the reference is copied or
// shipped.)  Interfaces implemented, all pure virtuals included:
//   IndexStreamer   src/include/zvec/core/framework/index_streamer.h:29-53   init/open/flush/close/meta
//   IndexSearcher   src/include/zvec/core/framework/index_searcher.h:30-54   init/meta/params/load
//   IndexRunner     src/include/zvec/core/framework/index_runner.h:400-740   stats/cleanup/unload/create_context/
//                   create_provider/get_vector*/add_impl/search_impl x2/search_bf_impl x2/search_bf_by_p_keys_impl
//   IndexContext    src/include/zvec/core/framework/index_context.h:57-262   set_topk/topk/result/mutable_result/
//                   update/magic/reset/set_fetch_vector (+ inherited filter(), threshold())
//   registration    src/include/zvec/core/framework/index_factory.h:237-250
//
// Division of labour: persistence stays with the reference's own operators and storage engine (out of scope for the
// GPU core) — the mutable flat streamer WRAPS the registered "FlatStreamer" for open / add / flush / get_vector and
// mirrors the rows into HBM; the immutable searchers read the dumped segments through IndexStorage exactly as
// FlatSearcher::load / IVFSearcher::load do and hand the payloads to the loaders of the C ABI.  Every search goes to
// the GPU; there is no CPU fallback (a missing device makes open / load fail with the ABI's error).
#include <zvec/core/framework/index_factory.h>
#include <zvec/core/framework/index_helper.h>
#include <zvec/core/framework/index_provider.h>
#include <zvec/core/framework/index_reformer.h>
#include <zvec/core/framework/index_searcher.h>
#include <zvec/core/framework/index_segment_storage.h>
#include <zvec/core/framework/index_streamer.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "algorithm/flat/flat_index_format.h"    // StreamerLinearMeta, BlockHeader, DeletionMap: the mutable streamer's persisted blocks
#include "algorithm/flat/flat_utility.h"         // its segment ids

#include "hip_plugin_common.h"
#include "zvec_hip.h"
#include "zvec_hip_operator.hpp"

namespace zvec {
namespace core {

namespace {

// parameter keys of the operators this plugin stands in for
const std::string kParamBruteForceThreshold("proxima.ivf.searcher.brute_force_threshold");  // ivf_params.h:46-47
const std::string kParamHipDeviceCount("proxima.hip.device_count");   // new: > 1 = shard the index over devices [device, device + count)
// new: the micro-batcher of single-query searches (zvec_hip_operator.hpp); window_us > 0 turns it on
const std::string kParamHipBatchWindowUs("proxima.hip.searcher.batch_window_us");
const std::string kParamHipMaxBatch("proxima.hip.searcher.max_batch");
const std::string kParamHipBatchLingerUs("proxima.hip.searcher.batch_linger_us");
// new: IVF searches pre-select on an fp16 twin of the fp32 lists, re-score in fp32, certify (zvec_hip_ivf_set_shadow); off by default
const std::string kParamHipHalfWidthPreselect("proxima.hip.searcher.half_width_preselect");
const std::string kParamHipPreselectRows("proxima.hip.searcher.preselect_rows");

// segment ids (flat_utility.h:32-34, ivf_index_format.h:152-164)
const std::string kFlatKeys("flat.keys"), kFlatFeatures("flat.features");
const std::string kIvfCentroid("ivf.centroid"), kIvfBody("ivf.inverted_body"), kIvfHeader("ivf.inverted_header"),
    kIvfMeta("ivf.inverted_meta"), kIvfKeys("hc.keys");

// whole payload of a segment as one contiguous host buffer.  Zero-copy Segment::read first, as the reference's own loaders
// do (flat_searcher.cc:139-145, ivf_entity.cc:443-570) — the storages that map or hold the file serve it without a copy, and
// IndexMemory::Block::fetch ignores its offset (index_memory.h:86-97), so "MemoryReadStorage" is only correct through read();
// storages that cannot lend a pointer (file reads) fall back to fetch
int read_segment(IndexStorage *stg, const std::string &id, std::string *out, int level = -1) {
  auto seg = stg->get(id, level);
  if (!seg) return IndexError_NoExist;
  const size_t size = seg->data_size();
  out->resize(size);
  if (size == 0) return 0;
  const void *p = nullptr;
  if (seg->read(0, &p, size) == size && p) {
    memcpy(&(*out)[0], p, size);
    return 0;
  }
  if (seg->fetch(0, &(*out)[0], size) != size) return IndexError_ReadData;
  return 0;
}

}  // namespace

/*! Search context: one per caller thread (index.cc:24-45 caches it thread-local per index type and may hand it to
 *  another index instance of that type => magic re-binding, flat_streamer.cc:319-321 / ivf_streamer.cc:198-202).
 *  The HIP side (stream + workspace, zvec_hip_ctx_t) is index-agnostic, so re-binding is only bookkeeping.  The op_*
 *  accessors are what the shared operator logic (include/zvec_hip_operator.hpp) reads a context through. */
class HipContext : public IndexContext {
 public:
  HipContext(int device, uint32_t magic) : magic_(magic) { rc_ = zvec_hip_ctx_create(device, &h_); }
  ~HipContext() override { if (h_) zvec_hip_ctx_destroy(h_); }

  void set_topk(uint32_t k) override { topk_ = k; }
  uint32_t topk() const override { return topk_; }
  void set_fetch_vector(bool on) override { fetch_vector_ = on; }
  bool fetch_vector() const override { return fetch_vector_; }
  const IndexDocumentList &result() const override { return results_[0]; }
  const IndexDocumentList &result(size_t i) const override { return results_[i]; }
  IndexDocumentList *mutable_result(size_t i) override { return &results_[i]; }
  uint32_t magic() const override { return magic_; }
  void reset() override { results_.assign(1, IndexDocumentList()); }
  //! IVFSearcherContext::update (ivf_searcher_context.h:61-79); flat contexts ignore it (flat_searcher_context.h)
  int update(const ailego::Params &params) override {
    params.get(kParamBruteForceThreshold, &bruteforce_threshold_);
    params.get(kParamScanRatio, &scan_ratio_);
    if (scan_ratio_ <= 0.0f) return IndexError_InvalidArgument;
    return 0;
  }
  void set_bruteforce_threshold(uint32_t v) override { bruteforce_threshold_ = v; }
  //! group-by search (flat_searcher_context.h:173-183): group_num > 0 switches the next searches to it
  void set_group_params(uint32_t group_num, uint32_t group_topk) override { group_num_ = group_num; group_topk_ = group_topk; }
  bool group_by_search() const { return group_num_ > 0; }
  const IndexGroupDocumentList &group_result() const override { return group_results_[0]; }
  const IndexGroupDocumentList &group_result(size_t i) const override { return group_results_[i]; }

  // ---- what zvec_hip_op::FlatOperator / IVFOperator read (zvec_hip_operator.hpp, "Requirements on the template arguments")
  zvec_hip_ctx_t hip() const { return h_; }
  bool op_has_filter() const { return filter().is_valid(); }
  bool op_filtered(uint64_t key) const { return filter()(key); }
  bool op_has_group_by() const { return group_by().is_valid(); }
  std::string op_group_of(uint64_t key) const { return group_by()(key); }
  uint32_t op_group_num() const { return group_num_; }
  uint32_t op_group_topk() const { return group_topk_; }
  const uint64_t *op_preset_bits() const { return nullptr; }             // (IndexContext has no side channel for a bitmap yet)
  const zvec_hip_doc_filter_t *op_doc_filter() const { return nullptr; }
  zvec_hip_op::Scratch &op_scratch() { return scratch_; }
  std::vector<IndexDocumentList> &op_results() { return results_; }
  std::vector<IndexGroupDocumentList> &op_group_results() { return group_results_; }

  zvec_hip_ctx_t h_{nullptr};
  int rc_{0};
  uint32_t topk_{0}, magic_{0};
  bool fetch_vector_{false};
  float scan_ratio_{0.1f};                  // ivf_searcher_context.h:211-213 defaults
  uint32_t bruteforce_threshold_{1000};
  std::vector<IndexDocumentList> results_{1};
  uint32_t group_num_{0}, group_topk_{0};
  std::vector<IndexGroupDocumentList> group_results_{1};
  zvec_hip_op::Scratch scratch_;            // result arrays, swept filter / groups, the fetch_vector payload the documents point
                                            // into (valid until the next search on this context)
  IndexContext::Pointer inner_;             // HipFlatStreamer: a context of the wrapped reference streamer (its add paths cast-check one)
};

//! the framework's document types, as the shared operator logic needs them
struct PluginDocs {
  using Document = IndexDocument;
  using DocumentList = IndexDocumentList;
  using GroupDocument = GroupIndexDocument;
  using GroupDocumentList = IndexGroupDocumentList;
  static Document make(uint64_t key, float score) { return IndexDocument(key, score); }
  //! fetch_vector: the document points into the context's payload (index_document.h:207-216; valid until the next search)
  static void attach(Document &d, uint32_t index, const char *row, size_t /*bytes*/) { d = IndexDocument(d.key(), d.score(), index, row); }
};
using HipFlatCore = zvec_hip_op::FlatOperator<HipContext, PluginDocs>;
using HipIVFOp = zvec_hip_op::IVFOperator<HipContext, PluginDocs>;

namespace {

HipContext *bind(IndexContext::Pointer &c, uint32_t magic) {
  auto *ctx = dynamic_cast<HipContext *>(c.get());
  if (!ctx || ctx->rc_ != 0 || (ctx->topk() == 0 && !ctx->group_by_search())) return nullptr;   // "Invalid context or topk not set yet"
  if (ctx->magic_ != magic) { ctx->magic_ = magic; ctx->reset(); }
  return ctx;
}

//! proxima.hip.searcher.batch_window_us (> 0 turns the micro-batcher on; off by default), .max_batch, .batch_linger_us
zvec_hip_op::BatcherOptions batcher_options(const ailego::Params &params) {
  zvec_hip_op::BatcherOptions bo;
  params.get(kParamHipBatchWindowUs, &bo.window_us);
  params.get(kParamHipMaxBatch, &bo.max_batch);
  params.get(kParamHipBatchLingerUs, &bo.linger_us);
  params.get(kParamHipHalfWidthPreselect, &bo.shadow);
  params.get(kParamHipPreselectRows, &bo.shadow_preselect);
  if (bo.max_batch == 0) bo.max_batch = 1024;
  return bo;
}

//! create the flat operator for an index of `meta`
int create_flat(HipFlatCore *core, const IndexMeta &meta, int device, uint32_t ndev, const ailego::Params &params) {
  const int metric = metric_of(meta), dtype = dtype_of(meta);
  if (metric < 0 || dtype < 0) return IndexError_Unsupported;
  return core->create(meta.dimension(), dtype, metric, meta.element_size(), device, ndev, /*position_is_id=*/false,
                      batcher_options(params));
}

}  // namespace

/*! IVFIndexProvider (ivf_index_provider.h:24-106) / the flat searcher's provider over the device-resident rows: documents in
 *  storage (list) order; the iterator pulls the rows to the host 4096 at a time (one gather launch each); like the
 *  reference's, data() is valid until next().  Core = HipFlatCore or HipIVFCore. */
template <typename Core>
class HipRowsProvider : public IndexProvider {
 public:
  HipRowsProvider(const IndexMeta &meta, const Core *core, const std::string &owner) : meta_(meta), core_(core), owner_(owner) {}
  Iterator::Pointer create_iterator() override { return Iterator::Pointer(new Walk(core_)); }
  size_t count() const override { return core_->count(); }
  size_t dimension() const override { return meta_.dimension(); }
  IndexMeta::DataType data_type() const override { return meta_.data_type(); }
  size_t element_size() const override { return meta_.element_size(); }
  const void *get_vector(const uint64_t key) const override {
    static thread_local std::string row;
    row.resize(core_->elem_size());
    return core_->vector_of_key(key, &row[0]) == 0 ? row.data() : nullptr;
  }
  int get_vector(const uint64_t key, IndexStorage::MemoryBlock &block) const override {
    const void *p = this->get_vector(key);
    if (!p) return IndexError_NoExist;
    block.reset(const_cast<void *>(p));
    return 0;
  }
  const std::string &owner_class() const override { return owner_; }

 private:
  class Walk : public Iterator {
   public:
    explicit Walk(const Core *core) : core_(core) { fetch(); }
    const void *data() const override { return chunk_.data() + (pos_ - chunk0_) * core_->elem_size(); }
    bool is_valid() const override { return ok_ && pos_ < core_->count(); }
    uint64_t key() const override { return core_->key_at(pos_); }
    void next() override {
      ++pos_;
      if (pos_ >= chunk0_ + chunk_n_) fetch();
    }
   private:
    void fetch() {
      chunk0_ = pos_;
      chunk_n_ = std::min<size_t>(4096, core_->count() > pos_ ? core_->count() - pos_ : 0);
      chunk_.resize(chunk_n_ * core_->elem_size());
      ok_ = chunk_n_ == 0 || core_->rows_at(chunk0_, chunk_n_, &chunk_[0]) == 0;
    }
    const Core *core_;
    std::string chunk_;
    size_t pos_{0}, chunk0_{0}, chunk_n_{0};
    bool ok_{true};
  };
  IndexMeta meta_;
  const Core *core_;
  std::string owner_;
};

//! create_provider / get_vector / get_vector_by_key / get_vector_by_id of the immutable operators (ivf_streamer.h:74-85: the id
//! is looked up as a key; flat_searcher.h:111,158)
#define ZVEC_HIP_VECTOR_ACCESSORS(CORE, OWNER)                                                                                 \
  Provider::Pointer create_provider() const override { return Provider::Pointer(new HipRowsProvider<CORE>(meta_, &core_, OWNER)); } \
  const void *get_vector(uint64_t key) const override {                                                                        \
    static thread_local std::string row;                                                                                       \
    row.resize(core_.elem_size());                                                                                             \
    return core_.vector_of_key(key, &row[0]) == 0 ? row.data() : nullptr;                                                      \
  }                                                                                                                            \
  int get_vector(const uint64_t key, IndexStorage::MemoryBlock &block) const override {                                        \
    const void *p = this->get_vector(key);                                                                                     \
    if (!p) return IndexError_NoExist;                                                                                         \
    block.reset(const_cast<void *>(p));                                                                                        \
    return 0;                                                                                                                  \
  }                                                                                                                            \
  int get_vector_by_key(const uint64_t key, IndexStorage::MemoryBlock &block) const override { return this->get_vector(key, block); } \
  int get_vector_by_id(const uint32_t id, IndexStorage::MemoryBlock &block) const override { return this->get_vector(id, block); }

/*! "HipFlatSearcher": stands where FlatSearcher<32> is registered (flat_searcher.cc:247-250). */
class HipFlatSearcher : public IndexSearcher {
 public:
  int init(const ailego::Params &params) override {
    params_ = params;
    params.get(kParamHipDevice, &device_);
    params.get(kParamHipDeviceCount, &ndev_);
    if (ndev_ == 0) ndev_ = 1;
    return 0;
  }
  int cleanup() override { return this->unload(); }
  //! FlatSearcher::load (flat_searcher.cc:68-157): meta + "flat.keys" + "flat.features" (row-major, or 32-row blocks
  //! transposed for a column-major index, flat_builder.cc:188-276) -> HBM
  int load(IndexStorage::Pointer stg, IndexMetric::Pointer /*metric*/) override {
    if (!stg) return IndexError_InvalidArgument;
    int rc = IndexHelper::DeserializeFromStorage(stg.get(), &meta_);
    if (rc != 0) return rc;
    std::string keys, features;
    if ((rc = read_segment(stg.get(), kFlatKeys, &keys)) != 0) return rc;
    if ((rc = read_segment(stg.get(), kFlatFeatures, &features)) != 0) return rc;
    if (keys.size() % sizeof(uint64_t) != 0) return IndexError_InvalidLength;
    const size_t n = keys.size() / sizeof(uint64_t);
    if (n * meta_.element_size() != features.size()) return IndexError_Mismatch;
    if ((rc = create_flat(&core_, meta_, device_, ndev_, params_)) != 0) return rc;
    rc = core_.load_features(features.data(), features.size(), n, meta_.major_order() == IndexMeta::MO_COLUMN,
                             reinterpret_cast<const uint64_t *>(keys.data()));
    if (rc != 0) return rc;
    magic_ = IndexContext::GenerateMagic();
    stats_.set_loaded_count(n);
    return 0;
  }
  int unload() override { core_.destroy(); return 0; }
  const Stats &stats() const override { return stats_; }
  const IndexMeta &meta() const override { return meta_; }
  const ailego::Params &params() const override { return params_; }
  Context::Pointer create_context() const override { return Context::Pointer(new HipContext(device_, magic_)); }
  ZVEC_HIP_VECTOR_ACCESSORS(HipFlatCore, "HipFlatSearcher")
  int search_impl(const void *q, const IndexQueryMeta &qm, Context::Pointer &c) const override { return search_impl(q, qm, 1, c); }
  int search_impl(const void *q, const IndexQueryMeta &qm, uint32_t count, Context::Pointer &c) const override {
    HipContext *ctx = bind(c, magic_);
    if (!ctx || !q || qm.element_size() != core_.elem_size()) return IndexError_InvalidArgument;
    return core_.search(q, count, ctx);
  }
  int search_bf_impl(const void *q, const IndexQueryMeta &qm, Context::Pointer &c) const override { return search_impl(q, qm, 1, c); }
  int search_bf_impl(const void *q, const IndexQueryMeta &qm, uint32_t count, Context::Pointer &c) const override {
    return search_impl(q, qm, count, c);
  }
  int search_bf_by_p_keys_impl(const void *q, const std::vector<std::vector<uint64_t>> &p_keys, const IndexQueryMeta &qm,
                               Context::Pointer &c) const override {
    return search_bf_by_p_keys_impl(q, p_keys, qm, 1, c);
  }
  int search_bf_by_p_keys_impl(const void *q, const std::vector<std::vector<uint64_t>> &p_keys, const IndexQueryMeta &qm,
                               uint32_t count, Context::Pointer &c) const override {
    HipContext *ctx = bind(c, magic_);
    if (!ctx || !q || qm.element_size() != core_.elem_size()) return IndexError_InvalidArgument;
    return core_.search_by_keys(q, p_keys, count, ctx);
  }

 private:
  IndexMeta meta_;
  ailego::Params params_;
  Stats stats_;
  int device_{0};
  uint32_t ndev_{1};
  uint32_t magic_{0};
  HipFlatCore core_;
};

/*! "HipFlatStreamer": stands where FlatStreamer<32> is registered (flat_streamer.cc:486-489).  Persistence (linked
 *  32-vector blocks in the storage, flat_streamer_entity.cc:43-47) stays with the wrapped reference streamer; its rows
 *  are mirrored in HBM, where every search runs. */
class HipFlatStreamer : public IndexStreamer {
 public:
  int init(const IndexMeta &meta, const ailego::Params &params) override {
    meta_ = meta;
    params_ = params;
    params.get(kParamHipDevice, &device_);
    params.get(kParamHipDeviceCount, &ndev_);
    if (ndev_ == 0) ndev_ = 1;
    if (metric_of(meta) < 0 || dtype_of(meta) < 0) return IndexError_Unsupported;
    store_ = IndexFactory::CreateStreamer("FlatStreamer");
    if (!store_) return IndexError_NoExist;
    return store_->init(meta, params);
  }
  int cleanup() override { core_.destroy(); return store_ ? store_->cleanup() : 0; }
  int open(IndexStorage::Pointer stg) override {
    stg_keep_ = stg;
    int rc = store_->open(std::move(stg));
    if (rc != 0) return rc;
    if ((rc = create_flat(&core_, meta_, device_, ndev_, params_)) != 0) return rc;
    // rows already persisted: the block runs of the storage's feature segments go to HBM whole (bulk_open); a storage whose
    // layout this reader does not recognise falls back to the reference streamer's provider walk (key, vector), row by row
    int bulk = this->bulk_open(stg_keep_.get());
    if (bulk < 0) return bulk;
    auto provider = bulk == 0 ? store_->create_provider() : Provider::Pointer();
    if (provider) {
      const size_t es = meta_.element_size(), chunk = 16384;
      std::string rows;
      std::vector<uint64_t> keys;
      for (auto it = provider->create_iterator(); it && it->is_valid(); it->next()) {
        rows.append(static_cast<const char *>(it->data()), es);
        keys.push_back(it->key());
        if (keys.size() == chunk) {
          if ((rc = core_.append(rows.data(), keys.size(), keys.data())) != 0) return rc;
          rows.clear();
          keys.clear();
        }
      }
      if (!keys.empty() && (rc = core_.append(rows.data(), keys.size(), keys.data())) != 0) return rc;
    }
    magic_ = IndexContext::GenerateMagic();
    return 0;
  }
  int flush(uint64_t check_point) override { return store_->flush(check_point); }
  int close() override { core_.destroy(); stg_keep_.reset(); return store_->close(); }
  const IndexMeta &meta() const override { return meta_; }
  const Stats &stats() const override { return store_->stats(); }
  int dump(const IndexDumper::Pointer &dumper) override { return store_->dump(dumper); }
  Context::Pointer create_context() const override { return Context::Pointer(new HipContext(device_, magic_)); }
  Provider::Pointer create_provider() const override { return store_->create_provider(); }
  const void *get_vector(uint64_t key) const override { return store_->get_vector(key); }
  int get_vector(const uint64_t key, IndexStorage::MemoryBlock &block) const override { return store_->get_vector(key, block); }
  int get_vector_by_key(const uint64_t key, IndexStorage::MemoryBlock &block) const override {
    return store_->get_vector_by_key(key, block);
  }
  int get_vector_by_id(const uint32_t id, IndexStorage::MemoryBlock &block) const override {
    return store_->get_vector_by_id(id, block);
  }
  //! add_impl / add_with_id_impl (index_runner.h:476-487): persist through the reference streamer, then mirror
  int add_impl(uint64_t key, const void *vec, const IndexQueryMeta &qm, Context::Pointer &c) override {
    if (!vec || qm.element_size() != meta_.element_size()) return IndexError_InvalidArgument;
    Context::Pointer *inner = inner_context(c);
    if (!inner) return IndexError_Cast;                      // flat_streamer.cc:226-231 "Failed to cast FlatStreamerContext"
    int rc = store_->add_impl(key, vec, qm, *inner);
    return rc != 0 ? rc : core_.append(vec, 1, &key);
  }
  int add_with_id_impl(uint32_t id, const void *vec, const IndexQueryMeta &qm, Context::Pointer &c) override {
    if (!vec || qm.element_size() != meta_.element_size()) return IndexError_InvalidArgument;
    Context::Pointer *inner = inner_context(c);
    if (!inner) return IndexError_Cast;
    int rc = store_->add_with_id_impl(id, vec, qm, *inner);
    return rc != 0 ? rc : core_.put(id, vec);
  }
  int search_impl(const void *q, const IndexQueryMeta &qm, Context::Pointer &c) const override { return search_impl(q, qm, 1, c); }
  int search_impl(const void *q, const IndexQueryMeta &qm, uint32_t count, Context::Pointer &c) const override {
    HipContext *ctx = bind(c, magic_);
    if (!ctx || !q || qm.element_size() != core_.elem_size()) return IndexError_InvalidArgument;
    return core_.search(q, count, ctx);
  }
  int search_bf_impl(const void *q, const IndexQueryMeta &qm, Context::Pointer &c) const override { return search_impl(q, qm, 1, c); }
  int search_bf_impl(const void *q, const IndexQueryMeta &qm, uint32_t count, Context::Pointer &c) const override {
    return search_impl(q, qm, count, c);
  }
  int search_bf_by_p_keys_impl(const void *q, const std::vector<std::vector<uint64_t>> &p_keys, const IndexQueryMeta &qm,
                               Context::Pointer &c) const override {
    return search_bf_by_p_keys_impl(q, p_keys, qm, 1, c);
  }
  int search_bf_by_p_keys_impl(const void *q, const std::vector<std::vector<uint64_t>> &p_keys, const IndexQueryMeta &qm,
                               uint32_t count, Context::Pointer &c) const override {
    HipContext *ctx = bind(c, magic_);
    if (!ctx || !q || qm.element_size() != core_.elem_size()) return IndexError_InvalidArgument;
    return core_.search_by_keys(q, p_keys, count, ctx);
  }

 private:
  IndexMeta meta_;
  ailego::Params params_;
  int device_{0};
  uint32_t ndev_{1};
  uint32_t magic_{0};
  //! The persisted rows of a FlatStreamer storage, read the way FlatStreamerEntity lays them out (flat_streamer_entity.cc:43-47,
  //! flat_streamer_entity.h:287-311; meta StreamerLinearMeta, flat_index_format.h:128-146): segments "flat.features1" ..
  //! "flat.features<segment_count>", each a run of blocks [bvc x element][bvc x key] ... [DeletionMap][BlockHeader].  Live rows =
  //! below the block's vector_count, not deleted, key valid — the rows FlatStreamerEntity::search scans (flat_streamer_entity.cc:
  //! 212-316), in the order its iterator walks them (:428-460).  Returns 1 when loaded, 0 to fall back to the provider walk.
  int bulk_open(IndexStorage *stg) {
    if (!stg) return 0;
    auto mseg = stg->get(FLAT_LINEAR_META_SEG_ID);
    if (!mseg || mseg->data_size() < sizeof(StreamerLinearMeta)) return 0;
    const void *mp = nullptr;
    if (mseg->read(0, &mp, sizeof(StreamerLinearMeta)) != sizeof(StreamerLinearMeta) || !mp) return 0;
    StreamerLinearMeta lm;
    memcpy(static_cast<void *>(&lm), mp, sizeof(lm));
    const uint32_t bvc = lm.header.block_vector_count, bs = lm.header.block_size;
    const size_t es = meta_.element_size();
    if (bvc == 0 || bvc > 32 || bs < bvc * (es + 8) + sizeof(DeletionMap) + sizeof(BlockHeader)) return 0;
    for (uint32_t si = 1; si <= lm.segment_count; ++si) {
      auto seg = stg->get(FLAT_SEGMENT_FEATURES_SEG_ID + std::to_string(si));
      if (!seg) return core_.count() == 0 ? 0 : (int)IndexError_InvalidFormat;
      const size_t nblk = seg->data_size() / bs;
      if (nblk == 0) continue;
      const void *p = nullptr;
      if (seg->read(0, &p, nblk * bs) != nblk * bs || !p) return core_.count() == 0 ? 0 : (int)IndexError_ReadData;
      std::vector<uint32_t> keep(nblk, 0);
      const char *base = static_cast<const char *>(p);
      for (size_t b = 0; b < nblk; ++b) {
        BlockHeader hd;
        DeletionMap dm;
        memcpy(static_cast<void *>(&hd), base + (b + 1) * bs - sizeof(BlockHeader), sizeof(hd));
        memcpy(static_cast<void *>(&dm), base + (b + 1) * bs - sizeof(BlockHeader) - sizeof(DeletionMap), sizeof(dm));
        const uint64_t *keys = reinterpret_cast<const uint64_t *>(base + b * bs + bvc * es);
        uint32_t m = 0;
        for (uint32_t r = 0; r < hd.vector_count && r < bvc; ++r) {
          uint64_t key;
          memcpy(&key, keys + r, 8);
          if (!dm.test(r) && key != kInvalidKey) m |= 1u << r;
        }
        keep[b] = m;
      }
      int rc = core_.load_blocks(p, nblk * bs, nblk, bs, bvc, keep);       // (the extent seg->read just lent)
      if (rc != 0) return rc;
    }
    return 1;
  }

  //! the wrapped streamer's add paths insist on a context of their own type (dynamic_cast, flat_streamer.cc:226-231,276-281):
  //! the caller's HipContext carries one, made on first use
  Context::Pointer *inner_context(Context::Pointer &c) const {
    auto *ctx = dynamic_cast<HipContext *>(c.get());
    if (!ctx) return nullptr;
    if (!ctx->inner_) ctx->inner_ = store_->create_context();
    return ctx->inner_ ? &ctx->inner_ : nullptr;
  }

  IndexStreamer::Pointer store_;          // the reference's FlatStreamer: storage engine side
  IndexStorage::Pointer stg_keep_;        // the storage it was opened on (bulk_open reads the block runs from it)
  HipFlatCore core_;
};

// =====================================================================================================================
// IVF-Flat
// =====================================================================================================================
/*! The storage side of the IVF operators: reads a dumped index through IndexStorage exactly as IVFSearcher::load does and hands
 *  the payloads to the shared operator (zvec_hip_op::IVFOperator), which owns every search. */
class HipIVFCore {
 public:
  void destroy() { op_.destroy(); creformer_.reset(); }
  //! IVFSearcher::load (ivf_searcher.cc:43-103) + IVFEntity::load (ivf_entity.cc:443-570): the centroid index is a
  //! nested flat index inside the "ivf.centroid" segment; the inverted lists come as header / meta / body / keys
  //! ndev > 1: whole inverted lists dealt over the devices (byte-balanced map), centroids replicated
  int load(IndexStorage *stg, IndexMeta *meta, int device, uint32_t ndev, const ailego::Params &params) {
    destroy();
    int rc = IndexHelper::DeserializeFromStorage(stg, meta);
    if (rc != 0) return rc;
    const int metric = metric_of(*meta), dtype = dtype_of(*meta);
    if (metric < 0 || dtype < 0) return IndexError_Unsupported;
    const uint32_t elem_size = meta->element_size();
    // centroid rows: features of the nested FlatSearcher index, put in centroid-id order
    auto cseg = stg->get(kIvfCentroid, 0);
    if (!cseg) return IndexError_InvalidFormat;
    IndexStorage::Pointer nested = std::make_shared<IndexSegmentStorage>(cseg);
    if ((rc = nested->open(std::string(), false)) != 0) return rc;
    IndexMeta cmeta;
    if ((rc = IndexHelper::DeserializeFromStorage(nested.get(), &cmeta)) != 0) return rc;
    // The centroid index may live in a space of its own: IVFBuilder trains inner-product indexes through a MipsConverter
    // (ivf_builder.cc:552-555), the nested index then holds converted centroids (more dimensions, squared-Euclidean) and names
    // the reformer every query goes through before the coarse scan (IVFCentroidIndex::load / search, ivf_centroid_index.cc:
    // 273-297,538-562).  Same element type only: quantised centroid indexes (int8 / int4 reformers) are not taken.
    const size_t celem = cmeta.element_size();
    if (cmeta.data_type() != meta->data_type()) return IndexError_Unsupported;
    if (!cmeta.reformer_name().empty()) {
      creformer_ = IndexFactory::CreateReformer(cmeta.reformer_name());
      if (!creformer_) return IndexError_NoExist;
      if ((rc = creformer_->init(cmeta.reformer_params())) != 0) return rc;
    } else if (celem != elem_size) {
      return IndexError_Unsupported;
    }
    std::string ckeys, cfeat;
    if ((rc = read_segment(nested.get(), kFlatKeys, &ckeys)) != 0) return rc;
    if ((rc = read_segment(nested.get(), kFlatFeatures, &cfeat)) != 0) return rc;
    const size_t nlist = ckeys.size() / sizeof(uint64_t);
    if (nlist == 0 || nlist * celem != cfeat.size()) return IndexError_Mismatch;
    std::string centroids(cfeat.size(), '\0');
    const uint64_t *ck = reinterpret_cast<const uint64_t *>(ckeys.data());
    const size_t unit = IndexMeta::AlignSizeof(cmeta.data_type()), cols = celem / unit;
    const bool colmajor = cmeta.major_order() == IndexMeta::MO_COLUMN;
    for (size_t i = 0; i < nlist; ++i) {
      if (ck[i] >= nlist) return IndexError_InvalidFormat;
      char *dst = &centroids[ck[i] * celem];
      const size_t blk = i / 32, r = i % 32;
      if (colmajor && (blk + 1) * 32 <= nlist) {             // full 32-row block, transposed in units (flat_builder.cc:231-262)
        const char *b0 = cfeat.data() + blk * 32 * celem;
        for (size_t u = 0; u < cols; ++u) memcpy(dst + u * unit, b0 + (u * 32 + r) * unit, unit);
      } else {
        memcpy(dst, cfeat.data() + i * celem, celem);
      }
    }
    const int cmetric = metric_of(cmeta);
    if (creformer_ && (cmetric != ZVEC_HIP_METRIC_L2 && cmetric != ZVEC_HIP_METRIC_IP)) return IndexError_Unsupported;
    std::string header, lmeta, body, keys;
    if ((rc = read_segment(stg, kIvfHeader, &header)) != 0) return rc;
    if ((rc = read_segment(stg, kIvfMeta, &lmeta)) != 0) return rc;
    if ((rc = read_segment(stg, kIvfBody, &body)) != 0) return rc;
    if ((rc = read_segment(stg, kIvfKeys, &keys)) != 0) return rc;
    if ((rc = op_.create(meta->dimension(), dtype, metric, elem_size, device, ndev, batcher_options(params))) != 0) return rc;
    HipIVFOp::CoarseReform reform;
    if (creformer_) {
      // IVFCentroidIndex::search: the queries reformed for the coarse space — in one call where the reformer has a batched
      // transform, one by one where it has not.  What comes back must be rows of the installed coarse space (element type and
      // dimension of the nested index's meta): anything else would be read past its end on the device.
      IndexReformer::Pointer rf = creformer_;
      const IndexQueryMeta in_qmeta(meta->data_type(), meta->dimension());
      const IndexMeta::DataType ctype = cmeta.data_type();
      const uint32_t cdim = cmeta.dimension();
      reform = [rf, in_qmeta, elem_size, ctype, cdim](const void *q, uint32_t count, std::string *cq) -> int {
        IndexQueryMeta ometa;
        int rrc = rf->transform(q, in_qmeta, count, cq, &ometa);
        if (rrc == IndexError_Unsupported || rrc == IndexError_NotImplemented) {
          cq->clear();
          for (uint32_t i = 0; i < count; ++i) {
            std::string one;
            if ((rrc = rf->transform(static_cast<const char *>(q) + size_t(i) * elem_size, in_qmeta, &one, &ometa)) != 0) return rrc;
            cq->append(one);
          }
        } else if (rrc != 0) {
          return rrc;
        }
        if (ometa.data_type() != ctype || ometa.dimension() != cdim) return IndexError_Mismatch;
        return 0;
      };
    }
    return op_.load_segments(header, lmeta, body, keys, centroids, (uint32_t)nlist, creformer_ ? cmeta.dimension() : 0, cmetric,
                             std::move(reform));
  }
  int vector_of_key(uint64_t key, void *out) const { return op_.vector_of_key(key, out); }
  int rows_at(uint64_t pos0, size_t n, void *out) const { return op_.rows_at(pos0, n, out); }
  uint64_t key_at(size_t pos) const { return op_.key_at(pos); }
  uint32_t elem_size() const { return op_.elem_size(); }
  size_t count() const { return op_.count(); }
  //! IVFSearcher::search_impl / search_bf_impl (ivf_searcher.cc:106-250)
  int search(const void *q, const IndexQueryMeta &qm, uint32_t count, HipContext *ctx, bool brute_force) const {
    if (!q || qm.element_size() != op_.elem_size()) return IndexError_InvalidArgument;
    return op_.search(q, count, ctx, brute_force, ctx->scan_ratio_, ctx->bruteforce_threshold_);
  }

 private:
  HipIVFOp op_;
  IndexReformer::Pointer creformer_;      // the centroid index's reformer when it lives in a converted space (MIPS), else null
};

/*! "HipIVFSearcher": stands where IVFSearcher is registered (ivf_searcher.cc). */
class HipIVFSearcher : public IndexSearcher {
 public:
  int init(const ailego::Params &params) override {
    params_ = params;
    params.get(kParamHipDevice, &device_);
    params.get(kParamHipDeviceCount, &ndev_);
    if (ndev_ == 0) ndev_ = 1;
    return 0;
  }
  int cleanup() override { return this->unload(); }
  int load(IndexStorage::Pointer stg, IndexMetric::Pointer /*metric*/) override {
    if (!stg) return IndexError_InvalidArgument;
    int rc = core_.load(stg.get(), &meta_, device_, ndev_, params_);
    if (rc != 0) return rc;
    magic_ = IndexContext::GenerateMagic();
    stats_.set_loaded_count(core_.count());
    return 0;
  }
  int unload() override { core_.destroy(); return 0; }
  const Stats &stats() const override { return stats_; }
  const IndexMeta &meta() const override { return meta_; }
  const ailego::Params &params() const override { return params_; }
  Context::Pointer create_context() const override {
    auto *ctx = new HipContext(device_, magic_);
    ctx->update(params_);
    return Context::Pointer(ctx);
  }
  ZVEC_HIP_VECTOR_ACCESSORS(HipIVFCore, "HipIVFSearcher")
  int search_impl(const void *q, const IndexQueryMeta &qm, Context::Pointer &c) const override { return search_impl(q, qm, 1, c); }
  int search_impl(const void *q, const IndexQueryMeta &qm, uint32_t count, Context::Pointer &c) const override {
    HipContext *ctx = bind(c, magic_);
    return ctx ? core_.search(q, qm, count, ctx, false) : (int)IndexError_InvalidArgument;
  }
  int search_bf_impl(const void *q, const IndexQueryMeta &qm, Context::Pointer &c) const override { return search_bf_impl(q, qm, 1, c); }
  int search_bf_impl(const void *q, const IndexQueryMeta &qm, uint32_t count, Context::Pointer &c) const override {
    HipContext *ctx = bind(c, magic_);
    return ctx ? core_.search(q, qm, count, ctx, true) : (int)IndexError_InvalidArgument;
  }

 private:
  IndexMeta meta_;
  ailego::Params params_;
  Stats stats_;
  int device_{0};
  uint32_t ndev_{1};
  uint32_t magic_{0};
  HipIVFCore core_;
};

/*! "HipIVFStreamer": what the product instantiates (indexes/ivf_index.cc:38-39).  Like the reference's IVFStreamer
 *  (ivf_streamer.h:28-85) it only opens, searches and closes a dumped index: no add_impl. */
class HipIVFStreamer : public IndexStreamer {
 public:
  int init(const IndexMeta &meta, const ailego::Params &params) override {
    meta_ = meta;
    params_ = params;
    params.get(kParamHipDevice, &device_);
    params.get(kParamHipDeviceCount, &ndev_);
    if (ndev_ == 0) ndev_ = 1;
    return 0;
  }
  int cleanup() override { core_.destroy(); return 0; }
  int open(IndexStorage::Pointer stg) override {
    if (!stg) return IndexError_InvalidArgument;
    int rc = core_.load(stg.get(), &meta_, device_, ndev_, params_);
    if (rc != 0) return rc;
    magic_ = IndexContext::GenerateMagic();
    stats_.set_loaded_count(core_.count());
    return 0;
  }
  int flush(uint64_t /*check_point*/) override { return 0; }      // immutable: nothing to persist
  int close() override { core_.destroy(); return 0; }
  const IndexMeta &meta() const override { return meta_; }
  const Stats &stats() const override { return stats_; }
  Context::Pointer create_context() const override {
    auto *ctx = new HipContext(device_, magic_);
    ctx->update(params_);
    return Context::Pointer(ctx);
  }
  ZVEC_HIP_VECTOR_ACCESSORS(HipIVFCore, "HipIVFStreamer")
  int search_impl(const void *q, const IndexQueryMeta &qm, Context::Pointer &c) const override { return search_impl(q, qm, 1, c); }
  int search_impl(const void *q, const IndexQueryMeta &qm, uint32_t count, Context::Pointer &c) const override {
    HipContext *ctx = bind(c, magic_);
    return ctx ? core_.search(q, qm, count, ctx, false) : (int)IndexError_InvalidArgument;
  }
  int search_bf_impl(const void *q, const IndexQueryMeta &qm, Context::Pointer &c) const override { return search_bf_impl(q, qm, 1, c); }
  int search_bf_impl(const void *q, const IndexQueryMeta &qm, uint32_t count, Context::Pointer &c) const override {
    HipContext *ctx = bind(c, magic_);
    return ctx ? core_.search(q, qm, count, ctx, true) : (int)IndexError_InvalidArgument;
  }

 private:
  IndexMeta meta_;
  ailego::Params params_;
  Stats stats_;
  int device_{0};
  uint32_t ndev_{1};
  uint32_t magic_{0};
  HipIVFCore core_;
};

static_assert(!std::is_abstract<HipContext>::value && !std::is_abstract<HipFlatStreamer>::value &&
                  !std::is_abstract<HipFlatSearcher>::value && !std::is_abstract<HipIVFSearcher>::value &&
                  !std::is_abstract<HipIVFStreamer>::value,
              "every pure virtual of IndexContext / IndexStreamer / IndexSearcher / IndexRunner is implemented");

// New names keep the CPU classes selectable; registering under "FlatStreamer" / "IVFStreamer" instead would shadow
// them (the factory map insert overwrites, src/include/zvec/ailego/pattern/factory.h:120-122).
INDEX_FACTORY_REGISTER_STREAMER_ALIAS(HipFlatStreamer, HipFlatStreamer);
INDEX_FACTORY_REGISTER_SEARCHER_ALIAS(HipFlatSearcher, HipFlatSearcher);
INDEX_FACTORY_REGISTER_SEARCHER_ALIAS(HipIVFSearcher, HipIVFSearcher);
INDEX_FACTORY_REGISTER_STREAMER_ALIAS(HipIVFStreamer, HipIVFStreamer);

}  // namespace core
}  // namespace zvec
