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

namespace zvec {
namespace core {

namespace {

// parameter keys of the operators this plugin stands in for
const std::string kParamBruteForceThreshold("proxima.ivf.searcher.brute_force_threshold");  // ivf_params.h:46-47
const std::string kParamHipDeviceCount("proxima.hip.device_count");   // new: > 1 = shard the index over devices [device, device + count)

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

/*! Reader/writer lock that cannot starve the writer (std::shared_mutex on glibc prefers readers; searches that overlap
 *  continuously would keep add_impl waiting): everybody passes a gate, a writer keeps it while the readers drain. */
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

/*! Search context: one per caller thread (index.cc:24-45 caches it thread-local per index type and may hand it to
 *  another index instance of that type => magic re-binding, flat_streamer.cc:319-321 / ivf_streamer.cc:198-202).
 *  The HIP side (stream + workspace, zvec_hip_ctx_t) is index-agnostic, so re-binding is only bookkeeping. */
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

  //! key/score arrays of one batched call -> per-query IndexDocumentList (topk_to_result: lists end at the
  //! RNN threshold, which the device gate already applied)
  void fill(uint32_t count, uint32_t k, const uint64_t *keys, const float *scores, const uint32_t *n) {
    results_.assign(count, IndexDocumentList());
    for (uint32_t q = 0; q < count; ++q) {
      results_[q].reserve(n[q]);
      for (uint32_t j = 0; j < n[q]; ++j) results_[q].emplace_back(keys[size_t(q) * k + j], scores[size_t(q) * k + j]);
    }
  }

  zvec_hip_ctx_t h_{nullptr};
  int rc_{0};
  uint32_t topk_{0}, magic_{0};
  bool fetch_vector_{false};
  float scan_ratio_{0.1f};                  // ivf_searcher_context.h:211-213 defaults
  uint32_t bruteforce_threshold_{1000};
  std::vector<IndexDocumentList> results_{1};
  uint32_t group_num_{0}, group_topk_{0};
  std::vector<IndexGroupDocumentList> group_results_{1};
  std::vector<uint32_t> group_of_, groups_, ngroups_;   // group number of every storage position; picked groups per query
  std::vector<std::string> group_ids_;                  // group number -> the caller's id
  std::vector<uint64_t> bits_, keys_;
  std::vector<float> scores_;
  std::vector<uint32_t> counts_;
  std::string vectors_;                     // fetch_vector payload the documents point into (valid until the next search)
  IndexContext::Pointer inner_;             // HipFlatStreamer: a context of the wrapped reference streamer (its add paths cast-check one)
};

namespace {

//! H4: IndexFilter is an opaque std::function<bool(uint64_t)> (index_filter.h:48-50, true = exclude); swept once over
//! the keys in storage order into the 1-bit-per-position exclude set the scan kernels gate on
const uint64_t *sweep_filter(HipContext *ctx, const uint64_t *keys, size_t n) {
  if (!ctx->filter().is_valid()) return nullptr;
  ctx->bits_.assign((n + 63) / 64, 0);
  for (size_t i = 0; i < n; ++i)
    if (ctx->filter()(keys[i])) ctx->bits_[i >> 6] |= 1ull << (i & 63);
  return ctx->bits_.data();
}

//! IndexGroupBy is an opaque std::function<std::string(uint64_t)> too: swept once over the keys into dense group numbers
void sweep_groups(HipContext *ctx, const uint64_t *keys, size_t n) {
  std::unordered_map<std::string, uint32_t> number_of;
  ctx->group_ids_.clear();
  ctx->group_of_.resize(n);
  for (size_t i = 0; i < n; ++i) {
    std::string id = ctx->group_by()(keys[i]);
    auto it = number_of.find(id);
    if (it == number_of.end()) {
      it = number_of.emplace(id, (uint32_t)ctx->group_ids_.size()).first;
      ctx->group_ids_.push_back(std::move(id));
    }
    ctx->group_of_[i] = it->second;
  }
}

HipContext *bind(IndexContext::Pointer &c, uint32_t magic) {
  auto *ctx = dynamic_cast<HipContext *>(c.get());
  if (!ctx || ctx->rc_ != 0 || (ctx->topk() == 0 && !ctx->group_by_search())) return nullptr;   // "Invalid context or topk not set yet"
  if (ctx->magic_ != magic) { ctx->magic_ = magic; ctx->reset(); }
  return ctx;
}

void size_outputs(HipContext *ctx, uint32_t count) {
  const size_t k = ctx->topk();
  ctx->keys_.resize(size_t(count) * k);
  ctx->scores_.resize(size_t(count) * k);
  ctx->counts_.resize(count);
}

}  // namespace

// =====================================================================================================================
// flat, shared by the searcher and the streamer
// =====================================================================================================================
class HipFlatCore {
 public:
  ~HipFlatCore() { destroy(); }
  //! ndev > 1: one zvec_hip_shards_t (a row-range shard, a worker thread and a stream per device) instead of one handle
  int create(const IndexMeta &meta, int device, uint32_t ndev = 1) {
    destroy();
    const int metric = metric_of(meta), dtype = dtype_of(meta);
    if (metric < 0 || dtype < 0) return IndexError_Unsupported;
    device_ = device;
    elem_size_ = meta.element_size();
    if (ndev > 1) {
      std::vector<int> devs(ndev);
      for (uint32_t g = 0; g < ndev; ++g) devs[g] = device + (int)g;
      return zvec_hip_shards_create(meta.dimension(), dtype, metric, ZVEC_HIP_SHARDS_FLAT, devs.data(), ndev, &sh_);
    }
    return zvec_hip_flat_create(meta.dimension(), dtype, metric, device, &h_);
  }
  void destroy() {
    if (h_) zvec_hip_flat_destroy(h_);
    if (sh_) zvec_hip_shards_destroy(sh_);
    h_ = nullptr;
    sh_ = nullptr;
    keys_.clear();
    pos_of_key_.clear();
  }
  int append(const void *rows, size_t n, const uint64_t *keys) {
    std::unique_lock<FairSharedMutex> w(mu_);
    int rc = sh_ ? zvec_hip_shards_flat_append(sh_, rows, n, keys) : zvec_hip_flat_append(h_, rows, n, keys);
    if (rc != 0) return rc;
    for (size_t i = 0; i < n; ++i) {
      pos_of_key_.emplace(keys[i], (uint32_t)keys_.size());
      keys_.push_back(keys[i]);
    }
    return 0;
  }
  //! group_by_search_impl / group_by_search_p_keys_impl (flat_streamer.cc:391-483): ids == nullptr scans every row
  int group_search(const void *q, uint32_t count, HipContext *ctx, const uint32_t *ids, const uint32_t *offs) const {
    if (!ctx->group_by().is_valid()) return IndexError_InvalidArgument;      // "Invalid group-by function"
    if (sh_) return IndexError_Unsupported;                                   // (group-by runs on one device)
    const uint32_t gnum = ctx->group_num_, gk = ctx->group_topk_;
    if (gk == 0) return IndexError_InvalidArgument;
    sweep_groups(ctx, keys_.data(), keys_.size());
    const uint64_t *bits = sweep_filter(ctx, keys_.data(), keys_.size());
    const size_t rows = size_t(count) * gnum;
    ctx->keys_.resize(rows * gk);
    ctx->scores_.resize(rows * gk);
    ctx->counts_.resize(rows);
    ctx->groups_.resize(rows);
    ctx->ngroups_.resize(count);
    const uint32_t ngroups = std::max<uint32_t>(1u, (uint32_t)ctx->group_ids_.size());
    const uint32_t none = 0;
    const uint32_t *gof = ctx->group_of_.empty() ? &none : ctx->group_of_.data();
    int rc = ids ? zvec_hip_flat_search_grouped_by_ids(h_, ctx->h_, q, count, ids, offs, gof, ngroups, gnum, gk, ctx->threshold(), bits,
                                                       ctx->groups_.data(), ctx->ngroups_.data(), ctx->keys_.data(),
                                                       ctx->scores_.data(), ctx->counts_.data())
                 : zvec_hip_flat_search_grouped(h_, ctx->h_, q, count, gof, ngroups, gnum, gk, ctx->threshold(), bits,
                                                ctx->groups_.data(), ctx->ngroups_.data(), ctx->keys_.data(), ctx->scores_.data(),
                                                ctx->counts_.data());
    if (rc != 0) return rc;
    // topk_to_group_result (flat_streamer_context.h:135-180)
    std::vector<uint64_t> pos;
    ctx->group_results_.assign(count, IndexGroupDocumentList());
    for (uint32_t qi = 0; qi < count; ++qi) {
      ctx->group_results_[qi].resize(ctx->ngroups_[qi]);
      for (uint32_t s = 0; s < ctx->ngroups_[qi]; ++s) {
        const size_t row = size_t(qi) * gnum + s;
        GroupIndexDocument &g = ctx->group_results_[qi][s];
        g.set_group_id(ctx->group_ids_[ctx->groups_[row]]);
        for (uint32_t j = 0; j < ctx->counts_[row]; ++j) {
          g.mutable_docs()->emplace_back(ctx->keys_[row * gk + j], ctx->scores_[row * gk + j]);
          if (ctx->fetch_vector()) pos.push_back(pos_of_key_.at(ctx->keys_[row * gk + j]));
        }
      }
    }
    if (!ctx->fetch_vector() || pos.empty()) return 0;
    ctx->vectors_.resize(pos.size() * elem_size_);
    if ((rc = zvec_hip_flat_get_vectors(h_, pos.data(), pos.size(), &ctx->vectors_[0])) != 0) return rc;
    size_t j = 0;
    for (auto &lst : ctx->group_results_)
      for (auto &g : lst)
        for (auto &d : *g.mutable_docs()) {
          d = IndexDocument(d.key(), d.score(), (uint32_t)pos[j], ctx->vectors_.data() + j * elem_size_);
          ++j;
        }
    return 0;
  }
  //! add_with_id_impl (FlatStreamerEntity::add_vector_with_id, flat_streamer_entity.cc:900-990): the document `id` is
  //! replaced when it exists, appended otherwise.  Results carry keys, never positions, so the mirror keeps its own
  //! id -> position map instead of the reference's "position == id" rule (no holes to pad, and a re-opened index whose
  //! rows were compacted by the provider walk stays addressable)
  int put(uint32_t id, const void *row) {
    std::unique_lock<FairSharedMutex> w(mu_);
    const uint64_t key = id;
    auto it = pos_of_key_.find(key);
    if (it == pos_of_key_.end()) {
      int rc = sh_ ? zvec_hip_shards_flat_append(sh_, row, 1, &key) : zvec_hip_flat_append(h_, row, 1, &key);
      if (rc != 0) return rc;
      pos_of_key_.emplace(key, (uint32_t)keys_.size());
      keys_.push_back(key);
      return 0;
    }
    if (sh_) return IndexError_Unsupported;                   // (in-place replacement runs on one device)
    const uint32_t pos = it->second;
    return zvec_hip_flat_put(h_, &pos, 1, row, &key);
  }
  int search(const void *q, const IndexQueryMeta &qm, uint32_t count, HipContext *ctx) const {
    if (!q || qm.element_size() != elem_size_) return IndexError_InvalidArgument;
    std::shared_lock<FairSharedMutex> r(mu_);
    if (ctx->group_by_search()) return group_search(q, count, ctx, nullptr, nullptr);   // flat_streamer.cc:323-324
    size_outputs(ctx, count);
    const uint64_t *bits = sweep_filter(ctx, keys_.data(), keys_.size());
    int rc = sh_ ? zvec_hip_shards_search(sh_, q, count, ctx->topk(), ctx->threshold(), 0, 0, bits, ctx->keys_.data(),
                                          ctx->scores_.data(), ctx->counts_.data())
                 : zvec_hip_flat_search(h_, ctx->h_, q, count, ctx->topk(), ctx->threshold(), bits, ctx->keys_.data(),
                                        ctx->scores_.data(), ctx->counts_.data());
    if (rc != 0) return rc;
    ctx->fill(count, ctx->topk(), ctx->keys_.data(), ctx->scores_.data(), ctx->counts_.data());
    return attach_vectors(ctx, count);
  }
  //! search_bf_by_p_keys_impl (flat_streamer.cc:346-389): unknown keys are skipped, as get_vector_by_key != 0 -> continue
  int search_by_keys(const void *q, const std::vector<std::vector<uint64_t>> &p_keys, const IndexQueryMeta &qm,
                     uint32_t count, HipContext *ctx) const {
    if (!q || qm.element_size() != elem_size_ || p_keys.size() != count) return IndexError_InvalidArgument;
    std::shared_lock<FairSharedMutex> r(mu_);
    std::vector<uint32_t> ids, offs(count + 1, 0);
    for (uint32_t i = 0; i < count; ++i) {
      for (uint64_t key : p_keys[i]) {
        auto it = pos_of_key_.find(key);
        if (it != pos_of_key_.end()) ids.push_back(it->second);
      }
      offs[i + 1] = (uint32_t)ids.size();
    }
    if (ids.empty()) ids.push_back(0);
    if (ctx->group_by_search()) return group_search(q, count, ctx, ids.data(), offs.data());   // flat_streamer.cc:365-366
    size_outputs(ctx, count);
    const uint64_t *bits = sweep_filter(ctx, keys_.data(), keys_.size());
    int rc;
    if (sh_) {
      std::vector<uint64_t> wide(ids.begin(), ids.end());
      rc = zvec_hip_shards_flat_search_by_ids(sh_, q, count, wide.data(), offs.data(), ctx->topk(), ctx->threshold(), bits,
                                              ctx->keys_.data(), ctx->scores_.data(), ctx->counts_.data());
    } else {
      rc = zvec_hip_flat_search_by_ids(h_, ctx->h_, q, count, ids.data(), offs.data(), ctx->topk(), ctx->threshold(), bits,
                                       ctx->keys_.data(), ctx->scores_.data(), ctx->counts_.data());
    }
    if (rc != 0) return rc;
    ctx->fill(count, ctx->topk(), ctx->keys_.data(), ctx->scores_.data(), ctx->counts_.data());
    return attach_vectors(ctx, count);
  }
  int vector_of_key(uint64_t key, void *out) const {
    std::shared_lock<FairSharedMutex> r(mu_);
    auto it = pos_of_key_.find(key);
    return it == pos_of_key_.end() ? (int)IndexError_NoExist : vector_of_pos(it->second, out);
  }
  int vector_of_pos(uint32_t pos, void *out) const {
    const uint64_t p = pos;
    return sh_ ? zvec_hip_shards_flat_get_vectors(sh_, &p, 1, out) : zvec_hip_flat_get_vector(h_, pos, out);
  }
  //! FlatSearcher::load's "flat.features" payload -> HBM (one device, or dealt over the shards)
  int load_features(const void *features, size_t bytes, size_t n, bool column_major, const uint64_t *keys) {
    return sh_ ? zvec_hip_shards_flat_load_features(sh_, features, bytes, n, column_major, 32, keys)
               : zvec_hip_flat_load_features(h_, features, bytes, n, column_major, 32, keys);
  }
  //! a run of FlatStreamerEntity blocks (one "flat.features<i>" segment) -> HBM: one strided copy + one pack launch
  int load_blocks(const void *blocks, size_t nblocks, uint32_t block_size, uint32_t bvc, const std::vector<uint32_t> &keep) {
    std::unique_lock<FairSharedMutex> w(mu_);
    if (sh_) {            // sharded mirror: rows dealt by the shards' own append (no strided path there)
      const char *p = static_cast<const char *>(blocks);
      for (size_t b = 0; b < nblocks; ++b)
        for (uint32_t m = keep[b]; m; m &= m - 1) {
          const uint32_t r = (uint32_t)__builtin_ctz(m);
          uint64_t key;
          memcpy(&key, p + b * block_size + (size_t)bvc * elem_size_ + (size_t)r * 8, 8);
          int rc = zvec_hip_shards_flat_append(sh_, p + b * block_size + (size_t)r * elem_size_, 1, &key);
          if (rc != 0) return rc;
          pos_of_key_.emplace(key, (uint32_t)keys_.size());
          keys_.push_back(key);
        }
      return 0;
    }
    int rc = zvec_hip_flat_load_blocks(h_, blocks, nblocks, block_size, bvc, keep.data());
    if (rc != 0) return rc;
    const char *p = static_cast<const char *>(blocks);
    for (size_t b = 0; b < nblocks; ++b)
      for (uint32_t m = keep[b]; m; m &= m - 1) {
        const uint32_t r = (uint32_t)__builtin_ctz(m);
        uint64_t key;
        memcpy(&key, p + b * block_size + (size_t)bvc * elem_size_ + (size_t)r * 8, 8);
        pos_of_key_.emplace(key, (uint32_t)keys_.size());
        keys_.push_back(key);
      }
    return 0;
  }
  size_t count() const { return keys_.size(); }
  uint64_t key_at(size_t pos) const { return keys_[pos]; }
  uint32_t elem_size() const { return elem_size_; }
  //! rows of storage positions [pos0, pos0 + n) -> host (the provider's iterator)
  int rows_at(uint64_t pos0, size_t n, void *out) const {
    std::vector<uint64_t> pos(n);
    for (size_t i = 0; i < n; ++i) pos[i] = pos0 + i;
    return sh_ ? zvec_hip_shards_flat_get_vectors(sh_, pos.data(), n, out) : zvec_hip_flat_get_vectors(h_, pos.data(), n, out);
  }
  void adopt_keys(const uint64_t *keys, size_t n) {
    keys_.assign(keys, keys + n);
    for (size_t i = 0; i < n; ++i) pos_of_key_.emplace(keys[i], (uint32_t)i);
  }

 private:
  //! IndexContext::set_fetch_vector (index.cc:635-647): the stored vectors of the result documents, one gather launch
  int attach_vectors(HipContext *ctx, uint32_t count) const {
    if (!ctx->fetch_vector()) return 0;
    std::vector<uint64_t> pos;
    for (uint32_t q = 0; q < count; ++q)
      for (auto &d : ctx->results_[q]) pos.push_back(pos_of_key_.at(d.key()));
    ctx->vectors_.resize(pos.size() * elem_size_);
    if (pos.empty()) return 0;
    int rc = sh_ ? zvec_hip_shards_flat_get_vectors(sh_, pos.data(), pos.size(), &ctx->vectors_[0])
                 : zvec_hip_flat_get_vectors(h_, pos.data(), pos.size(), &ctx->vectors_[0]);
    if (rc != 0) return rc;
    size_t j = 0;
    for (uint32_t q = 0; q < count; ++q)
      for (auto &d : ctx->results_[q]) {
        d = IndexDocument(d.key(), d.score(), (uint32_t)pos[j], ctx->vectors_.data() + j * elem_size_);
        ++j;
      }
    return 0;
  }

  zvec_hip_flat_t h_{nullptr};
  zvec_hip_shards_t sh_{nullptr};         // set instead of h_ when the index is sharded over several devices
  int device_{0};
  uint32_t elem_size_{0};
  mutable FairSharedMutex mu_;            // add (exclusive) vs search (shared): flat_streamer.cc:236-242
  std::vector<uint64_t> keys_;            // key of every storage position
  std::unordered_map<uint64_t, uint32_t> pos_of_key_;
};

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
    if ((rc = core_.create(meta_, device_, ndev_)) != 0) return rc;
    rc = core_.load_features(features.data(), features.size(), n, meta_.major_order() == IndexMeta::MO_COLUMN,
                             reinterpret_cast<const uint64_t *>(keys.data()));
    if (rc != 0) return rc;
    core_.adopt_keys(reinterpret_cast<const uint64_t *>(keys.data()), n);
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
    return ctx ? core_.search(q, qm, count, ctx) : (int)IndexError_InvalidArgument;
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
    return ctx ? core_.search_by_keys(q, p_keys, qm, count, ctx) : (int)IndexError_InvalidArgument;
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
    if ((rc = core_.create(meta_, device_, ndev_)) != 0) return rc;
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
    return ctx ? core_.search(q, qm, count, ctx) : (int)IndexError_InvalidArgument;
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
    return ctx ? core_.search_by_keys(q, p_keys, qm, count, ctx) : (int)IndexError_InvalidArgument;
  }

 private:
  IndexMeta meta_;
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
      int rc = core_.load_blocks(p, nblk, bs, bvc, keep);
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
class HipIVFCore {
 public:
  ~HipIVFCore() { destroy(); }
  void destroy() {
    if (h_) zvec_hip_ivf_destroy(h_);
    if (sh_) zvec_hip_shards_destroy(sh_);
    h_ = nullptr;
    sh_ = nullptr;
    keys_.clear();
  }
  //! IVFSearcher::load (ivf_searcher.cc:43-103) + IVFEntity::load (ivf_entity.cc:443-570): the centroid index is a
  //! nested flat index inside the "ivf.centroid" segment; the inverted lists come as header / meta / body / keys
  //! ndev > 1: whole inverted lists dealt over the devices (byte-balanced map), centroids replicated
  int load(IndexStorage *stg, IndexMeta *meta, int device, uint32_t ndev = 1) {
    destroy();
    int rc = IndexHelper::DeserializeFromStorage(stg, meta);
    if (rc != 0) return rc;
    const int metric = metric_of(*meta), dtype = dtype_of(*meta);
    if (metric < 0 || dtype < 0) return IndexError_Unsupported;
    elem_size_ = meta->element_size();
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
    creformer_.reset();
    if (cmeta.data_type() != meta->data_type()) return IndexError_Unsupported;
    if (!cmeta.reformer_name().empty()) {
      creformer_ = IndexFactory::CreateReformer(cmeta.reformer_name());
      if (!creformer_) return IndexError_NoExist;
      if ((rc = creformer_->init(cmeta.reformer_params())) != 0) return rc;
    } else if (celem != elem_size_) {
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
    if (creformer_ && ndev > 1) return IndexError_Unsupported;                 // (the coarse space runs on one device)
    in_qmeta_ = IndexQueryMeta(meta->data_type(), meta->dimension());
    std::string header, lmeta, body, keys;
    if ((rc = read_segment(stg, kIvfHeader, &header)) != 0) return rc;
    if ((rc = read_segment(stg, kIvfMeta, &lmeta)) != 0) return rc;
    if ((rc = read_segment(stg, kIvfBody, &body)) != 0) return rc;
    if ((rc = read_segment(stg, kIvfKeys, &keys)) != 0) return rc;
    if (ndev > 1) {
      std::vector<int> devs(ndev);
      for (uint32_t g = 0; g < ndev; ++g) devs[g] = device + (int)g;
      if ((rc = zvec_hip_shards_create(meta->dimension(), dtype, metric, ZVEC_HIP_SHARDS_IVF, devs.data(), ndev, &sh_)) != 0) return rc;
      rc = zvec_hip_shards_ivf_load_segments(sh_, header.data(), header.size(), lmeta.data(), lmeta.size(), body.data(),
                                             body.size(), keys.data(), keys.size(), centroids.data());
    } else {
      if ((rc = zvec_hip_ivf_create(meta->dimension(), dtype, metric, device, &h_)) != 0) return rc;
      rc = zvec_hip_ivf_load_segments(h_, header.data(), header.size(), lmeta.data(), lmeta.size(), body.data(), body.size(),
                                      keys.data(), keys.size(), creformer_ ? nullptr : centroids.data());
      if (rc == 0 && creformer_)
        rc = zvec_hip_ivf_set_coarse_space(h_, cmeta.dimension(), cmetric, centroids.data(), (uint32_t)nlist);
    }
    if (rc != 0) return rc;
    keys_.assign(reinterpret_cast<const uint64_t *>(keys.data()),
                 reinterpret_cast<const uint64_t *>(keys.data()) + keys.size() / sizeof(uint64_t));
    pos_of_key_.clear();
    pos_of_key_.reserve(keys_.size());
    for (size_t i = 0; i < keys_.size(); ++i) pos_of_key_.emplace(keys_[i], i);
    nlist_ = (uint32_t)nlist;
    return 0;
  }
  //! IVFEntity::get_vector_by_key: the stored row of a document (list-order position through the key map)
  int vector_of_key(uint64_t key, void *out) const {
    auto it = pos_of_key_.find(key);
    if (it == pos_of_key_.end()) return IndexError_NoExist;
    if (sh_) return IndexError_Unsupported;                  // (row fetches run on one device)
    return zvec_hip_ivf_get_vector(h_, it->second, out);
  }
  //! rows of list-order positions [pos0, pos0 + n) -> host (the provider's iterator walks the index in chunks)
  int rows_at(uint64_t pos0, size_t n, void *out) const {
    if (sh_) return IndexError_Unsupported;
    std::vector<uint64_t> pos(n);
    for (size_t i = 0; i < n; ++i) pos[i] = pos0 + i;
    return zvec_hip_ivf_get_vectors(h_, pos.data(), n, out);
  }
  uint64_t key_at(size_t pos) const { return keys_[pos]; }
  uint32_t elem_size() const { return elem_size_; }
  //! IVFSearcher::search_impl / search_bf_impl (ivf_searcher.cc:106-250)
  int search(const void *q, const IndexQueryMeta &qm, uint32_t count, HipContext *ctx, bool brute_force) const {
    if (!q || qm.element_size() != elem_size_) return IndexError_InvalidArgument;
    size_outputs(ctx, count);
    const uint64_t *bits = sweep_filter(ctx, keys_.data(), keys_.size());    // keys in list order (ivf_entity.cc:612)
    int rc;
    if (brute_force || keys_.size() <= ctx->bruteforce_threshold_) {         // ivf_searcher.cc:188-190
      rc = sh_ ? zvec_hip_shards_search(sh_, q, count, ctx->topk(), ctx->threshold(), nlist_, 0xffffffffu, bits,     // every list
                                        ctx->keys_.data(), ctx->scores_.data(), ctx->counts_.data())
               : zvec_hip_ivf_search_bf(h_, ctx->h_, q, count, ctx->topk(), ctx->threshold(), bits, ctx->keys_.data(),
                                        ctx->scores_.data(), ctx->counts_.data());
    } else {
      // IVFSearcherContext::update (ivf_searcher_context.h:70-78): float arithmetic, std::round / std::ceil
      const uint32_t nprobe = std::max(static_cast<uint32_t>(std::round(nlist_ * ctx->scan_ratio_)), 1u);
      uint32_t max_scan = static_cast<uint32_t>(std::ceil(keys_.size() * ctx->scan_ratio_));
      max_scan = std::max(ctx->bruteforce_threshold_, max_scan);
      if (creformer_) {
        // IVFCentroidIndex::search: the queries reformed for the coarse space — in one call where the reformer has a batched
        // transform, one by one where it has not
        std::string cq;
        IndexQueryMeta ometa;
        rc = creformer_->transform(q, in_qmeta_, count, &cq, &ometa);
        if (rc == IndexError_Unsupported || rc == IndexError_NotImplemented) {
          cq.clear();
          for (uint32_t i = 0; i < count; ++i) {
            std::string one;
            if ((rc = creformer_->transform(static_cast<const char *>(q) + size_t(i) * elem_size_, in_qmeta_, &one, &ometa)) != 0) return rc;
            cq.append(one);
          }
        } else if (rc != 0) {
          return rc;
        }
        rc = zvec_hip_ivf_search_coarse(h_, ctx->h_, q, cq.data(), count, ctx->topk(), ctx->threshold(), nprobe, max_scan, bits,
                                        ctx->keys_.data(), ctx->scores_.data(), ctx->counts_.data());
      } else {
        rc = sh_ ? zvec_hip_shards_search(sh_, q, count, ctx->topk(), ctx->threshold(), nprobe, max_scan, bits, ctx->keys_.data(),
                                          ctx->scores_.data(), ctx->counts_.data())
                 : zvec_hip_ivf_search(h_, ctx->h_, q, count, ctx->topk(), ctx->threshold(), nprobe, max_scan, bits,
                                       ctx->keys_.data(), ctx->scores_.data(), ctx->counts_.data());
      }
    }
    if (rc != 0) return rc;
    ctx->fill(count, ctx->topk(), ctx->keys_.data(), ctx->scores_.data(), ctx->counts_.data());
    return attach_vectors(ctx, count);
  }
  size_t count() const { return keys_.size(); }
 private:
  //! fetch_vector (ivf_searcher_context.h:186-197: entity_->get_vector_by_key per result): one gather for the batch
  int attach_vectors(HipContext *ctx, uint32_t count) const {
    if (!ctx->fetch_vector()) return 0;
    if (sh_) return IndexError_Unsupported;
    std::vector<uint64_t> pos;
    for (uint32_t q = 0; q < count; ++q)
      for (auto &d : ctx->results_[q]) pos.push_back(pos_of_key_.at(d.key()));
    ctx->vectors_.resize(pos.size() * elem_size_);
    if (pos.empty()) return 0;
    int rc = zvec_hip_ivf_get_vectors(h_, pos.data(), pos.size(), &ctx->vectors_[0]);
    if (rc != 0) return rc;
    size_t j = 0;
    for (uint32_t q = 0; q < count; ++q)
      for (auto &d : ctx->results_[q]) {
        d = IndexDocument(d.key(), d.score(), (uint32_t)d.key(), ctx->vectors_.data() + j * elem_size_);
        ++j;
      }
    return 0;
  }

  zvec_hip_ivf_t h_{nullptr};
  IndexReformer::Pointer creformer_;      // the centroid index's reformer when it lives in a converted space (MIPS), else null
  IndexQueryMeta in_qmeta_;               // the queries as they arrive
  zvec_hip_shards_t sh_{nullptr};         // set instead of h_ when the lists are dealt over several devices
  std::unordered_map<uint64_t, uint64_t> pos_of_key_;   // key -> list-order position
  uint32_t elem_size_{0}, nlist_{0};
  std::vector<uint64_t> keys_;
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
    int rc = core_.load(stg.get(), &meta_, device_, ndev_);
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
    int rc = core_.load(stg.get(), &meta_, device_, ndev_);
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
