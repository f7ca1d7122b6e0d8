// zvec_hip_operator.hpp — the operator logic ABOVE the C ABI (include/zvec_hip.h), written ONCE.
//
// Two host sides sit on the C ABI: the real zvec binding (plugin/hip_plugin.cc — IndexStreamer / IndexSearcher subclasses
// registered with the reference's factory) and the dependency-free C++ mirror of the same operator surface
// (zvec_amd/csrc/host/hip_index.h — what tests/cpp and tools/cpp build against on a box that has no zvec checkout).  Everything
// the two have in common lives here, templated on the framework's types, and both include it: the filter / group-by sweeps,
// the probe-parameter arithmetic, the p_keys mapping, result and group-result assembly, fetch_vector, add_with_id, the flat and
// IVF search sequences themselves, and the micro-batcher that folds zvec's one-query-per-call callers into batches.  A defect
// in any of it can no longer live in one copy only (round 3 found three in the plugin that the mirror's green suite could not see).
//
// What the operators do is the reference's (file:line at each function): FlatStreamer / FlatSearcher
// (src/core/algorithm/flat/flat_streamer.cc:304-483, flat_searcher.cc:162-211), IVFSearcher / IVFStreamer
// (src/core/algorithm/ivf/ivf_searcher.cc:106-250, ivf_searcher_context.h:61-79), contexts (index_context.h:123-262).
//
// Requirements on the template arguments
//   Ctx  — the caller's search context.  Non-virtual accessors, same names on both sides:
//            zvec_hip_ctx_t hip() const;            uint32_t topk() const;      float threshold() const;   bool fetch_vector() const;
//            bool op_has_filter() const;            bool op_filtered(uint64_t key) const;         // IndexFilter: true = EXCLUDE
//            bool op_has_group_by() const;          std::string op_group_of(uint64_t key) const;  // IndexGroupBy
//            uint32_t op_group_num() const;         uint32_t op_group_topk() const;
//            const uint64_t *op_preset_bits() const;                 // SURVEY H4 side channel: a materialised exclude set, or nullptr
//            const zvec_hip_doc_filter_t *op_doc_filter() const;     // the composite document filter as data, or nullptr
//            zvec_hip_op::Scratch &op_scratch();
//            std::vector<DT::DocumentList> &op_results();            std::vector<DT::GroupDocumentList> &op_group_results();
//   DT   — document traits: Document / DocumentList / GroupDocument / GroupDocumentList,
//            static Document make(uint64_t key, float score);
//            static void attach(Document &d, uint32_t index, const char *row, size_t bytes);   // fetch_vector
//          GroupDocument offers set_group_id(std::string) and mutable_docs() on both sides (index_document.h:278-314).
#pragma once
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "zvec_hip.h"

namespace zvec_hip_op {

constexpr uint64_t kInvalidKey = ~0ull;     // flat_index_format.h:29: the key of an add_with_id gap / a deleted slot

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

//! IVFSearcherContext::update (ivf_searcher_context.h:70-78): float arithmetic, std::round / std::ceil
struct ProbeParams { uint32_t nprobe, max_scan; };
inline ProbeParams probe_params(uint32_t nlist, uint64_t rows, float scan_ratio, uint32_t bruteforce_threshold) {
  ProbeParams p;
  p.nprobe = std::max(static_cast<uint32_t>(std::round(nlist * scan_ratio)), 1u);
  p.max_scan = std::max(bruteforce_threshold, static_cast<uint32_t>(std::ceil(rows * scan_ratio)));
  return p;
}

//! per-context scratch of the operators (the result arrays of the C ABI call, the swept filter / groups, the fetch_vector
//! payload the documents of the plugin point into — valid until the next search on the context, like the reference's)
struct Scratch {
  std::vector<uint64_t> bits, keys, pos;
  std::vector<float> scores;
  std::vector<uint32_t> counts, group_of, groups, ngroups, ids, offs;
  std::vector<std::string> group_ids;
  std::string vectors;
};

/*! Keys of the storage positions and the way back.  The map is built on first use (an immutable 10M-row index that is never
 *  asked for a vector never pays for it) and kept up to date afterwards.  Readers run concurrently (searches are const and
 *  hold the operator's lock shared); append / set / adopt need the lock exclusive. */
class KeyDirectory {
 public:
  void clear() { keys_.clear(); map_.clear(); built_.store(false); }
  size_t size() const { return keys_.size(); }
  const uint64_t *data() const { return keys_.data(); }
  uint64_t at(size_t pos) const { return keys_[pos]; }
  void append(uint64_t key) {
    if (built_.load(std::memory_order_relaxed) && key != kInvalidKey) map_.emplace(key, keys_.size());
    keys_.push_back(key);
  }
  //! position `pos` now carries `key`; positions skipped on the way are holes
  void set(uint64_t pos, uint64_t key) {
    if (keys_.size() <= pos) keys_.resize(pos + 1, kInvalidKey);
    if (built_.load(std::memory_order_relaxed)) {
      auto old = map_.find(keys_[pos]);
      if (old != map_.end() && old->second == pos) map_.erase(old);
      map_[key] = pos;
    }
    keys_[pos] = key;
  }
  void adopt(const uint64_t *keys, size_t n) {
    keys_.assign(keys, keys + n);
    map_.clear();
    built_.store(false);
  }
  void adopt_identity(size_t n) {
    keys_.resize(n);
    for (size_t i = 0; i < n; ++i) keys_[i] = i;
    map_.clear();
    built_.store(false);
  }
  bool find(uint64_t key, uint64_t *pos) const {
    ensure();
    auto it = map_.find(key);
    if (it == map_.end()) return false;
    *pos = it->second;
    return true;
  }

 private:
  void ensure() const {
    if (built_.load(std::memory_order_acquire)) return;
    std::lock_guard<std::mutex> g(mu_);
    if (built_.load(std::memory_order_relaxed)) return;
    map_.clear();
    map_.reserve(keys_.size());
    for (uint64_t i = 0; i < keys_.size(); ++i)
      if (keys_[i] != kInvalidKey) map_.emplace(keys_[i], i);        // (the first position of a key wins, as a scan would meet it first)
    built_.store(true, std::memory_order_release);
  }
  std::vector<uint64_t> keys_;
  mutable std::unordered_map<uint64_t, uint64_t> map_;
  mutable std::mutex mu_;
  mutable std::atomic<bool> built_{false};
};

// ---- micro-batcher ---------------------------------------------------------------------------------------------------
// zvec drives boundary B with ONE query per call from many threads, each with its own context (index.cc:24-45,605-619;
// tools/core/bench.cc:145-245), which leaves the GPU at a few thousand searches per second while one batched call answers
// 1024 queries in 5 ms.  The batcher turns the former into the latter behind the same single-query entry point: concurrent
// callers whose searches are interchangeable (same BatchKey: topk and probe parameters, no filter / radius / fetch_vector /
// group-by) join an open batch; the first one in leads it — keeps it open while an earlier batch is still searching (at most
// until it is full or `window_us` has passed), runs ONE batched search and hands every caller its own result list.  A lone
// caller on an idle index is not delayed (unless a linger is configured); under load the batches grow by themselves.
// Cost per member beyond the search: one short critical section to take a slot, a 3 KB copy outside it, and one wake-up —
// the members park on the batch's shared_mutex, which the leader holds exclusively from the batch's birth to its results:
// unlocking releases them all at once, with no mutex to re-acquire one after the other as a condition variable would.
struct BatchKey {
  uint32_t topk = 0, a = 0, b = 0, mode = 0;
  bool operator==(const BatchKey &o) const { return topk == o.topk && a == o.a && b == o.b && mode == o.mode; }
  bool operator!=(const BatchKey &o) const { return !(*this == o); }
};

template <class DT>
class MicroBatcher {
 public:
  using DocumentList = typename DT::DocumentList;
  //! runs a batched search of `count` queries (row-major, row_bytes each); fills keys / scores [count][topk] and counts
  using RunFn = std::function<int(const void *queries, uint32_t count, const BatchKey &key, std::vector<uint64_t> *keys,
                                  std::vector<float> *scores, std::vector<uint32_t> *counts)>;
  struct Stats { uint64_t batches = 0, queries = 0, largest = 0; };

  //! linger_us: how long a leader keeps its batch open even when nothing else is searching (0 = a lone caller is never delayed;
  //! a few tens of microseconds let callers that arrive in a burst share the first batch too)
  MicroBatcher(size_t row_bytes, uint32_t max_batch, uint32_t window_us, uint32_t linger_us, RunFn fn)
      : row_bytes_(row_bytes), max_batch_(std::max<uint32_t>(1, max_batch)), window_us_(window_us),
        linger_us_(std::min(linger_us, window_us)), fn_(std::move(fn)) {
    if (const char *e = getenv("ZVEC_HIP_BATCH_EARLY_GO")) early_go_ = atoi(e) != 0;
  }

  int search(const void *query, const BatchKey &key, DocumentList *out) {
    std::shared_ptr<Batch> b;
    bool leader = false;
    uint32_t slot = 0;
    {
      std::unique_lock<std::mutex> lk(mu_);
      // join the open batch if it takes this key and has room; otherwise wait for it to close and open a new one
      while (open_ && (open_->key != key || open_->n >= max_batch_)) cv_.wait(lk);
      b = open_;
      if (!b) {
        b = take_batch();
        b->key = key;
        b->n = 0;
        b->copied.store(0, std::memory_order_relaxed);
        b->rc = 0;
        b->gate.lock();                     // the members wait on this until the results are in
        const auto now = std::chrono::steady_clock::now();
        b->deadline = now + std::chrono::microseconds(window_us_);
        b->linger = now + std::chrono::microseconds(linger_us_);
        open_ = b;
        leader = true;
      }
      slot = b->n++;
      if (returning_ > 0) --returning_;
      if (b->n >= max_batch_ || returning_ == 0) cv_.notify_all();      // (full, or everybody a finished batch released is back: the leader may go)
    }
    memcpy(b->queries + (size_t)slot * row_bytes_, query, row_bytes_);        // outside the lock: 256 callers, 3 KB each
    b->copied.fetch_add(1, std::memory_order_release);
    if (leader) {
      uint32_t n;
      {
        std::unique_lock<std::mutex> lk(mu_);
        // collect while an earlier batch is still searching (no added latency on an idle index: a lone caller goes at once
        // unless a linger is configured), at most until the batch is full or the window has passed
        while (b->n < max_batch_) {
          const auto now = std::chrono::steady_clock::now();
          // Nothing searching: linger only if the previous batch had company (a lone caller in a steady state pays nothing), and
          // count the linger from the moment the device fell idle — the callers of the batch that has just ended are on their way
          // back, and a leader that left the instant it ended would take only those who had arrived meanwhile: closed-loop callers
          // then settle into two alternating half-size batches, each streaming most of the lists (256 callers, 10M x 768: 39 k
          // searches/s in two batches of ~128 against one of 256).
          auto until = b->deadline;
          if (inflight_ == 0) {
            if (early_go_ && returning_ == 0) break;     // every caller the finished batches released is back (closed-loop callers): nothing to linger for
            until = last_n_ > 1 ? std::min(b->deadline, std::max(b->linger, idle_since_ + std::chrono::microseconds(linger_us_))) : now;
          }
          if (now >= until) break;
          cv_.wait_until(lk, until);
        }
        open_.reset();                      // the next arrival opens (and leads) the next batch
        returning_ = 0;                     // (whoever has not come back by now is not awaited again)
        n = b->n;
        last_n_ = n;
        ++inflight_;
        stats_.batches++;
        stats_.queries += n;
        stats_.largest = std::max<uint64_t>(stats_.largest, n);
        cv_.notify_all();
      }
      while (b->copied.load(std::memory_order_acquire) < n) std::this_thread::yield();   // (a member still copying its row)
      b->rc = fn_(b->queries, n, key, &b->keys, &b->scores, &b->counts);
      {
        std::lock_guard<std::mutex> g(mu_);
        returning_ += n;                    // these callers are on their way back (an estimate: callers may also leave for good)
        if (--inflight_ == 0) idle_since_ = std::chrono::steady_clock::now();
        cv_.notify_all();                   // a leader that was collecting behind this batch may go now (after its linger)
      }
      b->gate.unlock();                     // every member at once
    } else {
      std::shared_lock<std::shared_mutex> wait(b->gate);
    }
    const int rc = b->rc;
    if (rc == 0) {
      out->clear();
      const uint32_t cnt = b->counts[slot];
      out->reserve(cnt);
      for (uint32_t j = 0; j < cnt; ++j)
        out->push_back(DT::make(b->keys[(size_t)slot * key.topk + j], b->scores[(size_t)slot * key.topk + j]));
    }
    if (b->left.fetch_add(1, std::memory_order_acq_rel) + 1 == b->n) recycle(b);      // (n is final: the batch closed before its results came)
    return rc;
  }
  Stats stats() const { std::lock_guard<std::mutex> g(mu_); return stats_; }

 private:
  struct Batch {
    BatchKey key;
    uint32_t n = 0;
    std::atomic<uint32_t> copied{0}, left{0};
    int rc = 0;
    std::shared_mutex gate;
    std::chrono::steady_clock::time_point deadline, linger;
    char *queries = nullptr;               // max_batch rows, page-locked (zvec_hip_host_alloc): the upload is one DMA, not a staged copy
    bool queries_pinned = false;
    ~Batch() {
      if (queries_pinned) zvec_hip_host_free(queries);
      else delete[] queries;
    }
    std::vector<uint64_t> keys;
    std::vector<float> scores;
    std::vector<uint32_t> counts;
  };
  // batches are recycled: a fresh 3 MB query block per batch would be page-faulted in by every leader
  std::shared_ptr<Batch> take_batch() {
    std::shared_ptr<Batch> b;
    if (!free_.empty()) { b = std::move(free_.back()); free_.pop_back(); }
    if (!b) {
      b = std::make_shared<Batch>();
      void *p = nullptr;
      if (zvec_hip_host_alloc((size_t)max_batch_ * row_bytes_, &p) == 0 && p) {
        b->queries = static_cast<char *>(p);
        b->queries_pinned = true;
      } else {
        b->queries = new char[(size_t)max_batch_ * row_bytes_];
      }
    }
    b->left.store(0, std::memory_order_relaxed);
    return b;
  }
  void recycle(std::shared_ptr<Batch> &b) {
    std::lock_guard<std::mutex> g(mu_);
    if (free_.size() < 16) free_.push_back(b);      // (never dropped in a steady state: a batch owns a page-locked block that is dear to allocate)
  }
  size_t row_bytes_;
  uint32_t max_batch_, window_us_, linger_us_;
  RunFn fn_;
  mutable std::mutex mu_;
  std::condition_variable cv_;
  std::shared_ptr<Batch> open_;
  std::vector<std::shared_ptr<Batch>> free_;
  uint32_t inflight_ = 0;      // batches currently searching
  uint32_t last_n_ = 0;        // size of the batch closed last
  bool early_go_ = true;
  uint64_t returning_ = 0;     // callers released by finished batches that have not come back yet (bounded by the linger)
  std::chrono::steady_clock::time_point idle_since_{};      // when the last searching batch ended
  Stats stats_;
};

// ---- sweeps ------------------------------------------------------------------------------------------------
//! The exclude set of a search, 1 bit per storage position: the caller's materialised set when it handed one over, else the
//! IndexFilter callback (an opaque std::function<bool(uint64_t)>, index_filter.h:48-50, true = exclude) swept once over the
//! keys in storage order — what the scan kernels gate on.  nullptr = nothing excluded.  Holes carry no key and are skipped
//! (the store excludes them itself).
template <class Ctx>
const uint64_t *exclude_bits(Ctx *ctx, const KeyDirectory &dir) {
  if (const uint64_t *preset = ctx->op_preset_bits()) return preset;
  if (!ctx->op_has_filter()) return nullptr;
  std::vector<uint64_t> &bits = ctx->op_scratch().bits;
  const size_t n = dir.size();
  bits.assign((n + 63) / 64, 0);
  const uint64_t *keys = dir.data();
  for (size_t i = 0; i < n; ++i)
    if (keys[i] != kInvalidKey && ctx->op_filtered(keys[i])) bits[i >> 6] |= 1ull << (i & 63);
  return bits.data();
}

//! IndexGroupBy is an opaque std::function<std::string(uint64_t)> too: swept once over the keys into dense group numbers
template <class Ctx>
void sweep_groups(Ctx *ctx, const KeyDirectory &dir) {
  Scratch &s = ctx->op_scratch();
  std::unordered_map<std::string, uint32_t> number_of;
  s.group_ids.clear();
  s.group_of.resize(dir.size());
  for (size_t i = 0; i < dir.size(); ++i) {
    std::string id = ctx->op_group_of(dir.at(i));
    auto it = number_of.find(id);
    if (it == number_of.end()) {
      it = number_of.emplace(id, (uint32_t)s.group_ids.size()).first;
      s.group_ids.push_back(std::move(id));
    }
    s.group_of[i] = it->second;
  }
}

//! key / score arrays of one batched call -> per-query document lists (topk_to_result: the lists end at the RNN threshold,
//! which the device gate already applied)
template <class Ctx, class DT>
void fill_results(Ctx *ctx, uint32_t count, uint32_t k) {
  const Scratch &s = ctx->op_scratch();
  auto &res = ctx->op_results();
  res.assign(count, typename DT::DocumentList());
  for (uint32_t q = 0; q < count; ++q) {
    res[q].reserve(s.counts[q]);
    for (uint32_t j = 0; j < s.counts[q]; ++j) res[q].push_back(DT::make(s.keys[(size_t)q * k + j], s.scores[(size_t)q * k + j]));
  }
}

inline void size_outputs(Scratch &s, size_t rows, size_t k) {
  s.keys.resize(rows * k);
  s.scores.resize(rows * k);
  s.counts.resize(rows);
}

//! IndexContext::set_fetch_vector (index.cc:635-647; ivf_searcher_context.h:186-197: get_vector_by_key per result): the stored
//! rows of every result document in ONE gather; `get_rows(positions, n, out)` is the index's row fetch
template <class Ctx, class DT, class GetRows>
int attach_vectors(Ctx *ctx, uint32_t count, size_t row_bytes, const KeyDirectory &dir, GetRows &&get_rows) {
  Scratch &s = ctx->op_scratch();
  auto &res = ctx->op_results();
  s.pos.clear();
  for (uint32_t q = 0; q < count; ++q)
    for (const auto &d : res[q]) {
      uint64_t pos;
      if (!dir.find(d.key(), &pos)) return ZVEC_HIP_ERR_NO_EXIST;
      s.pos.push_back(pos);
    }
  s.vectors.resize(s.pos.size() * row_bytes);
  if (s.pos.empty()) return 0;
  int rc = get_rows(s.pos.data(), s.pos.size(), &s.vectors[0]);
  if (rc != 0) return rc;
  size_t j = 0;
  for (uint32_t q = 0; q < count; ++q)
    for (auto &d : res[q]) {
      DT::attach(d, (uint32_t)s.pos[j], s.vectors.data() + j * row_bytes, row_bytes);
      ++j;
    }
  return 0;
}

// =====================================================================================================================
// flat: one body behind "FlatStreamer" and "FlatSearcher"
// =====================================================================================================================
struct BatcherOptions {
  uint32_t window_us = 0;      // proxima.hip.searcher.batch_window_us: > 0 turns the micro-batcher on
  uint32_t max_batch = 1024;   // proxima.hip.searcher.max_batch
  uint32_t linger_us = 0;      // proxima.hip.searcher.batch_linger_us
  // IVF, fp32 L2 / inner-product indexes on one device: searches pre-select on an fp16 twin of the lists, re-score in fp32 and
  // certify the result (zvec_hip_ivf_set_shadow: same results, about half the bytes per search, + half the index size in HBM)
  uint32_t shadow = 0;             // proxima.hip.searcher.half_width_preselect: != 0 turns it on (off by default)
  uint32_t shadow_preselect = 0;   // proxima.hip.searcher.preselect_rows: rows pre-selected per query (0: max(32, 3k); <= 64)
};

//! a few workspaces (stream + buffers) shared by the batch leaders: any caller thread may lead a batch, and a context per
//! thread sized for 1024-query batches would cost more than the searches
class CtxPool {
 public:
  ~CtxPool() { clear(); }
  void clear() {
    std::lock_guard<std::mutex> g(mu_);
    for (auto c : free_) zvec_hip_ctx_destroy(c);
    free_.clear();
  }
  int take(int device, zvec_hip_ctx_t *out) {
    {
      std::lock_guard<std::mutex> g(mu_);
      if (!free_.empty()) { *out = free_.back(); free_.pop_back(); return 0; }
    }
    return zvec_hip_ctx_create(device, out);
  }
  void give(zvec_hip_ctx_t c) {
    std::lock_guard<std::mutex> g(mu_);
    free_.push_back(c);
  }
 private:
  std::mutex mu_;
  std::vector<zvec_hip_ctx_t> free_;
};

template <class Ctx, class DT>
class FlatOperator {
 public:
  ~FlatOperator() { destroy(); }
  //! ndev > 1: one zvec_hip_shards_t (a row-range shard, a worker thread and a stream per device) instead of one handle.
  //! position_is_id: add_with_id keeps the reference's "storage position == id" rule (holes, in-place overwrites:
  //! FlatStreamerEntity::add_vector_with_id, flat_streamer_entity.cc:900-990); otherwise the operator keeps its own
  //! id -> position map (results carry keys, never positions: no holes to pad, and an index re-opened from rows a provider
  //! walk compacted stays addressable)
  int create(uint32_t dim, int dtype, int metric, uint32_t elem_size, int device, uint32_t ndev, bool position_is_id,
             const BatcherOptions &bo = BatcherOptions()) {
    destroy();
    device_ = device;
    elem_size_ = elem_size;
    position_is_id_ = position_is_id;
    shadow_ = bo.shadow;
    shadow_preselect_ = std::min<uint32_t>(bo.shadow_preselect, 64);
    int rc;
    if (ndev > 1) {
      std::vector<int> devs(ndev);
      for (uint32_t g = 0; g < ndev; ++g) devs[g] = device + (int)g;
      rc = zvec_hip_shards_create(dim, dtype, metric, ZVEC_HIP_SHARDS_FLAT, devs.data(), ndev, &sh_);
    } else {
      rc = zvec_hip_flat_create(dim, dtype, metric, device, &h_);
    }
    if (rc == 0 && bo.window_us > 0)
      batcher_.reset(new MicroBatcher<DT>(elem_size_, bo.max_batch, bo.window_us, bo.linger_us,
          [this](const void *q, uint32_t n, const BatchKey &key, std::vector<uint64_t> *ks, std::vector<float> *sc, std::vector<uint32_t> *cn) {
            ks->resize((size_t)n * key.topk);
            sc->resize((size_t)n * key.topk);
            cn->resize(n);
            std::shared_lock<FairSharedMutex> r(mu_);
            return raw_search(q, n, key.topk, FLT_MAX, nullptr, nullptr, ks->data(), sc->data(), cn->data());
          }));
    return rc;
  }
  void destroy() {
    batcher_.reset();
    pool_.clear();
    if (h_) zvec_hip_flat_destroy(h_);
    if (sh_) zvec_hip_shards_destroy(sh_);
    h_ = nullptr;
    sh_ = nullptr;
    dir_.clear();
  }
  bool ready() const { return h_ || sh_; }
  zvec_hip_flat_t handle() const { return h_; }
  size_t count() const { return dir_.size(); }
  uint64_t key_at(size_t pos) const { return dir_.at(pos); }
  uint32_t elem_size() const { return elem_size_; }

  //! IndexStreamer::add_impl (index_runner.h:476-480) / FlatBuilder::build in bulk; keys == nullptr: key = storage position
  int append(const void *rows, size_t n, const uint64_t *keys) {
    std::unique_lock<FairSharedMutex> w(mu_);
    int rc = sh_ ? zvec_hip_shards_flat_append(sh_, rows, n, keys) : zvec_hip_flat_append(h_, rows, n, keys);
    if (rc != 0) return rc;
    for (size_t i = 0; i < n; ++i) dir_.append(keys ? keys[i] : dir_.size());
    return 0;
  }
  //! add_with_id_impl (index_runner.h:483-487) — the call core_interface::Index::_dense_add makes for every document
  int put(uint32_t id, const void *row) {
    std::unique_lock<FairSharedMutex> w(mu_);
    const uint64_t key = id;
    if (position_is_id_) {
      if (sh_) return ZVEC_HIP_ERR_UNSUPPORTED;
      int rc = zvec_hip_flat_put(h_, &id, 1, row, nullptr);
      if (rc == 0) dir_.set(id, key);
      return rc;
    }
    uint64_t pos;
    if (!dir_.find(key, &pos)) {
      int rc = sh_ ? zvec_hip_shards_flat_append(sh_, row, 1, &key) : zvec_hip_flat_append(h_, row, 1, &key);
      if (rc == 0) dir_.append(key);
      return rc;
    }
    if (sh_) return ZVEC_HIP_ERR_UNSUPPORTED;                  // (in-place replacement runs on one device)
    const uint32_t p32 = (uint32_t)pos;
    return zvec_hip_flat_put(h_, &p32, 1, row, &key);
  }
  //! search_impl / search_bf_impl (flat_streamer.cc:304-344, flat_searcher.cc:162-211): the flat scan IS the brute force
  int search(const void *q, uint32_t count, Ctx *ctx) const {
    if (ctx->op_group_num() > 0) {                             // flat_streamer.cc:323-324
      std::shared_lock<FairSharedMutex> r(mu_);
      return group_search(q, count, ctx, nullptr, nullptr);
    }
    const uint32_t k = ctx->topk();
    if (batcher_ && count == 1 && plain(ctx)) {                // a single plain query rides a shared batch
      BatchKey key;
      key.topk = k;
      auto &res = ctx->op_results();
      res.assign(1, typename DT::DocumentList());
      return batcher_->search(q, key, &res[0]);
    }
    std::shared_lock<FairSharedMutex> r(mu_);
    Scratch &s = ctx->op_scratch();
    size_outputs(s, count, k);
    const uint64_t *bits = nullptr;
    int rc = filter_bits(ctx, &bits);
    if (rc != 0) return rc;
    rc = raw_search(q, count, k, ctx->threshold(), bits, ctx->hip(), s.keys.data(), s.scores.data(), s.counts.data());
    if (rc != 0) return rc;
    fill_results<Ctx, DT>(ctx, count, k);
    return ctx->fetch_vector() ? attach(ctx, count) : 0;
  }
  //! search_bf_by_p_keys_impl (flat_streamer.cc:346-389): unknown keys are skipped, as get_vector_by_key != 0 -> continue
  int search_by_keys(const void *q, const std::vector<std::vector<uint64_t>> &p_keys, uint32_t count, Ctx *ctx) const {
    if (p_keys.size() != count) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
    std::shared_lock<FairSharedMutex> r(mu_);
    Scratch &s = ctx->op_scratch();
    s.ids.clear();
    s.offs.assign(count + 1, 0);
    for (uint32_t i = 0; i < count; ++i) {
      for (uint64_t key : p_keys[i]) {
        uint64_t pos;
        if (dir_.find(key, &pos)) s.ids.push_back((uint32_t)pos);
      }
      s.offs[i + 1] = (uint32_t)s.ids.size();
    }
    if (s.ids.empty()) s.ids.push_back(0);
    if (ctx->op_group_num() > 0) return group_search(q, count, ctx, s.ids.data(), s.offs.data());   // flat_streamer.cc:365-366
    const uint32_t k = ctx->topk();
    size_outputs(s, count, k);
    const uint64_t *bits = nullptr;
    int rc = filter_bits(ctx, &bits);
    if (rc != 0) return rc;
    if (sh_) {
      std::vector<uint64_t> wide(s.ids.begin(), s.ids.end());
      rc = zvec_hip_shards_flat_search_by_ids(sh_, q, count, wide.data(), s.offs.data(), k, ctx->threshold(), bits, s.keys.data(),
                                              s.scores.data(), s.counts.data());
    } else {
      rc = zvec_hip_flat_search_by_ids(h_, ctx->hip(), q, count, s.ids.data(), s.offs.data(), k, ctx->threshold(), bits,
                                       s.keys.data(), s.scores.data(), s.counts.data());
    }
    if (rc != 0) return rc;
    fill_results<Ctx, DT>(ctx, count, k);
    return ctx->fetch_vector() ? attach(ctx, count) : 0;
  }
  int vector_of_key(uint64_t key, void *out) const {
    std::shared_lock<FairSharedMutex> r(mu_);
    uint64_t pos;
    return dir_.find(key, &pos) ? vector_of_pos(pos, out) : (int)ZVEC_HIP_ERR_NO_EXIST;
  }
  int vector_of_pos(uint64_t pos, void *out) const {
    return sh_ ? zvec_hip_shards_flat_get_vectors(sh_, &pos, 1, out) : zvec_hip_flat_get_vector(h_, pos, out);
  }
  //! rows of storage positions [pos0, pos0 + n) -> host (a provider's iterator)
  int rows_at(uint64_t pos0, size_t n, void *out) const {
    std::vector<uint64_t> pos(n);
    for (size_t i = 0; i < n; ++i) pos[i] = pos0 + i;
    return get_rows(pos.data(), n, out);
  }
  //! FlatSearcher::load's "flat.features" payload -> HBM (one device, or dealt over the shards)
  int load_features(const void *features, size_t bytes, size_t n, bool column_major, const uint64_t *keys) {
    std::unique_lock<FairSharedMutex> w(mu_);
    int rc = sh_ ? zvec_hip_shards_flat_load_features(sh_, features, bytes, n, column_major, 32, keys)
                 : zvec_hip_flat_load_features(h_, features, bytes, n, column_major, 32, keys);
    if (rc != 0) return rc;
    if (keys) dir_.adopt(keys, n); else dir_.adopt_identity(n);
    // FlatSearcher's one-shot load: the rows have settled — proxima.hip.searcher.half_width_preselect puts an fp16 twin beside them
    // (a store it cannot serve — fp16 rows, cosine, elements beyond the half range — keeps searching its own rows; a streamer's
    // later appends drop the twin: zvec_hip_flat_set_shadow)
    if (shadow_ && h_ && n > 0) {
      rc = zvec_hip_flat_set_shadow(h_, 1, shadow_preselect_);
      if (rc != 0 && rc != ZVEC_HIP_ERR_UNSUPPORTED) return rc;
    }
    return 0;
  }
  //! a run of FlatStreamerEntity blocks (one "flat.features<i>" segment: [bvc x element][bvc x key] ... [DeletionMap]
  //! [BlockHeader], flat_streamer_entity.cc:43-47) -> HBM: one strided copy + one pack launch; keep[b] = live rows of block b
  int load_blocks(const void *blocks, size_t bytes, size_t nblocks, uint32_t block_size, uint32_t bvc, const std::vector<uint32_t> &keep) {
    if (keep.size() < nblocks || (block_size && nblocks > bytes / block_size)) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
    std::unique_lock<FairSharedMutex> w(mu_);
    const char *p = static_cast<const char *>(blocks);
    if (!sh_) {
      int rc = zvec_hip_flat_load_blocks(h_, blocks, bytes, nblocks, block_size, bvc, keep.data());
      if (rc != 0) return rc;
    }
    for (size_t b = 0; b < nblocks; ++b)
      for (uint32_t m = keep[b]; m; m &= m - 1) {
        const uint32_t r = (uint32_t)__builtin_ctz(m);
        uint64_t key;
        memcpy(&key, p + b * block_size + (size_t)bvc * elem_size_ + (size_t)r * 8, 8);
        if (sh_) {            // sharded mirror: rows dealt by the shards' own append (no strided path there)
          int rc = zvec_hip_shards_flat_append(sh_, p + b * block_size + (size_t)r * elem_size_, 1, &key);
          if (rc != 0) return rc;
        }
        dir_.append(key);
      }
    return 0;
  }
  typename MicroBatcher<DT>::Stats batcher_stats() const { return batcher_ ? batcher_->stats() : typename MicroBatcher<DT>::Stats(); }

 private:
  static bool plain(const Ctx *ctx) {
    return !ctx->op_has_filter() && !ctx->op_preset_bits() && !ctx->op_doc_filter() && !ctx->fetch_vector() &&
           ctx->threshold() == FLT_MAX && ctx->topk() != 0;
  }
  int filter_bits(Ctx *ctx, const uint64_t **bits) const {
    if (const zvec_hip_doc_filter_t *df = ctx->op_doc_filter()) {       // composite filter as data: materialised on the GPU
      if (sh_) return ZVEC_HIP_ERR_UNSUPPORTED;
      std::vector<uint64_t> &b = ctx->op_scratch().bits;
      b.assign((dir_.size() + 63) / 64, 0);
      int rc = zvec_hip_flat_build_filter(h_, ctx->hip(), df, b.data(), 0, nullptr);
      *bits = b.data();
      return rc;
    }
    *bits = exclude_bits(ctx, dir_);
    return 0;
  }
  //! hctx == nullptr: a pooled workspace (the batcher's leaders)
  int raw_search(const void *q, uint32_t count, uint32_t k, float threshold, const uint64_t *bits, zvec_hip_ctx_t hctx,
                 uint64_t *keys, float *scores, uint32_t *counts) const {
    if (sh_) return zvec_hip_shards_search(sh_, q, count, k, threshold, 0, 0, bits, keys, scores, counts);
    zvec_hip_ctx_t pooled = nullptr;
    if (!hctx) {
      int rc = pool_.take(device_, &pooled);
      if (rc != 0) return rc;
      hctx = pooled;
    }
    int rc = zvec_hip_flat_search(h_, hctx, q, count, k, threshold, bits, keys, scores, counts);
    if (pooled) pool_.give(pooled);
    return rc;
  }
  int get_rows(const uint64_t *pos, size_t n, void *out) const {
    return sh_ ? zvec_hip_shards_flat_get_vectors(sh_, pos, n, out) : zvec_hip_flat_get_vectors(h_, pos, n, out);
  }
  int attach(Ctx *ctx, uint32_t count) const {
    return attach_vectors<Ctx, DT>(ctx, count, elem_size_, dir_, [this](const uint64_t *p, size_t n, void *out) { return get_rows(p, n, out); });
  }
  //! group_by_search_impl / group_by_search_p_keys_impl (flat_streamer.cc:391-483) + topk_to_group_result
  //! (flat_streamer_context.h:135-180); the caller holds mu_ shared.  ids == nullptr: every row competes
  int group_search(const void *q, uint32_t count, Ctx *ctx, const uint32_t *ids, const uint32_t *offs) const {
    if (!ctx->op_has_group_by()) return ZVEC_HIP_ERR_INVALID_ARGUMENT;       // "Invalid group-by function"
    if (sh_) return ZVEC_HIP_ERR_UNSUPPORTED;                                 // (group-by runs on one device)
    const uint32_t gnum = ctx->op_group_num(), gk = ctx->op_group_topk();
    if (gk == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
    Scratch &s = ctx->op_scratch();
    sweep_groups(ctx, dir_);
    const uint64_t *bits = nullptr;
    int rc = filter_bits(ctx, &bits);
    if (rc != 0) return rc;
    const size_t rows = size_t(count) * gnum;
    size_outputs(s, rows, gk);
    s.groups.resize(rows);
    s.ngroups.resize(count);
    const uint32_t ngroups = std::max<uint32_t>(1u, (uint32_t)s.group_ids.size());
    const uint32_t none = 0;
    const uint32_t *gof = s.group_of.empty() ? &none : s.group_of.data();
    rc = ids ? zvec_hip_flat_search_grouped_by_ids(h_, ctx->hip(), q, count, ids, offs, gof, ngroups, gnum, gk, ctx->threshold(), bits,
                                                   s.groups.data(), s.ngroups.data(), s.keys.data(), s.scores.data(), s.counts.data())
             : zvec_hip_flat_search_grouped(h_, ctx->hip(), q, count, gof, ngroups, gnum, gk, ctx->threshold(), bits, s.groups.data(),
                                            s.ngroups.data(), s.keys.data(), s.scores.data(), s.counts.data());
    if (rc != 0) return rc;
    auto &res = ctx->op_group_results();
    res.assign(count, typename DT::GroupDocumentList());
    s.pos.clear();
    for (uint32_t qi = 0; qi < count; ++qi) {
      res[qi].resize(s.ngroups[qi]);
      for (uint32_t g = 0; g < s.ngroups[qi]; ++g) {
        const size_t row = size_t(qi) * gnum + g;
        res[qi][g].set_group_id(s.group_ids[s.groups[row]]);
        for (uint32_t j = 0; j < s.counts[row]; ++j) {
          res[qi][g].mutable_docs()->push_back(DT::make(s.keys[row * gk + j], s.scores[row * gk + j]));
          if (ctx->fetch_vector()) {
            uint64_t pos;
            if (!dir_.find(s.keys[row * gk + j], &pos)) return ZVEC_HIP_ERR_NO_EXIST;
            s.pos.push_back(pos);
          }
        }
      }
    }
    if (!ctx->fetch_vector() || s.pos.empty()) return 0;
    s.vectors.resize(s.pos.size() * elem_size_);
    if ((rc = get_rows(s.pos.data(), s.pos.size(), &s.vectors[0])) != 0) return rc;
    size_t j = 0;
    for (auto &lst : res)
      for (auto &g : lst)
        for (auto &d : *g.mutable_docs()) {
          DT::attach(d, (uint32_t)s.pos[j], s.vectors.data() + j * elem_size_, elem_size_);
          ++j;
        }
    return 0;
  }

  zvec_hip_flat_t h_{nullptr};
  zvec_hip_shards_t sh_{nullptr};         // set instead of h_ when the index is sharded over several devices
  int device_{0};
  uint32_t elem_size_{0};
  bool position_is_id_{false};
  mutable FairSharedMutex mu_;            // add (exclusive) vs search (shared): flat_streamer.cc:236-242
  KeyDirectory dir_;                      // key of every storage position
  uint32_t shadow_{0}, shadow_preselect_{0};      // BatcherOptions::shadow / shadow_preselect
  std::unique_ptr<MicroBatcher<DT>> batcher_;
  mutable CtxPool pool_;
};

// =====================================================================================================================
// IVF-Flat: one body behind "IVFSearcher" and "IVFStreamer" (the reference's streamer only opens, searches and unloads a
// dumped index, ivf_streamer.h:28-85)
// =====================================================================================================================
template <class Ctx, class DT>
class IVFOperator {
 public:
  //! queries reformed for a centroid index that lives in a space of its own (IVFCentroidIndex::search through a
  //! MipsReformer, ivf_centroid_index.cc:273-297): count rows in, the coarse-space rows out
  using CoarseReform = std::function<int(const void *queries, uint32_t count, std::string *coarse_queries)>;

  ~IVFOperator() { destroy(); }
  void destroy() {
    batcher_.reset();
    pool_.clear();
    if (h_) zvec_hip_ivf_destroy(h_);
    if (sh_) zvec_hip_shards_destroy(sh_);
    h_ = nullptr;
    sh_ = nullptr;
    dir_.clear();
    reform_ = nullptr;
  }
  bool ready() const { return h_ || sh_; }
  size_t count() const { return dir_.size(); }
  uint64_t key_at(size_t pos) const { return dir_.at(pos); }
  uint32_t elem_size() const { return elem_size_; }
  uint32_t nlist() const { return nlist_; }

  //! create the device index (ndev > 1: whole inverted lists dealt over the devices, centroids replicated); then one of the loaders
  int create(uint32_t dim, int dtype, int metric, uint32_t elem_size, int device, uint32_t ndev, const BatcherOptions &bo = BatcherOptions()) {
    destroy();
    device_ = device;
    elem_size_ = elem_size;
    bo_ = bo;
    if (ndev > 1) {
      std::vector<int> devs(ndev);
      for (uint32_t g = 0; g < ndev; ++g) devs[g] = device + (int)g;
      return zvec_hip_shards_create(dim, dtype, metric, ZVEC_HIP_SHARDS_IVF, devs.data(), ndev, &sh_);
    }
    return zvec_hip_ivf_create(dim, dtype, metric, device, &h_);
  }
  //! what IVFSearcher::load reads from the ivf.* segments, as arrays (ivf_index_format.h:26-60,152-164)
  int load_arrays(const void *centroids, uint32_t nlist, const uint64_t *list_offsets, const void *vecs, const uint64_t *keys) {
    int rc = sh_ ? zvec_hip_shards_ivf_load(sh_, centroids, nlist, list_offsets, vecs, keys)
                 : zvec_hip_ivf_load(h_, centroids, nlist, list_offsets, vecs, keys);
    if (rc != 0) return rc;
    if (keys) dir_.adopt(keys, list_offsets[nlist]); else dir_.adopt_identity(list_offsets[nlist]);
    return loaded(nlist);
  }
  //! ... as the raw segment payloads of a dumped index (IVFSearcher::load, ivf_searcher.cc:43-103 + IVFEntity::load,
  //! ivf_entity.cc:443-570).  coarse_dim != 0: the centroid rows live in a space of their own (`reform` maps the queries there)
  int load_segments(const std::string &header, const std::string &lmeta, const std::string &body, const std::string &keys,
                    const std::string &centroids, uint32_t nlist, uint32_t coarse_dim, int coarse_metric, CoarseReform reform) {
    int rc;
    if (sh_) {
      if (coarse_dim) return ZVEC_HIP_ERR_UNSUPPORTED;                       // (the coarse space runs on one device)
      rc = zvec_hip_shards_ivf_load_segments(sh_, header.data(), header.size(), lmeta.data(), lmeta.size(), body.data(), body.size(),
                                             keys.data(), keys.size(), centroids.data());
    } else {
      rc = zvec_hip_ivf_load_segments(h_, header.data(), header.size(), lmeta.data(), lmeta.size(), body.data(), body.size(),
                                      keys.data(), keys.size(), coarse_dim ? nullptr : centroids.data());
      if (rc == 0 && coarse_dim) rc = zvec_hip_ivf_set_coarse_space(h_, coarse_dim, coarse_metric, centroids.data(), nlist);
    }
    if (rc != 0) return rc;
    dir_.adopt(reinterpret_cast<const uint64_t *>(keys.data()), keys.size() / sizeof(uint64_t));
    reform_ = coarse_dim ? std::move(reform) : nullptr;
    coarse_row_bytes_ = coarse_dim ? centroids.size() / std::max<uint32_t>(nlist, 1) : 0;
    return loaded(nlist);
  }
  //! IVFEntity::get_vector_by_key: the stored row of a document (list-order position through the key map)
  int vector_of_key(uint64_t key, void *out) const {
    uint64_t pos;
    if (!dir_.find(key, &pos)) return ZVEC_HIP_ERR_NO_EXIST;
    if (sh_) return ZVEC_HIP_ERR_UNSUPPORTED;                  // (row fetches run on one device)
    return zvec_hip_ivf_get_vector(h_, pos, out);
  }
  //! rows of list-order positions [pos0, pos0 + n) -> host (the provider's iterator walks the index in chunks)
  int rows_at(uint64_t pos0, size_t n, void *out) const {
    if (sh_) return ZVEC_HIP_ERR_UNSUPPORTED;
    std::vector<uint64_t> pos(n);
    for (size_t i = 0; i < n; ++i) pos[i] = pos0 + i;
    return zvec_hip_ivf_get_vectors(h_, pos.data(), n, out);
  }
  //! IVFSearcher::search_impl / search_bf_impl (ivf_searcher.cc:106-250); scan_ratio / bruteforce_threshold are the
  //! context's (IVFSearcherContext::update, ivf_searcher_context.h:61-79)
  int search(const void *q, uint32_t count, Ctx *ctx, bool brute_force, float scan_ratio, uint32_t bruteforce_threshold) const {
    const uint32_t k = ctx->topk();
    BatchKey key;
    key.topk = k;
    if (brute_force || dir_.size() <= bruteforce_threshold) {               // ivf_searcher.cc:188-190
      key.mode = 1;
    } else {
      const ProbeParams pp = probe_params(nlist_, dir_.size(), scan_ratio, bruteforce_threshold);
      key.a = pp.nprobe;
      key.b = pp.max_scan;
    }
    if (batcher_ && count == 1 && plain(ctx)) {                              // a single plain query rides a shared batch
      auto &res = ctx->op_results();
      res.assign(1, typename DT::DocumentList());
      return batcher_->search(q, key, &res[0]);
    }
    Scratch &s = ctx->op_scratch();
    size_outputs(s, count, k);
    const uint64_t *bits = nullptr;
    if (const zvec_hip_doc_filter_t *df = ctx->op_doc_filter()) {          // composite filter as data: materialised on the GPU
      if (sh_) return ZVEC_HIP_ERR_UNSUPPORTED;
      s.bits.assign((dir_.size() + 63) / 64, 0);
      int frc = zvec_hip_ivf_build_filter(h_, ctx->hip(), df, s.bits.data(), 0, nullptr);
      if (frc != 0) return frc;
      bits = s.bits.data();
    } else {
      bits = exclude_bits(ctx, dir_);                                       // keys in list order (ivf_entity.cc:612)
    }
    int rc = raw_search(q, count, key, ctx->threshold(), bits, ctx->hip(), s.keys.data(), s.scores.data(), s.counts.data());
    if (rc != 0) return rc;
    fill_results<Ctx, DT>(ctx, count, k);
    if (!ctx->fetch_vector()) return 0;
    if (sh_) return ZVEC_HIP_ERR_UNSUPPORTED;
    zvec_hip_ivf_t h = h_;
    return attach_vectors<Ctx, DT>(ctx, count, elem_size_, dir_,
                                   [h](const uint64_t *p, size_t n, void *out) { return zvec_hip_ivf_get_vectors(h, p, n, out); });
  }
  typename MicroBatcher<DT>::Stats batcher_stats() const { return batcher_ ? batcher_->stats() : typename MicroBatcher<DT>::Stats(); }

 private:
  static bool plain(const Ctx *ctx) {
    return !ctx->op_has_filter() && !ctx->op_preset_bits() && !ctx->op_doc_filter() && !ctx->fetch_vector() &&
           ctx->threshold() == FLT_MAX && ctx->topk() != 0;
  }
  int loaded(uint32_t nlist) {
    nlist_ = nlist;
    if (bo_.shadow && !sh_) {
      // (an index the shadow lists cannot serve — fp16 rows, cosine, elements beyond the half range — keeps searching its own lists)
      const int rc = zvec_hip_ivf_set_shadow(h_, 1, std::min<uint32_t>(bo_.shadow_preselect, 64));
      if (rc != 0 && rc != ZVEC_HIP_ERR_UNSUPPORTED) return rc;
    }
    if (bo_.window_us > 0)
      batcher_.reset(new MicroBatcher<DT>(elem_size_, bo_.max_batch, bo_.window_us, bo_.linger_us,
          [this](const void *q, uint32_t n, const BatchKey &key, std::vector<uint64_t> *ks, std::vector<float> *sc, std::vector<uint32_t> *cn) {
            ks->resize((size_t)n * key.topk);
            sc->resize((size_t)n * key.topk);
            cn->resize(n);
            return raw_search(q, n, key, FLT_MAX, nullptr, nullptr, ks->data(), sc->data(), cn->data());
          }));
    return 0;
  }
  //! hctx == nullptr: a pooled workspace (the batcher's leaders)
  int raw_search(const void *q, uint32_t count, const BatchKey &key, float threshold, const uint64_t *bits, zvec_hip_ctx_t hctx,
                 uint64_t *keys, float *scores, uint32_t *counts) const {
    if (sh_) {
      return key.mode == 1 ? zvec_hip_shards_search(sh_, q, count, key.topk, threshold, nlist_, 0xffffffffu, bits, keys, scores, counts)   // every list
                           : zvec_hip_shards_search(sh_, q, count, key.topk, threshold, key.a, key.b, bits, keys, scores, counts);
    }
    zvec_hip_ctx_t pooled = nullptr;
    if (!hctx) {
      int rc = pool_.take(device_, &pooled);
      if (rc != 0) return rc;
      hctx = pooled;
    }
    int rc;
    if (key.mode == 1) {
      rc = zvec_hip_ivf_search_bf(h_, hctx, q, count, key.topk, threshold, bits, keys, scores, counts);
    } else if (reform_) {
      std::string cq;
      rc = reform_(q, count, &cq);
      // the reformer's output must be rows of the installed coarse space: a width the device store does not have would be
      // read past its end (a MipsReformer whose m_value / forced type differs from the dumped meta)
      if (rc == 0 && cq.size() != (size_t)count * coarse_row_bytes_) rc = ZVEC_HIP_ERR_MISMATCH;
      if (rc == 0)
        rc = zvec_hip_ivf_search_coarse(h_, hctx, q, cq.data(), count, key.topk, threshold, key.a, key.b, bits, keys, scores, counts);
    } else {
      rc = zvec_hip_ivf_search(h_, hctx, q, count, key.topk, threshold, key.a, key.b, bits, keys, scores, counts);
    }
    if (pooled) pool_.give(pooled);
    return rc;
  }

  zvec_hip_ivf_t h_{nullptr};
  zvec_hip_shards_t sh_{nullptr};         // set instead of h_ when the lists are dealt over several devices
  int device_{0};
  uint32_t elem_size_{0}, nlist_{0};
  size_t coarse_row_bytes_{0};
  CoarseReform reform_;
  BatcherOptions bo_;
  KeyDirectory dir_;                      // keys in list order
  std::unique_ptr<MicroBatcher<DT>> batcher_;
  mutable CtxPool pool_;
};

}  // namespace zvec_hip_op
