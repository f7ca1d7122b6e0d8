// ref_core_shim.cc — TEST INFRASTRUCTURE ONLY.  extern "C" doors onto the REFERENCE's own core library, compiled IN PLACE from
// /root/reference by oracle/Makefile (target `ref_core`, output oracle/_ref/libzvec_ref_core.so; nothing of the reference is
// copied, no stand-in header / library / symbol is written, the reference's build system is not used).  The doors drive ANY
// class registered with the reference's factories BY NAME, so one harness runs
//   * the reference's CPU operators       "FlatSearcher" "FlatStreamer" "IVFSearcher" "IVFStreamer" "FlatBuilder" "IVFBuilder"
//   * the plugin's operators (plugin/*.cc) "HipFlatSearcher" "HipFlatStreamer" "HipIVFSearcher" "HipIVFStreamer" "HipIVFBuilder",
//     brought in the way the product would: IndexPluginBroker::emplace(path) -> dlopen (index_plugin.h:73-101)
// through the same IndexRunner virtuals (index_runner.h:476-585) on the same index files and compares what comes back.
// Index files live in the reference's IndexMemory ("MemoryDumper" writes, "MemoryReadStorage" reads) or on disk
// ("FileDumper", "MMapFileReadStorage", "MMapFileStorage").
#include <zvec/core/framework/index_builder.h>
#include <zvec/core/framework/index_dumper.h>
#include <zvec/core/framework/index_helper.h>
#include <zvec/core/framework/index_cluster.h>
#include <zvec/core/framework/index_converter.h>
#include <zvec/core/framework/index_reformer.h>
#include <zvec/core/framework/index_factory.h>
#include <zvec/core/framework/index_holder.h>
#include <zvec/core/framework/index_memory.h>
#include <zvec/core/framework/index_plugin.h>
#include <zvec/core/framework/index_searcher.h>
#include <zvec/core/framework/index_streamer.h>

#include "core/algorithm/ivf/ivf_dumper.h"

#include <atomic>
#include <map>
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

using namespace zvec;
using namespace zvec::core;

namespace {

// rows borrowed from the caller (numpy): no copy, multi-pass
class BorrowedHolder : public IndexHolder {
 public:
  BorrowedHolder(IndexMeta::DataType dt, uint32_t dim, size_t es, const void *rows, const uint64_t *keys, uint64_t n)
      : dt_(dt), dim_(dim), es_(es), rows_(static_cast<const char *>(rows)), keys_(keys), n_(n) {}
  size_t count() const override { return n_; }
  size_t dimension() const override { return dim_; }
  IndexMeta::DataType data_type() const override { return dt_; }
  size_t element_size() const override { return es_; }
  bool multipass() const override { return true; }
  Iterator::Pointer create_iterator() override { return Iterator::Pointer(new It(this)); }

 private:
  struct It : public Iterator {
    explicit It(const BorrowedHolder *h) : h_(h) {}
    const void *data() const override { return h_->rows_ + i_ * h_->es_; }
    bool is_valid() const override { return i_ < h_->n_; }
    uint64_t key() const override { return h_->keys_ ? h_->keys_[i_] : i_; }
    void next() override { ++i_; }
    const BorrowedHolder *h_;
    size_t i_{0};
  };
  IndexMeta::DataType dt_;
  uint32_t dim_;
  size_t es_;
  const char *rows_;
  const uint64_t *keys_;
  uint64_t n_;
};

IndexMeta make_meta(int dtype, uint32_t dim, const char *metric) {
  IndexMeta meta(dtype ? IndexMeta::DT_FP16 : IndexMeta::DT_FP32, dim);
  meta.set_metric(std::string(metric), 0, ailego::Params());
  return meta;
}

bool parse_params(const char *json, ailego::Params *p) {
  if (!json || !*json) return true;
  return ailego::Params::ParseFromBuffer(std::string(json), p);
}

struct Runner {
  IndexSearcher::Pointer searcher;
  IndexStreamer::Pointer streamer;
  IndexStorage::Pointer storage;
  IndexRunner *get() const { return searcher ? static_cast<IndexRunner *>(searcher.get()) : static_cast<IndexRunner *>(streamer.get()); }
};

struct Ctx {
  IndexContext::Pointer c;
  std::vector<uint8_t> exclude;        // by key: 1 = excluded (IndexFilter: true = exclude, index_filter.h:48-50)
  std::vector<uint32_t> group_of_key;  // by key
};


// ---- an index whose big segments stay where the caller has them -------------------------------------------------------------
// SkeletonDumper: an IndexDumper (index_dumper.h:25-53) that keeps every segment in memory EXCEPT the first `skip` bytes
// written, which it only counts: the reference's own IVFDumper then produces header / list meta / keys / offsets / mapping /
// centroid index / IndexMeta byte for byte while the 30 GB body it writes first is dropped.
class SkeletonDumper : public IndexDumper {
 public:
  explicit SkeletonDumper(size_t skip) : skip_(skip) {}
  int init(const ailego::Params &) override { return 0; }
  int cleanup() override { return 0; }
  int create(const std::string &) override { return 0; }
  int close() override { return 0; }
  uint32_t magic() const override { return 0; }
  size_t write(const void *data, size_t len) override {
    size_t drop = std::min(len, skip_ - skipped_);
    skipped_ += drop;
    if (len > drop) pending_.append(static_cast<const char *>(data) + drop, len - drop);
    written_ += len;
    return len;
  }
  int append(const std::string &id, size_t data_size, size_t padding_size, uint32_t crc) override {
    const size_t region = data_size + padding_size;
    if (first_) {                       // the skipped bytes are exactly the first segment
      first_ = false;
      if (skip_ != 0) {
        if (skipped_ != skip_ || region != skip_ || !pending_.empty()) return IndexError_Logic;
        skipped_id_ = id;
        return 0;
      }
    }
    if (pending_.size() != region) return IndexError_Logic;
    segments_[id] = std::make_pair(pending_.substr(0, data_size), crc);
    pending_.clear();
    return 0;
  }
  std::map<std::string, std::pair<std::string, uint32_t>> segments_;
  std::string skipped_id_;

 private:
  size_t skip_, skipped_{0}, written_{0};
  bool first_{true};
  std::string pending_;
};

// BorrowedStorage: an IndexStorage (index_storage.h:228-270) over segments that are either owned strings or memory the caller
// keeps alive (numpy arrays): read() lends the pointer, as the mmap storages do.
class BorrowedStorage : public IndexStorage {
 public:
  struct Seg : public IndexStorage::Segment, public std::enable_shared_from_this<Seg> {
    Seg(const void *p, size_t n, uint32_t crc) : p_(static_cast<const char *>(p)), n_(n), crc_(crc) {}
    size_t data_size() const override { return n_; }
    uint32_t data_crc() const override { return crc_; }
    size_t padding_size() const override { return 0; }
    size_t capacity() const override { return n_; }
    size_t fetch(size_t off, void *buf, size_t len) const override {
      if (off > n_) off = n_;
      len = std::min(len, n_ - off);
      memcpy(buf, p_ + off, len);
      return len;
    }
    size_t read(size_t off, const void **data, size_t len) override {
      if (off > n_) off = n_;
      len = std::min(len, n_ - off);
      *data = p_ + off;
      return len;
    }
    size_t read(size_t off, MemoryBlock &data, size_t len) override {
      const void *p = nullptr;
      size_t r = this->read(off, &p, len);
      data.reset(const_cast<void *>(p));
      return r;
    }
    bool read(SegmentData *iov, size_t count) override {
      for (auto *e = iov + count; iov != e; ++iov) {
        if (iov->offset + iov->length > n_) return false;
        iov->data = p_ + iov->offset;
      }
      return true;
    }
    size_t write(size_t, const void *, size_t) override { return 0; }
    size_t resize(size_t) override { return 0; }
    void update_data_crc(uint32_t) override {}
    IndexStorage::Segment::Pointer clone() override { return shared_from_this(); }
    const char *p_;
    size_t n_;
    uint32_t crc_;
  };
  void own(const std::string &id, std::string bytes, uint32_t crc) {
    owned_.push_back(std::make_unique<std::string>(std::move(bytes)));
    segs_[id] = std::make_shared<Seg>(owned_.back()->data(), owned_.back()->size(), crc);
  }
  void borrow(const std::string &id, const void *p, size_t n) { segs_[id] = std::make_shared<Seg>(p, n, 0); }
  int init(const ailego::Params &) override { return 0; }
  int cleanup() override { return 0; }
  int open(const std::string &, bool) override { return 0; }
  int flush() override { return 0; }
  int close() override { return 0; }
  int append(const std::string &, size_t) override { return IndexError_NotImplemented; }
  void refresh(uint64_t) override {}
  uint64_t check_point() const override { return 0; }
  IndexStorage::Segment::Pointer get(const std::string &id, int) override {
    auto it = segs_.find(id);
    return it == segs_.end() ? nullptr : it->second;
  }
  bool has(const std::string &id) const override { return segs_.count(id) != 0; }
  uint32_t magic() const override { return 0x5a564543u; }

 private:
  std::map<std::string, std::shared_ptr<Seg>> segs_;
  std::vector<std::unique_ptr<std::string>> owned_;
};

IndexPluginBroker &broker() {
  static IndexPluginBroker b;
  return b;
}

}  // namespace

extern "C" {

int zref_core_abi() { return 2; }

// ---- plugins / factory ----------------------------------------------------------------------------------------------
int zref_load_plugin(const char *path, char *err, uint64_t err_cap) {
  std::string e;
  bool ok = broker().emplace(std::string(path), &e);
  if (!ok && err && err_cap) snprintf(err, err_cap, "%s", e.c_str());
  return ok ? 0 : -1;
}
int zref_has(const char *kind, const char *name) {
  const std::string k(kind), n(name);
  if (k == "searcher") return IndexFactory::HasSearcher(n);
  if (k == "streamer") return IndexFactory::HasStreamer(n);
  if (k == "builder") return IndexFactory::HasBuilder(n);
  if (k == "metric") return IndexFactory::HasMetric(n);
  if (k == "storage") return IndexFactory::HasStorage(n);
  if (k == "cluster") return IndexFactory::HasCluster(n);
  return -1;
}

// ---- index file images in IndexMemory ---------------------------------------------------------------------------------
int zref_mem_put(const char *name, const void *bytes, uint64_t size) {
  IndexMemory::Instance()->remove(name);
  auto rope = IndexMemory::Instance()->create(std::string(name));
  if (!rope) return -1;
  auto &blk = rope->append(size);
  return blk.write(0, bytes, size) == size ? 0 : -2;
}
int zref_mem_size(const char *name, uint64_t *size) {
  auto rope = IndexMemory::Instance()->open(name);
  if (!rope || rope->count() != 1) return -1;
  *size = (*rope)[0].size();
  return 0;
}
int zref_mem_get(const char *name, void *out, uint64_t cap) {
  auto rope = IndexMemory::Instance()->open(name);
  if (!rope || rope->count() != 1) return -1;
  const size_t size = (*rope)[0].size();
  if (size > cap) return -2;
  return (*rope)[0].fetch(0, out, size) == size ? 0 : -3;
}
int zref_mem_remove(const char *name) {
  IndexMemory::Instance()->remove(name);
  return 0;
}

// ---- builders: IndexFactory::CreateBuilder(cls) -> init / train / build / dump (index_builder.h:37-56) -------------------
// dumper_cls "MemoryDumper" (target = IndexMemory name) or "FileDumper" (target = path)
int zref_build(const char *cls, int dtype, uint32_t dim, const char *metric, int column_major, const char *params_json,
               const void *rows, const uint64_t *keys, uint64_t n, const char *dumper_cls, const char *target, double *seconds) {
  IndexMeta meta = make_meta(dtype, dim, metric);
  meta.set_major_order(column_major ? IndexMeta::MO_COLUMN : IndexMeta::MO_ROW);
  ailego::Params params;
  if (!parse_params(params_json, &params)) return -1000;
  auto builder = IndexFactory::CreateBuilder(cls);
  if (!builder) return -1001;
  int rc = builder->init(meta, params);
  if (rc != 0) return rc;
  IndexHolder::Pointer holder = std::make_shared<BorrowedHolder>(meta.data_type(), dim, meta.element_size(), rows, keys, n);
  auto t0 = std::chrono::steady_clock::now();
  if ((rc = builder->train(holder)) != 0 && rc != IndexError_NotImplemented) return rc;
  if ((rc = builder->build(holder)) != 0) return rc;
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  auto dumper = IndexFactory::CreateDumper(dumper_cls);
  if (!dumper) return -1003;
  if ((rc = dumper->init(ailego::Params())) != 0) return rc;
  if (std::string(dumper_cls) == "MemoryDumper") IndexMemory::Instance()->remove(target);
  if ((rc = dumper->create(target)) != 0) return rc;
  if ((rc = builder->dump(dumper)) != 0) return rc;
  return dumper->close();
}

// ---- searchers: CreateSearcher(cls) -> init(params) -> load(storage, metric) (index_searcher.h:42-54) --------------------
void *zref_searcher_open(const char *cls, const char *params_json, const char *storage_cls, const char *target, int *rc_out) {
  auto r = std::make_unique<Runner>();
  ailego::Params params;
  int rc = 0;
  do {
    if (!parse_params(params_json, &params)) { rc = -1000; break; }
    r->searcher = IndexFactory::CreateSearcher(cls);
    if (!r->searcher) { rc = -1001; break; }
    if ((rc = r->searcher->init(params)) != 0) break;
    r->storage = IndexFactory::CreateStorage(storage_cls);
    if (!r->storage) { rc = -1002; break; }
    if ((rc = r->storage->init(ailego::Params())) != 0) break;
    if ((rc = r->storage->open(target, false)) != 0) break;
    rc = r->searcher->load(r->storage, IndexMetric::Pointer());
  } while (false);
  if (rc_out) *rc_out = rc;
  return rc == 0 ? r.release() : nullptr;
}

// ---- streamers: CreateStreamer(cls) -> init(meta, params) -> open(storage) (index_streamer.h:36-48) ----------------------
// storage_cls "MMapFileStorage" + a path (mutable flat streamer), or a read storage over a dumped index (IVFStreamer)
void *zref_streamer_open(const char *cls, int dtype, uint32_t dim, const char *metric, const char *params_json,
                         const char *storage_cls, const char *target, int create, int *rc_out) {
  auto r = std::make_unique<Runner>();
  ailego::Params params;
  int rc = 0;
  do {
    if (!parse_params(params_json, &params)) { rc = -1000; break; }
    r->streamer = IndexFactory::CreateStreamer(cls);
    if (!r->streamer) { rc = -1001; break; }
    IndexMeta meta = make_meta(dtype, dim, metric);
    if ((rc = r->streamer->init(meta, params)) != 0) break;
    r->storage = IndexFactory::CreateStorage(storage_cls);
    if (!r->storage) { rc = -1002; break; }
    if ((rc = r->storage->init(ailego::Params())) != 0) break;
    if ((rc = r->storage->open(target, create != 0)) != 0) break;
    rc = r->streamer->open(r->storage);
  } while (false);
  if (rc_out) *rc_out = rc;
  return rc == 0 ? r.release() : nullptr;
}
int zref_streamer_add(void *h, int dtype, uint32_t dim, const uint64_t *keys, const void *rows, uint64_t n, int with_id) {
  auto *r = static_cast<Runner *>(h);
  if (!r->streamer) return IndexError_Unsupported;
  IndexQueryMeta qm(dtype ? IndexMeta::DT_FP16 : IndexMeta::DT_FP32, dim);
  auto ctx = r->streamer->create_context();
  const char *p = static_cast<const char *>(rows);
  for (uint64_t i = 0; i < n; ++i) {
    int rc = with_id ? r->streamer->add_with_id_impl((uint32_t)keys[i], p + i * qm.element_size(), qm, ctx)
                     : r->streamer->add_impl(keys[i], p + i * qm.element_size(), qm, ctx);
    if (rc != 0) return rc;
  }
  return 0;
}
int zref_streamer_flush(void *h) {
  auto *r = static_cast<Runner *>(h);
  return r->streamer ? r->streamer->flush(0) : (int)IndexError_Unsupported;
}
int zref_streamer_dump(void *h, const char *dumper_cls, const char *target) {
  auto *r = static_cast<Runner *>(h);
  if (!r->streamer) return IndexError_Unsupported;
  auto dumper = IndexFactory::CreateDumper(dumper_cls);
  if (!dumper) return -1003;
  int rc = dumper->init(ailego::Params());
  if (rc != 0) return rc;
  if (std::string(dumper_cls) == "MemoryDumper") IndexMemory::Instance()->remove(target);
  if ((rc = dumper->create(target)) != 0) return rc;
  if ((rc = r->streamer->dump(dumper)) != 0) return rc;
  return dumper->close();
}
int zref_runner_close(void *h) {
  auto *r = static_cast<Runner *>(h);
  int rc = 0;
  if (r->searcher) rc = r->searcher->unload();
  if (r->streamer) rc = r->streamer->close();
  if (r->storage) r->storage->close();
  delete r;
  return rc;
}
uint64_t zref_runner_count(void *h) {
  auto *r = static_cast<Runner *>(h);
  auto p = r->get()->create_provider();
  return p ? p->count() : 0;
}
// IndexRunner::get_vector(key) (index_runner.h:440-453)
int zref_runner_get_vector(void *h, uint64_t key, void *out, uint32_t elem_size) {
  auto *r = static_cast<Runner *>(h);
  IndexStorage::MemoryBlock block;
  int rc = r->get()->get_vector(key, block);
  if (rc != 0) return rc;
  if (!block.data()) return IndexError_NoExist;
  memcpy(out, block.data(), elem_size);
  return 0;
}
// provider walk (index_provider.h): keys and rows in storage order
int64_t zref_runner_walk(void *h, uint64_t *keys, void *rows, uint32_t elem_size, uint64_t cap) {
  auto *r = static_cast<Runner *>(h);
  auto p = r->get()->create_provider();
  if (!p) return -1;
  uint64_t i = 0;
  for (auto it = p->create_iterator(); it && it->is_valid(); it->next(), ++i) {
    if (i >= cap) return -2;
    keys[i] = it->key();
    memcpy(static_cast<char *>(rows) + i * elem_size, it->data(), elem_size);
  }
  return (int64_t)i;
}

// ---- contexts (index_context.h:123-262) ---------------------------------------------------------------------------------
void *zref_ctx_create(void *h) {
  auto *r = static_cast<Runner *>(h);
  auto c = std::make_unique<Ctx>();
  c->c = r->get()->create_context();
  return c->c ? c.release() : nullptr;
}
void zref_ctx_destroy(void *c) { delete static_cast<Ctx *>(c); }
void zref_ctx_set_topk(void *c, uint32_t k) { static_cast<Ctx *>(c)->c->set_topk(k); }
void zref_ctx_set_threshold(void *c, int on, float v) {
  auto *x = static_cast<Ctx *>(c);
  if (on) x->c->set_threshold(v); else x->c->reset_threshold();
}
void zref_ctx_set_fetch_vector(void *c, int on) { static_cast<Ctx *>(c)->c->set_fetch_vector(on != 0); }
int zref_ctx_update(void *c, const char *params_json) {
  ailego::Params p;
  if (!parse_params(params_json, &p)) return -1000;
  return static_cast<Ctx *>(c)->c->update(p);
}
// filter: exclude_by_key[key] != 0 => excluded; keys >= n are kept.  n == 0 resets the filter
void zref_ctx_set_filter(void *c, const uint8_t *exclude_by_key, uint64_t n) {
  auto *x = static_cast<Ctx *>(c);
  if (n == 0) { x->exclude.clear(); x->c->reset_filter(); return; }
  x->exclude.assign(exclude_by_key, exclude_by_key + n);
  const std::vector<uint8_t> *ex = &x->exclude;
  x->c->set_filter([ex](uint64_t key) { return key < ex->size() && (*ex)[key] != 0; });
}
// group-by: group id of a key = decimal string of group_of_key[key] (index_groupby.h); n == 0 resets
void zref_ctx_set_group(void *c, const uint32_t *group_of_key, uint64_t n, uint32_t group_num, uint32_t group_topk) {
  auto *x = static_cast<Ctx *>(c);
  x->c->set_group_params(group_num, group_topk);
  if (n == 0) { x->group_of_key.clear(); x->c->reset_group_by(); return; }
  x->group_of_key.assign(group_of_key, group_of_key + n);
  const std::vector<uint32_t> *g = &x->group_of_key;
  x->c->set_group_by([g](uint64_t key) { return std::to_string(key < g->size() ? (*g)[key] : 0xffffffffu); });
}


// ---- a searcher over an IVF index given as ARRAYS (the structure the GPU built, exported): the reference's own IVFDumper walks
// the rows list by list exactly as IVFBuilder::dump does (ivf_builder.cc:652-729) and writes every small segment; the body — the
// rows in list order, which for a row-major index whose element size is a multiple of 32 bytes IS the dumped body byte for byte
// (blocks of 32 vectors, no padding: ivf_dumper.h:131-160) — is lent from the caller's array instead of being copied into a file.
// The caller keeps rows / keys alive while the runner lives.
void *zref_ivf_searcher_over_rows(const char *cls, const char *params_json, int dtype, uint32_t dim, const char *metric,
                                  const void *centroids, uint32_t nlist, const uint64_t *list_offsets, const void *rows,
                                  const uint64_t *keys, int *rc_out) {
  auto r = std::make_unique<Runner>();
  int rc = 0;
  do {
    ailego::Params params;
    if (!parse_params(params_json, &params)) { rc = -1000; break; }
    IndexMeta meta = make_meta(dtype, dim, metric);
    meta.set_major_order(IndexMeta::MO_ROW);
    const size_t es = meta.element_size();
    const uint64_t n = list_offsets[nlist];
    if (es % 32 != 0) { rc = IndexError_Unsupported; break; }       // (a partial block would carry padding the array lacks)
    auto sd = std::make_shared<SkeletonDumper>((size_t)n * es);
    IndexDumper::Pointer dumper = sd;
    {
      IVFDumper ivf(meta, dumper, nlist);
      const char *p = static_cast<const char *>(rows);
      for (uint32_t l = 0; l < nlist && rc == 0; ++l)
        for (uint64_t i = list_offsets[l]; i < list_offsets[l + 1]; ++i)
          if ((rc = ivf.dump_inverted_vector(l, keys ? keys[i] : i, p + (size_t)i * es)) != 0) break;
      if (rc != 0) break;
      if ((rc = ivf.dump_inverted_vector_finished()) != 0) break;
      if ((rc = ivf.dump_quantizer_params({})) != 0) break;
      // the centroid index: the file a FlatBuilder dumps (IVFCentroidIndex::build, ivf_centroid_index.cc:468-490)
      auto fb = IndexFactory::CreateBuilder("FlatBuilder");
      auto md = IndexFactory::CreateDumper("MemoryDumper");
      if (!fb || !md) { rc = -1001; break; }
      IndexMeta cmeta = make_meta(dtype, dim, metric);
      cmeta.set_major_order(IndexMeta::MO_ROW);
      if ((rc = fb->init(cmeta, ailego::Params())) != 0) break;
      IndexHolder::Pointer ch = std::make_shared<BorrowedHolder>(cmeta.data_type(), dim, es, centroids, nullptr, nlist);
      if ((rc = fb->train(ch)) != 0 && rc != IndexError_NotImplemented) break;
      if ((rc = fb->build(ch)) != 0) break;
      static std::atomic<uint32_t> serial{0};
      const std::string cpath = "zref_cent_" + std::to_string(serial.fetch_add(1));
      if ((rc = md->init(ailego::Params())) != 0 || (rc = md->create(cpath)) != 0) break;
      if ((rc = fb->dump(md)) != 0 || (rc = md->close()) != 0) break;
      auto crope = IndexMemory::Instance()->open(cpath);
      if (!crope || crope->count() != 1) { rc = -1004; break; }
      const void *cdata = nullptr;
      (*crope)[0].read(0, &cdata, 0);
      rc = ivf.dump_centroid_index(cdata, (*crope)[0].size());
      IndexMemory::Instance()->remove(cpath);
      if (rc != 0) break;
    }
    meta.set_searcher("IVFSearcher", 0, ailego::Params());
    meta.set_builder("IVFBuilder", 0, ailego::Params());
    if ((rc = IndexHelper::SerializeToDumper(meta, dumper.get())) != 0) break;
    if (sd->skipped_id_ != "ivf.inverted_body" && n != 0) { rc = IndexError_Logic; break; }
    auto st = std::make_shared<BorrowedStorage>();
    for (auto &kv : sd->segments_) st->own(kv.first, std::move(kv.second.first), kv.second.second);
    st->borrow("ivf.inverted_body", rows, (size_t)n * es);
    r->storage = st;
    r->searcher = IndexFactory::CreateSearcher(cls);
    if (!r->searcher) { rc = -1001; break; }
    if ((rc = r->searcher->init(params)) != 0) break;
    rc = r->searcher->load(r->storage, IndexMetric::Pointer());
  } while (false);
  if (rc_out) *rc_out = rc;
  return rc == 0 ? r.release() : nullptr;
}


// ---- boundary A's pre / post-processing around boundary B (src/core/interface/index.cc) ------------------------------------------
// zref_build_converted: what Index::Add + Train + Dump amount to for a converted index — IndexFactory::CreateConverter(converter),
// init(meta), TrainAndTransform over the caller's rows (index.cc:111-183: the converter's meta names the reformer), then the builder
// over the converter's result holder with the converter's meta.
int zref_build_converted(const char *builder_cls, const char *converter_cls, int dtype, uint32_t dim, const char *metric,
                         const char *builder_params_json, const void *rows, const uint64_t *keys, uint64_t n, const char *target) {
  IndexMeta meta = make_meta(dtype, dim, metric);
  ailego::Params bparams;
  if (!parse_params(builder_params_json, &bparams)) return -1000;
  auto conv = IndexFactory::CreateConverter(converter_cls);
  if (!conv) return -1001;
  meta.set_converter(converter_cls, 0, ailego::Params());
  int rc = conv->init(meta, ailego::Params());
  if (rc != 0) return rc;
  IndexHolder::Pointer holder = std::make_shared<BorrowedHolder>(meta.data_type(), dim, meta.element_size(), rows, keys, n);
  if ((rc = IndexConverter::TrainAndTransform(conv, holder)) != 0) return rc;
  IndexMeta cmeta = conv->meta();
  auto builder = IndexFactory::CreateBuilder(builder_cls);
  if (!builder) return -1002;
  if ((rc = builder->init(cmeta, bparams)) != 0) return rc;
  IndexHolder::Pointer conv_rows = conv->result();
  if (!conv_rows) return -1003;
  if ((rc = builder->train(conv_rows)) != 0 && rc != IndexError_NotImplemented) return rc;
  if ((rc = builder->build(conv_rows)) != 0) return rc;
  auto dumper = IndexFactory::CreateDumper("MemoryDumper");
  if (!dumper) return -1004;
  if ((rc = dumper->init(ailego::Params())) != 0) return rc;
  IndexMemory::Instance()->remove(target);
  if ((rc = dumper->create(target)) != 0) return rc;
  if ((rc = builder->dump(dumper)) != 0) return rc;
  return dumper->close();
}

// The search sequence of boundary A over any runner, with the reformer and the metric the index's own meta names (index.cc:87-108,
// 179-183).  batched = 0: Index::_dense_search once per query (index.cc:596-652: reformer->transform(1), search_impl(count = 1),
// metric->normalize, reformer->normalize) — what the product does today.  batched = 1: Index::SearchBatch of
// patches/boundary_a.diff: ONE reformer->transform(count), ONE search_impl(count), then per query the same two normalisations.
// `queries` are RAW (unconverted) rows of in_dtype / in_dim.  Results: [count][topk] keys / scores, counts.
int zref_search_sequence(void *h, void *c, int in_dtype, uint32_t in_dim, const void *queries, uint32_t count, int batched,
                         int linear, uint32_t topk, uint64_t *out_keys, float *out_scores, uint32_t *out_counts) {
  auto *r = static_cast<Runner *>(h);
  auto *x = static_cast<Ctx *>(c);
  const IndexMeta &im = r->searcher ? r->searcher->meta() : r->streamer->meta();
  auto metric = IndexFactory::CreateMetric(im.metric_name());
  if (!metric) return -1001;
  int rc = metric->init(im, im.metric_params());
  if (rc != 0) return rc;
  if (metric->query_metric()) metric = metric->query_metric();
  IndexReformer::Pointer reformer;
  if (!im.reformer_name().empty()) {
    reformer = IndexFactory::CreateReformer(im.reformer_name());
    if (!reformer) return -1002;
    if ((rc = reformer->init(im.reformer_params())) != 0) return rc;
  }
  IndexQueryMeta in_meta(in_dtype ? IndexMeta::DT_FP16 : IndexMeta::DT_FP32, in_dim);
  const size_t stride = in_meta.element_size();
  x->c->set_topk(topk);
  auto emit = [&](uint32_t q, IndexDocumentList list, const void *raw) -> int {
    if (metric->support_normalize())
      for (auto &d : list) metric->normalize(d.mutable_score());
    if (reformer && reformer->normalize(raw, in_meta, list) != 0) return IndexError_Runtime;
    const uint32_t m = (uint32_t)std::min<size_t>(list.size(), topk);
    out_counts[q] = m;
    for (uint32_t j = 0; j < m; ++j) {
      out_keys[(size_t)q * topk + j] = list[j].key();
      out_scores[(size_t)q * topk + j] = list[j].score();
    }
    return 0;
  };
  const char *qp = static_cast<const char *>(queries);
  if (batched) {
    const void *vectors = queries;
    std::string buf;
    IndexQueryMeta ometa = in_meta;
    if (reformer) {
      rc = reformer->transform(queries, in_meta, count, &buf, &ometa);
      if (rc == IndexError_Unsupported || rc == IndexError_NotImplemented) {
        // (CosineReformer has no batched transform, cosine_reformer.cc:146-150: one query at a time, rows concatenated)
        buf.clear();
        for (uint32_t q = 0; q < count; ++q) {
          std::string one;
          if ((rc = reformer->transform(qp + (size_t)q * stride, in_meta, &one, &ometa)) != 0) return rc;
          buf.append(one);
        }
      } else if (rc != 0) {
        return rc;
      }
      vectors = buf.data();
    }
    rc = linear ? r->get()->search_bf_impl(vectors, ometa, count, x->c) : r->get()->search_impl(vectors, ometa, count, x->c);
    if (rc != 0) return rc;
    for (uint32_t q = 0; q < count; ++q)
      if ((rc = emit(q, x->c->result(q), qp + (size_t)q * stride)) != 0) return rc;
    return 0;
  }
  for (uint32_t q = 0; q < count; ++q) {
    const void *raw = qp + (size_t)q * stride;
    const void *vector = raw;
    std::string buf;
    IndexQueryMeta ometa = in_meta;
    if (reformer) {
      if ((rc = reformer->transform(raw, in_meta, &buf, &ometa)) != 0) return rc;
      vector = buf.data();
    }
    rc = linear ? r->get()->search_bf_impl(vector, ometa, 1, x->c) : r->get()->search_impl(vector, ometa, 1, x->c);
    if (rc != 0) return rc;
    if ((rc = emit(q, x->c->result(), raw)) != 0) return rc;
  }
  return 0;
}

// ---- searches: mode 0 search_impl, 1 search_bf_impl, 2 search_bf_by_p_keys_impl (index_runner.h:490-585) ----------------
int zref_search(void *h, void *c, int mode, const void *q, int dtype, uint32_t dim, uint32_t count, const uint64_t *p_keys,
                const uint32_t *p_offs) {
  auto *r = static_cast<Runner *>(h);
  auto *x = static_cast<Ctx *>(c);
  IndexQueryMeta qm(dtype ? IndexMeta::DT_FP16 : IndexMeta::DT_FP32, dim);
  if (mode == 0) return r->get()->search_impl(q, qm, count, x->c);
  if (mode == 1) return r->get()->search_bf_impl(q, qm, count, x->c);
  if (mode == 2) {
    std::vector<std::vector<uint64_t>> pk(count);
    for (uint32_t i = 0; i < count; ++i) pk[i].assign(p_keys + p_offs[i], p_keys + p_offs[i + 1]);
    return r->get()->search_bf_by_p_keys_impl(q, pk, qm, count, x->c);
  }
  return IndexError_InvalidArgument;
}
uint32_t zref_ctx_result_size(void *c, uint32_t qi) { return (uint32_t)static_cast<Ctx *>(c)->c->result(qi).size(); }
// documents of query qi; vectors (nullable): elem_size bytes per document, from IndexDocument::vector()
int zref_ctx_result(void *c, uint32_t qi, uint64_t *keys, float *scores, uint32_t *index, void *vectors, uint32_t elem_size,
                    uint32_t *vectors_present) {
  const IndexDocumentList &lst = static_cast<Ctx *>(c)->c->result(qi);
  uint32_t present = 0;
  for (size_t j = 0; j < lst.size(); ++j) {
    keys[j] = lst[j].key();
    scores[j] = lst[j].score();
    if (index) index[j] = lst[j].index();
    if (vectors && lst[j].vector()) {
      memcpy(static_cast<char *>(vectors) + j * elem_size, lst[j].vector(), elem_size);
      ++present;
    }
  }
  if (vectors_present) *vectors_present = present;
  return 0;
}
uint32_t zref_ctx_group_count(void *c, uint32_t qi) { return (uint32_t)static_cast<Ctx *>(c)->c->group_result(qi).size(); }
// group s of query qi: its id (decimal, see zref_ctx_set_group) and documents; returns the document count, -1 past cap
int zref_ctx_group(void *c, uint32_t qi, uint32_t s, uint32_t *group, uint64_t *keys, float *scores, uint32_t cap) {
  const IndexGroupDocumentList &gl = static_cast<Ctx *>(c)->c->group_result(qi);
  if (s >= gl.size()) return -1;
  *group = (uint32_t)std::strtoul(gl[s].group_id().c_str(), nullptr, 10);
  const auto &docs = gl[s].docs();
  if (docs.size() > cap) return -1;
  for (size_t j = 0; j < docs.size(); ++j) {
    keys[j] = docs[j].key();
    scores[j] = docs[j].score();
  }
  return (int)docs.size();
}

// ---- the CPU baseline leg: queries dealt over T threads, each with its own context, ONE query per call — how the product
// calls boundary B (index.cc:605-619) and how tools/core/bench.cc:145-245 measures it.  Returns wall seconds. -------------
int zref_search_mt(void *h, int mode, const void *q, int dtype, uint32_t dim, uint32_t count, uint32_t topk, const char *ctx_params_json,
                   uint32_t threads, uint64_t *out_keys, float *out_scores, uint32_t *out_counts, double *seconds) {
  auto *r = static_cast<Runner *>(h);
  IndexQueryMeta qm(dtype ? IndexMeta::DT_FP16 : IndexMeta::DT_FP32, dim);
  ailego::Params cp;
  if (!parse_params(ctx_params_json, &cp)) return -1000;
  const bool has_cp = ctx_params_json && *ctx_params_json;
  std::atomic<uint32_t> next{0};
  std::atomic<int> err{0};
  const char *qp = static_cast<const char *>(q);
  auto work = [&]() {
    auto ctx = r->get()->create_context();
    if (!ctx) { err = -1; return; }
    if (has_cp) ctx->update(cp);
    ctx->set_topk(topk);
    for (;;) {
      uint32_t i = next.fetch_add(1);
      if (i >= count || err.load() != 0) break;
      int rc = mode == 1 ? r->get()->search_bf_impl(qp + size_t(i) * qm.element_size(), qm, ctx)
                         : r->get()->search_impl(qp + size_t(i) * qm.element_size(), qm, ctx);
      if (rc != 0) { err = rc; break; }
      const IndexDocumentList &lst = ctx->result();
      const uint32_t m = (uint32_t)std::min<size_t>(lst.size(), topk);
      out_counts[i] = m;
      for (uint32_t j = 0; j < m; ++j) {
        out_keys[size_t(i) * topk + j] = lst[j].key();
        out_scores[size_t(i) * topk + j] = lst[j].score();
      }
    }
  };
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> pool;
  for (uint32_t t = 1; t < threads; ++t) pool.emplace_back(work);
  work();
  for (auto &t : pool) t.join();
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return err.load();
}

}  // extern "C"
