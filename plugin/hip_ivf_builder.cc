// plugin/hip_ivf_builder.cc — "HipIVFBuilder": the IVF build on the GPU behind zvec's IndexBuilder interface
// (src/include/zvec/core/framework/index_builder.h, index_runner.h:653-740), to live beside hip_plugin.cc in
// libzvec_hip_plugin.so.  Unlike the streamers / searchers, this class needs nothing of zvec that is unbuildable in this
// container, so besides the syntax check of __graft_entry__.build() it is LINKED against the reference's own framework
// sources compiled in place (oracle/Makefile target `ref_core`: oracle/_ref/libzvec_hip_plugin.so, test infrastructure) and RUN on the GPU box by
// tests/test_gpu_plugin_builder.py: factory registration, holder walk, GPU build, the reference's IVFDumper / FlatBuilder /
// IndexMeta writers — the file it dumps is then opened by the loaders.
#include <zvec/core/framework/index_builder.h>
#include <zvec/core/framework/index_dumper.h>
#include <zvec/core/framework/index_error.h>
#include <zvec/core/framework/index_factory.h>
#include <zvec/core/framework/index_helper.h>
#include <zvec/core/framework/index_holder.h>
#include <zvec/core/framework/index_memory.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <vector>

#include "core/algorithm/flat/flat_utility.h"
#include "core/algorithm/ivf/ivf_dumper.h"
#include "hip_plugin_common.h"

namespace zvec {
namespace core {

namespace {
const std::string kParamCentroidCount("proxima.ivf.builder.centroid_count");   // ivf_params.h:25-26 ("4096" or "64*64")
const std::string kParamHipKmeansIters("proxima.hip.builder.kmeans_iters");   // new: Lloyd rounds (default 20 = OptKmeansCluster's max_iterations, opt_kmeans_cluster.cc:115)
const std::string kParamTrainSampleCount("proxima.ivf.builder.train_sample_count");   // ivf_params.h:33-36: 0 / absent = train on every row
const std::string kParamTrainSampleRatio("proxima.ivf.builder.train_sample_ratio");
}  // namespace

/*! "HipIVFBuilder": stands where IVFBuilder is registered (ivf_builder.cc).  train + build = k-means (20 Lloyd rounds over every row by default, like the reference's trainer; a strided sample when train_sample_count / _ratio say so),
 *  nearest-centroid labels and list packing on the device (zvec_hip_ivf_build); dump writes the SAME index file as
 *  IVFBuilder::dump_index / dump (ivf_builder.cc:405-440,652-729) through the reference's own IVFDumper, with the centroid
 *  index as the image its FlatBuilder dumps (IVFCentroidIndex::build, ivf_centroid_index.cc:468-490) — so IVFSearcher /
 *  HipIVFSearcher load it like any other.  No quantizers, one level of centroids (what indexes/ivf_index.cc:31-60 asks for). */
class HipIVFBuilder : public IndexBuilder {
 public:
  ~HipIVFBuilder() override { cleanup(); }
  int init(const IndexMeta &meta, const ailego::Params &params) override {
    meta_ = meta;
    params_ = params;
    if (metric_of(meta) < 0 || dtype_of(meta) < 0) return IndexError_Unsupported;
    params.get(kParamHipDevice, &device_);
    params.get(kParamHipKmeansIters, &kmeans_iters_);
    params.get(kParamTrainSampleCount, &sample_count_);
    params.get(kParamTrainSampleRatio, &sample_ratio_);
    std::string spec;
    params.get(kParamCentroidCount, &spec);
    uint64_t total = spec.empty() ? 0 : 1;
    for (size_t a = 0; a < spec.size();) {               // "a*b*…": the leaf count is the product
      size_t b = spec.find('*', a);
      if (b == std::string::npos) b = spec.size();
      total *= std::strtoull(spec.substr(a, b - a).c_str(), nullptr, 10);
      a = b + 1;
    }
    nlist_ = (uint32_t)total;
    return 0;
  }
  int cleanup() override {
    if (h_) zvec_hip_ivf_destroy(h_);
    h_ = nullptr;
    rows_.clear();
    keys_.clear();
    return 0;
  }
  const Stats &stats() const override { return stats_; }
  //! k-means runs inside build (one pass over the holder feeds both); train only checks the holder
  int train(IndexThreads::Pointer, IndexHolder::Pointer holder) override {
    if (!holder || !holder->is_matched(meta_)) return IndexError_Mismatch;
    stats_.set_trained_count(holder->count());
    return 0;
  }
  int train(const IndexTrainer::Pointer &) override { return IndexError_NotImplemented; }
  int build(IndexThreads::Pointer, IndexHolder::Pointer holder) override {
    if (!holder || !holder->is_matched(meta_)) return IndexError_Mismatch;
    const size_t es = meta_.element_size();
    rows_.clear();
    keys_.clear();
    for (auto it = holder->create_iterator(); it && it->is_valid(); it->next()) {
      rows_.append(static_cast<const char *>(it->data()), es);
      keys_.push_back(it->key());
    }
    const uint64_t n = keys_.size();
    if (n == 0) return IndexError_NoExist;
    // the reference's default when no count is given: about sqrt(n) lists (ivf_builder.cc centroid auto-tuning aside)
    uint32_t nlist = nlist_ ? nlist_ : (uint32_t)std::max<double>(1.0, std::floor(std::sqrt((double)n)));
    nlist = (uint32_t)std::min<uint64_t>(nlist, n);
    int rc = zvec_hip_ivf_create(meta_.dimension(), dtype_of(meta_), metric_of(meta_), device_, &h_);
    if (rc != 0) return rc;
    // training sample as StratifiedClusterTrainer takes it (stratified_cluster_trainer.cc:150-185): max(sample_count,
    // sample_ratio * n) rows when either is set, every row otherwise — the reference's default, and what the cluster-quality
    // fixture (tests/golden/kmeans_quality.json) pins the GPU trainer against
    uint64_t sample = std::max<uint64_t>(sample_count_, (uint64_t)((double)sample_ratio_ * (double)n));
    if (sample == 0 || sample > n) sample = n;
    const uint32_t per_list = (uint32_t)std::min<uint64_t>((sample + nlist - 1) / nlist, 0xffffffffu);
    if ((rc = zvec_hip_ivf_build(h_, rows_.data(), n, keys_.data(), nlist, kmeans_iters_, per_list, 20260320ull)) != 0) return rc;
    nlist_built_ = nlist;
    stats_.set_built_count(n);
    return 0;
  }
  int dump(const IndexDumper::Pointer &dumper) override {
    if (!h_ || !dumper) return IndexError_Runtime;                     // "Build the index before dump"
    const size_t es = meta_.element_size();
    uint64_t n = 0;
    uint32_t nlist = 0;
    int rc = zvec_hip_ivf_info(h_, &n, &nlist);
    if (rc != 0) return rc;
    std::string centroids(size_t(nlist) * es, '\0');
    std::vector<uint64_t> offs(size_t(nlist) + 1), row_of(n);
    if ((rc = zvec_hip_ivf_export(h_, &centroids[0], offs.data(), row_of.data())) != 0) return rc;
    IndexMeta imeta = meta_;
    imeta.set_major_order(IndexMeta::MO_ROW);
    {
      IVFDumper ivf(imeta, dumper, nlist);
      for (uint32_t l = 0; l < nlist; ++l)
        for (uint64_t i = offs[l]; i < offs[l + 1]; ++i) {
          const uint64_t r = row_of[i];                                // original row of this list-order position
          if ((rc = ivf.dump_inverted_vector(l, keys_[r], rows_.data() + r * es)) != 0) return rc;
        }
      if ((rc = ivf.dump_inverted_vector_finished()) != 0) return rc;
      if ((rc = ivf.dump_quantizer_params({})) != 0) return rc;
      // the centroid index: a flat index of the centroids, key = centroid id, as FlatBuilder dumps it into memory
      auto holder = std::make_shared<OnePassRows>(imeta, centroids.data(), nlist);
      auto fb = IndexFactory::CreateBuilder("FlatBuilder");
      auto md = IndexFactory::CreateDumper("MemoryDumper");
      if (!fb || !md) return IndexError_NoExist;
      ailego::Params fp;
      fp.set(PARAM_FLAT_COLUMN_MAJOR_ORDER, false);
      if ((rc = fb->init(imeta, fp)) != 0) return rc;
      rc = fb->train(holder);
      if (rc != 0 && rc != IndexError_NotImplemented) return rc;
      if ((rc = fb->build(holder)) != 0) return rc;
      const std::string path = "hip_ivf_centroids_" + std::to_string(reinterpret_cast<uintptr_t>(this));
      if ((rc = md->init(ailego::Params())) != 0 || (rc = md->create(path)) != 0) return rc;
      if ((rc = fb->dump(md)) != 0 || (rc = md->close()) != 0) return rc;
      auto rope = IndexMemory::Instance()->open(path);
      if (!rope || rope->count() != 1) return IndexError_Runtime;
      const void *image = nullptr;
      (*rope)[0].read(0, &image, 0);
      rc = ivf.dump_centroid_index(image, (*rope)[0].size());
      IndexMemory::Instance()->remove(path);
      if (rc != 0) return rc;
      stats_.set_dumped_count(ivf.dumped_count());
    }
    // searcher defaults as IVFBuilder::dump sets them (ivf_builder.cc:418-426)
    float scan_ratio = std::max(-0.004f * (float)std::log((double)n) + 0.0751f, 0.0001f);
    ailego::Params sp;
    sp.set(kParamScanRatio, scan_ratio);
    imeta.set_searcher("IVFSearcher", 0, std::move(sp));
    imeta.set_builder("HipIVFBuilder", 0, ailego::Params(params_));
    return IndexHelper::SerializeToDumper(imeta, dumper.get());
  }

 private:
  //! rows of a contiguous host matrix as a (multi-pass) holder; key = row number
  class OnePassRows : public IndexHolder {
   public:
    OnePassRows(const IndexMeta &m, const char *rows, size_t n) : meta_(m), rows_(rows), n_(n) {}
    size_t count() const override { return n_; }
    size_t dimension() const override { return meta_.dimension(); }
    IndexMeta::DataType data_type() const override { return meta_.data_type(); }
    size_t element_size() const override { return meta_.element_size(); }
    bool multipass() const override { return true; }
    Iterator::Pointer create_iterator() override { return Iterator::Pointer(new It(this)); }
   private:
    struct It : public Iterator {
      explicit It(const OnePassRows *o) : o_(o) {}
      const void *data() const override { return o_->rows_ + i_ * o_->meta_.element_size(); }
      bool is_valid() const override { return i_ < o_->n_; }
      uint64_t key() const override { return i_; }
      void next() override { ++i_; }
      const OnePassRows *o_;
      size_t i_{0};
    };
    IndexMeta meta_;
    const char *rows_;
    size_t n_;
  };

  IndexMeta meta_;
  ailego::Params params_;
  Stats stats_;
  int device_{0};
  uint32_t nlist_{0}, nlist_built_{0}, kmeans_iters_{20}, sample_count_{0};
  float sample_ratio_{0.0f};
  zvec_hip_ivf_t h_{nullptr};
  std::string rows_;                      // the holder's rows, kept for the dump (the reference keeps the holder)
  std::vector<uint64_t> keys_;
};

static_assert(!std::is_abstract<HipIVFBuilder>::value, "every pure virtual of IndexBuilder / IndexRunner is implemented");

INDEX_FACTORY_REGISTER_BUILDER_ALIAS(HipIVFBuilder, HipIVFBuilder);

}  // namespace core
}  // namespace zvec
