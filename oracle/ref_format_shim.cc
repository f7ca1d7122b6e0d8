// ref_format_shim.cc — TEST INFRASTRUCTURE ONLY.  extern "C" doors onto the reference's own index WRITERS, compiled IN
// PLACE from /root/reference by oracle/Makefile (target `ref_core`, part of oracle/_ref/libzvec_ref_core.so; nothing
// of the reference is copied, no stand-in headers, the reference's build system is not used):
//   FlatBuilder<32>::init / build / dump          src/core/algorithm/flat/flat_builder.cc:22-276
//   IVFDumper                                      src/core/algorithm/ivf/ivf_dumper.{h,cc}
//   MemoryDumper + IndexPacker / IndexFormat       src/core/utility/memory_dumper.cc, index_packer.h, index_format.h
//   IndexMeta::serialize, IndexHelper              src/core/framework/index_meta.cc, index_helper.cc
// They exist to make GOLDEN index files (tests/golden/make_ref_index_files.py): byte images of a flat index and of an
// IVF index exactly as the reference dumps them, which pin the container parser and the segment loaders of the product
// (SURVEY §8(f) next-2).  The IVF image is assembled with the call sequence of IVFBuilder::dump_index / dump
// (ivf_builder.cc:405-440,652-729): inverted vectors list by list, finish, (no quantizer params), the centroid index as
// the image a FlatBuilder dumps into a MemoryDumper (IVFCentroidIndex::build, ivf_centroid_index.cc:468-490), then the
// IndexMeta.  k-means / labelling are NOT run here: the caller passes centroids and lists, so the files hold exactly the
// structure the test states (empty, ragged and trailing-empty lists).  Whole IVFBuilder runs: ref_core_shim.cc zref_build.
#include <zvec/core/framework/index_factory.h>
#include <zvec/core/framework/index_helper.h>
#include <zvec/core/framework/index_holder.h>
#include <zvec/core/framework/index_memory.h>

#include <atomic>
#include <cstring>
#include <string>

#include "core/algorithm/flat/flat_utility.h"
#include "core/algorithm/ivf/ivf_dumper.h"

using namespace zvec;
using namespace zvec::core;

namespace {

std::string fresh_path() {
  static std::atomic<uint32_t> n{0};
  return "zref_format_" + std::to_string(n.fetch_add(1));
}

IndexMeta make_meta(int dtype, uint32_t dim, int column_major, const char *metric) {
  IndexMeta meta(dtype ? IndexMeta::DT_FP16 : IndexMeta::DT_FP32, dim);
  meta.set_metric(std::string(metric), 0, ailego::Params());
  meta.set_major_order(column_major ? IndexMeta::MO_COLUMN : IndexMeta::MO_ROW);
  return meta;
}

template <IndexMeta::DataType DT, typename T>
IndexHolder::Pointer make_holder(uint32_t dim, const void *rows, const uint64_t *keys, uint64_t n) {
  auto holder = std::make_shared<MultiPassIndexHolder<DT>>(dim);
  const T *p = static_cast<const T *>(rows);
  for (uint64_t i = 0; i < n; ++i) {
    ailego::NumericalVector<T> v(dim);
    memcpy(v.data(), p + (size_t)i * dim, sizeof(T) * dim);
    if (!holder->emplace(keys ? keys[i] : i, std::move(v))) return nullptr;
  }
  return holder;
}

// FlatBuilder -> MemoryDumper; returns the rope path holding the file image
int dump_flat_to_memory(int dtype, uint32_t dim, int column_major, const char *metric, const void *rows, const uint64_t *keys,
                        uint64_t n, std::string *path) {
  IndexMeta meta = make_meta(dtype, dim, column_major, metric);
  ailego::Params params;
  params.set(PARAM_FLAT_COLUMN_MAJOR_ORDER, column_major != 0);
  auto builder = IndexFactory::CreateBuilder("FlatBuilder");
  if (!builder) return -1001;
  int rc = builder->init(meta, params);
  if (rc != 0) return rc;
  IndexHolder::Pointer holder = dtype ? make_holder<IndexMeta::DT_FP16, ailego::Float16>(dim, rows, keys, n)
                                      : make_holder<IndexMeta::DT_FP32, float>(dim, rows, keys, n);
  if (!holder) return -1002;
  if ((rc = builder->train(holder)) != 0 && rc != IndexError_NotImplemented) return rc;
  if ((rc = builder->build(holder)) != 0) return rc;
  auto dumper = IndexFactory::CreateDumper("MemoryDumper");
  if (!dumper) return -1003;
  if ((rc = dumper->init(ailego::Params())) != 0) return rc;
  *path = fresh_path();
  if ((rc = dumper->create(*path)) != 0) return rc;
  if ((rc = builder->dump(dumper)) != 0) return rc;
  return dumper->close();
}

int copy_out(const std::string &path, void *out, uint64_t cap, uint64_t *out_size) {
  auto rope = IndexMemory::Instance()->open(path);
  if (!rope || rope->count() != 1) return -1004;
  const size_t size = (*rope)[0].size();
  *out_size = size;
  if (size > cap) return -1005;
  return (*rope)[0].fetch(0, out, size) == size ? 0 : -1006;
}

}  // namespace

extern "C" {

// a flat index FILE image as FlatBuilder<32> + MemoryDumper write it (IndexMeta, flat.keys, flat.features, …)
int zref_dump_flat_index(int dtype, uint32_t dim, int column_major, const char *metric, const void *rows, const uint64_t *keys,
                         uint64_t n, void *out, uint64_t cap, uint64_t *out_size) {
  std::string path;
  int rc = dump_flat_to_memory(dtype, dim, column_major, metric, rows, keys, n, &path);
  if (rc != 0) return rc;
  return copy_out(path, out, cap, out_size);
}

// an IVF index FILE image: rows in list order (list l = rows [list_offsets[l], list_offsets[l+1])), centroids [nlist][dim]
int zref_dump_ivf_index(int dtype, uint32_t dim, int column_major, int centroid_column_major, const char *metric,
                        const void *centroids, uint32_t nlist, const uint64_t *list_offsets, const void *rows,
                        const uint64_t *keys, void *out, uint64_t cap, uint64_t *out_size) {
  IndexMeta meta = make_meta(dtype, dim, column_major, metric);
  auto dumper = IndexFactory::CreateDumper("MemoryDumper");
  if (!dumper) return -1003;
  int rc = dumper->init(ailego::Params());
  if (rc != 0) return rc;
  const std::string path = fresh_path();
  if ((rc = dumper->create(path)) != 0) return rc;
  {
    IVFDumper ivf(meta, dumper, nlist);
    const size_t es = meta.element_size();
    const char *p = static_cast<const char *>(rows);
    for (uint32_t l = 0; l < nlist; ++l)
      for (uint64_t i = list_offsets[l]; i < list_offsets[l + 1]; ++i)
        if ((rc = ivf.dump_inverted_vector(l, keys ? keys[i] : i, p + (size_t)i * es)) != 0) return rc;
    if ((rc = ivf.dump_inverted_vector_finished()) != 0) return rc;
    if ((rc = ivf.dump_quantizer_params({})) != 0) return rc;
    std::string cpath;
    if ((rc = dump_flat_to_memory(dtype, dim, centroid_column_major, metric, centroids, nullptr, nlist, &cpath)) != 0) return rc;
    auto crope = IndexMemory::Instance()->open(cpath);
    if (!crope || crope->count() != 1) return -1004;
    const void *cdata = nullptr;
    (*crope)[0].read(0, &cdata, 0);
    if ((rc = ivf.dump_centroid_index(cdata, (*crope)[0].size())) != 0) return rc;
  }
  ailego::Params sp;
  sp.set("proxima.ivf.searcher.scan_ratio", 0.01f);
  meta.set_searcher("IVFSearcher", 0, std::move(sp));
  meta.set_builder("IVFBuilder", 0, ailego::Params());
  if ((rc = IndexHelper::SerializeToDumper(meta, dumper.get())) != 0) return rc;
  if ((rc = dumper->close()) != 0) return rc;
  return copy_out(path, out, cap, out_size);
}

}  // extern "C"
