// ref_plugin_shim.cc — TEST INFRASTRUCTURE ONLY.  Drives the plugin's "HipIVFBuilder" (plugin/hip_ivf_builder.cc) through the
// REFERENCE's own framework, compiled in place from /root/reference by oracle/Makefile (target `ref_plugin`, output
// oracle/_ref/libzvec_ref_plugin.so; no reference source copied, no stand-in headers): IndexFactory::CreateBuilder by its
// registered name, a MultiPassIndexHolder of the caller's rows, IndexBuilder::train / build / dump into the reference's
// MemoryDumper.  Returns the dumped index FILE image, which tests/test_gpu_plugin_builder.py opens with the loaders.
#include <zvec/core/framework/index_builder.h>
#include <zvec/core/framework/index_factory.h>
#include <zvec/core/framework/index_holder.h>
#include <zvec/core/framework/index_memory.h>

#include <atomic>
#include <cstring>
#include <string>

using namespace zvec;
using namespace zvec::core;

namespace {
template <IndexMeta::DataType DT, typename T>
IndexHolder::Pointer rows_holder(uint32_t dim, const void *rows, const uint64_t *keys, uint64_t n) {
  auto holder = std::make_shared<MultiPassIndexHolder<DT>>(dim);
  const T *p = static_cast<const T *>(rows);
  for (uint64_t i = 0; i < n; ++i) {
    ailego::NumericalVector<T> v(dim);
    memcpy(v.data(), p + (size_t)i * dim, sizeof(T) * dim);
    if (!holder->emplace(keys ? keys[i] : i, std::move(v))) return nullptr;
  }
  return holder;
}
}  // namespace

extern "C" int zref_plugin_ivf_build_and_dump(int dtype, uint32_t dim, const char *metric, const void *rows, const uint64_t *keys,
                                              uint64_t n, uint32_t nlist, uint32_t kmeans_iters, void *out, uint64_t cap,
                                              uint64_t *out_size) {
  IndexMeta meta(dtype ? IndexMeta::DT_FP16 : IndexMeta::DT_FP32, dim);
  meta.set_metric(std::string(metric), 0, ailego::Params());
  auto builder = IndexFactory::CreateBuilder("HipIVFBuilder");
  if (!builder) return -1001;
  ailego::Params params;
  params.set("proxima.ivf.builder.centroid_count", std::to_string(nlist));
  params.set("proxima.hip.builder.kmeans_iters", kmeans_iters);
  int rc = builder->init(meta, params);
  if (rc != 0) return rc;
  IndexHolder::Pointer holder = dtype ? rows_holder<IndexMeta::DT_FP16, ailego::Float16>(dim, rows, keys, n)
                                      : rows_holder<IndexMeta::DT_FP32, float>(dim, rows, keys, n);
  if (!holder) return -1002;
  if ((rc = builder->train(holder)) != 0) return rc;
  if ((rc = builder->build(holder)) != 0) return rc;
  auto dumper = IndexFactory::CreateDumper("MemoryDumper");
  if (!dumper) return -1003;
  static std::atomic<uint32_t> serial{0};
  const std::string path = "zref_plugin_" + std::to_string(serial.fetch_add(1));
  if ((rc = dumper->init(ailego::Params())) != 0 || (rc = dumper->create(path)) != 0) return rc;
  if ((rc = builder->dump(dumper)) != 0) return rc;
  if ((rc = dumper->close()) != 0) return rc;
  auto rope = IndexMemory::Instance()->open(path);
  if (!rope || rope->count() != 1) return -1004;
  const size_t size = (*rope)[0].size();
  *out_size = size;
  if (size > cap) return -1005;
  rc = (*rope)[0].fetch(0, out, size) == size ? 0 : -1006;
  IndexMemory::Instance()->remove(path);
  return rc;
}
