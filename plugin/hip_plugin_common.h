// hip_plugin_common.h — what the plugin translation units share: parameter keys and the IndexMeta -> C ABI enums.
#pragma once
#include <string>

#include <zvec/core/framework/index_meta.h>

#include "zvec_hip.h"

namespace zvec {
namespace core {
namespace {

const std::string kParamScanRatio("proxima.ivf.searcher.scan_ratio");   // ivf_params.h:44-45
const std::string kParamHipDevice("proxima.hip.device");                // new: HIP device ordinal

int metric_of(const IndexMeta &meta) {          // names chosen in src/core/interface/index.cc:47-106
  const std::string &m = meta.metric_name();
  if (m == "SquaredEuclidean") return ZVEC_HIP_METRIC_L2;
  if (m == "InnerProduct") return ZVEC_HIP_METRIC_IP;
  if (m == "Cosine") return ZVEC_HIP_METRIC_COSINE;
  return -1;
}

int dtype_of(const IndexMeta &meta) {
  if (meta.data_type() == IndexMeta::DT_FP32) return ZVEC_HIP_DT_FP32;
  if (meta.data_type() == IndexMeta::DT_FP16) return ZVEC_HIP_DT_FP16;
  return -1;
}

}  // namespace
}  // namespace core
}  // namespace zvec
