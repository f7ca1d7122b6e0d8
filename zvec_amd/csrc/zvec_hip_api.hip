// zvec_hip_api.hip — host side of the C ABI in include/zvec_hip.h: HBM-resident stores, search
// orchestration (prep -> [coarse -> plan] -> scan -> merge), IVF build, and the measurement hook.
// gfx950 only; no CPU fallback exists anywhere in this file: if HIP is unavailable the calls fail.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <sched.h>
#include <time.h>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <shared_mutex>
#include <new>
#include <vector>

#include "../../include/zvec_hip.h"
#include "scan_kernels.hip.h"

using namespace zvk;

#include "api_types.inc.h"
#include "api_flat_scan.inc.h"
#include "api_ivf_core.inc.h"

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

#include "api_entry_ctx_flat.inc.h"
#include "api_entry_ivf.inc.h"
#include "api_entry_merge_prof.inc.h"
#include "api_entry_filter.inc.h"
#include "api_entry_group.inc.h"
#include "api_entry_shards.inc.h"
#include "api_entry_container.inc.h"

}  // extern "C"
