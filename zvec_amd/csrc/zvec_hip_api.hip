// zvec_hip_api.hip — host side of the C ABI in include/zvec_hip.h: HBM-resident stores, search
// orchestration (prep -> [coarse -> plan] -> scan -> merge), IVF build, and the measurement hook.
// gfx950 only; no CPU fallback exists anywhere in this file: if HIP is unavailable the calls fail.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/zvec_hip.h"
#include "scan_kernels.hip.h"

using namespace zvk;

#define ZCHK(expr)                                                                               \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      fprintf(stderr, "[zvec_hip] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e),        \
              __FILE__, __LINE__);                                                               \
      return (_e == hipErrorOutOfMemory) ? ZVEC_HIP_ERR_NO_MEMORY : ZVEC_HIP_ERR_RUNTIME;        \
    }                                                                                            \
  } while (0)

#define ZRET(expr)            \
  do {                        \
    int _r = (expr);          \
    if (_r != 0) return _r;   \
  } while (0)

namespace {

constexpr size_t LDS_LIMIT = 160 * 1024;
constexpr int PROFILE_MAX = 8192;

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 4 + 256;
    ZCHK(hipMalloc(&p, want));
    cap = want;
    return 0;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// scope-owned device temporary: freed on every exit path of the enclosing function
template <typename T>
struct Scoped {
  T *p = nullptr;
  Scoped() {}
  Scoped(const Scoped &) = delete;
  Scoped &operator=(const Scoped &) = delete;
  ~Scoped() { if (p) (void)hipFree(p); }
  int alloc(size_t count) {
    ZCHK(hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T)));
    return 0;
  }
  operator T *() const { return p; }
};

// a blocked, HBM-resident set of rows (flat store, IVF centroids, IVF inverted lists)
struct Store {
  uint32_t dim_in = 0;   // element dimension at the ABI (cosine: d+1)
  uint32_t dscan = 0;    // scanned dims
  uint32_t dpad = 0;     // 4-byte WORDS per stored row, multiple of 32 (fp32: dscan up to 32; fp16: dscan up to 64, halved)
  uint32_t elem = 4;     // bytes per element: 4 (fp32) or 2 (fp16)
  bool f16 = false;
  int metric = 0;
  uint64_t n = 0;        // padded positions in use
  uint64_t cap_tiles = 0;
  float *base = nullptr;
  float *bnorm = nullptr;
  float *extra = nullptr;   // cosine: stored norm column
  uint64_t *keys = nullptr;

  void configure(uint32_t dim, int met, int dtype = ZVEC_HIP_DT_FP32) {
    dim_in = dim;
    metric = met;
    f16 = (dtype == ZVEC_HIP_DT_FP16);
    elem = f16 ? 2 : 4;
    // cosine rows end with the fp32 norm of the original vector: 1 float, or 2 half slots (cosine_converter.cc:205-212)
    dscan = (met == ZVEC_HIP_METRIC_COSINE) ? dim - (f16 ? 2 : 1) : dim;
    dpad = f16 ? ((dscan + 63) / 64 * 64) / 2 : (dscan + TILE_K - 1) / TILE_K * TILE_K;
  }
  size_t row_bytes() const { return (size_t)dim_in * elem; }
  int reserve(uint64_t rows, hipStream_t stream) {
    uint64_t tiles = (rows + TILE_N - 1) / TILE_N;
    if (tiles <= cap_tiles) return 0;
    uint64_t nt = std::max<uint64_t>(tiles, cap_tiles + cap_tiles / 2 + 1);
    float *nb = nullptr, *nn = nullptr, *ne = nullptr;
    uint64_t *nk = nullptr;
    ZCHK(hipMalloc(&nb, (size_t)nt * TILE_N * dpad * sizeof(float)));
    ZCHK(hipMalloc(&nn, (size_t)nt * TILE_N * sizeof(float)));
    ZCHK(hipMalloc(&nk, (size_t)nt * TILE_N * sizeof(uint64_t)));
    if (metric == ZVEC_HIP_METRIC_COSINE) ZCHK(hipMalloc(&ne, (size_t)nt * TILE_N * sizeof(float)));
    uint64_t used_tiles = (n + TILE_N - 1) / TILE_N;
    if (used_tiles) {
      ZCHK(hipMemcpyAsync(nb, base, (size_t)used_tiles * TILE_N * dpad * sizeof(float), hipMemcpyDeviceToDevice, stream));
      ZCHK(hipMemcpyAsync(nn, bnorm, (size_t)used_tiles * TILE_N * sizeof(float), hipMemcpyDeviceToDevice, stream));
      ZCHK(hipMemcpyAsync(nk, keys, (size_t)used_tiles * TILE_N * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream));
      if (ne) ZCHK(hipMemcpyAsync(ne, extra, (size_t)used_tiles * TILE_N * sizeof(float), hipMemcpyDeviceToDevice, stream));
      ZCHK(hipStreamSynchronize(stream));
    }
    release();
    base = nb; bnorm = nn; keys = nk; extra = ne; cap_tiles = nt;
    return 0;
  }
  void release() {
    if (base) (void)hipFree(base);
    if (bnorm) (void)hipFree(bnorm);
    if (extra) (void)hipFree(extra);
    if (keys) (void)hipFree(keys);
    base = bnorm = extra = nullptr; keys = nullptr; cap_tiles = 0;
  }
};

}  // namespace

struct zvec_hip_ctx_s {
  int device = 0;
  hipStream_t own = nullptr;
  hipStream_t cur = nullptr;
  std::mutex mu;
  // workspace
  DevBuf gtau, ridx;
  DevBuf seed_keys, seed_scores, seed_counts;   // sample scan that seeds the shared admission bounds
  DevBuf cmp_base, cmp_norm, cmp_extra, cmp_keys, cmp_pos, cmp_cnt;   // compacted keep-set (sparse filters)
  DevBuf qpad, qnorm, part_s, part_i, coarse_keys, coarse_scores, coarse_idx, coarse_cnt;
  DevBuf plan;        // all u32 plan arrays
  DevBuf io_q, io_ex, io_keys, io_scores, io_counts;   // staging for host-pointer entry points
  DevBuf stats;       // per-launch {distinct_rows, pair_rows} u64 x PROFILE_MAX
  uint32_t *q_scanned = nullptr, *q_nprobe = nullptr;  // inside plan
  uint32_t *last_list_count = nullptr;                 // inside plan
  uint32_t last_count = 0;
  // profiling
  bool profile = false;
  std::vector<hipEvent_t> ev0, ev1;
  std::vector<double> host_bytes, host_flops;   // flat launches: known on the host
  std::vector<int> launch_is_ivf;
  std::vector<uint32_t> prof_dscan;
  int nprof = 0;
  int cus = 0;
};

struct zvec_hip_flat_s {
  int device = 0;
  int dtype = 0;
  Store st;
  zvec_hip_ctx_s *defctx = nullptr;
  std::mutex mu;
};

struct zvec_hip_ivf_s {
  int device = 0;
  int dtype = 0;
  uint32_t dim = 0;
  int metric = 0;
  uint32_t nlist = 0;
  uint32_t shard = 0, nshards = 1;
  bool loaded = false;
  Store cent;     // centroids as a flat store
  Store lists;    // inverted lists, each padded to whole tiles
  uint64_t count_local = 0, count_global = 0;
  std::vector<uint32_t> h_size, h_size_global, h_tile0;
  std::vector<uint64_t> h_dense0;      // local dense offsets (nlist+1)
  std::vector<uint64_t> h_row_ids;     // local dense position -> original row
  std::vector<char> h_centroids;       // [nlist][dim] in the index element type
  uint32_t *d_size = nullptr, *d_size_global = nullptr, *d_tile0 = nullptr, *d_order = nullptr, *d_tail = nullptr;
  uint32_t tiles_per_chunk = 8;
  std::vector<uint32_t> h_tail;        // 1 = list belongs to the tail of the deal order (shorter chunks)
  uint64_t local_tiles = 0;            // tiles of the lists held by this shard
  uint64_t *d_dense0 = nullptr;
  zvec_hip_ctx_s *defctx = nullptr;
  std::mutex mu;
};

namespace {

struct KernelInfo {
  bool init = false;
  int cus = 0;
};
KernelInfo g_info[16];
std::mutex g_info_mu;

template <int NG, bool M16, bool EXCL, bool F16>
int launch_scan_t(const ScanArgs &a, uint32_t max_items, int cus, hipStream_t stream) {
  static bool attr_set[16] = {false};
  size_t lds = scan_lds_bytes(NG, a.k, M16);
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_set[dev & 15]) {
    ZCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_kernel<NG, M16, EXCL, F16>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT));
    attr_set[dev & 15] = true;
  }
  int occ = 0;
  ZCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, scan_kernel<NG, M16, EXCL, F16>, 256, lds));
  if (occ < 1) occ = 1;
  uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)max_items, (uint64_t)cus * (uint64_t)occ);
  if (grid == 0) return 0;
  hipLaunchKernelGGL((scan_kernel<NG, M16, EXCL, F16>), dim3(grid), dim3(256), lds, stream, a);
  ZCHK(hipGetLastError());
  return 0;
}

template <int NG, bool M16>
int launch_scan(const ScanArgs &a, bool f16, uint32_t max_items, int cus, hipStream_t stream) {
  if (f16)
    return a.exclude ? launch_scan_t<NG, M16, true, true>(a, max_items, cus, stream)
                     : launch_scan_t<NG, M16, false, true>(a, max_items, cus, stream);
  return a.exclude ? launch_scan_t<NG, M16, true, false>(a, max_items, cus, stream)
                   : launch_scan_t<NG, M16, false, false>(a, max_items, cus, stream);
}

// the 8-wave 128x128 flat tile (scan8_kernel); *occ_out = work-groups per CU it reaches for this k
template <bool EXCL, bool F16, bool GATHER>
int launch_scan8_t(const ScanArgs &a, uint32_t max_items, int cus, hipStream_t stream, int *occ_out) {
  static bool attr_set[16] = {false};
  size_t lds = scan8_lds_bytes(a.k);
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_set[dev & 15]) {
    ZCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&scan8_kernel<EXCL, F16, GATHER>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT));
    attr_set[dev & 15] = true;
  }
  int occ = 0;
  ZCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, scan8_kernel<EXCL, F16, GATHER>, 512, lds));
  if (occ < 1) occ = 1;
  if (occ_out) { *occ_out = occ; return 0; }
  uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)max_items, (uint64_t)cus * (uint64_t)occ);
  if (grid == 0) return 0;
  hipLaunchKernelGGL((scan8_kernel<EXCL, F16, GATHER>), dim3(grid), dim3(512), lds, stream, a);
  ZCHK(hipGetLastError());
  return 0;
}

// (the GATHER variant scans an already filtered position list: no exclude set)
int launch_scan8(const ScanArgs &a, bool f16, uint32_t max_items, int cus, hipStream_t stream, int *occ_out = nullptr) {
  if (a.gather_pos)
    return f16 ? launch_scan8_t<false, true, true>(a, max_items, cus, stream, occ_out)
               : launch_scan8_t<false, false, true>(a, max_items, cus, stream, occ_out);
  if (f16)
    return a.exclude ? launch_scan8_t<true, true, false>(a, max_items, cus, stream, occ_out)
                     : launch_scan8_t<false, true, false>(a, max_items, cus, stream, occ_out);
  return a.exclude ? launch_scan8_t<true, false, false>(a, max_items, cus, stream, occ_out)
                   : launch_scan8_t<false, false, false>(a, max_items, cus, stream, occ_out);
}

// ng == 0 selects the 16-row-halves (16x16 MFMA) shape
int launch_scan_ng(int ng, const ScanArgs &a, bool f16, uint32_t max_items, int cus, hipStream_t stream) {
  switch (ng) {
    case 0: return launch_scan<1, true>(a, f16, max_items, cus, stream);
    case 1: return launch_scan<1, false>(a, f16, max_items, cus, stream);
    case 2: return launch_scan<2, false>(a, f16, max_items, cus, stream);
    case 4: return launch_scan<4, false>(a, f16, max_items, cus, stream);
  }
  return ZVEC_HIP_ERR_INVALID_ARGUMENT;
}

// Tuning / test knobs, read once from the environment.  None is needed in production; they exist so that kernel
// variants can be A/B-timed on one GPU box (tools/ab_flat.sh) and so that tests can force a path onto small inputs.
struct Knobs {
  int max_ng = 4;             // ZVEC_HIP_MAX_NG      cap of the 4-wave kernel's query-row groups (1, 2, 4)
  bool no_wide = false;       // ZVEC_HIP_NO_WIDE     never take the 8-wave flat tile
  bool force_wide = false;    // ZVEC_HIP_FORCE_WIDE  take it on cache-resident bases too (tests)
  bool no_seed = false;       // ZVEC_HIP_NO_SEED     no prefix scan to seed the admission bounds
  bool no_gather = false;     // ZVEC_HIP_NO_GATHER   sparse filters: compact the kept rows instead of gathering them
  int ivf_tpc = 0;            // ZVEC_HIP_IVF_TPC     fixed tiles per IVF chunk (0 = adaptive)
  Knobs() {
    if (const char *e = getenv("ZVEC_HIP_MAX_NG")) max_ng = std::max(1, std::min(4, atoi(e)));
    no_wide = getenv("ZVEC_HIP_NO_WIDE") != nullptr;
    force_wide = getenv("ZVEC_HIP_FORCE_WIDE") != nullptr;
    no_seed = getenv("ZVEC_HIP_NO_SEED") != nullptr;
    no_gather = getenv("ZVEC_HIP_NO_GATHER") != nullptr;
    if (const char *e = getenv("ZVEC_HIP_IVF_TPC")) ivf_tpc = std::max(1, atoi(e));
  }
};
const Knobs &knobs() {
  static const Knobs k;
  return k;
}

int pick_ng(uint32_t rows_wanted, uint32_t k) {
  int ng = knobs().max_ng;
  // 128 query rows per work-group is the largest tile whose accumulators + staging fit 512 registers
  while (ng > 1 && (uint32_t)(ng / 2) * QGROUP >= rows_wanted) ng /= 2;
  while (ng >= 1 && scan_lds_bytes(ng, k) > LDS_LIMIT - 1024) ng /= 2;
  return ng;  // 0 => k too large for the LDS-resident lists
}

int device_cus(zvec_hip_ctx_s *ctx) {
  if (ctx->cus == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess) ctx->cus = prop.multiProcessorCount;
    if (ctx->cus <= 0) ctx->cus = 256;
  }
  return ctx->cus;
}

int prof_begin(zvec_hip_ctx_s *ctx, hipStream_t stream, double bytes, double flops, int is_ivf) {
  if (!ctx->profile || ctx->nprof >= PROFILE_MAX) return -1;
  int i = ctx->nprof;
  if ((int)ctx->ev0.size() <= i) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
    ctx->ev0.push_back(a);
    ctx->ev1.push_back(b);
    ctx->host_bytes.push_back(0);
    ctx->host_flops.push_back(0);
    ctx->launch_is_ivf.push_back(0);
    ctx->prof_dscan.push_back(0);
  }
  ctx->host_bytes[i] = bytes;
  ctx->host_flops[i] = flops;
  ctx->launch_is_ivf[i] = is_ivf;
  (void)hipEventRecord(ctx->ev0[i], stream);
  return i;
}
void prof_end(zvec_hip_ctx_s *ctx, hipStream_t stream, int i) {
  if (i < 0) return;
  (void)hipEventRecord(ctx->ev1[i], stream);
  ctx->nprof = i + 1;
}

// Outputs of a search on the device
struct SearchOut {
  uint64_t *keys;
  float *scores;
  uint32_t *idx;     // optional positions
  uint32_t *counts;
};

// flat scan of `st` for `count` prepared queries (ctx->qpad / qnorm already filled)
int refine_l2(zvec_hip_ctx_s *ctx, const Store &st, uint32_t count, uint32_t topk, float threshold, uint64_t *keys,
              float *scores, uint32_t *idx, uint32_t *counts, hipStream_t stream);

// partial-list merges of small batches: four waves per query gather the survivors (see merge_kernel)
inline uint32_t merge_threads(uint32_t count) { return count <= 256 ? 256u : 64u; }

// Sparse keep-set scan WITHOUT copying the kept rows: the wide kernel fetches the rows of a logical tile straight from
// their stored positions (LDS-DMA with per-lane source addresses: every 128-byte row segment is still one full line).
// `d_pos`: ascending kept positions, padded to whole tiles (+1 tile) with position 0; `kept` logical rows.
int flat_scan_gather(zvec_hip_ctx_s *ctx, const Store &st, const uint32_t *d_pos, uint32_t kept, uint32_t count,
                     uint32_t topk, float threshold, const SearchOut &out, hipStream_t stream, bool profile_it) {
  const int cus = device_cus(ctx);
  ScanArgs a{};
  a.base = st.base; a.bnorm = st.bnorm; a.exclude = nullptr; a.gather_pos = d_pos;
  a.queries = ctx->qpad.as<float>(); a.qnorm = ctx->qnorm.as<float>();
  a.dpad = st.dpad; a.nks = st.dpad / TILE_K; a.metric = st.metric; a.threshold = threshold;
  a.gtau = ctx->gtau.as<uint32_t>();
  a.mode = 0; a.nq = count;
  const uint32_t nqtiles = (count + W8_ROWS - 1) / W8_ROWS;
  // seeded bounds from the first SEED rows of the kept set (see flat_scan_prepared)
  constexpr uint32_t SEED_ROWS = 4096;
  if (kept >= 64 * SEED_ROWS && topk <= 64 && (size_t)topk * 12 + 16 <= 60 * 1024) {
    ZRET(ctx->seed_keys.ensure((size_t)count * topk * sizeof(uint64_t)));
    ZRET(ctx->seed_scores.ensure((size_t)count * topk * sizeof(float)));
    ZRET(ctx->seed_counts.ensure((size_t)count * sizeof(uint32_t)));
    ZRET(ctx->part_s.ensure((size_t)count * SEED_ROWS * sizeof(float)));
    ScanArgs d = a;
    d.k = 1; d.n = SEED_ROWS; d.ndense = SEED_ROWS; d.tiles_per_chunk = 1; d.nchunks = SEED_ROWS / TILE_N; d.nqtiles = nqtiles;
    d.dump = ctx->part_s.as<float>(); d.dump_stride = SEED_ROWS;
    ZRET(launch_scan8(d, st.f16, ((d.nchunks + 7) / 8) * 8 * nqtiles, cus, stream));
    MergeArgs m{};
    m.part_s = d.dump; m.slots_per_q = 1; m.slot_stride = 1; m.k = topk; m.slot_len = SEED_ROWS; m.threshold = threshold;
    m.out_keys = ctx->seed_keys.as<uint64_t>(); m.out_scores = ctx->seed_scores.as<float>(); m.out_counts = ctx->seed_counts.as<uint32_t>();
    hipLaunchKernelGGL(merge_kernel, dim3(count), dim3(64), (size_t)topk * 12 + 16, stream, m);
    hipLaunchKernelGGL(seed_gtau_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, ctx->gtau.as<uint32_t>(),
                       m.out_scores, m.out_counts, count, topk);
    ZCHK(hipGetLastError());
  }
  int occ8 = 1;
  a.k = topk;
  ZRET(launch_scan8(a, st.f16, 0, cus, stream, &occ8));
  const uint64_t ntiles = ((uint64_t)kept + TILE_N - 1) / TILE_N;
  const uint64_t resident = (uint64_t)cus * occ8;
  const uint64_t want_chunks = std::max<uint64_t>(1, (resident + nqtiles - 1) / nqtiles);
  uint64_t tpc = std::max<uint64_t>(1, (ntiles + want_chunks - 1) / want_chunks);
  tpc = std::max<uint64_t>(tpc, std::min<uint64_t>(ntiles, 4));
  const uint32_t nchunks = (uint32_t)((ntiles + tpc - 1) / tpc);
  const uint64_t slots = (uint64_t)count * nchunks;
  ZRET(ctx->part_s.ensure(slots * topk * sizeof(float)));
  ZRET(ctx->part_i.ensure(slots * topk * sizeof(uint32_t)));
  a.n = kept; a.ndense = kept; a.tiles_per_chunk = (uint32_t)tpc; a.nchunks = nchunks; a.nqtiles = nqtiles;
  a.part_s = ctx->part_s.as<float>(); a.part_i = ctx->part_i.as<uint32_t>();
  int pi = -1;
  if (profile_it) {
    double bytes = (double)kept * st.dscan * st.elem + (double)count * st.dscan * st.elem + (double)count * topk * 12.0;
    pi = prof_begin(ctx, stream, bytes, 2.0 * (double)count * (double)kept * st.dscan, 0);
  }
  ZRET(launch_scan8(a, st.f16, ((nchunks + 7) / 8) * 8 * nqtiles, cus, stream));
  prof_end(ctx, stream, pi);
  MergeArgs m{};
  m.part_s = a.part_s; m.part_i = a.part_i; m.slots_per_q = nchunks; m.slot_stride = 1; m.k = topk; m.slot_len = topk;
  m.threshold = threshold; m.bound_keys = a.gtau; m.keymap = st.keys;
  m.out_keys = out.keys; m.out_scores = out.scores; m.out_idx = out.idx; m.out_counts = out.counts;
  hipLaunchKernelGGL(merge_kernel, dim3(count), dim3(merge_threads(count)), (size_t)topk * 12 + 16, stream, m);
  ZCHK(hipGetLastError());
  return 0;
}

// `user_facing`: a search whose lists go back to the caller (profiled, L2-refined); false for the IVF
// coarse pass and the k-means labelling, which only need the ranking
int flat_scan_prepared(zvec_hip_ctx_s *ctx, const Store &st, uint32_t count, uint32_t topk, float threshold,
                       const uint64_t *d_exclude, const SearchOut &out_in, hipStream_t stream, bool user_facing) {
  const bool profile_it = user_facing;
  SearchOut out = out_in;
  if (user_facing && st.metric == ZVEC_HIP_METRIC_L2 && out.idx == nullptr) {
    ZRET(ctx->ridx.ensure((size_t)count * topk * sizeof(uint32_t)));
    out.idx = ctx->ridx.as<uint32_t>();
  }
  if (st.n == 0) {
    // no rows: empty results
    MergeArgs m{};
    ZCHK(hipMemsetAsync(out.counts, 0, sizeof(uint32_t) * count, stream));
    ZCHK(hipMemsetAsync(out.keys, 0xff, sizeof(uint64_t) * (size_t)count * topk, stream));
    return 0;
  }
  // Sparse keep-set: compact the kept rows and scan those (work ~ kept rows, like the CPU's skip-before-distance)
  if (d_exclude != nullptr && user_facing && st.n >= 65536) {
    const uint32_t nchunks_b = (uint32_t)((st.n + 2047) / 2048);
    ZRET(ctx->cmp_cnt.ensure(((size_t)2 * nchunks_b + 8) * sizeof(uint32_t)));
    uint32_t *d_cnt = ctx->cmp_cnt.as<uint32_t>(), *d_off = d_cnt + nchunks_b, *d_total = d_off + nchunks_b;
    const uint32_t *ex32 = reinterpret_cast<const uint32_t *>(d_exclude);
    hipLaunchKernelGGL(keep_count_kernel, dim3(nchunks_b), dim3(64), 0, stream, ex32, st.n, d_cnt);
    hipLaunchKernelGGL(u32_exclusive_scan_kernel, dim3(1), dim3(1024), 0, stream, d_cnt, d_off, nchunks_b, d_total);
    ZCHK(hipGetLastError());
    uint32_t kept = 0;
    ZCHK(hipMemcpyAsync(&kept, d_total, 4, hipMemcpyDeviceToHost, stream));
    ZCHK(hipStreamSynchronize(stream));
    // copying the kept rows pays below one half kept; gathering them inside the wide kernel costs ~1.5 % and pays
    // whenever a tenth of the rows can be skipped
    const bool can_gather = !knobs().no_gather && count > 2 * QGROUP && pick_ng(count, topk) == 4 && scan8_lds_bytes(topk) <= LDS_LIMIT - 1024;
    if ((double)kept <= (can_gather ? 0.9 : 0.5) * (double)st.n) {
      if (kept == 0) {
        ZCHK(hipMemsetAsync(out.counts, 0, sizeof(uint32_t) * count, stream));
        ZCHK(hipMemsetAsync(out.keys, 0xff, sizeof(uint64_t) * (size_t)count * topk, stream));
        return 0;
      }
      const uint64_t ktiles = ((uint64_t)kept + TILE_N - 1) / TILE_N;
      if (can_gather) {
        // wide batch: gather the kept rows inside the scan instead of copying them first
        const size_t padded = (size_t)(ktiles + 1) * TILE_N;
        ZRET(ctx->cmp_pos.ensure(padded * 4));
        ZCHK(hipMemsetAsync(ctx->cmp_pos.as<uint32_t>() + kept, 0, (padded - kept) * 4, stream));
        hipLaunchKernelGGL(keep_fill_kernel, dim3(nchunks_b), dim3(64), 0, stream, ex32, st.n, d_off, ctx->cmp_pos.as<uint32_t>());
        ZCHK(hipGetLastError());
        ZRET(flat_scan_gather(ctx, st, ctx->cmp_pos.as<uint32_t>(), kept, count, topk, threshold, out, stream, profile_it));
        if (user_facing) ZRET(refine_l2(ctx, st, count, topk, threshold, out.keys, out.scores, out.idx, out.counts, stream));
        return 0;
      }
      ZRET(ctx->cmp_pos.ensure((size_t)kept * 4));
      ZRET(ctx->cmp_base.ensure((size_t)ktiles * TILE_N * st.dpad * 4));
      ZRET(ctx->cmp_norm.ensure((size_t)ktiles * TILE_N * 4));
      ZRET(ctx->cmp_keys.ensure((size_t)ktiles * TILE_N * 8));
      if (st.extra) ZRET(ctx->cmp_extra.ensure((size_t)ktiles * TILE_N * 4));
      hipLaunchKernelGGL(keep_fill_kernel, dim3(nchunks_b), dim3(64), 0, stream, ex32, st.n, d_off, ctx->cmp_pos.as<uint32_t>());
      hipLaunchKernelGGL(compact_rows_kernel, dim3((kept + 3) / 4), dim3(256), 0, stream, st.base, st.bnorm, st.extra, st.keys,
                         ctx->cmp_pos.as<uint32_t>(), kept, st.dpad, ctx->cmp_base.as<float>(), ctx->cmp_norm.as<float>(),
                         st.extra ? ctx->cmp_extra.as<float>() : nullptr, ctx->cmp_keys.as<uint64_t>());
      ZCHK(hipGetLastError());
      Store tmp = st;                       // a view: same shape parameters, compacted arrays
      tmp.base = ctx->cmp_base.as<float>(); tmp.bnorm = ctx->cmp_norm.as<float>();
      tmp.extra = st.extra ? ctx->cmp_extra.as<float>() : nullptr; tmp.keys = ctx->cmp_keys.as<uint64_t>();
      tmp.n = kept; tmp.cap_tiles = ktiles;
      int rc = flat_scan_prepared(ctx, tmp, count, topk, threshold, nullptr, out_in, stream, user_facing);
      tmp.base = nullptr; tmp.bnorm = nullptr; tmp.extra = nullptr; tmp.keys = nullptr;   // the view owns nothing
      return rc;
    }
  }
  // Dense-score path: the scores of a sub-batch of queries are written once to a [queries][positions] matrix by
  // the same kernel in dump mode and every row is then selected by one wave of merge_kernel.  Used
  //  (a) for small cache-resident bases searched by many queries with a large k (the IVF coarse step: 1024 x 4096
  //      centroids, k = nprobe): the fused admission would spend longer warming up 1024 top-40 lists per tile
  //      run than the matrix cores need for the distances; the 16 MiB of scores stay in L2 / Infinity Cache;
  //  (b) as the large-k path: topk too big for the LDS-resident lists of the fused kernel (k up to ~5000).
  {
    const uint64_t ntiles_d = (st.n + TILE_N - 1) / TILE_N;
    const double row_bytes_d = (double)ntiles_d * TILE_N * 4.0;
    const bool small_base = (double)st.n * st.dpad * 4.0 <= 64.0 * 1024 * 1024;
    const bool k_fits_merge = (size_t)topk * 12 + 16 <= 60 * 1024;
    const bool want_a = small_base && d_exclude == nullptr && topk > 8 && row_bytes_d * count <= 128.0 * 1024 * 1024;
    const bool want_b = pick_ng(count, topk) < 1;
    if (want_b && !k_fits_merge) return ZVEC_HIP_ERR_UNSUPPORTED;
    if ((want_a || want_b) && k_fits_merge) {
      const int cus_d = device_cus(ctx);
      // sub-batches so that the score matrix stays <= 1 GiB
      const uint32_t sub = (uint32_t)std::max<double>(1.0, std::min<double>((double)count, std::floor(1073741824.0 / row_bytes_d)));
      ZRET(ctx->part_s.ensure((size_t)(row_bytes_d * sub)));
      for (uint32_t q0 = 0; q0 < count; q0 += sub) {
        const uint32_t cnt = std::min(sub, count - q0);
        int ngd = pick_ng(cnt, 1);
        // one item per (tile, query tile): halve the query tile while the items would not fill two work-groups per CU
        // (1024 x 4096 coarse scores: 256 items at 128 rows -> 512 at 64 rows, 92 -> 79 us)
        while (ngd > 2 && ntiles_d * ((cnt + ngd * QGROUP - 1) / (ngd * QGROUP)) < 2ull * cus_d) ngd /= 2;
        const uint32_t rows_d = ngd * QGROUP;
        const uint32_t nqt = (cnt + rows_d - 1) / rows_d;
        ScanArgs a{};
        a.base = st.base; a.bnorm = st.bnorm; a.exclude = reinterpret_cast<const uint32_t *>(d_exclude);
        a.queries = ctx->qpad.as<float>() + (size_t)q0 * st.dpad; a.qnorm = ctx->qnorm.as<float>() + q0;
        a.dpad = st.dpad; a.nks = st.dpad / TILE_K; a.metric = st.metric; a.k = 1; a.threshold = threshold;
        a.mode = 0; a.nq = cnt; a.n = st.n; a.ndense = st.n; a.tiles_per_chunk = 1; a.nchunks = (uint32_t)ntiles_d; a.nqtiles = nqt;
        a.gtau = ctx->gtau.as<uint32_t>() + q0;
        a.dump = ctx->part_s.as<float>(); a.dump_stride = (uint32_t)(ntiles_d * TILE_N);
        a.part_s = nullptr; a.part_i = nullptr;
        ZRET(launch_scan_ng(ngd, a, st.f16, (uint32_t)ntiles_d * nqt, cus_d, stream));
        MergeArgs m{};
        m.part_s = a.dump; m.part_i = nullptr; m.part_keys = nullptr; m.slot_begin = nullptr; m.slots_per_q = 1;
        m.slot_stride = 1; m.part_counts = nullptr; m.k = topk; m.slot_len = a.dump_stride; m.threshold = threshold;
        m.keymap = st.keys; m.out_keys = out.keys + (size_t)q0 * topk; m.out_scores = out.scores + (size_t)q0 * topk;
        m.out_idx = out.idx ? out.idx + (size_t)q0 * topk : nullptr; m.out_counts = out.counts + q0;
        hipLaunchKernelGGL(merge_kernel, dim3(cnt), dim3(64), (size_t)topk * 12 + 16, stream, m);
        ZCHK(hipGetLastError());
      }
      if (user_facing) ZRET(refine_l2(ctx, st, count, topk, threshold, out.keys, out.scores, out.idx, out.counts, stream));
      return 0;
    }
  }
  // Bound seeding: every work-group of the fused scan starts its lists empty, and filling a list costs ~k ln(rows/k)
  // sorted insertions per (query, chunk) — with hundreds of chunks in flight that warm-up is most of the admission
  // work.  A scan of a small prefix first (its k-th score bounds the final k-th from above) lets every chunk start
  // with a bound that only ~k * chunk_rows / sample_rows of its rows pass.
  constexpr uint64_t SEED_ROWS = 4096;
  if (!knobs().no_seed && st.n >= 64 * SEED_ROWS && topk <= 64 && count >= 16) {
    ZRET(ctx->seed_keys.ensure((size_t)count * topk * sizeof(uint64_t)));
    ZRET(ctx->seed_scores.ensure((size_t)count * topk * sizeof(float)));
    ZRET(ctx->seed_counts.ensure((size_t)count * sizeof(uint32_t)));
    Store view = st;                      // a view of the first SEED_ROWS rows (whole tiles of the same arrays)
    view.n = SEED_ROWS; view.cap_tiles = SEED_ROWS / TILE_N;
    SearchOut so{ctx->seed_keys.as<uint64_t>(), ctx->seed_scores.as<float>(), nullptr, ctx->seed_counts.as<uint32_t>()};
    int rc = flat_scan_prepared(ctx, view, count, topk, threshold, d_exclude, so, stream, false);
    view.base = nullptr; view.bnorm = nullptr; view.extra = nullptr; view.keys = nullptr;   // the view owns nothing
    ZRET(rc);
    hipLaunchKernelGGL(seed_gtau_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, ctx->gtau.as<uint32_t>(),
                       so.scores, so.counts, count, topk);
    ZCHK(hipGetLastError());
  }
  int ng = pick_ng(count, topk);
  // a base that stays in the 256 MiB Infinity Cache (IVF centroids, k-means codebooks) can be re-read by
  // every query tile for free: prefer many small query tiles (more work-groups, each with a long run of
  // tiles per top-k warm-up) over few large ones
  const bool cache_resident = (double)st.n * st.dpad * 4.0 <= 64.0 * 1024 * 1024;
  if (cache_resident && ng > 1) ng = 1;
  if (ng < 1) return ZVEC_HIP_ERR_UNSUPPORTED;
  const int cus = device_cus(ctx);
  // wide batches over a streamed base: the 8-wave 128x128 tile (two work-groups per CU while its lists fit)
  const bool wide = !knobs().no_wide && (!cache_resident || knobs().force_wide) && pick_ng(count, topk) == 4 && count > 2 * QGROUP && scan8_lds_bytes(topk) <= LDS_LIMIT - 1024;
  int occ8 = 1;
  ScanArgs probe{};
  probe.k = topk; probe.exclude = reinterpret_cast<const uint32_t *>(d_exclude);
  if (wide) ZRET(launch_scan8(probe, st.f16, 0, cus, stream, &occ8));
  const uint32_t rows = wide ? W8_ROWS : ng * QGROUP;
  const uint32_t nqtiles = (count + rows - 1) / rows;
  const uint64_t ntiles = (st.n + TILE_N - 1) / TILE_N;
  uint64_t resident = wide ? (uint64_t)cus * occ8
                           : (uint64_t)cus * (ng >= 4 ? 2 : (ng == 2 ? 2 : 3));   // work-groups per CU each shape reaches
  // items are equal-sized in a flat scan, so ONE wave of work-groups (items == resident slots) is the balanced
  // choice and gives the longest tile runs per top-k warm-up
  uint64_t want_chunks = std::max<uint64_t>(1, (resident + nqtiles - 1) / nqtiles);
  uint64_t tpc = std::max<uint64_t>(1, (ntiles + want_chunks - 1) / want_chunks);
  // >= 4 tiles per top-k warm-up — unless the base is too small to fill the chip that way (a single query over the
  // 4096 IVF centroids: 32 one-tile items instead of 8 four-tile ones, 141 -> 40 us)
  tpc = std::max<uint64_t>(tpc, std::min<uint64_t>(ntiles, ntiles >= 4 * resident ? 4 : 1));
  uint32_t nchunks = (uint32_t)((ntiles + tpc - 1) / tpc);
  uint64_t slots = (uint64_t)count * nchunks;
  ZRET(ctx->part_s.ensure(slots * topk * sizeof(float)));
  ZRET(ctx->part_i.ensure(slots * topk * sizeof(uint32_t)));

  ScanArgs a{};
  a.base = st.base; a.bnorm = st.bnorm; a.exclude = reinterpret_cast<const uint32_t *>(d_exclude);
  a.queries = ctx->qpad.as<float>(); a.qnorm = ctx->qnorm.as<float>();
  a.dpad = st.dpad; a.nks = st.dpad / TILE_K; a.metric = st.metric; a.k = topk; a.threshold = threshold;
  a.gtau = ctx->gtau.as<uint32_t>();
  a.mode = 0; a.nq = count; a.n = st.n; a.ndense = st.n; a.tiles_per_chunk = (uint32_t)tpc; a.nchunks = nchunks; a.nqtiles = nqtiles;
  a.part_s = ctx->part_s.as<float>(); a.part_i = ctx->part_i.as<uint32_t>();
  int pi = -1;
  if (profile_it) {
    double bytes = (double)st.n * st.dscan * st.elem + (double)count * st.dscan * st.elem + (double)count * topk * 12.0;
    double flops = 2.0 * (double)count * (double)st.n * st.dscan;
    pi = prof_begin(ctx, stream, bytes, flops, 0);
  }
  if (wide) ZRET(launch_scan8(a, st.f16, ((nchunks + 7) / 8) * 8 * nqtiles, cus, stream));   // ids padded to whole XCD groups
  else ZRET(launch_scan_ng(ng, a, st.f16, nchunks * nqtiles, cus, stream));
  prof_end(ctx, stream, pi);

  MergeArgs m{};
  m.part_s = a.part_s; m.part_i = a.part_i; m.part_keys = nullptr; m.slot_begin = nullptr;
  m.slots_per_q = nchunks; m.slot_stride = 1; m.part_counts = nullptr; m.k = topk; m.slot_len = topk; m.threshold = threshold;
  m.bound_keys = a.gtau;   // the scan's shared bounds: valid upper bounds of every query's final k-th score
  m.keymap = st.keys; m.out_keys = out.keys; m.out_scores = out.scores; m.out_idx = out.idx; m.out_counts = out.counts;
  hipLaunchKernelGGL(merge_kernel, dim3(count), dim3(merge_threads(count)), (size_t)topk * 12 + 16, stream, m);
  ZCHK(hipGetLastError());
  if (user_facing) ZRET(refine_l2(ctx, st, count, topk, threshold, out.keys, out.scores, out.idx, out.counts, stream));
  return 0;
}

// L2 only: direct re-scoring + re-sort of the final lists (see rescore_l2_kernel)
int refine_l2(zvec_hip_ctx_s *ctx, const Store &st, uint32_t count, uint32_t topk, float threshold, uint64_t *keys,
              float *scores, uint32_t *idx, uint32_t *counts, hipStream_t stream) {
  if (st.metric != ZVEC_HIP_METRIC_L2) return 0;
  if ((size_t)topk * 16 + 16 > 60 * 1024) return 0;   // huge k: keep the expansion scores
  const uint64_t pairs = (uint64_t)count * topk;
  if (st.f16)
    hipLaunchKernelGGL(rescore_l2_kernel<true>, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, stream, st.base,
                       ctx->qpad.as<float>(), st.dpad, idx, counts, count, topk, scores);
  else
    hipLaunchKernelGGL(rescore_l2_kernel<false>, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, stream, st.base,
                       ctx->qpad.as<float>(), st.dpad, idx, counts, count, topk, scores);
  hipLaunchKernelGGL(resort_kernel, dim3(count), dim3(64), (size_t)topk * 16 + 16, stream, keys, scores, idx, counts, topk,
                     threshold);
  ZCHK(hipGetLastError());
  return 0;
}

int prep_queries(zvec_hip_ctx_s *ctx, const Store &st, const void *d_queries, uint32_t count, float threshold,
                 hipStream_t stream) {
  ZRET(ctx->qpad.ensure((size_t)count * st.dpad * sizeof(float)));
  ZRET(ctx->qnorm.ensure((size_t)count * sizeof(float)));
  ZRET(ctx->gtau.ensure((size_t)count * sizeof(uint32_t)));
  if (st.f16)
    hipLaunchKernelGGL(prep_queries_kernel<true>, dim3((count + 3) / 4), dim3(256), 0, stream, d_queries, count,
                       st.dim_in, st.dscan, st.dpad, ctx->qpad.as<float>(), ctx->qnorm.as<float>(),
                       ctx->gtau.as<uint32_t>(), threshold);
  else
    hipLaunchKernelGGL(prep_queries_kernel<false>, dim3((count + 3) / 4), dim3(256), 0, stream, d_queries, count,
                       st.dim_in, st.dscan, st.dpad, ctx->qpad.as<float>(), ctx->qnorm.as<float>(),
                       ctx->gtau.as<uint32_t>(), threshold);
  ZCHK(hipGetLastError());
  return 0;
}

int launch_pack(const Store &st, const void *d_rows, uint64_t n, const uint64_t *d_src, uint64_t pos0,
                const uint64_t *d_dst, hipStream_t stream) {
  if (st.f16)
    hipLaunchKernelGGL(pack_rows_kernel<true>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, d_rows, n, st.dim_in,
                       st.dscan, st.dpad, d_src, pos0, d_dst, st.base, st.bnorm, st.extra);
  else
    hipLaunchKernelGGL(pack_rows_kernel<false>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, d_rows, n, st.dim_in,
                       st.dscan, st.dpad, d_src, pos0, d_dst, st.base, st.bnorm, st.extra);
  ZCHK(hipGetLastError());
  return 0;
}

int launch_unpack(const Store &st, uint64_t pos, void *d_out, hipStream_t stream) {
  if (st.f16)
    hipLaunchKernelGGL(unpack_row_kernel<true>, dim3(1), dim3(256), 0, stream, st.base, st.extra, pos, st.dscan, st.dim_in, st.dpad, d_out);
  else
    hipLaunchKernelGGL(unpack_row_kernel<false>, dim3(1), dim3(256), 0, stream, st.base, st.extra, pos, st.dscan, st.dim_in, st.dpad, d_out);
  ZCHK(hipGetLastError());
  return 0;
}

int store_append_dev(Store &st, const void *d_vecs, uint64_t n, const uint64_t *d_keys, hipStream_t stream) {
  if (n == 0) return 0;
  if (st.n + n >= 0xfffffff0ull) return ZVEC_HIP_ERR_OUT_OF_RANGE;   // positions are 32-bit (IDX_NONE reserved)
  ZRET(st.reserve(st.n + n, stream));
  ZRET(launch_pack(st, d_vecs, n, nullptr, st.n, nullptr, stream));
  ZCHK(hipGetLastError());
  hipLaunchKernelGGL(fill_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, st.keys, st.n, n, d_keys);
  ZCHK(hipGetLastError());
  st.n += n;
  return 0;
}

hipStream_t pick_stream(zvec_hip_ctx_s *ctx, void *stream) {
  return stream ? reinterpret_cast<hipStream_t>(stream) : ctx->cur;
}

int ctx_new(int device, zvec_hip_ctx_s **out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    fprintf(stderr, "[zvec_hip] no HIP device available: the zvec_hip core has no CPU fallback\n");
    return ZVEC_HIP_ERR_RUNTIME;
  }
  if (device < 0 || device >= ndev) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  ZCHK(hipSetDevice(device));
  zvec_hip_ctx_s *c = new (std::nothrow) zvec_hip_ctx_s();
  if (!c) return ZVEC_HIP_ERR_NO_MEMORY;
  c->device = device;
  if (hipStreamCreateWithFlags(&c->own, hipStreamNonBlocking) != hipSuccess) { delete c; return ZVEC_HIP_ERR_RUNTIME; }
  c->cur = c->own;
  *out = c;
  return 0;
}

void ctx_free(zvec_hip_ctx_s *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->own) (void)hipStreamSynchronize(c->own);
  c->gtau.release(); c->ridx.release(); c->cmp_base.release(); c->cmp_norm.release(); c->cmp_extra.release(); c->cmp_keys.release(); c->cmp_pos.release(); c->cmp_cnt.release(); c->qpad.release(); c->qnorm.release(); c->part_s.release(); c->part_i.release();
  c->coarse_keys.release(); c->coarse_scores.release(); c->coarse_idx.release(); c->coarse_cnt.release();
  c->plan.release(); c->io_q.release(); c->io_ex.release(); c->io_keys.release(); c->io_scores.release();
  c->io_counts.release(); c->stats.release();
  for (auto e : c->ev0) (void)hipEventDestroy(e);
  for (auto e : c->ev1) (void)hipEventDestroy(e);
  if (c->own) (void)hipStreamDestroy(c->own);
  delete c;
}

// ---- IVF search core (device pointers) ------------------------------------------------------
int ivf_search_core(zvec_hip_ivf_s *h, zvec_hip_ctx_s *ctx, const void *d_queries, uint32_t count, uint32_t topk,
                    float threshold, uint32_t nprobe, uint32_t max_scan_count, int brute_force,
                    const uint64_t *d_exclude, const SearchOut &out, hipStream_t stream) {
  const int cus = device_cus(ctx);
  const uint32_t nlist = h->nlist;
  if (nprobe < 1) nprobe = 1;
  if (nprobe > nlist) nprobe = nlist;
  ZRET(prep_queries(ctx, h->lists, d_queries, count, FLT_MAX, stream));   // coarse pass: no RNN radius

  // 1. coarse assign: flat scan over the centroids, k = nprobe (IVFCentroidIndex::search)
  if (!brute_force) {
    ZRET(ctx->coarse_keys.ensure((size_t)count * nprobe * sizeof(uint64_t)));
    ZRET(ctx->coarse_scores.ensure((size_t)count * nprobe * sizeof(float)));
    ZRET(ctx->coarse_idx.ensure((size_t)count * nprobe * sizeof(uint32_t)));
    ZRET(ctx->coarse_cnt.ensure((size_t)count * sizeof(uint32_t)));
    SearchOut co{ctx->coarse_keys.as<uint64_t>(), ctx->coarse_scores.as<float>(), ctx->coarse_idx.as<uint32_t>(),
                 ctx->coarse_cnt.as<uint32_t>()};
    ZRET(flat_scan_prepared(ctx, h->cent, count, nprobe, FLT_MAX, nullptr, co, stream, false));
  }

  // 2. plan: list-major work items
  // list scan shape: 16x16x4 MFMA tiles, 32 query rows per work item as two 16-row halves (the second is
  // skipped when the item has <= 16 rows); a list probed by more than 32 queries is dealt as several items
  const int ng = 0;
  if (scan_lds_bytes(1, topk, true) > LDS_LIMIT - 1024) {
    // Large k (beyond ~470): the result lists no longer fit beside the staging buffers.  Rare, so served by the plain
    // route: expand every query's probed lists into positions, score each (query, row) pair directly, select.
    if ((size_t)topk * 12 + 16 > 60 * 1024) return ZVEC_HIP_ERR_UNSUPPORTED;
    PlanArgs p{};
    p.coarse_idx = ctx->coarse_idx.as<uint32_t>(); p.coarse_cnt = ctx->coarse_cnt.as<uint32_t>();
    p.nq = count; p.nprobe = nprobe; p.nlist = nlist; p.max_scan_count = max_scan_count; p.brute_force = brute_force;
    p.list_size = h->d_size; p.list_size_global = h->d_size_global;
    ZRET(ctx->plan.ensure(((size_t)2 * count + 8) * sizeof(uint32_t)));
    uint32_t *d_rows = ctx->plan.as<uint32_t>(), *d_off = d_rows + count;
    // upper bound of the rows one query scans here: the np largest local lists
    uint64_t maxlen = 0;
    {
      std::vector<uint32_t> sz(h->h_size);
      const uint32_t np = brute_force ? nlist : nprobe;
      std::partial_sort(sz.begin(), sz.begin() + np, sz.end(), std::greater<uint32_t>());
      for (uint32_t i = 0; i < np; ++i) maxlen += sz[i];
    }
    if (maxlen == 0) maxlen = 1;
    if ((uint64_t)count * maxlen >= 0xffffffffull) return ZVEC_HIP_ERR_OUT_OF_RANGE;   // (slice the batch)
    hipLaunchKernelGGL(ivf_expand_kernel<false>, dim3((count + 3) / 4), dim3(256), 0, stream, p, h->d_tile0, h->d_dense0,
                       nullptr, d_rows, nullptr, nullptr);
    hipLaunchKernelGGL(u32_exclusive_scan_kernel, dim3(1), dim3(1024), 0, stream, d_rows, d_off, count, d_off + count);
    ZCHK(hipGetLastError());
    uint32_t total_rows = 0;
    ZCHK(hipMemcpyAsync(&total_rows, d_off + count, 4, hipMemcpyDeviceToHost, stream));
    ZCHK(hipStreamSynchronize(stream));
    Scoped<uint32_t> d_pos;
    ZRET(d_pos.alloc(std::max<uint32_t>(total_rows, 1)));
    hipLaunchKernelGGL(ivf_expand_kernel<true>, dim3((count + 3) / 4), dim3(256), 0, stream, p, h->d_tile0, h->d_dense0,
                       reinterpret_cast<const uint32_t *>(d_exclude), nullptr, d_off, d_pos);
    ZCHK(hipGetLastError());
    const uint64_t pairs = (uint64_t)count * maxlen;
    ZRET(ctx->part_s.ensure(pairs * 4));
    ZRET(ctx->part_i.ensure(pairs * 4));
    // (re-prepare the queries with the caller's RNN radius: the coarse pass ran without one)
    ZRET(prep_queries(ctx, h->lists, d_queries, count, threshold, stream));
    if (h->lists.f16)
      hipLaunchKernelGGL(pkeys_score_kernel<true>, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, stream, h->lists.base,
                         ctx->qpad.as<float>(), h->lists.dpad, h->metric, d_pos, d_off, count, (uint32_t)maxlen,
                         ctx->part_s.as<float>(), ctx->part_i.as<uint32_t>());
    else
      hipLaunchKernelGGL(pkeys_score_kernel<false>, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, stream, h->lists.base,
                         ctx->qpad.as<float>(), h->lists.dpad, h->metric, d_pos, d_off, count, (uint32_t)maxlen,
                         ctx->part_s.as<float>(), ctx->part_i.as<uint32_t>());
    ZCHK(hipGetLastError());
    MergeArgs m{};
    m.part_s = ctx->part_s.as<float>(); m.part_i = ctx->part_i.as<uint32_t>();
    m.slots_per_q = 1; m.slot_stride = 1; m.k = topk; m.slot_len = (uint32_t)maxlen; m.threshold = threshold;
    m.keymap = h->lists.keys; m.out_keys = out.keys; m.out_scores = out.scores; m.out_idx = out.idx; m.out_counts = out.counts;
    hipLaunchKernelGGL(merge_kernel, dim3(count), dim3(64), (size_t)topk * 12 + 16, stream, m);
    ZCHK(hipGetLastError());
    ZCHK(hipStreamSynchronize(stream));    // d_pos is freed on return
    ctx->q_nprobe = nullptr; ctx->q_scanned = nullptr; ctx->last_count = 0;
    return 0;
  }
  const uint32_t rows_per_group = 32;
  const uint64_t npairs = (uint64_t)count * (brute_force ? nlist : nprobe);
  // layout of the plan buffer (u32 words)
  size_t off = 0;
  auto take = [&](size_t words) { size_t o = off; off += (words + 3) & ~(size_t)3; return o; };
  size_t o_qnprobe = take(count), o_qscanned = take(count), o_qnslots = take(count), o_slotbegin = take(count + 1);
  size_t o_lcount = take(nlist), o_lfill = take(nlist), o_lqoff = take(nlist + 1), o_itemoff = take(nlist + 1);
  size_t o_queue = take(4);
  size_t o_ltpc = take(nlist);
  size_t o_total = take(4), o_csrq = take(npairs), o_csrslot = take(npairs);
  ZRET(ctx->plan.ensure(off * sizeof(uint32_t)));
  uint32_t *pb = ctx->plan.as<uint32_t>();
  // list_count + list_fill and the work-queue head zeroed, shared bounds reset (the coarse pass may have left
  // centroid-score bounds behind): one launch
  // chunk length of this search: the lists it can touch (at most count x nprobe of them) should give a few items per
  // resident work-group — a single query probing 40 lists needs one-tile items to use the chip at all, a batch of
  // 1024 the index-wide default
  uint32_t tpc = h->tiles_per_chunk;
  {
    const uint64_t lists_touched = std::min<uint64_t>(nlist, (uint64_t)count * (brute_force ? nlist : nprobe));
    const uint64_t est_tiles = std::max<uint64_t>(1, h->local_tiles * lists_touched / std::max<uint32_t>(nlist, 1));
    const uint64_t t = est_tiles / (4ull * (uint64_t)device_cus(ctx) * 3ull);
    tpc = (uint32_t)std::min<uint64_t>(h->tiles_per_chunk, std::max<uint64_t>(1, t));
    if (knobs().ivf_tpc) tpc = (uint32_t)knobs().ivf_tpc;
  }
  {
    ZRET(ctx->gtau.ensure((size_t)count * sizeof(uint32_t)));
    const uint32_t nzero = (uint32_t)(o_lqoff - o_lcount);
    const uint32_t nthr = std::max<uint32_t>(std::max<uint32_t>(nzero, count), std::max<uint32_t>(nlist, 4));
    hipLaunchKernelGGL(ivf_reset_kernel, dim3((nthr + 255) / 256), dim3(256), 0, stream, pb + o_lcount, nzero, pb + o_queue,
                       ctx->gtau.as<uint32_t>(), count, threshold, pb + o_ltpc, h->d_tail, nlist, tpc);
    ZCHK(hipGetLastError());
  }
  PlanArgs p{};
  p.coarse_idx = ctx->coarse_idx.as<uint32_t>(); p.coarse_cnt = ctx->coarse_cnt.as<uint32_t>();
  p.nq = count; p.nprobe = nprobe; p.nlist = nlist; p.max_scan_count = max_scan_count; p.brute_force = brute_force;
  p.list_size = h->d_size; p.list_size_global = h->d_size_global; p.list_order = h->d_order;
  p.list_tpc = pb + o_ltpc;
  p.rows_per_group = rows_per_group;
  p.q_nprobe = pb + o_qnprobe; p.q_scanned = pb + o_qscanned; p.q_nslots = pb + o_qnslots; p.slot_begin = pb + o_slotbegin;
  p.list_count = pb + o_lcount; p.list_fill = pb + o_lfill; p.list_qoff = pb + o_lqoff; p.item_off = pb + o_itemoff;
  p.total_items = pb + o_total; p.csr_q = pb + o_csrq; p.csr_slot = pb + o_csrslot;
  ctx->q_nprobe = p.q_nprobe; ctx->q_scanned = p.q_scanned; ctx->last_count = count;
  ctx->last_list_count = p.list_count;
  hipLaunchKernelGGL(plan_wave_kernel<false>, dim3((count + 3) / 4), dim3(256), 0, stream, p);
  hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(1024), 0, stream, p);
  hipLaunchKernelGGL(plan_wave_kernel<true>, dim3((count + 3) / 4), dim3(256), 0, stream, p);
  ZCHK(hipGetLastError());

  // 3. scan.  Upper bound of slots: every probed list contributes its chunks; the worst case is a query that
  //    probes the nprobe lists with the most chunks.
  std::vector<uint32_t> &sz = h->h_size;
  uint64_t slots_bound;
  {
    std::vector<uint32_t> chunks(nlist);
    for (uint32_t l = 0; l < nlist; ++l)
    {
      const uint32_t t = h->h_tail[l] ? std::max<uint32_t>(1, tpc >> 2) : tpc;
      chunks[l] = sz[l] ? (((sz[l] + TILE_N - 1) / TILE_N + t - 1) / t) : 0;
    }
    uint32_t np = brute_force ? nlist : nprobe;
    std::partial_sort(chunks.begin(), chunks.begin() + np, chunks.end(), std::greater<uint32_t>());
    uint64_t s = 0;
    for (uint32_t i = 0; i < np; ++i) s += chunks[i];
    slots_bound = s * count;
  }
  if (slots_bound == 0) slots_bound = 1;
  ZRET(ctx->part_s.ensure(slots_bound * topk * sizeof(float)));
  ZRET(ctx->part_i.ensure(slots_bound * topk * sizeof(uint32_t)));

  ScanArgs a{};
  a.base = h->lists.base; a.bnorm = h->lists.bnorm; a.exclude = reinterpret_cast<const uint32_t *>(d_exclude);
  a.queries = ctx->qpad.as<float>(); a.qnorm = ctx->qnorm.as<float>();
  a.dpad = h->lists.dpad; a.nks = h->lists.dpad / TILE_K; a.metric = h->metric; a.k = topk; a.threshold = threshold;
  a.gtau = ctx->gtau.as<uint32_t>();
  a.mode = 1; a.nq = count; a.n = h->lists.n; a.ndense = h->count_local; a.tiles_per_chunk = tpc; a.list_tpc = pb + o_ltpc;
  a.total_items = p.total_items; a.queue = pb + o_queue; a.list_order = h->d_order; a.item_off = p.item_off; a.list_tile0 = h->d_tile0; a.list_size = h->d_size;
  a.list_dense0 = h->d_dense0; a.list_qoff = p.list_qoff; a.csr_q = p.csr_q; a.csr_slot = p.csr_slot; a.nlist = nlist;
  a.part_s = ctx->part_s.as<float>(); a.part_i = ctx->part_i.as<uint32_t>();
  // algorithmic bytes of the list scan = rows of the DISTINCT probed lists (counted on device from
  // the plan, see ivf_work_stats_kernel) + the query rows + the result lists
  int pi = prof_begin(ctx, stream, (double)count * h->lists.dscan * h->lists.elem + (double)count * topk * 12.0, 0, 1);
  if (pi >= 0) ctx->prof_dscan[pi] = h->lists.dscan | (h->lists.f16 ? 0x80000000u : 0u);
  ZRET(launch_scan_ng(ng, a, h->lists.f16, 0x7fffffffu, cus, stream));
  prof_end(ctx, stream, pi);

  // 4. merge the per-(query, probe, chunk) partial lists in probe order
  MergeArgs m{};
  m.part_s = a.part_s; m.part_i = a.part_i; m.part_keys = nullptr; m.slot_begin = p.slot_begin; m.slots_per_q = 0;
  m.slot_stride = 1; m.part_counts = nullptr; m.k = topk; m.slot_len = topk; m.threshold = threshold; m.keymap = h->lists.keys;
  m.bound_keys = a.gtau;
  uint32_t *ridx = out.idx;
  if (h->metric == ZVEC_HIP_METRIC_L2 && ridx == nullptr) {
    ZRET(ctx->ridx.ensure((size_t)count * topk * sizeof(uint32_t)));
    ridx = ctx->ridx.as<uint32_t>();
  }
  m.out_keys = out.keys; m.out_scores = out.scores; m.out_idx = ridx; m.out_counts = out.counts;
  hipLaunchKernelGGL(merge_kernel, dim3(count), dim3(merge_threads(count)), (size_t)topk * 12 + 16, stream, m);
  ZCHK(hipGetLastError());
  ZRET(refine_l2(ctx, h->lists, count, topk, threshold, out.keys, out.scores, ridx, out.counts, stream));
  return 0;
}

// work statistics of an IVF launch (for the roofline line): distinct probed rows & pair rows
__global__ void ivf_work_stats_kernel(const uint32_t *list_count, const uint32_t *list_size, uint32_t nlist,
                                      unsigned long long *out2) {
  unsigned long long rows = 0, pairs = 0;
  for (uint32_t l = blockIdx.x * blockDim.x + threadIdx.x; l < nlist; l += gridDim.x * blockDim.x) {
    uint32_t c = list_count[l];
    if (c) { rows += list_size[l]; pairs += (unsigned long long)c * list_size[l]; }
  }
  atomicAdd(&out2[0], rows);
  atomicAdd(&out2[1], pairs);
}

int host_search_wrap_begin(zvec_hip_ctx_s *ctx, const void *queries, size_t qbytes, const uint64_t *exclude,
                           uint64_t nbits, uint32_t count, uint32_t topk, hipStream_t stream) {
  ZRET(ctx->io_q.ensure(qbytes));
  ZCHK(hipMemcpyAsync(ctx->io_q.p, queries, qbytes, hipMemcpyHostToDevice, stream));
  if (exclude) {
    size_t words = (size_t)((nbits + 63) / 64);
    ZRET(ctx->io_ex.ensure(words * 8 + 8));
    ZCHK(hipMemcpyAsync(ctx->io_ex.p, exclude, words * 8, hipMemcpyHostToDevice, stream));
  }
  ZRET(ctx->io_keys.ensure((size_t)count * topk * sizeof(uint64_t)));
  ZRET(ctx->io_scores.ensure((size_t)count * topk * sizeof(float)));
  ZRET(ctx->io_counts.ensure((size_t)count * sizeof(uint32_t)));
  return 0;
}

int host_search_wrap_end(zvec_hip_ctx_s *ctx, uint32_t count, uint32_t topk, uint64_t *out_keys, float *out_scores,
                         uint32_t *out_counts, hipStream_t stream) {
  ZCHK(hipMemcpyAsync(out_keys, ctx->io_keys.p, (size_t)count * topk * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
  ZCHK(hipMemcpyAsync(out_scores, ctx->io_scores.p, (size_t)count * topk * sizeof(float), hipMemcpyDeviceToHost, stream));
  ZCHK(hipMemcpyAsync(out_counts, ctx->io_counts.p, (size_t)count * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  ZCHK(hipStreamSynchronize(stream));
  return 0;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int zvec_hip_abi_version(void) { return ZVEC_HIP_ABI_VERSION; }

int zvec_hip_device_count(int *count) {
  if (!count) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { *count = 0; return ZVEC_HIP_ERR_RUNTIME; }
  *count = n;
  return 0;
}

const char *zvec_hip_error_string(int code) {
  switch (code) {
    case ZVEC_HIP_OK: return "Success";
    case ZVEC_HIP_ERR_RUNTIME: return "Runtime error";
    case ZVEC_HIP_ERR_UNSUPPORTED: return "Unsupported";
    case ZVEC_HIP_ERR_OUT_OF_RANGE: return "Out of range";
    case ZVEC_HIP_ERR_NO_MEMORY: return "Not enough space";
    case ZVEC_HIP_ERR_NO_READY: return "No ready";
    case ZVEC_HIP_ERR_NO_EXIST: return "No exist";
    case ZVEC_HIP_ERR_MISMATCH: return "Mismatch";
    case ZVEC_HIP_ERR_INVALID_ARGUMENT: return "Invalid argument";
    case ZVEC_HIP_ERR_NO_INDEX_LOADED: return "No index loaded";
    case ZVEC_HIP_ERR_NO_TRAINED: return "No trained";
  }
  return "Unknown error";
}

int zvec_hip_ctx_create(int device, zvec_hip_ctx_t *out) {
  if (!out) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  return ctx_new(device, out);
}
int zvec_hip_ctx_destroy(zvec_hip_ctx_t ctx) { ctx_free(ctx); return 0; }
int zvec_hip_ctx_synchronize(zvec_hip_ctx_t ctx) {
  if (!ctx) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  ZCHK(hipSetDevice(ctx->device));
  ZCHK(hipStreamSynchronize(ctx->cur));
  return 0;
}
int zvec_hip_ctx_set_stream(zvec_hip_ctx_t ctx, void *stream) {
  if (!ctx) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  ctx->cur = stream ? reinterpret_cast<hipStream_t>(stream) : ctx->own;
  return 0;
}

// ---- flat -----------------------------------------------------------------------------------
int zvec_hip_flat_create(uint32_t dim, int dtype, int metric, int device, zvec_hip_flat_t *out) {
  if (!out || dim == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (dtype != ZVEC_HIP_DT_FP32 && dtype != ZVEC_HIP_DT_FP16) return ZVEC_HIP_ERR_UNSUPPORTED;
  if (metric < 0 || metric > 2) return ZVEC_HIP_ERR_UNSUPPORTED;
  if (metric == ZVEC_HIP_METRIC_COSINE && dim < (dtype == ZVEC_HIP_DT_FP16 ? 3u : 2u)) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  zvec_hip_ctx_s *c = nullptr;
  ZRET(ctx_new(device, &c));
  zvec_hip_flat_s *h = new (std::nothrow) zvec_hip_flat_s();
  if (!h) { ctx_free(c); return ZVEC_HIP_ERR_NO_MEMORY; }
  h->device = device; h->dtype = dtype; h->defctx = c;
  h->st.configure(dim, metric, dtype);
  *out = h;
  return 0;
}

int zvec_hip_flat_destroy(zvec_hip_flat_t h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  h->st.release();
  ctx_free(h->defctx);
  delete h;
  return 0;
}

int zvec_hip_flat_reserve(zvec_hip_flat_t h, uint64_t capacity) {
  if (!h) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> g(h->mu);
  ZCHK(hipSetDevice(h->device));
  return h->st.reserve(capacity, h->defctx->own);
}

int zvec_hip_flat_append_dev(zvec_hip_flat_t h, const void *d_vecs, uint64_t n, const uint64_t *d_keys, void *stream) {
  if (!h || (!d_vecs && n)) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> g(h->mu);
  ZCHK(hipSetDevice(h->device));
  hipStream_t s = pick_stream(h->defctx, stream);
  return store_append_dev(h->st, d_vecs, n, d_keys, s);
}

// FlatSearcher::load of a dumped "flat.features"-style segment (FlatBuilder<32>::write_row_index / write_column_index,
// src/core/algorithm/flat/flat_builder.cc:186-276): [count][dim] rows, or — column-major index — full 32-row blocks
// transposed in units of the element type followed by a row-major remainder.  Appended to the store on the GPU.
int zvec_hip_flat_load_features(zvec_hip_flat_t h, const void *features, uint64_t bytes, uint64_t count, int column_major,
                                uint32_t batch_size, const uint64_t *keys) {
  if (!h || (!features && count) || batch_size == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (count == 0) return 0;
  const uint64_t elem = h->st.row_bytes();
  if (bytes < count * elem) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> g(h->mu);
  ZCHK(hipSetDevice(h->device));
  hipStream_t s = h->defctx->own;
  Scoped<uint8_t> d_body;
  Scoped<char> d_rows;
  Scoped<uint64_t> d_tab, d_keys;
  ZRET(d_body.alloc(count * elem));
  ZRET(d_rows.alloc(count * elem));
  ZRET(d_tab.alloc(3));
  const uint64_t tab[3] = {0, 0, count};                     // list_off[0]; row0[0], row0[1]
  ZCHK(hipMemcpyAsync(d_body, features, count * elem, hipMemcpyHostToDevice, s));
  ZCHK(hipMemcpyAsync(d_tab, tab, sizeof(tab), hipMemcpyHostToDevice, s));
  if (keys) {
    ZRET(d_keys.alloc(count));
    ZCHK(hipMemcpyAsync(d_keys, keys, count * 8, hipMemcpyHostToDevice, s));
  }
  IvfBodyArgs a{};
  a.body = d_body; a.list_off = d_tab; a.list_row0 = static_cast<uint64_t *>(d_tab) + 1; a.nlist = 1; a.bvc = batch_size;
  a.block_size = (uint32_t)(batch_size * elem); a.elem_size = (uint32_t)elem; a.unit = h->st.elem; a.column_major = column_major ? 1u : 0u;
  a.rows = reinterpret_cast<uint8_t *>(static_cast<char *>(d_rows)); a.total = count;
  hipLaunchKernelGGL(ivf_body_rows_kernel, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, s, a);
  ZCHK(hipGetLastError());
  int rc = store_append_dev(h->st, d_rows, count, keys ? static_cast<const uint64_t *>(d_keys) : nullptr, s);
  ZCHK(hipStreamSynchronize(s));
  return rc;
}

int zvec_hip_flat_append(zvec_hip_flat_t h, const void *vecs, uint64_t n, const uint64_t *keys) {
  if (!h || (!vecs && n)) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (n == 0) return 0;
  std::lock_guard<std::mutex> g(h->mu);
  ZCHK(hipSetDevice(h->device));
  hipStream_t s = h->defctx->own;
  // stage through the device in slices of <= 1 GiB
  const size_t rb = h->st.row_bytes();
  const uint64_t rows_per = std::max<uint64_t>(1, ((uint64_t)1 << 30) / (uint64_t)rb);
  DevBuf tmp, tk;
  for (uint64_t o = 0; o < n; o += rows_per) {
    uint64_t m = std::min(rows_per, n - o);
    int rc = tmp.ensure((size_t)m * rb);
    if (rc == 0 && keys) rc = tk.ensure((size_t)m * 8);
    if (rc != 0) { tmp.release(); tk.release(); return rc; }
    ZCHK(hipMemcpyAsync(tmp.p, reinterpret_cast<const char *>(vecs) + (size_t)o * rb, (size_t)m * rb, hipMemcpyHostToDevice, s));
    if (keys) ZCHK(hipMemcpyAsync(tk.p, keys + o, (size_t)m * 8, hipMemcpyHostToDevice, s));
    rc = store_append_dev(h->st, tmp.p, m, keys ? tk.as<uint64_t>() : nullptr, s);
    if (rc != 0) { tmp.release(); tk.release(); return rc; }
    ZCHK(hipStreamSynchronize(s));
  }
  tmp.release(); tk.release();
  return 0;
}

int zvec_hip_flat_count(zvec_hip_flat_t h, uint64_t *count) {
  if (!h || !count) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  *count = h->st.n;
  return 0;
}

int zvec_hip_flat_get_vector(zvec_hip_flat_t h, uint64_t pos, void *out) {
  if (!h || !out) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> g(h->mu);
  if (pos >= h->st.n) return ZVEC_HIP_ERR_NO_EXIST;
  ZCHK(hipSetDevice(h->device));
  zvec_hip_ctx_s *c = h->defctx;
  ZRET(c->io_q.ensure(h->st.row_bytes()));
  ZRET(launch_unpack(h->st, pos, c->io_q.p, c->own));
  ZCHK(hipMemcpyAsync(out, c->io_q.p, h->st.row_bytes(), hipMemcpyDeviceToHost, c->own));
  ZCHK(hipStreamSynchronize(c->own));
  return 0;
}

int zvec_hip_flat_search_dev(zvec_hip_flat_t h, zvec_hip_ctx_t ctx, const void *d_queries, uint32_t count,
                             uint32_t topk, float threshold, const uint64_t *d_exclude_bitset, uint64_t *d_out_keys,
                             float *d_out_scores, uint32_t *d_out_counts, void *stream) {
  if (!h || !d_queries || !d_out_keys || !d_out_scores || !d_out_counts) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (count == 0) return 0;
  if (topk == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;   // "Invalid context or topk not set yet" flat_searcher.cc:194
  zvec_hip_ctx_s *c = ctx ? ctx : h->defctx;
  // the kernels address the padded query matrix with 32-bit word offsets: very large batches go in slices
  const uint32_t maxq = std::max<uint32_t>(1u, 0x7fffffffu / std::max<uint32_t>(h->st.dpad, 1u));
  if (count > maxq) {
    for (uint32_t q0 = 0; q0 < count; q0 += maxq) {
      const uint32_t m = std::min(maxq, count - q0);
      ZRET(zvec_hip_flat_search_dev(h, ctx, reinterpret_cast<const char *>(d_queries) + (size_t)q0 * h->st.row_bytes(), m, topk,
                                    threshold, d_exclude_bitset, d_out_keys + (size_t)q0 * topk, d_out_scores + (size_t)q0 * topk,
                                    d_out_counts + q0, stream));
    }
    return 0;
  }
  std::lock_guard<std::mutex> g(c->mu);
  ZCHK(hipSetDevice(h->device));
  hipStream_t s = pick_stream(c, stream);
  ZRET(prep_queries(c, h->st, d_queries, count, threshold, s));
  SearchOut out{d_out_keys, d_out_scores, nullptr, d_out_counts};
  return flat_scan_prepared(c, h->st, count, topk, threshold, d_exclude_bitset, out, s, true);
}

int zvec_hip_flat_search(zvec_hip_flat_t h, zvec_hip_ctx_t ctx, const void *queries, uint32_t count, uint32_t topk,
                         float threshold, const uint64_t *exclude_bitset, uint64_t *out_keys, float *out_scores,
                         uint32_t *out_counts) {
  if (!h || !queries || !out_keys || !out_scores || !out_counts) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (count == 0) return 0;
  if (topk == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  zvec_hip_ctx_s *c = ctx ? ctx : h->defctx;
  {
    std::lock_guard<std::mutex> g(c->mu);
    ZCHK(hipSetDevice(h->device));
    ZRET(host_search_wrap_begin(c, queries, (size_t)count * h->st.row_bytes(), exclude_bitset, h->st.n, count, topk, c->cur));
  }
  ZRET(zvec_hip_flat_search_dev(h, c, c->io_q.p, count, topk, threshold, exclude_bitset ? c->io_ex.as<uint64_t>() : nullptr,
                                c->io_keys.as<uint64_t>(), c->io_scores.as<float>(), c->io_counts.as<uint32_t>(), c->cur));
  std::lock_guard<std::mutex> g(c->mu);
  return host_search_wrap_end(c, count, topk, out_keys, out_scores, out_counts, c->cur);
}

int zvec_hip_flat_search_by_ids(zvec_hip_flat_t h, zvec_hip_ctx_t ctx, const void *queries, uint32_t count,
                                const uint32_t *ids, const uint32_t *offsets, uint32_t topk, float threshold,
                                const uint64_t *exclude_bitset, uint64_t *out_keys, float *out_scores,
                                uint32_t *out_counts) {
  if (!h || !queries || !ids || !offsets || !out_keys || !out_scores || !out_counts) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (count == 0) return 0;
  if (topk == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if ((size_t)topk * 12 + 16 > 60 * 1024) return ZVEC_HIP_ERR_UNSUPPORTED;
  zvec_hip_ctx_s *c = ctx ? ctx : h->defctx;
  std::lock_guard<std::mutex> g(c->mu);
  ZCHK(hipSetDevice(h->device));
  hipStream_t s = c->cur;
  const Store &st = h->st;
  // host-side sanitising: positions out of range or excluded by the filter bitset become holes
  const uint32_t total = offsets[count];
  uint32_t maxlen = 1;
  for (uint32_t q = 0; q < count; ++q) {
    if (offsets[q + 1] < offsets[q]) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
    maxlen = std::max(maxlen, offsets[q + 1] - offsets[q]);
  }
  std::vector<uint32_t> clean(std::max<uint32_t>(total, 1));
  for (uint32_t i = 0; i < total; ++i) {
    uint32_t id = ids[i];
    bool ok = id < st.n;
    if (ok && exclude_bitset) ok = ((exclude_bitset[id >> 6] >> (id & 63)) & 1ull) == 0;
    clean[i] = ok ? id : IDX_NONE;
  }
  ZRET(host_search_wrap_begin(c, queries, (size_t)count * st.row_bytes(), nullptr, 0, count, topk, s));
  ZRET(prep_queries(c, st, c->io_q.p, count, threshold, s));
  ZRET(c->plan.ensure(((size_t)total + count + 8) * sizeof(uint32_t)));
  uint32_t *d_pos = c->plan.as<uint32_t>();
  uint32_t *d_off = d_pos + std::max<uint32_t>(total, 1);
  ZCHK(hipMemcpyAsync(d_pos, clean.data(), (size_t)std::max<uint32_t>(total, 1) * 4, hipMemcpyHostToDevice, s));
  ZCHK(hipMemcpyAsync(d_off, offsets, ((size_t)count + 1) * 4, hipMemcpyHostToDevice, s));
  const uint64_t pairs = (uint64_t)count * maxlen;
  ZRET(c->part_s.ensure(pairs * 4));
  ZRET(c->part_i.ensure(pairs * 4));
  if (st.f16)
    hipLaunchKernelGGL(pkeys_score_kernel<true>, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, s, st.base, c->qpad.as<float>(),
                       st.dpad, st.metric, d_pos, d_off, count, maxlen, c->part_s.as<float>(), c->part_i.as<uint32_t>());
  else
    hipLaunchKernelGGL(pkeys_score_kernel<false>, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, s, st.base, c->qpad.as<float>(),
                       st.dpad, st.metric, d_pos, d_off, count, maxlen, c->part_s.as<float>(), c->part_i.as<uint32_t>());
  ZCHK(hipGetLastError());
  MergeArgs m{};
  m.part_s = c->part_s.as<float>(); m.part_i = c->part_i.as<uint32_t>(); m.part_keys = nullptr; m.slot_begin = nullptr;
  m.slots_per_q = 1; m.slot_stride = 1; m.part_counts = nullptr; m.k = topk; m.slot_len = maxlen; m.threshold = threshold;
  m.keymap = st.keys; m.out_keys = c->io_keys.as<uint64_t>(); m.out_scores = c->io_scores.as<float>(); m.out_idx = nullptr;
  m.out_counts = c->io_counts.as<uint32_t>();
  hipLaunchKernelGGL(merge_kernel, dim3(count), dim3(64), (size_t)topk * 12 + 16, s, m);
  ZCHK(hipGetLastError());
  return host_search_wrap_end(c, count, topk, out_keys, out_scores, out_counts, s);
}

// ---- IVF ------------------------------------------------------------------------------------
int zvec_hip_ivf_create(uint32_t dim, int dtype, int metric, int device, zvec_hip_ivf_t *out) {
  if (!out || dim == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (dtype != ZVEC_HIP_DT_FP32 && dtype != ZVEC_HIP_DT_FP16) return ZVEC_HIP_ERR_UNSUPPORTED;
  if (metric < 0 || metric > 2) return ZVEC_HIP_ERR_UNSUPPORTED;
  if (metric == ZVEC_HIP_METRIC_COSINE && dim < (dtype == ZVEC_HIP_DT_FP16 ? 3u : 2u)) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  zvec_hip_ctx_s *c = nullptr;
  ZRET(ctx_new(device, &c));
  zvec_hip_ivf_s *h = new (std::nothrow) zvec_hip_ivf_s();
  if (!h) { ctx_free(c); return ZVEC_HIP_ERR_NO_MEMORY; }
  h->device = device; h->dtype = dtype; h->dim = dim; h->metric = metric; h->defctx = c;
  h->cent.configure(dim, metric, dtype);
  h->lists.configure(dim, metric, dtype);
  *out = h;
  return 0;
}

static void ivf_release(zvec_hip_ivf_s *h) {
  h->cent.release(); h->lists.release();
  h->cent.n = 0; h->lists.n = 0;
  if (h->d_size) (void)hipFree(h->d_size);
  if (h->d_size_global) (void)hipFree(h->d_size_global);
  if (h->d_tile0) (void)hipFree(h->d_tile0);
  if (h->d_order) (void)hipFree(h->d_order);
  if (h->d_tail) (void)hipFree(h->d_tail);
  if (h->d_dense0) (void)hipFree(h->d_dense0);
  h->d_size = h->d_size_global = h->d_tile0 = h->d_order = h->d_tail = nullptr; h->d_dense0 = nullptr;
  h->loaded = false;
}

int zvec_hip_ivf_destroy(zvec_hip_ivf_t h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  ivf_release(h);
  ctx_free(h->defctx);
  delete h;
  return 0;
}

int zvec_hip_ivf_keep_shard(zvec_hip_ivf_t h, uint32_t shard, uint32_t nshards) {
  if (!h || nshards == 0 || shard >= nshards) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (h->loaded) return ZVEC_HIP_ERR_NO_READY;   // must be set before load/build
  h->shard = shard; h->nshards = nshards;
  return 0;
}

// pack rows (device, row-major [n][dim]) given per-row labels (host) into the inverted-list store
static int ivf_pack(zvec_hip_ivf_s *h, const void *d_rows, uint64_t n, const uint64_t *keys,
                    const std::vector<uint32_t> &labels, const void *h_centroids, uint32_t nlist, hipStream_t s) {
  const size_t rb = h->lists.row_bytes();
  h->nlist = nlist;
  h->h_centroids.assign(reinterpret_cast<const char *>(h_centroids), reinterpret_cast<const char *>(h_centroids) + (size_t)nlist * rb);
  h->h_size_global.assign(nlist, 0);
  for (uint64_t i = 0; i < n; ++i) h->h_size_global[labels[i]] += 1;
  h->h_size.assign(nlist, 0);
  for (uint32_t l = 0; l < nlist; ++l)
    if (l % h->nshards == h->shard) h->h_size[l] = h->h_size_global[l];
  h->h_tile0.assign(nlist, 0);
  h->h_dense0.assign(nlist + 1, 0);
  uint64_t tiles = 0, dense = 0;
  for (uint32_t l = 0; l < nlist; ++l) {
    h->h_tile0[l] = (uint32_t)tiles;
    h->h_dense0[l] = dense;
    tiles += (h->h_size[l] + TILE_N - 1) / TILE_N;
    dense += h->h_size[l];
  }
  h->h_dense0[nlist] = dense;
  h->count_local = dense;
  h->count_global = n;
  if (tiles * TILE_N >= 0xffffffffull) return ZVEC_HIP_ERR_OUT_OF_RANGE;
  // stable counting sort of the owned rows into list order
  std::vector<uint64_t> cursor(nlist);
  for (uint32_t l = 0; l < nlist; ++l) cursor[l] = h->h_dense0[l];
  h->h_row_ids.assign(dense, 0);
  for (uint64_t i = 0; i < n; ++i) {
    uint32_t l = labels[i];
    if (l % h->nshards == h->shard) h->h_row_ids[cursor[l]++] = i;
  }
  std::vector<uint64_t> dst(dense), hkeys((size_t)tiles * TILE_N, ~0ull);
  for (uint32_t l = 0; l < nlist; ++l) {
    uint64_t pos0 = (uint64_t)h->h_tile0[l] * TILE_N;
    for (uint64_t j = 0; j < h->h_size[l]; ++j) {
      uint64_t d = h->h_dense0[l] + j;
      dst[d] = pos0 + j;
      hkeys[pos0 + j] = keys ? keys[h->h_row_ids[d]] : h->h_row_ids[d];
    }
  }
  // device side
  h->lists.n = 0;
  ZRET(h->lists.reserve(std::max<uint64_t>(tiles * TILE_N, 1), s));
  h->lists.n = tiles * TILE_N;
  if (dense) {
    Scoped<uint64_t> d_src, d_dst;
    ZRET(d_src.alloc(dense));
    ZRET(d_dst.alloc(dense));
    ZCHK(hipMemcpyAsync(d_src, h->h_row_ids.data(), dense * 8, hipMemcpyHostToDevice, s));
    ZCHK(hipMemcpyAsync(d_dst, dst.data(), dense * 8, hipMemcpyHostToDevice, s));
    ZRET(launch_pack(h->lists, d_rows, dense, d_src, 0, d_dst, s));
    ZCHK(hipMemcpyAsync(h->lists.keys, hkeys.data(), hkeys.size() * 8, hipMemcpyHostToDevice, s));
    ZCHK(hipStreamSynchronize(s));
  }
  // centroids as a flat store
  h->cent.n = 0;
  {
    Scoped<char> d_c;
    ZRET(d_c.alloc((size_t)nlist * rb));
    ZCHK(hipMemcpyAsync(d_c, h_centroids, (size_t)nlist * rb, hipMemcpyHostToDevice, s));
    ZRET(store_append_dev(h->cent, d_c, nlist, nullptr, s));
    ZCHK(hipStreamSynchronize(s));
  }
  // list tables
  if (h->d_size) { (void)hipFree(h->d_size); (void)hipFree(h->d_size_global); (void)hipFree(h->d_tile0); (void)hipFree(h->d_dense0); (void)hipFree(h->d_order); (void)hipFree(h->d_tail); }
  // largest lists are dealt first by the scan's work queue; chunk length adapts to the index size so
  // that a search has a few items per resident work-group yet long runs per top-k warm-up
  std::vector<uint32_t> order(nlist);
  for (uint32_t l = 0; l < nlist; ++l) order[l] = l;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return h->h_size[x] > h->h_size[y]; });
  {
    uint64_t tpc = tiles / (4ull * 256ull * 3ull);
    h->tiles_per_chunk = (uint32_t)std::min<uint64_t>(32, std::max<uint64_t>(4, tpc));
    if (knobs().ivf_tpc) h->tiles_per_chunk = (uint32_t)knobs().ivf_tpc;
    // The queue deals lists largest first, so the lists at the END of the order are the tail of every search: one
    // work-group streams only ~7 GB/s (5.7 TB/s over ~768 resident groups), i.e. a 4-tile item lasts ~200 us, and a
    // tail of such items leaves most of the chip idle.  The last quarter of the tiles is therefore cut into chunks
    // a quarter as long (guided self-scheduling: coarse items first, fine items last).
    h->h_tail.assign(nlist, 0);
    h->local_tiles = tiles;
    uint64_t acc = 0;
    for (uint32_t i = nlist; i-- > 0;) {
      const uint32_t l = order[i];
      if (acc * 4 >= tiles) break;
      h->h_tail[l] = 1;
      acc += (h->h_size[l] + TILE_N - 1) / TILE_N;
    }
  }
  ZCHK(hipMalloc(&h->d_tail, std::max<uint32_t>(nlist, 1) * 4));
  ZCHK(hipMemcpy(h->d_tail, h->h_tail.data(), nlist * 4, hipMemcpyHostToDevice));
  ZCHK(hipMalloc(&h->d_order, nlist * 4));
  ZCHK(hipMemcpy(h->d_order, order.data(), nlist * 4, hipMemcpyHostToDevice));
  ZCHK(hipMalloc(&h->d_size, nlist * 4));
  ZCHK(hipMalloc(&h->d_size_global, nlist * 4));
  ZCHK(hipMalloc(&h->d_tile0, nlist * 4));
  ZCHK(hipMalloc(&h->d_dense0, (nlist + 1) * 8));
  ZCHK(hipMemcpy(h->d_size, h->h_size.data(), nlist * 4, hipMemcpyHostToDevice));
  ZCHK(hipMemcpy(h->d_size_global, h->h_size_global.data(), nlist * 4, hipMemcpyHostToDevice));
  ZCHK(hipMemcpy(h->d_tile0, h->h_tile0.data(), nlist * 4, hipMemcpyHostToDevice));
  ZCHK(hipMemcpy(h->d_dense0, h->h_dense0.data(), (nlist + 1) * 8, hipMemcpyHostToDevice));
  h->loaded = true;
  return 0;
}

int zvec_hip_ivf_load(zvec_hip_ivf_t h, const void *centroids, uint32_t nlist, const uint64_t *list_offsets,
                      const void *vecs, const uint64_t *keys) {
  if (!h || !centroids || nlist == 0 || !list_offsets) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> g(h->mu);
  ZCHK(hipSetDevice(h->device));
  hipStream_t s = h->defctx->own;
  uint64_t n = list_offsets[nlist];
  if (n && !vecs) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::vector<uint32_t> labels(n);
  for (uint32_t l = 0; l < nlist; ++l) {
    if (list_offsets[l + 1] < list_offsets[l]) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
    for (uint64_t i = list_offsets[l]; i < list_offsets[l + 1]; ++i) labels[i] = l;
  }
  Scoped<char> d_rows;
  if (n) {
    ZRET(d_rows.alloc((size_t)n * h->lists.row_bytes()));
    ZCHK(hipMemcpyAsync(d_rows, vecs, (size_t)n * h->lists.row_bytes(), hipMemcpyHostToDevice, s));
  }
  if (h->loaded) ivf_release(h);
  return ivf_pack(h, d_rows, n, keys, labels, centroids, nlist, s);
}

namespace {
// ivf_index_format.h:26-37 / :41-47 and index_meta.cc:23-34, as plain structs of the same layout
struct RefInvertedIndexHeader {
  uint32_t header_size, total_vector_count;
  uint64_t inverted_body_size;
  uint32_t inverted_list_count, block_vector_count, block_size, block_count, index_meta_size;
  char reserved_[28];
};
static_assert(sizeof(RefInvertedIndexHeader) == 64, "InvertedIndexHeader is 64 bytes");
struct RefInvertedListMeta {
  uint64_t offset;
  uint32_t block_count, vector_count, id_offset;
  char reserved_[16];
};
static_assert(sizeof(RefInvertedListMeta) == 40, "InvertedListMeta is 40 bytes");
struct RefIndexMetaHeader {
  uint32_t header_size, meta_type, major_order, data_type, dimension, unit_size, space_id, attachment_offset, attachment_size;
};
}  // namespace

int zvec_hip_ivf_load_segments(zvec_hip_ivf_t h, const void *inverted_header, uint64_t header_bytes,
                               const void *inverted_meta, uint64_t meta_bytes, const void *inverted_body,
                               uint64_t body_bytes, const void *keys, uint64_t keys_bytes, const void *centroids) {
  if (!h || !inverted_header || !inverted_meta || !centroids) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (header_bytes < sizeof(RefInvertedIndexHeader) + sizeof(RefIndexMetaHeader)) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  RefInvertedIndexHeader hd;
  memcpy(&hd, inverted_header, sizeof(hd));
  RefIndexMetaHeader im;
  memcpy(&im, static_cast<const char *>(inverted_header) + sizeof(hd), sizeof(im));
  // IndexMeta::DataType: DT_FP16 = 1, DT_FP32 = 2 (index_meta.h:31-41); MajorOrder: MO_ROW = 1, MO_COLUMN = 2 (:45-49)
  const int dtype = im.data_type == 1 ? ZVEC_HIP_DT_FP16 : (im.data_type == 2 ? ZVEC_HIP_DT_FP32 : -1);
  if (dtype < 0) return ZVEC_HIP_ERR_UNSUPPORTED;
  if (dtype != h->dtype || im.dimension != h->dim) return ZVEC_HIP_ERR_MISMATCH;
  const uint32_t nlist = hd.inverted_list_count, bvc = hd.block_vector_count;
  const uint64_t total = hd.total_vector_count;
  const uint32_t unit = dtype == ZVEC_HIP_DT_FP16 ? 2u : 4u;
  const uint64_t elem = (uint64_t)h->dim * unit;
  if (nlist == 0 || bvc == 0 || meta_bytes < (uint64_t)nlist * sizeof(RefInvertedListMeta)) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (total && (!inverted_body || !keys || keys_bytes < total * 8)) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  const uint64_t block_size = (bvc * elem + 31) / 32 * 32;                   // IVFUtility::AlignedSize
  if (hd.block_size != 0 && hd.block_size != block_size) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  const bool column_major = im.major_order == 2;
  std::vector<uint64_t> list_off(nlist), row0(nlist + 1), list_offsets(nlist + 1);
  uint64_t seen = 0;
  for (uint32_t l = 0; l < nlist; ++l) {
    RefInvertedListMeta m;
    memcpy(&m, static_cast<const char *>(inverted_meta) + (size_t)l * sizeof(m), sizeof(m));
    if (m.id_offset != seen) return ZVEC_HIP_ERR_INVALID_ARGUMENT;          // lists are dumped in id order, back to back
    const uint64_t full = m.vector_count / bvc, rem = m.vector_count % bvc;
    const uint64_t bytes = full * block_size + (rem ? (rem * elem + 31) / 32 * 32 : 0);
    if (m.vector_count && (m.offset > body_bytes || bytes > body_bytes - m.offset)) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
    list_off[l] = m.offset;
    row0[l] = seen;
    list_offsets[l] = seen;
    seen += m.vector_count;
  }
  row0[nlist] = seen;
  list_offsets[nlist] = seen;
  if (seen != total) return ZVEC_HIP_ERR_INVALID_ARGUMENT;

  std::lock_guard<std::mutex> g(h->mu);
  ZCHK(hipSetDevice(h->device));
  hipStream_t s = h->defctx->own;
  Scoped<char> d_rows;
  if (total) {
    Scoped<uint8_t> d_body;
    Scoped<uint64_t> d_off, d_row0;
    ZRET(d_body.alloc(body_bytes));
    ZRET(d_off.alloc(nlist));
    ZRET(d_row0.alloc(nlist + 1));
    ZRET(d_rows.alloc((size_t)total * elem));
    ZCHK(hipMemcpyAsync(d_body, inverted_body, body_bytes, hipMemcpyHostToDevice, s));
    ZCHK(hipMemcpyAsync(d_off, list_off.data(), (size_t)nlist * 8, hipMemcpyHostToDevice, s));
    ZCHK(hipMemcpyAsync(d_row0, row0.data(), ((size_t)nlist + 1) * 8, hipMemcpyHostToDevice, s));
    IvfBodyArgs a{};
    a.body = d_body; a.list_off = d_off; a.list_row0 = d_row0; a.nlist = nlist; a.bvc = bvc; a.block_size = (uint32_t)block_size;
    a.elem_size = (uint32_t)elem; a.unit = unit; a.column_major = column_major ? 1u : 0u;
    a.rows = reinterpret_cast<uint8_t *>(static_cast<char *>(d_rows)); a.total = total;
    hipLaunchKernelGGL(ivf_body_rows_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, a);
    ZCHK(hipGetLastError());
    ZCHK(hipStreamSynchronize(s));      // the uploaded body and tables are freed here
  }
  std::vector<uint32_t> labels(total);
  for (uint32_t l = 0; l < nlist; ++l)
    for (uint64_t i = list_offsets[l]; i < list_offsets[l + 1]; ++i) labels[i] = l;
  if (h->loaded) ivf_release(h);
  return ivf_pack(h, d_rows, total, static_cast<const uint64_t *>(keys), labels, centroids, nlist, s);
}

int zvec_hip_ivf_build_dev(zvec_hip_ivf_t h, const void *d_vecs, uint64_t n, const uint64_t *keys, uint32_t nlist,
                           uint32_t kmeans_iters, uint32_t sample_per_list, uint64_t seed, void *stream) {
  if (!h || !d_vecs || n == 0 || nlist == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (nlist > n) nlist = (uint32_t)n;
  std::lock_guard<std::mutex> g(h->mu);
  ZCHK(hipSetDevice(h->device));
  zvec_hip_ctx_s *c = h->defctx;
  hipStream_t s = pick_stream(c, stream);
  const char *rows = reinterpret_cast<const char *>(d_vecs);
  const uint32_t dim = h->dim;
  const bool f16 = h->lists.f16;
  const size_t rb = h->lists.row_bytes();
  if (sample_per_list == 0) sample_per_list = 256;
  if (h->loaded) ivf_release(h);

  // ---- sample (deterministic stride) ----
  uint64_t S = std::min<uint64_t>(n, (uint64_t)sample_per_list * nlist);
  std::vector<uint64_t> sample_ids(S);
  for (uint64_t i = 0; i < S; ++i) sample_ids[i] = (uint64_t)(((unsigned __int128)i * n) / S);
  Scoped<uint64_t> d_ids;
  Scoped<char> d_sample, d_cent;
  ZRET(d_ids.alloc(S));
  ZRET(d_sample.alloc((size_t)S * rb));
  ZRET(d_cent.alloc((size_t)nlist * rb));
  ZCHK(hipMemcpyAsync(d_ids, sample_ids.data(), S * 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)S), dim3(256), 0, s, (const void *)rows, (uint32_t)rb, d_ids, S, (void *)d_sample);
  ZCHK(hipGetLastError());
  // ---- initial centroids: nlist distinct sample rows picked by a seeded partial shuffle ----
  {
    std::vector<uint64_t> perm(S);
    for (uint64_t i = 0; i < S; ++i) perm[i] = i;
    uint64_t x = seed * 6364136223846793005ull + 1442695040888963407ull;
    for (uint32_t i = 0; i < nlist; ++i) {
      x = x * 6364136223846793005ull + 1442695040888963407ull;
      uint64_t j = i + (x >> 33) % (S - i);
      std::swap(perm[i], perm[j]);
    }
    ZCHK(hipMemcpyAsync(d_ids, perm.data(), (size_t)nlist * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(gather_rows_kernel, dim3(nlist), dim3(256), 0, s, (const void *)d_sample, (uint32_t)rb, d_ids, (uint64_t)nlist, (void *)d_cent);
    ZCHK(hipGetLastError());
    ZCHK(hipStreamSynchronize(s));
  }
  // ---- Lloyd iterations on the sample ----
  Store cs;
  struct StoreGuard { Store &s; ~StoreGuard() { s.release(); } } cs_guard{cs};   // the k-means codebook store
  cs.configure(dim, h->metric, h->dtype);
  Scoped<uint64_t> d_lab_keys; Scoped<float> d_lab_scores; Scoped<uint32_t> d_lab_idx, d_lab_cnt;
  const uint64_t BATCH = 1u << 18;
  uint64_t maxq = std::max<uint64_t>(std::min<uint64_t>(S, BATCH), std::min<uint64_t>(n, BATCH));
  ZRET(d_lab_keys.alloc(maxq));
  ZRET(d_lab_scores.alloc(maxq));
  ZRET(d_lab_idx.alloc(maxq));
  ZRET(d_lab_cnt.alloc(maxq));
  Scoped<uint64_t> d_moff, d_members;
  ZRET(d_moff.alloc((size_t)nlist + 1));
  ZRET(d_members.alloc(S));
  std::vector<uint32_t> lab(std::max<uint64_t>(S, n));
  auto assign = [&](const char *q, uint64_t nq, uint32_t *host_labels) -> int {
    for (uint64_t o = 0; o < nq; o += BATCH) {
      uint32_t m = (uint32_t)std::min<uint64_t>(BATCH, nq - o);
      ZRET(prep_queries(c, cs, q + (size_t)o * rb, m, FLT_MAX, s));
      SearchOut out{d_lab_keys.p, d_lab_scores.p, d_lab_idx.p, d_lab_cnt.p};
      ZRET(flat_scan_prepared(c, cs, m, 1, FLT_MAX, nullptr, out, s, false));
      ZCHK(hipMemcpyAsync(host_labels + o, d_lab_idx, (size_t)m * 4, hipMemcpyDeviceToHost, s));
      ZCHK(hipStreamSynchronize(s));
    }
    return 0;
  };
  for (uint32_t it = 0; it < kmeans_iters; ++it) {
    cs.n = 0;
    ZRET(store_append_dev(cs, d_cent, nlist, nullptr, s));
    ZRET(assign(d_sample, S, lab.data()));
    std::vector<uint64_t> moff(nlist + 1, 0), members(S);
    for (uint64_t i = 0; i < S; ++i) moff[(lab[i] < nlist ? lab[i] : 0) + 1] += 1;
    for (uint32_t l = 0; l < nlist; ++l) moff[l + 1] += moff[l];
    std::vector<uint64_t> cur(moff.begin(), moff.end() - 1);
    for (uint64_t i = 0; i < S; ++i) members[cur[lab[i] < nlist ? lab[i] : 0]++] = i;
    ZCHK(hipMemcpyAsync(d_moff, moff.data(), moff.size() * 8, hipMemcpyHostToDevice, s));
    ZCHK(hipMemcpyAsync(d_members, members.data(), S * 8, hipMemcpyHostToDevice, s));
    if (f16) hipLaunchKernelGGL(centroid_mean_kernel<true>, dim3(nlist), dim3(256), 0, s, (const void *)d_sample, dim, d_moff, d_members, (void *)d_cent);
    else hipLaunchKernelGGL(centroid_mean_kernel<false>, dim3(nlist), dim3(256), 0, s, (const void *)d_sample, dim, d_moff, d_members, (void *)d_cent);
    ZCHK(hipGetLastError());
    ZCHK(hipStreamSynchronize(s));
    // empty clusters: split the currently largest one (tiny symmetric perturbation), as k-means trainers do
    std::vector<uint32_t> empties;
    std::vector<uint64_t> sizes(nlist);
    for (uint32_t l = 0; l < nlist; ++l) { sizes[l] = moff[l + 1] - moff[l]; if (sizes[l] == 0) empties.push_back(l); }
    if (!empties.empty() && it + 1 < kmeans_iters) {
      std::vector<char> hcb((size_t)nlist * rb);
      ZCHK(hipMemcpy(hcb.data(), d_cent, hcb.size(), hipMemcpyDeviceToHost));
      for (uint32_t e : empties) {
        uint32_t b = (uint32_t)(std::max_element(sizes.begin(), sizes.end()) - sizes.begin());
        if (sizes[b] < 2) break;
        for (uint32_t c = 0; c < dim; ++c) {
          if (f16) {
            _Float16 *hp = reinterpret_cast<_Float16 *>(hcb.data());
            float v = (float)hp[(size_t)b * dim + c];
            hp[(size_t)e * dim + c] = (_Float16)(v * (1.0f + 1.0f / 256.0f));
            hp[(size_t)b * dim + c] = (_Float16)(v * (1.0f - 1.0f / 256.0f));
          } else {
            float *hp = reinterpret_cast<float *>(hcb.data());
            float v = hp[(size_t)b * dim + c];
            hp[(size_t)e * dim + c] = v * (1.0f + 1.0f / 1024.0f);
            hp[(size_t)b * dim + c] = v * (1.0f - 1.0f / 1024.0f);
          }
        }
        sizes[e] = sizes[b] / 2;
        sizes[b] -= sizes[e];
      }
      ZCHK(hipMemcpy(d_cent, hcb.data(), hcb.size(), hipMemcpyHostToDevice));
    }
  }
  // ---- label every row with its nearest centroid (ivf_builder.h:253-274) ----
  cs.n = 0;
  ZRET(store_append_dev(cs, d_cent, nlist, nullptr, s));
  ZRET(assign(rows, n, lab.data()));
  std::vector<char> hc((size_t)nlist * rb);
  ZCHK(hipMemcpy(hc.data(), d_cent, hc.size(), hipMemcpyDeviceToHost));
  lab.resize(n);
  for (uint64_t i = 0; i < n; ++i) if (lab[i] >= nlist) lab[i] = 0;
  return ivf_pack(h, rows, n, keys, lab, hc.data(), nlist, s);
}

int zvec_hip_ivf_build(zvec_hip_ivf_t h, const void *vecs, uint64_t n, const uint64_t *keys, uint32_t nlist,
                       uint32_t kmeans_iters, uint32_t sample_per_list, uint64_t seed) {
  if (!h || !vecs || n == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  ZCHK(hipSetDevice(h->device));
  Scoped<char> d_rows;
  ZRET(d_rows.alloc((size_t)n * h->lists.row_bytes()));
  ZCHK(hipMemcpy(d_rows, vecs, (size_t)n * h->lists.row_bytes(), hipMemcpyHostToDevice));
  return zvec_hip_ivf_build_dev(h, d_rows, n, keys, nlist, kmeans_iters, sample_per_list, seed, nullptr);
}

int zvec_hip_ivf_info(zvec_hip_ivf_t h, uint64_t *count, uint32_t *nlist) {
  if (!h) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (count) *count = h->count_local;
  if (nlist) *nlist = h->nlist;
  return 0;
}

int zvec_hip_ivf_export(zvec_hip_ivf_t h, void *centroids, uint64_t *list_offsets, uint64_t *row_ids) {
  if (!h) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (!h->loaded) return ZVEC_HIP_ERR_NO_INDEX_LOADED;
  if (centroids) memcpy(centroids, h->h_centroids.data(), h->h_centroids.size());
  if (list_offsets) memcpy(list_offsets, h->h_dense0.data(), h->h_dense0.size() * 8);
  if (row_ids) memcpy(row_ids, h->h_row_ids.data(), h->h_row_ids.size() * 8);
  return 0;
}

int zvec_hip_ivf_get_vector(zvec_hip_ivf_t h, uint64_t list_pos, void *out) {
  if (!h || !out) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (!h->loaded) return ZVEC_HIP_ERR_NO_INDEX_LOADED;
  if (list_pos >= h->count_local) return ZVEC_HIP_ERR_NO_EXIST;
  std::lock_guard<std::mutex> g(h->mu);
  ZCHK(hipSetDevice(h->device));
  uint32_t l = (uint32_t)(std::upper_bound(h->h_dense0.begin(), h->h_dense0.end(), list_pos) - h->h_dense0.begin()) - 1;
  uint64_t pos = (uint64_t)h->h_tile0[l] * TILE_N + (list_pos - h->h_dense0[l]);
  zvec_hip_ctx_s *c = h->defctx;
  ZRET(c->io_q.ensure(h->lists.row_bytes()));
  ZRET(launch_unpack(h->lists, pos, c->io_q.p, c->own));
  ZCHK(hipMemcpyAsync(out, c->io_q.p, h->lists.row_bytes(), hipMemcpyDeviceToHost, c->own));
  ZCHK(hipStreamSynchronize(c->own));
  return 0;
}

static int ivf_search_dev_impl(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, const void *d_queries, uint32_t count,
                               uint32_t topk, float threshold, uint32_t nprobe, uint32_t max_scan_count,
                               int brute_force, const uint64_t *d_exclude, uint64_t *d_out_keys, float *d_out_scores,
                               uint32_t *d_out_counts, void *stream) {
  if (!h || !d_queries || !d_out_keys || !d_out_scores || !d_out_counts) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (!h->loaded) return ZVEC_HIP_ERR_NO_INDEX_LOADED;
  if (count == 0) return 0;
  if (topk == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;   // ivf_searcher.cc:197-200
  zvec_hip_ctx_s *c = ctx ? ctx : h->defctx;
  {
    const uint32_t maxq = std::max<uint32_t>(1u, 0x7fffffffu / std::max<uint32_t>(h->lists.dpad, 1u));
    if (count > maxq) {   // 32-bit word offsets into the padded query matrix: slice very large batches
      for (uint32_t q0 = 0; q0 < count; q0 += maxq) {
        const uint32_t m = std::min(maxq, count - q0);
        ZRET(ivf_search_dev_impl(h, ctx, reinterpret_cast<const char *>(d_queries) + (size_t)q0 * h->lists.row_bytes(), m, topk,
                                 threshold, nprobe, max_scan_count, brute_force, d_exclude, d_out_keys + (size_t)q0 * topk,
                                 d_out_scores + (size_t)q0 * topk, d_out_counts + q0, stream));
      }
      return 0;
    }
  }
  std::lock_guard<std::mutex> g(c->mu);
  ZCHK(hipSetDevice(h->device));
  hipStream_t s = pick_stream(c, stream);
  SearchOut out{d_out_keys, d_out_scores, nullptr, d_out_counts};
  int rc = ivf_search_core(h, c, d_queries, count, topk, threshold, nprobe,
                           max_scan_count, brute_force, d_exclude, out, s);
  if (rc == 0 && c->profile && c->nprof > 0 && c->nprof <= PROFILE_MAX && c->stats.p) {
    int i = c->nprof - 1;
    if (c->launch_is_ivf[i]) {
      unsigned long long *st = c->stats.as<unsigned long long>() + 2 * (size_t)i;
      ZCHK(hipMemsetAsync(st, 0, 16, s));
      hipLaunchKernelGGL(ivf_work_stats_kernel, dim3(16), dim3(256), 0, s, c->last_list_count, h->d_size, h->nlist, st);
      ZCHK(hipGetLastError());
    }
  }
  return rc;
}

int zvec_hip_ivf_search_dev(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, const void *d_queries, uint32_t count, uint32_t topk,
                            float threshold, uint32_t nprobe, uint32_t max_scan_count, const uint64_t *d_exclude_bitset,
                            uint64_t *d_out_keys, float *d_out_scores, uint32_t *d_out_counts, void *stream) {
  return ivf_search_dev_impl(h, ctx, d_queries, count, topk, threshold, nprobe, max_scan_count, 0, d_exclude_bitset,
                             d_out_keys, d_out_scores, d_out_counts, stream);
}

static int ivf_search_host_impl(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, const void *queries, uint32_t count, uint32_t topk,
                                float threshold, uint32_t nprobe, uint32_t max_scan_count, int brute_force,
                                const uint64_t *exclude_bitset, uint64_t *out_keys, float *out_scores,
                                uint32_t *out_counts) {
  if (!h || !queries || !out_keys || !out_scores || !out_counts) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (!h->loaded) return ZVEC_HIP_ERR_NO_INDEX_LOADED;
  if (count == 0) return 0;
  if (topk == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  zvec_hip_ctx_s *c = ctx ? ctx : h->defctx;
  {
    std::lock_guard<std::mutex> g(c->mu);
    ZCHK(hipSetDevice(h->device));
    ZRET(host_search_wrap_begin(c, queries, (size_t)count * h->lists.row_bytes(), exclude_bitset, h->count_local, count, topk, c->cur));
  }
  ZRET(ivf_search_dev_impl(h, c, c->io_q.p, count, topk, threshold, nprobe, max_scan_count, brute_force,
                           exclude_bitset ? c->io_ex.as<uint64_t>() : nullptr, c->io_keys.as<uint64_t>(),
                           c->io_scores.as<float>(), c->io_counts.as<uint32_t>(), c->cur));
  std::lock_guard<std::mutex> g(c->mu);
  return host_search_wrap_end(c, count, topk, out_keys, out_scores, out_counts, c->cur);
}

int zvec_hip_ivf_search(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, const void *queries, uint32_t count, uint32_t topk,
                        float threshold, uint32_t nprobe, uint32_t max_scan_count, const uint64_t *exclude_bitset,
                        uint64_t *out_keys, float *out_scores, uint32_t *out_counts) {
  return ivf_search_host_impl(h, ctx, queries, count, topk, threshold, nprobe, max_scan_count, 0, exclude_bitset,
                              out_keys, out_scores, out_counts);
}

int zvec_hip_ivf_search_bf(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, const void *queries, uint32_t count, uint32_t topk,
                           float threshold, const uint64_t *exclude_bitset, uint64_t *out_keys, float *out_scores,
                           uint32_t *out_counts) {
  return ivf_search_host_impl(h, ctx, queries, count, topk, threshold, 1, 0xffffffffu, 1, exclude_bitset, out_keys,
                              out_scores, out_counts);
}

int zvec_hip_ivf_last_stats(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, uint32_t count, uint32_t *scanned, uint32_t *probes) {
  if (!h) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  zvec_hip_ctx_s *c = ctx ? ctx : h->defctx;
  std::lock_guard<std::mutex> g(c->mu);
  if (!c->q_scanned || count > c->last_count) return ZVEC_HIP_ERR_NO_READY;
  ZCHK(hipSetDevice(h->device));
  ZCHK(hipStreamSynchronize(c->cur));
  if (scanned) ZCHK(hipMemcpy(scanned, c->q_scanned, (size_t)count * 4, hipMemcpyDeviceToHost));
  if (probes) ZCHK(hipMemcpy(probes, c->q_nprobe, (size_t)count * 4, hipMemcpyDeviceToHost));
  return 0;
}

// ---- merge ----------------------------------------------------------------------------------
int zvec_hip_merge_topk_dev(zvec_hip_ctx_t ctx, const uint64_t *d_keys, const float *d_scores, const uint32_t *d_counts,
                            uint32_t nparts, uint32_t count, uint32_t topk, uint64_t *d_out_keys, float *d_out_scores,
                            uint32_t *d_out_counts, void *stream) {
  if (!ctx || !d_keys || !d_scores || !d_counts || !d_out_keys || !d_out_scores || !d_out_counts)
    return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (count == 0) return 0;
  if (topk == 0 || nparts == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if ((size_t)topk * 12 + 16 > 64 * 1024) return ZVEC_HIP_ERR_UNSUPPORTED;
  std::lock_guard<std::mutex> g(ctx->mu);
  ZCHK(hipSetDevice(ctx->device));
  hipStream_t s = pick_stream(ctx, stream);
  MergeArgs m{};
  m.part_s = d_scores; m.part_i = nullptr; m.part_keys = d_keys; m.slot_begin = nullptr; m.slots_per_q = nparts;
  m.slot_stride = count; m.part_counts = d_counts; m.k = topk; m.slot_len = topk; m.threshold = FLT_MAX; m.keymap = nullptr;
  m.out_keys = d_out_keys; m.out_scores = d_out_scores; m.out_idx = nullptr; m.out_counts = d_out_counts;
  hipLaunchKernelGGL(merge_kernel, dim3(count), dim3(64), (size_t)topk * 12 + 16, s, m);
  ZCHK(hipGetLastError());
  return 0;
}

uint64_t zvec_hip_packed_bytes(uint32_t count, uint32_t topk) {
  uint64_t b = (uint64_t)count * topk * 12 + (uint64_t)count * 4;
  return (b + 15) & ~(uint64_t)15;
}

int zvec_hip_merge_topk_packed_dev(zvec_hip_ctx_t ctx, const void *d_packed, uint64_t part_stride, uint32_t nparts,
                                   uint32_t count, uint32_t topk, uint64_t *d_out_keys, float *d_out_scores,
                                   uint32_t *d_out_counts, void *stream) {
  if (!ctx || !d_packed || !d_out_keys || !d_out_scores || !d_out_counts) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (count == 0) return 0;
  if (topk == 0 || nparts == 0 || part_stride < zvec_hip_packed_bytes(count, topk) || (part_stride & 7)) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if ((size_t)topk * 12 + 16 > 64 * 1024) return ZVEC_HIP_ERR_UNSUPPORTED;
  std::lock_guard<std::mutex> g(ctx->mu);
  ZCHK(hipSetDevice(ctx->device));
  hipStream_t s = pick_stream(ctx, stream);
  const char *p0 = reinterpret_cast<const char *>(d_packed);
  MergeArgs m{};
  m.part_keys = reinterpret_cast<const uint64_t *>(p0);
  m.part_s = reinterpret_cast<const float *>(p0 + (size_t)count * topk * 8);
  m.part_counts = reinterpret_cast<const uint32_t *>(p0 + (size_t)count * topk * 12);
  m.part_i = nullptr; m.slot_begin = nullptr; m.slots_per_q = nparts; m.slot_stride = count; m.packed_stride = part_stride;
  m.k = topk; m.slot_len = topk; m.threshold = FLT_MAX; m.keymap = nullptr;
  m.out_keys = d_out_keys; m.out_scores = d_out_scores; m.out_idx = nullptr; m.out_counts = d_out_counts;
  hipLaunchKernelGGL(merge_kernel, dim3(count), dim3(64), (size_t)topk * 12 + 16, s, m);
  ZCHK(hipGetLastError());
  return 0;
}

int zvec_hip_merge_topk(zvec_hip_ctx_t ctx, const uint64_t *keys, const float *scores, const uint32_t *counts,
                        uint32_t nparts, uint32_t count, uint32_t topk, uint64_t *out_keys, float *out_scores,
                        uint32_t *out_counts) {
  if (!ctx || !keys || !scores || !counts || !out_keys || !out_scores || !out_counts) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (count == 0) return 0;
  ZCHK(hipSetDevice(ctx->device));
  size_t ne = (size_t)nparts * count * topk;
  Scoped<uint64_t> dk, dok; Scoped<float> ds, dos; Scoped<uint32_t> dc, doc;
  ZRET(dk.alloc(ne)); ZRET(ds.alloc(ne)); ZRET(dc.alloc((size_t)nparts * count));
  ZRET(dok.alloc((size_t)count * topk)); ZRET(dos.alloc((size_t)count * topk)); ZRET(doc.alloc(count));
  ZCHK(hipMemcpy(dk, keys, ne * 8, hipMemcpyHostToDevice));
  ZCHK(hipMemcpy(ds, scores, ne * 4, hipMemcpyHostToDevice));
  ZCHK(hipMemcpy(dc, counts, (size_t)nparts * count * 4, hipMemcpyHostToDevice));
  int rc = zvec_hip_merge_topk_dev(ctx, dk, ds, dc, nparts, count, topk, dok, dos, doc, nullptr);
  if (rc == 0) {
    ZCHK(hipStreamSynchronize(ctx->cur));
    ZCHK(hipMemcpy(out_keys, dok, (size_t)count * topk * 8, hipMemcpyDeviceToHost));
    ZCHK(hipMemcpy(out_scores, dos, (size_t)count * topk * 4, hipMemcpyDeviceToHost));
    ZCHK(hipMemcpy(out_counts, doc, (size_t)count * 4, hipMemcpyDeviceToHost));
  }
  return rc;
}

// ---- measurement hook -----------------------------------------------------------------------
int zvec_hip_ctx_profile(zvec_hip_ctx_t ctx, int enable) {
  if (!ctx) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> g(ctx->mu);
  ctx->profile = enable != 0;
  if (ctx->profile) {
    ZCHK(hipSetDevice(ctx->device));
    ZRET(ctx->stats.ensure(sizeof(uint64_t) * 2 * PROFILE_MAX));
  }
  return 0;
}

int zvec_hip_ctx_profile_read(zvec_hip_ctx_t ctx, uint64_t *launches, double *scan_ms, double *algorithmic_bytes,
                              double *algorithmic_flops, int reset) {
  if (!ctx) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> g(ctx->mu);
  ZCHK(hipSetDevice(ctx->device));
  ZCHK(hipStreamSynchronize(ctx->cur));
  double ms = 0, bytes = 0, flops = 0;
  std::vector<unsigned long long> st;
  if (ctx->nprof > 0 && ctx->stats.p) {
    st.resize((size_t)2 * ctx->nprof);
    ZCHK(hipMemcpy(st.data(), ctx->stats.p, st.size() * 8, hipMemcpyDeviceToHost));
  }
  for (int i = 0; i < ctx->nprof; ++i) {
    float t = 0;
    if (hipEventElapsedTime(&t, ctx->ev0[i], ctx->ev1[i]) == hipSuccess) ms += t;
    bytes += ctx->host_bytes[i];
    flops += ctx->host_flops[i];
  }
  for (int i = 0; i < ctx->nprof && !st.empty(); ++i) {
    if (!ctx->launch_is_ivf[i]) continue;
    const double ds = (double)(ctx->prof_dscan[i] & 0x7fffffffu), eb = (ctx->prof_dscan[i] & 0x80000000u) ? 2.0 : 4.0;
    bytes += (double)st[2 * (size_t)i] * ds * eb;            // distinct probed rows
    flops += (double)st[2 * (size_t)i + 1] * ds * 2.0;       // (query, row) pairs
  }
  if (launches) *launches = (uint64_t)ctx->nprof;
  if (scan_ms) *scan_ms = ms;
  if (algorithmic_bytes) *algorithmic_bytes = bytes;
  if (algorithmic_flops) *algorithmic_flops = flops;
  if (reset) ctx->nprof = 0;
  return 0;
}

// ---- predicate materialisation ------------------------------------------------------------------------------
namespace {

uint32_t crc32c_update(const void *data, uint64_t len, uint32_t crc) {
  static uint32_t table[256];
  static std::once_flag once;
  std::call_once(once, [] {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
      table[i] = c;
    }
  });
  const uint8_t *p = static_cast<const uint8_t *>(data);
  for (uint64_t i = 0; i < len; ++i) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
  return crc;
}

// container directory of one (or, for a 64-bit map, several) portable 32-bit roaring streams
struct RoaringDir {
  std::vector<uint64_t> ckey, coff;
  std::vector<uint32_t> cinfo;
};

inline uint32_t rd_u16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t rd_u32(const uint8_t *p) { return rd_u16(p) | (rd_u16(p + 2) << 16); }
inline uint64_t rd_u64(const uint8_t *p) { return (uint64_t)rd_u32(p) | ((uint64_t)rd_u32(p + 4) << 32); }

// RoaringFormatSpec "portable" layout (what roaring_bitmap_portable_serialize of CRoaring 2.0.4 writes):
//   cookie  : u32 12346 + u32 container count               (no run containers)
//           | u16 12347, u16 count-1, ceil(count/8) bytes of run flags
//   header  : count x (u16 key, u16 cardinality-1)
//   offsets : count x u32, present unless (run cookie && count < 4)
//   payload : per container — run: u16 n_runs + n_runs x (u16 start, u16 length-1);
//             cardinality > 4096: 1024 x u64 bitset; else cardinality x u16 sorted values
// Returns the bytes consumed, or 0 for a malformed stream.  `base` = offset of b[0] in the uploaded buffer.
uint64_t parse_roaring32(const uint8_t *b, uint64_t len, uint64_t high, uint64_t base, RoaringDir &dir) {
  if (len < 4) return 0;
  const uint32_t cookie = rd_u32(b);
  uint64_t pos;
  uint32_t n;
  const uint8_t *runflags = nullptr;
  if ((cookie & 0xffffu) == 12347u) {
    n = (cookie >> 16) + 1;
    runflags = b + 4;
    pos = 4 + (n + 7) / 8;
  } else if (cookie == 12346u) {
    if (len < 8) return 0;
    n = rd_u32(b + 4);
    pos = 8;
  } else {
    return 0;
  }
  if (n > 65536u || pos + (uint64_t)4 * n > len) return 0;
  const uint8_t *desc = b + pos;
  pos += (uint64_t)4 * n;
  if (runflags == nullptr || n >= 4) {
    if (pos + (uint64_t)4 * n > len) return 0;
    pos += (uint64_t)4 * n;
  }
  uint32_t prev_key = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t key = rd_u16(desc + 4 * i), card = rd_u16(desc + 4 * i + 2) + 1;
    if (i > 0 && key <= prev_key) return 0;          // keys strictly ascending
    prev_key = key;
    const bool is_run = runflags && ((runflags[i >> 3] >> (i & 7)) & 1u);
    uint32_t type, cnt;
    uint64_t size, payload = pos;
    if (is_run) {
      if (pos + 2 > len) return 0;
      cnt = rd_u16(b + pos);
      type = 2; payload = pos + 2; size = 2 + (uint64_t)4 * cnt;
    } else if (card > 4096u) {
      type = 1; cnt = card; size = 8192;
    } else {
      type = 0; cnt = card; size = (uint64_t)2 * card;
    }
    if (pos + size > len) return 0;
    dir.ckey.push_back((high << 16) | key);
    dir.cinfo.push_back(type | (cnt << 2));
    dir.coff.push_back(base + payload);
    pos += size;
  }
  return pos;
}

// roaring::Roaring64Map::write(portable): u64 map size, then per entry u32 high key + a portable 32-bit stream
bool parse_roaring64map(const uint8_t *b, uint64_t len, uint64_t base, RoaringDir &dir) {
  if (len < 8) return false;
  const uint64_t m = rd_u64(b);
  uint64_t pos = 8;
  uint64_t prev = 0;
  for (uint64_t i = 0; i < m; ++i) {
    if (pos + 4 > len) return false;
    const uint64_t high = rd_u32(b + pos);
    if (i > 0 && high <= prev) return false;
    prev = high;
    pos += 4;
    const uint64_t used = parse_roaring32(b + pos, len - pos, high, base + pos, dir);
    if (used == 0) return false;
    pos += used;
  }
  return true;
}

struct BitmapFileHeader {     // concurrent_roaring_bitmap.h:186-192
  uint64_t magic;
  uint32_t is_32bit;
  uint32_t checksum;
  uint64_t timestamp;
  uint32_t reserved_[10];
};
static_assert(sizeof(BitmapFileHeader) == 64, "BitmapMetaHeader is 64 bytes");
constexpr uint64_t ROARING_FILE_MAGIC = 0x362DDA444AC1B99Aull;

struct DeviceRoaring {
  Scoped<uint8_t> bytes;
  Scoped<uint64_t> ckey, coff;
  Scoped<uint32_t> cinfo;
  RoaringView view{};
};

int upload_roaring(const void *data, uint64_t len, int kind, DeviceRoaring &out, hipStream_t s) {
  out.view = RoaringView{};
  if (kind == ZVEC_HIP_ROARING_NONE || data == nullptr) return 0;
  const uint8_t *b = static_cast<const uint8_t *>(data);
  uint64_t off = 0;
  if (kind == ZVEC_HIP_ROARING_FILE) {
    if (len < sizeof(BitmapFileHeader)) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
    BitmapFileHeader hd;
    memcpy(&hd, b, sizeof(hd));
    if (hd.magic != ROARING_FILE_MAGIC) return ZVEC_HIP_ERR_MISMATCH;
    off = sizeof(hd);
    if (crc32c_update(b + off, len - off, 0u) != hd.checksum) return ZVEC_HIP_ERR_MISMATCH;
    kind = hd.is_32bit ? ZVEC_HIP_ROARING_32 : ZVEC_HIP_ROARING_64MAP;
  }
  RoaringDir dir;
  if (kind == ZVEC_HIP_ROARING_32) {
    if (parse_roaring32(b + off, len - off, 0, off, dir) == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
    out.view.trunc32 = 1;
  } else if (kind == ZVEC_HIP_ROARING_64MAP) {
    if (!parse_roaring64map(b + off, len - off, off, dir)) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  } else {
    return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  }
  const size_t nc = dir.ckey.size();
  out.view.present = 1;
  out.view.nc = (uint32_t)nc;
  ZRET(out.bytes.alloc(std::max<uint64_t>(len, 1)));
  ZCHK(hipMemcpyAsync(out.bytes, b, len, hipMemcpyHostToDevice, s));
  if (nc) {
    ZRET(out.ckey.alloc(nc));
    ZRET(out.coff.alloc(nc));
    ZRET(out.cinfo.alloc(nc));
    ZCHK(hipMemcpyAsync(out.ckey, dir.ckey.data(), nc * 8, hipMemcpyHostToDevice, s));
    ZCHK(hipMemcpyAsync(out.coff, dir.coff.data(), nc * 8, hipMemcpyHostToDevice, s));
    ZCHK(hipMemcpyAsync(out.cinfo, dir.cinfo.data(), nc * 4, hipMemcpyHostToDevice, s));
  }
  ZCHK(hipStreamSynchronize(s));     // `dir` (pageable host memory) may go away now
  out.view.ckey = out.ckey; out.view.coff = out.coff; out.view.cinfo = out.cinfo; out.view.bytes = out.bytes;
  return 0;
}

int build_filter(zvec_hip_ctx_s *c, int device, const uint64_t *d_keys, uint64_t n, const uint64_t *d_dense0,
                 const uint32_t *d_tile0, uint32_t nlist, const zvec_hip_doc_filter_t *f, uint64_t *out_words,
                 int out_on_device, void *stream) {
  if (!c || !f || !out_words) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> g(c->mu);
  ZCHK(hipSetDevice(device));
  hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : c->cur;
  const uint64_t words = (n + 63) / 64;
  if (words == 0) return 0;
  DeviceRoaring del, inv;
  ZRET(upload_roaring(f->delete_bitmap, f->delete_bytes, f->delete_kind, del, s));
  ZRET(upload_roaring(f->invert_bitmap, f->invert_bytes, f->invert_bitmap ? ZVEC_HIP_ROARING_32 : ZVEC_HIP_ROARING_NONE, inv, s));
  Scoped<uint8_t> fwd;
  if (f->forward_bits) {
    const uint64_t fb = (f->forward_len + 7) / 8;
    ZRET(fwd.alloc(std::max<uint64_t>(fb, 1)));
    ZCHK(hipMemcpyAsync(fwd, f->forward_bits, fb, hipMemcpyHostToDevice, s));
  }
  Scoped<uint64_t> tmp;
  uint64_t *d_out = out_words;
  if (!out_on_device) {
    ZRET(tmp.alloc(words));
    d_out = tmp;
  }
  DocFilterArgs a{};
  a.keys = d_keys; a.n = n; a.list_dense0 = d_dense0; a.list_tile0 = d_tile0; a.nlist = nlist;
  a.del = del.view; a.inv = inv.view;
  a.forward = f->forward_bits ? static_cast<const uint8_t *>(fwd) : nullptr; a.forward_len = f->forward_len;
  a.out = d_out;
  hipLaunchKernelGGL(doc_filter_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, a);
  ZCHK(hipGetLastError());
  if (!out_on_device) ZCHK(hipMemcpyAsync(out_words, d_out, words * 8, hipMemcpyDeviceToHost, s));
  ZCHK(hipStreamSynchronize(s));     // the temporaries above are freed on return
  return 0;
}

}  // namespace

extern "C" int zvec_hip_reform_queries_dev(zvec_hip_ctx_t ctx, const float *d_in, uint32_t count, uint32_t dim, int cosine,
                                           int out_dtype, void *d_out, void *stream) {
  if (!ctx || !d_in || !d_out || dim == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (out_dtype != ZVEC_HIP_DT_FP32 && out_dtype != ZVEC_HIP_DT_FP16) return ZVEC_HIP_ERR_UNSUPPORTED;
  if (count == 0) return 0;
  std::lock_guard<std::mutex> g(ctx->mu);
  ZCHK(hipSetDevice(ctx->device));
  hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : ctx->cur;
  hipLaunchKernelGGL(reform_queries_kernel, dim3((count + 15) / 16), dim3(256), 0, s, d_in, count, dim, cosine ? 1 : 0,
                     out_dtype == ZVEC_HIP_DT_FP16 ? 1 : 0, d_out);
  ZCHK(hipGetLastError());
  return 0;
}

extern "C" uint32_t zvec_hip_crc32c(const void *data, uint64_t len, uint32_t crc) {
  return (data || len == 0) ? crc32c_update(data, len, crc) : crc;
}

extern "C" int zvec_hip_flat_build_filter(zvec_hip_flat_t h, zvec_hip_ctx_t ctx, const zvec_hip_doc_filter_t *filter,
                                          uint64_t *out_words, int out_on_device, void *stream) {
  if (!h) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  return build_filter(ctx ? ctx : h->defctx, h->device, h->st.keys, h->st.n, nullptr, nullptr, 0, filter, out_words,
                      out_on_device, stream);
}

extern "C" int zvec_hip_ivf_build_filter(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, const zvec_hip_doc_filter_t *filter,
                                         uint64_t *out_words, int out_on_device, void *stream) {
  if (!h) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (!h->loaded) return ZVEC_HIP_ERR_NO_INDEX_LOADED;
  return build_filter(ctx ? ctx : h->defctx, h->device, h->lists.keys, h->count_local, h->d_dense0, h->d_tile0, h->nlist,
                      filter, out_words, out_on_device, stream);
}

}  // extern "C"
