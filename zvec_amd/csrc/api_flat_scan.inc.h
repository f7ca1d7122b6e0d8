// api_flat_scan.inc.h — kernel launchers, query preparation, the flat scan orchestration (dense / fused / wide / gather paths)
// Part of zvec_hip_api.hip (one translation unit; included in order, not standalone).

namespace {

struct KernelInfo {
  bool init = false;
  int cus = 0;
};
KernelInfo g_info[16];
std::mutex g_info_mu;

// Tuning knobs.  The shipped library has NONE: every value below is a compile-time constant.  Only a build with
// -DZVEC_HIP_TUNING (tools/build_variant.sh; kernel A/B timing on one GPU box, tools/ab_flat.sh) reads them from the
// environment, so a stray variable cannot change the product path.
struct Knobs {
  int max_ng = 4;             // ZVEC_HIP_MAX_NG      cap of the 4-wave kernel's query-row groups (1, 2, 4)
  bool no_wide = false;       // ZVEC_HIP_NO_WIDE     never take the 8-wave flat tile
  bool force_wide = false;    // ZVEC_HIP_FORCE_WIDE  take it on cache-resident bases too
  bool no_seed = false;       // ZVEC_HIP_NO_SEED     no prefix scan to seed the admission bounds
  bool no_gather = false;     // ZVEC_HIP_NO_GATHER   sparse filters: compact the kept rows instead of gathering them
  int ivf_tpc = 0;            // ZVEC_HIP_IVF_TPC     fixed tiles per IVF chunk (0 = adaptive)
  bool m16_small = true;      // ZVEC_HIP_NO_M16_SMALL  keep flat scans of <= 16 queries on the 32-row MFMA shape
  int ivf_head_pct = 50;      // ZVEC_HIP_IVF_HEAD_PCT  share of the tiles (deal-order head) dealt in double-length chunks
  int ivf_occ_cap = 0;        // ZVEC_HIP_IVF_OCC_CAP   IVF list scan: at most this many persistent work-groups per CU (0 = all)
  bool no_wide_dump = false;  // ZVEC_HIP_NO_WIDE_DUMP  dense-score path (IVF coarse step): never take the 8-wave tile
  int ivf_direct_q = 8;       // ZVEC_HIP_IVF_DIRECT_Q  IVF searches of at most this many queries take the direct (wave per row) route
  int seed_rows256 = 16384;   // ZVEC_HIP_SEED_ROWS256  rows of the bound-seeding prefix in front of the 256 x 256 fp16 tile (multiple of 128; 4096 elsewhere)
  Knobs() {
#ifdef ZVEC_HIP_TUNING
    if (const char *e = getenv("ZVEC_HIP_MAX_NG")) max_ng = std::max(1, std::min(4, atoi(e)));
    no_wide = getenv("ZVEC_HIP_NO_WIDE") != nullptr;
    force_wide = getenv("ZVEC_HIP_FORCE_WIDE") != nullptr;
    no_seed = getenv("ZVEC_HIP_NO_SEED") != nullptr;
    no_gather = getenv("ZVEC_HIP_NO_GATHER") != nullptr;
    if (const char *e = getenv("ZVEC_HIP_IVF_TPC")) ivf_tpc = std::max(1, atoi(e));
    m16_small = getenv("ZVEC_HIP_NO_M16_SMALL") == nullptr;
    no_wide_dump = getenv("ZVEC_HIP_NO_WIDE_DUMP") != nullptr;
    if (const char *e = getenv("ZVEC_HIP_IVF_OCC_CAP")) ivf_occ_cap = std::max(0, atoi(e));
    if (const char *e = getenv("ZVEC_HIP_IVF_HEAD_PCT")) ivf_head_pct = std::max(0, std::min(75, atoi(e)));
    if (const char *e = getenv("ZVEC_HIP_IVF_DIRECT_Q")) ivf_direct_q = std::max(0, atoi(e));
    if (const char *e = getenv("ZVEC_HIP_SEED_ROWS256")) seed_rows256 = std::max(128, atoi(e) / 128 * 128);
#endif
  }
};
const Knobs &knobs() {
  static const Knobs k;
  return k;
}

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
  if (a.mode == 1 && knobs().ivf_occ_cap > 0) occ = std::min(occ, knobs().ivf_occ_cap);
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

// the 8-wave 128x128 flat tile (scan8_kernel); *occ_out = work-groups per CU it reaches for this k.  (Round 3 measured a 4-wave
// form of the same tile — a 64 x 64 block per wave, the labelling kernel's shape — with and without an in-register tile
// pre-filter: 118.9 / 116.1 against 124.5 TFLOP/s on one box; DESIGN.md §3.  It is not kept.)
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

// wide fp16 flat scan on the 256 x 256 multi-phase tile (zvk_scan256.hip.h): ONE work-group of 8 waves per CU
int launch_scan256_f16(const ScanArgs &a, uint32_t max_items, int cus, hipStream_t stream) {
  static bool attr_set[16] = {false};
  const size_t lds = scan256_lds_bytes(a.k);
  if (lds > LDS_LIMIT || a.nq < (uint32_t)S256_ROWS || a.nks < 2) return ZVEC_HIP_ERR_UNSUPPORTED;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_set[dev & 15]) {
    ZCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&scan256_f16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT));
    attr_set[dev & 15] = true;
  }
  const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)max_items, (uint64_t)cus);
  if (grid == 0) return 0;
  hipLaunchKernelGGL(scan256_f16_kernel, dim3(grid), dim3(512), lds, stream, a);
  ZCHK(hipGetLastError());
  return 0;
}

// nearest-centroid assignment (zvk_assign.hip.h): one work item = 128 rows x every centroid, two work-groups per CU
template <bool F16>
int launch_assign(const AssignArgs &a, int cus, hipStream_t s) {
  static bool attr_set[16] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_set[dev & 15]) {
    ZCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&assign_kernel<F16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ASSIGN_LDS));
    attr_set[dev & 15] = true;
  }
  const uint32_t items = (a.nq + ASSIGN_ROWS - 1) / ASSIGN_ROWS;
  const uint32_t grid = std::min<uint32_t>(items, (uint32_t)cus * 2u);
  if (grid == 0) return 0;
  hipLaunchKernelGGL(assign_kernel<F16>, dim3(grid), dim3(256), ASSIGN_LDS, s, a);
  ZCHK(hipGetLastError());
  return 0;
}

// fp16 rows on the 256 x 256 multi-phase tile (zvk_assign256.hip.h): one work item = 256 rows x every centroid, ONE work-group
// of 8 waves per CU (129 KiB of LDS)
template <bool L2>
int launch_assign256_f16_t(const AssignArgs &a, int cus, hipStream_t s) {
  static bool attr_set[16] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_set[dev & 15]) {
    ZCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&assign256_f16_kernel<L2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)A256_LDS));
    attr_set[dev & 15] = true;
  }
  const uint32_t items = (a.nq + A256_ROWS - 1) / A256_ROWS;
  const uint32_t grid = std::min<uint32_t>(items, (uint32_t)cus);
  if (grid == 0) return 0;
  hipLaunchKernelGGL(assign256_f16_kernel<L2>, dim3(grid), dim3(512), A256_LDS, s, a);
  ZCHK(hipGetLastError());
  return 0;
}
int launch_assign256_f16(const AssignArgs &a, int cus, hipStream_t s) {
  return a.metric == METRIC_L2 ? launch_assign256_f16_t<true>(a, cus, s) : launch_assign256_f16_t<false>(a, cus, s);
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

int pick_ng(uint32_t rows_wanted, uint32_t k) {
  int ng = knobs().max_ng;
  // 128 query rows per work-group is the largest tile whose accumulators + staging fit 512 registers
  while (ng > 1 && (uint32_t)(ng / 2) * QGROUP >= rows_wanted) ng /= 2;
  while (ng >= 1 && scan_lds_bytes(ng, k) > LDS_LIMIT - 1024) ng /= 2;
  return ng;  // 0 => k too large for the LDS-resident lists
}

// Decomposition of a wide flat scan into (chunk x query tile) items: one item per resident work-group slot (equal
// items, ONE round).  Finer items dealt dynamically through per-XCD counters were measured in round 2: they even out the
// work-groups' end times but not the launch (116-124 vs 123-124 TFLOP/s), so the static round stays.
struct FlatSplit {
  uint32_t tpc = 1, nchunks = 1;
};
FlatSplit flat_split(uint64_t ntiles, uint32_t nqtiles, uint64_t resident) {
  FlatSplit f;
  const uint64_t slots_q = std::max<uint64_t>(1, (resident + nqtiles - 1) / nqtiles);    // chunks in flight per query tile
  uint64_t tpc = std::max<uint64_t>(1, (ntiles + slots_q - 1) / slots_q);
  // >= 4 tiles per top-k warm-up — unless the base is too small to fill the chip that way
  tpc = std::max<uint64_t>(tpc, std::min<uint64_t>(ntiles, ntiles >= 4 * resident ? 4 : 1));
  f.tpc = (uint32_t)tpc;
  f.nchunks = (uint32_t)((ntiles + tpc - 1) / tpc);
  return f;
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

// gate_enter / gate_leave bracket a gated scan launch on `stream` (no-ops without a gate); the gate's mutex is held in
// between so that wait, launch and record of one context are not interleaved with another thread's
void gate_enter(zvec_hip_ctx_s *ctx, hipStream_t stream) {
  if (!ctx->gate) return;
  ctx->gate->mu.lock();
  if (ctx->gate->armed) (void)hipStreamWaitEvent(stream, ctx->gate->ev, 0);
}
void gate_leave(zvec_hip_ctx_s *ctx, hipStream_t stream) {
  if (!ctx->gate) return;
  (void)hipEventRecord(ctx->gate->ev, stream);
  ctx->gate->armed = true;
  ctx->gate->mu.unlock();
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
    ZRET(ctx->seed_idx.ensure((size_t)count * topk * sizeof(uint32_t)));
    ZRET(ctx->part_s.ensure((size_t)count * SEED_ROWS * sizeof(float)));
    ScanArgs d = a;
    d.k = 1; d.n = SEED_ROWS; d.ndense = SEED_ROWS; d.tiles_per_chunk = 1; d.nchunks = SEED_ROWS / TILE_N; d.nqtiles = nqtiles;
    d.dump = ctx->part_s.as<float>(); d.dump_stride = SEED_ROWS;
    ZRET(launch_scan8(d, st.f16, ((d.nchunks + 7) / 8) * 8 * nqtiles, cus, stream));
    MergeArgs m{};
    m.part_s = d.dump; m.slots_per_q = 1; m.slot_stride = 1; m.k = topk; m.slot_len = SEED_ROWS; m.threshold = threshold;
    m.out_keys = ctx->seed_keys.as<uint64_t>(); m.out_scores = ctx->seed_scores.as<float>(); m.out_counts = ctx->seed_counts.as<uint32_t>();
    m.out_idx = ctx->seed_idx.as<uint32_t>();          // logical (gathered) positions: mapped through d_pos below
    hipLaunchKernelGGL(merge_kernel, dim3(count), dim3(64), (size_t)topk * 12 + 16, stream, m);
    hipLaunchKernelGGL(seed_gtau_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, ctx->gtau.as<uint32_t>(),
                       m.out_scores, m.out_idx, d_pos, m.out_counts, ctx->qnorm.as<float>(), st.bnorm, st.metric, count, topk);
    ZCHK(hipGetLastError());
  }
  int occ8 = 1;
  a.k = topk;
  ZRET(launch_scan8(a, st.f16, 0, cus, stream, &occ8));
  const uint64_t ntiles = ((uint64_t)kept + TILE_N - 1) / TILE_N;
  const uint64_t resident = (uint64_t)cus * occ8;
  const FlatSplit fs = flat_split(ntiles, nqtiles, resident);
  const uint64_t tpc = fs.tpc;
  const uint32_t nchunks = fs.nchunks;
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

// Dense scores of queries [q0, q0 + cnt) of the prepared batch against every row of the store: the scan kernel in dump
// mode writes score[query][padded position] once into ctx->part_s (excluded and padding positions hold +inf); the
// caller selects from the rows (merge_kernel: large k, coarse step; the group-by kernels).  The caller has sized
// ctx->part_s for cnt rows of ceil(n / 128) * 128 floats.
int flat_dense_scores(zvec_hip_ctx_s *ctx, const Store &st, uint32_t q0, uint32_t cnt, float threshold, const uint64_t *d_exclude,
                      hipStream_t stream, float **dump, uint32_t *dump_stride) {
  const uint64_t ntiles_d = (st.n + TILE_N - 1) / TILE_N;
  const int cus_d = device_cus(ctx);
  int ngd = pick_ng(cnt, 1);
  // wide batches: the 8-wave 128x128 tile in dump mode (LDS-DMA staging, 32 accumulators per wave) — one item per
  // (tile, 128-query tile); taken when that gives at least one work-group per two CUs
  const uint32_t nqt8 = (cnt + W8_ROWS - 1) / W8_ROWS;
  const bool wide_d = !knobs().no_wide_dump && cnt > 2 * QGROUP && ngd == 4 && d_exclude == nullptr &&
                      ntiles_d * nqt8 * 2 >= (uint64_t)cus_d && scan8_lds_bytes(1) <= LDS_LIMIT - 1024;
  // one item per (tile, query tile): halve the query tile while the items would not fill two work-groups per CU
  // (1024 x 4096 coarse scores: 256 items at 128 rows -> 512 at 64 rows, 92 -> 79 us)
  while (!wide_d && ngd > 2 && ntiles_d * ((cnt + ngd * QGROUP - 1) / (ngd * QGROUP)) < 2ull * cus_d) ngd /= 2;
  const uint32_t rows_d = wide_d ? W8_ROWS : ngd * QGROUP;
  const uint32_t nqt = (cnt + rows_d - 1) / rows_d;
  ScanArgs a{};
  a.base = st.base; a.bnorm = st.bnorm; a.exclude = reinterpret_cast<const uint32_t *>(d_exclude);
  a.queries = ctx->qpad.as<float>() + (size_t)q0 * st.dpad; a.qnorm = ctx->qnorm.as<float>() + q0;
  a.dpad = st.dpad; a.nks = st.dpad / TILE_K; a.metric = st.metric; a.k = 1; a.threshold = threshold;
  a.mode = 0; a.nq = cnt; a.n = st.n; a.ndense = st.n; a.tiles_per_chunk = 1; a.nchunks = (uint32_t)ntiles_d; a.nqtiles = nqt;
  a.gtau = ctx->gtau.as<uint32_t>() + q0;
  a.dump = ctx->part_s.as<float>(); a.dump_stride = (uint32_t)(ntiles_d * TILE_N);
  a.part_s = nullptr; a.part_i = nullptr;
  if (wide_d) ZRET(launch_scan8(a, st.f16, (uint32_t)((ntiles_d + 7) / 8) * 8 * nqt, cus_d, stream));
  else ZRET(launch_scan_ng(ngd, a, st.f16, (uint32_t)ntiles_d * nqt, cus_d, stream));
  *dump = a.dump;
  *dump_stride = a.dump_stride;
  return 0;
}

// `user_facing`: a search whose lists go back to the caller (profiled, L2-refined); false for the IVF
// coarse pass and the k-means labelling, which only need the ranking
int flat_scan_prepared(zvec_hip_ctx_s *ctx, const Store &st, uint32_t count, uint32_t topk, float threshold,
                       const uint64_t *d_exclude, const SearchOut &out_in, hipStream_t stream, bool user_facing) {
  const bool profile_it = user_facing || ctx->shadow_scan;      // (the scan over a shadow store IS the search's dominant scan)
  ctx->shadow_scan = false;                                     // (... its nested seeding pre-pass is not)
  SearchOut out = out_in;
  if (user_facing && st.metric == ZVEC_HIP_METRIC_L2 && out.idx == nullptr) {
    ZRET(ctx->ridx.ensure((size_t)count * topk * sizeof(uint32_t)));
    out.idx = ctx->ridx.as<uint32_t>();
  }
  if (st.n == 0) {
    // no rows: empty results
    ZCHK(hipMemsetAsync(out.counts, 0, sizeof(uint32_t) * count, stream));
    ZCHK(hipMemsetAsync(out.keys, 0xff, sizeof(uint64_t) * (size_t)count * topk, stream));
    return 0;
  }
  // Sparse keep-set: compact the kept rows and scan those (work ~ kept rows, like the CPU's skip-before-distance)
  // (a scan over shadow rows takes the gather variant only: its positions are stored positions, which the fp32 re-scoring needs; the
  // copying variant numbers the rows of the compacted copy)
  if (d_exclude != nullptr && profile_it && st.n >= 65536) {
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
    if ((double)kept <= (can_gather ? 0.9 : 0.5) * (double)st.n && (user_facing || can_gather)) {
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
    // (tests force the 256 x 256 tile onto small bases: option scan256 = 2)
    const bool forced256 = ropts().scan256.load(std::memory_order_relaxed) == 2 && st.f16 && count >= (uint32_t)S256_ROWS &&
                           st.dpad / TILE_K >= 2 && scan256_lds_bytes(topk) <= LDS_LIMIT;
    const bool want_a = small_base && d_exclude == nullptr && topk > 8 && row_bytes_d * count <= 128.0 * 1024 * 1024 && !forced256;
    const bool want_b = pick_ng(count, topk) < 1;
    if (want_b && !k_fits_merge) return ZVEC_HIP_ERR_UNSUPPORTED;
    if ((want_a || want_b) && k_fits_merge) {
      // sub-batches so that the score matrix stays <= 1 GiB
      const uint32_t sub = (uint32_t)std::max<double>(1.0, std::min<double>((double)count, std::floor(1073741824.0 / row_bytes_d)));
      ZRET(ctx->part_s.ensure((size_t)(row_bytes_d * sub)));
      for (uint32_t q0 = 0; q0 < count; q0 += sub) {
        const uint32_t cnt = std::min(sub, count - q0);
        float *dump = nullptr;
        uint32_t dump_stride = 0;
        ZRET(flat_dense_scores(ctx, st, q0, cnt, threshold, d_exclude, stream, &dump, &dump_stride));
        MergeArgs m{};
        m.part_s = dump; m.part_i = nullptr; m.part_keys = nullptr; m.slot_begin = nullptr; m.slots_per_q = 1;
        m.slot_stride = 1; m.part_counts = nullptr; m.k = topk; m.slot_len = dump_stride; m.threshold = threshold;
        m.keymap = st.keys; m.out_keys = out.keys + (size_t)q0 * topk; m.out_scores = out.scores + (size_t)q0 * topk;
        m.out_idx = out.idx ? out.idx + (size_t)q0 * topk : nullptr; m.out_counts = out.counts + q0;
        hipLaunchKernelGGL(merge_kernel, dim3(cnt), dim3(64), (size_t)topk * 12 + 16, stream, m);
        ZCHK(hipGetLastError());
      }
      if (user_facing) ZRET(refine_l2(ctx, st, count, topk, threshold, out.keys, out.scores, out.idx, out.counts, stream));
      return 0;
    }
  }
  // The fused shapes: a base that stays in the 256 MiB Infinity Cache (IVF centroids, k-means codebooks) can be re-read by
  // every query tile for free and prefers many small query tiles; wide batches over a streamed base take the 8-wave 128 x 128
  // tile; fp16 rows under at least 256 queries with a short list the 256 x 256 multi-phase tile (one work-group per CU).
  const bool cache_resident = (double)st.n * st.dpad * 4.0 <= 64.0 * 1024 * 1024;
  const int opt256 = ropts().scan256.load(std::memory_order_relaxed);      // (2: on cache-resident bases too — the tests' small cases)
  const bool can256 = opt256 != 0 && st.f16 && d_exclude == nullptr && count >= (uint32_t)S256_ROWS && st.dpad / TILE_K >= 2 &&
                      scan256_lds_bytes(topk) <= LDS_LIMIT && (uint64_t)count * st.dpad * 4ull < (1ull << 32);   // (32-bit query offsets)
  const bool wide = !knobs().no_wide && (!cache_resident || knobs().force_wide || (opt256 == 2 && can256)) && pick_ng(count, topk) == 4 &&
                    count > 2 * QGROUP && scan8_lds_bytes(topk) <= LDS_LIMIT - 1024;
  const bool wide256 = wide && can256;
  // Bound seeding: every work-group of the fused scan starts its lists empty, and filling a list costs ~k ln(rows/k)
  // sorted insertions per (query, chunk) — with hundreds of chunks in flight that warm-up is most of the admission
  // work.  A scan of a small prefix first (its k-th score bounds the final k-th from above) lets every chunk start
  // with a bound that only ~k * chunk_rows / sample_rows of its rows pass.
  const uint64_t SEED_ROWS = wide256 ? (uint64_t)knobs().seed_rows256 : 4096;
  if (!knobs().no_seed && st.n >= 64 * 4096 && topk <= 64 && count >= 16) {
    ZRET(ctx->seed_keys.ensure((size_t)count * topk * sizeof(uint64_t)));
    ZRET(ctx->seed_scores.ensure((size_t)count * topk * sizeof(float)));
    ZRET(ctx->seed_counts.ensure((size_t)count * sizeof(uint32_t)));
    ZRET(ctx->seed_idx.ensure((size_t)count * topk * sizeof(uint32_t)));
    Store view = st;                      // a view of the first SEED_ROWS rows (whole tiles of the same arrays)
    view.n = SEED_ROWS; view.cap_tiles = SEED_ROWS / TILE_N;
    int rc;
    if (d_exclude == nullptr && (double)SEED_ROWS * 4.0 * count <= 256.0 * 1024 * 1024) {
      // no selection: the prefix's scores once (dense), then the k-th smallest of 256 disjoint minima per query (seed_bound_kernel) —
      // a bound within a rank or two of the prefix's exact k-th for a sixth of the selection's time (16 384 rows x 256 queries:
      // 7 us against 37 + 6 for merge_kernel + seed_gtau_kernel)
      ZRET(ctx->part_s.ensure((size_t)SEED_ROWS * 4 * count));
      float *dump = nullptr;
      uint32_t dump_stride = 0;
      rc = flat_dense_scores(ctx, view, 0, count, threshold, nullptr, stream, &dump, &dump_stride);
      view.base = nullptr; view.bnorm = nullptr; view.extra = nullptr; view.keys = nullptr;   // the view owns nothing
      ZRET(rc);
      hipLaunchKernelGGL(seed_bound_kernel, dim3(count), dim3(256), 0, stream, dump, dump_stride, (uint32_t)SEED_ROWS, ctx->gtau.as<uint32_t>(),
                         ctx->qnorm.as<float>(), st.bnorm, st.metric, topk);
      ZCHK(hipGetLastError());
    } else {
      SearchOut so{ctx->seed_keys.as<uint64_t>(), ctx->seed_scores.as<float>(), ctx->seed_idx.as<uint32_t>(), ctx->seed_counts.as<uint32_t>()};
      rc = flat_scan_prepared(ctx, view, count, topk, threshold, d_exclude, so, stream, false);
      view.base = nullptr; view.bnorm = nullptr; view.extra = nullptr; view.keys = nullptr;   // the view owns nothing
      ZRET(rc);
      hipLaunchKernelGGL(seed_gtau_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, ctx->gtau.as<uint32_t>(),
                         so.scores, so.idx, (const uint32_t *)nullptr, so.counts, ctx->qnorm.as<float>(), st.bnorm, st.metric, count, topk);
      ZCHK(hipGetLastError());
    }
  }
  int ng = pick_ng(count, topk);
  if (cache_resident && ng > 1) ng = 1;
  if (ng < 1) return ZVEC_HIP_ERR_UNSUPPORTED;
  // a handful of queries (the product's count = 1): the 16-row MFMA shape (the IVF list-scan kernel in flat mode) does
  // half the matrix work and half the epilogue of the 32-row one, and streams the base around the L2
  // (1 query over 1M x 768: scan 0.64 -> 0.50 ms = 6.1 TB/s; 1M x 128: 0.155 -> 0.126 ms)
  const bool m16_small = knobs().m16_small && ng == 1 && count <= 16;
  const int cus = device_cus(ctx);
  int occ8 = 1;
  ScanArgs probe{};
  probe.k = topk; probe.exclude = reinterpret_cast<const uint32_t *>(d_exclude);
  if (wide && !wide256) ZRET(launch_scan8(probe, st.f16, 0, cus, stream, &occ8));
  const uint32_t rows = wide256 ? (uint32_t)S256_ROWS : wide ? W8_ROWS : ng * QGROUP;
  const uint32_t nqtiles = (count + rows - 1) / rows;
  const uint64_t ntiles = (st.n + TILE_N - 1) / TILE_N;
  uint64_t resident = wide256 ? (uint64_t)cus : wide ? (uint64_t)cus * occ8
                           : (uint64_t)cus * (ng >= 4 ? 2 : (ng == 2 ? 2 : 3));   // work-groups per CU each shape reaches
  // items are equal-sized in a flat scan, so ONE wave of work-groups (items == resident slots) is the balanced
  // choice and gives the longest tile runs per top-k warm-up
  FlatSplit fs;
  if (wide) {
    fs = flat_split(ntiles, nqtiles, resident);
    if (wide256 && (fs.tpc & 1)) {          // the kernel walks the chunk in PAIRS of tiles
      fs.tpc += 1;
      fs.nchunks = (uint32_t)((ntiles + fs.tpc - 1) / fs.tpc);
    }
  } else {
    uint64_t want_chunks = std::max<uint64_t>(1, (resident + nqtiles - 1) / nqtiles);
    uint64_t tpc1 = std::max<uint64_t>(1, (ntiles + want_chunks - 1) / want_chunks);
    // >= 4 tiles per top-k warm-up — unless the base is too small to fill the chip that way (a single query over the
    // 4096 IVF centroids: 32 one-tile items instead of 8 four-tile ones, 141 -> 40 us)
    tpc1 = std::max<uint64_t>(tpc1, std::min<uint64_t>(ntiles, ntiles >= 4 * resident ? 4 : 1));
    fs.tpc = (uint32_t)tpc1;
    fs.nchunks = (uint32_t)((ntiles + tpc1 - 1) / tpc1);
  }
  const uint64_t tpc = fs.tpc;
  uint32_t nchunks = fs.nchunks;
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
    gate_enter(ctx, stream);            // (user-facing scans only: the seeding pre-pass and coarse passes are not gated)
    double bytes = (double)st.n * st.dscan * st.elem + (double)count * st.dscan * st.elem + (double)count * topk * 12.0;
    double flops = 2.0 * (double)count * (double)st.n * st.dscan;
    pi = prof_begin(ctx, stream, bytes, flops, 0);
  }
  int lrc;
  if (wide256) lrc = launch_scan256_f16(a, ((nchunks + 7) / 8) * 8 * nqtiles, cus, stream);
  else if (wide) lrc = launch_scan8(a, st.f16, ((nchunks + 7) / 8) * 8 * nqtiles, cus, stream);   // ids padded to whole XCD groups
  else lrc = launch_scan_ng(m16_small ? 0 : ng, a, st.f16, nchunks * nqtiles, cus, stream);
  prof_end(ctx, stream, pi);
  if (profile_it) gate_leave(ctx, stream);
  ZRET(lrc);

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
                       ctx->qpad.as<float>(), st.dpad, idx, counts, count, topk, scores, 1u);
  else
    hipLaunchKernelGGL(rescore_l2_kernel<false>, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, stream, st.base,
                       ctx->qpad.as<float>(), st.dpad, idx, counts, count, topk, scores, 1u);
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
                const uint64_t *d_dst, hipStream_t stream, uint64_t *keys_out = nullptr, const uint64_t *key_src = nullptr) {
  if (st.f16)
    hipLaunchKernelGGL(pack_rows_kernel<true>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, d_rows, n, st.dim_in,
                       st.dscan, st.dpad, d_src, pos0, d_dst, st.base, st.bnorm, st.extra, keys_out, key_src);
  else
    hipLaunchKernelGGL(pack_rows_kernel<false>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, d_rows, n, st.dim_in,
                       st.dscan, st.dpad, d_src, pos0, d_dst, st.base, st.bnorm, st.extra, keys_out, key_src);
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

// rows at `n` padded positions (host array) -> host buffer, one gather launch + one copy back
int store_get_rows(zvec_hip_ctx_s *c, const Store &st, const std::vector<uint64_t> &pos, void *out) {
  const size_t n = pos.size();
  if (n == 0) return 0;
  const size_t rb = st.row_bytes();
  Scoped<uint64_t> d_pos;
  ZRET(d_pos.alloc(n));
  ZRET(c->io_q.ensure(n * rb));
  ZCHK(hipMemcpyAsync(d_pos, pos.data(), n * 8, hipMemcpyHostToDevice, c->own));
  if (st.f16)
    hipLaunchKernelGGL(unpack_rows_kernel<true>, dim3((unsigned)n), dim3(256), 0, c->own, st.base, st.extra, d_pos, st.dscan, st.dim_in, st.dpad, c->io_q.p);
  else
    hipLaunchKernelGGL(unpack_rows_kernel<false>, dim3((unsigned)n), dim3(256), 0, c->own, st.base, st.extra, d_pos, st.dscan, st.dim_in, st.dpad, c->io_q.p);
  ZCHK(hipGetLastError());
  ZCHK(hipMemcpyAsync(out, c->io_q.p, n * rb, hipMemcpyDeviceToHost, c->own));
  ZCHK(hipStreamSynchronize(c->own));
  return 0;
}

int store_append_dev(Store &st, const void *d_vecs, uint64_t n, const uint64_t *d_keys, hipStream_t stream) {
  if (n == 0) return 0;
  if (st.n + n >= 0xfffffff0ull) return ZVEC_HIP_ERR_OUT_OF_RANGE;   // positions are 32-bit (IDX_NONE reserved)
  ZRET(st.reserve(st.n + n, stream));
  ZRET(launch_pack(st, d_vecs, n, nullptr, st.n, nullptr, stream, st.keys, d_keys));    // rows, norms and keys in one launch
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
  c->gtau.release(); c->ridx.release(); c->seed_keys.release(); c->seed_scores.release(); c->seed_counts.release(); c->seed_idx.release(); c->cmp_base.release(); c->cmp_norm.release(); c->cmp_extra.release(); c->cmp_keys.release(); c->cmp_pos.release(); c->cmp_cnt.release(); c->qpad.release(); c->qnorm.release(); c->part_s.release(); c->part_i.release();
  c->coarse_keys.release(); c->coarse_scores.release(); c->coarse_idx.release(); c->coarse_cnt.release();
  c->plan.release(); c->io_q.release(); c->io_ex.release(); c->io_out.release(); c->io_cq.release();
  c->grp_ws.release(); c->grp_of.release(); c->grp_out.release(); c->grp_tab.release(); c->holes_ex.release(); c->direct_pos.release(); c->direct_keys.release(); c->direct_scores.release(); c->direct_idx.release(); c->direct_cnt.release(); c->stats.release(); c->sh_q16.release(); c->sh_qn16.release(); c->sh_qinfo.release(); c->sh_keys.release(); c->sh_scores.release(); c->sh_true.release(); c->sh_idx.release(); c->sh_counts.release(); c->sh_flags.release(); c->pin_in.release(); c->pin_out.release(); c->done_word.release();
  if (c->block_ev) (void)hipEventDestroy(c->block_ev);
  for (auto e : c->ev0) (void)hipEventDestroy(e);
  for (auto e : c->ev1) (void)hipEventDestroy(e);
  if (c->own) (void)hipStreamDestroy(c->own);
  delete c;
}

}  // namespace
