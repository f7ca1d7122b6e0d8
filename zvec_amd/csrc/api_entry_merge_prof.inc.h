// api_entry_merge_prof.inc.h — C ABI entry points: shard merge, measurement hook (inside extern "C")
// Part of zvec_hip_api.hip (one translation unit; included in order, not standalone).

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
  const bool was_on = ctx->profile && ctx->stats.p != nullptr;
  ctx->profile = enable != 0;
  if (ctx->profile && !was_on) {        // (already profiling: the recorded slots keep their counts; a reset goes through profile_read)
    ZCHK(hipSetDevice(ctx->device));
    ZRET(ctx->stats.ensure(sizeof(uint64_t) * 2 * PROFILE_MAX));
    // every launch slot starts at zero (the small-batch route ADDS its per-query row counts); zeroed here and at every reset,
    // not per launch: a fill kernel in front of every search is 5 us of a single query's 70 us chain
    ZCHK(hipMemsetAsync(ctx->stats.p, 0, sizeof(uint64_t) * 2 * PROFILE_MAX, ctx->cur));
    ZCHK(hipStreamSynchronize(ctx->cur));
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
  if (reset) {
    ctx->nprof = 0;
    if (ctx->stats.p) {
      ZCHK(hipMemsetAsync(ctx->stats.p, 0, sizeof(uint64_t) * 2 * PROFILE_MAX, ctx->cur));
      ZCHK(hipStreamSynchronize(ctx->cur));
    }
  }
  return 0;
}

// ---- per-box calibration ----------------------------------------------------------------------
// Boxes of one pool differ (power state, clocks: the same binary's list scan was timed at 4.67 - 5.07 ms, one box at 6.1 - 6.4):
// a bench line must let a reader tell box from code.  Two figures, measured in the same process as the timed region:
//   stream_gbs  what this box's HBM delivers to the simplest possible reader — every lane streams 16-byte non-temporal loads
//               over `bytes` of device memory, nothing else (the list scan's ceiling: it reads the same way, then computes)
//   clock_mhz   the shader clock the chip holds under that load: d(s_memtime) / d(s_memrealtime) x 100 MHz, median over the
//               work-groups (MI355X_MICROARCH.md, DVFS: the constant 100 MHz counter against the shader-clock counter)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) stream_read_kernel(const u32x4_t *p, unsigned long long n16, unsigned int *sink,
                                                          unsigned long long *stamps) {
  unsigned long long t0 = 0, r0 = 0;
  if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  u32x4_t acc = {0, 0, 0, 0};
  const unsigned long long stride = (unsigned long long)gridDim.x * 1024ull;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 1024ull + threadIdx.x; i + 768 < n16; i += stride) {
    const u32x4_t a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + 256),
                  c = __builtin_nontemporal_load(p + i + 512), d = __builtin_nontemporal_load(p + i + 768);
    acc ^= a ^ b ^ c ^ d;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u && sink) *sink = acc.x;     // (keeps the loads alive; practically never taken)
  if (threadIdx.x == 0 && stamps) {
    stamps[4 * (size_t)blockIdx.x + 0] = t0; stamps[4 * (size_t)blockIdx.x + 1] = r0;
    stamps[4 * (size_t)blockIdx.x + 2] = __builtin_amdgcn_s_memtime(); stamps[4 * (size_t)blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
  }
}

int zvec_hip_calibrate(int device, const void *d_buf, uint64_t bytes, uint32_t reps, double *clock_mhz, double *stream_gbs) {
  if (bytes < (1u << 20) || reps == 0 || reps > 64) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  ZCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  ZCHK(hipGetDeviceProperties(&prop, device));
  Scoped<char> own;
  if (!d_buf) {                               // (contents do not matter to a reader)
    ZRET(own.alloc(bytes));
    d_buf = own;
  }
  const unsigned grid = (unsigned)prop.multiProcessorCount * 8u;
  Scoped<unsigned long long> d_st;
  Scoped<unsigned int> d_sink;
  ZRET(d_st.alloc((size_t)grid * 4));
  ZRET(d_sink.alloc(1));
  hipStream_t s = nullptr;
  ZCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = 0;
  double best_ms = 1e30;
  auto run = [&]() -> int {
    ZCHK(hipEventCreate(&e0));
    ZCHK(hipEventCreate(&e1));
    const unsigned long long n16 = bytes / 16;
    for (uint32_t r = 0; r <= reps; ++r) {      // (the first launch warms up and is not counted)
      ZCHK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(stream_read_kernel, dim3(grid), dim3(256), 0, s, static_cast<const u32x4_t *>(d_buf), n16, (unsigned int *)d_sink,
                         (unsigned long long *)d_st);
      ZCHK(hipGetLastError());
      ZCHK(hipEventRecord(e1, s));
      ZCHK(hipEventSynchronize(e1));
      float ms = 0;
      ZCHK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0 && ms > 0 && ms < best_ms) best_ms = ms;
    }
    std::vector<unsigned long long> st((size_t)grid * 4);
    ZCHK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> mhz;
    for (unsigned b = 0; b < grid; ++b) {
      const double dt = (double)(st[4 * (size_t)b + 2] - st[4 * (size_t)b]), dr = (double)(st[4 * (size_t)b + 3] - st[4 * (size_t)b + 1]);
      if (dr > 0) mhz.push_back(dt / dr * 100.0);
    }
    std::sort(mhz.begin(), mhz.end());
    if (clock_mhz) *clock_mhz = mhz.empty() ? 0.0 : mhz[mhz.size() / 2];
    if (stream_gbs) *stream_gbs = best_ms < 1e29 ? (double)bytes / (best_ms * 1e-3) / 1e9 : 0.0;
    return 0;
  };
  rc = run();
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipStreamSynchronize(s);
  (void)hipStreamDestroy(s);
  return rc;
}

#ifdef ZVK_A256_STAMPS
// diagnostic build only: shader-clock cycles the waves 0 / 4 of work-group 0 of assign256_f16_kernel spent, summed over every step
// since the last call, in [group][phase 1..4][section]: 0 = from the previous stamp to the phase start (the barrier that ended the
// previous phase; for phase 1 also the fold), 1 = fragment reads + DMA issue + counted wait, 2 = the mid barrier (staggered builds),
// 3 = issuing the 16 MFMAs.  out[2*5*4] cycles; resets the counters.
int zvec_hip_debug_a256_stamps(double *out) {
  unsigned long long h[2][5][4];
  ZCHK(hipDeviceSynchronize());
  ZCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(zvk::zvk_a256_acc), sizeof(h)));
  for (int g = 0; g < 2; ++g) for (int p = 0; p < 5; ++p) for (int x = 0; x < 4; ++x) out[(g * 5 + p) * 4 + x] = (double)h[g][p][x];
  memset(h, 0, sizeof(h));
  ZCHK(hipMemcpyToSymbol(HIP_SYMBOL(zvk::zvk_a256_acc), h, sizeof(h)));
  return 0;
}
#endif

#ifdef ZVK_CLOCK_STAMP
// diagnostic build only: in-kernel clock (MHz, median over work-groups) of the LAST wide flat launch and the spread of
// the work-groups' start / end times in ms relative to the earliest start: out[0] clock, [1] median lifetime,
// [2] latest start, [3] earliest end, [4] median end, [5] latest end, [6] number of work-groups,
// [7..15] end-time percentiles 10, 25, 40, 60, 75, 90, 95, 99 and the lifetime p10
int zvec_hip_debug_flat_clock(double *out) {
  static unsigned long long h[1024][4];
  ZCHK(hipDeviceSynchronize());
  ZCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(zvk_clock_stamps), sizeof(h)));
  std::vector<double> clk, life, st, en;
  unsigned long long t0 = ~0ull;
  for (int i = 0; i < 1024; ++i)
    if (h[i][3] > h[i][1] && h[i][2] > h[i][0]) t0 = std::min(t0, h[i][1]);
  for (int i = 0; i < 1024; ++i)
    if (h[i][3] > h[i][1] && h[i][2] > h[i][0]) {
      clk.push_back((double)(h[i][2] - h[i][0]) / (double)(h[i][3] - h[i][1]) * 100.0);
      life.push_back((double)(h[i][3] - h[i][1]) / 100e3);
      st.push_back((double)(h[i][1] - t0) / 100e3);
      en.push_back((double)(h[i][3] - t0) / 100e3);
    }
  for (int i = 0; i < 16; ++i) out[i] = 0;
  if (clk.empty()) return 0;
  for (auto *v : {&clk, &life, &st, &en}) std::sort(v->begin(), v->end());
  out[0] = clk[clk.size() / 2]; out[1] = life[life.size() / 2]; out[2] = st.back(); out[3] = en.front();
  out[4] = en[en.size() / 2]; out[5] = en.back(); out[6] = (double)clk.size();
  const double pc[8] = {0.10, 0.25, 0.40, 0.60, 0.75, 0.90, 0.95, 0.99};
  for (int i = 0; i < 8; ++i) out[7 + i] = en[(size_t)(pc[i] * (en.size() - 1))];
  out[15] = life[(size_t)(0.10 * (life.size() - 1))];
  return 0;
}
#endif
