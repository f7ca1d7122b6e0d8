// api_ivf_core.inc.h — IVF search on device pointers: coarse pass, plan, list scan, merge, large-k fallback
// Part of zvec_hip_api.hip (one translation unit; included in order, not standalone).

namespace {

// ---- IVF search core (device pointers) ------------------------------------------------------
// d_coarse_queries: the queries in the centroid store's own space when it has one (h->coarse_sep: an inner-product index whose
// centroid index the reference's builder trained in MIPS-converted space, ivf_builder.cc:552-555, searched through a MipsReformer,
// ivf_centroid_index.cc:273-297); nullptr otherwise
// The coarse pass may be taken apart from the rest (sharded IVF, SURVEY §8(e): every rank needs the same probe sets, and the
// pass — 2 x Q x nlist x d flop against replicated centroids — is the one piece of a rank's step that does not shrink with the
// number of ranks): `coarse` with out_idx set = run ONLY the coarse pass and leave the probe lists there ([count][nprobe] centroid
// ids in coarse-score order + [count] valid entries); with given_idx set = skip the coarse pass and plan from those lists.
struct CoarseSplit {
  const uint32_t *given_idx = nullptr, *given_cnt = nullptr;
  uint32_t *out_idx = nullptr, *out_cnt = nullptr;
};

int ivf_search_core(zvec_hip_ivf_s *h, zvec_hip_ctx_s *ctx, const void *d_queries, uint32_t count, uint32_t topk,
                    float threshold, uint32_t nprobe, uint32_t max_scan_count, int brute_force,
                    const uint64_t *d_exclude, const SearchOut &out, hipStream_t stream, const void *d_coarse_queries = nullptr,
                    const CoarseSplit &coarse = CoarseSplit()) {
  const bool given = coarse.given_idx != nullptr, coarse_only = coarse.out_idx != nullptr;
  if ((given || coarse_only) && brute_force) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (h->coarse_sep && !brute_force && !given && d_coarse_queries == nullptr) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  const int cus = device_cus(ctx);
  const uint32_t nlist = h->nlist;
  if (nprobe < 1) nprobe = 1;
  if (nprobe > nlist) nprobe = nlist;

  // A handful of queries (the product's count = 1 calls): both steps go wave-per-row instead of through the MFMA tile
  // kernels, whose few work items would each be a chain of dependent HBM round trips (1 query, 2M x 768, nprobe 32:
  // coarse 90 us + list scan 113 us for 6 + 96 MB).  Bound of the rows one query can scan: the nprobe largest lists.
  const uint64_t direct_rows = h->h_rows_of_largest.size() > nprobe ? std::max<uint64_t>(1, h->h_rows_of_largest[nprobe]) : ~0ull;
  // (measured, 2M x 768 fp32, nprobe 32 of 2048 lists, host-pointer calls: 1 query 223 -> 116 us, 2: 244 -> 153, 4: 290 -> 222,
  // 8: 390 -> 369, 16: 586 -> 655 — the direct route costs the same for every query, the tile route shares the lists)
  const bool direct = !brute_force && !coarse_only && count <= (uint32_t)knobs().ivf_direct_q && (uint64_t)count * direct_rows <= (4u << 20) &&
                      (double)count * (double)direct_rows * (double)h->lists.row_bytes() <= 2.5e9 &&
                      (size_t)topk * 12 + 16 <= 60 * 1024 && (size_t)nprobe * 12 + 16 <= 60 * 1024;

  // The direct route scores rows straight from the prepared query rows and needs no norms: when the caller's rows already ARE
  // prepared rows (no padding: dim_in == the scanned dims == whole 128-byte k-steps; 16-byte aligned) the preparation launch is
  // skipped — one kernel and one launch gap less in a chain of eight short dependent kernels (single query, 10M x 768:
  // 5 us + gap of the ~70 us one lane needs between two of its scoring kernels).
  const bool sep = h->coarse_sep && !brute_force && !given;      // (given probe lists: no coarse pass, only the lists' own space)
  const bool raw_rows = !sep && direct && h->lists.dim_in == h->lists.dscan && (size_t)h->lists.dpad * 4 == h->lists.row_bytes() &&
                        (reinterpret_cast<uintptr_t>(d_queries) & 15u) == 0 &&
                        !(ctx->pin_in.dev && d_queries == ctx->pin_in.dev);    // (rows in the host-mapped slot are read ONCE, by prep_queries)
  const float *qrows = raw_rows ? reinterpret_cast<const float *>(d_queries) : nullptr;
  if (!raw_rows) {
    // (a separate coarse space: the coarse pass runs on the coarse queries, then the list queries are prepared in their place)
    ZRET(prep_queries(ctx, sep ? h->cent : h->lists, sep ? d_coarse_queries : d_queries, count, FLT_MAX, stream));   // coarse pass: no RNN radius
    qrows = ctx->qpad.as<float>();
  }

  // the small-batch route: stream geometry and probe-rule arguments
  constexpr uint32_t RUN = 1024;                                        // one gather round of merge_kernel
  uint32_t stride = 0, runs = 0, *d_off = nullptr;
  bool block_topk = false;
  PlanArgs dp{};
  if (direct) {
    stride = (uint32_t)((direct_rows + 63) / 64 * 64);
    if (stride > RUN) stride = (stride + RUN - 1) / RUN * RUN;          // whole runs for the two-step selection
    runs = (stride + RUN - 1) / RUN;
    // long streams, ordinary k: the scoring blocks keep only the k best of their PKEYS_BLOCK candidates (pkeys_topk_kernel) and
    // ONE merge finishes; otherwise the whole score matrix is written and selected from in one or two steps
    block_topk = runs > 1 && topk <= PKEYS_TOPK_MAX;
    ZRET(ctx->plan.ensure(((size_t)3 * count + 8) * sizeof(uint32_t)));
    uint32_t *pb = ctx->plan.as<uint32_t>();
    dp.coarse_idx = ctx->coarse_idx.as<uint32_t>(); dp.coarse_cnt = ctx->coarse_cnt.as<uint32_t>();
    dp.nq = count; dp.nprobe = nprobe; dp.nlist = nlist; dp.max_scan_count = max_scan_count; dp.brute_force = 0;
    dp.list_size = h->d_size; dp.list_size_global = h->d_size_global;
    dp.q_nprobe = pb; dp.q_scanned = pb + count;
    d_off = pb + 2 * (size_t)count;
    if (ctx->profile && ctx->nprof < PROFILE_MAX && ctx->stats.p) {       // the slot prof_begin will take for this launch
      dp.work_stats = ctx->stats.as<unsigned long long>() + 2 * (size_t)ctx->nprof;     // (zero since the last profile reset)
    }
  }

  // 1. coarse assign: flat scan over the centroids, k = nprobe (IVFCentroidIndex::search)
  const uint32_t *probe_idx = coarse.given_idx, *probe_cnt = coarse.given_cnt;
  if (given) { dp.coarse_idx = probe_idx; dp.coarse_cnt = probe_cnt; }
  if (!brute_force && !given) {
    ZRET(ctx->coarse_keys.ensure((size_t)count * nprobe * sizeof(uint64_t)));
    ZRET(ctx->coarse_scores.ensure((size_t)count * nprobe * sizeof(float)));
    ZRET(ctx->coarse_idx.ensure((size_t)count * nprobe * sizeof(uint32_t)));
    ZRET(ctx->coarse_cnt.ensure((size_t)count * sizeof(uint32_t)));
    SearchOut co{ctx->coarse_keys.as<uint64_t>(), ctx->coarse_scores.as<float>(),
                 coarse_only ? coarse.out_idx : ctx->coarse_idx.as<uint32_t>(), coarse_only ? coarse.out_cnt : ctx->coarse_cnt.as<uint32_t>()};
    probe_idx = co.idx; probe_cnt = co.counts;
    if (direct) {
      const uint32_t cstride = (nlist + 63) / 64 * 64;
      const uint64_t cpairs = (uint64_t)count * cstride;
      ZRET(ctx->part_s.ensure(cpairs * 4));
      if (h->cent.f16)
        hipLaunchKernelGGL(rows_score_kernel<true>, dim3((unsigned)((cpairs + 3) / 4)), dim3(256), 0, stream, h->cent.base,
                           qrows, h->cent.dpad, h->cent.metric, nlist, count, cstride, ctx->part_s.as<float>());
      else
        hipLaunchKernelGGL(rows_score_kernel<false>, dim3((unsigned)((cpairs + 3) / 4)), dim3(256), 0, stream, h->cent.base,
                           qrows, h->cent.dpad, h->cent.metric, nlist, count, cstride, ctx->part_s.as<float>());
      MergeArgs m{};
      m.part_s = ctx->part_s.as<float>(); m.slots_per_q = 1; m.slot_stride = 1; m.k = nprobe; m.slot_len = cstride; m.threshold = FLT_MAX;
      m.out_keys = co.keys; m.out_scores = co.scores; m.out_idx = co.idx; m.out_counts = co.counts;
      dp.coarse_idx = co.idx; dp.coarse_cnt = co.counts;                  // (the buffers may just have grown)
      hipLaunchKernelGGL(merge_kernel, dim3(count), dim3(256), (size_t)nprobe * 12 + 16, stream, m);     // four waves share the row
      ZCHK(hipGetLastError());
    } else {
      ZRET(flat_scan_prepared(ctx, h->cent, count, nprobe, FLT_MAX, nullptr, co, stream, false));
    }
    if (coarse_only) return 0;
  }
  if (sep) {
    ZRET(prep_queries(ctx, h->lists, d_queries, count, FLT_MAX, stream));      // from here on: the lists' own space
    qrows = ctx->qpad.as<float>();                                             // (the buffer may have grown: list rows wider than coarse rows)
  }
  if (direct) {
    // 2'. every query's probed rows as positions (same probe rule), 3'. one wave per (query, row): direct distance,
    // 4'. selection: the scoring blocks' own top-k lists -> the result, or (k > 64) two steps over the score matrix: runs of
    // 1024 candidates -> top-k lists -> the result (no refinement needed: the scores already are sum((q - b)^2))
    const uint64_t pairs = (uint64_t)count * stride;
    ZRET(ctx->direct_pos.ensure(pairs * 4));
    hipLaunchKernelGGL(ivf_expand_direct_kernel, dim3(nprobe, count), dim3(256), 0, stream, dp, h->d_tile0, h->d_dense0,
                       reinterpret_cast<const uint32_t *>(d_exclude), stride, d_off, ctx->direct_pos.as<uint32_t>());
    if (!block_topk) {
      ZRET(ctx->part_s.ensure(pairs * 4));
      ZRET(ctx->part_i.ensure(pairs * 4));
    }
    // NOT gated: handing the gate's event from one hardware queue to the other costs ~20 us, against a 50 us scoring kernel
    // (10M x 768, one query: two lanes 0.069 ms gated, 0.057 sharing the device freely).  (the radius is applied by the selection)
    const int pi = prof_begin(ctx, stream, (double)count * h->lists.dscan * h->lists.elem + (double)count * topk * 12.0, 0, 1);
    if (pi >= 0) ctx->prof_dscan[pi] = h->lists.dscan | (h->lists.f16 ? 0x80000000u : 0u);
    const uint32_t bpq = (stride + PKEYS_BLOCK - 1) / PKEYS_BLOCK;
    if (block_topk) {
      ZRET(ctx->direct_scores.ensure((uint64_t)count * bpq * topk * 4));
      ZRET(ctx->direct_idx.ensure((uint64_t)count * bpq * topk * 4));
      if (h->lists.f16)
        hipLaunchKernelGGL(pkeys_topk_kernel<true>, dim3(count * bpq), dim3(256), 0, stream, h->lists.base, qrows, h->lists.dpad,
                           h->metric, ctx->direct_pos.as<uint32_t>(), d_off, count, stride, topk, ctx->direct_scores.as<float>(),
                           ctx->direct_idx.as<uint32_t>());
      else
        hipLaunchKernelGGL(pkeys_topk_kernel<false>, dim3(count * bpq), dim3(256), 0, stream, h->lists.base, qrows, h->lists.dpad,
                           h->metric, ctx->direct_pos.as<uint32_t>(), d_off, count, stride, topk, ctx->direct_scores.as<float>(),
                           ctx->direct_idx.as<uint32_t>());
    } else if (h->lists.f16)
      hipLaunchKernelGGL(pkeys_score_kernel<true>, dim3(pkeys_score_blocks(count, stride)), dim3(256), 0, stream, h->lists.base,
                         qrows, h->lists.dpad, h->metric, ctx->direct_pos.as<uint32_t>(), d_off, count, stride,
                         ctx->part_s.as<float>(), ctx->part_i.as<uint32_t>());
    else
      hipLaunchKernelGGL(pkeys_score_kernel<false>, dim3(pkeys_score_blocks(count, stride)), dim3(256), 0, stream, h->lists.base,
                         qrows, h->lists.dpad, h->metric, ctx->direct_pos.as<uint32_t>(), d_off, count, stride,
                         ctx->part_s.as<float>(), ctx->part_i.as<uint32_t>());
    prof_end(ctx, stream, pi);
    ZCHK(hipGetLastError());
    if (block_topk) {
      // (equal scores: list order, then entry order = the order of the candidate stream, as below)
      MergeArgs f{};
      f.part_s = ctx->direct_scores.as<float>(); f.part_i = ctx->direct_idx.as<uint32_t>(); f.slots_per_q = bpq; f.slot_stride = 1;
      f.k = topk; f.slot_len = topk; f.threshold = threshold; f.order_by_ordinal = 1; f.keymap = h->lists.keys;
      f.out_keys = out.keys; f.out_scores = out.scores; f.out_idx = out.idx; f.out_counts = out.counts;
      hipLaunchKernelGGL(merge_kernel, dim3(count), dim3(256), (size_t)topk * 12 + 16, stream, f);
      ZCHK(hipGetLastError());
      ctx->q_nprobe = dp.q_nprobe; ctx->q_scanned = dp.q_scanned; ctx->last_count = count; ctx->last_list_count = nullptr;
      return 0;
    }
    // equal scores keep the reference's order — probe rank, then position in the list — which here is the ORDER of the
    // candidate stream, not the order of the positions
    MergeArgs m{};
    m.part_s = ctx->part_s.as<float>(); m.part_i = ctx->part_i.as<uint32_t>(); m.slots_per_q = 1; m.slot_stride = 1;
    m.k = topk; m.threshold = threshold; m.order_by_ordinal = 1;
    if (runs > 1) {
      // step 1: block (query, run) keeps the top-k of its 1024 candidates; step 2: a query's `runs` lists -> its result
      const uint64_t blocks = (uint64_t)count * runs;
      ZRET(ctx->direct_keys.ensure(blocks * topk * 8));
      ZRET(ctx->direct_scores.ensure(blocks * topk * 4));
      ZRET(ctx->direct_idx.ensure(blocks * topk * 4));
      ZRET(ctx->direct_cnt.ensure(blocks * 4));
      m.slot_len = RUN;
      m.out_keys = ctx->direct_keys.as<uint64_t>(); m.out_scores = ctx->direct_scores.as<float>();
      m.out_idx = ctx->direct_idx.as<uint32_t>(); m.out_counts = ctx->direct_cnt.as<uint32_t>();
      hipLaunchKernelGGL(merge_kernel, dim3((unsigned)blocks), dim3(64), (size_t)topk * 12 + 16, stream, m);
      MergeArgs f{};
      f.part_s = m.out_scores; f.part_i = m.out_idx; f.slots_per_q = runs; f.slot_stride = 1; f.k = topk; f.slot_len = topk;
      f.threshold = threshold; f.order_by_ordinal = 1; f.keymap = h->lists.keys;
      f.out_keys = out.keys; f.out_scores = out.scores; f.out_idx = out.idx; f.out_counts = out.counts;
      hipLaunchKernelGGL(merge_kernel, dim3(count), dim3(256), (size_t)topk * 12 + 16, stream, f);
    } else {
      m.slot_len = stride;
      m.keymap = h->lists.keys; m.out_keys = out.keys; m.out_scores = out.scores; m.out_idx = out.idx; m.out_counts = out.counts;
      hipLaunchKernelGGL(merge_kernel, dim3(count), dim3(64), (size_t)topk * 12 + 16, stream, m);
    }
    ZCHK(hipGetLastError());
    ctx->q_nprobe = dp.q_nprobe; ctx->q_scanned = dp.q_scanned; ctx->last_count = count; ctx->last_list_count = nullptr;
    return 0;
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
    p.coarse_idx = probe_idx; p.coarse_cnt = probe_cnt;
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
      hipLaunchKernelGGL(pkeys_score_kernel<true>, dim3(pkeys_score_blocks(count, maxlen)), dim3(256), 0, stream, h->lists.base,
                         ctx->qpad.as<float>(), h->lists.dpad, h->metric, d_pos, d_off, count, (uint32_t)maxlen,
                         ctx->part_s.as<float>(), ctx->part_i.as<uint32_t>());
    else
      hipLaunchKernelGGL(pkeys_score_kernel<false>, dim3(pkeys_score_blocks(count, maxlen)), dim3(256), 0, stream, h->lists.base,
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
  // Half-width pre-selection (zvk_shadow.hip.h): the scan reads the fp16 twin of the lists and keeps k' > k rows per query, which
  // are re-scored on the fp32 rows and certified below.  Not with an RNN radius (it would have to be widened by the rounding bound),
  // not for search_bf, not on the certify step's own re-run.
  uint32_t kp = 0;
  if (h->shadow_on && h->shadow.base && !ctx->shadow_skip && !brute_force && !(threshold < FLT_MAX) && topk <= 32 && h->shadow_gov.allow()) {
    kp = ctx->shadow_force_kp ? ctx->shadow_force_kp : h->shadow_kp ? h->shadow_kp : h->shadow_gov.kp_auto(topk);
    kp = std::min<uint32_t>(kp, 64);                                     // (shadow_select_kernel: one candidate per lane)
    if (kp <= topk || scan_lds_bytes(1, kp, true) > LDS_LIMIT - 1024) kp = 0;
  }
  const bool use_shadow = kp != 0;
  const uint32_t ks = use_shadow ? kp : topk;                            // k of the list scan and of its merge
  ctx->sh_count = 0;
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
    // never one-tile items: they double the slots the merge has to fold for nothing (measured, 10M x 768, nprobe 38:
    // batch 1 best at 2 tiles, batch 8 at 4, batch 32 at 4-8, batch 1024 at the index default)
    const uint64_t lo = est_tiles >= 4096 ? 4 : 2;
    tpc = (uint32_t)std::min<uint64_t>(h->tiles_per_chunk, std::max<uint64_t>(lo, t));
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
  p.coarse_idx = probe_idx; p.coarse_cnt = probe_cnt;
  p.nq = count; p.nprobe = nprobe; p.nlist = nlist; p.max_scan_count = max_scan_count; p.brute_force = brute_force;
  p.list_size = h->d_size; p.list_size_global = h->d_size_global; p.list_order = h->d_order;
  p.list_tpc = pb + o_ltpc;
  p.rows_per_group = rows_per_group;
  p.q_nprobe = pb + o_qnprobe; p.q_scanned = pb + o_qscanned; p.q_nslots = pb + o_qnslots; p.slot_begin = pb + o_slotbegin;
  p.list_count = pb + o_lcount; p.list_fill = pb + o_lfill; p.list_qoff = pb + o_lqoff; p.item_off = pb + o_itemoff;
  p.total_items = pb + o_total; p.csr_q = pb + o_csrq; p.csr_slot = pb + o_csrslot;
  ctx->q_nprobe = p.q_nprobe; ctx->q_scanned = p.q_scanned; ctx->last_count = count;
  ctx->last_list_count = p.list_count;
  if (ctx->profile && ctx->nprof < PROFILE_MAX && ctx->stats.p)      // the slot prof_begin will take for this launch
    p.work_stats = ctx->stats.as<unsigned long long>() + 2 * (size_t)ctx->nprof;
  hipLaunchKernelGGL(plan_wave_kernel<false>, dim3((count + 3) / 4), dim3(256), 0, stream, p);
  // (a pipelined context — one of several lanes, i.e. gated or driven on a caller's stream — runs beside another lane's
  // resident list scan: a 1024-thread work-group is not dispatched until that scan drains — 714 us for this 20 us kernel
  // in the round-2 trace — four waves are)
  const bool pipelined = ctx->gate != nullptr || stream != ctx->own;
  hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(pipelined ? 256 : 1024), 0, stream, p);
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
      const uint32_t t = level_tpc(h->h_tail[l], tpc);
      chunks[l] = sz[l] ? (((sz[l] + TILE_N - 1) / TILE_N + t - 1) / t) : 0;
    }
    uint32_t np = brute_force ? nlist : nprobe;
    std::partial_sort(chunks.begin(), chunks.begin() + np, chunks.end(), std::greater<uint32_t>());
    uint64_t s = 0;
    for (uint32_t i = 0; i < np; ++i) s += chunks[i];
    slots_bound = s * count;
  }
  if (slots_bound == 0) slots_bound = 1;
  ZRET(ctx->part_s.ensure(slots_bound * ks * sizeof(float)));
  ZRET(ctx->part_i.ensure(slots_bound * ks * sizeof(uint32_t)));

  const Store &sst = use_shadow ? h->shadow : h->lists;                  // the store the list scan streams
  ScanArgs a{};
  a.base = sst.base; a.bnorm = sst.bnorm; a.exclude = reinterpret_cast<const uint32_t *>(d_exclude);
  a.queries = ctx->qpad.as<float>(); a.qnorm = ctx->qnorm.as<float>();
  if (use_shadow) {
    ZRET(ctx->sh_q16.ensure((size_t)count * sst.dpad * sizeof(float)));
    ZRET(ctx->sh_qn16.ensure((size_t)count * sizeof(float)));
    ZRET(ctx->sh_qinfo.ensure((size_t)count * sizeof(f32x2)));
    hipLaunchKernelGGL(shadow_prep_queries_kernel, dim3((count + 3) / 4), dim3(256), 0, stream, reinterpret_cast<const float *>(d_queries),
                       count, h->lists.dim_in, sst.dscan, sst.dpad, ctx->sh_q16.as<float>(), ctx->sh_qn16.as<float>(),
                       ctx->sh_qinfo.as<f32x2>());
    ZCHK(hipGetLastError());
    a.queries = ctx->sh_q16.as<float>(); a.qnorm = ctx->sh_qn16.as<float>();
  }
  a.dpad = sst.dpad; a.nks = sst.dpad / TILE_K; a.metric = h->metric; a.k = ks; a.threshold = threshold;
  a.gtau = ctx->gtau.as<uint32_t>();
  a.mode = 1; a.nq = count; a.n = h->lists.n; a.ndense = h->count_local; a.tiles_per_chunk = tpc; a.list_tpc = pb + o_ltpc;
  a.total_items = p.total_items; a.queue = pb + o_queue; a.list_order = h->d_order; a.item_off = p.item_off; a.list_tile0 = h->d_tile0; a.list_size = h->d_size;
  a.list_dense0 = h->d_dense0; a.list_qoff = p.list_qoff; a.csr_q = p.csr_q; a.csr_slot = p.csr_slot; a.nlist = nlist;
  a.part_s = ctx->part_s.as<float>(); a.part_i = ctx->part_i.as<uint32_t>();
  // algorithmic bytes of the list scan = rows of the DISTINCT probed lists (counted on device from
  // the plan: plan_scan_kernel's work_stats) + the query rows + the result lists
  gate_enter(ctx, stream);
  int pi = prof_begin(ctx, stream, (double)count * sst.dscan * sst.elem + (double)count * ks * 12.0, 0, 1);
  if (pi >= 0) ctx->prof_dscan[pi] = sst.dscan | (sst.f16 ? 0x80000000u : 0u);
  const int lrc = launch_scan_ng(ng, a, sst.f16, 0x7fffffffu, cus, stream);
  prof_end(ctx, stream, pi);
  gate_leave(ctx, stream);
  ZRET(lrc);

  // 4. merge the per-(query, probe, chunk) partial lists in probe order
  MergeArgs m{};
  m.part_s = a.part_s; m.part_i = a.part_i; m.part_keys = nullptr; m.slot_begin = p.slot_begin; m.slots_per_q = 0;
  m.slot_stride = 1; m.part_counts = nullptr; m.k = ks; m.slot_len = ks; m.threshold = threshold; m.keymap = h->lists.keys;
  m.bound_keys = a.gtau;
  if (use_shadow) {
    // the k' pre-selected rows of every query in shadow-score order -> their true scores -> the k best + the certificate
    const size_t ck = (size_t)count * kp;
    ZRET(ctx->sh_keys.ensure(ck * sizeof(uint64_t)));
    ZRET(ctx->sh_scores.ensure(ck * sizeof(float)));
    ZRET(ctx->sh_true.ensure(ck * sizeof(float)));
    ZRET(ctx->sh_idx.ensure(ck * sizeof(uint32_t)));
    ZRET(ctx->sh_counts.ensure((size_t)count * sizeof(uint32_t)));
    ZRET(ctx->sh_flags.ensure(((size_t)count + 4) * sizeof(uint32_t)));
    m.out_keys = ctx->sh_keys.as<uint64_t>(); m.out_scores = ctx->sh_scores.as<float>(); m.out_idx = ctx->sh_idx.as<uint32_t>();
    m.out_counts = ctx->sh_counts.as<uint32_t>();
    hipLaunchKernelGGL(merge_kernel, dim3(count), dim3(merge_threads(count)), (size_t)kp * 12 + 16, stream, m);
    hipLaunchKernelGGL(shadow_rescore_kernel<false>, dim3((unsigned)((ck + 3) / 4)), dim3(256), 0, stream, h->lists.base,
                       ctx->qpad.as<float>(), h->lists.dpad, h->metric, m.out_idx, m.out_counts, count, kp, ctx->sh_true.as<float>());
    ZCHK(hipMemsetAsync(ctx->sh_flags.as<uint32_t>() + count, 0, sizeof(uint32_t), stream));
    ShadowSelectArgs sa{};
    sa.c_keys = m.out_keys; sa.c_shadow = m.out_scores; sa.c_true = ctx->sh_true.as<float>(); sa.c_idx = m.out_idx;
    sa.c_counts = m.out_counts; sa.qinfo = ctx->sh_qinfo.as<f32x2>(); sa.facts = static_cast<const ShadowFacts *>(h->d_shadow_facts);
    sa.kp = kp; sa.k = topk; sa.dscan = h->lists.dscan; sa.metric = h->metric;
    sa.out_keys = out.keys; sa.out_scores = out.scores; sa.out_idx = out.idx; sa.out_counts = out.counts;
    sa.flags = ctx->sh_flags.as<uint32_t>(); sa.nflag = sa.flags + count;
    hipLaunchKernelGGL(shadow_select_kernel, dim3(count), dim3(64), 0, stream, sa);
    ZCHK(hipGetLastError());
    ctx->sh_count = count;
    ctx->sh_kp = kp;
    return 0;
  }
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

constexpr size_t PIN_LIMIT = 256u << 10;  // staging through pinned memory up to 256 KiB per direction (above that the extra
                                          // host copy costs more than the pageable transfer: batch 1024 lost 2 %)

// wait policy 1: the last thing on the stream writes the call's epoch into a pinned word the host polls
__global__ void done_word_kernel(volatile uint32_t *word, uint32_t epoch) {
  __threadfence_system();
  *word = epoch;
}

// A host-pointer call waits for its stream here (RuntimeOpts::wait).  Everything the call enqueued is on `stream`.
int host_wait(zvec_hip_ctx_s *ctx, hipStream_t stream) {
  const int policy = ropts().wait.load(std::memory_order_relaxed);
  if (policy == 1 && ctx->done_word.ensure(64) == 0 && ctx->done_word.dev) {
    // a plain word in pinned memory: no runtime call (and none of its locks) per poll; a short spin for the single caller's
    // latency, then the CPU is handed on between polls — with more callers than CPUs the waiting threads no longer hold the
    // time slices the launching ones need
    volatile uint32_t *w = static_cast<volatile uint32_t *>(ctx->done_word.p);
    const uint32_t epoch = ++ctx->done_epoch ? ctx->done_epoch : ++ctx->done_epoch;      // (0 = the word's initial value)
    hipLaunchKernelGGL(done_word_kernel, dim3(1), dim3(1), 0, stream, static_cast<volatile uint32_t *>(ctx->done_word.dev), epoch);
    ZCHK(hipGetLastError());
    // Spin while the answer is close, sleep while it is not.  A lone caller's search ends ~0.1 ms after its last launch: it is
    // found by the spin.  With many callers sharing the GPU a call waits a millisecond or more; a thread that spun (or
    // sched_yield-ed on a core of its own) through that would burn a whole CPU — 64 such callers exhaust a 16-CPU quota in a
    // quarter of every scheduler period and are then all frozen for the rest of it (measured: p99 78 ms, throughput / 3).  So a
    // context whose previous wait was long goes to sleep at once, in steps that follow the time it has already waited.
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&]() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); };
    const uint64_t est = ctx->last_wait_ns;                 // how long this context's waits have been taking
    if (est > 300000) {
      // sleeping mode.  HALF the expected wait in one sleep, the rest in steps of 1/32 of it: the completion is then noticed within
      // ~3 % (+ the timer slack) and MEASURED (it falls into the polled half), so the estimate follows the truth both ways.  A
      // sleep that wakes to a finished call has only learnt "at most this long": the estimate is halved — it must never feed on
      // its own oversleeping (an estimate of 60 ms once, from a first call that allocated its workspace, would otherwise hold
      // every later 5 ms call for 45 ms).
      const uint64_t first = std::min<uint64_t>(est / 2, 20000000);
      struct timespec ts = {0, (long)first};
      nanosleep(&ts, nullptr);
      if (*w == epoch) { ctx->last_wait_ns = first; return 0; }
      const uint64_t step = std::min<uint64_t>(200000, std::max<uint64_t>(20000, est / 32));
      for (uint32_t it = 0;; ++it) {
        struct timespec t2 = {0, (long)step};
        nanosleep(&t2, nullptr);
        if (*w == epoch) { ctx->last_wait_ns = elapsed(); return 0; }
        if ((it & 63) == 63) {
          // a fault on the stream would leave the word unwritten for ever: ask the runtime now and then
          hipError_t q = hipStreamQuery(stream);
          if (q != hipSuccess && q != hipErrorNotReady) { ZCHK(q); }
          if (q == hipSuccess && *w != epoch && elapsed() > 5000000000ull) break;
        }
      }
    } else {
      // spinning mode: the answer is ~0.1 ms away
      for (uint32_t it = 0;; ++it) {
        if (*w == epoch) { ctx->last_wait_ns = elapsed(); return 0; }
        __builtin_ia32_pause();
        if ((it & 63) == 63 && elapsed() > 400000) { ctx->last_wait_ns = 600000; break; }     // longer than a lone call: sleep from now on
      }
      for (uint32_t it = 0;; ++it) {
        struct timespec t2 = {0, 50000};
        nanosleep(&t2, nullptr);
        if (*w == epoch) { ctx->last_wait_ns = elapsed(); return 0; }
        if ((it & 63) == 63) {
          hipError_t q = hipStreamQuery(stream);
          if (q != hipSuccess && q != hipErrorNotReady) { ZCHK(q); }
          if (q == hipSuccess && *w != epoch && elapsed() > 5000000000ull) break;
        }
      }
    }
    ZCHK(hipStreamSynchronize(stream));
    return 0;
  }
  if (policy == 2) {
    if (!ctx->block_ev) ZCHK(hipEventCreateWithFlags(&ctx->block_ev, hipEventBlockingSync | hipEventDisableTiming));
    ZCHK(hipEventRecord(ctx->block_ev, stream));
    ZCHK(hipEventSynchronize(ctx->block_ev));
    return 0;
  }
  ZCHK(hipStreamSynchronize(stream));
  return 0;
}

// Host-pointer entry points: queries (and an exclude set) in, keys | scores | counts out.  Small transfers skip the copy engine
// (RuntimeOpts::zerocopy): the rows are put into the context's pinned slot, which is mapped into the device's address space,
// and the first kernel of the search — prep_queries — reads them over the host link in place (ctx->io_qp); the result arrays
// are views into the mapped result slot, so the last kernels of the chain (merge / re-score / re-sort) write the answer straight
// into host memory and the call ends with ONE wait and three memcpys.  A copy command costs more than the 3 KB it moves: the
// copy engine's own dispatch and its hand-over to the compute queue and back (single query, 10M x 768: 0.109 -> see DESIGN §3).
int host_search_wrap_begin(zvec_hip_ctx_s *ctx, const void *queries, size_t qbytes, const uint64_t *exclude,
                           uint64_t nbits, uint32_t count, uint32_t topk, hipStream_t stream) {
  const int zc = ropts().zerocopy.load(std::memory_order_relaxed);      // bit 0: queries read in place, bit 1: results written in place
  ZRET(ctx->io_q.ensure(qbytes));
  ctx->io_qp = ctx->io_q.p;
  if (qbytes <= PIN_LIMIT && ctx->pin_in.ensure(qbytes) == 0) {
    memcpy(ctx->pin_in.p, queries, qbytes);              // (the previous call's transfer has been waited for)
    if ((zc & 1) && ctx->pin_in.dev)
      ctx->io_qp = ctx->pin_in.dev;
    else
      ZCHK(hipMemcpyAsync(ctx->io_q.p, ctx->pin_in.p, qbytes, hipMemcpyHostToDevice, stream));
  } else {
    ZCHK(hipMemcpyAsync(ctx->io_q.p, queries, qbytes, hipMemcpyHostToDevice, stream));
  }
  if (exclude) {
    size_t words = (size_t)((nbits + 63) / 64);
    ZRET(ctx->io_ex.ensure(words * 8 + 8));
    ZCHK(hipMemcpyAsync(ctx->io_ex.p, exclude, words * 8, hipMemcpyHostToDevice, stream));
  }
  // keys | scores | counts in ONE buffer (16-byte aligned parts): a single copy brings a result back — or none at all
  const size_t kb = (size_t)count * topk * sizeof(uint64_t), sb = ((size_t)count * topk * sizeof(float) + 15) & ~(size_t)15,
               cb = (size_t)count * sizeof(uint32_t);
  char *o = nullptr;
  ctx->out_mapped = false;
  if ((zc & 2) && kb + sb + cb <= PIN_LIMIT && ctx->pin_out.ensure(kb + sb + cb) == 0 && ctx->pin_out.dev) {
    o = static_cast<char *>(ctx->pin_out.dev);
    ctx->out_mapped = true;
  } else {
    ZRET(ctx->io_out.ensure(kb + sb + cb));
    o = ctx->io_out.as<char>();
  }
  ctx->io_keys.p = o;
  ctx->io_scores.p = o + kb;
  ctx->io_counts.p = o + kb + sb;
  return 0;
}

int host_search_wrap_end(zvec_hip_ctx_s *ctx, uint32_t count, uint32_t topk, uint64_t *out_keys, float *out_scores,
                         uint32_t *out_counts, hipStream_t stream) {
  const size_t kb = (size_t)count * topk * sizeof(uint64_t), sbytes = (size_t)count * topk * sizeof(float),
               sb = (sbytes + 15) & ~(size_t)15, cb = (size_t)count * sizeof(uint32_t);
  if (ctx->out_mapped) {
    ZRET(host_wait(ctx, stream));
    const char *h = static_cast<const char *>(ctx->pin_out.p);
    memcpy(out_keys, h, kb);
    memcpy(out_scores, h + kb, sbytes);
    memcpy(out_counts, h + kb + sb, cb);
    return 0;
  }
  if (kb + sb + cb <= PIN_LIMIT && ctx->pin_out.ensure(kb + sb + cb) == 0) {
    char *h = static_cast<char *>(ctx->pin_out.p);
    ZCHK(hipMemcpyAsync(h, ctx->io_out.p, kb + sb + cb, hipMemcpyDeviceToHost, stream));
    ZRET(host_wait(ctx, stream));
    memcpy(out_keys, h, kb);
    memcpy(out_scores, h + kb, sbytes);
    memcpy(out_counts, h + kb + sb, cb);
    return 0;
  }
  ZCHK(hipMemcpyAsync(out_keys, ctx->io_keys.p, kb, hipMemcpyDeviceToHost, stream));
  ZCHK(hipMemcpyAsync(out_scores, ctx->io_scores.p, sbytes, hipMemcpyDeviceToHost, stream));
  ZCHK(hipMemcpyAsync(out_counts, ctx->io_counts.p, cb, hipMemcpyDeviceToHost, stream));
  ZRET(host_wait(ctx, stream));
  return 0;
}

}  // namespace
