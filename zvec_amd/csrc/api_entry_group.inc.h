// api_entry_group.inc.h — C ABI entry points: group-by search on a flat index (inside extern "C")
// Part of zvec_hip_api.hip (one translation unit; included in order, not standalone).
//
// FlatSearcher / FlatStreamer::search_impl with IndexContext::set_group_params + set_group_by
// (flat_searcher.cc:178-207, flat_streamer.cc:323-324,391-483; flat_streamer_context.h:135-180): the `group_topk` best
// documents of each of the `group_num` groups whose best document is closest.  The group of every storage position is
// data here (the plugin sweeps the caller's group_by(key) callback once per key, like the filter sweep).

}  // extern "C"

namespace {

struct GroupOut { uint32_t *groups, *ngroups; uint64_t *keys; float *scores; uint32_t *counts; };   // device arrays of the whole batch

// selection over a candidate matrix of `cnt` queries (rows q0.. of the prepared batch); outputs at row offset q0
int group_select(zvec_hip_ctx_s *c, const Store &st, const float *cs, const uint32_t *ci, uint32_t stride, uint32_t len,
                 uint32_t q0, uint32_t cnt, const uint32_t *d_group_of, uint32_t ngroups, uint32_t gnum, uint32_t gk,
                 float threshold, bool refine, const GroupOut &out, hipStream_t s) {
  const size_t rows = (size_t)cnt * gnum;
  // workspace carve-up (8-byte items first)
  const size_t b_tmpk = rows * 8, b_rowk = rows * gk * 8, b_best = (size_t)cnt * ngroups * 4, b_tmps = rows * 4, b_sel = rows * 4,
               b_nsel = (size_t)cnt * 4, b_rows = rows * gk * 4, b_rowi = rows * gk * 4, b_rowc = rows * 4;
  ZRET(c->grp_ws.ensure(b_tmpk + b_rowk + b_best + b_tmps + b_sel + b_nsel + b_rows + b_rowi + b_rowc + 64));
  char *w = c->grp_ws.as<char>();
  uint64_t *tmpk = reinterpret_cast<uint64_t *>(w); w += b_tmpk;
  uint64_t *rowk = reinterpret_cast<uint64_t *>(w); w += b_rowk;
  uint32_t *best = reinterpret_cast<uint32_t *>(w); w += b_best;
  float *tmps = reinterpret_cast<float *>(w); w += b_tmps;
  uint32_t *sel = reinterpret_cast<uint32_t *>(w); w += b_sel;
  uint32_t *nsel = reinterpret_cast<uint32_t *>(w); w += b_nsel;
  float *rows_s = reinterpret_cast<float *>(w); w += b_rows;
  uint32_t *rows_i = reinterpret_cast<uint32_t *>(w); w += b_rowi;
  uint32_t *rows_c = reinterpret_cast<uint32_t *>(w);
  ZCHK(hipMemsetAsync(best, 0xff, b_best, s));
  const uint32_t splits = std::max<uint32_t>(1u, std::min<uint32_t>((len + 4095) / 4096, std::max<uint32_t>(1u, 2048u / std::max(cnt, 1u))));
  hipLaunchKernelGGL(group_best_kernel, dim3(splits, cnt), dim3(256), 0, s, cs, ci, stride, len, d_group_of, ngroups, best);
  const uint64_t nb = (uint64_t)cnt * ngroups;
  hipLaunchKernelGGL(group_keys_to_scores_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, best, nb);
  MergeArgs m{};
  m.part_s = reinterpret_cast<const float *>(best); m.slots_per_q = 1; m.slot_stride = 1; m.k = gnum; m.slot_len = ngroups;
  m.threshold = FLT_MAX;                                 // groups are ranked before the radius applies (topk_to_group_result)
  m.out_keys = tmpk; m.out_scores = tmps; m.out_idx = sel; m.out_counts = nsel;
  hipLaunchKernelGGL(merge_kernel, dim3(cnt), dim3(64), (size_t)gnum * 12 + 16, s, m);
  const bool l2 = refine && st.metric == ZVEC_HIP_METRIC_L2;
  // one pass per query when the per-wave lists of all its slots fit the LDS (4 waves, else 1), otherwise a wave per slot
  const size_t lds1 = ((size_t)gnum * gk * 8 + (size_t)gnum * 8 + 16);
  const int fill_waves = lds1 * 4 <= 60 * 1024 ? 4 : (lds1 <= 60 * 1024 ? 1 : 0);
  if (fill_waves && ngroups <= 0x7fffffffu) {
    ZRET(c->grp_tab.ensure((size_t)cnt * ngroups * 2));
    uint16_t *tab = c->grp_tab.as<uint16_t>();
    ZCHK(hipMemsetAsync(tab, 0xff, (size_t)cnt * ngroups * 2, s));
    hipLaunchKernelGGL(group_slot_kernel, dim3(cnt), dim3(64), 0, s, sel, nsel, gnum, ngroups, tab);
#define ZVEC_GROUP_FILL_Q(HAS, WV)                                                                                               \
    hipLaunchKernelGGL((group_fill_query_kernel<HAS, WV>), dim3(cnt), dim3(64 * WV), lds1 * WV, s, cs, ci, stride, len, d_group_of,  \
                       ngroups, tab, nsel, gnum, gk, threshold, !l2, st.keys, rowk, rows_s, rows_i, rows_c)
    if (ci) { if (fill_waves == 4) ZVEC_GROUP_FILL_Q(true, 4); else ZVEC_GROUP_FILL_Q(true, 1); }
    else { if (fill_waves == 4) ZVEC_GROUP_FILL_Q(false, 4); else ZVEC_GROUP_FILL_Q(false, 1); }
#undef ZVEC_GROUP_FILL_Q
  } else if (ci)
    hipLaunchKernelGGL(group_fill_kernel<true>, dim3((unsigned)rows), dim3(64), (size_t)gk * 8 + 16, s, cs, ci, stride, len, d_group_of,
                       sel, nsel, gnum, gk, threshold, !l2, st.keys, rowk, rows_s, rows_i, rows_c);
  else
    hipLaunchKernelGGL(group_fill_kernel<false>, dim3((unsigned)rows), dim3(64), (size_t)gk * 8 + 16, s, cs, ci, stride, len, d_group_of,
                       sel, nsel, gnum, gk, threshold, !l2, st.keys, rowk, rows_s, rows_i, rows_c);
  ZCHK(hipGetLastError());
  if (l2) {
    // the dense scores are |q|^2 + |b|^2 - 2 q.b: the documents that made the lists are re-scored directly and re-sorted
    const uint64_t pairs = (uint64_t)rows * gk;
    const float *qp = c->qpad.as<float>() + (size_t)q0 * st.dpad;
    if (st.f16)
      hipLaunchKernelGGL(rescore_l2_kernel<true>, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, s, st.base, qp, st.dpad, rows_i,
                         rows_c, (uint32_t)rows, gk, rows_s, gnum);
    else
      hipLaunchKernelGGL(rescore_l2_kernel<false>, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, s, st.base, qp, st.dpad, rows_i,
                         rows_c, (uint32_t)rows, gk, rows_s, gnum);
    hipLaunchKernelGGL(resort_kernel, dim3((unsigned)rows), dim3(64), (size_t)gk * 16 + 16, s, rowk, rows_s, rows_i, rows_c, gk, threshold);
  }
  hipLaunchKernelGGL(group_emit_kernel, dim3((unsigned)rows), dim3(64), 0, s, sel, nsel, gnum, gk, rowk, rows_s, rows_c,
                     out.groups + (size_t)q0 * gnum, out.ngroups + q0, out.keys + (size_t)q0 * gnum * gk,
                     out.scores + (size_t)q0 * gnum * gk, out.counts + (size_t)q0 * gnum);
  ZCHK(hipGetLastError());
  return 0;
}

int group_args_ok(uint32_t ngroups, uint32_t gnum, uint32_t gk) {
  if (ngroups == 0 || gnum == 0 || gk == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if ((size_t)gnum * 12 + 16 > 60 * 1024 || (size_t)gk * 16 + 16 > 60 * 1024) return ZVEC_HIP_ERR_UNSUPPORTED;
  return 0;
}

int group_outputs(zvec_hip_ctx_s *c, uint32_t count, uint32_t gnum, uint32_t gk, GroupOut *o) {
  const size_t rows = (size_t)count * gnum;
  ZRET(c->grp_out.ensure(rows * gk * 12 + rows * 8 + (size_t)count * 4 + 64));
  char *w = c->grp_out.as<char>();
  o->keys = reinterpret_cast<uint64_t *>(w); w += rows * gk * 8;
  o->scores = reinterpret_cast<float *>(w); w += rows * gk * 4;
  o->groups = reinterpret_cast<uint32_t *>(w); w += rows * 4;
  o->counts = reinterpret_cast<uint32_t *>(w); w += rows * 4;
  o->ngroups = reinterpret_cast<uint32_t *>(w);
  return 0;
}

int group_copy_out(zvec_hip_ctx_s *c, const GroupOut &o, uint32_t count, uint32_t gnum, uint32_t gk, uint32_t *out_groups,
                   uint32_t *out_ngroups, uint64_t *out_keys, float *out_scores, uint32_t *out_counts, hipStream_t s) {
  const size_t rows = (size_t)count * gnum;
  ZCHK(hipMemcpyAsync(out_keys, o.keys, rows * gk * 8, hipMemcpyDeviceToHost, s));
  ZCHK(hipMemcpyAsync(out_scores, o.scores, rows * gk * 4, hipMemcpyDeviceToHost, s));
  ZCHK(hipMemcpyAsync(out_groups, o.groups, rows * 4, hipMemcpyDeviceToHost, s));
  ZCHK(hipMemcpyAsync(out_counts, o.counts, rows * 4, hipMemcpyDeviceToHost, s));
  ZCHK(hipMemcpyAsync(out_ngroups, o.ngroups, (size_t)count * 4, hipMemcpyDeviceToHost, s));
  ZCHK(hipStreamSynchronize(s));
  return 0;
}

}  // namespace

extern "C" {

int zvec_hip_flat_search_grouped(zvec_hip_flat_t h, zvec_hip_ctx_t ctx, const void *queries, uint32_t count,
                                 const uint32_t *group_of_position, uint32_t ngroups, uint32_t group_num, uint32_t group_topk,
                                 float threshold, const uint64_t *exclude_bitset, uint32_t *out_groups, uint32_t *out_ngroups,
                                 uint64_t *out_keys, float *out_scores, uint32_t *out_counts) {
  if (!h || !queries || !group_of_position || !out_groups || !out_ngroups || !out_keys || !out_scores || !out_counts)
    return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (count == 0) return 0;
  ZRET(group_args_ok(ngroups, group_num, group_topk));
  zvec_hip_ctx_s *c = ctx ? ctx : h->defctx;
  std::lock_guard<std::mutex> g(c->mu);
  std::shared_lock<FairSharedMutex> r(h->rw);
  ZCHK(hipSetDevice(h->device));
  hipStream_t s = c->cur;
  const Store &st = h->st;
  ZRET(flat_wait_appends(h, s));
  GroupOut o{};
  ZRET(group_outputs(c, count, group_num, group_topk, &o));
  if (st.n == 0) {                                       // empty index: no groups
    ZCHK(hipMemsetAsync(o.ngroups, 0, (size_t)count * 4, s));
    ZCHK(hipMemsetAsync(o.counts, 0, (size_t)count * group_num * 4, s));
    return group_copy_out(c, o, count, group_num, group_topk, out_groups, out_ngroups, out_keys, out_scores, out_counts, s);
  }
  ZRET(c->grp_of.ensure((size_t)st.n * 4));
  ZCHK(hipMemcpyAsync(c->grp_of.p, group_of_position, (size_t)st.n * 4, hipMemcpyHostToDevice, s));
  ZRET(host_search_wrap_begin(c, queries, (size_t)count * st.row_bytes(), exclude_bitset, st.n, count, 1, s));
  ZRET(prep_queries(c, st, c->io_qp, count, threshold, s));
  const uint64_t ntiles = (st.n + TILE_N - 1) / TILE_N;
  const double row_bytes = (double)ntiles * TILE_N * 4.0;
  // query slices: the score matrix stays <= 1 GiB, and a slice is one grid dimension of group_best_kernel (<= 65535)
  const uint32_t sub = (uint32_t)std::max<double>(1.0, std::min<double>(std::min<double>((double)count, 32768.0), std::floor(1073741824.0 / row_bytes)));
  ZRET(c->part_s.ensure((size_t)(row_bytes * sub)));
  for (uint32_t q0 = 0; q0 < count; q0 += sub) {
    const uint32_t cnt = std::min(sub, count - q0);
    float *dump = nullptr;
    uint32_t stride = 0;
    const uint64_t *d_ex = exclude_bitset ? c->io_ex.as<uint64_t>() : nullptr;
    ZRET(flat_effective_exclude(h, c, d_ex, s, &d_ex));
    ZRET(flat_dense_scores(c, st, q0, cnt, threshold, d_ex, s, &dump, &stride));
    ZRET(group_select(c, st, dump, nullptr, stride, (uint32_t)st.n, q0, cnt, c->grp_of.as<uint32_t>(), ngroups, group_num, group_topk,
                      threshold, true, o, s));
  }
  return group_copy_out(c, o, count, group_num, group_topk, out_groups, out_ngroups, out_keys, out_scores, out_counts, s);
}

// FlatStreamer::group_by_search_p_keys_impl (flat_streamer.cc:439-483): as zvec_hip_flat_search_by_ids, grouped
int zvec_hip_flat_search_grouped_by_ids(zvec_hip_flat_t h, zvec_hip_ctx_t ctx, const void *queries, uint32_t count, const uint32_t *ids,
                                        const uint32_t *offsets, const uint32_t *group_of_position, uint32_t ngroups,
                                        uint32_t group_num, uint32_t group_topk, float threshold, const uint64_t *exclude_bitset,
                                        uint32_t *out_groups, uint32_t *out_ngroups, uint64_t *out_keys, float *out_scores,
                                        uint32_t *out_counts) {
  if (!h || !queries || !ids || !offsets || !group_of_position || !out_groups || !out_ngroups || !out_keys || !out_scores || !out_counts)
    return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (count == 0) return 0;
  ZRET(group_args_ok(ngroups, group_num, group_topk));
  zvec_hip_ctx_s *c = ctx ? ctx : h->defctx;
  std::lock_guard<std::mutex> g(c->mu);
  std::shared_lock<FairSharedMutex> r(h->rw);
  ZCHK(hipSetDevice(h->device));
  hipStream_t s = c->cur;
  const Store &st = h->st;
  ZRET(flat_wait_appends(h, s));
  const uint32_t total = offsets[count];
  uint32_t maxlen = 1;
  for (uint32_t q = 0; q < count; ++q) {
    if (offsets[q + 1] < offsets[q]) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
    maxlen = std::max(maxlen, offsets[q + 1] - offsets[q]);
  }
  std::vector<uint32_t> clean(std::max<uint32_t>(total, 1));
  for (uint32_t i = 0; i < total; ++i) {
    uint32_t id = ids[i];
    bool ok = id < st.n && !h->is_hole(id);
    if (ok && exclude_bitset) ok = ((exclude_bitset[id >> 6] >> (id & 63)) & 1ull) == 0;
    clean[i] = ok ? id : IDX_NONE;
  }
  GroupOut o{};
  ZRET(group_outputs(c, count, group_num, group_topk, &o));
  ZRET(c->grp_of.ensure(std::max<size_t>((size_t)st.n, 1) * 4));
  if (st.n) ZCHK(hipMemcpyAsync(c->grp_of.p, group_of_position, (size_t)st.n * 4, hipMemcpyHostToDevice, s));
  ZRET(host_search_wrap_begin(c, queries, (size_t)count * st.row_bytes(), nullptr, 0, count, 1, s));
  ZRET(prep_queries(c, st, c->io_qp, count, threshold, s));
  ZRET(c->plan.ensure(((size_t)total + count + 8) * sizeof(uint32_t)));
  uint32_t *d_pos = c->plan.as<uint32_t>();
  uint32_t *d_off = d_pos + std::max<uint32_t>(total, 1);
  ZCHK(hipMemcpyAsync(d_pos, clean.data(), (size_t)std::max<uint32_t>(total, 1) * 4, hipMemcpyHostToDevice, s));
  ZCHK(hipMemcpyAsync(d_off, offsets, ((size_t)count + 1) * 4, hipMemcpyHostToDevice, s));
  const uint64_t pairs = (uint64_t)count * maxlen;
  ZRET(c->part_s.ensure(pairs * 4));
  ZRET(c->part_i.ensure(pairs * 4));
  if (st.f16)
    hipLaunchKernelGGL(pkeys_score_kernel<true>, dim3(pkeys_score_blocks(count, maxlen)), dim3(256), 0, s, st.base, c->qpad.as<float>(),
                       st.dpad, st.metric, d_pos, d_off, count, maxlen, c->part_s.as<float>(), c->part_i.as<uint32_t>());
  else
    hipLaunchKernelGGL(pkeys_score_kernel<false>, dim3(pkeys_score_blocks(count, maxlen)), dim3(256), 0, s, st.base, c->qpad.as<float>(),
                       st.dpad, st.metric, d_pos, d_off, count, maxlen, c->part_s.as<float>(), c->part_i.as<uint32_t>());
  ZCHK(hipGetLastError());
  ZCHK(hipStreamSynchronize(s));                         // `clean` goes away
  // the pair scores are already direct distances: no refinement
  for (uint32_t q0 = 0; q0 < count; q0 += 32768)       // (a slice is one grid dimension of group_best_kernel)
    ZRET(group_select(c, st, c->part_s.as<float>() + (size_t)q0 * maxlen, c->part_i.as<uint32_t>() + (size_t)q0 * maxlen, maxlen, maxlen,
                      q0, std::min<uint32_t>(32768, count - q0), c->grp_of.as<uint32_t>(), ngroups, group_num, group_topk, threshold,
                      false, o, s));
  return group_copy_out(c, o, count, group_num, group_topk, out_groups, out_ngroups, out_keys, out_scores, out_counts, s);
}
