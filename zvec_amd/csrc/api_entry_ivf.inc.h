// api_entry_ivf.inc.h — C ABI entry points: IVF index — load, dumped-segment load, GPU build, export, search (inside extern "C")
// Part of zvec_hip_api.hip (one translation unit; included in order, not standalone).

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

int zvec_hip_ivf_get_vectors(zvec_hip_ivf_t h, const uint64_t *list_positions, uint64_t n, void *out) {
  if (!h || (n && (!list_positions || !out))) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (!h->loaded) return ZVEC_HIP_ERR_NO_INDEX_LOADED;
  if (n == 0) return 0;
  if (n > 0x7fffffffull) return ZVEC_HIP_ERR_OUT_OF_RANGE;
  std::lock_guard<std::mutex> g(h->mu);
  std::vector<uint64_t> pos(n);
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t lp = list_positions[i];
    if (lp >= h->count_local) return ZVEC_HIP_ERR_NO_EXIST;
    const uint32_t l = (uint32_t)(std::upper_bound(h->h_dense0.begin(), h->h_dense0.end(), lp) - h->h_dense0.begin()) - 1;
    pos[i] = (uint64_t)h->h_tile0[l] * TILE_N + (lp - h->h_dense0[l]);
  }
  ZCHK(hipSetDevice(h->device));
  return store_get_rows(h->defctx, h->lists, pos, out);
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
