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

// the centroid store back in the index's own space (after zvec_hip_ivf_set_coarse_space had put it into another one)
static void ivf_leave_coarse_space(zvec_hip_ivf_s *h) {
  if (!h->coarse_sep) return;
  h->cent.release();
  h->cent.n = 0;
  h->cent.configure(h->dim, h->metric, h->dtype);
  h->h_centroids.clear();
  h->coarse_sep = false;
  h->trained = false;
}

static void ivf_drop_shadow(zvec_hip_ivf_s *h) {
  h->shadow.keys = nullptr; h->shadow.extra = nullptr;      // (never its own)
  h->shadow.release();
  h->shadow.n = 0;
  if (h->d_shadow_facts) (void)hipFree(h->d_shadow_facts);
  h->d_shadow_facts = nullptr;
  h->shadow_on = false;
}

static void ivf_release(zvec_hip_ivf_s *h) {
  ivf_drop_shadow(h);
  h->cent.release(); h->lists.release();
  h->cent.n = 0; h->lists.n = 0;
  ivf_leave_coarse_space(h);
  if (h->d_size) (void)hipFree(h->d_size);
  if (h->d_size_global) (void)hipFree(h->d_size_global);
  if (h->d_tile0) (void)hipFree(h->d_tile0);
  if (h->d_order) (void)hipFree(h->d_order);
  if (h->d_tail) (void)hipFree(h->d_tail);
  if (h->d_dense0) (void)hipFree(h->d_dense0);
  h->d_size = h->d_size_global = h->d_tile0 = h->d_order = h->d_tail = nullptr; h->d_dense0 = nullptr;
  h->loaded = false; h->trained = false; h->filling = false;
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

// Greedy largest-first (LPT) assignment of inverted lists to shards, weighted by the 128-row tiles a list occupies in
// HBM (= the bytes a scan streams).  Deterministic: lists ordered by (tiles desc, list id asc), each given to the shard
// with the least tiles so far (lowest shard id on ties) — every rank computes the same map from the global list sizes.
// "whole inverted lists assigned to GPUs (balanced by bytes, list->GPU map)".
static void ivf_lpt_owner(const uint32_t *sizes, uint32_t nlist, uint32_t nshards, uint32_t *owner, uint64_t *rows_out) {
  std::vector<uint64_t> load(nshards, 0), rows(nshards, 0);
  if (nshards <= 1) {
    for (uint32_t l = 0; l < nlist; ++l) { owner[l] = 0; rows[0] += sizes[l]; }
  } else {
    std::vector<uint32_t> order(nlist);
    for (uint32_t l = 0; l < nlist; ++l) order[l] = l;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return sizes[x] > sizes[y]; });
    for (uint32_t i = 0; i < nlist; ++i) {
      const uint32_t l = order[i];
      uint32_t best = 0;
      for (uint32_t g = 1; g < nshards; ++g)
        if (load[g] < load[best]) best = g;
      owner[l] = best;
      load[best] += (sizes[l] + TILE_N - 1) / TILE_N;
      rows[best] += sizes[l];
    }
  }
  if (rows_out) for (uint32_t g = 0; g < nshards; ++g) rows_out[g] = rows[g];
}

int zvec_hip_ivf_shard_map(const uint32_t *list_sizes, uint32_t nlist, uint32_t nshards, uint32_t *owner_out,
                           uint64_t *shard_rows_out) {
  if (!list_sizes || !owner_out || nlist == 0 || nshards == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  ivf_lpt_owner(list_sizes, nlist, nshards, owner_out, shard_rows_out);
  return 0;
}

int zvec_hip_ivf_keep_shard(zvec_hip_ivf_t h, uint32_t shard, uint32_t nshards) {
  if (!h || nshards == 0 || shard >= nshards) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (h->loaded || h->filling) return ZVEC_HIP_ERR_NO_READY;   // must be set before load/build
  h->shard = shard; h->nshards = nshards;
  return 0;
}

int zvec_hip_ivf_list_owners(zvec_hip_ivf_t h, uint32_t *owner_out) {
  if (!h || !owner_out) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (!h->loaded && !h->filling) return ZVEC_HIP_ERR_NO_INDEX_LOADED;
  memcpy(owner_out, h->h_owner.data(), (size_t)h->nlist * 4);
  return 0;
}

// (re)build the centroid store from host rows; the index is "trained" afterwards
static int ivf_set_centroids(zvec_hip_ivf_s *h, const void *h_centroids, uint32_t nlist, hipStream_t s) {
  ivf_leave_coarse_space(h);            // centroids handed over here are rows of the index's own space
  const size_t rb = h->lists.row_bytes();
  h->nlist = nlist;
  if (h_centroids != h->h_centroids.data())
    h->h_centroids.assign(reinterpret_cast<const char *>(h_centroids), reinterpret_cast<const char *>(h_centroids) + (size_t)nlist * rb);
  h->cent.n = 0;
  Scoped<char> d_c;
  ZRET(d_c.alloc((size_t)nlist * rb));
  ZCHK(hipMemcpyAsync(d_c, h->h_centroids.data(), (size_t)nlist * rb, hipMemcpyHostToDevice, s));
  ZRET(store_append_dev(h->cent, d_c, nlist, nullptr, s));
  ZCHK(hipStreamSynchronize(s));
  h->trained = true;
  return 0;
}

// IVFDumper's job, step 1 (ivf_dumper.h:33-160: every list's extent is known before its blocks are written): lay out
// the lists this shard owns from the GLOBAL list sizes, allocate the blocked store, reset the per-list cursors.
static int ivf_begin_lists(zvec_hip_ivf_s *h, const uint32_t *sizes_global, hipStream_t s) {
  const uint32_t nlist = h->nlist;
  h->h_size_global.assign(sizes_global, sizes_global + nlist);
  h->h_owner.assign(nlist, 0);
  ivf_lpt_owner(sizes_global, nlist, h->nshards, h->h_owner.data(), nullptr);
  h->h_size.assign(nlist, 0);
  uint64_t total = 0;
  for (uint32_t l = 0; l < nlist; ++l) {
    total += sizes_global[l];
    if (h->h_owner[l] == h->shard) h->h_size[l] = sizes_global[l];
  }
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
  h->count_global = total;
  h->local_tiles = tiles;
  if (tiles * TILE_N >= 0xffffffffull) return ZVEC_HIP_ERR_OUT_OF_RANGE;
  h->h_cursor.assign(h->h_dense0.begin(), h->h_dense0.end() - 1);
  h->h_row_ids.assign(dense, 0);
  h->lists.n = 0;
  ZRET(h->lists.reserve(std::max<uint64_t>(tiles * TILE_N, 1), s));
  h->lists.n = tiles * TILE_N;
  if (tiles) ZCHK(hipMemsetAsync(h->lists.keys, 0xff, (size_t)tiles * TILE_N * 8, s));   // padding rows: invalid key
  h->filling = true;
  return 0;
}

// step 2: rows (device, row-major [n][dim]) with their labels (host); the rows of owned lists are appended to those
// lists in arrival order (stable, like the reference's per-list append).  `first_row` = global number of d_rows[0];
// keys: host, per row of this call, NULL -> global row number.
static int ivf_add_rows(zvec_hip_ivf_s *h, const void *d_rows, uint64_t n, const uint32_t *labels, const uint64_t *keys,
                        uint64_t first_row, hipStream_t s) {
  if (n == 0) return 0;
  std::vector<uint64_t> src, dst, kv;
  src.reserve(n / h->nshards + 16); dst.reserve(n / h->nshards + 16); kv.reserve(n / h->nshards + 16);
  for (uint64_t i = 0; i < n; ++i) {
    const uint32_t l = labels[i];
    if (l >= h->nlist) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
    if (h->h_owner[l] != h->shard) continue;
    const uint64_t d = h->h_cursor[l];
    if (d >= h->h_dense0[l + 1]) return ZVEC_HIP_ERR_INVALID_ARGUMENT;   // more rows than begin_lists was told
    h->h_cursor[l] = d + 1;
    h->h_row_ids[d] = first_row + i;
    src.push_back(i);
    dst.push_back((uint64_t)h->h_tile0[l] * TILE_N + (d - h->h_dense0[l]));
    kv.push_back(keys ? keys[i] : first_row + i);
  }
  const uint64_t kept = src.size();
  if (kept == 0) return 0;
  Scoped<uint64_t> d_src, d_dst, d_kv;
  ZRET(d_src.alloc(kept));
  ZRET(d_dst.alloc(kept));
  ZRET(d_kv.alloc(kept));
  ZCHK(hipMemcpyAsync(d_src, src.data(), kept * 8, hipMemcpyHostToDevice, s));
  ZCHK(hipMemcpyAsync(d_dst, dst.data(), kept * 8, hipMemcpyHostToDevice, s));
  ZCHK(hipMemcpyAsync(d_kv, kv.data(), kept * 8, hipMemcpyHostToDevice, s));
  ZRET(launch_pack(h->lists, d_rows, kept, d_src, 0, d_dst, s));
  hipLaunchKernelGGL(scatter_keys_kernel, dim3((unsigned)((kept + 255) / 256)), dim3(256), 0, s, h->lists.keys,
                     (const uint64_t *)d_dst, (const uint64_t *)d_kv, kept);
  ZCHK(hipGetLastError());
  ZCHK(hipStreamSynchronize(s));     // the staged arrays are freed on return
  return 0;
}

// step 3: every list must be complete; upload the list tables and the deal order of the scan's work queue
static int ivf_end_lists(zvec_hip_ivf_s *h) {
  const uint32_t nlist = h->nlist;
  for (uint32_t l = 0; l < nlist; ++l)
    if (h->h_cursor[l] != h->h_dense0[l + 1]) return ZVEC_HIP_ERR_NO_READY;   // fewer rows than announced
  const uint64_t tiles = h->local_tiles;
  if (h->d_size) { (void)hipFree(h->d_size); (void)hipFree(h->d_size_global); (void)hipFree(h->d_tile0); (void)hipFree(h->d_dense0); (void)hipFree(h->d_order); (void)hipFree(h->d_tail); h->d_size = nullptr; }
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
    // tail of such items leaves most of the chip idle.  Guided self-scheduling in three levels of the deal order:
    // the first half of the tiles (level 0) in chunks twice as long (fewer top-k warm-ups and list write-outs per byte
    // while everybody is busy anyway), the next quarter (level 1) at the base length, the last quarter (level 2) in
    // chunks a quarter as long.  Measured (head share 0 / 25 / 50 / 75 %): 1/8 shard 0.691 / 0.700 / 0.706 / 0.67 of
    // peak, whole 10M index 0.758 / 0.767 / 0.775 / 0.770 on one box.
    h->h_tail.assign(nlist, 1);
    const uint64_t head_num = knobs().ivf_head_pct;         // per cent of the tiles dealt in double-length chunks
    uint64_t acc = 0;
    for (uint32_t i = nlist; i-- > 0;) {
      const uint32_t l = order[i];
      if (acc * 4 >= tiles) break;
      h->h_tail[l] = 2;
      acc += (h->h_size[l] + TILE_N - 1) / TILE_N;
    }
    acc = 0;
    for (uint32_t i = 0; i < nlist; ++i) {
      const uint32_t l = order[i];
      if (acc * 100 >= tiles * head_num || h->h_tail[l] == 2) break;
      h->h_tail[l] = 0;
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
  h->filling = false;
  // rows of the i largest local lists: the bound of what i probes can scan (small-batch route); computed here, once,
  // because searches on different contexts read it concurrently
  {
    std::vector<uint32_t> sz(h->h_size);
    std::sort(sz.begin(), sz.end(), std::greater<uint32_t>());
    h->h_rows_of_largest.assign((size_t)h->nlist + 1, 0);
    for (uint32_t i = 0; i < h->nlist; ++i) h->h_rows_of_largest[i + 1] = h->h_rows_of_largest[i] + sz[i];
  }
  h->loaded = true;
  return 0;
}

// all rows at once (load, load_segments, the one-call build): centroids + layout + rows + tables
static int ivf_pack(zvec_hip_ivf_s *h, const void *d_rows, uint64_t n, const uint64_t *keys,
                    const std::vector<uint32_t> &labels, const void *h_centroids, uint32_t nlist, hipStream_t s) {
  if (h_centroids) {
    ZRET(ivf_set_centroids(h, h_centroids, nlist, s));
  } else {                               // (load_segments without centroids: zvec_hip_ivf_set_coarse_space brings them)
    h->nlist = nlist;
    h->cent.n = 0;
    h->trained = false;
  }
  std::vector<uint32_t> sizes(nlist, 0);
  for (uint64_t i = 0; i < n; ++i) {
    if (labels[i] >= nlist) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
    sizes[labels[i]] += 1;
  }
  ZRET(ivf_begin_lists(h, sizes.data(), s));
  ZRET(ivf_add_rows(h, d_rows, n, labels.data(), keys, 0, s));
  return ivf_end_lists(h);
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
  if (h->loaded || h->filling || h->coarse_sep) ivf_release(h);
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

// a dumped index that does not parse is reported like the reference's loaders do (LOG_ERROR, ivf_entity.cc:443-570): which check, on stderr
static int load_fail(int check, int code) {
  fprintf(stderr, "zvec_hip_ivf_load_segments: segment check %d failed (error %d)\n", check, code);
  return code;
}

int zvec_hip_ivf_load_segments(zvec_hip_ivf_t h, const void *inverted_header, uint64_t header_bytes,
                               const void *inverted_meta, uint64_t meta_bytes, const void *inverted_body,
                               uint64_t body_bytes, const void *keys, uint64_t keys_bytes, const void *centroids) {
  if (!h || !inverted_header || !inverted_meta) return load_fail(1, ZVEC_HIP_ERR_INVALID_ARGUMENT);
  if (header_bytes < sizeof(RefInvertedIndexHeader) + sizeof(RefIndexMetaHeader)) return load_fail(2, ZVEC_HIP_ERR_INVALID_ARGUMENT);
  RefInvertedIndexHeader hd;
  memcpy(&hd, inverted_header, sizeof(hd));
  RefIndexMetaHeader im;
  memcpy(&im, static_cast<const char *>(inverted_header) + sizeof(hd), sizeof(im));
  // IndexMeta::DataType: DT_FP16 = 1, DT_FP32 = 2 (index_meta.h:31-41); MajorOrder: MO_ROW = 1, MO_COLUMN = 2 (:45-49)
  const int dtype = im.data_type == 1 ? ZVEC_HIP_DT_FP16 : (im.data_type == 2 ? ZVEC_HIP_DT_FP32 : -1);
  if (dtype < 0) return load_fail(3, ZVEC_HIP_ERR_UNSUPPORTED);
  if (dtype != h->dtype || im.dimension != h->dim) return load_fail(4, ZVEC_HIP_ERR_MISMATCH);
  const uint32_t nlist = hd.inverted_list_count, bvc = hd.block_vector_count;
  const uint64_t total = hd.total_vector_count;
  const uint32_t unit = dtype == ZVEC_HIP_DT_FP16 ? 2u : 4u;
  const uint64_t elem = (uint64_t)h->dim * unit;
  if (nlist == 0 || bvc == 0 || meta_bytes < (uint64_t)nlist * sizeof(RefInvertedListMeta)) return load_fail(5, ZVEC_HIP_ERR_INVALID_ARGUMENT);
  if (total && (!inverted_body || !keys || keys_bytes < total * 8)) return load_fail(6, ZVEC_HIP_ERR_INVALID_ARGUMENT);
  const uint64_t block_size = (bvc * elem + 31) / 32 * 32;                   // IVFUtility::AlignedSize
  if (hd.block_size != 0 && hd.block_size != block_size) return load_fail(7, ZVEC_HIP_ERR_INVALID_ARGUMENT);
  const bool column_major = im.major_order == 2;
  std::vector<uint64_t> list_off(nlist), row0(nlist + 1), list_offsets(nlist + 1);
  uint64_t seen = 0;
  for (uint32_t l = 0; l < nlist; ++l) {
    RefInvertedListMeta m;
    memcpy(&m, static_cast<const char *>(inverted_meta) + (size_t)l * sizeof(m), sizeof(m));
    // lists are dumped in id order, back to back; an empty list AFTER the last dumped vector keeps the zeroed meta it was
    // created with (IVFDumper::check_dump_inverted_list only fills the skipped lists up to the next non-empty one,
    // ivf_dumper.cc:284-291), and the reference's reader never looks at the id_offset of an empty list
    if (m.vector_count && m.id_offset != seen) return load_fail(8, ZVEC_HIP_ERR_INVALID_ARGUMENT);
    const uint64_t full = m.vector_count / bvc, rem = m.vector_count % bvc;
    const uint64_t bytes = full * block_size + (rem ? (rem * elem + 31) / 32 * 32 : 0);
    if (m.vector_count && (m.offset > body_bytes || bytes > body_bytes - m.offset)) return load_fail(9, ZVEC_HIP_ERR_INVALID_ARGUMENT);
    list_off[l] = m.offset;
    row0[l] = seen;
    list_offsets[l] = seen;
    seen += m.vector_count;
  }
  row0[nlist] = seen;
  list_offsets[nlist] = seen;
  if (seen != total) return load_fail(10, ZVEC_HIP_ERR_INVALID_ARGUMENT);

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
  if (h->loaded || h->filling || h->coarse_sep) ivf_release(h);
  return ivf_pack(h, d_rows, total, static_cast<const uint64_t *>(keys), labels, centroids, nlist, s);
}

// nearest centroid of every row (IVFBuilder::label, ivf_builder.h:253-274: top-1 of the centroid index): device rows ->
// device labels, in batches through assign_kernel (zvk_assign.hip.h): 128 rows x every centroid per work item, arg-min kept
// in registers — no partial lists, no merge pass
static int ivf_label_rows(zvec_hip_ivf_s *h, zvec_hip_ctx_s *c, const Store &cs, const char *d_rows, uint64_t n,
                          uint32_t *d_labels, hipStream_t s) {
  const size_t rb = cs.row_bytes();
  const uint64_t BATCH = 1u << 18;
  if (cs.n == 0 || cs.n > 0xffffffffull) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  const int cus = device_cus(c);
  for (uint64_t o = 0; o < n; o += BATCH) {
    const uint32_t m = (uint32_t)std::min<uint64_t>(BATCH, n - o);
    ZRET(prep_queries(c, cs, d_rows + (size_t)o * rb, m, FLT_MAX, s));
    AssignArgs a{};
    a.base = cs.base; a.bnorm = cs.bnorm; a.queries = c->qpad.as<float>(); a.qnorm = c->qnorm.as<float>();
    a.dpad = cs.dpad; a.nks = cs.dpad / TILE_K; a.metric = cs.metric; a.nq = m; a.n = (uint32_t)cs.n;
    a.out_label = d_labels + o; a.out_score = nullptr;
    // fp16: the 256 x 256 multi-phase tile when there is a tile's worth of rows and centroids; the 128 x 128 tile otherwise
    const bool big = cs.f16 && ropts().assign256.load(std::memory_order_relaxed) != 0 && m >= 2u * A256_ROWS && cs.n >= 192 && a.nks >= 2;
    ZRET(big ? launch_assign256_f16(a, cus, s) : cs.f16 ? launch_assign<true>(a, cus, s) : launch_assign<false>(a, cus, s));
  }
  ZCHK(hipStreamSynchronize(s));      // callers read the labels with plain copies (the context's stream is non-blocking)
  (void)h;
  return 0;
}

// Lloyd's k-means over `S` device rows (IVFBuilder::train, ivf_builder.cc:212-267; the reference's trainer is
// OptKmeansCluster, opt_kmeans_cluster.cc): seeded choice of nlist distinct rows, `iters` assign / mean rounds, empty
// clusters re-seeded by splitting the largest.  Result -> h->h_centroids + the centroid store (h->trained).
static int ivf_train(zvec_hip_ivf_s *h, zvec_hip_ctx_s *c, const char *d_sample, uint64_t S, uint32_t nlist,
                     uint32_t kmeans_iters, uint64_t seed, hipStream_t s) {
  const uint32_t dim = h->dim;
  const bool f16 = h->lists.f16;
  const size_t rb = h->lists.row_bytes();
  Scoped<uint64_t> d_ids;
  Scoped<char> d_cent;
  ZRET(d_ids.alloc(std::max<uint64_t>(nlist, 1)));
  ZRET(d_cent.alloc((size_t)nlist * rb));
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
  Scoped<uint32_t> d_lab;
  ZRET(d_lab.alloc(S));
  Scoped<uint64_t> d_moff, d_members;
  ZRET(d_moff.alloc((size_t)nlist + 1));
  ZRET(d_members.alloc(S));
  std::vector<uint32_t> lab(S);
  for (uint32_t it = 0; it < kmeans_iters; ++it) {
    cs.n = 0;
    ZRET(store_append_dev(cs, d_cent, nlist, nullptr, s));
    ZRET(ivf_label_rows(h, c, cs, d_sample, S, d_lab, s));
    ZCHK(hipMemcpy(lab.data(), d_lab, S * 4, hipMemcpyDeviceToHost));
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
        for (uint32_t col = 0; col < dim; ++col) {
          if (f16) {
            _Float16 *hp = reinterpret_cast<_Float16 *>(hcb.data());
            float v = (float)hp[(size_t)b * dim + col];
            hp[(size_t)e * dim + col] = (_Float16)(v * (1.0f + 1.0f / 256.0f));
            hp[(size_t)b * dim + col] = (_Float16)(v * (1.0f - 1.0f / 256.0f));
          } else {
            float *hp = reinterpret_cast<float *>(hcb.data());
            float v = hp[(size_t)b * dim + col];
            hp[(size_t)e * dim + col] = v * (1.0f + 1.0f / 1024.0f);
            hp[(size_t)b * dim + col] = v * (1.0f - 1.0f / 1024.0f);
          }
        }
        sizes[e] = sizes[b] / 2;
        sizes[b] -= sizes[e];
      }
      ZCHK(hipMemcpy(d_cent, hcb.data(), hcb.size(), hipMemcpyHostToDevice));
    }
  }
  std::vector<char> hc((size_t)nlist * rb);
  ZCHK(hipMemcpy(hc.data(), d_cent, hc.size(), hipMemcpyDeviceToHost));
  return ivf_set_centroids(h, hc.data(), nlist, s);
}

int zvec_hip_ivf_build_dev(zvec_hip_ivf_t h, const void *d_vecs, uint64_t n, const uint64_t *keys, uint32_t nlist,
                           uint32_t kmeans_iters, uint32_t sample_per_list, uint64_t seed, void *stream) {
  if (!h || !d_vecs || n == 0 || nlist == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (nlist > n) nlist = (uint32_t)n;
  std::lock_guard<std::mutex> g(h->mu);
  ZCHK(hipSetDevice(h->device));
  zvec_hip_ctx_s *c = h->defctx;
  std::lock_guard<std::mutex> gc(c->mu);
  hipStream_t s = pick_stream(c, stream);
  const char *rows = reinterpret_cast<const char *>(d_vecs);
  const size_t rb = h->lists.row_bytes();
  if (sample_per_list == 0) sample_per_list = 256;
  if (h->loaded || h->filling || h->coarse_sep) ivf_release(h);

  // ---- sample (deterministic stride) ----
  uint64_t S = std::min<uint64_t>(n, (uint64_t)sample_per_list * nlist);
  {
    std::vector<uint64_t> sample_ids(S);
    for (uint64_t i = 0; i < S; ++i) sample_ids[i] = (uint64_t)(((unsigned __int128)i * n) / S);
    Scoped<uint64_t> d_ids;
    Scoped<char> d_sample;
    ZRET(d_ids.alloc(S));
    ZRET(d_sample.alloc((size_t)S * rb));
    ZCHK(hipMemcpyAsync(d_ids, sample_ids.data(), S * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)S), dim3(256), 0, s, (const void *)rows, (uint32_t)rb, d_ids, S, (void *)d_sample);
    ZCHK(hipGetLastError());
    ZRET(ivf_train(h, c, d_sample, S, nlist, kmeans_iters, seed, s));
  }
  // ---- label every row with its nearest centroid (ivf_builder.h:253-274), then pack the lists ----
  std::vector<uint32_t> lab(n);
  {
    Scoped<uint32_t> d_lab;
    ZRET(d_lab.alloc(n));
    ZRET(ivf_label_rows(h, c, h->cent, rows, n, d_lab, s));
    ZCHK(hipMemcpy(lab.data(), d_lab, n * 4, hipMemcpyDeviceToHost));
  }
  for (uint64_t i = 0; i < n; ++i) if (lab[i] >= nlist) lab[i] = 0;
  std::vector<uint32_t> sizes(nlist, 0);
  for (uint64_t i = 0; i < n; ++i) sizes[lab[i]] += 1;
  ZRET(ivf_begin_lists(h, sizes.data(), s));
  ZRET(ivf_add_rows(h, rows, n, lab.data(), keys, 0, s));
  return ivf_end_lists(h);
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

// ---- streamed build: train / label / begin_lists / add / end_lists ---------------------------------------------------
// The one-call build needs the whole corpus resident next to the store; a rank of a sharded index (100M x 768 fp16 over
// 8 GPUs: 153.6 GB of raw rows) only ever holds a chunk of raw rows, the labels, and the lists it owns.
int zvec_hip_ivf_train_dev(zvec_hip_ivf_t h, const void *d_sample, uint64_t n_sample, uint32_t nlist, uint32_t kmeans_iters,
                           uint64_t seed, void *stream) {
  if (!h || !d_sample || n_sample == 0 || nlist == 0 || nlist > n_sample) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> g(h->mu);
  ZCHK(hipSetDevice(h->device));
  zvec_hip_ctx_s *c = h->defctx;
  std::lock_guard<std::mutex> gc(c->mu);
  if (h->loaded || h->filling || h->coarse_sep) ivf_release(h);
  return ivf_train(h, c, reinterpret_cast<const char *>(d_sample), n_sample, nlist, kmeans_iters, seed, pick_stream(c, stream));
}

int zvec_hip_ivf_set_centroids(zvec_hip_ivf_t h, const void *centroids, uint32_t nlist) {
  if (!h || !centroids || nlist == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> g(h->mu);
  ZCHK(hipSetDevice(h->device));
  if (h->loaded || h->filling || h->coarse_sep) ivf_release(h);
  return ivf_set_centroids(h, centroids, nlist, h->defctx->own);
}

int zvec_hip_ivf_get_centroids(zvec_hip_ivf_t h, void *centroids, uint32_t *nlist) {
  if (!h) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (!h->trained) return ZVEC_HIP_ERR_NO_TRAINED;
  if (centroids && h->coarse_sep) return ZVEC_HIP_ERR_UNSUPPORTED;      // (rows of another width than the caller's [nlist][dim])
  if (nlist) *nlist = h->nlist;
  if (centroids) memcpy(centroids, h->h_centroids.data(), h->h_centroids.size());
  return 0;
}

int zvec_hip_ivf_label_dev(zvec_hip_ivf_t h, const void *d_rows, uint64_t n, uint32_t *d_labels, void *stream) {
  if (!h || (n && (!d_rows || !d_labels))) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (!h->trained) return ZVEC_HIP_ERR_NO_TRAINED;
  if (h->coarse_sep) return ZVEC_HIP_ERR_UNSUPPORTED;      // the centroid store is not in the rows' space: nothing to label against
  if (n == 0) return 0;
  std::lock_guard<std::mutex> g(h->mu);
  ZCHK(hipSetDevice(h->device));
  zvec_hip_ctx_s *c = h->defctx;
  std::lock_guard<std::mutex> gc(c->mu);
  return ivf_label_rows(h, c, h->cent, reinterpret_cast<const char *>(d_rows), n, d_labels, pick_stream(c, stream));
}

int zvec_hip_ivf_begin_lists(zvec_hip_ivf_t h, const uint32_t *list_sizes) {
  if (!h || !list_sizes) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (!h->trained) return ZVEC_HIP_ERR_NO_TRAINED;
  if (h->coarse_sep) return ZVEC_HIP_ERR_UNSUPPORTED;      // (a streamed fill labels against centroids of the rows' own space)
  std::lock_guard<std::mutex> g(h->mu);
  ZCHK(hipSetDevice(h->device));
  if (h->loaded) {                       // a new fill replaces the lists, the centroids stay
    std::vector<char> keep(h->h_centroids);
    const uint32_t nl = h->nlist;
    ivf_release(h);
    ZRET(ivf_set_centroids(h, keep.data(), nl, h->defctx->own));
  }
  int rc = ivf_begin_lists(h, list_sizes, h->defctx->own);
  if (rc == 0) ZCHK(hipStreamSynchronize(h->defctx->own));
  return rc;
}

int zvec_hip_ivf_add_dev(zvec_hip_ivf_t h, const void *d_rows, uint64_t n, const uint32_t *labels, const uint64_t *keys,
                         uint64_t first_row, void *stream) {
  if (!h || (n && (!d_rows || !labels))) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (!h->filling) return ZVEC_HIP_ERR_NO_READY;
  std::lock_guard<std::mutex> g(h->mu);
  ZCHK(hipSetDevice(h->device));
  return ivf_add_rows(h, d_rows, n, labels, keys, first_row, pick_stream(h->defctx, stream));
}

int zvec_hip_ivf_end_lists(zvec_hip_ivf_t h) {
  if (!h) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (!h->filling) return ZVEC_HIP_ERR_NO_READY;
  std::lock_guard<std::mutex> g(h->mu);
  ZCHK(hipSetDevice(h->device));
  return ivf_end_lists(h);
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
  if (centroids && h->coarse_sep) return ZVEC_HIP_ERR_UNSUPPORTED;      // (rows of another width than the caller's [nlist][dim])
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
  std::lock_guard<std::mutex> gc(h->defctx->mu);     // io_q is the built-in context's staging buffer
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
  std::lock_guard<std::mutex> gc(h->defctx->mu);
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

// the IVF search proper; the caller holds c->mu and has validated the arguments
static int ivf_search_dev_locked(zvec_hip_ivf_s *h, zvec_hip_ctx_s *c, const void *d_queries, uint32_t count, uint32_t topk,
                                 float threshold, uint32_t nprobe, uint32_t max_scan_count, int brute_force,
                                 const uint64_t *d_exclude, uint64_t *d_out_keys, float *d_out_scores, uint32_t *d_out_counts,
                                 hipStream_t s, const void *d_coarse_queries = nullptr, const CoarseSplit &coarse = CoarseSplit()) {
  if (!h->trained && !brute_force) return ZVEC_HIP_ERR_NO_TRAINED;      // (segments loaded, coarse space still to be set)
  const uint32_t np = std::max<uint32_t>(1u, std::min(nprobe, h->nlist));      // (the row stride of the probe lists)
  // 32-bit word offsets into the padded query matrix: very large batches go in slices
  const uint32_t maxq = std::max<uint32_t>(1u, 0x7fffffffu / std::max<uint32_t>(std::max(h->lists.dpad, h->cent.dpad), 1u));
  // (the flags of a shadow search belong to ONE call of the core: a sliced batch reads the fp32 lists)
  struct SkipGuard { zvec_hip_ctx_s *c; bool old; ~SkipGuard() { c->shadow_skip = old; } } guard{c, c->shadow_skip};
  if (count > maxq) c->shadow_skip = true;
  c->sh_count = 0;
  for (uint32_t q0 = 0; q0 < count; q0 += maxq) {
    const uint32_t m = std::min(maxq, count - q0);
    SearchOut out{d_out_keys ? d_out_keys + (size_t)q0 * topk : nullptr, d_out_scores ? d_out_scores + (size_t)q0 * topk : nullptr, nullptr,
                  d_out_counts ? d_out_counts + q0 : nullptr};
    CoarseSplit cs = coarse;
    if (cs.given_idx) { cs.given_idx += (size_t)q0 * np; cs.given_cnt += q0; }
    if (cs.out_idx) { cs.out_idx += (size_t)q0 * np; cs.out_cnt += q0; }
    ZRET(ivf_search_core(h, c, reinterpret_cast<const char *>(d_queries) + (size_t)q0 * h->lists.row_bytes(), m, topk, threshold,
                         nprobe, max_scan_count, brute_force, d_exclude, out, s,
                         d_coarse_queries ? reinterpret_cast<const char *>(d_coarse_queries) + (size_t)q0 * h->cent.row_bytes() : nullptr, cs));
  }
  return 0;
}

// The coarse pass on its own, and a search from given probe lists (see CoarseSplit): what a rank of a sharded index runs when the
// coarse pass is DEALT over the ranks — rank r scores queries [r x Q / G, (r + 1) x Q / G) against the replicated centroids, the
// probe lists (Q x nprobe x 4 bytes: 128 KB at 1024 x 32) are all-gathered, every rank plans from the gathered lists
// (IVFCentroidIndex::search, ivf_centroid_index.cc:273-297, then ivf_searcher.cc:217-247 per shard as
// combined_vector_column_indexer.cc:140-232 does per block).
int zvec_hip_ivf_coarse_dev(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, const void *d_queries, uint32_t count, uint32_t nprobe,
                            uint32_t *d_probe_idx, uint32_t *d_probe_cnt, void *stream) {
  if (!h || !d_queries || !d_probe_idx || !d_probe_cnt) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (!h->trained) return ZVEC_HIP_ERR_NO_TRAINED;
  if (h->coarse_sep) return ZVEC_HIP_ERR_UNSUPPORTED;
  if (count == 0) return 0;
  zvec_hip_ctx_s *c = ctx ? ctx : h->defctx;
  std::lock_guard<std::mutex> g(c->mu);
  ZCHK(hipSetDevice(h->device));
  CoarseSplit cs;
  cs.out_idx = d_probe_idx;
  cs.out_cnt = d_probe_cnt;
  return ivf_search_dev_locked(h, c, d_queries, count, 1, FLT_MAX, nprobe, 0xffffffffu, 0, nullptr, nullptr, nullptr, nullptr,
                               pick_stream(c, stream), nullptr, cs);
}

int zvec_hip_ivf_search_probes_dev(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, const void *d_queries, uint32_t count, uint32_t topk,
                                   float threshold, uint32_t nprobe, uint32_t max_scan_count, const uint32_t *d_probe_idx,
                                   const uint32_t *d_probe_cnt, const uint64_t *d_exclude_bitset, uint64_t *d_out_keys,
                                   float *d_out_scores, uint32_t *d_out_counts, void *stream) {
  if (!h || !d_queries || !d_probe_idx || !d_probe_cnt || !d_out_keys || !d_out_scores || !d_out_counts) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (!h->loaded) return ZVEC_HIP_ERR_NO_INDEX_LOADED;
  if (count == 0) return 0;
  if (topk == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  zvec_hip_ctx_s *c = ctx ? ctx : h->defctx;
  std::lock_guard<std::mutex> g(c->mu);
  ZCHK(hipSetDevice(h->device));
  CoarseSplit cs;
  cs.given_idx = d_probe_idx;
  cs.given_cnt = d_probe_cnt;
  return ivf_search_dev_locked(h, c, d_queries, count, topk, threshold, nprobe, max_scan_count, 0, d_exclude_bitset, d_out_keys,
                               d_out_scores, d_out_counts, pick_stream(c, stream), nullptr, cs);
}

int zvec_hip_ivf_search_dev(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, const void *d_queries, uint32_t count, uint32_t topk,
                            float threshold, uint32_t nprobe, uint32_t max_scan_count, const uint64_t *d_exclude_bitset,
                            uint64_t *d_out_keys, float *d_out_scores, uint32_t *d_out_counts, void *stream) {
  if (!h || !d_queries || !d_out_keys || !d_out_scores || !d_out_counts) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (!h->loaded) return ZVEC_HIP_ERR_NO_INDEX_LOADED;
  if (count == 0) return 0;
  if (topk == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;   // ivf_searcher.cc:197-200
  zvec_hip_ctx_s *c = ctx ? ctx : h->defctx;
  std::lock_guard<std::mutex> g(c->mu);
  ZCHK(hipSetDevice(h->device));
  return ivf_search_dev_locked(h, c, d_queries, count, topk, threshold, nprobe, max_scan_count, 0, d_exclude_bitset,
                               d_out_keys, d_out_scores, d_out_counts, pick_stream(c, stream));
}

// The second half of a search through the shadow lists: waits for it, reads how many queries could not be certified and runs those
// again on the fp32 lists (same probe rule, same exclude set), their results replacing the uncertified ones.  The caller holds c->mu.
static int ivf_shadow_certify_locked(zvec_hip_ivf_s *h, zvec_hip_ctx_s *c, const void *d_queries, uint32_t count, uint32_t topk,
                                     uint32_t nprobe, uint32_t max_scan_count, const uint64_t *d_exclude, uint64_t *d_out_keys,
                                     float *d_out_scores, uint32_t *d_out_counts, hipStream_t s, const void *d_coarse_queries,
                                     uint32_t *rerun_out) {
  if (rerun_out) *rerun_out = 0;
  if (c->sh_count == 0) return 0;                      // the last search on this context did not use the shadow lists
  if (c->sh_count != count) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  c->sh_count = 0;
  uint32_t nflag = 0;
  ZCHK(hipMemcpyAsync(&nflag, c->sh_flags.as<uint32_t>() + count, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  ZCHK(hipStreamSynchronize(s));
  const uint32_t used_kp = c->sh_kp;
  if (c->sh_tier == 0) {
    h->shadow_gov.report(nflag, count);
    if (h->shadow_kp == 0) h->shadow_gov.report_width(nflag, count);
  }
  if (nflag == 0) return 0;
  std::vector<uint32_t> flags(count);
  ZCHK(hipMemcpyAsync(flags.data(), c->sh_flags.p, (size_t)count * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  ZCHK(hipStreamSynchronize(s));
  std::vector<uint32_t> which;
  for (uint32_t q = 0; q < count; ++q) if (flags[q]) which.push_back(q);
  const uint32_t m = (uint32_t)which.size();
  if (m == 0) return 0;
  const size_t rb = h->lists.row_bytes(), crb = h->cent.row_bytes();
  Scoped<char> tq, tcq;
  Scoped<uint64_t> tk;
  Scoped<float> ts;
  Scoped<uint32_t> tc;
  ZRET(tq.alloc((size_t)m * rb));
  ZRET(tk.alloc((size_t)m * topk));
  ZRET(ts.alloc((size_t)m * topk));
  ZRET(tc.alloc(m));
  if (d_coarse_queries) ZRET(tcq.alloc((size_t)m * crb));
  for (uint32_t i = 0; i < m; ++i) {
    ZCHK(hipMemcpyAsync(tq.p + (size_t)i * rb, static_cast<const char *>(d_queries) + (size_t)which[i] * rb, rb, hipMemcpyDeviceToDevice, s));
    if (d_coarse_queries)
      ZCHK(hipMemcpyAsync(tcq.p + (size_t)i * crb, static_cast<const char *>(d_coarse_queries) + (size_t)which[i] * crb, crb, hipMemcpyDeviceToDevice, s));
  }
  // A failed certificate first costs a SECOND half-width pass over the flagged queries alone, at twice the pre-selection (32 .. 64 rows:
  // what separates the k-th row from the rest may simply lie beyond the first k'), and only what that pass cannot certify either is
  // answered by the fp32 lists.  (A handful of flagged queries take the small-batch route, which reads the fp32 lists anyway.)
  uint32_t answered_by_fp32 = m;
  int rc;
  if (c->sh_tier == 0 && used_kp < 64 && !c->shadow_skip) {
    c->sh_tier = 1;
    c->shadow_force_kp = std::min<uint32_t>(64, std::max<uint32_t>(32, 2 * used_kp));      // (wide lists are dear to keep: twice the first pass)
    rc = ivf_search_dev_locked(h, c, tq.p, m, topk, FLT_MAX, nprobe, max_scan_count, 0, d_exclude, tk, ts, tc, s,
                               d_coarse_queries ? tcq.p : nullptr);
    c->shadow_force_kp = 0;
    if (rc == 0 && c->sh_count)
      rc = ivf_shadow_certify_locked(h, c, tq.p, m, topk, nprobe, max_scan_count, d_exclude, tk, ts, tc, s,
                                     d_coarse_queries ? tcq.p : nullptr, &answered_by_fp32);
    c->sh_tier = 0;
  } else {
    const bool old = c->shadow_skip;
    c->shadow_skip = true;
    rc = ivf_search_dev_locked(h, c, tq.p, m, topk, FLT_MAX, nprobe, max_scan_count, 0, d_exclude, tk, ts, tc, s,
                               d_coarse_queries ? tcq.p : nullptr);
    c->shadow_skip = old;
  }
  ZRET(rc);
  for (uint32_t i = 0; i < m; ++i) {
    const size_t o = (size_t)which[i] * topk;
    ZCHK(hipMemcpyAsync(d_out_keys + o, tk.p + (size_t)i * topk, (size_t)topk * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
    ZCHK(hipMemcpyAsync(d_out_scores + o, ts.p + (size_t)i * topk, (size_t)topk * sizeof(float), hipMemcpyDeviceToDevice, s));
    ZCHK(hipMemcpyAsync(d_out_counts + which[i], tc.p + i, sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
  }
  ZCHK(hipStreamSynchronize(s));                       // (the temporaries are freed on return)
  if (rerun_out) *rerun_out = answered_by_fp32;        // queries that ended on the fp32 lists
  return 0;
}

int zvec_hip_ivf_set_shadow(zvec_hip_ivf_t h, int enable, uint32_t preselect) {
  if (!h) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> g(h->mu);
  ZCHK(hipSetDevice(h->device));
  if (!enable) {
    ZCHK(hipDeviceSynchronize());
    ivf_drop_shadow(h);
    return 0;
  }
  if (!h->loaded) return ZVEC_HIP_ERR_NO_INDEX_LOADED;
  if (h->dtype != ZVEC_HIP_DT_FP32 || h->metric == ZVEC_HIP_METRIC_COSINE) return ZVEC_HIP_ERR_UNSUPPORTED;
  if (preselect > 64) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (h->shadow_on) { h->shadow_kp = preselect; h->shadow_gov.reset(); return 0; }
  hipStream_t s = h->defctx->own;
  Store &sh = h->shadow;
  sh = Store();
  sh.configure(h->dim, h->metric, ZVEC_HIP_DT_FP16);
  const uint64_t tiles = (h->lists.n + TILE_N - 1) / TILE_N;
  if (tiles == 0) return ZVEC_HIP_ERR_NO_INDEX_LOADED;
  if (hipMalloc(&sh.base, (size_t)tiles * TILE_N * sh.dpad * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); sh.base = nullptr; return ZVEC_HIP_ERR_NO_MEMORY; }
  if (hipMalloc(&sh.bnorm, (size_t)tiles * TILE_N * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); ivf_drop_shadow(h); return ZVEC_HIP_ERR_NO_MEMORY; }
  if (hipMalloc(&h->d_shadow_facts, sizeof(ShadowFacts)) != hipSuccess) { (void)hipGetLastError(); ivf_drop_shadow(h); return ZVEC_HIP_ERR_NO_MEMORY; }
  sh.cap_tiles = tiles;
  sh.n = tiles * TILE_N;
  ZCHK(hipMemsetAsync(h->d_shadow_facts, 0, sizeof(ShadowFacts), s));
  // every position of the used tiles; the padding rows behind a list's last row become zero rows
  const uint64_t npos = tiles * TILE_N;
  hipLaunchKernelGGL(shadow_rows_kernel, dim3((unsigned)((npos + 3) / 4)), dim3(256), 0, s, h->lists.base, h->lists.dpad, sh.dscan,
                     sh.base, sh.dpad, sh.bnorm, npos, h->d_tile0, h->d_size, h->nlist, 0ull, static_cast<ShadowFacts *>(h->d_shadow_facts));
  ZCHK(hipGetLastError());
  ShadowFacts f{};
  ZCHK(hipMemcpyAsync(&f, h->d_shadow_facts, sizeof(f), hipMemcpyDeviceToHost, s));
  ZCHK(hipStreamSynchronize(s));
  const float mx = __builtin_bit_cast(float, f.max_abs);
  if (!(mx < 65504.f)) { ivf_drop_shadow(h); return ZVEC_HIP_ERR_UNSUPPORTED; }     // rows beyond the half range (or nan)
  h->shadow_max_err = __builtin_bit_cast(float, f.max_err);
  h->shadow_max_norm = __builtin_bit_cast(float, f.max_norm);
  h->shadow_kp = preselect;
  h->shadow_gov.reset();
  h->shadow_on = true;
  return 0;
}

int zvec_hip_ivf_shadow_info(zvec_hip_ivf_t h, int *enabled, uint64_t *bytes, float *max_row_error, float *max_row_norm) {
  if (!h) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> g(h->mu);
  if (enabled) *enabled = h->shadow_on ? 1 : 0;
  if (bytes) *bytes = h->shadow_on ? (uint64_t)h->shadow.cap_tiles * TILE_N * (h->shadow.dpad + 1) * sizeof(float) : 0;
  if (max_row_error) *max_row_error = h->shadow_on ? h->shadow_max_err : 0.f;
  if (max_row_norm) *max_row_norm = h->shadow_on ? h->shadow_max_norm : 0.f;
  return 0;
}

int zvec_hip_ivf_shadow_width(zvec_hip_ivf_t h, uint32_t topk, uint32_t *rows) {
  if (!h || !rows) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  *rows = !h->shadow_on ? 0 : std::min<uint32_t>(64, h->shadow_kp ? h->shadow_kp : h->shadow_gov.kp_auto(topk));
  return 0;
}

int zvec_hip_ivf_shadow_certify(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, const void *d_queries, uint32_t count, uint32_t topk,
                                uint32_t nprobe, uint32_t max_scan_count, const uint64_t *d_exclude_bitset, uint64_t *d_out_keys,
                                float *d_out_scores, uint32_t *d_out_counts, void *stream, uint32_t *rerun) {
  if (!h || !d_queries || !d_out_keys || !d_out_scores || !d_out_counts) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (!h->loaded) return ZVEC_HIP_ERR_NO_INDEX_LOADED;
  zvec_hip_ctx_s *c = ctx ? ctx : h->defctx;
  std::lock_guard<std::mutex> g(c->mu);
  ZCHK(hipSetDevice(h->device));
  return ivf_shadow_certify_locked(h, c, d_queries, count, topk, nprobe, max_scan_count, d_exclude_bitset, d_out_keys, d_out_scores,
                                   d_out_counts, pick_stream(c, stream), nullptr, rerun);
}

static int ivf_search_host_impl(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, const void *queries, uint32_t count, uint32_t topk,
                                float threshold, uint32_t nprobe, uint32_t max_scan_count, int brute_force,
                                const uint64_t *exclude_bitset, uint64_t *out_keys, float *out_scores,
                                uint32_t *out_counts, const void *coarse_queries = nullptr) {
  if (!h || !queries || !out_keys || !out_scores || !out_counts) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (!h->loaded) return ZVEC_HIP_ERR_NO_INDEX_LOADED;
  if (count == 0) return 0;
  if (topk == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  zvec_hip_ctx_s *c = ctx ? ctx : h->defctx;
  // ONE critical section from the upload to the copy-out (see zvec_hip_flat_search)
  std::lock_guard<std::mutex> g(c->mu);
  ZCHK(hipSetDevice(h->device));
  ZRET(host_search_wrap_begin(c, queries, (size_t)count * h->lists.row_bytes(), exclude_bitset, h->count_local, count, topk, c->cur));
  const void *d_cq = nullptr;
  if (coarse_queries && h->coarse_sep) {
    const size_t cb = (size_t)count * h->cent.row_bytes();
    ZRET(c->io_cq.ensure(cb));
    ZCHK(hipMemcpyAsync(c->io_cq.p, coarse_queries, cb, hipMemcpyHostToDevice, c->cur));
    d_cq = c->io_cq.p;
  }
  ZRET(ivf_search_dev_locked(h, c, c->io_qp, count, topk, threshold, nprobe, max_scan_count, brute_force,
                             exclude_bitset ? c->io_ex.as<uint64_t>() : nullptr, c->io_keys.as<uint64_t>(),
                             c->io_scores.as<float>(), c->io_counts.as<uint32_t>(), c->cur, d_cq));
  if (c->sh_count)       // shadow lists: the queries whose result could not be certified are re-run on the fp32 lists
    ZRET(ivf_shadow_certify_locked(h, c, c->io_qp, count, topk, nprobe, max_scan_count, exclude_bitset ? c->io_ex.as<uint64_t>() : nullptr,
                                   c->io_keys.as<uint64_t>(), c->io_scores.as<float>(), c->io_counts.as<uint32_t>(), c->cur, d_cq, nullptr));
  return host_search_wrap_end(c, count, topk, out_keys, out_scores, out_counts, c->cur);
}

// The centroid index in a space of its own.  IVFBuilder trains INNER-PRODUCT indexes through a MipsConverter
// (ivf_builder.cc:552-555): the nested "ivf.centroid" index then holds converted centroids — more dimensions, squared-Euclidean
// metric, a MipsReformer named in its meta — and IVFCentroidIndex::search transforms every query with that reformer before the
// coarse scan (ivf_centroid_index.cc:273-297), while the inverted lists keep the original rows and metric.  This entry installs
// such a centroid store (nlist rows of `coarse_dim` elements of the index's element type, in centroid-id order); searches then
// go through zvec_hip_ivf_search_coarse, which takes the reformed queries beside the original ones.
int zvec_hip_ivf_set_coarse_space(zvec_hip_ivf_t h, uint32_t coarse_dim, int coarse_metric, const void *centroids, uint32_t nlist) {
  if (!h || !centroids || coarse_dim == 0 || nlist == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (coarse_metric != ZVEC_HIP_METRIC_L2 && coarse_metric != ZVEC_HIP_METRIC_IP) return ZVEC_HIP_ERR_UNSUPPORTED;
  if (h->loaded && nlist != h->nlist) return ZVEC_HIP_ERR_MISMATCH;
  std::lock_guard<std::mutex> g(h->mu);
  ZCHK(hipSetDevice(h->device));
  hipStream_t s = h->defctx->own;
  h->cent.release();
  h->cent.n = 0;                          // (centroids of the rows' own space may have been loaded before)
  h->cent.configure(coarse_dim, coarse_metric, h->dtype);
  const size_t rb = h->cent.row_bytes();
  h->h_centroids.assign(static_cast<const char *>(centroids), static_cast<const char *>(centroids) + (size_t)nlist * rb);
  Scoped<char> d_c;
  ZRET(d_c.alloc((size_t)nlist * rb));
  ZCHK(hipMemcpyAsync(d_c, centroids, (size_t)nlist * rb, hipMemcpyHostToDevice, s));
  ZRET(store_append_dev(h->cent, d_c, nlist, nullptr, s));
  ZCHK(hipStreamSynchronize(s));
  h->nlist = nlist;
  h->coarse_sep = true;
  h->trained = true;
  return 0;
}

int zvec_hip_ivf_search_coarse(zvec_hip_ivf_t h, zvec_hip_ctx_t ctx, const void *queries, const void *coarse_queries, uint32_t count,
                               uint32_t topk, float threshold, uint32_t nprobe, uint32_t max_scan_count,
                               const uint64_t *exclude_bitset, uint64_t *out_keys, float *out_scores, uint32_t *out_counts) {
  if (!h || !h->coarse_sep || !coarse_queries) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  return ivf_search_host_impl(h, ctx, queries, count, topk, threshold, nprobe, max_scan_count, 0, exclude_bitset, out_keys,
                              out_scores, out_counts, coarse_queries);
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
