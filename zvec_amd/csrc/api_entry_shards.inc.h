// api_entry_shards.inc.h — C ABI entry points: ONE handle that owns G device shards of one index (inside extern "C")
// Part of zvec_hip_api.hip (one translation unit; included in order, not standalone).
//
// zvec is a single-process embedded library, so the drop-in plugin cannot rely on torchrun to use the 8 GPUs of a node:
// this handle does the fan-out / merge of CombinedVectorColumnIndexer::Search (combined_vector_column_indexer.cc:91-232)
// inside the process.  Partition = SURVEY §8(e): a flat index by contiguous row ranges (each append is cut into G
// pieces, key = caller key or global storage position), an IVF index by whole inverted lists (byte-balanced map,
// zvec_hip_ivf_shard_map), centroids replicated.  A search hands the batch to one persistent worker thread per shard
// (own device, own context + stream); every worker searches its shard, writes its candidate lists in the packed layout
// (zvec_hip_packed_bytes) and peer-copies them over xGMI into the gather buffer on the first device, which then merges
// them with the same kernel the RCCL path uses (part order = shard order).  The exchange is 124 KiB per shard at
// 1024 x 10: latency-bound point-to-point copies, no collective needed inside one process.

}  // extern "C"  (the worker pool below is C++)

namespace {

struct ShardRange { uint64_t global0, local0, len; };   // a run of global storage positions held by one shard

struct ShardWorker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int()> job;
  bool has_job = false, done = false, quit = false;
  int rc = 0;
};

}  // namespace

struct zvec_hip_shards_s {
  uint32_t dim = 0;
  int dtype = 0, metric = 0, kind = 0;
  uint32_t G = 0;
  std::vector<int> devices;
  std::vector<zvec_hip_flat_s *> flat;
  std::vector<zvec_hip_ivf_s *> ivf;
  std::vector<zvec_hip_ctx_s *> ctx;              // one context (stream + workspace) per shard
  std::vector<DevBuf> d_q, d_ex, d_packed;        // per shard: staged queries, exclude words, packed candidate lists
  std::vector<DevBuf> d_probe;                    // per shard: the probe lists of the whole batch (dealt coarse pass)
  bool deal_coarse = false;                       // IVF: the coarse pass dealt over the shards (zvec_hip_shards_deal_coarse)
  DevBuf d_gather, d_ok, d_os, d_oc;              // on devices[0]
  std::vector<std::vector<ShardRange>> ranges;    // flat: global position runs of every shard
  struct DirEntry { uint64_t global0, len, local0; uint32_t g; };
  std::vector<DirEntry> dir;                      // flat: all runs ordered by global0 (appends arrive in that order)
  uint64_t total = 0;                             // flat: rows appended so far (global storage positions)
  std::vector<std::unique_ptr<ShardWorker>> workers;
  std::mutex mu;                                  // one search / mutation at a time per handle
  size_t row_bytes = 0;
};

namespace {

void shard_worker_main(ShardWorker *w) {
  std::unique_lock<std::mutex> lk(w->mu);
  for (;;) {
    w->cv.wait(lk, [&] { return w->has_job || w->quit; });
    if (w->quit) return;
    std::function<int()> job = std::move(w->job);
    w->has_job = false;
    lk.unlock();
    const int rc = job();
    lk.lock();
    w->rc = rc;
    w->done = true;
    w->cv.notify_all();
  }
}

// run fn(g) on every shard's worker thread, wait for all, return the first error
int shards_parallel(zvec_hip_shards_s *h, const std::function<int(uint32_t)> &fn) {
  for (uint32_t g = 0; g < h->G; ++g) {
    ShardWorker *w = h->workers[g].get();
    std::lock_guard<std::mutex> lk(w->mu);
    w->job = [&fn, g]() { return fn(g); };
    w->has_job = true;
    w->done = false;
    w->cv.notify_all();
  }
  int rc = 0;
  for (uint32_t g = 0; g < h->G; ++g) {
    ShardWorker *w = h->workers[g].get();
    std::unique_lock<std::mutex> lk(w->mu);
    w->cv.wait(lk, [&] { return w->done; });
    if (rc == 0) rc = w->rc;
  }
  return rc;
}

// copy `len` bits from src (bit offset s0) to dst (bit offset d0); dst bits start zeroed
void copy_bits(const uint64_t *src, uint64_t s0, uint64_t *dst, uint64_t d0, uint64_t len) {
  for (uint64_t i = 0; i < len;) {
    const uint64_t sw = (s0 + i) >> 6, sb = (s0 + i) & 63, dw = (d0 + i) >> 6, db = (d0 + i) & 63;
    const uint64_t take = std::min<uint64_t>(len - i, std::min<uint64_t>(64 - sb, 64 - db));
    const uint64_t mask = take == 64 ? ~0ull : ((1ull << take) - 1);
    dst[dw] |= ((src[sw] >> sb) & mask) << db;
    i += take;
  }
}

// global storage position -> (shard, local position); false when nobody holds it
bool shard_locate(const zvec_hip_shards_s *h, uint64_t pos, uint32_t *g, uint64_t *local) {
  size_t lo = 0, hi = h->dir.size();
  while (lo < hi) {
    const size_t mid = (lo + hi) / 2;
    if (h->dir[mid].global0 + h->dir[mid].len <= pos) lo = mid + 1;
    else hi = mid;
  }
  if (lo == h->dir.size() || pos < h->dir[lo].global0) return false;
  *g = h->dir[lo].g;
  *local = h->dir[lo].local0 + (pos - h->dir[lo].global0);
  return true;
}

}  // namespace

extern "C" {

int zvec_hip_shards_create(uint32_t dim, int dtype, int metric, int kind, const int *devices, uint32_t ndev,
                           zvec_hip_shards_t *out) {
  if (!out || !devices || ndev == 0 || ndev > 64 || dim == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (kind != ZVEC_HIP_SHARDS_FLAT && kind != ZVEC_HIP_SHARDS_IVF) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  zvec_hip_shards_s *h = new (std::nothrow) zvec_hip_shards_s();
  if (!h) return ZVEC_HIP_ERR_NO_MEMORY;
  h->dim = dim; h->dtype = dtype; h->metric = metric; h->kind = kind; h->G = ndev;
  h->devices.assign(devices, devices + ndev);
  h->row_bytes = (size_t)dim * (dtype == ZVEC_HIP_DT_FP16 ? 2 : 4);
  h->d_q.resize(ndev); h->d_ex.resize(ndev); h->d_packed.resize(ndev); h->d_probe.resize(ndev); h->ranges.resize(ndev);
  int rc = 0;
  for (uint32_t g = 0; g < ndev && rc == 0; ++g) {
    zvec_hip_ctx_s *c = nullptr;
    rc = ctx_new(devices[g], &c);
    if (rc != 0) break;
    h->ctx.push_back(c);
    if (kind == ZVEC_HIP_SHARDS_FLAT) {
      zvec_hip_flat_t f = nullptr;
      rc = zvec_hip_flat_create(dim, dtype, metric, devices[g], &f);
      if (rc == 0) h->flat.push_back(f);
    } else {
      zvec_hip_ivf_t v = nullptr;
      rc = zvec_hip_ivf_create(dim, dtype, metric, devices[g], &v);
      if (rc == 0) rc = zvec_hip_ivf_keep_shard(v, g, ndev);
      if (v) h->ivf.push_back(v);
    }
  }
  if (rc == 0) {
    // peer access devices[g] -> devices[0] for the candidate-list copies (a no-op between equal devices)
    for (uint32_t g = 1; g < ndev; ++g) {
      if (devices[g] == devices[0]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, devices[g], devices[0]) == hipSuccess && can) {
        (void)hipSetDevice(devices[g]);
        hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
      }
    }
    for (uint32_t g = 0; g < ndev; ++g) {
      h->workers.emplace_back(new ShardWorker());
      ShardWorker *w = h->workers.back().get();
      w->th = std::thread(shard_worker_main, w);
    }
    *out = h;
    return 0;
  }
  zvec_hip_shards_destroy(h);
  return rc;
}

int zvec_hip_shards_destroy(zvec_hip_shards_t h) {
  if (!h) return 0;
  for (auto &wp : h->workers) {
    { std::lock_guard<std::mutex> lk(wp->mu); wp->quit = true; wp->cv.notify_all(); }
    if (wp->th.joinable()) wp->th.join();
  }
  for (uint32_t g = 0; g < h->G; ++g) {
    if (g < h->devices.size()) (void)hipSetDevice(h->devices[g]);
    if (g < h->d_q.size()) { h->d_q[g].release(); h->d_ex[g].release(); h->d_packed[g].release(); }
  }
  if (!h->devices.empty()) {
    (void)hipSetDevice(h->devices[0]);
    h->d_gather.release(); h->d_ok.release(); h->d_os.release(); h->d_oc.release();
  }
  for (auto f : h->flat) zvec_hip_flat_destroy(f);
  for (auto v : h->ivf) zvec_hip_ivf_destroy(v);
  for (auto c : h->ctx) ctx_free(c);
  delete h;
  return 0;
}

int zvec_hip_shards_count(zvec_hip_shards_t h, uint64_t *total, uint64_t *per_shard) {
  if (!h) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lk(h->mu);
  uint64_t sum = 0;
  for (uint32_t g = 0; g < h->G; ++g) {
    uint64_t n = 0;
    if (h->kind == ZVEC_HIP_SHARDS_FLAT) ZRET(zvec_hip_flat_count(h->flat[g], &n));
    else if (h->ivf[g]->loaded) n = h->ivf[g]->count_local;
    if (per_shard) per_shard[g] = n;
    sum += n;
  }
  if (total) *total = sum;
  return 0;
}

// FlatStreamer add path over G shards: the n rows of this call are cut into G contiguous pieces (row
// ranges; key = caller's key, or the global storage position = local position + range start as
// combined_vector_column_indexer.cc:140-145 rebases block-local ids)
int zvec_hip_shards_flat_append(zvec_hip_shards_t h, const void *vecs, uint64_t n, const uint64_t *keys) {
  if (!h || h->kind != ZVEC_HIP_SHARDS_FLAT || (!vecs && n)) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (n == 0) return 0;
  std::lock_guard<std::mutex> lk(h->mu);
  const uint64_t base0 = h->total;
  int rc = shards_parallel(h, [&](uint32_t g) -> int {
    const uint64_t a = (uint64_t)g * n / h->G, b = (uint64_t)(g + 1) * n / h->G;
    if (a == b) return 0;
    std::vector<uint64_t> k(b - a);
    for (uint64_t i = a; i < b; ++i) k[i - a] = keys ? keys[i] : base0 + i;
    return zvec_hip_flat_append(h->flat[g], static_cast<const char *>(vecs) + (size_t)a * h->row_bytes, b - a, k.data());
  });
  if (rc != 0) return rc;
  for (uint32_t g = 0; g < h->G; ++g) {
    const uint64_t a = (uint64_t)g * n / h->G, b = (uint64_t)(g + 1) * n / h->G;
    if (a == b) continue;
    uint64_t local0 = 0;
    for (const auto &r : h->ranges[g]) local0 += r.len;
    h->ranges[g].push_back(ShardRange{base0 + a, local0, b - a});
    h->dir.push_back({base0 + a, b - a, local0, g});
  }
  h->total += n;
  return 0;
}

// IVFBuilder over G shards from host rows: k-means on the first device, its centroids replicated, the rows labelled
// in G pieces (one per device), then every shard fills the lists the byte-balanced map gives it from a stream of chunks
int zvec_hip_shards_ivf_build(zvec_hip_shards_t h, const void *vecs, uint64_t n, const uint64_t *keys, uint32_t nlist,
                              uint32_t kmeans_iters, uint32_t sample_per_list, uint64_t seed) {
  if (!h || h->kind != ZVEC_HIP_SHARDS_IVF || !vecs || n == 0 || nlist == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (nlist > n) nlist = (uint32_t)n;
  if (sample_per_list == 0) sample_per_list = 256;
  std::lock_guard<std::mutex> lk(h->mu);
  const size_t rb = h->row_bytes;
  const char *rows = static_cast<const char *>(vecs);
  // ---- train on shard 0 (the strided sample of zvec_hip_ivf_build) ----
  const uint64_t S = std::min<uint64_t>(n, (uint64_t)sample_per_list * nlist);
  std::vector<char> cent((size_t)nlist * rb);
  {
    std::vector<char> sample((size_t)S * rb);
    for (uint64_t i = 0; i < S; ++i)
      memcpy(&sample[(size_t)i * rb], rows + (size_t)(((unsigned __int128)i * n) / S) * rb, rb);
    ZCHK(hipSetDevice(h->devices[0]));
    Scoped<char> d_sample;
    ZRET(d_sample.alloc((size_t)S * rb));
    ZCHK(hipMemcpy(d_sample, sample.data(), (size_t)S * rb, hipMemcpyHostToDevice));
    ZRET(zvec_hip_ivf_train_dev(h->ivf[0], d_sample, S, nlist, kmeans_iters, seed, nullptr));
    ZRET(zvec_hip_ivf_get_centroids(h->ivf[0], cent.data(), nullptr));
  }
  // ---- labels: piece g of the rows on device g ----
  std::vector<uint32_t> labels(n);
  const uint64_t CH = 1u << 18;
  ZRET(shards_parallel(h, [&](uint32_t g) -> int {
    ZCHK(hipSetDevice(h->devices[g]));
    if (g != 0) ZRET(zvec_hip_ivf_set_centroids(h->ivf[g], cent.data(), nlist));
    const uint64_t a = (uint64_t)g * n / h->G, b = (uint64_t)(g + 1) * n / h->G;
    Scoped<char> d_rows;
    Scoped<uint32_t> d_lab;
    ZRET(d_rows.alloc((size_t)std::min<uint64_t>(CH, std::max<uint64_t>(b - a, 1)) * rb));
    ZRET(d_lab.alloc(std::min<uint64_t>(CH, std::max<uint64_t>(b - a, 1))));
    for (uint64_t o = a; o < b; o += CH) {
      const uint64_t m = std::min<uint64_t>(CH, b - o);
      ZCHK(hipMemcpy(d_rows, rows + (size_t)o * rb, (size_t)m * rb, hipMemcpyHostToDevice));
      ZRET(zvec_hip_ivf_label_dev(h->ivf[g], d_rows, m, d_lab, nullptr));
      ZCHK(hipMemcpy(&labels[o], d_lab, (size_t)m * 4, hipMemcpyDeviceToHost));
    }
    return 0;
  }));
  std::vector<uint32_t> sizes(nlist, 0);
  for (uint64_t i = 0; i < n; ++i) {
    if (labels[i] >= nlist) labels[i] = 0;
    sizes[labels[i]] += 1;
  }
  // ---- fill: every shard streams the chunks and keeps the rows of the lists it owns ----
  return shards_parallel(h, [&](uint32_t g) -> int {
    ZCHK(hipSetDevice(h->devices[g]));
    ZRET(zvec_hip_ivf_begin_lists(h->ivf[g], sizes.data()));
    Scoped<char> d_rows;
    ZRET(d_rows.alloc((size_t)std::min<uint64_t>(CH, n) * rb));
    for (uint64_t o = 0; o < n; o += CH) {
      const uint64_t m = std::min<uint64_t>(CH, n - o);
      ZCHK(hipMemcpy(d_rows, rows + (size_t)o * rb, (size_t)m * rb, hipMemcpyHostToDevice));
      ZRET(zvec_hip_ivf_add_dev(h->ivf[g], d_rows, m, &labels[o], keys ? keys + o : nullptr, o, nullptr));
    }
    return zvec_hip_ivf_end_lists(h->ivf[g]);
  });
}

// IVFSearcher::load over G shards: same arrays as zvec_hip_ivf_load, every shard keeps its lists
int zvec_hip_shards_ivf_load(zvec_hip_shards_t h, const void *centroids, uint32_t nlist, const uint64_t *list_offsets,
                             const void *vecs, const uint64_t *keys) {
  if (!h || h->kind != ZVEC_HIP_SHARDS_IVF) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lk(h->mu);
  return shards_parallel(h, [&](uint32_t g) -> int { return zvec_hip_ivf_load(h->ivf[g], centroids, nlist, list_offsets, vecs, keys); });
}

// IVFSearcher::load over G shards from the raw segment payloads of a dumped index (see zvec_hip_ivf_load_segments):
// every shard parses the same payloads and keeps the lists the byte-balanced map gives it
int zvec_hip_shards_ivf_load_segments(zvec_hip_shards_t h, const void *inverted_header, uint64_t header_bytes,
                                      const void *inverted_meta, uint64_t meta_bytes, const void *inverted_body,
                                      uint64_t body_bytes, const void *keys, uint64_t keys_bytes, const void *centroids) {
  if (!h || h->kind != ZVEC_HIP_SHARDS_IVF) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lk(h->mu);
  return shards_parallel(h, [&](uint32_t g) -> int {
    return zvec_hip_ivf_load_segments(h->ivf[g], inverted_header, header_bytes, inverted_meta, meta_bytes, inverted_body, body_bytes,
                                      keys, keys_bytes, centroids);
  });
}

// FlatSearcher::load over G shards from the "flat.features" payload (see zvec_hip_flat_load_features): the rows are
// dealt in G contiguous ranges; a column-major payload (full 32-row blocks transposed) is cut at block boundaries so
// that every shard receives whole blocks plus, for the last shard, the row-major remainder
int zvec_hip_shards_flat_load_features(zvec_hip_shards_t h, const void *features, uint64_t bytes, uint64_t count,
                                       int column_major, uint32_t batch_size, const uint64_t *keys) {
  if (!h || h->kind != ZVEC_HIP_SHARDS_FLAT || (!features && count) || batch_size == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (count == 0) return 0;
  if (bytes < count * h->row_bytes) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lk(h->mu);
  const uint64_t base0 = h->total;
  // range boundaries: multiples of batch_size (so that transposed blocks stay whole), the last range takes the rest
  std::vector<uint64_t> cut(h->G + 1, count);
  cut[0] = 0;
  for (uint32_t g = 1; g < h->G; ++g) cut[g] = std::min<uint64_t>(count, ((uint64_t)g * count / h->G) / batch_size * batch_size);
  int rc = shards_parallel(h, [&](uint32_t g) -> int {
    const uint64_t a = cut[g], b = cut[g + 1];
    if (a >= b) return 0;
    std::vector<uint64_t> k(b - a);
    for (uint64_t i = a; i < b; ++i) k[i - a] = keys ? keys[i] : base0 + i;
    return zvec_hip_flat_load_features(h->flat[g], static_cast<const char *>(features) + (size_t)a * h->row_bytes,
                                       (b - a) * h->row_bytes, b - a, column_major, batch_size, k.data());
  });
  if (rc != 0) return rc;
  for (uint32_t g = 0; g < h->G; ++g) {
    const uint64_t a = cut[g], b = cut[g + 1];
    if (a >= b) continue;
    uint64_t local0 = 0;
    for (const auto &r : h->ranges[g]) local0 += r.len;
    h->ranges[g].push_back(ShardRange{base0 + a, local0, b - a});
    h->dir.push_back({base0 + a, b - a, local0, g});
  }
  h->total += count;
  return 0;
}

int zvec_hip_shards_deal_coarse(zvec_hip_shards_t h, int enable) {
  if (!h || h->kind != ZVEC_HIP_SHARDS_IVF) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lk(h->mu);
  h->deal_coarse = enable != 0;
  return 0;
}

// the search of CombinedVectorColumnIndexer::Search over device shards (flat: nprobe / max_scan_count ignored).
// exclude_bitset: 1 bit per GLOBAL storage position (flat: append order; IVF: list-order positions of the whole index).
int zvec_hip_shards_search(zvec_hip_shards_t h, const void *queries, uint32_t count, uint32_t topk, float threshold,
                           uint32_t nprobe, uint32_t max_scan_count, const uint64_t *exclude_bitset, uint64_t *out_keys,
                           float *out_scores, uint32_t *out_counts) {
  if (!h || !queries || !out_keys || !out_scores || !out_counts) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (count == 0) return 0;
  if (topk == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if ((size_t)topk * 12 + 16 > 64 * 1024) return ZVEC_HIP_ERR_UNSUPPORTED;
  const bool is_ivf = h->kind == ZVEC_HIP_SHARDS_IVF;
  std::lock_guard<std::mutex> lk(h->mu);
  if (is_ivf)
    for (auto v : h->ivf)
      if (!v->loaded) return ZVEC_HIP_ERR_NO_INDEX_LOADED;
  const uint64_t pb = zvec_hip_packed_bytes(count, topk);
  const size_t kb = (size_t)count * topk * 8, sb = (size_t)count * topk * 4;
  ZCHK(hipSetDevice(h->devices[0]));
  ZRET(h->d_gather.ensure((size_t)pb * h->G));
  ZRET(h->d_ok.ensure(kb));
  ZRET(h->d_os.ensure(sb));
  ZRET(h->d_oc.ensure((size_t)count * 4));
  char *gather = h->d_gather.as<char>();
  // The coarse pass DEALT over the shards (off by default; see zvec_hip_ivf_coarse_dev): shard g scores queries
  // [g x per, (g + 1) x per) against its replica of the centroids and peer-copies that slice of the probe lists (ids, then counts)
  // into every shard's table; the search proper then plans from the table instead of running the whole coarse pass G times.
  const bool dealt = is_ivf && h->deal_coarse && h->G > 1 && !h->ivf[0]->coarse_sep;
  const uint32_t np = is_ivf ? std::max<uint32_t>(1u, std::min(nprobe, h->ivf[0]->nlist)) : 0;
  const uint32_t per = (count + h->G - 1) / h->G;
  if (dealt) {
    for (uint32_t g = 0; g < h->G; ++g) {          // every table exists before the first peer copy into it
      ZCHK(hipSetDevice(h->devices[g]));
      ZRET(h->d_probe[g].ensure(((size_t)count * np + count) * 4));
      ZRET(h->d_q[g].ensure((size_t)count * h->row_bytes));
    }
    ZRET(shards_parallel(h, [&](uint32_t g) -> int {
      ZCHK(hipSetDevice(h->devices[g]));
      zvec_hip_ctx_s *c = h->ctx[g];
      hipStream_t s = c->own;
      ZCHK(hipMemcpyAsync(h->d_q[g].p, queries, (size_t)count * h->row_bytes, hipMemcpyHostToDevice, s));
      const uint32_t lo = std::min(count, g * per), hi = std::min(count, (g + 1) * per);
      if (hi > lo) {
        uint32_t *idx = h->d_probe[g].as<uint32_t>() + (size_t)lo * np, *cnt = h->d_probe[g].as<uint32_t>() + (size_t)count * np + lo;
        ZRET(zvec_hip_ivf_coarse_dev(h->ivf[g], c, h->d_q[g].as<char>() + (size_t)lo * h->row_bytes, hi - lo, nprobe, idx, cnt, s));
        for (uint32_t o = 0; o < h->G; ++o) {
          if (o == g) continue;
          uint32_t *oidx = h->d_probe[o].as<uint32_t>() + (size_t)lo * np, *ocnt = h->d_probe[o].as<uint32_t>() + (size_t)count * np + lo;
          if (h->devices[o] == h->devices[g]) {
            ZCHK(hipMemcpyAsync(oidx, idx, (size_t)(hi - lo) * np * 4, hipMemcpyDeviceToDevice, s));
            ZCHK(hipMemcpyAsync(ocnt, cnt, (size_t)(hi - lo) * 4, hipMemcpyDeviceToDevice, s));
          } else {
            ZCHK(hipMemcpyPeerAsync(oidx, h->devices[o], idx, h->devices[g], (size_t)(hi - lo) * np * 4, s));
            ZCHK(hipMemcpyPeerAsync(ocnt, h->devices[o], cnt, h->devices[g], (size_t)(hi - lo) * 4, s));
          }
        }
      }
      ZCHK(hipStreamSynchronize(s));            // every slice has landed everywhere before any shard plans
      return 0;
    }));
  }
  ZRET(shards_parallel(h, [&](uint32_t g) -> int {
    ZCHK(hipSetDevice(h->devices[g]));
    zvec_hip_ctx_s *c = h->ctx[g];
    hipStream_t s = c->own;
    ZRET(h->d_q[g].ensure((size_t)count * h->row_bytes));
    ZRET(h->d_packed[g].ensure(pb));
    if (!dealt) ZCHK(hipMemcpyAsync(h->d_q[g].p, queries, (size_t)count * h->row_bytes, hipMemcpyHostToDevice, s));
    // this shard's slice of the global exclude set
    const uint64_t *d_ex = nullptr;
    if (exclude_bitset) {
      uint64_t local_n = 0;
      std::vector<ShardRange> tmp;
      const std::vector<ShardRange> *rs = &h->ranges[g];
      if (is_ivf) {                      // owned lists: global list-order run -> local list-order run
        const zvec_hip_ivf_s *v = h->ivf[g];
        uint64_t gd = 0;
        for (uint32_t l = 0; l < v->nlist; ++l) {
          if (v->h_size[l]) tmp.push_back(ShardRange{gd, v->h_dense0[l], v->h_size[l]});
          gd += v->h_size_global[l];
        }
        rs = &tmp;
        local_n = v->count_local;
      } else {
        for (const auto &r : *rs) local_n += r.len;
      }
      std::vector<uint64_t> words((size_t)((local_n + 63) / 64) + 1, 0);
      for (const auto &r : *rs) copy_bits(exclude_bitset, r.global0, words.data(), r.local0, r.len);
      ZRET(h->d_ex[g].ensure(words.size() * 8));
      ZCHK(hipMemcpyAsync(h->d_ex[g].p, words.data(), words.size() * 8, hipMemcpyHostToDevice, s));
      ZCHK(hipStreamSynchronize(s));     // `words` goes away
      d_ex = h->d_ex[g].as<uint64_t>();
    }
    char *p = h->d_packed[g].as<char>();
    uint64_t *dk = reinterpret_cast<uint64_t *>(p);
    float *ds = reinterpret_cast<float *>(p + kb);
    uint32_t *dc = reinterpret_cast<uint32_t *>(p + kb + sb);
    int rc = dealt ? zvec_hip_ivf_search_probes_dev(h->ivf[g], c, h->d_q[g].p, count, topk, threshold, nprobe, max_scan_count,
                                                    h->d_probe[g].as<uint32_t>(), h->d_probe[g].as<uint32_t>() + (size_t)count * np, d_ex,
                                                    dk, ds, dc, s)
           : is_ivf ? zvec_hip_ivf_search_dev(h->ivf[g], c, h->d_q[g].p, count, topk, threshold, nprobe, max_scan_count, d_ex, dk, ds,
                                              dc, s)
                    : zvec_hip_flat_search_dev(h->flat[g], c, h->d_q[g].p, count, topk, threshold, d_ex, dk, ds, dc, s);
    if (rc != 0) return rc;
    // candidate lists -> the gather buffer on the first device (xGMI peer copy; plain D2D when the devices are equal)
    if (h->devices[g] == h->devices[0]) ZCHK(hipMemcpyAsync(gather + (size_t)g * pb, p, pb, hipMemcpyDeviceToDevice, s));
    else ZCHK(hipMemcpyPeerAsync(gather + (size_t)g * pb, h->devices[0], p, h->devices[g], pb, s));
    ZCHK(hipStreamSynchronize(s));
    return 0;
  }));
  ZCHK(hipSetDevice(h->devices[0]));
  zvec_hip_ctx_s *c0 = h->ctx[0];
  if (h->G == 1) {
    char *p = h->d_packed[0].as<char>();
    ZCHK(hipMemcpyAsync(out_keys, p, kb, hipMemcpyDeviceToHost, c0->own));
    ZCHK(hipMemcpyAsync(out_scores, p + kb, sb, hipMemcpyDeviceToHost, c0->own));
    ZCHK(hipMemcpyAsync(out_counts, p + kb + sb, (size_t)count * 4, hipMemcpyDeviceToHost, c0->own));
    ZCHK(hipStreamSynchronize(c0->own));
    return 0;
  }
  ZRET(zvec_hip_merge_topk_packed_dev(c0, gather, pb, h->G, count, topk, h->d_ok.as<uint64_t>(), h->d_os.as<float>(),
                                      h->d_oc.as<uint32_t>(), c0->own));
  ZCHK(hipMemcpyAsync(out_keys, h->d_ok.p, kb, hipMemcpyDeviceToHost, c0->own));
  ZCHK(hipMemcpyAsync(out_scores, h->d_os.p, sb, hipMemcpyDeviceToHost, c0->own));
  ZCHK(hipMemcpyAsync(out_counts, h->d_oc.p, (size_t)count * 4, hipMemcpyDeviceToHost, c0->own));
  ZCHK(hipStreamSynchronize(c0->own));
  return 0;
}

// FlatStreamer::search_bf_by_p_keys_impl over the shards: ids are GLOBAL storage positions; every shard scores the ones
// it holds (zvec_hip_flat_search_by_ids), the lists meet on the first device and are merged there
int zvec_hip_shards_flat_search_by_ids(zvec_hip_shards_t h, const void *queries, uint32_t count, const uint64_t *ids,
                                       const uint32_t *offsets, uint32_t topk, float threshold, const uint64_t *exclude_bitset,
                                       uint64_t *out_keys, float *out_scores, uint32_t *out_counts) {
  if (!h || h->kind != ZVEC_HIP_SHARDS_FLAT || !queries || !ids || !offsets || !out_keys || !out_scores || !out_counts)
    return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (count == 0) return 0;
  if (topk == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lk(h->mu);
  const uint64_t pb = zvec_hip_packed_bytes(count, topk);
  const size_t kb = (size_t)count * topk * 8, sb = (size_t)count * topk * 4;
  // the id lists of every shard, in local positions; excluded and unknown positions are dropped here
  std::vector<std::vector<uint32_t>> lid(h->G), loff(h->G, std::vector<uint32_t>(count + 1, 0));
  for (uint32_t q = 0; q < count; ++q) {
    if (offsets[q + 1] < offsets[q]) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
    for (uint32_t i = offsets[q]; i < offsets[q + 1]; ++i) {
      const uint64_t pos = ids[i];
      uint32_t g = 0;
      uint64_t local = 0;
      if (!shard_locate(h, pos, &g, &local)) continue;
      if (exclude_bitset && ((exclude_bitset[pos >> 6] >> (pos & 63)) & 1ull)) continue;
      lid[g].push_back((uint32_t)local);
    }
    for (uint32_t g = 0; g < h->G; ++g) loff[g][q + 1] = (uint32_t)lid[g].size();
  }
  std::vector<char> packed((size_t)pb * h->G, 0);
  ZRET(shards_parallel(h, [&](uint32_t g) -> int {
    char *p = packed.data() + (size_t)g * pb;
    if (lid[g].empty()) return 0;                       // counts stay 0
    return zvec_hip_flat_search_by_ids(h->flat[g], h->ctx[g], queries, count, lid[g].data(), loff[g].data(), topk, threshold, nullptr,
                                       reinterpret_cast<uint64_t *>(p), reinterpret_cast<float *>(p + kb),
                                       reinterpret_cast<uint32_t *>(p + kb + sb));
  }));
  ZCHK(hipSetDevice(h->devices[0]));
  zvec_hip_ctx_s *c0 = h->ctx[0];
  ZRET(h->d_gather.ensure((size_t)pb * h->G));
  ZRET(h->d_ok.ensure(kb));
  ZRET(h->d_os.ensure(sb));
  ZRET(h->d_oc.ensure((size_t)count * 4));
  ZCHK(hipMemcpyAsync(h->d_gather.p, packed.data(), packed.size(), hipMemcpyHostToDevice, c0->own));
  ZRET(zvec_hip_merge_topk_packed_dev(c0, h->d_gather.p, pb, h->G, count, topk, h->d_ok.as<uint64_t>(), h->d_os.as<float>(),
                                      h->d_oc.as<uint32_t>(), c0->own));
  ZCHK(hipMemcpyAsync(out_keys, h->d_ok.p, kb, hipMemcpyDeviceToHost, c0->own));
  ZCHK(hipMemcpyAsync(out_scores, h->d_os.p, sb, hipMemcpyDeviceToHost, c0->own));
  ZCHK(hipMemcpyAsync(out_counts, h->d_oc.p, (size_t)count * 4, hipMemcpyDeviceToHost, c0->own));
  ZCHK(hipStreamSynchronize(c0->own));
  return 0;
}

// IndexContext::set_fetch_vector over the shards: rows by GLOBAL storage position, one gather per shard
int zvec_hip_shards_flat_get_vectors(zvec_hip_shards_t h, const uint64_t *positions, uint64_t n, void *out) {
  if (!h || h->kind != ZVEC_HIP_SHARDS_FLAT || (n && (!positions || !out))) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (n == 0) return 0;
  std::lock_guard<std::mutex> lk(h->mu);
  std::vector<std::vector<uint64_t>> local(h->G), where(h->G);
  for (uint64_t i = 0; i < n; ++i) {
    uint32_t g = 0;
    uint64_t l = 0;
    if (!shard_locate(h, positions[i], &g, &l)) return ZVEC_HIP_ERR_NO_EXIST;
    local[g].push_back(l);
    where[g].push_back(i);
  }
  const size_t rb = h->row_bytes;
  return shards_parallel(h, [&](uint32_t g) -> int {
    if (local[g].empty()) return 0;
    std::vector<char> tmp(local[g].size() * rb);
    ZRET(zvec_hip_flat_get_vectors(h->flat[g], local[g].data(), local[g].size(), tmp.data()));
    for (size_t j = 0; j < local[g].size(); ++j) memcpy(static_cast<char *>(out) + (size_t)where[g][j] * rb, &tmp[j * rb], rb);
    return 0;
  });
}
