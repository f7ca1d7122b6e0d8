// api_entry_ctx_flat.inc.h — C ABI entry points: library, contexts, flat index (inside extern "C")
// Part of zvec_hip_api.hip (one translation unit; included in order, not standalone).

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

int zvec_hip_set_option(const char *name, int value) {
  if (!name) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (strcmp(name, "wait") == 0) {
    if (value < 0 || value > 2) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
    ropts().wait = value;
    return 0;
  }
  if (strcmp(name, "zerocopy") == 0) {
    if (value < 0 || value > 3) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
    ropts().zerocopy = value;
    return 0;
  }
  if (strcmp(name, "assign256") == 0) { ropts().assign256 = value != 0; return 0; }
  if (strcmp(name, "scan256") == 0) {
    if (value < 0 || value > 2) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
    ropts().scan256 = value;
    return 0;
  }
  return ZVEC_HIP_ERR_UNSUPPORTED;
}
int zvec_hip_get_option(const char *name, int *value) {
  if (!name || !value) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (strcmp(name, "wait") == 0) { *value = ropts().wait; return 0; }
  if (strcmp(name, "zerocopy") == 0) { *value = ropts().zerocopy; return 0; }
  if (strcmp(name, "assign256") == 0) { *value = ropts().assign256; return 0; }
  if (strcmp(name, "scan256") == 0) { *value = ropts().scan256; return 0; }
  return ZVEC_HIP_ERR_UNSUPPORTED;
}

int zvec_hip_host_alloc(uint64_t bytes, void **out) {
  if (!out || bytes == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return ZVEC_HIP_ERR_NO_MEMORY; }
  *out = p;
  return 0;
}
int zvec_hip_host_free(void *p) {
  if (p) (void)hipHostFree(p);
  return 0;
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

int zvec_hip_gate_create(int device, zvec_hip_gate_t *out) {
  if (!out) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  ZCHK(hipSetDevice(device));
  zvec_hip_gate_s *g = new (std::nothrow) zvec_hip_gate_s();
  if (!g) return ZVEC_HIP_ERR_NO_MEMORY;
  g->device = device;
  if (hipEventCreateWithFlags(&g->ev, hipEventDisableTiming) != hipSuccess) { delete g; return ZVEC_HIP_ERR_RUNTIME; }
  *out = g;
  return 0;
}
int zvec_hip_gate_destroy(zvec_hip_gate_t g) {
  if (!g) return 0;
  (void)hipSetDevice(g->device);
  if (g->ev) (void)hipEventDestroy(g->ev);
  delete g;
  return 0;
}
int zvec_hip_ctx_set_gate(zvec_hip_ctx_t ctx, zvec_hip_gate_t gate) {
  if (!ctx || (gate && gate->device != ctx->device)) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> g(ctx->mu);
  ctx->gate = gate;
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

static void flat_drop_shadow(zvec_hip_flat_s *h) {
  if (!h->shadow.base && !h->d_shadow_facts) { h->shadow_on = false; return; }
  h->shadow.keys = nullptr; h->shadow.extra = nullptr;      // (never its own)
  h->shadow.release();
  h->shadow.n = 0;
  if (h->d_shadow_facts) (void)hipFree(h->d_shadow_facts);
  h->d_shadow_facts = nullptr;
  h->shadow_on = false;
}

int zvec_hip_flat_destroy(zvec_hip_flat_t h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  flat_drop_shadow(h);
  h->st.release();
  h->d_holes.release();
  h->ring.release();
  for (uint32_t i = 0; i < zvec_hip_flat_s::RING / zvec_hip_flat_s::RING_GROUP; ++i)
    if (h->ring_used[i]) (void)hipEventDestroy(h->ring_ev[i]);
  if (h->append_ev) (void)hipEventDestroy(h->append_ev);
  ctx_free(h->defctx);
  delete h;
  return 0;
}

int zvec_hip_flat_reserve(zvec_hip_flat_t h, uint64_t capacity) {
  if (!h) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> g(h->mu);
  std::unique_lock<FairSharedMutex> w(h->rw);
  flat_drop_shadow(h);                               // (any mutation: the fp16 twin no longer mirrors the store)
  ZCHK(hipSetDevice(h->device));
  return h->st.reserve(capacity, h->defctx->own);
}

// the device copy of the hole bits has to span the store whenever holes exist (caller holds h->rw exclusively)
static int flat_holes_cover(zvec_hip_flat_s *h, hipStream_t s) {
  if (h->nholes == 0) return 0;
  const uint64_t words = (h->st.n + 63) / 64 + 1;
  h->h_holes.resize(words, 0);
  if (words * 8 <= h->d_holes.cap) return 0;
  ZRET(h->d_holes.ensure(words * 16));
  ZCHK(hipMemsetAsync(h->d_holes.p, 0, h->d_holes.cap, s));
  ZCHK(hipMemcpyAsync(h->d_holes.p, h->h_holes.data(), words * 8, hipMemcpyHostToDevice, s));
  ZCHK(hipStreamSynchronize(s));
  return 0;
}

// a pinned slot for a tiny add: rows at *rows, keys at *keys (device-visible host memory).  Slots are handed out in
// order; the event of a group of RING_GROUP slots is recorded behind the kernel that reads the group's last slot and
// waited for before the group's first slot is written again — by then it is RING - RING_GROUP adds old.
static int flat_ring_slot(zvec_hip_flat_s *h, char **rows, uint64_t **keys, uint32_t *slot) {
  const size_t rb = h->st.row_bytes();
  if (!h->ring.p) {
    h->ring_slot_bytes = (zvec_hip_flat_s::FAST_ROWS * (rb + 8) + 63) & ~(size_t)63;
    ZRET(h->ring.ensure(h->ring_slot_bytes * zvec_hip_flat_s::RING));
  }
  const uint32_t sl = h->ring_next++ % zvec_hip_flat_s::RING;
  const uint32_t grp = sl / zvec_hip_flat_s::RING_GROUP;
  if (sl % zvec_hip_flat_s::RING_GROUP == 0 && h->ring_used[grp]) ZCHK(hipEventSynchronize(h->ring_ev[grp]));
  *rows = static_cast<char *>(h->ring.p) + (size_t)sl * h->ring_slot_bytes;
  *keys = reinterpret_cast<uint64_t *>(*rows + zvec_hip_flat_s::FAST_ROWS * rb);
  *slot = sl;
  return 0;
}
// an asynchronous mutation has been enqueued on `s`: note it for the readers (flat_wait_appends); slot = the ring slot
// its kernels read, or ~0u
static int flat_publish_async(zvec_hip_flat_s *h, uint32_t slot, hipStream_t s) {
  if (slot != ~0u && slot % zvec_hip_flat_s::RING_GROUP == zvec_hip_flat_s::RING_GROUP - 1) {
    const uint32_t grp = slot / zvec_hip_flat_s::RING_GROUP;
    if (!h->ring_used[grp]) ZCHK(hipEventCreateWithFlags(&h->ring_ev[grp], hipEventDisableTiming));
    h->ring_used[grp] = true;
    ZCHK(hipEventRecord(h->ring_ev[grp], s));
  }
  h->append_stream = s;
  h->append_pending = true;
  h->append_dirty = true;
  return 0;
}
// a mutation is about to be enqueued on `s`: it must follow the ones enqueued on another stream (caller holds rw exclusively)
static int flat_order_after_appends(zvec_hip_flat_s *h, hipStream_t s) {
  if (!h->append_pending || h->append_stream == s) return 0;
  if (!h->append_ev) ZCHK(hipEventCreateWithFlags(&h->append_ev, hipEventDisableTiming));
  if (h->append_dirty) { ZCHK(hipEventRecord(h->append_ev, h->append_stream)); h->append_dirty = false; }
  ZCHK(hipStreamWaitEvent(s, h->append_ev, 0));
  return 0;
}
// a reader is about to enqueue work on `s` (caller holds rw shared): see zvec_hip_flat_s::append_ev
static int flat_wait_appends(zvec_hip_flat_s *h, hipStream_t s) {
  if (!h->append_pending || h->append_stream == s) return 0;
  {
    std::lock_guard<std::mutex> g(h->ev_mu);
    if (!h->append_ev) ZCHK(hipEventCreateWithFlags(&h->append_ev, hipEventDisableTiming));
    if (h->append_dirty) { ZCHK(hipEventRecord(h->append_ev, h->append_stream)); h->append_dirty = false; }
  }
  ZCHK(hipStreamWaitEvent(s, h->append_ev, 0));
  return 0;
}

int zvec_hip_flat_append_dev(zvec_hip_flat_t h, const void *d_vecs, uint64_t n, const uint64_t *d_keys, void *stream) {
  if (!h || (!d_vecs && n)) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> g(h->mu);
  std::unique_lock<FairSharedMutex> w(h->rw);
  flat_drop_shadow(h);                               // (any mutation: the fp16 twin no longer mirrors the store)
  ZCHK(hipSetDevice(h->device));
  hipStream_t s = pick_stream(h->defctx, stream);
  ZRET(flat_order_after_appends(h, s));                                    // (an earlier append on another stream)
  int rc = store_append_dev(h->st, d_vecs, n, d_keys, s);
  if (rc == 0) rc = flat_holes_cover(h, s);
  // the row count is published now, the pack kernel is only enqueued on `s`: readers on other streams wait for it
  if (rc == 0 && n) rc = flat_publish_async(h, ~0u, s);
  if (rc == 0 && n && s != h->defctx->own) {
    // the caller's stream may not outlive this call: its event is recorded now, not when a reader first needs it
    if (!h->append_ev) ZCHK(hipEventCreateWithFlags(&h->append_ev, hipEventDisableTiming));
    ZCHK(hipEventRecord(h->append_ev, s));
    h->append_dirty = false;
  }
  return rc;
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
  std::unique_lock<FairSharedMutex> w(h->rw);
  flat_drop_shadow(h);                               // (any mutation: the fp16 twin no longer mirrors the store)
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
  if (rc == 0) rc = flat_holes_cover(h, s);
  ZCHK(hipStreamSynchronize(s));
  return rc;
}

// FlatStreamerEntity's persisted rows (flat_streamer_entity.cc:43-47, flat_streamer_entity.h:287-311): a storage segment is a run
// of blocks of `block_size` bytes, each [block_vector_count x element][block_vector_count x u64 key] ... [DeletionMap 4 B]
// [BlockHeader 12 B]; the rows of a block are contiguous and always row-major (header.column_major is never set,
// flat_streamer_entity.cc:826).  `keep[b]`: bit r set = row r of block b is live (r < header.vector_count, not deleted, key
// valid) — the caller derives it from the 16 tail bytes of each block.  One strided H2D copy brings the row regions of the
// whole run over, one pack launch appends the kept rows (in block / row order = the reference iterator's order,
// flat_streamer_entity.cc:428-460) with the keys read from the run.
int zvec_hip_flat_load_blocks(zvec_hip_flat_t h, const void *blocks, uint64_t bytes, uint64_t nblocks, uint32_t block_size,
                              uint32_t block_vector_count, const uint32_t *keep) {
  if (!h || (nblocks && (!blocks || !keep)) || block_vector_count == 0 || block_vector_count > 32) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (nblocks == 0) return 0;
  const uint64_t elem = h->st.row_bytes();
  const uint64_t rows_bytes = (uint64_t)block_vector_count * elem;
  // a block = rows, keys, then DeletionMap (4 B) + BlockHeader (12 B) (flat_index_format.h:91-126); the run must lie inside `bytes`
  if (rows_bytes + (uint64_t)block_vector_count * 8 + 16 > block_size) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (nblocks > bytes / block_size) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::vector<uint64_t> src, keys;
  const char *p = static_cast<const char *>(blocks);
  for (uint64_t b = 0; b < nblocks; ++b) {
    uint32_t m = keep[b];
    if (block_vector_count < 32) m &= (1u << block_vector_count) - 1u;
    const char *kp = p + b * block_size + rows_bytes;
    for (; m; m &= m - 1) {
      const uint32_t r = (uint32_t)__builtin_ctz(m);
      uint64_t key;
      memcpy(&key, kp + (size_t)r * 8, 8);
      src.push_back(b * block_vector_count + r);
      keys.push_back(key);
    }
  }
  const uint64_t kept = src.size();
  if (kept == 0) return 0;
  std::lock_guard<std::mutex> g(h->mu);
  std::unique_lock<FairSharedMutex> w(h->rw);
  flat_drop_shadow(h);                               // (any mutation: the fp16 twin no longer mirrors the store)
  ZCHK(hipSetDevice(h->device));
  hipStream_t s = h->defctx->own;
  // everything that can refuse comes before the first asynchronous copy out of the local arrays
  if (h->st.n + kept >= 0xfffffff0ull) return ZVEC_HIP_ERR_OUT_OF_RANGE;
  ZRET(h->st.reserve(h->st.n + kept, s));
  Scoped<char> d_rows;
  Scoped<uint64_t> d_src, d_keys;
  ZRET(d_rows.alloc(nblocks * rows_bytes));
  ZRET(d_src.alloc(kept));
  ZRET(d_keys.alloc(kept));
  int rc = 0;
  auto copies = [&]() -> int {
    ZCHK(hipMemcpy2DAsync(d_rows, rows_bytes, blocks, block_size, rows_bytes, nblocks, hipMemcpyHostToDevice, s));
    ZCHK(hipMemcpyAsync(d_src, src.data(), kept * 8, hipMemcpyHostToDevice, s));
    ZCHK(hipMemcpyAsync(d_keys, keys.data(), kept * 8, hipMemcpyHostToDevice, s));
    return 0;
  };
  rc = copies();
  if (rc == 0) rc = launch_pack(h->st, d_rows, kept, d_src, h->st.n, nullptr, s, h->st.keys, d_keys);
  if (rc == 0) {
    h->st.n += kept;
    rc = flat_holes_cover(h, s);
  }
  (void)hipStreamSynchronize(s);          // on every path: the copies read `src` / `keys` / `blocks`, the device temporaries go away
  return rc;
}

int zvec_hip_flat_append(zvec_hip_flat_t h, const void *vecs, uint64_t n, const uint64_t *keys) {
  if (!h || (!vecs && n)) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (n == 0) return 0;
  std::lock_guard<std::mutex> g(h->mu);
  std::unique_lock<FairSharedMutex> w(h->rw);
  flat_drop_shadow(h);                               // (any mutation: the fp16 twin no longer mirrors the store)
  ZCHK(hipSetDevice(h->device));
  hipStream_t s = h->defctx->own;
  const size_t rb = h->st.row_bytes();
  if (n <= zvec_hip_flat_s::FAST_ROWS) {              // a document at a time: pinned ring, nothing to wait for
    char *prow = nullptr;
    uint64_t *pkey = nullptr;
    uint32_t slot = 0;
    ZRET(flat_ring_slot(h, &prow, &pkey, &slot));
    memcpy(prow, vecs, (size_t)n * rb);
    if (keys) memcpy(pkey, keys, (size_t)n * 8);
    ZRET(flat_order_after_appends(h, s));
    int rc = store_append_dev(h->st, prow, n, keys ? pkey : nullptr, s);
    if (rc == 0) rc = flat_holes_cover(h, s);
    if (rc == 0) rc = flat_publish_async(h, slot, s);
    return rc;
  }
  // stage through the device in slices of <= 1 GiB
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
    if (rc == 0) rc = flat_holes_cover(h, s);
    if (rc != 0) { tmp.release(); tk.release(); return rc; }
    ZCHK(hipStreamSynchronize(s));
  }
  tmp.release(); tk.release();
  return 0;
}

// IndexStreamer::add_with_id_impl in bulk: FlatStreamerEntity::add_vector_with_id (flat_streamer_entity.cc:900-990), one row
// after the other — the row of ids[i] lives at storage position ids[i] with key ids[i] (or keys[i], for callers that
// keep their own id -> position map):
//   id == count  appended;   id > count  positions [count, id) are padded with HOLES first (zero rows under kInvalidKey that
//   no search returns);   id < count  the row at that position is overwritten in place (a hole becomes a row).
int zvec_hip_flat_put(zvec_hip_flat_t h, const uint32_t *ids, uint64_t n, const void *vecs, const uint64_t *keys) {
  if (!h || (n && (!ids || !vecs))) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (n == 0) return 0;
  for (uint64_t i = 0; i < n; ++i)
    if (ids[i] >= 0xfffffff0u) return ZVEC_HIP_ERR_OUT_OF_RANGE;
  std::lock_guard<std::mutex> g(h->mu);
  std::unique_lock<FairSharedMutex> w(h->rw);
  flat_drop_shadow(h);                               // (any mutation: the fp16 twin no longer mirrors the store)
  ZCHK(hipSetDevice(h->device));
  hipStream_t s = h->defctx->own;
  Store &st = h->st;
  const size_t rb = st.row_bytes();
  const uint64_t rows_per = std::max<uint64_t>(1, ((uint64_t)1 << 28) / (uint64_t)rb);
  // a document at a time (the product's ingest): the rows are read in place from a pinned ring slot, nothing is allocated
  // and nothing waited for; larger calls are staged through a device buffer
  const bool fast = n <= zvec_hip_flat_s::FAST_ROWS;
  Scoped<char> tmp_buf, zero;
  Scoped<uint64_t> tk_buf;
  char *tmp = nullptr;
  uint64_t *tk = nullptr;
  uint32_t slot = 0;
  if (fast) {
    ZRET(flat_ring_slot(h, &tmp, &tk, &slot));
    memcpy(tmp, vecs, (size_t)n * rb);
    if (keys) memcpy(tk, keys, (size_t)n * 8);
    ZRET(flat_order_after_appends(h, s));
  } else {
    ZRET(tmp_buf.alloc((size_t)std::min<uint64_t>(n, rows_per) * rb));
    if (keys) ZRET(tk_buf.alloc(std::min<uint64_t>(n, rows_per)));
    tmp = tmp_buf;
    tk = tk_buf;
  }
  uint64_t lo = ~0ull, hi = 0;                    // hole words touched
  auto touch = [&](uint64_t pos) { lo = std::min(lo, pos >> 6); hi = std::max(hi, pos >> 6); };
  for (uint64_t o = 0; o < n; o += rows_per) {
    const uint64_t m = std::min(rows_per, n - o);
    if (!fast) {
      ZCHK(hipMemcpyAsync(tmp, static_cast<const char *>(vecs) + (size_t)o * rb, (size_t)m * rb, hipMemcpyHostToDevice, s));
      if (keys) ZCHK(hipMemcpyAsync(tk, keys + o, (size_t)m * 8, hipMemcpyHostToDevice, s));
    }
    for (uint64_t i = 0; i < m;) {
      const uint64_t id = ids[o + i];
      if (id > st.n) {                            // pad the gap with holes
        uint64_t gap = id - st.n;
        const uint64_t zrows = std::min<uint64_t>(gap, rows_per);
        if (!zero.p) {
          ZRET(zero.alloc((size_t)zrows * rb));
          ZCHK(hipMemsetAsync(zero, 0, (size_t)zrows * rb, s));
        }
        while (gap) {
          const uint64_t z = std::min(gap, zrows);
          const uint64_t pos0 = st.n;
          ZRET(store_append_dev(st, zero, z, nullptr, s));
          ZCHK(hipMemsetAsync(st.keys + pos0, 0xff, (size_t)z * 8, s));     // kInvalidKey (flat_index_format.h:29)
          h->h_holes.resize((st.n + 63) / 64 + 1, 0);
          for (uint64_t p = pos0; p < pos0 + z; ++p) { h->h_holes[p >> 6] |= 1ull << (p & 63); touch(p); }
          h->nholes += z;
          gap -= z;
        }
      }
      if (id == st.n) {                           // a run of consecutive new ids is one append (key = position)
        uint64_t j = i + 1;
        while (j < m && ids[o + j] == ids[o + j - 1] + 1u) ++j;
        ZRET(store_append_dev(st, tmp + (size_t)i * rb, j - i, keys ? tk + i : nullptr, s));
        i = j;
      } else {                                    // overwrite in place
        ZRET(launch_pack(st, tmp + (size_t)i * rb, 1, nullptr, id, nullptr, s, st.keys, keys ? tk + i : nullptr));
        if (h->is_hole(id)) { h->h_holes[id >> 6] &= ~(1ull << (id & 63)); h->nholes -= 1; touch(id); }
        ++i;
      }
    }
    if (!fast) ZCHK(hipStreamSynchronize(s));
  }
  if (fast) ZRET(flat_publish_async(h, slot, s));
  // device copy of the hole bits: everything after a reallocation, else the words that changed
  const uint64_t words = (st.n + 63) / 64 + 1;
  h->h_holes.resize(words, 0);
  if (h->nholes || h->d_holes.p) {
    if (words * 8 > h->d_holes.cap) {
      ZRET(h->d_holes.ensure(words * 16));
      ZCHK(hipMemsetAsync(h->d_holes.p, 0, h->d_holes.cap, s));
      lo = 0; hi = words - 1;
    }
    if (lo <= hi && lo != ~0ull) {
      ZCHK(hipMemcpyAsync(h->d_holes.as<uint64_t>() + lo, h->h_holes.data() + lo, (size_t)(std::min(hi, words - 1) - lo + 1) * 8, hipMemcpyHostToDevice, s));
      ZCHK(hipStreamSynchronize(s));             // (h_holes is pageable and may be resized by the next call)
      if (fast) ZRET(flat_publish_async(h, slot, s));
    }
  }
  return 0;
}

int zvec_hip_flat_holes(zvec_hip_flat_t h, uint64_t *count) {
  if (!h || !count) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  *count = h->nholes;
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
  zvec_hip_ctx_s *c = h->defctx;
  std::lock_guard<std::mutex> gc(c->mu);             // io_q is the built-in context's staging buffer
  std::shared_lock<FairSharedMutex> r(h->rw);
  if (pos >= h->st.n) return ZVEC_HIP_ERR_NO_EXIST;
  ZCHK(hipSetDevice(h->device));
  ZRET(flat_wait_appends(h, c->own));
  ZRET(c->io_q.ensure(h->st.row_bytes()));
  ZRET(launch_unpack(h->st, pos, c->io_q.p, c->own));
  ZCHK(hipMemcpyAsync(out, c->io_q.p, h->st.row_bytes(), hipMemcpyDeviceToHost, c->own));
  ZCHK(hipStreamSynchronize(c->own));
  return 0;
}

int zvec_hip_flat_get_vectors(zvec_hip_flat_t h, const uint64_t *positions, uint64_t n, void *out) {
  if (!h || (n && (!positions || !out))) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (n == 0) return 0;
  if (n > 0x7fffffffull) return ZVEC_HIP_ERR_OUT_OF_RANGE;
  std::lock_guard<std::mutex> g(h->mu);
  std::lock_guard<std::mutex> gc(h->defctx->mu);
  std::shared_lock<FairSharedMutex> r(h->rw);
  std::vector<uint64_t> pos(positions, positions + n);
  for (uint64_t p : pos)
    if (p >= h->st.n) return ZVEC_HIP_ERR_NO_EXIST;
  ZCHK(hipSetDevice(h->device));
  ZRET(flat_wait_appends(h, h->defctx->own));
  return store_get_rows(h->defctx, h->st, pos, out);
}

// the exclude set a scan of this store has to use: the caller's, plus the holes add-with-id left (caller holds h->rw)
static int flat_effective_exclude(zvec_hip_flat_s *h, zvec_hip_ctx_s *c, const uint64_t *d_exclude, hipStream_t s, const uint64_t **out) {
  *out = d_exclude;
  if (h->nholes == 0) return 0;
  if (!d_exclude) { *out = h->d_holes.as<uint64_t>(); return 0; }
  const uint64_t words = (h->st.n + 63) / 64;
  ZRET(c->holes_ex.ensure(words * 8 + 8));
  hipLaunchKernelGGL(or_bits_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, c->holes_ex.as<uint64_t>(), d_exclude,
                     h->d_holes.as<uint64_t>(), words);
  ZCHK(hipGetLastError());
  *out = c->holes_ex.as<uint64_t>();
  return 0;
}

// the flat search proper; the caller holds c->mu and h->rw (shared) and has validated the arguments
static int flat_search_dev_locked(zvec_hip_flat_s *h, zvec_hip_ctx_s *c, const void *d_queries, uint32_t count, uint32_t topk,
                                  float threshold, const uint64_t *d_exclude_bitset, uint64_t *d_out_keys, float *d_out_scores,
                                  uint32_t *d_out_counts, hipStream_t s) {
  // rows appended through the asynchronous device-pointer form may still be in flight on another stream
  ZRET(flat_wait_appends(h, s));
  ZRET(flat_effective_exclude(h, c, d_exclude_bitset, s, &d_exclude_bitset));
  // the kernels address the padded query matrix with 32-bit word offsets: very large batches go in slices
  const uint32_t maxq = std::max<uint32_t>(1u, 0x7fffffffu / std::max<uint32_t>(h->st.dpad, 1u));
  c->sh_count = 0;
  // Half-width pre-selection (zvk_shadow.hip.h, as for the IVF lists): the scan streams the fp16 twin of the store for k' rows per
  // query — half the bytes of a small batch's HBM-bound scan, and fp16 matrix work instead of fp32 for a wide one —, those are
  // re-scored on the fp32 rows, the k best certified; zvec_hip_flat_shadow_certify re-runs what fails.
  if (h->shadow_on && h->shadow.base && h->shadow.n == h->st.n && !c->shadow_skip && !(threshold < FLT_MAX) && topk <= 32 &&
      count <= maxq && h->st.n > 0 && h->shadow_gov.allow()) {
    uint32_t kp = c->shadow_force_kp ? c->shadow_force_kp : h->shadow_kp ? h->shadow_kp : h->shadow_gov.kp_auto(topk);
    kp = std::min<uint32_t>(kp, 64);
    if (kp > topk) {
      ZRET(prep_queries(c, h->st, d_queries, count, threshold, s));            // fp32 rows for the re-scoring; resets the shared bounds
      const Store &sst = h->shadow;
      const size_t ck = (size_t)count * kp;
      ZRET(c->sh_q16.ensure((size_t)count * sst.dpad * sizeof(float)));
      ZRET(c->sh_qn16.ensure((size_t)count * sizeof(float)));
      ZRET(c->sh_qinfo.ensure((size_t)count * sizeof(f32x2)));
      ZRET(c->sh_keys.ensure(ck * sizeof(uint64_t)));
      ZRET(c->sh_scores.ensure(ck * sizeof(float)));
      ZRET(c->sh_true.ensure(ck * sizeof(float)));
      ZRET(c->sh_idx.ensure(ck * sizeof(uint32_t)));
      ZRET(c->sh_counts.ensure((size_t)count * sizeof(uint32_t)));
      ZRET(c->sh_flags.ensure(((size_t)count + 4) * sizeof(uint32_t)));
      hipLaunchKernelGGL(shadow_prep_queries_kernel, dim3((count + 3) / 4), dim3(256), 0, s, reinterpret_cast<const float *>(d_queries),
                         count, h->st.dim_in, sst.dscan, sst.dpad, c->sh_q16.as<float>(), c->sh_qn16.as<float>(), c->sh_qinfo.as<f32x2>());
      ZCHK(hipGetLastError());
      Store view = sst;                          // the shadow rows under the store's keys (a view: owns nothing)
      view.keys = h->st.keys;
      SearchOut so{c->sh_keys.as<uint64_t>(), c->sh_scores.as<float>(), c->sh_idx.as<uint32_t>(), c->sh_counts.as<uint32_t>()};
      std::swap(c->qpad, c->sh_q16);             // the scan reads the context's prepared queries: the fp16 ones for this call
      std::swap(c->qnorm, c->sh_qn16);
      c->shadow_scan = true;
      const int rc = flat_scan_prepared(c, view, count, kp, FLT_MAX, d_exclude_bitset, so, s, false);
      c->shadow_scan = false;
      std::swap(c->qpad, c->sh_q16);
      std::swap(c->qnorm, c->sh_qn16);
      ZRET(rc);
      hipLaunchKernelGGL(shadow_rescore_kernel<false>, dim3((unsigned)((ck + 3) / 4)), dim3(256), 0, s, h->st.base, c->qpad.as<float>(),
                         h->st.dpad, h->st.metric, so.idx, so.counts, count, kp, c->sh_true.as<float>());
      ZCHK(hipMemsetAsync(c->sh_flags.as<uint32_t>() + count, 0, sizeof(uint32_t), s));
      ShadowSelectArgs sa{};
      sa.c_keys = so.keys; sa.c_shadow = so.scores; sa.c_true = c->sh_true.as<float>(); sa.c_idx = so.idx; sa.c_counts = so.counts;
      sa.qinfo = c->sh_qinfo.as<f32x2>(); sa.facts = static_cast<const ShadowFacts *>(h->d_shadow_facts);
      sa.kp = kp; sa.k = topk; sa.dscan = h->st.dscan; sa.metric = h->st.metric;
      sa.out_keys = d_out_keys; sa.out_scores = d_out_scores; sa.out_idx = nullptr; sa.out_counts = d_out_counts;
      sa.flags = c->sh_flags.as<uint32_t>(); sa.nflag = sa.flags + count;
      hipLaunchKernelGGL(shadow_select_kernel, dim3(count), dim3(64), 0, s, sa);
      ZCHK(hipGetLastError());
      c->sh_count = count;
      c->sh_kp = kp;
      return 0;
    }
  }
  for (uint32_t q0 = 0; q0 < count; q0 += maxq) {
    const uint32_t m = std::min(maxq, count - q0);
    ZRET(prep_queries(c, h->st, reinterpret_cast<const char *>(d_queries) + (size_t)q0 * h->st.row_bytes(), m, threshold, s));
    SearchOut out{d_out_keys + (size_t)q0 * topk, d_out_scores + (size_t)q0 * topk, nullptr, d_out_counts + q0};
    ZRET(flat_scan_prepared(c, h->st, m, topk, threshold, d_exclude_bitset, out, s, true));
  }
  return 0;
}

// the second half of a search through the shadow rows (as ivf_shadow_certify_locked): the caller holds c->mu and h->rw (shared)
static int flat_shadow_certify_locked(zvec_hip_flat_s *h, zvec_hip_ctx_s *c, const void *d_queries, uint32_t count, uint32_t topk,
                                      const uint64_t *d_exclude, uint64_t *d_out_keys, float *d_out_scores, uint32_t *d_out_counts,
                                      hipStream_t s, uint32_t *rerun_out) {
  if (rerun_out) *rerun_out = 0;
  if (c->sh_count == 0) return 0;
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
  const size_t rb = h->st.row_bytes();
  Scoped<char> tq;
  Scoped<uint64_t> tk;
  Scoped<float> ts;
  Scoped<uint32_t> tc;
  ZRET(tq.alloc((size_t)m * rb));
  ZRET(tk.alloc((size_t)m * topk));
  ZRET(ts.alloc((size_t)m * topk));
  ZRET(tc.alloc(m));
  for (uint32_t i = 0; i < m; ++i)
    ZCHK(hipMemcpyAsync(tq.p + (size_t)i * rb, static_cast<const char *>(d_queries) + (size_t)which[i] * rb, rb, hipMemcpyDeviceToDevice, s));
  // (as ivf_shadow_certify_locked: a second half-width pass over the flagged queries at the widest pre-selection, then the fp32 rows)
  uint32_t answered_by_fp32 = m;
  int rc;
  if (c->sh_tier == 0 && used_kp < 64 && !c->shadow_skip) {
    c->sh_tier = 1;
    c->shadow_force_kp = std::min<uint32_t>(64, std::max<uint32_t>(32, 2 * used_kp));      // (wide lists are dear to keep: twice the first pass)
    rc = flat_search_dev_locked(h, c, tq.p, m, topk, FLT_MAX, d_exclude, tk, ts, tc, s);
    c->shadow_force_kp = 0;
    if (rc == 0 && c->sh_count)
      rc = flat_shadow_certify_locked(h, c, tq.p, m, topk, d_exclude, tk, ts, tc, s, &answered_by_fp32);
    c->sh_tier = 0;
  } else {
    const bool old = c->shadow_skip;
    c->shadow_skip = true;
    rc = flat_search_dev_locked(h, c, tq.p, m, topk, FLT_MAX, d_exclude, tk, ts, tc, s);
    c->shadow_skip = old;
  }
  ZRET(rc);
  for (uint32_t i = 0; i < m; ++i) {
    const size_t o = (size_t)which[i] * topk;
    ZCHK(hipMemcpyAsync(d_out_keys + o, tk.p + (size_t)i * topk, (size_t)topk * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
    ZCHK(hipMemcpyAsync(d_out_scores + o, ts.p + (size_t)i * topk, (size_t)topk * sizeof(float), hipMemcpyDeviceToDevice, s));
    ZCHK(hipMemcpyAsync(d_out_counts + which[i], tc.p + i, sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
  }
  ZCHK(hipStreamSynchronize(s));
  if (rerun_out) *rerun_out = answered_by_fp32;        // queries that ended on the fp32 rows
  return 0;
}

int zvec_hip_flat_set_shadow(zvec_hip_flat_t h, int enable, uint32_t preselect) {
  if (!h) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> g(h->mu);
  std::unique_lock<FairSharedMutex> w_(h->rw);
  ZCHK(hipSetDevice(h->device));
  ZCHK(hipDeviceSynchronize());
  flat_drop_shadow(h);
  if (!enable) return 0;
  if (h->dtype != ZVEC_HIP_DT_FP32 || h->st.metric == ZVEC_HIP_METRIC_COSINE) return ZVEC_HIP_ERR_UNSUPPORTED;
  if (preselect > 64) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (h->st.n == 0) return ZVEC_HIP_ERR_NO_INDEX_LOADED;
  hipStream_t s = h->defctx->own;
  ZRET(flat_wait_appends(h, s));
  Store &sh = h->shadow;
  sh = Store();
  sh.configure(h->st.dim_in, h->st.metric, ZVEC_HIP_DT_FP16);
  const uint64_t tiles = (h->st.n + TILE_N - 1) / TILE_N;
  if (hipMalloc(&sh.base, (size_t)tiles * TILE_N * sh.dpad * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); sh.base = nullptr; return ZVEC_HIP_ERR_NO_MEMORY; }
  if (hipMalloc(&sh.bnorm, (size_t)tiles * TILE_N * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); flat_drop_shadow(h); return ZVEC_HIP_ERR_NO_MEMORY; }
  if (hipMalloc(&h->d_shadow_facts, sizeof(ShadowFacts)) != hipSuccess) { (void)hipGetLastError(); flat_drop_shadow(h); return ZVEC_HIP_ERR_NO_MEMORY; }
  sh.cap_tiles = tiles;
  sh.n = h->st.n;
  ZCHK(hipMemsetAsync(h->d_shadow_facts, 0, sizeof(ShadowFacts), s));
  const uint64_t npos = tiles * TILE_N;
  hipLaunchKernelGGL(shadow_rows_kernel, dim3((unsigned)((npos + 3) / 4)), dim3(256), 0, s, h->st.base, h->st.dpad, sh.dscan, sh.base, sh.dpad,
                     sh.bnorm, npos, (const uint32_t *)nullptr, (const uint32_t *)nullptr, 0u, h->st.n, static_cast<ShadowFacts *>(h->d_shadow_facts));
  ZCHK(hipGetLastError());
  ShadowFacts f{};
  ZCHK(hipMemcpyAsync(&f, h->d_shadow_facts, sizeof(f), hipMemcpyDeviceToHost, s));
  ZCHK(hipStreamSynchronize(s));
  if (!(__builtin_bit_cast(float, f.max_abs) < 65504.f)) { flat_drop_shadow(h); return ZVEC_HIP_ERR_UNSUPPORTED; }
  h->shadow_max_err = __builtin_bit_cast(float, f.max_err);
  h->shadow_max_norm = __builtin_bit_cast(float, f.max_norm);
  h->shadow_kp = preselect;
  h->shadow_gov.reset();
  h->shadow_on = true;
  return 0;
}

int zvec_hip_flat_shadow_info(zvec_hip_flat_t h, int *enabled, uint64_t *bytes, float *max_row_error, float *max_row_norm) {
  if (!h) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  std::shared_lock<FairSharedMutex> r(h->rw);
  if (enabled) *enabled = h->shadow_on ? 1 : 0;
  if (bytes) *bytes = h->shadow_on ? (uint64_t)h->shadow.cap_tiles * TILE_N * (h->shadow.dpad + 1) * sizeof(float) : 0;
  if (max_row_error) *max_row_error = h->shadow_on ? h->shadow_max_err : 0.f;
  if (max_row_norm) *max_row_norm = h->shadow_on ? h->shadow_max_norm : 0.f;
  return 0;
}

int zvec_hip_flat_shadow_width(zvec_hip_flat_t h, uint32_t topk, uint32_t *rows) {
  if (!h || !rows) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  *rows = !h->shadow_on ? 0 : std::min<uint32_t>(64, h->shadow_kp ? h->shadow_kp : h->shadow_gov.kp_auto(topk));
  return 0;
}

int zvec_hip_flat_shadow_certify(zvec_hip_flat_t h, zvec_hip_ctx_t ctx, const void *d_queries, uint32_t count, uint32_t topk,
                                 const uint64_t *d_exclude_bitset, uint64_t *d_out_keys, float *d_out_scores, uint32_t *d_out_counts,
                                 void *stream, uint32_t *rerun) {
  if (!h || !d_queries || !d_out_keys || !d_out_scores || !d_out_counts) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  zvec_hip_ctx_s *c = ctx ? ctx : h->defctx;
  std::lock_guard<std::mutex> g(c->mu);
  std::shared_lock<FairSharedMutex> r(h->rw);
  ZCHK(hipSetDevice(h->device));
  return flat_shadow_certify_locked(h, c, d_queries, count, topk, d_exclude_bitset, d_out_keys, d_out_scores, d_out_counts,
                                    pick_stream(c, stream), rerun);
}

int zvec_hip_flat_search_dev(zvec_hip_flat_t h, zvec_hip_ctx_t ctx, const void *d_queries, uint32_t count,
                             uint32_t topk, float threshold, const uint64_t *d_exclude_bitset, uint64_t *d_out_keys,
                             float *d_out_scores, uint32_t *d_out_counts, void *stream) {
  if (!h || !d_queries || !d_out_keys || !d_out_scores || !d_out_counts) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (count == 0) return 0;
  if (topk == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;   // "Invalid context or topk not set yet" flat_searcher.cc:194
  zvec_hip_ctx_s *c = ctx ? ctx : h->defctx;
  std::lock_guard<std::mutex> g(c->mu);
  std::shared_lock<FairSharedMutex> r(h->rw);
  ZCHK(hipSetDevice(h->device));
  return flat_search_dev_locked(h, c, d_queries, count, topk, threshold, d_exclude_bitset, d_out_keys, d_out_scores,
                                d_out_counts, pick_stream(c, stream));
}

int zvec_hip_flat_search(zvec_hip_flat_t h, zvec_hip_ctx_t ctx, const void *queries, uint32_t count, uint32_t topk,
                         float threshold, const uint64_t *exclude_bitset, uint64_t *out_keys, float *out_scores,
                         uint32_t *out_counts) {
  if (!h || !queries || !out_keys || !out_scores || !out_counts) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (count == 0) return 0;
  if (topk == 0) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  zvec_hip_ctx_s *c = ctx ? ctx : h->defctx;
  // ONE critical section from the upload to the copy-out: the staging buffers belong to the context, and a NULL ctx
  // means several threads share the handle's built-in one (include/zvec_hip.h: "under a mutex")
  std::lock_guard<std::mutex> g(c->mu);
  ZCHK(hipSetDevice(h->device));
  {
    std::shared_lock<FairSharedMutex> r(h->rw);      // the row count the bitset is sized for == the rows scanned
    ZRET(host_search_wrap_begin(c, queries, (size_t)count * h->st.row_bytes(), exclude_bitset, h->st.n, count, topk, c->cur));
    ZRET(flat_search_dev_locked(h, c, c->io_qp, count, topk, threshold, exclude_bitset ? c->io_ex.as<uint64_t>() : nullptr,
                                c->io_keys.as<uint64_t>(), c->io_scores.as<float>(), c->io_counts.as<uint32_t>(), c->cur));
    if (c->sh_count)       // shadow rows: the queries whose result could not be certified are re-run on the fp32 rows
      ZRET(flat_shadow_certify_locked(h, c, c->io_qp, count, topk, exclude_bitset ? c->io_ex.as<uint64_t>() : nullptr,
                                      c->io_keys.as<uint64_t>(), c->io_scores.as<float>(), c->io_counts.as<uint32_t>(), c->cur, nullptr));
  }
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
  std::shared_lock<FairSharedMutex> r(h->rw);
  ZCHK(hipSetDevice(h->device));
  hipStream_t s = c->cur;
  const Store &st = h->st;
  ZRET(flat_wait_appends(h, s));
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
    bool ok = id < st.n && !h->is_hole(id);
    if (ok && exclude_bitset) ok = ((exclude_bitset[id >> 6] >> (id & 63)) & 1ull) == 0;
    clean[i] = ok ? id : IDX_NONE;
  }
  ZRET(host_search_wrap_begin(c, queries, (size_t)count * st.row_bytes(), nullptr, 0, count, topk, s));
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
  MergeArgs m{};
  m.part_s = c->part_s.as<float>(); m.part_i = c->part_i.as<uint32_t>(); m.part_keys = nullptr; m.slot_begin = nullptr;
  m.slots_per_q = 1; m.slot_stride = 1; m.part_counts = nullptr; m.k = topk; m.slot_len = maxlen; m.threshold = threshold;
  m.keymap = st.keys; m.out_keys = c->io_keys.as<uint64_t>(); m.out_scores = c->io_scores.as<float>(); m.out_idx = nullptr;
  m.out_counts = c->io_counts.as<uint32_t>();
  hipLaunchKernelGGL(merge_kernel, dim3(count), dim3(64), (size_t)topk * 12 + 16, s, m);
  ZCHK(hipGetLastError());
  return host_search_wrap_end(c, count, topk, out_keys, out_scores, out_counts, s);
}

// IndexMetric::batch_distance (index_metric.h:85-87; ailego BaseDistance::ComputeBatch, math_batch/distance_batch.h:29-49):
// ONE query against `n` scattered stored rows, scores only, in the listed order — for L2 / IP the reference's batch form
// is a loop of the 1x1 kernel; here one wave per listed row scores it directly (the gather kernel of
// search_bf_by_p_keys_impl without the selection).  positions out of range score +inf.
int zvec_hip_flat_batch_distance(zvec_hip_flat_t h, zvec_hip_ctx_t ctx, const void *query, const uint32_t *positions,
                                 uint32_t n, float *out_scores) {
  if (!h || !query || (n && (!positions || !out_scores))) return ZVEC_HIP_ERR_INVALID_ARGUMENT;
  if (n == 0) return 0;
  zvec_hip_ctx_s *c = ctx ? ctx : h->defctx;
  std::lock_guard<std::mutex> g(c->mu);
  std::shared_lock<FairSharedMutex> r(h->rw);
  ZCHK(hipSetDevice(h->device));
  hipStream_t s = c->cur;
  const Store &st = h->st;
  ZRET(flat_wait_appends(h, s));
  std::vector<uint32_t> clean(positions, positions + n);
  for (auto &p : clean) if (p >= st.n) p = IDX_NONE;
  const uint32_t offs[2] = {0, n};
  ZRET(c->io_q.ensure(st.row_bytes()));
  ZCHK(hipMemcpyAsync(c->io_q.p, query, st.row_bytes(), hipMemcpyHostToDevice, s));
  ZRET(prep_queries(c, st, c->io_q.p, 1, FLT_MAX, s));
  ZRET(c->plan.ensure(((size_t)n + 8) * sizeof(uint32_t)));
  uint32_t *d_pos = c->plan.as<uint32_t>(), *d_off = d_pos + n;
  ZCHK(hipMemcpyAsync(d_pos, clean.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
  ZCHK(hipMemcpyAsync(d_off, offs, sizeof(offs), hipMemcpyHostToDevice, s));
  ZRET(c->part_s.ensure((size_t)n * 4));
  ZRET(c->part_i.ensure((size_t)n * 4));
  if (st.f16)
    hipLaunchKernelGGL(pkeys_score_kernel<true>, dim3(pkeys_score_blocks(1, n)), dim3(256), 0, s, st.base, c->qpad.as<float>(), st.dpad,
                       st.metric, d_pos, d_off, 1u, n, c->part_s.as<float>(), c->part_i.as<uint32_t>());
  else
    hipLaunchKernelGGL(pkeys_score_kernel<false>, dim3(pkeys_score_blocks(1, n)), dim3(256), 0, s, st.base, c->qpad.as<float>(), st.dpad,
                       st.metric, d_pos, d_off, 1u, n, c->part_s.as<float>(), c->part_i.as<uint32_t>());
  ZCHK(hipGetLastError());
  ZCHK(hipMemcpyAsync(out_scores, c->part_s.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  ZCHK(hipStreamSynchronize(s));
  return 0;
}
