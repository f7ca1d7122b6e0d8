// api_types.inc.h — error macros, device buffers, the blocked Store, context and index handle structs
// Part of zvec_hip_api.hip (one translation unit; included in order, not standalone).

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

// a typed window into somebody else's device buffer
struct DevView {
  void *p = nullptr;
  template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// pinned host staging (small transfers of the host-pointer entry points: a copy from / to pageable memory is staged
// and synchronised by the runtime, a pinned one is a plain asynchronous DMA)
struct PinnedBuf {
  void *p = nullptr;
  void *dev = nullptr;   // the same bytes as the device addresses them (host-mapped: kernels read / write the slot in place)
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) { (void)hipHostFree(p); p = nullptr; dev = nullptr; cap = 0; }
    size_t want = bytes + bytes / 4 + 256;
    ZCHK(hipHostMalloc(&p, want, hipHostMallocMapped));
    if (hipHostGetDevicePointer(&dev, p, 0) != hipSuccess) { (void)hipGetLastError(); dev = nullptr; }
    cap = want;
    return 0;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; dev = nullptr; cap = 0; }
};

// Process-wide run-time options of the host-pointer entry points (zvec_hip_set_option; environment at first use).
//   wait      how a host-pointer search waits for its stream.  The product calls boundary B from many threads, one query per
//             call (index.cc:605-619): the runtime's spinning hipStreamSynchronize keeps every waiting thread on a CPU, and
//             with more callers than CPUs the threads that have kernels to launch wait for time slices behind them.
//             0 = spin (hipStreamSynchronize), 1 = poll a completion word in pinned memory, yielding the CPU between polls
//             (a one-thread kernel at the end of the chain writes it), 2 = block on an event created with hipEventBlockingSync
//   zerocopy  small transfers skip the copy engine.  bit 1 (value 2, the default): the last kernels write keys | scores | counts
//             into the host-mapped result slot; bit 0: the first kernel reads the query rows from the host-mapped pinned slot —
//             measured SLOWER than the staged copy (single query, 10M x 768: 0.121 against 0.104 ms per call), off by default
struct RuntimeOpts {
  std::atomic<int> wait{1};
  std::atomic<int> zerocopy{2};
  std::atomic<int> assign256{1};     // fp16 labelling on the 256 x 256 multi-phase tile (0: always the 128 x 128 one-barrier tile)
  std::atomic<int> scan256{1};       // wide fp16 flat scans (>= 256 queries, k <= 11) on the 256 x 256 multi-phase tile (0: scan8_kernel; 2: on cache-resident bases too)
  RuntimeOpts() {
    if (const char *e = getenv("ZVEC_HIP_SCAN256")) scan256 = std::max(0, std::min(2, atoi(e)));
    if (const char *e = getenv("ZVEC_HIP_WAIT")) wait = std::max(0, std::min(2, atoi(e)));
    if (const char *e = getenv("ZVEC_HIP_ASSIGN256")) assign256 = atoi(e) != 0;
    if (const char *e = getenv("ZVEC_HIP_ZEROCOPY")) zerocopy = std::max(0, std::min(3, atoi(e)));
  }
};
inline RuntimeOpts &ropts() {
  static RuntimeOpts o;
  return o;
}

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

// reader/writer lock that cannot starve the writer (std::shared_mutex on glibc prefers readers: searches that overlap
// continuously would keep an append waiting): everybody passes a gate, a writer keeps it while the readers drain
class FairSharedMutex {
 public:
  void lock() { gate_.lock(); rw_.lock(); gate_.unlock(); }
  void unlock() { rw_.unlock(); }
  void lock_shared() { gate_.lock(); rw_.lock_shared(); gate_.unlock(); }
  void unlock_shared() { rw_.unlock_shared(); }
 private:
  std::mutex gate_;
  std::shared_mutex rw_;
};

// When the data does not let the half-width pre-selection certify its results (rows within the fp16 rounding of each other, a few
// rows of huge norm under inner product) every search pays the fp16 scan AND the fp32 re-run.  The governor watches the certify
// steps: four in a row that had to re-run more than half of their queries suspend the shadow route for the next 64 searches of the
// index (they read the fp32 rows directly), after which it is tried again.  Results are the same either way; this only bounds the
// cost of data the twin cannot serve.
// It also sizes the pre-selection when the caller left it open (preselect = 0): k' starts at max(32, 3k); six certify steps in a row
// without a single re-run narrow it by 8 (a narrower list is cheaper to keep: flat 1M x 768, batch 256: 194 k QPS at 32, 232 k at 16),
// a step that had to re-run more than 1/32 of its queries widens it by 8 and fixes the width it failed at as the floor from then on.
// Range: max(16, 1.5 k rounded up to 8) .. 64.
struct ShadowGovernor {
  std::atomic<uint32_t> bad{0}, pause{0}, clean{0};
  std::atomic<int> level{0}, floor_level{-8};
  uint32_t kp_auto(uint32_t topk) const {
    const int base = (int)std::max<uint32_t>(32, 3 * topk), lo = (int)std::max<uint32_t>(16, (topk * 3 / 2 + 7) / 8 * 8);
    return (uint32_t)std::min(64, std::max(lo, base + 8 * level.load(std::memory_order_relaxed)));
  }
  void report_width(uint32_t rerun, uint32_t count) {      // one call per certify step of a search whose width was left open
    if (rerun == 0) {
      if (clean.fetch_add(1, std::memory_order_relaxed) + 1 >= 6) {
        clean.store(0, std::memory_order_relaxed);
        const int l = level.load(std::memory_order_relaxed);
        if (l > floor_level.load(std::memory_order_relaxed) && l > -4) level.store(l - 1, std::memory_order_relaxed);
      }
    } else {
      clean.store(0, std::memory_order_relaxed);
      if ((uint64_t)rerun * 32 > count) {
        const int l = level.load(std::memory_order_relaxed);
        floor_level.store(std::max(floor_level.load(std::memory_order_relaxed), l + 1), std::memory_order_relaxed);
        if (l < 4) level.store(l + 1, std::memory_order_relaxed);
      }
    }
  }
  bool allow() {                                     // one call per search that could use the shadow rows
    uint32_t p = pause.load(std::memory_order_relaxed);
    while (p > 0 && !pause.compare_exchange_weak(p, p - 1, std::memory_order_relaxed)) {}
    return p == 0;
  }
  void report(uint32_t rerun, uint32_t count) {      // one call per certify step
    if ((uint64_t)rerun * 2 > count) {
      if (bad.fetch_add(1, std::memory_order_relaxed) + 1 >= 4) { bad.store(0, std::memory_order_relaxed); pause.store(64, std::memory_order_relaxed); }
    } else {
      bad.store(0, std::memory_order_relaxed);
    }
  }
  void reset() {
    bad.store(0, std::memory_order_relaxed); pause.store(0, std::memory_order_relaxed); clean.store(0, std::memory_order_relaxed);
    level.store(0, std::memory_order_relaxed); floor_level.store(-8, std::memory_order_relaxed);
  }
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

// Contexts that share a gate take turns on their dominant scan kernel (in call order) while everything else of their
// searches — query preparation, coarse pass, plan, merges, refinement, the caller's exchange — overlaps the other
// context's scan: one event, re-recorded behind every gated scan, waited for in front of the next one.
struct zvec_hip_gate_s {
  int device = 0;
  hipEvent_t ev = nullptr;
  bool armed = false;
  std::mutex mu;
};

struct zvec_hip_ctx_s {
  int device = 0;
  zvec_hip_gate_s *gate = nullptr;
  hipStream_t own = nullptr;
  hipStream_t cur = nullptr;
  std::mutex mu;
  // workspace
  DevBuf gtau, ridx;
  DevBuf seed_keys, seed_scores, seed_counts, seed_idx;   // sample scan that seeds the shared admission bounds
  DevBuf cmp_base, cmp_norm, cmp_extra, cmp_keys, cmp_pos, cmp_cnt;   // compacted keep-set (sparse filters)
  DevBuf qpad, qnorm, part_s, part_i, coarse_keys, coarse_scores, coarse_idx, coarse_cnt;
  DevBuf plan;        // all u32 plan arrays
  DevBuf io_q, io_ex, io_out;                          // staging for host-pointer entry points
  DevBuf io_cq;                                        // ... the coarse-space queries of zvec_hip_ivf_search_coarse
  DevView io_keys, io_scores, io_counts;               // the result arrays inside io_out: ONE copy brings them back
  DevBuf grp_ws, grp_of, grp_out, grp_tab;
  DevBuf direct_pos, direct_keys, direct_scores, direct_idx, direct_cnt;   // small-batch IVF route: positions, stage-1 lists
  // half-width pre-selection (zvec_hip_ivf_set_shadow): fp16 query rows + norms, per-query rounding facts, the k' pre-selected rows
  // of every query (keys | shadow scores | true scores | positions | counts), flags [count] + the flagged count [1]
  DevBuf sh_q16, sh_qn16, sh_qinfo, sh_keys, sh_scores, sh_true, sh_idx, sh_counts, sh_flags;
  uint32_t sh_count = 0;                               // queries of the last search that went through the shadow lists (0: none)
  bool shadow_skip = false;                            // the certify step's re-run: this search must read the fp32 lists
  uint32_t shadow_force_kp = 0;                        // the certify step's SECOND half-width pass over the flagged queries: this width
  uint32_t sh_kp = 0;                                  // width the last shadow search on this context used
  int sh_tier = 0;                                     // 1: inside the second pass (its own flagged queries go to the fp32 rows)
  bool shadow_scan = false;                            // flat_scan_prepared is running over a shadow store: profiled / gated like a user-facing scan
  DevBuf holes_ex;                                     // caller's exclude set OR the store's holes                      // group-by search: per-group bests / lists, group of every position, results
  PinnedBuf pin_in, pin_out;                           // (transfers up to PIN_LIMIT bytes go through pinned memory)
  const void *io_qp = nullptr;                         // where device code finds the uploaded queries: io_q or the mapped pin_in slot
  bool out_mapped = false;                             // io_keys / io_scores / io_counts point into the mapped pin_out slot
  PinnedBuf done_word;                                 // wait policy 1: completion word (epoch) a one-thread kernel writes
  uint32_t done_epoch = 0;
  uint64_t last_wait_ns = 0;                           // how long the previous call waited: spin (short) or sleep (long) next time
  hipEvent_t block_ev = nullptr;                       // wait policy 2: hipEventBlockingSync
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
  // fp16 twin of `st` at the same positions (own base + bnorm): zvec_hip_flat_set_shadow.  Built for the rows present at that moment;
  // any mutation of the store drops it (the store then searches its own rows until it is set again)
  Store shadow;
  bool shadow_on = false;
  uint32_t shadow_kp = 0;
  void *d_shadow_facts = nullptr;      // zvk::ShadowFacts
  float shadow_max_err = 0.f, shadow_max_norm = 0.f;
  ShadowGovernor shadow_gov;
  zvec_hip_ctx_s *defctx = nullptr;
  std::mutex mu;            // serialises the calls that use defctx's workspace (appends, get_vector)
  // The streamer is searched while it grows (flat_streamer_test.cc TestConcurrentAddAndSearch): searches hold `rw`
  // shared while they read the store's pointers / row count and enqueue their kernels; anything that may move or
  // extend the store holds it exclusive.  A growth reallocation frees the old arrays with hipFree, which waits for
  // the device, so kernels enqueued by earlier searches have finished with them.
  FairSharedMutex rw;
  // zvec_hip_flat_append_dev returns with its pack kernels only enqueued: recorded after them on the append stream,
  // waited for by every reader of the store on its own stream
  // asynchronous mutations (append_dev, single-document adds): `append_stream` carries kernels that are only enqueued.
  // A reader on ANOTHER stream makes its stream wait for them: it records append_ev behind them if nobody has since
  // the last mutation (append_dirty, under ev_mu — readers run concurrently) and waits for the event; a reader on the
  // same stream is ordered by the stream.  Writers hold `rw` exclusively, so these fields do not move under a reader.
  hipEvent_t append_ev = nullptr;
  bool append_pending = false;
  bool append_dirty = false;
  hipStream_t append_stream = nullptr;
  std::mutex ev_mu;
  // add-with-id gaps (FlatStreamerEntity::add_vector_with_id pads positions [count, id) with kInvalidKey rows that no scan
  // returns, flat_streamer_entity.cc:935-952): one bit per storage position, host copy + device copy, OR-ed into every
  // search's exclude set while any hole exists
  std::vector<uint64_t> h_holes;
  uint64_t nholes = 0;
  DevBuf d_holes;
  // single-document adds (the product ingests one add_with_id_impl per document): a ring of pinned slots that the pack
  // kernel reads in place — no staging copy, no allocation, no synchronisation per document
  static constexpr uint32_t FAST_ROWS = 8, RING = 64, RING_GROUP = 16;     // one event per group of slots
  PinnedBuf ring;
  size_t ring_slot_bytes = 0;
  hipEvent_t ring_ev[RING / RING_GROUP] = {};
  bool ring_used[RING / RING_GROUP] = {};
  uint32_t ring_next = 0;
  bool is_hole(uint64_t pos) const { return (pos >> 6) < h_holes.size() && ((h_holes[pos >> 6] >> (pos & 63)) & 1ull); }
};

struct zvec_hip_ivf_s {
  int device = 0;
  int dtype = 0;
  uint32_t dim = 0;
  int metric = 0;
  uint32_t nlist = 0;
  uint32_t shard = 0, nshards = 1;
  bool loaded = false;
  bool trained = false;                // centroids present (h_centroids + cent store): labelling possible
  bool filling = false;                // between begin_lists and end_lists of a streamed build
  std::vector<uint32_t> h_owner;       // list -> shard (byte-balanced, identical on every rank)
  std::vector<uint64_t> h_cursor;      // streamed build: next dense position of each list
  Store cent;     // centroids as a flat store
  bool coarse_sep = false;             // the centroid store lives in a space of its own (dimension / metric): zvec_hip_ivf_set_coarse_space
  Store lists;    // inverted lists, each padded to whole tiles
  // fp16 twin of `lists` at the same positions (own base + bnorm; keys / geometry are the lists'): zvec_hip_ivf_set_shadow
  Store shadow;
  bool shadow_on = false;
  uint32_t shadow_kp = 0;              // rows pre-selected per query (0: from k)
  void *d_shadow_facts = nullptr;      // zvk::ShadowFacts
  float shadow_max_err = 0.f, shadow_max_norm = 0.f;
  ShadowGovernor shadow_gov;
  uint64_t count_local = 0, count_global = 0;
  std::vector<uint32_t> h_size, h_size_global, h_tile0;
  std::vector<uint64_t> h_rows_of_largest;   // [i] = rows of the i largest local lists (bound of what i probes can scan)
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
