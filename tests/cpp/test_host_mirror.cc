// C++ parity test of the host mirror (zvec_amd/csrc/host/hip_index.h) over the C ABI, written like the
// reference's own tests.  Expectations are those of
//   tests/core/algorithm/flat/flat_streamer_test.cc:104-178 (TestLinearSearch), :731-801 (TestFilter)
//   tests/core/algorithm/flat/flat_streamer_test.cc:929-1037 (TestGroup), :1038-1117 (TestAddAndSearchWithID)
//   tests/core/algorithm/ivf/ivf_searcher_test.cc:200-321 (TestSimple), :2830-2886 (TestRnnSearch shape)
// Needs a GPU.  Exit code 0 = all checks passed.
#include <cstdio>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../zvec_amd/csrc/host/hip_index.h"

using namespace zvec_hip_host;
static int g_fail = 0;

// Bounded waits.  Round 2's reader/writer starvation showed up as a silent 300 s kill with no output: every thread of the
// concurrent tests now reports what it is about to do (TICK), and a watchdog ends the program with that picture — the test that
// is running, each worker's last reported step, how long nothing has moved — when NO thread has reported for STALL_SECONDS.
static std::atomic<uint64_t> g_progress{0};
static std::atomic<const char *> g_test{"(start)"};
constexpr int kSlots = 8;
static std::atomic<const char *> g_where[kSlots];
static std::atomic<uint64_t> g_steps[kSlots];
constexpr int STALL_SECONDS = 60;
#define TICK(slot, what)            \
  do {                              \
    g_where[(slot)] = (what);       \
    ++g_steps[(slot)];              \
    ++g_progress;                   \
  } while (0)
static void watchdog() {
  uint64_t last = g_progress.load();
  int idle = 0;
  for (;;) {
    std::this_thread::sleep_for(std::chrono::seconds(1));
    const uint64_t now = g_progress.load();
    if (now != last) { last = now; idle = 0; continue; }
    if (++idle < STALL_SECONDS) continue;
    printf("STALL: no thread has made progress for %d s in %s\n", STALL_SECONDS, g_test.load());
    for (int i = 0; i < kSlots; ++i)
      if (g_where[i].load()) printf("  worker %d: last step \"%s\" (%llu steps)\n", i, g_where[i].load(), (unsigned long long)g_steps[i].load());
    fflush(stdout);
    _exit(3);
  }
}
#define RUN(test)                                   \
  do {                                              \
    g_test = #test;                                 \
    for (int i_ = 0; i_ < kSlots; ++i_) { g_where[i_] = nullptr; g_steps[i_] = 0; } \
    ++g_progress;                                   \
    rc |= test();                                   \
    ++g_progress;                                   \
  } while (0)
#define EXPECT(cond)                                                              \
  do {                                                                            \
    if (!(cond)) { printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); ++g_fail; } \
  } while (0)
#define ASSERT(cond)                                                              \
  do {                                                                            \
    if (!(cond)) { printf("FATAL %s:%d  %s\n", __FILE__, __LINE__, #cond); return 1; } \
  } while (0)

static int TestLinearSearch() {
  constexpr size_t dim = 16;
  IndexMeta meta(IndexMeta::DT_FP32, dim);
  meta.set_metric("SquaredEuclidean");
  HipFlatStreamer streamer;
  ASSERT(0 == streamer.init(meta, Params()));
  ASSERT(0 == streamer.open());
  auto ctx = streamer.create_context();
  ASSERT(!!ctx);
  size_t cnt = 1000UL;
  IndexQueryMeta qmeta(IndexMeta::DT_FP32, dim);
  for (size_t i = 0; i < cnt; i++) {
    std::vector<float> vec(dim, (float)i);
    ASSERT(0 == streamer.add_impl(i, vec.data(), qmeta, ctx));
  }
  size_t topk = 3;
  for (size_t i = 0; i < cnt; i += 7) {
    std::vector<float> vec(dim, (float)i);
    ctx->set_topk(topk);
    ASSERT(0 == streamer.search_impl(vec.data(), qmeta, ctx));
    auto &result1 = ctx->result();
    ASSERT(topk == result1.size());
    EXPECT(i == result1[0].key());
    std::vector<float> got;
    ASSERT(0 == streamer.get_vector_by_id((uint32_t)result1[0].key(), &got));
    for (size_t j = 0; j < dim; ++j) EXPECT(got[j] == (float)i);
    for (size_t j = 0; j < dim; ++j) vec[j] = i + 0.1f;
    ASSERT(0 == streamer.search_impl(vec.data(), qmeta, ctx));
    auto &result2 = ctx->result();
    ASSERT(topk == result2.size());
    EXPECT(i == result2[0].key());
    EXPECT((i == cnt - 1 ? i - 1 : i + 1) == result2[1].key());
    EXPECT((i == 0 ? 2 : (i == cnt - 1 ? i - 2 : i - 1)) == result2[2].key());
  }
  ctx->set_topk(100U);
  std::vector<float> vec(dim, 10.1f);
  ASSERT(0 == streamer.search_bf_impl(vec.data(), qmeta, ctx));
  auto &result = ctx->result();
  ASSERT(100U == result.size());
  EXPECT(10 == result[0].key());
  EXPECT(11 == result[1].key());
  EXPECT(5 == result[10].key());
  EXPECT(0 == result[20].key());
  EXPECT(30 == result[30].key());
  EXPECT(35 == result[35].key());
  EXPECT(99 == result[99].key());
  // error behaviour: topk not set -> InvalidArgument; wrong qmeta -> InvalidArgument
  auto ctx2 = streamer.create_context();
  EXPECT(IndexError_InvalidArgument == streamer.search_impl(vec.data(), qmeta, ctx2));
  ctx2->set_topk(1);
  IndexQueryMeta bad(IndexMeta::DT_FP32, dim + 1);
  EXPECT(IndexError_InvalidArgument == streamer.search_impl(vec.data(), bad, ctx2));
  return 0;
}

static int TestFilter() {
  constexpr size_t dim = 16;
  IndexMeta meta(IndexMeta::DT_FP32, dim);
  meta.set_metric("SquaredEuclidean");
  HipFlatStreamer streamer;
  ASSERT(0 == streamer.init(meta, Params()));
  ASSERT(0 == streamer.open());
  auto ctx = streamer.create_context();
  ASSERT(!!ctx);
  ctx->set_topk(10U);
  IndexQueryMeta qmeta(IndexMeta::DT_FP32, dim);
  size_t cnt = 2000;
  std::vector<float> all(cnt * dim);
  for (size_t i = 0; i < cnt; i++) for (size_t j = 0; j < dim; ++j) all[i * dim + j] = (float)i;
  ASSERT(0 == streamer.add_batch(all.data(), cnt, nullptr));
  std::vector<float> vec(dim, 100.1f);
  ASSERT(0 == streamer.search_impl(vec.data(), qmeta, ctx));
  auto &results = ctx->result();
  ASSERT(10 == results.size());
  EXPECT(100 == results[0].key());
  EXPECT(101 == results[1].key());
  EXPECT(99 == results[2].key());
  auto filterFunc = [](uint64_t key) { return key == 100UL || key == 101UL; };
  ctx->set_filter(filterFunc);
  ASSERT(0 == streamer.search_impl(vec.data(), qmeta, ctx));
  auto &results1 = ctx->result();
  ASSERT(10 == results1.size());
  EXPECT(99 == results1[0].key());
  EXPECT(102 == results1[1].key());
  EXPECT(98 == results1[2].key());
  ASSERT(0 == streamer.search_bf_impl(vec.data(), qmeta, ctx));
  auto &results2 = ctx->result();
  ASSERT(10 == results2.size());
  EXPECT(99 == results2[0].key());
  EXPECT(102 == results2[1].key());
  EXPECT(98 == results2[2].key());
  // fetch_vector: documents carry their stored rows (index_context.h:139, index.cc:635-647)
  ctx->reset_filter();
  ctx->set_fetch_vector(true);
  ASSERT(0 == streamer.search_impl(vec.data(), qmeta, ctx));
  {
    auto &rf = ctx->result();
    ASSERT(10 == rf.size());
    for (auto &d : rf) {
      ASSERT(d.vector().size() == dim * sizeof(float));
      const float *v = reinterpret_cast<const float *>(d.vector().data());
      for (size_t j = 0; j < dim; ++j) EXPECT(v[j] == (float)d.key());
    }
  }
  ctx->set_fetch_vector(false);
  // the same predicate as data: a delete store holding {100, 101} in roaring portable form
  // (cookie 12346, 1 container; key 0, cardinality-1 = 1; offset 16; values) — materialised on the GPU
  const uint8_t deleted[20] = {0x3a, 0x30, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 16, 0, 0, 0, 100, 0, 101, 0};
  zvec_hip_doc_filter_t df{};
  df.delete_bitmap = deleted; df.delete_bytes = sizeof(deleted); df.delete_kind = ZVEC_HIP_ROARING_32;
  ctx->reset_filter();
  ctx->set_doc_filter(df);
  ASSERT(0 == streamer.search_impl(vec.data(), qmeta, ctx));
  auto &results3 = ctx->result();
  ASSERT(10 == results3.size());
  EXPECT(99 == results3[0].key());
  EXPECT(102 == results3[1].key());
  EXPECT(98 == results3[2].key());
  // an inverted-index result set {98, 99, 100}: everything else is excluded, 100 is deleted
  const uint8_t matched[22] = {0x3a, 0x30, 0, 0, 1, 0, 0, 0, 0, 0, 2, 0, 16, 0, 0, 0, 98, 0, 99, 0, 100, 0};
  df.invert_bitmap = matched; df.invert_bytes = sizeof(matched);
  ctx->set_doc_filter(df);
  ASSERT(0 == streamer.search_impl(vec.data(), qmeta, ctx));
  auto &results4 = ctx->result();
  ASSERT(2 == results4.size());
  EXPECT(99 == results4[0].key());
  EXPECT(98 == results4[1].key());
  return 0;
}

static int TestIVFSimple() {
  constexpr uint32_t dimension_ = 8;
  IndexMeta meta(IndexMeta::DT_FP32, dimension_);
  meta.set_metric("SquaredEuclidean");
  const size_t n = 33;
  std::vector<float> base(n * dimension_), centroid(dimension_, 16.0f);
  std::vector<uint64_t> keys(n), offs = {0, n};
  for (size_t i = 0; i < n; ++i) { keys[i] = i; for (size_t j = 0; j < dimension_; ++j) base[i * dimension_ + j] = 1.0f * i; }
  HipIVFSearcher searcher;
  Params params;
  params.set(PARAM_IVF_SEARCHER_SCAN_RATIO, 1.0);
  params.set(PARAM_IVF_SEARCHER_BRUTE_FORCE_THRESHOLD, 1);
  ASSERT(0 == searcher.init(params));
  EXPECT(nullptr == searcher.create_context());                  // not loaded yet
  ASSERT(0 == searcher.load(meta, centroid.data(), 1, offs.data(), base.data(), keys.data()));
  std::vector<float> query(dimension_, 32.0f);
  size_t qnum = 33;
  std::vector<float> query1;
  for (size_t i = 0; i < dimension_ * qnum; ++i) query1.push_back((float)(i / dimension_));
  auto context = searcher.create_context();
  ASSERT(!!context);
  IndexQueryMeta qmeta(IndexMeta::DT_FP32, dimension_);
  for (int pass = 0; pass < 2; ++pass) {   // pass 0: bf search, pass 1: knn search
    size_t topk = 33;
    context->set_topk(topk);
    int ret = pass == 0 ? searcher.search_bf_impl(query.data(), qmeta, context) : searcher.search_impl(query.data(), qmeta, context);
    ASSERT(0 == ret);
    const IndexDocumentList &result = context->result(0);
    ASSERT(topk == result.size());
    for (size_t i = 0; i < topk; ++i) {
      EXPECT((uint64_t)32 - i == result[i].key());
      EXPECT((float)i * i * dimension_ == result[i].score());
    }
    topk = 1;
    context->set_topk(topk);
    ret = pass == 0 ? searcher.search_bf_impl(query1.data(), qmeta, qnum, context) : searcher.search_impl(query1.data(), qmeta, qnum, context);
    ASSERT(0 == ret);
    for (size_t q = 0; q < qnum; ++q) {
      const IndexDocumentList &r = context->result(q);
      ASSERT(topk == r.size());
      EXPECT((uint64_t)q == r[0].key());
      EXPECT((float)0 == r[0].score());
    }
  }
  // RNN radius (TestRnnSearch shape): fewer than topk, all within the radius
  context->set_topk(33);
  context->set_threshold(100.0f);
  ASSERT(0 == searcher.search_impl(query.data(), qmeta, context));
  EXPECT(context->result().size() < 33 && context->result().size() > 0);
  for (auto &d : context->result()) EXPECT(d.score() <= 100.0f);
  EXPECT(0 == searcher.unload());
  return 0;
}

// Single-query callers on many threads, with and without the micro-batcher: identical result lists, and (printed) the
// searches per second of both — the product drives the boundary exactly like this (index.cc:617).
static int TestMicroBatcher() {
  const bool big = getenv("ZVEC_MIRROR_BIG") != nullptr;       // a larger index for a meaningful searches/s figure
  const uint32_t dim = big ? 128 : 32, nlist = big ? 256 : 24, per_list = big ? 4000 : 700, n = nlist * per_list;
  const uint32_t threads = big ? 64 : 32, per_thread = 100, topk = 10;
  IndexMeta meta(IndexMeta::DT_FP32, dim);
  meta.set_metric("SquaredEuclidean");
  std::vector<float> base((size_t)n * dim), cent((size_t)nlist * dim);
  std::vector<uint64_t> keys(n), offs(nlist + 1);
  uint32_t seed = 12345;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)((seed >> 9) & 1023) / 64.0f; };
  for (uint32_t l = 0; l < nlist; ++l) {
    offs[l] = (uint64_t)l * per_list;
    for (uint32_t j = 0; j < dim; ++j) cent[(size_t)l * dim + j] = (float)(l * 20);
    for (uint32_t i = 0; i < per_list; ++i) {
      const size_t r = (size_t)l * per_list + i;
      keys[r] = 7 * r + 1;
      for (uint32_t j = 0; j < dim; ++j) base[r * dim + j] = (float)(l * 20) + rnd();
    }
  }
  offs[nlist] = n;
  Params pa, pb;
  pa.set(PARAM_IVF_SEARCHER_SCAN_RATIO, 0.25);
  pa.set(PARAM_IVF_SEARCHER_BRUTE_FORCE_THRESHOLD, 10);
  pb = pa;
  pb.set(PARAM_HIP_SEARCHER_BATCH_WINDOW_US, 2000);
  pb.set(PARAM_HIP_SEARCHER_MAX_BATCH, 64);
  pb.set(PARAM_HIP_SEARCHER_BATCH_LINGER_US, 40);
  // ... and the shared batches through the certified half-width pre-selection (proxima.hip.searcher.half_width_preselect:
  // fp16 twin of the lists, fp32 re-scoring, certificate, fp32 re-run of what it cannot certify): the same lists again
  Params pc = pb;
  pc.set(PARAM_HIP_SEARCHER_HALF_WIDTH_PRESELECT, 1);
  HipIVFSearcher plain, batched, half;
  ASSERT(0 == plain.init(pa) && 0 == batched.init(pb) && 0 == half.init(pc));
  ASSERT(0 == plain.load(meta, cent.data(), nlist, offs.data(), base.data(), keys.data()));
  ASSERT(0 == batched.load(meta, cent.data(), nlist, offs.data(), base.data(), keys.data()));
  ASSERT(0 == half.load(meta, cent.data(), nlist, offs.data(), base.data(), keys.data()));
  IndexQueryMeta qmeta(IndexMeta::DT_FP32, dim);
  std::vector<float> queries((size_t)threads * per_thread * dim);
  for (auto &v : queries) v = rnd() + 20.0f * (float)((seed >> 20) % nlist);
  std::vector<IndexDocumentList> want((size_t)threads * per_thread);
  {
    auto ctx = plain.create_context();
    ctx->set_topk(topk);
    for (size_t i = 0; i < want.size(); ++i) {
      ASSERT(0 == plain.search_impl(&queries[i * dim], qmeta, ctx));
      want[i] = ctx->result();
    }
  }
  for (int pass = 0; pass < 3; ++pass) {
    HipIVFSearcher &se = pass == 2 ? half : pass ? batched : plain;
    std::atomic<int> bad{0};
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < threads; ++t)
      th.emplace_back([&, t]() {
        auto ctx = se.create_context();
        ctx->set_topk(topk);
        for (uint32_t i = 0; i < per_thread; ++i) {
          const size_t qi = (size_t)t * per_thread + i;
          TICK(t % kSlots, pass ? "micro-batched single query" : "direct single query");
          if (se.search_impl(&queries[qi * dim], qmeta, ctx) != 0) { ++bad; continue; }
          // same documents, same scores; documents with EQUAL scores may come in either order (a batch of > 8 queries
          // takes the list-major route, a single query the direct one: both keep the k best under (score, scan order), the
          // order inside a tie is unspecified in the reference too — heap.h:173-175 sorts unstably)
          auto r = ctx->result();
          auto w = want[qi];
          if (r.size() != w.size()) { ++bad; continue; }
          auto by_score_key = [](const IndexDocument &a, const IndexDocument &b) {
            return a.score() < b.score() || (a.score() == b.score() && a.key() < b.key());
          };
          for (size_t j = 1; j < r.size(); ++j) if (r[j - 1].score() > r[j].score()) ++bad;
          std::sort(r.begin(), r.end(), by_score_key);
          std::sort(w.begin(), w.end(), by_score_key);
          for (size_t j = 0; j < r.size(); ++j)
            if (r[j].key() != w[j].key() || r[j].score() != w[j].score()) {
              if (j + 1 == r.size() || r[j].score() == r.back().score()) continue;   // a tie at the k-th place may keep either document
              ++bad;
              break;
            }
        }
      });
    for (auto &x : th) x.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("  %u threads x %u single queries, %s: %.0f searches/s\n", threads, per_thread,
           pass == 2 ? "micro-batched, half-width pre-selection" : pass ? "micro-batched" : "direct",
           threads * per_thread / dt);
    EXPECT(bad.load() == 0);
  }
  return 0;
}

// ctx == NULL from several threads (include/zvec_hip.h: "uses the handle's built-in context under a mutex"): the staged
// queries / results of the built-in context must not be overwritten between upload, search and copy-out.  Flat and IVF,
// every thread its own queries, compared with what a private context returns.
static int TestNullContextFromManyThreads() {
  const uint32_t dim = 24, n = 6000, nq = 40, topk = 8, threads = 6, rounds = 30, nlist = 12;
  std::vector<float> base((size_t)n * dim);
  uint32_t seed = 777;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)((int)((seed >> 10) % 41) - 20); };
  for (auto &v : base) v = rnd();
  zvec_hip_flat_t flat = nullptr;
  ASSERT(0 == zvec_hip_flat_create(dim, ZVEC_HIP_DT_FP32, ZVEC_HIP_METRIC_L2, 0, &flat));
  ASSERT(0 == zvec_hip_flat_append(flat, base.data(), n, nullptr));
  // IVF over the same rows: centroids = the first nlist rows, lists = contiguous ranges
  std::vector<uint64_t> offs(nlist + 1);
  for (uint32_t l = 0; l <= nlist; ++l) offs[l] = (uint64_t)l * (n / nlist);
  zvec_hip_ivf_t ivf = nullptr;
  ASSERT(0 == zvec_hip_ivf_create(dim, ZVEC_HIP_DT_FP32, ZVEC_HIP_METRIC_L2, 0, &ivf));
  ASSERT(0 == zvec_hip_ivf_load(ivf, base.data(), nlist, offs.data(), base.data(), nullptr));
  std::vector<std::vector<float>> q(threads, std::vector<float>((size_t)nq * dim));
  for (auto &v : q) for (auto &x : v) x = rnd();
  struct Out { std::vector<uint64_t> k; std::vector<float> s; std::vector<uint32_t> c; };
  auto mk = [&]() { Out o; o.k.assign((size_t)nq * topk, 0); o.s.assign((size_t)nq * topk, 0.f); o.c.assign(nq, 0); return o; };
  std::vector<Out> want_f, want_i;
  zvec_hip_ctx_t own = nullptr;
  ASSERT(0 == zvec_hip_ctx_create(0, &own));
  for (uint32_t t = 0; t < threads; ++t) {
    Out a = mk(), b = mk();
    ASSERT(0 == zvec_hip_flat_search(flat, own, q[t].data(), nq, topk, FLT_MAX, nullptr, a.k.data(), a.s.data(), a.c.data()));
    ASSERT(0 == zvec_hip_ivf_search(ivf, own, q[t].data(), nq, topk, FLT_MAX, 4, n, nullptr, b.k.data(), b.s.data(), b.c.data()));
    want_f.push_back(a); want_i.push_back(b);
  }
  std::atomic<int> bad{0};
  std::vector<std::thread> th;
  for (uint32_t t = 0; t < threads; ++t)
    th.emplace_back([&, t]() {
      for (uint32_t r = 0; r < rounds; ++r) {
        Out a = mk(), b = mk();
        TICK(t % kSlots, "NULL-context flat + ivf search");
        if (zvec_hip_flat_search(flat, nullptr, q[t].data(), nq, topk, FLT_MAX, nullptr, a.k.data(), a.s.data(), a.c.data()) != 0 ||
            zvec_hip_ivf_search(ivf, nullptr, q[t].data(), nq, topk, FLT_MAX, 4, n, nullptr, b.k.data(), b.s.data(), b.c.data()) != 0) { ++bad; continue; }
        if (a.k != want_f[t].k || a.s != want_f[t].s || a.c != want_f[t].c) ++bad;
        if (b.k != want_i[t].k || b.s != want_i[t].s || b.c != want_i[t].c) ++bad;
        std::vector<float> row(dim);                         // get_vector shares the built-in context's staging buffer
        if (zvec_hip_flat_get_vector(flat, (t * 37 + r) % n, row.data()) != 0 ||
            memcmp(row.data(), &base[(size_t)((t * 37 + r) % n) * dim], dim * 4) != 0) ++bad;
      }
    });
  for (auto &x : th) x.join();
  EXPECT(bad.load() == 0);
  zvec_hip_ctx_destroy(own);
  zvec_hip_ivf_destroy(ivf);
  zvec_hip_flat_destroy(flat);
  return 0;
}

// add_impl from one thread while others search WITH a filter callback and fetch_vector (the mirror's host state —
// the keys of the storage positions, the lazily built key -> position map — is shared between them)
static int TestConcurrentAddWithFilterAndFetch() {
  const uint32_t dim = 16, n0 = 2000, n1 = 6000, topk = 6;
  IndexMeta meta(IndexMeta::DT_FP32, dim);
  meta.set_metric("SquaredEuclidean");
  HipFlatStreamer st;
  ASSERT(0 == st.init(meta, Params()));
  ASSERT(0 == st.open());
  IndexQueryMeta qmeta(IndexMeta::DT_FP32, dim);
  auto row = [&](uint64_t i) { std::vector<float> v(dim); for (uint32_t j = 0; j < dim; ++j) v[j] = (float)(i % 997) + (float)j; return v; };
  Context::Pointer none;
  for (uint64_t i = 0; i < n0; ++i) { auto v = row(i); ASSERT(0 == st.add_impl(i, v.data(), qmeta, none)); }
  std::atomic<int> bad{0};
  std::atomic<bool> done{false};
  std::thread adder([&]() {
    Context::Pointer c;
    for (uint64_t i = n0; i < n1; ++i) { auto v = row(i); TICK(0, "adder: add_impl"); if (st.add_impl(i, v.data(), qmeta, c) != 0) ++bad; }
    TICK(0, "adder: done");
    done = true;
  });
  std::vector<std::thread> th;
  for (int t = 0; t < 3; ++t)
    th.emplace_back([&, t]() {
      auto ctx = st.create_context();
      ctx->set_topk(topk);
      ctx->set_fetch_vector(true);
      ctx->set_filter([](uint64_t key) { return key % 2 == 1; });          // odd keys are filtered OUT
      std::vector<float> q = row(100 + t);
      int loops = 0;
      while (!done.load() || loops < 3) {
        ++loops;
        TICK(1 + t, "searcher: search_impl (filter + fetch_vector)");
        if (st.search_impl(q.data(), qmeta, ctx) != 0) { ++bad; continue; }
        TICK(1 + t, "searcher: checking the result");
        const auto &r = ctx->result();
        if (r.size() != topk) { ++bad; continue; }
        for (const auto &d : r) {
          if (d.key() % 2 == 1) ++bad;                                     // gate respected
          auto want = row(d.key());                                        // the fetched vector is the document's row
          if (d.vector().size() != dim * 4 || memcmp(d.vector().data(), want.data(), dim * 4) != 0) ++bad;
        }
        if (r[0].score() != 0.0f) ++bad;                                   // an even key with the query's row exists
      }
    });
  adder.join();
  for (auto &x : th) x.join();
  EXPECT(bad.load() == 0);
  EXPECT(st.count() == n1);
  return 0;
}

// add_with_id_impl from one thread — appends with gaps, gap fills and in-place overwrites — while three threads search
// (plain with a filter, p_keys, group-by).  Every row ever written for id i is (i % 500 + j): an overwrite changes
// nothing a searcher could tell apart, so the invariants hold at every moment: full result lists in ascending order, no
// hole ever returned (kInvalidKey, or an id never written), filtered keys absent, top score 0 for a query that equals a
// stored row, group documents inside their group.
static int TestConcurrentPutAndSearch() {
  const uint32_t dim = 16, n0 = 3000, n1 = 9000, topk = 8;
  IndexMeta meta(IndexMeta::DT_FP32, dim);
  meta.set_metric("SquaredEuclidean");
  HipFlatStreamer st;
  ASSERT(0 == st.init(meta, Params()));
  ASSERT(0 == st.open());
  IndexQueryMeta qmeta(IndexMeta::DT_FP32, dim);
  auto row = [&](uint64_t i) { std::vector<float> v(dim); for (uint32_t j = 0; j < dim; ++j) v[j] = (float)(i % 500) + (float)j; return v; };
  Context::Pointer none;
  for (uint32_t i = 0; i < n0; ++i) { auto v = row(i); ASSERT(0 == st.add_with_id_impl(i, v.data(), qmeta, none)); }
  std::atomic<int> bad{0};
  std::atomic<bool> done{false};
  std::thread writer([&]() {
    Context::Pointer c;
    uint32_t seed = 7;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    // multiples of 3 first (gaps of two holes each), overwrites of live rows in between, then the gaps
    for (uint32_t i = n0; i < n1; i += 3) {
      auto v = row(i);
      TICK(0, "writer: add_with_id_impl (append behind a gap)");
      if (st.add_with_id_impl(i, v.data(), qmeta, c) != 0) ++bad;
      const uint32_t o = rnd() % n0;
      auto w = row(o);
      TICK(0, "writer: add_with_id_impl (overwrite)");
      if (st.add_with_id_impl(o, w.data(), qmeta, c) != 0) ++bad;
    }
    for (uint32_t i = n0; i < n1; ++i)
      if (i % 3 != n0 % 3) { auto v = row(i); TICK(0, "writer: add_with_id_impl (gap fill)"); if (st.add_with_id_impl(i, v.data(), qmeta, c) != 0) ++bad; }
    TICK(0, "writer: done");
    done = true;
  });
  auto written = [&](uint64_t key) { return key < n1; };
  std::vector<std::thread> th;
  th.emplace_back([&]() {                                       // plain search with a filter
    auto ctx = st.create_context();
    ctx->set_topk(topk);
    ctx->set_filter([](uint64_t key) { return key % 5 == 0; });
    std::vector<float> q = row(123);
    int loops = 0;
    while (!done.load() || loops < 3) {
      ++loops;
      TICK(1, "filtered search: search_impl");
      if (st.search_impl(q.data(), qmeta, ctx) != 0) { ++bad; continue; }
      TICK(1, "filtered search: checking");
      const auto &r = ctx->result();
      if (r.size() != topk) { ++bad; continue; }
      if (r[0].score() != 0.0f) ++bad;
      for (size_t j = 0; j < r.size(); ++j) {
        if (!written(r[j].key()) || r[j].key() % 5 == 0) ++bad;
        if (j && r[j - 1].score() > r[j].score()) ++bad;
      }
    }
  });
  th.emplace_back([&]() {                                       // p_keys naming live rows, future rows and never-written ids
    auto ctx = st.create_context();
    ctx->set_topk(4);
    std::vector<float> q = row(10);
    std::vector<std::vector<uint64_t>> pk(1);
    pk[0] = {10, 510, 1010, n1 - 1, n1 - 2, (uint64_t)n1 + 50, 11, 12};
    int loops = 0;
    while (!done.load() || loops < 3) {
      ++loops;
      TICK(2, "p_keys search: search_bf_by_p_keys_impl");
      if (st.search_bf_by_p_keys_impl(q.data(), pk, qmeta, 1, ctx) != 0) { ++bad; continue; }
      TICK(2, "p_keys search: checking");
      const auto &r = ctx->result();
      if (r.size() != 4 || r[0].score() != 0.0f) { ++bad; continue; }
      for (const auto &d : r) if (!written(d.key())) ++bad;
    }
  });
  th.emplace_back([&]() {                                       // group-by
    auto ctx = st.create_context();
    ctx->set_group_params(4, 3);
    ctx->set_group_by([](uint64_t key) { return std::to_string(key % 7); });
    std::vector<float> q = row(250);
    int loops = 0;
    while (!done.load() || loops < 3) {
      ++loops;
      TICK(3, "group-by search: search_impl");
      if (st.search_impl(q.data(), qmeta, 1, ctx) != 0) { ++bad; continue; }
      TICK(3, "group-by search: checking");
      const auto &g = ctx->group_result();
      if (g.size() != 4) { ++bad; continue; }
      for (const auto &grp : g) {
        if (grp.docs().size() != 3) ++bad;
        for (const auto &d : grp.docs()) if (!written(d.key()) || std::to_string(d.key() % 7) != grp.group_id()) ++bad;
      }
      if (g[0].docs().empty() || g[0].docs()[0].score() != 0.0f) ++bad;
    }
  });
  writer.join();
  for (auto &x : th) x.join();
  EXPECT(bad.load() == 0);
  EXPECT(st.count() == n1);
  return 0;
}

// patches/boundary_a.diff: boundary A's nprobe arrives as scan_ratio = nprobe / nlist (ivf_searcher_context.h:61-79);
// with brute_force_threshold = N - 1 the operator probes exactly nprobe lists.  The operator-level search
// must equal the C ABI called with those two numbers.
static int TestBoundaryAMapping() {
  const uint32_t dim = 8, nlist = 200, per_list = 30, n = nlist * per_list, nprobe = 7, topk = 5, nq = 20;
  IndexMeta meta(IndexMeta::DT_FP32, dim);
  meta.set_metric("SquaredEuclidean");
  std::vector<float> base((size_t)n * dim), cent((size_t)nlist * dim);
  std::vector<uint64_t> offs(nlist + 1);
  uint32_t seed = 99;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)((seed >> 12) & 255) / 32.0f; };
  for (uint32_t l = 0; l < nlist; ++l) {
    offs[l] = (uint64_t)l * per_list;
    for (uint32_t j = 0; j < dim; ++j) cent[(size_t)l * dim + j] = (float)(l * 10);
    for (uint32_t i = 0; i < per_list; ++i)
      for (uint32_t j = 0; j < dim; ++j) base[((size_t)l * per_list + i) * dim + j] = (float)(l * 10) + rnd();
  }
  offs[nlist] = n;
  Params p;
  p.set(PARAM_IVF_SEARCHER_SCAN_RATIO, (double)((float)nprobe / (float)nlist));
  p.set(PARAM_IVF_SEARCHER_BRUTE_FORCE_THRESHOLD, (double)(n - 1));
  HipIVFSearcher se;
  ASSERT(0 == se.init(p));
  ASSERT(0 == se.load(meta, cent.data(), nlist, offs.data(), base.data(), nullptr));
  EXPECT(se.nprobe() == nprobe);
  EXPECT(se.max_scan_count() == n - 1);
  std::vector<float> q((size_t)nq * dim);
  for (uint32_t i = 0; i < nq; ++i)
    for (uint32_t j = 0; j < dim; ++j) q[(size_t)i * dim + j] = (float)((i * 9) % nlist * 10) + rnd();
  IndexQueryMeta qmeta(IndexMeta::DT_FP32, dim);
  auto ctx = se.create_context();
  ctx->set_topk(topk);
  ASSERT(0 == se.search_impl(q.data(), qmeta, nq, ctx));
  // the same search straight through the C ABI with (nprobe, N - 1), plus the per-query probe statistics
  zvec_hip_ivf_t raw = nullptr;
  ASSERT(0 == zvec_hip_ivf_create(dim, ZVEC_HIP_DT_FP32, ZVEC_HIP_METRIC_L2, 0, &raw));
  ASSERT(0 == zvec_hip_ivf_load(raw, cent.data(), nlist, offs.data(), base.data(), nullptr));
  zvec_hip_ctx_t rc_ctx = nullptr;
  ASSERT(0 == zvec_hip_ctx_create(0, &rc_ctx));
  std::vector<uint64_t> keys((size_t)nq * topk);
  std::vector<float> scores((size_t)nq * topk);
  std::vector<uint32_t> counts(nq), scanned(nq), probes(nq);
  ASSERT(0 == zvec_hip_ivf_search(raw, rc_ctx, q.data(), nq, topk, FLT_MAX, nprobe, n - 1, nullptr, keys.data(), scores.data(), counts.data()));
  ASSERT(0 == zvec_hip_ivf_last_stats(raw, rc_ctx, nq, scanned.data(), probes.data()));
  for (uint32_t i = 0; i < nq; ++i) {
    EXPECT(probes[i] == nprobe);
    EXPECT(scanned[i] == nprobe * per_list);
    const IndexDocumentList &r = ctx->result(i);
    ASSERT(r.size() == counts[i]);
    for (size_t j = 0; j < r.size(); ++j) {
      EXPECT(r[j].key() == keys[(size_t)i * topk + j]);
      EXPECT(r[j].score() == scores[(size_t)i * topk + j]);
    }
  }
  zvec_hip_ctx_destroy(rc_ctx);
  zvec_hip_ivf_destroy(raw);
  return 0;
}

// Expectations of flat_streamer_test.cc TestGroup: 5000 rows (row i = i/10 everywhere), query at 250.1; 5 groups of up
// to 20 documents without any set_topk; group ids are strings made from the key.  The full scan must give non-empty
// groups — here additionally: the query's own decade first, its best document key 2501, every document in its group;
// the p_keys leg over {4,3,2,1,5..10} with one key per group must list the keys 10, 9, 8, 7, 6 first.
static int TestGroup() {
  constexpr size_t dim = 16;
  IndexMeta meta(IndexMeta::DT_FP32, dim);
  meta.set_metric("SquaredEuclidean");
  HipFlatStreamer streamer;
  ASSERT(0 == streamer.init(meta, Params()));
  ASSERT(0 == streamer.open());
  const size_t rows = 5000;
  std::vector<float> all(rows * dim);
  std::vector<uint64_t> keys(rows);
  for (size_t i = 0; i < rows; ++i) {
    keys[i] = i;
    for (size_t j = 0; j < dim; ++j) all[i * dim + j] = i / 10.0f;
  }
  ASSERT(0 == streamer.add_batch(all.data(), rows, keys.data()));
  IndexQueryMeta qmeta(IndexMeta::DT_FP32, dim);
  std::vector<float> query(dim, (rows / 2) * 1.0f / 10 + 0.1f);
  const uint32_t group_num = 5, group_topk = 20;

  auto ctx = streamer.create_context();
  ASSERT(!!ctx);
  ctx->set_group_params(group_num, group_topk);
  EXPECT(IndexError_InvalidArgument == streamer.search_impl(query.data(), qmeta, 1, ctx));   // no group-by function yet
  ctx->set_group_by([](uint64_t key) { return std::string("g_") + std::to_string(key / 10 % 10); });
  ASSERT(0 == streamer.search_impl(query.data(), qmeta, 1, ctx));
  auto &by_decade = ctx->group_result();
  ASSERT(by_decade.size() == group_num);
  EXPECT(by_decade[0].group_id() == "g_0");
  EXPECT(by_decade[0].docs().size() == group_topk && by_decade[0].docs()[0].key() == 2501);
  for (auto &g : by_decade) {
    EXPECT(g.docs().size() > 0);
    for (size_t j = 0; j < g.docs().size(); ++j) {
      EXPECT(std::string("g_") + std::to_string(g.docs()[j].key() / 10 % 10) == g.group_id());
      if (j) EXPECT(g.docs()[j - 1].score() <= g.docs()[j].score());
    }
  }
  for (size_t i = 1; i < by_decade.size(); ++i) EXPECT(by_decade[i - 1].docs()[0].score() <= by_decade[i].docs()[0].score());

  auto pk = streamer.create_context();
  pk->set_group_params(group_num, group_topk);
  pk->set_group_by([](uint64_t key) { return std::string("g_") + std::to_string(key % 10); });
  std::vector<std::vector<uint64_t>> p_keys(1);
  p_keys[0] = {4, 3, 2, 1, 5, 6, 7, 8, 9, 10};
  ASSERT(0 == streamer.search_bf_by_p_keys_impl(query.data(), p_keys, qmeta, 1, pk));
  auto &by_unit = pk->group_result();
  ASSERT(by_unit.size() == group_num);
  for (uint32_t i = 0; i < by_unit.size(); ++i) {
    ASSERT(by_unit[i].docs().size() > 0);
    EXPECT(10 - i == by_unit[i].docs()[0].key());
  }
  // ungrouped p_keys search through the same entry: the three closest of the listed keys
  auto plain = streamer.create_context();
  plain->set_topk(3);
  ASSERT(0 == streamer.search_bf_by_p_keys_impl(query.data(), p_keys, qmeta, 1, plain));
  ASSERT(plain->result().size() == 3);
  EXPECT(plain->result()[0].key() == 10 && plain->result()[1].key() == 9 && plain->result()[2].key() == 8);
  return 0;
}

// Expectations of flat_streamer_test.cc TestAddAndSearchWithID: add_with_id_impl with the even ids first (odd positions
// become holes), then the odd ids (they land on the holes); row i = (i, ..., i); for queries i + 0.1 the linear search
// returns topk = 200 documents with key i first.  Here additionally: while the holes exist no odd key comes back, and a
// filter composes with them.
static int TestAddAndSearchWithID() {
  constexpr size_t dim = 16;
  IndexMeta meta(IndexMeta::DT_FP32, dim);
  meta.set_metric("SquaredEuclidean");
  HipFlatStreamer streamer;
  ASSERT(0 == streamer.init(meta, Params()));
  ASSERT(0 == streamer.open());
  auto ctx = streamer.create_context();
  ASSERT(!!ctx);
  const size_t cnt = 4000;
  IndexQueryMeta qmeta(IndexMeta::DT_FP32, dim);
  for (size_t i = 0; i < cnt; i += 2) {
    std::vector<float> vec(dim, (float)i);
    ASSERT(0 == streamer.add_with_id_impl((uint32_t)i, vec.data(), qmeta, ctx));
  }
  const size_t topk = 200;
  ctx->set_topk(topk);
  for (size_t i = 0; i < cnt; i += 500) {
    std::vector<float> vec(dim, i + 0.1f);
    ASSERT(0 == streamer.search_impl(vec.data(), qmeta, ctx));
    auto &res = ctx->result();
    ASSERT(res.size() == topk);
    EXPECT(res[0].key() == i);
    for (auto &d : res) EXPECT(d.key() % 2 == 0);
  }
  ctx->set_filter([](uint64_t key) { return key % 4 == 0; });
  {
    std::vector<float> vec(dim, 1000.1f);
    ASSERT(0 == streamer.search_impl(vec.data(), qmeta, ctx));
    for (auto &d : ctx->result()) EXPECT(d.key() % 4 == 2);
    EXPECT(ctx->result()[0].key() == 998 || ctx->result()[0].key() == 1002);
  }
  ctx->reset_filter();
  for (size_t i = 1; i < cnt; i += 2) {
    std::vector<float> vec(dim, (float)i);
    ASSERT(0 == streamer.add_with_id_impl((uint32_t)i, vec.data(), qmeta, ctx));
  }
  EXPECT(streamer.count() == cnt);
  for (size_t i = 0; i < cnt; i += 100) {
    std::vector<float> vec(dim, i + 0.1f);
    ASSERT(0 == streamer.search_bf_impl(vec.data(), qmeta, ctx));
    auto &res = ctx->result();
    ASSERT(res.size() == topk);
    EXPECT(res[0].key() == i);
    EXPECT(res[1].key() == i + 1);
  }
  return 0;
}

int main() {
  int rc = 0;
  std::thread(watchdog).detach();
  RUN(TestLinearSearch);
  RUN(TestAddAndSearchWithID);
  RUN(TestGroup);
  RUN(TestFilter);
  RUN(TestIVFSimple);
  RUN(TestNullContextFromManyThreads);
  RUN(TestConcurrentAddWithFilterAndFetch);
  RUN(TestConcurrentPutAndSearch);
  RUN(TestBoundaryAMapping);
  RUN(TestMicroBatcher);
  if (rc == 0 && g_fail == 0) { printf("host mirror: all tests passed\n"); return 0; }
  printf("host mirror: %d failures (rc=%d)\n", g_fail, rc);
  return 1;
}
