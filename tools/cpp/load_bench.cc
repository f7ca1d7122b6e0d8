// load_bench.cc — zvec's real call pattern against the IVF operators, measured in C++.
//
// zvec calls boundary B with ONE query per call from many threads, every thread with a context of its own
// (src/core/interface/index.cc:24-45 thread-local contexts, :605-619 count = 1; the reference's own load tool is
// tools/core/bench.cc:145-245: T threads, closed loop, QPS + latency percentiles).  This tool does the same against
//   --backend mirror   zvec_hip_host::HipIVFSearcher   (zvec_amd/csrc/host/hip_index.h, links libzvec_hip.so only)
//   --backend plugin   the REAL plugin class "HipIVFSearcher" (plugin/hip_plugin.cc) created by zvec's IndexFactory inside zvec's
//                      framework library, loaded at run time (dlopen): --framework <libzvec core .so> --plugin <libzvec_hip_plugin.so>.
//                      In this repository the only zvec framework library there is is the reference's core library compiled in place
//                      (oracle/_ref/libzvec_ref_core.so: the host application the plugin lives in, not a checker here); the tool
//                      drives the plugin through that library's by-name doors (create by registered name over an index given as
//                      arrays, create_context per thread, search_impl(count = 1), result()).
// Both backends run the same operator code (include/zvec_hip_operator.hpp).
//
// Index: `rows` x `dim` fp32, `nlist` inverted lists given directly as arrays (centre + noise rows in list order, list sizes
// spread like a k-means partition's), nprobe via scan_ratio = nprobe / nlist, brute_force_threshold = rows - 1.
// Every configuration = (threads T, micro-batcher window) runs `seconds` of closed-loop single-query searches over a pool of
// 4096 queries; every answer is compared with the answer one batched search gave for the same query.
//
//   g++ -std=c++17 -O2 -pthread -Iinclude -o tools/cpp/load_bench tools/cpp/load_bench.cc -Lzvec_amd -lzvec_hip -ldl \
//       -Wl,-rpath,'$ORIGIN/../../zvec_amd'
#include <dlfcn.h>
#include <sys/resource.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../zvec_amd/csrc/host/hip_index.h"

namespace {

using Clock = std::chrono::steady_clock;

struct Args {
  uint64_t rows = 10000000;
  uint32_t dim = 768, nlist = 4096, nprobe = 36, topk = 10, pool = 4096;
  double seconds = 3.0;
  std::string backend = "mirror", framework, plugin, json, threads = "1,16,64,256", windows = "0,2000", waits = "";
  uint32_t max_batch = 1024, linger_us = 100, half_width = 0;
  int device = 0;
};

struct Rng {          // xorshift64*: cheap, seeded per thread
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) {}
  uint64_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1Dull; }
  float uni() { return (float)((next() >> 40) * (1.0 / 16777216.0)) * 2.f - 1.f; }      // [-1, 1)
};

// the framework's doors (see the header comment)
struct Doors {
  void *lib = nullptr;
  int (*load_plugin)(const char *, char *, uint64_t) = nullptr;
  void *(*over_rows)(const char *, const char *, int, uint32_t, const char *, const void *, uint32_t, const uint64_t *, const void *,
                     const uint64_t *, int *) = nullptr;
  void *(*ctx_create)(void *) = nullptr;
  void (*ctx_destroy)(void *) = nullptr;
  void (*ctx_set_topk)(void *, uint32_t) = nullptr;
  int (*search)(void *, void *, int, const void *, int, uint32_t, uint32_t, const uint64_t *, const uint32_t *) = nullptr;
  uint32_t (*result_size)(void *, uint32_t) = nullptr;
  int (*result)(void *, uint32_t, uint64_t *, float *, uint32_t *, void *, uint32_t, uint32_t *) = nullptr;
  int (*runner_close)(void *) = nullptr;
  bool open(const std::string &path) {
    lib = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { fprintf(stderr, "dlopen %s: %s\n", path.c_str(), dlerror()); return false; }
#define DOOR(field, name) field = reinterpret_cast<decltype(field)>(dlsym(lib, name)); if (!field) { fprintf(stderr, "missing %s\n", name); return false; }
    DOOR(load_plugin, "zref_load_plugin") DOOR(over_rows, "zref_ivf_searcher_over_rows") DOOR(ctx_create, "zref_ctx_create")
    DOOR(ctx_destroy, "zref_ctx_destroy") DOOR(ctx_set_topk, "zref_ctx_set_topk") DOOR(search, "zref_search")
    DOOR(result_size, "zref_ctx_result_size") DOOR(result, "zref_ctx_result") DOOR(runner_close, "zref_runner_close")
#undef DOOR
    return true;
  }
};

// one searcher under test: backend-independent face
struct Searcher {
  virtual ~Searcher() {}
  virtual void *new_context() = 0;
  virtual void free_context(void *c) = 0;
  // one search of `count` queries; keys[count][topk] (missing entries ~0)
  virtual int search(void *c, const float *q, uint32_t count, uint64_t *keys) = 0;
  // micro-batcher counters so far (batches run, queries in them, largest batch); false when the backend cannot tell
  virtual bool batch_stats(uint64_t *batches, uint64_t *queries, uint64_t *largest) { (void)batches; (void)queries; (void)largest; return false; }
};

struct MirrorSearcher : Searcher {
  zvec_hip_host::HipIVFSearcher s;
  zvec_hip_host::IndexQueryMeta qm;
  uint32_t topk;
  void *new_context() override {
    auto c = s.create_context();
    if (!c) return nullptr;
    c->set_topk(topk);
    return new zvec_hip_host::Context::Pointer(std::move(c));
  }
  void free_context(void *c) override { delete static_cast<zvec_hip_host::Context::Pointer *>(c); }
  bool batch_stats(uint64_t *batches, uint64_t *queries, uint64_t *largest) override {
    const auto st = s.batcher_stats();
    *batches = st.batches; *queries = st.queries; *largest = st.largest;
    return true;
  }
  int search(void *c, const float *q, uint32_t count, uint64_t *keys) override {
    auto &ctx = *static_cast<zvec_hip_host::Context::Pointer *>(c);
    int rc = s.search_impl(q, qm, count, ctx);
    if (rc != 0) return rc;
    for (uint32_t i = 0; i < count; ++i) {
      const auto &r = ctx->result(i);
      for (uint32_t j = 0; j < topk; ++j) keys[(size_t)i * topk + j] = j < r.size() ? r[j].key() : ~0ull;
    }
    return 0;
  }
};

struct PluginSearcher : Searcher {
  Doors *d = nullptr;
  void *h = nullptr;
  uint32_t dim, topk;
  ~PluginSearcher() override { if (h) d->runner_close(h); }
  void *new_context() override {
    void *c = d->ctx_create(h);
    if (c) d->ctx_set_topk(c, topk);
    return c;
  }
  void free_context(void *c) override { d->ctx_destroy(c); }
  int search(void *c, const float *q, uint32_t count, uint64_t *keys) override {
    int rc = d->search(h, c, 0, q, 0, dim, count, nullptr, nullptr);
    if (rc != 0) return rc;
    float sc[1024];
    for (uint32_t i = 0; i < count; ++i) {
      uint32_t n = std::min<uint32_t>(d->result_size(c, i), topk);
      uint64_t *k = keys + (size_t)i * topk;
      uint64_t tmp[1024];
      d->result(c, i, tmp, sc, nullptr, nullptr, 0, nullptr);
      for (uint32_t j = 0; j < topk; ++j) k[j] = j < n ? tmp[j] : ~0ull;
    }
    return 0;
  }
};

struct RunResult {
  uint32_t threads = 0, window_us = 0;
  int wait = -1;
  uint64_t calls = 0, mismatched = 0;
  double seconds = 0, qps = 0, p50_us = 0, p90_us = 0, p99_us = 0, mean_us = 0, cpus_busy = 0, mean_batch = 0;
  uint64_t largest_batch = 0;
  int rc = 0;
};

RunResult run_load(Searcher *s, const std::vector<float> &pool, const std::vector<uint64_t> &expect, const Args &a, uint32_t threads) {
  RunResult r;
  r.threads = threads;
  std::atomic<bool> go{false}, stop{false};
  std::atomic<int> ready{0}, err{0};
  std::vector<std::vector<uint32_t>> lat(threads);
  std::vector<uint64_t> bad(threads, 0);
  std::vector<std::thread> th;
  for (uint32_t t = 0; t < threads; ++t)
    th.emplace_back([&, t]() {
      void *c = s->new_context();                    // a context per thread, as index.cc:24-45
      if (!c) { err = -1; ready++; return; }
      std::vector<uint64_t> keys(a.topk);
      Rng rng(1000 + t);
      // one untimed search: the context's workspace is allocated on first use
      if (int rc = s->search(c, pool.data(), 1, keys.data())) err = rc;
      ready++;
      while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
      lat[t].reserve(1 << 16);
      while (!stop.load(std::memory_order_relaxed) && err.load(std::memory_order_relaxed) == 0) {
        const uint32_t qi = (uint32_t)(rng.next() % a.pool);
        const auto t0 = Clock::now();
        int rc = s->search(c, pool.data() + (size_t)qi * a.dim, 1, keys.data());
        const auto t1 = Clock::now();
        if (rc != 0) { err = rc; break; }
        lat[t].push_back((uint32_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count());
        if (memcmp(keys.data(), expect.data() + (size_t)qi * a.topk, a.topk * 8) != 0) {
          // same id SET in another order (equal scores) is not a mismatch
          std::vector<uint64_t> x(keys), y(expect.begin() + (size_t)qi * a.topk, expect.begin() + (size_t)(qi + 1) * a.topk);
          std::sort(x.begin(), x.end());
          std::sort(y.begin(), y.end());
          if (x != y) bad[t]++;
        }
      }
      s->free_context(c);
    });
  while (ready.load() < (int)threads) std::this_thread::sleep_for(std::chrono::milliseconds(1));
  auto cpu_now = []() {
    struct rusage u;
    getrusage(RUSAGE_SELF, &u);
    return u.ru_utime.tv_sec + u.ru_stime.tv_sec + 1e-6 * (u.ru_utime.tv_usec + u.ru_stime.tv_usec);
  };
  const double cpu0 = cpu_now();
  uint64_t b0 = 0, q0 = 0, l0 = 0;
  const bool have_stats = s->batch_stats(&b0, &q0, &l0);
  const auto t0 = Clock::now();
  go.store(true, std::memory_order_release);
  std::this_thread::sleep_for(std::chrono::duration<double>(a.seconds));
  stop = true;
  for (auto &t : th) t.join();
  r.seconds = std::chrono::duration<double>(Clock::now() - t0).count();
  r.cpus_busy = (cpu_now() - cpu0) / r.seconds;             // process CPU time per wall second: what the callers cost the host
  r.rc = err.load();
  if (have_stats) {
    uint64_t b1 = 0, q1 = 0, l1 = 0;
    s->batch_stats(&b1, &q1, &l1);
    r.mean_batch = b1 > b0 ? (double)(q1 - q0) / (double)(b1 - b0) : 0.0;
    r.largest_batch = l1;
  }
  std::vector<uint32_t> all;
  for (uint32_t t = 0; t < threads; ++t) {
    all.insert(all.end(), lat[t].begin(), lat[t].end());
    r.mismatched += bad[t];
  }
  r.calls = all.size();
  if (!all.empty()) {
    std::sort(all.begin(), all.end());
    double sum = 0;
    for (uint32_t v : all) sum += v;
    r.mean_us = sum / all.size() / 1e3;
    r.p50_us = all[all.size() / 2] / 1e3;
    r.p90_us = all[(size_t)(all.size() * 0.90)] / 1e3;
    r.p99_us = all[std::min(all.size() - 1, (size_t)(all.size() * 0.99))] / 1e3;
    r.qps = all.size() / r.seconds;
  }
  return r;
}

std::vector<uint32_t> parse_list(const std::string &s) {
  std::vector<uint32_t> v;
  size_t i = 0;
  while (i < s.size()) {
    size_t j = s.find(',', i);
    if (j == std::string::npos) j = s.size();
    if (j > i) v.push_back((uint32_t)strtoul(s.substr(i, j - i).c_str(), nullptr, 10));
    i = j + 1;
  }
  return v;
}

}  // namespace

int main(int argc, char **argv) {
  Args a;
  for (int i = 1; i + 1 < argc; i += 2) {
    const std::string k = argv[i], v = argv[i + 1];
    if (k == "--rows") a.rows = strtoull(v.c_str(), nullptr, 10);
    else if (k == "--dim") a.dim = atoi(v.c_str());
    else if (k == "--nlist") a.nlist = atoi(v.c_str());
    else if (k == "--nprobe") a.nprobe = atoi(v.c_str());
    else if (k == "--topk") a.topk = atoi(v.c_str());
    else if (k == "--seconds") a.seconds = atof(v.c_str());
    else if (k == "--backend") a.backend = v;
    else if (k == "--framework") a.framework = v;
    else if (k == "--plugin") a.plugin = v;
    else if (k == "--threads") a.threads = v;
    else if (k == "--windows") a.windows = v;
    else if (k == "--waits") a.waits = v;
    else if (k == "--max-batch") a.max_batch = atoi(v.c_str());
    else if (k == "--linger-us") a.linger_us = atoi(v.c_str());
    else if (k == "--half-width") a.half_width = atoi(v.c_str());     // proxima.hip.searcher.half_width_preselect
    else if (k == "--json") a.json = v;
    else if (k == "--device") a.device = atoi(v.c_str());
    else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
  }
  if (a.topk > 1024 || a.dim % 8 != 0) { fprintf(stderr, "topk <= 1024, dim a multiple of 8\n"); return 2; }
  const unsigned hw = std::thread::hardware_concurrency();
  printf("load_bench: %llu x %u fp32, nlist %u, nprobe %u, topk %u, backend %s, %u hardware threads\n", (unsigned long long)a.rows, a.dim,
         a.nlist, a.nprobe, a.topk, a.backend.c_str(), hw);

  // ---- the index as arrays: list sizes like a k-means partition's (mean rows / nlist, spread ~ +-45 %), rows = centre + noise
  std::vector<uint64_t> offs(a.nlist + 1, 0);
  {
    Rng rng(7);
    std::vector<double> w(a.nlist);
    double tot = 0;
    for (auto &x : w) { x = 1.0 + 0.45 * rng.uni() + 0.3 * rng.uni() * rng.uni(); tot += x; }
    uint64_t used = 0;
    for (uint32_t l = 0; l < a.nlist; ++l) {
      uint64_t n = l + 1 == a.nlist ? a.rows - used : std::min<uint64_t>(a.rows - used, (uint64_t)(a.rows * (w[l] / tot)));
      used += n;
      offs[l + 1] = used;
    }
  }
  const auto tg = Clock::now();
  std::vector<float> cent((size_t)a.nlist * a.dim), rows((size_t)a.rows * a.dim);
  {
    Rng rng(11);
    for (auto &x : cent) x = 3.f * rng.uni();
    const unsigned nt = std::max(1u, std::min(hw, 32u));
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
      th.emplace_back([&, t]() {
        Rng r(100 + t);
        for (uint32_t l = t; l < a.nlist; l += nt) {
          const float *c = &cent[(size_t)l * a.dim];
          for (uint64_t i = offs[l]; i < offs[l + 1]; ++i) {
            float *x = &rows[(size_t)i * a.dim];
            for (uint32_t d = 0; d < a.dim; ++d) x[d] = c[d] + 0.6f * (r.uni() + r.uni());
          }
        }
      });
    for (auto &t : th) t.join();
  }
  // query pool: stored rows, slightly perturbed
  std::vector<float> pool((size_t)a.pool * a.dim);
  {
    Rng rng(13);
    for (uint32_t q = 0; q < a.pool; ++q) {
      const uint64_t src = rng.next() % a.rows;
      for (uint32_t d = 0; d < a.dim; ++d) pool[(size_t)q * a.dim + d] = rows[(size_t)src * a.dim + d] + 0.05f * rng.uni();
    }
  }
  printf("corpus generated in %.1f s\n", std::chrono::duration<double>(Clock::now() - tg).count());

  Doors doors;
  if (a.backend == "plugin") {
    if (a.framework.empty() || a.plugin.empty()) { fprintf(stderr, "--backend plugin needs --framework and --plugin\n"); return 2; }
    if (!doors.open(a.framework)) return 1;
    char err[512] = {0};
    if (doors.load_plugin(a.plugin.c_str(), err, sizeof(err)) != 0) { fprintf(stderr, "plugin: %s\n", err); return 1; }
  }
  const float ratio = (float)a.nprobe / (float)a.nlist;
  auto open_searcher = [&](uint32_t window_us) -> std::unique_ptr<Searcher> {
    const auto t0 = Clock::now();
    std::unique_ptr<Searcher> out;
    if (a.backend == "plugin") {
      char params[768];
      snprintf(params, sizeof(params),
               "{\"proxima.ivf.searcher.scan_ratio\": %.9g, \"proxima.ivf.searcher.brute_force_threshold\": %llu, "
               "\"proxima.hip.device\": %d, \"proxima.hip.searcher.batch_window_us\": %u, \"proxima.hip.searcher.max_batch\": %u, "
               "\"proxima.hip.searcher.batch_linger_us\": %u, \"proxima.hip.searcher.half_width_preselect\": %u}",
               (double)ratio, (unsigned long long)(a.rows - 1), a.device, window_us, a.max_batch, a.linger_us, (unsigned)a.half_width);
      auto p = std::make_unique<PluginSearcher>();
      p->d = &doors;
      p->dim = a.dim;
      p->topk = a.topk;
      int rc = 0;
      p->h = doors.over_rows("HipIVFSearcher", params, 0, a.dim, "SquaredEuclidean", cent.data(), a.nlist, offs.data(), rows.data(),
                             nullptr, &rc);
      if (!p->h) { fprintf(stderr, "HipIVFSearcher over rows failed: %d\n", rc); return nullptr; }
      out = std::move(p);
    } else {
      auto m = std::make_unique<MirrorSearcher>();
      zvec_hip_host::Params p;
      p.set(zvec_hip_host::PARAM_IVF_SEARCHER_SCAN_RATIO, ratio);
      p.set(zvec_hip_host::PARAM_IVF_SEARCHER_BRUTE_FORCE_THRESHOLD, (double)(a.rows - 1));
      p.set(zvec_hip_host::PARAM_HIP_SEARCHER_BATCH_WINDOW_US, window_us);
      p.set(zvec_hip_host::PARAM_HIP_SEARCHER_MAX_BATCH, a.max_batch);
      p.set(zvec_hip_host::PARAM_HIP_SEARCHER_BATCH_LINGER_US, a.linger_us);
      p.set(zvec_hip_host::PARAM_HIP_SEARCHER_HALF_WIDTH_PRESELECT, a.half_width);
      if (m->s.init(p) != 0) return nullptr;
      zvec_hip_host::IndexMeta meta(zvec_hip_host::IndexMeta::DT_FP32, a.dim);
      meta.set_metric("SquaredEuclidean");
      int rc = m->s.load(meta, cent.data(), a.nlist, offs.data(), rows.data(), nullptr, a.device);
      if (rc != 0) { fprintf(stderr, "load failed: %d\n", rc); return nullptr; }
      m->qm = zvec_hip_host::IndexQueryMeta(zvec_hip_host::IndexMeta::DT_FP32, a.dim);
      m->topk = a.topk;
      out = std::move(m);
    }
    printf("index open (window %u us) in %.1f s\n", window_us, std::chrono::duration<double>(Clock::now() - t0).count());
    fflush(stdout);
    return out;
  };

  std::vector<RunResult> results;
  std::vector<uint64_t> expect;
  const auto threads = parse_list(a.threads), windows = parse_list(a.windows), waits = parse_list(a.waits);
  for (uint32_t w : windows) {
    auto s = open_searcher(w);
    if (!s) return 1;
    if (expect.empty()) {
      // the answers the timed calls are compared with: ONE batched search of the whole pool
      expect.resize((size_t)a.pool * a.topk);
      void *c = s->new_context();
      if (!c || s->search(c, pool.data(), a.pool, expect.data()) != 0) { fprintf(stderr, "reference batch failed\n"); return 1; }
      s->free_context(c);
    }
    std::vector<int> wl;
    if (w == 0 && !waits.empty()) for (uint32_t x : waits) wl.push_back((int)x); else wl.push_back(waits.empty() ? -1 : (int)waits[0]);
    for (int wp : wl) {
      if (wp >= 0) zvec_hip_set_option("wait", wp);
      for (uint32_t t : threads) {
        RunResult r = run_load(s.get(), pool, expect, a, t);
        r.window_us = w;
        zvec_hip_get_option("wait", &r.wait);
        printf("T %4u  window %5u us  wait %d : %9.0f searches/s   p50 %8.1f  p90 %8.1f  p99 %8.1f us  %5.1f CPUs busy  (%llu calls, %llu differ, rc %d)\n",
               t, w, r.wait, r.qps, r.p50_us, r.p90_us, r.p99_us, r.cpus_busy, (unsigned long long)r.calls, (unsigned long long)r.mismatched, r.rc);
        if (r.mean_batch > 0) printf("        micro-batcher: mean batch %.1f queries (largest so far %llu)\n", r.mean_batch, (unsigned long long)r.largest_batch);
        fflush(stdout);
        results.push_back(r);
        if (r.rc != 0) return 1;
      }
    }
  }
  if (!a.json.empty()) {
    FILE *f = fopen(a.json.c_str(), "w");
    if (f) {
      fprintf(f, "{\"tool\": \"tools/cpp/load_bench.cc\", \"backend\": \"%s\", \"rows\": %llu, \"dim\": %u, \"nlist\": %u, \"nprobe\": %u, \"topk\": %u, "
                 "\"seconds_per_run\": %.2f, \"max_batch\": %u, \"linger_us\": %u, \"host_threads\": %u, \"runs\": [",
              a.backend.c_str(), (unsigned long long)a.rows, a.dim, a.nlist, a.nprobe, a.topk, a.seconds, a.max_batch, a.linger_us, hw);
      for (size_t i = 0; i < results.size(); ++i) {
        const RunResult &r = results[i];
        fprintf(f, "%s\n  {\"threads\": %u, \"batch_window_us\": %u, \"wait\": %d, \"searches_per_s\": %.1f, \"p50_us\": %.1f, \"p90_us\": %.1f, "
                   "\"p99_us\": %.1f, \"mean_us\": %.1f, \"cpus_busy\": %.2f, \"mean_batch\": %.1f, \"calls\": %llu, \"answers_differing\": %llu}",
                i ? "," : "", r.threads, r.window_us, r.wait, r.qps, r.p50_us, r.p90_us, r.p99_us, r.mean_us, r.cpus_busy, r.mean_batch,
                (unsigned long long)r.calls, (unsigned long long)r.mismatched);
      }
      fprintf(f, "\n]}\n");
      fclose(f);
    }
  }
  return 0;
}
