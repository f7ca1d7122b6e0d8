// test_batcher_cpu.cc — the micro-batcher of include/zvec_hip_operator.hpp without a GPU: T caller threads, one query per call
// (zvec's call pattern, index.cc:24-45,605-619), a stand-in batched search that answers every query from its own bytes.  Checks that
// every caller gets ITS answer whatever batch it rode in, that different keys never share a batch, that a lone caller is not
// delayed, that an error of the batched search reaches every member, and that batches do form under load.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#include "../../include/zvec_hip_operator.hpp"

struct Doc {
  uint64_t key;
  float score;
};
struct DT {
  using Document = Doc;
  using DocumentList = std::vector<Doc>;
  static Doc make(uint64_t k, float s) { return Doc{k, s}; }
};

static int g_fail = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++g_fail; } } while (0)

int main() {
  using namespace zvec_hip_op;
  const size_t row = 64;
  std::atomic<uint64_t> batches{0}, largest{0};
  std::atomic<int> fail_next{0};
  auto run = [&](const void *q, uint32_t n, const BatchKey &key, std::vector<uint64_t> *ks, std::vector<float> *sc, std::vector<uint32_t> *cn) {
    batches++;
    uint64_t l = largest.load();
    while (n > l && !largest.compare_exchange_weak(l, n)) {}
    if (fail_next.exchange(0)) return -31;
    std::this_thread::sleep_for(std::chrono::microseconds(300));          // a "search"
    ks->assign((size_t)n * key.topk, 0);
    sc->assign((size_t)n * key.topk, 0.f);
    cn->assign(n, 0);
    for (uint32_t i = 0; i < n; ++i) {
      uint64_t id;
      memcpy(&id, static_cast<const char *>(q) + i * row, 8);
      (*cn)[i] = key.topk;
      for (uint32_t j = 0; j < key.topk; ++j) {
        (*ks)[(size_t)i * key.topk + j] = id * 1000 + j + key.a;                // the answer names the query, the rank and the key
        (*sc)[(size_t)i * key.topk + j] = (float)j;
      }
    }
    return 0;
  };
  MicroBatcher<DT> mb(row, 64, 5000, 100, run);
  // a lone caller: answered, not delayed by the window (a batcher of its own with a 200 ms window: the bound below is far from both)
  {
    MicroBatcher<DT> lone(row, 64, 200000, 100, run);
    char q[row] = {0};
    uint64_t id = 7;
    memcpy(q, &id, 8);
    BatchKey k;
    k.topk = 3;
    std::vector<Doc> out;
    const auto t0 = std::chrono::steady_clock::now();
    CHECK(lone.search(q, k, &out) == 0);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    CHECK(out.size() == 3 && out[0].key == 7000 && out[2].key == 7002);
    CHECK(ms < 100.0);
  }
  // many callers, two different keys mixed: every answer is the caller's own, with its own key's parameters
  const int T = 48, rounds = 200;
  std::atomic<uint64_t> wrong{0};
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t]() {
      std::vector<Doc> out;
      for (int r = 0; r < rounds; ++r) {
        char q[row] = {0};
        const uint64_t id = (uint64_t)t * 100000 + r;
        memcpy(q, &id, 8);
        BatchKey k;
        k.topk = (t % 3 == 0) ? 5 : 10;
        k.a = (t % 3 == 0) ? 1 : 0;
        if (mb.search(q, k, &out) != 0) { wrong++; continue; }
        if (out.size() != k.topk) { wrong++; continue; }
        for (uint32_t j = 0; j < k.topk; ++j)
          if (out[j].key != id * 1000 + j + k.a || out[j].score != (float)j) { wrong++; break; }
      }
    });
  for (auto &x : th) x.join();
  CHECK(wrong.load() == 0);
  const auto st = mb.stats();
  CHECK(st.queries == (uint64_t)T * rounds);
  CHECK(st.batches < st.queries / 2);                       // batches formed (48 callers behind a 300 us search)
  CHECK(largest.load() > 8 && largest.load() <= 64);
  printf("batches %llu for %llu queries, largest %llu\n", (unsigned long long)st.batches, (unsigned long long)st.queries,
         (unsigned long long)largest.load());
  // an error of the batched search reaches every member of that batch and nobody else
  {
    std::atomic<int> errs{0}, oks{0};
    fail_next = 1;
    std::vector<std::thread> t2;
    for (int t = 0; t < 8; ++t)
      t2.emplace_back([&, t]() {
        char q[row] = {0};
        uint64_t id = 900 + t;
        memcpy(q, &id, 8);
        BatchKey k;
        k.topk = 2;
        std::vector<Doc> out;
        int rc = mb.search(q, k, &out);
        if (rc == -31) errs++;
        else if (rc == 0 && out.size() == 2 && out[0].key == id * 1000) oks++;
      });
    for (auto &x : t2) x.join();
    CHECK(errs.load() >= 1 && errs.load() + oks.load() == 8);
  }
  // probe parameters: IVFSearcherContext::update arithmetic (ivf_searcher_context.h:70-78)
  CHECK(probe_params(1024, 50000, 0.1f, 1000).nprobe == 102 && probe_params(1024, 50000, 0.1f, 1000).max_scan == 5000);
  CHECK(probe_params(4, 100, 0.1f, 1000).nprobe == 1 && probe_params(4, 100, 0.1f, 1000).max_scan == 1000);
  // key directory: lazily built map, holes, overwrites
  {
    KeyDirectory d;
    for (uint64_t i = 0; i < 10; ++i) d.append(100 + i);
    uint64_t pos = 99;
    CHECK(d.find(105, &pos) && pos == 5);
    d.append(500);
    CHECK(d.find(500, &pos) && pos == 10);
    d.set(14, 14);                                          // positions 11..13 become holes
    CHECK(d.size() == 15 && d.at(12) == kInvalidKey && d.find(14, &pos) && pos == 14 && !d.find(kInvalidKey, &pos));
    d.set(5, 777);                                          // overwrite: the old key is gone
    CHECK(!d.find(105, &pos) && d.find(777, &pos) && pos == 5);
  }
  printf(g_fail ? "FAILED (%d)\n" : "ok\n", g_fail);
  return g_fail ? 1 : 0;
}
