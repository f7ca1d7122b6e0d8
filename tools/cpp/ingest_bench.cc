// single-document ingest through the C ABI, timed from C++ (no interpreter in the loop)
//   g++ -O2 -std=c++17 -Iinclude -o /tmp/ingest_bench tools/cpp/ingest_bench.cc -Lzvec_amd -lzvec_hip -Wl,-rpath,$PWD/zvec_amd
#include <chrono>
#include <cstdio>
#include <vector>

#include "zvec_hip.h"

int main() {
  const uint32_t dim = 768, n = 50000;
  std::vector<float> rows((size_t)1024 * dim, 0.5f);
  for (int mode = 0; mode < 2; ++mode) {
    zvec_hip_flat_t h = nullptr;
    if (zvec_hip_flat_create(dim, ZVEC_HIP_DT_FP32, ZVEC_HIP_METRIC_L2, 0, &h) != 0) return 1;
    zvec_hip_flat_reserve(h, n + 1024);
    auto t0 = std::chrono::steady_clock::now();
    for (uint32_t i = 0; i < n; ++i) {
      const float *r = &rows[(size_t)(i & 1023) * dim];
      const uint64_t key = i;
      int rc = mode == 0 ? zvec_hip_flat_append(h, r, 1, &key) : zvec_hip_flat_put(h, &i, 1, r, nullptr);
      if (rc != 0) return 2;
    }
    uint64_t cnt = 0;
    zvec_hip_flat_count(h, &cnt);
    float q[768] = {0};
    uint64_t k;
    float s;
    uint32_t c;
    zvec_hip_flat_search(h, nullptr, q, 1, 1, 3.4e38f, nullptr, &k, &s, &c);      // waits for everything
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%s: %.2f us per document (%.0f documents/s), %llu rows\n", mode == 0 ? "append x1" : "put x1", dt / n * 1e6, n / dt,
           (unsigned long long)cnt);
    zvec_hip_flat_destroy(h);
  }
  return 0;
}
