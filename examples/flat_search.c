/* examples/flat_search.c — the drop-in boundary used from plain C: create a flat index, add documents the way the product does
 * (one add-with-id call per document), search a small batch, print the results.
 *   gcc -std=c99 -Iinclude -o flat_search examples/flat_search.c -Lzvec_amd -lzvec_hip -Wl,-rpath,$PWD/zvec_amd
 * Needs an MI355X at run time (there is no CPU fallback: zvec_hip_flat_create fails without a HIP device). */
#include <stdio.h>
#include <stdlib.h>

#include "zvec_hip.h"

int main(void) {
  enum { DIM = 64, N = 1000, NQ = 3, K = 5 };
  zvec_hip_flat_t index = NULL;
  int rc = zvec_hip_flat_create(DIM, ZVEC_HIP_DT_FP32, ZVEC_HIP_METRIC_L2, 0, &index);
  if (rc != 0) {
    fprintf(stderr, "zvec_hip_flat_create: %d (%s)\n", rc, zvec_hip_error_string(rc));
    return 1;
  }
  float *row = (float *)malloc(sizeof(float) * DIM);
  for (uint32_t id = 0; id < N; ++id) {                 /* document id: every component = id */
    for (int j = 0; j < DIM; ++j) row[j] = (float)id;
    if ((rc = zvec_hip_flat_put(index, &id, 1, row, NULL)) != 0) return 2;
  }
  float queries[NQ * DIM];
  for (int q = 0; q < NQ; ++q)
    for (int j = 0; j < DIM; ++j) queries[q * DIM + j] = 100.0f * (float)(q + 1) + 0.25f;
  uint64_t keys[NQ * K];
  float scores[NQ * K];
  uint32_t counts[NQ];
  rc = zvec_hip_flat_search(index, NULL, queries, NQ, K, 3.4e38f, NULL, keys, scores, counts);
  if (rc != 0) return 3;
  for (int q = 0; q < NQ; ++q) {
    printf("query %d:", q);
    for (uint32_t j = 0; j < counts[q]; ++j) printf(" (%llu, %.3f)", (unsigned long long)keys[q * K + j], scores[q * K + j]);
    printf("\n");
    if (keys[q * K] != (uint64_t)(100 * (q + 1))) return 4;      /* the nearest document is 100 (q + 1) */
  }
  free(row);
  zvec_hip_flat_destroy(index);
  return 0;
}
