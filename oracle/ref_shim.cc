// ref_shim.cc — thin extern "C" door onto the REFERENCE's own arithmetic kernels and heap, compiled
// in place from /root/reference by oracle/Makefile into oracle/_ref/libzvec_ref.so.
// TEST INFRASTRUCTURE ONLY.  This file is ours; it includes the reference's headers by path and
// copies none of its source.  Reference entry points used:
//   ailego::SquaredEuclideanDistanceMatrix<float,1,1>::Compute  src/ailego/math/euclidean_distance_matrix_fp32.cc:287
//   ailego::InnerProductMatrix<float,1,1>::Compute              src/ailego/math/inner_product_matrix_fp32.cc:588
//   ailego::MinusInnerProductMatrix<float,1,1>::Compute         src/ailego/math/inner_product_matrix_fp32.cc:870
//   ailego::CosineDistanceMatrix<float,1,1>::Compute            src/ailego/math/cosine_distance_matrix.h:32
//   ailego::Norm2Matrix<float,1>::Compute                       src/ailego/math/norm2_matrix_fp32.cc:47
//   ailego::Normalizer<float>::L2                               src/ailego/math/normalizer.h:46
//   ailego::Heap<T>                                             src/include/zvec/ailego/container/heap.h
//   ailego::SquaredEuclideanDistanceMatrix<Float16,1,1>         src/ailego/math/euclidean_distance_matrix_fp16.cc:137
//   ailego::MinusInnerProductMatrix<Float16,1,1>                src/ailego/math/inner_product_matrix_fp16.cc:166
//   ailego::FloatHelper::ToFP16                                 src/ailego/utility/float_helper.cc
//   ailego::{SquaredEuclideanDistance,MinusInnerProduct}Matrix<float,M,N>  (block kernels, M in 2..32, N in 1..32)
//   ailego::BaseDistance<CosineDistanceMatrix, float / Float16, 12, 2>::ComputeBatch   src/ailego/math_batch/distance_batch.h:29-49
//                                                                (what CosineMetric::batch_distance returns, cosine_metric.cc:202-212)
#include <cstddef>
#include <cstdint>
#include <limits>
#include <ailego/math/cosine_distance_matrix.h>
#include <ailego/math/euclidean_distance_matrix.h>
#include <ailego/math/inner_product_matrix.h>
#include <ailego/math/norm2_matrix.h>
#include <ailego/math/normalizer.h>
#include <ailego/math_batch/distance_batch.h>
#include <zvec/ailego/container/heap.h>
#include <zvec/ailego/utility/float_helper.h>

using namespace zvec::ailego;

namespace {
// Same ordering rule as core::IndexDocument (index_document.h:143: operator< on score only) and the
// same RNN gate as IndexDocumentHeap::emplace (index_document.h:250-261).
struct Doc {
  uint64_t key;
  float score;
  uint32_t index;
  Doc() : key(0), score(0), index(0) {}
  Doc(uint64_t k, float s, uint32_t i) : key(k), score(s), index(i) {}
  bool operator<(const Doc &rhs) const { return score < rhs.score; }
};
}  // namespace

extern "C" {

float zref_sqeuclid_f32(const float *m, const float *q, size_t dim) {
  float out;
  SquaredEuclideanDistanceMatrix<float, 1, 1>::Compute(m, q, dim, &out);
  return out;
}
float zref_ip_f32(const float *m, const float *q, size_t dim) {
  float out;
  InnerProductMatrix<float, 1, 1>::Compute(m, q, dim, &out);
  return out;
}
float zref_minus_ip_f32(const float *m, const float *q, size_t dim) {
  float out;
  MinusInnerProductMatrix<float, 1, 1>::Compute(m, q, dim, &out);
  return out;
}
float zref_cosine_f32(const float *m, const float *q, size_t dim_with_norm) {
  float out;
  CosineDistanceMatrix<float, 1, 1>::Compute(m, q, dim_with_norm, &out);
  return out;
}
float zref_norm2_f32(const float *m, size_t dim) {
  float out;
  Norm2Matrix<float, 1>::Compute(m, dim, &out);
  return out;
}
void zref_normalize_l2_f32(float *arr, size_t dim, float *norm) {
  Normalizer<float>::L2(arr, dim, norm);
}

// fp16 rows: Float16 is a 2-byte POD over uint16_t (src/include/zvec/ailego/utility/float_helper.h)
float zref_sqeuclid_f16(const uint16_t *m, const uint16_t *q, size_t dim) {
  float out;
  SquaredEuclideanDistanceMatrix<Float16, 1, 1>::Compute(reinterpret_cast<const Float16 *>(m),
                                                         reinterpret_cast<const Float16 *>(q), dim, &out);
  return out;
}
float zref_minus_ip_f16(const uint16_t *m, const uint16_t *q, size_t dim) {
  float out;
  MinusInnerProductMatrix<Float16, 1, 1>::Compute(reinterpret_cast<const Float16 *>(m),
                                                  reinterpret_cast<const Float16 *>(q), dim, &out);
  return out;
}
void zref_to_fp16(const float *in, size_t n, uint16_t *out) { FloatHelper::ToFP16(in, n, out); }

// M x N block kernels (column-major block of M vectors x N interleaved queries, out[n*M + m]):
//   SquaredEuclideanDistanceMatrix<float,M,N>::Compute   euclidean_distance_matrix_fp32.cc:323-929
//   MinusInnerProductMatrix<float,M,N>::Compute          inner_product_matrix_fp32.cc:588-1179
// Returns 0 when the (M, N) pair is one of the shapes the reference specialises (euclidean_metric.cc:25-80).
#define ZREF_BLOCK(KERNEL, M_, N_) \
  if (M == M_ && N == N_) { KERNEL<float, M_, N_>::Compute(m, q, dim, out); return 0; }
#define ZREF_BLOCK_ALL(KERNEL)                                                                        \
  ZREF_BLOCK(KERNEL, 2, 1) ZREF_BLOCK(KERNEL, 2, 2) ZREF_BLOCK(KERNEL, 4, 1) ZREF_BLOCK(KERNEL, 4, 2)  \
  ZREF_BLOCK(KERNEL, 4, 4) ZREF_BLOCK(KERNEL, 8, 1) ZREF_BLOCK(KERNEL, 8, 2) ZREF_BLOCK(KERNEL, 8, 4)  \
  ZREF_BLOCK(KERNEL, 8, 8) ZREF_BLOCK(KERNEL, 16, 1) ZREF_BLOCK(KERNEL, 16, 2) ZREF_BLOCK(KERNEL, 16, 4) \
  ZREF_BLOCK(KERNEL, 16, 8) ZREF_BLOCK(KERNEL, 16, 16) ZREF_BLOCK(KERNEL, 32, 1) ZREF_BLOCK(KERNEL, 32, 2) \
  ZREF_BLOCK(KERNEL, 32, 4) ZREF_BLOCK(KERNEL, 32, 8) ZREF_BLOCK(KERNEL, 32, 16) ZREF_BLOCK(KERNEL, 32, 32)
int zref_sqeuclid_block_f32(int M, int N, const float *m, const float *q, size_t dim, float *out) {
  ZREF_BLOCK_ALL(SquaredEuclideanDistanceMatrix)
  return -1;
}
int zref_minus_ip_block_f32(int M, int N, const float *m, const float *q, size_t dim, float *out) {
  ZREF_BLOCK_ALL(MinusInnerProductMatrix)
  return -1;
}
// fp16 blocks (euclidean_distance_matrix_fp16.cc / inner_product_matrix_fp16.cc; distance_matrix_accum_fp16.i)
#define ZREF_BLOCK16(KERNEL, M_, N_) \
  if (M == M_ && N == N_) { KERNEL<Float16, M_, N_>::Compute(reinterpret_cast<const Float16 *>(m), reinterpret_cast<const Float16 *>(q), dim, out); return 0; }
#define ZREF_BLOCK16_ALL(KERNEL)                                                                              \
  ZREF_BLOCK16(KERNEL, 2, 1) ZREF_BLOCK16(KERNEL, 2, 2) ZREF_BLOCK16(KERNEL, 4, 1) ZREF_BLOCK16(KERNEL, 4, 2)     \
  ZREF_BLOCK16(KERNEL, 4, 4)                                                                                  \
  ZREF_BLOCK16(KERNEL, 8, 1) ZREF_BLOCK16(KERNEL, 8, 2) ZREF_BLOCK16(KERNEL, 8, 4) ZREF_BLOCK16(KERNEL, 8, 8)     \
  ZREF_BLOCK16(KERNEL, 16, 1) ZREF_BLOCK16(KERNEL, 16, 2) ZREF_BLOCK16(KERNEL, 16, 4) ZREF_BLOCK16(KERNEL, 16, 8) \
  ZREF_BLOCK16(KERNEL, 16, 16) ZREF_BLOCK16(KERNEL, 32, 1) ZREF_BLOCK16(KERNEL, 32, 2) ZREF_BLOCK16(KERNEL, 32, 4) \
  ZREF_BLOCK16(KERNEL, 32, 8) ZREF_BLOCK16(KERNEL, 32, 16) ZREF_BLOCK16(KERNEL, 32, 32)
int zref_sqeuclid_block_f16(int M, int N, const uint16_t *m, const uint16_t *q, size_t dim, float *out) {
  ZREF_BLOCK16_ALL(SquaredEuclideanDistanceMatrix)
  return -1;
}
int zref_minus_ip_block_f16(int M, int N, const uint16_t *m, const uint16_t *q, size_t dim, float *out) {
  ZREF_BLOCK16_ALL(MinusInnerProductMatrix)
  return -1;
}

// Replays n emplace() calls through the reference Heap and returns the heap array as laid out.
size_t zref_heap_replay(const float *scores, size_t n, size_t limit, float threshold,
                        uint32_t *out_index, float *out_score) {
  Heap<Doc> heap;
  heap.limit(limit);
  for (size_t i = 0; i < n; ++i) {
    if (scores[i] <= threshold) heap.emplace((uint64_t)i, scores[i], (uint32_t)i);
  }
  for (size_t i = 0; i < heap.size(); ++i) {
    out_index[i] = heap[i].index;
    out_score[i] = heap[i].score;
  }
  return heap.size();
}


// one query against n rows through the Cosine metric's one-to-many function (dim counts the trailing norm slot(s))
void zref_cosine_batch_f32(const float *const *rows, const float *q, size_t n, size_t dim_with_norm, float *out) {
  BaseDistance<CosineDistanceMatrix, float, 12, 2>::ComputeBatch(const_cast<const float **>(rows), q, n, dim_with_norm, out);
}
void zref_cosine_batch_f16(const uint16_t *const *rows, const uint16_t *q, size_t n, size_t dim_with_norm, float *out) {
  BaseDistance<CosineDistanceMatrix, Float16, 12, 2>::ComputeBatch(reinterpret_cast<const Float16 **>(const_cast<const uint16_t **>(rows)),
                                                                    reinterpret_cast<const Float16 *>(q), n, dim_with_norm, out);
}

}  // extern "C"
