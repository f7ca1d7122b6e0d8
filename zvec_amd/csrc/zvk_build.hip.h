// zvk_build.hip.h — index ingestion and k-means helpers: dumped-segment body -> rows, centroid means, row gathers.
// Part of the device code of libzvec_hip (included through scan_kernels.hip.h).
#pragma once
#include "zvk_common.hip.h"
#include "zvk_rows.hip.h"

namespace zvk {

// ---------------------------------------------------------------------------------------------
// "ivf.inverted_body" of a dumped reference index -> plain rows in list order (SURVEY next-2).  Layout written by
// IVFDumper (src/core/algorithm/ivf/ivf_dumper.cc:19-81,388-406; ivf_dumper.h:33-160): per inverted list, at
// InvertedListMeta::offset, blocks of `bvc` (32) vectors, each block padded to 32 bytes; a FULL block of a
// column-major index is transposed in units of the element's alignment (unit u of vector i at (u*bvc + i)*unit),
// every other block is row-major.  One wave per row; pure byte movement.
// ---------------------------------------------------------------------------------------------
struct IvfBodyArgs {
  const uint8_t *body;
  const uint64_t *list_off;     // [nlist] byte offset of each list in the body
  const uint64_t *list_row0;    // [nlist + 1] first global row (InvertedListMeta::id_offset), last = total
  uint32_t nlist;
  uint32_t bvc;                 // block_vector_count
  uint32_t block_size;          // bytes of a full block
  uint32_t elem_size;           // bytes per vector
  uint32_t unit;                // alignment unit of the element type (2 = fp16, 4 = fp32)
  uint32_t column_major;
  uint8_t *rows;                // out: [total][elem_size]
  uint64_t total;
};

__global__ void __launch_bounds__(256) ivf_body_rows_kernel(const IvfBodyArgs a) {
  const int lane = threadIdx.x & 63;
  const uint64_t g = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= a.total) return;
  uint32_t lo = 0, hi = a.nlist;          // last list with row0 <= g
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a.list_row0[mid] <= g) lo = mid; else hi = mid;
  }
  while (lo + 1 < a.nlist && a.list_row0[lo + 1] <= g) ++lo;    // (empty lists share a row0)
  const uint64_t i = g - a.list_row0[lo];
  const uint64_t cnt = a.list_row0[lo + 1] - a.list_row0[lo];
  const uint64_t blk = i / a.bvc, r = i % a.bvc;
  const bool full = (blk + 1) * a.bvc <= cnt;
  const uint8_t *b0 = a.body + a.list_off[lo] + blk * a.block_size;
  uint8_t *dst = a.rows + g * a.elem_size;
  const uint32_t units = a.elem_size / a.unit;
  if (a.column_major && full) {
    if (a.unit == 4) {
      for (uint32_t u = lane; u < units; u += 64)
        reinterpret_cast<uint32_t *>(dst)[u] = reinterpret_cast<const uint32_t *>(b0)[(size_t)u * a.bvc + r];
    } else {
      for (uint32_t u = lane; u < units; u += 64)
        reinterpret_cast<uint16_t *>(dst)[u] = reinterpret_cast<const uint16_t *>(b0)[(size_t)u * a.bvc + r];
    }
  } else {
    const uint8_t *src = b0 + r * a.elem_size;
    if (a.unit == 4) {
      for (uint32_t u = lane; u < units; u += 64) reinterpret_cast<uint32_t *>(dst)[u] = reinterpret_cast<const uint32_t *>(src)[u];
    } else {
      for (uint32_t u = lane; u < units; u += 64) reinterpret_cast<uint16_t *>(dst)[u] = reinterpret_cast<const uint16_t *>(src)[u];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k-means helpers (IVF build): mean of member rows per cluster, members given as CSR of row ids.
// ---------------------------------------------------------------------------------------------
template <bool F16>
__global__ void __launch_bounds__(256) centroid_mean_kernel(const void *rows, uint32_t dim,
                                                            const uint64_t *member_off,
                                                            const uint64_t *members, void *centroids) {
  const uint32_t c = blockIdx.x;
  const uint64_t b = member_off[c], e = member_off[c + 1];
  if (e == b) return;  // empty cluster keeps its previous centroid
  const float inv = 1.0f / (float)(e - b);
  for (uint32_t col = threadIdx.x; col < dim; col += blockDim.x) {
    float acc = 0.f;
    for (uint64_t m = b; m < e; ++m) acc += load_row_elem<F16>(rows, members[m], dim, col);
    if constexpr (F16) reinterpret_cast<_Float16 *>(centroids)[(size_t)c * dim + col] = (_Float16)(acc * inv);   // RNE
    else reinterpret_cast<float *>(centroids)[(size_t)c * dim + col] = acc * inv;
  }
}

// row gather in bytes (element-type agnostic)
__global__ void gather_rows_kernel(const void *rows, uint32_t row_bytes, const uint64_t *ids, uint64_t n, void *out) {
  uint64_t i = blockIdx.x;
  if (i >= n) return;
  const uint16_t *src = reinterpret_cast<const uint16_t *>(rows) + (size_t)ids[i] * (row_bytes / 2);
  uint16_t *dst = reinterpret_cast<uint16_t *>(out) + (size_t)i * (row_bytes / 2);
  for (uint32_t c = threadIdx.x; c < row_bytes / 2; c += blockDim.x) dst[c] = src[c];
}

}  // namespace zvk
