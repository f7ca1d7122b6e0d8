// zvk_assign.hip.h — nearest-centroid assignment: the IVF build's labelling step and the assign half of every k-means round.
// Part of the device code of libzvec_hip (included through scan_kernels.hip.h).
//
// Reference: IVFBuilder::label (ivf_builder.h:253-274) asks the centroid index for the top-1 of every row
// (IVFCentroidIndex::search, ivf_centroid_index.cc:273-297): a dense rows x centroids^T contraction followed by an arg-min per
// row, ties to the lowest centroid id (heap.h:103-114 keeps the first).  At 100M x 768 fp16 over 16384 centroids that is
// 2.5e15 flop — the whole build is this kernel.
//
// Shape.  256 threads = 4 waves as 2 (row halves) x 2 (centroid halves) over a 128-row x 128-centroid tile; a wave owns a
// 64 x 64 block = 2 x 2 MFMA blocks of 32 x 32 (64 accumulators).  Compared with the search tile (8 waves, 64 x 32 per wave)
// every operand fragment read from LDS feeds two MFMAs instead of 1.33 (4 ds_read_b128 per 4 MFMAs against 3 per 2): a third
// fewer LDS operand bytes and wait points per MFMA, which counts once a 32x32x16 f16 MFMA retires in 32 cycles.  Staging is
// LDS-DMA exactly as in scan8_kernel (the centroid
// slab is stored as its LDS image; the row image's XOR swizzle is applied at the source), double-buffered, ONE barrier per
// k-step, 64 KB of LDS => two work-groups per CU.
//
// Arg-min without leaving the registers.  In the 32x32 C layout a lane holds ONE centroid column and 16 rows per MFMA block,
// so it keeps a running (best score, best centroid) per row slot — 32 slots — across the column blocks and across ALL centroid
// tiles of the sweep: per tile that is 64 fused multiply-adds and compare-selects per lane, no LDS transpose, no barrier, no
// list.  Only when a work item (128 rows x every centroid) ends are the 32 lanes that share a row reduced (5 xor-shuffles) and
// the two column halves combined through 2 KB of LDS.  Ties go to the lower centroid id at every level.
#pragma once
#include "zvk_common.hip.h"

namespace zvk {

struct AssignArgs {
  const float *base;        // centroid store, blocked rows
  const float *bnorm;       // [padded positions] squared norms (L2)
  const float *queries;     // [nq][dpad] the rows to label, prepared (padded, fp16 rows stay halves)
  const float *qnorm;       // [nq] squared norms (L2)
  uint32_t dpad, nks;
  int metric;
  uint32_t nq;              // rows
  uint32_t n;               // centroids
  uint32_t *out_label;      // [nq] nearest centroid (position in the centroid store = centroid id)
  float *out_score;         // nullable [nq] its score
};

constexpr int ASSIGN_ROWS = 128;
constexpr size_t ASSIGN_LDS = (2 * (size_t)ASSIGN_ROWS * TILE_K + 2 * (size_t)SLAB) * 4;   // 64 KiB

template <bool F16>
__global__ void __launch_bounds__(256, 2) assign_kernel(const AssignArgs a) {
  extern __shared__ f32x4 zvk_smem4[];
  float *smem = reinterpret_cast<float *>(zvk_smem4);
  float *Qs = smem;                              // [2][128 * 32]
  float *Bs = Qs + 2 * ASSIGN_ROWS * TILE_K;     // [2][SLAB]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wn = wave & 1, wm = wave >> 1;       // centroid half, row half
  const int r = lane & 31, h = lane >> 5;        // 32x32 operand coordinates
  const int srow = tid >> 3, schunk = tid & 7;   // staging: rows srow + 32 j, j = 0..3
  const int sswz = schunk ^ ((srow >> 1) & 7);
  const uint32_t dpad = a.dpad, nks = a.nks;
  const uint32_t ntiles = (a.n + TILE_N - 1) / TILE_N;
  const uint32_t nsteps = ntiles * nks;
  const uint32_t nitems = (a.nq + ASSIGN_ROWS - 1) / ASSIGN_ROWS;
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;

  const float m_alpha = (a.metric == METRIC_L2) ? -2.f : -1.f;
  const float m_beta = (a.metric == METRIC_COSINE) ? 1.f : 0.f;
  const float m_lo = (a.metric == METRIC_L2) ? 0.f : -__builtin_inff();
  const bool l2 = a.metric == METRIC_L2;

  for (uint32_t item = blockIdx.x; item < nitems; item += gridDim.x) {      // uniform exit
    const uint32_t r0 = item * ASSIGN_ROWS;
    const uint32_t nrows = min((uint32_t)ASSIGN_ROWS, a.nq - r0);
    // staging sources of this item's rows (rows past the end re-read the last one; their results are not written)
    uint32_t gq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) gq[j] = min(r0 + (uint32_t)(srow + 32 * j), a.nq - 1) * dpad + (uint32_t)sswz * 4u;
    auto stage = [&](uint32_t t_, uint32_t k_, float *Bb, float *Qb) {
      char *bl = reinterpret_cast<char *>(Bb) + wave * 1024;      // wave-uniform destinations: 1 KiB per wave-instruction
      char *ql = reinterpret_cast<char *>(Qb) + wave * 1024;
      const f32x4 *bsrc = reinterpret_cast<const f32x4 *>(a.base + (size_t)t_ * TILE_N * dpad + (size_t)k_ * SLAB) + tid;
#pragma unroll
      for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds((glb_void *)(bsrc + 256 * j), (lds_void *)(bl + 4096 * j), 16, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((glb_void *)(a.queries + (size_t)(gq[j] + k_ * TILE_K)), (lds_void *)(ql + 4096 * j), 16, 0, 0);
    };
    // row constants of the 32 row slots this lane holds: slot (mi, e) = row wm*64 + mi*32 + (e&3) + 8*(e>>2) + 4*h
    float qn[2][16];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const uint32_t row = min(r0 + (uint32_t)(wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h), a.nq - 1);
        qn[mi][e] = l2 ? a.qnorm[row] : 0.f;
      }
    float best_s[2][16];
    uint32_t best_i[2][16];
    floatx16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        best_s[mi][e] = __builtin_inff();
        best_i[mi][e] = IDX_NONE;
        acc[mi][0][e] = 0.f;
        acc[mi][1][e] = 0.f;
      }

    uint32_t tile = 0, ks = 0;          // step being computed
    uint32_t ptile = 0, pks = 0;        // next step to fetch
    auto advance = [&](uint32_t &t_, uint32_t &k_) { if (++k_ == nks) { k_ = 0; ++t_; } };
    __syncthreads();                    // the previous item's reduction scratch (operand buffers) is free again
    stage(0, 0, Bs, Qs);
    if (nsteps > 1) advance(ptile, pks);
    float bn[2] = {0.f, 0.f};
    for (uint32_t s0 = 0; s0 < nsteps; s0 += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint32_t s = s0 + u;
        if (s >= nsteps) break;                      // uniform
        const int buf = u;
        // the barrier (after each wave's vmcnt(0) for its own DMA pieces) publishes buffer `buf` and retires every read of the
        // other one, which is refilled with step s + 1 under this step's matrix work
        __syncthreads();
        if (s + 1 < nsteps) {
          stage(ptile, pks, Bs + (buf ^ 1) * SLAB, Qs + (buf ^ 1) * ASSIGN_ROWS * TILE_K);
          if (s + 2 < nsteps) advance(ptile, pks);
        }
        if (ks == nks - 1 && l2) {                   // column norms of this tile, landing under the last step's matrix work
          bn[0] = a.bnorm[(size_t)tile * TILE_N + wn * 64 + r];
          bn[1] = a.bnorm[(size_t)tile * TILE_N + wn * 64 + 32 + r];
        }
        {
          const float *Qb = Qs + buf * ASSIGN_ROWS * TILE_K + wm * 64 * TILE_K;
          const float *Bb = Bs + buf * SLAB + wn * 64 * TILE_K;
          const int swz = (r >> 1) & 7;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int c = (2 * kk + h) ^ swz;
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(Qb + (r * 8 + c) * 4);
            const f32x4 a1 = *reinterpret_cast<const f32x4 *>(Qb + ((32 + r) * 8 + c) * 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(Bb + (r * 8 + c) * 4);
            const f32x4 b1 = *reinterpret_cast<const f32x4 *>(Bb + ((32 + r) * 8 + c) * 4);
            if constexpr (F16) {
              acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b0), acc[0][0], 0, 0, 0);
              acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b1), acc[0][1], 0, 0, 0);
              acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, b0), acc[1][0], 0, 0, 0);
              acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, b1), acc[1][1], 0, 0, 0);
            } else {
#define ZVK_MFMA4(ACC, A, B)                                                   \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x, B.x, ACC, 0, 0, 0);          \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A.y, B.y, ACC, 0, 0, 0);          \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A.z, B.z, ACC, 0, 0, 0);          \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A.w, B.w, ACC, 0, 0, 0);
              ZVK_MFMA4(acc[0][0], a0, b0)
              ZVK_MFMA4(acc[0][1], a0, b1)
              ZVK_MFMA4(acc[1][0], a1, b0)
              ZVK_MFMA4(acc[1][1], a1, b1)
#undef ZVK_MFMA4
            }
          }
        }
        // ---- end of a centroid tile: fold the 64 x 64 block into the running arg-min (registers only) ----
        if (ks == nks - 1) {
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const uint32_t col = tile * TILE_N + (uint32_t)(wn * 64 + ni * 32 + r);
            const bool valid = col < a.n;                       // the last tile's padding columns never win
            const float nb = bn[ni];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int e = 0; e < 16; ++e) {
                const float dot = acc[mi][ni][e];
                float sc = fmaxf(fmaf(m_alpha, dot, l2 ? qn[mi][e] + nb : m_beta), m_lo);
                sc = valid ? sc : __builtin_inff();
                if (sc < best_s[mi][e]) {                       // strict: columns come in ascending order, the first stays
                  best_s[mi][e] = sc;
                  best_i[mi][e] = col;
                }
                acc[mi][ni][e] = 0.f;
              }
          }
        }
        advance(tile, ks);
      }
    }

    // ---- end of the item: 32 lanes share each row slot -> xor-shuffle arg-min; then the two centroid halves through LDS ----
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float s_ = best_s[mi][e];
        uint32_t i_ = best_i[mi][e];
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) {
          const float so = __shfl_xor(s_, m, 64);
          const uint32_t io = (uint32_t)__shfl_xor((int)i_, m, 64);
          if (so < s_ || (so == s_ && io < i_)) { s_ = so; i_ = io; }
        }
        best_s[mi][e] = s_;
        best_i[mi][e] = i_;
      }
    __syncthreads();                                  // every wave is done with the operand buffers
    float *red_s = Qs;                                // [2 halves][128 rows]
    uint32_t *red_i = reinterpret_cast<uint32_t *>(Qs + 2 * ASSIGN_ROWS);
    if (r == 0) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          red_s[wn * ASSIGN_ROWS + row] = best_s[mi][e];
          red_i[wn * ASSIGN_ROWS + row] = best_i[mi][e];
        }
    }
    __syncthreads();
    if ((uint32_t)tid < nrows) {
      float s_ = red_s[tid];
      uint32_t i_ = red_i[tid];
      const float s1 = red_s[ASSIGN_ROWS + tid];
      const uint32_t i1 = red_i[ASSIGN_ROWS + tid];
      if (s1 < s_ || (s1 == s_ && i1 < i_)) { s_ = s1; i_ = i1; }
      a.out_label[r0 + tid] = i_;
      if (a.out_score) a.out_score[r0 + tid] = s_;
    }
  }
}

}  // namespace zvk
