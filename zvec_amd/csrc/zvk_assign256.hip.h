// zvk_assign256.hip.h — nearest-centroid assignment of fp16 rows on a 256 x 256 multi-phase tile.
// Part of the device code of libzvec_hip (included through scan_kernels.hip.h).
//
// Same job and same answers as assign_kernel<true> (zvk_assign.hip.h; IVFBuilder::label, ivf_builder.h:253-274: the top-1 of
// every row over the centroid index, ties to the lowest centroid id, heap.h:103-114) — the f16 contraction of
// src/ailego/math/inner_product_matrix_fp16.cc / distance_matrix_accum_fp16.i:554-594 (halves multiplied exactly, fp32
// accumulation).  The 128 x 128 tile with ONE barrier per k-step stalls on the `vmcnt(0)` in front of every barrier while its
// LDS-DMA is in flight and reads every fragment right before the MFMAs that need it: 0.31 of the dense-f16 peak, matrix pipe
// 43 % issue-stalled (profiles/r3_label_pmc.json).  This kernel is the structure cdna_hip_programming.md gives for getting past
// that ("The 256^2 8-phase template"): one work-group of 8 waves per CU, a 256 x 256 tile, k-steps of 64 halves cut into FOUR
// phases of 16 MFMAs, each phase = { fragment reads of ONE quadrant's new operands · one quarter of a LATER k-step's LDS-DMA ·
// a COUNTED `s_waitcnt vmcnt(8)` (never 0 in the loop) · 16 x v_mfma_f32_16x16x32_f16 · one raw s_barrier }.
//
// Geometry.  8 waves as 4 (row quarters) x 2 (centroid halves): a wave owns 64 rows x 128 centroids = 4 x 8 blocks of 16 x 16
// (128 accumulators).  In the 16 x 16 C layout a lane holds ONE centroid column and 4 rows per block, so the running arg-min of
// assign_kernel carries over with 16 row slots per lane (32 registers) across ALL centroid tiles of a work item (256 rows x
// every centroid); the 16 lanes sharing a row and the two centroid halves are folded only when the item ends.
//
// LDS: two k-step buffers of 64 KiB = [A0 | A1 | B0 | B1], each a 128-row x 128-byte image in the store's own XOR-swizzled slab
// format (zvk_common.hip.h blocked_offset): the centroid slabs are copied verbatim, the rows' swizzle is applied at the source.
// The image of one k-step is staged in four 16 KiB groups, named by the phase that reads them:
//     a0 = rows wr*64 + [0,32) of every wave row quarter   (read in phase 1, kept in registers for phase 4)
//     b0 = centroid rows [0,64) of both halves             (phase 1; used in phases 1-2)
//     a1 = rows wr*64 + [32,64)                            (phase 2; used in phases 2-3)
//     b1 = centroid rows [64,128) of both halves           (phase 3; used in phases 3-4)
// A group's LDS region is dead two phases after its read, so the SAME buffer takes the k-step after next while the current one
// is still being multiplied: phase 3 of step t issues a0(t+2), phase 4 b0(t+2), phase 1 of step t+1 a1(t+2), phase 2 b1(t+2) —
// every group has five to six phases (>= 1300 MFMA cycles) to land.  The wait in front of the barrier that ends phase p retires
// exactly what phase p+1 reads; the four groups issued after it stay in flight: vmcnt(8).  Reads come one barrier after the
// wait that retired their data (never in the same phase).
#pragma once
#include "zvk_assign.hip.h"

namespace zvk {

constexpr int A256_ROWS = 256;
constexpr size_t A256_BUF = 64 * 1024;                                // one k-step image: A0 | A1 | B0 | B1
constexpr size_t A256_LDS = 2 * A256_BUF + 1024;                      // + the row norms of the item (256 floats)

typedef float floatx4_t __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512, 1) assign256_f16_kernel(const AssignArgs a) {
  extern __shared__ f32x4 zvk_smem4[];
  char *smem = reinterpret_cast<char *>(zvk_smem4);
  float *qn_lds = reinterpret_cast<float *>(smem + 2 * A256_BUF);
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (scalar: everything derived from it — LDS-DMA destinations, the wave's tile — stays in SGPRs)
  const int wr = wave & 3, wc = wave >> 2;          // row quarter, centroid half
  const int lr = lane & 15, lq = lane >> 4;         // 16x16x32 operand coordinates: row / column lr, k-chunk lq
  const uint32_t dpad = a.dpad, nks = a.nks;
  const uint32_t ntiles = (a.n + TILE_N - 1) / TILE_N;           // 128-row tiles the centroid store holds
  const uint32_t npairs = (ntiles + 1) / 2;
  const uint32_t nsteps = npairs * nks;
  const uint32_t nitems = (a.nq + A256_ROWS - 1) / A256_ROWS;

  const float m_alpha = (a.metric == METRIC_L2) ? -2.f : -1.f;
  const float m_beta = (a.metric == METRIC_COSINE) ? 1.f : 0.f;
  const float m_lo = (a.metric == METRIC_L2) ? 0.f : -__builtin_inff();
  const bool l2 = a.metric == METRIC_L2;

  // ---- fragment read offsets inside one buffer (bytes) ----
  // A fragment (i, kh): row = wr*64 + i*16 + lr of the 256-row image -> half wr >> 1, row-in-half rh = (wr&1)*64 + i*16 + lr; chunk
  // c = kh*4 + lq.  B fragment (n, kh): centroid row wc*128 + n*16 + lr of the pair -> half wc, row-in-half n*16 + lr.  A step of
  // 16 rows leaves the row's swizzle term (rh >> 1) & 7 unchanged, so i and n are immediate offsets of 2 KiB on one base per kh.
  // The second k-half is chunk c ^ 4 — the same address with byte bit 6 flipped: one base register per operand.
  const int rh0 = (wr & 1) * 64 + lr;
  const uint32_t a_base0 = (uint32_t)((wr >> 1) * 16384 + (rh0 * 8 + (lq ^ ((rh0 >> 1) & 7))) * 16);
  const uint32_t b_base0 = (uint32_t)(32768 + wc * 16384 + (lr * 8 + (lq ^ ((lr >> 1) & 7))) * 16);
#define ZVK_A256_FA(I, KH) (*reinterpret_cast<const f16x8 *>(buf + ((a_base0 ^ ((KH) * 64u)) + (I) * 2048)))
#define ZVK_A256_FB(N, KH) (*reinterpret_cast<const f16x8 *>(buf + ((b_base0 ^ ((KH) * 64u)) + (N) * 2048)))
// a raw barrier the compiler may not move LDS traffic across (the intrinsic alone is not a memory barrier to it)
#define ZVK_A256_BARRIER()              \
  asm volatile("" ::: "memory");         \
  __builtin_amdgcn_s_barrier();          \
  asm volatile("" ::: "memory")

  // ---- staging geometry: a group = 2 LDS-DMA instructions of 512 lanes x 16 B (8 KiB each) ----
  // A groups: instruction j covers the image half j; threads 0..255 -> piece of row quarter 2j, 256..511 -> quarter 2j+1;
  //           a0: rows q*64 + [0,32), a1: rows q*64 + [32,64) (q = 2j + (tid >> 8)); 32 rows x 8 chunks per piece
  // B groups: instruction j covers centroid half j; b0: rows [0,64), b1: rows [64,128); 64 rows x 8 chunks, linear copy
  const int ap_row = ((tid >> 8) & 1) * 64 + ((tid & 255) >> 3);      // row-in-half of the a0 piece (a1: + 32)
  const int ap_pos = tid & 7;                                          // stored chunk position
  const uint32_t b_lds0 = (uint32_t)(tid * 16);                        // within the half; b1: + 8192

  for (uint32_t item = blockIdx.x; item < nitems; item += gridDim.x) {      // uniform exit
    // every item has 256 rows of its own to stage: the LAST one of a ragged batch is moved back so that it ends with the batch
    // (it re-labels a few rows of its neighbour with the same answers) — no per-row clamping in the staging addresses
    const uint32_t r0 = min(item * A256_ROWS, a.nq - A256_ROWS);
    const uint32_t nrows = A256_ROWS;
    // source of this thread's A chunk: row r0 + j*128 + g*32 + ap_row; the row's swizzle term is the same for g = 0, 1 (a step
    // of 32 rows), so ONE per-thread offset serves all four (g, j) pieces with wave-uniform addends
    const uint32_t a_src0 = (r0 + (uint32_t)ap_row) * dpad + (uint32_t)((ap_pos ^ ((ap_row >> 1) & 7)) * 4);      // floats
    // group g of step s: 0 = a0, 1 = b0, 2 = a1, 3 = b1 (the order they are issued in).  Steps past the end re-stage the last
    // one (into regions nobody reads again) so that the counted waits stay exact to the very last phase.
    auto stage = [&](int g, uint32_t s) {
      s = min(s, nsteps - 1);
      const uint32_t pair = s / nks, ks = s - pair * nks;
      char *buf = smem + (s & 1) * A256_BUF;
      const int gi = g >> 1;
      if ((g & 1) == 0) {
        // A: uniform base (row block, k-step) + ONE 32-bit per-thread offset
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          char *dst = buf + j * 16384 + gi * 4096 + (wave >> 2) * 8192 + (wave & 3) * 1024;                 // wave-uniform
          const char *base = reinterpret_cast<const char *>(a.queries) + ((size_t)(j * 128 + gi * 32) * dpad + (size_t)ks * TILE_K) * 4;
          __builtin_amdgcn_global_load_lds((glb_void *)(base + a_src0 * 4u), (lds_void *)dst, 16, 0, 0);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const uint32_t tile = min(2 * pair + (uint32_t)j, ntiles - 1);
          const char *base = reinterpret_cast<const char *>(a.base + (size_t)tile * TILE_N * dpad + (size_t)ks * SLAB) + gi * 8192;
          char *dst = buf + 32768 + j * 16384 + gi * 8192 + wave * 1024;                                    // wave-uniform
          __builtin_amdgcn_global_load_lds((glb_void *)(base + b_lds0), (lds_void *)dst, 16, 0, 0);
        }
      }
    };

    __syncthreads();                       // the previous item's reduction scratch / norms are free again
    if (tid < A256_ROWS) qn_lds[tid] = l2 ? a.qnorm[r0 + (uint32_t)tid] : 0.f;
    float best_s[4][4];
    uint32_t best_i[4][4];
    floatx4_t acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { best_s[i][j] = __builtin_inff(); best_i[i][j] = IDX_NONE; }
#pragma unroll
      for (int n = 0; n < 8; ++n) acc[i][n] = floatx4_t{0.f, 0.f, 0.f, 0.f};
    }

    // ---- prologue: step 0 whole, a0 / b0 of step 1; a0(0), b0(0) retired and published ----
    stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0); stage(0, 1); stage(1, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    ZVK_A256_BARRIER();

    f16x8 fa[4][2], fb[4][2];
    for (uint32_t s = 0; s < nsteps; ++s) {
      const char *buf = smem + (s & 1) * A256_BUF;
      const uint32_t pair = s / nks, ks = s - pair * nks;
      // ---------------- phase 1: A rows i = 0,1 (kept for phase 4) + B columns n = 0..3; quadrant (i 0-1, n 0-3) ----------------
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) fa[i][kh] = ZVK_A256_FA(i, kh);
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) fb[n][kh] = ZVK_A256_FB(n, kh);
      stage(2, s + 1);                                     // a1(s+1)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // a1(s) has landed (this wave's pieces)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][kh], fb[n][kh], acc[i][n], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      ZVK_A256_BARRIER();
      // ---------------- phase 2: A rows i = 2,3; quadrant (i 2-3, n 0-3) ----------------
#pragma unroll
      for (int i = 2; i < 4; ++i)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) fa[i][kh] = ZVK_A256_FA(i, kh);
      stage(3, s + 1);                                     // b1(s+1)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // b1(s) has landed
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int i = 2; i < 4; ++i)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][kh], fb[n][kh], acc[i][n], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      ZVK_A256_BARRIER();
      // ---------------- phase 3: B columns n = 4..7 (over the registers of n = 0..3); quadrant (i 2-3, n 4-7) ----------------
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) fb[n][kh] = ZVK_A256_FB(4 + n, kh);
      stage(0, s + 2);                                     // a0(s+2): its region was last read in phase 1
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int i = 2; i < 4; ++i)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[i][4 + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][kh], fb[n][kh], acc[i][4 + n], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      ZVK_A256_BARRIER();
      // ---------------- phase 4: no reads; quadrant (i 0-1, n 4-7) ----------------
      stage(1, s + 2);                                     // b0(s+2)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // a0(s+1), b0(s+1) have landed
      float bn[8];
      if (ks == nks - 1 && l2) {                           // column norms of this centroid pair, landing under the last MFMAs
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          const uint32_t col = pair * 256 + (uint32_t)(wc * 128 + n * 16 + lr);
          bn[n] = a.bnorm[min(col, ntiles * TILE_N - 1)];
        }
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[i][4 + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][kh], fb[n][kh], acc[i][4 + n], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      // ---- end of a centroid pair: fold the 64 x 128 block into the running arg-min (registers only) ----
      if (ks == nks - 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4 q4 = *reinterpret_cast<const f32x4 *>(qn_lds + wr * 64 + i * 16 + lq * 4);
#pragma unroll
          for (int n = 0; n < 8; ++n) {
            const uint32_t col = pair * 256 + (uint32_t)(wc * 128 + n * 16 + lr);
            const bool valid = col < a.n;                  // padding columns of the last tile (or of a missing second tile) never win
            const float nb = l2 ? bn[n] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float dot = acc[i][n][j];
              float sc = fmaxf(fmaf(m_alpha, dot, l2 ? q4[j] + nb : m_beta), m_lo);
              sc = valid ? sc : __builtin_inff();
              if (sc < best_s[i][j]) {                     // strict: a lane's columns come in ascending order, the first stays
                best_s[i][j] = sc;
                best_i[i][j] = col;
              }
              acc[i][n][j] = 0.f;
            }
          }
        }
      }
      ZVK_A256_BARRIER();
    }

    // ---- end of the item: the 16 lanes sharing a row slot (xor-shuffles over lr), then the two centroid halves through LDS ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the re-staged tail has landed: the operand buffers become scratch
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s_ = best_s[i][j];
        uint32_t i_ = best_i[i][j];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
          const float so = __shfl_xor(s_, m, 64);
          const uint32_t io = (uint32_t)__shfl_xor((int)i_, m, 64);
          if (so < s_ || (so == s_ && io < i_)) { s_ = so; i_ = io; }
        }
        best_s[i][j] = s_;
        best_i[i][j] = i_;
      }
    __syncthreads();                                      // every wave is done with the operand buffers
    float *red_s = reinterpret_cast<float *>(smem);       // [2 halves][256 rows]
    uint32_t *red_i = reinterpret_cast<uint32_t *>(smem + 2 * A256_ROWS * 4);
    if (lr == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = wr * 64 + i * 16 + lq * 4 + j;
          red_s[wc * A256_ROWS + row] = best_s[i][j];
          red_i[wc * A256_ROWS + row] = best_i[i][j];
        }
    }
    __syncthreads();
    if ((uint32_t)tid < nrows) {
      float s_ = red_s[tid];
      uint32_t i_ = red_i[tid];
      const float s1 = red_s[A256_ROWS + tid];
      const uint32_t i1 = red_i[A256_ROWS + tid];
      if (s1 < s_ || (s1 == s_ && i1 < i_)) { s_ = s1; i_ = i1; }
      a.out_label[r0 + tid] = i_;
      if (a.out_score) a.out_score[r0 + tid] = s_;
    }
  }
}

#undef ZVK_A256_FA
#undef ZVK_A256_FB
#undef ZVK_A256_BARRIER

}  // namespace zvk
