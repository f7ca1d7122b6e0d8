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
constexpr size_t A256_LDS = 2 * A256_BUF + 2048;                      // + the squared norms of two centroid pairs (2 x 256 floats)

typedef float floatx4_t __attribute__((ext_vector_type(4)));
#ifdef ZVK_A256_STAMPS
__device__ unsigned long long zvk_a256_acc[2][5][4];
#endif

template <bool L2>
__global__ void __launch_bounds__(512, 1) assign256_f16_kernel(const AssignArgs a) {
  extern __shared__ f32x4 zvk_smem4[];
  char *smem = reinterpret_cast<char *>(zvk_smem4);
  float *bn_lds = reinterpret_cast<float *>(smem + 2 * A256_BUF);      // [pair & 1][256]: |c|^2 of the pair's centroids, by LDS-DMA
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

  const float m_beta = (a.metric == METRIC_COSINE) ? 1.f : 0.f;      // (scores: L2 max(|q|^2 + |c|^2 - 2 q.c, 0), IP -q.c, cosine 1 - q.c)
  constexpr bool l2 = L2;

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
// the barrier between a phase's reads / DMA issue / wait and its MFMAs exists for the staggered schedule only
#ifdef ZVK_A256_LOCKSTEP
#define ZVK_A256_MID_BARRIER()
#else
#define ZVK_A256_MID_BARRIER() ZVK_A256_BARRIER()
#endif
#ifdef ZVK_A256_STAMPS      // (diagnostic builds: where a step's time goes — shader-clock stamps of wave 0 / wave 4 of work-group 0)
#define ZVK_A256_STAMP(P, X)                                                                                  \
  if (blockIdx.x == 0 && (wave & 3) == 0) {                                                                    \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                               \
    if (lane == 0) atomicAdd(&zvk_a256_acc[wc][(P)][(X)], t_ - stamp_prev);                                   \
    stamp_prev = t_;                                                                                           \
  }
#else
#define ZVK_A256_STAMP(P, X)
#endif
#ifdef ZVK_A256_NOWAIT      // (diagnostic builds: timing without the counted waits — results are then wrong)
#define ZVK_A256_VMCNT8()
#else
#define ZVK_A256_VMCNT8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
#endif

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
    // A step's staging cursor: the wave-uniform source bases of its k-step (advanced by additions, no multiplications in the loop)
    struct Cursor {
      uint32_t pair, ks, par;
      const char *a;              // a.queries + ks * 128 bytes            (row block offsets are added per piece)
      const char *b[2];           // the k-step slab of store tile 2 * pair + j (the last tile again when that one is missing)
    };
    const size_t tile_bytes = (size_t)TILE_N * dpad * 4, slab_bytes = (size_t)SLAB * 4;
    const uint32_t a_piece[2][2] = {{0u, 128u * dpad * 4u}, {32u * dpad * 4u, (128u + 32u) * dpad * 4u}};      // [gi][j] bytes
    auto cursor_at0 = [&]() {
      Cursor c;
      c.pair = 0; c.ks = 0; c.par = 0;
      c.a = reinterpret_cast<const char *>(a.queries);
      c.b[0] = reinterpret_cast<const char *>(a.base);
      c.b[1] = reinterpret_cast<const char *>(a.base) + (size_t)min(1u, ntiles - 1) * tile_bytes;
      return c;
    };
    // the step after c; past the end it stays on the last one (re-staged into regions nobody reads again, so that the counted
    // waits stay exact to the very last phase)
    auto next_step = [&](Cursor &c) {
      if (c.pair == npairs - 1 && c.ks == nks - 1) return;
      c.par ^= 1;
      if (++c.ks == nks) {
        c.ks = 0;
        ++c.pair;
        c.a = reinterpret_cast<const char *>(a.queries);
        c.b[0] = reinterpret_cast<const char *>(a.base) + (size_t)min(2 * c.pair, ntiles - 1) * tile_bytes;
        c.b[1] = reinterpret_cast<const char *>(a.base) + (size_t)min(2 * c.pair + 1, ntiles - 1) * tile_bytes;
      } else {
        c.a += 128;
        c.b[0] += slab_bytes;
        c.b[1] += slab_bytes;
      }
    };
    // group g of the step at cursor c: 0 = a0, 1 = b0, 2 = a1, 3 = b1 (the order they are issued in)
    auto stage = [&](int g, const Cursor &c) {
#ifdef ZVK_A256_NODMA
      if (c.pair + c.ks > 1) return;
#endif
      char *buf = smem + c.par * A256_BUF;
      const int gi = g >> 1;
      if ((g & 1) == 0) {
        // A: uniform base (row block, k-step) + ONE 32-bit per-thread offset
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          char *dst = buf + j * 16384 + gi * 4096 + (wave >> 2) * 8192 + (wave & 3) * 1024;                 // wave-uniform
          __builtin_amdgcn_global_load_lds((glb_void *)(c.a + a_piece[gi][j] + a_src0 * 4u), (lds_void *)dst, 16, 0, 0);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          char *dst = buf + 32768 + j * 16384 + gi * 8192 + wave * 1024;                                    // wave-uniform
          __builtin_amdgcn_global_load_lds((glb_void *)(c.b[j] + gi * 8192 + b_lds0), (lds_void *)dst, 16, 0, 0);
        }
      }
    };

    __syncthreads();                       // the previous item's reduction scratch / norms are free again
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
    uint32_t pair = 0, ks = 0;                       // the step being multiplied (buffer s & 1)
    Cursor c1 = cursor_at0();                        // step s + 1
    {
      const Cursor c0 = c1;
      stage(0, c0); stage(1, c0); stage(2, c0); stage(3, c0);
    }
    next_step(c1);
    Cursor c2 = c1;                                  // step s + 2
    next_step(c2);
    stage(0, c1); stage(1, c1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    ZVK_A256_BARRIER();

#ifndef ZVK_A256_LOCKSTEP
    // The two centroid halves' wave groups (waves 0-3 / 4-7: the two waves of every SIMD) run ONE barrier apart: while one group
    // multiplies, the other reads its fragments, issues its DMA and sits out its counted wait.  Group 1 takes one barrier more
    // here, group 0 one more after the loop.
    if (wc == 1) { ZVK_A256_BARRIER(); }
#endif
    f16x8 fa[4][2], fb[4][2];
#ifdef ZVK_A256_STAMPS
    unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
#endif
    for (uint32_t s = 0; s < nsteps; ++s) {
      const char *buf = smem + (s & 1) * A256_BUF;
      // ---------------- phase 1: A rows i = 0,1 (kept for phase 4) + B columns n = 0..3; quadrant (i 0-1, n 0-3) ----------------
      ZVK_A256_STAMP(1, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) fa[i][kh] = ZVK_A256_FA(i, kh);
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) fb[n][kh] = ZVK_A256_FB(n, kh);
      if (l2 && ks == 0 && wave < 4) {
        // the pair's column norms, one dword per lane of waves 0-3, into the parity slot of bn_lds: in flight for at least a whole
        // k-step before the fold reads them (nks >= 2: the host takes the 128 x 128 kernel otherwise); the extra operation only
        // makes these waves' counted waits more conservative
        const uint32_t colg = min(pair * 256 + (uint32_t)tid, ntiles * TILE_N - 1);
        __builtin_amdgcn_global_load_lds((glb_void *)(a.bnorm + colg), (lds_void *)(bn_lds + (pair & 1) * 256 + wave * 64), 4, 0, 0);
      }
      stage(2, c1);                                        // a1(s+1)
      ZVK_A256_VMCNT8();                                   // a1(s) has landed (this wave's pieces)
      ZVK_A256_STAMP(1, 1);
      ZVK_A256_MID_BARRIER();
      ZVK_A256_STAMP(1, 2);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][kh], fb[n][kh], acc[i][n], 0, 0, 0);
#ifndef ZVK_A256_LOCKSTEP
        // phase 2's fragments (a1: retired before this phase's mid barrier) are fetched UNDER this phase's MFMAs: their registers are free
        if (kh == 0) {
#pragma unroll
          for (int i = 2; i < 4; ++i)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) fa[i][k2] = ZVK_A256_FA(i, k2);
        }
#endif
      }
      __builtin_amdgcn_s_setprio(0);
      ZVK_A256_STAMP(1, 3);
      ZVK_A256_BARRIER();
      // ---------------- phase 2: A rows i = 2,3; quadrant (i 2-3, n 0-3) ----------------
      ZVK_A256_STAMP(2, 0);
#ifdef ZVK_A256_LOCKSTEP
#pragma unroll
      for (int i = 2; i < 4; ++i)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) fa[i][kh] = ZVK_A256_FA(i, kh);
#endif
      stage(3, c1);                                        // b1(s+1)
      ZVK_A256_VMCNT8();                                   // b1(s) has landed
      ZVK_A256_STAMP(2, 1);
      ZVK_A256_MID_BARRIER();
      ZVK_A256_STAMP(2, 2);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
        for (int i = 2; i < 4; ++i)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][kh], fb[n][kh], acc[i][n], 0, 0, 0);
#ifndef ZVK_A256_LOCKSTEP
        // the first k-half of phase 3's B fragments (b1: retired before this phase's mid barrier) over the registers the MFMAs above
        // have just consumed, under the second k-half's MFMAs
        if (kh == 0) {
#pragma unroll
          for (int n = 0; n < 4; ++n) fb[n][0] = ZVK_A256_FB(4 + n, 0);
        }
#endif
      }
      __builtin_amdgcn_s_setprio(0);
      ZVK_A256_STAMP(2, 3);
      ZVK_A256_BARRIER();
      // ---------------- phase 3: B columns n = 4..7 (over the registers of n = 0..3); quadrant (i 2-3, n 4-7) ----------------
      ZVK_A256_STAMP(3, 0);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
#ifdef ZVK_A256_LOCKSTEP
        fb[n][0] = ZVK_A256_FB(4 + n, 0);
#endif
        fb[n][1] = ZVK_A256_FB(4 + n, 1);
      }
      stage(0, c2);                                        // a0(s+2): its region was last read in phase 1
      ZVK_A256_STAMP(3, 1);
      ZVK_A256_MID_BARRIER();
      ZVK_A256_STAMP(3, 2);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int i = 2; i < 4; ++i)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[i][4 + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][kh], fb[n][kh], acc[i][4 + n], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      ZVK_A256_STAMP(3, 3);
      ZVK_A256_BARRIER();
      // ---------------- phase 4: no reads; quadrant (i 0-1, n 4-7) ----------------
      ZVK_A256_STAMP(4, 0);
      stage(1, c2);                                        // b0(s+2)
      ZVK_A256_VMCNT8();                                   // a0(s+1), b0(s+1) have landed
      ZVK_A256_STAMP(4, 1);
      ZVK_A256_MID_BARRIER();
      ZVK_A256_STAMP(4, 2);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[i][4 + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][kh], fb[n][kh], acc[i][4 + n], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      ZVK_A256_STAMP(4, 3);
      // ---- end of a centroid pair: fold the 64 x 128 block into the running arg-min (registers only) ----
#ifdef ZVK_A256_NOFOLD
      if (ks == nks - 1 && a.n == 0xffffffffu) {
#else
      if (ks == nks - 1) {
#endif
        // Compared per row: t = |c|^2 - 2 q.c (L2) or -q.c (IP / cosine) — the row's own |q|^2 (and cosine's 1) is common to all
        // its candidates and joins the winner when the item ends; padding columns of the last tile (or of a missing second tile)
        // carry +inf and never win.
        float cadd[8];
        uint32_t col[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          col[n] = pair * 256 + (uint32_t)(wc * 128 + n * 16 + lr);
          const float nb = l2 ? bn_lds[(pair & 1) * 256 + wc * 128 + n * 16 + lr] : 0.f;
          cadd[n] = col[n] < a.n ? nb : __builtin_inff();
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int n = 0; n < 8; ++n) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float t = fmaf(l2 ? -2.f : -1.f, acc[i][n][j], cadd[n]);
              if (t < best_s[i][j]) {                      // strict: a lane's columns come in ascending order, the first stays
                best_s[i][j] = t;
                best_i[i][j] = col[n];
              }
              acc[i][n][j] = 0.f;
            }
          }
        }
      }
      pair = c1.pair; ks = c1.ks;
      c1 = c2;
      next_step(c2);
      ZVK_A256_BARRIER();
    }

#ifndef ZVK_A256_LOCKSTEP
    if (wc == 0) { ZVK_A256_BARRIER(); }
#endif
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
      if (a.out_score) a.out_score[r0 + tid] = l2 ? fmaxf(a.qnorm[r0 + tid] + s_, 0.f) : s_ + m_beta;
    }
  }
}

#undef ZVK_A256_FA
#undef ZVK_A256_FB
#undef ZVK_A256_BARRIER
#undef ZVK_A256_MID_BARRIER
#undef ZVK_A256_VMCNT8

}  // namespace zvk
