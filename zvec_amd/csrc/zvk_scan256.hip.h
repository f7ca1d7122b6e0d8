// zvk_scan256.hip.h — the wide flat scan of fp16 rows on the 256 x 256 multi-phase tile.
// Part of the device code of libzvec_hip (included through scan_kernels.hip.h).
//
// Same job, same lists and same write-out as scan8_kernel<false, true, false> (zvk_scan.hip.h; FlatSearcher's batched
// search, flat_searcher.cc:143-189 -> flat_distance_matrix.h, heap.h:103-114): every (query tile, chunk of base tiles) work item
// leaves the chunk's k best positions of each of its queries in part_s / part_i[query * nchunks + chunk][k], merged by
// merge_kernel.  The 128 x 128 one-barrier tile reaches 0.2 of the dense-f16 peak on a 256-query batch (profiles/
// r3_flat1m_fp16_*): this kernel gives the scan the matrix loop of assign256_f16_kernel (zvk_assign256.hip.h — 8 waves as
// 4 x 2 over 256 queries x 256 base rows, k-steps of 64 halves in four phases of 16 x v_mfma_f32_16x16x32_f16, LDS-DMA issued
// two k-steps ahead behind counted `s_waitcnt vmcnt(8)`) with the BASE as the streamed operand and the 256 query rows as the
// re-read one, and does this when a pair of base tiles ends (256 x 256 scores, 128 accumulators per lane):
//
//   1. every lane tests its scores where they are, against min(its row's k-th score, the query-wide bound gtau) — rows' bounds
//      read once per epilogue, the test in a shifted form that costs two packed FMAs, two minima and one compare per 4 scores,
//      the exact score and the exact test on the (rare) taken side;
//   2. a passing score is appended to the queue of the wave that OWNS its row (rows [32 q, 32 q + 32) -> wave q): owner q is fed
//      by exactly two waves, each appending to its own half of q's queue at (wave-uniform count + rank among the passing lanes) —
//      no atomics.  The queues live in 32 KiB of LDS the matrix loop is not using at that moment: the a1 / b1 regions of the k-step
//      buffer just multiplied (their next DMA is issued in phases 1 / 2 of the following k-step, after the epilogue's last barrier);
//   3. after one barrier every wave drains its own queue, 64 entries at a time, one LANE per entry: entries of the same row are
//      serialised through a claim word (ds_min of the lane id), a lane inserts into its row's sorted list on its own (lane_insert:
//      the list in registers after one round of loads, the tail moved by stores) — the same admission rule and kept set as
//      owner_row / sorted_insert (zvk_common.hip.h; heap.h:103-114);
//   4. a queue that would overflow (lists filling from empty without seeded bounds: small bases, the first pairs) sends the whole
//      pair down scan8's epilogue instead — eight rounds of 32 query rows transposed through the same 32 KiB, owner waves admitting
//      whole rows with owner_row.  The accumulators are cleared only after that decision.
//
// The base rows' norms and the rows' gtau keys arrive by 4-byte LDS-DMA one k-step ahead of the epilogue, and the LDS stores in front
// of the drain are written as asm: the compiler puts a full `vmcnt(0)` in front of every LDS store it can see while LDS-DMA is in
// flight, which would drain the staged k-steps at every epilogue.
//
// LDS: 2 x 64 KiB operand buffers, 2 KiB of base-row norms, 7 KiB of row state, 2 KiB per unit of k for the lists: k <= 11 fits
// the CU's 160 KiB (the host takes scan8 otherwise, and for filtered scans, fewer than 256 queries or a single k-step).
// The waves run in lockstep (one barrier per phase): the epilogue needs all eight in the same place.
// Measured steps from the first version (scan8's epilogue alone: no faster than scan8) to this one: DESIGN.md §3.
#pragma once
#include "zvk_assign256.hip.h"

namespace zvk {

constexpr int S256_ROWS = 256;
#ifndef ZVK_S256_BAUX
#define ZVK_S256_BAUX 0      // cache policy bits of the base tiles' LDS-DMA (diagnostic builds: 2 = nt)
#endif
__host__ __device__ inline size_t scan256_lds_bytes(uint32_t k) {
  return 2 * A256_BUF + 2048 + 7 * (size_t)S256_ROWS * 4 + 2 * (size_t)S256_ROWS * k * 4;
}

// The queue of one owner wave lives in 4 KiB of the epilogue scratch: two halves (one per producing wave) of 224 entries of
// {score, row << 8 | column}, the 32 claim words of its rows, the two halves' entry counts.
constexpr int S256_QHALF = 224;
constexpr int S256_QCLAIM = 2 * S256_QHALF * 8;       // byte offsets inside the 4 KiB
constexpr int S256_QCNT = S256_QCLAIM + 32 * 4;

__global__ void __launch_bounds__(512, 1) scan256_f16_kernel(const ScanArgs a) {
  extern __shared__ f32x4 zvk_smem4[];
  char *smem = reinterpret_cast<char *>(zvk_smem4);
  float *bn_lds = reinterpret_cast<float *>(smem + 2 * A256_BUF);      // [pair & 1][256]: |b|^2 of the pair's rows, by LDS-DMA
  float *qn_s = bn_lds + 2 * S256_ROWS;
  RowState st;
  st.tau = qn_s + S256_ROWS;
  st.cnt = reinterpret_cast<uint32_t *>(st.tau + S256_ROWS);
  uint32_t *qrow_s = st.cnt + S256_ROWS;
  uint32_t *gk_lds = qrow_s + S256_ROWS;            // [256] the rows' query-wide bounds (keys), by LDS-DMA one k-step ahead of their use
  st.k = a.k;
  st.gt = reinterpret_cast<float *>(gk_lds + S256_ROWS);
  st.tq = st.gt + S256_ROWS;
  st.gtau = a.gtau;
  st.qrow = qrow_s;
  st.Ls = st.tq + S256_ROWS;
  st.Li = reinterpret_cast<uint32_t *>(st.Ls + (size_t)S256_ROWS * a.k);
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave & 3, wc = wave >> 2;          // query-row quarter, base-row half of the pair
  const int lr = lane & 15, lq = lane >> 4;         // 16x16x32 operand coordinates
  const uint32_t dpad = a.dpad, nks = a.nks, k = a.k;
  const uint32_t ntiles_total = (uint32_t)((a.n + TILE_N - 1) / TILE_N);
  const uint32_t rows_valid_total = (uint32_t)min((uint64_t)0xffffffffu, a.n);

  // metric fix-up exactly as in scan8_kernel: L2 max(-2 q.b + |q|^2 + |b|^2, 0); IP -q.b; cosine 1 - q.b
  const bool l2 = a.metric == METRIC_L2;
  const float m_alpha = l2 ? -2.f : -1.f;
  const float m_beta = (a.metric == METRIC_COSINE) ? 1.f : 0.f;
  const float m_nrm = l2 ? 1.f : 0.f;
  const float m_lo = l2 ? 0.f : -__builtin_inff();

  // fragment read offsets inside one buffer (bytes): see assign256_f16_kernel
  const int rh0 = (wr & 1) * 64 + lr;
  const uint32_t a_base0 = (uint32_t)((wr >> 1) * 16384 + (rh0 * 8 + (lq ^ ((rh0 >> 1) & 7))) * 16);
  const uint32_t b_base0 = (uint32_t)(32768 + wc * 16384 + (lr * 8 + (lq ^ ((lr >> 1) & 7))) * 16);
#define ZVK_S256_FA(I, KH) (*reinterpret_cast<const f16x8 *>(buf + ((a_base0 ^ ((KH) * 64u)) + (I) * 2048)))
#define ZVK_S256_FB(N, KH) (*reinterpret_cast<const f16x8 *>(buf + ((b_base0 ^ ((KH) * 64u)) + (N) * 2048)))
#define ZVK_S256_BARRIER()              \
  asm volatile("" ::: "memory");         \
  __builtin_amdgcn_s_barrier();          \
  asm volatile("" ::: "memory")
#define ZVK_S256_VMCNT8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
#ifdef ZVK_A256_STAMPS      // (diagnostic builds: shader-clock cycles wave 0 of work-group 0 spends in the sections of a pair's end; tools/flat_stamps.py)
#define ZVK_S256_STAMP(X)                                                                                      \
  if (blockIdx.x == 0 && wave == 0) {                                                                          \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                               \
    if (lane == 0) atomicAdd(&zvk_a256_acc[0][(X) >> 2][(X) & 3], t_ - stamp_prev);                           \
    stamp_prev = t_;                                                                                           \
  }
#else
#define ZVK_S256_STAMP(X)
#endif

  const int ap_row = ((tid >> 8) & 1) * 64 + ((tid & 255) >> 3);
  const int ap_pos = tid & 7;
  const uint32_t b_lds0 = (uint32_t)(tid * 16);

  // ---- the score scratch of an epilogue round: 32 row slots of 1 KiB in the a1 / b1 regions of a k-step buffer ----
  // slot s < 16 : a1 piece s >> 2 (image half (s >> 3), row quarter (s >> 2) & 1) + (s & 3) KiB
  // slot s >= 16: b1 half (s - 16) >> 3 + ((s - 16) & 7) KiB
  // A writer's slot is wr * 8 + jb * 4 + lq (jb = j & 1); the column index is XORed with 16 * (slot & 3) so that the four lq
  // groups of a wave store to four different bank groups.
  auto slot_addr = [](int s_) -> uint32_t {
    return s_ < 16 ? (uint32_t)((s_ >> 3) * 16384 + ((s_ >> 2) & 1) * 8192 + 4096 + (s_ & 3) * 1024)
                   : (uint32_t)(32768 + ((s_ - 16) >> 3) * 16384 + 8192 + ((s_ - 16) & 7) * 1024);
  };
#ifdef ZVK_CLOCK_STAMP      // (diagnostic builds: start / end stamps of every work-group, tools/flat_clock.py fp16)
  if (tid == 0 && blockIdx.x < 1024) {
    zvk_clock_stamps[blockIdx.x][0] = __builtin_amdgcn_s_memtime();
    zvk_clock_stamps[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  const uint32_t vtotal = ((a.nchunks + 7) / 8) * 8 * a.nqtiles;      // XCD-aware item order: see scan8_kernel
  for (uint32_t v = blockIdx.x; v < vtotal; v += gridDim.x) {         // uniform exit
    const uint32_t qtile = (v >> 3) % a.nqtiles;
    const uint32_t chunk = ((v >> 3) / a.nqtiles) * 8 + (v & 7);
    if (chunk >= a.nchunks) continue;
    const uint32_t tile_begin = chunk * a.tiles_per_chunk;
    const uint32_t tile_end = min(tile_begin + a.tiles_per_chunk, ntiles_total);
    const uint32_t npairs = (tile_end - tile_begin + 1) / 2;
    const uint32_t nsteps = npairs * nks;
    // every item stages 256 query rows: the LAST tile of a ragged batch is moved back so that it ends with the batch; the rows
    // it shares with its neighbour are dead here (bound -inf, no slot)
    const uint32_t q_first = qtile * S256_ROWS;
    const uint32_t r0 = min(q_first, a.nq - S256_ROWS);
    const uint32_t pos_end = min(tile_end * (uint32_t)TILE_N, rows_valid_total);

    if (tid < S256_ROWS) {
      const uint32_t qrow = r0 + tid;
      const bool live = qrow >= q_first;
      const float t_ = live ? a.threshold : -__builtin_inff();
      qrow_s[tid] = qrow;
      qn_s[tid] = l2 ? a.qnorm[qrow] : 0.f;
      st.tau[tid] = t_;
      st.gt[tid] = t_;
      st.tq[tid] = t_;
      st.cnt[tid] = 0;
    }

    const uint32_t a_src0 = (r0 + (uint32_t)ap_row) * dpad + (uint32_t)((ap_pos ^ ((ap_row >> 1) & 7)) * 4);      // floats
    struct Cursor {
      uint32_t pair, ks, par;
      const char *a;
      const char *b[2];
    };
    const size_t tile_bytes = (size_t)TILE_N * dpad * 4, slab_bytes = (size_t)SLAB * 4;
    const uint32_t a_piece[2][2] = {{0u, 128u * dpad * 4u}, {32u * dpad * 4u, (128u + 32u) * dpad * 4u}};      // [gi][j] bytes
    auto cursor_at0 = [&]() {
      Cursor c;
      c.pair = 0; c.ks = 0; c.par = 0;
      c.a = reinterpret_cast<const char *>(a.queries);
      c.b[0] = reinterpret_cast<const char *>(a.base) + (size_t)min(tile_begin, ntiles_total - 1) * tile_bytes;
      c.b[1] = reinterpret_cast<const char *>(a.base) + (size_t)min(tile_begin + 1, ntiles_total - 1) * tile_bytes;
      return c;
    };
    auto next_step = [&](Cursor &c) {
      if (c.pair == npairs - 1 && c.ks == nks - 1) return;      // past the end: the last step again (keeps the counted waits exact)
      c.par ^= 1;
      if (++c.ks == nks) {
        c.ks = 0;
        ++c.pair;
        c.a = reinterpret_cast<const char *>(a.queries);
        c.b[0] = reinterpret_cast<const char *>(a.base) + (size_t)min(tile_begin + 2 * c.pair, ntiles_total - 1) * tile_bytes;
        c.b[1] = reinterpret_cast<const char *>(a.base) + (size_t)min(tile_begin + 2 * c.pair + 1, ntiles_total - 1) * tile_bytes;
      } else {
        c.a += 128;
        c.b[0] += slab_bytes;
        c.b[1] += slab_bytes;
      }
    };
    auto stage = [&](int g, const Cursor &c) {      // group g of the step at c: 0 = a0, 1 = b0, 2 = a1, 3 = b1
      char *buf = smem + c.par * A256_BUF;
      const int gi = g >> 1;
      if ((g & 1) == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          char *dst = buf + j * 16384 + gi * 4096 + (wave >> 2) * 8192 + (wave & 3) * 1024;
          __builtin_amdgcn_global_load_lds((glb_void *)(c.a + a_piece[gi][j] + a_src0 * 4u), (lds_void *)dst, 16, 0, 0);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          char *dst = buf + 32768 + j * 16384 + gi * 8192 + wave * 1024;
          __builtin_amdgcn_global_load_lds((glb_void *)(c.b[j] + gi * 8192 + b_lds0), (lds_void *)dst, 16, 0, ZVK_S256_BAUX);
        }
      }
    };

    floatx4_t acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int n = 0; n < 8; ++n) acc[i][n] = floatx4_t{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: step 0 whole, a0 / b0 of step 1; a0(0), b0(0) retired and published ----
    uint32_t pair = 0, ks = 0;
    Cursor c1 = cursor_at0();
    {
      const Cursor c0 = c1;
      stage(0, c0); stage(1, c0); stage(2, c0); stage(3, c0);
    }
    next_step(c1);
    Cursor c2 = c1;
    next_step(c2);
    stage(0, c1); stage(1, c1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();                                  // (also publishes the row state)

    // ---- end of a pair of base tiles: 256 x 256 scores against the rows' bounds; the few that pass into the lists ----
#ifdef ZVK_A256_STAMPS
    unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
#endif
    auto pair_end = [&](uint32_t s) {
      ZVK_S256_STAMP(0);                                   // 0: the matrix loop since the last stamp
      // (the lane's coordinates are re-derived here from an opaque copy: everything the epilogue computes from them would otherwise be
      // hoisted out of the matrix loop and held in registers across it — slot offsets, queue codes, column ids)
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      const int lr = lane_o & 15, lq = lane_o >> 4;
        // the very last step re-stages itself into THIS buffer (next_step stays put): those copies must have landed before their
        // regions become scratch
        if (s == nsteps - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        char *sc = smem + (s & 1) * A256_BUF;
        // (LDS stores the compiler can see are given a full `vmcnt(0)` while LDS-DMA is in flight; the ones in front of the drain go
        // out as asm so that the k-steps staged ahead keep landing under the tests)
        if (tid < S256_ROWS) {
          const float g_ = fkey_inv(gk_lds[tid]);
          lds_store_f32(st.gt + tid, g_);
          lds_store_f32(st.tq + tid, fminf(st.tau[tid], g_));
        }
        const uint32_t pos0 = (tile_begin + 2 * pair) * (uint32_t)TILE_N;
        // per column: what joins alpha * q.b — |b|^2 (L2), 1 (cosine), 0 (IP); NaN for a column past the rows (never passes)
        float inner[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          const uint32_t col = (uint32_t)(wc * 128 + n * 16 + lr);
          const float bn_ = l2 ? bn_lds[(pair & 1) * 256 + col] : m_beta;
          inner[n] = pos0 + col < pos_end ? bn_ : __builtin_nanf("");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ZVK_S256_BARRIER();                                // every wave is past its reads of this buffer's a1 / b1; bounds published
        ZVK_S256_STAMP(1);                                 // 1: bounds refresh + first barrier
        // ---- test in registers: score <= min(list's k-th, query-wide bound) of its row.  A passing score goes to the wave that owns
        // its row (rows [32 q, 32 q + 32) -> wave q) as {score, row << 8 | column}.  Owner q is fed by exactly two waves (row quarter
        // q >> 1, both column halves), each appending to its OWN half of q's queue: the append position is a wave-uniform counter
        // plus the lane's rank among the passing lanes — no atomics, no LDS round trip under the 128 unrolled tests.  A half that
        // would overflow (lists still filling without seeded bounds) sends the whole pair down the transposing path.
        {
          // (the lane's bases pass through an empty asm here so that the 128 unrolled tests below derive their constants inside the
          // epilogue instead of holding 128 loop-invariant registers across the matrix loop)
          uint32_t code0 = (uint32_t)((wr * 64 + lq * 4) << 8) | (uint32_t)(wc * 128 + lr);
          uint32_t row0 = (uint32_t)(wr * 64 + lq * 4);
          asm volatile("" : "+v"(code0), "+v"(row0));
          uint32_t nput[2] = {0u, 0u};                     // entries this wave has appended for owners 2 wr, 2 wr + 1 (wave-uniform)
#ifdef ZVK_S256_NOTEST      // (diagnostic builds: the epilogue's fixed part alone — results are then wrong)
          if (a.n == 0xffffffffu)
#endif
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f32x4 tq4 = *reinterpret_cast<const f32x4 *>(st.tq + row0 + i * 16);
            const f32x4 qn4 = *reinterpret_cast<const f32x4 *>(qn_s + row0 + i * 16);      // (0 unless L2)
            // The unrolled test runs in the shifted form  min_j (alpha q.b_j - (bound_j - |q_j|^2 + slack_j)) <= -(|b|^2)  over the 4
            // rows a lane holds of one column: two packed FMAs, two minima, ONE compare and branch per 4 scores.  The slack,
            // 2^-19 (|bound| + |q|^2), covers the roundings of both forms (a passing L2 score has |b|^2 <= 2 (|q|^2 + bound), so
            // every term is within a few 2^-24 of that scale); the exact scores and the exact tests follow on the taken side.
            // +inf bounds stay +inf; the -inf of a dead row becomes NaN, which the minima drop and no compare passes.
            f32x4 nts;
#pragma unroll
            for (int j = 0; j < 4; ++j) nts[j] = -((tq4[j] - qn4[j]) + (__builtin_fabsf(tq4[j]) + qn4[j]) * 0x1p-19f);
            const uint32_t qhalf = lds_off(sc + slot_addr((wr * 2 + (i >> 1)) * 4) + wc * (S256_QHALF * 8));
#pragma unroll
            for (int n = 0; n < 8; ++n) {
              const f32x2 al2 = {m_alpha, m_alpha};
              const f32x2 w01 = al2 * f32x2{acc[i][n][0], acc[i][n][1]} + f32x2{nts[0], nts[1]};
              const f32x2 w23 = al2 * f32x2{acc[i][n][2], acc[i][n][3]} + f32x2{nts[2], nts[3]};
              const float wmin = __builtin_fminf(__builtin_fminf(w01.x, w01.y), __builtin_fminf(w23.x, w23.y));
              if (__ballot(wmin <= -inner[n]) != 0) {      // (wave-uniform)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  // the same operations as the transposing path below: fma(alpha, dot, fma(nrm, |q|^2 + |b|^2, beta)), clamped at the
                  // metric's floor (a NaN — a column past the rows — must not be laundered by the clamp)
                  const float x_ = fmaf(m_alpha, acc[i][n][j], qn4[j] + inner[n]);
                  const float v_ = fmaxf(x_, m_lo);
                  const bool ok = x_ <= tq4[j] && v_ <= tq4[j];
                  const uint64_t m_ = __ballot(ok);
                  if (m_ != 0) {
                    const uint32_t e_ = nput[i >> 1] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_, 0u));
                    if (ok && e_ < (uint32_t)S256_QHALF)
                      lds_store_u64(qhalf + e_ * 8, (uint64_t)__builtin_bit_cast(uint32_t, v_) | ((uint64_t)(code0 + (uint32_t)(((i * 16 + j) << 8) + n * 16)) << 32));
                    nput[i >> 1] += (uint32_t)__popcll(m_);
                  }
                }
              }
            }
          }
          if (lane == 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h) lds_store_u32(lds_off(sc + slot_addr((wr * 2 + h) * 4) + S256_QCNT + wc * 4), nput[h]);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the asm stores above, in front of the barrier that publishes them
        }
        ZVK_S256_STAMP(2);                                 // 2: the tests and appends
        ZVK_S256_BARRIER();
        ZVK_S256_STAMP(3);                                 // 3: second barrier
        uint32_t qn_a, qn_b;                               // this wave's queue: entries in its two halves
        bool overflow;
        {
          const uint32_t c_ = *reinterpret_cast<const uint32_t *>(sc + slot_addr((lane & 7) * 4) + S256_QCNT + ((lane >> 3) & 1) * 4);
          overflow = __ballot(c_ > (uint32_t)S256_QHALF) != 0;
          qn_a = (uint32_t)__builtin_amdgcn_readlane((int)c_, wave);
          qn_b = (uint32_t)__builtin_amdgcn_readlane((int)c_, wave + 8);
        }
#ifdef ZVK_S256_NOINS
        if (a.n != 0xffffffffu) { qn_a = 0; qn_b = 0; }
#endif
        if (!overflow) {
          // ---- drain: this wave's queue, 64 entries at a time, one lane per entry.  Entries of the same row are serialised through a
          // claim word per row (the lowest lane goes first); a lane inserts into its row's sorted list on its own.
          const char *qbase = sc + slot_addr(wave * 4);
          uint32_t *claim = reinterpret_cast<uint32_t *>(sc + slot_addr(wave * 4) + S256_QCLAIM);
          if (lane < 32) claim[lane] = 0xffffffffu;
          const uint32_t qn_mine = qn_a + qn_b;
          for (uint32_t e0 = 0; e0 < qn_mine; e0 += 64) {      // (wave-uniform)
            const uint32_t e_ = e0 + (uint32_t)lane;
            bool pending = e_ < qn_mine;
            uint2 ent = make_uint2(0u, 0u);
            if (pending) ent = *reinterpret_cast<const uint2 *>(qbase + (e_ < qn_a ? e_ : e_ - qn_a + (uint32_t)S256_QHALF) * 8);
            const int row = (int)(ent.y >> 8);
            const float es_ = __builtin_bit_cast(float, ent.x);
            const uint32_t epos = pos0 + (ent.y & 255u);
            while (__ballot(pending) != 0) {
              if (pending) __hip_atomic_fetch_min(&claim[row & 31], (uint32_t)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
              __builtin_amdgcn_wave_barrier();
              const bool mine = pending && __hip_atomic_load(&claim[row & 31], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == (uint32_t)lane;
              if (mine) {
                lane_insert(st, row, es_, epos);
                __hip_atomic_store(&claim[row & 31], 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                pending = false;
              }
              __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
              __builtin_amdgcn_wave_barrier();
            }
          }
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[i][n] = floatx4_t{0.f, 0.f, 0.f, 0.f};
          ZVK_S256_STAMP(4);                               // 4: drain + clearing the accumulators
        } else {
          // ---- the transposing path (scan8's epilogue): eight rounds of 32 rows through the scratch, owner waves admit whole rows.
          // Round R: rows wr*64 + (R >> 1)*16 + lq*4 + 2 (R & 1) + {0, 1}
          uint32_t wslot[2];
#pragma unroll
          for (int jb = 0; jb < 2; ++jb) wslot[jb] = slot_addr(wr * 8 + jb * 4 + lq) + (uint32_t)(((wc * 128 + lr) ^ (lq * 16)) * 4);
          float bnv[8];
          uint32_t valid = 0;
#pragma unroll
          for (int n = 0; n < 8; ++n) {
            const uint32_t col = (uint32_t)(wc * 128 + n * 16 + lr);
            bnv[n] = l2 ? bn_lds[(pair & 1) * 256 + col] : 0.f;
            valid |= (pos0 + col < pos_end ? 1u : 0u) << n;
          }
          ZVK_S256_BARRIER();                              // every wave has read the queue counters
#pragma unroll
          for (int R = 0; R < 8; ++R) {
            {
              const int i = R >> 1, jj = R & 1;
#pragma unroll
              for (int jb = 0; jb < 2; ++jb) {
                const int j = 2 * jj + jb;
                const float qn = qn_s[wr * 64 + i * 16 + lq * 4 + j];
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                  const float v_ = fmaxf(fmaf(m_alpha, acc[i][n][j], fmaf(m_nrm, qn + bnv[n], m_beta)), m_lo);
                  *reinterpret_cast<float *>(sc + (wslot[jb] ^ (uint32_t)(n * 64))) = ((valid >> n) & 1u) ? v_ : __builtin_inff();
                  acc[i][n][j] = 0.f;
                }
              }
            }
            ZVK_S256_BARRIER();
            // this wave's four rows of the round (slot wave * 4 + t), each as two spans of 128 columns, 2 per lane: all eight spans
            // are read and tested against the rows' bounds at once; only a span with a passing score goes through owner_row
            {
              const int row_b = (wave >> 1) * 64 + (wave & 1) + (R >> 1) * 16 + 2 * (R & 1);
              constexpr int row_step = 4;                  // the row of slot wave * 4 + t is row_b + row_step * t
              auto span = [&](int u_) -> const f32x2 * {
                const int t_ = u_ >> 1, hc_ = u_ & 1;
                const int swz_ = t_ * 16;
                return reinterpret_cast<const f32x2 *>(sc + slot_addr(wave * 4 + t_) + (uint32_t)(((hc_ * 128 + 2 * lane) ^ swz_) * 4));
              };
              uint32_t hits = 0;
              float tqv[4];
              f32x2 vv[8];
#pragma unroll
              for (int t = 0; t < 4; ++t) tqv[t] = st.tq[row_b + row_step * t];
#pragma unroll
              for (int u = 0; u < 8; ++u) vv[u] = *span(u);
#pragma unroll
              for (int u = 0; u < 8; ++u) hits |= (__ballot(vv[u].x <= tqv[u >> 1] || vv[u].y <= tqv[u >> 1]) != 0 ? 1u : 0u) << u;
              while (hits != 0) {                          // (wave-uniform)
                const int u = __builtin_ctz(hits);
                hits &= hits - 1;
                const int row = row_b + row_step * (u >> 1);
                const f32x2 v2 = *span(u);
                owner_row(st, row, v2.x, v2.y, st.tq[row], pos0 + (uint32_t)((u & 1) * 128), lane);
              }
            }
            ZVK_S256_BARRIER();
          }
        }
    };
    f16x8 fa[4][2], fb[4][2];
    for (uint32_t s = 0; s < nsteps; ++s) {
      const char *buf = smem + (s & 1) * A256_BUF;
      // ---------------- phase 1: A rows i = 0,1 (kept for phase 4) + B columns n = 0..3 ----------------
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) fa[i][kh] = ZVK_S256_FA(i, kh);
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) fb[n][kh] = ZVK_S256_FB(n, kh);
      if (l2 && ks == 0 && wave < 4) {
        // the pair's base-row norms, one dword per lane of waves 0-3 (in flight for a whole k-step before the epilogue: nks >= 2)
        const uint32_t colg = min((tile_begin + 2 * pair) * (uint32_t)TILE_N + (uint32_t)tid, ntiles_total * (uint32_t)TILE_N - 1);
        __builtin_amdgcn_global_load_lds((glb_void *)(a.bnorm + colg), (lds_void *)(bn_lds + (pair & 1) * 256 + wave * 64), 4, 0, 0);
      }
      // the shared bounds the pair's epilogue will test against: fetched the same way in front of the last k-step's DMA — the
      // counted waits retire it, no register result for the compiler to put a `vmcnt(0)` in front of (a bound a k-step old is
      // only a little looser)
      if (ks == nks - 1 && wave < 4)
        __builtin_amdgcn_global_load_lds((glb_void *)(a.gtau + r0 + tid), (lds_void *)(gk_lds + wave * 64), 4, 0, 0);
      stage(2, c1);                                        // a1(s+1)
      ZVK_S256_VMCNT8();                                   // a1(s) has landed
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][kh], fb[n][kh], acc[i][n], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      ZVK_S256_BARRIER();
      // ---------------- phase 2: A rows i = 2,3 ----------------
#pragma unroll
      for (int i = 2; i < 4; ++i)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) fa[i][kh] = ZVK_S256_FA(i, kh);
      stage(3, c1);                                        // b1(s+1)
      ZVK_S256_VMCNT8();                                   // b1(s) has landed
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int i = 2; i < 4; ++i)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][kh], fb[n][kh], acc[i][n], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      ZVK_S256_BARRIER();
      // ---------------- phase 3: B columns n = 4..7 (over the registers of n = 0..3) ----------------
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) fb[n][kh] = ZVK_S256_FB(4 + n, kh);
      stage(0, c2);                                        // a0(s+2)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int i = 2; i < 4; ++i)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[i][4 + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][kh], fb[n][kh], acc[i][4 + n], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      ZVK_S256_BARRIER();
      // ---------------- phase 4: no reads ----------------
      stage(1, c2);                                        // b0(s+2)
      ZVK_S256_VMCNT8();                                   // a0(s+1), b0(s+1) have landed
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[i][4 + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][kh], fb[n][kh], acc[i][4 + n], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);

      // ---- end of a pair of base tiles ----
#ifdef ZVK_S256_NOEPI      // (diagnostic builds: the matrix loop alone — results are then wrong)
      if (ks == nks - 1 && a.n == 0xffffffffu) pair_end(s);
#else
      if (ks == nks - 1) pair_end(s);
#endif
      pair = c1.pair; ks = c1.ks;
      c1 = c2;
      next_step(c2);
      ZVK_S256_BARRIER();
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the re-staged tail has landed before the next item's prologue
    __syncthreads();
    // ---- write the partial lists ----
    for (uint32_t j = tid; j < (uint32_t)S256_ROWS * k; j += 512) {
      const uint32_t row = j / k, t = j - row * k;
      const uint32_t qrow = r0 + row;
      if (qrow < q_first) continue;                      // a row of the neighbouring tile
      const uint32_t slot = qrow * a.nchunks + chunk;
      const uint32_t c = st.cnt[row];
      const size_t o = (size_t)slot * k + t;
      a.part_s[o] = (t < c) ? st.Ls[(size_t)row * k + t] : __builtin_inff();
      a.part_i[o] = (t < c) ? st.Li[(size_t)row * k + t] : IDX_NONE;
    }
    __syncthreads();
  }
#ifdef ZVK_CLOCK_STAMP
  if (tid == 0 && blockIdx.x < 1024) {
    zvk_clock_stamps[blockIdx.x][2] = __builtin_amdgcn_s_memtime();
    zvk_clock_stamps[blockIdx.x][3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

#undef ZVK_S256_FA
#undef ZVK_S256_FB
#undef ZVK_S256_BARRIER
#undef ZVK_S256_VMCNT8
#undef ZVK_S256_STAMP

}  // namespace zvk
