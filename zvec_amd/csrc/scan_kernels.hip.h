// scan_kernels.hip.h — gfx950 (CDNA4, wave64) device code of the zvec flat / IVF-Flat scan core.
//
// One kernel does the whole hot path of SURVEY §8(a) rows 1-4, 6-11: the batched query x base
// distance matrix on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains, so the
// scores are fp32-faithful like the reference's AVX kernels, euclidean_distance_matrix_fp32.cc:229-320,
// inner_product_matrix_fp32.cc:509-556) fused with the per-query bounded top-k
// (ailego::Heap semantics, heap.h:103-114: keep the k smallest, first-seen wins ties).
//
// HBM layout ("blocked"): base rows are stored in tiles of 128 rows; inside a tile the K dimension
// is cut in steps of 32 floats, and each (tile, k-step) slab is 128 rows x 128 B = 16 KiB contiguous,
// already XOR-swizzled the way the LDS image wants it.  A work-group therefore streams a list as
// a sequence of contiguous 16 KiB slabs (perfectly coalesced, DRAM-page friendly) and the slab is
// copied to LDS verbatim.
//
//   float offset of (pos, kcol), row stride dpad (multiple of 32):
//     tile = pos / 128, r = pos % 128, ks = kcol / 32, c = (kcol % 32) / 4, e = kcol % 4
//     off  = tile*128*dpad + ks*4096 + (r*8 + (c ^ ((r >> 1) & 7)))*4 + e
//
// The swizzle makes the ds_read_b128 operand fetch conflict free: lanes 0..31 of a wave read 32
// different rows at the same 16-byte chunk; (r&1)*8 + (c ^ ((r>>1)&7)) is a bijection onto the 16
// 16-byte slots of a 256 B bank row for each of the instruction's 16-lane groups.
#pragma once
#include "zvk_common.hip.h"
#include "zvk_scan.hip.h"
#include "zvk_assign.hip.h"
#include "zvk_assign256.hip.h"
#include "zvk_scan256.hip.h"
#include "zvk_merge.hip.h"
#include "zvk_rows.hip.h"
#include "zvk_shadow.hip.h"
#include "zvk_filter.hip.h"
#include "zvk_plan.hip.h"
#include "zvk_build.hip.h"
#include "zvk_group.hip.h"
