/*
 * zvec_oracle.c — CPU restatement of the zvec flat / IVF-Flat scan path.  TEST INFRASTRUCTURE ONLY
 * (see zvec_oracle.h).  Plain C99; compile with -ffp-contract=off so that every fused
 * multiply-add below is an explicit fmaf() and nothing else is contracted.
 *
 * All file:line references are to /root/reference (sudo-flow/zvec @ 2026-03-20).
 */
#include "zvec_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================================== *
 * 1x1 distance kernels.
 *
 * The reference dispatches at run time (x86 build with AVX-512 enabled):
 *     dim > 15 -> *AVX512 ; dim > 7 -> *AVX ; else *SSE
 *   SquaredEuclideanDistanceMatrix<float,1,1>::Compute  euclidean_distance_matrix_fp32.cc:287-320
 *   InnerProductMatrix<float,1,1>::Compute              inner_product_matrix_fp32.cc:588-615
 * Each body keeps TWO vector accumulators of W lanes (W = 16/8/4), strides 2W, folds one more
 * W-chunk into accumulator 0 if it fits, adds the accumulators lane-wise, handles the tail
 * (AVX-512: one masked FMA into the combined vector, then the horizontal add;
 *  AVX/SSE: horizontal add first, then scalar `sum += x*x` steps from the LAST tail element
 *  down to the first — the switch falls through from case 7/3 to case 1), and reduces
 * horizontally as  ((v0+v1)+(v2+v3)) + ((v4+v5)+(v6+v7))  after adding the upper half onto the
 * lower half (matrix_utility.i:36-47,143-149,239-244).
 * The model below reproduces that order exactly in scalar code.
 * ======================================================================================== */

typedef float (*zo_step_fn)(float m, float q, float acc);

static inline float step_ssd(float m, float q, float acc) {
  float x = m - q; /* _mm*_sub_ps */
  return fmaf(x, x, acc);
}
static inline float step_fma(float m, float q, float acc) {
  return fmaf(m, q, acc);
}

/* horizontal add of W lanes in the reference's order */
static inline float hadd_lanes(const float *v, int w) {
  if (w == 16) {
    float t[8];
    for (int i = 0; i < 8; ++i) t[i] = v[i] + v[i + 8]; /* low256 + high256 */
    return ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  }
  if (w == 8) {
    return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  /* w == 4: hadd(v,v) twice */
  return (v[0] + v[1]) + (v[2] + v[3]);
}

#define ZO_DEFINE_LANE_KERNEL(NAME, STEP)                                                    \
  static float NAME(const float *lhs, const float *rhs, size_t size) {                       \
    int w = (size > 15) ? 16 : (size > 7 ? 8 : 4);                                           \
    float s0[16], s1[16];                                                                    \
    for (int i = 0; i < 16; ++i) s0[i] = s1[i] = 0.0f;                                       \
    size_t aligned = (size / (size_t)(2 * w)) * (size_t)(2 * w);                             \
    size_t p = 0;                                                                            \
    for (; p != aligned; p += (size_t)(2 * w)) {                                             \
      for (int i = 0; i < w; ++i) s0[i] = STEP(lhs[p + i], rhs[p + i], s0[i]);               \
      for (int i = 0; i < w; ++i) s1[i] = STEP(lhs[p + w + i], rhs[p + w + i], s1[i]);       \
    }                                                                                        \
    if (size >= aligned + (size_t)w) {                                                       \
      for (int i = 0; i < w; ++i) s0[i] = STEP(lhs[p + i], rhs[p + i], s0[i]);               \
      p += (size_t)w;                                                                        \
    }                                                                                        \
    for (int i = 0; i < w; ++i) s0[i] = s0[i] + s1[i];                                       \
    if (w == 16) {                                                                           \
      size_t left = size - p; /* masked fma into the combined accumulator */                 \
      for (size_t i = 0; i < left; ++i) s0[i] = STEP(lhs[p + i], rhs[p + i], s0[i]);         \
      return hadd_lanes(s0, 16);                                                             \
    }                                                                                        \
    float result = hadd_lanes(s0, w);                                                        \
    for (size_t i = size; i > p; --i) /* case N: ... case 1: fall-through order */           \
      result = STEP(lhs[i - 1], rhs[i - 1], result);                                         \
    return result;                                                                           \
  }

ZO_DEFINE_LANE_KERNEL(lane_ssd, step_ssd)
ZO_DEFINE_LANE_KERNEL(lane_ip, step_fma)

static zo_dist_fn g_override[3] = {NULL, NULL, NULL};

void zo_set_distance_override(int metric, zo_dist_fn fn) {
  if (metric >= 0 && metric < 3) g_override[metric] = fn;
}

float zo_sqeuclid_f32(const float *m, const float *q, size_t dim) {
  return lane_ssd(m, q, dim);
}
float zo_ip_f32(const float *m, const float *q, size_t dim) {
  return lane_ip(m, q, dim);
}
/* MinusInnerProductMatrix<float,1,1>::Compute  inner_product_matrix_fp32.cc:870-895 */
float zo_minus_ip_f32(const float *m, const float *q, size_t dim) {
  return -lane_ip(m, q, dim);
}
/* SquaredEuclideanDistanceMatrix<T,M,N> / MinusInnerProductMatrix<T,M,N> (M >= 2), T = float or Float16 (converted to
 * fp32 first, distance_matrix_accum_fp16.i): one accumulator per (vector i, query j); the step is
 *     sum = fma(m - q, m - q, sum)   resp.   sum = fma(m, q, sum)        (minus-ip negates at the end).
 * How the k steps of one pair are chained depends on M (AVX-512 build, distance_matrix_accum_fp32.i):
 *   M >= 8   one vector per SIMD lane, k sequential: a single chain over k = 0 .. dim-1;
 *   M == 4   a 256-bit register holds TWO k steps of the 4 vectors (ACCUM_FP32_4X1/4X2/4X4_AVX, :573-680): an even-k
 *            chain and an odd-k chain over the first dim & ~1 steps, added (low + high half), then the odd tail step
 *            is folded into the sum;
 *   M == 2   FOUR k steps per register (ACCUM_FP32_2X1/2X2_AVX, :496-571): chains c0..c3 over k = t mod 4 for the first
 *            dim & ~3 steps; x = c0 + c2, y = c1 + c3 (low + high half); a tail of >= 2 steps goes x <- k, y <- k+1;
 *            r = x + y (movehl); a last single step is folded into r. */
static inline float blk_step(int l2, float a, float b, float acc) {
  if (l2) { const float d = a - b; return fmaf(d, d, acc); }
  return fmaf(a, b, acc);
}
static float half_to_float(uint16_t h);   /* defined with the fp16 1x1 kernels below */
static inline float blk_elem(int half, const void *p, size_t idx) {
  return half ? half_to_float(((const uint16_t *)p)[idx]) : ((const float *)p)[idx];
}
static float blk_pair(int l2, int half, int M, int N, const void *m, const void *q, int i, int j, size_t dim) {
#define BLK_TERM(k, acc) blk_step(l2, blk_elem(half, m, (k) * (size_t)M + i), blk_elem(half, q, (k) * (size_t)N + j), (acc))
  if (M == 4) {
    float e = 0.0f, o = 0.0f;
    const size_t al = dim & ~(size_t)1;
    for (size_t k = 0; k < al; k += 2) { e = BLK_TERM(k, e); o = BLK_TERM(k + 1, o); }
    float r = e + o;
    if (al != dim) r = BLK_TERM(al, r);
    return r;
  }
  if (M == 2) {
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f;
    const size_t al = dim & ~(size_t)3;
    for (size_t k = 0; k < al; k += 4) { c0 = BLK_TERM(k, c0); c1 = BLK_TERM(k + 1, c1); c2 = BLK_TERM(k + 2, c2); c3 = BLK_TERM(k + 3, c3); }
    float x = c0 + c2, y = c1 + c3;
    size_t k = al;
    if (dim >= al + 2) { x = BLK_TERM(k, x); y = BLK_TERM(k + 1, y); k += 2; }
    float r = x + y;
    if (k != dim) r = BLK_TERM(k, r);
    return r;
  }
  float acc = 0.0f;
  for (size_t k = 0; k < dim; ++k) acc = BLK_TERM(k, acc);
  return acc;
#undef BLK_TERM
}
static void blk_all(int l2, int half, int M, int N, const void *m, const void *q, size_t dim, float *out) {
  for (int j = 0; j < N; ++j)
    for (int i = 0; i < M; ++i) {
      const float v = blk_pair(l2, half, M, N, m, q, i, j, dim);
      out[(size_t)j * M + i] = l2 ? v : -v;
    }
}
void zo_sqeuclid_block_f32(int M, int N, const float *m, const float *q, size_t dim, float *out) { blk_all(1, 0, M, N, m, q, dim, out); }
void zo_minus_ip_block_f32(int M, int N, const float *m, const float *q, size_t dim, float *out) { blk_all(0, 0, M, N, m, q, dim, out); }
void zo_sqeuclid_block_f16(int M, int N, const uint16_t *m, const uint16_t *q, size_t dim, float *out) { blk_all(1, 1, M, N, m, q, dim, out); }
void zo_minus_ip_block_f16(int M, int N, const uint16_t *m, const uint16_t *q, size_t dim, float *out) { blk_all(0, 1, M, N, m, q, dim, out); }

/* CosineDistanceMatrix<float,1,1>::Compute  cosine_distance_matrix.h:32-50 */
float zo_cosine_f32(const float *m, const float *q, size_t dim_with_norm) {
  size_t d = dim_with_norm - 1; /* extra_dim = sizeof(float)/sizeof(float) */
  return 1 - lane_ip(m, q, d);
}

/* ---- one-to-many ("batch") distances: IndexMetric::batch_distance (index_metric.h:85-87) ---------------------------
 * BaseDistance<…>::ComputeBatch (math_batch/distance_batch.h:29-49) is a plain loop of the 1x1 kernel for
 * SquaredEuclidean and MinusInnerProduct (euclidean_metric.cc:870-876, inner_product_metric.cc:353-362), i.e. zo_dist.
 * Cosine alone takes another route (cosine_metric.cc:202-212 -> cosine_distance_batch.h:33-47): 1 - ip, with ip from
 * InnerProductDistanceBatch (inner_product_distance_batch.h:143-163), whose per-vector arithmetic does not depend on the
 * batch size and differs from the 1x1 kernel's lane order:
 *   fp32, AVX2 (inner_product_distance_batch_impl.h:46-118): ONE 8-lane accumulator of NEGATED products
 *     (acc = -(q*b) + acc, fused), folded high + low half to 4 lanes; a 4-step, then a 2-step landing in lanes 2 and 3;
 *     (l0 + l2) + (l1 + l3); an odd last element is subtracted as a plain product; the sign is flipped at the end.
 *   fp16, AVX-512F (inner_product_distance_batch_impl_fp16.h:85-167): ONE 16-lane fp32 accumulator, two fused steps
 *     per 32 elements; one more 16-step only if MORE than 16 elements are left (strict <), fold to 8 lanes (low + high),
 *     an 8-step only if MORE than 8 are left, ((v0+v1)+(v2+v3))+((v4+v5)+(v6+v7)), the remaining elements one by one. */
static float half_to_float(uint16_t h);
float zo_ip_batch_f32(const float *m, const float *q, size_t d) {
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s[4];
  size_t k = 0;
  for (; k + 8 <= d; k += 8)
    for (int l = 0; l < 8; ++l) acc[l] = fmaf(-q[k + l], m[k + l], acc[l]);
  for (int l = 0; l < 4; ++l) s[l] = acc[l + 4] + acc[l];
  if (k + 4 <= d) {
    for (int l = 0; l < 4; ++l) s[l] = fmaf(-q[k + l], m[k + l], s[l]);
    k += 4;
  }
  if (k + 2 <= d) {
    s[0] = fmaf(-0.0f, 0.0f, s[0]);
    s[1] = fmaf(-0.0f, 0.0f, s[1]);
    s[2] = fmaf(-q[k], m[k], s[2]);
    s[3] = fmaf(-q[k + 1], m[k + 1], s[3]);
    k += 2;
  }
  float res = (s[0] + s[2]) + (s[1] + s[3]);
  if (k < d) res = fmaf(-q[k], m[k], res);   /* `res -= q * b` as this image's compiler contracts it (pinned live) */
  return -res;
}
float zo_ip_batch_f16(const uint16_t *m, const uint16_t *q, size_t d) {
  float acc[16], y[8];
  for (int l = 0; l < 16; ++l) acc[l] = 0.f;
  size_t k = 0;
  for (; k + 32 <= d; k += 32) {
    for (int l = 0; l < 16; ++l) acc[l] = fmaf(half_to_float(q[k + l]), half_to_float(m[k + l]), acc[l]);
    for (int l = 0; l < 16; ++l) acc[l] = fmaf(half_to_float(q[k + 16 + l]), half_to_float(m[k + 16 + l]), acc[l]);
  }
  if (k + 16 < d) {
    for (int l = 0; l < 16; ++l) acc[l] = fmaf(half_to_float(q[k + l]), half_to_float(m[k + l]), acc[l]);
    k += 16;
  }
  for (int l = 0; l < 8; ++l) y[l] = acc[l] + acc[l + 8];
  if (k + 8 < d) {
    for (int l = 0; l < 8; ++l) y[l] = fmaf(half_to_float(m[k + l]), half_to_float(q[k + l]), y[l]);
    k += 8;
  }
  float res = ((y[0] + y[1]) + (y[2] + y[3])) + ((y[4] + y[5]) + (y[6] + y[7]));
  for (; k < d; ++k) res = fmaf(half_to_float(q[k]), half_to_float(m[k]), res);   /* contracted `+=`, pinned live */
  return res;
}
float zo_cosine_batch_f32(const float *m, const float *q, size_t dim_with_norm) { return 1 - zo_ip_batch_f32(m, q, dim_with_norm - 1); }
float zo_cosine_batch_f16(const uint16_t *m, const uint16_t *q, size_t dim_with_norm) { return 1 - zo_ip_batch_f16(m, q, dim_with_norm - 2); }

/* ---- fp16 rows (IndexMeta::DT_FP16, HalfFloatConverter / HalfFloatReformer) ------------------------
 * SquaredEuclideanDistanceMatrix<Float16,1,1> / (Minus)InnerProductMatrix<Float16,1,1> on an AVX-512 CPU
 * WITHOUT AVX512-FP16 (euclidean_distance_matrix_fp16.cc:137-158, inner_product_matrix_fp16.cc:143-186):
 * ACCUM_FP16_1X1_AVX512 (distance_matrix_accum_fp16.i:554-594) converts halves to fp32 (exact) and keeps
 * ONE 16-lane fp32 accumulator: per 32 elements two FMA steps (low 16 then high 16), one more 16-step if
 * it fits, fold 16 -> 8 lanes (low + high), one 8-lane step if it fits, the last < 8 elements as one
 * zero-padded 8-lane step, then ((v0+v1)+(v2+v3))+((v4+v5)+(v6+v7)). */
static float half_to_float(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, bits;
  if (exp == 0) {
    if (man == 0) bits = sign;
    else {
      int e = -1;
      do { man <<= 1; ++e; } while (!(man & 0x400u));
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
    }
  } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
  else bits = sign | ((exp + 112) << 23) | (man << 13);
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

/* fp32 -> fp16, round to nearest even (what _mm_cvtps_ph(…, _MM_FROUND_TO_NEAREST_INT) / FloatHelper::ToFP16 do) */
uint16_t zo_float_to_half(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  int32_t exp = (int32_t)((x >> 23) & 0xff) - 127 + 15;
  uint32_t man = x & 0x7fffffu;
  if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (man ? 0x200u | (man >> 13) : 0));
  if (exp >= 31) return (uint16_t)(sign | 0x7c00u);
  if (exp <= 0) {
    if (exp < -10) return (uint16_t)sign;
    man |= 0x800000u;
    uint32_t shift = (uint32_t)(14 - exp);
    uint32_t half = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1), mid = 1u << (shift - 1);
    if (rem > mid || (rem == mid && (half & 1))) ++half;
    return (uint16_t)(sign | half);
  }
  uint32_t half = ((uint32_t)exp << 10) | (man >> 13);
  uint32_t rem = man & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (half & 1))) ++half;   /* may carry into the exponent: correct */
  return (uint16_t)(sign | half);
}
float zo_half_to_float(uint16_t h) { return half_to_float(h); }

#define ZO_DEFINE_F16_KERNEL(NAME, STEP)                                                          \
  static float NAME(const uint16_t *m, const uint16_t *q, size_t dim) {                           \
    float s[16];                                                                                  \
    for (int i = 0; i < 16; ++i) s[i] = 0.0f;                                                     \
    size_t aligned = (dim >> 5) << 5, p = 0;                                                      \
    for (; p != aligned; p += 32) {                                                               \
      for (int i = 0; i < 16; ++i) s[i] = STEP(half_to_float(m[p + i]), half_to_float(q[p + i]), s[i]);            \
      for (int i = 0; i < 16; ++i) s[i] = STEP(half_to_float(m[p + 16 + i]), half_to_float(q[p + 16 + i]), s[i]);  \
    }                                                                                             \
    if (dim >= aligned + 16) {                                                                    \
      for (int i = 0; i < 16; ++i) s[i] = STEP(half_to_float(m[p + i]), half_to_float(q[p + i]), s[i]);            \
      p += 16;                                                                                    \
    }                                                                                             \
    float y[8];                                                                                   \
    for (int i = 0; i < 8; ++i) y[i] = s[i] + s[i + 8];                                           \
    if (dim >= p + 8) {                                                                           \
      for (int i = 0; i < 8; ++i) y[i] = STEP(half_to_float(m[p + i]), half_to_float(q[p + i]), y[i]);             \
      p += 8;                                                                                     \
    }                                                                                             \
    if (p < dim) { /* MATRIX_FP16_MASK_AVX: one zero-padded 8-lane step; a tail of exactly ONE element */  \
      size_t left = dim - p; /* sits in lane 7 (its _mm_set_epi16 lists it first = highest lane, :96-107) */ \
      if (left == 1) {                                                                            \
        y[7] = STEP(half_to_float(m[p]), half_to_float(q[p]), y[7]);                              \
      } else {                                                                                    \
        for (size_t i = 0; i < left; ++i)                                                         \
          y[i] = STEP(half_to_float(m[p + i]), half_to_float(q[p + i]), y[i]);                    \
      }                                                                                           \
    }                                                                                             \
    return hadd_lanes(y, 8);                                                                      \
  }
ZO_DEFINE_F16_KERNEL(lane_ssd_f16, step_ssd)
ZO_DEFINE_F16_KERNEL(lane_ip_f16, step_fma)

float zo_sqeuclid_f16(const uint16_t *m, const uint16_t *q, size_t dim) { return lane_ssd_f16(m, q, dim); }
float zo_ip_f16(const uint16_t *m, const uint16_t *q, size_t dim) { return lane_ip_f16(m, q, dim); }
float zo_minus_ip_f16(const uint16_t *m, const uint16_t *q, size_t dim) { return -lane_ip_f16(m, q, dim); }

/* Norm2Matrix<float,1>::Compute, AVX-512 build: NORM_FP32_1_AVX512 (norm_matrix_fp32.i:120-157)
 * two 16-lane accumulators, one extra 16-chunk and the masked tail both go to accumulator 0,
 * then add + horizontal add + sqrt.  (With AVX512F compiled in, this macro is used for every dim.) */
float zo_norm2_f32(const float *m, size_t dim) {
  float s0[16], s1[16];
  for (int i = 0; i < 16; ++i) s0[i] = s1[i] = 0.0f;
  size_t aligned = (dim >> 5) << 5, p = 0;
  for (; p != aligned; p += 32) {
    for (int i = 0; i < 16; ++i) s0[i] = fmaf(m[p + i], m[p + i], s0[i]);
    for (int i = 0; i < 16; ++i) s1[i] = fmaf(m[p + 16 + i], m[p + 16 + i], s1[i]);
  }
  if (dim >= aligned + 16) {
    for (int i = 0; i < 16; ++i) s0[i] = fmaf(m[p + i], m[p + i], s0[i]);
    p += 16;
  }
  for (size_t i = 0; p + i < dim; ++i) s0[i] = fmaf(m[p + i], m[p + i], s0[i]);
  for (int i = 0; i < 16; ++i) s0[i] = s0[i] + s1[i];
  return sqrtf(hadd_lanes(s0, 16));
}

/* Normalizer<float>::L2  normalizer.h:46-51 (+ NormalizeAVX512: element-wise IEEE division) */
void zo_normalize_l2_f32(float *arr, size_t dim, float *norm) {
  float n = zo_norm2_f32(arr, dim);
  *norm = n;
  if (n > 0.0f) {
    for (size_t i = 0; i < dim; ++i) arr[i] = arr[i] / n;
  }
}

/* CosineConverter (fp32 -> fp32) cosine_converter.cc:112-127 and
 * CosineReformer::transform cosine_reformer.cc:66-101: normalised copy + trailing norm. */
void zo_cosine_transform_f32(const float *in, size_t dim, float *out) {
  memcpy(out, in, dim * sizeof(float));
  float norm = 0.0f;
  zo_normalize_l2_f32(out, dim, &norm);
  out[dim] = norm;
}

static zo_dist_fn g_override16[3] = {NULL, NULL, NULL};
void zo_set_distance_override_f16(int metric, zo_dist_fn fn) {
  if (metric >= 0 && metric < 3) g_override16[metric] = fn;
}

/* dtype: 0 = fp32 rows, 1 = fp16 rows.  Cosine over fp16 rows: CosineDistanceMatrix<Float16,1,1>
 * (cosine_distance_matrix.h:32-50): extra_dim = sizeof(float)/sizeof(Float16) = 2 trailing half slots hold the
 * norm, out = 1 - InnerProductMatrix<Float16,1,1>(m, q, dim - 2). */
static inline float zo_distance_t(int dtype, int metric, const void *m, const void *q, size_t dim) {
  if (dtype == 0) return zo_distance(metric, (const float *)m, (const float *)q, dim);
  if (metric == ZO_METRIC_COSINE) {
    if (g_override16[ZO_METRIC_IP]) return 1.0f - (-g_override16[ZO_METRIC_IP]((const float *)m, (const float *)q, dim - 2));
    return 1.0f - lane_ip_f16((const uint16_t *)m, (const uint16_t *)q, dim - 2);
  }
  if (g_override16[metric]) return g_override16[metric]((const float *)m, (const float *)q, dim);
  if (metric == ZO_METRIC_L2) return lane_ssd_f16((const uint16_t *)m, (const uint16_t *)q, dim);
  return -lane_ip_f16((const uint16_t *)m, (const uint16_t *)q, dim);
}

float zo_distance(int metric, const float *m, const float *q, size_t dim) {
  if (g_override[metric]) return g_override[metric](m, q, dim);
  switch (metric) {
    case ZO_METRIC_L2:
      return zo_sqeuclid_f32(m, q, dim);
    case ZO_METRIC_IP:
      return zo_minus_ip_f32(m, q, dim);
    default:
      return zo_cosine_f32(m, q, dim);
  }
}

/* ======================================================================================== *
 * Bounded heap: ailego::Heap<IndexDocument> (heap.h) with std::less on score
 * (IndexDocument::operator< compares score only, index_document.h:143), plus the RNN gate of
 * IndexDocumentHeap::emplace (index_document.h:250-261).
 * ======================================================================================== */

void zo_heap_init(zo_heap *h, zo_doc *storage, size_t limit, float threshold) {
  h->a = storage;
  h->n = 0;
  h->limit = limit < 1 ? 1 : limit; /* Heap::limit(): max(max,1) heap.h:155-158 */
  h->threshold = threshold;
}

/* std::__push_heap (libstdc++ bits/stl_heap.h): sift the new last element up while
 * parent < value. */
static void push_heap_up(zo_doc *a, size_t hole, zo_doc value) {
  while (hole > 0) {
    size_t parent = (hole - 1) / 2;
    if (!(a[parent].score < value.score)) break;
    a[hole] = a[parent];
    hole = parent;
  }
  a[hole] = value;
}

/* Heap::replace_heap  heap.h:180-205 */
static void replace_heap_top(zo_doc *a, size_t count, zo_doc val) {
  size_t hole = 0;
  if (count > 1) {
    size_t child = 1;
    while (child < count) {
      size_t right = child + 1;
      if (right < count && a[child].score < a[right].score) child = right;
      if (!(val.score < a[child].score)) break;
      a[hole] = a[child];
      hole = child;
      child = (hole << 1) + 1;
    }
  }
  a[hole] = val;
}

/* IndexDocumentHeap::emplace + Heap::emplace  heap.h:103-114 */
void zo_heap_emplace(zo_heap *h, uint64_t key, float score, uint32_t index) {
  if (!(score <= h->threshold)) return;
  zo_doc v;
  v.key = key;
  v.score = score;
  v.index = index;
  if (h->n == h->limit) {
    if (v.score < h->a[0].score) replace_heap_top(h->a, h->n, v);
  } else {
    h->n += 1;
    push_heap_up(h->a, h->n - 1, v);
  }
}

static int doc_cmp(const void *pa, const void *pb) {
  const zo_doc *a = (const zo_doc *)pa, *b = (const zo_doc *)pb;
  if (a->score < b->score) return -1;
  if (a->score > b->score) return 1;
  if (a->index < b->index) return -1;
  if (a->index > b->index) return 1;
  return 0;
}

void zo_heap_sort(zo_heap *h) {
  qsort(h->a, h->n, sizeof(zo_doc), doc_cmp);
}

size_t zo_heap_replay(const float *scores, size_t n, size_t limit, float threshold,
                      uint32_t *out_index, float *out_score) {
  zo_doc *st = (zo_doc *)malloc(sizeof(zo_doc) * (limit ? limit : 1));
  zo_heap h;
  zo_heap_init(&h, st, limit, threshold);
  for (size_t i = 0; i < n; ++i) zo_heap_emplace(&h, i, scores[i], (uint32_t)i);
  for (size_t i = 0; i < h.n; ++i) {
    out_index[i] = h.a[i].index;
    out_score[i] = h.a[i].score;
  }
  size_t r = h.n;
  free(st);
  return r;
}

static inline int bit_set(const uint64_t *bits, uint64_t pos) {
  return bits && ((bits[pos >> 6] >> (pos & 63)) & 1u);
}

/* ======================================================================================== *
 * Flat scan.  FlatSearcherContext::batch_search_row_nofilter / _filter
 * (flat_searcher_context.h:846-1003): for every stored vector in storage order, for every query:
 * 1x1 distance, heap.emplace(key, score, index); afterwards keys are mapped and the heap sorted.
 * A vector whose key the filter rejects is skipped for all queries (:949-963).
 * (The single-query loops :420-560 and FlatStreamerEntity::search flat_streamer_entity.cc:212-316
 * visit vectors in the same storage order, so the same restatement covers them.)
 * ======================================================================================== */
static int flat_search_range(int dtype, const void *base_v, const uint64_t *keys, uint64_t n, uint32_t dim,
                             int metric, const void *queries_v, uint32_t q0, uint32_t q1,
                             uint32_t topk, float threshold, const uint64_t *exclude_bits,
                             uint64_t *out_keys, float *out_scores, uint32_t *out_index,
                             uint32_t *out_counts) {
  if (topk == 0) return -31; /* IndexError_InvalidArgument */
  zo_doc *st = (zo_doc *)malloc(sizeof(zo_doc) * topk);
  if (!st) return -2;
  const size_t rb = (size_t)dim * (dtype ? 2 : 4);   /* row bytes */
  const char *base = (const char *)base_v, *queries = (const char *)queries_v;
  for (uint32_t q = q0; q < q1; ++q) {
    zo_heap h;
    zo_heap_init(&h, st, topk, threshold);
    const char *qv = queries + (size_t)q * rb;
    for (uint64_t i = 0; i < n; ++i) {
      if (bit_set(exclude_bits, i)) continue;
      float s = zo_distance_t(dtype, metric, base + (size_t)i * rb, qv, dim);
      zo_heap_emplace(&h, keys ? keys[i] : i, s, (uint32_t)i);
    }
    zo_heap_sort(&h);
    for (size_t j = 0; j < h.n; ++j) {
      out_keys[(size_t)q * topk + j] = h.a[j].key;
      out_scores[(size_t)q * topk + j] = h.a[j].score;
      if (out_index) out_index[(size_t)q * topk + j] = h.a[j].index;
    }
    out_counts[q] = (uint32_t)h.n;
  }
  free(st);
  return 0;
}

int zo_flat_search(const float *base, const uint64_t *keys, uint64_t n, uint32_t dim, int metric,
                   const float *queries, uint32_t nq, uint32_t topk, float threshold,
                   const uint64_t *exclude_bits, uint64_t *out_keys, float *out_scores,
                   uint32_t *out_index, uint32_t *out_counts) {
  return flat_search_range(0, base, keys, n, dim, metric, queries, 0, nq, topk, threshold,
                           exclude_bits, out_keys, out_scores, out_index, out_counts);
}

/* ---------------------------------------------------------------------------------------- *
 * Column-major flat scan: FlatSearcherContext<32>::batch_search_column_{nofilter,filter}
 * (flat_searcher_context.h:682-752, :754-845) — the reference's dense "32 x K tile" path for
 * indexes with d <= 512 (flat_builder.cc:27-66).  Restated with its own loop structure:
 *   - the features segment = full 32-row blocks TRANSPOSED ([dim][32], flat_builder.cc:188-276)
 *     followed by the n % 32 left-over rows row-major;
 *   - TransposeQueries<32> (flat_utility.h:106-158): the batch is cut greedily into groups of
 *     K = 32, 16, 8, 4, 2, 1 queries, each group interleaved ([dim][K]);
 *   - a full block against a group: the M=32 x N=K block kernel (batch_enqueue_*, :236-330),
 *     scores_[k*32 + j]; every query of the group then emplaces rows j = 0..31 in order
 *     (with a filter: only rows whose block_mask bit is set, mask bit = !filter(key), and a
 *     block whose mask is 0 is skipped entirely, :785-796);
 *   - a left-over row against a group of K > 1: the M=K x N=1 block kernel with the GROUP as the
 *     matrix and the row as the query (single_enqueue_nofilter, :336-357); against the K = 1
 *     tail query: the 1x1 kernel (:360-370) — the only place the lane-ordered 1x1 sum appears;
 *   - keys are mapped from the row index after the scan, then heap.sort() (:745-750).
 * The block kernels are the pinned zo_*_block_* restatements (one sequential FMA chain per pair).
 * `base` is given ROW-major here; the transposition is done inside, as FlatBuilder would.
 * ---------------------------------------------------------------------------------------- */
static void column_block_scores(int dtype, int metric, int M, int N, const void *m, const void *q, uint32_t dim, float *out) {
  if (dtype) {
    if (metric == ZO_METRIC_L2) zo_sqeuclid_block_f16(M, N, (const uint16_t *)m, (const uint16_t *)q, dim, out);
    else zo_minus_ip_block_f16(M, N, (const uint16_t *)m, (const uint16_t *)q, dim, out);
  } else {
    if (metric == ZO_METRIC_L2) zo_sqeuclid_block_f32(M, N, (const float *)m, (const float *)q, dim, out);
    else zo_minus_ip_block_f32(M, N, (const float *)m, (const float *)q, dim, out);
  }
}

int zo_flat_search_column_t(int dtype, const void *base_v, const uint64_t *keys, uint64_t n, uint32_t dim, int metric,
                            const void *queries_v, uint32_t nq, uint32_t topk, float threshold,
                            const uint64_t *exclude_bits, uint64_t *out_keys, float *out_scores,
                            uint32_t *out_index, uint32_t *out_counts) {
  if (topk == 0) return -31;
  if (metric != ZO_METRIC_L2 && metric != ZO_METRIC_IP) return -12;    /* the metrics with M x N tables restated here */
  const size_t es = dtype ? 2 : 4, rb = (size_t)dim * es;
  const char *base = (const char *)base_v, *queries = (const char *)queries_v;
  enum { B = 32 };
  /* TransposeQueries<32>: groups of 32, 16, ..., 1 */
  char *tq = (char *)malloc((size_t)nq * rb + 1);
  uint32_t *gstart = (uint32_t *)malloc(sizeof(uint32_t) * (nq + 1)), *gk = (uint32_t *)malloc(sizeof(uint32_t) * (nq + 1));
  zo_doc *st = (zo_doc *)malloc(sizeof(zo_doc) * (size_t)topk * nq);
  zo_heap *heaps = (zo_heap *)malloc(sizeof(zo_heap) * nq);
  char *blk = (char *)malloc((size_t)B * rb + 1);
  if (!tq || !gstart || !gk || !st || !heaps || !blk) return -2;
  uint32_t ngroups = 0;
  {
    uint32_t qi = 0, left = nq;
    for (uint32_t K = B; K >= 1; K >>= 1)          /* count / K groups of K, the remainder goes on with K / 2 */
      while (left >= K) {
        gstart[ngroups] = qi; gk[ngroups] = K; ++ngroups;
        for (uint32_t j = 0; j < K; ++j)
          for (uint32_t c = 0; c < dim; ++c)
            memcpy(tq + (size_t)qi * rb + ((size_t)c * K + j) * es, queries + (size_t)(qi + j) * rb + (size_t)c * es, es);
        qi += K; left -= K;
      }
  }
  for (uint32_t q = 0; q < nq; ++q) zo_heap_init(&heaps[q], st + (size_t)q * topk, topk, threshold);
  float scores[B * B];
  const uint64_t full = n / B * B;
  for (uint64_t b0 = 0; b0 < full; b0 += B) {
    uint32_t mask = 0;
    for (uint32_t j = 0; j < B; ++j)
      if (!bit_set(exclude_bits, b0 + j)) mask |= (1u << j);
    if (exclude_bits && mask == 0) continue;                       /* block skipped entirely */
    for (uint32_t j = 0; j < B; ++j)                               /* write_column_index: transpose the block */
      for (uint32_t c = 0; c < dim; ++c)
        memcpy(blk + ((size_t)c * B + j) * es, base + (size_t)(b0 + j) * rb + (size_t)c * es, es);
    for (uint32_t g = 0; g < ngroups; ++g) {
      const uint32_t K = gk[g];
      column_block_scores(dtype, metric, B, (int)K, blk, tq + (size_t)gstart[g] * rb, dim, scores);
      for (uint32_t k = 0; k < K; ++k)
        for (uint32_t j = 0; j < B; ++j)
          if (mask & (1u << j)) zo_heap_emplace(&heaps[gstart[g] + k], 0, scores[k * B + j], (uint32_t)(b0 + j));
    }
  }
  for (uint64_t r = full; r < n; ++r) {                            /* left-over rows, row-major */
    if (bit_set(exclude_bits, r)) continue;
    const char *row = base + (size_t)r * rb;
    for (uint32_t g = 0; g < ngroups; ++g) {
      const uint32_t K = gk[g];
      if (K > 1) {
        column_block_scores(dtype, metric, (int)K, 1, tq + (size_t)gstart[g] * rb, row, dim, scores);
        for (uint32_t k = 0; k < K; ++k) zo_heap_emplace(&heaps[gstart[g] + k], 0, scores[k], (uint32_t)r);
      } else {
        zo_heap_emplace(&heaps[gstart[g]], 0, zo_distance_t(dtype, metric, row, tq + (size_t)gstart[g] * rb, dim), (uint32_t)r);
      }
    }
  }
  for (uint32_t q = 0; q < nq; ++q) {
    zo_heap *h = &heaps[q];
    for (size_t j = 0; j < h->n; ++j) h->a[j].key = keys ? keys[h->a[j].index] : h->a[j].index;   /* it.set_key(owner_->key(it.index())) */
    zo_heap_sort(h);
    for (size_t j = 0; j < h->n; ++j) {
      out_keys[(size_t)q * topk + j] = h->a[j].key;
      out_scores[(size_t)q * topk + j] = h->a[j].score;
      if (out_index) out_index[(size_t)q * topk + j] = h->a[j].index;
    }
    out_counts[q] = (uint32_t)h->n;
  }
  free(tq); free(gstart); free(gk); free(st); free(heaps); free(blk);
  return 0;
}

/* ======================================================================================== *
 * IVF-Flat.
 *  coarse:   IVFCentroidIndex::search = FlatSearcher over the centroids, topk = nprobe
 *            (ivf_centroid_index.cc:273-297, ivf_searcher_context.h:70-74); result sorted by score.
 *  probing:  IVFSearcher::search_impl driver loop (ivf_searcher.cc:217-247):
 *              for i in centroids while total_scan_count < max_scan_count: scan list i;
 *              total_scan_count += list.vector_count   (the whole list, filtered or not :651,:715)
 *  list scan: IVFEntity::search (ivf_entity.cc:587-716): vectors in list order,
 *              heap->emplace(key, distance * norm_val, local_id); norm_val == 1 for fp32 lists;
 *              filtered keys are skipped (keeps mask :627-640).
 *  brute force: IVFSearcher::search_bf_impl -> IVFEntity::search(query, heap) scans every list
 *              in list-id order (ivf_entity.cc:719-745).
 *  final:    heap.sort(); topk_to_result truncates at score > threshold (ivf_searcher_context.h:184-208).
 * ======================================================================================== */
static int ivf_search_range(int dtype, const void *centroids_v, uint32_t nlist, const uint64_t *list_offsets,
                            const void *vecs_v, const uint64_t *keys, uint32_t dim, int metric,
                            const void *queries_v, uint32_t q0, uint32_t q1, uint32_t topk,
                            float threshold, uint32_t nprobe, uint32_t max_scan_count,
                            int brute_force, const uint64_t *exclude_bits, uint64_t *out_keys,
                            float *out_scores, uint32_t *out_index, uint32_t *out_counts,
                            uint32_t *out_scanned, uint32_t *out_probes) {
  if (topk == 0) return -31;
  if (nprobe < 1) nprobe = 1;
  if (nprobe > nlist) nprobe = nlist;
  zo_doc *st = (zo_doc *)malloc(sizeof(zo_doc) * topk);
  zo_doc *cst = (zo_doc *)malloc(sizeof(zo_doc) * nprobe);
  if (!st || !cst) return -2;
  const size_t rb = (size_t)dim * (dtype ? 2 : 4);
  const char *centroids = (const char *)centroids_v, *vecs = (const char *)vecs_v, *queries = (const char *)queries_v;
  for (uint32_t q = q0; q < q1; ++q) {
    const char *qv = queries + (size_t)q * rb;
    zo_heap h;
    zo_heap_init(&h, st, topk, threshold);
    uint32_t total_scan = 0;
    if (out_probes)
      for (uint32_t i = 0; i < nprobe; ++i) out_probes[(size_t)q * nprobe + i] = ~0u;
    if (brute_force) {
      for (uint32_t l = 0; l < nlist; ++l) {
        for (uint64_t p = list_offsets[l]; p < list_offsets[l + 1]; ++p) {
          if (bit_set(exclude_bits, p)) continue;
          float s = zo_distance_t(dtype, metric, vecs + (size_t)p * rb, qv, dim);
          zo_heap_emplace(&h, keys ? keys[p] : p, s, (uint32_t)p);
        }
        total_scan += (uint32_t)(list_offsets[l + 1] - list_offsets[l]);
      }
    } else {
      zo_heap ch;
      zo_heap_init(&ch, cst, nprobe, FLT_MAX);
      for (uint32_t c = 0; c < nlist; ++c) {
        float s = zo_distance_t(dtype, metric, centroids + (size_t)c * rb, qv, dim);
        zo_heap_emplace(&ch, c, s, c);
      }
      zo_heap_sort(&ch);
      for (size_t i = 0; i < ch.n && total_scan < max_scan_count; ++i) {
        uint32_t l = (uint32_t)ch.a[i].key;
        if (out_probes) out_probes[(size_t)q * nprobe + i] = l;
        for (uint64_t p = list_offsets[l]; p < list_offsets[l + 1]; ++p) {
          if (bit_set(exclude_bits, p)) continue;
          float s = zo_distance_t(dtype, metric, vecs + (size_t)p * rb, qv, dim);
          zo_heap_emplace(&h, keys ? keys[p] : p, s, (uint32_t)p);
        }
        total_scan += (uint32_t)(list_offsets[l + 1] - list_offsets[l]);
      }
    }
    zo_heap_sort(&h);
    size_t cnt = 0;
    for (size_t j = 0; j < h.n; ++j) {
      if (h.a[j].score > threshold) break;
      out_keys[(size_t)q * topk + j] = h.a[j].key;
      out_scores[(size_t)q * topk + j] = h.a[j].score;
      if (out_index) out_index[(size_t)q * topk + j] = h.a[j].index;
      ++cnt;
    }
    out_counts[q] = (uint32_t)cnt;
    if (out_scanned) out_scanned[q] = total_scan;
  }
  free(st);
  free(cst);
  return 0;
}

int zo_ivf_search(const float *centroids, uint32_t nlist, const uint64_t *list_offsets,
                  const float *vecs, const uint64_t *keys, uint32_t dim, int metric,
                  const float *queries, uint32_t nq, uint32_t topk, float threshold,
                  uint32_t nprobe, uint32_t max_scan_count, int brute_force,
                  const uint64_t *exclude_bits, uint64_t *out_keys, float *out_scores,
                  uint32_t *out_index, uint32_t *out_counts, uint32_t *out_scanned,
                  uint32_t *out_probes) {
  return ivf_search_range(0, centroids, nlist, list_offsets, vecs, keys, dim, metric, queries, 0, nq,
                          topk, threshold, nprobe, max_scan_count, brute_force, exclude_bits,
                          out_keys, out_scores, out_index, out_counts, out_scanned, out_probes);
}

/* ---- thread fan-out across queries (tools/core/bench.cc:145-245: T workers, one context each,
 * shared index; one query is always scanned by a single thread). */
typedef struct {
  int kind; /* 0 flat, 1 ivf */
  int dtype;
  const void *centroids;
  uint32_t nlist;
  const uint64_t *list_offsets;
  const void *base;
  const uint64_t *keys;
  uint64_t n;
  uint32_t dim;
  int metric;
  const void *queries;
  uint32_t q0, q1, topk;
  float threshold;
  uint32_t nprobe, max_scan;
  int brute_force;
  const uint64_t *exclude_bits;
  uint64_t *out_keys;
  float *out_scores;
  uint32_t *out_index, *out_counts, *out_scanned;
  int rc;
} zo_job;

static void *zo_worker(void *arg) {
  zo_job *j = (zo_job *)arg;
  if (j->kind == 0)
    j->rc = flat_search_range(j->dtype, j->base, j->keys, j->n, j->dim, j->metric, j->queries, j->q0, j->q1,
                              j->topk, j->threshold, j->exclude_bits, j->out_keys, j->out_scores,
                              j->out_index, j->out_counts);
  else
    j->rc = ivf_search_range(j->dtype, j->centroids, j->nlist, j->list_offsets, j->base, j->keys, j->dim,
                             j->metric, j->queries, j->q0, j->q1, j->topk, j->threshold,
                             j->nprobe, j->max_scan, j->brute_force, j->exclude_bits, j->out_keys,
                             j->out_scores, j->out_index, j->out_counts, j->out_scanned, NULL);
  return NULL;
}

static int run_jobs(zo_job proto, uint32_t nq, int threads) {
  if (threads < 1) threads = 1;
  if ((uint32_t)threads > nq) threads = (int)(nq ? nq : 1);
  zo_job *jobs = (zo_job *)malloc(sizeof(zo_job) * (size_t)threads);
  pthread_t *tids = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
  int rc = 0;
  for (int t = 0; t < threads; ++t) {
    jobs[t] = proto;
    jobs[t].q0 = (uint32_t)(((uint64_t)nq * (uint64_t)t) / (uint64_t)threads);
    jobs[t].q1 = (uint32_t)(((uint64_t)nq * (uint64_t)(t + 1)) / (uint64_t)threads);
    jobs[t].rc = 0;
    pthread_create(&tids[t], NULL, zo_worker, &jobs[t]);
  }
  for (int t = 0; t < threads; ++t) {
    pthread_join(tids[t], NULL);
    if (jobs[t].rc != 0) rc = jobs[t].rc;
  }
  free(jobs);
  free(tids);
  return rc;
}

int zo_flat_search_mt(const float *base, const uint64_t *keys, uint64_t n, uint32_t dim,
                      int metric, const float *queries, uint32_t nq, uint32_t topk,
                      float threshold, const uint64_t *exclude_bits, uint64_t *out_keys,
                      float *out_scores, uint32_t *out_index, uint32_t *out_counts, int threads) {
  zo_job p;
  memset(&p, 0, sizeof(p));
  p.kind = 0;
  p.base = base;
  p.keys = keys;
  p.n = n;
  p.dim = dim;
  p.metric = metric;
  p.queries = queries;
  p.topk = topk;
  p.threshold = threshold;
  p.exclude_bits = exclude_bits;
  p.out_keys = out_keys;
  p.out_scores = out_scores;
  p.out_index = out_index;
  p.out_counts = out_counts;
  return run_jobs(p, nq, threads);
}

int zo_ivf_search_mt(const float *centroids, uint32_t nlist, const uint64_t *list_offsets,
                     const float *vecs, const uint64_t *keys, uint32_t dim, int metric,
                     const float *queries, uint32_t nq, uint32_t topk, float threshold,
                     uint32_t nprobe, uint32_t max_scan_count, int brute_force,
                     const uint64_t *exclude_bits, uint64_t *out_keys, float *out_scores,
                     uint32_t *out_index, uint32_t *out_counts, uint32_t *out_scanned,
                     int threads) {
  zo_job p;
  memset(&p, 0, sizeof(p));
  p.kind = 1;
  p.centroids = centroids;
  p.nlist = nlist;
  p.list_offsets = list_offsets;
  p.base = vecs;
  p.keys = keys;
  p.dim = dim;
  p.metric = metric;
  p.queries = queries;
  p.topk = topk;
  p.threshold = threshold;
  p.nprobe = nprobe;
  p.max_scan = max_scan_count;
  p.brute_force = brute_force;
  p.exclude_bits = exclude_bits;
  p.out_keys = out_keys;
  p.out_scores = out_scores;
  p.out_index = out_index;
  p.out_counts = out_counts;
  p.out_scanned = out_scanned;
  return run_jobs(p, nq, threads);
}

int zo_flat_search_mt_t(int dtype, const void *base, const uint64_t *keys, uint64_t n, uint32_t dim, int metric,
                        const void *queries, uint32_t nq, uint32_t topk, float threshold,
                        const uint64_t *exclude_bits, uint64_t *out_keys, float *out_scores, uint32_t *out_index,
                        uint32_t *out_counts, int threads) {
  zo_job p;
  memset(&p, 0, sizeof(p));
  p.kind = 0; p.dtype = dtype; p.base = base; p.keys = keys; p.n = n; p.dim = dim; p.metric = metric;
  p.queries = queries; p.topk = topk; p.threshold = threshold; p.exclude_bits = exclude_bits;
  p.out_keys = out_keys; p.out_scores = out_scores; p.out_index = out_index; p.out_counts = out_counts;
  return run_jobs(p, nq, threads);
}

int zo_ivf_search_mt_t(int dtype, const void *centroids, uint32_t nlist, const uint64_t *list_offsets,
                       const void *vecs, const uint64_t *keys, uint32_t dim, int metric, const void *queries,
                       uint32_t nq, uint32_t topk, float threshold, uint32_t nprobe, uint32_t max_scan_count,
                       int brute_force, const uint64_t *exclude_bits, uint64_t *out_keys, float *out_scores,
                       uint32_t *out_index, uint32_t *out_counts, uint32_t *out_scanned, int threads) {
  zo_job p;
  memset(&p, 0, sizeof(p));
  p.kind = 1; p.dtype = dtype; p.centroids = centroids; p.nlist = nlist; p.list_offsets = list_offsets;
  p.base = vecs; p.keys = keys; p.dim = dim; p.metric = metric; p.queries = queries; p.topk = topk;
  p.threshold = threshold; p.nprobe = nprobe; p.max_scan = max_scan_count; p.brute_force = brute_force;
  p.exclude_bits = exclude_bits; p.out_keys = out_keys; p.out_scores = out_scores; p.out_index = out_index;
  p.out_counts = out_counts; p.out_scanned = out_scanned;
  return run_jobs(p, nq, threads);
}

/* ======================================================================================== *
 * Partial-result merge: CombinedVectorColumnIndexer::Search
 * (combined_vector_column_indexer.cc:91-232): concatenate the per-block lists (block order),
 * sort by score, truncate to topk.  Scores here are boundary-B scores (smaller is better for
 * every metric), so the sort is ascending; ties keep concatenation order (the reference's
 * std::sort leaves them unspecified).
 * ======================================================================================== */
typedef struct {
  uint64_t key;
  float score;
  uint32_t ord;
} zo_mrec;

static int mrec_cmp(const void *pa, const void *pb) {
  const zo_mrec *a = (const zo_mrec *)pa, *b = (const zo_mrec *)pb;
  if (a->score < b->score) return -1;
  if (a->score > b->score) return 1;
  if (a->ord < b->ord) return -1;
  if (a->ord > b->ord) return 1;
  return 0;
}

int zo_merge_topk(const uint64_t *keys, const float *scores, const uint32_t *counts,
                  uint32_t nparts, uint32_t nq, uint32_t topk, uint64_t *out_keys,
                  float *out_scores, uint32_t *out_counts) {
  zo_mrec *buf = (zo_mrec *)malloc(sizeof(zo_mrec) * (size_t)nparts * topk);
  if (!buf) return -2;
  for (uint32_t q = 0; q < nq; ++q) {
    size_t m = 0;
    for (uint32_t p = 0; p < nparts; ++p) {
      uint32_t c = counts[(size_t)p * nq + q];
      for (uint32_t j = 0; j < c; ++j) {
        size_t off = ((size_t)p * nq + q) * topk + j;
        buf[m].key = keys[off];
        buf[m].score = scores[off];
        buf[m].ord = (uint32_t)m;
        ++m;
      }
    }
    qsort(buf, m, sizeof(zo_mrec), mrec_cmp);
    size_t c = m < topk ? m : topk;
    for (size_t j = 0; j < c; ++j) {
      out_keys[(size_t)q * topk + j] = buf[j].key;
      out_scores[(size_t)q * topk + j] = buf[j].score;
    }
    out_counts[q] = (uint32_t)c;
  }
  free(buf);
  return 0;
}
