/*
 * woq_cpu_port.c — the reference's int4 weight-only decode GEMV restated for the HOST cores, on a CPU-friendly
 * layout. bench.py's `cpu_baseline` leg times this ("kind": "port"); it is also checked against the oracle's
 * definition (tests/test_oracle_golden.py). TEST / MEASUREMENT INFRASTRUCTURE ONLY — never linked into the product.
 *
 * What it restates. The reference's fp32 compute path runs BesTLA's `SCoreRowNAvx512f<48, 8>` core through
 * `WeightKBlockNInteger` (qbits/dispatcher/src/bestla_weightonly_dispatcher.cpp:151-177,237,316): per N tile x
 * K block, unpack int4 -> fp32, FMA against the broadcast activation, apply the per-(group, column) scale, and for
 * asymmetric weights subtract zp * scale * sum_k x_k with the row sums from the activation prologue (:154-160,
 * 205-207). BesTLA itself is not in /root/reference (fetched by CMake FetchContent, neural_speed.cmake:1-9), so this
 * is a port of that published scheme, not the reference binary; the only reference-kernel number available is the
 * published 35.84 ms / token (MPT-7B, 56-core Xeon 8480+, docs/release_data.md:131).
 *
 * Layout "CPK" (what a CPU kernel wants, like BesTLA's own N-tile-major repack): the matrix is cut into blocks of 32
 * output columns; a block stores its K rows contiguously, 16 bytes per row: byte j = u(k, n0 + j) | u(k, n0 + 16 + j)
 * << 4 with u = q + 8 the unsigned nibble. One thread streams one block: K x 16 contiguous bytes, 32 accumulators in
 * registers, no gathers. sum_k (q - zp) x = sum_k u x - (8 + zp) sum_k x per group.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

ORC_API size_t cpk_bytes(int K, int N) { return (size_t)((N + 31) / 32) * (size_t)K * 16u; }

/* q int8 [K][N] in the signed domain -> CPK. Parallel over blocks with the same static schedule as the GEMV, so on
 * a NUMA host every block is first touched by the thread that later streams it. */
ORC_API void cpk_pack(const int8_t* q, int K, int N, uint8_t* out) {
  const int nblk = (N + 31) / 32;
#pragma omp parallel for schedule(static)
  for (int nb = 0; nb < nblk; ++nb) {
    uint8_t* wb = out + (size_t)nb * K * 16u;
    for (int k = 0; k < K; ++k)
      for (int j = 0; j < 16; ++j) {
        const int n_lo = nb * 32 + j, n_hi = n_lo + 16;
        const unsigned lo = n_lo < N ? (unsigned)(q[(size_t)k * N + n_lo] + 8) & 15u : 8u;
        const unsigned hi = n_hi < N ? (unsigned)(q[(size_t)k * N + n_hi] + 8) & 15u : 8u;
        wb[(size_t)k * 16u + j] = (uint8_t)(lo | (hi << 4));
      }
  }
}

/* synthetic weights for the timing leg: every byte = two random nibbles, filled by the thread that will read it */
ORC_API void cpk_fill_random(uint8_t* out, int K, int N, uint64_t seed) {
  const int nblk = (N + 31) / 32;
#pragma omp parallel for schedule(static)
  for (int nb = 0; nb < nblk; ++nb) {
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + (uint64_t)nb * 0xBF58476D1CE4E5B9ull + 1u;
    uint64_t* w = (uint64_t*)(out + (size_t)nb * K * 16u);
    for (size_t i = 0; i < (size_t)K * 2u; ++i) {
      s ^= s << 13;
      s ^= s >> 7;
      s ^= s << 17;
      w[i] = s;
    }
  }
}

/* one 32-column block, all of K: plain C (any host), and the same loop with AVX2 / AVX-512 intrinsics. Every
 * variant accumulates a group in fp32 in k order per column, so they agree to fp32 rounding. */
typedef void (*cpk_block_fn)(int nb, const float* x, const uint8_t* w, const float* scales, const int8_t* zp,
                             const float* bias, int K, int N, int group, float* out);

/* group epilogue shared by the variants: tot += scale * (acc - (8 + zp) * sum_x) */
static inline void cpk_fold(float* tot, const float* acc, float sx, int g, int nb, const float* scales,
                            const int8_t* zp, int N) {
  for (int j = 0; j < 32; ++j) {
    const int n = nb * 32 + j;
    if (n >= N) break;
    const float z = 8.f + (zp ? (float)zp[(size_t)g * N + n] : 0.f);
    tot[j] += scales[(size_t)g * N + n] * (acc[j] - z * sx);
  }
}

static void cpk_block_plain(int nb, const float* x, const uint8_t* w, const float* scales, const int8_t* zp,
                            const float* bias, int K, int N, int group, float* out) {
  float tot[32];
  for (int j = 0; j < 32; ++j) tot[j] = 0.f;
  const uint8_t* wb = w + (size_t)nb * K * 16u;
  const int G = (K + group - 1) / group;
  for (int g = 0; g < G; ++g) {
    float acc[32];
    for (int j = 0; j < 32; ++j) acc[j] = 0.f;
    float sx = 0.f;
    const int k1 = (g + 1) * group < K ? (g + 1) * group : K;
    for (int k = g * group; k < k1; ++k) {
      const float xv = x[k];
      const uint8_t* b = wb + (size_t)k * 16u;
      sx += xv;
      for (int j = 0; j < 16; ++j) {
        acc[j] += (float)(b[j] & 15) * xv;
        acc[16 + j] += (float)(b[j] >> 4) * xv;
      }
    }
    cpk_fold(tot, acc, sx, g, nb, scales, zp, N);
  }
  for (int j = 0; j < 32 && nb * 32 + j < N; ++j) out[nb * 32 + j] = tot[j] + (bias ? bias[nb * 32 + j] : 0.f);
}

#include <immintrin.h>

__attribute__((target("avx512f,avx512bw,avx512vl,fma"))) static void cpk_block_avx512(
    int nb, const float* x, const uint8_t* w, const float* scales, const int8_t* zp, const float* bias, int K, int N,
    int group, float* out) {
  float tot[32];
  for (int j = 0; j < 32; ++j) tot[j] = 0.f;
  const uint8_t* wb = w + (size_t)nb * K * 16u;
  const int G = (K + group - 1) / group;
  const __m128i m4 = _mm_set1_epi8(15);
  for (int g = 0; g < G; ++g) {
    __m512 a0 = _mm512_setzero_ps(), a1 = _mm512_setzero_ps(), c0 = _mm512_setzero_ps(), c1 = _mm512_setzero_ps();
    float sx = 0.f;
    const int k0 = g * group, k1 = (g + 1) * group < K ? (g + 1) * group : K;
    int k = k0;
    for (; k + 1 < k1; k += 2) { /* two rows per trip: four independent FMA chains */
      const __m128i b0 = _mm_loadu_si128((const __m128i*)(wb + (size_t)k * 16u));
      const __m128i b1 = _mm_loadu_si128((const __m128i*)(wb + (size_t)(k + 1) * 16u));
      const __m512 x0 = _mm512_set1_ps(x[k]), x1 = _mm512_set1_ps(x[k + 1]);
      sx += x[k] + x[k + 1];
      a0 = _mm512_fmadd_ps(_mm512_cvtepi32_ps(_mm512_cvtepu8_epi32(_mm_and_si128(b0, m4))), x0, a0);
      a1 = _mm512_fmadd_ps(_mm512_cvtepi32_ps(_mm512_cvtepu8_epi32(_mm_and_si128(_mm_srli_epi16(b0, 4), m4))), x0, a1);
      c0 = _mm512_fmadd_ps(_mm512_cvtepi32_ps(_mm512_cvtepu8_epi32(_mm_and_si128(b1, m4))), x1, c0);
      c1 = _mm512_fmadd_ps(_mm512_cvtepi32_ps(_mm512_cvtepu8_epi32(_mm_and_si128(_mm_srli_epi16(b1, 4), m4))), x1, c1);
    }
    for (; k < k1; ++k) {
      const __m128i b0 = _mm_loadu_si128((const __m128i*)(wb + (size_t)k * 16u));
      const __m512 x0 = _mm512_set1_ps(x[k]);
      sx += x[k];
      a0 = _mm512_fmadd_ps(_mm512_cvtepi32_ps(_mm512_cvtepu8_epi32(_mm_and_si128(b0, m4))), x0, a0);
      a1 = _mm512_fmadd_ps(_mm512_cvtepi32_ps(_mm512_cvtepu8_epi32(_mm_and_si128(_mm_srli_epi16(b0, 4), m4))), x0, a1);
    }
    float acc[32];
    _mm512_storeu_ps(acc, _mm512_add_ps(a0, c0));
    _mm512_storeu_ps(acc + 16, _mm512_add_ps(a1, c1));
    cpk_fold(tot, acc, sx, g, nb, scales, zp, N);
  }
  for (int j = 0; j < 32 && nb * 32 + j < N; ++j) out[nb * 32 + j] = tot[j] + (bias ? bias[nb * 32 + j] : 0.f);
}

__attribute__((target("avx2,fma"))) static void cpk_block_avx2(int nb, const float* x, const uint8_t* w,
                                                               const float* scales, const int8_t* zp,
                                                               const float* bias, int K, int N, int group,
                                                               float* out) {
  float tot[32];
  for (int j = 0; j < 32; ++j) tot[j] = 0.f;
  const uint8_t* wb = w + (size_t)nb * K * 16u;
  const int G = (K + group - 1) / group;
  const __m128i m4 = _mm_set1_epi8(15);
  for (int g = 0; g < G; ++g) {
    __m256 a0 = _mm256_setzero_ps(), a1 = _mm256_setzero_ps(), a2 = _mm256_setzero_ps(), a3 = _mm256_setzero_ps();
    float sx = 0.f;
    const int k1 = (g + 1) * group < K ? (g + 1) * group : K;
    for (int k = g * group; k < k1; ++k) {
      const __m128i b = _mm_loadu_si128((const __m128i*)(wb + (size_t)k * 16u));
      const __m128i lo = _mm_and_si128(b, m4), hi = _mm_and_si128(_mm_srli_epi16(b, 4), m4);
      const __m256 xv = _mm256_set1_ps(x[k]);
      sx += x[k];
      a0 = _mm256_fmadd_ps(_mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(lo)), xv, a0);
      a1 = _mm256_fmadd_ps(_mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(_mm_srli_si128(lo, 8))), xv, a1);
      a2 = _mm256_fmadd_ps(_mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(hi)), xv, a2);
      a3 = _mm256_fmadd_ps(_mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(_mm_srli_si128(hi, 8))), xv, a3);
    }
    float acc[32];
    _mm256_storeu_ps(acc, a0);
    _mm256_storeu_ps(acc + 8, a1);
    _mm256_storeu_ps(acc + 16, a2);
    _mm256_storeu_ps(acc + 24, a3);
    cpk_fold(tot, acc, sx, g, nb, scales, zp, N);
  }
  for (int j = 0; j < 32 && nb * 32 + j < N; ++j) out[nb * 32 + j] = tot[j] + (bias ? bias[nb * 32 + j] : 0.f);
}

/* 2 = avx512, 1 = avx2, 0 = plain: what the dispatcher picked on this host (reported in the bench line) */
ORC_API int cpk_isa(void) {
  __builtin_cpu_init();
  if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl"))
    return 2;
  if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) return 1;
  return 0;
}

/* out[N] = x[K] . W_deq (+ bias): scales fp32 [G][N], zp int8 [G][N] (signed domain) or NULL, group along K */
ORC_API void cpk_gemv(const float* x, const uint8_t* w, const float* scales, const int8_t* zp, const float* bias,
                      int K, int N, int group, float* out) {
  if (group <= 0 || group > K) group = K;
  const int isa = cpk_isa();
  const cpk_block_fn fn = isa == 2 ? cpk_block_avx512 : (isa == 1 ? cpk_block_avx2 : cpk_block_plain);
  const int nblk = (N + 31) / 32;
#pragma omp parallel for schedule(static)
  for (int nb = 0; nb < nblk; ++nb) fn(nb, x, w, scales, zp, bias, K, N, group, out);
}
