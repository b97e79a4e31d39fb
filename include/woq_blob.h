/*
 * woq_blob.h — on-device packed-weight blob ("WQH1") for the MI355X int4 WOQ path.
 *
 * Replaces the opaque BesTLA StorageWeightKBlockNInteger blob that the reference
 * creates in qbits.repack_quantized_weight
 *   (intel_extension_for_transformers/qbits/qbits.cpp:61-77,
 *    qbits/dispatcher/src/bestla_packq_impl.cpp:20-41)
 * and re-parses on every call (bestla_weightonly_dispatcher.cpp:335-340).
 * The layout is opaque to callers of the reference too, so it is designed
 * for gfx950, not copied: one 1-KiB "tile" = 16 output columns x 128 K rows
 * is exactly one wave64 `buffer_load_dwordx4` (64 lanes x 16 B), and the lane
 * <-> (column, k) map inside a tile is the B-operand fragment map of
 * v_mfma_i32_16x16x64_i8 (lane l: column l & 15, 16 consecutive k starting at
 * (l >> 4) * 16), with every int4 stored as a SIGNED two's-complement nibble, so
 * that `(w << 4) & 0xf0f0f0f0` and `w & 0xf0f0f0f0` are directly four int8
 * values 16*q each: three VALU instructions turn 8 weights into MFMA operands.
 *
 * Blob = [256-B header][qdata][scales][zero points][shuffle indices]
 *
 * qdata   : [Npad/16][Kpad/128][64 lanes][4 x u32]
 *           lane l: column i = l & 15, k-sixteenth kq = l >> 4
 *           u32 #w (0..3): 64-k half h = w >> 1, part p = w & 1; it holds k-offsets j = 8p .. 8p+7 of
 *               k = kt*128 + h*64 + kq*16 + j,  n = tn*16 + i
 *           k-offset j = 8p + jj sits in byte c = jj & 3, low nibble for jj < 4, high nibble for jj >= 4
 *           (bit shift 8*c + 4*(jj >> 2)): the low nibbles of the 4 bytes are j = 8p..8p+3, the high
 *           nibbles j = 8p+4..8p+7 — B registers 2p and 2p+1 of the MFMA fragment.
 *           nibble value = q & 0xf, q in [-8,7] (the reference's signed-nibble domain,
 *           llm/quantization/nn/modules.py:225-227); padding (k >= K or n >= N) is q = 0.
 * scales  : scale_mode 0 (group % 128 == 0, or a single group):
 *               [Npad/16][n_groups][16]            element (tn, g, i)
 *           scale_mode 1 (any other group that is a multiple of 32):
 *               [Npad/16][Kpad/128][16][4]         element (tn, kt, i, s) = scale of
 *               the group that holds k = kt*128 + s*32 (expanded per 32-block)
 *           element type per header.scale_type (fp32 | bf16 | fp16); padding = 0.
 * zeros   : same indexing as scales, uint8, value uz = zp + 8 where zp is the
 *           signed-domain zero point handed to repack (range [-8, 7]: the reference's
 *           (zeros - 8) * 16 // 16 on int8 wraps 16 -> -8, modules.py:225-227, pinned by
 *           tests/golden/set_weights_bias.npz); absent when !asym.
 * shuffle : int32[K] activation shuffle indices (x'[j] = x[idx[j]]): convert_idx of the
 *           raw GPTQ g_idx the caller passed (bestla_packq_impl.cpp:37-38; what
 *           acquire_packed_weight_info(G_IDX) returns, qbits_ut/test_packq.py:98-100);
 *           absent otherwise.
 *
 * Dequantisation: w[k][n] = (q - (uz - 8)) * scale     (uz = 8 when !asym)
 *   == (q - zp) * scale of modules.py:264-295 / recover_qparms :349-352.
 */
#ifndef WOQ_BLOB_H_
#define WOQ_BLOB_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WOQ_BLOB_MAGIC 0x31485157u /* "WQH1" */
#define WOQ_BLOB_VERSION 2u
#define WOQ_HEADER_BYTES 256
#define WOQ_TILE_N 16
#define WOQ_TILE_K 128
#define WOQ_TILE_BYTES 1024

/* element types crossing the ABI (reference: qbits.cpp:31-37 allows fp32|bf16; fp16 added
 * because north_star asks for bf16/fp16 activations) */
enum woq_dtype { WOQ_F32 = 0, WOQ_BF16 = 1, WOQ_F16 = 2,
                 WOQ_FP8_E4M3 = 3 /* OCP e4m3fn; KV-cache storage only (woq_engine_config.kv_dtype) */ };

/* weight types (reference strings: bestla_weightonly_dispatcher.hpp:62-70) */
enum woq_weight_type {
  WOQ_W_INT4_CLIP = 0,
  WOQ_W_INT8 = 1,         /* composite of two int4 blobs, see woq_int8_headers */
  WOQ_W_NF4 = 2,          /* 4-bit table types: the nibble is an INDEX into a 16-entry table, w = table[code] * scale, */
  WOQ_W_FP4_E2M1 = 3,     /* symmetric only (no zero points), same qdata layout as int4                              */
  WOQ_W_FP4_E2M1_BNB = 4,
  WOQ_W_INT3_CLIP = 5,    /* narrow integers (reference strings "int3_clip" / "int2_clip"): values in [-4, 3] / [-2, 1]  */
  WOQ_W_INT2_CLIP = 6,    /* held in int4 storage — names at the ABI only: the blob header says INT4_CLIP plus          */
                          /* narrow_bits = 3 / 2, and every kernel runs its int4 path on it unchanged                    */
  WOQ_W_FP8_E4M3 = 7,     /* 8-bit float codes (OCP e4m3fn / e5m2), w = value(code) * scale, symmetric only: the code  */
  WOQ_W_FP8_E5M2 = 8      /* byte is stored as two nibble planes in the int8 composite layout, see woq_fp8_headers     */
};
static inline int woq_weight_is_fp8(uint32_t t) { return t == 7u || t == 8u; }
/* largest finite magnitude: RTN scale = absmax / max */
static inline float woq_fp8_max(uint32_t t) { return t == 7u ? 448.0f : 57344.0f; }
/* scale_type value accepted by the pack entry points for fp8 weights only (reference string "fp8_e8m0",
 * qbits_ut/test_weightonly.py:24-25): power-of-two scales. Stored as bf16 (exact for every power of two in range);
 * the header says scale_type = WOQ_BF16 and sets WOQ_FLAG_SCALE_E8M0 so the name survives. */
#define WOQ_SCALE_FP8_E8M0 4
static inline int woq_weight_is_table(uint32_t t) { return t >= 2u && t <= 4u; }
static inline int woq_weight_narrow_bits(uint32_t t) { return t == 5u ? 3 : (t == 6u ? 2 : 0); }

/* The tables (reference strings "nf4", "fp4_e2m1", "fp4_e2m1_bnb", bestla_weightonly_dispatcher.hpp:62-70; BesTLA's
 * own constants are not in the reference tree, so these are the published definitions: NF4 = the 16 normal-float
 * quantiles of bitsandbytes; e2m1 = sign | 2-bit exponent | 1-bit mantissa, values 0, .5, 1, 1.5, 2, 3, 4, 6;
 * the bitsandbytes fp4 variant = its sign-magnitude table over 12). `max` = the largest magnitude: scale = absmax / max. */
#define WOQ_LUT_NF4                                                                                                   \
  {-1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f, -0.28444138169288635f,                   \
   -0.18477343022823334f, -0.09105003625154495f, 0.0f, 0.07958029955625534f, 0.16093020141124725f,                    \
   0.24611230194568634f, 0.33791524171829224f, 0.44070982933044434f, 0.5626170039176941f, 0.7229568362236023f, 1.0f}
#define WOQ_LUT_FP4_E2M1 \
  {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f, -0.f, -0.5f, -1.f, -1.5f, -2.f, -3.f, -4.f, -6.f}
#define WOQ_LUT_FP4_BNB                                                                                               \
  {0.f, 0.0625f / 12.f, 8.f / 12.f, 1.f, 4.f / 12.f, 0.5f, 2.f / 12.f, 0.25f, -0.f,                      \
   -0.0625f / 12.f, -8.f / 12.f, -1.f, -4.f / 12.f, -0.5f, -2.f / 12.f, -0.25f}
static inline float woq_lut_max(uint32_t t) { return t == 3u ? 6.0f : 1.0f; }

/* compute types (reference strings "fp32" | "bf16" | "int8"; recorded) */
enum woq_compute_type { WOQ_C_FP32 = 0, WOQ_C_BF16 = 1, WOQ_C_INT8 = 2, WOQ_C_FP16 = 3 };

#define WOQ_FLAG_ASYM 1u
#define WOQ_FLAG_ACT_SHUFFLE 2u
#define WOQ_FLAG_SCALE_E8M0 4u

typedef struct woq_blob_header {
  uint32_t magic;
  uint32_t version;
  uint64_t total_bytes;
  int32_t K, N;
  int32_t group;     /* resolved group size (blocksize -1 -> K) */
  int32_t Kpad;      /* K rounded up to 128 */
  int32_t Npad;      /* N rounded up to 16 */
  int32_t n_groups;  /* ceil(K / group) */
  uint32_t weight_type;
  uint32_t scale_type;   /* enum woq_dtype */
  uint32_t compute_type; /* enum woq_compute_type */
  uint32_t flags;
  uint32_t scale_mode; /* 0 | 1, see above */
  uint32_t narrow_bits; /* 0, or 3 / 2: the blob was packed as int3_clip / int2_clip (weight_type is INT4_CLIP) */
  uint64_t off_q, off_scale, off_zp, off_shuffle; /* byte offsets from blob start; 0 = absent */
  uint8_t pad[WOQ_HEADER_BYTES - 96];
} woq_blob_header;

#ifdef __cplusplus
static_assert(sizeof(woq_blob_header) == WOQ_HEADER_BYTES, "header must be 256 B");
#else
_Static_assert(sizeof(woq_blob_header) == WOQ_HEADER_BYTES, "header must be 256 B");
#endif

/* acquire_packed_weight_info selector — same numbering as the reference's
 * PACKW_ACQUIRE_TYPE (qbits/dispatcher/include/bestla_packq_impl.hpp:18-31). */
enum woq_acquire_type {
  WOQ_ACQ_SIZE = 0,
  WOQ_ACQ_BLOCKSIZE = 1,
  WOQ_ACQ_K = 2,
  WOQ_ACQ_N = 3,
  WOQ_ACQ_ACT_SHUFFLE = 4,
  WOQ_ACQ_G_IDX = 5,
  WOQ_ACQ_WEI_TYPE = 6,
  WOQ_ACQ_CMPT_TYPE = 7,
  WOQ_ACQ_SCALE_TYPE = 8,
  WOQ_ACQ_SCALE_TENSOR = 9,
  WOQ_ACQ_ZP_TENSOR = 10,
  WOQ_ACQ_IS_ASYM = 11
};

/* bit shift of k-offset j (0..15 within a lane's 16-k run of one 64-k half) inside its u32 (#2h + (j >> 3)) */
static inline int woq_nibble_shift(int j) {
  int jj = j & 7;
  return 8 * (jj & 3) + 4 * (jj >> 2);
}

static inline size_t woq_dtype_size(uint32_t dt) { return dt == WOQ_F32 ? 4u : (dt == WOQ_FP8_E4M3 ? 1u : 2u); }
static inline size_t woq_round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

/* Fill every derived field of a header from (K, N, group, types, flags).
 * Returns 0, or -1 if the geometry is unsupported (group not a multiple of 32
 * and more than one group). Pure host arithmetic, shared by the HIP library
 * and by the CPU oracle so both agree on offsets by construction. */
static inline int woq_header_init(woq_blob_header* h, int K, int N, int group, uint32_t weight_type,
                                  uint32_t scale_type, uint32_t compute_type, int asym, int act_shuffle) {
  if (K <= 0 || N <= 0) return -1;
  if (group <= 0 || group > K) group = K;
  int n_groups = (K + group - 1) / group;
  if (n_groups > 1 && (group % 32) != 0) return -1;
  for (size_t i = 0; i < sizeof(*h); ++i) ((uint8_t*)h)[i] = 0;
  h->magic = WOQ_BLOB_MAGIC;
  h->version = WOQ_BLOB_VERSION;
  h->K = K;
  h->N = N;
  h->group = group;
  h->Kpad = (int32_t)woq_round_up((size_t)K, WOQ_TILE_K);
  h->Npad = (int32_t)woq_round_up((size_t)N, WOQ_TILE_N);
  h->n_groups = n_groups;
  h->weight_type = weight_type;
  h->scale_type = scale_type;
  h->compute_type = compute_type;
  h->flags = (asym ? WOQ_FLAG_ASYM : 0u) | (act_shuffle ? WOQ_FLAG_ACT_SHUFFLE : 0u);
  h->scale_mode = (n_groups == 1 || (group % WOQ_TILE_K) == 0) ? 0u : 1u;
  size_t tiles_n = (size_t)h->Npad / WOQ_TILE_N, tiles_k = (size_t)h->Kpad / WOQ_TILE_K;
  size_t n_scale = h->scale_mode == 0 ? tiles_n * (size_t)n_groups * 16u : tiles_n * tiles_k * 64u;
  size_t off = WOQ_HEADER_BYTES;
  h->off_q = off;
  off += tiles_n * tiles_k * WOQ_TILE_BYTES;
  h->off_scale = off;
  off = woq_round_up(off + n_scale * woq_dtype_size(scale_type), 256);
  if (asym) {
    h->off_zp = off;
    off = woq_round_up(off + n_scale, 256);
  }
  if (act_shuffle) {
    h->off_shuffle = off;
    off = woq_round_up(off + (size_t)K * 4u, 256);
  }
  h->total_bytes = off;
  return 0;
}

/* int8 weights (WOQ_W_INT8; reference weight type "int8", bestla_weightonly_dispatcher.hpp:62) are stored as a
 * COMPOSITE of two complete int4 blobs, because
 *     q8 - zp8 = 16 (hi - zhi) + (lo - zlo),   hi = q8 >> 4 (arithmetic), lo = (q8 & 15) - 8,
 *                                              zhi = zp8 >> 4,            zlo = (zp8 & 15) - 8
 * holds exactly with all four terms in the int4 range [-8, 7]: (q8 - zp8) * s = (hi - zhi) * 16s + (lo - zlo) * s,
 * i.e. an int8 linear is the sum of two int4 linears on the same activations — computed by the SAME exact kernels,
 * the second launch adding the first's fp32 result. Layout: [256-B outer header][HI int4 blob: scales 16s, zero
 * points zhi when asym][LO int4 blob: scales s, zero points zlo, always asym]. Outer header: weight_type INT8, the
 * usual geometry, off_q = offset of the HI blob, off_scale = offset of the LO blob, off_zp / off_shuffle = non-zero
 * (offset of the LO blob's section) when asymmetric / act-shuffled. Same bytes per weight as a flat int8 layout. */
static inline int woq_int8_headers(woq_blob_header* outer, woq_blob_header* hi, woq_blob_header* lo, int K, int N,
                                   int group, uint32_t scale_type, uint32_t compute_type, int asym, int act_shuffle) {
  if (woq_header_init(hi, K, N, group, 0u /* int4_clip */, scale_type, compute_type, asym, act_shuffle) != 0) return -1;
  if (woq_header_init(lo, K, N, group, 0u, scale_type, compute_type, 1, act_shuffle) != 0) return -1;
  *outer = *hi;
  outer->weight_type = 1u; /* WOQ_W_INT8 */
  outer->off_q = WOQ_HEADER_BYTES;
  outer->off_scale = WOQ_HEADER_BYTES + hi->total_bytes;
  outer->off_zp = asym ? outer->off_scale + lo->off_zp : 0;
  outer->off_shuffle = act_shuffle ? outer->off_scale + lo->off_shuffle : 0;
  outer->total_bytes = WOQ_HEADER_BYTES + hi->total_bytes + lo->total_bytes;
  return 0;
}

/* fp8 weights (WOQ_W_FP8_E4M3 / WOQ_W_FP8_E5M2): the same composite container as int8 — [outer header][HI blob][LO
 * blob], both in the int4 tile layout — holding the two nibbles of every code byte c: the HI plane stores c >> 4, the
 * LO plane (c & 15) ^ 8 (what the int8 splitter produces for the byte read as a signed value), so
 *     c = (hi_nibble << 4) | (lo_nibble ^ 8).
 * Both planes carry the scales s (the LO plane's zero-point section exists and is unused); symmetric only. A table
 * lookup is not linear in the nibbles, so unlike int8 the two planes are read TOGETHER by one kernel. */
static inline int woq_fp8_headers(woq_blob_header* outer, woq_blob_header* hi, woq_blob_header* lo, int K, int N,
                                  int group, uint32_t weight_type, uint32_t scale_type, uint32_t compute_type,
                                  int act_shuffle) {
  if (woq_int8_headers(outer, hi, lo, K, N, group, scale_type, compute_type, 0, act_shuffle) != 0) return -1;
  outer->weight_type = weight_type;
  return 0;
}

/* index of the scale / zero-point element for (column n, row k) */
static inline size_t woq_scale_index(const woq_blob_header* h, int k, int n) {
  size_t tn = (size_t)n / WOQ_TILE_N, i = (size_t)n % WOQ_TILE_N;
  if (h->scale_mode == 0) {
    size_t g = (size_t)k / (size_t)h->group;
    if (g >= (size_t)h->n_groups) g = (size_t)h->n_groups - 1;
    return (tn * (size_t)h->n_groups + g) * 16u + i;
  }
  size_t kt = (size_t)k / WOQ_TILE_K, s = ((size_t)k % WOQ_TILE_K) / 32u;
  return ((tn * ((size_t)h->Kpad / WOQ_TILE_K) + kt) * 16u + i) * 4u + s;
}

/* byte offset (from off_q) and nibble shift (0 | 4) of weight element (k, n) */
static inline size_t woq_q_byte(const woq_blob_header* h, int k, int n, int* shift) {
  size_t tn = (size_t)n / WOQ_TILE_N, i = (size_t)n % WOQ_TILE_N;
  size_t kt = (size_t)k / WOQ_TILE_K, r = (size_t)k % WOQ_TILE_K;
  size_t hh = r / 64u, kq = (r % 64u) / 16u, j = r % 16u;
  size_t lane = kq * 16u + i;
  size_t word = ((tn * ((size_t)h->Kpad / WOQ_TILE_K) + kt) * 64u + lane) * 4u + hh * 2u + j / 8u;
  int sh = woq_nibble_shift((int)j);
  *shift = sh & 4;
  return word * 4u + (size_t)(sh >> 3);
}

#ifdef __cplusplus
}
#endif
#endif /* WOQ_BLOB_H_ */
