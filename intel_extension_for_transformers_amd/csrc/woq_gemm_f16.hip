// woq_gemm_f16.hip — prefill-side int4 weight x activation GEMM on the matrix cores, M > 8: one fp16 product per
// operand pair for the reduced-precision compute modes (compute_dtype bf16 / fp16 / int8 in the blob header), three
// (hi + lo operands, fp32-class) for compute_dtype fp32.
//
// Replaces the arithmetic behind qbits.woq_linear at large M for those modes: qbits.cpp:113-140 ->
// bestla_weightonly_dispatcher.cpp:150-178, where the reference's HCoreRowNAmxbf16 core rounds BOTH operands to
// bf16 (8-bit significand) before the AMX tile product. Parity definition: autograd/functions.py:41-63 at the
// reference's own tolerance for these modes (qbits_ut/test_weightonly.py:82-88, rtol 0.03).
//
// Both operands are fp16 (11-bit significand, 8x tighter than the reference's bf16 operands; for compute_dtype fp32 a
// hi + lo PAIR of fp16 each, ~22 bits) and every scale that can be is folded into them, so
// the K loop is nothing but loads, a short dequantisation and MFMAs — v_mfma_f32_16x16x32_f16 accumulating ONE
// fp32 fragment set over all of K:
//  * A: a pack pass turns the activations into fp16 with one power-of-two scale PER ROW (exact scaling; elements
//    below rowmax * 2^-28 go subnormal — far under the 2^-9 relative steps of a bf16 activation), optionally gathers
//    the GPTQ act-order shuffle and applies RMSNorm on the way, and writes them as [M/128][K/128] tiles of
//    128 rows x 16 chunks x 8 halves = 32 KiB in exactly the image the workgroup wants in LDS: chunk c of row r sits
//    in 16-byte slot c ^ (r & 15) (every 16-lane ds_read_b128 group then covers all 64 banks), k inside a chunk in the
//    nibble-extraction order {0,2,4,6,1,3,5,7}. The GEMM moves a tile with 32 LDS-DMA instructions
//    (global_load_lds_dwordx4, lane-linear destination) — no staging registers, no ds_write pass;
//  * B: each wave loads ITS two 1-KiB blob tiles per K step straight to VGPRs. (w ^ 0x88888888) makes the signed
//    nibbles unsigned u; (x & 0x000f000f) | 0x64006400 = fp16 (1024 + u) pairs and (x & 0x00f000f0) | 0x54005400 =
//    fp16 (64 + u) pairs (the nibble sits 4 bits up, where 64.0's ulp is 1/16) — one v_pk_add_f16 of -(base + zp)
//    gives q - zp exactly, one v_pk_mul_f16 by the group's RELATIVE scale r = s / 2^E[col] <= 1 finishes the operand:
//    14 VALU per 8 x 16 fragment, each fragment feeding 8 MFMAs. Group-32 scales cost nothing extra: the lane
//    quarter kq of the MFMA for 64-k half h holds k in [64h + 16kq, 64h + 16kq + 16), i.e. group 2h + (kq >> 1), so a
//    lane simply carries its own group's r and zero point;
//  * epilogue: acc * 2^e[row] * 2^E[col] + bias, adjacent columns paired through DPP so stores are 4 (16-bit out)
//    or 8 bytes wide.
// Workgroup 128 x 128 (4 waves, wave w = all rows x columns [32w, 32w+32) -> 8 x 2 fragments = 64 accumulator VGPRs),
// two LDS buffers of 32 KiB, two workgroups per CU — the kernel in THIS file (hipcc's instruction order; kept for the
// fp32-class form and for odd K-tile counts). The one-product form normally runs the hand-scheduled K loop of
// woq_gemm_f16p.h: same tiles and epilogue, 16-KiB half-tiles through a ring of three LDS slots, three workgroups
// per CU, optionally fetching fp16 activation rows without a pack pass. Workgroup ids are laid out XCD-aware: the 64 workgroups that
// share an XCD's L2 at a time form an 8 x 8 super-tile (8 A row blocks x 8 B column blocks re-used 8x each).
// Measured alternatives (MI355X, M = 8192, gate/up shape, 849 TFLOP/s as built): 8 waves per workgroup sharing one
// A tile (128 x 256, one workgroup per CU) 799; 4 column tiles per wave (128 accumulator VGPRs, one workgroup per CU)
// 474; knock-outs of the built kernel: no dequantisation 1012, no A-tile DMA 997, no barrier 892 — i.e. ~17 % of
// the time is the 14-VALU dequantisation, ~15 % the LDS-DMA issue / landing, ~5 % barrier skew.
#include <algorithm>
#include <mutex>
#include <type_traits>

#include "woq_device.h"
#include "woq_launch.h"

namespace woq {
static hipEvent_t g_gemm_ev0 = nullptr, g_gemm_ev1 = nullptr;


typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

struct GemmF16Args {
  const u32x4* q;
  const void* scales;
  const uint8_t* zp;
  int K, N, tiles_k, tiles_n, n_groups, group, scale_type;
  const _Float16* ap;  // packed activation tiles [mb][kt][planes][128][16 slots][8] (planes: hi, and lo for NP == 3)
  const void* act_raw; // raw-A form (woq_gemm_f16p.h): the caller's row-major fp16 activations, no pack pass; else null
  int lda;             // its row stride in elements
  int ring;            // 1: half-tile ring form of the hand-scheduled kernel (packed tiles in the half-tile layout)
  const float* rs;     // [Mpad] 2^e per row; null = all 1 (raw-A)
  const float* cs;     // [Npad] 2^E per column
  int M, nb_m, nb_n, sup_n, n_sup;
  int nb_m128;         // 128-row blocks of the packed layout (gemm_f16t_kernel counts nb_m / n_sup in 256-row blocks)
  void* out;
  int out_dtype, ldo;
  const float* bias;
  const float* residual;  // fp32 [M][ld_res] added in the epilogue (may alias `out`), or null
  int ld_res;
  int epi;                // 0 plain; 1 SiLU(gate) * up over interleaved gate / up column tiles -> N/2 output columns
  // float weight types (nf4 / fp4 tables): the B operand pre-dequantised by deq_frag_kernel into the MFMA fragment
  // image [column tile][K tile][4 fragments][planes][64 lanes] x 16 B (planes: hi, and lo for the fp32-class form)
  const u32x4* bfrag;
  // split-K (few-row calls that would leave most CUs without a workgroup): blockIdx.y = K slice of kper tiles; every
  // slice leaves its scaled fp32 partial sums in part[z][M][ldp], splitk_reduce_kernel finishes (0 = no split)
  int kper;
  float* part;
  int ldp;
};

#ifndef WOQ_GEMM_HANDSCHED  // 1: the hand-scheduled K loop (woq_gemm_f16p.h) for NP = 1; 0: hipcc's schedule (A/B runs)
#define WOQ_GEMM_HANDSCHED 1
#endif
constexpr int FBM = 128;
constexpr int FTILE_BYTES = 128 * 128 * 2;

__device__ __forceinline__ int fperm(int e) { return (e < 4) ? 2 * e : 2 * (e - 4) + 1; }
__host__ __device__ inline int ht_swz(int row);  // chunk swizzle of the half-tile layout (woq_gemm_f16p.h)

// ---------------------------------------------------------------------------------------------------------------
// pack pass. Blocks [0, Mpad): one workgroup per activation row (row max / sum of squares, then convert and store).
// Blocks [Mpad, ...): one thread per weight column -> 2^E[col] >= max |scale| of the column.
// ---------------------------------------------------------------------------------------------------------------
struct PackF16Args {
  int planes;  // 1: fp16 plane; 2: hi + lo planes (x 2^-e = hi + lo to ~22 bits) for the three-product kernel
  int half_tiles;  // 1: [M/128][K/64] half-tiles of 16 KiB, row = 8 chunks, chunk c of row r in slot c ^ ht_swz(r)
                   // (the ring kernel of woq_gemm_f16p.h); 0: [M/128][K/128] tiles of 32 KiB, slot c ^ (r & 15)
  const void* x;
  int x_dtype, lda, M, Mpad, K, Kpad;
  const int32_t* shuffle;  // GPTQ act-order gather (converted g_idx) or null
  const float* norm_w;     // RMSNorm weight or null
  float eps;
  _Float16* ap;
  float* rs;
  // column-scale part
  const void* scales;
  int scale_type, scale_mode, n_groups, tiles_k, Npad, row_blocks;
  float* cs;
};

// MODE 0: fp32 rows, 16-B aligned; 1: bf16 / fp16 rows, 16-B aligned; 2: anything (shuffle, ragged alignment).
// Modes 0 / 1 keep the row in registers (one HBM read) when it fits PACK_MAXI chunks per thread, every load issued
// before the first use; the chunk index is clamped instead of branched on so the loads stay one batch.
constexpr int PACK_MAXI = 8;

template <int MODE>
__device__ __forceinline__ void pack_load8(const PackF16Args& a, size_t base, int k0, float (&v)[8]) {
  if constexpr (MODE == 0) {
    const float4_t lo4 = *(const float4_t*)((const float*)a.x + base + k0);
    const float4_t hi4 = *(const float4_t*)((const float*)a.x + base + k0 + 4);
    v[0] = lo4.x, v[1] = lo4.y, v[2] = lo4.z, v[3] = lo4.w, v[4] = hi4.x, v[5] = hi4.y, v[6] = hi4.z, v[7] = hi4.w;
  } else if constexpr (MODE == 1) {
    const u32x4 raw = *(const u32x4*)((const uint16_t*)a.x + base + k0);
    const bool bf = a.x_dtype == WOQ_BF16;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint16_t bits = (uint16_t)(raw[j >> 1] >> (16 * (j & 1)));
      const float fb = bf16_bits_to_f32(bits), fh = f16_bits_to_f32(bits);
      v[j] = bf ? fb : fh;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      v[j] = k0 + j < a.K ? load_f32(a.x, base + (a.shuffle ? a.shuffle[k0 + j] : k0 + j), a.x_dtype) : 0.f;
  }
}

template <int MODE, int NC>  // NC > 0: row cached in registers, NC chunks per thread; 0: two sweeps
__device__ __forceinline__ void pack_row(const PackF16Args& a, int r, float* red) {
  constexpr bool CACHED = NC > 0;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int mb = r >> 7, rl = r & 127;
  const int chunks = a.Kpad >> 3;
  const int full = a.K >> 3;  // chunks that lie wholly inside K (modes 0 / 1 load only those)
  const size_t tile_halves = (size_t)128 * 128 * a.planes;  // one (row block, K step): planes x 32 KiB
  _Float16* dst_row = a.ap + (size_t)mb * (a.Kpad >> 7) * tile_halves + (size_t)rl * 128;
  const bool ht = a.half_tiles != 0;  // planes == 1 there
  _Float16* ht_row = a.ap + (size_t)mb * (a.Kpad >> 6) * 8192 + (size_t)rl * 64;
  const size_t base = (size_t)r * a.lda;
  const bool norm = a.norm_w != nullptr;
  constexpr int NI = CACHED ? NC : 1;
  float v[NI][8];
  float amax = 0.f, ss = 0.f;
  auto fetch = [&](int c, bool valid, float (&d)[8]) {
    if constexpr (MODE == 2) {
      pack_load8<2>(a, base, c << 3, d);
      if (!valid) {
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = 0.f;
      }
    } else {
      const bool in = valid && c < full;
      pack_load8<MODE>(a, base, min(c, max(full - 1, 0)) << 3, d);
      if (!in) {  // K tail chunk (K % 8 != 0 never reaches modes 0 / 1 with a partial chunk) or K padding
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = 0.f;
      }
    }
    if (norm) {
      float g[8];
      if (MODE != 2 && c < full) {  // K % 8 == 0 here: whole chunk inside the weight vector
        const float4_t g0 = *(const float4_t*)(a.norm_w + c * 8), g1 = *(const float4_t*)(a.norm_w + c * 8 + 4);
        g[0] = g0.x, g[1] = g0.y, g[2] = g0.z, g[3] = g0.w, g[4] = g1.x, g[5] = g1.y, g[6] = g1.z, g[7] = g1.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {  // act-order rows: element k of the gathered row is x[shuffle[k]], its weight too
          const int k = min(c * 8 + j, a.K - 1);
          g[j] = a.norm_w[MODE == 2 && a.shuffle != nullptr ? a.shuffle[k] : k];
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ss = fmaf(d[j], d[j], ss);
        d[j] *= g[j];
      }
    }
  };
  if constexpr (CACHED) {
#pragma unroll
    for (int i = 0; i < NI; ++i) fetch(min(tid + 256 * i, chunks - 1), tid + 256 * i < chunks, v[i]);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[i][j]));
    }
  } else {
    for (int c = tid; c < chunks; c += 256) {
      fetch(c, true, v[0]);
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[0][j]));
    }
  }
  // block reduction (4 waves)
  amax = wave_max_dpp(amax);
  ss = wave_sum_dpp(ss);
  if (lane == 0) {
    red[wid] = amax;
    red[4 + wid] = ss;
  }
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  ss = (red[4] + red[5]) + (red[6] + red[7]);
  float nf = 1.f;
  if (norm) nf = rsqrtf(ss / (float)a.K + a.eps);
  amax *= nf;
  int e = 0;
  if (amax > 0.f && amax < INFINITY) e = max(-100, min(100, __builtin_amdgcn_frexp_expf(amax) - 14));
  const float p2 = ldexpf(1.f, -e) * nf;  // |x| nf 2^-e < 2^14
  if (tid == 0) a.rs[r] = ldexpf(1.f, e);
  auto emit = [&](int c, const float (&d)[8]) {
    h8 hh, ll;
#pragma unroll
    for (int e8 = 0; e8 < 8; ++e8) {
      const float xs = d[fperm(e8)] * p2;
      hh[e8] = (_Float16)xs;
      ll[e8] = (_Float16)(xs - (float)hh[e8]);
    }
    _Float16* dp = ht ? ht_row + (size_t)(c >> 3) * 8192 + (size_t)(((c & 7) ^ ht_swz(rl)) << 3)
                      : dst_row + (size_t)(c >> 4) * tile_halves + (size_t)(((c & 15) ^ (rl & 15)) << 3);
    *(h8*)dp = hh;
    if (a.planes == 2) *(h8*)(dp + 128 * 128) = ll;
  };
  if constexpr (CACHED) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (tid + 256 * i < chunks) emit(tid + 256 * i, v[i]);
  } else {
    for (int c = tid; c < chunks; c += 256) {  // second sweep: the row is L2-resident now
      fetch(c, true, v[0]);
      emit(c, v[0]);
    }
  }
}

__global__ __launch_bounds__(256) void pack_f16_kernel(PackF16Args a) {
  __shared__ float red[8];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= a.row_blocks) {  // ---- column scales: 32 columns per block, 8 threads per column ----
    // (each thread takes every 8th scale group of its column, the eight partial maxima meet in LDS: one thread per
    // column walking all groups was a 22-us chain of dependent misses when no activation rows ride along — the raw-A
    // calls — and 86 groups at K = 11008)
    __shared__ float cred[256];
    const int col = ((int)blockIdx.x - a.row_blocks) * 32 + (tid & 31), slice = tid >> 5;
    const int n = min(col, a.Npad - 1);
    const int tn = n >> 4, i16 = n & 15;
    float mx = 0.f;
    if (a.scale_mode == 0) {
      for (int g = slice; g < a.n_groups; g += 8)
        mx = fmaxf(mx, fabsf(load_f32(a.scales, ((size_t)tn * a.n_groups + g) * 16 + i16, a.scale_type)));
    } else {
      for (int kt = slice; kt < a.tiles_k; kt += 8)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          mx = fmaxf(mx, fabsf(load_f32(a.scales, ((((size_t)tn * a.tiles_k + kt) * 16 + i16) << 2) + u, a.scale_type)));
    }
    cred[tid] = mx;
    __syncthreads();
    if (slice == 0 && col < a.Npad) {
#pragma unroll
      for (int u = 1; u < 8; ++u) mx = fmaxf(mx, cred[u * 32 + tid]);
      int e = 0;
      if (mx > 0.f && mx < INFINITY) e = max(-120, min(120, __builtin_amdgcn_frexp_expf(mx)));  // mx < 2^e
      a.cs[col] = ldexpf(1.f, e);
    }
    return;
  }
  const int r = (int)blockIdx.x;
  if (r >= a.M) {  // padding rows of the last row block: zeros
    const int mb = r >> 7, rl = r & 127;
    const size_t tile_halves = (size_t)128 * 128 * a.planes;
    _Float16* dst_row = a.ap + (size_t)mb * (a.Kpad >> 7) * tile_halves + (size_t)rl * 128;
    _Float16* ht_row = a.ap + (size_t)mb * (a.Kpad >> 6) * 8192 + (size_t)rl * 64;
    for (int c = tid; c < (a.Kpad >> 3); c += 256) {
      _Float16* dp = a.half_tiles ? ht_row + (size_t)(c >> 3) * 8192 + (size_t)(((c & 7) ^ ht_swz(rl)) << 3)
                                  : dst_row + (size_t)(c >> 4) * tile_halves + (size_t)(((c & 15) ^ (rl & 15)) << 3);
      *(u32x4*)dp = (u32x4){0, 0, 0, 0};
      if (a.planes == 2) *(u32x4*)(dp + 128 * 128) = (u32x4){0, 0, 0, 0};
    }
    if (tid == 0) a.rs[r] = 0.f;
    return;
  }
  const int esz = a.x_dtype == WOQ_F32 ? 4 : 2;
  const bool vec_ok = a.shuffle == nullptr && (((uintptr_t)a.x) & 15) == 0 && (((size_t)a.lda * esz) & 15) == 0 &&
                      (a.K & 7) == 0;
  const int per_thread = ((a.Kpad >> 3) + 255) >> 8;  // chunks per thread
  if (!vec_ok) {
    pack_row<2, 0>(a, r, red);
  } else if (a.x_dtype == WOQ_F32) {
    if (per_thread <= 2)
      pack_row<0, 2>(a, r, red);
    else if (per_thread <= 4)
      pack_row<0, 4>(a, r, red);
    else if (per_thread <= PACK_MAXI)
      pack_row<0, PACK_MAXI>(a, r, red);
    else
      pack_row<0, 0>(a, r, red);
  } else {
    if (per_thread <= 2)
      pack_row<1, 2>(a, r, red);
    else if (per_thread <= 4)
      pack_row<1, 4>(a, r, red);
    else if (per_thread <= PACK_MAXI)
      pack_row<1, PACK_MAXI>(a, r, red);
    else
      pack_row<1, 0>(a, r, red);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// GEMM
// ---------------------------------------------------------------------------------------------------------------
template <int SMODE, bool S32, int CT>
struct BRegs {  // one K step of one wave's weight operand, as loaded
  u32x4 wv[CT];
  typename std::conditional<S32, float, uint32_t>::type sc[CT][SMODE == 0 ? 1 : 2];
  uint32_t zp[CT][SMODE == 0 ? 1 : 2];
};

// 8 signed nibbles of one blob word -> 8 fp16 (q - zp) * r, order {0,2,4,6,1,3,5,7}
__device__ __forceinline__ h8 dq8s(uint32_t w, h2 nlo, h2 nhi, h2 r) {
  // ((w ^ 0x88888888) & mask) | magic == (w & mask) ^ (magic | (mask & 0x88888888)): the nibble bits and the magic
  // bits do not overlap, so the sign flip, the mask and the magic are ONE three-input bit op per nibble pair
  const uint32_t y = w >> 8;
  const h2 f0 = (__builtin_bit_cast(h2, (w & 0x000f000fu) ^ 0x64086408u) + nlo) * r;
  const h2 f1 = (__builtin_bit_cast(h2, (w & 0x00f000f0u) ^ 0x54805480u) + nhi) * r;
  const h2 f2 = (__builtin_bit_cast(h2, (y & 0x000f000fu) ^ 0x64086408u) + nlo) * r;
  const h2 f3 = (__builtin_bit_cast(h2, (y & 0x00f000f0u) ^ 0x54805480u) + nhi) * r;
  return __builtin_bit_cast(h8, (u32x4){__builtin_bit_cast(uint32_t, f0), __builtin_bit_cast(uint32_t, f1),
                                        __builtin_bit_cast(uint32_t, f2), __builtin_bit_cast(uint32_t, f3)});
}

// fp32-class operand: (q - zp) * r with r = r_hi + r_lo (two fp16) -> hi = fl(d r_hi), lo = fl(d r_lo + (d r_hi - hi)).
// d r_hi has at most 16 significant bits, so the residual d r_hi - hi is exact in one v_pk_fma_f16.
__device__ __forceinline__ void dq8s_hl(uint32_t w, h2 nlo, h2 nhi, h2 rh, h2 rl, h8& hi, h8& lo) {
  const uint32_t y = w >> 8;
  h2 d[4];
  d[0] = __builtin_bit_cast(h2, (w & 0x000f000fu) ^ 0x64086408u) + nlo;
  d[1] = __builtin_bit_cast(h2, (w & 0x00f000f0u) ^ 0x54805480u) + nhi;
  d[2] = __builtin_bit_cast(h2, (y & 0x000f000fu) ^ 0x64086408u) + nlo;
  d[3] = __builtin_bit_cast(h2, (y & 0x00f000f0u) ^ 0x54805480u) + nhi;
  u32x4 uh, ul;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const h2 ph = d[i] * rh;
    const h2 res = __builtin_elementwise_fma(d[i], rh, -ph);
    const h2 pl = __builtin_elementwise_fma(d[i], rl, res);
    uh[i] = __builtin_bit_cast(uint32_t, ph);
    ul[i] = __builtin_bit_cast(uint32_t, pl);
  }
  hi = __builtin_bit_cast(h8, uh);
  lo = __builtin_bit_cast(h8, ul);
}

// epilogue shared by the GEMM kernels. D: lane (column i16, rows 4*kq + j) of every 16 x 16 fragment
template <int CT>
__device__ __forceinline__ void gemm_epilogue(const GemmF16Args& a, float4_t (&acc)[8][CT], int row0, int ct0, int i16,
                                              int kq) {
  const bool odd = (i16 & 1) != 0;
  const bool silu = a.epi == 1;
  const int n_out = silu ? (a.N >> 1) : a.N;
  const bool pair_ok = (a.ldo & 1) == 0 && (n_out & 1) == 0 && (a.ld_res & 1) == 0 &&
                       (((uintptr_t)a.out) & (a.out_dtype == WOQ_F32 ? 7 : 3)) == 0 && (((uintptr_t)a.residual) & 7) == 0;
  auto store_all = [&](auto put1, auto put2) {
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      if (silu && (c & 1)) continue;  // the up tile was consumed together with its gate tile
      const int n_in = (ct0 + c) * 16 + i16;                       // weight column
      const int n = silu ? ((ct0 + c) >> 1) * 16 + i16 : n_in;     // output column
      const bool live = ct0 + c < a.tiles_n && n_in < a.N && n < n_out;
      const float csv = live ? a.cs[n_in] : 0.f;
      const float csu = (silu && live) ? a.cs[n_in + 16] : 0.f;
      const float bsv = (live && a.bias) ? a.bias[n_in] : 0.f;
      const float bsu = (silu && live && a.bias) ? a.bias[n_in + 16] : 0.f;
#pragma unroll
      for (int rt = 0; rt < 8; ++rt) {
        const int mrow = row0 + rt * 16 + kq * 4;
        const float4_t rsv = a.rs ? *(const float4_t*)(a.rs + mrow) : (float4_t){1.f, 1.f, 1.f, 1.f};  // padded to the row block
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = fmaf(acc[rt][c][j], rsv[j] * csv, bsv);
          if (silu) {
            const float u = fmaf(acc[rt][(c | 1) < CT ? (c | 1) : c][j], rsv[j] * csu, bsu);
            v[j] = v[j] / (1.f + __expf(-v[j])) * u;
          }
        }
        if (pair_ok) {
          // even lanes keep rows 0,1 of (col, col+1); odd lanes rows 2,3 of (col-1, col)
          const float t0 = WOQ_DPP_F32(odd ? v[0] : v[2], 0xB1), t1 = WOQ_DPP_F32(odd ? v[1] : v[3], 0xB1);
          if (live) {
            const int m0 = mrow + (odd ? 2 : 0);
            const int nn = n & ~1;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              if (m0 + u >= a.M) continue;
              float x = odd ? (u ? t1 : t0) : v[u], y = odd ? v[2 + u] : (u ? t1 : t0);
              if (a.residual) {
                const float2 rr = *(const float2*)(a.residual + (size_t)(m0 + u) * a.ld_res + nn);
                x += rr.x;
                y += rr.y;
              }
              put2((size_t)(m0 + u) * a.ldo + nn, x, y);
            }
          }
        } else if (live) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (mrow + j < a.M)
              put1((size_t)(mrow + j) * a.ldo + n,
                   v[j] + (a.residual ? a.residual[(size_t)(mrow + j) * a.ld_res + n] : 0.f));
        }
      }
    }
  };
  // fp16 stores saturate instead of overflowing to inf (the engine keeps q / k / v / MLP activations in fp16)
  auto h16 = [](float v) { return f32_to_f16_bits(fminf(fmaxf(v, -65504.f), 65504.f)); };
  if (a.out_dtype == WOQ_F32)
    store_all([&](size_t i, float v) { ((float*)a.out)[i] = v; },
              [&](size_t i, float x, float y) { *(float2*)((float*)a.out + i) = make_float2(x, y); });
  else if (a.out_dtype == WOQ_BF16)
    store_all([&](size_t i, float v) { ((uint16_t*)a.out)[i] = f32_to_bf16_bits(v); },
              [&](size_t i, float x, float y) {
                *(uint32_t*)((uint16_t*)a.out + i) = (uint32_t)f32_to_bf16_bits(x) | ((uint32_t)f32_to_bf16_bits(y) << 16);
              });
  else
    store_all([&](size_t i, float v) { ((uint16_t*)a.out)[i] = h16(v); },
              [&](size_t i, float x, float y) {
                *(uint32_t*)((uint16_t*)a.out + i) = (uint32_t)h16(x) | ((uint32_t)h16(y) << 16);
              });
}

// Everything the epilogue needs is read from the kernel-argument segment at the TOP of a GEMM kernel and kept in SGPRs
// across the K loop (WOQ_PIN_EPILOGUE_ARGS there, WOQ_UNPIN_EPILOGUE_ARGS in front of gemm_epilogue). Left to itself
// hipcc re-loads some of it behind the loop (s_load_dword ..., s[0:1], 0x50 for M) and recycles the segment pointer's
// s0 with a v_readfirstlane three instructions later; in the first launches of a process (cold scalar cache) one wave
// in a few hundred workgroups of the hand-scheduled kernel then saw a garbage M and stored nothing: 10 of 80 process
// starts left output elements unwritten, 0 of 200 with the arguments pinned (tools/gemm_probe.hip, PROBE_CLEAR).
#define WOQ_PIN_EPILOGUE_ARGS(a)                                                                                       \
  int p_m = a.M, p_n = a.N, p_ldo = a.ldo, p_odt = a.out_dtype, p_ldr = a.ld_res, p_epi = a.epi, p_tn = a.tiles_n;     \
  void* p_out = a.out;                                                                                                 \
  const float *p_bias = a.bias, *p_res = a.residual, *p_rs = a.rs, *p_cs = a.cs;                                       \
  asm volatile("" : "+s"(p_m), "+s"(p_n), "+s"(p_ldo), "+s"(p_odt), "+s"(p_ldr), "+s"(p_epi), "+s"(p_tn));             \
  asm volatile("" : "+s"(p_out), "+s"(p_bias), "+s"(p_res), "+s"(p_rs), "+s"(p_cs));
#define WOQ_UNPIN_EPILOGUE_ARGS(a)                                                                                     \
  asm volatile("" : "+s"(p_m), "+s"(p_n), "+s"(p_ldo), "+s"(p_odt), "+s"(p_ldr), "+s"(p_epi), "+s"(p_tn));             \
  asm volatile("" : "+s"(p_out), "+s"(p_bias), "+s"(p_res), "+s"(p_rs), "+s"(p_cs));                                   \
  a.M = p_m, a.N = p_n, a.ldo = p_ldo, a.out_dtype = p_odt, a.ld_res = p_ldr, a.epi = p_epi, a.tiles_n = p_tn;         \
  a.out = p_out, a.bias = p_bias, a.residual = p_res, a.rs = p_rs, a.cs = p_cs;

// NP = 1: one fp16 product per fragment pair (compute_dtype bf16 / fp16 / int8). NP = 3: fp32-class — activations
// and scaled weights both carried as hi + lo fp16 pairs (~22 bits each), A_hi B_hi + A_hi B_lo + A_lo B_hi
// accumulated in the same fp32 fragments (the dropped A_lo B_lo term is 2^-22 of the product). Same data flow, same
// per-group scale folding, so group-32 blobs cost what group-128 blobs cost. One workgroup per CU (the A stage is
// 64 KiB: two planes).
// CT = 16-column tiles per wave: the workgroup covers 128 rows x 64 CT columns.
template <int SMODE, bool ASYM, bool S32, int NP = 1, int CT = 2>
__global__ __launch_bounds__(256, NP == 1 ? 2 : 1) void gemm_f16s_kernel(GemmF16Args a) {
  constexpr int PLANES = NP == 1 ? 1 : 2;
  constexpr int FBN = 64 * CT;
  constexpr int STAGE = FTILE_BYTES * PLANES;
  extern __shared__ __attribute__((aligned(1024))) unsigned char fsm[];  // 2 x 32 KiB A tiles
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kq = lane >> 4;

  // XCD-aware placement: workgroup ids go round-robin over the 8 XCDs; each XCD walks its own sequence of
  // 8 x 8 super-tiles, 64 consecutive local ids per super-tile
  const int bid = (int)blockIdx.x;
  const int sup = ((bid >> 3) >> 6) * 8 + (bid & 7), within = (bid >> 3) & 63;
  if (sup >= a.n_sup) return;
  const int mb = (sup / a.sup_n) * 8 + (within >> 3), nb = (sup % a.sup_n) * 8 + (within & 7);
  if (mb >= a.nb_m || nb >= a.nb_n) return;
  const int row0 = mb * FBM;
  const int ct0 = nb * (FBN / 16) + wid * CT;  // this wave's first column tile

  float4_t acc[8][CT];
#pragma unroll
  for (int rt = 0; rt < 8; ++rt)
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[rt][c] = (float4_t){0.f, 0.f, 0.f, 0.f};

  WOQ_PIN_EPILOGUE_ARGS(a)

  // ---- operand movers ----
  const _Float16* a_tiles = a.ap + (size_t)mb * a.tiles_k * (STAGE / 2);
  // Every load of the K loop is issued from inline asm and waited for by hand (wait_loads below). With the builtins
  // hipcc keeps its own count of what is in flight and, not knowing that the s_waitcnt at the top of a K step already
  // retired the CURRENT step's operands, puts an s_waitcnt vmcnt(0) in front of their first use — after the NEXT
  // step's loads have been issued, so every K step sat out a full load latency before its first MFMA.
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)fsm;
  auto issue_a = [&](int kt, int buf) {  // 8 LDS-DMA pieces of 1 KiB per wave
    const _Float16* src = a_tiles + (size_t)kt * (STAGE / 2) + (size_t)wid * (4096 * PLANES) + lane * 8;
    const uint32_t dst = lds0 + buf * STAGE + wid * (8192 * PLANES);
    // the instruction offset advances the global AND the LDS address: four 1-KiB pieces per address / M0 setup
#pragma unroll
    for (int j = 0; j < 8 * PLANES; j += 4) {
      uint32_t keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
          "global_load_lds_dwordx4 %1, off\n\t"
          "global_load_lds_dwordx4 %1, off offset:1024\n\t"
          "global_load_lds_dwordx4 %1, off offset:2048\n\t"
          "global_load_lds_dwordx4 %1, off offset:3072\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(src + j * 512), "s"(dst + j * 1024)
          : "memory");
    }
  };
  auto ld128 = [](u32x4& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); };
  auto ld32 = [](float& d, const void* p) { asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p) : "memory"); };
  auto ld16 = [](uint32_t& d, const void* p) { asm volatile("global_load_ushort %0, %1, off" : "=v"(d) : "v"(p) : "memory"); };
  auto ld8 = [](uint32_t& d, const void* p) { asm volatile("global_load_ubyte %0, %1, off" : "=v"(d) : "v"(p) : "memory"); };
  int tnc[CT];
  float icol[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    tnc[c] = min(ct0 + c, a.tiles_n - 1);
    icol[c] = 1.f / a.cs[tnc[c] * 16 + i16];  // exact: a power of two
  }
  const int lane_s = kq >> 1;  // group-32: which 32-k group of a 64-k half this lane quarter belongs to
  auto load_b = [&](int kt, BRegs<SMODE, S32, CT>& b) {
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      ld128(b.wv[c], a.q + ((size_t)tnc[c] * a.tiles_k + kt) * 64 + lane);
      if constexpr (SMODE == 0) {
        int g = (kt * 128) / a.group;
        g = g >= a.n_groups ? a.n_groups - 1 : g;
        const size_t si = ((size_t)tnc[c] * a.n_groups + g) * 16 + i16;
        if constexpr (S32)
          ld32(b.sc[c][0], (const float*)a.scales + si);
        else
          ld16(b.sc[c][0], (const uint16_t*)a.scales + si);
        if constexpr (ASYM) ld8(b.zp[c][0], a.zp + si);
      } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const size_t si = ((((size_t)tnc[c] * a.tiles_k + kt) * 16 + i16) << 2) + 2 * h + lane_s;
          if constexpr (S32)
            ld32(b.sc[c][h], (const float*)a.scales + si);
          else
            ld16(b.sc[c][h], (const uint16_t*)a.scales + si);
          if constexpr (ASYM) ld8(b.zp[c][h], a.zp + si);
        }
      }
    }
  };
  // everything in flight has landed (the loads of ONE K step are all that ever is); the "+v" ties make the register
  // operands' first use depend on the wait
  auto wait_loads = [&](BRegs<SMODE, S32, CT>& b) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      asm volatile("" : "+v"(b.wv[c]));
#pragma unroll
      for (int h = 0; h < (SMODE == 0 ? 1 : 2); ++h) {
        asm volatile("" : "+v"(b.sc[c][h]));
        if constexpr (ASYM) asm volatile("" : "+v"(b.zp[c][h]));
      }
    }
  };
  const bool sc_bf = a.scale_type == WOQ_BF16;
  // A fragment of (64-k half h, part p) for row tile rt: chunk h*8 + kq*2 + p of row rt*16 + i16, slot = chunk ^ i16
  int a_off[4];
#pragma unroll
  for (int hp = 0; hp < 4; ++hp) a_off[hp] = i16 * 256 + ((((hp >> 1) * 8 + kq * 2 + (hp & 1)) ^ i16) << 4);

  auto compute = [&](int buf, const BRegs<SMODE, S32, CT>& b) {
    const unsigned char* at = fsm + buf * STAGE;
    constexpr int NS = SMODE == 0 ? 1 : 2;
    h2 r2[CT][NS], r2l[CT][NS], nlo[CT][NS], nhi[CT][NS];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        float sv;
        if constexpr (S32) {
          sv = b.sc[c][s];
        } else {
          const float fb = bf16_bits_to_f32((uint16_t)b.sc[c][s]), fh = f16_bits_to_f32((uint16_t)b.sc[c][s]);
          sv = sc_bf ? fb : fh;
        }
        const float rf = sv * icol[c];
        const _Float16 rr = (_Float16)rf;
        r2[c][s] = (h2){rr, rr};
        if constexpr (NP == 3) {
          const _Float16 rl_ = (_Float16)(rf - (float)rr);
          r2l[c][s] = (h2){rl_, rl_};
        }
        const float uz = ASYM ? (float)(b.zp[c][s] & 0xff) : 8.f;
        const _Float16 l = (_Float16)(-(1024.f + uz)), hgh = (_Float16)(-(64.f + uz));
        nlo[c][s] = (h2){l, l};
        nhi[c][s] = (h2){hgh, hgh};
      }
#pragma unroll
    for (int hp = 0; hp < 4; ++hp) {
      const int s = SMODE == 0 ? 0 : (hp >> 1);
      h8 bfr[CT], bfl[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        if constexpr (NP == 3)
          dq8s_hl(b.wv[c][hp], nlo[c][s], nhi[c][s], r2[c][s], r2l[c][s], bfr[c], bfl[c]);
        else
          bfr[c] = dq8s(b.wv[c][hp], nlo[c][s], nhi[c][s], r2[c][s]);
      }
#pragma unroll
      for (int rt = 0; rt < 8; ++rt) {
        const h8 af = *(const h8*)(at + rt * 4096 + a_off[hp]);
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bfr[c], acc[rt][c], 0, 0, 0);
        if constexpr (NP == 3) {
          const h8 al = *(const h8*)(at + FTILE_BYTES + rt * 4096 + a_off[hp]);
#pragma unroll
          for (int c = 0; c < CT; ++c) {
            acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bfl[c], acc[rt][c], 0, 0, 0);
            acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bfr[c], acc[rt][c], 0, 0, 0);
          }
        }
      }
    }
  };

  // ---- K loop, two steps per trip (ping-pong register sets and LDS buffers; no register copies) ----
  BRegs<SMODE, S32, CT> b0, b1;
  // this workgroup's K tiles: all of them, or slice blockIdx.y of a split call
  const int kt_lo = a.kper > 0 ? (int)blockIdx.y * a.kper : 0;
  const int kt_hi = a.kper > 0 ? min(a.tiles_k, kt_lo + a.kper) : a.tiles_k;
  issue_a(kt_lo, 0);
  load_b(kt_lo, b0);
  const int last = kt_hi - 1;
  for (int kt = kt_lo; kt < kt_hi; kt += 2) {
    wait_loads(b0);   // this wave's share of tile kt and its weight registers have landed
    __syncthreads();  // everyone's has; everyone is done reading the other buffer
    issue_a(min(kt + 1, last), 1);
    load_b(min(kt + 1, last), b1);
    __builtin_amdgcn_sched_barrier(0);
    compute(0, b0);
    if (kt + 1 >= kt_hi) break;
    wait_loads(b1);
    __syncthreads();
    issue_a(min(kt + 2, last), 0);
    load_b(min(kt + 2, last), b0);
    __builtin_amdgcn_sched_barrier(0);
    compute(1, b1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may still be writing LDS when the workgroup retires

  WOQ_UNPIN_EPILOGUE_ARGS(a)
  if (a.kper > 0) {  // a K slice: scaled fp32 partial sums only; bias, SiLU * mul, residual and the cast come after the sum
    a.out = a.part + (size_t)blockIdx.y * a.M * a.ldp;
    a.out_dtype = WOQ_F32;
    a.ldo = a.ldp;
    a.bias = nullptr;
    a.residual = nullptr;
    a.epi = 0;
  }
  gemm_epilogue<CT>(a, acc, row0, ct0, i16, kq);
}

// out[m][n] = sum over the K slices (+ bias, SiLU(gate) * up over interleaved column tiles, + residual), cast
struct SplitKArgs {
  const float* part;
  int nz, M, N, ldp, epi;
  const float* bias;
  const float* residual;
  int ld_res;
  void* out;
  int out_dtype, ldo;
};
__global__ __launch_bounds__(256) void splitk_reduce_kernel(SplitKArgs a) {
  const int n_out = a.epi == 1 ? (a.N >> 1) : a.N;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)a.M * n_out) return;
  const int m = (int)(idx / n_out), n = (int)(idx % n_out);
  const int n_in = a.epi == 1 ? ((n >> 4) * 2) * 16 + (n & 15) : n;
  float v = 0.f, u = 0.f;
  for (int z = 0; z < a.nz; ++z) {  // fixed order: the result does not depend on which slice finished first
    const float* row = a.part + ((size_t)z * a.M + m) * a.ldp;
    v += row[n_in];
    if (a.epi == 1) u += row[n_in + 16];
  }
  if (a.bias) {
    v += a.bias[n_in];
    if (a.epi == 1) u += a.bias[n_in + 16];
  }
  if (a.epi == 1) v = v / (1.f + __expf(-v)) * u;
  if (a.residual) v += a.residual[(size_t)m * a.ld_res + n];
  if (a.out_dtype == WOQ_F16) v = fminf(fmaxf(v, -65504.f), 65504.f);
  store_f32(a.out, (size_t)m * a.ldo + n, a.out_dtype, v);
}

// ---------------------------------------------------------------------------------------------------------------
// float weight types at M > 8 (round 3). nf4 / fp4 weights are w = table[code] * scale: no integer identity turns the
// code into an fp16 in two VALU, and a 16-entry lookup per weight inside the K loop would cost more than the MFMAs it
// feeds. So the dequantisation is a PRE-PASS (deq_frag_kernel: blob -> fp16 MFMA fragments, scaled by the same
// power-of-two column factors as the int4 path, hi + lo planes for compute_dtype fp32) and the GEMM reads finished
// fragments: 16 B per lane per fragment, no VALU on the B side at all. Costs one pass over the weight (25 MB in,
// 100-200 MB out for a 4096 x 12288 projection, ~50 us) per call; before it these types ran the generic fp32 GEMV kernel
// at every M: 64 ms for 2048 rows of that projection (3.2 TFLOP/s, tools/wtypes_bench.py).
// ---------------------------------------------------------------------------------------------------------------
struct DeqFragArgs {
  const u32x4* q;
  const u32x4* q_lo;  // fp8 weight types: the low-nibble plane (code = hi nibble << 4 | lo nibble ^ 8), else null
  const void* scales;
  int scale_type, scale_mode, n_groups, group, tiles_k, tiles_n, planes;
  uint32_t weight_type;
  const float* cs;  // [Npad] 2^E per column (pack pass)
  u32x4* out;
};

__global__ __launch_bounds__(256) void deq_frag_kernel(DeqFragArgs a) {
  __shared__ float lut_s[256];  // table types: 16 values; fp8: all 256 codes
  const int tid = threadIdx.x, lane = tid & 63;
  const bool fp8 = a.q_lo != nullptr;
  if (fp8)
    lut_s[tid] = fp8_code_value(a.weight_type, tid);
  else if (tid < 16)
    lut_s[tid] = lut_value(a.weight_type, tid);
  __syncthreads();
  const size_t tile = (size_t)blockIdx.x * 4 + (tid >> 6);  // (column tile, K tile), K fastest
  if (tile >= (size_t)a.tiles_n * a.tiles_k) return;
  const int tn = (int)(tile / a.tiles_k), kt = (int)(tile % a.tiles_k);
  const int i16 = lane & 15, kq = lane >> 4;
  const u32x4 wv = a.q[tile * 64 + lane];
  u32x4 wl = {0u, 0u, 0u, 0u};
  if (fp8) wl = a.q_lo[tile * 64 + lane];
  const float icol = 1.f / a.cs[tn * 16 + i16];  // exact: a power of two
#pragma unroll
  for (int hp = 0; hp < 4; ++hp) {
    size_t si;
    if (a.scale_mode == 0) {
      int g = (kt * 128) / a.group;
      g = g >= a.n_groups ? a.n_groups - 1 : g;
      si = ((size_t)tn * a.n_groups + g) * 16 + i16;
    } else {
      si = ((((size_t)tn * a.tiles_k + kt) * 16 + i16) << 2) + 2 * (hp >> 1) + (kq >> 1);
    }
    const float r = load_f32(a.scales, si, a.scale_type) * icol;
    // element e of the fragment = nibble {0,4,1,5,2,6,3,7}[e] of the word: the order dq8s leaves its eight values in
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      uint32_t c0 = (wv[hp] >> (4 * pr)) & 0xfu, c1 = (wv[hp] >> (4 * pr + 16)) & 0xfu;
      if (fp8) {
        c0 = (c0 << 4) | (((wl[hp] >> (4 * pr)) & 0xfu) ^ 8u);
        c1 = (c1 << 4) | (((wl[hp] >> (4 * pr + 16)) & 0xfu) ^ 8u);
      }
      const float w0 = lut_s[c0] * r, w1 = lut_s[c1] * r;
      const _Float16 h0 = (_Float16)w0, h1 = (_Float16)w1;
      hi[pr] = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
      const _Float16 l0 = (_Float16)(w0 - (float)h0), l1 = (_Float16)(w1 - (float)h1);
      lo[pr] = (uint32_t)__builtin_bit_cast(uint16_t, l0) | ((uint32_t)__builtin_bit_cast(uint16_t, l1) << 16);
    }
    u32x4* dst = a.out + ((tile * 4 + hp) * a.planes) * 64 + lane;
    dst[0] = (u32x4){hi[0], hi[1], hi[2], hi[3]};
    if (a.planes == 2) dst[64] = (u32x4){lo[0], lo[1], lo[2], lo[3]};
  }
}

template <int NP, int CT = 2>
__global__ __launch_bounds__(256, NP == 1 ? 2 : 1) void gemm_f16frag_kernel(GemmF16Args a) {
  constexpr int PLANES = NP == 1 ? 1 : 2;
  constexpr int FBN = 64 * CT;
  constexpr int STAGE = FTILE_BYTES * PLANES;
  extern __shared__ __attribute__((aligned(1024))) unsigned char fsm[];  // 2 x 32 KiB A tiles (x planes)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kq = lane >> 4;
  const int bid = (int)blockIdx.x;  // XCD-aware 8 x 8 super-tiles, as gemm_f16s_kernel
  const int sup = ((bid >> 3) >> 6) * 8 + (bid & 7), within = (bid >> 3) & 63;
  if (sup >= a.n_sup) return;
  const int mb = (sup / a.sup_n) * 8 + (within >> 3), nb = (sup % a.sup_n) * 8 + (within & 7);
  if (mb >= a.nb_m || nb >= a.nb_n) return;
  const int row0 = mb * FBM;
  const int ct0 = nb * (FBN / 16) + wid * CT;
  float4_t acc[8][CT];
#pragma unroll
  for (int rt = 0; rt < 8; ++rt)
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[rt][c] = (float4_t){0.f, 0.f, 0.f, 0.f};
  WOQ_PIN_EPILOGUE_ARGS(a)
  const _Float16* a_tiles = a.ap + (size_t)mb * a.tiles_k * (STAGE / 2);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)fsm;
  auto issue_a = [&](int kt, int buf) {  // 8 LDS-DMA pieces of 1 KiB per wave and plane
    const _Float16* src = a_tiles + (size_t)kt * (STAGE / 2) + (size_t)wid * (4096 * PLANES) + lane * 8;
    const uint32_t dst = lds0 + buf * STAGE + wid * (8192 * PLANES);
#pragma unroll
    for (int j = 0; j < 2 * PLANES; ++j) {
      uint32_t keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
          "global_load_lds_dwordx4 %1, off\n\t"
          "global_load_lds_dwordx4 %1, off offset:1024\n\t"
          "global_load_lds_dwordx4 %1, off offset:2048\n\t"
          "global_load_lds_dwordx4 %1, off offset:3072\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(src + j * 2048), "s"(dst + j * 4096)
          : "memory");
    }
  };
  int tnc[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) tnc[c] = min(ct0 + c, a.tiles_n - 1);
  struct Frags {
    u32x4 f[CT][4][PLANES];
  };
  auto load_b = [&](int kt, Frags& b) {
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int hp = 0; hp < 4; ++hp)
#pragma unroll
        for (int p = 0; p < PLANES; ++p)
          asm volatile("global_load_dwordx4 %0, %1, off"
                       : "=v"(b.f[c][hp][p])
                       : "v"(a.bfrag + ((((size_t)tnc[c] * a.tiles_k + kt) * 4 + hp) * PLANES + p) * 64 + lane)
                       : "memory");
  };
  auto wait_loads = [&](Frags& b) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int hp = 0; hp < 4; ++hp)
#pragma unroll
        for (int p = 0; p < PLANES; ++p) asm volatile("" : "+v"(b.f[c][hp][p]));
  };
  int a_off[4];
#pragma unroll
  for (int hp = 0; hp < 4; ++hp) a_off[hp] = i16 * 256 + ((((hp >> 1) * 8 + kq * 2 + (hp & 1)) ^ i16) << 4);
  auto compute = [&](int buf, const Frags& b) {
    const unsigned char* at = fsm + buf * STAGE;
#pragma unroll
    for (int hp = 0; hp < 4; ++hp) {
#pragma unroll
      for (int rt = 0; rt < 8; ++rt) {
        const h8 af = *(const h8*)(at + rt * 4096 + a_off[hp]);
#pragma unroll
        for (int c = 0; c < CT; ++c)
          acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, __builtin_bit_cast(h8, b.f[c][hp][0]), acc[rt][c], 0, 0, 0);
        if constexpr (NP == 3) {
          const h8 al = *(const h8*)(at + FTILE_BYTES + rt * 4096 + a_off[hp]);
#pragma unroll
          for (int c = 0; c < CT; ++c) {
            acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, __builtin_bit_cast(h8, b.f[c][hp][PLANES - 1]), acc[rt][c], 0, 0, 0);
            acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, __builtin_bit_cast(h8, b.f[c][hp][0]), acc[rt][c], 0, 0, 0);
          }
        }
      }
    }
  };
  Frags b0, b1;
  issue_a(0, 0);
  load_b(0, b0);
  const int last = a.tiles_k - 1;
  for (int kt = 0; kt < a.tiles_k; kt += 2) {
    wait_loads(b0);
    __syncthreads();
    issue_a(min(kt + 1, last), 1);
    load_b(min(kt + 1, last), b1);
    __builtin_amdgcn_sched_barrier(0);
    compute(0, b0);
    if (kt + 1 >= a.tiles_k) break;
    wait_loads(b1);
    __syncthreads();
    issue_a(min(kt + 2, last), 0);
    load_b(min(kt + 2, last), b0);
    __builtin_amdgcn_sched_barrier(0);
    compute(1, b1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WOQ_UNPIN_EPILOGUE_ARGS(a)
  gemm_epilogue<CT>(a, acc, row0, ct0, i16, kq);
}

template <int NP>
static int launch_f16frag_t(GemmF16Args& a, hipStream_t st) {
  auto kern = gemm_f16frag_kernel<NP, 2>;
  static bool attr_set = false;
  static std::mutex attr_mu;
  {
    std::lock_guard<std::mutex> attr_lock(attr_mu);
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * FTILE_BYTES * 2);
      if (e != hipSuccess) return woq::fail(std::string("QBits: hipFuncSetAttribute: ") + hipGetErrorString(e));
      attr_set = true;
    }
  }
  const int n_sup8 = (a.n_sup + 7) / 8;
  hipLaunchKernelGGL(kern, dim3((unsigned)(n_sup8 * 8 * 64)), dim3(256), 2 * FTILE_BYTES * (NP == 1 ? 1 : 2), st, a);
  return 0;
}

#include "woq_gemm_f16p.h"
#include "woq_gemm_f16t.h"

template <int SMODE, bool ASYM, bool S32, int NP>
static int launch_f16_t(GemmF16Args& a, hipStream_t st) {
  auto kern = gemm_f16s_kernel<SMODE, ASYM, S32, NP, 2>;
  bool ring = false;
#if WOQ_GEMM_HANDSCHED
  if (NP == 1 && (a.tiles_k & 1) == 0 && a.kper == 0) {  // (K slices run the kernel above: its K loop takes a range)  // (its K loop runs two K steps per trip; odd tile counts keep the kernel above)
    // (the ring form needs <= 168 VGPRs for its third workgroup per CU; group-32 asymmetric blobs with fp32 scales do
    // not fit without spills and stay on the two-tile form)
    constexpr bool ring_fits = !(SMODE == 1 && ASYM && S32);
    ring = a.ring != 0 && ring_fits;
#define WOQ_PICK(RAW_, RING_)                                                                       \
  (S32 ? gemm_f16p_kernel<SMODE, ASYM, 2, RAW_, RING_>                                              \
       : (a.scale_type == WOQ_BF16 ? gemm_f16p_kernel<SMODE, ASYM, 1, RAW_, RING_>                  \
                                   : gemm_f16p_kernel<SMODE, ASYM, 0, RAW_, RING_>))
    if constexpr (ring_fits) {
      if (ring) kern = a.act_raw ? WOQ_PICK(true, true) : WOQ_PICK(false, true);
    }
    if (!ring)
      kern = a.act_raw ? WOQ_PICK(true, false) : WOQ_PICK(false, false);
#undef WOQ_PICK
  }
#endif
  int LDS = ring ? 3 * (FTILE_BYTES / 2) : 2 * FTILE_BYTES * (NP == 1 ? 1 : 2);
  // round 6: 256-row workgroup tiles (woq_gemm_f16t.h) — the ring layout's half-tile images (or the raw rows), twice the
  // rows per wave, half the weight unpack per MFMA. From 2048 rows (below that the 128-row tiles fill the chip better);
  // WOQ_GEMM_TALL=0: off (A/B runs).
  [[maybe_unused]] bool tall = false;
#if WOQ_GEMM_HANDSCHED
  if constexpr (NP == 1 && !(SMODE == 1 && ASYM && S32)) {
    static const bool tall_ok = !(getenv("WOQ_GEMM_TALL") && getenv("WOQ_GEMM_TALL")[0] == '0');
    static const int tall_rows = getenv("WOQ_GEMM_TALL_ROWS") ? atoi(getenv("WOQ_GEMM_TALL_ROWS")) : 2048;
    // raw-A calls (o / down of the prompt pass) keep the 128-row ring kernel: alone they gain 2-7 % on 256-row tiles, inside
    // the engine's pass they lose (0.4915 vs 0.4975 with them on, profiles/r06i_*); WOQ_GEMM_TALL_RAW=1 turns them on
    static const bool tall_raw = getenv("WOQ_GEMM_TALL_RAW") && getenv("WOQ_GEMM_TALL_RAW")[0] == '1';
    static const int tall_wgs = getenv("WOQ_GEMM_TALL_WGS") ? atoi(getenv("WOQ_GEMM_TALL_WGS")) : 1024;
    // enough 256-row workgroups for two full rounds of the chip's 512 slots (M = 2048 x N = 4096 is 256 of them: 132 us
    // against 82 us for the 128-row tiles, profiles/r06h_*)
    if (tall_ok && ring && a.M >= tall_rows && ((a.nb_m + 1) / 2) * a.nb_n >= tall_wgs && (tall_raw || !a.act_raw)) {
      tall = true;
#define WOQ_PICK_T(RAW_)                                                                   \
  (S32 ? gemm_f16t_kernel<SMODE, ASYM, 2, RAW_>                                            \
       : (a.scale_type == WOQ_BF16 ? gemm_f16t_kernel<SMODE, ASYM, 1, RAW_> : gemm_f16t_kernel<SMODE, ASYM, 0, RAW_>))
      kern = a.act_raw ? WOQ_PICK_T(true) : WOQ_PICK_T(false);
#undef WOQ_PICK_T
      a.nb_m128 = a.nb_m;
      a.nb_m = (a.nb_m + 1) / 2;
      a.n_sup = ((a.nb_m + 7) / 8) * a.sup_n;
      LDS = 2 * FTILE_BYTES;
    }
  }
#endif
  static const void* attr_set[12] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                     nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  static std::mutex attr_mu;  // (host threads launching concurrently)
  std::lock_guard<std::mutex> attr_lock(attr_mu);
  bool have = false;  // (attr_set: the kernels this instantiation can pick)
  int free_slot = 11;
  for (int i = 11; i >= 0; --i) {
    have = have || attr_set[i] == (const void*)kern;
    if (attr_set[i] == nullptr) free_slot = i;
  }
  if (!have) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * FTILE_BYTES * 2);
    if (e != hipSuccess) return woq::fail(std::string("QBits: hipFuncSetAttribute: ") + hipGetErrorString(e));
    attr_set[free_slot] = (const void*)kern;
  }
  const int n_sup8 = (a.n_sup + 7) / 8;
  const unsigned nz = a.kper > 0 ? (unsigned)((a.tiles_k + a.kper - 1) / a.kper) : 1u;
  hipLaunchKernelGGL(kern, dim3((unsigned)(n_sup8 * 8 * 64), nz), dim3(256), LDS, st, a);
  return 0;
}

// Workspace bytes for an [M, K] x [K, N] call (activation tiles + row scales + column scales).
// (+ SPLITK_WS for the partial sums of a split-K call: slices x rows x columns <= 512 / workgroups x 128 x 128 floats
// per workgroup of the unsplit grid, i.e. <= 34 MiB whatever the shape; a fixed 40 MiB keeps the caller's sizing simple)
constexpr size_t SPLITK_WS = (size_t)40 << 20;
size_t gemm_f16_workspace_bytes(int M, int Kpad, int Npad, int planes) {
  const size_t Mpad = ((size_t)M + FBM - 1) / FBM * FBM;
  const size_t base = Mpad * Kpad * sizeof(_Float16) * planes + Mpad * sizeof(float) + (size_t)Npad * sizeof(float);
  return ((base + 255) & ~(size_t)255) + SPLITK_WS;
}

// the same for one blob, fragment image of the float weight types included (4-8x the blob: what made table-type prompt
// passes allocate per call while the engine's workspace sat unused — ADVICE r04)
size_t gemm_f16_workspace_bytes_blob(int M, const woq_blob_header& h, int planes) {
  const size_t tiles = (size_t)(h.Npad / WOQ_TILE_N) * (h.Kpad / WOQ_TILE_K);
  return gemm_f16_workspace_bytes(M, h.Kpad, h.Npad, planes) + (is_table_type(h.weight_type) ? tiles * 4 * planes * 1024 : 0);
}

// out[M,N] = act[M,K] . W_deq (+ bias) with fp16 operands. `ws` = caller workspace of gemm_f16_workspace_bytes or
// null (stream-ordered allocation per call, like the reference's per-call amalloc,
// bestla_weightonly_dispatcher.cpp:108-118,179). norm_w/eps: RMSNorm fused into the pack pass (null = none);
// residual / epi: see GemmF16Args; fp32_class: the three-product hi + lo form (compute_dtype fp32).
int launch_gemm_f16(const void* act, int act_dtype, int lda, const void* blob, const woq_blob_header& h,
                    const float* bias, void* out, int out_dtype, int ldo, int M, const float* norm_w, float eps,
                    const float* residual, int ld_res, int epi, void* ws, int fp32_class, hipStream_t st,
                    const void* fp8_lo, uint32_t fp8_type, size_t ws_bytes) {
  const int planes = fp32_class ? 2 : 1;
  if (epi == 1 && (((h.Npad / WOQ_TILE_N) & 1) != 0 || (h.N & 31) != 0))
    return woq::fail("QBits: the SiLU*mul epilogue needs whole gate / up column-tile pairs");
  GemmF16Args a;
  const uint8_t* b = (const uint8_t*)blob;
  a.q = (const u32x4*)(b + h.off_q);
  a.scales = b + h.off_scale;
  a.zp = h.off_zp ? b + h.off_zp : nullptr;
  a.K = h.K;
  a.N = h.N;
  a.tiles_k = h.Kpad / WOQ_TILE_K;
  a.tiles_n = h.Npad / WOQ_TILE_N;
  a.n_groups = h.n_groups;
  a.group = h.group;
  a.scale_type = (int)h.scale_type;
  a.M = M;
  a.nb_m = (M + FBM - 1) / FBM;
  a.nb_m128 = a.nb_m;
  a.nb_n = (a.tiles_n * 16 + 127) / 128;
  const int sup_m = (a.nb_m + 7) / 8;
  a.sup_n = (a.nb_n + 7) / 8;
  a.n_sup = sup_m * a.sup_n;
  a.out = out;
  a.out_dtype = out_dtype;
  a.ldo = ldo;
  a.bias = bias;
  a.residual = residual;
  a.ld_res = ld_res;
  a.epi = epi;
  const size_t Mpad = (size_t)a.nb_m * FBM;
  // float weight types (4-bit tables; fp8 as two nibble planes, `h` = the high plane's header): B pre-dequantised into
  // MFMA fragments
  const bool frag = is_table_type(h.weight_type) || fp8_lo != nullptr;
  const size_t frag_bytes = frag ? (size_t)a.tiles_n * a.tiles_k * 4 * planes * 1024 : 0;
  const size_t base_bytes = gemm_f16_workspace_bytes(M, h.Kpad, h.Npad, planes) - SPLITK_WS;  // (a multiple of 256)
  // split-K: a call of one or two row blocks launches a few dozen workgroups on 256 CUs (M = 64: o / down 32, qkv 96,
  // gate / up 172 — 43-93 us per call); K slices bring the grid to ~512 workgroups, one round at two per CU. Each
  // slice keeps >= 4 K tiles, an even count.
  static const bool split_ok = !(getenv("WOQ_GEMM_SPLITK") && getenv("WOQ_GEMM_SPLITK")[0] == '0');
  int kper = 0, nz = 1;
  {
    const int wgs = a.nb_m * a.nb_n;
    // (measured on the Llama-2-7B prompt pass, 32 layers: 32 / 64 / 128 tokens 8.9 -> 5.6, 9.1 -> 6.0, 9.4 -> 6.6 ms;
    // at 512 tokens slicing the 128-workgroup o / down calls LOSES 10 %: the slices run the compiler-scheduled kernel
    // and pay the partials' round trip, so beyond one row block only really small grids are sliced)
    if (split_ok && !frag && wgs <= (a.nb_m == 1 ? 256 : 64) && a.tiles_k >= 8) {
      int want = std::min(16, std::max(2, 512 / wgs));
      kper = std::max(4, ((a.tiles_k + want - 1) / want + 1) & ~1);
      nz = (a.tiles_k + kper - 1) / kper;
      if (nz < 2) kper = 0, nz = 1;
    }
  }
  const int ldp = (h.Npad + 31) & ~31;
  size_t part_bytes = kper ? (size_t)nz * M * ldp * sizeof(float) : 0;
  if (part_bytes > SPLITK_WS) kper = 0, nz = 1, part_bytes = 0;  // (cannot happen: see SPLITK_WS)
  const size_t total = base_bytes + frag_bytes + part_bytes;  // [tiles | row scales | column scales][fragments][partials]
  if (frag && h.off_zp != 0) return woq::fail("QBits: float weight types are symmetric (no zero points)");
  // the fragment image (4-8x the blob): a caller's workspace holds it only when it says it is large enough (`ws_bytes`:
  // the engine sizes its prompt-pass workspace with gemm_f16_workspace_bytes_blob); otherwise per-call scratch
  if (frag && ws_bytes < total) ws = nullptr;
  unsigned char* w = (unsigned char*)ws;
  const bool mine = w == nullptr;  // no workspace passed in (the engine passes its own): take scratch
  bool own = false;
  if (mine) {
    w = (unsigned char*)scratch_take(total, st, &own);
    if (w == nullptr) return woq::fail("QBits: workspace allocation failed");
  }
  a.ap = (const _Float16*)w;
  a.rs = (const float*)(w + Mpad * h.Kpad * sizeof(_Float16) * planes);
  a.cs = a.rs + Mpad;
  a.bfrag = frag ? (const u32x4*)(w + base_bytes) : nullptr;
  a.kper = kper;
  a.part = kper ? (float*)(w + base_bytes + frag_bytes) : nullptr;
  a.ldp = ldp;

  // raw-A form: fp16 rows that need no gather, no RMSNorm and no rescale go to the hand-scheduled kernel as they are
  // (the o_proj / down_proj calls of the prompt pass); only the column scales are computed here
  static const bool raw_ok = !(getenv("WOQ_GEMM_RAW_A") && getenv("WOQ_GEMM_RAW_A")[0] == '0');
  const bool raw = WOQ_GEMM_HANDSCHED && raw_ok && !frag && !kper && !fp32_class && act_dtype == WOQ_F16 && norm_w == nullptr &&
                   h.off_shuffle == 0 && (h.K & 127) == 0 && ((h.Kpad / WOQ_TILE_K) & 1) == 0 && (lda & 7) == 0 &&
                   (((uintptr_t)act) & 15) == 0 && (size_t)M * lda * 2 < ((size_t)1 << 32);
  a.act_raw = raw ? act : nullptr;
  a.lda = lda;
  if (raw) a.rs = nullptr;
  static const bool ring_ok = !(getenv("WOQ_GEMM_RING") && getenv("WOQ_GEMM_RING")[0] == '0');
  const bool ring_fits = !(h.scale_mode == 1 && a.zp != nullptr && h.scale_type == WOQ_F32);  // (launch_f16_t: VGPRs)
  a.ring = (WOQ_GEMM_HANDSCHED && ring_ok && ring_fits && !frag && !kper && !fp32_class && ((h.Kpad / WOQ_TILE_K) & 1) == 0) ? 1 : 0;

  PackF16Args p;
  p.planes = planes;
  p.half_tiles = a.ring;
  p.x = act;
  p.x_dtype = act_dtype;
  p.lda = lda;
  p.M = M;
  p.Mpad = (int)Mpad;
  p.K = h.K;
  p.Kpad = h.Kpad;
  p.shuffle = h.off_shuffle ? (const int32_t*)(b + h.off_shuffle) : nullptr;
  p.norm_w = norm_w;
  p.eps = eps;
  p.ap = (_Float16*)a.ap;
  p.rs = (float*)a.rs;
  p.scales = a.scales;
  p.scale_type = a.scale_type;
  p.scale_mode = (int)h.scale_mode;
  p.n_groups = h.n_groups;
  p.tiles_k = a.tiles_k;
  p.Npad = h.Npad;
  p.row_blocks = raw ? 0 : (int)Mpad;  // raw-A: only the column-scale blocks run
  p.cs = (float*)a.cs;
  hipLaunchKernelGGL(pack_f16_kernel, dim3((unsigned)(p.row_blocks + (h.Npad + 31) / 32)), dim3(256), 0, st, p);

  if (g_gemm_ev0) hipEventRecord(g_gemm_ev0, st);  // measurement hook (woq_engine_time_prefill_gemm): the GEMM alone
  const bool asym = a.zp != nullptr;
  const int sm = (int)h.scale_mode;
  const bool s32 = h.scale_type == WOQ_F32;
  int rc = 1;
  if (frag) {
    DeqFragArgs d;
    d.q = a.q, d.q_lo = (const u32x4*)fp8_lo, d.scales = a.scales, d.scale_type = a.scale_type, d.scale_mode = sm, d.n_groups = h.n_groups;
    d.group = h.group, d.tiles_k = a.tiles_k, d.tiles_n = a.tiles_n, d.planes = planes, d.weight_type = fp8_lo ? fp8_type : h.weight_type;
    d.cs = a.cs, d.out = (u32x4*)a.bfrag;
    const size_t tiles = (size_t)a.tiles_n * a.tiles_k;
    hipLaunchKernelGGL(deq_frag_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, d);
    rc = fp32_class ? launch_f16frag_t<3>(a, st) : launch_f16frag_t<1>(a, st);
  } else {
#define WOQ_F16_CASE(SM, AS)                                                                   \
  if (sm == SM && asym == AS)                                                                   \
    rc = fp32_class ? (s32 ? launch_f16_t<SM, AS, true, 3>(a, st) : launch_f16_t<SM, AS, false, 3>(a, st))  \
                    : (s32 ? launch_f16_t<SM, AS, true, 1>(a, st) : launch_f16_t<SM, AS, false, 1>(a, st));
  WOQ_F16_CASE(0, false)
  WOQ_F16_CASE(0, true)
  WOQ_F16_CASE(1, false)
  WOQ_F16_CASE(1, true)
#undef WOQ_F16_CASE
  }
  if (rc == 0 && kper) {
    SplitKArgs r;
    r.part = a.part, r.nz = nz, r.M = M, r.N = h.N, r.ldp = ldp, r.epi = epi, r.bias = bias, r.residual = residual;
    r.ld_res = ld_res, r.out = out, r.out_dtype = out_dtype, r.ldo = ldo;
    const size_t elems = (size_t)M * (epi == 1 ? (h.N >> 1) : h.N);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, st, r);
  }
  if (g_gemm_ev1) hipEventRecord(g_gemm_ev1, st);
  if (mine) scratch_release(w, total, own, st);
  return rc;
}

// events recorded on the launch stream right before / after the GEMM kernel of the next launch_gemm_f16 calls
// (null = off): lets bench.py time the dominant prefill kernel without its pack pass
void set_gemm_time_events(hipEvent_t before, hipEvent_t after) {
  g_gemm_ev0 = before;
  g_gemm_ev1 = after;
}

}  // namespace woq
