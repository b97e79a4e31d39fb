// woq_gemm.hip — prefill-side int4 weight x floating activation GEMM on the matrix cores (M > 4).
//
// Replaces the arithmetic behind qbits.woq_linear at large M: qbits.cpp:113-140 ->
// bestla_weightonly_dispatcher.cpp:150-178 (BesTLA HCoreRowNAmxbf16 / SCoreRowNAvx512f cores: per K-block unpack
// int4, apply scale / zero point, tile dot product, fp32 accumulate). Parity definition: autograd/functions.py:41-63.
//
// Here the dequantised tile really is a dense fp16 contraction, so it goes to v_mfma_f32_16x16x32_f16:
//  * workgroup tile 128 rows x 128 columns, K step 128 (one blob tile deep); 4 waves, wave w owns all 128 rows x
//    columns [32w, 32w+32) (2 column tiles) -> 8 x 2 accumulator fragments; every A fragment read from LDS feeds two
//    MFMAs (LDS bytes per MFMA halved), every dequantised B fragment feeds eight;
//  * B: each wave loads ITS two 1-KiB blob tiles straight to VGPRs (nothing is shared between waves, no LDS round
//    trip) and converts nibbles to fp16 with the exact magic-number form: (w ^ 0x88888888) makes the signed nibbles
//    unsigned, (x & 0x000f000f) | 0x64006400 = (1024 + u_a, 1024 + u_b), one v_pk_add_f16 of -(1032 + zp) -> q - zp
//    exactly. The contraction order inside an MFMA is free as long as A and B agree, so the MFMA for 64-k half h,
//    part p takes lane (column i, sixteenth kq)'s k = kq*16 + 8p + {0,2,4,6,1,3,5,7} — exactly the order the four
//    and/shift extractions of blob word #(2h+p) produce — and the A tile is stored in LDS in that same order;
//  * A: a pack pass (act_pack_kernel, one read of the activations) converts them ONCE to fp16 planes in block
//    floating point — the GEMM workgroups of all N/128 column blocks then re-read 2 B (4 B with the lo plane) per
//    element from L2 instead of converting 4-B values again each: per (row, K step) a power-of-two
//    scale from the row segment's max keeps |x| in fp16 range (exact scaling), hi = fp16(x 2^-e); with
//    compute_dtype fp32 a second plane lo = fp16(x 2^-e - hi) is contracted as well (two MFMAs per fragment pair,
//    ~2^-21 relative on the activation); with compute_dtype bf16 / fp16 / int8 only hi (2^-11), like the reference's
//    reduced-precision compute modes;
//  * the step accumulator is folded into the fp32 total with weight-scale[col] x 2^e[row] once per K step, so group
//    scales (any multiple of 128) and the block exponents never touch the fp16 operands;
//  * group-32 scales: a 64-k MFMA mixes two groups, so each is issued per group with the other group's A lanes
//    reading a zero block (2x the MFMAs; exact).
// The A tile of step k+1 is fetched into registers before the MFMAs of step k and written to the other LDS buffer
// after them: one barrier per K step. Not yet tuned further (no XCD-aware tile order, B through VGPRs only).
#include "woq_device.h"
#include "woq_launch.h"

namespace woq {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

int launch_gemv_from_header(const void* act, int act_dtype, int lda, const void* blob, const woq_blob_header& h,
                            const float* bias, void* out, int out_dtype, int ldo, int M, const float* norm_w,
                            float eps, const float* residual, int ld_res, int epi, int nt, hipStream_t st);

struct GemmArgs {
  const u32x4* q;
  const void* scales;
  const uint8_t* zp;
  int K, N, tiles_k, tiles_n, n_groups, group, scale_type;
  const void* x;
  int x_dtype, lda, M;
  const _Float16* a_hi;  // packed planes [M][Kpad] (fragment order inside every 8-group), block floating point
  const _Float16* a_lo;
  const float* a_rs;     // [M][tiles_k] 2^e of every (row, K step)
  int Kpad;
  void* out;
  int out_dtype, ldo;
  const float* bias;
  const float* residual;  // fp32 [M][ld_res] added in the epilogue, or null
  int ld_res;
};

constexpr int GBM = 128, GBN = 128, GKS = 128;
constexpr int GRS = GKS + 8;  // LDS row stride in halves (+16 B: conflict-free ds_read_b128 across rows)

// LDS: [zero row GRS halves] + 2 buffers of {A hi 128 x GRS, A lo 128 x GRS (NPASS == 2), row scales 128 f32}
__host__ __device__ constexpr size_t gemm_buf_bytes(int npass) { return (size_t)npass * GBM * GRS * 2 + GBM * 4; }
__host__ __device__ constexpr size_t gemm_lds_bytes(int npass) { return (size_t)GRS * 2 + 2 * gemm_buf_bytes(npass); }

// order in which the nibble extraction delivers a word's 8 k-offsets: element e of a fragment <-> k-offset GPERM[e]
__device__ __forceinline__ int gperm(int e) { return (e < 4) ? 2 * e : 2 * (e - 4) + 1; }

// 8 signed nibbles of one blob word -> 8 fp16 values (q - zp), order {0,2,4,6,1,3,5,7}
__device__ __forceinline__ h8 gdq8(uint32_t w, h2 c) {
  const uint32_t x = w ^ 0x88888888u;
  const uint32_t o0 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, (x & 0x000f000fu) | 0x64006400u) + c);
  const uint32_t o1 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, ((x >> 4) & 0x000f000fu) | 0x64006400u) + c);
  const uint32_t o2 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, ((x >> 8) & 0x000f000fu) | 0x64006400u) + c);
  const uint32_t o3 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, ((x >> 12) & 0x000f000fu) | 0x64006400u) + c);
  return __builtin_bit_cast(h8, (u32x4){o0, o1, o2, o3});
}

// Activation pack pass: [M, K] (fp32 | bf16 | fp16) -> fp16 plane(s) [M][Kpad] + per-(row, K step) scales.
// 16 lanes per row segment of 128 k (8 consecutive k = one fragment group each), row max by four DPP steps.
template <int NPASS>
__global__ __launch_bounds__(256) void act_pack_kernel(const void* __restrict__ x, int x_dtype, int lda, int M, int K,
                                                       int Kpad, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                       float* __restrict__ rs) {
  const int tid = threadIdx.x;
  const int g8 = tid & 15;
  const int tiles_k = Kpad / 128;
  const size_t seg = (size_t)blockIdx.x * 16 + (tid >> 4);  // (row, K step) segment index
  if (seg >= (size_t)M * tiles_k) return;
  const int r = (int)(seg / tiles_k), kt = (int)(seg % tiles_k);
  const int k0 = kt * 128 + g8 * 8;
  const size_t base = (size_t)r * lda;
  float v[8];
  if (x_dtype == WOQ_F32 && k0 + 8 <= K && ((lda | (int)(((uintptr_t)x) >> 2)) & 3) == 0) {
    const float4_t lo4 = *(const float4_t*)((const float*)x + base + k0);
    const float4_t hi4 = *(const float4_t*)((const float*)x + base + k0 + 4);
    v[0] = lo4.x, v[1] = lo4.y, v[2] = lo4.z, v[3] = lo4.w, v[4] = hi4.x, v[5] = hi4.y, v[6] = hi4.z, v[7] = hi4.w;
  } else if (x_dtype != WOQ_F32 && k0 + 8 <= K && ((lda | (int)(((uintptr_t)x) >> 1)) & 7) == 0) {
    const u32x4 raw = *(const u32x4*)((const uint16_t*)x + base + k0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint16_t bits = (uint16_t)(raw[j >> 1] >> (16 * (j & 1)));
      v[j] = x_dtype == WOQ_BF16 ? bf16_bits_to_f32(bits) : f16_bits_to_f32(bits);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = k0 + j < K ? load_f32(x, base + k0 + j, x_dtype) : 0.f;
  }
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[j]));
  amax = fmaxf(amax, WOQ_DPP_F32(amax, 0xB1));
  amax = fmaxf(amax, WOQ_DPP_F32(amax, 0x4E));
  amax = fmaxf(amax, WOQ_DPP_F32(amax, 0x141));
  amax = fmaxf(amax, WOQ_DPP_F32(amax, 0x140));  // every lane of the 16-lane row now holds the segment max
  int e = 0;
  if (amax > 0.f && amax < INFINITY) e = max(-100, min(100, __builtin_amdgcn_frexp_expf(amax) - 14));
  const float p2 = ldexpf(1.f, -e);  // |x| 2^-e < 2^14
  if (g8 == 0) rs[seg] = ldexpf(1.f, e);
  h8 hh, ll;
#pragma unroll
  for (int e8 = 0; e8 < 8; ++e8) {
    const float xs = v[gperm(e8)] * p2;
    const _Float16 hv = (_Float16)xs;
    hh[e8] = hv;
    ll[e8] = (_Float16)(xs - (float)hv);
  }
  *(h8*)(hi + (size_t)r * Kpad + k0) = hh;
  if constexpr (NPASS == 2) *(h8*)(lo + (size_t)r * Kpad + k0) = ll;
}

template <int NPASS, int SMODE, bool ASYM, bool S32>
__global__ __launch_bounds__(256, 1) void gemm_mfma_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm_raw[];
  _Float16* zrow = (_Float16*)gsm_raw;
  unsigned char* bufs = gsm_raw + GRS * 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kq = lane >> 4;
  const int row0 = (int)blockIdx.y * GBM;
  const int ct0 = (int)blockIdx.x * (GBN / 16) + wid * 2;  // this wave's first column tile
  if (tid < GRS / 2) ((uint32_t*)zrow)[tid] = 0u;

  float4_t tot[8][2];
#pragma unroll
  for (int rt = 0; rt < 8; ++rt)
#pragma unroll
    for (int c = 0; c < 2; ++c) tot[rt][c] = (float4_t){0.f, 0.f, 0.f, 0.f};

  // A-tile mover: 16 lanes per row (one 16-B fragment group each), 8 passes of 32 rows... 4 waves x 4 rows x 8 passes
  const int mv_g8 = tid & 15, mv_r = tid >> 4;  // row within a pass of 16 rows
  u32x4 ph[GBM / 16], pl[GBM / 16];
  float prs = 0.f;
  auto fetch_a = [&](int kt) {  // global -> registers
#pragma unroll
    for (int pass = 0; pass < GBM / 16; ++pass) {
      const int r = min(row0 + pass * 16 + mv_r, a.M - 1);
      const size_t off = (size_t)r * a.Kpad + kt * 128 + mv_g8 * 8;
      ph[pass] = *(const u32x4*)(a.a_hi + off);
      if constexpr (NPASS == 2) pl[pass] = *(const u32x4*)(a.a_lo + off);
    }
    if (tid < GBM) prs = a.a_rs[(size_t)min(row0 + tid, a.M - 1) * a.tiles_k + kt];
  };
  auto store_a = [&](int buf) {  // registers -> LDS buffer
    _Float16* dh = (_Float16*)(bufs + (size_t)buf * gemm_buf_bytes(NPASS));
    _Float16* dl = dh + GBM * GRS;
    float* dr = (float*)(dh + (size_t)NPASS * GBM * GRS);
#pragma unroll
    for (int pass = 0; pass < GBM / 16; ++pass) {
      *(u32x4*)(dh + (size_t)(pass * 16 + mv_r) * GRS + mv_g8 * 8) = ph[pass];
      if constexpr (NPASS == 2) *(u32x4*)(dl + (size_t)(pass * 16 + mv_r) * GRS + mv_g8 * 8) = pl[pass];
    }
    if (tid < GBM) dr[tid] = prs;
  };
  // scale fetch without a branch around the load (a dtype switch there makes hipcc wait vmcnt(0) per load)
  const bool sc_bf = a.scale_type == WOQ_BF16;
  auto ldsc = [&](size_t si) -> float {
    if constexpr (S32) {
      return ((const float*)a.scales)[si];
    } else {
      const uint16_t bits = ((const uint16_t*)a.scales)[si];
      const float fb = bf16_bits_to_f32(bits), fh = f16_bits_to_f32(bits);
      return sc_bf ? fb : fh;
    }
  };
  auto fetch_b = [&](int kt, u32x4 (&wv)[2], float (&wsc)[2][4], float (&wz)[2][4]) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int tn = min(ct0 + c, a.tiles_n - 1);
      wv[c] = a.q[((size_t)tn * a.tiles_k + kt) * 64 + lane];
      if constexpr (SMODE == 0) {
        int g = (kt * 128) / a.group;
        g = g >= a.n_groups ? a.n_groups - 1 : g;
        const size_t si = ((size_t)tn * a.n_groups + g) * 16 + i16;
        wsc[c][0] = ldsc(si);
        wz[c][0] = ASYM ? (float)((int)a.zp[si] - 8) : 0.f;
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const size_t si = ((((size_t)tn * a.tiles_k + kt) * 16 + i16) << 2) + s;
          wsc[c][s] = ldsc(si);
          wz[c][s] = ASYM ? (float)((int)a.zp[si] - 8) : 0.f;
        }
      }
    }
  };
  u32x4 wv[2], wvn[2];
  float wsc[2][4], wz[2][4], wscn[2][4], wzn[2][4];
  fetch_a(0);
  fetch_b(0, wv, wsc, wz);
  store_a(0);
  __syncthreads();

  for (int kt = 0; kt < a.tiles_k; ++kt) {
    const int cur = kt & 1;
    const _Float16* a_hi = (const _Float16*)(bufs + (size_t)cur * gemm_buf_bytes(NPASS));
    const _Float16* a_lo = a_hi + GBM * GRS;
    const float* rsc = (const float*)(a_hi + (size_t)NPASS * GBM * GRS);
    const int ktn = min(kt + 1, a.tiles_k - 1);
    fetch_a(ktn);  // next step's operands fly during this step's MFMAs
    fetch_b(ktn, wvn, wscn, wzn);

    // ---- contraction of this K step ----
    if constexpr (SMODE == 0) {
      float4_t acc[8][2];
#pragma unroll
      for (int rt = 0; rt < 8; ++rt)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[rt][c] = (float4_t){0.f, 0.f, 0.f, 0.f};
      // A fragments of blob word #hp are read one word AHEAD of their MFMAs (a lone wave per SIMD has nobody to hide
      // the ~130-cycle ds_read latency behind)
      h8 afh[2][8], afl[2][8];
      auto read_a = [&](int hp, h8 (&dh)[8], h8 (&dl)[8]) {
        const int koff = (hp >> 1) * 64 + kq * 16 + (hp & 1) * 8;
#pragma unroll
        for (int rt = 0; rt < 8; ++rt) {
          dh[rt] = *(const h8*)(a_hi + (size_t)(rt * 16 + i16) * GRS + koff);
          if constexpr (NPASS == 2) dl[rt] = *(const h8*)(a_lo + (size_t)(rt * 16 + i16) * GRS + koff);
        }
      };
      read_a(0, afh[0], afl[0]);
#pragma unroll
      for (int hp = 0; hp < 4; ++hp) {  // (half h, part p) = blob word #hp
        if (hp < 3) read_a(hp + 1, afh[(hp + 1) & 1], afl[(hp + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);  // keep the reads AHEAD: hipcc otherwise sinks each next to its first use
        h8 bfr[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const _Float16 cz = (_Float16)(-(1032.f + wz[c][0]));
          bfr[c] = gdq8(wv[c][hp], (h2){cz, cz});
        }
#pragma unroll
        for (int rt = 0; rt < 8; ++rt) {
#pragma unroll
          for (int c = 0; c < 2; ++c)
            acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afh[hp & 1][rt], bfr[c], acc[rt][c], 0, 0, 0);
          if constexpr (NPASS == 2) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
              acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afl[hp & 1][rt], bfr[c], acc[rt][c], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int rt = 0; rt < 8; ++rt) {
        const float4_t rs = *(const float4_t*)(rsc + rt * 16 + kq * 4);  // D rows 4*kq + {0..3} of row tile rt
#pragma unroll
        for (int c = 0; c < 2; ++c) tot[rt][c] += acc[rt][c] * (rs * wsc[c][0]);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {  // 32-k group s of the tile: half h = s >> 1, lanes kq>>1 == (s & 1)
        const int h = s >> 1;
        const bool mine = (kq >> 1) == (s & 1);
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) {  // row tiles in two halves: keeps the step accumulator at 4 x 2 fragments
          float4_t acc[4][2];
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[r4][c] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            h8 bfr[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const _Float16 cz = (_Float16)(-(1032.f + wz[c][s]));
              bfr[c] = gdq8(wv[c][2 * h + p], (h2){cz, cz});
            }
            const int koff = h * 64 + kq * 16 + p * 8;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              const int rt = rh * 4 + r4;
              const h8 ah = *(const h8*)(mine ? a_hi + (size_t)(rt * 16 + i16) * GRS + koff : zrow);
#pragma unroll
              for (int c = 0; c < 2; ++c)
                acc[r4][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bfr[c], acc[r4][c], 0, 0, 0);
              if constexpr (NPASS == 2) {
                const h8 al = *(const h8*)(mine ? a_lo + (size_t)(rt * 16 + i16) * GRS + koff : zrow);
#pragma unroll
                for (int c = 0; c < 2; ++c)
                  acc[r4][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bfr[c], acc[r4][c], 0, 0, 0);
              }
            }
          }
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const int rt = rh * 4 + r4;
            const float4_t rs = *(const float4_t*)(rsc + rt * 16 + kq * 4);
#pragma unroll
            for (int c = 0; c < 2; ++c) tot[rt][c] += acc[r4][c] * (rs * wsc[c][s]);
          }
        }
      }
    }
    store_a(cur ^ 1);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      wv[c] = wvn[c];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        wsc[c][j] = wscn[c][j];
        wz[c][j] = wzn[c][j];
      }
    }
    __syncthreads();
  }

  // ---- epilogue: + bias, store. D: lane (column i16, rows 4*kq + j) of every 16 x 16 fragment. The output-type switch
  //      sits OUTSIDE the store loops: a per-element switch makes hipcc wait vmcnt(0) after every store ----
  auto store_all = [&](auto put) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int n = (ct0 + c) * 16 + i16;
      if (ct0 + c >= a.tiles_n || n >= a.N) continue;
      const float bsv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
      for (int rt = 0; rt < 8; ++rt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int m = row0 + rt * 16 + kq * 4 + j;
          if (m < a.M)
            put((size_t)m * a.ldo + n, tot[rt][c][j] + bsv + (a.residual ? a.residual[(size_t)m * a.ld_res + n] : 0.f));
        }
    }
  };
  if (a.out_dtype == WOQ_F32)
    store_all([&](size_t i, float v) { ((float*)a.out)[i] = v; });
  else if (a.out_dtype == WOQ_BF16)
    store_all([&](size_t i, float v) { ((uint16_t*)a.out)[i] = f32_to_bf16_bits(v); });
  else
    store_all([&](size_t i, float v) { ((uint16_t*)a.out)[i] = f32_to_f16_bits(v); });
}

template <int NPASS, int SMODE, bool ASYM, bool S32>
static int launch_gemm_t(GemmArgs& a, hipStream_t st) {
  auto kern = gemm_mfma_kernel<NPASS, SMODE, ASYM, S32>;
  const size_t lds = gemm_lds_bytes(NPASS);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return woq::fail(std::string("QBits: hipFuncSetAttribute: ") + hipGetErrorString(e));
    attr_set = true;
  }
  // stream-ordered scratch for the packed activation planes (the reference allocates its A-side workspace per call
  // too when none was registered, bestla_weightonly_dispatcher.cpp:108-118,179)
  const size_t plane = (size_t)a.M * a.Kpad * sizeof(_Float16);
  const size_t rs_bytes = (size_t)a.M * a.tiles_k * sizeof(float);
  const size_t total = (size_t)NPASS * plane + rs_bytes;
  unsigned char* ws = nullptr;
  hipError_t e = hipMallocAsync((void**)&ws, total, st);
  if (e != hipSuccess) return woq::fail(std::string("QBits: workspace allocation failed: ") + hipGetErrorString(e));
  a.a_hi = (const _Float16*)ws;
  a.a_lo = (const _Float16*)(ws + (NPASS == 2 ? plane : 0));
  a.a_rs = (const float*)(ws + (size_t)NPASS * plane);
  const size_t segs = (size_t)a.M * a.tiles_k;
  hipLaunchKernelGGL(act_pack_kernel<NPASS>, dim3((unsigned)((segs + 15) / 16)), dim3(256), 0, st, a.x, a.x_dtype, a.lda,
                     a.M, a.K, a.Kpad, (_Float16*)a.a_hi, (_Float16*)a.a_lo, (float*)a.a_rs);
  const dim3 grid((unsigned)((a.tiles_n * 16 + GBN - 1) / GBN), (unsigned)((a.M + GBM - 1) / GBM));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
  hipFreeAsync(ws, st);
  return 0;
}

// out[M,N] = act[M,K] . W_deq (+ bias) for M above the small-M kernels' range. compute_type (blob header):
// fp32 -> two fp16 planes (hi + lo) per activation, anything else -> one.
int launch_gemm_mfma(const void* act, int act_dtype, int lda, const void* blob, const woq_blob_header& h,
                     const float* bias, void* out, int out_dtype, int ldo, int M, const float* residual, int ld_res,
                     hipStream_t st) {
  static const bool force_gemv = getenv("WOQ_GEMM_AS_GEMV") != nullptr;  // A/B switch for tests
  if (force_gemv || h.off_shuffle != 0 || (h.scale_mode == 0 && h.n_groups > 1 && (h.group % WOQ_TILE_K) != 0))
    return launch_gemv_from_header(act, act_dtype, lda, blob, h, bias, out, out_dtype, ldo, M, nullptr, 0.f, residual,
                                   ld_res, 0, 0, st);
  GemmArgs a;
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
  a.x = act;
  a.x_dtype = act_dtype;
  a.lda = lda;
  a.M = M;
  a.Kpad = h.Kpad;
  a.out = out;
  a.out_dtype = out_dtype;
  a.ldo = ldo;
  a.bias = bias;
  a.residual = residual;
  a.ld_res = ld_res;
  const bool two = h.compute_type == WOQ_C_FP32;
  const bool asym = a.zp != nullptr;
  const int sm = (int)h.scale_mode;
  const bool s32 = h.scale_type == WOQ_F32;
#define WOQ_GEMM_CASE(NP, SM, AS)                                                          \
  if (two == (NP == 2) && sm == SM && asym == AS)                                           \
    return s32 ? launch_gemm_t<NP, SM, AS, true>(a, st) : launch_gemm_t<NP, SM, AS, false>(a, st);
  WOQ_GEMM_CASE(1, 0, false)
  WOQ_GEMM_CASE(1, 0, true)
  WOQ_GEMM_CASE(1, 1, false)
  WOQ_GEMM_CASE(1, 1, true)
  WOQ_GEMM_CASE(2, 0, false)
  WOQ_GEMM_CASE(2, 0, true)
  WOQ_GEMM_CASE(2, 1, false)
  WOQ_GEMM_CASE(2, 1, true)
#undef WOQ_GEMM_CASE
  return woq::fail("QBits: bad GEMM configuration");
}

}  // namespace woq
