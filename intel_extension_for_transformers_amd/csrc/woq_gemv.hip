// woq_gemv.hip — decode-side int4 weight x fp activation GEMV (M <= 8) for gfx950.
//
// Replaces the arithmetic behind qbits.woq_linear at small M:
//   qbits/qbits.cpp:113-140 -> bestla_weightonly_dispatcher.cpp:120-189 (do_compute) ->
//   BesTLA GemmRun<SchedulerBase> with LauncherBase<.., SCoreRowNAvx512f<48,8>, ShuffleActivationKBlockBaseF32,
//   WeightKBlockNInteger, AlphaBetaProcessStoreFp32> [external]: per N-tile x K-block unpack int4 -> fp32,
//   x scale, (asym: - zp * scale * sum(A) via the A-reduce prologue, dispatcher.cpp:154-160), fp32 FMA,
//   epilogue alpha*acc + beta*bias (bestla_customop.hpp:22-40).
// Parity definition: autograd/functions.py:41-63 (dequantize -> fp32 matmul -> + bias).
//
// MI355X design (HBM-bound: 0.5 B/weight, ~2 FLOP/weight; MFMA is NOT used here):
//  * one workgroup = CB adjacent 16-column tiles x all of K; its NW waves interleave over the
//    K tiles. One wave-level global_load_dwordx4 = one 1-KiB tile (16 columns x 128 k), fully
//    coalesced, issued straight to VGPRs, PF tiles deep, before anything waits (weights do not
//    depend on the activations, so the stream starts at wave launch).
//  * the activation vector is staged once per workgroup in LDS as fp32 (optionally RMSNorm'ed on
//    the way in: the fused prologue replaces a separate HF LlamaRMSNorm launch), together with
//    per-8 partial sums so that the -8 / zero-point offset costs one FMA per 8..32 weights:
//        sum_k (u_k - uz) x_k = sum_k u_k x_k - uz * sum_k x_k        (same identity BesTLA uses)
//  * nibble -> fp32 is v_cvt_f32_ubyteN on the even/odd-nibble split of a packed word (exact),
//    accumulate fp32, scale per group, then a 2-step wave shuffle over the 4 lanes sharing a
//    column and an LDS reduce over the waves.
//  * fused epilogues: bias, residual add (fp32 residual stream), SiLU(gate)*up over interleaved
//    gate/up tiles.
#include "woq_device.h"
#include "woq_launch.h"

namespace woq {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct GemvArgs {
  const u32x4* q;
  const void* scales;
  const uint8_t* zp;
  const int32_t* shuffle;
  int K, N, Kpad, tiles_k, n_groups, group, scale_type;
  const void* x;
  int x_dtype, lda, M;
  void* out;
  int out_dtype, ldo;
  const float* bias;
  const float* norm_w;  // fused RMSNorm prologue when non-null (fp32 [K])
  float eps;
  const float* residual;  // epilogue: out = residual + acc (fp32, row stride ld_res); may alias out
  int ld_res;
  int epi;  // 0 plain, 1 SiLU(gate)*up over (even, odd) tile pairs (needs CB == 2)
  int nt;   // reserved (weight loads are always non-temporal)
  int tpg_is1;          // group == 128
  unsigned tpg_magic;   // ceil(2^32 / (group / 128)) for the multiply-high division
};

constexpr int PF = 4;  // weight tiles in flight per wave per column block

template <int SMODE>
struct ScaleT;
template <>
struct ScaleT<0> {
  typedef float type;
};
template <>
struct ScaleT<1> {
  typedef float4_t type;
};

// 16-bit scale bits -> fp32 without a branch (both conversions are 1-2 VALU ops; select the right one)
__device__ __forceinline__ float scale16(uint32_t bits, bool is_bf16) {
  const float a = bf16_bits_to_f32((uint16_t)bits), b = f16_bits_to_f32((uint16_t)bits);
  return is_bf16 ? a : b;
}

// Branch-free tile fetch: out-of-range kt is clamped to the last tile and its scale forced to 0, so the
// instruction stream has no control flow around the loads (hipcc drains vmcnt(0) at every such branch —
// cdna_hip_programming.md "Three .s-level traps" (c)). S32: scales are fp32 (else 16-bit).
template <int SMODE, bool ASYM, bool S32>
__device__ __forceinline__ void load_tile(const GemvArgs& a, int tn, int kt, int lane, u32x4& w,
                                          typename ScaleT<SMODE>::type& sc, typename ScaleT<SMODE>::type& uz) {
  const int i = lane & 15;
  const bool valid = kt < a.tiles_k;
  const int ktc = valid ? kt : a.tiles_k - 1;
  const u32x4* p = a.q + ((size_t)tn * a.tiles_k + ktc) * 64 + lane;
  w = __builtin_nontemporal_load(p);  // streamed once: keep it out of the way of x / scales in L2
  const bool is_bf16 = a.scale_type == WOQ_BF16;
  if constexpr (SMODE == 0) {
    // group of this K tile = ktc / (group / 128), as a multiply-high (no divide, no branch)
    int grp = a.tpg_is1 ? ktc : (int)__umulhi((unsigned)ktc, a.tpg_magic);
    grp = a.n_groups > 1 ? min(grp, a.n_groups - 1) : 0;
    const size_t si = ((size_t)tn * a.n_groups + grp) * 16 + i;
    float v;
    if constexpr (S32)
      v = ((const float*)a.scales)[si];
    else
      v = scale16(((const uint16_t*)a.scales)[si], is_bf16);
    sc = valid ? v : 0.f;
    if constexpr (ASYM)
      uz = (float)a.zp[si];
    else
      uz = 8.f;
  } else {
    const size_t si = (((size_t)tn * a.tiles_k + ktc) * 16 + i) * 4;
    float4_t v;
    if constexpr (S32) {
      v = *(const float4_t*)((const float*)a.scales + si);
    } else {
      const uint2 r = *(const uint2*)((const uint16_t*)a.scales + si);
      v = (float4_t){scale16(r.x & 0xffff, is_bf16), scale16(r.x >> 16, is_bf16), scale16(r.y & 0xffff, is_bf16),
                     scale16(r.y >> 16, is_bf16)};
    }
    const float vm = valid ? 1.f : 0.f;
    sc = v * vm;
    if constexpr (ASYM) {
      const uint32_t z = *(const uint32_t*)(a.zp + si);
      uz = (float4_t){(float)(z & 0xff), (float)((z >> 8) & 0xff), (float)((z >> 16) & 0xff), (float)(z >> 24)};
    } else {
      uz = (float4_t){8.f, 8.f, 8.f, 8.f};
    }
  }
}

template <int MT, int SMODE>
__device__ __forceinline__ void consume_tile(const u32x4& w, const typename ScaleT<SMODE>::type& sc,
                                             const typename ScaleT<SMODE>::type& uz, int kt, int kq,
                                             const float* __restrict__ xs, const float* __restrict__ xsum, int Kpad,
                                             float (&tot)[MT]) {
  const float* xrow = xs + kt * 128 + kq * 8;
  const float* xsr = xsum + (kt * 4 + kq) * 4;
  const int xsum_ld = Kpad >> 3;
  float acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint32_t ws = w[s];
    const uint32_t e = ws & 0x0f0f0f0fu, o = (ws >> 4) & 0x0f0f0f0fu;
    // k-offset j lives at nibble position (j>>1)|((j&1)<<2): even positions are bytes of e, odd of o
    const float f0 = (float)(e & 0xffu), f1 = (float)((e >> 16) & 0xffu), f2 = (float)(o & 0xffu),
                f3 = (float)((o >> 16) & 0xffu), f4 = (float)((e >> 8) & 0xffu), f5 = (float)(e >> 24),
                f6 = (float)((o >> 8) & 0xffu), f7 = (float)(o >> 24);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float4_t xa = *(const float4_t*)(xrow + m * Kpad + s * 32);
      const float4_t xb = *(const float4_t*)(xrow + m * Kpad + s * 32 + 4);
      float p = f0 * xa.x;
      p = fmaf(f1, xa.y, p);
      p = fmaf(f2, xa.z, p);
      p = fmaf(f3, xa.w, p);
      p = fmaf(f4, xb.x, p);
      p = fmaf(f5, xb.y, p);
      p = fmaf(f6, xb.z, p);
      p = fmaf(f7, xb.w, p);
      if constexpr (SMODE == 0) {
        acc[m] += p;
      } else {
        const float xsv = xsr[m * xsum_ld + s];
        tot[m] = fmaf(sc[s], fmaf(-uz[s], xsv, p), tot[m]);
      }
    }
  }
  if constexpr (SMODE == 0) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float4_t x4 = *(const float4_t*)(xsr + m * xsum_ld);
      const float xsv = (x4.x + x4.y) + (x4.z + x4.w);
      tot[m] = fmaf(sc, fmaf(-uz, xsv, acc[m]), tot[m]);
    }
  }
}

template <int NW, int MT, int CB, int SMODE, bool ASYM, bool S32>
__global__ __launch_bounds__(NW * 64) void gemv_kernel(GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int T = NW * 64;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int kq = lane >> 4;
  const int tnb = blockIdx.x;
  const int m0 = blockIdx.y * MT;
  const int Kpad = a.Kpad;
  float* xs = smem;                       // [MT][Kpad]
  float* xsum = xs + MT * Kpad;           // [MT][Kpad/8]  order [kt][kq][s]
  float* red = xsum + MT * (Kpad >> 3);   // [NW][CB][MT][16]
  float* nrm = red + NW * CB * MT * 16;   // [NW][MT]

  typedef typename ScaleT<SMODE>::type sc_t;
  // ---- 1. start the weight stream: PF tiles per column block, nothing waits on them yet ----
  u32x4 wbuf[PF][CB];
  sc_t sbuf[PF][CB], zbuf[PF][CB];
#pragma unroll
  for (int p = 0; p < PF; ++p)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
      load_tile<SMODE, ASYM, S32>(a, tnb * CB + cb, wid + p * NW, lane, wbuf[p][cb], sbuf[p][cb], zbuf[p][cb]);

  // ---- 2. stage activations (fp32) + per-8 sums in LDS, optional fused RMSNorm ----
  const int nchunk = Kpad >> 3;
  const bool norm = a.norm_w != nullptr;
  float ss[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) ss[m] = 0.f;
  for (int c = tid; c < nchunk; c += T) {
    const int k0 = c * 8;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float v[8];
      const bool row_ok = (m0 + m) < a.M;
      const size_t rowoff = (size_t)(m0 + m) * a.lda;
      if (row_ok && !a.shuffle && k0 + 8 <= a.K && a.x_dtype != WOQ_F32 && ((rowoff + k0) & 7) == 0) {
        const uint4 r = *(const uint4*)((const uint16_t*)a.x + rowoff + k0);
        const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (a.x_dtype == WOQ_BF16) {
            v[2 * j] = bf16_bits_to_f32(rr[j] & 0xffff);
            v[2 * j + 1] = bf16_bits_to_f32(rr[j] >> 16);
          } else {
            v[2 * j] = f16_bits_to_f32(rr[j] & 0xffff);
            v[2 * j + 1] = f16_bits_to_f32(rr[j] >> 16);
          }
        }
      } else if (row_ok && !a.shuffle && k0 + 8 <= a.K && a.x_dtype == WOQ_F32 && ((rowoff + k0) & 3) == 0) {
        const float4_t r0 = *(const float4_t*)((const float*)a.x + rowoff + k0);
        const float4_t r1 = *(const float4_t*)((const float*)a.x + rowoff + k0 + 4);
        v[0] = r0.x; v[1] = r0.y; v[2] = r0.z; v[3] = r0.w;
        v[4] = r1.x; v[5] = r1.y; v[6] = r1.z; v[7] = r1.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = k0 + j;
          float t = 0.f;
          if (row_ok && k < a.K) t = load_f32(a.x, rowoff + (a.shuffle ? a.shuffle[k] : k), a.x_dtype);
          v[j] = t;
        }
      }
      float s8 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ss[m] = fmaf(v[j], v[j], ss[m]);
        s8 += v[j];
      }
      float* dst = xs + m * Kpad + k0;
      *(float4_t*)dst = (float4_t){v[0], v[1], v[2], v[3]};
      *(float4_t*)(dst + 4) = (float4_t){v[4], v[5], v[6], v[7]};
      if (!norm) xsum[m * nchunk + ((c >> 4) * 4 + (c & 3)) * 4 + ((c & 15) >> 2)] = s8;
    }
  }
  if (norm) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float t = wave_sum(ss[m]);
      if (lane == 0) nrm[wid * MT + m] = t;
    }
    __syncthreads();
    float inv[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float t = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) t += nrm[w2 * MT + m];
      inv[m] = 1.0f / sqrtf(t / (float)a.K + a.eps);  // HF LlamaRMSNorm: rsqrt(mean(x^2) + eps)
    }
    for (int c = tid; c < nchunk; c += T) {
      const int k0 = c * 8;
      float g[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = (k0 + j) < a.K ? a.norm_w[k0 + j] : 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        float* dst = xs + m * Kpad + k0;
        float s8 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = dst[j] * inv[m] * g[j];
          dst[j] = t;
          s8 += t;
        }
        xsum[m * nchunk + ((c >> 4) * 4 + (c & 3)) * 4 + ((c & 15) >> 2)] = s8;
      }
    }
  }
  __syncthreads();

  // ---- 3. stream the K tiles of this wave: consume tile, refill its slot ----
  float tot[CB][MT];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int m = 0; m < MT; ++m) tot[cb][m] = 0.f;

  for (int kt0 = wid; kt0 < a.tiles_k; kt0 += NW * PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int kt = kt0 + p * NW;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const u32x4 w = wbuf[p][cb];
        const sc_t sc = sbuf[p][cb], uz = zbuf[p][cb];
        load_tile<SMODE, ASYM, S32>(a, tnb * CB + cb, kt + NW * PF, lane, wbuf[p][cb], sbuf[p][cb], zbuf[p][cb]);
        // unconditional: a tile past the end was fetched clamped with scale 0 and contributes nothing
        consume_tile<MT, SMODE>(w, sc, uz, min(kt, a.tiles_k - 1), kq, xs, xsum, Kpad, tot[cb]);
      }
    }
  }

  // ---- 4. reduce: 4 lanes per column (shuffle), then NW waves (LDS) ----
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float t = reduce_kq(tot[cb][m]);
      if (lane < 16) red[((wid * CB + cb) * MT + m) * 16 + lane] = t;
    }
  __syncthreads();

  // ---- 5. epilogue ----
  for (int idx = tid; idx < CB * MT * 16; idx += T) {
    const int i = idx & 15, m = (idx >> 4) % MT, cb = idx / (MT * 16);
    if (m0 + m >= a.M) continue;
    float v = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) v += red[((w2 * CB + cb) * MT + m) * 16 + i];
    int n;
    if (a.epi == 1) {  // (gate, up) tile pair -> SiLU(gate) * up ; output column space is N/2
      if (cb != 0) continue;
      float u = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) u += red[((w2 * CB + 1) * MT + m) * 16 + i];
      n = tnb * 16 + i;
      if (n >= (a.N >> 1)) continue;
      if (a.bias) {
        v += a.bias[(tnb * CB) * 16 + i];
        u += a.bias[(tnb * CB + 1) * 16 + i];
      }
      v = v / (1.0f + __expf(-v)) * u;
    } else {
      n = (tnb * CB + cb) * 16 + i;
      if (n >= a.N) continue;
      if (a.bias) v += a.bias[n];
    }
    if (a.residual) v += a.residual[(size_t)(m0 + m) * a.ld_res + n];
    store_f32(a.out, (size_t)(m0 + m) * a.ldo + n, a.out_dtype, v);
  }
}

template <int NW, int MT, int CB, int SMODE, bool ASYM, bool S32>
static int launch_gemv_t(const GemvArgs& a, hipStream_t st) {
  const size_t lds = ((size_t)MT * a.Kpad + (size_t)MT * (a.Kpad >> 3) + (size_t)NW * CB * MT * 16 + NW * MT) * 4;
  if (lds > 160 * 1024) return woq::fail("QBits: activation tile does not fit LDS (K too large for this M tile)");
  auto kern = gemv_kernel<NW, MT, CB, SMODE, ASYM, S32>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return woq::fail(std::string("QBits: hipFuncSetAttribute: ") + hipGetErrorString(e));
    attr_set = true;
  }
  const int tiles_n = (a.N + 15) / 16;
  dim3 grid((tiles_n + CB - 1) / CB, (a.M + MT - 1) / MT);
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, st, a);
  return 0;
}

template <int NW, int MT, int CB>
static int launch_gemv_sm(const GemvArgs& a, int smode, bool asym, hipStream_t st) {
  const bool s32 = a.scale_type == WOQ_F32;
#define WOQ_GEMV_CASE(SM, AS, S3) \
  if (smode == SM && asym == AS && s32 == S3) return launch_gemv_t<NW, MT, CB, SM, AS, S3>(a, st);
  WOQ_GEMV_CASE(0, false, false)
  WOQ_GEMV_CASE(0, false, true)
  WOQ_GEMV_CASE(0, true, false)
  WOQ_GEMV_CASE(0, true, true)
  WOQ_GEMV_CASE(1, false, false)
  WOQ_GEMV_CASE(1, false, true)
  WOQ_GEMV_CASE(1, true, false)
  WOQ_GEMV_CASE(1, true, true)
#undef WOQ_GEMV_CASE
  return woq::fail("QBits: bad GEMV configuration");
}

// cb: 1 or 2 column tiles per workgroup (2 only with mt == 1); mt: rows per workgroup (1, 2 or 4)
int launch_gemv(const GemvArgs& a, int smode, bool asym, int cb, int mt, hipStream_t st) {
  constexpr int NW = 8;
  if (cb == 2) return launch_gemv_sm<NW, 1, 2>(a, smode, asym, st);
  if (mt == 1) return launch_gemv_sm<NW, 1, 1>(a, smode, asym, st);
  if (mt == 2) return launch_gemv_sm<NW, 2, 1>(a, smode, asym, st);
  return launch_gemv_sm<NW, 4, 1>(a, smode, asym, st);
}

}  // namespace woq

namespace woq {

int gemv_decode_max_rows(int Kpad);
int gemv_tile_max_rows(const void* act, int act_dtype, int lda, const woq_blob_header& h, const float* norm_w,
                       int epi);
int launch_gemv_tile(const void* act, int act_dtype, int lda, int M, const void* blob, const woq_blob_header& h,
                     const float* bias, void* out, int out_dtype, int ldo, const float* norm_w, float eps,
                     const float* residual, int ld_res, int epi, hipStream_t st);
int launch_gemv_decode(const void* act, int act_dtype, int lda, int M, const void* blob, const woq_blob_header& h,
                       const float* bias, void* out, int out_dtype, int ldo, const float* norm_w, float eps,
                       const float* residual, int ld_res, int epi, hipStream_t st);

// Build GemvArgs from a cached blob header and pick (CB, MT). Shared by woq_linear and the engine.
int launch_gemv_from_header(const void* act, int act_dtype, int lda, const void* blob, const woq_blob_header& h,
                            const float* bias, void* out, int out_dtype, int ldo, int M, const float* norm_w,
                            float eps, const float* residual, int ld_res, int epi, int nt, hipStream_t st) {
  // A/B switch (timing experiments): WOQ_GEMV_IMPL = tile (default) | persist | valu
  static const int impl = [] {
    const char* s = getenv("WOQ_GEMV_IMPL");
    return !s ? 0 : (s[0] == 'p' ? 1 : (s[0] == 'v' ? 2 : 0));
  }();
  // the per-token path: straight-line MFMA tile kernel (woq_gemv_tile.hip), up to 8 rows per launch
  const int mt_rows = impl == 0 ? gemv_tile_max_rows(act, act_dtype, lda, h, norm_w, epi) : 0;
  if (mt_rows > 0) {
    const size_t esz_o = out_dtype == WOQ_F32 ? 4 : 2;
    for (int m0 = 0; m0 < M; m0 += mt_rows) {
      const int mc = M - m0 < mt_rows ? M - m0 : mt_rows;
      int rc = launch_gemv_tile((const char*)act + (size_t)m0 * lda * 4, act_dtype, lda, mc, blob, h, bias,
                                (char*)out + (size_t)m0 * ldo * esz_o, out_dtype, ldo, norm_w, eps,
                                residual ? residual + (size_t)m0 * ld_res : nullptr, ld_res, epi, st);
      if (rc) return rc;
    }
    return 0;
  }
  const int mr = impl == 1 ? gemv_decode_max_rows(h.Kpad) : 0;
  if (mr > 0) {
    const size_t esz_a = act_dtype == WOQ_F32 ? 4 : 2, esz_o = out_dtype == WOQ_F32 ? 4 : 2;
    for (int m0 = 0; m0 < M; m0 += mr) {
      const int mc = M - m0 < mr ? M - m0 : mr;
      int rc = launch_gemv_decode((const char*)act + (size_t)m0 * lda * esz_a, act_dtype, lda, mc, blob, h, bias,
                                  (char*)out + (size_t)m0 * ldo * esz_o, out_dtype, ldo, norm_w, eps,
                                  residual ? residual + (size_t)m0 * ld_res : nullptr, ld_res, epi, st);
      if (rc) return rc;
    }
    return 0;
  }
  GemvArgs a;
  const uint8_t* b = (const uint8_t*)blob;
  a.q = (const u32x4*)(b + h.off_q);
  a.scales = b + h.off_scale;
  a.zp = h.off_zp ? b + h.off_zp : nullptr;
  a.shuffle = h.off_shuffle ? (const int32_t*)(b + h.off_shuffle) : nullptr;
  a.K = h.K;
  a.N = h.N;
  a.Kpad = h.Kpad;
  a.tiles_k = h.Kpad / WOQ_TILE_K;
  a.n_groups = h.n_groups;
  a.group = h.group;
  a.scale_type = (int)h.scale_type;
  a.x = act;
  a.x_dtype = act_dtype;
  a.lda = lda;
  a.M = M;
  a.out = out;
  a.out_dtype = out_dtype;
  a.ldo = ldo;
  a.bias = bias;
  a.norm_w = norm_w;
  a.eps = eps;
  a.residual = residual;
  a.ld_res = ld_res;
  a.epi = epi;
  a.nt = nt;
  {
    const unsigned tpg = h.group >= 128 ? (unsigned)(h.group >> 7) : 1u;
    a.tpg_is1 = tpg <= 1;
    a.tpg_magic = tpg <= 1 ? 0u : (unsigned)((0x100000000ull + tpg - 1) / tpg);
  }
  const int tiles_n = h.Npad / WOQ_TILE_N;
  // two column tiles per workgroup once there are enough tiles to keep > 2 workgroups per CU busy,
  // and always for the fused SiLU*mul epilogue (gate/up tile pairs)
  int cb = (epi == 1 || (M == 1 && tiles_n >= 1024)) ? 2 : 1;
  if (epi == 1 && (tiles_n & 1)) return woq::fail("QBits: fused gate/up weight needs an even number of column tiles");
  int mt = M >= 4 ? 4 : (M >= 2 ? 2 : 1);
  if (cb == 2) mt = 1;
  auto lds_bytes = [&](int mt_) { return ((size_t)mt_ * a.Kpad * 9 / 8 + 8 * 2 * mt_ * 16 + 64) * 4; };
  while (mt > 1 && lds_bytes(mt) > 150 * 1024) mt >>= 1;
  return launch_gemv(a, (int)h.scale_mode, a.zp != nullptr, cb, mt, st);
}

}  // namespace woq
