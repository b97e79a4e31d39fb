// woq_gemv_decode.hip — persistent, software-pipelined small-M (1..8 rows) int4 GEMV: the per-token hot kernel.
//
// Arithmetic and parity definition (reference): qbits.cpp:113-140 -> bestla_weightonly_dispatcher.cpp:120-189
// (per N-tile x K-block: unpack int4, apply scale / zero point, fp32 accumulate, epilogue
// alpha*acc + beta*bias, bestla_customop.hpp:22-40); definition autograd/functions.py:41-63.
//
// Schedule (what the batch-1 HBM roofline is about):
//  * the grid is sized to the chip (<= 2 workgroups per CU); each workgroup walks column steps
//    c = blockIdx.x, blockIdx.x + gridDim.x, ...; the activation rows (+ fused RMSNorm) are staged in LDS
//    ONCE per workgroup instead of once per 16 columns;
//  * a wave streams "units" of 4 consecutive K tiles (4 x global_load_dwordx4 nt = 4 KiB, straight to
//    VGPRs) and keeps two units in flight; the prefetch pointer runs across column-step boundaries, so
//    the HBM stream never drains at a reduction;
//  * one s_barrier per column step: waves drop their partial sums in a double-buffered LDS slab, wave 0
//    finishes (bias / residual / SiLU*mul) while the others already consume the next step.
//  * all control flow around the loads is wave-uniform and contains no VMEM, so hipcc keeps counted
//    vmcnt(N) waits (cdna_hip_programming.md "Three .s-level traps").
//
// Inner product. Measured on MI355X: the straightforward VALU form (v_cvt_f32_ubyte + v_fma, kept in
// woq_gemv.hip) needs ~150 VALU instructions per 1-KiB tile and is ISSUE-bound at ~2 TB/s, far under the
// HBM roof (profiles/r01_gemv_valu_*). This kernel therefore
//  (a) dequantises with the packed-fp16 magic-number form
//        (w & 0x000f000f) | 0x64006400 = (1024 + u_a, 1024 + u_b)   (w & 0x00f000f0) | 0x54005400 = (64 + u_c, 64 + u_d)
//      followed by one v_pk_add_f16 of -(1024|64 + uz): 9 VALU per 8 weights, exact small integers; and
//  (b) hands the multiply-accumulate to the otherwise idle matrix pipe: v_mfma_f32_16x16x32_f16 with
//      B = the dequantised 16-column x 32-k fragment (a blob tile IS four such fragments) and A = the
//      activation rows. Row 2m of A holds hi = fp16(x_m * 2^-e), row 2m+1 holds lo = fp16(x_m * 2^-e - hi):
//      both partial products are exact in fp32 and their sum carries x to ~2^-22 relative, i.e. fp32-class
//      accuracy for fp32, bf16 and fp16 activations alike (e: per-call power of two keeping |x| in fp16
//      range). Up to 8 rows ride along for free, and the column sums come out of the MFMA already reduced
//      over k, so this path has no cross-lane shuffle reduction at all.
#include "woq_device.h"
#include "woq_launch.h"

namespace woq {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

struct DecodeArgs {
  const u32x4* q;
  const void* scales;
  const uint8_t* zp;
  const int32_t* shuffle;
  int K, N, Kpad, tiles_k, n_groups, scale_is_bf16;
  int tpg_is1;
  unsigned tpg_magic;
  const void* x;
  int x_dtype, lda, M;
  void* out;
  int out_dtype, ldo, ld_res;
  const float* bias;
  const float* norm_w;
  float eps;
  const float* residual;
  int epi;         // 0 plain, 1 SiLU(gate)*up over (even, odd) tile pairs
  int cb_n;        // column tiles per step (1 or 2)
  int n_colsteps;  // ceil(tiles_n / cb_n)
  int upc;         // units (4 K tiles) per column per wave
  int debug;       // timing experiments only: 1 = no streaming, 3 = no activation staging, 4 = exit at once
};

constexpr int UT = 4;    // K tiles per unit
constexpr int MAXM = 8;  // activation rows per launch (16 MFMA rows / (hi, lo))

template <int SMODE>
struct DSc;
template <>
struct DSc<0> {
  typedef float type;
};
template <>
struct DSc<1> {
  typedef float4_t type;
};

template <int SMODE>
struct Unit {
  u32x4 w[UT];
  typename DSc<SMODE>::type sc[UT];
  typename DSc<SMODE>::type uz[UT];
};

__device__ __forceinline__ float dscale16(uint32_t bits, bool is_bf16) {
  const float a = bf16_bits_to_f32((uint16_t)bits), b = f16_bits_to_f32((uint16_t)bits);
  return is_bf16 ? a : b;
}

// fetch one unit: column tile tn, K tiles kt0 .. kt0+3. `live` = 0 turns the unit into a no-op
// (scale 0, clamped addresses) without any branch around the loads.
template <int SMODE, bool ASYM, bool S32>
__device__ __forceinline__ void load_unit(const DecodeArgs& a, int tn, int kt0, bool live, int lane,
                                          Unit<SMODE>& u) {
  const int i = lane & 15;
  const bool bf = a.scale_is_bf16 != 0;
#pragma unroll
  for (int t = 0; t < UT; ++t) {
    const int kt = kt0 + t;
    const bool valid = live && kt < a.tiles_k;
    const int ktc = min(kt, a.tiles_k - 1);
    u.w[t] = __builtin_nontemporal_load(a.q + ((size_t)tn * a.tiles_k + ktc) * 64 + lane);
    if constexpr (SMODE == 0) {
      int grp = a.tpg_is1 ? ktc : (int)__umulhi((unsigned)ktc, a.tpg_magic);
      grp = a.n_groups > 1 ? min(grp, a.n_groups - 1) : 0;
      const size_t si = ((size_t)tn * a.n_groups + grp) * 16 + i;
      float v;
      if constexpr (S32)
        v = ((const float*)a.scales)[si];
      else
        v = dscale16(((const uint16_t*)a.scales)[si], bf);
      u.sc[t] = valid ? v : 0.f;
      if constexpr (ASYM)
        u.uz[t] = (float)a.zp[si];
      else
        u.uz[t] = 8.f;
    } else {
      const size_t si = (((size_t)tn * a.tiles_k + ktc) * 16 + i) * 4;
      float4_t v;
      if constexpr (S32) {
        v = *(const float4_t*)((const float*)a.scales + si);
      } else {
        const uint2 r = *(const uint2*)((const uint16_t*)a.scales + si);
        v = (float4_t){dscale16(r.x & 0xffff, bf), dscale16(r.x >> 16, bf), dscale16(r.y & 0xffff, bf),
                       dscale16(r.y >> 16, bf)};
      }
      u.sc[t] = v * (valid ? 1.f : 0.f);
      if constexpr (ASYM) {
        const uint32_t z = *(const uint32_t*)(a.zp + si);
        u.uz[t] = (float4_t){(float)(z & 0xff), (float)((z >> 8) & 0xff), (float)((z >> 16) & 0xff), (float)(z >> 24)};
      } else {
        u.uz[t] = (float4_t){8.f, 8.f, 8.f, 8.f};
      }
    }
  }
}

// 8 packed int4 (one u32, interleaved nibble order) -> 8 fp16 values (u - uz), k-consecutive, as 4 dwords
__device__ __forceinline__ void dq8(uint32_t w, h2 c1024, h2 c64, uint32_t (&o)[4]) {
  const uint32_t w8 = w >> 8;
  o[0] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, (w & 0x000f000fu) | 0x64006400u) + c1024);
  o[1] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, (w & 0x00f000f0u) | 0x54005400u) + c64);
  o[2] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, (w8 & 0x000f000fu) | 0x64006400u) + c1024);
  o[3] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, (w8 & 0x00f000f0u) | 0x54005400u) + c64);
}

// One 1-KiB tile on the matrix pipe. a_base: this lane's A-row base in LDS (halves); a_mask: ~0 for a live
// row, 0 for an unused row (then every read lands in the 256-B zero block a_base points at). Accumulates the
// scaled partial sums of this lane's two activation rows (D rows 4*(lane>>4) + {0,1} and + {2,3}) for
// column lane & 15.
template <int SMODE>
__device__ __forceinline__ void tile_mfma(const u32x4& w, const typename DSc<SMODE>::type& sc,
                                          const typename DSc<SMODE>::type& uz, int kt, const _Float16* a_base,
                                          unsigned a_mask, float& tot_a, float& tot_b) {
  const _Float16* ap = a_base + ((unsigned)(kt * 128) & a_mask);
  float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float z;
    if constexpr (SMODE == 0)
      z = uz;
    else
      z = uz[s];
    const _Float16 z1 = (_Float16)(-(1024.f + z)), z2 = (_Float16)(-(64.f + z));
    const h2 c1024 = {z1, z1}, c64 = {z2, z2};
    uint32_t b[4];
    dq8(w[s], c1024, c64, b);
    const h8 bfr = __builtin_bit_cast(h8, (u32x4){b[0], b[1], b[2], b[3]});
    const h8 afr = *(const h8*)(ap + s * 32);
    if constexpr (SMODE == 0) {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr, bfr, acc, 0, 0, 0);
    } else {
      const float4_t zero = {0.f, 0.f, 0.f, 0.f};
      const float4_t d = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr, bfr, zero, 0, 0, 0);
      tot_a = fmaf(sc[s], d.x + d.y, tot_a);
      tot_b = fmaf(sc[s], d.z + d.w, tot_b);
    }
  }
  if constexpr (SMODE == 0) {
    tot_a = fmaf(sc, acc.x + acc.y, tot_a);
    tot_b = fmaf(sc, acc.z + acc.w, tot_b);
  }
}

// LDS (dynamic): [zero block 256 B][A rows: 2*M x Kpad halves][red: 2 x NW x 2 x 128 floats][misc]
__host__ __device__ inline size_t decode_lds_bytes(int M, int Kpad, int NW) {
  return 256 + (size_t)2 * M * Kpad * 2 + (size_t)2 * NW * 2 * 128 * 4 + (size_t)(2 * NW + 8 + 2 * MAXM) * 4;
}

constexpr int XV = 6;  // fast activation path: float4 loads per thread (covers K <= NW*64*4*XV = 12288 at NW = 8)

// FASTX: M == 1, fp32 activations, no shuffle, K % 4 == 0, K <= NW*64*4*XV, 16-B aligned row: the row (and the
// RMSNorm weight) is fetched with XV unconditional float4 loads per thread BEFORE the weight prefetch, stays in
// registers across the statistics and the hi/lo split, and costs two barriers. The generic path below handles
// everything else (any dtype / shuffle / M <= 8) but pays one dependent L2 round trip per element group.
template <int NW, int SMODE, bool ASYM, bool S32, bool FASTX>
__global__ __launch_bounds__(NW * 64) void gemv_decode_kernel(DecodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int T = NW * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Kpad = a.Kpad, M = a.M;
  _Float16* zero_blk = (_Float16*)smem_raw;             // 128 halves of zeros
  _Float16* arows = zero_blk + 128;                     // [2*M][Kpad]
  float* red = (float*)(arows + (size_t)2 * M * Kpad);  // [2][NW][2][128]
  float* misc = red + 2 * NW * 2 * 128;                 // [NW] sumsq partials, [NW] absmax partials, per-row scales

  const int upc = a.upc, cbn = a.cb_n;
  const int units_per_step = upc * cbn;
  const int my_steps =
      ((int)blockIdx.x < a.n_colsteps) ? (a.n_colsteps - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int total_units = a.debug == 1 ? 0 : my_steps * units_per_step;
  const int tiles_n = (a.N + 15) >> 4;
  if (a.debug == 4) return;

  // prefetch pointer (wave-uniform): step index pi, column-in-step pcb, unit-in-column pu
  int pi = 0, pcb = 0, pu = 0, pg = 0;
  auto fetch = [&](Unit<SMODE>& u) {
    const bool live = pg < total_units;
    const int step = min((int)blockIdx.x + pi * (int)gridDim.x, a.n_colsteps - 1);
    const int tn = min(step * cbn + pcb, tiles_n - 1);
    load_unit<SMODE, ASYM, S32>(a, tn, (pu * NW + wid) * UT, live, lane, u);
    ++pg;
    ++pu;
    if (pu == upc) {
      pu = 0;
      ++pcb;
      if (pcb == cbn) {
        pcb = 0;
        ++pi;
      }
    }
  };

  const bool norm = a.norm_w != nullptr;
  float* inv_s = misc + 2 * NW;  // [MAXM] rsqrt(mean(x^2)+eps) per row (1 when no norm)
  float* pow2_s = inv_s + MAXM;  // [MAXM] 2^-e per row

  // ---- 0. (fast path) issue the activation row loads first: they return first and need no branch ----
  float4_t xv[XV], gv[XV];
  if constexpr (FASTX) {
    const float* xp = (const float*)a.x;
    const float* gp = norm ? a.norm_w : xp;  // dummy (ignored) source when there is no norm: keeps it branch-free
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int k = (tid + j * T) * 4;
      const int kc = k < a.K ? k : 0;
      xv[j] = *(const float4_t*)(xp + kc);
      gv[j] = *(const float4_t*)(gp + kc);
    }
  }

  // ---- 1. start the weight stream (weights do not depend on the activations) ----
  Unit<SMODE> ua, ub;
  fetch(ua);
  fetch(ub);

  // ---- 2. stage the activation rows once per workgroup: (RMSNorm) -> power-of-two scale -> hi/lo fp16 ----
  if (tid < 64) ((uint32_t*)zero_blk)[tid] = 0u;
  if (a.debug == 3) {
    if (tid == 0) pow2_s[0] = 1.f;
  } else if constexpr (FASTX) {
    float ss = 0.f, amax = 0.f;
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const bool ok = (tid + j * T) * 4 < a.K;
      if (!ok) xv[j] = (float4_t){0.f, 0.f, 0.f, 0.f};
      if (!norm) gv[j] = (float4_t){1.f, 1.f, 1.f, 1.f};
      ss = fmaf(xv[j].x, xv[j].x, ss);
      ss = fmaf(xv[j].y, xv[j].y, ss);
      ss = fmaf(xv[j].z, xv[j].z, ss);
      ss = fmaf(xv[j].w, xv[j].w, ss);
      xv[j] = xv[j] * gv[j];
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(xv[j].x), fabsf(xv[j].y)), fmaxf(fabsf(xv[j].z), fabsf(xv[j].w))));
    }
    ss = wave_sum(ss);
    amax = wave_max(amax);
    if (lane == 0) {
      misc[wid] = ss;
      misc[NW + wid] = amax;
    }
    __syncthreads();
    float t = 0.f, mx = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) {
      t += misc[w2];
      mx = fmaxf(mx, misc[NW + w2]);
    }
    const float inv = norm ? 1.0f / sqrtf(t / (float)a.K + a.eps) : 1.f;  // HF LlamaRMSNorm
    mx *= inv;
    int e = 0;
    if (mx > 0.f && mx < INFINITY) e = max(-40, min(40, (int)ceilf(log2f(mx)) - 13));
    const float p2 = exp2f((float)-e);
    if (tid == 0) pow2_s[0] = p2;
    const float f = inv * p2;
    _Float16* hi = arows;
    _Float16* lo = hi + Kpad;
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int k = (tid + j * T) * 4;
      if (k < Kpad) {  // k >= K lanes hold zeros: they also clear the K padding
        const float4_t v = xv[j] * f;
        const _Float16 h0 = (_Float16)v.x, h1 = (_Float16)v.y, h2v = (_Float16)v.z, h3 = (_Float16)v.w;
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        *(h4*)(hi + k) = (h4){h0, h1, h2v, h3};
        *(h4*)(lo + k) = (h4){(_Float16)(v.x - (float)h0), (_Float16)(v.y - (float)h1),
                              (_Float16)(v.z - (float)h2v), (_Float16)(v.w - (float)h3)};
      }
    }
  } else
  for (int m = 0; m < M; ++m) {
    const size_t rowoff = (size_t)m * a.lda;
    float ss = 0.f, amax = 0.f;
    for (int k = tid; k < a.K; k += T) {
      const float v = load_f32(a.x, rowoff + (a.shuffle ? a.shuffle[k] : k), a.x_dtype);
      ss = fmaf(v, v, ss);
      amax = fmaxf(amax, fabsf(norm ? v * a.norm_w[k] : v));
    }
    ss = wave_sum(ss);
    amax = wave_max(amax);
    if (lane == 0) {
      misc[wid] = ss;
      misc[NW + wid] = amax;
    }
    __syncthreads();
    if (tid == 0) {
      float t = 0.f, mx = 0.f;
      for (int w2 = 0; w2 < NW; ++w2) {
        t += misc[w2];
        mx = fmaxf(mx, misc[NW + w2]);
      }
      const float inv = norm ? 1.0f / sqrtf(t / (float)a.K + a.eps) : 1.f;  // HF LlamaRMSNorm
      mx *= inv;
      // 2^-e with |x| * 2^-e <= 2^13: exact scaling, comfortably inside fp16 range
      int e = 0;
      if (mx > 0.f && mx < INFINITY) e = max(-40, min(40, (int)ceilf(log2f(mx)) - 13));
      inv_s[m] = inv;
      pow2_s[m] = exp2f((float)-e);
    }
    __syncthreads();
    const float inv = inv_s[m], p2 = pow2_s[m];
    _Float16* hi = arows + (size_t)(2 * m) * Kpad;
    _Float16* lo = hi + Kpad;
    for (int k = tid; k < Kpad; k += T) {
      float v = 0.f;
      if (k < a.K) {
        v = load_f32(a.x, rowoff + (a.shuffle ? a.shuffle[k] : k), a.x_dtype);
        if (norm) v = v * inv * a.norm_w[k];
        v *= p2;
      }
      const _Float16 h = (_Float16)v;
      hi[k] = h;
      lo[k] = (_Float16)(v - (float)h);
    }
  }
  __syncthreads();
  // this lane's D rows: 4*(lane>>4) + {0,1} -> activation row 2*(lane>>4); + {2,3} -> row 2*(lane>>4) + 1
  const int ma = 2 * (lane >> 4), mb = ma + 1;
  const float out_scale_a = ma < M ? 1.0f / pow2_s[ma] : 0.f;
  const float out_scale_b = mb < M ? 1.0f / pow2_s[mb] : 0.f;
  // this lane's A row: MFMA row r = lane & 15 -> activation row r >> 1, part (hi | lo) r & 1
  const int arow = lane & 15;
  const bool a_live = arow < 2 * M;
  const unsigned a_mask = a_live ? 0xffffffffu : 0u;
  const _Float16* a_base = (a_live ? arows + (size_t)arow * Kpad : zero_blk) + (lane >> 4) * 8;

  // ---- 3. stream: consume one unit, refill its slot, flush at column / step boundaries ----
  int ci = 0, ccb = 0, cu = 0, buf = 0;
  float tot_a = 0.f, tot_b = 0.f;
  auto consume = [&](const Unit<SMODE>& u) {
    const int kt0 = (cu * NW + wid) * UT;
#pragma unroll
    for (int t = 0; t < UT; ++t)
      tile_mfma<SMODE>(u.w[t], u.sc[t], u.uz[t], min(kt0 + t, a.tiles_k - 1), a_base, a_mask, tot_a, tot_b);
    ++cu;
    if (cu == upc) {  // column finished for this wave
      cu = 0;
      // slab layout [buf][wave][cb][row-pair slot lane>>4][a|b][16 columns] = 128 floats per (wave, cb)
      float* dst = red + (((size_t)buf * NW + wid) * 2 + ccb) * 128 + (lane >> 4) * 32 + (lane & 15);
      dst[0] = tot_a * out_scale_a;
      dst[16] = tot_b * out_scale_b;
      tot_a = 0.f;
      tot_b = 0.f;
      ++ccb;
      if (ccb == cbn) {  // column step finished for the workgroup
        ccb = 0;
        __syncthreads();
        if (wid == 0) {
          const int step = (int)blockIdx.x + ci * (int)gridDim.x;
          // 16 columns x M rows (x cbn column tiles when not fused): the 64 lanes loop over them
          const int ncb = a.epi == 1 ? 1 : cbn;
          for (int idx = lane; idx < ncb * M * 16; idx += 64) {
            const int i = idx & 15, m = (idx >> 4) % M, cb = idx / (16 * M);
            const int slot = (m >> 1) * 32 + (m & 1) * 16 + i;
            float v = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) v += red[(((size_t)buf * NW + w2) * 2 + cb) * 128 + slot];
            int n;
            bool ok;
            if (a.epi == 1) {
              float up = 0.f;
#pragma unroll
              for (int w2 = 0; w2 < NW; ++w2) up += red[(((size_t)buf * NW + w2) * 2 + 1) * 128 + slot];
              n = step * 16 + i;
              ok = n < (a.N >> 1);
              if (a.bias) {
                v += a.bias[min((step * 2) * 16 + i, a.N - 1)];
                up += a.bias[min((step * 2 + 1) * 16 + i, a.N - 1)];
              }
              v = v / (1.0f + __expf(-v)) * up;
            } else {
              n = (step * cbn + cb) * 16 + i;
              ok = n < a.N;
              if (a.bias) v += a.bias[min(n, a.N - 1)];
            }
            if (ok) {
              if (a.residual) v += a.residual[(size_t)m * a.ld_res + n];
              store_f32(a.out, (size_t)m * a.ldo + n, a.out_dtype, v);
            }
          }
        }
        buf ^= 1;
        ++ci;
      }
    }
  };

  for (int g = 0; g < total_units; g += 2) {
    {
      const Unit<SMODE> cur = ua;
      fetch(ua);
      consume(cur);
    }
    if (g + 1 < total_units) {
      const Unit<SMODE> cur = ub;
      fetch(ub);
      consume(cur);
    }
  }
}

template <int NW, int SMODE, bool ASYM, bool S32, bool FASTX>
static int launch_decode_t(const DecodeArgs& a, int grid, hipStream_t st) {
  const size_t lds = decode_lds_bytes(a.M, a.Kpad, NW);
  if (lds > 160 * 1024) return woq::fail("QBits: activation rows do not fit LDS");
  auto kern = gemv_decode_kernel<NW, SMODE, ASYM, S32, FASTX>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return woq::fail(std::string("QBits: hipFuncSetAttribute: ") + hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, a);
  return 0;
}

template <int NW>
static int launch_decode_nw(const DecodeArgs& a, int smode, bool asym, bool s32, int grid, hipStream_t st) {
  const bool fastx = a.M == 1 && a.x_dtype == WOQ_F32 && !a.shuffle && (a.K & 3) == 0 && a.K <= NW * 64 * 4 * XV &&
                     (a.Kpad <= NW * 64 * 4 * XV) && (((uintptr_t)a.x) & 15) == 0 &&
                     (!a.norm_w || (((uintptr_t)a.norm_w) & 15) == 0);
#define WOQ_DEC_CASE(SM, AS, S3)                                                              \
  if (smode == SM && asym == AS && s32 == S3)                                                 \
    return fastx ? launch_decode_t<NW, SM, AS, S3, true>(a, grid, st) : launch_decode_t<NW, SM, AS, S3, false>(a, grid, st);
  WOQ_DEC_CASE(0, false, false)
  WOQ_DEC_CASE(0, false, true)
  WOQ_DEC_CASE(0, true, false)
  WOQ_DEC_CASE(0, true, true)
  WOQ_DEC_CASE(1, false, false)
  WOQ_DEC_CASE(1, false, true)
  WOQ_DEC_CASE(1, true, false)
  WOQ_DEC_CASE(1, true, true)
#undef WOQ_DEC_CASE
  return woq::fail("QBits: bad decode GEMV configuration");
}

static int g_num_cus = 0;
static int g_decode_wgs_per_cu = 2;
static int g_debug = 0;  // WOQ_DEBUG_DECODE: 1 = staging only (no column steps), 2 = clamp to one step per WG

// largest M this kernel takes for a given K (LDS budget), 0 if not even one row fits
int gemv_decode_max_rows(int Kpad) {
  int m = MAXM;
  while (m > 0 && decode_lds_bytes(m, Kpad, 8) > 150 * 1024) --m;
  return m;
}

// rows 0..M-1 (M <= gemv_decode_max_rows). x: [M, lda]; out: [M, ldo]; residual: [M, ld_res] or null.
int launch_gemv_decode(const void* act, int act_dtype, int lda, int M, const void* blob, const woq_blob_header& h,
                       const float* bias, void* out, int out_dtype, int ldo, const float* norm_w, float eps,
                       const float* residual, int ld_res, int epi, hipStream_t st) {
  if (g_num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
      return woq::fail("QBits: cannot query the HIP device");
    g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char* s = getenv("WOQ_DECODE_WGS_PER_CU")) g_decode_wgs_per_cu = atoi(s) > 0 ? atoi(s) : 2;
    if (const char* s = getenv("WOQ_DEBUG_DECODE")) g_debug = atoi(s);
  }
  DecodeArgs a;
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
  a.scale_is_bf16 = h.scale_type == WOQ_BF16;
  const unsigned tpg = h.group >= 128 ? (unsigned)(h.group >> 7) : 1u;
  a.tpg_is1 = tpg <= 1;
  a.tpg_magic = tpg <= 1 ? 0u : (unsigned)((0x100000000ull + tpg - 1) / tpg);
  a.x = act;
  a.x_dtype = act_dtype;
  a.lda = lda;
  a.M = M;
  a.out = out;
  a.out_dtype = out_dtype;
  a.ldo = ldo;
  a.ld_res = ld_res;
  a.bias = bias;
  a.norm_w = norm_w;
  a.eps = eps;
  a.residual = residual;
  a.epi = epi;
  const int tiles_n = h.Npad / WOQ_TILE_N;
  if (epi == 1 && (tiles_n & 1)) return woq::fail("QBits: fused gate/up weight needs an even number of column tiles");
  a.cb_n = epi == 1 ? 2 : 1;
  a.n_colsteps = (tiles_n + a.cb_n - 1) / a.cb_n;
  constexpr int nw = 8;
  a.upc = (a.tiles_k + UT * nw - 1) / (UT * nw);
  // strided step assignment over a chip-sized grid: workgroups b and b + #CUs tend to share a CU, so a CU
  // sees steps b, b + #CUs, b + 2 #CUs, ... and the per-CU byte count stays within one step of the mean
  const int max_wgs = g_num_cus * g_decode_wgs_per_cu;
  const int grid = a.n_colsteps < max_wgs ? a.n_colsteps : max_wgs;
  a.debug = g_debug;  // timing experiments only (results are garbage when non-zero)
  if (g_debug == 2) a.n_colsteps = grid;
  const bool s32 = h.scale_type == WOQ_F32;
  return launch_decode_nw<nw>(a, (int)h.scale_mode, a.zp != nullptr, s32, grid, st);
}

}  // namespace woq
