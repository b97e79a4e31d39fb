// xq_r02_twin.h — the round-2 XQ decode GEMV (gemv_xq_kernel: every tile of a wave requested up front, offset-binary
// limbs with a ones row, per-tile scale requests, cross-lane shuffles for the block factors), frozen here as the
// TIMING twin of tools/xq_probe.hip. Not part of the product; its arithmetic expects the round-2 limb format, so only
// its launch time means anything (the probe feeds random bytes).
#pragma once
#include "../intel_extension_for_transformers_amd/csrc/woq_gemv_common.h"
#include "../intel_extension_for_transformers_amd/csrc/woq_xq.h"

namespace woq {
namespace r02 {

// LDS: [nw zero blocks of 256][nw ones blocks of 256][nw strips of TPW * 384][slab nw x CB x 16 f32][64 f32 scratch]
__host__ __device__ inline size_t xq_lds_bytes(int nw, int TPW, int CB) {
  return (size_t)nw * 512 + (size_t)nw * TPW * 384 + (size_t)nw * CB * 16 * 4 + 256;
}

// v of lane ^ 32 (valid in lanes 0..31) and of lane ^ 16 (valid in lanes 0..15), by gfx950's lane-swap VALU ops
// instead of two ds_bpermute round trips: v_permlane32_swap exchanges the upper half of its first operand with the
// lower half of the second, v_permlane16_swap the odd 16-lane rows of the first with the even rows of the second
__device__ __forceinline__ float quarter_swap32(float v) {
  const uint32_t b = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
  return __uint_as_float(r[1]);
}
__device__ __forceinline__ float quarter_swap16(float v) {
  const uint32_t b = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
  return __uint_as_float(r[1]);
}

// flags: bit 0 scales are bf16 (else fp16; ignored for fp32 scales), bit 1 SiLU(gate)*up epilogue (CB == 2)
template <int TPW, int CB, int SMODE, bool ASYM, bool S32>
__global__ __launch_bounds__(CB * TPW > 8 ? 512 : 1024) void gemv_xq_kernel(
    const u32x4* __restrict__ q, const void* __restrict__ scales, const uint8_t* __restrict__ xlimbs,
    const float* __restrict__ xu, int tiles_k, int K, int base_tiles, int rem_tiles, int n_groups, int tpg_shift,
    const uint8_t* __restrict__ zp, const float* __restrict__ xsx, float* __restrict__ out,
    const float* __restrict__ bias, const float* residual, float eps, int N, int flags, int kt_off,
    const float* __restrict__ ssq_in, int n_ssq, XqPtrs xo, const float* __restrict__ next_norm_w,
    float* __restrict__ ssq_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int SB = TPW * 384;  // strip bytes
  constexpr int ESZ = S32 ? 4 : 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = (int)blockDim.x >> 6;
  unsigned char* zero_blk = smem_raw + wid * 256;
  unsigned char* ones_blk = smem_raw + nw * 256 + wid * 256;
  unsigned char* strip = smem_raw + nw * 512 + (size_t)wid * SB;
  float* slab = (float*)(smem_raw + nw * 512 + (size_t)nw * SB);  // [nw][CB][16]
  float* red = slab + nw * CB * 16;                                // [64]
  const int kt0 = kt_off + wid * base_tiles + min(wid, rem_tiles);
  const int cnt = base_tiles + (wid < rem_tiles ? 1 : 0);
  const int i16 = lane & 15, kq = lane >> 4;
  const bool bf = (flags & 1) != 0, silu = (flags & 2) != 0;
  const int v16 = lane * 16;

  // the thread's residual element, fetched up front through a descriptor that is empty when there is none
  float e_res;
  {
    const int n0 = silu ? (int)blockIdx.x * 16 : (int)blockIdx.x * CB * 16;
    const int nlim = silu ? (N >> 1) : N;
    const rsrc_t rr = make_rsrc(residual ? residual + n0 : (const float*)xu,
                                residual ? max(0, min(nlim - n0, CB * 16)) * 4 : 0);
    e_res = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, min(tid, 63) * 4, 0, 0));
  }
  // ---- 0. this wave's K slice of the limb blocks, first in the wave's load queue (it returns first) ----
  // Plain loads + ds_write_b128 rather than LDS-DMA: hipcc turns every wait after a `global_load_lds` into vmcnt(0)
  // and does not order the strip's first ds_read behind it — the counted waits on the weight tiles below matter more
  // than three register hops.
  constexpr int XP = (SB + 1023) / 1024;  // 1-KiB pieces per strip
  u32x4 xl[XP];
  {
    const rsrc_t rl = make_rsrc(xlimbs + (size_t)kt0 * 384, max(0, min(cnt, tiles_k - kt0)) * 384);
#pragma unroll
    for (int j = 0; j < XP; ++j) xl[j] = __builtin_amdgcn_raw_buffer_load_b128(rl, v16 + j * 1024, 0, 0);
  }
  // block factors of the slice: lane L holds block kt0 * 8 + L (8 blocks per tile -> 64 = TPW 8; reads past the
  // slice return 0 through the descriptor, so tiles past the slice end contribute exactly 0)
  const int nblk = max(0, min(cnt * 8, tiles_k * 8 - kt0 * 8));
  const rsrc_t ru = make_rsrc(xu + (size_t)kt0 * 8, nblk * 4);
  const float uw = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ru, lane * 4, 0, 0));
  float sxw = 0.f;
  if constexpr (ASYM) {
    const rsrc_t rsx = make_rsrc(xsx + (size_t)kt0 * 8, nblk * 4);
    sxw = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsx, lane * 4, 0, 0));
  }
  // RMSNorm partials (wave 0 only; it also runs the epilogue): up to 1024 of them in four bounded 16-B loads, summed
  // after the inner products so that nothing here waits
  float4_t ssq_v[4];
  if (ssq_in != nullptr && wid == 0) {
    const rsrc_t rs = make_rsrc(ssq_in, n_ssq * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      ssq_v[j] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, v16 + j * 1024, 0, 0));
  }

  // ---- 1. scales / zero points, then all weight tiles ----
  u32x4 w[CB][TPW];
  typename RawSc<SMODE, S32>::type rsc[CB][TPW];
  uint32_t rzp[CB][TPW];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    const int tn = (int)blockIdx.x * CB + cb;
    if constexpr (SMODE == 0) {
      const rsrc_t rs = make_rsrc((const char*)scales + (size_t)tn * n_groups * 16 * ESZ, n_groups * 16 * ESZ);
      const rsrc_t rz = make_rsrc(ASYM ? zp + (size_t)tn * n_groups * 16 : (const uint8_t*)scales,
                                  ASYM ? n_groups * 16 : 0);
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int grp = min((kt0 + t) >> tpg_shift, n_groups - 1);
        if constexpr (S32)
          rsc[cb][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, i16 * 4, grp * 64, 0));
        else
          rsc[cb][t] = __builtin_amdgcn_raw_buffer_load_b16(rs, i16 * 2, grp * 32, 0);
        if constexpr (ASYM) rzp[cb][t] = __builtin_amdgcn_raw_buffer_load_b8(rz, i16, grp * 16, 0);
      }
    } else {
      const rsrc_t rs = make_rsrc((const char*)scales + (size_t)tn * tiles_k * 64 * ESZ, tiles_k * 64 * ESZ);
      const rsrc_t rz = make_rsrc(ASYM ? zp + (size_t)tn * tiles_k * 64 : (const uint8_t*)scales,
                                  ASYM ? tiles_k * 64 : 0);
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        if constexpr (S32)
          rsc[cb][t] = __builtin_bit_cast(
              float4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, i16 * 16 + t * 256, kt0 * 256, 0));
        else
          rsc[cb][t] =
              __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs, i16 * 8 + t * 128, kt0 * 128, 0));
        if constexpr (ASYM) rzp[cb][t] = __builtin_amdgcn_raw_buffer_load_b32(rz, i16 * 4 + t * 64, kt0 * 64, 0);
      }
    }
  }
  rsrc_t rq[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
    rq[cb] = make_rsrc(q + (size_t)((int)blockIdx.x * CB + cb) * tiles_k * 64, min(kt0 + cnt, tiles_k) * 1024);
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
      w[cb][t] = __builtin_amdgcn_raw_buffer_load_b128(rq[cb], v16 + t * 1024, kt0 * 1024, AUX_NT);

  // ---- 2. A-operand addresses: row r = lane & 15 -> quarter e = r >> 2, part r & 3 (limb 0..2 | ones) ----
  ((uint32_t*)zero_blk)[lane] = 0u;
  ((uint32_t*)ones_blk)[lane] = 0x01010101u;
#pragma unroll
  for (int j = 0; j < XP; ++j)
    if (v16 + j * 1024 < SB) *(u32x4*)(strip + v16 + j * 1024) = xl[j];  // past the slice: zeros (descriptor bounds)
  __builtin_amdgcn_wave_barrier();  // wave-private blocks and strip, in-order LDS: no workgroup barrier
  const int a_e = i16 >> 2, a_part = i16 & 3;
  const bool a_live = a_e == kq;
  const unsigned char* a_base =
      !a_live ? zero_blk + kq * 16 : (a_part == 3 ? ones_blk + kq * 16 : strip + kq * 48 + a_part * 16);
  const int a_step_t = (a_live && a_part != 3) ? 384 : 0, a_step_h = (a_live && a_part != 3) ? 192 : 0;
  const i32x4 izero = {0, 0, 0, 0};
  float tot[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) tot[cb] = 0.f;

  // this lane quarter's block factors (and block sums) of every tile, fetched across lanes ONCE: left next to
  // their uses, each tile would start with two LDS-crossbar round trips on the wave's critical path
  float ub[TPW][2], sb[TPW][2];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      ub[t][h2] = __shfl(uw, t * 8 + h2 * 4 + kq, 64);
      sb[t][h2] = ASYM ? __shfl(sxw, t * 8 + h2 * 4 + kq, 64) : 0.f;
    }
  __builtin_amdgcn_sched_barrier(0);

  // ---- 3. inner products, tiles in arrival order ----
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const i32x4 a0 = *(const i32x4*)(a_base + t * a_step_t);
    const i32x4 a1 = *(const i32x4*)(a_base + t * a_step_t + a_step_h);
    const float u0 = ub[t][0], u1 = ub[t][1];
    const float s0 = sb[t][0], s1 = sb[t][1];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const u32x4 wv = w[cb][t];
      const i32x4 b0 = {(int)((wv.x << 4) & 0xf0f0f0f0u), (int)(wv.x & 0xf0f0f0f0u), (int)((wv.y << 4) & 0xf0f0f0f0u),
                        (int)(wv.y & 0xf0f0f0f0u)};
      const i32x4 b1 = {(int)((wv.z << 4) & 0xf0f0f0f0u), (int)(wv.z & 0xf0f0f0f0u), (int)((wv.w << 4) & 0xf0f0f0f0u),
                        (int)(wv.w & 0xf0f0f0f0u)};
      const i32x4 d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, izero, 0, 0, 0);
      const i32x4 d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, izero, 0, 0, 0);
      float f0 = limb_combine(d0), f1 = limb_combine(d1);
      if constexpr (SMODE == 0) {
        if constexpr (ASYM) {  // the weights carry 16 * q: the zero point enters as 16 * zp
          const float z16 = -16.f * (float)((int)(rzp[cb][t] & 0xff) - 8);
          f0 = fmaf(z16, s0, f0);
          f1 = fmaf(z16, s1, f1);
        }
        float sc;
        if constexpr (S32)
          sc = rsc[cb][t];
        else
          sc = tscale16(rsc[cb][t], bf);
        tot[cb] = fmaf(sc, fmaf(f0, u0, f1 * u1), tot[cb]);
      } else {  // this lane quarter's 32-k group of each half: s = 2 h + (kq >> 1)
        if constexpr (ASYM) {
          const uint32_t z = rzp[cb][t] >> (8 * (kq >> 1));
          f0 = fmaf(-16.f * (float)((int)(z & 0xffu) - 8), s0, f0);
          f1 = fmaf(-16.f * (float)((int)((z >> 16) & 0xffu) - 8), s1, f1);
        }
        float sc0, sc1;
        if constexpr (S32) {
          sc0 = (kq >> 1) ? rsc[cb][t].y : rsc[cb][t].x;
          sc1 = (kq >> 1) ? rsc[cb][t].w : rsc[cb][t].z;
        } else {
          sc0 = tscale16((kq >> 1) ? (rsc[cb][t].x >> 16) : (rsc[cb][t].x & 0xffffu), bf);
          sc1 = tscale16((kq >> 1) ? (rsc[cb][t].y >> 16) : (rsc[cb][t].y & 0xffffu), bf);
        }
        tot[cb] = fmaf(sc0 * u0, f0, fmaf(sc1 * u1, f1, tot[cb]));
      }
    }
  }
  // the four lane quarters hold the four blocks' shares of each column
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    float v = tot[cb];
    v += quarter_swap32(v);  // lanes 0..31: this lane + lane ^ 32
    v += quarter_swap16(v);  // lanes 0..15: + lane ^ 16
    if (lane < 16) slab[((size_t)wid * CB + cb) * 16 + lane] = v;
  }
  if (ssq_in != nullptr && wid == 0) {
    float4_t t4 = (ssq_v[0] + ssq_v[1]) + (ssq_v[2] + ssq_v[3]);
    const float s = wave_sum_dpp((t4.x + t4.y) + (t4.z + t4.w));
    if (lane == 0) red[0] = s;
  }
  __syncthreads();

  // ---- 4. finish (lanes 0..15 of wave 0): sum over waves, RMSNorm factor, bias, SiLU*mul, residual, store, XQ ----
  if (tid < 16) {
    float v = 0.f, up = 0.f;
#pragma unroll 4
    for (int w2 = 0; w2 < nw; ++w2) {
      v += slab[((size_t)w2 * CB) * 16 + tid];
      if constexpr (CB == 2) up += slab[((size_t)w2 * CB + 1) * 16 + tid];
    }
    const float inv = ssq_in != nullptr ? 1.0f / sqrtf(red[0] / (float)K + eps) : 1.f;  // HF LlamaRMSNorm
    v *= inv;
    const int n = (int)blockIdx.x * 16 + tid;  // CB == 1 or the SiLU pair: one 16-column output tile per workgroup
    if (silu) {
      up *= inv;
      if (bias) {
        v += bias[min(((int)blockIdx.x * 2) * 16 + tid, N - 1)];
        up += bias[min(((int)blockIdx.x * 2 + 1) * 16 + tid, N - 1)];
      }
      v = v / (1.0f + __expf(-v)) * up;
    } else if (bias) {
      v += bias[min(n, N - 1)];
    }
    const bool live = n < (silu ? (N >> 1) : N);
    v = live ? v + e_res : 0.f;
    if (live && out) out[n] = v;
    if (xo.limbs != nullptr) {  // this tile IS block blockIdx.x of the next kernel's activation vector
      if (ssq_out != nullptr) {
        const float ss = row16_sum(v * v);
        if (tid == 0) ssq_out[blockIdx.x] = ss;
      }
      const float g = next_norm_w != nullptr ? next_norm_w[min(n, (silu ? (N >> 1) : N) - 1)] : 1.f;
      xq_emit16(v * g, xo, (int)blockIdx.x, tid);
    }
  }
}


}  // namespace r02
}  // namespace woq
