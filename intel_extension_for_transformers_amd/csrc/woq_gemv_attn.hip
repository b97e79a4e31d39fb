// woq_gemv_attn.hip — [RMSNorm + qkv GEMV] and [RoPE + KV append + single-query attention] of a decode step in ONE
// launch (round 3). Reference path replaced: qbits.cpp:113-140 at M = 1 followed by stock HF eager attention on the CPU.
//
// Why. Inside the captured decode step every launch costs ~2 us before it does anything (tools/xq_probe.hip: an empty
// kernel on the same grid, hipGraph replay: 1.7-2.4 us), and the batch-1 attention launch is a latency chain of 32
// workgroups that costs 5.0 us per layer in place (profiles/r03g_skip_masks.txt) — most of it waiting for its own
// first requests (position, cos / sin row, K / V rows) after the boundary. The qkv -> attention edge is the one edge of
// the layer that is NOT all-to-all: head h needs only the 24 column strips that hold its q, k and v rows.
// How. The grid is the GEMV's strips plus one attention workgroup per head at its END. An attention workgroup issues
// everything that does not depend on this token's q / k / v at once (position, K / V rows of its first passes, cos /
// sin), then its wave 0 re-reads the head's 384 {tag, value} granules until every tag is this (step, layer)'s; a strip
// workgroup's epilogue writes its 16 outputs as such granules — one 8-byte write-through agent-scope store each, the
// data is its own flag (cdna_hip_programming.md Guideline 16, form R2: no fence, no flag that can overtake its data).
// Tags come from a device-side step counter (the step's first kernel advances it), so a replayed graph sees fresh
// tags; the wait is bounded (~20 ms, then a sticky status word) and needs no dispatch order: strips never wait, and
// 32 waiting workgroups cannot keep 768 strips off a chip that holds all of them at once.
// A first form without resident attention workgroups — the strip that completes a head runs its attention — was
// correct but slower than two launches (profiles/r03j_fused_attn_last_arriver_negative.txt).
// Scope: multi-head attention shapes (heads == kv_heads), head_dim 128, K = 32 tiles at 8 per wave (4 waves),
// one context slice, no sliding window; everything else keeps the two launches.
#include <algorithm>
#include <cstdlib>

#include "woq_attn_decode.h"
#include "woq_gemv_common.h"
#include "woq_gemv_xqs.h"
#include "woq_launch.h"
#include "woq_xq.h"

namespace woq {

struct FusedAttnArgs {
  const unsigned int* seq;  // device-side step counter (advanced by the step's first kernel): tag = seq << 6 | layer
  int layer;
  int* status;              // sticky: set when an attention workgroup gave up waiting
  void* kcache;
  void* vcache;
  const int32_t* pos;
  const float* cs;
  const float* sn;
  int heads, kv_heads, window, spw;
  float* attn_out;  // fp32 [heads * 128] attention output
  XqPtrs xq_attn;
  int ns;           // context slices per head (round 6): heads * ns attention workgroups behind the strips
  unsigned long long* part_g;  // ns > 1: the slices' partials as tagged granules (woq_attn_decode.h attn_part_granule)
};

#ifndef WOQ_XQS_DEPTH
#define WOQ_XQS_DEPTH 4
#endif
constexpr int FUSED_TPW = 8;
// Round 6: the q strips of the fused launch are given the WHOLE slice as their window (8 tiles per wave up front) and the
// k / v strips a shallower one, so that q lands while k / v are still streaming: the attention workgroups then run
// everything over the cache on q alone (woq_attn_decode.h) and only the new position's score and value stand behind
// the launch's last strips. A/B builds: -DWOQ_FUSED_DQ=4 -DWOQ_FUSED_DKV=4 is the round-5 order of arrival.
#ifndef WOQ_FUSED_DQ
#define WOQ_FUSED_DQ 8
#endif
#ifndef WOQ_FUSED_DKV
#define WOQ_FUSED_DKV 1
#endif
// the fused launch's 14th argument dword: tpg_shift | flags << 8 | strips << 16 — the strip workgroups come first in the
// grid, one attention workgroup per head behind them — so that the role test needs nothing but preloaded arguments
// (gridDim is a hidden argument: it lives in the argument segment too)
__device__ __forceinline__ int fa_strips_of_grid(int tpg_flags) { return (tpg_flags >> 16) & 0xffff; }

template <int SMODE, bool ASYM, bool S32, typename KV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void gemv_xqs_attn_kernel(
    const u32x4* __restrict__ q, const void* __restrict__ scales, const uint8_t* __restrict__ xlimbs,
    const float* __restrict__ xu, int tiles_k, int n_q_strips, int base_tiles, int rem_tiles, int n_groups, int tpg_flags,
    XqsLate late_in_the_argument_segment, unsigned long long* __restrict__ qkv_g, int N, FusedAttnArgs fa) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n_strips = fa_strips_of_grid(tpg_flags);
  if ((int)blockIdx.x >= n_strips) {  // the attention workgroup of head blockIdx.x - n_strips
    const unsigned int tag = (fa.seq[0] << 6) | (unsigned int)fa.layer;
    const int a = (int)blockIdx.x - n_strips;
    if (fa.ns > 1) {
      // Context slices inside the fused launch (round 6, long contexts): heads * ns attention workgroups, each the
      // per-head flash-decoding slice of woq_attn_decode.h on a granule source — its K / V rows (they depend on the
      // position only) stream beside the launch's weight tiles while q is still in the making; the slices of a head then
      // merge among themselves (tagged partial granules, every slice finalises a share of the head's XQ blocks): no
      // combine launch. Workgroup ids go round-robin over the 8 XCDs: the query heads that share a kv head AND a slice
      // read the same cache rows, so they get ids that differ by multiples of 8 (one L2 pulls the rows from HBM once).
      const int rep = fa.heads / fa.kv_heads, groups = fa.kv_heads * fa.ns;
      int h, slice;
      if ((groups & 7) == 0 && (n_strips & 7) == 0) {
        const int y = a >> 3, gi = (a & 7) + 8 * (y / rep);
        h = (gi / fa.ns) * rep + y % rep, slice = gi % fa.ns;
      } else {
        h = a / fa.ns, slice = a % fa.ns;
      }
      attn_decode_body<KV, 128, true>((float*)smem_raw, h, slice, fa.ns, AttnGranule{qkv_g, tag, fa.status, fa.part_g},
                                      (KV*)fa.kcache, (KV*)fa.vcache, fa.pos, fa.cs, fa.sn, fa.heads, fa.kv_heads,
                                      fa.window, fa.spw, fa.attn_out, fa.xq_attn);
      return;
    }
    attn_decode_body<KV, 128, false>((float*)smem_raw, a, 0, 1,
                                     AttnGranule{qkv_g, tag, fa.status}, (KV*)fa.kcache, (KV*)fa.vcache, fa.pos, fa.cs,
                                     fa.sn, fa.heads, fa.kv_heads, fa.window, fa.spw, fa.attn_out, fa.xq_attn);
    return;
  }
  // (the K range always starts at tile 0 here: the kt_off slot of the preloaded dwords carries the number of q strips)
  if ((int)blockIdx.x < n_q_strips)
    gemv_xqs_body<FUSED_TPW, 1, WOQ_FUSED_DQ, SMODE, ASYM, S32, true>(smem_raw, q, scales, xlimbs, xu, tiles_k, 0, base_tiles,
                                                                     rem_tiles, n_groups, tpg_flags & 0xff,
                                                                     (tpg_flags >> 8) & 0xff, 4, xqs_late_ptr());
  else
    gemv_xqs_body<FUSED_TPW, 1, WOQ_FUSED_DKV, SMODE, ASYM, S32, true>(smem_raw, q, scales, xlimbs, xu, tiles_k, 0, base_tiles,
                                                                      rem_tiles, n_groups, tpg_flags & 0xff,
                                                                      (tpg_flags >> 8) & 0xff, 4, xqs_late_ptr());
}

// positions one attention workgroup of the fused launch may have to hold scores for (launch_attn_t's sizing)
static int fused_attn_span(int max_ctx, int window, int splits) {
  const int reach = window > 0 ? std::min(window, max_ctx) : max_ctx;
  return splits > 1 ? ((((reach + splits - 1) / splits) + 63) & ~63) + 64 : reach;
}

// does the fused launch take this (blob, attention) combination?
bool gemv_xq_attn_supported(const woq_blob_header& h, int heads, int kv_heads, int head_dim, int kv_dtype, int max_ctx,
                            int window, int splits) {
  if (h.weight_type != WOQ_W_INT4_CLIP || h.off_shuffle != 0 || h.K != h.Kpad || h.N != h.Npad) return false;
  if (h.Kpad / WOQ_TILE_K != 4 * FUSED_TPW) return false;  // four waves of eight tiles
  if (h.scale_mode == 0 && h.n_groups > 1) {
    const int tpg = h.group / WOQ_TILE_K;
    if (tpg < 1 || (tpg & (tpg - 1)) != 0) return false;
  }
  // round 4: grouped-query shapes (Mistral-7B: 32 query / 8 kv heads) and a sliding window are taken as well — the
  // attention body always handled both (kh = h / rep, re-based cache pointers); round 3 simply had not tested them here
  if (kv_heads < 1 || heads % kv_heads != 0 || head_dim != 128 || splits > ATTN_A2A_MAX_SLICES) return false;
  if (splits > 1 && heads * splits > 512) return false;  // the slices of a head wait for each other: resident together
  if (splits > 1 && (heads + 2 * kv_heads) * 8 + heads * splits > 65535) return false;  // the packed strip count
  if (h.N != (heads + 2 * kv_heads) * head_dim) return false;
  (void)window;
  if (kv_dtype != WOQ_F16 && kv_dtype != WOQ_BF16 && kv_dtype != WOQ_FP8_E4M3) return false;
  // every workgroup of the launch gets max(attention LDS, GEMV LDS): the 768 strip workgroups (three per CU, four on
  // the CUs that also hold an attention workgroup) inherit the attention's score buffer, which grows with max_ctx.
  // Up to 38 KiB (max_ctx 8192) four workgroups still share a CU's 160 KiB; beyond that the strips would lose
  // occupancy to a buffer they never touch, so such engines keep the two launches.
  return attn_dec_lds_floats(128, fused_attn_span(max_ctx, window, splits)) * 4 <= 38 * 1024;
}

struct FusedLaunch {
  const void* q;
  const void* scales;
  const void* zp;
  XqPtrs xin;
  int tiles_k, K, N, n_groups, tpg_shift, flags, n_ssq;
  unsigned long long* out;
  float eps;
  const float* ssq_in;
  FusedAttnArgs fa;
  size_t lds;
};

template <int SMODE, bool ASYM, bool S32, typename KV>
static int launch_fused_t(const FusedLaunch& a, hipStream_t st) {
  auto kern = gemv_xqs_attn_kernel<SMODE, ASYM, S32, KV>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return woq::fail(std::string("QBits: hipFuncSetAttribute: ") + hipGetErrorString(e));
    attr_set = true;
  }
  XqsLate late;
  late.zp = (const uint8_t*)a.zp, late.xsx = a.xin.sx, late.out = (float*)a.out, late.bias = nullptr, late.residual = nullptr;
  late.ssq_in = a.ssq_in, late.next_norm_w = nullptr, late.ssq_out = nullptr, late.tp = nullptr;
  late.tag_seq = a.fa.seq, late.tag_layer = a.fa.layer, late.xo = XqPtrs{nullptr, nullptr, nullptr}, late.eps = a.eps;
  late.N = a.N, late.K = a.K, late.n_ssq = a.n_ssq, late.lut = LutArgs{};
  hipLaunchKernelGGL(kern, dim3(a.N / 16 + a.fa.heads * a.fa.ns), dim3(256), a.lds, st, (const u32x4*)a.q, a.scales,
                     a.xin.limbs, a.xin.u, a.tiles_k, a.fa.heads * 8, FUSED_TPW, 0, a.n_groups,
                     a.tpg_shift | (a.flags << 8) | ((a.N / 16) << 16), late, a.out, a.N, a.fa);
  return 0;
}

template <typename KV>
static int launch_fused_kv(const FusedLaunch& a, int smode, bool asym, bool s32, hipStream_t st) {
#define WOQ_FA_CASE(SM, AS, S3) \
  if (smode == SM && asym == AS && s32 == S3) return launch_fused_t<SM, AS, S3, KV>(a, st);
  WOQ_FA_CASE(0, false, false)
  WOQ_FA_CASE(0, false, true)
  WOQ_FA_CASE(0, true, false)
  WOQ_FA_CASE(0, true, true)
  WOQ_FA_CASE(1, false, false)
  WOQ_FA_CASE(1, false, true)
  WOQ_FA_CASE(1, true, false)
  WOQ_FA_CASE(1, true, true)
#undef WOQ_FA_CASE
  return woq::fail("QBits: bad fused qkv + attention configuration");
}

// qkv_g ({tag, fp32} granules [(heads + 2 kv_heads) * 128]) = xin . W_qkv_deq * rsqrt(mean(x^2) + eps); per head, as its granules
// arrive: RoPE, KV append at *pos, attention over the cache -> attn_out (+ its XQ form).
// splits > 1: `splits` context slices per head; they merge among themselves through the tagged granules `part_g`
// ([heads][64][130] x 8 B) and write attn_out / xq_attn — no combine launch.
int launch_gemv_xq_attn(const XqPtrs& xin, const void* blob, const woq_blob_header& h, unsigned long long* qkv_g,
                        const float* ssq_in, float eps, const unsigned int* seq, int layer, int* status, void* kcache,
                        void* vcache, int kv_dtype, const int32_t* pos, const float* cs, const float* sn, int heads,
                        int kv_heads, int max_ctx, int window, float* attn_out, const XqPtrs& xq_attn, hipStream_t st,
                        int splits, unsigned long long* part_g) {
  FusedLaunch a;
  if (splits < 1) splits = 1;
  const uint8_t* b = (const uint8_t*)blob;
  a.q = b + h.off_q;
  a.scales = b + h.off_scale;
  a.zp = h.off_zp ? b + h.off_zp : nullptr;
  a.xin = xin;
  a.K = h.K;
  a.N = h.N;
  a.tiles_k = h.Kpad / WOQ_TILE_K;
  a.n_groups = h.n_groups;
  a.tpg_shift = 0;
  if (h.scale_mode == 0 && h.n_groups > 1) {
    int tpg = h.group / WOQ_TILE_K;
    while (tpg > 1) {
      tpg >>= 1;
      ++a.tpg_shift;
    }
  }
  a.flags = h.scale_type == WOQ_BF16 ? 1 : 0;
  a.eps = eps;
  a.n_ssq = h.K / 16;
  a.ssq_in = ssq_in;
  a.out = qkv_g;
  const int smode = (int)h.scale_mode;
  const bool asym = a.zp != nullptr, s32 = h.scale_type == WOQ_F32;
  const int span = fused_attn_span(max_ctx, window, splits);
  const int spw = attn_dec_spw(span);
  if (splits > 1 && part_g == nullptr) return woq::fail("QBits: context slices in the fused launch need the partial granules");
  a.fa = FusedAttnArgs{seq, layer, status, kcache, vcache, pos, cs, sn, heads, kv_heads, window, spw, attn_out, xq_attn,
                       splits, part_g};
  const size_t lds_attn = attn_dec_lds_floats(128, span) * 4;
  size_t lds_gemv = 0;
  if (smode == 0)
    lds_gemv = asym ? (s32 ? XqsLds<FUSED_TPW, 1, 0, true, true>::total(4) : XqsLds<FUSED_TPW, 1, 0, true, false>::total(4))
                    : (s32 ? XqsLds<FUSED_TPW, 1, 0, false, true>::total(4) : XqsLds<FUSED_TPW, 1, 0, false, false>::total(4));
  else
    lds_gemv = asym ? (s32 ? XqsLds<FUSED_TPW, 1, 1, true, true>::total(4) : XqsLds<FUSED_TPW, 1, 1, true, false>::total(4))
                    : (s32 ? XqsLds<FUSED_TPW, 1, 1, false, true>::total(4) : XqsLds<FUSED_TPW, 1, 1, false, false>::total(4));
  a.lds = std::max(lds_attn, lds_gemv);
  if (a.lds > 160 * 1024) return woq::fail("QBits: fused qkv + attention launch does not fit LDS");
  if (kv_dtype == WOQ_F16) return launch_fused_kv<_Float16>(a, smode, asym, s32, st);
  if (kv_dtype == WOQ_FP8_E4M3) return launch_fused_kv<Fp8>(a, smode, asym, s32, st);
  return launch_fused_kv<__bf16>(a, smode, asym, s32, st);
}


}  // namespace woq
