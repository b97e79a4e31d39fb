// MOVED OUT OF THE PRODUCT (round 4): the decode layer as two chained launches — a built,
// parity-tested and measured NEGATIVE of round 3 (profiles/r03y_chained_layer_negative.txt: launch 1 about even,
// launch 2 +4.5..6.4 us per layer). It was an opt-in path of csrc/woq_engine.hip up to commit fa9f694 (engine hooks
// `chain_ok`, `engine_layer_chained`, ABI `woq_engine_set_chain`; its GPU test arm was the "chained" modes of
// tests/test_gpu_fullsize_oracle.py::test_in_launch_handoffs_equal_separate_launches in that commit: bit-identical
// repeats, greedy tokens equal to the separate launches, logits within 2e-5). Kept here as the record of the form
// that was tried; it still compiles against csrc/ headers (`hipcc -I intel_extension_for_transformers_amd/csrc -I
// include -c tools/rejected/woq_gemv_chain.hip`) but nothing links it. The second negative of the same idea, the
// persistent one-launch engine (csrc/woq_persist.hip), stays in the library because bench.py re-measures it every run.
// woq_gemv_chain.hip — the decode layer as TWO launches whose workgroups are chained inside the launch (round 3):
//     launch 1: [RMSNorm + qkv GEMV strips | one attention workgroup per head | o_proj strips]
//     launch 2: [RMSNorm + gate/up GEMV pairs (SiLU * mul) | down_proj strips]
// Reference path replaced: qbits.cpp:113-140 at M = 1 around stock HF attention / MLP glue on the CPU.
//
// Why. Inside the captured step every launch costs ~2 us before it does anything and another ~1 us until its first
// weight bytes arrive (profiles/r03e_graph_vs_eager.txt, r03n_bench.json roofline.ceiling); four launches per layer spend
// a third of the layer there. The grid of a chained launch lists its roles in DEPENDENCY ORDER: a workgroup of a later
// role requests its first weight tiles at once — they cross HBM while the earlier role is still running — and only
// then waits (bounded) for the activation blocks of its K slice, which the producing workgroups publish block by
// block (woq_xq.h XqPub: write-through stores, drain, flag). Nothing waits on a HIGHER block index, so a workgroup that
// is resident can always be served by workgroups dispatched before it; if the dispatcher ever broke that order, or a
// producer died, the wait gives up after ~20 ms and raises the engine's sticky status instead of hanging.
// The qkv -> attention edge uses tagged 8-byte granules (woq_gemv_attn.hip); attention -> o_proj and gate/up -> down use
// one flag word per 16-value XQ block.
// Scope: what woq_gemv_attn.hip covers (multi-head shapes, head_dim 128, hidden 4096, one context slice, no window),
// fp16 / bf16 scales, an intermediate size of at most 112 K tiles; everything else keeps separate launches.
#include <algorithm>
#include <cstdlib>

#include "woq_attn_decode.h"
#include "woq_gemv_common.h"
#include "woq_gemv_xqs.h"
#include "woq_launch.h"
#include "woq_xq.h"

namespace woq {

// one projection's operands (passed by value, one per role)
struct ChainLin {
  const u32x4* q;
  const void* scales;
  const uint8_t* zp;
  int tiles_k, n_groups, tpg_shift, N, K, flags;
};

struct ChainAttn {
  const unsigned int* seq;  // device-side step counter: tags = seq << 8 | layer << 2 | edge
  int layer;
  int* status;
  unsigned long long* qkv_g;  // {tag, fp32} granules of q | k | v
  void* kcache;
  void* vcache;
  const int32_t* pos;
  const float* cs;
  const float* sn;
  int heads, kv_heads, window, spw;
  float* attn_out;
  XqPtrs xq_attn;
  unsigned int* flag_attn;  // [heads * 128 / 16]
};

constexpr int CH_TPW = 8, CH_D = 4;  // launch 1: four waves of eight tiles (K = 4096)

template <int SMODE, bool ASYM, typename KV>
__global__ __launch_bounds__(256) void chain_attn_kernel(ChainLin lq, XqPtrs xq_hidden, const float* __restrict__ ssq_in,
                                                         int n_ssq, float eps, ChainAttn fa, ChainLin lo,
                                                         float* __restrict__ hidden, const float* __restrict__ ln2,
                                                         float* __restrict__ ssq_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const unsigned int tag0 = (fa.seq[0] << 8) | ((unsigned int)fa.layer << 2);
  const int nq = lq.N >> 4, bid = (int)blockIdx.x;
  const XqPtrs no_xq = {nullptr, nullptr, nullptr};
  if (bid < nq) {  // ---- a column strip of the qkv projection: its 16 outputs leave as tagged granules
    gemv_xqs_body<CH_TPW, 1, CH_D, SMODE, ASYM, false, true>(
        smem_raw, lq.q, lq.scales, xq_hidden.limbs, xq_hidden.u, lq.tiles_k, 0, CH_TPW, 0, lq.n_groups, lq.tpg_shift,
        lq.zp, xq_hidden.sx, (float*)fa.qkv_g, nullptr, nullptr, eps, lq.N, lq.K, lq.flags, ssq_in, n_ssq, no_xq, nullptr,
        nullptr, tag0);
    return;
  }
  if (bid < nq + fa.heads) {  // ---- the attention of one head: publishes its 8 blocks of the o_proj input
    attn_decode_body<KV, 128, false>((float*)smem_raw, bid - nq, 0, 1, AttnGranule{fa.qkv_g, tag0, fa.status},
                                     (KV*)fa.kcache, (KV*)fa.vcache, fa.pos, fa.cs, fa.sn, fa.heads, fa.kv_heads,
                                     fa.window, fa.spw, fa.attn_out, fa.xq_attn, XqPub{fa.flag_attn, tag0 | 1u});
    return;
  }
  // ---- a column strip of o_proj: weights requested first, then the attention blocks of each wave's K slice
  const XqsChain ch = {fa.flag_attn, tag0 | 1u, XqPub{nullptr, 0u}, fa.status, bid - nq - fa.heads};
  gemv_xqs_body<CH_TPW, 1, CH_D, SMODE, ASYM, false, false, true>(
      smem_raw, lo.q, lo.scales, fa.xq_attn.limbs, fa.xq_attn.u, lo.tiles_k, 0, CH_TPW, 0, lo.n_groups, lo.tpg_shift,
      lo.zp, fa.xq_attn.sx, hidden, nullptr, hidden, eps, lo.N, lo.K, lo.flags, nullptr, 0, xq_hidden, ln2, ssq_out, 0u,
      nullptr, ch);
}

// launch 2: four waves per workgroup like launch 1 — gate/up pairs at 8 tiles per wave, down_proj strips at up to TPWD
// tiles per wave — so that the WHOLE grid is resident from the start (4 workgroups per CU at <= 128 VGPRs and <= 40 KiB
// of LDS: 1024 slots for 688 + 256 workgroups): the strips land one per CU behind the pairs (round robin) with their
// first six weight tiles in flight. A first form with eight-wave workgroups (three per CU) was 6 us per layer SLOWER
// than separate launches: the strips that only became resident at the end piled up on the CUs that freed first; with
// the strips dispatched first they held a third of every CU for the whole gate/up phase, +10 us (profiles/r03v).
template <int SMODE, bool ASYM, int TPWD>
__global__ __launch_bounds__(256) void chain_mlp_kernel(ChainLin lg, XqPtrs xq_hidden, const float* __restrict__ ssq_in,
                                                        int n_ssq, float eps, const unsigned int* __restrict__ seq,
                                                        int layer, int* status, XqPtrs xq_act,
                                                        unsigned int* __restrict__ flag_act, ChainLin ld,
                                                        float* __restrict__ hidden, XqPtrs xq_next,
                                                        const float* __restrict__ ln_next,
                                                        float* __restrict__ ssq_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const unsigned int tag = (seq[0] << 8) | ((unsigned int)layer << 2) | 2u;
  const int np = lg.N >> 5, bid = (int)blockIdx.x;  // gate/up pairs: 32 columns of the fused matrix each
  if (bid < np) {  // ---- a gate/up column pair: SiLU(gate) * up leaves as one published block of the down_proj input
    const XqsChain ch = {nullptr, 0u, XqPub{flag_act, tag}, status, -1};
    gemv_xqs_body<8, 2, 4, SMODE, ASYM, false, false, false>(
        smem_raw, lg.q, lg.scales, xq_hidden.limbs, xq_hidden.u, lg.tiles_k, 0, lg.tiles_k >> 2, 0, lg.n_groups,
        lg.tpg_shift, lg.zp, xq_hidden.sx, nullptr, nullptr, nullptr, eps, lg.N, lg.K, lg.flags, ssq_in, n_ssq, xq_act,
        nullptr, nullptr, 0u, nullptr, ch);
    return;
  }
  // ---- a column strip of down_proj
  const XqsChain ch = {flag_act, tag, XqPub{nullptr, 0u}, status, bid - np};
  gemv_xqs_body<TPWD, 1, 6, SMODE, ASYM, false, false, true>(
      smem_raw, ld.q, ld.scales, xq_act.limbs, xq_act.u, ld.tiles_k, 0, ld.tiles_k >> 2, ld.tiles_k & 3, ld.n_groups,
      ld.tpg_shift, ld.zp, xq_act.sx, hidden, nullptr, hidden, eps, ld.N, ld.K, ld.flags, nullptr, 0, xq_next, ln_next,
      ssq_out, 0u, nullptr, ch);
}

static ChainLin chain_lin(const void* blob, const woq_blob_header& h, int epi) {
  const uint8_t* b = (const uint8_t*)blob;
  ChainLin l;
  l.q = (const u32x4*)(b + h.off_q);
  l.scales = b + h.off_scale;
  l.zp = h.off_zp ? b + h.off_zp : nullptr;
  l.tiles_k = h.Kpad / WOQ_TILE_K;
  l.n_groups = h.n_groups;
  l.tpg_shift = 0;
  if (h.scale_mode == 0 && h.n_groups > 1) {
    int tpg = h.group / WOQ_TILE_K;
    while (tpg > 1) {
      tpg >>= 1;
      ++l.tpg_shift;
    }
  }
  l.N = h.N;
  l.K = h.K;
  l.flags = (h.scale_type == WOQ_BF16 ? 1 : 0) | (epi == 1 ? 2 : 0);
  return l;
}

static bool chain_blob_ok(const woq_blob_header& h) {
  if (h.weight_type != WOQ_W_INT4_CLIP || h.off_shuffle != 0 || h.K != h.Kpad || h.N != h.Npad) return false;
  if (h.scale_type == WOQ_F32) return false;
  if (h.scale_mode == 0 && h.n_groups > 1) {
    const int tpg = h.group / WOQ_TILE_K;
    if (tpg < 1 || (tpg & (tpg - 1)) != 0) return false;
  }
  return true;
}
static bool same_kind(const woq_blob_header& a, const woq_blob_header& b) {
  return a.scale_mode == b.scale_mode && (a.off_zp != 0) == (b.off_zp != 0) && a.scale_type == b.scale_type;
}

bool gemv_xq_attn_supported(const woq_blob_header& h, int heads, int kv_heads, int head_dim, int kv_dtype, int max_ctx,
                            int window, int splits);

// launch 2 needs its whole grid resident (4 workgroups per CU): the asymmetric variants of its kernel need more than
// 128 VGPRs, so they keep the two separate launches (launch 1 is chained all the same)
bool chain_mlp_supported(const woq_blob_header& hg, const woq_blob_header& hd) {
  const int td = hd.Kpad / WOQ_TILE_K;
  return hg.off_zp == 0 && hd.off_zp == 0 && td >= 4 && td <= 88;
}

// do the chained launches take this layer?
bool chain_layer_supported(const woq_blob_header& hq, const woq_blob_header& ho, const woq_blob_header& hg,
                           const woq_blob_header& hd, int heads, int kv_heads, int head_dim, int kv_dtype, int max_ctx,
                           int window, int splits) {
  if (!gemv_xq_attn_supported(hq, heads, kv_heads, head_dim, kv_dtype, max_ctx, window, splits)) return false;
  for (const woq_blob_header* h : {&hq, &ho, &hg, &hd})
    if (!chain_blob_ok(*h)) return false;
  if (!same_kind(hq, ho) || !same_kind(hg, hd)) return false;
  if (ho.Kpad / WOQ_TILE_K != 4 * CH_TPW || (ho.N & 15) != 0) return false;      // four waves of eight tiles
  if (hg.Kpad / WOQ_TILE_K != 32 || ((hg.N >> 4) & 1) != 0) return false;        // four waves of eight tiles, whole pairs
  const int td = hd.Kpad / WOQ_TILE_K;
  if (td < 4 || hd.K != (hg.N >> 1) || (hd.N & 15) != 0) return false;
  return true;
}

template <int SMODE, bool ASYM, typename KV>
static int launch_chain_attn_t(dim3 grid, size_t lds, hipStream_t st, const ChainLin& lq, const XqPtrs& xh,
                               const float* ssq_in, int n_ssq, float eps, const ChainAttn& fa, const ChainLin& lo,
                               float* hidden, const float* ln2, float* ssq_out) {
  auto kern = chain_attn_kernel<SMODE, ASYM, KV>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return woq::fail(std::string("QBits: hipFuncSetAttribute: ") + hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, lq, xh, ssq_in, n_ssq, eps, fa, lo, hidden, ln2, ssq_out);
  return 0;
}

template <int SMODE, bool ASYM, int TPWD>
static int launch_chain_mlp_t(dim3 grid, size_t lds, hipStream_t st, const ChainLin& lg, const XqPtrs& xh,
                              const float* ssq_in, int n_ssq, float eps, const unsigned int* seq, int layer, int* status,
                              const XqPtrs& xq_act, unsigned int* flag_act, const ChainLin& ld, float* hidden,
                              const XqPtrs& xq_next, const float* ln_next, float* ssq_out) {
  auto kern = chain_mlp_kernel<SMODE, ASYM, TPWD>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return woq::fail(std::string("QBits: hipFuncSetAttribute: ") + hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, lg, xh, ssq_in, n_ssq, eps, seq, layer, status, xq_act, flag_act, ld,
                     hidden, xq_next, ln_next, ssq_out);
  return 0;
}

template <int SMODE, bool ASYM>
static size_t lds_gemv(int tpw, int cb, int nw) {
  if (cb == 2) return XqsLds<8, 2, SMODE, ASYM, false>::total(nw);
  if (tpw <= 8) return XqsLds<8, 1, SMODE, ASYM, false>::total(nw);
  if (tpw <= 16) return XqsLds<16, 1, SMODE, ASYM, false>::total(nw);
  return XqsLds<22, 1, SMODE, ASYM, false>::total(nw);
}

// launch 1 of a layer: qkv strips, attention, o_proj strips (hidden += attn . W_o; the new hidden also as xq_hidden * ln2)
int launch_chain_attn(const XqPtrs& xq_hidden, const float* ssq_in, float eps, const void* qkv_blob,
                      const woq_blob_header& hq, const void* o_blob, const woq_blob_header& ho, const unsigned int* seq,
                      int layer, int* status, unsigned long long* qkv_g, void* kcache, void* vcache, int kv_dtype,
                      const int32_t* pos, const float* cs, const float* sn, int heads, int kv_heads, int max_ctx,
                      int window, float* attn_out, const XqPtrs& xq_attn, unsigned int* flag_attn, float* hidden,
                      const float* ln2, float* ssq_out, hipStream_t st) {
  const ChainLin lq = chain_lin(qkv_blob, hq, 0), lo = chain_lin(o_blob, ho, 0);
  const ChainAttn fa = {seq, layer, status, qkv_g, kcache, vcache, pos, cs, sn, heads, kv_heads, window,
                        attn_dec_spw(max_ctx), attn_out, xq_attn, flag_attn};
  const dim3 grid((unsigned)((hq.N >> 4) + heads + (ho.N >> 4)));
  const int sm = (int)hq.scale_mode;
  const bool asym = hq.off_zp != 0;
  const size_t lds_a = attn_dec_lds_floats(128, max_ctx) * 4;
  size_t lds_g = 0;
#define WOQ_CH_LDS(SM, AS) \
  if (sm == SM && asym == AS) lds_g = lds_gemv<SM, AS>(8, 1, 4);
  WOQ_CH_LDS(0, false) WOQ_CH_LDS(0, true) WOQ_CH_LDS(1, false) WOQ_CH_LDS(1, true)
#undef WOQ_CH_LDS
  const size_t lds = std::max(lds_a, lds_g);
  const int n_ssq = hq.K / 16;
#define WOQ_CH_CASE(SM, AS, KVT)                                                                                   \
  if (sm == SM && asym == AS)                                                                                       \
    return launch_chain_attn_t<SM, AS, KVT>(grid, lds, st, lq, xq_hidden, ssq_in, n_ssq, eps, fa, lo, hidden, ln2, \
                                            ssq_out);
  if (kv_dtype == WOQ_F16) {
    WOQ_CH_CASE(0, false, _Float16) WOQ_CH_CASE(0, true, _Float16) WOQ_CH_CASE(1, false, _Float16)
    WOQ_CH_CASE(1, true, _Float16)
  } else if (kv_dtype == WOQ_FP8_E4M3) {
    WOQ_CH_CASE(0, false, Fp8) WOQ_CH_CASE(0, true, Fp8) WOQ_CH_CASE(1, false, Fp8) WOQ_CH_CASE(1, true, Fp8)
  } else {
    WOQ_CH_CASE(0, false, __bf16) WOQ_CH_CASE(0, true, __bf16) WOQ_CH_CASE(1, false, __bf16)
    WOQ_CH_CASE(1, true, __bf16)
  }
#undef WOQ_CH_CASE
  return woq::fail("QBits: bad chained attention launch configuration");
}

// launch 2 of a layer: gate/up pairs, down_proj strips (hidden += act . W_down; the new hidden also as xq_next * ln_next)
int launch_chain_mlp(const XqPtrs& xq_hidden, const float* ssq_in, float eps, const void* gu_blob,
                     const woq_blob_header& hg, const void* down_blob, const woq_blob_header& hd,
                     const unsigned int* seq, int layer, int* status, const XqPtrs& xq_act, unsigned int* flag_act,
                     float* hidden, const XqPtrs& xq_next, const float* ln_next, float* ssq_out, hipStream_t st) {
  const ChainLin lg = chain_lin(gu_blob, hg, 1), ld = chain_lin(down_blob, hd, 0);
  const dim3 grid((unsigned)((hg.N >> 5) + (hd.N >> 4)));
  const int sm = (int)hg.scale_mode;
  const bool asym = hg.off_zp != 0;
  const int td = hd.Kpad / WOQ_TILE_K;
  const int tpwd = (td + 3) / 4;  // tiles per wave of a down_proj strip (four waves)
  size_t lds = 0;
#define WOQ_CH_LDS(SM, AS) \
  if (sm == SM && asym == AS) lds = std::max(lds_gemv<SM, AS>(8, 2, 4), lds_gemv<SM, AS>(tpwd, 1, 4));
  WOQ_CH_LDS(0, false) WOQ_CH_LDS(0, true) WOQ_CH_LDS(1, false) WOQ_CH_LDS(1, true)
#undef WOQ_CH_LDS
  const int n_ssq = hg.K / 16;
#define WOQ_CH_CASE(SM, AS)                                                                                          \
  if (sm == SM && asym == AS) {                                                                                       \
    if (tpwd <= 8)                                                                                                    \
      return launch_chain_mlp_t<SM, AS, 8>(grid, lds, st, lg, xq_hidden, ssq_in, n_ssq, eps, seq, layer, status, xq_act, \
                                           flag_act, ld, hidden, xq_next, ln_next, ssq_out);                         \
    if (tpwd <= 16)                                                                                                   \
      return launch_chain_mlp_t<SM, AS, 16>(grid, lds, st, lg, xq_hidden, ssq_in, n_ssq, eps, seq, layer, status,     \
                                            xq_act, flag_act, ld, hidden, xq_next, ln_next, ssq_out);                \
    return launch_chain_mlp_t<SM, AS, 22>(grid, lds, st, lg, xq_hidden, ssq_in, n_ssq, eps, seq, layer, status, xq_act, \
                                          flag_act, ld, hidden, xq_next, ln_next, ssq_out);                          \
  }
  WOQ_CH_CASE(0, false) WOQ_CH_CASE(0, true) WOQ_CH_CASE(1, false) WOQ_CH_CASE(1, true)
#undef WOQ_CH_CASE
  return woq::fail("QBits: bad chained MLP launch configuration");
}

}  // namespace woq
