// NOT PART OF THE PRODUCT (round 6): [combine of the context slices | o_proj strips] as ONE launch — built, parity-green
// (bit-identical to the combine launch + o_proj launch, tests of commit history: test_fused_launch_with_context_slices_*,
// test_grouped_slices_with_combine_led_o_proj_*) and measured SLOWER than the two launches it replaces:
//   profiles/r06s_fused_context_slices.txt — 7B @ 2048, 16 layers: +1.0 us per layer with one flag word per XQ block
//   (64 polled words per wave), +2.9 us with one flag word per head (8 polled words per wave, one more workgroup barrier
//   on the producer side); Mistral-7B @ 8k (grouped slices): +1.3 / +2.8 us per layer.
// The in-launch hand-off (write-through stores, drain, flag, poll, agent-scope loads) costs more than the 1.7-us kernel
// boundary it removes — the same finding as the chained layer of round 3 (tools/rejected/woq_gemv_chain.hip).
// This is the per-head-flag form; it compiled inside csrc/woq_gemv_attn.hip (namespace woq, same includes) with
// attn_combine_body<HD, PUBLISH> in woq_attn_merge.h, xq_emit16<2> (write-through stores without a flag) in woq_xq.h and
// XqsChain::blk_shift in woq_gemv_xqs.h; the engine hook was `comb` in engine_attn_block_xq (WOQ_FUSE_COMB).

// ---- [combine of the context slices | o_proj strips] in ONE launch (round 6) -------------------------------------
// Why. With context slices the attention ends in a combine launch: `heads` workgroups that read the slices' partials
// and write 4096 values — 5.1 us per layer, all of it the launch floor plus one dependent round trip
// (profiles/r04j_longctx_and_tp_kernel_stats.txt) — and o_proj, the next launch, cannot ask for a byte of its weights
// until it has ended. Here the combine workgroups come FIRST in o_proj's grid and publish each head's eight XQ blocks
// with their flag words (woq_xq.h XqPub: write-through stores, drain, flag); an o_proj strip requests its whole weight
// window at once and only then waits (bounded, CHAIN_IN of woq_gemv_xqs.h) for the blocks of each wave's K slice. The
// combine workgroups never wait and every workgroup of the launch (heads + N / 16 <= 1024) is resident at once.
struct CombArgs {
  const float* part;        // the slices' partials (woq_attn_merge.h)
  float* attn_out;          // fp32 [heads * 128]
  XqPtrs xq_attn;           // the o_proj input the combine workgroups write
  unsigned int* flag;       // [heads] one word per head (its eight XQ blocks), resting at older tags
  const unsigned int* seq;  // device-side step counter: tag = seq << 6 | layer
  int* status;
  int layer, ns, heads;
};

template <int SMODE, bool ASYM, bool S32>
__global__ __launch_bounds__(256) void gemv_xqs_comb_kernel(
    const u32x4* __restrict__ q, const void* __restrict__ scales, const uint8_t* __restrict__ xlimbs,
    const float* __restrict__ xu, int tiles_k, int n_comb, int base_tiles, int rem_tiles, int n_groups, int tpg_flags,
    XqsLate late_in_the_argument_segment, CombArgs ca) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const unsigned int tag = (ca.seq[0] << 6) | (unsigned int)ca.layer;
  if ((int)blockIdx.x < n_comb) {
    attn_combine_body<128, true>(ca.part, ca.heads, (int)blockIdx.x, ca.ns, ca.attn_out, ca.xq_attn,
                                 XqPub{ca.flag, tag}, (float*)smem_raw);
    return;
  }
  // (the K range always starts at tile 0: the kt_off slot of the preloaded dwords carries the number of combine workgroups)
  const XqsChain ch = {ca.flag, tag, XqPub{nullptr, 0u}, ca.status, (int)blockIdx.x - n_comb, 3};
  gemv_xqs_body<FUSED_TPW, 1, 8, SMODE, ASYM, S32, false, true>(smem_raw, q, scales, xlimbs, xu, tiles_k, 0, base_tiles,
                                                               rem_tiles, n_groups, tpg_flags & 0xff,
                                                               (tpg_flags >> 8) & 0xff, 4, xqs_late_ptr(), ch);
}

// does the [combine | o_proj] launch take this o_proj blob?
bool gemv_xq_comb_supported(const woq_blob_header& h, int heads, int head_dim) {
  if (h.weight_type != WOQ_W_INT4_CLIP || h.off_shuffle != 0 || h.K != h.Kpad || h.N != h.Npad) return false;
  if (h.Kpad / WOQ_TILE_K != 4 * FUSED_TPW || head_dim != 128 || h.K != heads * head_dim) return false;
  if (h.scale_mode == 0 && h.n_groups > 1) {
    const int tpg = h.group / WOQ_TILE_K;
    if (tpg < 1 || (tpg & (tpg - 1)) != 0) return false;
  }
  return heads + h.N / 16 <= 1024;
}

struct CombLaunch {
  const void* q;
  const void* scales;
  XqsLate late;
  int tiles_k, N, n_groups, tpg_shift, flags;
  CombArgs ca;
  size_t lds;
};

template <int SMODE, bool ASYM, bool S32>
static int launch_comb_t(const CombLaunch& a, hipStream_t st) {
  auto kern = gemv_xqs_comb_kernel<SMODE, ASYM, S32>;
  hipLaunchKernelGGL(kern, dim3(a.ca.heads + a.N / 16), dim3(256), a.lds, st, (const u32x4*)a.q, a.scales,
                     (const uint8_t*)a.ca.xq_attn.limbs, (const float*)a.ca.xq_attn.u, a.tiles_k, a.ca.heads, FUSED_TPW, 0,
                     a.n_groups, a.tpg_shift | (a.flags << 8) | (4 << 16), a.late, a.ca);
  return 0;
}

// hidden[N] += combine(part) . W_o; the new hidden leaves as the next GEMV's XQ vector (times next_norm_w) with its
// per-block sums of squares. part: `ns` slices per head; attn_out / xq_attn: the combined attention output (written by
// the combine workgroups, read by the strips of the same launch behind `flags`).
int launch_gemv_xq_comb(const void* blob, const woq_blob_header& h, const float* part, int ns, int heads, float* attn_out,
                        const XqPtrs& xq_attn, unsigned int* flags, const unsigned int* seq, int layer, int* status,
                        float* out, const float* residual, const XqPtrs& xo, const float* next_norm_w, float* ssq_out,
                        hipStream_t st) {
  CombLaunch a;
  const uint8_t* b = (const uint8_t*)blob;
  a.q = b + h.off_q;
  a.scales = b + h.off_scale;
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
  XqsLate& late = a.late;
  late.zp = h.off_zp ? b + h.off_zp : nullptr, late.xsx = xq_attn.sx, late.out = out, late.bias = nullptr;
  late.residual = residual, late.ssq_in = nullptr, late.next_norm_w = next_norm_w, late.ssq_out = ssq_out, late.tp = nullptr;
  late.tag_seq = nullptr, late.tag_layer = 0, late.xo = xo, late.eps = 0.f, late.N = h.N, late.K = h.K;
  late.n_ssq = h.K / 16, late.lut = LutArgs{};
  a.ca = CombArgs{part, attn_out, xq_attn, flags, seq, status, layer, ns, heads};
  const int smode = (int)h.scale_mode;
  const bool asym = late.zp != nullptr, s32 = h.scale_type == WOQ_F32;
  size_t lds_gemv = 0;
  if (smode == 0)
    lds_gemv = asym ? (s32 ? XqsLds<FUSED_TPW, 1, 0, true, true>::total(4) : XqsLds<FUSED_TPW, 1, 0, true, false>::total(4))
                    : (s32 ? XqsLds<FUSED_TPW, 1, 0, false, true>::total(4) : XqsLds<FUSED_TPW, 1, 0, false, false>::total(4));
  else
    lds_gemv = asym ? (s32 ? XqsLds<FUSED_TPW, 1, 1, true, true>::total(4) : XqsLds<FUSED_TPW, 1, 1, true, false>::total(4))
                    : (s32 ? XqsLds<FUSED_TPW, 1, 1, false, true>::total(4) : XqsLds<FUSED_TPW, 1, 1, false, false>::total(4));
  a.lds = std::max(lds_gemv, (size_t)448 * 4);
#define WOQ_CB_CASE(SM, AS, S3) \
  if (smode == SM && asym == AS && s32 == S3) return launch_comb_t<SM, AS, S3>(a, st);
  WOQ_CB_CASE(0, false, false)
  WOQ_CB_CASE(0, false, true)
  WOQ_CB_CASE(0, true, false)
  WOQ_CB_CASE(0, true, true)
  WOQ_CB_CASE(1, false, false)
  WOQ_CB_CASE(1, false, true)
  WOQ_CB_CASE(1, true, false)
  WOQ_CB_CASE(1, true, true)
#undef WOQ_CB_CASE
  return woq::fail("QBits: bad combine + o_proj configuration");
}

