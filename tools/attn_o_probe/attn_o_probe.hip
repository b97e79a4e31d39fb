// woq_attn_o.hip — decode attention and the o_proj GEMV in ONE launch.
//
// Per layer the decode step had [RoPE + KV append + attention] -> boundary -> [o_proj GEMV + residual]. The attention
// launch is 32 workgroups on a 256-CU chip for ~5 us (one dependent chain per head: position -> q/k/v -> scores ->
// softmax -> P.V), during which no weight byte moves; the o_proj launch behind it then pays its own ramp for 8.65 MB.
// Here both run in one grid of heads + N / 16 workgroups of 256 threads:
//   * workgroups [0, heads): one attention head each (woq_attn_body.h, unchanged arithmetic). The head's output leaves
//     as XQ limb blocks written with agent-scope (sc1, write-through) stores; every storing wave drains its stores
//     (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, one thread adds 1 to the arrival word;
//   * workgroups [heads, heads + N / 16): one 16-column strip of o_proj each (woq_gemv_xq_body.h with WAIT): scales and
//     ALL weight tiles are requested at once, so the 8.65 MB stream runs under the attention; wave 0 then polls the
//     arrival word (relaxed agent-scope load + s_sleep, bounded) until all heads have arrived, and only then are the
//     limb blocks read — with sc1 loads, the pairing MI355X_MICROARCH.md lists as valid for sc1 producers (no L1 / stale
//     L2 copy is consulted) — inner products, residual add, and the new hidden state leaves as the MLP's XQ input;
//   * the LAST o_proj workgroup to finish (ticket) zeroes the arrival word and the ticket for the next layer's launch.
// All heads + N / 16 workgroups are resident at once (256 threads, a few KiB of LDS each: up to 8 per CU), so the wait
// cannot starve the producers; it is bounded anyway (a timeout sets the engine's error word instead of hanging).
// What it saves per layer: one kernel boundary and the o_proj launch's ramp + weight stream (measured: see
// profiles/r02p_attn_o_fusion.txt).
#include "woq_attn_body.h"
#include "woq_gemv_xq_body.h"
#include "woq_launch.h"

namespace woq {

struct AttnOArgs {
  // attention
  const float* qkv;
  void* kcache;
  void* vcache;
  const int32_t* pos;
  const float* cs;
  const float* sn;
  int heads, kv_heads, window;
  float* attn_out;
  XqPtrs xq_attn;
  // o_proj
  const u32x4* q;
  const void* scales;
  const uint8_t* zp;
  int tiles_k, K, base_tiles, rem_tiles, n_groups, tpg_shift, N, flags, nw;
  float* out;
  const float* residual;
  XqPtrs xo;
  const float* next_norm_w;
  float* ssq_out;
  // hand-off
  uint32_t* sync;  // [0] heads arrived, [1] o_proj workgroups finished, [2] error
};

template <typename KV, int HD, int TPW, int SMODE, bool ASYM, bool S32>
__global__ __launch_bounds__(256) void attn_o_kernel(AttnOArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int bx = (int)blockIdx.x;
  if (bx < a.heads) {
    attn_decode_body<KV, HD, false, true>(a.qkv, (KV*)a.kcache, (KV*)a.vcache, a.pos, a.cs, a.sn, a.heads, a.kv_heads,
                                          a.window, a.attn_out, a.xq_attn, bx, 0, 1, (float*)smem_raw);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have left
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&a.sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  gemv_xq_body<TPW, 1, SMODE, ASYM, S32, true>(a.q, a.scales, a.xq_attn.limbs, a.xq_attn.u, a.tiles_k, a.K, a.base_tiles,
                                               a.rem_tiles, a.n_groups, a.tpg_shift, a.zp, a.xq_attn.sx, a.out, nullptr,
                                               a.residual, 0.f, a.N, a.flags, 0, nullptr, 0, a.xo, a.next_norm_w,
                                               a.ssq_out, bx - a.heads, a.nw, smem_raw, a.sync, (uint32_t)a.heads,
                                               a.sync + 2);
  if (threadIdx.x == 0) {  // the last strip to finish re-arms the hand-off for the next launch
    const uint32_t n_strips = gridDim.x - (uint32_t)a.heads;
    if (__hip_atomic_fetch_add(&a.sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_strips - 1) {
      __hip_atomic_store(&a.sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <typename KV, int HD, int TPW>
static int launch_attn_o_sm(const AttnOArgs& a, int smode, bool asym, bool s32, int grid, size_t lds, hipStream_t st) {
#define WOQ_AO_CASE(SM, AS, S3)                                                                                  \
  if (smode == SM && asym == AS && s32 == S3) {                                                                  \
    auto kern = attn_o_kernel<KV, HD, TPW, SM, AS, S3>;                                                          \
    static bool attr_set = false;                                                                                \
    if (!attr_set) {                                                                                             \
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      if (e != hipSuccess) return woq::fail(std::string("QBits: hipFuncSetAttribute: ") + hipGetErrorString(e)); \
      attr_set = true;                                                                                           \
    }                                                                                                            \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);                                                 \
    return 0;                                                                                                    \
  }
  WOQ_AO_CASE(0, false, false)
  WOQ_AO_CASE(0, false, true)
  WOQ_AO_CASE(0, true, false)
  WOQ_AO_CASE(0, true, true)
  WOQ_AO_CASE(1, false, false)
  WOQ_AO_CASE(1, false, true)
  WOQ_AO_CASE(1, true, false)
  WOQ_AO_CASE(1, true, true)
#undef WOQ_AO_CASE
  return woq::fail("QBits: bad attention + o_proj configuration");
}

// Does the fused launch take this step? One workgroup per head (no context slices), an fp16 KV cache, head_dim 64 or
// 128, and an o_proj blob the XQ kernel serves with at most four K-slice waves (K <= 4096) in one launch.
bool attn_o_supported(const woq_blob_header& h, int kv_dtype, int D, int splits, int heads) {
  if (kv_dtype != WOQ_F16 || (D != 64 && D != 128) || splits > 1) return false;
  if (h.weight_type != WOQ_W_INT4_CLIP || h.off_shuffle != 0 || (h.K % WOQ_TILE_K) != 0 || h.K != h.Kpad) return false;
  if (h.K != heads * D || h.N != h.Npad) return false;
  if (h.scale_mode == 0 && h.n_groups > 1) {
    const int tpg = h.group / WOQ_TILE_K;
    if (tpg < 1 || (tpg & (tpg - 1)) != 0) return false;
  }
  int nw, tpw;
  return gemv_tile_geometry(h.Kpad / WOQ_TILE_K, 1, (int)h.scale_mode, nw, tpw) && nw <= 4;
}

int launch_attn_o(const float* qkv, void* kcache, void* vcache, int kv_dtype, const int32_t* pos, const float* cs,
                  const float* sn, int heads, int kv_heads, int D, int max_ctx, int window, float* attn_out,
                  const XqPtrs& xq_attn, const void* blob, const woq_blob_header& h, float* out, const float* residual,
                  const XqPtrs& xo, const float* next_norm_w, float* ssq_out, uint32_t* sync, hipStream_t st) {
  if (!attn_o_supported(h, kv_dtype, D, 1, heads)) return woq::fail("QBits: step not covered by the fused attention + o_proj");
  AttnOArgs a;
  a.qkv = qkv;
  a.kcache = kcache;
  a.vcache = vcache;
  a.pos = pos;
  a.cs = cs;
  a.sn = sn;
  a.heads = heads;
  a.kv_heads = kv_heads;
  a.window = window;
  a.attn_out = attn_out;
  a.xq_attn = xq_attn;
  const uint8_t* b = (const uint8_t*)blob;
  a.q = (const u32x4*)(b + h.off_q);
  a.scales = b + h.off_scale;
  a.zp = h.off_zp ? b + h.off_zp : nullptr;
  a.tiles_k = h.Kpad / WOQ_TILE_K;
  a.K = h.K;
  a.N = h.N;
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
  int tpw;
  gemv_tile_geometry(a.tiles_k, 1, (int)h.scale_mode, a.nw, tpw);
  a.base_tiles = a.tiles_k / a.nw;
  a.rem_tiles = a.tiles_k % a.nw;
  a.out = out;
  a.residual = residual;
  a.xo = xo;
  a.next_norm_w = next_norm_w;
  a.ssq_out = ssq_out;
  a.sync = sync;
  const int reach = window > 0 ? std::min(window, max_ctx) : max_ctx;
  const int GP = 64 / (D / 8);
  const size_t lds_attn = (size_t)(3 * D + 8 + 4 * GP * D + ((reach + 4) & ~3)) * 4;
  const size_t lds = std::max(lds_attn, xq_lds_bytes(a.nw, tpw, 1));
  if (lds > 160 * 1024) return woq::fail("QBits: max_ctx too large for the fused attention + o_proj");
  const int grid = heads + h.Npad / WOQ_TILE_N;
  const bool asym = a.zp != nullptr, s32 = h.scale_type == WOQ_F32;
  const int smode = (int)h.scale_mode;
  if (D == 128)
    return tpw == 8 ? launch_attn_o_sm<_Float16, 128, 8>(a, smode, asym, s32, grid, lds, st)
                    : launch_attn_o_sm<_Float16, 128, 4>(a, smode, asym, s32, grid, lds, st);
  return tpw == 8 ? launch_attn_o_sm<_Float16, 64, 8>(a, smode, asym, s32, grid, lds, st)
                  : launch_attn_o_sm<_Float16, 64, 4>(a, smode, asym, s32, grid, lds, st);
}

}  // namespace woq
