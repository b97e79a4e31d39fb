// woq_gemv_xqs.h — the batch-1 decode GEMV of the fused engine (round 3): int4 weights x an activation vector that
// arrives in XQ form (woq_xq.h: balanced-digit limb blocks written by the producing kernel's epilogue).
//
// Arithmetic and parity definition: reference qbits.cpp:113-140, autograd/functions.py:41-63 — exact int8 x int8 ->
// int32 sums on v_mfma_i32_16x16x64_i8, one fp32 recombination per 16-k block, group scale (and zero point) applied
// per block, fp32 across blocks.
//
// What round 3 changed, and the measurement behind each change (tools/xq_probe.hip, profiles/r03b_xq_knockouts.txt):
// the round-2 kernel with every load's descriptor emptied — no memory traffic at all — still took 5.0 us per qkv
// launch, exactly what a load-only twin takes WITH the traffic, and its bare skeleton (no traffic, no arithmetic, no
// epilogue) 3.6 us. The kernel was bound by its own instruction stream: ~300 instructions of per-wave set-up and ~50
// per tile for 8 tiles of work, twelve waves per CU. So:
//   * the weight requests are the first thing a wave does (their addresses need only the preloaded kernel arguments),
//     and they move through a ROLLING WINDOW of D tiles — tile t + D is requested when tile t is consumed — so the
//     arithmetic of a tile runs under the stream of the next ones instead of after the last byte;
//   * balanced signed digits instead of offset-binary limbs (woq_xq.h): no ones row, no bias terms — the
//     recombination of an MFMA result is 4 VALU instead of 9, and the A operand needs no ones block;
//   * per-tile factors (block factors u, block sums sx, group scales, zero points) come from small wave-private LDS
//     tables filled by ONE vector request each, instead of 16 cross-lane shuffles and 8 tiny global requests per
//     column tile (as many trips through the address unit as the weight tiles themselves);
//   * what only the 16 epilogue lanes need (residual, next norm weight, RMSNorm partials) is requested by wave 0 only,
//     up front, so nothing in the serial tail waits on memory.
#pragma once
#include "woq_comm_dev.h"
#include "woq_gemv_common.h"
#include "woq_xq.h"

#ifdef WOQ_XQS_STAMPS
extern __device__ unsigned long long* g_xqs_probe;
// the buffer pointer is read ONCE per wave (WOQ_XQS_STAMP_INIT, before stamp 0): a stamp that loads it itself waits a
// global round trip (~0.5-1 us) in front of its clock read — round 6's first table (profiles/r06c_*) had that artefact
#define WOQ_XQS_STAMP_INIT()                                                            \
  unsigned long long* xqs_probe_ = g_xqs_probe;                                         \
  {                                                                                     \
    const unsigned long long a_ = (unsigned long long)xqs_probe_;                       \
    xqs_probe_ = (unsigned long long*)(((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(a_ >> 32)) << 32) | \
                                       (unsigned int)__builtin_amdgcn_readfirstlane((int)a_));                                  \
  }
#define WOQ_XQS_STAMP(k)                                                              \
  do {                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                \
    if (xqs_probe_ && lane == 0) {                                                    \
      unsigned long long* slot_ = xqs_probe_ + ((size_t)blockIdx.x * 16 + wid) * 16;  \
      slot_[(k)] = wall_clock64();                                                    \
    }                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                \
  } while (0)
#else
#define WOQ_XQS_STAMP_INIT() \
  do {                       \
  } while (0)
#define WOQ_XQS_STAMP(k) \
  do {                   \
  } while (0)
#endif

// experiment switches of the probe build (tools/xq_probe.hip, -DWOQ_XQS_KNOBS): bits 4.. of `flags` knock out one
// stage each by emptying its buffer descriptors or skipping its arithmetic. Nothing in the product build.
#ifdef WOQ_XQS_KNOBS
#define WOQ_XK(bit) (((flags) >> (bit)) & 1)
#else
#define WOQ_XK(bit) false
#endif

// how many of the first D weight tiles a wave requests BEFORE its small (L2-resident) requests; the rest follow them.
// Returns are in order, so whatever is in front of the limbs delays the first MFMA (tools/xq_probe.hip: A/B builds).
// polling interval of a wave that waits for its input blocks inside a chained launch, in units of 64 clocks: pollers
// share the memory pipe with the streaming producers (MI355X_MICROARCH.md, polling-cost)
#ifndef WOQ_CHAIN_SLEEP
#define WOQ_CHAIN_SLEEP 8
#endif
// Round 6 re-measured them on the round-6 kernel (profiles/r06ad_gemv_occupancy_and_window.txt, same-box A/Bs): single
// column tiles 2 -> 1 (0 / 1 / 2 / 4 / 6: 998 / 1005 / 1001 / 987 / 981 tokens/s), the gate/up pairs all of the window -> 0,
// the small requests first (0 / 1 / 2 beside singles at 1: 1016 / 1005 / 1000); the fused launch's q strips stay at 1
// (0 / 1 / 2: 995 / 998 / 997) and its one-tile k / v strips go to 0 (1003 vs 998)
#ifndef WOQ_XQS_PRE
#define WOQ_XQS_PRE 1
#endif
#ifndef WOQ_XQS_PRE_PAIR
#define WOQ_XQS_PRE_PAIR 0
#endif
#ifndef WOQ_XQS_PRE_Q
#define WOQ_XQS_PRE_Q 1
#endif
#ifndef WOQ_XQS_PRE_KV
#define WOQ_XQS_PRE_KV 0
#endif
// 1: no workgroup barrier behind the stream — every wave leaves its partial sums in the slab and bumps an LDS counter,
// the wave that arrives LAST runs the epilogue at once (A/B builds: tools/mkvariant_xq.sh last -DWOQ_XQS_LAST=1;
// record: profiles/r06c_*)
// A/B builds (tools/mkvariant_xq.sh): the scale / zero-point requests in front of the limb requests; their cache policy
#ifndef WOQ_XQS_SC_FIRST
#define WOQ_XQS_SC_FIRST 0
#endif
#ifndef WOQ_XQS_SC_AUX
#define WOQ_XQS_SC_AUX 0
#endif
#ifndef WOQ_XQS_LAST
#define WOQ_XQS_LAST 0
#endif
#ifndef WOQ_XQS_GU_WAVES  // A/B builds: -DWOQ_XQS_GU_WAVES=1 = the compiler's own choice (82 registers, five waves per SIMD)
#define WOQ_XQS_GU_WAVES 6
#endif

namespace woq {

// LDS: [zero block 256, shared: every wave writes the same zeros before it reads them][per wave: limb strip TPW x 384 |
// u | sx | scale slices | zero-point slices][slab nw x CB x 16 f32][64 f32 scratch]
template <int TPW, int CB, int SMODE, bool ASYM, bool S32>
struct XqsLds {
  static constexpr int ESZ = S32 ? 4 : 2;
  static constexpr int STRIP = TPW * 384;
  static constexpr int UTAB = ((TPW * 8 + 63) / 64) * 256;  // one fp32 per block of the slice
  static constexpr int SXTAB = ASYM ? UTAB : 0;
  static constexpr int SCB = (SMODE == 0 ? TPW * 16 : TPW * 64) * ESZ;  // one column tile's scale slice
  static constexpr int ZPB = ASYM ? (SMODE == 0 ? TPW * 16 : TPW * 64) : 0;
  static constexpr int O_STRIP = 0, O_U = O_STRIP + STRIP, O_SX = O_U + UTAB, O_SC = O_SX + SXTAB,
                       O_ZP = O_SC + CB * SCB;
  static constexpr int WAVE = O_ZP + CB * ZPB;
  static_assert(WAVE % 16 == 0, "wave region must keep 16-byte alignment");
  __host__ __device__ static constexpr size_t total(int nw) {
    return 256 + (size_t)nw * WAVE + (size_t)nw * CB * 16 * 4 + 256;
  }
};

__device__ __forceinline__ float xqs_swap32(float v) {
  const uint32_t b = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
  return __uint_as_float(r[1]);
}
__device__ __forceinline__ float xqs_swap16(float v) {
  const uint32_t b = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
  return __uint_as_float(r[1]);
}
// rows 4 e .. 4 e + 2 of the MFMA result = quarter e's three balanced digits against 16 q:
// sum_k 16 q_k v_k = d0 + 2^8 d1 + 2^16 d2 (|d0 + 2^8 d1| < 2^27: exact in int32), rounded to fp32 twice at most
__device__ __forceinline__ float digit_combine(const i32x4& d) {
  return fmaf((float)d.z, 65536.f, (float)(d.x + (d.y << 8)));
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// In-launch chaining (round 3; its one user left the library: tools/rejected/woq_gemv_chain.hip): the input vector may still be in the making by workgroups of the SAME launch
// (CHAIN_IN: every wave polls the flags of its K slice's blocks, bounded, and reads the blocks with agent-scope loads),
// and the output vector may have a consumer there (out.flag != null: published block by block, woq_xq.h XqPub).
struct XqsChain {
  const unsigned int* in_flag;  // [K / 16] flags of the input vector's blocks
  unsigned int in_tag;
  XqPub out;
  int* status;  // sticky give-up flag (bit 0)
  int strip;    // this workgroup's column strip (pair) when the grid holds several roles; -1 = blockIdx.x
};

// The kernel arguments a wave does NOT need to put its window and its small requests out. A kernel's first read of its
// argument segment costs ~1 us inside a replayed graph (profiles/r03d_kernarg_latency.txt), and hipcc loads every
// declared argument at the top of the kernel — the stage stamps (profiles/r06c_xqs_stage_stamps.txt) showed each wave
// sitting ~1 us between its first two weight requests and the rest of its window. So these travel as ONE by-value struct
// that the kernel never touches as a parameter: the body reads its fields from the argument segment by hand
// (XqsLatePtr, scalar loads) BEHIND the window. The first 14 argument dwords stay individual and preloaded.
struct XqsLate {
  const uint8_t* zp;  // read early by asymmetric blobs only
  const float* xsx;   // "
  float* out;         // fp32 [N], or (FUSED) {tag, fp32} granules
  const float* bias;
  const float* residual;
  const float* ssq_in;
  const float* next_norm_w;
  float* ssq_out;
  const CommDev* tp;
  const unsigned int* tag_seq;  // FUSED: device-side step counter the granule tags are made of
  XqPtrs xo;
  float eps;
  int N, K, n_ssq, tag_layer;
  LutArgs lut;
};
static_assert(alignof(XqsLate) == 8, "the struct starts at byte 56 of the argument segment, behind 14 preloaded dwords");
typedef const __attribute__((address_space(4))) XqsLate* XqsLatePtr;
__device__ __forceinline__ XqsLatePtr xqs_late_ptr() {
  return (XqsLatePtr)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + 56);
}

// FUSED (woq_gemv_attn.hip): the outputs are consumed by another workgroup of the SAME launch: `out` is then an array
// of 8-byte {tag, fp32} granules, each written by ONE write-through agent-scope store (the data is its own flag).
// NDIG: 0 = int4 weights; 1 | 2 | 3 = a 4-bit table type (nf4 / fp4) as that many digit planes (woq_gemv_common.h LutArgs)
template <int TPW, int CB, int D, int SMODE, bool ASYM, bool S32, bool FUSED, bool CHAIN_IN = false, int NDIG = 0>
__device__ __forceinline__ void gemv_xqs_body(
    unsigned char* smem_raw, const u32x4* __restrict__ q, const void* __restrict__ scales,
    const uint8_t* __restrict__ xlimbs, const float* __restrict__ xu, int tiles_k, int kt_off, int base_tiles,
    int rem_tiles, int n_groups, int tpg_shift, int flags, int nw, XqsLatePtr late,
    const XqsChain& chain = XqsChain{nullptr, 0u, XqPub{nullptr, 0u}, nullptr, -1}) {
  // asymmetric blobs: zero points and block sums are small requests of stage 2 — their pointers are read up front
  const uint8_t* zp = ASYM ? late->zp : nullptr;
  const float* xsx = ASYM ? late->xsx : nullptr;
  static_assert(!(ASYM && NDIG > 0), "table weight types are symmetric");
  typedef XqsLds<TPW, CB, SMODE, ASYM, S32> L;
  constexpr int ESZ = L::ESZ;
  constexpr int DD = D < TPW ? D : TPW;  // tiles requested before the first one is consumed
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = uni(tid >> 6);
  // (nw = waves of the workgroup arrives in a preloaded dword: blockDim is a hidden argument, i.e. argument-segment memory)
  const int bx = chain.strip >= 0 ? chain.strip : (int)blockIdx.x;
  WOQ_XQS_STAMP_INIT();
  WOQ_XQS_STAMP(0);
  const int kt0 = kt_off + wid * base_tiles + min(wid, rem_tiles);
  const int cnt = base_tiles + (wid < rem_tiles ? 1 : 0);
  const int v16 = lane * 16;

  // ---- 1. the first D weight tiles of the wave's slice: nothing is in front of them ----
  u32x4 w[CB][TPW];
  rsrc_t rq[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
    rq[cb] = make_rsrc(q + (size_t)(bx * CB + cb) * tiles_k * 64,
                       WOQ_XK(8) ? 0 : uni(min(kt0 + cnt, tiles_k) * 1024));
  // weight tiles requested in front of the small requests (WOQ_XQS_PRE* above): one for single column tiles and the fused
  // launch's q strips, none for the gate/up pairs and the one-tile-deep k / v strips (round 3 had two / all of the window:
  // profiles/r03i_xq_issue_order.txt; re-measured on the round-6 kernel: profiles/r06ad_gemv_occupancy_and_window.txt)
  constexpr int PRE_WANT = CB == 2 ? WOQ_XQS_PRE_PAIR : !FUSED ? WOQ_XQS_PRE : D == 1 ? WOQ_XQS_PRE_KV : WOQ_XQS_PRE_Q;
  constexpr int PRE = CHAIN_IN ? DD : (PRE_WANT < DD ? PRE_WANT : DD);
#pragma unroll
  for (int t = 0; t < PRE; ++t)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
      w[cb][t] = __builtin_amdgcn_raw_buffer_load_b128(rq[cb], v16 + t * 1024, kt0 * 1024, AUX_NT);
  __builtin_amdgcn_sched_barrier(0);

  // ---- 2. the small requests (L2-resident: written by the previous kernel, or shared by every workgroup) ----
  constexpr int XP = (L::STRIP + 1023) / 1024;  // 1-KiB pieces per limb strip
  constexpr int UL = (TPW * 8 + 63) / 64;       // block factors per lane (8 blocks per tile)
  constexpr int AUX_IN = CHAIN_IN ? 16 : 0;     // sc1: agent-scope loads of data another workgroup just published
  const int nblk = WOQ_XK(5) ? 0 : uni(max(0, min(cnt * 8, tiles_k * 8 - kt0 * 8)));
  if constexpr (CHAIN_IN) {
    // the blocks of this wave's K slice, one flag per lane and pass; the weight requests above are already in flight
    const unsigned int* f = chain.in_flag + (size_t)kt0 * 8;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
      bool good = true;
#pragma unroll
      for (int j = 0; j < UL; ++j)
        if (lane + 64 * j < nblk)
          good = good && __hip_atomic_load(f + lane + 64 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == chain.in_tag;
      if (__all(good)) break;
      if (wall_clock64() - t0 > 2000000ull) {  // 20 ms at 100 MHz: a producer is missing — say so, do not hang
        if (lane == 0 && chain.status != nullptr) atomicOr(chain.status, 1);
        break;
      }
      __builtin_amdgcn_s_sleep(WOQ_CHAIN_SLEEP);
    }
  }
  u32x4 xl[XP];
  float uw[UL], sxw[UL];
  constexpr int NSP = (L::SCB + 1023) / 1024;
  u32x4 sl[CB][NSP];
  u32x4 zl[CB];
  int g0 = 0;
  if constexpr (SMODE == 0) g0 = min(kt0 >> tpg_shift, n_groups - 1);
  auto req_x = [&]() {
    const rsrc_t rl =
        make_rsrc(xlimbs + (size_t)kt0 * 384, WOQ_XK(4) ? 0 : uni(max(0, min(cnt, tiles_k - kt0)) * 384));
#pragma unroll
    for (int j = 0; j < XP; ++j) xl[j] = __builtin_amdgcn_raw_buffer_load_b128(rl, v16 + j * 1024, 0, AUX_IN);
  // block factors of the slice: lane L holds blocks kt0 * 8 + L (+ 64 ...) (8 blocks per tile; reads past the slice
  // return 0 through the descriptor, so tiles past the slice end contribute exactly 0)
  const rsrc_t ru = make_rsrc(xu + (size_t)kt0 * 8, nblk * 4);
#pragma unroll
  for (int j = 0; j < UL; ++j) {
    uw[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ru, (lane + 64 * j) * 4, 0, AUX_IN));
    sxw[j] = 0.f;
  }
  if constexpr (ASYM) {
    const rsrc_t rsx = make_rsrc(xsx + (size_t)kt0 * 8, nblk * 4);
#pragma unroll
    for (int j = 0; j < UL; ++j)
      sxw[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsx, (lane + 64 * j) * 4, 0, AUX_IN));
  }
  };
  // the wave's slice of scales (and zero points) of each column tile: one vector request per KiB
  auto req_s = [&]() {
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    const int tn = bx * CB + cb;
    rsrc_t rs, rz;
    if constexpr (SMODE == 0) {
      rs = make_rsrc((const char*)scales + ((size_t)tn * n_groups + g0) * 16 * ESZ,
                     WOQ_XK(5) ? 0 : uni((n_groups - g0) * 16 * ESZ));
      rz = make_rsrc(zp + ((size_t)tn * n_groups + g0) * 16, uni((n_groups - g0) * 16));
    } else {
      const int left = uni(max(0, tiles_k - kt0));
      rs = make_rsrc((const char*)scales + ((size_t)tn * tiles_k + kt0) * 64 * ESZ, WOQ_XK(5) ? 0 : left * 64 * ESZ);
      rz = make_rsrc(zp + ((size_t)tn * tiles_k + kt0) * 64, left * 64);
    }
#pragma unroll
    for (int j = 0; j < NSP; ++j) sl[cb][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, v16 + j * 1024, 0, WOQ_XQS_SC_AUX);
    if constexpr (ASYM) zl[cb] = __builtin_amdgcn_raw_buffer_load_b128(rz, v16, 0, WOQ_XQS_SC_AUX);
  }
  };
#if WOQ_XQS_SC_FIRST
  req_s();
  __builtin_amdgcn_sched_barrier(0);
  req_x();
#else
  req_x();
  req_s();
#endif
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = PRE; t < DD; ++t)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
      w[cb][t] = __builtin_amdgcn_raw_buffer_load_b128(rq[cb], v16 + t * 1024, kt0 * 1024, AUX_NT);
  __builtin_amdgcn_sched_barrier(0);
  // Everything above needs only the 14 PRELOADED kernel-argument dwords (symmetric blobs). The rest of the argument
  // segment is read HERE, behind the window, and by wave 0 ALONE: it issues the epilogue's requests and runs the
  // epilogue; the other waves never touch the segment (table weight types: their digit planes, all waves).
  const float *residual = nullptr, *next_norm_w = nullptr, *ssq_in = nullptr, *bias = nullptr;
  int n_ssq = 0, N = 0, K = 0;
  float eps = 0.f;
  float *out = nullptr, *ssq_out = nullptr;
  const CommDev* tp = nullptr;
  XqPtrs xo = {nullptr, nullptr, nullptr};
  unsigned int fused_tag = 0u;
  if (wid == 0 || WOQ_XQS_LAST) {  // (the rejected last-arriver form: any wave may run the epilogue)
    residual = late->residual, next_norm_w = late->next_norm_w, ssq_in = late->ssq_in, bias = late->bias;
    n_ssq = late->n_ssq, N = late->N, K = late->K, eps = late->eps;
    out = late->out, ssq_out = late->ssq_out, tp = late->tp;
    xo = XqPtrs{late->xo.limbs, late->xo.u, late->xo.sx};
    if constexpr (FUSED) fused_tag = (late->tag_seq[0] << 6) | (unsigned int)late->tag_layer;
  }
  LutArgs lut;
  if constexpr (NDIG > 0) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) lut.d[j][i] = late->lut.d[j][i];
    lut.wmul = late->lut.wmul;
  }
  // epilogue inputs (wave 0 runs the epilogue on its lanes 0..15): residual element, next norm weight, RMSNorm
  // partials (256 per 16-B piece, only the pieces that exist) — requested now, used after the barrier
  const bool silu = (flags & 2) != 0;
  float e_res = 0.f, g_next = 1.f;
  float4_t ssq_v[4];
  if (wid == 0) {
    const int n0 = bx * 16;
    const int nlive = uni(max(0, min((silu ? (N >> 1) : N) - n0, 16)));
    if (residual != nullptr && !WOQ_XK(5)) {
      const rsrc_t rr = make_rsrc(residual + n0, nlive * 4);
      e_res = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, min(lane, 15) * 4, 0, 0));
    }
    if (next_norm_w != nullptr && !WOQ_XK(5)) {
      const rsrc_t rg = make_rsrc(next_norm_w + n0, nlive * 4);
      g_next = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, min(lane, 15) * 4, 0, 0));
    }
    if (ssq_in != nullptr) {
      const rsrc_t rs = make_rsrc(ssq_in, WOQ_XK(5) ? 0 : n_ssq * 4);
      ssq_v[0] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, v16, 0, 0));
#pragma unroll
      for (int j = 1; j < 4; ++j) {
        ssq_v[j] = float4_t{0.f, 0.f, 0.f, 0.f};
        if (n_ssq > j * 256)
          ssq_v[j] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, v16 + j * 1024, 0, 0));
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  WOQ_XQS_STAMP(1);

  // ---- 3. park the small pieces in the wave's LDS region (wave-private, in-order LDS: no workgroup barrier) ----
  unsigned char* zero_blk = smem_raw;
  unsigned char* wbase = smem_raw + 256 + (size_t)wid * L::WAVE;
  float* slab = (float*)(smem_raw + 256 + (size_t)nw * L::WAVE);  // [nw][CB][16]
  float* red = slab + nw * CB * 16;                          // [64]
  ((uint32_t*)zero_blk)[lane] = 0u;
#pragma unroll
  for (int j = 0; j < XP; ++j)
    if (v16 + j * 1024 < L::STRIP) *(u32x4*)(wbase + L::O_STRIP + v16 + j * 1024) = xl[j];  // past the slice: zeros
#pragma unroll
  for (int j = 0; j < UL; ++j) {
    ((float*)(wbase + L::O_U))[lane + 64 * j] = uw[j];
    if constexpr (ASYM) ((float*)(wbase + L::O_SX))[lane + 64 * j] = sxw[j];
  }
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
    for (int j = 0; j < NSP; ++j)
      if (v16 + j * 1024 < L::SCB) *(u32x4*)(wbase + L::O_SC + cb * L::SCB + v16 + j * 1024) = sl[cb][j];
    if constexpr (ASYM)
      if (v16 < L::ZPB) *(u32x4*)(wbase + L::O_ZP + cb * L::ZPB + v16) = zl[cb];
  }
  __builtin_amdgcn_wave_barrier();
#if WOQ_XQS_LAST
  // red[1]: arrival counter, zero before any wave can bump it (LDS-only barrier: the weight window stays in flight)
  if (tid == 0) ((unsigned int*)red)[1] = 0u;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
  // A-operand addresses: MFMA row r = lane & 15 -> quarter e = r >> 2, digit p = r & 3 (row 4 e + 3 stays zero);
  // a row is live in lane quarter kq == e only, everything else reads the zero block
  const int i16 = lane & 15, kq = lane >> 4;
  const bool a_live = (i16 >> 2) == kq && (i16 & 3) != 3;
  const unsigned char* a_base = a_live ? wbase + L::O_STRIP + kq * 48 + (i16 & 3) * 16 : zero_blk + kq * 16;
  const int a_step_t = a_live ? 384 : 0, a_step_h = a_live ? 192 : 0;
  const float* u_base = (const float*)(wbase + L::O_U) + kq;    // [t * 8 + h * 4]
  const float* sx_base = (const float*)(wbase + L::O_SX) + kq;  // (ASYM)
  const bool bf = (flags & 1) != 0;
  const i32x4 izero = {0, 0, 0, 0};
  float tot[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) tot[cb] = 0.f;
  WOQ_XQS_STAMP(2);

  // ---- 4. inner products: request tile t + D, consume tile t ----
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    if (t + DD < TPW) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
        w[cb][t + DD] = __builtin_amdgcn_raw_buffer_load_b128(rq[cb], v16 + (t + DD) * 1024, kt0 * 1024, AUX_NT);
    }
    const i32x4 a0 = *(const i32x4*)(a_base + t * a_step_t);
    const i32x4 a1 = *(const i32x4*)(a_base + t * a_step_t + a_step_h);
    const float u0 = u_base[t * 8], u1 = u_base[t * 8 + 4];
    float s0 = 0.f, s1 = 0.f;
    if constexpr (ASYM) s0 = sx_base[t * 8], s1 = sx_base[t * 8 + 4];
    int gi = t;
    if constexpr (SMODE == 0) gi = min((kt0 + t) >> tpg_shift, n_groups - 1) - g0;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const u32x4 wv = w[cb][t];
      if (WOQ_XK(6)) {  // probe: consume the tile, no arithmetic
        tot[cb] += __uint_as_float((wv.x ^ wv.y) ^ (wv.z ^ wv.w));
        continue;
      }
      const unsigned char* scp = wbase + L::O_SC + cb * L::SCB;
      const unsigned char* zpp = wbase + L::O_ZP + cb * L::ZPB;
      float f0, f1;
      if constexpr (NDIG > 0) {  // table weights: one MFMA per digit plane and half, most significant plane first
        i32x4 b0[NDIG], b1[NDIG];
        lut_b<NDIG>(lut, wv.x, wv.y, b0);
        lut_b<NDIG>(lut, wv.z, wv.w, b1);
        f0 = digit_combine(__builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0[NDIG - 1], izero, 0, 0, 0));
        f1 = digit_combine(__builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1[NDIG - 1], izero, 0, 0, 0));
#pragma unroll
        for (int j = NDIG - 2; j >= 0; --j) {
          f0 = fmaf(f0, 256.f, digit_combine(__builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0[j], izero, 0, 0, 0)));
          f1 = fmaf(f1, 256.f, digit_combine(__builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1[j], izero, 0, 0, 0)));
        }
      } else {
        const i32x4 b0 = {(int)((wv.x << 4) & 0xf0f0f0f0u), (int)(wv.x & 0xf0f0f0f0u),
                          (int)((wv.y << 4) & 0xf0f0f0f0u), (int)(wv.y & 0xf0f0f0f0u)};
        const i32x4 b1 = {(int)((wv.z << 4) & 0xf0f0f0f0u), (int)(wv.z & 0xf0f0f0f0u),
                          (int)((wv.w << 4) & 0xf0f0f0f0u), (int)(wv.w & 0xf0f0f0f0u)};
        const i32x4 d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, izero, 0, 0, 0);
        const i32x4 d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, izero, 0, 0, 0);
        f0 = digit_combine(d0), f1 = digit_combine(d1);
      }
      if constexpr (SMODE == 0) {
        if constexpr (ASYM) {  // the weights carry 16 * q: the zero point enters as 16 * zp
          const float z16 = -16.f * (float)((int)zpp[gi * 16 + i16] - 8);
          f0 = fmaf(z16, s0, f0);
          f1 = fmaf(z16, s1, f1);
        }
        float sc;
        if constexpr (S32)
          sc = *(const float*)(scp + (gi * 16 + i16) * 4);
        else
          sc = tscale16(*(const uint16_t*)(scp + (gi * 16 + i16) * 2), bf);
        tot[cb] = fmaf(sc, fmaf(f0, u0, f1 * u1), tot[cb]);
      } else {  // this lane quarter's 32-k group of each half: s = 2 h + (kq >> 1)
        const int o = (t * 16 + i16) * 4 + (kq >> 1);
        if constexpr (ASYM) {
          f0 = fmaf(-16.f * (float)((int)zpp[o] - 8), s0, f0);
          f1 = fmaf(-16.f * (float)((int)zpp[o + 2] - 8), s1, f1);
        }
        float sc0, sc1;
        if constexpr (S32) {
          sc0 = *(const float*)(scp + o * 4);
          sc1 = *(const float*)(scp + (o + 2) * 4);
        } else {
          sc0 = tscale16(*(const uint16_t*)(scp + o * 2), bf);
          sc1 = tscale16(*(const uint16_t*)(scp + (o + 2) * 2), bf);
        }
        tot[cb] = fmaf(sc0 * u0, f0, fmaf(sc1 * u1, f1, tot[cb]));
      }
      // three digit planes of two halves are 24 registers per column tile: keep one tile's planes from being built
      // under the previous one's MFMAs (the 128-register budget of the 1024-thread launches spills otherwise)
      if constexpr (NDIG == 3) __builtin_amdgcn_sched_barrier(0);
    }
    if (t == 0) WOQ_XQS_STAMP(3);
    // memory requests stay in their iteration (that IS the window); ALU, MFMA and LDS work may move across
    __builtin_amdgcn_sched_barrier(0x38f);
  }
  WOQ_XQS_STAMP(4);
  if (WOQ_XK(7)) {  // probe: no cross-wave sum, no epilogue
    if (lane < 16 && out) out[bx * 16 + lane] = tot[0] + tot[CB - 1];
    return;
  }
  // the four lane quarters hold the four blocks' shares of each column
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    float v = NDIG > 0 ? tot[cb] * lut.wmul : tot[cb];
    v += xqs_swap32(v);  // lanes 0..31: this lane + lane ^ 32
    v += xqs_swap16(v);  // lanes 0..15: + lane ^ 16
    if (lane < 16) slab[((size_t)wid * CB + cb) * 16 + lane] = v;
  }
  if (wid == 0 && ssq_in != nullptr) {
    float4_t t4 = (ssq_v[0] + ssq_v[1]) + (ssq_v[2] + ssq_v[3]);
    const float s = wave_sum_dpp((t4.x + t4.y) + (t4.z + t4.w));
    if (lane == 0) red[0] = s;
  }
#if WOQ_XQS_LAST
  if (wid == 0 && lane < 16) {  // what the epilogue lanes need from wave 0's registers: in LDS before wave 0 arrives
    red[16 + lane] = e_res;
    red[32 + lane] = g_next;
  }
  asm volatile("" ::: "memory");
  unsigned int arrived = 0u;
  if (lane == 0) {
    typedef __attribute__((address_space(3))) unsigned int lds_u32;
    arrived = __hip_atomic_fetch_add((lds_u32*)((unsigned int*)red + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  asm volatile("" ::: "memory");
  const bool finisher = uni((int)arrived) == nw - 1;  // LDS runs a wave's operations in order: the counter is behind the row
  if (finisher) {
    e_res = red[16 + (lane & 15)];
    g_next = red[32 + (lane & 15)];
  }
  const int tid_e = finisher ? lane : 64;
#else
  __syncthreads();
  const int tid_e = tid;
#endif
  WOQ_XQS_STAMP(5);

  // ---- 5. finish (lanes 0..15 of wave 0 | of the last wave to arrive): sum over waves, RMSNorm factor, bias, SiLU*mul,
  // residual, store, XQ ----
  if (tid_e < 16) {
    const int tid = tid_e;
    float v = 0.f, up = 0.f;
#pragma unroll 4
    for (int w2 = 0; w2 < nw; ++w2) {
      v += slab[((size_t)w2 * CB) * 16 + tid];
      if constexpr (CB == 2) up += slab[((size_t)w2 * CB + 1) * 16 + tid];
    }
    const float inv = ssq_in != nullptr ? 1.0f / sqrtf(red[0] / (float)K + eps) : 1.f;  // HF LlamaRMSNorm
    v *= inv;
    const int n = bx * 16 + tid;  // CB == 1 or the SiLU pair: one 16-column output tile per workgroup
    if (silu) {
      up *= inv;
      if (bias) {
        v += bias[min((bx * 2) * 16 + tid, N - 1)];
        up += bias[min((bx * 2 + 1) * 16 + tid, N - 1)];
      }
      v = v / (1.0f + __expf(-v)) * up;
    } else if (bias) {
      v += bias[min(n, N - 1)];
    }
    const bool live = n < (silu ? (N >> 1) : N);
    v = live ? v + e_res : 0.f;
    if (tp != nullptr && live) {
      // tensor parallel, row-parallel projection: this value is one rank's PARTIAL sum. Store it into every peer's
      // inbox right here ({fp32, tag} granule, system scope — woq_comm.hip's protocol, tag = the sequence number of
      // the all-reduce that follows): the fabric flight runs under the kernel boundary, the all-reduce kernel only
      // pulls and sums (allreduce_ll_kernel<PUSHED>)
      const unsigned int seq = tp->ctl[0];
      const int world = tp->world, rank = tp->rank;
      for (int r = 0; r < world; ++r)
        if (r != rank) push(tp->peer[r] + ar_slot(*tp, (int)(seq & 1u), rank, (unsigned int)n), __float_as_uint(v), seq);
    }
    if (live && out) {
      if constexpr (FUSED)
        __hip_atomic_store((unsigned long long*)out + n, ((unsigned long long)fused_tag << 32) | __float_as_uint(v),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
        out[n] = v;
    }
    if (xo.limbs != nullptr) {  // this tile IS block blockIdx.x of the next kernel's activation vector
      const float ss = ssq_out != nullptr ? row16_sum(v * v) : 0.f;
      if (chain.out.flag != nullptr)
        xq_emit16<true>(v * g_next, xo, bx, tid, chain.out, ssq_out, ss);
      else
        xq_emit16<false>(v * g_next, xo, bx, tid, chain.out, ssq_out, ss);
    }
  }
  WOQ_XQS_STAMP(6);
}

// flags: bit 0 scales are bf16 (else fp16; ignored for fp32 scales), bit 1 SiLU(gate)*up epilogue (CB == 2)
// The first 14 argument dwords are preloaded into SGPRs (-amdgpu-kernarg-preload-count=14): everything a wave needs to
// put its whole window and its small requests out sits there — `flags` rides in the upper bits of the 14th
// (tpg_flags = tpg_shift | flags << 8 | waves << 16), so the tile loop's scale conversion does not wait for the argument
// segment either.
// (three digit planes x two column tiles need more than the 128 registers of a 1024-thread workgroup: 512 there)
// (round 6: the int4 gate/up pairs — 688 workgroups of 8 waves at the 7B shape — need SIX waves per SIMD for three workgroups
// per CU, i.e. the whole grid resident in one round; at 82 registers the compiler stopped two short of that)
template <int TPW, int CB, int D, int SMODE, bool ASYM, bool S32, int NDIG>
__global__ __launch_bounds__((CB * TPW > 8 || (CB == 2 && NDIG == 3)) ? 512 : 1024)
__attribute__((amdgpu_waves_per_eu((CB == 2 && TPW == 4 && NDIG == 0 && SMODE == 0 && !ASYM && !S32) ? WOQ_XQS_GU_WAVES : 1)))
void gemv_xqs_kernel(
    const u32x4* __restrict__ q, const void* __restrict__ scales, const uint8_t* __restrict__ xlimbs,
    const float* __restrict__ xu, int tiles_k, int kt_off, int base_tiles, int rem_tiles, int n_groups, int tpg_flags,
    XqsLate late_in_the_argument_segment) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  gemv_xqs_body<TPW, CB, D, SMODE, ASYM, S32, false, false, NDIG>(smem_raw, q, scales, xlimbs, xu, tiles_k, kt_off,
                                                                 base_tiles, rem_tiles, n_groups, tpg_flags & 0xff,
                                                                 (tpg_flags >> 8) & 0xff, tpg_flags >> 16, xqs_late_ptr());
}

}  // namespace woq
