// woq_gemm_f16t.h — the prefill GEMM's hand-scheduled K loop on a 256-row x 128-column workgroup tile (round 6).
// Included by woq_gemm_f16.hip behind woq_gemm_f16p.h (inside namespace woq): same operand formats, same packed /
// raw activation layouts, same dequantisation and the same epilogue as gemm_f16p_kernel — twice the rows per wave.
//
// Why. gemm_f16p_kernel gives a wave 128 rows x 32 columns: per 32-k part sixteen MFMAs against the 26 VALU operations
// that dequantise the part's two weight fragments, and every one of the M / 128 row blocks of a call re-does that
// unpack for the same weights. Its MFMA skeleton alone runs 2082 TFLOP/s, the kernel 1343 (profiles/NOTEBOOK_r01_r04.md
// §3.3): the headroom is the unpack, not the matrix pipe. Here a wave owns 256 rows x 32 columns:
// 32 MFMAs per part for the same 26 VALU operations, and half as many workgroups unpack each weight tile.
//
// Shape of a part. Two asm blocks of eight row-tile pairs each (the proven block of woq_gemm_f16p.h: counted lgkmcnt
// waits, ONE set of eight A-fragment registers refilled right behind its MFMAs):
//   block A: row tiles 0..7  (rows 0..127 of the workgroup), refills the set with row tiles 8..15 of the SAME part,
//            dequantises column tile 0 of the NEXT part (13 / 14 VALU operations);
//   block B: row tiles 8..15 (rows 128..255), refills the set with row tiles 0..7 of the NEXT part, dequantises column
//            tile 1 of the next part.
// The accumulators are 128 registers, the whole wave ~240: two workgroups per CU (2 waves per SIMD).
// LDS: TWO slots of 32 KiB, each one 64-k half-tile of the 256 rows = the two 16-KiB half-tile images of the packed
// layout (row blocks 2 mb, 2 mb + 1) side by side; 64 KiB per workgroup. Half-tile q lives in slot q & 1. Per K step
// (half-tiles q0 = 2 kt, q1 = q0 + 1; parts 0, 1 read slot 0, parts 2, 3 slot 1):
//   P0A P0B P1A | vmcnt(0) lgkmcnt(0) barrier X: q1 has landed, nobody reads slot 0 any more |
//   P1B (+ LDS-DMA of q0 + 2 -> slot 0) P2A P2B P3A | vmcnt(0) lgkmcnt(0) barrier Y: q0 + 2 has landed, slot 1 is free |
//   weights of tile kt + 2 requested, P3B (+ LDS-DMA of q1 + 2 -> slot 1).
#pragma once
// (included inside namespace woq)
// development knock-outs (timing only, results are then garbage): bit 0 no LDS-DMA in the loop, bit 1 no workgroup
// barriers in the loop, bit 2 no dequantisation VALU, bit 3 no vmcnt waits in the loop
#ifndef WOQ_T_KNOCK
#define WOQ_T_KNOCK 0
#endif

#define WOQ_T_RD(i) "ds_read_b128 %[a" WOQ_S_(i) "], %[ad] offset:%[ob]+2048*" WOQ_S_(i) "\n\t"
// dequantisation of ONE fragment (blob word w -> q0..q3), the two element orders of woq_gemm_f16p.h
#define WOQ_T_DB(k, src, m, g) "v_bitop3_b32 %[q" WOQ_S_(k) "], %[" src "], %[" g "], %[" m "] bitop3:0x6c\n\t"
#define WOQ_T_DA(k, n) "v_pk_add_f16 %[q" WOQ_S_(k) "], %[q" WOQ_S_(k) "], %[" n "] op_sel_hi:[1,0]\n\t"
#define WOQ_T_DM(k) "v_pk_mul_f16 %[q" WOQ_S_(k) "], %[q" WOQ_S_(k) "], %[r] op_sel_hi:[1,0]\n\t"
#if WOQ_T_KNOCK & 4
#define WOQ_T_V_P0 "v_mov_b32 %[y], 0\n\tv_mov_b32 %[q0], 0\n\tv_mov_b32 %[q1], 0\n\tv_mov_b32 %[q2], 0\n\tv_mov_b32 %[q3], 0\n\t"
#define WOQ_T_V_P1 ""
#define WOQ_T_V_P2 ""
#define WOQ_T_V_P3 ""
#define WOQ_T_V_P4 ""
#define WOQ_T_V_P5 ""
#define WOQ_T_V_P6 ""
#define WOQ_T_V_P7 ""
#else
// packed-A order: 13 operations over the eight pairs
#define WOQ_T_V_P0 "v_lshrrev_b32 %[y], 8, %[w]\n\t" WOQ_T_DB(0, "w", "ml", "gl")
#define WOQ_T_V_P1 WOQ_T_DB(1, "w", "mh", "gh") WOQ_T_DB(2, "y", "ml", "gl")
#define WOQ_T_V_P2 WOQ_T_DB(3, "y", "mh", "gh") WOQ_T_DA(0, "nl")
#define WOQ_T_V_P3 WOQ_T_DA(1, "nh") WOQ_T_DA(2, "nl")
#define WOQ_T_V_P4 WOQ_T_DA(3, "nh") WOQ_T_DM(0)
#define WOQ_T_V_P5 WOQ_T_DM(1)
#define WOQ_T_V_P6 WOQ_T_DM(2)
#define WOQ_T_V_P7 WOQ_T_DM(3)
#endif
// raw-A order: 14 operations
#define WOQ_T_V_R0 "v_perm_b32 %[y], %[w], %[w], %[s1]\n\tv_perm_b32 %[z], %[w], %[w], %[s2]\n\t"
#define WOQ_T_V_R1 WOQ_T_DB(0, "y", "ml", "gl") WOQ_T_DB(1, "z", "ml", "gl")
#define WOQ_T_V_R2 WOQ_T_DB(2, "y", "mh", "gh") WOQ_T_DB(3, "z", "mh", "gh")
#define WOQ_T_V_R3 WOQ_T_DA(0, "nl") WOQ_T_DA(1, "nl")
#define WOQ_T_V_R4 WOQ_T_DA(2, "nh") WOQ_T_DA(3, "nh")
#define WOQ_T_V_R5 WOQ_T_DM(0) WOQ_T_DM(1)
#define WOQ_T_V_R6 WOQ_T_DM(2)
#define WOQ_T_V_R7 WOQ_T_DM(3)
// one block: pair i = wait for A fragment i (returns are in order: 7 reads behind it for pair 0, 6 after), its two
// MFMAs with the refill of fragment i - 1 between them, the LDS-DMA piece X_i (DMA blocks only), 1-2 VALU operations
#define WOQ_TBLOCK_TEXT(V, X0, X1, X2, X3, X4, X5, X6, X7)                                     \
  WOQ_W7 WOQ_MF(0, 0, 0) WOQ_MF(1, 0, 1) X0 WOQ_T_V_##V##0                                      \
  WOQ_W6 WOQ_MF(2, 1, 0) WOQ_T_RD(0) WOQ_MF(3, 1, 1) X1 WOQ_T_V_##V##1                          \
  WOQ_W6 WOQ_MF(4, 2, 0) WOQ_T_RD(1) WOQ_MF(5, 2, 1) X2 WOQ_T_V_##V##2                          \
  WOQ_W6 WOQ_MF(6, 3, 0) WOQ_T_RD(2) WOQ_MF(7, 3, 1) X3 WOQ_T_V_##V##3                          \
  WOQ_W6 WOQ_MF(8, 4, 0) WOQ_T_RD(3) WOQ_MF(9, 4, 1) X4 WOQ_T_V_##V##4                          \
  WOQ_W6 WOQ_MF(10, 5, 0) WOQ_T_RD(4) WOQ_MF(11, 5, 1) X5 WOQ_T_V_##V##5                        \
  WOQ_W6 WOQ_MF(12, 6, 0) WOQ_T_RD(5) WOQ_MF(13, 6, 1) X6 WOQ_T_V_##V##6                        \
  WOQ_W6 WOQ_MF(14, 7, 0) WOQ_T_RD(6) WOQ_MF(15, 7, 1) X7 WOQ_T_V_##V##7                        \
      WOQ_T_RD(7)
// raw-A pieces: each has its own lane offsets; the LDS address steps by 1 KiB in M0, by 16 KiB - 3 KiB between the
// two 128-row images
#define WOQ_T_DMAR(j, step) "global_load_lds_dwordx4 %[gr" WOQ_S_(j) "], %[g0]\n\ts_add_u32 m0, m0, " step "\n\t"

// acc: the sixteen accumulators of this block's eight row tiles; af: their A fragments on entry (possibly in flight),
// on exit the next block's (in flight), read from LDS address `ad` + OB + 2048 i; bc0 / bc1: this part's weight
// fragments; bn: ONE fragment of the next part, dequantised here from the blob word w with scales f.
// DMA: 1 = this block carries the eight LDS-DMA pieces of a half-tile; 2 (packed form, WOQ_T_SPLIT_DMA) = the four
// pieces of ONE 128-row image (g0 / m0a: its source and LDS base), one per two pairs.
template <int OB, int DMA, bool RAW>
__device__ __forceinline__ void gemm_tblock(float4_t (&acc)[8][2], h8 (&af)[8], const u32x4& bc0, const u32x4& bc1,
                                            uint32_t (&bn)[4], uint32_t ad, uint32_t w, const PhaseConst& k,
                                            const FragScale& f, uint32_t gv, const void* g0, const void* g1,
                                            uint32_t m0a, uint32_t m0b, const RawOffsets& ro) {
  uint32_t y, z;
#define WOQ_T_OPERANDS                                                                                               \
  [c0] "+v"(acc[0][0]), [c1] "+v"(acc[0][1]), [c2] "+v"(acc[1][0]), [c3] "+v"(acc[1][1]), [c4] "+v"(acc[2][0]),      \
      [c5] "+v"(acc[2][1]), [c6] "+v"(acc[3][0]), [c7] "+v"(acc[3][1]), [c8] "+v"(acc[4][0]), [c9] "+v"(acc[4][1]),  \
      [c10] "+v"(acc[5][0]), [c11] "+v"(acc[5][1]), [c12] "+v"(acc[6][0]), [c13] "+v"(acc[6][1]),                    \
      [c14] "+v"(acc[7][0]), [c15] "+v"(acc[7][1]), [a0] "+v"(af[0]), [a1] "+v"(af[1]), [a2] "+v"(af[2]),            \
      [a3] "+v"(af[3]), [a4] "+v"(af[4]), [a5] "+v"(af[5]), [a6] "+v"(af[6]), [a7] "+v"(af[7]),                      \
      [q0] "=&v"(bn[0]), [q1] "=&v"(bn[1]), [q2] "=&v"(bn[2]), [q3] "=&v"(bn[3]), [y] "=&v"(y)
#define WOQ_T_INPUTS                                                                                                 \
  [b0] "v"(bc0), [b1] "v"(bc1), [ad] "v"(ad), [ob] "n"(OB), [w] "v"(w), [ml] "s"(k.ml), [mh] "s"(k.mh),            \
      [gl] "v"(k.gl), [gh] "v"(k.gh), [r] "v"(f.r), [nl] "v"(f.nl), [nh] "v"(f.nh)
  if constexpr (DMA == 1 && !RAW && !(WOQ_T_KNOCK & 1)) {
    uint32_t keep;
    asm volatile("s_mov_b32 %[km], m0\n\t" WOQ_TBLOCK_TEXT(
                     P, "s_mov_b32 m0, %[m0a]\n\ts_nop 0\n\t" WOQ_DMA0("g0", 0), WOQ_DMA0("g0", 1024),
                     WOQ_DMA0("g0", 2048), WOQ_DMA0("g0", 3072), "s_mov_b32 m0, %[m0b]\n\ts_nop 0\n\t" WOQ_DMA0("g1", 0),
                     WOQ_DMA0("g1", 1024), WOQ_DMA0("g1", 2048), WOQ_DMA0("g1", 3072)) "s_mov_b32 m0, %[km]"
                 : WOQ_T_OPERANDS, [km] "=&s"(keep)
                 : WOQ_T_INPUTS, [gv] "v"(gv), [g0] "s"(g0), [g1] "s"(g1), [m0a] "s"(m0a), [m0b] "s"(m0b)
                 : "memory");
  } else if constexpr (DMA == 2 && !RAW && !(WOQ_T_KNOCK & 1)) {
    uint32_t keep;
    asm volatile("s_mov_b32 %[km], m0\n\t" WOQ_TBLOCK_TEXT(P, "s_mov_b32 m0, %[m0a]\n\ts_nop 0\n\t" WOQ_DMA0("g0", 0), "",
                                                           WOQ_DMA0("g0", 1024), "", WOQ_DMA0("g0", 2048), "",
                                                           WOQ_DMA0("g0", 3072), "") "s_mov_b32 m0, %[km]"
                 : WOQ_T_OPERANDS, [km] "=&s"(keep)
                 : WOQ_T_INPUTS, [gv] "v"(gv), [g0] "s"(g0), [m0a] "s"(m0a)
                 : "memory");
  } else if constexpr (DMA == 1 && RAW && !(WOQ_T_KNOCK & 1)) {
    uint32_t keep;
    asm volatile("s_mov_b32 %[km], m0\n\t" WOQ_TBLOCK_TEXT(R, "s_mov_b32 m0, %[m0a]\n\ts_nop 0\n\t" WOQ_T_DMAR(0, "0x400"),
                                                           WOQ_T_DMAR(1, "0x400"), WOQ_T_DMAR(2, "0x400"),
                                                           WOQ_T_DMAR(3, "0x3400"), WOQ_T_DMAR(4, "0x400"),
                                                           WOQ_T_DMAR(5, "0x400"), WOQ_T_DMAR(6, "0x400"),
                                                           WOQ_T_DMAR(7, "0x400")) "s_mov_b32 m0, %[km]"
                 : WOQ_T_OPERANDS, [z] "=&v"(z), [km] "=&s"(keep)
                 : WOQ_T_INPUTS, [s1] "s"(k.s1), [s2] "s"(k.s2), [g0] "s"(g0), [m0a] "s"(m0a), [gr0] "v"(ro.v[0]),
                   [gr1] "v"(ro.v[1]), [gr2] "v"(ro.v[2]), [gr3] "v"(ro.v[3]), [gr4] "v"(ro.v[4]), [gr5] "v"(ro.v[5]),
                   [gr6] "v"(ro.v[6]), [gr7] "v"(ro.v[7])
                 : "memory", "scc");
  } else if constexpr (RAW) {
    asm volatile(WOQ_TBLOCK_TEXT(R, "", "", "", "", "", "", "", "")
                 : WOQ_T_OPERANDS, [z] "=&v"(z)
                 : WOQ_T_INPUTS, [s1] "s"(k.s1), [s2] "s"(k.s2)
                 : "memory");
  } else {
    asm volatile(WOQ_TBLOCK_TEXT(P, "", "", "", "", "", "", "", "") : WOQ_T_OPERANDS : WOQ_T_INPUTS : "memory");
  }
#undef WOQ_T_OPERANDS
#undef WOQ_T_INPUTS
}

// ST: scale storage — 0 fp16, 1 bf16, 2 fp32. RAW: the caller's row-major fp16 matrix is the A operand (no pack pass).
// a.nb_m / a.n_sup / a.sup_n count 256-row blocks here; a.nb_m128 = 128-row blocks of the packed layout (the second
// image of the last workgroup may not exist: it then re-reads the last block's, and the epilogue masks rows >= M).
template <int SMODE, bool ASYM, int ST, bool RAW>
__global__ __launch_bounds__(256, 2) void gemm_f16t_kernel(GemmF16Args a) {
  constexpr bool S32 = ST == 2;
  constexpr int CT = 2, FBN = 128, NS = SMODE == 0 ? 1 : 2;
  constexpr int HT = FTILE_BYTES / 2;  // one 128-row half-tile image: 16 KiB
  constexpr int SLOT = 2 * HT;         // one 256-row half-tile: 32 KiB
  extern __shared__ __attribute__((aligned(1024))) unsigned char fsm[];  // 2 x 32 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kq = lane >> 4;
  const int bid = (int)blockIdx.x;  // XCD-aware placement, as gemm_f16s_kernel
  const int sup = ((bid >> 3) >> 6) * 8 + (bid & 7), within = (bid >> 3) & 63;
  if (sup >= a.n_sup) return;
  const int mb = (sup / a.sup_n) * 8 + (within >> 3), nb = (sup % a.sup_n) * 8 + (within & 7);
  if (mb >= a.nb_m || nb >= a.nb_n) return;
  const int row0 = mb * 256;
  const int ct0 = nb * (FBN / 16) + wid * CT;

  float4_t accA[8][CT], accB[8][CT];
#pragma unroll
  for (int rt = 0; rt < 8; ++rt)
#pragma unroll
    for (int c = 0; c < CT; ++c) accA[rt][c] = accB[rt][c] = (float4_t){0.f, 0.f, 0.f, 0.f};
  WOQ_PIN_EPILOGUE_ARGS(a)

  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)fsm;
  const int mbA = min(2 * mb, a.nb_m128 - 1), mbB = min(2 * mb + 1, a.nb_m128 - 1);
  const unsigned char* a_baseA =
      RAW ? (const unsigned char*)a.act_raw : (const unsigned char*)(a.ap + (size_t)mbA * a.tiles_k * (FTILE_BYTES / 2)) + wid * 4096;
  const unsigned char* a_baseB = RAW ? nullptr : (const unsigned char*)(a.ap + (size_t)mbB * a.tiles_k * (FTILE_BYTES / 2)) + wid * 4096;
  const uint32_t lane16 = lane * 16;
  const uint32_t dma_dst = lds0 + wid * 4096;  // + slot * SLOT (+ HT for the second image)
  auto srcA = [&](int q) { return a_baseA + (size_t)q * (RAW ? 128 : HT); };
  auto srcB = [&](int q) { return a_baseB + (size_t)q * HT; };
  RawOffsets ro;
#pragma unroll
  for (int j = 0; j < 8; ++j) {  // piece j: 8 rows x 128 B of image j >> 2; lane -> row lane / 8, slot lane % 8
    const int rl = (j >> 2) * 128 + wid * 32 + (j & 3) * 8 + (lane >> 3);
    ro.v[j] = RAW ? (uint32_t)min(row0 + rl, a.M - 1) * (uint32_t)(a.lda * 2) + (uint32_t)(((lane & 7) ^ ht_swz(rl)) << 4) : 0u;
  }
  auto issue_a = [&](int q, int slot) {  // a wave's whole share of a half-tile at once (prologue only)
    const uint32_t dst = dma_dst + slot * SLOT;
    if constexpr (RAW) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(ro.v[j]), "s"(srcA(q)), "s"(dst + (j >> 2) * HT + (j & 3) * 1024)
                     : "memory");
      }
    } else {
#pragma unroll
      for (int im = 0; im < 2; ++im) {
        uint32_t keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2\n\t"
            "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
            "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
            "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(lane16), "s"(im ? srcB(q) : srcA(q)), "s"(dst + im * HT)
            : "memory");
      }
    }
  };
  int tnc[CT], nexp[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    tnc[c] = min(ct0 + c, a.tiles_n - 1);
    nexp[c] = 1 - __builtin_amdgcn_frexp_expf(a.cs[tnc[c] * 16 + i16]);  // cs = 2^E: -E
  }
  const int lane_s = kq >> 1;
  struct BR {
    u32x4 wv[CT];
    u32x4 sc[CT];
    uint32_t zp[CT];
  };
  constexpr int SC_ESZ = S32 ? 4 : 2;
  const uint32_t sc_lane_k = SMODE == 0 ? i16 * SC_ESZ : i16 * 4 * SC_ESZ;
  const uint32_t zp_lane = SMODE == 0 ? i16 : i16 * 4;
  const int tpg = max(a.group >> 7, 1);
  int g_cnt = 0, g_idx = 0;
  auto load_b = [&](int kt, BR& b) {  // as gemm_f16p_kernel's
    const int g = g_idx;
    if constexpr (SMODE == 0) {
      if (++g_cnt == tpg) {
        g_cnt = 0;
        g_idx = min(g_idx + 1, a.n_groups - 1);
      }
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const size_t tile = (size_t)tnc[c] * a.tiles_k + kt;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b.wv[c]) : "v"(lane16), "s"(a.q + tile * 64) : "memory");
      const size_t s0 = SMODE == 0 ? ((size_t)tnc[c] * a.n_groups + g) * 16 : tile * 64;
      uint32_t sc_lane = sc_lane_k;
      if constexpr (ASYM && !S32) {
        sc_lane = zp_lane;
        asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(sc_lane));
      }
      const unsigned char* sp = (const unsigned char*)a.scales + s0 * SC_ESZ;
      if constexpr (SMODE == 0) {
        if constexpr (S32)
          asm volatile("global_load_dword %0, %1, %2" : "=v"(b.sc[c].x) : "v"(sc_lane), "s"(sp) : "memory");
        else
          asm volatile("global_load_ushort %0, %1, %2" : "=v"(b.sc[c].x) : "v"(sc_lane), "s"(sp) : "memory");
        if constexpr (ASYM)
          asm volatile("global_load_ubyte %0, %1, %2" : "=v"(b.zp[c]) : "v"(zp_lane), "s"(a.zp + s0) : "memory");
      } else {
        if constexpr (S32) {
          asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b.sc[c]) : "v"(sc_lane), "s"(sp) : "memory");
        } else {
          uint2 t;
          asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(t) : "v"(sc_lane), "s"(sp) : "memory");
          b.sc[c].x = t.x, b.sc[c].y = t.y;
        }
        if constexpr (ASYM)
          asm volatile("global_load_dword %0, %1, %2" : "=v"(b.zp[c]) : "v"(zp_lane), "s"(a.zp + s0) : "memory");
      }
    }
  };
  auto tie_b = [&](BR& b) {
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      asm volatile("" : "+v"(b.wv[c]));
      if constexpr (SMODE == 1 && S32) {
        asm volatile("" : "+v"(b.sc[c]));
      } else {
        asm volatile("" : "+v"(b.sc[c].x));
        if constexpr (SMODE == 1) asm volatile("" : "+v"(b.sc[c].y));
      }
      if constexpr (ASYM) asm volatile("" : "+v"(b.zp[c]));
    }
  };
  const uint32_t sh16 = 16 * lane_s;
  auto prep = [&](const BR& b, FragScale (&f)[CT][NS]) {  // as gemm_f16p_kernel's
    uint32_t sh8 = sh16;
    if constexpr (ASYM && SMODE == 1) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(sh8));
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        _Float16 rr;
        if constexpr (S32) {
          float sv = __builtin_bit_cast(float, b.sc[c].x);
          if constexpr (SMODE == 1) {
            const uint32_t lo = s == 0 ? b.sc[c].x : b.sc[c].z, hi = s == 0 ? b.sc[c].y : b.sc[c].w;
            sv = __builtin_bit_cast(float, lane_s ? hi : lo);
          }
          rr = (_Float16)__builtin_ldexpf(sv, nexp[c]);
        } else {
          uint32_t hw = b.sc[c].x;
          if constexpr (SMODE == 1) hw = (s == 0 ? b.sc[c].x : b.sc[c].y) >> sh16;
          if constexpr (ST == 1)
            rr = (_Float16)__builtin_ldexpf(bf16_bits_to_f32((uint16_t)hw), nexp[c]);
          else
            rr = __builtin_ldexpf16(__builtin_bit_cast(_Float16, (uint16_t)hw), nexp[c]);
        }
        f[c][s].r = (uint32_t)__builtin_bit_cast(uint16_t, rr);
        if constexpr (ASYM) {
          uint32_t uz = b.zp[c] & 0xffu;
          if constexpr (SMODE == 1) uz = (b.zp[c] >> (16 * s + sh8)) & 0xffu;
          f[c][s].nl = 0xE400u | uz;
          f[c][s].nh = 0xD400u | (uz << 4);
        } else {
          f[c][s].nl = 0xE408u;
          f[c][s].nh = 0xD480u;
        }
      }
  };
  PhaseConst pk;
  pk.ml = 0x000f000fu, pk.mh = 0x00f000f0u, pk.gl = 0x64086408u, pk.gh = 0x54805480u;
  pk.s1 = 0x0c010c00u, pk.s2 = 0x0c030c02u;
  asm volatile("" : "+v"(pk.gl), "+v"(pk.gh));
  h8 af[8];
  uint32_t bq[2][CT][4];
  BR B0, B1;
  FragScale F0[CT][NS], F1[CT][NS];
  auto bfrag = [&](int set, int c) { return (u32x4){bq[set][c][0], bq[set][c][1], bq[set][c][2], bq[set][c][3]}; };
  auto dequant_first = [&]() {  // part 0 of tile 0 by the compiler's hand
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      auto both = [](uint32_t lo16) { return __builtin_bit_cast(h2, (lo16 & 0xffffu) * 0x10001u); };
      u32x4 t = __builtin_bit_cast(u32x4, dq8s(B0.wv[c][0], both(F0[c][0].nl), both(F0[c][0].nh), both(F0[c][0].r)));
      if constexpr (RAW) {
        const uint32_t lo = 0x05040100u, hi = 0x07060302u;
        const u32x4 u = t;
        t.x = __builtin_amdgcn_perm(u.z, u.x, lo);
        t.y = __builtin_amdgcn_perm(u.z, u.x, hi);
        t.z = __builtin_amdgcn_perm(u.w, u.y, lo);
        t.w = __builtin_amdgcn_perm(u.w, u.y, hi);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) bq[0][c][i] = t[i];
    }
  };
  const int last = a.tiles_k - 1, qlast = 2 * a.tiles_k - 1;
  // A fragment of part p of a half-tile for row tile rt (0..15): chunk 2 kq + p of row (rt & 7) * 16 + i16 of image
  // rt >> 3 (128-B rows), slot = chunk ^ ht_swz(row); part 1's address is part 0's ^ 16
  const uint32_t a_adp0 = lds0 + i16 * 128 + (((2 * kq) ^ ht_swz(i16)) << 4);
  auto a_adp = [&](int pp, int slot) { return (a_adp0 + (uint32_t)slot * SLOT) ^ (uint32_t)(pp << 4); };

  // ---- prologue: half-tiles 0 and 1 in flight into slots 0 and 1, weights of tiles 0 and 1 ----
  issue_a(0, 0);
  load_b(0, B0);
  issue_a(min(1, qlast), 1);
  load_b(min(1, last), B1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  tie_b(B0);
  tie_b(B1);
  __syncthreads();
  prep(B0, F0);
#pragma unroll
  for (int rt = 0; rt < 8; ++rt) af[rt] = *(const h8*)(fsm + rt * 2048 + (a_adp0 - lds0));
  dequant_first();

#if WOQ_T_KNOCK & 8
#define WOQ_T_WAIT_BAR() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
#define WOQ_T_WAIT_BAR() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
#if WOQ_T_KNOCK & 2
#define WOQ_T_BARRIER()
#else
#define WOQ_T_BARRIER() __syncthreads();
#endif
  // One K step. Weight word j of a tile = its part j. bq[set]: set 0 holds the fragments of even parts, set 1 of odd.
#define WOQ_TSTEP_ALL(BCUR, FCUR, BNXT, FNXT, KT)                                                                           \
  {                                                                                                                    \
    const int q0 = 2 * (KT);                                                                                           \
    /* part 0 (slot 0): next = part 1, word 1 */                                                                        \
    gemm_tblock<HT, 0, RAW>(accA, af, bfrag(0, 0), bfrag(0, 1), bq[1][0], a_adp(0, 0), BCUR.wv[0][1], pk,           \
                                FCUR[0][0], 0, nullptr, nullptr, 0, 0, ro);                                             \
    gemm_tblock<0, 0, RAW>(accB, af, bfrag(0, 0), bfrag(0, 1), bq[1][1], a_adp(1, 0), BCUR.wv[1][1], pk,            \
                               FCUR[1][0], 0, nullptr, nullptr, 0, 0, ro);                                              \
    /* part 1 (slot 0): next = part 2 (slot 1), word 2 */                                                               \
    gemm_tblock<HT, 0, RAW>(accA, af, bfrag(1, 0), bfrag(1, 1), bq[0][0], a_adp(1, 0), BCUR.wv[0][2], pk,           \
                                FCUR[0][NS - 1], 0, nullptr, nullptr, 0, 0, ro);                                        \
    WOQ_T_WAIT_BAR() /* q1 (my pieces) landed; my reads of slot 0 done */                                               \
    tie_b(BNXT);                                                                                                        \
    WOQ_T_BARRIER() /* X: q1 complete, slot 0 free */                                                                   \
    gemm_tblock<0, 1, RAW>(accB, af, bfrag(1, 0), bfrag(1, 1), bq[0][1], a_adp(0, 1), BCUR.wv[1][2], pk,             \
                              FCUR[1][NS - 1], lane16, srcA(min(q0 + 2, qlast)), srcB(min(q0 + 2, qlast)), dma_dst,     \
                              dma_dst + HT, ro);                                                                        \
    /* part 2 (slot 1): next = part 3, word 3 */                                                                        \
    gemm_tblock<HT, 0, RAW>(accA, af, bfrag(0, 0), bfrag(0, 1), bq[1][0], a_adp(0, 1), BCUR.wv[0][3], pk,           \
                                FCUR[0][NS - 1], 0, nullptr, nullptr, 0, 0, ro);                                        \
    gemm_tblock<0, 0, RAW>(accB, af, bfrag(0, 0), bfrag(0, 1), bq[1][1], a_adp(1, 1), BCUR.wv[1][3], pk,            \
                               FCUR[1][NS - 1], 0, nullptr, nullptr, 0, 0, ro);                                         \
    /* part 3 (slot 1): next = part 0 of tile KT + 1 (slot 0), word 0 of BNXT */                                        \
    prep(BNXT, FNXT);                                                                                                   \
    gemm_tblock<HT, 0, RAW>(accA, af, bfrag(1, 0), bfrag(1, 1), bq[0][0], a_adp(1, 1), BNXT.wv[0][0], pk,           \
                                FNXT[0][0], 0, nullptr, nullptr, 0, 0, ro);                                             \
    WOQ_T_WAIT_BAR() /* q0 + 2 landed; my reads of slot 1 done */                                                       \
    WOQ_T_BARRIER() /* Y: q0 + 2 complete, slot 1 free */                                                               \
    load_b(min((KT) + 2, last), BCUR);                                                                                  \
    gemm_tblock<0, 1, RAW>(accB, af, bfrag(1, 0), bfrag(1, 1), bq[0][1], a_adp(0, 0), BNXT.wv[1][0], pk,             \
                              FNXT[1][0], lane16, srcA(min(q0 + 3, qlast)), srcB(min(q0 + 3, qlast)), dma_dst + SLOT,   \
                              dma_dst + SLOT + HT, ro);                                                                 \
  }
  // The same step with the LDS-DMA of a half-tile spread over TWO blocks, four pieces each, one per two pairs (packed
  // form): image A of q0 + 2 in P1B, image B in P2A; image A of q1 + 2 in P3B, image B in the NEXT step's P0A.
#define WOQ_TSTEP_SPLIT(BCUR, FCUR, BNXT, FNXT, KT)                                                                           \
  {                                                                                                                    \
    const int q0 = 2 * (KT);                                                                                           \
    /* part 0 (slot 0): next = part 1, word 1 */                                                                        \
    gemm_tblock<HT, 2, RAW>(accA, af, bfrag(0, 0), bfrag(0, 1), bq[1][0], a_adp(0, 0), BCUR.wv[0][1], pk,               \
                            FCUR[0][0], lane16, srcB(min(q0 + 1, qlast)), nullptr, dma_dst + SLOT + HT, 0, ro);         \
    gemm_tblock<0, 0, RAW>(accB, af, bfrag(0, 0), bfrag(0, 1), bq[1][1], a_adp(1, 0), BCUR.wv[1][1], pk,            \
                               FCUR[1][0], 0, nullptr, nullptr, 0, 0, ro);                                              \
    /* part 1 (slot 0): next = part 2 (slot 1), word 2 */                                                               \
    gemm_tblock<HT, 0, RAW>(accA, af, bfrag(1, 0), bfrag(1, 1), bq[0][0], a_adp(1, 0), BCUR.wv[0][2], pk,           \
                                FCUR[0][NS - 1], 0, nullptr, nullptr, 0, 0, ro);                                        \
    WOQ_T_WAIT_BAR() /* q1 (my pieces) landed; my reads of slot 0 done */                                               \
    tie_b(BNXT);                                                                                                        \
    WOQ_T_BARRIER() /* X: q1 complete, slot 0 free */                                                                   \
    gemm_tblock<0, 2, RAW>(accB, af, bfrag(1, 0), bfrag(1, 1), bq[0][1], a_adp(0, 1), BCUR.wv[1][2], pk,                \
                           FCUR[1][NS - 1], lane16, srcA(min(q0 + 2, qlast)), nullptr, dma_dst, 0, ro);                 \
    /* part 2 (slot 1): next = part 3, word 3 */                                                                        \
    gemm_tblock<HT, 2, RAW>(accA, af, bfrag(0, 0), bfrag(0, 1), bq[1][0], a_adp(0, 1), BCUR.wv[0][3], pk,               \
                            FCUR[0][NS - 1], lane16, srcB(min(q0 + 2, qlast)), nullptr, dma_dst + HT, 0, ro);           \
    gemm_tblock<0, 0, RAW>(accB, af, bfrag(0, 0), bfrag(0, 1), bq[1][1], a_adp(1, 1), BCUR.wv[1][3], pk,            \
                               FCUR[1][NS - 1], 0, nullptr, nullptr, 0, 0, ro);                                         \
    /* part 3 (slot 1): next = part 0 of tile KT + 1 (slot 0), word 0 of BNXT */                                        \
    prep(BNXT, FNXT);                                                                                                   \
    gemm_tblock<HT, 0, RAW>(accA, af, bfrag(1, 0), bfrag(1, 1), bq[0][0], a_adp(1, 1), BNXT.wv[0][0], pk,           \
                                FNXT[0][0], 0, nullptr, nullptr, 0, 0, ro);                                             \
    WOQ_T_WAIT_BAR() /* q0 + 2 landed; my reads of slot 1 done */                                                       \
    WOQ_T_BARRIER() /* Y: q0 + 2 complete, slot 1 free */                                                               \
    load_b(min((KT) + 2, last), BCUR);                                                                                  \
    gemm_tblock<0, 2, RAW>(accB, af, bfrag(1, 0), bfrag(1, 1), bq[0][1], a_adp(0, 0), BNXT.wv[1][0], pk,                \
                           FNXT[1][0], lane16, srcA(min(q0 + 3, qlast)), nullptr, dma_dst + SLOT, 0, ro);               \
  }
#ifndef WOQ_T_SPLIT_DMA  // A/B builds: tools/mkvariant_gemm.sh nosplit -DWOQ_T_SPLIT_DMA=0 (+1.1 % on the dominant GEMM, r06j)
#define WOQ_T_SPLIT_DMA 1
#endif
  if constexpr (WOQ_T_SPLIT_DMA && !RAW) {
    for (int kt = 0; kt < a.tiles_k; kt += 2) {  // tiles_k is even
      WOQ_TSTEP_SPLIT(B0, F0, B1, F1, kt)
      WOQ_TSTEP_SPLIT(B1, F1, B0, F0, kt + 1)
    }
  } else {
    for (int kt = 0; kt < a.tiles_k; kt += 2) {
      WOQ_TSTEP_ALL(B0, F0, B1, F1, kt)
      WOQ_TSTEP_ALL(B1, F1, B0, F0, kt + 1)
    }
  }
#undef WOQ_TSTEP_ALL
#undef WOQ_TSTEP_SPLIT
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // nothing may still be writing LDS when the workgroup retires
  WOQ_UNPIN_EPILOGUE_ARGS(a)
  gemm_epilogue<CT>(a, accA, row0, ct0, i16, kq);
  if (row0 + 128 < a.M) gemm_epilogue<CT>(a, accB, row0 + 128, ct0, i16, kq);
}
