// woq_gemm_f16p.h — the hand-scheduled K loop of the prefill GEMM (one fp16 product per operand pair, NP = 1).
// Included by woq_gemm_f16.hip after GemmF16Args, BRegs, dq8s and gemm_epilogue; same tiles, same operand formats,
// same epilogue as gemm_f16s_kernel there — only the order of the instructions inside a K step is different.
//
// Why. hipcc's schedule of gemm_f16s_kernel reads two A fragments, waits for them, issues four MFMAs, and repeats:
// sixteen exposed LDS round trips per K step (SQ_WAIT_ANY 33 % of the wave cycles, matrix pipe 48 % busy,
// profiles/r02t_gemm_counters.txt), and no source-level arrangement survived its scheduler. Here a K step is four
// asm blocks, one per 32-k part: part p's sixteen MFMAs (eight row-tile pairs) are interleaved with the eight
// ds_read_b128 of part p + 1's A fragments and the 26 VALU operations that dequantise part p + 1's two weight
// fragments. The A fragments live in ONE set of eight registers: fragment i is refilled with the next part's right
// after its two MFMAs have issued (hipcc's own schedule does the same), and the waits are counted (LDS returns in
// order), so six or seven reads stay in flight. The workgroup barrier sits once per K step between parts 2 and 3:
// behind it tile kt + 1 is complete in the other LDS buffer (part 3 already prefetches from it) and every wave is done
// reading this one, so the LDS-DMA of tile kt + 2 is issued from inside part 3's block, one 1-KiB piece per row-tile
// pair, a whole K step ahead of its use. The only vmcnt wait is the one before part 2, for loads issued a K step
// earlier. Registers are allocated by the compiler (operands); the instruction order inside a block is fixed by hand.
// Two further forms of the same loop, selected by the launcher: RAW (fp16 activation rows fetched straight from the
// caller's matrix, no pack pass) and RING (the A operand in 16-KiB half-tiles through three LDS slots: three
// workgroups per CU) — described at gemm_f16p_kernel below.
// gfx950, M = 8192 x K = 4096 x N = 22016, g128 sym, rocprofv3 kernel time: hipcc's schedule 1.40 ms, this loop 1.16,
// with the ring 1.10 = 0.54 of 2.5 PF (profiles/r01m_prefill_kernel_stats.txt, r02z_*, r02zz_bench_kernel_stats.txt).
#pragma once
// (included inside namespace woq)

#define WOQ_S_(x) #x
#define WOQ_MF(ci, ai, bi) \
  "v_mfma_f32_16x16x32_f16 %[c" WOQ_S_(ci) "], %[a" WOQ_S_(ai) "], %[b" WOQ_S_(bi) "], %[c" WOQ_S_(ci) "]\n\t"
// refill A fragment register i with the NEXT part's fragment (its two MFMAs of this part have been issued)
#define WOQ_RD_T(i) "ds_read_b128 %[a" WOQ_S_(i) "], %[ad] offset:%[ob]+4096*" WOQ_S_(i) "\n\t"  // 32-KiB tiles: 256-B rows
#define WOQ_RD_H(i) "ds_read_b128 %[a" WOQ_S_(i) "], %[ad] offset:%[ob]+2048*" WOQ_S_(i) "\n\t"  // 16-KiB half-tiles: 128-B rows
// dequantisation of fragment f: (w & mask) ^ magic -> + n -> * r   (see dq8s; bitop3 0x6c is (src0 & src2) ^ src1).
// Two element orders. P (packed A tiles, whose 16-byte chunks hold k in the order 0 2 4 6 1 3 5 7): the word and the
// word >> 8 give the pairs (n0 n4)(n1 n5)(n2 n6)(n3 n7), 13 operations. R (raw row-major A, k in natural order): two
// byte permutes [B0 0 B1 0], [B2 0 B3 0] give (n0 n2)(n4 n6)(n1 n3)(n5 n7) = k (0 1)(2 3)(4 5)(6 7) — the blob keeps
// k -> nibble 0 2 4 6 1 3 5 7 (include/woq_blob.h) — 14 operations.
#define WOQ_DH_P(f) "v_lshrrev_b32 %[y" WOQ_S_(f) "], 8, %[w" WOQ_S_(f) "]\n\t"
#define WOQ_DH_R(f)                                                                         \
  "v_perm_b32 %[y" WOQ_S_(f) "], %[w" WOQ_S_(f) "], %[w" WOQ_S_(f) "], %[s1]\n\t"           \
  "v_perm_b32 %[z" WOQ_S_(f) "], %[w" WOQ_S_(f) "], %[w" WOQ_S_(f) "], %[s2]\n\t"
#define WOQ_DB(f, k, src, m, g) \
  "v_bitop3_b32 %[q" WOQ_S_(f) WOQ_S_(k) "], %[" src WOQ_S_(f) "], %[" g "], %[" m "] bitop3:0x6c\n\t"
#define WOQ_DB_P0(f) WOQ_DB(f, 0, "w", "ml", "gl")
#define WOQ_DB_P1(f) WOQ_DB(f, 1, "w", "mh", "gh")
#define WOQ_DB_P2(f) WOQ_DB(f, 2, "y", "ml", "gl")
#define WOQ_DB_P3(f) WOQ_DB(f, 3, "y", "mh", "gh")
#define WOQ_DB_R0(f) WOQ_DB(f, 0, "y", "ml", "gl")
#define WOQ_DB_R1(f) WOQ_DB(f, 1, "z", "ml", "gl")
#define WOQ_DB_R2(f) WOQ_DB(f, 2, "y", "mh", "gh")
#define WOQ_DB_R3(f) WOQ_DB(f, 3, "z", "mh", "gh")
#define WOQ_DA(f, k, n) \
  "v_pk_add_f16 %[q" WOQ_S_(f) WOQ_S_(k) "], %[q" WOQ_S_(f) WOQ_S_(k) "], %[" n WOQ_S_(f) "] op_sel_hi:[1,0]\n\t"
#define WOQ_DA_P0(f) WOQ_DA(f, 0, "nl")
#define WOQ_DA_P1(f) WOQ_DA(f, 1, "nh")
#define WOQ_DA_P2(f) WOQ_DA(f, 2, "nl")
#define WOQ_DA_P3(f) WOQ_DA(f, 3, "nh")
#define WOQ_DA_R0(f) WOQ_DA(f, 0, "nl")
#define WOQ_DA_R1(f) WOQ_DA(f, 1, "nl")
#define WOQ_DA_R2(f) WOQ_DA(f, 2, "nh")
#define WOQ_DA_R3(f) WOQ_DA(f, 3, "nh")
#define WOQ_DM(f, k) \
  "v_pk_mul_f16 %[q" WOQ_S_(f) WOQ_S_(k) "], %[q" WOQ_S_(f) WOQ_S_(k) "], %[r" WOQ_S_(f) "] op_sel_hi:[1,0]\n\t"
#define WOQ_DMA0(g, off) "global_load_lds_dwordx4 %[gv], %[" g "] offset:" WOQ_S_(off) "\n\t"
#define WOQ_W7 "s_waitcnt lgkmcnt(7)\n\t"
#define WOQ_W6 "s_waitcnt lgkmcnt(6)\n\t"
// Row-tile pair i of a part: wait for A fragment i (LDS returns in order: the fragments read after it — 7 - i of the
// previous block, i - 1 of this one — may still be in flight, so the count is 7 for pair 0 and 6 after), its two
// MFMAs with the refill of fragment i - 1 between them, then three or four of the 26 VALU operations. X0 .. X7: the
// LDS-DMA pieces in the part-3 block, nothing elsewhere.
#define WOQ_PHASE_TEXT(V, L, X0, X1, X2, X3, X4, X5, X6, X7)                                                        \
  WOQ_W7 WOQ_MF(0, 0, 0) WOQ_MF(1, 0, 1) X0 WOQ_DH_##V(0) WOQ_DH_##V(1) WOQ_DB_##V##0(0)                            \
  WOQ_W6 WOQ_MF(2, 1, 0) WOQ_RD_##L(0) WOQ_MF(3, 1, 1) X1 WOQ_DB_##V##1(0) WOQ_DB_##V##2(0) WOQ_DB_##V##3(0)        \
  WOQ_W6 WOQ_MF(4, 2, 0) WOQ_RD_##L(1) WOQ_MF(5, 2, 1) X2 WOQ_DB_##V##0(1) WOQ_DB_##V##1(1) WOQ_DB_##V##2(1)        \
  WOQ_W6 WOQ_MF(6, 3, 0) WOQ_RD_##L(2) WOQ_MF(7, 3, 1) X3 WOQ_DB_##V##3(1) WOQ_DA_##V##0(0) WOQ_DA_##V##1(0)        \
  WOQ_W6 WOQ_MF(8, 4, 0) WOQ_RD_##L(3) WOQ_MF(9, 4, 1) X4 WOQ_DA_##V##2(0) WOQ_DA_##V##3(0) WOQ_DA_##V##0(1)        \
  WOQ_W6 WOQ_MF(10, 5, 0) WOQ_RD_##L(4) WOQ_MF(11, 5, 1) X5 WOQ_DA_##V##1(1) WOQ_DA_##V##2(1) WOQ_DA_##V##3(1)      \
  WOQ_W6 WOQ_MF(12, 6, 0) WOQ_RD_##L(5) WOQ_MF(13, 6, 1) X6 WOQ_DM(0, 0) WOQ_DM(0, 1) WOQ_DM(0, 2) WOQ_DM(0, 3)     \
  WOQ_W6 WOQ_MF(14, 7, 0) WOQ_RD_##L(6) WOQ_MF(15, 7, 1) X7 WOQ_DM(1, 0) WOQ_DM(1, 1) WOQ_DM(1, 2) WOQ_DM(1, 3)     \
      WOQ_RD_##L(7)
// one piece of a raw-A tile: its own lane offsets (row and swizzled chunk), the LDS address steps by 1 KiB in M0
#define WOQ_DMAR(j) "global_load_lds_dwordx4 %[gr" WOQ_S_(j) "], %[g0]\n\ts_add_u32 m0, m0, 0x400\n\t"

// Chunk swizzle of the half-tile layout (128-byte rows, 8 chunks): chunk c of row r sits in slot c ^ ht_swz(r). A
// ds_read_b128 is served in 16-lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... — lane quarters kq, kq ^ 1
// with complementary row sets), each of which has to cover the 64 banks once: with g = (r >> 1) & 7 the rows of one
// quarter have g in S = {0, 1, 6, 7}, the other's in S ^ 2, their chunks differ by 2, and f(g) = (g & 3) | ((g1 ^ g2) << 2)
// is a bijection that maps S onto {0..3}, so f(S) and f(S ^ 2) ^ 2 together are all eight slots (and a plain 16-lane
// grouping is conflict-free too, f being a bijection).
__host__ __device__ inline int ht_swz(int row) {
  const int g = (row >> 1) & 7;
  return (g & 3) | ((((g >> 1) ^ (g >> 2)) & 1) << 2);
}

struct PhaseConst {  // loop-invariant operands of the dequantisation
  uint32_t ml, mh;   // nibble masks (SGPR)
  uint32_t gl, gh;   // magic words (VGPR: one constant-bus operand per VOP3)
  uint32_t s1, s2;   // byte-permute selectors of the raw-A order (SGPR)
};
struct RawOffsets {  // raw-A: lane offset (row * lda * 2 + swizzled chunk * 16) of each of a wave's eight pieces
  uint32_t v[8];
};
struct FragScale {  // per (column tile, scale slot): fp16 values in the low halves
  uint32_t r, nl, nh;
};

// one 32-k part. af: this part's A fragments on entry (possibly still in flight), the next part's on exit (in
// flight), read from LDS address `ad` + immediate OB + 4096 rt; bc: this part's weight fragments; bn: the next
// part's, dequantised here from the blob words w0 / w1.
template <int OB, int DMA, bool RAW, bool RING = false>  // DMA: 0 none, 1 this block carries LDS-DMA pieces (8 of a tile, 4 of a half-tile)
__device__ __forceinline__ void gemm_phase(float4_t (&acc)[8][2], h8 (&af)[8], const u32x4& bc0, const u32x4& bc1,
                                           uint32_t (&bn)[2][4], uint32_t ad, uint32_t w0, uint32_t w1,
                                           const PhaseConst& k, const FragScale& f0, const FragScale& f1,
                                           uint32_t gv, const void* g0, const void* g1, uint32_t m0a, uint32_t m0b,
                                           const RawOffsets& ro) {
  uint32_t y0, y1, z0, z1;
#define WOQ_PHASE_OPERANDS                                                                                           \
  [c0] "+v"(acc[0][0]), [c1] "+v"(acc[0][1]), [c2] "+v"(acc[1][0]), [c3] "+v"(acc[1][1]), [c4] "+v"(acc[2][0]),      \
      [c5] "+v"(acc[2][1]), [c6] "+v"(acc[3][0]), [c7] "+v"(acc[3][1]), [c8] "+v"(acc[4][0]), [c9] "+v"(acc[4][1]),  \
      [c10] "+v"(acc[5][0]), [c11] "+v"(acc[5][1]), [c12] "+v"(acc[6][0]), [c13] "+v"(acc[6][1]),                    \
      [c14] "+v"(acc[7][0]), [c15] "+v"(acc[7][1]), [a0] "+v"(af[0]), [a1] "+v"(af[1]), [a2] "+v"(af[2]),            \
      [a3] "+v"(af[3]), [a4] "+v"(af[4]), [a5] "+v"(af[5]), [a6] "+v"(af[6]), [a7] "+v"(af[7]),                      \
      [q00] "=&v"(bn[0][0]), [q01] "=&v"(bn[0][1]), [q02] "=&v"(bn[0][2]), [q03] "=&v"(bn[0][3]),                    \
      [q10] "=&v"(bn[1][0]), [q11] "=&v"(bn[1][1]), [q12] "=&v"(bn[1][2]), [q13] "=&v"(bn[1][3]), [y0] "=&v"(y0),    \
      [y1] "=&v"(y1)
#define WOQ_PHASE_INPUTS                                                                                             \
  [b0] "v"(bc0), [b1] "v"(bc1), [ad] "v"(ad), [ob] "n"(OB), [w0] "v"(w0), [w1] "v"(w1), [ml] "s"(k.ml),             \
      [mh] "s"(k.mh), [gl] "v"(k.gl), [gh] "v"(k.gh), [r0] "v"(f0.r), [nl0] "v"(f0.nl), [nh0] "v"(f0.nh),            \
      [r1] "v"(f1.r), [nl1] "v"(f1.nl), [nh1] "v"(f1.nh)
#define WOQ_RAW_OUT , [z0] "=&v"(z0), [z1] "=&v"(z1)
#define WOQ_RAW_IN , [s1] "s"(k.s1), [s2] "s"(k.s2)
  if constexpr (RING) {
    if constexpr (DMA == 1 && !RAW) {
      uint32_t keep;
      asm volatile("s_mov_b32 %[km], m0\n\t" WOQ_PHASE_TEXT(P, H, "s_mov_b32 m0, %[m0a]\n\ts_nop 0\n\t" WOQ_DMA0("g0", 0),
                                                            WOQ_DMA0("g0", 1024), WOQ_DMA0("g0", 2048), WOQ_DMA0("g0", 3072),
                                                            "", "", "", "") "s_mov_b32 m0, %[km]"
                   : WOQ_PHASE_OPERANDS, [km] "=&s"(keep)
                   : WOQ_PHASE_INPUTS, [gv] "v"(gv), [g0] "s"(g0), [m0a] "s"(m0a)
                   : "memory");
    } else if constexpr (DMA == 1 && RAW) {
      uint32_t keep;
      asm volatile("s_mov_b32 %[km], m0\n\t" WOQ_PHASE_TEXT(R, H, "s_mov_b32 m0, %[m0a]\n\ts_nop 0\n\t" WOQ_DMAR(0),
                                                            WOQ_DMAR(1), WOQ_DMAR(2), WOQ_DMAR(3), "", "", "", "")
                   "s_mov_b32 m0, %[km]"
                   : WOQ_PHASE_OPERANDS WOQ_RAW_OUT, [km] "=&s"(keep)
                   : WOQ_PHASE_INPUTS WOQ_RAW_IN, [g0] "s"(g0), [m0a] "s"(m0a), [gr0] "v"(ro.v[0]), [gr1] "v"(ro.v[1]),
                     [gr2] "v"(ro.v[2]), [gr3] "v"(ro.v[3])
                   : "memory", "scc");
    } else if constexpr (RAW) {
      asm volatile(WOQ_PHASE_TEXT(R, H, "", "", "", "", "", "", "", "")
                   : WOQ_PHASE_OPERANDS WOQ_RAW_OUT
                   : WOQ_PHASE_INPUTS WOQ_RAW_IN
                   : "memory");
    } else {
      asm volatile(WOQ_PHASE_TEXT(P, H, "", "", "", "", "", "", "", "") : WOQ_PHASE_OPERANDS : WOQ_PHASE_INPUTS : "memory");
    }
  } else if constexpr (DMA == 1 && !RAW) {
    uint32_t keep;
    asm volatile("s_mov_b32 %[km], m0\n\t" WOQ_PHASE_TEXT(
                     P, T, "s_mov_b32 m0, %[m0a]\n\ts_nop 0\n\t" WOQ_DMA0("g0", 0), WOQ_DMA0("g0", 1024),
                     WOQ_DMA0("g0", 2048), WOQ_DMA0("g0", 3072), "s_mov_b32 m0, %[m0b]\n\ts_nop 0\n\t" WOQ_DMA0("g1", 0),
                     WOQ_DMA0("g1", 1024), WOQ_DMA0("g1", 2048), WOQ_DMA0("g1", 3072)) "s_mov_b32 m0, %[km]"
                 : WOQ_PHASE_OPERANDS, [km] "=&s"(keep)
                 : WOQ_PHASE_INPUTS, [gv] "v"(gv), [g0] "s"(g0), [g1] "s"(g1), [m0a] "s"(m0a), [m0b] "s"(m0b)
                 : "memory");
  } else if constexpr (DMA == 1 && RAW) {
    uint32_t keep;
    asm volatile("s_mov_b32 %[km], m0\n\t" WOQ_PHASE_TEXT(R, T, "s_mov_b32 m0, %[m0a]\n\ts_nop 0\n\t" WOQ_DMAR(0),
                                                          WOQ_DMAR(1), WOQ_DMAR(2), WOQ_DMAR(3), WOQ_DMAR(4), WOQ_DMAR(5),
                                                          WOQ_DMAR(6), WOQ_DMAR(7)) "s_mov_b32 m0, %[km]"
                 : WOQ_PHASE_OPERANDS WOQ_RAW_OUT, [km] "=&s"(keep)
                 : WOQ_PHASE_INPUTS WOQ_RAW_IN, [g0] "s"(g0), [m0a] "s"(m0a), [gr0] "v"(ro.v[0]), [gr1] "v"(ro.v[1]),
                   [gr2] "v"(ro.v[2]), [gr3] "v"(ro.v[3]), [gr4] "v"(ro.v[4]), [gr5] "v"(ro.v[5]), [gr6] "v"(ro.v[6]),
                   [gr7] "v"(ro.v[7])
                 : "memory", "scc");
  } else if constexpr (RAW) {
    asm volatile(WOQ_PHASE_TEXT(R, T, "", "", "", "", "", "", "", "")
                 : WOQ_PHASE_OPERANDS WOQ_RAW_OUT
                 : WOQ_PHASE_INPUTS WOQ_RAW_IN
                 : "memory");
  } else {
    asm volatile(WOQ_PHASE_TEXT(P, T, "", "", "", "", "", "", "", "") : WOQ_PHASE_OPERANDS : WOQ_PHASE_INPUTS : "memory");
  }
#undef WOQ_PHASE_OPERANDS
#undef WOQ_PHASE_INPUTS
#undef WOQ_RAW_OUT
#undef WOQ_RAW_IN
}

// ST: scale storage — 0 fp16, 1 bf16, 2 fp32. RAW: the A operand is the caller's row-major fp16 matrix itself
// (a.act_raw, a.lda; no pack pass, row scales all 1): every lane of an LDS-DMA piece fetches the 16-byte chunk that
// belongs at its LDS position — row 4 j + lane / 16 of the wave's 32, chunk (lane % 16) ^ (row % 16) — so the tile
// lands in the same swizzled image the packed tiles have; the k order inside a chunk is the natural one, which the
// weight side answers with its R order (above).
//
// RING: the A operand moves in 64-k HALF-tiles of 16 KiB through a ring of three LDS slots (48 KiB per workgroup:
// three workgroups per CU where the registers allow it, i.e. <= 168 VGPRs) instead of two 32-KiB tiles. Half-tile q
// (sequence number; its slot is q % 3) is issued 3 half-tiles ahead, from inside the second block of half-tile q - 3,
// whose slot every wave has released at the barrier in front of that block; one barrier per half-tile (two per K
// step), each with its counted vmcnt wait one block earlier.
template <int SMODE, bool ASYM, int ST, bool RAW = false, bool RING = false>
__global__ __launch_bounds__(256, RING ? 3 : 2) void gemm_f16p_kernel(GemmF16Args a) {
  constexpr bool S32 = ST == 2;
  constexpr int CT = 2, STAGE = FTILE_BYTES, FBN = 128, NS = SMODE == 0 ? 1 : 2;
  constexpr int HT = FTILE_BYTES / 2;  // half-tile bytes
  extern __shared__ __attribute__((aligned(1024))) unsigned char fsm[];  // 2 x 32 KiB A tiles (RING: 3 x 16 KiB)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kq = lane >> 4;
  const int bid = (int)blockIdx.x;  // XCD-aware placement, as gemm_f16s_kernel
  const int sup = ((bid >> 3) >> 6) * 8 + (bid & 7), within = (bid >> 3) & 63;
  if (sup >= a.n_sup) return;
  const int mb = (sup / a.sup_n) * 8 + (within >> 3), nb = (sup % a.sup_n) * 8 + (within & 7);
  if (mb >= a.nb_m || nb >= a.nb_n) return;
  const int row0 = mb * FBM;
  const int ct0 = nb * (FBN / 16) + wid * CT;

  float4_t acc[8][CT];
#pragma unroll
  for (int rt = 0; rt < 8; ++rt)
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[rt][c] = (float4_t){0.f, 0.f, 0.f, 0.f};
  WOQ_PIN_EPILOGUE_ARGS(a)

  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)fsm;
  // Every address of the K loop is a wave-uniform 64-bit base (SGPR pair, advanced by scalar adds) plus a lane
  // offset that never changes (VGPR): no vector address arithmetic per K step.
  const unsigned char* a_base =
      RAW ? (const unsigned char*)a.act_raw
          : (const unsigned char*)(a.ap + (size_t)mb * a.tiles_k * (FTILE_BYTES / 2)) + wid * (RING ? 4096 : 8192);
  const uint32_t lane16 = lane * 16;
  const uint32_t dma_dst = lds0 + wid * (RING ? 4096 : 8192);  // + buf * STAGE (RING: + slot * HT)
  // source of tile kt (RING: of half-tile kt)
  auto a_src = [&](int kt) { return a_base + (size_t)kt * (RAW ? (RING ? 128 : 256) : (RING ? HT : FTILE_BYTES)); };
  RawOffsets ro;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if constexpr (RING) {  // piece j < 4 of a half-tile: 8 rows x 128 B, lane -> row lane / 8, slot lane % 8
      const int rl = wid * 32 + (j & 3) * 8 + (lane >> 3);
      ro.v[j] = RAW ? (uint32_t)min(row0 + rl, a.M - 1) * (uint32_t)(a.lda * 2) + (uint32_t)(((lane & 7) ^ ht_swz(rl)) << 4)
                    : 0u;
    } else {
      const int rl = wid * 32 + j * 4 + kq;  // row of the tile this lane fetches in piece j (rows past M: the last row)
      ro.v[j] = RAW ? (uint32_t)min(row0 + rl, a.M - 1) * (uint32_t)(a.lda * 2) + (uint32_t)((i16 ^ (rl & 15)) << 4) : 0u;
    }
  }
  auto issue_a = [&](int kt, int buf) {  // a wave's whole share of a tile (RING: of a half-tile) at once (prologue only)
    constexpr int NPC = RING ? 4 : 8;
    const uint32_t dst = dma_dst + buf * (RING ? HT : STAGE);
    if constexpr (RAW) {
#pragma unroll
      for (int j = 0; j < NPC; ++j) {
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(ro.v[j]), "s"(a_src(kt)), "s"(dst + j * 1024)
                     : "memory");
      }
    } else {
#pragma unroll
      for (int j = 0; j < NPC; j += 4) {
        uint32_t keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2\n\t"
            "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
            "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
            "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(lane16), "s"(a_src(kt) + j * 1024), "s"(dst + j * 1024)
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
  const int lane_s = kq >> 1;  // group-32: which 32-k group of a 64-k half this lane quarter belongs to
  // One K step of this wave's weight operand as loaded. Scales / zero points of a (column, K step): group >= 128 one
  // value; group 32 / 64 the four values of the column in ONE load (fp32: 16 bytes, 16-bit: 8, zero points: 4) —
  // the lane picks its two (64-k half h, lane_s) in prep.
  struct BR {
    u32x4 wv[CT];
    u32x4 sc[CT];
    uint32_t zp[CT];
  };
  constexpr int SC_ESZ = S32 ? 4 : 2;
  const uint32_t sc_lane_k = SMODE == 0 ? i16 * SC_ESZ : i16 * 4 * SC_ESZ;  // byte offset of the lane's column
  const uint32_t zp_lane = SMODE == 0 ? i16 : i16 * 4;
  // scale group of the tile being loaded, kept incrementally (tiles are loaded in order 0, 1, 2, ...; the repeats of
  // the last tile at the end of the loop stay in the last group)
  const int tpg = max(a.group >> 7, 1);
  int g_cnt = 0, g_idx = 0;
  auto load_b = [&](int kt, BR& b) {
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
      const size_t s0 = SMODE == 0 ? ((size_t)tnc[c] * a.n_groups + g) * 16 : tile * 64;  // first scale of the block
      uint32_t sc_lane = sc_lane_k;
      if constexpr (ASYM && !S32) {  // scale offset = 2 x zero-point offset: formed here, one register across the loop
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
  auto tie_b = [&](BR& b) {  // the weight registers' first use depends on the s_waitcnt in front of this
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
  // r = scale * 2^-E as fp16, n = -(1024 + zp) and -(64 + zp) as fp16 BITS (1024 + u is 0x6400 + u, 64 + u is
  // 0x5400 + 16 u): low halves only, the block's packed operations broadcast them (op_sel_hi)
  const uint32_t sh16 = 16 * lane_s;
  auto prep = [&](const BR& b, FragScale (&f)[CT][NS]) {
    uint32_t sh8 = sh16;
    if constexpr (ASYM && SMODE == 1) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(sh8));  // (not kept across the loop)
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
          f[c][s].nl = 0xE408u;  // -(1024 + 8)
          f[c][s].nh = 0xD480u;  // -(64 + 8)
        }
      }
  };
  PhaseConst pk;
  pk.ml = 0x000f000fu, pk.mh = 0x00f000f0u, pk.gl = 0x64086408u, pk.gh = 0x54805480u;
  pk.s1 = 0x0c010c00u, pk.s2 = 0x0c030c02u;  // bytes [B0 0 B1 0], [B2 0 B3 0] of the blob word
  asm volatile("" : "+v"(pk.gl), "+v"(pk.gh));  // keep the magic words in registers
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
      if constexpr (RAW) {  // dq8s gives nibbles (0 4)(1 5)(2 6)(3 7); the raw order wants (0 2)(4 6)(1 3)(5 7)
        const uint32_t lo = 0x05040100u, hi = 0x07060302u;  // v_perm: low halves / high halves of two words
        const u32x4 u = t;
        t.x = __builtin_amdgcn_perm(u.z, u.x, lo);  // (n0 n2)
        t.y = __builtin_amdgcn_perm(u.z, u.x, hi);  // (n4 n6)
        t.z = __builtin_amdgcn_perm(u.w, u.y, lo);  // (n1 n3)
        t.w = __builtin_amdgcn_perm(u.w, u.y, hi);  // (n5 n7)
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) bq[0][c][i] = t[i];
    }
  };
  const int last = a.tiles_k - 1;

  if constexpr (RING) {
    // A fragment of part p of a half-tile for row tile rt: chunk 2 kq + p (k = 16 kq + 8 p .. + 8 of the 64) of row
    // rt*16 + i16 (128-B rows), slot = chunk ^ ht_swz(row)
    // (part 1's chunk is part 0's ^ 1: the address differs in bit 4 only, and is formed per block from part 0's so that
    // one register holds both — the ring kernels sit at the 168-VGPR edge of three waves per SIMD)
    const uint32_t a_adp0 = lds0 + i16 * 128 + (((2 * kq) ^ ht_swz(i16)) << 4);
    auto a_adp = [&](int pp, int slot) { return (a_adp0 + (uint32_t)slot * HT) ^ (uint32_t)(pp << 4); };
    const int qlast = 2 * a.tiles_k - 1;
    constexpr int NB = 2 * (2 + (ASYM ? 1 : 0));  // VMEM instructions of one load_b
    // ---- prologue: half-tiles 0..2 in flight into slots 0..2, weights of tiles 0 and 1 ----
    issue_a(0, 0);
    load_b(0, B0);
    issue_a(min(1, qlast), 1);
    issue_a(min(2, qlast), 2);
    load_b(min(1, last), B1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tie_b(B0);
    tie_b(B1);
    __syncthreads();
    prep(B0, F0);
#pragma unroll
    for (int rt = 0; rt < 8; ++rt) af[rt] = *(const h8*)(fsm + rt * 2048 + (a_adp0 - lds0));
    dequant_first();
    int s0 = 0;  // slot of the current K step's first half-tile: (2 kt) % 3
    // One K step = half-tiles q0 = 2 kt (parts 0, 1) and q1 = 2 kt + 1 (parts 2, 3) in slots s0, s1; s2 holds 2 kt + 2.
    // VMEM issue order per step: [P1] 4 LDS-DMA pieces of half-tile q0 + 3 -> s0, [behind Y] NB weight loads of tile
    // kt + 2, [P3] 4 pieces of half-tile q1 + 3 -> s1. The waits are counted against that order.
#define WOQ_RSTEP(BCUR, FCUR, BNXT, FNXT, KT)                                                                         \
  {                                                                                                                   \
    const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s1 == 2 ? 0 : s1 + 1;                                                   \
    const int q0 = 2 * (KT);                                                                                          \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB + 4) : "memory"); /* half-tile q1 (issued in the previous P1) */       \
    gemm_phase<0, 0, RAW, true>(acc, af, bfrag(0, 0), bfrag(0, 1), bq[1], a_adp(1, s0), BCUR.wv[0][1],          \
                                BCUR.wv[1][1], pk, FCUR[0][0], FCUR[1][0], 0, nullptr, nullptr, 0, 0, ro);            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* my reads of slot s0 are done */                             \
    __syncthreads();                                   /* X: q1 is complete, s0 is free */                            \
    gemm_phase<0, 1, RAW, true>(acc, af, bfrag(1, 0), bfrag(1, 1), bq[0], a_adp(0, s1), BCUR.wv[0][2],          \
                                BCUR.wv[1][2], pk, FCUR[0][NS - 1], FCUR[1][NS - 1], lane16, a_src(min(q0 + 3, qlast)), \
                                nullptr, dma_dst + s0 * HT, 0, ro);                                                   \
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); /* tile KT + 1's weights and half-tile q0 + 2 */                 \
    tie_b(BNXT);                                                                                                      \
    gemm_phase<0, 0, RAW, true>(acc, af, bfrag(0, 0), bfrag(0, 1), bq[1], a_adp(1, s1), BCUR.wv[0][3],          \
                                BCUR.wv[1][3], pk, FCUR[0][NS - 1], FCUR[1][NS - 1], 0, nullptr, nullptr, 0, 0, ro);  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* my reads of slot s1 are done */                             \
    __syncthreads();                                   /* Y: q0 + 2 is complete, s1 is free */                        \
    tie_b(BNXT); /* (keeps the next line behind part 2: FCUR and FNXT then never live together) */                    \
    prep(BNXT, FNXT);                                                                                                 \
    load_b(min((KT) + 2, last), BCUR);                                                                                \
    gemm_phase<0, 1, RAW, true>(acc, af, bfrag(1, 0), bfrag(1, 1), bq[0], a_adp(0, s2), BNXT.wv[0][0],          \
                                BNXT.wv[1][0], pk, FNXT[0][0], FNXT[1][0], lane16, a_src(min(q0 + 4, qlast)), nullptr, \
                                dma_dst + s1 * HT, 0, ro);                                                            \
    s0 = s2;                                                                                                          \
  }
    for (int kt = 0; kt < a.tiles_k; kt += 2) {  // tiles_k is even
      WOQ_RSTEP(B0, F0, B1, F1, kt)
      WOQ_RSTEP(B1, F1, B0, F0, kt + 1)
    }
#undef WOQ_RSTEP
  } else {
  // A fragment of (64-k half h, part p) for row tile rt: chunk h*8 + kq*2 + p of row rt*16 + i16, slot = chunk ^ i16
  uint32_t a_ad[4];
#pragma unroll
  for (int hp = 0; hp < 4; ++hp) a_ad[hp] = lds0 + i16 * 256 + ((((hp >> 1) * 8 + kq * 2 + (hp & 1)) ^ i16) << 4);

  // ---- prologue: tiles 0 and 1 in flight, part 0 of tile 0 in registers ----
  issue_a(0, 0);
  load_b(0, B0);
  issue_a(min(1, last), 1);
  load_b(min(1, last), B1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  tie_b(B0);
  tie_b(B1);
  __syncthreads();
  prep(B0, F0);
#pragma unroll
  for (int rt = 0; rt < 8; ++rt) af[rt] = *(const h8*)(fsm + rt * 4096 + (a_ad[0] - lds0));
  dequant_first();

  // one K step on LDS buffer BUF: parts 0..2, the barrier, part 3 (which starts tile kt + 1 and refills BUF).
  // The vmcnt wait that retires tile kt + 1 sits one whole block BEFORE the barrier behind which the tile is first
  // read (the guide's rule for staged buffers: read one phase after the wait that retires it, never in the same one);
  // the loads it waits for were issued a K step earlier, so it costs nothing there.
#define WOQ_KSTEP(BUF, BCUR, FCUR, BNXT, FNXT, KT)                                                                    \
  gemm_phase<BUF * STAGE, 0, RAW>(acc, af, bfrag(0, 0), bfrag(0, 1), bq[1], a_ad[1], BCUR.wv[0][1], BCUR.wv[1][1], pk, \
                                 FCUR[0][0], FCUR[1][0], 0, nullptr, nullptr, 0, 0, ro);                                  \
  gemm_phase<BUF * STAGE, 0, RAW>(acc, af, bfrag(1, 0), bfrag(1, 1), bq[0], a_ad[2], BCUR.wv[0][2], BCUR.wv[1][2], pk, \
                                 FCUR[0][NS - 1], FCUR[1][NS - 1], 0, nullptr, nullptr, 0, 0, ro);                        \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* tile KT + 1 and its weights, issued a K step ago */             \
  tie_b(BNXT);                                                                                                        \
  prep(BNXT, FNXT);                                                                                                   \
  gemm_phase<BUF * STAGE, 0, RAW>(acc, af, bfrag(0, 0), bfrag(0, 1), bq[1], a_ad[3], BCUR.wv[0][3], BCUR.wv[1][3], pk, \
                                 FCUR[0][NS - 1], FCUR[1][NS - 1], 0, nullptr, nullptr, 0, 0, ro);                        \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* my reads of BUF are done */                                   \
  __syncthreads();                                                                                                    \
  load_b(min((KT) + 2, last), BCUR);                                                                                  \
  gemm_phase<(1 - BUF) * STAGE, 1, RAW>(acc, af, bfrag(1, 0), bfrag(1, 1), bq[0], a_ad[0], BNXT.wv[0][0], BNXT.wv[1][0], \
                                      pk, FNXT[0][0], FNXT[1][0], lane16, a_src(min((KT) + 2, last)), \
                                      a_src(min((KT) + 2, last)) + 4096, dma_dst + BUF * STAGE,       \
                                      dma_dst + BUF * STAGE + 4096, ro);

  for (int kt = 0; kt < a.tiles_k; kt += 2) {  // tiles_k is even (launch_f16_t sends odd counts to gemm_f16s_kernel)
    WOQ_KSTEP(0, B0, F0, B1, F1, kt)
    WOQ_KSTEP(1, B1, F1, B0, F0, kt + 1)
  }
#undef WOQ_KSTEP
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // nothing may still be writing LDS when the workgroup retires
  WOQ_UNPIN_EPILOGUE_ARGS(a)
  gemm_epilogue<CT>(a, acc, row0, ct0, i16, kq);
}

