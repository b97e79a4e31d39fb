"""Host models of the decode-row kernels of the float weight types against the oracle — run without a GPU.

First the fp8-weight kernel (csrc/woq_gemv_fp8.hip), then the digit-plane unpack of the nf4 / fp4 kernels (last test).

The kernel feeds OCP fp8 code bytes straight into v_mfma_f32_16x16x32_fp8_{fp8,bf8} and brings the fp32 activation in as
six balanced base-16 digits (each an exact e4m3 value). What can be checked on the host is everything but the matrix
instruction itself: the code bytes a lane assembles from the blob's two nibble planes and the k index each byte stands
for (against the oracle's own reading of the blob), the digit decomposition of the fixed-point activation, and the
recombination sum_j 16^j S_j with per-tile scales — together against `oracle.woq_linear` (reference definition:
autograd/functions.py:41-63) at the bound the GPU test uses. The operand layout assumed for the instruction (lane
l: A[l & 15][8 (l >> 4) + byte], B[8 (l >> 4) + byte][l & 15], D rows 4 (l >> 4) .. + 3 of column l & 15) is the one
the int8 MFMA of the int4 kernels uses with 16 bytes per lane; the GPU suite checks the real thing.
"""
import numpy as np
import pytest

from oracle import woq_oracle as orc


def device_codes(blob):
    """codes [Kpad, Npad] as the kernel's lanes assemble them: b0..b3 of every (tile, lane, half) -> (k, n)."""
    bhi, blo = orc._int8_parts(blob)
    hh = orc.header(bhi)
    tiles_k, tiles_n = hh["Kpad"] // 128, hh["Npad"] // 16
    nwords = tiles_n * tiles_k * 256
    wh = bhi[hh["off_q"]:hh["off_q"] + nwords * 4].view(np.uint32).reshape(tiles_n, tiles_k, 64, 4)
    lh = orc.header(blo)
    wl = blo[lh["off_q"]:lh["off_q"] + nwords * 4].view(np.uint32).reshape(tiles_n, tiles_k, 64, 4)
    codes = np.zeros((hh["Kpad"], hh["Npad"]), np.uint8)
    lane = np.arange(64)
    i16, kq = lane & 15, lane >> 4
    m4 = np.uint32(0x0f0f0f0f)
    for h in range(2):
        h0, h1 = wh[..., 2 * h], wh[..., 2 * h + 1]
        l0, l1 = wl[..., 2 * h], wl[..., 2 * h + 1]
        b = [((h0 & m4) << np.uint32(4)) | ((l0 & m4) ^ np.uint32(0x08080808)),
             (h0 & np.uint32(0xf0f0f0f0)) | (((l0 >> np.uint32(4)) & m4) ^ np.uint32(0x08080808)),
             ((h1 & m4) << np.uint32(4)) | ((l1 & m4) ^ np.uint32(0x08080808)),
             (h1 & np.uint32(0xf0f0f0f0)) | (((l1 >> np.uint32(4)) & m4) ^ np.uint32(0x08080808))]
        for s in range(2):          # the two MFMAs of a half: B = {b[2 s], b[2 s + 1]} = 8 bytes
            for byte in range(8):
                word = b[2 * s + byte // 4]
                val = ((word >> np.uint32(8 * (byte % 4))) & np.uint32(0xff)).astype(np.uint8)  # [tn, kt, lane]
                for tn in range(tiles_n):
                    for kt in range(tiles_k):
                        k = kt * 128 + h * 64 + kq * 16 + s * 8 + byte
                        codes[k, tn * 16 + i16] = val[tn, kt]
    return codes, hh


def e4m3_exact(d):
    """every digit value in [-8, 8] is an OCP e4m3 value"""
    return bool(np.isin(np.abs(d), orc.FP8_TABLES[orc.W_FP8_E4M3][:0x80][np.isfinite(orc.FP8_TABLES[orc.W_FP8_E4M3][:0x80])]).all())


@pytest.mark.parametrize("wt", [orc.W_FP8_E4M3, orc.W_FP8_E5M2])
@pytest.mark.parametrize("K,N,group,tpw", [(512, 48, 128, 4), (384, 16, -1, 4), (2048, 32, 128, 4),
                                           (11008, 16, 128, 8),  # round 5: K > 8192 as eleven waves of eight tiles
                                           (512, 32, 32, 4), (768, 16, 64, 4), (768, 16, 96, 4)])  # one scale per 32-k block
def test_fp8_decode_kernel_model_vs_oracle(wt, K, N, group, tpw):
    rng = np.random.default_rng(5)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    q, s = orc.rtn_quantize_fp8(w, True, group, wt)
    blob = orc.repack_fp8(q, s, wt, None, group)
    codes, hh = device_codes(blob)
    assert np.array_equal(codes[:K, :N], orc.fp8_codes_of(blob))
    vals = orc.FP8_TABLES[wt][codes[:K, :N]].astype(np.float32)  # what the matrix core reads, [K, N]
    g = K if group == -1 else group
    scales = s.astype(np.float32)  # [G, N]
    tiles_k = hh["Kpad"] // 128
    assert tpw == (8 if tiles_k > 64 else 4)  # fp8_geometry (csrc/woq_gemv_fp8.hip)
    blocks = hh["scale_mode"] == 1  # groups of 32 / 64 / 96: each 64-k half is issued once per 32-k block, the other
    # block's two lane quarters read zero A rows, and every block is recombined and scaled on its own
    nw = (tiles_k + tpw - 1) // tpw
    base, rem = tiles_k // nw, tiles_k % nw
    x = rng.standard_normal(K).astype(np.float32)
    x[::7] *= 1e-3  # values far below the slice maximum
    out = np.zeros(N, np.float32)
    for wid in range(nw):  # a wave's K slice: its own exponent
        kt0 = wid * base + min(wid, rem)
        cnt = base + (1 if wid < rem else 0)
        lo, hi = kt0 * 128, min((kt0 + cnt) * 128, K)
        if lo >= hi:
            continue
        xs = x[lo:hi]
        amax = np.abs(xs).max()
        e = int(np.frexp(amax)[1]) if amax > 0 else 0
        sfix = np.float32(2.0 ** (21 - e))
        Q = (xs * sfix + np.float32(12582912.0)).astype(np.float32).view(np.uint32)  # one fp32 rounding, as the fma
        v = (Q & np.uint32(0x7fffff)).astype(np.int64) - (1 << 22)
        assert np.abs(v).max() <= 1 << 21
        digits, rest = [], v.copy()
        for _ in range(6):
            d = ((rest & 15) ^ 8) - 8  # low nibble, sign-extended
            rest = (rest - d) >> 4
            digits.append(d)
        assert not rest.any() and all(np.abs(d).max() <= 8 for d in digits) and all(e4m3_exact(d) for d in digits)
        assert np.array_equal(sum(d * 16 ** j for j, d in enumerate(digits)), v)
        tot = np.zeros(N, np.float32)
        for t in range(cnt):
            a, b = (kt0 + t) * 128, min((kt0 + t + 1) * 128, K)
            if a >= b:
                continue
            for a2 in (range(a, b, 32) if blocks else (a,)):
                b2 = min(a2 + 32, b) if blocks else b
                acc = [(digits[j][a2 - lo:b2 - lo].astype(np.float32)[:, None] * vals[a2:b2]).sum(0, dtype=np.float32)
                       for j in range(6)]
                comb = np.zeros(N, np.float32)
                for j in range(6):
                    comb += acc[j] * np.float32(16.0 ** j)
                tot += scales[min(a2 // g, scales.shape[0] - 1)] * comb
        out += tot * np.float32(2.0 ** (e - 21))
    ref = orc.woq_linear(x[None, :], blob, None)[0]
    mag = np.abs(x) @ np.abs(orc.dequantize_blob(blob))
    assert (np.abs(out - ref) <= 2e-6 * mag + 1e-5).all()


@pytest.mark.parametrize("wt,ct,tol", [(orc.W_NF4, 0, 2.0 ** -23), (orc.W_NF4, 1, 2.0 ** -16), (orc.W_FP4_E2M1, 0, 0.0),
                                        (orc.W_FP4_E2M1_BNB, 0, 1e-7)])
def test_table_digit_plane_unpack_model_vs_oracle(wt, ct, tol):
    """Host model of the nf4 / fp4 unpack of the decode GEMVs (csrc/woq_gemv_common.h lut_b): from the blob's weight
    dwords as a lane holds them, through the library's own digit planes (woq_table_digit_planes) and the byte lookup
    (two 8-entry permutes + select), to the value every (k, n) contributes — against the oracle's reading of the same
    blob (codes and table). Checks the k order of the B operand (low nibbles of w0, high nibbles of w0, low of w1, high of
    w1 = j 0..15 of the lane's run) and the planes together; the matrix instruction itself is the GPU suite's."""
    import ctypes
    import os

    from intel_extension_for_transformers_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libwoq_hip.so not built (python __graft_entry__.py)")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.woq_table_digit_planes.restype = ctypes.c_int
    lib.woq_table_digit_planes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    planes = (ctypes.c_uint32 * 12)()
    wmul = ctypes.c_float()
    ndig = lib.woq_table_digit_planes(wt, ct, planes, ctypes.byref(wmul))
    table_bytes = [np.array([(int(planes[j * 4 + c // 4]) >> (8 * (c % 4))) & 0xff for c in range(16)], np.uint8)
                   .view(np.int8).astype(np.int64) for j in range(3)]  # digit j of code c
    assert all(not table_bytes[j].any() for j in range(ndig, 3))
    K, N, group = 384, 48, 128
    rng = np.random.default_rng(9)
    codes = rng.integers(0, 16, (K, N)).astype(np.int8)
    scales = (rng.random((K // group, N), dtype=np.float32) + 0.5) * 0.01
    blob = orc.repack_table(codes, scales, wt, group)
    h = orc.header(blob)
    tiles_k, tiles_n = h["Kpad"] // 128, h["Npad"] // 16
    words = blob[h["off_q"]:h["off_q"] + tiles_n * tiles_k * 1024].view(np.uint32).reshape(tiles_n, tiles_k, 64, 4)
    lane = np.arange(64)
    i16, kq = lane & 15, lane >> 4
    S = 16.0 / wmul.value
    got = np.zeros((h["Kpad"], h["Npad"]))
    m4 = np.uint32(0x0f0f0f0f)
    for hh in range(2):
        w0, w1 = words[..., 2 * hh], words[..., 2 * hh + 1]
        quads = [w0 & m4, (w0 >> np.uint32(4)) & m4, w1 & m4, (w1 >> np.uint32(4)) & m4]  # idx of lut_quad, in B order
        for qi, idx in enumerate(quads):
            for byte in range(4):
                code = ((idx >> np.uint32(8 * byte)) & np.uint32(0xff)).astype(np.int64)  # [tn, kt, lane]
                # the lookup: codes 0..7 from the low table half, 8..15 from the high one, selected by bit 3
                val = sum(table_bytes[j][code] * 256 ** j for j in range(3)) / S
                j16 = qi * 4 + byte
                for tn in range(tiles_n):
                    for kt in range(tiles_k):
                        got[kt * 128 + hh * 64 + kq * 16 + j16, tn * 16 + i16] = val[tn, kt]
    want = orc.LUTS[wt][orc._codes_of(blob)].astype(np.float64)
    assert np.abs(got[:K, :N] - want).max() <= tol * float(np.abs(orc.LUTS[wt]).max()) + 1e-12


@pytest.mark.parametrize("K,N,group,nw,tpw", [(4096, 32, 128, 4, 8), (11008, 16, 128, 11, 8), (640, 16, 64, 2, 4),
                                              (1536, 16, -1, 2, 8)])
def test_act_order_gather_form_model_vs_oracle(K, N, group, nw, tpw):
    """Host model of the act-order form of the tile GEMV (csrc/woq_gemv_i8.hip, SHUF; round 5): the workgroup's copy pass
    (thread t, pass j -> four-element chunk t + j * threads, <= TPW / 2 chunks per thread) leaves x * norm_weight in LDS
    in natural order; wave w, lane l then picks elements kbase + 4 l + 256 j + i of ITS slice through the blob's index
    vector — `index_select(x, 1, g_idx)` of the parity definition (autograd/functions.py:41-63) — and the RMSNorm sum of
    squares comes from the copy pass (any partition of the vector sums to the same). Against `oracle.woq_linear` on the
    normalised row; the integer inner product itself is the GPU suite's."""
    rng = np.random.default_rng(9)
    q = rng.integers(-8, 8, (K, N), dtype=np.int8)
    g = K if group == -1 else group
    s = ((rng.random(((K + g - 1) // g, N), dtype=np.float32) + 0.5) * 0.01).astype(np.float32)
    g_idx = rng.permutation(np.arange(K, dtype=np.int32) // g).astype(np.int32)
    shuffle = orc.convert_idx(g_idx, K, g)
    blob = orc.repack(q, s, None, shuffle, group)
    x = rng.standard_normal(K).astype(np.float32)
    gw = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    threads, xj = nw * 64, tpw // 2
    tiles = (K + 127) // 128
    assert tiles <= nw * tpw and (K // 4 + threads - 1) // threads <= xj  # the chunk bound the kernel relies on
    # copy pass: every chunk exactly once, sum of squares per thread
    xs = np.full(K, np.nan, np.float32)
    ss = 0.0
    for t in range(threads):
        for j in range(xj):
            c = t + j * threads
            if c * 4 < K:
                assert np.isnan(xs[c * 4:c * 4 + 4]).all()
                xs[c * 4:c * 4 + 4] = x[c * 4:c * 4 + 4] * gw[c * 4:c * 4 + 4]
                ss += float((x[c * 4:c * 4 + 4].astype(np.float64) ** 2).sum())
    assert not np.isnan(xs).any() and abs(ss - float((x.astype(np.float64) ** 2).sum())) <= 1e-6 * ss
    # gather: wave slices as in the kernel (balanced contiguous K-tile ranges)
    base, rem = tiles // nw, tiles % nw
    row = np.zeros(K, np.float32)  # the activation each regrouped weight row meets
    seen = np.zeros(K, bool)
    for w in range(nw):
        kt0 = w * base + min(w, rem)
        cnt = base + (1 if w < rem else 0)
        kbase = kt0 * 128
        xlen = max(0, min(cnt * 128, K - kbase))
        for lane in range(64):
            for j in range(xj):
                for i in range(4):
                    off = lane * 4 + j * 256 + i
                    if off < xlen:
                        e = kbase + off
                        assert not seen[e]
                        seen[e] = True
                        row[e] = xs[shuffle[e]]
    assert seen.all()
    inv = np.float32(1.0 / np.sqrt(ss / K + 1e-5))
    got = (row.astype(np.float64) @ orc.dequantize_blob(blob).astype(np.float64)) * inv
    normed = (x * inv * gw).astype(np.float32)
    ref = orc.woq_linear(normed[None, :], blob, None)[0]
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max() + 1e-6
