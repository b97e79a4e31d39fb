"""numpy front-end of the CPU oracle (oracle/woq_oracle.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package. See the header of woq_oracle.c for what is
pinned against the reference and what is "parity unpinned".

Every function names the reference file:line it restates in woq_oracle.c; this module only
marshals numpy arrays through ctypes and adds the whole-decoder composition used for logits
parity (`LlamaOracle`).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libwoq_oracle.so")

F32, BF16, F16 = 0, 1, 2  # enum woq_dtype (include/woq_blob.h)
HEADER_BYTES = 256


def build(force=False):
    """Compile oracle/woq_oracle.c with gcc (idempotent)."""
    srcs = [os.path.join(_HERE, "woq_oracle.c"), os.path.join(_HERE, "woq_cpu_port.c"),
            os.path.join(_HERE, "..", "include", "woq_blob.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(f) for f in srcs)):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.orc_packed_size.restype = ctypes.c_size_t
        L.orc_packed_size.argtypes = [ctypes.c_int] * 6
        L.orc_load_scalar.restype = ctypes.c_float
        L.cpk_bytes.restype = ctypes.c_size_t
        L.cpk_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
        L.cpk_fill_random.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint64]
        _lib = L
    return _lib


def host_threads():
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (a container can show
    256 CPUs and grant 16 of them; an OpenMP team sized by the former then spends its time being descheduled)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, per = int(fq.read()), int(fp.read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    env = os.environ.get("OMP_NUM_THREADS")
    return min(n, int(env)) if env and env.isdigit() and int(env) > 0 else n


def set_threads(n):
    """Size the oracle's OpenMP team (the shared libgomp runtime of this process)."""
    lib()
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass
    return int(n)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


# ---- format decode (reference: llm/quantization/utils.py:82-125, nn/modules.py:225-227) ----
def unpack_weight(qweight, qzeros, K, N, G, bits=4, sym=True):
    qweight = _c(qweight, np.int32)
    qzeros = _c(qzeros, np.int32)
    w = np.empty((K, N), np.int8)
    z = np.empty((G, N), np.int8) if qzeros is not None else None
    lib().orc_unpack_weight(_p(qweight), _p(qzeros), K, N, G, bits, int(sym), _p(w), _p(z))
    return w, z


def to_signed_nibble(t):
    t = np.array(t, dtype=np.int8, copy=True)
    lib().orc_to_signed_nibble(_p(t), ctypes.c_size_t(t.size))
    return t


def convert_idx(g_idx, K, blocksize):
    g_idx = _c(g_idx, np.int32)
    ret = np.zeros(K, np.int32)
    lib().orc_convert_idx(_p(g_idx), K, blocksize, _p(ret))
    return ret


# ---- quantise / dequantise -------------------------------------------------------------
def rtn_quantize(w, transpose, group, asym):
    """PARITY UNPINNED rounding rule (see woq_oracle.c). w: [K,N] or [N,K] if transpose."""
    w = _c(w, np.float32)
    K, N = (w.shape[1], w.shape[0]) if transpose else w.shape
    g = K if group in (-1, 0) or group > K else group
    G = (K + g - 1) // g
    q = np.empty((K, N), np.int8)
    s = np.empty((G, N), np.float32)
    z = np.empty((G, N), np.int8) if asym else None
    lib().orc_rtn_quantize(_p(w), int(transpose), K, N, g, int(asym), _p(q), _p(s), _p(z))
    return q, s, z


def dequant_raw(q, scales, zp, group):
    q = _c(q, np.int8)
    K, N = q.shape
    out = np.empty((K, N), np.float32)
    lib().orc_dequant_raw(_p(q), _p(_c(scales, np.float32)), _p(_c(zp, np.int8)), K, N, group, _p(out))
    return out


def packed_size(K, N, group, scale_type=F32, asym=False, act_shuffle=False):
    return lib().orc_packed_size(K, N, group, scale_type, int(asym), int(act_shuffle))


def repack(q, scales, zp=None, shuffle=None, group=-1, scale_type=F32, compute_type=0):
    """reference: qbits.cpp:61-77 / bestla_packq_impl.cpp:20-41 (layout transform only)."""
    q = _c(q, np.int8)
    K, N = q.shape
    scales = _c(scales, np.float32)
    zp = _c(zp, np.int8)
    shuffle = _c(shuffle, np.int32)
    size = packed_size(K, N, group, scale_type, zp is not None, shuffle is not None)
    if size == 0:
        raise RuntimeError("QBits: unsupported blocksize %d for K=%d" % (group, K))
    blob = np.zeros(size, np.uint8)
    rc = lib().orc_repack(_p(q), _p(scales), _p(zp), _p(shuffle), K, N, group, scale_type, compute_type,
                          _p(blob), ctypes.c_size_t(size))
    if rc != 0:
        raise RuntimeError("orc_repack failed")
    return blob


# ---- 4-bit table weight types (nf4 / fp4_e2m1 / fp4_e2m1_bnb): w = table[code] * scale (include/woq_blob.h) -------------
W_NF4, W_FP4_E2M1, W_FP4_E2M1_BNB = 2, 3, 4
LUTS = {
    W_NF4: np.array([-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635,
                     -0.18477343022823334, -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725,
                     0.24611230194568634, 0.33791524171829224, 0.44070982933044434, 0.5626170039176941,
                     0.7229568362236023, 1.0], np.float32),
    W_FP4_E2M1: np.array([0, .5, 1, 1.5, 2, 3, 4, 6, -0.0, -.5, -1, -1.5, -2, -3, -4, -6], np.float32),
    W_FP4_E2M1_BNB: np.array([0, 0.0625 / 12, 8 / 12, 1, 4 / 12, .5, 2 / 12, .25,
                              -0.0, -0.0625 / 12, -8 / 12, -1, -4 / 12, -.5, -2 / 12, -.25], np.float32),
}
LUT_MAX = {W_NF4: 1.0, W_FP4_E2M1: 6.0, W_FP4_E2M1_BNB: 1.0}


def rtn_quantize_table(w, transpose, group, wtype):
    """scale = max|w| / table_max per group, code = nearest table entry of w / scale, lowest code on ties — fp32
    arithmetic like the device kernel (rounding rule parity-unpinned: BesTLA's quantiser is not in the reference tree)."""
    w = np.asarray(w, np.float32)
    w = w.T if transpose else w
    K, N = w.shape
    g = K if group in (-1, 0) or group > K else group
    G = (K + g - 1) // g
    lut = LUTS[wtype]
    q = np.empty((K, N), np.int8)
    s = np.empty((G, N), np.float32)
    for gi in range(G):
        blk = w[gi * g:min(K, (gi + 1) * g)]
        sc = (np.abs(blk).max(0) / np.float32(LUT_MAX[wtype])).astype(np.float32)
        sc[sc == 0] = 1
        d = np.abs((blk / sc)[..., None].astype(np.float32) - lut[None, None, :])
        q[gi * g:gi * g + blk.shape[0]] = d.argmin(-1).astype(np.int8)  # argmin returns the first (lowest) on ties
        s[gi] = sc
    return q, s


def repack_table(codes, scales, wtype, group=-1, scale_type=F32, compute_type=0):
    """codes int8 [K, N] in 0..15 -> blob: the int4 layout with the nibble holding the code, header weight_type set."""
    blob = repack(np.asarray(codes, np.int8), scales, None, None, group, scale_type, compute_type)
    blob[:HEADER_BYTES].view(np.uint32)[10] = wtype
    return blob


def _codes_of(blob):
    """raw 4-bit codes [K, N] of a blob (nibble (k, n) per include/woq_blob.h woq_q_byte)."""
    h = header(blob)
    K, N, tiles_k = h["K"], h["N"], h["Kpad"] // 128
    k = np.arange(K)[:, None]
    n = np.arange(N)[None, :]
    tn, i, kt, r = n // 16, n % 16, k // 128, k % 128
    hh, kq, j = r // 64, (r % 64) // 16, r % 16
    lane = kq * 16 + i
    word = ((tn * tiles_k + kt) * 64 + lane) * 4 + hh * 2 + j // 8
    jj = j % 8
    sh = 8 * (jj & 3) + 4 * (jj >> 2)
    words = blob[h["off_q"]:h["off_q"] + (h["Npad"] // 16) * tiles_k * 1024].view(np.uint32)
    return ((words[word] >> sh.astype(np.uint32)) & 15).astype(np.int64)


def dequantize_table(blob, transpose=False):
    h = header(blob)
    lut = LUTS[h["weight_type"]]
    codes = _codes_of(blob)
    # the scales live where an int4 blob keeps them: read them back by dequantising an all-ones int4 view
    probe = blob.copy()
    probe[:HEADER_BYTES].view(np.uint32)[10] = W_INT4
    probe[h["off_q"]:h["off_scale"]].view(np.uint32)[:] = 0x11111111  # every nibble = +1 -> dequant = 1 * scale
    sc = dequantize_blob(probe)
    w = (lut[codes] * sc).astype(np.float32)
    return w.T.copy() if transpose else w


# ---- int8 weights: composite of two int4 blobs (include/woq_blob.h woq_int8_headers) -------------------------------
W_INT4, W_INT8 = 0, 1


def split_int8(q8, scales, zp8):
    """q8 - zp8 = 16 (hi - zhi) + (lo - zlo) with every term in [-8, 7]; scales 16 s / s."""
    q8 = np.asarray(q8, np.int8).astype(np.int32)
    hi, lo = (q8 >> 4).astype(np.int8), ((q8 & 15) - 8).astype(np.int8)
    s = np.asarray(scales, np.float32)
    z = np.zeros(s.shape, np.int32) if zp8 is None else np.asarray(zp8, np.int8).astype(np.int32)
    zhi = None if zp8 is None else (z >> 4).astype(np.int8)
    zlo = ((z & 15) - 8).astype(np.int8)
    return (hi, (16.0 * s).astype(np.float32), zhi), (lo, s, zlo)


def repack_int8(q8, scales, zp8=None, shuffle=None, group=-1, scale_type=F32, compute_type=0):
    """int8 [K, N] + fp32 scales [G, N] + int8 zp [G, N] -> composite blob (outer header + HI blob + LO blob)."""
    (hi, shi, zhi), (lo, slo, zlo) = split_int8(q8, scales, zp8)
    bhi = repack(hi, shi, zhi, shuffle, group, scale_type, compute_type)
    blo = repack(lo, slo, zlo, shuffle, group, scale_type, compute_type)
    outer = bhi[:HEADER_BYTES].copy()
    lo_h = header(blo)
    outer.view(np.uint32)[10] = W_INT8
    u64 = outer.view(np.uint64)
    u64[8] = HEADER_BYTES
    u64[9] = HEADER_BYTES + bhi.size
    u64[10] = HEADER_BYTES + bhi.size + lo_h["off_zp"] if zp8 is not None else 0
    u64[11] = HEADER_BYTES + bhi.size + lo_h["off_shuffle"] if shuffle is not None else 0
    u64[1] = HEADER_BYTES + bhi.size + blo.size
    return np.concatenate([outer, bhi, blo])


def _int8_parts(blob):
    h = header(blob)
    return blob[h["off_q"]:h["off_scale"]], blob[h["off_scale"]:h["total_bytes"]]


def rtn_quantize_int8(w, transpose, group, asym):
    """8-bit form of the (parity-unpinned) RTN rule of woq_oracle.c, in fp32 arithmetic like the device kernel:
    sym s = max|w| / 127, q = clip(rne(w / s), -128, 127); asym s = (max - min) / 255, z = clip(rne(-min / s), 0, 255),
    q = clip(rne(w / s) + z, 0, 255) - 128, zp = z - 128."""
    return rtn_quantize_bits(w, transpose, group, asym, 8)


NARROW_BITS = {"int3_clip": 3, "int2_clip": 2}


def repack_narrow(q, scales, zp=None, shuffle=None, group=-1, scale_type=F32, compute_type=0, bits=3):
    """int3_clip / int2_clip (reference weight-type strings, bestla_weightonly_dispatcher.hpp:62-70): values in
    [-4, 3] / [-2, 1] in int4 storage — the int4 blob with header word 15 (narrow_bits) = bits."""
    q = np.asarray(q, np.int8)
    lim = 1 << (bits - 1)
    assert q.min() >= -lim and q.max() < lim, "values outside the %d-bit signed range" % bits
    blob = repack(q, scales, zp, shuffle, group, scale_type, compute_type)
    blob.view(np.uint32)[15] = bits
    return blob


def rtn_quantize_bits(w, transpose, group, asym, bits):
    """The (parity-unpinned) RTN rule of woq_oracle.c for any width, in fp32 arithmetic like the device kernel:
    qmax = 2^(bits-1) - 1, levels = 2^bits - 1, off = 2^(bits-1);
    sym s = max|w| / qmax, q = clip(rne(w / s), -off, qmax); asym s = (max - min) / levels,
    z = clip(rne(-min / s), 0, levels), q = clip(rne(w / s) + z, 0, levels) - off, zp = z - off."""
    qmax, levels, off = (1 << (bits - 1)) - 1, (1 << bits) - 1, 1 << (bits - 1)
    w = np.asarray(w, np.float32)
    w = w.T if transpose else w
    K, N = w.shape
    g = K if group in (-1, 0) or group > K else group
    G = (K + g - 1) // g
    q = np.empty((K, N), np.int8)
    s = np.empty((G, N), np.float32)
    z = np.empty((G, N), np.int8) if asym else None
    for gi in range(G):
        blk = w[gi * g:min(K, (gi + 1) * g)]
        if not asym:
            sc = (np.abs(blk).max(0) / np.float32(qmax)).astype(np.float32)
            sc[sc == 0] = 1
            q[gi * g:gi * g + blk.shape[0]] = np.clip(np.rint(blk / sc), -off, qmax).astype(np.int8)
        else:
            mx, mn = blk.max(0), blk.min(0)
            sc = ((mx - mn) / np.float32(levels)).astype(np.float32)
            sc[sc == 0] = 1
            zz = np.clip(np.rint(-mn / sc), 0, levels).astype(np.int32)
            q[gi * g:gi * g + blk.shape[0]] = (np.clip(np.rint(blk / sc).astype(np.int32) + zz, 0, levels) - off).astype(np.int8)
            z[gi] = (zz - off).astype(np.int8)
        s[gi] = sc
    return q, s, z


# ---- fp8 weights (fp8_e4m3 / fp8_e5m2, reference strings bestla_weightonly_dispatcher.hpp:62-70): two nibble planes in
# the int8 composite container, w = value(code) * scale, symmetric only (include/woq_blob.h woq_fp8_headers) -----------
W_FP8_E4M3, W_FP8_E5M2 = 7, 8
FLAG_SCALE_E8M0 = 4


def _fp8_table(wtype):
    """OCP e4m3fn (no infinities, S.1111.111 = NaN) / e5m2 (exponent 31 = inf / NaN), by the definition."""
    t = np.empty(256, np.float32)
    for c in range(256):
        s = -1.0 if c & 0x80 else 1.0
        if wtype == W_FP8_E4M3:
            e, m = (c >> 3) & 15, c & 7
            v = np.nan if (e == 15 and m == 7) else (m * 2.0 ** -9 if e == 0 else (8 + m) * 2.0 ** (e - 10))
        else:
            e, m = (c >> 2) & 31, c & 3
            v = (np.inf if m == 0 else np.nan) if e == 31 else (m * 2.0 ** -16 if e == 0 else (4 + m) * 2.0 ** (e - 17))
        t[c] = s * v
    return t


FP8_TABLES = {W_FP8_E4M3: _fp8_table(W_FP8_E4M3), W_FP8_E5M2: _fp8_table(W_FP8_E5M2)}
FP8_MAX = {W_FP8_E4M3: 448.0, W_FP8_E5M2: 57344.0}


def rtn_quantize_fp8(w, transpose, group, wtype, e8m0=False):
    """scale = max|w| / fp8_max per group (e8m0: the next power of two at or above it), code = the finite code whose
    value is nearest to w / scale in fp32, lowest code on ties — the device kernel's rule (parity unpinned: BesTLA's
    quantiser is not in the reference tree). Returns codes uint8 [K, N], scales fp32 [G, N]."""
    w = np.asarray(w, np.float32)
    w = w.T if transpose else w
    K, N = w.shape
    g = K if group in (-1, 0) or group > K else group
    G = (K + g - 1) // g
    tab = FP8_TABLES[wtype]
    finite = np.isfinite(tab)
    q = np.empty((K, N), np.uint8)
    s = np.empty((G, N), np.float32)
    for gi in range(G):
        blk = w[gi * g:min(K, (gi + 1) * g)]
        sc = (np.abs(blk).max(0) / np.float32(FP8_MAX[wtype])).astype(np.float32)
        sc[sc == 0] = 1
        if e8m0:
            f, e = np.frexp(sc)
            sc = np.ldexp(np.float32(1), np.where(f == 0.5, e - 1, e)).astype(np.float32)
        d = np.abs((blk / sc)[..., None].astype(np.float32) - tab[None, None, :])
        d[..., ~finite] = np.inf
        q[gi * g:gi * g + blk.shape[0]] = d.argmin(-1).astype(np.uint8)  # first (lowest) code on ties
        s[gi] = sc
    return q, s


def repack_fp8(codes, scales, wtype, shuffle=None, group=-1, scale_type=F32, compute_type=0, e8m0=False):
    """code bytes [K, N] + fp32 scales [G, N] -> composite blob: HI plane = code >> 4, LO plane = (code & 15) ^ 8 as
    stored nibbles (what the int8 splitter makes of the byte read as a signed value), scales on both planes; e8m0
    scales are stored as bf16 with the header flag set."""
    c = np.asarray(codes).astype(np.uint8).view(np.int8).astype(np.int32)
    hi, lo = (c >> 4).astype(np.int8), ((c & 15) - 8).astype(np.int8)
    st = BF16 if e8m0 else scale_type
    scales = np.asarray(scales, np.float32)
    bhi = repack(hi, scales, None, shuffle, group, st, compute_type)
    blo = repack(lo, scales, np.zeros(scales.shape, np.int8), shuffle, group, st, compute_type)
    outer = bhi[:HEADER_BYTES].copy()
    lo_h = header(blo)
    u32 = outer.view(np.uint32)
    u32[10] = wtype
    if e8m0:
        u32[13] |= FLAG_SCALE_E8M0
    u64 = outer.view(np.uint64)
    u64[8] = HEADER_BYTES
    u64[9] = HEADER_BYTES + bhi.size
    u64[10] = 0
    u64[11] = HEADER_BYTES + bhi.size + lo_h["off_shuffle"] if shuffle is not None else 0
    u64[1] = HEADER_BYTES + bhi.size + blo.size
    return np.concatenate([outer, bhi, blo])


def fp8_codes_of(blob):
    bhi, blo = _int8_parts(blob)
    return ((_codes_of(bhi) << 4) | (_codes_of(blo) ^ 8)).astype(np.uint8)


def dequantize_fp8(blob, transpose=False):
    h = header(blob)
    bhi, _ = _int8_parts(blob)
    probe = bhi.copy()  # scales: dequantise an all-ones int4 view of the HI plane
    hh = header(probe)
    probe[hh["off_q"]:hh["off_scale"]].view(np.uint32)[:] = 0x11111111
    sc = dequantize_blob(probe)
    w = (FP8_TABLES[h["weight_type"]][fp8_codes_of(blob)] * sc).astype(np.float32)
    return w.T.copy() if transpose else w


def header(blob):
    """Parse the WQH1 header (include/woq_blob.h) into a dict."""
    h = np.frombuffer(np.ascontiguousarray(blob[:HEADER_BYTES]).tobytes(), dtype=np.uint8)
    u32 = h.view(np.uint32)
    i32 = h.view(np.int32)
    u64 = h.view(np.uint64)
    return dict(magic=int(u32[0]), version=int(u32[1]), total_bytes=int(u64[1]), K=int(i32[4]), N=int(i32[5]),
                group=int(i32[6]), Kpad=int(i32[7]), Npad=int(i32[8]), n_groups=int(i32[9]),
                weight_type=int(u32[10]), scale_type=int(u32[11]), compute_type=int(u32[12]), flags=int(u32[13]),
                scale_mode=int(u32[14]), narrow_bits=int(u32[15]), off_q=int(u64[8]), off_scale=int(u64[9]), off_zp=int(u64[10]),
                off_shuffle=int(u64[11]))


def dequantize_blob(blob, transpose=False):
    """reference: qbits.cpp:102-111."""
    blob = _c(blob, np.uint8)
    h = header(blob)
    if h["weight_type"] == W_INT8:  # (hi - zhi) * 16s + (lo - zlo) * s, the two fp32 terms added in this order
        bhi, blo = _int8_parts(blob)
        return dequantize_blob(bhi, transpose) + dequantize_blob(blo, transpose)
    if h["weight_type"] in LUTS:
        return dequantize_table(blob, transpose)
    if h["weight_type"] in FP8_TABLES:
        return dequantize_fp8(blob, transpose)
    out = np.empty((h["N"], h["K"]) if transpose else (h["K"], h["N"]), np.float32)
    rc = lib().orc_dequantize_blob(_p(blob), _p(out), int(transpose))
    if rc != 0:
        raise RuntimeError("bad blob")
    return out


def _to_storage(a, dtype):
    if dtype == F32:
        return np.ascontiguousarray(a, np.float32)
    out = np.empty(a.shape, np.uint16)
    flat = np.ascontiguousarray(a, np.float32).ravel()
    of = out.ravel()
    for i in range(flat.size):  # small arrays only
        lib().orc_store_scalar(_p(of), ctypes.c_size_t(i), dtype, ctypes.c_float(float(flat[i])))
    return of.reshape(a.shape)


def bf16_round(a):
    """fp32 -> bf16 -> fp32 (round-to-nearest-even), vectorised."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(a))


def woq_linear(x, blob, bias=None, out_dtype=F32):
    """THE parity definition — reference: llm/quantization/autograd/functions.py:41-63."""
    x = _c(x, np.float32)
    blob = _c(blob, np.uint8)
    bias = _c(bias, np.float32)
    h = header(blob)
    if h["weight_type"] == W_INT8 or h["weight_type"] in LUTS or h["weight_type"] in FP8_TABLES:
        # dequantise -> fp32 matmul -> + bias
        w = dequantize_blob(blob).astype(np.float64)
        xs = x if not h["off_shuffle"] else x[:, np.frombuffer(
            blob[h["off_shuffle"]:h["off_shuffle"] + 4 * h["K"]].tobytes(), np.int32)]
        y = (xs.astype(np.float64) @ w).astype(np.float32)
        if bias is not None:
            y = y + bias
        if out_dtype == BF16:
            return bf16_round(y)
        if out_dtype == F16:
            return y.astype(np.float16).astype(np.float32)
        return y
    M = x.shape[0]
    out = np.empty((M, h["N"]), np.float32 if out_dtype == F32 else np.uint16)
    rc = lib().orc_woq_linear(_p(x), x.shape[1], _p(blob), _p(bias), _p(out), out_dtype, h["N"], M)
    if rc != 0:
        raise RuntimeError("orc_woq_linear failed")
    if out_dtype == BF16:
        return (out.astype(np.uint32) << 16).view(np.float32)
    if out_dtype == F16:
        return out.view(np.float16).astype(np.float32)
    return out


def woq_gemv_stream(x, blob, bias=None):
    """fp32-accumulate streaming GEMV straight from the packed nibbles (the cpu_baseline 'port')."""
    x = _c(x, np.float32).ravel()
    blob = _c(blob, np.uint8)
    bias = _c(bias, np.float32)
    h = header(blob)
    out = np.empty(h["N"], np.float32)
    rc = lib().orc_woq_gemv_stream(_p(x), _p(blob), _p(bias), _p(out))
    if rc != 0:
        raise RuntimeError("orc_woq_gemv_stream failed")
    return out


# ---- the CPU port timed as bench.py's cpu_baseline (oracle/woq_cpu_port.c) ---------------------
class CpuPortLinear:
    """One int4 linear on the CPU-friendly "CPK" layout (32-column blocks, K contiguous, nibble pairs): the port of
    the reference's BesTLA fp32 core that `cpu_baseline` times. From (q, scales, zp) for the parity check, or
    `synthetic(K, N, group, asym, seed)` for the timing leg (random nibbles filled in place, first-touched by the
    thread that streams them)."""

    def __init__(self, q, scales, zp, group):
        q = _c(q, np.int8)
        self.K, self.N = q.shape
        self.group = self.K if group in (-1, 0) or group > self.K else group
        self.scales = _c(scales, np.float32)
        self.zp = _c(zp, np.int8)
        self.w = np.empty(lib().cpk_bytes(self.K, self.N), np.uint8)
        lib().cpk_pack(_p(q), self.K, self.N, _p(self.w))

    @classmethod
    def synthetic(cls, K, N, group, asym, seed):
        self = cls.__new__(cls)
        self.K, self.N, self.group = K, N, group
        G = (K + group - 1) // group
        rng = np.random.default_rng(seed)
        self.scales = ((rng.random((G, N), dtype=np.float32) + 0.5) * 0.005).astype(np.float32)
        self.zp = rng.integers(-8, 8, (G, N), dtype=np.int8) if asym else None
        self.w = np.empty(lib().cpk_bytes(K, N), np.uint8)
        lib().cpk_fill_random(_p(self.w), K, N, seed)
        return self

    @property
    def nbytes(self):
        return self.w.nbytes + self.scales.nbytes + (self.zp.nbytes if self.zp is not None else 0)

    def __call__(self, x, bias=None, out=None):
        x = _c(x, np.float32).ravel()
        out = np.empty(self.N, np.float32) if out is None else out
        lib().cpk_gemv(_p(x), _p(self.w), _p(self.scales), _p(self.zp), _p(_c(bias, np.float32)), self.K, self.N,
                       self.group, _p(out))
        return out


def cpu_port_isa():
    return {2: "avx512", 1: "avx2", 0: "scalar"}[lib().cpk_isa()]


# ---- ops between the linears (HF semantics) ---------------------------------
def rmsnorm(x, weight, eps):
    x = _c(x, np.float32)
    out = np.empty_like(x)
    d = x.shape[-1]
    lib().orc_rmsnorm(_p(x), _p(_c(weight, np.float32)), ctypes.c_float(eps), x.size // d, d, _p(out))
    return out


def rope(x, pos, theta=10000.0):
    """x: [tokens, heads, D] -> rotated copy."""
    x = np.array(x, dtype=np.float32, copy=True, order="C")
    pos = _c(pos, np.int32)
    lib().orc_rope(_p(x), _p(pos), x.shape[0], x.shape[1], x.shape[2], ctypes.c_float(theta))
    return x


def silu_mul(gate, up):
    gate = _c(gate, np.float32)
    up = _c(up, np.float32)
    out = np.empty_like(gate)
    lib().orc_silu_mul(_p(gate), _p(up), ctypes.c_size_t(gate.size), _p(out))
    return out


def gelu(x, approximate="tanh"):
    x = _c(x, np.float32)
    out = np.empty_like(x)
    f = lib().orc_gelu_tanh if approximate == "tanh" else lib().orc_gelu_erf
    f(_p(x), ctypes.c_size_t(x.size), _p(out))
    return out


def attn_decode(q, kcache, vcache):
    """q [heads,D]; caches [ctx, kv_heads, D] -> [heads, D]."""
    q = _c(q, np.float32)
    kcache = _c(kcache, np.float32)
    vcache = _c(vcache, np.float32)
    out = np.empty_like(q)
    lib().orc_attn_decode(_p(q), _p(kcache), _p(vcache), q.shape[0], kcache.shape[1], q.shape[1], kcache.shape[0],
                          _p(out))
    return out


def linear_rows(x, blob):
    """Many-row form of `woq_linear` for prompt-sized inputs: dequantise -> matmul (autograd/functions.py:41-63, the
    reference's own definition), the contraction in fp64 through BLAS so that thousands of rows cost seconds. Act-order
    (GPTQ g_idx) int4 blobs: the activation columns are gathered by the blob's stored shuffle first —
    `index_select(x, 1, g_idx)` of that same definition (functions.py:48-50)."""
    blob = _c(blob, np.uint8)
    h = header(blob)
    x = np.asarray(x, np.float64)
    if h["off_shuffle"] and h["weight_type"] not in (W_INT8,) and h["weight_type"] not in FP8_TABLES:
        idx = np.frombuffer(blob[h["off_shuffle"]:h["off_shuffle"] + 4 * h["K"]].tobytes(), dtype=np.int32)
        x = x[:, idx]
    elif h["off_shuffle"]:
        raise NotImplementedError("linear_rows: act-order shuffle of composite blobs (use woq_linear)")
    w = dequantize_blob(blob).astype(np.float64)
    return (x @ w).astype(np.float32)


def attn_prompt(q, kcache, vcache, start, window=0, qblock=512):
    """Causal (optionally sliding-window) attention of T query rows over a cache — HF `eager_attention_forward` +
    `repeat_kv` semantics: scores q.k / sqrt(D), softmax, P.V; query i sits at position start + i
    and sees positions max(0, p - window + 1) .. p. q [T, heads, D]; caches [start + T, kv_heads, D] -> [T, heads, D].
    fp64 throughout (the many-row twin of orc_attn_decode)."""
    q = np.asarray(q, np.float64)
    T, H, D = q.shape
    KV = kcache.shape[1]
    rep = H // KV
    ctx = kcache.shape[0]
    out = np.empty((T, H, D), np.float32)
    scale = 1.0 / np.sqrt(float(D))
    kpos = np.arange(ctx)[None, :]
    for kh in range(KV):
        kk = np.ascontiguousarray(kcache[:, kh, :], np.float64)
        vv = np.ascontiguousarray(vcache[:, kh, :], np.float64)
        for hh in range(kh * rep, (kh + 1) * rep):
            for r0 in range(0, T, qblock):
                r1 = min(T, r0 + qblock)
                hi = start + r1  # positions beyond the block's last query are never visible
                lo = max(0, start + r0 - window + 1) if window else 0
                s = (q[r0:r1, hh, :] @ kk[lo:hi].T) * scale
                qp = (start + np.arange(r0, r1))[:, None]
                vis = kpos[:, lo:hi] <= qp
                if window:
                    vis &= kpos[:, lo:hi] > qp - window
                s = np.where(vis, s, -np.inf)
                s -= s.max(axis=1, keepdims=True)
                p = np.exp(s)
                p /= p.sum(axis=1, keepdims=True)
                out[r0:r1, hh, :] = (p @ vv[lo:hi]).astype(np.float32)
    return out


# ---- whole-decoder composition (Llama-class), fp32, used for logits parity ---------------------
class LlamaOracle:
    """fp32 CPU decoder built from the oracle ops, on the SAME (q, scale, zp) blobs as the GPU.

    Composition follows HF LlamaDecoderLayer (RMSNorm -> q/k/v -> RoPE -> attention + KV cache ->
    o ; RMSNorm -> gate/up -> SiLU*mul -> down), which is what the reference executes around its
    QuantizedLinearQBits modules. `layers` is a list of dicts with blobs
    'q','k','v','o','gate','up','down' (numpy uint8 WQH1 blobs) and 'ln1','ln2' fp32 vectors; 'qkv' (q | k | v
    along N) and 'gate_up' (16-column tiles interleaved) may replace the separate projections.
    """

    def __init__(self, cfg, embed, layers, norm, lm_head):
        self.cfg = cfg
        self.embed = np.asarray(embed, np.float32)
        self.layers = layers
        self.norm = np.asarray(norm, np.float32)
        self.lm_head = np.asarray(lm_head, np.float32)  # [vocab, hidden], NOT quantised (F9)
        L = len(layers)
        self.k = [np.zeros((0, cfg["kv_heads"], cfg["head_dim"]), np.float32) for _ in range(L)]
        self.v = [np.zeros((0, cfg["kv_heads"], cfg["head_dim"]), np.float32) for _ in range(L)]
        # kv_hook(layer, first_position, k_rows, v_rows) -> (k_rows, v_rows) to CACHE instead (None: cache what was
        # computed). Parity tests of a low-precision device cache (fp8) hand the device's own stored rows back here, so
        # that every layer is checked on the inputs the device really attended over and one layer's storage rounding
        # does not blur the comparison of the next.
        self.kv_hook = None

    def reset(self):
        for i in range(len(self.k)):
            self.k[i] = self.k[i][:0]
            self.v[i] = self.v[i][:0]

    def forward_token(self, token, pos):
        c = self.cfg
        H, KV, D = c["heads"], c["kv_heads"], c["head_dim"]
        h = self.embed[token].reshape(1, -1).copy()
        for li, ly in enumerate(self.layers):
            x = rmsnorm(h, ly["ln1"], c["eps"])
            if "qkv" in ly:  # one blob with q | k | v concatenated along N (the engine's fused layout)
                qkv = woq_linear(x, ly["qkv"])
                q = qkv[:, :H * D].reshape(1, H, D)
                k = qkv[:, H * D:(H + KV) * D].reshape(1, KV, D)
                v = qkv[:, (H + KV) * D:].reshape(1, KV, D)
            else:
                q = woq_linear(x, ly["q"]).reshape(1, H, D)
                k = woq_linear(x, ly["k"]).reshape(1, KV, D)
                v = woq_linear(x, ly["v"]).reshape(1, KV, D)
            q = rope(q, [pos], c["theta"])
            k = rope(k, [pos], c["theta"])
            if self.kv_hook is not None:
                k, v = self.kv_hook(li, pos, k, v)
            self.k[li] = np.concatenate([self.k[li], k], 0)
            self.v[li] = np.concatenate([self.v[li], v], 0)
            w = c.get("window", 0)  # HF Mistral sliding_window: a query sees the last `window` positions (itself included)
            kk, vv = (self.k[li][-w:], self.v[li][-w:]) if w else (self.k[li], self.v[li])
            a = attn_decode(q[0], kk, vv).reshape(1, H * D)
            h = h + woq_linear(a, ly["o"])
            x = rmsnorm(h, ly["ln2"], c["eps"])
            if "gate_up" in ly:  # 16-column tiles interleaved gate, up, gate, up, ... (runtime.fuse_gate_up)
                gu = woq_linear(x, ly["gate_up"]).reshape(1, -1, 2, 16)
                g = np.ascontiguousarray(gu[:, :, 0, :]).reshape(1, -1)
                u = np.ascontiguousarray(gu[:, :, 1, :]).reshape(1, -1)
            else:
                g = woq_linear(x, ly["gate"])
                u = woq_linear(x, ly["up"])
            h = h + woq_linear(silu_mul(g, u), ly["down"])
        x = rmsnorm(h, self.norm, c["eps"])
        return (x.astype(np.float64) @ self.lm_head.astype(np.float64).T).astype(np.float32)[0]

    def forward_prompt(self, tokens, start_pos=0, all_logits=False):
        """`forward_token` over a whole prompt (or one chunk of it, at positions start_pos ..) in many-row form: the
        same composition, linears through `linear_rows`, attention through `attn_prompt` over this oracle's cache
        (which it extends, so `forward_token` / further chunks continue from it). Returns the last position's logits
        (all positions' with all_logits)."""
        c = self.cfg
        H, KV, D = c["heads"], c["kv_heads"], c["head_dim"]
        tokens = [int(t) for t in tokens]
        T = len(tokens)
        pos = np.arange(start_pos, start_pos + T, dtype=np.int32)
        h = self.embed[tokens].astype(np.float32)
        for li, ly in enumerate(self.layers):
            assert self.k[li].shape[0] == start_pos, "chunks must arrive in order"
            x = rmsnorm(h, ly["ln1"], c["eps"])
            if "qkv" in ly:
                qkv = linear_rows(x, ly["qkv"])
                q, k, v = qkv[:, :H * D], qkv[:, H * D:(H + KV) * D], qkv[:, (H + KV) * D:]
            else:
                q, k, v = (linear_rows(x, ly[n]) for n in ("q", "k", "v"))
            q = rope(q.reshape(T, H, D), pos, c["theta"])
            k = rope(k.reshape(T, KV, D), pos, c["theta"])
            v = np.ascontiguousarray(v).reshape(T, KV, D)
            if self.kv_hook is not None:
                k, v = self.kv_hook(li, start_pos, k, v)
            self.k[li] = np.concatenate([self.k[li], k], 0)
            self.v[li] = np.concatenate([self.v[li], v], 0)
            a = attn_prompt(q, self.k[li], self.v[li], start_pos, c.get("window", 0)).reshape(T, H * D)
            h = h + linear_rows(a, ly["o"])
            x = rmsnorm(h, ly["ln2"], c["eps"])
            if "gate_up" in ly:
                gu = linear_rows(x, ly["gate_up"]).reshape(T, -1, 2, 16)
                g = np.ascontiguousarray(gu[:, :, 0, :]).reshape(T, -1)
                u = np.ascontiguousarray(gu[:, :, 1, :]).reshape(T, -1)
            else:
                g, u = linear_rows(x, ly["gate"]), linear_rows(x, ly["up"])
            h = h + linear_rows(silu_mul(g, u), ly["down"])
        if not all_logits:
            h = h[-1:]
        x = rmsnorm(h, self.norm, c["eps"])
        lg = (x.astype(np.float64) @ self.lm_head.astype(np.float64).T).astype(np.float32)
        return lg if all_logits else lg[0]
