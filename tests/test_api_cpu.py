"""CPU-side checks of the ITREX-compatible Python API (no GPU, no HIP calls): config semantics mirrored from the
reference's own unit tests, the optimum-format decode against the golden fixtures generated from the reference's
`unpack_weight`, and the module surface `_replace_linear` relies on."""
import json
import os

import numpy as np
import pytest
import torch

from intel_extension_for_transformers_amd.transformers import (AwqConfig, GPTQConfig, RtnConfig, TeqConfig,
                                                                  AutoRoundConfig, WeightOnlyQuantConfig)
from intel_extension_for_transformers_amd.transformers.llm.quantization import utils as qutils


def test_woq_config_diff_dict(tmp_path):
    """reference tests/CI/test_weight_only.py:93-103."""
    config = RtnConfig(bits=4, weight_dtype="int4", group_size=32)
    assert config.to_diff_dict() == {"weight_dtype": "int4"}
    json.loads(config.to_json_string())
    config.to_json_file(str(tmp_path / "config.json"))
    assert json.load(open(tmp_path / "config.json")) == {"weight_dtype": "int4"}
    assert "RtnConfig" in repr(config)


def test_woq_config_post_init_runtime():
    """reference tests/CI/test_weight_only.py:105-115."""
    config = RtnConfig(bits=4, weight_dtype="fp4", compute_dtype="int8", scheme="asym", scale_dtype="fp8")
    config.post_init_runtime()
    d = config.to_dict()
    assert (d["weight_dtype"], d["compute_dtype"], d["scheme"], d["scale_dtype"]) == ("fp4_e2m1", "fp32", "sym", "fp32")


def test_config_defaults_and_validators():
    """reference utils/config.py:794-842 defaults; :277-372 post_init_cpu rules."""
    c = RtnConfig()
    assert (c.bits, c.group_size, c.sym, c.scheme) == (4, 32, True, "sym")
    assert c.llm_int8_skip_modules == ["lm_head", "transformer.output_layer", "embed_out"]
    c.post_init_cpu()
    assert (c.weight_dtype, c.compute_dtype, c.scale_dtype, c.use_neural_speed) == ("int4_clip", "fp32", "fp32", False)
    bad = RtnConfig(sym=False, scale_dtype="bf16")
    with pytest.raises(ValueError):
        bad.post_init_cpu()  # asym needs fp32 scales on the CPU backend (:360-370)
    bad.post_init_hip()      # ... but not on the HIP backend
    with pytest.raises(ValueError):
        RtnConfig(bits=3).post_init_cpu()
    with pytest.raises(ValueError):
        RtnConfig(group_size=48).post_init_hip()
    with pytest.raises(ValueError):
        RtnConfig(compute_dtype="fp64").post_init_cpu()
    assert WeightOnlyQuantConfig is RtnConfig
    assert AwqConfig().scheme == "asym" and AwqConfig(zero_point=False).scheme == "sym"
    assert GPTQConfig(desc_act=True).desc_act and GPTQConfig().quant_method.value == "gptq"
    assert TeqConfig().quant_method.value == "teq" and AutoRoundConfig().group_size == 128
    assert RtnConfig(compute_dtype=torch.bfloat16).compute_dtype == "bf16"


def test_config_roundtrip_and_update(tmp_path):
    c = GPTQConfig(bits=4, group_size=128, sym=False, desc_act=True)
    c2 = GPTQConfig.from_dict(c.to_dict())
    assert c2.to_dict() == c.to_dict()
    assert c.update(group_size=64, nonsense=1) == {"nonsense": 1} and c.group_size == 64
    c.save_pretrained(str(tmp_path))
    saved = json.load(open(tmp_path / "quantize_config.json"))
    assert saved["quant_method"] == "gptq" and saved["desc_act"] is True
    c.remove_redundant_parameters()
    assert not hasattr(c, "tokenizer") and not hasattr(c, "scheme")


@pytest.mark.parametrize("tag", ["b4_sym", "b4_asym", "b4_asym_g128", "b8_sym", "b8_asym"])
def test_unpack_weight_matches_reference_golden(golden_dir, tag):
    """tests/golden/unpack_weight.npz was produced by running the reference's own unpack_weight
    (llm/quantization/utils.py:82-125) in the build container (tests/golden/make_golden.py). Bit-exact."""
    g = np.load(os.path.join(golden_dir, "unpack_weight.npz"))
    K, N, group, bits, sym = [int(v) for v in g[tag + "_meta"]]

    class Cfg:
        pass

    Cfg.bits, Cfg.sym = bits, bool(sym)
    w, s, z = qutils.unpack_weight(torch.from_numpy(g[tag + "_qweight"]), torch.from_numpy(g[tag + "_scales"]),
                                   torch.from_numpy(g[tag + "_qzeros"]), Cfg)
    assert np.array_equal(w.numpy().astype(np.int16), g[tag + "_w"])
    assert np.array_equal(z.numpy().astype(np.int16), g[tag + "_z"])


@pytest.mark.parametrize("sym", [True, False])
def test_pack_unpack_roundtrip(sym):
    torch.manual_seed(0)
    K, N, G = 96, 40, 3

    class Cfg:
        bits = 4

    Cfg.sym = sym
    w = torch.randint(0, 16, (K, N), dtype=torch.int8)
    z = torch.randint(1, 17, (G, N), dtype=torch.int8)
    s = torch.rand(G, N)
    qw, s16, qz = qutils.pack_weight(w, s, z)
    assert qw.dtype == torch.int32 and qw.shape == (K // 8, N) and qz.shape == (G, N // 8) and s16.dtype == torch.float16
    w2, _, z2 = qutils.unpack_weight(qw, s16, qz, Cfg)
    assert torch.equal(w, w2) and torch.equal(z, z2)


@pytest.mark.parametrize("tag", ["rtn_sym", "rtn_asym"])
def test_signed_nibble_matches_reference_golden(golden_dir, tag):
    """(t - 8) * 16 // 16 on int8 (modules.py:225-227), including the 16 -> -8 wrap of un-biased zero points."""
    from intel_extension_for_transformers_amd.transformers.llm.quantization.nn.modules import _signed_nibble

    g = np.load(os.path.join(golden_dir, "set_weights_bias.npz"))
    out = _signed_nibble(torch.from_numpy(g[tag + "_in_w"].astype(np.int8)))
    assert np.array_equal(out.numpy().astype(np.int16), g[tag + "_out_w"])
    if int(g[tag + "_out_asym"]):
        z = _signed_nibble(torch.from_numpy(g[tag + "_in_z"].astype(np.int8)))
        assert np.array_equal(z.numpy().astype(np.int16), g[tag + "_out_z"])


def test_module_surface_without_gpu():
    """The class contract `_replace_linear` relies on (reference modules.py:94-111, utils.py:364-366). Constructing
    the module needs no GPU; running it does, and must fail loudly rather than fall back."""
    from intel_extension_for_transformers_amd.transformers.llm.quantization.nn.modules import (ParamsQBits,
                                                                                               QuantizedLinearQBits)

    m = QuantizedLinearQBits(64, 32, bias=True, compute_dtype="fp32", weight_dtype="int4_clip", bits=4,
                             scale_dtype="fp32", blocksize=32, scheme="sym")
    assert isinstance(m, torch.nn.Linear) and isinstance(m.weight, ParamsQBits) and m.n_pack == 8
    for name in ("set_weights_bias", "set_fp_weights_bias", "recover_qparms", "forward"):
        assert callable(getattr(m, name))
    m.requires_grad_(False)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 64))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            m.set_fp_weights_bias(torch.zeros(32, 64))


def test_replace_linear_skips_and_requires_gpu():
    from intel_extension_for_transformers_amd.transformers.llm.quantization.utils import convert_to_quantized_model

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.linear = torch.nn.Linear(32, 2)
            self.lm_head = torch.nn.Linear(32, 2)

    cfg = RtnConfig(bits=4, group_size=32)
    cfg.post_init_hip()
    with pytest.raises(RuntimeError):
        convert_to_quantized_model(M(), cfg, device="cpu")
    with pytest.raises(NotImplementedError):
        g = GPTQConfig()
        g.post_init_hip()
        convert_to_quantized_model(M(), g, device="cuda")


# ---- NeuralChat serving seam: configuration surface (no GPU needed) ----------------------------------------------
def test_neural_chat_config_surface():
    """Field names / defaults of the reference's neural_chat/config.py for the fields this path reads
    (GenerationConfig :400-423, PipelineConfig :466-517) and the optimisation-config type check (:511-515)."""
    from intel_extension_for_transformers_amd.neural_chat import GenerationConfig, PipelineConfig
    from intel_extension_for_transformers_amd.transformers import MixedPrecisionConfig, RtnConfig

    g = GenerationConfig()
    assert (g.temperature, g.top_k, g.top_p, g.repetition_penalty, g.num_beams, g.max_new_tokens, g.do_sample,
            g.return_stats, g.format_version) == (0.1, 40, 0.75, 1.1, 1, 256, True, False, "v2")
    p = PipelineConfig(device="cuda")
    assert p.model_name_or_path == "Intel/neural-chat-7b-v3-1"
    assert isinstance(p.optimization_config, MixedPrecisionConfig) and p.optimization_config.dtype == "float16"
    q = PipelineConfig(model_name_or_path="meta-llama/Llama-2-7b-chat-hf", device="cuda",
                       optimization_config=RtnConfig(bits=4, group_size=128))
    assert q.optimization_config.bits == 4 and q.loading_config.use_cache
    with pytest.raises(AssertionError, match="optimization_config"):
        PipelineConfig(device="cuda", optimization_config=object())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        PipelineConfig(device="auto")  # this container has no GPU: the backend refuses instead of picking "cpu"


def test_neural_chat_prompt_templates():
    from intel_extension_for_transformers_amd.neural_chat.prompts import get_conv_template

    c = get_conv_template("neural-chat-7b-v3")
    c.append_message(c.roles[0], "Tell me about MI355X.")
    c.append_message(c.roles[1], None)
    p = c.get_prompt()
    assert p.startswith("### System:\n- You are a helpful assistant chatbot trained by Intel.")
    assert p.endswith("### User:\nTell me about MI355X.\n### Assistant:\n")
    c = get_conv_template("llama-2")
    c.append_message(c.roles[0], "hi")
    c.append_message(c.roles[1], None)
    assert c.get_prompt() == "[INST] hi [/INST]"
    assert get_conv_template("llama-2").messages == []  # templates are copied, not shared


@pytest.mark.parametrize("sym", [True, False])
def test_pack_unpack_weight_8bit_roundtrip(sym):
    """On-disk 8-bit convention of reference utils.py:82-125: values stored in the unsigned domain (q + 128), zeros
    as zp - 1; unpack returns signed int8. pack_weight (the save path) inverts it for what save_low_bit feeds it."""
    import types

    import torch

    from intel_extension_for_transformers_amd.transformers.llm.quantization.utils import pack_weight, unpack_weight

    g = torch.Generator().manual_seed(0)
    K, N, G = 64, 20, 2
    q = torch.randint(-128, 128, (K, N), generator=g, dtype=torch.int16)
    z = None if sym else torch.randint(-127, 128, (G, N), generator=g, dtype=torch.int16)
    s = torch.rand(G, N, generator=g) + 0.5
    qweight, _, qzeros = pack_weight(q + 128, s, None if z is None else z + 128, bits=8)
    assert qweight.shape == (K // 4, N) and qweight.dtype == torch.int32
    cfg = types.SimpleNamespace(bits=8, sym=sym)
    w, _, zeros = unpack_weight(qweight, s, qzeros, cfg)
    assert w.dtype == torch.int8 and torch.equal(w.to(torch.int16), q)
    if not sym:
        assert torch.equal(zeros.to(torch.int16), z)


def test_awq_gemm_pack_unpack_matches_writer_definition():
    """AutoAWQ "GEMM" packing: nibble i of word c (along N) = column 8c + (0,2,4,6,1,3,5,7)[i], zeros stored as they
    are. unpack_awq_gemm inverts it; checked against the writer's loop form on random tensors."""
    import torch

    from intel_extension_for_transformers_amd.transformers.llm.quantization.utils import pack_awq_gemm, unpack_awq_gemm

    g = torch.Generator().manual_seed(1)
    K, N, G = 32, 40, 2
    q = torch.randint(0, 16, (K, N), generator=g, dtype=torch.int8)
    z = torch.randint(0, 16, (G, N), generator=g, dtype=torch.int8)
    s = torch.rand(G, N, generator=g)
    order = [0, 2, 4, 6, 1, 3, 5, 7]
    want = torch.zeros(K, N // 8, dtype=torch.int64)
    for col in range(N // 8):          # the writer's definition, verbatim in loop form
        for i in range(8):
            want[:, col] |= q[:, col * 8 + order[i]].to(torch.int64) << (i * 4)
    want = torch.where(want >= 2 ** 31, want - 2 ** 32, want).to(torch.int32)
    qw, qz = pack_awq_gemm(q, z)
    assert torch.equal(qw, want)
    w, _, zz = unpack_awq_gemm(qw, s, qz)
    assert torch.equal(w, q) and torch.equal(zz, z)


def test_awq_gemm_known_answer_words():
    """External pin of the AutoAWQ "GEMM" nibble order: a hand-written 8 x 8 tensor and the int32
    words an AutoAWQ writer produces for it, written out as LITERALS — nibble i of a word holds column
    (0, 2, 4, 6, 1, 3, 5, 7)[i] (AutoAWQ `WQLinear_GEMM.from_linear`: `order_map = [0, 2, 4, 6, 1, 3, 5, 7]`,
    `qweight[:, col] |= intweight[:, col * 8 + order_map[i]] << (i * 4)`). Nothing below is produced by this repo's
    `pack_awq_gemm`; the GPTQ-style order (nibble i = column i) would read row 0 as 0xECA86420 and fail."""
    from intel_extension_for_transformers_amd.transformers.llm.quantization.utils import unpack_awq_gemm

    # w[k][n] = (2 n + 3 k) mod 16, k, n = 0..7
    w = torch.tensor([[(2 * n + 3 * k) % 16 for n in range(8)] for k in range(8)], dtype=torch.int8)
    # row 0 = [0,2,4,6,8,10,12,14]: nibbles 0..7 = columns 0,2,4,6,1,3,5,7 = 0,4,8,c,2,6,a,e -> 0xEA62C840
    # row 1 = [3,5,7,9,11,13,15,1]: 3,7,b,f,5,9,d,1 -> 0x1D95FB73;  row 2 = [6,8,10,12,14,0,2,4]: 6,a,e,2,8,c,0,4 -> 0x40C82EA6
    words = [0xEA62C840, 0x1D95FB73, 0x40C82EA6, 0x73FB51D9, 0xA62E840C, 0xD951B73F, 0x0C84EA62, 0x3FB71D95]
    for k in range(8):  # the literals against the definition, spelled out (guards the table above against typos)
        cols = [0, 2, 4, 6, 1, 3, 5, 7]
        assert words[k] == sum(((2 * cols[i] + 3 * k) % 16) << (4 * i) for i in range(8)), k
    qweight = torch.tensor([[x - (1 << 32) if x >= (1 << 31) else x] for x in words], dtype=torch.int32)  # [K = 8, N/8 = 1]
    # zero points of one group, same packing: z[n] = 15 - n -> nibbles f,d,b,9,e,c,a,8 -> 0x8ACE9BDF
    qzeros = torch.tensor([[0x8ACE9BDF - (1 << 32)]], dtype=torch.int32)
    scales = torch.ones(1, 8)
    got_w, _, got_z = unpack_awq_gemm(qweight, scales, qzeros)
    assert torch.equal(got_w, w)
    assert got_z.tolist() == [[15 - n for n in range(8)]]


def test_library_exports_every_symbol_the_headers_declare():
    """The C-ABI library loads without a GPU and exports every `WOQ_API` function of include/woq_hip.h (no compute
    call is made); the ctypes binding's EXPORTS list is exactly that set, and the header structs have the sizes
    the binding assumes."""
    import ctypes
    import os
    import re

    from intel_extension_for_transformers_amd import _lib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "woq_hip.h")).read()
    declared = set(re.findall(r"WOQ_API[^;(]*?\b(woq_\w+)\s*\(", header))
    assert len(declared) >= 30
    assert declared == set(_lib.EXPORTS)
    # round 6: the boundary is frozen — no lab switch or measurement hook in the public header
    assert not [n for n in declared if re.search(r"persist|prefetch|mall_probe|time_|_twin|attn_chunk", n)]
    exp_header = open(os.path.join(root, "include", "woq_hip_experimental.h")).read()
    experimental = set(re.findall(r"WOQ_API[^;(]*?\b(woq_\w+)\s*\(", exp_header))
    assert experimental == set(_lib.EXPERIMENTAL_EXPORTS) and not (experimental & declared)
    declared = declared | experimental
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libwoq_hip.so not built (python __graft_entry__.py)")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), name
    lib.woq_abi_version.restype = ctypes.c_int
    assert lib.woq_abi_version() == 4  # WOQ_ABI_VERSION of include/woq_hip.h
    lib.woq_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.woq_last_error(), bytes)
    assert ctypes.sizeof(_lib.BlobHeader) == 256 and ctypes.sizeof(_lib.EngineConfig) == 4 * 16


def test_table_digit_planes_reproduce_the_tables():
    """nf4 / fp4 at decode row counts (round 4): the kernels read table[c] * S as balanced base-256 int8 digit planes
    looked up with v_perm_b32 (csrc/woq_gemv_common.h). The planes the library hands its kernels, rebuilt here on the
    host (no device): digits in range, sum_j d_j 256^j / S == the oracle's table — exactly for both fp4 tables, to
    2^-23 of the largest entry for nf4 at compute fp32 (three planes) and 2^-16 for the two-plane form — and the byte
    lookup itself (two 8-entry permutes + a select by bit 3 of the code) emulated on every code."""
    import ctypes
    import os

    from intel_extension_for_transformers_amd import _lib
    from oracle import woq_oracle as orc

    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libwoq_hip.so not built (python __graft_entry__.py)")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.woq_table_digit_planes.restype = ctypes.c_int
    lib.woq_table_digit_planes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]

    def perm(hi, lo, sel):  # v_perm_b32: byte i of the result = byte sel_i of {hi:lo}; 0x0c = 0x00, >= 0x0d = 0xff
        src = [(lo >> (8 * i)) & 0xff for i in range(4)] + [(hi >> (8 * i)) & 0xff for i in range(4)]
        out = 0
        for i in range(4):
            b = (sel >> (8 * i)) & 0xff
            out |= (src[b] if b < 8 else (0 if b == 12 else 0xff)) << (8 * i)
        return out

    cases = [(orc.W_NF4, 0, 3, 2.0 ** -23), (orc.W_NF4, 1, 2, 2.0 ** -16), (orc.W_NF4, 3, 2, 2.0 ** -16),
             (orc.W_FP4_E2M1, 0, 1, 0.0), (orc.W_FP4_E2M1_BNB, 0, 2, 1e-7), (orc.W_FP4_E2M1_BNB, 1, 2, 1e-7)]
    for wt, ct, ndig, tol in cases:
        planes = (ctypes.c_uint32 * 12)()
        wmul = ctypes.c_float()
        assert lib.woq_table_digit_planes(wt, ct, planes, ctypes.byref(wmul)) == ndig
        d = [[int(planes[j * 4 + i]) for i in range(4)] for j in range(3)]
        S = 16.0 / wmul.value
        lut = orc.LUTS[wt]
        for base in range(0, 16, 4):
            codes = [base, base + 1, base + 2, base + 3]
            idx = sum(c << (8 * i) for i, c in enumerate(codes))
            sel = idx & 0x07070707
            mask = perm(0, 0, (idx & 0x08080808) | 0x05050505)
            for i, c in enumerate(codes):
                v = 0
                for j in range(3):
                    word = (perm(d[j][3], d[j][2], sel) & mask) | (perm(d[j][1], d[j][0], sel) & ~mask & 0xffffffff)
                    b = (word >> (8 * i)) & 0xff
                    dj = b - 256 if b >= 128 else b
                    assert j < ndig or dj == 0
                    v += dj * 256 ** j
                assert abs(v / S - float(lut[c])) <= tol * float(np.abs(lut).max()) + 1e-12, (wt, ct, c)
    assert lib.woq_table_digit_planes(0, 0, None, None) == 0  # int4_clip is not a table type


def test_fp8_weight_dtype_config_on_the_hip_path():
    """fp8 weights at the config level (reference config.py:298-299,313,335-340): "fp8" means fp8_e4m3, bits = 8,
    symmetric only, fp32 or fp8_e8m0 ("fp8") scales; fp8_e8m0 scales are refused for integer weights."""
    from intel_extension_for_transformers_amd.transformers import RtnConfig

    c = RtnConfig(weight_dtype="fp8", scale_dtype="fp8", group_size=64)
    c.post_init_hip()
    assert (c.bits, c.weight_dtype, c.scale_dtype, c.scheme) == (8, "fp8_e4m3", "fp8_e8m0", "sym")
    c = RtnConfig(bits=8, weight_dtype="fp8_e5m2", group_size=32)
    c.post_init_hip()
    assert (c.weight_dtype, c.scale_dtype) == ("fp8_e5m2", "fp32")
    for kw in (dict(bits=8, weight_dtype="fp8_e4m3", sym=False), dict(bits=8, weight_dtype="fp8_e4m3", scale_dtype="bf16"),
               dict(bits=4, scale_dtype="fp8_e8m0")):
        with pytest.raises(ValueError):
            RtnConfig(group_size=64, **kw).post_init_hip()


def test_plain_c_client_of_the_abi(tmp_path):
    """include/woq_blob.h + include/woq_hip.h are valid C99 (what a cgo / JNI / FFI binding would include), and a
    plain-C program links against libwoq_hip.so and agrees with the library on the blob geometry of every weight
    type (examples/c_client.c, host part; its device part — quantise, dequantise, woq_linear through the C ABI alone — runs in
    tests/test_gpu_api.py::test_plain_c_client_device_leg)."""
    import os
    import shutil
    import subprocess

    from intel_extension_for_transformers_amd import _lib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None or not os.path.exists(_lib.LIB_PATH):
        pytest.skip("needs gcc and a built libwoq_hip.so")
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "c_client")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "c_client.c"), "-L" + libdir, "-lwoq_hip", "-L/opt/rocm/lib",
                    "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "host checks ok" in out
    first = out.splitlines()[0].split()
    assert first[0] == "int4" and first[1] == first[2]
    from oracle import woq_oracle as orc

    assert int(first[1]) == orc.packed_size(4096, 11008, 128, orc.F16, False, False)


def test_matmul_kbit_seam_host_logic(monkeypatch):
    """autograd/functions.py under the reference's names: the acquire-type enum values (functions.py:29-38,
    bestla_packq_impl.hpp:18-31), the QBITS_DEBUG switch read per call (:108,:203), the training form refused."""
    from intel_extension_for_transformers_amd.transformers.llm.quantization import autograd
    from intel_extension_for_transformers_amd.transformers.llm.quantization.autograd import functions as F

    assert callable(autograd.matmul_kbit) and callable(autograd.qbits_woq_linear_ref_impl)
    T = F.qbits_acquire_type
    assert [T.SIZE.value, T.BLOCKSIZE.value, T.K.value, T.N.value, T.ACT_SHUFFLE.value, T.G_IDX.value,
            T.WEI_TYPE.value, T.CMPT_TYPE.value, T.SCALE_TYPE.value] == list(range(9))
    monkeypatch.delenv("QBITS_DEBUG", raising=False)
    assert not F.qbits_debug_enabled()
    monkeypatch.setenv("QBITS_DEBUG", "NULL")
    assert not F.qbits_debug_enabled()
    monkeypatch.setenv("QBITS_DEBUG", "1")
    assert F.qbits_debug_enabled()
    with pytest.raises(NotImplementedError):
        F.matmul_kbit(torch.zeros(1, 4), torch.zeros(4, dtype=torch.int8), None, torch.zeros(1, 4), "fp32", "int4_clip",
                      "fp32", "sym", do_dequant=True)


def test_device_sampler_matches_hf_logits_processors():
    """runtime.engine.DeviceSampler (the next-token choice of engine-backed sampling requests) against Hugging Face's own
    RepetitionPenaltyLogitsProcessor / TemperatureLogitsWarper / TopKLogitsWarper / TopPLogitsWarper in HF's order, on
    the reference's NeuralChat defaults (neural_chat/config.py:400-409) and a few other settings: identical processed
    scores (same -inf pattern, same finite values), and identical draws under the same RNG state. Pure torch: runs on
    the CPU."""
    import torch
    from transformers import (LogitsProcessorList, RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper,
                              TopKLogitsWarper, TopPLogitsWarper)

    from intel_extension_for_transformers_amd.runtime.engine import DeviceSampler

    g = torch.Generator().manual_seed(0)
    vocab = 500
    for kw in (dict(do_sample=True, temperature=0.1, top_k=40, top_p=0.75, repetition_penalty=1.1),
               dict(do_sample=True, temperature=0.8, top_k=0, top_p=0.9, repetition_penalty=1.0),
               dict(do_sample=True, temperature=1.0, top_k=5, top_p=1.0, repetition_penalty=1.3),
               dict(do_sample=False, temperature=1.0, top_k=50, top_p=1.0, repetition_penalty=1.2)):
        procs = LogitsProcessorList()
        if kw["repetition_penalty"] != 1.0:
            procs.append(RepetitionPenaltyLogitsProcessor(kw["repetition_penalty"]))
        if kw["do_sample"]:
            if kw["temperature"] != 1.0:
                procs.append(TemperatureLogitsWarper(kw["temperature"]))
            if kw["top_k"]:
                procs.append(TopKLogitsWarper(kw["top_k"]))
            if kw["top_p"] < 1.0:
                procs.append(TopPLogitsWarper(kw["top_p"]))
        sampler = DeviceSampler(**kw)
        for _ in range(5):
            logits = torch.randn(vocab, generator=g) * 3
            hist = torch.randint(0, vocab, (37,), generator=g)
            want = procs(hist[None], logits[None].clone())[0]
            got = sampler.processed(logits, hist)
            assert torch.equal(torch.isinf(got), torch.isinf(want))
            fin = ~torch.isinf(want)
            assert torch.allclose(got[fin], want[fin], rtol=1e-6, atol=1e-6)
            if kw["do_sample"] and not kw["top_k"]:
                torch.manual_seed(7)
                a = torch.multinomial(want.softmax(-1), 1)
                torch.manual_seed(7)
                b = sampler(logits, hist)
                assert int(a) == int(b)
            elif kw["do_sample"]:
                # top_k set: the draw runs on the candidates only (no full-vocabulary sort); same distribution — the
                # candidate probabilities scattered back to the vocabulary equal HF's, and draws land in its support
                want_p = want.softmax(-1)
                support = set(torch.nonzero(want_p > 0).ravel().tolist())
                for seed in range(20):
                    torch.manual_seed(seed)
                    assert int(sampler(logits, hist)) in support
                counts = torch.zeros(vocab)
                torch.manual_seed(3)
                for _ in range(4000):
                    counts[int(sampler(logits, hist))] += 1
                assert (counts / 4000 - want_p).abs().max() < 0.04
            else:
                assert int(sampler(logits, hist)) == int(want.argmax())


def test_bench_configs_summary_carries_every_baseline_config():
    """bench.py's `configs_summary`: <= 600 characters, one number per BASELINE.json config, built
    from the line's own objects — checked on the round's recorded line (profiles/r05z_bench_steps20.json)."""
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    line = json.load(open(os.path.join(root, "profiles", "r05z_bench_steps20.json")))
    s = bench.configs_summary(line)
    assert s == line["configs_summary"] and len(s) <= 600
    for needle in ("c1 7B g128 20-step", "128-step", "c1 pf 4x2048", "c2 g32asym", "c2 pf 32x2048", "c1@512", "c1@2048",
                   "act-order", "c4 pf 8k chunked", "c4 Mistral 8k fp8KV", "c3 70B 1-GPU"):
        assert needle in s, needle
    assert json.loads(bench.compact_line(line))["configs_summary"] == s  # rides in the (<= 6 KB) stdout line


def test_fused_projections_must_share_their_act_order():
    """optimize_transformers fuses q / k / v (and gate / up) along N: one activation shuffle per fused blob, so the
    modules' g_idx must agree — equal -> that g_idx, none -> None, different -> RuntimeError (module path keeps the model)."""
    from intel_extension_for_transformers_amd.runtime.engine import _same_order

    a = torch.tensor([0, 1, 0, 1], dtype=torch.int32)
    b = torch.tensor([1, 0, 0, 1], dtype=torch.int32)
    part = lambda g: (None, None, None, g)  # noqa: E731
    assert _same_order([part(None), part(None)], "x") is None
    assert torch.equal(_same_order([part(a), part(a.clone()), part(a.clone())], "x"), a)
    with pytest.raises(RuntimeError, match="different act-order permutations"):
        _same_order([part(a), part(b)], "layers.0 q_proj / k_proj")
    with pytest.raises(RuntimeError, match="different act-order permutations"):
        _same_order([part(a), part(None)], "layers.0 gate_proj / up_proj")
