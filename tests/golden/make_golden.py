"""Generate tests/golden/*.npz by RUNNING the reference's own Python code in this container.

The reference package cannot be imported (its __init__ chain needs neural_compressor / peft /
BesTLA), so this script lifts individual pure-torch functions out of the files
under /root/reference with `ast` at generation time and executes them unmodified. No reference
source is written into this repository — only the numeric inputs/outputs.

Pinned here:
  unpack_weight            intel_extension_for_transformers/transformers/llm/quantization/utils.py:82-125
  set_weights_bias         .../llm/quantization/nn/modules.py:195-262  (what reaches qbits.repack_quantized_weight)
  quant_weight_w_scale     .../llm/quantization/nn/modules.py:264-295
  convert_idx              intel_extension_for_transformers/qbits/qbits_ut/test_packq.py:22-28
  HF ops (the reference runs stock HF modules between the linears):
  LlamaRMSNorm, apply_rotary_pos_emb, SiLU*mul, gelu_new / gelu   (transformers installed here)

Run:  python tests/golden/make_golden.py     (needs /root/reference; output is committed)
"""
import ast
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/intel_extension_for_transformers"
OUT = os.path.dirname(os.path.abspath(__file__))


def lift(path, name, cls=None):
    """Return the source text of function `name` (optionally a method of `cls`) from `path`."""
    src = open(path).read()
    tree = ast.parse(src)
    nodes = tree.body
    if cls is not None:
        nodes = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    fn = next(n for n in nodes if isinstance(n, ast.FunctionDef) and n.name == name)
    text = ast.get_source_segment(src, fn)
    import textwrap

    return textwrap.dedent(text)


def gen_unpack_weight():
    ns = {"torch": torch}
    exec(lift(f"{REF}/transformers/llm/quantization/utils.py", "unpack_weight"), ns)
    out = {}
    g = torch.Generator().manual_seed(1234)
    for tag, (K, N, group, bits, sym) in {
        "b4_sym": (256, 64, 32, 4, True),
        "b4_asym": (256, 64, 32, 4, False),
        "b4_asym_g128": (512, 48, 128, 4, False),
        "b8_sym": (64, 32, 32, 8, True),
        "b8_asym": (64, 32, 32, 8, False),
    }.items():
        pack = 32 // bits
        G = K // group
        qweight = torch.randint(-(2**31), 2**31 - 1, (K // pack, N), generator=g, dtype=torch.int64).to(torch.int32)
        qzeros = torch.randint(-(2**31), 2**31 - 1, (G, N // pack), generator=g, dtype=torch.int64).to(torch.int32)
        if bits == 8:
            # keep stored zero-points away from the uint8/int8 wrap-around edge of the "+1" quirk
            qzeros = (qzeros & 0x7E7E7E7E).to(torch.int32)
        scales = torch.rand(G, N, generator=g)
        cfg = types.SimpleNamespace(sym=sym, bits=bits)
        w, s, z = ns["unpack_weight"](qweight, scales, qzeros, cfg)
        w = w.view(-1, w.shape[-1])
        out[f"{tag}_qweight"] = qweight.numpy()
        out[f"{tag}_qzeros"] = qzeros.numpy()
        out[f"{tag}_scales"] = scales.numpy()
        out[f"{tag}_w"] = w.numpy().astype(np.int16)
        out[f"{tag}_z"] = z.numpy().astype(np.int16)
        out[f"{tag}_meta"] = np.array([K, N, group, bits, int(sym)])
    np.savez_compressed(f"{OUT}/unpack_weight.npz", **out)


def gen_set_weights_bias():
    """Capture the exact tensors QuantizedLinearQBits.set_weights_bias hands to repack_quantized_weight."""
    captured = {}

    class _QB:
        @staticmethod
        def repack_quantized_weight(qw, sc, zp, gidx, wt, st, ct, asym, blocksize):
            captured.update(qw=qw.clone(), sc=sc.clone(), zp=zp.clone(), gidx=gidx.clone(), wt=wt, st=st, ct=ct,
                            asym=asym, blocksize=blocksize)
            return torch.zeros(1, dtype=torch.int8)

    class _Params:
        def __init__(self, **kw):
            self.kw = kw

    ns = {"torch": torch, "qbits": _QB, "ParamsQBits": _Params}
    exec(lift(f"{REF}/transformers/llm/quantization/nn/modules.py", "set_weights_bias", "QuantizedLinearQBits"), ns)
    fn = ns["set_weights_bias"]
    out = {}
    g = torch.Generator().manual_seed(4321)
    K, N, group = 128, 32, 32
    for tag, (method, sym, desc_act) in {
        "rtn_sym": ("rtn", True, False),
        "rtn_asym": ("rtn", False, False),
        "gptq_desc_act": ("gptq", False, True),
    }.items():
        int_weight = torch.randint(0, 16, (K, N), generator=g, dtype=torch.int8)
        scales = torch.rand(K // group, N, generator=g).to(torch.float16)
        zeros = torch.randint(1, 17, (K // group, N), generator=g, dtype=torch.int8)
        perm = torch.randperm(K, generator=g)
        g_idx = (perm // group).to(torch.int32) if desc_act else torch.zeros(K, dtype=torch.int32)
        qcfg = types.SimpleNamespace(
            quant_method=types.SimpleNamespace(value=method), desc_act=desc_act, static_groups=False,
            group_size=group, bits=4, weight_dtype="int4_clip", sym=sym, scale_dtype="fp32", compute_dtype="fp32")
        self = types.SimpleNamespace(blocksize=group, scheme="sym" if sym else "asym", compress_statistics=False,
                                     weight_dtype="int4_clip", scale_dtype="fp32")
        fn(self, int_weight.clone(), scales.clone(), zeros.clone(), g_idx.clone(), qcfg, bias=None)
        out[f"{tag}_in_w"] = int_weight.numpy().astype(np.int16)
        out[f"{tag}_in_s"] = scales.float().numpy()
        out[f"{tag}_in_z"] = zeros.numpy().astype(np.int16)
        out[f"{tag}_in_gidx"] = g_idx.numpy()
        out[f"{tag}_out_w"] = captured["qw"].numpy().astype(np.int16)
        out[f"{tag}_out_s"] = captured["sc"].numpy()
        out[f"{tag}_out_z"] = captured["zp"].numpy().astype(np.int16)
        out[f"{tag}_out_gidx"] = captured["gidx"].numpy()
        out[f"{tag}_out_asym"] = np.array(int(captured["asym"]))
    np.savez_compressed(f"{OUT}/set_weights_bias.npz", **out)


def gen_quant_weight_w_scale():
    ns = {"torch": torch}
    exec(lift(f"{REF}/transformers/llm/quantization/nn/modules.py", "quant_weight_w_scale", "QuantizedLinearQBits"), ns)
    fn = ns["quant_weight_w_scale"]
    g = torch.Generator().manual_seed(99)
    N, K, group = 24, 160, 64  # tail group of 32
    G = (K + group - 1) // group
    w = torch.randn(N, K, generator=g)
    scale = torch.rand(N, G, generator=g) * 0.1 + 0.01
    zp = torch.randint(0, 16, (N, G), generator=g).to(torch.uint8)
    q_asym = fn(None, w.clone(), scale, zp, group_size=group)
    q_sym = fn(None, w.clone(), scale, None, group_size=group)
    np.savez_compressed(f"{OUT}/quant_weight_w_scale.npz", w=w.numpy(), scale=scale.numpy(), zp=zp.numpy(),
                        q_asym=q_asym.numpy(), q_sym=q_sym.numpy(), group=np.array(group))


def gen_convert_idx():
    ns = {"torch": torch}
    exec(lift(f"{REF}/qbits/qbits_ut/test_packq.py", "convert_idx"), ns)
    out = {}
    # the reference test's own g_idx (test_packq.py:59-61) and a shuffled GPTQ-style one
    k, bs = 512, 128
    g_idx = torch.arange(k // bs, dtype=torch.int).repeat(bs)
    out["ut_gidx"] = g_idx.numpy()
    out["ut_ret"] = ns["convert_idx"](g_idx, k, bs).numpy()
    g = torch.Generator().manual_seed(7)
    perm = torch.randperm(256, generator=g)
    g_idx2 = (perm // 32).to(torch.int)
    out["rnd_gidx"] = g_idx2.numpy()
    out["rnd_ret"] = ns["convert_idx"](g_idx2, 256, 32).numpy()
    np.savez_compressed(f"{OUT}/convert_idx.npz", **out)


def gen_hf_ops():
    from transformers.models.llama.modeling_llama import LlamaRMSNorm, apply_rotary_pos_emb
    from transformers.activations import ACT2FN

    try:
        from transformers.models.llama.modeling_llama import LlamaRotaryEmbedding
        from transformers import LlamaConfig
    except Exception:  # pragma: no cover
        LlamaRotaryEmbedding = None
    g = torch.Generator().manual_seed(2024)
    out = {}
    d = 256
    x = torch.randn(5, d, generator=g) * 3
    norm = LlamaRMSNorm(d, eps=1e-5)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.1 * torch.randn(d, generator=g))
        out["rms_x"], out["rms_w"], out["rms_eps"] = x.numpy(), norm.weight.numpy().copy(), np.array(1e-5)
        out["rms_y"] = norm(x).numpy()
        # rope: [batch=1, heads, tokens, D]
        H, KV, T, D = 4, 2, 6, 64
        q = torch.randn(1, H, T, D, generator=g)
        k = torch.randn(1, KV, T, D, generator=g)
        pos = torch.tensor([[0, 1, 2, 17, 300, 4095]])
        cfg = LlamaConfig(hidden_size=H * D, num_attention_heads=H, num_key_value_heads=KV, rope_theta=10000.0,
                          max_position_embeddings=8192)
        try:
            cfg.rope_parameters = {"rope_type": "default", "rope_theta": 10000.0}
        except Exception:
            pass
        rot = LlamaRotaryEmbedding(config=cfg)
        cos, sin = rot(q, pos)
        qr, kr = apply_rotary_pos_emb(q, k, cos, sin)
        out["rope_q"], out["rope_k"], out["rope_pos"] = q.numpy(), k.numpy(), pos.numpy()
        out["rope_qr"], out["rope_kr"] = qr.numpy(), kr.numpy()
        gate = torch.randn(3, 512, generator=g) * 2
        up = torch.randn(3, 512, generator=g)
        out["silu_gate"], out["silu_up"] = gate.numpy(), up.numpy()
        out["silu_y"] = (ACT2FN["silu"](gate) * up).numpy()
        out["gelu_x"] = gate.numpy()
        out["gelu_new_y"] = ACT2FN["gelu_new"](gate).numpy()
        out["gelu_y"] = ACT2FN["gelu"](gate).numpy()
    import transformers

    out["transformers_version"] = np.array(transformers.__version__)
    np.savez_compressed(f"{OUT}/hf_ops.npz", **out)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    gen_unpack_weight()
    gen_set_weights_bias()
    gen_quant_weight_w_scale()
    gen_convert_idx()
    gen_hf_ops()
    print("golden fixtures written to", OUT)
