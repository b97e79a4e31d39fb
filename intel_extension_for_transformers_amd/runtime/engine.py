"""Python handle of the native batch-1 decode engine (csrc/woq_engine.hip).

The engine is the MI355X counterpart of `ipex.optimize_transformers(qmodel, ...)`, the reference's own
precedent for swapping HF's RMSNorm / RoPE / MLP glue for fused device kernels after `from_pretrained`
(docs/weightonlyquant.md:199-202, README.md:290): quantised-linear replacement happens at load, the fused
decode path is an optional post-pass over the same WQH1 blobs. torch is plumbing here (device memory,
streams); every kernel is in libwoq_hip.so.
"""
import ctypes
import os

import torch

from .. import _lib as L
from .. import qbits


def build_rope_tables(max_pos, head_dim, theta=10000.0, device="cuda"):
    """cos/sin [max_pos, head_dim/2] fp32, HF LlamaRotaryEmbedding's default parameterisation."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    ang = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv_freq[None, :]
    return ang.cos().contiguous().to(device), ang.sin().contiguous().to(device)


def fuse_gate_up(gate, up):
    """[*, I] x 2 -> [*, 2I] with 16-column tiles interleaved (gate tile t, up tile t, gate tile t+1, ...), the
    column order the fused SiLU*mul GEMV epilogue expects."""
    lead = gate.shape[:-1]
    inter = gate.shape[-1]
    assert inter % 16 == 0, "intermediate size must be a multiple of 16"
    g = gate.reshape(*lead, inter // 16, 1, 16)
    u = up.reshape(*lead, inter // 16, 1, 16)
    return torch.cat([g, u], dim=-2).reshape(*lead, 2 * inter).contiguous()


def _kv_code(dt):
    """KV-cache storage dtype -> ABI code: fp16 / bf16, or fp8 (torch.float8_e4m3fn or the string "fp8_e4m3")."""
    if dt in ("fp8", "fp8_e4m3") or dt == getattr(torch, "float8_e4m3fn", None):
        return L.FP8_E4M3
    return L.torch_dtype_code(dt)


class WoqDecoderEngine:
    """Owns the native engine plus the torch tensors whose device memory it points at."""

    def __init__(self, hidden, inter, heads, kv_heads, head_dim, layers, vocab, max_ctx=2048, rms_eps=1e-5,
                 rope_theta=10000.0, kv_dtype=torch.float16, tp_rank=0, tp_size=1, device=None, max_batch=1,
                 attn_splits=0, sliding_window=0, attn_grouped=False):
        L.require_gpu()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.cfg = L.EngineConfig(hidden=hidden, inter=inter, heads=heads, kv_heads=kv_heads, head_dim=head_dim,
                                  layers=layers, vocab=vocab, max_ctx=max_ctx, rms_eps=rms_eps, rope_theta=rope_theta,
                                  tp_rank=tp_rank, tp_size=tp_size, kv_dtype=_kv_code(kv_dtype))
        self.kv_dtype = kv_dtype
        self.cfg.reserved[0] = int(max_batch)
        self.cfg.reserved[1] = int(attn_splits)
        self.cfg.reserved[2] = int(sliding_window or 0)
        self.max_batch = int(max_batch)
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            L.check(L.lib().woq_engine_create(ctypes.byref(self.cfg), ctypes.byref(self._h)))
        if attn_grouped:
            self.set_attn_grouped(True)
        self.token = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.pos = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.logits = torch.zeros(vocab, dtype=torch.float32, device=self.device)
        self.hidden = torch.zeros(hidden, dtype=torch.float32, device=self.device)
        L.check(L.lib().woq_engine_bind_io(self._h, self.token.data_ptr(), self.pos.data_ptr(),
                                           self.logits.data_ptr(), self.hidden.data_ptr()))
        self._keep = []  # tensors the engine holds raw pointers to
        self.layer_tensors, self.head_tensors = {}, {}  # the same, by name (inspection / parity checks)
        self._allreduce_cb = None
        self._comm = None
        self.captured = False
        # hipGraph capture is not allowed on the legacy null stream torch uses by default
        self._stream = torch.cuda.Stream(device=self.device)

    def __del__(self):
        try:
            if self._h:
                L.lib().woq_engine_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_layer(self, idx, qkv_blob, o_blob, gate_up_blob, down_blob, ln1, ln2):
        ln1 = ln1.to(self.device, torch.float32).contiguous()
        ln2 = ln2.to(self.device, torch.float32).contiguous()
        w = L.LayerWeights(qkv_blob=qkv_blob.data_ptr(), o_blob=o_blob.data_ptr(),
                           gate_up_blob=gate_up_blob.data_ptr(), down_blob=down_blob.data_ptr(),
                           ln1=ln1.data_ptr(), ln2=ln2.data_ptr(), qkv_hdr=qbits.header_of(qkv_blob),
                           o_hdr=qbits.header_of(o_blob), gate_up_hdr=qbits.header_of(gate_up_blob),
                           down_hdr=qbits.header_of(down_blob))
        L.check(L.lib().woq_engine_set_layer(self._h, idx, ctypes.byref(w)))
        self._keep += [qkv_blob, o_blob, gate_up_blob, down_blob, ln1, ln2]
        self.layer_tensors[idx] = dict(qkv=qkv_blob, o=o_blob, gate_up=gate_up_blob, down=down_blob, ln1=ln1, ln2=ln2)

    def set_head(self, embed, final_norm, lm_head, cos=None, sin=None):
        embed = embed.to(self.device).contiguous()
        lm_head = lm_head.to(self.device).contiguous()
        final_norm = final_norm.to(self.device, torch.float32).contiguous()
        if cos is None:
            cos, sin = build_rope_tables(self.cfg.max_ctx, self.cfg.head_dim, self.cfg.rope_theta, self.device)
        L.check(L.lib().woq_engine_set_head(self._h, embed.data_ptr(), L.torch_dtype_code(embed.dtype),
                                            final_norm.data_ptr(), lm_head.data_ptr(),
                                            L.torch_dtype_code(lm_head.dtype), cos.data_ptr(), sin.data_ptr()))
        self._keep += [embed, lm_head, final_norm, cos, sin]
        self.head_tensors = dict(embed=embed, norm=final_norm, lm_head=lm_head)

    def reset(self, token=0, pos=0):
        self.token.fill_(int(token))
        self.pos.fill_(int(pos))

    def step(self, greedy=True):
        L.check(L.lib().woq_engine_step(self._h, int(greedy), L.stream_ptr()))

    # How a burst of decode steps is issued: "graph" = hipGraphLaunch of the captured step on the CALLER'S stream (no
    # host work per token: immune to host jitter), "eager" = `n` steps issued back to back by one native call
    # (`woq_engine_steps`; ~0.34 ms of launch calls per 7B token, ahead of the device). Same kernels, same device time:
    # 1.04 ms per Llama-2-7B token either way (profiles/r04ab_stream_mode_probe.txt). What made the replayed graph 12 %
    # slower through round 3 (1.17 ms) was not the graph but the way it was launched: on the engine's capture stream,
    # bracketed by `wait_stream` in both directions — after a cross-stream event wait every kernel boundary of the
    # following launches on that stream costs ~1 us more. Round 4 first switched to eager bursts (profiles/r04g_*), then
    # found the cause; the graph is launched on the current stream now. WOQ_ENGINE_LAUNCH=graph|eager picks the default.
    LAUNCH = os.environ.get("WOQ_ENGINE_LAUNCH", "graph")

    @property
    def launch(self):
        return getattr(self, "_launch", None) or type(self).LAUNCH

    @launch.setter
    def launch(self, mode):
        if mode not in ("eager", "graph"):
            raise ValueError("launch mode is 'eager' or 'graph'")
        self._launch = mode

    def capture(self, greedy=True):
        """Capture one token step into a hipGraph (on the engine's own stream)."""
        cur = torch.cuda.current_stream(self.device)
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            L.check(L.lib().woq_engine_capture(self._h, int(greedy), L.stream_ptr()))
        cur.wait_stream(self._stream)
        self.captured = True
        self._captured_greedy = bool(greedy)

    def prepare_decode(self, greedy=True):
        """Whatever `replay` needs before a burst: a captured graph in "graph" mode, nothing in "eager" mode."""
        if self.launch == "graph" and not self.captured:
            self.capture(greedy=greedy)
        self._burst_greedy = bool(greedy)

    def run(self, n=1, greedy=True):
        """`n` decode steps issued eagerly by one native call (greedy: token / position chain on the device)."""
        L.check(L.lib().woq_engine_steps(self._h, int(n), int(greedy), L.stream_ptr()))

    def replay_graph(self, n=1):
        """Replay the captured hipGraph n times on the CURRENT stream (greedy chaining stays on the device). Only the
        capture needs a stream of its own; replaying on that stream behind `wait_stream` calls was what made graph
        replays ~1 us per kernel boundary slower than eager launches (profiles/r04ab_stream_mode_probe.txt)."""
        L.check(L.lib().woq_engine_replay(self._h, int(n), L.stream_ptr()))

    def replay_graph_round3(self, n=1):
        """MEASUREMENT ONLY: the graph replayed the way rounds 1-3 did it — on the engine's capture stream, bracketed by
        `wait_stream` in both directions. bench.py times it next to the other two so that the cause of round 3's 12 %
        stays on record in every line."""
        self._on_own_stream(lambda: L.check(L.lib().woq_engine_replay(self._h, int(n), L.stream_ptr())))

    def replay(self, n=1):
        """A burst of `n` decode steps of the kind last captured / prepared (greedy chaining stays on the device): the
        captured graph in "graph" mode, `run` in "eager" mode (same kernels, same results — the tests hold graph replays
        and eager steps to bit-identical tokens)."""
        if self.launch == "graph":
            return self.replay_graph(n)
        self.run(n, greedy=getattr(self, "_burst_greedy", getattr(self, "_captured_greedy", True)))

    def token_log(self):
        """int32 view [max_ctx + 1] of the engine's token log: slot p = the greedy token of the step that fed position
        p. After `replay(k)` from position p0 the k new tokens are token_log()[p0 : p0 + k] — one host read."""
        return _device_view(L.lib().woq_engine_token_log_ptr(self._h), (self.cfg.max_ctx + 1,), self.device, "<i4")

    def uses_xq(self):
        """True when the decode step's kernels hand activations over as XQ limb blocks (csrc/woq_xq.h)."""
        return bool(L.lib().woq_engine_uses_xq(self._h))

    def phase(self, layer, phase, greedy=True):
        L.check(L.lib().woq_engine_phase(self._h, int(layer), int(phase), int(greedy), L.stream_ptr()))

    def _on_own_stream(self, fn):
        """Run `fn` with the engine's own stream current (captures are not allowed on the legacy default stream)."""
        cur = torch.cuda.current_stream(self.device)
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            out = fn()
        cur.wait_stream(self._stream)
        return out

    def set_time_eager(self, on=True):
        """time_gemv / time_twin: passes issued eagerly back to back on the current stream (default) or as a replayed
        graph on the engine's own stream (round 3's measure; that stream sits behind a `wait_stream`, which costs every
        kernel boundary ~1 us — see LAUNCH)."""
        L.check(L.lib().woq_engine_set_time_eager(self._h, int(bool(on))))
        self._time_eager = bool(on)

    def _timed_call(self, fn):
        """Timing helpers: eagerly issued passes need no stream of their own and must not sit behind a cross-stream wait;
        captured ones cannot run on the legacy default stream."""
        return fn() if getattr(self, "_time_eager", True) else self._on_own_stream(fn)

    def time_gemv(self, reps=1, mask=15):
        """(total ms, algorithmic bytes per pass, launches per pass) of `reps` replays of a captured pass over every
        layer's GEMV launches in the decode step's own forms; mask picks projections (bit 0 qkv, 1 o, 2 gate/up, 3 down)."""
        ms, by, n = ctypes.c_float(), ctypes.c_double(), ctypes.c_int()
        self._timed_call(lambda: L.check(L.lib().woq_engine_time_gemv_mask(
            self._h, int(mask), reps, L.stream_ptr(), ctypes.byref(ms), ctypes.byref(by), ctypes.byref(n))))
        return ms.value, by.value, n.value

    def time_twin(self, mode, reps=1):
        """Total ms of `reps` replays of a captured pass over every layer's 4 GEMV launches with the arithmetic taken
        out: mode 0 = load-only twins over the engine's own blobs, mode 1 = empty kernels on the same grids (bench.py
        roofline.ceiling)."""
        ms = ctypes.c_float()
        self._timed_call(lambda: L.check(L.lib().woq_engine_time_twin(
            self._h, int(mode), int(reps), L.stream_ptr(), ctypes.byref(ms))))
        return ms.value

    def time_prefill_gemm(self, layer, n_rows, reps=3):
        """(gemm_ms, call_ms) of the engine's own gate/up GEMM call over the `n_rows` rows a preceding prefill left in
        the residual stream: the MFMA GEMM kernel alone, and pack pass + GEMM (bench.py prefill.dominant_gemm)."""
        g, c = ctypes.c_float(), ctypes.c_float()
        L.check(L.lib().woq_engine_time_prefill_gemm(self._h, int(layer), int(n_rows), int(reps), L.stream_ptr(),
                                                     ctypes.byref(g), ctypes.byref(c)))
        return g.value, c.value

    # ---- tensor parallel: host-driven collectives between sub-blocks (RCCL via torch.distributed) ----
    def step_tp(self, group=None, greedy=True):
        """One token with the row-parallel partial sums all-reduced after o_proj and down_proj.
        One RCCL all-reduce (sum, fp32 [hidden]) per sub-block — SURVEY.md §8(e)."""
        import torch.distributed as dist

        for l in range(self.cfg.layers):
            self.phase(l, 0, greedy)
            dist.all_reduce(self.hidden, group=group)
            self.phase(l, 1, greedy)
            dist.all_reduce(self.hidden, group=group)
        self.phase(0, 2, greedy)

    def prefill(self, tokens, start_pos=0, greedy=True):
        """Prompt pass: `tokens` int [T] or [n_seq, T] at positions start_pos.. of each sequence's KV cache (all
        linears as MFMA GEMMs over n_seq*T rows, causal attention over the cache). Returns the last-position logits
        fp32 [n_seq, vocab] (a view of engine memory, valid until the next prefill). Sequence 0 is the one
        step()/replay() continue: with greedy its next token and position are already in place."""
        t = torch.as_tensor(tokens, dtype=torch.int32, device=self.device)
        if t.dim() == 1:
            t = t.unsqueeze(0)
        t = t.contiguous()
        n_seq, T = t.shape
        L.check(L.lib().woq_engine_prefill(self._h, t.data_ptr(), int(n_seq), int(T), int(start_pos), int(greedy),
                                           L.stream_ptr()))
        ptr = L.lib().woq_engine_prefill_logits_ptr(self._h)
        return _device_view(ptr, (n_seq, self.cfg.vocab), self.device)

    # ---- decode attention regime -----------------------------------------------------------------------------
    LONG_CTX = 480  # cached positions beyond which the sliced decode attention wins (r04l, V runs double-buffered: 448 -> 0.617 vs 0.631, 512 -> 0.635 vs 0.634, 640 -> 0.661 vs 0.634). Round 4 (eager bursts, V rows of two runs prefetched; Llama-2-7B shape, 16 layers, ms per token, one workgroup per head inside the fused launch vs context slices + combine: 256 -> 0.593 vs 0.618, 384 -> 0.615 vs 0.621, 512 -> 0.644 vs 0.635; profiles/r04k_*). Round 1 had measured the crossover near 256 on the two-launch form

    FUSED_SLICED_CTX = 192  # ... and where slices INSIDE the fused launch (self-merging, round 6) start to pay: 7B shape,
    # 16 layers, ms per token, 1 vs 4 slices (profiles/r06t_*): 96 -> 0.544-0.549 vs 0.558, 160 -> 0.557-0.562 vs
    # 0.561, 224 -> 0.570-0.574 vs 0.564, 320 -> 0.588 vs 0.570, 512 -> 0.625 vs 0.580

    def set_attn_splits(self, n):
        """1 = one workgroup per head, n > 1 = n context slices per head + combine. Invalidates a captured graph."""
        if int(n) != L.lib().woq_engine_attn_splits(self._h):
            L.check(L.lib().woq_engine_set_attn_splits(self._h, int(n)))
            self.captured = False

    GROUPED_CTX = 10240  # cached positions from which the grouped-query (matrix-core) slices are used where they apply
    # (a launch of their own: 186-247 registers, they cannot share a kernel with the strips). Round 6, Mistral-7B shape,
    # fp8 cache, 16 layers, ms per token, 16 per-head slices inside the fused launch vs grouped slices (both forms
    # merge among themselves; profiles/r06t_*): 4096 -> 0.644 vs 0.707, 8192 -> 0.720 vs 0.745, 16384 -> 0.850 vs 0.789

    def set_attn_grouped(self, on):
        """Sliced regime only: one workgroup per (kv head, slice) for all the query heads of the group (head_dim 128,
        2 / 4 / 8 query heads per kv head, fp16 / fp8 cache; ignored elsewhere). Invalidates a captured graph."""
        if bool(on) != bool(L.lib().woq_engine_attn_grouped(self._h)):
            L.check(L.lib().woq_engine_set_attn_grouped(self._h, int(bool(on))))
            self.captured = False

    GROUPED_CHUNK = 256  # positions per slice of the grouped form's position-independent geometry (two 32-position sub-tiles per wave, both requested before the position is read)

    def set_attn_chunk(self, chunk):
        """Grouped form: `chunk` > 0 = slice s owns the absolute positions [s * chunk, (s + 1) * chunk) whatever the
        current position is (K / V requested before it is read); 0 = slices cut evenly from the current span.
        Invalidates a captured graph."""
        if int(chunk) != L.lib().woq_engine_attn_chunk(self._h):
            L.check(L.lib().woq_engine_set_attn_chunk(self._h, int(chunk)))
            self.captured = False

    def set_tp_options(self, xq=True, fused_push=True):
        """Tensor-parallel decode with a device communicator: `xq` = the XQ decode kernels with an XQ-emitting
        all-reduce kernel (off: the fp32-activation kernels), `fused_push` = o_proj / down_proj push their partial sums
        into the peers' inboxes from their own epilogue (off: the all-reduce kernel pushes; bit-identical results).
        Invalidates a captured graph."""
        L.check(L.lib().woq_engine_set_tp_options(self._h, int(bool(xq)), int(bool(fused_push))))
        self.captured = False

    def set_fuse_attn(self, on):
        """Decode step: qkv GEMV + attention as one launch where the shape allows (csrc/woq_gemv_attn.hip; default
        on). Invalidates a captured graph."""
        L.check(L.lib().woq_engine_set_fuse_attn(self._h, int(bool(on))))
        self.captured = False

    def uses_fused_attn(self):
        """True when the next step / capture runs the fused qkv + attention launch."""
        return bool(L.lib().woq_engine_fuse_attn(self._h))

    def status(self):
        """Sticky device-side status of the decode step, 0 = fine. bit 0: an attention workgroup of the fused launch
        gave up waiting for its head's q / k / v; bit 1: a step found its position at or beyond max_ctx and ran at
        max_ctx - 1 instead (its outputs are meaningless, memory was not touched out of bounds)."""
        return int(L.lib().woq_engine_status(self._h, L.stream_ptr()))

    def clear_status(self):
        """Reset the sticky status (after the caller dealt with what it reported)."""
        L.check(L.lib().woq_engine_clear_status(self._h, L.stream_ptr()))

    def _grouped_applies(self):
        c = self.cfg
        return (c.head_dim == 128 and c.kv_heads > 0 and c.heads // c.kv_heads in (2, 4, 8)
                and c.kv_dtype in (L.F16, L.FP8_E4M3))

    def tune_attn_for(self, positions):
        """Pick the regime for a context of `positions` cached tokens (host-side hint: the position lives on the
        device): slices = enough workgroups to fill the chip, each at least 64 positions; from GROUPED_CTX on,
        grouped-query shapes take 256-position slices of the matrix-core form."""
        grouped = positions >= self.GROUPED_CTX and self._grouped_applies()
        self.set_attn_grouped(grouped)
        # position-independent slices: not with a sliding window (the slices move with the position), and only for
        # the fp8 cache, whose kernel keeps raw bytes in registers and fits two workgroups on a CU — the fp16 form
        # holds one (272 registers), and ceil(positions / 256) x kv_heads workgroups would then run a second round
        # measured slower (+2.5 us per layer, profiles/r04d_*): opt-in through WOQ_ATTN_FIXED_CHUNK=1 (parity:
        # test_grouped_attention_fixed_chunk_slices_vs_oracle drives the same geometry through set_attn_chunk)
        fixed = (os.environ.get("WOQ_ATTN_FIXED_CHUNK", "0") not in ("", "0") and grouped and not self.cfg.reserved[2]
                 and self.cfg.kv_dtype == L.FP8_E4M3)
        self.set_attn_chunk(self.GROUPED_CHUNK if fixed else 0)
        fused_sliced = not grouped and self.uses_fused_attn_sliced()
        if positions <= (self.FUSED_SLICED_CTX if fused_sliced else self.LONG_CTX):
            self.set_attn_splits(1)
        elif fixed:  # enough fixed slices to cover `positions`; the last one takes what lies beyond
            self.set_attn_splits(max(2, min(64, -(-positions // self.GROUPED_CHUNK))))
        elif grouped:  # (round 6, slices merging among themselves: 8k -> 0.765 / 0.732 / 0.739 ms for 16 / 24 / 32 slices)
            self.set_attn_splits(max(2, min(32, positions // 320)))
        elif fused_sliced:
            # slices as attention workgroups of the fused qkv launch, merging among themselves (round 6): fewer, longer
            # slices than launches of their own want (profiles/r06t_*, 16 layers, ms per token) — 7B shape 512: 0.580 /
            # 0.577 for 4 / 8 (0.594 for 2), 2048: 0.657 / 0.659 / 0.668 for 8 / 12 / 16 (0.707 for 4); Mistral shape
            # 2048: 0.633 / 0.628 for 8 / 16, 4096: 0.689 / 0.670, 8192: 0.785 / 0.751 for 12 / 16
            self.set_attn_splits(4 if positions < 1024 else 8 if positions < 3072 else 16)
        else:
            self.set_attn_splits(max(2, min(32, 1024 // max(1, self.cfg.heads), positions // 64)))

    def uses_fused_attn_sliced(self):
        """True when a sliced context would ride in the fused qkv launch (probe: the engine's own predicate at two slices)."""
        keep = L.lib().woq_engine_attn_splits(self._h)
        L.check(L.lib().woq_engine_set_attn_splits(self._h, 2))
        ok = bool(L.lib().woq_engine_fuse_attn(self._h))
        L.check(L.lib().woq_engine_set_attn_splits(self._h, keep))
        return ok

    def kv_cache(self, which="k"):
        """The engine's K or V cache as a torch view [max_batch, layers, max_ctx, kv_heads, head_dim] (no copy)."""
        c = self.cfg
        shape = (self.max_batch, c.layers, c.max_ctx, c.kv_heads, c.head_dim)
        ptr = L.lib().woq_engine_kv_cache_ptr(self._h, 0 if which == "k" else 1)
        code = c.kv_dtype
        if code == L.FP8_E4M3:
            return _device_view(ptr, shape, self.device, "|u1").view(torch.float8_e4m3fn)
        t = _device_view(ptr, shape, self.device, "<f2")
        return t if code == L.F16 else t.view(torch.bfloat16)

    # ---- tensor parallel through the engine's all-reduce seam (woq_engine_set_allreduce) --------------------------
    def bind_allreduce(self, group=None):
        """Let the NATIVE step / prompt pass call back into torch.distributed after o_proj and down_proj: the
        callback wraps the engine's fp32 buffer as a tensor view and issues `dist.all_reduce` (backend "nccl" = RCCL
        over xGMI) on the current stream. Needed for `prefill` under TP (the host-driven `step_tp` splits only the
        decode step); not graph-capturable."""
        import torch.distributed as dist

        def _cb(_user, buf, count, _stream):
            try:
                dist.all_reduce(_device_view(buf, (int(count),), self.device), group=group)
                return 0
            except Exception:  # pragma: no cover - surfaces as "all-reduce callback failed" from the C side
                return 1

        self._allreduce_cb = L.ALLREDUCE_FN(_cb)  # keep the ctypes thunk alive
        L.check(L.lib().woq_engine_set_allreduce(self._h, self._allreduce_cb, None))
        self.captured = False

    def bind_comm(self, comm, vocab_offset=0):
        """Tensor parallel with the exchange ON THE DEVICE (runtime/comm.py DeviceComm): the native step then issues
        the two all-reduce kernels per layer and the greedy-token exchange itself, so `capture()` / `replay()` cover a
        whole tensor-parallel token. `vocab_offset` = first vocabulary row of this rank's lm_head shard. The prompt
        pass keeps a bound RCCL callback when there is one (bandwidth-bound rows), else it uses the comm in pieces."""
        h = comm.handle if comm is not None else None
        L.check(L.lib().woq_engine_set_comm(self._h, h, int(vocab_offset)))
        self._comm = comm
        self.captured = False

    def unbind_allreduce(self):
        L.check(L.lib().woq_engine_set_allreduce(self._h, ctypes.cast(None, L.ALLREDUCE_FN), None))
        self._allreduce_cb = None

    def generate(self, prompt_ids, max_new_tokens, chunk=2048, burst=16):
        """Greedy decode: the prompt goes through the prefill pass in chunks of `chunk` tokens, then bursts of `burst`
        steps chained on the device (one host read of the token log per burst)."""
        out = []
        ids = [int(t) for t in prompt_ids]
        if len(ids) + max_new_tokens > self.cfg.max_ctx:  # the kernels index the KV cache by position, unchecked
            raise RuntimeError("QBits: prompt (%d) + max_new_tokens (%d) exceeds the engine's max_ctx (%d)"
                               % (len(ids), max_new_tokens, self.cfg.max_ctx))
        for s0 in range(0, len(ids), chunk):
            self.prefill(ids[s0:s0 + chunk], start_pos=s0, greedy=True)
        self.tune_attn_for(len(ids) + max_new_tokens)
        if max_new_tokens < 1:
            return out
        out.append(int(self.token.item()))  # the prompt pass's token
        self.prepare_decode(greedy=True)
        log = self.token_log()
        while len(out) < max_new_tokens:
            k = min(int(burst), max_new_tokens - len(out))
            p0 = len(ids) + len(out) - 1  # position the next step feeds
            self.replay(k)
            out += log[p0:p0 + k].tolist()
        return out


class DeviceSampler:
    """Next-token choice on the device from the engine's fp32 logits, with Hugging Face's semantics for the options the
    reference's NeuralChat passes by default (`neural_chat/config.py:400-409`: do_sample=True, temperature 0.1, top_k 40,
    top_p 0.75, repetition_penalty 1.1): RepetitionPenaltyLogitsProcessor over prompt + generated ids, then — when
    sampling — TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper (min_tokens_to_keep 1) and
    `torch.multinomial` on the softmax; argmax otherwise. Plain torch ops in HF's order (checked against the HF classes in
    tests/test_api_cpu.py); no host synchronisation — the caller reads tokens back in bursts."""

    def __init__(self, do_sample=False, temperature=1.0, top_k=0, top_p=1.0, repetition_penalty=1.0, generator=None):
        self.do_sample = bool(do_sample)
        self.temperature = float(temperature if temperature is not None else 1.0)
        self.top_k = int(top_k or 0)
        self.top_p = float(top_p if top_p is not None else 1.0)
        self.penalty = float(repetition_penalty if repetition_penalty is not None else 1.0)
        self.generator = generator
        if self.do_sample and self.temperature <= 0:
            raise ValueError("`temperature` has to be a strictly positive float when sampling")

    def processed(self, logits, history):
        """logits fp32 [vocab]; history int64 [n] (prompt + generated so far) -> the scores sampling draws from."""
        scores = logits.clone()
        if self.penalty != 1.0 and history.numel():
            picked = scores[history]
            scores[history] = torch.where(picked < 0, picked * self.penalty, picked / self.penalty)
        if not self.do_sample:
            return scores
        if self.temperature != 1.0:
            scores = scores / self.temperature
        if self.top_k > 0:
            k = min(self.top_k, scores.numel())
            kth = torch.topk(scores, k).values[-1]
            scores = scores.masked_fill(scores < kth, float("-inf"))
        if self.top_p < 1.0:
            sorted_scores, order = torch.sort(scores, descending=False)
            cum = sorted_scores.softmax(-1).cumsum(-1)
            drop = cum <= (1.0 - self.top_p)
            drop[-1] = False  # min_tokens_to_keep = 1
            scores = scores.masked_fill(torch.zeros_like(drop).scatter(0, order, drop), float("-inf"))
        return scores

    TIE_SLACK = 8  # candidates kept beyond top_k so that values tied with the k-th one survive like in HF's mask

    def __call__(self, logits, history):
        if self.do_sample and 0 < self.top_k < logits.numel():
            return self._draw_top_k(logits, history)
        scores = self.processed(logits, history)
        if not self.do_sample:
            return scores.argmax().reshape(1)
        return torch.multinomial(scores.softmax(-1), 1, generator=self.generator)

    def _draw_top_k(self, logits, history):
        """The same distribution as `processed` + softmax + multinomial when top_k is set, without sorting the whole
        vocabulary: after the top-k mask only the candidates matter (everything else has probability 0), so the
        temperature, the nucleus cut and the draw run on top_k (+ TIE_SLACK) values — a 32000-entry sort per token is
        most of what sampling costs otherwise (profiles/r04n_*). Ties beyond TIE_SLACK entries at the k-th value would
        be cut where HF keeps them all; fp32 logits do not tie in practice."""
        scores = logits
        if self.penalty != 1.0 and history.numel():
            scores = logits.clone()
            picked = scores[history]
            scores[history] = torch.where(picked < 0, picked * self.penalty, picked / self.penalty)
        k2 = min(self.top_k + self.TIE_SLACK, scores.numel())
        vals, idx = torch.topk(scores, k2)  # descending
        if self.temperature != 1.0:
            vals = vals / self.temperature
        vals = vals.masked_fill(vals < vals[self.top_k - 1], float("-inf"))
        if self.top_p < 1.0:  # HF: ascending order, drop while the cumulative probability <= 1 - top_p, keep the last
            asc = vals.flip(0)
            drop = asc.softmax(-1).cumsum(-1) <= (1.0 - self.top_p)
            drop[-1] = False
            vals = vals.masked_fill(drop.flip(0), float("-inf"))
        pick = torch.multinomial(vals.softmax(-1), 1, generator=self.generator)
        return idx[pick]


def iter_sampled(engine, prompt_ids, max_new_tokens, sampler, eos=(), burst=16, chunk=2048):
    """Prompt pass + decode steps with the next token chosen by `sampler` on the device (sampling and / or repetition
    penalty: the requests `generate`'s greedy chain does not cover). Every step is the engine's native step (logits
    only) followed by a dozen small torch kernels; nothing synchronises until `burst` tokens are read back (1 for a
    text stream). Yields each burst's new tokens; stops after the first id in `eos` (kept in the output, like HF)."""
    ids = [int(t) for t in prompt_ids]
    n = len(ids)
    if n + max_new_tokens > engine.cfg.max_ctx:
        raise RuntimeError("QBits: prompt (%d) + max_new_tokens (%d) exceeds the engine's max_ctx (%d)"
                           % (n, max_new_tokens, engine.cfg.max_ctx))
    dev = engine.device
    hist = torch.empty(n + max_new_tokens, dtype=torch.int64, device=dev)
    hist[:n] = torch.tensor(ids, dtype=torch.int64, device=dev)
    for s0 in range(0, n, chunk):
        engine.prefill(ids[s0:s0 + chunk], start_pos=s0, greedy=False)
    engine.tune_attn_for(n + max_new_tokens)
    eos = set(int(e) for e in eos)
    done, made = False, 0
    while not done and made < max_new_tokens:
        k = min(burst, max_new_tokens - made)
        for _ in range(k):
            if made > 0:
                engine.step(greedy=False)  # logits of the token just chosen, at position n + made - 1
            tok = sampler(engine.logits, hist[:n + made])
            hist[n + made] = tok[0]
            engine.token.copy_(tok.to(torch.int32))
            engine.pos.fill_(n + made)
            made += 1
        new = hist[n + made - k:n + made].tolist()  # the burst's one host synchronisation
        for j, t in enumerate(new):
            if t in eos:
                new, done = new[:j + 1], True
                break
        yield new


def generate_sampled(engine, prompt_ids, max_new_tokens, sampler, eos=(), on_tokens=None, burst=16, chunk=2048):
    """`iter_sampled` to the end; `on_tokens(list)` receives each burst. Returns the new tokens."""
    out = []
    for new in iter_sampled(engine, prompt_ids, max_new_tokens, sampler, eos=eos, burst=burst, chunk=chunk):
        out += new
        if on_tokens is not None:
            on_tokens(new)
    return out


def _device_view(ptr, shape, device, typestr="<f4"):
    """torch view of engine-owned device memory (no copy) through the CUDA array interface."""

    class _Raw:
        pass

    raw = _Raw()
    raw.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(raw, device=device)


def synth_llama_weights(engine, hidden, inter, heads, kv_heads, head_dim, layers, vocab, group=128, sym=True,
                        scale_dtype="fp16", seed=1234, model_dtype=torch.float16, embed_vocab=None,
                        shared_seed=None, weight_dtype="int4_clip", compute_dtype="fp32", act_order=False):
    """Synthetic random-init quantised Llama-shaped weights built directly on the device (no checkpoint, no
    network): int4 values uniform in [-8,7], scales ~ 0.02-ish/7 so dequantised weights look like N(0, 0.02^2),
    norms 1 + N(0, 0.02^2), embeddings N(0, 0.02^2). Returns the list of blobs (kept alive by the engine).
    Tensor-parallel shards: pass the PER-RANK heads / kv_heads / inter / vocab (what the engine was built with) and
    `embed_vocab` = the full vocabulary (the embedding table is replicated, lm_head is the rank's rows), and
    `shared_seed` = one seed for all ranks: the replicated tensors (norm weights, embedding) are drawn from it so
    that every rank holds the same copy, while `seed` (per rank) draws the shards.
    `weight_dtype` "nf4" / "fp4_e2m1" / "fp4_e2m1_bnb": uniform table codes 0..15 instead of int4 values (symmetric),
    scales sized so that the dequantised weights keep the same spread; `compute_dtype` is recorded in the blobs (nf4
    decodes with three digit planes at "fp32", two at "bf16" / "fp16" / "int8": csrc/woq_gemv_common.h)."""
    table = weight_dtype in TABLE_WEIGHT_DTYPES
    fp8 = weight_dtype in FP8_WEIGHT_DTYPES
    if (table or fp8) and not sym:
        raise RuntimeError("QBits: float weight types are symmetric")
    dev = engine.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    gs = g  # replicated tensors
    if shared_seed is not None:
        gs = torch.Generator(device=dev)
        gs.manual_seed(shared_seed)

    def rand_q(k, n):
        kg = (k + (k if group == -1 else group) - 1) // (k if group == -1 else group)
        if fp8:  # fp8 codes of N(0, 64^2) values (finite by construction), scales so that the weights look like N(0, 0.02^2)
            v = (64.0 * torch.randn(k, n, generator=g, device=dev)).clamp_(-400.0, 400.0)
            q = v.to(FP8_WEIGHT_DTYPES[weight_dtype]).view(torch.int8)
            s = (0.5 + torch.rand(kg, n, generator=g, device=dev)) * (0.02 / 64.0)
            return q, s, None
        q = torch.randint(0 if table else -8, 16 if table else 8, (k, n), generator=g, device=dev, dtype=torch.int8)
        s = (0.5 + torch.rand(kg, n, generator=g, device=dev)) * (0.02 / (TABLE_WEIGHT_DTYPES[weight_dtype] if table else 4.0))
        z = None if sym else torch.randint(-8, 8, (kg, n), generator=g, device=dev, dtype=torch.int8)
        return q, s, z

    def pack(q, s, z):
        # act_order: a GPTQ desc_act blob — every projection carries a random group assignment of its K rows (q / k / v and
        # gate / up are fused, so each fused blob has one, the way a GPTQ run leaves them); the rows are synthetic, so
        # "regrouped order" is whatever they are in
        k = q.shape[0]
        gsz = k if group == -1 else group
        g_idx = torch.empty(0, dtype=torch.int32)
        if act_order:
            g_idx = (torch.randperm(k, generator=g, device=dev) // gsz).to(torch.int32)
        return qbits.repack_quantized_weight(q, s, z if z is not None else torch.empty(0, dtype=torch.int8),
                                             g_idx, weight_dtype, scale_dtype, compute_dtype, z is not None, group)

    qkv_n = (heads + 2 * kv_heads) * head_dim
    for l in range(layers):
        q, s, z = rand_q(hidden, qkv_n)
        qkv_blob = pack(q, s, z)
        q, s, z = rand_q(heads * head_dim, hidden)
        o_blob = pack(q, s, z)
        gq, gsc, gz = rand_q(hidden, inter)
        uq, usc, uz = rand_q(hidden, inter)
        gu_blob = pack(fuse_gate_up(gq, uq), fuse_gate_up(gsc, usc), None if gz is None else fuse_gate_up(gz, uz))
        q, s, z = rand_q(inter, hidden)
        down_blob = pack(q, s, z)
        ln1 = 1 + 0.02 * torch.randn(hidden, generator=gs, device=dev)
        ln2 = 1 + 0.02 * torch.randn(hidden, generator=gs, device=dev)
        engine.set_layer(l, qkv_blob, o_blob, gu_blob, down_blob, ln1, ln2)
        del q, s, z, gq, gsc, gz, uq, usc, uz
    embed = (0.02 * torch.randn(embed_vocab or vocab, hidden, generator=gs, device=dev)).to(model_dtype)
    lm_head = (0.02 * torch.randn(vocab, hidden, generator=g, device=dev)).to(model_dtype)
    norm = 1 + 0.02 * torch.randn(hidden, generator=gs, device=dev)
    engine.set_head(embed, norm, lm_head)
    return engine


# ---- fused decode engine from a quantised HF model (the `ipex.optimize_transformers` analogue) ---------------------
# 4-bit table weight types the engine takes (round 4) -> the rms of their table over uniform codes (synthetic weights)
TABLE_WEIGHT_DTYPES = {"nf4": 0.5, "fp4_e2m1": 2.9, "fp4_e2m1_bnb": 0.47}
# 8-bit float weight types the engine takes (round 6; reference strings bestla_weightonly_dispatcher.hpp:62-72) -> torch dtype
FP8_WEIGHT_DTYPES = {"fp8_e4m3": torch.float8_e4m3fn, "fp8_e5m2": torch.float8_e5m2}
_TABLES = {
    "nf4": [-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635,
            -0.18477343022823334, -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725,
            0.24611230194568634, 0.33791524171829224, 0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0],
    "fp4_e2m1": [0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0, -0.0, -0.5, -1.0, -1.5, -2.0, -3.0, -4.0, -6.0],
    "fp4_e2m1_bnb": [0.0, 0.0625 / 12, 8.0 / 12, 1.0, 4.0 / 12, 0.5, 2.0 / 12, 0.25,
                     -0.0, -0.0625 / 12, -8.0 / 12, -1.0, -4.0 / 12, -0.5, -2.0 / 12, -0.25],
}


def _code_parts(mod):
    """QuantizedLinearQBits of a 4-bit table type -> (codes int8 [K,N] in 0..15, scales fp32 [G,N], None): the blob has
    no integer export (modules.py recover_qparms raises), so the codes are read back as the nearest table entry of
    dequantised / scale — exact, the entries are >= 5e-3 apart and the quotient is off by an fp32 rounding."""
    w = mod.weight.data
    info = lambda t: qbits.acquire_packed_weight_info(w, t)  # noqa: E731
    k, n = int(info(2)[0]), int(info(3)[0])
    if int(info(4)[0]):
        raise RuntimeError("QBits: the fused decode engine does not take act-order (g_idx) layers")
    scales = info(9)
    deq = torch.empty(k, n, dtype=torch.float32, device=w.device)
    qbits.dequantize_packed_weight(w, deq, False, mod.compute_dtype, mod.weight_dtype, mod.scale_dtype)
    s = scales[torch.arange(k, device=w.device) // int(info(1)[0])]
    x = deq / torch.where(s == 0, torch.ones_like(s), s)
    tab = torch.tensor(_TABLES[mod.weight_dtype], dtype=torch.float32, device=w.device)
    vals, order = torch.sort(tab, stable=True)
    pos = torch.bucketize(x, (vals[1:] + vals[:-1]) * 0.5)
    return order[pos].to(torch.int8), scales, None


def _fp8_parts(mod):
    """QuantizedLinearQBits of an fp8 weight type -> (code bytes int8 [K,N], scales fp32 [G,N], None): dequantised / scale
    IS an fp8 value up to an fp32 rounding of the quotient, so casting it to the torch fp8 dtype returns the code."""
    w = mod.weight.data
    info = lambda t: qbits.acquire_packed_weight_info(w, t)  # noqa: E731
    k, n = int(info(2)[0]), int(info(3)[0])
    if int(info(4)[0]):
        raise RuntimeError("QBits: the fused decode engine does not take act-order (g_idx) layers")
    scales = info(9)
    deq = torch.empty(k, n, dtype=torch.float32, device=w.device)
    qbits.dequantize_packed_weight(w, deq, False, mod.compute_dtype, mod.weight_dtype, mod.scale_dtype)
    s = scales[torch.arange(k, device=w.device) // int(info(1)[0])]
    x = deq / torch.where(s == 0, torch.ones_like(s), s)
    return x.to(FP8_WEIGHT_DTYPES[mod.weight_dtype]).view(torch.int8), scales, None


def _signed_parts(mod):
    """QuantizedLinearQBits -> (q int8 [K,N] in [-8,7], scales fp32 [G,N], zp int8 [G,N] signed or None, g_idx int32 [K]
    or None). Act-order (GPTQ desc_act) modules: rows in the regrouped order the blob holds, plus the raw g_idx that
    `repack_quantized_weight` turns into the activation shuffle (reference nn/modules.py:205-224)."""
    if getattr(mod, "weight_dtype", "int4_clip") in TABLE_WEIGHT_DTYPES:
        return _code_parts(mod) + (None,)
    if getattr(mod, "weight_dtype", "int4_clip") in FP8_WEIGHT_DTYPES:
        return _fp8_parts(mod) + (None,)
    int_w, scales, zeros, g_idx = mod.recover_qparms_kn()
    if g_idx is not None:
        # recover_qparms_kn hands the rows back in the CHECKPOINT's order; the blob wants them regrouped (group g owns
        # rows [g * group, (g + 1) * group), in order of appearance) exactly as set_weights_bias does before it repacks
        order = torch.argsort(g_idx.to(torch.int64), stable=True)
        int_w = int_w[order.to(int_w.device)].contiguous()
    return (int_w - 8).to(torch.int8), scales, None if zeros is None else (zeros - 8).to(torch.int8), g_idx


def _same_order(parts, what):
    """Projections fused along N share ONE activation shuffle: q / k / v (and gate / up) of a GPTQ act-order checkpoint
    see the same inputs, hence the same Hessian diagonal, hence the same permutation — checked, not assumed."""
    idx = [p[3] for p in parts]
    if all(i is None for i in idx):
        return None
    if any(i is None for i in idx) or any(not torch.equal(idx[0], i) for i in idx[1:]):
        raise RuntimeError("QBits: %s carry different act-order permutations; the fused decode engine needs one per fused "
                           "projection (the model keeps the module path)" % what)
    return idx[0]


def optimize_transformers(model, max_ctx=2048, kv_dtype=torch.float16):
    """Post-pass over a model returned by `AutoModelForCausalLM.from_pretrained(..., quantization_config=...)`:
    builds the native batch-1 decode engine over the SAME quantised weights (q/k/v fused along N, gate/up
    interleaved per 16-column tile for the fused SiLU*mul epilogue). Mirrors the two-step shape of the reference's
    Intel-GPU flow, `ipex.optimize_transformers(qmodel, ...)` after `from_pretrained`
    (docs/weightonlyquant.md:199-202). Llama-class decoders (Llama-2, Mistral: RMSNorm, RoPE, gated SiLU MLP)."""
    cfg = model.config
    if getattr(cfg, "model_type", "") not in ("llama", "mistral"):
        raise RuntimeError("QBits: the fused decode engine covers Llama-class decoders, got %r" % cfg.model_type)
    # anything the engine's kernels do not implement keeps the model on the module path instead of changing its maths
    rs = getattr(cfg, "rope_scaling", None) or (getattr(cfg, "rope_parameters", None) or {})
    rtype = (rs.get("rope_type") or rs.get("type") or "default") if isinstance(rs, dict) else "default"
    if rtype not in ("default", None):
        raise RuntimeError("QBits: the fused decode engine implements plain RoPE only (rope scaling %r)" % rtype)
    if getattr(cfg, "hidden_act", "silu") not in ("silu", "swish"):
        raise RuntimeError("QBits: the fused decode engine implements the SiLU-gated MLP only (%r)" % cfg.hidden_act)
    layers = model.model.layers
    first = layers[0].self_attn.q_proj
    wdt = getattr(first, "weight_dtype", "int4_clip")
    fp8 = wdt in FP8_WEIGHT_DTYPES
    if not fp8 and (getattr(first, "bits", 4) != 4 or (wdt != "int4_clip" and wdt not in TABLE_WEIGHT_DTYPES)):
        raise RuntimeError("QBits: the fused decode engine takes int4_clip / nf4 / fp4 / fp8 layers (int8 models run "
                           "on the module path)")
    hidden, inter = cfg.hidden_size, cfg.intermediate_size
    heads = cfg.num_attention_heads
    kv_heads = getattr(cfg, "num_key_value_heads", heads) or heads
    head_dim = getattr(cfg, "head_dim", None) or hidden // heads
    theta = getattr(cfg, "rope_theta", None)
    if theta is None:
        theta = (getattr(cfg, "rope_parameters", None) or {}).get("rope_theta", 10000.0)
    dev = first.weight.device
    window = getattr(cfg, "sliding_window", None) or 0  # Mistral v0.1: 4096; null / absent = full causal attention
    eng = WoqDecoderEngine(hidden, inter, heads, kv_heads, head_dim, len(layers), cfg.vocab_size, max_ctx=max_ctx,
                           rms_eps=cfg.rms_norm_eps, rope_theta=float(theta), kv_dtype=kv_dtype, device=dev,
                           sliding_window=int(window))
    # every projection of every layer is repacked under ONE (weight type, compute type, group, scale type, scheme): a
    # model whose modules differ would be repacked under layer 0's q_proj's and decode silently wrong —
    # refuse it here; such models keep the module path
    def _sig(m):
        return (getattr(m, "weight_dtype", "int4_clip"), getattr(m, "bits", 4), m.blocksize, m.scale_dtype, m.scheme,
                getattr(m, "compute_dtype", "fp32"))

    for l, layer in enumerate(layers):
        at, mlp = layer.self_attn, layer.mlp
        for name, m in (("q_proj", at.q_proj), ("k_proj", at.k_proj), ("v_proj", at.v_proj), ("o_proj", at.o_proj),
                        ("gate_proj", mlp.gate_proj), ("up_proj", mlp.up_proj), ("down_proj", mlp.down_proj)):
            if not hasattr(m, "blocksize") or _sig(m) != _sig(first):
                raise RuntimeError("QBits: the fused decode engine needs one quantisation recipe for every projection; "
                                   "layers.%d.%s is %r, layers.0.q_proj is %r" %
                                   (l, name, _sig(m) if hasattr(m, "blocksize") else type(m).__name__, _sig(first)))
    if (wdt in TABLE_WEIGHT_DTYPES or fp8) and first.scheme == "asym":
        raise RuntimeError("QBits: the float weight types are symmetric only (reference utils/config.py:360-370)")
    asym = first.scheme == "asym" and wdt not in TABLE_WEIGHT_DTYPES and not fp8
    group, sdt = first.blocksize, first.scale_dtype
    # the blobs' compute type picks nf4's digit-plane count (three at fp32, two otherwise); int4 kernels do not read it
    cdt = getattr(first, "compute_dtype", "fp32") if wdt in TABLE_WEIGHT_DTYPES else "fp32"

    def pack(q, s, z, g_idx=None):
        return qbits.repack_quantized_weight(q.contiguous(), s.contiguous(),
                                             z.contiguous() if z is not None else torch.empty(0, dtype=torch.int8),
                                             g_idx.to(torch.int32).contiguous() if g_idx is not None
                                             else torch.empty(0, dtype=torch.int32), wdt, sdt, cdt, asym, group)

    def cat(parts, what):
        return (torch.cat([p[0] for p in parts], 1), torch.cat([p[1] for p in parts], 1),
                torch.cat([p[2] for p in parts], 1) if asym else None, _same_order(parts, what))

    for l, layer in enumerate(layers):
        at, mlp = layer.self_attn, layer.mlp
        for m in (at.q_proj, at.k_proj, at.v_proj, at.o_proj, mlp.gate_proj, mlp.up_proj, mlp.down_proj):
            if m.bias is not None:
                raise RuntimeError("QBits: the fused decode engine does not take biased projections")
        qkv = pack(*cat([_signed_parts(at.q_proj), _signed_parts(at.k_proj), _signed_parts(at.v_proj)],
                        "layers.%d q_proj / k_proj / v_proj" % l))
        g, u = _signed_parts(mlp.gate_proj), _signed_parts(mlp.up_proj)
        gu = pack(fuse_gate_up(g[0], u[0]), fuse_gate_up(g[1], u[1]), fuse_gate_up(g[2], u[2]) if asym else None,
                  _same_order([g, u], "layers.%d gate_proj / up_proj" % l))
        eng.set_layer(l, qkv, at.o_proj.weight.data, gu, mlp.down_proj.weight.data, layer.input_layernorm.weight,
                      layer.post_attention_layernorm.weight)
    head_dtype = kv_dtype if kv_dtype in (torch.float16, torch.bfloat16) else torch.float16  # fp8 is cache-only
    eng.set_head(model.model.embed_tokens.weight.detach().to(head_dtype), model.model.norm.weight,
                 model.lm_head.weight.detach().to(head_dtype))
    model.woq_engine = eng
    return eng
