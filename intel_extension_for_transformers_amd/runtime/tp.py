"""Tensor-parallel sharding of the int4 decoder for one node of MI355X GPUs (one process per GPU, RCCL over xGMI).

The reference never combines weight-only quantisation with tensor parallelism (its multi-device inference is
DeepSpeed AutoTP over HCCL / oneCCL on fp models, neural_chat/models/model_utils.py:238-311),
so this is new, but the plan is the textbook one and the only exchange step is the one the maths requires:
  * q / k / v and gate / up are COLUMN-parallel: split N (whole heads; quantisation groups run along K, untouched);
  * o_proj and down_proj are ROW-parallel: split K at multiples of the group size, so every rank owns whole groups;
    each rank produces a partial [hidden] sum and ONE all-reduce (sum) follows each of the two — 2 per layer, 16 KB
    at batch 1 for the 70B shape (latency-bound, far below what the 7 x ~153 GB/s xGMI links carry);
  * lm_head is vocab-sharded; greedy decoding exchanges one (max, index) pair per rank, sampling gathers the row.
The host logic here is pure tensor slicing (numpy or torch) and is exercised on CPU with gloo at world size 2
(tests/test_tp_gloo.py); the device path (`TPDecoder`) feeds the shards to the same WoqDecoderEngine kernels and, in
its production form, leaves the exchange itself to kernels (runtime/comm.py, csrc/woq_comm.hip).
"""
import numpy as np


def _slice(a, sl, axis):
    if a is None:
        return None
    idx = [slice(None)] * a.ndim
    idx[axis] = sl
    return a[tuple(idx)]


def shard_columns(q, scales, zp, rank, world):
    """Column-parallel: q [K,N], scales/zp [G,N] -> the rank's N slice. N must divide evenly."""
    n = q.shape[1]
    if n % world:
        raise ValueError("QBits: N=%d is not divisible by the tensor-parallel degree %d" % (n, world))
    sl = slice(rank * n // world, (rank + 1) * n // world)
    return _slice(q, sl, 1), _slice(scales, sl, 1), _slice(zp, sl, 1)


def shard_rows(q, scales, zp, rank, world, group):
    """Row-parallel: the rank's K slice; the cut must fall on a group boundary so that groups stay whole
    (70B: 8192/8 = 8*128, 28672/8 = 28*128; 7B g32/g128 all divide)."""
    k = q.shape[0]
    g = k if group in (-1, 0) or group >= k else group
    if k % world or (k // world) % g:
        raise ValueError("QBits: K=%d cannot be row-sharded %d ways on group-%d boundaries" % (k, world, g))
    ks = slice(rank * k // world, (rank + 1) * k // world)
    gs = slice(rank * (k // g) // world, (rank + 1) * (k // g) // world)
    return _slice(q, ks, 0), _slice(scales, gs, 0), _slice(zp, gs, 0)


def shard_heads(q, scales, zp, rank, world, heads, head_dim):
    """q/k/v projection [K, heads*head_dim] -> the rank's whole heads."""
    if heads % world:
        raise ValueError("QBits: %d heads are not divisible by the tensor-parallel degree %d" % (heads, world))
    per = heads // world * head_dim
    sl = slice(rank * per, (rank + 1) * per)
    return _slice(q, sl, 1), _slice(scales, sl, 1), _slice(zp, sl, 1)


def shard_llama_layer(parts, rank, world, heads, kv_heads, head_dim, group):
    """parts: dict name -> (q, scales, zp) for q,k,v,o,gate,up,down of ONE decoder layer (full size).
    Returns the same dict for `rank`. kv_heads < world would need KV replication: rejected."""
    if kv_heads % world:
        raise ValueError("QBits: %d KV heads cannot be split %d ways (replication is not implemented)"
                         % (kv_heads, world))
    out = {}
    out["q"] = shard_heads(*parts["q"], rank, world, heads, head_dim)
    out["k"] = shard_heads(*parts["k"], rank, world, kv_heads, head_dim)
    out["v"] = shard_heads(*parts["v"], rank, world, kv_heads, head_dim)
    out["o"] = shard_rows(*parts["o"], rank, world, group)
    out["gate"] = shard_columns(*parts["gate"], rank, world)
    out["up"] = shard_columns(*parts["up"], rank, world)
    out["down"] = shard_rows(*parts["down"], rank, world, group)
    return out


def shard_vocab(lm_head, rank, world):
    """lm_head [vocab, hidden] -> the rank's vocab rows (padded split: the last rank may hold fewer)."""
    v = lm_head.shape[0]
    per = (v + world - 1) // world
    return lm_head[rank * per:min(v, (rank + 1) * per)]


def _host_staged(t, group):
    """gloo has no device all_gather: device tensors go through the host there (CPU tests, the two-ranks-on-one-GPU
    test); with RCCL the tensor is used where it is."""
    import torch.distributed as dist

    return t.cpu() if t.is_cuda and dist.get_backend(group) != "nccl" else t


def gather_logits(local_logits, vocab, group=None):
    """All-gather of the vocab-sharded logits (torch.distributed; backend "nccl" = RCCL on the GPUs, gloo on CPU)."""
    import torch
    import torch.distributed as dist

    dev = local_logits.device
    local_logits = _host_staged(local_logits, group)
    if local_logits.device != dev:
        return gather_logits(local_logits, vocab, group).to(dev)
    world = dist.get_world_size(group)
    per = (vocab + world - 1) // world
    buf = torch.zeros(per, dtype=local_logits.dtype, device=local_logits.device)
    buf[:local_logits.numel()] = local_logits
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    return torch.cat(parts)[:vocab]


def greedy_token(local_logits, vocab, group=None):
    """argmax over the vocab-sharded logits WITHOUT gathering them: every rank contributes one (max, global index)
    pair — 16 bytes per rank instead of vocab / world floats ("argmax-reduce for greedy"; 128 KiB vs
    128 B per token for a 32000-row lm_head on 8 ranks). Ties go to the lowest index, like a single-device argmax over
    the gathered row. Returns a 0-d int64 tensor on the logits' device; no host synchronisation."""
    import torch
    import torch.distributed as dist

    dev = local_logits.device
    local_logits = _host_staged(local_logits, group)
    if local_logits.device != dev:
        return greedy_token(local_logits, vocab, group).to(dev)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = (vocab + world - 1) // world
    n = max(0, min(per, vocab - rank * per, local_logits.numel()))  # live rows of this rank's (padded) shard
    pair = torch.empty(2, dtype=torch.float64, device=local_logits.device)
    if n > 0:
        x = torch.nan_to_num(local_logits[:n].to(torch.float64), nan=float("-inf"))  # a NaN never wins (argmax of NaN is not a token)
        val = x.max()
        ar = torch.arange(n, dtype=torch.float64, device=x.device)
        first = torch.where(x == val, ar, torch.full_like(ar, float(n))).min()  # lowest index among equal maxima
        pair[0], pair[1] = val, first + rank * per
    else:
        pair[0], pair[1] = float("-inf"), float(vocab)
    pairs = [torch.empty_like(pair) for _ in range(world)]
    dist.all_gather(pairs, pair, group=group)
    allp = torch.stack(pairs)  # [world, 2]
    top = allp[:, 0].max()
    # global indices grow with the rank, so the lowest index among the maximal pairs is the first occurrence overall
    return torch.where(allp[:, 0] == top, allp[:, 1], torch.full_like(allp[:, 1], float(vocab))).min().to(torch.int64)


class TPDecoder:
    """One rank of a tensor-parallel decoder: a WoqDecoderEngine over this rank's shards.

    Two transports for the per-layer exchange:
      * `comm` (runtime/comm.py DeviceComm) — the production form: the engine issues the all-reduce kernels and the
        greedy-token exchange itself, `step()` is ONE native call (or one graph replay after `capture()`), nothing
        crosses the host per collective;
      * no comm — host-driven: the engine's sub-blocks with one `dist.all_reduce` (RCCL) between them
        (engine.step_tp), the token from `greedy_token`. Kept as the fallback when the fabric self-test fails and as
        the cross-check of the device path.
    `vocab` is the full vocabulary; the engine was built with this rank's lm_head rows (`shard_vocab`)."""

    def __init__(self, engine, vocab, group=None, comm=None):
        import torch.distributed as dist

        self.engine, self.vocab, self.group, self.comm = engine, vocab, group, comm
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        self.vocab_offset = rank * ((vocab + world - 1) // world)
        if comm is not None:
            engine.bind_comm(comm, self.vocab_offset)
        self._since_check = 0

    CHECK_EVERY = 64  # device steps between two reads of the exchange's sticky status

    def check_status(self):
        """The device exchange never hangs: a peer that stays silent past the timeout contributes 0 and raises a
        sticky status word — after which the ranks' hidden states and KV caches have diverged. Read it (one small
        host synchronisation) and refuse to go on. Called at prompt-pass / capture / generation boundaries and every
        CHECK_EVERY steps; call it yourself after a burst of `engine.replay`."""
        self._since_check = 0
        if self.comm is not None and self.comm.status() != 0:
            raise RuntimeError("QBits: the tensor-parallel device exchange timed out on this rank (a peer stalled): "
                               "the step's results are invalid; re-create the communicator or use the RCCL transport")

    def prefill(self, tokens, start_pos=0):
        """Prompt pass over this rank's shards: the engine's native prefill with the all-reduce seam bound to the
        process group (sum of the row-parallel partials after o_proj / down_proj, [rows, hidden] fp32 each — RCCL:
        this one is bandwidth-bound), then the vocab-sharded last-position logits all-gathered. Leaves the engine
        where single-GPU `prefill(greedy=True)` leaves it: token = argmax of sequence 0's row, pos = start + T, so a
        following `step()` continues the sequence. Returns logits [n_seq, vocab]."""
        import torch

        import torch.distributed as dist

        e = self.engine
        if self.comm is not None and dist.get_backend(self.group) != "nccl":
            local = e.prefill(tokens, start_pos=start_pos, greedy=False)  # no RCCL: rows through the device comm
        else:
            e.bind_allreduce(self.group)
            try:
                local = e.prefill(tokens, start_pos=start_pos, greedy=False)
            finally:
                e.unbind_allreduce()
        self.check_status()
        nxt = greedy_token(local[0][:e.cfg.vocab], self.vocab, self.group)
        e.token.copy_(nxt.to(torch.int32).reshape(e.token.shape))
        e.pos.add_(1)  # prefill(greedy=False) left start + T - 1
        return torch.stack([gather_logits(row[:e.cfg.vocab], self.vocab, self.group) for row in local])

    def capture(self):
        """Device transport only: capture the whole tensor-parallel token (5 kernels + 2 all-reduce kernels per layer,
        head, token exchange) into one hipGraph. Collective on all ranks (the capture runs one eager step first)."""
        if self.comm is None:
            raise RuntimeError("QBits: graph capture of a tensor-parallel step needs the device communicator")
        self.engine.capture(greedy=True)
        self.check_status()

    def step(self, greedy=True, return_logits=True):
        """One token. greedy + return_logits=False is the production form: the next token is agreed from one
        (max, global index) pair per rank and written to the engine's token slot on the device — no logits gather, no
        host round trip. With return_logits the full row is all-gathered as well (tests, sampling)."""
        import torch

        e = self.engine
        if self.comm is not None:
            self._since_check += 1
            if self._since_check >= self.CHECK_EVERY:
                self.check_status()
            if greedy and not return_logits and e.captured and e.launch == "graph":
                e.replay_graph(1)
                return None
            e.step(greedy=greedy)  # all-reduces and, when greedy, the token exchange run inside the native step
            if return_logits:
                self.check_status()  # the host reads results here anyway
            return gather_logits(e.logits[:e.cfg.vocab], self.vocab, self.group) if return_logits else None
        e.step_tp(self.group, greedy=False)  # logits of this rank's vocab shard
        local = e.logits[:e.cfg.vocab]
        logits = gather_logits(local, self.vocab, self.group) if return_logits else None
        if greedy:
            nxt = greedy_token(local, self.vocab, self.group)  # the same value on every rank
            e.token.copy_(nxt.to(torch.int32).reshape(e.token.shape))
            e.pos.add_(1)
        return logits
