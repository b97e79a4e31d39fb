"""Multi-process (world_size 2, gloo, CPU) check of the tensor-parallel plan in runtime/tp.py.

No GPU here, so the per-rank "device" arithmetic is the CPU oracle — used strictly as the checker of the HOST logic
under test: which slices each rank owns (whole heads, whole quantisation groups), where the two all-reduces per layer
go, the rank-0-carries-the-residual rule of the engine (woq_engine.hip engine_attn_block / engine_mlp_block), and the
vocab-sharded logits all-gather. The sharded run must reproduce the unsharded oracle decoder.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


GEOMETRIES = {
    # the round-2 toy: 2 kv heads, cut in two
    "toy": dict(hidden=128, inter=256, heads=4, kv_heads=2, head_dim=32, layers=2, vocab=100, eps=1e-5, theta=10000.0),
    # Llama-2-70B's head / group geometry scaled down (BASELINE configs[3]): 8 kv heads — ONE per rank
    # at tensor-parallel degree 8 — with 2 query heads each; the row-parallel cuts fall on quantisation-group boundaries
    # with an ODD number of groups per rank on the MLP side (70B: 28672 / 8 = 28 groups of 128; here 768 / 8 = 3 groups
    # of 32) and the vocabulary does not divide by the rank count
    "70b_like": dict(hidden=512, inter=768, heads=16, kv_heads=8, head_dim=32, layers=2, vocab=203, eps=1e-5,
                     theta=10000.0),
}


def _model(seed=0, group=32, asym=True, geom="toy"):
    from oracle import woq_oracle as orc

    cfg = dict(GEOMETRIES[geom])
    rng = np.random.default_rng(seed)
    H, I, NH, KV, D = cfg["hidden"], cfg["inter"], cfg["heads"], cfg["kv_heads"], cfg["head_dim"]
    layers = []
    for _ in range(cfg["layers"]):
        shapes = dict(q=(H, NH * D), k=(H, KV * D), v=(H, KV * D), o=(NH * D, H), gate=(H, I), up=(H, I), down=(I, H))
        parts = {n: orc.rtn_quantize(rng.standard_normal(s).astype(np.float32) * 0.05, False, group, asym)
                 for n, s in shapes.items()}
        layers.append(dict(parts=parts, ln1=(1 + 0.1 * rng.standard_normal(H)).astype(np.float32),
                           ln2=(1 + 0.1 * rng.standard_normal(H)).astype(np.float32)))
    embed = rng.standard_normal((cfg["vocab"], H)).astype(np.float32)
    lm = (rng.standard_normal((cfg["vocab"], H)) * 0.1).astype(np.float32)
    norm = (1 + 0.1 * rng.standard_normal(H)).astype(np.float32)
    return cfg, layers, embed, lm, norm, group


def _blobs(parts, group):
    from oracle import woq_oracle as orc

    return {n: orc.repack(q, s, z, None, group) for n, (q, s, z) in parts.items()}


def _rank_main(rank, world, port, out_dir, geom="toy"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from intel_extension_for_transformers_amd.runtime import tp
    from oracle import woq_oracle as orc

    torch.set_num_threads(1)
    cfg, layers, embed, lm, norm, group = _model(geom=geom)
    NH, KV, D = cfg["heads"] // world, cfg["kv_heads"] // world, cfg["head_dim"]
    shards = []
    for ly in layers:
        sp = tp.shard_llama_layer(ly["parts"], rank, world, cfg["heads"], cfg["kv_heads"], D, group)
        shards.append(dict(b=_blobs(sp, group), ln1=ly["ln1"], ln2=ly["ln2"]))
    lm_local = tp.shard_vocab(lm, rank, world)
    kc = [np.zeros((0, KV, D), np.float32) for _ in layers]
    vc = [np.zeros((0, KV, D), np.float32) for _ in layers]

    def allreduce(a):
        t = torch.from_numpy(np.ascontiguousarray(a))
        dist.all_reduce(t)
        return t.numpy()

    toks = [3, 17, 42, 7]
    got = []
    for pos, tok in enumerate(toks):
        h = embed[tok].reshape(1, -1).copy()
        for li, ly in enumerate(shards):
            b = ly["b"]
            x = orc.rmsnorm(h, ly["ln1"], cfg["eps"])
            q = orc.rope(orc.woq_linear(x, b["q"]).reshape(1, NH, D), [pos], cfg["theta"])
            k = orc.rope(orc.woq_linear(x, b["k"]).reshape(1, KV, D), [pos], cfg["theta"])
            v = orc.woq_linear(x, b["v"]).reshape(1, KV, D)
            kc[li] = np.concatenate([kc[li], k], 0)
            vc[li] = np.concatenate([vc[li], v], 0)
            a = orc.attn_decode(q[0], kc[li], vc[li]).reshape(1, NH * D)
            part = orc.woq_linear(a, b["o"])            # row-parallel partial sum
            if rank == 0:
                part = part + h                          # rank 0 carries the residual: added exactly once
            h = allreduce(part)                          # all-reduce #1 of the layer
            x = orc.rmsnorm(h, ly["ln2"], cfg["eps"])
            act = orc.silu_mul(orc.woq_linear(x, b["gate"]), orc.woq_linear(x, b["up"]))
            part = orc.woq_linear(act, b["down"])
            if rank == 0:
                part = part + h
            h = allreduce(part)                          # all-reduce #2
        x = orc.rmsnorm(h, norm, cfg["eps"])
        local = (x.astype(np.float64) @ lm_local.astype(np.float64).T).astype(np.float32)[0]
        got.append(tp.gather_logits(torch.from_numpy(local), cfg["vocab"]).numpy())
    if rank == 0:
        np.save(os.path.join(out_dir, "tp_logits.npy"), np.stack(got))
    dist.barrier()
    dist.destroy_process_group()


def test_tp2_matches_unsharded_oracle(tmp_path):
    from oracle import woq_oracle as orc

    world, port = 2, 29500 + (os.getpid() % 2000)
    mp.spawn(_rank_main, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "tp_logits.npy")
    cfg, layers, embed, lm, norm, group = _model()
    full = orc.LlamaOracle(cfg, embed, [dict(_blobs(ly["parts"], group), ln1=ly["ln1"], ln2=ly["ln2"])
                                        for ly in layers], norm, lm)
    for pos, tok in enumerate([3, 17, 42, 7]):
        ref = full.forward_token(tok, pos)
        # same fp32 oracle arithmetic, different summation split (2 partial sums + all-reduce): 1e-5 relative
        assert np.abs(got[pos] - ref).max() <= 1e-5 * np.abs(ref).max() + 1e-6
        assert int(got[pos].argmax()) == int(ref.argmax())


@pytest.mark.parametrize("world", [4, 8])
def test_tp4_tp8_match_unsharded_oracle_at_the_70b_head_geometry(tmp_path, world):
    """the plan at tensor-parallel degrees 4 and 8 on the 70B head / group geometry (one kv head per
    rank at 8), gloo ranks on the CPU, against the unsharded oracle decoder. No multi-GPU hardware has run this path:
    the device exchange (csrc/woq_comm.hip) is covered separately by two ranks on one GPU (tests/test_gpu_tp_device.py)."""
    from oracle import woq_oracle as orc

    port = 33500 + (os.getpid() % 2000) + world
    mp.spawn(_rank_main, args=(world, port, str(tmp_path), "70b_like"), nprocs=world, join=True)
    got = np.load(tmp_path / "tp_logits.npy")
    cfg, layers, embed, lm, norm, group = _model(geom="70b_like")
    assert cfg["kv_heads"] // world >= 1 and (cfg["inter"] // world) % group == 0 and cfg["vocab"] % world != 0
    full = orc.LlamaOracle(cfg, embed, [dict(_blobs(ly["parts"], group), ln1=ly["ln1"], ln2=ly["ln2"])
                                        for ly in layers], norm, lm)
    for pos, tok in enumerate([3, 17, 42, 7]):
        ref = full.forward_token(tok, pos)
        assert np.abs(got[pos] - ref).max() <= 1e-5 * np.abs(ref).max() + 1e-6  # `world` partial sums + all-reduce
        assert int(got[pos].argmax()) == int(ref.argmax())


def test_sharding_rules():
    from intel_extension_for_transformers_amd.runtime import tp

    q = np.zeros((256, 96), np.int8)
    s = np.zeros((8, 96), np.float32)
    a, b, c = tp.shard_rows(q, s, None, 1, 2, 32)
    assert a.shape == (128, 96) and b.shape == (4, 96) and c is None
    with pytest.raises(ValueError):
        tp.shard_rows(np.zeros((192, 8), np.int8), np.zeros((3, 8), np.float32), None, 0, 2, 64)  # cuts a group
    with pytest.raises(ValueError):
        tp.shard_columns(q, s, None, 0, 5)
    # Llama-2-70B geometry at TP=8: every cut is on a group-128 boundary
    for k in (8192, 28672):
        assert (k // 8) % 128 == 0
    assert tp.shard_vocab(np.zeros((100, 4)), 1, 2).shape == (50, 4)


def _greedy_rank(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from intel_extension_for_transformers_amd.runtime import tp

    rng = np.random.default_rng(7)
    picks = []
    for vocab in (100, 101, 7, 2, 1):  # even split, ragged last shard, tiny vocabularies (an empty shard on rank 1)
        for case in range(4):
            row = rng.standard_normal(vocab).astype(np.float32)
            if case == 1:  # the maximum twice, once per shard: the lower index wins
                row[vocab - 1] = row[0] = 9.0
            if case == 2 and vocab > 3:  # twice inside one shard
                row[2] = row[1] = 9.0
            if case == 3:  # all equal
                row[:] = 0.5
            local = torch.from_numpy(np.ascontiguousarray(tp.shard_vocab(row[:, None], rank, world)[:, 0]))
            got = int(tp.greedy_token(local, vocab))
            gathered = tp.gather_logits(local, vocab).numpy()
            assert np.array_equal(gathered, row)
            picks.append((got, int(np.argmax(row))))
    if rank == 0:
        np.save(os.path.join(out_dir, "picks.npy"), np.array(picks))
    dist.barrier()
    dist.destroy_process_group()


def test_greedy_token_pair_exchange_matches_argmax_of_gathered_logits(tmp_path):
    """tp.greedy_token: one (max, global index) pair per rank instead of the logits all-gather; equals numpy's argmax
    (first maximum) of the full row for even and ragged vocab splits, ties across and inside shards, and a rank whose
    shard is empty."""
    world, port = 2, 31500 + (os.getpid() % 2000)
    mp.spawn(_greedy_rank, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    picks = np.load(tmp_path / "picks.npy")
    assert len(picks) == 20 and np.array_equal(picks[:, 0], picks[:, 1])


def test_bench_py_gpus_flag_spawns_its_own_ranks():
    """`python bench.py --gpus N` with no launcher around it must become N ranks: the command it
    re-executes is the driver's own N > 1 command line; with a launcher (WORLD_SIZE set) or N = 1 it spawns nothing, and
    a launcher that started a different number of ranks than --gpus is refused before any GPU work."""
    import subprocess

    sys.path.insert(0, ROOT)
    import bench

    cmd = bench.spawn_command(8, {}, ["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-7].endswith("bench.py") and cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert bench.spawn_command(8, {"MASTER_PORT": "29611"}, [])[cmd.index("--master-port") + 1] == "29611"
    assert bench.spawn_command(1, {}, []) is None
    assert bench.spawn_command(8, {"WORLD_SIZE": "8"}, []) is None
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert res.returncode != 0 and "--gpus 4" in res.stderr and "2 rank" in res.stderr, res.stderr[-500:]


def test_bench_py_refuses_more_gpus_than_are_visible():
    """`python bench.py --gpus 8` on a box with fewer than 8 visible devices (here: none) must say so and exit non-zero
    BEFORE it spawns ranks or touches a device — the N > 1 line is `unmeasured on hardware` until
    a node runs it, and a silent fallback to fewer ranks would print a line with the wrong n_gpus."""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert res.returncode != 0 and res.stdout.strip() == ""
    assert "--gpus 8" in res.stderr and "visible" in res.stderr, res.stderr[-500:]
