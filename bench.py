#!/usr/bin/env python3
"""bench.py — the driver's measurement contract for the int4-WOQ decode hot path.

A "step" is one batch-1 decode token of a Llama-2-7B-shaped decoder (hidden 4096, inter 11008, 32 heads,
32 layers, vocab 32000) whose 224 linears are int4 sym group-128 WQH1 blobs with fp16 scales (BASELINE.json
configs[1]); lm_head stays fp16 like the reference (utils/config.py:836-837). Weights are synthetic random-init
(there are no checkpoints and no network) and already resident in HBM when the timed region starts.

  python bench.py --gpus N --steps K --warmup W

N > 1: the 7B batch-1 decode path does not shard ("replicas only" for configs 1-3,5), so every
rank runs an independent replica on its own GPU, no data-path collective; value = N*K tokens / max-over-ranks
time ("weak" scaling). Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     : dominant kernel = the int4 decode GEMV (woq::gemv_tile_kernel, csrc/woq_gemv_i8.hip). achieved = algorithmic bytes per
                 launch (int4 payload + fp16 scales, SURVEY.md §8(d): 3 339 190 272 B / 128 launches per token)
                 / average launch duration: HIP events on the launch stream around passes of all 128 GEMV launches back to back
                 (boundaries between launches included, no per-launch event overhead). peak = 8000 GB/s.
                 traffic = HBM bytes per launch from the rocprofv3 --pmc pass (profiles/*_pmc_traffic.json), or null.
  cpu_baseline : the oracle's streaming int4 GEMV (oracle/woq_oracle.c orc_woq_gemv_stream, kind "port": the
                 reference's BesTLA kernels are not buildable here) timed on this host's cores over ONE decoder
                 layer's four fused linears (same bytes as the GPU's layer 0), extrapolated to 32 layers.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LLAMA2_7B = dict(hidden=4096, inter=11008, heads=32, kv_heads=32, head_dim=128, layers=32, vocab=32000)
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes_per_token(cfg, group=128, scale_bytes=2, asym=False):
    """int4 payload + scales (+ 4-bit zero points) of every quantised linear, unpadded."""
    h, i = cfg["hidden"], cfg["inter"]
    qkv_n = (cfg["heads"] + 2 * cfg["kv_heads"]) * cfg["head_dim"]
    shapes = [(h, qkv_n), (cfg["heads"] * cfg["head_dim"], h), (h, 2 * i), (i, h)]
    tot = 0
    for k, n in shapes:
        g = (k + group - 1) // group
        tot += k * n // 2 + g * n * scale_bytes + (g * n // 2 if asym else 0)
    return tot * cfg["layers"]


def cpu_baseline(eng, cfg, budget_s=12.0):
    """Time the oracle's streaming int4 GEMV on the host cores over layer 0's four fused linears."""
    import numpy as np

    from oracle import woq_oracle as orc

    blobs = [b.cpu().numpy().view(np.uint8) for b in eng._keep[:4]]  # qkv, o, gate_up, down of layer 0
    rng = np.random.default_rng(0)
    xs = [rng.standard_normal(orc.header(b)["K"]).astype(np.float32) for b in blobs]
    for b, x in zip(blobs, xs):  # warm (page in, thread pool)
        orc.woq_gemv_stream(x, b)
    reps, t0 = 0, time.perf_counter()
    while True:
        for b, x in zip(blobs, xs):
            orc.woq_gemv_stream(x, b)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or reps >= 100000:
            break
    t_layer = dt / reps
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    return {
        "value": 1.0 / (t_layer * cfg["layers"]),
        "unit": "tokens/s",
        "cores": cores,
        "kind": "port",
        "sample": "oracle orc_woq_gemv_stream (fp32-accumulate streaming int4 GEMV, OpenMP) over the 4 fused "
                  "linears of ONE Llama-2-7B layer x %d reps (%.1f s), extrapolated x32 layers; lm_head/attention "
                  "excluded" % (reps, dt),
    }


def prefill_measure(eng, cfg, n_seq, T, reps=2):
    """Prompt pass of the same model (north_star: MFMA utilisation on the prefill side): n_seq x T synthetic tokens
    through all layers (int4 x fp16-operand MFMA GEMMs + causal attention), after the decode timing. Not `value`."""
    import torch

    g = torch.Generator().manual_seed(4321)
    toks = torch.randint(0, cfg["vocab"], (n_seq, T), generator=g).cuda()
    eng.prefill(toks)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.prefill(toks)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    h, i, hd = cfg["hidden"], cfg["inter"], cfg["head_dim"]
    params = cfg["layers"] * (h * (cfg["heads"] + 2 * cfg["kv_heads"]) * hd + cfg["heads"] * hd * h + 3 * h * i)
    lin = 2.0 * params * n_seq * T
    att = cfg["layers"] * n_seq * 4.0 * cfg["heads"] * hd * T * (T + 1) / 2
    return {
        "workload": "prompt pass, %d x %d tokens, same int4 g128 weights, fp16-operand MFMA GEMMs + causal attention, "
                    "lm_head on the last positions" % (n_seq, T),
        "tokens_per_s": n_seq * T / dt, "ms": dt * 1e3,
        "achieved_tflops": (lin + att) / dt / 1e12, "linear_tflops": lin / dt / 1e12,
        "peak_tflops": 2500.0, "mfma_frac": (lin + att) / dt / 1e12 / 2500.0,
    }


def read_traffic():
    """HBM bytes per dominant-kernel launch from the committed PMC pass (None if there is none)."""
    pdir = os.path.join(ROOT, "profiles")
    best = None
    if os.path.isdir(pdir):
        for f in sorted(os.listdir(pdir)):
            if f.endswith("_pmc_traffic.json"):
                best = os.path.join(pdir, f)
    if not best:
        return None
    try:
        with open(best) as fh:
            return float(json.load(fh)["hbm_bytes_per_launch"])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--prompt", type=int, default=32, help="positions already in the KV cache when timing starts")
    ap.add_argument("--layers", type=int, default=LLAMA2_7B["layers"])
    ap.add_argument("--prefill-seqs", type=int, default=4, help="sequences in the prompt-pass measurement (0 = skip)")
    ap.add_argument("--prefill-len", type=int, default=2048, help="tokens per sequence in the prompt-pass measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from intel_extension_for_transformers_amd.runtime.engine import WoqDecoderEngine, synth_llama_weights

    cfg = dict(LLAMA2_7B, layers=args.layers)
    max_ctx = 1 << max(9, (args.prompt + args.warmup + args.steps + 8).bit_length())
    want_prefill = args.prefill_seqs > 0 and world == 1
    if want_prefill:
        max_ctx = max(max_ctx, args.prefill_len)
    eng = WoqDecoderEngine(cfg["hidden"], cfg["inter"], cfg["heads"], cfg["kv_heads"], cfg["head_dim"], cfg["layers"],
                           cfg["vocab"], max_ctx=max_ctx, max_batch=max(1, args.prefill_seqs) if want_prefill else 1)
    synth_llama_weights(eng, cfg["hidden"], cfg["inter"], cfg["heads"], cfg["kv_heads"], cfg["head_dim"],
                        cfg["layers"], cfg["vocab"], group=128, sym=True, scale_dtype="fp16", seed=1234 + rank)

    # synthetic prompt: feed `prompt` random tokens through the decode path so the KV cache holds real entries
    g = torch.Generator().manual_seed(1234)
    prompt = torch.randint(0, cfg["vocab"], (args.prompt,), generator=g).tolist()
    for i, t in enumerate(prompt):
        eng.token.fill_(int(t))
        eng.pos.fill_(i)
        eng.step(greedy=(i == len(prompt) - 1))
    use_graph = not args.no_graph
    if use_graph:
        eng.capture(greedy=True)

    def run(n):
        if use_graph:
            eng.replay(n)
        else:
            for _ in range(n):
                eng.step(greedy=True)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    fence()
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tok_s = world * args.steps / elapsed

    if rank == 0:
        qbytes = algorithmic_bytes_per_token(cfg)
        # dominant kernel, timed alone: HIP events on the launch stream around 4 passes of the 128 launches
        ms, by, n_launch = eng.time_gemv(reps=4)
        us_per_launch = ms * 1e3 / (4 * n_launch)
        achieved = (by / n_launch) / (us_per_launch * 1e-6) / 1e9
        out = {
            "metric": "decode tokens/sec + achieved HBM GB/s, Llama-2-7B int4 WOQ, batch=1",
            "value": tok_s,
            "unit": "tokens/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int4 weights x fp32 activations as 3 x int8 fixed-point limbs on i8 MFMA (exact int32 tile sums), fp32 across tiles",
            "data": "synthetic (random-init int4 weights of the Llama-2-7B shape, random prompt ids)",
            "config": {
                "workload": "Llama-2-7B int4 sym group_size=128 fp16 scales, batch=1 greedy decode, prompt %d, "
                            "lm_head fp16 unquantised, KV fp16%s" % (args.prompt, "" if args.layers == 32 else
                                                                     " [REDUCED to %d layers]" % args.layers),
                "global_batch": world,
                "parallelism": "replicas x%d (path does not shard at 7B)" % world if world > 1 else "single GPU",
                "hipgraph": use_graph,
            },
            "hbm_gbps_quantized_weight_stream": qbytes * (tok_s / world) / 1e9,
            "hbm_frac_of_peak_end_to_end": qbytes * (tok_s / world) / 1e9 / HBM_PEAK_GBPS,
            "roofline": {
                "bound": "hbm",
                "kernel": "woq::gemv_tile_kernel (int4 GEMV, M=1, csrc/woq_gemv_i8.hip)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": read_traffic(),
                "us_per_launch": us_per_launch,
                "algorithmic_bytes_per_launch": by / n_launch,
                "launches_per_token": n_launch,
            },
        }
        if args.prefill_seqs > 0 and world == 1:
            out["prefill"] = prefill_measure(eng, cfg, args.prefill_seqs, args.prefill_len)
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only (torchrun also pins OMP_NUM_THREADS=1)
            out["cpu_baseline"] = cpu_baseline(eng, cfg)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
