#!/usr/bin/env python3
"""bench.py — the driver's measurement contract for the int4-WOQ decode hot path.

  python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one batch-1 greedy decode token, weights synthetic (random-init, no checkpoints / network) and resident in
HBM when the timed region starts; barrier + synchronize on both sides, max over ranks, ONE JSON line from rank 0.
The K timed steps are K replays of the captured hipGraph on the current stream (token / position chained on the device,
no host work per token); `--eager` times ONE eager burst instead (woq_engine_steps: K x ~131 launches back to back) — the
same device time since round 4 found and removed what made graph replays ~1 us per kernel boundary slower (cross-stream
event waits in front of the replay, profiles/r04ab_stream_mode_probe.txt); the N = 1 line reports both (`launch_modes`).

Which workload (`--workload auto`, the default):
  * N = 1 on a one-GPU box  -> BASELINE.json configs[1]: Llama-2-7B int4 sym g128 (the configuration the metric is
    quoted on). The same line carries `extra_configs`: configs[2] decode (g32 asym) and its 32 x 2048 prompt pass,
    configs[4] (Mistral-7B shape, fp8 KV, 8k context, chunked 4 x 2048 prompt pass), and configs[3]'s single-GPU point
    (Llama-2-70B on one GPU); `parity` (logits of the measured 32-layer model against the CPU oracle on the same
    (q, scale, zp)), `roofline`, `cpu_baseline`.
  * N > 1 -> configs[3]: Llama-2-70B int4 sym g128 at tensor-parallel degree N (strong scaling: the SAME model cut N
    ways; column-parallel q/k/v/gate/up, row-parallel o/down, 2 all-reduces per layer + 1 token exchange per token,
    SURVEY.md §8(e)). The exchange runs on the device over xGMI as kernels of the native step (csrc/woq_comm.hip) after a
    start-up self-test; if that fails on this node the run falls back to host-issued RCCL all-reduces and says so.
  * N = 1 on a multi-GPU node (the first point of the driver's 1/2/4/8 sweep) -> the same 70B model on one GPU, so
    that the sweep's N = 1 value is the strong-scaling base; the 7B number rides along in `extra_configs`.

The stdout line is <= 6 KB (`compact_line`): the contract's scalars, `config`, `dtype`, `roofline`, `cpu_baseline`,
`prefill` / `parity` as numbers, `configs_summary`. Everything else (extra_configs, launch_modes, prose, per-config
objects) is written to `bench_extra.json` next to this file (and to gpurun_out/ when present); DESIGN.md §6.
roofline.achieved = algorithmic bytes per launch of the dominant kernel (woq::gemv_xqs_kernel) / its average duration
from HIP events on the launch stream around back-to-back passes of the step's own launches; roofline.ceiling = the
same launches as load-only twins / empty kernels, same run.
"""
import argparse
import gc
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LLAMA2_7B = dict(name="Llama-2-7B", hidden=4096, inter=11008, heads=32, kv_heads=32, head_dim=128, layers=32, vocab=32000)
LLAMA2_70B = dict(name="Llama-2-70B", hidden=8192, inter=28672, heads=64, kv_heads=8, head_dim=128, layers=80,
                  vocab=32000)
MISTRAL_7B = dict(name="Mistral-7B", hidden=4096, inter=14336, heads=32, kv_heads=8, head_dim=128, layers=32,
                  vocab=32000)
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_COPY_GBPS = 6290.0   # measured float4-copy ceiling on the same chip (same guide)
MFMA_PEAK_TFLOPS = 2500.0
METRIC = "decode tokens/sec + achieved HBM GB/s, Llama-2-7B int4 WOQ, batch=1"
DTYPE = "int4 weights x fp32 activations (3 int8 limbs on i8 MFMA, exact int32 tile sums, fp32 across tiles)"


def algorithmic_bytes_per_token(cfg, group=128, scale_bytes=2, asym=False, tp=1):
    """int4 payload + scales (+ 4-bit zero points) of every quantised linear, unpadded; per rank."""
    h, i = cfg["hidden"], cfg["inter"] // tp
    heads, kv = cfg["heads"] // tp, max(1, cfg["kv_heads"] // tp)
    shapes = [(h, (heads + 2 * kv) * cfg["head_dim"]), (heads * cfg["head_dim"], h), (h, 2 * i), (i, h)]
    tot = 0
    for k, n in shapes:
        g = (k + group - 1) // group
        tot += k * n // 2 + g * n * scale_bytes + (g * n // 2 if asym else 0)
    return tot * cfg["layers"]


def linear_params(cfg):
    h, i, hd = cfg["hidden"], cfg["inter"], cfg["head_dim"]
    return cfg["layers"] * (h * (cfg["heads"] + 2 * cfg["kv_heads"]) * hd + cfg["heads"] * hd * h + 3 * h * i)


def build_engine(cfg, group=128, sym=True, max_ctx=512, max_batch=1, kv_dtype=None, tp_rank=0, tp=1, seed=1234,
                 layers=None, weight_dtype="int4_clip", compute_dtype="fp32", act_order=False):
    import torch

    from intel_extension_for_transformers_amd.runtime.engine import WoqDecoderEngine, synth_llama_weights

    n_layers = layers or cfg["layers"]
    heads, kv, inter, vocab = cfg["heads"] // tp, max(1, cfg["kv_heads"] // tp), cfg["inter"] // tp, cfg["vocab"] // tp
    eng = WoqDecoderEngine(cfg["hidden"], inter, heads, kv, cfg["head_dim"], n_layers, vocab, max_ctx=max_ctx,
                           max_batch=max_batch, kv_dtype=kv_dtype or torch.float16, tp_rank=tp_rank, tp_size=tp)
    synth_llama_weights(eng, cfg["hidden"], inter, heads, kv, cfg["head_dim"], n_layers, vocab, group=group, sym=sym,
                        scale_dtype="fp16", seed=seed + tp_rank, embed_vocab=cfg["vocab"], shared_seed=seed,
                        weight_dtype=weight_dtype, compute_dtype=compute_dtype, act_order=act_order)
    return eng


LINE_LIMIT = 6144  # bytes of the ONE stdout JSON line (the driver's parser dropped a 21.7 KB line)
EXTRA_FILE = "bench_extra.json"


def _r(x, nd=4):
    """floats to `nd` significant digits (the line is a record, not a transport for doubles)"""
    if isinstance(x, float):
        return float(round(x)) if abs(x) >= 1e5 else float("%.*g" % (nd + 2, x))
    return x


def compact_roofline(r):
    """contract keys of `roofline` + the numbers the judge re-derives it from; no prose"""
    if not r:
        return r
    out = {k: _r(r[k]) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "us_per_launch",
                                   "algorithmic_bytes_per_launch", "launches_per_token") if k in r}
    out["kernel"] = str(r.get("kernel", "")).split(" (")[0]
    td = r.get("traffic_detail") or {}
    if td:
        out["traffic_source"] = td.get("source")
        out["traffic_measured_in_this_run"] = bool(td.get("measured_in_this_run"))
        if td.get("command"):
            out["traffic_command"] = str(td["command"])[:100]
    bp = r.get("by_projection") or {}
    if bp:
        out["by_projection"] = {k: {"us": _r(v["us_per_launch"], 3), "frac": _r(v["frac"], 3)}
                                for k, v in bp.items() if isinstance(v, dict)}
        out["dominant_by_time"] = bp.get("dominant_by_time")
    c = r.get("ceiling") or {}
    if c:
        out["ceiling"] = {"load_only_twin_us": _r(c["load_only_twin_us_per_launch"], 3),
                          "load_only_twin_frac": _r(c["load_only_twin_frac"], 3),
                          "empty_kernel_us_in_graph": _r(c["empty_kernel_us_per_launch"], 3)}
    return out


def compact_line(out, extra_path=EXTRA_FILE):
    """The ONE stdout line, <= LINE_LIMIT bytes: the contract's scalars, `config`, `dtype`, `roofline`, `cpu_baseline`,
    `prefill` / `parity` as numbers, `configs_summary`; everything else (extra_configs, launch_modes, prose) is in
    `extra_path`, named in the line. tests/test_bench_line.py holds the size and the required keys."""
    first = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data", "config"]
    res = {k: _r(out[k], 6) if k in ("value", "ms_per_step") else out[k] for k in first if k in out}
    for k in ("value_128_steps", "hbm_gbps_quantized_weight_stream", "hbm_frac_of_peak_end_to_end", "hbm_gbps_per_gpu"):
        if k in out:
            res[k] = _r(out[k], 5)
    if "note" in out:
        res["note"] = str(out["note"])[:300]
    if "headline_7b" in out:
        h = out["headline_7b"]
        res["headline_7b"] = {"value": _r(h["value"], 5), "unit": h["unit"], "ms_per_step": _r(h["ms_per_step"], 5),
                              "steps": h["steps"], "workload": str(h["workload"])[:120],
                              "roofline": compact_roofline(h.get("roofline"))}
    if "roofline" in out:
        res["roofline"] = compact_roofline(out["roofline"])
    cb = out.get("cpu_baseline")
    if cb:
        res["cpu_baseline"] = {"value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                               "sample": str(cb["sample"])[:120], "host_gbps": _r(cb.get("host_gbps", 0.0), 3),
                               "reference_published_tokens_per_s": 27.9}
    pf = out.get("prefill")
    if pf:
        res["prefill"] = {"workload": ", ".join(str(pf.get("workload", "prompt pass")).split(", ")[:2]),
                          "tokens_per_s": _r(pf["tokens_per_s"], 5), "achieved_tflops": _r(pf["achieved_tflops"]),
                          "peak_tflops": pf["peak_tflops"], "mfma_frac": _r(pf["mfma_frac"])}
        if "dominant_gemm" in pf:
            res["prefill"]["dominant_gemm_us"] = _r(pf["dominant_gemm"]["kernel_us"])
            res["prefill"]["dominant_gemm_mfma_frac"] = _r(pf["dominant_gemm"]["kernel_mfma_frac"])
    par = out.get("parity")
    if par:
        p = {"decode_logits_max_abs": _r(par["decode"]["logits_max_abs"], 3),
             "decode_logits_over_max_logit": _r(par["decode"]["logits_max_abs_over_max_logit"], 3),
             "greedy_tokens_equal": bool(par["decode"]["greedy_tokens_equal"])}
        if "prefill" in par:
            p["prefill_gemm_worst_row_over_rowmax"] = _r(par["prefill"]["worst_row_err_over_rowmax"], 3)
            att = par["prefill"].get("attention")
            if att:
                p["prefill_attention_kv_rows_worst_over_max"] = _r(
                    max(v["layer1_kv_rows_worst_over_max"] for v in att["per_sequence"].values()), 3)
        res["parity"] = p
    if "fused_attention_launch" in out:
        res["engine_status"] = out["fused_attention_launch"].get("engine_status")
    if "configs_summary" in out:
        res["configs_summary"] = out["configs_summary"]
    res["extra"] = extra_path
    line = json.dumps(res, separators=(",", ":"))
    if len(line) > LINE_LIMIT:  # never print an over-long line: drop the optional objects, keep the contract
        for k in ("headline_7b", "note", "configs_summary", "prefill", "parity"):
            res.pop(k, None)
            line = json.dumps(res, separators=(",", ":"))
            if len(line) <= LINE_LIMIT:
                break
    return line


def emit(out):
    """full record -> bench_extra.json (repo root, and gpurun_out/ when that exists so that it travels back from a GPU
    box); the compact line -> stdout."""
    full = json.dumps(out, indent=1)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, EXTRA_FILE), "w") as fh:
                    fh.write(full)
            except OSError:
                pass
    print(compact_line(out), flush=True)


def configs_summary(out):
    """<= 600 characters: one number per BASELINE config, placed right in front of `roofline` so that a parser that keeps
    only names of `extra_configs`, or a log tail that starts mid-line, still carries every configuration's value
    (VERDICT r04 item 4-ii). tok/s = batch-1 decode tokens/s; pf = prompt pass as a fraction of 2.5 PF dense fp16."""
    parts = []
    if "value_128_steps" in out:
        parts.append("c1 7B g128 %d-step %.0f / 128-step %.0f tok/s" % (out["steps"], out["value"], out["value_128_steps"]))
    else:
        parts.append("c1 7B g128 %d-step %.0f tok/s" % (out["steps"], out["value"]))
    if "prefill" in out:
        parts.append("c1 pf 4x2048 %.3f" % out["prefill"]["mfma_frac"])
    for e in out.get("extra_configs", []):
        c = e.get("config", "")
        if c.startswith("configs[2]") and "decode_tokens_per_s" in e:
            parts.append("c2 g32asym %.0f tok/s" % e["decode_tokens_per_s"])
        elif c.startswith("configs[2]"):
            parts.append("c2 pf 32x2048 %.3f" % e["mfma_frac"])
        elif c.startswith("configs[1]") and "decode_tokens_per_s" in e:
            parts.append("c1@%s %.0f" % (e["context"].split()[0], e["decode_tokens_per_s"]))
        elif c.startswith("configs[4]") and "decode_tokens_per_s" in e:
            parts.append("c4 Mistral 8k fp8KV %.0f tok/s" % e["decode_tokens_per_s"])
        elif c.startswith("configs[4]"):
            parts.append("c4 pf 8k chunked %.3f" % e["mfma_frac"])
        elif c.startswith("configs[3]"):
            parts.append("c3 70B 1-GPU %.1f tok/s" % e["tokens_per_s"])
        elif c.startswith("SURVEY 8(f)-2"):
            parts.append("act-order %.0f" % e["decode_tokens_per_s"])
        elif c.startswith("SURVEY 8(f)-4") and "decode_tokens_per_s" in e:
            parts.append("%s %.0f" % (c.split("Llama-2-7B ")[1].split(",")[0].replace(" sym g128", "").replace("compute ", ""),
                                      e["decode_tokens_per_s"]))
    return "; ".join(parts)[:600]


def free_gpu():
    """after `del engine`: run the finalisers (woq_engine_destroy frees the engine's own buffers) and hand torch's
    cached blocks back, so the next configuration starts from an empty device."""
    import torch

    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


def feed_prompt(eng, vocab, n, step=None):
    """`n` random tokens through the decode path so that the KV cache holds real entries; the last step is greedy."""
    import torch

    g = torch.Generator().manual_seed(1234)
    prompt = torch.randint(0, vocab, (n,), generator=g).tolist()
    for i, t in enumerate(prompt):
        eng.token.fill_(int(t))
        eng.pos.fill_(i)
        (step or eng.step)(i == len(prompt) - 1)


LAST_TIMED_REGION = [0.0, 0.0]  # unix time of the most recent timed region (for tools/clock_sampler.py)
CONDITION_MS = 800.0             # --condition-ms
LAST_CONDITIONING = {}


def read_sclk_mhz():
    """Current shader clock of the busiest GPU in sysfs (pp_dpm_sclk's starred line), right after GPU work: the active
    device is the one that is not idling at ~100 MHz. None when sysfs does not expose it."""
    import glob
    import re

    best = None
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for ln in open(f):
                if "*" in ln:
                    m = re.search(r"(\d+)\s*Mhz", ln, re.I)
                    if m:
                        best = max(best or 0, int(m.group(1)))
        except OSError:
            pass
    return best


def condition_clocks(run, eng, fence, dist=None, ms=None):
    """Untimed steady-state conditioning BEFORE the warm-up steps. Measured (profiles/r04a_sclk_ramp.txt): an MI355X that
    has been idle sits at ~160 MHz sclk and needs ~0.6 s of continuous load to reach 2.4 GHz (2.19 GHz after 0.25 s), and
    bench.py's GPU phases before the timed region are bursts between host-only legs — a 20-step timed region (23 ms)
    sits entirely on that ramp. For THIS workload it turned out not to matter (same box, profiles/r04b_conditioning_ab.txt:
    855.1 vs 856.5 tokens/s at 20 steps with sclk 2394 vs <= 2176 MHz behind the region, 837.6 vs 837.1 at 128): the
    decode step waits on memory and launch boundaries (mclk / fclk do not ramp), not on shader-clock cycles — which also
    retires the "profiled runs are faster because of clocks" reading of VERDICT r03 2(e). The conditioning stays so
    that the line states its clock: the same captured step is replayed for `ms` milliseconds in chunks that stay
    inside max_ctx, then the device-side token / position are put back, so the warm-up + timed steps that follow see
    exactly the contexts they would have seen. The number of conditioning steps is agreed over the process group (every
    rank replays the same collectives)."""
    import torch

    ms = CONDITION_MS if ms is None else ms
    LAST_CONDITIONING.clear()
    if ms <= 0:
        return
    tok0, pos0 = eng.token.clone(), eng.pos.clone()
    room = int(eng.cfg.max_ctx) - int(pos0.item()) - 2
    chunk = max(1, min(64, room))
    fence()
    t0 = time.perf_counter()
    run(min(8, chunk))
    fence()
    per = max((time.perf_counter() - t0) / min(8, chunk), 1e-5)
    n = int(ms * 1e-3 / per) + 1
    if dist is not None:
        on = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([n], dtype=torch.int64, device=on)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n = int(t.item())
    done = 0
    t0 = time.perf_counter()
    while done < n:
        eng.token.copy_(tok0)
        eng.pos.copy_(pos0)
        k = min(chunk, n - done)
        run(k)
        done += k
    fence()
    eng.token.copy_(tok0)
    eng.pos.copy_(pos0)
    fence()
    LAST_CONDITIONING.update({"untimed_steps": done, "ms": (time.perf_counter() - t0) * 1e3,
                              "why": "sclk ramps from idle (~160 MHz) to 2.4 GHz over ~0.6 s of load "
                                     "(profiles/r04a_sclk_ramp.txt); measured effect on this workload: none "
                                     "(profiles/r04b_conditioning_ab.txt)"})


def timed(run, steps, warmup, fence, condition=None):
    """`condition` = (engine, dist | None): clock conditioning (above) before the W warm-up steps."""
    if condition is not None:
        condition_clocks(run, condition[0], fence, condition[1])
    run(warmup)
    fence()
    LAST_TIMED_REGION[0] = time.time()
    t0 = time.perf_counter()
    run(steps)
    fence()
    dt = time.perf_counter() - t0
    LAST_TIMED_REGION[1] = time.time()
    if condition is not None:
        LAST_CONDITIONING["sclk_mhz_after_timed_region"] = read_sclk_mhz()
    return dt


def gemv_roofline(eng, traffic=None, ceiling=True):
    """Dominant kernel: the batch-1 int4 GEMV. `achieved` = algorithmic bytes per launch / average launch duration from
    HIP events on the launch stream around passes of the step's OWN four GEMV launches per layer (same kernels,
    epilogues, XQ outputs and residual chaining as the step; issued eagerly back to back like the step's bursts, so the
    boundaries a token pays are inside). `ceiling`: the same launches with the arithmetic taken out, timed the same way in the same run — load-only
    twins (what this launch structure reaches as a pure stream) and empty kernels (what its launches cost)."""
    ms, by, n_launch = eng.time_gemv(reps=8)
    us = ms * 1e3 / (8 * n_launch)
    per_launch = by / n_launch
    achieved = per_launch / (us * 1e-6) / 1e9
    out = {
        "bound": "hbm",
        "kernel": "woq::gemv_xqs_kernel (int4 GEMV over XQ digit blocks, M=1, csrc/woq_gemv_xqs.h)" if eng.uses_xq()
                  else "woq::gemv_tile_kernel (int4 GEMV, M=1, csrc/woq_gemv_i8.hip)",
        "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
        "frac_of_measured_copy_ceiling": achieved / HBM_COPY_GBPS,
        "traffic": traffic["bytes"] if traffic else None, "traffic_detail": traffic,
        "us_per_launch": us, "algorithmic_bytes_per_launch": per_launch,
        "launches_per_token": n_launch,
        "timed": "HIP events on the launch stream around 8 passes of %d launches issued eagerly back to back (the step's "
                 "own kernel variants, chained as in the step; boundaries included) — the way the step itself is "
                 "issued; round 3 timed replays of a captured pass on a stream behind a cross-stream event wait, where kernel boundaries cost ~1 us more" % n_launch,
    }
    # the four projections one at a time (the instantiations rocprofv3 lists separately): a pass of ONE projection per
    # layer, back to back — each launch then starts behind a launch of its own kind, not behind its real predecessor
    proj = {}
    for bit, name in enumerate(("qkv", "o", "gate_up", "down")):
        pms, pby, pn = eng.time_gemv(reps=8, mask=1 << bit)
        pus = pms * 1e3 / (8 * pn)
        proj[name] = {"us_per_launch": pus, "algorithmic_bytes_per_launch": pby / pn,
                      "achieved": pby / pn / (pus * 1e-6) / 1e9, "frac": pby / pn / (pus * 1e-6) / 1e9 / HBM_PEAK_GBPS}
    dom = max(proj, key=lambda k: proj[k]["us_per_launch"])
    out["by_projection"] = dict(proj, dominant_by_time=dom,
                                note="`achieved` above is the average over all four; the single instantiation with the "
                                     "largest share of the step's time is `%s`" % dom)
    if ceiling and eng.uses_xq():
        us_twin = eng.time_twin(0, reps=8) * 1e3 / (8 * n_launch)
        us_empty = eng.time_twin(1, reps=8) * 1e3 / (8 * n_launch)
        out["ceiling"] = {
            "load_only_twin_us_per_launch": us_twin,
            "load_only_twin_gbps": per_launch / (us_twin * 1e-6) / 1e9,
            "load_only_twin_frac": per_launch / (us_twin * 1e-6) / 1e9 / HBM_PEAK_GBPS,
            "empty_kernel_us_per_launch": us_empty,
            "empty_kernel_timed_as": "replays of a captured graph launched on the current stream",
            "copy_ceiling_frac": HBM_COPY_GBPS / HBM_PEAK_GBPS,
            "reading": "kernel %.3f of peak; the same four launches per layer as pure non-temporal streams %.3f; a "
                       "float4 copy %.3f; empty kernels on the same grids, replayed as a captured graph, %.2f us per launch (the "
                       "device's dependent-boundary cost; issued eagerly the empty pass is host-bound and reads the "
                       "host's ~2.7 us launch call instead)"
                       % (achieved / HBM_PEAK_GBPS, per_launch / (us_twin * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                          HBM_COPY_GBPS / HBM_PEAK_GBPS, us_empty),
        }
    return out


def read_traffic():
    """HBM bytes per dominant-kernel launch from the latest committed PMC pass (a separate rocprofv3 --pmc run: the
    counters cannot be read inside the timed run). None if there is none."""
    pdir = os.path.join(ROOT, "profiles")
    best = None
    if os.path.isdir(pdir):
        def visit_key(name):  # r06k < r06z < r06aa < r06ab: round number, then the visit letters in spreadsheet order
            m = re.match(r"r(\d+)([a-z]*)_", name)
            return (int(m.group(1)), len(m.group(2)), m.group(2)) if m else (-1, 0, name)

        for f in sorted(os.listdir(pdir), key=visit_key):
            if f.endswith("_pmc_traffic.json"):
                best = f
    if not best:
        return None
    try:
        with open(os.path.join(pdir, best)) as fh:
            rec = json.load(fh)
        return {"bytes": float(rec["hbm_bytes_per_launch"]), "source": "profiles/" + best,
                "measured_in_this_run": False, "command": rec.get("command"),
                "method": rec.get("method", "rocprofv3 --pmc FETCH_SIZE pass of the same bench command, x2 gfx950 "
                                            "correction")}
    except Exception:
        return None


def prefill_measure(eng, cfg, n_seq, T, chunk=None, reps=2, label=""):
    """Prompt pass: n_seq x T synthetic tokens through all layers (int4 x fp16-operand MFMA GEMMs + causal attention),
    in chunks of `chunk` positions when given. tokens/s and achieved TFLOP/s against the dense fp16 MFMA peak."""
    import torch

    g = torch.Generator().manual_seed(4321)
    toks = torch.randint(0, cfg["vocab"], (n_seq, T), generator=g).cuda()
    chunk = chunk or T

    def run():
        for s0 in range(0, T, chunk):
            eng.prefill(toks[:, s0:s0 + chunk], start_pos=s0)

    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    lin = 2.0 * linear_params(cfg) * n_seq * T
    att = cfg["layers"] * n_seq * 4.0 * cfg["heads"] * cfg["head_dim"] * T * (T + 1) / 2
    return {
        "workload": "prompt pass, %d x %d tokens%s%s, fp16-operand MFMA GEMMs + causal attention, lm_head on the last "
                    "positions" % (n_seq, T, " in chunks of %d" % chunk if chunk != T else "", label),
        "tokens_per_s": n_seq * T / dt, "ms": dt * 1e3,
        "achieved_tflops": (lin + att) / dt / 1e12, "linear_tflops": lin / dt / 1e12,
        "peak_tflops": MFMA_PEAK_TFLOPS, "mfma_frac": (lin + att) / dt / 1e12 / MFMA_PEAK_TFLOPS,
    }


# ---- host-only leg: the CPU baseline ---------------------------------------------------------------------------------
def cpu_baseline(cfg, budget_s=10.0):
    """The reference's CPU path restated on this host's cores (oracle/woq_cpu_port.c, kind "port": BesTLA itself is
    not buildable here): every quantised linear of the FULL model (3.3 GB of int4 + fp32 scales, far beyond the
    caches), one token = 128 GEMV calls, for ~budget_s seconds. Beside it the torch-CPU restatement SURVEY.md §8(d)
    asked for — `x.float() @ dequant(W)` with cached fp32 weights and with on-the-fly dequantisation — on a bounded
    sample, and the reference's only published kernel number."""
    import numpy as np
    import torch

    from oracle import woq_oracle as orc

    h, i = cfg["hidden"], cfg["inter"]
    shapes = [(h, (cfg["heads"] + 2 * cfg["kv_heads"]) * cfg["head_dim"]), (cfg["heads"] * cfg["head_dim"], h),
              (h, 2 * i), (i, h)]
    cores = orc.set_threads(orc.host_threads())  # affinity mask capped by the cgroup quota, not os.cpu_count()
    rng = np.random.default_rng(0)
    t_build = time.perf_counter()
    lins = [[orc.CpuPortLinear.synthetic(k, n, 128, False, l * 4 + j) for j, (k, n) in enumerate(shapes)]
            for l in range(cfg["layers"])]
    xs = [rng.standard_normal(k).astype(np.float32) for k, _ in shapes]
    outs = [np.empty(n, np.float32) for _, n in shapes]
    t_build = time.perf_counter() - t_build

    def token():
        for layer in lins:
            for lin, x, o in zip(layer, xs, outs):
                lin(x, out=o)

    token()  # warm: thread pool, page tables
    reps, t0 = 0, time.perf_counter()
    while True:
        token()
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or reps >= 10000:
            break
    nbytes = sum(lin.nbytes for layer in lins for lin in layer)
    out = {
        "value": reps / dt, "unit": "tokens/s", "cores": cores, "visible_cpus": os.cpu_count(), "kind": "port",
        "sample": "oracle/woq_cpu_port.c (%s, OpenMP): all %d layers x 4 fused int4 g128 linears of the %s shape "
                  "(%.2f GB streamed per token, fp32 scales), %d tokens in %.1f s; lm_head / attention / norms excluded"
                  % (orc.cpu_port_isa(), cfg["layers"], cfg["name"], nbytes / 1e9, reps, dt),
        "host_gbps": nbytes * reps / dt / 1e9,
        "reference_published": "35.84 ms/token = 27.9 tokens/s: MPT-7B int4 g128 next-token latency of the "
                               "reference's own kernels on a 56-core Xeon Platinum 8480+ (docs/release_data.md:131)",
    }
    del lins
    # torch-CPU restatement (autograd/functions.py:41-63 on torch 2.x CPU ops), bounded sample
    try:
        torch.set_num_threads(cores)
        n_l = 2
        ws = [[torch.randn(k, n) * 0.02 for k, n in shapes] for _ in range(n_l)]  # cached dequantised fp32 weights
        xt = [torch.randn(1, k) for k, _ in shapes]
        for layer in ws:
            for w, x in zip(layer, xt):
                torch.matmul(x, w)
        r, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 3.0:
            for layer in ws:
                for w, x in zip(layer, xt):
                    torch.matmul(x, w)
            r += 1
        dt_c = (time.perf_counter() - t0) / r / n_l
        del ws
        k, n = shapes[1]  # on-the-fly: unpack nibbles -> (q - zp) * scale -> matmul, one o_proj-sized matrix
        packed = torch.randint(0, 256, (k // 2, n), dtype=torch.uint8)
        sc = torch.rand(k // 128, n) * 0.01
        r, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 2.0:
            q = torch.stack([packed & 15, packed >> 4], 1).reshape(k, n).to(torch.float32) - 8.0
            w = (q.reshape(k // 128, 128, n) * sc[:, None, :]).reshape(k, n)
            torch.matmul(xt[1], w)
            r += 1
        dt_f = (time.perf_counter() - t0) / r
        per_token_f = dt_f * sum(kk * nn for kk, nn in shapes) / (k * n) * cfg["layers"]
        out["torch_cpu_restatement"] = {
            "cached_dequant_fp32_tokens_per_s": 1.0 / (dt_c * cfg["layers"]),
            "on_the_fly_dequant_tokens_per_s": 1.0 / per_token_f,
            "threads": cores,
            "sample": "x.float() @ dequant(W) as torch CPU ops: cached fp32 weights over %d layers' linears (timed, "
                      "x%d), on-the-fly dequantisation of one %d x %d matrix (timed, scaled by parameter count)"
                      % (n_l, cfg["layers"] // n_l, k, n),
        }
    except Exception as ex:  # the restatement is an extra; the port above is the baseline
        out["torch_cpu_restatement"] = {"error": str(ex)[:200]}
    return out


# ---- parity of the measured model against the oracle ----------------------------------------------------------------
def oracle_of(eng, cfg, window=0):
    """oracle.LlamaOracle over the ENGINE'S OWN blobs (fused q|k|v and interleaved gate/up layouts, read by the oracle's
    own dequantiser), norm weights, embedding and lm_head. Checker only."""
    import numpy as np

    from oracle import woq_oracle as orc

    orc.set_threads(orc.host_threads())
    host = lambda t: t.detach().cpu().numpy()  # noqa: E731
    layers = []
    for l in range(eng.cfg.layers):
        lt = eng.layer_tensors[l]
        layers.append(dict(qkv=host(lt["qkv"]).view(np.uint8), o=host(lt["o"]).view(np.uint8),
                           gate_up=host(lt["gate_up"]).view(np.uint8), down=host(lt["down"]).view(np.uint8),
                           ln1=host(lt["ln1"]), ln2=host(lt["ln2"])))
    ht = eng.head_tensors
    ocfg = dict(heads=cfg["heads"], kv_heads=cfg["kv_heads"], head_dim=cfg["head_dim"], eps=float(eng.cfg.rms_eps),
                theta=float(eng.cfg.rope_theta), window=window)
    return orc.LlamaOracle(ocfg, host(ht["embed"].float()), layers, host(ht["norm"]), host(ht["lm_head"].float()))


def prefill_attention_check(cfg, n_seq, T, seqs=None):
    """The prompt pass's ATTENTION at the geometry it is measured at: a 2-layer twin of the timed
    engine — same hidden / heads / kv heads / head_dim, hence the same `attn_prefill_kernel` grid, XCD-aware work order
    and query-block count per (sequence, head); `inter` and the vocabulary cut to 512 so the CPU oracle stays cheap —
    runs the SAME n_seq x T prompt pass; layer 1's K and V rows at EVERY position (functions of layer 0's attention
    output at every query row) and the last-position logits of the first and last sequence are compared with
    oracle.LlamaOracle.forward_prompt on the engine's own blobs. Two layers because with one the logits would depend
    on a single query row. Bounds: rows 1e-2 * max|row tensor| (+ half an fp16 ulp) over ~8M elements each (a skipped or misplaced tile shows
    up at 1e-1), logits 1e-2 * max|logit| + 1e-3."""
    import numpy as np
    import torch

    t0 = time.perf_counter()
    twin = dict(cfg, inter=512, vocab=512, layers=2)
    eng = build_engine(twin, max_ctx=T, max_batch=n_seq, seed=4242)
    g = torch.Generator().manual_seed(99)
    prompts = torch.randint(0, twin["vocab"], (n_seq, T), generator=g)
    got = eng.prefill(prompts.cuda(), greedy=True).cpu().numpy().copy()
    oracle = oracle_of(eng, twin)
    res = {}
    for seq in (seqs if seqs is not None else sorted({0, n_seq - 1})):
        oracle.reset()
        ref = oracle.forward_prompt(prompts[seq].tolist())
        rows, rms = 0.0, 0.0
        for which, rc in (("k", oracle.k[1]), ("v", oracle.v[1])):
            c = eng.kv_cache(which)[seq, 1, :T].float().cpu().numpy()
            err = np.abs(c - rc)
            scale = float(np.abs(rc).max())
            if (err > 1e-2 * scale + 2.0 ** -11 * np.abs(rc)).any():
                p, hd, d = (int(v) for v in np.argwhere(err > 1e-2 * scale + 2.0 ** -11 * np.abs(rc))[0])
                raise RuntimeError("prefill attention parity FAILED: layer-1 %s row of sequence %d, position %d, head %d, "
                                   "d %d: %.3e of the tensor's maximum" % (which, seq, p, hd, d, float(err.max()) / scale))
            rows = max(rows, float(err.max()) / scale)
            rms = max(rms, float(np.sqrt((err.astype(np.float64) ** 2).mean())) / scale)
        lerr = float(np.abs(got[seq] - ref).max()) / float(np.abs(ref).max())
        if lerr > 1e-2 or int(got[seq].argmax()) != int(ref.argmax()):
            raise RuntimeError("prefill attention parity FAILED: logits of sequence %d off by %.3e of the largest" % (seq, lerr))
        res["sequence_%d" % seq] = {"layer1_kv_rows_worst_over_max": rows, "layer1_kv_rows_rms_over_max": rms,
                                    "logits_max_abs_over_max_logit": lerr}
    del eng
    free_gpu()
    return {
        "checked": "2-layer twin of the timed engine (hidden %d, %d / %d heads x %d; inter 512, vocab 512), the same %d x %d "
                   "prompt pass: layer 1's K / V rows at all %d positions x %d kv heads and the last-position logits of "
                   "sequences %s vs oracle.LlamaOracle.forward_prompt on the twin's own blobs"
                   % (cfg["hidden"], cfg["heads"], cfg["kv_heads"], cfg["head_dim"], n_seq, T, T, cfg["kv_heads"],
                      sorted(int(k.split("_")[1]) for k in res)),
        "per_sequence": res, "greedy_tokens_equal": True,
        "tol": "rows 1e-2 * max + fp16 half-ulp; logits 1e-2 * max|logit| (fp16-operand GEMMs, fp16 q/k/v between kernels)",
        "seconds": time.perf_counter() - t0,
    }


def parity_check(eng, cfg, tokens=(11, 20000, 317)):
    """Logits of the measured engine (all layers, its own synthetic weights) against oracle.LlamaOracle on the SAME
    blobs (fused q|k|v and interleaved gate/up layouts read by the oracle's own dequantiser): north_star's
    "logits max-abs" at the full BASELINE size. The oracle is the checker only."""
    import numpy as np

    from oracle import woq_oracle as orc

    t0 = time.perf_counter()
    oracle = oracle_of(eng, cfg)
    worst_abs, worst_rel, same = 0.0, 0.0, True
    for i, t in enumerate(tokens):
        eng.token.fill_(int(t))
        eng.pos.fill_(i)
        eng.step(greedy=False)
        got = eng.logits.cpu().numpy()
        ref = oracle.forward_token(int(t), i)
        err = float(np.abs(got - ref).max())
        worst_abs = max(worst_abs, err)
        worst_rel = max(worst_rel, err / float(np.abs(ref).max()))
        same = same and int(got.argmax()) == int(ref.argmax())
        tol = 2e-3 * float(np.abs(ref).max()) + 1e-4
        if err > tol:
            raise RuntimeError("parity FAILED: logits max-abs %.3e > tolerance %.3e at token %d" % (err, tol, i))
    return {
        "checked": "%s shape, all %d layers, %d decode steps vs oracle.LlamaOracle (fp32 / double accumulate) on the "
                   "same (q, scale, zp)" % (cfg["name"], eng.cfg.layers, len(tokens)),
        "logits_max_abs": worst_abs, "logits_max_abs_over_max_logit": worst_rel,
        "tol": "2e-3 * max|logit| + 1e-4 (fp16 KV cache / embedding / lm_head storage vs fp32 oracle)",
        "greedy_tokens_equal": same,
        "unpinned": "RTN rounding rule and BesTLA blob bytes are not on this path (identical (q, scale, zp) on both "
                    "sides); see DESIGN.md §5",
        "seconds": time.perf_counter() - t0,
    }


def prefill_gemm_check(eng, cfg, M, layer=0, n_rows=64, time_reps=5):
    """The prompt pass's GEMMs at the size they are measured at. For every projection of one layer: woq_linear with the
    one-product (fp16-operand, hand-scheduled ring) kernel over M rows — fp32 rows for qkv / gate-up (the pack pass),
    fp16 rows for o / down (the raw-A form) — and `n_rows` sampled rows (first / last row of the first, a middle and
    the last 128-row tile, the rest random) against oracle.woq_linear on the same blob; bound 2e-3 * rowmax|ref|.
    Also the dominant GEMM (gate/up) timed alone with HIP events on the launch stream."""
    import ctypes

    import numpy as np
    import torch

    from intel_extension_for_transformers_amd import _lib as L
    from intel_extension_for_transformers_amd import qbits
    from oracle import woq_oracle as orc

    t0 = time.perf_counter()
    orc.set_threads(orc.host_threads())
    lt = eng.layer_tensors[layer]
    rng = np.random.default_rng(77)
    rows = sorted({0, 127, (M // 2) & ~127, ((M // 2) & ~127) + 127, M - 128, M - 1}
                  | set(int(r) for r in rng.integers(0, M, n_rows)))[:max(n_rows, 6)]
    res, worst = {}, 0.0
    n_seq = max(1, M // 2048)
    g = torch.Generator().manual_seed(4321)
    eng.prefill(torch.randint(0, cfg["vocab"], (n_seq, M // n_seq), generator=g).cuda())  # rows for the timing below
    gemm_ms, call_ms = eng.time_prefill_gemm(layer, M, reps=time_reps)
    flops = 2.0 * M * cfg["hidden"] * 2 * cfg["inter"]
    timing = {"gemm": "gate/up projection of layer %d in place, M=%d K=%d N=%d: woq::gemm_f16p_kernel (ring form, SiLU*mul "
                      "epilogue)" % (layer, M, cfg["hidden"], 2 * cfg["inter"]),
              "kernel_us": gemm_ms * 1e3, "kernel_tflops": flops / (gemm_ms * 1e-3) / 1e12,
              "kernel_mfma_frac": flops / (gemm_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
              "call_us_with_pack_pass": call_ms * 1e3,
              "timed": "HIP events on the launch stream right around the GEMM kernel's launch, median of %d calls after a warm-up "
                       "one (woq_engine_time_prefill_gemm)" % time_reps}
    for name, adt in (("qkv", torch.float32), ("o", torch.float16), ("gate_up", torch.float32), ("down", torch.float16)):
        blob = lt[name]
        hdr = qbits.header_of(blob)
        hdr.compute_type = L.C_BF16  # the one-product MFMA form the prompt pass runs
        g = torch.Generator(device="cuda").manual_seed(5)
        x = torch.randn(M, hdr.K, generator=g, device="cuda", dtype=torch.float32).to(adt)
        out = torch.empty(M, hdr.N, device="cuda", dtype=torch.float32)

        def call():
            L.check(L.lib().woq_linear(x.data_ptr(), L.torch_dtype_code(x.dtype), x.stride(0), blob.data_ptr(),
                                       ctypes.byref(hdr), None, out.data_ptr(), L.torch_dtype_code(out.dtype),
                                       out.stride(0), M, L.stream_ptr()))

        call()
        torch.cuda.synchronize()
        got = out[rows].cpu().numpy()
        ref = orc.woq_linear(x[rows].float().cpu().numpy(), blob.cpu().numpy().view(np.uint8))
        rel = float((np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1)).max())
        res[name] = rel
        worst = max(worst, rel)
        if rel > 2e-3:
            raise RuntimeError("prefill parity FAILED: %s GEMM at M=%d, worst row error %.3e of its row maximum" % (name, M, rel))
        del x, out
    return {
        "checked": "layer %d's four GEMMs at M=%d through woq_linear's one-product MFMA form (fp32 rows: pack pass; "
                   "fp16 rows: raw-A), %d sampled rows each vs oracle.woq_linear on the same blob"
                   % (layer, M, len(rows)),
        "worst_row_err_over_rowmax": worst, "per_gemm": res, "tol": "2e-3 * rowmax|ref| (fp16-operand products)",
        "seconds": time.perf_counter() - t0,
    }, timing


# ---- the other BASELINE configurations (N = 1) ----------------------------------------------------------------------
def decode_entry(name, cfg, eng, steps, warmup, group, asym, ctx_note, kv_bytes_per_token=0):
    import torch

    def fence():
        torch.cuda.synchronize()

    # graph replays on the current stream (runtime/engine.py LAUNCH). Three timed regions from the same cache state, the
    # median reported: one 64-step region of an eagerly issued burst read 586 tokens/s in r04zz between two visits'
    # 786-800 for the same row (a ~28 ms stall); the headline keeps the contract's single region
    eng.capture(greedy=True)
    tok0, pos0 = eng.token.clone(), eng.pos.clone()
    regions = []
    for r in range(3):
        eng.token.copy_(tok0)
        eng.pos.copy_(pos0)
        regions.append(timed(eng.replay_graph, steps, warmup, fence, condition=(eng, None) if r == 0 else None) / steps)
        if r == 0:
            cond = dict(LAST_CONDITIONING)
    dt = sorted(regions)[1]
    wbytes = algorithmic_bytes_per_token(cfg, group=group, asym=asym)
    return {
        "config": name, "decode_tokens_per_s": 1.0 / dt, "ms_per_token": dt * 1e3,
        "ms_per_token_regions": [x * 1e3 for x in regions],
        "clock_conditioning": cond,
        "algorithmic_weight_bytes_per_token": wbytes, "hbm_gbps_weights": wbytes / dt / 1e9,
        "hbm_frac_weights": wbytes / dt / 1e9 / HBM_PEAK_GBPS,
        "kv_bytes_per_token": kv_bytes_per_token,
        "hbm_frac_weights_plus_kv": (wbytes + kv_bytes_per_token) / dt / 1e9 / HBM_PEAK_GBPS, "context": ctx_note,
    }


def extra_configs(args):
    import torch

    out = []
    # configs[2]: Llama-2-7B int4 asym group 32 (AWQ-style): batch-1 decode, then the 32 x 2048 prompt pass
    cfg = LLAMA2_7B
    eng = build_engine(cfg, group=32, sym=False, max_ctx=2048, max_batch=32)
    feed_prompt(eng, cfg["vocab"], 32)
    e = decode_entry("configs[2] Llama-2-7B int4 asym g32 fp16 scales, batch-1 decode", cfg, eng, 64, 8, 32, True,
                     "prompt 32")
    e["roofline_gemv"] = gemv_roofline(eng)
    out.append(e)
    pf = prefill_measure(eng, cfg, 32, 2048, reps=1, label=", int4 asym g32")
    pf["config"] = "configs[2] Llama-2-7B int4 asym g32, batch 32 x 2048-token prompt pass (MFMA prefill tile)"
    out.append(pf)
    del eng
    free_gpu()
    # configs[1] again, with longer contexts in the cache: the headline is quoted at <= 200 cached
    # positions; these rows say what the same model does at 512 and 2048 (sliced decode attention, merged in-launch)
    cfg = LLAMA2_7B
    eng = build_engine(cfg, group=128, sym=True, max_ctx=2304)
    g = torch.Generator().manual_seed(2)
    toks = torch.randint(0, cfg["vocab"], (2048,), generator=g).cuda()
    for ctx in (512, 2048):
        eng.prefill(toks[:ctx], start_pos=0, greedy=True)
        eng.tune_attn_for(ctx + 128)
        kvb = 2 * cfg["layers"] * cfg["kv_heads"] * cfg["head_dim"] * ctx * 2
        out.append(decode_entry("configs[1] Llama-2-7B int4 sym g128, batch-1 decode at %d cached positions" % ctx, cfg,
                                eng, 64, 8, 128, False, "%d cached positions, fp16 KV" % ctx, kv_bytes_per_token=kvb))
    del eng
    free_gpu()
    # a GPTQ act-order (desc_act) model of the same shape on the engine (round 5: the tile GEMV gathers its
    # activations by the blob's shuffle; such layers ran the fp32 VALU kernel on the module path before)
    eng = build_engine(LLAMA2_7B, group=128, sym=True, max_ctx=512, act_order=True)
    feed_prompt(eng, LLAMA2_7B["vocab"], 32)
    e = decode_entry("SURVEY 8(f)-2: Llama-2-7B int4 sym g128 GPTQ act-order (g_idx on every projection), batch-1 decode",
                     LLAMA2_7B, eng, 64, 8, 128, False, "prompt 32")
    e["kernels"] = "fp32-activation tile GEMV with the act-order gather (xq=%s, fused attention=%s)" % (
        eng.uses_xq(), eng.uses_fused_attn())
    out.append(e)
    del eng
    free_gpu()
    # SURVEY §8(f)-4, the float 4-bit weight types at the same shape (round 4: digit-plane unpack of the table codes on
    # the int8 MFMA, csrc/woq_gemv_common.h LutArgs; through round 3 these models ran the fp32 VALU kernel outside the
    # engine, 35-87 us per projection): nf4 as three planes (compute fp32) and two (compute bf16), fp4_e2m1 as one
    for wname, cname in (("nf4", "fp32"), ("nf4", "bf16"), ("fp4_e2m1", "fp32")):
        eng = build_engine(LLAMA2_7B, group=128, sym=True, max_ctx=512, weight_dtype=wname, compute_dtype=cname)
        feed_prompt(eng, LLAMA2_7B["vocab"], 32)
        out.append(decode_entry("SURVEY 8(f)-4: Llama-2-7B %s sym g128 (compute %s), batch-1 decode" % (wname, cname),
                                LLAMA2_7B, eng, 64, 8, 128, False, "prompt 32"))
        del eng
        free_gpu()
    # round 6: fp8_e4m3 WEIGHTS inside the engine (fp8 matrix-core GEMVs with RMSNorm / residual fused, csrc/woq_gemv_fp8.hip;
    # 6 launches per layer on fp32 activations). One code byte per weight: the algorithmic bytes are twice the int4 payload.
    eng = build_engine(LLAMA2_7B, group=128, sym=True, max_ctx=512, weight_dtype="fp8_e4m3")
    feed_prompt(eng, LLAMA2_7B["vocab"], 32)
    e = decode_entry("SURVEY 8(f)-4: Llama-2-7B fp8_e4m3 sym g128 (compute fp32), batch-1 decode", LLAMA2_7B, eng, 64, 8, 128,
                     False, "prompt 32")
    wb = e["algorithmic_weight_bytes_per_token"] + linear_params(LLAMA2_7B) // 2
    for k in ("hbm_gbps_weights", "hbm_frac_weights", "hbm_frac_weights_plus_kv"):
        e[k] *= wb / e["algorithmic_weight_bytes_per_token"]
    e["algorithmic_weight_bytes_per_token"] = wb
    out.append(e)
    del eng
    free_gpu()
    # configs[4]: Mistral-7B shape, int4 sym g128, fp8 (e4m3) KV cache, 8k context: chunked prompt pass then decode
    cfg, ctx = MISTRAL_7B, 8192
    eng = build_engine(cfg, group=128, sym=True, max_ctx=ctx + 256, kv_dtype=torch.float8_e4m3fn)
    pf = prefill_measure(eng, cfg, 1, ctx, chunk=2048, reps=1, label=", int4 sym g128, fp8 KV")
    pf["config"] = "configs[4] Mistral-7B int4 g128 + fp8 KV-cache, 8k-context chunked prompt pass (4 x 2048)"
    out.append(pf)
    g = torch.Generator().manual_seed(1)
    toks = torch.randint(0, cfg["vocab"], (ctx,), generator=g).cuda()
    for s0 in range(0, ctx, 2048):
        eng.prefill(toks[s0:s0 + 2048], start_pos=s0, greedy=True)
    eng.tune_attn_for(ctx + 128)
    kvb = 2 * cfg["layers"] * cfg["kv_heads"] * cfg["head_dim"] * ctx
    out.append(decode_entry("configs[4] Mistral-7B int4 g128 + fp8 KV-cache, batch-1 decode at 8k context", cfg, eng,
                            64, 8, 128, False, "8192 cached positions, fp8 e4m3 KV", kv_bytes_per_token=kvb))
    del eng
    free_gpu()
    if not args.no_70b:
        out.append(tp_point(args, LLAMA2_70B, rank=0, world=1, dist=None, steps=32, warmup=4, as_extra=True))
    return out


# ---- configs[3]: Llama-2-70B at tensor-parallel degree = world -------------------------------------------------------
def tp_point(args, cfg, rank, world, dist, steps, warmup, as_extra=False):
    import torch

    from intel_extension_for_transformers_amd.runtime.tp import TPDecoder

    n_layers = args.layers if args.layers and not as_extra else cfg["layers"]
    ranks_verified = 1  # no process group at N = 1: this process is the one rank
    if dist is not None:  # how many ranks actually answer a collective (not just the group's nominal size)
        on = "cuda" if dist.get_backend() == "nccl" else "cpu"
        ones = torch.ones(1, dtype=torch.float32, device=on)
        dist.all_reduce(ones)
        ranks_verified = int(round(float(ones.item())))
    max_ctx = 1 << max(9, (args.prompt + warmup + steps + 8).bit_length())
    eng = build_engine(cfg, group=128, sym=True, max_ctx=max_ctx, tp_rank=rank, tp=world, layers=n_layers)
    transport, comm, dec = "none (one GPU)", None, None
    if world > 1:
        from intel_extension_for_transformers_amd.runtime.comm import DeviceComm

        comm = DeviceComm(cfg["hidden"])
        if not args.host_allreduce and comm.self_test():
            transport = "device one-shot all-reduce kernels over xGMI (hipIpc inboxes), inside the captured graph"
            dec = TPDecoder(eng, cfg["vocab"], comm=comm)
        else:
            transport = "RCCL all-reduce issued by the host between sub-blocks (device exchange %s)" % (
                "disabled by flag" if args.host_allreduce else "failed its self-test: %s" % (comm.error or "mismatch"))
            comm = None
            dec = TPDecoder(eng, cfg["vocab"])

    def step(greedy):
        if dec is None:
            eng.step(greedy=greedy)
        else:
            dec.step(greedy=greedy, return_logits=False)

    feed_prompt(eng, cfg["vocab"], args.prompt, step=step)
    use_graph = not args.no_graph and (world == 1 or comm is not None)
    native = world == 1 or comm is not None  # the whole token (exchange included) is one native call
    if use_graph:
        eng.capture(greedy=True)

    def run(n):
        if use_graph:
            eng.replay_graph(n)
        elif native:
            eng.run(n)  # n tokens issued eagerly by one native call; the device exchange's kernels are part of the step
        else:
            for _ in range(n):
                step(True)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    elapsed = timed(run, steps, warmup, fence, condition=(eng, dist) if native else None)
    conditioning = dict(LAST_CONDITIONING)
    agree = True
    if dist is not None:
        on = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=on)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        toks = [torch.zeros(1, dtype=torch.int32, device=on) for _ in range(world)]
        dist.all_gather(toks, eng.token.to(on))
        agree = len({int(x.item()) for x in toks}) == 1
        if comm is not None and comm.status() != 0:
            raise RuntimeError("tensor-parallel exchange reported a timeout during the timed region")
        if not agree:
            raise RuntimeError("tensor-parallel ranks disagree on the greedy token")
    tok_s = steps / elapsed
    wbytes = algorithmic_bytes_per_token(cfg, tp=world) * n_layers // cfg["layers"]
    point = {
        "workload": "%s int4 sym group_size=128 fp16 scales, batch=1 greedy decode, prompt %d, tensor parallel x%d%s"
                    % (cfg["name"], args.prompt, world, "" if n_layers == cfg["layers"] else
                       " [REDUCED to %d layers]" % n_layers),
        "tokens_per_s": tok_s, "ms_per_token": elapsed * 1e3 / steps, "elapsed_s": elapsed,
        "algorithmic_weight_bytes_per_token_per_gpu": wbytes,
        "hbm_gbps_per_gpu": wbytes * tok_s / 1e9, "hbm_frac_per_gpu": wbytes * tok_s / 1e9 / HBM_PEAK_GBPS,
        "clock_conditioning": conditioning,
        "process_group_ranks": world, "process_group_backend": dist.get_backend() if dist is not None else None,
        "rccl_ranks_verified": ranks_verified, "allreduces_per_token": 2 * n_layers if world > 1 else 0,
        "token_exchanges_per_token": 1 if world > 1 else 0, "allreduce_bytes": cfg["hidden"] * 4,
        "allreduce_transport": transport, "hipgraph": use_graph, "ranks_agree_on_token": agree,
    }
    if rank == 0:
        point["roofline_gemv"] = gemv_roofline(eng)
    if as_extra:
        point = dict(config="configs[3] single-GPU point: " + point.pop("workload"), **point)
    del eng, dec, comm
    free_gpu()
    return point


def spawn_command(gpus, env, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): the command that runs the
    same arguments as N ranks of one node under torch.distributed.run — what the driver's own N > 1 command line is
    (one rank per GPU over RCCL, rendezvous on 127.0.0.1). None when there is nothing to spawn."""
    if gpus <= 1 or "WORLD_SIZE" in env:
        return None
    port = env.get("MASTER_PORT") or str(29500 + os.getpid() % 2000)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--prompt", type=int, default=32, help="positions already in the KV cache when timing starts")
    ap.add_argument("--layers", type=int, default=0, help="override the layer count (marks the line REDUCED)")
    ap.add_argument("--workload", choices=["auto", "7b", "70b"], default="auto")
    ap.add_argument("--prefill-seqs", type=int, default=4, help="sequences in the main line's prompt-pass object (0 = skip)")
    ap.add_argument("--prefill-len", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip extra_configs (configs[2], [3] single GPU, [4])")
    ap.add_argument("--no-70b", action="store_true", help="skip the 70B single-GPU point of extra_configs")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-structures", action="store_true", help="(no-op since round 6: the persistent launch left the library)")
    ap.add_argument("--graph", action="store_true", help="(the default: replays of the captured hipGraph on the current stream)")
    ap.add_argument("--no-graph", "--eager", dest="no_graph", action="store_true",
                    help="time eager bursts (one native call issuing every launch) instead of graph replays; same device "
                         "time (profiles/r04ab_stream_mode_probe.txt), but the host has to keep ahead of the device")
    ap.add_argument("--condition-ms", type=float, default=800.0,
                    help="untimed replays before the warm-up steps so that sclk has left its idle ramp (0 = off)")
    ap.add_argument("--host-allreduce", action="store_true", help="N > 1: force the RCCL host-driven transport")
    ap.add_argument("--counters-run", action="store_true",
                    help="the headline path alone for a `rocprofv3 --pmc` pass: the same 32-layer engine, prompt, warm-up and "
                         "timed decode steps issued eagerly (no graph capture: rocprofv3's counter tool crashed on the captured "
                         "step in round 5), nothing else measured; prints a short line")
    args = ap.parse_args()
    global CONDITION_MS
    CONDITION_MS = args.condition_ms
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("WOQ_BENCH_BACKEND", "nccl") == "nccl":
        import torch

        seen = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if seen < args.gpus:  # before any rank is spawned or any device is touched
            raise SystemExit("bench.py: --gpus %d but only %d GPU(s) are visible on this node; one rank per GPU over RCCL "
                             "needs %d (the N > 1 line is configs[3], Llama-2-70B at tensor-parallel degree N: unmeasured "
                             "on hardware until a node runs it)" % (args.gpus, seen, args.gpus))
    respawn = spawn_command(args.gpus, os.environ, sys.argv[1:])
    if respawn is not None:  # plain `python bench.py --gpus N`, N > 1: become torch.distributed.run's N ranks
        sys.stdout.flush()
        os.execvpe(respawn[0], respawn, dict(os.environ, MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1")))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); the line's n_gpus would "
                         "not be what was asked for" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if world > 1 and os.environ.get("WOQ_BENCH_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) are visible on this node (one rank per GPU over RCCL)"
                         % (world, torch.cuda.device_count()))
    # WOQ_BENCH_BACKEND=gloo (development): the N > 1 code path with several ranks on ONE GPU — RCCL refuses that, the
    # device exchange does not need it (tests/test_gpu_tp_device.py). Never what the driver runs.
    backend = os.environ.get("WOQ_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    workload = args.workload
    if workload == "auto":
        workload = "70b" if world > 1 or torch.cuda.device_count() > 1 else "7b"
    if world > 1 and workload == "7b":
        raise SystemExit("the 7B batch-1 path does not shard (SURVEY.md §8(e)); N > 1 measures configs[3] (--workload 70b)")

    base = {"metric": METRIC, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "vs_baseline": None, "dtype": DTYPE}
    if workload == "70b":
        p = tp_point(args, LLAMA2_70B, rank, world, dist, args.steps, args.warmup)
        if rank == 0:
            out = dict(base, value=p["tokens_per_s"], ms_per_step=p["ms_per_token"], scaling="strong",
                       data="synthetic (random-init int4 weights of the Llama-2-70B shape, random prompt ids)",
                       config={"workload": p["workload"], "global_batch": 1, "parallelism": "tp%d" % world,
                               "process_group_ranks": world, "rccl_ranks_verified": p["rccl_ranks_verified"],
                               "allreduces_per_token": p["allreduces_per_token"],
                               "token_exchanges_per_token": p["token_exchanges_per_token"],
                               "allreduce_bytes": p["allreduce_bytes"], "allreduce_transport": p["allreduce_transport"],
                               "hipgraph": p["hipgraph"], "visible_gpus": torch.cuda.device_count()},
                       hbm_gbps_per_gpu=p["hbm_gbps_per_gpu"], hbm_frac_of_peak_end_to_end=p["hbm_frac_per_gpu"],
                       roofline=dict(p["roofline_gemv"], traffic=None))
            out["note"] = ("`value` is Llama-2-70B at tensor-parallel degree %d: the strong-scaling curve of BASELINE "
                           "configs[3] (the 7B batch-1 path of the metric's name does not shard, SURVEY.md §8(e))" % world)
            if world == 1:
                out["note"] += ("; this is the curve's N = 1 base on a multi-GPU node — the headline configuration "
                                "(configs[1], Llama-2-7B, what BENCH_rNN.json's `value` is) rides along as `headline_7b`")
            if world == 1 and not args.no_extra:  # first point of a sweep on a multi-GPU node: the 7B headline rides along
                eng = build_engine(LLAMA2_7B, max_ctx=512)
                feed_prompt(eng, LLAMA2_7B["vocab"], args.prompt)
                e7 = decode_entry("configs[1] Llama-2-7B int4 sym g128, batch-1 decode", LLAMA2_7B, eng, args.steps,
                                  args.warmup, 128, False, "prompt %d" % args.prompt)
                out["headline_7b"] = {"value": e7["decode_tokens_per_s"], "unit": "tokens/s",
                                      "ms_per_step": e7["ms_per_token"], "steps": args.steps, "warmup": args.warmup,
                                      "workload": "Llama-2-7B int4 sym group_size=128 fp16 scales, batch=1 greedy "
                                                  "decode, prompt %d (same as the one-GPU run's `value`)" % args.prompt,
                                      "hbm_frac_of_peak_end_to_end": e7["hbm_frac_weights"],
                                      "roofline": gemv_roofline(eng, read_traffic())}
                del eng
                free_gpu()
            emit(out)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- configs[1]: Llama-2-7B int4 sym g128, one GPU --------------------------------------------------------------
    cfg = dict(LLAMA2_7B, layers=args.layers or LLAMA2_7B["layers"])
    cpu = None
    if not args.no_cpu_baseline and not args.counters_run:  # host-only leg first: the rest of the run keeps the GPU busy
        cpu = cpu_baseline(LLAMA2_7B)
    max_ctx = 1 << max(9, (args.prompt + args.warmup + max(args.steps, 128) + 8).bit_length())
    want_prefill = args.prefill_seqs > 0 and not args.counters_run
    if want_prefill:
        max_ctx = max(max_ctx, args.prefill_len)
    eng = build_engine(cfg, max_ctx=max_ctx, max_batch=max(1, args.prefill_seqs) if want_prefill else 1,
                       layers=cfg["layers"])
    feed_prompt(eng, cfg["vocab"], args.prompt)
    use_graph = not args.no_graph and not args.counters_run
    if use_graph:
        eng.capture(greedy=True)
    run = eng.replay_graph if use_graph else eng.run  # eng.run(n): n steps issued eagerly by one native call
    if args.counters_run:
        run(args.warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"counters_run": True, "steps": args.steps, "warmup": args.warmup, "prompt": args.prompt,
                          "tokens_per_s_under_the_profiler": args.steps / dt,
                          "gemv_launches": (args.prompt + args.steps + args.warmup) * cfg["layers"] * 4}), flush=True)
        return

    elapsed = timed(run, args.steps, args.warmup, torch.cuda.synchronize, condition=(eng, None))
    timed_region = list(LAST_TIMED_REGION)
    conditioning = dict(LAST_CONDITIONING)
    tok_s = args.steps / elapsed
    value_128 = tok_s
    if args.steps != 128 and max_ctx >= args.prompt + args.warmup + 128 + 8:
        # the driver runs 20 steps; the 128-step value of the same engine rides along (contexts grow to ~180 positions:
        # the one-workgroup-per-head attention of the fused launch is context-linear)
        feed_prompt(eng, cfg["vocab"], args.prompt)
        value_128 = 128 / timed(run, 128, args.warmup, torch.cuda.synchronize, condition=None)
    # the other way of issuing the same steps, same engine, same prompt, right after the headline
    feed_prompt(eng, cfg["vocab"], args.prompt)
    if not use_graph:
        eng.capture(greedy=True)
    other = eng.run if use_graph else eng.replay_graph
    el2 = timed(other, args.steps, args.warmup, torch.cuda.synchronize, condition=(eng, None))
    feed_prompt(eng, cfg["vocab"], args.prompt)
    eng.capture(greedy=True)
    el3 = timed(eng.replay_graph_round3, args.steps, args.warmup, torch.cuda.synchronize, condition=(eng, None))
    launch_modes = {"eager_bursts_tokens_per_s": args.steps / (el2 if use_graph else elapsed),
                    "hipgraph_replay_tokens_per_s": args.steps / (elapsed if use_graph else el2),
                    "hipgraph_replay_behind_cross_stream_waits_tokens_per_s": args.steps / el3,
                    "headline": "hipgraph_replay" if use_graph else "eager_bursts",
                    "note": "the same kernels and device-side token chain either way, and the same device time since the "
                            "graph is launched on the caller's stream: through round 3 it was replayed on the engine's "
                            "capture stream behind cross-stream event waits, which cost every kernel boundary ~1 us "
                            "(1.17 vs 1.04 ms per token; profiles/r04g_graph_vs_eager_steps.txt found the gap, "
                            "profiles/r04ab_stream_mode_probe.txt its cause). Eager bursts cost ~0.34 ms of host launch "
                            "calls per token, issued ahead of the device; graph replays none"}
    qbytes = algorithmic_bytes_per_token(cfg)
    out = dict(base, value=tok_s, ms_per_step=elapsed * 1e3 / args.steps, scaling="weak",
               data="synthetic (random-init int4 weights of the Llama-2-7B shape, random prompt ids)",
               config={"workload": "Llama-2-7B int4 sym group_size=128 fp16 scales, batch=1 greedy decode, prompt %d, "
                                   "lm_head fp16 unquantised, KV fp16%s"
                                   % (args.prompt, "" if cfg["layers"] == 32 else " [REDUCED to %d layers]" % cfg["layers"]),
                       "global_batch": 1, "parallelism": "single GPU", "hipgraph": use_graph},
               value_128_steps=value_128,
               timed_region_unix=timed_region, clock_conditioning=conditioning, launch_modes=launch_modes,
               hbm_gbps_quantized_weight_stream=qbytes * tok_s / 1e9,
               hbm_frac_of_peak_end_to_end=qbytes * tok_s / 1e9 / HBM_PEAK_GBPS,
               hbm_frac_of_measured_copy_ceiling_end_to_end=qbytes * tok_s / 1e9 / HBM_COPY_GBPS,
               roofline=gemv_roofline(eng, read_traffic()))
    if want_prefill:
        out["prefill"] = prefill_measure(eng, cfg, args.prefill_seqs, args.prefill_len, label=", same int4 g128 weights")
    if not args.no_parity:
        out["parity"] = {"decode": parity_check(eng, cfg)}
        if want_prefill:
            pp, timing = prefill_gemm_check(eng, cfg, args.prefill_seqs * args.prefill_len)
            out["parity"]["prefill"] = pp
            out["prefill"]["dominant_gemm"] = timing
        out["fused_attention_launch"] = {"in_use": eng.uses_fused_attn(), "engine_status": eng.status()}
    del eng
    free_gpu()
    if not args.no_parity and want_prefill:
        out["parity"]["prefill"]["attention"] = prefill_attention_check(cfg, args.prefill_seqs, args.prefill_len)
    if not args.no_extra:
        out["extra_configs"] = extra_configs(args)
    if cpu is not None:
        out["cpu_baseline"] = cpu
    out["configs_summary"] = configs_summary(out)
    emit(out)


if __name__ == "__main__":
    main()
