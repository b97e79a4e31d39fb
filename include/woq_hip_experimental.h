/*
 * woq_hip_experimental.h — measurement hooks and lab switches of libwoq_hip.so.
 *
 * NOT part of the drop-in boundary (include/woq_hip.h) and NOT covered by WOQ_ABI_VERSION: these entry points exist
 * for bench.py's `roofline` / `prefill.dominant_gemm` objects, the A/B records under profiles/ and the GPU tests that
 * pin an experiment's results. They may change or disappear between rounds. No reference counterpart (the reference
 * has no timing hooks on qbits.cpp:113-140's path).
 */
#ifndef WOQ_HIP_EXPERIMENTAL_H_
#define WOQ_HIP_EXPERIMENTAL_H_

#include "woq_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* grouped form only: `chunk` > 0 (a multiple of 32) = position-independent slice geometry — slice s owns the absolute
 * cached positions [s * chunk, (s + 1) * chunk) (the last slice also whatever lies beyond), so its K / V rows are
 * requested before the device-side position is read; 0 = slices cut evenly from the current span (the default; always
 * used with a sliding window). Pick splits >= ceil(positions / chunk). Same capture rule as attn_splits. */
WOQ_API int woq_engine_set_attn_chunk(woq_engine* e, int chunk);
WOQ_API int woq_engine_attn_chunk(woq_engine* e);
/* time the dominant kernel (int4 GEMV) alone over all layers with HIP events on `stream`: one pass = every layer's 4
 * GEMV launches in the forms the decode step uses (same kernels, epilogues, XQ outputs, residual chaining), captured
 * into a hipGraph and replayed `reps` times between one event pair (after an untimed replay); returns total ms, the
 * algorithmic bytes of one pass and its launch count (average launch duration = total_ms / (reps * launches_per_pass),
 * boundaries included). Overwrites the residual stream / XQ vectors (the next step's embedding rewrites them). */
WOQ_API int woq_engine_time_gemv(woq_engine* e, int reps, void* stream, float* total_ms, double* bytes_per_pass,
                                 int* launches_per_pass);
/* the same with a pass restricted to some of the layer's projections (mask bit 0 qkv, 1 o, 2 gate/up, 3 down): the
 * per-instantiation numbers rocprofv3's kernel stats list separately (bench.py roofline.by_projection) */
WOQ_API int woq_engine_time_gemv_mask(woq_engine* e, int mask, int reps, void* stream, float* total_ms,
                                      double* bytes_per_pass, int* launches_per_pass);
/* the same four launches per layer with the arithmetic taken out, timed the same way (`reps` passes after a warm-up
 * pass, total milliseconds): mode 0 = load-only twins (same grids, waves, K slices, non-temporal 16-byte requests over
 * the engine's own blobs: what this launch structure reaches as a pure stream), mode 1 = empty kernels on the same
 * grids (what the launches cost before they do anything). bench.py reports both as roofline.ceiling. */
WOQ_API int woq_engine_time_twin(woq_engine* e, int mode, int reps, void* stream, float* total_ms);
/* how woq_engine_time_gemv / _gemv_mask / _twin issue their timed passes: on != 0 (default) eagerly back to back on
 * the stream — the way decode bursts run by default since round 4 — else as replays of a captured graph. */
WOQ_API int woq_engine_set_time_eager(woq_engine* e, int on);
/* the prompt pass's dominant GEMM in place: the engine's own gate/up call of `layer` over n_rows rows of the residual
 * stream a preceding woq_engine_prefill left (RMSNorm pack pass + MFMA GEMM + SiLU * mul epilogue), the MEDIAN of `reps`
 * calls after a warm-up one: gemm_ms = the GEMM kernel alone (HIP events on the launch stream right around its launch),
 * call_ms = pack pass + GEMM. */
WOQ_API int woq_engine_time_prefill_gemm(woq_engine* e, int layer, int n_rows, int reps, void* stream, float* gemm_ms,
                                         float* call_ms);

#ifdef __cplusplus
}
#endif
#endif /* WOQ_HIP_EXPERIMENTAL_H_ */
