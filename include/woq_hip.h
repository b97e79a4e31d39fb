/*
 * woq_hip.h — C ABI of libwoq_hip.so, the MI355X (gfx950) replacement for the reference's
 * `qbits_py` operator module on the int4 weight-only-quantized linear path.
 *
 * Every entry point below replaces one function of the reference boundary
 *   intel_extension_for_transformers/qbits/qbits.cpp:192-206   (PYBIND11_MODULE qbits_py)
 * or, for the ops between the linears, one stock-HF module forward that the reference executes
 * on the CPU. The reference-side binding a maintainer would add is shown in
 * INTEGRATION.md.
 *
 * Conventions
 *  - plain C types only; every `*_dev` / `const void*` tensor argument is a DEVICE pointer owned by
 *    the caller (torch allocates it); the library never frees or keeps caller memory, except the
 *    pointers stored in a woq_engine, which must outlive it (same rule as the reference's
 *    set_woq_workspace, bestla_weightonly_dispatcher.cpp:394-397).
 *  - `stream` is a hipStream_t passed as void* (0 = default stream). All work is asynchronous on
 *    that stream. The per-token calls (woq_linear with int4 blobs, fp32 activations and M <= 8, the
 *    woq_engine step) neither synchronise nor allocate and can be captured into a hipGraph; the
 *    prefill GEMM (M > 8), int8 blobs and 16-bit activations at M <= 8 need scratch: it comes from the
 *    caller's workspace (woq_set_workspace, the reference's set_woq_workspace contract) when that has
 *    room — then these calls do not allocate either and are capturable — and otherwise from
 *    stream-ordered allocation (hipMallocAsync / hipFreeAsync on `stream`) per call, like the reference's
 *    per-call amalloc (bestla_weightonly_dispatcher.cpp:108-118,179); the engine's prompt pass owns its scratch.
 *  - return value: 0 on success, non-zero on error; woq_last_error() returns a thread-local
 *    message that starts with "QBits:" like the reference's TORCH_CHECK strings
 *    (bestla_weightonly_dispatcher.cpp:289,368; qbits.cpp:35,150). The Python shim raises
 *    RuntimeError with it, which is what c10::Error becomes in the reference.
 *  - dtype codes: enum woq_dtype in woq_blob.h (0 fp32, 1 bf16, 2 fp16).
 */
#ifndef WOQ_HIP_H_
#define WOQ_HIP_H_

#include <stddef.h>
#include <stdint.h>

#include "woq_blob.h"

#ifdef __cplusplus
extern "C" {
#endif

#define WOQ_API __attribute__((visibility("default")))

WOQ_API const char* woq_last_error(void);
/* ABI revision of this header: 1 = rounds 1-3; 2 = round 4; 3 = round 5; 4 = round 6: FROZEN to what the boundary
 * needs (the qbits operator functions, the ops between the linears, the decode engine, the tensor-
 * parallel exchange). Every lab switch and measurement hook left it — the persistent launch and the token-long
 * prefetcher left the library altogether (tools/rejected/, both measured slower), the timing / probe entry points
 * live in woq_hip_experimental.h, which is NOT covered by this number. A client checks woq_abi_version() against
 * WOQ_ABI_VERSION at load time. */
#define WOQ_ABI_VERSION 4
WOQ_API int woq_abi_version(void);
/* number of visible HIP devices (0 when there is no GPU); never fails. */
WOQ_API int woq_device_count(void);

/* How the decode GEMVs read a 4-bit table weight type (nf4 / fp4_e2m1 / fp4_e2m1_bnb; no counterpart in the reference,
 * whose BesTLA kernels look the value up per weight): table[c] * S as `return value` balanced base-256 int8 digits —
 * planes[j][c >> 2] byte (c & 3) = digit j of code c — one v_mfma_i32_16x16x64_i8 per digit plane, and *wmul = 16 / S,
 * the factor that undoes S (and the 16 of the int4 kernels' 16 q operand). compute_type (enum woq_compute_type): nf4
 * takes three planes at fp32, two otherwise. Host-only (no device needed); 0 = not a table type. */
WOQ_API int woq_table_digit_planes(int weight_type, int compute_type, uint32_t planes[3][4], float* wmul);

/* ---- packed-weight management (load time) ---------------------------------------------------- */

/* replaces qbits.get_packed_weight_size (qbits.cpp:79-88). 0 = unsupported geometry. */
WOQ_API size_t woq_packed_weight_size(int K, int N, int blocksize, int weight_type, int scale_type, int asym,
                                      int act_shuffle);

/* replaces qbits.repack_quantized_weight (qbits.cpp:61-77 -> bestla_packq_impl.cpp:20-41).
 * weight_type (enum woq_weight_type): int4_clip — qweight int8 [K,N] holds int4 values in the signed domain
 * (modules.py:225-227); int3_clip / int2_clip — values in [-4, 3] / [-2, 1], kept in int4 storage with a header tag
 * (woq_blob.h narrow_bits: same bytes and kernels as int4; the name survives acquire_packed_weight_info);
 * int8 — full int8 values (stored as a composite of two int4 blobs, woq_blob.h); nf4 /
 * fp4_e2m1 / fp4_e2m1_bnb — table codes 0..15, no zero points; fp8_e4m3 / fp8_e5m2 — code bytes (as int8), no zero
 * points, scale_type may be WOQ_SCALE_FP8_E8M0 (power-of-two scales, stored as bf16). scale fp32 [G,N],
 * zp int8 [G,N] or NULL (sym), g_idx int32 [K] = GPTQ act-order group id of every K row (each group exactly
 * `blocksize` rows) or NULL; like BesTLA, repack converts it to activation shuffle indices
 * (qbits_ut/test_packq.py:22-28,59-64) and the blob keeps those. blob_dev: caller-allocated,
 * woq_packed_weight_size() bytes, 256-B aligned. Layout transform only; device-side. */
WOQ_API int woq_repack_quantized_weight(const int8_t* qweight_dev, const float* scale_dev, const int8_t* zp_dev,
                                        const int32_t* g_idx_dev, int K, int N, int blocksize, int weight_type,
                                        int scale_type, int compute_type, void* blob_dev, size_t blob_bytes,
                                        void* stream);

/* replaces qbits.quantize_to_packed_weight (qbits.cpp:90-100): RTN of an fp32 weight straight into a
 * blob. transpose != 0: weight is [N,K] (nn.Linear layout). Rounding rule: DESIGN.md (parity unpinned). */
WOQ_API int woq_quantize_to_packed_weight(const float* weight_dev, int transpose, int K, int N, int blocksize,
                                          int weight_type, int scale_type, int compute_type, int asym,
                                          void* blob_dev, size_t blob_bytes, void* stream);

/* replaces qbits.dequantize_packed_weight (qbits.cpp:102-111): blob -> fp32 [K,N] or [N,K]. The
 * geometry is passed by value (hdr = host copy of the blob header) so the call never reads device
 * memory from the host. */
WOQ_API int woq_dequantize_packed_weight(const void* blob_dev, const woq_blob_header* hdr, float* out_dev,
                                         int transpose, void* stream);

/* copy the 256-B header of a device blob to host memory (synchronises `stream`). Load-time only;
 * the Python module caches the result so the hot path never re-parses it (the reference re-parses
 * per call, bestla_weightonly_dispatcher.cpp:335). */
WOQ_API int woq_read_header(const void* blob_dev, woq_blob_header* hdr_out, void* stream);

/* replaces qbits.acquire_packed_weight_info for the tensor-valued selectors
 * (bestla_packq_impl.cpp:152-204): SCALE_TENSOR -> fp32 [G,N], ZP_TENSOR -> int8 [G,N] (signed
 * domain), G_IDX -> int32 [K]. Scalar / string selectors are answered from the cached header in
 * the Python shim. out_dev is caller-allocated. */
WOQ_API int woq_blob_extract(const void* blob_dev, const woq_blob_header* hdr, int what, void* out_dev,
                             void* stream);

/* replaces qbits.set_woq_workspace (qbits.cpp:142-144 -> bestla_weightonly_dispatcher.cpp:394-397): a raw pointer
 * into caller-owned device memory that must outlive every later call; woq_linear carves its scratch from it (int8
 * composite: M * N * 4 bytes; M > 8: the packed activation planes, about 2 * Mpad * Kpad * (1 or 2 planes) +
 * 8 * Mpad + 4 * Npad bytes; fp16 / bf16 activation rows at M <= 8 are read natively and need none). Process-wide like the reference's (calls from
 * several host threads are not safe with it). NULL / 0 removes it. */
WOQ_API int woq_set_workspace(void* workspace_dev, size_t bytes);

/* ---- the hot call ------------------------------------------------------------------------------ */

/* replaces qbits.woq_linear (qbits.cpp:113-140): out[M,N] = act[M,K] . W_deq[K,N] (+ bias).
 *   act: act_dtype [M, lda]; out: out_dtype [M, ldo], written in place; bias fp32 [N] or NULL
 *   (alpha = 1, beta = bias ? 1 : 0, bestla_customop.hpp:22-40). Activation shuffle (g_idx) is
 *   applied when the blob carries indices (autograd/functions.py:52-57).
 * Kernel selection by M: M <= 8 decode GEMV (int8-MFMA inner product over exact fixed-point activation limbs,
 * csrc/woq_gemv_i8.hip; the generic fp32-VALU kernel for shuffled / table / fp8 blobs), M > 8 MFMA GEMM
 * (csrc/woq_gemm_f16.hip; its K loop for the one-product form is scheduled by hand, csrc/woq_gemm_f16p.h). */
WOQ_API int woq_linear(const void* act_dev, int act_dtype, int lda, const void* blob_dev,
                       const woq_blob_header* hdr, const float* bias_dev, void* out_dev, int out_dtype, int ldo,
                       int M, void* stream);

/* ---- ops between the linears (replace stock-HF module forwards) -------------- */

/* HF LlamaRMSNorm: out = x * rsqrt(mean(x^2) + eps) * weight ; x/out dtype [rows, d], weight fp32 [d]. */
WOQ_API int woq_rmsnorm(const void* x_dev, int dtype, const float* weight_dev, float eps, int rows, int d,
                        void* out_dev, int out_dtype, void* stream);
/* HF apply_rotary_pos_emb (rotate_half) in place on x [tokens, heads, D]; cos/sin fp32 tables
 * [max_pos, D/2]; pos int32 [tokens] (device). */
WOQ_API int woq_rope(void* x_dev, int dtype, const int32_t* pos_dev, const float* cos_dev, const float* sin_dev,
                     int tokens, int heads, int D, void* stream);
/* HF LlamaMLP elementwise part: out = silu(gate) * up, n elements. */
WOQ_API int woq_silu_mul(const void* gate_dev, const void* up_dev, int dtype, size_t n, void* out_dev, void* stream);
/* GeLU: approximate != 0 -> tanh form ("gelu_new", GPT-2), else erf form. */
WOQ_API int woq_gelu(const void* x_dev, int dtype, size_t n, int approximate, void* out_dev, void* stream);

/* ---- fused batch-1 decode engine (Llama-class decoder; the measured path of bench.py) ----------
 * A woq_engine owns no weights: it holds device pointers to the per-layer WQH1 blobs and norm
 * vectors plus its own activation scratch and KV cache, and launches the whole token step
 * (5 kernels per layer) from C++, optionally replayed as one hipGraph. */
typedef struct woq_engine woq_engine;

typedef struct woq_engine_config {
  int32_t hidden, inter, heads, kv_heads, head_dim, layers, vocab, max_ctx;
  float rms_eps, rope_theta;
  int32_t tp_rank, tp_size; /* tensor parallel: heads/inter are PER-RANK sizes when tp_size > 1 */
  int32_t kv_dtype;         /* WOQ_F16 | WOQ_BF16 | WOQ_FP8_E4M3 (unscaled e4m3fn, saturating at +-448) */
  int32_t reserved[3];      /* [0] = max_batch: sequences the KV cache holds for woq_engine_prefill (0 / 1 = one);
                             * [1] = attn_splits: context slices per head in the decode attention (0 = automatic:
                             *       1 up to max_ctx 4096, else 1024 / heads clamped to [2, 32]);
                             * [2] = sliding window (HF Mistral `sliding_window`): a query sees the last `window`
                             *       positions, itself included; 0 = full causal attention */
} woq_engine_config;

typedef struct woq_layer_weights {
  const void* qkv_blob;     /* [hidden -> (heads + 2 kv_heads) * head_dim] */
  const void* o_blob;       /* [heads*head_dim -> hidden] */
  const void* gate_up_blob; /* [hidden -> 2*inter], 16-column tiles interleaved gate/up */
  const void* down_blob;    /* [inter -> hidden] */
  const float* ln1;         /* input_layernorm weight fp32 [hidden] */
  const float* ln2;         /* post_attention_layernorm weight fp32 [hidden] */
  woq_blob_header qkv_hdr, o_hdr, gate_up_hdr, down_hdr;
} woq_layer_weights;

WOQ_API int woq_engine_create(const woq_engine_config* cfg, woq_engine** out);
WOQ_API void woq_engine_destroy(woq_engine* e);
WOQ_API int woq_engine_set_layer(woq_engine* e, int layer, const woq_layer_weights* w);
/* embed: fp16/bf16 [vocab, hidden]; final norm fp32 [hidden]; lm_head fp16/bf16 [vocab_shard, hidden]
 * (NOT quantised, like the reference: utils/config.py:836-837 llm_int8_skip_modules). */
WOQ_API int woq_engine_set_head(woq_engine* e, const void* embed_dev, int embed_dtype, const float* final_norm_dev,
                                const void* lm_head_dev, int lm_head_dtype, const float* cos_dev,
                                const float* sin_dev);
/* optional: make the engine use caller-owned device buffers (e.g. torch tensors) for its I/O instead of
 * its own: token int32[1], pos int32[1], logits fp32[vocab], hidden fp32[hidden]. NULL keeps the internal one. */
WOQ_API int woq_engine_bind_io(woq_engine* e, void* token_dev, void* pos_dev, void* logits_dev, void* hidden_dev);
/* device buffers the caller reads/writes between steps */
WOQ_API void* woq_engine_token_ptr(woq_engine* e);  /* int32[1]: token fed to the next step */
/* int32[max_ctx + 1]: a greedy step that fed position p leaves its new token in slot p, so a host that replays k
 * steps back to back reads k tokens with ONE synchronisation (the reference's loop pays a host round trip per token,
 * transformers/generation: greedy_search.py:148-150,374-377) */
WOQ_API void* woq_engine_token_log_ptr(woq_engine* e);
WOQ_API void* woq_engine_pos_ptr(woq_engine* e);    /* int32[1]: its position */
WOQ_API void* woq_engine_logits_ptr(woq_engine* e); /* fp32[vocab] of the last step */
WOQ_API void* woq_engine_hidden_ptr(woq_engine* e); /* fp32[hidden] residual stream (TP all-reduce target) */
/* one decode step: embed(token) -> layers -> final norm -> lm_head -> logits; if greedy != 0 also
 * argmax -> token_ptr and pos_ptr += 1, so steps can be chained with no host round trip. */
WOQ_API int woq_engine_step(woq_engine* e, int greedy, void* stream);
/* prompt pass (what HF's first generate() forward does with the reference's linears at M = batch * T rows,
 * nn/modules.py:140-169): n_seq sequences x T new tokens each, tokens int32 [n_seq][T] on the device, at positions
 * start_pos .. start_pos + T - 1 (each sequence's cache must already hold [0, start_pos): chunked prefill = calls
 * with growing start_pos). Fills the KV caches; the last position's logits of every sequence go to
 * woq_engine_prefill_logits_ptr() (fp32 [n_seq][vocab]); sequence 0's also to logits_ptr / hidden_ptr, with
 * pos_ptr = start_pos + T - 1, and if greedy != 0 token_ptr = argmax and pos_ptr = start_pos + T so that
 * woq_engine_step continues sequence 0. Linears run on the fp16-operand MFMA GEMM whatever the blobs' compute_type;
 * q/k/v, attention output and MLP activations are fp16 between kernels, the residual stream fp32. */
WOQ_API int woq_engine_prefill(woq_engine* e, const int32_t* tokens_dev, int n_seq, int T, int start_pos, int greedy,
                               void* stream);
WOQ_API void* woq_engine_prefill_logits_ptr(woq_engine* e);
/* decode attention regime: 1 = one workgroup per head (fastest up to a few hundred cached positions), n > 1 = n
 * context slices per head + a combine launch (long contexts). Takes effect at the next step / capture; a graph
 * captured earlier keeps the regime it was captured with. */
WOQ_API int woq_engine_set_attn_splits(woq_engine* e, int splits);
WOQ_API int woq_engine_attn_splits(woq_engine* e);
/* sliced regime only: on = one workgroup per (kv head, slice) computing all the query heads of the group on the
 * matrix cores, so each K / V row is fetched once per group instead of once per query head (measured faster at
 * 8k cached positions, Mistral-7B shape; applies to head_dim 128 with 2 / 4 / 8 query heads per kv head and an fp16 or
 * fp8 cache — silently the per-query-head slices otherwise). Default off. Same capture rule as attn_splits. */
WOQ_API int woq_engine_set_attn_grouped(woq_engine* e, int on);
WOQ_API int woq_engine_attn_grouped(woq_engine* e);
/* decode step, XQ path: [RMSNorm + qkv GEMV] and [RoPE + KV append + attention] as ONE launch — the workgroup whose
 * column strip completes a head's q / k / v runs that head's attention (csrc/woq_gemv_attn.hip). Applies to multi-head
 * shapes (heads == kv_heads), head_dim 128, hidden 4096, one context slice, no sliding window; the two launches
 * otherwise. Default on (WOQ_ENGINE_FUSE_ATTN=0 turns it off). Same capture rule as attn_splits.
 * woq_engine_fuse_attn: 1 when the next step / capture will use it. */
WOQ_API int woq_engine_set_fuse_attn(woq_engine* e, int on);
WOQ_API int woq_engine_fuse_attn(woq_engine* e);
/* sticky device-side status of the decode step, 0 = fine; bit 0: an attention workgroup of the fused launch gave up
 * waiting for its head's q / k / v after its bound (its outputs are then wrong); bit 1: a step started with its
 * position at or beyond max_ctx — it ran at max_ctx - 1 instead (outputs meaningless, nothing written out of bounds;
 * woq_engine_step / _replay cannot check a position that lives on the device); bit 2: a greedy step found no
 * winning logit (all NaN) and fed token 0; -1 = the read itself failed.
 * Synchronises `stream`. */
WOQ_API int woq_engine_status(woq_engine* e, void* stream);
/* resets the sticky status to 0 (stream-ordered): a caller that read a non-zero status, changed what caused it
 * (e.g. woq_engine_set_fuse_attn(e, 0)) and re-runs the request starts from a clean word. */
WOQ_API int woq_engine_clear_status(woq_engine* e, void* stream);
/* KV cache base pointers (which: 0 = K, 1 = V), layout [sequence][layer][position][kv_head][head_dim] in kv_dtype:
 * inspection / tests, and the seam for an external cache manager. */
WOQ_API void* woq_engine_kv_cache_ptr(woq_engine* e, int which);
/* 1 when the decode step hands activations between its kernels as XQ limb blocks (csrc/woq_xq.h; every layer's blobs
 * qualify, one GPU, WOQ_ENGINE_XQ != 0), 0 when it runs the fp32-activation kernels. */
WOQ_API int woq_engine_uses_xq(woq_engine* e);
/* capture one step into a hipGraph and replay it `n` times (greedy chaining). Round 5: a greedy capture also holds the
 * step chained 8 times as ONE graph (WOQ_ENGINE_GRAPH_UNROLL, 1 = off; interior token boundaries: argmax + next
 * embedding as one launch); replay(n) then issues n / 8 launches of it and n % 8 single steps — same kernels, same
 * tokens, +0.6 % tokens/s (each hipGraphLaunch costs the device a few us between tokens). */
WOQ_API int woq_engine_capture(woq_engine* e, int greedy, void* stream);
WOQ_API int woq_engine_replay(woq_engine* e, int n, void* stream);
/* `n` decode steps issued eagerly back to back (no graph, no host synchronisation; with greedy != 0 the token /
 * position chain on the device like a replayed graph). Same device time as woq_engine_replay on the same stream; costs
 * the host ~2.6 us per launch. (Round 4 note: graph replays looked ~1 us per kernel boundary slower until their launch
 * was moved off a stream that sat behind cross-stream event waits — profiles/r04ab_stream_mode_probe.txt.) */
WOQ_API int woq_engine_steps(woq_engine* e, int n, int greedy, void* stream);
/* tensor-parallel seam: when set, the engine calls `fn(user, buf_dev, count_f32, stream)` after
 * o_proj and after down_proj (row-parallel partial sums -> sum over ranks). The Python host binds it
 * to RCCL via torch.distributed. NULL = single GPU. */
typedef int (*woq_allreduce_fn)(void* user, void* buf_dev, size_t count, void* stream);
WOQ_API int woq_engine_set_allreduce(woq_engine* e, woq_allreduce_fn fn, void* user);
/* ---- device-side tensor-parallel exchange (one process per GPU) ----------------------------------
 * The reference has no counterpart (its multi-device inference is DeepSpeed AutoTP on fp models,
 * neural_chat/models/model_utils.py:238-311); this is the exchange step the sharded int4 path needs: the sum over ranks
 * of the row-parallel [hidden] partials after o_proj and down_proj, and the greedy-token pick over a vocab-sharded
 * lm_head. Both are kernels on the caller's stream (capturable): every rank stores 8-byte {fp32, tag} granules straight
 * into every peer's inbox over xGMI (hipIpc-mapped) and sums what arrives in its own, rank order 0..W-1 (bit-identical
 * results on all ranks). Set-up, per rank: create -> handle (64 bytes, exchanged by the host, e.g. torch.distributed
 * all_gather) -> connect(all handles, device ordinal of every rank). All ranks must issue the same sequence of
 * collectives. A peer that does not show up within the timeout (default 2 s) sets a sticky status instead of hanging. */
typedef struct woq_comm woq_comm;
WOQ_API int woq_comm_create(int rank, int world, size_t max_elems, woq_comm** out);
WOQ_API int woq_comm_handle(woq_comm* c, void* handle_out, size_t bytes);
WOQ_API int woq_comm_connect(woq_comm* c, const void* handles, const int* peer_devices);
/* in-place sum over ranks of buf[0..n) fp32, n <= max_elems */
WOQ_API int woq_comm_allreduce_f32(woq_comm* c, float* buf_dev, size_t n, void* stream);
/* synchronises `stream`; *status_out: 0 = every collective so far completed, bit 0 / 1 = an all-reduce / a token
 * exchange timed out (results after that are undefined) */
WOQ_API int woq_comm_status(woq_comm* c, void* stream, int* status_out);
WOQ_API int woq_comm_set_timeout_ms(woq_comm* c, int ms);
WOQ_API void woq_comm_destroy(woq_comm* c);
/* make the engine's decode step issue the exchange itself (after o_proj and down_proj, and for the greedy token:
 * token id = vocab_offset + local argmax of this rank's lm_head rows), so woq_engine_capture / replay cover a whole
 * tensor-parallel token. The comm must outlive the engine's use of it. NULL = back to the callback / single GPU. */
WOQ_API int woq_engine_set_comm(woq_engine* e, woq_comm* comm, int vocab_offset);
/* tensor-parallel decode step with a bound device communicator. xq (default on, WOQ_TP_XQ=0 off): the ranks run the
 * XQ decode kernels (csrc/woq_gemv_xqs.h) and the all-reduce kernel emits the next GEMV's XQ vector itself; off = the
 * fp32-activation kernels. fused_push (default on, WOQ_TP_FUSED_PUSH=0 off; XQ path only): the row-parallel GEMVs
 * (o_proj, down_proj) store their partial sums into the peers' inboxes from their own epilogue, so the fabric flight runs
 * under the kernel boundary and the all-reduce kernel only pulls and sums; off = the all-reduce kernel pushes (the A/B
 * twin; results are bit-identical). Takes effect at the next step / capture. */
WOQ_API int woq_engine_set_tp_options(woq_engine* e, int xq, int fused_push);
/* run only the kernels of one sub-block, for TP where the collective is issued by the host
 * between them: phase 0 = embed + attention block up to o_proj partial, 1 = MLP block up to
 * down partial, 2 = head. layer ignored for phase 2. */
WOQ_API int woq_engine_phase(woq_engine* e, int layer, int phase, int greedy, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WOQ_HIP_H_ */
