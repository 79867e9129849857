/*
 * genrec_hip.h -- C ABI of libgenrec_hip.so: trie-constrained semantic-ID (SID)
 * decoding for MI355X (gfx950).
 *
 * Every entry point takes plain pointers and sizes.  Pointers documented as
 * "device" must be HIP device memory on the device that is current when the
 * call is made; "host" pointers are ordinary memory.  All functions return
 * GRX_OK (0) on success or a negative GRX_E_* code; grx_last_error() gives a
 * message for the calling thread's last failure.  Kernels are enqueued on the
 * caller's stream (hipStream_t passed as void*) and never synchronise unless
 * the function says so.
 *
 * Each declaration cites the reference code (WncFht/GenRec, paths relative to
 * the reference checkout) whose behaviour it replaces.
 */
#ifndef GENREC_HIP_H
#define GENREC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRX_ABI_VERSION 15

#if defined(GRX_BUILDING)
#define GRX_API __attribute__((visibility("default")))
#else
#define GRX_API
#endif

/* status codes */
#define GRX_OK 0
#define GRX_E_INVALID (-1)     /* bad argument (null pointer, negative size, ...) */
#define GRX_E_HIP (-2)         /* a HIP runtime call failed                         */
#define GRX_E_UNSUPPORTED (-3) /* argument outside what the kernels handle          */
#define GRX_E_WORKSPACE (-4)   /* workspace too small                               */

/* element types of logit rows */
#define GRX_F32 0
#define GRX_BF16 1
#define GRX_F16 2

/* how the allowed-token set of a row is resolved */
#define GRX_MODE_COUNT 0      /* logit_processor.py:29-49 (count-based window)      */
#define GRX_MODE_FIXED_HINT 1 /* fixed_hint_logit_processor.py:18-61 (prefix scan)  */

/* per-row resolution failures, reported through grx_state (lowest row wins) */
#define GRX_ROWERR_NONE 0
#define GRX_ROWERR_EMPTY 1   /* AssertionError "No valid tokens for prefix_ids"
                                (logit_processor.py:43-44,
                                 fixed_hint_logit_processor.py:59-60)               */
#define GRX_ROWERR_NOMATCH 2 /* ValueError "Failed to find a valid trie prefix
                                match" (fixed_hint_logit_processor.py:37-40)        */
#define GRX_ROWERR_OVERFLOW 3 /* allowed set larger than the compact buffer (cmax)  */

#define GRX_MAX_PREFIX 16 /* longest assistant-prefix token list accepted           */

typedef struct grx_trie grx_trie;   /* flat CSR child table resident in HBM         */
typedef struct grx_state grx_state; /* device-resident processor state words        */

GRX_API int grx_abi_version(void);
GRX_API const char* grx_last_error(void);

/* ------------------------------------------------------------------ trie --- */

/*
 * Upload a prefix trie as a CSR child table.  Replaces the nested-dict
 * llamafactory.extras.trie.Trie (src/llamafactory/extras/trie.py:18-97).
 *
 * Node 0 is the root.  The children of node n are the edges
 * child_off[n] .. child_off[n+1]-1; child_tok[e] is the token on edge e and
 * is strictly increasing inside one node; the node reached through edge e has
 * id e + 1 (so n_nodes == n_edges + 1).  child_off has n_nodes + 1 entries.
 * Host pointers; the arrays are copied.  Synchronous.
 */
GRX_API int grx_trie_create(const int32_t* child_off, const int32_t* child_tok, int32_t n_nodes, int32_t n_edges,
                    grx_trie** out);
GRX_API int grx_trie_destroy(grx_trie* trie);
/* largest number of children of any node (upper bound for cmax below) */
GRX_API int32_t grx_trie_max_fanout(const grx_trie* trie);
/* largest token id on any edge (-1 for an empty trie) */
GRX_API int32_t grx_trie_max_token(const grx_trie* trie);

/* ----------------------------------------------------------------- state --- */

/*
 * Device-resident words that carry what ConstrainedLogitsProcessor keeps in
 * Python attributes: the shared step counter `count`
 * (logit_processor.py:24,27,40,48) and the first per-row failure of a call.
 * Keeping them on the device lets consecutive decode steps run without a
 * host round trip.  grx_state_create/destroy/read synchronise;
 * grx_state_clear_error is enqueued on `stream`; grx_state_set_count enqueues
 * nothing (the value is handed to the kernel that plans the next call, on that
 * call's stream, and grx_state_read reports it at once).
 */
GRX_API int grx_state_create(grx_state** out);
GRX_API int grx_state_destroy(grx_state* st);
GRX_API int grx_state_set_count(grx_state* st, int32_t count, void* stream);
GRX_API int grx_state_clear_error(grx_state* st, void* stream);
/* out[0]=count, out[1]=row error code (GRX_ROWERR_*), out[2]=row index of it, out[3]=calls so far */
GRX_API int grx_state_read(grx_state* st, int32_t out[4], void* stream);

/* ------------------------------------------------- constrained log-softmax --- */

typedef struct grx_rows {
    const int64_t* input_ids; /* device [R, T], row stride ids_stride (elements)   */
    int64_t ids_stride;
    int32_t R;                /* rows = batch * beams                               */
    int32_t T;                /* tokens per row so far                              */
    const void* scores;       /* device [R, V] logits, row stride scores_stride     */
    int64_t scores_stride;    /* elements                                           */
    int32_t V;
    int32_t dtype;            /* GRX_F32 / GRX_BF16 / GRX_F16                       */
    int32_t planned;          /* 1: the kernel that produced exactly these rows has already
                                 prepared this call (see grx_plan_request); 0: prepare it now */
    int32_t reserved;
} grx_rows;

typedef struct grx_constraint {
    int32_t mode;             /* GRX_MODE_*                                         */
    int32_t prefix_index;     /* logit_processor.py:15                              */
    int32_t n_prefix;         /* -1: prefix_ids is None; else length (<= GRX_MAX_PREFIX) */
    int32_t eos_token_id;     /* -1: None                                           */
    int64_t prefix_ids[GRX_MAX_PREFIX];
    /* Optional fused pre-processor: transformers' RepetitionPenaltyLogitsProcessor
     * (generation/logits_process.py, the 2-D path of __call__) applied to
     * log_softmax(scores) before the masked log-softmax, which is where generate()
     * runs it ahead of the reference's processor when repetition_penalty != 1
     * (evaluate.py:218,363-371).  1.0 (or 0) switches it off.  Tokens of
     * input_ids[:, penalty_skip:] are penalised; rows may hold at most
     * GRX_MAX_PENALTY_TOKENS tokens then, and V at most GRX_MAX_PENALTY_VOCAB (a
     * workgroup marks the row's tokens in a bitmap over the vocabulary that has to
     * fit its LDS). */
    float repetition_penalty;
    int32_t penalty_skip;     /* prompt_ignore_length of the HF processor; 0 = none  */
    /* 1: the caller vouches that every row of `scores` already is a log-softmax
     * (log-sum-exp 0), which is what generate() hands a processor when nothing ran
     * before it (generation/utils.py:3388-3389; the reference's second log_softmax,
     * logit_processor.py:30, then changes values by < 1e-6).  The rows are not
     * streamed: allowed positions receive scores[r, v] as is, the dense output costs
     * one write pass instead of a read and a write.  Not combinable with the fused
     * repetition penalty. */
    int32_t assume_normalized;
    /* Compact form only (grx_constrained_candidates*): 0 = emit every allowed token of a row, tokens
     * ascending.  k > 0: a row with more than k + GRX_KEEP_TOP_SLACK allowed tokens emits only its k
     * best, in no particular order -- ranked by (logp + row_bias[r]) descending, token ascending,
     * which is how grx_beam_step ranks the candidates of a beam -- and its header count is k; rows
     * with fewer are emitted whole (shedding a handful of entries costs more than writing them).
     * A beam step that keeps n_keep candidates per group never takes more than n_keep from one
     * row, so with k = n_keep and row_bias = the running beam scores
     * (grx_beam_step_args.run_score_in; NULL when every row's score is the same, as on the shared
     * first step) its results are unchanged while candidate rows never exceed
     * k + GRX_KEEP_TOP_SLACK entries: buffers (cmax), their writes and the beam step's reads shrink
     * accordingly.  k <= 512; tries whose widest node has more than 1024 children are refused. */
#define GRX_KEEP_TOP_SLACK 64
    int32_t keep_top;
    const float* row_bias;    /* device fp32 [R] or NULL; read only when keep_top > 0 */
    /* Optional promise for the fused repetition penalty, 0 = none: the rows come in groups of
     * shared_group consecutive rows (the beams of one prompt) whose first shared_prefix_len tokens are
     * equal within a group -- what transformers' beam search holds for the prompt part of its rows
     * (generation/utils.py:3316-3351 expands every prompt into num_beams equal rows and only ever
     * re-orders whole rows inside a group).  The kernel then reads those tokens from the group's first
     * row: one 8-byte-per-token fetch per group and XCD instead of one per row, the others are L2 hits.
     * Results do not depend on it as long as the promise holds.  R must be a multiple of shared_group. */
    int32_t shared_prefix_len;
    int32_t shared_group;
} grx_constraint;
#define GRX_MAX_PENALTY_TOKENS 1536
#define GRX_MAX_PENALTY_VOCAB 320000

/* Bytes of device scratch the two calls below need for R rows (required when
 * R > 0).  It holds the per-row plans that a small kernel writes ahead of the
 * streaming kernel in the fixed-hint mode (the trie node that governs each
 * row; count-based calls walk their rows inside the streaming kernel), and the
 * arrival counters and partial (max, sum) pairs of rows that are streamed by
 * several workgroups (launches of few rows).  It must be 16-byte aligned and ZERO-FILLED
 * when first handed over; every call leaves the counters zero-filled again, so
 * one buffer serves all calls of a stream.  Calls that share a workspace must not
 * overlap. */
GRX_API size_t grx_workspace_bytes(int32_t R);

/*
 * Processor surface.  out[r, v] = log_softmax(scores[r, :])[v] for the tokens
 * v the trie allows after row r, -inf elsewhere; `count` advances by one.
 * Replaces ConstrainedLogitsProcessor.__call__ (logit_processor.py:29-49) and
 * FixedHintConstrainedLogitsProcessor.__call__
 * (fixed_hint_logit_processor.py:79-96).  `out` is device [R, V] with row
 * stride out_stride and element type out_dtype; it must not alias `scores`.
 * `state` may be NULL in GRX_MODE_FIXED_HINT only if errors need not be
 * reported.
 */
GRX_API int grx_constrained_logsoftmax(const grx_trie* trie, grx_state* state, const grx_rows* rows,
                               const grx_constraint* con, void* out, int64_t out_stride, int32_t out_dtype,
                               void* workspace, size_t workspace_bytes, void* stream);

/*
 * The same call for a caller that keeps ONE output tensor and hands it to every call (no
 * counterpart in the reference, whose logit_processor.py:31,49 builds a new mask and a new
 * result per call -- transformers' beam loop reads the result once and drops it, so one tensor
 * can serve every step).  A dense result is -inf except at the few allowed positions of each
 * row; when `out` still holds the previous call's result, writing V x -inf again is all waste:
 * this form resets just the positions the previous call made finite and writes the new ones,
 * so the call costs the read of `scores` alone (with assume_normalized: next to nothing).
 * last_rows: device buffer of grx_reuse_bytes(R) bytes that travels with `out` -- the library
 * records in it which trie node every row of `out` currently shows.  fresh = 1: `out` holds
 * anything (first call, new tensor, other shape): it is filled like grx_constrained_logsoftmax
 * does and last_rows is initialised; fresh = 0: `out` and last_rows are exactly as the previous
 * call on them left them (same R, V, out_dtype, eos_token_id, trie) and nobody wrote to either.
 * Everything else as grx_constrained_logsoftmax.
 */
GRX_API size_t grx_reuse_bytes(int32_t R);
GRX_API int grx_constrained_logsoftmax_reuse(const grx_trie* trie, grx_state* state, const grx_rows* rows,
                                             const grx_constraint* con, void* out, int64_t out_stride,
                                             int32_t out_dtype, void* last_rows, int32_t fresh, void* workspace,
                                             size_t workspace_bytes, void* stream);

/*
 * Compact candidate rows, the hand-off between the fused log-softmax and the
 * beam step.  One row per logit row, made of 8-byte entries {int32 x, int32 y}:
 *   entry 0      = {count, bits of the row's log-sum-exp (fp32)}
 *   entry 1 + i  = {token_i, bits of logp_i (fp32)},  i < count, tokens ascending.
 * The row stride (in entries) is a multiple of 16 and at least cmax + 1, and the
 * buffer is 128-byte aligned, so every row starts on its own 128-byte line and is
 * written in whole lines (entries between count + 1 and the next multiple of 16
 * are zero-filled; anything beyond is left untouched): no store ever covers part
 * of a memory line.
 */
typedef struct grx_cand {
    int32_t x;
    int32_t y;
} grx_cand;

/* smallest valid row stride (entries) for rows of up to cmax candidates */
GRX_API int64_t grx_candidate_stride(int32_t cmax);

/*
 * Fused form of the same step: instead of the dense [R, V] result it emits,
 * per row, the allowed tokens and their log-probabilities in the compact
 * candidate format above -- the finite entries of what
 * grx_constrained_logsoftmax would produce -- which is everything the beam
 * step below consumes.  Reads every logit once and writes O(R * cmax).
 * cand is device [R, cand_stride].
 */
GRX_API int grx_constrained_candidates(const grx_trie* trie, grx_state* state, const grx_rows* rows,
                                       const grx_constraint* con, grx_cand* cand, int64_t cand_stride, int32_t cmax,
                                       void* workspace, size_t workspace_bytes, void* stream);

/*
 * grx_constrained_candidates for callers that cannot allocate device memory (no
 * counterpart in the reference): code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  Same inputs, outputs and results, but no
 * workspace argument: the scratch the kernels need (arrival counters and partial
 * sums of split rows, fixed-hint row plans) is a device array inside the library
 * image, GRX_NOALLOC_MAX_ROWS rows large (about 1 MB, held by every GPU the library
 * runs on until the process ends).  The call makes no hipMalloc / hipFree and calls
 * no allocating routine; the first call on a device looks the array's address up.
 * More than GRX_NOALLOC_MAX_ROWS rows are refused with GRX_E_UNSUPPORTED, and so is
 * grx_rows.planned = 1 (plans made ahead live in the planner's workspace).  All
 * calls on one GPU share the array: calls that overlap on one GPU (two streams, two
 * threads) race -- serialise them.  The trie must live on the current device.
 */
#define GRX_NOALLOC_MAX_ROWS 65536
GRX_API int32_t grx_noalloc_max_rows(void);
GRX_API int grx_constrained_candidates_noalloc(const grx_trie* trie, grx_state* state, const grx_rows* rows,
                                               const grx_constraint* con, grx_cand* cand, int64_t cand_stride,
                                               int32_t cmax, void* stream);

/*
 * The same step over logits that arrive in column slices (no counterpart in the
 * reference, where the model's forward pass materialises the whole [R, V] tensor
 * before logit_processor.py:30 reads it: MIMIGenRec.py:79, evaluate.py:363-371).
 * The caller runs the LM-head matmul over slices of the vocabulary -- sized so that a
 * slice [R, Vc] stays in the memory-side cache -- and hands each slice to
 * grx_constrained_partial as soon as it exists; the slice buffer can be reused for
 * the next one, so the full logits never reach HBM.  A partial call streams its
 * slice once, leaves per row `parts` partial (max, sum-of-exp) pairs in
 * partials[r, slot .. slot + parts) and copies the raw logits of the row's allowed
 * tokens inside the slice into the row's candidate entries.  grx_constrained_finish
 * merges a row's n_slots pairs in slot order and turns the candidate row into the
 * format grx_constrained_candidates emits (same tokens; log-probabilities equal up
 * to fp32 summation order).
 *
 * rows->scores is the slice ([R, Vc], rows->V = Vc); column j is token v_begin + j.
 * The slices of one step must cover [0, v_total) exactly once, all calls of a step
 * share rows->input_ids, cand, partials and workspace, every slot in [0, n_slots)
 * must be filled by exactly one (call, part), and the call with first = 1 -- it
 * opens the processor call: `count` advances, rows are resolved, row errors are
 * reported -- must be ordered before the others and grx_constrained_finish after
 * all of them (one stream, or events).  partials is device fp32 [R, n_slots, 2].
 * Not available with assume_normalized or keep_top (both need whole rows).
 * The library cannot tell whether every slice was fed: grx_constrained_finish merges all
 * n_slots pairs and reads candidate entries 1 .. count whatever wrote them, so slices that do not
 * cover [0, v_total) exactly once, or a slot left unfilled, yield log-probabilities built from
 * stale memory without any error (the Python ChunkedStep checks coverage and the fed count).
 *
 * With the fused repetition penalty (grx_constraint.repetition_penalty) every partial
 * call also copies the raw logits of the row's own tokens ids[r, penalty_skip:T) that
 * lie inside its slice into grx_chunk.pen_logits (device fp32 [R, pen_stride],
 * pen_stride >= T - penalty_skip; shared by the calls of a step, contents need no
 * initialisation), and the step ends with grx_constrained_finish_penalised -- given
 * the same rows (input_ids, T), constraint and pen_logits -- instead of
 * grx_constrained_finish; results equal grx_constrained_candidates' with the same
 * penalty up to fp32 summation order.
 */
#define GRX_MAX_PARTIAL_SLOTS 4096
typedef struct grx_chunk {
    int32_t v_begin; /* token id of column 0 of this slice                          */
    int32_t v_total; /* width of the whole vocabulary                                */
    int32_t slot;    /* first partial slot this call fills                          */
    int32_t n_slots; /* partial slots per row                                       */
    int32_t parts;   /* waves that share one row of the slice (>= 1; see grx_chunk_parts) */
    int32_t first;   /* 1: this call opens the step (exactly one call per step)     */
    float* pen_logits;  /* repetition penalty only (else null): see above            */
    int64_t pen_stride;
} grx_chunk;
/* a good `parts` for R rows of Vc columns: 1 when the rows alone fill the GPU */
GRX_API int32_t grx_chunk_parts(int32_t R, int32_t Vc, int32_t dtype);
GRX_API int grx_constrained_partial(const grx_trie* trie, grx_state* state, const grx_rows* rows,
                                    const grx_constraint* con, const grx_chunk* chunk, float* partials,
                                    grx_cand* cand, int64_t cand_stride, int32_t cmax, void* workspace,
                                    size_t workspace_bytes, void* stream);
GRX_API int grx_constrained_finish(int32_t R, int32_t n_slots, const float* partials, grx_cand* cand,
                                   int64_t cand_stride, int32_t cmax, const void* workspace, size_t workspace_bytes,
                                   void* stream);
GRX_API int grx_constrained_finish_penalised(const grx_rows* rows, const grx_constraint* con, int32_t v_total,
                                             int32_t n_slots, const float* partials, const float* pen_logits,
                                             int64_t pen_stride, grx_cand* cand, int64_t cand_stride, int32_t cmax,
                                             const void* workspace, size_t workspace_bytes, void* stream);

/*
 * The LM head inside the step (no counterpart in the reference, where the model's forward pass
 * writes the whole [R, V] logits tensor -- MIMIGenRec.py:79, evaluate.py:363-371 -- and
 * logit_processor.py:30 reads it back): logits = hidden [R, H] x weight^T, weight being the
 * [V, H] matrix of the output projection (nn.Linear.weight), are formed tile by tile on the matrix
 * cores (bf16 operands, fp32 accumulation) and consumed on the spot.  Per row and 256-column tile
 * only a (max, sum-of-exp) pair leaves the multiplication, plus the fp32 logits of the column
 * tiles at the top of the vocabulary that hold the trie's tokens (token ids 151,665 + rank for
 * semantic-ID tokens: at most the last 4096 columns are kept; a trie token below them gets its
 * logit from a dot product in the finishing kernel).  The result is the candidate rows
 * grx_constrained_candidates would emit for scores = fp32(hidden x weight^T): same tokens, same
 * order, log-probabilities equal up to fp32 summation order.  rows->scores is ignored,
 * rows->V is the vocabulary width, rows->dtype must be GRX_BF16 (the operands' type); H must be a
 * multiple of 64 and operand rows 16-byte aligned.  scratch: device buffer of
 * grx_lm_head_scratch_bytes(trie, R, V) bytes, contents need no initialisation.  Not available
 * with the fused repetition penalty, assume_normalized or keep_top.  Counter, row errors and
 * plans made ahead behave as in grx_constrained_candidates.
 */
typedef struct grx_lm_head {
    const void* hidden;     /* device bf16 [R, H], row stride hidden_stride (elements)       */
    int64_t hidden_stride;
    const void* weight;     /* device bf16 [V, H], row stride weight_stride (elements)       */
    int64_t weight_stride;
    int32_t H;
    int32_t reserved;
} grx_lm_head;
GRX_API size_t grx_lm_head_scratch_bytes(const grx_trie* trie, int32_t R, int32_t V);
GRX_API int grx_lm_head_candidates(const grx_trie* trie, grx_state* state, const grx_rows* rows,
                                   const grx_constraint* con, const grx_lm_head* head, void* scratch,
                                   size_t scratch_bytes, grx_cand* cand, int64_t cand_stride, int32_t cmax,
                                   void* workspace, size_t workspace_bytes, void* stream);

/*
 * Kernel timing for benchmarks (no counterpart in the reference).  While
 * enabled with every = k > 0, every k-th of the two calls above (the k-th,
 * 2k-th, ... after the enable) attaches HIP events to the dispatch of its
 * streaming kernel (compact form) or to the fill kernel's start and the
 * streaming kernel's end (dense form), up to 4096 timed calls per enable; the
 * row planning kernel is not included.  An event pair costs about 5 us of
 * stream time, hence the sampling.  every = 0 switches timing off.
 * grx_timing_read synchronises on the recorded events and returns the summed
 * time in milliseconds and the number of timed calls.  Process-wide and not
 * thread-safe: a measurement hook, not an API.
 */
GRX_API int grx_timing_enable(int every);
GRX_API int grx_timing_read(double* total_ms, int32_t* launches);

/*
 * Read probe for benchmarks (no counterpart in the reference): streams R rows of
 * row_bytes bytes (device, 16-byte aligned, row stride stride_bytes) with the
 * streaming kernel's access pattern -- one workgroup per row, the same 16-byte
 * non-temporal loads -- and nothing else; out (device uint32 [R * 4]) receives a
 * checksum word per wave.  Its throughput is the ceiling this GPU offers that
 * pattern, which puts the streaming kernel's figure in proportion.  Timed like
 * the calls above while timing is enabled.
 */
GRX_API int grx_read_probe(const void* rows, int64_t stride_bytes, int32_t R, int64_t row_bytes, uint32_t* out,
                           void* stream);

/*
 * Planning ahead (no counterpart in the reference; used by the fused loop).  The
 * two calls above start with a small kernel that prepares the call: in the
 * fixed-hint mode it works out which trie node governs every row, in the
 * count-based mode with prefix_ids it finds the lowest row that ends in them (the
 * counter reset of logit_processor.py:39-40; the rows themselves are walked by the
 * streaming kernel, and without prefix_ids there is no such kernel at all).  The
 * kernel that PRODUCES the rows can do that work instead: grx_beam_step /
 * grx_beam_init given a grx_plan_request prepare the call on their output rows
 * (the running sequences, cur_len + 1 resp. prompt_len tokens each; row plans go
 * into `workspace`), and the next processor call on those rows passes
 * grx_rows.planned = 1 with the same trie / state / constraint / workspace.
 * A plan made ahead may be dropped: a call that passes planned = 0 (other rows, or
 * the counter was set in between) plans for itself and first erases what the
 * dropped plan recorded in the state, and so does a second plan made ahead of the
 * same call.  planned = 1 is only valid for exactly the rows that were planned,
 * with no grx_state_set_count since.
 * One workspace per search: in the fixed-hint mode grx_beam_step derives a row's
 * plan from the plan of its parent row, which it reads from this same `workspace`
 * -- so the processor call that produced the candidates handed to grx_beam_step
 * must have run on B * K rows with this workspace (or, with cand_shared, on B rows;
 * then nothing is read back).  A workspace that did not serve that call yields
 * plans for the wrong rows.
 */
typedef struct grx_plan_request {
    const grx_trie* trie;
    grx_state* state;          /* required in GRX_MODE_COUNT                          */
    const grx_constraint* con;
    void* workspace;           /* at least grx_workspace_bytes(B * K) bytes           */
    size_t workspace_bytes;
} grx_plan_request;

/* -------------------------------------------------------------- beam step --- */

/*
 * One iteration of beam search over the compact candidates: the arithmetic of
 * transformers.GenerationMixin._beam_search steps b-g
 * (generation/utils.py:3418-3502 of transformers 5.15.0: accumulate, top-2K,
 * parent/token split, stopping criteria, running top-K, finished merge,
 * early-stop heuristic), called by the reference from evaluate.py:363-371 and
 * MIMIGenRec.py:79.  One workgroup per prompt group.  Ties are broken by
 * (score descending, beam*V+token ascending).
 *
 * All pointers are device memory.  *_in and *_out must be different buffers.
 * B groups of K beams; n_keep = max(2, 1 + n_eos) * K candidates are kept.
 */
typedef struct grx_beam_step_args {
    int32_t B, K, V, n_keep;
    int32_t cmax;
    int32_t cur_len;        /* tokens per running sequence before this step        */
    int32_t prompt_len;     /* decoder_prompt_len                                   */
    int32_t max_length;     /* row length of the sequence buffers                   */
    int32_t copy_from;      /* first position of a sequence row that must be
                               copied when beams are re-ordered (0 = whole row;
                               prompt_len when all beams of a group share the
                               prompt and the out buffers already hold it)         */
    int32_t eos_token_id;   /* -1: none                                             */
    int32_t early_stopping; /* 0 False, 1 True, 2 "never"                           */
    float length_penalty;
    const grx_cand* cand;      /* [B*K, cand_stride] compact candidate rows         */
    int64_t cand_stride;       /* entries per row                                   */
    /* running beams */
    const int64_t* run_seq_in;  int64_t* run_seq_out;      /* [B, K, max_length]     */
    const float* run_score_in;  float* run_score_out;      /* [B, K]                 */
    const int32_t* run_bidx_in; int32_t* run_bidx_out;     /* [B, K, max_length - prompt_len] */
    /* finished beams */
    const int64_t* fin_seq_in;  int64_t* fin_seq_out;      /* [B, K, max_length]     */
    const float* fin_score_in;  float* fin_score_out;      /* [B, K]                 */
    const int32_t* fin_bidx_in; int32_t* fin_bidx_out;     /* [B, K, max_length - prompt_len] */
    const uint8_t* fin_done_in; uint8_t* fin_done_out;     /* [B, K] is_sent_finished */
    uint8_t* unsat;            /* [B] is_early_stop_heuristic_unsatisfied, in/out   */
    /* per-step results (may be NULL) */
    float* topk_score;         /* [B, n_keep]                                       */
    int32_t* topk_parent;      /* [B, n_keep] beam index inside the group           */
    int32_t* topk_token;       /* [B, n_keep]                                       */
    uint8_t* topk_hit;         /* [B, n_keep] next_token_hits_stopping_criteria     */
    int32_t* next_parent;      /* [B*K] flat cache-reorder index (b*K + parent)     */
    const grx_plan_request* plan_next; /* NULL, or: also plan the next processor call  */
    int32_t cand_shared;       /* 1: cand is [B, cand_stride], ONE candidate row per group
                                  that all K beams extend from.  For the first step of a
                                  search whose K beams are copies of one prompt (what
                                  generate() builds in _expand_inputs_for_generation): the
                                  K logit rows are identical, so the processor call needs
                                  to see only one of them.  Same results as K equal rows. */
    int32_t reserved;
} grx_beam_step_args;

GRX_API int grx_beam_step(const grx_beam_step_args* args, void* stream);

/*
 * Initial state of a beam search (generation/utils.py:3316-3351 of transformers
 * 5.15.0): sequences = prompt + fill, running scores [0, -1e9, ...], finished
 * scores -1e9, parent trails -1, nothing finished.  Both halves [0] and [1] of
 * every double buffer used by grx_beam_step are written, so either can be the
 * first *_in.  input_ids is device [B*K, prompt_len] with row stride ids_stride.
 */
typedef struct grx_beam_init_args {
    int32_t B, K, prompt_len, max_length;
    int64_t fill_token;
    const int64_t* input_ids;
    int64_t ids_stride;
    int64_t* run_seq[2];
    int64_t* fin_seq[2];
    float* run_score[2];
    float* fin_score[2];
    int32_t* run_bidx[2];
    int32_t* fin_bidx[2];
    uint8_t* fin_done[2];
    uint8_t* unsat;
    const grx_plan_request* plan_next; /* NULL, or: also plan the first processor call */
    int32_t plan_one_per_group; /* with plan_next: that call will see one row per group (rows
                                   0, K, 2K, ... of input_ids; see grx_beam_step_args.cand_shared),
                                   so B rows are planned instead of B * K */
    int32_t reserved;
} grx_beam_init_args;

GRX_API int grx_beam_init(const grx_beam_init_args* args, void* stream);

/*
 * A whole decode step in one call: grx_constrained_candidates(trie, state, rows, con, cand, ...)
 * followed by grx_beam_step(beam) on those candidates -- the loop body of _beam_search
 * (generation/utils.py:3384-3508 of transformers 5.15.0) that the reference enters from
 * MIMIGenRec.py:79 and evaluate.py:363-371, with logit_processor.py:29-49 inside it.  Same
 * arguments, same results and same side effects (counter, row errors, the plan made for the next
 * call) as the two calls in that order; beam->cand, beam->cand_stride and beam->cmax must be the
 * candidate buffer given here, and rows->R == beam->B * beam->K (beam->B with cand_shared).
 * Where it can, the call is ONE launch: the workgroup that emits the last candidate row of a
 * prompt group carries out that group's beam step on the spot, while the rows of other groups are
 * still being streamed, so only the last groups' steps add to the kernel's duration.  That form
 * serves searches of at most 16 beams that keep at most 64 candidates, 16-byte aligned logit rows,
 * no repetition penalty, no assume_normalized, no cand_shared, and count-based calls that arrive
 * planned (grx_rows.planned = 1) or carry prefix_ids; anything else runs the two kernels back to
 * back.  The workspace must be the one whose plan entries this state's calls use (zero-filled when
 * first handed over): the spare word of a group's first plan entry counts the group's finished rows.
 */
GRX_API int grx_constrained_beam_step(const grx_trie* trie, grx_state* state, const grx_rows* rows,
                                      const grx_constraint* con, grx_cand* cand, int64_t cand_stride, int32_t cmax,
                                      void* workspace, size_t workspace_bytes, const grx_beam_step_args* beam,
                                      void* stream);

/* ----------------------------------------------------------------- scoring --- */

/*
 * Rule / prefix / NDCG rewards on SID code tuples (the integers the reference
 * extracts with re.findall(r"_(\d+)") -- util.py:133-134).  Replaces
 * rule_reward, prefix_rule_reward and ndcg_rule_reward
 * (rewards/ranking_reward.py:42-110).  pred [R, Lp] and gt [R, Lg] are
 * contiguous device int32 code rows with per-row lengths pred_len / gt_len
 * (NULL: the full width).  Any of the three outputs may be NULL; output r is
 * written at out[r * out_stride] (fp32), so the outputs can be columns of one
 * [R, n_funcs] matrix.  ndcg_weights is device fp32 [K] (-1/log2(i+2)
 * normalised to sum -1, computed by the caller in float64 as
 * ranking_reward.py:83-84 does).  Rows beyond the last full group of K get no
 * NDCG value, as in the reference loop (ranking_reward.py:101-107).  K <= 256.
 */
GRX_API int grx_sid_rewards(const int32_t* pred, const int32_t* pred_len, int32_t Lp, const int32_t* gt,
                            const int32_t* gt_len, int32_t Lg, int32_t R, int32_t K, int32_t prefix_normalize,
                            const float* ndcg_weights, float* rule_out, float* prefix_out, float* ndcg_out,
                            int32_t out_stride, void* stream);

/*
 * The same rewards straight from token ids: pred_tok / gt_tok are device int64
 * rows (as generate() returns them) with row strides in elements, and
 * code_table[token] (device int32 [table_size]) is the SID code of a token,
 * negative for tokens that carry none (EOS, padding, ordinary text): those are
 * skipped, like text the regular expression does not match.  Oracle hint
 * tokens are simply part of the pred row (ranking_reward.py:25-30).
 */
GRX_API int grx_sid_rewards_tokens(const int64_t* pred_tok, int64_t pred_stride, int32_t Lp, const int64_t* gt_tok,
                                   int64_t gt_stride, int32_t Lg, const int32_t* code_table, int32_t table_size,
                                   int32_t R, int32_t K, int32_t prefix_normalize, const float* ndcg_weights,
                                   float* rule_out, float* prefix_out, float* ndcg_out, int32_t out_stride,
                                   void* stream);

/*
 * Rollout scoring in one launch: the rewards of grx_sid_rewards_tokens written as the columns of
 * rewards_per_func (device fp32 [R, n_funcs]; func_kinds, host int32 [n_funcs], names the reward of
 * each column: 0 rule, 1 prefix, 2 ndcg, each at most once) and, from the same values, what
 * grx_group_advantage(rewards_per_func, reward_weights, R, n_funcs, K, scale, eps, ...) would
 * compute -- bit for bit -- for scale 0 ("none") or 1 ("group").  Groups never straddle ranks
 * (the rollout shards whole prompt groups), so these are the local rows' final advantages
 * (fixed_hint_grpo_trainer.py:383-408); the all-gather of rewards_per_func that the reference
 * runs first only feeds logging and scale "batch".  R must be a multiple of K.
 */
GRX_API int grx_rollout_score_tokens(const int64_t* pred_tok, int64_t pred_stride, int32_t Lp, const int64_t* gt_tok,
                                     int64_t gt_stride, int32_t Lg, const int32_t* code_table, int32_t table_size,
                                     int32_t R, int32_t K, int32_t prefix_normalize, const float* ndcg_weights,
                                     float* rewards_per_func, int32_t n_funcs, const int32_t* func_kinds,
                                     const float* reward_weights, int32_t scale, float eps, float* rewards,
                                     float* advantages, float* group_mean, float* group_std, void* stream);

/*
 * HR@k / NDCG@k support: rank of the first beam equal to the target
 * (evaluate.py:162-166).  beams [n, n_beam, L] and target [n, L] are device
 * int32 code rows (a beam equals the target when all L codes match).
 * first_hit [n] receives the rank, or n_beam when no beam matches (may be
 * NULL); rank_hist [n_beam + 1] (device int32, zeroed by the caller) counts
 * the samples per rank.  HR@k and NDCG@k (evaluate.py:168-180) follow from the
 * histogram on the host in float64.
 */
GRX_API int grx_hit_ranks(const int32_t* beams, const int32_t* target, int32_t n, int32_t n_beam, int32_t L,
                  int32_t* first_hit, int32_t* rank_hist, void* stream);

/*
 * Per-group advantage: rewards[r] = nansum_f(rewards_per_func[r, f] *
 * weights[f]); adv = (reward - mean_g) / (std_g + eps) with the unbiased std
 * of each group of K consecutive rows (scale 1 "group"), the unbiased std of
 * all R rows (scale 2 "batch") or no division (scale 0 "none").  Replaces
 * fixed_hint_grpo_trainer.py:383-401 and
 * TokenPrefixGRPOTrainer._group_normalize
 * (token_prefix_grpo_trainer.py:41-45).  rewards_per_func is device fp32
 * [R, n_funcs] (after the all-gather), weights device fp32 [n_funcs].
 * Outputs (device fp32 [R], any may be NULL): rewards, advantages,
 * group_mean, group_std.  Scale 2 first reduces all R rows once (one extra
 * single-workgroup kernel) into a scratch word inside the library image (64
 * words per GPU, handed out round robin: up to 64 such calls may be in flight
 * on one GPU).  No scale allocates anything, so the call can be captured in a
 * HIP graph.
 */
GRX_API int grx_group_advantage(const float* rewards_per_func, const float* weights, int32_t R, int32_t n_funcs,
                        int32_t K, int32_t scale, float eps, float* rewards, float* advantages,
                        float* group_mean, float* group_std, void* stream);

/*
 * The exchange step of a rollout shard inside the library: the all-gather of the per-rank reward
 * matrices that the reference issues through accelerate (`gather(rewards_per_func)`,
 * fixed_hint_grpo_trainer.py:382; token_prefix_grpo_trainer.py:215,241-242,249,274), as an RCCL
 * all-gather enqueued on the caller's stream -- reward kernel, gather and advantage kernel are then
 * ordered by that one stream.  The communicator is the library's own: if the process also drives
 * torch.distributed's NCCL communicator over the same ranks (on its own stream), collectives of the
 * two must not be in flight at the same time on a rank -- every rank has to issue them in one common
 * order (synchronise the stream before switching communicators).  grx_comm_destroy synchronises the
 * device first, so no gather of this communicator is cut short.
 * One process per GPU: rank 0 draws an id with grx_comm_unique_id
 * (GRX_COMM_ID_BYTES opaque bytes), hands it to the other ranks by any means (a file, a
 * torch.distributed store), and every rank calls grx_comm_create with the device it will use
 * current (collective: returns when all n_ranks have called).  grx_comm_allgather places rank i's
 * bytes_per_rank bytes at recv + i * bytes_per_rank on every rank; device buffers.  RCCL is bound
 * at run time (librccl.so, the copy already in the process if there is one); without it these
 * calls return GRX_E_UNSUPPORTED and the rest of the library is unaffected.
 */
#define GRX_COMM_ID_BYTES 128
typedef struct grx_comm grx_comm;
GRX_API int grx_comm_unique_id(void* id_out);
GRX_API int grx_comm_create(const void* id, int32_t n_ranks, int32_t rank, grx_comm** out);
GRX_API int grx_comm_destroy(grx_comm* comm);
GRX_API int grx_comm_allgather(grx_comm* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream);

/*
 * Token-wise masked group normalisation
 * (TokenPrefixGRPOTrainer._group_normalize_tokenwise,
 * token_prefix_grpo_trainer.py:47-63): for every group of K rows and token
 * position t, mean/var over the rows with mask 1 (biased variance, denominator
 * clamped to 1), out = (v - mean) / sqrt(var + eps) * mask.  values / mask /
 * out are device fp32 [R, Tc].
 */
GRX_API int grx_group_normalize_tokenwise(const float* values, const float* mask, int32_t R, int32_t Tc, int32_t K,
                                  float eps, float* out, void* stream);

/*
 * Token-level prefix advantages: what TokenPrefixGRPOTrainer._generate_and_score_completions
 * computes after generation (token_prefix_grpo_trainer.py:196-278) from the token-space
 * reward builders (_build_prefix_token_rewards_with_error_mask, :107-149; the older
 * _build_prefix_token_rewards, :65-105, yields the same token and sequence rewards) --
 * there per-row Python loops with an .item() read-back per compared token.
 *
 * Row i's completion (completion_ids[i, :Tc], valid where completion_mask[i, p] > 0, a
 * terminal eos_token_id ignored) is matched against gt_ids[i, :gt_len[i]], the token ids
 * of its ground-truth text (:123): the matched prefix earns 1 / gt_len per token
 * (normalize = 1) or 1, the row's sequence reward is their sum, valid positions behind
 * the matched prefix form the error-token mask.  Then, for groups of K consecutive rows:
 *   total_token_normalize = 1 (:209-246): raw token rewards
 *       prefix_weight * prefix + ndcg_weight * ndcg_tok, ndcg_tok = 0 (has_ndcg = 0),
 *       rank_weights[(i + rank_offset) % K] on the row's error tokens if any row of its
 *       group has a positive sequence reward (error_token_penalty = 1; rank_weights are the
 *       values of _build_ndcg_rank_penalties, :151-161, rank_offset = process_index * R),
 *       or ndcg[i] on every valid token; normalised per (group, position) as
 *       grx_group_normalize_tokenwise does.  Needs scratch [R, Tc].
 *   total_token_normalize = 0 (:247-277): the sequence rewards normalised per group
 *       (mean, unbiased std, eps) and spread over the matched tokens in proportion to their
 *       token rewards, times prefix_weight; plus, with has_ndcg, ndcg_weight times the
 *       group-normalised ndcg[i] on every valid token.
 * The reference gathers the rewards of all ranks first and slices its own rows out again;
 * groups never straddle ranks, so the local rows give the same values.  All pointers are
 * device memory; completion_mask / outputs are fp32 [R, Tc] contiguous, seq_rewards [R];
 * token_rewards, seq_rewards, error_mask may be null.  gt_len[i] must not exceed the gt_stride
 * entries row i owns (larger values are cut to it); with Tc == 0 only seq_rewards is written (zeros).
 */
typedef struct grx_token_prefix_args {
    const int64_t* completion_ids;
    int64_t ids_stride;
    const float* completion_mask;
    const int64_t* gt_ids;
    int64_t gt_stride;
    const int32_t* gt_len;
    const float* ndcg;          /* [R] sequence-level NDCG rewards (ndcg_rule_reward) or null */
    const float* rank_weights;  /* [K] or null                                               */
    int32_t R, Tc, K;
    int32_t eos_token_id;       /* -1: none                                                  */
    int32_t normalize;          /* prefix_reward_normalize                                   */
    int32_t total_token_normalize, has_ndcg, error_token_penalty;
    int32_t rank_offset;
    float prefix_weight, ndcg_weight, eps;
    float* token_rewards;
    float* seq_rewards;
    float* error_mask;
    float* scratch;
    float* token_advantages;
} grx_token_prefix_args;
GRX_API int grx_token_prefix_advantages(const grx_token_prefix_args* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GENREC_HIP_H */
