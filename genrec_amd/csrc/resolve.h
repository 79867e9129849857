// Internal interface between the translation units of libgenrec_hip.so.
#pragma once

#include "common.h"

PrefixArg grx_prefix_arg(const grx_constraint* con);

// Open a processor call on `stream`: picks the state slot a count-based call owns, advances the
// host-side call counter and prepares the call -- fixed-hint: the kernel that writes one grx_row_plan
// per row; count-based: the kernel that finds the rows ending in prefix_ids (the streaming kernel
// walks the rows itself with *ctx_out, and does the once-per-call bookkeeping when *bookkeep_out).
int grx_plan_rows(const grx_trie* trie, grx_state* state, const grx_rows* rows, const grx_constraint* con,
                  grx_row_plan* plan, hipStream_t stream, PlanCtx* ctx_out, int* bookkeep_out);

// Planning context for the processor call that will follow on `state` (its slot is the one the
// next count-based call owns; the host-side call counter is left alone), and the plan buffer
// inside a caller workspace.  Used by the kernels that plan ahead for a grx_plan_request.
int grx_plan_ctx_for_next_call(const char* who, const grx_plan_request* req, int32_t n_rows, PlanCtx* ctx,
                               grx_row_plan** plan,
                               hipStream_t stream);
// The argument checks of the above on their own (no state is touched).
int grx_plan_request_check(const char* who, const grx_plan_request* req, int32_t n_rows);
// Plan `R` rows of `T` tokens for the next call with the stand-alone planning kernel (no call counting).
int grx_launch_plan_kernel(const PlanCtx& ctx, const int64_t* ids, int64_t ids_stride, int R, int T,
                           grx_row_plan* plan, hipStream_t stream);

// `vocab`: width of the whole vocabulary when `rows` is a slice of it, -1 when rows->V is that width.
int grx_check_step_args(const char* who, const grx_trie* trie, const grx_state* state, const grx_rows* rows,
                        const grx_constraint* con, const void* workspace, size_t workspace_bytes, int32_t vocab = -1);

// Split rows.  With few rows, one workgroup per row leaves most of the chip idle,
// so a row is streamed by S workgroups (S <= 8, see grx_split_factor) whose partial
// (max, sum) pairs the last one to arrive merges through device-scope atomics.
// The caller's workspace holds the arrival counters ([GRX_SPLIT_MAX_ROWS] int32, zero
// between calls) followed by two 8-byte words per part -- the partial (max, sum) pair and, with the
// repetition penalty, the part's two penalty sums; the row plans come after them.
#define GRX_SPLIT_MAX_ROWS 512
#define GRX_SPLIT_MAX_PARTS 2048
#define GRX_SPLIT_COUNT_BYTES (GRX_SPLIT_MAX_ROWS * 4)
#define GRX_WS_PLAN_OFFSET (GRX_SPLIT_COUNT_BYTES + GRX_SPLIT_MAX_PARTS * 16)
int grx_split_factor(int R, int row_vecs);

// Device view of a trie handle for kernel arguments.
TrieView grx_trie_view(const grx_trie* trie);

// Beam step pieces shared with the streaming kernel's fused form (beam_step.hip).
struct BeamDerived;
int grx_beam_step_check(const char* who, const grx_beam_step_args* args);
bool grx_beam_step_is_small(const grx_beam_step_args& a);
int grx_beam_step_derive(const char* who, const grx_beam_step_args& a, hipStream_t stream, BeamDerived* d_out, PlanCtx* pc,
                         grx_row_plan** plan_out);
void grx_beam_step_launch(const grx_beam_step_args& a, const BeamDerived& d, const PlanCtx& pc, grx_row_plan* plan_out,
                          hipStream_t stream);
