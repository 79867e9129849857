// Internal interface between the translation units of libgenrec_hip.so.
#pragma once

#include "common.h"

// Fill res[0..R) / res[R..2R) (device int32) with the trie node whose children
// are the allowed tokens of each row, or a negative GRX_RES_* code.
int grx_resolve_rows(const grx_trie* trie, grx_state* state, const grx_rows* rows, const grx_constraint* con,
                     int32_t* res, hipStream_t stream);

// Close a processor call: count <- (a row matched prefix_ids ? 0 : count) + 1
// when bump_count, and re-arm first_match.
int grx_state_advance(grx_state* state, int bump_count, hipStream_t stream);

int grx_check_step_args(const char* who, const grx_trie* trie, const grx_state* state, const grx_rows* rows,
                        const grx_constraint* con, const void* workspace, size_t workspace_bytes);
