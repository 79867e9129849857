// Internal interface between the translation units of libgenrec_hip.so.
#pragma once

#include "common.h"

PrefixArg grx_prefix_arg(const grx_constraint* con);

// Open a count-based processor call on `stream`: picks the state slot the call
// owns, enqueues the scan for rows that end in prefix_ids (when prefix_ids is
// given) and advances the host-side call counter.  With no rows it only moves
// the counter forward.
int grx_begin_count_call(grx_state* state, const grx_rows* rows, const grx_constraint* con, hipStream_t stream,
                         int* slot_out);

int grx_check_step_args(const char* who, const grx_trie* trie, const grx_state* state, const grx_rows* rows,
                        const grx_constraint* con, const void* workspace, size_t workspace_bytes);
