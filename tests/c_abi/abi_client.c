/*
 * A plain C client of libgenrec_hip.so: shows that include/genrec_hip.h is a C
 * header (this file is compiled with gcc -std=c99 -pedantic-errors) and that the
 * library can be driven without Python or torch.
 *
 *   abi_client host   no GPU needed: version, struct sizes, the pure host helpers,
 *                     and argument validation that fails before any device work
 *   abi_client gpu    one constrained decoding step on device 0 (trie upload,
 *                     count-based compact candidates, beam step), checked against
 *                     a float64 computation in this file; the same candidates once
 *                     more through the entry point that allocates nothing (byte
 *                     for byte) and through the step over column slices (tokens
 *                     equal, log-probs within 1e-5)
 *
 * Prints one "key=value" line per fact and "OK" last; exit status 0 on success.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "genrec_hip.h"

/* The HIP runtime entry points this client needs, declared by hand so that the
 * file stays plain C99 (no HIP headers); hipError_t 0 is success, hipMemcpyKind
 * 1 / 2 are host-to-device / device-to-host. */
extern int hipMalloc(void** ptr, size_t size);
extern int hipFree(void* ptr);
extern int hipMemcpy(void* dst, const void* src, size_t size, int kind);
extern int hipMemset(void* dst, int value, size_t size);
extern int hipDeviceSynchronize(void);
extern int hipSetDevice(int device);

#define CHECK(cond, msg)                                              \
    do {                                                              \
        if (!(cond)) {                                                \
            fprintf(stderr, "FAILED: %s (%s)\n", msg, grx_last_error()); \
            return 1;                                                 \
        }                                                             \
    } while (0)

static int host_checks(void) {
    grx_rows rows;
    grx_constraint con;
    int rc;
    printf("abi_version=%d\n", grx_abi_version());
    CHECK(grx_abi_version() == GRX_ABI_VERSION, "library and header disagree on the ABI version");
    printf("sizeof_grx_rows=%zu\n", sizeof(grx_rows));
    printf("sizeof_grx_constraint=%zu\n", sizeof(grx_constraint));
    printf("sizeof_grx_plan_request=%zu\n", sizeof(grx_plan_request));
    printf("sizeof_grx_beam_step_args=%zu\n", sizeof(grx_beam_step_args));
    printf("sizeof_grx_beam_init_args=%zu\n", sizeof(grx_beam_init_args));
    printf("sizeof_grx_cand=%zu\n", sizeof(grx_cand));
    printf("sizeof_grx_chunk=%zu\n", sizeof(grx_chunk));
    printf("sizeof_grx_lm_head=%zu\n", sizeof(grx_lm_head));
    printf("noalloc_max_rows=%d\n", (int)grx_noalloc_max_rows());
    CHECK(grx_noalloc_max_rows() == GRX_NOALLOC_MAX_ROWS, "library and header disagree on the scratch size");
    CHECK(grx_chunk_parts(8192, 8192, GRX_BF16) == 1 && grx_chunk_parts(16, 65536, GRX_BF16) > 1,
          "rows alone fill the GPU at 8192 rows; few rows are shared by several waves");
    printf("candidate_stride_242=%lld\n", (long long)grx_candidate_stride(242));
    printf("workspace_bytes_8192=%zu\n", grx_workspace_bytes(8192));
    CHECK(grx_candidate_stride(15) == 16 && grx_candidate_stride(16) == 32, "candidate stride rule");
    /* argument validation happens on the host, before any device call */
    memset(&rows, 0, sizeof(rows));
    memset(&con, 0, sizeof(con));
    rc = grx_constrained_candidates(NULL, NULL, &rows, &con, NULL, 16, 4, NULL, 0, NULL);
    CHECK(rc == GRX_E_INVALID, "a null trie must be rejected with GRX_E_INVALID");
    printf("null_trie_message=%s\n", grx_last_error());
    rc = grx_beam_step(NULL, NULL);
    CHECK(rc == GRX_E_INVALID, "null beam-step arguments must be rejected");
    CHECK(grx_trie_max_fanout(NULL) == 0 && grx_trie_max_token(NULL) == -1, "null trie accessors");
    return 0;
}

/* xorshift PRNG: the same numbers on every run */
static unsigned long long g_seed = 88172645463325252ULL;
static double uniform01(void) {
    g_seed ^= g_seed << 13;
    g_seed ^= g_seed >> 7;
    g_seed ^= g_seed << 17;
    return (double)(g_seed >> 11) / 9007199254740992.0;
}

#define NV 5000
#define NK 4
#define NB 3
#define NR (NB * NK)
#define NT 6       /* 3 filler tokens + the 3-token prefix */
#define NCMAX 8

static int gpu_checks(void) {
    /* trie: root -(10)-> -(11)-> -(12)-> node 3 with children {100, 200, 300, 4000}; each child
     * continues with one token 7 (edge ids in BFS order; edge e leads to node e + 1) */
    const int32_t child_off[] = {0, 1, 2, 3, 7, 8, 9, 10, 11, 11, 11, 11, 11};
    const int32_t child_tok[] = {10, 11, 12, 100, 200, 300, 4000, 7, 7, 7, 7};
    const int32_t n_edges = 11, n_nodes = 12;
    const int32_t allowed[4] = {100, 200, 300, 4000};
    grx_trie* trie = NULL;
    grx_state* state = NULL;
    grx_rows rows;
    grx_constraint con;
    grx_beam_init_args init;
    grx_beam_step_args step;
    int64_t* h_ids = (int64_t*)malloc(sizeof(int64_t) * NR * NT);
    float* h_logits = (float*)malloc(sizeof(float) * NR * NV);
    const int64_t stride = grx_candidate_stride(NCMAX);
    grx_cand* h_cand = (grx_cand*)malloc(sizeof(grx_cand) * NR * (size_t)stride);
    void *d_ids, *d_logits, *d_cand, *d_ws;
    void* buf[2][7];
    void *d_unsat, *d_score, *d_parent, *d_token;
    float h_score[NB * 2 * NK];
    int32_t h_token[NB * 2 * NK], h_parent[NB * 2 * NK];
    const int L = NT + 4;
    const size_t ws_bytes = grx_workspace_bytes(NR);
    int r, v, i, h, b;
    int32_t st[4];

    CHECK(hipSetDevice(0) == 0, "hipSetDevice");
    CHECK(grx_trie_create(child_off, child_tok, n_nodes, n_edges, &trie) == GRX_OK, "grx_trie_create");
    CHECK(grx_trie_max_fanout(trie) == 4 && grx_trie_max_token(trie) == 4000, "trie summary");
    CHECK(grx_state_create(&state) == GRX_OK, "grx_state_create");

    for (r = 0; r < NR; ++r) {
        for (i = 0; i < 3; ++i) h_ids[r * NT + i] = 20 + (r / NK) + i;  /* one prompt per group */
        h_ids[r * NT + 3] = 10;
        h_ids[r * NT + 4] = 11;
        h_ids[r * NT + 5] = 12;
        for (v = 0; v < NV; ++v) h_logits[(size_t)r * NV + v] = (float)(8.0 * uniform01() - 4.0);
    }
    CHECK(hipMalloc(&d_ids, sizeof(int64_t) * NR * NT) == 0 && hipMalloc(&d_logits, sizeof(float) * NR * NV) == 0 &&
              hipMalloc(&d_cand, sizeof(grx_cand) * NR * (size_t)stride) == 0 && hipMalloc(&d_ws, ws_bytes) == 0,
          "hipMalloc");
    CHECK(hipMemcpy(d_ids, h_ids, sizeof(int64_t) * NR * NT, 1) == 0 &&
              hipMemcpy(d_logits, h_logits, sizeof(float) * NR * NV, 1) == 0 && hipMemset(d_ws, 0, ws_bytes) == 0,
          "upload");

    /* beam state, both halves of every double buffer */
    {
        const size_t sizes[7] = {sizeof(int64_t) * NR * L, sizeof(int64_t) * NR * L, sizeof(float) * NR, sizeof(float) * NR,
                                 sizeof(int32_t) * NR * (L - NT), sizeof(int32_t) * NR * (L - NT), NR};
        for (h = 0; h < 2; ++h)
            for (i = 0; i < 7; ++i) CHECK(hipMalloc(&buf[h][i], sizes[i]) == 0, "hipMalloc beam state");
    }
    CHECK(hipMalloc(&d_unsat, NB) == 0 && hipMalloc(&d_score, sizeof(float) * NB * 2 * NK) == 0 &&
              hipMalloc(&d_parent, sizeof(int32_t) * NB * 2 * NK) == 0 &&
              hipMalloc(&d_token, sizeof(int32_t) * NB * 2 * NK) == 0,
          "hipMalloc outputs");
    memset(&init, 0, sizeof(init));
    init.B = NB; init.K = NK; init.prompt_len = NT; init.max_length = L; init.fill_token = 1;
    init.input_ids = (const int64_t*)d_ids; init.ids_stride = NT;
    for (h = 0; h < 2; ++h) {
        init.run_seq[h] = (int64_t*)buf[h][0]; init.fin_seq[h] = (int64_t*)buf[h][1];
        init.run_score[h] = (float*)buf[h][2]; init.fin_score[h] = (float*)buf[h][3];
        init.run_bidx[h] = (int32_t*)buf[h][4]; init.fin_bidx[h] = (int32_t*)buf[h][5];
        init.fin_done[h] = (uint8_t*)buf[h][6];
    }
    init.unsat = (uint8_t*)d_unsat;
    CHECK(grx_beam_init(&init, NULL) == GRX_OK, "grx_beam_init");

    /* the processor call: count-based window of prefix_index + count = 3 + 0 tokens */
    memset(&rows, 0, sizeof(rows));
    rows.input_ids = (const int64_t*)d_ids; rows.ids_stride = NT; rows.R = NR; rows.T = NT;
    rows.scores = d_logits; rows.scores_stride = NV; rows.V = NV; rows.dtype = GRX_F32;
    memset(&con, 0, sizeof(con));
    con.mode = GRX_MODE_COUNT; con.prefix_index = 3; con.n_prefix = -1; con.eos_token_id = 1;
    con.repetition_penalty = 1.0f;
    CHECK(grx_state_set_count(state, 0, NULL) == GRX_OK, "grx_state_set_count");
    CHECK(grx_constrained_candidates(trie, state, &rows, &con, (grx_cand*)d_cand, stride, NCMAX, d_ws, ws_bytes, NULL) ==
              GRX_OK,
          "grx_constrained_candidates");
    CHECK(grx_state_read(state, st, NULL) == GRX_OK, "grx_state_read");
    CHECK(st[0] == 1 && st[1] == GRX_ROWERR_NONE && st[3] == 1, "count advanced once, no row error");
    CHECK(hipMemcpy(h_cand, d_cand, sizeof(grx_cand) * NR * (size_t)stride, 2) == 0, "download candidates");
    {
        double worst = 0.0;
        for (r = 0; r < NR; ++r) {
            const grx_cand* row = h_cand + (size_t)r * stride;
            double m = -1e300, s = 0.0, lse;
            float f;
            for (v = 0; v < NV; ++v) m = h_logits[(size_t)r * NV + v] > m ? h_logits[(size_t)r * NV + v] : m;
            for (v = 0; v < NV; ++v) s += exp((double)h_logits[(size_t)r * NV + v] - m);
            lse = m + log(s);
            CHECK(row[0].x == 4, "four allowed tokens per row");
            memcpy(&f, &row[0].y, 4);
            worst = fabs((double)f - lse) > worst ? fabs((double)f - lse) : worst;
            for (i = 0; i < 4; ++i) {
                CHECK(row[1 + i].x == allowed[i], "allowed token ids, ascending");
                memcpy(&f, &row[1 + i].y, 4);
                worst = fabs((double)f - ((double)h_logits[(size_t)r * NV + allowed[i]] - lse)) > worst
                            ? fabs((double)f - ((double)h_logits[(size_t)r * NV + allowed[i]] - lse))
                            : worst;
            }
        }
        printf("max_abs_logp_error=%.3g\n", worst);
        CHECK(worst <= 1e-5, "log-probs within 1e-5 of the float64 log-softmax");
    }

    /* the same call without a workspace: byte-identical candidate rows */
    {
        grx_cand* h_again = (grx_cand*)malloc(sizeof(grx_cand) * NR * (size_t)stride);
        void* d_again;
        CHECK(hipMalloc(&d_again, sizeof(grx_cand) * NR * (size_t)stride) == 0, "hipMalloc");
        CHECK(hipMemcpy(d_again, d_cand, sizeof(grx_cand) * NR * (size_t)stride, 3) == 0, "device copy");
        CHECK(hipMemset(d_again, 0xEE, sizeof(grx_cand) * 8) == 0, "scribble");
        CHECK(grx_state_set_count(state, 0, NULL) == GRX_OK, "grx_state_set_count");
        CHECK(grx_constrained_candidates_noalloc(trie, state, &rows, &con, (grx_cand*)d_again, stride, NCMAX, NULL) == GRX_OK,
              "grx_constrained_candidates_noalloc");
        CHECK(hipMemcpy(h_again, d_again, sizeof(grx_cand) * NR * (size_t)stride, 2) == 0, "download");
        for (r = 0; r < NR; ++r)
            CHECK(memcmp(h_again + (size_t)r * stride, h_cand + (size_t)r * stride, sizeof(grx_cand) * 16) == 0,
                  "the entry point that allocates nothing returns the same rows");
        printf("noalloc_rows_identical=%d\n", NR);

        /* and over two column slices of the logits, copied out as [NR, width] blocks */
        {
            const int cut = 2048;  /* tokens 100..300 in the first slice, 4000 in the second */
            const int widths[2] = {cut, NV - cut};
            float* h_slice = (float*)malloc(sizeof(float) * NR * (size_t)(NV - cut));
            void *d_slice, *d_part;
            grx_chunk ch;
            double worst = 0.0;
            int c;
            CHECK(hipMalloc(&d_slice, sizeof(float) * NR * (size_t)(NV - cut)) == 0 &&
                      hipMalloc(&d_part, sizeof(float) * 2 * NR * 8) == 0,
                  "hipMalloc slices");
            CHECK(hipMemset(d_again, 0xEE, sizeof(grx_cand) * NR * (size_t)stride) == 0, "scribble");
            CHECK(grx_state_set_count(state, 0, NULL) == GRX_OK, "grx_state_set_count");
            memset(&ch, 0, sizeof(ch));
            ch.v_total = NV; ch.n_slots = 0;
            for (c = 0; c < 2; ++c) ch.n_slots += grx_chunk_parts(NR, widths[c], GRX_F32);
            CHECK(ch.n_slots <= 8, "partial buffer sized for 8 slots");
            for (c = 0; c < 2; ++c) {
                grx_rows part = rows;
                const int begin = c == 0 ? 0 : cut;
                for (r = 0; r < NR; ++r)
                    memcpy(h_slice + (size_t)r * widths[c], h_logits + (size_t)r * NV + begin, sizeof(float) * widths[c]);
                CHECK(hipDeviceSynchronize() == 0, "the previous slice has been consumed");
                CHECK(hipMemcpy(d_slice, h_slice, sizeof(float) * NR * (size_t)widths[c], 1) == 0, "upload slice");
                part.scores = d_slice; part.scores_stride = widths[c]; part.V = widths[c];
                ch.v_begin = begin; ch.first = c == 0; ch.parts = grx_chunk_parts(NR, widths[c], GRX_F32);
                CHECK(grx_constrained_partial(trie, state, &part, &con, &ch, (float*)d_part, (grx_cand*)d_again, stride, NCMAX,
                                              d_ws, ws_bytes, NULL) == GRX_OK,
                      "grx_constrained_partial");
                ch.slot += ch.parts;
            }
            CHECK(grx_constrained_finish(NR, ch.n_slots, (const float*)d_part, (grx_cand*)d_again, stride, NCMAX, d_ws,
                                         ws_bytes, NULL) == GRX_OK,
                  "grx_constrained_finish");
            CHECK(grx_state_read(state, st, NULL) == GRX_OK && st[0] == 1 && st[1] == GRX_ROWERR_NONE,
                  "the sliced step counts as one processor call");
            CHECK(hipMemcpy(h_again, d_again, sizeof(grx_cand) * NR * (size_t)stride, 2) == 0, "download");
            for (r = 0; r < NR; ++r)
                for (i = 0; i < 16; ++i) {
                    const grx_cand a = h_again[(size_t)r * stride + i], w = h_cand[(size_t)r * stride + i];
                    float fa, fw;
                    memcpy(&fa, &a.y, 4);
                    memcpy(&fw, &w.y, 4);
                    CHECK(a.x == w.x, "sliced step: same counts, tokens and padding");
                    if (i <= 4) worst = fabs((double)fa - (double)fw) > worst ? fabs((double)fa - (double)fw) : worst;
                    else CHECK(a.y == 0, "zero padding");
                }
            printf("sliced_max_abs_logp_difference=%.3g\n", worst);
            CHECK(worst <= 1e-5, "sliced log-probs within 1e-5 of the whole-row call");
            hipFree(d_slice); hipFree(d_part);
            free(h_slice);
        }
        hipFree(d_again);
        free(h_again);
    }

    /* the dense form, into a new buffer and into one kept from call to call: the second kept call
     * (rows moved one level down the trie) must leave exactly what a fresh call leaves */
    {
        const size_t out_bytes = sizeof(float) * NR * (size_t)NV;
        float* h_fresh = (float*)malloc(out_bytes);
        float* h_kept = (float*)malloc(out_bytes);
        int64_t* h_ids2 = (int64_t*)malloc(sizeof(int64_t) * NR * (NT + 1));
        void *d_fresh, *d_kept, *d_last, *d_ids2;
        grx_rows deeper = rows;
        int call;
        CHECK(hipMalloc(&d_fresh, out_bytes) == 0 && hipMalloc(&d_kept, out_bytes) == 0 &&
                  hipMalloc(&d_last, grx_reuse_bytes(NR)) == 0 && hipMalloc(&d_ids2, sizeof(int64_t) * NR * (NT + 1)) == 0,
              "hipMalloc dense outputs");
        CHECK(hipMemset(d_kept, 0x5A, out_bytes) == 0, "scribble");
        for (r = 0; r < NR; ++r) {  /* the same rows one token further: the first allowed token appended */
            memcpy(h_ids2 + (size_t)r * (NT + 1), h_ids + (size_t)r * NT, sizeof(int64_t) * NT);
            h_ids2[(size_t)r * (NT + 1) + NT] = allowed[r % 4];
        }
        CHECK(hipMemcpy(d_ids2, h_ids2, sizeof(int64_t) * NR * (NT + 1), 1) == 0, "upload deeper rows");
        deeper.input_ids = (const int64_t*)d_ids2; deeper.ids_stride = NT + 1; deeper.T = NT + 1;
        for (call = 0; call < 2; ++call) {
            const grx_rows* now = call == 0 ? &rows : &deeper;
            size_t n_finite = 0;
            CHECK(grx_state_set_count(state, call, NULL) == GRX_OK, "grx_state_set_count");
            CHECK(grx_constrained_logsoftmax(trie, state, now, &con, d_fresh, NV, GRX_F32, d_ws, ws_bytes, NULL) == GRX_OK,
                  "grx_constrained_logsoftmax");
            CHECK(grx_state_set_count(state, call, NULL) == GRX_OK, "grx_state_set_count");
            CHECK(grx_constrained_logsoftmax_reuse(trie, state, now, &con, d_kept, NV, GRX_F32, d_last, call == 0, d_ws,
                                                   ws_bytes, NULL) == GRX_OK,
                  "grx_constrained_logsoftmax_reuse");
            CHECK(grx_state_read(state, st, NULL) == GRX_OK && st[1] == GRX_ROWERR_NONE, "no row error");
            CHECK(hipMemcpy(h_fresh, d_fresh, out_bytes, 2) == 0 && hipMemcpy(h_kept, d_kept, out_bytes, 2) == 0,
                  "download dense outputs");
            CHECK(memcmp(h_fresh, h_kept, out_bytes) == 0, "the kept output equals a fresh one, bit for bit");
            for (v = 0; v < NR * NV; ++v) n_finite += isfinite(h_kept[v]) ? 1 : 0;
            CHECK(n_finite > 0 && n_finite <= (size_t)NR * 4, "a few finite positions per row, -inf elsewhere");
            printf("dense_kept_call_%d_finite=%zu\n", call, n_finite);
        }
        CHECK(grx_constrained_logsoftmax_reuse(trie, state, &rows, &con, d_kept, NV, GRX_F32, NULL, 0, d_ws, ws_bytes,
                                               NULL) == GRX_E_INVALID,
              "the kept form needs its row record");
        hipFree(d_fresh); hipFree(d_kept); hipFree(d_last); hipFree(d_ids2);
        free(h_fresh); free(h_kept); free(h_ids2);
    }

    /* the beam step over those candidates: at step 0 only beam 0 of a group is live */
    memset(&step, 0, sizeof(step));
    step.B = NB; step.K = NK; step.V = NV; step.n_keep = 2 * NK; step.cmax = NCMAX;
    step.cur_len = NT; step.prompt_len = NT; step.max_length = L; step.copy_from = NT;
    step.eos_token_id = 1; step.early_stopping = 0; step.length_penalty = 1.0f;
    step.cand = (const grx_cand*)d_cand; step.cand_stride = stride;
    step.run_seq_in = (const int64_t*)buf[0][0]; step.run_seq_out = (int64_t*)buf[1][0];
    step.fin_seq_in = (const int64_t*)buf[0][1]; step.fin_seq_out = (int64_t*)buf[1][1];
    step.run_score_in = (const float*)buf[0][2]; step.run_score_out = (float*)buf[1][2];
    step.fin_score_in = (const float*)buf[0][3]; step.fin_score_out = (float*)buf[1][3];
    step.run_bidx_in = (const int32_t*)buf[0][4]; step.run_bidx_out = (int32_t*)buf[1][4];
    step.fin_bidx_in = (const int32_t*)buf[0][5]; step.fin_bidx_out = (int32_t*)buf[1][5];
    step.fin_done_in = (const uint8_t*)buf[0][6]; step.fin_done_out = (uint8_t*)buf[1][6];
    step.unsat = (uint8_t*)d_unsat;
    step.topk_score = (float*)d_score; step.topk_parent = (int32_t*)d_parent; step.topk_token = (int32_t*)d_token;
    CHECK(grx_beam_step(&step, NULL) == GRX_OK, "grx_beam_step");
    CHECK(hipDeviceSynchronize() == 0, "hipDeviceSynchronize");
    CHECK(hipMemcpy(h_score, d_score, sizeof(h_score), 2) == 0 && hipMemcpy(h_token, d_token, sizeof(h_token), 2) == 0 &&
              hipMemcpy(h_parent, d_parent, sizeof(h_parent), 2) == 0,
          "download beam step");
    for (b = 0; b < NB; ++b) {
        /* the four live candidates are beam 0's, ordered by log-prob descending */
        const grx_cand* row = h_cand + (size_t)(b * NK) * stride;
        int order[4] = {0, 1, 2, 3}, j;
        for (i = 0; i < 4; ++i)
            for (j = i + 1; j < 4; ++j) {
                float fi, fj;
                memcpy(&fi, &row[1 + order[i]].y, 4);
                memcpy(&fj, &row[1 + order[j]].y, 4);
                if (fj > fi) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
            }
        for (i = 0; i < 4; ++i) {
            float f;
            memcpy(&f, &row[1 + order[i]].y, 4);
            CHECK(h_parent[b * 2 * NK + i] == 0 && h_token[b * 2 * NK + i] == row[1 + order[i]].x &&
                      h_score[b * 2 * NK + i] == f,
                  "top candidates of a group = beam 0's allowed tokens by log-prob");
        }
    }
    printf("beam_step_groups_checked=%d\n", NB);

    /* the same step as ONE call (grx_constrained_beam_step), the processor call planned ahead by the beam
     * initialisation so that the streaming kernel itself carries the groups' beam steps: same candidate
     * rows, same top candidates, bit for bit */
    {
        grx_plan_request req;
        void *d_score2, *d_parent2, *d_token2;
        float h_score2[NB * 2 * NK];
        int32_t h_token2[NB * 2 * NK], h_parent2[NB * 2 * NK];
        grx_cand* h_again = (grx_cand*)malloc(sizeof(grx_cand) * NR * (size_t)stride);
        CHECK(hipMalloc(&d_score2, sizeof(h_score2)) == 0 && hipMalloc(&d_parent2, sizeof(h_parent2)) == 0 &&
                  hipMalloc(&d_token2, sizeof(h_token2)) == 0,
              "hipMalloc");
        memset(&req, 0, sizeof(req));
        req.trie = trie; req.state = state; req.con = &con; req.workspace = d_ws; req.workspace_bytes = ws_bytes;
        CHECK(grx_state_set_count(state, 0, NULL) == GRX_OK, "grx_state_set_count");
        init.plan_next = &req;
        CHECK(grx_beam_init(&init, NULL) == GRX_OK, "grx_beam_init with a plan request");
        init.plan_next = NULL;
        CHECK(hipMemset(d_cand, 0xEE, sizeof(grx_cand) * NR * (size_t)stride) == 0, "scribble");
        rows.planned = 1;
        step.topk_score = (float*)d_score2; step.topk_parent = (int32_t*)d_parent2; step.topk_token = (int32_t*)d_token2;
        CHECK(grx_constrained_beam_step(trie, state, &rows, &con, (grx_cand*)d_cand, stride, NCMAX, d_ws, ws_bytes, &step,
                                        NULL) == GRX_OK,
              "grx_constrained_beam_step");
        rows.planned = 0;
        CHECK(grx_state_read(state, st, NULL) == GRX_OK && st[0] == 1 && st[1] == GRX_ROWERR_NONE,
              "the one-call step counts as one processor call");
        CHECK(hipMemcpy(h_score2, d_score2, sizeof(h_score2), 2) == 0 && hipMemcpy(h_token2, d_token2, sizeof(h_token2), 2) == 0 &&
                  hipMemcpy(h_parent2, d_parent2, sizeof(h_parent2), 2) == 0 &&
                  hipMemcpy(h_again, d_cand, sizeof(grx_cand) * NR * (size_t)stride, 2) == 0,
              "download");
        for (r = 0; r < NR; ++r)
            CHECK(memcmp(h_again + (size_t)r * stride, h_cand + (size_t)r * stride, sizeof(grx_cand) * 16) == 0,
                  "one-call step: the candidate rows of the processor call");
        for (b = 0; b < NB; ++b)
            for (i = 0; i < 4; ++i)  /* the live slots (the rest are -1e9 placeholders) */
                CHECK(h_score2[b * 2 * NK + i] == h_score[b * 2 * NK + i] && h_token2[b * 2 * NK + i] == h_token[b * 2 * NK + i] &&
                          h_parent2[b * 2 * NK + i] == h_parent[b * 2 * NK + i],
                      "one-call step: the top candidates of the beam step");
        step.cmax = NCMAX + 1;
        CHECK(grx_constrained_beam_step(trie, state, &rows, &con, (grx_cand*)d_cand, stride, NCMAX, d_ws, ws_bytes, &step,
                                        NULL) == GRX_E_INVALID,
              "a beam step that does not read this call's candidate buffer is refused");
        step.cmax = NCMAX;
        printf("one_call_step_checked=1\n");
        hipFree(d_score2); hipFree(d_parent2); hipFree(d_token2);
        free(h_again);
    }

    for (h = 0; h < 2; ++h)
        for (i = 0; i < 7; ++i) hipFree(buf[h][i]);
    hipFree(d_unsat); hipFree(d_score); hipFree(d_parent); hipFree(d_token);
    hipFree(d_ids); hipFree(d_logits); hipFree(d_cand); hipFree(d_ws);
    CHECK(grx_state_destroy(state) == GRX_OK && grx_trie_destroy(trie) == GRX_OK, "destroy");
    free(h_ids); free(h_logits); free(h_cand);
    return 0;
}

int main(int argc, char** argv) {
    int rc;
    if (argc != 2 || (strcmp(argv[1], "host") != 0 && strcmp(argv[1], "gpu") != 0)) {
        fprintf(stderr, "usage: %s host|gpu\n", argv[0]);
        return 2;
    }
    rc = host_checks();
    if (rc == 0 && strcmp(argv[1], "gpu") == 0) rc = gpu_checks();
    if (rc == 0) printf("OK\n");
    return rc;
}
