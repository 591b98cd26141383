// C ABI of libsnkhip.so, part 6: the waveform-side gather, timers, options, introspection and self tests.
#include "snk_engine.h"

// ---------------------------------------------------------------------------
// waveform-side gather
// ---------------------------------------------------------------------------
int snk_upload_frames(snk_handle h, const float *spec, const double *fzv, int64_t rows, int H)
{
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    if (!spec || !fzv || rows < 1 || H < 1) return fail("snk_upload_frames: null/empty argument");
    const size_t W = (size_t)3 * H;
    CHK(h->frames_spec.ensure((size_t)rows * W * sizeof(float)));
    CHK(h->frames_fzv.ensure((size_t)rows * 2 * sizeof(double)));
    CHK(h2d(h, h->frames_spec.p, spec, (size_t)rows * W * sizeof(float), h->stream));
    CHK(h2d(h, h->frames_fzv.p, fzv, (size_t)rows * 2 * sizeof(double), h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->frames_rows = rows;
    h->frames_W = (int)W;
    return 0;
}

int snk_concat_fragments(snk_handle h, const int64_t *first_row, const int64_t *utt_lo, const int64_t *utt_hi,
                         int64_t n, int multiepoch, int overlap, const double *in_taper,
                         double *spec_out, double *fz_out)
{
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    if (h->frames_rows < 1) return fail("snk_concat_fragments: no analysis frames uploaded (snk_upload_frames)");
    if (!first_row || !utt_lo || !utt_hi || !spec_out || !fz_out || n < 1) return fail("snk_concat_fragments: null/empty argument");
    if (multiepoch < 1) return fail("snk_concat_fragments: multiepoch < 1");
    if (overlap < 0 || (overlap % 2) != 0) return fail("snk_concat_fragments: frame overlap should be even number");
    if (overlap > 0 && !in_taper) return fail("snk_concat_fragments: null taper");
    if (overlap > 0 && 2 * overlap > multiepoch + overlap)
        return fail("snk_concat_fragments: taper_length (%d) too long for (padded) unit length (%d)", overlap, multiepoch + overlap);
    for (int64_t k = 0; k < n; ++k) {
        if (utt_lo[k] < 0 || utt_hi[k] > h->frames_rows || utt_lo[k] >= utt_hi[k] || first_row[k] < utt_lo[k] ||
            first_row[k] >= utt_hi[k])
            return fail("snk_concat_fragments: unit %lld lies outside its utterance / the uploaded frames", (long long)k);
        // the reference slices silently short (and asserts) when a window runs past its utterance without overlap
        if (overlap == 0 && first_row[k] + multiepoch > utt_hi[k])
            return fail("snk_concat_fragments: unit %lld runs past the end of its utterance (needs overlap > 0)", (long long)k);
    }
    const int64_t rows_out = n * multiepoch;
    const size_t W = (size_t)h->frames_W;
    const size_t in_bytes = (size_t)n * 3 * sizeof(int64_t) + (size_t)(overlap > 0 ? overlap : 1) * sizeof(double);
    CHK(h->cc_in.ensure(in_bytes));
    CHK(h->cc_out.ensure((size_t)rows_out * (W + 1) * sizeof(double)));
    int64_t *d_first = h->cc_in.as<int64_t>(), *d_lo = d_first + n, *d_hi = d_lo + n;
    double *d_taper = reinterpret_cast<double *>(d_hi + n);
    CHK(h2d(h, d_first, first_row, (size_t)n * sizeof(int64_t), h->stream));
    CHK(h2d(h, d_lo, utt_lo, (size_t)n * sizeof(int64_t), h->stream));
    CHK(h2d(h, d_hi, utt_hi, (size_t)n * sizeof(int64_t), h->stream));
    if (overlap > 0) CHK(h2d(h, d_taper, in_taper, (size_t)overlap * sizeof(double), h->stream));
    double *d_spec = h->cc_out.as<double>(), *d_fz = d_spec + (size_t)rows_out * W;
    launch_concat_fragments(h->frames_spec.as<float>(), (int)W, h->frames_fzv.as<double>(), d_first, d_lo, d_hi, n,
                            multiepoch, overlap, d_taper, d_spec, d_fz, h->stream);
    HIPCHK(hipGetLastError());
    {
        D2HPart parts[2] = {{spec_out, d_spec, (size_t)rows_out * W * sizeof(double)}, {fz_out, d_fz, (size_t)rows_out * sizeof(double)}};
        CHK(staged_d2h(h, h->stream, parts, 2));
    }
    return 0;
}

// ---------------------------------------------------------------------------
// timers / options / self test
// ---------------------------------------------------------------------------

const char *kTimerNames[TM_COUNT] = {
    "h2d_queries", "prepare_queries", "knn_minima", "knn_threshold", "knn_filter", "knn_bucket", "knn_finalize",
    "join_costs", "viterbi_dp", "d2h_results", "greedy_target_gemm", "greedy_steps", "set_weights",
    "merge_topk", "join_lower_bounds", "viterbi_lower_bound", "join_exact_sparse", "viterbi_sparse", "knn_ball_bound"};

int snk_timer_count(void) { return TM_COUNT; }
const char *snk_timer_name(int i) { return (i >= 0 && i < TM_COUNT) ? kTimerNames[i] : ""; }

int snk_get_timers(snk_handle h, double *ms_out, int capacity)
{
    if (!h || !ms_out) return -1;
    // layout: [total_ms x TM_COUNT][launch count x TM_COUNT]
    int n = 0;
    for (int i = 0; i < TM_COUNT && n < capacity; ++i) ms_out[n++] = h->tm_ms[i];
    for (int i = 0; i < TM_COUNT && n < capacity; ++i) ms_out[n++] = (double)h->tm_n[i];
    return n;
}

int snk_reset_timers(snk_handle h)
{
    if (!h) return fail("null handle");
    for (int i = 0; i < TM_COUNT; ++i) { h->tm_ms[i] = 0; h->tm_n[i] = 0; }
    h->greedy_bound_violations = 0; h->greedy_bound_max_used = 0.0;
    if (h->margin_stat.p && !any_batch_busy(h) && !h->sticket[0].busy && !h->sticket[1].busy) {
        const unsigned int init[8] = {0u, 0x7f800000u, 0u, 0u, 0u, 0u, (unsigned int)h->roofline_counters, 0u};
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipStreamSynchronize(h->stream));
        CHK(h2d_sync(h, h->margin_stat.p, init, sizeof(init)));
        if (h->vstats.p) {                                     // ... and the tripwire of the join bounds (joinfast_kernels.hip: stats[4], [5])
            for (int i = 0; i < 2; ++i) HIPCHK(hipStreamSynchronize(h->dp_stream[i]));
            const unsigned long long init2[4] = {0ull, 0xffffffffull, 0ull, 0ull};      // ... and pass 3's counters ([6], [7])
            CHK(h2d_sync(h, reinterpret_cast<char *>(h->vstats.p) + 4 * sizeof(unsigned long long), init2, sizeof(init2)));
        }
    }
    return 0;
}

int snk_set_option(snk_handle h, const char *name, double value)
{
    if (!h || !name) return fail("snk_set_option: null argument");
    if (!strcmp(name, "list_capacity")) {
        if (value < 64 || value > 8192) return fail("list_capacity must be in 64..8192");
        h->cap = (int)value;
    } else if (!strcmp(name, "sample_fraction")) {
        if (!(value > 0.0 && value <= 1.0)) return fail("sample_fraction must be in (0,1]");
        h->sample_frac = value;
    } else if (!strcmp(name, "min_sample_slabs")) {
        if (!(value >= 1.0 && value <= 65536.0)) return fail("min_sample_slabs must be in 1..65536");
        h->min_sample_slabs = (int)value;
    } else if (!strcmp(name, "db_tiles_per_wave")) {
        h->nt_override = (int)value;
    } else if (!strcmp(name, "f32_tiles_per_wave")) {
        if (value != 2.0 && value != 4.0 && value != 8.0) return fail("f32_tiles_per_wave must be 2, 4 or 8");
        h->nt16 = (int)value;
        h->have_weights = false;          // operands are laid out per slab: set_weights must be called again
    } else if (!strcmp(name, "precision")) {
        if (value != 0.0 && value != 1.0) return fail("precision must be 0 (f64 sweep) or 1 (f32 prefilter + exact f64 re-rank)");
        h->precision = (int)value;
    } else if (!strcmp(name, "prefilter")) {
        if (value != 0.0 && value != 1.0 && value != 2.0) return fail("prefilter must be 0 (float32 operands), 1 or 2 (bf16-split operands where the shape has a variant: 3 / 4 MFMA terms per product)");
        CHK(no_batch_in_flight(h, "snk_set_option(prefilter)"));
        h->prefilter = (int)value;
        h->have_weights = false;          // the bf16 operands are built by set_weights
    } else if (!strcmp(name, "prefilter_balls")) {
        if (value != 0.0 && value != 1.0) return fail("prefilter_balls must be 0 or 1");
        CHK(no_batch_in_flight(h, "snk_set_option(prefilter_balls)"));
        h->prefilter_balls = (int)value;
        h->have_weights = false;          // the ball operand is built by set_weights
    } else if (!strcmp(name, "prefilter_super_balls")) {
        if (value != 0.0 && value != 1.0) return fail("prefilter_super_balls must be 0 or 1");
        h->prefilter_super_balls = (int)value;
    } else if (!strcmp(name, "prefilter_ball_bound")) {
        if (value != 0.0 && value != 1.0) return fail("prefilter_ball_bound must be 0 or 1");
        h->prefilter_ball_bound = (int)value;
    } else if (!strcmp(name, "coarse_gate_fraction")) {
        if (!(value >= 0.0 && value <= 1.0)) return fail("coarse_gate_fraction must be in 0..1");
        h->coarse_gate_fraction = value;
    } else if (!strcmp(name, "prefilter_two_pass")) {
        if (value != 0.0 && value != 1.0) return fail("prefilter_two_pass must be 0 or 1");
        CHK(no_batch_in_flight(h, "snk_set_option(prefilter_two_pass)"));
        h->prefilter_two_pass = (int)value;
    } else if (!strcmp(name, "reserved_cus")) {
        if (value < 0 || value > 64) return fail("reserved_cus must be in 0..64");
        h->reserved_cus = (int)value;
    } else if (!strcmp(name, "batch_rows")) {
        if (value < 0 || value > SNK_KNN_MAX_ROWS) return fail("batch_rows must be in 0..%d (0: one K-NN call per utterance)", (int)SNK_KNN_MAX_ROWS);
        h->batch_rows = (int)value;
    } else if (!strcmp(name, "greedy_f16")) {
        if (value != 0.0 && value != 1.0 && value != 2.0) return fail("greedy_f16 must be 0, 1 (streamed databases) or 2 (always)");
        h->greedy_f16 = (int)value;
    } else if (!strcmp(name, "greedy_test_stall")) {
        h->greedy_test_stall = value != 0.0;
    } else if (!strcmp(name, "greedy_fenced")) {
        h->greedy_fenced = value != 0.0;
    } else if (!strcmp(name, "greedy_resident")) {
        if (value != 0.0 && value != 1.0) return fail("greedy_resident must be 0 or 1");
        h->greedy_resident = (int)value;
    } else if (!strcmp(name, "greedy_hoist")) {
        if (value != 0.0 && value != 1.0) return fail("greedy_hoist must be 0 or 1");
        h->greedy_hoist = (int)value;
    } else if (!strcmp(name, "greedy_speculate")) {
        if (value != 0.0 && value != 1.0) return fail("greedy_speculate must be 0 or 1");
        h->greedy_speculate = (int)value;
    } else if (!strcmp(name, "greedy_hoist_fast")) {
        if (value != 0.0 && value != 1.0) return fail("greedy_hoist_fast must be 0 or 1");
        h->greedy_hoist_fast = (int)value;
    } else if (!strcmp(name, "greedy_hoist_max_gb")) {
        if (!(value >= 0.0)) return fail("greedy_hoist_max_gb must be >= 0");
        h->greedy_hoist_max_gb = value;
    } else if (!strcmp(name, "greedy_mode")) {
        if (value != 0.0 && value != 1.0 && value != 2.0) return fail("greedy_mode must be 0 (exact scan, a launch per step), 1 (float32 prefilter scan, one launch) or 2 (auto)");
        h->greedy_mode = (int)value;
    } else if (!strcmp(name, "viterbi_mode")) {
        if (value != 0.0 && value != 1.0 && value != 2.0) return fail("viterbi_mode must be 0 (dense exact join + recursion), 1 (lower bounds + sparse exact recursion) or 2 (auto)");
        CHK(no_batch_in_flight(h, "snk_set_option(viterbi_mode)"));
        h->viterbi_mode = (int)value;
    } else if (!strcmp(name, "reorder")) {
        if (value != 0.0 && value != 1.0) return fail("reorder must be 0 or 1");
        CHK(no_batch_in_flight(h, "snk_set_option(reorder)"));
        h->reorder = (int)value;
        if (!h->reorder && h->perm_ready) { h->perm_ready = false; h->have_weights = false; }     // back to the database order with the next snk_set_weights
    } else if (!strcmp(name, "reorder_now")) {
        h->reorder_pending = value != 0.0; h->reorder_done = false; h->reorder_useless = false;      // developer aid: cluster at the next K-NN call
    } else if (!strcmp(name, "reorder_iterations")) {
        if (!(value >= 1.0 && value <= 64.0)) return fail("reorder_iterations must be in 1..64");
        h->reorder_iters = (int)value;
    } else if (!strcmp(name, "viterbi_latch") || !strcmp(name, "latch_rearm")) {
        if (value != 0.0 && value != 1.0) return fail("%s must be 0 or 1", name);
        CHK(no_batch_in_flight(h, "snk_set_option(latch)"));
        if (name[0] == 'v') { h->viterbi_latch = (int)value; h->vit = snk_engine::VitLatch(); }
        else h->latch_rearm = (int)value;
    } else if (!strcmp(name, "viterbi_refine_gate")) {
        if (!(value >= 0.0 && value <= 1.0)) return fail("viterbi_refine_gate must be in 0..1");
        h->vit_refine_gate = value;
    } else if (!strcmp(name, "viterbi_fst32_slack")) {
        if (!(value >= 0.0 && value <= 1e-3)) return fail("viterbi_fst32_slack must be in 0..1e-3");
        h->fst32_slack = value;
    } else if (!strcmp(name, "viterbi_lb_chunk") || !strcmp(name, "viterbi_lb_warm") || !strcmp(name, "viterbi_lb_chunk_max_utts")) {
        if (!(value >= 0.0 && value <= 1e6) || value != (double)(int)value) return fail("%s must be a small non-negative integer", name);
        if (!strcmp(name, "viterbi_lb_warm") && value < 1.0) return fail("viterbi_lb_warm must be >= 1");
        CHK(no_batch_in_flight(h, "snk_set_option(viterbi_lb_*)"));
        (!strcmp(name, "viterbi_lb_chunk") ? h->lb_chunk : !strcmp(name, "viterbi_lb_warm") ? h->lb_warm : h->lb_chunk_max_utts) = (int)value;
    } else if (!strcmp(name, "viterbi_sparse_waves")) {
        // process-wide (a debugging / A-B switch): which form of the sparse exact recursion runs; same results
        if (value != 1.0 && value != 4.0) return fail("viterbi_sparse_waves must be 1 (one compute wavefront per utterance) or 4");
        CHK(no_batch_in_flight(h, "snk_set_option(viterbi_sparse_waves)"));
        set_viterbi_sparse_waves((int)value);
    } else if (!strcmp(name, "shard_gather_queries")) {
        if (value != 0.0 && value != 1.0) return fail("shard_gather_queries must be 0 or 1 (the same on every rank)");
        h->shard_gather_queries = (int)value;
    } else if (!strcmp(name, "shard_compact")) {
        if (value != 0.0 && value != 1.0) return fail("shard_compact must be 0 or 1 (the same on every rank)");
        h->shard_compact = (int)value;
    } else if (!strcmp(name, "shard_refine")) {
        if (value != 0.0 && value != 1.0) return fail("shard_refine must be 0 or 1 (the same on every rank)");
        h->shard_refine = (int)value;
    } else if (!strcmp(name, "join_bounds_stream")) {
        if (value != 0.0 && value != 1.0) return fail("join_bounds_stream must be 0 (main stream) or 1 (side stream of the group)");
        CHK(no_batch_in_flight(h, "snk_set_option(join_bounds_stream)"));
        h->join_bounds_stream = (int)value;
    } else if (!strcmp(name, "join_bounds_delay")) {
        if (!(value >= 0.0 && value <= 5.0) || value != (double)(int)value) return fail("join_bounds_delay must be 0 .. 5");
        CHK(no_batch_in_flight(h, "snk_set_option(join_bounds_delay)"));
        h->join_bounds_delay = (int)value;
    } else if (!strcmp(name, "tau_optimism")) {
        if (value != 0.0 && value != 1.0) return fail("tau_optimism must be 0 or 1");
        CHK(no_batch_in_flight(h, "snk_set_option(tau_optimism)"));
        h->tau_optimism = (int)value;
        h->opt_off = false; h->opt_calls = 0; h->opt_fails = 0;
    } else if (!strcmp(name, "tau_optimism_rank")) {
        if (!(value >= 0.0 && value <= 4096.0) || value != (double)(int)value) return fail("tau_optimism_rank must be 0 (automatic) .. 4096");
        CHK(no_batch_in_flight(h, "snk_set_option(tau_optimism_rank)"));
        h->tau_rank_override = (int)value;
        h->opt_off = false; h->opt_calls = 0; h->opt_fails = 0;
    } else if (!strcmp(name, "roofline_counters")) {
        if (value != 0.0 && value != 1.0) return fail("roofline_counters must be 0 or 1");
        CHK(no_batch_in_flight(h, "snk_set_option(roofline_counters)"));
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipDeviceSynchronize());
        h->roofline_counters = (int)value;
        const unsigned int on = (unsigned int)h->roofline_counters;
        CHK(h2d_sync(h, reinterpret_cast<char *>(h->margin_stat.p) + 6 * sizeof(unsigned int), &on, sizeof(on)));
        if (!h->vstats.p) {
            CHK(h->vstats.ensure((128 + 16 * 1024) * sizeof(unsigned long long)));
            HIPCHK(hipMemset(h->vstats.p, 0, 128 * sizeof(unsigned long long)));
            HIPCHK(hipMemset(reinterpret_cast<char *>(h->vstats.p) + 5 * sizeof(unsigned long long), 0xff, 4));
        }
        const unsigned long long on64 = on;
        CHK(h2d_sync(h, reinterpret_cast<char *>(h->vstats.p) + 8 * sizeof(unsigned long long), &on64, sizeof(on64)));
    } else if (!strcmp(name, "tail_defer")) {
        if (value != 0.0 && value != 1.0 && value != 2.0) return fail("tail_defer must be 0 (never), 1 (always) or 2 (while the host keeps up)");
        CHK(no_batch_in_flight(h, "snk_set_option(tail_defer)"));
        h->tail_defer = (int)value; h->starved_ema = 0.0;
    } else if (!strcmp(name, "results_by_kernel")) {
        if (value != 0.0 && value != 1.0) return fail("results_by_kernel must be 0 or 1");
        CHK(no_batch_in_flight(h, "snk_set_option(results_by_kernel)"));
        h->results_by_kernel = (int)value;
    } else if (!strcmp(name, "upload_stream")) {
        if (value != 0.0 && value != 1.0) return fail("upload_stream must be 0 or 1");
        CHK(no_batch_in_flight(h, "snk_set_option(upload_stream)"));
        h->upload_stream = (int)value;
    } else if (!strcmp(name, "wide_one_group")) {
        if (value != 0.0 && value != 1.0) return fail("wide_one_group must be 0 or 1");
        CHK(no_batch_in_flight(h, "snk_set_option(wide_one_group)"));
        h->wide_one_group = (int)value;
    } else if (!strcmp(name, "split_one_group")) {
        if (value != 0.0 && value != 1.0) return fail("split_one_group must be 0 or 1");
        CHK(no_batch_in_flight(h, "snk_set_option(split_one_group)"));
        h->split_one_group = (int)value;
    } else if (!strcmp(name, "join_lb_quadrants")) {
        if (value != 0.0 && value != 1.0) return fail("join_lb_quadrants must be 0 or 1");
        CHK(no_batch_in_flight(h, "snk_set_option(join_lb_quadrants)"));
        set_join_lb_quadrants((int)value);
    } else if (!strcmp(name, "join_lb_one_set")) {
        if (value != 0.0 && value != 1.0) return fail("join_lb_one_set must be 0 or 1");
        CHK(no_batch_in_flight(h, "snk_set_option(join_lb_one_set)"));
        set_join_lb_one_set((int)value);
    } else if (!strcmp(name, "join_exact_form")) {
        if (value != 0.0 && value != 1.0) return fail("join_exact_form must be 0 or 1");
        CHK(no_batch_in_flight(h, "snk_set_option(join_exact_form)"));
        set_join_exact_form((int)value);
    } else if (!strcmp(name, "viterbi_weights")) {
        if (value != 0.0 && value != 1.0) return fail("viterbi_weights must be 0 (float64) or 1 (OpenFST's float32 weights)");
        CHK(no_batch_in_flight(h, "snk_set_option(viterbi_weights)"));
        h->viterbi_weights = (int)value;
    } else if (!strcmp(name, "join_lb_variant")) {
        if (value != 0.0 && value != 1.0) return fail("join_lb_variant must be 0 or 1");
        h->join_lb_variant = (int)value;
    } else if (!strcmp(name, "join_lb_test_scale")) {
        if (!(value >= 0.0 && value <= 100.0)) return fail("join_lb_test_scale must be in 0..100");
        h->join_lb_test_scale = value;
    } else if (!strcmp(name, "join_beta")) {
        if (!(value >= 0.0 && value <= 10.0)) return fail("join_beta must be in 0..10");
        h->join_beta = value;
    } else if (!strcmp(name, "pool_chunk_limit")) {
        if (value < 0 || value > 1e6) return fail("pool_chunk_limit must be in 0..1e6");
        h->pool_chunk_limit = (int)value;
    } else if (!strcmp(name, "timers_mask")) {
        if (!(value >= 0.0 && value < 4294967296.0)) return fail("timers_mask must be a 32-bit mask of stage ids");
        h->timers_mask = (unsigned int)value;
    } else if (!strcmp(name, "timers")) {
        if (value != 0.0 && value != 1.0 && value != 2.0 && value != 3.0) return fail("timers must be 0 (none), 1 (every stage), 2 (the Viterbi side's bounds pass only) or 3 (the stages of timers_mask)");
        h->timers_on = (int)value;
    } else {
        return fail("snk_set_option: unknown option '%s'", name);
    }
    return 0;
}

int snk_get_info(snk_handle h, const char *name, double *out)
{
    if (!h || !name || !out) return fail("snk_get_info: null argument");
    if (!strcmp(name, "n_units")) *out = (double)h->N;
    else if (!strcmp(name, "target_dim")) *out = h->Dt;
    else if (!strcmp(name, "join_dim")) *out = h->Dj;
    else if (!strcmp(name, "last_knn_retries")) *out = h->last_retries;
    else if (!strcmp(name, "list_capacity")) *out = h->cap;
    else if (!strcmp(name, "tie_overflow")) *out = h->tie_overflow;
    else if (!strcmp(name, "exact_row_fallbacks")) *out = h->exact_row_fallbacks;
    else if (!strcmp(name, "batch_redos")) *out = h->batch_redos;
    else if (!strcmp(name, "pool_overflows")) *out = h->pool_overflows;
    else if (!strcmp(name, "viterbi_mode")) *out = h->viterbi_mode;
    else if (!strcmp(name, "join_lb_variant")) *out = h->join_lb_variant;
    else if (!strcmp(name, "viterbi_weights")) *out = h->viterbi_weights;
    else if (!strcmp(name, "greedy_mode")) *out = h->greedy_mode;
    else if (!strcmp(name, "greedy_hoist")) *out = h->greedy_hoist;
    else if (!strcmp(name, "greedy_hoist_launches")) *out = (double)h->greedy_hoist_launches;
    else if (!strcmp(name, "greedy_hoist_fast")) *out = h->greedy_hoist_fast;
    else if (!strcmp(name, "greedy_speculate")) *out = h->greedy_speculate;
    else if (!strcmp(name, "greedy_hoist16_launches")) *out = (double)h->greedy_hoist16_launches;
    else if (!strcmp(name, "greedy_last_undecided_step")) *out = (double)h->greedy_last_status[0] - 1.0;     // -1: every step was decided
    else if (!strcmp(name, "greedy_last_watchdog")) *out = (double)h->greedy_last_status[3];
    // (words 4 .. 7 belong to the kernel that served the last launch: the streamed scan's counters read 0 after a resident launch,
    // the resident scan's `why` words 0 / NaN after a streamed one)
    else if (!strcmp(name, "greedy_last_kernel")) *out = h->greedy_last_kernel;                       // 1 streamed scan, 2 resident scan, 0 none yet
    else if (!strcmp(name, "greedy_last_speculated")) *out = h->greedy_last_kernel == 1 ? (double)h->greedy_last_status[4] : 0.0;         // streamed scan: steps decided before the gather
    else if (!strcmp(name, "greedy_last_several_holders")) *out = h->greedy_last_kernel == 1 ? (double)h->greedy_last_status[5] : 0.0;   // streamed scan: steps with windows inside the bound in several workgroups
    else if (!strcmp(name, "greedy_last_why_candidates")) *out = h->greedy_last_kernel == 2 ? (double)h->greedy_last_status[4] : 0.0;
    else if (!strcmp(name, "greedy_last_why_third")) *out = h->greedy_last_kernel == 2 ? (double)h->greedy_last_status[5] : 0.0;
    else if (!strcmp(name, "greedy_last_why_min")) { double v; memcpy(&v, &h->greedy_last_status[6], 8); *out = h->greedy_last_kernel == 2 ? v : (double)NAN; }
    else if (!strcmp(name, "greedy_last_why_tau")) { double v; memcpy(&v, &h->greedy_last_status[7], 8); *out = h->greedy_last_kernel == 2 ? v : (double)NAN; }
    else if (!strcmp(name, "greedy_bound_violations")) *out = (double)h->greedy_bound_violations;     // tripwire of the float32 scans' bound (snk_engine.h)
    else if (!strcmp(name, "greedy_bound_max_used")) *out = h->greedy_bound_max_used;
    else if (!strcmp(name, "greedy_resident")) *out = h->greedy_resident;
    else if (!strcmp(name, "greedy_resident_launches")) *out = (double)h->greedy_resident_launches;
    else if (!strcmp(name, "greedy_f16")) *out = h->greedy_f16;
    else if (!strcmp(name, "greedy_f16_launches")) *out = (double)h->greedy_f16_launches;
    else if (!strcmp(name, "greedy_f16_delta")) *out = h->g16_delta;
    else if (!strcmp(name, "greedy_exact_windows")) *out = (double)h->greedy_exact_windows;
    else if (!strcmp(name, "greedy_second_rounds")) *out = (double)h->greedy_second_rounds;
    else if (!strcmp(name, "greedy_stalls")) *out = (double)h->greedy_stalls;
    else if (!strcmp(name, "greedy_fallbacks")) *out = h->greedy_fallbacks;
    else if (!strcmp(name, "greedy_second_phase_rounds") || !strcmp(name, "greedy_exact_windows")) {
        // statistics of the most recent float32 scan launch: steps that needed every lane's candidates; windows
        // whose canonical float64 totals decided a step
        int64_t v[3] = {0, 0, 0};
        if (h->g32_ctl.p) CHK(d2h_sync(h, v, reinterpret_cast<char *>(h->g32_ctl.p) + 16, sizeof(v), h->stream));
        *out = (double)v[!strcmp(name, "greedy_exact_windows") ? 2 : 1];
    }
    else if (!strcmp(name, "dense_cells") || !strcmp(name, "dense_steps") || !strcmp(name, "dense_exact_costs") || !strcmp(name, "set_overflows")) {
        unsigned long long v[4] = {0, 0, 0, 0};
        if (h->vstats.p) { HIPCHK(hipDeviceSynchronize()); CHK(d2h_sync(h, v, h->vstats.p, sizeof(v), h->stream)); }
        *out = (double)v[!strcmp(name, "dense_steps") ? 1 : (!strcmp(name, "dense_exact_costs") ? 2 : (!strcmp(name, "set_overflows") ? 3 : 0))];
    }
    else if (!strcmp(name, "join_bound_violations") || !strcmp(name, "join_bound_min_margin")) {
        // tripwire of pass 1's bounds (joinfast_kernels.hip), since the engine was created or the last snk_reset_timers: exact join
        // costs that came out BELOW their float32 bound (must be 0), and the smallest (exact^2 - lo^2) / (2 ceps scale^2) seen
        // (+inf: no exact cost was compared yet)
        unsigned long long v[6] = {0, 0, 0, 0, 0, 0xffffffffull};
        if (h->vstats.p) { HIPCHK(hipSetDevice(h->device)); HIPCHK(hipDeviceSynchronize()); CHK(d2h_sync(h, v, h->vstats.p, sizeof(v), h->stream)); }
        if (name[11] == 'v') *out = (double)v[4];
        else {
            const unsigned int im = (unsigned int)v[5];
            const unsigned int bits = (im & 0x80000000u) ? (im & 0x7fffffffu) : ~im;
            float f; memcpy(&f, &bits, 4);
            *out = im == 0xffffffffu ? (double)INFINITY : (double)f;
        }
    }
    else if (!strcmp(name, "f16_ready")) *out = h->f16_ready ? 1 : 0;
    else if (!strcmp(name, "f16_fallbacks")) *out = h->f16_fallbacks;
    else if (!strcmp(name, "knn_level")) *out = h->knn_level;                       // rung of the voice's list-overflow ladder (snk_engine.h)
    else if (!strcmp(name, "knn_escalations")) *out = (double)h->knn_escalations;
    else if (!strcmp(name, "last_f16_status")) *out = h->last_f16_status;
    else if (!strcmp(name, "pool_chunks_used")) { unsigned int v[2] = {0, 0}; CHK(d2h_sync(h, v, h->poolctl.p, sizeof(v), h->stream)); *out = v[0] + 1e6 * v[1]; }
    else if (!strcmp(name, "precision")) *out = h->precision;
    else if (!strcmp(name, "prefilter")) *out = h->prefilter;
    else if (!strcmp(name, "prefilter_bf16_active")) *out = h->bf16_ready ? 1 : 0;
    else if (!strcmp(name, "prefilter_rho_lo") || !strcmp(name, "prefilter_rho_res")) {
        // sqrt of the largest ||fl||^2 / ||f||^2 (||rf||^2 / ||f||^2) over the database rows: 2^-8 (2^-16) at worst
        double rho[2] = {0.0, 0.0};
        if (h->bf16_ready && h->rho16.p) {
            HIPCHK(hipSetDevice(h->device));
            HIPCHK(hipStreamSynchronize(h->stream));
            CHK(d2h_sync(h, rho, h->rho16.p, sizeof(rho), h->stream));
        }
        *out = sqrt(rho[name[14] == 'l' ? 0 : 1]);
    }
    else if (!strcmp(name, "prefilter_margin_rows") || !strcmp(name, "prefilter_min_margin")) {
        // since the engine was created (or the last snk_reset_timers): rows of prefilter K-NN calls whose exact K-th key
        // came within 2 eps of the filter threshold, and the smallest (threshold - exact K-th key) / eps seen
        unsigned int v[2] = {0u, 0x7f800000u};
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipStreamSynchronize(h->stream));
        CHK(d2h_sync(h, v, h->margin_stat.p, sizeof(v), h->stream));
        float r; memcpy(&r, &v[1], 4);
        *out = name[10] == 'm' && name[11] == 'a' ? (double)v[0] : (double)r;
    }
    else if (!strcmp(name, "finalize_list_entries") || !strcmp(name, "finalize_reranked")) {
        // since the last snk_reset_timers: list entries the re-rank read ([2..3]) and entries it gave exact float64 distances
        // ([4..5]) -- what the roofline of knn_finalize_kernel is priced on (bench.py)
        unsigned long long v[4] = {0, 0, 0, 0};
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipStreamSynchronize(h->stream));
        CHK(d2h_sync(h, v, h->margin_stat.p, sizeof(v), h->stream));
        *out = (double)v[name[9] == 'l' ? 1 : 2];
    }
    else if (!strcmp(name, "sparse_exact_costs") || !strcmp(name, "sparse_set_members")) {
        // pass 3 of the sparse Viterbi path since the last snk_reset_timers: exact costs it took from the rows, members of the
        // predecessor sets it looked at (natural successors and unusable units cost no row)
        unsigned long long v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (h->vstats.p) { HIPCHK(hipSetDevice(h->device)); HIPCHK(hipDeviceSynchronize()); CHK(d2h_sync(h, v, h->vstats.p, sizeof(v), h->stream)); }
        *out = (double)v[name[7] == 'e' ? 6 : 7];
    }
    else if (!strcmp(name, "prefilter_two_pass")) *out = h->prefilter_two_pass;
    else if (!strcmp(name, "prefilter_balls")) *out = h->prefilter_balls;
    else if (!strcmp(name, "prefilter_ball_bound")) *out = h->prefilter_ball_bound;
    else if (!strcmp(name, "prefilter_super_balls")) *out = h->prefilter_super_balls;
    else if (!strcmp(name, "wide_launches")) *out = (double)h->wide_launches;
    else if (!strcmp(name, "shard_compact")) *out = h->shard_compact;
    else if (!strcmp(name, "shard_last_sent_mb")) *out = h->shard_last_sent_mb;
    else if (!strcmp(name, "shard_last_padded_mb")) *out = h->shard_last_padded_mb;
    else if (!strcmp(name, "wide_ready")) *out = h->wide16_ready ? 1 : 0;
    else if (!strcmp(name, "filter_onepass")) *out = h->filter_onepass ? 1 : 0;     // 1: the coarse sweep listed most pairs for this voice: one-pass sweep since
    else if (!strcmp(name, "filter_coarse")) *out = h->filter_coarse ? 1 : 0;       // 1: the ball pass listed too many pairs for this voice
    else if (!strcmp(name, "ball_switches")) *out = (double)h->ball_switches;
    else if (!strcmp(name, "onepass_switches")) *out = (double)h->onepass_switches;
    else if (!strcmp(name, "reorder")) *out = h->reorder;
    else if (!strcmp(name, "reordered")) *out = h->perm_ready ? 1 : 0;               // 1: the prefilter's operands stand in an order the engine chose (kmeans_kernels.hip)
    else if (!strcmp(name, "reorders")) *out = (double)h->reorders;
    else if (!strcmp(name, "reorder_failures")) *out = (double)h->reorder_failures;
    else if (!strcmp(name, "reorder_useless")) *out = h->reorder_useless ? 1 : 0;
    else if (!strcmp(name, "debug_perm_ptr")) *out = (double)(uintptr_t)(h->perm_ready ? h->perm.p : nullptr);     // developer aid: snk_copy_to_host reads it
    else if (!strcmp(name, "debug_ball_rad_ptr")) *out = (double)(uintptr_t)h->ball_rad.p;
    else if (!strcmp(name, "reorder_radius_before")) *out = h->reorder_radius_before;
    else if (!strcmp(name, "reorder_radius_after")) *out = h->reorder_radius_after;
    else if (!strcmp(name, "filter_rearms")) *out = (double)h->filter_rearms;        // times a counting probe took the voice back to a faster filter
    else if (!strcmp(name, "filter_probe_period")) *out = h->probe_period;
    else if (!strcmp(name, "latch_rearm")) *out = h->latch_rearm;
    else if (!strcmp(name, "viterbi_latch")) *out = h->viterbi_latch;
    else if (!strcmp(name, "viterbi_fst32_slack")) *out = h->fst32_slack;
    else if (!strcmp(name, "join_lb_quadrants")) *out = get_join_lb_quadrants();
    else if (!strcmp(name, "join_lb_one_set")) *out = get_join_lb_one_set();
    else if (!strcmp(name, "join_bounds_delay")) *out = h->join_bounds_delay;
    else if (!strcmp(name, "split_one_group")) *out = h->split_one_group;
    else if (!strcmp(name, "wide_one_group")) *out = h->wide_one_group;
    else if (!strcmp(name, "upload_stream")) *out = h->upload_stream;
    else if (!strcmp(name, "results_by_kernel")) *out = h->results_by_kernel;
    else if (!strcmp(name, "tail_defer")) *out = h->tail_defer;
    else if (!strcmp(name, "submits_starved")) *out = (double)h->submits_starved;     // pipelined submits that found the K-NN stream idle (the host was late)
    else if (!strcmp(name, "submits_pipelined")) *out = (double)h->submits_seen;
    else if (!strcmp(name, "submits_starved_recent")) *out = h->starved_ema;
    else if (!strcmp(name, "roofline_counters")) *out = h->roofline_counters;
    else if (!strcmp(name, "tau_optimism")) *out = h->tau_optimism;
    else if (!strcmp(name, "tau_optimism_rank")) *out = h->opt_last_rank;               // j of the most recent call (0: guaranteed thresholds)
    else if (!strcmp(name, "tau_optimism_failures")) *out = (double)h->opt_fails_total;   // calls / groups redone with guaranteed thresholds
    else if (!strcmp(name, "tau_optimism_off")) *out = h->opt_off ? 1 : 0;              // this voice went back to guaranteed thresholds
    else if (!strcmp(name, "viterbi_refine_gate")) *out = h->vit_refine_gate;
    else if (!strcmp(name, "viterbi_latch_mode")) *out = h->vit.mode;                 // 0: batches take the sparse path, 1: the dense kernels (judged, snk_engine.h)
    else if (!strcmp(name, "viterbi_lb_warm_now")) *out = h->lb_warm_eff > h->lb_warm ? h->lb_warm_eff : h->lb_warm;       // warm-up of pass 2's chunks this voice runs with
    else if (!strcmp(name, "viterbi_lb_warm_raises")) *out = (double)h->lb_warm_raises;
    else if (!strcmp(name, "viterbi_latch_switches")) *out = (double)h->vit.switches;
    else if (!strcmp(name, "viterbi_latch_trials")) *out = (double)h->vit.trials;
    else if (!strcmp(name, "viterbi_latch_ms_row_sparse")) *out = h->vit.ms_row[0];
    else if (!strcmp(name, "viterbi_latch_ms_row_dense")) *out = h->vit.ms_row[1];
    else if (!strcmp(name, "coarse_pairs") || !strcmp(name, "coarse_pair_overflow")) {
        // tile pairs the coarse pass of the most recent two-pass filter let through (debug / tuning aid)
        unsigned int v[2] = {0u, 0u};
        if (h->cpairctl.p) { HIPCHK(hipSetDevice(h->device)); HIPCHK(hipStreamSynchronize(h->stream)); CHK(d2h_sync(h, v, h->cpairctl.p, sizeof(v), h->stream)); }
        *out = (double)v[name[11] == 's' ? 0 : 1];
    }
    else if (!strcmp(name, "prefilter_mfma_unit")) *out = SNK_BF16_MFMA_UNIT;
    else if (!strcmp(name, "prefilter_eps_c")) *out = (h->bf16_ready && h->prefilter >= 1) ? h->eps_c_bf : h->eps_c;
    else if (!strcmp(name, "batch_rows")) *out = h->batch_rows;
    else if (!strcmp(name, "last_list_mean") || !strcmp(name, "last_list_max")) {
        // candidate-list lengths of the most recent K-NN call (debug / tuning aid)
        const int64_t n = h->last_T;
        if (n <= 0) { *out = 0; return 0; }
        std::vector<int> c((size_t)n);
        CHK(d2h_sync(h, c.data(), h->cnt.p, (size_t)n * sizeof(int), h->stream));
        double sum = 0, mx = 0;
        for (int64_t i = 0; i < n; ++i) { sum += c[i]; if (c[i] > mx) mx = c[i]; }
        *out = !strcmp(name, "last_list_max") ? mx : sum / (double)n;
    }
    else if (!strcmp(name, "device")) *out = h->device;
    else if (!strcmp(name, "db_tiles_per_wave")) { KnnPlan p = make_plan(h, 100); *out = p.nt; }
    else if (!strcmp(name, "sample_slabs")) { KnnPlan p = make_plan(h, 100); *out = (double)p.a_count; }
    else if (!strcmp(name, "n_slabs")) { KnnPlan p = make_plan(h, 100); *out = (double)p.n_slabs; }
    else return fail("snk_get_info: unknown item '%s'", name);
    return 0;
}

static int selftest_mfma16(snk_engine *h, double *err_out)
{
    float A[64], B[64], C[1024], R[1024];
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 2; ++k) A[i * 2 + k] = (float)((3 * i + 7 * k + 1) % 11 - 5);
    for (int k = 0; k < 2; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (float)((5 * k - 2 * j + (k * j) % 3) % 7);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        float acc = 0; for (int k = 0; k < 2; ++k) acc += A[i * 2 + k] * B[k * 32 + j]; R[i * 32 + j] = acc; }
    DevBuf dA, dB, dC;
    CHK(dA.ensure(sizeof(A))); CHK(dB.ensure(sizeof(B))); CHK(dC.ensure(sizeof(C)));
    CHK(h2d_sync(h, dA.p, A, sizeof(A)));
    CHK(h2d_sync(h, dB.p, B, sizeof(B)));
    HIPCHK(hipMemset(dC.p, 0, sizeof(C)));
    launch_mfma16_selftest(dA.as<float>(), dB.as<float>(), dC.as<float>(), h->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    CHK(d2h_sync(h, C, dC.p, sizeof(C), h->stream));
    double err = 0;
    for (int i = 0; i < 1024; ++i) err = fmax(err, fabs((double)C[i] - (double)R[i]));
    dA.release(); dB.release(); dC.release();
    *err_out = err;
    return 0;
}

// One v_mfma_f32_32x32x16_bf16 on the caller's bit patterns (include/snk.h): the probe behind the accumulation term of
// the bf16-split prefilter's bound.
int snk_probe_mfma_bf16(snk_handle h, const uint16_t *A, const uint16_t *B, const float *C, float *D_out)
{
    if (!h) return fail("null handle");
    if (!A || !B || !C || !D_out) return fail("snk_probe_mfma_bf16: null argument");
    HIPCHK(hipSetDevice(h->device));
    DevBuf dA, dB, dC, dD;
    CHK(dA.ensure(512 * 2)); CHK(dB.ensure(512 * 2)); CHK(dC.ensure(1024 * 4)); CHK(dD.ensure(1024 * 4));
    CHK(h2d_sync(h, dA.p, A, 512 * 2));
    CHK(h2d_sync(h, dB.p, B, 512 * 2));
    CHK(h2d_sync(h, dC.p, C, 1024 * 4));
    launch_mfma_bf16_probe(dA.as<unsigned short>(), dB.as<unsigned short>(), dC.as<float>(), dD.as<float>(), h->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    CHK(d2h_sync(h, D_out, dD.p, 1024 * 4, h->stream));
    dA.release(); dB.release(); dC.release(); dD.release();
    return 0;
}

int snk_selftest_mfma(snk_handle h, double *max_abs_err_out)
{
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    double A[64], B[64], C[256], R[256];
    for (int i = 0; i < 16; ++i)
        for (int k = 0; k < 4; ++k) A[i * 4 + k] = (double)(3 * i + 7 * k + 1);      // asymmetric
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 16; ++j) B[k * 16 + j] = (double)(5 * k - 2 * j + (k * j) % 3);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double acc = 0;
            for (int k = 0; k < 4; ++k) acc += A[i * 4 + k] * B[k * 16 + j];
            R[i * 16 + j] = acc;
        }
    DevBuf dA, dB, dC;
    CHK(dA.ensure(sizeof(A))); CHK(dB.ensure(sizeof(B))); CHK(dC.ensure(sizeof(C)));
    CHK(h2d_sync(h, dA.p, A, sizeof(A)));
    CHK(h2d_sync(h, dB.p, B, sizeof(B)));
    HIPCHK(hipMemset(dC.p, 0, sizeof(C)));
    launch_mfma_selftest(dA.as<double>(), dB.as<double>(), dC.as<double>(), h->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    CHK(d2h_sync(h, C, dC.p, sizeof(C), h->stream));
    double err = 0;
    for (int i = 0; i < 256; ++i) err = fmax(err, fabs(C[i] - R[i]));
    dA.release(); dB.release(); dC.release();
    double err16 = 0;
    CHK(selftest_mfma16(h, &err16));          // f16 32x32x16 operand / result maps of the prefilter
    // the bf16 instruction of the split prefilter: operand map (small integers: exact) and the accumulation assumption
    // behind its key bound on the pattern that shows the unit's cut (one product of 1, fifteen just under 2^-24)
    double errbf = 0;
    {
        uint16_t A[512], B[512];
        float C[1024], D[1024];
        auto bits = [](float x) { unsigned int u; memcpy(&u, &x, 4); return (uint16_t)(u >> 16); };   // exact for the values used
        for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = bits((float)((3 * i + 5 * k) % 9 - 4));
        for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = bits((float)((7 * k - 2 * j) % 5));
        for (int i = 0; i < 1024; ++i) C[i] = (float)(i % 7);
        CHK(snk_probe_mfma_bf16(h, A, B, C, D));
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double acc = (double)C[i * 32 + j];
            for (int k = 0; k < 16; ++k) acc += (double)((3 * i + 5 * k) % 9 - 4) * (double)((7 * k - 2 * j) % 5);
            errbf = fmax(errbf, fabs((double)D[i * 32 + j] - acc));
        }
        for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = bits(k == 0 ? 1.0f : 0.000244140625f);          // 2^-12
        for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = bits(k == 0 ? 1.0f : 0.000236511230469f);        // 1.9375 2^-13
        for (int i = 0; i < 1024; ++i) C[i] = 0.f;
        CHK(snk_probe_mfma_bf16(h, A, B, C, D));
        const double small = 0.000244140625 * 0.000236511230469, exact = 1.0 + 15.0 * small, mass = exact;
        for (int i = 0; i < 1024; ++i)
            if (fabs((double)D[i] - exact) > SNK_BF16_MFMA_UNIT * mass) errbf = fmax(errbf, fabs((double)D[i] - exact));
    }
    if (max_abs_err_out) *max_abs_err_out = fmax(fmax(err, err16), errbf);
    return 0;
}
