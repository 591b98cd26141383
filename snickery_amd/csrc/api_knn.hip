// C ABI of libsnkhip.so, part 2: the K-NN preselection pipeline on the device (knn_device) and its entry points
// (preselect_units_acoustic / _monophone_then_acoustic / _quinphone, synth_halfphone.py:1305-1396).
#include "snk_engine.h"

// ---------------------------------------------------------------------------
// K-NN pipeline on device.  Q must already be on the device (Qraw).  Results go to the
// given device buffers.  Synchronises the stream once per attempt to read the status word.
// ---------------------------------------------------------------------------
// SNK_TRACE: the entry pool the filter sweep left, checked on the host before the bucket kernel scatters it
static int debug_check_pool(snk_engine *h, int max_chunks, int64_t Tpad, int64_t idx_limit, hipStream_t s)
{
    HIPCHK(hipStreamSynchronize(s));
    unsigned int ctl[2] = {0, 0};
    CHK(d2h_sync(h, ctl, h->poolctl.p, sizeof(ctl), s));
    int used = (int)ctl[0];
    if (used > max_chunks) used = max_chunks;
    std::vector<int> fill((size_t)max_chunks);
    CHK(d2h_sync(h, fill.data(), h->chunkfill.p, fill.size() * sizeof(int), s));
    struct E { double key; int idx; int row; };
    const int chunk = knn_pool_chunk_entries();
    std::vector<E> en((size_t)chunk);
    int64_t total = 0, bad = 0;
    for (int c = 0; c < used; ++c) {
        const int n = fill[(size_t)c];
        if (n < 0 || n > chunk) { fprintf(stderr, "[snk-trace] pool: chunk %d of %d has fill %d (chunk size %d)\n", c, used, n, chunk); ++bad; continue; }
        if (n == 0) continue;
        CHK(d2h_sync(h, en.data(), (const char *)h->pool.p + (size_t)c * chunk * sizeof(E), (size_t)n * sizeof(E), s));
        for (int e = 0; e < n; ++e) {
            ++total;
            if (en[(size_t)e].row < 0 || en[(size_t)e].row >= Tpad || en[(size_t)e].idx < 0 || en[(size_t)e].idx >= idx_limit) {
                if (bad < 8) fprintf(stderr, "[snk-trace] pool: chunk %d entry %d of %d: row %d idx %d key %g (Tpad %lld, idx limit %lld)\n", c, e, n,
                                     en[(size_t)e].row, en[(size_t)e].idx, en[(size_t)e].key, (long long)Tpad, (long long)idx_limit);
                ++bad;
            }
        }
    }
    fprintf(stderr, "[snk-trace] pool: %u chunks handed out (max %d, overflow %u), %lld entries, %lld bad\n", ctl[0], max_chunks, ctl[1], (long long)total, (long long)bad);
    return 0;
}

// what the ball pass of a call listed, seen at the call's next host synchronisation: beyond the limit this voice's
// filter goes back to the coarse sweep (until the weights change)
void judge_filter(snk_engine *h, bool ran_balls, double ball_limit, bool ran_coarse, double coarse_limit, unsigned int listed,
                  int probe_kind, double probe_limit, unsigned int probe_listed)
{
    auto left = [&]() { h->probe_period = 16; h->probe_next = h->filter_calls + 16; };
    if (ran_balls && !h->filter_coarse && (double)listed > ball_limit) {
        h->filter_coarse = true; h->ball_switches += 1; left();
        // the tiles are not compact: give the voice an order of its own before the next call (once per set of weights; a voice the
        // clustering did not help stays as it is)
        if (h->reorder && !h->reorder_useless && !h->reorder_done) h->reorder_pending = true;
    }
    if (ran_coarse && !h->filter_onepass && (double)listed > coarse_limit) { h->filter_onepass = true; h->onepass_switches += 1; left(); }
    // the counting probe of the pass the voice left (snk_engine.h: latch_rearm)
    const bool applies = (probe_kind == 1 && h->filter_coarse && !h->filter_onepass) || (probe_kind == 2 && h->filter_onepass);
    if (!applies) return;
    if ((double)probe_listed <= probe_limit) {
        if (probe_kind == 1) h->filter_coarse = false; else h->filter_onepass = false;
        h->filter_rearms += 1;
        left();
    } else {
        h->probe_period = h->probe_period < 128 ? 2 * h->probe_period : 256;
        h->probe_next = h->filter_calls + h->probe_period;
    }
}

// sync paths that read the first control word only: what the most recent call listed
void note_ball_pairs(snk_engine *h, unsigned int listed)
{
    judge_filter(h, h->ball_pass_ran, h->ball_limit, h->coarse_pass_ran, h->coarse_limit, listed, 0, 0.0, 0u);
}

// a call (or a group of a batch) under optimistic thresholds had a row flagged: counted, and a voice that fails more often than
// once in 200 calls goes back to the guaranteed thresholds until its weights change
void note_optimism_failure(snk_engine *h)
{
    h->opt_fails += 1; h->opt_fails_total += 1;
    if (h->opt_fails >= 3 && h->opt_fails * 200 > h->opt_calls) h->opt_off = true;
}

KnnPlan make_plan(snk_engine *h, int K)
{
    KnnPlan p{};
    p.dch = h->Dpad / 64;
    int nt = (p.dch == 1) ? 4 : (p.dch == 2) ? 4 : (p.dch == 3) ? 2 : 1;
    if (h->nt_override > 0 && p.dch <= 2) {
        if (h->nt_override == 2 || h->nt_override == 4 || (h->nt_override == 8 && p.dch == 1)) nt = h->nt_override;
    }
    p.nt = nt;
    const int64_t slab_rows = 16 * nt;
    p.n_slabs = (h->N + slab_rows - 1) / slab_rows;
    p.row_limit = h->N;
    p.grid_cus = h->n_cus - h->reserved_cus;
    if (p.grid_cus < 1) p.grid_cus = 1;
    p.slab_counter = h->slabctr.as<unsigned int>();
    // stage-A sample: every a_stride-th database row (uniform at single-unit granularity),
    // at least 4K groups of sampled-slab-lane minima (16 groups per slab of 16*nt sampled rows)
    int64_t stride = (int64_t)floor(1.0 / h->sample_frac + 0.5);
    if (stride < 1) stride = 1;
    const int64_t min_rows = ((4 * (int64_t)K + 15) / 16) * slab_rows;
    while (stride > 1 && (h->N + stride - 1) / stride < min_rows) --stride;
    p.a_stride = stride;
    const int64_t sample_rows = (h->N + stride - 1) / stride;
    p.a_count = (sample_rows + slab_rows - 1) / slab_rows;
    return p;
}

// deferred_status != nullptr: enqueue the first attempt only, leave its status word in that device
// int and do NOT synchronise (batch pipeline: the caller checks all words at the end of the batch
// and redoes the rare overflowed utterance synchronously).
//
// Row-sharded databases (snk_knn_local_batch_bounds_dev / _bounded_dev): `bound_out` != nullptr runs stage
// A only and leaves, per row, an upper bound of the K-th nearest key of THIS shard (DBL_MAX where the
// f32 path cannot give one); `bound_in` != nullptr skips stage A and filters against bound_in + eps --
// the caller passes the minimum of the bounds of all shards, which still bounds the K-th nearest key
// of the whole database.  Lists may then hold fewer than K entries (padded with id -1).
// gs (with bound_out): stage A runs against the replicated GLOBAL sample (snk_upload_global_sample) instead of
// this shard's own one -- the bound is then that of the whole database, as on a single GPU.

// refine (with bound_in, inside snk_sharded_knn_viterbi_batch): between bucket and re-rank the shards agree on a
// second, tighter bound -- the smallest of their lists' K-th keys (knn_kernels.hip knn_local_kth_kernel) -- with one
// more all-reduce per call, and prune their lists to it.
int knn_device(snk_engine *h, const double *Qdev, int64_t T, int K, const int32_t *qclass_dev,
               int64_t *cand_dev, double *dist_dev, double *d2_dev, int *deferred_status,
               const double *bound_in, double *bound_out, bool gs, bool refine,
               unsigned int *pairs_listed_dev, unsigned int *probe_listed_dev)
{
    if (K < 1 || K > 208) return fail("K-NN: n_candidates=%d outside the supported range 1..208", K);
    if (T > SNK_KNN_MAX_ROWS) {
        // very long query matrices: the per-row workspaces (and the bucket kernel's LDS histogram)
        // are sized for SNK_KNN_MAX_ROWS rows; the search is per row, so it is cut into calls
        for (int64_t r0 = 0; r0 < T; r0 += SNK_KNN_MAX_ROWS) {
            const int64_t rows = (T - r0 < SNK_KNN_MAX_ROWS) ? T - r0 : SNK_KNN_MAX_ROWS;
            CHK(knn_device(h, Qdev + r0 * h->Dt, rows, K, qclass_dev ? qclass_dev + r0 : nullptr,
                           cand_dev ? cand_dev + r0 * K : nullptr, dist_dev ? dist_dev + r0 * K : nullptr,
                           d2_dev ? d2_dev + r0 * K : nullptr, nullptr, bound_in ? bound_in + r0 : nullptr,
                           bound_out ? bound_out + r0 : nullptr, gs, refine));
        }
        if (deferred_status) HIPCHK(hipMemsetAsync(deferred_status, 0, sizeof(int), h->stream));
        return 0;
    }
    if (h->Dpad > 256 && h->precision == 1 && !qclass_dev && !bound_out && !bound_in && !deferred_status) CHK(ensure_wide_operands(h));
    if (h->Dpad > 256 && h->wide16_ready && h->precision == 1 && !qclass_dev && !bound_out && !bound_in && !deferred_status &&
        2 * h->n_slabs16_a >= K) {
        // 257 .. 512 columns: stage A and the filter as a blocked bf16-split product (knn_wide16b), bucket and the exact
        // float64 re-rank as for every other width; a list or pool overflow sends the call to the exact selection below
        const int64_t Tpad = roundup(T, 32);
        const KnnPlan p0 = make_plan(h, K);
        int cap = h->cap;
        if (cap < 40 * K) cap = 40 * K < 8192 ? 40 * K : 8192;
        const int64_t G16 = 2 * h->n_slabs16_a;
        const int terms = h->prefilter == 2 ? 4 : 3;
        CHK(h->Qp.ensure((size_t)Tpad * h->Dpad * sizeof(double)));
        CHK(h->qnorm.ensure((size_t)Tpad * sizeof(double)));
        CHK(h->thr.ensure((size_t)Tpad * sizeof(double)));
        CHK(h->cnt.ensure((size_t)Tpad * sizeof(int)));
        CHK(h->lkey.ensure((size_t)Tpad * cap * sizeof(double)));
        CHK(h->lidx.ensure((size_t)Tpad * cap * sizeof(int)));
        CHK(h->status.ensure(sizeof(int)));
        int max_chunks = h->pool_chunks;
        {
            const int64_t want_row = 20 * (int64_t)K > 3072 ? 20 * (int64_t)K : 3072;
            const int64_t need = (Tpad * (cap < want_row ? cap : want_row)) / knn_pool_chunk_entries() + 2048;
            if (need > max_chunks) max_chunks = (int)need;
        }
        if (h->pool_chunk_limit > 0 && max_chunks > h->pool_chunk_limit) max_chunks = h->pool_chunk_limit;
        CHK(h->pool.ensure(knn_pool_bytes(max_chunks)));
        CHK(h->poolctl.ensure(2 * sizeof(unsigned int)));
        CHK(h->chunkfill.ensure((size_t)max_chunks * sizeof(int)));
        CHK(h->b16l.ensure((size_t)(Tpad / 32) * 8 * 64 * 16 * (h->Dpad / 64)));
        CHK(h->eps16.ensure((size_t)Tpad * sizeof(double)));
        CHK(h->cq16.ensure((size_t)Tpad * sizeof(double)));
        CHK(h->thr32.ensure((size_t)Tpad * sizeof(float)));
        CHK(h->gmin32.ensure((size_t)Tpad * G16 * sizeof(float)));
        hipStream_t s = h->stream;
        {
            StageTimer t(h, s, TM_PREP);
            launch_prepare_queries(Qdev, T, h->Dt, h->Qp.as<double>(), nullptr, h->qnorm.as<double>(), Tpad, h->Dpad, s);
            launch_knn_reset(h->cnt.as<int>(), Tpad, h->status.as<int>(), h->poolctl.as<unsigned int>(), h->slabctr.as<unsigned int>(),
                             h->chunkfill.as<int>(), max_chunks, s);
            launch_prepare_queries16b(h->Qp.as<double>(), h->qnorm.as<double>(), T, h->Dt, h->Dpad, h->fmax2.as<double>(),
                                      h->rho16.as<double>(), h->eps_c_bf, h->b16l.p, h->eps16.as<double>(), h->cq16.as<double>(), s);
        }
        {
            StageTimer t(h, s, TM_KNN_MINIMA);
            launch_knn_wide16b(0, terms, p0.grid_cus, h->s16l.p, h->b16l.p, h->Dpad, nullptr, Tpad, h->n_slabs16_a, h->gmin32.as<float>(), G16,
                               nullptr, nullptr, nullptr, 0, knn_pool_chunk_entries(), s);
        }
        {
            StageTimer t(h, s, TM_KNN_THRESHOLD);
            launch_knn_threshold16(h->gmin32.as<float>(), G16, T, Tpad, K, h->eps16.as<double>(), h->thr.as<double>(), h->thr32.as<float>(),
                                   nullptr, nullptr, s);
        }
        {
            StageTimer t(h, s, TM_KNN_FILTER);
            launch_knn_wide16b(1, terms, p0.grid_cus, h->a16l.p, h->b16l.p, h->Dpad, h->thr32.as<float>(), Tpad, h->n_slabs16, nullptr, 0,
                               h->pool.p, h->poolctl.as<unsigned int>(), h->chunkfill.as<int>(), max_chunks, knn_pool_chunk_entries(), s);
        }
        {
            StageTimer t(h, s, TM_KNN_BUCKET);
            launch_knn_bucket(h->pool.p, h->poolctl.as<unsigned int>(), h->chunkfill.as<int>(), max_chunks, Tpad, h->N, h->cnt.as<int>(),
                              h->lkey.as<double>(), h->lidx.as<int>(), cap, h->status.as<int>(), s, nullptr, true);
        }
        {
            StageTimer t(h, s, TM_KNN_FINALIZE);
            launch_knn_finalize(h->Fw.as<double>(), h->F_unw.as<float>(), h->Fp, h->wt.as<double>(), h->Dpad, h->Dt, h->Qp.as<double>(),
                                h->qnorm.as<double>(), T, K, h->cnt.as<int>(), h->lkey.as<double>(), h->lidx.as<int>(), cap, h->shard_offset,
                                h->eps16.as<double>(), h->fnorm.as<double>(), h->eps_c_bf, h->cq16.as<double>(), cand_dev, dist_dev, d2_dev,
                                h->status.as<int>(), nullptr, s, false, h->thr.as<double>(), h->margin_stat.as<unsigned int>());
        }
        int status = 0;
        CHK(d2h_sync(h, &status, h->status.p, sizeof(int), s));
        HIPCHK(hipGetLastError());
        h->last_retries = 0; h->last_T = T;
        h->last_f16_status = status;
        if (status == 0) { h->wide_launches += 1; return 0; }
        h->f16_fallbacks += 1;               // overflow or too many near ties: the exact selection below serves the call
    }
    if (h->Dpad > 256) {
        // Rows wider than a database row's fragments fit a wavefront's registers: every row through the exact selection
        // (knn_exact_rows_kernel: canonical distances to every unit by one workgroup per query row, radix select, ties by
        // lowest id) -- the all-pairs join K-NN of active_learning_join.py:184-212 on 302-column join rows.  No bounds
        // for a sharded caller (nothing is pruned), nothing deferred.
        if (bound_out) { launch_fill_threshold(bound_out, T, T, DBL_MAX, h->stream); return 0; }
        const int64_t Tp = roundup(T, 16);
        CHK(h->Qp.ensure((size_t)Tp * h->Dpad * sizeof(double)));
        CHK(h->qnorm.ensure((size_t)Tp * sizeof(double)));
        launch_prepare_queries(Qdev, T, h->Dt, h->Qp.as<double>(), nullptr, h->qnorm.as<double>(), Tp, h->Dpad, h->stream);
        // rows per launch: as many workgroups as the scratch (one float64 per unit and row) allows within 2 GB
        int64_t per = ((int64_t)2 << 30) / ((int64_t)h->Nalloc * 8);
        per = per < 1 ? 1 : per > 512 ? 512 : per;
        CHK(h->exact_rows.ensure((size_t)per * sizeof(int)));
        CHK(h->exact_scratch.ensure((size_t)per * h->Nalloc * sizeof(double)));
        std::vector<int> rows((size_t)per);
        for (int64_t r0 = 0; r0 < T; r0 += per) {
            const int n = (int)(T - r0 < per ? T - r0 : per);
            for (int i = 0; i < n; ++i) rows[(size_t)i] = (int)(r0 + i);
            CHK(h2d(h, h->exact_rows.p, rows.data(), (size_t)n * sizeof(int), h->stream));
            launch_knn_exact_rows(h->Fw.as<double>(), h->Dpad, h->Dt, h->N, h->Qp.as<double>(), h->exact_rows.as<int>(), n, K,
                                  h->exact_scratch.as<double>(), h->Nalloc, qclass_dev ? h->unit_class.as<int32_t>() : nullptr,
                                  qclass_dev, h->shard_offset, cand_dev, dist_dev, d2_dev, h->stream);
            HIPCHK(hipStreamSynchronize(h->stream));          // (the host array of row numbers is reused)
        }
        HIPCHK(hipGetLastError());
        if (deferred_status) HIPCHK(hipMemsetAsync(deferred_status, 0, sizeof(int), h->stream));
        h->last_retries = 0; h->last_T = T;
        return 0;
    }
    if (h->reorder_pending && h->precision == 1) CHK(reorder_units(h));      // (queued on this stream: behind every call that still reads the old operands)
    const int64_t Tpad = roundup(T, 32);
    const KnnPlan p0 = make_plan(h, K);
    const bool cls = qclass_dev != nullptr;
    const int32_t *uc = cls ? h->unit_class.as<int32_t>() : nullptr;
    int cap = h->cap;
    if (cap < 2 * K) cap = 2 * K;
    // the sampled thresholds let ~17 K candidates per row through (mean; 3200 at K = 200): large K needs
    // longer lists and a bigger pool share than the defaults sized for K <= 128
    // (40 K: the bf16-split prefilter's wider key margin lengthens the lists by a fifth)
    if (cap < 40 * K) cap = 40 * K < 8192 ? 40 * K : 8192;
    if (h->knn_level >= 1 && cap < 8192) cap = 8192;          // this voice's lists overflowed before (snk_engine.h: knn_level)
    if (K > 4096) return fail("K-NN: K too large");
    KnnPlan p = p0;
    int64_t G = p.a_count * 16;
    CHK(h->Qp.ensure((size_t)Tpad * h->Dpad * sizeof(double)));
    CHK(h->Qf.ensure((size_t)Tpad * h->Dpad * sizeof(double)));
    CHK(h->qnorm.ensure((size_t)Tpad * sizeof(double)));
    CHK(h->thr.ensure((size_t)Tpad * sizeof(double)));
    CHK(h->gmin.ensure((size_t)Tpad * G * sizeof(double)));
    CHK(h->cnt.ensure((size_t)Tpad * sizeof(int)));
    CHK(h->lkey.ensure((size_t)Tpad * cap * sizeof(double)));
    CHK(h->lidx.ensure((size_t)Tpad * cap * sizeof(int)));
    CHK(h->status.ensure(sizeof(int)));
    CHK(h->rowflag.ensure((size_t)Tpad * sizeof(int)));
    // entry pool: room for ~3K survivors per row plus one partly filled chunk per resident wave
    int max_chunks = h->pool_chunks;
    {
        const int64_t want_row = 20 * (int64_t)K > 3072 ? 20 * (int64_t)K : 3072;
        const int64_t per_row = cap < want_row ? cap : want_row;
        const int64_t need = (Tpad * per_row) / knn_pool_chunk_entries() + 2048;
        if (need > max_chunks) max_chunks = (int)need;
    }
    if (h->pool_chunk_limit > 0 && max_chunks > h->pool_chunk_limit) max_chunks = h->pool_chunk_limit;
    CHK(h->pool.ensure(knn_pool_bytes(max_chunks)));
    CHK(h->poolctl.ensure(2 * sizeof(unsigned int)));
    CHK(h->chunkfill.ensure((size_t)max_chunks * sizeof(int)));
    hipStream_t s = h->stream;
    {
        StageTimer t(h, s, TM_PREP);
        launch_prepare_queries(Qdev, T, h->Dt, h->Qp.as<double>(), h->Qf.as<double>(), h->qnorm.as<double>(),
                               Tpad, h->Dpad, s);
    }
    h->last_retries = 0;
    h->last_T = T;
    int *status_dev = deferred_status ? deferred_status : h->status.as<int>();

    // ---- fast path: f16-split prefilter (exact results through the float64 re-rank) ----
    // class-restricted searches run the 2-tile (one chunk) / 2- / 1-tile variants of the f32 sweep
    const int dch16 = h->Dpad / 64;
    const int nt_run = (cls && dch16 == 1) ? 2 : h->nt16_eff;
    const int slab_factor = h->nt16_eff / (nt_run > 0 ? nt_run : 1);
    const bool use_gs = gs && bound_out && !cls && h->gs_ready && 2 * h->gs_slabs >= K;
    if (h->precision == 1 && h->f16_ready && (!cls || (h->nt16_eff % nt_run) == 0) &&
        (use_gs || 2 * h->n_slabs16_a * slab_factor >= K)) {
        const int64_t n_slabs_a = use_gs ? h->gs_slabs : h->n_slabs16_a * slab_factor, n_slabs_b = h->n_slabs16 * slab_factor;
        const int64_t G16 = 2 * n_slabs_a;
        const int32_t *cls_full = nullptr, *cls_samp = nullptr;
        if (cls) {
            if (!h->cls16_ready) {
                const int64_t tiles_b = h->n_slabs16 * h->nt16_eff, tiles_a = h->n_slabs16_a * h->nt16_eff;
                CHK(h->cls16_full.ensure((size_t)tiles_b * 32 * sizeof(int32_t)));
                CHK(h->cls16_samp.ensure((size_t)tiles_a * 32 * sizeof(int32_t)));
                const int32_t *perm = h->perm_ready ? h->perm.as<int32_t>() : nullptr;
                launch_build_class16(h->unit_class.as<int32_t>(), h->N, tiles_b, 0, 0, h->nt16_eff,
                                     h->cls16_full.as<int32_t>(), s, perm);
                launch_build_class16(h->unit_class.as<int32_t>(), h->N, tiles_a, h->stride16, 2 * h->n_slabs16_a,
                                     h->nt16_eff, h->cls16_samp.as<int32_t>(), s, perm);
                h->cls16_ready = true;
            }
            cls_full = h->cls16_full.as<int32_t>();
            cls_samp = h->cls16_samp.as<int32_t>();
        }
        const bool bf = h->bf16_ready && h->prefilter >= 1 && !cls && nt_run == h->nt16_eff && h->knn_level < 2;
        const double eps_c_run = bf ? h->eps_c_bf : h->eps_c;
        // two-pass filter (knn16_kernels.hip): not for stage-A-only calls
        const bool twopass_ok = bf && h->prefilter_two_pass && !bound_out && knn_coarse16b_supported(nt_run, dch16);
        const bool coarse = twopass_ok && !h->filter_onepass;
        const int64_t n_tiles_b = n_slabs_b * nt_run;
        unsigned int pair_cap = 0;
        if (!coarse) { h->ball_pass_ran = false; h->coarse_pass_ran = false; }      // (nothing listed by this call: nothing to judge the voice by)
        // a voice on a slower filter: now and then the pass it left runs beside it as a probe that only counts (snk_engine.h)
        int probe = 0;
        h->probe_ran = 0;
        if (twopass_ok && h->latch_rearm && h->prefilter_balls && (h->filter_coarse || h->filter_onepass)) {
            h->filter_calls += 1;
            if (h->filter_calls >= h->probe_next) {
                probe = h->filter_onepass ? 2 : 1;
                if (probe == 1 && !(h->ball_tiles > 0 && nt_run == h->nt16_eff)) probe = 0;
                h->probe_next = h->filter_calls + h->probe_period;        // (until this probe is judged)
            }
        }
        if (probe && !coarse) {
            CHK(h->cpairs.ensure(64));
            CHK(h->cpairctl.ensure(4 * sizeof(unsigned int)));
            HIPCHK(hipMemsetAsync(h->cpairctl.p, 0, 4 * sizeof(unsigned int), s));
            if (probe == 2) { CHK(h->e1_16.ensure((size_t)Tpad * sizeof(double))); CHK(h->thr1_32.ensure((size_t)Tpad * sizeof(float))); }
        }
        const bool want_thr1 = coarse || probe == 2;
        if (coarse) {
            const int64_t all = (Tpad / 32) * n_tiles_b;
            int64_t capp = all / 4 > ((int64_t)4 << 20) ? all / 4 : ((int64_t)4 << 20);
            if (capp > all) capp = all;
            if (capp > ((int64_t)1 << 31) - 1) capp = ((int64_t)1 << 31) - 1;
            pair_cap = (unsigned int)capp;
            CHK(h->cpairs.ensure((size_t)pair_cap * knn_coarse_pair_bytes()));
            CHK(h->cpairctl.ensure(4 * sizeof(unsigned int)));
            CHK(h->e1_16.ensure((size_t)Tpad * sizeof(double)));
            CHK(h->thr1_32.ensure((size_t)Tpad * sizeof(float)));
            // (cleared by the call's knn_reset below, with the super balls' mask and the re-rank's retry flags: three dispatches
            // less on the K-NN stream per call)
        }
        // accumulation of the coarse pass's own chain (one MFMA per k-block through C) on top of the three-term chain's
        const double c_coarse = eps_c_run + 1.02 * SNK_BF16_MFMA_UNIT * (double)(h->Dpad / 16 + 1);
        CHK((bf ? h->b16l : h->b16h).ensure((size_t)(Tpad / 32) * 8 * 64 * 16 * (h->Dpad / 64)));
        CHK(h->eps16.ensure((size_t)Tpad * sizeof(double)));
        if (bf) CHK(h->cq16.ensure((size_t)Tpad * sizeof(double)));
        CHK(h->thr32.ensure((size_t)Tpad * sizeof(float)));
        CHK(h->gmin32.ensure((size_t)Tpad * G16 * sizeof(float)));
        // the balls of 32 tiles run first where the ball pass does (same test as below): their mask is cleared with the rest
        const bool will_super = coarse && h->prefilter_balls && h->ball_tiles > 0 && slab_factor == 1 && !h->filter_coarse &&
                                h->prefilter_super_balls && h->ball_supers > 0;
        const size_t mask_words = will_super ? (size_t)h->ball_supers * ((Tpad / 32 + 31) / 32) : 0;
        if (will_super) CHK(h->ball_mask.ensure(mask_words * sizeof(unsigned int)));
        launch_knn_reset(h->cnt.as<int>(), Tpad, status_dev, h->poolctl.as<unsigned int>(),
                         h->slabctr.as<unsigned int>(), h->chunkfill.as<int>(), max_chunks, s,
                         coarse ? h->cpairctl.as<unsigned int>() : nullptr, 4,
                         will_super ? h->ball_mask.as<unsigned int>() : nullptr, (long long)mask_words,
                         h->rowflag.as<unsigned int>(), (long long)T);
        {
            StageTimer t(h, s, TM_PREP);
            if (bf)
                launch_prepare_queries16b(h->Qp.as<double>(), h->qnorm.as<double>(), T, h->Dt, h->Dpad,
                                          use_gs ? h->gs_fmax2.as<double>() : h->fmax2.as<double>(),
                                          use_gs ? h->gs_rho16.as<double>() : h->rho16.as<double>(), eps_c_run, h->b16l.p,
                                          h->eps16.as<double>(), h->cq16.as<double>(), s, c_coarse, want_thr1 ? h->e1_16.as<double>() : nullptr);
            else
            launch_prepare_queries16(h->Qp.as<double>(), h->qnorm.as<double>(), T, h->Dt, h->Dpad,
                                     use_gs ? h->gs_fmax2.as<double>() : h->fmax2.as<double>(), h->eps_c, h->b16h.p,
                                     h->eps16.as<double>(), s);
        }
        if (!bound_in) {
            StageTimer t(h, s, TM_KNN_MINIMA);
            if (bf)
                launch_knn_sweep16b(0, h->prefilter == 2 ? 4 : 3, nt_run, dch16, p0.grid_cus, use_gs ? h->gs_tiles_b.p : h->s16l.p, h->b16l.p, nullptr, Tpad,
                                    n_slabs_a, h->slabctr.as<unsigned int>(), h->gmin32.as<float>(), G16, nullptr, nullptr,
                                    nullptr, 0, knn_pool_chunk_entries(), s);
            else
            launch_knn_sweep16(0, nt_run, dch16, (h->Dt + 2) / 2, p0.grid_cus, use_gs ? h->gs_tiles.p : h->s16h.p, h->b16h.p, cls_samp, qclass_dev,
                               nullptr, Tpad, n_slabs_a, h->slabctr.as<unsigned int>(), h->gmin32.as<float>(),
                               G16, nullptr, nullptr, nullptr, 0, knn_pool_chunk_entries(), s);
        }
        // stage A': where the ball pass is going to list the tile pairs (compact tiles), the K-th smallest key among the units
        // of the tiles nearest to a row is a second, much tighter bound of its K-th nearest key
        const bool balls = coarse && h->prefilter_balls && h->ball_tiles > 0 && slab_factor == 1 && !h->filter_coarse;
        const bool ball_bound = balls && h->prefilter_ball_bound && !bound_in && knn_scout_keys_per_row() >= K;
        if (ball_bound) {
            StageTimer t(h, s, TM_KNN_BALLMIN);
            const int Gb = knn_scout_groups(Tpad, h->ball_tiles), G2 = knn_scout_keys_per_row();
            CHK(h->ball_gmin.ensure((size_t)Tpad * Gb * sizeof(float)));
            CHK(h->ball_aq.ensure(knn_scout_list_bytes(Tpad)));
            CHK(h->ball_nql.ensure((size_t)Tpad * G2 * sizeof(float)));
            CHK(h->ball_bound.ensure((size_t)Tpad * sizeof(double)));
            launch_knn_scout16b(h->prefilter == 2 ? 4 : 3, dch16, p0.grid_cus, h->ball_c16.p, h->a16l.p, h->b16l.p, T, Tpad, h->ball_tiles,
                                h->ball_gmin.as<float>(), h->ball_aq.as<unsigned int>(), h->ball_nql.as<float>(), s);
            // K-th smallest of the row's keys + eps -> ball_bound (thr / thr32 are written again below)
            launch_knn_threshold16(h->ball_nql.as<float>(), G2, T, Tpad, K, h->eps16.as<double>(), h->thr.as<double>(),
                                   h->thr32.as<float>(), nullptr, h->ball_bound.as<double>(), s);
        }
        // ---- optimistic thresholds (snk_engine.h): which minimum of the sample bounds -- or only estimates -- the K-th nearest key.
        // The sample is every stride-th unit, its groups are scattered over it: at the true K-th key a row has lam = K / stride
        // sample units under it, Poisson-like, so the j-th smallest minimum with j = lam + 9 sqrt(lam) + 4 lies ABOVE the K-th
        // nearest key except with a probability of ~1e-10 per row (K = 100, stride 16: j = 33, lists of ~520 entries instead of
        // ~1 800).  Not a proof: the re-rank proves each row (knn_finalize_kernel, status bit 8) and a flagged call is redone.
        int k_rank = K;
        {
            const bool may = h->tau_optimism && !h->opt_suppress && !h->opt_off && !cls && !bound_in && !bound_out && !refine && !use_gs &&
                             G16 >= K && h->stride16 > 1 && h->N >= 64 * (int64_t)K;
            if (may) {
                const double lam = (double)K / (double)h->stride16;
                int j = h->tau_rank_override > 0 ? h->tau_rank_override : (int)ceil(lam + 9.0 * sqrt(lam) + 4.0);
                if (j < 1) j = 1;
                if (j < K) k_rank = j;
            }
        }
        const bool optimistic = k_rank < K;
        h->opt_last_rank = optimistic ? k_rank : 0;
        if (optimistic) h->opt_calls += 1;
        {
            StageTimer t(h, s, TM_KNN_THRESHOLD);
            launch_knn_threshold16(h->gmin32.as<float>(), G16, T, Tpad, k_rank, h->eps16.as<double>(), h->thr.as<double>(),
                                   h->thr32.as<float>(), bound_in, bound_out, s, want_thr1 ? h->e1_16.as<double>() : nullptr,
                                   want_thr1 ? h->thr1_32.as<float>() : nullptr, ball_bound ? h->ball_bound.as<double>() : nullptr);
        }
        if (bound_out) return 0;             // stage A only
        h->knn_mid_recorded = false;
        // (a voice on the coarse sweep: behind that sweep instead -- below)
        if (((h->join_bounds_delay == 1 && !(coarse && !balls)) || h->join_bounds_delay == 3) && deferred_status) {
            if (!h->knn_mid) HIPCHK(hipEventCreateWithFlags(&h->knn_mid, hipEventDisableTiming));
            HIPCHK(hipEventRecord(h->knn_mid, s));
            h->knn_mid_recorded = true;
        }
        {
            StageTimer t(h, s, TM_KNN_FILTER);
            if (coarse) {
                if (balls) {
                    CHK(h->ball_tq.ensure((size_t)Tpad * sizeof(float)));
                    CHK(h->ball_nq.ensure((size_t)Tpad * sizeof(float)));
                    launch_ball_query_terms(h->thr32.as<float>(), h->eps16.as<double>(), h->qnorm.as<double>(), T, Tpad, h->ball_tq.as<float>(),
                                            h->ball_nq.as<float>(), s);
                    const unsigned int *visit = nullptr;
                    if (h->prefilter_super_balls && h->ball_supers > 0) {
                        // the balls of 32 tiles first: a bit per (super ball, query tile); the tile pass visits the marked blocks
                        const size_t words = (size_t)h->ball_supers * ((Tpad / 32 + 31) / 32);
                        CHK(h->ball_mask.ensure(words * sizeof(unsigned int)));
                        if (!(will_super && words == mask_words)) HIPCHK(hipMemsetAsync(h->ball_mask.p, 0, words * sizeof(unsigned int), s));      // (else: cleared by knn_reset)
                        launch_knn_balls16b(h->prefilter == 2 ? 4 : 3, dch16, p0.grid_cus, h->ball_s16.p, h->b16l.p, h->ball_rad2.as<float>(),
                                            h->ball_tq.as<float>(), h->ball_nq.as<float>(), Tpad, h->ball_supers, nullptr, nullptr, 0u, s,
                                            h->ball_mask.as<unsigned int>(), nullptr);
                        visit = h->ball_mask.as<unsigned int>();
                    }
                    launch_knn_balls16b(h->prefilter == 2 ? 4 : 3, dch16, p0.grid_cus, h->ball_c16.p, h->b16l.p, h->ball_rad.as<float>(),
                                        h->ball_tq.as<float>(), h->ball_nq.as<float>(), Tpad, h->ball_tiles, h->cpairs.p,
                                        h->cpairctl.as<unsigned int>(), pair_cap, s, nullptr, visit);
                    h->ball_limit = h->coarse_gate_fraction * (double)(Tpad / 32) * (double)h->ball_tiles;
                }
                h->ball_pass_ran = balls;
                h->coarse_pass_ran = !balls;
                if (!balls) {
                    // beyond half of all pairs the one-pass sweep is the cheaper filter; the list must not overflow either
                    const double all = (double)(Tpad / 32) * (double)n_tiles_b;
                    h->coarse_limit = h->onepass_gate_fraction * all < 0.9 * (double)pair_cap ? h->onepass_gate_fraction * all : 0.9 * (double)pair_cap;
                }
                if (!balls && (h->join_bounds_delay == 1 || h->join_bounds_delay == 5) && deferred_status) {
                    // the Viterbi side of the group before starts behind this group's COARSE sweep (a persistent whole-database
                    // sweep like stage A: pass 1's workgroups would starve it; behind the thresholds the AR(1) voice lost 6 %,
                    // here it gains 3 %)
                    launch_knn_filter16c(h->prefilter == 2 ? 4 : 3, dch16, p0.grid_cus, h->a16l.p, h->b16l.p, h->thr32.as<float>(), h->thr1_32.as<float>(),
                                         Tpad, n_tiles_b, h->slabctr.as<unsigned int>() + 1, h->cpairs.p, h->cpairctl.as<unsigned int>(), pair_cap,
                                         h->pool.p, h->poolctl.as<unsigned int>(), h->chunkfill.as<int>(), max_chunks, knn_pool_chunk_entries(), s,
                                         true, false);
                    if (!h->knn_mid) HIPCHK(hipEventCreateWithFlags(&h->knn_mid, hipEventDisableTiming));
                    HIPCHK(hipEventRecord(h->knn_mid, s));
                    h->knn_mid_recorded = true;
                    launch_knn_filter16c(h->prefilter == 2 ? 4 : 3, dch16, p0.grid_cus, h->a16l.p, h->b16l.p, h->thr32.as<float>(), h->thr1_32.as<float>(),
                                         Tpad, n_tiles_b, h->slabctr.as<unsigned int>() + 1, h->cpairs.p, h->cpairctl.as<unsigned int>(), pair_cap,
                                         h->pool.p, h->poolctl.as<unsigned int>(), h->chunkfill.as<int>(), max_chunks, knn_pool_chunk_entries(), s,
                                         false, true);
                } else
                launch_knn_filter16c(h->prefilter == 2 ? 4 : 3, dch16, p0.grid_cus, h->a16l.p, h->b16l.p, h->thr32.as<float>(), h->thr1_32.as<float>(),
                                     Tpad, n_tiles_b, h->slabctr.as<unsigned int>() + 1, h->cpairs.p, h->cpairctl.as<unsigned int>(), pair_cap,
                                     h->pool.p, h->poolctl.as<unsigned int>(), h->chunkfill.as<int>(), max_chunks, knn_pool_chunk_entries(), s,
                                     !balls);
            }
            else if (bf)
                launch_knn_sweep16b(1, h->prefilter == 2 ? 4 : 3, nt_run, dch16, p0.grid_cus, h->a16l.p, h->b16l.p, h->thr32.as<float>(), Tpad, n_slabs_b,
                                    h->slabctr.as<unsigned int>() + 1, nullptr, 0, h->pool.p, h->poolctl.as<unsigned int>(),
                                    h->chunkfill.as<int>(), max_chunks, knn_pool_chunk_entries(), s);
            else
            launch_knn_sweep16(1, nt_run, dch16, (h->Dt + 2) / 2, p0.grid_cus, h->a16h.p, h->b16h.p, cls_full, qclass_dev,
                               h->thr32.as<float>(), Tpad, n_slabs_b, h->slabctr.as<unsigned int>() + 1, nullptr, 0,
                               h->pool.p, h->poolctl.as<unsigned int>(), h->chunkfill.as<int>(), max_chunks,
                               knn_pool_chunk_entries(), s);
        }
        if (probe == 1) {
            // the ball pass as a counting probe: pair_cap 0, its count in word 2 of the control block
            CHK(h->ball_tq.ensure((size_t)Tpad * sizeof(float)));
            CHK(h->ball_nq.ensure((size_t)Tpad * sizeof(float)));
            launch_ball_query_terms(h->thr32.as<float>(), h->eps16.as<double>(), h->qnorm.as<double>(), T, Tpad, h->ball_tq.as<float>(),
                                    h->ball_nq.as<float>(), s);
            const unsigned int *visit = nullptr;
            if (h->prefilter_super_balls && h->ball_supers > 0) {
                const size_t words = (size_t)h->ball_supers * ((Tpad / 32 + 31) / 32);
                CHK(h->ball_mask.ensure(words * sizeof(unsigned int)));
                HIPCHK(hipMemsetAsync(h->ball_mask.p, 0, words * sizeof(unsigned int), s));
                launch_knn_balls16b(h->prefilter == 2 ? 4 : 3, dch16, p0.grid_cus, h->ball_s16.p, h->b16l.p, h->ball_rad2.as<float>(),
                                    h->ball_tq.as<float>(), h->ball_nq.as<float>(), Tpad, h->ball_supers, nullptr, nullptr, 0u, s,
                                    h->ball_mask.as<unsigned int>(), nullptr);
                visit = h->ball_mask.as<unsigned int>();
            }
            launch_knn_balls16b(h->prefilter == 2 ? 4 : 3, dch16, p0.grid_cus, h->ball_c16.p, h->b16l.p, h->ball_rad.as<float>(),
                                h->ball_tq.as<float>(), h->ball_nq.as<float>(), Tpad, h->ball_tiles, h->cpairs.p,
                                h->cpairctl.as<unsigned int>() + 2, 0u, s, nullptr, visit);
            h->probe_limit = 0.5 * h->coarse_gate_fraction * (double)(Tpad / 32) * (double)h->ball_tiles;
        } else if (probe == 2) {
            // the coarse sweep as a counting probe (its own slab dispenser: word 2)
            HIPCHK(hipMemsetAsync(h->slabctr.as<unsigned int>() + 2, 0, sizeof(unsigned int), s));
            launch_knn_filter16c(h->prefilter == 2 ? 4 : 3, dch16, p0.grid_cus, h->a16l.p, h->b16l.p, h->thr32.as<float>(), h->thr1_32.as<float>(),
                                 Tpad, n_tiles_b, h->slabctr.as<unsigned int>() + 2, h->cpairs.p, h->cpairctl.as<unsigned int>() + 2, 0u,
                                 nullptr, nullptr, nullptr, 0, knn_pool_chunk_entries(), s, true, false);
            const double all = (double)(Tpad / 32) * (double)n_tiles_b;
            int64_t capp = (int64_t)all / 4 > ((int64_t)4 << 20) ? (int64_t)all / 4 : ((int64_t)4 << 20);
            if ((double)capp > all) capp = (int64_t)all;
            h->probe_limit = 0.5 * (h->onepass_gate_fraction * all < 0.9 * (double)capp ? h->onepass_gate_fraction * all : 0.9 * (double)capp);
        }
        h->probe_ran = probe;
        if (trace_on()) CHK(debug_check_pool(h, max_chunks, Tpad, n_slabs_b * 32 * nt_run, s));
        {
            StageTimer t(h, s, TM_KNN_BUCKET);
            launch_knn_bucket(h->pool.p, h->poolctl.as<unsigned int>(), h->chunkfill.as<int>(), max_chunks,
                              Tpad, h->N, h->cnt.as<int>(), h->lkey.as<double>(), h->lidx.as<int>(), cap, status_dev, s,
                              h->perm_ready ? h->perm.as<int32_t>() : nullptr, true);
        }
        if ((h->join_bounds_delay == 2 || h->join_bounds_delay == 4) && deferred_status) {
            if (!h->knn_mid) HIPCHK(hipEventCreateWithFlags(&h->knn_mid, hipEventDisableTiming));
            HIPCHK(hipEventRecord(h->knn_mid, s));
            h->knn_mid_recorded = true;
        }
        if (refine && bound_in && h->comm_ranks > 1 && h->shard_refine) {
            StageTimer t(h, s, TM_KNN_BUCKET);
            CHK(h->kth16.ensure((size_t)Tpad * sizeof(double)));
            launch_knn_local_kth(h->cnt.as<int>(), h->lkey.as<double>(), cap, K, h->eps16.as<double>(), T, h->kth16.as<double>(), s);
            CHK(comm_all_reduce_min(h, h->kth16.as<double>(), T));
            launch_knn_list_prune(h->cnt.as<int>(), h->lkey.as<double>(), h->lidx.as<int>(), cap, h->kth16.as<double>(),
                                  h->eps16.as<double>(), T, s);
        }
        {
            StageTimer t(h, s, TM_KNN_FINALIZE);
            launch_knn_finalize(h->Fw.as<double>(), h->F_unw.as<float>(), h->Fp, h->wt.as<double>(), h->Dpad, h->Dt, h->Qp.as<double>(), h->qnorm.as<double>(), T, K,
                                h->cnt.as<int>(), h->lkey.as<double>(), h->lidx.as<int>(), cap,
                                h->shard_offset, h->eps16.as<double>(), h->fnorm.as<double>(), eps_c_run, bf ? h->cq16.as<double>() : nullptr, cand_dev, dist_dev, d2_dev, status_dev, nullptr, s,
                                bound_in != nullptr,         // a shard's lists under the shared bound are short
                                bound_in ? nullptr : h->thr.as<double>(), h->margin_stat.as<unsigned int>(), h->rowflag.as<int>(),
                                h->knn_level >= 1, optimistic, true);
        }
        if (deferred_status) {               // the batch caller redoes failures with precision 0
            // (and learns how many tile pairs the ball pass listed)
            if (pairs_listed_dev) {
                if (coarse) HIPCHK(hipMemcpyAsync(pairs_listed_dev, h->cpairctl.p, sizeof(unsigned int), hipMemcpyDeviceToDevice, s));
                else HIPCHK(hipMemsetAsync(pairs_listed_dev, 0, sizeof(unsigned int), s));
            }
            if (probe_listed_dev) {
                if (probe) HIPCHK(hipMemcpyAsync(probe_listed_dev, h->cpairctl.as<unsigned int>() + 2, sizeof(unsigned int), hipMemcpyDeviceToDevice, s));
                else HIPCHK(hipMemsetAsync(probe_listed_dev, 0xff, sizeof(unsigned int), s));
            }
            return 0;
        }
        int status = 0;
        {
            unsigned int ctl[4] = {0u, 0u, 0u, 0u};
            D2HPart parts[2] = {{&status, h->status.p, sizeof(int)}, {ctl, h->cpairctl.p, (coarse || probe) ? sizeof(ctl) : 0}};
            CHK(staged_d2h(h, s, parts, 2));
            judge_filter(h, h->ball_pass_ran, h->ball_limit, h->coarse_pass_ran, h->coarse_limit, ctl[0], probe, h->probe_limit, ctl[2]);
        }
        HIPCHK(hipGetLastError());
        h->last_f16_status = status;
        if (status == 0) return 0;
        // (bit 8 beside another bit: an overflowed pair list, pool or row list left rows unproven -- that bit's handling below is the
        // remedy, not the thresholds' fault)
        if (status != 8) status &= ~8;
        if (status == 8) {
            // a row's list was not proven complete under the optimistic thresholds: the call once more with the guaranteed ones
            note_optimism_failure(h);
            struct Guard { snk_engine *e; bool was; ~Guard() { e->opt_suppress = was; } } guard{h, h->opt_suppress};
            h->opt_suppress = true;
            return knn_device(h, Qdev, T, K, qclass_dev, cand_dev, dist_dev, d2_dev, nullptr, bound_in, nullptr, gs, refine, nullptr, nullptr);
        }
        if ((status & ~3) == 0 && h->knn_level < 2 && !refine) {
            // a candidate list overflowed, or a row held more near ties than the exact re-rank takes, and nothing else went wrong:
            // the next rung of the voice's ladder (longer lists + a re-rank tier for 8 192 ties, then the float32 operands) serves
            // this call and every later one
            h->knn_level += 1; h->knn_escalations += 1;
            return knn_device(h, Qdev, T, K, qclass_dev, cand_dev, dist_dev, d2_dev, nullptr, bound_in, nullptr, gs, false, nullptr, nullptr);
        }
        h->f16_fallbacks += 1;               // overflow or too many near ties: exact f64 sweep below
    }

    if (bound_out) {                          // no f32 path for this shape: no bound, nothing is pruned
        launch_fill_threshold(bound_out, T, T, DBL_MAX, s);
        return 0;
    }
    if (refine && bound_in && h->comm_ranks > 1 && h->shard_refine) {
        // this rank's shard has no prefilter lists (shape without a variant): it still takes part in the other
        // ranks' all-reduce of the second bound, contributing nothing
        CHK(h->kth16.ensure((size_t)Tpad * sizeof(double)));
        launch_fill_threshold(h->kth16.as<double>(), T, T, DBL_MAX, s);
        CHK(comm_all_reduce_min(h, h->kth16.as<double>(), T));
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        // attempt 0: thresholds from a strided sample of slabs (stage A).
        // attempt 1 (a candidate list overflowed): stage A over EVERY slab -- at most
        //   nt*K database rows then lie under each threshold, which the lists always hold.
        launch_knn_reset(h->cnt.as<int>(), Tpad, status_dev, h->poolctl.as<unsigned int>(),
                         h->slabctr.as<unsigned int>(), h->chunkfill.as<int>(), max_chunks, s);
        if (attempt == 1) {
            p.a_stride = 1; p.a_count = p.n_slabs;
            G = p.a_count * 16;
            CHK(h->gmin.ensure((size_t)Tpad * G * sizeof(double)));
            h->last_retries = 1;
            // the first attempt may have exhausted the entry pool (mass ties at the thresholds): give the
            // retry room for a full list per row, so that only the lists themselves can still overflow
            const int64_t full = (Tpad * (int64_t)cap) / knn_pool_chunk_entries() + 4096;
            if (full > max_chunks && h->pool_chunk_limit <= 0) {
                max_chunks = (int)full;
                CHK(h->pool.ensure(knn_pool_bytes(max_chunks)));
                CHK(h->chunkfill.ensure((size_t)max_chunks * sizeof(int)));
            }
        }
        if (G >= K) {       // tiny databases: fewer than K groups cannot bound the K-th neighbour
            {
                StageTimer t(h, s, TM_KNN_MINIMA);
                launch_knn_minima(p, h->Fw.as<double>(), h->fnorm.as<double>(), h->Qf.as<double>(), Tpad,
                                  h->gmin.as<double>(), G, uc, qclass_dev, s);
            }
            {
                StageTimer t(h, s, TM_KNN_THRESHOLD);
                launch_knn_threshold(h->gmin.as<double>(), G, T, Tpad, K, h->thr.as<double>(), attempt, s);
            }
        } else {
            launch_fill_threshold(h->thr.as<double>(), T, Tpad, DBL_MAX, s);
        }
        {
            StageTimer t(h, s, TM_KNN_FILTER);
            launch_knn_filter(p, h->Fw.as<double>(), h->fnorm.as<double>(), h->Qf.as<double>(),
                              h->thr.as<double>(), Tpad, h->pool.p, h->poolctl.as<unsigned int>(),
                              h->chunkfill.as<int>(), max_chunks, uc, qclass_dev, s);
        }
        {
            StageTimer t(h, s, TM_KNN_BUCKET);
            launch_knn_bucket(h->pool.p, h->poolctl.as<unsigned int>(), h->chunkfill.as<int>(), max_chunks,
                              Tpad, h->N, h->cnt.as<int>(), h->lkey.as<double>(), h->lidx.as<int>(), cap,
                              status_dev, s);
        }
        {
            StageTimer t(h, s, TM_KNN_FINALIZE);
            launch_knn_finalize(h->Fw.as<double>(), h->F_unw.as<float>(), h->Fp, h->wt.as<double>(), h->Dpad, h->Dt, h->Qp.as<double>(), h->qnorm.as<double>(), T, K,
                                h->cnt.as<int>(), h->lkey.as<double>(), h->lidx.as<int>(), cap,
                                h->shard_offset, nullptr, nullptr, 0.0, nullptr, cand_dev, dist_dev, d2_dev, status_dev, h->rowflag.as<int>(), s);
        }
        if (deferred_status) return 0;
        int status = 0;
        CHK(d2h_sync(h, &status, h->status.p, sizeof(int), s));
        HIPCHK(hipGetLastError());
        if (status == 0) return 0;
        if ((status & 5) && attempt == 0) continue;          // a list or the pool overflowed: exact thresholds next
        // Rows the list pipeline cannot serve: more units tied with (or within rounding of) the K-th
        // neighbour than a list or the exact re-rank holds -- mass duplicates.  They get the
        // one-workgroup-per-row exact selection (slow, exact, ties by lowest id).  If even the enlarged
        // pool overflowed, the sweep dropped entries of rows that cannot be told apart: every row goes.
        std::vector<int> flags((size_t)T);
        CHK(d2h_sync(h, flags.data(), h->rowflag.p, (size_t)T * sizeof(int), s));
        std::vector<int> rows;
        for (int64_t t = 0; t < T; ++t) if (flags[(size_t)t] || (status & 4)) rows.push_back((int)t);
        if (status & 4) h->pool_overflows += 1;
        if (status & 2) h->tie_overflow = 1;
        for (size_t r0 = 0; r0 < rows.size(); r0 += 64) {    // 64 rows (x Nalloc doubles of scratch) at a time
            const int n = (int)((rows.size() - r0 < 64) ? rows.size() - r0 : 64);
            CHK(h->exact_rows.ensure((size_t)64 * sizeof(int)));
            CHK(h->exact_scratch.ensure((size_t)64 * h->Nalloc * sizeof(double)));
            CHK(h2d(h, h->exact_rows.p, rows.data() + r0, (size_t)n * sizeof(int), s));
            launch_knn_exact_rows(h->Fw.as<double>(), h->Dpad, h->Dt, h->N, h->Qp.as<double>(), h->exact_rows.as<int>(), n, K,
                                  h->exact_scratch.as<double>(), h->Nalloc, uc, qclass_dev, h->shard_offset,
                                  cand_dev, dist_dev, d2_dev, s);
            HIPCHK(hipStreamSynchronize(s));
        }
        HIPCHK(hipGetLastError());
        h->exact_row_fallbacks += (int)rows.size();
        return 0;
    }
    return fail("K-NN: internal error (attempt loop fell through)");
}

int upload_queries(snk_engine *h, const double *Q, int64_t T, int D)
{
    if (!Q) return fail("null query matrix");
    if (D != h->Dt) return fail("query matrix has %d columns, database has %d", D, h->Dt);
    if (T < 1) return fail("query matrix has no rows");
    CHK(h->Qraw.ensure((size_t)T * D * sizeof(double)));
    StageTimer t(h, h->stream, TM_H2D);
    CHK(h2d(h, h->Qraw.p, Q, (size_t)T * D * sizeof(double), h->stream));
    if (!h->tsel.empty()) launch_mask_columns(h->Qraw.as<double>(), T, D, h->tmask.as<double>(), h->stream);
    return 0;
}

int snk_knn(snk_handle h, const double *Q, int64_t T, int D, int K, int64_t *cand_out, double *dist_out)
{
    CHK(check_ready(h, true, false));
    CHK(no_batch_in_flight(h, "snk_knn"));
    HIPCHK(hipSetDevice(h->device));
    if (!cand_out || !dist_out) return fail("snk_knn: null output");
    CHK(upload_queries(h, Q, T, D));
    UttSlot &s = h->slot[0];
    CHK(s.cand.ensure((size_t)T * K * sizeof(int64_t)));
    CHK(s.tdist.ensure((size_t)T * K * sizeof(double)));
    {
        const int rc = knn_device(h, h->Qraw.as<double>(), T, K, nullptr, s.cand.as<int64_t>(), s.tdist.as<double>(), nullptr);
        if (rc) { (void)hipStreamSynchronize(h->stream); collect_timers(h); return rc; }
    }
    {
        StageTimer t(h, h->stream, TM_D2H);
        D2HPart parts[2] = {{cand_out, s.cand.p, (size_t)T * K * sizeof(int64_t)},
                            {dist_out, s.tdist.p, (size_t)T * K * sizeof(double)}};
        CHK(staged_d2h(h, h->stream, parts, 2));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    collect_timers(h);
    return 0;
}

// Diagnostic of the prefilter's error bound (include/snk.h): the prefilter's minimum key of every (query row, slab of
// rows_per_slab consecutive units) pair and the bound eps[t] it is trusted to; the caller compares with float64 keys.
int snk_prefilter_minima(snk_handle h, const double *Q, int64_t T, int D, float *slab_min, int64_t slab_min_len,
                         double *eps_out, int64_t *n_slabs_out, int *rows_per_slab_out)
{
    CHK(check_ready(h, true, false));
    CHK(no_batch_in_flight(h, "snk_prefilter_minima"));
    HIPCHK(hipSetDevice(h->device));
    if (!h->f16_ready) return fail("snk_prefilter_minima: this database shape has no float32 / bf16 prefilter");
    if (h->perm_ready) return fail("snk_prefilter_minima: the engine gave this voice an order of its own (slabs are not runs of consecutive units; option reorder 0)");
    if (T < 1 || T > SNK_KNN_MAX_ROWS) return fail("snk_prefilter_minima: T outside 1..%d", (int)SNK_KNN_MAX_ROWS);
    const int64_t n_slabs = h->n_slabs16;
    if (n_slabs_out) *n_slabs_out = n_slabs;
    if (rows_per_slab_out) *rows_per_slab_out = 32 * h->nt16_eff;
    if (!slab_min) return 0;                   // size query
    if (slab_min_len < T * n_slabs || !eps_out) return fail("snk_prefilter_minima: output too small");
    CHK(upload_queries(h, Q, T, D));
    const int64_t Tpad = roundup(T, 32), G16 = 2 * n_slabs;
    const bool bf = h->bf16_ready && h->prefilter >= 1;
    const int dch16 = h->Dpad / 64;
    hipStream_t s = h->stream;
    CHK(h->Qp.ensure((size_t)Tpad * h->Dpad * sizeof(double)));
    CHK(h->Qf.ensure((size_t)Tpad * h->Dpad * sizeof(double)));
    CHK(h->qnorm.ensure((size_t)Tpad * sizeof(double)));
    CHK((bf ? h->b16l : h->b16h).ensure((size_t)(Tpad / 32) * 8 * 64 * 16 * dch16));
    CHK(h->eps16.ensure((size_t)Tpad * sizeof(double)));
    CHK(h->gmin32.ensure((size_t)Tpad * G16 * sizeof(float)));
    CHK(h->slabctr.ensure(4 * sizeof(unsigned int)));
    HIPCHK(hipMemsetAsync(h->slabctr.p, 0, 4 * sizeof(unsigned int), s));
    launch_prepare_queries(h->Qraw.as<double>(), T, h->Dt, h->Qp.as<double>(), h->Qf.as<double>(), h->qnorm.as<double>(),
                           Tpad, h->Dpad, s);
    const KnnPlan p0 = make_plan(h, 1);
    if (bf) {
        CHK(h->cq16.ensure((size_t)Tpad * sizeof(double)));
        launch_prepare_queries16b(h->Qp.as<double>(), h->qnorm.as<double>(), T, h->Dt, h->Dpad, h->fmax2.as<double>(),
                                  h->rho16.as<double>(), h->eps_c_bf, h->b16l.p, h->eps16.as<double>(), h->cq16.as<double>(), s);
        StageTimer tm(h, s, TM_KNN_MINIMA);
        launch_knn_sweep16b(0, h->prefilter == 2 ? 4 : 3, h->nt16_eff, dch16, p0.grid_cus, h->a16l.p, h->b16l.p, nullptr, Tpad, n_slabs,
                            h->slabctr.as<unsigned int>(), h->gmin32.as<float>(), G16, nullptr, nullptr, nullptr, 0,
                            knn_pool_chunk_entries(), s);
    } else {
        launch_prepare_queries16(h->Qp.as<double>(), h->qnorm.as<double>(), T, h->Dt, h->Dpad, h->fmax2.as<double>(),
                                 h->eps_c, h->b16h.p, h->eps16.as<double>(), s);
        launch_knn_sweep16(0, h->nt16_eff, dch16, (h->Dt + 2) / 2, p0.grid_cus, h->a16h.p, h->b16h.p, nullptr, nullptr,
                           nullptr, Tpad, n_slabs, h->slabctr.as<unsigned int>(), h->gmin32.as<float>(), G16, nullptr,
                           nullptr, nullptr, 0, knn_pool_chunk_entries(), s);
    }
    HIPCHK(hipGetLastError());
    std::vector<float> g((size_t)T * G16);
    {
        D2HPart parts[2] = {{g.data(), h->gmin32.p, g.size() * sizeof(float)}, {eps_out, h->eps16.p, (size_t)T * sizeof(double)}};
        CHK(staged_d2h(h, s, parts, 2));
    }
    collect_timers(h);
    for (int64_t t = 0; t < T; ++t)
        for (int64_t w = 0; w < n_slabs; ++w) {
            const float a = g[t * G16 + 2 * w], b = g[t * G16 + 2 * w + 1];
            slab_min[t * n_slabs + w] = a < b ? a : b;
        }
    return 0;
}

int snk_knn_by_class(snk_handle h, const double *Q, int64_t T, int D, int K, const int32_t *query_class,
                     int64_t *cand_out, double *dist_out)
{
    CHK(check_ready(h, true, false));
    CHK(no_batch_in_flight(h, "snk_knn_by_class"));
    HIPCHK(hipSetDevice(h->device));
    if (!h->have_classes) return fail("snk_knn_by_class: unit classes not set (snk_set_unit_classes)");
    if (!query_class || !cand_out || !dist_out) return fail("snk_knn_by_class: null argument");
    CHK(upload_queries(h, Q, T, D));
    const int64_t Tpad = roundup(T, 32);
    CHK(h->qclass.ensure((size_t)Tpad * sizeof(int32_t)));
    HIPCHK(hipMemsetAsync(h->qclass.p, 0xfe, (size_t)Tpad * sizeof(int32_t), h->stream));
    CHK(h2d(h, h->qclass.p, query_class, (size_t)T * sizeof(int32_t), h->stream));
    UttSlot &s = h->slot[0];
    CHK(s.cand.ensure((size_t)T * K * sizeof(int64_t)));
    CHK(s.tdist.ensure((size_t)T * K * sizeof(double)));
    CHK(knn_device(h, h->Qraw.as<double>(), T, K, h->qclass.as<int32_t>(), s.cand.as<int64_t>(), s.tdist.as<double>(), nullptr));
    {
        D2HPart parts[2] = {{cand_out, s.cand.p, (size_t)T * K * sizeof(int64_t)}, {dist_out, s.tdist.p, (size_t)T * K * sizeof(double)}};
        CHK(staged_d2h(h, h->stream, parts, 2));
    }
    collect_timers(h);
    return 0;
}

int snk_candidate_distances(snk_handle h, const double *Q, int64_t T, int D, const int64_t *cand, int K,
                            double *dist_out)
{
    CHK(check_ready(h, true, false));
    HIPCHK(hipSetDevice(h->device));
    if (!cand || !dist_out || K < 1) return fail("snk_candidate_distances: null/empty argument");
    CHK(upload_queries(h, Q, T, D));
    const int64_t Tpad = roundup(T, 16);
    CHK(h->Qp.ensure((size_t)Tpad * h->Dpad * sizeof(double)));
    CHK(h->Qf.ensure((size_t)Tpad * h->Dpad * sizeof(double)));
    CHK(h->qnorm.ensure((size_t)Tpad * sizeof(double)));
    UttSlot &s = h->slot[0];
    CHK(s.cand.ensure((size_t)T * K * sizeof(int64_t)));
    CHK(s.tdist.ensure((size_t)T * K * sizeof(double)));
    launch_prepare_queries(h->Qraw.as<double>(), T, h->Dt, h->Qp.as<double>(), h->Qf.as<double>(), h->qnorm.as<double>(),
                           Tpad, h->Dpad, h->stream);
    CHK(h2d(h, s.cand.p, cand, (size_t)T * K * sizeof(int64_t), h->stream));
    launch_candidate_dist(h->Fw.as<double>(), h->Dpad, h->Dt, h->N, h->Qp.as<double>(), s.cand.as<int64_t>(), T, K,
                          s.tdist.as<double>(), h->stream);
    HIPCHK(hipGetLastError());
    D2HPart parts[1] = {{dist_out, s.tdist.p, (size_t)T * K * sizeof(double)}};
    CHK(staged_d2h(h, h->stream, parts, 1));
    collect_timers(h);
    return 0;
}
