// C ABI of libsnkhip.so, part 4: greedy search (greedy_joint_search, synth_simple.py:458-503) and per-stream path scores.
#include "snk_engine.h"

// ---------------------------------------------------------------------------
// greedy search
// ---------------------------------------------------------------------------
int snk_set_greedy_layout(snk_handle h, int multiepoch, int last_frame_as_target, int join_split_mode)
{
    if (!h) return fail("null handle");
    if (!h->have_db || !h->have_join) return fail("snk_set_greedy_layout: upload the database first");
    if (multiepoch < 1 || multiepoch > 16) return fail("snk_set_greedy_layout: multiepoch=%d outside 1..16", multiepoch);
    if (join_split_mode != 0 && join_split_mode != 1) return fail("snk_set_greedy_layout: join_split_mode must be 0 or 1");
    if (join_split_mode == 1 && (h->Dj % 2)) return fail("snk_set_greedy_layout: join_split_mode 1 needs an even number of join columns");
    const int64_t Nrep = h->Njc - 1;            // rows of unit_start_data / unit_end_data
    if (Nrep != h->N) return fail("snk_set_greedy_layout: join_contexts rows (%lld) != N+1", (long long)h->Njc);
    if (h->N < multiepoch) return fail("snk_set_greedy_layout: database shorter than one multiepoch window");
    GreedyLayout g{};
    g.me = multiepoch;
    g.last_frame_as_target = last_frame_as_target ? 1 : 0;
    g.join_split_mode = join_split_mode;
    g.Nwin = h->N - multiepoch + 1;
    if (join_split_mode == 0) {
        // prev = unit_start_data = JC[:-1] (row i); current = unit_end_data = JC[1:], shifted by
        // the multiepoch overlap (synth_simple.py:194-195,213-214): window i -> JC row i + me
        g.jdim = h->Dj; g.prev_col0 = 0; g.cur_col0 = 0; g.prev_row0 = 0; g.cur_row0 = multiepoch;
    } else {
        // synth_halfphone.py:552-553: halves of the unit_start_data columns
        g.jdim = h->Dj / 2; g.prev_col0 = 0; g.cur_col0 = h->Dj / 2; g.prev_row0 = 0; g.cur_row0 = multiepoch - 1;
    }
    if (!greedy_supported(g, h->Dt))
        return fail("snk_set_greedy_layout: too many scan columns for the greedy step's table (join %d + %d epochs x %d)", g.jdim, multiepoch, h->Dt);
    h->glay = g;
    h->have_glay = true;
    h->gtiles_ready = false; h->gt16_ready = false;
    h->gh_ready = false; h->gj_ready = false;
    return 0;
}

// Up to greedy32_max_utts() utterances through the float32 persistent scan (greedy32_kernels.hip).  Returns in
// *undecided whether the launch stopped at a step it could not decide (the caller then runs the exact scan).
static int greedy32_group(snk_engine *h, int nu, const int64_t *q_off, const int64_t *ns, const int64_t *oo, const int64_t *st,
                          bool approx, bool want_dist, bool *undecided)
{
    const GreedyLayout &g = h->glay;
    // the target term of all steps as one matrix product per utterance; the scan then streams the join columns only
    G32Hoist hst{};
    bool hoist = h->greedy_hoist && greedy_hoist_supported(g, h->Dt);
    if (hoist) {
        const int64_t Wp = greedy_hoist_pitch(g);
        const int KA = greedy_hoist_k(g, h->Dt);
        int64_t rows = 0, prows = 0;
        for (int u = 0; u < nu; ++u) { rows += ns[u]; prows += greedy_hoist_rows(ns[u]); }
        const double w_bytes = (double)rows * (double)Wp * 4.0;
        if (w_bytes > h->greedy_hoist_max_gb * 1e9) hoist = false;
        if (hoist && w_bytes > (double)h->gh_W.bytes) {
            // the product must fit beside the voice: a device that cannot hold it keeps the scan that computes the target term itself
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || 1.125 * w_bytes - (double)h->gh_W.bytes > 0.9 * (double)free_b) hoist = false;      // (ensure() frees, then asks for 9/8)
        }
        if (hoist) {
            if (!h->gh_ready) {
                CHK(h->gh_nw.ensure((size_t)Wp * sizeof(double)));
                CHK(h->gh_max.ensure(64));
                launch_hoist_window_norms(g, h->fnorm.as<double>(), h->gh_nw.as<double>(), h->gh_max.as<unsigned long long>(), h->stream);
                HIPCHK(hipGetLastError());
                CHK(d2h_sync(h, &h->gh_fwmax2, h->gh_max.p, sizeof(double), h->stream));
                h->gh_ready = true;
            }
            // float16 join tiles: databases that are streamed from HBM (scans beyond 192 MB; or forced), up to three utterances
            // per scan.  Decided before the product: such a scan takes the target values from the bf16 pipe
            bool scan16 = false;
            if (h->greedy_f16 && nu <= 3) {
                CHK(h->gh_max.ensure(64));
                if (!h->gt16_ready) {
                    // once per database and layout: the float16 tiles (refused if a value leaves the float16 range)
                    CHK(h->gtiles16.ensure(greedy_tile16_bytes(g)));
                    unsigned int *mx = reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(h->gh_max.p) + 32);
                    launch_greedy_tiles16(g, h->JC_unw.as<float>(), h->Jp, h->gtiles16.p, mx, h->stream);
                    HIPCHK(hipGetLastError());
                    float mabs = 0.f;
                    CHK(d2h_sync(h, &mabs, mx, sizeof(float), h->stream));
                    h->gt16_ok = mabs < 6.0e4f;
                    h->gt16_ready = true;
                }
                if (h->gt16_ok && !h->gj_ready) {
                    // once per set of weights: max ||w o S'[i]||^2 and ||w||^2
                    unsigned long long *o2 = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(h->gh_max.p) + 16);
                    launch_greedy_join_norms(g, h->gtiles.as<float>(), h->wj.as<double>(), o2, h->stream);
                    HIPCHK(hipGetLastError());
                    double v[2] = {0.0, 0.0};
                    CHK(d2h_sync(h, v, o2, sizeof(v), h->stream));
                    h->g16_delta = 4.8828125e-4 * sqrt(v[0]) + 2.98023223876953125e-8 * sqrt(v[1]);
                    h->gj_ready = true;
                }
                if (h->gt16_ok) {
                    hst.JT16 = h->gtiles16.p; hst.f16_delta = h->g16_delta; hst.f16_force = h->greedy_f16 == 2;
                    // (launch_greedy32 takes the float16 tiles under the same condition)
                    scan16 = hst.f16_force || (double)g.Nwin * (double)(g.jdim + 1) * 4.0 > (double)((size_t)192 << 20);
                    if (scan16) h->greedy_f16_launches += 1;
                }
            }
            const bool fast = scan16 && h->greedy_hoist_fast && greedy_hoist16_supported(g, h->Dt);
            CHK(h->gh_aq.ensure((size_t)prows * KA * sizeof(double)));
            CHK(h->gh_qn2.ensure((size_t)prows * sizeof(double)));
            CHK(h->gh_W.ensure((size_t)rows * (size_t)Wp * sizeof(float)));
            int64_t r0 = 0, p0 = 0;
            for (int u = 0; u < nu; ++u) {
                float *W = h->gh_W.as<float>() + (size_t)r0 * Wp;
                double *qn2 = h->gh_qn2.as<double>() + p0;
                (fast ? launch_hoist_product16 : launch_hoist_product)(g, h->F_unw.as<float>(), h->Fp, h->N, h->Dt, h->wt.as<double>(),
                                     h->Qraw.as<double>(), q_off[u], ns[u], h->gh_nw.as<double>(), h->gh_aq.as<double>() + (size_t)p0 * KA, qn2, W, h->stream);
                hst.W[u] = W; hst.qn2[u] = qn2;
                r0 += ns[u]; p0 += greedy_hoist_rows(ns[u]);
            }
            HIPCHK(hipGetLastError());
            hst.Wp = Wp; hst.c = fast ? greedy_hoist_c16(g, h->Dt) : greedy_hoist_c(g, h->Dt); hst.fwmax2 = h->gh_fwmax2;
            if (fast) h->greedy_hoist16_launches += 1;
            h->greedy_hoist_launches += 1;
        }
    }
    if (!hoist && nu > greedy32_max_utts(false)) {
        // without the product a scan serves three utterances: two launches
        const int n1 = greedy32_max_utts(false);
        bool u1 = false, u2 = false;
        CHK(greedy32_group(h, n1, q_off, ns, oo, st, approx, want_dist, &u1));
        CHK(greedy32_group(h, nu - n1, q_off + n1, ns + n1, oo + n1, st + n1, approx, want_dist, &u2));
        *undecided = u1 || u2;
        return 0;
    }
    const int nblk = greedy32_blocks(g, h->Dt, h->n_cus, hoist);
    CHK(h->g32_blk.ensure(greedy32_block_bytes(nblk)));
    CHK(h->g32_ctl.ensure(256));
    unsigned int *gen = h->g32_ctl.as<unsigned int>();
    int64_t *status = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(h->g32_ctl.p) + 16);
    // one utterance against a database that fits the chip's LDS: resident scan, every workgroup decides for itself
    // (float16 tiles forced on by the caller -- tests -- are the streamed scan's)
    const bool resident = hoist && nu == 1 && h->greedy_resident && greedy_res_supported(g, h->Dt, h->n_cus) && !hst.f16_force;
    if (resident) {
        CHK(h->g32_res.ensure(greedy_res_record_bytes(g) + 256));
        launch_greedy_res(g, h->F_unw.as<float>(), h->Fp, h->Dt, h->wt.as<double>(), h->JC_unw.as<float>(), h->Jp, h->Dj,
                          h->wj.as<double>(), h->gtiles.as<float>(), h->Qraw.as<double>(), q_off[0], ns[0], oo[0], st[0],
                          (approx ? 1 : 0) | (h->greedy_test_stall ? 256 : 0) | (h->greedy_fenced ? 512 : 0), h->g32_res.p, status, h->gpath.as<int64_t>(), &hst, h->stream);
        h->greedy_resident_launches += 1;
    } else
    launch_greedy32(g, h->F_unw.as<float>(), h->Fp, h->Dt, h->wt.as<double>(), h->JC_unw.as<float>(), h->Jp, h->Dj,
                    h->wj.as<double>(), h->gtiles.as<float>(), h->Qraw.as<double>(), nu, q_off, ns, oo, st, (approx ? 1 : 0) | (h->greedy_test_stall ? 256 : 0) | (h->greedy_fenced ? 512 : 0) | (h->greedy_speculate ? 0 : 2048),
                    h->g32_blk.p, h->n_cus, gen,
                    status, h->gpath.as<int64_t>(), hoist ? &hst : nullptr, h->stream);
    HIPCHK(hipGetLastError());
    int64_t stv[16] = {0};         // undecided step + 1 | second-phase rounds | windows decided by exact totals | watchdog | [4..7] per kernel | [8], [9] the bound's tripwire
    CHK(d2h_sync(h, stv, status, sizeof(stv), h->stream));
    greedy32_trace_dump();
    greedy_res_trace_dump();
    *undecided = stv[0] != 0;
    for (int i = 0; i < 16; ++i) h->greedy_last_status[i] = stv[i];
    h->greedy_last_kernel = resident ? 2 : 1;
    h->greedy_bound_violations += stv[8];
    { float uf; const unsigned int ub = (unsigned int)(unsigned long long)stv[9]; memcpy(&uf, &ub, 4); if ((double)uf > h->greedy_bound_max_used) h->greedy_bound_max_used = (double)uf; }
    h->greedy_second_rounds += stv[1];
    h->greedy_exact_windows += stv[2];
    h->greedy_stalls += stv[3];
    if (!*undecided && want_dist) {
        for (int u = 0; u < nu; ++u)
            launch_greedy32_dist(g, h->F_unw.as<float>(), h->Fp, h->Dt, h->wt.as<double>(), h->JC_unw.as<float>(), h->Jp, h->Dj,
                                 h->wj.as<double>(), h->Qraw.as<double>(), u, q_off[u], ns[u], oo[u], st[u], h->gpath.as<int64_t>(),
                                 h->gdist.as<double>(), h->stream);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// greedy_mode 2 (default): the float32 scan where it pays -- always with the hoisted target term (the scan streams the
// join columns only: 202 against 262 us per step at 1.5 M units, 22 against 28 at 65 536); without it, when several
// utterances share every pass over the database (three per scan); a single utterance then goes through the exact
// scan.  Same paths and distances either way.
static bool use_greedy32(const snk_engine *h, int n_utts)
{
    if (!greedy32_supported(h->glay, h->Dt)) return false;
    const bool hoist = h->greedy_hoist && greedy_hoist_supported(h->glay, h->Dt);
    return h->greedy_mode == 1 || (h->greedy_mode == 2 && (n_utts >= 2 || hoist));
}

int snk_greedy(snk_handle h, const double *Q, int64_t T, int D, int64_t start_state, double eps,
               int64_t *path_out, double *dist_out, int64_t *nsteps_out)
{
    CHK(check_ready(h, true, true));
    CHK(no_batch_in_flight(h, "snk_greedy"));
    HIPCHK(hipSetDevice(h->device));
    if (!h->have_glay) return fail("snk_greedy: greedy layout not set (snk_set_greedy_layout)");
    if (!path_out || !nsteps_out) return fail("snk_greedy: null output");
    if (!(eps >= 0.0)) return fail("snk_greedy: search_epsilon must be >= 0");
    const GreedyLayout &g = h->glay;
    if (start_state >= g.Nwin) return fail("snk_greedy: start_state %lld out of range", (long long)start_state);
    CHK(upload_queries(h, Q, T, D));
    const int64_t nsteps = T / g.me;          // py2 integer division: tail frames dropped
    *nsteps_out = nsteps;
    if (nsteps == 0) { HIPCHK(hipStreamSynchronize(h->stream)); collect_timers(h); return 0; }
    if (!h->gtiles_ready) {
        // the scan reads a lane-major copy of its columns (greedy_kernels.hip); built on first use
        CHK(h->gtiles.ensure(greedy_tile_bytes(g, h->Dt)));
        launch_greedy_tiles(g, h->F_unw.as<float>(), h->Fp, h->Dt, h->JC_unw.as<float>(), h->Jp, h->gtiles.as<float>(), h->stream);
        HIPCHK(hipGetLastError());
        h->gtiles_ready = true;
    }
    const int nblk = greedy_blocks(g, h->Dt, h->n_cus);
    CHK(h->gprev.ensure(2 * greedy_table_doubles(g, h->Dt) * sizeof(double) + 512));   // + slack: the scan warms whole 512-byte spans     // (weight, reference) tables
    CHK(h->gsync.ensure(greedy_counter_bytes()));
    CHK(h->gblkmin.ensure((size_t)nblk * sizeof(double)));
    CHK(h->gblkarg.ensure((size_t)nblk * sizeof(int64_t)));
    CHK(h->gpath.ensure((size_t)nsteps * sizeof(int64_t)));
    CHK(h->gdist.ensure((size_t)nsteps * sizeof(double)));
    bool exact_scan = true;
    if (use_greedy32(h, 1)) {
        // float32 prefilter scan, one persistent launch; search_epsilon >= 1e-3: the float32 minimum is the answer
        const int64_t zero = 0;
        bool undecided = false;
        {
            StageTimer t(h, h->stream, TM_GREEDY_STEPS);
            CHK(greedy32_group(h, 1, &zero, &nsteps, &zero, &start_state, eps >= 1e-3, dist_out != nullptr, &undecided));
        }
        exact_scan = undecided;
        if (undecided) h->greedy_fallbacks += 1;
    }
    if (exact_scan) {
        StageTimer t(h, h->stream, TM_GREEDY_STEPS);
        launch_greedy(g, h->F_unw.as<float>(), h->Fp, h->Dt, h->wt.as<double>(), h->JC_unw.as<float>(), h->Jp, h->Dj,
                      h->wj.as<double>(), h->gtiles.as<float>(), h->Qraw.as<double>(), nsteps, start_state, h->gprev.as<double>(),
                      h->gblkmin.as<double>(), h->gblkarg.as<int64_t>(), nblk, h->n_cus, h->gsync.as<unsigned int>(), h->gpath.as<int64_t>(),
                      h->gdist.as<double>(), h->stream);
    }
    HIPCHK(hipGetLastError());
    {
        StageTimer t(h, h->stream, TM_D2H);
        D2HPart parts[2] = {{path_out, h->gpath.p, (size_t)nsteps * sizeof(int64_t)},
                            {dist_out, h->gdist.p, dist_out ? (size_t)nsteps * sizeof(double) : 0}};
        CHK(staged_d2h(h, h->stream, parts, 2));
    }
    collect_timers(h);
    return 0;
}

// Several utterances through the greedy search together: up to three share every scan of the database
// (one weighted value per column and window, one comparison per utterance), so the database is read
// once per step for all of them.  Results equal snk_greedy utterance by utterance.
int snk_greedy_batch(snk_handle h, const double *Q, const int64_t *row_offsets, int n_utts, int D,
                     const int64_t *start_states, double eps, int64_t *path_out, double *dist_out,
                     int64_t *nsteps_out)
{
    CHK(check_ready(h, true, true));
    CHK(no_batch_in_flight(h, "snk_greedy_batch"));
    HIPCHK(hipSetDevice(h->device));
    if (!h->have_glay) return fail("snk_greedy_batch: greedy layout not set (snk_set_greedy_layout)");
    if (!Q || !row_offsets || n_utts < 1 || !path_out || !nsteps_out) return fail("snk_greedy_batch: null/empty argument");
    if (D != h->Dt) return fail("query matrix has %d columns, database has %d", D, h->Dt);
    if (!(eps >= 0.0)) return fail("snk_greedy_batch: search_epsilon must be >= 0");
    const GreedyLayout &g = h->glay;
    const int64_t total = row_offsets[n_utts];
    std::vector<int64_t> nsteps((size_t)n_utts), out_off((size_t)n_utts + 1, 0);
    for (int u = 0; u < n_utts; ++u) {
        const int64_t T = row_offsets[u + 1] - row_offsets[u];
        if (T < 1) return fail("snk_greedy_batch: utterance %d has no rows", u);
        if (start_states && start_states[u] >= g.Nwin) return fail("snk_greedy_batch: start_state %lld out of range", (long long)start_states[u]);
        nsteps[(size_t)u] = T / g.me;             // py2 integer division: tail frames dropped
        nsteps_out[u] = nsteps[(size_t)u];
        out_off[(size_t)u + 1] = out_off[(size_t)u] + nsteps[(size_t)u];
    }
    const int64_t total_steps = out_off[(size_t)n_utts];
    CHK(upload_queries(h, Q, total, D));
    if (total_steps == 0) { HIPCHK(hipStreamSynchronize(h->stream)); collect_timers(h); return 0; }
    if (!h->gtiles_ready) {
        CHK(h->gtiles.ensure(greedy_tile_bytes(g, h->Dt)));
        launch_greedy_tiles(g, h->F_unw.as<float>(), h->Fp, h->Dt, h->JC_unw.as<float>(), h->Jp, h->gtiles.as<float>(), h->stream);
        HIPCHK(hipGetLastError());
        h->gtiles_ready = true;
    }
    const int ub = greedy_max_utts(g, h->Dt);
    const int nblk = greedy_blocks(g, h->Dt, h->n_cus, ub);
    CHK(h->gprev.ensure(2 * greedy_table_doubles(g, h->Dt, ub) * sizeof(double) + 512));
    CHK(h->gsync.ensure(greedy_counter_bytes()));
    CHK(h->gblkmin.ensure((size_t)ub * nblk * sizeof(double)));
    CHK(h->gblkarg.ensure((size_t)ub * nblk * sizeof(int64_t)));
    CHK(h->gpath.ensure((size_t)total_steps * sizeof(int64_t)));
    CHK(h->gdist.ensure((size_t)total_steps * sizeof(double)));
    // utterances of similar length share a scan (the scan runs for the longest of its group)
    std::vector<int> order((size_t)n_utts);
    for (int u = 0; u < n_utts; ++u) order[(size_t)u] = u;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return nsteps[(size_t)x] > nsteps[(size_t)y]; });
    const bool g32 = use_greedy32(h, n_utts);
    const int ub_run = g32 ? greedy32_max_utts(h->greedy_hoist && greedy_hoist_supported(g, h->Dt)) : ub;
    {
        StageTimer t(h, h->stream, TM_GREEDY_STEPS);
        for (int i = 0; i < n_utts; i += ub_run) {
            int nu = 0;
            int64_t q_off[6], ns[6], oo[6], st[6];
            for (; nu < ub_run && i + nu < n_utts; ++nu) {
                const int u = order[(size_t)(i + nu)];
                if (nsteps[(size_t)u] == 0) break;             // sorted: the rest have no steps either
                q_off[nu] = row_offsets[u]; ns[nu] = nsteps[(size_t)u]; oo[nu] = out_off[(size_t)u];
                st[nu] = start_states ? start_states[u] : -1;
            }
            if (nu == 0) break;
            if (g32) {
                bool undecided = false;
                CHK(greedy32_group(h, nu, q_off, ns, oo, st, eps >= 1e-3, dist_out != nullptr, &undecided));
                if (!undecided) continue;
                h->greedy_fallbacks += 1;
            }
            // exact scan, a launch per step (two utterances per scan at most)
            for (int j = 0; j < nu; j += ub) {
                const int n2 = (nu - j < ub) ? nu - j : ub;
                launch_greedy_batch(g, h->F_unw.as<float>(), h->Fp, h->Dt, h->wt.as<double>(), h->JC_unw.as<float>(), h->Jp, h->Dj,
                                    h->wj.as<double>(), h->gtiles.as<float>(), h->Qraw.as<double>(), n2, q_off + j, ns + j, oo + j, st + j,
                                    h->gprev.as<double>(), h->gblkmin.as<double>(), h->gblkarg.as<int64_t>(),
                                    greedy_blocks(g, h->Dt, h->n_cus, n2), h->n_cus, h->gsync.as<unsigned int>(),
                                    h->gpath.as<int64_t>(), h->gdist.as<double>(), h->stream);
            }
        }
    }
    HIPCHK(hipGetLastError());
    {
        StageTimer t(h, h->stream, TM_D2H);
        D2HPart parts[2] = {{path_out, h->gpath.p, (size_t)total_steps * sizeof(int64_t)},
                            {dist_out, h->gdist.p, dist_out ? (size_t)total_steps * sizeof(double) : 0}};
        CHK(staged_d2h(h, h->stream, parts, 2));
    }
    collect_timers(h);
    return 0;
}

int snk_path_scores(snk_handle h, const double *Q, const int64_t *path, int64_t L, int mode,
                    double *tsq_out, double *jsq_out)
{
    CHK(check_ready(h, true, true));
    HIPCHK(hipSetDevice(h->device));
    if (!Q || !path || !tsq_out || L < 1) return fail("snk_path_scores: null/empty argument");
    if (mode != 0 && mode != 1) return fail("snk_path_scores: mode must be 0 (viterbi) or 1 (greedy)");
    if (mode == 1 && !h->have_glay) return fail("snk_path_scores: greedy layout not set");
    GreedyLayout g = h->glay;
    int me = 1, nep = 1, jcols = h->Dj;
    if (mode == 1) { me = g.me; nep = (g.last_frame_as_target && me > 1) ? 2 : me; jcols = g.jdim; }
    const int64_t limit = (mode == 1) ? g.Nwin : h->N;
    for (int64_t l = 0; l < L; ++l)
        if (path[l] < 0 || path[l] >= limit) return fail("snk_path_scores: path[%lld]=%lld out of range", (long long)l, (long long)path[l]);
    const size_t qrows = (size_t)L * me;
    CHK(h->Qraw.ensure(qrows * h->Dt * sizeof(double)));
    CHK(h->gpath.ensure((size_t)L * sizeof(int64_t)));
    const size_t tbytes = (size_t)L * nep * h->Dt * sizeof(double);
    const size_t jbytes = (size_t)(L > 1 ? L - 1 : 1) * jcols * sizeof(double);
    CHK(h->d2tmp.ensure(tbytes + jbytes));
    CHK(h2d(h, h->Qraw.p, Q, qrows * h->Dt * sizeof(double), h->stream));
    if (!h->tsel.empty()) launch_mask_columns(h->Qraw.as<double>(), (int64_t)qrows, h->Dt, h->tmask.as<double>(), h->stream);
    CHK(h2d(h, h->gpath.p, path, (size_t)L * sizeof(int64_t), h->stream));
    double *tsq = h->d2tmp.as<double>();
    double *jsq = reinterpret_cast<double *>(reinterpret_cast<char *>(h->d2tmp.p) + tbytes);
    launch_path_scores(g, mode, h->F_unw.as<float>(), h->Fp, h->Dt, h->wt.as<double>(), h->JC_unw.as<float>(), h->Jp, h->Dj,
                       h->wj.as<double>(), h->Qraw.as<double>(), h->gpath.as<int64_t>(), L, tsq, jsq, jcols, h->stream);
    HIPCHK(hipGetLastError());
    {
        D2HPart parts[2] = {{tsq_out, tsq, tbytes}, {jsq_out, jsq, (jsq_out && L > 1) ? (size_t)(L - 1) * jcols * sizeof(double) : 0}};
        CHK(staged_d2h(h, h->stream, parts, 2));
    }
    return 0;
}
