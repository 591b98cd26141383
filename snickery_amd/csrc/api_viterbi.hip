// C ABI of libsnkhip.so, part 3: join costs, Viterbi (viterbi_search, synth_halfphone.py:1399-1436) and the batch pipeline
// (K-NN of a group on the main stream, the recursions of the group before on a side stream).
#include "snk_engine.h"

static int slot_ensure(snk_engine *h, UttSlot &s, int64_t T, int K)
{
    CHK(s.cand.ensure((size_t)T * K * sizeof(int64_t)));
    CHK(s.tdist.ensure((size_t)T * K * sizeof(double)));
    if (!use_sparse_viterbi(h, K, 1)) CHK(s.J.ensure((size_t)(T > 1 ? T - 1 : 1) * K * K * sizeof(double)));
    CHK(s.bp.ensure((size_t)T * K));
    CHK(s.path.ensure((size_t)T * sizeof(int64_t)));
    CHK(s.plen.ensure(sizeof(int64_t)));
    CHK(s.cost.ensure(sizeof(double)));
    return 0;
}

static int64_t join_units(snk_engine *h)
{
    // data_frames of unit_end_data = rows of join_contexts - 1 (synth_halfphone.py:3227)
    return h->Njc - 1;
}

// viterbi_mode 2 (default): the sparse path wherever it is supported -- batches (its first pass runs over the whole
// chip while the per-utterance passes hide beside the next group's K-NN) and, since pass 2 runs in chunks side by
// side and pass 4 on one wavefront, a single utterance too (T = 600, K = 100: 0.75 ms against 1.65 ms through the
// dense kernels).  1 forces it, 0 forces the dense exact path.  Same results.
bool use_sparse_viterbi(const snk_engine *h, int K, int n_utts)
{
    (void)n_utts;
    if (!join_lb_supported(h->Dj, K)) return false;
    if (h->viterbi_mode == 2 && h->vit_now_dense) return false;       // this voice's batches were judged faster through the dense kernels (snk_engine.h: vit)
    return h->viterbi_mode == 1 || h->viterbi_mode == 2;
}

static int sparse_ensure(snk_engine *h, UttSlot &s, int64_t rows, int K)
{
    CHK(s.Jlo.ensure((size_t)(rows > 1 ? rows - 1 : 1) * K * K * sizeof(float)));
    CHK(s.scale.ensure((size_t)rows * sizeof(float)));
    CHK(s.sets.ensure((size_t)rows * K * 16));
    CHK(s.cex.ensure((size_t)rows * K * join_record_bytes() + 4096));
    CHK(s.bp.ensure((size_t)rows * K));
    if (!h->vstats.p) {
        CHK(h->vstats.ensure((128 + 16 * 1024) * sizeof(unsigned long long)));    // + the stamps of a -DSNK_JF_TRACE build
        HIPCHK(hipMemset(h->vstats.p, 0, 128 * sizeof(unsigned long long)));
        HIPCHK(hipMemset(reinterpret_cast<char *>(h->vstats.p) + 5 * sizeof(unsigned long long), 0xff, 4));     // [5]: smallest margin of the bounds' tripwire (image of +inf and beyond)
    }
    return 0;
}

static int ensure_jw32(snk_engine *h, hipStream_t st)
{
    if (h->jw32_ready) return 0;
    // once per set of weights; waited for: the groups of a batch launch pass 1 on different streams
    const int Jq = join_lb2_pitch(h->Dj);
    CHK(h->JW32.ensure((size_t)h->Njc * Jq * sizeof(float)));
    CHK(h->jw_umax.ensure(64));
    launch_join_weight32(h->JC_unw.as<float>(), h->Jp, h->Njc, h->Dj, h->wj.as<double>(), h->JW32.as<float>(), Jq,
                         h->jw_umax.as<unsigned int>(), st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    h->jw32_ready = true;
    return 0;
}

static void join_bounds_launch(snk_engine *h, const int64_t *cand, int64_t rows, int K, float *Jlo, float *scale, hipStream_t st)
{
    if (h->join_lb_variant == 1)
        launch_join_lb2(h->JW32.as<float>(), h->Dj, h->jw_umax.as<unsigned int>(), join_units(h), cand, rows, K, Jlo, scale, st);
    else
        launch_join_lb(h->JC_unw.as<float>(), h->Jp, h->Dj, h->wj.as<double>(), join_units(h), cand, rows, K, Jlo, scale, st);
}

// Passes 1..4 of joinfast_kernels.hip over `rows` candidate rows holding n_utts utterances (off: n_utts + 1
// row offsets).  Pass 1 runs on `main` (the whole chip, in parallel over the rows); the three per-utterance
// passes on `side` behind `knn_done` when the two streams differ.
static int viterbi_sparse_rows(snk_engine *h, UttSlot &s, const int64_t *cand, const double *tdist, int64_t rows,
                               const int64_t *off, int n_utts, int first_utt, int K, hipStream_t main, hipStream_t side,
                               int64_t *path, int64_t *plen, double *cost, bool knn_done_recorded = false, hipEvent_t also_behind = nullptr)
{
    const float *JC = h->JC_unw.as<float>();
    const double *wj = h->wj.as<double>();
    const bool lb_side = h->join_bounds_stream == 1 && side != main;
    const float ceps = (float)join_lb_ceps(h->join_lb_variant, h->Dj, K);       // unit of the bounds' tripwire (joinfast_kernels.hip)
    if (h->join_lb_variant == 1) CHK(ensure_jw32(h, main));
    if (lb_side) {
        if (!knn_done_recorded) HIPCHK(hipEventRecord(s.knn_done, main));
        HIPCHK(hipStreamWaitEvent(side, s.knn_done, 0));
        if (also_behind) HIPCHK(hipStreamWaitEvent(side, also_behind, 0));          // (join_bounds_delay: where the next group's K-NN stands)
    }
    {
        StageTimer t(h, lb_side ? side : main, TM_JOIN_LB);
        join_bounds_launch(h, cand, rows, K, s.Jlo.as<float>(), s.scale.as<float>(), lb_side ? side : main);
        if (h->join_lb_test_scale != 1.0 && rows > 1)
            launch_scale_f32(s.Jlo.as<float>(), (rows - 1) * (int64_t)K * K, (float)h->join_lb_test_scale, lb_side ? side : main);
    }
    if (side != main && !lb_side) {
        HIPCHK(hipEventRecord(s.knn_done, main));
        HIPCHK(hipStreamWaitEvent(side, s.knn_done, 0));
    }
    {
        StageTimer t(h, side, TM_DP_LB);
        launch_viterbi_lb(cand, tdist, s.Jlo.as<float>(), s.scale.as<float>(), off, n_utts, K, join_units(h),
                          (float)h->join_beta, s.sets.p, side,
                          n_utts <= h->lb_chunk_max_utts ? (n_utts <= 4 && h->lb_chunk > 32 ? 32 : h->lb_chunk) : 0, h->lb_warm_eff > h->lb_warm ? h->lb_warm_eff : h->lb_warm,
                          h->viterbi_weights == 1 ? (float)h->fst32_slack : 0.f);
    }
    {
        StageTimer t(h, side, TM_JOIN_SPARSE);
        launch_join_exact_sparse(JC, h->Jp, h->Dj, wj, join_units(h), cand, tdist, rows, K, s.sets.p, s.cex.p, side,
                                 s.Jlo.as<float>(), s.scale.as<float>(), ceps, h->vstats.as<unsigned long long>());
    }
    {
        StageTimer t(h, side, TM_DP_SPARSE);
        launch_viterbi_sparse(cand, s.cex.p, s.Jlo.as<float>(), JC, h->Jp, h->Dj, wj, off, n_utts, first_utt, K,
                              join_units(h), s.bp.as<unsigned char>(), path, plen, cost,
                              h->vstats.as<unsigned long long>(), side, s.scale.as<float>(), ceps, h->viterbi_weights == 1);
    }
    return 0;
}

static int viterbi_device(snk_engine *h, UttSlot &s, int64_t T, int K, hipStream_t st)
{
    if (K > 208) return fail("viterbi: n_candidates=%d > 208 not supported", K);
    if (use_sparse_viterbi(h, K, 1)) {
        CHK(sparse_ensure(h, s, T, K));
        const int64_t off[2] = {0, T};
        return viterbi_sparse_rows(h, s, s.cand.as<int64_t>(), s.tdist.as<double>(), T, off, 1, 0, K, st, st,
                                   s.path.as<int64_t>(), s.plen.as<int64_t>(), s.cost.as<double>());
    }
    {
        StageTimer t(h, st, TM_JOIN);
        launch_join_costs(h->JCw.as<double>(), h->Djpad, h->Dj, join_units(h), s.cand.as<int64_t>(), T, K,
                          s.J.as<double>(), st);
    }
    {
        StageTimer t(h, st, TM_VITERBI_DP);
        launch_viterbi_dp(s.cand.as<int64_t>(), s.tdist.as<double>(), s.J.as<double>(), T, K, join_units(h),
                          s.bp.as<unsigned char>(), s.path.as<int64_t>(), s.plen.as<int64_t>(),
                          s.cost.as<double>(), st, h->viterbi_weights == 1);
    }
    return 0;
}

int snk_join_costs(snk_handle h, const int64_t *cand, int64_t T, int K, double *J_out)
{
    CHK(check_ready(h, false, true));
    CHK(no_batch_in_flight(h, "snk_join_costs"));
    HIPCHK(hipSetDevice(h->device));
    if (!cand || !J_out) return fail("snk_join_costs: null argument");
    if (T < 2) return fail("snk_join_costs: need at least 2 columns");
    if (K < 1 || K > 208) return fail("snk_join_costs: K=%d outside 1..208", K);
    UttSlot &s = h->slot[0];
    CHK(slot_ensure(h, s, T, K));
    CHK(s.J.ensure((size_t)(T - 1) * K * K * sizeof(double)));
    CHK(h2d(h, s.cand.p, cand, (size_t)T * K * sizeof(int64_t), h->stream));
    {
        StageTimer t(h, h->stream, TM_JOIN);
        launch_join_costs(h->JCw.as<double>(), h->Djpad, h->Dj, join_units(h), s.cand.as<int64_t>(), T, K,
                          s.J.as<double>(), h->stream);
    }
    HIPCHK(hipGetLastError());
    CHK(d2h_sync(h, J_out, s.J.p, (size_t)(T - 1) * K * K * sizeof(double), h->stream));
    collect_timers(h);
    return 0;
}

int snk_join_bounds(snk_handle h, const int64_t *cand, int64_t T, int K, float *lo_out, float *scale_out)
{
    CHK(check_ready(h, false, true));
    CHK(no_batch_in_flight(h, "snk_join_bounds"));
    HIPCHK(hipSetDevice(h->device));
    if (!cand || !lo_out || !scale_out) return fail("snk_join_bounds: null argument");
    if (T < 2) return fail("snk_join_bounds: need at least 2 columns");
    if (!join_lb_supported(h->Dj, K)) return fail("snk_join_bounds: no bounds variant for %d join columns, K=%d", h->Dj, K);
    UttSlot &s = h->slot[0];
    CHK(slot_ensure(h, s, T, K));
    CHK(sparse_ensure(h, s, T, K));
    CHK(h2d(h, s.cand.p, cand, (size_t)T * K * sizeof(int64_t), h->stream));
    if (h->join_lb_variant == 1) CHK(ensure_jw32(h, h->stream));
    join_bounds_launch(h, s.cand.as<int64_t>(), T, K, s.Jlo.as<float>(), s.scale.as<float>(), h->stream);
    HIPCHK(hipGetLastError());
    D2HPart parts[2] = {{lo_out, s.Jlo.p, (size_t)(T - 1) * K * K * sizeof(float)}, {scale_out, s.scale.p, (size_t)(T - 1) * sizeof(float)}};
    CHK(staged_d2h(h, h->stream, parts, 2));
    collect_timers(h);
    return 0;
}

int snk_viterbi(snk_handle h, const int64_t *cand, const double *tdist, int64_t T, int K,
                int64_t *path_out, int64_t *path_len_out, double *cost_out)
{
    CHK(check_ready(h, false, true));
    CHK(no_batch_in_flight(h, "snk_viterbi"));
    HIPCHK(hipSetDevice(h->device));
    if (!cand || !tdist || !path_out || !path_len_out) return fail("snk_viterbi: null argument");
    if (T < 1 || K < 1) return fail("snk_viterbi: empty trellis");
    UttSlot &s = h->slot[0];
    CHK(slot_ensure(h, s, T, K));
    CHK(h2d(h, s.cand.p, cand, (size_t)T * K * sizeof(int64_t), h->stream));
    CHK(h2d(h, s.tdist.p, tdist, (size_t)T * K * sizeof(double), h->stream));
    CHK(viterbi_device(h, s, T, K, h->stream));
    HIPCHK(hipGetLastError());
    double cost = 0;
    {
        D2HPart parts[3] = {{path_len_out, s.plen.p, sizeof(int64_t)}, {&cost, s.cost.p, sizeof(double)},
                            {path_out, s.path.p, (size_t)T * sizeof(int64_t)}};
        CHK(staged_d2h(h, h->stream, parts, 3));
    }
    if (cost_out) *cost_out = cost;
    collect_timers(h);
    return 0;
}

int snk_knn_viterbi(snk_handle h, const double *Q, int64_t T, int D, int K, int64_t *cand_out,
                    double *dist_out, int64_t *path_out, int64_t *path_len_out, double *cost_out)
{
    CHK(check_ready(h, true, true));
    CHK(no_batch_in_flight(h, "snk_knn_viterbi"));
    HIPCHK(hipSetDevice(h->device));
    if (!path_out || !path_len_out) return fail("snk_knn_viterbi: null output");
    if (h->Njc != h->N + 1) return fail("snk_knn_viterbi: join_contexts rows (%lld) != N+1", (long long)h->Njc);
    CHK(upload_queries(h, Q, T, D));
    UttSlot &s = h->slot[0];
    CHK(slot_ensure(h, s, T, K));
    CHK(knn_device(h, h->Qraw.as<double>(), T, K, nullptr, s.cand.as<int64_t>(), s.tdist.as<double>(), nullptr));
    CHK(viterbi_device(h, s, T, K, h->stream));
    HIPCHK(hipGetLastError());
    double cost = 0;
    {
        StageTimer t(h, h->stream, TM_D2H);
        D2HPart parts[5] = {{cand_out, s.cand.p, cand_out ? (size_t)T * K * sizeof(int64_t) : 0},
                            {dist_out, s.tdist.p, dist_out ? (size_t)T * K * sizeof(double) : 0},
                            {path_len_out, s.plen.p, sizeof(int64_t)},
                            {&cost, s.cost.p, sizeof(double)},
                            {path_out, s.path.p, (size_t)T * sizeof(int64_t)}};
        CHK(staged_d2h(h, h->stream, parts, 5));
    }
    if (cost_out) *cost_out = cost;
    collect_timers(h);
    return 0;
}

// Groups consecutive utterances into K-NN calls of about h->batch_rows rows: the search is per row,
// so one sweep over the database serves every utterance of the group and the per-call stages
// (sample minima, thresholds, bucket, re-rank) amortise.  first[g] .. first[g+1] are the utterances
// of group g.
std::vector<int> group_utterances(const snk_engine *h, const int64_t *row_offsets, int n_utts, bool knn_beside, int K)
{
    std::vector<int> first(1, 0);
    // as few groups as batch_rows allows, of equal size and an even number of them: two groups of 16 utterances take a
    // B* step 13 % less time than 13 + 13 + 6 (11.9 against 13.7 ms; three of 11 / 11 / 10: 13.2, four of 8: 12.0,
    // one of 32: 18.2 -- nothing of its own step to run beside)
    const int64_t total = row_offsets[n_utts] - row_offsets[0];
    int64_t target = h->batch_rows;
    if (h->batch_rows > 0) {
        int64_t n_groups = (total + h->batch_rows - 1) / h->batch_rows;
        if (n_groups > 1 && (n_groups & 1)) ++n_groups;       // groups alternate between two workspaces and side streams
        // a batch that fits one group, but is long: two, so that the first one's Viterbi side runs beside the second one's K-NN
        // (B5, 64 x 120 rows: 1.86 -> 1.91 M frames/s).  Only where there IS a K-NN to run beside (the K-NN batch entry points:
        // knn_beside); callers that bring their candidates (snk_viterbi_batch, the merged lists of a sharded step) keep one group,
        // one launch of every pass.  Option split_one_group 0 keeps one group everywhere.
        if (knn_beside && h->split_one_group && n_groups == 1 && total >= 6144 && n_utts >= 2) n_groups = 2;
        // option wide_one_group (default 0): K > 128, where the Viterbi side of a group is what a step waits for (pass 1 on one
        // accumulator set, pass 4 a chain of T steps that refines a tenth of them: 8 ms per group of 16 utterances at B4, and a chain
        // takes as long for 32 utterances as for 16) -- the whole batch as ONE group where it fits a K-NN call, consecutive batches
        // taking the two side streams in turn (BatchSlot::gbase), so that two batches' chains run side by side.  Measured at B4:
        // slower (9.6 -> 11.2 ms per step, profiles/r06k_b4.log: the single group's passes 1 and 3 are twice as long and nothing
        // of its own batch runs beside them); with the quadrant form of pass 1 it breaks even (9.3 ms).  Kept as an option.
        if (knn_beside && h->wide_one_group && K > 128 && total <= SNK_KNN_MAX_ROWS) n_groups = 1;
        target = (total + n_groups - 1) / n_groups;
    }
    int64_t rows = 0;
    for (int u = 0; u < n_utts; ++u) {
        const int64_t T = row_offsets[u + 1] - row_offsets[u];
        // close the group when adding this utterance would overshoot the even share by more than it undershoots
        if (u > first.back() && (h->batch_rows <= 0 || rows + T > h->batch_rows || rows + T - target > target - rows)) {
            first.push_back(u);
            rows = 0;
        }
        rows += T;
    }
    first.push_back(n_utts);
    return first;
}

// Join costs and recursions of the utterances [u0, u1) whose candidate rows are resident in
// cand_all / tdist_all (batch row numbering).  The join costs of the whole group are ONE launch on
// the main stream (its rows form one long sequence; the slab between two utterances is never read),
// the recursions ONE launch (a workgroup per utterance) on the side stream of the group's parity,
// where they land on the compute units the persistent K-NN sweep of the next group leaves free.
int viterbi_group(snk_engine *h, int g, const int64_t *row_offsets, int u0, int u1, int K,
                         const int64_t *cand_all, const double *tdist_all, bool side_stream,
                         int64_t *res_path, int64_t *res_plen, double *res_cost, int n_batch_utts,
                         bool knn_done_recorded, hipEvent_t also_behind)
{
    if (!res_path) { res_path = h->res_path.as<int64_t>(); res_plen = h->res_plen.as<int64_t>(); res_cost = h->res_cost.as<double>(); }
    const int64_t r0 = row_offsets[u0], rows = row_offsets[u1] - r0;
    UttSlot &s = h->slot[g & 1];
    hipStream_t dps = side_stream ? h->dp_stream[g & 1] : h->stream;
    // workspace reuse: the join costs of this group overwrite what the last recursion queued on this
    // workspace reads (an earlier group of this batch, or the tail of the batch submitted before)
    const bool sparse = use_sparse_viterbi(h, K, n_batch_utts);
    // (the sparse path with its bounds on the side stream touches the workspace on that stream only: in order)
    if (s.vit_recorded && !(sparse && side_stream && h->join_bounds_stream == 1)) HIPCHK(hipStreamWaitEvent(h->stream, s.vit_done, 0));
    if (sparse) {
        CHK(sparse_ensure(h, s, rows, K));
        std::vector<int64_t> off((size_t)(u1 - u0) + 1);
        for (int u = u0; u <= u1; ++u) off[(size_t)(u - u0)] = row_offsets[u] - r0;
        CHK(viterbi_sparse_rows(h, s, cand_all + r0 * K, tdist_all + r0 * K, rows, off.data(), u1 - u0, u0, K, h->stream, dps,
                                res_path + r0, res_plen, res_cost, knn_done_recorded, also_behind));
        if (side_stream) { HIPCHK(hipEventRecord(s.vit_done, dps)); s.vit_recorded = true; }
        else s.vit_recorded = false;
        return 0;
    }
    CHK(s.J.ensure((size_t)(rows > 1 ? rows - 1 : 1) * K * K * sizeof(double)));
    CHK(s.bp.ensure((size_t)rows * K));
    {
        StageTimer t(h, h->stream, TM_JOIN);
        launch_join_costs(h->JCw.as<double>(), h->Djpad, h->Dj, join_units(h), cand_all + r0 * K, rows, K,
                          s.J.as<double>(), h->stream);
    }
    if (side_stream) {
        HIPCHK(hipEventRecord(s.knn_done, h->stream));
        HIPCHK(hipStreamWaitEvent(dps, s.knn_done, 0));
    }
    std::vector<int64_t> off((size_t)(u1 - u0) + 1);
    for (int u = u0; u <= u1; ++u) off[(size_t)(u - u0)] = row_offsets[u] - r0;
    {
        StageTimer t(h, dps, TM_VITERBI_DP);
        launch_viterbi_dp_batch(cand_all + r0 * K, tdist_all + r0 * K, s.J.as<double>(), off.data(), u1 - u0, u0, K,
                                join_units(h), s.bp.as<unsigned char>(), res_path + r0, res_plen, res_cost, dps,
                                h->viterbi_weights == 1);
    }
    if (side_stream) { HIPCHK(hipEventRecord(s.vit_done, dps)); s.vit_recorded = true; }
    else s.vit_recorded = false;             // ran on the main stream: ordered with everything that follows
    return 0;
}

// results of a batch -> pinned memory, behind its K-NN status words (main stream) and its last recursions
static int batch_queue_results(snk_engine *h, BatchSlot &b)
{
    const int64_t total = b.total;
    const int n_utts = b.n_utts;
    const size_t sz_path = ((size_t)total * sizeof(int64_t) + 63) & ~(size_t)63;
    const size_t sz_u = ((size_t)n_utts * 8 + 63) & ~(size_t)63, sz_st = ((size_t)3 * b.n_groups * sizeof(int) + 63) & ~(size_t)63;
    HIPCHK(hipStreamWaitEvent(h->copy_stream, b.knn_end, 0));
    for (int i = 0; i < 2; ++i)
        if (h->slot[i].vit_recorded) HIPCHK(hipStreamWaitEvent(h->copy_stream, h->slot[i].vit_done, 0));
    {
        StageTimer t(h, h->copy_stream, TM_D2H);
        char *st = (char *)b.stage.p;
        if (h->results_by_kernel) {
            // (a kernel's stores, not DMA copies: viterbi_kernels.hip results_to_host_kernel says why)
            void *dst[5] = {st, st + sz_path, st + sz_path + sz_u, st + sz_path + 2 * sz_u, st + sz_path + 2 * sz_u + sz_st};
            const void *src[5] = {b.path.p, b.plen.p, b.cost.p, b.status.p, h->vstats.p};
            const size_t nb[5] = {(size_t)total * sizeof(int64_t), (size_t)n_utts * sizeof(int64_t), (size_t)n_utts * sizeof(double),
                                  (size_t)3 * b.n_groups * sizeof(int), 4 * sizeof(unsigned long long)};
            launch_results_to_host(dst, src, nb, 5, h->copy_stream);
        } else {
        HIPCHK(hipMemcpyAsync(st, b.path.p, (size_t)total * sizeof(int64_t), hipMemcpyDeviceToHost, h->copy_stream));
        HIPCHK(hipMemcpyAsync(st + sz_path, b.plen.p, (size_t)n_utts * sizeof(int64_t), hipMemcpyDeviceToHost, h->copy_stream));
        HIPCHK(hipMemcpyAsync(st + sz_path + sz_u, b.cost.p, (size_t)n_utts * sizeof(double), hipMemcpyDeviceToHost, h->copy_stream));
        HIPCHK(hipMemcpyAsync(st + sz_path + 2 * sz_u, b.status.p, (size_t)3 * b.n_groups * sizeof(int), hipMemcpyDeviceToHost, h->copy_stream));
        HIPCHK(hipMemcpyAsync(st + sz_path + 2 * sz_u + sz_st, h->vstats.p, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->copy_stream));
        }
    }
    HIPCHK(hipEventRecord(h->vit_t1[b.seq & 3], h->copy_stream));
    HIPCHK(hipEventRecord(b.done, h->copy_stream));
    HIPCHK(hipGetLastError());
    return 0;
}

// join_bounds_delay: the Viterbi side of a batch's last group, left unqueued by its submit (BatchSlot::tail_pending), and the
// copy of the batch's results behind it.  also_behind: the point inside the NEXT batch's first K-NN call it starts behind
// (nullptr: a collect came first -- it starts where it stands).
static int batch_flush_tail(snk_engine *h, BatchSlot &b, hipEvent_t also_behind)
{
    if (!b.busy || !b.tail_pending) return 0;
    b.tail_pending = false;
    const int g = b.n_groups - 1;
    struct Restore { snk_engine *e; bool v; ~Restore() { e->vit_now_dense = v; } } restore{h, h->vit_now_dense};
    h->vit_now_dense = false;                  // (a batch with a pending tail took the sparse path)
    CHK(viterbi_group(h, g + b.gbase, b.offs.data(), b.first[g], b.first[g + 1], b.K, b.cand.as<int64_t>(), b.dist.as<double>(), true,
                      b.path.as<int64_t>(), b.plen.as<int64_t>(), b.cost.as<double>(), b.n_utts, true, also_behind));
    return batch_queue_results(h, b);
}

// Batch pipeline.  The main stream runs the K-NN of a group of utterances and their join costs; the
// T-step recursions run on a side stream, overlapping the K-NN of the next group.  All results stay
// on the device until the end of the batch; a copy stream moves them to pinned host memory behind
// the last recursions.  submit() only queues work (two batches may be in flight, each with its own
// query / candidate / result buffers), collect() waits for one batch: a caller that submits batch
// i+1 before collecting batch i hides the tail of batch i (its last group's recursions, the copy
// and the host-side hand-over) behind the K-NN of batch i+1.
int snk_knn_viterbi_batch_submit(snk_handle h, const double *Q, const int64_t *row_offsets, int n_utts, int D,
                                 int K, int *ticket_out)
{
    CHK(check_ready(h, true, true));
    if (h->sticket[0].busy || h->sticket[1].busy)
        return fail("snk_knn_viterbi_batch_submit: a submitted sharded step is still in flight (snk_sharded_knn_viterbi_batch_collect it first)");
    HIPCHK(hipSetDevice(h->device));
    if (!row_offsets || n_utts < 1 || !ticket_out)
        return fail("snk_knn_viterbi_batch_submit: null/empty argument");
    if (D != h->Dt) return fail("query matrix has %d columns, database has %d", D, h->Dt);
    if (h->Njc != h->N + 1) return fail("snk_knn_viterbi_batch: join_contexts rows != N+1");
    if (K > 208) return fail("viterbi: n_candidates=%d > 208 not supported", K);
    if (row_offsets[0] != 0) return fail("snk_knn_viterbi_batch: row_offsets[0] must be 0 (got %lld)", (long long)row_offsets[0]);
    const int64_t total = row_offsets[n_utts];
    for (int u = 0; u < n_utts; ++u)
        if (row_offsets[u + 1] - row_offsets[u] < 1) return fail("snk_knn_viterbi_batch: utterance %d has no rows", u);
    int slot = h->bnext;
    for (int k = 0; k < SNK_BATCH_SLOTS && h->bslot[slot].busy; ++k) slot = (slot + 1) % SNK_BATCH_SLOTS;
    BatchSlot &b = h->bslot[slot];
    if (b.busy) return fail("snk_knn_viterbi_batch_submit: %d batches are in flight already (collect one first)", SNK_BATCH_SLOTS);
    b.first = group_utterances(h, row_offsets, n_utts, true, K);
    b.n_groups = (int)b.first.size() - 1;
    b.n_utts = n_utts; b.K = K; b.D = D; b.total = total;
    b.offs.assign(row_offsets, row_offsets + n_utts + 1);
    { void *before = b.Qall.p; CHK(b.Qall.ensure((size_t)total * D * sizeof(double))); if (b.Qall.p != before) b.q_rows = -1; }
    CHK(b.cand.ensure((size_t)total * K * sizeof(int64_t)));
    CHK(b.dist.ensure((size_t)total * K * sizeof(double)));
    CHK(b.path.ensure((size_t)total * sizeof(int64_t)));
    CHK(b.plen.ensure((size_t)n_utts * sizeof(int64_t)));
    CHK(b.cost.ensure((size_t)n_utts * sizeof(double)));
    CHK(b.status.ensure((size_t)3 * b.n_groups * sizeof(int)));          // per group: K-NN status word | tile pairs the first pass listed | pairs its probe counted
    const size_t sz_path = ((size_t)total * sizeof(int64_t) + 63) & ~(size_t)63;
    const size_t sz_u = ((size_t)n_utts * 8 + 63) & ~(size_t)63, sz_st = ((size_t)3 * b.n_groups * sizeof(int) + 63) & ~(size_t)63;
    CHK(b.stage.ensure(sz_path + 2 * sz_u + sz_st + 64));                 // + the Viterbi statistics words
    // the Viterbi latch (snk_engine.h: vit): which exact path this batch takes, and the events its period is read from
    const bool vit_auto = h->viterbi_mode == 2 && h->viterbi_latch && join_lb_supported(h->Dj, K) && n_utts >= 2;
    b.vit_trial = vit_auto && h->vit.trial_left > 0;
    b.vit_dense = vit_auto && (b.vit_trial ? h->vit.trial_mode : h->vit.mode) == 1;
    b.vit_judged = vit_auto;
    b.seq = h->vit_seq++;
    b.gbase = (b.n_groups == 1) ? (int)(b.seq & 1) : 0;       // one-group batches alternate between the two side streams / Viterbi workspaces
    if (!h->vit_t0[0])
        for (int i = 0; i < 4; ++i) { HIPCHK(hipEventCreate(&h->vit_t0[i])); HIPCHK(hipEventCreate(&h->vit_t1[i])); }
    HIPCHK(hipEventRecord(h->vit_t0[b.seq & 3], h->stream));
    struct DenseGuard { snk_engine *e; ~DenseGuard() { e->vit_now_dense = false; } } dense_guard{h};
    h->vit_now_dense = b.vit_dense;
    if (b.vit_dense || !h->vstats.p) {
        // (the statistics words are copied below whichever path runs)
        if (!h->vstats.p) {
            CHK(h->vstats.ensure((128 + 16 * 1024) * sizeof(unsigned long long)));
            HIPCHK(hipMemset(h->vstats.p, 0, 128 * sizeof(unsigned long long)));
            HIPCHK(hipMemset(reinterpret_cast<char *>(h->vstats.p) + 5 * sizeof(unsigned long long), 0xff, 4));
        }
    }
    b.probe_kind.assign((size_t)b.n_groups, 0);
    b.probe_limit.assign((size_t)b.n_groups, 0.0);
    if (Q) {
        // option upload_stream 1: the rows travel on a stream of their own -- this workspace is idle (its last batch was collected), so
        // the copy needs to wait for nothing and runs on a DMA engine beside the batch before this one; the main stream only waits
        // for its event.  Default 0: at the head of the main stream, where the K-NN stream stands still for the 0.3 ms the 9.4 MB
        // of a B* step take to cross PCIe (why: snk_engine.h upload_stream).
        hipStream_t us = h->stream;
        if (h->upload_stream) {
            if (!h->up_stream) HIPCHK(hipStreamCreateWithFlags(&h->up_stream, hipStreamNonBlocking));
            if (!b.q_up) HIPCHK(hipEventCreateWithFlags(&b.q_up, hipEventDisableTiming));
            us = h->up_stream;
        }
        {
            StageTimer t(h, us, TM_H2D);
            CHK(h2d_via(b.qstage, b.Qall.p, Q, (size_t)total * D * sizeof(double), us));
        }
        if (us != h->stream) {
            HIPCHK(hipEventRecord(b.q_up, us));
            HIPCHK(hipStreamWaitEvent(h->stream, b.q_up, 0));
        }
        if (!h->tsel.empty()) launch_mask_columns(b.Qall.as<double>(), total, D, h->tmask.as<double>(), h->stream);
        b.q_rows = total; b.q_D = D;
        b.q_offs.assign(row_offsets, row_offsets + n_utts + 1);
    } else if (b.q_rows != total || b.q_D != D || b.q_offs.size() != (size_t)n_utts + 1 ||
               !std::equal(b.q_offs.begin(), b.q_offs.end(), row_offsets)) {
        return fail("snk_knn_viterbi_batch_submit: no query matrix given and this workspace holds no rows of that shape "
                    "(the first submit on each of the three workspaces must carry Q)");
    }
    bool tail = false;
    const bool pipelined = any_batch_busy(h);
    BatchSlot *const prev = (h->blast >= 0 && h->blast != slot && h->bslot[h->blast].busy) ? &h->bslot[h->blast] : nullptr;      // the batch submitted before this one
    // Is the host keeping up?  The batch before this one is still in flight; if everything it queued on the main stream has
    // already run, the K-NN stream stood idle while the host got here (it could not submit earlier: it was waiting for the
    // batch before that one) -- and a tail left to THIS submit would have stood idle with it.  tail_defer 2 (default): a voice
    // whose submits find the stream idle more often than not queues its tails at once instead (they then run in the gap);
    // 1: always leave them to the next submit (round 5), 0: never.
    if (prev && prev->knn_end) {
        const bool idle = hipEventQuery(prev->knn_end) == hipSuccess;
        (void)hipGetLastError();
        h->submits_seen += 1; h->submits_starved += idle ? 1 : 0;
        h->starved_ema = 0.9 * h->starved_ema + (idle ? 0.1 : 0.0);
    }
    const bool defer_tail = h->tail_defer == 1 || (h->tail_defer == 2 && h->starved_ema < 0.5);
    for (int g = 0; g < b.n_groups; ++g) {
        const int64_t r0 = row_offsets[b.first[g]], rows = row_offsets[b.first[g + 1]] - r0;
        CHK(knn_device(h, b.Qall.as<double>() + r0 * D, rows, K, nullptr, b.cand.as<int64_t>() + r0 * K,
                       b.dist.as<double>() + r0 * K, nullptr, b.status.as<int>() + g, nullptr, nullptr, false, false,
                       reinterpret_cast<unsigned int *>(b.status.as<int>() + b.n_groups + g),
                       reinterpret_cast<unsigned int *>(b.status.as<int>() + 2 * b.n_groups + g)));
        b.ball_limit = h->ball_pass_ran ? h->ball_limit : -1.0;
        b.coarse_limit = h->coarse_pass_ran ? h->coarse_limit : -1.0;
        b.probe_kind[(size_t)g] = h->probe_ran; b.probe_limit[(size_t)g] = h->probe_limit;
        b.operand_gen = h->operand_gen;
        // join_bounds_delay: the Viterbi side of group g - 1 is queued only now, behind a point inside THIS group's K-NN; its
        // candidates' event was recorded when they were queued (sparse path with pass 1 on the side stream only: nothing of it
        // touches the main stream)
        // Where it pays (measured, DESIGN.md 4.3: B*, B2 and the reordered voices gain 3-7 %; B5 loses 15 %, the AR(1) voice 6 %,
        // K = 200 1 %): batches of several long groups (a batch of one group is all tail: every Viterbi side would wait for the
        // next submit, and a short step leaves that tail no slack before the host must submit again), not behind a one-pass
        // sweep of the whole database (api_knn.hip: a voice on the COARSE sweep starts it behind that sweep), K <= 128 (beyond it
        // the Viterbi side is what a step waits for).  join_bounds_delay 3 / 4: 1 / 2 whatever the shape, 5: the coarse point only.
        const int64_t rows_per_group = total / (b.n_groups > 0 ? b.n_groups : 1);
        const bool fits = h->join_bounds_delay == 3 || h->join_bounds_delay == 4 || (h->join_bounds_delay == 5 && h->filter_coarse) ||
                          (b.n_groups >= 2 && rows_per_group >= 4096 && K <= 128 && !h->filter_onepass);
        const bool delay = h->join_bounds_delay > 0 && fits && h->join_bounds_stream == 1 && !b.vit_dense && use_sparse_viterbi(h, K, n_utts);
        if (g == 0 && prev) CHK(batch_flush_tail(h, *prev, h->knn_mid_recorded ? h->knn_mid : nullptr));      // the batch before this one
        if (delay) {
            if (g > 0)
                CHK(viterbi_group(h, g - 1 + b.gbase, row_offsets, b.first[g - 1], b.first[g], K, b.cand.as<int64_t>(), b.dist.as<double>(), true,
                                  b.path.as<int64_t>(), b.plen.as<int64_t>(), b.cost.as<double>(), n_utts, true,
                                  h->knn_mid_recorded ? h->knn_mid : nullptr));
            HIPCHK(hipEventRecord(h->slot[(g + b.gbase) & 1].knn_done, h->stream));
            if (g == b.n_groups - 1) {
                // The last group: left to the next submit (or to this batch's collect) where the caller keeps two batches in
                // flight -- the other workspace is busy right now, so the next thing it does is very likely another submit.  A
                // caller with one batch at a time (host work between submit and collect) gets it queued here, as before.
                if (pipelined && defer_tail) tail = true;
                else CHK(viterbi_group(h, g + b.gbase, row_offsets, b.first[g], b.first[g + 1], K, b.cand.as<int64_t>(), b.dist.as<double>(), true,
                                       b.path.as<int64_t>(), b.plen.as<int64_t>(), b.cost.as<double>(), n_utts, true, nullptr));
            }
            continue;
        }
        CHK(viterbi_group(h, g + b.gbase, row_offsets, b.first[g], b.first[g + 1], K, b.cand.as<int64_t>(), b.dist.as<double>(), true,
                          b.path.as<int64_t>(), b.plen.as<int64_t>(), b.cost.as<double>(), n_utts));
    }
    if (!b.knn_end) HIPCHK(hipEventCreateWithFlags(&b.knn_end, hipEventDisableTiming));
    HIPCHK(hipEventRecord(b.knn_end, h->stream));
    b.busy = true;                                 // (batch_flush_tail looks at it)
    b.tail_pending = tail;
    if (!tail) CHK(batch_queue_results(h, b));
    HIPCHK(hipGetLastError());
    b.busy = true;
    h->bnext = (slot + 1) % SNK_BATCH_SLOTS;
    h->blast = slot;
    *ticket_out = slot;
    return 0;
}

int snk_knn_viterbi_batch_collect(snk_handle h, int ticket, int64_t *path_out, int64_t *path_len_out, double *cost_out)
{
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    if (ticket < 0 || ticket >= SNK_BATCH_SLOTS || !h->bslot[ticket].busy) return fail("snk_knn_viterbi_batch_collect: no batch behind ticket %d", ticket);
    if (!path_out || !path_len_out || !cost_out) return fail("snk_knn_viterbi_batch_collect: null output");
    BatchSlot &b = h->bslot[ticket];
    const size_t sz_path = ((size_t)b.total * sizeof(int64_t) + 63) & ~(size_t)63;
    const size_t sz_u = ((size_t)b.n_utts * 8 + 63) & ~(size_t)63;
    CHK(batch_flush_tail(h, b, nullptr));           // (no batch was submitted behind this one: its last group starts where it stands)
    HIPCHK(hipEventSynchronize(b.done));          // this batch only: the one submitted after it may still run
    HIPCHK(hipGetLastError());
    b.busy = false;
    char *st = (char *)b.stage.p;
    // deferred K-NN status words: redo the (rare) group whose sampled thresholds overflowed a list
    const int *status = reinterpret_cast<const int *>(st + sz_path + 2 * sz_u);
    for (int g = 0; g < b.n_groups && b.operand_gen == h->operand_gen; ++g)       // (a batch that ran on operands since rebuilt says nothing about the new ones)
        judge_filter(h, b.ball_limit >= 0.0, b.ball_limit, b.coarse_limit >= 0.0, b.coarse_limit, (unsigned int)status[b.n_groups + g],
                     (unsigned int)status[2 * b.n_groups + g] != 0xffffffffu ? b.probe_kind[(size_t)g] : 0, b.probe_limit[(size_t)g],
                     (unsigned int)status[2 * b.n_groups + g]);
    // ---- the Viterbi latch (snk_engine.h: vit): this batch's period in stream time, per row ----
    if (b.vit_judged && h->viterbi_mode == 2 && h->viterbi_latch) {
        const size_t sz_st = ((size_t)3 * b.n_groups * sizeof(int) + 63) & ~(size_t)63;
        const unsigned long long *vs = reinterpret_cast<const unsigned long long *>(st + sz_path + 2 * sz_u + sz_st);
        float lat = 0.f, per = 0.f;
        double ms = -1.0;
        if (hipEventElapsedTime(&lat, h->vit_t0[b.seq & 3], h->vit_t1[b.seq & 3]) == hipSuccess) ms = lat;
        if (b.seq > 0 && h->vit_last_collected == b.seq - 1 &&
            hipEventElapsedTime(&per, h->vit_t1[(b.seq - 1) & 3], h->vit_t1[b.seq & 3]) == hipSuccess && per > 0.f && (ms < 0.0 || per < ms)) ms = per;
        (void)hipGetLastError();
        h->vit_last_collected = b.seq;
        const double cells = (double)(vs[0] - h->vit_cells_prev);
        h->vit_cells_prev = vs[0];
        snk_engine::VitLatch &v = h->vit;
        if (ms > 0.0 && b.total > 0) {
            const double row_ms = ms / (double)b.total;
            const int m = b.vit_dense ? 1 : 0;
            v.batches += 1;
            if (b.vit_trial && m == v.trial_mode) {
                // a batch of a trial: the first one overlaps a batch of the other path, the best of the rest counts
                if (v.trial_left <= 2) v.trial_best = (v.trial_best <= 0.0 || row_ms < v.trial_best) ? row_ms : v.trial_best;
                if (--v.trial_left == 0) {
                    if (v.trial_best > 0.0 && v.ms_row[v.mode] > 0.0 && v.trial_best < 0.95 * v.ms_row[v.mode]) {
                        v.ms_row[v.trial_mode] = v.trial_best;
                        v.mode = v.trial_mode; v.switches += 1; v.period = 32;
                    } else v.period = v.period < 512 ? 2 * v.period : 1024;
                    v.next_probe = v.batches + v.period;
                    v.trial_mode = -1;
                }
            } else if (!b.vit_trial && m == v.mode) {
                v.ms_row[m] = v.ms_row[m] > 0.0 ? 0.75 * v.ms_row[m] + 0.25 * row_ms : row_ms;
                // a trial of the other path: from the sparse path only where its bounds do not prune (cells refined in pass 4)
                const bool gate = v.mode == 1 || cells > h->vit_refine_gate * (double)b.total * (double)b.K;
                // before the dense kernels are tried: a longer warm-up of pass 2's chunks (snk_engine.h lb_warm_eff)
                if (v.mode == 0 && gate && h->lb_chunk > 0 && (h->lb_warm_eff > h->lb_warm ? h->lb_warm_eff : h->lb_warm) < h->lb_warm_long) {
                    h->lb_warm_eff = h->lb_warm_long; h->lb_warm_raises += 1;
                    v.ms_row[0] = 0.0;                             // (the sparse path's period is measured afresh)
                    v.next_probe = v.batches + 3;                  // (the batches in flight still ran with the short one)
                } else
                if (v.trial_left == 0 && v.batches >= v.next_probe && gate) {
                    v.trial_mode = 1 - v.mode; v.trial_left = 3; v.trial_best = 0.0; v.trials += 1;
                }
            }
        }
    }
    bool redone = false;
    for (int g = 0; g < b.n_groups; ++g) {
        if (status[g] == 0) continue;
        if (status[g] & 2) h->tie_overflow = 1;
        const int64_t r0 = b.offs[b.first[g]], rows = b.offs[b.first[g + 1]] - r0;
        const int saved = h->precision;
        const bool saved_sup = h->opt_suppress;
        // bit 8: a row's list was not proven complete under the optimistic thresholds (snk_engine.h): the fast path once more with
        // the guaranteed ones.  A lone list overflow: the fast path once more, non-deferred -- it takes the voice up its ladder
        // (longer lists, float32 operands: snk_engine.h knn_level) and later batches start there; anything else: the exact
        // float64 sweep
        // (beside another bit -- an overflowed pair list, pool or row list -- bit 8 is that overflow's doing, not the thresholds')
        if (status[g] == 8) { note_optimism_failure(h); h->opt_suppress = true; }
        const int st = status[g] & ~8;
        if (st && !((st & ~3) == 0 && h->knn_level < 2)) h->precision = 0;
        const int rc = knn_device(h, b.Qall.as<double>() + r0 * b.D, rows, b.K, nullptr,
                                  b.cand.as<int64_t>() + r0 * b.K, b.dist.as<double>() + r0 * b.K, nullptr);
        h->precision = saved;
        h->opt_suppress = saved_sup;
        if (rc) return rc;
        CHK(viterbi_group(h, g, b.offs.data(), b.first[g], b.first[g + 1], b.K, b.cand.as<int64_t>(), b.dist.as<double>(), false,
                          b.path.as<int64_t>(), b.plen.as<int64_t>(), b.cost.as<double>(), b.n_utts));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->batch_redos += 1;
        redone = true;
    }
    if (redone) {
        HIPCHK(hipMemcpyAsync(st, b.path.p, (size_t)b.total * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(st + sz_path, b.plen.p, (size_t)b.n_utts * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(st + sz_path + sz_u, b.cost.p, (size_t)b.n_utts * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    memcpy(path_out, st, (size_t)b.total * sizeof(int64_t));
    memcpy(path_len_out, st + sz_path, (size_t)b.n_utts * sizeof(int64_t));
    memcpy(cost_out, st + sz_path + sz_u, (size_t)b.n_utts * sizeof(double));
    collect_timers(h);
    return 0;
}

int snk_knn_viterbi_batch(snk_handle h, const double *Q, const int64_t *row_offsets, int n_utts, int D,
                          int K, int64_t *path_out, int64_t *path_len_out, double *cost_out)
{
    if (!path_out || !path_len_out || !cost_out) return fail("snk_knn_viterbi_batch: null/empty argument");
    if (h && any_batch_busy(h))
        return fail("snk_knn_viterbi_batch: a submitted batch is still in flight (collect it first)");
    int ticket = -1;
    CHK(snk_knn_viterbi_batch_submit(h, Q, row_offsets, n_utts, D, K, &ticket));
    return snk_knn_viterbi_batch_collect(h, ticket, path_out, path_len_out, cost_out);
}

// Viterbi of a batch of utterances whose candidates the caller already has (label-driven preselection:
// preselect_units_quinphone / monophone_then_acoustic, synth_halfphone.py:1315-1396): what the tail of
// snk_knn_viterbi_batch does -- join bounds, sparse exact recursion per group on the side streams -- without the K-NN.
int snk_viterbi_batch(snk_handle h, const int64_t *cand, const double *tdist, const int64_t *row_offsets, int n_utts, int K,
                      int64_t *path_out, int64_t *path_len_out, double *cost_out)
{
    CHK(check_ready(h, false, true));
    CHK(no_batch_in_flight(h, "snk_viterbi_batch"));
    HIPCHK(hipSetDevice(h->device));
    if (!cand || !tdist || !row_offsets || n_utts < 1 || !path_out || !path_len_out || !cost_out)
        return fail("snk_viterbi_batch: null/empty argument");
    if (K < 1 || K > 208) return fail("viterbi: n_candidates=%d outside 1..208", K);
    if (h->Njc != h->N + 1 && h->N > 0) return fail("snk_viterbi_batch: join_contexts rows != N+1");
    const int64_t total = row_offsets[n_utts];
    for (int u = 0; u < n_utts; ++u)
        if (row_offsets[u + 1] - row_offsets[u] < 1) return fail("snk_viterbi_batch: utterance %d has no rows", u);
    CHK(h->mcand.ensure((size_t)total * K * sizeof(int64_t)));
    CHK(h->mdist.ensure((size_t)total * K * sizeof(double)));
    CHK(h->res_path.ensure((size_t)total * sizeof(int64_t)));
    CHK(h->res_plen.ensure((size_t)n_utts * sizeof(int64_t)));
    CHK(h->res_cost.ensure((size_t)n_utts * sizeof(double)));
    {
        StageTimer t(h, h->stream, TM_H2D);
        CHK(h2d(h, h->mcand.p, cand, (size_t)total * K * sizeof(int64_t), h->stream));
        CHK(h2d(h, h->mdist.p, tdist, (size_t)total * K * sizeof(double), h->stream));
    }
    {
        const std::vector<int> first = group_utterances(h, row_offsets, n_utts, false, K);
        for (int g = 0; g + 1 < (int)first.size(); ++g)
            CHK(viterbi_group(h, g, row_offsets, first[g], first[g + 1], K, h->mcand.as<int64_t>(), h->mdist.as<double>(), true,
                              nullptr, nullptr, nullptr, n_utts));
    }
    for (int i = 0; i < 2; ++i) HIPCHK(hipStreamSynchronize(h->dp_stream[i]));
    HIPCHK(hipGetLastError());
    {
        StageTimer t(h, h->stream, TM_D2H);
        D2HPart parts[3] = {{path_out, h->res_path.p, (size_t)total * sizeof(int64_t)},
                            {path_len_out, h->res_plen.p, (size_t)n_utts * sizeof(int64_t)},
                            {cost_out, h->res_cost.p, (size_t)n_utts * sizeof(double)}};
        CHK(staged_d2h(h, h->stream, parts, 3));
    }
    collect_timers(h);
    return 0;
}
