// C ABI of libsnkhip.so (include/snk.h), part 1: errors, the engine's lifetime, host <-> device copies, database upload,
// stream weights and everything derived from them (operands of the K-NN prefilter, tile balls).
#include "snk_engine.h"

static thread_local std::string g_err;

int fail(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

const std::string &last_error_string() { return g_err; }

hipEvent_t ev_get(snk_engine *h)
{
    if (!h->ev_pool.empty()) { hipEvent_t e = h->ev_pool.back(); h->ev_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

void collect_timers(snk_engine *h)   // call after the streams were synchronised
{
    std::vector<EvPair> later;                // stages of a batch that is still in flight (submit / collect)
    for (auto &ep : h->pending) {
        float ms = 0.f;
        const hipError_t e = hipEventElapsedTime(&ms, ep.a, ep.b);
        if (e == hipErrorNotReady) { later.push_back(ep); continue; }
        if (e == hipSuccess) { h->tm_ms[ep.id] += ms; h->tm_n[ep.id] += 1; }
        h->ev_pool.push_back(ep.a);
        h->ev_pool.push_back(ep.b);
    }
    (void)hipGetLastError();
    h->pending.swap(later);
}

int staged_d2h(snk_engine *h, hipStream_t st, const D2HPart *parts, int n)
{
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += (parts[i].bytes + 63) & ~(size_t)63;
    CHK(h->hstage.ensure(total));
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        if (parts[i].bytes)
            HIPCHK(hipMemcpyAsync((char *)h->hstage.p + off, parts[i].src, parts[i].bytes, hipMemcpyDeviceToHost, st));
        off += (parts[i].bytes + 63) & ~(size_t)63;
    }
    HIPCHK(hipStreamSynchronize(st));
    off = 0;
    for (int i = 0; i < n; ++i) {
        if (parts[i].bytes) memcpy(parts[i].dst, (char *)h->hstage.p + off, parts[i].bytes);
        off += (parts[i].bytes + 63) & ~(size_t)63;
    }
    return 0;
}

// ---------------------------------------------------------------------------
// Host <-> device copies.  The GPU (its DMA engines, the runtime's copy kernels) only ever touches page-locked memory
// that THIS library allocated (hipHostMalloc) or that the caller registered (snk_host_register): caller buffers are
// ordinary pageable memory (numpy arrays, stack variables), and handing those to hipMemcpyAsync makes the runtime pin
// and map them on the fly -- a process that frees and reuses such memory all the time (a Python test session, a tuning
// loop) was seen to die with "Memory access fault by GPU node ... on address <an address of the host heap>" inside an
// unrelated call (HISTORY.md section 8.1).  Uploads go caller -> pinned staging (memcpy) -> device, results device ->
// pinned staging -> caller (staged_d2h).
// ---------------------------------------------------------------------------
bool host_memory_is_pinned(const void *p)
{
    static int bypass = -1;                                // developer switch: SNK_NO_STAGING=1 hands caller memory to the runtime as before
    if (bypass < 0) { const char *e = getenv("SNK_NO_STAGING"); bypass = (e && *e == '1') ? 1 : 0; }
    if (bypass) return true;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

// the engine's upload staging: a bump allocator over one pinned buffer; it wraps (after waiting for the stream: the
// transfers queued so far read it) when full.  Transfers are queued on `st` (always the engine's main stream).
int h2d(snk_engine *h, void *dst_dev, const void *src_host, size_t bytes, hipStream_t st)
{
    if (!bytes) return 0;
    if (host_memory_is_pinned(src_host)) {                 // registered by the caller / pinned by us: a plain queued copy
        HIPCHK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, st));
        return 0;
    }
    const size_t chunk_max = (size_t)64 << 20;
    const char *src = static_cast<const char *>(src_host);
    char *dst = static_cast<char *>(dst_dev);
    while (bytes) {
        const size_t n = bytes < chunk_max ? bytes : chunk_max;
        const size_t need = (n + 255) & ~(size_t)255;
        if (h->up_used + need > h->up.bytes) {
            HIPCHK(hipStreamSynchronize(st));              // everything queued out of the buffer has been read
            h->up_used = 0;
            if (need > h->up.bytes) CHK(h->up.ensure(need));
        }
        char *stage = static_cast<char *>(h->up.p) + h->up_used;
        memcpy(stage, src, n);
        HIPCHK(hipMemcpyAsync(dst, stage, n, hipMemcpyHostToDevice, st));
        h->up_used += need;
        src += n; dst += n; bytes -= n;
    }
    return 0;
}

// rows of row_bytes at src_pitch -> rows at dst_pitch (zero-filled padding); dst rows are contiguous at dst_pitch
int h2d_rows(snk_engine *h, void *dst_dev, size_t dst_pitch, const void *src_host, size_t src_pitch, size_t row_bytes,
                    size_t n_rows, hipStream_t st)
{
    if (!n_rows || !row_bytes) return 0;
    if (dst_pitch == src_pitch && dst_pitch == row_bytes) return h2d(h, dst_dev, src_host, row_bytes * n_rows, st);
    size_t per = ((size_t)32 << 20) / dst_pitch;
    if (per < 1) per = 1;
    const char *src = static_cast<const char *>(src_host);
    char *dst = static_cast<char *>(dst_dev);
    for (size_t r0 = 0; r0 < n_rows; r0 += per) {
        const size_t n = n_rows - r0 < per ? n_rows - r0 : per;
        const size_t need = (n * dst_pitch + 255) & ~(size_t)255;
        if (h->up_used + need > h->up.bytes) {
            HIPCHK(hipStreamSynchronize(st));
            h->up_used = 0;
            if (need > h->up.bytes) CHK(h->up.ensure(need));
        }
        char *stage = static_cast<char *>(h->up.p) + h->up_used;
        for (size_t r = 0; r < n; ++r) {
            memcpy(stage + r * dst_pitch, src + (r0 + r) * src_pitch, row_bytes);
            if (dst_pitch > row_bytes) memset(stage + r * dst_pitch + row_bytes, 0, dst_pitch - row_bytes);
        }
        HIPCHK(hipMemcpyAsync(dst + r0 * dst_pitch, stage, n * dst_pitch, hipMemcpyHostToDevice, st));
        h->up_used += need;
    }
    return 0;
}

// small synchronous upload (weights, masks, counters): staged, waited for
int h2d_sync(snk_engine *h, void *dst_dev, const void *src_host, size_t bytes)
{
    CHK(h2d(h, dst_dev, src_host, bytes, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// one device -> host transfer through the pinned staging, waited for
int d2h_sync(snk_engine *h, void *dst_host, const void *src_dev, size_t bytes, hipStream_t st)
{
    const D2HPart part = {dst_host, src_dev, bytes};
    return staged_d2h(h, st, &part, 1);
}

// an upload whose staging must outlive the call (submit / collect): the caller's own pinned buffer takes the copy
int h2d_via(HostBuf &stage, void *dst_dev, const void *src_host, size_t bytes, hipStream_t st)
{
    if (!bytes) return 0;
    if (host_memory_is_pinned(src_host)) {
        HIPCHK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, st));
        return 0;
    }
    CHK(stage.ensure(bytes));
    memcpy(stage.p, src_host, bytes);
    HIPCHK(hipMemcpyAsync(dst_dev, stage.p, bytes, hipMemcpyHostToDevice, st));
    return 0;
}

static int create_streams(snk_engine *h);

int snk_abi_version(void) { return 1; }
const char *snk_last_error(void) { return g_err.c_str(); }

int snk_device_count(int *count_out)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count_out = 0; return fail("hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count_out = n;
    return 0;
}

int snk_create(int device_id, snk_handle *out)
{
    if (!out) return fail("snk_create: null handle_out");
    *out = nullptr;
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (n <= 0) return fail("snk_create: no HIP device available (this engine has no CPU fallback)");
    if (device_id < 0 || device_id >= n) return fail("snk_create: device %d out of range (0..%d)", device_id, n - 1);
    HIPCHK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail("snk_create: device %d is %s; this library is built for gfx950 (MI355X) only",
                    device_id, prop.gcnArchName);
    snk_engine *h = new snk_engine();
    h->device = device_id;
    h->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (h->slabctr.ensure(64)) { delete h; return 1; }
    if (h->margin_stat.ensure(8 * sizeof(unsigned int))) { delete h; return 1; }
    int rc = create_streams(h);
    if (!rc) { const unsigned int init[8] = {0u, 0x7f800000u, 0u, 0u, 0u, 0u, 0u, 0u}; rc = h2d_sync(h, h->margin_stat.p, init, sizeof(init)); }
    if (rc) { (void)snk_destroy(h); return rc; }
    *out = h;
    return 0;
}

static int create_streams(snk_engine *h)
{
    HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    // Two side streams take the T-step recursions of alternate utterance groups.  (ROCm multiplexes
    // the streams of a process onto 4 hardware queues: with more side streams one of them shares a
    // queue with the main stream and a recursion stalls the K-NN sweep queued behind it.)
    HIPCHK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
    for (int i = 0; i < 8; ++i) {
        HIPCHK(hipEventCreateWithFlags(&h->slot[i].knn_done, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->slot[i].vit_done, hipEventDisableTiming));
    }
    h->dp_stream[0] = h->stream2;
    HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&h->knn_all_done, hipEventDisableTiming));
    for (auto &b : h->bslot) HIPCHK(hipEventCreateWithFlags(&b.done, hipEventDisableTiming));
    HIPCHK(hipStreamCreateWithFlags(&h->dp_stream[1], hipStreamNonBlocking));
    return 0;
}

int snk_destroy(snk_handle h)
{
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    collect_timers(h);
    h->tmask.release(); h->mcand.release(); h->mdist.release(); h->vstats.release(); h->perm.release(); h->perm2.release(); h->km_ws.release();
    (void)snk_comm_destroy(h);
    { DevBuf *cb[] = {&h->sh_d2, &h->sh_id, &h->sh_bound, &h->sh_rd2, &h->sh_rid, &h->sh_res, &h->sh_resall,
                      &h->gs_unw, &h->gs_w, &h->gs_norm, &h->gs_tiles, &h->gs_fmax2};
      for (auto *b : cb) b->release(); }
    DevBuf *bufs[] = {&h->F_unw, &h->JC_unw, &h->Fw, &h->fnorm, &h->JCw, &h->wt, &h->wj, &h->unit_class, &h->JW32, &h->jw_umax,
                      &h->Qraw, &h->Qp, &h->Qf, &h->qnorm, &h->thr, &h->gmin, &h->cnt, &h->lkey, &h->lidx,
                      &h->status, &h->qclass, &h->d2tmp, &h->slabctr, &h->pool, &h->poolctl, &h->chunkfill, &h->Dm, &h->gprev, &h->gblkmin, &h->gblkarg,
                      &h->gpath, &h->gdist, &h->gsync, &h->gtiles, &h->cls16_full, &h->cls16_samp,
                      &h->g32_blk, &h->g32_ctl, &h->g32_res,
                      &h->gh_nw, &h->gh_max, &h->gh_aq, &h->gh_qn2, &h->gh_W, &h->gtiles16};
    for (auto *b : bufs) b->release();
    if (h->dp_stream[1]) (void)hipStreamDestroy(h->dp_stream[1]);
    h->res_path.release(); h->res_plen.release(); h->res_cost.release(); h->Qall.release();
    h->rowflag.release(); h->exact_rows.release(); h->exact_scratch.release();
    h->frames_spec.release(); h->frames_fzv.release(); h->cc_in.release(); h->cc_out.release();
    h->res_status.release(); h->hstage.release(); h->up.release();
    { DevBuf *fb[] = {&h->a16h, &h->a16l, &h->s16h, &h->s16l, &h->b16h, &h->b16l, &h->eps16, &h->thr32, &h->gmin32, &h->fmax2, &h->gs_tiles_b, &h->cq16, &h->rho16, &h->gs_rho16, &h->kth16, &h->margin_stat, &h->e1_16, &h->thr1_32, &h->cpairs, &h->cpairctl,
                      &h->ball_c, &h->ball_cn, &h->ball_rad, &h->ball_c16, &h->ball_tq, &h->ball_nq,
                      &h->ball_aq, &h->ball_nql, &h->ball_gmin, &h->ball_bound,
                      &h->ball_c2, &h->ball_cn2, &h->ball_rad2, &h->ball_s16, &h->ball_mask,
                      &h->sh_cnt, &h->sh_off, &h->sh_tot, &h->sh_totall, &h->sh_plan, &h->sh_pack, &h->sh_rpack, &h->sh_offq};
      for (auto *b : fb) b->release(); }
    for (int i = 0; i < 8; ++i) {
        UttSlot &s = h->slot[i];
        DevBuf *sb[] = {&s.cand, &s.tdist, &s.J, &s.bp, &s.path, &s.plen, &s.cost, &s.Jlo, &s.scale, &s.sets, &s.cex};
        for (auto *b : sb) b->release();
        if (s.knn_done) (void)hipEventDestroy(s.knn_done);
        if (s.vit_done) (void)hipEventDestroy(s.vit_done);
    }
    for (auto e : h->ev_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < 4; ++i) { if (h->vit_t0[i]) (void)hipEventDestroy(h->vit_t0[i]); if (h->vit_t1[i]) (void)hipEventDestroy(h->vit_t1[i]); }
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->up_stream) (void)hipStreamDestroy(h->up_stream);
    for (auto &b : h->bslot) if (b.q_up) (void)hipEventDestroy(b.q_up);
    if (h->knn_all_done) (void)hipEventDestroy(h->knn_all_done);
    if (h->knn_mid) (void)hipEventDestroy(h->knn_mid);
    for (auto &t : h->sticket) {
        t.mcand.release(); t.mdist.release(); t.res_path.release(); t.res_plen.release(); t.res_cost.release(); t.status.release();
        t.stage.release(); t.qstage.release();
        if (t.done) (void)hipEventDestroy(t.done);
        if (t.main_done) (void)hipEventDestroy(t.main_done);
        for (int i = 0; i < 2; ++i) if (t.side_done[i]) (void)hipEventDestroy(t.side_done[i]);
    }
    for (auto &b : h->bslot) { b.Qall.release(); b.cand.release(); b.dist.release(); b.path.release(); b.plen.release(); b.cost.release(); b.status.release(); b.stage.release(); b.qstage.release(); if (b.done) (void)hipEventDestroy(b.done); if (b.knn_end) (void)hipEventDestroy(b.knn_end); }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    delete h;
    return 0;
}

static int upload_join(snk_engine *h, const float *JC_unw, int64_t Njc, int Dj)
{
    if (!JC_unw || Njc < 2 || Dj < 1) return fail("upload: bad join matrix (Njc=%lld Dj=%d)", (long long)Njc, Dj);
    h->Njc = Njc; h->Dj = Dj; h->Djpad = roundup(Dj, 32);
    h->Jp = roundup(Dj, 4);     // 16-byte aligned rows, zero-filled padding (greedy scan: 128-bit loads)
    CHK(h->JC_unw.ensure((size_t)Njc * h->Jp * sizeof(float)));
    if (h->Jp != Dj) HIPCHK(hipMemsetAsync(h->JC_unw.p, 0, (size_t)Njc * h->Jp * sizeof(float), h->stream));
    CHK(h->JCw.ensure((size_t)Njc * h->Djpad * sizeof(double)));
    CHK(h2d_rows(h, h->JC_unw.p, (size_t)h->Jp * sizeof(float), JC_unw, (size_t)Dj * sizeof(float), (size_t)Dj * sizeof(float),
                 (size_t)Njc, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->have_join = true;
    h->have_weights = false;
    // everything derived from the join matrix (greedy layout, float32 / float16 join tiles, their norms and range check)
    h->have_glay = false;
    h->gtiles_ready = false; h->gt16_ready = false; h->gt16_ok = false; h->gj_ready = false;
    h->jw32_ready = false;
    return 0;
}

int snk_upload_db(snk_handle h, const float *F_unw, int64_t N, int Dt, const float *JC_unw,
                  int64_t Njc, int Dj)
{
    CHK(no_batch_in_flight(h, "snk_upload_db"));
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    if (!F_unw || N < 1 || Dt < 1) return fail("snk_upload_db: bad target matrix (N=%lld Dt=%d)", (long long)N, Dt);
    // up to 256 columns the matrix sweeps serve the K-NN; wider rows (the doubled join rows of an epoch voice from
    // train_halfphone as a K-NN database: 2 x 151 columns, Synthesiser.join_knn) go through the canonical-distance
    // selection, a workgroup per query row (knn_device)
    if (Dt > 512) return fail("snk_upload_db: Dt=%d > 512 columns is not supported", Dt);
    if (N >= (1LL << 31) - 4096) return fail("snk_upload_db: N=%lld exceeds the 31-bit unit id range", (long long)N);
    if (JC_unw && Njc != N + 1) return fail("snk_upload_db: join_contexts must have N+1 rows (got %lld, N=%lld)", (long long)Njc, (long long)N);
    h->N = N; h->Dt = Dt; h->Dpad = roundup(Dt, SNK_DPAD);
    h->Nalloc = roundup(N, 16) + 16 * SNK_NT_MAX;
    h->Fp = roundup(Dt, 4);
    CHK(h->F_unw.ensure((size_t)N * h->Fp * sizeof(float)));
    if (h->Fp != Dt) HIPCHK(hipMemsetAsync(h->F_unw.p, 0, (size_t)N * h->Fp * sizeof(float), h->stream));
    CHK(h->Fw.ensure((size_t)h->Nalloc * h->Dpad * sizeof(double)));
    CHK(h->fnorm.ensure((size_t)h->Nalloc * sizeof(double)));
    CHK(h2d_rows(h, h->F_unw.p, (size_t)h->Fp * sizeof(float), F_unw, (size_t)Dt * sizeof(float), (size_t)Dt * sizeof(float),
                 (size_t)N, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->have_db = true;
    h->have_weights = false;
    h->have_classes = false;
    for (auto &b : h->bslot) b.q_rows = -1;                  // a new voice: the next batch submit carries its query rows
    h->have_glay = false;
    h->gtiles_ready = false; h->gt16_ready = false;
    h->gh_ready = false; h->gj_ready = false;
    h->gs_rows = 0; h->gs_ready = false;
    h->perm_ready = false; h->reorder_useless = false; h->reorder_done = false; h->reorder_pending = false;      // a new voice: the database order again
    h->reorder_radius_before = 0.0; h->reorder_radius_after = 0.0;
    if (h->global_N < 0) { h->shard_offset = 0; }
    if (JC_unw) CHK(upload_join(h, JC_unw, Njc, Dj));
    return 0;
}

int snk_upload_join_only(snk_handle h, const float *JC_unw, int64_t Njc, int Dj)
{
    if (!h) return fail("null handle");
    CHK(no_batch_in_flight(h, "snk_upload_join_only"));
    HIPCHK(hipSetDevice(h->device));
    return upload_join(h, JC_unw, Njc, Dj);
}

int snk_set_shard(snk_handle h, int64_t global_row_offset, int64_t global_N)
{
    if (!h) return fail("null handle");
    if (global_row_offset < 0 || global_N < 1) return fail("snk_set_shard: bad arguments");
    // merged lists carry unit ids through 32-bit sort keys (merge_topk_kernel)
    if (global_N >= (1LL << 31)) return fail("snk_set_shard: global_N=%lld exceeds the 31-bit unit id range", (long long)global_N);
    h->shard_offset = global_row_offset;
    h->global_N = global_N;
    return 0;
}

// Stream truncation (truncate_target_streams / truncate_join_streams, synth_simple.py:982-992; the reference
// drops the columns from its weighted copies and from the query rows).  Here the columns stay in place:
// the next snk_set_weights gives them weight 0 and uploaded query rows get them zeroed, so each adds
// exactly +0.0 to every squared distance -- same candidates, distances and paths as dropping them.
// cols: ascending indices of the columns that take part; n < 0: all columns.
int snk_set_column_selection(snk_handle h, const int *tcols, int nt, const int *jcols, int nj)
{
    CHK(no_batch_in_flight(h, "snk_set_column_selection"));
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    auto build = [&](const int *cols, int n, int width, std::vector<double> &sel, const char *what) -> int {
        sel.clear();
        if (n < 0) return 0;
        if (width <= 0) return fail("snk_set_column_selection: no %s matrix uploaded", what);
        if (n > 0 && !cols) return fail("snk_set_column_selection: null %s column list", what);
        sel.assign((size_t)width, 0.0);
        for (int i = 0; i < n; ++i) {
            if (cols[i] < 0 || cols[i] >= width) return fail("snk_set_column_selection: %s column %d outside 0..%d", what, cols[i], width - 1);
            sel[(size_t)cols[i]] = 1.0;
        }
        return 0;
    };
    CHK(build(tcols, nt, h->have_db ? h->Dt : 0, h->tsel, "target"));
    CHK(build(jcols, nj, h->have_join ? h->Dj : 0, h->jsel, "join"));
    if (!h->tsel.empty()) {
        CHK(h->tmask.ensure(h->tsel.size() * sizeof(double)));
        CHK(h2d_sync(h, h->tmask.p, h->tsel.data(), h->tsel.size() * sizeof(double)));
    }
    h->have_weights = false;                 // takes effect with the next snk_set_weights
    // rows resident in the batch workspaces were masked with the selection of their upload: the next submit must carry Q
    for (auto &b : h->bslot) b.q_rows = -1;
    return 0;
}

// Rows of 257 .. 512 columns: the operands of the blocked bf16-split product (knn_wide16b), built at the FIRST call that can take
// that path -- plain snk_knn calls; batch, sharded and class-restricted engines never pay the memory (N x Dpad x 4 bytes) or the
// host synchronisation.  (nt16_eff, n_slabs16, stride16, eps_c_bf are shared with the narrow-row plan: a database is one
// or the other.)
int ensure_wide_operands(snk_engine *h)
{
    if (h->wide16_ready || h->wide16_tried) return 0;
    h->wide16_tried = true;
    if (h->have_db && h->prefilter >= 1 && knn_wide16b_supported(h->Dt, h->Dpad)) {
        // rows of 257 .. 512 columns (Synthesiser.join_knn on the doubled join rows of an epoch voice): bf16-split operands
        // of the whole database and of the stage-A sample, one tile per slab; the blocked product of knn_wide16b serves
        // both stages, the exact float64 re-rank is the one of every other width
        const int terms = h->prefilter == 2 ? 4 : 3;
        h->nt16_eff = 1;
        CHK(h->fmax2.ensure(sizeof(double)));
        launch_fmax(h->fnorm.as<double>(), h->N, h->fmax2.as<double>(), h->stream);
        double fmax2 = 0.0;
        CHK(d2h_sync(h, &fmax2, h->fmax2.p, sizeof(double), h->stream));
        h->n_slabs16 = (h->N + 31) / 32;
        int64_t stride = (int64_t)floor(1.0 / h->sample_frac + 0.5);
        if (stride < 1) stride = 1;
        while (stride > 1 && (h->N / stride) / 32 < h->min_sample_slabs) --stride;
        h->stride16 = stride;
        h->n_slabs16_a = (h->N / stride) / 32;
        if (fmax2 < 1.0e30 && h->n_slabs16_a >= 1) {
            const size_t per_tile = (size_t)8 * 64 * 16 * (h->Dpad / 64);
            h->eps_c_bf = 1.02 * (SNK_BF16_MFMA_UNIT * (double)(terms * 4 + 1) + 6e-8 * (double)(2 * (h->Dpad / 64) + 1));
            CHK(h->rho16.ensure(2 * sizeof(double)));
            launch_db16b_ratios(h->Fw.as<double>(), h->N, h->Dt, h->Dpad, h->rho16.as<double>(), h->stream);
            CHK(h->a16l.ensure(h->n_slabs16 * per_tile));
            CHK(h->s16l.ensure(h->n_slabs16_a * per_tile));
            launch_build_db16b(h->Fw.as<double>(), h->fnorm.as<double>(), h->N, h->Dt, h->Dpad, h->n_slabs16, 0, 0, 1, h->a16l.p, h->stream);
            launch_build_db16b(h->Fw.as<double>(), h->fnorm.as<double>(), h->N, h->Dt, h->Dpad, h->n_slabs16_a, stride,
                               2 * h->n_slabs16_a, 1, h->s16l.p, h->stream);
            HIPCHK(hipGetLastError());
            h->wide16_ready = true;
        }
    }
    return 0;
}

// The operands of the K-NN prefilter for the current weights (knn16_kernels.hip): float32 and bf16-split copies of the weighted
// database and of its stage-A sample, tile balls and the balls of 32 tiles -- in the database's order, or in the order the engine
// gave the voice (h->perm: kmeans_kernels.hip).  Resets the filter's latches: the voice is judged afresh on these operands.
int build_prefilter_operands(snk_engine *h)
{
    const int32_t *perm = h->perm_ready ? h->perm.as<int32_t>() : nullptr;
    h->f16_ready = false;
    h->cls16_ready = false;
    h->operand_gen += 1;
    if (h->have_db && h->Dpad <= 256 && h->Dpad - h->Dt >= 1) {
        // tiles per wavefront: the database fragments of a slab stay in registers (32 * Dpad / 64
        // floats per tile and lane), so wider rows leave room for fewer tiles
        const int dch16 = h->Dpad / 64;
        const int nt = (dch16 == 1) ? h->nt16 : (dch16 == 2) ? 2 : 1;
        h->nt16_eff = nt;
        // |key~ - key| <= c (2 |q| Fmax + Fmax^2): operand rounding 2^-24 each and an f32 FMA chain of
        // Dpad + 1 terms; c = 2 x that (8e-6 at Dpad = 64)
        h->eps_c = 2.0 * (double)(h->Dpad + 3) * 5.9604644775390625e-08;
        CHK(h->fmax2.ensure(sizeof(double)));
        launch_fmax(h->fnorm.as<double>(), h->N, h->fmax2.as<double>(), h->stream);
        double fmax2 = 0.0;
        CHK(d2h_sync(h, &fmax2, h->fmax2.p, sizeof(double), h->stream));
        const int64_t slab_rows = 32 * nt;
        h->n_slabs16 = (h->N + slab_rows - 1) / slab_rows;
        int64_t stride = (int64_t)floor(1.0 / h->sample_frac + 0.5);
        if (stride < 1) stride = 1;
        // small databases (one rank's shard of a row-sharded one): keep >= 512 sample groups so the
        // K-th smallest group minimum stays close to the K-th nearest sampled unit
        while (stride > 1 && (h->N / stride) / slab_rows < h->min_sample_slabs) --stride;
        h->stride16 = stride;
        h->n_slabs16_a = (h->N / stride) / slab_rows;
        if (fmax2 < 1.0e30 && h->n_slabs16_a >= 1) {
            const int64_t tiles_b = h->n_slabs16 * nt, tiles_a = h->n_slabs16_a * nt;
            const size_t per_tile = (size_t)8 * 64 * 16 * dch16;
            CHK(h->a16h.ensure(tiles_b * per_tile));
            CHK(h->s16h.ensure(tiles_a * per_tile));
            launch_build_db16(h->Fw.as<double>(), h->fnorm.as<double>(), h->N, h->Dt, h->Dpad, tiles_b, 0, 0, nt,
                              h->a16h.p, h->stream, perm);
            launch_build_db16(h->Fw.as<double>(), h->fnorm.as<double>(), h->N, h->Dt, h->Dpad, tiles_a, stride,
                              2 * h->n_slabs16_a, nt, h->s16h.p, h->stream, perm);
            HIPCHK(hipGetLastError());
            h->f16_ready = true;
            h->bf16_ready = false;
            if (h->prefilter >= 1 && knn_sweep16b_supported(nt, dch16, h->Dt, h->Dpad, false)) {
                // key bound = cq ||f|| (what the split drops: measured, prepare_queries16b_kernel) + c_acc (...):
                // 2^-20 per MFMA over the 4 `terms` MFMAs of a chunk's chain and the norm pieces' 2^-24 (knn16_kernels.hip)
                const int terms = h->prefilter == 2 ? 4 : 3;
                // (chains of one 64-column chunk: 4 `terms` MFMAs; the chunks' sums are added in float32)
                h->eps_c_bf = 1.02 * (SNK_BF16_MFMA_UNIT * (double)(terms * 4 + 1) + 6e-8 * (double)(2 * (h->Dpad / 64) + 1));
                CHK(h->rho16.ensure(2 * sizeof(double)));
                launch_db16b_ratios(h->Fw.as<double>(), h->N, h->Dt, h->Dpad, h->rho16.as<double>(), h->stream);
                CHK(h->a16l.ensure(tiles_b * per_tile));
                CHK(h->s16l.ensure(tiles_a * per_tile));
                launch_build_db16b(h->Fw.as<double>(), h->fnorm.as<double>(), h->N, h->Dt, h->Dpad, tiles_b, 0, 0, nt,
                                   h->a16l.p, h->stream, perm);
                launch_build_db16b(h->Fw.as<double>(), h->fnorm.as<double>(), h->N, h->Dt, h->Dpad, tiles_a, stride,
                                   2 * h->n_slabs16_a, nt, h->s16l.p, h->stream, perm);
                HIPCHK(hipGetLastError());
                h->bf16_ready = true;
                // pass 0 of the two-pass filter: centre and radius of every 32-unit tile, the centres as one more bf16-split
                // operand (its dropped-piece ratios join the database's: one key bound serves both)
                h->ball_tiles = 0;
                h->filter_coarse = false; h->filter_onepass = false;
                h->filter_calls = 0; h->probe_next = 16; h->probe_period = 16; h->probe_ran = 0;
                if (h->prefilter_balls) {
                    const int64_t vt = (h->N + 31) / 32, ct = (vt + 31) / 32;
                    CHK(h->ball_c.ensure((size_t)vt * h->Dpad * sizeof(double)));
                    CHK(h->ball_cn.ensure((size_t)vt * sizeof(double)));
                    CHK(h->ball_rad.ensure((size_t)vt * sizeof(float)));
                    CHK(h->ball_c16.ensure((size_t)ct * per_tile));
                    launch_build_tile_balls(h->Fw.as<double>(), h->N, h->Dt, h->Dpad, vt, h->ball_c.as<double>(), h->ball_cn.as<double>(),
                                            h->ball_rad.as<float>(), h->stream, perm);
                    launch_db16b_ratios(h->ball_c.as<double>(), vt, h->Dt, h->Dpad, h->rho16.as<double>(), h->stream, true);
                    launch_build_db16b(h->ball_c.as<double>(), h->ball_cn.as<double>(), vt, h->Dt, h->Dpad, ct, 0, 0, nt, h->ball_c16.p, h->stream);
                    HIPCHK(hipGetLastError());
                    h->ball_tiles = vt;
                    // one level up: the balls of 32 consecutive tiles, as one more operand in the same format
                    h->ball_supers = 0;
                    if (ct >= 64) {
                        const int64_t ct2 = (ct + 31) / 32;
                        CHK(h->ball_c2.ensure((size_t)ct * h->Dpad * sizeof(double)));
                        CHK(h->ball_cn2.ensure((size_t)ct * sizeof(double)));
                        CHK(h->ball_rad2.ensure((size_t)ct * sizeof(float)));
                        CHK(h->ball_s16.ensure((size_t)ct2 * per_tile));
                        launch_build_super_balls(h->ball_c.as<double>(), h->ball_rad.as<float>(), h->N, vt, h->Dt, h->Dpad, ct, h->ball_c2.as<double>(),
                                                 h->ball_cn2.as<double>(), h->ball_rad2.as<float>(), h->stream);
                        launch_db16b_ratios(h->ball_c2.as<double>(), ct, h->Dt, h->Dpad, h->rho16.as<double>(), h->stream, true);
                        launch_build_db16b(h->ball_c2.as<double>(), h->ball_cn2.as<double>(), ct, h->Dt, h->Dpad, ct2, 0, 0, nt, h->ball_s16.p, h->stream);
                        HIPCHK(hipGetLastError());
                        h->ball_supers = ct;
                    }
                }
            }
        }
    }
    return 0;
}

// A voice whose tiles are not compact (the ball pass listed more than coarse_gate_fraction of all tile pairs): cluster its units
// and lay the prefilter's operands out cluster by cluster.  Once per set of weights; a voice the clustering does not help (the
// ball pass lists too much again: units spread like a cloud, not like a curve) is not clustered again until a new database comes.
// mean radius of the tiles' balls of the operands as they stand (0: no balls)
static int mean_tile_radius(snk_engine *h, double *out)
{
    *out = 0.0;
    if (h->ball_tiles < 1) return 0;
    std::vector<float> r((size_t)h->ball_tiles);
    CHK(d2h_sync(h, r.data(), h->ball_rad.p, r.size() * sizeof(float), h->stream));
    double s = 0.0;
    for (float v : r) s += (double)v;
    *out = s / (double)r.size();
    return 0;
}

// The order only buys speed: whatever goes wrong in here (an allocation on a nearly full device, a launch, a copy, an order that
// is no permutation) must not fail the caller's K-NN call -- reorder_units() below catches it, puts the operands back in the
// order they had and carries on.  *swapped: h->perm / h->perm2 are exchanged at the moment of the failure.
static int reorder_units_try(snk_engine *h, bool *swapped, bool had)
{
    double r_old = 0.0, r_new = 0.0;
    CHK(mean_tile_radius(h, &r_old));
    CHK(h->perm2.ensure((size_t)h->N * sizeof(int32_t)));
    size_t ws = kmeans_workspace_bytes(h->N, h->Dt);
    if (ws < kmeans_perm_check_bytes(h->N)) ws = kmeans_perm_check_bytes(h->N);
    CHK(h->km_ws.ensure(ws));
    {
        StageTimer t(h, h->stream, TM_WEIGHTS);
        launch_kmeans_order(h->Fw.as<double>(), h->N, h->Dt, h->Dpad, h->reorder_iters, h->km_ws.p, h->perm2.as<int32_t>(), h->stream);
    }
    HIPCHK(hipGetLastError());
    {
        // a bad order would silently drop units from the prefilter: every unit must appear exactly once (non-finite features
        // can break the chains' argmins)
        launch_perm_check(h->perm2.as<int32_t>(), h->N, h->km_ws.p, h->stream);
        unsigned int bad = 1u;
        CHK(d2h_sync(h, &bad, h->km_ws.p, sizeof(bad), h->stream));
        if (bad) return fail("reorder: the clustered order is not a permutation of the units (%u bad entries)", bad);
    }
    std::swap(h->perm, h->perm2);
    *swapped = true;
    h->perm_ready = true;
    CHK(build_prefilter_operands(h));
    HIPCHK(hipGetLastError());
    CHK(mean_tile_radius(h, &r_new));
    h->reorder_radius_before = r_old; h->reorder_radius_after = r_new;
    if (r_new < 0.8 * r_old) { h->reorders += 1; return 0; }
    // the clusters' tiles are no tighter than the order the voice had (consecutive frames of a frame-level voice questioned by a
    // batch of far-away rows; units spread like a cloud): that order stays, and the voice is not clustered again
    std::swap(h->perm, h->perm2);
    *swapped = false;
    h->perm_ready = had;
    h->reorder_useless = true;
    CHK(build_prefilter_operands(h));
    HIPCHK(hipGetLastError());
    return 0;
}

int reorder_units(snk_engine *h)
{
    h->reorder_pending = false;
    h->reorder_done = true;
    if (!h->have_db || !h->have_weights || !kmeans_supported(h->Dt) || h->N < 4096 || h->ball_tiles < 1) return 0;
    const bool had = h->perm_ready;
    bool swapped = false;
    if (reorder_units_try(h, &swapped, had) == 0) return 0;
    // failed: the voice keeps the order it had and is not asked again (until the weights change); the caller's search goes on
    (void)hipGetLastError();
    h->reorder_failures += 1;
    h->reorder_useless = true;
    if (swapped) std::swap(h->perm, h->perm2);
    h->perm_ready = had;
    CHK(build_prefilter_operands(h));         // (this one must work: the operands may be half rebuilt)
    HIPCHK(hipGetLastError());
    return 0;
}

int snk_set_weights(snk_handle h, const double *wt, int n_wt, const double *wj, int n_wj)
{
    CHK(no_batch_in_flight(h, "snk_set_weights"));
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    if (!h->have_db && !h->have_join) return fail("snk_set_weights: no database uploaded");
    if (h->have_db) {
        if (!wt || n_wt != h->Dt)
            return fail("snk_set_weights: target weight vector has %d entries, database has %d columns", n_wt, h->Dt);
        CHK(h->wt.ensure((size_t)n_wt * sizeof(double)));
        std::vector<double> eff(wt, wt + n_wt);
        if (!h->tsel.empty()) {
            if ((int)h->tsel.size() != n_wt) return fail("snk_set_weights: the target column selection was made for %d columns", (int)h->tsel.size());
            for (int c = 0; c < n_wt; ++c) eff[(size_t)c] *= h->tsel[(size_t)c];
        }
        CHK(h2d_sync(h, h->wt.p, eff.data(), (size_t)n_wt * sizeof(double)));
    }
    if (h->have_join) {
        if (!wj || n_wj != h->Dj)
            return fail("snk_set_weights: join weight vector has %d entries, join_contexts has %d columns", n_wj, h->Dj);
        CHK(h->wj.ensure((size_t)n_wj * sizeof(double)));
        std::vector<double> eff(wj, wj + n_wj);
        if (!h->jsel.empty()) {
            if ((int)h->jsel.size() != n_wj) return fail("snk_set_weights: the join column selection was made for %d columns", (int)h->jsel.size());
            for (int c = 0; c < n_wj; ++c) eff[(size_t)c] *= h->jsel[(size_t)c];
        }
        CHK(h2d_sync(h, h->wj.p, eff.data(), (size_t)n_wj * sizeof(double)));
    }
    {
        StageTimer t(h, h->stream, TM_WEIGHTS);
        if (h->have_db)
            launch_weight_target(h->F_unw.as<float>(), h->Fp, h->N, h->Dt, h->wt.as<double>(), h->Fw.as<double>(),
                                 h->fnorm.as<double>(), h->Nalloc, h->Dpad, nullptr, h->stream);
        if (h->have_join)
            launch_weight_join(h->JC_unw.as<float>(), h->Jp, h->Njc, h->Dj, h->wj.as<double>(), h->JCw.as<double>(),
                               h->Djpad, h->stream);
    }
    HIPCHK(hipGetLastError());
    h->gh_ready = false; h->gj_ready = false;                  // window norms of the hoisted greedy target term follow the target weights
    h->jw32_ready = false;                                     // ... and the float32 copy of the weighted join rows the join weights
    // float32 operands of the prefilter (knn16_kernels.hip): ||f||^2 rides in ONE spare padding column
    h->knn_level = 0;
    h->opt_off = false; h->opt_calls = 0; h->opt_fails = 0;    // a new set of weights: the optimistic thresholds get another chance
    h->reorder_done = false; h->reorder_pending = false;     // (the order a voice was given stays: any order is valid, and usually still a good one)
    h->vit = snk_engine::VitLatch();                         // a new set of weights: the Viterbi latch starts over
    h->lb_warm_eff = 0;                                      // ... and pass 2's warm-up is the short one again
    CHK(build_prefilter_operands(h));
    h->wide16_ready = false; h->wide16_tried = false;              // (built at the first call that can use them: ensure_wide_operands)
    h->gs_ready = false;
    if (h->gs_rows > 0 && h->f16_ready) {
        // the replicated global sample in the operand layout of stage A (groups scattered over the sample)
        const int nt = h->nt16_eff;
        h->gs_slabs = h->gs_rows / (32 * nt);
        if (h->gs_slabs >= 1) {
            const int64_t rows_alloc = roundup(h->gs_rows, 16) + 16 * SNK_NT_MAX;
            CHK(h->gs_w.ensure((size_t)rows_alloc * h->Dpad * sizeof(double)));
            CHK(h->gs_norm.ensure((size_t)rows_alloc * sizeof(double)));
            CHK(h->gs_fmax2.ensure(sizeof(double)));
            launch_weight_target(h->gs_unw.as<float>(), h->Fp, h->gs_rows, h->Dt, h->wt.as<double>(), h->gs_w.as<double>(),
                                 h->gs_norm.as<double>(), rows_alloc, h->Dpad, nullptr, h->stream);
            launch_fmax(h->gs_norm.as<double>(), h->gs_rows, h->gs_fmax2.as<double>(), h->stream);
            const int64_t tiles = h->gs_slabs * nt;
            CHK(h->gs_tiles.ensure((size_t)tiles * 8 * 64 * 16 * (h->Dpad / 64)));
            launch_build_db16(h->gs_w.as<double>(), h->gs_norm.as<double>(), h->gs_rows, h->Dt, h->Dpad, tiles, 1,
                              2 * h->gs_slabs, nt, h->gs_tiles.p, h->stream);
            if (h->bf16_ready) {
                CHK(h->gs_rho16.ensure(2 * sizeof(double)));
                launch_db16b_ratios(h->gs_w.as<double>(), h->gs_rows, h->Dt, h->Dpad, h->gs_rho16.as<double>(), h->stream);
                CHK(h->gs_tiles_b.ensure((size_t)tiles * 8 * 64 * 16 * (h->Dpad / 64)));
                launch_build_db16b(h->gs_w.as<double>(), h->gs_norm.as<double>(), h->gs_rows, h->Dt, h->Dpad, tiles, 1,
                                   2 * h->gs_slabs, nt, h->gs_tiles_b.p, h->stream);
            }
            HIPCHK(hipGetLastError());
            h->gs_ready = true;
        }
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    collect_timers(h);
    h->have_weights = true;
    return 0;
}

// Row-sharded databases: every rank also keeps a SAMPLE of the whole database (every s-th unit, chosen by the
// caller: 1/16 of 1 M units x 61 columns is 17 MB) and bounds the K-th nearest key of ITS share of the query
// rows against it -- the bound a single GPU would compute -- instead of every rank bounding every row against
// its own shard's sample.  Takes effect with the next snk_set_weights; rows in the database's column layout.
int snk_upload_global_sample(snk_handle h, const float *F_sample_unw, int64_t n_rows, int Dt)
{
    if (!h) return fail("null handle");
    CHK(no_batch_in_flight(h, "snk_upload_global_sample"));
    HIPCHK(hipSetDevice(h->device));
    if (!h->have_db) return fail("snk_upload_global_sample: upload the database shard first");
    if (!F_sample_unw || n_rows < 1 || Dt != h->Dt) return fail("snk_upload_global_sample: bad sample matrix (rows=%lld Dt=%d, database Dt=%d)", (long long)n_rows, Dt, h->Dt);
    CHK(h->gs_unw.ensure((size_t)n_rows * h->Fp * sizeof(float)));
    if (h->Fp != Dt) HIPCHK(hipMemsetAsync(h->gs_unw.p, 0, (size_t)n_rows * h->Fp * sizeof(float), h->stream));
    CHK(h2d_rows(h, h->gs_unw.p, (size_t)h->Fp * sizeof(float), F_sample_unw, (size_t)Dt * sizeof(float), (size_t)Dt * sizeof(float),
                 (size_t)n_rows, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->gs_rows = n_rows;
    h->gs_ready = false;
    h->have_weights = false;
    return 0;
}

int snk_set_unit_classes(snk_handle h, const int32_t *unit_class, int64_t N)
{
    CHK(no_batch_in_flight(h, "snk_set_unit_classes"));
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    if (!h->have_db || N != h->N) return fail("snk_set_unit_classes: N=%lld does not match the database (%lld)", (long long)N, (long long)h->N);
    CHK(h->unit_class.ensure((size_t)h->Nalloc * sizeof(int32_t)));
    HIPCHK(hipMemsetAsync(h->unit_class.p, 0xff, (size_t)h->Nalloc * sizeof(int32_t), h->stream));
    CHK(h2d(h, h->unit_class.p, unit_class, (size_t)N * sizeof(int32_t), h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->have_classes = true;
    h->cls16_ready = false;
    return 0;
}

int check_ready(snk_engine *h, bool need_target, bool need_join)
{
    if (!h) return fail("null handle");
    if (need_target && !h->have_db) return fail("no unit database uploaded (snk_upload_db)");
    if (need_join && !h->have_join) return fail("no join_contexts uploaded");
    if (!h->have_weights) return fail("weights not set (snk_set_weights)");
    return 0;
}

// state changes (database, weights, classes) while a submitted batch is still in flight would be seen by it
int no_batch_in_flight(snk_engine *h, const char *who)
{
    if (h && any_batch_busy(h))
        return fail("%s: a submitted batch is still in flight (snk_knn_viterbi_batch_collect it first)", who);
    if (h && (h->sticket[0].busy || h->sticket[1].busy))
        return fail("%s: a submitted sharded step is still in flight (snk_sharded_knn_viterbi_batch_collect it first)", who);
    return 0;
}

// Page-lock a caller buffer that is uploaded again and again (the query rows of a tune set): a copy
// from pageable memory makes the host wait for the stream, a copy from registered memory is queued.
int snk_host_register(void *ptr, size_t bytes)
{
    if (!ptr || !bytes) return fail("snk_host_register: null/empty buffer");
    HIPCHK(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return 0;
}

int snk_host_unregister(void *ptr)
{
    if (!ptr) return fail("snk_host_unregister: null buffer");
    HIPCHK(hipHostUnregister(ptr));
    return 0;
}
