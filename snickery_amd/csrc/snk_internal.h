// Internal declarations shared by the HIP translation units of libsnkhip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// every launch of the library goes through this form (see trace_launch below)
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel_, grid_, block_, lds_, stream_, ...)                          \
    do {                                                                                        \
        kernel_<<<(grid_), (block_), (lds_), (stream_)>>>(__VA_ARGS__);                         \
        snk::trace_launch(#kernel_, __FILE__, __LINE__, (stream_));                             \
    } while (0)

#define SNK_DPAD 64          // K-NN feature chunk: columns padded to multiples of 64 doubles
#define SNK_NT_MAX 8         // max DB tiles (16 rows each) a wave keeps in registers
#define SNK_VERY_BIG 1000000000000000.0   // const.py:3

#include <mutex>
#include <stdio.h>
#include <stdlib.h>

namespace snk {

// Debug aid (environment SNK_TRACE=1, read once): every kernel launch of the library is named on stderr and waited for,
// so the line before the runtime's "Memory access fault" message is the kernel that faulted.  Off: one predictable branch.
inline bool trace_on()
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("SNK_TRACE"); on = (e && *e && *e != '0') ? 1 : 0; }
    return on == 1;
}
inline void trace_launch(const char *kernel, const char *file, int line, hipStream_t s)
{
    if (!trace_on()) return;
    fprintf(stderr, "[snk-trace] %s (%s:%d) ...", kernel, file, line);
    fflush(stderr);
    const hipError_t e1 = hipGetLastError();
    const hipError_t e2 = hipStreamSynchronize(s);
    fprintf(stderr, " %s\n", e1 != hipSuccess ? hipGetErrorString(e1) : e2 != hipSuccess ? hipGetErrorString(e2) : "ok");
    fflush(stderr);
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the device that is current when it is called: the
// "already raised to" state of a kernel is kept per device (a process may hold engines on several), under a lock
// (engines on different devices may be driven from different threads).  `set` raises the attribute(s) and runs UNDER the
// lock, before the new state is recorded: a second thread never sees "raised" ahead of the call that raises it.
template <typename F>
inline void lds_attr_ensure(size_t (&per_device)[32], size_t want, F &&set)
{
    static std::mutex m;
    int d = 0;
    (void)hipGetDevice(&d);
    std::lock_guard<std::mutex> lock(m);
    size_t &have = per_device[d & 31];
    if (want <= have) return;
    set();
    have = want;
}

// ---- database preparation -------------------------------------------------
void launch_weight_target(const float *F_unw, int Fp, int64_t N, int Dt, const double *wt,
                          double *Fw, double *fnorm, int64_t Nalloc, int Dpad,
                          const int32_t *unit_class, hipStream_t s);
void launch_weight_join(const float *JC_unw, int Jp, int64_t Njc, int Dj, const double *wj,
                        double *JCw, int Djpad, hipStream_t s);

// ---- K-NN -----------------------------------------------------------------
struct KnnPlan {
    int nt;              // DB tiles per wave (slab = 16*nt rows)
    int dch;             // Dpad / 64
    int64_t n_slabs;     // slabs covering the DB
    int64_t a_stride, a_count;   // stage-A sample: every a_stride-th DB row, a_count virtual slabs
    int64_t row_limit;           // first padding row (rows >= N have norm +inf)
    int grid_cus;                // compute units the persistent sweep may occupy
    unsigned int *slab_counter;  // device word: dynamic slab dispenser
};

void launch_mask_columns(double *Q, int64_t T, int D, const double *mask, hipStream_t s);
void launch_prepare_queries(const double *Q, int64_t T, int D, double *Qp, double *Qf, double *qnorm,
                            int64_t Tpad, int Dpad, hipStream_t s);
// Qf = fragment-order copy of the padded queries (see prepare_queries_kernel)
// stage A: per-(row, lane-group) minima over the sampled slabs
void launch_knn_minima(const KnnPlan &p, const double *Fw, const double *fnorm,
                       const double *Qp, int64_t Tpad, double *gmin, int64_t G,
                       const int32_t *unit_class, const int32_t *query_class, hipStream_t s);
// K-th smallest of the minima -> per-row key-space threshold
void launch_knn_threshold(const double *gmin, int64_t G, int64_t T, int64_t Tpad, int K,
                          double *thr, int keep_min, hipStream_t s);
void launch_fill_threshold(double *thr, int64_t T, int64_t Tpad, double value, hipStream_t s);
// stage B: filtered sweep of the whole DB; passing (row, unit, key) entries go to wave-private
// chunks of a global entry pool; bucket scatters the pool into per-row lists
void launch_knn_filter(const KnnPlan &p, const double *Fw, const double *fnorm,
                       const double *Qp, const double *thr, int64_t Tpad,
                       void *pool, unsigned int *pool_ctl, int *chunk_fill, int max_chunks,
                       const int32_t *unit_class, const int32_t *query_class, hipStream_t s);
size_t knn_pool_bytes(int max_chunks);
void launch_knn_reset(int *cnt, int64_t Tpad, int *status, unsigned int *pool_ctl,
                      unsigned int *slab_counter, int *chunk_fill, int max_chunks, hipStream_t s,
                      unsigned int *x0 = nullptr, long long n0 = 0, unsigned int *x1 = nullptr, long long n1 = 0,
                      unsigned int *x2 = nullptr, long long n2 = 0);      // + up to three regions of words to clear
void launch_knn_bucket(const void *pool, const unsigned int *pool_ctl, const int *chunk_fill,
                       int max_chunks, int64_t Tpad, int64_t n_valid, int *cnt, double *lkey, int *lidx,
                       int cap, int *status, hipStream_t s, const int32_t *perm = nullptr,      // perm: the entries carry POSITIONS of a reordered operand
                       bool keys_f32 = false);                                                  // the entries' keys are float32 (the knn16_kernels.hip filters)
// stage C: per-row select + exact re-rank in canonical order + sort
void launch_knn_exact_rows(const double *Fw, int Dpad, int D, int64_t N, const double *Qp, const int *rows,
                           int n_rows, int K, double *scratch, int64_t scratch_pitch, const int32_t *unit_class,
                           const int32_t *query_class, int64_t id_offset, int64_t *cand, double *dist,
                           double *d2_out, hipStream_t s);
void launch_knn_local_kth(const int *cnt, const double *lkey, int cap, int K, const double *eps, int64_t T, double *kth,
                          hipStream_t s);
void launch_knn_list_prune(int *cnt, double *lkey, int *lidx, int cap, const double *bound, const double *eps, int64_t T,
                           hipStream_t s);
void launch_knn_finalize(const double *Fw, const float *F_unw, int Fp, const double *wt, int Dpad, int D, const double *Qp, const double *qnorm,
                         int64_t T, int K, const int *cnt, const double *lkey, const int *lidx,
                         int cap, int64_t id_offset, const double *eps, const double *fnorm, double eps_c, const double *cq,
                         int64_t *cand, double *dist, double *d2_out, int *status, int *rowflag, hipStream_t s,
                         bool split_short = false,    // split_short: rows of at most 512 entries through the small-LDS instance
                         const double *thr = nullptr, unsigned int *margin_stat = nullptr,    // tripwire of the prefilter's key bound
                         int *retry = nullptr,      // T ints of scratch: the lean form (keys only in LDS) first, the full form for the rows it flags
                         bool big_tier = false,     // ... and a third form (selection of 8 192) for the rows with more near ties than the full form holds
                         bool verify_thr = false,   // optimistic thresholds: rows whose list is not PROVEN to hold the K nearest set status bit 8
                         bool retry_cleared = false);   // `retry` was zeroed by the call's knn_reset already
void launch_candidate_dist(const double *Fw, int Dpad, int D, int64_t N, const double *Qp,
                           const int64_t *cand, int64_t T, int K, double *dist, hipStream_t s);
void launch_results_to_host(void *const *dst, const void *const *src, const size_t *bytes, int n, hipStream_t s);      // viterbi_kernels.hip
void launch_merge_topk(const double *d2, const int64_t *id, int G, int64_t T, int K,
                       int64_t *cand, double *dist, hipStream_t s);
// compacted exchange of the shards' lists (knn_kernels.hip)
void launch_shard_count(const int64_t *ids, int64_t R, int K, unsigned char *cnt, hipStream_t s);
void launch_shard_scan(const unsigned char *cnt, const int64_t *cnt_off, const int64_t *rows, int *off, const int64_t *off_off,
                       int64_t *tot, int G, hipStream_t s);
void launch_shard_pack(const double *d2, const int64_t *ids, const unsigned char *cnt, const int *off, const int64_t *row0,
                       const int64_t *rows, const int64_t *poff, const int64_t *tot, int G, int64_t R, int K, unsigned char *out,
                       hipStream_t s);
void launch_shard_unpack(const unsigned char *in, const int64_t *roff, const int64_t *totq, const int *offq, int64_t r_own, int K, int G,
                         double *d2_out, int64_t *id_out, hipStream_t s);

// ---- float32 prefilter (knn16_kernels.hip) ---------------------------------------------
// perm (nullable, here and below): the order the engine gave the prefilter's operands (kmeans_kernels.hip): position -> unit
void launch_build_db16(const double *Fw, const double *fnorm, int64_t N, int Dt, int Dpad, int64_t n_tiles,
                       int64_t sample_stride, int64_t G, int nt_a, void *A32, hipStream_t s, const int32_t *perm = nullptr);
// an order for the units of a voice whose tiles are not compact: k-means clusters laid out one after the other (kmeans_kernels.hip)
int kmeans_clusters(int64_t N, int Dt);
bool kmeans_supported(int Dt);
size_t kmeans_workspace_bytes(int64_t N, int Dt);
void launch_kmeans_order(const double *Fw, int64_t N, int Dt, int Dpad, int iters, void *workspace, int *perm, hipStream_t s);
size_t kmeans_perm_check_bytes(int64_t N);
void launch_perm_check(const int *perm, int64_t N, void *scratch, hipStream_t s);      // scratch word 0 <- entries that make perm no permutation
void launch_fmax(const double *fnorm, int64_t N, double *out, hipStream_t s);
void launch_prepare_queries16(const double *Qp, const double *qnorm, int64_t T, int Dt, int Dpad,
                              const double *fmax2, double eps_c, void *B32, double *eps, hipStream_t s);
void launch_build_class16(const int32_t *unit_class, int64_t N, int64_t n_tiles, int64_t sample_stride, int64_t G,
                          int nt_a, int32_t *out, hipStream_t s, const int32_t *perm = nullptr);
bool launch_knn_sweep16(int mode, int nt, int dch, int k_steps, int grid_cus, const void *A32, const void *B32,
                        const int32_t *tile_class, const int32_t *query_class,
                        const float *thr32, int64_t T32, int64_t n_slabs, unsigned int *ctr,
                        float *gmin32, int64_t G, void *pool, unsigned int *pool_ctl, int *chunk_fill,
                        int max_chunks, int pool_chunk, hipStream_t s);
void launch_knn_threshold16(const float *gmin32, int64_t G, int64_t T, int64_t T32, int K, const double *eps,
                            double *thr, float *thr32, const double *bound_in, double *bound_out, hipStream_t s,
                            const double *e1 = nullptr, float *thr1 = nullptr,     // thr1: the coarse pass's threshold (thr32 + e1)
                            const double *bound2 = nullptr);                       // a second upper bound per row (stage A'): the smaller one serves
// stage A': keys of the units of the tiles nearest to each query row (knn16_kernels.hip): a second bound per row
int knn_scout_groups(int64_t T32, int64_t n_tiles);
int knn_scout_keys_per_row();
size_t knn_scout_list_bytes(int64_t T32);
bool launch_knn_scout16b(int terms, int dch, int grid_cus, const void *C16, const void *A16, const void *B16, int64_t T, int64_t T32,
                         int64_t n_tiles, float *smin, unsigned int *list, float *gkeys, hipStream_t s);
// rows of 257 .. 512 columns: blocked bf16-split product (knn16_kernels.hip)
bool knn_wide16b_supported(int Dt, int Dpad);
void launch_knn_wide16b(int mode, int terms, int grid_cus, const void *A16, const void *B16, int Dpad, const float *thr32, int64_t T32,
                        int64_t n_tiles, float *gmin32, int64_t G, void *pool, unsigned int *pool_ctl, int *chunk_fill,
                        int max_chunks, int pool_chunk, hipStream_t s);
// bf16-split prefilter (knn16_kernels.hip)
bool knn_sweep16b_supported(int nt, int dch, int Dt, int Dpad, bool cls);
void launch_build_db16b(const double *Fw, const double *fnorm, int64_t N, int Dt, int Dpad, int64_t n_tiles,
                        int64_t sample_stride, int64_t G, int nt_a, void *A16, hipStream_t s, const int32_t *perm = nullptr);
void launch_prepare_queries16b(const double *Qp, const double *qnorm, int64_t T, int Dt, int Dpad, const double *fmax2,
                               const double *rho, double c_acc, void *B16, double *eps, double *cq, hipStream_t s,
                               double c_coarse = 0.0, double *e1 = nullptr);      // e1: what the hi.hi term alone may be off by
// two-pass filter on the bf16-split operands: hi.hi sweep -> (database tile, query tile) pairs -> three-term keys of those
bool knn_coarse16b_supported(int nt, int dch);
size_t knn_coarse_pair_bytes();
bool launch_knn_filter16c(int terms, int dch, int grid_cus, const void *A16, const void *B16, const float *thr32, const float *thr1,
                          int64_t T32, int64_t n_tiles, unsigned int *ctr, void *pairs, unsigned int *pair_ctl, unsigned int pair_cap,
                          void *pool, unsigned int *pool_ctl, int *chunk_fill, int max_chunks, int pool_chunk, hipStream_t s,
                          bool run_coarse = true,      // false: the pair list is already there (the ball pass wrote it)
                          bool run_refine = true);     // false: the first pass alone (a probe that counts pairs: pair_cap 0)
// pass 0: the tiles' balls (centre, radius) against the query rows; lists pairs, then gates the coarse sweep
void launch_build_tile_balls(const double *Fw, int64_t N, int Dt, int Dpad, int64_t n_tiles, double *C, double *cnorm, float *rad,
                             hipStream_t s, const int32_t *perm = nullptr);
void launch_ball_query_terms(const float *thr32, const double *eps, const double *qnorm, int64_t T, int64_t T32, float *tq, float *nq,
                             hipStream_t s);
bool launch_knn_balls16b(int terms, int dch, int grid_cus, const void *C16, const void *B16, const float *rad, const float *tq,
                         const float *nq, int64_t T32, int64_t n_tiles, void *pairs, unsigned int *pair_ctl, unsigned int pair_cap,
                         hipStream_t s, unsigned int *mask_out = nullptr, const unsigned int *visit = nullptr);
// the balls of 32 consecutive tiles: the ball pass's own first level (mask_out / visit above)
void launch_build_super_balls(const double *C, const float *rad, int64_t N, int64_t n_tiles, int Dt, int Dpad, int64_t n_super, double *C2,
                              double *cnorm2, float *rad2, hipStream_t s);
void launch_db16b_ratios(const double *Fw, int64_t N, int Dt, int Dpad, double *rho, hipStream_t s, bool accumulate = false);   // accumulate: max with what rho holds
bool launch_knn_sweep16b(int mode, int terms, int nt, int dch, int grid_cus, const void *A16, const void *B16, const float *thr32,
                         int64_t T32, int64_t n_slabs, unsigned int *ctr, float *gmin32, int64_t G, void *pool,
                         unsigned int *pool_ctl, int *chunk_fill, int max_chunks, int pool_chunk, hipStream_t s);
void launch_mfma_bf16_probe(const unsigned short *A, const unsigned short *B, const float *C, float *D, hipStream_t s);
void launch_mfma16_selftest(const float *A, const float *B, float *C, hipStream_t s);
int knn_pool_chunk_entries();
void sweep_tail_split(int64_t n_slabs, int qsplit, int64_t waves, int nQT, int64_t *n_main, int *qtail);

// ---- join costs + Viterbi -------------------------------------------------
void launch_join_costs(const double *JCw, int Djpad, int Dj, int64_t n_units,
                       const int64_t *cand, int64_t T, int K, double *J, hipStream_t s);
struct DpBatch {                      // kernel-argument descriptor of one batched recursion launch
    static constexpr int MAX = 24;
    int64_t off[MAX + 1];
    int first;
};
void launch_viterbi_dp_batch(const int64_t *cand, const double *tdist, const double *J, const int64_t *off,
                             int n_utts, int first_utt, int K, int64_t n_units, unsigned char *bp_global,
                             int64_t *path, int64_t *path_len, double *cost, hipStream_t s,
                             bool fst32 = false);    // fst32: OpenFST's float32 weight chain (option viterbi_weights 1)
void launch_viterbi_dp(const int64_t *cand, const double *tdist, const double *J,
                       int64_t T, int K, int64_t n_units, unsigned char *bp_global,
                       int64_t *path, int64_t *path_len, double *cost, hipStream_t s, bool fst32 = false);

// ---- join lower bounds + sparse exact recursion (joinfast_kernels.hip) ------------------------
bool join_lb_supported(int Dj, int K);
void launch_join_lb(const float *JC_unw, int Jp, int Dj, const double *wj, int64_t n_units, const int64_t *cand,
                    int64_t R, int K, float *Jlo, float *scale, hipStream_t s);
// second form of pass 1: bf16 matrix pipe over a float32 copy of the weighted join rows (built once per set of weights)
int join_lb2_pitch(int Dj);
double join_lb2_ceps(int Dj, int K);
bool join_lb2_supported(int Dj, int K);
void launch_join_weight32(const float *JC_unw, int Jp, int64_t Njc, int Dj, const double *wj, float *JW, int Jq,
                          unsigned int *umax_bits, hipStream_t s);
void launch_join_lb2(const float *JW, int Dj, const unsigned int *umax_bits, int64_t n_units, const int64_t *cand, int64_t R,
                     int K, float *Jlo, float *scale, hipStream_t s);
void launch_viterbi_lb(const int64_t *cand, const double *tdist, const float *Jlo, const float *scale, const int64_t *off,
                       int n_utts, int K, int64_t n_units, float beta, void *sets, hipStream_t s,
                       int chunk_len = 0, int warm = 32,
                       float slack32 = 0.f);        // viterbi_weights 1: sets wider by this fraction of the (estimated) absolute total   // chunk_len > 0: the approximate recursion in chunks of that many steps, side by side
size_t join_record_bytes();
// Jlo / scale / ceps / stats (pass 3 and pass 4): the tripwire of pass 1's bounds -- every exact cost computed is held against the
// bound of its cell: stats[4] += cells with lo > exact, stats[5] = smallest margin (joinfast_kernels.hip); ceps from join_lb_ceps
double join_lb_ceps(int variant, int Dj, int K);
void launch_scale_f32(float *x, int64_t n, float f, hipStream_t s);      // test hook (option join_lb_test_scale)
void launch_join_exact_sparse(const float *JC_unw, int Jp, int Dj, const double *wj, int64_t n_units, const int64_t *cand,
                              const double *tdist, int64_t R, int K, const void *sets, void *rec, hipStream_t s,
                              const float *Jlo = nullptr, const float *scale = nullptr, float ceps = 0.f, unsigned long long *stats = nullptr);
void set_join_lb_quadrants(int q);         // 1: pass 1 of K > 128 as 2 x 2 quadrants (joinlb2_kernels.hip); 0 (default): one workgroup per row pair
int get_join_lb_quadrants();
void set_join_lb_one_set(int v);          // 1: pass 1 of K <= 128 on ONE accumulator set, one k-block in flight: half the registers, twice the workgroups per compute unit, a looser error constant
int get_join_lb_one_set();
void set_join_exact_form(int f);           // 1 (default): cooperative pass 3 (rows in coalesced chunks through LDS); 0: a lane per cell
void set_viterbi_sparse_waves(int w);      // 1 (default) or 4: which form of pass 4 launch_viterbi_sparse runs (same results)
void launch_viterbi_sparse(const int64_t *cand, const void *rec, const float *Jlo, const float *JC_unw, int Jp, int Dj,
                           const double *wj, const int64_t *off, int n_utts, int first_utt, int K, int64_t n_units,
                           unsigned char *bp_global, int64_t *path, int64_t *path_len, double *cost,
                           unsigned long long *stats, hipStream_t s, const float *scale, float ceps,
                           bool fst32 = false);    // fst32: OpenFST's float32 weight chain (option viterbi_weights 1; the one-wavefront form)

// ---- greedy ---------------------------------------------------------------
struct GreedyLayout {
    int me, last_frame_as_target, join_split_mode;
    int64_t Nwin;         // N - me + 1
    int jdim;             // columns of prev/current join rep
    int prev_col0, cur_col0;      // first column inside a JCw row
    int64_t prev_row0, cur_row0;  // JCw row of window 0 for prev / current
};
// Fp / Jp: row pitch (floats, multiple of 4) of the unweighted device matrices
void launch_greedy(const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const double *wt,
                   const float *JC_unw, int Jp, int Dj, const double *wj, const float *tiles, const double *Q,
                   int64_t nsteps, int64_t start_state, double *tables, double *blk_min,
                   int64_t *blk_arg, int nblk, int n_cus, unsigned int *arrive, int64_t *path, double *dist, hipStream_t s);
size_t greedy_table_doubles(const GreedyLayout &g, int Dt, int ub = 1);
int greedy_max_utts(const GreedyLayout &g, int Dt);
void launch_greedy_batch(const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const double *wt,
                         const float *JC_unw, int Jp, int Dj, const double *wj, const float *tiles, const double *Q,
                         int nu, const int64_t *q_off, const int64_t *nsteps_u, const int64_t *out_off,
                         const int64_t *start, double *tables, double *blk_min,
                         int64_t *blk_arg, int nblk, int n_cus, unsigned int *arrive, int64_t *path, double *dist, hipStream_t s);
size_t greedy_counter_bytes();
// snk_sharded_greedy (greedy_kernels.hip): a rank's share of a step's scan, then the pick among the ranks' winners
int greedy_shard_blocks(const GreedyLayout &g, int Dt, int n_cus, int64_t tile_n);
void launch_greedy_shard_init(const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const double *wt, const float *JC_unw, int Jp,
                              int Dj, const double *wj, const float *tiles, const double *Q, int64_t nsteps, int64_t start_state,
                              double *tables, unsigned int *arrive, hipStream_t s);
void launch_greedy_shard_step(const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const double *wt, const float *JC_unw, int Jp,
                              int Dj, const double *wj, const float *tiles, const double *Q, int64_t step, int64_t nsteps,
                              int64_t tile_lo, int64_t tile_n, double *tables, double *blk_min, int64_t *blk_arg, int nblk, int n_cus,
                              unsigned int *arrive, double *shard_out, hipStream_t s);
void launch_greedy_shard_pick(const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const double *wt, const float *JC_unw, int Jp,
                              int Dj, const double *wj, const double *Q, int64_t step, int64_t nsteps, const double *gathered, int G,
                              double *tables, int64_t *path, double *dist, hipStream_t s);
// lane-major copy of the scan columns (built once per database + layout, read by the scan)
size_t greedy_tile_bytes(const GreedyLayout &g, int Dt);
void launch_greedy_tiles(const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const float *JC_unw, int Jp,
                         float *tiles, hipStream_t s);
int greedy_blocks(const GreedyLayout &g, int Dt, int n_cus, int ub = 1);
// float16 copy of the join tiles and the norms its bound needs (the float32 scan with the hoisted target term)
size_t greedy_tile16_bytes(const GreedyLayout &g);
void launch_greedy_tiles16(const GreedyLayout &g, const float *JC_unw, int Jp, void *tiles16, unsigned int *max_abs_bits, hipStream_t s);
void launch_greedy_join_norms(const GreedyLayout &g, const float *tiles, const double *wj, unsigned long long *out, hipStream_t s);
bool greedy_supported(const GreedyLayout &g, int Dt);

// float32 prefilter scan, one persistent launch per utterance group (greedy32_kernels.hip)
bool greedy32_supported(const GreedyLayout &g, int Dt);
int greedy32_max_utts(bool hoist = false);      // utterances per scan: 3, 6 with the hoisted target term
int greedy32_blocks(const GreedyLayout &g, int Dt, int n_cus, bool hoist = false);
size_t greedy32_table_floats(const GreedyLayout &g, int Dt, bool hoist = false);
// hoisted target term (greedy_hoist_kernels.hip): what the scan reads instead of the target columns
struct G32Hoist {
    const float *W[6];        // per utterance: (nsteps x Wp) float32 target terms
    const double *qn2[6];     // per utterance: ||target vector of the step||^2
    int64_t Wp;
    double c, fwmax2;         // bound |W~ - W| <= 2^-24 W + c (||q|| + sqrt(fwmax2))^2
    const void *JT16;         // float16 join tiles (nullptr: none) and the bound's term 2^-11 max ||w o S'|| + 2^-25 ||w||
    double f16_delta; int f16_force;
};
int64_t greedy_hoist_pitch(const GreedyLayout &g);
int greedy_hoist_k(const GreedyLayout &g, int Dt);
bool greedy_hoist_supported(const GreedyLayout &g, int Dt);
double greedy_hoist_c(const GreedyLayout &g, int Dt);
bool greedy_hoist16_supported(const GreedyLayout &g, int Dt);      // the product on the bf16 pipe (scans of float16 join tiles)
double greedy_hoist_c16(const GreedyLayout &g, int Dt);
int64_t greedy_hoist_rows(int64_t nsteps);
void launch_hoist_window_norms(const GreedyLayout &g, const double *fnorm, double *nw, unsigned long long *max_bits, hipStream_t s);
void launch_hoist_product(const GreedyLayout &g, const float *F_unw, int Fp, int64_t n_f_rows, int Dt, const double *wt, const double *Q,
                          int64_t q_off, int64_t nsteps, const double *nw, double *Aq, double *qn2, float *W, hipStream_t s);
void launch_hoist_product16(const GreedyLayout &g, const float *F_unw, int Fp, int64_t n_f_rows, int Dt, const double *wt, const double *Q,
                          int64_t q_off, int64_t nsteps, const double *nw, double *Aq, double *qn2, float *W, hipStream_t s);
size_t greedy32_block_bytes(int nblk);
void greedy32_trace_dump();          // developer aid (SNK_G32_TRACE=file): timeline of the last launch
void launch_greedy32(const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const double *wt, const float *JC_unw, int Jp,
                     int Dj, const double *wj, const float *tiles, const double *Q, int nu, const int64_t *q_off,
                     const int64_t *nsteps_u, const int64_t *out_off, const int64_t *start, int approx, void *blk, int n_cus, unsigned int *gen, int64_t *status,
                     int64_t *path, const G32Hoist *hoist, hipStream_t s);
// database resident in LDS for the whole launch (greedy_res_kernels.hip): one utterance, hoisted target term required
bool greedy_res_supported(const GreedyLayout &g, int Dt, int n_cus);
size_t greedy_res_record_bytes(const GreedyLayout &g);
void launch_greedy_res(const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const double *wt, const float *JC_unw, int Jp,
                       int Dj, const double *wj, const float *tiles, const double *Q, int64_t q_off, int64_t nsteps,
                       int64_t out_off, int64_t start, int flags, void *rec, int64_t *status, int64_t *path,
                       const G32Hoist *hoist, hipStream_t s);
void greedy_res_trace_dump();
void launch_greedy32_dist(const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const double *wt, const float *JC_unw, int Jp,
                          int Dj, const double *wj, const double *Q, int u_slot, int64_t q_off, int64_t nsteps, int64_t out_off,
                          int64_t start, const int64_t *path, double *dist, hipStream_t s);

void launch_path_scores(const GreedyLayout &g, int mode, const float *F_unw, int Fp, int Dt, const double *wt,
                        const float *JC_unw, int Jp, int Dj, const double *wj, const double *Q,
                        const int64_t *path, int64_t L, double *tsq, double *jsq, int jcols,
                        hipStream_t s);

// ---- waveform-side gather ----------------------------------------------------
void launch_concat_fragments(const float *spec, int W, const double *fzv, const int64_t *first_row,
                             const int64_t *utt_lo, const int64_t *utt_hi, int64_t n, int me, int ov,
                             const double *in_taper, double *out_spec, double *out_fz, hipStream_t s);

// ---- self test --------------------------------------------------------------
void launch_mfma_selftest(const double *A, const double *B, double *C, hipStream_t s);

}  // namespace snk
