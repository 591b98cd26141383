/*
 * snk.h -- C ABI of libsnkhip.so: the MI355X (gfx950) unit-selection search
 * engine that replaces the CPU hot path of CSTR-Edinburgh/snickery
 * (script/synth_simple.py, script/synth_halfphone.py).
 *
 * The reference has no FFI for this path: the search sits behind plain Python
 * methods of `Synthesiser`.  Each entry point below names the reference
 * method / third-party call it replaces (file:line in the reference checkout).
 * INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; the message is
 *     available from snk_last_error() (thread-local).
 *   - all pointers are caller-owned HOST memory unless the name ends in _dev.
 *   - matrices are dense row-major.
 *   - one handle = one device + one HIP stream; a handle is not thread-safe,
 *     different handles are independent (the reference's only concurrency is
 *     process-level, synth_halfphone.py:897-903).
 *   - there is NO CPU fallback: every compute entry point fails if no gfx950
 *     device is usable.
 */
#ifndef SNK_H
#define SNK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct snk_engine *snk_handle;

/* library */
int         snk_abi_version(void);
const char *snk_last_error(void);
int         snk_device_count(int *count_out);

/* lifetime */
int snk_create(int device_id, snk_handle *handle_out);
int snk_destroy(snk_handle h);

/* Replaces the HDF5 arrays the Synthesiser keeps after loading
 * (synth_simple.py:87-97, synth_halfphone.py:184-225):
 *   F_unw  (N, Dt)    float32  train_unit_features (unweighted)
 *   JC_unw (Njc, Dj)  float32  join_contexts (unweighted), Njc = N+1
 * Device copies are made; the host arrays may be freed afterwards. */
int snk_upload_db(snk_handle h, const float *F_unw, int64_t N, int Dt,
                  const float *JC_unw, int64_t Njc, int Dj);

/* Replaces set_target_weights / set_join_weights (synth_simple.py:234-274,
 * synth_halfphone.py:682-737) + weight() (speech_manip.py:209-213):
 * per-COLUMN weight vectors, F = F_unw * wt (float64), JC = JC_unw * wj,
 * unit_end_data = JC[1:], unit_start_data = JC[:-1].  O(N*D) on device, no index
 * rebuild (the reference rebuilds its KD-trees here). */
int snk_set_weights(snk_handle h, const double *wt, int n_wt, const double *wj, int n_wj);

/* Replaces truncate_target_streams / truncate_join_streams (synth_simple.py:982-992,
 * get_selection_vector :968-980; synth_halfphone.py:756-790): tcols / jcols list the columns that
 * take part (ascending; n < 0: all).  The reference drops the other columns from its weighted copies
 * and from the query rows; here they stay in place with weight 0 and zeroed query entries, which
 * adds exactly +0.0 per column -- identical candidates, distances and paths.  Takes effect with the
 * next snk_set_weights; query matrices keep their full width. */
int snk_set_column_selection(snk_handle h, const int *tcols, int nt, const int *jcols, int nj);

/* Replaces `self.tree.query(unit_features, k=n_candidates)` in
 * preselect_units_acoustic (synth_halfphone.py:1359-1366; tree built at :379):
 *   Q (T, D) float64 weighted target vectors, D == Dt
 *   cand_out (T, K) int64 unit ids ascending by distance (ties: lower id first)
 *   dist_out (T, K) float64 Euclidean distances (not squared)
 * K > N pads with id -1 / distance 1e15 (const.VERY_BIG_WEIGHT_VALUE). */
int snk_knn(snk_handle h, const double *Q, int64_t T, int D, int K,
            int64_t *cand_out, double *dist_out);

/* Replaces preselect_units_monophone_then_acoustic (synth_halfphone.py:1369-1396,
 * per-phone trees :385-402): K-NN restricted to DB units whose class id equals the
 * query's; short classes padded with -1 / 1e15.
 *   unit_class (N) int32 set once with snk_set_unit_classes; query_class (T) int32 */
int snk_set_unit_classes(snk_handle h, const int32_t *unit_class, int64_t N);
int snk_knn_by_class(snk_handle h, const double *Q, int64_t T, int D, int K,
                     const int32_t *query_class, int64_t *cand_out, double *dist_out);

/* Diagnostic of the K-NN prefilter (no reference counterpart: the reference's KD-tree is exact by construction,
 * synth_halfphone.py:379): the approximate key ||f||^2 - 2 q.f the prefilter computes is trusted to eps[t]; this
 * returns the prefilter's MINIMUM key per (query row, slab of *rows_per_slab consecutive units) and eps, so a test
 * can hold them against float64 keys.  slab_min == NULL: only *n_slabs / *rows_per_slab are written.
 *   slab_min (T, n_slabs) float32, eps (T) float64 */
int snk_prefilter_minima(snk_handle h, const double *Q, int64_t T, int D, float *slab_min, int64_t slab_min_len,
                         double *eps, int64_t *n_slabs, int *rows_per_slab);

/* Distance part of preselect_units_quinphone (synth_halfphone.py:1343-1349): the candidate ids come
 * from the label index on the host (quinphone -> triphone -> diphone -> monophone back-off,
 * :1315-1334); dist_out[t,k] = ||F[cand[t,k]] - Q[t]||_2.  As in the reference's numpy fancy
 * indexing a negative id counts from the end of the database (padding -1 -> last unit). */
int snk_candidate_distances(snk_handle h, const double *Q, int64_t T, int D, const int64_t *cand, int K,
                            double *dist_out);

/* Replaces make_on_the_fly_join_lattice_BLOCK_DIRECT's cost computation
 * (synth_halfphone.py:3206-3322; get_natural_distance_vectorised :2942-2951):
 *   J_out (T-1, K, K) float64, J[t,a,b] = ||unit_end[cand[t,a]] - unit_start[cand[t+1,b]]||_2,
 *   +inf where either unit is unusable (id == -1, id < 1 or id >= N-1, :3238-3268). */
int snk_join_costs(snk_handle h, const int64_t *cand, int64_t T, int K, double *J_out);

/* Diagnostic of the sparse Viterbi path (no reference counterpart: the reference computes every join cost exactly,
 * synth_halfphone.py:2942-2951): the float32 LOWER BOUNDS of the join costs that pass 1 computes on the matrix pipe
 * (option join_lb_variant: 1 = bf16 pieces of a weighted float32 copy, 0 = float32 operands weighted per gather), so a
 * test can hold lo[t,a,b] <= J[t,a,b] of snk_join_costs for every cell and measure how tight they are.
 *   lo_out (T-1, K, K) float32 (+inf where either unit is unusable), scale_out (T-1) float32: the step's largest
 *   centred norm (the unit of the pass-2 margin join_beta). */
int snk_join_bounds(snk_handle h, const int64_t *cand, int64_t T, int K, float *lo_out, float *scale_out);

/* Replaces viterbi_search (synth_halfphone.py:1399-1436): target sausage lattice
 * (fst_functions_wrapped.py:28-58) o join lattice (:172-217), openfst.compose (:368)
 * and openfst.shortestpath (:389), as one dynamic programme over the T x K trellis.
 *   cand (T,K) int64, tdist (T,K) float64 -> path_out (T) int64 unit ids,
 *   *path_len_out = T, or 0 when no complete path exists (incl. T < 2),
 *   *cost_out = accumulated path cost (float64). */
int snk_viterbi(snk_handle h, const int64_t *cand, const double *tdist, int64_t T, int K,
                int64_t *path_out, int64_t *path_len_out, double *cost_out);

/* snk_viterbi for a batch of utterances whose candidates the caller already has (label-driven preselection:
 * preselect_units_quinphone / _monophone_then_acoustic, synth_halfphone.py:1315-1396, then viterbi_search per utterance
 * in the reference's tuning loop): cand / tdist (row_offsets[n_utts], K) row-concatenated, outputs as snk_knn_viterbi_batch. */
int snk_viterbi_batch(snk_handle h, const int64_t *cand, const double *tdist, const int64_t *row_offsets, int n_utts, int K,
                      int64_t *path_out, int64_t *path_len_out, double *cost_out);

/* snk_knn + snk_viterbi without the host round trip between them (what synth_utt does
 * at synth_halfphone.py:1601-1625).  cand_out / dist_out may be NULL. */
int snk_knn_viterbi(snk_handle h, const double *Q, int64_t T, int D, int K,
                    int64_t *cand_out, double *dist_out,
                    int64_t *path_out, int64_t *path_len_out, double *cost_out);

/* Batched form of snk_knn_viterbi for throughput callers (balance_stream_weights.py:82-92
 * loops synth_utt over a tune set): n_utts utterances, rows concatenated in Q,
 * row_offsets (n_utts+1).  path_out has row_offsets[n_utts] entries; path_len_out and
 * cost_out have n_utts entries.  Stages of consecutive utterances overlap on device. */
int snk_knn_viterbi_batch(snk_handle h, const double *Q, const int64_t *row_offsets,
                          int n_utts, int D, int K,
                          int64_t *path_out, int64_t *path_len_out, double *cost_out);
/* optional: page-lock a host buffer that is uploaded repeatedly (queued instead of blocking copies) */
int snk_host_register(void *ptr, size_t bytes);
int snk_host_unregister(void *ptr);
/* The same in two halves: submit() queues the whole batch and returns, collect() waits for it.  Three
 * batches may be in flight; submitting batch i+1 before collecting batch i hides the tail of batch i
 * (last recursions, copy to the host) behind the K-NN of batch i+1 -- the shape of a tuning loop
 * over a large tune set (balance_stream_weights.py:82-92 searches the same utterances every iteration) --
 * and submitting batch i+2 as well keeps the K-NN stream fed while the host waits for batch i (with two
 * in flight the next submit can only follow that wait: the stream runs dry whenever the host is slower
 * than the 0.3 ms of slack a B* step leaves).
 * Q must stay valid until submit() returns; tickets are 0 .. 2.
 * Q == NULL: the query rows this ticket's workspace received with its previous submit are still on the device and are
 * searched again (same row_offsets and D required): a caller whose targets do not change between two searches -- a
 * tuning loop that only moves join weights, a benchmark with its inputs resident in HBM -- uploads them once per
 * workspace (three workspaces take turns: the first three submits carry Q). */
int snk_knn_viterbi_batch_submit(snk_handle h, const double *Q, const int64_t *row_offsets, int n_utts,
                                 int D, int K, int *ticket_out);
int snk_knn_viterbi_batch_collect(snk_handle h, int ticket, int64_t *path_out, int64_t *path_len_out,
                                  double *cost_out);

/* Replaces get_tree_for_greedy_search's data layout (synth_simple.py:190-225,
 * synth_halfphone.py:539-596).  Nothing is materialised: the windowed DB
 * (N-me+1, me*Dt) is addressed in place.
 *   join_split_mode 0: synth_simple (prev = unit_start_data, current = unit_end_data)
 *   join_split_mode 1: synth_halfphone epoch DB (first/second half of the columns) */
int snk_set_greedy_layout(snk_handle h, int multiepoch, int last_frame_as_target,
                          int join_split_mode);

/* Replaces greedy_joint_search (synth_simple.py:458-503 == synth_halfphone.py:1900-1945),
 * i.e. the per-step joint_tree.query(k=1) at :488-490.
 *   Q (T, Dt) float64 weighted targets (NOT yet reshaped by multiepoch)
 *   path_out  (T / multiepoch) int64 indices into the windowed DB
 *   dist_out  (T / multiepoch) float64 Euclidean distance of each pick (may be NULL)
 *   eps: search_epsilon (`joint_tree.query(..., eps=...)`, :488-490; shipped as 10.0 in
 *        config/slt_simplified_mini.cfg:92).  0: the exact nearest neighbour, lowest index on exact ties.
 *        >= 1e-3 on the float32 scan (the default wherever the hoisted target term applies -- join streams of
 *        65 columns and more --, any batch, option greedy_mode 1): the window with the smallest FLOAT32 total is
 *        returned without re-evaluation where its error bound is below 1e-3 of the nearest distance (inside the
 *        (1 + eps) contract) and decided exactly otherwise; dist_out still holds the pick's exact distance.
 *        The exact scan (greedy_mode 0) returns the exact neighbour for every eps.
 *   The target term of all steps is computed up front as one float64 matrix product per utterance
 *   (greedy_hoist_kernels.hip; T / multiepoch x N floats of device memory, option greedy_hoist_max_gb). */
int snk_greedy(snk_handle h, const double *Q, int64_t T, int D, int64_t start_state,
               double eps, int64_t *path_out, double *dist_out, int64_t *nsteps_out);
/* snk_greedy for several utterances in one call (balance_stream_weights.py:82-92 runs the greedy search
 * over a tune set, iteration after iteration).  Up to six utterances share every scan of the database (three
 * where the hoisted target term does not apply): the weighted value of a column is computed once per window
 * and compared with each utterance's reference, and the database is read once per step for all of them.  Rows concatenated in Q,
 * row_offsets (n_utts+1); start_states (n_utts) may be NULL (= -1 each); path_out / dist_out hold the
 * utterances' steps one after the other (nsteps_out[u] = rows_u / multiepoch each).  Same results as
 * n_utts calls of snk_greedy. */
int snk_greedy_batch(snk_handle h, const double *Q, const int64_t *row_offsets, int n_utts, int D,
                     const int64_t *start_states, double eps, int64_t *path_out, double *dist_out,
                     int64_t *nsteps_out);

/* Replaces get_target_scores_per_stream / get_join_scores_per_stream
 * (synth_halfphone.py:1964-1981): squared errors along a path, per column.
 *   mode 0: Viterbi  (E[p[:-1]] - S[p[1:]])^2 ; mode 1: greedy (prev[p[1:]] - cur[p[:-1]])^2
 *   tsq_out (L, Dt), jsq_out (L-1, Dj') float64 */
int snk_path_scores(snk_handle h, const double *Q, const int64_t *path, int64_t L, int mode,
                    double *tsq_out, double *jsq_out);

/* Per-stage device times of the most recent call, measured with HIP events on the
 * engine's stream (names via snk_timer_name): the reference's start_clock/stop_clock
 * stage log (synth_halfphone.py:1953-1961). Returns number of timers written. */
int         snk_get_timers(snk_handle h, double *ms_out, int capacity);
const char *snk_timer_name(int index);
int         snk_timer_count(void);
int         snk_reset_timers(snk_handle h);

/* Multi-GPU (database row-sharded over ranks).  The shard holds rows
 * [row_offset, row_offset+N) of the global DB: candidate ids reported by the *_dev calls
 * are global ids.  Device-pointer entry points so that the exchange step
 * (RCCL all-gather of the per-rank top-K) can run on device buffers owned by the caller. */
int snk_set_shard(snk_handle h, int64_t global_row_offset, int64_t global_N);
int snk_knn_local_dev(snk_handle h, const double *Q, int64_t T, int D, int K,
                      double *d2_dev_out /* (T,K) squared distances */,
                      int64_t *id_dev_out /* (T,K) global ids, -1 padded */);
/* merge G gathered lists (G,T,K) -> (T,K), ordered by (distance, id); host outputs */
int snk_merge_topk_dev(snk_handle h, const double *d2_dev, const int64_t *id_dev,
                       int G, int64_t T, int K, int64_t *cand_out, double *dist_out);
/* Batch form of the two calls above for throughput callers (the sharded counterpart of
 * snk_knn_viterbi_batch): rows of n_utts utterances concatenated in Q (host), row_offsets (n_utts+1).
 *   step 1, every rank:   snk_knn_local_batch_dev -> (R, K) lists in caller device buffers, R = all rows
 *   exchange (caller):    RCCL all-to-all, rank r receives the rows of the utterances it owns from
 *                         every shard -> (G, R_own, K)
 *   step 2, owner rank:   snk_merge_viterbi_batch_dev -> merged candidates, join costs, Viterbi;
 *                         paths / lengths / costs of the owned utterances on the host. */
int snk_knn_local_batch_dev(snk_handle h, const double *Q, const int64_t *row_offsets, int n_utts,
                            int D, int K, double *d2_dev_out, int64_t *id_dev_out);
/* Step 1 with shared bounds (fewer survivors per shard, the more the more shards there are):
 *   snk_knn_local_batch_bounds_dev  -> per row, an upper bound of the K-th nearest key of THIS shard
 *                                      (stage A only; DBL_MAX where no bound is available), device buffer;
 *   caller:                            all-reduce MIN of the (R,) bounds over the shards -- the smallest
 *                                      still bounds the K-th nearest key of the whole database;
 *   snk_knn_local_batch_bounded_dev -> as snk_knn_local_batch_dev, filtering against those bounds; a
 *                                      shard's list may hold fewer than K entries (id -1 padding, as for
 *                                      shards smaller than K).  Q == NULL: the query rows of the bounds
 *                                      call are still resident.
 * Replaces nothing in the reference (SURVEY 8e sketches the plain all-gather); results are identical. */
int snk_knn_local_batch_bounds_dev(snk_handle h, const double *Q, const int64_t *row_offsets, int n_utts,
                                   int D, int K, double *bound_dev_out /* (R,) */);
int snk_knn_local_batch_bounded_dev(snk_handle h, const double *Q /* nullable */, const int64_t *row_offsets,
                                    int n_utts, int D, int K, const double *bound_dev_in /* (R,) */,
                                    double *d2_dev_out, int64_t *id_dev_out);
int snk_merge_viterbi_batch_dev(snk_handle h, const double *d2_dev, const int64_t *id_dev, int G,
                                const int64_t *row_offsets, int n_utts, int K,
                                int64_t *path_out, int64_t *path_len_out, double *cost_out);
/* upload the FULL join matrix only (ranks that run the Viterbi of an utterance) */
int snk_upload_join_only(snk_handle h, const float *JC_unw, int64_t Njc, int Dj);

/* ---- collectives inside the library (one process per GPU; RCCL over xGMI) ------------------------
 * The reference's only fan-out is a multiprocessing.Pool over utterances (synth_halfphone.py:897-903); the
 * sharded search of SURVEY 8e needs an exchange step, which a C caller gets here without torch:
 *   rank 0: snk_comm_unique_id -> hand the 128 bytes to every rank (file, MPI, socket ...) ->
 *   every rank: snk_comm_init(h, nranks, rank, id)  [ncclCommInitRank; librccl is loaded at this call]
 *   every rank: snk_sharded_knn_viterbi_batch(...) with the SAME batch.
 * snk_sharded_knn_viterbi_batch, all on the engine's stream with no host synchronisation between its
 * stages: bounds of the K-th nearest key (own share of the rows against the replicated global sample when
 * snk_upload_global_sample was called, else all rows against the shard's sample) -> all-reduce MIN ->
 * filter + re-rank of ALL rows against this rank's shard -> all-to-all of the (R, K) lists to the ranks
 * that own the utterances (contiguous blocks, snk_shard_plan) -> merge, join costs, Viterbi of the owned
 * utterances -> all-gather of paths and costs.  Every rank returns the results of ALL utterances.
 * snk_transport: the same flow over caller-provided collectives on device buffers (the functional tests run
 * two ranks on ONE GPU, which RCCL refuses); each function returns 0 on success and must be complete (or
 * ordered on `stream`) when it returns. */
typedef struct snk_transport {
    void *ctx;
    int (*all_reduce_min_f64)(void *ctx, double *buf_dev, int64_t n, void *stream);
    int (*all_gather)(void *ctx, const void *send_dev, void *recv_dev, int64_t bytes_per_rank, void *stream);
    /* rank p's block: send_bytes[p] from send_dev + send_off[p]; recv_bytes[p] to recv_dev + recv_off[p].  Nothing but the
     * received blocks may be written: send_dev and recv_dev can be the same buffer with disjoint blocks (query rows). */
    int (*all_to_all_v)(void *ctx, const void *send_dev, const int64_t *send_off, const int64_t *send_bytes,
                        void *recv_dev, const int64_t *recv_off, const int64_t *recv_bytes, void *stream);
} snk_transport;
int snk_comm_unique_id(void *id_out, int capacity, int *bytes_out);
int snk_comm_init(snk_handle h, int nranks, int rank, const void *unique_id);
int snk_comm_init_transport(snk_handle h, int nranks, int rank, const snk_transport *transport);
int snk_comm_destroy(snk_handle h);
/* contiguous block [lo, hi) of n items that rank `rank` of `nranks` owns (sizes differ by at most one):
 * database rows of a shard, utterances of a batch */
int snk_shard_plan(int64_t n_items, int nranks, int rank, int64_t *lo_out, int64_t *hi_out);
/* every s-th unit of the WHOLE database, replicated on every rank (see above); before snk_set_weights */
int snk_upload_global_sample(snk_handle h, const float *F_sample_unw, int64_t n_rows, int Dt);
int snk_sharded_knn_viterbi_batch(snk_handle h, const double *Q, const int64_t *row_offsets, int n_utts, int D, int K,
                                  int64_t *path_out, int64_t *path_len_out, double *cost_out);
/* The same step in two halves (as snk_knn_viterbi_batch_submit / _collect on one GPU): submit queues the K-NN of all rows,
 * the collectives and the recursions of the owned utterances and returns a ticket (0 / 1); collect waits for them,
 * gathers every rank's results and hands them out.  Two steps may be in flight: submitting step i + 1 before collecting
 * step i runs the Viterbi side of step i beside the K-NN of step i + 1.  Every rank must issue the same sequence of
 * submits and collects (the collectives are matched by order); Q must stay valid until its step is collected. */
int snk_sharded_knn_viterbi_batch_submit(snk_handle h, const double *Q, const int64_t *row_offsets, int n_utts, int D, int K,
                                         int *ticket_out);
int snk_sharded_knn_viterbi_batch_collect(snk_handle h, int ticket, int64_t *path_out, int64_t *path_len_out, double *cost_out);
/* greedy_joint_search (synth_simple.py:458-503, the loop :486-501) with the scan of every step split over the ranks of
 * the communicator (snk_comm_init / snk_comm_init_transport): every rank holds the WHOLE database (snk_upload_db,
 * snk_set_weights, snk_set_greedy_layout as for snk_greedy -- the scan is bound by the bytes a step streams, not by
 * memory: 12 M units are 15 GB of the 288) and scans the windows of its share; per step one all-gather of 16 bytes per
 * rank (squared distance, window), every rank picks the same winner (lowest window on exact ties, as one scan does) and
 * writes the next step's reference itself.  The exact float64 scan, one launch per step and rank.  Every rank calls it
 * with the same arguments and receives the whole path; results equal snk_greedy's bit for bit.  No reference
 * counterpart (SURVEY 8e marks it optional). */
int snk_sharded_greedy(snk_handle h, const double *Q, int64_t T, int D, int64_t start_state,
                       int64_t *path_out, double *dist_out, int64_t *nsteps_out);
/* plain synchronous copies for transport implementations (device pointers handed to the callbacks) */
int snk_copy_to_host(void *dst_host, const void *src_dev, int64_t bytes);
int snk_copy_to_device(void *dst_dev, const void *src_host, int64_t bytes);

/* Waveform side (the step after the search).  Replaces retrieve_magphase_frag
 * (synth_simple.py:538-652) and the overlap-add loop of concatenateMagPhaseEpoch_sep_files
 * (:677-747) up to the call of the external vocoder.
 *   snk_upload_frames: the analysis frames of ALL database utterances, concatenated:
 *     spec (rows, 3*H) float32 = [mag | real | imag] per frame, fzv (rows, 2) float64 =
 *     [interpolated f0, voicing flag] (speech_manip.lin_interp_f0 on the host).
 *   snk_concat_fragments: n selected units; first_row[k] = frame row of unit k's first frame,
 *     [utt_lo[k], utt_hi[k]) = frame rows of its utterance; multiepoch frames per unit, `overlap`
 *     (even, <= multiepoch) cross-fade frames with the Hann weights in_taper (overlap entries,
 *     matrix_operations.py:19).  Outputs (n*multiepoch rows): spec_out (rows, 3*H) float64,
 *     fz_out (rows) float64 with unvoiced frames zeroed. */
int snk_upload_frames(snk_handle h, const float *spec, const double *fzv, int64_t rows, int H);
int snk_concat_fragments(snk_handle h, const int64_t *first_row, const int64_t *utt_lo, const int64_t *utt_hi,
                         int64_t n, int multiepoch, int overlap, const double *in_taper,
                         double *spec_out, double *fz_out);

/* Engine tuning / introspection (not part of the reference surface).  Options of the greedy search:
 *   greedy_mode 0 / 1 / 2      exact scan, a launch per step / float32 scan in one persistent launch / auto (default)
 *   greedy_hoist 0 / 1         target term of all steps as one float64 matrix product per utterance (default 1)
 *   greedy_hoist_max_gb        largest product kept on the device (default 48)
 *   greedy_f16 0 / 1 / 2       float16 join tiles: never / for databases streamed from HBM (default) / always
 *   greedy_resident 0 / 1      one utterance against a database whose windowed join matrix fits the chip's LDS: the resident
 *                              scan (greedy_res_kernels.hip; default 1)
 *   greedy_fenced 0 / 1        release / acquire fences around every hand-off of the one-launch scans (cross-check; default 0)
 * Options of the Viterbi search:
 *   viterbi_mode 0 / 1 / 2     dense float64 join + recursion / bounds + sparse exact recursion / sparse wherever supported (default)
 *   viterbi_lb_chunk, viterbi_lb_warm, viterbi_lb_chunk_max_utts
 *                              the approximate recursion (pass 2) in chunks of that many steps side by side (default 48; 0: one
 *                              chain per utterance), each started viterbi_lb_warm steps early (default 16), for launches of up to
 *                              that many utterances (default 24 = all)
 *   viterbi_sparse_waves 1 / 4 which form of the exact sparse recursion runs (process-wide; default 1)
 *   join_lb_variant 0 / 1      pass 1 of the sparse path: float32 matrix pipe, weights applied per gathered row / bf16 matrix pipe
 *                              over a float32 copy of the weighted join rows built once per snk_set_weights (default 1)
 *   viterbi_weights 0 / 1      0 (default): float64 recursion, the restatement every bit-exact test refers to.  1: the
 *                              reference's own arithmetic -- OpenFST's float32 tropical weights (fst_functions_wrapped.py:47,201:
 *                              every lattice weight is parsed into float32; :368 compose adds the two arc weights, :389
 *                              shortestpath accumulates them from the start state, all in float32): arc weight
 *                              fl32(fl32(tdist[t-1,k']) + fl32(c(k',k))), totals accumulated in float32, the last row's target
 *                              cost on the exit arc; ties as before.  This IS a different result on near ties (the cost returned
 *                              is the float32 total).  Runs wherever the float64 recursion does: on the dense kernels (viterbi_mode
 *                              0) and on the sparse path, whose proof folds the float32 roundings in (joinfast_kernels.hip).
 *   join_exact_form 0 / 1      pass 3 of the sparse path: a lane per cell / a cooperative workgroup per step (default 1)
 * Options of the K-NN filter: prefilter 0 / 1 / 2, prefilter_two_pass 0 / 1, prefilter_balls 0 / 1, prefilter_super_balls 0 / 1,
 * prefilter_ball_bound 0 / 1, coarse_gate_fraction; of the sharded search: shard_compact 0 / 1; of the greedy search:
 * greedy_hoist 0 / 1, greedy_hoist_fast 0 / 1, greedy_f16 0 / 1 / 2, greedy_resident 0 / 1, greedy_speculate 0 / 1, greedy_fenced 0 / 1
 * (INTEGRATION.md).
 * None of these changes a result (viterbi_weights excepted, which selects the arithmetic).
 * Latches (none changes a result): viterbi_latch 0 / 1 (batches in viterbi_mode 2: the dense kernels are tried where the bounds do not
 * prune, the faster path is kept, the other one re-tried after 32 .. 1 024 batches), viterbi_refine_gate, latch_rearm 0 / 1 (a voice on
 * the coarse / one-pass filter is probed with a counting form of the pass it left every 16 .. 256 calls).
 * Tripwires of the probed MFMA accumulation bound (snk_reset_timers clears them): prefilter_margin_rows / prefilter_min_margin (below),
 * join_bound_violations / join_bound_min_margin (exact join costs against pass 1's bounds), greedy_bound_violations /
 * greedy_bound_max_used (exact totals against the float32 scans' bound).
 * infos: greedy_fallbacks, greedy_stalls, greedy_exact_windows, greedy_second_rounds, greedy_hoist_launches, greedy_hoist16_launches,
 * greedy_f16_launches, greedy_f16_delta, greedy_resident_launches, greedy_last_speculated, greedy_last_several_holders, filter_coarse,
 * filter_onepass, shard_last_sent_mb.  The other names are listed in INTEGRATION.md.
 * Tripwire of the K-NN prefilter's key bound (snk_reset_timers clears it): prefilter_margin_rows = rows of prefilter K-NN
 * calls whose exact K-th key came within 2 eps of the filter threshold (eps: the largest error the approximate keys are
 * ASSUMED to have -- for the bf16-split operands that rests on the probed accumulation property below);
 * prefilter_min_margin = the smallest (threshold - exact K-th key) / eps seen: the factor by which the true key errors could
 * exceed eps before a row could lose a neighbour (the threshold is a sampled approximate key + 2 eps, so a row whose
 * threshold came from its K-th neighbour itself sits between 1 and 3; rows bounded by the sample sit far above). */
int snk_set_option(snk_handle h, const char *name, double value);
int snk_get_info(snk_handle h, const char *name, double *value_out);
/* One v_mfma_f32_32x32x16_bf16 on caller-chosen bit patterns: D = A B + C with A (32, 16) and B (16, 32) bf16 bit
 * patterns, C and D (32, 32) float32, all row-major.  The hardware's internal summation order is not documented; the
 * accumulation term of the bf16-split prefilter's key bound ASSUMES an error of at most 2^-20 of the sum of the
 * |products| and |C| per instruction (a first version assumed 2^-22; the probe refuted it at 1.6 x that), and tests/test_gpu_prefilter.py holds this probe against float64 sums on
 * adversarial patterns (one large product beside fifteen just below its rounding unit, cancellation, a large C). */
int snk_probe_mfma_bf16(snk_handle h, const uint16_t *A, const uint16_t *B, const float *C, float *D_out);
/* gfx950 self-test of the f64 MFMA fragment mapping the K-NN kernel relies on */
int snk_selftest_mfma(snk_handle h, double *max_abs_err_out);

#ifdef __cplusplus
}
#endif
#endif /* SNK_H */
