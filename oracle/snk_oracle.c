/*
 * CPU ORACLE in C (test infrastructure only -- never linked or called by the product).
 * Same algorithms and the same canonical float order as oracle/snk_oracle.py (which is pinned
 * against reference-generated golden vectors); exists so that parity checks at sizes where the
 * numpy loops are too slow still finish in seconds.  Build: -ffp-contract=off (no FMA).
 *
 * Restates (paths relative to the reference checkout, script/...):
 *   knn       cKDTree(F).query(U, k=K)                 synth_halfphone.py:379,1364
 *   join      get_natural_distance_vectorised          synth_halfphone.py:2942-2951, :3238-3301
 *   viterbi   T o J shortest path                      fst_functions_wrapped.py:28-58,172-217,368,389
 *   greedy    greedy_joint_search                      synth_simple.py:458-503
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define VERY_BIG 1000000000000000.0

static double sqdist(const double *a, const double *b, int d)
{
    double acc = 0.0;
    for (int c = 0; c < d; ++c) {
        double x = a[c] - b[c];
        acc = acc + x * x;
    }
    return acc;
}

typedef struct { double d; int64_t i; } pair_t;
static int pair_cmp(const void *pa, const void *pb)
{
    const pair_t *a = (const pair_t *)pa, *b = (const pair_t *)pb;
    if (a->d < b->d) return -1;
    if (a->d > b->d) return 1;
    return (a->i > b->i) - (a->i < b->i);
}

/* F (N,D) f64 weighted, U (T,D); cand (T,K) i64, dist (T,K) f64; order (distance, id) */
int snko_knn(const double *F, int64_t N, int D, const double *U, int64_t T, int K, int64_t *cand, double *dist)
{
#pragma omp parallel
    {
        pair_t *buf = (pair_t *)malloc((size_t)N * sizeof(pair_t));
#pragma omp for schedule(dynamic, 1)
        for (int64_t t = 0; t < T; ++t) {
            for (int64_t i = 0; i < N; ++i) { buf[i].d = sqdist(F + i * D, U + t * D, D); buf[i].i = i; }
            qsort(buf, (size_t)N, sizeof(pair_t), pair_cmp);
            for (int k = 0; k < K; ++k) {
                if (k < N) { cand[t * K + k] = buf[k].i; dist[t * K + k] = sqrt(buf[k].d); }
                else { cand[t * K + k] = -1; dist[t * K + k] = VERY_BIG; }
            }
        }
        free(buf);
    }
    return 0;
}

static int usable(int64_t id, int64_t n_units) { return id >= 1 && id < n_units - 1; }

/* JC (n_units+1, Dj) weighted; J (T-1,K,K) */
int snko_join(const double *JC, int64_t n_units, int Dj, const int64_t *cand, int64_t T, int K, double *J)
{
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t t = 0; t < T - 1; ++t)
        for (int a = 0; a < K; ++a)
            for (int b = 0; b < K; ++b) {
                const int64_t f = cand[t * K + a], s = cand[(t + 1) * K + b];
                double v = INFINITY;
                if (usable(f, n_units) && usable(s, n_units))
                    v = sqrt(sqdist(JC + (f + 1) * Dj, JC + s * Dj, Dj));
                J[(t * K + a) * K + b] = v;
            }
    return 0;
}

/* returns path length (T or 0) */
int64_t snko_viterbi(const int64_t *cand, const double *tdist, const double *J, int64_t T, int K,
                     int64_t n_units, int64_t *path, double *cost)
{
    *cost = INFINITY;
    if (T < 2) return 0;
    double *delta = (double *)malloc((size_t)K * sizeof(double));
    double *nd = (double *)malloc((size_t)K * sizeof(double));
    int *bp = (int *)malloc((size_t)T * K * sizeof(int));
    for (int k = 0; k < K; ++k) delta[k] = usable(cand[k], n_units) ? tdist[k] : INFINITY;
    for (int64_t t = 1; t < T; ++t) {
        for (int k = 0; k < K; ++k) {
            double best = INFINITY; int arg = 0;
            for (int kp = 0; kp < K; ++kp) {
                double tot = delta[kp] + J[((t - 1) * K + kp) * K + k];
                if (tot < best) { best = tot; arg = kp; }
            }
            bp[t * K + k] = arg;
            nd[k] = usable(cand[t * K + k], n_units) ? tdist[t * K + k] + best : INFINITY;
        }
        memcpy(delta, nd, (size_t)K * sizeof(double));
    }
    double best = INFINITY; int slot = 0;
    for (int k = 0; k < K; ++k) if (delta[k] < best) { best = delta[k]; slot = k; }
    int64_t len = 0;
    if (best < INFINITY) {
        *cost = best; len = T;
        for (int64_t t = T - 1; t >= 0; --t) { path[t] = cand[t * K + slot]; if (t > 0) slot = bp[t * K + slot]; }
    }
    free(delta); free(nd); free(bp);
    return len;
}

/* prev_rep/cur_rep (Nwin, jd) and Fwin addressed through F (N,Dt) with window me; Q (steps, nep*Dt) */
int snko_greedy(const double *prev_rep, const double *cur_rep, int jd, const double *F, int Dt, int me,
                const int *ep, int nep, int64_t Nwin, const double *Q, int64_t steps, int64_t start_state,
                int64_t *path, double *dists)
{
    double *prev = (double *)calloc((size_t)jd, sizeof(double));
    double *d2 = (double *)malloc((size_t)Nwin * sizeof(double));
    if (start_state >= 0) memcpy(prev, prev_rep + start_state * jd, (size_t)jd * sizeof(double));
    for (int64_t s = 0; s < steps; ++s) {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < Nwin; ++i) {
            double aj = sqdist(prev_rep + i * jd, prev, jd);
            double at = 0.0;
            for (int k = 0; k < nep; ++k) {
                const double *f = F + (i + ep[k]) * Dt, *q = Q + (s * nep + k) * Dt;
                for (int c = 0; c < Dt; ++c) { double x = f[c] - q[c]; at = at + x * x; }
            }
            d2[i] = aj + at;
        }
        int64_t arg = 0;
        for (int64_t i = 1; i < Nwin; ++i) if (d2[i] < d2[arg]) arg = i;
        path[s] = arg; dists[s] = sqrt(d2[arg]);
        memcpy(prev, cur_rep + arg * jd, (size_t)jd * sizeof(double));
    }
    free(prev); free(d2);
    return 0;
}

/* The same search straight from the UNWEIGHTED float32 matrices (what the HDF5 holds), weights applied
 * on the fly: fl64(f32 * w) is the value speech_manip.weight() produces (synth_simple.py:245,269), so
 * this equals snko_greedy on the weighted float64 copies without needing 8 bytes per database cell --
 * the form the parity tests use at N = 1.5 M.  Layout as get_tree_for_greedy_search
 * (synth_simple.py:190-225, synth_halfphone.py:539-596):
 *   split 0: prev_join_rep[i] = JCw[i], current_join_rep[i] = JCw[i + me]   (unit_start / unit_end, :194-195,213-214)
 *   split 1: prev = first half of the columns of JCw[i], current = second half of JCw[i + me - 1]  (:552-553)
 * Q (steps * nep, Dt) weighted query rows in window order; lowest index wins exact ties.
 * dists may be NULL; d2_first (Nwin) may be NULL, else it receives the squared distances of step
 * `d2_step` (for tests of the (1+eps) contract of approximate searches). */
int snko_greedy_f32(const float *F_unw, int64_t N, int Dt, const double *wt, const float *JC_unw, int Dj,
                    const double *wj, int me, const int *ep, int nep, int split, const double *Q, int64_t steps,
                    int64_t start_state, int64_t *path, double *dists, int64_t d2_step, double *d2_out)
{
    const int64_t Nwin = N - me + 1;
    const int jd = split ? Dj / 2 : Dj;
    const int pcol0 = 0, ccol0 = split ? Dj / 2 : 0;
    const int64_t crow0 = split ? me - 1 : me;
    double *prev = (double *)calloc((size_t)jd, sizeof(double));
    double *d2 = (double *)malloc((size_t)Nwin * sizeof(double));
    if (start_state >= 0)
        for (int c = 0; c < jd; ++c) prev[c] = (double)JC_unw[start_state * Dj + pcol0 + c] * wj[pcol0 + c];
    for (int64_t s = 0; s < steps; ++s) {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < Nwin; ++i) {
            double aj = 0.0, at = 0.0;
            const float *jr = JC_unw + i * Dj + pcol0;
            for (int c = 0; c < jd; ++c) {
                double x = (double)jr[c] * wj[pcol0 + c];
                x = x - prev[c];
                aj = aj + x * x;
            }
            for (int k = 0; k < nep; ++k) {
                const float *f = F_unw + (i + ep[k]) * Dt;
                const double *q = Q + (s * nep + k) * Dt;
                for (int c = 0; c < Dt; ++c) {
                    double x = (double)f[c] * wt[c];
                    x = x - q[c];
                    at = at + x * x;
                }
            }
            d2[i] = aj + at;
        }
        int64_t arg = 0;
        for (int64_t i = 1; i < Nwin; ++i) if (d2[i] < d2[arg]) arg = i;
        path[s] = arg;
        if (dists) dists[s] = sqrt(d2[arg]);
        if (d2_out && s == d2_step) memcpy(d2_out, d2, (size_t)Nwin * sizeof(double));
        for (int c = 0; c < jd; ++c) prev[c] = (double)JC_unw[(crow0 + arg) * Dj + ccol0 + c] * wj[ccol0 + c];
    }
    free(prev); free(d2);
    return 0;
}
