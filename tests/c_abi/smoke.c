/* A caller of libsnkhip.so that is neither Python nor torch: plain C against include/snk.h.
 * Deterministic database and queries (a fixed LCG), every search entry point once; prints the results
 * as text.  tests/test_gpu_c_abi.py compiles it with gcc, runs it and compares the printed ids and
 * costs with the same calls made through the ctypes binding (and hence with the oracle). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "snk.h"

static uint64_t lcg_state = 12345;
static double lcg(void)                       /* uniform in [-1, 1) */
{
    lcg_state = lcg_state * 6364136223846793005ULL + 1442695040888963407ULL;
    return (double)((lcg_state >> 11) & ((1ULL << 53) - 1)) / (double)(1ULL << 52) - 1.0;
}

#define CHECK(call) do { if ((call) != 0) { fprintf(stderr, "%s failed: %s\n", #call, snk_last_error()); return 1; } } while (0)

int main(void)
{
    const int64_t N = 5000;
    const int Dt = 61, Dj = 40, K = 12, T = 30, me = 3;
    float *F = malloc(sizeof(float) * N * Dt), *JC = malloc(sizeof(float) * (N + 1) * Dj);
    double *wt = malloc(sizeof(double) * Dt), *wj = malloc(sizeof(double) * Dj);
    double *Q = malloc(sizeof(double) * T * Dt);
    /* random walks (speech-like continuity), as single precision like the HDF5 arrays */
    for (int c = 0; c < Dt; ++c) { double v = 0; for (int64_t i = 0; i < N; ++i) { v += 0.1 * lcg(); F[i * Dt + c] = (float)v; } }
    for (int c = 0; c < Dj; ++c) { double v = 0; for (int64_t i = 0; i <= N; ++i) { v += 0.1 * lcg(); JC[i * Dj + c] = (float)v; } }
    for (int c = 0; c < Dt; ++c) wt[c] = 0.4;
    for (int c = 0; c < Dj; ++c) wj[c] = 0.05;
    for (int t = 0; t < T; ++t)
        for (int c = 0; c < Dt; ++c) Q[t * Dt + c] = ((double)F[(1000 + t) * Dt + c] + 0.05 * lcg()) * wt[c];

    snk_handle h;
    CHECK(snk_create(0, &h));
    CHECK(snk_upload_db(h, F, N, Dt, JC, N + 1, Dj));
    CHECK(snk_set_weights(h, wt, Dt, wj, Dj));

    int64_t *cand = malloc(sizeof(int64_t) * T * K), *path = malloc(sizeof(int64_t) * T), plen = 0;
    double *dist = malloc(sizeof(double) * T * K), cost = 0;
    CHECK(snk_knn(h, Q, T, Dt, K, cand, dist));
    printf("knn_row0");
    for (int k = 0; k < K; ++k) printf(" %lld", (long long)cand[k]);
    printf("\nknn_dist0 %.17g %.17g\n", dist[0], dist[K - 1]);
    CHECK(snk_viterbi(h, cand, dist, T, K, path, &plen, &cost));
    printf("viterbi %lld %.17g", (long long)plen, cost);
    for (int64_t t = 0; t < plen; ++t) printf(" %lld", (long long)path[t]);
    printf("\n");

    /* the batch entry point: two utterances (rows 0..17 and 18..29) */
    const int64_t offs[3] = {0, 18, 30};
    int64_t *bpath = malloc(sizeof(int64_t) * T), blen[2];
    double bcost[2];
    CHECK(snk_knn_viterbi_batch(h, Q, offs, 2, Dt, K, bpath, blen, bcost));
    printf("batch %lld %lld %.17g %.17g", (long long)blen[0], (long long)blen[1], bcost[0], bcost[1]);
    for (int t = 0; t < T; ++t) printf(" %lld", (long long)bpath[t]);
    printf("\n");

    /* greedy search over multiepoch windows */
    int64_t *gpath = malloc(sizeof(int64_t) * T), nsteps = 0;
    double *gdist = malloc(sizeof(double) * T);
    CHECK(snk_set_greedy_layout(h, me, 0, 0));
    CHECK(snk_greedy(h, Q, T, Dt, -1, 0.0, gpath, gdist, &nsteps));
    printf("greedy %lld", (long long)nsteps);
    for (int64_t s = 0; s < nsteps; ++s) printf(" %lld", (long long)gpath[s]);
    printf(" %.17g\n", gdist[0]);
    CHECK(snk_destroy(h));
    return 0;
}
