// Shared by the two greedy scans (greedy_kernels.hip: exact float64; greedy32_kernels.hip: float32 prefilter).
#pragma once
#include "snk_internal.h"
#include <float.h>

namespace snk {

#define GR_CC 32          // columns per chunk (8 float4 per window)
#define GR_NSTG 3         // ring stages (chunks) in registers per thread: GR_NSTG - 1 requests ahead of the arithmetic
#define GR_MAX_EP 16      // max multiepoch
#define GR_S1 256          // arrival counters of the first / second level (see greedy_finish_step)
#define GR_S2 16

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GreedyArgs {
    const float *JC_unw; int Jp, Dj; const double *wj;   // Jp / Fp: row pitch in floats (multiple of 4,
    const float *F_unw; int Fp, Dt; const double *wt;    // zero-filled padding columns)
    const f32x4 *JT, *FT;                 // lane-major tiles of the scan columns (greedy_tile_kernel)
    int me, nep; int ep[GR_MAX_EP];       // epochs of the window that enter the target term
    int prev_col0, cur_col0, jdim;
    int64_t prev_row0, cur_row0, Nwin;
    int64_t n_jc_rows, n_f_rows;          // matrix heights (clamp for the ragged last workgroup)
    const double *Q;                      // (rows, Dt) weighted targets of all utterances, row-major
    int lds_mode;                         // 1: target rows in LDS, interleaved chunk order (greedy_step_kernel)
    // utterances of this scan (snk_greedy_batch: up to GR_MAXU share one pass over the database)
    int nu;
    int64_t q_off[6], nsteps_u[6], out_off[6];      // first query row, steps, first slot in path / dist
    int64_t start[6];                               // start states of the float32 scan's utterances (-1: none)
    // hoisted target term (greedy_hoist_kernels.hip): W[u][step][window] float32, row pitch Wp; the scan then reads
    // the join columns only.  qn2[u][step] = ||target vector of the step||^2, fwmax2 = max ||window||^2 (weighted),
    // hoist_c: relative constant of the bound |W~ - W| <= hoist_c (||q|| + ||f||max)^2
    int hoist;
    const float *W[6]; int64_t Wp;
    const double *qn2[6];
    double hoist_c, fwmax2;
    // float16 copy of the join tiles (8 columns per 16 bytes, chunks of 64 columns) and the bound's term for it
    const f32x4 *JT16; int f16; double f16_delta;
    // scan range of the exact step kernel in 64-window tiles (tile_n 0: every tile): a rank of snk_sharded_greedy scans its share
    // of the windows; its winner then goes to shard_out ({float64 squared distance, int64 window}) instead of the path, and the
    // next step's table is written by greedy_shard_pick_kernel once every rank's winner is there
    int64_t tile_lo, tile_n;
    double *shard_out;
};
#define GR_MAXU 2          // utterances per scan: one weight and the references share 32 table bytes per column

__device__ __forceinline__ int greedy_join_chunks(const GreedyArgs &a) { return (a.jdim + GR_CC - 1) / GR_CC; }
__device__ __forceinline__ int greedy_target_chunks(const GreedyArgs &a) { return (a.Dt + GR_CC - 1) / GR_CC; }

// Chunk order of a window's scan.  Plain order (lds_mode 0): the join chunks, then the
// target chunks epoch by epoch.  Interleaved order (lds_mode 1): join chunk j is followed
// by the target chunks [j*nT/jch, (j+1)*nT/jch) -- the join chunks come from HBM, the target chunks
// from LDS, and spreading the first among the second gives every HBM request several chunks of
// arithmetic to land behind.  The two partial sums are separate accumulators, so only the order
// within each kind matters for the result.
// Slot c -> join chunk (returns true, *idx = j) or target chunk (returns false, *idx = k*tch + cc).
__device__ __forceinline__ bool greedy_chunk_slot(const GreedyArgs &a, int jch, int nT, int c, int *idx)
{
    if (!a.lds_mode) {
        if (c < jch) { *idx = c; return true; }
        *idx = c - jch; return false;
    }
    int j = 0;
    while (j + 1 < jch && (j + 1) + ((j + 1) * nT) / jch <= c) ++j;      // last join chunk at or before slot c
    if (c == j + (j * nT) / jch) { *idx = j; return true; }
    *idx = c - j - 1;
    return false;
}


// host side (greedy_kernels.hip)
void greedy_fill_args(GreedyArgs &a, const GreedyLayout &g, const float *F_unw, int Fp, int Dt, const double *wt,
                      const float *JC_unw, int Jp, int Dj, const double *wj, const double *Q, bool greedy_mode,
                      const float *tiles);
bool greedy_lds_mode(const GreedyLayout &g, int Dt);

}  // namespace snk
