// Waveform-side gather for gfx950: the selected units' analysis frames, cross-faded and
// overlap-added into the matrices the vocoder consumes.
//
// Replaces retrieve_magphase_frag (script/synth_simple.py:538-652) and the accumulation loop of
// concatenateMagPhaseEpoch_sep_files (:677-747) up to the call of the external vocoder
// (magphase.synthesis_from_lossless):
//   fragment k = frames [start_k - ov/2, start_k + me + ov/2) of the utterance that holds unit
//   path[k], zero-padded where that runs past the utterance, its first / last `ov` frames weighted
//   by a Hann cross-fade (matrix_operations.py:16-30); fragment k is ADDED at output frame k*me;
//   ov/2 frames are trimmed at both ends; f0 is zeroed where the summed voicing flag is < 0.5.
// Every output element has at most ceil((me+ov)/me) contributions, added here in path order like the
// reference's `+=`.  Rounding follows the reference exactly: an unpadded fragment is a float32 view
// there, so its tapered rows are products rounded to float32; a padded fragment has been promoted to
// float64 by the zero padding, so its products stay float64; f0 and voicing are float64 throughout.
// HBM-bound gather: (me+ov)/me x 3H x 4 bytes read per output frame.
#include "snk_internal.h"

namespace snk {

__global__ void __launch_bounds__(256)
concat_fragments_kernel(const float *__restrict__ spec, int W /* 3H */, const double *__restrict__ fzv,
                        const int64_t *__restrict__ first_row, const int64_t *__restrict__ utt_lo,
                        const int64_t *__restrict__ utt_hi, int64_t n, int me, int ov,
                        const double *__restrict__ in_taper, double *__restrict__ out_spec /* (n*me, W) */,
                        double *__restrict__ out_fz /* (n*me) */)
{
    const int64_t jo = blockIdx.x;                 // output frame after trimming
    const int64_t j = jo + ov / 2;                 // frame of the untrimmed accumulation
    const int m = me + ov;
    int64_t k_lo = (j - m + me) / me;              // ceil((j - m + 1) / me) for j - m + 1 > 0
    if (j - m + 1 <= 0) k_lo = 0;
    int64_t k_hi = j / me;
    if (k_hi > n - 1) k_hi = n - 1;
    for (int c = threadIdx.x; c < W + 1; c += blockDim.x) {
        double acc = 0.0, vacc = 0.0;
        for (int64_t k = k_lo; k <= k_hi; ++k) {
            const int r = (int)(j - k * me);       // row inside fragment k
            if (r < 0 || r >= m) continue;
            const int64_t start = first_row[k] - ov / 2;
            const int64_t src = start + r;
            const bool padded = start < utt_lo[k] || start + m > utt_hi[k];
            if (src < utt_lo[k] || src >= utt_hi[k]) continue;      // zero padding: adds +0.0
            double w = 1.0;
            if (ov > 0) {
                if (r < ov) w = in_taper[r];
                else if (r >= m - ov) w = in_taper[m - 1 - r];
            }
            if (c < W) {
                const float x = spec[src * W + c];
                double v = (double)x;
                if (w != 1.0) {
                    v = __dmul_rn(v, w);
                    if (!padded) v = (double)(float)v;              // in-place multiply of a float32 view
                }
                acc = __dadd_rn(acc, v);
            } else {
                acc = __dadd_rn(acc, (w != 1.0) ? __dmul_rn(fzv[2 * src], w) : fzv[2 * src]);
                vacc = __dadd_rn(vacc, (w != 1.0) ? __dmul_rn(fzv[2 * src + 1], w) : fzv[2 * src + 1]);
            }
        }
        if (c < W) out_spec[jo * W + c] = acc;
        else out_fz[jo] = (vacc < 0.5) ? 0.0 : acc;
    }
}

void launch_concat_fragments(const float *spec, int W, const double *fzv, const int64_t *first_row,
                             const int64_t *utt_lo, const int64_t *utt_hi, int64_t n, int me, int ov,
                             const double *in_taper, double *out_spec, double *out_fz, hipStream_t s)
{
    hipLaunchKernelGGL(concat_fragments_kernel, dim3((unsigned)(n * me)), dim3(256), 0, s, spec, W, fzv, first_row,
                       utt_lo, utt_hi, n, me, ov, in_taper, out_spec, out_fz);
}

}  // namespace snk
