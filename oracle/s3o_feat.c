/*
 * s3o_feat.c -- CPU restatement (TEST INFRASTRUCTURE ONLY) of sphinxbase's feature computation
 * for the stream type sphinx3's continuous models use, "1s_c_d_dd":
 *
 *   sphinxbase/src/libsphinxbase/feat/feat.c:1111-1123  feat_compute_utt (CMN, AGC, then frames)
 *   feat.c:396-516      feat_s2mfc_read: the utterance is PADDED with `win` (= 3) copies of its
 *                       first and last frame before anything else -- so the copies take part in
 *                       the cepstral mean, the variance and the AGC maximum
 *   feat/cmn.c:141-208  cmn(): float32 sums in frame order, mean = sum / n, optional variance
 *                       normalisation with invstd = (float32) sqrt((float64) n / var)
 *   feat/agc.c:109-126  agc_max(): c0 -= max c0
 *   feat.c:726-769      feat_1s_c_d_dd_cep2feat: cep | c[t+2]-c[t-2] | (c[t+3]-c[t-1]) - (c[t+1]-c[t-3])
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "s3o.h"

#define WIN 3       /* FEAT_DCEP_WIN + 1 */

void
s3o_feat_1s_c_d_dd(const float *cep, int32_t n, int32_t cs, int32_t cmn, int32_t varnorm, int32_t agc_max,
                   float *feat)
{
    int32_t nfr = n + 2 * WIN, f, i;
    float *m, *mean, *var;
    if (n <= 0) return;
    m = (float *)malloc(sizeof(float) * (size_t)nfr * cs);
    mean = (float *)calloc(cs, sizeof(float));
    var = (float *)calloc(cs, sizeof(float));
    for (f = 0; f < nfr; f++) {
        int32_t src = f - WIN;
        if (src < 0) src = 0;
        if (src > n - 1) src = n - 1;
        memcpy(m + (size_t)f * cs, cep + (size_t)src * cs, sizeof(float) * cs);
    }
    if (cmn) {
        for (f = 0; f < nfr; f++)
            for (i = 0; i < cs; i++) mean[i] += m[(size_t)f * cs + i];
        for (i = 0; i < cs; i++) mean[i] /= nfr;
        if (!varnorm) {
            for (f = 0; f < nfr; f++)
                for (i = 0; i < cs; i++) m[(size_t)f * cs + i] -= mean[i];
        }
        else {
            for (f = 0; f < nfr; f++)
                for (i = 0; i < cs; i++) { float t = m[(size_t)f * cs + i] - mean[i]; var[i] += t * t; }
            for (i = 0; i < cs; i++) var[i] = (float)sqrt((double)nfr / var[i]);
            for (f = 0; f < nfr; f++)
                for (i = 0; i < cs; i++) m[(size_t)f * cs + i] = (m[(size_t)f * cs + i] - mean[i]) * var[i];
        }
    }
    if (agc_max) {
        float mx = m[0];
        for (f = 1; f < nfr; f++) if (m[(size_t)f * cs] > mx) mx = m[(size_t)f * cs];
        for (f = 0; f < nfr; f++) m[(size_t)f * cs] -= mx;
    }
    for (f = WIN; f < nfr - WIN; f++) {
        float *o = feat + (size_t)(f - WIN) * 3 * cs;
        const float *c = m + (size_t)f * cs;
        memcpy(o, c, sizeof(float) * cs);
        for (i = 0; i < cs; i++) o[cs + i] = c[2 * cs + i] - c[-2 * cs + i];
        for (i = 0; i < cs; i++) {
            float d1 = c[3 * cs + i] - c[-1 * cs + i], d2 = c[1 * cs + i] - c[-3 * cs + i];
            o[2 * cs + i] = d1 - d2;
        }
    }
    free(m); free(mean); free(var);
}
