/*
 * s3o_fe.c -- CPU restatement (TEST INFRASTRUCTURE ONLY) of sphinxbase's MFCC front end, the
 * step in front of feat_s2mfc2feat, floating-point build (frame_t =
 * powspec_t = window_t = float64, mfcc_t = float32):
 *
 *   sphinxbase/src/libsphinxbase/fe/fe_interface.c:69-127   fe_parse_general_params
 *   fe_interface.c:129-163                                   fe_parse_melfb_params
 *   fe_interface.c:212-283                                   fe_init_auto_r (frame_shift / frame_size rounding)
 *   fe_interface.c:336-483                                   fe_process_frames / fe_process_utt (framing)
 *   fe_interface.c:486-502                                   fe_end_utt (the final, partial frame)
 *   fe_sigproc.c:288-301                                     fe_mel / fe_melinv (neutral warp)
 *   fe_sigproc.c:303-427                                     fe_build_melfilters
 *   fe_sigproc.c:429-466                                     fe_compute_melcosine (+ lifter weights)
 *   fe_sigproc.c:469-498, :516-568                           fe_pre_emphasis, fe_create_hamming, fe_hamming_window
 *   fe_sigproc.c:570-594                                     fe_spch_to_frame
 *   fe_sigproc.c:645-667, :792-889                           fe_create_twiddle, fe_fft_real
 *   fe_sigproc.c:891-934, :936-966                           fe_spec_magnitude, fe_mel_spec
 *   fe_sigproc.c:968-1014                                    fe_mel_cep
 *   fe_sigproc.c:1016-1042, :1044-1067, :1069-1080, :1082-1095   fe_spec2cep, fe_dct2, fe_lifter, fe_dct3
 *
 * Whole-utterance view of the streaming code: frame i covers samples [i*shift, i*shift +
 * frame_size); there are 1 + (n - frame_size) / shift full frames (n >= frame_size), then
 * fe_end_utt turns the remaining samples from (number of full frames) * shift on -- always at
 * least frame_size - shift of them -- into ONE more frame, zero-padded.  The pre-emphasis
 * "prior" carried from frame to frame is exactly the sample in front of the frame's first, so
 * pre-emphasis is the utterance-wide  y[k] = x[k] - alpha * x[k-1], x[-1] = 0.
 *
 * Types that fix the bit patterns: everything up to the log mel spectrum is float64; the
 * mel filter weights, the DCT basis and the normalisers are float32; the cepstra ACCUMULATE
 * in float32 storage (every += is a float64 add rounded to float32).
 * Not restated: dithering (random), frequency warping with parameters, swapped input.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "s3o.h"

s3o_fe_t *
s3o_fe_init(const s3o_fe_params_t *p)
{
    s3o_fe_t *fe = (s3o_fe_t *)calloc(1, sizeof *fe);
    int32_t i, j, nf = p->nfilt, n_coeffs;
    float melmin, melmax, melbw, fftfreq;
    fe->p = *p;
    for (j = p->nfft, fe->fft_order = 0; j > 1; j >>= 1, fe->fft_order++)
        if ((j % 2) != 0 || p->nfft <= 0) { free(fe); return NULL; }
    if (p->nfft < (int)(p->wlen * p->samprate)) { free(fe); return NULL; }
    fe->frame_shift = (int32_t)(p->samprate / (int16_t)p->frate + 0.5);     /* fe_interface.c:231 */
    fe->frame_size = (int32_t)(p->wlen * p->samprate + 0.5);
    if (fe->frame_size > p->nfft) { free(fe); return NULL; }
    fe->feature_dimension = p->logspec ? nf : p->ncep;
    /* fe_create_hamming */
    fe->hamming = (double *)calloc(fe->frame_size / 2 + 1, sizeof(double));
    for (i = 0; i < fe->frame_size / 2; i++)
        fe->hamming[i] = (0.54 - 0.46 * cos(2 * M_PI * i / ((double)fe->frame_size - 1.0)));
    /* fe_create_twiddle */
    fe->ccc = (double *)calloc(p->nfft / 4 + 1, sizeof(double));
    fe->sss = (double *)calloc(p->nfft / 4 + 1, sizeof(double));
    for (i = 0; i < p->nfft / 4; ++i) {
        double a = 2 * M_PI * i / p->nfft;
        fe->ccc[i] = cos(a);
        fe->sss[i] = sin(a);
    }
    /* fe_build_melfilters (neutral warp: warped == unwarped) */
#define MEL(x) ((float)(2595.0 * log10(1.0 + (float)(x) / 700.0)))
#define MELINV(x) ((float)(700.0 * (pow(10.0, (float)(x) / 2595.0) - 1.0)))
    fe->spec_start = (int16_t *)calloc(nf, sizeof(int16_t));
    fe->filt_start = (int16_t *)calloc(nf, sizeof(int16_t));
    fe->filt_width = (int16_t *)calloc(nf, sizeof(int16_t));
    melmin = MEL(p->lowerf);
    melmax = MEL(p->upperf);
    melbw = (melmax - melmin) / (nf + 1);
    if (p->doublebw) {
        melmin -= melbw;
        melmax += melbw;
        if (MELINV(melmin) < 0 || MELINV(melmax) > p->samprate / 2) { s3o_fe_free(fe); return NULL; }
    }
    fftfreq = p->samprate / (float)p->nfft;
    n_coeffs = 0;
    for (i = 0; i < nf; ++i) {
        float freqs[3];
        for (j = 0; j < 3; ++j) {
            if (p->doublebw) freqs[j] = MELINV((i + j * 2) * melbw + melmin);
            else freqs[j] = MELINV((i + j) * melbw + melmin);
            if (p->round_filters) freqs[j] = ((int)(freqs[j] / fftfreq + 0.5)) * fftfreq;
        }
        fe->spec_start[i] = -1;
        for (j = 0; j < p->nfft / 2 + 1; ++j) {
            float hz = j * fftfreq;
            if (hz < freqs[0]) continue;
            else if (hz > freqs[2] || j == p->nfft / 2) {
                fe->filt_width[i] = (int16_t)(j - fe->spec_start[i]);
                fe->filt_start[i] = (int16_t)n_coeffs;
                n_coeffs += fe->filt_width[i];
                break;
            }
            if (fe->spec_start[i] == -1) fe->spec_start[i] = (int16_t)j;
        }
    }
    fe->n_coeffs = n_coeffs;
    fe->filt_coeffs = (float *)calloc(n_coeffs > 0 ? n_coeffs : 1, sizeof(float));
    n_coeffs = 0;
    for (i = 0; i < nf; ++i) {
        float freqs[3];
        for (j = 0; j < 3; ++j) {
            if (p->doublebw) freqs[j] = MELINV((i + j * 2) * melbw + melmin);
            else freqs[j] = MELINV((i + j) * melbw + melmin);
            if (p->round_filters) freqs[j] = ((int)(freqs[j] / fftfreq + 0.5)) * fftfreq;
        }
        for (j = 0; j < fe->filt_width[i]; ++j) {
            float hz, loslope, hislope;
            hz = (fe->spec_start[i] + j) * fftfreq;
            loslope = (hz - freqs[0]) / (freqs[1] - freqs[0]);
            hislope = (freqs[2] - hz) / (freqs[2] - freqs[1]);
            if (p->unit_area) {
                loslope *= 2 / (freqs[2] - freqs[0]);
                hislope *= 2 / (freqs[2] - freqs[0]);
            }
            fe->filt_coeffs[n_coeffs++] = (loslope < hislope) ? loslope : hislope;
        }
    }
    /* fe_compute_melcosine */
    {
        double freqstep = M_PI / nf;
        fe->mel_cosine = (float *)calloc((size_t)p->ncep * nf, sizeof(float));
        for (i = 0; i < p->ncep; i++)
            for (j = 0; j < nf; j++)
                fe->mel_cosine[i * nf + j] = (float)cos(freqstep * i * (j + 0.5));
        fe->sqrt_inv_n = (float)sqrt(1.0 / nf);
        fe->sqrt_inv_2n = (float)sqrt(2.0 / nf);
        fe->lifter = (float *)calloc(p->ncep, sizeof(float));
        if (p->lifter)
            for (i = 0; i < p->ncep; ++i)
                fe->lifter[i] = (float)(1 + p->lifter / 2 * sin(i * M_PI / p->lifter));  /* (integer lifter / 2) */
    }
    return fe;
}

void
s3o_fe_free(s3o_fe_t *fe)
{
    if (!fe) return;
    free(fe->hamming); free(fe->ccc); free(fe->sss); free(fe->spec_start); free(fe->filt_start);
    free(fe->filt_width); free(fe->filt_coeffs); free(fe->mel_cosine); free(fe->lifter);
    free(fe);
}

int32_t
s3o_fe_n_frames(const s3o_fe_t *fe, int64_t nsamps)
{
    int64_t full = nsamps < fe->frame_size ? 0 : 1 + (nsamps - fe->frame_size) / fe->frame_shift;
    /* fe_end_utt: whatever is left in the overflow buffer becomes one more frame */
    int64_t left = nsamps - full * fe->frame_shift;
    return (int32_t)(full + (left > 0 ? 1 : 0));
}

/* fe_fft_real, fe_sigproc.c:792-889 */
static void
fft_real(const s3o_fe_t *fe, double *x)
{
    int i, j, k, m = fe->fft_order, n = fe->p.nfft;
    double xt;
    j = 0;
    for (i = 0; i < n - 1; ++i) {
        if (i < j) { xt = x[j]; x[j] = x[i]; x[i] = xt; }
        k = n / 2;
        while (k <= j) { j -= k; k /= 2; }
        j += k;
    }
    for (i = 0; i < n; i += 2) {
        xt = x[i];
        x[i] = (xt + x[i + 1]);
        x[i + 1] = (xt - x[i + 1]);
    }
    for (k = 1; k < m; ++k) {
        int n4 = k - 1, n2 = k, n1 = k + 1;
        for (i = 0; i < n; i += (1 << n1)) {
            xt = x[i];
            x[i] = (xt + x[i + (1 << n2)]);
            x[i + (1 << n2)] = (xt - x[i + (1 << n2)]);
            x[i + (1 << n2) + (1 << n4)] = -x[i + (1 << n2) + (1 << n4)];
            for (j = 1; j < (1 << n4); ++j) {
                double cc, ss, t1, t2;
                int i1 = i + j, i2 = i + (1 << n2) - j, i3 = i + (1 << n2) + j, i4 = i + (1 << n2) + (1 << n2) - j;
                cc = fe->ccc[j << (m - n1)];
                ss = fe->sss[j << (m - n1)];
                t1 = x[i3] * cc + x[i4] * ss;
                t2 = x[i3] * ss - x[i4] * cc;
                x[i4] = (x[i2] - t2);
                x[i3] = (-x[i2] - t2);
                x[i2] = (x[i1] - t1);
                x[i1] = (x[i1] + t1);
            }
        }
    }
}

/* one frame: samples in[0..len) (prior = the sample in front), len <= frame_size */
static void
one_frame(const s3o_fe_t *fe, const int16_t *in, int32_t len, int16_t prior, float *mfcep)
{
    const s3o_fe_params_t *p = &fe->p;
    int32_t i, j, nf = p->nfilt, n = p->nfft, fs = fe->frame_size;
    double *x = (double *)calloc(n, sizeof(double));
    double *spec = (double *)calloc(n / 2 + 1, sizeof(double));
    double *mfspec = (double *)calloc(nf, sizeof(double));
    /* fe_spch_to_frame */
    if (p->alpha != 0.0) {
        x[0] = (double)in[0] - (double)prior * p->alpha;
        for (i = 1; i < len; i++) x[i] = (double)in[i] - (double)in[i - 1] * p->alpha;
    }
    else
        for (i = 0; i < len; i++) x[i] = (double)in[i];
    /* (zero padded by calloc) ; fe_hamming_window over frame_size */
    if (p->remove_dc) {
        double mean = 0;
        for (i = 0; i < fs; i++) mean += x[i];
        mean /= fs;
        for (i = 0; i < fs; i++) x[i] -= mean;
    }
    for (i = 0; i < fs / 2; i++) {
        x[i] = x[i] * fe->hamming[i];
        x[fs - 1 - i] = x[fs - 1 - i] * fe->hamming[i];
    }
    fft_real(fe, x);
    /* fe_spec_magnitude */
    spec[0] = x[0] * x[0];
    for (j = 1; j <= n / 2; j++) spec[j] = x[j] * x[j] + x[n - j] * x[n - j];
    /* fe_mel_spec */
    for (i = 0; i < nf; i++) {
        mfspec[i] = 0;
        for (j = 0; j < fe->filt_width[i]; j++)
            mfspec[i] += spec[fe->spec_start[i] + j] * fe->filt_coeffs[fe->filt_start[i] + j];
    }
    /* fe_mel_cep */
    for (i = 0; i < nf; ++i) {
        if (mfspec[i] > 0) mfspec[i] = log(mfspec[i]);
        else mfspec[i] = -10.0;
    }
    if (p->logspec == 1) {
        for (i = 0; i < fe->feature_dimension; i++) mfcep[i] = (float)mfspec[i];
    }
    else if (p->logspec == 2) {
        /* fe_dct2(htk = 0) then fe_dct3 */
        float *c = (float *)calloc(p->ncep, sizeof(float));
        c[0] = (float)mfspec[0];
        for (j = 1; j < nf; j++) c[0] = (float)(c[0] + mfspec[j]);
        c[0] = c[0] * fe->sqrt_inv_n;
        for (i = 1; i < p->ncep; ++i) {
            c[i] = 0;
            for (j = 0; j < nf; j++) c[i] = (float)(c[i] + mfspec[j] * fe->mel_cosine[i * nf + j]);
            c[i] = c[i] * fe->sqrt_inv_2n;
        }
        for (i = 0; i < nf; ++i) {
            mfspec[i] = c[0] * 0.707106781186548;                 /* COSMUL(mfcep[0], SQRT_HALF): float * double constant */
            for (j = 1; j < p->ncep; j++) mfspec[i] += c[j] * fe->mel_cosine[j * nf + i];    /* float * float */
            mfspec[i] = mfspec[i] * fe->sqrt_inv_2n;
        }
        for (i = 0; i < fe->feature_dimension; i++) mfcep[i] = (float)mfspec[i];
        free(c);
    }
    else if (p->transform == 1 || p->transform == 2) {      /* fe_dct2 */
        mfcep[0] = (float)mfspec[0];
        for (j = 1; j < nf; j++) mfcep[0] = (float)(mfcep[0] + mfspec[j]);
        mfcep[0] = mfcep[0] * (p->transform == 2 ? fe->sqrt_inv_2n : fe->sqrt_inv_n);
        for (i = 1; i < p->ncep; ++i) {
            mfcep[i] = 0;
            for (j = 0; j < nf; j++) mfcep[i] = (float)(mfcep[i] + mfspec[j] * fe->mel_cosine[i * nf + j]);
            mfcep[i] = mfcep[i] * fe->sqrt_inv_2n;
        }
    }
    else {                                                   /* fe_spec2cep */
        mfcep[0] = (float)(mfspec[0] / 2);
        for (j = 1; j < nf; j++) mfcep[0] = (float)(mfcep[0] + mfspec[j]);
        mfcep[0] = (float)(mfcep[0] / (double)nf);
        for (i = 1; i < p->ncep; ++i) {
            mfcep[i] = 0;
            for (j = 0; j < nf; j++) {
                int beta = (j == 0) ? 1 : 2;
                mfcep[i] = (float)(mfcep[i] + mfspec[j] * fe->mel_cosine[i * nf + j] * beta);
            }
            mfcep[i] = (float)(mfcep[i] / ((double)nf * 2));
        }
    }
    /* fe_lifter (applied to whatever fe_mel_cep produced) */
    if (p->lifter)
        for (i = 0; i < p->ncep; ++i) mfcep[i] = mfcep[i] * fe->lifter[i];
    free(x); free(spec); free(mfspec);
}

/* fe_process_utt + fe_end_utt: cep [s3o_fe_n_frames][feature_dimension]; returns the frame count */
int32_t
s3o_fe_process_utt(const s3o_fe_t *fe, const int16_t *spch, int64_t nsamps, float *cep)
{
    const int32_t n = s3o_fe_n_frames(fe, nsamps), D = fe->feature_dimension;
    const int64_t full = nsamps < fe->frame_size ? 0 : 1 + (nsamps - fe->frame_size) / fe->frame_shift;
    int32_t i;
    for (i = 0; i < n; i++) {
        const int64_t s0 = (int64_t)i * fe->frame_shift;
        const int32_t len = (i < full) ? fe->frame_size : (int32_t)(nsamps - s0);
        one_frame(fe, spch + s0, len, s0 > 0 ? spch[s0 - 1] : 0, cep + (size_t)i * D);
    }
    return n;
}
