/*
 * s3a_fe.hip -- the MFCC front end on the device: 16-bit samples -> cepstra, the step in front of
 * feat_s2mfc2feat (second half).
 *
 * Reference (sphinxbase/src/libsphinxbase/fe, floating-point build): fe_init_auto_r
 * (fe_interface.c:212-283), fe_process_utt + fe_end_utt (:470-502) over fe_read_frame /
 * fe_shift_frame / fe_spch_to_frame (fe_sigproc.c:570-643), fe_write_frame (:1097-1106) =
 * fe_spec_magnitude (:891-934, with fe_fft_real :792-889) + fe_mel_spec (:936-966) + fe_mel_cep
 * (:968-1014: fe_spec2cep :1016-1042 / fe_dct2 :1044-1067 / fe_dct3 :1082-1095) + fe_lifter
 * (:1069-1080); tables: fe_create_hamming (:516-532), fe_create_twiddle (:645-667),
 * fe_build_melfilters (:303-427), fe_compute_melcosine (:429-466).
 *
 * One workgroup per frame; the frame lives in LDS as float64.  The streaming reference is, seen
 * over a whole utterance, a pure function of the samples: frame i = samples [i * shift, i * shift
 * + frame_size), the carried pre-emphasis "prior" is the sample in front of the frame, and
 * fe_end_utt's last frame is the zero-padded rest -- so all frames are independent.  Every
 * arithmetic step keeps the reference's types and ORDER (float64 signal path with the reference's
 * own real-FFT butterfly schedule, no contraction; mel and cepstral sums are sequential chains, one
 * lane each; cepstra accumulate through float32 roundings), so the only place a result can differ
 * from the CPU's is log(): libm's and the device library's float64 log are both within 1 ulp of the
 * true value but not of each other, and that ulp survives the float32 rounding of a cepstrum in
 * roughly one value per 10^5 (tests/test_gpu_fe.py states the tolerance and counts them).
 * The tables are built on the host with libm exactly as the reference builds them.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "s3a_device.h"

#pragma clang fp contract(off)

#define FE_THREADS 256

struct s3a_fe_s {
    s3a_fe_params_t p;
    int32_t fft_order, frame_shift, frame_size, out_dim, n_coeffs;
    /* device tables */
    double *d_hamming, *d_cc, *d_ss;
    int16_t *d_spec_start, *d_filt_start, *d_filt_width;
    float *d_filt, *d_cos, *d_lifter;
    float sqrt_inv_n, sqrt_inv_2n;
    /* scratch for the host-pointer entry point */
    int16_t *d_spch; size_t spch_cap;
    float *d_cep; size_t cep_cap;
    hipStream_t stream;
};

struct FeDev {
    int32_t nfft, order, shift, fsize, nfilt, ncep, out_dim, transform, logspec, remove_dc, has_lifter;
    float alpha, sqrt_inv_n, sqrt_inv_2n;
    const double *hamming, *cc, *ss;
    const int16_t *spec_start, *filt_start, *filt_width;
    const float *filt, *cosine, *lifter;
};

__device__ __forceinline__ uint32_t
bit_reverse(uint32_t v, int32_t bits)
{
    return __brev(v) >> (32 - bits);
}

/* LDS: x[nfft] | spec[nfft / 2 + 1] | mfspec[nfilt] (float64) | c[ncep] (float32) */
__global__ void __launch_bounds__(FE_THREADS)
k_fe_frames(FeDev fe, const int16_t *__restrict__ spch, long long nsamps, int32_t n_full, int32_t n_frames,
            float *__restrict__ cep)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fe_smem[];
    double *x = (double *)fe_smem;
    double *spec = x + fe.nfft;
    double *mfspec = spec + (fe.nfft / 2 + 1);
    float *c = (float *)(mfspec + fe.nfilt);
    const int32_t f = blockIdx.x, tid = threadIdx.x, n = fe.nfft, fs = fe.fsize;
    const long long s0 = (long long)f * fe.shift;
    const int32_t len = f < n_full ? fs : (int32_t)(nsamps - s0);

    /* fe_spch_to_frame: pre-emphasis (prior = the sample in front of the frame), zero padding */
    for (int32_t i = tid; i < n; i += FE_THREADS) {
        double v = 0.0;
        if (i < len) {
            v = (double)spch[s0 + i];
            if (fe.alpha != 0.0f) {
                const int16_t prev = (s0 + i > 0) ? spch[s0 + i - 1] : (int16_t)0;
                v = v - (double)prev * fe.alpha;
            }
        }
        x[i] = v;
    }
    __syncthreads();
    if (fe.remove_dc) {                     /* the mean is a float64 sum in sample order */
        __shared__ double s_mean;
        if (tid == 0) {
            double m = 0;
            for (int32_t i = 0; i < fs; i++) m += x[i];
            s_mean = m / fs;
        }
        __syncthreads();
        for (int32_t i = tid; i < fs; i += FE_THREADS) x[i] -= s_mean;
        __syncthreads();
    }
    /* fe_hamming_window: symmetric halves (an odd frame's middle sample stays as it is) */
    for (int32_t i = tid; i < fs / 2; i += FE_THREADS) {
        const double w = fe.hamming[i];
        x[i] = x[i] * w;
        x[fs - 1 - i] = x[fs - 1 - i] * w;
    }
    __syncthreads();

    /* fe_fft_real: bit reversal, the 2-point butterflies, then stages 1 .. order-1; within a stage
     * every butterfly owns its four points, so the stage is one parallel step */
    for (int32_t i = tid; i < n; i += FE_THREADS) {
        const int32_t j = (int32_t)bit_reverse((uint32_t)i, fe.order);
        if (i < j) { const double t = x[j]; x[j] = x[i]; x[i] = t; }
    }
    __syncthreads();
    for (int32_t i = 2 * tid; i < n; i += 2 * FE_THREADS) {
        const double a = x[i], b = x[i + 1];
        x[i] = a + b;
        x[i + 1] = a - b;
    }
    __syncthreads();
    for (int32_t k = 1; k < fe.order; k++) {
        const int32_t half = 1 << k, quarter = 1 << (k - 1), span = 1 << (k + 1);
        for (int32_t t = tid; t < n / 4; t += FE_THREADS) {
            const int32_t grp = t / quarter, j = t - grp * quarter, i = grp * span;
            if (j == 0) {
                const double a = x[i], b = x[i + half];
                x[i] = a + b;
                x[i + half] = a - b;
                x[i + half + quarter] = -x[i + half + quarter];
            }
            else {
                const int32_t i1 = i + j, i2 = i + half - j, i3 = i + half + j, i4 = i + span - j;
                const double cc = fe.cc[j << (fe.order - (k + 1))], ss = fe.ss[j << (fe.order - (k + 1))];
                const double x3 = x[i3], x4 = x[i4], x2 = x[i2], x1 = x[i1];
                const double t1 = x3 * cc + x4 * ss;
                const double t2 = x3 * ss - x4 * cc;
                x[i4] = x2 - t2;
                x[i3] = -x2 - t2;
                x[i2] = x1 - t1;
                x[i1] = x1 + t1;
            }
        }
        __syncthreads();
    }
    /* fe_spec_magnitude */
    for (int32_t j = tid; j <= n / 2; j += FE_THREADS)
        spec[j] = (j == 0) ? x[0] * x[0] : x[j] * x[j] + x[n - j] * x[n - j];
    __syncthreads();
    /* fe_mel_spec + the log of fe_mel_cep: a filter per lane, its sum in bin order */
    for (int32_t i = tid; i < fe.nfilt; i += FE_THREADS) {
        const int32_t s = fe.spec_start[i], fstart = fe.filt_start[i], w = fe.filt_width[i];
        double m = 0;
        for (int32_t j = 0; j < w; j++) m += spec[s + j] * fe.filt[fstart + j];
        mfspec[i] = m > 0 ? log(m) : -10.0;
    }
    __syncthreads();
    float *out = cep + (size_t)f * fe.out_dim;
    const int32_t nf = fe.nfilt;
    if (fe.logspec == 1) {
        for (int32_t i = tid; i < fe.out_dim; i += FE_THREADS) {
            const float v = (float)mfspec[i];
            out[i] = (fe.has_lifter && i < fe.ncep) ? v * fe.lifter[i] : v;     /* (fe_lifter runs whatever the output is) */
        }
        return;
    }
    /* a cepstral coefficient per lane; the accumulator is float32 storage: each += rounds */
    for (int32_t i = tid; i < fe.ncep; i += FE_THREADS) {
        float v;
        if (fe.logspec == 2 || fe.transform != 0) {                     /* fe_dct2 */
            if (i == 0) {
                v = (float)mfspec[0];
                for (int32_t j = 1; j < nf; j++) v = (float)(v + mfspec[j]);
                v = v * ((fe.transform == 2 && fe.logspec != 2) ? fe.sqrt_inv_2n : fe.sqrt_inv_n);
            }
            else {
                v = 0;
                for (int32_t j = 0; j < nf; j++) v = (float)(v + mfspec[j] * fe.cosine[i * nf + j]);
                v = v * fe.sqrt_inv_2n;
            }
        }
        else {                                                          /* fe_spec2cep */
            if (i == 0) {
                v = (float)(mfspec[0] / 2);
                for (int32_t j = 1; j < nf; j++) v = (float)(v + mfspec[j]);
                v = (float)(v / (double)nf);
            }
            else {
                v = 0;
                for (int32_t j = 0; j < nf; j++) v = (float)(v + mfspec[j] * fe.cosine[i * nf + j] * (j == 0 ? 1 : 2));
                v = (float)(v / ((double)nf * 2));
            }
        }
        if (fe.logspec == 2) c[i] = v;
        else out[i] = fe.has_lifter ? v * fe.lifter[i] : v;
    }
    if (fe.logspec != 2) return;
    __syncthreads();
    /* -smoothspec: fe_dct3 of the cepstra, a filter per lane */
    for (int32_t i = tid; i < nf; i += FE_THREADS) {
        double m = c[0] * 0.707106781186548;            /* float32 x the float64 constant SQRT_HALF */
        for (int32_t j = 1; j < fe.ncep; j++) m += c[j] * fe.cosine[j * nf + i];    /* float32 product */
        m = m * fe.sqrt_inv_2n;
        float v = (float)m;
        /* (fe_lifter runs over the first ncep outputs whatever they are) */
        if (fe.has_lifter && i < fe.ncep) v = v * fe.lifter[i];
        out[i] = v;
    }
}

/* ------------------------------------------------------------------ */
/* host                                                                */
/* ------------------------------------------------------------------ */
extern "C" void
s3a_fe_default_params(s3a_fe_params_t *p)
{
    /* sphinxbase/include/sphinxbase/fe.h:100-215 (waveform_to_cepstral_command_line_macro) */
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->samprate = 16000.0f; p->frate = 100; p->wlen = 0.025625f; p->alpha = 0.97f;
    p->ncep = 13; p->nfft = 512; p->nfilt = 40; p->lowerf = 133.33334f; p->upperf = 6855.4976f;
    p->transform = S3A_FE_LEGACY; p->round_filters = 1; p->unit_area = 1;
}

/* -warp_type / -warp_params (fe_warp.c; the three shapes: fe_warp_inverse_linear.c:138-170, fe_warp_affine.c:139-170,
 * fe_warp_piecewise_linear.c:127-215): float32 arithmetic as there */
struct FeWarp { int32_t type; float a, b, f0, f1; };
static FeWarp
make_warp(const s3a_fe_params_t *p)
{
    FeWarp w = { 0, 0.0f, 0.0f, 0.0f, 0.0f };
    if (p->warp_type == S3A_FE_WARP_NONE || p->warp_params[0] == 0.0f) return w;          /* (slope zero: "warping not applied") */
    w.type = p->warp_type; w.a = p->warp_params[0]; w.b = p->warp_params[1];
    if (w.type == S3A_FE_WARP_PIECEWISE) {
        const float nyq = p->samprate / 2;
        if (w.b < p->samprate) {
            if (w.b == 0) w.b = p->samprate * 0.85f;
            w.f0 = (nyq - w.a * w.b) / (nyq - w.b);
            w.f1 = nyq * w.b * (w.a - 1.0f) / (nyq - w.b);
        }
    }
    return w;
}
static float
unwarped_to_warped(const FeWarp &w, float x)
{
    float t;
    switch (w.type) {
    case S3A_FE_WARP_INVERSE: return x / w.a;
    case S3A_FE_WARP_AFFINE: t = x * w.a; t += w.b; return t;
    case S3A_FE_WARP_PIECEWISE: return x < w.b ? x * w.a : w.f0 * x + w.f1;
    default: return x;
    }
}
static float
warped_to_unwarped(const FeWarp &w, float x)
{
    float t;
    switch (w.type) {
    case S3A_FE_WARP_INVERSE: return x * w.a;
    case S3A_FE_WARP_AFFINE: t = x - w.b; t /= w.a; return t;
    case S3A_FE_WARP_PIECEWISE: if (x < w.a * w.b) return x / w.a; t = x - w.f1; t /= w.f0; return t;
    default: return x;
    }
}
/* Hz <-> mel (fe_mel / fe_melinv, fe_sigproc.c:288-301): float32 in and out, float64 inside, the warping on the Hz side */
static float hz2mel(const FeWarp &w, float hz) { const float wd = unwarped_to_warped(w, hz); return (float)(2595.0 * log10(1.0 + wd / 700.0)); }
static float mel2hz(const FeWarp &w, float mel) { const float wd = (float)(700.0 * (pow(10.0, mel / 2595.0) - 1.0)); return warped_to_unwarped(w, wd); }

template <typename T>
static int32_t
upload(T **dst, const std::vector<T> &v)
{
    const size_t n = v.size() ? v.size() : 1;
    if (hipMalloc((void **)dst, n * sizeof(T)) != hipSuccess) return S3A_EHIP;
    if (v.size() && hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return S3A_EHIP;
    return S3A_OK;
}

extern "C" s3a_fe_t *
s3a_fe_init(const s3a_fe_params_t *p)
{
    if (!p) { s3a_set_error("s3a_fe_init: NULL parameters"); return NULL; }
    /* fe_parse_general_params' checks */
    int32_t order = 0;
    for (int32_t j = p->nfft; j > 1; j >>= 1, order++)
        if ((j % 2) != 0 || p->nfft <= 0) { s3a_set_error("s3a_fe_init: -nfft must be a power of 2 (is %d)", p->nfft); return NULL; }
    if (p->nfft < 4 || p->nfft > 8192) { s3a_set_error("s3a_fe_init: -nfft %d outside 4..8192", p->nfft); return NULL; }
    if (p->nfft < (int)(p->wlen * p->samprate)) {
        s3a_set_error("s3a_fe_init: -nfft must be >= the frame size (%d samples)", (int)(p->wlen * p->samprate));
        return NULL;
    }
    if (p->transform < 0 || p->transform > 2 || p->logspec < 0 || p->logspec > 2 || p->frate <= 0
        || p->nfilt <= 0 || p->nfilt > 1024 || p->ncep <= 0 || p->ncep > p->nfilt) {
        s3a_set_error("s3a_fe_init: bad -transform / -logspec / -frate / -nfilt / -ncep");
        return NULL;
    }
    s3a_fe_t *fe = (s3a_fe_t *)calloc(1, sizeof *fe);
    fe->p = *p;
    fe->fft_order = order;
    fe->frame_shift = (int32_t)(p->samprate / (int16_t)p->frate + 0.5);
    fe->frame_size = (int32_t)(p->wlen * p->samprate + 0.5);
    if (fe->frame_size > p->nfft || fe->frame_size < 2 || fe->frame_shift < 1) {
        s3a_set_error("s3a_fe_init: frame size %d / shift %d do not fit -nfft %d", fe->frame_size, fe->frame_shift, p->nfft);
        free(fe);
        return NULL;
    }
    fe->out_dim = p->logspec ? p->nfilt : p->ncep;
    const int32_t nf = p->nfilt, nbin = p->nfft / 2 + 1;

    /* window (first half) and twiddles */
    std::vector<double> ham(fe->frame_size / 2), cc(p->nfft / 4), ss(p->nfft / 4);
    for (size_t i = 0; i < ham.size(); i++)
        ham[i] = 0.54 - 0.46 * cos(2 * M_PI * i / ((double)fe->frame_size - 1.0));
    for (size_t i = 0; i < cc.size(); i++) {
        const double a = 2 * M_PI * i / p->nfft;
        cc[i] = cos(a); ss[i] = sin(a);
    }
    /* mel filters: the three corner frequencies of every filter, then its bins and weights */
    const FeWarp warp = make_warp(p);
    float melmin = hz2mel(warp, p->lowerf), melmax = hz2mel(warp, p->upperf);
    const float melbw = (melmax - melmin) / (nf + 1);
    if (p->doublebw) {
        melmin -= melbw; melmax += melbw;
        if (mel2hz(warp, melmin) < 0 || mel2hz(warp, melmax) > p->samprate / 2) {
            s3a_set_error("s3a_fe_init: -doublebw filter edges out of range (%f .. %f)", mel2hz(warp, melmin), mel2hz(warp, melmax));
            free(fe);
            return NULL;
        }
    }
    const float fftfreq = p->samprate / (float)p->nfft;
    std::vector<float> corner((size_t)nf * 3);
    for (int32_t i = 0; i < nf; i++)
        for (int32_t j = 0; j < 3; j++) {
            float hz = mel2hz(warp, (i + (p->doublebw ? 2 * j : j)) * melbw + melmin);
            if (p->round_filters) hz = ((int)(hz / fftfreq + 0.5)) * fftfreq;
            corner[i * 3 + j] = hz;
        }
    std::vector<int16_t> sstart(nf, -1), fstart(nf, 0), fwidth(nf, 0);
    std::vector<float> coeff;
    for (int32_t i = 0; i < nf; i++) {
        const float lo = corner[i * 3], mid = corner[i * 3 + 1], hi = corner[i * 3 + 2];
        for (int32_t j = 0; j < nbin; j++) {
            const float hz = j * fftfreq;
            if (hz < lo) continue;
            if (hz > hi || j == p->nfft / 2) {
                fwidth[i] = (int16_t)(j - sstart[i]);
                break;
            }
            if (sstart[i] == -1) sstart[i] = (int16_t)j;
        }
        if (sstart[i] < 0 || fwidth[i] < 0 || sstart[i] + fwidth[i] > nbin) {
            /* (the reference would index out of its arrays here: a filter bank that does not fit the FFT) */
            s3a_set_error("s3a_fe_init: mel filter %d does not fit the spectrum (edges %g .. %g Hz)", i, lo, hi);
            free(fe);
            return NULL;
        }
        fstart[i] = (int16_t)coeff.size();
        for (int32_t j = 0; j < fwidth[i]; j++) {
            const float hz = (sstart[i] + j) * fftfreq;
            float up = (hz - lo) / (mid - lo), down = (hi - hz) / (hi - mid);
            if (p->unit_area) { up *= 2 / (hi - lo); down *= 2 / (hi - lo); }
            coeff.push_back(up < down ? up : down);
        }
    }
    fe->n_coeffs = (int32_t)coeff.size();
    /* DCT basis, normalisers, lifter */
    std::vector<float> cosine((size_t)p->ncep * nf), lift(p->ncep, 0.0f);
    const double freqstep = M_PI / nf;
    for (int32_t i = 0; i < p->ncep; i++)
        for (int32_t j = 0; j < nf; j++)
            cosine[(size_t)i * nf + j] = (float)cos(freqstep * i * (j + 0.5));
    fe->sqrt_inv_n = (float)sqrt(1.0 / nf);
    fe->sqrt_inv_2n = (float)sqrt(2.0 / nf);
    if (p->lifter)
        for (int32_t i = 0; i < p->ncep; i++)
            lift[i] = (float)(1 + p->lifter / 2 * sin(i * M_PI / p->lifter));   /* (the reference's integer lifter / 2) */

    if (upload(&fe->d_hamming, ham) || upload(&fe->d_cc, cc) || upload(&fe->d_ss, ss)
        || upload(&fe->d_spec_start, sstart) || upload(&fe->d_filt_start, fstart) || upload(&fe->d_filt_width, fwidth)
        || upload(&fe->d_filt, coeff) || upload(&fe->d_cos, cosine) || upload(&fe->d_lifter, lift)
        || hipStreamCreateWithFlags(&fe->stream, hipStreamNonBlocking) != hipSuccess) {
        s3a_set_error("s3a_fe_init: device allocation failed: %s", hipGetErrorString(hipGetLastError()));
        s3a_fe_free(fe);
        return NULL;
    }
    return fe;
}

extern "C" void
s3a_fe_free(s3a_fe_t *fe)
{
    if (!fe) return;
    (void)hipFree(fe->d_hamming); (void)hipFree(fe->d_cc); (void)hipFree(fe->d_ss);
    (void)hipFree(fe->d_spec_start); (void)hipFree(fe->d_filt_start); (void)hipFree(fe->d_filt_width);
    (void)hipFree(fe->d_filt); (void)hipFree(fe->d_cos); (void)hipFree(fe->d_lifter);
    (void)hipFree(fe->d_spch); (void)hipFree(fe->d_cep);
    if (fe->stream) (void)hipStreamDestroy(fe->stream);
    free(fe);
}

extern "C" int32_t s3a_fe_output_size(const s3a_fe_t *fe) { return fe ? fe->out_dim : S3A_EINVAL; }
extern "C" int32_t s3a_fe_frame_shift(const s3a_fe_t *fe) { return fe ? fe->frame_shift : S3A_EINVAL; }
extern "C" int32_t s3a_fe_frame_size(const s3a_fe_t *fe) { return fe ? fe->frame_size : S3A_EINVAL; }

/* frames fe_process_utt + fe_end_utt make of nsamps samples */
extern "C" int32_t
s3a_fe_n_frames(const s3a_fe_t *fe, int64_t nsamps)
{
    if (!fe || nsamps < 0) return S3A_EINVAL;
    const int64_t full = nsamps < fe->frame_size ? 0 : 1 + (nsamps - fe->frame_size) / fe->frame_shift;
    return (int32_t)(full + (nsamps - full * fe->frame_shift > 0 ? 1 : 0));
}

/* the stream the front end's launches go to by default */
extern "C" void *
s3a_fe_stream(const s3a_fe_t *fe)
{
    return fe ? (void *)fe->stream : NULL;
}

/* samples and cepstra in DEVICE memory; enqueued on `stream` (NULL: the front end's own), not synchronised */
extern "C" int32_t
s3a_fe_process_utt_dev(s3a_fe_t *fe, const int16_t *spch_dev, int64_t nsamps, float *cep_dev, int32_t max_frames,
                       int32_t *n_frames, void *stream)
{
    if (!fe || !n_frames || nsamps < 0 || (nsamps > 0 && (!spch_dev || !cep_dev))) return S3A_EINVAL;
    const int32_t n = s3a_fe_n_frames(fe, nsamps);
    *n_frames = n;
    if (n > max_frames) { s3a_set_error("s3a_fe_process_utt: %d frames, room for %d", n, max_frames); return S3A_EINVAL; }
    if (n == 0) return S3A_OK;
    const int64_t full = nsamps < fe->frame_size ? 0 : 1 + (nsamps - fe->frame_size) / fe->frame_shift;
    FeDev d;
    d.nfft = fe->p.nfft; d.order = fe->fft_order; d.shift = fe->frame_shift; d.fsize = fe->frame_size;
    d.nfilt = fe->p.nfilt; d.ncep = fe->p.ncep; d.out_dim = fe->out_dim; d.transform = fe->p.transform;
    d.logspec = fe->p.logspec; d.remove_dc = fe->p.remove_dc; d.has_lifter = fe->p.lifter != 0;
    d.alpha = fe->p.alpha; d.sqrt_inv_n = fe->sqrt_inv_n; d.sqrt_inv_2n = fe->sqrt_inv_2n;
    d.hamming = fe->d_hamming; d.cc = fe->d_cc; d.ss = fe->d_ss;
    d.spec_start = fe->d_spec_start; d.filt_start = fe->d_filt_start; d.filt_width = fe->d_filt_width;
    d.filt = fe->d_filt; d.cosine = fe->d_cos; d.lifter = fe->d_lifter;
    const size_t lds = ((size_t)fe->p.nfft + fe->p.nfft / 2 + 1 + fe->p.nfilt) * sizeof(double) + (size_t)fe->p.ncep * sizeof(float);
    hipStream_t st = stream ? (hipStream_t)stream : fe->stream;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)k_fe_frames, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_fe_frames, dim3(n), dim3(FE_THREADS), lds, st, d, spch_dev, (long long)nsamps, (int32_t)full, n, cep_dev);
    HIPCHK(hipGetLastError());
    return S3A_OK;
}

/* fe_process_utt + fe_end_utt with host buffers: cep [n_frames][s3a_fe_output_size] */
extern "C" int32_t
s3a_fe_process_utt(s3a_fe_t *fe, const int16_t *spch, int64_t nsamps, float *cep, int32_t max_frames, int32_t *n_frames)
{
    if (!fe || !n_frames || nsamps < 0 || (nsamps > 0 && (!spch || !cep))) return S3A_EINVAL;
    const int32_t n = s3a_fe_n_frames(fe, nsamps);
    *n_frames = n;
    if (n > max_frames) { s3a_set_error("s3a_fe_process_utt: %d frames, room for %d", n, max_frames); return S3A_EINVAL; }
    if (n == 0) return S3A_OK;
    int32_t rc;
    if ((rc = s3a_dev_grow((void **)&fe->d_spch, &fe->spch_cap, (size_t)nsamps * 2)) != S3A_OK) return rc;
    if ((rc = s3a_dev_grow((void **)&fe->d_cep, &fe->cep_cap, (size_t)n * fe->out_dim * 4)) != S3A_OK) return rc;
    HIPCHK(hipMemcpyAsync(fe->d_spch, spch, (size_t)nsamps * 2, hipMemcpyHostToDevice, fe->stream));
    if ((rc = s3a_fe_process_utt_dev(fe, fe->d_spch, nsamps, fe->d_cep, max_frames, n_frames, fe->stream)) != S3A_OK) return rc;
    HIPCHK(hipMemcpyAsync(cep, fe->d_cep, (size_t)n * fe->out_dim * 4, hipMemcpyDeviceToHost, fe->stream));
    HIPCHK(hipStreamSynchronize(fe->stream));
    return S3A_OK;
}
