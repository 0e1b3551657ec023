/*
 * s3a_feat.hip -- feature computation on the device for the stream type "1s_c_d_dd": the step
 * immediately before the scoring path.
 *
 * Reference: feat_compute_utt (sphinxbase/src/libsphinxbase/feat/feat.c:1111-1123) over the padded
 * utterance that feat_s2mfc_read builds (feat.c:396-516: `win` = 3 copies of the first and of the
 * last frame, which therefore take part in the statistics), cmn() (feat/cmn.c:141-208), agc_max()
 * (feat/agc.c:109-126), feat_1s_c_d_dd_cep2feat (feat.c:726-769).
 *
 * Bit-exact: the cepstral sums are float32 sums IN FRAME ORDER, so k_feat_stats gives every
 * cepstral dimension one lane that walks the frames sequentially (13 chains of n adds; the frames
 * are staged through LDS in coalesced tiles); k_feat_apply is elementwise: normalise the seven
 * frames t-3..t+3 (indices clamped = the reference's padding), then the two difference streams.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "s3a_device.h"

#define FWIN 3
#define FTILE 64            /* frames per LDS tile in the statistics kernel */
#define FMAXC 64            /* cepstral dimensions supported */

/* stats[0..cs) = mean, [cs..2cs) = inverse standard deviation (1 if no varnorm), [2cs] = AGC maximum.
 * cmn == 2 (-cmn prior, cmn_prior.c:143-170): every frame loses the PRIOR mean stats[3 FMAXC ..) and the running sums
 * stats[4 FMAXC ..) (cmn_t.sum, filled by the caller) take the padded frames in frame order; they are left there for the caller */
__global__ void __launch_bounds__(256)
k_feat_stats(const float *__restrict__ cep, int32_t n, int32_t cs, int32_t cmn, int32_t varnorm,
             int32_t agc_max, float *stats)
{
    __shared__ float tile[FTILE][FMAXC + 1];
    const int32_t nfr = n + 2 * FWIN, i = threadIdx.x;
    float mean = 0.0f, inv = 1.0f;
    if (cmn) {
        float sum = (cmn == 2 && i < cs) ? stats[4 * FMAXC + i] : 0.0f;
        for (int32_t f0 = 0; f0 < nfr; f0 += FTILE) {
            for (int32_t k = threadIdx.x; k < FTILE * cs; k += 256) {
                const int32_t f = f0 + k / cs, d = k % cs;
                if (f < nfr) tile[k / cs][d] = cep[(size_t)min(max(f - FWIN, 0), n - 1) * cs + d];
            }
            __syncthreads();
            if (i < cs)
                for (int32_t f = 0; f < FTILE && f0 + f < nfr; f++) sum += tile[f][i];
            __syncthreads();
        }
        mean = sum / nfr;
        if (cmn == 2 && i < cs) { stats[4 * FMAXC + i] = sum; mean = stats[3 * FMAXC + i]; }
        if (varnorm && cmn != 2) {
            float var = 0.0f;
            for (int32_t f0 = 0; f0 < nfr; f0 += FTILE) {
                for (int32_t k = threadIdx.x; k < FTILE * cs; k += 256) {
                    const int32_t f = f0 + k / cs, d = k % cs;
                    if (f < nfr) tile[k / cs][d] = cep[(size_t)min(max(f - FWIN, 0), n - 1) * cs + d];
                }
                __syncthreads();
                if (i < cs)
                    for (int32_t f = 0; f < FTILE && f0 + f < nfr; f++) { const float t = tile[f][i] - mean; var += t * t; }
                __syncthreads();
            }
            inv = (float)sqrt((double)nfr / (double)var);
        }
    }
    if (i < cs) { stats[i] = mean; stats[cs + i] = inv; }
    __syncthreads();
    if (agc_max && i == 0) {
        /* maximum of the NORMALISED c0 over the padded frames (clamped copies cannot change a maximum) */
        const bool vn = cmn == 1 && varnorm;
        float mx = -INFINITY;
        for (int32_t f = 0; f < n; f++) {
            float v = cep[(size_t)f * cs];
            if (cmn) v = vn ? (v - mean) * inv : v - mean;
            mx = fmaxf(mx, v);
        }
        stats[2 * cs] = mx;
    }
    else if (i == 0) stats[2 * cs] = 0.0f;
}

__global__ void __launch_bounds__(256)
k_feat_apply(const float *__restrict__ cep, int32_t n, int32_t cs, int32_t cmn, int32_t varnorm, int32_t agc_max,
             const float *__restrict__ stats, float *feat, int32_t feat_stride)
{
    const int32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n * cs) return;
    const int32_t t = k / cs, i = k - t * cs;
    const float mean = stats[i], inv = stats[cs + i], mx = (i == 0) ? stats[2 * cs] : 0.0f;
    const bool vn = cmn == 1 && varnorm;
    float c[7];
#pragma unroll
    for (int o = -3; o <= 3; o++) {
        float v = cep[(size_t)min(max(t + o, 0), n - 1) * cs + i];
        if (cmn) v = vn ? (v - mean) * inv : v - mean;
        if (agc_max && i == 0) v = v - mx;
        c[o + 3] = v;
    }
    float *o = feat + (size_t)t * feat_stride;
    o[i] = c[3];
    o[cs + i] = c[5] - c[1];
    const float d1 = c[6] - c[2], d2 = c[4] - c[0];
    o[2 * cs + i] = d1 - d2;
}

/* cep: HOST [n][cepsize]; feat_dev: DEVICE rows of feat_stride floats (>= 3 * cepsize; the columns
 * beyond are left untouched -- pass the padded stride s3a_mgau_score_frames_dev expects) */
extern "C" int32_t
s3a_feat_1s_c_d_dd_dev(const float *cep, int32_t n_frames, int32_t cepsize, int32_t cmn_current,
                       int32_t varnorm, int32_t agc_max, float *feat_dev, int32_t feat_stride, void *stream)
{
    if (!cep || !feat_dev || n_frames <= 0 || cepsize <= 0 || cepsize > FMAXC || feat_stride < 3 * cepsize) {
        s3a_set_error("s3a_feat_1s_c_d_dd: bad arguments (1..%d cepstral dimensions)", FMAXC);
        return S3A_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    float *cep_d = NULL, *stats = NULL;
    HIPCHK(hipMalloc((void **)&cep_d, (size_t)n_frames * cepsize * 4));
    HIPCHK(hipMalloc((void **)&stats, (size_t)(2 * cepsize + 1) * 4));
    HIPCHK(hipMemcpyAsync(cep_d, cep, (size_t)n_frames * cepsize * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_feat_stats, dim3(1), dim3(256), 0, st, cep_d, n_frames, cepsize, cmn_current, varnorm,
                       agc_max, stats);
    hipLaunchKernelGGL(k_feat_apply, dim3((n_frames * cepsize + 255) / 256), dim3(256), 0, st, cep_d, n_frames,
                       cepsize, cmn_current, varnorm, agc_max, stats, feat_dev, feat_stride);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    (void)hipFree(cep_d); (void)hipFree(stats);
    return S3A_OK;
}

/* the same with the features copied back: feat [n][3 * cepsize] on the host */
extern "C" int32_t
s3a_feat_1s_c_d_dd(const float *cep, int32_t n_frames, int32_t cepsize, int32_t cmn_current, int32_t varnorm,
                   int32_t agc_max, float *feat)
{
    if (!feat || n_frames <= 0 || cepsize <= 0) return S3A_EINVAL;
    float *fd = NULL;
    HIPCHK(hipMalloc((void **)&fd, (size_t)n_frames * 3 * cepsize * 4));
    int32_t rc = s3a_feat_1s_c_d_dd_dev(cep, n_frames, cepsize, cmn_current, varnorm, agc_max, fd, 3 * cepsize, NULL);
    if (rc == S3A_OK && hipMemcpy(feat, fd, (size_t)n_frames * 3 * cepsize * 4, hipMemcpyDeviceToHost) != hipSuccess) {
        s3a_set_error("s3a_feat_1s_c_d_dd: read-back failed");
        rc = S3A_EHIP;
    }
    (void)hipFree(fd);
    return rc;
}

/*
 * 16-bit samples (host) -> the utterance's 1s_c_d_dd features RESIDENT IN HBM, ready for s3a_uttdec_decode_dev /
 * s3a_uttdec_decode_queue_dev: what utt_decode does with -adcin (sphinx3/src/libs3decoder/libAPI/utt.c:208-233) --
 * fe_start_utt + fe_process_utt (NO fe_end_utt when drop_partial_frame: the samples behind the last whole frame are
 * dropped, as that function drops them) + feat_s2mfc2feat_live(beginutt, endutt) = feat_s2mfc2feat_block_utt
 * (feat.c:1241-1265: first and last frame replicated `win` times, then feat_compute_utt) -- with the cepstra never
 * leaving the device: k_fe_frames -> k_feat_stats -> k_feat_apply on the front end's stream.  *feat_dev_out: rows of
 * *feat_stride = 4 * ceil(3 * cepsize / 4) floats, zero padded; the caller releases it with s3a_dev_free.
 */
static int32_t
audio_to_feat_dev(s3a_fe_t *fe, const int16_t *spch, int64_t nsamps, int32_t drop_partial_frame, int32_t cmn_current,
                  int32_t varnorm, int32_t agc_max, float **feat_dev_out, int32_t *n_frames, int32_t *feat_stride,
                  const float *prior_mean, float *prior_sum)
{
    if (!fe || !spch || !feat_dev_out || !n_frames || !feat_stride || nsamps <= 0) return S3A_EINVAL;
    const int32_t cs = s3a_fe_output_size(fe), fsize = s3a_fe_frame_size(fe), fshift = s3a_fe_frame_shift(fe);
    const int32_t n_all = s3a_fe_n_frames(fe, nsamps);
    const int32_t n = drop_partial_frame ? (nsamps < fsize ? 0 : (int32_t)(1 + (nsamps - fsize) / fshift)) : n_all;
    *feat_dev_out = NULL; *n_frames = n; *feat_stride = 4 * ((3 * cs + 3) / 4);
    if (cs <= 0 || cs > FMAXC) { s3a_set_error("s3a_audio_to_feat_dev: %d cepstral dimensions (1..%d)", cs, FMAXC); return S3A_EINVAL; }
    if (n <= 0) { s3a_set_error("s3a_audio_to_feat_dev: %lld samples are less than one frame of %d", (long long)nsamps, fsize); return S3A_EINVAL; }
    hipStream_t st = (hipStream_t)s3a_fe_stream(fe);
    int16_t *spch_d = NULL;
    float *cep_d = NULL, *out = NULL;
    const size_t row = (size_t)*feat_stride, out_floats = (size_t)n * row + 5 * FMAXC + 8;     /* (+ the statistics behind the rows) */
    int32_t rc = S3A_OK, k = 0;
    if (hipMalloc((void **)&spch_d, (size_t)nsamps * 2) != hipSuccess || hipMalloc((void **)&cep_d, (size_t)n_all * cs * 4) != hipSuccess
        || hipMalloc((void **)&out, out_floats * 4) != hipSuccess) {
        s3a_set_error("s3a_audio_to_feat_dev: out of device memory");
        rc = S3A_ENOMEM;
    }
    if (rc == S3A_OK && (hipMemsetAsync(out, 0, out_floats * 4, st) != hipSuccess
                         || hipMemcpyAsync(spch_d, spch, (size_t)nsamps * 2, hipMemcpyHostToDevice, st) != hipSuccess)) rc = S3A_EHIP;
    if (rc == S3A_OK) rc = s3a_fe_process_utt_dev(fe, spch_d, nsamps, cep_d, n_all, &k, (void *)st);
    if (rc == S3A_OK && prior_mean
        && (hipMemcpyAsync(out + (size_t)n * row + 3 * FMAXC, prior_mean, (size_t)cs * 4, hipMemcpyHostToDevice, st) != hipSuccess
            || hipMemcpyAsync(out + (size_t)n * row + 4 * FMAXC, prior_sum, (size_t)cs * 4, hipMemcpyHostToDevice, st) != hipSuccess)) rc = S3A_EHIP;
    if (rc == S3A_OK) {
        float *stats = out + (size_t)n * row;
        hipLaunchKernelGGL(k_feat_stats, dim3(1), dim3(256), 0, st, cep_d, n, cs, cmn_current, varnorm, agc_max, stats);
        hipLaunchKernelGGL(k_feat_apply, dim3((n * cs + 255) / 256), dim3(256), 0, st, cep_d, n, cs, cmn_current, varnorm, agc_max,
                           stats, out, *feat_stride);
        if (hipGetLastError() != hipSuccess) rc = S3A_EHIP;
    }
    if (rc == S3A_OK && prior_sum && hipMemcpyAsync(prior_sum, out + (size_t)n * row + 4 * FMAXC, (size_t)cs * 4, hipMemcpyDeviceToHost, st) != hipSuccess) rc = S3A_EHIP;
    if (hipStreamSynchronize(st) != hipSuccess && rc == S3A_OK) rc = S3A_EHIP;
    if (spch_d) (void)hipFree(spch_d);
    if (cep_d) (void)hipFree(cep_d);
    if (rc != S3A_OK) { if (out) (void)hipFree(out); return rc; }
    *feat_dev_out = out;
    return S3A_OK;
}

extern "C" int32_t
s3a_audio_to_feat_dev(s3a_fe_t *fe, const int16_t *spch, int64_t nsamps, int32_t drop_partial_frame, int32_t cmn_current,
                      int32_t varnorm, int32_t agc_max, float **feat_dev_out, int32_t *n_frames, int32_t *feat_stride)
{
    return audio_to_feat_dev(fe, spch, nsamps, drop_partial_frame, cmn_current ? 1 : 0, varnorm, agc_max, feat_dev_out, n_frames, feat_stride,
                             (const float *)NULL, (float *)NULL);
}

/* -cmn prior on whole utterances (feat_s2mfc2feat_block_utt -> feat_cmn -> cmn_prior, cmn_prior.c:143-170): every frame of the
 * padded utterance loses cmn_mean[] (the mean the decoder has learnt from EARLIER utterances), and cmn_sum[] (cmn_t.sum) takes
 * the padded frames in frame order, float32, starting from the value handed in.  The caller owns the state between utterances:
 * nframe += n_frames + 6, then cmn_prior's window shift and cmn_prior_update (cmn_prior.c:95-141). */
extern "C" int32_t
s3a_audio_to_feat_dev_prior(s3a_fe_t *fe, const int16_t *spch, int64_t nsamps, int32_t drop_partial_frame, const float *cmn_mean,
                            float *cmn_sum, int32_t agc_max, float **feat_dev_out, int32_t *n_frames, int32_t *feat_stride)
{
    if (!cmn_mean || !cmn_sum) return S3A_EINVAL;
    return audio_to_feat_dev(fe, spch, nsamps, drop_partial_frame, 2, 0, agc_max, feat_dev_out, n_frames, feat_stride, cmn_mean, cmn_sum);
}

/* feat_lda_transform (sphinxbase feat/lda.c:137-160): every row of the feature matrix times the transposed LDA matrix --
 * out[j] = sum over k, IN ORDER, of feat[k] * lda[j][k] (a float32 product and a float32 sum per term, no FMA) -- into rows of
 * the output dimension rounded up to four floats (zero padded: the stride the engines take for a model of that dimension) */
__global__ void __launch_bounds__(256)
k_feat_lda(const float *__restrict__ feat, int32_t n, int32_t stride, const float *__restrict__ lda, int32_t in_dim, int32_t out_dim,
           float *out, int32_t out_stride)
{
    __shared__ float row[4][FMAXC * 3 + 4];
    const int32_t sub = threadIdx.x >> 6, lane = threadIdx.x & 63, t = blockIdx.x * 4 + sub;
    if (t < n) for (int32_t k = lane; k < in_dim; k += 64) row[sub][k] = feat[(size_t)t * stride + k];
    __syncthreads();
    if (t >= n) return;
    for (int32_t j = lane; j < out_stride; j += 64) {
        float acc = 0.0f;
        if (j < out_dim)
            for (int32_t k = 0; k < in_dim; k++) { const float p = row[sub][k] * lda[(size_t)j * in_dim + k]; acc = acc + p; }
        out[(size_t)t * out_stride + j] = acc;
    }
}

extern "C" int32_t
s3a_feat_lda_dev(float **feat_dev, int32_t n_frames, int32_t *feat_stride, const float *lda, int32_t in_dim, int32_t out_dim, void *stream)
{
    if (!feat_dev || !*feat_dev || !feat_stride || !lda || n_frames < 1 || in_dim < 1 || in_dim > 3 * FMAXC || out_dim < 1 || out_dim > in_dim
        || *feat_stride < in_dim) {
        s3a_set_error("s3a_feat_lda_dev: bad arguments (%d x %d matrix)", out_dim, in_dim);
        return S3A_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    const int32_t os = 4 * ((out_dim + 3) / 4);
    float *lda_d = NULL, *out = NULL;
    if (hipMalloc((void **)&lda_d, (size_t)out_dim * in_dim * 4) != hipSuccess || hipMalloc((void **)&out, ((size_t)n_frames * os + 8) * 4) != hipSuccess) {
        if (lda_d) (void)hipFree(lda_d);
        s3a_set_error("s3a_feat_lda_dev: out of device memory");
        return S3A_ENOMEM;
    }
    int32_t rc = S3A_OK;
    if (hipMemcpyAsync(lda_d, lda, (size_t)out_dim * in_dim * 4, hipMemcpyHostToDevice, st) != hipSuccess) rc = S3A_EHIP;
    if (rc == S3A_OK) {
        hipLaunchKernelGGL(k_feat_lda, dim3((n_frames + 3) / 4), dim3(256), 0, st, *feat_dev, n_frames, *feat_stride, lda_d, in_dim, out_dim, out, os);
        if (hipGetLastError() != hipSuccess) rc = S3A_EHIP;
    }
    if (hipStreamSynchronize(st) != hipSuccess && rc == S3A_OK) rc = S3A_EHIP;
    (void)hipFree(lda_d);
    if (rc != S3A_OK) { (void)hipFree(out); return rc; }
    (void)hipFree(*feat_dev);
    *feat_dev = out; *feat_stride = os;
    return S3A_OK;
}
