/*
 * s3a_hmm.hip -- batched HMM Viterbi update (one lane per HMM).
 *
 * Replaces hmm_t / hmm_context_t (sphinx3/include/hmm.h:156-197) and
 * hmm_vit_eval with its topology-specific bodies
 * (sphinx3/src/libs3decoder/libam/hmm.c:285-412 5-state, :418-587 5-state mpx,
 *  :592-674 3-state, :677-776 3-state mpx, :779-852 any topology, :855-873
 *  dispatch) plus hmm_init / hmm_clear / hmm_enter (hmm.c:130-147, 225-250).
 *
 * Data layout (structure of arrays, N HMMs, NS = 5 state slots):
 *   score[st][i], hist[st][i] (int64: the reference's union {long; void*}),
 *   out_score[i], out_hist[i], bestscore[i], ssid[i] (non-mpx) or
 *   mpx_ssid[st][i], tmatid[i], frame[i], mpx[i]
 * so that lane i of a wave reads consecutive words for every field.  The
 * transition matrices (48 x 3 x 4 int32 = 2.3 KB for hub4) are staged in LDS;
 * senone scores are gathered through sseq[ssid][st].
 *
 * Integer semantics: all score arithmetic wraps modulo 2^32 exactly like the
 * reference's int32 adds of WORST_SCORE + senscr on x86-64; comparisons are
 * signed.  Tie-breaking follows the reference's strict '>' tests verbatim,
 * because which predecessor's history survives an exact tie is observable.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#include "s3a_device.h"
#include "s3a_vit.h"

#define NS S3A_NS
#define WORST S3A_WORST

struct s3a_hmm_batch_s {
    int32_t n, ne, n_tmat, n_sseq, n_sen;
    int32_t *score, *out_score, *bestscore, *ssid, *mpx_ssid, *tmatid, *frame, *ret;
    int64_t *hist, *out_hist;
    uint8_t *mpx;
    int32_t *tp;        /* [n_tmat][ne][ne+1] */
    int16_t *sseq;      /* [n_sseq][ne] */
    int32_t *senscr;    /* [n_sen] */
    hipStream_t stream;
};

typedef HmmRegsT<int64_t> HmmRegs;

/* hmm_vit_eval_3st_lr_mpx, hmm.c:677-776; e[st] is only read when ssid[st] != -1 */
__device__ __forceinline__ int32_t
vit3_mpx(HmmRegs &r, const int32_t *tp, const int32_t *e)
{
    int32_t s3, s2, s1, s0, t2, t1, t0, best;
    t2 = INT_MIN;
    if (r.ssid[2] == -1) s2 = t1 = WORST;
    else { s2 = add32(r.s[2], e[2]); if (s2 < WORST) s2 = WORST; t1 = add32(s2, tp[2 * 4 + 3]); }
    if (r.ssid[1] == -1) s1 = WORST;
    else { s1 = add32(r.s[1], e[1]); if (s1 < WORST) s1 = WORST; t2 = add32(s1, tp[1 * 4 + 3]); }
    if (t1 > t2) { s3 = t1; r.outh = r.h[2]; }
    else         { s3 = t2; r.outh = r.h[1]; }
    if (s3 < WORST) s3 = WORST;
    r.out = s3;
    best = s3;

    s0 = add32(r.s[0], e[0]);
    if (s0 < WORST) s0 = WORST;
    t0 = t1 = WORST;
    if (s2 != WORST) t0 = add32(s2, tp[2 * 4 + 2]);
    if (s1 != WORST) t1 = add32(s1, tp[1 * 4 + 2]);
    if (tp[0 * 4 + 2] > WORST) t2 = add32(s0, tp[0 * 4 + 2]);
    if (t0 > t1) {
        if (t2 > t0) { s2 = t2; r.h[2] = r.h[0]; r.ssid[2] = r.ssid[0]; } else s2 = t0;
    }
    else {
        if (t2 > t1) { s2 = t2; r.h[2] = r.h[0]; r.ssid[2] = r.ssid[0]; }
        else { s2 = t1; r.h[2] = r.h[1]; r.ssid[2] = r.ssid[1]; }
    }
    if (s2 < WORST) s2 = WORST;
    if (s2 > best) best = s2;
    r.s[2] = s2;

    t0 = WORST;
    if (s1 != WORST) t0 = add32(s1, tp[1 * 4 + 1]);
    t1 = add32(s0, tp[0 * 4 + 1]);
    if (t0 > t1) s1 = t0;
    else { s1 = t1; r.h[1] = r.h[0]; r.ssid[1] = r.ssid[0]; }
    if (s1 < WORST) s1 = WORST;
    if (s1 > best) best = s1;
    r.s[1] = s1;

    s0 = add32(s0, tp[0]);
    if (s0 < WORST) s0 = WORST;
    if (s0 > best) best = s0;
    r.s[0] = s0;
    return best;
}

/* three-way predecessor choice shared by the 5-state bodies:
 * candidates t0 (self), t1 (from st-1), t2 (from st-2); ties resolved as
 * hmm.c:330-345: t0 wins only on strict >, t2 beats the winner only on strict > */
/* hmm_vit_eval_5st_lr_mpx, hmm.c:418-587 */
__device__ __forceinline__ int32_t
vit5_mpx(HmmRegs &r, const int32_t *tp, const int32_t *e)
{
    int32_t s5, s4, s3, s2, s1, s0, t2, t1, t0, best;
    if (r.ssid[4] == -1) s4 = t1 = WORST;
    else { s4 = add32(r.s[4], e[4]); t1 = add32(s4, tp[4 * 6 + 5]); }
    if (r.ssid[3] == -1) s3 = t2 = WORST;
    else { s3 = add32(r.s[3], e[3]); t2 = add32(s3, tp[3 * 6 + 5]); }
    if (t1 > t2) { s5 = t1; r.outh = r.h[4]; }
    else         { s5 = t2; r.outh = r.h[3]; }
    if (s5 < WORST) s5 = WORST;
    r.out = s5;
    best = s5;

    if (r.ssid[2] == -1) s2 = t2 = WORST;
    else { s2 = add32(r.s[2], e[2]); t2 = add32(s2, tp[2 * 6 + 4]); }
    t0 = t1 = WORST;
    if (s4 != WORST) t0 = add32(s4, tp[4 * 6 + 4]);
    if (s3 != WORST) t1 = add32(s3, tp[3 * 6 + 4]);
    PICK3(s4, t0, t1, t2, (r.h[4] = r.h[3], r.ssid[4] = r.ssid[3]),
          (r.h[4] = r.h[2], r.ssid[4] = r.ssid[2]));
    if (s4 < WORST) s4 = WORST;
    if (s4 > best) best = s4;
    r.s[4] = s4;

    if (r.ssid[1] == -1) s1 = t2 = WORST;
    else { s1 = add32(r.s[1], e[1]); t2 = add32(s1, tp[1 * 6 + 3]); }
    t0 = t1 = WORST;
    if (s3 != WORST) t0 = add32(s3, tp[3 * 6 + 3]);
    if (s2 != WORST) t1 = add32(s2, tp[2 * 6 + 3]);
    PICK3(s3, t0, t1, t2, (r.h[3] = r.h[2], r.ssid[3] = r.ssid[2]),
          (r.h[3] = r.h[1], r.ssid[3] = r.ssid[1]));
    if (s3 < WORST) s3 = WORST;
    if (s3 > best) best = s3;
    r.s[3] = s3;

    s0 = add32(r.s[0], e[0]);
    t0 = t1 = WORST;
    if (s2 != WORST) t0 = add32(s2, tp[2 * 6 + 2]);
    if (s1 != WORST) t1 = add32(s1, tp[1 * 6 + 2]);
    t2 = add32(s0, tp[0 * 6 + 2]);
    PICK3(s2, t0, t1, t2, (r.h[2] = r.h[1], r.ssid[2] = r.ssid[1]),
          (r.h[2] = r.h[0], r.ssid[2] = r.ssid[0]));
    if (s2 < WORST) s2 = WORST;
    if (s2 > best) best = s2;
    r.s[2] = s2;

    t0 = WORST;
    if (s1 != WORST) t0 = add32(s1, tp[1 * 6 + 1]);
    t1 = add32(s0, tp[0 * 6 + 1]);
    if (t0 > t1) s1 = t0;
    else { s1 = t1; r.h[1] = r.h[0]; r.ssid[1] = r.ssid[0]; }
    if (s1 < WORST) s1 = WORST;
    if (s1 > best) best = s1;
    r.s[1] = s1;

    s0 = add32(s0, tp[0]);
    if (s0 < WORST) s0 = WORST;
    if (s0 > best) best = s0;
    r.s[0] = s0;
    return best;
}

/* hmm_vit_eval_anytopo, hmm.c:779-852 */
__device__ __forceinline__ int32_t
vit_any(HmmRegs &r, const int32_t *tp, const int32_t *e, int32_t ne, bool mpx)
{
    const int32_t nd = ne + 1;
    int32_t ss[NS];
    int32_t to, from, bestfrom, newscr, scr, bestscr;
    int64_t newh[NS];
    int32_t news[NS], newssid[NS];
    ss[0] = add32(r.s[0], e[0]);
    for (from = 1; from < ne; ++from) {
        ss[from] = add32(r.s[from], e[from]);
        if (ss[from] < WORST) ss[from] = WORST;
    }
    to = ne;
    scr = WORST;
    bestfrom = -1;
    for (from = to - 1; from >= 0; --from) {
        int32_t t = tp[from * nd + to];
        if (t > WORST && (newscr = add32(ss[from], t)) > scr) { scr = newscr; bestfrom = from; }
    }
    r.out = scr;
    if (bestfrom >= 0) r.outh = r.h[bestfrom];
    bestscr = scr;
    /* The reference updates states from the last to the first IN PLACE; a state's
     * history is read only from lower-numbered states, which are updated later,
     * so buffering the new values is equivalent and keeps this loop simple. */
    for (to = ne - 1; to >= 0; --to) {
        int32_t tself = tp[to * nd + to];
        scr = (tself > WORST) ? add32(ss[to], tself) : WORST;
        bestfrom = -1;
        for (from = to - 1; from >= 0; --from) {
            int32_t t = tp[from * nd + to];
            if (t > WORST && (newscr = add32(ss[from], t)) > scr) { scr = newscr; bestfrom = from; }
        }
        news[to] = scr;
        newh[to] = (bestfrom >= 0) ? r.h[bestfrom] : r.h[to];
        newssid[to] = (bestfrom >= 0 && mpx) ? r.ssid[bestfrom] : r.ssid[to];
        if (bestscr < scr) bestscr = scr;
    }
    for (to = 0; to < ne; ++to) { r.s[to] = news[to]; r.h[to] = newh[to]; r.ssid[to] = newssid[to]; }
    return bestscr;
}

__global__ void __launch_bounds__(256)
k_hmm_vit_eval(int32_t n, int32_t ne, int32_t n_tmat,
               int32_t *__restrict__ score, int64_t *__restrict__ hist,
               int32_t *__restrict__ out_score, int64_t *__restrict__ out_hist,
               int32_t *__restrict__ bestscore, const int32_t *__restrict__ ssid_g,
               int32_t *__restrict__ mpx_ssid, const int32_t *__restrict__ tmatid,
               const uint8_t *__restrict__ mpx_g, const int32_t *__restrict__ tp_g,
               const int16_t *__restrict__ sseq, const int32_t *__restrict__ senscr,
               int32_t *__restrict__ ret)
{
    extern __shared__ int32_t tp_s[];
    const int32_t nd = ne + 1, tpsz = ne * nd;
    for (int32_t i = threadIdx.x; i < n_tmat * tpsz; i += blockDim.x)
        tp_s[i] = tp_g[i];
    __syncthreads();
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;

    HmmRegs r;
    const bool mpx = mpx_g[i] != 0;
    const int32_t *tp = tp_s + tmatid[i] * tpsz;
    int32_t e[NS];
#pragma unroll
    for (int st = 0; st < NS; st++) {
        if (st < ne) {
            r.s[st] = score[(size_t)st * n + i];
            r.h[st] = hist[(size_t)st * n + i];
            r.ssid[st] = mpx ? mpx_ssid[(size_t)st * n + i] : ssid_g[i];
            /* hmm_senscr: S3_LOGPROB_ZERO for an unset multiplex state, hmm.h:223-226 */
            e[st] = (r.ssid[st] == -1) ? S3A_LOGPROB_ZERO
                                       : senscr[sseq[(size_t)r.ssid[st] * ne + st]];
        }
        else { r.s[st] = WORST; r.h[st] = -1; r.ssid[st] = -1; e[st] = WORST; }
    }
    r.out = out_score[i];
    r.outh = out_hist[i];

    int32_t best;
    if (ne == 3)
        best = mpx ? vit3_mpx(r, tp, e) : vit3(r, tp, e[0], e[1], e[2]);
    else if (ne == 5) {
        int32_t dummy = 0;
        best = mpx ? vit5_mpx(r, tp, e) : vit5(r, tp, e, dummy);
    }
    else
        best = vit_any(r, tp, e, ne, mpx);

#pragma unroll
    for (int st = 0; st < NS; st++)
        if (st < ne) {
            score[(size_t)st * n + i] = r.s[st];
            hist[(size_t)st * n + i] = r.h[st];
            if (mpx) mpx_ssid[(size_t)st * n + i] = r.ssid[st];
        }
    out_score[i] = r.out;
    out_hist[i] = r.outh;
    bestscore[i] = best;
    if (ret) ret[i] = best;
}

__global__ void
k_hmm_clear(int32_t n, int32_t ne, const int32_t *__restrict__ idx, int32_t n_idx,
            int32_t *score, int64_t *hist, int32_t *out_score, int64_t *out_hist,
            int32_t *bestscore, int32_t *frame)
{
    int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_idx) return;
    int32_t i = idx ? idx[j] : j;
    for (int32_t st = 0; st < ne; st++) {
        score[(size_t)st * n + i] = WORST;
        hist[(size_t)st * n + i] = -1;
    }
    out_score[i] = WORST;
    out_hist[i] = -1;
    bestscore[i] = WORST;
    frame[i] = -1;
}

__global__ void
k_hmm_enter(int32_t n, const int32_t *__restrict__ idx, const int32_t *__restrict__ scr,
            const int64_t *__restrict__ hid, int32_t n_idx, int32_t fr,
            int32_t *score, int64_t *hist, int32_t *frame)
{
    int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_idx) return;
    int32_t i = idx[j];
    score[i] = scr[j];          /* state 0 */
    hist[i] = hid[j];
    frame[i] = fr;
}

extern "C" s3a_hmm_batch_t *
s3a_hmm_batch_init(int32_t n_hmm, int32_t n_emit_state, const s3a_tmat_t *tmat,
                   const int16_t *sseq, int32_t n_sseq, int32_t n_sen)
{
    s3a_hmm_batch_t *b;
    size_t n = (size_t)n_hmm;
    if (n_hmm <= 0 || n_emit_state < 1 || n_emit_state > NS || !tmat || !sseq || n_sseq <= 0
        || n_sen <= 0 || tmat->n_state != n_emit_state) {
        s3a_set_error("s3a_hmm_batch_init: bad arguments (n_emit_state must be 1..5 and match the tmat)");
        return NULL;
    }
    for (int32_t i = 0; i < n_sseq * n_emit_state; i++)
        if (sseq[i] < 0 || sseq[i] >= n_sen) {
            s3a_set_error("s3a_hmm_batch_init: sseq entry %d out of range", i);
            return NULL;
        }
    b = (s3a_hmm_batch_t *)calloc(1, sizeof *b);
    b->n = n_hmm; b->ne = n_emit_state; b->n_tmat = tmat->n_tmat; b->n_sseq = n_sseq; b->n_sen = n_sen;
    size_t tpn = (size_t)tmat->n_tmat * n_emit_state * (n_emit_state + 1);
    if (hipMalloc(&b->score, 4 * NS * n) != hipSuccess || hipMalloc(&b->hist, 8 * NS * n) != hipSuccess
        || hipMalloc(&b->out_score, 4 * n) != hipSuccess || hipMalloc(&b->out_hist, 8 * n) != hipSuccess
        || hipMalloc(&b->bestscore, 4 * n) != hipSuccess || hipMalloc(&b->ssid, 4 * n) != hipSuccess
        || hipMalloc(&b->mpx_ssid, 4 * NS * n) != hipSuccess || hipMalloc(&b->tmatid, 4 * n) != hipSuccess
        || hipMalloc(&b->frame, 4 * n) != hipSuccess || hipMalloc(&b->ret, 4 * n) != hipSuccess
        || hipMalloc(&b->mpx, n) != hipSuccess || hipMalloc(&b->tp, 4 * tpn) != hipSuccess
        || hipMalloc(&b->sseq, 2 * (size_t)n_sseq * n_emit_state) != hipSuccess
        || hipMalloc(&b->senscr, 4 * (size_t)n_sen) != hipSuccess
        || hipMemcpy(b->tp, tmat->tp, 4 * tpn, hipMemcpyHostToDevice) != hipSuccess
        || hipMemcpy(b->sseq, sseq, 2 * (size_t)n_sseq * n_emit_state, hipMemcpyHostToDevice) != hipSuccess
        || hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) {
        s3a_set_error("s3a_hmm_batch_init: device allocation failed (no HIP device?)");
        s3a_hmm_batch_free(b);
        return NULL;
    }
    return b;
}

extern "C" void
s3a_hmm_batch_free(s3a_hmm_batch_t *b)
{
    if (!b) return;
    (void)hipFree(b->score); (void)hipFree(b->hist); (void)hipFree(b->out_score);
    (void)hipFree(b->out_hist); (void)hipFree(b->bestscore); (void)hipFree(b->ssid);
    (void)hipFree(b->mpx_ssid); (void)hipFree(b->tmatid); (void)hipFree(b->frame);
    (void)hipFree(b->ret); (void)hipFree(b->mpx); (void)hipFree(b->tp); (void)hipFree(b->sseq);
    (void)hipFree(b->senscr);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    free(b);
}

extern "C" int32_t
s3a_hmm_batch_clear(s3a_hmm_batch_t *b, const int32_t *idx, int32_t n)
{
    int32_t *d_idx = NULL;
    if (!b) return S3A_EINVAL;
    if (idx == NULL) n = b->n;
    if (n <= 0) return S3A_OK;
    if (idx) {
        HIPCHK(hipMalloc(&d_idx, 4 * (size_t)n));
        HIPCHK(hipMemcpyAsync(d_idx, idx, 4 * (size_t)n, hipMemcpyHostToDevice, b->stream));
    }
    hipLaunchKernelGGL(k_hmm_clear, dim3((n + 255) / 256), dim3(256), 0, b->stream, b->n, b->ne,
                       d_idx, n, b->score, b->hist, b->out_score, b->out_hist, b->bestscore,
                       b->frame);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(b->stream));
    if (d_idx) (void)hipFree(d_idx);
    return S3A_OK;
}

extern "C" int32_t
s3a_hmm_batch_setup(s3a_hmm_batch_t *b, const uint8_t *mpx, const int32_t *ssid,
                    const int32_t *tmatid)
{
    if (!b || !mpx || !ssid || !tmatid) return S3A_EINVAL;
    size_t n = (size_t)b->n;
    int32_t *ms = (int32_t *)malloc(4 * NS * n);
    for (size_t i = 0; i < n; i++) {
        if (ssid[i] < 0 || ssid[i] >= b->n_sseq || tmatid[i] < 0 || tmatid[i] >= b->n_tmat) {
            free(ms);
            s3a_set_error("s3a_hmm_batch_setup: ssid/tmatid of HMM %zu out of range", i);
            return S3A_EINVAL;
        }
        for (int st = 0; st < NS; st++)
            ms[(size_t)st * n + i] = (st == 0) ? ssid[i] : -1;     /* hmm.c:138-140 */
    }
    hipError_t e1 = hipMemcpy(b->mpx_ssid, ms, 4 * NS * n, hipMemcpyHostToDevice);
    free(ms);
    HIPCHK(e1);
    HIPCHK(hipMemcpy(b->mpx, mpx, n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->ssid, ssid, 4 * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->tmatid, tmatid, 4 * n, hipMemcpyHostToDevice));
    return s3a_hmm_batch_clear(b, NULL, b->n);
}

extern "C" int32_t
s3a_hmm_batch_enter(s3a_hmm_batch_t *b, const int32_t *idx, const int32_t *score,
                    const int64_t *histid, int32_t n, int32_t frame)
{
    int32_t *d_idx, *d_scr;
    int64_t *d_hid;
    if (!b || n < 0 || (n && (!idx || !score || !histid))) return S3A_EINVAL;
    if (n == 0) return S3A_OK;
    for (int32_t j = 0; j < n; j++)
        if (idx[j] < 0 || idx[j] >= b->n) {
            s3a_set_error("s3a_hmm_batch_enter: index out of range");
            return S3A_EINVAL;
        }
    HIPCHK(hipMalloc(&d_idx, 4 * (size_t)n));
    HIPCHK(hipMalloc(&d_scr, 4 * (size_t)n));
    HIPCHK(hipMalloc(&d_hid, 8 * (size_t)n));
    HIPCHK(hipMemcpyAsync(d_idx, idx, 4 * (size_t)n, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(d_scr, score, 4 * (size_t)n, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(d_hid, histid, 8 * (size_t)n, hipMemcpyHostToDevice, b->stream));
    hipLaunchKernelGGL(k_hmm_enter, dim3((n + 255) / 256), dim3(256), 0, b->stream, b->n, d_idx,
                       d_scr, d_hid, n, frame, b->score, b->hist, b->frame);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(b->stream));
    (void)hipFree(d_idx); (void)hipFree(d_scr); (void)hipFree(d_hid);
    return S3A_OK;
}

extern "C" int32_t
s3a_hmm_batch_vit_eval(s3a_hmm_batch_t *b, const int32_t *senscr, int32_t *ret)
{
    if (!b || !senscr) return S3A_EINVAL;
    size_t lds = 4 * (size_t)b->n_tmat * b->ne * (b->ne + 1);
    if (lds > 64 * 1024) {
        s3a_set_error("transition matrices (%zu bytes) exceed the LDS budget", lds);
        return S3A_EUNSUP;
    }
    HIPCHK(hipMemcpyAsync(b->senscr, senscr, 4 * (size_t)b->n_sen, hipMemcpyHostToDevice, b->stream));
    hipLaunchKernelGGL(k_hmm_vit_eval, dim3((b->n + 255) / 256), dim3(256), lds, b->stream, b->n,
                       b->ne, b->n_tmat, b->score, b->hist, b->out_score, b->out_hist, b->bestscore,
                       b->ssid, b->mpx_ssid, b->tmatid, b->mpx, b->tp, b->sseq, b->senscr, b->ret);
    HIPCHK(hipGetLastError());
    if (ret)
        HIPCHK(hipMemcpyAsync(ret, b->ret, 4 * (size_t)b->n, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return S3A_OK;
}

extern "C" int32_t
s3a_hmm_batch_get(const s3a_hmm_batch_t *b, int32_t *score, int64_t *hist, int32_t *out_score,
                  int64_t *out_hist, int32_t *bestscore, int32_t *mpx_ssid, int32_t *frame)
{
    if (!b) return S3A_EINVAL;
    size_t n = (size_t)b->n;
    HIPCHK(hipStreamSynchronize(b->stream));
    /* device is [st][i]; the ABI hands back [i][5] like an array of hmm_t */
    if (score || mpx_ssid) {
        int32_t *tmp = (int32_t *)malloc(4 * NS * n);
        if (score) {
            hipError_t e = hipMemcpy(tmp, b->score, 4 * NS * n, hipMemcpyDeviceToHost);
            if (e != hipSuccess) { free(tmp); HIPCHK(e); }
            for (size_t i = 0; i < n; i++)
                for (int st = 0; st < NS; st++)
                    score[i * NS + st] = (st < b->ne) ? tmp[(size_t)st * n + i] : 0;
        }
        if (mpx_ssid) {
            hipError_t e = hipMemcpy(tmp, b->mpx_ssid, 4 * NS * n, hipMemcpyDeviceToHost);
            if (e != hipSuccess) { free(tmp); HIPCHK(e); }
            for (size_t i = 0; i < n; i++)
                for (int st = 0; st < NS; st++)
                    mpx_ssid[i * NS + st] = (st < b->ne) ? tmp[(size_t)st * n + i] : -1;
        }
        free(tmp);
    }
    if (hist) {
        int64_t *tmp = (int64_t *)malloc(8 * NS * n);
        hipError_t e = hipMemcpy(tmp, b->hist, 8 * NS * n, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { free(tmp); HIPCHK(e); }
        for (size_t i = 0; i < n; i++)
            for (int st = 0; st < NS; st++)
                hist[i * NS + st] = (st < b->ne) ? tmp[(size_t)st * n + i] : 0;
        free(tmp);
    }
    if (out_score) HIPCHK(hipMemcpy(out_score, b->out_score, 4 * n, hipMemcpyDeviceToHost));
    if (out_hist) HIPCHK(hipMemcpy(out_hist, b->out_hist, 8 * n, hipMemcpyDeviceToHost));
    if (bestscore) HIPCHK(hipMemcpy(bestscore, b->bestscore, 4 * n, hipMemcpyDeviceToHost));
    if (frame) HIPCHK(hipMemcpy(frame, b->frame, 4 * n, hipMemcpyDeviceToHost));
    return S3A_OK;
}
