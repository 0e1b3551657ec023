/*
 * s3a_ms.hip -- device half of the multi-stream senone scorer (-senmgau .s3cont. / .semi.).
 *
 * Reference: ms_cont_mgau_frame_eval (sphinx3 libam/ms_mgau.c:242-329) = flag the codebooks of
 * the active senones, gauden_dist (ms_gauden.c:541-644) for each, senone_eval
 * (ms_senone.c:442-490) for each active senone, best, normalise.
 *
 * Layout: codebook parameters are TRANSPOSED per (codebook, stream) to [dim][P] with
 * P = n_density rounded up to a power of two, so that the P lanes that evaluate one
 * codebook read consecutive floats for every dimension (a 32-byte sector per 8-density
 * continuous codebook); the frame's feature vector sits in LDS.
 *
 *   k_ms_mark     codebook flags from the senone mask (skipped when the mapping is 1-to-1)
 *   k_ms_dist     lane = (codebook, stream, density): float64 distance chain in dimension
 *                 order (no FMA), then the ordered top-N list -- the reference's insertion
 *                 sort keeps (distance, codeword) order, i.e. each density's slot is its
 *                 RANK, computed by every lane against its codebook's values in LDS --
 *                 floor, logmath_ln_to_log truncation
 *   k_ms_senone   thread = active senone: per stream the ordered log-add of dist - pdf,
 *                 summed over streams; block max -> atomicMax
 *   k_ms_norm     senscr -= best for the active senones
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <vector>

#include "s3a_device.h"

#define MSB 256

struct s3a_ms_dev_s {
    int32_t P;                      /* padded densities per codebook-stream */
    float *meanT, *precT;           /* per (m, f): [featlen f][P], blocks in (m, f) order */
    float *det;                     /* [m][f][P] */
    int32_t *featlen, *featoff;     /* device copies */
    int32_t *pdf, *mgau;
    uint32_t *tab;                  /* log-add table widened to u32 */
    uint32_t tab_size;
    int32_t lm_zero;
    uint8_t *sen_active, *mgau_active;
    float *feat;
    int32_t *dist, *dist_id, *scr, *best;
    int32_t *scr_h, *best_h;        /* pinned */
    hipStream_t stream;
};

struct LogAdd32 {
    const uint32_t *tab;
    uint32_t size;
    int32_t zero;
    __device__ __forceinline__ int32_t operator()(int32_t x, int32_t y) const
    {
        if (x <= zero) return y;
        if (y <= zero) return x;
        const int32_t hi = x > y ? x : y, lo = x > y ? y : x;
        const uint32_t d = (uint32_t)hi - (uint32_t)lo;
        if (d >= size) return hi;
        return hi + (int32_t)tab[d];
    }
};

__global__ void
k_ms_mark(const uint8_t *__restrict__ sen_active, const int32_t *__restrict__ mgau, int32_t n_sen,
          uint8_t *mgau_active)
{
    const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_sen && sen_active[s]) mgau_active[mgau[s]] = 1;
}

__global__ void __launch_bounds__(MSB)
k_ms_dist(int32_t n_mgau, int32_t n_feat, int32_t nd, int32_t P, int32_t veclen, int32_t topn,
          const int32_t *__restrict__ featlen, const int32_t *__restrict__ featoff,
          const float *__restrict__ meanT, const float *__restrict__ precT, const float *__restrict__ det,
          const uint8_t *__restrict__ mgau_active, const float *__restrict__ feat, double min_density,
          double inv_log_of_base, int32_t shift, int32_t *dist, int32_t *dist_id)
{
    extern __shared__ float x_s[];              /* [veclen] */
    __shared__ double dv[MSB];
    for (int32_t i = threadIdx.x; i < veclen; i += MSB) x_s[i] = feat[i];
    __syncthreads();
    const int32_t item = blockIdx.x * MSB + threadIdx.x;
    const int32_t job = item / P, d = item % P;             /* job = m * n_feat + f */
    const int32_t m = job / n_feat, f = job % n_feat;
    const bool live = job < n_mgau * n_feat && mgau_active[m] != 0;
    double dval = 0.0;
    if (live && d < nd) {
        const int32_t flen = featlen[f], fo = featoff[f];
        const size_t base = ((size_t)m * veclen + fo) * P;      /* start of the (m, f) block */
        dval = (double)det[(size_t)job * P + d];
        for (int32_t i = 0; i < flen; i++) {
            const float df = x_s[fo + i] - meanT[base + (size_t)i * P + d];     /* float32 subtract */
            const double dd = (double)df;
            const double t = (dd * dd) * (double)precT[base + (size_t)i * P + d];
            dval = dval + t;                                                    /* not an fma */
        }
    }
    dv[threadIdx.x] = dval;
    __syncthreads();
    if (!live || d >= nd) return;
    int32_t rank = d;
    if (topn < nd) {
        /* slot in the reference's ordered list: densities with a smaller distance, or an equal
         * one and a smaller codeword id, come first */
        const double *mine = dv + (threadIdx.x - d);
        rank = 0;
        for (int32_t k = 0; k < nd; k++) {
            const double o = mine[k];
            rank += (o < dval || (o == dval && k < d)) ? 1 : 0;
        }
        if (rank >= topn) return;
    }
    double v = -dval;
    if (v < min_density) v = min_density;
    const size_t o = (size_t)job * topn + rank;
    dist_id[o] = d;
    dist[o] = (int32_t)(v * inv_log_of_base) >> shift;
}

__global__ void __launch_bounds__(MSB)
k_ms_senone(int32_t n_sen, int32_t n_feat, int32_t nd, int32_t topn, const uint8_t *__restrict__ sen_active,
            const int32_t *__restrict__ mgau, const int32_t *__restrict__ pdf, const int32_t *__restrict__ dist,
            const int32_t *__restrict__ dist_id, LogAdd32 la, int32_t *scr, int32_t *best)
{
    __shared__ int32_t red[MSB / 64];
    const int32_t s = blockIdx.x * MSB + threadIdx.x;
    int32_t v = INT_MIN;
    if (s < n_sen && sen_active[s]) {
        const int32_t m = mgau[s];
        uint32_t tot = 0;
        for (int32_t f = 0; f < n_feat; f++) {
            const int32_t *fd = dist + ((size_t)m * n_feat + f) * topn, *fi = dist_id + ((size_t)m * n_feat + f) * topn;
            const int32_t *p = pdf + ((size_t)s * n_feat + f) * nd;
            int32_t fscr = (int32_t)((uint32_t)fd[0] - (uint32_t)p[fi[0]]);
            for (int32_t t = 1; t < topn; t++)
                fscr = la(fscr, (int32_t)((uint32_t)fd[t] - (uint32_t)p[fi[t]]));
            tot += (uint32_t)fscr;
        }
        v = (int32_t)tot;
        scr[s] = v;
    }
    int32_t b = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b = max(b, __shfl_xor(b, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < MSB / 64; w++) b = max(b, red[w]);
        if (b != INT_MIN) atomicMax(best, b);
    }
}

__global__ void
k_ms_norm(int32_t n_sen, const uint8_t *__restrict__ sen_active, const int32_t *__restrict__ best, int32_t *scr)
{
    const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_sen && sen_active[s]) scr[s] = (int32_t)((uint32_t)scr[s] - (uint32_t)*best);
}

/* ------------------------------------------------------------------ */
#define DM(ptr, bytes) HIPCHK(hipMalloc((void **)&(ptr), (bytes) ? (bytes) : 4))

extern "C" int32_t
s3a_ms_dev_create(s3a_ms_mgau_t *ms)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        s3a_set_error("no HIP device: libcmusphinx_amd has no CPU fallback");
        return S3A_ENODEV;
    }
    int32_t P = 1;
    while (P < ms->n_density) P <<= 1;
    if (P > MSB) {
        s3a_set_error("s3a_ms_mgau_init: %d densities per codebook exceed the kernel's %d", ms->n_density, MSB);
        return S3A_EUNSUP;
    }
    if (ms->veclen * 4 > 48 * 1024) { s3a_set_error("s3a_ms_mgau_init: feature vector too long"); return S3A_EUNSUP; }
    s3a_ms_dev_s *dv = new s3a_ms_dev_s();
    memset(dv, 0, sizeof *dv);
    ms->dev = dv;
    dv->P = P;
    const int32_t M = ms->n_mgau, F = ms->n_feat, nd = ms->n_density, D = ms->veclen, S = ms->n_sen;
    const size_t nT = (size_t)M * D * P;
    std::vector<float> mt(nT, 0.0f), pt(nT, 0.0f), dt((size_t)M * F * P, 0.0f);
    for (int32_t m = 0; m < M; m++)
        for (int32_t f = 0; f < F; f++) {
            const size_t src = (size_t)m * nd * D + (size_t)nd * ms->featoff[f];
            const size_t dst = ((size_t)m * D + ms->featoff[f]) * P;
            for (int32_t d = 0; d < nd; d++) {
                dt[((size_t)m * F + f) * P + d] = ms->det[((size_t)m * F + f) * nd + d];
                for (int32_t i = 0; i < ms->featlen[f]; i++) {
                    mt[dst + (size_t)i * P + d] = ms->mean[src + (size_t)d * ms->featlen[f] + i];
                    pt[dst + (size_t)i * P + d] = ms->prec[src + (size_t)d * ms->featlen[f] + i];
                }
            }
        }
    HIPCHK(hipStreamCreateWithFlags(&dv->stream, hipStreamNonBlocking));
    DM(dv->meanT, nT * 4); DM(dv->precT, nT * 4); DM(dv->det, (size_t)M * F * P * 4);
    DM(dv->featlen, (size_t)F * 4); DM(dv->featoff, (size_t)(F + 1) * 4);
    DM(dv->pdf, (size_t)S * F * nd * 4); DM(dv->mgau, (size_t)S * 4);
    DM(dv->sen_active, (size_t)S); DM(dv->mgau_active, (size_t)M); DM(dv->feat, (size_t)D * 4);
    DM(dv->dist, (size_t)M * F * ms->topn * 4); DM(dv->dist_id, (size_t)M * F * ms->topn * 4);
    DM(dv->scr, (size_t)S * 4); DM(dv->best, 4);
    HIPCHK(hipHostMalloc((void **)&dv->scr_h, (size_t)S * 4));
    HIPCHK(hipHostMalloc((void **)&dv->best_h, 4));
    HIPCHK(hipMemcpy(dv->meanT, mt.data(), nT * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dv->precT, pt.data(), nT * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dv->det, dt.data(), dt.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dv->featlen, ms->featlen, (size_t)F * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dv->featoff, ms->featoff, (size_t)(F + 1) * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dv->pdf, ms->pdf, (size_t)S * F * nd * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dv->mgau, ms->mgau, (size_t)S * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dv->dist, 0, (size_t)M * F * ms->topn * 4));
    HIPCHK(hipMemset(dv->dist_id, 0, (size_t)M * F * ms->topn * 4));
    {
        uint32_t size = 0, width = 0, shift = 0;
        s3a_logmath_get_table_shape(ms->lm, &size, &width, &shift);
        dv->tab_size = size;
        dv->lm_zero = s3a_logmath_get_zero(ms->lm);
        if (size == 0) { s3a_set_error("s3a_ms_mgau_init: a logmath without an add table is not supported"); return S3A_EUNSUP; }
        std::vector<uint32_t> tab(size);
        s3a_logmath_copy_table(ms->lm, tab.data(), size);
        DM(dv->tab, (size_t)size * 4);
        HIPCHK(hipMemcpy(dv->tab, tab.data(), (size_t)size * 4, hipMemcpyHostToDevice));
    }
    return S3A_OK;
}

extern "C" void
s3a_ms_dev_destroy(s3a_ms_mgau_t *ms)
{
    s3a_ms_dev_s *dv = ms ? ms->dev : NULL;
    if (!dv) return;
    void *ptrs[] = { dv->meanT, dv->precT, dv->det, dv->featlen, dv->featoff, dv->pdf, dv->mgau, dv->tab,
                     dv->sen_active, dv->mgau_active, dv->feat, dv->dist, dv->dist_id, dv->scr, dv->best };
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (dv->scr_h) (void)hipHostFree(dv->scr_h);
    if (dv->best_h) (void)hipHostFree(dv->best_h);
    if (dv->stream) (void)hipStreamDestroy(dv->stream);
    delete dv;
    ms->dev = NULL;
}

extern "C" int32_t
s3a_ms_cont_mgau_frame_eval(s3a_ms_mgau_t *ms, const uint8_t *sen_active, int32_t *senscr, const float *feat,
                            int32_t frame, int32_t *best)
{
    (void)frame;
    if (!ms || !ms->dev || !sen_active || !senscr || !feat || !best) return S3A_EINVAL;
    s3a_ms_dev_s *dv = ms->dev;
    const int32_t M = ms->n_mgau, F = ms->n_feat, nd = ms->n_density, S = ms->n_sen, P = dv->P;
    const int32_t init = INT_MIN;
    HIPCHK(hipMemcpyAsync(dv->sen_active, sen_active, (size_t)S, hipMemcpyHostToDevice, dv->stream));
    HIPCHK(hipMemcpyAsync(dv->feat, feat, (size_t)ms->veclen * 4, hipMemcpyHostToDevice, dv->stream));
    HIPCHK(hipMemcpyAsync(dv->best, &init, 4, hipMemcpyHostToDevice, dv->stream));
    const uint8_t *cb_active = dv->sen_active;             /* ".s3cont.": codebook s <=> senone s */
    if (!ms->one_to_one) {
        HIPCHK(hipMemsetAsync(dv->mgau_active, 0, (size_t)M, dv->stream));
        hipLaunchKernelGGL(k_ms_mark, dim3((S + 255) / 256), dim3(256), 0, dv->stream, dv->sen_active, dv->mgau, S,
                           dv->mgau_active);
        cb_active = dv->mgau_active;
    }
    const int64_t items = (int64_t)M * F * P;
    hipLaunchKernelGGL(k_ms_dist, dim3((uint32_t)((items + MSB - 1) / MSB)), dim3(MSB), (size_t)ms->veclen * 4,
                       dv->stream, M, F, nd, P, ms->veclen, ms->topn, dv->featlen, dv->featoff, dv->meanT,
                       dv->precT, dv->det, cb_active, dv->feat, ms->min_density, ms->lm->inv_log_of_base,
                       ms->lm->shift, dv->dist, dv->dist_id);
    LogAdd32 la = { dv->tab, dv->tab_size, dv->lm_zero };
    hipLaunchKernelGGL(k_ms_senone, dim3((S + MSB - 1) / MSB), dim3(MSB), 0, dv->stream, S, F, nd, ms->topn,
                       dv->sen_active, dv->mgau, dv->pdf, dv->dist, dv->dist_id, la, dv->scr, dv->best);
    hipLaunchKernelGGL(k_ms_norm, dim3((S + 255) / 256), dim3(256), 0, dv->stream, S, dv->sen_active, dv->best,
                       dv->scr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(dv->scr_h, dv->scr, (size_t)S * 4, hipMemcpyDeviceToHost, dv->stream));
    HIPCHK(hipMemcpyAsync(dv->best_h, dv->best, 4, hipMemcpyDeviceToHost, dv->stream));
    HIPCHK(hipStreamSynchronize(dv->stream));
    for (int32_t s = 0; s < S; s++)
        if (sen_active[s]) senscr[s] = dv->scr_h[s];
    *best = *dv->best_h;
    return S3A_OK;
}

extern "C" int32_t
s3a_ms_mgau_get_dist(s3a_ms_mgau_t *ms, int32_t *dist, int32_t *dist_id)
{
    if (!ms || !ms->dev || !dist || !dist_id) return S3A_EINVAL;
    const size_t n = (size_t)ms->n_mgau * ms->n_feat * ms->topn * 4;
    HIPCHK(hipStreamSynchronize(ms->dev->stream));
    HIPCHK(hipMemcpy(dist, ms->dev->dist, n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(dist_id, ms->dev->dist_id, n, hipMemcpyDeviceToHost));
    return S3A_OK;
}
