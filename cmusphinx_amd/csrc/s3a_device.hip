/*
 * s3a_device.hip -- CDNA4 (gfx950) kernels and launchers of libcmusphinx_amd.
 *
 * Hot path: Gaussian-mixture senone scoring.
 *
 * Device layout of a model (built once in s3a_mgau_dev_create):
 *   - one LANE per Gaussian.  Gaussian index g = senone * CP + component,
 *     CP = components per senone rounded up to a power of two (<= 64), so the
 *     CP components of a senone sit in adjacent lanes of one wavefront and a
 *     64-lane wave covers 64/CP whole senones.
 *   - mean4[k][g], prec4[k][g] : float4 = dims 4k..4k+3 of Gaussian g
 *     (veclen padded to a multiple of 4 with mean = prec = 0, which adds an
 *     exact 0.0 to the distance).  For a fixed k the 64 lanes of a wave read
 *     1 KiB of consecutive memory with one global_load_dwordx4.
 *   - lrd[g], mixw[g]; padded components carry mixw = S3_LOGPROB_ZERO so the
 *     reference's own log-add ignores them (x <= lmath->zero returns y).
 *   - the log-add table (58.7 KB of uint16 at base 1.0003) is copied into LDS
 *     by every workgroup that scores more than a few frames.
 *
 * Kernel k_score_frames is MODEL-STATIONARY: a lane loads its Gaussian's
 * 2 x 40 parameters into VGPRs once and then streams frames past them; the
 * feature vectors of the workgroup's frame chunk are staged in LDS and read
 * back with wave-uniform ds_read_b128 (one LDS broadcast per 4 dimensions).
 * Bit-exact arithmetic (mode S3A_GMM_EXACT) follows cont_mgau.c:1058-1063:
 * float32 subtract, widen, two float64 multiplies, one float64 subtract --
 * never an FMA (this file is compiled with -ffp-contract=off and the exact
 * path uses no fma builtin) -- then (int32)(f * dval) + mixw and a log-add
 * over the components IN COMPONENT ORDER (logmath_add is not associative).
 * Frames are processed FB = 8 at a time: 8 independent accumulator chains per
 * lane give the ILP that hides the float64 latency, and the 8 x CP scores of a
 * senone are transposed through a small per-wave LDS tile so that each lane
 * runs ONE frame's sequential log-add chain (CP table look-ups per lane per 8
 * frames instead of CP per frame).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <vector>

#include "s3a_device.h"
#include "s3a_structs.h"

#pragma clang fp contract(off)

/* ------------------------------------------------------------------ */
/* device objects                                                      */
/* ------------------------------------------------------------------ */
#define FB 8                /* frames per inner group */
#define HYB_MAX_PIECES 10    /* 16-byte pieces of the repacked log-add table per lane of k_score_frame_sync (40 KB) */
#define GPAD_ALIGN 1024     /* Gaussians padded to a whole number of the largest workgroup */

static int32_t
pow2_ceil(int32_t v)
{
    int32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

extern "C" int32_t
s3a_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

extern "C" int32_t
s3a_set_device(int32_t ordinal)
{
    HIPCHK(hipSetDevice(ordinal));
    return S3A_OK;
}

extern "C" void *
s3a_dev_malloc(size_t nbytes)
{
    void *p = NULL;
    if (hipMalloc(&p, nbytes ? nbytes : 1) != hipSuccess) {
        s3a_set_error("hipMalloc(%zu) failed", nbytes);
        return NULL;
    }
    return p;
}

extern "C" int32_t s3a_dev_free(void *p) { HIPCHK(hipFree(p)); return S3A_OK; }
extern "C" int32_t s3a_dev_upload(void *d, const void *s, size_t n)
{
    HIPCHK(hipMemcpy(d, s, n, hipMemcpyHostToDevice));
    return S3A_OK;
}
extern "C" int32_t s3a_dev_download(void *d, const void *s, size_t n)
{
    HIPCHK(hipMemcpy(d, s, n, hipMemcpyDeviceToHost));
    return S3A_OK;
}
extern "C" int32_t s3a_dev_sync(void) { HIPCHK(hipDeviceSynchronize()); return S3A_OK; }

extern "C" int32_t
s3a_mgau_dev_create(s3a_mgau_model_t *g)
{
    struct s3a_mgau_dev_s *d;
    int ndev = 0;
    hipDeviceProp_t prop;
    int dev = 0;

    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        s3a_set_error("no HIP device: libcmusphinx_amd has no CPU fallback");
        return S3A_ENODEV;
    }
    if (g->max_comp > 64) {
        s3a_set_error("%d components per senone: more than 64 is not supported", g->max_comp);
        return S3A_EUNSUP;
    }
    d = (struct s3a_mgau_dev_s *)calloc(1, sizeof *d);
    g->dev = d;
    d->S = g->n_mgau;
    d->C = g->max_comp;
    d->CP = pow2_ceil(g->max_comp);
    d->D = g->veclen;
    d->D4 = (g->veclen + 3) / 4;
    d->G = d->S * d->CP;
    d->Gpad = ((d->G + GPAD_ALIGN - 1) / GPAD_ALIGN) * GPAD_ALIGN;
    HIPCHK(hipGetDevice(&dev));
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    d->n_cu = prop.multiProcessorCount;

    {
        size_t nv = (size_t)d->D4 * d->Gpad;
        std::vector<float4> hm(nv), hp(nv);
        std::vector<float> hl(d->Gpad, 0.0f);
        std::vector<int32_t> hw(d->Gpad, S3A_LOGPROB_ZERO);
        memset(hm.data(), 0, nv * sizeof(float4));
        memset(hp.data(), 0, nv * sizeof(float4));
        for (int32_t s = 0; s < d->S; s++)
            for (int32_t c = 0; c < g->n_comp[s]; c++) {
                size_t gi = (size_t)s * d->CP + c;
                const float *m = g->mean + ((size_t)s * d->C + c) * d->D;
                const float *p = g->prec + ((size_t)s * d->C + c) * d->D;
                for (int32_t i = 0; i < d->D; i++) {
                    ((float *)&hm[(size_t)(i >> 2) * d->Gpad + gi])[i & 3] = m[i];
                    ((float *)&hp[(size_t)(i >> 2) * d->Gpad + gi])[i & 3] = p[i];
                }
                hl[gi] = g->lrd[(size_t)s * d->C + c];
                hw[gi] = g->mixw[(size_t)s * d->C + c];
            }
        HIPCHK(hipMalloc(&d->mean4, nv * sizeof(float4)));
        HIPCHK(hipMalloc(&d->prec4, nv * sizeof(float4)));
        HIPCHK(hipMalloc(&d->lrd, d->Gpad * sizeof(float)));
        HIPCHK(hipMalloc(&d->mixw, d->Gpad * sizeof(int32_t)));
        HIPCHK(hipMemcpy(d->mean4, hm.data(), nv * sizeof(float4), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d->prec4, hp.data(), nv * sizeof(float4), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d->lrd, hl.data(), d->Gpad * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d->mixw, hw.data(), d->Gpad * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    {
        const s3a_logmath_t *lm = g->lm;
        d->lm_zero = lm->zero;
        d->tab_size = lm->table_size;
        if (lm->table == NULL) {
            s3a_set_error("a logmath without an add-table is not supported on the device");
            return S3A_EUNSUP;
        }
        if (lm->width <= 2) {
            std::vector<uint16_t> t(((size_t)lm->table_size + 7) & ~(size_t)7, 0);  /* 16-byte multiple */
            for (uint32_t i = 0; i < lm->table_size; i++) t[i] = (uint16_t)lm->table[i];
            HIPCHK(hipMalloc(&d->tab16, t.size() * 2));
            HIPCHK(hipMemcpy(d->tab16, t.data(), t.size() * 2, hipMemcpyHostToDevice));
        }
        else {
            HIPCHK(hipMalloc(&d->tab32, (size_t)lm->table_size * 4));
            HIPCHK(hipMemcpy(d->tab32, lm->table, (size_t)lm->table_size * 4,
                             hipMemcpyHostToDevice));
        }
    }
    /* the table repacked for the frame-synchronous pass: 16-bit entries while the values need them, 8-bit ones
     * from the first multiple-of-1024 index whose value (and, the table being monotone, every later one) fits a byte */
    d->hyb_ok = 0;
    if (d->tab16 != NULL) {
        const s3a_logmath_t *lm = g->lm;
        uint32_t head = 0;
        while (head < lm->table_size && lm->table[head] > 255) head += 1024;
        if (head > lm->table_size) head = (lm->table_size + 7) & ~7u;
        bool ok = true;
        for (uint32_t i = head; i < lm->table_size && ok; i++) ok = lm->table[i] <= 255;
        const size_t bytes = (((size_t)head * 2 + (lm->table_size > head ? lm->table_size - head : 0)) + 15) & ~(size_t)15;
        if (ok && bytes <= 40 * 1024) {
            std::vector<uint8_t> pk(bytes, 0);
            for (uint32_t i = 0; i < head && i < lm->table_size; i++) ((uint16_t *)pk.data())[i] = (uint16_t)lm->table[i];
            for (uint32_t i = head; i < lm->table_size; i++) pk[(size_t)head * 2 + (i - head)] = (uint8_t)lm->table[i];
            HIPCHK(hipMalloc(&d->hyb_tab, bytes));
            HIPCHK(hipMemcpy(d->hyb_tab, pk.data(), bytes, hipMemcpyHostToDevice));
            d->hyb_ok = 1; d->hyb_head = (int32_t)head; d->hyb_bytes = (int32_t)bytes;
        }
    }
    HIPCHK(hipMalloc(&d->bstidx, d->S * sizeof(int32_t)));
    HIPCHK(hipMalloc(&d->bstscr, d->S * sizeof(int32_t)));
    HIPCHK(hipMalloc(&d->updatetime, d->S * sizeof(int32_t)));
    HIPCHK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    return s3a_mgau_reset_state(g);
}

/*
 * An adapted model: the Gaussians' means, precisions (1 / (2 sigma^2)) and log determinant terms replaced in place,
 * host AoS [n_mgau][max_comp][veclen] / [n_mgau][max_comp] as mgau_model_t holds them after adapt_set_mllr
 * (libam/adaptor.c:106-170: means + variances reloaded, mllr_norm_mgau, variance floor, mgau_precomp -- all of it the
 * reference's own host code; the replacement backend takes the result).  The mixture weights and the senones' component
 * counts do not change (checked by the caller against s3a_mgau_n_comp).  Every scorer / engine built on this model sees
 * the new parameters from its next launch on (they read the model's device arrays); not to be called while one runs.
 */
extern "C" int32_t
s3a_mgau_set_params(s3a_mgau_model_t *g, const float *mean, const float *prec, const float *lrd)
{
    if (!g || !g->dev || !mean || !prec || !lrd) return S3A_EINVAL;
    struct s3a_mgau_dev_s *d = g->dev;
    const size_t nv = (size_t)d->D4 * d->Gpad;
    std::vector<float4> hm(nv), hp(nv);
    std::vector<float> hl(d->Gpad, 0.0f);
    memset(hm.data(), 0, nv * sizeof(float4));
    memset(hp.data(), 0, nv * sizeof(float4));
    memcpy(g->mean, mean, (size_t)d->S * d->C * d->D * sizeof(float));
    memcpy(g->prec, prec, (size_t)d->S * d->C * d->D * sizeof(float));
    memcpy(g->lrd, lrd, (size_t)d->S * d->C * sizeof(float));
    for (int32_t s = 0; s < d->S; s++)
        for (int32_t c = 0; c < g->n_comp[s]; c++) {
            const size_t gi = (size_t)s * d->CP + c;
            const float *m = g->mean + ((size_t)s * d->C + c) * d->D, *p = g->prec + ((size_t)s * d->C + c) * d->D;
            for (int32_t i = 0; i < d->D; i++) {
                ((float *)&hm[(size_t)(i >> 2) * d->Gpad + gi])[i & 3] = m[i];
                ((float *)&hp[(size_t)(i >> 2) * d->Gpad + gi])[i & 3] = p[i];
            }
            hl[gi] = g->lrd[(size_t)s * d->C + c];
        }
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(d->mean4, hm.data(), nv * sizeof(float4), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d->prec4, hp.data(), nv * sizeof(float4), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d->lrd, hl.data(), d->Gpad * sizeof(float), hipMemcpyHostToDevice));
    return S3A_OK;
}

extern "C" void
s3a_mgau_dev_destroy(s3a_mgau_model_t *g)
{
    struct s3a_mgau_dev_s *d = g->dev;
    if (!d)
        return;
    hipFree(d->mean4); hipFree(d->prec4); hipFree(d->lrd); hipFree(d->mixw);
    hipFree(d->tab16); hipFree(d->tab32); hipFree(d->hyb_tab);
    hipFree(d->bstidx); hipFree(d->bstscr); hipFree(d->updatetime);
    hipFree(d->feat_buf); hipFree(d->scr_buf); hipFree(d->best_buf);
    if (d->stream) hipStreamDestroy(d->stream);
    if (d->ev0) { hipEventDestroy(d->ev0); hipEventDestroy(d->ev1); }
    free(d);
    g->dev = NULL;
}

extern "C" int32_t
s3a_mgau_set_precision(s3a_mgau_model_t *g, int32_t mode)
{
    if (mode != S3A_GMM_EXACT && mode != S3A_GMM_FAST)
        return S3A_EINVAL;
    g->precision = mode;
    return S3A_OK;
}

/* ------------------------------------------------------------------ */
/* device helpers                                                      */
/* ------------------------------------------------------------------ */
/* ------------------------------------------------------------------ */
/* k_score_frames: all senones x a chunk of frames                     */
/* ------------------------------------------------------------------ */
/*
 * grid.x enumerates (Gaussian tile, frame chunk) with the XCD-aware mapping
 * described at the launcher.  LDS: [features of the chunk: fpc x DP floats]
 * [per-wave transpose tiles: FB x 65 ints] [table (if TAB_LDS)].
 */
template <int CP, bool EXACT, bool TAB_LDS, int NT>
__global__ void __launch_bounds__(NT)
k_score_frames(const float4 *__restrict__ mean4, const float4 *__restrict__ prec4,
               const float *__restrict__ lrd, const int32_t *__restrict__ mixw_g,
               const uint16_t *__restrict__ tab_g, uint32_t tab_size, int32_t lm_zero,
               double f, double distfloor,
               const float *__restrict__ feat, int32_t feat_stride, int32_t veclen,
               int32_t n_frames, int32_t fpc, int32_t n_chunks, int32_t n_tiles,
               int32_t *__restrict__ senscr, int32_t S, int32_t Gpad)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int DP = D4MAIN * 4;
    typedef typename Acc<EXACT>::T acc_t;

    /* XCD-aware decode of blockIdx.x: all chunks of tile t run on XCD t % 8 */
    int32_t b = blockIdx.x;
    int32_t xcd = b & 7, r = b >> 3;
    int32_t chunk = r % n_chunks;
    int32_t tile = (r / n_chunks) * 8 + xcd;
    if (tile >= n_tiles)
        return;

    const int32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t g = tile * NT + tid;
    const int32_t f0 = chunk * fpc;
    const int32_t nf = min(fpc, n_frames - f0);

    /* ---- LDS carve: [features fpc x DP f32][transpose tiles][table] ---- */
    float *xs = (float *)smem;                                  /* 16-byte aligned rows */
    size_t off = (size_t)fpc * DP * sizeof(float);
    int32_t *tr = (int32_t *)(smem + off) + wave * (FB * 65);   /* [FB][65] per wave */
    off += (size_t)(NT / 64) * FB * 65 * sizeof(int32_t);
    off = (off + 15) & ~(size_t)15;
    uint16_t *tab_s = (uint16_t *)(smem + off);

    /* ---- stage table + features ---- */
    if (TAB_LDS) {
        /* 16-byte copies; tab_g is padded to a multiple of 8 entries by the host */
        const uint4 *src = (const uint4 *)tab_g;
        uint4 *dst = (uint4 *)tab_s;
        int32_t n16 = (int32_t)((tab_size * 2 + 15) >> 4);
        for (int32_t i = tid; i < n16; i += NT)
            dst[i] = src[i];
    }
    for (int32_t i = tid; i < nf * DP; i += NT) {
        int32_t fr = i / DP, k = i - fr * DP;
        xs[i] = (k < veclen) ? feat[(size_t)(f0 + fr) * feat_stride + k] : 0.0f;
    }

    /* ---- this lane's Gaussian, resident in VGPRs for the whole chunk ---- */
    float4 M[D4MAIN], P[D4MAIN];
#pragma unroll
    for (int k = 0; k < D4MAIN; k++) {
        M[k] = mean4[(size_t)k * Gpad + g];
        P[k] = prec4[(size_t)k * Gpad + g];
    }
    const acc_t lrd_g = (acc_t)lrd[g];
    const int32_t mixw = mixw_g[g];
    LogAdd la;
    la.tab = TAB_LDS ? tab_s : tab_g;
    la.size = tab_size;
    la.zero = lm_zero;
    __syncthreads();

    const int32_t c = lane & (CP - 1);          /* component == frame slot in the transposed phase */
    const int32_t sl = lane / CP;               /* senone within the wave */
    const int32_t sen = g / CP;
    const float4 *xs4 = (const float4 *)xs;

    int32_t fr = 0;
    /* ---- groups of FB frames ---- */
    for (; fr + FB <= nf; fr += FB) {
        acc_t a[FB];
#pragma unroll
        for (int j = 0; j < FB; j++) a[j] = lrd_g;
#pragma unroll
        for (int k = 0; k < D4MAIN; k++) {
#pragma unroll
            for (int j = 0; j < FB; j++) {
                float4 x = xs4[(fr + j) * D4MAIN + k];      /* wave-uniform: LDS broadcast */
                a[j] = Acc<EXACT>::step(a[j], x.x, M[k].x, P[k].x);
                a[j] = Acc<EXACT>::step(a[j], x.y, M[k].y, P[k].y);
                a[j] = Acc<EXACT>::step(a[j], x.z, M[k].z, P[k].z);
                a[j] = Acc<EXACT>::step(a[j], x.w, M[k].w, P[k].w);
            }
        }
        /* transpose: tr[j][lane] <- gauscr of frame j */
#pragma unroll
        for (int j = 0; j < FB; j++)
            tr[j * 65 + lane] = gau_to_int((double)a[j], f, distfloor, mixw);
        /* same-wave LDS hand-off: no barrier needed, but order the accesses */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        /* lane (sl, c) runs the chain of frame slot j = c, c+CP, ... */
        for (int j = c; j < FB; j += CP) {
            int32_t score = S3A_LOGPROB_ZERO;
            const int32_t *row = tr + j * 65 + sl * CP;
#pragma unroll
            for (int cc = 0; cc < CP; cc++)
                score = la(score, row[cc]);
            if (score <= S3A_LOGPROB_ZERO) score = S3A_LOGPROB_ZERO;
            if (sen < S)
                senscr[(size_t)(f0 + fr + j) * S + sen] = score;
        }
        __builtin_amdgcn_wave_barrier();
    }
    /* ---- remaining frames one at a time: shuffle chain ---- */
    for (; fr < nf; fr++) {
        acc_t a = lrd_g;
#pragma unroll
        for (int k = 0; k < D4MAIN; k++) {
            float4 x = xs4[fr * D4MAIN + k];
            a = Acc<EXACT>::step(a, x.x, M[k].x, P[k].x);
            a = Acc<EXACT>::step(a, x.y, M[k].y, P[k].y);
            a = Acc<EXACT>::step(a, x.z, M[k].z, P[k].z);
            a = Acc<EXACT>::step(a, x.w, M[k].w, P[k].w);
        }
        int32_t gs = gau_to_int((double)a, f, distfloor, mixw);
        int32_t score = S3A_LOGPROB_ZERO;
#pragma unroll
        for (int cc = 0; cc < CP; cc++)
            score = la(score, __shfl(gs, sl * CP + cc, 64));
        if (score <= S3A_LOGPROB_ZERO) score = S3A_LOGPROB_ZERO;
        if (c == 0 && sen < S)
            senscr[(size_t)(f0 + fr) * S + sen] = score;
    }
}

/* ------------------------------------------------------------------ */
/* k_score_frame_sync: all senones x ONE frame                         */
/* ------------------------------------------------------------------ */
/*
 * The frame-synchronous pass is the read of the model plus, per senone, CP DEPENDENT log-add look-ups that
 * start when the wave's last parameter has arrived; from global memory each link costs ~0.25 us while
 * the model streams through the same L2, and the whole 58 KB table in LDS costs half the resident waves.  The
 * table is monotone and its values drop below 256 after ~9 k entries: 16-bit entries up to there and 8-bit ones
 * beyond are 38 KB, four workgroups per CU still fit, and every link is an LDS read.
 */
template <int CP, bool EXACT>
__global__ void __launch_bounds__(256)
k_score_frame_sync(const float4 *__restrict__ mean4, const float4 *__restrict__ prec4,
                   const float *__restrict__ lrd, const int32_t *__restrict__ mixw_g,
                   const uint8_t *__restrict__ hyb_g, int32_t hyb_bytes, uint32_t head, uint32_t tab_size,
                   int32_t lm_zero, double f, double distfloor, const float *__restrict__ feat, int32_t veclen,
                   int32_t *__restrict__ senscr, int32_t S, int32_t Gpad)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t hyb_s[];
    __shared__ __attribute__((aligned(16))) float xs[D4MAIN * 4];
    typedef typename Acc<EXACT>::T acc_t;
    const int32_t tid = threadIdx.x, lane = tid & 63;
    const int32_t g = blockIdx.x * 256 + tid;
    /* the lane's Gaussian first: the copies below run while these loads are in flight */
    float4 M[D4MAIN], P[D4MAIN];
#pragma unroll
    for (int k = 0; k < D4MAIN; k++) {
        M[k] = mean4[(size_t)k * Gpad + g];
        P[k] = prec4[(size_t)k * Gpad + g];
    }
    const acc_t lrd_g = (acc_t)lrd[g];
    const int32_t mixw = mixw_g[g];
    {
        /* <= HYB_MAX_PIECES 16-byte pieces per lane: all loads in flight, then the LDS writes (a load / wait /
         * write loop costs a round trip per piece) */
        const int32_t n16 = hyb_bytes / 16;
        uint4 piece[HYB_MAX_PIECES];
#pragma unroll
        for (int q = 0; q < HYB_MAX_PIECES; q++)
            piece[q] = (tid + q * 256 < n16) ? ((const uint4 *)hyb_g)[tid + q * 256] : make_uint4(0, 0, 0, 0);
        const float xv = (tid < D4MAIN * 4 && tid < veclen) ? feat[tid] : 0.0f;
#pragma unroll
        for (int q = 0; q < HYB_MAX_PIECES; q++)
            if (tid + q * 256 < n16) ((uint4 *)hyb_s)[tid + q * 256] = piece[q];
        if (tid < D4MAIN * 4) xs[tid] = xv;
    }
    __syncthreads();
    acc_t a = lrd_g;
    const float4 *xs4 = (const float4 *)xs;
#pragma unroll
    for (int k = 0; k < D4MAIN; k++) {
        const float4 x = xs4[k];
        a = Acc<EXACT>::step(a, x.x, M[k].x, P[k].x);
        a = Acc<EXACT>::step(a, x.y, M[k].y, P[k].y);
        a = Acc<EXACT>::step(a, x.z, M[k].z, P[k].z);
        a = Acc<EXACT>::step(a, x.w, M[k].w, P[k].w);
    }
    const int32_t gs = gau_to_int((double)a, f, distfloor, mixw);
    const int32_t c = lane & (CP - 1), sl = lane / CP, sen = g / CP;
    const uint16_t *head_s = (const uint16_t *)hyb_s;
    const uint8_t *tail_s = hyb_s + (size_t)head * 2;
    int32_t score = S3A_LOGPROB_ZERO;
#pragma unroll
    for (int cc = 0; cc < CP; cc++) {
        const int32_t y = __shfl(gs, sl * CP + cc, 64), x = score;
        /* logmath_add (logmath.c:391-436) on the repacked table */
        if (x <= lm_zero) { score = y; continue; }
        if (y <= lm_zero) continue;
        const int32_t hi = x > y ? x : y, lo = x > y ? y : x;
        const uint32_t d = (uint32_t)hi - (uint32_t)lo;
        if (d >= tab_size) { score = hi; continue; }
        score = hi + (d < head ? (int32_t)head_s[d] : (int32_t)tail_s[d - head]);
    }
    if (score <= S3A_LOGPROB_ZERO) score = S3A_LOGPROB_ZERO;
    if (c == 0 && sen < S)
        senscr[sen] = score;
}

/*
 * Generic-veclen fallback (any D4): parameters are re-read from memory for
 * every frame (L2-resident), one frame at a time.  Correct for every model the
 * loader accepts; only the 39/40-dim case above is tuned.
 */
template <bool EXACT>
__global__ void __launch_bounds__(256)
k_score_frames_generic(const float4 *__restrict__ mean4, const float4 *__restrict__ prec4,
                       const float *__restrict__ lrd, const int32_t *__restrict__ mixw_g,
                       const uint16_t *__restrict__ tab_g, uint32_t tab_size, int32_t lm_zero,
                       double f, double distfloor,
                       const float *__restrict__ feat, int32_t feat_stride, int32_t veclen,
                       int32_t D4, int32_t CP, int32_t *__restrict__ senscr, int32_t S,
                       int32_t Gpad, int32_t G)
{
    typedef typename Acc<EXACT>::T acc_t;
    const int32_t g = blockIdx.x * 256 + threadIdx.x;
    const int32_t fr = blockIdx.y;
    const int32_t lane = threadIdx.x & 63;
    const float *x = feat + (size_t)fr * feat_stride;
    LogAdd la;
    la.tab = tab_g; la.size = tab_size; la.zero = lm_zero;
    int32_t gs = S3A_LOGPROB_ZERO;
    if (g < Gpad) {
        acc_t a = (acc_t)lrd[g];
        for (int32_t k = 0; k < D4; k++) {
            float4 m = mean4[(size_t)k * Gpad + g], p = prec4[(size_t)k * Gpad + g];
            int32_t i = 4 * k;
            a = Acc<EXACT>::step(a, (i < veclen) ? x[i] : 0.0f, m.x, p.x);
            a = Acc<EXACT>::step(a, (i + 1 < veclen) ? x[i + 1] : 0.0f, m.y, p.y);
            a = Acc<EXACT>::step(a, (i + 2 < veclen) ? x[i + 2] : 0.0f, m.z, p.z);
            a = Acc<EXACT>::step(a, (i + 3 < veclen) ? x[i + 3] : 0.0f, m.w, p.w);
        }
        gs = gau_to_int((double)a, f, distfloor, mixw_g[g]);
    }
    const int32_t sl = lane / CP, c = lane & (CP - 1);
    int32_t score = S3A_LOGPROB_ZERO;
    for (int32_t cc = 0; cc < CP; cc++)
        score = la(score, __shfl(gs, sl * CP + cc, 64));
    if (score <= S3A_LOGPROB_ZERO) score = S3A_LOGPROB_ZERO;
    if (c == 0 && g < G)
        senscr[(size_t)fr * S + g / CP] = score;
}

/* per-frame maximum over senones: one workgroup per frame */
__global__ void __launch_bounds__(256)
k_frame_best(const int32_t *__restrict__ senscr, int32_t S, int32_t *__restrict__ best)
{
    __shared__ int32_t red[4];
    const int32_t *row = senscr + (size_t)blockIdx.x * S;
    int32_t m = INT_MIN;
    for (int32_t s = threadIdx.x; s < S; s += 256)
        m = max(m, row[s]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        m = max(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        best[blockIdx.x] = max(max(red[0], red[1]), max(red[2], red[3]));
}

/* ------------------------------------------------------------------ */
/* launchers                                                           */
/* ------------------------------------------------------------------ */
static size_t
score_lds_bytes(const struct s3a_mgau_dev_s *d, int32_t fpc, bool tab_lds, int32_t nt)
{
    size_t b = (size_t)fpc * D4MAIN * 4 * sizeof(float);
    b += (size_t)(nt / 64) * FB * 65 * sizeof(int32_t);
    b = (b + 15) & ~(size_t)15;
    if (tab_lds) b += ((size_t)d->tab_size * 2 + 15) & ~(size_t)15;
    return b;
}

template <int CP, bool EXACT, bool TAB_LDS, int NT>
static hipError_t
launch_score_t(const s3a_mgau_model_t *g, const float *feat_dev, int32_t feat_stride,
               int32_t n_frames, int32_t fpc, int32_t *senscr_dev, hipStream_t st)
{
    const struct s3a_mgau_dev_s *d = g->dev;
    int32_t n_tiles = d->Gpad / NT;
    int32_t n_chunks = (n_frames + fpc - 1) / fpc;
    int32_t grid = ((n_tiles + 7) / 8) * n_chunks * 8;
    size_t lds = score_lds_bytes(d, fpc, TAB_LDS, NT);
    auto kern = k_score_frames<CP, EXACT, TAB_LDS, NT>;
    static bool attr_set[64];           /* per device: the opt-in belongs to the device that is current */
    int dev_now = 0;
    (void)hipGetDevice(&dev_now);
    if (dev_now < 0 || dev_now >= 64 || !attr_set[dev_now]) {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
        if (dev_now >= 0 && dev_now < 64) attr_set[dev_now] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st,
                       d->mean4, d->prec4, d->lrd, d->mixw, d->tab16, d->tab_size, d->lm_zero,
                       g->f, g->distfloor, feat_dev, feat_stride, d->D, n_frames, fpc, n_chunks,
                       n_tiles, senscr_dev, d->S, d->Gpad);
    return hipGetLastError();
}

template <bool EXACT, bool TAB_LDS, int NT>
static hipError_t
launch_score_cp(const s3a_mgau_model_t *g, const float *feat_dev, int32_t feat_stride,
                int32_t n_frames, int32_t fpc, int32_t *senscr_dev, hipStream_t st)
{
#define S3A_CASE(cp) case cp: return launch_score_t<cp, EXACT, TAB_LDS, NT>(g, feat_dev, feat_stride, n_frames, fpc, senscr_dev, st)
    switch (g->dev->CP) {
    S3A_CASE(1); S3A_CASE(2); S3A_CASE(4); S3A_CASE(8); S3A_CASE(16); S3A_CASE(32);
    default: return launch_score_t<64, EXACT, TAB_LDS, NT>(g, feat_dev, feat_stride, n_frames, fpc, senscr_dev, st);
    }
#undef S3A_CASE
}

template <bool EXACT, bool TAB_LDS>
static hipError_t
launch_score_nt(const s3a_mgau_model_t *g, const float *feat_dev, int32_t feat_stride,
                int32_t n_frames, int32_t fpc, int32_t nt, int32_t *senscr_dev, hipStream_t st)
{
    switch (nt) {
    case 256:  return launch_score_cp<EXACT, TAB_LDS, 256>(g, feat_dev, feat_stride, n_frames, fpc, senscr_dev, st);
    case 1024: return launch_score_cp<EXACT, TAB_LDS, 1024>(g, feat_dev, feat_stride, n_frames, fpc, senscr_dev, st);
    default:   return launch_score_cp<EXACT, TAB_LDS, 512>(g, feat_dev, feat_stride, n_frames, fpc, senscr_dev, st);
    }
}

/* choose the frames-per-chunk so that the grid fills the chip (>= ~2 workgroups
 * of 8 waves per CU) without shrinking chunks below one FB group */
/* tuning knobs (s3a_set_variants): workgroup size and frames per chunk */
static int32_t
pick_nt(void)
{
    const int32_t nt = s3a_variants()->score_nt;
    return (nt == 256 || nt == 512 || nt == 1024) ? nt : 512;
}

/*
 * Frames per chunk.  A workgroup of the utterance kernel needs ~100 KB of LDS
 * (table + transpose tiles + its chunk's features), so ONE workgroup is resident
 * per CU and the grid executes in rounds of n_cu workgroups.  Time is
 * proportional to rounds x (frames per chunk + a fixed start-up cost of loading
 * the table and the Gaussians, worth about 10 frames): pick the chunk count that
 * minimises it, so that e.g. 96 tiles x 8 chunks = 768 = 3 full rounds of 256.
 */
static int32_t
pick_fpc(const struct s3a_mgau_dev_s *d, int32_t n_frames, int32_t nt)
{
    const int32_t forced = s3a_variants()->score_fpc;
    int32_t n_tiles = d->Gpad / nt;
    int32_t best_fpc = FB, nc;
    int64_t best_cost = -1;
    if (forced > 0) {
        int32_t f = ((forced + FB - 1) / FB) * FB;
        return f > 256 ? 256 : f;
    }
    for (nc = 1; nc <= 128; nc++) {
        int32_t fpc = (n_frames + nc - 1) / nc;
        int32_t chunks, rounds;
        int64_t cost;
        fpc = ((fpc + FB - 1) / FB) * FB;
        if (fpc > 256) continue;            /* 40 KB of LDS for features at most */
        chunks = (n_frames + fpc - 1) / fpc;
        rounds = (n_tiles * chunks + d->n_cu - 1) / d->n_cu;
        cost = (int64_t)rounds * (fpc + 10);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best_fpc = fpc;
        }
        if (fpc <= FB) break;
    }
    return best_fpc;
}

static int32_t
launch_score(const s3a_mgau_model_t *g, const float *feat_dev, int32_t feat_stride,
             int32_t n_frames, int32_t *senscr_dev, int32_t *best_dev, hipStream_t st)
{
    const struct s3a_mgau_dev_s *d = g->dev;
    const bool exact = g->precision == S3A_GMM_EXACT;
    hipError_t e;

    if (n_frames <= 0)
        return S3A_OK;
    /* one frame, long mixtures: the pass whose tail is the senone's chain of look-ups (16+ links; with 8 the
     * general kernel is as fast: 4.2 vs 4.4 us on the hub4 shape, 20.1 vs 17.0 us with 32) */
    const bool no_frame_sync = s3a_variants()->no_frame_sync_kernel != 0;
    if (d->D4 == D4MAIN && d->tab16 != NULL && n_frames == 1 && d->hyb_ok && d->CP >= 16 && !no_frame_sync) {
        const dim3 grid(d->Gpad / 256);
        const size_t lds = (size_t)d->hyb_bytes;
#define S3A_FS(cp)                                                                                          \
        case cp:                                                                                            \
            if (exact) hipLaunchKernelGGL((k_score_frame_sync<cp, true>), grid, dim3(256), lds, st, d->mean4, d->prec4, d->lrd, \
                           d->mixw, d->hyb_tab, d->hyb_bytes, (uint32_t)d->hyb_head, d->tab_size, d->lm_zero, g->f,       \
                           g->distfloor, feat_dev, d->D, senscr_dev, d->S, d->Gpad);                        \
            else hipLaunchKernelGGL((k_score_frame_sync<cp, false>), grid, dim3(256), lds, st, d->mean4, d->prec4, d->lrd, \
                           d->mixw, d->hyb_tab, d->hyb_bytes, (uint32_t)d->hyb_head, d->tab_size, d->lm_zero, g->f,       \
                           g->distfloor, feat_dev, d->D, senscr_dev, d->S, d->Gpad);                        \
            break
        switch (d->CP) { S3A_FS(1); S3A_FS(2); S3A_FS(4); S3A_FS(8); S3A_FS(16); S3A_FS(32); default: S3A_FS(64); }
#undef S3A_FS
        e = hipGetLastError();
    }
    else if (d->D4 == D4MAIN && d->tab16 != NULL) {
        int32_t nt = pick_nt();
        int32_t fpc = pick_fpc(d, n_frames, nt);
        /* the table is worth staging in LDS only if a workgroup does enough
         * log-adds to amortise the 58 KB copy, and only if it fits */
        bool tab_lds = n_frames >= 2 * FB && score_lds_bytes(d, fpc, true, nt) <= 160 * 1024;
        if (n_frames < 2 * FB) nt = 256;    /* few frames: more, smaller workgroups */
        if (exact)
            e = tab_lds ? launch_score_nt<true, true>(g, feat_dev, feat_stride, n_frames, fpc, nt, senscr_dev, st)
                        : launch_score_nt<true, false>(g, feat_dev, feat_stride, n_frames, fpc, nt, senscr_dev, st);
        else
            e = tab_lds ? launch_score_nt<false, true>(g, feat_dev, feat_stride, n_frames, fpc, nt, senscr_dev, st)
                        : launch_score_nt<false, false>(g, feat_dev, feat_stride, n_frames, fpc, nt, senscr_dev, st);
    }
    else {
        if (d->tab16 == NULL) {
            s3a_set_error("32-bit log-add tables are not supported by the scoring kernels");
            return S3A_EUNSUP;
        }
        dim3 grid((d->Gpad + 255) / 256, n_frames);
        if (exact)
            hipLaunchKernelGGL(k_score_frames_generic<true>, grid, dim3(256), 0, st, d->mean4,
                               d->prec4, d->lrd, d->mixw, d->tab16, d->tab_size, d->lm_zero, g->f,
                               g->distfloor, feat_dev, feat_stride, d->D, d->D4, d->CP, senscr_dev,
                               d->S, d->Gpad, d->G);
        else
            hipLaunchKernelGGL(k_score_frames_generic<false>, grid, dim3(256), 0, st, d->mean4,
                               d->prec4, d->lrd, d->mixw, d->tab16, d->tab_size, d->lm_zero, g->f,
                               g->distfloor, feat_dev, feat_stride, d->D, d->D4, d->CP, senscr_dev,
                               d->S, d->Gpad, d->G);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        s3a_set_error("score kernel launch failed: %s", hipGetErrorString(e));
        return S3A_EHIP;
    }
    if (best_dev) {
        hipLaunchKernelGGL(k_frame_best, dim3(n_frames), dim3(256), 0, st, senscr_dev, d->S,
                           best_dev);
        e = hipGetLastError();
        if (e != hipSuccess) {
            s3a_set_error("k_frame_best launch failed: %s", hipGetErrorString(e));
            return S3A_EHIP;
        }
    }
    return S3A_OK;
}

extern "C" int32_t
s3a_mgau_score_frames_dev(s3a_mgau_model_t *g, const float *feat_dev, int32_t n_frames,
                          int32_t *senscr_dev, int32_t *best_dev, void *stream)
{
    if (g && !g->dev) {
        s3a_set_error("host-only model handle: scoring needs a GPU (no CPU fallback)");
        return S3A_ENODEV;
    }
    if (!g || !feat_dev || !senscr_dev || n_frames < 0)
        return S3A_EINVAL;
    return launch_score(g, feat_dev, g->veclen, n_frames, senscr_dev, best_dev,
                        stream ? (hipStream_t)stream : g->dev->stream);
}

int32_t
s3a_dev_grow(void **buf, size_t *cap, size_t need)
{
    if (*cap >= need)
        return S3A_OK;
    if (*buf) hipFree(*buf);
    *buf = NULL;
    *cap = 0;
    HIPCHK(hipMalloc(buf, need));
    *cap = need;
    return S3A_OK;
}

extern "C" int32_t
s3a_mgau_score_frames(s3a_mgau_model_t *g, const float *feat, int32_t n_frames,
                      int32_t *senscr, int32_t *best)
{
    struct s3a_mgau_dev_s *d;
    int32_t rc;
    size_t fb, sb;

    if (g && !g->dev) {
        s3a_set_error("host-only model handle: scoring needs a GPU (no CPU fallback)");
        return S3A_ENODEV;
    }
    if (!g || !feat || !senscr || n_frames < 0)
        return S3A_EINVAL;
    if (n_frames == 0)
        return S3A_OK;
    d = g->dev;
    fb = (size_t)n_frames * d->D * sizeof(float);
    sb = (size_t)n_frames * d->S * sizeof(int32_t);
    if ((rc = s3a_dev_grow((void **)&d->feat_buf, &d->feat_cap, fb)) != S3A_OK) return rc;
    if ((rc = s3a_dev_grow((void **)&d->scr_buf, &d->scr_cap, sb)) != S3A_OK) return rc;
    if ((rc = s3a_dev_grow((void **)&d->best_buf, &d->best_cap, (size_t)n_frames * 4)) != S3A_OK) return rc;
    HIPCHK(hipMemcpyAsync(d->feat_buf, feat, fb, hipMemcpyHostToDevice, d->stream));
    rc = launch_score(g, d->feat_buf, d->D, n_frames, d->scr_buf, best ? d->best_buf : NULL,
                      d->stream);
    if (rc != S3A_OK)
        return rc;
    HIPCHK(hipMemcpyAsync(senscr, d->scr_buf, sb, hipMemcpyDeviceToHost, d->stream));
    if (best)
        HIPCHK(hipMemcpyAsync(best, d->best_buf, (size_t)n_frames * 4, hipMemcpyDeviceToHost,
                              d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    return S3A_OK;
}

/* ------------------------------------------------------------------ */
/* per-senone state                                                    */
/* ------------------------------------------------------------------ */
__global__ void
k_reset_state(int32_t *bstidx, int32_t *bstscr, int32_t *updatetime, int32_t S)
{
    int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < S) {
        bstidx[s] = S3A_NO_BSTIDX;
        bstscr[s] = S3A_LOGPROB_ZERO;
        updatetime[s] = S3A_NOT_UPDATED;
    }
}

/* the same reset on state arrays a scorer owns itself (s3a_scorer_init_private); asynchronous */
int32_t
s3a_reset_state_arrays(hipStream_t stream, int32_t *bstidx, int32_t *bstscr, int32_t *updatetime, int32_t S)
{
    hipLaunchKernelGGL(k_reset_state, dim3((S + 255) / 256), dim3(256), 0, stream, bstidx, bstscr, updatetime, S);
    HIPCHK(hipGetLastError());
    return S3A_OK;
}

extern "C" int32_t
s3a_mgau_reset_state(s3a_mgau_model_t *g)
{
    struct s3a_mgau_dev_s *d = g->dev;
    if (!d) return S3A_EINVAL;
    hipLaunchKernelGGL(k_reset_state, dim3((d->S + 255) / 256), dim3(256), 0, d->stream,
                       d->bstidx, d->bstscr, d->updatetime, d->S);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(d->stream));
    return S3A_OK;
}

extern "C" int32_t
s3a_mgau_get_state(const s3a_mgau_model_t *g, int32_t *bstidx, int32_t *bstscr,
                   int32_t *updatetime)
{
    const struct s3a_mgau_dev_s *d = g->dev;
    if (!d) return S3A_EINVAL;
    HIPCHK(hipStreamSynchronize(d->stream));
    if (bstidx) HIPCHK(hipMemcpy(bstidx, d->bstidx, d->S * 4, hipMemcpyDeviceToHost));
    if (bstscr) HIPCHK(hipMemcpy(bstscr, d->bstscr, d->S * 4, hipMemcpyDeviceToHost));
    if (updatetime) HIPCHK(hipMemcpy(updatetime, d->updatetime, d->S * 4, hipMemcpyDeviceToHost));
    return S3A_OK;
}

/* ------------------------------------------------------------------ */
/* mgau_eval: one senone, one vector (drop-in completeness)            */
/* ------------------------------------------------------------------ */
/*
 * One wave.  Lane c < CP evaluates component c; lane 0 then walks either all
 * components or the -1-terminated active list in order, log-adding and
 * tracking the best component with the reference's exact update rules
 * (including cont_mgau.c:1076-1079: inside mgau_eval_all the first component of
 * each pair updates bstidx regardless of update_best_id).
 */
template <bool EXACT>
__global__ void __launch_bounds__(64)
k_mgau_eval_one(const float4 *__restrict__ mean4, const float4 *__restrict__ prec4,
                const float *__restrict__ lrd, const int32_t *__restrict__ mixw_g,
                const uint16_t *__restrict__ tab_g, uint32_t tab_size, int32_t lm_zero,
                double f, double distfloor, const float *__restrict__ x, int32_t veclen,
                int32_t D4, int32_t CP, int32_t Gpad, int32_t m, int32_t n_comp,
                const int32_t *__restrict__ active, int32_t n_active,
                int32_t fr, int32_t update_best_id,
                int32_t *bstidx, int32_t *bstscr, int32_t *updatetime, int32_t *out)
{
    typedef typename Acc<EXACT>::T acc_t;
    __shared__ int32_t gs_s[64];
    const int32_t lane = threadIdx.x;
    LogAdd la;
    la.tab = tab_g; la.size = tab_size; la.zero = lm_zero;
    if (lane < CP) {
        int32_t g = m * CP + lane;
        acc_t a = (acc_t)lrd[g];
        for (int32_t k = 0; k < D4; k++) {
            float4 mm = mean4[(size_t)k * Gpad + g], pp = prec4[(size_t)k * Gpad + g];
            int32_t i = 4 * k;
            a = Acc<EXACT>::step(a, (i < veclen) ? x[i] : 0.0f, mm.x, pp.x);
            a = Acc<EXACT>::step(a, (i + 1 < veclen) ? x[i + 1] : 0.0f, mm.y, pp.y);
            a = Acc<EXACT>::step(a, (i + 2 < veclen) ? x[i + 2] : 0.0f, mm.z, pp.z);
            a = Acc<EXACT>::step(a, (i + 3 < veclen) ? x[i + 3] : 0.0f, mm.w, pp.w);
        }
        gs_s[lane] = gau_to_int((double)a, f, distfloor, mixw_g[g]);
    }
    __syncthreads();
    if (lane == 0) {
        int32_t bi = bstidx[m], bs = bstscr[m];
        int32_t score = S3A_LOGPROB_ZERO;
        if (update_best_id) {
            bi = S3A_NO_BSTIDX;
            bs = S3A_LOGPROB_ZERO;
            updatetime[m] = fr;
        }
        if (active == NULL) {
            for (int32_t c = 0; c < n_comp; c++) {
                int32_t gs = gs_s[c];
                bool pair_first = ((c & 1) == 0) && (c + 1 < n_comp);
                score = la(score, gs);
                if ((pair_first || update_best_id) && gs > bs) { bi = c; bs = gs; }
            }
        }
        else {
            for (int32_t j = 0; j < n_active; j++) {
                int32_t c = active[j];
                int32_t gs = gs_s[c];
                score = la(score, gs);
                if (update_best_id && gs > bs) { bi = c; bs = gs; }
            }
        }
        if (score <= S3A_LOGPROB_ZERO) score = S3A_LOGPROB_ZERO;
        bstidx[m] = bi;
        bstscr[m] = bs;
        *out = score;
    }
}

extern "C" int32_t
s3a_mgau_eval(s3a_mgau_model_t *g, int32_t m, const int32_t *active_comp, const float *x,
              int32_t fr, int32_t update_best_id)
{
    struct s3a_mgau_dev_s *d;
    int32_t n_active = 0, rc, result = S3A_LOGPROB_ZERO;
    float *dx;
    int32_t *dact = NULL, *dout;
    size_t need;

    if (!g || !g->dev || !x || m < 0 || m >= g->n_mgau) {
        s3a_set_error("s3a_mgau_eval: bad arguments");
        return S3A_LOGPROB_ZERO;
    }
    d = g->dev;
    if (d->tab16 == NULL) {
        s3a_set_error("32-bit log-add tables are not supported by the scoring kernels");
        return S3A_LOGPROB_ZERO;
    }
    if (active_comp)
        while (active_comp[n_active] >= 0) {
            if (active_comp[n_active] >= g->n_comp[m]) {
                s3a_set_error("s3a_mgau_eval: active component out of range");
                return S3A_LOGPROB_ZERO;
            }
            n_active++;
        }
    /* scratch: [x: D floats][active: n ints][out: 1 int] in feat_buf */
    need = (size_t)(d->D + n_active + 2) * 4;
    if ((rc = s3a_dev_grow((void **)&d->feat_buf, &d->feat_cap, need)) != S3A_OK)
        return S3A_LOGPROB_ZERO;
    dx = d->feat_buf;
    dout = (int32_t *)(dx + d->D);
    if (hipMemcpyAsync(dx, x, d->D * 4, hipMemcpyHostToDevice, d->stream) != hipSuccess)
        return S3A_LOGPROB_ZERO;
    if (active_comp) {
        dact = dout + 1;
        if (n_active && hipMemcpyAsync(dact, active_comp, n_active * 4, hipMemcpyHostToDevice,
                                       d->stream) != hipSuccess)
            return S3A_LOGPROB_ZERO;
    }
    if (g->precision == S3A_GMM_EXACT)
        hipLaunchKernelGGL(k_mgau_eval_one<true>, dim3(1), dim3(64), 0, d->stream, d->mean4,
                           d->prec4, d->lrd, d->mixw, d->tab16, d->tab_size, d->lm_zero, g->f,
                           g->distfloor, dx, d->D, d->D4, d->CP, d->Gpad, m, g->n_comp[m], dact,
                           n_active, fr, update_best_id, d->bstidx, d->bstscr, d->updatetime, dout);
    else
        hipLaunchKernelGGL(k_mgau_eval_one<false>, dim3(1), dim3(64), 0, d->stream, d->mean4,
                           d->prec4, d->lrd, d->mixw, d->tab16, d->tab_size, d->lm_zero, g->f,
                           g->distfloor, dx, d->D, d->D4, d->CP, d->Gpad, m, g->n_comp[m], dact,
                           n_active, fr, update_best_id, d->bstidx, d->bstscr, d->updatetime, dout);
    if (hipGetLastError() != hipSuccess
        || hipMemcpyAsync(&result, dout, 4, hipMemcpyDeviceToHost, d->stream) != hipSuccess
        || hipStreamSynchronize(d->stream) != hipSuccess) {
        s3a_set_error("s3a_mgau_eval: HIP failure");
        return S3A_LOGPROB_ZERO;
    }
    return result;
}

/* ------------------------------------------------------------------ */
/* measurement hook                                                    */
/* ------------------------------------------------------------------ */
extern "C" int32_t
s3a_bench_score_frames(s3a_mgau_model_t *g, const float *feat_dev, int32_t n_frames,
                       int32_t *senscr_dev, int32_t *best_dev, int32_t frames_per_launch,
                       int32_t iters, double *avg_us, double *avg_kernel_us, int32_t *n_launches)
{
    struct s3a_mgau_dev_s *d;
    hipEvent_t e0, e1;
    float ms = 0.0f;
    int32_t it, rc = S3A_OK, launches = 0;

    if (!g || !g->dev || !feat_dev || !senscr_dev || n_frames <= 0 || iters <= 0)
        return S3A_EINVAL;
    d = g->dev;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipStreamSynchronize(d->stream));
    HIPCHK(hipEventRecord(e0, d->stream));
    for (it = 0; it < iters && rc == S3A_OK; it++) {
        if (frames_per_launch <= 0) {
            rc = launch_score(g, feat_dev, d->D, n_frames, senscr_dev, best_dev, d->stream);
            launches = best_dev ? 2 : 1;
        }
        else {
            launches = 0;
            for (int32_t f0 = 0; f0 < n_frames && rc == S3A_OK; f0 += frames_per_launch) {
                int32_t nf = n_frames - f0 < frames_per_launch ? n_frames - f0 : frames_per_launch;
                rc = launch_score(g, feat_dev + (size_t)f0 * d->D, d->D, nf,
                                  senscr_dev + (size_t)f0 * d->S,
                                  best_dev ? best_dev + f0 : NULL, d->stream);
                launches += best_dev ? 2 : 1;
            }
        }
    }
    HIPCHK(hipEventRecord(e1, d->stream));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (rc != S3A_OK)
        return rc;
    if (avg_us) *avg_us = (double)ms * 1000.0 / iters;
    if (avg_kernel_us) *avg_kernel_us = (double)ms * 1000.0 / iters / launches;
    if (n_launches) *n_launches = launches;
    return S3A_OK;
}

extern "C" int32_t
s3a_stream_timer_begin(s3a_mgau_model_t *g)
{
    struct s3a_mgau_dev_s *d;
    if (!g || !g->dev) return S3A_EINVAL;
    d = g->dev;
    if (!d->ev0) {
        HIPCHK(hipEventCreate(&d->ev0));
        HIPCHK(hipEventCreate(&d->ev1));
    }
    HIPCHK(hipEventRecord(d->ev0, d->stream));
    return S3A_OK;
}

extern "C" int32_t
s3a_stream_timer_end(s3a_mgau_model_t *g, double *elapsed_us)
{
    struct s3a_mgau_dev_s *d;
    float ms = 0.0f;
    if (!g || !g->dev || !g->dev->ev0 || !elapsed_us) return S3A_EINVAL;
    d = g->dev;
    HIPCHK(hipEventRecord(d->ev1, d->stream));
    HIPCHK(hipEventSynchronize(d->ev1));
    HIPCHK(hipEventElapsedTime(&ms, d->ev0, d->ev1));
    *elapsed_us = (double)ms * 1000.0;
    return S3A_OK;
}
