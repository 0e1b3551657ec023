/*
 * s3a_scorer.hip -- the per-frame scoring driver with CI gating and the
 * composite-senone pass, on the device.
 *
 * Replaces (paths relative to cjac/cmusphinx, sphinx3/src/libs3decoder):
 *   libam/approx_cont_mgau.c:367-428   approx_cont_mgau_ci_eval
 *   libam/approx_cont_mgau.c:434-616   approx_cont_mgau_frame_eval
 *   libam/approx_cont_mgau.c:94-143    approx_isskip        (host, s3a_host side below)
 *   libam/approx_cont_mgau.c:303-357   approx_compute_dyn_ci_pbeam (host: <= 200 CI senones)
 *   libsearch/dict2pid.c:1029-1048     dict2pid_comsenscr
 *
 * k_gated_frame scores ONE frame.  Lane = Gaussian (as in s3a_device.hip); a
 * wave owns 64/CP senones.  Per senone the reference's three-way gate is
 * evaluated once (by every lane of the senone, identically):
 *   full   : senscr[ci] >= pbest + beam  -> all components, ordered log-add,
 *            bstidx/bstscr/updatetime updated (update_best_id = 1)
 *   single : else if scored last frame   -> only component bstidx; state is
 *            re-written only on a skipped (down-sampled) frame
 *   ci     : else                        -> copy of the parent CI senone's score
 * The frame maximum is reduced per wave and merged with one atomicMax; the
 * normalisation senscr[s] -= best for active senones is a second tiny kernel.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#include "s3a_device.h"
#include "s3a_structs.h"
#include "s3a_gated.h"

#pragma clang fp contract(off)

/* the frame's feature vector as a kernel argument (fused decoder path): 160 bytes ride in the launch packet
 * instead of a host-to-device copy (4 us of copy engine + its stream dependency per frame) */
struct GatedFeat { float v[D4MAIN * 4]; };

template <bool EXACT>
__global__ void __launch_bounds__(256)
k_gated_frame(const float4 *__restrict__ mean4, const float4 *__restrict__ prec4,
              const float *__restrict__ lrd, const int32_t *__restrict__ mixw_g,
              const uint16_t *__restrict__ tab_g, uint32_t tab_size, int32_t lm_zero,
              double f, double distfloor, const float *__restrict__ x, int32_t D4, int32_t CP,
              int32_t Gpad, int32_t sen_lo, int32_t sen_hi, int32_t ci_phase,
              const uint8_t *__restrict__ ncomp, const int16_t *__restrict__ cd2cisen,
              const uint8_t *__restrict__ sen_active, int32_t *__restrict__ senscr,
              int32_t pbest_plus_beam, const int32_t *__restrict__ pbest_ptr, int32_t beam,
              int32_t frame, int32_t is_skip,
              int32_t *bstidx, int32_t *bstscr, int32_t *updatetime, int32_t *misc, int32_t best_slot,
              uint8_t *clear_active, int32_t *gpart, int32_t gp_n, GatedFeat fa)
{
    if (x == NULL) x = fa.v;            /* (only the D4 == D4MAIN shape is launched this way) */
    if (D4 == D4MAIN)
        d_gated_frame<EXACT, D4MAIN>(mean4, prec4, lrd, mixw_g, tab_g, tab_size, lm_zero, f, distfloor, x, D4, CP, Gpad, sen_lo, sen_hi, ci_phase, ncomp, cd2cisen, sen_active, senscr, pbest_plus_beam, pbest_ptr, beam, frame, is_skip, bstidx, bstscr, updatetime, misc, best_slot, clear_active, gpart, gp_n, blockIdx.x);
    else
        d_gated_frame<EXACT, 0>(mean4, prec4, lrd, mixw_g, tab_g, tab_size, lm_zero, f, distfloor, x, D4, CP, Gpad, sen_lo, sen_hi, ci_phase, ncomp, cd2cisen, sen_active, senscr, pbest_plus_beam, pbest_ptr, beam, frame, is_skip, bstidx, bstscr, updatetime, misc, best_slot, clear_active, gpart, gp_n, blockIdx.x);
}

/* approx_cont_mgau.c:597-600 */
__global__ void
k_normalise(int32_t *senscr, uint8_t *sen_active, int32_t *misc, int32_t S, int32_t n_ci,
            int32_t ci_slot)
{
    int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    int32_t best = misc[0];
    if (ci_slot >= 0) best = max(best, misc[ci_slot]);   /* CI maximum kept in its own slot */
    if (s < n_ci) sen_active[s] = 1;    /* CI senones are forced active, approx_cont_mgau.c:537 */
    if (s < n_ci || sen_active[s])
        senscr[s] -= best;
    if (s == 0) misc[6] = best;         /* the frame's normaliser (srch->senscale) */
}

__global__ void
k_misc_reset(int32_t *misc)
{
    if (threadIdx.x < 8) misc[threadIdx.x] = (threadIdx.x == 0 || threadIdx.x == 5) ? INT_MIN : 0;
}

extern "C" s3a_scorer_t *
s3a_scorer_init(s3a_mgau_model_t *g, const int16_t *cd2cisen, int32_t n_sen, int32_t n_ci_sen,
                int32_t ds_ratio, int32_t cond_ds, double ci_pbeam, float tighten_factor,
                int32_t max_cd)
{
    s3a_scorer_t *sc;
    struct s3a_mgau_dev_s *d;
    int32_t s;

    if (!g || !g->dev || !cd2cisen || n_sen != g->n_mgau || n_ci_sen < 0 || n_ci_sen > n_sen
        || ds_ratio < 1) {
        s3a_set_error("s3a_scorer_init: bad arguments");
        return NULL;
    }
    if (cond_ds) {
        s3a_set_error("-cond_ds needs a Gaussian selector, which is not supported");
        return NULL;
    }
    /* mdef_is_cisenone(s) == (s == cd2cisen[s]) must hold exactly on [0, n_ci_sen):
     * the CD gate then sees pbest = max over ALL CI senones, as in the reference
     * where the CI senones come first (mdef.h:206-208) */
    for (s = 0; s < n_sen; s++) {
        int ok = (s < n_ci_sen) ? (cd2cisen[s] == s)
                                : (cd2cisen[s] >= 0 && cd2cisen[s] < n_ci_sen);
        if (!ok) {
            s3a_set_error("cd2cisen: CI senones must be exactly the first %d senones "
                          "(violated at senone %d -> %d)", n_ci_sen, s, cd2cisen[s]);
            return NULL;
        }
    }
    d = g->dev;
    if (d->tab16 == NULL) {
        s3a_set_error("32-bit log-add tables are not supported by the scoring kernels");
        return NULL;
    }
    sc = (s3a_scorer_t *)calloc(1, sizeof *sc);
    sc->g = g;
    sc->n_sen = n_sen;
    sc->n_ci_sen = n_ci_sen;
    sc->ds_ratio = ds_ratio;
    sc->cond_ds = cond_ds;
    sc->ci_pbeam = s3a_logs3(g->lm, ci_pbeam);      /* fast_algo_struct.c:438 */
    sc->dyn_ci_pbeam = sc->ci_pbeam;
    sc->tighten_factor = tighten_factor;
    sc->max_cd = max_cd;
    sc->cd2cisen_h = (int16_t *)malloc(sizeof(int16_t) * n_sen);
    memcpy(sc->cd2cisen_h, cd2cisen, sizeof(int16_t) * n_sen);
    sc->ci_occ_h = (int32_t *)calloc(n_sen, sizeof(int32_t));
    sc->idx_h = (int32_t *)calloc(n_ci_sen > 0 ? n_ci_sen : 1, sizeof(int32_t));
    {
        uint8_t *nc = (uint8_t *)malloc(n_sen);
        for (s = 0; s < n_sen; s++) nc[s] = (uint8_t)g->n_comp[s];
        if (hipMalloc(&sc->cd2cisen_d, sizeof(int16_t) * n_sen) != hipSuccess
            || hipMalloc(&sc->ncomp_d, n_sen) != hipSuccess
            || hipMalloc(&sc->x_d, sizeof(float) * d->D4 * 4) != hipSuccess
            || hipMalloc(&sc->act_d, n_sen) != hipSuccess
            || hipMalloc(&sc->scr_d, sizeof(int32_t) * n_sen) != hipSuccess
            || hipMalloc(&sc->ci_d, sizeof(int32_t) * (n_ci_sen > 0 ? n_ci_sen : 1)) != hipSuccess
            || hipMalloc(&sc->misc_d, sizeof(int32_t) * 8) != hipSuccess
            || hipMalloc(&sc->gpart_d, sizeof(int32_t) * 3 * (sc->gp_n = ((n_sen - n_ci_sen) * d->CP + 255) / 256, sc->gp_n > 0 ? sc->gp_n : 1)) != hipSuccess
            || hipHostMalloc(&sc->misc_h, sizeof(int32_t) * 8) != hipSuccess
            || hipMemcpy(sc->cd2cisen_d, cd2cisen, sizeof(int16_t) * n_sen, hipMemcpyHostToDevice) != hipSuccess
            || hipMemcpy(sc->ncomp_d, nc, n_sen, hipMemcpyHostToDevice) != hipSuccess
            || hipMemset(sc->x_d, 0, sizeof(float) * d->D4 * 4) != hipSuccess
            || hipMemset(sc->scr_d, 0, sizeof(int32_t) * n_sen) != hipSuccess) {
            free(nc);
            s3a_set_error("s3a_scorer_init: device allocation failed");
            s3a_scorer_free(sc);
            return NULL;
        }
        free(nc);
    }
    sc->bstidx_d = d->bstidx; sc->bstscr_d = d->bstscr; sc->updatetime_d = d->updatetime;
    return sc;
}

/* A scorer whose per-utterance Gaussian-selection state (mgau_t.bstidx / bstscr / updatetime,
 * cont_mgau.h:174-176) is its own: several decoders can then share ONE model on the device
 * (the batched engine reads the 16 MB of a hub4 model once per step instead of once per decoder). */
extern "C" s3a_scorer_t *
s3a_scorer_init_private(s3a_mgau_model_t *g, const int16_t *cd2cisen, int32_t n_sen, int32_t n_ci_sen,
                        int32_t ds_ratio, int32_t cond_ds, double ci_pbeam, float tighten_factor,
                        int32_t max_cd)
{
    s3a_scorer_t *sc = s3a_scorer_init(g, cd2cisen, n_sen, n_ci_sen, ds_ratio, cond_ds, ci_pbeam, tighten_factor, max_cd);
    if (!sc) return NULL;
    sc->bstidx_d = sc->bstscr_d = sc->updatetime_d = NULL;
    if (hipMalloc(&sc->bstidx_d, sizeof(int32_t) * n_sen) != hipSuccess
        || hipMalloc(&sc->bstscr_d, sizeof(int32_t) * n_sen) != hipSuccess
        || hipMalloc(&sc->updatetime_d, sizeof(int32_t) * n_sen) != hipSuccess) {
        s3a_set_error("s3a_scorer_init_private: device allocation failed");
        sc->own_state = 1;
        s3a_scorer_free(sc);
        return NULL;
    }
    sc->own_state = 1;
    if (s3a_scorer_utt_begin(sc) != S3A_OK) { s3a_scorer_free(sc); return NULL; }
    return sc;
}

extern "C" void
s3a_scorer_free(s3a_scorer_t *sc)
{
    if (!sc) return;
    (void)hipFree(sc->cd2cisen_d); (void)hipFree(sc->ncomp_d); (void)hipFree(sc->x_d);
    (void)hipFree(sc->act_d); (void)hipFree(sc->scr_d); (void)hipFree(sc->ci_d);
    (void)hipFree(sc->misc_d); (void)hipFree(sc->gpart_d);
    if (sc->own_state) { (void)hipFree(sc->bstidx_d); (void)hipFree(sc->bstscr_d); (void)hipFree(sc->updatetime_d); }
    if (sc->misc_h) (void)hipHostFree(sc->misc_h);
    free(sc->cd2cisen_h); free(sc->ci_occ_h); free(sc->idx_h);
    free(sc);
}

extern "C" int32_t
s3a_scorer_utt_begin(s3a_scorer_t *sc)
{
    if (!sc) return S3A_EINVAL;
    sc->skip_count = 0;
    if (sc->own_state)
        return s3a_reset_state_arrays(sc->g->dev->stream, sc->bstidx_d, sc->bstscr_d, sc->updatetime_d, sc->n_sen);
    return s3a_mgau_reset_state(sc->g);
}

static void
launch_gated(s3a_scorer_t *sc, int32_t lo, int32_t hi, int32_t ci_phase, int32_t thresh,
             int32_t frame, int32_t is_skip, const int32_t *pbest_ptr = NULL, int32_t beam = 0,
             int32_t best_slot = 0, uint8_t *clear_active = NULL, int32_t *gpart = NULL,
             const float *feat_host = NULL)
{
    s3a_mgau_model_t *g = sc->g;
    struct s3a_mgau_dev_s *d = g->dev;
    int32_t n_gau = (hi - lo) * d->CP;
    int32_t grid = (n_gau + 255) / 256;
    if (grid <= 0) return;
    GatedFeat fa;
    memset(&fa, 0, sizeof fa);
    const bool by_arg = feat_host != NULL && d->D4 == D4MAIN;
    if (by_arg) memcpy(fa.v, feat_host, sizeof(float) * d->D);
    const float *xp = by_arg ? (const float *)NULL : sc->x_d;
    if (g->precision == S3A_GMM_EXACT)
        hipLaunchKernelGGL(k_gated_frame<true>, dim3(grid), dim3(256), 0, d->stream, d->mean4,
                           d->prec4, d->lrd, d->mixw, d->tab16, d->tab_size, d->lm_zero, g->f,
                           g->distfloor, xp, d->D4, d->CP, d->Gpad, lo, hi, ci_phase,
                           sc->ncomp_d, sc->cd2cisen_d, sc->act_d, sc->scr_d, thresh, pbest_ptr, beam,
                           frame, is_skip, sc->bstidx_d, sc->bstscr_d, sc->updatetime_d, sc->misc_d, best_slot, clear_active, gpart, sc->gp_n, fa);
    else
        hipLaunchKernelGGL(k_gated_frame<false>, dim3(grid), dim3(256), 0, d->stream, d->mean4,
                           d->prec4, d->lrd, d->mixw, d->tab16, d->tab_size, d->lm_zero, g->f,
                           g->distfloor, xp, d->D4, d->CP, d->Gpad, lo, hi, ci_phase,
                           sc->ncomp_d, sc->cd2cisen_d, sc->act_d, sc->scr_d, thresh, pbest_ptr, beam,
                           frame, is_skip, sc->bstidx_d, sc->bstscr_d, sc->updatetime_d, sc->misc_d, best_slot, clear_active, gpart, sc->gp_n, fa);
}

static const int32_t k_misc_init[8] = { INT_MIN, 0, 0, 0, 0, 0, 0, 0 };

extern "C" int32_t
s3a_approx_cont_mgau_ci_eval(s3a_scorer_t *sc, const float *feat, int32_t *ci_senscr,
                             int32_t *best_score, int32_t fr)
{
    struct s3a_mgau_dev_s *d;
    if (!sc || !feat || !ci_senscr || !best_score)
        return S3A_EINVAL;
    d = sc->g->dev;
    HIPCHK(hipMemcpyAsync(sc->x_d, feat, sizeof(float) * d->D, hipMemcpyHostToDevice, d->stream));
    HIPCHK(hipMemcpyAsync(sc->misc_d, k_misc_init, sizeof k_misc_init, hipMemcpyHostToDevice, d->stream));
    launch_gated(sc, 0, sc->n_ci_sen, 1, 0, fr, 0);
    HIPCHK(hipGetLastError());
    if (sc->n_ci_sen)
        HIPCHK(hipMemcpyAsync(ci_senscr, sc->scr_d, sizeof(int32_t) * sc->n_ci_sen,
                              hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipMemcpyAsync(sc->misc_h, sc->misc_d, sizeof(int32_t) * 8, hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    *best_score = sc->misc_h[0];        /* MAX_NEG_INT32 when there are no CI senones */
    return S3A_OK;
}

/* approx_compute_dyn_ci_pbeam, approx_cont_mgau.c:303-357 (host: n_ci_sen is ~150) */
static __thread const int32_t *g_sort_key;    /* (qsort has no context argument) */
static int
cmp_ci_desc(const void *a, const void *b)
{
    return g_sort_key[*(const int32_t *)b] - g_sort_key[*(const int32_t *)a];
}

static int32_t
dyn_ci_pbeam(s3a_scorer_t *sc, const uint8_t *sen_active, const int32_t *ci)
{
    int32_t s, total = 0, pbest;
    for (s = 0; s < sc->n_sen; s++) {
        if (s < sc->n_ci_sen)
            sc->ci_occ_h[s] = 0;
        else if (sen_active[s])
            sc->ci_occ_h[sc->cd2cisen_h[s]]++;
    }
    for (s = 0; s < sc->n_ci_sen; s++)
        sc->idx_h[s] = s;
    g_sort_key = ci;
    qsort(sc->idx_h, sc->n_ci_sen, sizeof(int32_t), cmp_ci_desc);
    pbest = ci[sc->idx_h[0]];
    sc->dyn_ci_pbeam = sc->ci_pbeam;
    for (s = 0; s < sc->n_ci_sen && ci[sc->idx_h[s]] > pbest + sc->ci_pbeam; s++) {
        total += sc->ci_occ_h[sc->idx_h[s]];
        if (total > sc->max_cd) {
            sc->dyn_ci_pbeam = ci[sc->idx_h[s]] - pbest;
            break;
        }
    }
    return sc->dyn_ci_pbeam;
}

extern "C" int32_t
s3a_approx_cont_mgau_frame_eval(s3a_scorer_t *sc, uint8_t *sen_active, uint8_t *rec_sen_active,
                                int32_t *senscr, const float *feat, int32_t frame,
                                const int32_t *cache_ci_senscr, int32_t *best,
                                int32_t *n_sen_eval, int32_t *n_gau_eval)
{
    struct s3a_mgau_dev_s *d;
    int32_t beam, is_skip, pbest, s, ci_best;
    int32_t init[8];

    if (!sc || !sen_active || !senscr || !feat || !cache_ci_senscr || !best)
        return S3A_EINVAL;
    d = sc->g->dev;

    /* host-side scalar logic of approx_cont_mgau.c:487-511 */
    if (sc->max_cd < sc->n_sen - sc->n_ci_sen && sc->n_ci_sen > 0)
        beam = dyn_ci_pbeam(sc, sen_active, cache_ci_senscr);
    else
        beam = sc->ci_pbeam;
    is_skip = (frame % sc->ds_ratio == 0) ? 0 : 1;          /* approx_isskip without a selector */
    if (is_skip)
        beam = (int32_t)((float)beam * sc->tighten_factor);

    /* CI senones: copied from the cache, forced active (approx_cont_mgau.c:529-538) */
    pbest = S3A_MAX_NEG_INT32;
    for (s = 0; s < sc->n_ci_sen; s++) {
        if (pbest < cache_ci_senscr[s]) pbest = cache_ci_senscr[s];
        sen_active[s] = 1;
    }
    ci_best = pbest;
    memcpy(init, k_misc_init, sizeof init);
    init[0] = ci_best;              /* best starts from the CI maximum */

    HIPCHK(hipMemcpyAsync(sc->x_d, feat, sizeof(float) * d->D, hipMemcpyHostToDevice, d->stream));
    HIPCHK(hipMemcpyAsync(sc->act_d, sen_active, sc->n_sen, hipMemcpyHostToDevice, d->stream));
    if (sc->n_ci_sen)
        HIPCHK(hipMemcpyAsync(sc->scr_d, cache_ci_senscr, sizeof(int32_t) * sc->n_ci_sen,
                              hipMemcpyHostToDevice, d->stream));
    HIPCHK(hipMemcpyAsync(sc->misc_d, init, sizeof init, hipMemcpyHostToDevice, d->stream));
    launch_gated(sc, sc->n_ci_sen, sc->n_sen, 0,
                 (int32_t)((uint32_t)pbest + (uint32_t)beam), frame, is_skip);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_normalise, dim3((sc->n_sen + 255) / 256), dim3(256), 0, d->stream,
                       sc->scr_d, sc->act_d, sc->misc_d, sc->n_sen, sc->n_ci_sen, -1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(senscr, sc->scr_d, sizeof(int32_t) * sc->n_sen, hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipMemcpyAsync(sc->misc_h, sc->misc_d, sizeof(int32_t) * 8, hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    if (rec_sen_active)
        memcpy(rec_sen_active, sen_active, sc->n_sen);
    *best = sc->misc_h[0];
    if (n_sen_eval) *n_sen_eval = sc->misc_h[1];
    if (n_gau_eval) *n_gau_eval = sc->misc_h[2];
    return S3A_OK;
}

extern "C" uint8_t *s3a_scorer_sen_active_dev(s3a_scorer_t *sc) { return sc ? sc->act_d : NULL; }
extern "C" int32_t *s3a_scorer_senscr_dev(s3a_scorer_t *sc) { return sc ? sc->scr_d : NULL; }
extern "C" void *s3a_mgau_stream(s3a_mgau_model_t *g) { return (g && g->dev) ? (void *)g->dev->stream : NULL; }

/* ------------------------------------------------------------------ */
/* composite senones                                                   */
/* ------------------------------------------------------------------ */
__global__ void
k_comsenscr(int32_t n, const int32_t *__restrict__ off, const int16_t *__restrict__ list,
            const int32_t *__restrict__ wt, const int32_t *__restrict__ senscr,
            int32_t *__restrict__ out)
{
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t b = off[i], e = off[i + 1];
    int32_t best = senscr[list[b]];
    for (int32_t j = b + 1; j < e; j++)
        best = max(best, senscr[list[j]]);
    out[i] = (int32_t)((uint32_t)best + (uint32_t)wt[i]);
}

extern "C" s3a_comsen_t *
s3a_comsen_init(int32_t n_comstate, const int32_t *comstate_off, const int16_t *comstate,
                const int32_t *comwt)
{
    s3a_comsen_t *cs;
    int32_t i, n_list;
    if (n_comstate <= 0 || !comstate_off || !comstate || !comwt) {
        s3a_set_error("s3a_comsen_init: bad arguments");
        return NULL;
    }
    n_list = comstate_off[n_comstate];
    for (i = 0; i < n_comstate; i++)
        if (comstate_off[i + 1] <= comstate_off[i]) {
            s3a_set_error("s3a_comsen_init: composite state %d has no member senone", i);
            return NULL;
        }
    cs = (s3a_comsen_t *)calloc(1, sizeof *cs);
    cs->n_comstate = n_comstate;
    cs->n_list = n_list;
    if (hipMalloc(&cs->off_d, 4 * (n_comstate + 1)) != hipSuccess
        || hipMalloc(&cs->wt_d, 4 * n_comstate) != hipSuccess
        || hipMalloc(&cs->out_d, 4 * n_comstate) != hipSuccess
        || hipMalloc(&cs->list_d, 2 * n_list) != hipSuccess
        || hipMemcpy(cs->off_d, comstate_off, 4 * (n_comstate + 1), hipMemcpyHostToDevice) != hipSuccess
        || hipMemcpy(cs->wt_d, comwt, 4 * n_comstate, hipMemcpyHostToDevice) != hipSuccess
        || hipMemcpy(cs->list_d, comstate, 2 * n_list, hipMemcpyHostToDevice) != hipSuccess
        || hipStreamCreateWithFlags(&cs->stream, hipStreamNonBlocking) != hipSuccess) {
        s3a_set_error("s3a_comsen_init: device allocation failed");
        s3a_comsen_free(cs);
        return NULL;
    }
    return cs;
}

extern "C" void
s3a_comsen_free(s3a_comsen_t *cs)
{
    if (!cs) return;
    (void)hipFree(cs->off_d); (void)hipFree(cs->wt_d); (void)hipFree(cs->out_d);
    (void)hipFree(cs->list_d); (void)hipFree(cs->scr_d);
    if (cs->stream) (void)hipStreamDestroy(cs->stream);
    free(cs);
}

extern "C" int32_t
s3a_dict2pid_comsenscr(s3a_comsen_t *cs, const int32_t *senscr, int32_t n_sen, int32_t *comsenscr)
{
    int32_t rc;
    if (!cs || !senscr || !comsenscr || n_sen <= 0)
        return S3A_EINVAL;
    if ((rc = s3a_dev_grow((void **)&cs->scr_d, &cs->scr_cap, (size_t)n_sen * 4)) != S3A_OK)
        return rc;
    HIPCHK(hipMemcpyAsync(cs->scr_d, senscr, (size_t)n_sen * 4, hipMemcpyHostToDevice, cs->stream));
    hipLaunchKernelGGL(k_comsenscr, dim3((cs->n_comstate + 255) / 256), dim3(256), 0, cs->stream,
                       cs->n_comstate, cs->off_d, cs->list_d, cs->wt_d, cs->scr_d, cs->out_d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(comsenscr, cs->out_d, (size_t)cs->n_comstate * 4, hipMemcpyDeviceToHost, cs->stream));
    HIPCHK(hipStreamSynchronize(cs->stream));
    return S3A_OK;
}

extern "C" int32_t *s3a_comsen_dev(s3a_comsen_t *cs) { return cs ? cs->out_d : NULL; }

extern "C" int32_t
s3a_approx_cont_mgau_frame_eval_dev(s3a_scorer_t *sc, s3a_comsen_t *cs, const float *feat,
                                    int32_t frame, const int32_t *cache_ci_senscr, int32_t *best,
                                    int32_t *n_sen_eval, int32_t *n_gau_eval)
{
    struct s3a_mgau_dev_s *d;
    int32_t beam, is_skip, pbest, s;
    int32_t init[8];

    if (!sc || !feat || !cache_ci_senscr || !best)
        return S3A_EINVAL;
    if (sc->max_cd < sc->n_sen - sc->n_ci_sen) {
        s3a_set_error("-maxcdsenpf (dynamic CI beam) needs the host senone mask: use the host-pointer entry point");
        return S3A_EUNSUP;
    }
    d = sc->g->dev;
    beam = sc->ci_pbeam;
    is_skip = (frame % sc->ds_ratio == 0) ? 0 : 1;
    if (is_skip)
        beam = (int32_t)((float)beam * sc->tighten_factor);
    pbest = S3A_MAX_NEG_INT32;
    for (s = 0; s < sc->n_ci_sen; s++)
        if (pbest < cache_ci_senscr[s]) pbest = cache_ci_senscr[s];
    memcpy(init, k_misc_init, sizeof init);
    init[0] = pbest;
    HIPCHK(hipMemcpyAsync(sc->x_d, feat, sizeof(float) * d->D, hipMemcpyHostToDevice, d->stream));
    if (sc->n_ci_sen)
        HIPCHK(hipMemcpyAsync(sc->scr_d, cache_ci_senscr, sizeof(int32_t) * sc->n_ci_sen,
                              hipMemcpyHostToDevice, d->stream));
    HIPCHK(hipMemcpyAsync(sc->misc_d, init, sizeof init, hipMemcpyHostToDevice, d->stream));
    launch_gated(sc, sc->n_ci_sen, sc->n_sen, 0, (int32_t)((uint32_t)pbest + (uint32_t)beam), frame, is_skip);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_normalise, dim3((sc->n_sen + 255) / 256), dim3(256), 0, d->stream,
                       sc->scr_d, sc->act_d, sc->misc_d, sc->n_sen, sc->n_ci_sen, -1);
    HIPCHK(hipGetLastError());
    if (cs) {
        /* cs has its own stream object but the composite pass must follow the scores: run it here */
        hipLaunchKernelGGL(k_comsenscr, dim3((cs->n_comstate + 255) / 256), dim3(256), 0, d->stream,
                           cs->n_comstate, cs->off_d, cs->list_d, cs->wt_d, sc->scr_d, cs->out_d);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(sc->misc_h, sc->misc_d, sizeof(int32_t) * 8, hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    *best = sc->misc_h[0];
    if (n_sen_eval) *n_sen_eval = sc->misc_h[1];
    if (n_gau_eval) *n_gau_eval = sc->misc_h[2];
    return S3A_OK;
}

extern "C" int32_t *s3a_scorer_misc_dev(s3a_scorer_t *sc) { return sc ? sc->misc_d : NULL; }

/*
 * Fully asynchronous frame: CI senones, the CI gate, CD senones, normalisation and the
 * composite pass are enqueued back to back; nothing is read back.  The frame's results
 * stay on the device: senscr (s3a_scorer_senscr_dev), comsen (s3a_comsen_dev) and
 * misc[8] = {.., [1] #CD senones evaluated, [2] #CD Gaussians, [3] #CI senones,
 * [4] #CI Gaussians, [5] CI best, [6] frame normaliser = srch->senscale}.
 */
extern "C" int32_t
s3a_approx_cont_mgau_frame_eval_async(s3a_scorer_t *sc, s3a_comsen_t *cs, const float *feat,
                                      int32_t frame)
{
    struct s3a_mgau_dev_s *d;
    int32_t beam, is_skip;
    if (!sc || !feat) return S3A_EINVAL;
    if (sc->max_cd < sc->n_sen - sc->n_ci_sen) {
        s3a_set_error("-maxcdsenpf (dynamic CI beam) needs the host senone mask: use the host-pointer entry point");
        return S3A_EUNSUP;
    }
    d = sc->g->dev;
    beam = sc->ci_pbeam;
    is_skip = (frame % sc->ds_ratio == 0) ? 0 : 1;
    if (is_skip)
        beam = (int32_t)((float)beam * sc->tighten_factor);
    HIPCHK(hipMemcpyAsync(sc->x_d, feat, sizeof(float) * d->D, hipMemcpyHostToDevice, d->stream));
    hipLaunchKernelGGL(k_misc_reset, dim3(1), dim3(64), 0, d->stream, sc->misc_d);
    launch_gated(sc, 0, sc->n_ci_sen, 1, 0, frame, 0, NULL, 0, 5);                 /* CI -> misc[5] */
    launch_gated(sc, sc->n_ci_sen, sc->n_sen, 0, 0, frame, is_skip, sc->misc_d + 5, beam, 0);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_normalise, dim3((sc->n_sen + 255) / 256), dim3(256), 0, d->stream,
                       sc->scr_d, sc->act_d, sc->misc_d, sc->n_sen, sc->n_ci_sen, 5);
    if (cs)
        hipLaunchKernelGGL(k_comsenscr, dim3((cs->n_comstate + 255) / 256), dim3(256), 0, d->stream,
                           cs->n_comstate, cs->off_d, cs->list_d, cs->wt_d, sc->scr_d, cs->out_d);
    HIPCHK(hipGetLastError());
    return S3A_OK;
}

/*
 * Internal (fused decoder frame, s3a_decoder.hip): enqueue the CI phase and the gated CD
 * phase only.  Scores stay RAW in scr_d (the search kernels subtract the frame best on the
 * fly, which is what approx_cont_mgau.c:597-600 does for every senone an active HMM can
 * read); misc[0] = best over evaluated CD senones, misc[5] = CI best; the senone mask is
 * consumed and cleared.  misc must have been reset (k_misc_reset / the previous frame's
 * finishing kernel).
 */
int32_t
s3a_scorer_enqueue_raw(s3a_scorer_t *sc, const float *feat, int32_t frame)
{
    struct s3a_mgau_dev_s *d = sc->g->dev;
    int32_t beam = sc->ci_pbeam, is_skip = (frame % sc->ds_ratio == 0) ? 0 : 1;
    if (sc->max_cd < sc->n_sen - sc->n_ci_sen) {
        s3a_set_error("-maxcdsenpf (dynamic CI beam) needs the host senone mask: use the host-pointer entry point");
        return S3A_EUNSUP;
    }
    if (is_skip)
        beam = (int32_t)((float)beam * sc->tighten_factor);
    /* 39/40-dimensional features travel as a kernel argument; other shapes through the device buffer */
    if (d->D4 != D4MAIN)
        HIPCHK(hipMemcpyAsync(sc->x_d, feat, sizeof(float) * d->D, hipMemcpyHostToDevice, d->stream));
    launch_gated(sc, 0, sc->n_ci_sen, 1, 0, frame, 0, NULL, 0, 5, NULL, NULL, feat);
    launch_gated(sc, sc->n_ci_sen, sc->n_sen, 0, 0, frame, is_skip, sc->misc_d + 5, beam, 0, sc->act_d, sc->gpart_d, feat);
    sc->gpart_valid = sc->gp_n > 0;
    HIPCHK(hipGetLastError());
    return S3A_OK;
}

int32_t
s3a_scorer_reset_frame_state(s3a_scorer_t *sc)
{
    struct s3a_mgau_dev_s *d = sc->g->dev;
    hipLaunchKernelGGL(k_misc_reset, dim3(1), dim3(64), 0, d->stream, sc->misc_d);
    HIPCHK(hipMemsetAsync(sc->act_d, 0, sc->n_sen, d->stream));
    HIPCHK(hipGetLastError());
    return S3A_OK;
}
