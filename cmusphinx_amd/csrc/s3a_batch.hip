/*
 * s3a_batch.hip -- B decoders per kernel launch: the fused frame of s3a_decoder.hip, batched.
 *
 * Why: one mode-4 search frame is ~11 dependent, latency-bound launches that leave the chip
 * almost idle (a few thousand HMMs against 256 CUs); independent decoders on separate HIP
 * streams overlap only as far as the 4 hardware queues HIP multiplexes them onto (measured:
 * 4 streams 2.7x one stream, 8 and 16 streams no better).  Utterances are independent, so B
 * decoders can share EVERY launch instead: the grid gets a z dimension (= the decoders that are
 * ready), each workgroup reads its decoder's pointers from a descriptor table and its frame
 * parameters (beams, frame number, this frame's lextree_enter calls, the feature vector) from a
 * per-step array uploaded with one copy, and runs the very same kernel bodies
 * (s3a_decoder_kernels.h, s3a_gated.h).  One synchronisation per STEP serves all B decoders.
 *
 * Host side: each decoder stays the reference's own single-threaded C (one kb_t, one host
 * thread: vithist, LM, word transitions).  Per frame a thread records its lextree_enter calls
 * (s3a_batch_transition: host only) and then calls s3a_batch_step with the frame's features and
 * beams; the call blocks until every decoder that is inside an utterance has arrived, the last
 * arrival runs the batch for all of them, and each thread returns with its own frame record.
 * Decoders between utterances (backtrace, file I/O) are not waited for.  Per-decoder state is
 * untouched by the batching, so the results are those of the single-decoder path -- which the
 * tests check (tests/test_gpu_batch.py; the drop-in with S3A_BATCH=1).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <pthread.h>
#include <time.h>
#include <vector>

#include "s3a_device.h"
#include "s3a_structs.h"
#include "s3a_decoder_kernels.h"
#include "s3a_gated.h"

#define BMAXC 96            /* lextree_enter calls per decoder per frame (#CI phones + 1) */
#define BMAXSLOT 256

struct BSlot {              /* one decoder: static model + its state, all device pointers */
    int32_t N, T, n_tmat, maxn;
    const int32_t *node_base, *ssid, *tmatid, *wid, *prob, *child_off, *child, *par_off, *par, *tree_of,
        *rootlist, *tp;
    const uint8_t *comp;
    const int16_t *sseq, *comsseq;
    int32_t *sc, *hist, *outs, *outh, *bests, *frame, *pos, *posf, *act[2], *nact[2], *turn, *selfemit,
        *cnt, *base, *best, *exits, *nexit, *first, *eflag, *hbin, *done, *ctot, *n0, *pstamp, *propf, *poswid, *posout, *scan_flag;
    unsigned long long *scan_agg, *scan_pre;
    int32_t scan_chunks, pad3;
    const int32_t *rootnodes, *ps, *psof_off, *psof;
    int32_t n_rootnodes;
    unsigned long long *key;
    const int32_t *cs_off, *cs_wt;
    const int16_t *cs_list;
    const float4 *mean4, *prec4;
    const float *lrd;
    const int32_t *mixw;
    const uint16_t *tab16;
    uint32_t tab_size;
    int32_t lm_zero;
    double f, distfloor;
    int32_t D4, CP, Gpad, n_sen, n_ci_sen;
    const uint8_t *ncomp;
    const int16_t *cd2cisen;
    uint8_t *sen_act;
    int32_t *scr, *misc, *bstidx, *bstscr, *updatetime;
    int32_t *gpart;         /* [3][gp_n]: per-workgroup maxima / counters of kb_gated_cd_multi */
    int32_t gp_n, pad1;
};

struct BFrame {             /* one decoder's parameters for one step */
    float feat[64];         /* first: 16-byte aligned for the kernels' float4 reads */
    int32_t slot, cur;      /* cur = index of the list searched in this step */
    int32_t cf, thresh, n_calls, n_ent, n_groups, pad0, groups[8];
    int32_t frm, may_hist;
    FrameBeams bm;
    int32_t sc_frame, sc_beam, sc_is_skip, mark_rows;   /* mark_rows: bound on the list lengths before the entries */
    int32_t gpart_n, scan_epoch, scan_nc, pad2;     /* scan_nc: k_dec_scan workgroups per tree this decoder needs */   /* > 0: this step's CD maxima / counters are in the slot's gpart[] */
    int32_t calls[4 * BMAXC];
};

/* ------------------------------------------------------------------ */
#define SLOT_FRAME const BFrame &f = frames[blockIdx.z]; const BSlot &s = slots[f.slot]

__global__ void
kb_enter1(const BSlot *__restrict__ slots, const BFrame *__restrict__ frames)
{
    SLOT_FRAME;
    if ((int32_t)(blockIdx.x * blockDim.x) >= f.n_ent) return;
    const Entries ent = { f.calls, s.rootlist, f.n_calls };
    d_dec_enter1(ent, f.n_ent, f.calls, s.prob, s.sc, f.thresh, s.key, s.first, blockIdx.x, 0);
}

__global__ void __launch_bounds__(SCAN_THREADS)
kb_enter2(const BSlot *__restrict__ slots, const BFrame *__restrict__ frames)
{
    SLOT_FRAME;
    if ((int32_t)blockIdx.x >= f.n_calls || f.n_ent == 0) return;
    const Entries ent = { f.calls, s.rootlist, f.n_calls };
    d_dec_enter2(ent, f.n_ent, f.calls, s.prob, s.sc, s.frame, s.first, f.thresh, f.cf + 1, s.T, s.nact[f.cur],
                 s.eflag, s.ctot, s.n0, blockIdx.x, 0);
}

__global__ void __launch_bounds__(M3BLOCK)
kb_enter3_mark(const BSlot *__restrict__ slots, const BFrame *__restrict__ frames)
{
    SLOT_FRAME;
    const int32_t n_ent_blocks = (f.n_ent + M3BLOCK - 1) / M3BLOCK, bpt = (f.mark_rows + M3BLOCK - 1) / M3BLOCK;
    if ((int32_t)blockIdx.x >= n_ent_blocks + bpt * s.T) return;
    const Entries ent = { f.calls, s.rootlist, f.n_calls };
    d_dec_enter3_mark(n_ent_blocks, ent, f.n_ent, f.calls, f.groups, f.n_groups, f.cf + 1, s.key, s.first, s.eflag,
                      s.ctot, f.n_ent > 0 ? s.n0 : s.nact[f.cur], s.sc, s.hist, s.frame, s.T, bpt, s.node_base,
                      s.act[f.cur], s.nact[f.cur], s.pos, s.posf, s.ssid, s.comp, s.sseq, s.comsseq, s.cs_off,
                      s.cs_list, s.sen_act, blockIdx.x, 0);
}

#define KB_GATED_ARGS s.mean4, s.prec4, s.lrd, s.mixw, s.tab16, s.tab_size, s.lm_zero, s.f, s.distfloor,            \
                         f.feat, s.D4, s.CP, s.Gpad, lo, hi, CI ? 1 : 0, s.ncomp, s.cd2cisen, s.sen_act, s.scr,     \
                         0, CI ? (const int32_t *)NULL : s.misc + 5, CI ? 0 : f.sc_beam, f.sc_frame,              \
                         CI ? 0 : f.sc_is_skip, s.bstidx, s.bstscr, s.updatetime, s.misc, CI ? 5 : 0,             \
                         CI ? (uint8_t *)NULL : s.sen_act, CI ? (int32_t *)NULL : s.gpart, s.gp_n
template <bool EXACT, bool CI>
__global__ void __launch_bounds__(256)
kb_gated(const BSlot *__restrict__ slots, const BFrame *__restrict__ frames)
{
    SLOT_FRAME;
    const int32_t lo = CI ? 0 : s.n_ci_sen, hi = CI ? s.n_ci_sen : s.n_sen;
    if ((int32_t)(blockIdx.x * 256) >= (hi - lo) * s.CP) return;
    if (s.D4 == D4MAIN)
        d_gated_frame<EXACT, D4MAIN>(KB_GATED_ARGS, blockIdx.x);
    else
        d_gated_frame<EXACT, 0>(KB_GATED_ARGS, blockIdx.x);
}

/*
 * The gated CD senones of the step's decoders when they share ONE model (the usual case: one
 * acoustic model, many utterances).  A wave holds the CP Gaussians of one senone for 64/CP
 * DECODERS side by side: lanes with the same Gaussian read the same parameter address, which the
 * memory pipeline serves with one request, so the model streams through the caches once per
 * 64/CP decoders instead of once per decoder, at the same parallelism (a lane per (Gaussian,
 * decoder) pair).  Gate, distance, ordered log-add: the code of d_gated_frame (s3a_gated.h),
 * bit-identical results; the per-decoder maxima / counters are reduced per 8-lane group, then
 * per workgroup in LDS, before they touch the decoders' misc[] words.
 * (A first version kept the parameters in registers and LOOPED over the decoders: with < 1 wave
 * per SIMD every decoder's gate -> distance -> log-add chain was exposed latency; 78 us vs 45.)
 */
#define GX_THREADS 512      /* 256 VGPRs per lane: the Gaussian (80) stays in registers without spilling */
#define GX_SEN (GX_THREADS / 64)    /* senones per tile: one per wave */
struct GxDec {              /* one decoder of the workgroup, staged in LDS */
    uint8_t *sen_act;
    int32_t *scr, *misc, *bstidx, *bstscr, *updatetime;    /* misc: the scorer's counters (kb_gated_cd_shared) or
                                                            * its per-workgroup columns gpart[] (kb_gated_cd_multi) */
    int32_t frame, is_skip, thresh, gp_n;   /* thresh = the CI maximum + the CI beam; gp_n = columns of gpart[] */
};
/*
 * D4C > 0: the Gaussian is fetched before the gate is known and the gate only selects (see s3a_gated.h).
 * A workgroup walks several senone tiles (grid.x of them apart): the table is staged once, and the
 * decoders' maxima / counters leave the workgroup as ONE set of atomics per decoder -- their words share
 * a cache line per decoder and read-modify-writes of a line are served one after another, which is what
 * this kernel's time was made of when every tile sent its own.
 */
template <bool EXACT, int D4C>
__global__ void __launch_bounds__(GX_THREADS)
kb_gated_cd_shared(const BSlot *__restrict__ slots, const BFrame *__restrict__ frames, int32_t n)
{
    typedef typename Acc<EXACT>::T acc_t;
    constexpr int NK = D4C > 0 ? D4C : 1;
    __shared__ int32_t red[64][4];              /* per decoder: best, #senones, #Gaussians */
    __shared__ GxDec dec[64];
    /* dynamic: [features of this workgroup's decoders: 64 / CP rows of 64 + 4][the log-add table: the
     * ordered log-add is a chain of dependent look-ups, from LDS instead of L2] */
    extern __shared__ __attribute__((aligned(16))) unsigned char gx_smem[];
    const BSlot &s0 = slots[frames[0].slot];
    const int32_t CP = s0.CP, D4 = s0.D4, Gpad = s0.Gpad, per_wave = 64 / CP;
    float (*xs)[64 + 4] = (float (*)[64 + 4])gx_smem;
    uint16_t *tab_s = (uint16_t *)(gx_smem + (size_t)per_wave * (64 + 4) * sizeof(float));
    const int32_t z0 = blockIdx.y * per_wave;
    const int32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int32_t q = lane / CP, c = lane - q * CP, z = z0 + q;
    const int32_t n_tiles = (s0.n_sen - s0.n_ci_sen + GX_SEN - 1) / GX_SEN;
    int32_t tile = blockIdx.x;
    int32_t sen = s0.n_ci_sen + tile * GX_SEN + wave;
    bool valid = tile < n_tiles && sen < s0.n_sen && z < n;
    int32_t g = sen * CP + c;

    /* ---- everything whose address is known now: the lane's Gaussian, the senone's constants ---- */
    float4 M[NK], P[NK];
    float lrd_g = 0.0f;
    int32_t mixw = 0, nc = 0, ci_id = 0;
#define GX_FETCH                                                                                     \
    if (valid) {                                                                                     \
        if (D4C > 0) {                                                                               \
            _Pragma("unroll")                                                                        \
            for (int k = 0; k < NK; k++) {                                                           \
                M[k] = s0.mean4[(size_t)k * Gpad + g];                                               \
                P[k] = s0.prec4[(size_t)k * Gpad + g];                                               \
            }                                                                                        \
            lrd_g = s0.lrd[g];                                                                       \
            mixw = s0.mixw[g];                                                                       \
        }                                                                                            \
        nc = (int32_t)s0.ncomp[sen];                                                                 \
        ci_id = s0.cd2cisen[sen];                                                                    \
    }                                                                                                \
    else if (D4C > 0) {                                                                              \
        _Pragma("unroll")                                                                            \
        for (int k = 0; k < NK; k++) M[k] = P[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);             \
    }
    GX_FETCH
    /* ---- while those are in flight: features, the decoders' pointers, the table ---- */
    for (int32_t i = threadIdx.x; i < per_wave * 64; i += GX_THREADS) {
        const int32_t qq = i >> 6, k = i & 63;
        xs[qq][k] = (z0 + qq < n && k < D4 * 4) ? frames[z0 + qq].feat[k] : 0.0f;
    }
    if (threadIdx.x < 64) { red[threadIdx.x][0] = INT_MIN; red[threadIdx.x][1] = 0; red[threadIdx.x][2] = 0; }
    if ((int32_t)threadIdx.x < per_wave && z0 + (int32_t)threadIdx.x < n) {
        const BFrame &f = frames[z0 + threadIdx.x];
        const BSlot &sp = slots[f.slot];
        GxDec d;
        d.sen_act = sp.sen_act; d.scr = sp.scr; d.misc = sp.misc; d.bstidx = sp.bstidx; d.bstscr = sp.bstscr;
        d.updatetime = sp.updatetime; d.frame = f.sc_frame; d.is_skip = f.sc_is_skip;
        d.thresh = (int32_t)((uint32_t)sp.misc[5] + (uint32_t)f.sc_beam); d.gp_n = 0;
        dec[threadIdx.x] = d;
    }
    {
        const int32_t n16 = (int32_t)((s0.tab_size * 2 + 15) >> 4);     /* padded to 8 entries by the host */
        for (int32_t i = threadIdx.x; i < n16; i += GX_THREADS)
            ((uint4 *)tab_s)[i] = ((const uint4 *)s0.tab16)[i];
    }
    __syncthreads();
    LogAdd la;
    la.tab = tab_s; la.size = s0.tab_size; la.zero = s0.lm_zero;
    GxDec d;
    if (z < n) d = dec[q];
    int32_t rbest = INT_MIN, rns = 0, rng = 0;      /* lane c == 0 of each decoder accumulates */
    for (;;) {
        /* ---- the gate: one round trip ---- */
        int32_t mode = 0, ci_scr = 0, bi = S3A_NO_BSTIDX;
        if (valid) {
            const int32_t act = d.sen_act[sen], ut = d.updatetime[sen];
            ci_scr = d.scr[ci_id];
            bi = d.bstidx[sen];
            if (act) {
                if (ci_scr >= d.thresh)
                    mode = 1;
                else
                    mode = (bi == S3A_NO_BSTIDX || ut != d.frame - 1) ? 3 : 2;
            }
        }
        if (mode != 2) bi = S3A_NO_BSTIDX;
        const bool wanted = mode == 1 || (mode == 2 && c == bi);
        int32_t gs = S3A_LOGPROB_ZERO;
        if (D4C > 0) {
            acc_t a = (acc_t)lrd_g;
            const float *x = xs[q];
#pragma unroll
            for (int k = 0; k < NK; k++) {
                const float4 xv = *(const float4 *)(x + 4 * k);
                a = Acc<EXACT>::step(a, xv.x, M[k].x, P[k].x);
                a = Acc<EXACT>::step(a, xv.y, M[k].y, P[k].y);
                a = Acc<EXACT>::step(a, xv.z, M[k].z, P[k].z);
                a = Acc<EXACT>::step(a, xv.w, M[k].w, P[k].w);
            }
            if (wanted) gs = gau_to_int((double)a, s0.f, s0.distfloor, mixw);
        }
        else if (wanted) {
            acc_t a = (acc_t)s0.lrd[g];
            const float *x = xs[q];
            for (int32_t k = 0; k < D4; k++) {
                const float4 m = s0.mean4[(size_t)k * Gpad + g], p = s0.prec4[(size_t)k * Gpad + g];
                const float4 xv = *(const float4 *)(x + 4 * k);
                a = Acc<EXACT>::step(a, xv.x, m.x, p.x);
                a = Acc<EXACT>::step(a, xv.y, m.y, p.y);
                a = Acc<EXACT>::step(a, xv.z, m.z, p.z);
                a = Acc<EXACT>::step(a, xv.w, m.w, p.w);
            }
            gs = gau_to_int((double)a, s0.f, s0.distfloor, s0.mixw[g]);
        }
        int32_t score = S3A_LOGPROB_ZERO, bs = S3A_LOGPROB_ZERO, bidx = S3A_NO_BSTIDX;
        for (int32_t cc = 0; cc < CP; cc++) {
            const int32_t v = __shfl(gs, q * CP + cc, 64);
            if (mode == 1 && cc < nc) {
                score = la(score, v);
                if (v > bs) { bs = v; bidx = cc; }
            }
            else if (mode == 2 && cc == bi) {
                score = la(score, v);
                if (v > bs) { bs = v; bidx = cc; }
            }
        }
        if (score <= S3A_LOGPROB_ZERO) score = S3A_LOGPROB_ZERO;
        if (mode == 3) score = ci_scr;
        if (valid && c == 0) {
            d.sen_act[sen] = 0;                             /* the mask is consumed */
            if (mode != 0) {
                d.scr[sen] = score;
                rbest = max(rbest, score);
                if (mode == 1) {
                    d.bstidx[sen] = bidx; d.bstscr[sen] = bs; d.updatetime[sen] = d.frame;
                    rns++; rng += nc;
                }
                else if (mode == 2) {
                    if (d.is_skip) { d.bstidx[sen] = bidx; d.bstscr[sen] = bs; d.updatetime[sen] = d.frame; }
                    rng++;
                }
            }
        }
        tile += gridDim.x;
        if (tile >= n_tiles) break;         /* (workgroup-uniform) */
        sen = s0.n_ci_sen + tile * GX_SEN + wave;
        valid = sen < s0.n_sen && z < n;
        g = sen * CP + c;
        GX_FETCH
    }
#undef GX_FETCH
    if (c == 0 && z < n) {
        if (rbest != INT_MIN) atomicMax(&red[q][0], rbest);
        if (rns) atomicAdd(&red[q][1], rns);
        if (rng) atomicAdd(&red[q][2], rng);
    }
    __syncthreads();
    if ((int32_t)threadIdx.x < per_wave && z0 + (int32_t)threadIdx.x < n) {
        int32_t *misc = dec[threadIdx.x].misc;
        if (red[threadIdx.x][0] != INT_MIN) atomicMax(&misc[0], red[threadIdx.x][0]);
        if (red[threadIdx.x][1]) atomicAdd(&misc[1], red[threadIdx.x][1]);
        if (red[threadIdx.x][2]) atomicAdd(&misc[2], red[threadIdx.x][2]);
    }
}

/*
 * The same, model-stationary: the multi-frame scoring kernel (k_score_frames, s3a_device.hip) with
 * the decoders of the step in the place of the frames.  A lane keeps ONE Gaussian in registers
 * and evaluates it for a group of GM_FB decoders (features broadcast from LDS), the values are
 * transposed through LDS, and lane (senone, c) then runs the gate and the ordered log-add of
 * decoder c of the group for its senone -- so the model is read once per GM_FB decoders with
 * fully coalesced loads, at 8 x the arithmetic per byte.  Every Gaussian is computed, the gate
 * selects (see s3a_gated.h).  The per-decoder maxima / counters leave the workgroup as plain
 * stores into the decoder's gpart[] (one column per workgroup), merged by the consumers
 * (d_dec_hmm_eval: the normaliser; d_dec_scan: the frame record): atomics on the decoders'
 * misc[] words -- one cache line per decoder -- were served one after another and cost more
 * than the scoring.
 */
#define GM_FB 8             /* decoders per group (= accumulators per lane) */
#define GM_MAXDEC 32        /* decoders per launch (LDS: features + descriptors) */
template <bool EXACT>
__global__ void __launch_bounds__(256)
kb_gated_cd_multi(const BSlot *__restrict__ slots, const BFrame *__restrict__ frames, int32_t n)
{
    typedef typename Acc<EXACT>::T acc_t;
    __shared__ GxDec dec[GM_MAXDEC];
    __shared__ int32_t red[4][GM_MAXDEC][3];    /* per wave, per decoder: best, #senones, #Gaussians */
    __shared__ int32_t tr_s[4][GM_FB * 65];     /* per wave: [decoder of the group][lane] */
    __shared__ float4 xs4[GM_MAXDEC * D4MAIN];  /* the decoders' features */
    const BSlot &s0 = slots[frames[0].slot];
    const int32_t CP = s0.CP, Gpad = s0.Gpad;
    const int32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t g = s0.n_ci_sen * CP + blockIdx.x * 256 + tid;
    const int32_t sen = g / CP, c = g - sen * CP, sl = lane / CP;
    const bool valid = sen < s0.n_sen;

    /* ---- the lane's Gaussian + the senone's constants ---- */
    float4 M[D4MAIN], P[D4MAIN];
    float lrd_g = 0.0f;
    int32_t mixw = 0, nc = 0, ci_id = 0;
    if (valid) {
#pragma unroll
        for (int k = 0; k < D4MAIN; k++) {
            M[k] = s0.mean4[(size_t)k * Gpad + g];
            P[k] = s0.prec4[(size_t)k * Gpad + g];
        }
        lrd_g = s0.lrd[g];
        mixw = s0.mixw[g];
        nc = (int32_t)s0.ncomp[sen];
        ci_id = s0.cd2cisen[sen];
    }
    else {
#pragma unroll
        for (int k = 0; k < D4MAIN; k++) M[k] = P[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    /* ---- while those are in flight: features, the decoders' descriptors, the table ---- */
    for (int32_t i = tid; i < GM_MAXDEC * D4MAIN * 4; i += 256) {
        const int32_t zz = i / (D4MAIN * 4), k = i - zz * (D4MAIN * 4);
        ((float *)xs4)[i] = zz < n ? frames[zz].feat[k] : 0.0f;
    }
    for (int32_t i = tid; i < 4 * GM_MAXDEC; i += 256) {
        red[i / GM_MAXDEC][i % GM_MAXDEC][0] = INT_MIN; red[i / GM_MAXDEC][i % GM_MAXDEC][1] = 0;
        red[i / GM_MAXDEC][i % GM_MAXDEC][2] = 0;
    }
    if (tid < n) {
        const BFrame &f = frames[tid];
        const BSlot &sp = slots[f.slot];
        GxDec d;
        d.sen_act = sp.sen_act; d.scr = sp.scr; d.misc = sp.gpart; d.bstidx = sp.bstidx; d.bstscr = sp.bstscr;
        d.updatetime = sp.updatetime; d.frame = f.sc_frame; d.is_skip = f.sc_is_skip;
        d.thresh = (int32_t)((uint32_t)sp.misc[5] + (uint32_t)f.sc_beam); d.gp_n = sp.gp_n;
        dec[tid] = d;
    }
    __syncthreads();
    LogAdd la;              /* (the table from global memory: see s3a_gated.h) */
    la.tab = s0.tab16; la.size = s0.tab_size; la.zero = s0.lm_zero;
    int32_t *tr = tr_s[wave];
    const int32_t n_groups = (n + GM_FB - 1) / GM_FB;
    for (int32_t grp = blockIdx.y; grp < n_groups; grp += gridDim.y) {
        const int32_t z0 = grp * GM_FB, nd = min(GM_FB, n - z0);
        /* ---- the gate of (this senone, decoder c of the group): in flight during the arithmetic ---- */
        const bool mine = valid && c < nd;
        int32_t act = 0, ut = 0, ci_scr = 0, bi = S3A_NO_BSTIDX;
        GxDec d;
        if (mine) {
            d = dec[z0 + c];
            act = d.sen_act[sen]; ut = d.updatetime[sen];
            ci_scr = d.scr[ci_id];
            bi = d.bstidx[sen];
        }
        /* ---- the lane's Gaussian for every decoder of the group ---- */
        acc_t a[GM_FB];
#pragma unroll
        for (int j = 0; j < GM_FB; j++) a[j] = (acc_t)lrd_g;
#pragma unroll
        for (int k = 0; k < D4MAIN; k++) {
#pragma unroll
            for (int j = 0; j < GM_FB; j++) {
                if (j < nd) {
                    const float4 x = xs4[(z0 + j) * D4MAIN + k];        /* wave-uniform: LDS broadcast */
                    a[j] = Acc<EXACT>::step(a[j], x.x, M[k].x, P[k].x);
                    a[j] = Acc<EXACT>::step(a[j], x.y, M[k].y, P[k].y);
                    a[j] = Acc<EXACT>::step(a[j], x.z, M[k].z, P[k].z);
                    a[j] = Acc<EXACT>::step(a[j], x.w, M[k].w, P[k].w);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < GM_FB; j++)
            if (j < nd) tr[j * 65 + lane] = gau_to_int((double)a[j], s0.f, s0.distfloor, mixw);
        /* same-wave LDS hand-off: no barrier needed, but order the accesses */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        /* ---- gate + ordered log-add: 0 = untouched, 1 = full, 2 = single Gaussian, 3 = CI copy ---- */
        int32_t mode = 0;
        if (mine && act) {
            if (ci_scr >= d.thresh)
                mode = 1;
            else
                mode = (bi == S3A_NO_BSTIDX || ut != d.frame - 1) ? 3 : 2;
        }
        int32_t score = S3A_LOGPROB_ZERO, bs = S3A_LOGPROB_ZERO, bidx = S3A_NO_BSTIDX;
        const int32_t *row = tr + c * 65 + sl * CP;     /* (only read when mine: c < nd <= GM_FB) */
        if (mode == 1) {
            for (int32_t cc = 0; cc < nc; cc++) {
                const int32_t v = row[cc];
                score = la(score, v);
                if (v > bs) { bs = v; bidx = cc; }      /* update_best_id = 1: strict >, first max wins */
            }
        }
        else if (mode == 2) {
            const int32_t v = row[bi];
            score = la(score, v);
            if (v > bs) { bs = v; bidx = bi; }
        }
        if (score <= S3A_LOGPROB_ZERO) score = S3A_LOGPROB_ZERO;
        if (mode == 3) score = ci_scr;
        int32_t rbest = INT_MIN, rns = 0, rng = 0;
        if (mine) {
            d.sen_act[sen] = 0;                             /* the mask is consumed */
            if (mode != 0) {
                d.scr[sen] = score;
                rbest = score;
                if (mode == 1) {
                    d.bstidx[sen] = bidx; d.bstscr[sen] = bs; d.updatetime[sen] = d.frame;
                    rns = 1; rng = nc;
                }
                else if (mode == 2) {
                    if (d.is_skip) { d.bstidx[sen] = bidx; d.bstscr[sen] = bs; d.updatetime[sen] = d.frame; }
                    rng = 1;
                }
            }
        }
        /* over the wave's senones (lanes with the same c) */
        for (int32_t o = CP; o < 64; o <<= 1) {
            rbest = max(rbest, __shfl_xor(rbest, o, 64));
            rns += __shfl_xor(rns, o, 64);
            rng += __shfl_xor(rng, o, 64);
        }
        if (sl == 0 && c < nd) { red[wave][z0 + c][0] = rbest; red[wave][z0 + c][1] = rns; red[wave][z0 + c][2] = rng; }
        __builtin_amdgcn_wave_barrier();                    /* tr is rewritten by the next group */
    }
    __syncthreads();
    /* this workgroup's column of every decoder it served */
    if (tid < n && (tid / GM_FB) % (int32_t)gridDim.y == (int32_t)blockIdx.y) {
        int32_t *gp = dec[tid].misc;
        const int32_t gp_n = dec[tid].gp_n;
        gp[blockIdx.x] = max(max(red[0][tid][0], red[1][tid][0]), max(red[2][tid][0], red[3][tid][0]));
        gp[gp_n + blockIdx.x] = red[0][tid][1] + red[1][tid][1] + red[2][tid][1] + red[3][tid][1];
        gp[2 * gp_n + blockIdx.x] = red[0][tid][2] + red[1][tid][2] + red[2][tid][2] + red[3][tid][2];
    }
}

template <int EB>
__global__ void __launch_bounds__(EB)
kb_hmm_eval(const BSlot *__restrict__ slots, const BFrame *__restrict__ frames)
{
    SLOT_FRAME;
    if ((int32_t)blockIdx.y >= s.T || (int32_t)(blockIdx.x * EB) >= s.maxn) return;   /* (grid sized by the host bound) */
    d_dec_hmm_eval<EB>(s.node_base, s.act[f.cur], s.nact[f.cur], s.N, s.n_tmat, s.ssid, s.tmatid, s.wid, s.comp,
                   s.tp, s.sseq, s.comsseq, s.cs_off, s.cs_list, s.cs_wt, s.scr, s.misc, s.sc, s.hist, s.outs,
                   s.outh, s.bests, s.best, f.frm, s.psof_off, s.psof, s.pstamp, s.gpart, f.gpart_n ? s.gp_n : 0, s.poswid, s.posout, blockIdx.x, blockIdx.y);
}

__global__ void __launch_bounds__(DBLOCK)
kb_hist_count(const BSlot *__restrict__ slots, const BFrame *__restrict__ frames)
{
    SLOT_FRAME;
    if (!f.may_hist || (int32_t)blockIdx.y >= s.T || (int32_t)(blockIdx.x * DBLOCK) >= s.maxn) return;
    d_dec_hist_count(s.node_base, s.act[f.cur], s.nact[f.cur], s.T, f.bm, s.best, s.bests, s.exits + s.N, s.hbin,
                     -1, 0, 1, NBIN, blockIdx.x, blockIdx.y);
}

__global__ void __launch_bounds__(SCAN_THREADS)
kb_hist_sort(const BSlot *__restrict__ slots, const BFrame *__restrict__ frames)
{
    SLOT_FRAME;
    if (!f.may_hist || (int32_t)blockIdx.x >= s.T) return;
    d_dec_hist_sort(s.node_base, s.act[f.cur], s.nact[f.cur], s.T, f.bm, s.exits + s.N, s.exits, s.hbin, s.pos, -1,
                    NBIN, blockIdx.x, 0);
}

__global__ void __launch_bounds__(RSBLOCK)
kb_resolve(const BSlot *__restrict__ slots, const BFrame *__restrict__ frames)
{
    SLOT_FRAME;
    if ((int32_t)(blockIdx.x * RSBLOCK) >= s.N) return;
    d_dec_resolve(s.N, s.T, f.frm, f.bm, s.best, s.nact[f.cur], s.node_base, s.tree_of, s.prob, s.par_off, s.par,
                  s.pos, s.posf, s.sc, s.hist, s.outs, s.outh, s.bests, s.frame, s.turn, s.selfemit, s.cnt, s.key,
                  s.first, s.hbin, s.ps, s.pstamp, s.rootnodes, s.n_rootnodes, s.propf, s.posout, blockIdx.x, 0);
}


__global__ void __launch_bounds__(SCAN_THREADS)
kb_weak(const BSlot *__restrict__ slots, const BFrame *__restrict__ frames)
{
    SLOT_FRAME;
    if (!(f.bm.phone_uses_wbeam || f.bm.pbeam < f.bm.hmmbeam)) return;
    d_dec_weak(s.N, s.T, f.frm, f.bm, s.best, s.nact[f.cur], s.node_base, s.act[f.cur], s.prob, s.par_off, s.par, s.pos,
               s.posf, s.sc, s.outs, s.bests, s.wid, s.hbin, s.propf, s.exits + 2 * (size_t)s.N, 0, 0);
}

__global__ void __launch_bounds__(SCAN_THREADS)
kb_scan(const BSlot *__restrict__ slots, const BFrame *__restrict__ frames, int32_t *pack_all,
        int32_t pack_stride, int32_t max_exits, int32_t NC)
{
    SLOT_FRAME;
    /* the launch has NC workgroups per tree (the largest bound of the step); this decoder needs f.scan_nc of them */
    const int32_t bt = blockIdx.x / NC, bj = blockIdx.x - bt * NC;
    if (bt >= s.T || bj >= f.scan_nc) return;
    d_dec_scan(s.N, s.T, f.frm, f.bm, s.node_base, s.act[f.cur], s.nact[f.cur], s.wid, s.prob, s.outs, s.outh,
               s.selfemit, s.cnt, s.base, s.act[f.cur ^ 1], s.nact[f.cur ^ 1], s.pos, s.posf, s.best, s.exits,
               s.nexit, s.hbin, s.misc, s.done, pack_all + (size_t)blockIdx.z * pack_stride, max_exits,
               s.gpart, f.gpart_n ? s.gp_n : 0, s.poswid, s.posout, f.may_hist, s.scan_agg, s.scan_pre, s.scan_flag,
               s.scan_chunks, f.scan_epoch, f.scan_nc, f.scan_nc, bt * f.scan_nc + bj, 0);
}

__global__ void __launch_bounds__(DBLOCK)
kb_emit(const BSlot *__restrict__ slots, const BFrame *__restrict__ frames)
{
    SLOT_FRAME;
    if ((int32_t)blockIdx.y >= s.T) return;
    d_dec_emit(f.frm, s.node_base, s.act[f.cur], s.nact[f.cur], s.child_off, s.child, s.turn, s.selfemit, s.base,
               s.act[f.cur ^ 1], s.nact[f.cur ^ 1], s.pos, s.posf, blockIdx.x, blockIdx.y);
}

/* ------------------------------------------------------------------ */
struct BOut {
    s3a_frame_result_t *res;
    int32_t *n_exit, *wid, *scr, *hist, max_exits, frm, may_hist, rc;
    char err[256];
};

struct s3a_batch_s {
    int32_t max_slots, n_slots;
    int opt_scan_chained;               /* S3A_SCAN_CHAINED when the batch was created (tests) */
    s3a_lexsearch_t *ls[BMAXSLOT];
    s3a_scorer_t *sc[BMAXSLOT];
    s3a_comsen_t *cs[BMAXSLOT];
    BSlot *d_slots;
    BFrame *d_frames, *h_frames;        /* device / pinned host, [max_slots] */
    BFrame stage[BMAXSLOT];             /* per slot: pending transition + this frame's request */
    BOut out[BMAXSLOT];
    uint8_t has_trans[BMAXSLOT], active[BMAXSLOT], arrived[BMAXSLOT];
    int32_t n_active, n_arrived, order[BMAXSLOT], rows[BMAXSLOT], zof[BMAXSLOT], last_order[BMAXSLOT], last_n;
    int32_t *h_pack, pack_stride, pack_max_exits, hdr_max;     /* pinned host: the kernels write the frame records there */
    int32_t g_ent, g_ci, g_cd, g_maxn, g_N, g_T, g_mark, g_tmat, exact;
    unsigned long long gen;
    long steps, slot_frames;
    hipStream_t stream;
    hipEvent_t ev;
    pthread_mutex_t mu;
    pthread_cond_t cv;
};

extern "C" s3a_batch_t *
s3a_batch_create(int32_t max_slots)
{
    if (max_slots <= 0 || max_slots > BMAXSLOT) { s3a_set_error("s3a_batch_create: 1..%d slots", BMAXSLOT); return NULL; }
    s3a_batch_t *b = new s3a_batch_s();
    memset((void *)b, 0, sizeof *b);
    b->max_slots = max_slots;
    b->opt_scan_chained = s3a_variants()->scan_chained != 0;
    pthread_mutex_init(&b->mu, NULL);
    pthread_cond_init(&b->cv, NULL);
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess
        || hipEventCreateWithFlags(&b->ev, hipEventDisableTiming) != hipSuccess
        || hipMalloc((void **)&b->d_slots, sizeof(BSlot) * max_slots) != hipSuccess
        || hipMalloc((void **)&b->d_frames, sizeof(BFrame) * max_slots) != hipSuccess
        || hipHostMalloc((void **)&b->h_frames, sizeof(BFrame) * max_slots) != hipSuccess) {
        s3a_set_error("s3a_batch_create: no usable HIP device (libcmusphinx_amd has no CPU fallback)");
        delete b;
        return NULL;
    }
    return b;
}

extern "C" void
s3a_batch_free(s3a_batch_t *b)
{
    if (!b) return;
    (void)hipStreamSynchronize(b->stream);
    (void)hipEventDestroy(b->ev);
    (void)hipFree(b->d_slots); (void)hipFree(b->d_frames); (void)hipHostFree(b->h_frames);
    if (b->h_pack) (void)hipHostFree(b->h_pack);

    /* the attached decoders now own a dead stream handle: they must be freed by their owners
     * WITHOUT further use; their own streams were replaced at attach time */
    pthread_mutex_destroy(&b->mu);
    pthread_cond_destroy(&b->cv);
    delete b;
}

/* Attach a decoder (its lextrees, scorer and composite table).  All its work from now on is
 * ordered on the engine's stream. */
extern "C" int32_t
s3a_batch_attach(s3a_batch_t *b, s3a_lexsearch_t *ls, s3a_scorer_t *sc, s3a_comsen_t *cs)
{
    LS_NEED_3ST(ls, "s3a_batch_attach");
    if (!b || !ls || !sc || !cs) return S3A_EINVAL;
    pthread_mutex_lock(&b->mu);
    int32_t rc = S3A_OK, slot = b->n_slots;
    struct s3a_mgau_dev_s *d = sc->g->dev;
    do {
        if (slot >= b->max_slots) { s3a_set_error("s3a_batch_attach: all %d slots taken", b->max_slots); rc = S3A_EINVAL; break; }
        if (d->tab16 == NULL) { s3a_set_error("s3a_batch_attach: 32-bit log-add tables are not supported"); rc = S3A_EUNSUP; break; }
        if (d->D4 * 4 > 64) { s3a_set_error("s3a_batch_attach: feature vectors longer than 64 are not supported"); rc = S3A_EUNSUP; break; }
        if (sc->max_cd < sc->n_sen - sc->n_ci_sen) { s3a_set_error("s3a_batch_attach: -maxcdsenpf needs the host-pointer scorer"); rc = S3A_EUNSUP; break; }
        if (slot > 0 && (sc->g->precision == S3A_GMM_EXACT) != (b->exact != 0)) {
            s3a_set_error("s3a_batch_attach: all decoders must use the same GMM precision"); rc = S3A_EINVAL; break;
        }
        /* whatever the decoder enqueued on its own streams so far (uploads, resets) must be done */
        if (hipStreamSynchronize(ls->stream) != hipSuccess || hipStreamSynchronize(d->stream) != hipSuccess) { rc = S3A_EHIP; break; }
        ls->stream = b->stream;         /* (the replaced stream objects stay allocated: harmless) */
        ls->own_stream = 0;
        d->stream = b->stream;
        cs->stream = b->stream;
        int32_t maxn = 0;
        for (int32_t t = 0; t < ls->n_tree; t++) maxn = max(maxn, ls->node_base[t + 1] - ls->node_base[t]);
        BSlot s;
        memset((void *)&s, 0, sizeof s);
        s.N = ls->N; s.T = ls->n_tree; s.n_tmat = ls->n_tmat; s.maxn = maxn;
        s.node_base = ls->d_node_base; s.ssid = ls->d_ssid; s.tmatid = ls->d_tmatid; s.wid = ls->d_wid;
        s.prob = ls->d_prob; s.child_off = ls->d_child_off; s.child = ls->d_child; s.par_off = ls->d_par_off;
        s.par = ls->d_par; s.tree_of = ls->d_tree_of; s.rootlist = ls->d_rootlist; s.tp = ls->d_tp;
        s.comp = ls->d_comp; s.sseq = ls->d_sseq; s.comsseq = ls->d_comsseq;
        s.sc = ls->d_sc; s.hist = ls->d_hist; s.outs = ls->d_outs; s.outh = ls->d_outh; s.bests = ls->d_bests;
        s.frame = ls->d_frame; s.pos = ls->d_pos; s.posf = ls->d_posf; s.act[0] = ls->d_act[0]; s.act[1] = ls->d_act[1];
        s.nact[0] = ls->d_nact[0]; s.nact[1] = ls->d_nact[1]; s.turn = ls->d_turn; s.selfemit = ls->d_selfemit;
        s.cnt = ls->d_cnt; s.base = ls->d_cand; s.best = ls->d_best; s.exits = ls->d_exit; s.nexit = ls->d_nexit;
        s.first = ls->d_first; s.eflag = ls->d_eflag; s.hbin = ls->d_hbin; s.done = ls->d_done; s.key = ls->d_key;
        s.ctot = ls->d_ctot; s.n0 = ls->d_n0; s.pstamp = ls->d_pstamp; s.propf = ls->d_candf; s.rootnodes = ls->d_rootnodes;
        s.poswid = ls->d_poswid; s.posout = ls->d_posout;
        s.scan_agg = ls->d_scan_agg; s.scan_pre = ls->d_scan_pre; s.scan_flag = ls->d_scan_flag; s.scan_chunks = ls->scan_chunks;
        s.ps = ls->d_ps; s.psof_off = ls->d_psof_off; s.psof = ls->d_psof;
        s.n_rootnodes = ls->n_rootnodes;
        s.cs_off = cs->off_d; s.cs_wt = cs->wt_d; s.cs_list = cs->list_d;
        s.mean4 = d->mean4; s.prec4 = d->prec4; s.lrd = d->lrd; s.mixw = d->mixw; s.tab16 = d->tab16;
        s.tab_size = d->tab_size; s.lm_zero = d->lm_zero; s.f = sc->g->f; s.distfloor = sc->g->distfloor;
        s.D4 = d->D4; s.CP = d->CP; s.Gpad = d->Gpad; s.n_sen = sc->n_sen; s.n_ci_sen = sc->n_ci_sen;
        s.ncomp = sc->ncomp_d; s.cd2cisen = sc->cd2cisen_d; s.sen_act = sc->act_d; s.scr = sc->scr_d;
        s.misc = sc->misc_d; s.bstidx = sc->bstidx_d; s.bstscr = sc->bstscr_d; s.updatetime = sc->updatetime_d;
        s.gpart = sc->gpart_d; s.gp_n = sc->gp_n;
        if (hipMemcpy(b->d_slots + slot, &s, sizeof s, hipMemcpyHostToDevice) != hipSuccess) { rc = S3A_EHIP; break; }
        b->ls[slot] = ls; b->sc[slot] = sc; b->cs[slot] = cs;
        b->exact = sc->g->precision == S3A_GMM_EXACT;
        /* launch geometry = the maximum over the attached decoders */
        b->g_ci = max(b->g_ci, (s.n_ci_sen * s.CP + 255) / 256);
        b->g_cd = max(b->g_cd, ((s.n_sen - s.n_ci_sen) * s.CP + 255) / 256);
        b->g_maxn = max(b->g_maxn, maxn); b->g_N = max(b->g_N, s.N); b->g_T = max(b->g_T, s.T);
        b->g_ent = max(b->g_ent, (ls->ent_cap + 255) / 256);
        b->g_mark = max(b->g_mark, (ls->ent_cap + DBLOCK - 1) / DBLOCK + ((maxn + DBLOCK - 1) / DBLOCK) * s.T);
        b->g_tmat = max(b->g_tmat, s.n_tmat);
        const int32_t hdr = 6 * s.T + 16;
        if (hdr > b->hdr_max || ls->pack_max_exits > b->pack_max_exits) {
            if (b->h_pack) (void)hipHostFree(b->h_pack);
            b->hdr_max = max(b->hdr_max, hdr);
            b->pack_max_exits = max(b->pack_max_exits, ls->pack_max_exits);
            b->pack_stride = b->hdr_max + 3 * b->pack_max_exits;
            if (hipHostMalloc((void **)&b->h_pack, (size_t)b->pack_stride * b->max_slots * 4, hipHostMallocCoherent) != hipSuccess) { rc = S3A_EHIP; break; }
        }
        b->n_slots++;
    } while (0);
    pthread_mutex_unlock(&b->mu);
    return rc == S3A_OK ? slot : rc;
}

/* run one step for every arrived decoder; called with the lock held */
static int32_t
run_batch(s3a_batch_t *b)
{
    const int32_t n = b->n_arrived;
    int32_t any_hist = 0, any_weak = 0, g_ent = 0, g_calls = 0, g_rows = 1, g_mark = 1, rc = S3A_OK;
    for (int32_t z = 0; z < n; z++) {
        const int32_t slot = b->order[z];
        b->h_frames[z] = b->stage[slot];
        any_hist |= b->stage[slot].may_hist;
        any_weak |= (b->stage[slot].bm.phone_uses_wbeam || b->stage[slot].bm.pbeam < b->stage[slot].bm.hmmbeam) ? 1 : 0;
        g_ent = max(g_ent, (b->stage[slot].n_ent + 255) / 256);
        g_calls = max(g_calls, b->stage[slot].n_calls);
        g_rows = max(g_rows, b->rows[slot]);
        g_mark = max(g_mark, b->stage[slot].mark_rows);
    }
#define CHK(expr) do { if ((expr) != hipSuccess) { s3a_set_error("s3a_batch: %s failed: %s", #expr, hipGetErrorString(hipGetLastError())); rc = S3A_EHIP; goto done; } } while (0)
    {
        hipStream_t st = b->stream;
        const BSlot *S = b->d_slots;
        const BFrame *F = b->d_frames;
        /* one model for every decoder of the step?  then the CD senones of all of them are one pass
         * over the model: kb_gated_cd_multi (39/40-dimensional features, >= GM_FB Gaussians per senone
         * slot), else kb_gated_cd_shared */
        const bool no_shared = s3a_variants()->batch_no_shared != 0, no_multi = s3a_variants()->batch_no_multi != 0;
        bool shared = n > 1 && n <= 64 && !no_shared;
        for (int32_t z = 0; z < n && shared; z++) {
            const s3a_scorer_t *sc = b->sc[b->order[z]], *sc0 = b->sc[b->order[0]];
            shared = sc->g == sc0->g && sc->n_sen == sc0->n_sen && sc->n_ci_sen == sc0->n_ci_sen
                && sc->cd2cisen_d != NULL && sc0->g->dev->D4 * 4 <= 64 && sc0->g->dev->CP <= 64;
        }
        const struct s3a_mgau_dev_s *d0 = b->sc[b->order[0]]->g->dev;
        /* (measured, hub4 shape: one launch per decoder -- kb_gated, grid z -- is as fast up to ~7 decoders:
         * 10 / 11 / 16 us for 2 / 4 / 7 against 11 / 13 / 17; the shared pass wins from there: 19 us for 13.5) */
        const bool multi = shared && n >= GM_FB && n <= GM_MAXDEC && d0->D4 == D4MAIN && d0->CP >= GM_FB
            && !no_multi;
        if (!multi && n < GM_FB) shared = false;
        /* (kb_gated_cd_shared, the fallback for other shapes, still merges with atomics) */
        for (int32_t z = 0; z < n; z++) b->h_frames[z].gpart_n = (multi || !shared) ? 1 : 0;
        CHK(hipMemcpyAsync(b->d_frames, b->h_frames, sizeof(BFrame) * n, hipMemcpyHostToDevice, st));
        if (g_ent > 0) {
            hipLaunchKernelGGL(kb_enter1, dim3(g_ent, 1, n), dim3(256), 0, st, S, F);
            hipLaunchKernelGGL(kb_enter2, dim3(g_calls, 1, n), dim3(SCAN_THREADS), 0, st, S, F);
        }
        hipLaunchKernelGGL(kb_enter3_mark, dim3(g_ent * (256 / M3BLOCK) + ((g_mark + M3BLOCK - 1) / M3BLOCK) * b->g_T, 1, n),
                           dim3(M3BLOCK), 0, st, S, F);
        dim3 gx_grid(1, 1, 1);
        size_t gx_lds = 0;
        const bool d4main = shared && b->sc[b->order[0]]->g->dev->D4 == D4MAIN;
        if (shared) {
            static bool attr_set = false;
            if (!attr_set) {        /* static + dynamic LDS exceed the 64 KB default */
                (void)hipFuncSetAttribute((const void *)kb_gated_cd_shared<true, D4MAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
                (void)hipFuncSetAttribute((const void *)kb_gated_cd_shared<false, D4MAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
                (void)hipFuncSetAttribute((const void *)kb_gated_cd_shared<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
                (void)hipFuncSetAttribute((const void *)kb_gated_cd_shared<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
                attr_set = true;
            }
            gx_lds = (((size_t)b->sc[b->order[0]]->g->dev->tab_size + 7) & ~(size_t)7) * 2
                + (size_t)(64 / b->sc[b->order[0]]->g->dev->CP) * (64 + 4) * sizeof(float);
            const s3a_scorer_t *sc0 = b->sc[b->order[0]];
            const int32_t per_wave = 64 / sc0->g->dev->CP;
            const int32_t n_tiles = (sc0->n_sen - sc0->n_ci_sen + GX_SEN - 1) / GX_SEN, ny = (n + per_wave - 1) / per_wave;
            /* two workgroups fit a CU (LDS): one round of them, each walking its share of the tiles */
            const int32_t per_row = max(1, 2 * sc0->g->dev->n_cu / ny);
            const int32_t walk = (n_tiles + per_row - 1) / per_row;
            gx_grid = dim3((n_tiles + walk - 1) / walk, ny, 1);
        }
        dim3 gm_grid(1, 1, 1);
        if (multi) {
            /* the groups of GM_FB decoders spread over grid.y as far as that keeps the launch within one round
             * of workgroups (two per CU), the rest is walked */
            const int32_t n_groups = (n + GM_FB - 1) / GM_FB;
            const int32_t gp_n = b->sc[b->order[0]]->gp_n;
            gm_grid = dim3(gp_n, max(1, min(n_groups, 2 * d0->n_cu / max(1, gp_n))), 1);
        }
        if (b->exact) {
            if (b->g_ci) hipLaunchKernelGGL((kb_gated<true, true>), dim3(b->g_ci, 1, n), dim3(256), 0, st, S, F);
            if (multi && gm_grid.x > 0)
                hipLaunchKernelGGL((kb_gated_cd_multi<true>), gm_grid, dim3(256), 0, st, S, F, n);
            else if (b->g_cd && shared && d4main)
                hipLaunchKernelGGL((kb_gated_cd_shared<true, D4MAIN>), gx_grid, dim3(GX_THREADS), gx_lds, st, S, F, n);
            else if (b->g_cd && shared)
                hipLaunchKernelGGL((kb_gated_cd_shared<true, 0>), gx_grid, dim3(GX_THREADS), gx_lds, st, S, F, n);
            else if (b->g_cd) hipLaunchKernelGGL((kb_gated<true, false>), dim3(b->g_cd, 1, n), dim3(256), 0, st, S, F);
        }
        else {
            if (b->g_ci) hipLaunchKernelGGL((kb_gated<false, true>), dim3(b->g_ci, 1, n), dim3(256), 0, st, S, F);
            if (multi && gm_grid.x > 0)
                hipLaunchKernelGGL((kb_gated_cd_multi<false>), gm_grid, dim3(256), 0, st, S, F, n);
            else if (b->g_cd && shared && d4main)
                hipLaunchKernelGGL((kb_gated_cd_shared<false, D4MAIN>), gx_grid, dim3(GX_THREADS), gx_lds, st, S, F, n);
            else if (b->g_cd && shared)
                hipLaunchKernelGGL((kb_gated_cd_shared<false, 0>), gx_grid, dim3(GX_THREADS), gx_lds, st, S, F, n);
            else if (b->g_cd) hipLaunchKernelGGL((kb_gated<false, false>), dim3(b->g_cd, 1, n), dim3(256), 0, st, S, F);
        }
        if (g_rows >= EVBLOCK_LONG_LIST)
            hipLaunchKernelGGL(kb_hmm_eval<256>, dim3((g_rows + 255) / 256, b->g_T, n), dim3(256),
                               0, st, S, F);
        else
            hipLaunchKernelGGL(kb_hmm_eval<64>, dim3((g_rows + 63) / 64, b->g_T, n), dim3(64),
                               0, st, S, F);
        if (any_hist) {
            hipLaunchKernelGGL(kb_hist_count, dim3((g_rows + DBLOCK - 1) / DBLOCK, b->g_T, n), dim3(DBLOCK), 0, st, S, F);
            hipLaunchKernelGGL(kb_hist_sort, dim3(b->g_T, 1, n), dim3(SCAN_THREADS), 0, st, S, F);
        }
        if (any_weak) hipLaunchKernelGGL(kb_weak, dim3(1, 1, n), dim3(SCAN_THREADS), 0, st, S, F);
        hipLaunchKernelGGL(kb_resolve, dim3((b->g_N + RSBLOCK - 1) / RSBLOCK, 1, n), dim3(RSBLOCK), 0, st, S, F);
        {
            const int32_t scan_nc = scan_workgroups(g_rows, b->opt_scan_chained);
            hipLaunchKernelGGL(kb_scan, dim3(b->g_T * scan_nc, 1, n), dim3(SCAN_THREADS), 0, st, S, F, b->h_pack,
                               b->pack_stride, b->pack_max_exits, scan_nc);
        }
        CHK(hipGetLastError());
        /* the records (header + every exit) were written by kb_scan's last workgroups straight into pinned host
         * memory; the hosts wait for that kernel only, so k_dec_emit overlaps their word-level work (the next
         * step's kernels follow it in stream order) */
        CHK(hipEventRecord(b->ev, st));
        hipLaunchKernelGGL(kb_emit, dim3(EMIT_BLOCKS, b->g_T, n), dim3(DBLOCK), 0, st, S, F);
        CHK(hipGetLastError());
        CHK(hipEventSynchronize(b->ev));
        /* every decoder unpacks its own record (in its own thread, in parallel) after the wake-up */
        for (int32_t z = 0; z < n; z++) { b->zof[b->order[z]] = z; b->last_order[z] = b->order[z]; }
        b->last_n = n;
    }
done:
    if (rc != S3A_OK)
        for (int32_t z = 0; z < n; z++) { BOut &o = b->out[b->order[z]]; o.rc = rc; strncpy(o.err, s3a_last_error(), sizeof o.err - 1); }
    b->steps++;
    b->slot_frames += n;
    for (int32_t z = 0; z < n; z++) b->arrived[b->order[z]] = 0;
    b->n_arrived = 0;
    __atomic_store_n(&b->gen, b->gen + 1, __ATOMIC_RELEASE);   /* the waiters spin on this word */
    pthread_cond_broadcast(&b->cv);
    return rc;
#undef CHK
}

/* the calling decoder's frame record of the last step -> its result structures */
static int32_t
unpack_own(s3a_batch_t *b, int32_t slot)
{
    BOut &o = b->out[slot];
    if (o.rc != S3A_OK) { s3a_set_error("%s", o.err); return o.rc; }
    s3a_lexsearch_t *ls = b->ls[slot];
    const int32_t z = b->zof[slot], hdr = 6 * ls->n_tree + 16;
    const int32_t *p = b->h_pack + (size_t)z * b->pack_stride;
    int32_t total = 0;
    int32_t rc = s3a_dec_unpack(ls, p, o.may_hist != 0, o.frm, o.res, o.n_exit, o.max_exits, &total);
    if (rc != S3A_OK) return rc;
    for (int32_t k = 0; k < total; k++) {
        o.wid[k] = p[hdr + 3 * k]; o.scr[k] = p[hdr + 3 * k + 1]; o.hist[k] = p[hdr + 3 * k + 2];
    }
    return S3A_OK;
}

/* srch_TST_begin's device side for one decoder; the decoder now takes part in the steps */
extern "C" int32_t
s3a_batch_utt_begin(s3a_batch_t *b, int32_t slot)
{
    if (!b || slot < 0 || slot >= b->n_slots) return S3A_EINVAL;
    pthread_mutex_lock(&b->mu);
    int32_t rc = s3a_decoder_utt_begin(b->ls[slot], b->sc[slot]);
    if (rc == S3A_OK && !b->active[slot]) { b->active[slot] = 1; b->n_active++; }
    b->has_trans[slot] = 0;
    pthread_mutex_unlock(&b->mu);
    return rc;
}

/* lextree_utt_end: the decoder leaves the steps (its last recorded transition is dropped: the
 * reference swaps and then clears those lists anyway) */
extern "C" int32_t
s3a_batch_utt_end(s3a_batch_t *b, int32_t slot)
{
    if (!b || slot < 0 || slot >= b->n_slots) return S3A_EINVAL;
    pthread_mutex_lock(&b->mu);
    int32_t rc = s3a_lexsearch_utt_end(b->ls[slot]);
    if (b->active[slot]) { b->active[slot] = 0; b->n_active--; }
    b->has_trans[slot] = 0;
    if (b->n_arrived > 0 && b->n_arrived == b->n_active)
        (void)run_batch(b);             /* the others were only waiting for this decoder */
    pthread_mutex_unlock(&b->mu);
    return rc;
}

/* record this frame's lextree_enter calls + the swap (host only; executed by the next step) */
extern "C" int32_t
s3a_batch_transition(s3a_batch_t *b, int32_t slot, int32_t cf, int32_t thresh, int32_t tree_a, int32_t n_a,
                     const int32_t *lc_a, const int32_t *scr_a, const int32_t *hist_a, int32_t tree_b,
                     int32_t n_b, const int32_t *lc_b, const int32_t *scr_b, const int32_t *hist_b)
{
    if (!b || slot < 0 || slot >= b->n_slots) return S3A_EINVAL;
    BFrame &f = b->stage[slot];
    if (b->has_trans[slot]) { s3a_set_error("s3a_batch_transition: two transitions without a step"); return S3A_EINVAL; }
    int32_t rc = s3a_dec_stage_calls(b->ls[slot], tree_a, n_a, lc_a, scr_a, hist_a, tree_b, n_b, lc_b, scr_b,
                                     hist_b, f.groups, f.calls, BMAXC, &f.n_calls, &f.n_ent, &f.n_groups);
    if (rc != S3A_OK) return rc;
    f.cf = cf; f.thresh = thresh;
    {
        s3a_lexsearch_t *ls = b->ls[slot];
        int32_t maxn = 0;
        for (int32_t t = 0; t < ls->n_tree; t++) maxn = max(maxn, ls->node_base[t + 1] - ls->node_base[t]);
        f.mark_rows = min(maxn, max(ls->last_nnxt, 1));
        b->rows[slot] = min(maxn, max(min(ls->hist_bound, ls->row_bound), 1));  /* bound on the coming frame's list lengths */
    }
    b->ls[slot]->cur ^= 1;              /* lextree_active_swap */
    b->has_trans[slot] = 1;
    return S3A_OK;
}

static int32_t
submit(s3a_batch_t *b, int32_t slot, const float *feat, int32_t frame, int32_t frm, int32_t hmmbeam,
       int32_t pbeam, int32_t wbeam, int32_t phone_uses_wbeam, int32_t maxhmmpf, s3a_frame_result_t *res,
       int32_t *n_exit, int32_t *exit_wid, int32_t *exit_score, int32_t *exit_hist, int32_t max_exits)
{
    if (!b || slot < 0 || slot >= b->n_slots || !feat || !res || !n_exit || !exit_wid || !exit_score || !exit_hist)
        return S3A_EINVAL;
    s3a_lexsearch_t *ls = b->ls[slot];
    s3a_scorer_t *sc = b->sc[slot];
    BFrame &f = b->stage[slot];
    if (!b->active[slot] || !b->has_trans[slot]) {
        s3a_set_error("s3a_batch_step: slot %d needs utt_begin and a transition before every step", slot);
        return S3A_EINVAL;
    }
    f.slot = slot; f.cur = ls->cur; f.frm = frm;
    f.bm.hmmbeam = hmmbeam; f.bm.pbeam = pbeam; f.bm.wbeam = wbeam; f.bm.phone_uses_wbeam = phone_uses_wbeam;
    f.bm.maxhmmpf = maxhmmpf;
    f.may_hist = ls->hist_bound > maxhmmpf + (maxhmmpf >> 1);
    f.scan_epoch = ++ls->scan_epoch;
    f.scan_nc = scan_workgroups(b->rows[slot], b->opt_scan_chained);
    if (f.may_hist && -hmmbeam / NBIN == 0) { s3a_set_error("s3a_batch_step: -beam too narrow for histogram pruning"); return S3A_EUNSUP; }
    f.sc_frame = frame;
    f.sc_is_skip = (frame % sc->ds_ratio == 0) ? 0 : 1;
    f.sc_beam = f.sc_is_skip ? (int32_t)((float)sc->ci_pbeam * sc->tighten_factor) : sc->ci_pbeam;
    memset(f.feat, 0, sizeof f.feat);
    memcpy(f.feat, feat, sizeof(float) * sc->g->veclen);
    BOut &o = b->out[slot];
    o.res = res; o.n_exit = n_exit; o.wid = exit_wid; o.scr = exit_score; o.hist = exit_hist;
    o.max_exits = max_exits; o.frm = frm; o.may_hist = f.may_hist; o.rc = S3A_OK; o.err[0] = 0;
    b->has_trans[slot] = 0;
    b->order[b->n_arrived++] = slot;
    b->arrived[slot] = 1;
    return S3A_OK;
}

/* gmm_compute_lv1/lv2 + hmm_compute_lv2 + propagate_graph_ph_lv2 + the word-exit half of
 * propagate_graph_wd_lv2 for this decoder's frame -- executed together with the same frame step
 * of every other decoder inside an utterance.  Blocks until the step has run. */
extern "C" int32_t
s3a_batch_step(s3a_batch_t *b, int32_t slot, const float *feat, int32_t frame, int32_t frm, int32_t hmmbeam,
               int32_t pbeam, int32_t wbeam, int32_t phone_uses_wbeam, int32_t maxhmmpf,
               s3a_frame_result_t *res, int32_t *n_exit, int32_t *exit_wid, int32_t *exit_score,
               int32_t *exit_hist, int32_t max_exits)
{
    if (!b) return S3A_EINVAL;
    pthread_mutex_lock(&b->mu);
    int32_t rc = submit(b, slot, feat, frame, frm, hmmbeam, pbeam, wbeam, phone_uses_wbeam, maxhmmpf, res, n_exit,
                        exit_wid, exit_score, exit_hist, max_exits);
    if (rc == S3A_OK) {
        const unsigned long long my_gen = b->gen;
        if (b->n_arrived == b->n_active) {
            (void)run_batch(b);
            pthread_mutex_unlock(&b->mu);
        }
        else {
            /* wait OUTSIDE the lock, spinning: a step is a few hundred microseconds and there is a
             * core per decoder thread; a condition variable costs a wake-up plus a convoy on the
             * mutex for every waiter.  After ~2 ms of spinning (a decoder stuck in file I/O) sleep. */
            pthread_mutex_unlock(&b->mu);
            for (long spins = 0; __atomic_load_n(&b->gen, __ATOMIC_ACQUIRE) == my_gen; spins++) {
                if (spins < 200000) __builtin_ia32_pause();
                else { struct timespec ts = { 0, 50000 }; nanosleep(&ts, NULL); }
            }
        }
        return unpack_own(b, slot);
    }
    pthread_mutex_unlock(&b->mu);
    return rc;
}

/* single-threaded drivers (tests): submit every active decoder's frame, then run the step */
extern "C" int32_t
s3a_batch_submit(s3a_batch_t *b, int32_t slot, const float *feat, int32_t frame, int32_t frm, int32_t hmmbeam,
                 int32_t pbeam, int32_t wbeam, int32_t phone_uses_wbeam, int32_t maxhmmpf,
                 s3a_frame_result_t *res, int32_t *n_exit, int32_t *exit_wid, int32_t *exit_score,
                 int32_t *exit_hist, int32_t max_exits)
{
    if (!b) return S3A_EINVAL;
    pthread_mutex_lock(&b->mu);
    int32_t rc = submit(b, slot, feat, frame, frm, hmmbeam, pbeam, wbeam, phone_uses_wbeam, maxhmmpf, res, n_exit,
                        exit_wid, exit_score, exit_hist, max_exits);
    pthread_mutex_unlock(&b->mu);
    return rc;
}

extern "C" int32_t
s3a_batch_run(s3a_batch_t *b)
{
    if (!b) return S3A_EINVAL;
    pthread_mutex_lock(&b->mu);
    int32_t rc = S3A_OK;
    int32_t n = 0, order[BMAXSLOT];
    if (b->n_arrived > 0) {
        rc = run_batch(b);
        n = b->last_n;
        memcpy(order, b->last_order, sizeof(int32_t) * n);
    }
    pthread_mutex_unlock(&b->mu);
    for (int32_t z = 0; z < n; z++) {
        const int32_t r = unpack_own(b, order[z]);
        if (rc == S3A_OK) rc = r;
    }
    return rc;
}

/* steps run so far and decoder-frames they served (mean batch size = frames / steps) */
extern "C" int32_t
s3a_batch_stats(s3a_batch_t *b, int64_t *steps, int64_t *slot_frames)
{
    if (!b) return S3A_EINVAL;
    pthread_mutex_lock(&b->mu);
    if (steps) *steps = b->steps;
    if (slot_frames) *slot_frames = b->slot_frames;
    pthread_mutex_unlock(&b->mu);
    return S3A_OK;
}
