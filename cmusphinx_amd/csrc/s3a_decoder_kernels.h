/*
 * s3a_decoder_kernels.h -- the bodies of the fused-frame kernels as device functions, so that
 * the single-decoder kernels (s3a_decoder.hip: one decoder per launch, explicit arguments) and
 * the batched ones (s3a_batch.hip: B decoders per launch, arguments read from per-slot
 * descriptors) run the very same code.  BX / BY = the workgroup's coordinates within ONE
 * decoder's grid.  See s3a_decoder.hip for what each kernel replaces in the reference.
 */
#ifndef S3A_DECODER_KERNELS_H
#define S3A_DECODER_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <limits.h>
#include "s3a_device.h"
#include "s3a_structs.h"
#include "s3a_vit.h"
#include "s3a_scan.h"

#define WORST S3A_WORST
#define DBLOCK 256
/* a word that ATOMICS of an earlier phase of the SAME launch may have changed (they execute in L2; this CU's vector L1 can still
 * hold the line from an earlier plain load): read past the L1 (sc1).  Across a kernel boundary a plain load does; the bodies below
 * also run as phases of one persistent launch (ku_frames, s3a_utt.hip), so they read such words this way. */
#define S3A_ALD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define S3A_ALDU(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
/* k_dec_scan: workgroups per tree.  One (walking the list 1024 positions at a time) unless the host's bound on the
 * list length is long: a chained multi-workgroup scan costs ~3 us per link, which pays from ~16 k positions on
 * (56 k HMMs per frame: 44 -> 23 us; 3 k HMMs: 14 us either way, and slower when batched) */
#define SCAN_LONG_LIST 16384
static inline int32_t scan_workgroups(int32_t rows, int chained)
{
    /* (chained = S3A_SCAN_CHAINED when the search object was made: the chained scan whatever the length -- for the tests) */
    return (rows >= SCAN_LONG_LIST || chained) ? (rows + SCAN_THREADS - 1) / SCAN_THREADS : 1;
}
#define M3BLOCK 64          /* k_dec_enter3_mark: same reason (a composite leaf marks ~140 scattered senones) */
#define RSBLOCK 64          /* k_dec_resolve: the nodes something happens to are neighbours; small workgroups spread them over more CUs */
/* k_dec_hmm_eval's workgroup size EB (template): 64 while the lists are short -- a few thousand HMMs are a dozen
 * workgroups of 256, and a CU's memory pipeline serialises their ~50 scattered accesses per HMM; a wave per
 * workgroup puts them on four times as many CUs (20.6 -> 18.5 us) -- 256 for long lists (bound >= 32 k positions; 56 k HMMs: 34 vs 41 us) */
#define EVBLOCK_LONG_LIST 32768

struct FrameBeams {
    int32_t hmmbeam, pbeam, wbeam, phone_uses_wbeam, maxhmmpf;
};

#define NBIN 1000        /* srch_time_switch_tree.c:858 */
#define EMIT_WAVES (DBLOCK / 64)
#define EMIT_NARROW 8           /* children a lane walks by itself in k_dec_emit */
#define EMIT_BLOCKS 32          /* workgroups per tree in the list sweeps: 128 waves, 8192 positions per pass */

/* thresholds of srch_TST_hmm_compute_lv2 (srch_time_switch_tree.c:849-905) from the per-tree
 * maxima hmm_eval left in best[]; when the frame holds more than 1.5 x maxhmmpf HMMs the
 * beam found by the histogram kernels (hbin[NBIN]) replaces -beam and bounds the others */
__device__ __forceinline__ bool
frame_thresholds(const int32_t *best, const int32_t *nact, int32_t T, const FrameBeams &bm,
                 const int32_t *hbin, int32_t &bh, int32_t &bw, int32_t &n, int32_t &th,
                 int32_t &pth, int32_t &wth)
{
    bh = INT_MIN; bw = INT_MIN; n = 0;
    for (int32_t t = 0; t < T; t++) {
        bh = max(bh, best[2 * t]);
        bw = max(bw, best[2 * t + 1]);
        n += nact[t];
    }
    int32_t hb = bm.hmmbeam, pb = bm.pbeam, wb = bm.wbeam;
    const bool hist = n > bm.maxhmmpf + (bm.maxhmmpf >> 1);
    if (hist) {
        hb = hbin[NBIN];
        pb = max(hb, pb);
        wb = max(hb, wb);
    }
    th = add32(bh, hb);
    wth = add32(bw, wb);
    pth = bm.phone_uses_wbeam ? wth : add32(bh, pb);
    return hist;
}

/* ------------------------------------------------------------------ */
/* one HMM: node v of the active list, evaluated against the frame's senone scores (raw: the scorer's row; norm: the frame's
 * normaliser); returns the HMM's best score, w = its word id (< 0: not a word-final node) and out = its exit score */
/* (TAG: the record's frame tag is written along, as if the HMM survived the frame -- ku_frames: its propagation pass then only has to
 * touch the records of the HMMs it clears or enters, not of every survivor) */
template <int NE, bool TAG = false>
__device__ __forceinline__ int32_t
d_dec_hmm_eval_nd(int32_t v, const int4 nd, int32_t N,
                    const int32_t *__restrict__ tp_g, const int16_t *__restrict__ sseq,
                    const int16_t *__restrict__ comsseq, const int32_t *__restrict__ cs_off,
                    const int16_t *__restrict__ cs_list, const int32_t *__restrict__ cs_wt,
                    const int32_t *__restrict__ raw, int32_t norm,
                    int32_t *sc, int32_t *hist, int32_t *outs, int32_t *outh, int32_t *bests, int32_t cf,
                    const int32_t *__restrict__ psof_off, const int32_t *__restrict__ psof, int32_t *pstamp,
                    const int32_t *__restrict__ cs_val, int32_t &w, int32_t &out)
{
    /* (nd: the node's static words -- senone-sequence id, transition matrix, word id, composite? -- loaded by the caller, who may
     * have asked for them a turn ahead: ku_frames) */
    const int32_t ss = nd.x;
    HmmRegsT<int32_t> r;
    int32_t e[NE];
    /* the HMM's own state first: these loads do not depend on the senone scores and stay in flight
     * while the (longer) senone chain below runs */
#pragma unroll
    for (int st = 0; st < NE; st++) { r.s[st] = sc[NSI(st, N, v)]; r.h[st] = hist[NSI(st, N, v)]; }
    r.out = outs[NSV(v)];
    r.outh = outh[NSV(v)];
    int32_t tp[NS_TPW(NE)];
    {
        const int4 *tq = (const int4 *)(tp_g + nd.y * NS_TPW(NE));       /* 48- / 128-byte rows of a 16-byte aligned array */
#pragma unroll
        for (int q = 0; q < NS_TPW(NE) / 4; q++) { const int4 a = tq[q]; tp[4 * q] = a.x; tp[4 * q + 1] = a.y; tp[4 * q + 2] = a.z; tp[4 * q + 3] = a.w; }
    }
    const int32_t q_lo = psof_off ? psof_off[v] : 0, q_hi = psof_off ? psof_off[v + 1] : 0;
    w = nd.z;
    if (nd.w && cs_val) {                /* (the maxima were worked out once per composite senone: d_comsen_max) */
#pragma unroll
        for (int st = 0; st < NE; st++) {
            const int32_t cs = comsseq[ss * NE + st];
            e[st] = add32(add32(cs_val[cs], -norm), cs_wt[cs]);
        }
    }
    else if (nd.w) {
        /* composite senone = max over its member senones (dict2pid.c:1029-1048), up to one member
         * per context (~46): the three states' lists are walked together, 8 members each per round,
         * ids first and then scores, so a round is two round trips instead of 48 */
        int32_t lo[NE], hi[NE], m[NE], wt[NE];
#pragma unroll
        for (int st = 0; st < NE; st++) {
            const int32_t cs = comsseq[ss * NE + st];
            lo[st] = cs_off[cs]; hi[st] = cs_off[cs + 1]; wt[st] = cs_wt[cs]; m[st] = INT_MIN;
        }
        for (;;) {
            bool more = false;
#pragma unroll
            for (int st = 0; st < NE; st++) more = more || lo[st] < hi[st];
            if (!more) break;
            int32_t id[NE][8];
#pragma unroll
            for (int st = 0; st < NE; st++)
#pragma unroll
                for (int u = 0; u < 8; u++) id[st][u] = (lo[st] + u < hi[st]) ? (int32_t)cs_list[lo[st] + u] : -1;
#pragma unroll
            for (int st = 0; st < NE; st++) {
#pragma unroll
                for (int u = 0; u < 8; u++) if (id[st][u] >= 0) m[st] = max(m[st], raw[id[st][u]]);
                lo[st] += 8;
            }
        }
#pragma unroll
        for (int st = 0; st < NE; st++) e[st] = add32(add32(m[st], -norm), wt[st]);
    }
    else {
#pragma unroll
        for (int st = 0; st < NE; st++)
            e[st] = add32(raw[sseq[ss * NE + st]], -norm);
    }
    int32_t k;
    if (NE == 5) { int32_t out_written = 0; k = vit5(r, tp, e, out_written); (void)out_written; }
    else k = vit3(r, tp, e[0], e[1], e[2]);
#pragma unroll
    for (int st = 0; st < NE; st++) { sc[NSI(st, N, v)] = r.s[st]; hist[NSI(st, N, v)] = r.h[st]; }
    outs[NSV(v)] = r.out;
    outh[NSV(v)] = r.outh;
    bests[NSV(v)] = k;
    if (TAG) sc[NSV(v) + NS_FRAME(NE)] = cf + 1;
    out = r.out;
    /* this node is active in frame cf: stamp the parent sets its children belong to (k_dec_resolve
     * skips every node whose parent set carries no stamp of this frame).  (psof_off == NULL: the caller stamps
     * after the thresholds are known, and only for the nodes that propagate: d_dec_stamp) */
    for (int32_t q = q_lo; q < q_hi; q++) pstamp[psof[q]] = cf;
    return k;
}

template <int NE, bool TAG = false>
__device__ __forceinline__ int32_t
d_dec_hmm_eval_node(int32_t v, int32_t N, const int32_t *__restrict__ ssid, const int32_t *__restrict__ tmatid,
                    const int32_t *__restrict__ wid, const uint8_t *__restrict__ comp,
                    const int32_t *__restrict__ tp_g, const int16_t *__restrict__ sseq,
                    const int16_t *__restrict__ comsseq, const int32_t *__restrict__ cs_off,
                    const int16_t *__restrict__ cs_list, const int32_t *__restrict__ cs_wt,
                    const int32_t *__restrict__ raw, int32_t norm,
                    int32_t *sc, int32_t *hist, int32_t *outs, int32_t *outh, int32_t *bests, int32_t cf,
                    const int32_t *__restrict__ psof_off, const int32_t *__restrict__ psof, int32_t *pstamp,
                    const int32_t *__restrict__ cs_val, const int4 *__restrict__ node4, int32_t &w, int32_t &out)
{
    /* the node's static words: one 16-byte load when the caller keeps them packed (node4), else four arrays */
    int4 nd;
    if (node4) nd = node4[v]; else { nd.x = ssid[v]; nd.y = tmatid[v]; nd.z = wid[v]; nd.w = comp[v]; }
    return d_dec_hmm_eval_nd<NE, TAG>(v, nd, N, tp_g, sseq, comsseq, cs_off, cs_list, cs_wt, raw, norm, sc, hist, outs, outh, bests, cf, psof_off, psof,
                                     pstamp, cs_val, w, out);
}

template <int EB, int NE = 3>
__device__ __forceinline__ void
d_dec_hmm_eval(const int32_t *__restrict__ node_base, const int32_t *__restrict__ act,
               const int32_t *__restrict__ nact, int32_t N, int32_t n_tmat,
               const int32_t *__restrict__ ssid, const int32_t *__restrict__ tmatid,
               const int32_t *__restrict__ wid, const uint8_t *__restrict__ comp,
               const int32_t *__restrict__ tp_g, const int16_t *__restrict__ sseq,
               const int16_t *__restrict__ comsseq, const int32_t *__restrict__ cs_off,
               const int16_t *__restrict__ cs_list, const int32_t *__restrict__ cs_wt,
               const int32_t *__restrict__ raw, const int32_t *__restrict__ misc,
               int32_t *sc, int32_t *hist, int32_t *outs, int32_t *outh, int32_t *bests,
               int32_t *best_out, int32_t cf, const int32_t *__restrict__ psof_off,
               const int32_t *__restrict__ psof, int32_t *pstamp,
               const int32_t *__restrict__ gpart, int32_t gpart_n, int32_t *poswid, int32_t *posout,
        const int32_t BX, const int32_t BY, const int32_t *__restrict__ cs_val = NULL, const int4 *__restrict__ node4 = NULL)
{
    __shared__ int32_t red[2][EB / 64];
    __shared__ int32_t s_gb[EB / 64];
    /* (the transition matrices are read from global memory -- 12 cached words per HMM, independent of the senone
     * chain: staging them in LDS put a copy loop and a barrier in front of every workgroup) */
    /* the batched scorer leaves the CD maximum as one value per workgroup (s3a_batch.hip) */
    int32_t gb = INT_MIN;
    for (int32_t i = threadIdx.x; i < gpart_n; i += EB) gb = max(gb, gpart[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gb = max(gb, __shfl_xor(gb, o, 64));
    if ((threadIdx.x & 63) == 0) s_gb[threadIdx.x >> 6] = gb;
    __syncthreads();
    const int32_t t = BY, i = BX * EB + threadIdx.x;
    int32_t norm = max(misc[0], misc[5]);               /* the frame's normaliser */
    for (int w = 0; w < EB / 64; w++) norm = max(norm, s_gb[w]);
    int32_t best = INT_MIN, wbest = INT_MIN;
    if (i < nact[t]) {
        const int32_t v = act[node_base[t] + i];
        int32_t w, out;
        const int32_t k = d_dec_hmm_eval_node<NE>(v, N, ssid, tmatid, wid, comp, tp_g, sseq, comsseq, cs_off, cs_list, cs_wt, raw, norm,
                                                  sc, hist, outs, outh, bests, cf, psof_off, psof, pstamp, cs_val, node4, w, out);
        best = k;
        if (w >= 0) wbest = k;
        /* by list position (coalesced): k_dec_scan finds the word exits without chasing the node ids again */
        poswid[node_base[t] + i] = w;
        posout[node_base[t] + i] = out;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        best = max(best, __shfl_xor(best, o, 64));
        wbest = max(wbest, __shfl_xor(wbest, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = best; red[1][threadIdx.x >> 6] = wbest; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 0; w < EB / 64; w++) { best = max(best, red[0][w]); wbest = max(wbest, red[1][w]); }
        if (best != INT_MIN) atomicMax(&best_out[2 * t], best);
        if (wbest != INT_MIN) atomicMax(&best_out[2 * t + 1], wbest);
    }
}

/* -pheurtype > 0 (lextree.c:1443-1486): a transition parent -> child must also satisfy
 * out(parent) + (prob(child) - prob(parent)) + phn_heur[ci(child)] >= hth(parent), hth = the running maximum of that
 * expression over the propagating HMMs up to the parent in active-list order (kbc->maxNewHeurScore, reset per tree)
 * + pl_beam; hth_pos is indexed by list position (tree slices), heur by CI phone.  All NULL: -pheurtype 0. */
/* listed parent sets with this many parents or more (up to 64) are resolved set-wise (d_dec_resolve_children) */
#ifndef SET_NP_MIN
#define SET_NP_MIN 2
#endif

struct HeurArgs {
    const uint8_t *node_ci;
    const int32_t *heur, *hth_pos;
};

/* the frame's thresholds when the histogram beam is known to the caller (hb_hist <= 0) or not in force (> 0) */
__device__ __forceinline__ void
frame_thresholds_hb(const int32_t *best, int32_t T, const FrameBeams &bm, int32_t hb_hist, int32_t &th, int32_t &pth)
{
    int32_t bh = INT_MIN, bw = INT_MIN;
    for (int32_t t = 0; t < T; t++) { bh = max(bh, best[2 * t]); bw = max(bw, best[2 * t + 1]); }
    int32_t hb = bm.hmmbeam, pb = bm.pbeam, wb = bm.wbeam;
    if (hb_hist <= 0) { hb = hb_hist; pb = max(hb, pb); wb = max(hb, wb); }
    th = add32(bh, hb);
    pth = bm.phone_uses_wbeam ? add32(bw, wb) : add32(bh, pb);
}

/*
 * Which nodes can be entered in this frame: an active HMM whose exit score reaches the phone threshold stamps the
 * parent sets its children belong to (a superset of the HMMs that propagate: lextree.c:1445-1452; one that is
 * cleared first -- phone threshold below the HMM threshold -- is sorted out by d_dec_resolve_node).  Run once the
 * thresholds are final: k_dec_resolve then walks the parent lists of those nodes only (a few hundred instead of
 * every child of every active HMM, each with up to ~46 parents).  A lane per list position i of tree t.
 */
/* the stamp of frame cf in a stamp array of type PS: the frame number itself, or (8-bit stamps) cf mod 255 -- 255 is
 * "never stamped" */
template <typename PS> __device__ __forceinline__ PS ps_val(int32_t cf) { return (PS)cf; }
template <> __device__ __forceinline__ uint8_t ps_val<uint8_t>(int32_t cf) { return (uint8_t)(cf % 255); }

template <typename PS>
__device__ __forceinline__ void
d_dec_stamp(const int32_t *__restrict__ act, const int32_t *__restrict__ outs, const int32_t *__restrict__ psof_off,
            const int32_t *__restrict__ psof, PS *pstamp, int32_t b, int32_t na, int32_t i, int32_t pth, int32_t cf)
{
    if (i >= na) return;
    const int32_t u = act[b + i];
    if (outs[NSV(u)] < pth) return;
    for (int32_t q = psof_off[u], q_hi = psof_off[u + 1]; q < q_hi; q++) pstamp[psof[q]] = ps_val<PS>(cf);
}

/* ------------------------------------------------------------------ */
/*
 * lextree_hmm_histbin (lextree.c:1314-1358) + the bin scan of srch_TST_hmm_compute_lv2
 * (srch_time_switch_tree.c:870-892).  Two side effects matter: the bins (they fix the
 * frame's beam) and the ORDER of every tree's active list afterwards -- the reference
 * rebuilds the list bin by bin, and within a bin in reverse insertion order (glist_add_ptr
 * prepends), which later decides ties in the propagation.  Kernel 1 bins every active HMM
 * (LDS-private histogram per workgroup); kernel 2, one workgroup per tree, finds the beam
 * and performs that stable reordering as a counting sort whose ranks are computed in list
 * order (wave-level peeling + per-wave bin counts).  Both return at once unless the frame
 * is over 1.5 x maxhmmpf (or `force`: the stand-alone entry point).
 */
/* (NT threads; s_bin: NBIN words of the workgroup's LDS) */
template <int NT>
__device__ __forceinline__ void
d_dec_hist_count_t(const int32_t *__restrict__ node_base, const int32_t *__restrict__ act,
                 const int32_t *__restrict__ nact, int32_t T, FrameBeams bm,
                 const int32_t *__restrict__ best, const int32_t *__restrict__ bests,
                 int32_t *binof, int32_t *hbin, int32_t force_tree, int32_t fbest, int32_t fbw,
                 int32_t nbin,
        const int32_t BX, const int32_t BY, int32_t *s_bin)
{
    __shared__ int32_t s_go, s_bh, s_bw;
    const int32_t t = BY;
    if (threadIdx.x == 0) {
        int32_t bh = INT_MIN, n = 0;
        for (int32_t k = 0; k < T; k++) { bh = max(bh, best[2 * k]); n += nact[k]; }
        if (force_tree >= 0) { s_go = (t == force_tree); s_bh = fbest; s_bw = fbw; }
        else { s_go = n > bm.maxhmmpf + (bm.maxhmmpf >> 1); s_bh = bh; s_bw = -bm.hmmbeam / NBIN; }
    }
    __syncthreads();
    if (!s_go) return;
    for (int32_t i = threadIdx.x; i < nbin; i += NT) s_bin[i] = 0;
    __syncthreads();
    const int32_t i = BX * NT + threadIdx.x, b = node_base[t];
    if (i < nact[t]) {
        int32_t k = (s_bh - bests[NSV(act[b + i])]) / s_bw;
        if (k >= nbin) k = nbin - 1;
        if (k < 0) k = 0;               /* cannot happen with bestscr = the frame's maximum */
        binof[b + i] = k;
        atomicAdd(&s_bin[k], 1);
    }
    __syncthreads();
    for (int32_t k = threadIdx.x; k < nbin; k += NT)
        if (s_bin[k]) atomicAdd(&hbin[k], s_bin[k]);
}

__device__ __forceinline__ void
d_dec_hist_count(const int32_t *__restrict__ node_base, const int32_t *__restrict__ act,
                 const int32_t *__restrict__ nact, int32_t T, FrameBeams bm,
                 const int32_t *__restrict__ best, const int32_t *__restrict__ bests,
                 int32_t *binof, int32_t *hbin, int32_t force_tree, int32_t fbest, int32_t fbw,
                 int32_t nbin,
        const int32_t BX, const int32_t BY)
{
    __shared__ int32_t s_bin[NBIN];
    d_dec_hist_count_t<DBLOCK>(node_base, act, nact, T, bm, best, bests, binof, hbin, force_tree, fbest, fbw, nbin, BX, BY, s_bin);
}

/* inclusive prefix sum over the SCAN_THREADS values of one workgroup */
__device__ __forceinline__ int32_t
block_inclusive_sum(int32_t x, int32_t *wsum /* [SCAN_THREADS / 64] shared */)
{
    const int32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t incl = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int32_t y = __shfl_up(incl, o, 64);
        if (lane >= o) incl += y;
    }
    __syncthreads();
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int32_t add = 0;
    for (int32_t w = 0; w < wave; w++) add += wsum[w];
    return incl + add;
}

/* the sort's LDS: the caller's (the persistent frame kernel lends an area other phases use too) */
template <int NT>
struct HistSortWs {
    int32_t s_cnt[NBIN], s_base[NBIN], s_run[NBIN];
    uint16_t s_cntw[NT / 64][NBIN];
    int32_t s_wsum[NT / 64];
    int32_t s_go, s_i, s_tot;
};

template <int NT>
__device__ __forceinline__ int32_t
d_dec_hist_sort_ws(const int32_t *__restrict__ node_base, int32_t *act, const int32_t *__restrict__ nact,
                int32_t T, FrameBeams bm, const int32_t *__restrict__ binof, int32_t *tmp,
                int32_t *hbin, int32_t *pos, int32_t force_tree, int32_t nbin,
        const int32_t BX, const int32_t BY, HistSortWs<NT> &ws)
{
    int32_t (&s_cnt)[NBIN] = ws.s_cnt, (&s_base)[NBIN] = ws.s_base, (&s_run)[NBIN] = ws.s_run;
    uint16_t (&s_cntw)[NT / 64][NBIN] = ws.s_cntw;
    int32_t (&s_wsum)[NT / 64] = ws.s_wsum;
    int32_t &s_go = ws.s_go, &s_i = ws.s_i, &s_tot = ws.s_tot;
    const int32_t t = BX, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        int32_t n = 0;
        for (int32_t k = 0; k < T; k++) n += nact[k];
        s_go = force_tree >= 0 ? (t == force_tree) : (n > bm.maxhmmpf + (bm.maxhmmpf >> 1));
        s_i = nbin;
    }
    __syncthreads();
    if (!s_go) return 1;                        /* (no histogram beam in force: beams are <= 0) */
    if (force_tree < 0) {
        /* for (i = 0, j = 0; i < nbin && j < maxhmmpf; i++, j += bin[i]);  -- bin[0] is never
         * counted and the read of bin[nbin] after the last increment decides nothing */
        /* (the bins in chunks of NT with a running total: the workgroup may be smaller than the histogram) */
        int32_t carry = 0;
        for (int32_t k0 = 0; k0 < nbin; k0 += NT) {
            const int32_t k = k0 + tid;
            const int32_t x = (k >= 1 && k < nbin) ? S3A_ALD(&hbin[k]) : 0;    /* (the counting pass's atomics) */
            const int32_t J = carry + block_inclusive_sum(x, s_wsum);       /* j after i reached k */
            if (k < nbin && J >= bm.maxhmmpf) atomicMin(&s_i, bm.maxhmmpf <= 0 ? 0 : k);
            if (tid == NT - 1) s_tot = J;
            __syncthreads();
            carry = s_tot;
            __syncthreads();
        }
        if (t == 0 && tid == 0) hbin[NBIN] = -(s_i * (-bm.hmmbeam / NBIN));
    }
    /* this tree's own bin counts and bin bases */
    const int32_t b = node_base[t], na = nact[t];
    for (int32_t k = tid; k < nbin; k += NT) { s_cnt[k] = 0; s_run[k] = 0; }
    for (int32_t k = tid; k < (NT / 64) * NBIN; k += NT) (&s_cntw[0][0])[k] = 0;
    __syncthreads();
    for (int32_t i = tid; i < na; i += NT) atomicAdd(&s_cnt[binof[b + i]], 1);
    __syncthreads();
    {
        int32_t carry = 0;
        for (int32_t k0 = 0; k0 < nbin; k0 += NT) {
            const int32_t k = k0 + tid;
            const int32_t x = k < nbin ? s_cnt[k] : 0;
            const int32_t incl = carry + block_inclusive_sum(x, s_wsum);
            if (k < nbin) s_base[k] = incl - x;
            if (tid == NT - 1) s_tot = incl;
            __syncthreads();
            carry = s_tot;
            __syncthreads();
        }
    }
    __syncthreads();
    for (int32_t c0 = 0; c0 < na; c0 += NT) {
        const int32_t i = c0 + tid;
        const bool valid = i < na;
        const int32_t k = valid ? binof[b + i] : -1;
        int32_t rank = 0, tot = 0;
        bool leader = false;
        unsigned long long remaining = __ballot(valid);
        while (remaining) {                                     /* wave-uniform */
            const int src = __ffsll((long long)remaining) - 1;
            const int32_t kb = __shfl(k, src, 64);
            const unsigned long long m = __ballot(valid && k == kb);
            if (valid && k == kb) {
                rank = __popcll(m & ((1ull << lane) - 1ull));
                tot = __popcll(m);
                leader = (lane == src);
            }
            remaining &= ~m;
        }
        if (leader) s_cntw[wave][k] = (uint16_t)tot;
        __syncthreads();
        if (valid) {
            int32_t before = s_run[k] + rank;
            for (int32_t w = 0; w < wave; w++) before += s_cntw[w][k];
            tmp[b + s_base[k] + s_cnt[k] - 1 - before] = act[b + i];
        }
        __syncthreads();
        if (leader) { atomicAdd(&s_run[k], tot); s_cntw[wave][k] = 0; }
        __syncthreads();
    }
    __syncthreads();
    for (int32_t i = tid; i < na; i += NT) {
        const int32_t v = tmp[b + i];
        act[b + i] = v;
        pos[PPX(v)] = i;
    }
    __syncthreads();
    return force_tree < 0 ? -(s_i * (-bm.hmmbeam / NBIN)) : 1;      /* the histogram beam (hbin[NBIN]) */
}

template <int NT>
__device__ __forceinline__ int32_t
d_dec_hist_sort_t(const int32_t *__restrict__ node_base, int32_t *act, const int32_t *__restrict__ nact,
                int32_t T, FrameBeams bm, const int32_t *__restrict__ binof, int32_t *tmp,
                int32_t *hbin, int32_t *pos, int32_t force_tree, int32_t nbin,
        const int32_t BX, const int32_t BY)
{
    __shared__ HistSortWs<NT> ws;
    return d_dec_hist_sort_ws<NT>(node_base, act, nact, T, bm, binof, tmp, hbin, pos, force_tree, nbin, BX, BY, ws);
}

__device__ __forceinline__ int32_t
d_dec_hist_sort(const int32_t *__restrict__ node_base, int32_t *act, const int32_t *__restrict__ nact,
                int32_t T, FrameBeams bm, const int32_t *__restrict__ binof, int32_t *tmp,
                int32_t *hbin, int32_t *pos, int32_t force_tree, int32_t nbin,
        const int32_t BX, const int32_t BY)
{
    return d_dec_hist_sort_t<SCAN_THREADS>(node_base, act, nact, T, bm, binof, tmp, hbin, pos, force_tree, nbin, BX, BY);
}

/* ------------------------------------------------------------------ */
/*
 * Only when the phone threshold lies BELOW the HMM threshold (-ptranskip frames use the word threshold
 * for phone transitions, srch_time_switch_tree.c:975-1003; or -pbeam wider than -beam).  Then an
 * active node under the HMM beam but over the phone threshold ("weak") matters: the reference clears
 * it at its turn -- and a cleared HMM propagates nothing -- UNLESS a parent earlier in the list
 * re-entered it first, in which case it survives and does propagate; and that parent may be weak
 * itself.  The dependency runs along the tree depth, so it is settled here before k_dec_resolve:
 * one workgroup collects the weak nodes and iterates "entered early by a propagating parent" to
 * its fixed point (monotone; at most tree-depth rounds), stamping propf[v] = cf.
 */
template <int NT>
__device__ __forceinline__ void
d_dec_weak_t(int32_t N, int32_t T, int32_t cf, FrameBeams bm, const int32_t *__restrict__ best,
           const int32_t *__restrict__ nact, const int32_t *__restrict__ node_base,
           const int32_t *__restrict__ act, const int32_t *__restrict__ prob,
           const int32_t *__restrict__ par_off, const int32_t *__restrict__ par,
           const int32_t *__restrict__ pos, const int32_t *__restrict__ posf, const int32_t *__restrict__ sc,
           const int32_t *__restrict__ outs, const int32_t *__restrict__ bests, const int32_t *__restrict__ wid,
           const int32_t *__restrict__ hbin, int32_t *propf, int32_t *weaklist,
        const int32_t BX, const int32_t BY)
{
    __shared__ int32_t n_weak, changed;
    int32_t th, pth;
    {
        int32_t bh, bw, n, wth;
        (void)frame_thresholds(best, nact, T, bm, hbin, bh, bw, n, th, pth, wth);
    }
    if (pth >= th) return;                              /* the usual geometry: nothing to do (uniform) */
    if (threadIdx.x == 0) n_weak = 0;
    __syncthreads();
    for (int32_t t = 0; t < T; t++)
        for (int32_t i = threadIdx.x; i < nact[t]; i += NT) {
            const int32_t v = act[node_base[t] + i];
            if (wid[v] < 0 && bests[NSV(v)] < th && outs[NSV(v)] >= pth) weaklist[atomicAdd(&n_weak, 1)] = v;
        }
    __syncthreads();
    const int32_t nw = n_weak;
    for (int32_t round = 0; round < N; round++) {       /* (ends after at most tree-depth rounds) */
        __syncthreads();
        if (threadIdx.x == 0) changed = 0;
        __syncthreads();
        for (int32_t k = threadIdx.x; k < nw; k += NT) {
            const int32_t v = weaklist[k];
            if (((volatile int32_t *)propf)[v] == cf) continue;
            const int32_t in0 = sc[NSV(v)], j = pos[PPX(v)];
            bool early = false;
            for (int32_t q = par_off[v]; q < par_off[v + 1] && !early; q++) {
                const int32_t g = par[q];
                if (posf[PPX(g)] != cf || pos[PPX(g)] >= j) continue;
                const int32_t po = outs[NSV(g)];
                if (po < pth) continue;
                if (bests[NSV(g)] < th && ((volatile int32_t *)propf)[g] != cf) continue;
                const int32_t ns = add32(po, add32(prob[v], -prob[g]));
                early = ns >= th && ns > in0;
            }
            if (early) { ((volatile int32_t *)propf)[v] = cf; changed = 1; }
        }
        __syncthreads();
        if (!changed) break;
    }
}

__device__ __forceinline__ void
d_dec_weak(int32_t N, int32_t T, int32_t cf, FrameBeams bm, const int32_t *__restrict__ best,
           const int32_t *__restrict__ nact, const int32_t *__restrict__ node_base,
           const int32_t *__restrict__ act, const int32_t *__restrict__ prob,
           const int32_t *__restrict__ par_off, const int32_t *__restrict__ par,
           const int32_t *__restrict__ pos, const int32_t *__restrict__ posf, const int32_t *__restrict__ sc,
           const int32_t *__restrict__ outs, const int32_t *__restrict__ bests, const int32_t *__restrict__ wid,
           const int32_t *__restrict__ hbin, int32_t *propf, int32_t *weaklist,
        const int32_t BX, const int32_t BY)
{
    d_dec_weak_t<SCAN_THREADS>(N, T, cf, bm, best, nact, node_base, act, prob, par_off, par, pos, posf, sc, outs, bests, wid, hbin, propf, weaklist, BX, BY);
}

/* ------------------------------------------------------------------ */
/*
 * lextree_hmm_propagate_non_leaves from every node's point of view (see the header of
 * s3a_lextree.hip for the rule); one thread per node of every tree: inactive nodes
 * without a propagating parent fall through after two loads.  Also resets the root-entry
 * scratch (key / first) for this frame's transitions.
 */
/* what the parents found mean for node v (the second half of the rule): the best entry by a parent earlier in the list (mE from
 * list position pE with history hE; firstE = the first earlier parent that improves on v's own entry score), the same for the
 * later parents, v's own survival at its turn j, the clear, and the turn at which v joins the next list */
__device__ __forceinline__ void
d_dec_resolve_finish(int32_t N, int32_t cf, int32_t th, int32_t b, int32_t v, bool is_active, int32_t j, int32_t in0,
                     int32_t mE, int32_t hE, int32_t firstE, int32_t mL, int32_t hL, int32_t firstL,
                     int32_t *sc, int32_t *hist, int32_t *outs, int32_t *outh, int32_t *bests,
                     int32_t *frame, int32_t *turn, int32_t *selfemit, int32_t *cnt, int32_t *posout)
{
    const int32_t nf = cf + 1;
    if (!is_active && mE == INT_MIN)
        return;                                         /* nothing happens to this node */
    int32_t cur = in0, h0 = hist[NSV(v)], my_turn = -1;
    bool in_list = false, cleared = false, entered = false;
    if (mE > in0) {
        cur = mE; h0 = hE; entered = true; in_list = true; my_turn = firstE;
    }
    else if (is_active) {
        if (bests[NSV(v)] >= th) { in_list = true; selfemit[b + j] = 1; atomicAdd(&cnt[b + j], 1); }
        else { cleared = true; cur = WORST; h0 = -1; }
    }
    if (mL > cur) {
        cur = mL; h0 = hL; entered = true;
        if (!in_list) { in_list = true; my_turn = firstL; }
    }
    if (cleared) {
        const int32_t ne = (int32_t)(hist - sc);
        if (ne == 3) {
            sc[NSI(1, N, v)] = WORST; sc[NSI(2, N, v)] = WORST;
            hist[NSI(1, N, v)] = -1; hist[NSI(2, N, v)] = -1;
        }
        else for (int32_t st = 1; st < ne; st++) { sc[NSI(st, N, v)] = WORST; hist[NSI(st, N, v)] = -1; }
        outs[NSV(v)] = WORST; outh[NSV(v)] = -1; bests[NSV(v)] = WORST;
        posout[b + j] = WORST;                  /* k_dec_scan reads the exit scores by list position */
    }
    if (cleared || entered) { sc[NSV(v)] = cur; hist[NSV(v)] = h0; }
    frame[NSV(v)] = in_list ? nf : (cleared ? -1 : frame[NSV(v)]);
    if (my_turn >= 0) { turn[v] = my_turn; atomicAdd(&cnt[b + my_turn], 1); }
}

/* what happens to node v (active, or with an active parent) in this frame: the rule of s3a_lextree.hip.
 * (PS = the type of the parent sets' stamps: int32 frame numbers, or their low 8 bits in the whole-utterance engine --
 * a stale stamp that happens to match only costs a parent walk that finds nothing) */
template <typename PS, bool HEUR = false>
__device__ __forceinline__ void
d_dec_resolve_node(int32_t N, int32_t T, int32_t cf, FrameBeams bm, const int32_t *__restrict__ best,
              const int32_t *__restrict__ nact, const int32_t *__restrict__ node_base,
              const int32_t *__restrict__ tree_of, const int32_t *__restrict__ prob,
              const int32_t *__restrict__ par_off, const int32_t *__restrict__ par,
              const int32_t *__restrict__ pos, const int32_t *__restrict__ posf,
              int32_t *sc, int32_t *hist, int32_t *outs, int32_t *outh, int32_t *bests,
              int32_t *frame, int32_t *turn, int32_t *selfemit, int32_t *cnt,
              unsigned long long *key, int32_t *first, int32_t *hbin,
              const int32_t *__restrict__ ps, const PS *__restrict__ pstamp,
              const int32_t *__restrict__ rootnodes, int32_t n_rootnodes,
              const int32_t *__restrict__ propf, int32_t *posout,
        const int32_t v, const bool is_active, const bool has_par, const int32_t j_known = -1, const int32_t b_known = -1,
        const HeurArgs hx = HeurArgs{ NULL, NULL, NULL }, const int32_t *thp = NULL)
{
    /* (thp: the frame's {HMM threshold, phone threshold} when the caller has worked them out once -- ku_frames) */
    const int32_t nf = cf + 1;
    int32_t th, pth;
    if (thp) { th = thp[0]; pth = thp[1]; }
    else {
        int32_t bh, bw, n, wth;
        (void)frame_thresholds(best, nact, T, bm, hbin, bh, bw, n, th, pth, wth);
    }
    if (is_active && !has_par) {
        /* no parent can enter v (the usual active HMM): it survives or is cleared at its own turn -- one gather */
        const int32_t j = j_known >= 0 ? j_known : pos[PPX(v)], b = b_known >= 0 ? b_known : node_base[tree_of[v]];
        if (bests[NSV(v)] >= th) { selfemit[b + j] = 1; atomicAdd(&cnt[b + j], 1); frame[NSV(v)] = nf; }
        else {
            const int32_t ne = (int32_t)(hist - sc);        /* (the record's layout: s3a_structs.h) */
            if (ne == 3) {
                sc[NSV(v)] = WORST; sc[NSI(1, N, v)] = WORST; sc[NSI(2, N, v)] = WORST;
                hist[NSV(v)] = -1; hist[NSI(1, N, v)] = -1; hist[NSI(2, N, v)] = -1;
            }
            else for (int32_t st = 0; st < ne; st++) { sc[NSI(st, N, v)] = WORST; hist[NSI(st, N, v)] = -1; }
            outs[NSV(v)] = WORST; outh[NSV(v)] = -1; bests[NSV(v)] = WORST;
            posout[b + j] = WORST;
            frame[NSV(v)] = -1;
        }
        return;
    }
    const int32_t j = is_active ? (j_known >= 0 ? j_known : pos[PPX(v)]) : INT_MAX;     /* (known when v was taken from the list) */
    const int32_t in0 = sc[NSV(v)];
    int32_t mE = INT_MIN, pE = INT_MAX, hE = -1, firstE = INT_MAX;
    int32_t mL = INT_MIN, pL = INT_MAX, hL = -1, firstL = INT_MAX;
    /* parents in batches of 8: the ids, then their list stamps, are independent loads (a first-level
     * node has one parent per left-context variant of its root, ~46: a one-at-a-time walk is 46
     * dependent round trips); the comparisons below do not depend on the visiting order */
    for (int32_t k0 = has_par ? par_off[v] : 0, kend = has_par ? par_off[v + 1] : 0; k0 < kend; k0 += 8) {
        int32_t pid[8], pf[8];
#pragma unroll
        for (int u = 0; u < 8; u++) pid[u] = (k0 + u < kend) ? par[k0 + u] : -1;
#pragma unroll
        for (int u = 0; u < 8; u++) pf[u] = (pid[u] >= 0) ? posf[PPX(pid[u])] : INT_MIN;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (pf[u] != cf || pid[u] < 0) continue;
            const int32_t p = pid[u];
            const int32_t po = outs[NSV(p)];
            if (po < pth) continue;
            /* phone threshold BELOW the HMM threshold (-ptranskip frames, -pbeam wider than -beam): a
             * parent under the HMM beam is cleared at its turn (hmm_clear resets its exit score) before
             * it could propagate -- unless one of ITS parents re-entered it earlier in this frame
             * (k_dec_weak worked that out and stamped it) */
            if (pth < th && bests[NSV(p)] < th && propf[p] != cf) continue;
            const int32_t ns = add32(po, add32(prob[v], -prob[p]));
            if (ns < th) continue;
            const int32_t pp = pos[PPX(p)];
            if (HEUR && add32(ns, hx.heur[hx.node_ci[v]]) < hx.hth_pos[(b_known >= 0 ? b_known : node_base[tree_of[v]]) + pp]) continue;
            if (pp < j) {
                if (ns > mE || (ns == mE && pp < pE)) { mE = ns; pE = pp; hE = outh[NSV(p)]; }
                if (ns > in0 && pp < firstE) firstE = pp;
            }
            else {
                if (ns > mL || (ns == mL && pp < pL)) { mL = ns; pL = pp; hL = outh[NSV(p)]; }
                if (pp < firstL) firstL = pp;
            }
        }
    }
    d_dec_resolve_finish(N, cf, th, b_known >= 0 ? b_known : node_base[tree_of[v]], v, is_active, j, in0, mE, hE, firstE, mL, hL, firstL,
                         sc, hist, outs, outh, bests, frame, turn, selfemit, cnt, posout);
}

template <typename PS, bool HEUR = false>
__device__ __forceinline__ void
d_dec_resolve(int32_t N, int32_t T, int32_t cf, FrameBeams bm, const int32_t *__restrict__ best,
              const int32_t *__restrict__ nact, const int32_t *__restrict__ node_base,
              const int32_t *__restrict__ tree_of, const int32_t *__restrict__ prob,
              const int32_t *__restrict__ par_off, const int32_t *__restrict__ par,
              const int32_t *__restrict__ pos, const int32_t *__restrict__ posf,
              int32_t *sc, int32_t *hist, int32_t *outs, int32_t *outh, int32_t *bests,
              int32_t *frame, int32_t *turn, int32_t *selfemit, int32_t *cnt,
              unsigned long long *key, int32_t *first, int32_t *hbin,
              const int32_t *__restrict__ ps, const PS *__restrict__ pstamp,
              const int32_t *__restrict__ rootnodes, int32_t n_rootnodes,
              const int32_t *__restrict__ propf, int32_t *posout,
        const int32_t BX, const int32_t BY, const HeurArgs hx = HeurArgs{ NULL, NULL, NULL })
{
    /* (the thresholds come from uniform addresses: computed per thread with scalar loads, and only by
     * the few workgroups that have anything to do -- no LDS, no barrier in front of the early exit) */
    if (BX == 0) {                              /* the bins were consumed by k_dec_hist_sort */
        int32_t bh, bw, n, th0, pth0, wth0;
        if (frame_thresholds(best, nact, T, bm, hbin, bh, bw, n, th0, pth0, wth0))
            for (int32_t i = threadIdx.x; i < NBIN; i += RSBLOCK) hbin[i] = 0;
    }
    const int32_t v = BX * RSBLOCK + threadIdx.x;
    if (v >= N) return;
    if (v < n_rootnodes) {                              /* lextree_enter only ever touches root nodes */
        const int32_t r = rootnodes[v];
        key[r] = 0ull;
        first[r] = INT_MAX;
    }
    const bool is_active = posf[PPX(v)] == cf;
    const int32_t q = ps[v];
    const bool has_par = q >= 0 && pstamp[q] == ps_val<PS>(cf); /* some parent may enter v (its parent set is stamped) */
    if (!is_active && !has_par) return;                 /* nothing can happen to v */
    d_dec_resolve_node<PS, HEUR>(N, T, cf, bm, best, nact, node_base, tree_of, prob, par_off, par, pos, posf, sc, hist, outs, outh, bests, frame, turn, selfemit, cnt, key, first, hbin, ps, pstamp, rootnodes, n_rootnodes, propf, posout, v, is_active, has_par, -1, -1, hx);
}

/*
 * k_dec_resolve for many lanes per launch (the whole-utterance engine).  With dozens of lanes the one-node-per-thread
 * sweep is hundreds of thousands of waves that live for the chain list stamp -> parent set -> its stamp and find
 * nothing to do: the launch lasts [waves / resident waves] x that chain.  Here
 *   - workgroups [0, GA) take the ACTIVE HMMs by list position, 64 per work item (dense waves: every lane has work);
 *   - workgroups [GA, GA + GB) sweep all nodes for the NOT active ones a propagating parent may enter, K nodes per
 *     thread, N/K apart (static parent-set id -> the set's stamp: two trips to the early exit, K times fewer waves);
 *     the wave's candidates -- siblings, so they come in runs -- are compacted through LDS and handled 64 at a time.
 */
template <int K, typename PS, bool HEUR = false>
__device__ __forceinline__ void
d_dec_resolve_utt(int32_t N, int32_t T, int32_t cf, FrameBeams bm, const int32_t *__restrict__ best,
              const int32_t *__restrict__ nact, const int32_t *__restrict__ node_base,
              const int32_t *__restrict__ tree_of, const int32_t *__restrict__ prob,
              const int32_t *__restrict__ par_off, const int32_t *__restrict__ par,
              const int32_t *__restrict__ pos, const int32_t *__restrict__ posf,
              int32_t *sc, int32_t *hist, int32_t *outs, int32_t *outh, int32_t *bests,
              int32_t *frame, int32_t *turn, int32_t *selfemit, int32_t *cnt,
              unsigned long long *key, int32_t *first, int32_t *hbin,
              const int32_t *__restrict__ ps, const PS *__restrict__ pstamp,
              const int32_t *__restrict__ rootnodes, int32_t n_rootnodes,
              const int32_t *__restrict__ propf, int32_t *posout, const int32_t *__restrict__ act,
        const int32_t BX, const int32_t GA, const int32_t GB, const HeurArgs hx = HeurArgs{ NULL, NULL, NULL },
        const int32_t *__restrict__ claim = NULL)
{
#define RS_ARGS N, T, cf, bm, best, nact, node_base, tree_of, prob, par_off, par, pos, posf, sc, hist, outs, outh, bests, frame, turn,  \
        selfemit, cnt, key, first, hbin, ps, pstamp, rootnodes, n_rootnodes, propf, posout
    static_assert(RSBLOCK == 64, "d_dec_resolve_utt: one wave per workgroup");
    if (BX == 0) {                              /* the bins were consumed by k_dec_hist_sort */
        int32_t bh, bw, n, th0, pth0, wth0;
        if (frame_thresholds(best, nact, T, bm, hbin, bh, bw, n, th0, pth0, wth0))
            for (int32_t i = threadIdx.x; i < NBIN; i += RSBLOCK) hbin[i] = 0;
    }
    if (BX < GA) {
        for (int32_t v = BX * RSBLOCK + threadIdx.x; v < n_rootnodes; v += GA * RSBLOCK) {     /* lextree_enter only ever touches root nodes */
            const int32_t r = rootnodes[v];
            key[r] = 0ull;
            first[r] = INT_MAX;
        }
        int32_t w0 = 0;                         /* first work item of tree t */
        for (int32_t t = 0; t < T; t++) {
            const int32_t na = nact[t], nw = (na + RSBLOCK - 1) / RSBLOCK, b = node_base[t];
            /* this workgroup's items of tree t: w = BX, BX + GA, ... within [w0, w0 + nw) */
            int32_t w = BX >= w0 % GA ? w0 - w0 % GA + BX : w0 - w0 % GA + GA + BX;
            for (; w < w0 + nw; w += GA) {
                const int32_t i = (w - w0) * RSBLOCK + threadIdx.x;
                if (i < na) {
                    const int32_t v = act[b + i], q = ps[v];
                    const bool has_par = q >= 0 && pstamp[q] == ps_val<PS>(cf);
                    /* (claim: the listed sets' members with 2..64 parents, active or not, are d_dec_resolve_children's) */
                    if (has_par && claim && claim[q] == cf) {
                        const int32_t np = par_off[v + 1] - par_off[v];
                        if (np >= SET_NP_MIN && np <= 64) continue;
                    }
                    d_dec_resolve_node<PS, HEUR>(RS_ARGS, v, true, has_par, i, b, hx);
                }
            }
            w0 += nw;
        }
        return;
    }
    const int32_t stride = GB * RSBLOCK, v0 = (BX - GA) * RSBLOCK + threadIdx.x;
    int32_t q[K];
    PS st[K];
#pragma unroll
    for (int k = 0; k < K; k++) { const int32_t v = v0 + k * stride; q[k] = v < N ? ps[v] : -1; }
#pragma unroll
    for (int k = 0; k < K; k++) st[k] = q[k] >= 0 ? pstamp[q[k]] : (PS)(ps_val<PS>(cf) + 1);
    __shared__ int32_t s_cand[64 * K];
    int32_t total = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const bool c = st[k] == ps_val<PS>(cf);
        const unsigned long long m = __ballot(c);
        if (c) s_cand[total + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = v0 + k * stride;
        total += __popcll(m);
    }
    if (total == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int32_t i = threadIdx.x; i < total; i += 64) {
        const int32_t v = s_cand[i];
        if (posf[PPX(v)] != cf) d_dec_resolve_node<PS, HEUR>(RS_ARGS, v, false, true, -1, -1, hx);     /* (the active ones: by list position) */
    }
#undef RS_ARGS
}

/*
 * The nodes a propagating parent may ENTER while they are not on the list, found from the parents' side (round 4): the stamping
 * pass has listed the frame's stamped PARENT SETS (plist, any order, each once); a wave takes a set and its lanes take the set's
 * members (psmem: the nodes that share that parent list -- the children of one interior node, or the ~340 first-level nodes under
 * the ~46 left-context variants of one root).  A member that is on the list is the other half's (by list position).  Replaces
 * the sweep over ALL nodes for stamped parent sets (216 k nodes per lane and frame for a few hundred propagating HMMs); the
 * byte stamps stay for the active nodes' "can a parent enter me".  (Listing the propagating HMMs and walking their child
 * lists visited a first-level node once per variant, with an atomic claim each time: 375 k frames/s in the bench.)
 */
/* (WS: the wave's own 5 x 64 words of LDS when the workgroup holds several waves -- ku_frames --, else NULL: one wave per workgroup) */
template <typename PS, bool HEUR = false>
__device__ __forceinline__ void
d_dec_resolve_children(int32_t N, int32_t T, int32_t cf, FrameBeams bm, const int32_t *__restrict__ best,
              const int32_t *__restrict__ nact, const int32_t *__restrict__ node_base,
              const int32_t *__restrict__ tree_of, const int32_t *__restrict__ prob,
              const int32_t *__restrict__ par_off, const int32_t *__restrict__ par,
              const int32_t *__restrict__ pos, const int32_t *__restrict__ posf,
              int32_t *sc, int32_t *hist, int32_t *outs, int32_t *outh, int32_t *bests,
              int32_t *frame, int32_t *turn, int32_t *selfemit, int32_t *cnt,
              unsigned long long *key, int32_t *first, int32_t *hbin,
              const int32_t *__restrict__ ps, const PS *__restrict__ pstamp,
              const int32_t *__restrict__ rootnodes, int32_t n_rootnodes,
              const int32_t *__restrict__ propf, int32_t *posout,
              const int32_t *__restrict__ plist, int32_t n_plist, const int32_t *__restrict__ psmem_off,
              const int32_t *__restrict__ psmem, int32_t W, int32_t NW, const HeurArgs hx = HeurArgs{ NULL, NULL, NULL },
              int32_t *WS = NULL, const int32_t *thp = NULL)
{
    const int32_t lane = threadIdx.x & 63;
    /* the qualifying parents of a several-parent set, loaded once for all of its members */
    __shared__ int32_t s_own[5][64];
    int32_t *s_po = WS ? WS : s_own[0], *s_pp = s_po + 64, *s_ph = s_po + 128, *s_pr = s_po + 192, *s_ht = s_po + 256;
    /* (a wave's stores to its LDS area seen by its other lanes) */
#define RC_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();   \
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
    int32_t th, pth;
    if (thp) { th = thp[0]; pth = thp[1]; }
    else {
        int32_t bh, bw, n, wth;
        (void)frame_thresholds(best, nact, T, bm, hbin, bh, bw, n, th, pth, wth);
    }
    for (int32_t k = W; k < n_plist; k += NW) {
        const int32_t q = plist[k];
        const int32_t m_lo = psmem_off[q], m_hi = psmem_off[q + 1];
        const int32_t x0 = psmem[m_lo], kp0 = par_off[x0], np = par_off[x0 + 1] - kp0;
        if (np >= SET_NP_MIN && np <= 64) {
            /* a first-level set: ~340 members under the ~46 left-context variants of one root.  Every member walking the
             * same 46 parents was 46 x (id + list stamp) gathers per member; the few variants that do propagate are found
             * once, and a member only combines their exit scores with its own probability (the parent loop of
             * d_dec_resolve_node from LDS; the members that are ON the list included: d_dec_resolve_utt skips them) */
            bool qual = false;
            int32_t po = 0, pp = 0, ph = 0, pr = 0, ht = INT_MIN, g = -1;
            if (lane < np) {
                g = par[kp0 + lane];
                if (posf[PPX(g)] == cf) {
                    po = outs[NSV(g)];
                    qual = po >= pth && !(pth < th && bests[NSV(g)] < th && propf[g] != cf);
                }
            }
            const unsigned long long qm = __ballot(qual);
            const int32_t b = node_base[tree_of[x0]];
            if (qual) {
                pp = pos[PPX(g)]; ph = outh[NSV(g)]; pr = prob[g];
                if (HEUR) ht = hx.hth_pos[b + pp];
                const int32_t at = __popcll(qm & ((1ull << lane) - 1ull));
                s_po[at] = po; s_pp[at] = pp; s_ph[at] = ph; s_pr[at] = pr; s_ht[at] = ht;
            }
            RC_WAVE_SYNC();
            const int32_t nq = __popcll(qm);
            for (int32_t c = m_lo + lane; c < m_hi; c += 64) {
                const int32_t x = psmem[c];
                const bool on_list = posf[PPX(x)] == cf;                         /* (the list position pass leaves these members to us) */
                if (!on_list && nq == 0) continue;
                const int32_t j = on_list ? pos[PPX(x)] : INT_MAX;
                const int32_t in0 = sc[NSV(x)], px = prob[x];
                const int32_t hv = HEUR ? hx.heur[hx.node_ci[x]] : 0;
                int32_t mE = INT_MIN, pE = INT_MAX, hE = -1, firstE = INT_MAX;
                int32_t mL = INT_MIN, pL = INT_MAX, hL = -1, firstL = INT_MAX;
                for (int32_t u = 0; u < nq; u++) {
                    const int32_t ns = add32(s_po[u], add32(px, -s_pr[u]));
                    if (ns < th) continue;
                    if (HEUR && add32(ns, hv) < s_ht[u]) continue;
                    const int32_t up = s_pp[u];
                    if (up < j) {
                        if (ns > mE || (ns == mE && up < pE)) { mE = ns; pE = up; hE = s_ph[u]; }
                        if (ns > in0 && up < firstE) firstE = up;
                    }
                    else {
                        if (ns > mL || (ns == mL && up < pL)) { mL = ns; pL = up; hL = s_ph[u]; }
                        if (up < firstL) firstL = up;
                    }
                }
                d_dec_resolve_finish(N, cf, th, b, x, on_list, j, in0, mE, hE, firstE, mL, hL, firstL,
                                     sc, hist, outs, outh, bests, frame, turn, selfemit, cnt, posout);
            }
            RC_WAVE_SYNC();
            continue;
        }
        for (int32_t c = m_lo + lane; c < m_hi; c += 64) {
            const int32_t x = psmem[c];
            if (posf[PPX(x)] == cf) continue;                                    /* on the list: resolved by list position */
            d_dec_resolve_node<PS, HEUR>(N, T, cf, bm, best, nact, node_base, tree_of, prob, par_off, par, pos, posf, sc, hist, outs, outh,
                                         bests, frame, turn, selfemit, cnt, key, first, hbin, ps, pstamp, rootnodes, n_rootnodes, propf,
                                         posout, x, false, true, -1, -1, hx, thp);
        }
    }
#undef RC_WAVE_SYNC
}

/* ------------------------------------------------------------------ */
/*
 * The ordered emission of the next active list, in two kernels.
 *
 * k_dec_scan, one workgroup per tree: prefix-sums the per-turn counts (base[i] = where the
 * nodes emitted during the turn of active-list position i start in the next list), writes the
 * self-emitted nodes, and compacts the word exits in active-list order; the LAST workgroup to
 * finish (agent-scope release by every workgroup, acquire by the last) assembles the frame
 * record for the host and resets the per-frame accumulators.
 *
 * k_dec_emit, a fixed grid of waves sweeping the active list: the children a parent put on the
 * list during its turn follow in child-list order; a lane per list position finds the few
 * turns that attributed children, then the whole wave walks such a parent's child list 64
 * links at a time and ranks the attributed children with a ballot.  (A lextree root has
 * hundreds of children -- 341 in the 20 k-word task -- and a one-thread walk of such a list,
 * one dependent HBM access per link, cost 600 us per frame.)
 * record = [best,wbest] x T | nact x T | thr[8] | n_exit x T | err x T | misc[8] | n_next x T | exits
 */
/*
 * Several workgroups per tree: workgroup (t, j) owns list positions [j * 1024, (j + 1) * 1024) of tree t and needs
 * the totals of the chunks in front of it.  Single-pass chained scan: it publishes its own totals (st_agg, flag =
 * 2 * epoch), then walks back over its predecessors -- adding their totals until one has already published its
 * inclusive prefix (st_pre, flag = 2 * epoch + 1) -- and publishes its own prefix.  `epoch` grows with every launch
 * on this decoder, so the flags never need a reset.  Predecessors have smaller workgroup ids (dispatched first) and
 * the whole grid fits the chip, so the wait always ends; the spin is bounded all the same (error word 2 in the frame record).
 * With GC < NC workgroups per tree a workgroup takes every GC-th chunk in turn (the whole-utterance engine: the host
 * does not know the list lengths, and a workgroup per possible chunk is thousands of workgroups that only find out
 * that they have nothing to do).  A chunk still only waits for chunks with smaller numbers, which belong to workgroups
 * of the same tree that are dispatched together with it (GC is small), so the wait ends as before.
 */
template <int NT>
__device__ __forceinline__ void
d_dec_scan_t(int32_t N, int32_t T, int32_t cf, FrameBeams bm, const int32_t *__restrict__ node_base,
           const int32_t *__restrict__ act, const int32_t *__restrict__ nact,
           const int32_t *__restrict__ wid, const int32_t *__restrict__ prob,
           const int32_t *__restrict__ outs, const int32_t *__restrict__ outh,
           const int32_t *__restrict__ selfemit, int32_t *cnt, int32_t *base, int32_t *nxt, int32_t *nnxt,
           int32_t *pos, int32_t *posf, int32_t *best, int32_t *exits, int32_t *nexit,
           const int32_t *hbin, int32_t *misc, int32_t *done, int32_t *pack, int32_t max_exits,
           const int32_t *gpart, int32_t gpart_n, const int32_t *poswid, const int32_t *posout, int32_t reordered,
           unsigned long long *st_agg, unsigned long long *st_pre, int32_t *st_flag, int32_t st_stride,
           int32_t epoch, int32_t NC, int32_t GC,
        const int32_t BX, const int32_t BY)
{
    __shared__ int32_t s_wth, s_last;
    __shared__ int32_t s_gp[3];
    __shared__ int32_t s_thr[8];
    __shared__ unsigned long long s_wsum[NT / 64], s_chunk, s_prefix;
    __shared__ int32_t s_exit_open;
    /* GC <= NC workgroups per tree are launched; workgroup j owns chunks j, j + GC, ... (GC == NC: one each) */
    const int32_t t = BX / GC, j = BX - t * GC, b = node_base[t], na = nact[t];
    const int32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t c_step = (NC == 1 ? 1 : GC) * NT;
    const bool live = j * NT < na || j == 0;      /* (chunk 0 also reports an empty tree's totals) */
    /* a chunk's list entries: turn count; word id / exit score by list position (after a histogram reordering --
     * the evaluation wrote them before it -- through the node).  The first chunk's loads are issued before the
     * thresholds are worked out, the next chunk's before the scan of the current one. */
    int32_t u2 = 0, c2 = 0, w2 = -1, os2 = 0;
#define SCAN_FETCH(i_)                                                                              \
    do {                                                                                            \
        u2 = 0; c2 = 0; w2 = -1; os2 = 0;                                                           \
        if ((i_) < na) {                                                                            \
            u2 = act[b + (i_)]; c2 = S3A_ALD(&cnt[b + (i_)]);  /* (d_dec_resolve_finish's atomicAdd) */ \
            if (reordered) { w2 = wid[u2]; os2 = outs[NSV(u2)]; }                                        \
            else { w2 = poswid[b + (i_)]; os2 = posout[b + (i_)]; }                                 \
            cnt[b + (i_)] = 0;                                  /* the accumulator of the next frame */ \
        }                                                                                           \
    } while (0)
    SCAN_FETCH(j * NT + tid);
    /* (a workgroup past the end of its tree's list -- the grid is sized by the host's bound -- only reports in) */
    if (live) {
        if (tid == 0) {
            int32_t bh, bw, n, th, pth, wth;
            const bool hist = frame_thresholds(best, nact, T, bm, hbin, bh, bw, n, th, pth, wth);
            s_wth = wth;
            s_thr[0] = th; s_thr[1] = pth; s_thr[2] = wth; s_thr[3] = bh; s_thr[4] = bw; s_thr[5] = n;
            s_thr[6] = hist ? 1 : 0;
            s_thr[7] = 0;
            s_exit_open = 0;
        }
        __syncthreads();
        const int32_t wth = s_wth;
        unsigned long long carry = 0ull;        /* NC == 1: this workgroup walks all chunks, totals in a register */
        /* NC > 1: one chunk per workgroup (the host sizes NC by its bound on the list length), chained */
        for (int32_t c0 = j * NT; c0 == j * NT || c0 < na; c0 += c_step) {
            const int32_t i = c0 + tid, jc = c0 / NT;
            const int32_t u = u2, c = c2, w = w2, os = os2;
            SCAN_FETCH(i + c_step);
            /* both ordered compactions in one scan: the turn bases (exclusive sum of the turn counts; the
             * self-emitted nodes are written at theirs by k_dec_emit) and the word exits in list order (exclusive
             * sum of the exit flags) travel as the halves of one 64-bit value */
            const bool ex = i < na && w >= 0 && os >= wth;
            const unsigned long long x = (unsigned long long)(uint32_t)c | ((unsigned long long)(ex ? 1u : 0u) << 32);
            unsigned long long incl = x;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned long long y = __shfl_up(incl, o, 64);
                if (lane >= o) incl += y;
            }
            if (lane == 63) s_wsum[wave] = incl;
            __syncthreads();
            if (wave == 0) {
                const unsigned long long ws = (lane < NT / 64) ? s_wsum[lane] : 0ull;
                unsigned long long wi = ws;
#pragma unroll
                for (int o = 1; o < NT / 64; o <<= 1) {
                    const unsigned long long y = __shfl_up(wi, o, 64);
                    if (lane >= o) wi += y;
                }
                if (lane < NT / 64) s_wsum[lane] = wi - ws;   /* exclusive wave offsets */
                if (lane == NT / 64 - 1) s_chunk = wi;        /* the chunk's totals */
            }
            __syncthreads();
            unsigned long long prefix = carry;
            if (NC > 1) {
                if (tid == 0) {
                    const unsigned long long A = s_chunk;
                    unsigned long long pre = 0ull;
                    const int32_t me = t * st_stride + jc;   /* (st_stride >= the tree's chunks: the arrays' row length) */
                    if (jc > 0) {
                        st_agg[me] = A;
                        __hip_atomic_store(&st_flag[me], 2 * epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                        for (int32_t p = jc - 1; p >= 0; p--) {
                            int32_t f, spins = 0;
                            while ((f = __hip_atomic_load(&st_flag[t * st_stride + p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < 2 * epoch) {
                                if (++spins > (1 << 22)) { nexit[T + t] = 2; break; }   /* (cannot happen: see above) */
                                __builtin_amdgcn_s_sleep(1);
                            }
                            if (f == 2 * epoch + 1) { pre += ((volatile unsigned long long *)st_pre)[t * st_stride + p]; break; }
                            pre += ((volatile unsigned long long *)st_agg)[t * st_stride + p];
                        }
                    }
                    st_pre[me] = pre + A;
                    __hip_atomic_store(&st_flag[me], 2 * epoch + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    s_prefix = pre;
                }
                __syncthreads();
                prefix = s_prefix;
            }
            /* (a wave only re-reads its own s_wsum entry, and s_chunk is rewritten after the next chunk's first
             * barrier: no further barrier is needed before the next chunk) */
            const unsigned long long excl = prefix + s_wsum[wave] + incl - x;
            carry = prefix + s_chunk;
            if (i < na) {
                base[b + i] = (int32_t)(uint32_t)excl;
                if (ex) {
                    const int32_t e = b + (int32_t)(excl >> 32);
                    const int32_t oh = outh[NSV(u)];
                    exits[e] = w;
                    exits[N + e] = add32(os, -prob[u]);
                    exits[2 * N + e] = oh;
                    if (oh == -1) s_exit_open = 1;
                }
            }
            if (tid == 0 && na <= c0 + NT) {      /* the tree's last chunk: its totals */
                nnxt[t] = (int32_t)(uint32_t)carry;
                nexit[t] = (int32_t)(carry >> 32);
            }
        }
#undef SCAN_FETCH
        __syncthreads();
        if (tid == 0 && s_exit_open) nexit[T + t] = 1;
    }
    /* (whole-utterance engine, s3a_utt.hip: the word-level kernel that follows assembles the frame record itself
     * -- d_dec_pack_frame -- so no workgroup needs to know that it is the last one: no fences, no counter) */
    if (done == NULL) return;
    /* publish this workgroup's results, find out whether it is the last of the launch */
    __syncthreads();
    if (tid == 0) {
        __threadfence();                                        /* agent-scope release */
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = (atomicAdd(done, 1) == T * GC - 1) ? 1 : 0;
        if (s_last) __threadfence();                            /* agent-scope acquire */
    }
    __syncthreads();
    if (!s_last) return;

    /* the batched scorer's per-workgroup maxima / counters (s3a_batch.hip) */
    if (threadIdx.x == 0) {
        if (!live) {                            /* (a workgroup that had no chunk has not worked them out yet) */
            int32_t bh, bw, n, th, pth, wth;
            const bool hist = frame_thresholds(best, nact, T, bm, hbin, bh, bw, n, th, pth, wth);
            s_thr[0] = th; s_thr[1] = pth; s_thr[2] = wth; s_thr[3] = bh; s_thr[4] = bw; s_thr[5] = n;
            s_thr[6] = hist ? 1 : 0;
            s_thr[7] = 0;
        }
        s_gp[0] = INT_MIN; s_gp[1] = 0; s_gp[2] = 0;
    }
    __syncthreads();
    if (gpart_n > 0) {
        volatile const int32_t *vg = gpart;
        int32_t gb = INT_MIN, gs = 0, gg = 0;
        for (int32_t q = threadIdx.x; q < gpart_n; q += NT) {
            gb = max(gb, vg[q]); gs += vg[gpart_n + q]; gg += vg[2 * gpart_n + q];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            gb = max(gb, __shfl_xor(gb, o, 64)); gs += __shfl_xor(gs, o, 64); gg += __shfl_xor(gg, o, 64);
        }
        if ((threadIdx.x & 63) == 0 && gb != INT_MIN) { atomicMax(&s_gp[0], gb); atomicAdd(&s_gp[1], gs); atomicAdd(&s_gp[2], gg); }
    }
    __syncthreads();
    const int32_t hdr = 6 * T + 16;
    volatile const int32_t *vbest = best, *vnexit = nexit, *vex = exits, *vmisc = misc, *vnnxt = nnxt;
    for (int32_t q = threadIdx.x; q < 2 * T; q += NT) pack[q] = vbest[q];
    for (int32_t q = threadIdx.x; q < T; q += NT) {
        pack[2 * T + q] = nact[q];
        pack[3 * T + 8 + q] = vnexit[q];
        pack[4 * T + 8 + q] = vnexit[T + q];
        pack[5 * T + 16 + q] = vnnxt[q];
    }
    if (threadIdx.x < 8) {
        pack[3 * T + threadIdx.x] = s_thr[threadIdx.x];
        int32_t m = vmisc[threadIdx.x];
        if (threadIdx.x == 0) m = max(m, s_gp[0]);
        if (threadIdx.x == 1 || threadIdx.x == 2) m += s_gp[threadIdx.x];
        if (threadIdx.x == 6) m = max(max(vmisc[0], s_gp[0]), vmisc[5]);    /* srch->senscale */
        pack[5 * T + 8 + threadIdx.x] = m;
    }
    int32_t off = 0;
    for (int32_t tt = 0; tt < T; tt++) {
        const int32_t n = vnexit[tt], bb = node_base[tt];
        for (int32_t q = threadIdx.x; q < n; q += NT) {
            const int32_t k = off + q;
            if (k < max_exits) {
                pack[hdr + 3 * k] = vex[bb + q];
                pack[hdr + 3 * k + 1] = vex[N + bb + q];
                pack[hdr + 3 * k + 2] = vex[2 * N + bb + q];
            }
        }
        off += n;
    }
    __syncthreads();
    /* reset the per-frame accumulators for the next frame */
    for (int32_t q = threadIdx.x; q < 2 * T; q += NT) { best[q] = INT_MIN; nexit[q] = 0; }
    if (threadIdx.x < 8) misc[threadIdx.x] = (threadIdx.x == 0 || threadIdx.x == 5) ? INT_MIN : 0;
    if (threadIdx.x == 0) *done = 0;
}

__device__ __forceinline__ void
d_dec_scan(int32_t N, int32_t T, int32_t cf, FrameBeams bm, const int32_t *__restrict__ node_base,
           const int32_t *__restrict__ act, const int32_t *__restrict__ nact,
           const int32_t *__restrict__ wid, const int32_t *__restrict__ prob,
           const int32_t *__restrict__ outs, const int32_t *__restrict__ outh,
           const int32_t *__restrict__ selfemit, int32_t *cnt, int32_t *base, int32_t *nxt, int32_t *nnxt,
           int32_t *pos, int32_t *posf, int32_t *best, int32_t *exits, int32_t *nexit,
           const int32_t *hbin, int32_t *misc, int32_t *done, int32_t *pack, int32_t max_exits,
           const int32_t *gpart, int32_t gpart_n, const int32_t *poswid, const int32_t *posout, int32_t reordered,
           unsigned long long *st_agg, unsigned long long *st_pre, int32_t *st_flag, int32_t st_stride,
           int32_t epoch, int32_t NC, int32_t GC,
        const int32_t BX, const int32_t BY)
{
    d_dec_scan_t<SCAN_THREADS>(N, T, cf, bm, node_base, act, nact, wid, prob, outs, outh, selfemit, cnt, base, nxt, nnxt, pos, posf, best, exits, nexit, hbin, misc, done, pack, max_exits, gpart, gpart_n, poswid, posout, reordered, st_agg, st_pre, st_flag, st_stride, epoch, NC, GC, BX, BY);
}

/*
 * The frame record for the whole-utterance engine's word level, which runs in the SAME workgroup right after: every
 * input is loaded up front (one round trip: per-tree maxima, list lengths, exit counts, the scorer's per-workgroup
 * columns), the record is assembled in LDS (hdr_s: the header, 6 T + 16 words; ex_s: the exits when there are at most
 * ex_cap of them) and written to `pack` on the side (the wide-beam launches and the host read it there).  Returns the
 * number of exits.  blockDim.x threads; hdr_s / ex_s are the caller's shared arrays.
 */
__device__ __forceinline__ int32_t
d_dec_pack_frame_lds(int32_t N, int32_t T, FrameBeams bm, const int32_t *__restrict__ node_base,
                     const int32_t *__restrict__ nact, int32_t *best, const int32_t *exits, int32_t *nexit,
                     const int32_t *hbin, int32_t *misc, int32_t *pack, int32_t max_exits, const int32_t *gpart,
                     int32_t gpart_n, const int32_t *nnxt, int32_t *hdr_s, int32_t *ex_s, int32_t ex_cap)
{
    __shared__ int32_t sp_gp[3];
    const int32_t nt = blockDim.x, tid = threadIdx.x, hdr = 6 * T + 16;
    int32_t gb = INT_MIN, gs = 0, gg = 0;
    for (int32_t q = tid; q < gpart_n; q += nt) { gb = max(gb, gpart[q]); gs += gpart[gpart_n + q]; gg += gpart[2 * gpart_n + q]; }
    for (int32_t q = tid; q < hdr; q += nt) hdr_s[q] = 0;
    if (tid < 3) sp_gp[tid] = tid == 0 ? INT_MIN : 0;
    __syncthreads();
    if (tid < 2 * T) hdr_s[tid] = S3A_ALD(&best[tid]);               /* (d_dec_hmm_eval's atomicMax) */
    if (tid < T) {
        hdr_s[2 * T + tid] = nact[tid];
        hdr_s[3 * T + 8 + tid] = nexit[tid];
        hdr_s[4 * T + 8 + tid] = nexit[T + tid];
        hdr_s[5 * T + 16 + tid] = nnxt[tid];
    }
    if (tid >= 64 && tid < 72) hdr_s[5 * T + 8 + (tid - 64)] = misc[tid - 64];      /* raw: merged with the columns below */
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        gb = max(gb, __shfl_xor(gb, o, 64)); gs += __shfl_xor(gs, o, 64); gg += __shfl_xor(gg, o, 64);
    }
    if ((tid & 63) == 0 && gb != INT_MIN) { atomicMax(&sp_gp[0], gb); atomicAdd(&sp_gp[1], gs); atomicAdd(&sp_gp[2], gg); }
    __syncthreads();
    if (tid == 0) {
        int32_t bh, bw, n, th, pth, wth;
        const bool hist = frame_thresholds(hdr_s, hdr_s + 2 * T, T, bm, hbin, bh, bw, n, th, pth, wth);
        hdr_s[3 * T + 0] = th; hdr_s[3 * T + 1] = pth; hdr_s[3 * T + 2] = wth; hdr_s[3 * T + 3] = bh; hdr_s[3 * T + 4] = bw;
        hdr_s[3 * T + 5] = n; hdr_s[3 * T + 6] = hist ? 1 : 0; hdr_s[3 * T + 7] = 0;
        const int32_t m0 = hdr_s[5 * T + 8 + 0], m5 = hdr_s[5 * T + 8 + 5];
        hdr_s[5 * T + 8 + 0] = max(m0, sp_gp[0]);
        hdr_s[5 * T + 8 + 1] += sp_gp[1];
        hdr_s[5 * T + 8 + 2] += sp_gp[2];
        hdr_s[5 * T + 8 + 6] = max(max(m0, sp_gp[0]), m5);      /* srch->senscale */
    }
    int32_t off = 0, total = 0;
    for (int32_t tt = 0; tt < T; tt++) total += hdr_s[3 * T + 8 + tt];
    for (int32_t tt = 0; tt < T; tt++) {
        const int32_t n = hdr_s[3 * T + 8 + tt], bb = node_base[tt];
        for (int32_t q = tid; q < n; q += nt) {
            const int32_t k = off + q;
            const int32_t w = exits[bb + q], sc = exits[N + bb + q], h = exits[2 * N + bb + q];
            if (k < max_exits) { pack[hdr + 3 * k] = w; pack[hdr + 3 * k + 1] = sc; pack[hdr + 3 * k + 2] = h; }
            if (total <= ex_cap) { ex_s[3 * k] = w; ex_s[3 * k + 1] = sc; ex_s[3 * k + 2] = h; }
        }
        off += n;
    }
    __syncthreads();
    for (int32_t q = tid; q < hdr; q += nt) pack[q] = hdr_s[q];
    for (int32_t q = tid; q < 2 * T; q += nt) { best[q] = INT_MIN; nexit[q] = 0; }
    if (tid < 8) misc[tid] = (tid == 0 || tid == 5) ? INT_MIN : 0;
    __syncthreads();
    return total;
}

/* The frame record of d_dec_scan's last workgroup, assembled instead by ONE workgroup of a LATER kernel
 * (blockDim.x threads; everything the scan kernels wrote is visible across the kernel boundary), followed by
 * the same reset of the per-frame accumulators. */
__device__ __forceinline__ void
d_dec_pack_frame(int32_t N, int32_t T, FrameBeams bm, const int32_t *__restrict__ node_base,
                 const int32_t *__restrict__ nact, int32_t *best, const int32_t *exits, int32_t *nexit,
                 const int32_t *hbin, int32_t *misc, int32_t *pack, int32_t max_exits, const int32_t *gpart,
                 int32_t gpart_n, const int32_t *nnxt)
{
    __shared__ int32_t sp_gp[3], sp_thr[8];
    const int32_t nt = blockDim.x;
    if (threadIdx.x == 0) {
        int32_t bh, bw, n, th, pth, wth;
        const bool hist = frame_thresholds(best, nact, T, bm, hbin, bh, bw, n, th, pth, wth);
        sp_thr[0] = th; sp_thr[1] = pth; sp_thr[2] = wth; sp_thr[3] = bh; sp_thr[4] = bw; sp_thr[5] = n;
        sp_thr[6] = hist ? 1 : 0; sp_thr[7] = 0;
        sp_gp[0] = INT_MIN; sp_gp[1] = 0; sp_gp[2] = 0;
    }
    __syncthreads();
    if (gpart_n > 0) {
        int32_t gb = INT_MIN, gs = 0, gg = 0;
        for (int32_t q = threadIdx.x; q < gpart_n; q += nt) { gb = max(gb, gpart[q]); gs += gpart[gpart_n + q]; gg += gpart[2 * gpart_n + q]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            gb = max(gb, __shfl_xor(gb, o, 64)); gs += __shfl_xor(gs, o, 64); gg += __shfl_xor(gg, o, 64);
        }
        if ((threadIdx.x & 63) == 0 && gb != INT_MIN) { atomicMax(&sp_gp[0], gb); atomicAdd(&sp_gp[1], gs); atomicAdd(&sp_gp[2], gg); }
    }
    __syncthreads();
    const int32_t hdr = 6 * T + 16;
    for (int32_t q = threadIdx.x; q < 2 * T; q += nt) pack[q] = best[q];
    for (int32_t q = threadIdx.x; q < T; q += nt) {
        pack[2 * T + q] = nact[q];
        pack[3 * T + 8 + q] = nexit[q];
        pack[4 * T + 8 + q] = nexit[T + q];
        pack[5 * T + 16 + q] = nnxt[q];
    }
    if (threadIdx.x < 8) {
        pack[3 * T + threadIdx.x] = sp_thr[threadIdx.x];
        int32_t m = misc[threadIdx.x];
        if (threadIdx.x == 0) m = max(m, sp_gp[0]);
        if (threadIdx.x == 1 || threadIdx.x == 2) m += sp_gp[threadIdx.x];
        if (threadIdx.x == 6) m = max(max(misc[0], sp_gp[0]), misc[5]);     /* srch->senscale */
        pack[5 * T + 8 + threadIdx.x] = m;
    }
    int32_t off = 0;
    for (int32_t tt = 0; tt < T; tt++) {
        const int32_t n = nexit[tt], bb = node_base[tt];
        for (int32_t q = threadIdx.x; q < n; q += nt) {
            const int32_t k = off + q;
            if (k < max_exits) {
                pack[hdr + 3 * k] = exits[bb + q];
                pack[hdr + 3 * k + 1] = exits[N + bb + q];
                pack[hdr + 3 * k + 2] = exits[2 * N + bb + q];
            }
        }
        off += n;
    }
    __syncthreads();
    for (int32_t q = threadIdx.x; q < 2 * T; q += nt) { best[q] = INT_MIN; nexit[q] = 0; }
    if (threadIdx.x < 8) misc[threadIdx.x] = (threadIdx.x == 0 || threadIdx.x == 5) ? INT_MIN : 0;
    __syncthreads();
}

/* (wave_id of n_waves waves sweep tree t's list: the kernels map their workgroups onto that) */
__device__ __forceinline__ void
d_dec_emit_w(int32_t cf, const int32_t *__restrict__ node_base, const int32_t *__restrict__ act,
             const int32_t *__restrict__ nact, const int32_t *__restrict__ child_off,
             const int32_t *__restrict__ child, int32_t *turn, int32_t *selfemit,
             const int32_t *__restrict__ base, int32_t *nxt, const int32_t *nnxt, int32_t *pos, int32_t *posf,
        const int32_t t, const int32_t wave_id, const int32_t n_waves)
{
    const int32_t b = node_base[t], na = nact[t], nf = cf + 1, total = nnxt[t];
    const int32_t lane = threadIdx.x & 63;
    /* each wave sweeps 64 list positions at a time: a lane per position finds the (few) turns that
     * attributed children, then the whole wave walks those parents' child lists */
    for (int32_t i0 = wave_id * 64; i0 < na; i0 += n_waves * 64) {
        const int32_t i = i0 + lane;
        int32_t lo = 0, hi = 0;
        if (i < na) {
            lo = base[b + i];
            hi = (i + 1 < na) ? base[b + i + 1] : total;
            if (selfemit[b + i]) {
                /* the node put itself on the list at its own turn: position = its turn's base (the scattered
                 * stores happen here, spread over the sweep's workgroups, not in k_dec_scan's one per tree) */
                const int32_t u = act[b + i];
                nxt[b + lo] = u; PP_SET(pos, u, lo, nf);
                selfemit[b + i] = 0;
                lo++;
            }
        }
        /* a parent with a handful of children (most nodes) is finished by its own lane: the child ids, then their
         * turns, are independent loads; only the wide ones (a root has hundreds of children) take the whole wave */
        bool wide = false;
        if (lo < hi) {
            const int32_t u = act[b + i], c_lo = child_off[u], c_hi = child_off[u + 1];
            if (c_hi - c_lo <= EMIT_NARROW) {
                int32_t cid[EMIT_NARROW], ct[EMIT_NARROW];
#pragma unroll
                for (int q = 0; q < EMIT_NARROW; q++) cid[q] = (c_lo + q < c_hi) ? child[c_lo + q] : -1;
#pragma unroll
                for (int q = 0; q < EMIT_NARROW; q++) ct[q] = (cid[q] >= 0) ? turn[cid[q]] : -1;
                int32_t k = lo;
#pragma unroll
                for (int q = 0; q < EMIT_NARROW; q++)
                    if (cid[q] >= 0 && ct[q] == i && k < hi) {
                        nxt[b + k] = cid[q]; PP_SET(pos, cid[q], k, nf);
                        turn[cid[q]] = -1;
                        k++;
                    }
            }
            else wide = true;
        }
        unsigned long long todo = __ballot(wide);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            int32_t k = __shfl(lo, src, 64);
            const int32_t kend = __shfl(hi, src, 64), ip = i0 + src, u = act[b + ip];
            const int32_t c_lo = child_off[u], c_hi = child_off[u + 1];
            for (int32_t j0 = c_lo; j0 < c_hi && k < kend; j0 += 64) {
                const int32_t j = j0 + lane;
                const int32_t c = (j < c_hi) ? child[j] : -1;
                const bool mine = c >= 0 && turn[c] == ip;
                const unsigned long long m = __ballot(mine);
                if (mine) {
                    const int32_t q = k + __popcll(m & ((1ull << lane) - 1ull));
                    nxt[b + q] = c; PP_SET(pos, c, q, nf);
                    turn[c] = -1;
                }
                k += __popcll(m);
            }
        }
    }
}

__device__ __forceinline__ void
d_dec_emit(int32_t cf, const int32_t *__restrict__ node_base, const int32_t *__restrict__ act,
           const int32_t *__restrict__ nact, const int32_t *__restrict__ child_off,
           const int32_t *__restrict__ child, int32_t *turn, int32_t *selfemit,
           const int32_t *__restrict__ base, int32_t *nxt, const int32_t *nnxt, int32_t *pos, int32_t *posf,
        const int32_t BX, const int32_t BY)
{
    d_dec_emit_w(cf, node_base, act, nact, child_off, child, turn, selfemit, base, nxt, nnxt, pos, posf, BY,
                 BX * EMIT_WAVES + (threadIdx.x >> 6), EMIT_BLOCKS * EMIT_WAVES);
}

/* ------------------------------------------------------------------ */
/* lextree_enter for up to two trees of one frame + next frame's senone marks */
/* calls[c] = {inscore, inhist, offset of the call's root list in rootlist[], first entry index};
 * entry e of the frame = the (e - first)-th root of its call; groups: {tree, ent_lo, ent_hi}.
 * (The root lists of a 20 k-word lextree hold ~2000 nodes per left context: expanding ~90 k
 * entries on the host and copying them every frame cost more than the search itself.) */
struct Entries {
    const int32_t *calls, *rootlist;
    int32_t n_calls;
    const int32_t *rootprob = nullptr;      /* prob[] of the roots IN LIST ORDER (optional): the entry test of the many
                                             * entries that fail it is then two coalesced loads and no gather */
    /* entry e: its call and its place in the root lists */
    __device__ __forceinline__ void locate_idx(int32_t e, int32_t &idx, int32_t &c) const
    {
        int32_t lo = 0, hi = n_calls - 1;
        while (lo < hi) {
            const int32_t mid = (lo + hi + 1) >> 1;
            if (calls[4 * mid + 3] <= e) lo = mid; else hi = mid - 1;
        }
        c = lo;
        idx = calls[4 * c + 2] + (e - calls[4 * c + 3]);
    }
    /* the same for a whole wave at once (EVERY lane of the wave must call): the calls' first entries are loaded once,
     * 64 per lane-register, and each lane counts the calls that start at or before its entry -- one round trip instead
     * of the bisection's five dependent ones (the entry kernels are that chain times tens of thousands of waves) */
    __device__ __forceinline__ void locate_idx_wave(int32_t e, int32_t &idx, int32_t &c) const
    {
        if (n_calls > 128) { locate_idx(e, idx, c); return; }
        const int32_t lane = threadIdx.x & 63;
        const int32_t o0 = lane < n_calls ? calls[4 * lane + 3] : INT_MAX, r0 = lane < n_calls ? calls[4 * lane + 2] : 0;
        const int32_t o1 = lane + 64 < n_calls ? calls[4 * (lane + 64) + 3] : INT_MAX, r1 = lane + 64 < n_calls ? calls[4 * (lane + 64) + 2] : 0;
        int32_t cnt = 0;
        const int32_t n0 = min(n_calls, 64);
        for (int32_t k = 0; k < n0; k++) cnt += __shfl(o0, k, 64) <= e ? 1 : 0;
        for (int32_t k = 64; k < n_calls; k++) cnt += __shfl(o1, k - 64, 64) <= e ? 1 : 0;
        c = max(cnt - 1, 0);
        const int32_t oc = c < 64 ? __shfl(o0, c, 64) : __shfl(o1, c - 64, 64), rc = c < 64 ? __shfl(r0, c, 64) : __shfl(r1, c - 64, 64);
        idx = rc + (e - oc);
    }
    /* the look-ahead probability of the root at place idx (node v when it has been loaded already, else -1) */
    __device__ __forceinline__ int32_t root_prob(int32_t idx, const int32_t *__restrict__ prob) const
    {
        return rootprob ? rootprob[idx] : prob[rootlist[idx]];
    }
    __device__ __forceinline__ void locate(int32_t e, int32_t &v, int32_t &c) const
    {
        int32_t lo = 0, hi = n_calls - 1;
        while (lo < hi) {
            const int32_t mid = (lo + hi + 1) >> 1;
            if (calls[4 * mid + 3] <= e) lo = mid; else hi = mid - 1;
        }
        c = lo;
        v = rootlist[calls[4 * c + 2] + (e - calls[4 * c + 3])];
    }
};

__device__ __forceinline__ void
d_dec_enter1(Entries ent, int32_t n_ent, const int32_t *__restrict__ calls,
             const int32_t *__restrict__ prob, const int32_t *__restrict__ sc, int32_t thresh,
             unsigned long long *key, int32_t *first,
        const int32_t BX, const int32_t BY)
{
    const int32_t e = BX * blockDim.x + threadIdx.x;
    int32_t idx, c;
    ent.locate_idx_wave(e < n_ent ? e : 0, idx, c);
    if (e >= n_ent) return;
    const int32_t scr = add32(calls[4 * c], ent.root_prob(idx, prob));
    if (scr < thresh) return;
    const int32_t v = ent.rootlist[idx];
    if (!(sc[NSV(v)] < scr)) return;
    atomicMax(&key[v], ((unsigned long long)((uint32_t)scr ^ 0x80000000u) << 32) | (uint32_t)(0x7fffffff - c));
    atomicMin(&first[v], c);
}

/* The composite senones wanted in this frame (cs_need[cs] == stamp): their member senones are marked for scoring
 * (d_comsen_mark, before the scoring kernels) and, after scoring, their score -- the maximum over the members,
 * dict2pid.c:1029-1048 -- is left in cs_val[] for d_dec_hmm_eval (d_comsen_max): once per composite senone and frame
 * instead of once per HMM that carries it (a word-final HMM has three, of ~46 members each).  A wave looks at 64
 * composite senones with one load (few are wanted: a wave per senone would be thousands of waves that only find that
 * out), then its four 16-lane groups take the wanted ones in turn.  cs0 = the wave's first composite senone. */
template <bool MAXOP>
__device__ __forceinline__ void
d_comsen_wave(int32_t n_cs, const int32_t *__restrict__ cs_need, int32_t stamp, const int32_t *__restrict__ cs_off,
              const int16_t *__restrict__ cs_list, uint8_t *sen_active, const int32_t *__restrict__ raw, int32_t *cs_val,
              int32_t cs0)
{
    const int32_t lane = threadIdx.x & 63, sub = lane >> 4, l16 = lane & 15;
    const int32_t mine = cs0 + lane;
    const unsigned long long mask = __ballot(mine < n_cs && cs_need[mine] == stamp);
    const int32_t n = __popcll(mask);
    for (int32_t k0 = 0; k0 < n; k0 += 4) {            /* wave-uniform trip count: the shuffles below see all lanes */
        const int32_t k = k0 + sub;
        unsigned long long m = mask;
        for (int32_t i = 0; i < k && m; i++) m &= m - 1ull;
        const bool on = k < n;
        const int32_t cs = on ? cs0 + (__ffsll((long long)m) - 1) : 0;
        int32_t mx = INT_MIN;
        if (on)
            for (int32_t q = cs_off[cs] + l16, hi = cs_off[cs + 1]; q < hi; q += 16) {
                const int32_t id = cs_list[q];
                if (MAXOP) mx = max(mx, raw[id]); else sen_active[id] = 1;
            }
        if (MAXOP) {
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
            if (on && l16 == 0) cs_val[cs] = mx;
        }
    }
}

/* the same for a LIST of wanted composite senones (ku_frames): 16-lane group `grp` of `n_grp` takes every n_grp-th entry, U entries
 * per turn and two runs of 16 members of each at a time (the chains list entry -> member range -> member ids -> scores run side by
 * side: a turn is four round trips whatever it holds) */
template <bool MAXOP, int U = 4>
__device__ __forceinline__ void
d_comsen_list(const int32_t *__restrict__ wl, int32_t n_w, const int32_t *__restrict__ cs_off, const int16_t *__restrict__ cs_list,
              uint8_t *sen_active, const int32_t *__restrict__ raw, int32_t *cs_val, int32_t grp, int32_t n_grp,
              const int32_t *__restrict__ cs_wt = NULL, uint32_t *actbits = NULL)
{
    /* (actbits: the mask as bits (ku_frames: in LDS) instead of bytes.  cs_wt: the composite senone's weight is added to the maximum -- add32 wraps, so the evaluation's
     * (score - normaliser) + weight comes out the same whichever is added first) */
    const int32_t l16 = threadIdx.x & 15;
    for (int32_t j0 = 0; j0 < n_w; j0 += U * n_grp) {       /* (trip count uniform over the wave: the shuffles below see all lanes) */
        int32_t cs[U], lo[U], hi[U], mx[U];
        bool on[U];
#pragma unroll
        /* (every load unconditional, at a harmless index for a group without an entry: a load under a condition is a branch with its
         * own wait inside, and the U chains would run one after the other) */
        for (int u = 0; u < U; u++) { const int32_t j = j0 + u * n_grp + grp; on[u] = j < n_w; cs[u] = ((const __attribute__((address_space(1))) int32_t *)wl)[on[u] ? j : 0]; }
#pragma unroll
        for (int u = 0; u < U; u++) { const __attribute__((address_space(1))) int32_t *co = (const __attribute__((address_space(1))) int32_t *)cs_off; const int32_t a0 = co[cs[u]], a1 = co[cs[u] + 1]; lo[u] = on[u] ? a0 : 0; hi[u] = on[u] ? a1 : 0; mx[u] = INT_MIN; }
        for (int32_t q0 = 0; ; q0 += 32) {
            int32_t id[U][2];
            bool any = false;
#pragma unroll
            for (int u = 0; u < U; u++) {
#pragma unroll
                for (int h = 0; h < 2; h++) { const int32_t q = lo[u] + q0 + 16 * h + l16; const int32_t x = (int32_t)((const __attribute__((address_space(1))) int16_t *)cs_list)[q < hi[u] ? q : 0]; id[u][h] = q < hi[u] ? x : -1; }
                any = any || lo[u] + q0 < hi[u];
            }
            if (!any) break;
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int h = 0; h < 2; h++)
                {
                    if (MAXOP) { const int32_t x = ((const __attribute__((address_space(1))) int32_t *)raw)[max(id[u][h], 0)]; if (id[u][h] >= 0) mx[u] = max(mx[u], x); }
                    else if (id[u][h] >= 0) { if (actbits) (void)__hip_atomic_fetch_or((__attribute__((address_space(3))) uint32_t *)&actbits[id[u][h] >> 5], 1u << (id[u][h] & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); else sen_active[id[u][h]] = 1; }
                }
        }
        if (MAXOP) {
#pragma unroll
            for (int u = 0; u < U; u++) {
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) mx[u] = max(mx[u], __shfl_xor(mx[u], o, 64));
                if (on[u] && l16 == 0) ((__attribute__((address_space(1))) int32_t *)cs_val)[cs[u]] = cs_wt ? add32(mx[u], ((const __attribute__((address_space(1))) int32_t *)cs_wt)[cs[u]]) : mx[u];
            }
        }
    }
}

/* senones of one node (srch_TST_select_active_gmm's per-node step) */
__device__ __forceinline__ void
mark_node_senones(int32_t v, const int32_t *__restrict__ ssid, const uint8_t *__restrict__ comp,
                  const int16_t *__restrict__ sseq, const int16_t *__restrict__ comsseq,
                  const int32_t *__restrict__ cs_off, const int16_t *__restrict__ cs_list, uint8_t *sen_active,
                  int32_t *cs_need = NULL, int32_t stamp = 0, int32_t ne = 3, int32_t *cs_wl = NULL, int32_t *cs_wn = NULL)
{
    const int32_t ss = ssid[v];
    if (comp[v] && cs_need && cs_wl) {
        /* ... and LISTED once per frame by whoever stamps it first (ku_frames: the list is what the member marks and the maxima walk,
         * instead of a sweep over all composite senones for the few that are wanted) */
        for (int st = 0; st < ne; st++) {
            const int32_t cs = comsseq[ss * ne + st];
            if (atomicExch(&cs_need[cs], stamp) != stamp) cs_wl[atomicAdd(cs_wn, 1)] = cs;
        }
    }
    else if (comp[v] && cs_need) {
        /* the whole-utterance engine: a composite senone is WANTED (stamp); its members are marked once per frame by
         * d_comsen_mark however many HMMs share it */
        if (ne == 3) {
#pragma unroll
            for (int st = 0; st < 3; st++) cs_need[comsseq[ss * 3 + st]] = stamp;
        }
        else for (int st = 0; st < ne; st++) cs_need[comsseq[ss * ne + st]] = stamp;
    }
    else if (comp[v] && ne != 3) {
        for (int st = 0; st < ne; st++) {
            const int32_t cs = comsseq[ss * ne + st];
            for (int32_t q = cs_off[cs]; q < cs_off[cs + 1]; q++) sen_active[cs_list[q]] = 1;
        }
    }
    else if (comp[v]) {
        /* the three states' member lists together, 8 members each per round (as in d_dec_hmm_eval): a round is
         * one trip for 24 ids instead of three */
        int32_t lo[3], hi[3];
#pragma unroll
        for (int st = 0; st < 3; st++) {
            const int32_t cs = comsseq[ss * 3 + st];
            lo[st] = cs_off[cs]; hi[st] = cs_off[cs + 1];
        }
        while (lo[0] < hi[0] || lo[1] < hi[1] || lo[2] < hi[2]) {
            int32_t id[3][8];
#pragma unroll
            for (int st = 0; st < 3; st++)
#pragma unroll
                for (int u = 0; u < 8; u++) id[st][u] = (lo[st] + u < hi[st]) ? (int32_t)cs_list[lo[st] + u] : -1;
#pragma unroll
            for (int st = 0; st < 3; st++) {
#pragma unroll
                for (int u = 0; u < 8; u++) if (id[st][u] >= 0) sen_active[id[st][u]] = 1;
                lo[st] += 8;
            }
        }
    }
    else if (ne == 3) {
        for (int st = 0; st < 3; st++)
            sen_active[sseq[ss * 3 + st]] = 1;
    }
    else {
        for (int st = 0; st < ne; st++)
            sen_active[sseq[ss * ne + st]] = 1;
    }
}

/*
 * One workgroup per lextree_enter CALL (BX = call): which of its roots does this call put on the
 * next list (first qualifying call of a node that is not listed yet), ranked in root-list order
 * by a block scan: flag[e] = (rank within the call << 1) | listed, ctot[c] = the call's count.
 * Workgroup 0 also snapshots the list lengths before the entries (n0[t]).  (One workgroup
 * scanning all ~90 k entries of a 20 k-word frame took 30 us; the calls are independent.)
 */
template <int NT>
__device__ __forceinline__ void
d_dec_enter2_t(Entries ent, int32_t n_ent, const int32_t *__restrict__ calls,
             const int32_t *__restrict__ prob, const int32_t *__restrict__ sc,
             const int32_t *__restrict__ frame, const int32_t *__restrict__ first, int32_t thresh,
             int32_t nf, int32_t T, const int32_t *__restrict__ nnxt, int32_t *flag, int32_t *ctot,
             int32_t *n0,
        const int32_t BX, const int32_t BY)
{
    __shared__ int32_t s_ws[NT / 64], s_chunk;
    const int32_t c = BX;
    const int32_t lo = calls[4 * c + 3], hi = (c + 1 < ent.n_calls) ? calls[4 * (c + 1) + 3] : n_ent;
    const int32_t n = hi - lo, roots = calls[4 * c + 2], in = calls[4 * c];
    const int32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (c == 0)
        for (int32_t t = tid; t < T; t += NT) n0[t] = nnxt[t];
    /* one sweep: the entry's test (four gathers) once, its rank among the call's listed entries by a scan that
     * never leaves registers / LDS, one store of (rank << 1 | listed) */
    int32_t carry = 0;
    for (int32_t c0 = 0; c0 < n; c0 += NT) {
        const int32_t i = c0 + tid;
        int32_t q = 0;
        if (i < n) {
            const int32_t scr = add32(in, ent.root_prob(roots + i, prob));
            if (scr >= thresh) {
                const int32_t v = ent.rootlist[roots + i];
                q = (sc[NSV(v)] < scr && S3A_ALD(&first[v]) == c && frame[NSV(v)] != nf) ? 1 : 0;   /* (d_dec_enter1's atomicMin) */
            }
        }
        const unsigned long long m = __ballot(q);
        const int32_t below = __popcll(m & ((1ull << lane) - 1ull));   /* listed entries in front, this wave */
        if (lane == 0) s_ws[wave] = __popcll(m);
        __syncthreads();
        if (wave == 0) {
            const int32_t w = (lane < NT / 64) ? s_ws[lane] : 0;
            int32_t wi = w;
#pragma unroll
            for (int o = 1; o < NT / 64; o <<= 1) {
                const int32_t y = __shfl_up(wi, o, 64);
                if (lane >= o) wi += y;
            }
            if (lane < NT / 64) s_ws[lane] = wi - w;
            if (lane == NT / 64 - 1) s_chunk = wi;
        }
        __syncthreads();
        if (i < n) flag[lo + i] = ((carry + s_ws[wave] + below) << 1) | q;
        carry += s_chunk;
    }
    if (tid == 0) ctot[c] = carry;
}

__device__ __forceinline__ void
d_dec_enter2(Entries ent, int32_t n_ent, const int32_t *__restrict__ calls,
             const int32_t *__restrict__ prob, const int32_t *__restrict__ sc,
             const int32_t *__restrict__ frame, const int32_t *__restrict__ first, int32_t thresh,
             int32_t nf, int32_t T, const int32_t *__restrict__ nnxt, int32_t *flag, int32_t *ctot,
             int32_t *n0,
        const int32_t BX, const int32_t BY)
{
    d_dec_enter2_t<SCAN_THREADS>(ent, n_ent, calls, prob, sc, frame, first, thresh, nf, T, nnxt, flag, ctot, n0, BX, BY);
}

/* blocks [0, n_ent_blocks): write the listed roots to the next list (position = list length before
 * the entries + the counts of the group's earlier calls + rank within the call), mark their
 * senones, apply the winning entries; the remaining blocks mark the senones of the nodes that
 * were on the NEXT lists before the entries (srch_TST_select_active_gmm for the coming frame).
 * groups[g] = {tree, first entry, last entry + 1, first call}; n0 = list lengths before the
 * entries (== nnxt when there are none). */
__device__ __forceinline__ void
d_dec_enter3_mark(int32_t n_ent_blocks, Entries ent, int32_t n_ent,
                  const int32_t *__restrict__ calls, const int32_t *__restrict__ groups, int32_t n_groups,
                  int32_t nf, const unsigned long long *__restrict__ key, const int32_t *__restrict__ first,
                  const int32_t *__restrict__ flag, const int32_t *__restrict__ ctot,
                  const int32_t *__restrict__ n0,
                  int32_t *sc, int32_t *hist, int32_t *frame,
                  int32_t T, int32_t blocks_per_tree, const int32_t *__restrict__ node_base,
                  int32_t *nxt, int32_t *nnxt, int32_t *pos, int32_t *posf,
                  const int32_t *__restrict__ ssid, const uint8_t *__restrict__ comp,
                  const int16_t *__restrict__ sseq, const int16_t *__restrict__ comsseq,
                  const int32_t *__restrict__ cs_off, const int16_t *__restrict__ cs_list,
                  uint8_t *sen_active,
        const int32_t BX, const int32_t BY, int32_t *cs_need = NULL, int32_t thresh = INT_MIN, int32_t TX = -1,
        int32_t *cs_wl = NULL, int32_t *cs_wn = NULL)
{
    /* (TX: the thread's place in its M3BLOCK-wide virtual workgroup when that is not the real one: ku_frames) */
    const int32_t tx = TX >= 0 ? TX : (int32_t)threadIdx.x;
    if ((int32_t)BX < n_ent_blocks) {
        const int32_t e = BX * M3BLOCK + tx;
        int32_t idx, c;
        ent.locate_idx_wave(e < n_ent ? e : 0, idx, c);
        if (e >= n_ent) return;
        const int32_t g = (n_groups > 1 && e >= groups[4 + 1]) ? 1 : 0;
        const int32_t t = groups[4 * g], c_lo = groups[4 * g + 3];
        if (e == groups[4 * g + 1]) {                    /* first entry of the group: new list length */
            int32_t tot = 0;
            for (int32_t cc = c_lo; cc < ent.n_calls && calls[4 * cc + 3] < groups[4 * g + 2]; cc++) tot += ctot[cc];
            nnxt[t] = n0[t] + tot;
        }
        /* an entry under the threshold entered nothing (with the roots' probabilities in list order: no gather) */
        if (ent.rootprob && add32(calls[4 * c], ent.rootprob[idx]) < thresh) return;
        const int32_t v = ent.rootlist[idx];
        const int32_t fl = flag[e];
        if (fl & 1) {
            int32_t k = n0[t] + (fl >> 1);
            for (int32_t cc = c_lo; cc < c; cc++) k += ctot[cc];
            nxt[node_base[t] + k] = v; PP_SET(pos, v, k, nf);
            mark_node_senones(v, ssid, comp, sseq, comsseq, cs_off, cs_list, sen_active, cs_need, nf, (int32_t)(hist - sc), cs_wl, cs_wn);
        }
        const unsigned long long k = S3A_ALD(&key[v]);                  /* (d_dec_enter1's atomicMax / atomicMin) */
        if (k == 0ull) return;
        const int32_t win_c = 0x7fffffff - (int32_t)(uint32_t)(k & 0xffffffffu);
        if (c == win_c) { sc[NSV(v)] = (int32_t)((uint32_t)(k >> 32) ^ 0x80000000u); hist[NSV(v)] = calls[4 * c + 1]; }
        if (c == S3A_ALD(&first[v])) frame[NSV(v)] = nf;
        return;
    }
    const int32_t bb = BX - n_ent_blocks;
    const int32_t t = bb / blocks_per_tree, i = (bb % blocks_per_tree) * M3BLOCK + tx;
    if (t >= T || i >= n0[t]) return;
    mark_node_senones(nxt[node_base[t] + i], ssid, comp, sseq, comsseq, cs_off, cs_list, sen_active, cs_need, nf, (int32_t)(hist - sc), cs_wl, cs_wn);
}


#endif
