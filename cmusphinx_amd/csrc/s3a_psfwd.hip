/*
 * s3a_psfwd.hip -- pocketsphinx's first pass (the lexicon-tree Viterbi search of ngram_search_fwdtree.c) on the
 * device, behind ps_searchfuncs_t {start, step, finish} (pocketsphinx/src/libpocketsphinx/pocketsphinx_internal.h:68-81).
 * SURVEY.md 8(f).3, the search half.  References below are to pocketsphinx/src/libpocketsphinx/.
 *
 * Design (MI355X-first, not the reference's): one LANE = one utterance = one WORKGROUP.  A lane's search state
 * never meets another lane's, so the whole frame -- activate senones, normalise their scores, evaluate every
 * active HMM, prune, cross phone and word boundaries with trigram look-ups, write backpointers -- runs inside
 * ONE kernel with workgroup barriers between its phases; a launch covers a window of frames of every lane.
 * No launch per phase (a single utterance is not bound by launch latency), no cross-workgroup ordering
 * (nothing like the sphinx3 engine's stamp/scan protocol is needed), and 256 CUs run >= 256 utterances at once.
 *
 * What the reference does sequentially is reproduced in its ORDER, because the order is observable: active
 * lists are evaluated and pruned in list order, a channel entered by its parent before or after its own turn
 * ends in different states (prune_nonroot_chan :822-867), the first of equal scores wins (strict >), and the
 * backpointer table is the output.  Ordered outputs (next active list, last-phone candidates, next active word
 * list, backpointer entries and their right-context score stack) are produced by workgroup prefix sums over the
 * current lists; every channel's next state is decided from values that are immutable during the phase
 * (evaluated scores, positions in the current list), entries into active channels are parked and applied after
 * a barrier.  Interior channels have ONE parent, so "who came first" is a comparison of two list positions.
 *
 * Channel numbering: [0, n_root) roots (multiplexed), [n_root, n_ch) interior, [sp_base, +n_1ph) the
 * single-phone words' channels (multiplexed), [rc_base, n_hmm) the last-phone channels of every word that can
 * leave the tree, one per right context (ngram_search_alloc_all_rc's channels, preallocated: "allocated" is
 * frame in {f, f+1}, "freed" is the cleared state).  Per lane the state is structure-of-arrays [field][channel].
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <vector>
#include <algorithm>

#include "s3a_device.h"

#ifndef NT
#define NT 512          /* threads of a lane's workgroup (build parameter: tools/psfwd_variants.sh) */
#endif
#ifndef PSF_WPE
#define PSF_WPE 4       /* minimum waves per SIMD the search kernels are compiled for (caps the VGPRs: 2 lanes per CU need 4 with NT 512) */
#endif
#define PS_WORST ((int32_t)0xE0000000)      /* hmm.h:74 */
#define PS_TMAT_WORST (-255)                /* hmm.h:80 */
#define PS_BAD_SSID 0xffff
#define NO_BP (-1)
#define MAX_SEG 4096

struct PsfScalars {
    int32_t n_acl[2], n_awl[2];
    int32_t n_cand, bpidx, bss_head, n_frame, best_score, lp_best, dyn_beam, renorm, status, tick;
    int32_t st_root, st_nonroot, st_last, st_wlast, st_cand, st_sen;
    int32_t n_total;            /* frames of the utterance (whole-utterance mode) */
    int32_t exit_bp, exit_score, n_seg, finished;
    int32_t clean;              /* the last utterance was finished without failure: a new decoder's state is a light reset away */
    int32_t pl_best;            /* -pl_window: phone_loop_search_t.best_score */
#ifdef PSF_TIMING
    unsigned long long t_phase[16], t_last;     /* wall_clock64 ticks (100 MHz) per phase of the frame; diagnostics build */
#endif
};

struct PsfLane {
    int32_t *st;                                /* [n_hmm][CH_STRIDE]: a channel's hmm_t state in ONE 64-byte record (channels are
                                                   visited at random: one cache line per visit, not one per field) */
    uint16_t *mpxid;                            /* [n_mpx][MPX_STRIDE]: roots, then single-phone words; a channel's ids side by side (one line per visit) */
    int32_t *acl[2];                            /* active_chan_list (a channel's position in it: CH_AP, in its record) */
    int32_t *awl[2];                            /* active_word_list */
    int32_t *wstamp;                            /* word entered by last_phone_transition in tick .. */
    int32_t *lt_sf, *lt_dscr, *lt_bp;           /* last_ltrans_t */
    int32_t *ent_score, *ent_hist, *ent_stamp;  /* parked entries into active interior channels */
    int32_t *cand_wid, *cand_score, *cand_bp, *cand_ef;
    int32_t *bp_frame, *bp_wid, *bp_bp, *bp_score, *bp_sidx, *bp_realwid;
    uint8_t *bp_valid;
    int32_t *bss, *bp_idx;                      /* bp_idx[1 + frame]; [0] is the reference's bp_table_idx[-1] */
    uint8_t *flags;                             /* [n_sen] acmod->senone_active_vec as bytes (frame-synchronous mode's output) */
    int32_t *rl;                                /* [n_root] the roots active in the current frame, in index order */
    int32_t *arc;                               /* the allocated last-phone channels of the active words, in (word list, right context) order */
    int16_t *senscr;                            /* [n_sen] frame-synchronous mode: acmod_score's output */
    const int16_t *raw;                         /* whole-utterance mode: [window][n_sen] scores before normalisation */
    int32_t *pl_host;                           /* [n_ci] -pl_window, frame-synchronous mode: phone_loop_search_score of the next step (s3a_psfwd_set_lookahead) */
    uint32_t *senkeep;                          /* [(n_sen + 31) / 32] -pl_window, whole utterances: acmod->senone_active_vec as the lane's last search step left it
                                                 * (acmod_start_utt does not clear it: the next utterance's first phone loop steps score those senones too) */
    PsfScalars *sc;
    s3a_psfwd_seg_t *seg;
};

struct PsfModel {
    int32_t n_ci, sil_ci, n_emit, n_sen;
    const uint16_t *sseq;
    const uint8_t *tp;
    int32_t n_words, start_wid, finish_wid, silence_wid;
    const int32_t *w_basewid, *w_lmwid, *w_rcsize, *w_rc_base, *w_rc_row;
    const int32_t *w_lmcw;                      /* class-based LMs: a word's in-class weight (NULL: no class words) */
    const int16_t *w_first_ci, *w_last_ci, *w_last2_ci, *rc_cimap;
    const uint8_t *w_flags;
    int32_t n_root, n_nonroot, n_ch, sp_base, rc_base, n_hmm, n_mpx;
    const uint16_t *ch_ssid;                    /* [n_hmm] senone sequence of plain channels */
    const int16_t *ch_tmat;                     /* [n_hmm] */
    const int32_t *ch_par;                      /* [n_ch] parent of an interior channel */
    const int16_t *root_ci;
    const uint16_t *root_lc_ssid, *sp_lc_ssid, *root_ssid0, *sp_ssid0;
    const int32_t *ch_child_off, *ch_child, *ch_pen_off, *ch_pen_wid;
    int32_t n_1ph, n_1ph_lm, sil_sp, start_sp;
    const int32_t *sp_wid;
    const uint8_t *sp_kind;                     /* 1: entered with an LM score (word_transition :1339-1364), 2: <sil>, 4: filler loop */
    int32_t lm_order, lm_zero;
    const int32_t *ug_prob, *ug_bowt, *ug_firstbg, *bg_wid, *bg_prob, *bg_bowt, *bg_firsttg, *tg_wid, *tg_prob;
    int32_t beam, pbeam, wbeam, lpbeam, lponlybeam, fillpen, silpen, nwpen, pip, maxwpf, maxhmmpf;
    int32_t bp_cap, bss_cap, max_frames, cand_cap;
    /* -pl_window > 0 (phone_loop_search.c): the CI phones' HMMs are the channels [pl_base, pl_base + n_ci) of a lane */
    int32_t pl_window, pl_beam, pl_pbeam, pl_pip, pl_base;
    const int16_t *ch_ci;                       /* [n_ch] chan_t.ciphone / root_chan_t.ciphone */
    const int16_t *sp_ci;                       /* [n_1ph] the single-phone words' phone */
};

/* a channel's record: score[5], history[5], out_score, out_history, bestscore, frame */
#define CH_STRIDE 16
#define MPX_STRIDE 8        /* senone-sequence ids of a multiplexed channel's states (<= 5), 16 bytes per channel */
#define MPX_ID(L, k, m) (L).mpxid[(size_t)(m) * MPX_STRIDE + (k)]
#define CH_SC(L, c, k) (L).st[(size_t)(c) * CH_STRIDE + (k)]
#define CH_HI(L, c, k) (L).st[(size_t)(c) * CH_STRIDE + 5 + (k)]
#define CH_OS(L, c) (L).st[(size_t)(c) * CH_STRIDE + 10]
#define CH_OH(L, c) (L).st[(size_t)(c) * CH_STRIDE + 11]
#define CH_BE(L, c) (L).st[(size_t)(c) * CH_STRIDE + 12]
#define CH_FR(L, c) (L).st[(size_t)(c) * CH_STRIDE + 13]
/* the two spare words: an interior channel's position in active list 0 / 1.  What is looked at together lives together: a channel is
 * visited through "is it on the list, where, and what are its scores", and a position in an array of its own was one more 64-byte
 * line fetched -- and, written, one more line written back -- per visit (PMC: 792 KB of traffic per lane-frame for 220 KB of records) */
#define CH_AP(L, par, c) (L).st[(size_t)(c) * CH_STRIDE + 14 + (par)]
/* a parked entry (score, history, the tick it was parked in): with three emitting states the record's unused words 3, 8, 4;
 * with five, arrays of their own */
template <int NE> __device__ __forceinline__ int32_t &ent_score_ref(PsfLane &L, int32_t c, int32_t n_root) { return NE == 3 ? CH_SC(L, c, 3) : L.ent_score[c - n_root]; }
template <int NE> __device__ __forceinline__ int32_t &ent_hist_ref(PsfLane &L, int32_t c, int32_t n_root) { return NE == 3 ? CH_HI(L, c, 3) : L.ent_hist[c - n_root]; }
template <int NE> __device__ __forceinline__ int32_t &ent_stamp_ref(PsfLane &L, int32_t c, int32_t n_root) { return NE == 3 ? CH_SC(L, c, 4) : L.ent_stamp[c - n_root]; }

__device__ __forceinline__ int32_t add32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int32_t sub32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }

/* ------------------------------------------------------------------ */
/* workgroup primitives                                               */
/* ------------------------------------------------------------------ */
struct Wg {
    int32_t red[NT / 64][4];
    unsigned long long sred[NT / 64];
};

/* exclusive prefix sums of two counters at once (a in the high, b in the low half); totals through ta / tb */
__device__ __forceinline__ void
wg_scan2(Wg &w, int32_t a, int32_t b, int32_t &oa, int32_t &ob, int32_t &ta, int32_t &tb)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long x = ((unsigned long long)(uint32_t)a << 32) | (uint32_t)b, incl = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        unsigned long long y = __shfl_up(incl, o, 64);
        if (lane >= o) incl += y;
    }
    if (lane == 63) w.sred[wv] = incl;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; i++) { if (i < wv) base += w.sred[i]; tot += w.sred[i]; }
    __syncthreads();
    const unsigned long long ex = base + incl - x;
    oa = (int32_t)(ex >> 32); ob = (int32_t)(ex & 0xffffffffu);
    ta = (int32_t)(tot >> 32); tb = (int32_t)(tot & 0xffffffffu);
}

/* four reductions at once: max, max, sum, sum */
__device__ __forceinline__ void
wg_reduce4(Wg &w, int32_t &m0, int32_t &m1, int32_t &s0, int32_t &s1)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m0 = max(m0, __shfl_xor(m0, o, 64)); m1 = max(m1, __shfl_xor(m1, o, 64));
        s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64);
    }
    if (lane == 0) { w.red[wv][0] = m0; w.red[wv][1] = m1; w.red[wv][2] = s0; w.red[wv][3] = s1; }
    __syncthreads();
    m0 = w.red[0][0]; m1 = w.red[0][1]; s0 = w.red[0][2]; s1 = w.red[0][3];
#pragma unroll
    for (int i = 1; i < NT / 64; i++) { m0 = max(m0, w.red[i][0]); m1 = max(m1, w.red[i][1]); s0 += w.red[i][2]; s1 += w.red[i][3]; }
    __syncthreads();
}

/* ------------------------------------------------------------------ */
/* the language model: ngram_tg_score >> SENSCR_SHIFT (sphinxbase ngram_model.c:555 -> ngram_model_set.c:709  */
/* -> lm3g_templates.c:155-195, :68-95).  find_bg / find_tg step for step (:46-66, :134-152): LM files whose  */
/* runs are not sorted by word id exist, and what the reference finds in them is what these steps find.        */
/* ------------------------------------------------------------------ */
__device__ __forceinline__ int32_t
lm_find(const int32_t *__restrict__ wids, int32_t b, int32_t e, int32_t w)
{
    int32_t i;
    while (e - b > 16) {
        i = (b + e) >> 1;
        const int32_t v = wids[i];
        if (v < w) b = i + 1;
        else if (v > w) e = i;
        else return i;
    }
    for (i = b; i < e && wids[i] != w; i++) ;
    return i < e ? i : -1;
}
__device__ __forceinline__ int32_t
lm_bg(const PsfModel &M, int32_t lw1, int32_t lw2)
{
    if (lw1 < 0 || M.lm_order < 2) return M.ug_prob[lw2];
    const int32_t i = lm_find(M.bg_wid, M.ug_firstbg[lw1], M.ug_firstbg[lw1 + 1], lw2);
    if (i >= 0) return M.bg_prob[i];
    return add32(M.ug_bowt[lw1], M.ug_prob[lw2]);
}
/* w3, w2, w1: DICTIONARY base word ids (w2, w1 may be -1) */
__device__ int32_t
lm_tg_score(const PsfModel &M, int32_t w3, int32_t w2, int32_t w1)
{
    const int32_t m3 = M.w_lmwid[w3], m2 = w2 < 0 ? -1 : M.w_lmwid[w2], m1 = w1 < 0 ? -1 : M.w_lmwid[w1];
    /* class-based LMs (ngram_ng_score, sphinxbase ngram_model.c:494-521): a class word's id is its class's TAG word (target and history
     * alike), its in-class weight is added to the tag's score; weight 1 = not in the class = the LM's zero */
    const int32_t cw = M.w_lmcw ? M.w_lmcw[w3] : 0;
    if (m3 < 0 || cw == 1) return M.lm_zero;
    if (M.lm_order < 2) return add32(M.ug_prob[m3], cw);
    if (M.lm_order < 3 || m1 < 0 || m2 < 0) return add32(lm_bg(M, m2, m3), cw);
    int32_t bowt = 0, tb = 0, te = 0;
    const int32_t b = lm_find(M.bg_wid, M.ug_firstbg[m1], M.ug_firstbg[m1 + 1], m2);
    if (b >= 0) { bowt = M.bg_bowt[b]; tb = M.bg_firsttg[b]; te = M.bg_firsttg[b + 1]; }
    const int32_t i = lm_find(M.tg_wid, tb, te, m3);
    if (i >= 0) return add32(M.tg_prob[i], cw);
    return add32(add32(bowt, lm_bg(M, m2, m3)), cw);
}

/* ngram_search_exit_score, ngram_search.c:601-622 */
__device__ __forceinline__ int32_t
exit_score(const PsfModel &M, const PsfLane &L, int32_t bp, int32_t rcphone)
{
    const int32_t w = L.bp_wid[bp];
    if (M.w_last2_ci[w] == -1) return L.bss[L.bp_sidx[bp]];
    return L.bss[L.bp_sidx[bp] + M.rc_cimap[M.w_rc_row[w] * M.n_ci + rcphone]];
}
__device__ __forceinline__ int32_t
prev_real_wid(const PsfLane &L, int32_t bp)
{
    const int32_t p = L.bp_bp[bp];
    return p == NO_BP ? -1 : L.bp_realwid[p];
}

/* ------------------------------------------------------------------ */
/* hmm.c in pocketsphinx's conventions                                */
/* ------------------------------------------------------------------ */
struct SenScr {
    const int16_t *p;
    int32_t best;
    int raw;
    /* the NEGATED score hmm_vit_eval adds (hmm.h:139); raw: ms_cont_mgau_frame_eval's normalisation (ms_mgau.c:196-204) */
    __device__ __forceinline__ int32_t operator()(int32_t s) const
    {
        int32_t v = p[s];
        if (raw) { v -= best; v = v > 32767 ? 32767 : v; v = v < -32768 ? -32768 : v; }
        return -v;
    }
};

/* hmm_clear_scores :176-187 / hmm_clear :189-204 */
template <int NE> __device__ __forceinline__ void
hmm_clear_scores(const PsfModel &M, PsfLane &L, int32_t c)
{
#pragma unroll
    for (int k = 0; k < NE; k++) CH_SC(L, c, k) = PS_WORST;
    CH_OS(L, c) = PS_WORST; CH_BE(L, c) = PS_WORST;
}
template <int NE> __device__ __forceinline__ void
hmm_clear(const PsfModel &M, PsfLane &L, int32_t c)
{
#pragma unroll
    for (int k = 0; k < NE; k++) { CH_SC(L, c, k) = PS_WORST; CH_HI(L, c, k) = -1; }
    CH_OS(L, c) = PS_WORST; CH_OH(L, c) = -1; CH_BE(L, c) = PS_WORST; CH_FR(L, c) = -1;
}

/*
 * hmm_vit_eval (:789-809) for the hard-wired topologies: plain 3-state :532-609, multiplexed :612-712, plain
 * 5-state :230-345, multiplexed :350-528.  V[k] = state score + negated senone score before any update;
 * plain: state j of a 5-state model is updated only if V[j-2] is alive, the exit only if V[NE-2] is; multiplexed:
 * a state without a senone sequence contributes WORST_SCORE, a dead source WORST_SCORE instead of a sum.
 * 3-state models honour skip arcs only where the matrix has them, and the skip term survives from the exit
 * stage into state 2's (:547, :565-567).  The winner among (self, previous, skip) is self only on a strict
 * win over previous, skip only on a strict win over that winner; histories (and multiplexed ids) follow.
 */
template <int NE, bool MPX> __device__ int32_t
hmm_vit_eval(const PsfModel &M, PsfLane &L, int32_t c, int32_t m, const SenScr &sen)
{
    const uint8_t *tp = M.tp + (size_t)M.ch_tmat[c] * NE * (NE + 1);
    int32_t sc[NE], hi[NE], V[NE];
    uint16_t id[NE];
    bool bad[NE];
#define TPV(i, j) (-(int32_t)tp[(i) * (NE + 1) + (j)])
    const uint32_t ssid = MPX ? 0u : M.ch_ssid[c];
#pragma unroll
    for (int k = 0; k < NE; k++) {
        sc[k] = CH_SC(L, c, k); hi[k] = CH_HI(L, c, k);
        if (MPX) {
            id[k] = MPX_ID(L, k, m);
            bad[k] = k > 0 && id[k] == PS_BAD_SSID;
            V[k] = bad[k] ? PS_WORST : add32(sc[k], sen(M.sseq[(uint32_t)id[k] * NE + k]));
        }
        else { bad[k] = false; id[k] = 0; V[k] = add32(sc[k], sen(M.sseq[ssid * NE + k])); }
    }
    int32_t best = PS_WORST, t0, t1, t2, v;
    /* the exit state: sources NE-1 and NE-2 */
    t2 = INT_MIN;
    if (MPX || V[NE - 2] > PS_WORST) {
        t1 = bad[NE - 1] ? PS_WORST : add32(V[NE - 1], TPV(NE - 1, NE));
        if (bad[NE - 2]) t2 = PS_WORST;
        else if (NE == 5 || TPV(NE - 2, NE) > PS_TMAT_WORST) t2 = add32(V[NE - 2], TPV(NE - 2, NE));
        int32_t oh;
        if (t1 > t2) { v = t1; oh = hi[NE - 1]; } else { v = t2; oh = hi[NE - 2]; }
        if (v < PS_WORST) v = PS_WORST;
        CH_OS(L, c) = v; CH_OH(L, c) = oh;
        best = v;
    }
    /* states NE-1 .. 2: sources j, j-1, j-2 */
#pragma unroll
    for (int j = NE - 1; j >= 2; j--) {
        if (!MPX && NE == 5 && j > 2 && !(V[j - 2] > PS_WORST)) continue;
        if (MPX) {
            t0 = V[j] != PS_WORST ? add32(V[j], TPV(j, j)) : PS_WORST;
            t1 = V[j - 1] != PS_WORST ? add32(V[j - 1], TPV(j - 1, j)) : PS_WORST;
        }
        else { t0 = add32(V[j], TPV(j, j)); t1 = add32(V[j - 1], TPV(j - 1, j)); }
        if (NE == 5) t2 = (MPX && j > 2 && bad[j - 2]) ? PS_WORST : add32(V[j - 2], TPV(j - 2, j));
        else if (TPV(0, 2) > PS_TMAT_WORST) t2 = add32(V[0], TPV(0, 2));      /* else: the exit stage's t2 */
        int32_t nh = hi[j];
        uint16_t nid = id[j];
        if (t0 > t1) {
            if (t2 > t0) { v = t2; nh = hi[j - 2]; nid = id[j - 2]; } else v = t0;
        }
        else {
            if (t2 > t1) { v = t2; nh = hi[j - 2]; nid = id[j - 2]; } else { v = t1; nh = hi[j - 1]; nid = id[j - 1]; }
        }
        if (v < PS_WORST) v = PS_WORST;
        if (v > best) best = v;
        /* sources of the lower states are V[] and the OLD hi[] / id[] of lower indices: safe to store now */
        CH_SC(L, c, j) = v; CH_HI(L, c, j) = nh;
        if (MPX) MPX_ID(L, j, m) = nid;
    }
    /* state 1 */
    t0 = MPX ? (V[1] != PS_WORST ? add32(V[1], TPV(1, 1)) : PS_WORST) : add32(V[1], TPV(1, 1));
    t1 = add32(V[0], TPV(0, 1));
    if (t0 > t1) v = t0;
    else {
        v = t1;
        CH_HI(L, c, 1) = hi[0];
        if (MPX) MPX_ID(L, 1, m) = id[0];
    }
    if (v < PS_WORST) v = PS_WORST;
    if (v > best) best = v;
    CH_SC(L, c, 1) = v;
    /* state 0 */
    v = add32(V[0], TPV(0, 0));
    if (v < PS_WORST) v = PS_WORST;
    if (v > best) best = v;
    CH_SC(L, c, 0) = v;
    CH_BE(L, c) = best;
    return best;
#undef TPV
}

/* acmod_activate_hmm, acmod.c:1173-1214: acmod->senone_active_vec is a bit vector in LDS */
#define SENBITS_WORDS 2048      /* 65536 senones (s3senid_t is 16 bits) */
#define ROOTBITS_WORDS 1024     /* 32768 root channels */
__device__ __forceinline__ void
setbit(uint32_t *bits, uint32_t i) { atomicOr(&bits[i >> 5], 1u << (i & 31)); }

template <int NE, bool MPX> __device__ __forceinline__ void
activate(const PsfModel &M, const PsfLane &L, int32_t c, int32_t m, uint32_t *bits)
{
    if (MPX) {
#pragma unroll
        for (int k = 0; k < NE; k++) {
            const uint16_t id = MPX_ID(L, k, m);
            if (id != PS_BAD_SSID) setbit(bits, M.sseq[(uint32_t)id * NE + k]);
        }
    }
    else {
        const uint32_t ssid = M.ch_ssid[c];
#pragma unroll
        for (int k = 0; k < NE; k++) setbit(bits, M.sseq[ssid * NE + k]);
    }
}

/* the roots active in frame f, in index order, from their bit vector: L.rl[0 .. return) */
__device__ int32_t
d_root_list(const PsfModel &M, PsfLane &L, Wg &wg, const uint32_t *rootbits)
{
    const int32_t nw = (M.n_root + 31) >> 5;
    int32_t base = 0;
    for (int32_t w0 = 0; w0 < nw; w0 += NT) {
        const int32_t w = w0 + threadIdx.x;
        uint32_t v = w < nw ? rootbits[w] : 0u;
        int32_t off, o2, tot, t2;
        wg_scan2(wg, __popc(v), 0, off, o2, tot, t2);
        off += base;
        while (v) { const int b = __ffs(v) - 1; v &= v - 1; L.rl[off++] = (w << 5) + b; }
        base += tot;
    }
    __syncthreads();
    return base;
}

/* compute_sen_active :513-552 into the bit vector */
template <int NE> __device__ void
d_sen_active(const PsfModel &M, PsfLane &L, const PsfScalars &S, int32_t f, uint32_t *senbits, int32_t n_rl, int32_t n_arc)
{
    const int tid = threadIdx.x, cur = f & 1;
    for (int32_t i = tid; i < ((M.n_sen + 31) >> 5); i += NT) senbits[i] = 0;
    __syncthreads();
    for (int32_t j = tid; j < n_rl; j += NT) { const int32_t i = L.rl[j]; activate<NE, true>(M, L, i, i, senbits); }
    for (int32_t j = tid; j < S.n_acl[cur]; j += NT) activate<NE, false>(M, L, L.acl[cur][j], 0, senbits);
    if (n_arc >= 0)
        for (int32_t j = tid; j < n_arc; j += NT) activate<NE, false>(M, L, L.arc[j], 0, senbits);
    else
        for (int32_t j = tid; j < S.n_awl[cur]; j += NT) {
            const int32_t w = L.awl[cur][j], c0 = M.rc_base + M.w_rc_base[w];
            for (int32_t r = 0; r < M.w_rcsize[w]; r++)
                if (CH_FR(L, c0 + r) == f) activate<NE, false>(M, L, c0 + r, 0, senbits);
        }
    for (int32_t i = tid; i < M.n_1ph; i += NT)
        if (CH_FR(L, M.sp_base + i) == f) activate<NE, true>(M, L, M.sp_base + i, M.n_root + i, senbits);
    __syncthreads();
}

/*
 * What acmod_score computes for a frame in whole-utterance mode: acmod_flags2list (acmod.c:1220-1271) turns the
 * flags into a delta list in which a gap over 255 is bridged by extra entries -- senones that are then scored and
 * take part in the normalisation like any other (ms_mgau.c:219-233); the best (smallest) score over all listed
 * senones is the frame's normaliser.  Returns it through *best, the list length through *count.
 * One thread per 32-senone word of the bit vector; the previous active senone of a word's first one comes from a
 * workgroup max-scan of the words' highest set bits (the list starts its deltas at senone 0).
 */
__device__ void
d_normaliser(const PsfModel &M, Wg &wg, int32_t *sh, const uint32_t *senbits, const int16_t *raw, int compallsen,
             int32_t *best, int32_t *count)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int32_t mn = INT_MAX, cnt = 0;
    if (compallsen) {
        for (int32_t s = tid; s < M.n_sen; s += NT) mn = min(mn, (int32_t)raw[s]);
        cnt = tid == 0 ? M.n_sen : 0;
    }
    else {
        const int32_t nw = (M.n_sen + 31) >> 5;
        int32_t carry = 0;          /* the last active senone before this chunk of words (0: the list's origin) */
        for (int32_t w0 = 0; w0 < nw; w0 += NT) {
            const int32_t w = w0 + tid;
            uint32_t v = w < nw ? senbits[w] : 0u;
            /* inclusive max-scan of the highest active senone per word */
            int32_t hi = v ? (w << 5) + 31 - __clz(v) : -1, x = hi;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int32_t y = __shfl_up(x, o, 64); if (lane >= o) x = max(x, y); }
            if (lane == 63) sh[wv] = x;
            __syncthreads();
            int32_t before = -1, all = -1;
#pragma unroll
            for (int i = 0; i < NT / 64; i++) { if (i < wv) before = max(before, sh[i]); all = max(all, sh[i]); }
            int32_t prevx = __shfl_up(x, 1, 64);
            if (lane == 0) prevx = -1;
            int32_t prev = max(before, prevx);
            if (prev < 0) prev = carry;
            __syncthreads();
            while (v) {
                const int32_t s = (w << 5) + __ffs(v) - 1;
                v &= v - 1;
                for (int32_t b = prev + 255; b < s; b += 255) { mn = min(mn, (int32_t)raw[b]); cnt++; }
                mn = min(mn, (int32_t)raw[s]); cnt++;
                prev = s;
            }
            if (all >= 0) carry = all;
        }
    }
    int32_t a = -mn, b = INT_MIN, z = 0;
    wg_reduce4(wg, a, b, cnt, z);
    *best = -a; *count = cnt;
}

#ifdef PSF_TIMING
#define TPHASE(S, k) do { if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); (S).t_phase[k] += t_ - (S).t_last; (S).t_last = t_; } } while (0)
#else
#define TPHASE(S, k) do { } while (0)
#endif

/* ------------------------------------------------------------------ */
/* one frame of one lane: ngram_fwdtree_search :1446-1488             */
/* ------------------------------------------------------------------ */
struct FrameShared {
    Wg wg;
    PsfScalars S;
    int32_t sh[2 * NT];
    unsigned long long key[NT];
    int32_t bins[256];
    int32_t brc_score[256], brc_path[256], brc_lc[256];
    int32_t carryA, carryB, carryC, misc[8];
    int32_t n_rl, n_arc;
    int32_t pa[4][NT];                          /* a chunk of parents in prune: first item, channel, list position, appends itself */
    uint32_t senbits[SENBITS_WORDS];            /* acmod->senone_active_vec */
    uint32_t rootbits[ROOTBITS_WORDS];          /* roots active in the NEXT frame (set while the frame runs) */
    int32_t pl[256];                            /* -pl_window: phone_loop_search_score(ci) of the frame (phone_loop_search.h:103-105) */
};

/* at a kernel's start: the bit vector of the roots active in frame f, from the channels' frame numbers */
__device__ void
d_root_bits_from_frames(const PsfModel &M, const PsfLane &L, uint32_t *rootbits, int32_t f)
{
    for (int32_t i = threadIdx.x; i < ((M.n_root + 31) >> 5); i += NT) rootbits[i] = 0;
    __syncthreads();
    for (int32_t i = threadIdx.x; i < M.n_root; i += NT) if (CH_FR(L, i) == f) setbit(rootbits, i);
    __syncthreads();
}
/* at a frame's start: rootbits -> L.rl, then cleared to collect the next frame's */
__device__ void
d_frame_roots(const PsfModel &M, PsfLane &L, FrameShared &F)
{
    const int32_t n = d_root_list(M, L, F.wg, F.rootbits);
    if (threadIdx.x == 0) F.n_rl = n;
    for (int32_t i = threadIdx.x; i < ((M.n_root + 31) >> 5); i += NT) F.rootbits[i] = 0;
    __syncthreads();
}
/* at a frame's start: the word_chan lists of the active words (eval_word_chan's walk, :645-662) as one flat list */
__device__ void
d_frame_word_chans(const PsfModel &M, PsfLane &L, FrameShared &F, int32_t f)
{
    const int32_t n_awl = F.S.n_awl[f & 1];
    int32_t base = 0;
    for (int32_t j0 = 0; j0 < n_awl; j0 += NT) {
        const int32_t j = j0 + threadIdx.x;
        int32_t cnt = 0, c0 = 0, rcs = 0;
        unsigned long long live = 0ull;             /* (the first 64 right contexts' frame tests, kept: a record is visited once) */
        if (j < n_awl) {
            const int32_t w = L.awl[f & 1][j];
            c0 = M.rc_base + M.w_rc_base[w]; rcs = M.w_rcsize[w];
            for (int32_t r = 0; r < rcs; r++) {
                const bool on = CH_FR(L, c0 + r) == f;
                cnt += on ? 1 : 0;
                if (on && r < 64) live |= 1ull << r;
            }
        }
        int32_t off, o2, tot, t2;
        wg_scan2(F.wg, cnt, 0, off, o2, tot, t2);
        off += base;
        for (int32_t r = 0; r < rcs; r++) if (r < 64 ? (live >> r) & 1ull : CH_FR(L, c0 + r) == f) L.arc[off++] = c0 + r;
        base += tot;
    }
    if (threadIdx.x == 0) F.n_arc = base;
    __syncthreads();
}

/* PL: -pl_window > 0 -- every transition into a phone adds F.pl[its CI phone] (phone_loop_search_score), and the tests that stand in
 * front of the transitions' loops in the reference (`pls != NULL || ...`, :745, :765, :824, :847) are not made */
template <int NE, bool PL> __device__ void
d_frame(const PsfModel &M, PsfLane &L, FrameShared &F, int32_t f, const SenScr &sen, int32_t n_senone_active)
{
    PsfScalars &S = F.S;
    const int tid = threadIdx.x, cur = f & 1, nxt = cur ^ 1, nf = f + 1;

    if (tid == 0) {
        S.st_sen += n_senone_active;
        L.bp_idx[1 + f] = S.bpidx;                                      /* ngram_search_mark_bptable */
        S.tick++;
    }
    __syncthreads();
    if (S.best_score == PS_WORST || S.best_score < PS_WORST) return;    /* :1463: recognition has failed */
    const int32_t tick = S.tick;

    /* ---- renormalize_scores :555-592 ---- */
    if (add32(S.best_score, 2 * M.beam) < PS_WORST) {
        const int32_t norm = S.best_score;
        auto normalize = [&](int32_t c) {
#pragma unroll
            for (int k = 0; k < NE; k++) { const int32_t v = CH_SC(L, c, k); if (v > PS_WORST) CH_SC(L, c, k) = sub32(v, norm); }
            const int32_t o = CH_OS(L, c);
            if (o > PS_WORST) CH_OS(L, c) = sub32(o, norm);
        };
        for (int32_t j = tid; j < F.n_rl; j += NT) normalize(L.rl[j]);
        for (int32_t j = tid; j < S.n_acl[cur]; j += NT) normalize(L.acl[cur][j]);
        for (int32_t j = tid; j < F.n_arc; j += NT) normalize(L.arc[j]);
        for (int32_t i = tid; i < M.n_1ph; i += NT) if (CH_FR(L, M.sp_base + i) == f) normalize(M.sp_base + i);
        if (tid == 0) S.renorm = 1;
        __syncthreads();
    }

    TPHASE(S, 2);
    /* ---- evaluate_channels :694-706 ---- */
    {
        int32_t mx = PS_WORST, lp = PS_WORST, n_rt = 0, kj = 0;
        for (int32_t j = tid; j < F.n_rl; j += NT) { const int32_t i = L.rl[j]; mx = max(mx, hmm_vit_eval<NE, true>(M, L, i, i, sen)); n_rt++; }
        for (int32_t j = tid; j < S.n_acl[cur]; j += NT) mx = max(mx, hmm_vit_eval<NE, false>(M, L, L.acl[cur][j], 0, sen));
        for (int32_t j = tid; j < F.n_arc; j += NT) { lp = max(lp, hmm_vit_eval<NE, false>(M, L, L.arc[j], 0, sen)); kj++; }
        int32_t j1 = 0;
        for (int32_t i = tid; i < M.n_1ph; i += NT) {
            const int32_t c = M.sp_base + i;
            if (CH_FR(L, c) < f) continue;
            const int32_t b = hmm_vit_eval<NE, true>(M, L, c, M.n_root + i, sen);
            if (M.sp_wid[i] != M.finish_wid) lp = max(lp, b);
            j1++;
        }
        kj += j1;
        /* sums: roots evaluated; k + j.  j alone is needed too: a second reduction */
        wg_reduce4(F.wg, mx, lp, n_rt, kj);
        int32_t d0 = INT_MIN, d1 = INT_MIN, z = 0;
        wg_reduce4(F.wg, d0, d1, j1, z);
        if (tid == 0) {
            S.st_root += n_rt;
            S.st_nonroot += S.n_acl[cur] + kj;
            S.st_last += kj;
            S.st_wlast += S.n_awl[cur] + j1;
            S.best_score = max(mx, lp);
            S.lp_best = lp;
        }
        __syncthreads();
    }

    TPHASE(S, 3);
    /* ---- prune_channels :1125-1177: the dynamic beam ---- */
    if (tid == 0) { S.n_cand = 0; S.dyn_beam = M.beam; }
    const int32_t bw = -M.beam / 256;
    if (M.maxhmmpf != -1 && S.st_root + S.st_nonroot > M.maxhmmpf && bw != 0) {     /* block-uniform */
        for (int i = tid; i < 256; i += NT) F.bins[i] = 0;
        __syncthreads();
        /* every root channel is binned (:1143-1151); one that is not active has bestscore WORST_SCORE */
        for (int32_t j = tid; j < F.n_rl; j += NT) {
            int32_t b = sub32(S.best_score, CH_BE(L, L.rl[j])) / bw;
            atomicAdd(&F.bins[b >= 256 ? 255 : b], 1);
        }
        if (tid == 0) {
            int32_t b = sub32(S.best_score, PS_WORST) / bw;
            atomicAdd(&F.bins[b >= 256 ? 255 : b], M.n_root - F.n_rl);
        }
        for (int32_t j = tid; j < S.n_acl[cur]; j += NT) {
            int32_t b = sub32(S.best_score, CH_BE(L, L.acl[cur][j])) / bw;
            atomicAdd(&F.bins[b >= 256 ? 255 : b], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int32_t i, n = 0;
            for (i = 0; i < 256; ++i) { n += F.bins[i]; if (n > M.maxhmmpf) break; }
            S.dyn_beam = -(i * bw);
        }
    }
    __syncthreads();

    const int32_t thresh = add32(S.best_score, S.dyn_beam);
    const int32_t newphone_thresh = add32(S.best_score, M.pbeam), lastphn_thresh = add32(S.best_score, M.lpbeam);
    const int32_t n_acl = S.n_acl[cur];
    int32_t *acl = L.acl[cur], *nacl = L.acl[nxt];
#define APOS(c) CH_AP(L, cur, c)

    /* ---- prune_root_chan :714-788 + prune_nonroot_chan :794-870, outputs in the reference's order ---- */
    auto in_acl = [&](int32_t c) -> bool { const int32_t p = APOS(c); return p >= 0 && p < n_acl && acl[p] == c; };
    /* does parent p (position ppos in the list, -1 = root) enter child x?  everything read is immutable in this phase */
    auto enters = [&](int32_t p, int32_t ppos, int32_t x, int32_t *ns_out) -> bool {
        if (!(CH_BE(L, p) > thresh)) return false;
        int32_t ns = add32(CH_OS(L, p), M.pip);
        if (PL) ns = add32(ns, F.pl[M.ch_ci[x]]);
        *ns_out = ns;
        if (!(ns > newphone_thresh)) return false;
        if (!in_acl(x)) return (CH_FR(L, x) < f) || (ns > CH_SC(L, x, 0));
        const bool pfirst = ppos < 0 || ppos < APOS(x);
        if (pfirst || CH_BE(L, x) > thresh) return ns > CH_SC(L, x, 0);
        return ns > PS_WORST;       /* x's turn came first and cleared it (:866-867) */
    };
    if (tid == 0) { F.carryA = 0; F.carryC = 0; }
    __syncthreads();
    const int32_t n_rl = F.n_rl, n_par = n_rl + n_acl;
    /* the next list's entries come in the order (parent in list order: itself, then its entered children in sibling
     * order).  Per chunk of NT parents: every parent's item count (1 + children if it propagates), then the items
     * FLAT over the threads -- a root with dozens of children does not hold a thread for dozens of turns. */
    for (int32_t base = 0; base < n_par; base += NT) {
        const int32_t p = base + tid;
        int32_t items = 0, cntC = 0, c = -1, pos = -1;
        bool surv = false, selfapp = false;
        if (p < n_par) {
            if (p < n_rl) c = L.rl[p];
            else { pos = p - n_rl; c = acl[pos]; }
            items = 1;
            surv = CH_BE(L, c) > thresh;
            if (pos >= 0 && surv) {
                /* :822-826 `if (hmm_frame != nf)`: unless the parent's turn came first and entered this channel */
                const int32_t par = M.ch_par[c];
                int32_t ns;
                bool par_first_entered = false;
                if (par < M.n_root) par_first_entered = !(CH_FR(L, par) < f) && enters(par, -1, c, &ns);
                else if (in_acl(par) && APOS(par) < pos) par_first_entered = enters(par, APOS(par), c, &ns);
                selfapp = !par_first_entered;
            }
            if (surv) {
                const int32_t ns0 = add32(CH_OS(L, c), M.pip);
                if (PL || ns0 > newphone_thresh) items += M.ch_child_off[c + 1] - M.ch_child_off[c];
                if (PL) {       /* (a word is a candidate when its own last phone's look-ahead lets it: :768-772, :850-853) */
                    for (int32_t e = M.ch_pen_off[c]; e < M.ch_pen_off[c + 1]; e++)
                        cntC += add32(ns0, F.pl[M.w_last_ci[M.ch_pen_wid[e]]]) > lastphn_thresh ? 1 : 0;
                }
                else if (ns0 > lastphn_thresh) cntC = M.ch_pen_off[c + 1] - M.ch_pen_off[c];
            }
        }
        int32_t io, oc, ti, tc;
        wg_scan2(F.wg, items, cntC, io, oc, ti, tc);
        F.pa[0][tid] = io; F.pa[1][tid] = c; F.pa[2][tid] = pos; F.pa[3][tid] = selfapp ? 1 : 0;
        oc += F.carryC;
        if (surv) {
            if (pos < 0) { CH_FR(L, c) = nf; setbit(F.rootbits, c); }                    /* :733 */
            if (cntC > 0) {
                const int32_t ns0 = add32(CH_OS(L, c), M.pip), ch = CH_OH(L, c);
                for (int32_t e = M.ch_pen_off[c]; e < M.ch_pen_off[c + 1]; e++) {
                    const int32_t w = M.ch_pen_wid[e];
                    int32_t v = ns0;
                    if (PL) { v = add32(ns0, F.pl[M.w_last_ci[w]]); if (!(v > lastphn_thresh)) continue; }
                    L.cand_wid[oc] = w; L.cand_score[oc] = sub32(v, M.nwpen); L.cand_bp[oc] = ch;
                    oc++;
                }
            }
        }
        __syncthreads();
        for (int32_t qb = 0; qb < ti; qb += NT) {
            const int32_t q = qb + tid;
            int32_t flag = 0, item = -1;
            if (q < ti) {
                int lo = 0, hi = NT - 1;            /* the last parent whose first item is <= q */
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (F.pa[0][mid] <= q) lo = mid; else hi = mid - 1; }
                const int32_t k = q - F.pa[0][lo], pc = F.pa[1][lo], ppos = F.pa[2][lo];
                if (k == 0) { flag = F.pa[3][lo]; item = pc; }
                else {
                    const int32_t x = M.ch_child[M.ch_child_off[pc] + k - 1];
                    int32_t ns;
                    if (enters(pc, ppos, x, &ns)) {
                        const bool xin = in_acl(x);
                        bool app;
                        if (ppos < 0) app = true;                                       /* :750-752 */
                        else if (xin) app = !(CH_BE(L, x) > thresh && APOS(x) < ppos);
                        else app = CH_FR(L, x) != nf;                                    /* :833-836 */
                        flag = app ? 1 : 0; item = x;
                        if (xin) {      /* parked: x's own state is still being read by others */
                            ent_score_ref<NE>(L, x, M.n_root) = ns; ent_hist_ref<NE>(L, x, M.n_root) = CH_OH(L, pc); ent_stamp_ref<NE>(L, x, M.n_root) = tick;
                        }
                        else { CH_SC(L, x, 0) = ns; CH_HI(L, x, 0) = CH_OH(L, pc); CH_FR(L, x) = nf; }  /* hmm_enter; only its parent touches an inactive channel */
                    }
                }
            }
            int32_t oa, oz, ta, tz;
            wg_scan2(F.wg, flag, 0, oa, oz, ta, tz);
            if (flag) { nacl[F.carryA + oa] = item; CH_AP(L, nxt, item) = F.carryA + oa; }
            __syncthreads();
            if (tid == 0) F.carryA += ta;
            __syncthreads();
        }
        if (tid == 0) F.carryC += tc;
        __syncthreads();
    }
    const int32_t n_nacl = F.carryA, n_cand = F.carryC;
    /* channels of this frame that did not survive: cleared unless the parent came first and entered them (:866-867) */
    for (int32_t j = tid; j < n_acl; j += NT) {
        const int32_t c = acl[j];
        if (CH_BE(L, c) > thresh) continue;
        bool keep = false;
        if (ent_stamp_ref<NE>(L, c, M.n_root) == tick) {
            const int32_t par = M.ch_par[c];
            keep = par < M.n_root || APOS(par) < j;
        }
        if (!keep) hmm_clear_scores<NE>(M, L, c);
    }
    __syncthreads();
    /* the parked entries; every channel of the next list is active in f + 1 */
    for (int32_t j = tid; j < n_nacl; j += NT) {
        const int32_t x = nacl[j];
        if (ent_stamp_ref<NE>(L, x, M.n_root) == tick) { CH_SC(L, x, 0) = ent_score_ref<NE>(L, x, M.n_root); CH_HI(L, x, 0) = ent_hist_ref<NE>(L, x, M.n_root); }
        CH_FR(L, x) = nf;
    }
    if (tid == 0) { S.n_acl[nxt] = n_nacl; S.n_cand = n_cand; S.st_cand += n_cand; }
    __syncthreads();
    TPHASE(S, 4);

    /* ---- last_phone_transition :877-1030 ---- */
    {
        /* start scores off; which (word, start frame) pairs are new */
        for (int32_t i = tid; i < n_cand; i += NT) {
            const int32_t bp = L.cand_bp[i], w = L.cand_wid[i];
            int32_t ef = -2;
            if (bp != -1) {
                const int32_t e = L.bp_frame[bp];
                L.cand_score[i] = sub32(L.cand_score[i], exit_score(M, L, bp, M.w_first_ci[w]));
                if (L.lt_sf[w] != e + 1) { L.lt_dscr[w] = PS_WORST; L.lt_sf[w] = e + 1; ef = e; }
            }
            L.cand_ef[i] = ef;
        }
        __syncthreads();
        /* the best predecessor of every new pair: all (candidate, backpointer of its start frame) pairs, flat */
        for (int32_t base = 0; base < n_cand; base += NT) {
            const int32_t i = base + tid;
            int32_t cnt = 0, b0 = 0;
            if (i < n_cand && L.cand_ef[i] >= -1) { b0 = L.bp_idx[1 + L.cand_ef[i]]; cnt = L.bp_idx[1 + L.cand_ef[i] + 1] - b0; }
            int32_t off, o2, tot, t2;
            wg_scan2(F.wg, cnt, 0, off, o2, tot, t2);
            F.sh[tid] = off; F.sh[NT + tid] = b0;
            F.key[tid] = 0ull;
            __syncthreads();
            for (int32_t q = tid; q < tot; q += NT) {
                int lo = 0, hi = NT - 1;            /* the last candidate whose offset is <= q */
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (F.sh[mid] <= q) lo = mid; else hi = mid - 1; }
                const int32_t ci = base + lo, bp = F.sh[NT + lo] + (q - F.sh[lo]);
                if (!L.bp_valid[bp]) continue;
                const int32_t w = L.cand_wid[ci];
                int32_t dscr = exit_score(M, L, bp, M.w_first_ci[w]);
                if (dscr != PS_WORST)
                    dscr = add32(dscr, lm_tg_score(M, M.w_basewid[w], L.bp_realwid[bp], prev_real_wid(L, bp)) >> 10);
                if (dscr > PS_WORST)        /* strict: the earliest of equals wins (:965-968) */
                    atomicMax(&F.key[lo], ((unsigned long long)((uint32_t)dscr ^ 0x80000000u) << 32) | (uint32_t)(0x7fffffff - bp));
            }
            __syncthreads();
            if (i < n_cand && F.key[tid] != 0ull) {
                const int32_t w = L.cand_wid[i];
                L.lt_dscr[w] = (int32_t)((uint32_t)(F.key[tid] >> 32) ^ 0x80000000u);
                L.lt_bp[w] = 0x7fffffff - (int32_t)(F.key[tid] & 0xffffffffu);
            }
            __syncthreads();
        }
        /* totals with the LM; the best last-phone score */
        int32_t mx = S.lp_best, d1 = INT_MIN, z0 = 0, z1 = 0;
        for (int32_t i = tid; i < n_cand; i += NT) {
            const int32_t w = L.cand_wid[i], sc = add32(L.cand_score[i], L.lt_dscr[w]);
            L.cand_score[i] = sc; L.cand_bp[i] = L.lt_bp[w];
            mx = max(mx, sc);
        }
        wg_reduce4(F.wg, mx, d1, z0, z1);
        if (tid == 0) { S.lp_best = mx; F.carryA = 0; }
        __syncthreads();
        const int32_t lthresh = add32(mx, M.lponlybeam);
        /* into every right context of the candidates within the beam (:1001-1027) */
        for (int32_t base = 0; base < n_cand; base += NT) {
            const int32_t i = base + tid;
            int32_t k = 0, w = -1;
            if (i < n_cand && L.cand_score[i] > lthresh) {
                w = L.cand_wid[i];
                const int32_t sc = L.cand_score[i], bp = L.cand_bp[i], c0 = M.rc_base + M.w_rc_base[w];
                for (int32_t r = 0; r < M.w_rcsize[w]; r++) {
                    const int32_t c = c0 + r;
                    if (CH_FR(L, c) < f || sc > CH_SC(L, c, 0)) { CH_SC(L, c, 0) = sc; CH_HI(L, c, 0) = bp; CH_FR(L, c) = nf; k++; }
                }
            }
            int32_t off, o2, tot, t2;
            wg_scan2(F.wg, k > 0 ? 1 : 0, 0, off, o2, tot, t2);
            if (k > 0) { L.awl[nxt][F.carryA + off] = w; L.wstamp[w] = tick; }
            __syncthreads();
            if (tid == 0) F.carryA += tot;
            __syncthreads();
        }
    }

    TPHASE(S, 5);
    /* ---- prune_word_chan :1037-1122 with ngram_search_save_bp (ngram_search.c:360-441) ---- */
    {
        const int32_t newword_thresh = add32(S.lp_best, M.wbeam), lpo_thresh = add32(S.lp_best, M.lponlybeam);
        const int32_t n_awl = S.n_awl[cur], n_items = n_awl + M.n_1ph;
        if (tid == 0) { F.carryB = S.bpidx; F.carryC = S.bss_head; }
        __syncthreads();
        for (int32_t base = 0; base < n_items; base += NT) {
            const int32_t q = base + tid;
            int32_t w = -1, k = 0, ex_score = 0, ex_hist = 0, n_ex = 0, rcs = 0, c0 = 0;
            if (q < n_awl) {
                w = L.awl[cur][q]; c0 = M.rc_base + M.w_rc_base[w]; rcs = M.w_rcsize[w];
                for (int32_t r = 0; r < rcs; r++) {
                    const int32_t c = c0 + r, fr = CH_FR(L, c);
                    if (fr < f) continue;                                   /* not allocated */
                    if (CH_BE(L, c) > lpo_thresh) {
                        CH_FR(L, c) = nf; k++;
                        const int32_t o = CH_OS(L, c);
                        if (o > newword_thresh) { if (n_ex == 0 || ex_score < o) { ex_score = o; ex_hist = CH_OH(L, c); } n_ex++; }
                    }
                    else if (fr != nf) hmm_clear<NE>(M, L, c);              /* listelem_free */
                }
            }
            else if (q < n_items) {
                const int32_t i = q - n_awl, c = M.sp_base + i;
                if (!(CH_FR(L, c) < f) && CH_BE(L, c) > lpo_thresh) {
                    CH_FR(L, c) = nf;
                    const int32_t o = CH_OS(L, c);
                    if (o > newword_thresh) { w = M.sp_wid[i]; ex_score = o; ex_hist = CH_OH(L, c); n_ex = 1; rcs = 1; }
                }
            }
            const int32_t has = n_ex > 0 ? 1 : 0;
            const int32_t app = (q < n_awl && k > 0 && L.wstamp[w] != tick) ? 1 : 0;
            int32_t ob, os, tb, ts, oa, oz, ta, tz;
            wg_scan2(F.wg, has, has ? rcs : 0, ob, os, tb, ts);
            wg_scan2(F.wg, app, 0, oa, oz, ta, tz);
            if (app) L.awl[nxt][F.carryA + oa] = w;
            if (has) {
                const int32_t bp = F.carryB + ob, si = F.carryC + os;
                if (bp < M.bp_cap && si + rcs <= M.bss_cap) {
                    L.bp_wid[bp] = w; L.bp_frame[bp] = f; L.bp_bp[bp] = ex_hist; L.bp_score[bp] = ex_score;
                    L.bp_sidx[bp] = si; L.bp_valid[bp] = 1;
                    if (q < n_awl) {
                        for (int32_t r = 0; r < rcs; r++) {
                            const int32_t c = c0 + r, o = CH_OS(L, c);
                            L.bss[si + r] = (CH_BE(L, c) > lpo_thresh && o > newword_thresh) ? o : PS_WORST;
                        }
                    }
                    else L.bss[si] = ex_score;
                    /* set_real_wid, ngram_search.c:343-358 */
                    if (M.w_flags[w] & S3A_PSW_FILLER) { if (ex_hist != NO_BP) L.bp_realwid[bp] = L.bp_realwid[ex_hist]; }
                    else L.bp_realwid[bp] = M.w_basewid[w];
                }
            }
            __syncthreads();
            if (tid == 0) { F.carryA += ta; F.carryB += tb; F.carryC += ts; }
            __syncthreads();
        }
        if (tid == 0) {
            S.n_awl[nxt] = F.carryA;
            if (F.carryB > M.bp_cap || F.carryC > M.bss_cap) { S.status = S3A_ENOMEM; F.carryB = S.bpidx; F.carryC = S.bss_head; }
            F.misc[0] = S.bpidx;        /* first entry of this frame */
            S.bpidx = F.carryB; S.bss_head = F.carryC;
        }
        __syncthreads();
    }
    const int32_t bp0 = F.misc[0], bp1 = S.bpidx, n_ent = bp1 - bp0;
    TPHASE(S, 6);

    /* ---- bptable_maxwpf :1183-1233 ---- */
    if (M.maxwpf != -1 && M.maxwpf != M.n_words && n_ent > 0) {
        /* one filler exit per frame: the best, the first of equals */
        unsigned long long kb = 0ull;
        int32_t nfill = 0;
        for (int32_t e = bp0 + tid; e < bp1; e += NT)
            if (M.w_flags[L.bp_wid[e]] & S3A_PSW_FILLER) {
                nfill++;
                const unsigned long long k = ((unsigned long long)((uint32_t)L.bp_score[e] ^ 0x80000000u) << 32) | (uint32_t)(0x7fffffff - e);
                kb = k > kb ? k : kb;
                L.bp_valid[e] = 0;
            }
        if (tid == 0) F.key[0] = 0ull;
        __syncthreads();
        if (kb) atomicMax(&F.key[0], kb);
        int32_t d0 = INT_MIN, d1 = INT_MIN, z = 0;
        wg_reduce4(F.wg, d0, d1, nfill, z);
        __syncthreads();
        /* bestscr starts at INT_MIN and the test is strict: a filler with that score is never the best */
        const bool have_best = F.key[0] != 0ull && (int32_t)((uint32_t)(F.key[0] >> 32) ^ 0x80000000u) > INT_MIN;
        if (tid == 0 && have_best) L.bp_valid[0x7fffffff - (int32_t)(F.key[0] & 0xffffffffu)] = 1;
        __syncthreads();
        const int32_t n_valid = n_ent - (nfill - (have_best ? 1 : 0)), n_drop = n_valid - M.maxwpf;
        if (n_drop > 0) {
            /* the n_drop worst valid entries go, the first of equals first: rank by (score, index) */
            for (int32_t e = bp0 + tid; e < bp1; e += NT) {
                if (!L.bp_valid[e]) continue;
                const int32_t se = L.bp_score[e];
                int32_t rank = 0;
                for (int32_t o = bp0; o < bp1; o++)
                    if (L.bp_valid[o]) {
                        const int32_t so = L.bp_score[o];
                        rank += (so < se || (so == se && o < e)) ? 1 : 0;
                    }
                if (rank < n_drop) L.bp_valid[e] = 3;      /* marked; still counts as valid for the others */
            }
            __syncthreads();
            for (int32_t e = bp0 + tid; e < bp1; e += NT) if (L.bp_valid[e] == 3) L.bp_valid[e] = 0;
        }
        __syncthreads();
    }

    TPHASE(S, 7);
    /* ---- word_transition :1236-1405 ---- */
    {
        /* the best exit per right-context phone over ALL entries of the frame (valid or not) except </s> */
        int32_t k = 0;
        for (int32_t e = bp0 + tid; e < bp1; e += NT) k += L.bp_wid[e] != M.finish_wid ? 1 : 0;
        int32_t d0 = INT_MIN, d1 = INT_MIN, z = 0;
        wg_reduce4(F.wg, d0, d1, k, z);
        if (k > 0) {        /* block-uniform */
            for (int32_t rc = tid; rc < M.n_ci; rc += NT) {
                int32_t bs = PS_WORST, bpth = 0, blc = 0;
                for (int32_t e = bp0; e < bp1; e++) {
                    const int32_t w = L.bp_wid[e];
                    if (w == M.finish_wid) continue;
                    const int32_t *rcss = L.bss + L.bp_sidx[e];
                    const int32_t v = M.w_last2_ci[w] == -1 ? rcss[0] : rcss[M.rc_cimap[M.w_rc_row[w] * M.n_ci + rc]];
                    if (v > bs) { bs = v; bpth = e; blc = M.w_last_ci[w]; }
                }
                F.brc_score[rc] = bs; F.brc_path[rc] = bpth; F.brc_lc[rc] = blc;
            }
            __syncthreads();
            const int32_t wthresh = add32(S.best_score, S.dyn_beam);
            /* into the roots (:1301-1317) */
            for (int32_t i = tid; i < M.n_root; i += NT) {
                const int32_t ci = M.root_ci[i];
                int32_t ns = add32(add32(F.brc_score[ci], M.nwpen), M.pip);
                if (PL) ns = add32(ns, F.pl[ci]);
                if (ns > wthresh && (CH_FR(L, i) < f || ns > CH_SC(L, i, 0))) {
                    CH_SC(L, i, 0) = ns; CH_HI(L, i, 0) = F.brc_path[ci]; CH_FR(L, i) = nf;
                    MPX_ID(L, 0, i) = M.root_lc_ssid[i * M.n_ci + F.brc_lc[ci]];
                    setbit(F.rootbits, i);
                }
            }
            /* single-phone words of the LM: the best predecessor with its trigram score (:1323-1349) */
            const int32_t n_pair = M.n_1ph_lm * n_ent;
            for (int32_t i = tid; i < M.n_1ph_lm; i += NT) F.key[i % NT] = 0ull;
            __syncthreads();
            for (int32_t q = tid; q < n_pair; q += NT) {
                const int32_t i = q / n_ent, e = bp0 + (q - i * n_ent);
                if (!L.bp_valid[e]) continue;
                const int32_t w = M.sp_wid[i];
                int32_t ns = exit_score(M, L, e, M.w_first_ci[w]);
                if (ns != PS_WORST) ns = add32(ns, lm_tg_score(M, M.w_basewid[w], L.bp_realwid[e], prev_real_wid(L, e)) >> 10);
                if (ns > INT_MIN)
                    atomicMax(&F.key[i], ((unsigned long long)((uint32_t)ns ^ 0x80000000u) << 32) | (uint32_t)(0x7fffffff - e));
            }
            __syncthreads();
            /* enter the single-phone channels: LM words first, then <sil> / the filler loop on top (:1351-1402) */
            for (int32_t i = tid; i < M.n_1ph; i += NT) {
                const int32_t c = M.sp_base + i, kind = M.sp_kind[i], w = M.sp_wid[i];
                if (i < M.n_1ph_lm) {
                    int32_t dscr = INT_MIN;
                    if (F.key[i] != 0ull) {
                        dscr = (int32_t)((uint32_t)(F.key[i] >> 32) ^ 0x80000000u);
                        L.lt_bp[w] = 0x7fffffff - (int32_t)(F.key[i] & 0xffffffffu);
                    }
                    L.lt_dscr[w] = dscr;
                    if (kind & 1) {
                        int32_t ns = add32(dscr, M.pip);
                        if (PL) ns = add32(ns, F.pl[M.sp_ci[i]]);
                        if (ns > wthresh && (CH_FR(L, c) < f || ns > CH_SC(L, c, 0))) {
                            const int32_t pb = L.lt_bp[w];
                            CH_SC(L, c, 0) = ns; CH_HI(L, c, 0) = pb; CH_FR(L, c) = nf;
                            MPX_ID(L, 0, M.n_root + i) = M.sp_lc_ssid[i * M.n_ci + M.w_last_ci[L.bp_wid[pb]]];
                        }
                    }
                }
                if (kind & 6) {
                    int32_t ns = add32(add32(F.brc_score[M.sil_ci], (kind & 2) ? M.silpen : M.fillpen), M.pip);
                    if (PL) ns = add32(ns, F.pl[M.sp_ci[i]]);
                    if (ns > wthresh && (CH_FR(L, c) < f || ns > CH_SC(L, c, 0))) { CH_SC(L, c, 0) = ns; CH_HI(L, c, 0) = F.brc_path[M.sil_ci]; CH_FR(L, c) = nf; }
                }
            }
        }
        __syncthreads();
    }

    TPHASE(S, 8);
    /* ---- deactivate_channels :1421-1443 ---- */
    for (int32_t j = tid; j < F.n_rl; j += NT) { const int32_t i = L.rl[j]; if (CH_FR(L, i) == f) hmm_clear_scores<NE>(M, L, i); }
    for (int32_t i = tid; i < M.n_1ph; i += NT) if (CH_FR(L, M.sp_base + i) == f) hmm_clear_scores<NE>(M, L, M.sp_base + i);
    if (tid == 0) S.n_frame++;
    __syncthreads();
    TPHASE(S, 9);
}

/* ------------------------------------------------------------------ */
/* the phone loop look-ahead (-pl_window; phone_loop_search.c).  Whole-utterance mode: the loop -- one plain HMM per CI    */
/* phone, all entered at the utterance's start, every exit entering every phone -- runs inside the lane's launch, pl_window */
/* frames ahead of the search (pocketsphinx.c:704-712: the loop is stepped for frame F, then the search for F - pl_window; */
/* :823-826: the last pl_window frames of the search see the loop's final state).  Its HMMs are the lane's channels        */
/* [pl_base, pl_base + n_ci); a thread per phone.  Histories are not kept apart from what hmm_vit_eval moves: nothing      */
/* reads the loop's backtrace (phone_loop_search_hyp returns nothing).                                                      */
/* ------------------------------------------------------------------ */
/* phone_loop_search_start :166-181 */
template <int NE> __device__ void
d_pl_start(const PsfModel &M, PsfLane &L, FrameShared &F)
{
    for (int32_t i = threadIdx.x; i < M.n_ci; i += NT) {
        const int32_t c = M.pl_base + i;
        hmm_clear<NE>(M, L, c);
        CH_SC(L, c, 0) = 0; CH_HI(L, c, 0) = -1; CH_FR(L, c) = 0;         /* hmm_enter(hmm, 0, -1, 0) */
    }
    if (threadIdx.x == 0) F.S.pl_best = 0;
    __syncthreads();
}

/* phone_loop_search_step :279-311 for frame fi, whose unnormalised scores are raw[].  F.senbits: acmod->senone_active_vec as the
 * search's last step left it (nothing clears it in between: acmod_score then lists -- and normalises over -- those senones AND
 * the loop's, :289-293; ms_mgau.c:219-246); the loop's senones are OR-ed in.  Leaves F.pl[ci] = phone_loop_search_score. */
template <int NE> __device__ void
d_pl_step(const PsfModel &M, PsfLane &L, FrameShared &F, int32_t fi, const int16_t *raw, int compallsen)
{
    const int tid = threadIdx.x, nf = fi + 1;
    if (!compallsen) {
        for (int32_t i = tid; i < M.n_ci; i += NT) activate<NE, false>(M, L, M.pl_base + i, 0, F.senbits);
        __syncthreads();
    }
    int32_t best = 0, count = 0;
    d_normaliser(M, F.wg, F.sh, F.senbits, raw, compallsen, &best, &count);
    const SenScr sen = { raw, best, 1 };
    /* renormalize_hmms :183-197: every phone, whatever its frame */
    const int32_t old_best = F.S.pl_best;
    if (add32(old_best, 2 * M.pl_beam) < PS_WORST)
        for (int32_t i = tid; i < M.n_ci; i += NT) {
            const int32_t c = M.pl_base + i;
#pragma unroll
            for (int k = 0; k < NE; k++) { const int32_t v = CH_SC(L, c, k); if (v > PS_WORST) CH_SC(L, c, k) = sub32(v, old_best); }
            const int32_t o = CH_OS(L, c);
            if (o > PS_WORST) CH_OS(L, c) = sub32(o, old_best);
        }
    /* evaluate_hmms :199-223 (a thread keeps its phones through the steps below: no barrier between them) */
    int32_t bs = PS_WORST, d1 = INT_MIN, z0 = 0, z1 = 0;
    for (int32_t i = tid; i < M.n_ci; i += NT) {
        const int32_t c = M.pl_base + i;
        if (CH_FR(L, c) < fi) continue;
        bs = max(bs, hmm_vit_eval<NE, false>(M, L, c, 0, sen));
    }
    wg_reduce4(F.wg, bs, d1, z0, z1);
    /* prune_hmms :225-244, and who leaves (phone_transition :246-277: out + pip inside the exit beam) */
    const int32_t thresh = add32(bs, M.pl_beam), xthresh = add32(bs, M.pl_pbeam);
    int32_t mx = INT_MIN, any = 0;
    d1 = INT_MIN; z1 = 0;
    for (int32_t i = tid; i < M.n_ci; i += NT) {
        const int32_t c = M.pl_base + i;
        if (CH_FR(L, c) < fi) continue;
        if (CH_BE(L, c) > thresh) {
            CH_FR(L, c) = nf;
            const int32_t ns = add32(CH_OS(L, c), M.pl_pip);
            if (ns > xthresh) { mx = max(mx, ns); any = 1; }
        }
        else hmm_clear_scores<NE>(M, L, c);
    }
    wg_reduce4(F.wg, mx, d1, any, z1);
    if (add32(PS_WORST, M.pl_pip) > xthresh) {
        /* (a phone that was pruned holds WORST_SCORE as its exit score; entered by an earlier phone of this very loop it is looked
         * at again, :258, and with an exit threshold below WORST_SCORE + pip it would leave too: the reference's loop as it stands) */
        if (tid == 0)
            for (int32_t i = 0; i < M.n_ci; i++) {
                const int32_t c = M.pl_base + i;
                if (CH_FR(L, c) != nf) continue;
                const int32_t ns = add32(CH_OS(L, c), M.pl_pip);
                if (!(ns > xthresh)) continue;
                for (int32_t j = 0; j < M.n_ci; j++) {
                    const int32_t x = M.pl_base + j;
                    if (CH_FR(L, x) < fi || ns > CH_SC(L, x, 0)) { CH_SC(L, x, 0) = ns; CH_FR(L, x) = nf; }
                }
            }
    }
    else if (any) {
        /* every phone is entered with the best of the leaving scores: unconditionally by the first leaving phone when it was not
         * active in this frame, by a strictly better score otherwise (:268-271) -- the maximum either way */
        for (int32_t i = tid; i < M.n_ci; i += NT) {
            const int32_t x = M.pl_base + i;
            if (CH_FR(L, x) < fi || mx > CH_SC(L, x, 0)) { CH_SC(L, x, 0) = mx; CH_FR(L, x) = nf; }
        }
    }
    __syncthreads();
    for (int32_t i = tid; i < M.n_ci; i += NT) F.pl[i] = sub32(CH_BE(L, M.pl_base + i), bs);
    if (tid == 0) F.S.pl_best = bs;
    __syncthreads();
}

/* ngram_fwdtree_start :464-507; fresh: what a new decoder's channels look like (init_search_tree :66-148) */
template <int NE> __device__ void
d_start(const PsfModel &M, PsfLane &L, PsfScalars &S, int fresh)
{
    const int tid = threadIdx.x;
    /* after an utterance that was finished (ngram_fwdtree_finish cleared the roots, the last active lists and every
     * word's last-phone channels; pruning cleared the scores of everything else) the state differs from a new decoder's
     * only in the interior channels' stale frame numbers and histories, the multiplexed ids, the last-transition cache and
     * the real-word ids of the old table: reset those, not 20+ MB of channel records */
    const bool light = fresh && S.clean, full = fresh && !light;
    const int32_t old_bpidx = S.bpidx;
    __syncthreads();
    if (light) {
        for (int32_t c = M.n_root + tid; c < M.n_ch; c += NT) hmm_clear<NE>(M, L, c);
        for (int32_t i = tid; i < M.n_mpx; i += NT) {
            MPX_ID(L, 0, i) = i < M.n_root ? M.root_ssid0[i] : M.sp_ssid0[i - M.n_root];
            for (int k = 1; k < NE; k++) MPX_ID(L, k, i) = PS_BAD_SSID;
        }
        for (int32_t w = tid; w < M.n_words; w += NT) { L.lt_dscr[w] = 0; L.lt_bp[w] = 0; }
        for (int32_t i = tid; i < old_bpidx && i < M.bp_cap; i += NT) L.bp_realwid[i] = 0;
    }
    if (full) {
        for (int32_t c = tid; c < M.n_hmm; c += NT) hmm_clear<NE>(M, L, c);
        for (int32_t i = tid; i < M.n_mpx; i += NT) {
            MPX_ID(L, 0, i) = i < M.n_root ? M.root_ssid0[i] : M.sp_ssid0[i - M.n_root];
            for (int k = 1; k < NE; k++) MPX_ID(L, k, i) = PS_BAD_SSID;
        }
        for (int32_t w = tid; w < M.n_words; w += NT) { L.lt_dscr[w] = 0; L.lt_bp[w] = 0; L.wstamp[w] = 0; }
        for (int32_t i = tid; i < M.n_nonroot; i += NT) { CH_AP(L, 0, M.n_root + i) = -1; CH_AP(L, 1, M.n_root + i) = -1; ent_stamp_ref<NE>(L, M.n_root + i, M.n_root) = 0; }
        for (int32_t i = tid; i < M.bp_cap; i += NT) L.bp_realwid[i] = 0;
    }
    for (int32_t w = tid; w < M.n_words; w += NT) L.lt_sf[w] = -1;
    for (int32_t i = tid; i < M.n_1ph; i += NT) hmm_clear<NE>(M, L, M.sp_base + i);
    __syncthreads();
    if (tid == 0) {
        const int32_t keep_tick = S.tick, n_total = S.n_total;
        memset(&S, 0, sizeof(S));           /* (clean = 0 until this utterance is finished) */
        S.tick = keep_tick + 1; S.n_total = n_total;
        S.exit_bp = NO_BP;
        const int32_t c = M.sp_base + M.start_sp;
        CH_SC(L, c, 0) = 0; CH_HI(L, c, 0) = NO_BP; CH_FR(L, c) = 0;                 /* hmm_enter(<s>, 0, NO_BP, 0) */
    }
    __syncthreads();
}

/* ngram_fwdtree_finish :1490-1551 */
template <int NE> __device__ void
d_finish(const PsfModel &M, PsfLane &L, PsfScalars &S, int32_t cf)
{
    const int tid = threadIdx.x;
    if (tid == 0) L.bp_idx[1 + cf] = S.bpidx;
    for (int32_t i = tid; i < M.n_root; i += NT) hmm_clear<NE>(M, L, i);
    for (int32_t j = tid; j < S.n_acl[cf & 1]; j += NT) hmm_clear<NE>(M, L, L.acl[cf & 1][j]);
    for (int32_t j = tid; j < S.n_awl[cf & 1]; j += NT) {
        const int32_t w = L.awl[cf & 1][j], c0 = M.rc_base + M.w_rc_base[w];
        for (int32_t r = 0; r < M.w_rcsize[w]; r++) hmm_clear<NE>(M, L, c0 + r);     /* ngram_search_free_all_rc */
    }
    __syncthreads();
    if (tid == 0) { S.finished = 1; S.clean = (S.status == 0 && S.best_score > PS_WORST) ? 1 : 0; }
}

/* ngram_search_find_exit (ngram_search.c:444-484) + ngram_search_bp_iter / _bp2itor (:862-903, :777-818), lwf = 1 */
__device__ void
d_hyp(const PsfModel &M, PsfLane &L, PsfScalars &S)
{
    if (threadIdx.x != 0) return;
    S.exit_bp = NO_BP; S.n_seg = 0; S.exit_score = PS_WORST;
    if (S.n_frame == 0) return;
    int32_t fi = S.n_frame - 1;
    const int32_t end = L.bp_idx[1 + fi];
    while (fi >= 0 && L.bp_idx[1 + fi] == end) --fi;
    if (fi < 0) return;
    int32_t best = PS_WORST, ex = NO_BP;
    for (int32_t bp = L.bp_idx[1 + fi]; bp < end; ++bp) {
        const bool fin = L.bp_wid[bp] == M.finish_wid;
        if (fin || L.bp_score[bp] > best) { best = L.bp_score[bp]; ex = bp; }
        if (fin) break;
    }
    S.exit_bp = ex; S.exit_score = best;
    int32_t n = 0;
    for (int32_t bp = ex; bp != NO_BP; bp = L.bp_bp[bp]) n++;
    S.n_seg = n;
    if (n > MAX_SEG) return;
    int32_t cur = n - 1;
    for (int32_t bp = ex; bp != NO_BP; bp = L.bp_bp[bp], cur--) {
        const int32_t pbe = L.bp_bp[bp], w = L.bp_wid[bp];
        s3a_psfwd_seg_t sg;
        sg.wid = w; sg.ef = L.bp_frame[bp]; sg.sf = pbe == NO_BP ? 0 : L.bp_frame[pbe] + 1; sg.bp = bp;
        if (pbe == NO_BP) { sg.ascr = L.bp_score[bp]; sg.lscr = 0; }
        else {
            const int32_t start_score = exit_score(M, L, pbe, M.w_first_ci[w]);
            if (w == M.silence_wid) sg.lscr = M.silpen;
            else if (M.w_flags[w] & S3A_PSW_FILLER) sg.lscr = M.fillpen;
            else sg.lscr = (int32_t)((float)(lm_tg_score(M, L.bp_realwid[bp], L.bp_realwid[pbe], prev_real_wid(L, pbe)) >> 10) * 1.0f);
            sg.ascr = sub32(sub32(L.bp_score[bp], start_score), sg.lscr);
        }
        L.seg[cur] = sg;
    }
}

/* ------------------------------------------------------------------ */
/* kernels: one workgroup per lane                                    */
/* ------------------------------------------------------------------ */
template <int NE> __global__ void __launch_bounds__(NT)
k_psf_start(PsfModel M, PsfLane *lanes, const int32_t *lane_ids, int fresh)
{
    PsfLane L = lanes[lane_ids[blockIdx.x]];
    __shared__ PsfScalars S;
    if (threadIdx.x == 0) S = *L.sc;
    __syncthreads();
    d_start<NE>(M, L, S, fresh);
    if (threadIdx.x == 0) *L.sc = S;
}

template <int NE> __global__ void __launch_bounds__(NT)
k_psf_sen_active(PsfModel M, PsfLane *lanes, int32_t lane, int32_t f)
{
    PsfLane L = lanes[lane];
    __shared__ PsfScalars S;
    __shared__ Wg wg;
    __shared__ uint32_t senbits[SENBITS_WORDS], rootbits[ROOTBITS_WORDS];
    if (threadIdx.x == 0) S = *L.sc;
    __syncthreads();
    d_root_bits_from_frames(M, L, rootbits, f);
    const int32_t n_rl = d_root_list(M, L, wg, rootbits);
    d_sen_active<NE>(M, L, S, f, senbits, n_rl, -1);
    for (int32_t s = threadIdx.x; s < M.n_sen; s += NT) L.flags[s] = (senbits[s >> 5] >> (s & 31)) & 1;
}

/* frame-synchronous: one frame of one lane with the caller's (normalised) senone scores in L.senscr */
template <int NE, bool PL> __global__ void __launch_bounds__(NT)
k_psf_step(PsfModel M, PsfLane *lanes, int32_t lane, int32_t f, int32_t n_senone_active)
{
    PsfLane L = lanes[lane];
    __shared__ FrameShared F;
    if (threadIdx.x == 0) F.S = *L.sc;
    if (PL) for (int32_t i = threadIdx.x; i < M.n_ci; i += NT) F.pl[i] = L.pl_host[i];      /* (the decoder's own phone loop: s3a_psfwd_set_lookahead) */
    __syncthreads();
    d_root_bits_from_frames(M, L, F.rootbits, f);
    d_frame_roots(M, L, F);
    d_frame_word_chans(M, L, F, f);
    SenScr sen = { L.senscr, 0, 0 };
    d_frame<NE, PL>(M, L, F, f, sen, n_senone_active);
    __syncthreads();
    if (threadIdx.x == 0) *L.sc = F.S;
}

template <int NE> __global__ void __launch_bounds__(NT)
k_psf_finish(PsfModel M, PsfLane *lanes, int32_t lane, int32_t cf)
{
    PsfLane L = lanes[lane];
    __shared__ PsfScalars S;
    if (threadIdx.x == 0) S = *L.sc;
    __syncthreads();
    d_finish<NE>(M, L, S, cf);
    __syncthreads();
    d_hyp(M, L, S);
    __syncthreads();
    if (threadIdx.x == 0) *L.sc = S;
}

/* whole utterances: frames [f0, f0 + n_win) of every lane of the batch; a lane's utterance may end inside */
/* (PL: the window is the whole utterance, f0 = 0 -- the phone loop reads rows ahead of the search's; pl_fresh: the lane is a new
 * decoder, whose acmod->senone_active_vec is empty) */
template <int NE, bool PL> __global__ void __launch_bounds__(NT, PSF_WPE)
k_psf_window(PsfModel M, PsfLane *lanes, const int32_t *lane_ids, int32_t f0, int32_t n_win, int compallsen, int pl_fresh)
{
    PsfLane L = lanes[lane_ids[blockIdx.x]];
    __shared__ FrameShared F;
    if (threadIdx.x == 0) F.S = *L.sc;
    __syncthreads();
    const int32_t n_total = F.S.n_total;
    if (F.S.finished) return;
    d_root_bits_from_frames(M, L, F.rootbits, f0);
    if (PL) {
        for (int32_t i = threadIdx.x; i < ((M.n_sen + 31) >> 5); i += NT) F.senbits[i] = pl_fresh ? 0u : L.senkeep[i];
        d_pl_start<NE>(M, L, F);
        for (int32_t fi = 0; fi < M.pl_window && fi < n_total; fi++) d_pl_step<NE>(M, L, F, fi, L.raw + (size_t)fi * M.n_sen, compallsen);
    }
    for (int32_t f = f0; f < f0 + n_win && f < n_total; f++) {
        const int16_t *raw = L.raw + (size_t)(f - f0) * M.n_sen;
        int32_t best = 0, count = 0;
#ifdef PSF_TIMING
        if (threadIdx.x == 0) F.S.t_last = wall_clock64();
#endif
        if (PL && f + M.pl_window < n_total) d_pl_step<NE>(M, L, F, f + M.pl_window, L.raw + (size_t)(f + M.pl_window) * M.n_sen, compallsen);
        d_frame_roots(M, L, F);
        d_frame_word_chans(M, L, F, f);
        if (!compallsen) d_sen_active<NE>(M, L, F.S, f, F.senbits, F.n_rl, F.n_arc);
        TPHASE(F.S, 0);
        d_normaliser(M, F.wg, F.sh, F.senbits, raw, compallsen, &best, &count);
        TPHASE(F.S, 1);
        SenScr sen = { raw, best, 1 };
        d_frame<NE, PL>(M, L, F, f, sen, count);
        __syncthreads();
    }
    if (PL && !compallsen) for (int32_t i = threadIdx.x; i < ((M.n_sen + 31) >> 5); i += NT) L.senkeep[i] = F.senbits[i];
    if (f0 + n_win >= n_total) {
        d_finish<NE>(M, L, F.S, n_total);
        __syncthreads();
        d_hyp(M, L, F.S);
        __syncthreads();
    }
    if (threadIdx.x == 0) *L.sc = F.S;
}

/* whole utterances from a QUEUE: the lanes are persistent workgroups, a lane takes the next utterance of the batch when its
 * own has ended (ps_decode_raw's loop over a control file, batch.c:720-760, one decoder per lane; every utterance starts
 * from a new decoder's state).  Hypotheses are made on the device and kept per utterance. */
#define PSF_QRES 8
struct PsfQueue {
    int32_t n_utt, seg_cap;
    int32_t *next;                  /* the queue's head */
    const int32_t *order;           /* the k-th utterance the lanes take (longest first: the queue's tail is short ones); NULL: k */
    const int32_t *ready;           /* utterances whose scores are complete (the scoring runs beside the search); NULL: all */
    const int32_t *nfr;             /* [n_utt] */
    const long long *row0;          /* [n_utt] first row of the utterance in the score matrix */
    const int16_t *raw;             /* [rows][n_sen] */
    int32_t *res;                   /* [n_utt][PSF_QRES]: status, exit score, #segments, frames searched, #backpointers, lane */
    s3a_psfwd_seg_t *seg;           /* [n_utt][seg_cap] */
};

template <int NE, bool PL> __global__ void __launch_bounds__(NT, PSF_WPE)
k_psf_queue(PsfModel M, PsfLane *lanes, PsfQueue Q, int compallsen)
{
    PsfLane L = lanes[blockIdx.x];
    __shared__ FrameShared F;
    __shared__ int32_t s_u;
    if (Q.ready) __builtin_amdgcn_s_setprio(3);     /* the scoring shares the CU: this lane's few waves go first at instruction issue */
    if (threadIdx.x == 0) F.S = *L.sc;
    __syncthreads();
    for (;;) {
        if (threadIdx.x == 0) s_u = atomicAdd(Q.next, 1);
        __syncthreads();
        const int32_t k_ = s_u;
        __syncthreads();
        if (k_ >= Q.n_utt) break;
        const int32_t u = Q.order ? Q.order[k_] : k_;
        if (Q.ready) {
            /* the utterance's scores come from a kernel on another stream: wait for its completion mark (a store behind that
             * kernel, so everything it wrote is in memory), then drop what this CU may hold of those lines */
            if (threadIdx.x == 0)
                while (__hip_atomic_load(Q.ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) <= k_) __builtin_amdgcn_s_sleep(32);
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        const int32_t n_total = Q.nfr[u];
        const int16_t *raw0 = Q.raw + (size_t)Q.row0[u] * M.n_sen;
        if (threadIdx.x == 0) F.S.n_total = n_total;
        __syncthreads();
        d_start<NE>(M, L, F.S, 1);
        d_root_bits_from_frames(M, L, F.rootbits, 0);
        if (PL) {           /* (every utterance of a queue starts from a new decoder: no senone is flagged yet) */
            for (int32_t i = threadIdx.x; i < ((M.n_sen + 31) >> 5); i += NT) F.senbits[i] = 0u;
            d_pl_start<NE>(M, L, F);
            for (int32_t fi = 0; fi < M.pl_window && fi < n_total; fi++) d_pl_step<NE>(M, L, F, fi, raw0 + (size_t)fi * M.n_sen, compallsen);
        }
        for (int32_t f = 0; f < n_total; f++) {
            const int16_t *raw = raw0 + (size_t)f * M.n_sen;
            int32_t best = 0, count = 0;
            if (PL && f + M.pl_window < n_total) d_pl_step<NE>(M, L, F, f + M.pl_window, raw0 + (size_t)(f + M.pl_window) * M.n_sen, compallsen);
            d_frame_roots(M, L, F);
            d_frame_word_chans(M, L, F, f);
            if (!compallsen) d_sen_active<NE>(M, L, F.S, f, F.senbits, F.n_rl, F.n_arc);
            d_normaliser(M, F.wg, F.sh, F.senbits, raw, compallsen, &best, &count);
            SenScr sen = { raw, best, 1 };
            d_frame<NE, PL>(M, L, F, f, sen, count);
            __syncthreads();
        }
        d_finish<NE>(M, L, F.S, n_total);
        __syncthreads();
        d_hyp(M, L, F.S);
        __syncthreads();
        if (threadIdx.x == 0) {
            int32_t *r = Q.res + (size_t)u * PSF_QRES;
            r[0] = F.S.status; r[1] = F.S.exit_score; r[2] = F.S.exit_bp == NO_BP ? 0 : F.S.n_seg; r[3] = F.S.n_frame; r[4] = F.S.bpidx; r[5] = blockIdx.x;
        }
        const int32_t ns = F.S.exit_bp == NO_BP ? 0 : (F.S.n_seg < Q.seg_cap ? F.S.n_seg : Q.seg_cap);
        for (int32_t i = threadIdx.x; i < ns && i < MAX_SEG; i += NT) Q.seg[(size_t)u * Q.seg_cap + i] = L.seg[i];
        __syncthreads();
    }
    if (threadIdx.x == 0) *L.sc = F.S;
}

__global__ void
k_psf_mark_ready(int32_t *ready, int32_t n)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(ready, n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

/* ------------------------------------------------------------------ */
/* host side                                                          */
/* ------------------------------------------------------------------ */
struct s3a_psfwd_s {
    PsfModel M;
    int32_t n_lanes, n_sseq, n_tmat;
    std::vector<void *> dev_static;         /* model arrays on the device */
    std::vector<void *> dev_lane;           /* every lane's buffers */
    std::vector<PsfLane> lanes_h;
    PsfLane *lanes_d;
    int32_t *lane_ids_d;
    hipStream_t stream, stream_sc;          /* the search's stream; the scoring's when it runs beside the search (queue) */
    hipEvent_t ev0, ev1, ev_sc;
    int32_t *q_ready_d;
    double last_ms, last_score_ms;          /* the last decode: whole region; its scoring launches (0 when they ran beside the search) */
    hipEvent_t ev_mid;
    /* host mirrors for s3a_psfwd_table */
    std::vector<int32_t> t_frame, t_wid, t_bp, t_score, t_sidx, t_realwid, t_bss, t_idx;
    std::vector<uint8_t> t_valid;
    /* whole-utterance staging */
    float *feat_d; size_t feat_cap;
    int32_t *slot_row_d; size_t slot_cap;
    int16_t *raw_d; size_t raw_cap;
    int32_t win;
    /* the queue's per-utterance results */
    int32_t *q_next_d, *q_nfr_d, *q_res_d, *q_order_d; long long *q_row0_d; s3a_psfwd_seg_t *q_seg_d;
    size_t q_cap;
    int32_t q_n, q_seg_cap;
    std::vector<int32_t> q_res_h;
    std::vector<s3a_psfwd_seg_t> q_seg_h;
};

template <typename T> static T *
up(s3a_psfwd_t *e, const T *src, size_t n, bool lane = false)
{
    T *d = NULL;
    if (hipMalloc((void **)&d, (n ? n : 1) * sizeof(T)) != hipSuccess) return NULL;
    if (src && n) { if (hipMemcpy(d, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return NULL; }
    else if (hipMemset(d, 0, (n ? n : 1) * sizeof(T)) != hipSuccess) return NULL;
    (lane ? e->dev_lane : e->dev_static).push_back(d);
    return d;
}

#define UP(dst, src, n) do { (dst) = up(e, (src), (size_t)(n)); if (!(dst)) { s3a_set_error("s3a_psfwd_init: device allocation failed"); s3a_psfwd_free(e); return NULL; } } while (0)

extern "C" void
s3a_psfwd_free(s3a_psfwd_t *e)
{
    if (!e) return;
    for (void *p : e->dev_static) (void)hipFree(p);
    for (void *p : e->dev_lane) (void)hipFree(p);
    if (e->lanes_d) (void)hipFree(e->lanes_d);
    if (e->lane_ids_d) (void)hipFree(e->lane_ids_d);
    if (e->feat_d) (void)hipFree(e->feat_d);
    if (e->slot_row_d) (void)hipFree(e->slot_row_d);
    if (e->raw_d) (void)hipFree(e->raw_d);
    { void *qp[] = { e->q_next_d, e->q_nfr_d, e->q_res_d, e->q_row0_d, e->q_seg_d, e->q_order_d }; for (void *q : qp) if (q) (void)hipFree(q); }
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->ev_mid) (void)hipEventDestroy(e->ev_mid);
    if (e->ev_sc) (void)hipEventDestroy(e->ev_sc);
    if (e->stream_sc) (void)hipStreamDestroy(e->stream_sc);
    if (e->q_ready_d) (void)hipFree(e->q_ready_d);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

extern "C" s3a_psfwd_t *
s3a_psfwd_init(const s3a_psfwd_desc_t *d, int32_t n_lanes, int32_t max_frames, int32_t bp_cap, int32_t bss_cap)
{
    int ndev = 0;
    if (!d || n_lanes < 1 || max_frames < 1) { s3a_set_error("s3a_psfwd_init: bad argument"); return NULL; }
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        s3a_set_error("no HIP device: libcmusphinx_amd has no CPU fallback");
        return NULL;
    }
    if (d->n_emit != 3 && d->n_emit != 5) { s3a_set_error("s3a_psfwd_init: %d emitting states (3 or 5 served)", d->n_emit); return NULL; }
    if (d->n_ci > 256) { s3a_set_error("s3a_psfwd_init: %d CI phones exceed the kernel's 256", d->n_ci); return NULL; }
    if (d->n_root > 32 * ROOTBITS_WORDS || d->n_sen > 32 * SENBITS_WORDS) { s3a_set_error("s3a_psfwd_init: %d roots / %d senones exceed the kernel's bit vectors", d->n_root, d->n_sen); return NULL; }
    if (d->n_1ph > NT) { s3a_set_error("s3a_psfwd_init: %d single-phone words exceed the kernel's %d", d->n_1ph, NT); return NULL; }
    if (max_frames > 32767) { s3a_set_error("s3a_psfwd_init: max_frames %d (the reference's frame numbers are int16)", max_frames); return NULL; }
    s3a_psfwd_t *e = new s3a_psfwd_t();
    e->lanes_d = NULL; e->lane_ids_d = NULL; e->stream = NULL; e->stream_sc = NULL; e->ev0 = e->ev1 = e->ev_sc = NULL; e->q_ready_d = NULL; e->last_ms = 0; e->last_score_ms = 0; e->ev_mid = NULL;
    e->feat_d = NULL; e->feat_cap = 0; e->slot_row_d = NULL; e->slot_cap = 0; e->raw_d = NULL; e->raw_cap = 0; e->win = 0;
    e->q_next_d = e->q_nfr_d = e->q_res_d = e->q_order_d = NULL; e->q_row0_d = NULL; e->q_seg_d = NULL; e->q_cap = 0; e->q_n = 0; e->q_seg_cap = 256;
    PsfModel &M = e->M;
    memset(&M, 0, sizeof(M));
    e->n_lanes = n_lanes;
    const int32_t NE = d->n_emit, W = d->n_words, n_ch = d->n_root + d->n_nonroot;
    M.n_ci = d->n_ci; M.sil_ci = d->sil_ci; M.n_emit = NE; M.n_sen = d->n_sen;
    M.n_words = W; M.start_wid = d->start_wid; M.finish_wid = d->finish_wid; M.silence_wid = d->silence_wid;
    M.n_root = d->n_root; M.n_nonroot = d->n_nonroot; M.n_ch = n_ch; M.sp_base = n_ch; M.rc_base = n_ch + d->n_1ph;
    M.n_mpx = d->n_root + d->n_1ph; M.n_1ph = d->n_1ph; M.n_1ph_lm = d->n_1ph_lm;
    M.lm_order = d->lm_order; M.lm_zero = d->lm_zero;
    M.beam = d->beam; M.pbeam = d->pbeam; M.wbeam = d->wbeam; M.lpbeam = d->lpbeam; M.lponlybeam = d->lponlybeam;
    M.fillpen = d->fillpen; M.silpen = d->silpen; M.nwpen = d->nwpen; M.pip = d->pip; M.maxwpf = d->maxwpf; M.maxhmmpf = d->maxhmmpf;
    M.max_frames = max_frames;
    M.bp_cap = bp_cap > 0 ? bp_cap : 64 * max_frames;
    M.bss_cap = bss_cap > 0 ? bss_cap : M.bp_cap * (d->n_ci < 24 ? d->n_ci : 24);

    /* the words that can leave the tree get last-phone channels, one per right context */
    std::vector<int32_t> rcsize(W), rcbase(W, -1), par(n_ch, -1);
    std::vector<uint16_t> ch_ssid;
    std::vector<int16_t> ch_tmat;
    for (int32_t w = 0; w < W; w++) rcsize[w] = (d->w_flags[w] & S3A_PSW_SINGLE) ? 1 : d->w_rc_off[w + 1] - d->w_rc_off[w];
    ch_ssid.assign(M.rc_base, 0); ch_tmat.assign(M.rc_base, 0);
    for (int32_t i = 0; i < d->n_root; i++) ch_tmat[i] = d->root_tmat[i];
    for (int32_t i = 0; i < d->n_nonroot; i++) { ch_ssid[d->n_root + i] = d->nr_ssid[i]; ch_tmat[d->n_root + i] = d->nr_tmat[i]; }
    for (int32_t i = 0; i < d->n_1ph; i++) ch_tmat[M.sp_base + i] = d->sp_tmat[i];
    int32_t n_rc = 0, n_pen = d->ch_pen_off[n_ch];
    for (int32_t c = 0; c < n_ch; c++) {
        for (int32_t x = d->ch_child_off[c]; x < d->ch_child_off[c + 1]; x++) par[d->ch_child[x]] = c;
        for (int32_t x = d->ch_pen_off[c]; x < d->ch_pen_off[c + 1]; x++) {
            const int32_t w = d->ch_pen_wid[x];
            if (w < 0 || w >= W || (d->w_flags[w] & S3A_PSW_SINGLE) || rcbase[w] >= 0) {
                s3a_set_error("s3a_psfwd_init: word %d is in two penultimate lists or is a single-phone word", w);
                delete e;
                return NULL;
            }
            rcbase[w] = n_rc;
            for (int32_t r = 0; r < rcsize[w]; r++) { ch_ssid.push_back(d->rc_ssid[d->w_rc_off[w] + r]); ch_tmat.push_back(d->w_rc_tmat[w]); }
            n_rc += rcsize[w];
        }
    }
    M.n_hmm = M.rc_base + n_rc;
    M.cand_cap = n_pen > 0 ? n_pen : 1;
    /* the phone loop's HMMs behind the search's channels */
    M.pl_window = d->pl_window; M.pl_beam = d->pl_beam; M.pl_pbeam = d->pl_pbeam; M.pl_pip = d->pl_pip; M.pl_base = M.n_hmm;
    if (d->pl_window < 0 || (d->pl_window > 0 && (!d->ci_ssid || !d->ci_tmat))) {
        s3a_set_error("s3a_psfwd_init: pl_window %d needs ci_ssid / ci_tmat", d->pl_window);
        delete e;
        return NULL;
    }
    std::vector<int16_t> ch_ci(n_ch > 0 ? n_ch : 1, 0);
    for (int32_t i = 0; i < d->n_root; i++) ch_ci[i] = d->root_ci[i];
    for (int32_t i = 0; i < d->n_nonroot; i++) ch_ci[d->n_root + i] = d->nr_ci[i];
    if (d->pl_window > 0) {
        for (int32_t i = 0; i < d->n_ci; i++) {
            if (d->ci_ssid[i] >= d->n_sseq || d->ci_tmat[i] < 0 || d->ci_tmat[i] >= d->n_tmat) {
                s3a_set_error("s3a_psfwd_init: CI phone %d: senone sequence %d / transition matrix %d", i, d->ci_ssid[i], d->ci_tmat[i]);
                delete e;
                return NULL;
            }
            ch_ssid.push_back(d->ci_ssid[i]); ch_tmat.push_back(d->ci_tmat[i]);
        }
        for (int32_t c = 0; c < n_ch; c++)
            if (ch_ci[c] < 0 || ch_ci[c] >= d->n_ci) { s3a_set_error("s3a_psfwd_init: channel %d: CI phone %d of %d", c, ch_ci[c], d->n_ci); delete e; return NULL; }
        for (int32_t i = 0; i < d->n_1ph; i++)
            if (d->sp_ci[i] < 0 || d->sp_ci[i] >= d->n_ci) { s3a_set_error("s3a_psfwd_init: single-phone word %d: CI phone %d of %d", i, d->sp_ci[i], d->n_ci); delete e; return NULL; }
    }
    std::vector<uint8_t> sp_kind(d->n_1ph > 0 ? d->n_1ph : 1, 0);
    M.sil_sp = -1; M.start_sp = -1;
    for (int32_t i = 0; i < d->n_1ph; i++) {
        if (i < d->n_1ph_lm && d->sp_wid[i] != d->start_wid) sp_kind[i] |= 1;
        if (d->sp_wid[i] == d->silence_wid) { sp_kind[i] |= 2; M.sil_sp = i; }
        if (d->sp_wid[i] == d->start_wid) M.start_sp = i;
    }
    for (int32_t i = 0; i < d->n_fill; i++) sp_kind[d->fill_sp[i]] |= 4;
    if (M.sil_sp < 0 || M.start_sp < 0) { s3a_set_error("s3a_psfwd_init: <sil> or <s> is not a listed single-phone word"); delete e; return NULL; }

    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&e->ev0) != hipSuccess
        || hipStreamCreateWithFlags(&e->stream_sc, hipStreamNonBlocking) != hipSuccess
        || hipEventCreateWithFlags(&e->ev_sc, hipEventDisableTiming) != hipSuccess || hipMalloc((void **)&e->q_ready_d, 4) != hipSuccess
        || hipEventCreate(&e->ev1) != hipSuccess || hipEventCreate(&e->ev_mid) != hipSuccess) { s3a_set_error("s3a_psfwd_init: stream / event creation failed"); s3a_psfwd_free(e); return NULL; }
    UP(M.sseq, d->sseq, (size_t)d->n_sseq * NE);
    UP(M.tp, d->tp, (size_t)d->n_tmat * NE * (NE + 1));
    UP(M.w_basewid, d->w_basewid, W); UP(M.w_lmwid, d->w_lmwid, W);
    if (d->w_lmcw) UP(M.w_lmcw, d->w_lmcw, W);
    UP(M.w_rcsize, rcsize.data(), W); UP(M.w_rc_base, rcbase.data(), W); UP(M.w_rc_row, d->w_rc_row, W);
    UP(M.w_first_ci, d->w_first_ci, W); UP(M.w_last_ci, d->w_last_ci, W); UP(M.w_last2_ci, d->w_last2_ci, W);
    UP(M.rc_cimap, d->rc_cimap, (size_t)d->n_rc_rows * d->n_ci);
    UP(M.w_flags, d->w_flags, W);
    UP(M.ch_ssid, ch_ssid.data(), ch_ssid.size()); UP(M.ch_tmat, ch_tmat.data(), ch_tmat.size());
    UP(M.ch_par, par.data(), n_ch);
    UP(M.ch_ci, ch_ci.data(), ch_ci.size()); UP(M.sp_ci, d->sp_ci, d->n_1ph);
    UP(M.root_ci, d->root_ci, d->n_root);
    UP(M.root_lc_ssid, d->root_lc_ssid, (size_t)d->n_root * d->n_ci); UP(M.sp_lc_ssid, d->sp_lc_ssid, (size_t)d->n_1ph * d->n_ci);
    UP(M.root_ssid0, d->root_ssid0, d->n_root); UP(M.sp_ssid0, d->sp_ssid0, d->n_1ph);
    UP(M.ch_child_off, d->ch_child_off, n_ch + 1); UP(M.ch_child, d->ch_child, d->ch_child_off[n_ch]);
    UP(M.ch_pen_off, d->ch_pen_off, n_ch + 1); UP(M.ch_pen_wid, d->ch_pen_wid, n_pen);
    UP(M.sp_wid, d->sp_wid, d->n_1ph); UP(M.sp_kind, sp_kind.data(), sp_kind.size());
    UP(M.ug_prob, d->ug_prob, d->lm_n_ug); UP(M.ug_bowt, d->ug_bowt, d->lm_n_ug); UP(M.ug_firstbg, d->ug_firstbg, d->lm_n_ug + 1);
    UP(M.bg_wid, d->bg_wid, d->lm_n_bg); UP(M.bg_prob, d->bg_prob, d->lm_n_bg); UP(M.bg_bowt, d->bg_bowt, d->lm_n_bg);
    UP(M.bg_firsttg, d->bg_firsttg, d->lm_n_bg + 1);
    UP(M.tg_wid, d->tg_wid, d->lm_n_tg); UP(M.tg_prob, d->tg_prob, d->lm_n_tg);

    e->lanes_h.resize(n_lanes);
    {
        /* a lane's arrays as ONE allocation (256-byte aligned pieces, zeroed once): 36 allocations per lane were 18 000 for an engine of
         * 512 lanes -- slow to make, and more than the profiler's allocation tracking takes (rocprofv3 --pmc did not get through such an
         * engine's construction: profiles/r6_experiments.txt 10, 21) */
        auto layout = [&](PsfLane &L, char *base) -> size_t {
            size_t at = 0;
            auto take = [&](size_t n, size_t sz) -> char * { char *p_ = base ? base + at : (char *)NULL; at += ((n ? n : 1) * sz + 255) & ~(size_t)255; return p_; };
#define PIECE(dst, T, n) (dst) = (T *)take((size_t)(n), sizeof(T))
            const size_t H = (size_t)M.n_hmm + (M.pl_window > 0 ? M.n_ci : 0);
            PIECE(L.st, int32_t, CH_STRIDE * H);
            PIECE(L.pl_host, int32_t, M.n_ci); PIECE(L.senkeep, uint32_t, (M.n_sen + 31) / 32);
            PIECE(L.mpxid, uint16_t, (size_t)MPX_STRIDE * M.n_mpx);
            for (int k = 0; k < 2; k++) { PIECE(L.acl[k], int32_t, d->n_nonroot + 1); PIECE(L.awl[k], int32_t, M.cand_cap + 1); }
            PIECE(L.wstamp, int32_t, W); PIECE(L.lt_sf, int32_t, W); PIECE(L.lt_dscr, int32_t, W); PIECE(L.lt_bp, int32_t, W);
            PIECE(L.ent_score, int32_t, d->n_nonroot + 1); PIECE(L.ent_hist, int32_t, d->n_nonroot + 1); PIECE(L.ent_stamp, int32_t, d->n_nonroot + 1);
            PIECE(L.cand_wid, int32_t, M.cand_cap); PIECE(L.cand_score, int32_t, M.cand_cap); PIECE(L.cand_bp, int32_t, M.cand_cap); PIECE(L.cand_ef, int32_t, M.cand_cap);
            PIECE(L.bp_frame, int32_t, M.bp_cap); PIECE(L.bp_wid, int32_t, M.bp_cap); PIECE(L.bp_bp, int32_t, M.bp_cap); PIECE(L.bp_score, int32_t, M.bp_cap);
            PIECE(L.bp_sidx, int32_t, M.bp_cap); PIECE(L.bp_realwid, int32_t, M.bp_cap); PIECE(L.bp_valid, uint8_t, M.bp_cap);
            PIECE(L.bss, int32_t, M.bss_cap); PIECE(L.bp_idx, int32_t, max_frames + 3);
            PIECE(L.flags, uint8_t, d->n_sen); PIECE(L.senscr, int16_t, d->n_sen); PIECE(L.rl, int32_t, d->n_root + 1); PIECE(L.arc, int32_t, n_rc + 1);
            PIECE(L.sc, PsfScalars, 1); PIECE(L.seg, s3a_psfwd_seg_t, MAX_SEG);
#undef PIECE
            L.raw = NULL;
            return at;
        };
        PsfLane probe;
        const size_t per_lane = layout(probe, (char *)NULL);
        for (int32_t z = 0; z < n_lanes; z++) {
            char *slab = NULL;
            if (hipMalloc((void **)&slab, per_lane) != hipSuccess || hipMemset(slab, 0, per_lane) != hipSuccess) {
                if (slab) (void)hipFree(slab);
                s3a_set_error("s3a_psfwd_init: device allocation failed (lane %d: %zu bytes)", z, per_lane);
                s3a_psfwd_free(e);
                return NULL;
            }
            e->dev_lane.push_back(slab);
            (void)layout(e->lanes_h[z], slab);
        }
    }
    if (hipMalloc((void **)&e->lanes_d, sizeof(PsfLane) * n_lanes) != hipSuccess
        || hipMalloc((void **)&e->lane_ids_d, sizeof(int32_t) * n_lanes) != hipSuccess
        || hipMemcpy(e->lanes_d, e->lanes_h.data(), sizeof(PsfLane) * n_lanes, hipMemcpyHostToDevice) != hipSuccess) {
        s3a_set_error("s3a_psfwd_init: device allocation failed (lane table)");
        s3a_psfwd_free(e);
        return NULL;
    }
    /* every lane starts as a new decoder */
    std::vector<int32_t> ids(n_lanes);
    for (int32_t z = 0; z < n_lanes; z++) ids[z] = z;
    if (hipMemcpy(e->lane_ids_d, ids.data(), sizeof(int32_t) * n_lanes, hipMemcpyHostToDevice) != hipSuccess) { s3a_psfwd_free(e); return NULL; }
    (void)hipDeviceSynchronize();       /* the uploads above went through the null stream */
    if (NE == 3) hipLaunchKernelGGL(k_psf_start<3>, dim3(n_lanes), dim3(NT), 0, e->stream, M, e->lanes_d, e->lane_ids_d, 1);
    else hipLaunchKernelGGL(k_psf_start<5>, dim3(n_lanes), dim3(NT), 0, e->stream, M, e->lanes_d, e->lane_ids_d, 1);
    if (hipStreamSynchronize(e->stream) != hipSuccess) { s3a_set_error("s3a_psfwd_init: the reset kernel failed: %s", hipGetErrorString(hipGetLastError())); s3a_psfwd_free(e); return NULL; }
    return e;
}

extern "C" int32_t s3a_psfwd_n_lanes(const s3a_psfwd_t *e) { return e ? e->n_lanes : 0; }
extern "C" double s3a_psfwd_last_decode_ms(const s3a_psfwd_t *e) { return e ? e->last_ms : 0.0; }
extern "C" double s3a_psfwd_last_score_ms(const s3a_psfwd_t *e) { return e ? e->last_score_ms : 0.0; }

#define LANECHK(fn) do { if (!e || lane < 0 || lane >= e->n_lanes) { s3a_set_error(fn ": bad lane"); return S3A_EINVAL; } } while (0)
#define NEPL_LAUNCH(kern, grid, ...) do { const bool pl_ = e->M.pl_window > 0; \
        if (e->M.n_emit == 3) { if (pl_) hipLaunchKernelGGL((kern<3, true>), grid, dim3(NT), 0, e->stream, __VA_ARGS__); \
                                else hipLaunchKernelGGL((kern<3, false>), grid, dim3(NT), 0, e->stream, __VA_ARGS__); } \
        else { if (pl_) hipLaunchKernelGGL((kern<5, true>), grid, dim3(NT), 0, e->stream, __VA_ARGS__); \
               else hipLaunchKernelGGL((kern<5, false>), grid, dim3(NT), 0, e->stream, __VA_ARGS__); } } while (0)
#define NE_LAUNCH(kern, grid, ...) do { if (e->M.n_emit == 3) hipLaunchKernelGGL(kern<3>, grid, dim3(NT), 0, e->stream, __VA_ARGS__); \
                                        else hipLaunchKernelGGL(kern<5>, grid, dim3(NT), 0, e->stream, __VA_ARGS__); } while (0)

static int32_t
start_lanes(s3a_psfwd_t *e, const int32_t *ids, int32_t n, int fresh)
{
    HIPCHK(hipMemcpyAsync(e->lane_ids_d, ids, sizeof(int32_t) * n, hipMemcpyHostToDevice, e->stream));
    NE_LAUNCH(k_psf_start, dim3(n), e->M, e->lanes_d, e->lane_ids_d, fresh);
    HIPCHK(hipGetLastError());
    return S3A_OK;
}

extern "C" int32_t
s3a_psfwd_start(s3a_psfwd_t *e, int32_t lane)
{
    LANECHK("s3a_psfwd_start");
    int32_t rc = start_lanes(e, &lane, 1, 0);
    if (rc != S3A_OK) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    return S3A_OK;
}

extern "C" int32_t
s3a_psfwd_reset(s3a_psfwd_t *e, int32_t lane)
{
    LANECHK("s3a_psfwd_reset");
    int32_t rc = start_lanes(e, &lane, 1, 1);
    if (rc != S3A_OK) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    return S3A_OK;
}

extern "C" int32_t
s3a_psfwd_sen_active(s3a_psfwd_t *e, int32_t lane, int32_t frame_idx, uint8_t *flags)
{
    LANECHK("s3a_psfwd_sen_active");
    if (!flags || frame_idx < 0 || frame_idx >= e->M.max_frames) { s3a_set_error("s3a_psfwd_sen_active: bad argument"); return S3A_EINVAL; }
    NE_LAUNCH(k_psf_sen_active, dim3(1), e->M, e->lanes_d, lane, frame_idx);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(flags, e->lanes_h[lane].flags, (size_t)e->M.n_sen, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return S3A_OK;
}

extern "C" int32_t
s3a_psfwd_step(s3a_psfwd_t *e, int32_t lane, const int16_t *senscr, int32_t frame_idx, int32_t n_senone_active)
{
    LANECHK("s3a_psfwd_step");
    if (!senscr || frame_idx < 0 || frame_idx >= e->M.max_frames) { s3a_set_error("s3a_psfwd_step: frame %d outside [0, %d)", frame_idx, e->M.max_frames); return S3A_EINVAL; }
    PsfScalars sc;
    HIPCHK(hipMemcpyAsync(e->lanes_h[lane].senscr, senscr, (size_t)e->M.n_sen * 2, hipMemcpyHostToDevice, e->stream));
    NEPL_LAUNCH(k_psf_step, dim3(1), e->M, e->lanes_d, lane, frame_idx, n_senone_active);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&sc, e->lanes_h[lane].sc, sizeof(sc), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (sc.status != 0) { s3a_set_error("s3a_psfwd_step: the backpointer table (%d entries) or score stack (%d) is full", e->M.bp_cap, e->M.bss_cap); return sc.status; }
    return (sc.best_score == PS_WORST || sc.best_score < PS_WORST) ? 0 : 1;
}

extern "C" int32_t
s3a_psfwd_set_lookahead(s3a_psfwd_t *e, int32_t lane, const int32_t *pl_score)
{
    LANECHK("s3a_psfwd_set_lookahead");
    if (e->M.pl_window <= 0) { s3a_set_error("s3a_psfwd_set_lookahead: the engine was built without the phone loop look-ahead (pl_window 0)"); return S3A_EUNSUP; }
    if (pl_score) HIPCHK(hipMemcpyAsync(e->lanes_h[lane].pl_host, pl_score, (size_t)e->M.n_ci * 4, hipMemcpyHostToDevice, e->stream));
    else HIPCHK(hipMemsetAsync(e->lanes_h[lane].pl_host, 0, (size_t)e->M.n_ci * 4, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));            /* (pl_score is the caller's) */
    return S3A_OK;
}

extern "C" int32_t
s3a_psfwd_finish(s3a_psfwd_t *e, int32_t lane, int32_t n_frames)
{
    LANECHK("s3a_psfwd_finish");
    if (n_frames < 0 || n_frames > e->M.max_frames) { s3a_set_error("s3a_psfwd_finish: bad frame count"); return S3A_EINVAL; }
    NE_LAUNCH(k_psf_finish, dim3(1), e->M, e->lanes_d, lane, n_frames);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    return S3A_OK;
}

extern "C" int32_t
s3a_psfwd_get_sp_ssid(s3a_psfwd_t *e, int32_t lane, uint16_t *ssid)
{
    LANECHK("s3a_psfwd_get_sp_ssid");
    const PsfModel &M = e->M;
    std::vector<uint16_t> all((size_t)MPX_STRIDE * M.n_mpx);
    HIPCHK(hipMemcpy(all.data(), e->lanes_h[lane].mpxid, all.size() * 2, hipMemcpyDeviceToHost));
    for (int32_t i = 0; i < M.n_1ph; i++)
        for (int32_t k = 0; k < M.n_emit; k++) ssid[i * M.n_emit + k] = all[(size_t)(M.n_root + i) * MPX_STRIDE + k];
    return S3A_OK;
}

extern "C" int32_t
s3a_psfwd_set_sp_ssid(s3a_psfwd_t *e, int32_t lane, const uint16_t *ssid)
{
    LANECHK("s3a_psfwd_set_sp_ssid");
    const PsfModel &M = e->M;
    std::vector<uint16_t> all((size_t)MPX_STRIDE * M.n_mpx);
    HIPCHK(hipMemcpy(all.data(), e->lanes_h[lane].mpxid, all.size() * 2, hipMemcpyDeviceToHost));
    for (int32_t i = 0; i < M.n_1ph; i++)
        for (int32_t k = 0; k < M.n_emit; k++) all[(size_t)(M.n_root + i) * MPX_STRIDE + k] = ssid[i * M.n_emit + k];
    HIPCHK(hipMemcpy(e->lanes_h[lane].mpxid, all.data(), all.size() * 2, hipMemcpyHostToDevice));
    return S3A_OK;
}

extern "C" int32_t
s3a_psfwd_table(s3a_psfwd_t *e, int32_t lane, s3a_psfwd_table_t *out)
{
    LANECHK("s3a_psfwd_table");
    if (!out) return S3A_EINVAL;
    PsfScalars sc;
    const PsfLane &L = e->lanes_h[lane];
    HIPCHK(hipMemcpy(&sc, L.sc, sizeof(sc), hipMemcpyDeviceToHost));
    const size_t n = sc.bpidx, ns = sc.bss_head, ni = (size_t)e->M.max_frames + 2;
    e->t_frame.resize(n + 1); e->t_wid.resize(n + 1); e->t_bp.resize(n + 1); e->t_score.resize(n + 1); e->t_sidx.resize(n + 1);
    e->t_realwid.resize(n + 1); e->t_valid.resize(n + 1); e->t_bss.resize(ns + 1); e->t_idx.resize(ni + 1);
    if (n) {
        HIPCHK(hipMemcpy(e->t_frame.data(), L.bp_frame, n * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(e->t_wid.data(), L.bp_wid, n * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(e->t_bp.data(), L.bp_bp, n * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(e->t_score.data(), L.bp_score, n * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(e->t_sidx.data(), L.bp_sidx, n * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(e->t_realwid.data(), L.bp_realwid, n * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(e->t_valid.data(), L.bp_valid, n, hipMemcpyDeviceToHost));
    }
    if (ns) HIPCHK(hipMemcpy(e->t_bss.data(), L.bss, ns * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(e->t_idx.data(), L.bp_idx, ni * 4, hipMemcpyDeviceToHost));
    memset(out, 0, sizeof(*out));
    out->status = sc.status; out->n_frame = sc.n_frame; out->bpidx = sc.bpidx; out->bss_head = sc.bss_head;
    out->best_score = sc.best_score; out->last_phone_best_score = sc.lp_best; out->renormalized = sc.renorm;
    out->st[1] = sc.st_root; out->st[2] = sc.st_nonroot; out->st[3] = sc.st_last; out->st[4] = sc.st_wlast; out->st[5] = sc.st_cand; out->st[6] = sc.st_sen;
    out->n_mark = (int32_t)ni - 1;
    out->frame = e->t_frame.data(); out->wid = e->t_wid.data(); out->bp = e->t_bp.data(); out->score = e->t_score.data();
    out->s_idx = e->t_sidx.data(); out->real_wid = e->t_realwid.data(); out->valid = e->t_valid.data();
    out->bscore_stack = e->t_bss.data(); out->bp_table_idx = e->t_idx.data() + 1;
    return S3A_OK;
}

extern "C" int32_t
s3a_psfwd_hyp(s3a_psfwd_t *e, int32_t lane, int32_t *out_score, s3a_psfwd_seg_t *seg, int32_t max_seg)
{
    LANECHK("s3a_psfwd_hyp");
    PsfScalars sc;
    HIPCHK(hipMemcpy(&sc, e->lanes_h[lane].sc, sizeof(sc), hipMemcpyDeviceToHost));
    if (sc.status != 0) { s3a_set_error("s3a_psfwd_hyp: the lane stopped: backpointer table or score stack full"); return sc.status; }
    if (!sc.finished) { s3a_set_error("s3a_psfwd_hyp: the utterance is not finished"); return S3A_EINVAL; }
    if (out_score) *out_score = sc.exit_score;
    if (sc.exit_bp == NO_BP) return 0;
    if (sc.n_seg > max_seg || sc.n_seg > MAX_SEG) { s3a_set_error("s3a_psfwd_hyp: %d segments, room for %d", sc.n_seg, max_seg < MAX_SEG ? max_seg : MAX_SEG); return -3; }
    if (seg) HIPCHK(hipMemcpy(seg, e->lanes_h[lane].seg, sizeof(s3a_psfwd_seg_t) * sc.n_seg, hipMemcpyDeviceToHost));
    return sc.n_seg;
}

extern "C" int32_t
s3a_psfwd_decode(s3a_psfwd_t *e, s3a_ps_mgau_t *scorer, int32_t n_utt, const float *const *feat, const int32_t *n_frames,
                 int32_t compallsen, int32_t fresh)
{
    if (!e || !scorer || !feat || !n_frames || n_utt < 1 || n_utt > e->n_lanes) { s3a_set_error("s3a_psfwd_decode: bad argument"); return S3A_EINVAL; }
    const PsfModel &M = e->M;
    const int32_t D = s3a_ps_ms_mgau_veclen(scorer);
    int32_t W = e->win;
    if (s3a_ps_ms_mgau_n_sen(scorer) != M.n_sen) { s3a_set_error("s3a_psfwd_decode: the scorer has %d senones, the search %d", s3a_ps_ms_mgau_n_sen(scorer), M.n_sen); return S3A_EINVAL; }
    size_t total = 0;
    int32_t longest = 0;
    std::vector<size_t> row0(n_utt);
    for (int32_t z = 0; z < n_utt; z++) {
        if (n_frames[z] < 0 || n_frames[z] > M.max_frames) { s3a_set_error("s3a_psfwd_decode: utterance %d has %d frames (max_frames %d)", z, n_frames[z], M.max_frames); return S3A_EINVAL; }
        if (M.pl_window > 0 && n_frames[z] > 0 && n_frames[z] <= M.pl_window) {
            s3a_set_error("s3a_psfwd_decode: utterance %d has %d frames, not more than -pl_window %d", z, n_frames[z], M.pl_window);
            return S3A_EUNSUP;
        }
        row0[z] = total; total += n_frames[z];
        longest = n_frames[z] > longest ? n_frames[z] : longest;
    }
    if (e->feat_cap < total * D) {
        if (e->feat_d) (void)hipFree(e->feat_d);
        e->feat_d = NULL; e->feat_cap = 0;
        HIPCHK(hipMalloc((void **)&e->feat_d, (total * D + 1) * 4));
        e->feat_cap = total * D;
    }
    /* every lane's senone scores of the WHOLE utterance before its search starts (int16 per senone and frame: 12 MB per
     * 10 s of a 6144-senone model): the lanes then run their utterances from the first frame to the last in ONE launch,
     * none waiting for another at window boundaries (a frame's cost varies several-fold along an utterance) */
    if (e->win <= 0 || M.pl_window > 0) W = longest > 0 ? longest : 1;      /* (the phone loop reads rows ahead of the search's: one window) */
    const size_t n_slots = (size_t)n_utt * W, n_windows = (size_t)(longest + W - 1) / W + 1;
    if (e->slot_cap < n_slots * n_windows) {
        if (e->slot_row_d) (void)hipFree(e->slot_row_d);
        e->slot_row_d = NULL; e->slot_cap = 0;
        HIPCHK(hipMalloc((void **)&e->slot_row_d, n_slots * n_windows * 4));
        e->slot_cap = n_slots * n_windows;
    }
    if (e->raw_cap < n_slots * M.n_sen) {
        if (e->raw_d) (void)hipFree(e->raw_d);
        e->raw_d = NULL; e->raw_cap = 0;
        HIPCHK(hipMalloc((void **)&e->raw_d, n_slots * M.n_sen * 2));
        e->raw_cap = n_slots * M.n_sen;
    }
    for (int32_t z = 0; z < n_utt; z++)
        if (n_frames[z]) HIPCHK(hipMemcpyAsync(e->feat_d + row0[z] * D, feat[z], (size_t)n_frames[z] * D * 4, hipMemcpyHostToDevice, e->stream));
    std::vector<int32_t> ids(n_utt);
    for (int32_t z = 0; z < n_utt; z++) {
        ids[z] = z;
        e->lanes_h[z].raw = e->raw_d + (size_t)z * W * M.n_sen;
        HIPCHK(hipMemcpyAsync((char *)e->lanes_h[z].sc + offsetof(PsfScalars, n_total), &n_frames[z], 4, hipMemcpyHostToDevice, e->stream));
    }
    HIPCHK(hipMemcpyAsync(e->lanes_d, e->lanes_h.data(), sizeof(PsfLane) * n_utt, hipMemcpyHostToDevice, e->stream));
    int32_t rc = start_lanes(e, ids.data(), n_utt, fresh);
    if (rc != S3A_OK) return rc;
    /* which feature row every (window, lane, frame of the window) slot scores; -1 = past the utterance's end */
    std::vector<int32_t> slot_row(n_slots * n_windows);
    for (size_t wi = 0; wi < n_windows; wi++)
        for (int32_t z = 0; z < n_utt; z++)
            for (int32_t t = 0; t < W; t++) {
                const int32_t f = (int32_t)wi * W + t;
                slot_row[wi * n_slots + (size_t)z * W + t] = f < n_frames[z] ? (int32_t)(row0[z] + f) : -1;
            }
    HIPCHK(hipMemcpyAsync(e->slot_row_d, slot_row.data(), slot_row.size() * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipEventRecord(e->ev0, e->stream));
    for (int32_t f0 = 0, wi = 0; f0 < (longest > 0 ? longest : 1); f0 += W, wi++) {
        rc = s3a_ps_score_slots_dev(scorer, e->feat_d, e->slot_row_d + (size_t)wi * n_slots, (int32_t)n_slots, e->raw_d, e->stream);
        if (rc != S3A_OK) return rc;
        NEPL_LAUNCH(k_psf_window, dim3(n_utt), M, e->lanes_d, e->lane_ids_d, f0, W, compallsen, fresh ? 1 : 0);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(e->ev1, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    e->last_ms = ms;
    for (int32_t z = 0; z < n_utt; z++) {
        PsfScalars sc;
        HIPCHK(hipMemcpy(&sc, e->lanes_h[z].sc, sizeof(sc), hipMemcpyDeviceToHost));
#ifdef PSF_TIMING
        if (z == 0) {
            static const char *nm[10] = { "sen_active", "normaliser", "renorm", "eval", "beam+prune_tree", "last_phone", "prune_word", "maxwpf", "word_trans", "deactivate" };
            fprintf(stderr, "PSF_TIMING lane 0, %d frames, us per frame:", sc.n_frame);
            for (int k = 0; k < 10; k++) fprintf(stderr, " %s %.1f", nm[k], sc.t_phase[k] * 0.01 / (sc.n_frame ? sc.n_frame : 1));
            fprintf(stderr, "\n");
        }
#endif
        if (sc.status != 0) { s3a_set_error("s3a_psfwd_decode: utterance %d: the backpointer table (%d entries) or score stack (%d) is full", z, M.bp_cap, M.bss_cap); return sc.status; }
    }
    return S3A_OK;
}

/* n_utt utterances (any number) through the engine's lanes as a queue: all senone scores first (one model-stationary pass
 * over the batch's frames), then ONE launch in which every lane decodes utterance after utterance.  Every utterance starts
 * from a new decoder's state.  Results: s3a_psfwd_queue_hyp. */
extern "C" int32_t
s3a_psfwd_decode_queue(s3a_psfwd_t *e, s3a_ps_mgau_t *scorer, int32_t n_utt, const float *const *feat, const int32_t *n_frames,
                       int32_t compallsen)
{
    if (!e || !scorer || !feat || !n_frames || n_utt < 1) { s3a_set_error("s3a_psfwd_decode_queue: bad argument"); return S3A_EINVAL; }
    const PsfModel &M = e->M;
    const int32_t D = s3a_ps_ms_mgau_veclen(scorer);
    if (s3a_ps_ms_mgau_n_sen(scorer) != M.n_sen) { s3a_set_error("s3a_psfwd_decode_queue: the scorer has %d senones, the search %d", s3a_ps_ms_mgau_n_sen(scorer), M.n_sen); return S3A_EINVAL; }
    size_t total = 0;
    std::vector<long long> row0(n_utt);
    for (int32_t z = 0; z < n_utt; z++) {
        if (n_frames[z] < 0 || n_frames[z] > M.max_frames) { s3a_set_error("s3a_psfwd_decode_queue: utterance %d has %d frames (max_frames %d)", z, n_frames[z], M.max_frames); return S3A_EINVAL; }
        if (M.pl_window > 0 && n_frames[z] > 0 && n_frames[z] <= M.pl_window) {
            s3a_set_error("s3a_psfwd_decode_queue: utterance %d has %d frames, not more than -pl_window %d", z, n_frames[z], M.pl_window);
            return S3A_EUNSUP;
        }
        row0[z] = (long long)total; total += n_frames[z];
    }
    if (e->feat_cap < total * D) {
        if (e->feat_d) (void)hipFree(e->feat_d);
        e->feat_d = NULL; e->feat_cap = 0;
        HIPCHK(hipMalloc((void **)&e->feat_d, (total * D + 1) * 4));
        e->feat_cap = total * D;
    }
    if (e->slot_cap < total + 1) {
        if (e->slot_row_d) (void)hipFree(e->slot_row_d);
        e->slot_row_d = NULL; e->slot_cap = 0;
        HIPCHK(hipMalloc((void **)&e->slot_row_d, (total + 1) * 4));
        e->slot_cap = total + 1;
    }
    if (e->raw_cap < (total + 1) * M.n_sen) {
        if (e->raw_d) (void)hipFree(e->raw_d);
        e->raw_d = NULL; e->raw_cap = 0;
        HIPCHK(hipMalloc((void **)&e->raw_d, (total + 1) * M.n_sen * 2));
        e->raw_cap = (total + 1) * M.n_sen;
    }
    if (e->q_cap < (size_t)n_utt) {
        void *qp[] = { e->q_next_d, e->q_nfr_d, e->q_res_d, e->q_row0_d, e->q_seg_d, e->q_order_d };
        for (void *q : qp) if (q) (void)hipFree(q);
        e->q_next_d = e->q_nfr_d = e->q_res_d = e->q_order_d = NULL; e->q_row0_d = NULL; e->q_seg_d = NULL; e->q_cap = 0;
        HIPCHK(hipMalloc((void **)&e->q_next_d, 4)); HIPCHK(hipMalloc((void **)&e->q_nfr_d, (size_t)n_utt * 4)); HIPCHK(hipMalloc((void **)&e->q_order_d, (size_t)n_utt * 4));
        HIPCHK(hipMalloc((void **)&e->q_res_d, (size_t)n_utt * PSF_QRES * 4)); HIPCHK(hipMalloc((void **)&e->q_row0_d, (size_t)n_utt * 8));
        HIPCHK(hipMalloc((void **)&e->q_seg_d, (size_t)n_utt * e->q_seg_cap * sizeof(s3a_psfwd_seg_t)));
        e->q_cap = n_utt;
    }
    for (int32_t z = 0; z < n_utt; z++)
        if (n_frames[z]) HIPCHK(hipMemcpyAsync(e->feat_d + (size_t)row0[z] * D, feat[z], (size_t)n_frames[z] * D * 4, hipMemcpyHostToDevice, e->stream));
    std::vector<int32_t> rows(total + 1);
    for (size_t i = 0; i < total; i++) rows[i] = (int32_t)i;
    HIPCHK(hipMemcpyAsync(e->slot_row_d, rows.data(), total * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->q_nfr_d, n_frames, (size_t)n_utt * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->q_row0_d, row0.data(), (size_t)n_utt * 8, hipMemcpyHostToDevice, e->stream));
    /* the lanes take the utterances longest first: whatever is taken last, when the other lanes run dry, is short (pocketsphinx_batch
     * walks the control file in order; an utterance's result does not depend on when it is decoded -- every one starts from a new
     * decoder's state -- and the results are kept by utterance index) */
    std::vector<int32_t> order(n_utt);
    for (int32_t z = 0; z < n_utt; z++) order[z] = z;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return n_frames[a] > n_frames[b]; });
    HIPCHK(hipMemcpyAsync(e->q_order_d, order.data(), (size_t)n_utt * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemsetAsync(e->q_next_d, 0, 4, e->stream));
    HIPCHK(hipMemsetAsync(e->q_res_d, 0xff, (size_t)n_utt * PSF_QRES * 4, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));            /* (rows is a local) */
    HIPCHK(hipEventRecord(e->ev0, e->stream));
    const int32_t n_wg = n_utt < e->n_lanes ? n_utt : e->n_lanes;
    PsfQueue Q;
    Q.n_utt = n_utt; Q.seg_cap = e->q_seg_cap; Q.next = e->q_next_d; Q.nfr = e->q_nfr_d; Q.row0 = e->q_row0_d; Q.raw = e->raw_d;
    Q.res = e->q_res_d; Q.seg = e->q_seg_d; Q.ready = NULL; Q.order = NULL;
    int32_t rc;
    if (n_utt <= n_wg || !s3a_variants()->ps_overlap) {
        rc = s3a_ps_score_slots_dev(scorer, e->feat_d, e->slot_row_d, (int32_t)total, e->raw_d, e->stream);
        if (rc != S3A_OK) return rc;
        HIPCHK(hipEventRecord(e->ev_mid, e->stream));
    }
    else {
        /* The scoring is float32 arithmetic that fills the vector units; the search is one workgroup per lane waiting on
         * dependent memory accesses.  They share the chip: the first utterance of every lane is scored, then the search
         * starts, and the rest of the queue is scored on a second stream beside it, a group of utterances per launch with a
         * completion mark behind each (k_psf_mark_ready) that a lane checks before it takes an utterance. */
        Q.ready = e->q_ready_d;        /* (the marks count utterances in index order: no reordering here) */
        HIPCHK(hipMemsetAsync(e->q_ready_d, 0, 4, e->stream));
        HIPCHK(hipEventRecord(e->ev_sc, e->stream));
        HIPCHK(hipStreamWaitEvent(e->stream_sc, e->ev_sc, 0));
        const int32_t group = n_wg / 4 > 0 ? n_wg / 4 : 1;
        for (int32_t u0 = 0; u0 < n_utt; ) {
            const int32_t u1 = u0 == 0 ? n_wg : (u0 + group < n_utt ? u0 + group : n_utt);
            const long long r0 = row0[u0], r1 = u1 < n_utt ? row0[u1] : (long long)total;
            if (r1 > r0) {
                rc = s3a_ps_score_slots_dev(scorer, e->feat_d, e->slot_row_d + r0, (int32_t)(r1 - r0), e->raw_d + (size_t)r0 * M.n_sen, e->stream_sc);
                if (rc != S3A_OK) { (void)hipStreamSynchronize(e->stream_sc); return rc; }
            }
            hipLaunchKernelGGL(k_psf_mark_ready, dim3(1), dim3(64), 0, e->stream_sc, e->q_ready_d, u1);
            if (u0 == 0) {
                HIPCHK(hipEventRecord(e->ev_sc, e->stream_sc));
                HIPCHK(hipStreamWaitEvent(e->stream, e->ev_sc, 0));
            }
            u0 = u1;
        }
    }
    if (!Q.ready) Q.order = e->q_order_d;
    NEPL_LAUNCH(k_psf_queue, dim3(n_wg), M, e->lanes_d, Q, compallsen);
    HIPCHK(hipGetLastError());
    if (Q.ready) {                  /* the scoring stream ends before the search can, but the timed region closes on both */
        HIPCHK(hipEventRecord(e->ev_sc, e->stream_sc));
        HIPCHK(hipStreamWaitEvent(e->stream, e->ev_sc, 0));
    }
    HIPCHK(hipEventRecord(e->ev1, e->stream));
    e->q_res_h.resize((size_t)n_utt * PSF_QRES); e->q_seg_h.resize((size_t)n_utt * e->q_seg_cap);
    HIPCHK(hipMemcpyAsync(e->q_res_h.data(), e->q_res_d, (size_t)n_utt * PSF_QRES * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(e->q_seg_h.data(), e->q_seg_d, (size_t)n_utt * e->q_seg_cap * sizeof(s3a_psfwd_seg_t), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    e->last_ms = ms;
    e->last_score_ms = 0;
    if (!Q.ready) { HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev_mid)); e->last_score_ms = ms; }
    e->q_n = n_utt;
    for (int32_t z = 0; z < n_utt; z++) {
        const int32_t *r = &e->q_res_h[(size_t)z * PSF_QRES];
        if (r[0] != 0) { s3a_set_error("s3a_psfwd_decode_queue: utterance %d: %s", z, r[0] == -1 ? "not decoded" : "the backpointer table or score stack is full"); return r[0] == -1 ? S3A_EHIP : r[0]; }
    }
    return S3A_OK;
}

extern "C" int32_t
s3a_psfwd_queue_hyp(s3a_psfwd_t *e, int32_t utt, int32_t *out_score, s3a_psfwd_seg_t *seg, int32_t max_seg)
{
    if (!e || utt < 0 || utt >= e->q_n) { s3a_set_error("s3a_psfwd_queue_hyp: bad utterance index"); return S3A_EINVAL; }
    const int32_t *r = &e->q_res_h[(size_t)utt * PSF_QRES];
    if (r[0] != 0) { s3a_set_error("s3a_psfwd_queue_hyp: utterance %d failed (status %d)", utt, r[0]); return S3A_EINVAL; }
    if (out_score) *out_score = r[1];
    if (r[2] > e->q_seg_cap || r[2] > max_seg) { s3a_set_error("s3a_psfwd_queue_hyp: %d segments, room for %d", r[2], max_seg < e->q_seg_cap ? max_seg : e->q_seg_cap); return -3; }
    if (seg && r[2] > 0) memcpy(seg, &e->q_seg_h[(size_t)utt * e->q_seg_cap], sizeof(s3a_psfwd_seg_t) * r[2]);
    return r[2];
}
