/*
 * s3a_utt.hip -- WHOLE UTTERANCES on the device: the `decode` slot of srch_funcs_t
 * (sphinx3/include/srch.h:552-555; srch.c:673-675 hands the whole block to it).
 *
 * One engine = one acoustic model, one set of lextrees, one trigram, L decoder LANES.  A lane
 * decodes one utterance from its first to its last frame without the host: per frame the
 * lextree_enter calls of the previous frame, the senone marks, CI + gated CD senone scoring,
 * HMM evaluation, histogram pruning, phone-level propagation, the ordered compaction of the
 * next active list and of the word exits (the kernel bodies of s3a_decoder_kernels.h /
 * s3a_gated.h: the very code of the frame-synchronous path) and then the WORD LEVEL
 * (s3a_wordlevel.h: trigram look-ups, Viterbi history, pruning, word transitions), which
 * leaves the next frame's lextree_enter calls in device memory.  Everything a frame needs to
 * know about the previous one lives in HBM (UCtx), so the host only ENQUEUES: L lanes share
 * every launch (grid z), there is no synchronisation inside an utterance, and the kernels'
 * grids are fixed (virtual workgroups looped over the list lengths found in memory) so that the
 * per-frame launch sequence is the same for every frame.  At the end the host reads each lane's
 * history table back; the reference's own vithist_utt_end / backtrace (or s3a_uttdec_hyp) turn
 * it into the hypothesis.
 *
 * Parity: tests/test_gpu_wordlevel.py (the word level against the oracle, frame by frame, on
 * recorded RM1 / tidigits traces and on random frames with tied scores) and
 * tests/test_gpu_dropin.py (S3A_UTT: -hyp / -hypseg byte-identical to the unmodified reference).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <limits.h>
#include <vector>
#include <map>
#include <atomic>
#include <mutex>
#include <algorithm>
#include <fcntl.h>
#include <unistd.h>
#include <sys/file.h>
#include <sys/stat.h>

#include "s3a_device.h"
#include "s3a_structs.h"
#include "s3a_decoder_kernels.h"
#include "s3a_gated.h"
#include "s3a_wordlevel.h"
#include "s3a_lm3g.h"
#include "s3a_dag.h"

/* A pointer that came out of a structure in memory is a GENERIC pointer to the compiler; what it makes of an access through one is
 * flat_load / flat_store -- counted by the LDS counter as well as the memory counter, so that every wait for an LDS read waits for
 * all flat accesses in flight (loads AND stores), and a step that mixes LDS reads with gathers runs them one after the other.
 * Everything the lanes' and the engine's structures point to is device memory: the hot loops of ku_frames say so (GM / GMC: the
 * same pointer in the global address space -> global_load / global_store, the memory counter only), and read LDS as LDS (LM). */
#define S3A_AS1 __attribute__((address_space(1)))
#define S3A_AS3 __attribute__((address_space(3)))
typedef int s3a_v4i __attribute__((ext_vector_type(4)));
typedef int s3a_v2i __attribute__((ext_vector_type(2)));
template <class T> __device__ __forceinline__ S3A_AS1 T *GM(T *p) { return (S3A_AS1 T *)p; }
template <class T> __device__ __forceinline__ const S3A_AS1 T *GMC(const T *p) { return (const S3A_AS1 T *)p; }
template <class T> __device__ __forceinline__ const S3A_AS3 T *LM(const T *p) { return (const S3A_AS3 T *)p; }
__device__ __forceinline__ void lds_or(uint32_t *p, uint32_t v) { (void)__hip_atomic_fetch_or((S3A_AS3 uint32_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

/* one lane: the lextree state of a clone, a private scorer state, its history table */
struct ULane {
    /* lextree state */
    int32_t *sc, *hist, *outs, *outh, *bests, *frame, *pos, *posf, *act[2], *nact[2], *turn, *selfemit, *cnt, *base,
        *best, *exits, *nexit, *first, *eflag, *hbin, *done, *ctot, *n0, *pstamp, *propf, *poswid, *posout, *scan_flag;
    unsigned long long *scan_agg, *scan_pre, *key;
    /* scorer state */
    uint8_t *sen_act;
    int32_t *scr, *misc, *bstidx, *bstscr, *updatetime, *gpart;
    int32_t *cs_need, *cs_val;  /* [n_cs] composite senones: wanted in frame (stamp) | score of the frame */
    int32_t *cs_wl, *cs_wn;     /* [n_cs + 1], [1] ku_frames: the composite senones wanted in the frame, each once, any order | their number */
    int32_t *posbest;           /* [N] ku_frames: by list position, the HMM's best score of the frame (as poswid / posout) */
    uint32_t *senbits;          /* [KF_SENBITS / 32] ku_frames with clusters: the frame's active senones, a bit each (the workgroups' masks OR-ed together; zero between frames) */
    int32_t *posps;             /* [N] ku_frames, 3-state HMMs: by list position, the node's parent set + 1 (0: none) and, bit 29, whether it is a member of
                                 * a several-parent set (from the packed node) */
    int32_t *ent;               /* [2 ent_cap] ku_frames: lextree_enter's scratch (the entries that pass the threshold test) */
    uint8_t *pstamp8;           /* [n_pset] the parent sets' stamps, the frame number's low 8 bits (a quarter of the
                                 * sweep's gathers' footprint; a stale match costs a walk that finds nothing) */
    int32_t *dynbeam;           /* [1] the frame's CI beam when -maxcdsenpf is in force (ku_dyn_ci_beam) */
    int32_t *claim;             /* [N >= n_pset] the frame in which a parent set was last listed (d_stamp_and_list) */
    int32_t *plist, *pcnt;      /* [N], [2]: the parent sets stamped in the frame (by an HMM whose exit score reaches the phone threshold), any order;
                                 * their number by frame parity (the stamping pass appends; ku_resolve_plist walks their members) */
    int32_t *win;               /* [K][n_sen] look-ahead window: every senone's score for the frames f0 .. f0 + K - 1 */
    uint8_t *winb;              /* [K][n_sen] ... and the best component of its mixture (255: none) */
    /* -pheurtype > 0 (s3a_uttdec_enable_pheur) */
    int32_t *ci_all;            /* [max_frames][n_ci_sen] the utterance's raw CI senone scores (gmm_compute_lv1 for every frame) */
    int32_t *heur_all;          /* [max_frames][n_ci] phn_heur_list of every frame (pl_computePhnHeur) */
    int32_t *hth_pos;           /* [3][N] by list position (tree slices): the heuristic threshold of the HMM listed there; ku_weak_heur's survivors */
    int32_t *ph_scratch;        /* [3 n_ci_sen + 8] the look-ahead pass's throw-away best-Gaussian state and counters */
    /* this utterance */
    UCtx *ctx;
    int32_t *pack;
    WLane w;
};

/* what every lane shares */
struct UShared {
    int32_t N, T, n_tmat, maxn, n_rootnodes, scan_chunks, pack_max_exits, gp_n, n_cs, n_pset_bytes;
    int32_t ne;                 /* emitting states per HMM (3 or 5): the node record's layout, s3a_structs.h */
    const int32_t *node_base, *ssid, *tmatid, *wid, *prob, *child_off, *child, *par_off, *par, *tree_of, *rootlist, *tp,
        *rootnodes, *ps, *psof_off, *psof, *psmem_off, *psmem, *cs_off, *cs_wt, *rootprob;
    const uint8_t *comp;
    const int16_t *sseq, *comsseq, *cs_list;
    const int4 *node4;          /* per node: {ssid, tmatid, wid, composite}: ku_hmm_eval's static words as one load */
    const int32_t *nodesen;     /* per node, 2 (3 states) or 4 (5 states) words: its senone ids -- of a composite node its composite-senone ids --
                                 * as 16-bit halves (ku_frames: one load instead of the chain node -> sequence id -> three 2-byte gathers) */
    const int4 *pshdr;          /* per parent set ONE 16-byte word for what ku_frames' propagation asks of a listed set: its members' run in psmem
                                 * (first, end), its parents' run in par (first, count) -- instead of the chain set -> psmem_off -> psmem -> par_off */
    const int4 *nodepk;         /* 3-state HMMs, per node ONE 16-byte word for everything ku_frames' steps read of it: senone ids 0 | 1 << 16,
                                 * id 2 | transition matrix << 16, word id, (parent set + 1) << 1 | composite (what is asked for together
                                 * lives together: one cache line per visit instead of node4's + nodesen's + ps's three) */
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
    int32_t ds_ratio, ci_pbeam, ci_pbeam_tight, ptranskip;
    int32_t max_cd;             /* -maxcdsenpf; >= the number of CD senones: no dynamic beam */
    float tighten;
    FrameBeams bm;              /* phone_uses_wbeam is worked out per frame */
    /* what decides whether a workgroup has anything to do sits at addresses known from the kernel arguments alone: the
     * lanes' contexts and active-list lengths are two arrays (ONE round trip to the early exit instead of lane struct ->
     * pointer -> value: with 64 lanes most workgroups of a fixed grid only find out that they are not needed, and
     * that chain times the number of such waves over the chip's resident waves WAS the launch) */
    int32_t pheurtype, pl_beam, pl_window, n_ci;        /* -pheurtype (0: off), logs3(-pl_beam), -pl_window, #CI phones */
    const uint8_t *node_ci;     /* [N] CI phone of every lextree node (lextree_node_t.ci) */
    const int16_t *sen2cimap;   /* [n_ci_sen + 1] mdef_t.sen2cimap of the CI senones and the first CD senone */
    int32_t win_K;              /* > 0: look-ahead scoring, K frames per window (ku_score_window / ku_select); 0: per-frame scoring */
    UCtx *ctx_all;              /* [n_lanes] */
    int32_t *nact_all;          /* [n_lanes][2][WL_MAXT] */
    const int32_t *fgbase;      /* [1] what the kernels add to their frame argument: 0 in stream mode (the argument is the engine's
                                 * frame counter); in graph mode a block of frames is captured ONCE with the arguments 0 .. G - 1 and
                                 * replayed, the block's last node (ku_advance) moves this counter on by G */
};

/* the lanes share every launch: the kernels' argument fg is the ENGINE's frame counter, a lane's own frame is
 * f = fg - ctx->f0 (f0 = 0 when all lanes start together; a lane that took its next utterance from the queue when the
 * last one ended -- s3a_uttdec_decode_queue -- started later), the list searched in frame f is list f & 1
 * (lextree_active_swap flips it every frame) -- no kernel has to read what the word level of the previous frame wrote
 * last, so the word level can share a launch with the emission sweep */
#define LANE UCtx *ctx = S.ctx_all + blockIdx.z; const int32_t f = fg + *S.fgbase - ctx->f0;                         \
    if (f < 0 || f >= ctx->nfr || !ctx->active) return;                                                           \
    const int32_t cur = f & 1; const int32_t *nact_cur = S.nact_all + ((size_t)blockIdx.z * 2 + cur) * WL_MAXT;       \
    const ULane &L = lanes[blockIdx.z]; (void)cur; (void)nact_cur
/* the frame's senone scores: the lane's row of the look-ahead window, or the per-frame scorer's array */
#define SCR_ROW (S.win_K > 0 ? L.win + (size_t)(f % S.win_K) * S.n_sen : L.scr)

__device__ __forceinline__ FrameBeams
frame_beams(const UShared &S, int32_t cf)
{
    FrameBeams bm = S.bm;
    bm.phone_uses_wbeam = (S.ptranskip != 0 && (cf % S.ptranskip) == 0) ? 1 : 0;   /* srch_time_switch_tree.c:975-1003 */
    return bm;
}

/* ---- utterance boundaries of ALL lanes in two launches ----
 * (lextree_utt_end + srch_TST_begin per lane were ~25 small launches, a copy and a stream synchronisation EACH: with
 * 128 lanes per engine several percent of a batch)
 * ku_lanes_end: hmm_clear on whatever the last utterance left active (lextree_utt_end, lextree.c:1666-1700), both lists;
 * ku_lanes_begin: the frame-tagged scratch, the scorer's per-senone state (cont_mgau.c:1183-1190 as srch_TST_begin
 * :485-490 resets it), the masks, and the history table's entry 0 (vithist_utt_begin, vithist.c:300-335). */
struct UBegin { int32_t e0[10], lmc[5], n_pset; };

/* lextree_utt_end for lane z by the threads (vt, vt + vstride, ...) of whatever grid the caller has; scrub: the lane's utterance
 * stopped on an error in mid-frame -- everything from scratch, as the host does for a dirty lane (lane_scrub: s3a_lexsearch_reset +
 * the word level's hash / per-word scratch) */
__device__ __forceinline__ void
d_lane_end(const ULane &L, const UShared &S, int32_t z, bool scrub, int32_t n_word, int32_t vt, int32_t vstride)
{
    const int32_t ne = S.ne;
    if (scrub) {
        for (int32_t v = vt; v < S.N; v += vstride) {
            int32_t *r = L.sc + NSV(v);
            for (int32_t st = 0; st < ne; st++) { r[st] = WORST; r[NS_HIST(ne) + st] = -1; }
            r[NS_OUTS(ne)] = WORST; r[NS_OUTH(ne)] = -1; r[NS_BESTS(ne)] = WORST; r[NS_FRAME(ne)] = -1;
            L.pos[PPX(v)] = -1; L.turn[v] = -1; L.selfemit[v] = 0; L.cnt[v] = 0; L.first[v] = INT_MAX; L.key[v] = 0ull;
        }
        for (int32_t i = vt; i <= L.w.hmask; i += vstride) { L.w.hkey[i] = 0ull; L.w.hbest[i] = 0ull; L.w.hfirst[i] = 0xffffffffu; }
        for (int32_t i = vt; i < n_word; i += vstride) { L.w.wfirst[i] = INT_MAX; L.w.wbest[i] = INT_MIN; }
        for (int32_t i = vt; i < 2 * S.T; i += vstride) { L.nexit[i] = 0; L.best[i] = INT_MIN; }
        for (int32_t i = vt; i < 1024; i += vstride) L.hbin[i] = 0;
        if (vt < 4) L.done[vt] = 0;
        return;
    }
    for (int32_t w = 0; w < 2; w++)
        for (int32_t t = 0; t < S.T; t++) {
            const int32_t na = S.nact_all[((size_t)z * 2 + w) * WL_MAXT + t], b = S.node_base[t];
            for (int32_t i = vt; i < na; i += vstride) {
                const int32_t v = L.act[w][b + i];
                int32_t *r = L.sc + NSV(v);
                for (int32_t st = 0; st < ne; st++) { r[st] = WORST; r[NS_HIST(ne) + st] = -1; }
                r[NS_OUTS(ne)] = WORST; r[NS_OUTH(ne)] = -1; r[NS_BESTS(ne)] = WORST; r[NS_FRAME(ne)] = -1;
            }
        }
}

/* (sub: the lanes of a refill event, s3a_uttdec_decode_queue -- NULL: lanes 0 .. gridDim.z - 1) */
__global__ void __launch_bounds__(256)
ku_lanes_end(const ULane *__restrict__ lanes, UShared S, const int32_t *__restrict__ sub, int32_t n_word)
{
    const int32_t z = sub ? sub[blockIdx.z] : (int32_t)blockIdx.z;
    const int32_t nb = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
    d_lane_end(lanes[z], S, z, sub && S.ctx_all[z].err, n_word, b * 256 + (int32_t)threadIdx.x, nb * 256);
}

/* srch_TST_begin's resets for lane z (threads vt, vt + vstride, ...; lead: the one workgroup -- threads tid of it -- that also writes
 * the small things): the frame-tagged scratch, the scorer's per-senone state, the masks, the history table's entry 0; src: the
 * utterance's staged context (a queue), copied word by word into the lane's */
__device__ __forceinline__ void
d_lane_begin(const ULane &L, const UShared &S, const UBegin &B, int32_t z, const UCtx *src, int32_t vt, int32_t vstride, bool lead, int32_t tid, int32_t lead_nt)
{
    for (int32_t i = vt; i < S.N; i += vstride) { L.posf[PPX(i)] = INT_MIN; L.propf[i] = INT_MIN; L.claim[i] = INT_MIN; }
    for (int32_t i = vt; i < B.n_pset; i += vstride) L.pstamp[i] = INT_MIN;
    for (int32_t i = vt; i < S.n_pset_bytes; i += vstride) L.pstamp8[i] = 0xff;
    if (vt == 0) { L.pcnt[0] = 0; L.pcnt[1] = 0; L.cs_wn[0] = 0; }
    for (int32_t i = vt; i < S.n_sen; i += vstride) {
        L.bstidx[i] = S3A_NO_BSTIDX; L.bstscr[i] = S3A_LOGPROB_ZERO; L.updatetime[i] = S3A_NOT_UPDATED; L.sen_act[i] = 0;
    }
    for (int32_t i = vt; i <= S.n_cs; i += vstride) L.cs_need[i] = -1;
    if (lead) {
        if (tid < 2 * WL_MAXT) S.nact_all[(size_t)z * 2 * WL_MAXT + tid] = 0;
        if (tid < 8) L.misc[tid] = (tid == 0 || tid == 5) ? INT_MIN : 0;
        if (tid == 32) {
            int32_t *arr[10] = { L.w.score, L.w.pred, L.w.lw0, L.w.lw1, L.w.wid, L.w.sf, L.w.ef, L.w.ascr, L.w.lscr, L.w.type };
            for (int k = 0; k < 10; k++) arr[k][0] = B.e0[k];
            for (int k = 0; k < 5; k++) L.w.lmc[(size_t)k * L.w.cap] = B.lmc[k];
            L.w.frame_start[0] = 1; L.w.bestscore[0] = INT_MIN; L.w.bestvh[0] = -1; L.w.st[0] = 1; L.w.st[1] = 0;
        }
        if (src) {
            static_assert(sizeof(UCtx) % 4 == 0, "UCtx is copied word by word");
            const int32_t *s4 = (const int32_t *)src;
            int32_t *dst = (int32_t *)(S.ctx_all + z);
            for (int32_t i = tid; i < (int32_t)(sizeof(UCtx) / 4); i += lead_nt) dst[i] = s4[i];
        }
    }
}

/* (sub / utt / stage: a refill event -- lane sub[z] takes utterance utt[z] of the queue, whose context was staged on the
 * device before the first frame) */
__global__ void __launch_bounds__(256)
ku_lanes_begin(const ULane *__restrict__ lanes, UShared S, UBegin B, const int32_t *__restrict__ sub,
               const int32_t *__restrict__ utt, const UCtx *__restrict__ stage)
{
    const int32_t z = sub ? sub[blockIdx.z] : (int32_t)blockIdx.z;
    d_lane_begin(lanes[z], S, B, z, stage ? stage + utt[blockIdx.z] : (const UCtx *)NULL, blockIdx.x * 256 + (int32_t)threadIdx.x, gridDim.x * 256,
                 blockIdx.x == 0, threadIdx.x, 256);
}

/* debugging (S3A_UTT_FRAMECHECK=1): after every frame, every node record that is not an inactive HMM must be on the NEXT
 * list at the place its record names; the first violation of a lane is kept: dbg[0] = frame + 1 (0: none), [1] node,
 * [2..7] sc0 sc1 outs bests frame-tag posf, [8] pos, [9] turn, [10] list length of its tree, [11] what sits at pos */
__global__ void __launch_bounds__(256)
ku_framecheck(const ULane *__restrict__ lanes, UShared S, int32_t fg, int32_t *dbg_all)
{
    LANE;
    int32_t *dbg = dbg_all + 16 * blockIdx.z;
    const int32_t nf = f + 1, nxt = cur ^ 1;
    for (int32_t v = blockIdx.x * 256 + threadIdx.x; v < S.N; v += gridDim.x * 256) {
        const int32_t *r = L.sc + NSV(v);
        const int32_t ne = S.ne;
        bool clean = r[NS_OUTS(ne)] == WORST && r[NS_BESTS(ne)] == WORST;
        for (int32_t st = 0; st < ne; st++) clean = clean && r[st] == WORST;
        const int32_t t = S.tree_of[v], b = S.node_base[t], p = L.pos[PPX(v)];
        const int32_t nn = S.nact_all[((size_t)blockIdx.z * 2 + nxt) * WL_MAXT + t];
        const bool listed = L.posf[PPX(v)] == nf && p >= 0 && p < nn && L.act[nxt][b + p] == v;
        /* the propagation scratch must be consumed by the end of the frame: turn by node; selfemit / cnt by list position */
        if (L.turn[v] != -1 || L.selfemit[v] != 0 || L.cnt[v] != 0) {
            if (atomicCAS(&dbg[15], 0, f + 1) == 0) {
                /* (reported through the same record when the node check finds nothing) */
                if (atomicCAS(&dbg[0], 0, -(f + 1)) == 0) {
                    dbg[1] = v; dbg[2] = L.turn[v]; dbg[3] = L.selfemit[v]; dbg[4] = L.cnt[v]; dbg[5] = ctx->nfr; dbg[6] = S.nact_all[((size_t)blockIdx.z * 2 + cur) * WL_MAXT + S.tree_of[v]];
                    dbg[7] = v - S.node_base[S.tree_of[v]]; dbg[14] = S.tree_of[v];
                }
            }
        }
        if ((!clean && !listed) || (listed && r[NS_FRAME(ne)] != nf)) {
            if (atomicCAS(&dbg[0], 0, f + 1) == 0) {
                dbg[1] = v; dbg[2] = r[0]; dbg[3] = r[1]; dbg[4] = r[NS_OUTS(ne)]; dbg[5] = r[NS_BESTS(ne)]; dbg[6] = r[NS_FRAME(ne)];
                dbg[7] = L.posf[PPX(v)]; dbg[8] = p; dbg[9] = L.turn[v]; dbg[10] = nn; dbg[11] = (p >= 0 && p < nn) ? L.act[nxt][b + p] : -7;
                dbg[12] = clean ? 1 : 0; dbg[13] = listed ? 1 : 0; dbg[14] = t;
            }
        }
    }
}

/* self-check (tests / debugging): after ku_lanes_end every node record of a lane must be an inactive HMM (hmm_clear);
 * out[0..5] = nodes whose state-0 / -1 / -2 score, exit score, best score is not WORST, whose frame tag is not -1;
 * out[6] = the first such node; out[7] = nodes with turn != -1 | selfemit / cnt left set (by position) */
__global__ void __launch_bounds__(256)
ku_selfcheck(const ULane *__restrict__ lanes, UShared S, int32_t lane, int32_t *out)
{
    const ULane &L = lanes[lane];
    for (int32_t v = blockIdx.x * 256 + threadIdx.x; v < S.N; v += gridDim.x * 256) {
        const int32_t *r = L.sc + NSV(v);
        bool bad = false;
        const int32_t ne = S.ne;
        for (int32_t st = 0; st < ne; st++)         /* (states 2 .. ne - 1 are counted together) */
            if (r[st] != WORST) { atomicAdd(&out[st < 2 ? st : 2], 1); bad = true; }
        if (r[NS_OUTS(ne)] != WORST) { atomicAdd(&out[3], 1); bad = true; }
        if (r[NS_BESTS(ne)] != WORST) { atomicAdd(&out[4], 1); bad = true; }
        if (r[NS_FRAME(ne)] != -1) { atomicAdd(&out[5], 1); bad = true; }
        if (bad) atomicMin(&out[6], v);
        if (L.turn[v] != -1 || L.selfemit[v] != 0 || L.cnt[v] != 0) atomicAdd(&out[7], 1);
    }
}

__global__ void
ku_pack_pshdr(const int32_t *__restrict__ psmem_off, const int32_t *__restrict__ psmem, const int32_t *__restrict__ par_off, int4 *__restrict__ out, int32_t n_pset)
{
    const int32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_pset) return;
    const int32_t m_lo = psmem_off[q], m_hi = psmem_off[q + 1];
    int32_t kp0 = 0, np = 0;
    if (m_hi > m_lo) { const int32_t x0 = psmem[m_lo]; kp0 = par_off[x0]; np = par_off[x0 + 1] - kp0; }
    out[q] = make_int4(m_lo, m_hi, kp0, np);
}

/* ---- lextree_enter calls left by the previous frame's word level (or by utterance begin) ---- */
/* the entry test (d_dec_enter1) with the calls' table in LDS: an entry finds its call by a bisection in LDS and fails the
 * test -- almost all do -- after ONE global round trip (its root's look-ahead probability, in list order) */
__global__ void __launch_bounds__(256)
ku_enter1(const ULane *__restrict__ lanes, UShared S, int32_t fg)
{
    __shared__ int32_t s_off[WL_MAXCALL], s_root[WL_MAXCALL], s_in[WL_MAXCALL];
    LANE;
    const int32_t n_ent = ctx->n_ent, n_calls = min(ctx->n_calls, WL_MAXCALL), thresh = ctx->thresh;
    if ((int32_t)blockIdx.x * 256 >= n_ent) return;
    if ((int32_t)threadIdx.x < n_calls) {
        s_in[threadIdx.x] = ctx->calls[4 * threadIdx.x]; s_root[threadIdx.x] = ctx->calls[4 * threadIdx.x + 2];
        s_off[threadIdx.x] = ctx->calls[4 * threadIdx.x + 3];
    }
    __syncthreads();
    for (int32_t e = blockIdx.x * 256 + threadIdx.x; e < n_ent; e += gridDim.x * 256) {
        int32_t lo = 0, hi = n_calls - 1;
        while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (s_off[mid] <= e) lo = mid; else hi = mid - 1; }
        const int32_t c = lo, idx = s_root[c] + (e - s_off[c]);
        const int32_t scr = add32(s_in[c], S.rootprob[idx]);
        if (scr < thresh) continue;
        const int32_t v = S.rootlist[idx];
        if (!(L.sc[NSV(v)] < scr)) continue;
        atomicMax(&L.key[v], ((unsigned long long)((uint32_t)scr ^ 0x80000000u) << 32) | (uint32_t)(0x7fffffff - c));
        atomicMin(&L.first[v], c);
    }
}

/* (a workgroup per lextree_enter call; with many lanes fewer workgroups that take the calls in turn) */
template <int NT>
__global__ void __launch_bounds__(NT)
ku_enter2(const ULane *__restrict__ lanes, UShared S, int32_t fg)
{
    LANE;
    const int32_t n_calls = ctx->n_calls;
    if ((int32_t)blockIdx.x >= n_calls || ctx->n_ent == 0) return;
    const Entries ent = { ctx->calls, S.rootlist, n_calls, S.rootprob };
    for (int32_t c = blockIdx.x; c < n_calls; c += gridDim.x) {
        d_dec_enter2_t<NT>(ent, ctx->n_ent, ctx->calls, S.prob, L.sc, L.frame, L.first, ctx->thresh, f, S.T,
                     L.nact[cur], L.eflag, L.ctot, L.n0, c, 0);
        __syncthreads();
    }
}

__global__ void __launch_bounds__(M3BLOCK)
ku_enter3_mark(const ULane *__restrict__ lanes, UShared S, int32_t fg)
{
    LANE;
    const int32_t n_ent = ctx->n_ent;
    const int32_t *n0 = n_ent > 0 ? L.n0 : L.nact[cur];
    int32_t rows = 0;
    for (int32_t t = 0; t < S.T; t++) rows = max(rows, n0[t]);
    const int32_t n_ent_blocks = (n_ent + M3BLOCK - 1) / M3BLOCK, bpt = (rows + M3BLOCK - 1) / M3BLOCK;
    const Entries ent = { ctx->calls, S.rootlist, ctx->n_calls, S.rootprob };
    for (int32_t vb = blockIdx.x; vb < n_ent_blocks + bpt * S.T; vb += gridDim.x)
        d_dec_enter3_mark(n_ent_blocks, ent, n_ent, ctx->calls, ctx->groups, ctx->n_groups, f, L.key, L.first, L.eflag,
                          L.ctot, n0, L.sc, L.hist, L.frame, S.T, bpt, S.node_base, L.act[cur], L.nact[cur], L.pos,
                          L.posf, S.ssid, S.comp, S.sseq, S.comsseq, S.cs_off, S.cs_list, L.sen_act, vb, 0, L.cs_need, ctx->thresh);
}

/* ---- approx_cont_mgau_ci_eval / _frame_eval for the lane's frame (s3a_gated.h) ---- */
template <bool EXACT, bool CI>
__global__ void __launch_bounds__(256)
ku_gated(const ULane *__restrict__ lanes, UShared S, int32_t fg)
{
    LANE;
    const int32_t lo = CI ? 0 : S.n_ci_sen, hi = CI ? S.n_ci_sen : S.n_sen, cf = f;
    if (CI) {               /* the CI launch's extra workgroups: the members of the composite senones wanted in this frame */
        const int32_t g_ci = (S.n_ci_sen * S.CP + 255) / 256;
        if ((int32_t)blockIdx.x >= g_ci) {
            d_comsen_wave<false>(S.n_cs, L.cs_need, f, S.cs_off, S.cs_list, L.sen_act, (const int32_t *)NULL, (int32_t *)NULL,
                                 ((int32_t)blockIdx.x - g_ci) * 256 + (int32_t)(threadIdx.x & ~63));
            return;
        }
    }
    if ((int32_t)(blockIdx.x * 256) >= (hi - lo) * S.CP) return;
    const float *x = ctx->feat + (size_t)cf * S.D4 * 4;
    const int32_t is_skip = (cf % S.ds_ratio == 0) ? 0 : 1;
    const int32_t beam = (!CI && S.max_cd < S.n_sen - S.n_ci_sen) ? L.dynbeam[0] : (is_skip ? S.ci_pbeam_tight : S.ci_pbeam);
#define KU_GATED_ARGS S.mean4, S.prec4, S.lrd, S.mixw, S.tab16, S.tab_size, S.lm_zero, S.f, S.distfloor, x, S.D4, S.CP,  \
        S.Gpad, lo, hi, CI ? 1 : 0, S.ncomp, S.cd2cisen, L.sen_act, L.scr, 0, CI ? (const int32_t *)NULL : L.misc + 5,     \
        CI ? 0 : beam, cf, CI ? 0 : is_skip, L.bstidx, L.bstscr, L.updatetime, L.misc, CI ? 5 : 0,                        \
        CI ? (uint8_t *)NULL : L.sen_act, CI ? (int32_t *)NULL : L.gpart, S.gp_n
    if (S.D4 == D4MAIN)
        d_gated_frame<EXACT, D4MAIN>(KU_GATED_ARGS, blockIdx.x);
    else
        d_gated_frame<EXACT, 0>(KU_GATED_ARGS, blockIdx.x);
}

/*
 * The gated CD senones of ALL lanes in one pass over the model (approx_cont_mgau_frame_eval x lanes): the
 * model-stationary scoring kernel (k_score_frames, s3a_device.hip) with the lanes' frames in the place of an
 * utterance's frames.  A lane of the wave keeps ONE Gaussian in registers and evaluates it for a group of UG_FB
 * decoder lanes (their feature vectors broadcast from LDS), the values are transposed through a per-wave LDS tile,
 * and wave lane (senone, c) then runs the gate and the ordered log-add of decoder lane c of the group for its
 * senone.  The model is read once per launch-row instead of once per decoder lane (per-lane launches moved
 * lanes x 15.7 MB per frame: 32 lanes = 0.5 GB), with coalesced 16-byte loads; every Gaussian is computed and the
 * gate only selects (s3a_gated.h).  Maxima / counters leave the workgroup as plain stores into the decoder lane's
 * gpart[] column (merged by d_dec_hmm_eval / d_dec_pack_frame): no atomics.  Same results as ku_gated, bit for bit.
 */
#define UG_FB 8
#define UG_MAX 32
struct UgDec {
    uint8_t *sen_act;
    int32_t *scr, *gpart, *bstidx, *bstscr, *updatetime;
    int32_t frame, is_skip, thresh, active;
};

template <bool EXACT>
__global__ void __launch_bounds__(256)
ku_gated_cd_multi(const ULane *__restrict__ lanes, UShared S, int32_t n_lanes, int32_t fg)
{
    typedef typename Acc<EXACT>::T acc_t;
    __shared__ UgDec dec[UG_MAX];
    __shared__ int32_t red[4][UG_MAX][3];
    __shared__ int32_t tr_s[4][UG_FB * 65];
    __shared__ float4 xs4[UG_MAX * D4MAIN];
    const int32_t zb = blockIdx.z * UG_MAX, n = min(UG_MAX, n_lanes - zb);
    const int32_t CP = S.CP, Gpad = S.Gpad;
    const int32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t g = S.n_ci_sen * CP + blockIdx.x * 256 + tid;
    const int32_t sen = g / CP, c = g - sen * CP, sl = lane / CP;
    const bool valid = sen < S.n_sen;
    float4 M[D4MAIN], P[D4MAIN];
    float lrd_g = 0.0f;
    int32_t mixw = 0, nc = 0, ci_id = 0;
    if (valid) {
#pragma unroll
        for (int k = 0; k < D4MAIN; k++) { M[k] = S.mean4[(size_t)k * Gpad + g]; P[k] = S.prec4[(size_t)k * Gpad + g]; }
        lrd_g = S.lrd[g];
        mixw = S.mixw[g];
        nc = (int32_t)S.ncomp[sen];
        ci_id = S.cd2cisen[sen];
    }
    else {
#pragma unroll
        for (int k = 0; k < D4MAIN; k++) M[k] = P[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (tid < UG_MAX) {
        UgDec d;
        memset(&d, 0, sizeof d);
        if (tid < n) {
            const ULane &Lz = lanes[zb + tid];
            const UCtx *cx = Lz.ctx;
            const int32_t cf = fg + *S.fgbase - cx->f0;        /* the lane's own frame */
            d.active = (cx->active && cf >= 0 && cf < cx->nfr) ? 1 : 0;
            if (d.active) {
                d.sen_act = Lz.sen_act; d.scr = Lz.scr; d.gpart = Lz.gpart; d.bstidx = Lz.bstidx; d.bstscr = Lz.bstscr;
                d.updatetime = Lz.updatetime; d.frame = cf; d.is_skip = (cf % S.ds_ratio == 0) ? 0 : 1;
                d.thresh = add32(Lz.misc[5], S.max_cd < S.n_sen - S.n_ci_sen ? Lz.dynbeam[0]
                                                                                  : (d.is_skip ? S.ci_pbeam_tight : S.ci_pbeam));
            }
        }
        dec[tid] = d;
    }
    for (int32_t i = tid; i < 4 * UG_MAX; i += 256) {
        red[i / UG_MAX][i % UG_MAX][0] = INT_MIN; red[i / UG_MAX][i % UG_MAX][1] = 0; red[i / UG_MAX][i % UG_MAX][2] = 0;
    }
    __syncthreads();
    for (int32_t i = tid; i < UG_MAX * D4MAIN * 4; i += 256) {
        const int32_t zz = i / (D4MAIN * 4), k = i - zz * (D4MAIN * 4);
        float v = 0.0f;
        if (zz < n && dec[zz].active) { const UCtx *cx = lanes[zb + zz].ctx; v = cx->feat[(size_t)dec[zz].frame * (D4MAIN * 4) + k]; }
        ((float *)xs4)[i] = v;
    }
    __syncthreads();
    LogAdd la;
    la.tab = S.tab16; la.size = S.tab_size; la.zero = S.lm_zero;
    int32_t *tr = tr_s[wave];
    const int32_t n_groups = (n + UG_FB - 1) / UG_FB;
    for (int32_t grp = blockIdx.y; grp < n_groups; grp += gridDim.y) {
        const int32_t z0 = grp * UG_FB, nd = min(UG_FB, n - z0);
        bool mine = valid && c < nd;
        int32_t act = 0, ut = 0, ci_scr = 0, bi = S3A_NO_BSTIDX;
        UgDec d;
        if (mine) {
            d = dec[z0 + c];
            mine = d.active != 0;
        }
        if (mine) {
            act = d.sen_act[sen]; ut = d.updatetime[sen];
            ci_scr = d.scr[ci_id];
            bi = d.bstidx[sen];
        }
        acc_t a[UG_FB];
#pragma unroll
        for (int j = 0; j < UG_FB; j++) a[j] = (acc_t)lrd_g;
#pragma unroll
        for (int k = 0; k < D4MAIN; k++) {
#pragma unroll
            for (int j = 0; j < UG_FB; j++) {
                if (j < nd) {
                    const float4 x = xs4[(z0 + j) * D4MAIN + k];
                    a[j] = Acc<EXACT>::step(a[j], x.x, M[k].x, P[k].x);
                    a[j] = Acc<EXACT>::step(a[j], x.y, M[k].y, P[k].y);
                    a[j] = Acc<EXACT>::step(a[j], x.z, M[k].z, P[k].z);
                    a[j] = Acc<EXACT>::step(a[j], x.w, M[k].w, P[k].w);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < UG_FB; j++)
            if (j < nd) tr[j * 65 + lane] = gau_to_int((double)a[j], S.f, S.distfloor, mixw);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int32_t mode = 0;
        if (mine && act) {
            if (ci_scr >= d.thresh) mode = 1;
            else mode = (bi == S3A_NO_BSTIDX || ut != d.frame - 1) ? 3 : 2;
        }
        int32_t score = S3A_LOGPROB_ZERO, bs = S3A_LOGPROB_ZERO, bidx = S3A_NO_BSTIDX;
        const int32_t *row = tr + c * 65 + sl * CP;
        if (mode == 1) {
            for (int32_t cc = 0; cc < nc; cc++) {
                const int32_t v = row[cc];
                score = la(score, v);
                if (v > bs) { bs = v; bidx = cc; }
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
            d.sen_act[sen] = 0;
            if (mode != 0) {
                d.scr[sen] = score;
                rbest = score;
                if (mode == 1) { d.bstidx[sen] = bidx; d.bstscr[sen] = bs; d.updatetime[sen] = d.frame; rns = 1; rng = nc; }
                else if (mode == 2) {
                    if (d.is_skip) { d.bstidx[sen] = bidx; d.bstscr[sen] = bs; d.updatetime[sen] = d.frame; }
                    rng = 1;
                }
            }
        }
        for (int32_t o = CP; o < 64; o <<= 1) {
            rbest = max(rbest, __shfl_xor(rbest, o, 64));
            rns += __shfl_xor(rns, o, 64);
            rng += __shfl_xor(rng, o, 64);
        }
        if (sl == 0 && c < nd) { red[wave][z0 + c][0] = rbest; red[wave][z0 + c][1] = rns; red[wave][z0 + c][2] = rng; }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (tid < n && dec[tid].active && (tid / UG_FB) % (int32_t)gridDim.y == (int32_t)blockIdx.y) {
        int32_t *gp = dec[tid].gpart;
        gp[blockIdx.x] = max(max(red[0][tid][0], red[1][tid][0]), max(red[2][tid][0], red[3][tid][0]));
        gp[S.gp_n + blockIdx.x] = red[0][tid][1] + red[1][tid][1] + red[2][tid][1] + red[3][tid][1];
        gp[2 * S.gp_n + blockIdx.x] = red[0][tid][2] + red[1][tid][2] + red[2][tid][2] + red[3][tid][2];
    }
}

/*
 * LOOK-AHEAD SCORING: every senone of the coming K frames of every lane in ONE pass over the model.
 *
 * A Gaussian's value does not depend on the search (only WHICH scores are used does: the active-senone mask and the
 * CI gate of approx_cont_mgau_frame_eval select among them), and the lanes' features are resident in HBM, so the
 * Gaussians are taken out of the frame's chain of launches altogether: every K frames ONE launch of the
 * model-stationary kernel (k_score_frames, s3a_device.hip: a lane of the wave keeps its Gaussian in VGPRs, the frames
 * stream past from LDS, eight at a time, and the senone's ordered log-add runs on transposed values with the log-add
 * table in LDS) scores n_lanes x K (lane, frame) slots for ALL senones -- a batch of 128 lanes x 8 frames is the
 * same 1024 frames per model pass as a 10 s utterance -- into the lanes' window buffers (win: score per senone, winb:
 * the best component, mgau_eval's update_best_id).  Per frame ku_select then applies the gate to the ACTIVE senones
 * (the very decisions of d_gated_frame, s3a_gated.h) and patches the few back-off scores in place; ku_hmm_eval reads
 * the window row.  The model is read once per K frames of all lanes (per-frame passes: once per frame and 32 lanes).
 * Bit for bit the scores of ku_gated / ku_gated_cd_multi (tests/test_gpu_uttdec.py, test_gpu_dropin.py).
 */
#define UW_FB 8
struct UwGroup {
    const float *feat;      /* the lane's features at the group's first frame */
    int32_t *win;           /* the lane's window rows from the group's first row */
    uint8_t *winb;
    int32_t nv, pad;        /* valid frames of the group (0: lane finished / idle) */
};

template <int CP, bool EXACT, bool TAB_LDS, int NT>
__global__ void __launch_bounds__(NT)
ku_score_window(const ULane *__restrict__ lanes, UShared S, int32_t n_lanes, int32_t f0, int32_t K, int32_t fpc,
                int32_t n_chunks, int32_t n_tiles, const UwGroup *__restrict__ gdesc, int32_t n_g)
{
    /* (gdesc: n_g groups of up to 8 consecutive frames of ANY utterances, written by the host -- the whole call's frames scored before
     * the search starts, rows in one buffer (ku_frames, SCORES FIRST); NULL: the lanes' coming K frames into their window rows) */
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int DP = D4MAIN * 4;
    typedef typename Acc<EXACT>::T acc_t;
    /* XCD-aware decode of blockIdx.x (as k_score_frames): all chunks of a Gaussian tile run on XCD tile % 8 */
    const int32_t b = blockIdx.x, xcd = b & 7, r = b >> 3;
    const int32_t chunk = r % n_chunks, tile = (r / n_chunks) * 8 + xcd;
    if (tile >= n_tiles) return;
    const int32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t g = tile * NT + tid;
    const int32_t q0 = chunk * fpc, total = gdesc ? n_g * UW_FB : n_lanes * K;
    const int32_t nslot = min(fpc, total - q0), ngrp = (nslot + UW_FB - 1) / UW_FB;

    float *xs = (float *)smem;
    size_t off = (size_t)fpc * DP * sizeof(float);
    int32_t *tr = (int32_t *)(smem + off) + wave * (UW_FB * 65);
    off += (size_t)(NT / 64) * UW_FB * 65 * sizeof(int32_t);
    off = (off + 15) & ~(size_t)15;
    UwGroup *grp = (UwGroup *)(smem + off);
    off += (size_t)(fpc / UW_FB) * sizeof(UwGroup);
    off = (off + 15) & ~(size_t)15;
    uint16_t *tab_s = (uint16_t *)(smem + off);

    /* the chunk's groups: 8 consecutive frames of one lane each (K is a multiple of 8) */
    int32_t *s_any = (int32_t *)xs;             /* (the feature area is filled after the early exit) */
    if (tid == 0) *s_any = 0;
    __syncthreads();
    if (tid < ngrp && gdesc) {
        const UwGroup gr = gdesc[q0 / UW_FB + tid];
        grp[tid] = gr;
        if (gr.nv) *s_any = 1;
    }
    else if (tid < ngrp) {
        const int32_t q = q0 + tid * UW_FB, z = q / K, j0 = q - z * K;
        UwGroup gr;
        gr.feat = NULL; gr.win = NULL; gr.winb = NULL; gr.nv = 0; gr.pad = 0;
        const UCtx *cx = S.ctx_all + z;
        const int32_t fl = f0 + *S.fgbase - cx->f0 + j0;        /* the lane's own frame (its utterance began at a window's first frame) */
        if (cx->active && fl >= 0) {
            const int32_t left = cx->nfr - fl;
            if (left > 0) {
                gr.nv = min(left, UW_FB);
                gr.feat = cx->feat + (size_t)fl * DP;
                gr.win = lanes[z].win + (size_t)j0 * S.n_sen;
                gr.winb = lanes[z].winb + (size_t)j0 * S.n_sen;
            }
        }
        grp[tid] = gr;
        if (gr.nv) *s_any = 1;
    }
    __syncthreads();
    if (!*s_any) return;                        /* every lane of the chunk has finished its utterance */
    __syncthreads();

    if (TAB_LDS) {
        const uint4 *src = (const uint4 *)S.tab16;
        uint4 *dst = (uint4 *)tab_s;
        const int32_t n16 = (int32_t)((S.tab_size * 2 + 15) >> 4);
        for (int32_t i = tid; i < n16; i += NT) dst[i] = src[i];
    }
    for (int32_t i = tid; i < ngrp * UW_FB * (DP / 4); i += NT) {      /* 16-byte pieces: rows are DP floats, zero padded */
        const int32_t slot = i / (DP / 4), k = i - slot * (DP / 4), gi = slot / UW_FB, j = slot - gi * UW_FB;
        const UwGroup &gr = grp[gi];
        ((float4 *)xs)[i] = j < gr.nv ? ((const float4 *)(gr.feat + (size_t)j * DP))[k] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    float4 M[D4MAIN], P[D4MAIN];
#pragma unroll
    for (int k = 0; k < D4MAIN; k++) { M[k] = S.mean4[(size_t)k * S.Gpad + g]; P[k] = S.prec4[(size_t)k * S.Gpad + g]; }
    const acc_t lrd_g = (acc_t)S.lrd[g];
    const int32_t mixw = S.mixw[g];
    const int32_t c = lane & (CP - 1), sl = lane / CP, sen = g / CP;
    const int32_t nc = sen < S.n_sen ? (int32_t)S.ncomp[sen] : 0;
    LogAdd la;
    la.tab = TAB_LDS ? tab_s : S.tab16; la.size = S.tab_size; la.zero = S.lm_zero;
    __syncthreads();
    const float4 *xs4 = (const float4 *)xs;
    for (int32_t gi = 0; gi < ngrp; gi++) {
        const int32_t nv = grp[gi].nv;
        if (nv == 0) continue;
        const int32_t fr = gi * UW_FB;
        acc_t a[UW_FB];
#pragma unroll
        for (int j = 0; j < UW_FB; j++) a[j] = lrd_g;
#pragma unroll
        for (int k = 0; k < D4MAIN; k++) {
#pragma unroll
            for (int j = 0; j < UW_FB; j++) {
                const float4 x = xs4[(fr + j) * D4MAIN + k];       /* wave-uniform: LDS broadcast */
                a[j] = Acc<EXACT>::step(a[j], x.x, M[k].x, P[k].x);
                a[j] = Acc<EXACT>::step(a[j], x.y, M[k].y, P[k].y);
                a[j] = Acc<EXACT>::step(a[j], x.z, M[k].z, P[k].z);
                a[j] = Acc<EXACT>::step(a[j], x.w, M[k].w, P[k].w);
            }
        }
#pragma unroll
        for (int j = 0; j < UW_FB; j++) tr[j * 65 + lane] = gau_to_int((double)a[j], S.f, S.distfloor, mixw);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int32_t *win = grp[gi].win;
        uint8_t *winb = grp[gi].winb;
        for (int j = c; j < UW_FB; j += CP) {
            int32_t score = S3A_LOGPROB_ZERO, bs = S3A_LOGPROB_ZERO, bidx = 255;
            const int32_t *row = tr + j * 65 + sl * CP;
#pragma unroll
            for (int cc = 0; cc < CP; cc++) {
                const int32_t v = row[cc];
                if (cc < nc) {                      /* (padded slots: not components of the senone) */
                    score = la(score, v);
                    if (v > bs) { bs = v; bidx = cc; }      /* update_best_id: strict >, the first maximum */
                }
            }
            if (score <= S3A_LOGPROB_ZERO) score = S3A_LOGPROB_ZERO;
            if (sen < S.n_sen && j < nv) {
                win[(size_t)j * S.n_sen + sen] = score;
                winb[(size_t)j * S.n_sen + sen] = (uint8_t)bidx;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

/* one Gaussian for one frame (the gate's best-Gaussian back-off; a few senones per frame when the gate fires at all) */
template <bool EXACT>
__device__ __forceinline__ int32_t
uw_one_gaussian(const UShared &S, int32_t g, const float *__restrict__ x)
{
    typedef typename Acc<EXACT>::T acc_t;
    acc_t a = (acc_t)S.lrd[g];
    for (int32_t k = 0; k < S.D4; k++) {
        const float4 m = S.mean4[(size_t)k * S.Gpad + g], p = S.prec4[(size_t)k * S.Gpad + g];
        const float4 xv = *(const float4 *)(x + 4 * k);
        a = Acc<EXACT>::step(a, xv.x, m.x, p.x);
        a = Acc<EXACT>::step(a, xv.y, m.y, p.y);
        a = Acc<EXACT>::step(a, xv.z, m.z, p.z);
        a = Acc<EXACT>::step(a, xv.w, m.w, p.w);
    }
    return gau_to_int((double)a, S.f, S.distfloor, S.mixw[g]);
}

/*
 * approx_cont_mgau_ci_eval + approx_cont_mgau_frame_eval on the window row of the lane's frame: the CI senones (always
 * evaluated; their maximum is the gate's reference), then per ACTIVE CD senone the three-way gate of
 * approx_cont_mgau.c:188-284 -- inside the CI beam: the full mixture (already in the row) and its best component;
 * outside: the best Gaussian of the previous frame alone if the senone was evaluated then, else the CI senone's score
 * -- with bstidx / updatetime kept as the reference keeps them (-ds skip frames included), the frame's best score and
 * the evaluation counters as columns of gpart[] (merged by ku_hmm_eval and the frame record), the mask consumed.
 * USEL_G workgroups per lane; every one works out the CI maximum for itself (n_ci_sen loads from one row).
 */
#define USEL_G 8
#ifndef G_ENTER2_MANY
#define G_ENTER2_MANY 12   /* ku_enter2's workgroups per lane from 64 lanes on (they take the calls in turn) */
#endif
#ifndef G_HIST_MANY
#define G_HIST_MANY 9      /* ku_hist_count's workgroups per (tree, lane) with many lanes (5: 440.1, 7: 445.1, 8: 440.4, 9: 447.5 k frames/s) */
#endif
/* (NT threads; workgroup bx of G takes every G-th run of NT CD senones and leaves its maxima / counters in column bx of gpart[]) */
template <bool EXACT, int NT>
__device__ __forceinline__ void
d_select(const ULane &L, const UShared &S, const UCtx *ctx, int32_t f, int32_t *row, const uint8_t *brow, int32_t bx, int32_t G,
         const uint32_t *actbits = NULL,     /* (ku_frames: the mask as bits in LDS, cleared by the caller) */
         const int32_t *dynbeam = NULL)      /* (ku_frames: -maxcdsenpf's beam of the frame as this workgroup worked it out, kf_dyn_ci_beam) */
{
    __shared__ int32_t red[3][NT / 64];
    __shared__ int32_t s_pb;
    const int32_t tid = threadIdx.x, ln = tid & 63;
    int32_t pb = INT_MIN, cig = 0;
    for (int32_t ci = tid; ci < S.n_ci_sen; ci += NT) { pb = max(pb, row[ci]); cig += (int32_t)S.ncomp[ci]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { pb = max(pb, __shfl_xor(pb, o, 64)); cig += __shfl_xor(cig, o, 64); }
    if (ln == 0) { red[0][tid >> 6] = pb; red[1][tid >> 6] = cig; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NT / 64; w++) { pb = max(pb, red[0][w]); cig += red[1][w]; }
        s_pb = pb;
        if (bx == 0) {          /* the CI phase's outputs: best CI score, senones / Gaussians evaluated */
            L.misc[5] = pb; L.misc[3] = S.n_ci_sen; L.misc[4] = cig;
        }
    }
    __syncthreads();
    pb = s_pb;
    const int32_t is_skip = (f % S.ds_ratio == 0) ? 0 : 1;
    const int32_t beam = (S.max_cd < S.n_sen - S.n_ci_sen) ? (dynbeam ? *dynbeam : L.dynbeam[0]) : (is_skip ? S.ci_pbeam_tight : S.ci_pbeam);
    const int32_t thresh = add32(pb, beam);
    LogAdd la;
    la.tab = S.tab16; la.size = S.tab_size; la.zero = S.lm_zero;
    int32_t rbest = INT_MIN, rns = 0, rng = 0;
    /* (four runs of NT senones per turn: what a senone's decision reads -- its mark, its CI senone's score, its own score and best
     * component, last frame's best component and when that was -- is asked for together for all four, coalesced; one senone per turn
     * was a chain of four round trips per turn) */
    for (int32_t s0 = S.n_ci_sen + bx * NT; s0 < S.n_sen; s0 += 4 * G * NT) {
        int32_t senq[4], ciq[4], cisq[4], rowq[4], obq[4], utq[4];
        uint8_t actq[4], nbq[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            senq[u] = s0 + u * G * NT + tid;
            const bool in = senq[u] < S.n_sen;
            const int32_t sq = in ? senq[u] : S.n_ci_sen;            /* (loads unconditional, a thread past the end reads a senone that exists: a load under a
                                                                       * condition is a branch with its own wait inside) */
            actq[u] = actbits ? (uint8_t)((LM(actbits)[sq >> 5] >> (sq & 31)) & 1u) : L.sen_act[sq];
            if (!in) actq[u] = 0;
            ciq[u] = GMC(S.cd2cisen)[sq];
            senq[u] = sq;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            cisq[u] = GMC(row)[ciq[u]]; rowq[u] = GMC(row)[senq[u]]; nbq[u] = GMC(brow)[senq[u]];
            obq[u] = GMC(L.bstidx)[senq[u]]; utq[u] = GMC(L.updatetime)[senq[u]];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (!actq[u]) continue;
            const int32_t sen = senq[u];
            if (!actbits) L.sen_act[sen] = 0;           /* the mask is consumed: clean for the next frame's marks */
            const int32_t ci_scr = cisq[u];
            if (ci_scr >= thresh) {                     /* full evaluation */
                const int32_t bi = (int32_t)nbq[u];
                GM(L.bstidx)[sen] = bi == 255 ? S3A_NO_BSTIDX : bi;
                GM(L.updatetime)[sen] = f;
                rbest = max(rbest, rowq[u]); rns++; rng += (int32_t)GMC(S.ncomp)[sen];
                continue;
            }
            const int32_t bi = obq[u], ut = utq[u];
            if (bi == S3A_NO_BSTIDX || ut != f - 1) {   /* the CI senone stands in */
                GM(row)[sen] = ci_scr;
                rbest = max(rbest, ci_scr);
                continue;
            }
            /* the best Gaussian of the previous frame alone */
            const int32_t v = uw_one_gaussian<EXACT>(S, sen * S.CP + bi, ctx->feat + (size_t)f * S.D4 * 4);
            int32_t score = la(S3A_LOGPROB_ZERO, v);
            if (score <= S3A_LOGPROB_ZERO) score = S3A_LOGPROB_ZERO;
            row[sen] = score;
            rbest = max(rbest, score); rng++;
            if (is_skip) { L.bstidx[sen] = v > S3A_LOGPROB_ZERO ? bi : S3A_NO_BSTIDX; L.updatetime[sen] = f; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        rbest = max(rbest, __shfl_xor(rbest, o, 64)); rns += __shfl_xor(rns, o, 64); rng += __shfl_xor(rng, o, 64);
    }
    __syncthreads();
    if (ln == 0) { red[0][tid >> 6] = rbest; red[1][tid >> 6] = rns; red[2][tid >> 6] = rng; }
    __syncthreads();
    if (tid == 0 && bx < S.gp_n) {
        for (int w = 1; w < NT / 64; w++) { rbest = max(rbest, red[0][w]); rns += red[1][w]; rng += red[2][w]; }
        L.gpart[bx] = rbest;
        L.gpart[S.gp_n + bx] = rns;
        L.gpart[2 * S.gp_n + bx] = rng;
    }
    /* (the columns beyond this grid stay neutral: written once at init) */
}

template <bool EXACT>
__global__ void __launch_bounds__(256)
ku_select(const ULane *__restrict__ lanes, UShared S, int32_t fg, int32_t K)
{
    LANE;
    d_select<EXACT, 256>(L, S, ctx, f, L.win + (size_t)(f % K) * S.n_sen, L.winb + (size_t)(f % K) * S.n_sen, blockIdx.x, gridDim.x);
}

/* the members of the composite senones wanted in this frame join the mask (before ku_select) */
__global__ void __launch_bounds__(256)
ku_comsen_mark(const ULane *__restrict__ lanes, UShared S, int32_t fg)
{
    LANE;
    d_comsen_wave<false>(S.n_cs, L.cs_need, f, S.cs_off, S.cs_list, L.sen_act, (const int32_t *)NULL, (int32_t *)NULL,
                         (int32_t)blockIdx.x * 256 + (int32_t)(threadIdx.x & ~63));
}

/* ---- approx_compute_dyn_ci_pbeam (approx_cont_mgau.c:303-357, -maxcdsenpf): the CI senones in descending score order,
 * the active CD senones each of them stands for counted along the way; the beam is cut where the count passes the cap.
 * (Ties need no order: the cut is a score.)  One workgroup per lane, between the CI and the CD scoring launches. ---- */
#define UDB_MAXCI 1024
__global__ void __launch_bounds__(1024)
ku_dyn_ci_beam(const ULane *__restrict__ lanes, UShared S, int32_t fg)
{
    __shared__ int32_t s_occ[UDB_MAXCI], s_scr[UDB_MAXCI], s_ord[UDB_MAXCI], s_cut;
    LANE;
    const int32_t n_ci = S.n_ci_sen, tid = threadIdx.x;
    for (int32_t c = tid; c < n_ci; c += 1024) { s_occ[c] = 0; s_scr[c] = SCR_ROW[c]; }
    if (tid == 0) s_cut = INT_MAX;
    __syncthreads();
    for (int32_t s = n_ci + tid; s < S.n_sen; s += 1024)
        if (L.sen_act[s]) atomicAdd(&s_occ[S.cd2cisen[s]], 1);
    __syncthreads();
    for (int32_t c = tid; c < n_ci; c += 1024) {
        const int32_t v = s_scr[c];
        int32_t r = 0;
        for (int32_t c2 = 0; c2 < n_ci; c2++) { const int32_t v2 = s_scr[c2]; r += (v2 > v || (v2 == v && c2 < c)) ? 1 : 0; }
        s_ord[r] = c;
    }
    __syncthreads();
    const int32_t pbest = s_scr[s_ord[0]];
    for (int32_t r = tid; r < n_ci; r += 1024) {
        int32_t total = 0;
        for (int32_t r2 = 0; r2 <= r; r2++) total += s_occ[s_ord[r2]];
        if (total > S.max_cd) atomicMin(&s_cut, r);          /* the first rank at which the count passes the cap */
    }
    __syncthreads();
    if (tid == 0) {
        int32_t beam = S.ci_pbeam;
        if (s_cut != INT_MAX) {
            /* (the reference's loop runs while the score is above pbest + ci_pbeam: a cut beyond that is none) */
            const int32_t v = s_scr[s_ord[s_cut]];
            bool in_beam = true;
            for (int32_t r = 0; r <= s_cut && in_beam; r++) in_beam = s_scr[s_ord[r]] > add32(pbest, S.ci_pbeam);
            if (in_beam) beam = v - pbest;
        }
        if (f % S.ds_ratio != 0) beam = (int32_t)((float)beam * S.tighten);
        L.dynbeam[0] = beam;
    }
}

/* ---- the scores of the composite senones wanted in this frame (after the scoring kernels) ---- */
__global__ void __launch_bounds__(256)
ku_comsen_max(const ULane *__restrict__ lanes, UShared S, int32_t fg)
{
    LANE;
    d_comsen_wave<true>(S.n_cs, L.cs_need, f, S.cs_off, S.cs_list, (uint8_t *)NULL, SCR_ROW, L.cs_val,
                        (int32_t)blockIdx.x * 256 + (int32_t)(threadIdx.x & ~63));
}

/* ---- phoneme look-ahead (-pheurtype 1..3, -pl_window, -pl_beam) ----
 * The reference scores the CI senones -pl_window frames ahead (gmm_compute_lv1 into ascr->cache_ci_senscr, srch.c:738-741,
 * :813-817), sums a per-phone figure over the window into pl->phn_heur_list before every frame's search
 * (srch_TST_compute_heuristic, srch_time_switch_tree.c:755-775 -> pl_computePhnHeur, fast_algo_struct.c:219-300) and lets
 * lextree_hmm_propagate_non_leaves (lextree.c:1443-1486) drop transitions whose score + heuristic of the child's phone
 * falls below the running maximum + pl_beam.  A lane's features are resident, so here BOTH tables are made for the whole
 * utterance in two launches at its beginning: ku_ci_ahead = approx_cont_mgau_ci_eval for every frame (raw scores, as the
 * cache holds them); ku_phn_heur = pl_computePhnHeur for every frame t over the frames [t, min(t + window, n_frames)).
 */
template <bool EXACT>
__global__ void __launch_bounds__(256)
ku_ci_ahead(const ULane *__restrict__ lanes, UShared S, const int32_t *__restrict__ sub)
{
    const ULane &L = lanes[sub ? sub[blockIdx.z] : (int32_t)blockIdx.z];
    const UCtx *ctx = L.ctx;
    const int32_t cf = blockIdx.y;
    if (cf >= ctx->nfr) return;
    if ((int32_t)(blockIdx.x * 256) >= S.n_ci_sen * S.CP) return;
    const float *x = ctx->feat + (size_t)cf * S.D4 * 4;
    int32_t *tmp = L.ph_scratch;
#define KU_AHEAD_ARGS S.mean4, S.prec4, S.lrd, S.mixw, S.tab16, S.tab_size, S.lm_zero, S.f, S.distfloor, x, S.D4, S.CP, S.Gpad, 0,   \
        S.n_ci_sen, 1, S.ncomp, S.cd2cisen, (const uint8_t *)NULL, L.ci_all + (size_t)cf * S.n_ci_sen, 0, (const int32_t *)NULL, 0, cf, 0,  \
        tmp, tmp + S.n_ci_sen, tmp + 2 * S.n_ci_sen, tmp + 3 * S.n_ci_sen, 5, (uint8_t *)NULL, (int32_t *)NULL, 0
    if (S.D4 == D4MAIN) d_gated_frame<EXACT, D4MAIN>(KU_AHEAD_ARGS, blockIdx.x);
    else d_gated_frame<EXACT, 0>(KU_AHEAD_ARGS, blockIdx.x);
}

/* NO_UFLOW_ADD, fast_algo_struct.c:206-216 (int32 wrap-around, then the underflow patch) */
__device__ __forceinline__ int32_t
ph_add(int32_t a, int32_t b)
{
    const int32_t c = add32(a, b);
    return (c > 0 && a < 0 && b < 0) ? INT_MIN : c;
}

#define PH_T 64
#define PH_MAXCI 96
__global__ void __launch_bounds__(PH_T)
ku_phn_heur(const ULane *__restrict__ lanes, UShared S, const int32_t *__restrict__ sub)
{
    const ULane &L = lanes[sub ? sub[blockIdx.z] : (int32_t)blockIdx.z];
    const int32_t nfr = L.ctx->nfr, t = blockIdx.x * PH_T + threadIdx.x, n_cis = S.n_ci_sen, nci = S.n_ci;
    __shared__ int32_t s_ph[PH_MAXCI][PH_T + 1];
    if (t >= nfr) return;
    const int16_t *s2c = S.sen2cimap;
    for (int32_t p = 0; p < nci; p++) s_ph[p][threadIdx.x] = 0;         /* (every CI senone's phone: all CI phones) */
    const int32_t t_end = min(t + S.pl_window, nfr);
#define PH(p) s_ph[p][threadIdx.x]
    for (int32_t i = t; i < t_end; i++) {
        const int32_t *row = L.ci_all + (size_t)i * n_cis;
        int32_t cur = 0, var = INT_MIN;
        if (S.pheurtype == 1) {                 /* sum over the window of the phone's best senone */
            for (int32_t j = 0; j < n_cis; j++) {
                const int32_t v = row[j];
                if (var < v) var = v;
                cur = s2c[j];
                if (cur != s2c[j + 1]) { PH(cur) = ph_add(PH(cur), var); var = INT_MIN; }
            }
        }
        else if (S.pheurtype == 2) {
            /* "sum of averages" as the reference computes it: the phone's running sum starts from MAX_NEG_INT32, so its
             * first addition overflows -- undefined in C; the pinned reference build (gcc -O2) drops NO_UFLOW_ADD's
             * patch where an operand is that constant and keeps the wrapped sum: that is what is restated here */
            for (int32_t j = 0; j < n_cis; j++) {
                var = var == INT_MIN && (j == 0 || s2c[j - 1] != s2c[j]) ? add32(row[j], INT_MIN) : ph_add(row[j], var);
                cur = s2c[j];
                if (cur != s2c[j + 1]) { var /= S.ne; PH(cur) = ph_add(PH(cur), var); var = INT_MIN; }
            }
        }
        else {                                  /* type 3, with its "dangerous hack" as written */
            for (int32_t j = 0; j < n_cis; j++) {
                const int32_t v = row[j];
                if (cur == 0 || cur != s2c[j - 1]) PH(cur) = ph_add(PH(cur), v);
                cur = s2c[j];
                if (var < v) var = v;
                if (s2c[j] != s2c[j + 1]) { PH(cur) = ph_add(PH(cur), var); var = INT_MIN; }
            }
        }
    }
    for (int32_t p = 0; p < nci; p++) L.heur_all[(size_t)t * nci + p] = PH(p);
#undef PH
}

/* per frame, behind the thresholds: the heuristic threshold of every propagating HMM by list position -- the running maximum
 * over the active list (per tree: kbc->maxNewHeurScore is reset by every lextree_hmm_propagate_non_leaves call) of
 * max over children (out + (prob(child) - prob) + phn_heur[ci(child)]), plus pl_beam (lextree.c:1443-1462).  One workgroup
 * per (tree, lane); an HMM propagates when it is no leaf and its exit score reaches the phone threshold (this kernel: the phone
 * threshold never below the HMM threshold, so such an HMM is never cleared first; otherwise ku_weak_heur). */
__global__ void __launch_bounds__(1024)
ku_heur_thresh(const ULane *__restrict__ lanes, UShared S, int32_t fg)
{
    LANE;
    const int32_t t = blockIdx.x, na = nact_cur[t], b = S.node_base[t], tid = threadIdx.x;
    if (na == 0) return;
    int32_t th, pth;
    {
        int32_t bh, bw, n, wth;
        (void)frame_thresholds(L.best, nact_cur, S.T, frame_beams(S, f), L.hbin, bh, bw, n, th, pth, wth);
    }
    const int32_t *heur = L.heur_all + (size_t)f * S.n_ci;
    const int32_t *act = L.act[cur];
    __shared__ int32_t s_w[16], s_carry;
    if (tid == 0) s_carry = INT_MIN;
    __syncthreads();
    for (int32_t i0 = 0; i0 < na; i0 += 1024) {
        const int32_t i = i0 + tid;
        int32_t m = INT_MIN;
        if (i < na) {
            const int32_t p = act[b + i];
            const int32_t po = L.outs[NSV(p)];
            if (S.wid[p] < 0 && po >= pth) {
                const int32_t pp = S.prob[p];
                for (int32_t q = S.child_off[p]; q < S.child_off[p + 1]; q++) {
                    const int32_t c = S.child[q];
                    m = max(m, add32(add32(po, add32(S.prob[c], -pp)), heur[S.node_ci[c]]));
                }
            }
        }
        /* inclusive running maximum over the chunk: wave scan, then the waves' totals */
        int32_t x = m;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int32_t y = __shfl_up(x, o, 64); if ((tid & 63) >= o) x = max(x, y); }
        if ((tid & 63) == 63) s_w[tid >> 6] = x;
        __syncthreads();
        int32_t pre = s_carry;
        for (int32_t w = 0; w < (tid >> 6); w++) pre = max(pre, s_w[w]);
        x = max(x, pre);
        if (i < na) L.hth_pos[b + i] = add32(x, S.pl_beam);
        __syncthreads();
        if (tid == 1023) s_carry = x;
        __syncthreads();
    }
}

/*
 * -pheurtype together with a phone threshold BELOW the HMM threshold (-ptranskip frames: bestwordscore + -wbeam; -pbeam wider
 * than -beam): the two dependencies meet.  A "weak" HMM (under the HMM beam, exit score over the phone threshold) propagates only
 * when a parent EARLIER in the list entered it first (d_dec_weak) -- and with the look-ahead that entry must also pass the
 * heuristic threshold at the parent's list position, which is the running maximum over the HMMs that propagate up to there,
 * surviving weak ones included (lextree.c:1424-1458: the clear at the HMM's turn, then maxNewHeurScore).  Every dependency
 * points to a smaller list position of the same tree, so one workgroup per (tree, lane):
 *   1. the running maximum over the HMMs that propagate whatever happens (not weak) -> hth_pos (raw), and the weak HMMs
 *      compacted in list order with their own contribution (position, max over children);
 *   2. ONE wave walks the weak HMMs in list order, its lanes over the HMM's parents: entered early by a propagating parent
 *      whose threshold (hth_pos at the parent's position, or the best surviving weak HMM at or before it) the entry passes
 *      -> propf stamp, and the survivor joins the list of (position, running maximum);
 *   3. hth_pos = max(raw, survivors at or before the position) + pl_beam.
 * Replaces ku_weak + ku_heur_thresh when both options are on (in a frame with the usual geometry no HMM is weak and 2. is empty).
 */
/* (the body: NT threads of one workgroup for tree t of the lane; best = the trees' best scores of the frame -- the lane's array, or
 * ku_frames' copy in LDS; s_w / s_c: NT / 64 words each, s_x: 3 words of the caller's LDS) */
template <int NT>
__device__ void
d_weak_heur_t(const ULane &L, const UShared &S, int32_t f, int32_t cur, const int32_t *nact_cur, const int32_t *best, int32_t t,
              int32_t *s_w, int32_t *s_c, int32_t *s_x)
{
    const int32_t na = nact_cur[t], b = S.node_base[t], tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (na == 0) return;
    int32_t th, pth;
    {
        int32_t bh, bw, n, wth;
        (void)frame_thresholds(best, nact_cur, S.T, frame_beams(S, f), L.hbin, bh, bw, n, th, pth, wth);
    }
    const bool weak_frame = pth < th;
    const int32_t *heur = L.heur_all + (size_t)f * S.n_ci;
    const int32_t *act = L.act[cur];
    int32_t *wl_v = L.exits + b, *wl_pos = L.exits + (size_t)S.N + b, *wl_H = L.exits + 2 * (size_t)S.N + b;   /* (free between the histogram and the scan) */
    int32_t *sv_pos = L.hth_pos + (size_t)S.N + b, *sv_max = L.hth_pos + 2 * (size_t)S.N + b;
    int32_t &s_carry = s_x[0], &s_nw = s_x[1], &s_ns = s_x[2];
    __syncthreads();
    if (tid == 0) { s_carry = INT_MIN; s_nw = 0; s_ns = 0; }
    __syncthreads();
    for (int32_t i0 = 0; i0 < na; i0 += NT) {
        const int32_t i = i0 + tid;
        int32_t m = INT_MIN, p = -1;
        bool wk = false;
        if (i < na) {
            p = act[b + i];
            const int32_t po = L.outs[NSV(p)];
            if (S.wid[p] < 0 && po >= pth) {
                const int32_t pp = S.prob[p];
                for (int32_t q = S.child_off[p]; q < S.child_off[p + 1]; q++) {
                    const int32_t c = S.child[q];
                    m = max(m, add32(add32(po, add32(S.prob[c], -pp)), heur[S.node_ci[c]]));
                }
                wk = weak_frame && L.bests[NSV(p)] < th;
            }
        }
        const unsigned long long wm = __ballot(wk);
        int32_t x = wk ? INT_MIN : m;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int32_t y = __shfl_up(x, o, 64); if (lane >= o) x = max(x, y); }
        if (lane == 63) s_w[wv] = x;
        if (lane == 0) s_c[wv] = __popcll(wm);
        __syncthreads();
        int32_t pre = s_carry, at = s_nw;
        for (int32_t w = 0; w < wv; w++) { pre = max(pre, s_w[w]); at += s_c[w]; }
        x = max(x, pre);
        if (i < na) L.hth_pos[b + i] = x;
        if (wk) {
            const int32_t k = at + __popcll(wm & ((1ull << lane) - 1ull));
            wl_v[k] = p; wl_pos[k] = i; wl_H[k] = m;
        }
        __syncthreads();
        if (tid == NT - 1) { s_carry = x; s_nw = at + s_c[NT / 64 - 1]; }
        __syncthreads();
    }
    const int32_t nw = s_nw;
    if (nw == 0) {
        for (int32_t i = tid; i < na; i += NT) L.hth_pos[b + i] = add32(L.hth_pos[b + i], S.pl_beam);
        return;
    }
    __threadfence_block();
    if (wv == 0) {
        int32_t ns = 0, smax = INT_MIN;
        for (int32_t k = 0; k < nw; k++) {
            const int32_t v = wl_v[k], j = wl_pos[k];
            const int32_t in0 = L.sc[NSV(v)], hv = heur[S.node_ci[v]];
            bool early = false;
            for (int32_t q0 = S.par_off[v], q_hi = S.par_off[v + 1]; q0 < q_hi && !early; q0 += 64) {
                bool pass = false;
                const int32_t q = q0 + lane;
                if (q < q_hi) {
                    const int32_t g = S.par[q];
                    const int32_t pp = L.pos[PPX(g)];
                    if (L.posf[PPX(g)] == f && pp < j) {
                        const int32_t po = L.outs[NSV(g)];
                        if (po >= pth && (L.bests[NSV(g)] >= th || ((volatile int32_t *)L.propf)[g] == f)) {
                            const int32_t nsc = add32(po, add32(S.prob[v], -S.prob[g]));
                            if (nsc >= th && nsc > in0) {
                                int32_t hm = L.hth_pos[b + pp];
                                if (ns > 0 && ((volatile int32_t *)sv_pos)[0] <= pp) {       /* the last survivor at or before pp */
                                    int32_t lo = 0, hi = ns - 1;
                                    while (lo < hi) {
                                        const int32_t mid = (lo + hi + 1) >> 1;
                                        if (((volatile int32_t *)sv_pos)[mid] <= pp) lo = mid; else hi = mid - 1;
                                    }
                                    hm = max(hm, ((volatile int32_t *)sv_max)[lo]);
                                }
                                pass = add32(nsc, hv) >= add32(hm, S.pl_beam);
                            }
                        }
                    }
                }
                early = __any(pass);
            }
            if (early) {
                smax = max(smax, wl_H[k]);
                if (lane == 0) {
                    ((volatile int32_t *)L.propf)[v] = f;
                    ((volatile int32_t *)sv_pos)[ns] = j; ((volatile int32_t *)sv_max)[ns] = smax;
                }
                ns++;
                __threadfence_block();
            }
        }
        if (lane == 0) s_ns = ns;
    }
    __syncthreads();
    const int32_t ns = s_ns;
    for (int32_t i = tid; i < na; i += NT) {
        int32_t hm = L.hth_pos[b + i];
        if (ns > 0 && sv_pos[0] <= i) {
            int32_t lo = 0, hi = ns - 1;
            while (lo < hi) {
                const int32_t mid = (lo + hi + 1) >> 1;
                if (sv_pos[mid] <= i) lo = mid; else hi = mid - 1;
            }
            hm = max(hm, sv_max[lo]);
        }
        L.hth_pos[b + i] = add32(hm, S.pl_beam);
    }
}

__global__ void __launch_bounds__(1024)
ku_weak_heur(const ULane *__restrict__ lanes, UShared S, int32_t fg)
{
    LANE;
    __shared__ int32_t s_w[16], s_c[16], s_x[3];
    d_weak_heur_t<1024>(L, S, f, cur, nact_cur, L.best, blockIdx.x, s_w, s_c, s_x);
}

/* ---- lextree_hmm_eval ---- */
template <int EB, int NE>
__global__ void __launch_bounds__(EB)
ku_hmm_eval(const ULane *__restrict__ lanes, UShared S, int32_t fg)
{
    LANE;
    const int32_t t = blockIdx.y, na = nact_cur[t];
    for (int32_t vb = blockIdx.x; vb * EB < na; vb += gridDim.x) {
        d_dec_hmm_eval<EB, NE>(S.node_base, L.act[cur], L.nact[cur], S.N, S.n_tmat, S.ssid, S.tmatid, S.wid, S.comp, S.tp,
                           S.sseq, S.comsseq, S.cs_off, S.cs_list, S.cs_wt, SCR_ROW, L.misc, L.sc, L.hist, L.outs, L.outh,
                           L.bests, L.best, f, (const int32_t *)NULL /* ku_hist_count stamps */, S.psof, L.pstamp, L.gpart, S.gp_n, L.poswid, L.posout,
                           vb, t, L.cs_val, S.node4);
        __syncthreads();
    }
}

/* after the evaluation: the histogram bins when the frame holds more than 1.5 x -maxhmmpf HMMs (lextree_hmm_histbin);
 * otherwise the thresholds are final and the HMMs that can propagate stamp their children's parent sets (d_dec_stamp) */
/* The stamps of tree t's HMMs that can propagate (d_dec_stamp) AND the stamped sets appended to the lane's list of the frame, each
 * once (claim[set] = the frame number, exchanged: whoever finds an older frame there lists the set; the order does not matter,
 * ku_resolve_plist visits the members of every listed set).  i0 / stride: this workgroup's first list position and the step to
 * its next, uniform over the workgroup. */
__device__ __forceinline__ void
d_stamp_and_list(const ULane &L, const UShared &S, int32_t cur, int32_t t, int32_t na, int32_t pth, int32_t f, int32_t i0, int32_t stride, int32_t list_sets)
{
    const int32_t b = S.node_base[t];
    int32_t *cnt = &L.pcnt[f & 1];
    for (int32_t ib = i0; ib < na; ib += stride) {
        const int32_t i = ib + (int32_t)threadIdx.x;
        int32_t u = -1;
        bool prop = false;
        if (i < na) { u = L.act[cur][b + i]; prop = L.outs[NSV(u)] >= pth; }
        if (prop)
            for (int32_t q = S.psof_off[u], q_hi = S.psof_off[u + 1]; q < q_hi; q++) {
                const int32_t ps = S.psof[q];
                L.pstamp8[ps] = ps_val<uint8_t>(f);
                if (list_sets && atomicExch(&L.claim[ps], f) != f) L.plist[atomicAdd(cnt, 1)] = ps;     /* the first HMM to stamp the set lists it (only when ku_resolve_plist follows) */
            }
    }
}

/* what ku_hist_sort does for a lane (below): the histogram beam, the lists reordered, the stamps of such a frame */
template <int NT>
__device__ __forceinline__ void
d_hist_sort_lane(const ULane &L, const UShared &S, const FrameBeams &bm, const int32_t *nact_cur, int32_t cur, int32_t f, int32_t list_sets)
{
    for (int32_t t = 0; t < S.T; t++) {
        const int32_t hb = d_dec_hist_sort_t<NT>(S.node_base, L.act[cur], L.nact[cur], S.T, bm, L.exits + S.N, L.exits, L.hbin,
                                           L.pos, -1, NBIN, t, 0);
        __syncthreads();
        if (hb <= 0) {
            int32_t th, pth;
            frame_thresholds_hb(L.best, S.T, bm, hb, th, pth);
            d_stamp_and_list(L, S, cur, t, nact_cur[t], pth, f, 0, NT, list_sets);
        }
        __syncthreads();
    }
}

/* own_sort: the frame's histogram sort has no launch of its own -- almost no frame needs it (37 of 1.15 M in the bench), and with
 * several engines on the chip a launch that only finds that out still waits its turn (57 us on average in the four-engine bench) --:
 * in a frame under the histogram beam the LAST workgroup of the lane to finish its bins (a counter in the lane's context, release
 * / acquire around it) runs the sort; in every other frame nothing is counted at all */
__global__ void __launch_bounds__(DBLOCK)
ku_hist_count(const ULane *__restrict__ lanes, UShared S, int32_t fg, int32_t own_sort, int32_t list_sets)
{
    LANE;
    __shared__ int32_t s_last;
    const int32_t t = blockIdx.y, na = nact_cur[t];
    const bool has_work = (int32_t)blockIdx.x * DBLOCK < na;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) L.pcnt[(f + 1) & 1] = 0;       /* (the coming frame's list) */
    const FrameBeams bm = frame_beams(S, f);
    int32_t n = 0;
    for (int32_t k = 0; k < S.T; k++) n += nact_cur[k];
    if (n > bm.maxhmmpf + (bm.maxhmmpf >> 1)) {
        if (has_work)
            for (int32_t vb = blockIdx.x; vb * DBLOCK < na; vb += gridDim.x) {
                d_dec_hist_count(S.node_base, L.act[cur], L.nact[cur], S.T, bm, L.best, L.bests, L.exits + S.N, L.hbin, -1, 0, 1,
                                 NBIN, vb, t);
                __syncthreads();
            }
        if (!own_sort) return;
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) s_last = atomicAdd(&ctx->hist_wg, 1) == (int32_t)(gridDim.x * gridDim.y) - 1;
        __syncthreads();
        if (!s_last) return;
        if (threadIdx.x == 0) ctx->hist_wg = 0;
        __threadfence();
        d_hist_sort_lane<DBLOCK>(L, S, bm, nact_cur, cur, f, list_sets);
        return;
    }
    if (!has_work) return;
    int32_t th, pth;
    frame_thresholds_hb(L.best, S.T, bm, 1, th, pth);
    d_stamp_and_list(L, S, cur, t, na, pth, f, blockIdx.x * DBLOCK, gridDim.x * DBLOCK, list_sets);
}

/* the histogram beam + the reordering of the lists (frames over 1.5 x -maxhmmpf only), then the stamps of such a frame */
template <int NT>
__global__ void __launch_bounds__(NT)
ku_hist_sort(const ULane *__restrict__ lanes, UShared S, int32_t fg, int32_t list_sets)
{
    /* ONE workgroup per lane that takes the trees in turn: the usual frame has nothing to sort, and a workgroup per (tree,
     * lane) was 768 workgroups of 1024 threads per launch whose only act is to leave -- cheap alone (7 us), 43 us on average
     * with four engines on the chip (a 1024-thread workgroup waits for half a CU to be free at once) */
    LANE;
    const FrameBeams bm = frame_beams(S, f);
    int32_t n = 0;
    for (int32_t k = 0; k < S.T; k++) n += nact_cur[k];
    if (n <= bm.maxhmmpf + (bm.maxhmmpf >> 1)) return;          /* (uniform: no histogram beam in this frame) */
    d_hist_sort_lane<NT>(L, S, bm, nact_cur, cur, f, list_sets);
}

__global__ void __launch_bounds__(SCAN_THREADS)
ku_weak(const ULane *__restrict__ lanes, UShared S, int32_t fg)
{
    LANE;
    const FrameBeams bm = frame_beams(S, f);
    if (!(bm.phone_uses_wbeam || bm.pbeam < bm.hmmbeam)) return;
    d_dec_weak(S.N, S.T, f, bm, L.best, L.nact[cur], S.node_base, L.act[cur], S.prob, S.par_off, S.par, L.pos,
               L.posf, L.sc, L.outs, L.bests, S.wid, L.hbin, L.propf, L.exits + 2 * (size_t)S.N, 0, 0);
}

/* the phoneme look-ahead's inputs of the frame (all NULL with -pheurtype 0) */
#define UHX (HEUR ? HeurArgs{ S.node_ci, L.heur_all + (size_t)f * S.n_ci, L.hth_pos } : HeurArgs{ NULL, NULL, NULL })

/* few lanes: one node per thread over all nodes (the chain is what counts) */
template <bool HEUR>
__global__ void __launch_bounds__(RSBLOCK)
ku_resolve(const ULane *__restrict__ lanes, UShared S, int32_t fg)
{
    LANE;
    d_dec_resolve<uint8_t, HEUR>(S.N, S.T, f, frame_beams(S, f), L.best, nact_cur, S.node_base, S.tree_of, S.prob,
                  S.par_off, S.par, L.pos, L.posf, L.sc, L.hist, L.outs, L.outh, L.bests, L.frame, L.turn, L.selfemit,
                  L.cnt, L.key, L.first, L.hbin, S.ps, L.pstamp8, S.rootnodes, S.n_rootnodes, L.propf, L.posout,
                  blockIdx.x, 0, UHX);
}

/* many lanes: the active HMMs by list position + a K-nodes-per-thread sweep for the rest (the number of waves counts) */
#define UR_K 8
/* ... and the not active ones from the frame's list of stamped parent sets instead of the sweep (GL one-wave workgroups per lane
 * behind the GA of the active HMMs) */
#ifndef UR_GL
#define UR_GL 192     /* one-wave workgroups per lane that walk the listed sets (one box: 128: 353.5 k, 256: 362.2 k, 512: 363.9 k frames/s; the sweep: 355.9 k) */
#endif
template <bool HEUR>
__global__ void __launch_bounds__(RSBLOCK)
ku_resolve_plist(const ULane *__restrict__ lanes, UShared S, int32_t fg)
{
    LANE;
    const int32_t GA = (int32_t)gridDim.x - UR_GL;
    if ((int32_t)blockIdx.x < GA) {
        d_dec_resolve_utt<UR_K, uint8_t, HEUR>(S.N, S.T, f, frame_beams(S, f), L.best, nact_cur, S.node_base, S.tree_of, S.prob,
                      S.par_off, S.par, L.pos, L.posf, L.sc, L.hist, L.outs, L.outh, L.bests, L.frame, L.turn, L.selfemit,
                      L.cnt, L.key, L.first, L.hbin, S.ps, L.pstamp8, S.rootnodes, S.n_rootnodes, L.propf, L.posout,
                      L.act[cur], blockIdx.x, GA, 0, UHX, L.claim);
        return;
    }
    d_dec_resolve_children<uint8_t, HEUR>(S.N, S.T, f, frame_beams(S, f), L.best, nact_cur, S.node_base, S.tree_of, S.prob,
                  S.par_off, S.par, L.pos, L.posf, L.sc, L.hist, L.outs, L.outh, L.bests, L.frame, L.turn, L.selfemit,
                  L.cnt, L.key, L.first, L.hbin, S.ps, L.pstamp8, S.rootnodes, S.n_rootnodes, L.propf, L.posout,
                  L.plist, L.pcnt[f & 1], S.psmem_off, S.psmem, (int32_t)blockIdx.x - GA, UR_GL, UHX);
}

template <bool HEUR, int URK = UR_K>
__global__ void __launch_bounds__(RSBLOCK)
ku_resolve_lists(const ULane *__restrict__ lanes, UShared S, int32_t fg)
{
    LANE;
    const int32_t GB = ((S.N + URK - 1) / URK + RSBLOCK - 1) / RSBLOCK;
    d_dec_resolve_utt<URK, uint8_t, HEUR>(S.N, S.T, f, frame_beams(S, f), L.best, nact_cur, S.node_base, S.tree_of, S.prob,
                  S.par_off, S.par, L.pos, L.posf, L.sc, L.hist, L.outs, L.outh, L.bests, L.frame, L.turn, L.selfemit,
                  L.cnt, L.key, L.first, L.hbin, S.ps, L.pstamp8, S.rootnodes, S.n_rootnodes, L.propf, L.posout,
                  L.act[cur], blockIdx.x, (int32_t)gridDim.x - GB, GB, UHX);
}

template <int NT>
__global__ void __launch_bounds__(NT)
ku_scan(const ULane *__restrict__ lanes, UShared S, int32_t NC, int32_t GC, int32_t fg)
{
    LANE;
    const FrameBeams bm = frame_beams(S, f);
    /* after a histogram reordering the position-indexed word ids / exit scores are stale */
    int32_t n = 0;
    for (int32_t t = 0; t < S.T; t++) n += nact_cur[t];
    const int32_t reordered = n > bm.maxhmmpf + (bm.maxhmmpf >> 1) ? 1 : 0;
    d_dec_scan_t<NT>(S.N, S.T, f, bm, S.node_base, L.act[cur], L.nact[cur], S.wid, S.prob, L.outs, L.outh, L.selfemit,
               L.cnt, L.base, L.act[cur ^ 1], L.nact[cur ^ 1], L.pos, L.posf, L.best, L.exits, L.nexit, L.hbin, L.misc,
               (int32_t *)NULL /* no tail: ku_wordlevel assembles the frame record */, L.pack, S.pack_max_exits, L.gpart, S.gp_n, L.poswid, L.posout, reordered, L.scan_agg, L.scan_pre,
               L.scan_flag, S.scan_chunks, ctx->scan_epoch, NC, GC, blockIdx.x, 0);
}

/* ---- the ordered emission of the next list AND the word level, one launch ----
 * workgroup 0 of a lane: the frame record (d_dec_pack_frame) + the word level, which closes the frame and leaves the
 * next frame's lextree_enter calls; workgroups 1 .. 8 T: the emission sweep (16 waves each, 8 workgroups per tree).
 * The two read nothing the other writes. */
#define UE_WG_PER_TREE (8 * 1024 / WL_THREADS)      /* the emission sweep keeps its 128 waves per tree */
#ifndef WL_EMIT_WPE
#define WL_EMIT_WPE 1
#endif
__global__ void __launch_bounds__(WL_THREADS, WL_EMIT_WPE)
ku_emit_word(const ULane *__restrict__ lanes, UShared S, WLm lm, WDict dict, WPar par, int32_t fg, int32_t big)
{
    LANE;
    if (blockIdx.x > 0) {
        const int32_t wgpt = ((int32_t)gridDim.x - 1) / S.T;        /* emission workgroups per tree */
        const int32_t t = (blockIdx.x - 1) / wgpt, bx = (blockIdx.x - 1) % wgpt;
        d_dec_emit_w(f, S.node_base, L.act[cur], L.nact[cur], S.child_off, S.child, L.turn, L.selfemit, L.base,
                     L.act[cur ^ 1], L.nact[cur ^ 1], L.pos, L.posf, t, bx * WL_WAVES + (threadIdx.x >> 6),
                     wgpt * WL_WAVES);
        return;
    }
    __shared__ int32_t s_hdr[6 * WL_MAXT + 16], s_ex[3 * WL_LDS_EX];
    const long long t_in = (long long)wall_clock64();
    if (threadIdx.x == 0) ctx->scan_epoch++;        /* (k_dec_scan's flags are stamped per launch) */
    const int32_t nx = d_dec_pack_frame_lds(S.N, S.T, frame_beams(S, f), S.node_base, L.nact[cur], L.best, L.exits, L.nexit, L.hbin,
                                            L.misc, L.pack, S.pack_max_exits, L.gpart, S.gp_n, L.nact[cur ^ 1], s_hdr, s_ex, WL_LDS_EX);
    if (big) d_wl_big_begin(L.w, ctx, L.pack, dict, par);   /* wide beams: the candidate phases follow as their own launches */
    else d_wordlevel_frame(L.w, ctx, L.pack, s_hdr, nx <= WL_LDS_EX ? s_ex : (const int32_t *)NULL, lm, dict, par, f, t_in);
}

/* ------------------------------------------------------------------ */
/* the hypothesis of a finished lane, on the device                    */
/* ------------------------------------------------------------------ */
/*
 * vithist_utt_end (vithist.c:766-860) + vithist_backtrace (vithist.c:1066-1100) + compute_scale (srch_output.c:52-60)
 * for every lane behind its last frame: the best transition into </s> from the last frame that has entries (the
 * earliest of equals), a silence entry over the rest when the search died early, the backtrace, every word's sum of
 * frame normalisers.  Nothing is added to the lane's table: the final entries exist in the record only.  What the
 * host reads back per utterance is UH_N words + 24 bytes per hypothesis word, not the history table: the lanes' words
 * are packed one behind the other (a lane reserves its place with one atomicAdd on the word counter behind the headers),
 * so that ONE linear copy brings them over.
 */
enum { UH_STATUS, UH_NENTRY, UH_NFRM, UH_TSCALE, UH_NWORDS, UH_SCORE, UH_EXIT, UH_WOFF,
       UH_ERR, UH_CF, UH_MAXCAND, UH_MAXNEW, UH_NTIE, UH_NFR, UH_PAD0, UH_PAD1, UH_N };   /* (from UH_ERR on: the lane's context when it ended) */
#define UH_FIRST 96          /* words per lane that travel with the headers (more: a second copy) */
struct UHypPar { int32_t finish_lwid, finishwid, silwid, wcap, wtotal; };     /* wcap: words per hypothesis; wtotal: of the packed buffer */
#define UH_T 256
#define UH_IDS 2048

/* (NT threads of one workgroup; hdr: the utterance's header slot) */
template <int NT>
__device__ __forceinline__ void
d_hyp(const ULane &L, const UCtx *ctx, const WLm &lm, const WDict &dict, const UHypPar &P, int32_t *__restrict__ hdr, int32_t *__restrict__ words_all,
      int32_t *__restrict__ wcount, int32_t *s_ids /* [UH_IDS] of the workgroup's LDS */)
{
    const WLane &w = L.w;
    const int32_t tid = threadIdx.x, nfr = ctx->nfr, n_entry = w.st[0], n_frm = w.st[1];
    __shared__ uint32_t s_scale;
    __shared__ int32_t s_woff;
    __shared__ unsigned long long s_best;
    __shared__ int32_t s_f, s_n;
    if (tid == 0) { s_scale = 0u; s_best = 0ull; s_n = 0; }
    __syncthreads();
    uint32_t part = 0u;
    for (int32_t f = tid; f < nfr; f += NT) part += (uint32_t)w.fstat[(size_t)f * 8];
    atomicAdd(&s_scale, part);
    if (tid == 0) {
        int32_t f;
        for (f = n_frm - 1; f >= 0; --f)
            if (w.frame_start[f] < w.frame_start[f + 1]) break;
        s_f = f;
    }
    __syncthreads();
    const int32_t f = s_f, err = ctx->err;
    if (tid == 0) {
        hdr[UH_STATUS] = err ? -1 : (f < 0 ? -2 : 0); hdr[UH_NENTRY] = n_entry; hdr[UH_NFRM] = n_frm; hdr[UH_TSCALE] = (int32_t)s_scale;
        hdr[UH_NWORDS] = 0; hdr[UH_SCORE] = 0; hdr[UH_EXIT] = -1; hdr[UH_WOFF] = 0;
        hdr[UH_ERR] = err; hdr[UH_CF] = ctx->cf; hdr[UH_MAXCAND] = ctx->max_cand; hdr[UH_MAXNEW] = ctx->max_new;
        hdr[UH_NTIE] = ctx->n_tie_frames; hdr[UH_NFR] = nfr; hdr[UH_PAD0] = 0; hdr[UH_PAD1] = 0;
    }
    if (err || f < 0) return;               /* (f < 0: no word exit at all -- vithist_utt_end returns -1) */
    const int32_t sv = w.frame_start[f], nsv = w.frame_start[f + 1];
    for (int32_t i = sv + tid; i < nsv; i += NT) {
        const int32_t sc = add32(w.score[i], wl_tg_score(lm, w.lw1[i], w.lw0[i], P.finish_lwid, P.finishwid));
        atomicMax(&s_best, wl_pack(sc, (uint32_t)i));            /* best < s: the earliest of equals */
    }
    __syncthreads();
    const int32_t bestvh = (int32_t)(0xffffffffu - (uint32_t)(s_best & 0xffffffffull));
    int32_t best = (int32_t)((uint32_t)(s_best >> 32) ^ 0x80000000u);
    const bool have_sil = f != n_frm - 1;   /* the search died early: a silence entry over the rest (vithist.c:817-826) */
    if (tid == 0) {
        int32_t n = 0;
        for (int32_t i = bestvh; i > 0; i = w.pred[i], n++) if (n < UH_IDS) s_ids[n] = i;
        s_n = n;
        const int32_t tot = n + (have_sil ? 1 : 0) + 1;
        s_woff = tot <= P.wcap ? atomicAdd(wcount, tot) : -1;
        if (s_woff >= 0 && (long long)s_woff + tot > (long long)P.wtotal) s_woff = -1;
    }
    __syncthreads();
    const int32_t n = s_n, total = n + (have_sil ? 1 : 0) + 1;
    int32_t *words = words_all + (size_t)max(s_woff, 0) * 6;
    int32_t last_ef = w.ef[bestvh], last_score = w.score[bestvh];
    int32_t sil_lscr = 0;
    if (have_sil) {
        sil_lscr = dict.fillpen[P.silwid];
        const int32_t sil_score = add32(w.score[bestvh], sil_lscr);
        best = add32(sil_score, wl_tg_score(lm, w.lw1[bestvh], w.lw0[bestvh], P.finish_lwid, P.finishwid));
        last_ef = n_frm - 1; last_score = sil_score;
    }
    if (tid == 0) { hdr[UH_NWORDS] = total; hdr[UH_SCORE] = best; hdr[UH_EXIT] = n_entry + (have_sil ? 1 : 0); hdr[UH_WOFF] = max(s_woff, 0); }
    if (s_woff < 0) { if (tid == 0) hdr[UH_STATUS] = -3; return; }
    if (n <= UH_IDS) {
        for (int32_t q = tid; q < n; q += NT) {
            const int32_t i = s_ids[n - 1 - q];
            int32_t *o = words + (size_t)q * 6;
            o[0] = w.wid[i]; o[1] = w.sf[i]; o[2] = w.ef[i]; o[3] = w.ascr[i]; o[4] = w.lscr[i];
        }
    }
    else if (tid == 0) {
        int32_t k = n - 1;
        for (int32_t i = bestvh; i > 0; i = w.pred[i], k--) {
            int32_t *o = words + (size_t)k * 6;
            o[0] = w.wid[i]; o[1] = w.sf[i]; o[2] = w.ef[i]; o[3] = w.ascr[i]; o[4] = w.lscr[i];
        }
    }
    if (tid == 0) {
        int32_t k = n;
        if (have_sil) {
            int32_t *o = words + (size_t)k * 6;
            o[0] = P.silwid; o[1] = w.ef[bestvh] + 1; o[2] = n_frm - 1; o[3] = add32(w.score[bestvh], -w.score[bestvh]); o[4] = sil_lscr;
            k++;
        }
        int32_t *o = words + (size_t)k * 6;
        o[0] = P.finishwid; o[1] = last_ef + 1; o[2] = n_frm; o[3] = 0; o[4] = add32(best, -last_score);
    }
    __syncthreads();
    for (int32_t q = tid; q < total; q += NT) {           /* compute_scale */
        int32_t *o = words + (size_t)q * 6;
        uint32_t sc = 0u;
        for (int32_t i = max(o[1], 0); i < o[2] && i < nfr; i++) sc += (uint32_t)w.fstat[(size_t)i * 8];
        o[5] = (int32_t)sc;
    }
}


__global__ void __launch_bounds__(UH_T)
ku_hyp(const ULane *__restrict__ lanes, WLm lm, WDict dict, UHypPar P, int32_t *__restrict__ hdr_all, int32_t *__restrict__ words_all,
       int32_t *__restrict__ wcount, const int32_t *__restrict__ sub, const int32_t *__restrict__ slot)
{
    /* (sub / slot: a refill event -- lane sub[x] has finished the utterance whose header goes to slot[x]; NULL: lane x, header x) */
    const ULane &L = lanes[sub ? sub[blockIdx.x] : (int32_t)blockIdx.x];
    __shared__ int32_t s_ids[UH_IDS];
    d_hyp<UH_T>(L, L.ctx, lm, dict, P, hdr_all + (size_t)(slot ? slot[blockIdx.x] : (int32_t)blockIdx.x) * UH_N, words_all, wcount, s_ids);
}

/* ====================================================================================================================
 * ku_frames: THE LANE'S FRAMES IN ONE LAUNCH -- the frame as the phases of a persistent workgroup (round 5).
 *
 * The reference's frame loop (srch.c:746-835; lextree.c:1253-1597) has no boundary between its steps; the launches above
 * have twelve per frame, each a grid over all lanes whose workgroups mostly find out that they are not needed, and several
 * engines take turns on the chip.  Here a lane is ONE 512-thread workgroup (or a cluster of C of them) that stays on its CU
 * and walks the same steps with a workgroup barrier between them: lextree_enter (test / rank / apply + senone marks), the
 * composite senones' members, the CI gate (d_select), the composite maxima, lextree_hmm_eval, the stamps of the HMMs that
 * propagate + the list of stamped parent sets (or the histogram beam and the reordering), -ptranskip's weak HMMs,
 * lextree_hmm_propagate_non_leaves from the node's point of view (by list position + a wave per listed parent set), the
 * ordered scan, the emission of the next list and the word level.  The bodies are the launch path's
 * (s3a_decoder_kernels.h, s3a_wordlevel.h): same rule, same bits.
 *
 * SCORES FIRST.  A senone's score does not depend on the search, so the look-ahead pass (ku_score_window) scores EVERY frame
 * of every utterance of the call before the search starts (rows [frame][senone] in HBM, 5 bytes per senone and frame: what a
 * 288 GB device is for), and the lanes never meet again: measured on the hub4-shaped batch, the time lanes take for the same
 * eight frames spreads 1 : 4.7 : 16 (fastest : median : slowest), so every boundary all lanes must reach together costs
 * the median lane more than its own work.  Three modes:
 *   KF_WINDOW  frames [fg0, fg0 + n_fr) of every lane from its K-frame window rows (the launch path's buffers): what is left
 *              when the scores of the whole call do not fit the device;
 *   KF_STATIC  lane z decodes utterance z from its first to its last frame (s3a_uttdec_decode: at most n_lanes utterances,
 *              the tables stay in the lanes);
 *   KF_QUEUE   a lane TAKES the queue's next utterance (one atomic), begins it (srch_TST_begin's resets, d_lane_begin),
 *              decodes it, leaves its hypothesis in the utterance's slot (d_hyp) and ends it (lextree_utt_end, d_lane_end)
 *              -- s3a_uttdec_decode_queue without a schedule: no lane waits for a boundary, the chip drains only once.
 *
 * A lane's data is its own, so C = 1 needs nothing but __syncthreads() (two lanes per CU: 512 lanes fill the chip).  With
 * fewer lanes than workgroup slots a lane is a CLUSTER of C workgroups on one XCD (block b runs on XCD b % 8: observed, used
 * for speed only) with a counter barrier between the phases: every workgroup arrives behind an agent-scope release and leaves
 * through an agent-scope acquire (a CU's L1 is not refreshed by other CUs' stores, the XCDs' L2s are not coherent with each
 * other).  Words that ATOMICS of an earlier phase changed are read past the L1 (S3A_ALD); the per-tree maxima are copied to
 * LDS once per frame and every later phase reads the copy.  LDS: the word level's arrays + one pool the other phases share
 * (two workgroups per CU).
 * Not served here (the engine then keeps the launch path): per-frame scoring (window = 0), the invariant checker, per-launch profiling.
 * ==================================================================================================================== */
#define KF_NT WL_THREADS
#define KF_WAVES (KF_NT / 64)
#define KF_SPIN_MAX (1 << 21)
#define KF_MAXC 32
#define KF_MAXSEG (KF_MAXC * KF_WAVES)
#define KF_PSBITS 4096
#define KF_PSTAB 512
#define KF_SENBITS 16384          /* senones ku_frames keeps an activity bit for in LDS (more: the launches stay) */
#ifndef KF_ER
#define KF_ER 4                 /* runs of 64 entries a wave tests per turn of lextree_enter's sweep */
#endif
#ifndef KF_RL
#define KF_RL 8                 /* list positions a thread classifies per turn of the propagation step's first pass */
#endif
#ifndef KF_SK
#define KF_SK 8                 /* list positions a thread stamps per pass */
#endif
#ifndef KF_RK
#define KF_RK 4                 /* kept entries a thread ranks per pass of lextree_enter's ranking (8: 706.6, 4: 713.1, 2: 711.7 k frames/s, same box: profiles/r6_experiments.txt 12) */
#endif
#define KF_SETS 512            /* listed parent sets a workgroup takes per pass of the propagation step */
#define KF_TP_LDS 1024          /* words of transition matrices kept in LDS */
#define KF_ENT 640              /* propagating parents of a pass's several-parent sets kept in LDS (a set whose parents find no room goes the wave-per-set way) */
#define KF_BIG 256              /* ... of which several-parent sets whose headers stay in LDS (the others: d_dec_resolve_children) */
static_assert(KF_NT == 512, "ku_frames: the word level's workgroup is the frame's workgroup");
enum { KF_WINDOW, KF_STATIC, KF_QUEUE };
#define KF_ST_WORDS 8
#define KF_STAGES 2             /* launches a call's relay may have behind its first */

union KfPool {                  /* phases that never overlap share this LDS */
    struct { int32_t off[WL_MAXCALL], root[WL_MAXCALL], in[WL_MAXCALL], hist[WL_MAXCALL]; } e1;
    int32_t bin[NBIN];
    HistSortWs<KF_NT> hs;
    struct {                    /* the propagation pass: the big sets' per-wave parent tables; the listed sets of a pass */
        union {                     /* (the tables are read before the wave-per-set routine borrows the area) */
            int32_t rc[KF_WAVES][5 * 64];
            int32_t ent[KF_ENT][5];         /* the several-parent sets' propagating parents, chained per set: exit score, list position, exit
                                             * history, probability, next entry of the set (-1: none) */
        };
        int32_t mlo[KF_SETS], pre[KF_SETS + 1], big[KF_SETS], nbig, m_all;
        int32_t bmlo[KF_BIG], bnm[KF_BIG], bkp0[KF_BIG], bnp[KF_BIG];      /* the several-parent sets' headers: members, parents */
        int32_t bnq[KF_BIG], bpre[KF_BIG + 1], bmpre[KF_BIG + 1], leg[KF_BIG], nleg;       /* first entry of the set's chain (-1: none; -2: no room), work-item prefixes */
        int32_t n_ent;
    } rs;
    struct { int32_t hdr[6 * WL_MAXT + 16], ex[3 * WL_LDS_EX]; } wl;
};

struct KfSh {                   /* the workgroup's LDS outside the word level's own arrays */
    KfPool pool;
    int32_t best[2 * WL_MAXT], acc[2 * WL_MAXT], pre[WL_MAXT + 1], red[KF_WAVES], dead, u, u2, dynbeam;
    ULane Lc;                   /* the lane's structure (its ~60 pointers): a copy in LDS -- a field is a ds_read at a constant address; out of
                                 * memory it was the structure's spilled address back from scratch, then the pointer, then the data: two
                                 * round trips in front of many a step's first access */
    int32_t nb[WL_MAXT + 1];    /* the trees' first nodes (UShared.node_base: in LDS, a list position's place in the arrays is then LDS arithmetic only) */
    int32_t seg[KF_MAXSEG + 1], ws[KF_WAVES + 1], gq[4];    /* lextree_enter: the waves' segments of passing entries, scan scratch */
    int32_t rk[KF_RK][KF_WAVES];  /* ... the ranking pass's counts per (run, wave) */
    uint32_t senbits[KF_SENBITS / 32];  /* srch_TST_select_active_gmm's mask of the frame, a bit per senone (the launch path: a byte each in HBM) */
    int32_t pstab[KF_PSTAB];        /* the frame's stamped parent sets, an open-addressed table (-1: free): membership is exact */
    int32_t ps_exact;               /* ... when they all found room (else the bit filter in front of pstamp8 / claim) */
    uint32_t psbits[KF_PSBITS / 32];    /* the frame's stamped parent sets, a bit per set id modulo KF_PSBITS (a filter in front of pstamp8) */
    long long kacc[16];         /* the steps' clock of the utterance so far (UCtx.kacc) */
    int32_t thr[4];             /* the frame's thresholds: HMM, phone, word (final once the histogram beam is known) */
    int32_t tp[KF_TP_LDS];      /* the transition matrices (when they fit: 48 of hub4's 3-state topology are 2.3 KB) */
    /* the frame's operands (KF_CALL: the frame is a function of its own -- its registers are then allocated for the frame alone, not for the
     * frame inside the kernel's loops over utterances and frames; a function's arguments travel in VGPRs, so what is uniform arrives here
     * and is read back through readfirstlane) */
    struct { const void *A; UCtx *ctx; int32_t *row; const uint8_t *brow; int32_t *bar_cnt; int32_t z, r, C, f, weak, bar_target, bar_local; } fa;
};

struct KfBar { int32_t *cnt; int32_t C, target; int32_t *dead; int32_t local; };

/* what a call's launch works on besides the lanes (by value) */
struct KfJob {
    int32_t mode, fg0, n_fr;    /* KF_WINDOW: the engine frames of this launch */
    int32_t n_utt;              /* KF_QUEUE: utterances of this launch (the queue's, or a part of it that fits the score buffer) */
    int32_t u0;                 /* ... the first one's place in the queue */
    const int32_t *order;       /* ... [queue] which utterance the k-th take of the counter is: a part's utterances longest first (the launch ends with
                                 * its slowest lane: what is taken last should be short) */
    /* THE RELAY (KF_QUEUE / KF_STATIC): a launch ends with its slowest lane, and while the last lanes finish their utterances most of the
     * chip idles.  So the call is a CHAIN of launches: when no utterance is left to take and at most `stop_at` lanes are still decoding,
     * those lanes leave at their next frame boundary -- everything a frame needs of the last one is in device memory --, hand themselves
     * over (lane, utterance, next frame), and the next launch of the chain continues them as clusters of more workgroups each. */
    int32_t stop_at;            /* > 0: hand over when that few lanes are left (0: this launch runs to the end) */
    int32_t resume;             /* this launch's lanes are the ones the previous launch handed over (st_cur's list) */
    int32_t *st_cur, *st_next;  /* [KF_ST_WORDS + n_lanes] this launch's / the next one's relay words: [0] lanes still at work, [1] the stop flag,
                                 * [2] lanes handed TO this launch, [KF_ST_WORDS ...] which */
    int32_t *lane_f, *lane_uq, *lane_stop;  /* [n_lanes] the frame a handed-over lane goes on with | its utterance | the flag as its cluster saw it */
    int32_t *next;              /* ... [1] the counter the lanes take their utterances from */
    int32_t *lane_u;            /* ... [n_lanes] what a lane's first workgroup took (read by the rest of its cluster) */
    const UCtx *stage;          /* ... [queue] the utterances' staged contexts */
    const long long *row0;      /* KF_STATIC: [n_lanes], KF_QUEUE: [queue]: the utterance's first row in the score buffer */
    int32_t *scores;            /* [rows][n_sen] every frame's senone scores (ku_score_window); d_select patches the back-offs in place */
    const uint8_t *bests;       /* [rows][n_sen] ... and the best component of every mixture */
    int32_t *hdr, *words, *wcount;  /* KF_QUEUE: the hypotheses (ku_hyp's buffers) */
    UHypPar P;
    UBegin B;
    int32_t n_word;
};

/* all workgroups of the lane's cluster have finished the phase and see what the others wrote.
 * Two forms.  The general one: arrive behind an agent-scope release (buffer_wbl2: the L2 writes its dirty lines back -- the XCDs' L2s
 * are not coherent with each other), leave through an agent-scope acquire.  The XCD-LOCAL one (B.local: every workgroup of the cluster
 * read the same XCC_ID at the start of the launch -- checked, not assumed): the cluster's CUs share ONE L2, a CU's L1 writes through,
 * so a store whose wave has counted it down (s_waitcnt vmcnt(0), EVERY wave before the workgroup barrier: __syncthreads() is a bare
 * s_barrier here) is where the others' loads find it once their L1 is invalidated (buffer_inv sc1); no write-back of the L2. */
__device__ __forceinline__ void
kf_barrier(KfBar &B)
{
    if (B.C == 1) { __syncthreads(); return; }
    if (B.local) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    B.target += B.C;
    if (threadIdx.x == 0 && !*B.dead) {
        if (!B.local) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        (void)__hip_atomic_fetch_add(B.cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int32_t spins = 0;
        while (__hip_atomic_load(B.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < B.target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > KF_SPIN_MAX) { *B.dead = 1; break; }        /* (cannot happen while the cluster is resident: the host sizes the grid) */
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

/* list position g of the lane's trees laid end to end -> (tree, position in its list); pre[t] = positions in front of tree t */
__device__ __forceinline__ void
kf_locate(const int32_t *pre, int32_t T, int32_t g, int32_t &t, int32_t &i)
{
    t = 0;
    while (t + 1 < T && g >= pre[t + 1]) t++;
    i = g - pre[t];
}

/*
 * lextree_hmm_eval's step for one HMM, as ku_frames runs it.  What bounds the frame's steps on this chip is the NUMBER of scattered
 * accesses -- a gather or scatter instruction costs a cycle per lane in the CU's address path whatever it moves (measured: asking for
 * the chain's loads turns ahead changes nothing) --, so the HMM's 64-byte record travels as 16-byte pieces (2 + 3 accesses for a
 * 3-state HMM instead of 8 loads + 10 stores), its senone ids come packed with the node (nodesen: one access instead of the sequence
 * id and three 2-byte gathers), a composite senone's score arrives with its weight added (cs_val: d_comsen_list), and the transition
 * matrices are read from LDS when they fit.  Same arithmetic as d_dec_hmm_eval_nd (vit3 / vit5); the frame tag is written along.
 */
template <int NE>
__device__ __forceinline__ int32_t
kf_hmm_eval(int32_t v, const int4 nd, const int32_t *__restrict__ nodesen, const int32_t *__restrict__ tp_g, const int32_t *tp_lds, const bool tp_in_lds,
            const int32_t *__restrict__ raw, const int32_t *raw_lds, const bool raw_in_lds, int32_t norm,
            const int32_t *__restrict__ cs_valw, int32_t *rec_all, int32_t cf, int32_t &w, int32_t &out)
{
    constexpr int NV = NE == 3 ? 2 : 3;             /* 16-byte pieces that hold scores, histories, exit score, exit history */
    S3A_AS1 s3a_v4i *rec = (S3A_AS1 s3a_v4i *)(rec_all + NSV(v));
    /* (everything that comes from memory is asked for first -- the ids are in the packed node --, as GLOBAL accesses (GM: above), the
     * row of scores and the transition matrix come from LDS behind them) */
    int32_t id[NE], tmat, wid_, comp;
    if (NE == 3) {              /* (nd = the node's packed word, UShared.nodepk) */
        id[0] = nd.x & 0xffff; id[1] = (int32_t)((uint32_t)nd.x >> 16); id[2] = nd.y & 0xffff;
        tmat = (int32_t)((uint32_t)nd.y >> 16); wid_ = nd.z; comp = nd.w & 1;
    }
    else {                      /* (nd = node4) */
        const s3a_v4i a5 = *(const S3A_AS1 s3a_v4i *)(nodesen + (size_t)v * 4);
        const int32_t h[5] = { a5.x & 0xffff, (int32_t)((uint32_t)a5.x >> 16), a5.y & 0xffff, (int32_t)((uint32_t)a5.y >> 16), a5.z & 0xffff };
#pragma unroll
        for (int st = 0; st < NE; st++) id[st] = h[st];
        tmat = nd.y; wid_ = nd.z; comp = nd.w;
    }
    int32_t e[NE], eg[NE];
    /* (a composite node's maxima from memory; a plain node's scores from the row in LDS, or from memory when the row does not fit:
     * one global load per state, at a harmless index for the lanes that do not want it) */
    const S3A_AS1 int32_t *gsrc = GMC(comp ? cs_valw : raw);
#pragma unroll
    for (int st = 0; st < NE; st++) eg[st] = gsrc[(comp || !raw_in_lds) ? id[st] : 0];
    int32_t wd[4 * NV];
#pragma unroll
    for (int q = 0; q < NV; q++) { const s3a_v4i a = rec[q]; wd[4 * q] = a.x; wd[4 * q + 1] = a.y; wd[4 * q + 2] = a.z; wd[4 * q + 3] = a.w; }
#pragma unroll
    for (int st = 0; st < NE; st++) e[st] = raw_in_lds ? LM(raw_lds)[comp ? 0 : id[st]] : 0;
    int32_t tp[NS_TPW(NE)];
    if (tp_in_lds) {
#pragma unroll
        for (int q = 0; q < NS_TPW(NE); q++) tp[q] = LM(tp_lds)[tmat * NS_TPW(NE) + q];
    }
    else {
        const S3A_AS1 s3a_v4i *tq = (const S3A_AS1 s3a_v4i *)(tp_g + tmat * NS_TPW(NE));
#pragma unroll
        for (int q = 0; q < NS_TPW(NE) / 4; q++) { const s3a_v4i a = tq[q]; tp[4 * q] = a.x; tp[4 * q + 1] = a.y; tp[4 * q + 2] = a.z; tp[4 * q + 3] = a.w; }
    }
#pragma unroll
    for (int st = 0; st < NE; st++) if (comp || !raw_in_lds) e[st] = eg[st];
#pragma unroll
    for (int st = 0; st < NE; st++) e[st] = add32(e[st], -norm);
    HmmRegsT<int32_t> r;
#pragma unroll
    for (int st = 0; st < NE; st++) { r.s[st] = wd[st]; r.h[st] = wd[NE + st]; }
    r.out = wd[2 * NE]; r.outh = wd[2 * NE + 1];
    int32_t k;
    if (NE == 5) { int32_t out_written = 0; k = vit5(r, tp, e, out_written); (void)out_written; }
    else k = vit3(r, tp, e[0], e[1], e[2]);
#pragma unroll
    for (int st = 0; st < NE; st++) { wd[st] = r.s[st]; wd[NE + st] = r.h[st]; }
    wd[2 * NE] = r.out; wd[2 * NE + 1] = r.outh;
#pragma unroll
    for (int q = 0; q < NV; q++) { s3a_v4i o; o.x = wd[4 * q]; o.y = wd[4 * q + 1]; o.z = wd[4 * q + 2]; o.w = wd[4 * q + 3]; rec[q] = o; }
    /* the best score and the frame tag (as if the HMM survived the frame: kf_frame's propagation pass corrects the ones it clears) */
    static_assert(2 * 3 + 2 == 8 && 2 * 5 + 2 == 12, "the record's layout: s3a_structs.h");
    { s3a_v4i o; o.x = k; o.y = cf + 1; o.z = 0; o.w = 0; rec[NV] = o; }
    w = wid_; out = r.out;
    return k;
}

/* srch_TST_select_active_gmm's step for one node from its packed ids (mark_node_senones with one gather instead of five): a plain
 * node's senones join the mask, a composite node's composite senones are stamped and -- by whoever stamps one first -- listed */
template <int NE>
__device__ __forceinline__ int4
kf_mark_load(int32_t v, const int32_t *__restrict__ nodesen)
{
    /* (3 states: `nodesen` is UShared.nodepk, the node's packed word; brought to nodesen's form: ids, then the composite flag) */
    if (NE == 3) { const s3a_v4i a = ((const S3A_AS1 s3a_v4i *)nodesen)[v]; return make_int4(a.x, (a.y & 0xffff) | ((a.w & 1) << 16), 0, 0); }
    { const s3a_v4i a = *(const S3A_AS1 s3a_v4i *)(nodesen + (size_t)v * 4); return make_int4(a.x, a.y, a.z, a.w); }
}
template <int NE>
__device__ __forceinline__ void
kf_mark_apply(const int4 a, uint32_t *senbits, int32_t *cs_need, int32_t stamp, int32_t *cs_wl, int32_t *cs_wn)
{
    int32_t id[NE], comp;
    if (NE == 3) {
        id[0] = a.x & 0xffff; id[1] = (int32_t)((uint32_t)a.x >> 16); id[2] = a.y & 0xffff; comp = (int32_t)((uint32_t)a.y >> 16);
    }
    else {
        const int32_t h[5] = { a.x & 0xffff, (int32_t)((uint32_t)a.x >> 16), a.y & 0xffff, (int32_t)((uint32_t)a.y >> 16), a.z & 0xffff };
#pragma unroll
        for (int st = 0; st < NE; st++) id[st] = h[st];
        comp = (int32_t)((uint32_t)a.z >> 16);
    }
    if (comp) {
#pragma unroll
        for (int st = 0; st < NE; st++)
            if (atomicExch(&cs_need[id[st]], stamp) != stamp) cs_wl[atomicAdd(cs_wn, 1)] = id[st];
    }
    else {
#pragma unroll
        for (int st = 0; st < NE; st++) lds_or(&senbits[id[st] >> 5], 1u << (id[st] & 31));       /* (LDS: three scattered stores less per node) */
    }
}

/* the same for four nodes at once (on[u]: node u counts): the plain nodes' bits; the composite nodes' stamps ALL asked for together,
 * then the list's counter once per wave and (node, state) -- a composite node taken by itself is three times exchange -> add -> store,
 * nine dependent round trips, and four of them in a row held a wave for 36 */
template <int NE>
__device__ __forceinline__ void
kf_mark_apply4(const int4 (&a)[4], const bool (&on)[4], uint32_t *senbits, int32_t *cs_need, int32_t stamp, int32_t *cs_wl, int32_t *cs_wn)
{
    const int32_t lane = threadIdx.x & 63;
    int32_t id[4][NE], old[4][NE];
    bool cp[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        int32_t comp;
        if (NE == 3) {
            id[u][0] = a[u].x & 0xffff; id[u][1] = (int32_t)((uint32_t)a[u].x >> 16); id[u][2] = a[u].y & 0xffff; comp = (int32_t)((uint32_t)a[u].y >> 16);
        }
        else {
            const int32_t h[5] = { a[u].x & 0xffff, (int32_t)((uint32_t)a[u].x >> 16), a[u].y & 0xffff, (int32_t)((uint32_t)a[u].y >> 16), a[u].z & 0xffff };
#pragma unroll
            for (int st = 0; st < NE; st++) id[u][st] = h[st];
            comp = (int32_t)((uint32_t)a[u].z >> 16);
        }
        cp[u] = on[u] && comp != 0;
        if (on[u] && !comp) {
#pragma unroll
            for (int st = 0; st < NE; st++) lds_or(&senbits[id[u][st] >> 5], 1u << (id[u][st] & 31));
        }
    }
    bool any = false;
#pragma unroll
    for (int u = 0; u < 4; u++) any = any || cp[u];
    if (!__ballot(any)) return;
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int st = 0; st < NE; st++) {
            old[u][st] = stamp;
            if (cp[u]) old[u][st] = __hip_atomic_exchange(GM(cs_need) + id[u][st], stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    int32_t at[4][NE];
    unsigned long long mm[4][NE];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int st = 0; st < NE; st++) {
            mm[u][st] = __ballot(old[u][st] != stamp);
            at[u][st] = 0;
            if (mm[u][st] && lane == __ffsll((long long)mm[u][st]) - 1)
                at[u][st] = __hip_atomic_fetch_add(GM(cs_wn), __popcll(mm[u][st]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int st = 0; st < NE; st++) {
            if (!mm[u][st]) continue;
            const int32_t base = __shfl(at[u][st], __ffsll((long long)mm[u][st]) - 1, 64);
            if (old[u][st] != stamp) GM(cs_wl)[base + __popcll(mm[u][st] & ((1ull << lane) - 1ull))] = id[u][st];
        }
}

/*
 * -maxcdsenpf inside ku_frames (approx_compute_dyn_ci_pbeam, approx_cont_mgau.c:303-357; ku_dyn_ci_beam of the launch path): the CI senones by
 * score, the frame's active CD senones counted per CI senone -- from the mask's bits in LDS --, the beam cut where the count passes the cap.
 * Every workgroup of a cluster holds the whole mask and works the same beam out for itself (*out, LDS): no barrier beside the frame's own.
 * ws: 3 UDB_MAXCI + 1 words of LDS (the pool: no other step's data lives there between the composite senones' marks and the evaluation).
 */
template <int NT>
__device__ __forceinline__ void
kf_dyn_ci_beam(const UShared &S, int32_t f, const int32_t *row, const uint32_t *senbits, int32_t *ws, int32_t *out)
{
    int32_t *s_occ = ws, *s_scr = ws + UDB_MAXCI, *s_ord = ws + 2 * UDB_MAXCI, *s_cut = ws + 3 * UDB_MAXCI;
    const int32_t n_ci = S.n_ci_sen, tid = threadIdx.x;
    for (int32_t c = tid; c < n_ci; c += NT) { s_occ[c] = 0; s_scr[c] = row[c]; }
    if (tid == 0) *s_cut = INT_MAX;
    __syncthreads();
    for (int32_t s = n_ci + tid; s < S.n_sen; s += NT)
        if ((senbits[s >> 5] >> (s & 31)) & 1u) atomicAdd(&s_occ[S.cd2cisen[s]], 1);
    __syncthreads();
    for (int32_t c = tid; c < n_ci; c += NT) {
        const int32_t v = s_scr[c];
        int32_t r = 0;
        for (int32_t c2 = 0; c2 < n_ci; c2++) { const int32_t v2 = s_scr[c2]; r += (v2 > v || (v2 == v && c2 < c)) ? 1 : 0; }
        s_ord[r] = c;
    }
    __syncthreads();
    const int32_t pbest = s_scr[s_ord[0]];
    for (int32_t r = tid; r < n_ci; r += NT) {
        int32_t total = 0;
        for (int32_t r2 = 0; r2 <= r; r2++) total += s_occ[s_ord[r2]];
        if (total > S.max_cd) atomicMin(s_cut, r);           /* the first rank at which the count passes the cap */
    }
    __syncthreads();
    if (tid == 0) {
        int32_t beam = S.ci_pbeam;
        if (*s_cut != INT_MAX) {
            /* (the reference's loop runs while the score is above pbest + ci_pbeam: a cut beyond that is none) */
            const int32_t v = s_scr[s_ord[*s_cut]];
            bool in_beam = true;
            for (int32_t r = 0; r <= *s_cut && in_beam; r++) in_beam = s_scr[s_ord[r]] > add32(pbest, S.ci_pbeam);
            if (in_beam) beam = v - pbest;
        }
        if (f % S.ds_ratio != 0) beam = (int32_t)((float)beam * S.tighten);
        *out = beam;
    }
    __syncthreads();
}

/* frame f of lane z (workgroup r of its C): row / brow = the frame's senone scores and best components */
/* HEUR: -pheurtype 1..3 (the phoneme look-ahead: the lane's heur_all / hth_pos, s3a_uttdec_enable_pheur) -- kernels of their own, so that the
 * ones without it are compiled as before */
template <int NE, bool EXACT, bool HEUR = false>
__device__ __forceinline__ void
kf_frame(const ULane &L, const UShared &S, UCtx *ctx, const WLm &lm, const WDict &dict, const WPar &par, KfSh &sh, KfBar &B, int32_t z,
         int32_t r, int32_t C, int32_t f, int32_t *row, const uint8_t *brow, int32_t weak_flags)
{
    const int32_t weak_possible = weak_flags & 1, big_wl = weak_flags & 2;
    const int32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, T = S.T;
    const int32_t gtid = r * KF_NT + tid, gstride = C * KF_NT, gwave = r * KF_WAVES + wave, gwaves = C * KF_WAVES;
    /* where the frame's time goes (thread 0 of the cluster's first workgroup: one clock read per step) */
    long long t_prev = 0;
    /* (the clock's sums stay in LDS and go to the lane's context once per utterance / launch: a read-modify-write of HBM by thread 0
     * in every step was 3 us each on the path to the step's barrier) */
#define KF_STAMP(i) do { if (r == 0 && tid == 0) { const long long t_ = (long long)wall_clock64(); sh.kacc[i] += t_ - t_prev; t_prev = t_; } } while (0)
    const int32_t cur = f & 1;
    const int32_t *nact_cur = S.nact_all + ((size_t)z * 2 + cur) * WL_MAXT;
    const FrameBeams bm = frame_beams(S, f);
    const int32_t n_ent = ctx->n_ent, n_calls_all = ctx->n_calls, thresh = ctx->thresh;
    if (r == 0 && tid == 0) { t_prev = (long long)wall_clock64(); sh.kacc[13]++; }
    if (f == 0) {               /* (an utterance that was stopped inside a frame may have left marks behind) */
        for (int32_t i = tid; i < KF_SENBITS / 32; i += KF_NT) sh.senbits[i] = 0u;
        if (C > 1 && r == 0) for (int32_t i = tid; i < KF_SENBITS / 32; i += KF_NT) L.senbits[i] = 0u;
        kf_barrier(B);
    }

    /* ---- lextree_enter (lextree.c:1093-1236; ku_enter1 / 2 / 3 of the launch path): the bench's task has ~7 calls (one per tree and
     * left context that a word left) with ~13.6 k (call, root) entries per frame, ~3 000 of which pass the threshold and ~2 600 improve
     * on their root -- ONE sweep tests them all (coalesced loads) and keeps those, in entry order -- every wave compacts its own
     * stretch --; ranking and applying then only see the kept ones.  Key / first: as d_dec_enter1.  (Looking only at the entries
     * that can pass -- prefixes of the root lists sorted by probability -- was built and is slower: its stages are gathers,
     * profiles/r6_experiments.txt 18.) ---- */
    if (n_ent > 0) {
        const int32_t n_calls = min(n_calls_all, WL_MAXCALL), nf = f;
        if (tid < n_calls) {
            sh.pool.e1.in[tid] = ctx->calls[4 * tid]; sh.pool.e1.hist[tid] = ctx->calls[4 * tid + 1];
            sh.pool.e1.root[tid] = ctx->calls[4 * tid + 2]; sh.pool.e1.off[tid] = ctx->calls[4 * tid + 3];
        }
        if (r == 0 && tid < T) L.n0[tid] = L.nact[cur][tid];            /* the list lengths before the entries */
        __syncthreads();
        const int32_t R = (((n_ent + gwaves - 1) / gwaves) + 63) & ~63, e_lo = gwave * R, e_hi = min(e_lo + R, n_ent);
        int32_t cntw = 0, c = 0;
        if (e_lo < e_hi) {          /* the call of the stretch's first entry; the lanes then only step forward */
            int32_t lo = 0, hi = n_calls - 1;
            while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (sh.pool.e1.off[mid] <= e_lo) lo = mid; else hi = mid - 1; }
            c = lo;
        }
        /* (four runs of 64 entries per turn: their probabilities, then the passing ones' roots, then those roots' entry scores are asked
         * for together -- a turn is three round trips whatever it holds; the order of the kept entries is the entries') */
        for (int32_t e0 = e_lo; e0 < e_hi; e0 += 64 * KF_ER) {
            /* an entry counts when it passes the threshold AND improves on its root's entry score: only such an entry can list the
             * root, win it, or tag it (the others pass through lextree_enter without a trace) */
            int32_t scr[KF_ER], vv[KF_ER], cc[KF_ER], idx[KF_ER], s0[KF_ER];
            bool keep[KF_ER];
#pragma unroll
            for (int u = 0; u < KF_ER; u++) {
                const int32_t e = e0 + 64 * u + lane;
                keep[u] = false; scr[u] = INT_MIN; vv[u] = 0; cc[u] = c; idx[u] = 0;
                if (e < e_hi) {
                    while (c + 1 < n_calls && sh.pool.e1.off[c + 1] <= e) c++;
                    cc[u] = c; idx[u] = sh.pool.e1.root[c] + (e - sh.pool.e1.off[c]);
                }
            }
            /* (the loads unconditional -- a lane past the end asks for entry 0's words --: a load under a condition is a branch with its
             * own wait inside, and the four runs' round trips would follow one another instead of running side by side) */
            int32_t rp[KF_ER];
#pragma unroll
            for (int u = 0; u < KF_ER; u++) { rp[u] = GMC(S.rootprob)[idx[u]]; vv[u] = GMC(S.rootlist)[idx[u]]; }
#pragma unroll
            for (int u = 0; u < KF_ER; u++) {
                if (e0 + 64 * u + lane < e_hi) scr[u] = add32(sh.pool.e1.in[cc[u]], rp[u]);
                keep[u] = e0 + 64 * u + lane < e_hi && scr[u] >= thresh;
            }
            /* (... and an entry under the threshold asks for ONE root's score, its run's first: no line of its own) */
#pragma unroll
            for (int u = 0; u < KF_ER; u++) s0[u] = GMC(L.sc)[NSV(keep[u] ? vv[u] : __shfl(vv[u], 0, 64))];
#pragma unroll
            for (int u = 0; u < KF_ER; u++) {
                keep[u] = keep[u] && s0[u] < scr[u];
                const unsigned long long m = __ballot(keep[u]);
                if (keep[u]) {
                    /* (what the later steps need of the entry, side by side: its root, its score, its call) */
                    const int32_t p_ = e_lo + cntw + __popcll(m & ((1ull << lane) - 1ull));
                    GM(L.eflag)[p_] = vv[u]; GM(L.ent)[2 * p_] = scr[u]; GM(L.ent)[2 * p_ + 1] = cc[u];
                    atomicMax(&L.key[vv[u]], ((unsigned long long)((uint32_t)scr[u] ^ 0x80000000u) << 32) | (uint32_t)(0x7fffffff - cc[u]));
                    atomicMin(&L.first[vv[u]], cc[u]);
                }
                cntw += __popcll(m);
            }
        }
        if (lane == 0) L.ctot[gwave] = cntw;
        kf_barrier(B);
        KF_STAMP(0);
        /* the kept entries: (1) the lane's first workgroup finds which of them list their root (the first qualifying call of a node
         * that is not listed yet), ranked in entry order by a scan -- the others meanwhile mark the senones of the list as it stood
         * before the entries --; (2) every workgroup applies its share: the listed roots, the winning entries, the frame tags (the
         * ranking READS the frame tags the applying WRITES: a barrier between them); the roots' scratch is cleaned one step later */
        {
            /* (every workgroup: where the waves' stretches of kept entries lie) */
            const int32_t x = tid < gwaves ? L.ctot[tid] : 0;
            int32_t incl = x;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int32_t y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
            if (lane == 63) sh.ws[wave] = incl;
            __syncthreads();
            int32_t add = 0;
            for (int32_t w = 0; w < wave; w++) add += sh.ws[w];
            if (tid < gwaves) sh.seg[tid] = add + incl - x;
            if (tid == gwaves - 1) sh.seg[gwaves] = add + incl;
            if (tid < 4) sh.gq[tid] = 0;
            __syncthreads();
        }
        const int32_t P = sh.seg[gwaves], c1 = ctx->n_groups > 1 ? ctx->groups[4 + 3] : INT_MAX;      /* (group 1's first call) */
        if (r == 0) {
            /* (KF_RK entries per thread and pass, their loads asked for together: a pass is two round trips and two barriers whatever
             * it holds -- one entry per thread and pass was six passes for the usual ~2 600 kept entries, each waiting on its own chain) */
            int32_t carry = 0, p0 = 0;
            if (P > 0) {            /* (the first kept entry's place: where a thread without an entry of its own sends its loads) */
                int32_t lo = 0, hi = gwaves - 1;
                while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (sh.seg[mid] <= 0) lo = mid; else hi = mid - 1; }
                p0 = lo * R;
            }
            for (int32_t i0 = 0; i0 < P; i0 += KF_RK * KF_NT) {
                int32_t q[KF_RK], c[KF_RK], p_[KF_RK], v[KF_RK], fs[KF_RK], fr[KF_RK];
#pragma unroll
                for (int k = 0; k < KF_RK; k++) {
                    const int32_t i = i0 + k * KF_NT + tid;
                    p_[k] = -1;
                    if (i < P) {
                        int32_t lo = 0, hi = gwaves - 1;
                        while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (sh.seg[mid] <= i) lo = mid; else hi = mid - 1; }
                        p_[k] = lo * R + (i - sh.seg[lo]);
                    }
                }
#pragma unroll
                /* (every load unconditional, at a harmless place for a thread without an entry: a load under a condition is a branch
                 * with its own wait inside -- eight round trips one after the other instead of one) */
                for (int k = 0; k < KF_RK; k++) { const int32_t pk = p_[k] >= 0 ? p_[k] : p0; v[k] = GMC(L.eflag)[pk]; c[k] = GMC(L.ent)[2 * pk + 1]; }
#pragma unroll
                for (int k = 0; k < KF_RK; k++) { fs[k] = S3A_ALD(&GM(L.first)[v[k]]); fr[k] = GMC(L.frame)[NSV(v[k])]; }
                unsigned long long m[KF_RK];
                int32_t n0 = 0;
#pragma unroll
                for (int k = 0; k < KF_RK; k++) {
                    q[k] = (p_[k] >= 0 && fs[k] == c[k] && fr[k] != nf) ? 1 : 0;       /* (sc < scr: true of every kept entry) */
                    m[k] = __ballot(q[k]);
                    n0 += __popcll(__ballot(q[k] && c[k] < c1));
                    if (lane == 0) sh.rk[k][wave] = __popcll(m[k]);
                }
                if (lane == 0 && n0) atomicAdd(&sh.gq[0], n0);
                __syncthreads();
                int32_t before = carry;
#pragma unroll
                for (int k = 0; k < KF_RK; k++) {
                    int32_t mine = before;
                    for (int32_t w = 0; w < KF_WAVES; w++) { const int32_t x = sh.rk[k][w]; if (w < wave) mine += x; before += x; }
                    if (p_[k] >= 0) GM(L.ent)[2 * p_[k] + 1] = ((mine + __popcll(m[k] & ((1ull << lane) - 1ull))) << 8) | (q[k] << 7) | c[k];
                }
                carry = before;
                __syncthreads();
            }
            /* the groups' new list lengths (group 0 = the unigram tree's calls, group 1 = the filler tree's: consecutive entries) */
            if (tid == 0) { sh.gq[1] = carry - sh.gq[0]; L.ctot[KF_MAXSEG] = sh.gq[0]; }
            __syncthreads();
            if (tid < ctx->n_groups) { const int32_t t = ctx->groups[4 * tid]; L.nact[cur][t] = L.n0[t] + sh.gq[tid]; }
        }
    }
    /* ---- the senone marks of the nodes that were on the frame's list before the entries (srch_TST_select_active_gmm): with a
     * cluster, by the workgroups that do not rank ---- */
    if (C == 1 || r > 0) {
        const int32_t *n0 = n_ent > 0 ? L.n0 : L.nact[cur];
        const int32_t mt = C == 1 ? tid : gtid - KF_NT, ms = C == 1 ? KF_NT : gstride - KF_NT;
        int32_t a = 0;
        for (int32_t t = 0; t < T; t++) {
            const int32_t na = n0[t], b = sh.nb[t];
            /* (the trees laid end to end: a thread's positions are mt, mt + ms, ... of the concatenation) */
            /* (four nodes per turn: their chains position -> node -> packed ids run side by side) */
            for (int32_t i = mt - a % ms + (mt < a % ms ? ms : 0); i < na; i += 4 * ms) {
                int32_t vq[4];
                int4 aq[4];
                bool oq[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { oq[u] = i + u * ms < na; vq[u] = GMC(L.act[cur])[b + (oq[u] ? i + u * ms : 0)]; }      /* (loads unconditional: see the ranking) */
#pragma unroll
                for (int u = 0; u < 4; u++) aq[u] = kf_mark_load<NE>(vq[u], NE == 3 ? (const int32_t *)S.nodepk : S.nodesen);
                kf_mark_apply4<NE>(aq, oq, sh.senbits, L.cs_need, f, L.cs_wl, L.cs_wn);
            }
            a += na;
        }
    }
    if (n_ent > 0) {
        kf_barrier(B);
        KF_STAMP(1);
        /* (2) this workgroup's share of the kept entries */
        const int32_t nf = f, n_calls = min(n_calls_all, WL_MAXCALL);
        const int32_t R = (((n_ent + gwaves - 1) / gwaves) + 63) & ~63;
        const int32_t P = sh.seg[gwaves], c1 = ctx->n_groups > 1 ? ctx->groups[4 + 3] : INT_MAX, gq0 = L.ctot[KF_MAXSEG];
        (void)n_calls;
        /* (four entries per turn, their loads side by side: entry -> root / flags -> the root's key and first call; what the turn
         * stores -- list places, scores, tags, senone marks -- none of these loads reads) */
        int32_t p0 = 0;
        if (P > 0) {
            int32_t lo = 0, hi = gwaves - 1;
            while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (sh.seg[mid] <= 0) lo = mid; else hi = mid - 1; }
            p0 = lo * R;
        }
        /* (the two groups' trees and list lengths once, not per entry: inside the turn they were two dependent loads per entry, under the
         * entry's condition -- eight round trips in a row per turn) */
        const int32_t tg0 = ctx->groups[0], tg1 = ctx->n_groups > 1 ? ctx->groups[4] : ctx->groups[0];
        const int32_t n00 = L.n0[tg0], n01 = L.n0[tg1];
        for (int32_t i0 = gtid; i0 < P; i0 += 4 * gstride) {
            int32_t pq[4], vq[4], flq[4], fsq[4];
            unsigned long long kq[4];
            int4 aq[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int32_t i = i0 + u * gstride;
                pq[u] = -1;
                if (i < P) {
                    int32_t lo = 0, hi = gwaves - 1;
                    while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (sh.seg[mid] <= i) lo = mid; else hi = mid - 1; }
                    pq[u] = lo * R + (i - sh.seg[lo]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) { const int32_t pk = pq[u] >= 0 ? pq[u] : p0; vq[u] = GMC(L.eflag)[pk]; flq[u] = GMC(L.ent)[2 * pk + 1]; }       /* (loads unconditional: see the ranking) */
#pragma unroll
            for (int u = 0; u < 4; u++) {
                kq[u] = S3A_ALD(&GM(L.key)[vq[u]]); fsq[u] = S3A_ALD(&GM(L.first)[vq[u]]);
                aq[u] = kf_mark_load<NE>(vq[u], NE == 3 ? (const int32_t *)S.nodepk : S.nodesen);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (pq[u] < 0) continue;
                const int32_t v = vq[u], fl = flq[u], c = fl & 127;
                const int32_t g = c >= c1 ? 1 : 0, t = g ? tg1 : tg0;
                if (fl & 128) {
                    const int32_t k = (g ? n01 : n00) + (fl >> 8) - (g ? gq0 : 0);
                    GM(L.act[cur])[sh.nb[t] + k] = v; { s3a_v2i o; o.x = k; o.y = nf; *(S3A_AS1 s3a_v2i *)(L.pos + PPX(v)) = o; }
                }
                const unsigned long long key = kq[u];
                const int32_t win_c = 0x7fffffff - (int32_t)(uint32_t)(key & 0xffffffffu);
                if (c == win_c) { GM(L.sc)[NSV(v)] = (int32_t)((uint32_t)(key >> 32) ^ 0x80000000u); GM(L.hist)[NSV(v)] = LM(sh.pool.e1.hist)[c]; }
                if (c == fsq[u]) GM(L.frame)[NSV(v)] = nf;
            }
            /* (the senone marks of the roots this turn listed: the four together, kf_mark_apply4) */
            bool lq[4];
#pragma unroll
            for (int u = 0; u < 4; u++) lq[u] = pq[u] >= 0 && (flq[u] & 128) != 0;
            kf_mark_apply4<NE>(aq, lq, sh.senbits, L.cs_need, nf, L.cs_wl, L.cs_wn);
        }
    }
    kf_barrier(B);
    KF_STAMP(2);
    /* the frame's lists are final: their lengths end to end */
    if (tid == 0) {
        int32_t a = 0;
        for (int32_t t = 0; t < T; t++) { sh.pre[t] = a; a += nact_cur[t]; }
        sh.pre[T] = a;
    }
    /* (lextree_enter's scratch of the roots it touched is clean again -- every entry's key / first call have been read: the launch path
     * sweeps all root nodes in its resolve) */
    if (n_ent > 0) {
        const int32_t R = (((n_ent + gwaves - 1) / gwaves) + 63) & ~63, P = sh.seg[gwaves];
        for (int32_t i = gtid; i < P; i += gstride) {
            int32_t lo = 0, hi = gwaves - 1;
            while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (sh.seg[mid] <= i) lo = mid; else hi = mid - 1; }
            const int32_t v = L.eflag[lo * R + (i - sh.seg[lo])];
            L.key[v] = 0ull; L.first[v] = INT_MAX;
        }
    }
    /* ---- the members of the composite senones wanted in the frame join the mask (ku_comsen_mark): from the frame's list ---- */
    const int32_t n_csw = S3A_ALD(&L.cs_wn[0]);
#ifndef KF_CSU
#define KF_CSU 4                /* composite senones a 16-thread group walks per turn */
#endif
    d_comsen_list<false, KF_CSU>(L.cs_wl, n_csw, S.cs_off, S.cs_list, (uint8_t *)NULL, (const int32_t *)NULL, (int32_t *)NULL, gtid >> 4, gstride >> 4,
                         (const int32_t *)NULL, sh.senbits);
    /* (a cluster's workgroups marked their shares: the masks meet in the lane's words, and every workgroup reads the union back) */
    if (C > 1) {
        __syncthreads();
        for (int32_t i = tid; i < KF_SENBITS / 32; i += KF_NT) if (sh.senbits[i]) atomicOr(&L.senbits[i], sh.senbits[i]);
    }
    kf_barrier(B);
    if (C > 1) {
        for (int32_t i = tid; i < KF_SENBITS / 32; i += KF_NT) sh.senbits[i] = S3A_ALDU(&L.senbits[i]);
        __syncthreads();
    }
    KF_STAMP(3);
    const int32_t n_tot = sh.pre[T];
    const bool hist_frame = n_tot > bm.maxhmmpf + (bm.maxhmmpf >> 1);
    /* ---- approx_cont_mgau_ci_eval / _frame_eval on the window row (ku_select); the launch path's columns of gpart[] that this
     * cluster does not write are made neutral, so that an engine may take either path from frame to frame ---- */
    {
        const int32_t g_all = max(1, min(USEL_G, min(S.gp_n, (S.n_sen - S.n_ci_sen + 255) / 256))), G = min(C, g_all);
        const bool dyn = S.max_cd < S.n_sen - S.n_ci_sen;              /* -maxcdsenpf: the frame's CI beam from the mask and the CI scores */
        if (dyn && r < G) kf_dyn_ci_beam<KF_NT>(S, f, row, sh.senbits, (int32_t *)&sh.pool, &sh.dynbeam);
        if (dyn && r == 0 && tid == 0) L.dynbeam[0] = sh.dynbeam;      /* (where the launch path keeps it: an engine may take either path from frame to frame) */
        if (r < G) d_select<EXACT, KF_NT>(L, S, ctx, f, row, brow, r, G, sh.senbits, dyn ? &sh.dynbeam : (const int32_t *)NULL);
        if (r == 0 && tid >= G && tid < g_all) { L.gpart[tid] = INT_MIN; L.gpart[S.gp_n + tid] = 0; L.gpart[2 * S.gp_n + tid] = 0; }
    }
    kf_barrier(B);
    /* (the mask is consumed: clean for the next frame's marks) */
    for (int32_t i = tid; i < KF_SENBITS / 32; i += KF_NT) sh.senbits[i] = 0u;
    if (C > 1 && r == 0) for (int32_t i = tid; i < KF_SENBITS / 32; i += KF_NT) L.senbits[i] = 0u;
    KF_STAMP(4);
    /* ---- the scores of the composite senones wanted in this frame (ku_comsen_max) ---- */
    d_comsen_list<true, KF_CSU>(L.cs_wl, n_csw, S.cs_off, S.cs_list, (uint8_t *)NULL, row, L.cs_val, gtid >> 4, gstride >> 4, S.cs_wt);
    kf_barrier(B);
    KF_STAMP(5);
    /* ---- lextree_hmm_eval (ku_hmm_eval): a thread per list position of the trees laid end to end; the per-tree maxima are
     * gathered in LDS and leave the workgroup as one atomic per tree ---- */
    {
        int32_t gb = INT_MIN;
        for (int32_t i = tid; i < S.gp_n; i += KF_NT) gb = max(gb, L.gpart[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gb = max(gb, __shfl_xor(gb, o, 64));
        if (lane == 0) sh.red[wave] = gb;
        if (tid < 2 * T) sh.acc[tid] = INT_MIN;
        if (r == 0 && tid == 0) L.cs_wn[0] = 0;                 /* (the frame's list of composite senones is consumed) */
        /* the frame's senone scores: into the pool when they fit (no other step's data lives there now) */
        const bool row_in_lds = (size_t)S.n_sen * 4 <= sizeof(KfPool);
        int32_t *row_lds = (int32_t *)&sh.pool;
        if (row_in_lds)
            for (int32_t i = tid; i < S.n_sen; i += KF_NT) row_lds[i] = row[i];
        __syncthreads();
        int32_t norm = max(L.misc[0], L.misc[5]);               /* the frame's normaliser */
        for (int w = 0; w < KF_WAVES; w++) norm = max(norm, sh.red[w]);
        const int32_t *act = L.act[cur];
        const bool tp_in_lds = S.n_tmat * NS_TPW(NE) <= KF_TP_LDS;
        /* (the chain list position -> node -> its static words -> senone ids -> scores: the node is asked for two turns ahead, its
         * static words one turn ahead, so that a turn only waits for the ids and the scores) */
        int32_t t0 = -1, i0 = 0, b0 = 0, v0 = -1, t1 = -1, i1 = 0, b1 = 0, v1 = -1;
        int4 nd0 = make_int4(0, 0, 0, 0);
        {
            const int32_t ga = r * KF_NT + tid, gb_ = ga + gstride;
            if (ga < n_tot) { kf_locate(sh.pre, T, ga, t0, i0); b0 = sh.nb[t0]; v0 = act[b0 + i0]; }
            if (gb_ < n_tot) { kf_locate(sh.pre, T, gb_, t1, i1); b1 = sh.nb[t1]; v1 = act[b1 + i1]; }
            if (v0 >= 0) nd0 = (NE == 3 ? S.nodepk : S.node4)[v0];
        }
        for (int32_t g0 = r * KF_NT; g0 < n_tot; g0 += gstride) {
            /* (unconditional, a thread past the end asks for position 0 / node 0: a load under a condition is a branch with its own wait) */
            int32_t t2 = -1, i2 = 0, b2 = 0, v2 = -1;
            const int32_t gc = g0 + 2 * gstride + tid;
            kf_locate(sh.pre, T, gc < n_tot ? gc : 0, t2, i2);          /* (the LDS reads first: a wait for one waits for every flat load in flight) */
            b2 = sh.nb[t2];
            int4 nd1;
            { const s3a_v4i a = ((const S3A_AS1 s3a_v4i *)(NE == 3 ? S.nodepk : S.node4))[max(v1, 0)]; nd1 = make_int4(a.x, a.y, a.z, a.w); }
            {
                const int32_t vx = GMC(act)[b2 + i2];
                if (gc < n_tot) v2 = vx; else t2 = -1;
            }
            const int32_t t = t0;
            int32_t k = INT_MIN, w = -1;
            if (v0 >= 0) {
                int32_t out;
                k = kf_hmm_eval<NE>(v0, nd0, S.nodesen, S.tp, sh.tp, tp_in_lds, row, row_lds, row_in_lds, norm, L.cs_val, L.sc, f, w, out);
                GM(L.poswid)[b0 + i0] = w;
                GM(L.posout)[b0 + i0] = out;
                GM(L.posbest)[b0 + i0] = k;
                if (NE == 3) GM(L.posps)[b0 + i0] = (int32_t)((uint32_t)nd0.w >> 1);           /* (parent set + 1, bit 29: member of a several-parent set) */
            }
            /* a wave's 64 positions belong to one tree, or to two or three at the seams */
            unsigned long long todo = __ballot(t >= 0);
            while (todo) {
                const int32_t tt = __shfl(t, __ffsll((long long)todo) - 1, 64);
                const bool mine = t == tt;
                int32_t x = mine ? k : INT_MIN, y = (mine && w >= 0) ? k : INT_MIN;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { x = max(x, __shfl_xor(x, o, 64)); y = max(y, __shfl_xor(y, o, 64)); }
                if (lane == 0) { atomicMax(&sh.acc[2 * tt], x); if (y != INT_MIN) atomicMax(&sh.acc[2 * tt + 1], y); }
                todo &= ~__ballot(mine);
            }
            t0 = t1; i0 = i1; b0 = b1; v0 = v1; nd0 = nd1;
            t1 = t2; i1 = i2; b1 = b2; v1 = v2;
        }
        __syncthreads();
        if (tid < 2 * T && sh.acc[tid] != INT_MIN) atomicMax(&L.best[tid], sh.acc[tid]);
    }
    kf_barrier(B);
    KF_STAMP(6);
    /* the frame's per-tree maxima: final; every later phase reads this copy */
    if (tid < 2 * T) sh.best[tid] = S3A_ALD(&L.best[tid]);
    if (r == 0 && tid == 0) L.pcnt[(f + 1) & 1] = 0;           /* (the coming frame's list of stamped parent sets) */
    __syncthreads();
    /* ---- lextree_hmm_histbin + the histogram beam (frames over 1.5 x -maxhmmpf), else the stamps of the HMMs that can
     * propagate and the list of stamped parent sets (ku_hist_count / ku_hist_sort) ---- */
    if (hist_frame) {
        for (int32_t t = 0; t < T; t++)
            for (int32_t vb = r; vb * KF_NT < nact_cur[t]; vb += C) {
                d_dec_hist_count_t<KF_NT>(S.node_base, L.act[cur], L.nact[cur], T, bm, sh.best, L.bests, L.exits + S.N, L.hbin, -1, 0, 1, NBIN, vb, t, sh.pool.bin);
                __syncthreads();
            }
        kf_barrier(B);
        if (r == 0)
            for (int32_t t = 0; t < T; t++) {
                const int32_t hb = d_dec_hist_sort_ws<KF_NT>(S.node_base, L.act[cur], L.nact[cur], T, bm, L.exits + S.N, L.exits, L.hbin, L.pos, -1, NBIN, t, 0, sh.pool.hs);
                __syncthreads();
                if (hb <= 0) {
                    int32_t th, pth;
                    frame_thresholds_hb(sh.best, T, bm, hb, th, pth);
                    d_stamp_and_list(L, S, cur, t, nact_cur[t], pth, f, 0, KF_NT, 1);
                }
                __syncthreads();
            }
    }
    else {
        /* (d_stamp_and_list with the exit scores by list position: coalesced, and the few HMMs that propagate then chase their node) */
        int32_t th, pth;
        frame_thresholds_hb(sh.best, T, bm, 1, th, pth);
        int32_t *pc = &L.pcnt[f & 1];
        /* (KF_SK positions per thread and pass, stage by stage -- exit scores, the passing HMMs' nodes, their ranges of parent sets, then
         * set after set: one position per pass was seven passes of up to six dependent round trips; the list's counter is taken once
         * per wave and stage) */
        int32_t ix0 = 0;
        if (n_tot > 0) { int32_t t, i; kf_locate(sh.pre, T, 0, t, i); ix0 = sh.nb[t] + i; }       /* (a list position that exists: the loads of threads past the end) */
        for (int32_t gb = 0; gb < n_tot; gb += KF_SK * gstride) {
            int32_t ix[KF_SK], q0[KF_SK], q1[KF_SK];
            bool ok[KF_SK];
            /* (stage by stage, and no LDS read between the loads of a stage: these are flat loads -- counted by the LDS counter too --,
             * so waiting for an LDS read waits for all of them) */
#pragma unroll
            for (int k = 0; k < KF_SK; k++) {
                const int32_t g = gb + k * gstride + gtid;
                ok[k] = g < n_tot; ix[k] = ix0;
                if (ok[k]) { int32_t t, i; kf_locate(sh.pre, T, g, t, i); ix[k] = sh.nb[t] + i; }
            }
            int32_t po[KF_SK];
#pragma unroll
            for (int k = 0; k < KF_SK; k++) po[k] = GMC(L.posout)[ix[k]];
#pragma unroll
            for (int k = 0; k < KF_SK; k++) ix[k] = GMC(L.act[cur])[ix[k]];
#pragma unroll
            for (int k = 0; k < KF_SK; k++) ok[k] = ok[k] && po[k] >= pth;
            int32_t nq = 0;
            {
                int32_t a0[KF_SK], a1[KF_SK];
#pragma unroll
                for (int k = 0; k < KF_SK; k++) { a0[k] = GMC(S.psof_off)[ix[k]]; a1[k] = GMC(S.psof_off)[ix[k] + 1]; }
#pragma unroll
                for (int k = 0; k < KF_SK; k++) { q0[k] = ok[k] ? a0[k] : 0; q1[k] = ok[k] ? a1[k] : 0; nq = max(nq, q1[k] - q0[k]); }
            }
            (void)nq;
            /* the passing HMMs' (first set, number of sets) pairs go to a list in LDS, and the workgroup walks all their sets as flat
             * work items -- a thread per (HMM, set) --: a propagating root variant stamps dozens of sets, and a thread that walked its
             * own HMMs' sets one after the other held its wave for ~11 turns of three dependent round trips each (measured) */
            auto &rs = sh.pool.rs;
            if (tid == 0) rs.nbig = 0;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < KF_SK; k++) {
                const int32_t cnt = q1[k] - q0[k];
                if (cnt <= 0) continue;
                const int32_t slot = atomicAdd(&rs.nbig, 1);
                if (slot < KF_SETS) { rs.mlo[slot] = q0[k]; rs.big[slot] = cnt; }
                else                /* (no room: this thread walks them itself) */
                    for (int32_t q = q0[k]; q < q1[k]; q++) {
                        const int32_t ps = GMC(S.psof)[q];
                        GM(L.pstamp8)[ps] = ps_val<uint8_t>(f);
                        if (atomicExch(&L.claim[ps], f) != f) L.plist[atomicAdd(pc, 1)] = ps;
                    }
            }
            __syncthreads();
            const int32_t n_e = min(rs.nbig, KF_SETS);
            {
                static_assert(KF_SETS == KF_NT, "an entry per thread");
                const int32_t x = tid < n_e ? rs.big[tid] : 0;
                int32_t incl = x;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const int32_t y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
                if (lane == 63) sh.ws[wave] = incl;
                __syncthreads();
                int32_t add = 0;
                for (int32_t w = 0; w < wave; w++) add += sh.ws[w];
                rs.pre[tid] = add + incl - x;
                if (tid == KF_NT - 1) rs.m_all = add + incl;
                __syncthreads();
            }
            const int32_t I = rs.m_all;
            for (int32_t it0 = 0; it0 < I; it0 += KF_NT) {
                const int32_t it = it0 + tid;
                int32_t lo = 0, hi = max(n_e - 1, 0);
                while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (rs.pre[mid] <= it) lo = mid; else hi = mid - 1; }
                const int32_t qx = it < I ? rs.mlo[lo] + (it - rs.pre[lo]) : rs.mlo[0];
                const int32_t psx = GMC(S.psof)[qx];
                const int32_t ps = it < I ? psx : -1;
                int32_t old = f;
                if (ps >= 0) { GM(L.pstamp8)[ps] = ps_val<uint8_t>(f); old = atomicExch(&L.claim[ps], f); }
                const unsigned long long mm = __ballot(old != f);
                if (mm) {
                    int32_t at = 0;
                    if (lane == __ffsll((long long)mm) - 1) at = atomicAdd(pc, __popcll(mm));
                    const int32_t base = __shfl(at, __ffsll((long long)mm) - 1, 64);
                    if (old != f) GM(L.plist)[base + __popcll(mm & ((1ull << lane) - 1ull))] = ps;
                }
            }
            __syncthreads();
        }
    }
    kf_barrier(B);
    KF_STAMP(7);
    /* ---- -ptranskip frames / -pbeam wider than -beam: the weak HMMs that a parent earlier in the list re-entered (ku_weak) ---- */
    if (weak_possible && (bm.phone_uses_wbeam || bm.pbeam < bm.hmmbeam)) {
        if (HEUR) {
            /* ... together with the look-ahead (ku_weak_heur): which weak HMMs a parent re-entered AND the heuristic thresholds by list
             * position, a tree per workgroup in turn */
            static_assert(KF_WAVES <= 16, "d_weak_heur_t's per-wave words");
            for (int32_t t = r; t < T; t += C) d_weak_heur_t<KF_NT>(L, S, f, cur, nact_cur, sh.best, t, sh.ws, sh.seg, sh.gq);
        }
        else if (r == 0)
            d_dec_weak_t<KF_NT>(S.N, T, f, bm, sh.best, L.nact[cur], S.node_base, L.act[cur], S.prob, S.par_off, S.par, L.pos, L.posf, L.sc, L.outs,
                                L.bests, S.wid, L.hbin, L.propf, L.exits + 2 * (size_t)S.N, 0, 0);
        kf_barrier(B);
        KF_STAMP(8);
    }
    /* ---- lextree_hmm_propagate_non_leaves from the node's point of view (ku_resolve_plist) ---- */
    {
        if (tid == 0) {
            int32_t bh, bw, n, th, pth, wth;
            (void)frame_thresholds(sh.best, nact_cur, T, bm, L.hbin, bh, bw, n, th, pth, wth);
            sh.thr[0] = th; sh.thr[1] = pth; sh.thr[2] = wth;
        }
        __syncthreads();
        if (r == 0 && hist_frame)
            for (int32_t i = tid; i < NBIN; i += KF_NT) L.hbin[i] = 0;         /* (the bins were consumed by the sort; hbin[NBIN] stays) */
        const int32_t *act = L.act[cur];
        const HeurArgs hx = HEUR ? HeurArgs{ S.node_ci, L.heur_all + (size_t)f * S.n_ci, L.hth_pos } : HeurArgs{ NULL, NULL, NULL };
        if (HEUR && !(weak_possible && (bm.phone_uses_wbeam || bm.pbeam < bm.hmmbeam))) {       /* (a frame with weak HMMs: d_weak_heur_t made them) */
            /* the heuristic threshold of every propagating HMM by list position (ku_heur_thresh; lextree.c:1443-1462): per tree the running
             * maximum over the list of max over children (out + (prob(child) - prob) + phn_heur[ci(child)]), plus pl_beam -- a tree per
             * workgroup in turn, KF_NT positions per pass, the passes' carry through LDS */
            const int32_t pth_ = sh.thr[1];
            for (int32_t t = r; t < T; t += C) {
                const int32_t na = nact_cur[t], b = sh.nb[t];
                if (tid == 0) sh.gq[2] = INT_MIN;
                __syncthreads();
                for (int32_t i0 = 0; i0 < na; i0 += KF_NT) {
                    const int32_t i = i0 + tid;
                    int32_t m = INT_MIN;
                    if (i < na) {
                        const int32_t p = act[b + i], po = L.outs[NSV(p)];
                        if (S.wid[p] < 0 && po >= pth_) {
                            const int32_t pp = S.prob[p];
                            for (int32_t q = S.child_off[p]; q < S.child_off[p + 1]; q++) {
                                const int32_t c = S.child[q];
                                m = max(m, add32(add32(po, add32(S.prob[c], -pp)), hx.heur[hx.node_ci[c]]));
                            }
                        }
                    }
                    int32_t x = m;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) { const int32_t y = __shfl_up(x, o, 64); if (lane >= o) x = max(x, y); }
                    if (lane == 63) sh.ws[wave] = x;
                    __syncthreads();
                    int32_t pre = sh.gq[2];
                    for (int32_t w = 0; w < wave; w++) pre = max(pre, sh.ws[w]);
                    x = max(x, pre);
                    if (i < na) L.hth_pos[b + i] = add32(x, S.pl_beam);
                    __syncthreads();
                    if (tid == KF_NT - 1) sh.gq[2] = x;
                    __syncthreads();
                }
            }
            kf_barrier(B);
        }
        /* the frame's stamped parent sets as a filter in LDS (every stamp was listed: d_stamp_and_list and the stamping pass above), so
         * that the usual HMM -- nobody stamped its set -- is settled without a visit to pstamp8; a 3-state HMM's set id came with its
         * packed node and lies by list position (a histogram frame has reordered the positions: through the node then) */
        {
            for (int32_t i = tid; i < KF_PSBITS / 32; i += KF_NT) sh.psbits[i] = 0u;
            for (int32_t i = tid; i < KF_PSTAB; i += KF_NT) sh.pstab[i] = -1;
            __syncthreads();
            const int32_t n_pl = S3A_ALD(&L.pcnt[f & 1]);
            if (tid == 0) sh.ps_exact = n_pl <= (3 * KF_PSTAB) / 4 ? 1 : 0;
            for (int32_t k = tid; k < n_pl; k += KF_NT) {
                const int32_t q_ = L.plist[k];
                const uint32_t q = (uint32_t)q_ % KF_PSBITS;
                atomicOr(&sh.psbits[q >> 5], 1u << (q & 31));
                if (n_pl <= (3 * KF_PSTAB) / 4)         /* (every listed set once: no two threads insert the same id) */
                    for (uint32_t sl = ((uint32_t)q_ * 2654435761u) >> 23; ; sl = (sl + 1) & (KF_PSTAB - 1))
                        if (atomicCAS(&sh.pstab[sl], -1, q_) == -1) break;
            }
            __syncthreads();
        }
        const bool ps_by_pos = NE == 3 && !hist_frame;
        /* the active HMMs by list position, in two passes.  (A) KF_RL positions per thread and turn, their words asked for together:
         * the usual HMM -- nobody stamped its parent set, and it survives -- is settled at once (it joins the next list at its own turn;
         * its record carries the frame tag since the evaluation); the others -- entered by a parent, or dying -- go to a list in LDS.
         * (B) a thread per listed HMM walks d_dec_resolve_node's chain: dense waves, once per block, instead of every wave paying the
         * chain in every turn for the few lanes that need it (two positions per turn were 3.2 turns x ~11 round trips) */
#ifdef KF_DIAG
        long long td_ = (long long)wall_clock64();
#define KF_DT(n) do { __syncthreads(); if (KF_DIAG == (n) && r == 0 && tid == 0) { const long long t_ = (long long)wall_clock64(); sh.kacc[8] += t_ - td_; } if (r == 0 && tid == 0) td_ = (long long)wall_clock64(); } while (0)
#else
#define KF_DT(n) do { } while (0)
#endif
        {
            int32_t *wl_ = (int32_t *)&sh.pool;
            static_assert(sizeof(KfPool) >= KF_RL * KF_NT * 4, "the list pass's work list lives in the pool");
            for (int32_t g00 = r * KF_RL * KF_NT; g00 < n_tot; g00 += KF_RL * gstride) {
                if (tid == 0) sh.gq[3] = 0;
                __syncthreads();
                {
                    int32_t gg[KF_RL], ii[KF_RL], bb[KF_RL], vv[KF_RL], qq[KF_RL], pbv[KF_RL];
#pragma unroll
                    for (int u = 0; u < KF_RL; u++) {       /* (the LDS reads first: a wait for one waits for every flat load in flight) */
                        int32_t t_;
                        gg[u] = g00 + u * KF_NT + tid;
                        kf_locate(sh.pre, T, gg[u] < n_tot ? gg[u] : 0, t_, ii[u]);
                        bb[u] = sh.nb[t_];
                    }
#pragma unroll
                    for (int u = 0; u < KF_RL; u++) {       /* (loads unconditional, a thread past the end asks for position 0) */
                        const int32_t vx = GMC(act)[bb[u] + ii[u]];
                        vv[u] = gg[u] < n_tot ? vx : -1;
                        qq[u] = ps_by_pos ? GMC(L.posps)[bb[u] + ii[u]] : 0;
                        pbv[u] = GMC(L.posbest)[bb[u] + ii[u]];
                    }
                    if (!ps_by_pos) {
#pragma unroll
                        for (int u = 0; u < KF_RL; u++) qq[u] = GMC(S.ps)[max(vv[u], 0)];
                    }
                    const bool exact = ps_by_pos && sh.ps_exact != 0;       /* (the stamped sets' table is complete: membership decides) */
#pragma unroll
                    for (int u = 0; u < KF_RL; u++) {
                        if (vv[u] < 0) continue;
                        /* (by position: parent set + 1, bit 29 = a member of a several-parent set; through the node: the set id) */
                        const int32_t q = ps_by_pos ? (qq[u] & 0x1fffffff) - 1 : qq[u];
                        const bool in_set = ps_by_pos && ((qq[u] >> 29) & 1);
                        bool has_par = false;
                        if (q >= 0) {
                            if (exact) {
                                for (uint32_t sl = ((uint32_t)q * 2654435761u) >> 23; ; sl = (sl + 1) & (KF_PSTAB - 1)) {
                                    const int32_t x = sh.pstab[sl];
                                    if (x == q) { has_par = true; break; }
                                    if (x == -1) break;
                                }
                            }
                            else {
                                const uint32_t qb = (uint32_t)q % KF_PSBITS;
                                has_par = ((sh.psbits[qb >> 5] >> (qb & 31)) & 1u) != 0;
                                if (has_par) has_par = GMC(L.pstamp8)[q] == ps_val<uint8_t>(f);
                            }
                        }
                        if (!hist_frame && !has_par) {
                            /* the usual active HMM -- no parent can enter it: it survives (it joins the next list at its own turn; its record
                             * carries the frame tag since the evaluation) or it is cleared (d_dec_resolve_node's first case: stores only) */
                            if (pbv[u] >= sh.thr[0]) { GM(L.selfemit)[bb[u] + ii[u]] = 1; atomicAdd(&L.cnt[bb[u] + ii[u]], 1); continue; }
                            if (NE == 3) {
                                S3A_AS1 s3a_v4i *rec = (S3A_AS1 s3a_v4i *)(L.sc + NSV(vv[u]));
                                s3a_v4i o0, o1; s3a_v2i o2;
                                o0.x = WORST; o0.y = WORST; o0.z = WORST; o0.w = -1; o1.x = -1; o1.y = -1; o1.z = WORST; o1.w = -1; o2.x = WORST; o2.y = -1;
                                static_assert(NS_HIST(3) == 3 && NS_OUTS(3) == 6 && NS_OUTH(3) == 7 && NS_BESTS(3) == 8 && NS_FRAME(3) == 9, "the 3-state record's layout");
                                rec[0] = o0; rec[1] = o1; *(S3A_AS1 s3a_v2i *)(rec + 2) = o2;
                                GM(L.posout)[bb[u] + ii[u]] = WORST;
                                continue;
                            }
                        }
                        /* (the listed sets' members with 2..64 parents, active or not, are the set passes': with the exact table a stamped set IS a
                         * listed set) */
                        if (exact && has_par && in_set && !hist_frame) continue;
                        wl_[atomicAdd(&sh.gq[3], 1)] = gg[u] | (has_par ? (int32_t)0x80000000 : 0);
                    }
                }
                __syncthreads();
                const int32_t n_wl = sh.gq[3];
#if defined(KF_DIAG) && KF_DIAG == 1
                if (r == 0 && tid == 0) sh.kacc[8] += (long long)n_wl * 100;
#endif
                for (int32_t k = tid; k < n_wl; k += KF_NT) {
                    const int32_t e_ = wl_[k], g = e_ & 0x7fffffff;
                    const bool has_par = e_ < 0;
                    int32_t t_, i;
                    kf_locate(sh.pre, T, g, t_, i);
                    const int32_t b = sh.nb[t_], v = GMC(act)[b + i];
                    /* (the listed sets' members with 2..64 parents, active or not, are d_dec_resolve_children's) */
                    if (has_par) {
                        const int32_t q = ps_by_pos ? (GMC(L.posps)[b + i] & 0x1fffffff) - 1 : GMC(S.ps)[v];
                        if (S3A_ALD(&L.claim[q]) == f) {
                            const int32_t np = S.par_off[v + 1] - S.par_off[v];
                            if (np >= SET_NP_MIN && np <= 64) continue;
                        }
                    }
                    d_dec_resolve_node<uint8_t, HEUR>(S.N, T, f, bm, sh.best, nact_cur, S.node_base, S.tree_of, S.prob, S.par_off, S.par, L.pos, L.posf,
                                                      L.sc, L.hist, L.outs, L.outh, L.bests, L.frame, L.turn, L.selfemit, L.cnt, L.key, L.first, L.hbin,
                                                      S.ps, L.pstamp8, S.rootnodes, S.n_rootnodes, L.propf, L.posout, v, true, has_par, i, b,
                                                      hx, sh.thr);
                }
                __syncthreads();
            }
        }
        KF_DT(5);
        /* the frame's listed parent sets (d_stamp_and_list): this workgroup's share, up to KF_SETS per pass -- everything as flat work
         * items (measured: ~130 listed sets per frame, 87 of them several-parent sets of ~10 members and one or two propagating
         * parents each; a wave per such set was 11 sets in a row per wave, each a chain of five round trips: 98 us of the frame):
         *   1. the sets' headers, all at once;
         *   2. the one-parent sets' members (an interior node's children), a thread each;
         *   3. a thread per (several-parent set, parent): the PROPAGATING parents go to the set's table in LDS (exit score, list
         *      position, exit history, probability: up to KF_TQ; a set with more goes the wave-per-set way);
         *   4. a thread per (several-parent set, member), on the list or not: d_dec_resolve_children's rule with the table. */
        {
            auto &rs = sh.pool.rs;
            const int32_t th = sh.thr[0], pth = sh.thr[1];
            const int32_t n_pl = S3A_ALD(&L.pcnt[f & 1]), per = (n_pl + C - 1) / C, k_lo = min(n_pl, r * per), k_hi = min(n_pl, k_lo + per);
            for (int32_t k0 = k_lo; k0 < k_hi; k0 += KF_SETS) {
                const int32_t nk = min(KF_SETS, k_hi - k0);
                if (tid == 0) { rs.nbig = 0; rs.nleg = 0; rs.n_ent = 0; }
                __syncthreads();
                {
                    static_assert(KF_SETS == KF_NT, "a set per thread");
                    int32_t cm = 0, m_lo = 0;
                    if (tid < nk) {
                        const int32_t q = GMC(L.plist)[k0 + tid];
#ifndef KF_PSHDR
#define KF_PSHDR 1
#endif
#if KF_PSHDR
                        const s3a_v4i h_ = *(const S3A_AS1 s3a_v4i *)(S.pshdr + q);      /* (the set's header in one word: two dependent gathers less than the chain below) */
                        m_lo = h_.x;
                        const int32_t m_hi = h_.y, kp0 = h_.z, np = h_.w;
#else
                        m_lo = GMC(S.psmem_off)[q];
                        const int32_t m_hi = GMC(S.psmem_off)[q + 1], x0 = GMC(S.psmem)[m_lo], kp0 = GMC(S.par_off)[x0], np = GMC(S.par_off)[x0 + 1] - kp0;
#endif
                        if (np >= SET_NP_MIN && np <= 64) {
                            const int32_t at = atomicAdd(&rs.nbig, 1);
                            rs.big[at] = q;
                            if (at < KF_BIG) { rs.bmlo[at] = m_lo; rs.bnm[at] = m_hi - m_lo; rs.bkp0[at] = kp0; rs.bnp[at] = np; rs.bnq[at] = -1; }
                        }
                        else cm = m_hi - m_lo;
                    }
                    rs.mlo[tid] = m_lo; rs.pre[tid] = cm;
                }
                __syncthreads();
                const int32_t nb = min(rs.nbig, KF_BIG);
                {   /* exclusive sums: the one-parent sets' members | the several-parent sets' parents | their members */
                    const int32_t a0 = rs.pre[tid], pn = tid < nb ? rs.bnp[tid] : 0, pm = tid < nb ? rs.bnm[tid] : 0;
                    int32_t i0 = a0, i1 = pn, i2 = pm;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const int32_t y0 = __shfl_up(i0, o, 64), y1 = __shfl_up(i1, o, 64), y2 = __shfl_up(i2, o, 64);
                        if (lane >= o) { i0 += y0; i1 += y1; i2 += y2; }
                    }
                    if (lane == 63) { sh.ws[wave] = i0; sh.seg[wave] = i1; sh.seg[16 + wave] = i2; }
                    __syncthreads();
                    int32_t a_ = 0, b_ = 0, c_ = 0;
                    for (int32_t w = 0; w < wave; w++) { a_ += sh.ws[w]; b_ += sh.seg[w]; c_ += sh.seg[16 + w]; }
                    rs.pre[tid] = a_ + i0 - a0;
                    if (tid < KF_BIG) { rs.bpre[tid] = b_ + i1 - pn; rs.bmpre[tid] = c_ + i2 - pm; }
                    if (tid == KF_NT - 1) { rs.m_all = a_ + i0; rs.bpre[KF_BIG] = b_ + i1; rs.bmpre[KF_BIG] = c_ + i2; }
                    __syncthreads();
                }
                /* 2. the one-parent sets' members */
                const int32_t M = rs.m_all;
#if defined(KF_DIAG) && KF_DIAG == 2
                if (r == 0 && tid == 0) sh.kacc[8] += (long long)M * 100;
#endif
#if defined(KF_DIAG) && KF_DIAG == 3
                if (r == 0 && tid == 0) sh.kacc[8] += (long long)(rs.bpre[KF_BIG] + rs.bmpre[KF_BIG]) * 100;
#endif
#if defined(KF_DIAG) && KF_DIAG == 4
                if (r == 0 && tid == 0) sh.kacc[8] += (long long)(rs.nleg + max(0, rs.nbig - KF_BIG)) * 100;
#endif
                for (int32_t m = tid; m < M; m += KF_NT) {
                    int32_t lo = 0, hi = nk - 1;
                    while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (rs.pre[mid] <= m) lo = mid; else hi = mid - 1; }
                    const int32_t x = GMC(S.psmem)[rs.mlo[lo] + (m - rs.pre[lo])];
                    if (L.posf[PPX(x)] == f) continue;                               /* on the list: resolved by list position */
                    d_dec_resolve_node<uint8_t, HEUR>(S.N, T, f, bm, sh.best, nact_cur, S.node_base, S.tree_of, S.prob, S.par_off, S.par, L.pos, L.posf,
                                                      L.sc, L.hist, L.outs, L.outh, L.bests, L.frame, L.turn, L.selfemit, L.cnt, L.key, L.first, L.hbin,
                                                      S.ps, L.pstamp8, S.rootnodes, S.n_rootnodes, L.propf, L.posout, x, false, true, -1, -1,
                                                      hx, sh.thr);
                }
                KF_DT(6);
                /* 3. the several-parent sets' propagating parents */
                {
                    const int32_t I = rs.bpre[KF_BIG];
                    for (int32_t it0 = 0; it0 < I; it0 += 2 * KF_NT) {
                        int32_t sb[2], gp[2], pfv[2];
#pragma unroll
                        for (int u = 0; u < 2; u++) {
                            const int32_t it = it0 + u * KF_NT + tid;
                            sb[u] = -1; gp[u] = -1;
                            if (it < I) {
                                int32_t lo = 0, hi = nb - 1;
                                while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (rs.bpre[mid] <= it) lo = mid; else hi = mid - 1; }
                                sb[u] = lo; gp[u] = GMC(S.par)[rs.bkp0[lo] + (it - rs.bpre[lo])];
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 2; u++) pfv[u] = gp[u] >= 0 ? GMC(L.posf)[PPX(gp[u])] : INT_MIN;
#pragma unroll
                        for (int u = 0; u < 2; u++) {
                            if (pfv[u] != f) continue;
                            const int32_t g = gp[u], po = GMC(L.outs)[NSV(g)];
                            if (po < pth || (pth < th && GMC(L.bests)[NSV(g)] < th && GMC(L.propf)[g] != f)) continue;
                            const int32_t at = atomicAdd(&rs.n_ent, 1);
                            if (at < KF_ENT) {
                                int32_t *e = rs.ent[at];
                                e[0] = po; e[1] = GMC(L.pos)[PPX(g)]; e[2] = GMC(L.outh)[NSV(g)]; e[3] = GMC(S.prob)[g];
                                e[4] = atomicExch(&rs.bnq[sb[u]], at);              /* (the chain's order does not matter: maxima with position tie-breaks) */
                            }
                            else atomicMin(&rs.bnq[sb[u]], -2);
                        }
                    }
                    __syncthreads();
                    /* (a set one of whose parents found no room: its chain may be cut short -- the wave-per-set way) */
                    if (tid < nb && rs.n_ent > KF_ENT) {
                        bool cut = rs.bnq[tid] == -2;
                        for (int32_t e = rs.bnq[tid]; e >= 0 && !cut; e = rs.ent[e][4]) cut = rs.ent[e][4] == -2;
                        if (cut) { rs.bnq[tid] = -2; rs.leg[atomicAdd(&rs.nleg, 1)] = tid; }
                    }
                    __syncthreads();
                }
                KF_DT(7);
                /* 4. their members */
                {
                    const int32_t MB = rs.bmpre[KF_BIG];
                    for (int32_t m = tid; m < MB; m += KF_NT) {
                        int32_t lo = 0, hi = nb - 1;
                        while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (rs.bmpre[mid] <= m) lo = mid; else hi = mid - 1; }
                        const int32_t k = lo, e0 = rs.bnq[k];
                        if (e0 == -2) continue;                                 /* the wave-per-set way, below */
                        const int32_t x = GMC(S.psmem)[rs.bmlo[k] + (m - rs.bmpre[k])];
                        const bool on_list = GMC(L.posf)[PPX(x)] == f;                    /* (the list position pass leaves these members to us) */
                        if (!on_list && e0 < 0) continue;
                        const int32_t j = on_list ? GMC(L.pos)[PPX(x)] : INT_MAX, in0 = GMC(L.sc)[NSV(x)], px = GMC(S.prob)[x], b = sh.nb[GMC(S.tree_of)[x]];
                        const int32_t hv = HEUR ? hx.heur[hx.node_ci[x]] : 0;
                        int32_t mE = INT_MIN, pE = INT_MAX, hE = -1, firstE = INT_MAX;
                        int32_t mL = INT_MIN, pL = INT_MAX, hL = -1, firstL = INT_MAX;
                        for (int32_t q = e0; q >= 0; q = rs.ent[q][4]) {
                            const int32_t *e = rs.ent[q];
                            const int32_t ns = add32(e[0], add32(px, -e[3]));
                            if (ns < th) continue;
                            const int32_t up = e[1];
                            if (HEUR && add32(ns, hv) < GMC(L.hth_pos)[b + up]) continue;       /* (the parent's threshold, by ITS list position) */
                            if (up < j) {
                                if (ns > mE || (ns == mE && up < pE)) { mE = ns; pE = up; hE = e[2]; }
                                if (ns > in0 && up < firstE) firstE = up;
                            }
                            else {
                                if (ns > mL || (ns == mL && up < pL)) { mL = ns; pL = up; hL = e[2]; }
                                if (up < firstL) firstL = up;
                            }
                        }
                        d_dec_resolve_finish(S.N, f, th, b, x, on_list, j, in0, mE, hE, firstE, mL, hL, firstL,
                                             L.sc, L.hist, L.outs, L.outh, L.bests, L.frame, L.turn, L.selfemit, L.cnt, L.posout);
                    }
                }
                KF_DT(8);
                /* ... and the sets that have no table, a wave each (d_dec_resolve_children) */
                if (rs.nleg > 0 || rs.nbig > KF_BIG) {
                    __syncthreads();
                    if (tid < rs.nleg) sh.seg[32 + tid] = rs.big[rs.leg[tid]];
                    __syncthreads();
                    d_dec_resolve_children<uint8_t, HEUR>(S.N, T, f, bm, sh.best, nact_cur, S.node_base, S.tree_of, S.prob, S.par_off, S.par, L.pos, L.posf,
                                                          L.sc, L.hist, L.outs, L.outh, L.bests, L.frame, L.turn, L.selfemit, L.cnt, L.key, L.first, L.hbin,
                                                          S.ps, L.pstamp8, S.rootnodes, S.n_rootnodes, L.propf, L.posout, sh.seg + 32, rs.nleg,
                                                          S.psmem_off, S.psmem, wave, KF_WAVES, hx, rs.rc[wave], sh.thr);
                    if (rs.nbig > KF_BIG)
                        d_dec_resolve_children<uint8_t, HEUR>(S.N, T, f, bm, sh.best, nact_cur, S.node_base, S.tree_of, S.prob, S.par_off, S.par, L.pos, L.posf,
                                                              L.sc, L.hist, L.outs, L.outh, L.bests, L.frame, L.turn, L.selfemit, L.cnt, L.key, L.first, L.hbin,
                                                              S.ps, L.pstamp8, S.rootnodes, S.n_rootnodes, L.propf, L.posout, rs.big + KF_BIG, rs.nbig - KF_BIG,
                                                              S.psmem_off, S.psmem, wave, KF_WAVES, hx, rs.rc[wave], sh.thr);
                }
                __syncthreads();
            }
        }
    }
    kf_barrier(B);
    KF_STAMP(9);
    /* ---- the ordered compaction of the next list and of the word exits (ku_scan): a tree per workgroup in turn ---- */
    for (int32_t t = r; t < T; t += C) {
        d_dec_scan_t<KF_NT>(S.N, T, f, bm, S.node_base, L.act[cur], L.nact[cur], S.wid, S.prob, L.outs, L.outh, L.selfemit, L.cnt, L.base,
                            L.act[cur ^ 1], L.nact[cur ^ 1], L.pos, L.posf, sh.best, L.exits, L.nexit, L.hbin, L.misc, (int32_t *)NULL, L.pack,
                            S.pack_max_exits, L.gpart, S.gp_n, L.poswid, L.posout, hist_frame ? 1 : 0, L.scan_agg, L.scan_pre, L.scan_flag,
                            S.scan_chunks, 0, 1, 1, t, 0);
        __syncthreads();
    }
    kf_barrier(B);
    KF_STAMP(10);
    /* ---- the emission of the next list (every workgroup but the first when there are several) and the word level (the
     * first), which closes the frame and leaves the next frame's lextree_enter calls (ku_emit_word) ---- */
    if (C == 1 || r > 0) {
        const int32_t ew = C == 1 ? wave : gwave - KF_WAVES, enw = C == 1 ? KF_WAVES : gwaves - KF_WAVES;
        for (int32_t t = 0; t < T; t++)
            d_dec_emit_w(f, S.node_base, L.act[cur], L.nact[cur], S.child_off, S.child, L.turn, L.selfemit, L.base, L.act[cur ^ 1], L.nact[cur ^ 1],
                         L.pos, L.posf, t, ew, enw);
    }
    if (r == 0) {
        __syncthreads();
        if (C == 1) KF_STAMP(15);
        const long long t_in = (long long)wall_clock64();
        const int32_t nx = d_dec_pack_frame_lds(S.N, T, bm, S.node_base, L.nact[cur], L.best, L.exits, L.nexit, L.hbin, L.misc, L.pack,
                                                S.pack_max_exits, L.gpart, S.gp_n, L.nact[cur ^ 1], sh.pool.wl.hdr, sh.pool.wl.ex, WL_LDS_EX);
        if (big_wl) d_wl_big_begin(L.w, ctx, L.pack, dict, par);       /* wide beams: the candidate phases follow, chunked over the cluster */
        else d_wordlevel_frame(L.w, ctx, L.pack, sh.pool.wl.hdr, nx <= WL_LDS_EX ? sh.pool.wl.ex : (const int32_t *)NULL, lm, dict, par, f, t_in);
    }
    if (big_wl) {
        /* the wide-beam word level (tens of thousands of candidates per frame: configs[4]): the launch path's chip-wide phases
         * ku_wl_p2 .. ku_wl_finish with the cluster's workgroups as the chunks' owners and the cluster barrier for the launch boundaries */
        kf_barrier(B);
        d_wl_big_p2(L.w, ctx, L.pack, lm, dict, par, f, r, C);
        kf_barrier(B);
        d_wl_big_p3(L.w, ctx, L.pack, dict, par, f, r, C);
        kf_barrier(B);
        d_wl_big_p4a(L.w, ctx, L.pack, par, f, r, C);
        kf_barrier(B);
        d_wl_big_p4b(L.w, ctx, L.pack, par, f, r, C);
        kf_barrier(B);
        d_wl_big_p5(L.w, ctx, L.pack, dict, par, f, r, C);
        kf_barrier(B);
        if (r == 0) d_wl_big_finish(L.w, ctx, L.pack, lm, dict, par, f);
    }
    kf_barrier(B);
    KF_STAMP(11);
#undef KF_STAMP
}

struct KfArgs { UShared S; WLm lm; WDict dict; WPar par; KfJob J; };
#define KF_ARG_SLOTS 8

/* the workgroup's LDS, one object per instantiation of the kernel (at namespace scope so that the frame's function addresses it as LDS) */
template <int NE, bool EXACT> __shared__ KfSh g_kfsh;

#ifndef KF_CALL
#define KF_CALL 0               /* (measured, profiles/r6_experiments.txt 4: the frame as a function has no scratch access in its hot loops and is 0.9 % slower) */
#endif
template <class T> __device__ __forceinline__ T *kf_uniform_ptr(T *p)
{
    const unsigned long long v = (unsigned long long)p;
    return (T *)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)v));
}
/* the frame as a function of its own: operands from LDS (KfSh.fa), the cluster barrier's count back there */
template <int NE, bool EXACT>
__device__ __noinline__ void
kf_frame_call()
{
    KfSh &sh = g_kfsh<NE, EXACT>;
    const KfArgs *A = (const KfArgs *)kf_uniform_ptr(sh.fa.A);
    UCtx *ctx = kf_uniform_ptr(sh.fa.ctx);
    const int32_t z = __builtin_amdgcn_readfirstlane(sh.fa.z), r = __builtin_amdgcn_readfirstlane(sh.fa.r), C = __builtin_amdgcn_readfirstlane(sh.fa.C),
        f = __builtin_amdgcn_readfirstlane(sh.fa.f), weak = __builtin_amdgcn_readfirstlane(sh.fa.weak);
    KfBar B = { kf_uniform_ptr(sh.fa.bar_cnt), C, __builtin_amdgcn_readfirstlane(sh.fa.bar_target), &sh.dead, __builtin_amdgcn_readfirstlane(sh.fa.bar_local) };
    kf_frame<NE, EXACT>(sh.Lc, A->S, ctx, A->lm, A->dict, A->par, sh, B, z, r, C, f, kf_uniform_ptr(sh.fa.row), kf_uniform_ptr(sh.fa.brow), weak);
    if (threadIdx.x == 0) sh.fa.bar_target = B.target;
    __syncthreads();
}

/* (KF_OCC: waves per SIMD the register budget is set for -- 4 = 128 VGPRs, two lanes per CU; -DKF_OCC=2 = 256 VGPRs, one lane per CU: the
 * build profiles/r6_experiments.txt uses to tell what the spills cost) */
#ifndef KF_OCC
#define KF_OCC 4
#endif

template <int NE, bool EXACT, bool HEUR = false>
__global__ void __launch_bounds__(KF_NT, KF_OCC)
ku_frames(const ULane *__restrict__ lanes, const KfArgs *__restrict__ A, int32_t n_lanes, int32_t C, int32_t *bar,
          int32_t weak_possible, int32_t local_ok)
{
    /* (what every lane shares arrives through memory, not as ~300 words of kernel arguments the compiler then tries to keep in
     * registers for the whole frame loop: SGPR spills 1 493 -> 647, VGPR spills 342 -> 248, scratch 524 -> 308 B per lane) */
    const UShared &S = A->S; const WLm &lm = A->lm; const WDict &dict = A->dict; const WPar &par = A->par; const KfJob &J = A->J;
    KfSh &sh = g_kfsh<NE, EXACT>;
    /* the lane and this workgroup's place in its cluster: a cluster's workgroups share an XCD */
    int32_t z, r;
    if (C == 1) { z = blockIdx.x; r = 0; }
    else { const int32_t b = blockIdx.x, xcd = b & 7, j = b >> 3; z = (j / C) * 8 + xcd; r = j % C; }
    if (z >= n_lanes) return;
    if (J.resume) {             /* (the relay: slot z of this launch continues the lane the previous launch listed there) */
        if (z >= __hip_atomic_load(&J.st_cur[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        z = J.st_cur[KF_ST_WORDS + z];
    }
    const int32_t tid = threadIdx.x;
    static_assert(sizeof(ULane) % 4 == 0, "the lane's structure is copied to LDS word by word");
    for (int32_t i = tid; i < (int32_t)(sizeof(ULane) / 4); i += KF_NT) ((int32_t *)&sh.Lc)[i] = ((const int32_t *)&lanes[z])[i];
    const ULane &L = sh.Lc;
    UCtx *ctx = S.ctx_all + z;
    if (tid == 0) sh.dead = 0;
    if (tid < 16) sh.kacc[tid] = 0;
    for (int32_t i = tid; i < KF_SENBITS / 32; i += KF_NT) sh.senbits[i] = 0u;
    if (tid <= S.T) sh.nb[tid] = S.node_base[tid];
    if (S.n_tmat * NS_TPW(NE) <= KF_TP_LDS)
        for (int32_t i = tid; i < S.n_tmat * NS_TPW(NE); i += KF_NT) sh.tp[i] = S.tp[i];
    KfBar B = { bar + 2 * z, C, 0, &sh.dead, 0 };
    __syncthreads();
    if (C > 1) {
        /* where the cluster's workgroups run: each leaves its XCD's bit, and behind one general barrier all of them read the same word --
         * one bit: the cheap barrier serves (the observed placement, block b on XCD b % 8, is what the grid is laid out for) */
        if (tid == 0) atomicOr(bar + 2 * z + 1, 1 << (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15));
        kf_barrier(B);
        if (tid == 0) sh.u = __hip_atomic_load(bar + 2 * z + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        B.local = local_ok && __popc(sh.u) == 1;
        __syncthreads();
    }
    const long long t_launch = (long long)wall_clock64();
    /* (ONE call site of the frame for the three modes: the frame's code is ~150 KB, and a copy per mode was three times that in a
     * kernel whose instruction cache holds 64 KB) */
    const int32_t mode = J.mode;
    const int32_t gtid = r * KF_NT + tid, gstride = C * KF_NT;
    bool resumed = J.resume != 0, handed = false;
    if (resumed && C > 1) {     /* (the lane's mask of a frame: clean between frames -- unless an utterance once stopped inside one) */
        if (r == 0) for (int32_t i = tid; i < KF_SENBITS / 32; i += KF_NT) L.senbits[i] = 0u;
        kf_barrier(B);
    }
    for (;;) {
        int32_t u = 0, f_lo = 0, f_hi = 0;
        size_t r0 = 0;
        if (resumed) {          /* the utterance the lane was handed over with, from the frame it had reached */
            u = mode == KF_QUEUE ? J.lane_uq[z] : z;
            f_lo = J.lane_f[z]; f_hi = ctx->nfr;
            r0 = (size_t)J.row0[u];
            resumed = false;
        }
        else if (mode == KF_QUEUE) {
            /* the queue's next utterance: taken by the lane's first workgroup */
            if (r == 0 && tid == 0) {
                const int32_t u_ = atomicAdd(J.next, 1);
                sh.u = u_;
                if (C > 1) __hip_atomic_store(&J.lane_u[z], u_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            kf_barrier(B);
            if (r > 0 && tid == 0) sh.u = __hip_atomic_load(&J.lane_u[z], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            u = sh.u;
            if (u >= J.n_utt || sh.dead) break;
            u = J.order[J.u0 + u];              /* (the take's utterance: its index in the queue) */
            if (r == 0 && tid == 0 && J.lane_uq) J.lane_uq[z] = u;
            /* srch_utt_begin (srch.c:453-479): every per-utterance state reset, the utterance's context */
            d_lane_begin(L, S, J.B, z, J.stage + u, gtid, gstride, r == 0, tid, KF_NT);
            kf_barrier(B);
            f_hi = ctx->nfr;
            r0 = (size_t)J.row0[u];
        }
        else if (mode == KF_STATIC) { f_hi = ctx->nfr; r0 = (size_t)J.row0[z]; }
        else { const int32_t f0 = ctx->f0; f_lo = max(0, J.fg0 - f0); f_hi = min(ctx->nfr, J.fg0 + J.n_fr - f0); }
        for (int32_t f = f_lo; f < f_hi; f++) {
            /* (the word level ends an utterance that ran into an error: uniform over the cluster behind the frame's last barrier) */
            if (!((volatile UCtx *)ctx)->active || sh.dead) break;
            if (J.stop_at > 0) {
                /* the relay's stop flag, as ONE thread of the lane read it (a cluster's workgroups must leave at the same frame) */
                if (r == 0 && tid == 0) {
                    const int32_t st = __hip_atomic_load(&J.st_cur[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sh.u2 = st;
                    if (C > 1) __hip_atomic_store(&J.lane_stop[z], st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                kf_barrier(B);
                if (C > 1 && r > 0 && tid == 0) sh.u2 = __hip_atomic_load(&J.lane_stop[z], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (C > 1) __syncthreads();
                if (sh.u2) {
                    if (r == 0 && tid == 0) {
                        J.lane_f[z] = f;
                        J.st_next[KF_ST_WORDS + atomicAdd(&J.st_next[2], 1)] = z;
                        atomicAdd(&J.st_next[0], 1);
                    }
                    handed = true;
                    break;
                }
            }
            int32_t *row = mode == KF_WINDOW ? L.win + (size_t)(f % S.win_K) * S.n_sen : J.scores + (r0 + f) * S.n_sen;
            const uint8_t *brow = mode == KF_WINDOW ? L.winb + (size_t)(f % S.win_K) * S.n_sen : J.bests + (r0 + f) * S.n_sen;
#if KF_CALL
            if (tid == 0) {
                sh.fa.A = A; sh.fa.ctx = ctx; sh.fa.row = row; sh.fa.brow = brow; sh.fa.bar_cnt = B.cnt; sh.fa.z = z; sh.fa.r = r; sh.fa.C = C; sh.fa.f = f;
                sh.fa.weak = weak_possible; sh.fa.bar_target = B.target; sh.fa.bar_local = B.local;
            }
            __syncthreads();
            kf_frame_call<NE, EXACT>();
            B.target = sh.fa.bar_target;
#else
            kf_frame<NE, EXACT, HEUR>(L, S, ctx, lm, dict, par, sh, B, z, r, C, f, row, brow, weak_possible);
#endif
        }
        if (handed) break;
        if (mode != KF_QUEUE) break;
        if (r == 0 && tid < 16 && tid != 12 && tid != 14) { ctx->kacc[tid] += sh.kacc[tid]; sh.kacc[tid] = 0; }
        /* srch_utt_end (srch.c:482-560): the hypothesis goes to the utterance's slot, the lane's lists are cleared */
        static_assert(3 * WL_LDS_EX >= UH_IDS, "d_hyp's backtrace ids borrow the word level's exit area");
        if (r == 0) d_hyp<KF_NT>(L, ctx, lm, dict, J.P, J.hdr + (size_t)u * UH_N, J.words, J.wcount, sh.pool.wl.ex);
        kf_barrier(B);
        d_lane_end(L, S, z, ctx->err != 0, J.n_word, gtid, gstride);
        kf_barrier(B);
    }
    /* (the relay: a lane that is through counts itself out; the one that leaves `stop_at` lanes at work raises the flag for them) */
    if (J.stop_at > 0 && !handed && r == 0 && tid == 0) {
        const int32_t left = atomicSub(&J.st_cur[0], 1) - 1;
        if (left <= J.stop_at && left > 0) __hip_atomic_store(&J.st_cur[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (r == 0 && tid == 0) { sh.kacc[12] += (long long)wall_clock64() - t_launch; sh.kacc[14]++; }
    if (r == 0 && tid < 16) ctx->kacc[tid] += sh.kacc[tid];             /* (KF_QUEUE: the lane's last utterance has its frames' share already) */
    if (r == 0 && tid == 0 && J.mode == KF_WINDOW && J.fg0 == 512) {
        ctx->kdbg[0] = t_launch; ctx->kdbg[1] = (long long)wall_clock64();
        ctx->kdbg[2] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4); ctx->kdbg[3] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
    if (sh.dead && tid == 0) { ctx->err |= WL_E_SCAN; ctx->active = 0; }
}

#define LANE_W const ULane &L = lanes[blockIdx.z]; UCtx *ctx = L.ctx; const int32_t f = fg - ctx->f0;               \
    if (f < 0 || f >= ctx->nfr || !ctx->active) return
/* the wide-beam word level: WL_BIG_G workgroups per lane and phase (s3a_wordlevel.h) */
__global__ void __launch_bounds__(WL_THREADS)
ku_wl_p2(const ULane *__restrict__ lanes, WLm lm, WDict dict, WPar par, int32_t fg)
{
    LANE_W;
    d_wl_big_p2(L.w, ctx, L.pack, lm, dict, par, f, blockIdx.x, gridDim.x);
}
__global__ void __launch_bounds__(WL_THREADS)
ku_wl_p3(const ULane *__restrict__ lanes, WDict dict, WPar par, int32_t fg)
{
    LANE_W;
    d_wl_big_p3(L.w, ctx, L.pack, dict, par, f, blockIdx.x, gridDim.x);
}
__global__ void __launch_bounds__(WL_THREADS)
ku_wl_p4a(const ULane *__restrict__ lanes, WPar par, int32_t fg)
{
    LANE_W;
    d_wl_big_p4a(L.w, ctx, L.pack, par, f, blockIdx.x, gridDim.x);
}
__global__ void __launch_bounds__(WL_THREADS)
ku_wl_p4b(const ULane *__restrict__ lanes, WPar par, int32_t fg)
{
    LANE_W;
    d_wl_big_p4b(L.w, ctx, L.pack, par, f, blockIdx.x, gridDim.x);
}
__global__ void __launch_bounds__(WL_THREADS)
ku_wl_p5(const ULane *__restrict__ lanes, WDict dict, WPar par, int32_t fg)
{
    LANE_W;
    d_wl_big_p5(L.w, ctx, L.pack, dict, par, f, blockIdx.x, gridDim.x);
}
__global__ void __launch_bounds__(WL_THREADS)
ku_wl_finish(const ULane *__restrict__ lanes, WLm lm, WDict dict, WPar par, int32_t fg)
{
    LANE_W;
    d_wl_big_finish(L.w, ctx, L.pack, lm, dict, par, f);
}

/* stand-alone word-level frame on a caller-filled record (tests: lock step with the oracle) */
__global__ void __launch_bounds__(WL_THREADS)
ku_wordlevel_only(const ULane *__restrict__ lanes, WLm lm, WDict dict, WPar par)
{
    __shared__ int32_t s_hdr[6 * WL_MAXT + 16], s_ex[3 * WL_LDS_EX];
    const ULane &L = lanes[blockIdx.z];
    if (!L.ctx->active) return;
    const int32_t hdr = 6 * par.T + 16;
    int32_t nx = 0;
    for (int32_t t = 0; t < par.T; t++) nx += L.pack[3 * par.T + 8 + t];
    for (int32_t i = threadIdx.x; i < hdr; i += WL_THREADS) s_hdr[i] = L.pack[i];
    if (nx <= WL_LDS_EX) for (int32_t i = threadIdx.x; i < 3 * nx; i += WL_THREADS) s_ex[i] = L.pack[hdr + i];
    __syncthreads();
    d_wordlevel_frame(L.w, L.ctx, L.pack, s_hdr, nx <= WL_LDS_EX ? s_ex : (const int32_t *)NULL, lm, dict, par, L.ctx->cf,
                      (long long)wall_clock64());
}

/* the last node of a captured block of frames: the engine's frame counter moves on */
__global__ void
ku_advance(int32_t *fgbase, int32_t by)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *fgbase += by;
}

__global__ void
ku_pack_node4(const int32_t *__restrict__ ssid, const int32_t *__restrict__ tmatid, const int32_t *__restrict__ wid,
              const uint8_t *__restrict__ comp, int4 *out, int32_t N)
{
    const int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < N) out[v] = make_int4(ssid[v], tmatid[v], wid[v], (int32_t)comp[v]);
}

/* 3-state HMMs: everything ku_frames reads of a node in one 16-byte word (UShared.nodepk) */
__global__ void
ku_pack_nodepk(const int32_t *__restrict__ ssid, const int32_t *__restrict__ tmatid, const int32_t *__restrict__ wid,
               const uint8_t *__restrict__ comp, const int16_t *__restrict__ sseq, const int16_t *__restrict__ comsseq,
               const int32_t *__restrict__ ps, const int32_t *__restrict__ par_off, int4 *out, int32_t N)
{
    const int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    const int16_t *row = (comp[v] ? comsseq : sseq) + (size_t)ssid[v] * 3;
    const uint32_t i0 = (uint16_t)row[0], i1 = (uint16_t)row[1], i2 = (uint16_t)row[2];
    /* (bit 30: the node is a member of a several-parent set the set passes resolve -- SET_NP_MIN .. 64 parents) */
    const int32_t np = par_off[v + 1] - par_off[v];
    out[v] = make_int4((int32_t)(i0 | (i1 << 16)), (int32_t)(i2 | ((uint32_t)tmatid[v] << 16)), wid[v],
                       (int32_t)(((np >= SET_NP_MIN && np <= 64) ? 0x40000000u : 0u) | ((uint32_t)(ps[v] + 1) << 1) | (comp[v] ? 1u : 0u)));
}

/* (ne = 3: 2 words per node, ne = 5: 4) */
__global__ void
ku_pack_nodesen(const int32_t *__restrict__ ssid, const uint8_t *__restrict__ comp, const int16_t *__restrict__ sseq,
                const int16_t *__restrict__ comsseq, int32_t *out, int32_t N, int32_t ne)
{
    const int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    const int16_t *row = (comp[v] ? comsseq : sseq) + (size_t)ssid[v] * ne;
    const int32_t sv = ne == 3 ? 2 : 4;
    for (int32_t k = 0; k < sv; k++) {
        uint32_t lo = 2 * k < ne ? (uint32_t)(uint16_t)row[2 * k] : 0u, hi = 2 * k + 1 < ne ? (uint32_t)(uint16_t)row[2 * k + 1] : 0u;
        if (2 * k + 1 == ne) hi = comp[v] ? 1u : 0u;            /* (the half behind the last id: composite?) */
        out[(size_t)v * sv + k] = (int32_t)(lo | (hi << 16));
    }
}

__global__ void
ku_gather32(const int32_t *__restrict__ src, const int32_t *__restrict__ idx, int32_t *dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

__global__ void
ku_fill32(int32_t *p, int32_t v, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

/* ------------------------------------------------------------------ */
/* host side                                                           */
/* ------------------------------------------------------------------ */

#define DM(ptr, bytes) do { if (hipMalloc((void **)&(ptr), (bytes) > 0 ? (bytes) : 4) != hipSuccess) { \
        s3a_set_error("s3a_utt: device allocation of %zu bytes failed", (size_t)(bytes)); goto fail; } } while (0)
#define UPV(dst, vec) do { DM(dst, (vec).size() * 4); \
        if ((vec).size() && hipMemcpy((void *)(dst), (vec).data(), (vec).size() * 4, hipMemcpyHostToDevice) != hipSuccess) { \
            s3a_set_error("s3a_utt: upload failed"); goto fail; } } while (0)

/* the host copy, checked (what the second pass's host side -- s3a_lattice_nbest -- reads; no device needed) */
static s3a_lm3g_t *
lm3g_host(int32_t n_ug, const int32_t *ug_prob, const int32_t *ug_bowt, const int32_t *ug_firstbg, int32_t n_bg,
              const int32_t *bg_wid, const int32_t *bg_prob, const int32_t *bg_bowt, const int32_t *bg_firsttg,
              int32_t n_tg, const int32_t *tg_wid, const int32_t *tg_prob, const int32_t *inclass_ugscore,
              int32_t n_dictword)
{
    if (n_ug <= 0 || !ug_prob || !ug_bowt || n_bg < 0 || n_tg < 0 || (n_bg > 0 && (!ug_firstbg || !bg_wid || !bg_prob))
        || (n_tg > 0 && (n_bg == 0 || !bg_bowt || !bg_firsttg || !tg_wid || !tg_prob))) {
        s3a_set_error("s3a_lm3g_init: bad arguments");
        return NULL;
    }
    s3a_lm3g_t *lm = new s3a_lm3g_s();
    lm->ug_prob.assign(ug_prob, ug_prob + n_ug); lm->ug_bowt.assign(ug_bowt, ug_bowt + n_ug);
    if (n_bg > 0) lm->ug_firstbg.assign(ug_firstbg, ug_firstbg + n_ug + 1); else lm->ug_firstbg.assign(n_ug + 1, 0);
    if (n_bg > 0) { lm->bg_wid.assign(bg_wid, bg_wid + n_bg); lm->bg_prob.assign(bg_prob, bg_prob + n_bg); }
    if (n_tg > 0) {
        lm->bg_bowt.assign(bg_bowt, bg_bowt + n_bg); lm->bg_firsttg.assign(bg_firsttg, bg_firsttg + n_bg + 1);
        lm->tg_wid.assign(tg_wid, tg_wid + n_tg); lm->tg_prob.assign(tg_prob, tg_prob + n_tg);
    }
    if (inclass_ugscore) lm->inclass.assign(inclass_ugscore, inclass_ugscore + n_dictword);
    lm->n_dictword = n_dictword;
    /* the look-ups bisect: every run must be sorted and duplicate-free, as lm_3g_dmp.c writes them
     * (the reference's find_bg / find_tg, lm.c:1132-1178, assume the same) */
    for (int32_t w = 0; w < n_ug && n_bg > 0; w++) {
        const int32_t b0 = lm->ug_firstbg[w], b1 = lm->ug_firstbg[w + 1];
        if (b0 < 0 || b1 < b0 || b1 > n_bg) { s3a_set_error("s3a_lm3g_init: bigram offsets of unigram %d out of range", w); delete lm; return NULL; }
        for (int32_t b = b0 + 1; b < b1; b++)
            if (lm->bg_wid[b] <= lm->bg_wid[b - 1]) { s3a_set_error("s3a_lm3g_init: bigrams of unigram %d are not sorted", w); delete lm; return NULL; }
    }
    for (int32_t b = 0; b < n_bg && n_tg > 0; b++) {
        const int32_t t0 = lm->bg_firsttg[b], t1 = lm->bg_firsttg[b + 1];
        if (t0 < 0 || t1 < t0 || t1 > n_tg) { s3a_set_error("s3a_lm3g_init: trigram offsets of bigram %d out of range", b); delete lm; return NULL; }
        for (int32_t t = t0 + 1; t < t1; t++)
            if (lm->tg_wid[t] <= lm->tg_wid[t - 1]) { s3a_set_error("s3a_lm3g_init: trigrams of bigram %d are not sorted", b); delete lm; return NULL; }
    }
    memset(&lm->d, 0, sizeof lm->d);
    lm->d.n_ug = n_ug; lm->d.n_bg = n_bg; lm->d.n_tg = n_tg;
    return lm;
}

extern "C" s3a_lm3g_t *
s3a_lm3g_init_host(int32_t n_ug, const int32_t *ug_prob, const int32_t *ug_bowt, const int32_t *ug_firstbg, int32_t n_bg,
                   const int32_t *bg_wid, const int32_t *bg_prob, const int32_t *bg_bowt, const int32_t *bg_firsttg,
                   int32_t n_tg, const int32_t *tg_wid, const int32_t *tg_prob, const int32_t *inclass_ugscore,
                   int32_t n_dictword)
{
    return lm3g_host(n_ug, ug_prob, ug_bowt, ug_firstbg, n_bg, bg_wid, bg_prob, bg_bowt, bg_firsttg, n_tg, tg_wid, tg_prob, inclass_ugscore, n_dictword);
}

extern "C" s3a_lm3g_t *
s3a_lm3g_init(int32_t n_ug, const int32_t *ug_prob, const int32_t *ug_bowt, const int32_t *ug_firstbg, int32_t n_bg,
              const int32_t *bg_wid, const int32_t *bg_prob, const int32_t *bg_bowt, const int32_t *bg_firsttg,
              int32_t n_tg, const int32_t *tg_wid, const int32_t *tg_prob, const int32_t *inclass_ugscore,
              int32_t n_dictword)
{
    s3a_lm3g_t *lm = lm3g_host(n_ug, ug_prob, ug_bowt, ug_firstbg, n_bg, bg_wid, bg_prob, bg_bowt, bg_firsttg, n_tg, tg_wid, tg_prob, inclass_ugscore, n_dictword);
    if (!lm) return NULL;
    UPV(lm->d.ug_prob, lm->ug_prob); UPV(lm->d.ug_bowt, lm->ug_bowt); UPV(lm->d.ug_firstbg, lm->ug_firstbg);
    UPV(lm->d.bg_wid, lm->bg_wid); UPV(lm->d.bg_prob, lm->bg_prob); UPV(lm->d.bg_bowt, lm->bg_bowt);
    UPV(lm->d.bg_firsttg, lm->bg_firsttg); UPV(lm->d.tg_wid, lm->tg_wid); UPV(lm->d.tg_prob, lm->tg_prob);
    if (inclass_ugscore) UPV(lm->d.inclass, lm->inclass);
    return lm;
fail:
    s3a_lm3g_free(lm);
    return NULL;
}

extern "C" void
s3a_lm3g_free(s3a_lm3g_t *lm)
{
    if (!lm) return;
    const int32_t *p[] = { lm->d.ug_prob, lm->d.ug_bowt, lm->d.ug_firstbg, lm->d.bg_wid, lm->d.bg_prob, lm->d.bg_bowt,
                           lm->d.bg_firsttg, lm->d.tg_wid, lm->d.tg_prob, lm->d.inclass };
    for (auto q : p) if (q) (void)hipFree((void *)q);
    delete lm;
}

/* lm_tg_score on the host copy (lm.c:1661-1833): the utterance's final </s> transition, tests */
static int32_t
h_find(const int32_t *v, int32_t n, int32_t w)
{
    int32_t lo = 0, hi = n;
    while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (v[mid] < w) lo = mid + 1; else hi = mid; }
    return (lo < n && v[lo] == w) ? lo : -1;
}
static int32_t
h_add(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static int32_t
h_bg(const s3a_lm3g_t *lm, int32_t lw1, int32_t lw2, int32_t wid)
{
    int32_t s;
    if (lm->d.n_bg == 0 || lw1 < 0) s = lm->ug_prob[lw2];
    else {
        const int32_t b0 = lm->ug_firstbg[lw1], n = lm->ug_firstbg[lw1 + 1] - b0;
        const int32_t i = n > 0 ? h_find(lm->bg_wid.data() + b0, n, lw2) : -1;
        s = i >= 0 ? lm->bg_prob[b0 + i] : h_add(lm->ug_bowt[lw1], lm->ug_prob[lw2]);
    }
    if (!lm->inclass.empty()) s = h_add(s, lm->inclass[wid]);
    return s;
}
extern "C" int32_t
s3a_lm3g_tg_score(const s3a_lm3g_t *lm, int32_t lw1, int32_t lw2, int32_t lw3, int32_t wid)
{
    if (lm->d.n_tg == 0 || lw1 < 0) return h_bg(lm, lw2, lw3, wid);
    const int32_t b0 = lm->ug_firstbg[lw1], nb = lm->ug_firstbg[lw1 + 1] - b0;
    int32_t b = nb > 0 ? h_find(lm->bg_wid.data() + b0, nb, lw2) : -1, bowt = 0;
    if (b >= 0) {
        b += b0;
        bowt = lm->bg_bowt[b];
        const int32_t t0 = lm->bg_firsttg[b], nt = lm->bg_firsttg[b + 1] - t0;
        const int32_t i = nt > 0 ? h_find(lm->tg_wid.data() + t0, nt, lw3) : -1;
        if (i >= 0) return lm->inclass.empty() ? lm->tg_prob[t0 + i] : h_add(lm->tg_prob[t0 + i], lm->inclass[wid]);
    }
    return h_add(bowt, h_bg(lm, lw2, lw3, wid));
}

/*
 * A queue with the second pass (s3a_uttdec_enable_bestpath + s3a_uttdec_decode_queue): the pass has just run for the lanes that
 * ended at this refill event; what it left in the lane's arena -- status words, the best path end first -- goes to the
 * UTTERANCE's slot before the lane's next utterance overwrites it: 16 status words per utterance, the words packed behind one
 * counter like ku_hyp's, in utterance order, each with its sum of frame normalisers (compute_scale).
 */
#define UD_WOFF 15          /* (a free io word: where the utterance's words start in the packed buffer; -1: no room) */
__global__ void __launch_bounds__(UH_T)
ku_dag_store(const ULane *__restrict__ lanes, const DagLane *__restrict__ dl, int32_t hyp_cap, int32_t *__restrict__ io_all,
             int32_t *__restrict__ words_all, int32_t *__restrict__ wcount, int32_t wtotal, const int32_t *__restrict__ sub,
             const int32_t *__restrict__ slot)
{
    const int32_t z = sub[blockIdx.x], u = slot[blockIdx.x], tid = threadIdx.x;
    const ULane &L = lanes[z];
    const DagLane &D = dl[z];
    const int32_t nfr = L.ctx->nfr;
    int32_t *io = io_all + (size_t)u * DG_IO_N;
    __shared__ int32_t s_woff;
    const int32_t status = D.io[DG_IO_STATUS], n = status == 0 ? D.io[DG_IO_NWORDS] : 0;
    if (tid < DG_IO_N && tid != UD_WOFF) io[tid] = D.io[tid];
    if (tid == 0) {
        s_woff = n > 0 ? atomicAdd(wcount, n) : 0;
        if (n > 0 && (long long)s_woff + n > (long long)wtotal) s_woff = -1;
        io[UD_WOFF] = s_woff;
    }
    __syncthreads();
    if (s_woff < 0 || n <= 0) return;
    int32_t *words = words_all + (size_t)s_woff * 6;
    for (int32_t q = tid; q < n; q += UH_T) {
        const int32_t r = n - 1 - q;            /* (dag_backtrace prepends: the pass leaves the words end first) */
        int32_t *o = words + (size_t)q * 6;
        for (int k = 0; k < 5; k++) o[k] = D.out[(size_t)k * hyp_cap + r];
        uint32_t sc = 0u;
        for (int32_t i = max(o[1], 0); i < o[2] && i < nfr; i++) sc += (uint32_t)L.w.fstat[(size_t)i * 8];
        o[5] = (int32_t)sc;
    }
}

struct HostLane {
    s3a_lexsearch_t *ls;
    s3a_scorer_t *sc;
    ULane d;
    UCtx *h_ctx;                /* pinned */
    float *d_feat;
    size_t feat_cap;
    float *h_feat;              /* pinned staging */
    size_t h_feat_cap;
    int32_t *h_tab;             /* pinned: the downloaded history table [13 arrays] */
    size_t h_tab_cap;
    int32_t *h_st, *h_fstat;    /* pinned */
    int32_t nfr, n_entry, epoch, dirty;
};

struct s3a_uttdec_s {
    s3a_lm3g_t *lm;
    s3a_comsen_t *cs;
    s3a_mgau_model_t *g;
    int32_t n_lanes, max_frames, vh_cap, cand_cap, ex_cap, new_cap, exact, veclen;
    UShared S;
    WDict dict;
    WPar par;
    s3a_wordlevel_cfg_t cfg;
    std::vector<int32_t> h_lwid, h_fillpen, h_last_ci, h_tree_type;
    std::vector<uint8_t> h_is_filler;
    std::vector<HostLane> lane;
    ULane *d_lanes;
    int32_t *d_lcmap;
    std::vector<int32_t> h_lcmap;
    int device;                 /* the device the engine lives on: made current at every entry point (HIP's current
                                 * device is per host thread, and engines are driven from several) */
    int32_t many;               /* from this many lanes on: the grids / kernels for many lanes per launch (S3A_UTT_MANY; tests) */
    int32_t g_eval, eval_block, g_ent, g_mark, g_res, scan_nc, scan_gc, hist_possible, weak_possible;
    hipStream_t stream;
    int32_t n_utt;              /* lanes in use by the last decode */
    double last_decode_ms;
    int32_t prof_every;         /* > 0: every prof_every-th frame is bracketed by events */
    struct ProfEv { int32_t cls; hipEvent_t a, b; };
    std::vector<ProfEv> prof_ev;
    double prof_us[24];
    int64_t prof_n[24];
    int32_t big_wl;             /* the word level's candidate phases as their own launches (wide beams) */
    hipEvent_t ev0, ev1;        /* around the frames of a decode (last_decode_ms) */
    int32_t urk;                /* S3A_UTT_URK: nodes per thread of the resolve sweep (experiments) */
    int32_t no_multi, gy;       /* tuning switches, read ONCE at init (S3A_UTT_NO_MULTI, S3A_UTT_GY; tests) */
    int32_t *d_dbg;             /* S3A_UTT_FRAMECHECK: [n_lanes][16] first broken invariant per lane */
    int32_t times;              /* S3A_UTT_TIMES: print the host-side phases of every decode call to stderr */
    int32_t win_fpc;            /* S3A_UTT_WIN_FPC: slots per chunk of the look-ahead scoring (0: the cost model's) */
    UCtx *h_ctx_dn;             /* pinned [n_lanes]: the lanes' contexts after a decode (HostLane.h_ctx points into it) */
    int32_t *d_hyp_hdr, *h_hyp_hdr;     /* [n_lanes][UH_N]: ku_hyp's header per lane (device / pinned) */
    int32_t *d_hyp_words, *h_hyp_words; /* [n_lanes][hyp_wcap][6]: its words */
    int32_t hyp_wcap;
    UCtx *h_ctx_up;             /* pinned [n_lanes]: the lanes' contexts of the coming decode, uploaded with ONE copy */
    int32_t n_pset;
    s3a_dagpass_t *dag;         /* the second pass after every decode (s3a_uttdec_enable_bestpath), or NULL */
    int32_t keep_tables;        /* 0: with the second pass enabled the history tables stay on the device */
    int32_t tables_fetched, fstat_fetched;
    /* graph mode (s3a_uttdec_opts_t.graph): a block of frames captured once per lane count, replayed block after block */
    int32_t use_graph;
    int32_t scan_small_from;    /* lanes per launch from which ku_scan runs with 256-thread workgroups */
    int32_t *d_fgbase;
    struct FrameGraph { int32_t n, frames; hipGraphExec_t exec; UShared S; };
    std::vector<FrameGraph> graphs;
    /* s3a_uttdec_decode_queue (lane refill): everything of a queue lives in buffers that only grow */
    int32_t q_n;                /* utterances of the last decode when it was a queue (0: a plain decode) */
    std::vector<int32_t> q_nfr; /* their frame counts */
    float *q_feat_d, *q_feat_h; size_t q_feat_cap;      /* the queue's features, rows padded to the scorer's stride */
    UCtx *q_ctx_d, *q_ctx_h; size_t q_ctx_cap;          /* the utterances' staged contexts */
    int32_t *q_sched_d, *q_sched_h; size_t q_sched_cap; /* the refill events' lane / utterance lists */
    int32_t *q_hdr_d, *q_hdr_h; size_t q_hdr_cap;       /* [n_utt][UH_N] + the word counter */
    int32_t *q_words_d, *q_words_h; size_t q_words_cap, q_words_hcap;   /* hypothesis words, packed (6 int32 each) */
    int32_t *q_dio_d, *q_dio_h; size_t q_dio_cap;       /* the second pass inside a queue: [n_utt][DG_IO_N] + the word counter */
    int32_t *q_dw_d, *q_dw_h; size_t q_dw_cap, q_dw_hcap;   /* its words, packed */
    int32_t q_dag;              /* the last queue ran the second pass */
    int32_t q_keep_lat;         /* s3a_uttdec_queue_keep_lattices: the queue's lattices are read back group by group (event by event) */
    struct QLat { s3a_lat_info_t info; std::vector<s3a_lat_node_t> nodes; std::vector<s3a_lat_link_t> links; int32_t have; };
    std::vector<QLat> q_lat;    /* [queue] ... and kept per utterance until the next decode */
    /* ku_frames: the frames of a window as one launch */
    int32_t persist;            /* the engine's configuration is served and the option allows it */
    int32_t kf_cluster_opt;     /* s3a_uttdec_opts_t.cluster */
    int32_t kf_slots;           /* workgroups of ku_frames the device holds at once */
    int32_t *d_kfbar;           /* [n_lanes][2] the clusters' barrier counters, the XCDs their workgroups run on (a bit each) */
    int32_t kf_last_c;          /* workgroups per lane of the last launch (diagnostics) */
    int32_t kf_counted;         /* this engine is counted in g_kf_live */
    int32_t *d_kfnext;          /* [16 + n_lanes] the queue's counter | what a lane's first workgroup took */
    void *d_kfargs, *h_kfargs;  /* [KF_ARG_SLOTS] KfArgs: ku_frames' shared arguments (device / pinned) */
    int32_t kf_arg_at;
    /* SCORES FIRST: every frame's senone scores of a call (ku_frames, KF_STATIC / KF_QUEUE) */
    int32_t *sb_scores; uint8_t *sb_bests; size_t sb_rows_cap;
    size_t sb_rows_max;         /* s3a_uttdec_opts_t.score_rows_max: a cap on the buffer's rows (0: half of the free device memory) */
    int32_t *kf_order_d, *kf_order_h; size_t kf_order_cap;      /* KF_QUEUE: the order in which the lanes take the queue's utterances (longest first) */
    int32_t kf_shared;          /* another PROCESS holds this device's cluster lock: this engine's lanes stay one workgroup each */
    int32_t *d_kfrelay;         /* [(KF_STAGES + 1) (KF_ST_WORDS + n_lanes) + 3 n_lanes] the relay's words per launch of the chain | lane_f | lane_uq | lane_stop */
    int32_t kf_n_relay;         /* launches the last call's chain had behind its first (diagnostics) */
    UwGroup *sb_gdesc_d, *sb_gdesc_h; size_t sb_g_cap;
    long long *sb_row0_d, *sb_row0_h; size_t sb_row0_cap;
    /* where the last call's device time went: events around the scoring launches and around ku_frames (s3a_uttdec_last_parts) */
    std::vector<hipEvent_t> kf_evs;
    int32_t kf_ev_n, kf_n_score, kf_n_frames;
    double kf_score_ms, kf_frames_ms;
};

/* engines with ku_frames alive per device: an engine that is alone on its device may give a lane a cluster of workgroups */
static std::atomic<int> g_kf_live[64];

/* ... and alone means: no other PROCESS runs clusters there either.  The clusters of a launch spin on one another, so all of them must be
 * resident at once; two processes that each sized a grid for the whole device would leave each other's clusters half resident (the
 * spin then runs into KF_SPIN_MAX and the utterances end with WL_E_SCAN).  One advisory lock per device -- a file named by the device's
 * PCI bus id under /dev/shm, flock()ed exclusively by the first process that creates a ku_frames engine there and kept while it has one --
 * tells the others: they keep their lanes at one workgroup each (s3a_uttdec_opts_t.cluster still overrides: the caller then answers for
 * co-residency).  No /dev/shm, no lock: taken as alone, as before. */
static std::mutex g_kf_lock_mu;
static int g_kf_lock_fd[64], g_kf_lock_n[64], g_kf_lock_own[64];
static bool
kf_device_lock(int dev)
{
    std::lock_guard<std::mutex> lk(g_kf_lock_mu);
    if (g_kf_lock_n[dev]++ > 0) return g_kf_lock_own[dev] != 0;
    char bus[64] = "", path[160];
    if (hipDeviceGetPCIBusId(bus, sizeof bus, dev) != hipSuccess) snprintf(bus, sizeof bus, "dev%d", dev);
    for (char *c = bus; *c; c++) if (*c == ':' || *c == '/' || *c == '.') *c = '_';
    snprintf(path, sizeof path, "/dev/shm/cmusphinx_amd.kf.%s.lock", bus);
    const int fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC, 0666);
    g_kf_lock_fd[dev] = fd;
    if (fd < 0) { g_kf_lock_own[dev] = 1; return true; }
    (void)fchmod(fd, 0666);
    g_kf_lock_own[dev] = flock(fd, LOCK_EX | LOCK_NB) == 0 ? 1 : 0;
    return g_kf_lock_own[dev] != 0;
}
static void
kf_device_unlock(int dev)
{
    std::lock_guard<std::mutex> lk(g_kf_lock_mu);
    if (--g_kf_lock_n[dev] > 0) return;
    if (g_kf_lock_fd[dev] >= 0) { (void)flock(g_kf_lock_fd[dev], LOCK_UN); (void)close(g_kf_lock_fd[dev]); }
    g_kf_lock_fd[dev] = -1; g_kf_lock_own[dev] = 0; g_kf_lock_n[dev] = 0;
}

static int32_t
fill32(hipStream_t st, int32_t *p, int32_t v, size_t n)
{
    if (n == 0) return S3A_OK;
    hipLaunchKernelGGL(ku_fill32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, v, n);
    HIPCHK(hipGetLastError());
    return S3A_OK;
}

static void
wlane_free(WLane &w)
{
    void *p[] = { w.score, w.pred, w.lw0, w.lw1, w.wid, w.sf, w.ef, w.ascr, w.lscr, w.type, w.lmc, w.frame_start, w.bestscore,
                  w.bestvh, w.st, w.ex_off, w.ex_info, w.cand_i, w.cand_pref, w.cand_e, w.cand_score, w.cand_slot, w.hkey, w.hbest, w.hfirst,
                  w.hlead_rank, w.sg, w.srt, w.wfirst, w.wbest, w.part, w.part2, w.tb, w.heap, w.nrl, w.fstat };
    for (auto q : p) if (q) (void)hipFree(q);
    memset((void *)&w, 0, sizeof w);
}

/* one lane's history table and per-frame scratch */
static int32_t
wlane_alloc(WLane &w, int32_t vh_cap, int32_t max_frames, int32_t ex_cap, int32_t cand_cap, int32_t new_cap, int32_t n_word,
            hipStream_t st)
{
    const size_t vc = (size_t)vh_cap * 4, mf = (size_t)(max_frames + 2) * 4;
    size_t hs = 1024;
    memset((void *)&w, 0, sizeof w);
    DM(w.score, vc); DM(w.pred, vc); DM(w.lw0, vc); DM(w.lw1, vc); DM(w.wid, vc); DM(w.sf, vc); DM(w.ef, vc);
    DM(w.ascr, vc); DM(w.lscr, vc); DM(w.type, vc); DM(w.lmc, 5 * vc);
    w.cap = vh_cap;
    DM(w.frame_start, mf); DM(w.bestscore, mf); DM(w.bestvh, mf); DM(w.st, 16 * 4);
    DM(w.ex_off, (size_t)(ex_cap + 1) * 4); DM(w.ex_info, (size_t)3 * ex_cap * 4);
    w.ex_cap = ex_cap;
    DM(w.cand_score, (size_t)cand_cap * 4); DM(w.cand_slot, (size_t)cand_cap * 4);
    DM(w.cand_pref, (size_t)cand_cap * 4); DM(w.cand_e, (size_t)cand_cap * 4); DM(w.cand_i, (size_t)cand_cap * 4);
    w.cand_cap = cand_cap;
    while (hs < (size_t)2 * cand_cap) hs <<= 1;
    w.hmask = (int32_t)(hs - 1);
    DM(w.hkey, hs * 8); DM(w.hbest, hs * 8); DM(w.hfirst, hs * 4); DM(w.hlead_rank, hs * 4);
    if (hipMemset(w.hkey, 0, hs * 8) != hipSuccess || hipMemset(w.hbest, 0, hs * 8) != hipSuccess
        || hipMemset(w.hfirst, 0xff, hs * 4) != hipSuccess) { s3a_set_error("s3a_utt: memset failed"); goto fail; }
    w.new_cap = new_cap;
    DM(w.sg, (size_t)11 * new_cap * 4); DM(w.srt, (size_t)6 * new_cap * 4); DM(w.heap, (size_t)18 * new_cap * 4); DM(w.nrl, (size_t)new_cap * 4);
    DM(w.wfirst, (size_t)n_word * 4); DM(w.wbest, (size_t)n_word * 4);
    DM(w.part, WL_BIG_G * 4); DM(w.part2, WL_BIG_G * 4); DM(w.tb, (WL_MAXT + 1) * 4);
    DM(w.fstat, (size_t)max_frames * 8 * 4);
    if (fill32(st, w.wfirst, INT_MAX, n_word) != S3A_OK || fill32(st, w.wbest, INT_MIN, n_word) != S3A_OK) goto fail;
    return S3A_OK;
fail:
    wlane_free(w);
    return S3A_ENOMEM;
}

extern "C" void
s3a_uttdec_free(s3a_uttdec_t *ud)
{
    if (!ud) return;
    (void)hipSetDevice(ud->device);
    (void)hipStreamSynchronize(ud->stream);
    for (auto &hl : ud->lane) {
        wlane_free(hl.d.w);
        if (hl.d.pack) (void)hipFree(hl.d.pack);
        if (hl.d.cs_need) (void)hipFree(hl.d.cs_need);
        if (hl.d.cs_val) (void)hipFree(hl.d.cs_val);
        if (hl.d.cs_wl) (void)hipFree(hl.d.cs_wl);
        if (hl.d.cs_wn) (void)hipFree(hl.d.cs_wn);
        if (hl.d.posbest) (void)hipFree(hl.d.posbest);
        if (hl.d.posps) (void)hipFree(hl.d.posps);
        if (hl.d.senbits) (void)hipFree(hl.d.senbits);
        if (hl.d.dynbeam) (void)hipFree(hl.d.dynbeam);
        if (hl.d.pstamp8) (void)hipFree(hl.d.pstamp8);
        if (hl.d.plist) (void)hipFree(hl.d.plist);
        if (hl.d.claim) (void)hipFree(hl.d.claim);
        if (hl.d.pcnt) (void)hipFree(hl.d.pcnt);
        if (hl.d.ci_all) (void)hipFree(hl.d.ci_all);
        if (hl.d.heur_all) (void)hipFree(hl.d.heur_all);
        if (hl.d.hth_pos) (void)hipFree(hl.d.hth_pos);
        if (hl.d.ph_scratch) (void)hipFree(hl.d.ph_scratch);
        if (hl.d.win) (void)hipFree(hl.d.win);
        if (hl.d.winb) (void)hipFree(hl.d.winb);
        if (hl.ls && ud->S.nact_all && hl.ls->d_nact[0] >= ud->S.nact_all
            && hl.ls->d_nact[0] < ud->S.nact_all + (size_t)ud->n_lanes * 2 * WL_MAXT)
            hl.ls->d_nact[0] = hl.ls->d_nact[1] = NULL;     /* borrowed from nact_all */
        if (hl.d_feat) (void)hipFree(hl.d_feat);
        if (hl.h_feat) (void)hipHostFree(hl.h_feat);
        if (hl.h_tab) (void)hipHostFree(hl.h_tab);
        if (hl.h_st) (void)hipHostFree(hl.h_st);
        if (hl.h_fstat) (void)hipHostFree(hl.h_fstat);
        if (hl.sc) s3a_scorer_free(hl.sc);
        if (hl.ls) s3a_lexsearch_free(hl.ls);
    }
    if (ud->kf_counted) { g_kf_live[ud->device]--; ud->kf_counted = 0; kf_device_unlock(ud->device); }
    if (ud->kf_order_d) (void)hipFree(ud->kf_order_d);
    if (ud->kf_order_h) (void)hipHostFree(ud->kf_order_h);
    for (auto e : ud->kf_evs) (void)hipEventDestroy(e);
    ud->kf_evs.clear();
    if (ud->d_kfbar) (void)hipFree(ud->d_kfbar);
    if (ud->d_kfrelay) (void)hipFree(ud->d_kfrelay);
    if (ud->d_kfnext) (void)hipFree(ud->d_kfnext);
    if (ud->d_kfargs) (void)hipFree(ud->d_kfargs);
    if (ud->h_kfargs) (void)hipHostFree(ud->h_kfargs);
    if (ud->sb_scores) (void)hipFree(ud->sb_scores);
    if (ud->sb_bests) (void)hipFree(ud->sb_bests);
    if (ud->sb_gdesc_d) (void)hipFree(ud->sb_gdesc_d);
    if (ud->sb_gdesc_h) (void)hipHostFree(ud->sb_gdesc_h);
    if (ud->sb_row0_d) (void)hipFree(ud->sb_row0_d);
    if (ud->sb_row0_h) (void)hipHostFree(ud->sb_row0_h);
    if (ud->q_dio_d) (void)hipFree(ud->q_dio_d);
    if (ud->q_dio_h) (void)hipHostFree(ud->q_dio_h);
    if (ud->q_dw_d) (void)hipFree(ud->q_dw_d);
    if (ud->q_dw_h) (void)hipHostFree(ud->q_dw_h);
    if (ud->dag) s3a_dagpass_free(ud->dag);
    {
        void *qd[] = { ud->q_feat_d, ud->q_ctx_d, ud->q_sched_d, ud->q_hdr_d, ud->q_words_d };
        void *qh[] = { ud->q_feat_h, ud->q_ctx_h, ud->q_sched_h, ud->q_hdr_h, ud->q_words_h };
        for (auto q : qd) if (q) (void)hipFree(q);
        for (auto q : qh) if (q) (void)hipHostFree(q);
    }
    for (auto &fg_ : ud->graphs) if (fg_.exec) (void)hipGraphExecDestroy(fg_.exec);
    if (ud->d_fgbase) (void)hipFree(ud->d_fgbase);
    if (ud->h_ctx_up) (void)hipHostFree(ud->h_ctx_up);
    if (ud->h_ctx_dn) (void)hipHostFree(ud->h_ctx_dn);
    if (ud->h_hyp_hdr) (void)hipHostFree(ud->h_hyp_hdr);
    if (ud->h_hyp_words) (void)hipHostFree(ud->h_hyp_words);
    if (ud->d_hyp_hdr) (void)hipFree(ud->d_hyp_hdr);
    if (ud->d_hyp_words) (void)hipFree(ud->d_hyp_words);
    if (ud->ev0) (void)hipEventDestroy(ud->ev0);
    if (ud->ev1) (void)hipEventDestroy(ud->ev1);
    if (ud->d_lanes) (void)hipFree(ud->d_lanes);
    if (ud->S.node_ci) (void)hipFree((void *)ud->S.node_ci);
    if (ud->S.sen2cimap) (void)hipFree((void *)ud->S.sen2cimap);
    if (ud->S.rootprob) (void)hipFree((void *)ud->S.rootprob);
    if (ud->S.pshdr) (void)hipFree((void *)ud->S.pshdr);
    if (ud->S.node4) (void)hipFree((void *)ud->S.node4);
    if (ud->S.nodesen) (void)hipFree((void *)ud->S.nodesen);
    if (ud->S.nodepk) (void)hipFree((void *)ud->S.nodepk);
    if (ud->S.ctx_all) (void)hipFree(ud->S.ctx_all);
    if (ud->S.nact_all) (void)hipFree(ud->S.nact_all);
    if (ud->d_lcmap) (void)hipFree(ud->d_lcmap);
    const void *q[] = { ud->dict.lwid, ud->dict.fillpen, ud->dict.last_ci, ud->dict.is_filler };
    for (auto p : q) if (p) (void)hipFree((void *)p);
    delete ud;
}

/* the tuning options as the environment names them (S3A_UTT_*): what the drop-in program and the tests pass to
 * s3a_uttdec_init_opts; the library itself never reads the environment for an engine's configuration */
extern "C" void
s3a_uttdec_opts_default(s3a_uttdec_opts_t *o)
{
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->big_wl = -1; o->window = -1;
}

extern "C" void
s3a_uttdec_opts_from_env(s3a_uttdec_opts_t *o)
{
    if (!o) return;
    s3a_uttdec_opts_default(o);
    auto num = [](const char *name, int32_t dflt) { const char *v = getenv(name); return v ? (int32_t)atoi(v) : dflt; };
    o->many = num("S3A_UTT_MANY", 0); o->big_wl = num("S3A_UTT_BIGWL", -1); o->window = num("S3A_UTT_WIN", -1);
    o->window_fpc = num("S3A_UTT_WIN_FPC", 0); o->g_eval = num("S3A_UTT_GEVAL", 0); o->g_res = num("S3A_UTT_GRES", 0);
    o->scan_g = num("S3A_UTT_SCAN_G", 0); o->gy = num("S3A_UTT_GY", 0); o->sweep_k = num("S3A_UTT_URK", 0);
    o->no_multi = getenv("S3A_UTT_NO_MULTI") != NULL; o->framecheck = getenv("S3A_UTT_FRAMECHECK") != NULL;
    o->times = num("S3A_UTT_TIMES", 0); o->graph = num("S3A_UTT_GRAPH", 0); o->window_max = num("S3A_UTT_WIN_MAX", 0); o->scan_small_from = num("S3A_UTT_SCAN_SMALL", 0);
    o->persist = num("S3A_UTT_PERSIST", 0); o->cluster = num("S3A_UTT_CLUSTER", 0); o->score_rows_max = num("S3A_UTT_SCORE_ROWS", 0);
}

extern "C" s3a_uttdec_t *
s3a_uttdec_init(const s3a_lexsearch_t *proto, s3a_mgau_model_t *g, const int16_t *cd2cisen, int32_t n_sen,
                int32_t n_ci_sen, int32_t ds_ratio, int32_t cond_ds, double ci_pbeam, float tighten_factor,
                int32_t max_cd, s3a_comsen_t *cs, s3a_lm3g_t *lm, const s3a_wordlevel_cfg_t *cfg, int32_t n_lanes,
                int32_t max_frames, int32_t vh_cap, int32_t cand_cap)
{
    return s3a_uttdec_init_opts(proto, g, cd2cisen, n_sen, n_ci_sen, ds_ratio, cond_ds, ci_pbeam, tighten_factor, max_cd, cs, lm, cfg,
                                n_lanes, max_frames, vh_cap, cand_cap, NULL);
}

extern "C" s3a_uttdec_t *
s3a_uttdec_init_opts(const s3a_lexsearch_t *proto, s3a_mgau_model_t *g, const int16_t *cd2cisen, int32_t n_sen,
                int32_t n_ci_sen, int32_t ds_ratio, int32_t cond_ds, double ci_pbeam, float tighten_factor,
                int32_t max_cd, s3a_comsen_t *cs, s3a_lm3g_t *lm, const s3a_wordlevel_cfg_t *cfg, int32_t n_lanes,
                int32_t max_frames, int32_t vh_cap, int32_t cand_cap, const s3a_uttdec_opts_t *opts)
{
    s3a_uttdec_opts_t o_;
    if (opts) o_ = *opts; else s3a_uttdec_opts_default(&o_);
    const s3a_uttdec_opts_t &O = o_;
    if (!proto || !g || !g->dev || !cs || !lm || !cfg || n_lanes <= 0 || n_lanes > 1024 || max_frames <= 0) {
        s3a_set_error("s3a_uttdec_init: bad arguments");
        return NULL;
    }
    struct s3a_mgau_dev_s *d = g->dev;
    const int32_t T = proto->n_tree;
    if (T > WL_MAXT || T != 2 * cfg->n_lextree || cfg->n_ci > 255 || cfg->n_ci <= 0 || cfg->epl <= 0 || cfg->n_lextree <= 0) {
        s3a_set_error("s3a_uttdec_init: %d lextrees / %d CI phones outside what the word level supports", T, cfg->n_ci);
        return NULL;
    }
    if (cfg->wbeam_vh > 0 || cfg->wbeam > 0 || cfg->hmmbeam > 0 || cfg->pbeam > 0) {
        s3a_set_error("s3a_uttdec_init: beams must be log probabilities (<= 0)");
        return NULL;
    }
    if (d->tab16 == NULL) { s3a_set_error("s3a_uttdec_init: 32-bit log-add tables are not supported"); return NULL; }
    if (max_cd < n_sen - n_ci_sen && n_ci_sen > UDB_MAXCI) { s3a_set_error("s3a_uttdec_init: -maxcdsenpf with more than %d CI senones", UDB_MAXCI); return NULL; }
    if (lm->n_dictword != 0 && lm->n_dictword != cfg->n_word && !lm->inclass.empty()) {
        s3a_set_error("s3a_uttdec_init: the LM's class table and the dictionary disagree");
        return NULL;
    }
    s3a_uttdec_t *ud = new s3a_uttdec_s();
    ud->lm = lm; ud->cs = cs; ud->g = g; ud->n_lanes = n_lanes; ud->max_frames = max_frames;
    ud->cfg = *cfg;
    ud->d_lanes = NULL; ud->d_lcmap = NULL; ud->n_utt = 0; ud->last_decode_ms = 0.0; ud->prof_every = 0;
    ud->device = 0; ud->ev0 = ud->ev1 = NULL; ud->dag = NULL; ud->keep_tables = 1; ud->tables_fetched = 0; ud->use_graph = 0; ud->d_fgbase = NULL; ud->q_n = 0; ud->q_feat_d = ud->q_feat_h = NULL; ud->q_ctx_d = ud->q_ctx_h = NULL; ud->q_sched_d = ud->q_sched_h = NULL; ud->q_hdr_d = ud->q_hdr_h = NULL; ud->q_words_d = ud->q_words_h = NULL; ud->q_dio_d = ud->q_dio_h = NULL; ud->q_dw_d = ud->q_dw_h = NULL; ud->q_dio_cap = ud->q_dw_cap = ud->q_dw_hcap = 0; ud->q_dag = 0; ud->q_keep_lat = 0; ud->q_feat_cap = ud->q_ctx_cap = ud->q_sched_cap = ud->q_hdr_cap = ud->q_words_cap = ud->q_words_hcap = 0; ud->h_ctx_up = NULL; ud->h_ctx_dn = NULL; ud->d_hyp_hdr = ud->h_hyp_hdr = ud->d_hyp_words = ud->h_hyp_words = NULL; ud->hyp_wcap = 0; ud->n_pset = proto->n_pset;
    (void)hipGetDevice(&ud->device);
    memset(ud->prof_us, 0, sizeof ud->prof_us); memset(ud->prof_n, 0, sizeof ud->prof_n);
    memset(&ud->dict, 0, sizeof ud->dict);
    ud->stream = d->stream;             /* (the model's stream: engines that are to overlap each need a model of their own) */
    ud->exact = g->precision == S3A_GMM_EXACT;
    ud->veclen = g->veclen;
    int32_t maxn = 0, n_leaf_bound = proto->pack_max_exits;
    for (int32_t t = 0; t < T; t++) maxn = max(maxn, proto->node_base[t + 1] - proto->node_base[t]);
    ud->vh_cap = vh_cap > 0 ? vh_cap : (int32_t)min((long long)max_frames * max(cfg->maxhistpf, 1) + 1024, (long long)(4 << 20));
    ud->cand_cap = cand_cap > 0 ? cand_cap : (1 << 20);
    ud->ex_cap = n_leaf_bound;
    ud->new_cap = min(ud->cand_cap, 1 << 18);

    UShared &S = ud->S;
    memset(&S, 0, sizeof S);
    S.N = proto->N; S.T = T; S.n_tmat = proto->n_tmat; S.maxn = maxn; S.n_rootnodes = proto->n_rootnodes;
    S.scan_chunks = proto->scan_chunks; S.pack_max_exits = proto->pack_max_exits;
    S.node_base = proto->d_node_base; S.ssid = proto->d_ssid; S.tmatid = proto->d_tmatid; S.wid = proto->d_wid;
    S.prob = proto->d_prob; S.child_off = proto->d_child_off; S.child = proto->d_child; S.par_off = proto->d_par_off;
    S.par = proto->d_par; S.tree_of = proto->d_tree_of; S.rootlist = proto->d_rootlist; S.tp = proto->d_tp; S.ne = proto->n_emit;
    {   /* the nodes' static words, packed */
        int4 *n4 = NULL;
        if (hipMalloc((void **)&n4, (size_t)(proto->N > 0 ? proto->N : 1) * sizeof(int4)) != hipSuccess) { s3a_set_error("s3a_uttdec_init: out of device memory"); goto fail; }
        hipLaunchKernelGGL(ku_pack_node4, dim3((unsigned)((proto->N + 255) / 256)), dim3(256), 0, ud->stream, proto->d_ssid, proto->d_tmatid,
                           proto->d_wid, proto->d_comp, n4, proto->N);
        S.node4 = n4;
        int32_t *ns = NULL;
        const int32_t ne = proto->n_emit;
        DM(ns, (size_t)proto->N * (ne == 3 ? 2 : 4) * 4);
        S.nodesen = ns;
        hipLaunchKernelGGL(ku_pack_nodesen, dim3((unsigned)((proto->N + 255) / 256)), dim3(256), 0, ud->stream, proto->d_ssid, proto->d_comp, proto->d_sseq,
                           proto->d_comsseq, ns, proto->N, ne);
        if (ne == 3 && proto->n_tmat < 65536 && proto->d_ps && proto->N < (1 << 28)) {
            int4 *pk = NULL;
            DM(pk, (size_t)(proto->N > 0 ? proto->N : 1) * sizeof(int4));
            S.nodepk = pk;
            hipLaunchKernelGGL(ku_pack_nodepk, dim3((unsigned)((proto->N + 255) / 256)), dim3(256), 0, ud->stream, proto->d_ssid, proto->d_tmatid, proto->d_wid,
                               proto->d_comp, proto->d_sseq, proto->d_comsseq, proto->d_ps, proto->d_par_off, pk, proto->N);
        }
    }
    S.pshdr = NULL;
    if (proto->d_ps && proto->n_pset > 0) {
        int4 *ph = NULL;
        if (hipMalloc((void **)&ph, (size_t)proto->n_pset * sizeof(int4)) != hipSuccess) { s3a_set_error("s3a_uttdec_init: out of device memory"); goto fail; }
        hipLaunchKernelGGL(ku_pack_pshdr, dim3((unsigned)((proto->n_pset + 255) / 256)), dim3(256), 0, ud->stream, proto->d_psmem_off, proto->d_psmem, proto->d_par_off,
                           ph, proto->n_pset);
        S.pshdr = ph;
    }
    {   /* the roots' look-ahead probabilities in root-list order (Entries::rootprob) */
        const size_t nr = proto->h_rootlist.size();
        int32_t *rp = NULL;
        if (nr > 0) {
            if (hipMalloc((void **)&rp, nr * 4) != hipSuccess) { s3a_set_error("s3a_uttdec_init: out of device memory"); goto fail; }
            hipLaunchKernelGGL(ku_gather32, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, ud->stream, proto->d_prob, proto->d_rootlist, rp, nr);
        }
        S.rootprob = rp;
    }
    S.rootnodes = proto->d_rootnodes; S.ps = proto->d_ps; S.psof_off = proto->d_psof_off; S.psof = proto->d_psof; S.psmem_off = proto->d_psmem_off; S.psmem = proto->d_psmem;
    S.cs_off = cs->off_d; S.cs_wt = cs->wt_d; S.cs_list = cs->list_d; S.n_cs = cs->n_comstate;
    S.comp = proto->d_comp; S.sseq = proto->d_sseq; S.comsseq = proto->d_comsseq;
    S.mean4 = d->mean4; S.prec4 = d->prec4; S.lrd = d->lrd; S.mixw = d->mixw; S.tab16 = d->tab16; S.tab_size = d->tab_size;
    S.lm_zero = d->lm_zero; S.f = g->f; S.distfloor = g->distfloor; S.D4 = d->D4; S.CP = d->CP; S.Gpad = d->Gpad;
    S.n_sen = n_sen; S.n_ci_sen = n_ci_sen;
    S.ds_ratio = ds_ratio > 0 ? ds_ratio : 1; S.ptranskip = cfg->ptranskip;
    S.bm.hmmbeam = cfg->hmmbeam; S.bm.pbeam = cfg->pbeam; S.bm.wbeam = cfg->wbeam; S.bm.phone_uses_wbeam = 0;
    S.bm.maxhmmpf = cfg->maxhmmpf;

    /* launch geometry: fixed grids, the kernels loop over the list lengths they find in memory */
    ud->eval_block = (cfg->maxhmmpf >= EVBLOCK_LONG_LIST && maxn >= EVBLOCK_LONG_LIST) ? 256 : 64;
    ud->no_multi = O.no_multi != 0;
    ud->d_dbg = NULL;
    if (O.framecheck) { if (hipMalloc((void **)&ud->d_dbg, (size_t)n_lanes * 64) != hipSuccess || hipMemset(ud->d_dbg, 0, (size_t)n_lanes * 64) != hipSuccess) ud->d_dbg = NULL; }
    ud->gy = O.gy;
    ud->many = O.many > 0 ? O.many : 32;
    /* fixed grids, sized for the usual frame: a workgroup loops when a list is longer (virtual workgroups); with many
     * lanes the idle workgroups of a generous grid cost more than the loop */
    /* (many lanes: NOT a multiple of 16 -- measured with four 128-lane engines: 32 / 48 / 64 / 96 workgroups per (tree, lane) 406 /
     * 424 / 411 / 409 k frames/s, 24 ... 72 otherwise 428 ... 438 k, odd counts best: consecutive (tree, lane) groups then start on
     * rotating XCDs instead of piling every group's first, always busy, workgroup onto the same one) */
#ifndef G_EVAL_MANY
#define G_EVAL_MANY 57
#endif
    ud->g_eval = max(1, min((maxn + ud->eval_block - 1) / ud->eval_block, n_lanes >= ud->many ? G_EVAL_MANY : 2048 / max(1, min(n_lanes, 8))));
    if (O.g_eval > 0) ud->g_eval = O.g_eval;
#ifndef G_RES_MANY
#define G_RES_MANY 49         /* (128: 401 k, 96: 403 k, 64: 405 k / 408.5 k, 48: 411 k, 32: 410 k frames/s; + UR_GL an odd total) */
#endif
    ud->g_res = max(1, min((proto->N + RSBLOCK - 1) / RSBLOCK, n_lanes >= ud->many ? G_RES_MANY : 1024));
    if (O.g_res > 0) ud->g_res = O.g_res;
    ud->urk = O.sweep_k > 0 ? O.sweep_k : UR_K;
#ifndef G_ENT
#define G_ENT 256
#endif
    ud->g_ent = max(1, min((proto->ent_cap + 255) / 256, G_ENT));
    ud->g_mark = max(1, min((proto->ent_cap + M3BLOCK - 1) / M3BLOCK + ((maxn + M3BLOCK - 1) / M3BLOCK) * T, 1024));
#ifndef G_MARK_MANY
#define G_MARK_MANY 128
#endif
    if (n_lanes >= ud->many) ud->g_mark = min(ud->g_mark, G_MARK_MANY);     /* (many lanes: 1024 workgroups per lane were 131 k per launch, most of them idle: 50 -> 30 us per 128-lane launch, 331 -> 349 k frames/s) */
    ud->scan_nc = (cfg->maxhmmpf >= SCAN_LONG_LIST && maxn >= SCAN_LONG_LIST) ? (maxn + SCAN_THREADS - 1) / SCAN_THREADS : 1;
    ud->scan_gc = O.scan_g;
    /* lextree_hmm_histbin can only fire when more than 1.5 x maxhmmpf HMMs can be active at all */
    ud->hist_possible = (long long)proto->N > (long long)cfg->maxhmmpf + (cfg->maxhmmpf >> 1);
    ud->weak_possible = cfg->ptranskip != 0 || cfg->pbeam < cfg->hmmbeam;
    /* wide beams (thousands of word exits, 10^5..10^6 (exit, predecessor) candidates per frame): the word level's
     * candidate phases run chip-wide as their own launches; opts->big_wl = 0 / 1 overrides */
    ud->big_wl = O.big_wl >= 0 ? (O.big_wl != 0) : (cfg->maxhmmpf >= 50000 && maxn >= 50000);
    /* (wide beams keep the round-3 SWEEP in the resolve launch, whose workgroups G_RES_MANY was never meant to size: 128 there, as
     * before the list-position grids were retuned -- configs[4]: 363 us per launch against 659) */
    if (ud->big_wl && O.g_res <= 0 && n_lanes >= ud->many) ud->g_res = max(1, min((proto->N + RSBLOCK - 1) / RSBLOCK, 128));
    if (ud->hist_possible && -cfg->hmmbeam / NBIN == 0) {
        s3a_set_error("s3a_uttdec_init: -beam too narrow for histogram pruning (bin width 0)");
        goto fail;
    }

    /* dictionary facts + root lists per (tree, left context) */
    ud->h_lwid.assign(cfg->lwid, cfg->lwid + cfg->n_word); ud->h_fillpen.assign(cfg->fillpen, cfg->fillpen + cfg->n_word);
    ud->h_last_ci.assign(cfg->last_ci, cfg->last_ci + cfg->n_word);
    ud->h_is_filler.assign(cfg->is_filler, cfg->is_filler + cfg->n_word);
    ud->h_tree_type.assign(cfg->tree_type, cfg->tree_type + T);
    ud->cfg.lwid = ud->h_lwid.data(); ud->cfg.fillpen = ud->h_fillpen.data(); ud->cfg.last_ci = ud->h_last_ci.data();
    ud->cfg.is_filler = ud->h_is_filler.data(); ud->cfg.tree_type = ud->h_tree_type.data();
    if (!lm->d.ug_prob) { s3a_set_error("s3a_uttdec_init: the LM handle has no device arrays (s3a_lm3g_init_host)"); goto fail; }
    for (int32_t w = 0; w < cfg->n_word; w++)
        if (cfg->last_ci[w] < 0 || cfg->last_ci[w] >= cfg->n_ci || (cfg->lwid[w] >= lm->d.n_ug)) {
            s3a_set_error("s3a_uttdec_init: dictionary word %d has a bad final phone / LM id", w);
            goto fail;
        }
    ud->h_lcmap.assign((size_t)T * (cfg->n_ci + 1) * 2, -1);
    for (int32_t t = 0; t < T; t++) {
        if (proto->n_lc[t] > 0) {
            for (int32_t k = 0; k < proto->n_lc[t]; k++) {
                const int32_t p = proto->lc[t][k];
                if (p < 0 || p >= cfg->n_ci) continue;
                ud->h_lcmap[((size_t)t * (cfg->n_ci + 1) + p) * 2] = proto->rootbuf_base[t] + proto->lcroot_off[t][k];
                ud->h_lcmap[((size_t)t * (cfg->n_ci + 1) + p) * 2 + 1] = proto->lcroot_off[t][k + 1] - proto->lcroot_off[t][k];
            }
        }
        else {
            /* no left contexts (filler trees): every call enters the one root list, whatever lc says */
            for (int32_t p = 0; p <= cfg->n_ci; p++) {
                ud->h_lcmap[((size_t)t * (cfg->n_ci + 1) + p) * 2] = proto->rootbuf_base[t] + proto->lcroot_off[t][0];
                ud->h_lcmap[((size_t)t * (cfg->n_ci + 1) + p) * 2 + 1] = proto->lcroot_off[t][1] - proto->lcroot_off[t][0];
            }
        }
    }
    {
        int32_t *p0 = NULL, *p1 = NULL, *p2 = NULL;
        uint8_t *p3 = NULL;
        DM(p0, (size_t)cfg->n_word * 4); DM(p1, (size_t)cfg->n_word * 4); DM(p2, (size_t)cfg->n_word * 4); DM(p3, (size_t)cfg->n_word);
        ud->dict.lwid = p0; ud->dict.fillpen = p1; ud->dict.last_ci = p2; ud->dict.is_filler = p3;
        ud->dict.n_word = cfg->n_word; ud->dict.n_ci = cfg->n_ci;
        if (hipMemcpy(p0, cfg->lwid, (size_t)cfg->n_word * 4, hipMemcpyHostToDevice) != hipSuccess
            || hipMemcpy(p1, cfg->fillpen, (size_t)cfg->n_word * 4, hipMemcpyHostToDevice) != hipSuccess
            || hipMemcpy(p2, cfg->last_ci, (size_t)cfg->n_word * 4, hipMemcpyHostToDevice) != hipSuccess
            || hipMemcpy(p3, cfg->is_filler, (size_t)cfg->n_word, hipMemcpyHostToDevice) != hipSuccess) {
            s3a_set_error("s3a_uttdec_init: upload failed");
            goto fail;
        }
        UPV(ud->d_lcmap, ud->h_lcmap);
    }
    {
        WPar &P = ud->par;
        memset(&P, 0, sizeof P);
        P.wbeam = cfg->wbeam_vh; P.bghist = cfg->bghist; P.maxwpf = cfg->maxwpf; P.maxhist = cfg->maxhistpf;
        P.wordend = cfg->wordend_beam; P.n_lextree = cfg->n_lextree; P.epl = cfg->epl; P.T = T; P.hmmbeam = cfg->hmmbeam;
        for (int32_t t = 0; t < T; t++) P.tree_type[t] = cfg->tree_type[t];
        P.lcmap = ud->d_lcmap;
    }

    /* look-ahead scoring (ku_score_window): the 39/40-dimensional case with a 16-bit log-add table and at most 64
     * Gaussian slots per senone; K = 8 frames per window: from 128 lanes on that is the ~1024 (lane, frame) slots a model pass
     * amortises over, and with fewer lanes a longer window buys < 1 % (measured: one lane 94.6 x real time with K = 64, 93.5
     * with 8; 8 and 32 lanes: no difference) while a refilled lane has to wait for a window boundary.  opts->window_max > 0:
     * up to that many frames so that a pass has ~1024 slots; opts->window: exactly that (0: the per-frame scoring kernels). */
    {
        int32_t K = 0;
        if (d->D4 == D4MAIN && d->CP <= 64 && d->Gpad % 512 == 0) {
            K = 8;
            if (O.window_max > 8) K = min(((O.window_max + 7) / 8) * 8, max(8, (((1024 + n_lanes - 1) / n_lanes + 7) / 8) * 8));
            if (O.window >= 0) K = (O.window + 7) / 8 * 8;
        }
        S.win_K = K;
        ud->times = O.times;
        ud->win_fpc = O.window_fpc;
    }
    ud->lane.resize(n_lanes);
    for (auto &hl : ud->lane) memset((void *)&hl, 0, sizeof hl);
    if (T > WL_MAXT) { s3a_set_error("s3a_uttdec_init: more than %d lextrees", WL_MAXT); goto fail; }
    DM(ud->S.ctx_all, sizeof(UCtx) * n_lanes);
    DM(ud->S.nact_all, (size_t)n_lanes * 2 * WL_MAXT * 4);
    DM(ud->d_fgbase, 64);
    if (hipMemset(ud->d_fgbase, 0, 64) != hipSuccess) goto fail;
    ud->S.fgbase = ud->d_fgbase;
    ud->use_graph = O.graph != 0 && !ud->big_wl && !O.framecheck;
    ud->persist = O.persist >= 0 && !ud->use_graph && !O.framecheck ? (O.persist > 1 ? 3 : O.persist > 0 ? 2 : 1) : 0;     /* (2: whatever the lane count; 3: and with the general barrier only) */
    ud->kf_cluster_opt = O.cluster; ud->kf_last_c = 0; ud->kf_slots = 0; ud->d_kfbar = NULL; ud->kf_counted = 0; ud->d_kfnext = NULL; ud->d_kfargs = ud->h_kfargs = NULL; ud->kf_arg_at = 0;
    ud->sb_scores = NULL; ud->sb_bests = NULL; ud->sb_rows_cap = 0; ud->sb_gdesc_d = ud->sb_gdesc_h = NULL; ud->sb_g_cap = 0;
    ud->sb_rows_max = O.score_rows_max > 0 ? (size_t)O.score_rows_max : 0; ud->kf_order_d = ud->kf_order_h = NULL; ud->kf_order_cap = 0; ud->kf_shared = 0;
    ud->d_kfrelay = NULL; ud->kf_n_relay = 0;
    ud->sb_row0_d = ud->sb_row0_h = NULL; ud->sb_row0_cap = 0;
    ud->kf_ev_n = ud->kf_n_score = ud->kf_n_frames = 0; ud->kf_score_ms = ud->kf_frames_ms = 0.0;
    if (ud->persist) {
        DM(ud->d_kfnext, (size_t)(16 + n_lanes) * 4);
        DM(ud->d_kfargs, sizeof(KfArgs) * KF_ARG_SLOTS);
        if (hipHostMalloc(&ud->h_kfargs, sizeof(KfArgs) * KF_ARG_SLOTS) != hipSuccess) { s3a_set_error("s3a_uttdec_init: pinned allocation failed"); goto fail; }
        DM(ud->d_kfbar, (size_t)n_lanes * 8);
        if (hipMemset(ud->d_kfbar, 0, (size_t)n_lanes * 8) != hipSuccess) goto fail;
        DM(ud->d_kfrelay, ((size_t)(KF_STAGES + 1) * (KF_ST_WORDS + n_lanes) + (size_t)3 * n_lanes) * 4);
        if (ud->device >= 0 && ud->device < 64) { g_kf_live[ud->device]++; ud->kf_counted = 1; ud->kf_shared = kf_device_lock(ud->device) ? 0 : 1; }
    }
    ud->scan_small_from = O.scan_small_from > 0 ? O.scan_small_from : 64;
    ud->hyp_wcap = max_frames + 4;
    if (hipHostMalloc((void **)&ud->h_ctx_up, sizeof(UCtx) * n_lanes) != hipSuccess
        || hipHostMalloc((void **)&ud->h_ctx_dn, sizeof(UCtx) * n_lanes) != hipSuccess
        || hipHostMalloc((void **)&ud->h_hyp_hdr, ((size_t)n_lanes * UH_N + 1) * 4) != hipSuccess
        || hipHostMalloc((void **)&ud->h_hyp_words, (size_t)n_lanes * ud->hyp_wcap * 6 * 4) != hipSuccess) { s3a_set_error("s3a_uttdec_init: pinned allocation failed"); goto fail; }
    memset(ud->h_ctx_dn, 0, sizeof(UCtx) * n_lanes);
    DM(ud->d_hyp_hdr, ((size_t)n_lanes * UH_N + 1) * 4);
    DM(ud->d_hyp_words, (size_t)n_lanes * ud->hyp_wcap * 6 * 4);
    if (hipMemset(ud->S.nact_all, 0, (size_t)n_lanes * 2 * WL_MAXT * 4) != hipSuccess
        || hipMemset(ud->S.ctx_all, 0, sizeof(UCtx) * n_lanes) != hipSuccess) goto fail;
    for (int32_t z = 0; z < n_lanes; z++) {
        HostLane &hl = ud->lane[z];
        hl.ls = s3a_lexsearch_clone(proto, (void *)ud->stream);
        hl.sc = s3a_scorer_init_private(g, cd2cisen, n_sen, n_ci_sen, ds_ratio, cond_ds, ci_pbeam, tighten_factor, max_cd);
        if (!hl.ls || !hl.sc) goto fail;
        if (z == 0) {
            S.gp_n = hl.sc->gp_n;
            S.ci_pbeam = hl.sc->ci_pbeam;
            S.ci_pbeam_tight = (int32_t)((float)hl.sc->ci_pbeam * hl.sc->tighten_factor);
            S.max_cd = max_cd; S.tighten = hl.sc->tighten_factor;
            S.ncomp = hl.sc->ncomp_d;           /* identical in every lane's scorer (model facts) */
            S.cd2cisen = hl.sc->cd2cisen_d;
        }
        s3a_lexsearch_t *ls = hl.ls;
        ULane &u = hl.d;
        u.sc = ls->d_sc; u.hist = ls->d_hist; u.outs = ls->d_outs; u.outh = ls->d_outh; u.bests = ls->d_bests;
        u.frame = ls->d_frame; u.pos = ls->d_pos; u.posf = ls->d_posf; u.act[0] = ls->d_act[0]; u.act[1] = ls->d_act[1];
        /* the clone's list lengths move into the engine's array (borrowed: s3a_uttdec_free takes them back) */
        (void)hipFree(ls->d_nact[0]); (void)hipFree(ls->d_nact[1]);
        ls->d_nact[0] = ud->S.nact_all + ((size_t)z * 2) * WL_MAXT; ls->d_nact[1] = ud->S.nact_all + ((size_t)z * 2 + 1) * WL_MAXT;
        u.nact[0] = ls->d_nact[0]; u.nact[1] = ls->d_nact[1]; u.turn = ls->d_turn; u.selfemit = ls->d_selfemit;
        u.cnt = ls->d_cnt; u.base = ls->d_cand; u.best = ls->d_best; u.exits = ls->d_exit; u.nexit = ls->d_nexit;
        u.first = ls->d_first; u.eflag = ls->d_eflag; u.hbin = ls->d_hbin; u.done = ls->d_done; u.ctot = ls->d_ctot;
        u.n0 = ls->d_n0; u.pstamp = ls->d_pstamp; u.propf = ls->d_candf; u.poswid = ls->d_poswid; u.posout = ls->d_posout;
        u.scan_flag = ls->d_scan_flag; u.scan_agg = ls->d_scan_agg; u.scan_pre = ls->d_scan_pre; u.key = ls->d_key;
        u.sen_act = hl.sc->act_d; u.scr = hl.sc->scr_d; u.misc = hl.sc->misc_d; u.bstidx = hl.sc->bstidx_d;
        u.bstscr = hl.sc->bstscr_d; u.updatetime = hl.sc->updatetime_d; u.gpart = hl.sc->gpart_d;
        DM(u.cs_need, (size_t)(cs->n_comstate + 1) * 4); DM(u.cs_val, (size_t)(cs->n_comstate + 1) * 4); DM(u.dynbeam, 16);
        DM(u.cs_wl, (size_t)(cs->n_comstate + 1) * 4); DM(u.cs_wn, 16); u.ent = ls->d_ent; DM(u.posbest, (size_t)(proto->N + 64) * 4); DM(u.posps, (size_t)(proto->N + 64) * 4);
        DM(u.senbits, KF_SENBITS / 8); if (hipMemset(u.senbits, 0, KF_SENBITS / 8) != hipSuccess) goto fail;
        if (hipMemset(u.cs_wn, 0, 16) != hipSuccess) goto fail; DM(u.pstamp8, (size_t)proto->n_pset + 64); S.n_pset_bytes = proto->n_pset + 64;
        DM(u.plist, (size_t)(proto->N + 64) * 4); DM(u.pcnt, 16); DM(u.claim, (size_t)(proto->N + 64) * 4);
        if (hipMemset(u.pcnt, 0, 16) != hipSuccess) goto fail;
        if (hipMemset(u.pstamp8, 0xff, (size_t)proto->n_pset + 64) != hipSuccess) goto fail;
        if (S.win_K > 0) {
            DM(u.win, (size_t)S.win_K * n_sen * 4); DM(u.winb, (size_t)S.win_K * n_sen);
            /* ku_select fills the first USEL_G columns of gpart[]; the others stay neutral */
            if (fill32(ud->stream, u.gpart, INT_MIN, (size_t)max(S.gp_n, 0)) != S3A_OK
                || fill32(ud->stream, u.gpart + max(S.gp_n, 0), 0, (size_t)2 * max(S.gp_n, 0)) != S3A_OK) goto fail;
        }
        if (fill32(ud->stream, u.cs_need, -1, (size_t)cs->n_comstate + 1) != S3A_OK) goto fail;
        u.ctx = ud->S.ctx_all + z;
        DM(u.pack, (size_t)(6 * T + 16 + 3 * proto->pack_max_exits) * 4);
        if (wlane_alloc(u.w, ud->vh_cap, max_frames, ud->ex_cap, ud->cand_cap, ud->new_cap, cfg->n_word, ud->stream) != S3A_OK) goto fail;
        hl.h_ctx = ud->h_ctx_dn + z;
        if (hipHostMalloc((void **)&hl.h_st, 16 * 4) != hipSuccess
            || hipHostMalloc((void **)&hl.h_fstat, (size_t)max_frames * 8 * 4) != hipSuccess) {
            s3a_set_error("s3a_uttdec_init: pinned allocation failed");
            goto fail;
        }
    }
    {
        std::vector<ULane> tmp(n_lanes);
        for (int32_t z = 0; z < n_lanes; z++) tmp[z] = ud->lane[z].d;
        DM(ud->d_lanes, sizeof(ULane) * n_lanes);
        if (hipMemcpy(ud->d_lanes, tmp.data(), sizeof(ULane) * n_lanes, hipMemcpyHostToDevice) != hipSuccess) goto fail;
    }
    if (hipStreamSynchronize(ud->stream) != hipSuccess) goto fail;
    return ud;
fail:
    s3a_uttdec_free(ud);
    return NULL;
}

/* a lane whose previous utterance stopped on an error in mid-frame: everything from scratch */
static int32_t
lane_scrub(s3a_uttdec_t *ud, int32_t z)
{
    HostLane &hl = ud->lane[z];
    int32_t rc;
    if (!hl.dirty) return S3A_OK;
    const WLane &w = hl.d.w;
    const size_t hs = (size_t)w.hmask + 1;
    if ((rc = s3a_lexsearch_reset(hl.ls)) != S3A_OK) return rc;
    HIPCHK(hipMemsetAsync(w.hkey, 0, hs * 8, ud->stream)); HIPCHK(hipMemsetAsync(w.hbest, 0, hs * 8, ud->stream));
    HIPCHK(hipMemsetAsync(w.hfirst, 0xff, hs * 4, ud->stream));
    if ((rc = fill32(ud->stream, w.wfirst, INT_MAX, ud->cfg.n_word)) != S3A_OK || (rc = fill32(ud->stream, w.wbest, INT_MIN, ud->cfg.n_word)) != S3A_OK) return rc;
    hl.dirty = 0;
    return S3A_OK;
}

/* the context an utterance starts from: srch_TST_begin's lextree_enter calls (silence as left context into unigram tree 0
 * and filler tree n_lextree), frame 0, the first beam */
static int32_t
utt_context(const s3a_uttdec_t *ud, UCtx &x, const float *d_feat, int32_t nfr, int32_t scan_epoch, int32_t f0, int32_t utt)
{
    const s3a_wordlevel_cfg_t &c = ud->cfg;
    memset(&x, 0, sizeof x);
    x.feat = d_feat;
    x.active = 1; x.cf = 0; x.nfr = nfr; x.cur = 0; x.n_lextrans = 1; x.thresh = c.hmmbeam;
    x.scan_epoch = scan_epoch;      /* k_dec_scan's flags are never reset: the stamps keep growing */
    x.f0 = f0; x.utt = utt;
    const int32_t nci = c.n_ci;
    const int32_t *m0 = &ud->h_lcmap[((size_t)0 * (nci + 1) + c.sil_ci) * 2];
    const int32_t *m1 = &ud->h_lcmap[((size_t)c.n_lextree * (nci + 1) + nci) * 2];
    if (m0[1] < 0 || m1[1] < 0) { s3a_set_error("s3a_uttdec_decode: silence is not a left context of the unigram lextree"); return S3A_EINVAL; }
    x.calls[0] = 0; x.calls[1] = 0; x.calls[2] = m0[0]; x.calls[3] = 0;
    x.calls[4] = 0; x.calls[5] = 0; x.calls[6] = m1[0]; x.calls[7] = m0[1];
    x.groups[0] = 0; x.groups[1] = 0; x.groups[2] = m0[1]; x.groups[3] = 0;
    x.groups[4] = c.n_lextree; x.groups[5] = m0[1]; x.groups[6] = m0[1] + m1[1]; x.groups[7] = 1;
    x.n_calls = 2; x.n_groups = 2; x.n_ent = m0[1] + m1[1];
    return S3A_OK;
}

/* srch_TST_begin (srch_time_switch_tree.c:457-512) for one lane: the dummy <s> history entry, the
 * root entries with silence as left context into unigram tree 0 and filler tree n_lextree */
static int32_t
lane_begin(s3a_uttdec_t *ud, int32_t z, const float *feat, int32_t nfr, int32_t feat_stride, bool feat_on_device)
{
    HostLane &hl = ud->lane[z];
    const int32_t D4x4 = ud->S.D4 * 4;
    int32_t rc;
    if (nfr <= 0 || nfr > ud->max_frames) { s3a_set_error("s3a_uttdec_decode: %d frames (1..%d)", nfr, ud->max_frames); return S3A_EINVAL; }
    /* features, rows padded to the scorer's float4 stride */
    const float *d_feat_use = feat;
    if (!feat_on_device) {
        const size_t need = (size_t)nfr * D4x4;
        /* (buffers grow with slack: every hipMalloc / hipHostMalloc / free stalls ALL streams of the device, the other
         * engines' too -- a lane must not pay that each time its utterance is a little longer than the last) */
        const size_t grow = (size_t)min(ud->max_frames, max(nfr + nfr / 2, 1024)) * D4x4;
        if (need > hl.feat_cap) {
            if (hl.d_feat) (void)hipFree(hl.d_feat);
            hl.d_feat = NULL; hl.feat_cap = 0;
            if (hipMalloc((void **)&hl.d_feat, grow * 4) != hipSuccess) { s3a_set_error("s3a_uttdec_decode: feature buffer"); return S3A_ENOMEM; }
            hl.feat_cap = grow;
        }
        if (need > hl.h_feat_cap) {
            if (hl.h_feat) (void)hipHostFree(hl.h_feat);
            hl.h_feat = NULL; hl.h_feat_cap = 0;
            if (hipHostMalloc((void **)&hl.h_feat, grow * 4) != hipSuccess) { s3a_set_error("s3a_uttdec_decode: pinned feature buffer"); return S3A_ENOMEM; }
            hl.h_feat_cap = grow;
        }
        for (int32_t t = 0; t < nfr; t++) {
            float *row = hl.h_feat + (size_t)t * D4x4;
            memcpy(row, feat + (size_t)t * feat_stride, sizeof(float) * ud->veclen);
            for (int32_t k = ud->veclen; k < D4x4; k++) row[k] = 0.0f;
        }
        HIPCHK(hipMemcpyAsync(hl.d_feat, hl.h_feat, need * 4, hipMemcpyHostToDevice, ud->stream));
        d_feat_use = hl.d_feat;
    }
    else if (feat_stride != D4x4) {
        s3a_set_error("s3a_uttdec_decode_dev: device features must have rows of %d floats (zero padded)", D4x4);
        return S3A_EINVAL;
    }
    hl.nfr = nfr;
    /* lextree + scorer state as after lextree_utt_end / at srch_TST_begin */
    if ((rc = lane_scrub(ud, z)) != S3A_OK) return rc;
    /* (lextree_utt_end, srch_TST_begin's resets and the history table's entry 0: ku_lanes_end / ku_lanes_begin, all
     * lanes in two launches -- uttdec_decode; here only the host-side bookkeeping of the lane's objects) */
    hl.ls->cur = 0;
    hl.ls->last_nnxt = 0; hl.ls->hist_bound = 0; hl.ls->row_bound = 1;
    hl.ls->nnxt_t.assign(hl.ls->n_tree, 0);
    hl.sc->skip_count = 0;
    UCtx &x = *hl.h_ctx;
    if (hl.epoch < 1) hl.epoch = 1;
    if ((rc = utt_context(ud, x, d_feat_use, nfr, hl.epoch, 0, -1)) != S3A_OK) return rc;
    hl.epoch += nfr + 1;
    ud->h_ctx_up[z] = x;            /* (uploaded with the other lanes': uttdec_decode) */
    return S3A_OK;
}

/* ---- look-ahead scoring: geometry + launch of ku_score_window ---- */
struct UwGeom { int32_t nt, fpc, n_chunks, n_tiles, grid, tab_lds; size_t lds; };

static size_t
uw_lds_bytes(int32_t fpc, int32_t nt, bool tab_lds, uint32_t tab_size)
{
    size_t b = (size_t)fpc * D4MAIN * 4 * sizeof(float);
    b += (size_t)(nt / 64) * UW_FB * 65 * sizeof(int32_t);
    b = (b + 15) & ~(size_t)15;
    b += (size_t)(fpc / UW_FB) * sizeof(UwGroup);
    b = (b + 15) & ~(size_t)15;
    if (tab_lds) b += ((size_t)tab_size * 2 + 15) & ~(size_t)15;
    return b;
}

/* (lane, frame) slots per chunk: a workgroup needs ~100 KB of LDS, so one is resident per CU and the grid runs in rounds
 * of n_cu workgroups; time ~ rounds x (slots per chunk + a start-up worth ~10 slots): s3a_device.hip, pick_fpc */
static UwGeom
uw_geometry(const s3a_uttdec_t *ud, int32_t n_lanes, int32_t total_slots = 0)
{
    const UShared &S = ud->S;
    const int32_t total = total_slots > 0 ? total_slots : n_lanes * S.win_K, n_cu = max(1, ud->g->dev->n_cu);
    UwGeom q;
    q.tab_lds = total >= 2 * UW_FB ? 1 : 0;
    q.nt = q.tab_lds ? 512 : 256;
    q.n_tiles = S.Gpad / q.nt;
    int32_t best_fpc = UW_FB;
    int64_t best_cost = -1;
    for (int32_t nc = 1; nc <= 512; nc++) {
        int32_t fpc = (total + nc - 1) / nc;
        fpc = ((fpc + UW_FB - 1) / UW_FB) * UW_FB;
        if (fpc > 256) continue;
        const int32_t chunks = (total + fpc - 1) / fpc;
        const int64_t rounds = ((int64_t)q.n_tiles * chunks + n_cu - 1) / n_cu;
        const int64_t cost = rounds * (fpc + 10);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_fpc = fpc; }
        if (fpc <= UW_FB) break;
    }
    if (ud->win_fpc > 0) best_fpc = min(256, ((ud->win_fpc + UW_FB - 1) / UW_FB) * UW_FB);
    q.fpc = best_fpc;
    q.n_chunks = (total + q.fpc - 1) / q.fpc;
    q.grid = ((q.n_tiles + 7) / 8) * q.n_chunks * 8;
    q.lds = uw_lds_bytes(q.fpc, q.nt, q.tab_lds != 0, S.tab_size);
    if (q.lds > 160 * 1024) { q.tab_lds = 0; q.lds = uw_lds_bytes(q.fpc, q.nt, false, S.tab_size); }
    return q;
}

template <int CP, bool EXACT>
static hipError_t
uw_launch_cp(const s3a_uttdec_t *ud, const UwGeom &q, int32_t n, int32_t f0, hipStream_t st, bool attr_only, const UwGroup *gdesc = NULL, int32_t n_g = 0)
{
    /* (attr_only: the function attribute alone -- before a stream capture, where it may not be set) */
#define UW_GO(TAB, NT) do { auto kern = ku_score_window<CP, EXACT, TAB, NT>;                                            \
        if (q.lds > 64 * 1024 && hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                                       160 * 1024) != hipSuccess) return hipGetLastError();             \
        if (!attr_only) hipLaunchKernelGGL(kern, dim3(q.grid), dim3(NT), q.lds, st, ud->d_lanes, ud->S, n, f0, ud->S.win_K, q.fpc,        \
                           q.n_chunks, q.n_tiles, gdesc, n_g); } while (0)
    if (q.nt == 512) { if (q.tab_lds) UW_GO(true, 512); else UW_GO(false, 512); }
    else UW_GO(false, 256);
#undef UW_GO
    return hipGetLastError();
}

static int32_t
uw_launch(const s3a_uttdec_t *ud, int32_t n, int32_t f0, bool attr_only = false, const UwGroup *gdesc = NULL, int32_t n_g = 0)
{
    /* (gdesc: n_g groups of 8 frames from the host's table instead of the lanes' windows) */
    const UwGeom q = uw_geometry(ud, n, gdesc ? n_g * UW_FB : 0);
    hipError_t e;
#define UW_CASE(cp) case cp: e = ud->exact ? uw_launch_cp<cp, true>(ud, q, n, f0, ud->stream, attr_only, gdesc, n_g) : uw_launch_cp<cp, false>(ud, q, n, f0, ud->stream, attr_only, gdesc, n_g); break
    switch (ud->S.CP) {
    UW_CASE(1); UW_CASE(2); UW_CASE(4); UW_CASE(8); UW_CASE(16); UW_CASE(32);
    default: e = ud->exact ? uw_launch_cp<64, true>(ud, q, n, f0, ud->stream, attr_only, gdesc, n_g) : uw_launch_cp<64, false>(ud, q, n, f0, ud->stream, attr_only, gdesc, n_g); break;
    }
#undef UW_CASE
    if (e != hipSuccess) { s3a_set_error("ku_score_window launch failed: %s", hipGetErrorString(e)); return S3A_EHIP; }
    return S3A_OK;
}

/* kernel classes of a frame (s3a_uttdec_profile) */
enum { UK_ENTER1, UK_ENTER2, UK_ENTER3, UK_GATED_CI, UK_GATED_CD, UK_COMSEN, UK_HMM_EVAL, UK_HIST_COUNT, UK_HIST_SORT, UK_WEAK,
       UK_RESOLVE, UK_SCAN, UK_EMIT, UK_WORD, UK_WL_P2, UK_WL_P3, UK_WL_P4, UK_WL_P5, UK_WL_FIN, UK_WINDOW, UK_N };
static const char *const uk_names[UK_N] = { "ku_enter1", "ku_enter2", "ku_enter3_mark", "ku_gated_ci", "ku_gated_cd",
    "ku_comsen_max", "ku_hmm_eval", "ku_hist_count", "ku_hist_sort", "ku_weak", "ku_resolve", "ku_scan", "ku_emit", "ku_emit_word", "ku_wl_p2", "ku_wl_p3", "ku_wl_p4ab", "ku_wl_p5", "ku_wl_finish", "ku_score_window" };

static int32_t
enqueue_frame(s3a_uttdec_t *ud, int32_t n, int32_t f, bool prof)
{
    hipStream_t st = ud->stream;
    const ULane *LN = ud->d_lanes;
    const UShared &S = ud->S;
    const int32_t T = S.T;
    /* profiled frames bracket every launch with HIP events on the launch stream */
#define UKL(cls, ...) do { hipEvent_t a_ = NULL, b_ = NULL;                                                   \
        if (prof) { (void)hipEventCreate(&a_); (void)hipEventCreate(&b_); (void)hipEventRecord(a_, st); }        \
        hipLaunchKernelGGL(__VA_ARGS__);                                                                         \
        if (prof) { (void)hipEventRecord(b_, st); ud->prof_ev.push_back({ cls, a_, b_ }); } } while (0)
    UKL(UK_ENTER1, ku_enter1, dim3(ud->g_ent, 1, n), dim3(256), 0, st, LN, S, f);
    /* (from 64 lanes on: 256-thread workgroups, as the scan below: 365 -> 371 k frames/s with four engines on the chip) */
    if (n >= ud->scan_small_from)
        UKL(UK_ENTER2, ku_enter2<256>, dim3(G_ENTER2_MANY, 1, n), dim3(256), 0, st, LN, S, f);
    else
        UKL(UK_ENTER2, ku_enter2<SCAN_THREADS>, dim3(n >= ud->many ? 12 : WL_MAXCALL, 1, n), dim3(SCAN_THREADS), 0, st, LN, S, f);
    UKL(UK_ENTER3, ku_enter3_mark, dim3(ud->g_mark, 1, n), dim3(M3BLOCK), 0, st, LN, S, f);
    const int32_t g_ci = (S.n_ci_sen * S.CP + 255) / 256, g_cd = ((S.n_sen - S.n_ci_sen) * S.CP + 255) / 256;
    const int32_t g_cs = (S.n_cs + 255) / 256;         /* composite senones: a wave looks at 64 */
    if (S.win_K > 0) {
        /* look-ahead scoring: every K frames one pass over the model for all lanes' coming K frames; per frame the
         * composite senones' members join the mask and the gate selects (ku_select) */
        if (f % S.win_K == 0) {
            hipEvent_t a_ = NULL, b_ = NULL;
            if (prof) { (void)hipEventCreate(&a_); (void)hipEventCreate(&b_); (void)hipEventRecord(a_, st); }
            const int32_t rc = uw_launch(ud, n, f);
            if (rc != S3A_OK) return rc;
            if (prof) { (void)hipEventRecord(b_, st); ud->prof_ev.push_back({ UK_WINDOW, a_, b_ }); }
        }
        if (g_cs) UKL(UK_GATED_CI, ku_comsen_mark, dim3(g_cs, 1, n), dim3(256), 0, st, LN, S, f);
        if (S.max_cd < S.n_sen - S.n_ci_sen) UKL(UK_GATED_CI, ku_dyn_ci_beam, dim3(1, 1, n), dim3(1024), 0, st, LN, S, f);
        const int32_t g_sel = max(1, min(USEL_G, min(S.gp_n, (S.n_sen - S.n_ci_sen + 255) / 256)));
        if (ud->exact) UKL(UK_GATED_CD, ku_select<true>, dim3(g_sel, 1, n), dim3(256), 0, st, LN, S, f, S.win_K);
        else UKL(UK_GATED_CD, ku_select<false>, dim3(g_sel, 1, n), dim3(256), 0, st, LN, S, f, S.win_K);
    }
    else {
    /* from UG_FB lanes on the CD senones of all lanes are ONE pass over the model (39/40-dimensional features,
     * >= UG_FB Gaussian slots per senone); (S3A_UTT_NO_MULTI: the per-lane kernel whatever the lane count -- tests) */
    const bool multi = n >= UG_FB && S.D4 == D4MAIN && S.CP >= UG_FB && S.CP <= 64 && g_cd > 0 && S.gp_n == g_cd
        && !ud->no_multi;
    const int32_t gz = (n + UG_MAX - 1) / UG_MAX, groups = (min(n, UG_MAX) + UG_FB - 1) / UG_FB;
    const int32_t gy_env = ud->gy;
    const dim3 gm(g_cd, gy_env > 0 ? min(groups, gy_env) : max(1, min(groups, 2 * ud->g->dev->n_cu / max(1, g_cd * gz))), gz);
    if (ud->exact) {
        if (g_ci) UKL(UK_GATED_CI, (ku_gated<true, true>), dim3(g_ci + g_cs, 1, n), dim3(256), 0, st, LN, S, f);
        if (S.max_cd < S.n_sen - S.n_ci_sen) UKL(UK_GATED_CI, ku_dyn_ci_beam, dim3(1, 1, n), dim3(1024), 0, st, LN, S, f);
        if (multi) UKL(UK_GATED_CD, (ku_gated_cd_multi<true>), gm, dim3(256), 0, st, LN, S, n, f);
        else if (g_cd) UKL(UK_GATED_CD, (ku_gated<true, false>), dim3(g_cd, 1, n), dim3(256), 0, st, LN, S, f);
    }
    else {
        if (g_ci) UKL(UK_GATED_CI, (ku_gated<false, true>), dim3(g_ci + g_cs, 1, n), dim3(256), 0, st, LN, S, f);
        if (S.max_cd < S.n_sen - S.n_ci_sen) UKL(UK_GATED_CI, ku_dyn_ci_beam, dim3(1, 1, n), dim3(1024), 0, st, LN, S, f);
        if (multi) UKL(UK_GATED_CD, (ku_gated_cd_multi<false>), gm, dim3(256), 0, st, LN, S, n, f);
        else if (g_cd) UKL(UK_GATED_CD, (ku_gated<false, false>), dim3(g_cd, 1, n), dim3(256), 0, st, LN, S, f);
    }
    }
    if (g_cs) UKL(UK_COMSEN, ku_comsen_max, dim3(g_cs, 1, n), dim3(256), 0, st, LN, S, f);
    if (S.ne == 5) {
        if (ud->eval_block == 256) UKL(UK_HMM_EVAL, (ku_hmm_eval<256, 5>), dim3(ud->g_eval, T, n), dim3(256), 0, st, LN, S, f);
        else UKL(UK_HMM_EVAL, (ku_hmm_eval<64, 5>), dim3(ud->g_eval, T, n), dim3(64), 0, st, LN, S, f);
    }
    else if (ud->eval_block == 256)
        UKL(UK_HMM_EVAL, (ku_hmm_eval<256, 3>), dim3(ud->g_eval, T, n), dim3(256), 0, st, LN, S, f);
    else
        UKL(UK_HMM_EVAL, (ku_hmm_eval<64, 3>), dim3(ud->g_eval, T, n), dim3(64), 0, st, LN, S, f);
    /* the not active nodes from the list of stamped parent sets (the stamping pass then also lists the sets) -- not with wide
     * beams (tens of thousands of active HMMs, thousands of listed sets per frame: configs[4] is faster with the sweep) */
    const int32_t by_parents = n >= ud->many && !ud->big_wl && !s3a_variants()->resolve_sweep ? 1 : 0;
    {
        /* many lanes: the (rare) histogram sort rides on the count's launch (its 256-thread form is the one used there anyway) */
        const int32_t own_sort = ud->hist_possible && n >= ud->scan_small_from && !s3a_variants()->hist_sort_launch ? 1 : 0;
        UKL(UK_HIST_COUNT, ku_hist_count, dim3(max(1, min((S.maxn + DBLOCK - 1) / DBLOCK, n >= ud->many ? G_HIST_MANY : 64)), T, n), dim3(DBLOCK), 0, st, LN, S, f, own_sort, by_parents);
        if (ud->hist_possible && !own_sort) {
            /* (from 64 lanes on 256 threads: the launch's 128 workgroups mostly only leave, and small ones find a slot sooner) */
            if (n >= ud->scan_small_from) UKL(UK_HIST_SORT, ku_hist_sort<256>, dim3(1, 1, n), dim3(256), 0, st, LN, S, f, by_parents);
            else UKL(UK_HIST_SORT, ku_hist_sort<SCAN_THREADS>, dim3(1, 1, n), dim3(SCAN_THREADS), 0, st, LN, S, f, by_parents);
        }
    }
    if (ud->weak_possible && S.pheurtype > 0) UKL(UK_WEAK, ku_weak_heur, dim3(T, 1, n), dim3(1024), 0, st, LN, S, f);
    else {
        if (ud->weak_possible) UKL(UK_WEAK, ku_weak, dim3(1, 1, n), dim3(SCAN_THREADS), 0, st, LN, S, f);
        if (S.pheurtype > 0) UKL(UK_RESOLVE, ku_heur_thresh, dim3(T, 1, n), dim3(1024), 0, st, LN, S, f);
    }
    {   /* (the active HMMs by list position: ud->g_res workgroups that loop; the rest: a sweep, UR_K nodes per thread) */
        const int32_t GB = ((S.N + UR_K - 1) / UR_K + RSBLOCK - 1) / RSBLOCK;
        if (by_parents) {
            if (S.pheurtype > 0) UKL(UK_RESOLVE, ku_resolve_plist<true>, dim3(ud->g_res + UR_GL, 1, n), dim3(RSBLOCK), 0, st, LN, S, f);
            else UKL(UK_RESOLVE, ku_resolve_plist<false>, dim3(ud->g_res + UR_GL, 1, n), dim3(RSBLOCK), 0, st, LN, S, f);
        }
        else if (S.pheurtype > 0) {
            if (n >= ud->many) UKL(UK_RESOLVE, ku_resolve_lists<true>, dim3(ud->g_res + GB, 1, n), dim3(RSBLOCK), 0, st, LN, S, f);
            else UKL(UK_RESOLVE, ku_resolve<true>, dim3((S.N + RSBLOCK - 1) / RSBLOCK, 1, n), dim3(RSBLOCK), 0, st, LN, S, f);
        }
        else if (n >= ud->many && ud->urk == 16) { const int32_t GB16 = ((S.N + 15) / 16 + RSBLOCK - 1) / RSBLOCK; UKL(UK_RESOLVE, (ku_resolve_lists<false, 16>), dim3(ud->g_res + GB16, 1, n), dim3(RSBLOCK), 0, st, LN, S, f); }
        else if (n >= ud->many && ud->urk == 4) { const int32_t GB4 = ((S.N + 3) / 4 + RSBLOCK - 1) / RSBLOCK; UKL(UK_RESOLVE, (ku_resolve_lists<false, 4>), dim3(ud->g_res + GB4, 1, n), dim3(RSBLOCK), 0, st, LN, S, f); }
        else if (n >= ud->many) UKL(UK_RESOLVE, ku_resolve_lists<false>, dim3(ud->g_res + GB, 1, n), dim3(RSBLOCK), 0, st, LN, S, f);
        else UKL(UK_RESOLVE, ku_resolve<false>, dim3((S.N + RSBLOCK - 1) / RSBLOCK, 1, n), dim3(RSBLOCK), 0, st, LN, S, f);
    }
    /* the scan over long lists: either one workgroup per possible chunk, chained (a chunk waits for chunks with SMALLER
     * numbers only = workgroups dispatched before it: the wait ends whatever else runs on the chip), when that is no more
     * workgroups than the chip holds -- a single lane, wide beams --, or one workgroup per tree that walks the chunks itself.
     * (Round 2 had a few workgroups per tree taking the chunks in turn: workgroup 0's second chunk then waits for the LAST
     * workgroup's first, a workgroup dispatched after it -- with several engines on the chip every slot can end up held by
     * such a waiter whose partner is not dispatched yet: observed as WL_E_SCAN time-outs with 8 engines of 64 lanes;
     * opts->scan_g still asks for that variant.) */
    const int32_t scan_gc = ud->scan_gc > 0 ? min(ud->scan_gc, ud->scan_nc) : ((long long)T * n * ud->scan_nc <= 768 ? ud->scan_nc : 1);
    /* (one workgroup per tree: it walks the chunks with the totals in a register -- no flags, no look-back) */
    /* (from 64 lanes on, one workgroup per tree: 256 threads -- four times the chunks to walk, 53 instead of 29 us per 128-lane
     * launch alone, but with four engines on the chip a 256-thread workgroup finds a slot where a 1024-thread one waits for half
     * a CU: 355 -> 362 k frames/s; with 32-lane engines the other way round: 268 -> 259 k) */
    if (scan_gc == 1 && n >= ud->scan_small_from)
        UKL(UK_SCAN, ku_scan<256>, dim3(T, 1, n), dim3(256), 0, st, LN, S, 1, 1, f);
    else
        UKL(UK_SCAN, ku_scan<SCAN_THREADS>, dim3(T * scan_gc, 1, n), dim3(SCAN_THREADS), 0, st, LN, S, scan_gc == 1 ? 1 : ud->scan_nc, scan_gc, f);
    /* (many lanes: fewer emission workgroups per tree -- each sweeps further -- instead of thousands of idle ones) */
    UKL(UK_WORD, ku_emit_word, dim3(1 + (n >= ud->many && !ud->big_wl ? 1024 / WL_THREADS : UE_WG_PER_TREE) * T, 1, n), dim3(WL_THREADS), 0, st, LN, S, ud->lm->d, ud->dict, ud->par, f, ud->big_wl);
    if (ud->big_wl) {
        const dim3 gb(WL_BIG_G, 1, n), one(1, 1, n), tb(WL_THREADS);
        UKL(UK_WL_P2, ku_wl_p2, gb, tb, 0, st, LN, ud->lm->d, ud->dict, ud->par, f);
        UKL(UK_WL_P3, ku_wl_p3, gb, tb, 0, st, LN, ud->dict, ud->par, f);
        UKL(UK_WL_P4, ku_wl_p4a, gb, tb, 0, st, LN, ud->par, f);
        UKL(UK_WL_P4, ku_wl_p4b, gb, tb, 0, st, LN, ud->par, f);
        UKL(UK_WL_P5, ku_wl_p5, gb, tb, 0, st, LN, ud->dict, ud->par, f);
        UKL(UK_WL_FIN, ku_wl_finish, one, tb, 0, st, LN, ud->lm->d, ud->dict, ud->par, f);
    }
    if (ud->d_dbg) hipLaunchKernelGGL(ku_framecheck, dim3(64, 1, n), dim3(256), 0, st, LN, S, f, ud->d_dbg);
#undef UKL
    HIPCHK(hipGetLastError());
    return S3A_OK;
}

template <typename TP>
static int32_t
q_grow(TP **d, TP **h, size_t *cap, size_t need, const char *what)
{
    if (need <= *cap) return S3A_OK;
    const size_t grow = need + need / 4 + 64;
    if (*d) (void)hipFree(*d);
    if (h && *h) (void)hipHostFree(*h);
    *d = NULL; if (h) *h = NULL; *cap = 0;
    if (hipMalloc((void **)d, grow * sizeof(TP)) != hipSuccess || (h && hipHostMalloc((void **)h, grow * sizeof(TP)) != hipSuccess)) {
        s3a_set_error("s3a_uttdec_decode_queue: out of memory (%s, %zu bytes)", what, grow * sizeof(TP));
        return S3A_ENOMEM;
    }
    *cap = grow;
    return S3A_OK;
}

/* ---- ku_frames: the frames [fg0, fg0 + nf) of all lanes as ONE launch (behind the look-ahead pass that scores them) ---- */
/* does this engine, as it is configured now, run its frames through ku_frames? */
/* (n: the lanes the call keeps busy.  Measured with one engine on the hub4-shaped task, ku_frames : launches, k frames/s -- 256 lanes
 * 443 : 288, 128 lanes (clusters of 3) 407 : 223, 64 lanes (4) 273 : 153, 32 lanes (4) 159 : 110, 16 lanes (8) 91 : 73, 8 lanes (8)
 * 54.7 : 45.2, 4 lanes (16) 28.2 : 27.7, one lane (16) 7.6 : 9.4 -- with the clusters' XCD-local barrier (the general one, an L2
 * write-back per workgroup and step: 128 lanes 284, 64 lanes 216, and the launches won below 96 lanes).  Larger clusters do not pay
 * (32 lanes: 159 / 146 / 94 k with 4 / 8 / 15 workgroups; 64 lanes: 273 / 219 k with 4 / 7): the entry ranking and the word level
 * stay one workgroup's.  Below KF_MIN_LANES the launches stay) */
#define KF_MIN_LANES 8
#define KF_CLUSTER_MAX(n) ((n) <= 16 ? 8 : 4)
static bool
kf_served(const s3a_uttdec_t *ud, int32_t n)
{
    const UShared &S = ud->S;
    /* (a wide-beam engine -- big_wl, configs[4] -- is SERVED, word level and all, but keeps the launches unless asked: 23 000 HMMs and 300 000
     * word-level candidates per lane-frame want the whole chip per step, not a cluster of 8 workgroups -- 64 lanes: 10.4 k frames/s through
     * ku_frames against 30.5 k through the launches, 128 lanes 19.4 k : 36.1 k; profiles/r6_experiments.txt 8) */
    return ud->persist && (ud->persist > 1 || (n >= KF_MIN_LANES && !ud->big_wl)) && S.win_K > 0 && !ud->d_dbg
        && ud->prof_every == 0 && ((S.ne == 3 && S.nodepk) || S.ne == 5) && S.T <= WL_MAXT && S.n_sen <= KF_SENBITS;
}

/* workgroups of ku_frames the device holds at once: a cluster's workgroups wait for one another */
template <int NE, bool EXACT>
static void
kf_slots_t(s3a_uttdec_t *ud)
{
    if (ud->kf_slots == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ku_frames<NE, EXACT>, KF_NT, 0) != hipSuccess || nb < 1) nb = 1;
        ud->kf_slots = nb * ud->g->dev->n_cu;
    }
}
/* ... per XCD, less a margin (the occupancy query can be one workgroup per CU high, MI355X_MICROARCH "Residency") */
static int32_t
kf_usable_per_xcd(s3a_uttdec_t *ud)
{
    if (ud->kf_slots == 0) {
        if (ud->S.ne == 5) { if (ud->exact) kf_slots_t<5, true>(ud); else kf_slots_t<5, false>(ud); }
        else { if (ud->exact) kf_slots_t<3, true>(ud); else kf_slots_t<3, false>(ud); }
    }
    const int32_t per_xcd = max(1, ud->kf_slots / 8);
    return per_xcd - (per_xcd > 8 ? 4 : 0);
}
/* is this engine free to size clusters for the whole device?  (alone in its process AND no other process holds the device's lock) */
static bool
kf_alone(const s3a_uttdec_t *ud)
{
    return ud->device >= 0 && ud->device < 64 && g_kf_live[ud->device].load() <= 1 && !ud->kf_shared;
}
/* workgroups per lane for n lanes: the lanes' clusters share the XCDs evenly (lane z on XCD z % 8) and must all be resident */
static int32_t
kf_choose_c(s3a_uttdec_t *ud, int32_t n)
{
    const int32_t usable = kf_usable_per_xcd(ud), per_xcd = max(1, ud->kf_slots / 8), lanes_per_xcd = (n + 7) / 8;
    int32_t C = 1;
    if (ud->kf_cluster_opt > 0) C = ud->kf_cluster_opt;
    else if (kf_alone(ud)) {
        const int32_t cmax = ud->big_wl ? 8 : KF_CLUSTER_MAX(n);       /* (wide beams: seven times the HMMs, a word level in chunks) */
        C = min(per_xcd / lanes_per_xcd, cmax);
        /* ... but an EVEN fill first: a cluster is as fast as its slowest workgroup, and with one and a half workgroups per CU some share
         * a CU and some do not -- 128 lanes: 429.5 k frames/s as clusters of 2 (one workgroup per CU) against 406.9 k as clusters of 3
         * (profiles/r6_experiments.txt 16).  So: as many workgroups per lane as still leave every workgroup a CU of its own, when that is
         * a cluster at all */
        const int32_t c1 = min(max(1, ud->g->dev->n_cu / 8) / lanes_per_xcd, cmax);
        if (c1 >= 2) C = c1;
    }
    /* (KF_MAXC: what the kernel's LDS arrays -- the waves' segments, the cluster's scan -- are sized for) */
    return max(1, min(min(C, KF_MAXC), usable / lanes_per_xcd));
}

/* n: lane slots of the launch (lanes 0 .. n - 1, or -- J.resume -- that many places of the relay's list); C: workgroups per lane */
template <int NE, bool EXACT, bool HEUR = false>
static int32_t
kf_launch_t(s3a_uttdec_t *ud, int32_t n, const KfJob &J, int32_t C)
{
    auto kern = ku_frames<NE, EXACT, HEUR>;
    const int32_t lanes_per_xcd = (n + 7) / 8;
    if (!J.resume) ud->kf_last_c = C;
    const int32_t grid = C == 1 ? n : 8 * C * lanes_per_xcd;
    if (C > 1) HIPCHK(hipMemsetAsync(ud->d_kfbar, 0, (size_t)ud->n_lanes * 8, ud->stream));
    /* the launch's shared arguments: a slot of a small ring (pinned + device) so that launches may queue up behind one another */
    if (ud->kf_arg_at > 0 && ud->kf_arg_at % KF_ARG_SLOTS == 0) HIPCHK(hipStreamSynchronize(ud->stream));
    KfArgs *ha = (KfArgs *)ud->h_kfargs + ud->kf_arg_at % KF_ARG_SLOTS, *da = (KfArgs *)ud->d_kfargs + ud->kf_arg_at % KF_ARG_SLOTS;
    ud->kf_arg_at++;
    ha->S = ud->S; ha->lm = ud->lm->d; ha->dict = ud->dict; ha->par = ud->par; ha->J = J;
    HIPCHK(hipMemcpyAsync(da, ha, sizeof(KfArgs), hipMemcpyHostToDevice, ud->stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(KF_NT), 0, ud->stream, ud->d_lanes, (const KfArgs *)da, n, C,
                       ud->d_kfbar, (ud->weak_possible ? 1 : 0) | (ud->big_wl ? 2 : 0), ud->persist < 3 ? 1 : 0);
    HIPCHK(hipGetLastError());
    return S3A_OK;
}

static int32_t
kf_launch_c(s3a_uttdec_t *ud, int32_t n, const KfJob &J, int32_t C)
{
    if (ud->S.ne == 5) {
        if (ud->S.pheurtype > 0) return ud->exact ? kf_launch_t<5, true, true>(ud, n, J, C) : kf_launch_t<5, false, true>(ud, n, J, C);
        return ud->exact ? kf_launch_t<5, true>(ud, n, J, C) : kf_launch_t<5, false>(ud, n, J, C);
    }
    if (ud->S.pheurtype > 0) return ud->exact ? kf_launch_t<3, true, true>(ud, n, J, C) : kf_launch_t<3, false, true>(ud, n, J, C);
    return ud->exact ? kf_launch_t<3, true>(ud, n, J, C) : kf_launch_t<3, false>(ud, n, J, C);
}

/* one launch, the cluster size the library's (KF_WINDOW blocks; a call without the relay) */
static int32_t
kf_launch(s3a_uttdec_t *ud, int32_t n, const KfJob &J)
{
    return kf_launch_c(ud, n, J, kf_choose_c(ud, n));
}

/* THE RELAY: a KF_QUEUE / KF_STATIC call as a chain of launches (KfJob.stop_at).  The first launch is what kf_launch would have made; when
 * nothing is left to take and no more lanes are at work than fit the chip as clusters of 2 (then 4) workgroups, they hand themselves over
 * at their next frame boundary and the next launch -- enqueued here, up front, behind the first -- continues them that way.  A launch of
 * the chain that is handed nothing ends at once.  Only an engine that may size clusters for the whole device plans a chain (kf_alone; not
 * with s3a_uttdec_opts_t.cluster set: the caller's cluster size holds for the whole call); s3a_variants_t.kf_no_relay: never. */
static int32_t
kf_launch_chain(s3a_uttdec_t *ud, int32_t n, KfJob J)
{
    /* (kf_relay_at, the tests' switch: the chain from one workgroup per lane on, however few the lanes) */
    const int32_t C0 = s3a_variants()->kf_relay_at > 0 && ud->kf_cluster_opt == 0 ? 1 : kf_choose_c(ud, n), usable = kf_usable_per_xcd(ud), W = KF_ST_WORDS + ud->n_lanes;
    int32_t st_c[KF_STAGES], st_cap[KF_STAGES], n_st = 0;
    /* (kf_relay_at, the tests' switch: the chain whatever else runs on the device -- the caller answers for co-residency, as with .cluster) */
    if (ud->d_kfrelay && ud->kf_cluster_opt == 0 && (kf_alone(ud) || s3a_variants()->kf_relay_at > 0) && !s3a_variants()->kf_no_relay) {
        int32_t prev_n = n, prev_c = C0;
        for (int32_t c = 2; c <= 4 && n_st < KF_STAGES; c *= 2) {
            int32_t cap = 8 * (usable / c);
            if (s3a_variants()->kf_relay_at > 0) cap = min(cap, max(1, s3a_variants()->kf_relay_at / (c / 2)));
            if (c <= prev_c || (long long)cap * 5 > (long long)prev_n * 4) continue;      /* (fewer lanes than 0.8 of the launch before) */
            st_c[n_st] = c; st_cap[n_st] = cap; n_st++;
            prev_n = cap; prev_c = c;
        }
    }
    if (n_st == 0) return kf_launch_c(ud, n, J, C0);
    HIPCHK(hipMemsetAsync(ud->d_kfrelay, 0, ((size_t)(KF_STAGES + 1) * W + (size_t)3 * ud->n_lanes) * 4, ud->stream));
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)ud->d_kfrelay, n, 1, ud->stream));           /* (the first launch's lanes at work) */
    J.lane_f = ud->d_kfrelay + (size_t)(KF_STAGES + 1) * W; J.lane_uq = J.lane_f + ud->n_lanes; J.lane_stop = J.lane_uq + ud->n_lanes;
    int32_t rc = S3A_OK;
    for (int32_t k = 0; k <= n_st && rc == S3A_OK; k++) {
        J.resume = k > 0; J.stop_at = k < n_st ? st_cap[k] : 0;
        J.st_cur = ud->d_kfrelay + (size_t)k * W; J.st_next = ud->d_kfrelay + (size_t)(k + 1) * W;
        rc = kf_launch_c(ud, k == 0 ? n : st_cap[k - 1], J, k == 0 ? C0 : st_c[k - 1]);
    }
    ud->kf_n_relay = n_st;
    return rc;
}

/* the frames [fg0, fg0 + nf) of lanes 0 .. n - 1 from their window rows: fg0 is a window boundary, nf at most the window */
static int32_t
enqueue_block(s3a_uttdec_t *ud, int32_t n, int32_t fg0, int32_t nf)
{
    int32_t rc = uw_launch(ud, n, fg0, false);
    if (rc != S3A_OK) return rc;
    KfJob J;
    memset(&J, 0, sizeof J);
    J.mode = KF_WINDOW; J.fg0 = fg0; J.n_fr = nf;
    ud->kf_n_score++; ud->kf_n_frames++;            /* (s3a_uttdec_last_parts: a block is a scoring pass + a launch) */
    return kf_launch(ud, n, J);
}

/* a time mark on the engine's stream (pairs of them bracket the scoring launches and ku_frames: kf_collect) */
static void
kf_mark(s3a_uttdec_t *ud)
{
    if ((size_t)ud->kf_ev_n >= ud->kf_evs.size()) {
        hipEvent_t e = NULL;
        if (hipEventCreate(&e) != hipSuccess) return;
        ud->kf_evs.push_back(e);
    }
    (void)hipEventRecord(ud->kf_evs[ud->kf_ev_n++], ud->stream);
}

/* behind the stream's synchronisation: marks come in triples (before scoring, between, behind ku_frames) */
static void
kf_collect(s3a_uttdec_t *ud)
{
    ud->kf_score_ms = ud->kf_frames_ms = 0.0;
    for (int32_t i = 0; i + 2 < ud->kf_ev_n; i += 3) {
        float a = 0.0f, b = 0.0f;
        if (hipEventElapsedTime(&a, ud->kf_evs[i], ud->kf_evs[i + 1]) == hipSuccess) ud->kf_score_ms += a;
        if (hipEventElapsedTime(&b, ud->kf_evs[i + 1], ud->kf_evs[i + 2]) == hipSuccess) ud->kf_frames_ms += b;
    }
    ud->kf_ev_n = 0;
}

/* ---- SCORES FIRST: every frame of the call's utterances scored into one buffer before the search starts ---- */
/* rows of senone scores the device can hold beside everything else (5 bytes per senone and frame; half of what is free now) */
static size_t
sb_budget_rows(const s3a_uttdec_t *ud)
{
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return 0;
    const size_t per_row = (size_t)ud->S.n_sen * 5, rows = (fr / 2 + ud->sb_rows_cap * per_row) / per_row;
    return ud->sb_rows_max > 0 ? min(rows, ud->sb_rows_max) : rows;
}

static int32_t
sb_reserve(s3a_uttdec_t *ud, size_t rows, size_t groups, size_t n_utt)
{
    const size_t S_ = (size_t)ud->S.n_sen;
    if (rows > ud->sb_rows_cap) {
        if (ud->sb_scores) (void)hipFree(ud->sb_scores);
        if (ud->sb_bests) (void)hipFree(ud->sb_bests);
        ud->sb_scores = NULL; ud->sb_bests = NULL; ud->sb_rows_cap = 0;
        const size_t grow = rows + rows / 16 + 64;
        if (hipMalloc((void **)&ud->sb_scores, grow * S_ * 4) != hipSuccess || hipMalloc((void **)&ud->sb_bests, grow * S_) != hipSuccess) {
            if (ud->sb_scores) (void)hipFree(ud->sb_scores);
            ud->sb_scores = NULL; ud->sb_bests = NULL;
            s3a_set_error("s3a_uttdec: the score buffer (%zu rows of %zu senones) does not fit the device", grow, S_);
            return S3A_ENOMEM;
        }
        ud->sb_rows_cap = grow;
    }
    int32_t rc;
    if ((rc = q_grow(&ud->sb_gdesc_d, &ud->sb_gdesc_h, &ud->sb_g_cap, groups, "score groups")) != S3A_OK) return rc;
    if ((rc = q_grow(&ud->sb_row0_d, &ud->sb_row0_h, &ud->sb_row0_cap, n_utt + 1, "score rows")) != S3A_OK) return rc;
    return S3A_OK;
}

/* the groups of utterances [u0, u1): 8 consecutive frames each, rows from `row` on; returns the groups written */
static size_t
sb_describe(s3a_uttdec_t *ud, const float *const *feat_dev, const int32_t *n_frames, int32_t u0, int32_t u1, size_t g_at)
{
    const size_t S_ = (size_t)ud->S.n_sen;
    const int32_t DP = ud->S.D4 * 4;
    size_t row = 0, g = g_at;
    for (int32_t u = u0; u < u1; u++) {
        ud->sb_row0_h[u] = (long long)row;
        for (int32_t j0 = 0; j0 < n_frames[u]; j0 += UW_FB, g++) {
            UwGroup &gr = ud->sb_gdesc_h[g];
            gr.feat = feat_dev[u] + (size_t)j0 * DP;
            gr.win = ud->sb_scores + (row + j0) * S_;
            gr.winb = ud->sb_bests + (row + j0) * S_;
            gr.nv = min(UW_FB, n_frames[u] - j0); gr.pad = 0;
        }
        row += (size_t)n_frames[u];
    }
    return g - g_at;
}

/* score the groups [g0, g0 + n_g) of the uploaded table: launches of up to SB_PIECE groups */
#define SB_PIECE 1024
static int32_t
sb_score(s3a_uttdec_t *ud, size_t g0, size_t n_g)
{
    for (size_t g = 0; g < n_g; g += SB_PIECE) {
        const int32_t rc = uw_launch(ud, 0, 0, false, ud->sb_gdesc_d + g0 + g, (int32_t)min((size_t)SB_PIECE, n_g - g));
        if (rc != S3A_OK) return rc;
        ud->kf_n_score++;
    }
    return S3A_OK;
}

/* s3a_uttdec_decode through ku_frames: lane z decodes utterance z, all of its frames in one launch.  feat_dev[z]: the lane's
 * features on the device.  Returns S3A_EUNSUP when the scores do not fit (the caller then decodes in window blocks). */
static int32_t
kf_decode_static(s3a_uttdec_t *ud, int32_t n_utt, const int32_t *n_frames)
{
    size_t rows = 0, groups = 0;
    std::vector<const float *> fd((size_t)n_utt);
    for (int32_t z = 0; z < n_utt; z++) { rows += (size_t)n_frames[z]; groups += (size_t)(n_frames[z] + UW_FB - 1) / UW_FB; fd[z] = ud->h_ctx_up[z].feat; }
    if (rows > sb_budget_rows(ud)) return S3A_EUNSUP;
    int32_t rc;
    if ((rc = sb_reserve(ud, rows, groups, (size_t)n_utt)) != S3A_OK) return rc;
    const size_t ng = sb_describe(ud, fd.data(), n_frames, 0, n_utt, 0);
    HIPCHK(hipMemcpyAsync(ud->sb_gdesc_d, ud->sb_gdesc_h, ng * sizeof(UwGroup), hipMemcpyHostToDevice, ud->stream));
    HIPCHK(hipMemcpyAsync(ud->sb_row0_d, ud->sb_row0_h, (size_t)n_utt * 8, hipMemcpyHostToDevice, ud->stream));
    ud->kf_n_score = ud->kf_n_frames = 0;
    kf_mark(ud);
    if ((rc = sb_score(ud, 0, ng)) != S3A_OK) return rc;
    kf_mark(ud);
    KfJob J;
    memset(&J, 0, sizeof J);
    J.mode = KF_STATIC; J.row0 = ud->sb_row0_d; J.scores = ud->sb_scores; J.bests = ud->sb_bests;
    rc = kf_launch_chain(ud, n_utt, J);
    ud->kf_n_frames++;
    kf_mark(ud);
    return rc;
}

/* s3a_uttdec_decode_queue through ku_frames: the lanes take the utterances themselves.  The queue goes through in as many parts as
 * the score buffer needs (one, unless the queue's frames outgrow half the free device memory). */
static int32_t
kf_decode_queue(s3a_uttdec_t *ud, int32_t n_utt, const float *const *feat_dev, const int32_t *n_frames, const UBegin &B, const UHypPar &P,
                int32_t *wcount)
{
    const size_t budget = sb_budget_rows(ud);
    /* the parts: consecutive utterances whose rows fit */
    std::vector<int32_t> cut(1, 0);
    size_t rows = 0, max_rows = 0, groups = 0;
    for (int32_t u = 0; u < n_utt; u++) {
        if (rows > 0 && rows + (size_t)n_frames[u] > budget) { cut.push_back(u); rows = 0; }
        rows += (size_t)n_frames[u];
        max_rows = max(max_rows, rows);
        groups += (size_t)(n_frames[u] + UW_FB - 1) / UW_FB;
    }
    cut.push_back(n_utt);
    if (max_rows > budget) { s3a_set_error("s3a_uttdec_decode_queue: an utterance's scores (%zu rows) do not fit the device", max_rows); return S3A_ENOMEM; }
    int32_t rc;
    if ((rc = sb_reserve(ud, max_rows, groups, (size_t)n_utt)) != S3A_OK) return rc;
    std::vector<size_t> g_at(cut.size(), 0);
    for (size_t k = 0; k + 1 < cut.size(); k++) g_at[k + 1] = g_at[k] + sb_describe(ud, feat_dev, n_frames, cut[k], cut[k + 1], g_at[k]);
    HIPCHK(hipMemcpyAsync(ud->sb_gdesc_d, ud->sb_gdesc_h, g_at.back() * sizeof(UwGroup), hipMemcpyHostToDevice, ud->stream));
    HIPCHK(hipMemcpyAsync(ud->sb_row0_d, ud->sb_row0_h, (size_t)n_utt * 8, hipMemcpyHostToDevice, ud->stream));
    /* the order of the takes: within a part the longest utterance first (stable: equal lengths in queue order) -- the launch ends with its
     * slowest lane, and an utterance taken when the counter is nearly through should be a short one */
    if ((rc = q_grow(&ud->kf_order_d, &ud->kf_order_h, &ud->kf_order_cap, (size_t)n_utt, "queue order")) != S3A_OK) return rc;
    for (size_t k = 0; k + 1 < cut.size(); k++) {
        for (int32_t u = cut[k]; u < cut[k + 1]; u++) ud->kf_order_h[u] = u;
        if (!s3a_variants()->kf_queue_in_order)
            std::stable_sort(ud->kf_order_h + cut[k], ud->kf_order_h + cut[k + 1], [&](int32_t a, int32_t b) { return n_frames[a] > n_frames[b]; });
    }
    HIPCHK(hipMemcpyAsync(ud->kf_order_d, ud->kf_order_h, (size_t)n_utt * 4, hipMemcpyHostToDevice, ud->stream));
    ud->kf_n_score = ud->kf_n_frames = 0;
    for (size_t k = 0; k + 1 < cut.size(); k++) {
        const int32_t u0 = cut[k], nu = cut[k + 1] - cut[k];
        kf_mark(ud);
        if ((rc = sb_score(ud, g_at[k], g_at[k + 1] - g_at[k])) != S3A_OK) return rc;
        kf_mark(ud);
        HIPCHK(hipMemsetAsync(ud->d_kfnext, 0, 4, ud->stream));
        KfJob J;
        memset(&J, 0, sizeof J);
        J.mode = KF_QUEUE; J.n_utt = nu; J.u0 = u0; J.order = ud->kf_order_d; J.next = ud->d_kfnext; J.lane_u = ud->d_kfnext + 16; J.stage = ud->q_ctx_d;
        J.row0 = ud->sb_row0_d; J.scores = ud->sb_scores; J.bests = ud->sb_bests;
        J.hdr = ud->q_hdr_d; J.words = ud->q_words_d; J.wcount = wcount; J.P = P; J.B = B; J.n_word = ud->cfg.n_word;
        if ((rc = kf_launch_chain(ud, min(ud->n_lanes, nu), J)) != S3A_OK) return rc;
        ud->kf_n_frames++;
        kf_mark(ud);
    }
    return S3A_OK;
}

/* ---- graph mode: the launches of a BLOCK of frames as one HIP graph ----
 * The launch sequence of a frame is the same for every frame (fixed grids; what a kernel has to do it finds in memory), and
 * the frame number reaches the kernels as argument + *S.fgbase: a block of G frames (G = the look-ahead window, so that the
 * block holds exactly one scoring pass) is captured once per lane count with the arguments 0 .. G - 1, its last node moves
 * the counter on by G, and a decode is ceil(frames / G) graph launches (the frames behind an utterance's end find nothing to
 * do).  The host's part of a frame drops from ~13 launches to 1 / G of a graph launch. */
static int32_t
graph_block_frames(const s3a_uttdec_t *ud)
{
    return ud->S.win_K > 0 ? ud->S.win_K : 8;
}

static int32_t
graph_for(s3a_uttdec_t *ud, int32_t n, hipGraphExec_t *out)
{
    const int32_t G = graph_block_frames(ud);
    for (auto &g : ud->graphs)
        if (g.n == n && g.frames == G && memcmp(&g.S, &ud->S, sizeof(UShared)) == 0) { *out = g.exec; return S3A_OK; }
    /* (function attributes must not be set while the stream captures: the scoring pass's dynamic LDS, once, up front) */
    if (ud->S.win_K > 0) {
        const int32_t rc0 = uw_launch(ud, n, 0, true);
        if (rc0 != S3A_OK) return rc0;
    }
    hipGraph_t graph = NULL;
    hipGraphExec_t exec = NULL;
    if (hipStreamBeginCapture(ud->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { s3a_set_error("s3a_uttdec: hipStreamBeginCapture failed: %s", hipGetErrorString(hipGetLastError())); return S3A_EHIP; }
    int32_t rc = S3A_OK;
    for (int32_t j = 0; j < G && rc == S3A_OK; j++) rc = enqueue_frame(ud, n, j, false);
    if (rc == S3A_OK) hipLaunchKernelGGL(ku_advance, dim3(1), dim3(64), 0, ud->stream, ud->d_fgbase, G);
    if (hipStreamEndCapture(ud->stream, &graph) != hipSuccess || !graph) { s3a_set_error("s3a_uttdec: hipStreamEndCapture failed: %s", hipGetErrorString(hipGetLastError())); return S3A_EHIP; }
    if (rc != S3A_OK) { (void)hipGraphDestroy(graph); return rc; }
    if (hipGraphInstantiate(&exec, graph, NULL, NULL, 0) != hipSuccess || !exec) { (void)hipGraphDestroy(graph); s3a_set_error("s3a_uttdec: hipGraphInstantiate failed: %s", hipGetErrorString(hipGetLastError())); return S3A_EHIP; }
    (void)hipGraphDestroy(graph);
    s3a_uttdec_s::FrameGraph fgr;
    fgr.n = n; fgr.frames = G; fgr.exec = exec; fgr.S = ud->S;
    ud->graphs.push_back(fgr);
    *out = exec;
    return S3A_OK;
}

/* download lane z's history table + frame statistics (after the frames have been enqueued) */
static int32_t
lane_fetch_fstat(s3a_uttdec_t *ud, int32_t z)
{
    HostLane &hl = ud->lane[z];
    HIPCHK(hipMemcpyAsync(hl.h_fstat, hl.d.w.fstat, (size_t)hl.nfr * 8 * 4, hipMemcpyDeviceToHost, ud->stream));
    return S3A_OK;
}

static int32_t
lane_fetch_table(s3a_uttdec_t *ud, int32_t z)
{
    HostLane &hl = ud->lane[z];
    const int32_t n = hl.h_st[0], nf = hl.nfr + 2;
    hl.n_entry = n;
    const size_t need = (size_t)10 * (n + 1) + (size_t)3 * nf;
    if (need > hl.h_tab_cap) {
        const size_t grow = max(need * 2, (size_t)1 << 18);        /* (with slack: see lane_begin) */
        if (hl.h_tab) (void)hipHostFree(hl.h_tab);
        hl.h_tab = NULL; hl.h_tab_cap = 0;
        if (hipHostMalloc((void **)&hl.h_tab, grow * 4 + 64) != hipSuccess) { s3a_set_error("s3a_uttdec: pinned table buffer"); return S3A_ENOMEM; }
        hl.h_tab_cap = grow;
    }
    const WLane &w = hl.d.w;
    const int32_t *src[10] = { w.score, w.pred, w.lw0, w.lw1, w.wid, w.sf, w.ef, w.ascr, w.lscr, w.type };
    for (int k = 0; k < 10; k++)
        HIPCHK(hipMemcpyAsync(hl.h_tab + (size_t)k * (n + 1), src[k], (size_t)n * 4, hipMemcpyDeviceToHost, ud->stream));
    int32_t *tail = hl.h_tab + (size_t)10 * (n + 1);
    HIPCHK(hipMemcpyAsync(tail, w.frame_start, (size_t)nf * 4, hipMemcpyDeviceToHost, ud->stream));
    HIPCHK(hipMemcpyAsync(tail + nf, w.bestscore, (size_t)nf * 4, hipMemcpyDeviceToHost, ud->stream));
    HIPCHK(hipMemcpyAsync(tail + 2 * nf, w.bestvh, (size_t)nf * 4, hipMemcpyDeviceToHost, ud->stream));
    return S3A_OK;
}

static int32_t
uttdec_decode(s3a_uttdec_t *ud, int32_t n_utt, const float *const *feat, const int32_t *n_frames,
              int32_t feat_stride, bool feat_on_device)
{
    if (!ud || n_utt <= 0 || n_utt > ud->n_lanes || !feat || !n_frames || feat_stride < ud->veclen) {
        s3a_set_error("s3a_uttdec_decode: bad arguments (%d utterances, %d lanes)", n_utt, ud ? ud->n_lanes : 0);
        return S3A_EINVAL;
    }
    HIPCHK(hipSetDevice(ud->device));
    int32_t rc, maxT = 0;
    ud->q_n = 0;
    ud->kf_n_score = ud->kf_n_frames = ud->kf_last_c = ud->kf_n_relay = 0;         /* (s3a_uttdec_last_parts: all zero unless THIS call goes through ku_frames) */
    double tm[6] = { 0, 0, 0, 0, 0, 0 };
    auto now = []() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
    tm[0] = now();
    for (int32_t z = 0; z < n_utt; z++) {
        if ((rc = lane_begin(ud, z, feat[z], n_frames[z], feat_stride, feat_on_device)) != S3A_OK) return rc;
        maxT = max(maxT, n_frames[z]);
    }
    {
        const s3a_wordlevel_cfg_t &c = ud->cfg;
        UBegin B;
        const int32_t e0[10] = { 0 /*score*/, -1 /*pred*/, c.start_lwid, -1 /*lw1*/, c.startwid, -1 /*sf*/, -1 /*ef*/, 0, 0, 0 };
        memcpy(B.e0, e0, sizeof e0);
        /* wl_lm_context of entry 0: state (<s>, none): no trigram run, the bigrams of <s> */
        B.lmc[0] = B.lmc[1] = B.lmc[2] = B.lmc[3] = 0; B.lmc[4] = -1;
        if (ud->lm->d.n_bg > 0 && c.start_lwid >= 0) { B.lmc[3] = ud->lm->ug_firstbg[c.start_lwid]; B.lmc[4] = ud->lm->ug_firstbg[c.start_lwid + 1] - B.lmc[3]; }
        B.n_pset = ud->n_pset > 0 ? ud->n_pset : 1;
        hipLaunchKernelGGL(ku_lanes_end, dim3(8, 2 * ud->S.T, n_utt), dim3(256), 0, ud->stream, ud->d_lanes, ud->S, (const int32_t *)NULL, 0);
        hipLaunchKernelGGL(ku_lanes_begin, dim3(32, 1, n_utt), dim3(256), 0, ud->stream, ud->d_lanes, ud->S, B, (const int32_t *)NULL, (const int32_t *)NULL, (const UCtx *)NULL);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(ud->S.ctx_all, ud->h_ctx_up, sizeof(UCtx) * n_utt, hipMemcpyHostToDevice, ud->stream));
        if (ud->S.pheurtype > 0) {
            /* the phoneme look-ahead's tables for the whole utterance: every frame's CI senone scores, every frame's phn_heur_list */
            const dim3 g((ud->S.n_ci_sen * ud->S.CP + 255) / 256, maxT, n_utt);
            if (ud->exact) hipLaunchKernelGGL(ku_ci_ahead<true>, g, dim3(256), 0, ud->stream, ud->d_lanes, ud->S, (const int32_t *)NULL);
            else hipLaunchKernelGGL(ku_ci_ahead<false>, g, dim3(256), 0, ud->stream, ud->d_lanes, ud->S, (const int32_t *)NULL);
            hipLaunchKernelGGL(ku_phn_heur, dim3((maxT + PH_T - 1) / PH_T, 1, n_utt), dim3(PH_T), 0, ud->stream, ud->d_lanes, ud->S, (const int32_t *)NULL);
            HIPCHK(hipGetLastError());
        }
    }
    /* (the engine's two timing events live as long as the engine; the per-launch events of profiled frames are destroyed
     * on every way out) */
    if (!ud->ev0 && (hipEventCreate(&ud->ev0) != hipSuccess || hipEventCreate(&ud->ev1) != hipSuccess)) {
        s3a_set_error("s3a_uttdec_decode: hipEventCreate failed");
        return S3A_EHIP;
    }
    rc = S3A_OK;
    tm[1] = now();
    if (hipEventRecord(ud->ev0, ud->stream) != hipSuccess) rc = S3A_EHIP;
    if (ud->use_graph && ud->prof_every == 0) {
        hipGraphExec_t ge = NULL;
        const int32_t G = graph_block_frames(ud);
        rc = graph_for(ud, n_utt, &ge);
        if (rc == S3A_OK && hipMemsetAsync(ud->d_fgbase, 0, 4, ud->stream) != hipSuccess) rc = S3A_EHIP;
        for (int32_t f = 0; f < maxT && rc == S3A_OK; f += G)
            if (hipGraphLaunch(ge, ud->stream) != hipSuccess) { s3a_set_error("s3a_uttdec_decode: hipGraphLaunch failed: %s", hipGetErrorString(hipGetLastError())); rc = S3A_EHIP; }
        if (rc == S3A_OK && hipMemsetAsync(ud->d_fgbase, 0, 4, ud->stream) != hipSuccess) rc = S3A_EHIP;
    }
    else if (kf_served(ud, n_utt)) {
        /* every frame scored first, then each lane's utterance as one launch; in window blocks when the scores do not fit */
        rc = kf_decode_static(ud, n_utt, n_frames);
        if (rc == S3A_EUNSUP) {
            rc = S3A_OK;
            for (int32_t f = 0; f < maxT && rc == S3A_OK; f += ud->S.win_K)
                rc = enqueue_block(ud, n_utt, f, min(ud->S.win_K, maxT - f));
        }
    }
    else
    for (int32_t f = 0; f < maxT && rc == S3A_OK; f++)
        rc = enqueue_frame(ud, n_utt, f, ud->prof_every > 0 && f % ud->prof_every == 0);
    if (rc == S3A_OK && hipEventRecord(ud->ev1, ud->stream) != hipSuccess) rc = S3A_EHIP;
    /* (the second pass -- vithist_utt_end + lattice + best path for every lane -- follows the first pass's hypotheses below) */
    tm[2] = now();
    /* every lane's hypothesis on the device (before the second pass appends its final entries to the tables); what
     * comes back: the lanes' contexts (error bits, counters), the hypothesis headers, then the words */
    if (rc == S3A_OK) {
        const s3a_wordlevel_cfg_t &c = ud->cfg;
        const UHypPar P = { c.finish_lwid, c.finishwid, c.silwid, ud->hyp_wcap, (int32_t)min((long long)ud->n_lanes * ud->hyp_wcap, (long long)INT_MAX) };
        if (hipMemsetAsync(ud->d_hyp_hdr + (size_t)n_utt * UH_N, 0, 4, ud->stream) != hipSuccess) rc = S3A_EHIP;
        hipLaunchKernelGGL(ku_hyp, dim3(n_utt), dim3(UH_T), 0, ud->stream, ud->d_lanes, ud->lm->d, ud->dict, P, ud->d_hyp_hdr, ud->d_hyp_words,
                           ud->d_hyp_hdr + (size_t)n_utt * UH_N, (const int32_t *)NULL, (const int32_t *)NULL);
        if (hipGetLastError() != hipSuccess) { s3a_set_error("s3a_uttdec_decode: ku_hyp launch failed"); rc = S3A_EHIP; }
    }
    if (rc == S3A_OK && ud->dag) rc = s3a_dagpass_enqueue(ud->dag, n_utt, ud->stream, 1);
    const size_t first_words = min((size_t)n_utt * UH_FIRST, (size_t)ud->n_lanes * ud->hyp_wcap);
    if (rc == S3A_OK && (hipMemcpyAsync(ud->h_ctx_dn, ud->S.ctx_all, sizeof(UCtx) * n_utt, hipMemcpyDeviceToHost, ud->stream) != hipSuccess
                         || hipMemcpyAsync(ud->h_hyp_hdr, ud->d_hyp_hdr, ((size_t)n_utt * UH_N + 1) * 4, hipMemcpyDeviceToHost, ud->stream) != hipSuccess
                         || hipMemcpyAsync(ud->h_hyp_words, ud->d_hyp_words, first_words * 24, hipMemcpyDeviceToHost, ud->stream) != hipSuccess)) rc = S3A_EHIP;
    if (hipStreamSynchronize(ud->stream) != hipSuccess && rc == S3A_OK) { s3a_set_error("s3a_uttdec_decode: %s", hipGetErrorString(hipGetLastError())); rc = S3A_EHIP; }
    if (rc == S3A_OK) {
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, ud->ev0, ud->ev1);
        ud->last_decode_ms = ms;
    }
    kf_collect(ud);
    for (auto &e : ud->prof_ev) {
        float ms = 0.0f;
        if (rc == S3A_OK && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { ud->prof_us[e.cls] += 1e3 * ms; ud->prof_n[e.cls]++; }
        (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
    }
    ud->prof_ev.clear();
    if (rc != S3A_OK) {
        for (int32_t z = 0; z < n_utt; z++) ud->lane[z].dirty = 1;      /* nothing is known about the lanes' state */
        return rc;
    }
    tm[3] = now();
    /* the history tables and the per-frame statistics stay on the device until someone asks (s3a_uttdec_result) */
    ud->tables_fetched = 0; ud->fstat_fetched = 0;
    for (int32_t z = 0; z < n_utt; z++) {
        const int32_t *h = ud->h_hyp_hdr + (size_t)z * UH_N;
        ud->lane[z].h_st[0] = h[UH_NENTRY]; ud->lane[z].h_st[1] = h[UH_NFRM];
    }
    {
        const size_t total_words = (size_t)ud->h_hyp_hdr[(size_t)n_utt * UH_N];
        if (total_words > first_words)
            HIPCHK(hipMemcpyAsync(ud->h_hyp_words + first_words * 6, ud->d_hyp_words + first_words * 6, (total_words - first_words) * 24,
                                  hipMemcpyDeviceToHost, ud->stream));
    }
    if (ud->dag && (rc = s3a_dagpass_finish(ud->dag, n_utt, ud->stream)) != S3A_OK) return rc;
    tm[4] = now();
    HIPCHK(hipStreamSynchronize(ud->stream));
    tm[5] = now();
    if (ud->times)
        fprintf(stderr, "s3a_uttdec_decode: %d lanes x %d frames: begin %.1f ms, enqueue %.1f, wait %.1f (device %.1f), hypothesis copies issued %.1f, done %.1f\n",
                n_utt, maxT, 1e3 * (tm[1] - tm[0]), 1e3 * (tm[2] - tm[1]), 1e3 * (tm[3] - tm[2]), ud->last_decode_ms, 1e3 * (tm[4] - tm[3]), 1e3 * (tm[5] - tm[4]));
    ud->n_utt = n_utt;
    /* EVERY lane that stopped in mid-frame starts its next utterance from scratch (a capacity overflow usually hits several
     * lanes of a batch); then the first one is reported */
    for (int32_t z = 0; z < n_utt; z++)
        if (ud->lane[z].h_ctx->err) ud->lane[z].dirty = 1;
    for (int32_t z = 0; z < n_utt; z++) {
        const int32_t e = ud->lane[z].h_ctx->err;
        if (e) {
            s3a_set_error("s3a_uttdec_decode: utterance %d stopped at frame %d: error bits 0x%x%s%s%s%s%s%s", z,
                          ud->lane[z].h_ctx->cf, e, (e & WL_E_OPEN_EXIT) ? " (out.history == -1 at a word exit)" : "",
                          (e & WL_E_EXITS) ? " (too many word exits)" : "", (e & WL_E_CAND) ? " (candidate buffer full)" : "",
                          (e & WL_E_TABLE) ? " (history table full)" : "", (e & WL_E_LC) ? " (phone is no left context)" : "",
                          (e & WL_E_NOLM) ? " (word without LM id)" : "");
            return (e & (WL_E_EXITS | WL_E_CAND | WL_E_TABLE | WL_E_CALLS)) ? S3A_ENOMEM : S3A_EINVAL;
        }
    }
    return S3A_OK;
}

extern "C" int32_t
s3a_uttdec_decode(s3a_uttdec_t *ud, int32_t n_utt, const float *const *feat, const int32_t *n_frames, int32_t feat_stride)
{
    return uttdec_decode(ud, n_utt, feat, n_frames, feat_stride, false);
}

/* the same with the features already in HBM: feat_dev[z] = n_frames[z] rows of feat_stride floats on the device,
 * feat_stride = the scorer's row length (feature length rounded up to a multiple of 4, padding zero) */
extern "C" int32_t
s3a_uttdec_decode_dev(s3a_uttdec_t *ud, int32_t n_utt, const float *const *feat_dev, const int32_t *n_frames, int32_t feat_stride)
{
    return uttdec_decode(ud, n_utt, feat_dev, n_frames, feat_stride, true);
}

/* ------------------------------------------------------------------ */
/* a QUEUE of utterances: lanes are refilled                           */
/* ------------------------------------------------------------------ */
/*
 * s3a_uttdec_decode decodes at most n_lanes utterances and the lanes of a call finish with the longest of them.  Here
 * the engine takes ANY number of utterances (ctl_process, libcommon/corpus.c:538, knows no coupling between them either):
 * the first n_lanes start at frame 0, and a lane whose utterance has ended takes the next one of the queue at the next
 * window boundary (with look-ahead scoring a lane's frames are scored K at a time, so utterances begin at multiples of K;
 * without it: at the next frame) -- every utterance's length is known before the first launch, so the whole schedule is
 * made on the host up front and, as in a plain decode, the host only enqueues: per frame the usual launches (a lane's own
 * frame is fg - ctx->f0), per refill event the finished lanes' hypotheses (ku_hyp into the UTTERANCE's header slot),
 * lextree_utt_end (ku_lanes_end; a lane whose utterance stopped on an error is scrubbed on the device), and
 * srch_TST_begin + the staged context of the utterance the lane takes (ku_lanes_begin).  What comes back: a header +
 * words per utterance (s3a_uttdec_queue_hyp); the history tables are reused by the lanes' next utterances and are not
 * available (no s3a_uttdec_result, no second pass).
 */
/* The refill schedule, host arithmetic alone (no device): utterance u of the queue runs in lane[u] from engine frame f0[u]
 * on.  The first n_lanes utterances start at frame 0; from then on, at every multiple of `boundary` frames, every lane whose
 * utterance has ended takes the queue's next one (lanes in index order).  Returns the engine frame at which the last lane is
 * done (a multiple of `boundary`), or S3A_EINVAL. */
extern "C" int32_t
s3a_queue_schedule(int32_t n_lanes, int32_t boundary, int32_t n_utt, const int32_t *n_frames, int32_t *lane, int32_t *f0)
{
    if (n_lanes <= 0 || boundary <= 0 || n_utt <= 0 || !n_frames || !lane || !f0) return S3A_EINVAL;
    const int32_t n = min(n_lanes, n_utt);
    std::vector<int32_t> busy_until((size_t)n, 0);
    std::vector<char> busy((size_t)n, 0);
    int32_t next = 0;
    for (int32_t u = 0; u < n_utt; u++) if (n_frames[u] <= 0) return S3A_EINVAL;
    for (long long F = 0;; F += boundary) {
        if (F > INT_MAX - boundary) return S3A_EINVAL;
        bool any = false;
        for (int32_t z = 0; z < n; z++) {
            if (busy[z] && busy_until[z] <= F) busy[z] = 0;
            if (!busy[z] && next < n_utt) {
                lane[next] = z; f0[next] = (int32_t)F;
                busy_until[z] = (int32_t)F + n_frames[next]; busy[z] = 1;
                next++;
            }
            any = any || busy[z];
        }
        if (!any) return (int32_t)F;
    }
}

/* s3a_uttdec_queue_keep_lattices: the lanes `lanes[0 .. n)` have just gone through the second pass for the utterances utts[]: wait for the
 * stream and read their lattices back, before the lanes' tables are reused (the price of lattices out of a queue: the host waits once per
 * group / refill event) */
static int32_t
q_keep_lattices(s3a_uttdec_t *ud, const int32_t *lanes, const int32_t *utts, int32_t n)
{
    HIPCHK(hipStreamSynchronize(ud->stream));
    for (int32_t i = 0; i < n; i++) {
        s3a_uttdec_s::QLat &q = ud->q_lat[(size_t)utts[i]];
        q.have = 0;
        if (s3a_dagpass_lattice_lane(ud->dag, lanes[i], &q.info, NULL, 0, NULL, 0) != S3A_OK) continue;      /* (no lattice: the pass's status says why) */
        q.nodes.resize((size_t)q.info.n_nodes + 1); q.links.resize((size_t)q.info.n_links + 1);
        if (s3a_dagpass_lattice_lane(ud->dag, lanes[i], &q.info, q.nodes.data(), q.info.n_nodes, q.links.data(), q.info.n_links) == S3A_OK) q.have = 1;
    }
    return S3A_OK;
}

static int32_t
uttdec_decode_queue(s3a_uttdec_t *ud, int32_t n_utt, const float *const *feat, const int32_t *n_frames, int32_t feat_stride,
                    bool feat_on_device)
{
    if (!ud || n_utt <= 0 || !feat || !n_frames || feat_stride < ud->veclen) {
        s3a_set_error("s3a_uttdec_decode_queue: bad arguments (%d utterances)", n_utt);
        return S3A_EINVAL;
    }
    HIPCHK(hipSetDevice(ud->device));
    ud->kf_n_score = ud->kf_n_frames = ud->kf_last_c = ud->kf_n_relay = 0;
    const UShared &S = ud->S;
    const bool graph_mode = ud->use_graph && ud->prof_every == 0;
    /* ku_frames (KF_QUEUE): the lanes take the utterances themselves -- no schedule, no refill events, no window boundaries */
    const bool kfq = !graph_mode && kf_served(ud, min(ud->n_lanes, n_utt)) && !ud->dag && S.pheurtype == 0;
    /* ... and with the second pass (round 6): the queue as GROUPS of at most n_lanes utterances, longest first -- a group is lane z = its
     * z-th utterance from the first frame to the last in ONE ku_frames launch (KF_STATIC, with the relay), then every lane's hypothesis,
     * vithist_utt_end + the second pass while the tables are still the utterances' own, and lextree_utt_end; everything enqueued, nothing
     * waited for between groups (utterances of like length share a group, so a group's lanes end together).  Before: the frame as launches
     * with refill events, or -- asked for -- window blocks, the regime that lost to the launches. */
    /* (-pheurtype goes the same way: the look-ahead's tables are a LANE's -- made when the lane begins its utterance, ku_ci_ahead / ku_phn_heur) */
    bool kfd = !graph_mode && kf_served(ud, min(ud->n_lanes, n_utt)) && (ud->dag != NULL || S.pheurtype > 0);
    std::vector<int32_t> kfd_order;
    if (kfd) {
        kfd_order.resize((size_t)n_utt);
        for (int32_t u = 0; u < n_utt; u++) kfd_order[u] = u;
        std::stable_sort(kfd_order.begin(), kfd_order.end(), [&](int32_t a, int32_t b) { return n_frames[a] > n_frames[b]; });
        size_t rows = 0, worst = 0;
        for (int32_t k = 0; k < n_utt; k++) {
            if (k % ud->n_lanes == 0) rows = 0;
            rows += (size_t)max(n_frames[kfd_order[k]], 0);
            worst = max(worst, rows);
        }
        if (worst > sb_budget_rows(ud)) kfd = false;            /* (a group's scores do not fit: the launches) */
    }
    /* utterances begin at window boundaries (the look-ahead pass scores K frames of all lanes); in graph mode at the blocks' */
    const int32_t n = min(ud->n_lanes, n_utt), E = graph_mode ? graph_block_frames(ud) : (S.win_K > 0 ? S.win_K : 1), D4x4 = S.D4 * 4, T = S.T;
    int32_t rc;
    ud->n_utt = 0; ud->q_n = 0;
    ud->q_lat.clear();
    if (ud->q_keep_lat && ud->dag) { ud->q_lat.resize((size_t)n_utt); for (auto &q : ud->q_lat) q.have = 0; }
    std::vector<size_t> row0((size_t)n_utt + 1, 0);
    for (int32_t u = 0; u < n_utt; u++) {
        if (n_frames[u] <= 0 || n_frames[u] > ud->max_frames) { s3a_set_error("s3a_uttdec_decode_queue: utterance %d has %d frames (1..%d)", u, n_frames[u], ud->max_frames); return S3A_EINVAL; }
        if (!feat[u]) { s3a_set_error("s3a_uttdec_decode_queue: utterance %d has no features", u); return S3A_EINVAL; }
        row0[u + 1] = row0[u] + (size_t)n_frames[u];
    }
    /* features: one buffer, rows padded to the scorer's float4 stride, one copy */
    std::vector<const float *> fd((size_t)n_utt);
    if (!feat_on_device) {
        if ((rc = q_grow(&ud->q_feat_d, &ud->q_feat_h, &ud->q_feat_cap, row0[n_utt] * D4x4, "features")) != S3A_OK) return rc;
        for (int32_t u = 0; u < n_utt; u++) {
            for (int32_t t = 0; t < n_frames[u]; t++) {
                float *row = ud->q_feat_h + (row0[u] + t) * D4x4;
                memcpy(row, feat[u] + (size_t)t * feat_stride, sizeof(float) * ud->veclen);
                for (int32_t k = ud->veclen; k < D4x4; k++) row[k] = 0.0f;
            }
            fd[u] = ud->q_feat_d + row0[u] * D4x4;
        }
        HIPCHK(hipMemcpyAsync(ud->q_feat_d, ud->q_feat_h, row0[n_utt] * D4x4 * 4, hipMemcpyHostToDevice, ud->stream));
    }
    else {
        if (feat_stride != D4x4) { s3a_set_error("s3a_uttdec_decode_queue_dev: device features must have rows of %d floats (zero padded)", D4x4); return S3A_EINVAL; }
        for (int32_t u = 0; u < n_utt; u++) fd[u] = feat[u];
    }
    for (int32_t z = 0; z < n; z++) {
        if ((rc = lane_scrub(ud, z)) != S3A_OK) return rc;
        HostLane &hl = ud->lane[z];
        hl.ls->cur = 0; hl.ls->last_nnxt = 0; hl.ls->hist_bound = 0; hl.ls->row_bound = 1;
        hl.ls->nnxt_t.assign(hl.ls->n_tree, 0);
        hl.sc->skip_count = 0;
        if (hl.epoch < 1) hl.epoch = 1;
    }
    /* the schedule (s3a_queue_schedule: host arithmetic on the lengths), then who ends and who begins at which boundary */
    struct Ev { int32_t f, o_el, o_bl, n_end, n_beg, max_nfr; };
    std::vector<Ev> evs;
    std::vector<int32_t> sched, u_lane((size_t)n_utt), u_f0((size_t)n_utt), last_utt((size_t)n, -1);
    if ((rc = q_grow(&ud->q_ctx_d, &ud->q_ctx_h, &ud->q_ctx_cap, (size_t)n_utt, "contexts")) != S3A_OK) return rc;
    if (kfq || kfd) {
        for (int32_t u = 0; u < n_utt; u++)
            if ((rc = utt_context(ud, ud->q_ctx_h[u], fd[u], n_frames[u], 1, 0, u)) != S3A_OK) return rc;
        for (int32_t z = 0; z < n; z++) ud->lane[z].nfr = 0;
        if (kfd) {              /* the groups' lane / utterance lists: [lanes 0 .. m - 1][their utterances] per group */
            for (int32_t k0 = 0; k0 < n_utt; k0 += ud->n_lanes) {
                const int32_t m = min(ud->n_lanes, n_utt - k0);
                for (int32_t z = 0; z < m; z++) sched.push_back(z);
                for (int32_t z = 0; z < m; z++) { sched.push_back(kfd_order[k0 + z]); last_utt[z] = kfd_order[k0 + z]; }
            }
        }
    }
    else {
        const int32_t fe = s3a_queue_schedule(n, E, n_utt, n_frames, u_lane.data(), u_f0.data());
        if (fe < 0) return fe;
        struct Lists { std::vector<int32_t> el, eu, bl, bu; int32_t mx = 0; };
        std::map<int32_t, Lists> at;
        for (int32_t u = 0; u < n_utt; u++) {           /* (queue order = the order of a lane's utterances) */
            const int32_t z = u_lane[u], F = u_f0[u], Fe = ((F + n_frames[u] + E - 1) / E) * E;
            HostLane &hl = ud->lane[z];
            if ((rc = utt_context(ud, ud->q_ctx_h[u], fd[u], n_frames[u], hl.epoch, F, u)) != S3A_OK) return rc;
            hl.epoch += n_frames[u] + 1;
            hl.nfr = n_frames[u];
            last_utt[z] = u;
            Lists &b = at[F];
            b.bl.push_back(z); b.bu.push_back(u); b.mx = max(b.mx, n_frames[u]);
            Lists &e = at[Fe];
            e.el.push_back(z); e.eu.push_back(u);
        }
        for (auto &kv : at) {
            const Lists &l = kv.second;
            Ev e = { kv.first, (int32_t)sched.size(), 0, (int32_t)l.el.size(), (int32_t)l.bl.size(), l.mx };
            sched.insert(sched.end(), l.el.begin(), l.el.end()); sched.insert(sched.end(), l.eu.begin(), l.eu.end());
            e.o_bl = (int32_t)sched.size();
            sched.insert(sched.end(), l.bl.begin(), l.bl.end()); sched.insert(sched.end(), l.bu.begin(), l.bu.end());
            evs.push_back(e);
        }
    }
    const int32_t F_end = evs.empty() ? 0 : evs.back().f;           /* (the last event only ends lanes) */
    if ((rc = q_grow(&ud->q_sched_d, &ud->q_sched_h, &ud->q_sched_cap, sched.size(), "schedule")) != S3A_OK) return rc;
    if (!sched.empty()) {
        memcpy(ud->q_sched_h, sched.data(), sched.size() * 4);
        HIPCHK(hipMemcpyAsync(ud->q_sched_d, ud->q_sched_h, sched.size() * 4, hipMemcpyHostToDevice, ud->stream));
    }
    HIPCHK(hipMemcpyAsync(ud->q_ctx_d, ud->q_ctx_h, sizeof(UCtx) * n_utt, hipMemcpyHostToDevice, ud->stream));
    /* results: a header per utterance + the word counter; the words packed (an utterance's hypothesis has at most one word
     * per frame + silence + </s>) */
    const size_t wtotal = min(row0[n_utt] + (size_t)4 * n_utt, (size_t)INT_MAX / 8);
    if ((rc = q_grow(&ud->q_hdr_d, &ud->q_hdr_h, &ud->q_hdr_cap, (size_t)n_utt * UH_N + 1, "hypothesis headers")) != S3A_OK) return rc;
    if ((rc = q_grow(&ud->q_words_d, (int32_t **)NULL, &ud->q_words_cap, wtotal * 6, "hypothesis words")) != S3A_OK) return rc;
    HIPCHK(hipMemsetAsync(ud->q_hdr_d, 0, ((size_t)n_utt * UH_N + 1) * 4, ud->stream));
    int32_t *wcount = ud->q_hdr_d + (size_t)n_utt * UH_N;
    ud->q_dag = 0;
    if (ud->dag) {
        /* the second pass of the lanes that end at a refill event runs right there, before the lane's table is reused */
        if ((rc = q_grow(&ud->q_dio_d, &ud->q_dio_h, &ud->q_dio_cap, (size_t)n_utt * DG_IO_N + 1, "second-pass status")) != S3A_OK) return rc;
        if ((rc = q_grow(&ud->q_dw_d, (int32_t **)NULL, &ud->q_dw_cap, wtotal * 6, "second-pass words")) != S3A_OK) return rc;
        HIPCHK(hipMemsetAsync(ud->q_dio_d, 0xff, ((size_t)n_utt * DG_IO_N) * 4, ud->stream));
        HIPCHK(hipMemsetAsync(ud->q_dio_d + (size_t)n_utt * DG_IO_N, 0, 4, ud->stream));
        if ((rc = s3a_dagpass_prepare(ud->dag, ud->stream)) != S3A_OK) return rc;
    }
    const s3a_wordlevel_cfg_t &c = ud->cfg;
    UBegin B;
    {
        const int32_t e0[10] = { 0 /*score*/, -1 /*pred*/, c.start_lwid, -1 /*lw1*/, c.startwid, -1 /*sf*/, -1 /*ef*/, 0, 0, 0 };
        memcpy(B.e0, e0, sizeof e0);
        B.lmc[0] = B.lmc[1] = B.lmc[2] = B.lmc[3] = 0; B.lmc[4] = -1;
        if (ud->lm->d.n_bg > 0 && c.start_lwid >= 0) { B.lmc[3] = ud->lm->ug_firstbg[c.start_lwid]; B.lmc[4] = ud->lm->ug_firstbg[c.start_lwid + 1] - B.lmc[3]; }
        B.n_pset = ud->n_pset > 0 ? ud->n_pset : 1;
    }
    const UHypPar P = { c.finish_lwid, c.finishwid, c.silwid, ud->hyp_wcap, (int32_t)wtotal };
    /* whatever the engine's last decode left active (it ends its lanes at the NEXT decode's beginning) */
    hipLaunchKernelGGL(ku_lanes_end, dim3(8, 2 * T, n), dim3(256), 0, ud->stream, ud->d_lanes, ud->S, (const int32_t *)NULL, 0);
    HIPCHK(hipGetLastError());
    auto run_event = [&](const Ev &e) -> int32_t {
        const int32_t *el = ud->q_sched_d + e.o_el, *eu = el + e.n_end, *bl = ud->q_sched_d + e.o_bl, *bu = bl + e.n_beg;
        if (e.n_end > 0) {
            hipLaunchKernelGGL(ku_hyp, dim3(e.n_end), dim3(UH_T), 0, ud->stream, ud->d_lanes, ud->lm->d, ud->dict, P, ud->q_hdr_d, ud->q_words_d, wcount, el, eu);
            if (ud->dag) {
                const int32_t drc = s3a_dagpass_enqueue_lanes(ud->dag, el, e.n_end, ud->stream);
                if (drc != S3A_OK) return drc;
                hipLaunchKernelGGL(ku_dag_store, dim3(e.n_end), dim3(UH_T), 0, ud->stream, ud->d_lanes, s3a_dagpass_dev_lanes(ud->dag), s3a_dagpass_hyp_cap(ud->dag),
                                   ud->q_dio_d, ud->q_dw_d, ud->q_dio_d + (size_t)n_utt * DG_IO_N, (int32_t)wtotal, el, eu);
                if (ud->q_keep_lat) {
                    const int32_t krc = q_keep_lattices(ud, sched.data() + e.o_el, sched.data() + e.o_el + e.n_end, e.n_end);
                    if (krc != S3A_OK) return krc;
                }
            }
            hipLaunchKernelGGL(ku_lanes_end, dim3(8, 2 * T, e.n_end), dim3(256), 0, ud->stream, ud->d_lanes, ud->S, el, c.n_word);
        }
        if (e.n_beg > 0) {
            hipLaunchKernelGGL(ku_lanes_begin, dim3(32, 1, e.n_beg), dim3(256), 0, ud->stream, ud->d_lanes, ud->S, B, bl, bu, (const UCtx *)ud->q_ctx_d);
            if (S.pheurtype > 0) {
                const dim3 g((S.n_ci_sen * S.CP + 255) / 256, e.max_nfr, e.n_beg);
                if (ud->exact) hipLaunchKernelGGL(ku_ci_ahead<true>, g, dim3(256), 0, ud->stream, ud->d_lanes, ud->S, bl);
                else hipLaunchKernelGGL(ku_ci_ahead<false>, g, dim3(256), 0, ud->stream, ud->d_lanes, ud->S, bl);
                hipLaunchKernelGGL(ku_phn_heur, dim3((e.max_nfr + PH_T - 1) / PH_T, 1, e.n_beg), dim3(PH_T), 0, ud->stream, ud->d_lanes, ud->S, bl);
            }
        }
        HIPCHK(hipGetLastError());
        return S3A_OK;
    };
    if (!ud->ev0 && (hipEventCreate(&ud->ev0) != hipSuccess || hipEventCreate(&ud->ev1) != hipSuccess)) {
        s3a_set_error("s3a_uttdec_decode_queue: hipEventCreate failed");
        return S3A_EHIP;
    }
    rc = S3A_OK;
    if (hipEventRecord(ud->ev0, ud->stream) != hipSuccess) rc = S3A_EHIP;
    size_t ei = 0;
    if (kfq) {
        if (rc == S3A_OK) rc = kf_decode_queue(ud, n_utt, fd.data(), n_frames, B, P, wcount);
    }
    else if (kfd) {
        /* every group's scoring groups and rows described up front (the pinned tables must not change under the copies in flight) */
        size_t groups = 0, max_rows = 0;
        std::vector<size_t> g_at(1, 0);
        std::vector<const float *> gf((size_t)n_utt);
        std::vector<int32_t> gn((size_t)n_utt);
        for (int32_t k = 0; k < n_utt; k++) { gf[k] = fd[kfd_order[k]]; gn[k] = n_frames[kfd_order[k]]; groups += (size_t)(gn[k] + UW_FB - 1) / UW_FB; }
        for (int32_t k0 = 0; k0 < n_utt; k0 += ud->n_lanes) {
            size_t rows = 0;
            for (int32_t k = k0; k < min(n_utt, k0 + ud->n_lanes); k++) rows += (size_t)gn[k];
            max_rows = max(max_rows, rows);
        }
        if (rc == S3A_OK) rc = sb_reserve(ud, max_rows, groups, (size_t)n_utt);
        for (int32_t k0 = 0; k0 < n_utt && rc == S3A_OK; k0 += ud->n_lanes)         /* (rows restart per group: sb_describe; row0 by position in the order) */
            g_at.push_back(g_at.back() + sb_describe(ud, gf.data(), gn.data(), k0, min(n_utt, k0 + ud->n_lanes), g_at.back()));
        if (rc == S3A_OK && (hipMemcpyAsync(ud->sb_gdesc_d, ud->sb_gdesc_h, g_at.back() * sizeof(UwGroup), hipMemcpyHostToDevice, ud->stream) != hipSuccess
                             || hipMemcpyAsync(ud->sb_row0_d, ud->sb_row0_h, (size_t)n_utt * 8, hipMemcpyHostToDevice, ud->stream) != hipSuccess)) rc = S3A_EHIP;
        ud->kf_n_score = ud->kf_n_frames = 0;
        size_t gi = 0;
        for (int32_t k0 = 0; k0 < n_utt && rc == S3A_OK; k0 += ud->n_lanes, gi++) {
            const int32_t m = min(ud->n_lanes, n_utt - k0);
            const int32_t *bl = ud->q_sched_d + (size_t)2 * k0, *bu = bl + m;
            hipLaunchKernelGGL(ku_lanes_begin, dim3(32, 1, m), dim3(256), 0, ud->stream, ud->d_lanes, ud->S, B, bl, bu, (const UCtx *)ud->q_ctx_d);
            if (S.pheurtype > 0) {      /* (the group's longest utterance is its first) */
                const dim3 g((S.n_ci_sen * S.CP + 255) / 256, gn[k0], m);
                if (ud->exact) hipLaunchKernelGGL(ku_ci_ahead<true>, g, dim3(256), 0, ud->stream, ud->d_lanes, ud->S, bl);
                else hipLaunchKernelGGL(ku_ci_ahead<false>, g, dim3(256), 0, ud->stream, ud->d_lanes, ud->S, bl);
                hipLaunchKernelGGL(ku_phn_heur, dim3((gn[k0] + PH_T - 1) / PH_T, 1, m), dim3(PH_T), 0, ud->stream, ud->d_lanes, ud->S, bl);
            }
            kf_mark(ud);
            if ((rc = sb_score(ud, g_at[gi], g_at[gi + 1] - g_at[gi])) != S3A_OK) break;
            kf_mark(ud);
            KfJob J;
            memset(&J, 0, sizeof J);
            J.mode = KF_STATIC; J.row0 = ud->sb_row0_d + k0; J.scores = ud->sb_scores; J.bests = ud->sb_bests;
            if ((rc = kf_launch_chain(ud, m, J)) != S3A_OK) break;
            ud->kf_n_frames++;
            kf_mark(ud);
            hipLaunchKernelGGL(ku_hyp, dim3(m), dim3(UH_T), 0, ud->stream, ud->d_lanes, ud->lm->d, ud->dict, P, ud->q_hdr_d, ud->q_words_d, wcount, bl, bu);
            if (ud->dag) {
                if ((rc = s3a_dagpass_enqueue_lanes(ud->dag, bl, m, ud->stream)) != S3A_OK) break;
                hipLaunchKernelGGL(ku_dag_store, dim3(m), dim3(UH_T), 0, ud->stream, ud->d_lanes, s3a_dagpass_dev_lanes(ud->dag), s3a_dagpass_hyp_cap(ud->dag),
                                   ud->q_dio_d, ud->q_dw_d, ud->q_dio_d + (size_t)n_utt * DG_IO_N, (int32_t)wtotal, bl, bu);
                if (ud->q_keep_lat && (rc = q_keep_lattices(ud, sched.data() + (size_t)2 * k0, sched.data() + (size_t)2 * k0 + m, m)) != S3A_OK) break;
            }
            hipLaunchKernelGGL(ku_lanes_end, dim3(8, 2 * T, m), dim3(256), 0, ud->stream, ud->d_lanes, ud->S, bl, c.n_word);
            if (hipGetLastError() != hipSuccess) { s3a_set_error("s3a_uttdec_decode_queue: a launch of the grouped second pass failed"); rc = S3A_EHIP; }
        }
    }
    else if (graph_mode) {
        hipGraphExec_t ge = NULL;
        rc = graph_for(ud, n, &ge);
        if (rc == S3A_OK && hipMemsetAsync(ud->d_fgbase, 0, 4, ud->stream) != hipSuccess) rc = S3A_EHIP;
        for (int32_t fg = 0; fg < F_end && rc == S3A_OK; fg += E) {        /* (E = the graph's block: refill events fall between blocks) */
            if (ei < evs.size() && evs[ei].f == fg) rc = run_event(evs[ei++]);
            if (rc == S3A_OK && hipGraphLaunch(ge, ud->stream) != hipSuccess) { s3a_set_error("s3a_uttdec_decode_queue: hipGraphLaunch failed: %s", hipGetErrorString(hipGetLastError())); rc = S3A_EHIP; }
        }
    }
    else if (kf_served(ud, n) && ud->persist > 1)           /* (a queue with the second pass: ku_frames only when asked for -- window blocks keep all lanes in step) */
        for (int32_t fg = 0; fg < F_end && rc == S3A_OK; fg += E) {        /* (E = the window: refill events fall on its boundaries) */
            if (ei < evs.size() && evs[ei].f == fg) rc = run_event(evs[ei++]);
            if (rc == S3A_OK) rc = enqueue_block(ud, n, fg, min(E, F_end - fg));
        }
    else
    for (int32_t fg = 0; fg < F_end && rc == S3A_OK; fg++) {
        if (ei < evs.size() && evs[ei].f == fg) rc = run_event(evs[ei++]);
        if (rc == S3A_OK) rc = enqueue_frame(ud, n, fg, ud->prof_every > 0 && fg % ud->prof_every == 0);
    }
    while (rc == S3A_OK && ei < evs.size()) rc = run_event(evs[ei++]);
    if (graph_mode && rc == S3A_OK && hipMemsetAsync(ud->d_fgbase, 0, 4, ud->stream) != hipSuccess) rc = S3A_EHIP;
    if (rc == S3A_OK && hipEventRecord(ud->ev1, ud->stream) != hipSuccess) rc = S3A_EHIP;
    if (rc == S3A_OK && hipMemcpyAsync(ud->q_hdr_h, ud->q_hdr_d, ((size_t)n_utt * UH_N + 1) * 4, hipMemcpyDeviceToHost, ud->stream) != hipSuccess) rc = S3A_EHIP;
    if (rc == S3A_OK && ud->dag && hipMemcpyAsync(ud->q_dio_h, ud->q_dio_d, ((size_t)n_utt * DG_IO_N + 1) * 4, hipMemcpyDeviceToHost, ud->stream) != hipSuccess) rc = S3A_EHIP;
    if (hipStreamSynchronize(ud->stream) != hipSuccess && rc == S3A_OK) { s3a_set_error("s3a_uttdec_decode_queue: %s", hipGetErrorString(hipGetLastError())); rc = S3A_EHIP; }
    if (rc == S3A_OK) {
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, ud->ev0, ud->ev1);
        ud->last_decode_ms = ms;
    }
    kf_collect(ud);
    for (auto &e : ud->prof_ev) {
        float ms = 0.0f;
        if (rc == S3A_OK && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { ud->prof_us[e.cls] += 1e3 * ms; ud->prof_n[e.cls]++; }
        (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
    }
    ud->prof_ev.clear();
    if (rc != S3A_OK) {
        for (int32_t z = 0; z < n; z++) ud->lane[z].dirty = 1;      /* nothing is known about the lanes' state */
        return rc;
    }
    {
        const size_t total_words = (size_t)ud->q_hdr_h[(size_t)n_utt * UH_N];
        if (total_words * 6 > ud->q_words_hcap) {
            if (ud->q_words_h) (void)hipHostFree(ud->q_words_h);
            ud->q_words_h = NULL; ud->q_words_hcap = 0;
            const size_t grow = total_words * 6 + total_words + 1024;
            if (hipHostMalloc((void **)&ud->q_words_h, grow * 4) != hipSuccess) { s3a_set_error("s3a_uttdec_decode_queue: pinned word buffer"); return S3A_ENOMEM; }
            ud->q_words_hcap = grow;
        }
        if (total_words > 0) HIPCHK(hipMemcpy(ud->q_words_h, ud->q_words_d, total_words * 24, hipMemcpyDeviceToHost));
    }
    if (ud->dag) {
        const size_t total_words = (size_t)ud->q_dio_h[(size_t)n_utt * DG_IO_N];
        if (total_words * 6 > ud->q_dw_hcap) {
            if (ud->q_dw_h) (void)hipHostFree(ud->q_dw_h);
            ud->q_dw_h = NULL; ud->q_dw_hcap = 0;
            const size_t grow = total_words * 6 + total_words + 1024;
            if (hipHostMalloc((void **)&ud->q_dw_h, grow * 4) != hipSuccess) { s3a_set_error("s3a_uttdec_decode_queue: pinned word buffer (second pass)"); return S3A_ENOMEM; }
            ud->q_dw_hcap = grow;
        }
        if (total_words > 0) HIPCHK(hipMemcpy(ud->q_dw_h, ud->q_dw_d, min(total_words, wtotal) * 24, hipMemcpyDeviceToHost));
        ud->q_dag = 1;
    }
    ud->q_n = n_utt;
    ud->q_nfr.assign(n_frames, n_frames + n_utt);
    /* a lane whose LAST utterance stopped in mid-frame starts the next decode from scratch (the others of its utterances
     * that stopped were scrubbed on the device when the lane took its next one) */
    for (int32_t z = 0; z < n; z++)
        if (last_utt[z] >= 0 && ud->q_hdr_h[(size_t)last_utt[z] * UH_N + UH_ERR]) ud->lane[z].dirty = 1;
    for (int32_t u = 0; u < n_utt; u++) {
        const int32_t *h = ud->q_hdr_h + (size_t)u * UH_N;
        const int32_t e = h[UH_ERR];
        if (e) {
            s3a_set_error("s3a_uttdec_decode_queue: utterance %d stopped at frame %d: error bits 0x%x%s%s%s%s%s%s", u, h[UH_CF], e,
                          (e & WL_E_OPEN_EXIT) ? " (out.history == -1 at a word exit)" : "",
                          (e & WL_E_EXITS) ? " (too many word exits)" : "", (e & WL_E_CAND) ? " (candidate buffer full)" : "",
                          (e & WL_E_TABLE) ? " (history table full)" : "", (e & WL_E_LC) ? " (phone is no left context)" : "",
                          (e & WL_E_NOLM) ? " (word without LM id)" : "");
            return (e & (WL_E_EXITS | WL_E_CAND | WL_E_TABLE | WL_E_CALLS)) ? S3A_ENOMEM : S3A_EINVAL;
        }
    }
    return S3A_OK;
}

extern "C" int32_t
s3a_uttdec_decode_queue(s3a_uttdec_t *ud, int32_t n_utt, const float *const *feat, const int32_t *n_frames, int32_t feat_stride)
{
    return uttdec_decode_queue(ud, n_utt, feat, n_frames, feat_stride, false);
}

extern "C" int32_t
s3a_uttdec_decode_queue_dev(s3a_uttdec_t *ud, int32_t n_utt, const float *const *feat_dev, const int32_t *n_frames, int32_t feat_stride)
{
    return uttdec_decode_queue(ud, n_utt, feat_dev, n_frames, feat_stride, true);
}

/* the first s3a_uttdec_result (or second-pass hypothesis) after a decode brings the per-frame statistics -- and the
 * history tables -- of ALL lanes of that decode to the host; decodes whose caller only wants hypotheses never move them */
static int32_t
uttdec_fetch(s3a_uttdec_t *ud, bool tables)
{
    HIPCHK(hipSetDevice(ud->device));
    int32_t rc;
    bool any = false;
    if (!ud->fstat_fetched) {
        for (int32_t z = 0; z < ud->n_utt; z++) if ((rc = lane_fetch_fstat(ud, z)) != S3A_OK) return rc;
        any = true;
    }
    if (tables && !ud->tables_fetched) {
        for (int32_t z = 0; z < ud->n_utt; z++) if ((rc = lane_fetch_table(ud, z)) != S3A_OK) return rc;
        any = true;
    }
    if (any) HIPCHK(hipStreamSynchronize(ud->stream));
    ud->fstat_fetched = 1;
    if (tables) ud->tables_fetched = 1;
    return S3A_OK;
}

extern "C" int32_t
s3a_uttdec_result(s3a_uttdec_t *ud, int32_t lane, s3a_utt_result_t *out)
{
    if (!ud || !out || lane < 0 || lane >= ud->n_utt) return S3A_EINVAL;
    if (ud->dag && !ud->keep_tables) { s3a_set_error("s3a_uttdec_result: the history tables were left on the device (s3a_uttdec_enable_bestpath, keep_tables = 0)"); return S3A_EUNSUP; }
    int32_t rc;
    if ((rc = uttdec_fetch(ud, true)) != S3A_OK) return rc;
    const HostLane &hl = ud->lane[lane];
    const int32_t n = hl.n_entry, nf = hl.nfr + 2;
    const int32_t *t = hl.h_tab;
    memset(out, 0, sizeof *out);
    out->err = hl.h_ctx->err; out->n_entry = n; out->n_frm = hl.h_st[1]; out->n_frames = hl.nfr;
    out->score = t; out->pred = t + (size_t)(n + 1); out->lw0 = t + (size_t)2 * (n + 1); out->lw1 = t + (size_t)3 * (n + 1);
    out->wid = t + (size_t)4 * (n + 1); out->sf = t + (size_t)5 * (n + 1); out->ef = t + (size_t)6 * (n + 1);
    out->ascr = t + (size_t)7 * (n + 1); out->lscr = t + (size_t)8 * (n + 1); out->type = t + (size_t)9 * (n + 1);
    const int32_t *tail = t + (size_t)10 * (n + 1);
    out->frame_start = tail; out->bestscore = tail + nf; out->bestvh = tail + 2 * nf;
    out->frame_stat = hl.h_fstat;
    out->max_cand = hl.h_ctx->max_cand; out->max_new = hl.h_ctx->max_new; out->n_tie_frames = hl.h_ctx->n_tie_frames;
    return S3A_OK;
}

extern "C" int32_t
s3a_uttdec_wl_ticks(s3a_uttdec_t *ud, int32_t lane, long long *out16)
{
    if (!ud || !out16 || lane < 0 || lane >= ud->n_utt) return S3A_EINVAL;
    for (int i = 0; i < 16; i++) out16[i] = ud->lane[lane].h_ctx->tacc[i];
    return S3A_OK;
}

/* ku_frames' clock: [0..11] lextree_enter test / rank / apply + marks, composite members, CI gate, composite maxima, HMM evaluation,
 * stamps (or histogram), weak HMMs, propagation, scan, emission + word level; [15] the emission alone when the lane is one
 * workgroup; [12] inside the launches, [13] frames, [14] launches -- of the utterance lane `lane` decoded last (read from the
 * device); out16[15] < 0 when the engine does not run ku_frames */
extern "C" int32_t
s3a_uttdec_frame_ticks(s3a_uttdec_t *ud, int32_t lane, long long *out16, int32_t *cluster)
{
    if (!ud || !out16 || lane < 0 || lane >= ud->n_lanes) return S3A_EINVAL;
    HIPCHK(hipSetDevice(ud->device));
    UCtx x;
    HIPCHK(hipMemcpy(&x, ud->S.ctx_all + lane, sizeof x, hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; i++) out16[i] = x.kacc[i];
    if (cluster) *cluster = ud->kf_last_c;
    return S3A_OK;
}

/* where the last decode's device time went when it ran through ku_frames: milliseconds and launches of the up-front scoring
 * (ku_score_window) and of ku_frames; cluster = workgroups per lane of its last launch.  All zero: the call ran the frame as launches. */
extern "C" int32_t
s3a_uttdec_last_relay(const s3a_uttdec_t *ud)
{
    return ud ? ud->kf_n_relay : 0;
}

extern "C" int32_t
s3a_uttdec_last_parts(s3a_uttdec_t *ud, double *score_ms, int32_t *n_score, double *frames_ms, int32_t *n_frames, int32_t *cluster)
{
    if (!ud) return S3A_EINVAL;
    if (score_ms) *score_ms = ud->kf_score_ms;
    if (frames_ms) *frames_ms = ud->kf_frames_ms;
    if (n_score) *n_score = ud->kf_n_score;
    if (n_frames) *n_frames = ud->kf_n_frames;
    if (cluster) *cluster = ud->kf_last_c;
    return S3A_OK;
}

/* diagnostics: lane's ku_frames launch that began at engine frame 512: clock (100 MHz) at entry and exit, HW_ID, XCC_ID */
extern "C" int32_t
s3a_uttdec_frame_dbg(s3a_uttdec_t *ud, int32_t lane, long long *out4)
{
    if (!ud || !out4 || lane < 0 || lane >= ud->n_lanes) return S3A_EINVAL;
    HIPCHK(hipSetDevice(ud->device));
    UCtx x;
    HIPCHK(hipMemcpy(&x, ud->S.ctx_all + lane, sizeof x, hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; i++) out4[i] = x.kdbg[i];
    return S3A_OK;
}

/* ---- phoneme look-ahead (-pheurtype 1..3) ---- */
extern "C" int32_t
s3a_uttdec_enable_pheur(s3a_uttdec_t *ud, int32_t pheurtype, int32_t pl_beam, int32_t pl_window, const uint8_t *const *node_ci,
                        const int16_t *sen2cimap, int32_t n_ci)
{
    if (!ud || pheurtype < 0 || pheurtype > 3) return S3A_EINVAL;
    HIPCHK(hipSetDevice(ud->device));
    if (pheurtype == 0) { ud->S.pheurtype = 0; return S3A_OK; }
    if (!node_ci || !sen2cimap || n_ci <= 0 || n_ci > PH_MAXCI || pl_window < 1) {
        s3a_set_error("s3a_uttdec_enable_pheur: bad arguments (%d CI phones, 1..%d; -pl_window %d >= 1)", n_ci, PH_MAXCI, pl_window);
        return S3A_EINVAL;
    }
    if (ud->S.n_sen <= ud->S.n_ci_sen) { s3a_set_error("s3a_uttdec_enable_pheur: the model has no CD senones"); return S3A_EUNSUP; }
    UShared &S = ud->S;
    if (!S.node_ci) {
        std::vector<uint8_t> h((size_t)S.N);
        for (int32_t t = 0; t < S.T; t++) {
            const int32_t b = ud->lane[0].ls->node_base[t], n = ud->lane[0].ls->node_base[t + 1] - b;
            if (!node_ci[t]) { s3a_set_error("s3a_uttdec_enable_pheur: no CI phones for tree %d", t); return S3A_EINVAL; }
            for (int32_t i = 0; i < n; i++) {
                if (node_ci[t][i] >= n_ci) { s3a_set_error("s3a_uttdec_enable_pheur: tree %d node %d: CI phone %d of %d", t, i, node_ci[t][i], n_ci); return S3A_EINVAL; }
                h[(size_t)b + i] = node_ci[t][i];
            }
        }
        for (int32_t j = 0; j < S.n_ci_sen; j++)         /* (entry n_ci_sen is only compared with) */
            if (sen2cimap[j] < 0 || sen2cimap[j] >= n_ci) { s3a_set_error("s3a_uttdec_enable_pheur: sen2cimap[%d] = %d", j, sen2cimap[j]); return S3A_EINVAL; }
        uint8_t *d_ci = NULL; int16_t *d_s2c = NULL;
        if (hipMalloc((void **)&d_ci, (size_t)S.N) != hipSuccess || hipMalloc((void **)&d_s2c, (size_t)(S.n_ci_sen + 1) * 2) != hipSuccess
            || hipMemcpy(d_ci, h.data(), (size_t)S.N, hipMemcpyHostToDevice) != hipSuccess
            || hipMemcpy(d_s2c, sen2cimap, (size_t)(S.n_ci_sen + 1) * 2, hipMemcpyHostToDevice) != hipSuccess) {
            if (d_ci) (void)hipFree(d_ci);
            if (d_s2c) (void)hipFree(d_s2c);
            s3a_set_error("s3a_uttdec_enable_pheur: device allocation failed");
            return S3A_ENOMEM;
        }
        S.node_ci = d_ci; S.sen2cimap = d_s2c;
        std::vector<ULane> tmp((size_t)ud->n_lanes);
        for (int32_t z = 0; z < ud->n_lanes; z++) {
            ULane &u = ud->lane[z].d;
            if (hipMalloc((void **)&u.ci_all, (size_t)ud->max_frames * S.n_ci_sen * 4) != hipSuccess
                || hipMalloc((void **)&u.heur_all, (size_t)ud->max_frames * n_ci * 4) != hipSuccess
                || hipMalloc((void **)&u.hth_pos, (size_t)3 * S.N * 4) != hipSuccess      /* (+ ku_weak_heur's survivors) */
                || hipMalloc((void **)&u.ph_scratch, ((size_t)3 * S.n_ci_sen + 8) * 4) != hipSuccess) {
                s3a_set_error("s3a_uttdec_enable_pheur: device allocation failed");
                return S3A_ENOMEM;
            }
            tmp[z] = u;
        }
        HIPCHK(hipMemcpy(ud->d_lanes, tmp.data(), sizeof(ULane) * ud->n_lanes, hipMemcpyHostToDevice));
    }
    S.pheurtype = pheurtype; S.pl_beam = pl_beam; S.pl_window = pl_window; S.n_ci = n_ci;
    return S3A_OK;
}

/* ---- the second pass behind every decode ---- */
extern "C" int32_t
s3a_uttdec_enable_bestpath(s3a_uttdec_t *ud, const s3a_dag_cfg_t *cfg, int32_t link_cap, int32_t pair_cap, int32_t keep_tables)
{
    if (!ud || !cfg) return S3A_EINVAL;
    HIPCHK(hipSetDevice(ud->device));
    if (ud->dag) { s3a_dagpass_free(ud->dag); ud->dag = NULL; }
    ud->dag = s3a_dagpass_init(ud->lm, cfg, ud->n_lanes, ud->vh_cap, ud->max_frames, link_cap, pair_cap);
    if (!ud->dag) return S3A_ENOMEM;
    for (int32_t z = 0; z < ud->n_lanes; z++) {
        const WLane &w = ud->lane[z].d.w;
        DagTab t;
        t.score = w.score; t.pred = w.pred; t.lw0 = w.lw0; t.lw1 = w.lw1; t.wid = w.wid; t.sf = w.sf; t.ef = w.ef; t.ascr = w.ascr;
        t.lscr = w.lscr; t.type = w.type; t.frame_start = w.frame_start; t.st = w.st; t.cap = w.cap;
        s3a_dagpass_bind(ud->dag, z, t);
    }
    ud->keep_tables = keep_tables ? 1 : 0;
    return S3A_OK;
}

extern "C" int32_t
s3a_uttdec_bestpath_result(s3a_uttdec_t *ud, int32_t lane, s3a_dag_result_t *out)
{
    if (!ud || !ud->dag || lane < 0 || lane >= ud->n_utt) return S3A_EINVAL;
    return s3a_dagpass_result(ud->dag, lane, out);
}

/* the second pass's hypothesis of utterance `utt` of the last queue (s3a_uttdec_bestpath_hyp's record and status codes) */
extern "C" int32_t
s3a_uttdec_queue_bestpath_hyp(s3a_uttdec_t *ud, int32_t utt, const char *uttid, int32_t utt_index, s3a_hyp_header_t *hdr,
                              s3a_hyp_word_t *words, int32_t max_words)
{
    if (!ud || !hdr || utt < 0 || utt >= ud->q_n || max_words < 0 || (max_words > 0 && !words)) return S3A_EINVAL;
    if (!ud->q_dag) { s3a_set_error("s3a_uttdec_queue_bestpath_hyp: the last queue ran without the second pass (s3a_uttdec_enable_bestpath)"); return S3A_EINVAL; }
    const int32_t *h = ud->q_hdr_h + (size_t)utt * UH_N, *io = ud->q_dio_h + (size_t)utt * DG_IO_N;
    memset(hdr, 0, sizeof *hdr);
    if (uttid) strncpy(hdr->uttid, uttid, sizeof hdr->uttid - 1);
    hdr->utt_index = utt_index; hdr->n_frames = ud->q_nfr[utt]; hdr->n_entry = io[DG_IO_NENT]; hdr->exit_id = io[DG_IO_ENDID];
    hdr->score = io[DG_IO_SCORE]; hdr->total_scale = h[UH_TSCALE];
    if (h[UH_ERR]) { hdr->status = -1; return S3A_OK; }
    const int32_t st = io[DG_IO_STATUS];
    if (st == DG_E_NOEXIT) { hdr->status = -2; return S3A_OK; }
    if (st == DG_E_NOPATH) { hdr->status = -4; return S3A_OK; }
    if (st == DG_E_CAP || st == DG_E_POSEDGE) {
        s3a_set_error("s3a_uttdec_queue_bestpath_hyp: the second pass of utterance %d gave up with status %d (3: a capacity of the pass or "
                      "-maxedge in the filler bypass, 5: positive bypass edge): no hypothesis from it", utt, st);
        hdr->status = -5;
        return S3A_OK;
    }
    if (st != 0) { s3a_set_error("s3a_uttdec_queue_bestpath_hyp: the second pass of utterance %d stopped with status %d", utt, st); return S3A_EUNSUP; }
    hdr->n_words = io[DG_IO_NWORDS];
    if (io[UD_WOFF] < 0) { s3a_set_error("s3a_uttdec_queue_bestpath_hyp: the packed word buffer was too small"); return S3A_ENOMEM; }
    if (hdr->n_words > max_words) { hdr->status = -3; return S3A_OK; }
    const int32_t *w = ud->q_dw_h + (size_t)io[UD_WOFF] * 6;
    for (int32_t q = 0; q < hdr->n_words; q++) {
        s3a_hyp_word_t &o = words[q];
        o.wid = w[6 * q]; o.sf = w[6 * q + 1]; o.ef = w[6 * q + 2]; o.ascr = w[6 * q + 3]; o.lscr = w[6 * q + 4]; o.scale = w[6 * q + 5];
    }
    return S3A_OK;
}

/* lattices out of a queue: asked for before s3a_uttdec_decode_queue*, the lanes' lattices are read back behind every group's (refill
 * event's) second pass and kept per utterance until the next decode */
extern "C" int32_t
s3a_uttdec_queue_keep_lattices(s3a_uttdec_t *ud, int32_t on)
{
    if (!ud || !ud->dag) { s3a_set_error("s3a_uttdec_queue_keep_lattices: no second pass on this engine (s3a_uttdec_enable_bestpath)"); return S3A_EINVAL; }
    ud->q_keep_lat = on != 0;
    return S3A_OK;
}

extern "C" int32_t
s3a_uttdec_queue_lattice(s3a_uttdec_t *ud, int32_t utt, s3a_lat_info_t *info, s3a_lat_node_t *nodes, int32_t node_cap, s3a_lat_link_t *links,
                         int32_t link_cap)
{
    if (!ud || !info || utt < 0 || utt >= ud->q_n || (size_t)utt >= ud->q_lat.size()) {
        s3a_set_error("s3a_uttdec_queue_lattice: no kept lattices (s3a_uttdec_queue_keep_lattices before the queue's decode) or bad utterance");
        return S3A_EINVAL;
    }
    const s3a_uttdec_s::QLat &q = ud->q_lat[(size_t)utt];
    if (!q.have) { s3a_set_error("s3a_uttdec_queue_lattice: utterance %d has no lattice (its second pass's status says why)", utt); return S3A_EUNSUP; }
    *info = q.info;
    if (!nodes || !links) return S3A_OK;
    if (node_cap < q.info.n_nodes || link_cap < q.info.n_links) { s3a_set_error("s3a_uttdec_queue_lattice: buffers too small"); return S3A_EINVAL; }
    memcpy(nodes, q.nodes.data(), (size_t)q.info.n_nodes * sizeof(s3a_lat_node_t));
    memcpy(links, q.links.data(), (size_t)q.info.n_links * sizeof(s3a_lat_link_t));
    return S3A_OK;
}

extern "C" int32_t
s3a_uttdec_lattice(s3a_uttdec_t *ud, int32_t lane, s3a_lat_info_t *info, s3a_lat_node_t *nodes, int32_t node_cap,
                   s3a_lat_link_t *links, int32_t link_cap)
{
    if (!ud || !ud->dag || lane < 0 || lane >= ud->n_utt) { s3a_set_error("s3a_uttdec_lattice: no second pass on this engine (s3a_uttdec_enable_bestpath) or bad lane"); return S3A_EINVAL; }
    if (ud->q_n) { s3a_set_error("s3a_uttdec_lattice: a queue keeps no lattices (decode in lock-step batches)"); return S3A_EUNSUP; }
    return s3a_dagpass_lattice(ud->dag, lane, info, nodes, node_cap, links, link_cap);
}

extern "C" int32_t
s3a_uttdec_bestpath_hyp(s3a_uttdec_t *ud, int32_t lane, const char *uttid, int32_t utt_index, s3a_hyp_header_t *hdr,
                        s3a_hyp_word_t *words, int32_t max_words)
{
    if (!ud || !ud->dag || !hdr || lane < 0 || lane >= ud->n_utt || max_words < 0 || (max_words > 0 && !words)) return S3A_EINVAL;
    const HostLane &hl = ud->lane[lane];
    s3a_dag_result_t r;
    int32_t rc = s3a_dagpass_result(ud->dag, lane, &r);
    if (rc != S3A_OK) return rc;
    if ((rc = uttdec_fetch(ud, false)) != S3A_OK) return rc;       /* (the frame normalisers of this decode's lanes) */
    memset(hdr, 0, sizeof *hdr);
    if (uttid) strncpy(hdr->uttid, uttid, sizeof hdr->uttid - 1);
    hdr->utt_index = utt_index; hdr->n_frames = hl.nfr; hdr->n_entry = r.n_entry; hdr->exit_id = r.endid; hdr->score = r.score;
    for (int32_t f = 0; f < hl.nfr; f++) hdr->total_scale = h_add(hdr->total_scale, hl.h_fstat[8 * f]);
    if (hl.h_ctx->err) { hdr->status = -1; return S3A_OK; }
    if (r.status == DG_E_NOEXIT) { hdr->status = -2; return S3A_OK; }
    if (r.status == DG_E_NOPATH) { hdr->status = -4; return S3A_OK; }     /* "Bestpath search failed": the reference writes no line */
    if (r.status == DG_E_CAP || r.status == DG_E_POSEDGE) {
        /* THIS utterance's second pass gave up (a capacity of the pass, -maxedge during the filler bypass, a positive bypass
         * edge): a failed utterance, as "Bestpath search failed" is in the reference -- never a failed batch */
        s3a_set_error("s3a_uttdec_bestpath_hyp: the second pass of lane %d gave up with status %d (3: a capacity of the pass or "
                      "-maxedge in the filler bypass, 5: positive bypass edge): no hypothesis from it", lane, r.status);
        hdr->status = -5;
        return S3A_OK;
    }
    if (r.status != 0) {
        s3a_set_error("s3a_uttdec_bestpath_hyp: the second pass of lane %d stopped with status %d (4: inconsistent table)", lane, r.status);
        return S3A_EUNSUP;
    }
    hdr->n_words = r.n_words;
    if (r.n_words > max_words) { hdr->status = -3; return S3A_OK; }
    for (int32_t q = 0; q < r.n_words; q++) {
        s3a_hyp_word_t &w = words[q];
        w.wid = r.wid[q]; w.sf = r.sf[q]; w.ef = r.ef[q]; w.ascr = r.ascr[q]; w.lscr = r.lscr[q];
        int32_t sc = 0;
        for (int32_t i = w.sf; i < w.ef && i < hl.nfr; i++) if (i >= 0) sc = h_add(sc, hl.h_fstat[8 * i]);
        w.scale = sc;
    }
    return S3A_OK;
}

extern "C" double
s3a_uttdec_last_decode_ms(const s3a_uttdec_t *ud)
{
    return ud ? ud->last_decode_ms : 0.0;
}

/* diagnostics: lextree_utt_end on every lane, then the node records of `lane` that are not an inactive HMM
 * (see ku_selfcheck); all zeros (out[6] = INT_MAX) on a healthy lane */
extern "C" int32_t
s3a_uttdec_selfcheck(s3a_uttdec_t *ud, int32_t lane, int32_t *out8)
{
    if (!ud || !out8 || lane < 0 || lane >= ud->n_lanes) return S3A_EINVAL;
    HIPCHK(hipSetDevice(ud->device));
    int32_t *d = NULL, init[8] = { 0, 0, 0, 0, 0, 0, INT_MAX, 0 };
    HIPCHK(hipMalloc((void **)&d, sizeof init));
    HIPCHK(hipMemcpyAsync(d, init, sizeof init, hipMemcpyHostToDevice, ud->stream));
    hipLaunchKernelGGL(ku_lanes_end, dim3(8, 2 * ud->S.T, ud->n_lanes), dim3(256), 0, ud->stream, ud->d_lanes, ud->S, (const int32_t *)NULL, 0);
    hipLaunchKernelGGL(ku_selfcheck, dim3(64), dim3(256), 0, ud->stream, ud->d_lanes, ud->S, lane, d);
    HIPCHK(hipMemcpyAsync(out8, d, sizeof init, hipMemcpyDeviceToHost, ud->stream));
    HIPCHK(hipStreamSynchronize(ud->stream));
    (void)hipFree(d);
    if (ud->d_dbg) {
        int32_t g[16];
        HIPCHK(hipMemcpy(g, ud->d_dbg + 16 * lane, sizeof g, hipMemcpyDeviceToHost));
        if (g[0] < 0) fprintf(stderr, "framecheck lane %d: scratch left set first at frame %d (utterance of %d frames): index %d (tree %d, offset %d; current list length %d): turn %d selfemit %d cnt %d\n",
                              lane, -g[0] - 1, g[5], g[1], g[14], g[7], g[6], g[2], g[3], g[4]);
        else if (g[0]) fprintf(stderr, "framecheck lane %d: first violation at frame %d node %d (tree %d): sc0 %d sc1 %d outs %d bests %d frame-tag %d posf %d pos %d turn %d | next list length %d, act[pos] %d | clean %d listed %d\n",
                          lane, g[0] - 1, g[1], g[14], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[9], g[10], g[11], g[12], g[13]);
    }
    return S3A_OK;
}

extern "C" int32_t
s3a_uttdec_n_lanes(const s3a_uttdec_t *ud)
{
    return ud ? ud->n_lanes : 0;
}

extern "C" int32_t
s3a_uttdec_window(const s3a_uttdec_t *ud)
{
    return ud ? ud->S.win_K : 0;
}

/* ------------------------------------------------------------------ */
/* the word level on its own (parity tests: frame by frame against the oracle) */
/* ------------------------------------------------------------------ */
struct s3a_wltest_s {
    s3a_lm3g_t *lm;
    WDict dict;
    WPar par;
    ULane lane, *d_lane;
    UCtx *h_ctx;
    int32_t *d_lcmap, *h_pack, *h_tab;
    int32_t T, n_ci, n_word, start_lwid, max_exits, cap, max_frames;
    size_t h_tab_cap;
    hipStream_t stream;
};

extern "C" void
s3a_wltest_free(s3a_wltest_t *wt)
{
    if (!wt) return;
    wlane_free(wt->lane.w);
    const void *q[] = { wt->dict.lwid, wt->dict.fillpen, wt->dict.last_ci, wt->dict.is_filler, wt->d_lcmap, wt->lane.ctx,
                        wt->lane.pack, wt->d_lane };
    for (auto p : q) if (p) (void)hipFree((void *)p);
    if (wt->h_ctx) (void)hipHostFree(wt->h_ctx);
    if (wt->h_pack) (void)hipHostFree(wt->h_pack);
    if (wt->h_tab) (void)hipHostFree(wt->h_tab);
    if (wt->stream) (void)hipStreamDestroy(wt->stream);
    delete wt;
}

/* cfg: the dictionary / pruning fields of s3a_wordlevel_cfg_t; lcmap_len[t * (n_ci + 1) + p] = length of the root list
 * lextree_enter(tree t, left context p) walks (p == n_ci: none), negative = not a context of that tree */
extern "C" s3a_wltest_t *
s3a_wltest_init(s3a_lm3g_t *lm, const s3a_wordlevel_cfg_t *cfg, const int32_t *lcmap_len, int32_t vh_cap,
                int32_t cand_cap, int32_t max_exits, int32_t max_frames)
{
    if (!lm || !cfg || !lcmap_len || cfg->n_ci <= 0 || cfg->n_ci > 255 || 2 * cfg->n_lextree > WL_MAXT) {
        s3a_set_error("s3a_wltest_init: bad arguments");
        return NULL;
    }
    s3a_wltest_t *wt = new s3a_wltest_s();
    memset((void *)wt, 0, sizeof *wt);
    wt->lm = lm; wt->T = 2 * cfg->n_lextree; wt->n_ci = cfg->n_ci; wt->n_word = cfg->n_word; wt->start_lwid = cfg->start_lwid;
    wt->max_exits = max_exits; wt->cap = vh_cap; wt->max_frames = max_frames;
    const int32_t T = wt->T, hdr = 6 * T + 16;
    std::vector<int32_t> lcmap((size_t)T * (cfg->n_ci + 1) * 2);
    {
        int32_t off = 0;
        for (size_t i = 0; i < (size_t)T * (cfg->n_ci + 1); i++) { lcmap[2 * i] = off; lcmap[2 * i + 1] = lcmap_len[i]; if (lcmap_len[i] > 0) off += lcmap_len[i]; }
    }
    int32_t *p0 = NULL, *p1 = NULL, *p2 = NULL;
    uint8_t *p3 = NULL;
    if (hipStreamCreateWithFlags(&wt->stream, hipStreamNonBlocking) != hipSuccess) { s3a_set_error("s3a_wltest_init: no HIP device"); delete wt; return NULL; }
    DM(p0, (size_t)cfg->n_word * 4); DM(p1, (size_t)cfg->n_word * 4); DM(p2, (size_t)cfg->n_word * 4); DM(p3, (size_t)cfg->n_word);
    wt->dict.lwid = p0; wt->dict.fillpen = p1; wt->dict.last_ci = p2; wt->dict.is_filler = p3;
    wt->dict.n_word = cfg->n_word; wt->dict.n_ci = cfg->n_ci;
    if (hipMemcpy(p0, cfg->lwid, (size_t)cfg->n_word * 4, hipMemcpyHostToDevice) != hipSuccess
        || hipMemcpy(p1, cfg->fillpen, (size_t)cfg->n_word * 4, hipMemcpyHostToDevice) != hipSuccess
        || hipMemcpy(p2, cfg->last_ci, (size_t)cfg->n_word * 4, hipMemcpyHostToDevice) != hipSuccess
        || hipMemcpy(p3, cfg->is_filler, (size_t)cfg->n_word, hipMemcpyHostToDevice) != hipSuccess) goto fail;
    UPV(wt->d_lcmap, lcmap);
    wt->par.wbeam = cfg->wbeam_vh; wt->par.bghist = cfg->bghist; wt->par.maxwpf = cfg->maxwpf; wt->par.maxhist = cfg->maxhistpf;
    wt->par.wordend = cfg->wordend_beam; wt->par.n_lextree = cfg->n_lextree; wt->par.epl = cfg->epl; wt->par.T = T;
    wt->par.hmmbeam = cfg->hmmbeam; wt->par.lcmap = wt->d_lcmap;
    for (int32_t t = 0; t < T; t++) wt->par.tree_type[t] = cfg->tree_type[t];
    if (wlane_alloc(wt->lane.w, vh_cap, max_frames, max_exits, cand_cap, cand_cap < (1 << 18) ? cand_cap : (1 << 18), cfg->n_word, wt->stream) != S3A_OK) goto fail;
    DM(wt->lane.ctx, sizeof(UCtx));
    DM(wt->lane.pack, (size_t)(hdr + 3 * max_exits) * 4);
    DM(wt->d_lane, sizeof(ULane));
    if (hipMemcpy(wt->d_lane, &wt->lane, sizeof(ULane), hipMemcpyHostToDevice) != hipSuccess
        || hipHostMalloc((void **)&wt->h_ctx, sizeof(UCtx)) != hipSuccess
        || hipHostMalloc((void **)&wt->h_pack, (size_t)(hdr + 3 * max_exits) * 4) != hipSuccess) goto fail;
    return wt;
fail:
    s3a_wltest_free(wt);
    return NULL;
}

/* vithist_utt_begin: entry 0 */
extern "C" int32_t
s3a_wltest_begin(s3a_wltest_t *wt, int32_t startwid, int32_t n_frames)
{
    if (!wt || n_frames <= 0 || n_frames > wt->max_frames) return S3A_EINVAL;
    WLane &w = wt->lane.w;
    int32_t rc;
    const int32_t e0[10] = { 0, -1, wt->start_lwid, -1, startwid, -1, -1, 0, 0, 0 };
    int32_t *arr[10] = { w.score, w.pred, w.lw0, w.lw1, w.wid, w.sf, w.ef, w.ascr, w.lscr, w.type };
    for (int k = 0; k < 10; k++) if ((rc = fill32(wt->stream, arr[k], e0[k], 1)) != S3A_OK) return rc;
    int32_t c5[5] = { 0, 0, 0, 0, -1 };
    if (wt->lm->d.n_bg > 0 && wt->start_lwid >= 0) { c5[3] = wt->lm->ug_firstbg[wt->start_lwid]; c5[4] = wt->lm->ug_firstbg[wt->start_lwid + 1] - c5[3]; }
    for (int k = 0; k < 5; k++) if ((rc = fill32(wt->stream, w.lmc + (size_t)k * w.cap, c5[k], 1)) != S3A_OK) return rc;
    if ((rc = fill32(wt->stream, w.frame_start, 1, 1)) || (rc = fill32(wt->stream, w.bestscore, INT_MIN, 1))
        || (rc = fill32(wt->stream, w.bestvh, -1, 1)) || (rc = fill32(wt->stream, w.st, 1, 1)) || (rc = fill32(wt->stream, w.st + 1, 0, 1)))
        return rc;
    memset(wt->h_ctx, 0, sizeof(UCtx));
    wt->h_ctx->active = 1; wt->h_ctx->nfr = n_frames; wt->h_ctx->n_lextrans = 1;
    HIPCHK(hipMemcpyAsync(wt->lane.ctx, wt->h_ctx, sizeof(UCtx), hipMemcpyHostToDevice, wt->stream));
    HIPCHK(hipStreamSynchronize(wt->stream));
    return S3A_OK;
}

/* One frame: n_exit[T] word exits per tree, exits = (wid, score, history) x total in tree-then-list order;
 * best_hmm / best_word / word_thres = beam_t.bestscore / .bestwordscore / .word_thres of the frame.
 * Out: the lextree_enter calls it leaves: calls[4c] = {score, history, root-list offset, first entry}. */
extern "C" int32_t
s3a_wltest_frame(s3a_wltest_t *wt, const int32_t *n_exit, const int32_t *exits, int32_t best_hmm, int32_t best_word,
                 int32_t word_thres, int32_t *n_calls, int32_t *calls, int32_t *thresh, int32_t *n_ent)
{
    if (!wt || !n_exit || !n_calls || !calls) return S3A_EINVAL;
    const int32_t T = wt->T, hdr = 6 * T + 16;
    int32_t total = 0;
    memset(wt->h_pack, 0, (size_t)hdr * 4);
    for (int32_t t = 0; t < T; t++) { wt->h_pack[3 * T + 8 + t] = n_exit[t]; total += n_exit[t]; }
    if (total > wt->max_exits) return S3A_EINVAL;
    wt->h_pack[3 * T + 2] = word_thres; wt->h_pack[3 * T + 3] = best_hmm; wt->h_pack[3 * T + 4] = best_word;
    if (total) memcpy(wt->h_pack + hdr, exits, (size_t)3 * total * 4);
    HIPCHK(hipMemcpyAsync(wt->lane.pack, wt->h_pack, (size_t)(hdr + 3 * total) * 4, hipMemcpyHostToDevice, wt->stream));
    hipLaunchKernelGGL(ku_wordlevel_only, dim3(1, 1, 1), dim3(WL_THREADS), 0, wt->stream, wt->d_lane, wt->lm->d, wt->dict, wt->par);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(wt->h_ctx, wt->lane.ctx, sizeof(UCtx), hipMemcpyDeviceToHost, wt->stream));
    HIPCHK(hipStreamSynchronize(wt->stream));
    if (wt->h_ctx->err) { s3a_set_error("s3a_wltest_frame: error bits 0x%x", wt->h_ctx->err); return S3A_EINVAL; }
    *n_calls = wt->h_ctx->n_calls;
    memcpy(calls, wt->h_ctx->calls, (size_t)4 * wt->h_ctx->n_calls * 4);
    if (thresh) *thresh = wt->h_ctx->thresh;
    if (n_ent) *n_ent = wt->h_ctx->n_ent;
    return S3A_OK;
}

/* the table so far: out[k * n_entry + id], k = score pred lw0 lw1 wid sf ef ascr lscr type; frames[f] for
 * f <= n_frm: frame_start | bestscore | bestvh (each max_out_frames long) */
extern "C" int32_t
s3a_wltest_fetch(s3a_wltest_t *wt, int32_t *n_entry, int32_t *n_frm, int32_t *out, int32_t max_entries,
                 int32_t *frames, int32_t max_out_frames, int32_t *n_tie_frames)
{
    if (!wt || !n_entry || !n_frm) return S3A_EINVAL;
    int32_t st[2];
    const WLane &w = wt->lane.w;
    HIPCHK(hipMemcpy(st, w.st, 8, hipMemcpyDeviceToHost));
    *n_entry = st[0]; *n_frm = st[1];
    if (n_tie_frames) *n_tie_frames = wt->h_ctx->n_tie_frames;
    if (out) {
        if (st[0] > max_entries) return S3A_EINVAL;
        const int32_t *src[10] = { w.score, w.pred, w.lw0, w.lw1, w.wid, w.sf, w.ef, w.ascr, w.lscr, w.type };
        for (int k = 0; k < 10; k++) HIPCHK(hipMemcpy(out + (size_t)k * st[0], src[k], (size_t)st[0] * 4, hipMemcpyDeviceToHost));
    }
    if (frames) {
        if (st[1] + 1 > max_out_frames) return S3A_EINVAL;
        HIPCHK(hipMemcpy(frames, w.frame_start, (size_t)(st[1] + 1) * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(frames + max_out_frames, w.bestscore, (size_t)(st[1] + 1) * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(frames + 2 * (size_t)max_out_frames, w.bestvh, (size_t)(st[1] + 1) * 4, hipMemcpyDeviceToHost));
    }
    return S3A_OK;
}

/* ------------------------------------------------------------------ */
/* the hypothesis of a finished lane (host): vithist_utt_end + vithist_backtrace */
/* ------------------------------------------------------------------ */
/* what ku_hyp left for the lane (vithist_utt_end + vithist_backtrace + compute_scale on the device): header + words.
 * Status 0 ok, -1 the decode stopped with an error, -2 no word exit at all (vithist_utt_end returns -1), -3 more words
 * than the caller has room for (n_words = what it takes). */
static int32_t
hyp_from_header(const int32_t *h, const int32_t *words_base, int32_t nfr, int32_t which, const char *uttid, int32_t utt_index,
                s3a_hyp_header_t *rec, s3a_hyp_word_t *words, int32_t max_words)
{
    memset(rec, 0, sizeof *rec);
    if (uttid) strncpy(rec->uttid, uttid, sizeof rec->uttid - 1);
    rec->utt_index = utt_index; rec->n_frames = nfr; rec->n_entry = h[UH_NENTRY]; rec->status = h[UH_STATUS];
    rec->total_scale = h[UH_TSCALE];
    if (h[UH_STATUS] == -3) {
        s3a_set_error("s3a_uttdec_hyp: utterance %d: a hypothesis of %d words in an utterance of %d frames", which, h[UH_NWORDS], nfr);
        return S3A_EINVAL;
    }
    if (rec->status != 0) return S3A_OK;
    rec->n_words = h[UH_NWORDS]; rec->score = h[UH_SCORE]; rec->exit_id = h[UH_EXIT];
    if (rec->n_words > max_words) { rec->status = -3; return S3A_OK; }
    static_assert(sizeof(s3a_hyp_word_t) == 24, "s3a_hyp_word_t is six int32");
    memcpy(words, words_base + (size_t)h[UH_WOFF] * 6, (size_t)rec->n_words * sizeof(s3a_hyp_word_t));
    return S3A_OK;
}

static int32_t
uttdec_hyp(s3a_uttdec_t *ud, int32_t lane, const char *uttid, int32_t utt_index, s3a_hyp_header_t *rec,
           s3a_hyp_word_t *words, int32_t max_words)
{
    if (!ud || !rec || lane < 0 || lane >= ud->n_utt || max_words < 0 || (max_words > 0 && !words)) return S3A_EINVAL;
    return hyp_from_header(ud->h_hyp_hdr + (size_t)lane * UH_N, ud->h_hyp_words, ud->lane[lane].nfr, lane, uttid, utt_index, rec, words, max_words);
}

/* after s3a_uttdec_decode_queue: the hypothesis of utterance `utt` of the queue (as s3a_uttdec_hyp_var for a lane) */
extern "C" int32_t
s3a_uttdec_queue_hyp(s3a_uttdec_t *ud, int32_t utt, const char *uttid, int32_t utt_index, s3a_hyp_header_t *hdr,
                     s3a_hyp_word_t *words, int32_t max_words)
{
    if (!ud || !hdr || utt < 0 || utt >= ud->q_n || max_words < 0 || (max_words > 0 && !words)) return S3A_EINVAL;
    return hyp_from_header(ud->q_hdr_h + (size_t)utt * UH_N, ud->q_words_h, ud->q_nfr[utt], utt, uttid, utt_index, hdr, words, max_words);
}

/* ... and what stopped it, if anything: the word level's error bits (0: decoded), the frame it stopped at, the largest
 * candidate / new-entry counts of a frame (capacity planning) */
extern "C" int32_t
s3a_uttdec_queue_status(s3a_uttdec_t *ud, int32_t utt, int32_t *err, int32_t *stopped_at, int32_t *max_cand, int32_t *max_new)
{
    if (!ud || utt < 0 || utt >= ud->q_n) return S3A_EINVAL;
    const int32_t *h = ud->q_hdr_h + (size_t)utt * UH_N;
    if (err) *err = h[UH_ERR];
    if (stopped_at) *stopped_at = h[UH_CF];
    if (max_cand) *max_cand = h[UH_MAXCAND];
    if (max_new) *max_new = h[UH_MAXNEW];
    return S3A_OK;
}

extern "C" int32_t
s3a_uttdec_hyp(s3a_uttdec_t *ud, int32_t lane, const char *uttid, int32_t utt_index, s3a_hyp_record_t *rec)
{
    if (!rec) return S3A_EINVAL;
    memset(rec, 0, sizeof *rec);
    const int32_t rc = uttdec_hyp(ud, lane, uttid, utt_index, (s3a_hyp_header_t *)rec, rec->word, S3A_HYP_MAXW);
    if (rc == S3A_OK && rec->status == -3) rec->n_words = 0;
    return rc;
}

/* the same without the word limit: the header (what s3a_hyp_record_t begins with) + as many words as the hypothesis has.
 * max_words too small (0 to ask): status -3, hdr->n_words = the number it takes, nothing written to words[] */
extern "C" int32_t
s3a_uttdec_hyp_var(s3a_uttdec_t *ud, int32_t lane, const char *uttid, int32_t utt_index, s3a_hyp_header_t *hdr,
                   s3a_hyp_word_t *words, int32_t max_words)
{
    return uttdec_hyp(ud, lane, uttid, utt_index, hdr, words, max_words);
}

/* match_write / matchseg_write (libsearch/srch_output.c:74-161) for one record: the -hyp and -hypseg lines.
 * wordstr / basewid / is_filler by dictionary word id; lw, wip = lm_t.lw, lm_t.wip (lm_rawscore, lm.c:2171-2178). */
extern "C" int32_t
s3a_hyp_format_var(const s3a_hyp_header_t *rec, const s3a_hyp_word_t *words, const char *const *wordstr, const int32_t *basewid,
                   const uint8_t *is_filler, int32_t startwid, int32_t finishwid, float lw, int32_t wip, int32_t unscale,
                   char *match_line, size_t match_cap, char *seg_line, size_t seg_cap)
{
    if (!rec || !wordstr || !basewid || !is_filler || !match_line || !seg_line || (rec->n_words > 0 && !words)) return S3A_EINVAL;
    size_t mp = 0, sp = 0;
#define APP(buf, pos, cap, ...) do { int w_ = snprintf((buf) + (pos), (pos) < (cap) ? (cap) - (pos) : 0, __VA_ARGS__); \
        if (w_ < 0 || (pos) + (size_t)w_ >= (cap)) return S3A_EINVAL; (pos) += (size_t)w_; } while (0)
    int counter = 0;
    if (rec->n_words == 0) APP(match_line, mp, match_cap, "(null)");
    for (int32_t q = 0; q < rec->n_words; q++) {
        const s3a_hyp_word_t &w = words[q];
        if (w.sf == w.ef) continue;
        if (!is_filler[w.wid] && w.wid != finishwid && w.wid != startwid) APP(match_line, mp, match_cap, "%s ", wordstr[basewid[w.wid]]);
        counter++;
    }
    if (counter == 0) APP(match_line, mp, match_cap, " ");
    APP(match_line, mp, match_cap, "(%s)\n", rec->uttid);
    int32_t ascr = 0, lscr = 0, gscale = 0;
    auto raw = [&](int32_t s) { s -= wip; float fs = (float)s; fs /= lw; return (int32_t)fs; };
    for (int32_t q = 0; q < rec->n_words; q++) {
        const s3a_hyp_word_t &w = words[q];
        if (w.sf == w.ef) continue;
        ascr += w.ascr; lscr += raw(w.lscr);
        if (unscale) gscale += w.scale;
    }
    APP(seg_line, sp, seg_cap, "%s S %d T %d A %d L %d", rec->uttid, rec->total_scale, ascr + lscr + gscale, ascr + gscale, lscr);
    for (int32_t q = 0; q < rec->n_words; q++) {
        const s3a_hyp_word_t &w = words[q];
        if (w.sf == w.ef) continue;
        APP(seg_line, sp, seg_cap, " %d %d %d %s", w.sf, w.ascr + (unscale ? w.scale : 0), raw(w.lscr), wordstr[w.wid]);
    }
    APP(seg_line, sp, seg_cap, " %d\n", rec->n_frames);
#undef APP
    return S3A_OK;
}

extern "C" int32_t
s3a_hyp_format(const s3a_hyp_record_t *rec, const char *const *wordstr, const int32_t *basewid, const uint8_t *is_filler,
               int32_t startwid, int32_t finishwid, float lw, int32_t wip, int32_t unscale, char *match_line,
               size_t match_cap, char *seg_line, size_t seg_cap)
{
    if (!rec) return S3A_EINVAL;
    return s3a_hyp_format_var((const s3a_hyp_header_t *)rec, rec->word, wordstr, basewid, is_filler, startwid, finishwid, lw,
                              wip, unscale, match_line, match_cap, seg_line, seg_cap);
}

/* per-kernel timing: from now on every `every`-th frame of a decode is bracketed, launch by launch, by HIP events
 * on the launch stream (0 = off; totals are reset).  s3a_uttdec_profile returns, per kernel class, the summed
 * microseconds and the number of timed launches since; *names (optional) = the class names. */
extern "C" int32_t
s3a_uttdec_set_profile(s3a_uttdec_t *ud, int32_t every)
{
    if (!ud || every < 0) return S3A_EINVAL;
    ud->prof_every = every;
    memset(ud->prof_us, 0, sizeof ud->prof_us); memset(ud->prof_n, 0, sizeof ud->prof_n);
    return S3A_OK;
}

extern "C" int32_t
s3a_uttdec_profile(const s3a_uttdec_t *ud, double *us, int64_t *launches, const char **names, int32_t max_classes)
{
    if (!ud || !us || !launches) return S3A_EINVAL;
    const int32_t n = max_classes < UK_N ? max_classes : UK_N;
    for (int32_t k = 0; k < n; k++) { us[k] = ud->prof_us[k]; launches[k] = ud->prof_n[k]; if (names) names[k] = uk_names[k]; }
    return n;
}

/* what one launch of the scoring kernels moves, for roofline arithmetic: Gaussians of the model (padded lanes
 * excluded), feature dimension, senones, CI senones */
extern "C" int32_t
s3a_uttdec_shape(const s3a_uttdec_t *ud, int32_t *n_sen, int32_t *n_ci_sen, int32_t *n_comp_padded, int32_t *veclen,
                 int32_t *n_node, int32_t *n_tree)
{
    if (!ud) return S3A_EINVAL;
    if (n_sen) *n_sen = ud->S.n_sen;
    if (n_ci_sen) *n_ci_sen = ud->S.n_ci_sen;
    if (n_comp_padded) *n_comp_padded = ud->S.CP;
    if (veclen) *veclen = ud->veclen;
    if (n_node) *n_node = ud->S.N;
    if (n_tree) *n_tree = ud->S.T;
    return S3A_OK;
}
