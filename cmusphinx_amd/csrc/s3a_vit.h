/*
 * s3a_vit.h -- the 3- and 5-state left-to-right HMM Viterbi updates shared by the batched
 * hmm_vit_eval (s3a_hmm.hip, int64 histories like the reference's union) and the
 * lexical-tree search (s3a_lextree.hip, int32 vithist ids).
 * Restates hmm_vit_eval_3st_lr, sphinx3/src/libs3decoder/libam/hmm.c:592-674, and hmm_vit_eval_5st_lr, hmm.c:285-412.
 */
#ifndef S3A_VIT_H
#define S3A_VIT_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include "cmusphinx_amd.h"

#define S3A_NS 5
#define S3A_WORST S3A_LOGPROB_ZERO

__device__ __forceinline__ int32_t
add32(int32_t a, int32_t b)
{
    return (int32_t)((uint32_t)a + (uint32_t)b);
}

/* per-lane working copy of one HMM; H = history id type */
template <typename H>
struct HmmRegsT {
    int32_t s[S3A_NS];
    H h[S3A_NS];
    int32_t out;
    H outh;
    int32_t ssid[S3A_NS];
};

#define PICK3(dst, t0, t1, t2, on_t1, on_t2)     \
    do {                                         \
        if ((t0) > (t1)) {                       \
            if ((t2) > (t0)) { dst = (t2); on_t2; } else dst = (t0); \
        }                                        \
        else {                                   \
            if ((t2) > (t1)) { dst = (t2); on_t2; } else { dst = (t1); on_t1; } \
        }                                        \
    } while (0)

/*
 * hmm_vit_eval_3st_lr, hmm.c:592-674.  V[k] = state k's score plus its senone score, all three taken before any
 * state is updated.  A dead source (V <= WORST_SCORE) contributes WORST_SCORE, not a sum; the skip arcs 0 -> 2 and
 * 1 -> exit count only where the matrix has them (the 0 -> 2 arc then without looking at V[0]); into the exit state an
 * absent skip arc loses against anything (INT_MIN), into state 2 it ties with a dead source.  Among (self, previous,
 * skip) self wins only strictly over previous, skip only strictly over that winner (PICK3): the history follows.
 */
template <typename H>
__device__ __forceinline__ int32_t
vit3(HmmRegsT<H> &r, const int32_t *tp, int32_t e0, int32_t e1, int32_t e2)
{
    const int32_t V0 = add32(r.s[0], e0), V1 = add32(r.s[1], e1), V2 = add32(r.s[2], e2);
    const bool live0 = V0 > S3A_WORST, live1 = V1 > S3A_WORST, live2 = V2 > S3A_WORST;
#define ARC3(i, j) tp[(i) * 4 + (j)]
    /* exit state: from 2, or over the skip arc from 1 */
    const int32_t x2 = live2 ? add32(V2, ARC3(2, 3)) : S3A_WORST;
    const int32_t x1 = (live1 && ARC3(1, 3) > S3A_WORST) ? add32(V1, ARC3(1, 3)) : INT_MIN;
    int32_t best = x2 > x1 ? x2 : x1;
    r.outh = x2 > x1 ? r.h[2] : r.h[1];
    if (best < S3A_WORST) best = S3A_WORST;
    r.out = best;
    /* state 2: self, from 1, over the skip arc from 0 */
    const int32_t c_self = live2 ? add32(V2, ARC3(2, 2)) : S3A_WORST;
    const int32_t c_prev = live1 ? add32(V1, ARC3(1, 2)) : S3A_WORST;
    const int32_t c_skip = ARC3(0, 2) > S3A_WORST ? add32(V0, ARC3(0, 2)) : S3A_WORST;
    int32_t n2;
    PICK3(n2, c_self, c_prev, c_skip, r.h[2] = r.h[1], r.h[2] = r.h[0]);
    if (n2 < S3A_WORST) n2 = S3A_WORST;
    if (n2 > best) best = n2;
    r.s[2] = n2;
    /* state 1: self, from 0 */
    const int32_t d_self = live1 ? add32(V1, ARC3(1, 1)) : S3A_WORST;
    const int32_t d_prev = live0 ? add32(V0, ARC3(0, 1)) : S3A_WORST;
    int32_t n1 = d_self;
    if (!(d_self > d_prev)) { n1 = d_prev; r.h[1] = r.h[0]; }
    if (n1 < S3A_WORST) n1 = S3A_WORST;
    if (n1 > best) best = n1;
    r.s[1] = n1;
    /* state 0: self */
    int32_t n0 = add32(V0, ARC3(0, 0));
    if (n0 < S3A_WORST) n0 = S3A_WORST;
    if (n0 > best) best = n0;
    r.s[0] = n0;
    return best;
#undef ARC3
}

/* hmm_vit_eval_5st_lr, hmm.c:285-412 (note: the exit state and states 4, 3 are
 * only re-computed when the state two below them is alive, exactly as there) */
template <typename H>
__device__ __forceinline__ int32_t
vit5(HmmRegsT<H> &r, const int32_t *tp, const int32_t *e, int32_t &out_written)
{
    int32_t s5, s4, s3, s2, s1, s0, t2, t1, t0, best = S3A_WORST;
    s4 = add32(r.s[4], e[4]);
    s3 = add32(r.s[3], e[3]);
    if (s3 > S3A_WORST) {
        t1 = add32(s4, tp[4 * 6 + 5]);
        t2 = add32(s3, tp[3 * 6 + 5]);
        if (t1 > t2) { s5 = t1; r.outh = r.h[4]; }
        else         { s5 = t2; r.outh = r.h[3]; }
        if (s5 < S3A_WORST) s5 = S3A_WORST;
        r.out = s5;
        best = s5;
        out_written = 1;
    }
    s2 = add32(r.s[2], e[2]);
    if (s2 > S3A_WORST) {
        t0 = add32(s4, tp[4 * 6 + 4]);
        t1 = add32(s3, tp[3 * 6 + 4]);
        t2 = add32(s2, tp[2 * 6 + 4]);
        PICK3(s4, t0, t1, t2, r.h[4] = r.h[3], r.h[4] = r.h[2]);
        if (s4 < S3A_WORST) s4 = S3A_WORST;
        if (s4 > best) best = s4;
        r.s[4] = s4;
    }
    s1 = add32(r.s[1], e[1]);
    if (s1 > S3A_WORST) {
        t0 = add32(s3, tp[3 * 6 + 3]);
        t1 = add32(s2, tp[2 * 6 + 3]);
        t2 = add32(s1, tp[1 * 6 + 3]);
        PICK3(s3, t0, t1, t2, r.h[3] = r.h[2], r.h[3] = r.h[1]);
        if (s3 < S3A_WORST) s3 = S3A_WORST;
        if (s3 > best) best = s3;
        r.s[3] = s3;
    }
    s0 = add32(r.s[0], e[0]);
    t0 = add32(s2, tp[2 * 6 + 2]);
    t1 = add32(s1, tp[1 * 6 + 2]);
    t2 = add32(s0, tp[0 * 6 + 2]);
    PICK3(s2, t0, t1, t2, r.h[2] = r.h[1], r.h[2] = r.h[0]);
    if (s2 < S3A_WORST) s2 = S3A_WORST;
    if (s2 > best) best = s2;
    r.s[2] = s2;

    t0 = add32(s1, tp[1 * 6 + 1]);
    t1 = add32(s0, tp[0 * 6 + 1]);
    if (t0 > t1) s1 = t0;
    else { s1 = t1; r.h[1] = r.h[0]; }
    if (s1 < S3A_WORST) s1 = S3A_WORST;
    if (s1 > best) best = s1;
    r.s[1] = s1;

    s0 = add32(s0, tp[0]);
    if (s0 < S3A_WORST) s0 = S3A_WORST;
    if (s0 > best) best = s0;
    r.s[0] = s0;
    return best;
}


#endif
