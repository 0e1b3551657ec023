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

/* hmm_vit_eval_3st_lr, hmm.c:592-674 */
template <typename H>
__device__ __forceinline__ int32_t
vit3(HmmRegsT<H> &r, const int32_t *tp, int32_t e0, int32_t e1, int32_t e2)
{
    int32_t s3, s2, s1, s0, t2, t1, t0, best;
    s2 = add32(r.s[2], e2);
    s1 = add32(r.s[1], e1);
    s0 = add32(r.s[0], e0);
    t0 = t1 = best = S3A_WORST;
    t2 = INT_MIN;
    if (s2 > S3A_WORST) { t1 = add32(s2, tp[2 * 4 + 3]); t0 = add32(s2, tp[2 * 4 + 2]); }
    if (s1 > S3A_WORST && tp[1 * 4 + 3] > S3A_WORST) t2 = add32(s1, tp[1 * 4 + 3]);
    if (t1 > t2) { s3 = t1; r.outh = r.h[2]; }
    else         { s3 = t2; r.outh = r.h[1]; }
    if (s3 < S3A_WORST) s3 = S3A_WORST;
    r.out = s3;
    best = s3;

    t1 = t2 = S3A_WORST;
    if (s1 > S3A_WORST) t1 = add32(s1, tp[1 * 4 + 2]);
    if (tp[0 * 4 + 2] > S3A_WORST) t2 = add32(s0, tp[0 * 4 + 2]);
    if (t0 > t1) {
        if (t2 > t0) { s2 = t2; r.h[2] = r.h[0]; } else s2 = t0;
    }
    else {
        if (t2 > t1) { s2 = t2; r.h[2] = r.h[0]; } else { s2 = t1; r.h[2] = r.h[1]; }
    }
    if (s2 < S3A_WORST) s2 = S3A_WORST;
    if (s2 > best) best = s2;
    r.s[2] = s2;

    t0 = t1 = S3A_WORST;
    if (s1 > S3A_WORST) t0 = add32(s1, tp[1 * 4 + 1]);
    if (s0 > S3A_WORST) t1 = add32(s0, tp[0 * 4 + 1]);
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

#define PICK3(dst, t0, t1, t2, on_t1, on_t2)     \
    do {                                         \
        if ((t0) > (t1)) {                       \
            if ((t2) > (t0)) { dst = (t2); on_t2; } else dst = (t0); \
        }                                        \
        else {                                   \
            if ((t2) > (t1)) { dst = (t2); on_t2; } else { dst = (t1); on_t1; } \
        }                                        \
    } while (0)

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
