/*
 * s3a_gated.h -- body of the gated per-frame senone kernel (approx_cont_mgau_frame_eval /
 * _ci_eval on the device, see s3a_scorer.hip) as a device function shared by the single-decoder
 * kernel and the batched one (s3a_batch.hip).  BX = workgroup index within one decoder's grid.
 */
#ifndef S3A_GATED_H
#define S3A_GATED_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include "s3a_device.h"

#pragma clang fp contract(off)

/*
 * D4C > 0 (the 39/40-dimensional case, D4C == D4): the lane's Gaussian is fetched BEFORE the gate is
 * known, next to the gate's own inputs.  The gate (active? CI score within the beam? best Gaussian
 * of the previous frame?) is a chain of dependent loads; with the parameter fetch behind it the
 * kernel was that chain plus the fetch plus the log-add chain, ~15 memory round trips for a few
 * microseconds of work.  A Gaussian's value does not depend on whether it is wanted, so it is computed
 * unconditionally and the gate only SELECTS (same scores, same counters); nearly every active senone
 * is inside the CI beam anyway (41 of 41.5 k Gaussians per frame on the hub4-shaped task).
 * The log-add table stays in global memory (L2 / vL1D): staging its 58 KB into LDS per workgroup was
 * measured at twice the kernel's time (13.8 vs 6.8 us for the CD phase of one decoder) -- it halves
 * the waves a CU can hold and every workgroup waits for the copy -- while most steps of the ordered
 * log-add skip the table anyway (|difference| beyond its end).
 */
template <bool EXACT, int D4C>
__device__ __forceinline__ void
d_gated_frame(const float4 *__restrict__ mean4, const float4 *__restrict__ prec4,
              const float *__restrict__ lrd, const int32_t *__restrict__ mixw_g,
              const uint16_t *__restrict__ tab_g, uint32_t tab_size, int32_t lm_zero,
              double f, double distfloor, const float *__restrict__ x, int32_t D4, int32_t CP,
              int32_t Gpad, int32_t sen_lo, int32_t sen_hi, int32_t ci_phase,
              const uint8_t *__restrict__ ncomp, const int16_t *__restrict__ cd2cisen,
              const uint8_t *__restrict__ sen_active, int32_t *__restrict__ senscr,
              int32_t pbest_plus_beam, const int32_t *__restrict__ pbest_ptr, int32_t beam,
              int32_t frame, int32_t is_skip,
              int32_t *bstidx, int32_t *bstscr, int32_t *updatetime, int32_t *misc, int32_t best_slot,
              uint8_t *clear_active, int32_t *gpart, int32_t gp_n,
              const int32_t BX)
{
    typedef typename Acc<EXACT>::T acc_t;
    constexpr int NK = D4C > 0 ? D4C : 1;
    const int32_t lane = threadIdx.x & 63;
    const int32_t g = sen_lo * CP + BX * 256 + threadIdx.x;
    const int32_t sen = g / CP, c = g - sen * CP, sl = lane / CP;
    const bool valid = sen < sen_hi;
    LogAdd la;
    la.tab = tab_g; la.size = tab_size; la.zero = lm_zero;

    /* ---- round trip 1: everything whose address is known now ---- */
    float4 M[NK], P[NK];
    float lrd_g = 0.0f;
    int32_t mixw = 0, act = 0, ci_id = 0, bi = S3A_NO_BSTIDX, ut = 0, nc = 0;
    if (valid) {
        if (D4C > 0) {
#pragma unroll
            for (int k = 0; k < NK; k++) {
                M[k] = mean4[(size_t)k * Gpad + g];
                P[k] = prec4[(size_t)k * Gpad + g];
            }
            lrd_g = lrd[g];
            mixw = mixw_g[g];
        }
        nc = (int32_t)ncomp[sen];
        if (ci_phase)
            act = 1;
        else {
            act = sen_active[sen];
            ci_id = cd2cisen[sen];
            bi = bstidx[sen];
            ut = updatetime[sen];
        }
    }
    else if (D4C > 0) {
#pragma unroll
        for (int k = 0; k < NK; k++) M[k] = P[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    /* device-resident path: the CI maximum was left in memory by the CI phase */
    if (pbest_ptr)
        pbest_plus_beam = (int32_t)((uint32_t)*pbest_ptr + (uint32_t)beam);
    /* ---- round trip 2: the CI senone's score ---- */
    /* 0 = untouched, 1 = full, 2 = single Gaussian, 3 = CI copy */
    int32_t mode = 0, ci_scr = 0;
    if (valid && act) {
        if (ci_phase)
            mode = 1;
        else {
            ci_scr = senscr[ci_id];
            if (ci_scr >= pbest_plus_beam)
                mode = 1;
            else
                mode = (bi == S3A_NO_BSTIDX || ut != frame - 1) ? 3 : 2;
        }
    }
    if (mode != 2) bi = S3A_NO_BSTIDX;
    const bool wanted = mode == 1 || (mode == 2 && c == bi);
    int32_t gs = S3A_LOGPROB_ZERO;
    if (D4C > 0) {
        acc_t a = (acc_t)lrd_g;
#pragma unroll
        for (int k = 0; k < NK; k++) {
            const float4 xv = *(const float4 *)(x + 4 * k);
            a = Acc<EXACT>::step(a, xv.x, M[k].x, P[k].x);
            a = Acc<EXACT>::step(a, xv.y, M[k].y, P[k].y);
            a = Acc<EXACT>::step(a, xv.z, M[k].z, P[k].z);
            a = Acc<EXACT>::step(a, xv.w, M[k].w, P[k].w);
        }
        if (wanted) gs = gau_to_int((double)a, f, distfloor, mixw);
    }
    else if (wanted) {
        acc_t a = (acc_t)lrd[g];
        for (int32_t k = 0; k < D4; k++) {
            float4 m = mean4[(size_t)k * Gpad + g], p = prec4[(size_t)k * Gpad + g];
            const float4 xv = *(const float4 *)(x + 4 * k);
            a = Acc<EXACT>::step(a, xv.x, m.x, p.x);
            a = Acc<EXACT>::step(a, xv.y, m.y, p.y);
            a = Acc<EXACT>::step(a, xv.z, m.z, p.z);
            a = Acc<EXACT>::step(a, xv.w, m.w, p.w);
        }
        gs = gau_to_int((double)a, f, distfloor, mixw_g[g]);
    }
    /* ordered chain over the senone's lanes; every lane of the senone runs it */
    int32_t score = S3A_LOGPROB_ZERO, bs = S3A_LOGPROB_ZERO, bidx = S3A_NO_BSTIDX;
    for (int32_t cc = 0; cc < CP; cc++) {
        int32_t v = __shfl(gs, sl * CP + cc, 64);
        if (mode == 1 && cc < nc) {
            score = la(score, v);
            if (v > bs) { bs = v; bidx = cc; }      /* update_best_id = 1: strict >, first max wins */
        }
        else if (mode == 2 && cc == bi) {
            score = la(score, v);
            if (v > bs) { bs = v; bidx = cc; }
        }
    }
    if (score <= S3A_LOGPROB_ZERO) score = S3A_LOGPROB_ZERO;
    if (mode == 3) score = ci_scr;

    /* fused decoder path: the mask is consumed here, leave it clean for the next frame's marks
     * (every lane of the senone has read it above; same wave, program order) */
    if (clear_active && valid && c == 0) clear_active[sen] = 0;
    int32_t wbest = INT_MIN, ns = 0, ng = 0;
    if (mode != 0 && c == 0) {
        senscr[sen] = score;
        wbest = score;
        if (mode == 1) {
            bstidx[sen] = bidx; bstscr[sen] = bs; updatetime[sen] = frame;
            ns = 1; ng = nc;
        }
        else if (mode == 2) {
            if (is_skip) { bstidx[sen] = bidx; bstscr[sen] = bs; updatetime[sen] = frame; }
            ng = 1;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        wbest = max(wbest, __shfl_xor(wbest, o, 64));
        ns += __shfl_xor(ns, o, 64);
        ng += __shfl_xor(ng, o, 64);
    }
    /* one set of atomics per WORKGROUP: the three words share a cache line, and read-modify-writes of
     * one line are served one after another (~14 ns each; one set per wave -- 768 waves -- made this
     * kernel 32 us long) */
    __shared__ int32_t red[3][4];
    if (lane == 0) { red[0][threadIdx.x >> 6] = wbest; red[1][threadIdx.x >> 6] = ns; red[2][threadIdx.x >> 6] = ng; }
    __syncthreads();
    if (threadIdx.x == 0) {
        wbest = max(max(red[0][0], red[0][1]), max(red[0][2], red[0][3]));
        ns = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        ng = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        if (gpart) {        /* fused decoder: this workgroup's column, merged by the consumers -- no atomics */
            gpart[BX] = wbest; gpart[gp_n + BX] = ns; gpart[2 * gp_n + BX] = ng;
            return;
        }
        if (wbest != INT_MIN) atomicMax(&misc[best_slot], wbest);
        if (!ci_phase && ns) atomicAdd(&misc[1], ns);
        if (!ci_phase && ng) atomicAdd(&misc[2], ng);
        if (ci_phase && ns) atomicAdd(&misc[3], ns);
        if (ci_phase && ng) atomicAdd(&misc[4], ng);
    }
}


#endif
