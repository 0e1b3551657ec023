/*
 * s3a_psms.hip -- device half of pocketsphinx's continuous scorer (ps_mgaufuncs_t "ms",
 * pocketsphinx/src/libpocketsphinx/ms_mgau.c:163-252): float32 everywhere, log-domain precisions,
 * 8-bit mixture weights, int16 negated scores normalised to best = 0.
 *
 *   k_ps_dist    lane = (codebook, stream, density): dval = det - sum((x-m)^2 * prec), each product
 *                and the subtraction rounded to float32 (the reference's single C expression on SSE2,
 *                no FMA); the top-N list is in DESCENDING order and a tie goes in front of the entry
 *                it ties with (compute_dist's insertion, ms_gauden.c:497-519), so a density's slot is
 *                its rank under (value desc, codeword desc); parameters transposed to [dim][P] as in
 *                s3a_ms.hip
 *   k_ps_senone  thread = active senone: per stream ((int32)dist + 1023 >> 10) - weight8, log-add on
 *                the 10-bit-shifted table, negated sum, / aw, int16 clamp; block MIN -> atomicMin
 *   k_ps_norm    clamp(score - best) into the int16 output
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <vector>

#include "s3a_device.h"

#define PSB 256
#define PS_SHIFT 10

struct s3a_ps_dev_s {
    int32_t P;
    float *meanT, *precT, *det;
    float *meanS, *precS; int32_t VP;       /* [codebook][density][veclen padded to 4]: k_ps_cont_tr (one feature stream) */
    int32_t *featlen, *featoff, *pdf, *mgau;
    uint32_t *tab;
    uint32_t tab_size;
    int32_t lm_zero;
    uint8_t *sen_active, *mgau_active;
    float *feat, *dist;
    int32_t *dist_id, *scr, *best;
    int16_t *out, *out_h;
    hipStream_t stream;
    float *bdist; int32_t *bdist_id; size_t bdist_cap;     /* s3a_ps_score_slots_dev's top-N lists */
};

__global__ void
k_ps_mark(const uint8_t *__restrict__ sen_active, const int32_t *__restrict__ mgau, int32_t n_sen, uint8_t *mgau_active)
{
    const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_sen && sen_active[s]) mgau_active[mgau[s]] = 1;
}

__global__ void __launch_bounds__(PSB)
k_ps_dist(int32_t n_mgau, int32_t n_feat, int32_t nd, int32_t P, int32_t veclen, int32_t topn,
          const int32_t *__restrict__ featlen, const int32_t *__restrict__ featoff,
          const float *__restrict__ meanT, const float *__restrict__ precT, const float *__restrict__ det,
          const uint8_t *__restrict__ mgau_active, const float *__restrict__ feat, float *dist, int32_t *dist_id)
{
    extern __shared__ float x_s[];
    __shared__ float dv[PSB];
    for (int32_t i = threadIdx.x; i < veclen; i += PSB) x_s[i] = feat[i];
    __syncthreads();
    const int32_t item = blockIdx.x * PSB + threadIdx.x;
    const int32_t job = item / P, d = item % P, m = job / n_feat, f = job % n_feat;
    const bool live = job < n_mgau * n_feat && mgau_active[m] != 0;
    float dval = 0.0f;
    if (live && d < nd) {
        const int32_t flen = featlen[f], fo = featoff[f];
        const size_t base = ((size_t)m * veclen + fo) * P;
        dval = det[(size_t)job * P + d];
        for (int32_t i = 0; i < flen; i++) {
            const float df = x_s[fo + i] - meanT[base + (size_t)i * P + d];
            const float t = (df * df) * precT[base + (size_t)i * P + d];
            dval = dval - t;
        }
    }
    dv[threadIdx.x] = dval;
    __syncthreads();
    if (!live || d >= nd) return;
    int32_t rank = d;
    if (topn < nd) {
        const float *mine = dv + (threadIdx.x - d);
        rank = 0;
        for (int32_t k = 0; k < nd; k++) {
            const float o = mine[k];
            rank += (o > dval || (o == dval && k > d)) ? 1 : 0;
        }
        if (rank >= topn) return;
    }
    const size_t o = (size_t)job * topn + rank;
    dist[o] = dval;
    dist_id[o] = d;
}

struct LogAddShifted {
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

__global__ void __launch_bounds__(PSB)
k_ps_senone(int32_t n_sen, int32_t n_feat, int32_t nd, int32_t topn, int32_t aw,
            const uint8_t *__restrict__ sen_active, const int32_t *__restrict__ mgau,
            const int32_t *__restrict__ pdf, const float *__restrict__ dist, const int32_t *__restrict__ dist_id,
            LogAddShifted la, int32_t *scr, int32_t *best)
{
    __shared__ int32_t red[PSB / 64];
    const int32_t s = blockIdx.x * PSB + threadIdx.x;
    int32_t v = INT_MAX;
    if (s < n_sen && sen_active[s]) {
        const int32_t m = mgau[s];
        int32_t tot = 0;
        for (int32_t f = 0; f < n_feat; f++) {
            const float *fd = dist + ((size_t)m * n_feat + f) * topn;
            const int32_t *fi = dist_id + ((size_t)m * n_feat + f) * topn;
            const int32_t *p = pdf + ((size_t)s * n_feat + f) * nd;
            int32_t fscr = (((int32_t)fd[0] + ((1 << PS_SHIFT) - 1)) >> PS_SHIFT) - p[fi[0]];
            for (int32_t t = 1; t < topn; t++)
                fscr = la(fscr, (((int32_t)fd[t] + ((1 << PS_SHIFT) - 1)) >> PS_SHIFT) - p[fi[t]]);
            tot -= fscr;
        }
        tot /= aw;
        tot = min(max(tot, -32768), 32767);
        scr[s] = tot;
        v = tot;
    }
    int32_t b = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b = min(b, __shfl_xor(b, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < PSB / 64; w++) b = min(b, red[w]);
        if (b != INT_MAX) atomicMin(best, b);
    }
}

__global__ void
k_ps_norm(int32_t n_sen, const uint8_t *__restrict__ sen_active, const int32_t *__restrict__ best,
          const int32_t *__restrict__ scr, int16_t *out)
{
    const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_sen && sen_active[s]) out[s] = (int16_t)min(max(scr[s] - *best, -32768), 32767);
}


/* ------------------------------------------------------------------ */
/* many frames at once for the whole-utterance search (s3a_psfwd.hip): slot q scores feature row slot_row[q]  */
/* (negative: empty slot) for EVERY senone; out[q][n_sen] = the int16-clamped score BEFORE normalisation      */
/* (ms_mgau.c:185-189 / :228-233 normalise by the best of the senones the frame's active list names; the      */
/* search does that per frame from these).  A senone's score does not depend on which others are computed.    */
/* ------------------------------------------------------------------ */
#define PS_FT 8     /* frames per tile: a thread keeps its density's partial sums of PS_FT frames */
typedef float ps_f2 __attribute__((ext_vector_type(2)));
typedef float ps_f4 __attribute__((ext_vector_type(4)));
/* thread = (codebook, stream, density), PS_FT frames at once: the density's mean / precision are read ONCE per tile, the
 * frames' components come as two 16-byte LDS broadcasts per dimension, and the four operations of a dimension
 * (x - m, squared, times the precision, subtracted from the sum: each rounded to float32 as the reference's scalar SSE2
 * code rounds them, never fused) run two frames wide (packed float32 VALU) */
__global__ void __launch_bounds__(PSB)
k_ps_dist_slots(int32_t n_mgau, int32_t n_feat, int32_t nd, int32_t P, int32_t veclen, int32_t topn,
                const int32_t *__restrict__ featlen, const int32_t *__restrict__ featoff,
                const float *__restrict__ meanT, const float *__restrict__ precT, const float *__restrict__ det,
                const float *__restrict__ feat, const int32_t *__restrict__ slot_row, int32_t slot0, int32_t n_slots,
                float *dist, int32_t *dist_id)
{
    extern __shared__ ps_f4 x_s4[];         /* [veclen][PS_FT] floats: a dimension's PS_FT frames are contiguous */
    float *x_s = (float *)x_s4;
    __shared__ float dv[PS_FT][PSB];
    const int32_t q0 = slot0 + blockIdx.y * PS_FT;
    for (int32_t i = threadIdx.x; i < PS_FT * veclen; i += PSB) {
        const int32_t t = i / veclen, k = i - t * veclen, q = q0 + t;
        const int32_t row = (q < slot0 + n_slots) ? slot_row[q] : -1;
        x_s[k * PS_FT + t] = row >= 0 ? feat[(size_t)row * veclen + k] : 0.0f;
    }
    __syncthreads();
    const int32_t item = blockIdx.x * PSB + threadIdx.x;
    const int32_t job = item / P, d = item % P, m = job / n_feat, f = job % n_feat;
    const bool live = job < n_mgau * n_feat && d < nd;
    ps_f2 acc[PS_FT / 2];
#pragma unroll
    for (int t = 0; t < PS_FT / 2; t++) acc[t] = ps_f2{ 0.0f, 0.0f };
    if (live) {
        const int32_t flen = featlen[f], fo = featoff[f];
        const size_t base = ((size_t)m * veclen + fo) * P;
        const float dt = det[(size_t)job * P + d];
#pragma unroll
        for (int t = 0; t < PS_FT / 2; t++) acc[t] = ps_f2{ dt, dt };
        for (int32_t i = 0; i < flen; i++) {
            const float mu = meanT[base + (size_t)i * P + d], pr = precT[base + (size_t)i * P + d];
            const ps_f2 mu2 = ps_f2{ mu, mu }, pr2 = ps_f2{ pr, pr };
            const ps_f4 xa = x_s4[(fo + i) * (PS_FT / 4)], xb = x_s4[(fo + i) * (PS_FT / 4) + 1];
            const ps_f2 x[PS_FT / 2] = { xa.xy, xa.zw, xb.xy, xb.zw };
#pragma unroll
            for (int t = 0; t < PS_FT / 2; t++) {
                const ps_f2 df = x[t] - mu2;
                const ps_f2 sq = df * df;
                const ps_f2 tt = sq * pr2;
                acc[t] = acc[t] - tt;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < PS_FT / 2; t++) { dv[2 * t][threadIdx.x] = acc[t].x; dv[2 * t + 1][threadIdx.x] = acc[t].y; }
    __syncthreads();
    if (!live) return;
    for (int t = 0; t < PS_FT; t++) {
        const int32_t q = q0 + t;
        if (q >= slot0 + n_slots) break;
        const float dval = dv[t][threadIdx.x];
        int32_t rank = d;
        if (topn < nd) {
            const float *mine = &dv[t][threadIdx.x - d];
            rank = 0;
            for (int32_t k = 0; k < nd; k++) {
                const float o = mine[k];
                rank += (o > dval || (o == dval && k > d)) ? 1 : 0;
            }
            if (rank >= topn) continue;
        }
        const size_t o = ((size_t)(q - slot0) * n_mgau * n_feat + job) * topn + rank;
        dist[o] = dval;
        dist_id[o] = d;
    }
}

__global__ void __launch_bounds__(PSB)
k_ps_senone_slots(int32_t n_sen, int32_t n_mgau, int32_t n_feat, int32_t nd, int32_t topn, int32_t aw,
                  const int32_t *__restrict__ mgau, const int32_t *__restrict__ pdf, const float *__restrict__ dist,
                  const int32_t *__restrict__ dist_id, LogAddShifted la, const int32_t *__restrict__ slot_row,
                  int32_t slot0, int16_t *out)
{
    const int32_t s = blockIdx.x * PSB + threadIdx.x, q = slot0 + blockIdx.y;
    if (s >= n_sen || slot_row[q] < 0) return;
    const int32_t m = mgau[s];
    int32_t tot = 0;
    for (int32_t f = 0; f < n_feat; f++) {
        const size_t o = ((size_t)blockIdx.y * n_mgau * n_feat + (size_t)m * n_feat + f) * topn;
        const float *fd = dist + o;
        const int32_t *fi = dist_id + o;
        const int32_t *p = pdf + ((size_t)s * n_feat + f) * nd;
        int32_t fscr = (((int32_t)fd[0] + ((1 << PS_SHIFT) - 1)) >> PS_SHIFT) - p[fi[0]];
        for (int32_t t = 1; t < topn; t++)
            fscr = la(fscr, (((int32_t)fd[t] + ((1 << PS_SHIFT) - 1)) >> PS_SHIFT) - p[fi[t]]);
        tot -= fscr;
    }
    tot /= aw;
    out[(size_t)q * n_sen + s] = (int16_t)min(max(tot, -32768), 32767);
}

/*
 * Continuous models (-senmgau .cont.: senone s has codebook s, one stream): distances, top-N and the senone's score in
 * ONE kernel -- the top-N lists never leave the workgroup (the two-kernel path writes and re-reads 8 bytes per
 * (slot, codebook, rank): 196 KB per frame of a 6144 x 8 model, more than the model itself).  A tile is PS_CT frames: the
 * Gaussians' parameters are read once per PS_CT frames (PS_CT = 32: the 15.7 MB model once per 0.32 s of audio).
 * Arithmetic as k_ps_dist_slots + k_ps_senone_slots, value for value.  (Measured round 4: feeding the frames' components
 * through scalar loads instead of LDS broadcasts is slower, 109 vs 96 ms per 290 k frames.)
 */
#define PS_CT 32
#define PS_CH 8         /* frames ranked and scored at a time */
__global__ void __launch_bounds__(PSB)
k_ps_cont_slots(int32_t n_sen, int32_t nd, int32_t P, int32_t veclen, int32_t topn, int32_t aw,
                const float *__restrict__ meanT, const float *__restrict__ precT, const float *__restrict__ det,
                const int32_t *__restrict__ pdf, LogAddShifted la, const float *__restrict__ feat,
                const int32_t *__restrict__ slot_row, int32_t n_slots, int16_t *out)
{
    extern __shared__ ps_f4 x_s4[];         /* [veclen][PS_CT] */
    float *x_s = (float *)x_s4;
    __shared__ float dv[PS_CH][PSB];
    __shared__ float topv[PS_CH][PSB / 2][4];       /* (P >= 2: at most PSB / 2 codebooks per block; top-N <= 4) */
    __shared__ int32_t topi[PS_CH][PSB / 2][4];
    __shared__ int32_t s_row[PS_CT];
    const int32_t q0 = blockIdx.y * PS_CT;
    if (threadIdx.x < PS_CT) s_row[threadIdx.x] = (q0 + (int32_t)threadIdx.x < n_slots) ? slot_row[q0 + threadIdx.x] : -1;
    __syncthreads();
    for (int32_t i = threadIdx.x; i < PS_CT * veclen; i += PSB) {
        const int32_t t = i / veclen, k = i - t * veclen, row = s_row[t];
        x_s[k * PS_CT + t] = row >= 0 ? feat[(size_t)row * veclen + k] : 0.0f;
    }
    __syncthreads();
    const int32_t CB = PSB / P, item = blockIdx.x * PSB + threadIdx.x, m = item / P, d = item % P, cbl = threadIdx.x / P;
    const bool live = m < n_sen && d < nd;
    ps_f2 acc[PS_CT / 2];
    {
        const float dt = live ? det[(size_t)m * P + d] : 0.0f;
#pragma unroll
        for (int t = 0; t < PS_CT / 2; t++) acc[t] = ps_f2{ dt, dt };
    }
    if (live) {
        const size_t base = (size_t)m * veclen * P;
        for (int32_t i = 0; i < veclen; i++) {
            const float mu = meanT[base + (size_t)i * P + d], pr = precT[base + (size_t)i * P + d];
            const ps_f2 mu2 = ps_f2{ mu, mu }, pr2 = ps_f2{ pr, pr };
#pragma unroll
            for (int g = 0; g < PS_CT / 4; g++) {
                const ps_f4 xv = x_s4[i * (PS_CT / 4) + g];
                ps_f2 df = xv.xy - mu2, sq = df * df, tt = sq * pr2;
                acc[2 * g] = acc[2 * g] - tt;
                df = xv.zw - mu2; sq = df * df; tt = sq * pr2;
                acc[2 * g + 1] = acc[2 * g + 1] - tt;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < PS_CT / PS_CH; c++) {
#pragma unroll
        for (int t = 0; t < PS_CH / 2; t++) { dv[2 * t][threadIdx.x] = acc[c * (PS_CH / 2) + t].x; dv[2 * t + 1][threadIdx.x] = acc[c * (PS_CH / 2) + t].y; }
        __syncthreads();
        if (live) {
            for (int t = 0; t < PS_CH; t++) {
                const float dval = dv[t][threadIdx.x];
                int32_t rank = d;
                if (topn < nd) {
                    const float *mine = &dv[t][threadIdx.x - d];
                    rank = 0;
                    for (int32_t k = 0; k < nd; k++) {
                        const float o = mine[k];
                        rank += (o > dval || (o == dval && k > d)) ? 1 : 0;
                    }
                }
                if (rank < topn) { topv[t][cbl][rank] = dval; topi[t][cbl][rank] = d; }
            }
        }
        __syncthreads();
        /* the senone's score: one (codebook, frame) per thread */
        for (int32_t pr_ = threadIdx.x; pr_ < CB * PS_CH; pr_ += PSB) {
            const int32_t t = pr_ / CB, cb = pr_ - t * CB, sm = blockIdx.x * CB + cb, row = s_row[c * PS_CH + t];
            if (sm >= n_sen || row < 0) continue;
            const int32_t *pw = pdf + (size_t)sm * nd;
            int32_t fscr = (((int32_t)topv[t][cb][0] + ((1 << PS_SHIFT) - 1)) >> PS_SHIFT) - pw[topi[t][cb][0]];
            for (int32_t r = 1; r < topn; r++)
                fscr = la(fscr, (((int32_t)topv[t][cb][r] + ((1 << PS_SHIFT) - 1)) >> PS_SHIFT) - pw[topi[t][cb][r]]);
            int32_t tot = -fscr;
            tot /= aw;
            out[(size_t)(q0 + c * PS_CH + t) * n_sen + sm] = (int16_t)min(max(tot, -32768), 32767);
        }
        __syncthreads();
    }
}

/*
 * The same scores with the roles turned round: a LANE IS A FRAME (PT_FL frames per lane, two per packed float32 operation), the
 * Gaussian is uniform over the wave.  k_ps_cont_slots gives every lane a Gaussian and broadcasts the frames' components from LDS:
 * one 16-byte LDS read per eight packed operations on each of four SIMDs is all the LDS delivers, and the kernel stops at 0.29 of
 * the packed-float32 rate.  Here a lane keeps its frames' vectors in VGPRs for the whole launch (PT_FL x veclen registers), the
 * means and precisions come through the scalar cache (a Gaussian's parameters are contiguous: meanS / precS = [codebook][density]
 * [veclen padded to 4]) and the inner loop touches no memory: per dimension and frame pair subtract, square, scale, accumulate.
 * A lane then holds all densities of the codebook for its frames: the rank of a density (value descending, a tie in front of the
 * entry it ties with: codeword descending) is counted in registers, the top-N mixture is summed in that order, and the int16
 * scores leave through an LDS tile [frame][senone] so that a frame's row is written 128 bytes at a time.
 * Arithmetic: k_ps_cont_slots's, value for value (same operations on the same operands in the same order).
 */
#ifndef PT_FL
#define PT_FL 2         /* frames per lane: one packed pair (78 registers of features: four waves per SIMD hide the latency of the dependent operations) */
#endif
#define PT_SG 64        /* senones per workgroup (the tile's row: 128 bytes of int16) */
template <int ND, int VL>
__global__ void __launch_bounds__(PSB)
k_ps_cont_tr(int32_t n_sen, int32_t P, int32_t VP, int32_t topn, int32_t aw,
             const float *__restrict__ meanS, const float *__restrict__ precS, const float *__restrict__ det,
             const int32_t *__restrict__ pdf, LogAddShifted la, const float *__restrict__ feat,
             const int32_t *__restrict__ slot_row, int32_t n_slots, int16_t *out)
{
    constexpr int TF = 64 * PT_FL;                  /* frames of a workgroup: all four waves take the same frames, other senones */
    __shared__ int16_t tile[TF][PT_SG + 2];
    __shared__ float dvs[ND][PT_FL][PSB];          /* a thread's densities of the senone in hand (private slots) */
    __shared__ int32_t s_row[TF];
    const int32_t q0 = blockIdx.y * TF, s0 = blockIdx.x * PT_SG, lane = threadIdx.x & 63,
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     /* (uniform: the parameter addresses below are scalar) */
    for (int32_t i = threadIdx.x; i < TF; i += PSB) s_row[i] = (q0 + i < n_slots) ? slot_row[q0 + i] : -1;
    __syncthreads();
    /* this lane's frames: lane + 64 j, j < PT_FL */
    ps_f2 x[PT_FL / 2][VL];                         /* frames (lane + 128 j, lane + 128 j + 64) side by side */
#pragma unroll
    for (int j = 0; j < PT_FL / 2; j++) {
        const int32_t r0 = s_row[lane + 128 * j], r1 = s_row[lane + 128 * j + 64];
#pragma unroll
        for (int i = 0; i < VL; i++) x[j][i] = ps_f2{ r0 >= 0 ? feat[(size_t)r0 * VL + i] : 0.0f, r1 >= 0 ? feat[(size_t)r1 * VL + i] : 0.0f };      /* (a cast of (a, b) would splat b) */
    }
    for (int32_t sl = wave; sl < PT_SG; sl += PSB / 64) {
        const int32_t m = s0 + sl;                  /* (uniform over the wave) */
        if (m >= n_sen) break;
        /* one Gaussian at a time (the loop over the densities stays a loop: eight Gaussians' parameters unrolled side by side
         * want 640 scalar registers), its parameters eight dimensions ahead of the arithmetic: the scalar loads of chunk c + 1
         * are issued before chunk c is computed (the scheduling barriers keep them there) */
#pragma unroll 1
        for (int d = 0; d < ND; d++) {
            const float *mu = meanS + ((size_t)m * ND + d) * VP, *pr = precS + ((size_t)m * ND + d) * VP;
            const float dt = det[(size_t)m * P + d];
            ps_f2 acc[PT_FL / 2];
#pragma unroll
            for (int j = 0; j < PT_FL / 2; j++) acc[j] = ps_f2{ dt, dt };
            constexpr int NC = (VL + 7) / 8;
            float pm[2][8], pp[2][8];
#pragma unroll
            for (int k = 0; k < 8; k++) { pm[0][k] = mu[k]; pp[0][k] = pr[k]; }       /* (VP >= 8) */
#pragma unroll
            for (int c = 0; c < NC; c++) {
                if (c + 1 < NC) {
#pragma unroll
                    for (int k = 0; k < 8; k++) if ((c + 1) * 8 + k < ((VL + 3) & ~3)) { pm[(c + 1) & 1][k] = mu[(c + 1) * 8 + k]; pp[(c + 1) & 1][k] = pr[(c + 1) * 8 + k]; }
                }
                __builtin_amdgcn_sched_barrier(0);
                /* the chunk in phases over independent values (a dependent chain of four operations per value would wait for
                 * the pipeline at every step); only the accumulation keeps its order: dimension by dimension */
                ps_f2 t[8][PT_FL / 2];
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (c * 8 + k < VL) {
                        const float mi = pm[c & 1][k];
                        const ps_f2 mu2 = ps_f2{ mi, mi };
#pragma unroll
                        for (int j = 0; j < PT_FL / 2; j++) t[k][j] = x[j][c * 8 + k] - mu2;
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (c * 8 + k < VL) {
#pragma unroll
                        for (int j = 0; j < PT_FL / 2; j++) t[k][j] = t[k][j] * t[k][j];
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (c * 8 + k < VL) {
                        const float pi = pp[c & 1][k];
                        const ps_f2 pr2 = ps_f2{ pi, pi };
#pragma unroll
                        for (int j = 0; j < PT_FL / 2; j++) t[k][j] = t[k][j] * pr2;
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (c * 8 + k < VL) {
#pragma unroll
                        for (int j = 0; j < PT_FL / 2; j++) acc[j] = acc[j] - t[k][j];
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < PT_FL / 2; j++) { dvs[d][2 * j][threadIdx.x] = acc[j].x; dvs[d][2 * j + 1][threadIdx.x] = acc[j].y; }
        }
        float dval[ND][PT_FL];                      /* (a thread reads back what it wrote: no barrier) */
#pragma unroll
        for (int d = 0; d < ND; d++)
#pragma unroll
            for (int j = 0; j < PT_FL; j++) dval[d][j] = dvs[d][j][threadIdx.x];
        int32_t pw[ND];
#pragma unroll
        for (int d = 0; d < ND; d++) pw[d] = pdf[(size_t)m * ND + d];
#pragma unroll
        for (int j = 0; j < PT_FL; j++) {
            /* ranks, then the mixture over the ranks 0 .. topn - 1 in that order */
            int32_t rank[ND];
#pragma unroll
            for (int d = 0; d < ND; d++) {
                int32_t r = 0;
#pragma unroll
                for (int k = 0; k < ND; k++) r += (dval[k][j] > dval[d][j] || (dval[k][j] == dval[d][j] && k > d)) ? 1 : 0;
                rank[d] = topn < ND ? r : d;
            }
            int32_t fscr = 0;
            for (int32_t r = 0; r < topn; r++) {
                float v = 0.0f;
                int32_t w = 0;
#pragma unroll
                for (int d = 0; d < ND; d++) if (rank[d] == r) { v = dval[d][j]; w = pw[d]; }
                const int32_t term = (((int32_t)v + ((1 << PS_SHIFT) - 1)) >> PS_SHIFT) - w;
                fscr = r == 0 ? term : la(fscr, term);
            }
            int32_t tot = -fscr;
            tot /= aw;
            tile[lane + 64 * j][sl] = (int16_t)min(max(tot, -32768), 32767);
        }
    }
    __syncthreads();
    /* a frame's row of the tile: PT_SG int16 = 128 bytes, four bytes per thread, 32 threads per row */
    const int32_t ns = n_sen - s0 < PT_SG ? n_sen - s0 : PT_SG;
    for (int32_t e = threadIdx.x; e < TF * (PT_SG / 2); e += PSB) {
        const int32_t fr = e / (PT_SG / 2), c = (e - fr * (PT_SG / 2)) * 2;
        if (s_row[fr] < 0) continue;
        int16_t *o = out + (size_t)(q0 + fr) * n_sen + s0 + c;
        if (c + 1 < ns && ((uintptr_t)o & 3) == 0) *(int32_t *)o = (int32_t)(uint16_t)tile[fr][c] | ((int32_t)tile[fr][c + 1] << 16);
        else { if (c < ns) o[0] = tile[fr][c]; if (c + 1 < ns) o[1] = tile[fr][c + 1]; }
    }
}

/* internal (s3a_internal.h): feat_dev [rows][veclen], slot_row_dev [n_slots], raw_dev [n_slots][n_sen], on `stream` */
extern "C" int32_t
s3a_ps_score_slots_dev(s3a_ps_mgau_t *ps, const float *feat_dev, const int32_t *slot_row_dev, int32_t n_slots,
                       int16_t *raw_dev, void *stream)
{
    if (!ps || !ps->dev || n_slots < 0) return S3A_EINVAL;
    s3a_ps_dev_s *dv = ps->dev;
    hipStream_t st = (hipStream_t)stream;
    const int32_t M = ps->n_mgau, F = ps->n_feat, nd = ps->n_density, S = ps->n_sen, P = dv->P;
    const size_t per_slot = (size_t)M * F * ps->topn;
    /* the top-N lists of a tile of slots: at most 256 MB at a time */
    int32_t tile = (int32_t)(((size_t)256 << 20) / (per_slot * 8));
    tile = tile < PS_FT ? PS_FT : (tile / PS_FT) * PS_FT;
    if (tile > n_slots) tile = ((n_slots + PS_FT - 1) / PS_FT) * PS_FT;
    if (tile < PS_FT) tile = PS_FT;
    if (dv->bdist_cap < (size_t)tile * per_slot) {
        if (dv->bdist) (void)hipFree(dv->bdist);
        if (dv->bdist_id) (void)hipFree(dv->bdist_id);
        dv->bdist = NULL; dv->bdist_id = NULL; dv->bdist_cap = 0;
        HIPCHK(hipMalloc((void **)&dv->bdist, (size_t)tile * per_slot * 4));
        HIPCHK(hipMalloc((void **)&dv->bdist_id, (size_t)tile * per_slot * 4));
        dv->bdist_cap = (size_t)tile * per_slot;
    }
    LogAddShifted la = { dv->tab, dv->tab_size, dv->lm_zero };
    const int64_t items = (int64_t)M * F * P;
    if (ps->one_to_one && F == 1 && nd == 8 && ps->veclen == 39 && ps->topn <= nd && M == S && dv->meanS && !s3a_variants()->ps_score_by_gaussian) {
        /* (a grid's y dimension ends at 65 535: batches beyond 8.4 M frames go in several launches) */
        const int32_t per_launch = 65535 * 64 * PT_FL;
        for (int32_t s0 = 0; s0 < n_slots; s0 += per_launch) {
            const int32_t n = n_slots - s0 < per_launch ? n_slots - s0 : per_launch;
            hipLaunchKernelGGL((k_ps_cont_tr<8, 39>), dim3((S + PT_SG - 1) / PT_SG, (n + 64 * PT_FL - 1) / (64 * PT_FL)), dim3(PSB), 0, st, S, P,
                               dv->VP, ps->topn, ps->aw, dv->meanS, dv->precS, dv->det, dv->pdf, la, feat_dev, slot_row_dev + s0, n,
                               raw_dev + (size_t)s0 * S);
        }
        HIPCHK(hipGetLastError());
        return S3A_OK;
    }
    if (ps->one_to_one && F == 1 && P >= 2 && ps->topn <= 4 && ps->topn <= nd && M == S && (size_t)PS_CT * ps->veclen * 4 <= 32 * 1024) {
        const int32_t per_launch = 65535 * PS_CT;
        for (int32_t s0 = 0; s0 < n_slots; s0 += per_launch) {
            const int32_t n = n_slots - s0 < per_launch ? n_slots - s0 : per_launch;
            hipLaunchKernelGGL(k_ps_cont_slots, dim3((uint32_t)((items + PSB - 1) / PSB), (n + PS_CT - 1) / PS_CT), dim3(PSB),
                               (size_t)PS_CT * ps->veclen * 4 + 16, st, S, nd, P, ps->veclen, ps->topn, ps->aw, dv->meanT, dv->precT, dv->det,
                               dv->pdf, la, feat_dev, slot_row_dev + s0, n, raw_dev + (size_t)s0 * S);
        }
        HIPCHK(hipGetLastError());
        return S3A_OK;
    }
    for (int32_t s0 = 0; s0 < n_slots; s0 += tile) {
        const int32_t n = n_slots - s0 < tile ? n_slots - s0 : tile;
        hipLaunchKernelGGL(k_ps_dist_slots, dim3((uint32_t)((items + PSB - 1) / PSB), (n + PS_FT - 1) / PS_FT), dim3(PSB),
                           (size_t)PS_FT * ps->veclen * 4 + 16, st, M, F, nd, P, ps->veclen, ps->topn, dv->featlen, dv->featoff,
                           dv->meanT, dv->precT, dv->det, feat_dev, slot_row_dev, s0, n, dv->bdist, dv->bdist_id);
        hipLaunchKernelGGL(k_ps_senone_slots, dim3((S + PSB - 1) / PSB, n), dim3(PSB), 0, st, S, M, F, nd, ps->topn, ps->aw,
                           dv->mgau, dv->pdf, dv->bdist, dv->bdist_id, la, slot_row_dev, s0, raw_dev);
    }
    HIPCHK(hipGetLastError());
    return S3A_OK;
}

#define DM(ptr, bytes) HIPCHK(hipMalloc((void **)&(ptr), (bytes) ? (bytes) : 4))

extern "C" int32_t
s3a_ps_dev_create(s3a_ps_mgau_t *ps)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        s3a_set_error("no HIP device: libcmusphinx_amd has no CPU fallback");
        return S3A_ENODEV;
    }
    int32_t P = 1;
    while (P < ps->n_density) P <<= 1;
    if (P > PSB) { s3a_set_error("s3a_ps_ms_mgau_init: %d densities per codebook exceed the kernel's %d", ps->n_density, PSB); return S3A_EUNSUP; }
    if (ps->veclen * 4 > 48 * 1024) { s3a_set_error("s3a_ps_ms_mgau_init: feature vector too long"); return S3A_EUNSUP; }
    s3a_ps_dev_s *dv = new s3a_ps_dev_s();
    memset(dv, 0, sizeof *dv);
    ps->dev = dv;
    dv->P = P;
    const int32_t M = ps->n_mgau, F = ps->n_feat, nd = ps->n_density, D = ps->veclen, S = ps->n_sen;
    const size_t nT = (size_t)M * D * P;
    std::vector<float> mt(nT, 0.0f), pt(nT, 0.0f), dt((size_t)M * F * P, 0.0f);
    for (int32_t m = 0; m < M; m++)
        for (int32_t f = 0; f < F; f++) {
            const size_t src = (size_t)m * nd * D + (size_t)nd * ps->featoff[f];
            const size_t dst = ((size_t)m * D + ps->featoff[f]) * P;
            for (int32_t d = 0; d < nd; d++) {
                dt[((size_t)m * F + f) * P + d] = ps->det[((size_t)m * F + f) * nd + d];
                for (int32_t i = 0; i < ps->featlen[f]; i++) {
                    mt[dst + (size_t)i * P + d] = ps->mean[src + (size_t)d * ps->featlen[f] + i];
                    pt[dst + (size_t)i * P + d] = ps->prec[src + (size_t)d * ps->featlen[f] + i];
                }
            }
        }
    HIPCHK(hipStreamCreateWithFlags(&dv->stream, hipStreamNonBlocking));
    DM(dv->meanT, nT * 4); DM(dv->precT, nT * 4); DM(dv->det, (size_t)M * F * P * 4);
    DM(dv->featlen, (size_t)F * 4); DM(dv->featoff, (size_t)(F + 1) * 4);
    DM(dv->pdf, (size_t)S * F * nd * 4); DM(dv->mgau, (size_t)S * 4);
    DM(dv->sen_active, (size_t)S); DM(dv->mgau_active, (size_t)M); DM(dv->feat, (size_t)D * 4);
    DM(dv->dist, (size_t)M * F * ps->topn * 4); DM(dv->dist_id, (size_t)M * F * ps->topn * 4);
    DM(dv->scr, (size_t)S * 4); DM(dv->best, 4); DM(dv->out, (size_t)S * 2);
    HIPCHK(hipHostMalloc((void **)&dv->out_h, (size_t)S * 2));
    HIPCHK(hipMemcpy(dv->meanT, mt.data(), nT * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dv->precT, pt.data(), nT * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dv->det, dt.data(), dt.size() * 4, hipMemcpyHostToDevice));
    if (F == 1) {           /* a Gaussian's parameters contiguous, for the kernel that reads them through the scalar cache */
        const int32_t VP = (D + 3) & ~3;
        std::vector<float> ms((size_t)M * nd * VP + 64, 0.0f), pss((size_t)M * nd * VP + 64, 0.0f);
        for (int32_t m = 0; m < M; m++)
            for (int32_t d = 0; d < nd; d++)
                for (int32_t i = 0; i < D; i++) {
                    ms[((size_t)m * nd + d) * VP + i] = ps->mean[((size_t)m * nd + d) * D + i];
                    pss[((size_t)m * nd + d) * VP + i] = ps->prec[((size_t)m * nd + d) * D + i];
                }
        dv->VP = VP;
        DM(dv->meanS, ms.size() * 4); DM(dv->precS, pss.size() * 4);
        HIPCHK(hipMemcpy(dv->meanS, ms.data(), ms.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dv->precS, pss.data(), pss.size() * 4, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemcpy(dv->featlen, ps->featlen, (size_t)F * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dv->featoff, ps->featoff, (size_t)(F + 1) * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dv->pdf, ps->pdf, (size_t)S * F * nd * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dv->mgau, ps->mgau, (size_t)S * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dv->dist, 0, (size_t)M * F * ps->topn * 4));
    HIPCHK(hipMemset(dv->dist_id, 0, (size_t)M * F * ps->topn * 4));
    HIPCHK(hipMemset(dv->out, 0, (size_t)S * 2));
    {
        uint32_t size = 0, width = 0, shift = 0;
        s3a_logmath_get_table_shape(ps->lm8, &size, &width, &shift);
        if (size == 0) { s3a_set_error("s3a_ps_ms_mgau_init: no log-add table"); return S3A_EUNSUP; }
        std::vector<uint32_t> tab(size);
        s3a_logmath_copy_table(ps->lm8, tab.data(), size);
        dv->tab_size = size;
        dv->lm_zero = s3a_logmath_get_zero(ps->lm8);
        DM(dv->tab, (size_t)size * 4);
        HIPCHK(hipMemcpy(dv->tab, tab.data(), (size_t)size * 4, hipMemcpyHostToDevice));
    }
    return S3A_OK;
}

extern "C" void
s3a_ps_dev_destroy(s3a_ps_mgau_t *ps)
{
    s3a_ps_dev_s *dv = ps ? ps->dev : NULL;
    if (!dv) return;
    void *ptrs[] = { dv->meanS, dv->precS, dv->meanT, dv->precT, dv->det, dv->featlen, dv->featoff, dv->pdf, dv->mgau, dv->tab,
                     dv->sen_active, dv->mgau_active, dv->feat, dv->dist, dv->dist_id, dv->scr, dv->best, dv->out };
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (dv->bdist) (void)hipFree(dv->bdist);
    if (dv->bdist_id) (void)hipFree(dv->bdist_id);
    if (dv->out_h) (void)hipHostFree(dv->out_h);
    if (dv->stream) (void)hipStreamDestroy(dv->stream);
    delete dv;
    ps->dev = NULL;
}

extern "C" int32_t
s3a_ps_ms_cont_mgau_frame_eval(s3a_ps_mgau_t *ps, int16_t *senscr, const uint8_t *senone_active,
                               int32_t n_senone_active, const float *feat, int32_t frame, int32_t compallsen)
{
    (void)frame;
    if (!ps || !ps->dev || !senscr || !feat || (!compallsen && (!senone_active || n_senone_active < 0))) return S3A_EINVAL;
    s3a_ps_dev_s *dv = ps->dev;
    const int32_t M = ps->n_mgau, F = ps->n_feat, nd = ps->n_density, S = ps->n_sen, P = dv->P;
    /* the delta-encoded list (acmod_flags2list) -> flags; out-of-range ids would be a caller bug */
    if (compallsen) memset(ps->flags, 1, S);
    else {
        memset(ps->flags, 0, S);
        for (int32_t i = 0, n = 0; i < n_senone_active; i++) {
            const int32_t s = senone_active[i] + n;
            if (s >= S) { s3a_set_error("s3a_ps_ms_cont_mgau_frame_eval: senone %d out of range", s); return S3A_EINVAL; }
            ps->flags[s] = 1;
            n = s;
        }
    }
    const int32_t init = INT_MAX;
    HIPCHK(hipMemcpyAsync(dv->sen_active, ps->flags, (size_t)S, hipMemcpyHostToDevice, dv->stream));
    HIPCHK(hipMemcpyAsync(dv->feat, feat, (size_t)ps->veclen * 4, hipMemcpyHostToDevice, dv->stream));
    HIPCHK(hipMemcpyAsync(dv->best, &init, 4, hipMemcpyHostToDevice, dv->stream));
    const uint8_t *cb_active = dv->sen_active;
    if (!ps->one_to_one) {
        HIPCHK(hipMemsetAsync(dv->mgau_active, 0, (size_t)M, dv->stream));
        hipLaunchKernelGGL(k_ps_mark, dim3((S + 255) / 256), dim3(256), 0, dv->stream, dv->sen_active, dv->mgau, S, dv->mgau_active);
        cb_active = dv->mgau_active;
    }
    const int64_t items = (int64_t)M * F * P;
    hipLaunchKernelGGL(k_ps_dist, dim3((uint32_t)((items + PSB - 1) / PSB)), dim3(PSB), (size_t)ps->veclen * 4, dv->stream,
                       M, F, nd, P, ps->veclen, ps->topn, dv->featlen, dv->featoff, dv->meanT, dv->precT, dv->det,
                       cb_active, dv->feat, dv->dist, dv->dist_id);
    LogAddShifted la = { dv->tab, dv->tab_size, dv->lm_zero };
    hipLaunchKernelGGL(k_ps_senone, dim3((S + PSB - 1) / PSB), dim3(PSB), 0, dv->stream, S, F, nd, ps->topn, ps->aw,
                       dv->sen_active, dv->mgau, dv->pdf, dv->dist, dv->dist_id, la, dv->scr, dv->best);
    hipLaunchKernelGGL(k_ps_norm, dim3((S + 255) / 256), dim3(256), 0, dv->stream, S, dv->sen_active, dv->best, dv->scr, dv->out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(dv->out_h, dv->out, (size_t)S * 2, hipMemcpyDeviceToHost, dv->stream));
    HIPCHK(hipStreamSynchronize(dv->stream));
    for (int32_t s = 0; s < S; s++)
        if (ps->flags[s]) senscr[s] = dv->out_h[s];
    return S3A_OK;
}
