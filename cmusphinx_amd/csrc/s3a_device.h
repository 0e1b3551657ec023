/*
 * s3a_device.h -- C++/HIP-only internals shared by the .hip translation units:
 * the device half of a model, the HIP error macro and the device-side
 * log-add / Gaussian primitives.
 */
#ifndef S3A_DEVICE_H
#define S3A_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "s3a_internal.h"

#define HIPCHK(expr)                                                              \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess) {                                                   \
            s3a_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                          __FILE__, __LINE__);                                    \
            return (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice         \
                    || e_ == hipErrorInsufficientDriver) ? S3A_ENODEV : S3A_EHIP; \
        }                                                                         \
    } while (0)

struct s3a_mgau_dev_s {
    int32_t S, C, CP, D, D4, G, Gpad;
    float4 *mean4, *prec4;      /* [D4][Gpad] */
    float *lrd;                 /* [Gpad] */
    int32_t *mixw;              /* [Gpad] */
    uint16_t *tab16;            /* log-add table, width <= 2 */
    uint32_t *tab32;            /* log-add table, width 4 */
    uint32_t tab_size;
    int32_t lm_zero;
    int32_t *bstidx, *bstscr, *updatetime;  /* [S] */
    hipStream_t stream;
    /* scratch for the host-pointer API */
    float *feat_buf;   size_t feat_cap;     /* padded [T][D4*4] */
    int32_t *scr_buf;  size_t scr_cap;
    int32_t *best_buf; size_t best_cap;
    int32_t n_cu;
    hipEvent_t ev0, ev1;        /* s3a_stream_timer_* */
    /* the log-add table repacked for the frame-synchronous pass (k_score_frame_sync): 16-bit entries up to
     * hyb_head, 8-bit entries from there on (the values have dropped below 256): 38 KB instead of 58 */
    int32_t hyb_ok, hyb_head, hyb_bytes;
    uint8_t *hyb_tab;           /* device: [hyb_head uint16][tab_size - hyb_head uint8], padded to 16 bytes */
};


/* logmath_add, logmath.c:391-436, on a uint16 table in LDS or global memory */
struct LogAdd {
    const uint16_t *tab;
    uint32_t size;
    int32_t zero;
    __device__ __forceinline__ int32_t operator()(int32_t x, int32_t y) const
    {
        if (x <= zero) return y;
        if (y <= zero) return x;
        int32_t hi = x > y ? x : y;
        int32_t lo = x > y ? y : x;
        uint32_t d = (uint32_t)hi - (uint32_t)lo;
        if (d >= size) return hi;           /* also covers the wrapped (d < 0) case */
        return hi + (int32_t)tab[d];
    }
};

/* one dimension of cont_mgau.c:1058-1063, bit-exact (compile with -ffp-contract=off) */
__device__ __forceinline__ double
gau_step_exact(double acc, float x, float m, float p)
{
    float df = x - m;               /* float32 subtract */
    double d = (double)df;
    double d2 = d * d;              /* exact: 24-bit significand squared */
    double t = d2 * (double)p;      /* rounded once */
    return acc - t;                 /* rounded once; NOT an fma */
}

__device__ __forceinline__ float
gau_step_fast(float acc, float x, float m, float p)
{
    float df = x - m;
    return __builtin_fmaf(-(df * df), p, acc);
}

template <bool EXACT> struct Acc;
template <> struct Acc<true> {
    typedef double T;
    static __device__ __forceinline__ double step(double a, float x, float m, float p)
    { return gau_step_exact(a, x, m, p); }
};
template <> struct Acc<false> {
    typedef float T;
    static __device__ __forceinline__ float step(float a, float x, float m, float p)
    { return gau_step_fast(a, x, m, p); }
};

/* gauscr = (int32)(f * max(dval, distfloor)) + mixw : cont_mgau.c:1066-1073 */
__device__ __forceinline__ int32_t
gau_to_int(double dval, double f, double distfloor, int32_t mixw)
{
    if (dval < distfloor) dval = distfloor;
    return (int32_t)((uint32_t)(int32_t)(f * dval) + (uint32_t)mixw);
}

#define D4MAIN 10           /* ceil(39/4): the 1s_c_d_dd case gets the unrolled kernels */

int32_t s3a_dev_grow(void **buf, size_t *cap, size_t need);

#endif
