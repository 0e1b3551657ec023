/* s3a_scan.h -- one-workgroup exclusive prefix sum used by the ordered compactions. */
#ifndef S3A_SCAN_H
#define S3A_SCAN_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#ifndef SCAN_THREADS
#define SCAN_THREADS 1024
#endif

/* exclusive scan of src[0..n) into dst[0..n) (may alias) by one workgroup; the total goes to *total */
__device__ __forceinline__ void
block_exclusive_scan_to(const int32_t *src, int32_t *dst, int32_t n, int32_t *total)
{
    __shared__ int32_t wsum[SCAN_THREADS / 64];
    __shared__ int32_t carry;
    const int32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int32_t base = 0; base < n; base += SCAN_THREADS) {
        int32_t i = base + tid;
        int32_t x = (i < n) ? src[i] : 0;
        int32_t incl = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int32_t y = __shfl_up(incl, o, 64);
            if (lane >= o) incl += y;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        if (wave == 0) {
            int32_t w = (lane < SCAN_THREADS / 64) ? wsum[lane] : 0;
            int32_t wi = w;
#pragma unroll
            for (int o = 1; o < SCAN_THREADS / 64; o <<= 1) {
                int32_t y = __shfl_up(wi, o, 64);
                if (lane >= o) wi += y;
            }
            if (lane < SCAN_THREADS / 64) wsum[lane] = wi - w;  /* exclusive wave offsets */
        }
        __syncthreads();
        int32_t excl = carry + wsum[wave] + incl - x;
        if (i < n) dst[i] = excl;
        __syncthreads();
        if (tid == SCAN_THREADS - 1) carry = excl + x;
        __syncthreads();
    }
    if (tid == 0) *total = carry;
}

__device__ __forceinline__ void
block_exclusive_scan(int32_t *v, int32_t n, int32_t *total)
{
    block_exclusive_scan_to(v, v, n, total);
}

#endif
