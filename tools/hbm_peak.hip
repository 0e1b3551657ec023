// Achievable streaming-read bandwidth on this GPU (context for the roofline fractions in DESIGN.md):
// every lane reads float4's with a grid-stride loop and folds them; sizes: one that fits the 256 MB
// infinity cache (the senone model does) and one that does not.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void __launch_bounds__(256)
k_read(const float4 *__restrict__ p, size_t n, float *out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}
// the scorer's shape: 20 independent 16-byte loads per lane, no loop
__global__ void __launch_bounds__(256)
k_read20(const float4 *__restrict__ p, size_t plane, float *out)
{
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    float4 v[20];
#pragma unroll
    for (int k = 0; k < 20; k++) v[k] = p[k * plane + g];
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 20; k++) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    if (acc == 123.456f) out[0] = acc;
}
int main()
{
    const size_t sizes[] = { 16u << 20, 82u << 20, 1024u << 20 };
    float *out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t s : sizes) {
        float4 *p; hipMalloc(&p, s); hipMemset(p, 0, s);
        for (int blocks : { 256, 1024, 4096, 16384 }) {
            const int reps = 50;
            hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, p, s / 16, out);
            hipEventRecord(e0);
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, p, s / 16, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("grid-stride read %5zu MB, %5d blocks: %7.2f us/launch  %6.0f GB/s\n", s >> 20, blocks, ms * 1e3 / reps, s / (ms * 1e-3 / reps) / 1e9);
        }
        {
            const size_t plane = s / 16 / 20 / 256 * 256;
            const int reps = 50;
            hipLaunchKernelGGL(k_read20, dim3(plane / 256), dim3(256), 0, 0, p, plane, out);
            hipEventRecord(e0);
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_read20, dim3(plane / 256), dim3(256), 0, 0, p, plane, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("20 loads per lane %5zu MB, %6zu blocks: %7.2f us/launch  %6.0f GB/s\n", s >> 20, plane / 256, ms * 1e3 / reps, plane * 20 * 16 / (ms * 1e-3 / reps) / 1e9);
        }
        hipFree(p);
    }
    return 0;
}
