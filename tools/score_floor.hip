// The floor of ONE-FRAME senone scoring on the hub4 shape (49 152 Gaussians x (39 means + 39 precisions + 2) x 4 B = 15.73 MB per frame),
// for profiles/r6_score_floor.txt: what a launch that re-reads the model for a single frame cannot go under on this GPU --
//   (1) an empty kernel, launches back to back: the launch boundary;
//   (2) the bare read: every lane 20 independent 16-byte loads (the scorer's own access shape, one Gaussian per lane), nothing else;
//   (3) the same + what a mixture's ordered log-add needs behind the last load: 8 DEPENDENT table look-ups (the 58.7 KB logs3 add table,
//       L2-resident), the value of one deciding the index of the next -- cont_mgau.c:1080-1122 adds the 8 densities of a senone in order;
//   (4) the same with the table in LDS (each workgroup copies the 14 680 words it may index: what TAB_LDS costs a one-frame launch).
// The north star's 0.60 of 8 TB/s is 3.28 us per frame for these bytes.   hipcc --offload-arch=gfx950 -O3 tools/score_floor.hip -o /tmp/score_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define TAB_N 14680
__global__ void k_empty(float *out) { if (out == (float *)1) out[0] = 0.f; }
template <int CHAIN, bool LDS>
__global__ void __launch_bounds__(256)
k_read20(const float4 *__restrict__ p, size_t plane, const int *__restrict__ tab, float *out)
{
    __shared__ int s_tab[LDS ? TAB_N : 1];
    if (LDS) { for (int i = threadIdx.x; i < TAB_N; i += 256) s_tab[i] = tab[i]; __syncthreads(); }
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    float4 v[20];
#pragma unroll
    for (int k = 0; k < 20; k++) v[k] = p[k * plane + g];
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 20; k++) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    int idx = (int)(__float_as_uint(acc) % TAB_N);
#pragma unroll
    for (int c = 0; c < CHAIN; c++) idx = (LDS ? s_tab[idx] : tab[idx]) % TAB_N;      // (dependent: the sum so far picks the next entry)
    if (acc == 123.456f || idx == -7) out[0] = acc + idx;
}
template <class F> static double timed(F launch, int reps)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / reps;
}
int main()
{
    const size_t G = 49152, plane = G, bytes = G * 80 * 4;
    float4 *p; hipMalloc(&p, bytes); hipMemset(p, 0, bytes);
    int *tab; hipMalloc(&tab, TAB_N * 4);
    { int *h = (int *)malloc(TAB_N * 4); for (int i = 0; i < TAB_N; i++) h[i] = (i * 7919 + 13) % TAB_N; hipMemcpy(tab, h, TAB_N * 4, hipMemcpyHostToDevice); free(h); }
    float *out; hipMalloc(&out, 4);
    const int reps = 200, blocks = (int)(plane / 256);
    const double e = timed([&] { hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, 0, out); }, reps);
    const double r0 = timed([&] { hipLaunchKernelGGL((k_read20<0, false>), dim3(blocks), dim3(256), 0, 0, p, plane, tab, out); }, reps);
    const double r8 = timed([&] { hipLaunchKernelGGL((k_read20<8, false>), dim3(blocks), dim3(256), 0, 0, p, plane, tab, out); }, reps);
    const double l8 = timed([&] { hipLaunchKernelGGL((k_read20<8, true>), dim3(blocks), dim3(256), 0, 0, p, plane, tab, out); }, reps);
    printf("model bytes per frame %zu (%.2f MB), %d workgroups of 256\n", bytes, bytes / 1048576.0, blocks);
    printf("(1) empty kernel, back to back            %6.2f us per launch\n", e);
    printf("(2) bare read, 20 x 16 B per lane         %6.2f us per launch = %5.0f GB/s = %.3f of 8 TB/s\n", r0, bytes / (r0 * 1e-6) / 1e9, bytes / (r0 * 1e-6) / 8e12);
    printf("(3) + 8 dependent look-ups (table in L2)  %6.2f us per launch = %5.0f GB/s = %.3f of 8 TB/s\n", r8, bytes / (r8 * 1e-6) / 1e9, bytes / (r8 * 1e-6) / 8e12);
    printf("(4) + 8 dependent look-ups (table in LDS) %6.2f us per launch = %5.0f GB/s = %.3f of 8 TB/s\n", l8, bytes / (l8 * 1e-6) / 1e9, bytes / (l8 * 1e-6) / 8e12);
    printf("the 0.60 mark: %.2f us per launch\n", bytes / (0.6 * 8e12) * 1e6);
    return 0;
}
