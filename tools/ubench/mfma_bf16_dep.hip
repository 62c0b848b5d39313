// MI355X: issue interval of bf16 MFMAs as a function of the number of independent accumulators they rotate over (one wave per SIMD):
// how far apart must two MFMAs on the same accumulator be?   hipcc --offload-arch=gfx950 -O3 mfma_bf16_dep.hip -o mfma_bf16_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int BIG>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    u32x4 a = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
    f32x4 acc[8]; f32x16 big[4];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) big[i][e] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            if (BIG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(big[m % NACC]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m % NACC]) : "v"(a), "v"(b));
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i].x;
    for (int i = 0; i < 4; ++i) r += big[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int NACC, int BIG>
void run() {
    int iters = 4000;
    float* out; (void)hipMalloc(&out, 256 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, BIG>), dim3(256), dim3(256), 0, 0, out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, BIG>), dim3(256), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%s rotating over %d accumulator(s): %.1f cycles@2.4GHz per MFMA\n", BIG ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_16x16x32_bf16", NACC, ms * 1e-3 * 2.4e9 / (24.0 * iters));
    (void)hipFree(out);
}
template <int NACC>
__global__ void __launch_bounds__(256) k16(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 a = {0x3f803f80u + lane, 0x3f803f80u}, b = a;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+a"(acc[m % NACC]) : "v"(a), "v"(b));
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i].x;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int NACC>
void run16() {
    int iters = 4000;
    float* out; (void)hipMalloc(&out, 256 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k16<NACC>), dim3(256), dim3(256), 0, 0, out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k16<NACC>), dim3(256), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("v_mfma_f32_16x16x16_bf16 (K = 16, 2-register operands) rotating over %d accumulator(s): %.1f cycles@2.4GHz per MFMA\n", NACC, ms * 1e-3 * 2.4e9 / (24.0 * iters));
    (void)hipFree(out);
}
int main() {
    run16<1>(); run16<4>();
    run<1, 0>(); run<2, 0>(); run<3, 0>(); run<4, 0>(); run<6, 0>(); run<8, 0>();
    run<1, 1>(); run<2, 1>(); run<3, 1>(); run<4, 1>();
    return 0;
}
