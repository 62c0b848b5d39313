// Microbenchmarks on MI355X: (a) pure v_mfma_f32_16x16x4_f32 issue rate, (b) the same with the conv kernel's LDS
// read pattern (4 x ds_read_b128 per 16 MFMAs).  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = seed * i;
    __syncthreads();
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = seed + lane, b = seed * 2 + lane;
    const float* lp = lds + ((lane & 15) * 24 + (lane >> 4) * 4);
    for (int it = 0; it < iters; ++it) {
        f32x4 bb[NACC];
        if (LDS) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) bb[i] = *reinterpret_cast<const f32x4*>(lp + i * 432 + (it & 7) * 24);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, LDS ? bb[i][j] : b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
}

template <int NACC, bool LDS>
void run(const char* name, int blocks_per_cu) {
    int iters = 20000;
    float* out; hipMalloc(&out, 256 * 256 * 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL((k<NACC, LDS>), dim3(grid), dim3(256), 65536, 0, out, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, LDS>), dim3(grid), dim3(256), 65536, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 2.0 * 16 * 16 * 4 * 4.0 * NACC * iters * (grid * 4.0);
    double cyc_per_mfma = ms * 1e-3 * 2.4e9 / (4.0 * NACC * iters * blocks_per_cu);
    printf("%-28s blocks/CU=%d  %.1f TFLOP/s  (%.1f cycles@2.4GHz per MFMA per SIMD)\n", name, blocks_per_cu, flops / ms / 1e9, cyc_per_mfma);
    hipFree(out);
}
int main() {
    hipFuncSetAttribute((const void*)k<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    run<4, false>("mfma only, 4 acc", 1);
    run<4, false>("mfma only, 4 acc", 2);
    run<2, false>("mfma only, 2 acc", 1);
    run<1, false>("mfma only, 1 acc", 1);
    run<4, true>("mfma + 4 ds_read_b128/16", 1);
    run<4, true>("mfma + 4 ds_read_b128/16", 2);
    return 0;
}
