// Microbenchmark on MI355X: can one wave overlap VALU work with its own MFMAs?  Exact instruction streams via inline asm:
// per iteration 16 x { v_mfma_f32_16x16x4_f32 (4 rotating accumulators), K x VALU op } with 1 or 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MFMA(ACC) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(ACC) : "v"(a), "v"(b))
// MODE 0: v_add_f32, 1: v_pk_add_f32, 2: v_accvgpr_read (from an unrelated AGPR), 3: v_mov_b32, 4: ds_read_b128
template <int K, int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed * i;
    __syncthreads();
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0, spare = {seed, seed, seed, seed};
    float a = seed + lane, b = seed * 2 + lane;
    float x[8]; f32x2 p[8]; f32x4 q[8];
    for (int i = 0; i < 8; ++i) { x[i] = seed * i; p[i] = (f32x2){seed, seed * i}; q[i] = (f32x4){0, 0, 0, 0}; }
    const float y = seed * 0.5f; const f32x2 y2 = {y, y};
    const unsigned la = (threadIdx.x & 255) * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if ((m & 3) == 0) MFMA(acc0); else if ((m & 3) == 1) MFMA(acc1); else if ((m & 3) == 2) MFMA(acc2); else MFMA(acc3);
#pragma unroll
            for (int v = 0; v < K; ++v) {
                if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[v]) : "v"(y));
                if (MODE == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[v]) : "v"(y2));
                if (MODE == 2) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x[v]) : "a"(spare.x));
                if (MODE == 3) asm volatile("v_mov_b32 %0, %1" : "=v"(x[v]) : "v"(y));
                if (MODE == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(q[v]) : "v"(la));
            }
        }
        if (MODE == 4) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    f32x4 s = acc0 + acc1 + acc2 + acc3;
    float r = s.x + s.y + s.z + s.w;
    for (int i = 0; i < 8; ++i) r += x[i] + p[i].x + p[i].y + q[i].x;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int K, int MODE>
void run(const char* name, int blocks_per_cu) {
    int iters = 4000;
    float* out; hipMalloc(&out, 256 * 256 * 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu;
    int lds = blocks_per_cu == 1 ? 100 * 1024 : 64 * 1024;     // 100 KB: one workgroup per CU
    hipFuncSetAttribute((const void*)k<K, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((k<K, MODE>), dim3(grid), dim3(256), lds, 0, out, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<K, MODE>), dim3(grid), dim3(256), lds, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double cyc = ms * 1e-3 * 2.4e9 / (16.0 * iters * blocks_per_cu);
    printf("%-14s K=%d waves/SIMD=%d: %.1f cycles@2.4GHz per {MFMA + K ops} per wave\n", name, K, blocks_per_cu, cyc);
    hipFree(out);
}
int main() {
    run<0, 0>("mfma only", 1);
    run<2, 0>("v_add_f32", 1); run<4, 0>("v_add_f32", 1); run<6, 0>("v_add_f32", 1); run<7, 0>("v_add_f32", 1); run<8, 0>("v_add_f32", 1);
    run<4, 1>("v_pk_add_f32", 1); run<6, 1>("v_pk_add_f32", 1); run<8, 1>("v_pk_add_f32", 1);
    run<4, 2>("accvgpr_read", 1); run<8, 2>("accvgpr_read", 1);
    run<4, 3>("v_mov_b32", 1); run<8, 3>("v_mov_b32", 1);
    run<1, 4>("ds_read_b128", 1); run<2, 4>("ds_read_b128", 1);
    run<4, 0>("v_add_f32", 2); run<8, 0>("v_add_f32", 2); run<8, 1>("v_pk_add_f32", 2);
    return 0;
}
