// Microbenchmark on MI355X (round 6): what does a VALU / LDS / VMEM instruction cost beside v_mfma_f32_16x16x32_f16 of the same wave (one wave per
// SIMD), depending on WHERE the MFMA's operands live?  ACCA: accumulators in AccVGPRs, A / B in arch VGPRs (the usual arrangement);
// ACCV: accumulators in arch VGPRs, A in AccVGPRs (conv_dir_f16s.hip: 216 registers of resident weights), B in arch VGPRs.
// per iteration 16 x { MFMA (8 rotating accumulators), K x filler }.      hipcc --offload-arch=gfx950 -O3 mfma_f16_regfile.hip -o mfma_f16_regfile
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
enum { MUL = 0, CVT, DSW, DSR, MIX };

template <int K, int MODE, bool ACCV>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed * i;
    __syncthreads();
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    u32x4 a = {0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, b = a, aa;
    asm volatile("; def %0" : "=a"(aa));      // (an AccVGPR tuple with whatever it holds: only the register class matters)
    float x[8]; unsigned u[8]; f32x4 q[8];
    for (int i = 0; i < 8; ++i) { x[i] = seed * (i + 1) + lane * 1e-3f; u[i] = 0; q[i] = (f32x4){seed, seed, seed, seed}; }
    const float y = seed * 0.5f;
    const unsigned la = (threadIdx.x & 255) * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (ACCV) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "a"(aa), "v"(b));
            else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[m & 7]) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < K; ++v) {
                if (MODE == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[v]) : "v"(y));
                if (MODE == CVT) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[v]) : "v"(x[v]), "v"(y));
                if (MODE == DSW) asm volatile("ds_write_b128 %0, %1" :: "v"(la), "v"(q[v]));
                if (MODE == DSR) asm volatile("ds_read_b128 %0, %1" : "=v"(q[v]) : "v"(la));
                if (MODE == MIX) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(u[v]) : "v"(u[(v + 1) & 7]), "v"(x[v]));
            }
        }
        if (MODE == DSW || MODE == DSR) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7];
    float r = s.x + s.y + s.z + s.w;
    for (int i = 0; i < 8; ++i) r += x[i] + u[i] + q[i].x;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int K, int MODE, bool ACCV>
void run(const char* name) {
    int iters = 4000;
    float* out; hipMalloc(&out, 256 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int lds = 100 * 1024;     // one workgroup per CU
    hipFuncSetAttribute((const void*)k<K, MODE, ACCV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((k<K, MODE, ACCV>), dim3(256), dim3(256), lds, 0, out, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<K, MODE, ACCV>), dim3(256), dim3(256), lds, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-18s acc in %s K=%d: %.1f ns per {MFMA + K fillers} per wave\n", name, ACCV ? "VGPR (A in AGPR)" : "AGPR (A in VGPR)", K, ms * 1e6 / (16.0 * iters));
    hipFree(out);
}
template <bool ACCV>
void all() {
    run<0, MUL, ACCV>("mfma only");
    run<1, MUL, ACCV>("v_mul_f32"); run<2, MUL, ACCV>("v_mul_f32"); run<4, MUL, ACCV>("v_mul_f32");
    run<1, CVT, ACCV>("v_cvt_pk_f16_f32"); run<2, CVT, ACCV>("v_cvt_pk_f16_f32");
    run<1, MIX, ACCV>("v_fma_mixlo_f16"); run<2, MIX, ACCV>("v_fma_mixlo_f16");
    run<1, DSW, ACCV>("ds_write_b128"); run<1, DSR, ACCV>("ds_read_b128");
}
int main() { all<false>(); all<true>(); return 0; }
