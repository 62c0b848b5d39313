// Microbenchmark on MI355X: what does an instruction cost between v_mfma_f32_16x16x32_bf16 of the same wave (one wave per
// SIMD, the shape of the 512-register Winograd kernels)?  Exact instruction streams via inline asm: per iteration
// 16 x { MFMA (8 rotating accumulators), K x filler }.  Decides how the split-bf16 Winograd kernel (conv_wino_bf16.hip) builds
// its operand pieces.      hipcc --offload-arch=gfx950 -O3 mfma_bf16_valu.hip -o mfma_bf16_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#ifdef NOMFMA
#define MFMA(ACC)
#else
#define MFMA(ACC) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(a), "v"(b))
#endif
enum { ADD = 0, PKADD, ACCRD, MOV, DS128, CVTPK, AND, PERM, DOT2C, LSHL, DS64, SPLIT_RN, SPLIT_DOT, SPLIT_TRUNC, FMA };

template <int K, int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed * i;
    __syncthreads();
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    f32x4 spare = {seed, seed, seed, seed};
    u32x4 a = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
    float x[8]; f32x2 p[8]; f32x4 q[8]; u32x2 q2[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { x[i] = seed * (i + 1) + lane * 1e-3f; p[i] = (f32x2){seed, seed * i}; q[i] = (f32x4){0, 0, 0, 0}; q2[i] = (u32x2){0, 0}; u[i] = 0; }
    const float y = seed * 0.5f; const f32x2 y2 = {y, y};
    const unsigned msk = 0xffff0000u, sel = 0x07060302u, negone_lo = 0x0000bf80u, negone_hi = 0xbf800000u;
    const unsigned la = (threadIdx.x & 255) * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            MFMA(acc[m & 7]);
            if (MODE == SPLIT_RN || MODE == SPLIT_DOT || MODE == SPLIT_TRUNC) {
                // K = number of value PAIRS split into three bf16 pieces behind this MFMA
#pragma unroll
                for (int v = 0; v < K; ++v) {
                    float& A = x[(2 * v) & 7]; float& B = x[(2 * v + 1) & 7];
                    unsigned H, M, L, t0, t1;
                    if (MODE == SPLIT_RN) {
                        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(H) : "v"(A), "v"(B));
                        asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(t0) : "v"(H));
                        asm volatile("v_and_b32 %0, %1, %2" : "=v"(t1) : "v"(msk), "v"(H));
                        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(A) : "v"(t0));
                        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(B) : "v"(t1));
                        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(M) : "v"(A), "v"(B));
                        asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(t0) : "v"(M));
                        asm volatile("v_and_b32 %0, %1, %2" : "=v"(t1) : "v"(msk), "v"(M));
                        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(A) : "v"(t0));
                        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(B) : "v"(t1));
                        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(L) : "v"(A), "v"(B));
                    } else if (MODE == SPLIT_DOT) {
                        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(H) : "v"(A), "v"(B));
                        asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(A) : "v"(H), "v"(negone_lo));
                        asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(B) : "v"(H), "v"(negone_hi));
                        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(M) : "v"(A), "v"(B));
                        asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(A) : "v"(M), "v"(negone_lo));
                        asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(B) : "v"(M), "v"(negone_hi));
                        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(L) : "v"(A), "v"(B));
                    } else {
                        asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(H) : "v"(B), "v"(A), "v"(sel));
                        asm volatile("v_and_b32 %0, %1, %2" : "=v"(t0) : "v"(msk), "v"(A));
                        asm volatile("v_and_b32 %0, %1, %2" : "=v"(t1) : "v"(msk), "v"(B));
                        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(A) : "v"(t0));
                        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(B) : "v"(t1));
                        asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(M) : "v"(B), "v"(A), "v"(sel));
                        asm volatile("v_and_b32 %0, %1, %2" : "=v"(t0) : "v"(msk), "v"(A));
                        asm volatile("v_and_b32 %0, %1, %2" : "=v"(t1) : "v"(msk), "v"(B));
                        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(A) : "v"(t0));
                        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(B) : "v"(t1));
                        asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(L) : "v"(B), "v"(A), "v"(sel));
                    }
                    u[v & 7] ^= H ^ M ^ L;
                    A += seed; B += seed;      // (2 more VALU ops: keeps the values alive and away from zero)
                }
                continue;
            }
#pragma unroll
            for (int v = 0; v < K; ++v) {
                if (MODE == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[v & 7]) : "v"(y));
                if (MODE == FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[v & 7]) : "v"(y));
                if (MODE == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[v & 7]) : "v"(y2));
                if (MODE == ACCRD) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x[v & 7]) : "a"(spare.x));
                if (MODE == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x[v & 7]) : "v"(y));
                if (MODE == DS128) asm volatile("ds_read_b128 %0, %1" : "=v"(q[v & 7]) : "v"(la));
                if (MODE == DS64) asm volatile("ds_read_b64 %0, %1" : "=v"(q2[v & 7]) : "v"(la));
                if (MODE == CVTPK) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[v & 7]) : "v"(x[v & 7]), "v"(y));
                if (MODE == AND) asm volatile("v_and_b32 %0, %1, %2" : "=v"(u[v & 7]) : "v"(msk), "v"(x[v & 7]));
                if (MODE == LSHL) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[v & 7]) : "v"(x[v & 7]));
                if (MODE == PERM) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[v & 7]) : "v"(x[v & 7]), "v"(y), "v"(sel));
                if (MODE == DOT2C) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x[v & 7]) : "v"(msk), "v"(negone_lo));
            }
        }
        if (MODE == DS128 || MODE == DS64) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    f32x4 s = acc[0];
    for (int i = 1; i < 8; ++i) s += acc[i];
    float r = s.x + s.y + s.z + s.w;
    for (int i = 0; i < 8; ++i) r += x[i] + p[i].x + p[i].y + q[i].x + (float)q2[i].x + (float)u[i];
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
    printf("%-22s K=%d waves/SIMD=%d: %.1f cycles@2.4GHz per {MFMA + K} per wave\n", name, K, blocks_per_cu, cyc);
    hipFree(out);
}
#define SWEEP(MODE, NAME) run<1, MODE>(NAME, 1); run<2, MODE>(NAME, 1); run<4, MODE>(NAME, 1); run<6, MODE>(NAME, 1); run<8, MODE>(NAME, 1);
int main() {
    run<0, ADD>("bf16 mfma only", 1);
    SWEEP(ADD, "v_add_f32") SWEEP(FMA, "v_fma_f32") SWEEP(PKADD, "v_pk_add_f32") SWEEP(ACCRD, "v_accvgpr_read") SWEEP(MOV, "v_mov_b32")
    SWEEP(CVTPK, "v_cvt_pk_bf16_f32") SWEEP(AND, "v_and_b32") SWEEP(LSHL, "v_lshlrev_b32") SWEEP(PERM, "v_perm_b32") SWEEP(DOT2C, "v_dot2c_f32_bf16")
    run<1, DS128>("ds_read_b128", 1); run<2, DS128>("ds_read_b128", 1); run<3, DS128>("ds_read_b128", 1);
    run<1, DS64>("ds_read_b64", 1); run<2, DS64>("ds_read_b64", 1); run<3, DS64>("ds_read_b64", 1);
    // K value pairs split per MFMA: 13 / 9 / 13 VALU ops per pair (11 / 7 / 11 + 2 keep-alive adds)
    run<1, SPLIT_RN>("split RN (13 ops)", 1); run<2, SPLIT_RN>("split RN (13 ops)", 1);
    run<1, SPLIT_DOT>("split dot2c (9 ops)", 1); run<2, SPLIT_DOT>("split dot2c (9 ops)", 1);
    run<1, SPLIT_TRUNC>("split trunc (13 ops)", 1); run<2, SPLIT_TRUNC>("split trunc (13 ops)", 1);
    run<4, ADD>("v_add_f32", 2); run<8, ADD>("v_add_f32", 2); run<8, PKADD>("v_pk_add_f32", 2);
    // two waves per SIMD: does the second wave fill the issue slots the "slow" classes leave?
    run<0, ADD>("bf16 mfma only", 2);
    run<4, MOV>("v_mov_b32", 2); run<8, MOV>("v_mov_b32", 2); run<4, ACCRD>("v_accvgpr_read", 2); run<8, ACCRD>("v_accvgpr_read", 2);
    run<4, CVTPK>("v_cvt_pk_bf16_f32", 2); run<8, CVTPK>("v_cvt_pk_bf16_f32", 2); run<4, DOT2C>("v_dot2c_f32_bf16", 2); run<8, DOT2C>("v_dot2c_f32_bf16", 2);
    run<4, PKADD>("v_pk_add_f32", 2); run<2, SPLIT_DOT>("split dot2c (9 ops)", 2); run<2, DS128>("ds_read_b128", 2);
    return 0;
}
