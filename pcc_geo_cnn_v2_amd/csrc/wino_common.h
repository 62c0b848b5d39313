// Shared pieces of the Winograd F(2x2,3x3)+z kernels (conv_wino.hip: exact-fp32 MFMA; conv_wino_bf16.hip: split-bf16 MFMA):
// buffer-descriptor helpers, the LDS plane geometry, the launch arguments and the fp32 input / output transforms.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace pccwino {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, f32x4 v, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff, (int)soff, 0);
}
constexpr unsigned kOOB = 0x80000000u;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

constexpr int NT = 256;
constexpr int PLANE_VOX = 18 * 18;
constexpr int PLANE_ITEMS = PLANE_VOX * 4;             // float4 slots per plane (1296)
constexpr int CHUNKS = (PLANE_ITEMS + 63) / 64;        // 21 wave-sized (1 KB) chunks; the last one is padding beyond slot 1295
constexpr int PLANE_BYTES = CHUNKS * 1024;             // 21504: three planes end below 64 KB -> every plane offset is a DS immediate
constexpr int U_BASE = 3 * PLANE_BYTES;
constexpr int U_BYTES = 48 * 1024;                     // 3 z taps x 16 points x 64 lanes x float4
constexpr int LDS_BYTES = U_BASE + U_BYTES;            // 113664
constexpr int LDS_BYTES_CIN = U_BASE + 2 * U_BYTES;    // 162816 <= 160 KB (conv16_wino_cin_kernel): tile ring + two U buffers
constexpr int ITEMS = 6;                               // chunks per wave: wave w stages chunks 5w .. 5w+5 (5, 10, 15 twice: same data)

struct WinoArgs {
    const float* in;
    const float* u;     // packed transformed weights of this cin group: [cout group][48][64][4]
    const float* bias;
    const float* res;
    float* out;
    int N, D, H, W;
    int nty, ntx, zsplit, zlen;
    int nco;            // cout groups of 16 handled by this launch (grid dimension)
    int flags, ocs, oco;
    int ics, ico;       // input channel stride / offset of this launch's 16-channel cin group
    int rcs;            // residual channel stride (its channel offset follows the cout group)
    int ncig;           // cin groups processed by one launch of conv16_wino_cin_kernel (u: [cin group][cout group][48][64][4])
    // fp16-split kernels (conv_wino_f16s.hip): per-block max |x| of the input (fp32 bit patterns, one per block n) that fixes the block's
    // power-of-two pre-scale; where to record the same for the output (nullptr: not recorded); the weight image's scale (device, 1 float)
    const unsigned* amax_in = nullptr;
    unsigned* amax_out = nullptr;
    const float* utail = nullptr;
    int ig0 = 0;        // first cin group this launch marches (the 64-channel layers run as two launches of two groups)
};

// Measured on MI355X (tools/ubench/mfma_valu.hip): a wave's VALU instructions do NOT overlap with its own fp32 MFMAs
// (v_mfma_f32_16x16x4_f32 runs at the packed-FMA rate of the same SIMD): every VALU op costs ~5 cycles of MFMA time,
// v_mov / v_accvgpr_read ~8.  Hence: packed adds everywhere (the compiler turns a-b into two scalar v_sub), no
// register copies, no per-lane address arithmetic in the loop.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 sub4(const f32x4& a, const f32x4& b) {
    f32x2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(__builtin_shufflevector(a, a, 0, 1)), "v"(__builtin_shufflevector(b, b, 0, 1)));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(__builtin_shufflevector(a, a, 2, 3)), "v"(__builtin_shufflevector(b, b, 2, 3)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ f32x4 add4(const f32x4& a, const f32x4& b) {
    f32x2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(__builtin_shufflevector(a, a, 0, 1)), "v"(__builtin_shufflevector(b, b, 0, 1)));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"(__builtin_shufflevector(a, a, 2, 3)), "v"(__builtin_shufflevector(b, b, 2, 3)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

// AccVGPR -> VGPR at a place of OUR choosing (the register allocator otherwise splits the live range right behind the
// defining MFMA, i.e. in the middle of an MFMA block).  Inline asm is invisible to the hazard recogniser: callers keep at least
// one slot of 16 MFMAs between the MFMA that wrote the accumulator and this read.
__device__ __forceinline__ f32x4 acc_read(const f32x4& a) {
    f32x4 d;
    asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]) : "a"(a[0]), "a"(a[1]), "a"(a[2]), "a"(a[3]));
    return d;
}

// B^T along x, in place, on a 4x4 array of float4 (4 input channels each)
__device__ __forceinline__ void transform_x_rows(f32x4 (&P)[16], int y0, int y1) {
#pragma unroll
    for (int y = y0; y < y1; ++y) {
        const f32x4 d0 = P[y * 4 + 0], d1 = P[y * 4 + 1], d2 = P[y * 4 + 2], d3 = P[y * 4 + 3];
        P[y * 4 + 0] = sub4(d0, d2); P[y * 4 + 1] = add4(d1, d2); P[y * 4 + 2] = sub4(d2, d1); P[y * 4 + 3] = sub4(d1, d3);
    }
}
// B^T along y, one output row: V[r][x] from the x-transformed patch
__device__ __forceinline__ void transform_y_row(f32x4 (&V)[16], const f32x4 (&P)[16], int r) {
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        if (r == 0) V[x] = sub4(P[x], P[8 + x]);
        else if (r == 1) V[4 + x] = add4(P[4 + x], P[8 + x]);
        else if (r == 2) V[8 + x] = sub4(P[8 + x], P[4 + x]);
        else V[12 + x] = sub4(P[4 + x], P[12 + x]);
    }
}

}  // namespace pccwino
