// Winograd F(2x2, 3x3) in (x, y) + direct 3 taps in z: 3-D convolution for gfx950 (CDNA4), fp32,
// Cin = Cout = 16, k = 3, stride 1.
//
// Why: the 16->16 k3 layers at 64^3 are bound by the fp32 MFMA rate (157 TFLOP/s).  The minimal-filtering form
// in the x-y plane needs 16 multiplies per 2x2 outputs, z tap and (cin, cout) pair instead of 36: 2.25x fewer
// MFMAs, all still exact-fp32 v_mfma_f32_16x16x4_f32.  (The full 3-D form F(2^3,3^3) would save 3.375x but
// needs > 256 VALU-visible registers per lane for the 4x4x4 patch pipeline; VALU cannot read AccVGPRs.)  The
// transforms use +-1 (input, output) and {1, 1/2} (weights, once on the host in double), so results stay
// inside the stated fp32 tolerance of the direct kernels (tests/test_conv_gpu.py) and are bit-deterministic.
//
//   out tile (2x2) = A^T [ sum_dz (G g_dz G^T) . (B^T d_{z+dz-1} B) ] A          per axis:
//   B^T d = (d0-d2, d1+d2, d2-d1, d1-d3)      G g = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2)
//   A^T m = (m0+m1+m2, m1-m2-m3)
//
// GEMM view per (dz, point p):  M_zo[cout][tile] += U_{dz,p}[cout][cin] * V_zi,p[cin][tile],  zo = zi - dz + 1
//   * A operand = U (host-packed, 48 KB, resident in LDS); B operand = V: lane (tile = lane&15, cin quad =
//     lane>>4) transforms ITS OWN 4x4 input patch in registers -> V never touches LDS.
//   * D: lane holds 4 consecutive couts of one tile; three output planes are in flight in 3 x 16 accumulators
//     (AccVGPRs); a finished plane is reduced by A^T lane-locally and stored as float4 per voxel.
//
// Workgroup = 4 waves = 16x16 (x,y) outputs, marching along z one INPUT plane per step; each wave owns 4x4
// tiles.  LDS: U + a ring of 3 input planes 18x18x16ch (x de-interleaved by parity so that the stride-2 tile
// origins become unit stride; 16-byte slots XOR-swizzled by voxel -> conflict-free ds_read_b128).
// Software pipeline over 12 row slots (dz, py) of 16 MFMAs: under the MFMAs of plane s, plane s+1 is read
// from LDS and transformed, plane s+2 is fetched global->registers->LDS, and the output plane completed by
// the dz=2 rows is reduced and stored.  One barrier per step.
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
constexpr int PLANE_BYTES = PLANE_VOX * 64;            // 20736
constexpr int U_BYTES = 48 * 1024;                     // 3 z taps x 16 points x 64 lanes x float4
constexpr int LDS_BYTES = U_BYTES + 3 * PLANE_BYTES;   // 111360
constexpr int PLANE_ITEMS = PLANE_VOX * 4;             // float4 items per plane
constexpr int ITEMS = (PLANE_ITEMS + NT - 1) / NT;     // 6 per thread and plane

struct WinoArgs {
    const float* in;
    const float* u;     // packed transformed weights
    const float* bias;
    const float* res;
    float* out;
    int N, D, H, W;
    int nty, ntx, zsplit, zlen;
    int flags, ocs, oco;
};

// B^T d B on a 4x4 array of float4 (4 input channels each): along x, then along y
__device__ __forceinline__ void transform_x(f32x4 (&P)[16]) {
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        const f32x4 d0 = P[y * 4 + 0], d1 = P[y * 4 + 1], d2 = P[y * 4 + 2], d3 = P[y * 4 + 3];
        P[y * 4 + 0] = d0 - d2; P[y * 4 + 1] = d1 + d2; P[y * 4 + 2] = d2 - d1; P[y * 4 + 3] = d1 - d3;
    }
}
__device__ __forceinline__ void transform_y(f32x4 (&P)[16]) {
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const f32x4 d0 = P[0 + x], d1 = P[4 + x], d2 = P[8 + x], d3 = P[12 + x];
        P[0 + x] = d0 - d2; P[4 + x] = d1 + d2; P[8 + x] = d2 - d1; P[12 + x] = d1 - d3;
    }
}

__global__ void __launch_bounds__(NT, 1) conv16_wino_kernel(WinoArgs a, int nwg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = lane & 15, g = lane >> 4;
    auto ldsr = [&](unsigned off) -> f32x4 { return *reinterpret_cast<const f32x4*>(smem + off); };
    auto ldsw = [&](unsigned off, const f32x4& v) { *reinterpret_cast<f32x4*>(smem + off) = v; };

    int wg = xcd_remap(blockIdx.x, nwg);
    const int tx_ = wg % a.ntx; wg /= a.ntx;
    const int ty_ = wg % a.nty; wg /= a.nty;
    const int zs = wg % a.zsplit;
    const int n = wg / a.zsplit;
    const int X0 = tx_ * 16, Y0 = ty_ * 16, zb = zs * a.zlen;
    const int nsteps = a.zlen + 2;             // input planes zb-1 .. zb+zlen
    const unsigned HW64 = (unsigned)a.H * a.W * 64u;
    const unsigned img_bytes = (unsigned)a.D * HW64;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(a.in + (size_t)n * a.D * a.H * a.W * 16, img_bytes);

    // ---- U -> LDS (12 float4 per thread)
    {
        const __amdgpu_buffer_rsrc_t ru = make_rsrc(a.u, (unsigned)U_BYTES);
        f32x4 tmp[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) tmp[i] = buf_load4(ru, (unsigned)(i * NT + tid) * 16u, 0);
#pragma unroll
        for (int i = 0; i < 12; ++i) ldsw((unsigned)(i * NT + tid) * 16u, tmp[i]);
    }

    // ---- staging items of one plane: global offset inside a z-plane (or OOB) and swizzled LDS slot
    unsigned rel[ITEMS], wr[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int idx = it * NT + tid;
        const int yrow = idx / 72, rem = idx - yrow * 72, xi = rem >> 2, c4 = rem & 3;
        const int y = Y0 - 1 + yrow, x = X0 - 1 + xi;
        const bool ok = y >= 0 && y < a.H && x >= 0 && x < a.W;
        rel[it] = ok ? (unsigned)((y * a.W + x) * 64 + c4 * 16) : kOOB;
        const int v = (yrow * 2 + (xi & 1)) * 9 + (xi >> 1);
        wr[it] = (unsigned)(U_BYTES + v * 64 + ((c4 ^ ((v >> 1) & 3)) << 4));
        if (idx >= PLANE_ITEMS) { rel[it] = rel[it - 1]; wr[it] = wr[it - 1]; }   // tail: repeat the previous item (same data, same slot)
    }
    auto load_plane = [&](f32x4 (&dst)[ITEMS], int z) __attribute__((always_inline)) {
        const bool zok = (unsigned)z < (unsigned)a.D;
        const unsigned zflag = zok ? 0u : kOOB, zoff = zok ? (unsigned)z * HW64 : 0u;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) dst[it] = buf_load4(rin, rel[it] | zflag, zoff);
    };

    // ---- per-lane patch read addresses (ring slot 0), tile of this lane
    const int wx = wave & 1, wy = wave >> 1;
    const int TX = 4 * wx + (t & 3), TY = 4 * wy + (t >> 2);
    unsigned ra[16];
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const int v = 36 * TY + 18 * dy + 9 * (dx & 1) + TX + (dx >> 1);
            ra[dy * 4 + dx] = (unsigned)(U_BYTES + v * 64 + ((g ^ ((v >> 1) & 3)) << 4));
        }
    const unsigned ua = (unsigned)lane * 16u;

    // ---- epilogue addressing: lane writes couts 4g..4g+3 of the 2x2 voxels of its tile
    const int ox0 = X0 + 2 * TX, oy0 = Y0 + 2 * TY;
    unsigned ovo[4], rvo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned vox = (unsigned)((oy0 + (q >> 1)) * a.W + ox0 + (q & 1));
        ovo[q] = (vox * (unsigned)a.ocs + (unsigned)a.oco + 4u * g) * 4u;
        rvo[q] = (vox * 16u + 4u * g) * 4u;
    }
    const unsigned HWO = (unsigned)a.H * a.W * (unsigned)a.ocs * 4u;
    const bool has_res = (a.flags & PCC_CONV_ADD) != 0;
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(has_res ? a.res + (size_t)n * a.D * a.H * a.W * 16 : a.in, has_res ? img_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(a.out + (size_t)n * a.D * a.H * a.W * a.ocs, (unsigned)a.D * HWO);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bias4 = (a.flags & PCC_CONV_BIAS) ? *reinterpret_cast<const f32x4*>(a.bias + g * 4) : zero4;
    const float relu_lo = (a.flags & PCC_CONV_RELU) ? 0.f : -__builtin_inff();
    const float clip_lo = (a.flags & PCC_CONV_CLIP01) ? 0.f : -__builtin_inff();
    const float clip_hi = (a.flags & PCC_CONV_CLIP01) ? 1.f : __builtin_inff();

    // ---- prologue: input planes s = 0, 1 (z = zb-1, zb) -> ring slots 0, 1;  V(0) -> registers
    f32x4 stg[ITEMS];
    {
        f32x4 stg1[ITEMS];
        load_plane(stg, zb - 1);
        load_plane(stg1, zb);
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) { ldsw(wr[it], stg[it]); ldsw(wr[it] + PLANE_BYTES, stg1[it]); }
    }
    __syncthreads();

    f32x4 Vc[16], Vn[16];       // transformed patch of the current / next input plane
    f32x4 Ub[2][4];
    f32x4 acc[3][16];           // three output planes in flight
    f32x4 S[2][2];              // A^T-reduced 2x2 outputs of the finished plane
    f32x4 resv[4], ost[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) Vc[i] = ldsr(ra[i]);
    transform_x(Vc);
    transform_y(Vc);
#pragma unroll
    for (int px = 0; px < 4; ++px) Ub[0][px] = ldsr(ua + (unsigned)((2 * 16 + px) * 1024));   // first row: dz = 2, py = 0
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][i] = zero4; acc[1][i] = zero4; acc[2][i] = zero4; }

    // one input plane: s = step index (input plane z = zb-1+s), PH = s mod 3
    auto step = [&](auto ph_tag, int s) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_tag)::value;
        constexpr unsigned slotN = (unsigned)((PH + 1) % 3) * PLANE_BYTES;   // plane s+1 (read)
        constexpr unsigned slotW = (unsigned)((PH + 2) % 3) * PLANE_BYTES;   // plane s+2 (written)
        constexpr int AF = PH;                                               // acc slot of the plane finished by dz = 2
        const int zo = zb - 2 + s;                                           // that plane (valid when s >= 2)
        const unsigned sflag = s >= 2 ? 0u : kOOB;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int dz = 2 - (j >> 2), py = j & 3;
            const int as = (PH + 2 - dz) % 3;
            // (1) U fragments of the next row (wraps to the first row of the next step)
            {
                const int jn = (j + 1) % 12, dzn = 2 - (jn >> 2), pyn = jn & 3;
#pragma unroll
                for (int px = 0; px < 4; ++px) Ub[(j + 1) & 1][px] = ldsr(ua + (unsigned)(((dzn * 4 + pyn) * 4 + px) * 1024));
            }
            // (2) the 16 MFMAs of this row: 4 independent accumulators, k-chained; dz = 0 opens a new output plane
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    const f32x4 c = (dz == 0 && kk == 0) ? zero4 : acc[as][py * 4 + px];
                    acc[as][py * 4 + px] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ub[j & 1][px][kk], Vc[py * 4 + px][kk], c, 0, 0, 0);
                }
            // (3) everything else, spread over the slots
            if (j == 0) {
                load_plane(stg, zb - 1 + s + 2);
#pragma unroll
                for (int q = 0; q < 4; ++q) resv[q] = buf_load4(rres, rvo[q] | sflag, s >= 2 ? (unsigned)zo * HW64 : 0u);
#pragma unroll
                for (int i = 0; i < 16; ++i) Vn[i] = ldsr(ra[i] + slotN);
            }
            else if (j == 1) transform_x(Vn);
            else if (j == 2) transform_y(Vn);
            else if (j >= 5 && j <= 8) {
                // A^T along x on row r of the finished plane, accumulate A^T along y
                const int r = j - 5;
                const f32x4 m0 = acc[AF][r * 4 + 0], m1 = acc[AF][r * 4 + 1], m2 = acc[AF][r * 4 + 2], m3 = acc[AF][r * 4 + 3];
                const f32x4 r0 = m0 + m1 + m2, r1 = m1 - m2 - m3;
                if (r == 0) { S[0][0] = r0; S[0][1] = r1; }
                else if (r == 1) { S[0][0] += r0; S[0][1] += r1; S[1][0] = r0; S[1][1] = r1; }
                else if (r == 2) { S[0][0] += r0; S[0][1] += r1; S[1][0] -= r0; S[1][1] -= r1; }
                else { S[1][0] -= r0; S[1][1] -= r1; }
                if (j == 8) {
#pragma unroll
                    for (int it = 0; it < ITEMS; ++it) ldsw(wr[it] + slotW, stg[it]);
                }
            } else if (j == 9) {
                // epilogue of the finished plane: bias, ReLU, residual, clip (all branch-free), float4 stores
                // (dropped by the range check while s < 2)
                const unsigned zoff = s >= 2 ? (unsigned)zo * HWO : 0u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 o = S[q >> 1][q & 1] + bias4;
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = fmaxf(o[c], relu_lo);
                    o += resv[q];     // zeros without PCC_CONV_ADD (zero-sized buffer)
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = fminf(fmaxf(o[c], clip_lo), clip_hi);
                    ost[q] = o;
                }
                // gfx950: a buffer_store_dwordx4 whose soffset is an SGPR reads its data late; a VALU write to the data
                // registers in the next issue slots corrupts the last dword of the last lanes (seen as out[q].w <-
                // out[q+1].w).  The compiler only guards the immediate-soffset form, so: immediate soffset (z offset
                // folded into the address) AND the data registers stay live until the next slot.
#pragma unroll
                for (int q = 0; q < 4; ++q) buf_store4(rout, ost[q], (ovo[q] + zoff) | sflag, 0);
            } else if (j == 10) {
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("" ::"v"(ost[q]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // V(s+1) becomes the current patch (all MFMAs that read Vc are issued); planes in LDS are published
#pragma unroll
        for (int i = 0; i < 16; ++i) Vc[i] = Vn[i];
        __syncthreads();
    };

    for (int s = 0; s < nsteps; s += 3) {
        step(std::integral_constant<int, 0>{}, s);
        if (s + 1 < nsteps) step(std::integral_constant<int, 1>{}, s + 1);
        if (s + 2 < nsteps) step(std::integral_constant<int, 2>{}, s + 2);
    }
}

}  // namespace pccwino

using namespace pccwino;

bool pcc_wino_eligible(const pcc_conv_desc* d) {
    if (d->Cin != 16 || d->Cout != 16 || d->k != 3 || d->stride != 1) return false;
    if (d->W % 16 || d->H % 16) return false;
    if ((double)d->D * d->H * d->W * 64.0 >= 2147483648.0) return false;
    const int ocs = d->out_cstride ? d->out_cstride : d->Cout;
    if (ocs % 4 || d->out_coffset % 4) return false;
    if ((double)d->D * d->H * d->W * ocs * 4.0 >= 2147483648.0) return false;
    return true;
}

int pcc_conv16_wino(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* u_packed, const float* bias,
                    const float* residual, float* out, hipStream_t st) {
    PCC_REQUIRE(pcc_wino_eligible(d), "pcc_conv16_wino: shape not covered");
    WinoArgs a;
    a.in = in; a.u = u_packed; a.bias = bias; a.res = residual; a.out = out;
    a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W;
    a.nty = d->H / 16; a.ntx = d->W / 16;
    a.flags = d->flags;
    a.ocs = d->out_cstride ? d->out_cstride : d->Cout;
    a.oco = d->out_coffset;
    // split z so that every CU gets a workgroup (each split pays the 48 KB U load and two halo planes)
    const int base = d->N * a.nty * a.ntx;
    int zs = 1;
    while (base * zs < ctx->num_cu && d->D % (zs * 2) == 0 && d->D / (zs * 2) >= 8) zs *= 2;
    a.zsplit = zs; a.zlen = d->D / zs;
    const int nwg = base * zs;
    static thread_local bool configured = false;
    if (!configured) {
        PCC_CHECK_HIP(hipFuncSetAttribute((const void*)conv16_wino_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        configured = true;
    }
    hipLaunchKernelGGL(conv16_wino_kernel, dim3((unsigned)nwg), dim3(NT), LDS_BYTES, st, a, nwg);
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}
