// fp16-storage 3-D convolution, k = 3, stride 1, Cin = Cout = C in {16, 32, 64} -- the mid-network layers of the c3p / c3 blocks
// (/root/reference/src/model_transforms.py:62-81: the second and third Conv3D / Conv3DTranspose of AnalysisBlock /
// SynthesisBlock) in the fp16 mode of BASELINE.json configs[4] ("fp16 MFMA").  NOT the reference's arithmetic (fp32): a
// separate, labelled mode.  Activations are fp16 in HBM and in LDS, the contraction runs on v_mfma_f32_16x16x32_f16 (fp32
// accumulate), bias / ReLU / residual are applied in fp32, the result is stored as fp16 (or fp32 for the layer that hands over
// to an fp32 consumer).
//
// Why a kernel of its own: with fp16 matrix instructions (16x the fp32 MFMA rate) these layers are bound by HBM, and the
// direct kernels of conv_mfma.hip (fp32 tiles in LDS, one ds_read per MFMA) become LDS/issue-bound long before that.  Here
//   * a voxel is 32 B (C = 16) / 64 B (C = 32): half the HBM and LDS traffic of the fp32 path;
//   * GEMM view per input plane: D[cout][voxel] += W[cout][(tap, cin)] . In[(tap, cin)][voxel], A = weights (resident in
//     registers for the life of the workgroup), B = 16 consecutive voxels of one input row.  K = 32 of one MFMA = two x-taps x
//     16 channels (C = 16; the odd third tap is paired with a zero tap) or one tap x 32 channels (C = 32);
//   * a B fragment (one ds_read_b128 per lane) is used by NINE MFMAs: the workgroup marches along z with three output planes
//     in flight (as conv_wino.hip), so input row (z, y) feeds outputs (z+1-kz, y+1-ky) for all kz, ky -- LDS reads per MFMA
//     drop to 1/9 and the kernel is left with HBM as its only bound;
//   * planes are fetched global -> LDS directly (buffer_load ... lds) into a ring of three, SAME padding and tile overhang
//     through the buffer descriptor's range check.
// Workgroup = 4 waves; wave = 4 rows x 16 voxels x 16 couts: a 16 x 16 (x, y) tile for C = 16, 16 x 8 for C = 32 (two waves per
// row group, one per cout group: 108 of a wave's registers hold its 27 weight fragments).  C = 64 runs as 2 x 2 sub-blocks of
// the 32-channel kernel on strided views of the 64-channel tensors (SUB): the launch of input half 0 leaves raw fp16 partial
// sums in a context-owned scratch tensor, the launch of input half 1 adds them (fp32) before bias / ReLU / residual; both cout
// halves are workgroups of the same launch.  Deterministic: fixed k order, no atomics.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace pccf16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
constexpr unsigned kOOB = 0x80000000u;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// timing probes (tools/build_variant.sh, tools/r06_f16_probe.sh): 1 no MFMAs, 2 no plane loads, 4 no residual loads / stores, 8 no vmcnt wait / barrier
// (2 | 4 leaves nothing observable: the compiler removes the kernel).  C = 16 @128^3 x 8 alone, round 6, us avg | min of 51 launches, with / without
// residual: as shipped 364 | 305 / 317 | 256;  1: 297 | 293 / 210 | 200;  2: 308 | 242 / 254 | 223;  4: 102 | 99 / 102 | 99;  8: 357 | 320 / 305 | 240.
// I.e. input side + matrix work 100 us (5.4 TB/s of reads), the output side (fp16 stores, residual rows) takes the launch to 200 - 300 us, the
// MFMAs add 50 - 60 us on top of the memory-only time instead of hiding under it, and launches are bimodal (max 400 - 470 us).
#ifndef PCC_F16_PROBE
#define PCC_F16_PROBE 0
#endif

struct F16Args {
    const void* in;      // fp16 NDHWC
    const void* w;       // packed fp16 fragments (pcc_f16_pack)
    const float* bias;
    const void* res;     // fp16 NDHWC (same shape as out) or NULL
    void* out;           // fp16, or fp32 with OUT32
    int N, D, H, W;
    int nty, ntx, zsplit, zlen;
    int flags;
    // SUB (C = 64 as 2 x 2 blocks of 32): input channel offset of this launch, partial sums in / out (fp16, 64-channel voxels)
    int ico;
    const void* pre;     // partial sums of the previous input half, or NULL
    void* partial;       // != NULL: store the raw accumulators here instead of the epilogue into `out`
};

// s_waitcnt immediate: vmcnt(n) (6 bits: [3:0] and [15:14]), expcnt(7), lgkmcnt(15)
constexpr int imin(int a, int b) { return a < b ? a : b; }
constexpr int vmcnt_imm(int n) { return 0x0F70 | (n & 15) | ((n >> 4) << 14); }

template <int C>
struct Cfg {
    static constexpr int NCT = C / 16;                 // cout groups of 16
    static constexpr int R = 4;                        // output rows per wave
    static constexpr int TY = (4 / NCT) * R;           // tile rows: 16 (C = 16), 8 (C = 32: two of the four waves per cout group)
    static constexpr int LY = TY + 2;                  // haloed plane rows
    static constexpr int VB = C * 2;                   // bytes per voxel
    static constexpr int SPV = VB / 16;                // 16-byte slots per voxel
    static constexpr int SLOTS = LY * 18 * SPV;
    static constexpr int CHUNKS = (SLOTS + 63) / 64 + 1;   // +1: the zero-tap of the last row reads one voxel past the plane
    static constexpr int PLANE_BYTES = CHUNKS * 1024;
    // Input planes in flight: HBM latency under load (2-3 us) is several steps of this kernel (0.3-0.6 us of MFMAs), so the
    // plane ring is deep: plane s + NRING - 1 is requested while plane s is consumed.  A multiple of 3 (the three output planes
    // in flight rotate with period 3) so that every ring / accumulator index is a compile-time constant of the unrolled loop.
    static constexpr int NRING = 6;                        // 72 KB (C = 16) / 78 KB (C = 32) of LDS: two workgroups per CU
    static constexpr int LDS_BYTES = NRING * PLANE_BYTES;
    static constexpr int ITEMS = (CHUNKS + 3) / 4;     // chunks per wave
    static constexpr int LASTN = CHUNKS - 3 * ITEMS;   // ... of the last wave (waves 0-2 stage ITEMS each)
    static_assert(LASTN >= 0 && LASTN <= ITEMS, "chunk split");
    static constexpr int NF = C == 16 ? 2 : 3;         // B fragments per input row (x-tap pairs / x-taps)
    static constexpr int NA = 9 * NF;                  // A fragments per cout group
};

template <int C, bool OUT32, bool SUB>
__global__ void __launch_bounds__(256, 2) conv_f16_kernel(F16Args a, int nwg) {
    using K = Cfg<C>;
    static_assert(!SUB || C == 32, "sub-block launches use the 32-channel configuration");
    constexpr int GS = SUB ? 64 : C;                       // channels per voxel of the tensors in memory
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kb = lane >> 4;
    const int ct = wave % K::NCT, rg = wave / K::NCT;      // cout group, row group

    int wg = xcd_remap(blockIdx.x, nwg);
    const int cog = SUB ? (wg & 1) : 0;                    // cout half (SUB): neighbours share their input planes in L2
    if (SUB) wg >>= 1;
    const int tx_ = wg % a.ntx; wg /= a.ntx;
    const int ty_ = wg % a.nty; wg /= a.nty;
    const int zs = wg % a.zsplit;
    const int n = wg / a.zsplit;
    const int X0 = tx_ * 16, Y0 = ty_ * K::TY, zb = zs * a.zlen;
    const int nsteps = a.zlen + 2;                         // input planes zb-1 .. zb+zlen
    const size_t HW = (size_t)a.H * a.W;
    const unsigned PLB = (unsigned)(HW * GS * 2);          // bytes per input / fp16 output plane
    const unsigned PLO = OUT32 ? PLB * 2u : PLB;
    const unsigned char* in_n = (const unsigned char*)a.in + (size_t)n * a.D * PLB;

    // ---- A fragments of this wave's cout group: resident for the life of the workgroup
    h16x8 A[K::NA];
    {
        const __amdgpu_buffer_rsrc_t rw = make_rsrc((const unsigned char*)a.w + (size_t)(cog * K::NCT + ct) * K::NA * 1024, (unsigned)K::NA * 1024u);
#pragma unroll
        for (int i = 0; i < K::NA; ++i)
            A[i] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, (int)(i * 1024 + lane * 16), 0, 0));
    }

    // ---- staging: global -> LDS directly; the plane image in LDS is the 18 x 18 haloed tile in memory order
    unsigned rel[K::ITEMS];
#pragma unroll
    for (int it = 0; it < K::ITEMS; ++it) {
        const int slot = (wave * K::ITEMS + it) * 64 + lane;
        const int v = slot / K::SPV, q = slot - v * K::SPV;
        const int yrow = v / 18, xi = v - yrow * 18;
        const int y = Y0 - 1 + yrow, x = X0 - 1 + xi;
        const bool ok = v < K::LY * 18 && y >= 0 && y < a.H && x >= 0 && x < a.W;
        rel[it] = ok ? (unsigned)((y * a.W + x) * (GS * 2) + (SUB ? a.ico * 2 : 0) + q * 16) : kOOB;      // out of range: zeros (SAME padding / pad slots)
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto stage_plane = [&](unsigned plane_off, int z) __attribute__((always_inline)) {
        const bool ok = (unsigned)z < (unsigned)a.D;
        const __amdgpu_buffer_rsrc_t rp = make_rsrc(in_n + (ok ? (size_t)z * PLB : 0), ok ? PLB : 0u);
#pragma unroll
        for (int it = 0; it < K::ITEMS; ++it)
            if ((wave * K::ITEMS + it) < K::CHUNKS)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lds_ptr)(smem + plane_off + (wave * K::ITEMS + it) * 1024), 16, (int)rel[it], 0, 0, 0);
    };

    // ---- B fragment addresses: input row yi (0 .. R+1) of this wave's row group, fragment f
    //      C = 16: lane (n16, kb): x-tap 2f + (kb >> 1), channels 8 (kb & 1) ..;  C = 32: x-tap f, channels 8 kb ..
    const unsigned brow0 = (unsigned)((rg * K::R) * 18 * K::VB);
    unsigned bfo[K::NF];
#pragma unroll
    for (int f = 0; f < K::NF; ++f)
        bfo[f] = C == 16 ? (unsigned)((n16 + 2 * f + (kb >> 1)) * K::VB + (kb & 1) * 16) : (unsigned)((n16 + f) * K::VB + kb * 16);

    // ---- epilogue addressing: lane holds couts 16 ct + 4 kb .. +3 of voxel (row, X0 + n16)
    const int ox = X0 + n16;
    const bool x_ok = ox < a.W;
    const int cofs = 32 * cog + 16 * ct + 4 * kb;
    const bool raw = SUB && a.partial != nullptr;          // first input half: raw partial sums, no epilogue
    const bool has_pre = SUB && a.pre != nullptr;
    const bool has_res = (a.flags & PCC_CONV_ADD) != 0 && !raw;
    const unsigned char* res_n = has_res ? (const unsigned char*)a.res + (size_t)n * a.D * PLB : (const unsigned char*)a.in;
    unsigned char* out_n = (unsigned char*)a.out + (size_t)n * a.D * PLO;
    const unsigned char* pre_n = has_pre ? (const unsigned char*)a.pre + (size_t)n * a.D * PLB : (const unsigned char*)a.in;
    unsigned char* part_n = raw ? (unsigned char*)a.partial + (size_t)n * a.D * PLB : nullptr;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bias4 = ((a.flags & PCC_CONV_BIAS) && !raw) ? *reinterpret_cast<const f32x4*>(a.bias + cofs) : zero4;
    const float relu_lo = ((a.flags & PCC_CONV_RELU) && !raw) ? 0.f : -__builtin_inff();

#pragma unroll
    for (int i = 0; i < K::NRING - 1; ++i) stage_plane((unsigned)i * K::PLANE_BYTES, zb - 1 + i);
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): this wave's plane loads have landed
    __syncthreads();

    // residual rows of the plane that finishes three steps from now (ring of 3: compile-time indices)
    u32x2 resq[3][K::R];
    u32x2 preq[SUB ? 3 : 1][K::R];      // SUB: partial sums of the other input half, same schedule
    auto load_rows = [&](u32x2 (&dst)[K::R], const unsigned char* base, bool on, int zo) __attribute__((always_inline)) {
        const bool zok = on && zo >= zb && zo < zb + a.zlen;
        const __amdgpu_buffer_rsrc_t rres = make_rsrc(base + (zok ? (size_t)zo * PLB : 0), zok ? PLB : 0u);
#pragma unroll
        for (int i = 0; i < K::R; ++i) {
            const int oy = Y0 + rg * K::R + i;
            dst[i] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rres, (int)((x_ok && oy < a.H) ? (unsigned)((oy * a.W + ox) * GS + cofs) * 2u : kOOB), 0, 0));
        }
    };
    auto load_res = [&](int slot, int zo) __attribute__((always_inline)) {
        load_rows(resq[slot], res_n, has_res, zo);
        if (SUB) load_rows(preq[slot], pre_n, has_pre, zo);
    };
    load_res(0, zb - 2);       // (planes zb-2, zb-1 do not exist: zero-sized descriptors, no traffic)
    load_res(1, zb - 1);
    load_res(2, zb);

    f32x4 acc[3][K::R];          // three output planes in flight
#pragma unroll
    for (int i = 0; i < K::R; ++i) { acc[0][i] = zero4; acc[1][i] = zero4; acc[2][i] = zero4; }

    // one input plane: s = step index (input plane z = zb - 1 + s), RS = s mod NRING (ring slot of that plane), PH = s mod 3
    auto step = [&](auto rs_tag, int s) __attribute__((always_inline)) {
        constexpr int RS = decltype(rs_tag)::value, PH = RS % 3;
        constexpr unsigned slotC = (unsigned)RS * K::PLANE_BYTES;                                  // plane s (read now)
        constexpr unsigned slotW = (unsigned)((RS + K::NRING - 1) % K::NRING) * K::PLANE_BYTES;    // plane s + NRING - 1 (requested now)
        if (!(PCC_F16_PROBE & 2)) stage_plane(slotW, zb - 1 + s + K::NRING - 1);
        const unsigned char* pl = smem + slotC + brow0;
#pragma unroll
        for (int yi = 0; yi < K::R + 2; ++yi) {
            h16x8 B[K::NF];
#pragma unroll
            for (int f = 0; f < K::NF; ++f) B[f] = *reinterpret_cast<const h16x8*>(pl + yi * 18 * K::VB + bfo[f]);
#pragma unroll
            for (int f = 0; f < K::NF; ++f)
#pragma unroll
                for (int kz = 0; kz < 3; ++kz)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int yo = yi - ky;                       // output row fed by input row yi through tap ky
                        if (yo < 0 || yo >= K::R) continue;
                        const int as = (PH + 2 - kz) % 3;             // output plane zo = zi + 1 - kz
                        // the kz = 0 taps open a new output plane: its first MFMA starts from 0
                        const bool first = kz == 0 && f == 0 && ky == 0;
                        if (!(PCC_F16_PROBE & 1) || first) acc[as][yo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[(kz * 3 + ky) * K::NF + f], B[f], first ? zero4 : acc[as][yo], 0, 0, 0);
                    }
        }
        // ---- the plane completed by the kz = 2 taps: zo = zb - 2 + s, acc slot PH
        const int zo = zb - 2 + s;
        if (s >= 2 && !(PCC_F16_PROBE & 4)) {
            const __amdgpu_buffer_rsrc_t rout = raw ? make_rsrc(part_n + (size_t)zo * PLB, PLB) : make_rsrc(out_n + (size_t)zo * PLO, PLO);
#pragma unroll
            for (int i = 0; i < K::R; ++i) {
                const int oy = Y0 + rg * K::R + i;
                const bool ok = x_ok && oy < a.H;
                const unsigned vox = (unsigned)(oy * a.W + ox);
                f32x4 o = acc[PH][i];
                if (SUB) {
                    const h16x4 ph = __builtin_bit_cast(h16x4, preq[PH][i]);      // zeros when there is no previous half
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] += (float)ph[c];
                }
                o += bias4;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = fmaxf(o[c], relu_lo);
                if (has_res) {
                    const h16x4 rh = __builtin_bit_cast(h16x4, resq[PH][i]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] += (float)rh[c];
                }
                if (a.flags & PCC_CONV_CLIP01) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = fminf(fmaxf(o[c], 0.f), 1.f);
                }
                if (OUT32 && !raw) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rout, (int)(ok ? (vox * GS + cofs) * 4u : kOOB), 0, 0);
                } else {
                    h16x4 oh;
#pragma unroll
                    for (int c = 0; c < 4; ++c) oh[c] = (_Float16)o[c];      // round to nearest even
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, oh), rout, (int)(ok ? (vox * GS + cofs) * 2u : kOOB), 0, 0);
                }
            }
        }
        if (!(PCC_F16_PROBE & 4)) load_res(PH, zo + 3);               // consumed three steps from now
        // vmcnt retires in order.  Plane s + 1, which the next step reads, was requested NRING - 2 steps ago (at the head of step
        // s - NRING + 2); everything issued in the NRING - 2 steps since may stay in flight across the barrier.  A step issues this
        // wave's plane loads (LASTN for the last wave), R residual (and R partial-sum) loads, and R stores once s >= 2 -- the
        // count must never exceed what was really issued, hence the two cases.
        constexpr int PER = (SUB ? 2 : 1) * K::R;
        const bool steady = s >= K::NRING - 1;          // steps s - NRING + 3 .. s all stored a plane
        if (PCC_F16_PROBE & 8) {
        } else if (wave == 3) {
            if (steady) __builtin_amdgcn_s_waitcnt(vmcnt_imm(imin((K::NRING - 2) * (K::LASTN + PER + K::R), 63)));
            else __builtin_amdgcn_s_waitcnt(vmcnt_imm(imin((K::NRING - 2) * (K::LASTN + PER), 63)));
        } else {
            if (steady) __builtin_amdgcn_s_waitcnt(vmcnt_imm(imin((K::NRING - 2) * (K::ITEMS + PER + K::R), 63)));
            else __builtin_amdgcn_s_waitcnt(vmcnt_imm(imin((K::NRING - 2) * (K::ITEMS + PER), 63)));
        }
        if (!(PCC_F16_PROBE & 8)) __syncthreads();
    };

    for (int s = 0; s < nsteps; s += K::NRING) {
        step(std::integral_constant<int, 0>{}, s);
        if (s + 1 < nsteps) step(std::integral_constant<int, 1>{}, s + 1);
        if (s + 2 < nsteps) step(std::integral_constant<int, 2>{}, s + 2);
        {
            if (s + 3 < nsteps) step(std::integral_constant<int, 3>{}, s + 3);
            if (s + 4 < nsteps) step(std::integral_constant<int, 4>{}, s + 4);
            if (s + 5 < nsteps) step(std::integral_constant<int, 5>{}, s + 5);
        }
    }
}

}  // namespace pccf16

using namespace pccf16;

bool pcc_f16_eligible(const pcc_conv_desc* d) {
    if (d->Cin != d->Cout || (d->Cin != 16 && d->Cin != 32 && d->Cin != 64) || d->k != 3 || d->stride != 1) return false;
    if (d->W % 16 || d->H % 16) return false;
    if (d->out_cstride && d->out_cstride != d->Cout) return false;
    if (d->out_coffset) return false;
    if ((double)d->H * d->W * d->Cin * 4.0 >= 2147483648.0) return false;       // one z-plane per buffer descriptor
    return true;
}

size_t pcc_f16_packed_bytes(int C) {
    if (C == 64) return 4 * pcc_f16_packed_bytes(32);          // [cin half][cout half] images of the 32-channel kernel
    return (size_t)(C / 16) * 9 * (C == 16 ? 2 : 3) * 1024;
}

// wlog: logical forward weights [kz][ky][kx][ci][co] (already flipped for transposed layers).
// Image: [cout group][(kz*3 + ky) * NF + f][lane][8 halfs];  lane (m = lane & 15 -> cout, kb = lane >> 4):
//   C = 16: slot j <-> x-tap 2f + (kb >> 1), cin 8 (kb & 1) + j (zero for the 4th tap);  C = 32: x-tap f, cin 8 kb + j
void pcc_f16_pack(int C, const float* wlog, unsigned short* out) {
    if (C == 64) {
        float* sub = (float*)malloc((size_t)27 * 32 * 32 * sizeof(float));
        if (!sub) return;
        for (int cig = 0; cig < 2; ++cig) for (int cog = 0; cog < 2; ++cog) {
            for (int t = 0; t < 27; ++t) for (int ci = 0; ci < 32; ++ci) for (int co = 0; co < 32; ++co)
                sub[((size_t)t * 32 + ci) * 32 + co] = wlog[((size_t)t * 64 + 32 * cig + ci) * 64 + 32 * cog + co];
            pcc_f16_pack(32, sub, out + (size_t)(cig * 2 + cog) * (pcc_f16_packed_bytes(32) / 2));
        }
        free(sub);
        return;
    }
    const int NCT = C / 16, NF = C == 16 ? 2 : 3;
    for (int ct = 0; ct < NCT; ++ct)
        for (int kz = 0; kz < 3; ++kz) for (int ky = 0; ky < 3; ++ky) for (int f = 0; f < NF; ++f)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int m = lane & 15, kb = lane >> 4;
                    const int kx = C == 16 ? 2 * f + (kb >> 1) : f;
                    const int ci = C == 16 ? 8 * (kb & 1) + j : 8 * kb + j;
                    const float v = kx < 3 ? wlog[((((size_t)kz * 3 + ky) * 3 + kx) * C + ci) * C + ct * 16 + m] : 0.f;
                    const _Float16 h = (_Float16)v;
                    unsigned short bits;
                    __builtin_memcpy(&bits, &h, 2);
                    out[((((size_t)ct * 9 + kz * 3 + ky) * NF + f) * 64 + lane) * 8 + j] = bits;
                }
}

int pcc_conv_f16(pcc_ctx* ctx, const pcc_conv_desc* d, const void* in, const void* w_packed, const float* bias,
                 const void* residual, void* out, bool out32, hipStream_t st) {
    PCC_REQUIRE(pcc_f16_eligible(d), "pcc_conv_f16: shape not covered");
    const bool sub = d->Cin == 64;
    F16Args a;
    a.in = in; a.w = w_packed; a.bias = bias; a.res = residual; a.out = out;
    a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W;
    a.nty = d->H / (d->Cin == 16 ? 16 : 8); a.ntx = d->W / 16;
    a.flags = d->flags;
    a.ico = 0; a.pre = nullptr; a.partial = nullptr;
    const int base = d->N * a.nty * a.ntx * (sub ? 2 : 1);
    int zs = 1;
    // Split z until every CU has a workgroup (slabs down to 4 planes: an idle CU costs more than the two halo planes a split
    // re-reads), then until it has two (slabs of >= 8).  Measured r02 (us, 32->32 @16^3 x32 / @32^3 x8): 30 / 33 -> 15 / 22.
    while (base * zs < ctx->num_cu && d->D % (zs * 2) == 0 && d->D / (zs * 2) >= 4) zs *= 2;
    while (base * zs < 2 * ctx->num_cu && d->D % (zs * 2) == 0 && d->D / (zs * 2) >= 8) zs *= 2;
    a.zsplit = zs; a.zlen = d->D / zs;
    const int nwg = base * zs;
#define PCC_F16_LAUNCH(CC, O32, SUB)                                                                                  \
    {                                                                                                                 \
        { const int rc_ = pcc_enable_big_lds((const void*)conv_f16_kernel<CC, O32, SUB>, Cfg<CC>::LDS_BYTES); if (rc_ != PCC_OK) return rc_; } \
        hipLaunchKernelGGL((conv_f16_kernel<CC, O32, SUB>), dim3((unsigned)nwg), dim3(256), Cfg<CC>::LDS_BYTES, st, a, nwg); \
    }
    if (sub) {
        // input half 0 -> raw fp16 partial sums in the context's scratch tensor; input half 1 adds them and finishes the layer
        void* part = nullptr;
        const int rc = pcc_ctx_scratch(ctx, (size_t)d->N * d->D * d->H * d->W * 64 * 2, &part);
        if (rc != PCC_OK) return rc;
        a.partial = part;
        if (out32) PCC_F16_LAUNCH(32, true, true) else PCC_F16_LAUNCH(32, false, true)
        a.partial = nullptr; a.pre = part; a.ico = 32;
        a.w = (const unsigned char*)w_packed + 2 * pcc_f16_packed_bytes(32);
        if (out32) PCC_F16_LAUNCH(32, true, true) else PCC_F16_LAUNCH(32, false, true)
    } else if (d->Cin == 16) { if (out32) PCC_F16_LAUNCH(16, true, false) else PCC_F16_LAUNCH(16, false, false) }
    else { if (out32) PCC_F16_LAUNCH(32, true, false) else PCC_F16_LAUNCH(32, false, false) }
#undef PCC_F16_LAUNCH
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}
