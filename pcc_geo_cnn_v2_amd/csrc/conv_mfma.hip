// MFMA implicit-GEMM 3-D convolution / transposed convolution for gfx950 (CDNA4), fp32.
//
// GEMM view:  D[cout][voxel] = sum_k  Wt[cout][k] * In[k][voxel],   k = (tap, cin)
//   * MFMA: v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered FMA chain -> bit-deterministic results that do
//     not depend on tile position, batch size or launch geometry; SURVEY.md §5 determinism requirement).
//   * A operand = weights, pre-packed on the host in fragment order (one float4 per lane covers 4 MFMAs);
//     B operand = a 16-voxel row of the NDHWC input tile staged in LDS (one ds_read_b128 per lane covers
//     the 4 MFMAs of a 16-channel group: MFMA j contracts channels {j, 4+j, 8+j, 12+j}).
//   * D layout: lane holds 4 consecutive output channels of one voxel -> float4 epilogue loads/stores
//     (bias, ReLU, residual add, clip fused).
//   * LDS voxel stride = staged channels + 8 floats: the +8 makes every ds_read_b128 lane group hit 16
//     distinct 16-byte bank slots (stride = 2 mod 4 slots), so tap offsets stay immediates.
//
// Kernels:
//   conv_fwd_kernel : Conv3D stride 1|2 (and Conv3DTranspose stride 1 through host-flipped weights),
//                     Cin,Cout multiples of 16; 16 input channels staged per pass.
//   conv_tr2_kernel : Conv3DTranspose stride 2 by output-parity decomposition (8 classes, each a small
//                     stride-1 gather conv on the input grid); all Cin staged at once.
//   conv_cin1_kernel: Conv3D with Cin = 1 (first layer): k-slots of the MFMA are kernel taps along x.
//   conv_cout1_kernel: Conv3DTranspose with Cout = 1 (last layer): VALU dot products from an LDS tile.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"

// Compiled several times with -DPCC_PART=k (see Makefile): part 0 holds the dispatcher, the weight packer and the
// first/last-layer kernels; parts 1.. hold the explicit instantiations of launch_fwd<> / launch_tr2<> so that the
// heavily unrolled kernels build in parallel.
#ifndef PCC_PART
#define PCC_PART 0
#endif

namespace pccmfma {

typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// PCC_CONV_F16 (BASELINE.json configs[4]): the same fragments -- a lane's float4 holds k-slots 4*(lane>>4)..+3 of a
// 16-channel group, exactly the k layout of v_mfma_f32_16x16x16_f16 -- are rounded to fp16 (v_cvt_pk_f16_f32, RTN) and
// contracted by ONE matrix instruction instead of four; accumulation, bias, activations in HBM and LDS stay fp32.
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16h(const f32x4& a, const f32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_convertvector(a, h16x4), __builtin_convertvector(b, h16x4), c, 0, 0, 0);
}

// Raw buffer resources: the hardware range check returns 0 for offsets >= num_records, which implements the
// SAME zero padding (and the tile overhang) without a single branch; the descriptor is wave-uniform (SGPRs).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
constexpr unsigned kOOB = 0x80000000u;                 // >= any per-image byte size the planner admits
// sched_barrier mask: VALU, SALU, DS and transcendental ops may cross; vector-memory ops and MFMAs may not ->
// a prefetch load stays in front of the MFMAs of the tap it was written in (two taps before its use).
#define PCC_PIN_VMEM() __builtin_amdgcn_sched_barrier(0x786)
// stricter: only VALU / SALU / transcendental ops may cross (memory ops and MFMAs keep their written order)
#define PCC_PIN_MEM_MFMA() __builtin_amdgcn_sched_barrier(0x406)

struct ConvArgs {
    const float* in;
    const float* w;  // packed
    const float* bias;
    const float* res;
    float* out;
    int N, D, H, W;     // input dims
    int OD, OH, OW;     // output dims
    int ntz, nty, ntx;  // tiles per dim (base grid)
    int flags, ocs, oco;
    // 16 -> 1 last layer with the occupancy decision folded into its epilogue (pcc_thr_fuse): bit (z,y,x) of `mask` = x_hat > thr[n]
    const float* thr = nullptr;
    unsigned short* mask = nullptr;
    int thr_clip = 0;
    // per-block max |out| for the fp16-split layer that consumes `out` (common.h, pcc_conv_ext); filled by conv_cin1_kernel
    unsigned* amax_out = nullptr;
};

__device__ __forceinline__ f32x4 store_out(const ConvArgs& a, f32x4 v, size_t vox, int c0, int COUT) {
    // v = 4 consecutive output channels c0..c0+3 of voxel `vox`; returns what was stored (fp32 path)
    if (a.flags & PCC_CONV_BIAS) v += *reinterpret_cast<const f32x4*>(a.bias + c0);
    if (a.flags & PCC_CONV_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (a.flags & PCC_CONV_ADD) v += *reinterpret_cast<const f32x4*>(a.res + vox * COUT + c0);
    if (a.flags & PCC_CONV_CLIP01) {
        v.x = fminf(fmaxf(v.x, 0.f), 1.f); v.y = fminf(fmaxf(v.y, 0.f), 1.f);
        v.z = fminf(fmaxf(v.z, 0.f), 1.f); v.w = fminf(fmaxf(v.w, 0.f), 1.f);
    }
    if (a.flags & PCC_CONV_OUT16) {      // fp16 hand-over to conv_f16.hip (fp16 mode): 8 bytes per lane
        h16x4 h;
        h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
        *reinterpret_cast<h16x4*>(reinterpret_cast<_Float16*>(a.out) + vox * a.ocs + a.oco + c0) = h;
    } else {
        *reinterpret_cast<f32x4*>(a.out + vox * a.ocs + a.oco + c0) = v;
    }
    return v;
}

// XCD-aware tile index: consecutive tile ids go to the same XCD (blocks are dispatched round-robin over
// the 8 XCDs), so that neighbouring tiles share their halos in one L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// =====================================================================================================
// forward conv (stride 1 or 2), Cin % 16 == 0, Cout % 16 == 0
//   tile = TZ x TY x TXT output voxels; a "row" = 16 voxels = RY(=16/TX) y-lines x TX voxels along x;
//   each wave owns R rows that are consecutive in y.
// =====================================================================================================
template <int CIN, int COUT, int KS, int S, int TX, int TZ, int TY, int TXT, int R, int CTW = COUT / 16>
struct FwdCfg {
    static constexpr int NG = CIN / 16, NCT = COUT / 16;
    static constexpr int RY = 16 / TX;
    static constexpr int NYB = TY / RY, NXB = TXT / TX;
    static constexpr int NCG = NCT / CTW;              // cout-tile groups: waves also split the output channels
    static constexpr int NW = TZ * (NYB / R) * NXB * NCG;
    static constexpr int NT = NW * 64;
    static constexpr int PL = (S == 1) ? (KS - 1) / 2 : (KS - 2) / 2;  // SAME pad_low (even input dims for S=2)
    static constexpr int LZ = (TZ - 1) * S + KS, LY = (TY - 1) * S + KS, LX = (TXT - 1) * S + KS;
    static constexpr int VS = 24;  // floats per voxel in LDS: 16 staged channels + 8 pad
    static constexpr int NV = LZ * LY * LX;
    static constexpr int LDS_BYTES = NV * VS * 4;
    static constexpr int ITEMS = (NV * 4 + NT - 1) / NT;
    static_assert(TY % RY == 0 && NYB % R == 0 && TXT % TX == 0, "bad tile");
};

template <int CIN, int COUT, int KS, int S, int TX, int TZ, int TY, int TXT, int R, int CTW = COUT / 16, bool F16 = false>
__global__ void __launch_bounds__((FwdCfg<CIN, COUT, KS, S, TX, TZ, TY, TXT, R, CTW>::NT))
conv_fwd_kernel(ConvArgs a) {
    using C = FwdCfg<CIN, COUT, KS, S, TX, TZ, TY, TXT, R, CTW>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int v = lane & 15, cq = lane >> 4;

    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = t % a.ntx; t /= a.ntx;
    const int ty = t % a.nty; t /= a.nty;
    const int tz = t % a.ntz;
    const int n = t / a.ntz;
    const int oz0 = tz * TZ, oy0 = ty * TY, ox0 = tx * TXT;          // output-tile origin
    const int iz0 = oz0 * S - C::PL, iy0 = oy0 * S - C::PL, ix0 = ox0 * S - C::PL;  // LDS-tile origin (input)

    // wave -> (z, y-block group, x-block)
    int wv = wave;
    const int ct0 = (wv % C::NCG) * CTW; wv /= C::NCG;   // first cout tile of this wave
    const int w_xb = wv % C::NXB; wv /= C::NXB;
    const int w_yg = wv % (C::NYB / R);
    const int w_z = wv / (C::NYB / R);
    const int ry = v / TX, rx = v % TX;
    const int ly0 = (w_yg * R * C::RY + ry), lx0 = (w_xb * TX + rx);  // local output coords of row 0
    const float* lbase = lds + ((w_z * S * C::LY + ly0 * S) * C::LX + lx0 * S) * C::VS + cq * 4;
    constexpr int ROW_OFF = C::RY * S * C::LX * C::VS;  // floats between consecutive rows of a wave

    f32x4 acc[R][CTW];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) acc[i][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* inb = a.in + (size_t)n * a.D * a.H * a.W * CIN;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(inb, (unsigned)a.D * a.H * a.W * CIN * 4u);
    constexpr int NTAP = KS * KS * KS;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, (unsigned)(C::NG * NTAP * C::NCT) * 1024u);
    const unsigned wlane = lane * 16;
    // weights: RING-deep register prefetch ring over the linear (g, tap) sequence; the loads are pinned
    // RING - 1 taps ahead of their use so that L2 latency (500-900 cycles) never reaches the MFMA pipe.  A tap is R * CTW * 4 MFMAs:
    // two taps ahead are >= 1000 cycles for the big tiles, but only 256 for the R = CTW = 1 tiles of the 4^3 / 8^3 grids (round 3:
    // those layers were latency-bound on exactly this -- 64 -> 64 @4^3 26.5 us for 5 us of MFMAs), hence the deeper rings there.
    constexpr int RING = (KS == 3) ? (R * CTW == 1 ? (TX <= 4 ? 27 : 9) : (R * CTW == 2 ? 9 : 3)) : 5;      // (8-wide grids: 27 costs occupancy, 39.5 vs 36.4 us)
    constexpr int SLAB = (KS == 3) ? NTAP : KS * KS;   // taps unrolled per dynamic iteration
    static_assert(SLAB % RING == 0, "ring phase must be static");
    const int q_last = C::NG * NTAP - 1;
    auto tap_off = [](int kz, int ky, int kx) { return ((kz * C::LY + ky) * C::LX + kx) * C::VS; };

    // per-thread staging items: byte offset of channel group 0 inside the image, or kOOB (reads as zeros)
    unsigned soff[C::ITEMS];
#pragma unroll
    for (int it = 0; it < C::ITEMS; ++it) {
        const int item = it * C::NT + tid;
        const int u = item >> 2, q = item & 3;
        const int lz = u / (C::LY * C::LX), rem = u - lz * (C::LY * C::LX);
        const int ly = rem / C::LX, lx = rem - ly * C::LX;
        const int gz = iz0 + lz, gy = iy0 + ly, gx = ix0 + lx;
        const bool ok = (item < C::NV * 4) & (gz >= 0) & (gz < a.D) & (gy >= 0) & (gy < a.H) & (gx >= 0) & (gx < a.W);
        soff[it] = ok ? (unsigned)(((gz * a.H + gy) * a.W + gx) * CIN + q * 4) * 4u : kOOB;
    }
    auto commit = [&](const f32x4 (&stg)[C::ITEMS]) {
#pragma unroll
        for (int it = 0; it < C::ITEMS; ++it) {
            const int item = it * C::NT + tid;
            if (item < C::NV * 4) *reinterpret_cast<f32x4*>(lds + (item >> 2) * C::VS + (item & 3) * 4) = stg[it];
        }
    };

    if constexpr (KS == 3) {
        // ---- k3: software pipeline over the channel groups of the tile.  While group g is contracted (27 taps),
        //      the staging loads of group g+1 (one item per tap) and, in the last group, the residual rows are in
        //      flight; the weight ring runs continuously across groups.
        constexpr int NRES = R * CTW;                       // residual float4 per lane
        constexpr int RES0 = (27 - NRES) > 0 ? 27 - NRES : 0;
        static_assert(C::ITEMS <= 27 && NRES <= 27, "prefetch is spread over the tap sections");
        f32x4 stg[C::ITEMS];
#pragma unroll
        for (int it = 0; it < C::ITEMS; ++it) stg[it] = buf_load4(rin, soff[it], 0);
        f32x4 wf[RING][CTW];
#pragma unroll
        for (int r = 0; r < RING - 1; ++r)
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct) wf[r][ct] = buf_load4(rw, wlane, (unsigned)(min(r, q_last) * C::NCT + ct0 + ct) * 1024u);
        commit(stg);
        __syncthreads();

        // residual rows of this wave (prefetched during the last group)
        const int gzo = oz0 + w_z;
        const bool has_res = (a.flags & PCC_CONV_ADD) != 0;
        const __amdgpu_buffer_rsrc_t rres = make_rsrc(has_res ? a.res + (size_t)n * a.OD * a.OH * a.OW * COUT : a.in,
                                                      has_res ? (unsigned)a.OD * a.OH * a.OW * COUT * 4u : 0u);
        f32x4 resv[R][CTW];

#pragma unroll 1
        for (int g = 0; g < C::NG; ++g) {
            const unsigned gnext = (unsigned)min(g + 1, C::NG - 1) * 64u;     // last group: harmless re-read
            const bool last = g == C::NG - 1;
            f32x4 bb[2][R];
#pragma unroll
            for (int i = 0; i < R; ++i) bb[0][i] = *reinterpret_cast<const f32x4*>(lbase + i * ROW_OFF);
#pragma unroll
            for (int ts = 0; ts < NTAP; ++ts) {
                {
                    const int q = min(g * NTAP + ts + RING - 1, q_last);
#pragma unroll
                    for (int ct = 0; ct < CTW; ++ct)
                        wf[(ts + RING - 1) % RING][ct] = buf_load4(rw, wlane, (unsigned)(q * C::NCT + ct0 + ct) * 1024u);
                    const int tn = (ts + 1 < NTAP) ? ts + 1 : ts;   // last tap: harmless re-read
                    const int toff = tap_off(tn / 9, (tn / 3) % 3, tn % 3);
#pragma unroll
                    for (int i = 0; i < R; ++i)
                        bb[(ts + 1) & 1][i] = *reinterpret_cast<const f32x4*>(lbase + toff + i * ROW_OFF);
                    if (ts < C::ITEMS) stg[ts] = buf_load4(rin, soff[ts], gnext);
                    if (ts >= RES0 && ts < RES0 + NRES) {
                        const int i = (ts - RES0) / CTW, ct = (ts - RES0) % CTW;
                        const int gy = oy0 + ly0 + i * C::RY, gx = ox0 + lx0;
                        const bool ok = last & has_res & (gzo < a.OD) & (gy < a.OH) & (gx < a.OW);
                        const unsigned off = (unsigned)(((gzo * a.OH + gy) * a.OW + gx) * COUT + (ct0 + ct) * 16 + cq * 4) * 4u;
                        resv[i][ct] = buf_load4(rres, ok ? off : kOOB, 0);
                    }
                }
                // k-slot quarter j outermost: consecutive MFMAs go to different accumulators (the 40-cycle
                // dependent-accumulator latency of v_mfma_f32_16x16x4_f32 never stalls the 32-cycle issue)
                if constexpr (F16) {
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int ct = 0; ct < CTW; ++ct) acc[i][ct] = mfma16h(wf[ts % RING][ct], bb[ts & 1][i], acc[i][ct]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = 0; i < R; ++i)
#pragma unroll
                            for (int ct = 0; ct < CTW; ++ct)
                                acc[i][ct] = mfma16(wf[ts % RING][ct][j], bb[ts & 1][i][j], acc[i][ct]);
                }
                PCC_PIN_MEM_MFMA();
            }
            if (!last) {
                __syncthreads();   // every wave finished reading group g
                commit(stg);
                __syncthreads();
            }
        }
        // ---- epilogue (residual already in registers)
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int gy = oy0 + ly0 + i * C::RY, gx = ox0 + lx0;
            if (gzo < a.OD && gy < a.OH && gx < a.OW) {
                const size_t vox = (((size_t)n * a.OD + gzo) * a.OH + gy) * a.OW + gx;
#pragma unroll
                for (int ct = 0; ct < CTW; ++ct) {
                    f32x4 o = acc[i][ct];
                    const int c0 = (ct0 + ct) * 16 + cq * 4;
                    if (a.flags & PCC_CONV_BIAS) o += *reinterpret_cast<const f32x4*>(a.bias + c0);
                    if (a.flags & PCC_CONV_RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                    if (has_res) o += resv[i][ct];
                    if (a.flags & PCC_CONV_CLIP01) {
                        o.x = fminf(fmaxf(o.x, 0.f), 1.f); o.y = fminf(fmaxf(o.y, 0.f), 1.f);
                        o.z = fminf(fmaxf(o.z, 0.f), 1.f); o.w = fminf(fmaxf(o.w, 0.f), 1.f);
                    }
                    if (a.flags & PCC_CONV_OUT16) {      // (wave-uniform) fp16 hand-over, as store_out
                        h16x4 h;
                        h[0] = (_Float16)o.x; h[1] = (_Float16)o.y; h[2] = (_Float16)o.z; h[3] = (_Float16)o.w;
                        *reinterpret_cast<h16x4*>(reinterpret_cast<_Float16*>(a.out) + vox * a.ocs + a.oco + c0) = h;
                    } else {
                        *reinterpret_cast<f32x4*>(a.out + vox * a.ocs + a.oco + c0) = o;
                    }
                }
            }
        }
    } else {
#pragma unroll 1
        for (int g = 0; g < C::NG; ++g) {
            f32x4 stg[C::ITEMS];
#pragma unroll
            for (int it = 0; it < C::ITEMS; ++it) stg[it] = buf_load4(rin, soff[it], (unsigned)g * 64u);
            f32x4 wf[RING][CTW];
#pragma unroll
            for (int r = 0; r < RING - 1; ++r) {
                const int q = min(g * NTAP + r, q_last);
#pragma unroll
                for (int ct = 0; ct < CTW; ++ct) wf[r][ct] = buf_load4(rw, wlane, (unsigned)(q * C::NCT + ct0 + ct) * 1024u);
            }
            if (g > 0) __syncthreads();  // all waves finished reading the previous group
            commit(stg);
            __syncthreads();
#pragma unroll 1
            for (int sl = 0; sl < NTAP / SLAB; ++sl) {
#pragma unroll
                for (int ts = 0; ts < SLAB; ++ts) {
                    const int t = sl * SLAB + ts;
                    const int q = min(g * NTAP + t + RING - 1, q_last);
#pragma unroll
                    for (int ct = 0; ct < CTW; ++ct)
                        wf[(ts + RING - 1) % RING][ct] = buf_load4(rw, wlane, (unsigned)(q * C::NCT + ct0 + ct) * 1024u);
                    PCC_PIN_VMEM();
                    const int toff = tap_off(sl, ts / KS, ts % KS);
                    f32x4 b[R];
#pragma unroll
                    for (int i = 0; i < R; ++i) b[i] = *reinterpret_cast<const f32x4*>(lbase + toff + i * ROW_OFF);
                    if constexpr (F16) {
#pragma unroll
                        for (int i = 0; i < R; ++i)
#pragma unroll
                            for (int ct = 0; ct < CTW; ++ct) acc[i][ct] = mfma16h(wf[ts % RING][ct], b[i], acc[i][ct]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int i = 0; i < R; ++i)
#pragma unroll
                                for (int ct = 0; ct < CTW; ++ct)
                                    acc[i][ct] = mfma16(wf[ts % RING][ct][j], b[i][j], acc[i][ct]);
                    }
                }
            }
        }
        // ---- epilogue
        const int gz = oz0 + w_z;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int gy = oy0 + ly0 + i * C::RY, gx = ox0 + lx0;
            if (gz < a.OD && gy < a.OH && gx < a.OW) {
                const size_t vox = (((size_t)n * a.OD + gz) * a.OH + gy) * a.OW + gx;
#pragma unroll
                for (int ct = 0; ct < CTW; ++ct) store_out(a, acc[i][ct], vox, (ct0 + ct) * 16 + cq * 4, COUT);
            }
        }
    }
}

// =====================================================================================================
// Persistent 16 -> 16, k3, stride 1 kernel (the dominant layer shape: Conv3DTranspose 16->16 @64^3 is 47 % of
// all c3p MACs).  One workgroup per CU walks tiles g, g+G, ...:
//   * all 27 weight fragments live in registers for the lifetime of the workgroup (108 VGPRs) -> no weight
//     traffic and no vmcnt coupling inside the tap loop;
//   * the haloed input tile is double-buffered in LDS: the global loads of tile i+1 (and the residual of
//     tile i) are issued BEFORE the 432 MFMAs of tile i and land under them;
//   * one barrier per tile.
// Accumulation order per output element is identical to conv_fwd_kernel (tap-major, 4 k-slots): results are
// bit-identical between the two kernels.
// =====================================================================================================
template <int TZ, int TY, int R, int VS_>
struct P16Cfg {
    static constexpr int CIN = 16, COUT = 16, KS = 3;
    static constexpr int NW = TZ * (TY / R);
    static constexpr int NT = NW * 64;
    static constexpr int LZ = TZ + 2, LY = TY + 2, LX = 18;
    static constexpr int VS = VS_;   // floats per voxel in LDS: 24 = conflict-free, 20 = smaller (some 2-way conflicts)
    static constexpr int NV = LZ * LY * LX;
    static constexpr int ITEMS = (NV * 4 + NT - 1) / NT;
    static constexpr int BUF = ITEMS * NT / 4 * VS;   // floats per LDS buffer (rounded up: no store guards)
    static constexpr int LDS_BYTES = 2 * BUF * 4;
    static_assert(ITEMS + R <= 27, "prefetch is spread over the 27 tap sections");
};

template <int TZ, int TY, int R, int VS_, int WGS_PER_CU>
__global__ void __launch_bounds__((P16Cfg<TZ, TY, R, VS_>::NT), (WGS_PER_CU * P16Cfg<TZ, TY, R, VS_>::NT / 256)) conv16_pers_kernel(ConvArgs a, int ntiles) {
    using C = P16Cfg<TZ, TY, R, VS_>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int v = lane & 15, cq = lane >> 4;
    const int G = gridDim.x;
    int tile = xcd_remap(blockIdx.x, G);
    if (tile >= ntiles) return;

    const int w_yg = wave % (TY / R), w_z = wave / (TY / R);
    const int ly0 = w_yg * R;
    const int lane_off = ((w_z * C::LY + ly0) * C::LX + v) * C::VS + cq * 4;
    constexpr int ROW_OFF = C::LX * C::VS;

    // ---- all weights -> registers
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, 27u * 1024u);
    f32x4 wreg[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) wreg[t] = buf_load4(rw, lane * 16, t * 1024u);

    // ---- per-thread staging items: fixed (lz,ly,lx,quarter) -> relative byte offset inside an image, LDS slot
    const unsigned img_bytes = (unsigned)a.D * a.H * a.W * 64u;     // 16 channels x 4 B per voxel, in == out size
    unsigned rel[C::ITEMS];      // byte offset relative to the tile's (z-1, y-1, x-1) corner voxel
    unsigned lyx[C::ITEMS];      // ly | lx << 8 (for the y/x range test; z is covered by the buffer range check)
#pragma unroll
    for (int it = 0; it < C::ITEMS; ++it) {
        const int item = it * C::NT + tid;
        const int u = item >> 2, q = item & 3;
        const int lz = u / (C::LY * C::LX), rem = u - lz * (C::LY * C::LX);
        const int ly = rem / C::LX, lx = rem - ly * C::LX;
        rel[it] = (unsigned)(((lz * a.H + ly) * a.W + lx) * 16 + q * 4) * 4u;
        lyx[it] = (item < C::NV * 4) ? (unsigned)(ly | (lx << 8)) : 0xFFFFu;   // tail items: lx = 255 -> always out of range
    }

    // tile coordinates, advanced incrementally by G tiles per iteration (mixed radix, no divisions in the loop)
    int tx = tile % a.ntx, ty = (tile / a.ntx) % a.nty, tz = (tile / (a.ntx * a.nty)) % a.ntz, n = tile / (a.ntx * a.nty * a.ntz);
    const int gx_ = G % a.ntx, gy_ = (G / a.ntx) % a.nty, gz_ = (G / (a.ntx * a.nty)) % a.ntz, gn_ = G / (a.ntx * a.nty * a.ntz);

    struct Prefetch { __amdgpu_buffer_rsrc_t rin; unsigned base; int ylo, ny1, xlo, nx1; };
    auto setup = [&](int n_, int tz_, int ty_, int tx_) {
        Prefetch p;
        p.rin = make_rsrc(a.in + (size_t)n_ * a.D * a.H * a.W * 16, img_bytes);
        const int oz0 = tz_ * TZ, oy0 = ty_ * TY, ox0 = tx_ * 16;
        p.base = (unsigned)((((oz0 - 1) * a.H + (oy0 - 1)) * a.W + (ox0 - 1)) * 64);   // may wrap: unsigned arithmetic
        p.ylo = oy0 == 0 ? 1 : 0;                                   // valid local rows:    ylo <= ly <= ylo + ny1
        p.ny1 = min(C::LY, a.H - oy0 + 1) - p.ylo - 1;
        p.xlo = ox0 == 0 ? 1 : 0;                                   // valid local columns: xlo <= lx <= xlo + nx1
        p.nx1 = min(C::LX, a.W - ox0 + 1) - p.xlo - 1;
        return p;
    };
    auto load_item = [&](const Prefetch& p, int it) {
        // pure-VALU range test: a negative term sets the sign bit, and the sign bit IS the out-of-range offset.
        // z below 0 wraps to a huge offset, z >= D runs past the image: both hit the hardware range check.
        const int dy = (int)(lyx[it] & 0xFFu) - p.ylo, dx = (int)(lyx[it] >> 8) - p.xlo;
        const unsigned neg = (unsigned)(dy | (p.ny1 - dy) | dx | (p.nx1 - dx)) & kOOB;
        return buf_load4(p.rin, (p.base + rel[it]) | neg, 0);
    };
    auto commit = [&](float* buf, const f32x4 (&stg)[C::ITEMS]) {
#pragma unroll
        for (int it = 0; it < C::ITEMS; ++it) {
            const int item = it * C::NT + tid;
            *reinterpret_cast<f32x4*>(buf + (item >> 2) * C::VS + (item & 3) * 4) = stg[it];
        }
    };

    f32x4 stg[C::ITEMS];
    {
        const Prefetch p = setup(n, tz, ty, tx);
#pragma unroll
        for (int it = 0; it < C::ITEMS; ++it) stg[it] = load_item(p, it);
    }
    commit(lds, stg);
    __syncthreads();
    int cur = 0;
    const bool has_res = (a.flags & PCC_CONV_ADD) != 0;
    const unsigned out_img_bytes = (unsigned)a.OD * a.OH * a.OW * (unsigned)a.ocs * 4u;
    const f32x4 bias4 = (a.flags & PCC_CONV_BIAS) ? *reinterpret_cast<const f32x4*>(a.bias + cq * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wr_off = (tid >> 2) * C::VS + (tid & 3) * 4;      // LDS slot of this thread's staging item 0
    constexpr int WR_STRIDE = (C::NT / 4) * C::VS;               // floats between consecutive items
    constexpr int COMMIT_LAG = (C::ITEMS + 14 <= 27) ? 14 : 27 - C::ITEMS;                             // item k is loaded in tap k and written in tap k + LAG
    static_assert(C::ITEMS + COMMIT_LAG <= 27, "commit must fit in the tap loop");

    // deferred epilogue state of the PREVIOUS tile (its stores are issued inside this tile's tap loop)
    f32x4 pacc[R], pres[R];
    unsigned poff[R];                 // byte offset inside the output image, or kOOB (store dropped by hardware)
    __amdgpu_buffer_rsrc_t prout = make_rsrc(a.out, 0u);
#pragma unroll
    for (int i = 0; i < R; ++i) { pacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; pres[i] = pacc[i]; poff[i] = kOOB; }

    auto finish_row = [&](const f32x4& accv, const f32x4& resv_, unsigned off, __amdgpu_buffer_rsrc_t ro) {
        f32x4 o = accv + bias4;
        if (a.flags & PCC_CONV_RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        o += resv_;   // zeros when PCC_CONV_ADD is clear
        if (a.flags & PCC_CONV_CLIP01) {
            o.x = fminf(fmaxf(o.x, 0.f), 1.f); o.y = fminf(fmaxf(o.y, 0.f), 1.f);
            o.z = fminf(fmaxf(o.z, 0.f), 1.f); o.w = fminf(fmaxf(o.w, 0.f), 1.f);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ro, (int)off, 0, 0);
    };

    for (;;) {
        const int next = tile + G;
        const bool has_next = next < ntiles;
        int ntx_ = tx + gx_, nty_ = ty + gy_, ntz_ = tz + gz_, nn_ = n + gn_;
        if (ntx_ >= a.ntx) { ntx_ -= a.ntx; ++nty_; }
        if (nty_ >= a.nty) { nty_ -= a.nty; ++ntz_; }
        if (ntz_ >= a.ntz) { ntz_ -= a.ntz; ++nn_; }
        const Prefetch pn = has_next ? setup(nn_, ntz_, nty_, ntx_) : setup(n, tz, ty, tx);   // no next: harmless re-read
        const int oz0 = tz * TZ, oy0 = ty * TY, ox0 = tx * 16;
        const int gz = oz0 + w_z, gx = ox0 + v;
        const unsigned lvox0 = (unsigned)((gz * a.OH + oy0 + ly0) * a.OW + gx);
        const __amdgpu_buffer_rsrc_t rres = make_rsrc(has_res ? a.res + (size_t)n * a.OD * a.OH * a.OW * 16 : a.in, has_res ? img_bytes : 0u);
        const __amdgpu_buffer_rsrc_t rout = make_rsrc(a.out + (size_t)n * a.OD * a.OH * a.OW * a.ocs, out_img_bytes);
        const bool col_ok = gz < a.OD && gx < a.OW;
        f32x4 resv[R];

        const float* lbase = lds + cur * C::BUF + lane_off;
        float* wbase = lds + (cur ^ 1) * C::BUF + wr_off;
        f32x4 acc[R];
#pragma unroll
        for (int i = 0; i < R; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 bb[2][R];
#pragma unroll
        for (int i = 0; i < R; ++i) bb[0][i] = *reinterpret_cast<const f32x4*>(lbase + i * ROW_OFF);
#pragma unroll
        for (int ts = 0; ts < 27; ++ts) {
            // Everything that is not an MFMA is spread over the 27 tap sections and interleaved with the 16 MFMAs
            // of the section (sched_group_barrier below), so the matrix pipe never waits for the issue of:
            //   rows of tap ts+1 (4 ds_read) | staging load k of the NEXT tile (ts = k < ITEMS) and its LDS commit
            //   (ts = k + LAG) | residual load of THIS tile (ts = ITEMS..ITEMS+R-1) | the epilogue row of the PREVIOUS
            //   tile (ts = 27-R..26, branch-free buffer store).
            const int tn = (ts + 1 < 27) ? ts + 1 : ts;
            const int toff = (((tn / 9) * C::LY + (tn / 3) % 3) * C::LX + tn % 3) * C::VS;
#pragma unroll
            for (int i = 0; i < R; ++i)
                bb[(ts + 1) & 1][i] = *reinterpret_cast<const f32x4*>(lbase + toff + i * ROW_OFF);
            if (ts < C::ITEMS) stg[ts] = load_item(pn, ts);
            if (ts >= COMMIT_LAG && ts < C::ITEMS + COMMIT_LAG)
                *reinterpret_cast<f32x4*>(wbase + (ts - COMMIT_LAG) * WR_STRIDE) = stg[ts - COMMIT_LAG];
            if (ts >= C::ITEMS && ts < C::ITEMS + R) {
                const int i = ts - C::ITEMS;
                const bool ok = col_ok && (oy0 + ly0 + i) < a.OH && has_res;
                resv[i] = buf_load4(rres, ok ? ((lvox0 + (unsigned)(i * a.OW)) * 16 + cq * 4) * 4u : kOOB, 0);
            }
            if (ts >= 27 - R) finish_row(pacc[ts - (27 - R)], pres[ts - (27 - R)], poff[ts - (27 - R)], prout);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < R; ++i) acc[i] = mfma16(wreg[ts][j], bb[ts & 1][i][j], acc[i]);
            // issue order inside the section: 1 LDS read, 4 MFMA, ... (VMEM / LDS write / VALU wherever they fit)
#pragma unroll
            for (int k = 0; k < R; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // 4 MFMA
            }
            PCC_PIN_MEM_MFMA();
        }

        // hand the finished accumulators over to the deferred epilogue of the next iteration
#pragma unroll
        for (int i = 0; i < R; ++i) {
            pacc[i] = acc[i];
            pres[i] = resv[i];
            const bool ok = col_ok && (oy0 + ly0 + i) < a.OH;
            poff[i] = ok ? ((lvox0 + (unsigned)(i * a.OW)) * (unsigned)a.ocs + a.oco + cq * 4) * 4u : kOOB;
        }
        prout = rout;
        if (!has_next) break;
        __syncthreads();   // every wave's commits of the next tile are in LDS; nobody still reads `cur`
        cur ^= 1;
        tile = next; n = nn_; tz = ntz_; ty = nty_; tx = ntx_;
    }
    // epilogue of the last tile
#pragma unroll
    for (int i = 0; i < R; ++i) finish_row(pacc[i], pres[i], poff[i], prout);
}

// =====================================================================================================
// transposed conv, stride 2: parity decomposition on the INPUT (base) grid.
//   out[2b + p] = sum over taps kappa = (p + PL) mod 2 (step 2) of W[kappa] * in[b + (p + PL - kappa)/2]
//   PL = SAME pad_low of the adjoint forward conv = (KS-2)/2.
// =====================================================================================================
template <int KS>
struct Tr2Geo {
    static constexpr int PL = (KS - 2) / 2;
    // delta range over both parities: kappa in [0,KS): delta = (p + PL - kappa)/2
    static constexpr int HL = (KS - 1 - PL) / 2;      // max(-delta)  (kappa = KS-1 or KS-2)
    static constexpr int HH = (1 + PL) / 2;           // max(+delta)  (p = 1, kappa = 0 or 1)
};

template <int CIN, int COUT, int KS, int TX, int TZ, int TY, int TXT, int R, int CTW = COUT / 16>
struct Tr2Cfg {
    using G = Tr2Geo<KS>;
    static constexpr int NG = CIN / 16, NCT = COUT / 16;
    static constexpr int RY = 16 / TX;
    static constexpr int NYB = TY / RY, NXB = TXT / TX;
    static constexpr int NCG = NCT / CTW;
    static constexpr int NW = TZ * (NYB / R) * NXB * NCG;
    static constexpr int NT = NW * 64;
    static constexpr int LZ = TZ + G::HL + G::HH, LY = TY + G::HL + G::HH, LX = TXT + G::HL + G::HH;
    static constexpr int VS = CIN + 8;
    static constexpr int NV = LZ * LY * LX;
    static constexpr int LDS_BYTES = NV * VS * 4;
    static constexpr int Q = CIN / 4;  // float4 per voxel
    static constexpr int ITEMS = (NV * Q + NT - 1) / NT;
};

template <int CIN, int COUT, int KS, int TX, int TZ, int TY, int TXT, int R, int CTW = COUT / 16, bool F16 = false>
__global__ void __launch_bounds__((Tr2Cfg<CIN, COUT, KS, TX, TZ, TY, TXT, R, CTW>::NT))
conv_tr2_kernel(ConvArgs a) {
    using C = Tr2Cfg<CIN, COUT, KS, TX, TZ, TY, TXT, R, CTW>;
    using G = Tr2Geo<KS>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int v = lane & 15, cq = lane >> 4;

    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = t % a.ntx; t /= a.ntx;
    const int ty = t % a.nty; t /= a.nty;
    const int tz = t % a.ntz;
    const int n = t / a.ntz;
    const int bz0 = tz * TZ, by0 = ty * TY, bx0 = tx * TXT;  // base (input-grid) tile origin

    int wv = wave;
    const int ct0 = (wv % C::NCG) * CTW; wv /= C::NCG;
    const int w_xb = wv % C::NXB; wv /= C::NXB;
    const int w_yg = wv % (C::NYB / R);
    const int w_z = wv / (C::NYB / R);
    const int ry = v / TX, rx = v % TX;
    const int ly0 = w_yg * R * C::RY + ry, lx0 = w_xb * TX + rx;
    const float* lbase = lds + (((w_z + G::HL) * C::LY + ly0 + G::HL) * C::LX + lx0 + G::HL) * C::VS + cq * 4;
    constexpr int ROW_OFF = C::RY * C::LX * C::VS;

    // ---- stage the whole haloed tile, all channels (buffer loads: out-of-range voxels read as zeros)
    const float* inb = a.in + (size_t)n * a.D * a.H * a.W * CIN;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(inb, (unsigned)a.D * a.H * a.W * CIN * 4u);
#pragma unroll 1
    for (int it0 = 0; it0 < C::ITEMS; it0 += 8) {
        f32x4 stg[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int item = (it0 + k) * C::NT + tid;
            const int u = item / C::Q, q = item - u * C::Q;
            const int lz = u / (C::LY * C::LX), rem = u - lz * (C::LY * C::LX);
            const int ly = rem / C::LX, lx = rem - ly * C::LX;
            const int gz = bz0 - G::HL + lz, gy = by0 - G::HL + ly, gx = bx0 - G::HL + lx;
            const bool ok = (item < C::NV * C::Q) && gz >= 0 && gz < a.D && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            const unsigned off = (unsigned)(((gz * a.H + gy) * a.W + gx) * CIN + q * 4) * 4u;
            stg[k] = buf_load4(rin, ok ? off : kOOB, 0);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int item = (it0 + k) * C::NT + tid;
            const int u = item / C::Q, q = item - u * C::Q;
            if (item < C::NV * C::Q) *reinterpret_cast<f32x4*>(lds + u * C::VS + q * 4) = stg[k];
        }
    }

    // weights are packed in consumption order [class][tap in class][g][ct]
    constexpr int NSEQ = KS * KS * KS * C::NG;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, (unsigned)(NSEQ * C::NCT) * 1024u);
    const unsigned wlane = lane * 16;
    const int gzb = bz0 + w_z;
    // FULL: the whole (class, tap, g) sequence is unrolled so that a static 3-deep register ring prefetches two
    // units ahead.  For the widest shape (64 -> 64: 3456 MFMAs per wave) that would not fit the instruction
    // cache, so the cin-group loop stays dynamic there and the weights are loaded per (tap, group).
    // (round 3) the one-row, one-cout-tile configuration of the 4-wide grids is small enough to unroll whatever the width (432
    // MFMAs per wave) and needs a deep ring: a unit is 4 MFMAs = 128 cycles there, and loading per (tap, group) on demand left the
    // 64 -> 64 4^3 -> 8^3 layer at 53 us for 6 us of MFMAs.
    constexpr bool FULL = (C::NG * CTW < 16 && C::NG * C::NCT < 16) || R * CTW == 1;
    constexpr int RING = R * CTW == 1 ? 12 : 3;
    f32x4 wf[RING][CTW];
    if constexpr (FULL) {
#pragma unroll
        for (int r = 0; r < RING - 1; ++r)
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct) wf[r][ct] = buf_load4(rw, wlane, (unsigned)(r * C::NCT + ct0 + ct) * 1024u);
    }
    __syncthreads();

    int seq = 0;  // compile-time after unrolling
#pragma unroll
    for (int pz = 0; pz < 2; ++pz)
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                f32x4 acc[R][CTW];
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int ct = 0; ct < CTW; ++ct) acc[i][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kz = (pz + G::PL) & 1; kz < KS; kz += 2)
#pragma unroll
                    for (int ky = (py + G::PL) & 1; ky < KS; ky += 2)
#pragma unroll
                        for (int kx = (px + G::PL) & 1; kx < KS; kx += 2) {
                            const int dz = (pz + G::PL - kz) / 2, dy = (py + G::PL - ky) / 2, dx = (px + G::PL - kx) / 2;
                            const int toff = ((dz * C::LY + dy) * C::LX + dx) * C::VS;
                            if constexpr (FULL) {
#pragma unroll
                                for (int g = 0; g < C::NG; ++g, ++seq) {
                                    {
                                        const int qn = (seq + RING - 1 < NSEQ) ? seq + RING - 1 : NSEQ - 1;
#pragma unroll
                                        for (int ct = 0; ct < CTW; ++ct)
                                            wf[(seq + RING - 1) % RING][ct] = buf_load4(rw, wlane, (unsigned)(qn * C::NCT + ct0 + ct) * 1024u);
                                        PCC_PIN_VMEM();
                                    }
                                    f32x4 b[R];
#pragma unroll
                                    for (int i = 0; i < R; ++i)
                                        b[i] = *reinterpret_cast<const f32x4*>(lbase + toff + i * ROW_OFF + g * 16);
                                    if constexpr (F16) {
#pragma unroll
                                        for (int i = 0; i < R; ++i)
#pragma unroll
                                            for (int ct = 0; ct < CTW; ++ct) acc[i][ct] = mfma16h(wf[seq % RING][ct], b[i], acc[i][ct]);
                                    } else {
#pragma unroll
                                    for (int j = 0; j < 4; ++j)
#pragma unroll
                                        for (int i = 0; i < R; ++i)
#pragma unroll
                                            for (int ct = 0; ct < CTW; ++ct)
                                                acc[i][ct] = mfma16(wf[seq % RING][ct][j], b[i][j], acc[i][ct]);
                                    }
                                }
                            } else {
                                const int seq0 = seq;
                                seq += C::NG;
#pragma unroll 1
                                for (int g = 0; g < C::NG; ++g) {
                                    f32x4 w1[CTW];
#pragma unroll
                                    for (int ct = 0; ct < CTW; ++ct)
                                        w1[ct] = buf_load4(rw, wlane, (unsigned)((seq0 + g) * C::NCT + ct0 + ct) * 1024u);
                                    f32x4 b[R];
#pragma unroll
                                    for (int i = 0; i < R; ++i)
                                        b[i] = *reinterpret_cast<const f32x4*>(lbase + toff + i * ROW_OFF + g * 16);
                                    if constexpr (F16) {
#pragma unroll
                                        for (int i = 0; i < R; ++i)
#pragma unroll
                                            for (int ct = 0; ct < CTW; ++ct) acc[i][ct] = mfma16h(w1[ct], b[i], acc[i][ct]);
                                    } else {
#pragma unroll
                                    for (int j = 0; j < 4; ++j)
#pragma unroll
                                        for (int i = 0; i < R; ++i)
#pragma unroll
                                            for (int ct = 0; ct < CTW; ++ct)
                                                acc[i][ct] = mfma16(w1[ct][j], b[i][j], acc[i][ct]);
                                    }
                                }
                            }
                        }
                // epilogue of this parity class
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int gyb = by0 + ly0 + i * C::RY, gxb = bx0 + lx0;
                    if (gzb < a.D && gyb < a.H && gxb < a.W) {
                        const size_t vox = (((size_t)n * a.OD + 2 * gzb + pz) * a.OH + 2 * gyb + py) * a.OW + 2 * gxb + px;
#pragma unroll
                        for (int ct = 0; ct < CTW; ++ct) store_out(a, acc[i][ct], vox, (ct0 + ct) * 16 + cq * 4, COUT);
                    }
                }
            }
}

// =====================================================================================================
// transposed conv, stride 2, k = 3: channel-group pipelined variant (conv_tr2g_kernel).
//   Same parity decomposition, but the loop nest is  cin group (16 ch) -> parity class -> tap:
//   * one 16-channel group of the haloed input tile is staged at a time, global -> LDS directly
//     (buffer_load ... lds: no staging registers, no ds_write, no per-item index math in the loop) into a double
//     buffer; group g+1 is in flight while group g feeds the MFMAs;
//   * the accumulators of ALL 8 parity classes stay live across the groups (8 x R x CTW float4);
//   * the epilogue uses per-row offsets computed once and one buffer descriptor per parity class.
// =====================================================================================================
template <int CIN, int COUT, int TX, int TZ, int TY, int TXT, int R, int CTW>
struct Tr2gCfg {
    static constexpr int NG = CIN / 16, NCT = COUT / 16;
    static constexpr int RY = 16 / TX;
    static constexpr int NYB = TY / RY, NXB = TXT / TX;
    static constexpr int NCG = NCT / CTW;
    static constexpr int NW = TZ * (NYB / R) * NXB * NCG;
    static constexpr int NT = NW * 64;
    static constexpr int LZ = TZ + 1, LY = TY + 1, LX = TXT + 1;     // k3: taps reach b-1 only
    static constexpr int VSQ = 5;                                     // 16-byte slots per voxel: 4 data + 1 pad (80 B stride)
    static constexpr int VS = VSQ * 4;
    static constexpr int NV = LZ * LY * LX;
    static constexpr int CHUNKS = (NV * VSQ + 63) / 64;               // 1 KB wave-sized chunks of one group image
    static constexpr int ITEMS = (CHUNKS + NW - 1) / NW;              // chunks per wave
    static constexpr int BUF_BYTES = ITEMS * NW * 1024;
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;
};

// tap seq (0..26) of the parity-class order [class (pz,py,px)][kz][ky][kx]: its class, input offsets (0 / -1 per dim), and
// whether it is the first / last tap of its class.  Evaluated at compile time (seq is a constant of the unrolled loops).
struct Tr2gTap { int cls, dz, dy, dx; bool first, last; };
__host__ __device__ constexpr Tr2gTap tr2g_tap(int want) {
    int seq = 0;
    for (int cls = 0; cls < 8; ++cls) {
        const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
        const int ntap = (pz ? 1 : 2) * (py ? 1 : 2) * (px ? 1 : 2);      // even outputs take taps 0 and 2, odd ones tap 1
        int t = 0;
        for (int kz = pz; kz < 3; kz += 2)
            for (int ky = py; ky < 3; ky += 2)
                for (int kx = px; kx < 3; kx += 2, ++seq, ++t)
                    if (seq == want) return Tr2gTap{cls, (pz - kz) / 2, (py - ky) / 2, (px - kx) / 2, t == 0, t == ntap - 1};
    }
    return Tr2gTap{0, 0, 0, 0, false, false};
}

template <int... I, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

// EPI: the epilogue is compiled for the layer's flags (the per-store flag tests of a generic epilogue cost this kernel more
// scalar registers and branches than it has to spare): 0 = bias / ReLU, fp32 store (every stride-2 transposed layer of the c*
// graphs); 1 = bias / ReLU, fp16 store (PCC_CONV_OUT16, the fp16 mode); 2 = any flags (residual, clip), tested at run time.
enum { TR2G_EPI_F32 = 0, TR2G_EPI_F16 = 1, TR2G_EPI_ANY = 2 };

template <int CIN, int COUT, int TX, int TZ, int TY, int TXT, int R, int CTW, bool F16 = false, int EPI = TR2G_EPI_F32>
__global__ void __launch_bounds__((Tr2gCfg<CIN, COUT, TX, TZ, TY, TXT, R, CTW>::NT), 2)   // two waves per SIMD: <= 256 registers
conv_tr2g_kernel(ConvArgs a, int ntiles) {
    using C = Tr2gCfg<CIN, COUT, TX, TZ, TY, TXT, R, CTW>;
    static_assert(C::NG % 2 == 0, "the LDS double buffer alternates per cin group across tiles");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int v = lane & 15, cq = lane >> 4;
    const int G = gridDim.x;
    int tile = xcd_remap(blockIdx.x, G);
    if (tile >= ntiles) return;

    int wv = wave;
    const int ct0 = (wv % C::NCG) * CTW; wv /= C::NCG;
    const int w_xb = wv % C::NXB; wv /= C::NXB;
    const int w_yg = wv % (C::NYB / R);
    const int w_z = wv / (C::NYB / R);
    const int ry = v / TX, rx = v % TX;
    const int ly0 = w_yg * R * C::RY + ry, lx0 = w_xb * TX + rx;
    // B operand base: voxel (w_z+1, ly0+1, lx0+1) of the haloed tile, channel quad cq
    const int lane_off = (((w_z + 1) * C::LY + ly0 + 1) * C::LX + lx0 + 1) * C::VS + cq * 4;
    constexpr int ROW_OFF = C::RY * C::LX * C::VS;

    // ---- staging items of this lane (fixed for the life of the workgroup): the tile-local voxel packed in 10-bit fields
    //      (lz | ly << 10 | lx << 20; pad slots carry an lz that fails every range test) and its byte offset relative to the
    //      tile's first haloed voxel.  Per tile the range test of all three dims is two packed adds: with a bias of 512 per
    //      field, bit 9 of (p + lo) says g >= 0 and bit 9 of (hi - p) says g <= dim - 1 (fields neither carry nor borrow:
    //      coordinates and dims stay below 256 -- the launcher checks).
    unsigned pk[C::ITEMS], rel[C::ITEMS];
    {
        const int HWc = a.H * a.W * CIN * 4, Wc = a.W * CIN * 4;
#pragma unroll
        for (int it = 0; it < C::ITEMS; ++it) {
            const int slot = (wave * C::ITEMS + it) * 64 + lane;
            const int u = slot / C::VSQ, q = slot - u * C::VSQ;
            const int lz = u / (C::LY * C::LX), rem = u - lz * (C::LY * C::LX);
            const int ly = rem / C::LX, lx = rem - ly * C::LX;
            const bool data = u < C::NV && q < 4;
            pk[it] = data ? (unsigned)(lz | (ly << 10) | (lx << 20)) : 511u;
            rel[it] = data ? (unsigned)(lz * HWc + ly * Wc + lx * (CIN * 4) + q * 16) : 0u;
        }
    }
    constexpr unsigned kBit9 = (1u << 9) | (1u << 19) | (1u << 29);
    const unsigned in_bytes = (unsigned)a.D * a.H * a.W * CIN * 4u;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // global -> LDS of cin group g of tile (n, bz0, by0, bx0): zeros land in LDS for SAME padding, tile overhang, pad slots
    auto stage_group = [&](int n_, int bz0, int by0, int bx0, int g, int buf) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rg = make_rsrc(a.in + (size_t)n_ * a.D * a.H * a.W * CIN + g * 16, in_bytes - (unsigned)g * 64u);
        const int HWc = a.H * a.W * CIN * 4, Wc = a.W * CIN * 4;
        const unsigned lo = (unsigned)((bz0 - 1 + 512) | ((by0 - 1 + 512) << 10) | ((bx0 - 1 + 512) << 20));
        const unsigned hi = (unsigned)((a.D - bz0 + 512) | ((a.H - by0 + 512) << 10) | ((a.W - bx0 + 512) << 20));
        const unsigned base = (unsigned)((bz0 - 1) * HWc + (by0 - 1) * Wc + (bx0 - 1) * (CIN * 4));     // (may wrap below 0: rel brings it back)
#pragma unroll
        for (int it = 0; it < C::ITEMS; ++it) {
            const bool ok = (((pk[it] + lo) & (hi - pk[it])) & kBit9) == kBit9;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lds_ptr)((char*)lds + buf * C::BUF_BYTES + (wave * C::ITEMS + it) * 1024), 16,
                                                     (int)(ok ? base + rel[it] : kOOB), 0, 0, 0);
        }
    };
    auto decode = [&](int t, int& n_, int& bz0, int& by0, int& bx0) {
        const int tx = t % a.ntx; t /= a.ntx;
        const int ty = t % a.nty; t /= a.nty;
        bz0 = (t % a.ntz) * TZ; by0 = ty * TY; bx0 = tx * TXT; n_ = t / a.ntz;
    };

    int n, bz0, by0, bx0;
    decode(tile, n, bz0, by0, bx0);
    stage_group(n, bz0, by0, bx0, 0, 0);

    // weights packed in consumption order [g][class][tap in class][ct]
    constexpr int NSEQ = 27;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, (unsigned)(NSEQ * C::NG * C::NCT) * 1024u);
    const unsigned wlane = lane * 16;
    // vmcnt retires in order: a weight load issued AFTER the staging loads of the next group cannot be consumed before
    // those have landed (HBM latency).  A deep ring (8 taps ahead = ~4500 cycles) keeps that wait out of the tap loop.
    constexpr int RING = (CTW == 1) ? 9 : 3;
    static_assert(NSEQ % RING == 0, "the weight ring position must repeat per group");
    f32x4 wf[RING][CTW];
    const size_t ovox_n = (size_t)a.OD * a.OH * a.OW;
    const bool any_res = EPI == TR2G_EPI_ANY && (a.flags & PCC_CONV_ADD) != 0;
    const bool any_out16 = EPI == TR2G_EPI_ANY && (a.flags & PCC_CONV_OUT16) != 0;
    const bool out16 = EPI == TR2G_EPI_F16 || any_out16;
    const float relu_lo = (a.flags & PCC_CONV_RELU) ? 0.f : -__builtin_inff();
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 bias4[CTW];
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct)
        bias4[ct] = (a.flags & PCC_CONV_BIAS) ? *reinterpret_cast<const f32x4*>(a.bias + (ct0 + ct) * 16 + cq * 4) : zero4;

    // The weight stream wraps around: the last group of a tile prefetches the first RING - 1 taps of the NEXT tile (same
    // weights).  Restarting it at the top of a tile would put those loads behind the tile's last stores, and the in-order
    // vmcnt wait at the first barrier would then wait for the store acknowledgements.
#pragma unroll
    for (int r = 0; r < RING - 1; ++r)
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) wf[r][ct] = buf_load4(rw, wlane, (unsigned)(r * C::NCT + ct0 + ct) * 1024u);
    f32x4 acc[8][R][CTW];
#pragma unroll 1
    for (;;) {
        const int next = tile + G;
        const bool has_next = next < ntiles;
        int nn = n, nbz0 = bz0, nby0 = by0, nbx0 = bx0;
        if (has_next) decode(next, nn, nbz0, nby0, nbx0);
        // per-row output BYTE offsets of this tile for class (0,0,0) (the epilogue of a parity class runs inside the last
        // group, right after the class's taps: the 8 x R x CTW stores of a tile are spread over that group instead of bursting
        // at its end).  One descriptor per tile (image n); the class adds a wave-uniform offset.  kOOB plus those offsets
        // stays beyond the descriptor's range (the launcher admits images below 2^31 bytes only).
        const int gzb = bz0 + w_z, gxb = bx0 + lx0;
        const unsigned esz = out16 ? 2u : 4u;
        unsigned ooff[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int gyb = by0 + ly0 + i * C::RY;
            const bool ok = gzb < a.D && gyb < a.H && gxb < a.W;
            const unsigned vox = (unsigned)((2 * gzb * a.OH + 2 * gyb) * a.OW + 2 * gxb);
            ooff[i] = ok ? (vox * (unsigned)a.ocs + (unsigned)a.oco + cq * 4) * esz : kOOB;
        }
        const __amdgpu_buffer_rsrc_t rout = make_rsrc((const char*)a.out + (size_t)n * ovox_n * a.ocs * esz, (unsigned)(ovox_n * a.ocs * esz));
        // residual (EPI_ANY only; no layer of the c* graphs adds one to a stride-2 transposed conv): offsets computed on demand
        const __amdgpu_buffer_rsrc_t rres = make_rsrc(any_res ? a.res + (size_t)n * ovox_n * COUT : a.in, any_res ? (unsigned)(ovox_n * COUT * 4) : 0u);
        auto roff_of = [&](int i) -> unsigned {
            const int gyb = by0 + ly0 + i * C::RY;
            const bool ok = gzb < a.D && gyb < a.H && gxb < a.W;
            return ok ? ((unsigned)((2 * gzb * a.OH + 2 * gyb) * a.OW + 2 * gxb) * (unsigned)COUT + cq * 4) * 4u : kOOB;
        };

        // one cin group of the tile.  FIRST: the accumulators start from 0 (srcC = inline 0, no zero-init pass);
        // LAST: each parity class is finished and stored right after its taps.
        auto group = [&](auto first_tag, auto last_tag, int g) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
            // group g has landed in LDS (this wave's loads: vmcnt; the other waves': barrier); nobody reads the other buffer any more
            __builtin_amdgcn_s_waitcnt(0x0F70 | ((RING - 1) * CTW));    // vmcnt(weights still in flight) expcnt(7) lgkmcnt(15)
            __syncthreads();
            if (!LAST) stage_group(n, bz0, by0, bx0, g + 1, (g + 1) & 1);
            else if (has_next) stage_group(nn, nbz0, nby0, nbx0, 0, 0);          // next tile's first group under this tile's last
            const float* lbase = lds + (g & 1) * (C::BUF_BYTES / 4) + lane_off;
            const unsigned wg_off = (unsigned)(g * NSEQ * C::NCT) * 1024u;
            // B operands are read one tap ahead (double buffer): the LDS latency of tap t + 1 runs under the MFMAs of tap t
            f32x4 b[2][R];
#pragma unroll
            for (int i = 0; i < R; ++i) b[0][i] = *reinterpret_cast<const f32x4*>(lbase + i * ROW_OFF);       // tap 0: class (0,0,0), no offset
            static_for(std::make_integer_sequence<int, NSEQ>{}, [&](auto seq_tag) __attribute__((always_inline)) {
                constexpr int seq = decltype(seq_tag)::value;
                constexpr Tr2gTap T = tr2g_tap(seq);
                constexpr Tr2gTap Tn = tr2g_tap(seq + 1 < NSEQ ? seq + 1 : seq);
                constexpr int next_off = ((Tn.dz * C::LY + Tn.dy) * C::LX + Tn.dx) * C::VS;
                constexpr int cls = T.cls;
                // weights RING - 1 taps ahead: runs into the next group's first taps, and from the last group into the next tile's
                constexpr bool wrap = LAST && seq + RING - 1 >= NSEQ;
#pragma unroll
                for (int ct = 0; ct < CTW; ++ct)
                    wf[(seq + RING - 1) % RING][ct] = buf_load4(rw, wlane, (wrap ? 0u : wg_off) + (unsigned)((seq + RING - 1 - (wrap ? NSEQ : 0)) * C::NCT + ct0 + ct) * 1024u);
                if constexpr (seq + 1 < NSEQ) {
#pragma unroll
                    for (int i = 0; i < R; ++i) b[(seq + 1) & 1][i] = *reinterpret_cast<const f32x4*>(lbase + next_off + i * ROW_OFF);
                }
                PCC_PIN_MEM_MFMA();
                constexpr bool open = FIRST && T.first;  // first tap of the class in the first group: start from the bias
                if constexpr (F16) {
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int ct = 0; ct < CTW; ++ct) acc[cls][i][ct] = mfma16h(wf[seq % RING][ct], b[seq & 1][i], open ? bias4[ct] : acc[cls][i][ct]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = 0; i < R; ++i)
#pragma unroll
                            for (int ct = 0; ct < CTW; ++ct)
                                acc[cls][i][ct] = mfma16(wf[seq % RING][ct][j], b[seq & 1][i][j], (open && j == 0) ? bias4[ct] : acc[cls][i][ct]);
                }
                if constexpr (LAST && T.last) {      // this class is complete: ReLU (/ residual / clip), stores (the bias is already in)
                    constexpr int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
                    const unsigned coff = (unsigned)(((pz * a.OH + py) * a.OW + px) * a.ocs) * esz;  // (wave-uniform) bytes
#pragma unroll
                    for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
                        for (int i = 0; i < R; ++i) {
                            f32x4 o = acc[cls][i][ct];
                            // one v_maximum3_f32 per element (fmaxf on a raw MFMA result costs a second v_max that quiets NaNs)
                            o = __builtin_elementwise_maximum(o, (f32x4){relu_lo, relu_lo, relu_lo, relu_lo});
                            if constexpr (EPI == TR2G_EPI_ANY) {
                                if (any_res) o += buf_load4(rres, roff_of(i), (unsigned)((((pz * a.OH + py) * a.OW + px) * COUT + (ct0 + ct) * 16) * 4));
                                if (a.flags & PCC_CONV_CLIP01) {
#pragma unroll
                                    for (int c = 0; c < 4; ++c) o[c] = fminf(fmaxf(o[c], 0.f), 1.f);
                                }
                            }
                            const unsigned boff = ooff[i] + coff + (unsigned)((ct0 + ct) * 16) * esz;
                            if (EPI == TR2G_EPI_F16 || (EPI == TR2G_EPI_ANY && any_out16)) {
                                h16x4 oh;
#pragma unroll
                                for (int c = 0; c < 4; ++c) oh[c] = (_Float16)o[c];
                                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, oh), rout, (int)boff, 0, 0);
                            } else {
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rout, (int)boff, 0, 0);
                            }
                        }
                }
            });
        };
        group(std::true_type{}, std::false_type{}, 0);
#pragma unroll 1
        for (int g = 1; g < C::NG - 1; ++g) group(std::false_type{}, std::false_type{}, g);
        group(std::false_type{}, std::true_type{}, C::NG - 1);

        if (!has_next) break;
        tile = next; n = nn; bz0 = nbz0; by0 = nby0; bx0 = nbx0;
    }
}

// =====================================================================================================
// forward conv with Cin = 1, stride 2 (first layer of every analysis transform).
//   k-slots of each MFMA = 4 consecutive taps along x (x taps padded to a multiple of 4 with zero
//   weights), so the B operand is one ds_read_b32 with an immediate offset.
//   tile = TZ x TY x 16 output voxels; wave owns R rows consecutive in y.
// =====================================================================================================
template <int COUT, int KS, int TZ, int TY, int R>
struct Cin1Cfg {
    static constexpr int S = 2;
    static constexpr int NCT = COUT / 16;
    static constexpr int KXG = (KS + 3) / 4;  // groups of 4 x-taps
    static constexpr int NW = TZ * (TY / R);
    static constexpr int NT = NW * 64;
    static constexpr int PL = (KS - 2) / 2;
    static constexpr int LZ = (TZ - 1) * S + KS, LY = (TY - 1) * S + KS;
    static constexpr int LXU = 15 * S + KXG * 4;             // x extent actually addressed
    static constexpr int LX = (LXU | 1);                     // odd row stride: fewer bank conflicts
    static constexpr int NV = LZ * LY * LX;
    static constexpr int LDS_BYTES = NV * 4;
    static constexpr int ITEMS = (NV + NT - 1) / NT;
};

template <int COUT, int KS, int TZ, int TY, int R>
__global__ void __launch_bounds__((Cin1Cfg<COUT, KS, TZ, TY, R>::NT)) conv_cin1_kernel(ConvArgs a) {
    using C = Cin1Cfg<COUT, KS, TZ, TY, R>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int v = lane & 15, kq = lane >> 4;

    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = t % a.ntx; t /= a.ntx;
    const int ty = t % a.nty; t /= a.nty;
    const int tz = t % a.ntz;
    const int n = t / a.ntz;
    const int oz0 = tz * TZ, oy0 = ty * TY, ox0 = tx * 16;
    const int iz0 = oz0 * 2 - C::PL, iy0 = oy0 * 2 - C::PL, ix0 = ox0 * 2 - C::PL;

    const float* inb = a.in + (size_t)n * a.D * a.H * a.W;
    if constexpr (KS == 3) {
        // k3: no low-side halo (PL = 0) and ix0 = 32 tx, so a row of the tile is 9 aligned float4 of one image row: all loads of a
        // thread are in flight together (the scalar loop below spent ~50 VALU instructions per element on index arithmetic: the
        // kernel was VALU-bound at 0.14 of the MFMA peak with its matrix pipe 4 % busy)
        constexpr int Q = (C::LXU + 3) / 4, ROWS = C::LZ * C::LY, NITEM = ROWS * Q, PER = (NITEM + C::NT - 1) / C::NT;
        const size_t img = (size_t)a.D * a.H * a.W;
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(inb, (unsigned)(img * 4));       // (the planner admits < 2 GiB per image)
        f32x4 v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int it = i * C::NT + tid, row = it / Q, q = it - row * Q;
            const int lz = row / C::LY, ly = row - lz * C::LY;
            const int gz = iz0 + lz, gy = iy0 + ly, gx = ix0 + 4 * q;
            const bool ok = it < NITEM && gz < a.D && gy < a.H && gx < a.W;
            v[i] = buf_load4(rin, ok ? (unsigned)((((size_t)gz * a.H + gy) * a.W + gx) * 4) : kOOB, 0);
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int it = i * C::NT + tid, row = it / Q, q = it - row * Q;
            if (it < NITEM) {
                float* lp = lds + row * C::LX + 4 * q;
                const int gx = ix0 + 4 * q;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * q + e < C::LX) lp[e] = (gx + e < a.W) ? v[i][e] : 0.f;       // beyond the image row: SAME padding
            }
        }
    } else {
#pragma unroll 1
    for (int it = 0; it < C::ITEMS; ++it) {
        const int u = it * C::NT + tid;
        const int lz = u / (C::LY * C::LX), rem = u - lz * (C::LY * C::LX);
        const int ly = rem / C::LX, lx = rem - ly * C::LX;
        const int gz = iz0 + lz, gy = iy0 + ly, gx = ix0 + lx;
        if (u < C::NV) {
            const bool ok = gz >= 0 && gz < a.D && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            lds[u] = ok ? inb[((size_t)gz * a.H + gy) * a.W + gx] : 0.f;
        }
    }
    }
    __syncthreads();

    const int w_yg = wave % (TY / R), w_z = wave / (TY / R);
    const float* lbase = lds + ((w_z * 2) * C::LY + (w_yg * R) * 2) * C::LX + v * 2 + kq;
    constexpr int ROW_OFF = 2 * C::LX;

    f32x4 acc[R][C::NCT];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int ct = 0; ct < C::NCT; ++ct) acc[i][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* wp = a.w + lane;  // packed [kz][ky][kxg][ct][lane]
#pragma unroll 1
    for (int kz = 0; kz < KS; ++kz) {
#pragma unroll 1
        for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
            for (int kg = 0; kg < C::KXG; ++kg) {
                float wf[C::NCT];
#pragma unroll
                for (int ct = 0; ct < C::NCT; ++ct) wf[ct] = wp[(size_t)((((kz * KS + ky) * C::KXG + kg) * C::NCT) + ct) * 64];
                const float* lp = lbase + (kz * C::LY + ky) * C::LX + kg * 4;
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const float b = lp[i * ROW_OFF];
#pragma unroll
                    for (int ct = 0; ct < C::NCT; ++ct) acc[i][ct] = mfma16(wf[ct], b, acc[i][ct]);
                }
            }
        }
    }

    const int gz = oz0 + w_z;
    float mx = 0.f;      // max |stored value| of this lane (NaNs skipped: fmaxf returns the other operand)
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int gy = oy0 + w_yg * R + i, gx = ox0 + v;
        if (gz < a.OD && gy < a.OH && gx < a.OW) {
            const size_t vox = (((size_t)n * a.OD + gz) * a.OH + gy) * a.OW + gx;
#pragma unroll
            for (int ct = 0; ct < C::NCT; ++ct) {
                const f32x4 o = store_out(a, acc[i][ct], vox, ct * 16 + kq * 4, COUT);
                mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            }
        }
    }
    // the fp16-split layer behind this one scales block n by its max |x| (conv_wino_f16s.hip): order-independent atomicMax of bit patterns
    if (a.amax_out != nullptr) pcc_amax_record(a.amax_out + (size_t)n * PCC_AMAX_SLOTS, mx, (int)blockIdx.x + wave);
}

// =====================================================================================================
// transposed conv with Cout = 1 (last synthesis layer): VALU.  One thread = one output voxel... each
// thread accumulates CIN x taps FMAs reading float4 channel quads from an LDS tile; weights are read
// through the scalar cache (uniform addresses).
//   S = 1: gather conv with (host-)flipped weights, pad (KS-1)/2.
//   S = 2: parity decomposition; a thread produces the 8 outputs of one base voxel.
// Accumulation order: taps (kz,ky,kx) outer, channels inner, fp32 FMA chain -> deterministic.
// =====================================================================================================
template <int CIN, int KS, int S, int TZ, int TY, int TXT>
struct Cout1Cfg {
    static constexpr int NT = TZ * TY * TXT;  // one thread per base voxel
    static constexpr int PLO = (S == 1) ? (KS - 1) / 2 : Tr2Geo<KS>::HL;
    static constexpr int PHI = (S == 1) ? (KS - 1) / 2 : Tr2Geo<KS>::HH;
    static constexpr int LZ = TZ + PLO + PHI, LY = TY + PLO + PHI, LX = TXT + PLO + PHI;
    static constexpr int VS = CIN + 4;  // +4: consecutive voxels land on different bank slots
    static constexpr int NV = LZ * LY * LX;
    static constexpr int LDS_BYTES = NV * VS * 4;
    static constexpr int Q = CIN / 4;
};

template <int CIN, int KS, int S, int TZ, int TY, int TXT>
__global__ void __launch_bounds__((Cout1Cfg<CIN, KS, S, TZ, TY, TXT>::NT)) conv_cout1_kernel(ConvArgs a) {
    using C = Cout1Cfg<CIN, KS, S, TZ, TY, TXT>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = t % a.ntx; t /= a.ntx;
    const int ty = t % a.nty; t /= a.nty;
    const int tz = t % a.ntz;
    const int n = t / a.ntz;
    const int bz0 = tz * TZ, by0 = ty * TY, bx0 = tx * TXT;

    const float* inb = a.in + (size_t)n * a.D * a.H * a.W * CIN;
#pragma unroll 1
    for (int item = tid; item < C::NV * C::Q; item += C::NT) {
        const int u = item / C::Q, q = item - u * C::Q;
        const int lz = u / (C::LY * C::LX), rem = u - lz * (C::LY * C::LX);
        const int ly = rem / C::LX, lx = rem - ly * C::LX;
        const int gz = bz0 - C::PLO + lz, gy = by0 - C::PLO + ly, gx = bx0 - C::PLO + lx;
        const bool ok = gz >= 0 && gz < a.D && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        f32x4 val = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (ok) val = *reinterpret_cast<const f32x4*>(inb + (((size_t)gz * a.H + gy) * a.W + gx) * CIN + q * 4);
        *reinterpret_cast<f32x4*>(lds + u * C::VS + q * 4) = val;
    }
    __syncthreads();

    const int lx = tid % TXT, ly = (tid / TXT) % TY, lz = tid / (TXT * TY);
    const float* lbase = lds + (((lz + C::PLO) * C::LY + ly + C::PLO) * C::LX + lx + C::PLO) * C::VS;
    const int gz = bz0 + lz, gy = by0 + ly, gx = bx0 + lx;
    const bool inb_ok = gz < a.D && gy < a.H && gx < a.W;
    const float bias = (a.flags & PCC_CONV_BIAS) ? a.bias[0] : 0.f;
    const f32x4* wq = reinterpret_cast<const f32x4*>(a.w);  // packed [tap][CIN/4] float4 (S=1: already flipped)

    auto finish = [&](float s, size_t vox) {
        s += bias;
        if (a.flags & PCC_CONV_RELU) s = fmaxf(s, 0.f);
        if (a.flags & PCC_CONV_ADD) s += a.res[vox];
        if (a.flags & PCC_CONV_CLIP01) s = fminf(fmaxf(s, 0.f), 1.f);
        a.out[vox * a.ocs + a.oco] = s;
    };

    if constexpr (S == 1) {
        float s = 0.f;
#pragma unroll 1
        for (int kz = 0; kz < KS; ++kz)
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const float* lp = lbase + (((kz - C::PLO) * C::LY + (ky - C::PLO)) * C::LX + (kx - C::PLO)) * C::VS;
                    const f32x4* wt = wq + (size_t)((kz * KS + ky) * KS + kx) * C::Q;
#pragma unroll
                    for (int q = 0; q < C::Q; ++q) {
                        const f32x4 x = *reinterpret_cast<const f32x4*>(lp + q * 4);
                        const f32x4 w = wt[q];
                        s = fmaf(x.x, w.x, s); s = fmaf(x.y, w.y, s); s = fmaf(x.z, w.z, s); s = fmaf(x.w, w.w, s);
                    }
                }
        if (inb_ok) finish(s, (((size_t)n * a.OD + gz) * a.OH + gy) * a.OW + gx);
    } else {
        using G = Tr2Geo<KS>;
#pragma unroll 1
        for (int pz = 0; pz < 2; ++pz)
#pragma unroll 1
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    float s = 0.f;
                    for (int kz = (pz + G::PL) & 1; kz < KS; kz += 2)
                        for (int ky = (py + G::PL) & 1; ky < KS; ky += 2)
#pragma unroll
                            for (int kx = (px + G::PL) & 1; kx < KS; kx += 2) {
                                const int dz = (pz + G::PL - kz) / 2, dy = (py + G::PL - ky) / 2, dx = (px + G::PL - kx) / 2;
                                const float* lp = lbase + ((dz * C::LY + dy) * C::LX + dx) * C::VS;
                                const f32x4* wt = wq + (size_t)((kz * KS + ky) * KS + kx) * C::Q;
#pragma unroll
                                for (int q = 0; q < C::Q; ++q) {
                                    const f32x4 x = *reinterpret_cast<const f32x4*>(lp + q * 4);
                                    const f32x4 w = wt[q];
                                    s = fmaf(x.x, w.x, s); s = fmaf(x.y, w.y, s); s = fmaf(x.z, w.z, s); s = fmaf(x.w, w.w, s);
                                }
                            }
                    if (inb_ok) finish(s, (((size_t)n * a.OD + 2 * gz + pz) * a.OH + 2 * gy + py) * a.OW + 2 * gx + px);
                }
    }
}

#if PCC_PART == 0
// =====================================================================================================
// Conv3DTranspose 16 -> 1, k3, stride 1 (last layer of the V2 synthesis transforms) on the matrix cores.
//   A GEMV-shaped layer has no N dimension for an implicit GEMM, so the contraction is split:
//     (1) P[tap][voxel] = sum_c w[tap][c] * in[voxel][c]      -- MFMA: M = 27 taps (2 tiles), N = 16 voxels,
//         K = 16 channels; every input voxel is read ONCE from global memory (coalesced 1 KiB per wave load);
//     (2) out[z,y,x] = sum_tap P[tap][voxel + offset(tap)]      -- 27 LDS reads + adds per output voxel.
//   A workgroup owns a T x T (y,x) column block of one image (and, with a z split, a slab of it) and marches along z: input
//   plane p feeds the three output planes p-1, p, p+1 through rolling accumulators.  T = 32 (round 3, when the grid still fills
//   the CUs): the haloed plane is 34^2 = 1156 voxels for 1024 outputs instead of 18^2 = 324 for 256 -- 13 % halo instead of 27 %
//   in both the MFMA work (which costs this layer as much time as its HBM floor) and the input reads.
//   Summation order per output: channels (MFMA chain) -> (ky,kx) -> kz, fixed => deterministic.
// =====================================================================================================
template <int T>
struct Cout1M {
    static constexpr int TYX = T, NT = T * T, NW = NT / 64;   // T = 16: 4 waves, 2-3 workgroups per CU; T = 32: 16 waves, one workgroup per CU
    static constexpr int LYX = TYX + 2;                 // haloed plane edge
    static constexpr int NTILE = (LYX * LYX + 15) / 16; // N-tiles of 16 voxels: 21 (324 voxels) / 73 (1156)
    static constexpr int NU = NTILE * 16;
    static constexpr int RS = NU + 20;                  // row pitch of P in floats: 4 * RS = 16 (mod 32) -> the four channel quads of a
                                                        // ds_write_b32 fall on two bank halves (2 cycles, the minimum for 64 lanes) instead of one;
                                                        // columns [NU + 4, NU + 20) of a row take the writes of the N-tiles past the plane (the last,
                                                        // partial round of tiles): every wave writes every round, no branch in the MFMA stream
    static_assert((4 * RS) % 32 == 16, "row pitch");
    static constexpr int LDS_BYTES = 32 * RS * 4;       // 32 tap rows: 27 + the zero rows of the second M tile (45.5 KB / 148.5 KB)
    static constexpr int PER_WAVE = (NTILE + NW - 1) / NW;    // N-tiles per wave (6,5,5,5 / 5 x9, 4 x7)
};

// IN16 (fp16 mode, PCC_CONV_IN16): the input is fp16 NDHWC; a lane's 4 channels are one 8-byte load and feed ONE
// v_mfma_f32_16x16x16_f16 per tap tile instead of four fp32 MFMAs.
// THR (pcc_thr_fuse, fixed-threshold extraction): the epilogue also decides occupancy -- (clipped) x_hat > thr[n], the float32
// compare of model_types.py:202,209 / :232-234 -- and stores it as one bit per voxel in (z,y,x) order: a wave's ballot is T-bit
// pieces of 64 / T output rows, each written once by its row's first lane.  The compaction pass then reads 32 KB per 64^3 block
// instead of x_hat twice.
template <bool IN16, int T, bool THR>
__global__ void __launch_bounds__((Cout1M<T>::NT)) conv_cout1_mfma_kernel(ConvArgs a) {
    using C = Cout1M<T>;
    extern __shared__ __attribute__((aligned(16))) float P[];   // [32][RS]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int v = lane & 15, cq = lane >> 4;
    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = t % a.ntx; t /= a.ntx;
    const int ty = t % a.nty; t /= a.nty;
    const int zs = t % a.ntz;                          // z slab [zb, ze) of output planes
    const int n = t / a.ntz;
    const int y0 = ty * C::TYX, x0 = tx * C::TYX;
    const int zlen = (a.D + a.ntz - 1) / a.ntz, zb = zs * zlen, ze = min(zb + zlen, a.D);
    const int p0 = max(zb - 1, 0), p1 = min(ze, a.D - 1);     // input planes of this slab

    // A operand: w[tap = 16*mt + (lane & 15)][channel 4*(lane>>4) + j], taps >= 27 are zero rows
    const f32x4 wA0 = *reinterpret_cast<const f32x4*>(a.w + (0 * 64 + lane) * 4);
    const f32x4 wA1 = *reinterpret_cast<const f32x4*>(a.w + (1 * 64 + lane) * 4);

    // this lane's voxels (one per N-tile it serves): byte offset inside a plane, sign bit set when outside H x W
    unsigned voff[C::PER_WAVE];
    int uidx[C::PER_WAVE];
#pragma unroll
    for (int k = 0; k < C::PER_WAVE; ++k) {
        const int nt = wave + C::NW * k;
        const int u = nt * 16 + v;
        const int ly = u / C::LYX, lx = u - ly * C::LYX;
        const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
        const bool ok = nt < C::NTILE && u < C::LYX * C::LYX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        voff[k] = ok ? (unsigned)((gy * a.W + gx) * 16 + cq * 4) * (IN16 ? 2u : 4u) : kOOB;
        uidx[k] = nt < C::NTILE ? u : C::NU + 4 + v;
    }
    const unsigned plane_bytes = (unsigned)a.H * a.W * (IN16 ? 32u : 64u);
    const unsigned char* inb = (const unsigned char*)a.in + (size_t)n * a.D * plane_bytes;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(inb, (unsigned)a.D * plane_bytes);
    // fp16 input arrives as 4 halfs in the low half of the float4 slot (bit pattern), converted weights beside it
    const h16x4 wA0h = __builtin_convertvector(wA0, h16x4), wA1h = __builtin_convertvector(wA1, h16x4);
    auto load_in = [&](unsigned voff_, unsigned soff) -> f32x4 {
        if constexpr (IN16) {
            const u32x2 r = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rin, (int)voff_, (int)soff, 0));
            return __builtin_bit_cast(f32x4, (u32x4){r[0], r[1], 0u, 0u});
        } else {
            return buf_load4(rin, voff_, soff);
        }
    };

    // gather side: thread -> output column (y, x)
    const int oy = tid / C::TYX, ox = tid % C::TYX;
    const bool col_ok = (y0 + oy) < a.OH && (x0 + ox) < a.OW;
    const float* pcol = P + oy * C::LYX + ox;
    const float bias = (a.flags & PCC_CONV_BIAS) ? a.bias[0] : 0.f;
    const size_t out_plane = (size_t)a.OH * a.OW;
    float* ob = a.out + ((size_t)n * a.OD * out_plane + (size_t)(y0 + oy) * a.OW + x0 + ox) * a.ocs + a.oco;
    const float* rb = (a.flags & PCC_CONV_ADD) ? a.res + (size_t)n * a.OD * out_plane + (size_t)(y0 + oy) * a.OW + x0 + ox : nullptr;

    float thr_n = 0.f;
    unsigned short* mrow = nullptr;        // this lane's row of the bit mask (meaningful on the first lane of each T-lane group)
    if constexpr (THR) {
        thr_n = a.thr[n];
        mrow = a.mask + ((size_t)n * a.OD * out_plane + (size_t)(y0 + oy) * a.OW + x0) / 16;
    }

    auto finish = [&](float s, int z) {
        s += bias;
        if (a.flags & PCC_CONV_RELU) s = fmaxf(s, 0.f);
        if (rb) s += rb[(size_t)z * out_plane];
        if (a.flags & PCC_CONV_CLIP01) s = fminf(fmaxf(s, 0.f), 1.f);
        if (col_ok) ob[(size_t)z * out_plane * a.ocs] = s;
        if constexpr (THR) {
            const float v = a.thr_clip ? fminf(fmaxf(s, 0.f), 1.f) : s;
            const unsigned long long hits = __ballot(col_ok && v > thr_n);
            if ((lane & (T - 1)) == 0 && (y0 + oy) < a.OH) {
                const unsigned piece = (unsigned)(hits >> (lane & 63 & ~(T - 1)));
                unsigned short* mp = mrow + (size_t)z * out_plane / 16;
                if constexpr (T == 32) *reinterpret_cast<unsigned*>(mp) = piece;
                else *mp = (unsigned short)piece;
            }
        }
    };

#ifndef PCC_C1_PROBE
#define PCC_C1_PROBE 0      // timing probes (tools/build_variant.sh): 1 no P writes, 2 no gather reads, 4 no MFMA, 8 no barriers, 16 no plane loads
#endif
    // ---- (1) P[tap][voxel] = W x in, one N-tile (16 voxels) at a time: d0 = taps 0..15, d1 = taps 16..31 of tile k
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};      // first k-slot starts from the inline constant 0: no zero-init pass (VALU costs MFMA time)
    f32x4 x[C::PER_WAVE], d0[C::PER_WAVE], d1[C::PER_WAVE];
    auto taps = [&](auto k0_tag, auto k1_tag) __attribute__((always_inline)) {      // tiles [K0, K1)
        constexpr int K0 = decltype(k0_tag)::value, K1 = decltype(k1_tag)::value;
        if constexpr ((PCC_C1_PROBE & 4) != 0) {
#pragma unroll
            for (int k = K0; k < K1; ++k) { d0[k] = x[k]; d1[k] = x[k] * wA1; }
        } else if constexpr (IN16) {
#pragma unroll
            for (int k = K0; k < K1; ++k) {
                const u32x4 cb = __builtin_bit_cast(u32x4, x[k]);
                const h16x4 bh = __builtin_bit_cast(h16x4, (u32x2){cb[0], cb[1]});
                d0[k] = __builtin_amdgcn_mfma_f32_16x16x16f16(wA0h, bh, zero4, 0, 0, 0);
                d1[k] = __builtin_amdgcn_mfma_f32_16x16x16f16(wA1h, bh, zero4, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = K0; k < K1; ++k) {
                    d0[k] = mfma16(wA0[j], x[k][j], j == 0 ? zero4 : d0[k]);
                    d1[k] = mfma16(wA1[j], x[k][j], j == 0 ? zero4 : d1[k]);
                }
        }
    };
    auto fetch = [&](auto k0_tag, auto k1_tag, int plane) __attribute__((always_inline)) {
        constexpr int K0 = decltype(k0_tag)::value, K1 = decltype(k1_tag)::value;
        const unsigned soff = (unsigned)(((PCC_C1_PROBE & 16) != 0) ? p0 : min(plane, p1)) * plane_bytes;     // (clamped: a plane past the slab is loaded, never used)
#pragma unroll
        for (int k = K0; k < K1; ++k) x[k] = load_in(voff[k], soff);
    };
    // all 32 tap rows are written (rows 27..31 are never read), tiles past the plane go to the pad columns: no branch around the ds_writes
    auto store = [&](auto k0_tag, auto k1_tag) __attribute__((always_inline)) {
        constexpr int K0 = decltype(k0_tag)::value, K1 = decltype(k1_tag)::value;
#pragma unroll
        for (int k = K0; k < K1; ++k) {
            if constexpr ((PCC_C1_PROBE & 1) != 0) { if (d0[k][0] == 1.2345f && d1[k][1] == 5.4321f) P[uidx[k]] = d0[k][2]; continue; }
            float* pw = P + uidx[k] + 4 * cq * C::RS;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pw[r * C::RS] = d0[k][r];
                pw[(16 + r) * C::RS] = d1[k][r];
            }
        }
    };
    // ---- (2) gather: the plane in P is tap kz = 0 of output p+1, kz = 1 of output p, kz = 2 of output p-1
    float accA = 0.f, accB = 0.f, accC = 0.f;   // outputs z = p+1, p, p-1
    auto gather = [&]() __attribute__((always_inline)) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float* q = pcol + ky * C::LYX + kx;
                if constexpr ((PCC_C1_PROBE & 2) != 0) { if (ky + kx > 0) continue; }
                s0 += q[(0 * 9 + ky * 3 + kx) * C::RS];
                s1 += q[(1 * 9 + ky * 3 + kx) * C::RS];
                s2 += q[(2 * 9 + ky * 3 + kx) * C::RS];
            }
        accA += s0; accB += s1; accC += s2;
    };
    auto sync = [&]() __attribute__((always_inline)) { if constexpr ((PCC_C1_PROBE & 8) == 0) __syncthreads(); };

    // Software pipeline (round 3; round 2 ran load -> MFMA -> LDS write -> barrier -> gather -> barrier strictly in turn, with the
    // matrix pipe idle 45 % of the time).  The N-tiles of a wave are split into a front group [0, KB) and a back group [KB, PER_WAVE).
    // At the top of iteration q:  P = tap planes of plane q;  d[front] = tap planes of plane q+1 (in flight);  d[back] = stale
    // (plane q, already in P);  x[back] = inputs of plane q+1;  x[front] = inputs of plane q+2 (loads in flight).
    //   phase A:  back-group MFMAs of plane q+1  ||  the 27 LDS reads of the gather of plane q;  finish(q-1);  refill x[back]
    //   barrier   (every wave has read P)
    //   phase B:  per front tile: write its plane-(q+1) rows, then its MFMAs of plane q+2 into the same registers;  refill x[front];
    //             the back group's rows are written between those MFMAs
    //   barrier   (P = plane q+1)
    // so the matrix pipe always has work while LDS is read or written, and every input tile is loaded one iteration before use.
    constexpr int KB = C::PER_WAVE - 2;
    using I0 = std::integral_constant<int, 0>;
    using IB = std::integral_constant<int, KB>;
    using IE = std::integral_constant<int, C::PER_WAVE>;
    fetch(I0{}, IE{}, p0);
    taps(I0{}, IE{});
    store(I0{}, IE{});
    fetch(I0{}, IB{}, p0 + 1);
    sync();
    taps(I0{}, IB{});
    fetch(I0{}, IB{}, p0 + 2);
    fetch(IB{}, IE{}, p0 + 1);
#pragma unroll 1
    for (int q = p0; q < p1; ++q) {
        // ---- phase A
        taps(IB{}, IE{});
        gather();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // 2 DS reads
        }
        fetch(IB{}, IE{}, q + 2);
        if (q - 1 >= zb) finish(accC, q - 1);          // (q - 1 < ze always: q <= ze)
        accC = accB; accB = accA; accA = 0.f;
        sync();
        // ---- phase B
        store(I0{}, IB{});
        taps(I0{}, IB{});
        store(IB{}, IE{});
        if constexpr (!IN16 && (PCC_C1_PROBE & 5) == 0) {
            // a front tile's 8 row writes go out before its first MFMA pair overwrites the registers; the remaining MFMAs carry the back rows
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x200, 8, 0);  // 8 DS writes
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // 2 MFMA (j = 0 of tile k)
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
        }
        fetch(I0{}, IB{}, q + 3);
        sync();
    }
    gather();                                          // plane p1
    if (p1 - 1 >= zb) finish(accC, p1 - 1);
    accC = accB; accB = accA; accA = 0.f;
    if (ze == a.D) finish(accC, a.D - 1);              // the last plane of the volume has no plane behind it
}

#endif  // PCC_PART == 0 (non-template kernel)

// =====================================================================================================
// host side: kernel selection, launch, weight packing
// =====================================================================================================
enum Kind { K_NONE = 0, K_FWD, K_TR2, K_CIN1, K_COUT1, K_COUT1M };

struct Plan {
    Kind kind = K_NONE;
    int tx = 0;           // 16 / 8 / 4: voxels of a row along x
    bool flip = false;    // transposed stride-1 -> forward with flipped taps
};

inline int base_w(const pcc_conv_desc* d) {  // x extent of the grid the rows live on
    if (d->transposed) return d->W;          // TR2: input grid; TR s1: same
    return pcc_same_out(d->W, d->stride);
}

static Plan make_plan(const pcc_conv_desc* d) {
    Plan p;
    if ((double)d->D * d->H * d->W * d->Cin * 4.0 >= 2147483648.0) return p;
    const bool out_vec_ok = d->out_cstride == 0 || (d->out_cstride % 4 == 0 && d->out_coffset % 4 == 0);
    const int k = d->k, s = d->stride;
    const bool even = (d->D % 2 == 0) && (d->H % 2 == 0) && (d->W % 2 == 0);
    if (!d->transposed && d->Cin == 1) {
        if (out_vec_ok && s == 2 && even && (k == 3 || k == 9) && (d->Cout == 16 || d->Cout == 32) && base_w(d) % 16 == 0) p.kind = K_CIN1;
        return p;
    }
    if (d->transposed && d->Cout == 1) {
        if (s == 1 && k == 3 && d->Cin == 16) { p.kind = K_COUT1M; p.flip = true; }
        else if (s == 1 && k == 3 && d->Cin == 32 && d->W % 8 == 0) { p.kind = K_COUT1; p.flip = true; }
        else if (s == 2 && k == 9 && d->Cin == 32 && d->W % 8 == 0) p.kind = K_COUT1;
        return p;
    }
    if (!out_vec_ok) return p;
    if (d->Cin % 16 || d->Cout % 16 || d->Cin > 64 || d->Cout > 64 || d->Cin == 48 || d->Cout == 48) return p;
    const int bw = base_w(d);
    const int tx = bw % 16 == 0 ? 16 : (bw == 8 ? 8 : (bw == 4 ? 4 : 0));
    if (!tx) return p;
    if (!d->transposed) {
        if (s == 1 && k == 3) p.kind = K_FWD;
        else if (s == 2 && even && (k == 3 || k == 5)) p.kind = K_FWD;
    } else {
        if (s == 1 && k == 3) { p.kind = K_FWD; p.flip = true; }
        else if (s == 2 && (k == 3 || k == 5)) p.kind = K_TR2;
    }
    if (p.kind == K_FWD && k == 5 && !(d->Cin == 32 && d->Cout == 32)) p.kind = K_NONE;
    if (p.kind == K_TR2 && k == 5 && !(d->Cin == 32 && d->Cout == 32)) p.kind = K_NONE;
    p.tx = tx;
    return p;
}

template <typename KernelT, typename... Extra>
int launch(KernelT kern, int nt, int lds_bytes, int tiles, const ConvArgs& a, hipStream_t st, Extra... extra) {
    { const int rc = pcc_enable_big_lds((const void*)kern, lds_bytes); if (rc != PCC_OK) return rc; }
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(nt), lds_bytes, st, a, extra...);
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

template <int CIN, int COUT, int KS, int S>
int launch_fwd(int tx, ConvArgs a, hipStream_t st, int num_cu, uint32_t numerics) {
    // tile shapes per row width: (TX, TZ, TY, TXT, R)
#define PCC_FWD(TX, TZ, TY, TXT, R) PCC_FWDC(TX, TZ, TY, TXT, R, (COUT / 16))
#define PCC_FWDC(TX, TZ, TY, TXT, R, CTW)                                                               \
    {                                                                                                   \
        using C = FwdCfg<CIN, COUT, KS, S, TX, TZ, TY, TXT, R, CTW>;                                    \
        a.ntz = cdiv(a.OD, TZ); a.nty = cdiv(a.OH, TY); a.ntx = cdiv(a.OW, TXT);                        \
        if (a.flags & PCC_CONV_F16)                                                                     \
            return launch(conv_fwd_kernel<CIN, COUT, KS, S, TX, TZ, TY, TXT, R, CTW, true>, C::NT, C::LDS_BYTES, \
                          a.N * a.ntz * a.nty * a.ntx, a, st);                                          \
        return launch(conv_fwd_kernel<CIN, COUT, KS, S, TX, TZ, TY, TXT, R, CTW>, C::NT, C::LDS_BYTES,  \
                      a.N * a.ntz * a.nty * a.ntx, a, st);                                              \
    }
    if constexpr (S == 1) {
        if (tx == 16) {
            if constexpr (COUT >= 64) {
                // pick the tile whose workgroup count fills the CU slots best (avoids a mostly empty last round)
                const long vox = (long)a.N * a.OD * a.OH * a.OW;
                const long wg_small = vox / 128, wg_big = vox / 256;          // (2,4,16) vs (2,8,16)
                const double t_small = (double)((wg_small + 3 * num_cu - 1) / (3 * num_cu)) * 1.0;
                const double t_big = (double)((wg_big + 2 * num_cu - 1) / (2 * num_cu)) * 2.0;
                const bool big = t_big <= t_small;
                if (big) PCC_FWD(16, 2, 8, 16, 4)
                PCC_FWD(16, 2, 4, 16, 2)
            }
            else if (COUT == 16 && CIN == 16 && KS == 3 && !(a.flags & PCC_CONV_F16)) {
                // persistent kernel, 2 workgroups per CU (tile 2x4x16, 80-byte LDS voxel stride)
#define PCC_P16(TZ, TY, R, VS, WPC)                                                                     \
    {                                                                                                   \
        using P = P16Cfg<TZ, TY, R, VS>;                                                                \
        a.ntz = cdiv(a.OD, TZ); a.nty = cdiv(a.OH, TY); a.ntx = cdiv(a.OW, 16);                         \
        const int ntiles = a.N * a.ntz * a.nty * a.ntx;                                                 \
        const int grid = ntiles < num_cu * WPC ? ntiles : num_cu * WPC;                                 \
        return launch(conv16_pers_kernel<TZ, TY, R, VS, WPC>, P::NT, P::LDS_BYTES, grid, a, st, ntiles); \
    }
                if (numerics & PCC_NUM_P16) PCC_P16(2, 8, 2, 20, 1)     // one 8-wave workgroup per CU (same speed, fewer halo re-reads)
                PCC_P16(2, 4, 2, 20, 2)
#undef PCC_P16
            }
            else PCC_FWD(16, 2, 8, 16, 4)
        }
        // small grids: few voxels per workgroup and the cout tiles split over waves, so that every CU gets work
        if (tx == 8) {
            if constexpr (COUT >= 64) PCC_FWDC(8, 2, 4, 8, 1, 1)
            else PCC_FWD(8, 2, 8, 8, 2)
        }
        if constexpr (COUT >= 32) PCC_FWDC(4, 1, 4, 4, 1, 1)
        else PCC_FWD(4, 4, 4, 4, 1)
    } else if constexpr (KS == 3) {  // stride 2: the staged input tile is 2x larger per dim
        if (tx == 16) PCC_FWD(16, 2, 2, 16, 1)
        if (tx == 8) PCC_FWD(8, 2, 4, 8, 1)
        if constexpr (COUT >= 32) PCC_FWDC(4, 1, 4, 4, 1, 1)
        else PCC_FWD(4, 4, 4, 4, 1)
    } else {
        if (tx == 16) PCC_FWD(16, 1, 2, 16, 1)
        if (tx == 8) PCC_FWD(8, 1, 4, 8, 1)
        PCC_FWD(4, 4, 4, 4, 1)
    }
#undef PCC_FWD
#undef PCC_FWDC
}

template <int CIN, int COUT, int KS>
int launch_tr2(int tx, ConvArgs a, hipStream_t st, int num_cu, uint32_t numerics) {
#define PCC_TR2(TX, TZ, TY, TXT, R) PCC_TR2C(TX, TZ, TY, TXT, R, (COUT / 16))
#define PCC_TR2C(TX, TZ, TY, TXT, R, CTW)                                                               \
    {                                                                                                   \
        using C = Tr2Cfg<CIN, COUT, KS, TX, TZ, TY, TXT, R, CTW>;                                       \
        a.ntz = cdiv(a.D, TZ); a.nty = cdiv(a.H, TY); a.ntx = cdiv(a.W, TXT);                           \
        if (a.flags & PCC_CONV_F16)                                                                     \
            return launch(conv_tr2_kernel<CIN, COUT, KS, TX, TZ, TY, TXT, R, CTW, true>, C::NT, C::LDS_BYTES, \
                          a.N * a.ntz * a.nty * a.ntx, a, st);                                          \
        return launch(conv_tr2_kernel<CIN, COUT, KS, TX, TZ, TY, TXT, R, CTW>, C::NT, C::LDS_BYTES,     \
                      a.N * a.ntz * a.nty * a.ntx, a, st);                                              \
    }
#define PCC_TR2G_EPI(TX, TZ, TY, TXT, R, CTW, F16, EPI)                                                  \
    return launch(conv_tr2g_kernel<CIN, COUT, TX, TZ, TY, TXT, R, CTW, F16, EPI>, C::NT, C::LDS_BYTES,       \
                  ntiles < slots ? ntiles : slots, a, st, ntiles);
#define PCC_TR2G(TX, TZ, TY, TXT, R, CTW)                                                               \
    if ((double)a.OD * a.OH * a.OW * a.ocs * 4.0 < 2147483648.0 && a.D < 512 && a.H < 512 && a.W < 512) { \
        using C = Tr2gCfg<CIN, COUT, TX, TZ, TY, TXT, R, CTW>;                                          \
        a.w += 27 * CIN * COUT;                                                                         \
        a.ntz = cdiv(a.D, TZ); a.nty = cdiv(a.H, TY); a.ntx = cdiv(a.W, TXT);                           \
        const int ntiles = a.N * a.ntz * a.nty * a.ntx;                                                 \
        const int slots = num_cu * (C::NT > 256 ? 1 : 2);                                               \
        const bool plain = !(a.flags & (PCC_CONV_ADD | PCC_CONV_CLIP01));                               \
        if (a.flags & PCC_CONV_F16) {                                                                   \
            if (plain && (a.flags & PCC_CONV_OUT16)) PCC_TR2G_EPI(TX, TZ, TY, TXT, R, CTW, true, TR2G_EPI_F16) \
            if (plain) PCC_TR2G_EPI(TX, TZ, TY, TXT, R, CTW, true, TR2G_EPI_F32)                        \
            PCC_TR2G_EPI(TX, TZ, TY, TXT, R, CTW, true, TR2G_EPI_ANY)                                   \
        }                                                                                               \
        if (plain) PCC_TR2G_EPI(TX, TZ, TY, TXT, R, CTW, false, TR2G_EPI_F32)                           \
        PCC_TR2G_EPI(TX, TZ, TY, TXT, R, CTW, false, TR2G_EPI_ANY)                                      \
    }
    const bool tr2_old = (numerics & PCC_NUM_TR2_OLD) != 0;
    if (tx == 16) {
        if constexpr (KS == 3) { if (!tr2_old) { if constexpr (CIN >= 64) { PCC_TR2G(16, 2, 4, 16, 2, 2) } else { PCC_TR2G(16, 2, 8, 16, 4, 1) } } }
        if constexpr (CIN >= 64) PCC_TR2(16, 2, 4, 16, 2)
        else PCC_TR2(16, 2, 8, 16, 4)
    }
    if (tx == 8) {
        if constexpr (KS == 3) { if (!tr2_old) { if constexpr (COUT >= 64) { PCC_TR2G(8, 2, 4, 8, 1, 1) } else { PCC_TR2G(8, 2, 8, 8, 2, 1) } } }
        if constexpr (COUT >= 64 && KS == 3) PCC_TR2C(8, 2, 4, 8, 1, 1)
        else PCC_TR2(8, 2, 8, 8, 2)
    }
#undef PCC_TR2G
#undef PCC_TR2G_EPI
    if constexpr (COUT >= 32 && KS == 3) PCC_TR2C(4, 1, 4, 4, 1, 1)
    else PCC_TR2(4, 4, 4, 4, 1)
#undef PCC_TR2
#undef PCC_TR2C
}


// ---- explicit instantiation lists: (CIN, COUT, KS, S) for launch_fwd, (CIN, COUT, KS) for launch_tr2 ------------
#define PCC_FWD_P1(X) X(16, 16, 3, 1) X(16, 32, 3, 2)
#define PCC_FWD_P2(X) X(32, 32, 3, 1) X(32, 32, 3, 2)
#define PCC_FWD_P3(X) X(64, 64, 3, 1) X(64, 64, 3, 2) X(32, 64, 3, 2)
#define PCC_FWD_P4(X) X(32, 32, 5, 2)
#define PCC_TR2_P4(X) X(32, 32, 5)
#define PCC_TR2_P5(X) X(64, 64, 3) X(64, 32, 3)
#define PCC_TR2_P6(X) X(32, 16, 3) X(32, 32, 3)
#define PCC_FWD_ALL(X) PCC_FWD_P1(X) PCC_FWD_P2(X) PCC_FWD_P3(X) PCC_FWD_P4(X)
#define PCC_TR2_ALL(X) PCC_TR2_P4(X) PCC_TR2_P5(X) PCC_TR2_P6(X)
#define PCC_INST_FWD(CI, CO, K, S) template int launch_fwd<CI, CO, K, S>(int, ConvArgs, hipStream_t, int, uint32_t);
#define PCC_INST_TR2(CI, CO, K) template int launch_tr2<CI, CO, K>(int, ConvArgs, hipStream_t, int, uint32_t);
#define PCC_EXT_FWD(CI, CO, K, S) extern template int launch_fwd<CI, CO, K, S>(int, ConvArgs, hipStream_t, int, uint32_t);
#define PCC_EXT_TR2(CI, CO, K) extern template int launch_tr2<CI, CO, K>(int, ConvArgs, hipStream_t, int, uint32_t);
#if PCC_PART == 0
PCC_FWD_ALL(PCC_EXT_FWD)
PCC_TR2_ALL(PCC_EXT_TR2)
#elif PCC_PART == 1
PCC_FWD_P1(PCC_INST_FWD)
#elif PCC_PART == 2
PCC_FWD_P2(PCC_INST_FWD)
#elif PCC_PART == 3
PCC_FWD_P3(PCC_INST_FWD)
#elif PCC_PART == 4
PCC_FWD_P4(PCC_INST_FWD)
PCC_TR2_P4(PCC_INST_TR2)
#elif PCC_PART == 5
PCC_TR2_P5(PCC_INST_TR2)
#elif PCC_PART == 6
PCC_TR2_P6(PCC_INST_TR2)
#endif

}  // namespace pccmfma

#if PCC_PART == 0
using namespace pccmfma;

PCC_API int pcc_conv_mfma_supported(const pcc_conv_desc* d) {
    if (!d) return 0;
    return make_plan(d).kind != K_NONE ? 1 : 0;
}

#define NGROUPS(c) ((c) / 16)
// stride-2 transposed k3 layers that carry a split-bf16 image behind their two fp32 images: 32 -> 16 (conv_tr2m_bf16.hip), 64 -> 32 and
// 64 -> 64 (conv_tr2_split_kernel, conv_split.hip)
static bool tr2_has_f16s_image(int Cin, int Cout) { return (Cin == 32 && Cout == 16) || (Cin == 64 && Cout == 32); }
static bool tr2_has_split_image(int Cin, int Cout) { return (Cin == 32 && Cout == 16) || (Cin == 64 && (Cout == 32 || Cout == 64)); }

// two-piece fp16 image of U behind everything else of a k3 stride-1 layer (conv_wino_f16s.hip): the 16- and 32-channel layers carry one
static size_t wino_f16s_floats(int C) { return (size_t)NGROUPS(C) * NGROUPS(C) * PCC_WINO_UH_FLOATS + PCC_WINO_UH_TAIL; }
static size_t wino_f16s_offset(int C) {
    return (size_t)27 * C * C + (size_t)NGROUPS(C) * NGROUPS(C) * (PCC_WINO_U_FLOATS + PCC_WINO_UB_FLOATS) + pcc_f16_packed_bytes(C) / 4 +
           (C >= 32 ? pcc_split_packed_floats(C) : 0);
}

PCC_API size_t pcc_conv_packed_floats(const pcc_conv_desc* d) {
    if (!d) return 0;
    const Plan p = make_plan(d);
    const size_t k3 = (size_t)d->k * d->k * d->k;
    switch (p.kind) {
        case K_FWD:
            // 16->16 / 32->32 k3 stride-1 layers also carry the Winograd-transformed weights (conv_wino.hip)
            if (pcc_wino_channels(d->Cin, d->Cout) && d->k == 3 && d->stride == 1)
                return k3 * d->Cin * d->Cout + (size_t)NGROUPS(d->Cin) * NGROUPS(d->Cout) * PCC_WINO_U_FLOATS +
                       pcc_f16_packed_bytes(d->Cin) / 4 +     // + the fp16 fragments of conv_f16.hip
                       (size_t)NGROUPS(d->Cin) * NGROUPS(d->Cout) * PCC_WINO_UB_FLOATS +    // + the split-bf16 image of U (conv_wino_bf16.hip)
                       (d->Cin >= 32 ? pcc_split_packed_floats(d->Cin) : 0) +              // + the split image of the direct kernel (conv_split.hip)
                       wino_f16s_floats(d->Cin);                                            // + the two-piece fp16 image of U (conv_wino_f16s.hip)
            return k3 * d->Cin * d->Cout;
        case K_TR2:     // k3: second copy in the order of conv_tr2g_kernel; 32 -> 16: + its split-bf16 image (conv_tr2m_bf16.hip)
            return k3 * d->Cin * d->Cout * (d->k == 3 ? 2 : 1) + (d->k == 3 && tr2_has_split_image(d->Cin, d->Cout) ? pcc_tr2m_bf16_packed_floats(d->Cin, d->Cout) : 0) +
                   (d->k == 3 && tr2_has_f16s_image(d->Cin, d->Cout) ? pcc_tr2m_f16s_packed_floats(d->Cin, d->Cout) : 0);      // 32 -> 16: + its two-piece fp16 image (conv_tr2m_f16s.hip)
        case K_CIN1: return (size_t)d->k * d->k * ((d->k + 3) / 4) * 4 * d->Cout;
        case K_COUT1: return k3 * d->Cin;
        case K_COUT1M: return 2 * 64 * 4;
        default: return 0;
    }
}

// Keras layouts: forward (k,k,k,Cin,Cout); transposed (k,k,k,Cout,Cin).
PCC_API int pcc_conv_pack_weights(const pcc_conv_desc* d, const float* w, float* pk) {
    PCC_REQUIRE(d && w && pk, "pcc_conv_pack_weights: NULL argument");
    const Plan p = make_plan(d);
    PCC_REQUIRE(p.kind != K_NONE, "pcc_conv_pack_weights: shape not covered by the MFMA path");
    const int k = d->k, Cin = d->Cin, Cout = d->Cout;
    const int NG = Cin / 16, NCT = Cout / 16;
    // logical forward-style weight W(tap, ci, co) for the gather formulation
    auto Wf = [&](int kz, int ky, int kx, int ci, int co) -> float {
        if (!d->transposed) return w[((((size_t)kz * k + ky) * k + kx) * Cin + ci) * Cout + co];
        if (p.flip) { kz = k - 1 - kz; ky = k - 1 - ky; kx = k - 1 - kx; }
        return w[((((size_t)kz * k + ky) * k + kx) * Cout + co) * Cin + ci];
    };
    if (p.kind == K_FWD) {
        // [g][tap][ct][lane][j] : cin = g*16 + 4*(lane>>4) + j, cout = ct*16 + (lane&15)
        for (int g = 0; g < NG; ++g)
            for (int kz = 0; kz < k; ++kz) for (int ky = 0; ky < k; ++ky) for (int kx = 0; kx < k; ++kx) {
                const int tap = (kz * k + ky) * k + kx;
                for (int ct = 0; ct < NCT; ++ct)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 4; ++j)
                            pk[((((size_t)g * k * k * k + tap) * NCT + ct) * 64 + lane) * 4 + j] =
                                Wf(kz, ky, kx, g * 16 + 4 * (lane >> 4) + j, ct * 16 + (lane & 15));
            }
        if (pcc_wino_channels(Cin, Cout) && k == 3 && d->stride == 1) {
            // [cin group][cout group] U[dz][py][px][lane][kk] = (G (x) G) g_dz  for cin = 16 cig + 4*(lane>>4) + kk,
            // cout = 16 cog + (lane & 15); double precision
            static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
            float* u = pk + (size_t)27 * Cin * Cout;
            for (int cig = 0; cig < NG; ++cig) for (int cog = 0; cog < NCT; ++cog)
                for (int kz = 0; kz < 3; ++kz) for (int py = 0; py < 4; ++py) for (int px = 0; px < 4; ++px)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int kk = 0; kk < 4; ++kk) {
                            double s = 0;
                            for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx)
                                s += G[py][ky] * G[px][kx] * (double)Wf(kz, ky, kx, 16 * cig + 4 * (lane >> 4) + kk, 16 * cog + (lane & 15));
                            u[(((size_t)(cig * NCT + cog) * 48 + (kz * 4 + py) * 4 + px) * 64 + lane) * 4 + kk] = (float)s;
                        }
            {      // fp16 fragment image of conv_f16.hip behind the Winograd block
                float* wlog = (float*)malloc((size_t)27 * Cin * Cout * sizeof(float));
                PCC_REQUIRE(wlog != nullptr, "pcc_conv_pack_weights: out of memory");
                for (int kz = 0; kz < 3; ++kz) for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx)
                    for (int ci = 0; ci < Cin; ++ci) for (int co = 0; co < Cout; ++co)
                        wlog[((((size_t)kz * 3 + ky) * 3 + kx) * Cin + ci) * Cout + co] = Wf(kz, ky, kx, ci, co);
                pcc_f16_pack(Cin, wlog, (unsigned short*)(u + (size_t)NG * NCT * PCC_WINO_U_FLOATS));
                // split-bf16 image of U behind the fp16 block, the direct kernel's split taps behind that
                float* ub = u + (size_t)NG * NCT * PCC_WINO_U_FLOATS + pcc_f16_packed_bytes(Cin) / 4;
                pcc_wino_bf16_pack(NG, u, ub);
                if (Cin >= 32) pcc_split_pack(Cin, wlog, ub + (size_t)NG * NCT * PCC_WINO_UB_FLOATS);
                if (wino_f16s_floats(Cin)) pcc_wino_f16s_pack(NG, u, pk + wino_f16s_offset(Cin));
                free(wlog);
            }
        }
    } else if (p.kind == K_TR2) {
        // consumption order of conv_tr2_kernel: [parity class (pz,py,px)][taps of the class (kz,ky,kx)][g][ct][lane][j]
        const int PL = (k - 2) / 2;
        size_t seq = 0;
        for (int pz = 0; pz < 2; ++pz) for (int py = 0; py < 2; ++py) for (int px = 0; px < 2; ++px)
            for (int kz = (pz + PL) & 1; kz < k; kz += 2) for (int ky = (py + PL) & 1; ky < k; ky += 2)
                for (int kx = (px + PL) & 1; kx < k; kx += 2)
                    for (int g = 0; g < NG; ++g, ++seq)
                        for (int ct = 0; ct < NCT; ++ct)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int j = 0; j < 4; ++j)
                                    pk[(((seq * NCT) + ct) * 64 + lane) * 4 + j] =
                                        Wf(kz, ky, kx, g * 16 + 4 * (lane >> 4) + j, ct * 16 + (lane & 15));
        if (k == 3) {
            // conv_tr2g_kernel: [g][parity class][taps of the class][ct][lane][j]
            float* pg = pk + (size_t)27 * Cin * Cout;
            for (int g = 0; g < NG; ++g) {
                size_t sq = 0;
                for (int pz = 0; pz < 2; ++pz) for (int py = 0; py < 2; ++py) for (int px = 0; px < 2; ++px)
                    for (int kz = pz; kz < 3; kz += 2) for (int ky = py; ky < 3; ky += 2) for (int kx = px; kx < 3; kx += 2, ++sq)
                        for (int ct = 0; ct < NCT; ++ct)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int j = 0; j < 4; ++j)
                                    pg[((((size_t)g * 27 + sq) * NCT + ct) * 64 + lane) * 4 + j] =
                                        Wf(kz, ky, kx, g * 16 + 4 * (lane >> 4) + j, ct * 16 + (lane & 15));
            }
            if (tr2_has_split_image(Cin, Cout)) pcc_tr2m_bf16_pack(Cin, Cout, pg, pg + (size_t)27 * Cin * Cout);
            if (tr2_has_f16s_image(Cin, Cout)) pcc_tr2m_f16s_pack(Cin, Cout, pg, pg + (size_t)27 * Cin * Cout + pcc_tr2m_bf16_packed_floats(Cin, Cout));
        }
    } else if (p.kind == K_CIN1) {
        // [kz][ky][kxg][ct][lane] : kx = kxg*4 + (lane>>4) (zero beyond k), cout = ct*16 + (lane&15)
        const int KXG = (k + 3) / 4;
        for (int kz = 0; kz < k; ++kz) for (int ky = 0; ky < k; ++ky) for (int kg = 0; kg < KXG; ++kg)
            for (int ct = 0; ct < NCT; ++ct)
                for (int lane = 0; lane < 64; ++lane) {
                    const int kx = kg * 4 + (lane >> 4);
                    pk[((((size_t)(kz * k + ky) * KXG + kg) * NCT) + ct) * 64 + lane] =
                        kx < k ? Wf(kz, ky, kx, 0, ct * 16 + (lane & 15)) : 0.f;
                }
    } else if (p.kind == K_COUT1M) {
        // [mt][lane][j]: tap = 16*mt + (lane & 15) (zero rows beyond 27), channel = 4*(lane>>4) + j
        for (int mt = 0; mt < 2; ++mt)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const int tap = 16 * mt + (lane & 15);
                    pk[(mt * 64 + lane) * 4 + j] = tap < 27 ? Wf(tap / 9, (tap / 3) % 3, tap % 3, 4 * (lane >> 4) + j, 0) : 0.f;
                }
    } else {  // K_COUT1: [tap][ci]
        for (int kz = 0; kz < k; ++kz) for (int ky = 0; ky < k; ++ky) for (int kx = 0; kx < k; ++kx)
            for (int ci = 0; ci < Cin; ++ci)
                pk[((size_t)((kz * k + ky) * k + kx)) * Cin + ci] = Wf(kz, ky, kx, ci, 0);
    }
    return PCC_OK;
}

int pcc_conv3d_mfma(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w_packed, const float* bias,
                    const float* residual, float* out, hipStream_t st) {
    return pcc_conv3d_mfma_thr(ctx, d, in, w_packed, bias, residual, out, nullptr, nullptr, nullptr, st);
}

// The dispatch rules of the k3 stride-1 layers with Cin = Cout in {16, 32, 64} (shape + context state only), in the order
// pcc_conv3d_mfma_thr applies them: 0 = a direct kernel, 1 = the direct split-bf16 kernel (conv_split.hip), 2 = Winograd exact fp32,
// 3 = Winograd split-bf16 (16 channels), 4 = Winograd two-piece fp16 (16 / 32 channels, conv_wino_f16s.hip)
static bool f16s64_preferred(const pcc_ctx* ctx, const pcc_conv_desc* d) {
    return !(d->flags & (PCC_CONV_F16 | PCC_CONV_OUT16 | PCC_CONV_CLIP01)) && !ctx->num(PCC_NUM_NO_SPLIT | PCC_NUM_NO_F16S | PCC_NUM_NO_WINOGRAD | PCC_NUM_NO_WINOGRAD64) &&
           pcc_wino_eligible(d) && pcc_wino_f16s_covers(d);
}
static int k3s1_route(const pcc_ctx* ctx, const pcc_conv_desc* d) {
    const int ci = d->Cin;
    if (d->impl == PCC_IMPL_SPLIT) return 1;
    // 64 channels on grids of 16-multiples: the two-piece fp16 Winograd kernel as two launches of two cin groups (conv_wino_f16s.hip)
    // ahead of the direct bf16 kernel; PCC_NO_F16S=1 / PCC_NO_WINOGRAD64=1: the direct kernel (A/B)
    if (d->impl == PCC_IMPL_AUTO && ci == 64 && f16s64_preferred(ctx, d)) return 4;
    if (d->impl == PCC_IMPL_AUTO && ci >= 32 && !(d->flags & (PCC_CONV_F16 | PCC_CONV_OUT16)) && !ctx->num(PCC_NUM_NO_SPLIT | PCC_NUM_NO_SPLIT_DIRECT) &&
        pcc_split_covers(d) && pcc_split_preferred(ctx, d))
        return 1;
    const bool no_wino = ctx->num(PCC_NUM_NO_WINOGRAD), no_wino32 = ctx->num(PCC_NUM_NO_WINOGRAD32), wino64 = !ctx->num(PCC_NUM_NO_WINOGRAD64);
    const bool want = d->impl == PCC_IMPL_WINOGRAD || (d->impl == PCC_IMPL_AUTO && !(d->flags & PCC_CONV_F16) && !no_wino && !(ci == 32 && (no_wino32 || d->D < 16)) && !(ci == 64 && !wino64));
    if (!(want && pcc_wino_eligible(d))) return 0;
    if (!ctx->num(PCC_NUM_NO_SPLIT | PCC_NUM_NO_F16S) && pcc_wino_f16s_covers(d) && (ci == 16 || !(d->flags & PCC_CONV_CLIP01)) && !(ci == 64 && (d->flags & PCC_CONV_F16))) return 4;
    if (!ctx->num(PCC_NUM_NO_SPLIT) && pcc_wino_bf16_covers(d)) return 3;
    return 2;
}

// The dispatch rules of the k3 stride-2 transposed layers (shape + context state only), in the order pcc_conv3d_mfma_thr applies them:
// 0 = tiled exact-fp32 kernels, 1 = z march in the fp16 MODE (conv_tr2m_f16.hip), 2 = z march with two fp16 pieces (conv_tr2m_f16s.hip:
// 32 -> 16, 64 -> 32 on grids of 16-multiples), 3 = parity-class tiles with three bf16 pieces (conv_tr2_split_kernel), 4 = z march with
// three bf16 pieces (32 -> 16), 5 = z march exact fp32
static int tr2_route(const pcc_ctx* ctx, const pcc_conv_desc* d) {
    if (d->k != 3) return 0;
    const bool autoi = d->impl == PCC_IMPL_AUTO;
    if (autoi && !ctx->num(PCC_NUM_NO_TR2M) && pcc_tr2m_f16_covers(d)) return 1;
    const bool plain = !(d->flags & (PCC_CONV_F16 | PCC_CONV_OUT16));
    if (autoi && plain && !ctx->num(PCC_NUM_NO_TR2M | PCC_NUM_NO_SPLIT | PCC_NUM_NO_SPLIT_TR2 | PCC_NUM_NO_F16S) && pcc_tr2m_f16s_covers(d)) return 2;
    if (autoi && plain && pcc_tr2_split_covers(d) && !ctx->num(PCC_NUM_NO_SPLIT | PCC_NUM_NO_SPLIT_TR2)) return 3;
    if (!ctx->num(PCC_NUM_NO_TR2M) && (ctx->num(PCC_NUM_TR2M) ? pcc_tr2m_eligible(d) : pcc_tr2m_preferred(ctx, d))) {
        if (plain && !ctx->num(PCC_NUM_NO_SPLIT | PCC_NUM_NO_SPLIT_TR2 | PCC_NUM_NO_F16S) && pcc_tr2m_f16s_covers(d)) return 2;      // (PCC_IMPL_MFMA callers)
        return pcc_tr2m_bf16_covers(d) && !ctx->num(PCC_NUM_NO_SPLIT | PCC_NUM_NO_SPLIT_TR2) ? 4 : 5;
    }
    return 0;
}

// Does the kernel picked for this layer take the fp16-split path (it then wants the per-block max of its input)?
bool pcc_conv_wants_amax(const pcc_ctx* ctx, const pcc_conv_desc* d) {
    if (d->flags & (PCC_CONV_IN16 | PCC_CONV_OUT16)) return false;
    const Plan p = make_plan(d);
    if (p.kind == K_TR2) return tr2_route(ctx, d) == 2;
    if (p.kind != K_FWD || !(pcc_wino_channels(d->Cin, d->Cout) && d->k == 3 && (p.flip ? 1 : d->stride) == 1)) return false;
    return k3s1_route(ctx, d) == 4;
}

// The kernel family pcc_conv3d takes for a layer on this context (same tests, same order as the dispatch below): what bench.py prints
// beside every layer's time, and what a maintainer asks when two builds disagree in the last bits.
PCC_API int pcc_conv_kernel_family(pcc_ctx* ctx, const pcc_conv_desc* d, char* buf, int32_t cap) {
    PCC_REQUIRE(ctx && d && buf && cap > 0, "pcc_conv_kernel_family: NULL argument");
    const char* name = "generic (reference-order fp32 FMA chain)";
    const Plan p = make_plan(d);
    const int ci = d->Cin, co = d->Cout, k = d->k;
    if (d->impl == PCC_IMPL_GENERIC || p.kind == K_NONE) {
    } else if ((d->flags & (PCC_CONV_IN16 | PCC_CONV_RES16)) && p.kind != K_COUT1M) name = "conv_f16 (fp16 storage, f16 MFMA)";
    else if (p.kind == K_FWD) {
        name = (d->flags & PCC_CONV_F16) ? "conv_fwd (f16 MFMA)" : "conv_fwd (exact fp32 MFMA)";
        if (pcc_wino_channels(ci, co) && k == 3 && (p.flip ? 1 : d->stride) == 1) {
            switch (k3s1_route(ctx, d)) {
                case 1: name = (ctx->num(PCC_NUM_SPLIT_MFMA32) || (!ctx->num(PCC_NUM_SPLIT_MFMA16) && ci == 32)) && d->W != 8 ? "conv_k3s1_split32 (direct, bf16 x 3, 32x32x16 MFMA)" : "conv_k3s1_split (direct, bf16 x 3, 16x16x32 MFMA)"; break;
                case 2: name = "conv16_wino (Winograd, exact fp32 MFMA)"; break;
                case 3: name = "conv16_wino_bf16 (Winograd, bf16 x 3)"; break;
                case 4: name = "conv16_wino_f16s (Winograd, fp16 x 2 under a per-block pre-scale)"; break;
                default: break;
            }
        }
    } else if (p.kind == K_TR2) {
        static const char* const tn[6] = {"conv_tr2 (exact fp32 MFMA)", "conv_tr2m_f16 (z march, f16 MFMA)", "conv_tr2m_f16s (z march, fp16 x 2 under a per-block pre-scale)",
                                          "conv_tr2_split (parity classes, bf16 x 3)", "conv_tr2m_bf16 (z march, bf16 x 3)", "conv_tr2m (z march, exact fp32 MFMA)"};
        name = tn[tr2_route(ctx, d)];
    } else if (p.kind == K_CIN1) name = "conv_cin1 (exact fp32 MFMA)";
    else if (p.kind == K_COUT1M) name = "conv_cout1_mfma (exact fp32 MFMA)";
    else if (p.kind == K_COUT1) name = "conv_cout1 (fp32 VALU)";
    snprintf(buf, (size_t)cap, "%s", name);
    return PCC_OK;
}

int pcc_conv3d_mfma_thr(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w_packed, const float* bias,
                        const float* residual, float* out, const pcc_thr_fuse* fuse, bool* fused, pcc_conv_ext* ext, hipStream_t st) {
    if (fused) *fused = false;
    if (ext) ext->out_recorded = false;
    const Plan p = make_plan(d);
    PCC_REQUIRE(p.kind != K_NONE, "pcc_conv3d_mfma: shape not covered");
    ConvArgs a;
    a.in = in; a.w = w_packed; a.bias = bias; a.res = residual; a.out = out;
    a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W;
    pcc_conv_out_dims(d, &a.OD, &a.OH, &a.OW);
    a.flags = d->flags;
    a.ocs = d->out_cstride ? d->out_cstride : d->Cout;
    a.oco = d->out_coffset;
    a.ntz = a.nty = a.ntx = 0;
    const int ci = d->Cin, co = d->Cout, k = d->k, s = d->stride;

#define PCC_CASE_FWD(CI, CO, K, S) if (ci == CI && co == CO && k == K && fs == S) return launch_fwd<CI, CO, K, S>(p.tx, a, st, ctx->num_cu, ctx->numerics);
#define PCC_CASE_TR2(CI, CO, K) if (ci == CI && co == CO && k == K) return launch_tr2<CI, CO, K>(p.tx, a, st, ctx->num_cu, ctx->numerics);
    PCC_REQUIRE(!(d->flags & PCC_CONV_OUT16) || p.kind == K_FWD || p.kind == K_TR2 || p.kind == K_CIN1,
                "pcc_conv3d: PCC_CONV_OUT16 needs a layer with Cout a multiple of 16");
    if ((d->flags & (PCC_CONV_IN16 | PCC_CONV_RES16)) && p.kind != K_COUT1M) {
        // fp16-storage layer (conv_f16.hip): fp16 input (and residual), fp16 or fp32 output
        PCC_REQUIRE(p.kind == K_FWD && (d->flags & PCC_CONV_IN16) && pcc_f16_eligible(d),
                    "pcc_conv3d: PCC_CONV_IN16 covers k3 stride-1 layers with Cin = Cout in {16, 32, 64} (H, W multiples of 16) and the 16 -> 1 transposed layer");
        const float* f16w = w_packed + (size_t)27 * ci * co + (size_t)(ci / 16) * (co / 16) * PCC_WINO_U_FLOATS;
        return pcc_conv_f16(ctx, d, in, f16w, bias, residual, out, !(d->flags & PCC_CONV_OUT16), st);
    }
    if (p.kind == K_FWD) {
        PCC_REQUIRE(!(d->flags & PCC_CONV_OUT16) || d->impl != PCC_IMPL_WINOGRAD, "pcc_conv3d: PCC_CONV_OUT16 is not implemented by the Winograd kernel (fp32)");
        const int fs = p.flip ? 1 : s;
        if (pcc_wino_channels(ci, co) && k == 3 && fs == 1) {
            // 32- / 64-channel layers: direct convolution on the bf16 MFMA pipe with split operands (conv_split.hip) where it beats the
            // fp32-MFMA Winograd kernel.  PCC_NO_SPLIT=1 (all split paths) / PCC_NO_SPLIT_DIRECT=1 (this one) for A/B runs
            const float* w_split = w_packed + (size_t)27 * ci * co + (size_t)(ci / 16) * (co / 16) * (PCC_WINO_U_FLOATS + PCC_WINO_UB_FLOATS) + pcc_f16_packed_bytes(ci) / 4;
            const int route = k3s1_route(ctx, d);
            if (route == 1) {
                PCC_REQUIRE(ci >= 32 && pcc_split_covers(d) && !(d->flags & (PCC_CONV_F16 | PCC_CONV_OUT16)), "pcc_conv3d: PCC_IMPL_SPLIT covers fp32 k3 stride-1 layers with Cin = Cout in {32, 64}, W % 16 == 0");
                return pcc_conv_split(ctx, d, in, w_split, bias, residual, out, st);
            }
            const float* u32 = w_packed + (size_t)27 * ci * co;
            // two fp16 pieces under a per-block power-of-two pre-scale (conv_wino_f16s.hip, round 6: 16- and 32-channel layers);
            // PCC_NO_F16S=1: three bf16 pieces (16 channels) / exact fp32; PCC_NO_SPLIT=1: exact-fp32 MFMA everywhere (A/B)
            if (route == 4) return pcc_conv_wino_f16s(ctx, d, in, w_packed + wino_f16s_offset(ci), bias, residual, out, ext, st);
            if (route == 3) return pcc_conv_wino_bf16(ctx, d, in, u32 + (size_t)(ci / 16) * (co / 16) * PCC_WINO_U_FLOATS + pcc_f16_packed_bytes(ci) / 4, bias, residual, out, st);
            if (route == 2) return pcc_conv_wino(ctx, d, in, u32, bias, residual, out, st);
            PCC_REQUIRE(d->impl != PCC_IMPL_WINOGRAD, "pcc_conv3d: PCC_IMPL_WINOGRAD needs W and H multiples of 16");
        } else {
            PCC_REQUIRE(d->impl != PCC_IMPL_WINOGRAD, "pcc_conv3d: PCC_IMPL_WINOGRAD covers Cin = Cout in {16,32,64} k3 stride-1 layers only");
        }
        PCC_CASE_FWD(16, 16, 3, 1) PCC_CASE_FWD(32, 32, 3, 1) PCC_CASE_FWD(64, 64, 3, 1)
        PCC_CASE_FWD(16, 32, 3, 2) PCC_CASE_FWD(32, 64, 3, 2) PCC_CASE_FWD(64, 64, 3, 2)
        PCC_CASE_FWD(32, 32, 3, 2) PCC_CASE_FWD(32, 32, 5, 2)
    } else if (p.kind == K_TR2) {
        // tr2_route: fp16-mode march | two-piece fp16 march (32 -> 16, 64 -> 32; PCC_NO_F16S=1 off) | parity-class tiles, bf16 x 3 (64 -> 32 / 64 -> 64;
        // PCC_NO_SPLIT=1 / PCC_NO_SPLIT_TR2=1 off) | bf16 x 3 march (32 -> 16) | exact-fp32 march (PCC_NO_TR2M=1: the tiled kernels below)
        switch (tr2_route(ctx, d)) {
            case 1: return pcc_conv_tr2m_f16(ctx, d, in, w_packed + (size_t)27 * ci * co, bias, out, st);
            case 2: return pcc_conv_tr2m_f16s(ctx, d, in, w_packed + (size_t)2 * 27 * ci * co + pcc_tr2m_bf16_packed_floats(ci, co), bias, out, ext, st);
            case 3: return pcc_conv_tr2_split(ctx, d, in, w_packed + (size_t)2 * 27 * ci * co, bias, out, ext, st);
            case 4: return pcc_conv_tr2m_bf16(ctx, d, in, w_packed + (size_t)2 * 27 * ci * co, bias, out, ext, st);
            case 5: return pcc_conv_tr2m(ctx, d, in, w_packed + (size_t)27 * ci * co, bias, out, st);
            default: break;
        }
        PCC_CASE_TR2(64, 64, 3) PCC_CASE_TR2(64, 32, 3) PCC_CASE_TR2(32, 16, 3) PCC_CASE_TR2(32, 32, 3)
        PCC_CASE_TR2(32, 32, 5)
    } else if (p.kind == K_CIN1) {
#define PCC_CIN1(CO, K, TZ, TY, R)                                                                      \
    if (co == CO && k == K) {                                                                           \
        using C = Cin1Cfg<CO, K, TZ, TY, R>;                                                            \
        if (ext && ext->out_amax && !(d->flags & PCC_CONV_OUT16)) { a.amax_out = ext->out_amax; ext->out_recorded = true; } \
        a.ntz = cdiv(a.OD, TZ); a.nty = cdiv(a.OH, TY); a.ntx = cdiv(a.OW, 16);                         \
        return launch(conv_cin1_kernel<CO, K, TZ, TY, R>, C::NT, C::LDS_BYTES, a.N * a.ntz * a.nty * a.ntx, a, st); \
    }
        PCC_CIN1(16, 3, 2, 8, 4) PCC_CIN1(32, 3, 2, 8, 4) PCC_CIN1(16, 9, 2, 8, 4) PCC_CIN1(32, 9, 2, 8, 4)
#undef PCC_CIN1
    } else if (p.kind == K_COUT1M) {
        // 32 x 32 columns (one 16-wave workgroup per CU) when H, W allow it and the grid, z-split into slabs of >= 16 planes, still
        // gives every CU a workgroup; else 16 x 16 columns, whole z range.  PCC_COUT1_T16=1 forces the latter (A/B runs).
        const bool t16 = ctx->num(PCC_NUM_COUT1_T16);
        typedef void (*kern_t)(ConvArgs);
        const bool in16 = (d->flags & PCC_CONV_IN16) != 0;
        // the occupancy bits ride along when the caller asked for them and whole T-voxel rows map to whole mask pieces
        const bool thr = fuse != nullptr && a.ocs == 1 && a.oco == 0 && a.W % 16 == 0 && ((size_t)a.H * a.W) % 32 == 0;
        if (thr) { a.thr = fuse->thr; a.mask = (unsigned short*)fuse->mask; a.thr_clip = fuse->clip; }
        if (fused) *fused = thr;
        if (!t16 && a.H % 32 == 0 && a.W % 32 == 0) {
            const int base = a.N * (a.H / 32) * (a.W / 32);
            int zsp = 1;
            while (base * zsp < ctx->num_cu && a.D / (zsp * 2) >= 16) zsp *= 2;
            if (base * zsp >= ctx->num_cu) {
                using C = Cout1M<32>;
                a.ntz = zsp; a.nty = a.H / 32; a.ntx = a.W / 32;
                static const kern_t k32[4] = {conv_cout1_mfma_kernel<false, 32, false>, conv_cout1_mfma_kernel<true, 32, false>,
                                              conv_cout1_mfma_kernel<false, 32, true>, conv_cout1_mfma_kernel<true, 32, true>};
                const kern_t kern = k32[(in16 ? 1 : 0) + (thr ? 2 : 0)];
                { const int rc = pcc_enable_big_lds((const void*)kern, C::LDS_BYTES); if (rc != PCC_OK) return rc; }
                return launch(kern, C::NT, C::LDS_BYTES, base * zsp, a, st);
            }
        }
        using C = Cout1M<16>;
        a.ntz = 1; a.nty = cdiv(a.H, C::TYX); a.ntx = cdiv(a.W, C::TYX);
        static const kern_t k16[4] = {conv_cout1_mfma_kernel<false, 16, false>, conv_cout1_mfma_kernel<true, 16, false>,
                                      conv_cout1_mfma_kernel<false, 16, true>, conv_cout1_mfma_kernel<true, 16, true>};
        return launch(k16[(in16 ? 1 : 0) + (thr ? 2 : 0)], C::NT, C::LDS_BYTES, a.N * a.nty * a.ntx, a, st);
    } else if (p.kind == K_COUT1) {
#define PCC_COUT1(CI, K, S, TZ, TY, TXT)                                                                \
    if (ci == CI && k == K && s == S) {                                                                 \
        using C = Cout1Cfg<CI, K, S, TZ, TY, TXT>;                                                      \
        a.ntz = cdiv(a.D, TZ); a.nty = cdiv(a.H, TY); a.ntx = cdiv(a.W, TXT);                           \
        return launch(conv_cout1_kernel<CI, K, S, TZ, TY, TXT>, C::NT, C::LDS_BYTES, a.N * a.ntz * a.nty * a.ntx, a, st); \
    }
        PCC_COUT1(16, 3, 1, 4, 8, 8) PCC_COUT1(32, 3, 1, 4, 8, 8) PCC_COUT1(32, 9, 2, 2, 8, 8)
#undef PCC_COUT1
    }
    pcc_set_error("pcc_conv3d_mfma: no instantiation for Cin=%d Cout=%d k=%d s=%d transposed=%d", ci, co, k, s, d->transposed);
    return PCC_ERR_ARG;
}

#endif  // PCC_PART == 0
