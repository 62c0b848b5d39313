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
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

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
};

__device__ __forceinline__ void store_out(const ConvArgs& a, f32x4 v, size_t vox, int c0, int COUT) {
    // v = 4 consecutive output channels c0..c0+3 of voxel `vox`
    if (a.flags & PCC_CONV_BIAS) v += *reinterpret_cast<const f32x4*>(a.bias + c0);
    if (a.flags & PCC_CONV_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (a.flags & PCC_CONV_ADD) v += *reinterpret_cast<const f32x4*>(a.res + vox * COUT + c0);
    if (a.flags & PCC_CONV_CLIP01) {
        v.x = fminf(fmaxf(v.x, 0.f), 1.f); v.y = fminf(fmaxf(v.y, 0.f), 1.f);
        v.z = fminf(fmaxf(v.z, 0.f), 1.f); v.w = fminf(fmaxf(v.w, 0.f), 1.f);
    }
    *reinterpret_cast<f32x4*>(a.out + vox * a.ocs + a.oco + c0) = v;
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
template <int CIN, int COUT, int KS, int S, int TX, int TZ, int TY, int TXT, int R>
struct FwdCfg {
    static constexpr int NG = CIN / 16, NCT = COUT / 16;
    static constexpr int RY = 16 / TX;
    static constexpr int NYB = TY / RY, NXB = TXT / TX;
    static constexpr int NW = TZ * (NYB / R) * NXB;
    static constexpr int NT = NW * 64;
    static constexpr int PL = (S == 1) ? (KS - 1) / 2 : (KS - 2) / 2;  // SAME pad_low (even input dims for S=2)
    static constexpr int LZ = (TZ - 1) * S + KS, LY = (TY - 1) * S + KS, LX = (TXT - 1) * S + KS;
    static constexpr int VS = 24;  // floats per voxel in LDS: 16 staged channels + 8 pad
    static constexpr int NV = LZ * LY * LX;
    static constexpr int LDS_BYTES = NV * VS * 4;
    static constexpr int ITEMS = (NV * 4 + NT - 1) / NT;
    static_assert(TY % RY == 0 && NYB % R == 0 && TXT % TX == 0, "bad tile");
};

template <int CIN, int COUT, int KS, int S, int TX, int TZ, int TY, int TXT, int R>
__global__ void __launch_bounds__((FwdCfg<CIN, COUT, KS, S, TX, TZ, TY, TXT, R>::NT))
conv_fwd_kernel(ConvArgs a) {
    using C = FwdCfg<CIN, COUT, KS, S, TX, TZ, TY, TXT, R>;
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
    const int w_xb = wv % C::NXB; wv /= C::NXB;
    const int w_yg = wv % (C::NYB / R);
    const int w_z = wv / (C::NYB / R);
    const int ry = v / TX, rx = v % TX;
    const int ly0 = (w_yg * R * C::RY + ry), lx0 = (w_xb * TX + rx);  // local output coords of row 0
    const float* lbase = lds + ((w_z * S * C::LY + ly0 * S) * C::LX + lx0 * S) * C::VS + cq * 4;
    constexpr int ROW_OFF = C::RY * S * C::LX * C::VS;  // floats between consecutive rows of a wave

    f32x4 acc[R][C::NCT];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int ct = 0; ct < C::NCT; ++ct) acc[i][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const f32x4* wp = reinterpret_cast<const f32x4*>(a.w) + lane;
    const float* inb = a.in + (size_t)n * a.D * a.H * a.W * CIN;

#pragma unroll 1
    for (int g = 0; g < C::NG; ++g) {
        // ---- stage 16 channels of the haloed input tile (zero fill = SAME padding)
        f32x4 stg[C::ITEMS];
#pragma unroll
        for (int it = 0; it < C::ITEMS; ++it) {
            const int item = it * C::NT + tid;
            const int u = item >> 2, q = item & 3;
            const int lz = u / (C::LY * C::LX), rem = u - lz * (C::LY * C::LX);
            const int ly = rem / C::LX, lx = rem - ly * C::LX;
            const int gz = iz0 + lz, gy = iy0 + ly, gx = ix0 + lx;
            const bool ok = (item < C::NV * 4) && gz >= 0 && gz < a.D && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            f32x4 val = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (ok) val = *reinterpret_cast<const f32x4*>(inb + (((size_t)gz * a.H + gy) * a.W + gx) * CIN + g * 16 + q * 4);
            stg[it] = val;
        }
        if (g > 0) __syncthreads();  // all waves finished reading the previous group
#pragma unroll
        for (int it = 0; it < C::ITEMS; ++it) {
            const int item = it * C::NT + tid;
            if (item < C::NV * 4) *reinterpret_cast<f32x4*>(lds + (item >> 2) * C::VS + (item & 3) * 4) = stg[it];
        }
        __syncthreads();

        const f32x4* wg = wp + (size_t)g * (KS * KS * KS) * C::NCT * 64;
#pragma unroll 1
        for (int kz = 0; kz < KS; ++kz) {
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const int tap = (kz * KS + ky) * KS + kx;
                    f32x4 wf[C::NCT];
#pragma unroll
                    for (int ct = 0; ct < C::NCT; ++ct) wf[ct] = wg[(size_t)(tap * C::NCT + ct) * 64];
                    const int toff = ((kz * C::LY + ky) * C::LX + kx) * C::VS;
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        const f32x4 b = *reinterpret_cast<const f32x4*>(lbase + toff + i * ROW_OFF);
#pragma unroll
                        for (int ct = 0; ct < C::NCT; ++ct) {
                            acc[i][ct] = mfma16(wf[ct].x, b.x, acc[i][ct]);
                            acc[i][ct] = mfma16(wf[ct].y, b.y, acc[i][ct]);
                            acc[i][ct] = mfma16(wf[ct].z, b.z, acc[i][ct]);
                            acc[i][ct] = mfma16(wf[ct].w, b.w, acc[i][ct]);
                        }
                    }
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
            for (int ct = 0; ct < C::NCT; ++ct) store_out(a, acc[i][ct], vox, ct * 16 + cq * 4, COUT);
        }
    }
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

template <int CIN, int COUT, int KS, int TX, int TZ, int TY, int TXT, int R>
struct Tr2Cfg {
    using G = Tr2Geo<KS>;
    static constexpr int NG = CIN / 16, NCT = COUT / 16;
    static constexpr int RY = 16 / TX;
    static constexpr int NYB = TY / RY, NXB = TXT / TX;
    static constexpr int NW = TZ * (NYB / R) * NXB;
    static constexpr int NT = NW * 64;
    static constexpr int LZ = TZ + G::HL + G::HH, LY = TY + G::HL + G::HH, LX = TXT + G::HL + G::HH;
    static constexpr int VS = CIN + 8;
    static constexpr int NV = LZ * LY * LX;
    static constexpr int LDS_BYTES = NV * VS * 4;
    static constexpr int Q = CIN / 4;  // float4 per voxel
    static constexpr int ITEMS = (NV * Q + NT - 1) / NT;
};

template <int CIN, int COUT, int KS, int TX, int TZ, int TY, int TXT, int R>
__global__ void __launch_bounds__((Tr2Cfg<CIN, COUT, KS, TX, TZ, TY, TXT, R>::NT))
conv_tr2_kernel(ConvArgs a) {
    using C = Tr2Cfg<CIN, COUT, KS, TX, TZ, TY, TXT, R>;
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
    const int w_xb = wv % C::NXB; wv /= C::NXB;
    const int w_yg = wv % (C::NYB / R);
    const int w_z = wv / (C::NYB / R);
    const int ry = v / TX, rx = v % TX;
    const int ly0 = w_yg * R * C::RY + ry, lx0 = w_xb * TX + rx;
    const float* lbase = lds + (((w_z + G::HL) * C::LY + ly0 + G::HL) * C::LX + lx0 + G::HL) * C::VS + cq * 4;
    constexpr int ROW_OFF = C::RY * C::LX * C::VS;

    // ---- stage the whole haloed tile, all channels
    const float* inb = a.in + (size_t)n * a.D * a.H * a.W * CIN;
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
            const bool ok = (it0 + k < C::ITEMS) && (item < C::NV * C::Q) && gz >= 0 && gz < a.D && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            f32x4 val = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (ok) val = *reinterpret_cast<const f32x4*>(inb + (((size_t)gz * a.H + gy) * a.W + gx) * CIN + q * 4);
            stg[k] = val;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int item = (it0 + k) * C::NT + tid;
            const int u = item / C::Q, q = item - u * C::Q;
            if ((it0 + k < C::ITEMS) && item < C::NV * C::Q) *reinterpret_cast<f32x4*>(lds + u * C::VS + q * 4) = stg[k];
        }
    }
    __syncthreads();

    const f32x4* wp = reinterpret_cast<const f32x4*>(a.w) + lane;
    const int gzb = bz0 + w_z;

#pragma unroll
    for (int pz = 0; pz < 2; ++pz)
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                f32x4 acc[R][C::NCT];
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int ct = 0; ct < C::NCT; ++ct) acc[i][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                for (int g = 0; g < C::NG; ++g) {
#pragma unroll
                    for (int kz = (pz + G::PL) & 1; kz < KS; kz += 2)
#pragma unroll
                        for (int ky = (py + G::PL) & 1; ky < KS; ky += 2)
#pragma unroll
                            for (int kx = (px + G::PL) & 1; kx < KS; kx += 2) {
                                const int dz = (pz + G::PL - kz) / 2, dy = (py + G::PL - ky) / 2, dx = (px + G::PL - kx) / 2;
                                const int tap = (kz * KS + ky) * KS + kx;
                                f32x4 wf[C::NCT];
#pragma unroll
                                for (int ct = 0; ct < C::NCT; ++ct) wf[ct] = wp[(size_t)((tap * C::NG + g) * C::NCT + ct) * 64];
                                const int toff = ((dz * C::LY + dy) * C::LX + dx) * C::VS;
#pragma unroll
                                for (int i = 0; i < R; ++i) {
                                    const f32x4 b = *reinterpret_cast<const f32x4*>(lbase + toff + i * ROW_OFF + g * 16);
#pragma unroll
                                    for (int ct = 0; ct < C::NCT; ++ct) {
                                        acc[i][ct] = mfma16(wf[ct].x, b.x, acc[i][ct]);
                                        acc[i][ct] = mfma16(wf[ct].y, b.y, acc[i][ct]);
                                        acc[i][ct] = mfma16(wf[ct].z, b.z, acc[i][ct]);
                                        acc[i][ct] = mfma16(wf[ct].w, b.w, acc[i][ct]);
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
                        for (int ct = 0; ct < C::NCT; ++ct) store_out(a, acc[i][ct], vox, ct * 16 + cq * 4, COUT);
                    }
                }
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
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int gy = oy0 + w_yg * R + i, gx = ox0 + v;
        if (gz < a.OD && gy < a.OH && gx < a.OW) {
            const size_t vox = (((size_t)n * a.OD + gz) * a.OH + gy) * a.OW + gx;
#pragma unroll
            for (int ct = 0; ct < C::NCT; ++ct) store_out(a, acc[i][ct], vox, ct * 16 + kq * 4, COUT);
        }
    }
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

// =====================================================================================================
// host side: kernel selection, launch, weight packing
// =====================================================================================================
enum Kind { K_NONE = 0, K_FWD, K_TR2, K_CIN1, K_COUT1 };

struct Plan {
    Kind kind = K_NONE;
    int tx = 0;           // 16 / 8 / 4: voxels of a row along x
    bool flip = false;    // transposed stride-1 -> forward with flipped taps
};

inline int base_w(const pcc_conv_desc* d) {  // x extent of the grid the rows live on
    if (d->transposed) return d->W;          // TR2: input grid; TR s1: same
    return pcc_same_out(d->W, d->stride);
}

Plan make_plan(const pcc_conv_desc* d) {
    Plan p;
    const bool out_vec_ok = d->out_cstride == 0 || (d->out_cstride % 4 == 0 && d->out_coffset % 4 == 0);
    const int k = d->k, s = d->stride;
    const bool even = (d->D % 2 == 0) && (d->H % 2 == 0) && (d->W % 2 == 0);
    if (!d->transposed && d->Cin == 1) {
        if (out_vec_ok && s == 2 && even && (k == 3 || k == 9) && (d->Cout == 16 || d->Cout == 32) && base_w(d) % 16 == 0) p.kind = K_CIN1;
        return p;
    }
    if (d->transposed && d->Cout == 1) {
        if (s == 1 && k == 3 && (d->Cin == 16 || d->Cin == 32) && d->W % 8 == 0) { p.kind = K_COUT1; p.flip = true; }
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

template <typename KernelT>
int launch(KernelT kern, int nt, int lds_bytes, int tiles, const ConvArgs& a, hipStream_t st) {
    static thread_local const void* configured[64];
    static thread_local int nconf = 0;
    bool done = false;
    for (int i = 0; i < nconf; ++i) done |= (configured[i] == (const void*)kern);
    if (!done) {
        if (lds_bytes > 64 * 1024)
            PCC_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        if (nconf < 64) configured[nconf++] = (const void*)kern;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(nt), lds_bytes, st, a);
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

template <int CIN, int COUT, int KS, int S>
int launch_fwd(int tx, ConvArgs a, hipStream_t st) {
    // tile shapes per row width: (TX, TZ, TY, TXT, R)
#define PCC_FWD(TX, TZ, TY, TXT, R)                                                                     \
    {                                                                                                   \
        using C = FwdCfg<CIN, COUT, KS, S, TX, TZ, TY, TXT, R>;                                         \
        a.ntz = cdiv(a.OD, TZ); a.nty = cdiv(a.OH, TY); a.ntx = cdiv(a.OW, TXT);                        \
        return launch(conv_fwd_kernel<CIN, COUT, KS, S, TX, TZ, TY, TXT, R>, C::NT, C::LDS_BYTES,       \
                      a.N * a.ntz * a.nty * a.ntx, a, st);                                              \
    }
    if constexpr (S == 1) {
        if (tx == 16) {
            if constexpr (COUT >= 64) PCC_FWD(16, 2, 4, 16, 2)
            else PCC_FWD(16, 2, 8, 16, 4)
        }
        if (tx == 8) PCC_FWD(8, 2, 8, 8, 2)
        PCC_FWD(4, 4, 4, 4, 1)
    } else if constexpr (KS == 3) {  // stride 2: the staged input tile is 2x larger per dim
        if (tx == 16) PCC_FWD(16, 2, 2, 16, 1)
        if (tx == 8) PCC_FWD(8, 2, 4, 8, 1)
        PCC_FWD(4, 4, 4, 4, 1)
    } else {
        if (tx == 16) PCC_FWD(16, 1, 2, 16, 1)
        if (tx == 8) PCC_FWD(8, 1, 4, 8, 1)
        PCC_FWD(4, 4, 4, 4, 1)
    }
#undef PCC_FWD
}

template <int CIN, int COUT, int KS>
int launch_tr2(int tx, ConvArgs a, hipStream_t st) {
#define PCC_TR2(TX, TZ, TY, TXT, R)                                                                     \
    {                                                                                                   \
        using C = Tr2Cfg<CIN, COUT, KS, TX, TZ, TY, TXT, R>;                                            \
        a.ntz = cdiv(a.D, TZ); a.nty = cdiv(a.H, TY); a.ntx = cdiv(a.W, TXT);                           \
        return launch(conv_tr2_kernel<CIN, COUT, KS, TX, TZ, TY, TXT, R>, C::NT, C::LDS_BYTES,          \
                      a.N * a.ntz * a.nty * a.ntx, a, st);                                              \
    }
    if (tx == 16) {
        if constexpr (CIN >= 64) PCC_TR2(16, 2, 4, 16, 2)
        else PCC_TR2(16, 2, 8, 16, 4)
    }
    if (tx == 8) PCC_TR2(8, 2, 8, 8, 2)
    PCC_TR2(4, 4, 4, 4, 1)
#undef PCC_TR2
}

}  // namespace

PCC_API int pcc_conv_mfma_supported(const pcc_conv_desc* d) {
    if (!d) return 0;
    return make_plan(d).kind != K_NONE ? 1 : 0;
}

PCC_API size_t pcc_conv_packed_floats(const pcc_conv_desc* d) {
    if (!d) return 0;
    const Plan p = make_plan(d);
    const size_t k3 = (size_t)d->k * d->k * d->k;
    switch (p.kind) {
        case K_FWD:
        case K_TR2: return k3 * d->Cin * d->Cout;
        case K_CIN1: return (size_t)d->k * d->k * ((d->k + 3) / 4) * 4 * d->Cout;
        case K_COUT1: return k3 * d->Cin;
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
    } else if (p.kind == K_TR2) {
        // [tap][g][ct][lane][j]
        for (int kz = 0; kz < k; ++kz) for (int ky = 0; ky < k; ++ky) for (int kx = 0; kx < k; ++kx) {
            const int tap = (kz * k + ky) * k + kx;
            for (int g = 0; g < NG; ++g)
                for (int ct = 0; ct < NCT; ++ct)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 4; ++j)
                            pk[((((size_t)tap * NG + g) * NCT + ct) * 64 + lane) * 4 + j] =
                                Wf(kz, ky, kx, g * 16 + 4 * (lane >> 4) + j, ct * 16 + (lane & 15));
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
    } else {  // K_COUT1: [tap][ci]
        for (int kz = 0; kz < k; ++kz) for (int ky = 0; ky < k; ++ky) for (int kx = 0; kx < k; ++kx)
            for (int ci = 0; ci < Cin; ++ci)
                pk[((size_t)((kz * k + ky) * k + kx)) * Cin + ci] = Wf(kz, ky, kx, ci, 0);
    }
    return PCC_OK;
}

int pcc_conv3d_mfma(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w_packed, const float* bias,
                    const float* residual, float* out, hipStream_t st) {
    (void)ctx;
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

#define PCC_CASE_FWD(CI, CO, K, S) if (ci == CI && co == CO && k == K && fs == S) return launch_fwd<CI, CO, K, S>(p.tx, a, st);
#define PCC_CASE_TR2(CI, CO, K) if (ci == CI && co == CO && k == K) return launch_tr2<CI, CO, K>(p.tx, a, st);
    if (p.kind == K_FWD) {
        const int fs = p.flip ? 1 : s;
        PCC_CASE_FWD(16, 16, 3, 1) PCC_CASE_FWD(32, 32, 3, 1) PCC_CASE_FWD(64, 64, 3, 1)
        PCC_CASE_FWD(16, 32, 3, 2) PCC_CASE_FWD(32, 64, 3, 2) PCC_CASE_FWD(64, 64, 3, 2)
        PCC_CASE_FWD(32, 32, 3, 2) PCC_CASE_FWD(32, 32, 5, 2)
    } else if (p.kind == K_TR2) {
        PCC_CASE_TR2(64, 64, 3) PCC_CASE_TR2(64, 32, 3) PCC_CASE_TR2(32, 16, 3) PCC_CASE_TR2(32, 32, 3)
        PCC_CASE_TR2(32, 32, 5)
    } else if (p.kind == K_CIN1) {
#define PCC_CIN1(CO, K, TZ, TY, R)                                                                      \
    if (co == CO && k == K) {                                                                           \
        using C = Cin1Cfg<CO, K, TZ, TY, R>;                                                            \
        a.ntz = cdiv(a.OD, TZ); a.nty = cdiv(a.OH, TY); a.ntx = cdiv(a.OW, 16);                         \
        return launch(conv_cin1_kernel<CO, K, TZ, TY, R>, C::NT, C::LDS_BYTES, a.N * a.ntz * a.nty * a.ntx, a, st); \
    }
        PCC_CIN1(16, 3, 2, 8, 4) PCC_CIN1(32, 3, 2, 8, 4) PCC_CIN1(16, 9, 2, 8, 4) PCC_CIN1(32, 9, 2, 8, 4)
#undef PCC_CIN1
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
