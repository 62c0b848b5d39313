// Direct 3x3x3 stride-1 convolution on the bf16 MFMA pipe with fp32-EQUIVALENT operands, Cin = Cout in {32, 64} (round 4).
//
// The multi-group k3 stride-1 layers of /root/reference/src/model_transforms.py:62-81 (32 -> 32 @32^3 / 16^3, 64 -> 64 @16^3) ran
// on conv16_wino_cin_kernel (Winograd, exact-fp32 MFMA, 0.58 - 0.70 of the 157 TFLOP/s pipe executed).  The split-bf16 Winograd
// kernel (conv_wino_bf16.hip) does not extend to them: its split weights are 96 KB per (cin group, cout group) and a launch
// needs all cin groups of a cout group in LDS.  A DIRECT convolution has no such coupling, and on the bf16 pipe its 2.25x larger
// multiply count is cheaper than Winograd on the fp32 pipe: three v_mfma_f32_16x16x32_bf16 (48 cycles) per tap, 16 voxels and
// (16 cin, 16 cout) pair against 4 x 32 / 2.25 = 57 cycles of fp32 Winograd MFMAs -- with none of Winograd's VALU work: no input /
// output transforms, no AccVGPR shuffles, and the operand split is paid ONCE per input element when the tile is staged
// (every staged element then feeds 27 taps x Cout multiply-adds) instead of once per (tile, point) element.
//
// Operand split as in conv_wino_bf16.hip: x = h + m + l exactly (three bf16 pieces), six product terms
//     acc += [Wh | Wm] . [dh | dm]   (hh + mm);   acc += [Wh | Wm] . [dl | dh]   (hl + mh);   acc += [Wl | Wh] . [dh | dm]   (lh + hm)
// fp32 accumulation in a fixed order (cin group -> tap -> the three MFMAs): bit-deterministic, independent of tile position and
// launch geometry.
//
// Structure = conv_fwd_kernel's k3 path (conv_mfma.hip): workgroup = 4 waves = tile of 2 x 8 x 16 output voxels; a wave owns the
// R = 8 rows of 16 voxels of one z plane and HALF of the cout tiles.  (First version: 4 rows x all cout tiles per wave -- every wave
// then pulls all weights of a tap from L2, 42 B/clk/CU, and the MFMA pipe sat at 0.46 / 0.64 busy waiting for them; with 8 rows per
// weight fragment it is 21 B/clk/CU, and the LDS serves the 2 x 8 row fragments per tap at 85 - 170 B/clk of its 256.)  Per cin
// group the haloed input tile (4 x 10 x 18 voxels x 16 channels) is staged global -> registers -> split -> LDS, 160 bytes per voxel:
// [B1: 4 cin quads x 16 B][B2: 4 x 16 B][32 B pad] -- the stride 40 dwords = 8 * 5 keeps the 16 lanes of every ds_read_b128 group
// on distinct bank quads, like the 24-dword stride of the fp32 kernel.  Weights stream from L2 through a register ring two taps
// ahead; the next group's staging loads and the residual rows ride in the tap sections.
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

#include "common.h"

namespace pccsplit {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ u32x4 buf_load4u(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
constexpr unsigned kOOB = 0x80000000u;
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}
__device__ __forceinline__ f32x4 mfma_bf16(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// only VALU / SALU may cross: memory operations and MFMAs keep their written order (as PCC_PIN_MEM_MFMA in conv_mfma.hip)
#define PCC_SPLIT_PIN() __builtin_amdgcn_sched_barrier(0x406)
// timing probes (tools/build_variant.sh): 1 no weight loads in the tap loop, 2 no LDS operand reads in the tap loop, 4 no staging loads,
// 8 (16x16x32 kernel only) MFMAs of tap 0 only.  64 -> 64 @8^3 x 32 alone, round 6: 23.8 us; 1: 19.0; 2: 22.5; 4: 22.5; 7: 17.5; 8: 8.5; 15: 7.9
#ifndef PCC_SPLIT_PROBE
#define PCC_SPLIT_PROBE 0
#endif

// Two staging items (2 x 4 input channels of a voxel each): fp32 -> B1 = [dh | dm], B2 = [dl | dh] each.  ONE asm block, because
// v_dot2c_f32_bf16 is a DOT instruction: a different VALU op that reads its result needs 3 wait states behind it
// (GCNHazardRecognizer: DotWriteDifferentVALURead) and the hazard recogniser cannot see into inline asm.  Inside the block every
// reader sits >= 3 instructions behind its writer; K0 / K1 = the bf16 pairs {-1, 0} / {0, -1}: x -= lo(h) / hi(h), exactly.
__device__ __forceinline__ void split_items2(u32x4& p_b1, u32x4& p_b2, u32x4& q_b1, u32x4& q_b2, const f32x4& pv, const f32x4& qv) {
    float a = pv[0], b = pv[1], c = pv[2], d = pv[3], e = qv[0], f = qv[1], g = qv[2], h = qv[3];
    unsigned ph01, ph23, pm01, pm23, pl01, pl23, pg01, pg23, qh01, qh23, qm01, qm23, ql01, ql23, qg01, qg23;
    asm volatile(
        "v_cvt_pk_bf16_f32 %8, %0, %1\n\tv_cvt_pk_bf16_f32 %9, %2, %3\n\tv_cvt_pk_bf16_f32 %16, %4, %5\n\tv_cvt_pk_bf16_f32 %17, %6, %7\n\t"
        "v_cvt_pk_bf16_f32 %14, %0, %1\n\tv_cvt_pk_bf16_f32 %15, %2, %3\n\tv_cvt_pk_bf16_f32 %22, %4, %5\n\tv_cvt_pk_bf16_f32 %23, %6, %7\n\t"
        "v_dot2c_f32_bf16 %0, %24, %8\n\tv_dot2c_f32_bf16 %1, %25, %8\n\tv_dot2c_f32_bf16 %2, %24, %9\n\tv_dot2c_f32_bf16 %3, %25, %9\n\t"
        "v_dot2c_f32_bf16 %4, %24, %16\n\tv_dot2c_f32_bf16 %5, %25, %16\n\tv_dot2c_f32_bf16 %6, %24, %17\n\tv_dot2c_f32_bf16 %7, %25, %17\n\t"
        "v_cvt_pk_bf16_f32 %10, %0, %1\n\tv_cvt_pk_bf16_f32 %11, %2, %3\n\tv_cvt_pk_bf16_f32 %18, %4, %5\n\ts_nop 0\n\tv_cvt_pk_bf16_f32 %19, %6, %7\n\t"
        "v_dot2c_f32_bf16 %0, %24, %10\n\tv_dot2c_f32_bf16 %1, %25, %10\n\tv_dot2c_f32_bf16 %2, %24, %11\n\tv_dot2c_f32_bf16 %3, %25, %11\n\t"
        "v_dot2c_f32_bf16 %4, %24, %18\n\tv_dot2c_f32_bf16 %5, %25, %18\n\tv_dot2c_f32_bf16 %6, %24, %19\n\tv_dot2c_f32_bf16 %7, %25, %19\n\t"
        "v_cvt_pk_bf16_f32 %12, %0, %1\n\tv_cvt_pk_bf16_f32 %13, %2, %3\n\tv_cvt_pk_bf16_f32 %20, %4, %5\n\ts_nop 0\n\tv_cvt_pk_bf16_f32 %21, %6, %7\n\ts_nop 2"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h),
          "=&v"(ph01), "=&v"(ph23), "=&v"(pm01), "=&v"(pm23), "=&v"(pl01), "=&v"(pl23), "=&v"(pg01), "=&v"(pg23),
          "=&v"(qh01), "=&v"(qh23), "=&v"(qm01), "=&v"(qm23), "=&v"(ql01), "=&v"(ql23), "=&v"(qg01), "=&v"(qg23)
        : "s"(0x0000bf80u), "s"(0xbf800000u));
    p_b1 = (u32x4){ph01, ph23, pm01, pm23}; p_b2 = (u32x4){pl01, pl23, pg01, pg23};
    q_b1 = (u32x4){qh01, qh23, qm01, qm23}; q_b2 = (u32x4){ql01, ql23, qg01, qg23};
}


struct SplitArgs {
    const float* in;
    const float* w;      // split image: [cin group][tap][cout tile][operand][lane][8 bf16]
    const float* bias;
    const float* res;
    float* out;
    int N, D, H, W;
    int ntz, nty, ntx;
    int flags, ocs, oco;
    unsigned* amax_out = nullptr;      // per-block max |out| for the fp16-split layer behind this one (common.h, pcc_conv_ext); conv_tr2_split_kernel
};

// (round 5) staging item -> (voxel, cin quad): the four voxels of 16 consecutive items are taken in the order 0, 2, 1, 3.  A ds_write_b128 is served
// in groups of 8 lanes on 32 banks; two voxels 160 B apart overlap in 8 of them (2-way conflicts on every staging write: the 0.10 - 0.15 of
// the PMC rows), 320 B apart they use the other 16 banks.  A bijection inside every aligned group of 16 items.
__host__ __device__ constexpr int stage_perm(int item) { return (item & ~0xC) | ((item & 4) << 1) | ((item & 8) >> 1); }

// TXW = 16: an MFMA row = 16 voxels of one y line.  TXW = 8 (the 8^3 grids of the 64-channel layers): a row = 2 y lines of 8 voxels.
template <int CH, int TZ, int TY, int R, int CTW, int TXW = 16>
struct SplitCfg {
    static constexpr int NG = CH / 16, NCT = CH / 16;
    static constexpr int NCG = NCT / CTW;               // waves also split the cout tiles
    static constexpr int LPR = 16 / TXW;                // y lines per row
    static constexpr int NYG = TY / (R * LPR);          // y groups (of R rows) of the tile
    static constexpr int NW = TZ * NYG * NCG, NT = NW * 64;
    static constexpr int LZ = TZ + 2, LY = TY + 2, LX = TXW + 2;
    // Line pitch of the LDS tile in voxels.  TXW = 8 (round 5): an MFMA row is TWO lines, and with the natural pitch of 10 voxels the second
    // line starts 100 bank quads = 4 (mod 16) behind the first: the ds_read_b128 lane groups {0-3, 12-15, 20-27} ... (MI355X_MICROARCH.md, LDS)
    // then hit {0,4,10,14} with lanes 0-3 and {12,6,0,10} with lanes 12-15 -- measured SQ_LDS_BANK_CONFLICT = 0.50 of the LDS cycles.  A pitch of
    // 16 voxels (160 quads = 0 mod 16) puts the second line on the complementary quads {2,6,8,12} / {1,5,11,15}: conflict-free for every tap.
    static constexpr int LXP = TXW == 8 ? 16 : LX;
    static constexpr int VS = 40;                       // dwords per voxel in LDS: B1 (16) + B2 (16) + 8 pad
    static constexpr int NV = LZ * LY * LX;             // staged voxels
    static constexpr int LDS_BYTES = LZ * LY * LXP * VS * 4;
    static constexpr int ITEMS = ((NV * 4 + NT - 1) / NT + 1) & ~1;      // (voxel, cin quad) items per thread, even: split two at a time
    static constexpr int RING = 3;                      // weight ring depth (taps); 27 % RING == 0 (9 on the 8^3 grids: measured equal, 250 VGPRs)
    static_assert(TY % (R * LPR) == 0 && NCT % CTW == 0 && 27 % RING == 0 && ITEMS <= 26 && (TXW == 16 || TXW == 8), "bad tile");
};

template <int CH, int TZ, int TY, int R, int CTW, int TXW = 16>
__global__ void __launch_bounds__((SplitCfg<CH, TZ, TY, R, CTW, TXW>::NT), (SplitCfg<CH, TZ, TY, R, CTW, TXW>::LDS_BYTES <= 80 * 1024 ? 2 : 1))
conv_k3s1_split_kernel(SplitArgs a) {
    using C = SplitCfg<CH, TZ, TY, R, CTW, TXW>;
    constexpr int NTAP = 27, RING = C::RING;
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int v = lane & 15, cq = lane >> 4;

    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = t % a.ntx; t /= a.ntx;
    const int ty = t % a.nty; t /= a.nty;
    const int tz = t % a.ntz;
    const int n = t / a.ntz;
    const int oz0 = tz * TZ, oy0 = ty * TY, ox0 = tx * TXW;
    const int iz0 = oz0 - 1, iy0 = oy0 - 1, ix0 = ox0 - 1;
    const int ct0 = (wave % C::NCG) * CTW;               // first cout tile of this wave
    const int w_yg = (wave / C::NCG) % C::NYG, w_z = wave / C::NCG / C::NYG;
    const int ly0 = w_yg * R * C::LPR + v / TXW, lx0 = v % TXW;      // row i of the wave: line ly0 + i * LPR
    const unsigned* lbase = lds + ((w_z * C::LY + ly0) * C::LXP + lx0) * C::VS + cq * 4;
    constexpr int ROW_OFF = C::LPR * C::LXP * C::VS;

    f32x4 acc[R][CTW];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) acc[i][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* inb = a.in + (size_t)n * a.D * a.H * a.W * CH;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(inb, (unsigned)a.D * a.H * a.W * CH * 4u);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, (unsigned)(C::NG * NTAP * C::NCT) * 2048u);
    const unsigned wlane = lane * 16;
    const int q_last = C::NG * NTAP - 1;
    auto tap_off = [](int kz, int ky, int kx) { return ((kz * C::LY + ky) * C::LXP + kx) * C::VS; };

    unsigned soff[C::ITEMS];
    unsigned lw[C::LXP != C::LX ? C::ITEMS : 1];          // LDS dword offset of the item's voxel when the line pitch is padded
#pragma unroll
    for (int it = 0; it < C::ITEMS; ++it) {
        const int item = stage_perm(it * C::NT + tid);
        const int u = item >> 2, q = item & 3;
        const int lz = u / (C::LY * C::LX), rem = u - lz * (C::LY * C::LX);
        const int ly = rem / C::LX, lx = rem - ly * C::LX;
        if constexpr (C::LXP != C::LX) lw[it] = (unsigned)((((lz * C::LY + ly) * C::LXP + lx) * C::VS) + q * 4);
        const int gz = iz0 + lz, gy = iy0 + ly, gx = ix0 + lx;
        const bool ok = (item < C::NV * 4) & (gz >= 0) & (gz < a.D) & (gy >= 0) & (gy < a.H) & (gx >= 0) & (gx < a.W);
        soff[it] = ok ? (unsigned)(((gz * a.H + gy) * a.W + gx) * CH + q * 4) * 4u : kOOB;
    }
    // split the staged fp32 items into their bf16 pieces and write B1 / B2 of (voxel, cin quad)
    auto commit = [&](const f32x4 (&stg)[C::ITEMS]) {
#pragma unroll
        for (int it = 0; it < C::ITEMS; it += 2) {
            u32x4 p1, p2, q1, q2;
            split_items2(p1, p2, q1, q2, stg[it], stg[it + 1]);
            const int i0 = stage_perm(it * C::NT + tid), i1 = stage_perm((it + 1) * C::NT + tid);
            const unsigned o0 = C::LXP != C::LX ? lw[C::LXP != C::LX ? it : 0] : (unsigned)((i0 >> 2) * C::VS + (i0 & 3) * 4);
            const unsigned o1 = C::LXP != C::LX ? lw[C::LXP != C::LX ? it + 1 : 0] : (unsigned)((i1 >> 2) * C::VS + (i1 & 3) * 4);
            if (i0 < C::NV * 4) {
                *reinterpret_cast<u32x4*>(lds + o0) = p1;
                *reinterpret_cast<u32x4*>(lds + o0 + 16) = p2;
            }
            if (i1 < C::NV * 4) {
                *reinterpret_cast<u32x4*>(lds + o1) = q1;
                *reinterpret_cast<u32x4*>(lds + o1 + 16) = q2;
            }
        }
    };

    constexpr int NRES = R * CTW;
    constexpr int RES0 = 27 - NRES;
    static_assert(NRES <= 27, "residual prefetch is spread over the tap sections");
    f32x4 stg[C::ITEMS];
#pragma unroll
    for (int it = 0; it < C::ITEMS; ++it) stg[it] = buf_load4(rin, soff[it], 0);
    u32x4 wf1[RING][CTW], wf2[RING][CTW];
#pragma unroll
    for (int r = 0; r < RING - 1; ++r)
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) {
            wf1[r][ct] = buf_load4u(rw, wlane, (unsigned)(min(r, q_last) * C::NCT + ct0 + ct) * 2048u);
            wf2[r][ct] = buf_load4u(rw, wlane, (unsigned)(min(r, q_last) * C::NCT + ct0 + ct) * 2048u + 1024u);
        }
    commit(stg);
    __syncthreads();

    const int gzo = oz0 + w_z;
    const bool has_res = (a.flags & PCC_CONV_ADD) != 0;
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(has_res ? a.res + (size_t)n * a.D * a.H * a.W * CH : a.in,
                                                  has_res ? (unsigned)a.D * a.H * a.W * CH * 4u : 0u);
    f32x4 resv[R][CTW];

#pragma unroll 1
    for (int g = 0; g < C::NG; ++g) {
        const unsigned gnext = (unsigned)min(g + 1, C::NG - 1) * 64u;     // last group: harmless re-read
        const bool last = g == C::NG - 1;
        u32x4 b1[2][R], b2[2][R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            b1[0][i] = *reinterpret_cast<const u32x4*>(lbase + i * ROW_OFF);
            b2[0][i] = *reinterpret_cast<const u32x4*>(lbase + i * ROW_OFF + 16);
        }
#pragma unroll
        for (int ts = 0; ts < NTAP; ++ts) {
            {
                const int q = min(g * NTAP + ts + RING - 1, q_last);
#pragma unroll
                for (int ct = 0; ct < CTW; ++ct) if (!(PCC_SPLIT_PROBE & 1)) {
                    wf1[(ts + RING - 1) % RING][ct] = buf_load4u(rw, wlane, (unsigned)(q * C::NCT + ct0 + ct) * 2048u);
                    wf2[(ts + RING - 1) % RING][ct] = buf_load4u(rw, wlane, (unsigned)(q * C::NCT + ct0 + ct) * 2048u + 1024u);
                }
                const int tn = (ts + 1 < NTAP) ? ts + 1 : ts;   // last tap: harmless re-read
                const int toff = tap_off(tn / 9, (tn / 3) % 3, tn % 3);
#pragma unroll
                for (int i = 0; i < R; ++i) if (!(PCC_SPLIT_PROBE & 2)) {
                    b1[(ts + 1) & 1][i] = *reinterpret_cast<const u32x4*>(lbase + toff + i * ROW_OFF);
                    b2[(ts + 1) & 1][i] = *reinterpret_cast<const u32x4*>(lbase + toff + i * ROW_OFF + 16);
                }
                if (ts < C::ITEMS && !(PCC_SPLIT_PROBE & 4)) stg[ts] = buf_load4(rin, soff[ts], gnext);
                if (ts >= RES0 && ts < RES0 + NRES) {
                    const int i = (ts - RES0) / CTW, ct = (ts - RES0) % CTW;
                    const int gy = oy0 + ly0 + i * C::LPR, gx = ox0 + lx0;
                    const bool ok = last & has_res & (gzo < a.D) & (gy < a.H) & (gx < a.W);
                    const unsigned off = (unsigned)(((gzo * a.H + gy) * a.W + gx) * CH + (ct0 + ct) * 16 + cq * 4) * 4u;
                    resv[i][ct] = buf_load4(rres, ok ? off : kOOB, 0);
                }
            }
            // the prefetches of the NEXT tap are issued before this tap's MFMAs, not sunk behind them (the scheduler otherwise moves
            // them to the end of the region, i.e. right in front of their first use: every tap then waits for its LDS reads)
            PCC_SPLIT_PIN();
            // three MFMAs per (row, cout tile); term outermost: consecutive MFMAs go to different accumulators
#pragma unroll
            for (int tm = 0; tm < 3; ++tm)
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int ct = 0; ct < CTW; ++ct)
                        if (!(PCC_SPLIT_PROBE & 8) || ts == 0)
                        acc[i][ct] = mfma_bf16(tm == 2 ? wf2[ts % RING][ct] : wf1[ts % RING][ct], tm == 1 ? b2[ts & 1][i] : b1[ts & 1][i], acc[i][ct]);
            PCC_SPLIT_PIN();
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
        const int gy = oy0 + ly0 + i * C::LPR, gx = ox0 + lx0;
        if (gzo < a.D && gy < a.H && gx < a.W) {
            const size_t vox = (((size_t)n * a.D + gzo) * a.H + gy) * a.W + gx;
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
                *reinterpret_cast<f32x4*>(a.out + vox * a.ocs + a.oco + c0) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Second formulation (round 4, late): v_mfma_f32_32x32x16_bf16.  The 16x16x32 kernel above stacks two product terms along K and
// needs [Wh | Wm], [Wl | Wh] x [dh | dm], [dl | dh] (the h pieces twice on both sides: 8 bytes per element) and re-fetches a 2 KB
// operand for every 3 MFMAs of 16 cycles.  With M = N = 32 and K = 16 (one product term, 16 input channels) the three pieces of both
// operands are used as they are -- acc += Wa . db for (a, b) in hh, hm, mh, mm, hl, lh: six MFMAs of 32 cycles per tap, 32 voxels,
// 32 output channels and 16-channel cin group, the same matrix time -- at 6 bytes per element and 3 KB of operand per side for 192
// MFMA cycles.  Per wave and tap: A 3 KB x CTW from L2 (32 B/clk/CU), B 3 KB x R rows from LDS (64 B/clk of 256): the launch is
// bound by the matrix pipe, not by its operand traffic.
//   lane (n = lane & 31, o = lane >> 5): B = the 8 input channels 8 o .. 8 o + 7 (of the staged 16-channel group) of voxel n of the
//   row; A = the same 8 input channels of output channel (32 half + n); D: voxel n, output channels 32 half + 8 q + 4 o .. + 3, q = 0..3.
//   LDS: 112 bytes per voxel: [h: 16 ch x 2 B][m][l][16 B pad]; 28 dwords = 4 x 7 keeps the 16 lanes of a ds_read_b128 group on
//   distinct bank quads.
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int CH, int TZ, int TY, int TXW, int R>
struct Split32Cfg {
    static constexpr int NG = CH / 16, NH = CH / 32;   // cin groups of 16, cout halves of 32
    static constexpr int LPR = 32 / TXW;                // y lines per row of 32 voxels
    static constexpr int NROW = TZ * TY / LPR;           // rows of the tile
    static constexpr int NW = NROW / R, NT = NW * 64;
    static constexpr int LZ = TZ + 2, LY = TY + 2, LX = TXW + 2;
    static constexpr int VS = 28;                       // dwords per voxel
    static constexpr int NV = LZ * LY * LX;
    static constexpr int LDS_BYTES = NV * VS * 4;
    static constexpr int ITEMS = (NV * 2 + NT - 1) / NT;      // (voxel, channel octet) items per thread: 2 loads, 3 stores each
    static constexpr int RING = 3;
    static_assert(NROW % R == 0 && NW == 4 && (TY % (R * LPR) == 0) && 2 * ITEMS <= 27, "bad tile");
};

template <int CH, int TZ, int TY, int TXW, int R>
__global__ void __launch_bounds__((Split32Cfg<CH, TZ, TY, TXW, R>::NT), 1) conv_k3s1_split32_kernel(SplitArgs a) {
    using C = Split32Cfg<CH, TZ, TY, TXW, R>;
    constexpr int NH = C::NH, NTAP = 27, RING = C::RING, LPR = C::LPR;
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 31, o = lane >> 5;

    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = t % a.ntx; t /= a.ntx;
    const int ty = t % a.nty; t /= a.nty;
    const int tz = t % a.ntz;
    const int n = t / a.ntz;
    const int oz0 = tz * TZ, oy0 = ty * TY, ox0 = tx * TXW;
    const int iz0 = oz0 - 1, iy0 = oy0 - 1, ix0 = ox0 - 1;
    // wave -> z plane and first row inside it; lane -> voxel of the row
    constexpr int RPZ = TY / LPR;                        // rows per z plane
    const int w_z = (wave * R) / RPZ, w_r0 = (wave * R) % RPZ;
    const int ly0 = w_r0 * LPR + (LPR == 2 ? (nl >> 4) : 0), lx0 = LPR == 2 ? (nl & 15) : nl;
    const unsigned* lbase = lds + ((w_z * C::LY + ly0) * C::LX + lx0) * C::VS + o * 4;
    constexpr int ROW_OFF = LPR * C::LX * C::VS;

    f32x16 acc[R][NH];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][h][e] = 0.f;

    const float* inb = a.in + (size_t)n * a.D * a.H * a.W * CH;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(inb, (unsigned)a.D * a.H * a.W * CH * 4u);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, (unsigned)(C::NG * NTAP * NH) * 3072u);
    const unsigned wlane = lane * 16;
    const int q_last = C::NG * NTAP - 1;
    auto tap_off = [](int kz, int ky, int kx) { return ((kz * C::LY + ky) * C::LX + kx) * C::VS; };

    unsigned soff[C::ITEMS];
#pragma unroll
    for (int it = 0; it < C::ITEMS; ++it) {
        const int item = it * C::NT + tid;
        const int u = item >> 1, oc = item & 1;
        const int lz = u / (C::LY * C::LX), rem = u - lz * (C::LY * C::LX);
        const int ly = rem / C::LX, lx = rem - ly * C::LX;
        const int gz = iz0 + lz, gy = iy0 + ly, gx = ix0 + lx;
        const bool ok = (item < C::NV * 2) & (gz >= 0) & (gz < a.D) & (gy >= 0) & (gy < a.H) & (gx >= 0) & (gx < a.W);
        soff[it] = ok ? (unsigned)(((gz * a.H + gy) * a.W + gx) * CH + oc * 8) * 4u : kOOB;
    }
    auto commit = [&](const f32x4 (&stg)[2 * C::ITEMS]) {
#pragma unroll
        for (int it = 0; it < C::ITEMS; ++it) {
            u32x4 p1, p2, q1, q2;
            split_items2(p1, p2, q1, q2, stg[2 * it], stg[2 * it + 1]);
            const int item = it * C::NT + tid;
            if (item < C::NV * 2) {
                unsigned* dst = lds + (item >> 1) * C::VS + (item & 1) * 4;
                *reinterpret_cast<u32x4*>(dst) = (u32x4){p1[0], p1[1], q1[0], q1[1]};          // h of the 8 channels
                *reinterpret_cast<u32x4*>(dst + 8) = (u32x4){p1[2], p1[3], q1[2], q1[3]};      // m
                *reinterpret_cast<u32x4*>(dst + 16) = (u32x4){p2[0], p2[1], q2[0], q2[1]};     // l
            }
        }
    };

    f32x4 stg[2 * C::ITEMS];
#pragma unroll
    for (int it = 0; it < C::ITEMS; ++it) { stg[2 * it] = buf_load4(rin, soff[it], 0); stg[2 * it + 1] = buf_load4(rin, soff[it], 16); }
    u32x4 wf[RING][NH][3];
#pragma unroll
    for (int r = 0; r < RING - 1; ++r)
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int p = 0; p < 3; ++p) wf[r][h][p] = buf_load4u(rw, wlane, (unsigned)((min(r, q_last) * NH + h) * 3 + p) * 1024u);
    commit(stg);
    __syncthreads();

    const int gzo = oz0 + w_z;
    const bool has_res = (a.flags & PCC_CONV_ADD) != 0;

#pragma unroll 1
    for (int g = 0; g < C::NG; ++g) {
        const unsigned gnext = (unsigned)min(g + 1, C::NG - 1) * 64u;
        const bool last = g == C::NG - 1;
        u32x4 b[2][R][3];
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) b[0][i][p] = *reinterpret_cast<const u32x4*>(lbase + i * ROW_OFF + p * 8);
#pragma unroll
        for (int ts = 0; ts < NTAP; ++ts) {
            {
                const int q = min(g * NTAP + ts + RING - 1, q_last);
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        if (!(PCC_SPLIT_PROBE & 1)) wf[(ts + RING - 1) % RING][h][p] = buf_load4u(rw, wlane, (unsigned)((q * NH + h) * 3 + p) * 1024u);
                const int tn = (ts + 1 < NTAP) ? ts + 1 : ts;
                const int toff = tap_off(tn / 9, (tn / 3) % 3, tn % 3);
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        if (!(PCC_SPLIT_PROBE & 2)) b[(ts + 1) & 1][i][p] = *reinterpret_cast<const u32x4*>(lbase + toff + i * ROW_OFF + p * 8);
                if (ts < 2 * C::ITEMS && !(PCC_SPLIT_PROBE & 4)) stg[ts] = buf_load4(rin, soff[ts >> 1], gnext + (unsigned)((ts & 1) * 16));
            }
            PCC_SPLIT_PIN();         // (prefetches first, see the 16x16x32 kernel)
            // six product terms; the term outermost so that consecutive MFMAs go to different accumulators
#pragma unroll
            for (int tm = 0; tm < 6; ++tm) {
                constexpr int PA[6] = {0, 1, 0, 1, 0, 2}, PB[6] = {0, 1, 1, 0, 2, 0};      // (W piece, d piece): hh mm hm mh hl lh
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int h = 0; h < NH; ++h) acc[i][h] = mfma32_bf16(wf[ts % RING][h][PA[tm]], b[ts & 1][i][PB[tm]], acc[i][h]);
            }
            PCC_SPLIT_PIN();
        }
        if (!last) {
            __syncthreads();
            commit(stg);
            __syncthreads();
        }
    }
    // ---- epilogue: lane (voxel nl of the row, octet o) holds output channels 32 h + 8 q + 4 o .. + 3
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(has_res ? a.res + (size_t)n * a.D * a.H * a.W * CH : a.in,
                                                  has_res ? (unsigned)a.D * a.H * a.W * CH * 4u : 0u);
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int gy = oy0 + ly0 + i * LPR, gx = ox0 + lx0;
        if (gzo < a.D && gy < a.H && gx < a.W) {
            const size_t vox = (((size_t)n * a.D + gzo) * a.H + gy) * a.W + gx;
            const unsigned rvo = (unsigned)(((gzo * a.H + gy) * a.W + gx) * CH) * 4u;
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = 32 * h + 8 * q + 4 * o;
                    f32x4 v = {acc[i][h][4 * q], acc[i][h][4 * q + 1], acc[i][h][4 * q + 2], acc[i][h][4 * q + 3]};
                    if (a.flags & PCC_CONV_BIAS) v += *reinterpret_cast<const f32x4*>(a.bias + c0);
                    if (a.flags & PCC_CONV_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if (has_res) v += buf_load4(rres, rvo + (unsigned)c0 * 4u, 0);
                    if (a.flags & PCC_CONV_CLIP01) {
                        v.x = fminf(fmaxf(v.x, 0.f), 1.f); v.y = fminf(fmaxf(v.y, 0.f), 1.f);
                        v.z = fminf(fmaxf(v.z, 0.f), 1.f); v.w = fminf(fmaxf(v.w, 0.f), 1.f);
                    }
                    *reinterpret_cast<f32x4*>(a.out + vox * a.ocs + a.oco + c0) = v;
                }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Conv3DTranspose k3 stride 2 (64 -> 32 @16^3 -> 32^3, 64 -> 64 @8^3 -> 16^3 of /root/reference/src/model_transforms.py:126-137) with
// split-bf16 operands.  Parity decomposition as in conv_tr2g_kernel (conv_mfma.hip): output voxel 2 b + p takes, per dimension,
// tap 1 of input b for odd p and taps 0 / 2 of inputs b / b - 1 for even p -- 8 parity classes of 8, 4, 4, 2, 4, 2, 2, 1 taps, 27 in
// all per input voxel, i.e. the multiply count of a k3 stride-1 layer per INPUT voxel and 8 outputs for it.  Same tile loop as
// conv_k3s1_split_kernel: per cin group the haloed input tile (halo on the low side only) is staged global -> registers -> split ->
// LDS once and feeds all 27 taps; the accumulators of all 8 classes stay live across the groups (8 x R x CTW float4), a class is
// stored right after its last tap in the last group, so the 8 x denser output stream is spread over that group.  Weights: the
// tr2g-order image [g][class][tap of the class][cout tile] split by pcc_tr2m_bf16_pack, 2 KB per fragment, ring of 3 from L2.
// fp32 accumulation in a fixed order (cin group -> tap -> three MFMAs): bit-deterministic and independent of the tiling.
struct Tr2Tap { int cls, dz, dy, dx; bool last; };
__host__ __device__ constexpr Tr2Tap tr2_tap(int want) {
    int seq = 0;
    for (int cls = 0; cls < 8; ++cls) {
        const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
        const int ntap = (pz ? 1 : 2) * (py ? 1 : 2) * (px ? 1 : 2);
        int t = 0;
        for (int kz = pz; kz < 3; kz += 2)
            for (int ky = py; ky < 3; ky += 2)
                for (int kx = px; kx < 3; kx += 2, ++seq, ++t)
                    if (seq == want) return Tr2Tap{cls, (pz - kz) / 2, (py - ky) / 2, (px - kx) / 2, t == ntap - 1};
    }
    return Tr2Tap{0, 0, 0, 0, false};
}
template <int... I, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

template <int CIN, int COUT, int TZ, int TY, int R, int CTW, int TXW>
struct Tr2SplitCfg {
    static constexpr int NG = CIN / 16, NCT = COUT / 16;
    static constexpr int NCG = NCT / CTW;
    static constexpr int LPR = 16 / TXW;
    static constexpr int NYG = TY / (R * LPR);
    static constexpr int NW = TZ * NYG * NCG, NT = NW * 64;
    static constexpr int LZ = TZ + 1, LY = TY + 1, LX = TXW + 1;      // taps reach b - 1 only
    static constexpr int VS = 40;
    static constexpr int NV = LZ * LY * LX;
    static constexpr int LDS_BYTES = NV * VS * 4;
    static constexpr int ITEMS = ((NV * 4 + NT - 1) / NT + 1) & ~1;
    static constexpr int RING = 3;
    static_assert(TY % (R * LPR) == 0 && NCT % CTW == 0 && ITEMS <= 26 && (TXW == 16 || TXW == 8), "bad tile");
};

template <int CIN, int COUT, int TZ, int TY, int R, int CTW, int TXW>
__global__ void __launch_bounds__((Tr2SplitCfg<CIN, COUT, TZ, TY, R, CTW, TXW>::NT), 1)      // 8 classes of accumulators: 512 registers per lane
conv_tr2_split_kernel(SplitArgs a) {
    using C = Tr2SplitCfg<CIN, COUT, TZ, TY, R, CTW, TXW>;
    constexpr int NTAP = 27, RING = C::RING;
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int v = lane & 15, cq = lane >> 4;

    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = t % a.ntx; t /= a.ntx;
    const int ty = t % a.nty; t /= a.nty;
    const int tz = t % a.ntz;
    const int n = t / a.ntz;
    const int bz0 = tz * TZ, by0 = ty * TY, bx0 = tx * TXW;
    const int ct0 = (wave % C::NCG) * CTW;
    const int w_yg = (wave / C::NCG) % C::NYG, w_z = wave / C::NCG / C::NYG;
    const int ly0 = w_yg * R * C::LPR + v / TXW, lx0 = v % TXW;
    // B operand base: voxel (w_z + 1, ly0 + 1, lx0 + 1) of the haloed tile (the halo is on the low side), channel quad cq
    const unsigned* lbase = lds + (((w_z + 1) * C::LY + ly0 + 1) * C::LX + lx0 + 1) * C::VS + cq * 4;
    constexpr int ROW_OFF = C::LPR * C::LX * C::VS;

    const float* inb = a.in + (size_t)n * a.D * a.H * a.W * CIN;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(inb, (unsigned)a.D * a.H * a.W * CIN * 4u);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, (unsigned)(C::NG * NTAP * C::NCT) * 2048u);
    const unsigned wlane = lane * 16;
    constexpr int q_last = C::NG * NTAP - 1;
    unsigned soff[C::ITEMS];
#pragma unroll
    for (int it = 0; it < C::ITEMS; ++it) {
        const int item = stage_perm(it * C::NT + tid);
        const int u = item >> 2, q = item & 3;
        const int lz = u / (C::LY * C::LX), rem = u - lz * (C::LY * C::LX);
        const int ly = rem / C::LX, lx = rem - ly * C::LX;
        const int gz = bz0 - 1 + lz, gy = by0 - 1 + ly, gx = bx0 - 1 + lx;
        const bool ok = (item < C::NV * 4) & (gz >= 0) & (gz < a.D) & (gy >= 0) & (gy < a.H) & (gx >= 0) & (gx < a.W);
        soff[it] = ok ? (unsigned)(((gz * a.H + gy) * a.W + gx) * CIN + q * 4) * 4u : kOOB;
    }
    auto commit = [&](const f32x4 (&stg)[C::ITEMS]) {
#pragma unroll
        for (int it = 0; it < C::ITEMS; it += 2) {
            u32x4 p1, p2, q1, q2;
            split_items2(p1, p2, q1, q2, stg[it], stg[it + 1]);
            const int i0 = stage_perm(it * C::NT + tid), i1 = stage_perm((it + 1) * C::NT + tid);
            if (i0 < C::NV * 4) {
                *reinterpret_cast<u32x4*>(lds + (i0 >> 2) * C::VS + (i0 & 3) * 4) = p1;
                *reinterpret_cast<u32x4*>(lds + (i0 >> 2) * C::VS + 16 + (i0 & 3) * 4) = p2;
            }
            if (i1 < C::NV * 4) {
                *reinterpret_cast<u32x4*>(lds + (i1 >> 2) * C::VS + (i1 & 3) * 4) = q1;
                *reinterpret_cast<u32x4*>(lds + (i1 >> 2) * C::VS + 16 + (i1 & 3) * 4) = q2;
            }
        }
    };

    f32x4 stg[C::ITEMS];
#pragma unroll
    for (int it = 0; it < C::ITEMS; ++it) stg[it] = buf_load4(rin, soff[it], 0);
    u32x4 wf1[RING][CTW], wf2[RING][CTW];
#pragma unroll
    for (int r = 0; r < RING - 1; ++r)
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) {
            wf1[r][ct] = buf_load4u(rw, wlane, (unsigned)(r * C::NCT + ct0 + ct) * 2048u);
            wf2[r][ct] = buf_load4u(rw, wlane, (unsigned)(r * C::NCT + ct0 + ct) * 2048u + 1024u);
        }
    f32x4 acc[8][R][CTW];
    {
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) {
            const f32x4 b4 = (a.flags & PCC_CONV_BIAS) ? *reinterpret_cast<const f32x4*>(a.bias + (ct0 + ct) * 16 + cq * 4) : zero4;
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int i = 0; i < R; ++i) acc[c][i][ct] = b4;
        }
    }
    commit(stg);
    __syncthreads();

    // per-row output byte offsets of class (0, 0, 0); a class adds a wave-uniform offset
    const int OH = 2 * a.H, OW = 2 * a.W;
    const size_t ovox_n = (size_t)8 * a.D * a.H * a.W;
    const int gzb = bz0 + w_z, gxb = bx0 + lx0;
    unsigned ooff[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int gyb = by0 + ly0 + i * C::LPR;
        const bool ok = gzb < a.D && gyb < a.H && gxb < a.W;
        const unsigned vox = (unsigned)((2 * gzb * OH + 2 * gyb) * OW + 2 * gxb);
        ooff[i] = ok ? (vox * (unsigned)a.ocs + (unsigned)a.oco + cq * 4) * 4u : kOOB;
    }
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(a.out + (size_t)n * ovox_n * a.ocs, (unsigned)(ovox_n * a.ocs * 4u));
    const float relu_lo = (a.flags & PCC_CONV_RELU) ? 0.f : -__builtin_inff();
    float mx = 0.f;      // max |stored value| of this lane (amax_out)

    auto group = [&](auto last_tag, int g) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_tag)::value;
        const unsigned gnext = (unsigned)(g + 1) * 64u;
        u32x4 b1[2][R], b2[2][R];
#pragma unroll
        for (int i = 0; i < R; ++i) {       // tap 0: class (0,0,0), (dz, dy, dx) = (0, 0, 0)
            b1[0][i] = *reinterpret_cast<const u32x4*>(lbase + i * ROW_OFF);
            b2[0][i] = *reinterpret_cast<const u32x4*>(lbase + i * ROW_OFF + 16);
        }
        static_for(std::make_integer_sequence<int, NTAP>{}, [&](auto seq_tag) __attribute__((always_inline)) {
            constexpr int ts = decltype(seq_tag)::value;
            constexpr Tr2Tap T = tr2_tap(ts);
            constexpr Tr2Tap Tn = tr2_tap(ts + 1 < NTAP ? ts + 1 : ts);
            constexpr int next_off = ((Tn.dz * C::LY + Tn.dy) * C::LX + Tn.dx) * C::VS;
            constexpr int cls = T.cls;
            constexpr int WS = ts % RING;      // register slot of this tap's weights
            {
                const int q = min(g * NTAP + ts + RING - 1, q_last);
#pragma unroll
                for (int ct = 0; ct < CTW; ++ct) {
                    wf1[(ts + RING - 1) % RING][ct] = buf_load4u(rw, wlane, (unsigned)(q * C::NCT + ct0 + ct) * 2048u);
                    wf2[(ts + RING - 1) % RING][ct] = buf_load4u(rw, wlane, (unsigned)(q * C::NCT + ct0 + ct) * 2048u + 1024u);
                }
                if constexpr (ts + 1 < NTAP) {
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        b1[(ts + 1) & 1][i] = *reinterpret_cast<const u32x4*>(lbase + next_off + i * ROW_OFF);
                        b2[(ts + 1) & 1][i] = *reinterpret_cast<const u32x4*>(lbase + next_off + i * ROW_OFF + 16);
                    }
                }
                if constexpr (!LAST && ts < C::ITEMS) stg[ts] = buf_load4(rin, soff[ts], gnext);
            }
            PCC_SPLIT_PIN();
#pragma unroll
            for (int tm = 0; tm < 3; ++tm)
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int ct = 0; ct < CTW; ++ct)
                        acc[cls][i][ct] = mfma_bf16(tm == 2 ? wf2[WS][ct] : wf1[WS][ct], tm == 1 ? b2[ts & 1][i] : b1[ts & 1][i], acc[cls][i][ct]);
            PCC_SPLIT_PIN();
            if constexpr (LAST && T.last) {      // the class is complete: ReLU and stores (the bias is in the accumulators)
                constexpr int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
                const unsigned coff = (unsigned)(((pz * OH + py) * OW + px) * a.ocs) * 4u;
#pragma unroll
                for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        f32x4 o = acc[cls][i][ct];
                        o = __builtin_elementwise_maximum(o, (f32x4){relu_lo, relu_lo, relu_lo, relu_lo});
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rout, (int)(ooff[i] + coff + (unsigned)((ct0 + ct) * 16) * 4u), 0, 0);
                        // rows beyond the volume (their stores are dropped by the range check) stay out of the block's maximum
                        const float m4 = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
                        mx = ooff[i] < kOOB ? fmaxf(mx, m4) : mx;
                    }
            }
        });
    };
#pragma unroll 1
    for (int g = 0; g < C::NG - 1; ++g) {
        group(std::false_type{}, g);
        __syncthreads();   // every wave finished reading group g
        commit(stg);
        __syncthreads();
    }
    group(std::true_type{}, C::NG - 1);
    // ---- max |out| of block n for the fp16-split layer behind this one (conv_wino_f16s.hip)
    if (a.amax_out != nullptr) pcc_amax_record(a.amax_out + (size_t)n * PCC_AMAX_SLOTS, mx, (int)blockIdx.x * (C::NT / 64) + wave);
}

}  // namespace pccsplit

using namespace pccsplit;

// ---- host: split image of the logical forward weights.  wlog: [kz][ky][kx][ci][co] (already flipped for transposed layers);
//      out: [cin group][tap][cout tile][operand][lane][8 bf16]; operand 0 = [Wh c0..c3 | Wm], 1 = [Wl | Wh]; cin = 16 g + 4 (lane >> 4) + c,
//      cout = 16 ct + (lane & 15)
static inline unsigned short bf16_rn(float v) {
    unsigned b;
    memcpy(&b, &v, 4);
    if ((b & 0x7f800000u) == 0x7f800000u) return (unsigned short)(b >> 16);
    b += 0x7fffu + ((b >> 16) & 1u);
    return (unsigned short)(b >> 16);
}
static inline float bf16_f(unsigned short h) {
    const unsigned b = (unsigned)h << 16;
    float v;
    memcpy(&v, &b, 4);
    return v;
}
// two images: the 16x16x32 kernel's ([g][tap][ct][2 operands][lane][8 bf16]) and, behind it, the 32x32x16 kernel's
// ([g][tap][cout half][piece h, m, l][lane][8 bf16]: lane (n, o) = output channel 32 half + n, input channels 16 g + 8 o .. + 7)
static size_t split16_floats(int C) { return (size_t)(C / 16) * 27 * (C / 16) * 2 * 64 * 4; }
static size_t split32_floats(int C) { return (size_t)(C / 16) * 27 * (C / 32) * 3 * 64 * 4; }
size_t pcc_split_packed_floats(int C) { return split16_floats(C) + split32_floats(C); }
void pcc_split_pack(int C, const float* wlog, float* out) {
    unsigned short* o = reinterpret_cast<unsigned short*>(out);
    const int NG = C / 16;
    for (int g = 0; g < NG; ++g)
        for (int tap = 0; tap < 27; ++tap)
            for (int ct = 0; ct < NG; ++ct)
                for (int lane = 0; lane < 64; ++lane) {
                    unsigned short h[4], m[4], l[4];
                    for (int c = 0; c < 4; ++c) {
                        const float x = wlog[((size_t)tap * C + g * 16 + 4 * (lane >> 4) + c) * C + ct * 16 + (lane & 15)];
                        h[c] = bf16_rn(x);
                        const float r1 = x - bf16_f(h[c]);
                        m[c] = bf16_rn(r1);
                        l[c] = bf16_rn(r1 - bf16_f(m[c]));
                    }
                    unsigned short* a1 = o + (((((size_t)g * 27 + tap) * NG + ct) * 2 + 0) * 64 + lane) * 8;
                    unsigned short* a2 = o + (((((size_t)g * 27 + tap) * NG + ct) * 2 + 1) * 64 + lane) * 8;
                    for (int c = 0; c < 4; ++c) { a1[c] = h[c]; a1[4 + c] = m[c]; a2[c] = l[c]; a2[4 + c] = h[c]; }
                }
    unsigned short* o3 = o + split16_floats(C) * 2;
    for (int g = 0; g < NG; ++g)
        for (int tap = 0; tap < 27; ++tap)
            for (int half = 0; half < C / 32; ++half)
                for (int lane = 0; lane < 64; ++lane)
                    for (int c = 0; c < 8; ++c) {
                        const float x = wlog[((size_t)tap * C + g * 16 + 8 * (lane >> 5) + c) * C + half * 32 + (lane & 31)];
                        const unsigned short hh = bf16_rn(x);
                        const float r1 = x - bf16_f(hh);
                        const unsigned short mm = bf16_rn(r1);
                        const unsigned short ll = bf16_rn(r1 - bf16_f(mm));
                        const size_t base = ((((size_t)g * 27 + tap) * (C / 32) + half) * 3) * 64;
                        o3[((base + 0 * 64 + lane) * 8) + c] = hh;
                        o3[((base + 1 * 64 + lane) * 8) + c] = mm;
                        o3[((base + 2 * 64 + lane) * 8) + c] = ll;
                    }
}

// ---- Conv3DTranspose k3 stride 2 with split operands: 64 -> 32 and 64 -> 64 (weights: pcc_tr2m_bf16_pack of the tr2g-order image)
bool pcc_tr2_split_covers(const pcc_conv_desc* d) {
    if (!(d->transposed && d->k == 3 && d->stride == 2 && d->Cin == 64 && (d->Cout == 32 || d->Cout == 64))) return false;
    if (d->flags & ~(PCC_CONV_BIAS | PCC_CONV_RELU)) return false;
    if (d->W % 16 && d->W != 8) return false;
    const int ocs = d->out_cstride ? d->out_cstride : d->Cout;
    if (ocs % 4 || d->out_coffset % 4) return false;
    return (double)d->D * d->H * d->W * 8.0 * ocs * 4.0 < 2147483648.0 && (double)d->D * d->H * d->W * d->Cin * 4.0 < 2147483648.0;
}

int pcc_conv_tr2_split(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w_split, const float* bias, float* out,
                       pcc_conv_ext* ext, hipStream_t st) {
    (void)ctx;
    PCC_REQUIRE(pcc_tr2_split_covers(d), "pcc_conv_tr2_split: shape not covered");
    SplitArgs a;
    a.in = in; a.w = w_split; a.bias = bias; a.res = nullptr; a.out = out;
    a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W;
    a.flags = d->flags; a.ocs = d->out_cstride ? d->out_cstride : d->Cout; a.oco = d->out_coffset;
    if (ext && ext->out_amax) { a.amax_out = ext->out_amax; ext->out_recorded = true; }
#define PCC_TR2S_LAUNCH(CO, TZ, TY, R, CTW, TXW)                                                                  \
    {                                                                                                              \
        using C = Tr2SplitCfg<64, CO, TZ, TY, R, CTW, TXW>;                                                        \
        a.ntz = (d->D + TZ - 1) / TZ; a.nty = (d->H + TY - 1) / TY; a.ntx = d->W / TXW;                            \
        const int grid = d->N * a.ntz * a.nty * a.ntx;                                                             \
        const void* kern = (const void*)conv_tr2_split_kernel<64, CO, TZ, TY, R, CTW, TXW>;                        \
        { const int rc = pcc_enable_big_lds(kern, C::LDS_BYTES); if (rc != PCC_OK) return rc; }                    \
        hipLaunchKernelGGL((conv_tr2_split_kernel<64, CO, TZ, TY, R, CTW, TXW>), dim3((unsigned)grid), dim3(C::NT), C::LDS_BYTES, st, a); \
        PCC_CHECK_HIP(hipGetLastError());                                                                          \
        return PCC_OK;                                                                                             \
    }
    if (d->Cout == 32) { if (d->W == 8) PCC_TR2S_LAUNCH(32, 2, 8, 2, 2, 8) else PCC_TR2S_LAUNCH(32, 2, 8, 4, 2, 16) }
    else { if (d->W == 8) PCC_TR2S_LAUNCH(64, 1, 8, 2, 2, 8) else PCC_TR2S_LAUNCH(64, 1, 8, 4, 2, 16) }
#undef PCC_TR2S_LAUNCH
}

bool pcc_split_covers(const pcc_conv_desc* d) {
    if (!(d->Cin == d->Cout && (d->Cin == 32 || d->Cin == 64) && d->k == 3 && d->stride == 1)) return false;
    if (d->W % 16 && !(d->Cin == 64 && d->W == 8)) return false;      // (8-wide grids: the 64-channel layers at 8^3)
    const int ocs = d->out_cstride ? d->out_cstride : d->Cout;
    if (ocs % 4 || d->out_coffset % 4) return false;
    return (double)d->D * d->H * d->W * d->Cin * 4.0 < 2147483648.0;        // one image per buffer descriptor
}

// AUTO takes this path for the 64-channel layers (tile = 2 x 4 x 16 voxels, two workgroups per CU).
// Measured (batch 32, tools/bench_one.py): 64 -> 64 @16^3 128 us against 138 - 146 us for conv16_wino_cin_kernel<4> and 234 us for the
// exact-fp32 direct kernel; 32 -> 32 @32^3 333 us against 241 us (Winograd) and 468 us (fp32 direct): with R x CTW = 4 x 1 MFMA
// groups per operand fetch the 32-channel launch is bound by its operand traffic (5 GB of weight fragments from L2, 15 GB of input
// fragments from LDS per launch -- MFMA busy 0.29 - 0.46), so the Winograd kernel keeps those layers (DESIGN_HISTORY.md 3.0d).
// The choice depends on the layer shape ONLY, never on the batch: encoder and decoder run with different batch sizes and must produce
// the same bits (tests/test_codec_gpu.py::test_blocks_128_cubed_roundtrip_and_layer_parity caught a batch-dependent rule).
bool pcc_split_preferred(const pcc_ctx* ctx, const pcc_conv_desc* d) {
    (void)ctx;
    // 64 channels: 128 us against 138 - 146 (Winograd fp32) @16^3 x 32.  32 channels: only on the small grids (38.5 against 45 us @16^3;
    // at 32^3 the Winograd kernel's 241 us stand against 332: every tile of this kernel pays its staging, split and epilogue
    // un-overlapped -- at bf16 MFMA rates they are as long as the 27 taps themselves, see DESIGN_HISTORY.md 3.0d)
    return d->Cin == 64 || (d->Cin == 32 && d->D <= 16 && (d->W % 32 == 0 || d->W == 16));
}

int pcc_conv_split(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w_split, const float* bias, const float* residual,
                   float* out, hipStream_t st) {
    PCC_REQUIRE(pcc_split_covers(d), "pcc_conv_split: shape not covered");
    SplitArgs a;
    a.in = in; a.w = w_split; a.bias = bias; a.res = residual; a.out = out;
    a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W;
    a.flags = d->flags; a.ocs = d->out_cstride ? d->out_cstride : d->Cout; a.oco = d->out_coffset;
    // Which formulation: measured at batch 32 (tools/bench_one.py) -- 64 -> 64 @16^3: 128 us (16x16x32, tile 2 x 4 x 16) / 134 - 148 (32x32x16);
    // 32 -> 32 @16^3: 41 / 38.5 us; 32 -> 32 @32^3: 332 / 373 us.  A function of the layer shape only.  PCC_SPLIT_MFMA=16 | 32 overrides (A/B).
    const int force = ctx->num(PCC_NUM_SPLIT_MFMA16) ? 16 : ctx->num(PCC_NUM_SPLIT_MFMA32) ? 32 : 0;
    const bool use32 = force ? force == 32 : d->Cin == 32;
    if (d->W == 8) {
        // 64 -> 64 on the 8^3 grids (analysis block 3, first / last hyper layers): one z plane x 8 lines x 8 voxels per workgroup =
        // 8 planes x 32 blocks = 256 workgroups of 4 waves (2 rows x 2 cout tiles each); the fp32 kernel took 36.7 us per launch
        using C = SplitCfg<64, 1, 8, 2, 2, 8>;
        a.ntz = d->D; a.nty = (d->H + 7) / 8; a.ntx = 1;
        const void* kern = (const void*)conv_k3s1_split_kernel<64, 1, 8, 2, 2, 8>;
        { const int rc = pcc_enable_big_lds(kern, C::LDS_BYTES); if (rc != PCC_OK) return rc; }
        hipLaunchKernelGGL((conv_k3s1_split_kernel<64, 1, 8, 2, 2, 8>), dim3((unsigned)(d->N * a.ntz * a.nty)), dim3(C::NT), C::LDS_BYTES, st, a);
        PCC_CHECK_HIP(hipGetLastError());
        return PCC_OK;
    }
    if (use32 && (d->W % 32 == 0 || d->W == 16)) {
        a.w = w_split + split16_floats(d->Cin);
#define PCC_SPLIT32_LAUNCH(CH, TZ, TY, TXW, R)                                                                     \
    {                                                                                                              \
        using C = Split32Cfg<CH, TZ, TY, TXW, R>;                                                                  \
        a.ntz = (d->D + TZ - 1) / TZ; a.nty = (d->H + TY - 1) / TY; a.ntx = d->W / TXW;                            \
        const int grid = d->N * a.ntz * a.nty * a.ntx;                                                             \
        const void* kern = (const void*)conv_k3s1_split32_kernel<CH, TZ, TY, TXW, R>;                              \
        { const int rc = pcc_enable_big_lds(kern, C::LDS_BYTES); if (rc != PCC_OK) return rc; }                    \
        hipLaunchKernelGGL((conv_k3s1_split32_kernel<CH, TZ, TY, TXW, R>), dim3((unsigned)grid), dim3(C::NT), C::LDS_BYTES, st, a); \
        PCC_CHECK_HIP(hipGetLastError());                                                                          \
        return PCC_OK;                                                                                             \
    }
        if (d->Cin == 32) { if (d->W % 32 == 0) PCC_SPLIT32_LAUNCH(32, 2, 4, 32, 2) else PCC_SPLIT32_LAUNCH(32, 2, 8, 16, 2) }
        else { if (d->W % 32 == 0) PCC_SPLIT32_LAUNCH(64, 2, 4, 32, 2) else PCC_SPLIT32_LAUNCH(64, 2, 8, 16, 2) }
#undef PCC_SPLIT32_LAUNCH
    }
    // tile 2 x 4 x 16 (69 KB of LDS, two workgroups per CU: the second wave of a SIMD covers the first one's LDS / L2 waits) or
    // 2 x 8 x 16 (115 KB, one workgroup per CU, fewer halo voxels); PCC_SPLIT_TILE=8 selects the large one (A/B)
    const bool big = ctx->num(PCC_NUM_SPLIT_TILE8);
#define PCC_SPLIT_LAUNCH(CH, TZ, TY, R)                                                                            \
    {                                                                                                              \
        using C = SplitCfg<CH, TZ, TY, R, CH / 32>;                                                                \
        a.ntz = (d->D + TZ - 1) / TZ; a.nty = (d->H + TY - 1) / TY; a.ntx = d->W / 16;                             \
        const int grid = d->N * a.ntz * a.nty * a.ntx;                                                             \
        const void* kern = (const void*)conv_k3s1_split_kernel<CH, TZ, TY, R, CH / 32>;                            \
        { const int rc = pcc_enable_big_lds(kern, C::LDS_BYTES); if (rc != PCC_OK) return rc; }                    \
        hipLaunchKernelGGL((conv_k3s1_split_kernel<CH, TZ, TY, R, CH / 32>), dim3((unsigned)grid), dim3(C::NT), C::LDS_BYTES, st, a); \
    }
    if (d->Cin == 32) { if (big) PCC_SPLIT_LAUNCH(32, 2, 8, 8) else PCC_SPLIT_LAUNCH(32, 2, 4, 4) }
    else { if (big) PCC_SPLIT_LAUNCH(64, 2, 8, 8) else PCC_SPLIT_LAUNCH(64, 2, 4, 4) }
#undef PCC_SPLIT_LAUNCH
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}
