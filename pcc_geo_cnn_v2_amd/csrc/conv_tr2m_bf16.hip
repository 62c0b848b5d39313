// Conv3DTranspose, k = 3, stride 2, 32 -> 16, marching along z, on the bf16 MFMA pipe with fp32-EQUIVALENT operands (round 4).
//
// conv_tr2m_kernel<2> (conv_tr2m.hip) ran the first layer of the last SynthesisBlock (/root/reference/src/model_transforms.py:78:
// 32 -> 16 @32^3 -> 64^3) at 0.67 of the fp32-MFMA peak: 432 fp32 MFMAs per micro-step behind 16 B-operand vectors that are read
// from LDS once.  That structure is what the operand split wants: the 16 operand vectors of a micro-step feed 27 taps x 4 rows x 3 =
// 324 v_mfma_f32_16x16x32_bf16 of 16 cycles instead of 432 x 32: 5.2 k instead of 13.8 k MFMA cycles per micro-step.  The split itself
// happens ONCE PER TILE ELEMENT (late round 4): the tile of the next micro-step travels global -> registers -> split -> LDS in operand
// form (160 bytes per voxel as in conv_split.hip: [dh | dm] x 4 channel quads, [dl | dh] x 4, pad), 5 items per thread, and the 16
// vectors of a lane are 32 plain ds_read_b128.  (First version: the fp32 tile by LDS-direct loads and 8 split blocks per lane and
// micro-step on the 16 vectors -- the same voxel split up to four times, for its two x and two y offsets; 203 -> 173 us of that launch
// were the split, a probe build showed.)  x = h + m + l (three bf16 pieces), terms hh hm mh hl mm lh, fp32
// accumulation in a fixed order, as conv_wino_bf16.hip / conv_split.hip:
//     acc += [Wh | Wm] . [dh | dm];   acc += [Wh | Wm] . [dl | dh];   acc += [Wl | Wh] . [dh | dm]
// The split weights of the cout tile (2 x 27 fragments of 2 KB = 108 KB) stay LDS-resident beside the two-tile ring (40 KB); NG = 4
// (64 -> 32: 216 KB) does not fit and keeps the fp32 kernel / conv_tr2g_kernel.  Everything else -- parity decomposition, three
// accumulator sets, epilogue under the first taps of the next plane, compile-time plane parity -- is conv_tr2m_kernel's.
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

#include "common.h"

namespace pcctr2mb {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma_bf16(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Two B-operand vectors (4 input channels of a voxel each): fp32 -> B1 = [dh | dm], B2 = [dl | dh] each.  ONE asm block, because
// v_dot2c_f32_bf16 is a DOT instruction: a different VALU op that reads its result needs 3 wait states behind it
// (GCNHazardRecognizer: DotWriteDifferentVALURead) and the hazard recogniser cannot see into inline asm.  Inside the block every
// reader sits >= 3 instructions behind its writer; K0 / K1 = the bf16 pairs {-1, 0} / {0, -1}: x -= lo(h) / hi(h), exactly.
__device__ __forceinline__ void split_vec2(u32x4& p_b1, u32x4& p_b2, u32x4& q_b1, u32x4& q_b2, const f32x4& pv, const f32x4& qv) {
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


__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
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
__device__ __forceinline__ f32x4 acc_read(const f32x4& a) {
    f32x4 d;
    asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]) : "a"(a[0]), "a"(a[1]), "a"(a[2]), "a"(a[3]));
    return d;
}

constexpr int NT = 256;
constexpr int LXY = 17;                                 // tile edge incl. the low-side halo (taps reach b - 1 only)
constexpr int TILE_SLOTS = LXY * LXY * 4;               // 16-byte slots of one (plane, cin group) tile: 1156
constexpr int ITEMS = 5;                                // (voxel, channel quad) items per thread: 5 x 256 = 1280 >= 1156
constexpr int VSB = 160;                                // bytes per voxel of the operand-form tile: B1 x 4 quads, B2 x 4 quads, 32 pad (10 bank quads:
                                                        // the 8 voxels x 2 quads of a ds_read_b128 lane group fall on 16 different ones)
constexpr int TILE_BYTES = ITEMS * 64 * VSB;            // 51200: 320 voxel slots (289 used; the items past the tile write zeros into the rest)
constexpr int W_BASE = TILE_BYTES;                      // one tile, then the weights
constexpr int ROWB = LXY * VSB;                         // bytes per tile row

struct Tr2mArgs {
    const float* in;
    const float* w;      // split image in conv_tr2g order: [cin group][27 taps, class-major][cout tile][operand][64 lanes][8 bf16]
    const float* bias;
    float* out;
    int N, D, H, W;      // input dims (output = 2x)
    int nty, ntx, zsplit, zlen, nct;
    int flags, ocs, oco;
    unsigned* amax_out = nullptr;      // per-block max |out| for the fp16-split layer behind this one (common.h, pcc_conv_ext)
};

// tap t = 0..26 of a micro-step, kz-major; within a kz the (ky, kx) order keeps equal input offsets together and lets the
// first four taps open the four parity classes
struct Tap { int kz, ky, kx, cls, dyi, dxi, sq; bool opens; };
__host__ __device__ constexpr int tr2g_seq(int kz, int ky, int kx) {      // position in the packed (class-major) weight order
    int seq = 0;
    for (int cls = 0; cls < 8; ++cls) {
        const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
        for (int z = pz; z < 3; z += 2)
            for (int y = py; y < 3; y += 2)
                for (int x = px; x < 3; x += 2, ++seq)
                    if (z == kz && y == ky && x == kx) return seq;
    }
    return -1;
}
__host__ __device__ constexpr Tap tap_of(int t) {
    constexpr int KY[9] = {0, 0, 1, 1, 0, 1, 2, 2, 2}, KX[9] = {0, 1, 0, 1, 2, 2, 0, 1, 2};
    const int kz = t / 9, r = t % 9, ky = KY[r], kx = KX[r];
    return Tap{kz, ky, kx, (ky & 1) * 2 + (kx & 1), ky == 2 ? 1 : 0, kx == 2 ? 1 : 0, tr2g_seq(kz, ky, kx), r < 4};
}

template <int... I, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

template <int NG, bool RELU>
__global__ void __launch_bounds__(NT, 1) conv_tr2m_bf16_kernel(Tr2mArgs a, int nwg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int v = lane & 15, cq = lane >> 4;
    auto ldsu = [&](unsigned off) -> u32x4 { return *reinterpret_cast<const u32x4*>(smem + off); };
    typedef __attribute__((address_space(3))) void* lds_ptr;

    int wg = xcd_remap(blockIdx.x, nwg);
    const int ct = wg % a.nct; wg /= a.nct;          // cout tile: neighbours in the grid share their input tiles in L2
    const int tx_ = wg % a.ntx; wg /= a.ntx;
    const int ty_ = wg % a.nty; wg /= a.nty;
    const int zs = wg % a.zsplit;
    const int n = wg / a.zsplit;
    const int X0 = tx_ * 16, Y0 = ty_ * 16, zb = zs * a.zlen;
    const int nsteps = a.zlen + 1;                   // input planes zb-1 .. zb+zlen-1 (the first one only opens output plane 2 zb)
    constexpr int CIN = NG * 16;
    const size_t HW = (size_t)a.H * a.W;
    const unsigned HWI = (unsigned)(HW * CIN * 4);
    const float* in_n = a.in + (size_t)n * a.D * HW * CIN;

    // ---- split weights of this cout tile -> LDS (resident): NG x 27 fragments of 2 KB ([Wh | Wm] then [Wl | Wh]), fragment (g, sq) at
    //      W_BASE + (g * 27 + sq) * 2 KB
    {
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, (unsigned)(NG * 27 * a.nct) * 2048u);
        for (int p = wave; p < NG * 27 * 2; p += 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(smem + W_BASE + p * 1024), 16, lane * 16, ((p >> 1) * a.nct + ct) * 2048 + (p & 1) * 1024, 0, 0);
    }
    // ---- tile staging: global -> registers (one micro-step ahead) -> split -> LDS.  Item it of thread tid = (voxel u, channel quad q) =
    //      ((it * 256 + tid) >> 2, tid & 3); its operands go to u * 160 + q * 16 (B1) and + 64 (B2).  OOB items read zeros.
    // (round 5) the four voxels of 16 consecutive lanes are taken in the order 0, 2, 1, 3: a ds_write_b128 is served in groups of 8 lanes on 32
    // banks, and two voxels 160 B apart overlap in 8 of them (2-way conflicts on every staging write: the 0.10 - 0.18 of the PMC rows) -- 320 B apart
    // they use the other 16 banks
    const int gperm = ((tid >> 2) & ~3) | (((tid >> 2) & 1) << 1) | (((tid >> 2) >> 1) & 1);
    unsigned rel[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int item = it * NT + tid;
        const int u = it * 64 + gperm, q = item & 3, ly = u / LXY, lx = u - ly * LXY;
        const int y = Y0 - 1 + ly, x = X0 - 1 + lx;
        const bool ok = item < TILE_SLOTS && y >= 0 && y < a.H && x >= 0 && x < a.W;
        rel[it] = ok ? (unsigned)(((y * a.W + x) * CIN + q * 4) * 4) : kOOB;
    }
    const unsigned cw = (unsigned)(gperm * VSB + (tid & 3) * 16);       // item it: cw + it * 64 * VSB
    f32x4 stg[ITEMS];
    // tile of (step sp, cin group cg): `addr` = address of channel 16 cg of input plane zb - 1 + sp (kept incrementally)
    auto fetch_tile = [&](int sp, int cg, unsigned long long addr) __attribute__((always_inline)) {
        const bool ok = (unsigned)(zb - 1 + sp) < (unsigned)a.D && sp < nsteps;
        const __amdgpu_buffer_rsrc_t rp = make_rsrc((const void*)(ok ? addr : (unsigned long long)in_n), ok ? HWI - (unsigned)(64 * cg) : 0u);
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) stg[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, (int)rel[it], 0, 0));
    };
    // items j, j + 1 (j = 0, 2) or the single item 4: split and written
    auto commit2 = [&](auto j_tag) __attribute__((always_inline)) {
        constexpr int j = decltype(j_tag)::value, j1 = j + 1 < ITEMS ? j + 1 : j;
        u32x4 p1, p2, q1, q2;
        split_vec2(p1, p2, q1, q2, stg[j], stg[j1]);
        *reinterpret_cast<u32x4*>(smem + cw + (unsigned)(j * 64 * VSB)) = p1;
        *reinterpret_cast<u32x4*>(smem + cw + (unsigned)(j * 64 * VSB + 64)) = p2;
        if constexpr (j1 != j) {
            *reinterpret_cast<u32x4*>(smem + cw + (unsigned)(j1 * 64 * VSB)) = q1;
            *reinterpret_cast<u32x4*>(smem + cw + (unsigned)(j1 * 64 * VSB + 64)) = q2;
        }
    };

    // ---- B operand addresses: lane (v, cq) reads voxel (row, lx = v + 1 - dx) of the tile, its channel quad cq; one base per x offset,
    //      rows / y offsets and the operand (B2 = + 64) are immediates.  ba[dxi] points at tile row 4 * wave.
    unsigned ba[2];
#pragma unroll
    for (int dxi = 0; dxi < 2; ++dxi) ba[dxi] = (unsigned)((4 * wave * LXY + v + 1 - dxi) * VSB + cq * 16);
    unsigned wa = (unsigned)(W_BASE + lane * 16);      // + cin group * 54 KB (per micro-step), + tap * 2 KB (immediate)

    // ---- epilogue addressing: lane writes couts 4 cq .. 4 cq + 3 (of this cout tile) of output voxel (2 z + pz, 2 y + py, 2 x + px)
    const int OH = 2 * a.H, OW = 2 * a.W;
    const unsigned ob = (unsigned)((((2 * (Y0 + 4 * wave)) * OW + 2 * (X0 + v)) * a.ocs + a.oco + 16 * ct + 4 * cq) * 4);
    const unsigned PLANE_O = (unsigned)((size_t)OH * OW * a.ocs * 4);          // bytes per output plane (launcher: two planes < 2^31)
    float* out_n = a.out + (size_t)n * (2 * a.D) * OH * OW * a.ocs;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bias_l = (a.flags & PCC_CONV_BIAS) ? *reinterpret_cast<const f32x4*>(a.bias + 16 * ct + 4 * cq) : zero4;

    // ---- prologue
    // (s2, c2, tile_pl): input plane step / cin group / address of the next tile to fetch (wave-uniform)
    int s2 = 0, c2 = 0;
    unsigned long long tile_pl = (unsigned long long)in_n + (unsigned long long)(long long)(zb - 1) * HWI;     // tile of micro-step 0
    auto fetch_next = [&]() __attribute__((always_inline)) {
        fetch_tile(s2, c2, tile_pl);
        if (++c2 == NG) { c2 = 0; ++s2; tile_pl += HWI - 64 * (NG - 1); } else tile_pl += 64;
    };
    fetch_next();
    commit2(std::integral_constant<int, 0>{}); commit2(std::integral_constant<int, 2>{}); commit2(std::integral_constant<int, 4>{});
    fetch_next();                                     // raw tile of micro-step 1 waits in registers
    __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): the weights' LDS-direct loads
    __syncthreads();

    f32x4 E[2][4][4], O[4][4];                        // accumulators [set][class py * 2 + px][row]
    u32x4 b1[2][2][4], b2[2][2][4];                   // split B operands [dyi][dxi][row] of the current micro-step: [dh | dm], [dl | dh]
    u32x4 wf1[3], wf2[3];                             // weight fragment ring, two taps ahead (a tap is 12 MFMAs = 192 cycles: one tap does not cover an LDS read): [Wh | Wm], [Wl | Wh]
    // the 16 operand vectors of the micro-step: 32 plain LDS reads
    auto load_b = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int dyi = 0; dyi < 2; ++dyi)
#pragma unroll
            for (int dxi = 0; dxi < 2; ++dxi)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    b1[dyi][dxi][i] = ldsu(ba[dxi] + (unsigned)((i + 1 - dyi) * ROWB));
                    b2[dyi][dxi][i] = ldsu(ba[dxi] + (unsigned)((i + 1 - dyi) * ROWB + 64));
                }
    };
    auto load_w = [&](int slot, int sq) __attribute__((always_inline)) {
        wf1[slot] = ldsu(wa + (unsigned)(sq * 2048));
        wf2[slot] = ldsu(wa + (unsigned)(sq * 2048 + 1024));
    };
    load_b();
    load_w(0, tap_of(18).sq); load_w(1, tap_of(19).sq);      // step 0 is the halo plane: its first tap is 18 (ring slot = (t - T0) % 3)

    // wave-uniform march state
    int s = 0, c = 0;                                 // input plane step / cin group of the current micro-step
    unsigned long long out_pl = (unsigned long long)out_n + (unsigned long long)(long long)(2 * (zb - 2)) * PLANE_O;   // planes 2 (z - 1), 2 (z - 1) + 1 of step s = 0

    float mx = 0.f, mxt = 0.f;      // max |stored value|: all planes so far / the plane pair in flight (dropped with its stores when s < 2)
    // epilogue item e of a finished plane pair: e < 16: odd set (pz = 1), class e >> 2, row e & 3; else the old even_cur (pz = 0)
    auto finish = [&](auto ph_tag, auto e_tag, const __amdgpu_buffer_rsrc_t& rout, f32x4& keep) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_tag)::value, e = decltype(e_tag)::value;
        constexpr int pz = e < 16 ? 1 : 0, cls = (e >> 2) & 3, i = e & 3, py = cls >> 1, px = cls & 1;
        f32x4 o = acc_read(pz ? O[cls][i] : E[PH ^ 1][cls][i]);
        if (RELU) o = __builtin_elementwise_maximum(o, zero4);
        keep = o;
        asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mxt) : "v"(o[0]), "v"(o[1]));
        asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mxt) : "v"(o[2]), "v"(o[3]));
        const unsigned soff = (unsigned)pz * PLANE_O + (unsigned)(((2 * i + py) * OW + px) * a.ocs * 4);
        buf_store4(rout, keep, ob, soff);
    };

    // HALO: the slab's first input plane (zb - 1) only contributes its kz = 2 taps (to output plane 2 zb): 9 taps instead of 27
    auto micro = [&](auto ph_tag, auto first_tag, auto halo_tag) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value, HALO = decltype(halo_tag)::value;
        constexpr int T0 = HALO ? 18 : 0;
        // FIRST: the planes finished by the previous input plane leave under the first taps (zero-sized: stores dropped, s < 2)
        const bool prev_ok = s >= 2;
        const __amdgpu_buffer_rsrc_t rout = make_rsrc((const void*)(prev_ok ? out_pl : (unsigned long long)out_n), FIRST && prev_ok ? 2u * PLANE_O : 0u);
        f32x4 ost[2][3];
        static_for(std::make_integer_sequence<int, 27 - T0>{}, [&](auto t_tag) __attribute__((always_inline)) {
            constexpr int t = decltype(t_tag)::value + T0;
            constexpr Tap T = tap_of(t);
            // weight fragment of the next tap (wraps to tap 0 of the next micro-step: loaded after the barrier instead)
            if constexpr (t + 2 < 27) load_w((t + 2 - T0) % 3, tap_of(t + 2).sq);
            constexpr bool open = FIRST && T.opens && T.kz != 0;     // first tap of a class of the odd / even_next set in this plane
            // three MFMAs per row, term outermost: a dependent MFMA is four issue slots away
#pragma unroll
            for (int tm = 0; tm < 3; ++tm)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4& acc = T.kz == 0 ? E[PH][T.cls][i] : T.kz == 1 ? O[T.cls][i] : E[PH ^ 1][T.cls][i];
                    acc = mfma_bf16(tm == 2 ? wf2[(t - T0) % 3] : wf1[(t - T0) % 3], tm == 1 ? b2[T.dyi][T.dxi][i] : b1[T.dyi][T.dxi][i], (open && tm == 0) ? bias_l : acc);
                }
#ifndef PCC_TR2MB_INTERLEAVE
            __builtin_amdgcn_sched_barrier(0);
#endif
            // the tile of the NEXT micro-step (raw, in registers since the previous micro-step) replaces this one in LDS.  Every wave read
            // its 16 vectors before tap T0: the barrier behind that tap -- where the waves are still close together -- orders the
            // overwrite behind those reads; split + writes + the request for the tile after next run behind the last tap.  (Measured,
            // min of bench_one on one box: this placement 182 - 185 us, barrier and blocks all behind the last tap 187 - 190, the three
            // split blocks spread over taps T0 + 3 .. 5 220 - 228 -- their LDS writes sit between the weight-fragment reads of the
            // ring, which return in order -- against 202 - 206 for the version that split the 16 vectors per micro-step.)
            if constexpr (t == T0) __syncthreads();
            if constexpr (t == 26) {
                commit2(std::integral_constant<int, 0>{}); commit2(std::integral_constant<int, 2>{}); commit2(std::integral_constant<int, 4>{});
                fetch_next();
            }
            if constexpr (FIRST && !HALO) {
                if constexpr (t < 8) {
                    finish(ph_tag, std::integral_constant<int, 2 * t>{}, rout, ost[t & 1][0]);
                    finish(ph_tag, std::integral_constant<int, 2 * t + 1>{}, rout, ost[t & 1][1]);
                }
                if constexpr (t < 16) finish(ph_tag, std::integral_constant<int, 16 + t>{}, rout, ost[t & 1][2]);
                if constexpr (t >= 1 && t < 17) {     // store data registers stay untouched for one more tap (late read, see conv_wino.hip)
                    asm volatile("" ::"v"(ost[(t - 1) & 1][2]));
                    if constexpr (t < 9) { asm volatile("" ::"v"(ost[(t - 1) & 1][0])); asm volatile("" ::"v"(ost[(t - 1) & 1][1])); }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (FIRST && !HALO) { mx = fmaxf(mx, prev_ok ? mxt : 0.f); mxt = 0.f; }
        // the operand-form tile of the next micro-step is complete in LDS (this wave's ds_writes: lgkmcnt; the others': barrier)
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0); vmcnt / expcnt untouched (stores and the raw loads stay in flight)
        __syncthreads();
        // ---- advance (wave-uniform) and fetch the operands of the next micro-step
        if (++c == NG) { c = 0; ++s; out_pl += 2ull * PLANE_O; wa -= (unsigned)((NG - 1) * 27 * 2048); } else wa += 27u * 2048u;
        load_b();
        // first weight fragment of the next micro-step (buffer parity follows its first tap: 0, or 18 inside the halo plane)
        if (s == 0) { load_w(0, tap_of(18).sq); load_w(1, tap_of(19).sq); }
        else { load_w(0, tap_of(0).sq); load_w(1, tap_of(1).sq); }
    };
    // one input plane = NG micro-steps, unrolled (a run-time loop over the middle ones made the register allocator shuttle the
    // accumulators between AccVGPRs and VGPRs at the loop boundary)
    auto plane = [&](auto ph_tag, auto halo_tag) __attribute__((always_inline)) {
        micro(ph_tag, std::true_type{}, halo_tag);
#pragma unroll
        for (int k = 1; k < NG; ++k) micro(ph_tag, std::false_type{}, halo_tag);
    };
    plane(std::integral_constant<int, 0>{}, std::true_type{});          // step 0: the halo plane
#pragma nounroll
    for (int sp = 1; sp < nsteps; sp += 2) {
        plane(std::integral_constant<int, 1>{}, std::false_type{});
        if (sp + 1 < nsteps) plane(std::integral_constant<int, 0>{}, std::false_type{});
    }
    // ---- drain: the planes finished by the last input plane (s == nsteps here; its parity decides which even set is complete)
    {
        const __amdgpu_buffer_rsrc_t rout = make_rsrc((const void*)out_pl, nsteps >= 2 ? 2u * PLANE_O : 0u);
        f32x4 keep[32];
        if (nsteps & 1) {      // last plane had PH = 0: its even_cur is E[0] = "E[PH ^ 1]" of a PH = 1 epilogue
            static_for(std::make_integer_sequence<int, 32>{}, [&](auto e_tag) __attribute__((always_inline)) {
                finish(std::integral_constant<int, 1>{}, e_tag, rout, keep[decltype(e_tag)::value]); });
        } else {
            static_for(std::make_integer_sequence<int, 32>{}, [&](auto e_tag) __attribute__((always_inline)) {
                finish(std::integral_constant<int, 0>{}, e_tag, rout, keep[decltype(e_tag)::value]); });
        }
#pragma unroll
        for (int e = 0; e < 32; ++e) asm volatile("" ::"v"(keep[e]));
        mx = fmaxf(mx, nsteps >= 2 ? mxt : 0.f);
    }
    // ---- max |out| of block n for the fp16-split layer behind this one (conv_wino_f16s.hip): atomicMax of non-negative fp32 bit patterns
    if (a.amax_out != nullptr) pcc_amax_record(a.amax_out + (size_t)n * PCC_AMAX_SLOTS, mx, (int)blockIdx.x * 4 + wave);
}

}  // namespace pcctr2mb

using namespace pcctr2mb;

// ---- host: split image of the class-major weights.  w_tr2g: [cin group][27][cout tile][64 lanes][4 floats] (conv_tr2g order, fp32)
//      -> out: [cin group][27][cout tile][operand][64 lanes][8 bf16]; operand 0 = [Wh | Wm], 1 = [Wl | Wh]
static inline unsigned short bf16_rn_(float v) {
    unsigned b;
    memcpy(&b, &v, 4);
    if ((b & 0x7f800000u) == 0x7f800000u) return (unsigned short)(b >> 16);
    b += 0x7fffu + ((b >> 16) & 1u);
    return (unsigned short)(b >> 16);
}
static inline float bf16_f_(unsigned short h) {
    const unsigned b = (unsigned)h << 16;
    float v;
    memcpy(&v, &b, 4);
    return v;
}
size_t pcc_tr2m_bf16_packed_floats(int Cin, int Cout) { return (size_t)(Cin / 16) * 27 * (Cout / 16) * 2 * 64 * 4; }
void pcc_tr2m_bf16_pack(int Cin, int Cout, const float* w_tr2g, float* out) {
    unsigned short* o = reinterpret_cast<unsigned short*>(out);
    const size_t nfrag = (size_t)(Cin / 16) * 27 * (Cout / 16);
    for (size_t f = 0; f < nfrag; ++f)
        for (int lane = 0; lane < 64; ++lane) {
            unsigned short h[4], m[4], l[4];
            for (int c = 0; c < 4; ++c) {
                const float x = w_tr2g[(f * 64 + lane) * 4 + c];
                h[c] = bf16_rn_(x);
                const float r1 = x - bf16_f_(h[c]);
                m[c] = bf16_rn_(r1);
                l[c] = bf16_rn_(r1 - bf16_f_(m[c]));
            }
            unsigned short* a1 = o + ((f * 2 + 0) * 64 + lane) * 8;
            unsigned short* a2 = o + ((f * 2 + 1) * 64 + lane) * 8;
            for (int c = 0; c < 4; ++c) { a1[c] = h[c]; a1[4 + c] = m[c]; a2[c] = l[c]; a2[4 + c] = h[c]; }
        }
}

// the split kernel covers the 32 -> 16 layer (its 108 KB of split weights fit LDS beside the tile ring); shape-only rule
bool pcc_tr2m_bf16_covers(const pcc_conv_desc* d) { return pcc_tr2m_eligible(d) && d->Cin == 32 && d->Cout == 16; }

int pcc_conv_tr2m_bf16(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w_split, const float* bias, float* out,
                       pcc_conv_ext* ext, hipStream_t st) {
    PCC_REQUIRE(pcc_tr2m_bf16_covers(d), "pcc_conv_tr2m_bf16: shape not covered");
    Tr2mArgs a;
    a.in = in; a.w = w_split; a.bias = bias; a.out = out;
    a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W;
    a.nty = d->H / 16; a.ntx = d->W / 16; a.nct = d->Cout / 16;
    a.flags = d->flags;
    a.ocs = d->out_cstride ? d->out_cstride : d->Cout;
    a.oco = d->out_coffset;
    if (ext && ext->out_amax) { a.amax_out = ext->out_amax; ext->out_recorded = true; }
    const int base = d->N * (d->H / 16) * (d->W / 16) * (d->Cout / 16);
    int zs = 1;
    while (base * zs < ctx->num_cu && d->D % (zs * 2) == 0 && d->D / (zs * 2) >= 4) zs *= 2;
    a.zsplit = zs; a.zlen = d->D / zs;
    const int nwg = base * zs;
    const int lds = W_BASE + 2 * 27 * 2048;
    typedef void (*kern_t)(Tr2mArgs, int);
    const kern_t kern = (d->flags & PCC_CONV_RELU) ? (kern_t)conv_tr2m_bf16_kernel<2, true> : (kern_t)conv_tr2m_bf16_kernel<2, false>;
    { const int rc = pcc_enable_big_lds((const void*)kern, lds); if (rc != PCC_OK) return rc; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(NT), lds, st, a, nwg);
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}
