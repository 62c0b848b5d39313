// Conv3DTranspose, k = 3, stride 2, 32 -> 16, marching along z, on the f16 MFMA pipe from TWO fp16 pieces per fp32 operand (round 6).
//
// conv_tr2m_bf16.hip (round 4) with the operand form of conv_wino_f16s.hip: x = h + l, h = fp16_rn(s x), l = fp16_rn(s x - h) under an exact
// power-of-two pre-scale (weights: one su per layer, host; activations: one s per 64^3 block from the maximum the producing kernel
// recorded -- common.h, pcc_conv_ext), all four product terms in two v_mfma_f32_16x16x32_f16 per (row, tap):
//     acc += [Wh | Wh] . [dh | dl];   acc += [Wl | Wl] . [dh | dl]
// Against the three-piece bf16 form: 216 instead of 324 MFMAs per micro-step, ONE operand vector per (voxel, channel quad) -- 64 instead
// of 128 B-operand registers, 96 instead of 160 bytes per staged voxel (30 KB tile), 16 instead of 32 LDS reads per micro-step -- and the
// split of a staged item is 2 packed muls (the scale) + 2 cvt_pk + 4 fma_mix instead of 20 ops with DOT hazards.  The scale is undone
// in the epilogue (one packed mul pair per stored float4; ReLU commutes with it), the bias enters the accumulators scaled by su s.
// Weights (late round 6): LDS-resident, lane-contiguous [cin group][lane][27 taps][4 h | 4 l] fp16 + 8 B pad (440 B per lane), read by
// ds_read2_b64 with offset0 == offset1 -- the 8 bytes of a piece arrive twice in four consecutive registers, the duplicated operands
// [Wh | Wh] / [Wl | Wl] without a duplicated image (conv_wino_f16s.hip): 27.5 KB per cin group instead of 54, so the 64 -> 32 layer
// (NG = 4: 110 KB + the 30 KB tile) marches too.  Those reads are inline asm (the compiler inserts no waits for them); inside the tap
// loop they are the ONLY LDS operations, two taps ahead, so "at most 4 younger reads outstanding" is the wait in front of every tap.
// Everything else -- /root/reference/src/model_transforms.py:78 inside :126-137, parity decomposition, three accumulator sets, epilogue
// under the first taps of the next plane, compile-time plane parity, the recorded maximum of the output -- is conv_tr2m_bf16.hip's.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

#include "common.h"

namespace pcctr2mh {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 mfma_f16(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// One B-operand vector (4 input channels of a voxel): fp32 -> [dh c0..c3 | dl c0..c3] under the block's scale.  l = fp16(s x - h) comes out of
// v_fma_mix{lo,hi}_f16 (h x -1.0 + s x, the fp16 operand widened exactly, one rounding: the bits of cvt(s x - h), conv_wino_f16s.hip).
__device__ __forceinline__ u32x4 split_quad(const f32x4& v, const f32x2& s2) {
    f32x2 lo, hi;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(__builtin_shufflevector(v, v, 0, 1)), "v"(s2));
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(__builtin_shufflevector(v, v, 2, 3)), "v"(s2));
    const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, f16x2)), h23 = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, f16x2));
    unsigned l01, l23;
    asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(l01) : "v"(h01), "v"(lo[0]), "v"(lo[1]));
    asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(l23) : "v"(h23), "v"(hi[0]), "v"(hi[1]));
    return (u32x4){h01, h23, l01, l23};
}
template <int OFF>
__device__ __forceinline__ u32x4 lds_read2_dup(unsigned addr) {      // {8 bytes at addr + OFF} twice
    u32x4 v;
    asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%2" : "=v"(v) : "v"(addr), "n"(OFF / 8));
    return v;
}
template <int N>
__device__ __forceinline__ void lgkm_wait_w(u32x4& a, u32x4& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
__device__ __forceinline__ f32x4 mul4s(const f32x4& a, const f32x2& s) {
    f32x2 lo, hi;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(__builtin_shufflevector(a, a, 0, 1)), "v"(s));
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(__builtin_shufflevector(a, a, 2, 3)), "v"(s));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
// the block's pre-scale from its recorded max |x| (conv_wino_f16s.hip, f16s_scale_bits): here |operand| = |x| <= max, no transform in front
// of the split, so s max lands in [2^14, 2^15)
__device__ __forceinline__ unsigned tr2m_scale_bits(unsigned m, int lsu) {
    if (m == 0u) return 0x3f800000u;
    int se = 268 - (int)(m >> 23);
    const int lo = 7 - lsu > 1 ? 7 - lsu : 1, hi = 247 - lsu < 254 ? 247 - lsu : 254;
    se = se < lo ? lo : se > hi ? hi : se;
    return (unsigned)se << 23;
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
constexpr int VSB = 96;                                 // bytes per voxel of the operand-form tile: [dh | dl] x 4 quads + 32 pad (6 bank quads: brute-forced over the
                                                        // ds_read_b128 lane groups and the 8-lane groups of the staging ds_write_b128: no conflicts)
constexpr int TILE_BYTES = ITEMS * 64 * VSB;            // 30720: 320 voxel slots (289 used; the items past the tile write zeros into the rest)
constexpr int W_BASE = TILE_BYTES;                      // one tile, then the weights
constexpr int WL_LANE = 27 * 16 + 8;                    // bytes per lane and cin group: 27 taps x (4 h + 4 l fp16) + 8 pad (440 = 55 x 8: 32 lanes on 64 distinct banks)
constexpr int WG_BYTES = 64 * WL_LANE;                  // 28160 per (cin group, cout tile)
constexpr int ROWB = LXY * VSB;                         // bytes per tile row

struct Tr2mArgs {
    const float* in;
    const float* w;      // two-piece image: [cout tile][cin group][lane][27 taps, class-major (conv_tr2g order)][4 h | 4 l fp16] + 8 B pad per lane
    const float* wtail;  // [0] = su, the power of two the weight pieces were scaled by
    const unsigned* amax_in;      // per-block maxima of the input (PCC_AMAX_SLOTS partial maxima per block)
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
__global__ void __launch_bounds__(NT, 1) conv_tr2m_f16s_kernel(Tr2mArgs a, int nwg) {
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

    // ---- the weight pieces of this cout tile -> LDS (resident): NG x 27.5 KB, one linear copy
    {
        constexpr int WBYTES = NG * WG_BYTES, NCH = (WBYTES + 1023) / 1024;
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w + (size_t)ct * (WBYTES / 4), (unsigned)WBYTES);      // beyond the image: zeros
        for (int p = wave; p < NCH; p += 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(smem + W_BASE + p * 1024), 16, lane * 16, p * 1024, 0, 0);
    }
    // ---- scales (wave-uniform)
    const float su = a.wtail[0];
    const unsigned sbits = tr2m_scale_bits(pcc_amax_read(a.amax_in + (size_t)n * PCC_AMAX_SLOTS), (int)(__builtin_bit_cast(unsigned, su) >> 23) - 127);
    const float sv = __builtin_bit_cast(float, sbits);
    const float sprod = su * sv, sinv = 1.0f / sprod;          // powers of two inside 2^+-120: exact
    const f32x2 sc2 = {sv, sv}, inv2 = {sinv, sinv};
    // ---- tile staging: global -> registers (one micro-step ahead) -> split -> LDS.  Item it of thread tid = (voxel u, channel quad q) =
    //      ((it * 256 + tid) >> 2, tid & 3); its operand vector goes to u * 96 + q * 16.  OOB items read zeros.
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
        *reinterpret_cast<u32x4*>(smem + cw + (unsigned)(j * 64 * VSB)) = split_quad(stg[j], sc2);
        if constexpr (j1 != j) *reinterpret_cast<u32x4*>(smem + cw + (unsigned)(j1 * 64 * VSB)) = split_quad(stg[j1], sc2);
    };

    // ---- B operand addresses: lane (v, cq) reads voxel (row, lx = v + 1 - dx) of the tile, its channel quad cq; one base per x offset,
    //      rows / y offsets are immediates.  ba[dxi] points at tile row 4 * wave.
    unsigned ba[2];
#pragma unroll
    for (int dxi = 0; dxi < 2; ++dxi) ba[dxi] = (unsigned)((4 * wave * LXY + v + 1 - dxi) * VSB + cq * 16);
    unsigned wa = (unsigned)(unsigned long long)(lds_ptr)smem + (unsigned)(W_BASE + lane * WL_LANE);      // absolute; + cin group * 27.5 KB (per micro-step), + tap * 16 B (immediate)

    // ---- epilogue addressing: lane writes couts 4 cq .. 4 cq + 3 (of this cout tile) of output voxel (2 z + pz, 2 y + py, 2 x + px)
    const int OH = 2 * a.H, OW = 2 * a.W;
    const unsigned ob = (unsigned)((((2 * (Y0 + 4 * wave)) * OW + 2 * (X0 + v)) * a.ocs + a.oco + 16 * ct + 4 * cq) * 4);
    const unsigned PLANE_O = (unsigned)((size_t)OH * OW * a.ocs * 4);          // bytes per output plane (launcher: two planes < 2^31)
    float* out_n = a.out + (size_t)n * (2 * a.D) * OH * OW * a.ocs;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 bias_l = (a.flags & PCC_CONV_BIAS) ? *reinterpret_cast<const f32x4*>(a.bias + 16 * ct + 4 * cq) : zero4;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) bias_l[cc] *= sprod;          // enters accumulators that hold su s times the sums

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
    u32x4 b1[2][2][4];                                // B operands [dyi][dxi][row] of the current micro-step: [dh | dl]
    u32x4 wf1[3], wf2[3];                             // weight fragment ring, two taps ahead (a tap is 8 MFMAs = 128 cycles: one tap does not cover an LDS read): [Wh | Wh], [Wl | Wl]
    // the 16 operand vectors of the micro-step: 16 plain LDS reads
    auto load_b = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int dyi = 0; dyi < 2; ++dyi)
#pragma unroll
            for (int dxi = 0; dxi < 2; ++dxi)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    b1[dyi][dxi][i] = ldsu(ba[dxi] + (unsigned)((i + 1 - dyi) * ROWB));
    };
    auto load_w = [&](auto slot_tag, auto sq_tag) __attribute__((always_inline)) {
        constexpr int slot = decltype(slot_tag)::value, sq = decltype(sq_tag)::value;
        wf2[slot] = lds_read2_dup<sq * 16 + 8>(wa);
        wf1[slot] = lds_read2_dup<sq * 16>(wa);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    load_b();
    load_w(I0{}, std::integral_constant<int, tap_of(18).sq>{}); load_w(I1{}, std::integral_constant<int, tap_of(19).sq>{});      // step 0 is the halo plane: its first tap is 18 (ring slot = (t - T0) % 3)

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
        o = mul4s(o, inv2);          // un-scale (a power of two: commutes with the ReLU and with every rounding before it)
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
            if constexpr (t + 2 < 27) load_w(std::integral_constant<int, (t + 2 - T0) % 3>{}, std::integral_constant<int, tap_of(t + 2).sq>{});
            // this tap's fragments were requested two taps ago; younger LDS reads: the fragments of taps t + 1 and t + 2 (two each)
            lgkm_wait_w<(t + 2 < 27 ? 4 : t + 1 < 27 ? 2 : 0)>(wf1[(t - T0) % 3], wf2[(t - T0) % 3]);
            constexpr bool open = FIRST && T.opens && T.kz != 0;     // first tap of a class of the odd / even_next set in this plane
            // two MFMAs per row, term outermost: a dependent MFMA is four issue slots away
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4& acc = T.kz == 0 ? E[PH][T.cls][i] : T.kz == 1 ? O[T.cls][i] : E[PH ^ 1][T.cls][i];
                    acc = mfma_f16(tm == 1 ? wf2[(t - T0) % 3] : wf1[(t - T0) % 3], b1[T.dyi][T.dxi][i], (open && tm == 0) ? bias_l : acc);
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
        if (++c == NG) { c = 0; ++s; out_pl += 2ull * PLANE_O; wa -= (unsigned)((NG - 1) * WG_BYTES); } else wa += (unsigned)WG_BYTES;
        load_b();
        // first weight fragments of the next micro-step (buffer parity follows its first tap: 0, or 18 inside the halo plane)
        if (s == 0) { load_w(I0{}, std::integral_constant<int, tap_of(18).sq>{}); load_w(I1{}, std::integral_constant<int, tap_of(19).sq>{}); }
        else { load_w(I0{}, std::integral_constant<int, tap_of(0).sq>{}); load_w(I1{}, std::integral_constant<int, tap_of(1).sq>{}); }
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

}  // namespace pcctr2mh

using namespace pcctr2mh;

// ---- host: two-piece fp16 image of the class-major weights.  w_tr2g: [cin group][27][cout tile][64 lanes][4 floats] (conv_tr2g order, fp32)
//      -> out: [cin group][27][cout tile][operand][64 lanes][8 fp16]; operand 0 = [Wh | Wh], 1 = [Wl | Wl], all scaled by su (max |W| su in
//      [2^13, 2^14)); behind the fragments PCC_TR2M_F16S_TAIL floats, [0] = su
static inline unsigned short f16_bits_(float v) {
    const _Float16 h = (_Float16)v;
    unsigned short b;
    memcpy(&b, &h, 2);
    return b;
}
static inline float f16_value_(unsigned short b) {
    _Float16 h;
    memcpy(&h, &b, 2);
    return (float)h;
}
size_t pcc_tr2m_f16s_packed_floats(int Cin, int Cout) { return (size_t)(Cout / 16) * (Cin / 16) * (WG_BYTES / 4) + PCC_TR2M_F16S_TAIL; }
// w_tr2g: [cin group][27][cout tile][64 lanes][4 floats] -> out: [cout tile][cin group][lane][27][4 h | 4 l] (+ 8 B pad per lane), tail[0] = su
void pcc_tr2m_f16s_pack(int Cin, int Cout, const float* w_tr2g, float* out) {
    const int NGi = Cin / 16, NCT = Cout / 16;
    const size_t nfrag = (size_t)NGi * 27 * NCT;
    float wmax = 0.f;
    for (size_t i = 0; i < nfrag * 256; ++i) {
        const float v = fabsf(w_tr2g[i]);
        if (v > wmax && v <= 3.0e38f) wmax = v;
    }
    int e = 0;
    float su = 1.f;
    if (wmax > 0.f) {
        frexpf(wmax, &e);
        int se = 14 - e;
        se = se < -100 ? -100 : se > 100 ? 100 : se;
        su = ldexpf(1.f, se);
    }
    memset(out, 0, pcc_tr2m_f16s_packed_floats(Cin, Cout) * sizeof(float));
    unsigned short* o = reinterpret_cast<unsigned short*>(out);
    for (int g = 0; g < NGi; ++g)
        for (int sq = 0; sq < 27; ++sq)
            for (int ct = 0; ct < NCT; ++ct)
                for (int lane = 0; lane < 64; ++lane) {
                    unsigned short* d = o + ((size_t)(ct * NGi + g) * WG_BYTES + (size_t)lane * WL_LANE + (size_t)sq * 16) / 2;
                    for (int c = 0; c < 4; ++c) {
                        const float x = w_tr2g[((((size_t)g * 27 + sq) * NCT + ct) * 64 + lane) * 4 + c] * su;          // exact
                        d[c] = f16_bits_(x);
                        d[4 + c] = f16_bits_(x - f16_value_(d[c]));                                                    // the difference is exact
                    }
                }
    out[(size_t)NCT * NGi * (WG_BYTES / 4)] = su;
}

// 32 -> 16 and 64 -> 32 on grids of 16-multiples (the layers conv_tr2m.hip marches); shape-only rule
bool pcc_tr2m_f16s_covers(const pcc_conv_desc* d) { return pcc_tr2m_eligible(d); }

int pcc_conv_tr2m_f16s(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w_f16s, const float* bias, float* out,
                       pcc_conv_ext* ext, hipStream_t st) {
    PCC_REQUIRE(pcc_tr2m_f16s_covers(d), "pcc_conv_tr2m_f16s: shape not covered");
    const int NGi = d->Cin / 16;
    Tr2mArgs a;
    a.in = in; a.w = w_f16s; a.bias = bias; a.out = out;
    a.wtail = w_f16s + (size_t)(d->Cout / 16) * NGi * (WG_BYTES / 4);
    a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W;
    a.nty = d->H / 16; a.ntx = d->W / 16; a.nct = d->Cout / 16;
    a.flags = d->flags;
    a.ocs = d->out_cstride ? d->out_cstride : d->Cout;
    a.oco = d->out_coffset;
    if (ext) ext->out_recorded = false;
    if (ext && ext->out_amax) { a.amax_out = ext->out_amax; ext->out_recorded = true; }
    if (ext && ext->in_amax) a.amax_in = ext->in_amax;
    else {
        unsigned* am = nullptr;
        { const int rc = pcc_ctx_amax(ctx, d->N * PCC_AMAX_SLOTS, &am); if (rc != PCC_OK) return rc; }
        { const int rc = pcc_block_amax(ctx, in, d->N, (size_t)d->D * d->H * d->W * d->Cin, am, st); if (rc != PCC_OK) return rc; }
        a.amax_in = am;
    }
    const int base = d->N * (d->H / 16) * (d->W / 16) * (d->Cout / 16);
    int zs = 1;
    while (base * zs < ctx->num_cu && d->D % (zs * 2) == 0 && d->D / (zs * 2) >= 4) zs *= 2;
    a.zsplit = zs; a.zlen = d->D / zs;
    const int nwg = base * zs;
    const int lds = W_BASE + ((NGi * WG_BYTES + 1023) / 1024) * 1024;
    typedef void (*kern_t)(Tr2mArgs, int);
    static const kern_t kerns[4] = {conv_tr2m_f16s_kernel<2, false>, conv_tr2m_f16s_kernel<2, true>, conv_tr2m_f16s_kernel<4, false>, conv_tr2m_f16s_kernel<4, true>};
    const kern_t kern = kerns[(NGi == 4 ? 2 : 0) + ((d->flags & PCC_CONV_RELU) ? 1 : 0)];
    { const int rc = pcc_enable_big_lds((const void*)kern, lds); if (rc != PCC_OK) return rc; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(NT), lds, st, a, nwg);
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}
