// Winograd F(2x2, 3x3) in (x, y) + direct z taps on the bf16 MFMA pipe with fp32-EQUIVALENT operands (round 4).
//
// conv_wino.hip runs the k3 stride-1 layers of /root/reference/src/model_transforms.py:62-81 (AnalysisBlock / SynthesisBlock) on
// v_mfma_f32_16x16x4_f32 and sits on that pipe's issue rate (157 TFLOP/s; 0.60-0.71 of it executed in round 3, 76 % of the
// step).  gfx950's bf16 MFMA is 16x faster.  Here every fp32 operand is split EXACTLY into three bf16 pieces
//     x = h + m + l,   h = bf16_rn(x),  m = bf16_rn(x - h),  l = bf16_rn(x - h - m)        (8 + 8 + 8 significand bits)
// and a product keeps the six terms down to 2^-16 of it (hh, hm, mh, hl, mm, lh; dropped: ml, lm ~2^-24, ll ~2^-32 -- below
// the rounding of the fp32 kernel's own operands).  Two terms are stacked along K of one v_mfma_f32_16x16x32_bf16
// (K = 2 terms x 16 input channels), products are exact in fp32 and accumulate in fp32, in a fixed order:
//     acc += [Uh | Um] . [Vh | Vm]        (hh + mm)
//     acc += [Uh | Um] . [Vl | Vh]        (hl + mh)
//     acc += [Ul | Uh] . [Vh | Vm]        (lh + hm)
// = 3 MFMAs of 16 cycles per (z tap, Winograd point) instead of 4 of 32: two A operands (U: split once on the host, 32 B per
// lane in LDS) and two B operands (V: 8 VGPRs per point; Vh is written twice) -- no operand needs a register copy.
// V = B^T d B is computed in fp32 exactly as in conv_wino.hip and split in registers:
//     H = v_cvt_pk_bf16_f32(a, b);  a -= H.lo, b -= H.hi  (v_dot2c_f32_bf16 with the constants {-1, 0} / {0, -1}: the residual
//     of a rounding is exactly representable, so the subtraction is exact);  M = cvt(a, b);  a -= M.lo ...;  L = cvt(a, b)
// -- 7 VALU ops per value pair (measured beside bf16 MFMAs, tools/ubench/mfma_bf16_valu.hip: 47 cycles per pair against 65 for
// shift / mask / v_sub).  Accuracy: the split adds nothing measurable to the fp32 Winograd kernel's error (numpy model and
// tests/test_conv_gpu.py at the bench geometry); results are bit-deterministic and independent of the launch geometry.
//
// Everything else follows conv_wino.hip: workgroup = 4 waves = 16 x 16 (x, y) outputs marching along z, ring of 3 input planes
// in LDS filled by buffer_load ... lds, lane = (tile, cin quad) transforms its own 4 x 4 patch, three output planes in flight in
// 192 AccVGPRs, A^T . A lane-local.  What changed with the 2.7x shorter matrix work:
//   * rows run py-major ((py, dz) = slot q: py = q / 3, dz = 2 - q % 3), so the pieces of V row r are dead after slot 3r + 2
//     and the next plane's row r is built right behind it: the split work (56 ops per row) is spread over the step instead of
//     crowding its end, and only 3 of the 4 patch rows are ever live (48 registers instead of 64 -- the V pieces need 128);
//   * the summation order per output element is the one of conv_wino.hip per plane (dz = 2, 1, 0 of consecutive input planes),
//     with three MFMAs per row instead of four.
#include <cstdlib>
#include <cstring>
#include "wino_common.h"

namespace pccwino {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int UB_ROW_BYTES = 2048;                       // per (dz, point): A1 = [Uh | Um] (64 lanes x 16 B), A2 = [Ul | Uh]
constexpr int UB_BYTES = 48 * UB_ROW_BYTES;              // 98304
constexpr int LDS_BYTES_B = U_BASE + UB_BYTES;           // 162816 <= 160 KB

__device__ __forceinline__ f32x4 mfma_bf16(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// Two Winograd points (2 x 4 input channels of one tile): fp32 -> B1 = [Vh | Vm], B2 = [Vl | Vh] each.  ONE asm block, because
// v_dot2c_f32_bf16 is a DOT instruction: a different VALU op that reads its result needs 3 wait states behind it
// (GCNHazardRecognizer: DotWriteDifferentVALURead) and the hazard recogniser cannot see into inline asm.  Inside the block every
// reader sits >= 3 instructions behind its writer; K0 / K1 = the bf16 pairs {-1, 0} / {0, -1}: x -= lo(h) / hi(h), exactly.
__device__ __forceinline__ void split_points2(u32x4& p_b1, u32x4& p_b2, u32x4& q_b1, u32x4& q_b2, const f32x4& pv, const f32x4& qv) {
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
// B^T along x on one patch row (4 voxels x 4 channels), in place
__device__ __forceinline__ void transform_x_row(f32x4 (&P)[4]) {
    const f32x4 d0 = P[0], d1 = P[1], d2 = P[2], d3 = P[3];
    P[0] = sub4(d0, d2); P[1] = add4(d1, d2); P[2] = sub4(d2, d1); P[3] = sub4(d1, d3);
}

// Long-lived per-lane constants (addresses) are PARKED in AccVGPRs and read back right before their use: the 128 registers of
// the V pieces leave the arch VGPRs no room for them (the register allocator otherwise spills into scratch).
__device__ __forceinline__ unsigned park(unsigned v) {
    unsigned a;
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v));
    return a;
}
__device__ __forceinline__ unsigned unpark(unsigned a) {
    unsigned v;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}

// compile-time loop: the slot index must be a constant expression (row masks, U offsets and accumulator slots follow from it)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

enum { MB_ALL = 0, MB_S0 = 1, MB_S1 = 2, MB_S1O = 3, MB_FIN = 4 };
__host__ __device__ constexpr bool mb_row_active(int mode, int dz) {
    return mode == MB_ALL || (mode == MB_S0 && dz == 0) || ((mode == MB_S1 || mode == MB_S1O) && dz <= 1);
}
__host__ __device__ constexpr int mb_next_mode(int mode) { return mode == MB_S0 ? MB_S1 : MB_ALL; }
// first active slot q' > q of this step, or 12 + the first active slot of the next step
__host__ __device__ constexpr int mb_next_slot(int mode, int q) {
    for (int n = q + 1; n < 12; ++n)
        if (mb_row_active(mode, 2 - n % 3)) return n;
    for (int n = 0; n < 12; ++n)
        if (mb_row_active(mb_next_mode(mode), 2 - n % 3)) return 12 + n;
    return 12;
}
__host__ __device__ constexpr int mb_first_slot(int mode) {
    for (int n = 0; n < 12; ++n)
        if (mb_row_active(mode, 2 - n % 3)) return n;
    return 0;
}
__host__ __device__ constexpr unsigned ub_row_off(int q) { return (unsigned)((((2 - q % 3) * 4 + q / 3) * 4) * UB_ROW_BYTES); }   // (dz, py) row, px = 0

template <bool RELU, bool CLIP>
__global__ void __launch_bounds__(NT, 1) conv16_wino_bf16_kernel(WinoArgs a, int nwg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = lane & 15, g = lane >> 4;
    auto ldsr = [&](unsigned off) -> f32x4 { return *reinterpret_cast<const f32x4*>(smem + off); };
    auto ldsu = [&](unsigned off) -> u32x4 { return *reinterpret_cast<const u32x4*>(smem + off); };
    typedef __attribute__((address_space(3))) void* lds_ptr;

    int wg = xcd_remap(blockIdx.x, nwg);
    const int cog = wg % a.nco; wg /= a.nco;
    const int tx_ = wg % a.ntx; wg /= a.ntx;
    const int ty_ = wg % a.nty; wg /= a.nty;
    const int zs = wg % a.zsplit;
    const int n = wg / a.zsplit;
    const int X0 = tx_ * 16, Y0 = ty_ * 16, zb = zs * a.zlen;
    const int nsteps = a.zlen + 2;             // input planes zb-1 .. zb+zlen
    const size_t HW = (size_t)a.H * a.W;
    const unsigned HWI = (unsigned)(HW * a.ics * 4), HWR = (unsigned)(HW * a.rcs * 4), HWO = (unsigned)(HW * a.ocs * 4);
    const float* in_n = a.in + (size_t)n * a.D * HW * a.ics + a.ico;

    // ---- U -> LDS: 96 KB straight global -> LDS (24 pieces of 1 KB per wave)
    {
        const __amdgpu_buffer_rsrc_t ru = make_rsrc(a.u + (size_t)cog * (UB_BYTES / 4), (unsigned)UB_BYTES);
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            const int chunk = wave * 24 + k;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr)(smem + U_BASE + chunk * 1024), 16, (int)(lane * 16), chunk * 1024, 0, 0);
        }
    }

    // ---- plane staging (as conv16_wino_kernel): lane L of chunk c fetches whatever belongs into LDS slot 64c + L
    unsigned rel[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int slot = (wave * 5 + it) * 64 + lane;
        const int v = slot >> 2, c4 = (slot & 3) ^ ((v >> 1) & 3);
        const int yrow = v / 18, r = v - yrow * 18, par = r >= 9 ? 1 : 0, col = r - 9 * par, xi = 2 * col + par;
        const int y = Y0 - 1 + yrow, x = X0 - 1 + xi;
        const bool ok = v < PLANE_VOX && y >= 0 && y < a.H && x >= 0 && x < a.W;
        rel[it] = ok ? (unsigned)(((y * a.W + x) * a.ics + c4 * 4) * 4) : kOOB;
    }
    auto stage_plane = [&](unsigned plane_off, int z) __attribute__((always_inline)) {
        const bool ok = (unsigned)z < (unsigned)a.D;
        const __amdgpu_buffer_rsrc_t rp = make_rsrc(in_n + (ok ? (size_t)z * HW * a.ics : 0), ok ? HWI : 0u);
#pragma unroll
        for (int it = 0; it < ITEMS; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lds_ptr)(smem + plane_off + (wave * 5 + it) * 1024), 16, (int)rel[it], 0, 0, 0);
    };
    unsigned long long in_pl = (unsigned long long)in_n + (unsigned long long)(long long)(zb + 1) * HWI;   // plane s+2 of step s = 0

    // ---- per-lane patch read addresses (ring slot 0), tile of this lane
    const int wx = wave & 1, wy = wave >> 1;
    const int TX = 4 * wx + (t & 3), TY = 4 * wy + (t >> 2);
    unsigned ra[16];
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const int v = 36 * TY + 18 * dy + 9 * (dx & 1) + TX + (dx >> 1);
            ra[dy * 4 + dx] = (unsigned)(v * 64 + ((g ^ ((v >> 1) & 3)) << 4));
        }
    const unsigned ua = (unsigned)(U_BASE + lane * 16);

    // ---- epilogue addressing: lane writes couts 4g..4g+3 of the 2x2 voxels of its tile
    const int ox0 = X0 + 2 * TX, oy0 = Y0 + 2 * TY;
    unsigned ovo[4], rvo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned vox = (unsigned)((oy0 + (q >> 1)) * a.W + ox0 + (q & 1));
        ovo[q] = (vox * (unsigned)a.ocs + (unsigned)a.oco + 16u * cog + 4u * g) * 4u;
        rvo[q] = (vox * (unsigned)a.rcs + 16u * cog + 4u * g) * 4u;
    }
    const bool has_res = (a.flags & PCC_CONV_ADD) != 0;
    const float* res_n = has_res ? a.res + (size_t)n * a.D * HW * a.rcs : a.in;
    float* out_n = a.out + (size_t)n * a.D * HW * a.ocs;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bias4 = (a.flags & PCC_CONV_BIAS) ? *reinterpret_cast<const f32x4*>(a.bias + 16 * cog + g * 4) : zero4;

    // head / tail planes as in conv16_wino_kernel (wave-uniform, decided outside the MFMA stream)
    const bool first_zero = zb == 0, last_zero = zb + a.zlen == a.D;
    const int s0 = first_zero ? 1 : 0;
    stage_plane((unsigned)s0 * PLANE_BYTES, zb - 1 + s0);
    stage_plane((unsigned)(s0 + 1) * PLANE_BYTES, zb + s0);
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): U and both planes have landed
    __syncthreads();

    u32x4 B1[16], B2[16];       // pieces of B^T d B of the current input plane: [Vh | Vm], [Vl | Vh]
    f32x4 P0[4], P1[4], P2[4], P3[4];      // x-transformed patch rows of the NEXT plane (at most three are live)
    u32x4 A1[4], A2[4];         // U fragments of the row in flight: [Uh | Um], [Ul | Uh] per point px
    f32x4 acc[3][16];           // three output planes in flight
    f32x4 S[2][2];
    f32x4 resv[4], ost[4];
    // V row r of the plane whose patch rows are in P*: y-transform + split
    auto vrow = [&](auto r_tag) __attribute__((always_inline)) {
        constexpr int r = decltype(r_tag)::value;
        f32x4 y[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) y[x] = r == 0 ? sub4(P0[x], P2[x]) : r == 1 ? add4(P1[x], P2[x]) : r == 2 ? sub4(P2[x], P1[x]) : sub4(P1[x], P3[x]);
        split_points2(B1[r * 4 + 0], B2[r * 4 + 0], B1[r * 4 + 1], B2[r * 4 + 1], y[0], y[1]);
        split_points2(B1[r * 4 + 2], B2[r * 4 + 2], B1[r * 4 + 3], B2[r * 4 + 3], y[2], y[3]);
    };
    auto load_prow = [&](f32x4 (&P)[4], int dy, unsigned slot_off) __attribute__((always_inline)) {
#pragma unroll
        for (int x = 0; x < 4; ++x) P[x] = ldsr(ra[dy * 4 + x] + slot_off);
    };
    using R0 = std::integral_constant<int, 0>;
    using R1 = std::integral_constant<int, 1>;
    using R2 = std::integral_constant<int, 2>;
    using R3 = std::integral_constant<int, 3>;
    {
        const unsigned so = (unsigned)s0 * PLANE_BYTES;
        load_prow(P0, 0, so); load_prow(P1, 1, so); load_prow(P2, 2, so); load_prow(P3, 3, so);
        transform_x_row(P0); transform_x_row(P1); transform_x_row(P2); transform_x_row(P3);
        vrow(R0{}); vrow(R1{}); vrow(R2{});        // row 3 follows in slot 0 of the first step
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][i] = zero4; acc[1][i] = zero4; acc[2][i] = zero4; }
    {
        // U fragments of the first active row of the first step that runs
        const unsigned fo = first_zero ? ub_row_off(mb_first_slot(MB_S1O)) : ub_row_off(mb_first_slot(MB_S0));
#pragma unroll
        for (int px = 0; px < 4; ++px) { A1[px] = ldsu(ua + fo + (unsigned)(px * UB_ROW_BYTES)); A2[px] = ldsu(ua + fo + (unsigned)(px * UB_ROW_BYTES + 1024)); }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { resv[q] = zero4; ost[q] = zero4; }

    // park the per-lane address constants (see park())
    unsigned rel_p[ITEMS], ra_p[16], ovo_p[4], rvo_p[4];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) rel_p[i] = park(rel[i]);
#pragma unroll
    for (int i = 0; i < 16; ++i) ra_p[i] = park(ra[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) { ovo_p[i] = park(ovo[i]); rvo_p[i] = park(rvo[i]); }
    auto load_prow_p = [&](f32x4 (&P)[4], int dy, unsigned slot_off) __attribute__((always_inline)) {
#pragma unroll
        for (int x = 0; x < 4; ++x) P[x] = ldsr(unpark(ra_p[dy * 4 + x]) + slot_off);
    };

    unsigned long long res_pl = (unsigned long long)res_n + (unsigned long long)(long long)(zb - 2 + s0) * HWR;     // plane zo of step s0
    unsigned long long out_pl = (unsigned long long)out_n + (unsigned long long)(long long)(zb - 2 + s0) * HWO;
    in_pl += (unsigned long long)s0 * HWI;

    // one input plane: s = step index (input plane z = zb-1+s), PH = s mod 3; MODE as in conv16_wino_kernel
    auto step = [&](auto ph_tag, int s, auto mode_tag) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_tag)::value;
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool FIN = MODE == MB_FIN;
        constexpr unsigned slotN = (unsigned)((PH + 1) % 3) * PLANE_BYTES;   // plane s+1 (read)
        constexpr unsigned slotW = (unsigned)((PH + 2) % 3) * PLANE_BYTES;   // plane s+2 (written)
        constexpr int AF = PH;                                               // acc slot of the plane finished by dz = 2
        const bool zo_ok = s >= 2;                                           // the finished plane zo = zb - 2 + s exists
        const __amdgpu_buffer_rsrc_t rout = make_rsrc((const void*)(zo_ok ? out_pl : (unsigned long long)out_n), zo_ok ? HWO : 0u);
        // residual of the plane the NEXT step finishes (a full step of lead)
        const bool zn_ok = s + 1 >= 2 && s + 1 < nsteps && has_res;
        const __amdgpu_buffer_rsrc_t rres = make_rsrc((const void*)(zn_ok ? res_pl + HWR : (unsigned long long)res_n), zn_ok ? HWR : 0u);
        res_pl += HWR; out_pl += HWO;
        static_for<0, 12>([&](auto q_tag) __attribute__((always_inline)) {
            constexpr int q = decltype(q_tag)::value;
            constexpr int py = q / 3, dz = 2 - q % 3;
            constexpr int as = (PH + 2 - dz) % 3;
            constexpr bool active = mb_row_active(MODE, dz);
            if constexpr (active) {
                constexpr int qn = mb_next_slot(MODE, q) % 12;
                const unsigned un = ua + ub_row_off(qn);
                constexpr bool opens = dz == 0 || (MODE == MB_S1O && dz == 1);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // 6 MFMAs of two points, alternating between them (a dependent MFMA is two issue slots away)
#pragma unroll
                    for (int tm = 0; tm < 3; ++tm)
#pragma unroll
                        for (int px = 2 * h; px < 2 * h + 2; ++px) {
                            const int i = py * 4 + px;
                            // a new output plane starts from 0, except point (1,1), which enters all four outputs with weight +1 and carries the bias
                            const f32x4 c = (opens && tm == 0) ? ((py == 1 && px == 1) ? bias4 : zero4) : acc[as][i];
                            acc[as][i] = mfma_bf16(tm == 2 ? A2[px] : A1[px], tm == 1 ? B2[i] : B1[i], c);
                        }
                    // the fragments of this half are dead: the same registers receive the next row's
#pragma unroll
                    for (int px = 2 * h; px < 2 * h + 2; ++px) {
                        A1[px] = ldsu(un + (unsigned)(px * UB_ROW_BYTES));
                        A2[px] = ldsu(un + (unsigned)(px * UB_ROW_BYTES + 1024));
                    }
                }
            }
            // ---- everything else.  V row r of the next plane is built right behind the last use of the current one (slot 3r + 2).
            if (!FIN) {
                if (q == 0) vrow(R3{});                                   // row 3 of THIS plane (patch rows of the previous step)
                else if (q == 1) {
                    const bool ok = (unsigned)(zb + 1 + s) < (unsigned)a.D;
                    const __amdgpu_buffer_rsrc_t rp = make_rsrc((const void*)(ok ? in_pl : (unsigned long long)in_n), ok ? HWI : 0u);
                    in_pl += HWI;
#pragma unroll
                    for (int it = 0; it < ITEMS; ++it)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lds_ptr)(smem + slotW + (wave * 5 + it) * 1024), 16, (int)unpark(rel_p[it]), 0, 0, 0);
                    load_prow_p(P0, 0, slotN); load_prow_p(P2, 2, slotN);
                } else if (q == 2) { transform_x_row(P0); transform_x_row(P2); }
                else if (q == 3) { vrow(R0{}); load_prow_p(P1, 1, slotN); }
                else if (q == 4) transform_x_row(P1);
                else if (q == 6) vrow(R1{});
                else if (q == 9) { vrow(R2{}); load_prow_p(P3, 3, slotN); }
                else if (q == 10) transform_x_row(P3);
            }
            if (q == 2 || q == 5 || q == 8 || q == 11) {
                // A^T along x on row r of the finished plane (its dz = 2 MFMAs ran in slot 3r), accumulate A^T along y
                const int r = q / 3;
                const f32x4 m0 = acc_read(acc[AF][r * 4 + 0]), m1 = acc_read(acc[AF][r * 4 + 1]), m2 = acc_read(acc[AF][r * 4 + 2]), m3 = acc_read(acc[AF][r * 4 + 3]);
                const f32x4 r0 = add4(add4(m0, m1), m2), r1 = sub4(sub4(m1, m2), m3);
                if (r == 0) { S[0][0] = r0; S[0][1] = r1; }
                else if (r == 1) { S[0][0] = add4(S[0][0], r0); S[0][1] = add4(S[0][1], r1); S[1][0] = r0; S[1][1] = r1; }
                else if (r == 2) { S[0][0] = add4(S[0][0], r0); S[0][1] = add4(S[0][1], r1); S[1][0] = sub4(S[1][0], r0); S[1][1] = sub4(S[1][1], r1); }
                else { S[1][0] = sub4(S[1][0], r0); S[1][1] = sub4(S[1][1], r1); }
            }
            if (q == 8 || q == 11) {
                // epilogue of output row oy (complete after reduction row 2 resp. 3): ReLU, residual, clip, float4 stores; then the
                // residual of the next plane into the registers just consumed
                const int oy = q == 8 ? 0 : 1;
#pragma unroll
                for (int v = 2 * oy; v < 2 * oy + 2; ++v) {
                    f32x4 o = S[v >> 1][v & 1];
                    if (RELU) o = __builtin_elementwise_maximum(o, zero4);
                    o = add4(o, resv[v]);     // zeros without PCC_CONV_ADD (zero-sized buffer)
                    if (CLIP) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] = fminf(fmaxf(o[c], 0.f), 1.f);
                    }
                    ost[v] = o;
                }
#pragma unroll
                for (int v = 2 * oy; v < 2 * oy + 2; ++v) buf_store4(rout, ost[v], unpark(ovo_p[v]), 0);
#pragma unroll
                for (int v = 2 * oy; v < 2 * oy + 2; ++v) resv[v] = buf_load4(rres, unpark(rvo_p[v]), 0);
            }
            // gfx950: a buffer_store_dwordx4 reads its data registers late (conv16_wino_kernel): keep them unwritten for one more slot
            if (q == 9) { asm volatile("" ::"v"(ost[0])); asm volatile("" ::"v"(ost[1])); }
            if (q == 0) { asm volatile("" ::"v"(ost[2])); asm volatile("" ::"v"(ost[3])); }
            __builtin_amdgcn_sched_barrier(0);
        });
        // the LDS-direct loads of plane s+2 (slot 1) must have landed before the barrier publishes them; younger: 4 stores and 4
        // residual loads (slots 8, 11)
        if (!FIN) {
            __builtin_amdgcn_s_waitcnt(0x0F78);      // vmcnt(8) expcnt(7) lgkmcnt(15)
            __syncthreads();
        }
    };
    using P0t = std::integral_constant<int, 0>;
    using P1t = std::integral_constant<int, 1>;
    using P2t = std::integral_constant<int, 2>;
    using MAll = std::integral_constant<int, MB_ALL>;
    using MFin = std::integral_constant<int, MB_FIN>;

    if (first_zero) step(P1t{}, 1, std::integral_constant<int, MB_S1O>{});
    else {
        step(P0t{}, 0, std::integral_constant<int, MB_S0>{});
        step(P1t{}, 1, std::integral_constant<int, MB_S1>{});
    }
    const int nloop = nsteps - (last_zero ? 1 : 0);
    for (int s = 2; s < nloop; s += 3) {
        step(P2t{}, s, MAll{});
        if (s + 1 < nloop) step(P0t{}, s + 1, MAll{});
        if (s + 2 < nloop) step(P1t{}, s + 2, MAll{});
    }
    if (last_zero) {
        const int sl = nsteps - 1, ph = sl % 3;
        if (ph == 0) step(P0t{}, sl, MFin{});
        else if (ph == 1) step(P1t{}, sl, MFin{});
        else step(P2t{}, sl, MFin{});
    }
}

}  // namespace pccwino

using namespace pccwino;

// ---- host: split-bf16 image of the Winograd-transformed weights.  Per (cin group, cout group): [dz][py][px][operand][lane][8 bf16]
//      operand 0 = [Uh c0..c3 | Um c0..c3], operand 1 = [Ul | Uh];  cin = 16 cig + 4 (lane >> 4) + c, cout = 16 cog + (lane & 15)
static inline unsigned short bf16_rn_bits(float v) {
    unsigned b;
    memcpy(&b, &v, 4);
    if ((b & 0x7f800000u) == 0x7f800000u) return (unsigned short)(b >> 16);      // inf / nan: truncate
    b += 0x7fffu + ((b >> 16) & 1u);
    return (unsigned short)(b >> 16);
}
static inline float bf16_bits_to_float(unsigned short h) {
    const unsigned b = (unsigned)h << 16;
    float v;
    memcpy(&v, &b, 4);
    return v;
}
// u_f32: the fp32 Winograd image of conv_wino.hip ([cin group][cout group][48][64 lanes][4]) -> out: PCC_WINO_UB_FLOATS per pair
void pcc_wino_bf16_pack(int ngroups, const float* u_f32, float* out) {
    unsigned short* o = reinterpret_cast<unsigned short*>(out);
    for (int pair = 0; pair < ngroups * ngroups; ++pair)
        for (int row = 0; row < 48; ++row)
            for (int lane = 0; lane < 64; ++lane) {
                unsigned short h[4], m[4], l[4];
                for (int c = 0; c < 4; ++c) {
                    const float x = u_f32[(((size_t)pair * 48 + row) * 64 + lane) * 4 + c];
                    h[c] = bf16_rn_bits(x);
                    const float r1 = x - bf16_bits_to_float(h[c]);          // exact
                    m[c] = bf16_rn_bits(r1);
                    const float r2 = r1 - bf16_bits_to_float(m[c]);         // exact
                    l[c] = bf16_rn_bits(r2);
                }
                unsigned short* a1 = o + ((((size_t)pair * 48 + row) * 2 + 0) * 64 + lane) * 8;
                unsigned short* a2 = o + ((((size_t)pair * 48 + row) * 2 + 1) * 64 + lane) * 8;
                for (int c = 0; c < 4; ++c) { a1[c] = h[c]; a1[4 + c] = m[c]; a2[c] = l[c]; a2[4 + c] = h[c]; }
            }
}

bool pcc_wino_bf16_covers(const pcc_conv_desc* d) { return d->Cin == 16 && d->Cout == 16; }

int pcc_conv_wino_bf16(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* ub_packed, const float* bias,
                       const float* residual, float* out, hipStream_t st) {
    PCC_REQUIRE(pcc_wino_eligible(d) && d->Cin == 16, "pcc_conv_wino_bf16: shape not covered");
    WinoArgs a;
    a.in = in; a.bias = bias; a.res = residual; a.out = out;
    a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W;
    a.nty = d->H / 16; a.ntx = d->W / 16;
    a.ocs = d->out_cstride ? d->out_cstride : d->Cout;
    a.oco = d->out_coffset;
    a.nco = 1; a.ics = d->Cin; a.rcs = d->Cout; a.ico = 0; a.ncig = 1;
    a.u = ub_packed; a.flags = d->flags;
    const int base = d->N * a.nty * a.ntx;
    int zs = 1;
    while (base * zs < ctx->num_cu && d->D % (zs * 2) == 0 && d->D / (zs * 2) >= 8) zs *= 2;
    a.zsplit = zs; a.zlen = d->D / zs;
    const int nwg = base * zs;
    typedef void (*kern_t)(WinoArgs, int);
    static const kern_t kerns[4] = {conv16_wino_bf16_kernel<false, false>, conv16_wino_bf16_kernel<true, false>,
                                    conv16_wino_bf16_kernel<false, true>, conv16_wino_bf16_kernel<true, true>};
    const kern_t kern = kerns[((d->flags & PCC_CONV_RELU) ? 1 : 0) + ((d->flags & PCC_CONV_CLIP01) ? 2 : 0)];
    { const int rc = pcc_enable_big_lds((const void*)kern, LDS_BYTES_B); if (rc != PCC_OK) return rc; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(NT), LDS_BYTES_B, st, a, nwg);
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}
