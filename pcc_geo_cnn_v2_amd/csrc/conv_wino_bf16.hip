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

// timing probes (tools/build_variant.sh): 1 no step barrier / vmcnt wait, 2 no plane loads, 4 no residual loads / stores, 8 no DOT
// blocks, 16 no MFMAs, 32 no cvts, 64 no output side (AccVGPR reads, A^T, epilogue), 128 no input transforms,
// 256 no U refills, 512 no patch reads, 1024 AccVGPR reads in the K blocks
#ifndef PCC_WB_PROBE
#define PCC_WB_PROBE 0
#endif

namespace pccwino {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int UB_ROW_BYTES = 2048;                       // per (dz, point): A1 = [Uh | Um] (64 lanes x 16 B), A2 = [Ul | Uh]
constexpr int UB_BYTES = 48 * UB_ROW_BYTES;              // 98304
constexpr int LDS_BYTES_B = U_BASE + UB_BYTES;           // 162816 <= 160 KB

__device__ __forceinline__ f32x4 mfma_bf16(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma_bf16_k16(const u32x2& a, const u32x2& b, const f32x4& c) {
    f32x4 d;
    asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %3" : "=a"(d) : "v"(a), "v"(b), "a"(c));
    return d;
}
// Y -= bf16 piece pairs p0..p7 (pair j covers Y[2j], Y[2j+1]), exactly.  v_dot2c_f32_bf16 is a DOT instruction: another VALU op
// that reads its result needs 3 wait states behind it (GCNHazardRecognizer: DotWriteDifferentVALURead), which the hazard
// recogniser cannot see in inline asm -- the block ends with them.  K0 / K1 = the bf16 pairs {-1, 0} / {0, -1}: x -= lo(p) / hi(p).
__device__ __forceinline__ void dot2c_sub16(float (&Y)[16], unsigned p0, unsigned p1, unsigned p2, unsigned p3, unsigned p4, unsigned p5,
                                            unsigned p6, unsigned p7) {
    asm volatile(
        "v_dot2c_f32_bf16 %0, %24, %16\n\tv_dot2c_f32_bf16 %1, %25, %16\n\tv_dot2c_f32_bf16 %2, %24, %17\n\tv_dot2c_f32_bf16 %3, %25, %17\n\t"
        "v_dot2c_f32_bf16 %4, %24, %18\n\tv_dot2c_f32_bf16 %5, %25, %18\n\tv_dot2c_f32_bf16 %6, %24, %19\n\tv_dot2c_f32_bf16 %7, %25, %19\n\t"
        "v_dot2c_f32_bf16 %8, %24, %20\n\tv_dot2c_f32_bf16 %9, %25, %20\n\tv_dot2c_f32_bf16 %10, %24, %21\n\tv_dot2c_f32_bf16 %11, %25, %21\n\t"
        "v_dot2c_f32_bf16 %12, %24, %22\n\tv_dot2c_f32_bf16 %13, %25, %22\n\tv_dot2c_f32_bf16 %14, %24, %23\n\tv_dot2c_f32_bf16 %15, %25, %23\n\ts_nop 2"
        : "+v"(Y[0]), "+v"(Y[1]), "+v"(Y[2]), "+v"(Y[3]), "+v"(Y[4]), "+v"(Y[5]), "+v"(Y[6]), "+v"(Y[7]),
          "+v"(Y[8]), "+v"(Y[9]), "+v"(Y[10]), "+v"(Y[11]), "+v"(Y[12]), "+v"(Y[13]), "+v"(Y[14]), "+v"(Y[15])
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7), "s"(0x0000bf80u), "s"(0xbf800000u));
}
__device__ __forceinline__ void acc_read1(float& d, const float& a) { asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(d) : "a"(a)); }

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
// a parked LDS address: below the plane size, so that base + ring-slot offset folds into the DS offset field
__device__ __forceinline__ unsigned unpark_lds(unsigned a) {
    const unsigned v = unpark(a);
    __builtin_assume(v < (unsigned)PLANE_BYTES + 4096u);
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
__host__ __device__ constexpr unsigned ub_row_off(int q) { return (unsigned)(q * 4 * UB_ROW_BYTES); }       // the image is stored in slot order: row (py, dz) of slot q, px = 0

template <bool RELU, bool CLIP>
__global__ void __launch_bounds__(NT, 1) conv16_wino_bf16_kernel(WinoArgs a, int nwg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = lane & 15, g = lane >> 4;
    auto ldsr = [&](unsigned off) -> f32x4 { return *reinterpret_cast<const f32x4*>(smem + off); };
    auto ldsu = [&](unsigned off) -> u32x4 { return *reinterpret_cast<const u32x4*>(smem + off); };
    typedef const f32x4 __attribute__((address_space(3)))* lds_cf4;
    auto lds_abs = [&](unsigned addr) -> f32x4 { return *(lds_cf4)(unsigned long long)addr; };     // absolute LDS address
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

    // ---- plane staging: lane L of chunk c fetches whatever belongs into the 16-byte LDS unit 64c + L.  Layout of a plane (this kernel's
    //      own): voxel v = 18 yrow + 9 (x & 1) + (x >> 1) of the haloed 18 x 18 plane (even / odd columns apart: a patch's dx = 0, 2 and
    //      1, 3 are neighbours), channel quad c4 -> unit 4 v + c4 + 2 (v / 36), i.e. two pad units behind every pair of rows.  A lane's 16
    //      patch addresses are then ONE per-lane base + compile-time constants (no swizzle to precompute, no address registers to park).
    //      Bank quads: byte / 16 = 4 TX + g + 2 TY + const (mod 16).  A ds_read_b128 is served in four groups of 16 lanes
    //      ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32: MI355X_MICROARCH.md, LDS) -- tiles of two y rows x two channel
    //      quads each: 16 different bank quads.  (With ONE pad unit the counter showed 0.13 of the LDS cycles in conflicts.)
    unsigned rel[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int slot = (wave * 5 + it) * 64 + lane;
        const int R = slot / 146, rem = slot - R * 146;
        const int v = 36 * R + (rem >> 2), c4 = rem & 3;
        const int yrow = v / 18, r = v - yrow * 18, par = r >= 9 ? 1 : 0, col = r - 9 * par, xi = 2 * col + par;
        const int y = Y0 - 1 + yrow, x = X0 - 1 + xi;
        const bool ok = rem < 144 && v < PLANE_VOX && y >= 0 && y < a.H && x >= 0 && x < a.W;
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
    const unsigned pa0 = (unsigned)(64 * (36 * TY + TX) + 16 * (g + 2 * TY));       // patch (dy, dx) = (0, 0) of this lane in ring slot 0
    auto pa_off = [](int dy, int dx) constexpr -> unsigned { return (unsigned)(64 * (18 * dy + 9 * (dx & 1) + (dx >> 1)) + (dy >= 2 ? 32 : 0)); };
    const unsigned ua = (unsigned)(U_BASE + lane * 16);

    // ---- epilogue addressing: lane writes couts 4g..4g+3 of the 2x2 voxels of its tile
    const int ox0 = X0 + 2 * TX, oy0 = Y0 + 2 * TY;
    // voxel q = (oy, ox) of the 2 x 2 outputs: per-lane offset of voxel 0 + a wave-uniform offset (the buffer instructions' soffset)
    const unsigned vox0 = (unsigned)(oy0 * a.W + ox0);
    const unsigned ovo0 = (vox0 * (unsigned)a.ocs + (unsigned)a.oco + 16u * cog + 4u * g) * 4u;
    const unsigned rvo0 = (vox0 * (unsigned)a.rcs + 16u * cog + 4u * g) * 4u;
    unsigned oso[4], rso[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        oso[q] = (unsigned)(((q >> 1) * a.W + (q & 1)) * a.ocs) * 4u;
        rso[q] = (unsigned)(((q >> 1) * a.W + (q & 1)) * a.rcs) * 4u;
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
    f32x4 P0[4], P1[4], P2[4], P3[4];      // x-transformed patch rows of the NEXT plane (two are live at a time)
    float Y[16];                // fp32 row of V on its way through the split: (point x, channel c) at 4x + c
    float Mr[16];               // accumulators of the finished plane's row on their way through A^T
    u32x4 A1[4], A2[4];         // U fragments of the row in flight: [Uh | Um], [Ul | Uh] per point px
    f32x4 acc[3][16];           // three output planes in flight
    f32x4 S[2][2];
    f32x4 resv[4], ost[4];
    // ---- pieces of the schedule.  V row r of a plane: Y = its fp32 values (yrow), then H = cvt(Y), Y -= H, M = cvt(Y), Y -= M,
    //      L = cvt(Y).  The cvts are single VALU ops the compiler sees (fillers between MFMAs); the subtractions are asm blocks.
    auto yrow = [&](auto r_tag) __attribute__((always_inline)) {
        constexpr int r = decltype(r_tag)::value;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const f32x4 y = r == 0 ? sub4(P0[x], P2[x]) : r == 1 ? add4(P1[x], P2[x]) : r == 2 ? sub4(P2[x], P1[x]) : sub4(P1[x], P3[x]);
            Y[4 * x + 0] = y[0]; Y[4 * x + 1] = y[1]; Y[4 * x + 2] = y[2]; Y[4 * x + 3] = y[3];
        }
    };
    // pair j = (point x = j >> 1, channel pair hf = j & 1) of row r: stage 0 = h (also into B2), 1 = m, 2 = l
    auto cvt_task = [&](auto r_tag, auto st_tag, auto j_tag) __attribute__((always_inline)) {
        constexpr int r = decltype(r_tag)::value, st = decltype(st_tag)::value, j = decltype(j_tag)::value;
        constexpr int x = j >> 1, hf = j & 1;
        const unsigned v = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){Y[4 * x + 2 * hf], Y[4 * x + 2 * hf + 1]}, bf16x2));
        if (st == 0) { B1[r * 4 + x][hf] = v; B2[r * 4 + x][2 + hf] = v; }
        else if (st == 1) B1[r * 4 + x][2 + hf] = v;
        else B2[r * 4 + x][hf] = v;
    };
    auto dot_block = [&](auto r_tag, auto st_tag) __attribute__((always_inline)) {      // Y -= piece `st` (0: h, 1: m) of row r
        constexpr int r = decltype(r_tag)::value, o = decltype(st_tag)::value == 0 ? 0 : 2;
        dot2c_sub16(Y, B1[r * 4 + 0][o], B1[r * 4 + 0][o + 1], B1[r * 4 + 1][o], B1[r * 4 + 1][o + 1], B1[r * 4 + 2][o], B1[r * 4 + 2][o + 1],
                    B1[r * 4 + 3][o], B1[r * 4 + 3][o + 1]);
    };
    auto load_prow = [&](f32x4 (&P)[4], int dy, unsigned slot_off) __attribute__((always_inline)) {
#pragma unroll
        for (int x = 0; x < 4; ++x) P[x] = ldsr(pa0 + pa_off(dy, x) + slot_off);
    };
    using R0 = std::integral_constant<int, 0>;
    using R1 = std::integral_constant<int, 1>;
    using R2 = std::integral_constant<int, 2>;
    using R3 = std::integral_constant<int, 3>;
    using ST0 = std::integral_constant<int, 0>;
    using ST1 = std::integral_constant<int, 1>;
    using ST2 = std::integral_constant<int, 2>;
    // a whole row at once (prologue only)
    auto split_row = [&](auto r_tag) __attribute__((always_inline)) {
        yrow(r_tag);
        static_for<0, 8>([&](auto j) __attribute__((always_inline)) { cvt_task(r_tag, ST0{}, j); });
        dot_block(r_tag, ST0{});
        static_for<0, 8>([&](auto j) __attribute__((always_inline)) { cvt_task(r_tag, ST1{}, j); });
        dot_block(r_tag, ST1{});
        static_for<0, 8>([&](auto j) __attribute__((always_inline)) { cvt_task(r_tag, ST2{}, j); });
    };
    {
        const unsigned so = (unsigned)s0 * PLANE_BYTES;
        load_prow(P0, 0, so); load_prow(P1, 1, so); load_prow(P2, 2, so); load_prow(P3, 3, so);
        transform_x_row(P0); transform_x_row(P1); transform_x_row(P2); transform_x_row(P3);
        split_row(R0{}); split_row(R1{}); split_row(R2{});
        yrow(R3{});                                  // row 3 goes through the split in slots 0..2 of the first step
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

    // park the per-lane address constants (see park()): 9 of them since the plane layout gives every lane ONE patch base
    unsigned rel_p[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) rel_p[i] = park(rel[i]);
    const unsigned pa_p = park((unsigned)(unsigned long long)(lds_ptr)smem + pa0);       // absolute LDS address: no base add at the use
    const unsigned ovo_p = park(ovo0), rvo_p = park(rvo0);
    unsigned long long res_pl = (unsigned long long)res_n + (unsigned long long)(long long)(zb - 2 + s0) * HWR;     // plane zo of step s0
    unsigned long long out_pl = (unsigned long long)out_n + (unsigned long long)(long long)(zb - 2 + s0) * HWO;
    in_pl += (unsigned long long)s0 * HWI;

    // One input plane: s = step index (input plane z = zb-1+s), PH = s mod 3; MODE as in conv16_wino_kernel.
    // Slot q = row (py = q / 3, dz = 2 - q % 3): 12 MFMAs, each followed by at most a few single-issue instructions that run in
    // its shadow (F: cvts, AccVGPR reads, LDS reads, DMA issue), then a block of packed adds / DOT ops (K).  Measured beside
    // v_mfma_f32_16x16x32_bf16 (tools/ubench/mfma_bf16_valu.hip): two cvt / and / mov / accvgpr_read class ops per MFMA are free,
    // a v_pk_add_f32 or v_dot2c next to an MFMA costs ~15 cycles of pipeline switch however many follow it, 4.1 - 4.5 each.
    //   input side, V row r of plane s+1 (its pieces are dead after slot 3r + 2):
    //     K(3r+2) yrow   F(3r+3) h   K(3r+3) Y -= h   F(3r+4) m   K(3r+4) Y -= m   F(3r+5) l        (r = 3 wraps into slots 0..2)
    //     patch rows: P2 F(0), P0 F(1), x-transforms K(1) / K(2);  P1 F(4), K(4);  P3 F(9), K(10);  plane s+2 -> LDS: F(3)
    //   output side, row r of the finished plane (its dz = 2 MFMAs ran in slot 3r): F(3r+2) AccVGPR reads, K(3r+2) A^T;
    //     K(8) / K(11): epilogue of output rows 0 / 1 and the residual loads of the NEXT plane into the registers just consumed.
    auto step = [&](auto ph_tag, int s, auto mode_tag) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_tag)::value;
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool FIN = MODE == MB_FIN;
        constexpr unsigned slotN = (unsigned)((PH + 1) % 3) * PLANE_BYTES;   // plane s+1 (read)
        constexpr unsigned slotW = (unsigned)((PH + 2) % 3) * PLANE_BYTES;   // plane s+2 (written)
        constexpr int AF = PH;                                               // acc slot of the plane finished by dz = 2
        const bool zo_ok = s >= 2;                                           // the finished plane zo = zb - 2 + s exists
        const __amdgpu_buffer_rsrc_t rout = make_rsrc((const void*)(zo_ok ? out_pl : (unsigned long long)out_n), zo_ok ? HWO : 0u);
        // residual of the finished plane: requested in K(0) / K(3), eight slots ahead of its use.  (No load is left in flight across
        // the step boundary: with VMEM loads pending at the loop back-edge the compiler's waitcnt pass can no longer tell them from the
        // LDS-direct loads and put vmcnt(0) in front of the first patch read of every third step.)
        const __amdgpu_buffer_rsrc_t rres = make_rsrc((const void*)(zo_ok ? res_pl : (unsigned long long)res_n), zo_ok && has_res ? HWR : 0u);
        res_pl += HWR; out_pl += HWO;
        const bool in_ok = (unsigned)(zb + 1 + s) < (unsigned)a.D;
        const __amdgpu_buffer_rsrc_t rin = make_rsrc((const void*)(in_ok ? in_pl : (unsigned long long)in_n), in_ok ? HWI : 0u);
        if (!FIN) in_pl += HWI;
        static_for<0, 12>([&](auto q_tag) __attribute__((always_inline)) {
            constexpr int q = decltype(q_tag)::value;
            constexpr int py = q / 3, dz = 2 - q % 3;
            constexpr int as = (PH + 2 - dz) % 3;
            constexpr bool active = mb_row_active(MODE, dz);
            constexpr int qn = mb_next_slot(MODE, q) % 12;
            const unsigned un = ua + ub_row_off(qn);
            constexpr bool opens = dz == 0 || (MODE == MB_S1O && dz == 1);
            using RS = std::integral_constant<int, (q / 3 + 3) % 4>;        // V row in the split pipeline during this slot
            using STG = std::integral_constant<int, q % 3>;                  // its stage
            unsigned pa_s = 0, ovo_s = 0, rvo_s = 0;                         // parked addresses this slot needs (one AccVGPR read each)
            if constexpr (!FIN && (q == 0 || q == 1 || q == 4 || q == 9)) pa_s = unpark_lds(pa_p);
            if constexpr (q == 8 || q == 11) ovo_s = unpark(ovo_p);
            if constexpr (q == 0 || q == 3) rvo_s = unpark(rvo_p);
            // ---- F(q): MFMA i, then what runs in its shadow
            static_for<0, 12>([&](auto i_tag) __attribute__((always_inline)) {
                constexpr int i = decltype(i_tag)::value;
                if constexpr (active && !(PCC_WB_PROBE & 16)) {
                    // two halves of two points each, three MFMAs per point, alternating between the points of a half (a dependent MFMA
                    // is two issue slots away); a new output plane starts from 0, except point (1,1), which enters all four outputs
                    // with weight +1 and carries the bias
                    constexpr int h = i / 6, tm = (i % 6) / 2, px = 2 * h + (i & 1);
                    constexpr int k = py * 4 + px;
                    const f32x4 c = (opens && tm == 0) ? ((py == 1 && px == 1) ? bias4 : zero4) : acc[as][k];
                    acc[as][k] = mfma_bf16(tm == 2 ? A2[px] : A1[px], tm == 1 ? B2[k] : B1[k], c);
                    // the fragments of a point are dead behind its third MFMA: the same registers receive the next row's
                    // (A2 first: LDS reads return in order, so the wait in front of the point's first MFMA -- on A1 -- covers both)
                    if constexpr (tm == 2 && !(PCC_WB_PROBE & 256)) {
                        A2[px] = ldsu(un + (unsigned)(px * UB_ROW_BYTES + 1024));
                        A1[px] = ldsu(un + (unsigned)(px * UB_ROW_BYTES));
                    }
                }
                if constexpr (!FIN) {
                    if constexpr (i < 8 && !(PCC_WB_PROBE & 32)) cvt_task(RS{}, STG{}, i_tag);
                    if constexpr (q == 0 && i >= 8 && !(PCC_WB_PROBE & 512)) P2[i - 8] = lds_abs(pa_s + pa_off(2, i - 8) + slotN);
                    if constexpr (q == 1 && i >= 8 && !(PCC_WB_PROBE & 512)) P0[i - 8] = lds_abs(pa_s + pa_off(0, i - 8) + slotN);
                    if constexpr (q == 4 && i >= 8 && !(PCC_WB_PROBE & 512)) P1[i - 8] = lds_abs(pa_s + pa_off(1, i - 8) + slotN);
                    if constexpr (q == 9 && i >= 8 && !(PCC_WB_PROBE & 512)) P3[i - 8] = lds_abs(pa_s + pa_off(3, i - 8) + slotN);
                    if constexpr (q == 3 && i >= 6 && !(PCC_WB_PROBE & 2))
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_ptr)(smem + slotW + (wave * 5 + i - 6) * 1024), 16, (int)unpark(rel_p[i - 6]), 0, 0, 0);
                }
                if constexpr (q % 3 == 2 && !(PCC_WB_PROBE & 64)) {
                    // AccVGPR reads of row q / 3 of the finished plane: one per MFMA in gaps 0..7, two in gaps 8..11
                    constexpr int r = q / 3;
                    constexpr int e0 = i < 8 ? i : 8 + 2 * (i - 8), ne = i < 8 ? 1 : 2;
#pragma unroll
                    for (int e = e0; e < e0 + ne; ++e)
                        acc_read1(Mr[e], acc[AF][r * 4 + (e >> 2)][e & 3]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            // ---- K(q)
            if constexpr (!FIN) {
                if constexpr (q % 3 == 0 && !(PCC_WB_PROBE & 8)) dot_block(RS{}, ST0{});
                if constexpr (q % 3 == 1 && !(PCC_WB_PROBE & 8)) dot_block(RS{}, ST1{});
                if constexpr (!(PCC_WB_PROBE & 128)) {
                    if constexpr (q == 1) transform_x_row(P2);
                    if constexpr (q == 2) { transform_x_row(P0); yrow(R0{}); }
                    if constexpr (q == 4) transform_x_row(P1);
                    if constexpr (q == 5) yrow(R1{});
                    if constexpr (q == 8) yrow(R2{});
                    if constexpr (q == 10) transform_x_row(P3);
                    if constexpr (q == 11) yrow(R3{});
                }
            }
            if constexpr (q % 3 == 2 && !(PCC_WB_PROBE & 64)) {
                // A^T along x on row r of the finished plane, accumulate A^T along y
                constexpr int r = q / 3;
                const f32x4 m0 = {Mr[0], Mr[1], Mr[2], Mr[3]}, m1 = {Mr[4], Mr[5], Mr[6], Mr[7]}, m2 = {Mr[8], Mr[9], Mr[10], Mr[11]}, m3 = {Mr[12], Mr[13], Mr[14], Mr[15]};
                const f32x4 r0 = add4(add4(m0, m1), m2), r1 = sub4(sub4(m1, m2), m3);
                if (r == 0) { S[0][0] = r0; S[0][1] = r1; }
                else if (r == 1) { S[0][0] = add4(S[0][0], r0); S[0][1] = add4(S[0][1], r1); S[1][0] = r0; S[1][1] = r1; }
                else if (r == 2) { S[0][0] = add4(S[0][0], r0); S[0][1] = add4(S[0][1], r1); S[1][0] = sub4(S[1][0], r0); S[1][1] = sub4(S[1][1], r1); }
                else { S[1][0] = sub4(S[1][0], r0); S[1][1] = sub4(S[1][1], r1); }
            }
            if constexpr ((q == 8 || q == 11) && !(PCC_WB_PROBE & (64 | 4))) {
                // epilogue of output row oy (complete after reduction row 2 resp. 3): ReLU, residual, clip, float4 stores
                constexpr int oy = q == 8 ? 0 : 1;
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
                for (int v = 2 * oy; v < 2 * oy + 2; ++v) buf_store4(rout, ost[v], ovo_s, oso[v]);
            }
            if constexpr ((q == 0 || q == 3) && !(PCC_WB_PROBE & 4)) {
#pragma unroll
                for (int v = 2 * (q / 3); v < 2 * (q / 3) + 2; ++v) resv[v] = buf_load4(rres, rvo_s, rso[v]);
            }
            // gfx950: a buffer_store_dwordx4 reads its data registers late (conv16_wino_kernel): keep them unwritten for one more slot
            if constexpr (q == 9) { asm volatile("" ::"v"(ost[0])); asm volatile("" ::"v"(ost[1])); }
            if constexpr (q == 0) { asm volatile("" ::"v"(ost[2])); asm volatile("" ::"v"(ost[3])); }
            __builtin_amdgcn_sched_barrier(0);
        });
        // the LDS-direct loads of plane s+2 (F(3)) must have landed before the barrier publishes them; younger: the residual loads of
        // K(3) and the 4 stores of K(8) / K(11)
        if (!FIN && !(PCC_WB_PROBE & 1)) {
            __builtin_amdgcn_s_waitcnt(0x0F76);      // vmcnt(6) expcnt(7) lgkmcnt(15)
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

// ---- host: split-bf16 image of the Winograd-transformed weights.  Per (cin group, cout group): [slot q = 3 py + 2 - dz][px][operand][lane][8 bf16]
//      (the order the kernel consumes it in: the first 48 KB are the rows of slots 0..5)
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
                const int dz = row / 16, py = (row / 4) % 4, px = row % 4, orow = (3 * py + 2 - dz) * 4 + px;
                unsigned short* a1 = o + ((((size_t)pair * 48 + orow) * 2 + 0) * 64 + lane) * 8;
                unsigned short* a2 = o + ((((size_t)pair * 48 + orow) * 2 + 1) * 64 + lane) * 8;
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
