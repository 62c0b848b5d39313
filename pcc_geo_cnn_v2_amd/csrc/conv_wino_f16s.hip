// Winograd F(2x2, 3x3) in (x, y) + direct z taps on the f16 MFMA pipe with TWO-piece fp16 operands under an exact power-of-two
// pre-scale (round 6; the layers of /root/reference/src/model_transforms.py:62-81, AnalysisBlock / SynthesisBlock).
//
// conv_wino_bf16.hip splits every fp32 operand into three bf16 pieces: 3 MFMAs per (z tap, point), 7 VALU ops per split value
// pair, 128 registers of V pieces.  Here
//     x = h + l,   h = fp16_rn(s x),   l = fp16_rn(s x - h)          (11 + 11 significand bits; s x - h is exact)
// and ALL FOUR product terms are kept, two of them stacked along K of one v_mfma_f32_16x16x32_f16 (K = 2 terms x 16 cin):
//     acc += [Uh | Uh] . [Vh | Vl]        (hh + hl)
//     acc += [Ul | Ul] . [Vh | Vl]        (lh + ll)
// = 2 MFMAs per (z tap, point) with ONE B operand (V: 4 VGPRs per point, 64 in all), the duplication sits in the weight image the
// host prepares (U: 32 B per lane and row, 96 KB in LDS as before).  Split of a value pair: cvt_pk + v_fma_mixlo / mixhi = 3 VALU ops.
// Products of fp16 pieces are exact in fp32; accumulation is fp32 in a fixed order.  Operand error 2^-22 relative (numpy model
// tools/model_f16_split.py: 1.0-1.3e-7 of 1 + max|ref| against 4-6e-8 for fp32 operands; the gate is 8e-6).
//
// fp16's exponent range is the price.  s is an exact power of two, so it commutes with every rounding in the path:
//   * weights: ONE scale per layer, chosen on the host so that max |U| lands in [2^13, 2^14) (pcc_wino_f16s_pack; the image's tail
//     holds it);
//   * activations: one scale PER BLOCK n (never per launch: a per-launch max would make the low piece's denormal rounding depend on
//     what else is in the batch and break the encoder-chunk / decoder-chunk / shard bit-equality the codec needs).  The producer of
//     the input tensor records max |x| of each block (atomicMax on the fp32 bit pattern: order-independent; this kernel does it for
//     its own output), or pcc_block_amax computes it; s = 2^(12 - floor(log2 max)), so that |V| = |B^T d B| <= 4 max lands below 2^15.
//     The low piece's denormal step is then 2^-37 of the block's max -- below fp32 accumulation noise.  NaNs are left out of the max
//     (a NaN stays local), an all-zero block takes s = 1.
// The scale is applied inside the x transform (packed fma: s d0 - s d2 ...: exact, +16 packed ops per plane) and undone in the
// epilogue (the residual add becomes a packed fma).  Scaling the operands of a launch by powers of two therefore scales the result
// bit for bit, as with the bf16 pieces.
//
// Everything else follows conv_wino_bf16.hip: workgroup = 4 waves = 16 x 16 (x, y) outputs marching along z, ring of 3 input planes
// in LDS filled by buffer_load ... lds, lane = (tile, cin quad) transforms its own 4 x 4 patch, three output planes in flight in
// 192 AccVGPRs, A^T . A lane-local, rows py-major.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include "wino_common.h"

// timing probes (tools/build_variant.sh): 1 no step barrier / vmcnt wait, 2 no plane loads, 4 no residual loads / stores, 8 no DOT
// blocks (none here), 16 no MFMAs, 32 no split, 64 no output side, 128 no input transforms, 256 no U refills, 512 no patch reads
#ifndef PCC_WH_PROBE
#define PCC_WH_PROBE 0
#endif
// where the 16 v_fma_mix ops of a V row go: 0 = two per MFMA gap of slot 3r + 4, 1 = one block behind that slot's MFMAs (all-asm LDS form: 0 is 3 % faster, wait_lds 0.038 vs 0.086 of the wave cycles)
#ifndef PCC_WH_MIXK
#define PCC_WH_MIXK 0
#endif

namespace pccwino {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// U in LDS, lane-contiguous: lane L of (cin group, cout group) owns 776 B = [slot q = 3 py + 2 - dz][px][4 h | 4 l fp16] (768 B) + 8 B
// pad.  One ds_read2_b64 with offset0 == offset1 delivers the 8 bytes of a piece TWICE into four consecutive registers: the
// duplicated MFMA operands [Uh | Uh] / [Ul | Ul] without a duplicated image (48.5 KB per group pair instead of 96 KB: two cin groups
// fit beside the tile ring).  Every offset fits the 8-bit x 8 B field of that instruction, so a lane needs ONE address per cin group;
// 776 = 3 x 256 + 8 spreads 32 lanes over all 64 banks (no conflicts).
constexpr int UL_LANE_BYTES = 776;
constexpr int UG_BYTES = 64 * UL_LANE_BYTES;             // 49664 per (cin group, cout group)
__host__ __device__ constexpr int f16s_lds_bytes(int G) { return U_BASE + ((G * UG_BYTES + 1023) / 1024) * 1024; }      // 114688 / 163840 (= 160 KB)

__device__ __forceinline__ f32x4 mfma_f16(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// low pieces of one value pair: L = {fp16_rn(a - H.lo), fp16_rn(b - H.hi)} -- subtraction and rounding FUSED in v_fma_mix{lo,hi}_f16
// (H.lo x -1.0 + a with the fp16 operand widened exactly, one rounding to fp16; a - H.lo is exact in fp32 anyway, so these are the
// bits of cvt(a - H.lo)): 3 VALU ops per value pair with the cvt_pk of H, against cvt_pk + two dot2c + cvt_pk.  Not DOT
// instructions: no wait states behind them.
__device__ __forceinline__ unsigned mix_low_pair(unsigned H, float a, float b) {
    unsigned L;
    asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(L) : "v"(H), "v"(a), "v"(b));
    return L;
}
// ---- LDS reads as inline asm.  The compiler does not know these are LDS operations: it inserts no s_waitcnt for them -- in
// particular not the lgkmcnt(0) it puts in front of every s_barrier while reads it knows about are in flight (the U prefetch of the
// next step's first row used to be drained there, once per step) -- and the kernel waits by count: LDS reads return in order, ALL
// LDS reads of the march are issued here, at fixed places of the compile-time schedule, so "at most N younger reads outstanding"
// is a constant per place.  The wait carries the registers it covers as in-out operands: nothing that uses them can be moved above it.
template <int OFF>
__device__ __forceinline__ u32x4 lds_read2_dup(unsigned addr) {      // {8 bytes at addr + OFF} twice
    u32x4 v;
#if PCC_WH_PROBE & 2048      // timing only: one 16-byte read at the same place (wrong operands)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF & ~15));
#else
    asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%2" : "=v"(v) : "v"(addr), "n"(OFF / 8));
#endif
    return v;
}
template <int OFF>
__device__ __forceinline__ f32x4 lds_read128(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N>
__device__ __forceinline__ void lgkm_wait_u(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"((PCC_WH_PROBE & 4096) ? 15 : N));
}
template <int N>
__device__ __forceinline__ void lgkm_wait_p(f32x4 (&P)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(P[0]), "+v"(P[1]), "+v"(P[2]), "+v"(P[3]) : "n"(N));
}
__device__ __forceinline__ void acc_read1h(float& d, const float& a) { asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(d) : "a"(a)); }

// packed helpers with the block's scale s2 = {s, s}: a s, a s - c, c - a s (c already scaled)
__device__ __forceinline__ f32x4 mul4s(const f32x4& a, const f32x2& s) {
    f32x2 lo, hi;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(__builtin_shufflevector(a, a, 0, 1)), "v"(s));
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(__builtin_shufflevector(a, a, 2, 3)), "v"(s));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ f32x4 fms4s(const f32x4& a, const f32x2& s, const f32x4& c) {
    f32x2 lo, hi;
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(lo) : "v"(__builtin_shufflevector(a, a, 0, 1)), "v"(s), "v"(__builtin_shufflevector(c, c, 0, 1)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(hi) : "v"(__builtin_shufflevector(a, a, 2, 3)), "v"(s), "v"(__builtin_shufflevector(c, c, 2, 3)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ f32x4 fnma4s(const f32x4& a, const f32x2& s, const f32x4& c) {
    f32x2 lo, hi;
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(lo) : "v"(__builtin_shufflevector(a, a, 0, 1)), "v"(s), "v"(__builtin_shufflevector(c, c, 0, 1)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(hi) : "v"(__builtin_shufflevector(a, a, 2, 3)), "v"(s), "v"(__builtin_shufflevector(c, c, 2, 3)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ f32x4 fma4s(const f32x4& a, const f32x2& s, const f32x4& c) {
    f32x2 lo, hi;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(__builtin_shufflevector(a, a, 0, 1)), "v"(s), "v"(__builtin_shufflevector(c, c, 0, 1)));
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(__builtin_shufflevector(a, a, 2, 3)), "v"(s), "v"(__builtin_shufflevector(c, c, 2, 3)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
// s B^T along x on one patch row (4 voxels x 4 channels), in place: s (d0 - d2), s (d1 + d2), s (d2 - d1), s (d1 - d3) -- the same
// bits as s times the unscaled transform (s is a power of two)
__device__ __forceinline__ void transform_x_row_s(f32x4 (&P)[4], const f32x2& s) {
    const f32x4 d1 = mul4s(P[1], s), d2 = mul4s(P[2], s);
    P[0] = fms4s(P[0], s, d2); P[3] = fnma4s(P[3], s, d1); P[1] = add4(d1, d2); P[2] = sub4(d2, d1);
}
// running max |o| over the stored values (NaNs are skipped: v_max3_f32 returns the other operands)
__device__ __forceinline__ void amax4(float& m, const f32x4& o) {
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(o[0]), "v"(o[1]));
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(o[2]), "v"(o[3]));
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for_h(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_h<I + 1, N>(f);
    }
}

enum { MH_ALL = 0, MH_S0 = 1, MH_S1 = 2, MH_S1O = 3, MH_FIN = 4 };
__host__ __device__ constexpr bool mh_row_active(int mode, int dz) {
    return mode == MH_ALL || (mode == MH_S0 && dz == 0) || ((mode == MH_S1 || mode == MH_S1O) && dz <= 1);
}
__host__ __device__ constexpr int mh_next_mode(int mode) { return mode == MH_S0 ? MH_S1 : MH_ALL; }
// first active slot q' > q of this step, or 12 + the first active slot of the next step
__host__ __device__ constexpr int mh_next_slot(int mode, int q) {
    for (int n = q + 1; n < 12; ++n)
        if (mh_row_active(mode, 2 - n % 3)) return n;
    for (int n = 0; n < 12; ++n)
        if (mh_row_active(mh_next_mode(mode), 2 - n % 3)) return 12 + n;
    return 12;
}
__host__ __device__ constexpr int mh_first_slot(int mode) {
    for (int n = 0; n < 12; ++n)
        if (mh_row_active(mode, 2 - n % 3)) return n;
    return 0;
}

// The block's pre-scale from its recorded max |x| (fp32 bits m, finite): biased exponent of s = 127 + 12 - (e - 127), kept inside
// [1, 254] and such that s su stays inside 2^+-120 (su = the weight image's scale, exponent lsu).  m = 0: s = 1.
__device__ __forceinline__ unsigned f16s_scale_bits(unsigned m, int lsu) {
    if (m == 0u) return 0x3f800000u;
    int se = 266 - (int)(m >> 23);
    const int lo = 7 - lsu > 1 ? 7 - lsu : 1, hi = 247 - lsu < 254 ? 247 - lsu : 254;
    se = se < lo ? lo : se > hi ? hi : se;
    return (unsigned)se << 23;
}

// G = Cin / 16 = Cout / 16 in {1, 2}.  The march is a sequence of MICRO-STEPS m = (input plane s, cin group c), c fastest (G = 1: one per
// plane): micro-step m multiplies the pieces of tile m (registers B) with U[c]; meanwhile tile m + 1 is read from the LDS ring,
// transformed and split into the same registers behind their last use, and tile m + 2 arrives global -> LDS.  The three output planes
// in flight (192 AccVGPRs, ONE cout group per workgroup) stay live across the cin groups; only the last group of a plane reduces and
// stores the finished output plane.  Ring slot of tile m = m mod 3 = (G (s mod 3) + c) mod 3: compile time, like the accumulator phase.
// PRE (64-channel layers, second launch): `out` already holds the partial sums of the cin groups a first launch of this kernel
// marched (flags cleared: raw A^T sums, un-scaled); they are added before the ReLU, in that fixed order.
template <bool RELU, bool CLIP, int G, bool PRE = false>
__global__ void __launch_bounds__(NT, 1) conv16_wino_f16s_kernel(WinoArgs a, int nwg) {
    static_assert(G == 1 || G == 2, "one or two 16-channel cin groups");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = lane & 15, g = lane >> 4;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_ptr)smem;       // absolute LDS address of the dynamic segment

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

    // ---- U -> LDS: the G images of this cout group ([cout group][cin group][lane][776 B], contiguous) straight global -> LDS
    {
        // a.ncig = cin groups of the LAYER (the image holds [cout group][all cin groups]), a.ig0 = the first one this launch marches
        constexpr int UBYTES = G * UG_BYTES, NCH = (UBYTES + 1023) / 1024;
        const __amdgpu_buffer_rsrc_t ru = make_rsrc(a.u + ((size_t)cog * a.ncig + a.ig0) * (UG_BYTES / 4), (unsigned)UBYTES);      // beyond the image: zeros
#pragma unroll
        for (int k = 0; k < (NCH + 3) / 4; ++k) {
            const int chunk = k * 4 + wave;
            if (chunk < NCH)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr)(smem + U_BASE + chunk * 1024), 16, (int)(lane * 16), chunk * 1024, 0, 0);
        }
    }
    // ---- scales (wave-uniform): su of the weight image, s of this block, their product for the bias, its inverse for the epilogue
    const float su = a.utail[0];
    const unsigned sbits = f16s_scale_bits(pcc_amax_read(a.amax_in + (size_t)n * PCC_AMAX_SLOTS), (int)(__builtin_bit_cast(unsigned, su) >> 23) - 127);
    const float sv = __builtin_bit_cast(float, sbits);
    const float sprod = su * sv, sinv = 1.0f / sprod;          // powers of two inside 2^+-120: exact
    const f32x2 s2 = {sv, sv}, inv2 = {sinv, sinv};

    // ---- tile staging (layout and addressing: conv_wino_bf16.hip)
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
    // tile (input plane index sp of this slab, cin group cg) -> ring slot `ring_off`
    auto stage_tile = [&](unsigned ring_off, int sp, int cg) __attribute__((always_inline)) {
        const int z = zb - 1 + sp;
        const bool ok = (unsigned)z < (unsigned)a.D;
        const __amdgpu_buffer_rsrc_t rp = make_rsrc(in_n + (ok ? (size_t)z * HW * a.ics + 16 * cg : 0), ok ? HWI - (unsigned)(64 * cg) : 0u);
#pragma unroll
        for (int it = 0; it < ITEMS; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lds_ptr)(smem + ring_off + (wave * 5 + it) * 1024), 16, (int)rel[it], 0, 0, 0);
    };

    // ---- per-lane LDS addresses (absolute): patch (dy, dx) = (0, 0) of this lane's tile in ring slot 0; this lane's U rows per cin group
    const int wx = wave & 1, wy = wave >> 1;
    const int TX = 4 * wx + (t & 3), TY = 4 * wy + (t >> 2);
    const unsigned pa0 = lds0 + (unsigned)(64 * (36 * TY + TX) + 16 * (g + 2 * TY));
    constexpr auto pa_off = [](int dy, int dx) constexpr -> int { return 64 * (18 * dy + 9 * (dx & 1) + (dx >> 1)) + (dy >= 2 ? 32 : 0); };
    unsigned ua[G];
#pragma unroll
    for (int c = 0; c < G; ++c) ua[c] = lds0 + (unsigned)(U_BASE + c * UG_BYTES + lane * UL_LANE_BYTES);

    // ---- epilogue addressing: lane writes couts 4g..4g+3 of the 2x2 voxels of its tile
    const int ox0 = X0 + 2 * TX, oy0 = Y0 + 2 * TY;
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
    f32x4 bias4 = (a.flags & PCC_CONV_BIAS) ? *reinterpret_cast<const f32x4*>(a.bias + 16 * cog + g * 4) : zero4;
#pragma unroll
    for (int c = 0; c < 4; ++c) bias4[c] *= sprod;          // rides in the accumulator of point (1,1), which is scaled by su s

    // head / tail planes as in conv16_wino_kernel (wave-uniform, decided outside the MFMA stream)
    const bool first_zero = zb == 0, last_zero = zb + a.zlen == a.D;
    const int s0 = first_zero ? 1 : 0;
    // tiles m0 = G s0 and m0 + 1 -> their ring slots
    stage_tile((unsigned)((G * s0) % 3) * PLANE_BYTES, s0, 0);
    stage_tile((unsigned)((G * s0 + 1) % 3) * PLANE_BYTES, G == 1 ? s0 + 1 : s0, G == 1 ? 0 : 1);
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): U and both tiles have landed
    __syncthreads();

    u32x4 B[16];                // pieces of s B^T d B of the current tile: [Vh | Vl]
    f32x4 P0[4], P1[4], P2[4], P3[4];      // x-transformed (and scaled) patch rows of the NEXT tile
    float Y[16];                // fp32 row of V on its way through the split: (point x, channel c) at 4x + c
    float Mr[16];               // accumulators of the finished plane's row on their way through A^T
    u32x4 A1[4], A2[4];         // U fragments of the row in flight: [Uh | Uh], [Ul | Ul] per point px
    f32x4 acc[3][16];           // three output planes in flight
    f32x4 S[2][2];
    f32x4 resv[4], ost[4], prev[PRE ? 4 : 1];
    float mx = 0.f;             // max |stored value| of this lane (amax_out)
    // ---- pieces of the schedule.  V row r of a tile: Y = its fp32 values (yrow), H = cvt_pk(Y), L = fma_mix(Y - H).
    auto yrow = [&](auto r_tag) __attribute__((always_inline)) {
        constexpr int r = decltype(r_tag)::value;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const f32x4 y = r == 0 ? sub4(P0[x], P2[x]) : r == 1 ? add4(P1[x], P2[x]) : r == 2 ? sub4(P2[x], P1[x]) : sub4(P1[x], P3[x]);
            Y[4 * x + 0] = y[0]; Y[4 * x + 1] = y[1]; Y[4 * x + 2] = y[2]; Y[4 * x + 3] = y[3];
        }
    };
    // pair j = (point x = j >> 1, channel pair hf = j & 1) of row r: stage 0 = h, 1 = l
    auto cvt_task = [&](auto r_tag, auto st_tag, auto j_tag) __attribute__((always_inline)) {
        constexpr int r = decltype(r_tag)::value, st = decltype(st_tag)::value, j = decltype(j_tag)::value;
        constexpr int x = j >> 1, hf = j & 1;
        if constexpr (st == 0)
            B[r * 4 + x][hf] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){Y[4 * x + 2 * hf], Y[4 * x + 2 * hf + 1]}, f16x2));
        else if constexpr (st == 1)
            B[r * 4 + x][2 + hf] = mix_low_pair(B[r * 4 + x][hf], Y[4 * x + 2 * hf], Y[4 * x + 2 * hf + 1]);
    };
    using R0 = std::integral_constant<int, 0>;
    using R1 = std::integral_constant<int, 1>;
    using R2 = std::integral_constant<int, 2>;
    using R3 = std::integral_constant<int, 3>;
    using ST0 = std::integral_constant<int, 0>;
    using ST1 = std::integral_constant<int, 1>;
    // a whole row at once (prologue only)
    auto split_row = [&](auto r_tag) __attribute__((always_inline)) {
        yrow(r_tag);
        static_for_h<0, 8>([&](auto j) __attribute__((always_inline)) { cvt_task(r_tag, ST0{}, j); });
        static_for_h<0, 8>([&](auto j) __attribute__((always_inline)) { cvt_task(r_tag, ST1{}, j); });
    };
    // U fragments of (slot q, point px) of cin group c
    auto load_u = [&](auto q_tag, auto px_tag, int c) __attribute__((always_inline)) {
        constexpr int q = decltype(q_tag)::value, px = decltype(px_tag)::value;
        A2[px] = lds_read2_dup<(q * 4 + px) * 16 + 8>(ua[c]);
        A1[px] = lds_read2_dup<(q * 4 + px) * 16>(ua[c]);
    };
    // patch row dy of the tile in ring slot SLOT: piece x
    auto load_p = [&](f32x4 (&P)[4], auto dy_tag, auto x_tag, auto slot_tag) __attribute__((always_inline)) {
        constexpr int dy = decltype(dy_tag)::value, x = decltype(x_tag)::value, SLOT = decltype(slot_tag)::value;
        P[x] = lds_read128<pa_off(dy, x) + SLOT * PLANE_BYTES>(pa0);
    };
    auto prologue = [&](auto slot_tag, auto first_slot_tag) __attribute__((always_inline)) {
        static_for_h<0, 4>([&](auto x) __attribute__((always_inline)) {
            load_p(P0, R0{}, x, slot_tag); load_p(P1, R1{}, x, slot_tag); load_p(P2, R2{}, x, slot_tag); load_p(P3, R3{}, x, slot_tag); });
        static_for_h<0, 4>([&](auto px) __attribute__((always_inline)) { load_u(first_slot_tag, px, 0); });
        lgkm_wait_p<0>(P0); lgkm_wait_p<0>(P1); lgkm_wait_p<0>(P2); lgkm_wait_p<0>(P3);
        lgkm_wait_u<0>(A1[0], A2[0], A1[1], A2[1]); lgkm_wait_u<0>(A1[2], A2[2], A1[3], A2[3]);
    };
    // tile m0 sits in ring slot (G s0) mod 3; the first step that runs is S1O (slab at z = 0) or S0
    if (first_zero) prologue(std::integral_constant<int, G % 3>{}, std::integral_constant<int, mh_first_slot(MH_S1O)>{});
    else prologue(std::integral_constant<int, 0>{}, std::integral_constant<int, mh_first_slot(MH_S0)>{});
    transform_x_row_s(P0, s2); transform_x_row_s(P1, s2); transform_x_row_s(P2, s2); transform_x_row_s(P3, s2);
    split_row(R0{}); split_row(R1{}); split_row(R2{});
    yrow(R3{});                                  // row 3 goes through the split in slots 0..1 of the first step
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][i] = zero4; acc[1][i] = zero4; acc[2][i] = zero4; }
#pragma unroll
    for (int q = 0; q < 4; ++q) { resv[q] = zero4; ost[q] = zero4; }

    unsigned long long res_pl = (unsigned long long)res_n + (unsigned long long)(long long)(zb - 2 + s0) * HWR;     // plane zo of step s0
    unsigned long long out_pl = (unsigned long long)out_n + (unsigned long long)(long long)(zb - 2 + s0) * HWO;
    unsigned long long in_pl = (unsigned long long)in_n + (unsigned long long)(long long)(zb - 1 + s0) * HWI;       // input plane of step s0 (cin group 0)

    // One micro-step: s = input plane index (z = zb-1+s), PH = s mod 3, C = cin group; MODE as in conv16_wino_kernel.
    // Slot q = row (py = q / 3, dz = 2 - q % 3): 8 MFMAs, each followed by a few single-issue instructions that run in its shadow
    // (F: cvts, AccVGPR reads, LDS reads, DMA issue), then a block of packed adds / fma_mix ops (K).
    //   input side, V row r of tile m+1 (its pieces are dead after slot 3r + 2):
    //     K(3r+2) yrow   F(3r+3) h = cvt_pk(Y)   K(3r+4) l = fma_mix(Y - h)                        (r = 3 wraps into slots 0..1)
    //     patch rows: P2 F(0), P0 F(1), x-transforms K(1) / K(2);  P1 F(4), K(5);  P3 F(9), K(10);  tile m+2 -> LDS: F(3)
    //   output side (last cin group only), row r of the finished plane (its dz = 2 MFMAs ran in slot 3r): F(3r+2) AccVGPR reads,
    //     K(3r+2) A^T;  K(8) / K(11): epilogue of output rows 0 / 1 and the residual loads of the NEXT plane.
    //   LDS reads, all inline asm, in program order per slot: gaps 0, 1, 4, 5: one patch piece each (slots 0, 1, 4, 9 only);
    //     gaps 2, 3, 6, 7: the two fragments of point px = 0, 1, 2, 3 of the NEXT active row.  Waits by count (steady state, MODE ALL):
    //     before MFMA 0 of a slot: at most 4 + 2 [previous slot read patches] younger reads may be outstanding; before MFMA 4:
    //     4 + 2 [this slot reads patches]; a patch row is used a slot after its reads: 12.  Head / tail modes wait for everything.
    auto step = [&](auto ph_tag, auto c_tag, int s, auto mode_tag) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_tag)::value, C = decltype(c_tag)::value;
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool FIN = MODE == MH_FIN, FIRST = C == 0, LAST = C == G - 1, EXACT = MODE == MH_ALL;
        constexpr int TN = (G * PH + C + 1) % 3, TW = (G * PH + C + 2) % 3;      // ring slots of tile m+1 (read) / m+2 (written)
        constexpr int AF = PH;                                               // acc slot of the plane finished by dz = 2
        constexpr bool ZO = MODE == MH_ALL || MODE == MH_FIN;                // the finished plane zo = zb - 2 + s exists (s >= 2)
        constexpr int NMODE = LAST ? mh_next_mode(MODE) : MODE;              // mode of the next micro-step
        const bool zo_ok = s >= 2;
        const __amdgpu_buffer_rsrc_t rout = make_rsrc((const void*)(zo_ok ? out_pl : (unsigned long long)out_n), zo_ok && LAST ? HWO : 0u);
        const __amdgpu_buffer_rsrc_t rres = make_rsrc((const void*)(zo_ok ? res_pl : (unsigned long long)res_n), zo_ok && has_res && LAST ? HWR : 0u);
        // tile m + 2 = (plane s + (C + 2) / G, cin group (C + 2) % G)
        constexpr int DS2 = (C + 2) / G, C2 = (C + 2) % G;
        const bool in_ok = (unsigned)(zb - 1 + s + DS2) < (unsigned)a.D;
        const __amdgpu_buffer_rsrc_t rin = make_rsrc((const void*)(in_ok ? in_pl + (unsigned long long)DS2 * HWI + 64ull * C2 : (unsigned long long)in_n), in_ok ? HWI - 64u * C2 : 0u);
        if constexpr (LAST) { res_pl += HWR; out_pl += HWO; in_pl += HWI; }
        static_for_h<0, 12>([&](auto q_tag) __attribute__((always_inline)) {
            constexpr int q = decltype(q_tag)::value;
            constexpr int py = q / 3, dz = 2 - q % 3;
            constexpr int as = (PH + 2 - dz) % 3;
            constexpr bool active = mh_row_active(MODE, dz);
            // the next active row: in this micro-step (same cin group) or the first one of the next micro-step
            constexpr int qraw = mh_next_slot(MODE, q);
            constexpr bool wraps = qraw >= 12;
            constexpr int qn = wraps ? mh_first_slot(NMODE) : qraw;
            constexpr int cn = wraps ? (C + 1) % G : C;
            constexpr bool opens = FIRST && (dz == 0 || (MODE == MH_S1O && dz == 1));
            constexpr bool preads = !FIN && (q == 0 || q == 1 || q == 4 || q == 9);                 // this slot reads a patch row
            constexpr bool pprev = !FIN && (q == 1 || q == 2 || q == 5 || q == 10);                  // the previous slot did
            using RS = std::integral_constant<int, (q / 3 + 3) % 4>;        // V row in the split pipeline during this slot
            using STG = std::integral_constant<int, q % 3>;                  // its stage (2: none)
            using TNt = std::integral_constant<int, TN>;
            using QN = std::integral_constant<int, qn>;
            // ---- F(q): MFMA i, then what runs in its shadow
            static_for_h<0, 8>([&](auto i_tag) __attribute__((always_inline)) {
                constexpr int i = decltype(i_tag)::value;
                if constexpr (active && !(PCC_WH_PROBE & 16)) {
                    // two halves of two points each, two MFMAs per point, alternating between the points of a half (a dependent MFMA
                    // is two issue slots away); a new output plane starts from 0, except point (1,1), which enters all four outputs
                    // with weight +1 and carries the bias
                    constexpr int h = i / 4, tm = (i % 4) / 2, px = 2 * h + (i & 1);
                    constexpr int k = py * 4 + px;
                    if constexpr (i == 0) lgkm_wait_u<EXACT ? 4 + 2 * (pprev ? 1 : 0) : 0>(A1[0], A2[0], A1[1], A2[1]);
                    if constexpr (i == 4) lgkm_wait_u<EXACT ? 4 + 2 * (preads ? 1 : 0) : 0>(A1[2], A2[2], A1[3], A2[3]);
                    const f32x4 c = (opens && tm == 0) ? ((py == 1 && px == 1) ? bias4 : zero4) : acc[as][k];
                    acc[as][k] = mfma_f16(tm == 1 ? A2[px] : A1[px], B[k], c);
                    // the fragments of a point are dead behind its second MFMA: the same registers receive the next row's
                    if constexpr (tm == 1 && !(PCC_WH_PROBE & 256)) load_u(QN{}, std::integral_constant<int, px>{}, cn);
                }
                if constexpr (!FIN) {
                    if constexpr (!(PCC_WH_PROBE & 32) && (STG::value == 0 || (STG::value == 1 && !PCC_WH_MIXK))) cvt_task(RS{}, STG{}, i_tag);
                    if constexpr (preads && (i == 0 || i == 1 || i == 4 || i == 5) && !(PCC_WH_PROBE & 512)) {
                        using X = std::integral_constant<int, i < 2 ? i : i - 2>;
                        if constexpr (q == 0) load_p(P2, R2{}, X{}, TNt{});
                        if constexpr (q == 1) load_p(P0, R0{}, X{}, TNt{});
                        if constexpr (q == 4) load_p(P1, R1{}, X{}, TNt{});
                        if constexpr (q == 9) load_p(P3, R3{}, X{}, TNt{});
                    }
                    if constexpr (q == 3 && i >= 2 && !(PCC_WH_PROBE & 2))
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_ptr)(smem + TW * PLANE_BYTES + (wave * 5 + i - 2) * 1024), 16, (int)rel[i - 2], 0, 0, 0);
                }
                if constexpr (LAST && q % 3 == 2 && !(PCC_WH_PROBE & 64)) {
                    // AccVGPR reads of row q / 3 of the finished plane: two per MFMA gap
                    constexpr int r = q / 3;
#pragma unroll
                    for (int e = 2 * i; e < 2 * i + 2; ++e)
                        acc_read1h(Mr[e], acc[AF][r * 4 + (e >> 2)][e & 3]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            // ---- K(q)
            if constexpr (!FIN) {
                if constexpr (q % 3 == 1 && PCC_WH_MIXK && !(PCC_WH_PROBE & 32))
                    static_for_h<0, 8>([&](auto j) __attribute__((always_inline)) { cvt_task(RS{}, ST1{}, j); });
                if constexpr (!(PCC_WH_PROBE & 128)) {
                    if constexpr (q == 1) { lgkm_wait_p<EXACT ? 12 : 0>(P2); transform_x_row_s(P2, s2); }
                    if constexpr (q == 2) { lgkm_wait_p<EXACT ? 12 : 0>(P0); transform_x_row_s(P0, s2); yrow(R0{}); }
                    if constexpr (q == 5) { lgkm_wait_p<EXACT ? 12 : 0>(P1); transform_x_row_s(P1, s2); yrow(R1{}); }
                    if constexpr (q == 8) yrow(R2{});
                    if constexpr (q == 10) { lgkm_wait_p<EXACT ? 12 : 0>(P3); transform_x_row_s(P3, s2); }
                    if constexpr (q == 11) yrow(R3{});
                }
            }
            if constexpr (LAST && q % 3 == 2 && !(PCC_WH_PROBE & 64)) {
                // A^T along x on row r of the finished plane, accumulate A^T along y
                constexpr int r = q / 3;
                const f32x4 m0 = {Mr[0], Mr[1], Mr[2], Mr[3]}, m1 = {Mr[4], Mr[5], Mr[6], Mr[7]}, m2 = {Mr[8], Mr[9], Mr[10], Mr[11]}, m3 = {Mr[12], Mr[13], Mr[14], Mr[15]};
                const f32x4 r0 = add4(add4(m0, m1), m2), r1 = sub4(sub4(m1, m2), m3);
                if (r == 0) { S[0][0] = r0; S[0][1] = r1; }
                else if (r == 1) { S[0][0] = add4(S[0][0], r0); S[0][1] = add4(S[0][1], r1); S[1][0] = r0; S[1][1] = r1; }
                else if (r == 2) { S[0][0] = add4(S[0][0], r0); S[0][1] = add4(S[0][1], r1); S[1][0] = sub4(S[1][0], r0); S[1][1] = sub4(S[1][1], r1); }
                else { S[1][0] = sub4(S[1][0], r0); S[1][1] = sub4(S[1][1], r1); }
            }
            if constexpr (LAST && (q == 8 || q == 11) && !(PCC_WH_PROBE & (64 | 4))) {
                // epilogue of output row oy (complete after reduction row 2 resp. 3): ReLU, un-scale + residual (one packed fma), clip, stores
                constexpr int oy = q == 8 ? 0 : 1;
#pragma unroll
                for (int v = 2 * oy; v < 2 * oy + 2; ++v) {
                    f32x4 o = S[v >> 1][v & 1];
                    if constexpr (PRE) {
                        o = fma4s(o, inv2, prev[v]);     // + the partial sums of the first launch, then ReLU, then the residual
                        if (RELU) o = __builtin_elementwise_maximum(o, zero4);
                        o = add4(o, resv[v]);
                    } else {
                        if (RELU) o = __builtin_elementwise_maximum(o, zero4);
                        o = fma4s(o, inv2, resv[v]);     // zeros without PCC_CONV_ADD (zero-sized buffer)
                    }
                    if (CLIP) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] = fminf(fmaxf(o[c], 0.f), 1.f);
                    }
                    ost[v] = o;
                    if constexpr (ZO) amax4(mx, o);
                }
#pragma unroll
                for (int v = 2 * oy; v < 2 * oy + 2; ++v) buf_store4(rout, ost[v], ovo0, oso[v]);
            }
            if constexpr (LAST && (q == 0 || q == 3) && !(PCC_WH_PROBE & 4)) {
#pragma unroll
                for (int v = 2 * (q / 3); v < 2 * (q / 3) + 2; ++v) {
                    resv[v] = buf_load4(rres, rvo0, rso[v]);
                    if constexpr (PRE) prev[v] = buf_load4(rout, ovo0, oso[v]);
                }
            }
            // gfx950: a buffer_store_dwordx4 reads its data registers late (conv16_wino_kernel): keep them unwritten for one more slot
            if constexpr (LAST && q == 9) { asm volatile("" ::"v"(ost[0])); asm volatile("" ::"v"(ost[1])); }
            if constexpr (LAST && q == 0) { asm volatile("" ::"v"(ost[2])); asm volatile("" ::"v"(ost[3])); }
            __builtin_amdgcn_sched_barrier(0);
        });
        // the LDS-direct loads of tile m+2 (F(3)) must have landed before the barrier publishes them; younger (last cin group only):
        // the residual loads of K(3) and the 4 stores of K(8) / K(11).  LDS reads stay in flight across the barrier.
        if (!FIN && !(PCC_WH_PROBE & 1)) {
            if constexpr (LAST && PRE) __builtin_amdgcn_s_waitcnt(0x0F78);       // vmcnt(8): + the two partial-sum loads of K(3)
            else if constexpr (LAST) __builtin_amdgcn_s_waitcnt(0x0F76);      // vmcnt(6) expcnt(7) lgkmcnt(15)
            else __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
            __syncthreads();
        }
    };
    using P0t = std::integral_constant<int, 0>;
    using P1t = std::integral_constant<int, 1>;
    using P2t = std::integral_constant<int, 2>;
    using MAll = std::integral_constant<int, MH_ALL>;
    using MFin = std::integral_constant<int, MH_FIN>;
    // one input plane = G micro-steps
    auto plane = [&](auto ph_tag, int s, auto mode_tag) __attribute__((always_inline)) {
        static_for_h<0, G>([&](auto c_tag) __attribute__((always_inline)) { step(ph_tag, c_tag, s, mode_tag); });
    };
    using CL = std::integral_constant<int, G - 1>;

    if (first_zero) plane(P1t{}, 1, std::integral_constant<int, MH_S1O>{});
    else {
        plane(P0t{}, 0, std::integral_constant<int, MH_S0>{});
        plane(P1t{}, 1, std::integral_constant<int, MH_S1>{});
    }
    const int nloop = nsteps - (last_zero ? 1 : 0);
    for (int s = 2; s < nloop; s += 3) {
        plane(P2t{}, s, MAll{});
        if (s + 1 < nloop) plane(P0t{}, s + 1, MAll{});
        if (s + 2 < nloop) plane(P1t{}, s + 2, MAll{});
    }
    if (last_zero) {
        // the padding plane behind the volume: no matrix work; its last micro-step reduces and stores output plane D - 1
        const int sl = nsteps - 1, ph = sl % 3;
        if (ph == 0) step(P0t{}, CL{}, sl, MFin{});
        else if (ph == 1) step(P1t{}, CL{}, sl, MFin{});
        else step(P2t{}, CL{}, sl, MFin{});
    }
    // ---- max |out| of block n for the next layer's pre-scale (order-independent: atomicMax on non-negative fp32 bit patterns)
    if (a.amax_out != nullptr) pcc_amax_record(a.amax_out + (size_t)n * PCC_AMAX_SLOTS, mx, (int)blockIdx.x * 4 + wave);
}

// per-block max |x| over finite values, as the producers record it: one workgroup per (block, chunk), atomicMax of the bit pattern
__global__ void __launch_bounds__(256) block_amax_kernel(const float* __restrict__ x, size_t per_block, int chunks, unsigned* __restrict__ amax) {
    const int n = blockIdx.x / chunks, c = blockIdx.x % chunks;
    const f32x4* p = reinterpret_cast<const f32x4*>(x + (size_t)n * per_block);
    const size_t nv = per_block / 4;
    float m = 0.f;
    for (size_t i = (size_t)c * 256 + threadIdx.x; i < nv; i += (size_t)chunks * 256) {
        const f32x4 v = p[i];
        amax4(m, v);
    }
    for (size_t i = nv * 4 + (size_t)c * 256 + threadIdx.x; i < per_block; i += (size_t)chunks * 256) m = fmaxf(m, fabsf(x[(size_t)n * per_block + i]));
    pcc_amax_record(amax + (size_t)n * PCC_AMAX_SLOTS, m, c * 4 + (int)(threadIdx.x >> 6));
}

}  // namespace pccwino

using namespace pccwino;

// row n of amax (PCC_AMAX_SLOTS partial maxima) = bits of max |x| over block n (N blocks of per_block floats, contiguous); zeroed first
int pcc_block_amax(pcc_ctx* ctx, const float* x, int N, size_t per_block, unsigned* amax, hipStream_t st) {
    PCC_CHECK_HIP(hipMemsetAsync(amax, 0, (size_t)N * PCC_AMAX_SLOTS * sizeof(unsigned), st));
    if (N <= 0 || per_block == 0) return PCC_OK;
    size_t want = (per_block / 4 + 2047) / 2048;          // >= 8 float4 per thread
    const size_t cap = (size_t)(ctx->num_cu * 8 + N - 1) / (size_t)N;
    int chunks = (int)(want < 1 ? 1 : want > cap ? cap : want);
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(block_amax_kernel, dim3((unsigned)(N * chunks)), dim3(256), 0, st, x, per_block, chunks, amax);
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}

// ---- host: two-piece fp16 image of the Winograd-transformed weights, lane-contiguous (the LDS layout: one linear copy per workgroup).
//      [cout group][cin group][lane][slot q = 3 py + 2 - dz][px][4 h | 4 l fp16] + 8 B pad per lane;
//      cin = 16 cig + 4 (lane >> 4) + c, cout = 16 cog + (lane & 15); behind the pairs: PCC_WINO_UH_TAIL floats, [0] = su (the power of
//      two all pieces were scaled by)
static inline unsigned short f16_bits(float v) {
    const _Float16 h = (_Float16)v;          // round to nearest even, denormals kept
    unsigned short b;
    memcpy(&b, &h, 2);
    return b;
}
static inline float f16_value(unsigned short b) {
    _Float16 h;
    memcpy(&h, &b, 2);
    return (float)h;
}
// u_f32: the fp32 Winograd image of conv_wino.hip ([cin group][cout group][48][64 lanes][4]) -> out: PCC_WINO_UH_FLOATS per pair + tail
void pcc_wino_f16s_pack(int ngroups, const float* u_f32, float* out) {
    static_assert(PCC_WINO_UH_FLOATS * 4 == UG_BYTES, "image size");
    const size_t nu = (size_t)ngroups * ngroups * 48 * 64 * 4;
    float umax = 0.f;
    for (size_t i = 0; i < nu; ++i) {
        const float v = fabsf(u_f32[i]);
        if (v > umax && v <= 3.0e38f) umax = v;
    }
    int e = 0;
    float su = 1.f;
    if (umax > 0.f) {
        frexpf(umax, &e);                      // umax = f 2^e, f in [0.5, 1)
        int se = 14 - e;                       // su umax in [2^13, 2^14)
        se = se < -100 ? -100 : se > 100 ? 100 : se;
        su = ldexpf(1.f, se);
    }
    memset(out, 0, ((size_t)ngroups * ngroups * PCC_WINO_UH_FLOATS + PCC_WINO_UH_TAIL) * sizeof(float));
    unsigned short* o = reinterpret_cast<unsigned short*>(out);
    for (int cig = 0; cig < ngroups; ++cig)
        for (int cog = 0; cog < ngroups; ++cog)
            for (int row = 0; row < 48; ++row)
                for (int lane = 0; lane < 64; ++lane) {
                    const int dz = row / 16, py = (row / 4) % 4, px = row % 4, q = 3 * py + 2 - dz;
                    unsigned short* d = o + ((size_t)(cog * ngroups + cig) * UG_BYTES + (size_t)lane * UL_LANE_BYTES + (size_t)(q * 4 + px) * 16) / 2;
                    for (int c = 0; c < 4; ++c) {
                        const float x = u_f32[((((size_t)cig * ngroups + cog) * 48 + row) * 64 + lane) * 4 + c] * su;      // exact
                        d[c] = f16_bits(x);
                        d[4 + c] = f16_bits(x - f16_value(d[c]));                                                    // the difference is exact
                    }
                }
    out[(size_t)ngroups * ngroups * PCC_WINO_UH_FLOATS] = su;
}

bool pcc_wino_f16s_covers(const pcc_conv_desc* d) { return d->Cin == d->Cout && (d->Cin == 16 || d->Cin == 32 || d->Cin == 64); }

// 64 channels: two launches of the two-group kernel (cin groups 0, 1 -> raw partial sums in `out`; groups 2, 3 add them before the
// epilogue).  The partial sums of these layers are small (33 MB at 16^3 x 32): their read-modify-write costs nothing, and two resident U
// images per workgroup is what LDS holds.  Both launches scale block n by the same s (the same input tensor's maximum).
int pcc_conv_wino_f16s(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* uh_packed, const float* bias,
                       const float* residual, float* out, pcc_conv_ext* ext, hipStream_t st) {
    PCC_REQUIRE(pcc_wino_eligible(d) && pcc_wino_f16s_covers(d), "pcc_conv_wino_f16s: shape not covered");
    const int NG = d->Cin / 16, G = NG >= 2 ? 2 : 1;
    PCC_REQUIRE(NG == 1 || !(d->flags & PCC_CONV_CLIP01), "pcc_conv_wino_f16s: the multi-group kernels do not clip");
    WinoArgs a;
    a.in = in; a.bias = bias; a.res = residual; a.out = out;
    a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W;
    a.nty = d->H / 16; a.ntx = d->W / 16;
    a.ocs = d->out_cstride ? d->out_cstride : d->Cout;
    a.oco = d->out_coffset;
    a.nco = NG; a.ics = d->Cin; a.rcs = d->Cout; a.ico = 0; a.ncig = NG; a.ig0 = 0;
    a.u = uh_packed; a.flags = d->flags;
    a.utail = uh_packed + (size_t)NG * NG * PCC_WINO_UH_FLOATS;
    a.amax_out = ext ? ext->out_amax : nullptr;
    if (ext) ext->out_recorded = ext->out_amax != nullptr;
    if (ext && ext->in_amax) a.amax_in = ext->in_amax;
    else {
        unsigned* am = nullptr;
        { const int rc = pcc_ctx_amax(ctx, d->N * PCC_AMAX_SLOTS, &am); if (rc != PCC_OK) return rc; }
        { const int rc = pcc_block_amax(ctx, in, d->N, (size_t)d->D * d->H * d->W * d->Cin, am, st); if (rc != PCC_OK) return rc; }
        a.amax_in = am;
    }
    const int base = d->N * a.nty * a.ntx * NG;
    int zs = 1;
    const int min_zlen = NG >= 2 ? 4 : 8;
    while (base * zs < ctx->num_cu && d->D % (zs * 2) == 0 && d->D / (zs * 2) >= min_zlen) zs *= 2;
    a.zsplit = zs; a.zlen = d->D / zs;
    const int nwg = base * zs;
    typedef void (*kern_t)(WinoArgs, int);
    const bool relu = (d->flags & PCC_CONV_RELU) != 0;
    const int lds = f16s_lds_bytes(G);
    auto launch1 = [&](kern_t kern, const WinoArgs& aa) -> int {
        { const int rc = pcc_enable_big_lds((const void*)kern, lds); if (rc != PCC_OK) return rc; }
        hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(NT), lds, st, aa, nwg);
        PCC_CHECK_HIP(hipGetLastError());
        return PCC_OK;
    };
    if (NG == 1) {
        static const kern_t k1[4] = {conv16_wino_f16s_kernel<false, false, 1>, conv16_wino_f16s_kernel<true, false, 1>,
                                     conv16_wino_f16s_kernel<false, true, 1>, conv16_wino_f16s_kernel<true, true, 1>};
        return launch1(k1[(relu ? 1 : 0) + ((d->flags & PCC_CONV_CLIP01) ? 2 : 0)], a);
    }
    if (NG == 2) return launch1(relu ? (kern_t)conv16_wino_f16s_kernel<true, false, 2> : (kern_t)conv16_wino_f16s_kernel<false, false, 2>, a);
    // NG == 4: cin groups {0, 1} -> raw partial sums, then {2, 3} + partial sums -> bias (in the accumulators) / ReLU / residual
    WinoArgs p0 = a;
    p0.flags = 0; p0.bias = nullptr; p0.res = nullptr; p0.amax_out = nullptr;
    { const int rc = launch1((kern_t)conv16_wino_f16s_kernel<false, false, 2>, p0); if (rc != PCC_OK) return rc; }
    a.ig0 = 2; a.ico = 32;
    return launch1(relu ? (kern_t)conv16_wino_f16s_kernel<true, false, 2, true> : (kern_t)conv16_wino_f16s_kernel<false, false, 2, true>, a);
}
