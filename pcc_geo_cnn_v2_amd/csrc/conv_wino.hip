// Winograd F(2x2, 3x3) in (x, y) + direct 3 taps in z: 3-D convolution for gfx950 (CDNA4), fp32,
// Cin = Cout in {16, 32, 64} (as g x g sub-convolutions of 16 channels), k = 3, stride 1 -- Conv3D and, through
// host-flipped weights, Conv3DTranspose of /root/reference/src/model_transforms.py:62-70,93,107 (the k3 stride-1
// layers of AnalysisBlock / SynthesisBlock and the last analysis conv).
//
// Why: the k3 stride-1 layers are bound by the fp32 MFMA rate (157 TFLOP/s).  The minimal-filtering form
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
// from LDS and transformed, plane s+2 is fetched global->LDS directly (buffer_load ... lds), and the output plane
// completed by the dz=2 rows is reduced and stored.  One barrier per step.
#include <cstdlib>
#include "wino_common.h"

namespace pccwino {

// PRE: this launch handles a later cin group of a multi-group layer launched group by group (clip layers; PCC_WINO_PER_GROUP):
// the partial sums of the earlier groups are read back from `out` (same lane, same address as its own earlier store).
enum { M_ALL = 0, M_S0 = 1, M_S1 = 2, M_S1O = 3, M_FIN = 4 };

template <bool RELU, bool CLIP, bool PRE>
__global__ void __launch_bounds__(NT, 1) conv16_wino_kernel(WinoArgs a, int nwg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = lane & 15, g = lane >> 4;
    auto ldsr = [&](unsigned off) -> f32x4 { return *reinterpret_cast<const f32x4*>(smem + off); };
    auto ldsw = [&](unsigned off, const f32x4& v) { *reinterpret_cast<f32x4*>(smem + off) = v; };

    int wg = xcd_remap(blockIdx.x, nwg);
    const int cog = wg % a.nco; wg /= a.nco;       // cout group: neighbours in the grid share their input planes in L2
    const int tx_ = wg % a.ntx; wg /= a.ntx;
    const int ty_ = wg % a.nty; wg /= a.nty;
    const int zs = wg % a.zsplit;
    const int n = wg / a.zsplit;
    const int X0 = tx_ * 16, Y0 = ty_ * 16, zb = zs * a.zlen;
    const int nsteps = a.zlen + 2;             // input planes zb-1 .. zb+zlen
    const size_t HW = (size_t)a.H * a.W;
    const unsigned HWI = (unsigned)(HW * a.ics * 4), HWR = (unsigned)(HW * a.rcs * 4);      // bytes per z-plane
    const float* in_n = a.in + (size_t)n * a.D * HW * a.ics + a.ico;
    // plane-sized descriptors: z validity selects the descriptor (wave-uniform), y/x validity is the per-lane offset
    auto plane_rsrc = [&](int z) { const bool ok = (unsigned)z < (unsigned)a.D; return make_rsrc(in_n + (ok ? (size_t)z * HW * a.ics : 0), ok ? HWI : 0u); };

    // ---- U -> LDS (12 float4 per thread)
    {
        const __amdgpu_buffer_rsrc_t ru = make_rsrc(a.u + (size_t)cog * (U_BYTES / 4), (unsigned)U_BYTES);
        f32x4 tmp[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) tmp[i] = buf_load4(ru, (unsigned)(i * NT + tid) * 16u, 0);
#pragma unroll
        for (int i = 0; i < 12; ++i) ldsw((unsigned)(U_BASE + (i * NT + tid) * 16), tmp[i]);
    }

    // ---- staging: global -> LDS directly (buffer_load ... lds, no registers, no ds_write).  The LDS side of such a load
    //      is linear (M0 base + lane*16), so the XOR swizzle and the x de-interleave are applied on the GLOBAL side:
    //      lane L of chunk c fetches whatever belongs into LDS slot 64c + L.
    unsigned rel[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int slot = (wave * 5 + it) * 64 + lane;
        const int v = slot >> 2, c4 = (slot & 3) ^ ((v >> 1) & 3);
        const int yrow = v / 18, r = v - yrow * 18, par = r >= 9 ? 1 : 0, col = r - 9 * par, xi = 2 * col + par;
        const int y = Y0 - 1 + yrow, x = X0 - 1 + xi;
        const bool ok = v < PLANE_VOX && y >= 0 && y < a.H && x >= 0 && x < a.W;
        rel[it] = ok ? (unsigned)(((y * a.W + x) * a.ics + c4 * 4) * 4) : kOOB;   // OOB lanes write zeros (SAME padding / chunk padding)
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // (main loop) the plane addresses advance by one plane per step: 2 SALU adds instead of a 64-bit multiply per descriptor
    unsigned long long in_pl = (unsigned long long)in_n + (unsigned long long)(long long)(zb + 1) * HWI;   // plane s+2 of step s = 0
    auto stage_plane = [&](unsigned plane_off, int z) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rp = plane_rsrc(z);
#pragma unroll
        for (int it = 0; it < ITEMS; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lds_ptr)(smem + plane_off + (wave * 5 + it) * 1024), 16, (int)rel[it], 0, 0, 0);
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
    const unsigned HWO = (unsigned)(HW * a.ocs * 4);
    const bool has_res = (a.flags & PCC_CONV_ADD) != 0;
    const float* res_n = has_res ? a.res + (size_t)n * a.D * HW * a.rcs : a.in;
    float* out_n = a.out + (size_t)n * a.D * HW * a.ocs;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bias_l = (a.flags & PCC_CONV_BIAS) ? *reinterpret_cast<const f32x4*>(a.bias + 16 * cog + g * 4) : zero4;
    const f32x4 bias4 = bias_l;

    // Planes outside the volume are zero (SAME padding): a slab that starts at z = 0 skips step 0 (its plane z = -1 would only
    // deposit the bias) and a slab that ends at z = D runs its last step without matrix work (plane z = D adds nothing; the
    // step only reduces and stores output plane D - 1).  Both conditions are wave-uniform and decided OUTSIDE the MFMA stream.
    const bool first_zero = zb == 0, last_zero = zb + a.zlen == a.D;
    const int s0 = first_zero ? 1 : 0;                                   // first step that runs
    // ---- prologue: input planes s0, s0 + 1 -> their ring slots (plane s lives in slot s mod 3)
    stage_plane((unsigned)s0 * PLANE_BYTES, zb - 1 + s0);
    stage_plane((unsigned)(s0 + 1) * PLANE_BYTES, zb + s0);
    __syncthreads();

    f32x4 Vc[16];               // B^T d B of the current input plane (B operand of the MFMAs)
    f32x4 Vn[16];               // next plane: raw patch -> x-transformed; its y-transform is written row by row into Vc
    f32x4 Ub[2][4];
    f32x4 acc[3][16];           // three output planes in flight
    f32x4 S[2][2];              // A^T-reduced 2x2 outputs of the finished plane
    f32x4 resv[4], prev[4], ost[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) Vn[i] = ldsr(ra[i] + (unsigned)s0 * PLANE_BYTES);
    transform_x_rows(Vn, 0, 4);
    transform_y_row(Vc, Vn, 0);
    transform_y_row(Vc, Vn, 1);
    transform_y_row(Vc, Vn, 2);          // row 3 follows in slot 0 of the first step
    // the first step that runs starts at its first ACTIVE row: (dz 0, py 0) = slot 8 of step 0, (dz 1, py 0) = slot 4 of step 1; both use buffer 0
#pragma unroll
    for (int px = 0; px < 4; ++px) Ub[0][px] = ldsr(ua + (unsigned)((s0 * 16 + px) * 1024));
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][i] = zero4; acc[1][i] = zero4; acc[2][i] = zero4; }   // (planes finishing at s < 2 are never stored)

    unsigned long long res_pl = (unsigned long long)res_n + (unsigned long long)(long long)(zb - 2 + s0) * HWR;     // plane zo of step s0
    unsigned long long out_pl = (unsigned long long)out_n + (unsigned long long)(long long)(zb - 2 + s0) * HWO;
    in_pl += (unsigned long long)s0 * HWI;                                                                          // plane s0 + 2
    // one input plane: s = step index (input plane z = zb-1+s), PH = s mod 3
    // MODE: which of the 12 MFMA rows (dz = 2, 1, 0 x 4 point rows) a step runs.
    //   M_ALL  every interior step
    //   M_S0   step 0 (plane zb - 1 of a slab inside the volume): only its dz = 0 rows feed an output plane of this slab
    //   M_S1   step 1 behind M_S0: its dz = 2 rows would finish output plane zb - 1, which belongs to the slab below
    //   M_S1O  step 1 of a slab that starts at z = 0 (step 0 skipped): as M_S1, and its dz = 1 rows OPEN their accumulators
    //   M_FIN  last step of a slab that ends at z = D: no matrix work, only the reduction + store of output plane D - 1
    // (66 -> 65 plane-equivalents per 64-plane slab inside the volume, 66 -> 64.8 for a whole-volume slab)
    auto step = [&](auto ph_tag, int s, auto mode_tag) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_tag)::value;
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool FIN = MODE == M_FIN;
        constexpr unsigned slotN = (unsigned)((PH + 1) % 3) * PLANE_BYTES;   // plane s+1 (read)
        constexpr unsigned slotW = (unsigned)((PH + 2) % 3) * PLANE_BYTES;   // plane s+2 (written)
        constexpr int AF = PH;                                               // acc slot of the plane finished by dz = 2
        const bool zo_ok = s >= 2;                                           // the finished plane zo = zb - 2 + s exists
        // plane-sized descriptors of the finished output plane; zero-sized (loads return 0, stores are dropped) while s < 2
        const __amdgpu_buffer_rsrc_t rres = make_rsrc((const void*)(zo_ok ? res_pl : (unsigned long long)res_n), zo_ok && has_res ? HWR : 0u);
        const __amdgpu_buffer_rsrc_t rout = make_rsrc((const void*)(zo_ok ? out_pl : (unsigned long long)out_n), zo_ok ? HWO : 0u);
        res_pl += HWR; out_pl += HWO;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int dz = 2 - (j >> 2), py = j & 3;
            const int as = (PH + 2 - dz) % 3;
            const bool active = MODE == M_ALL || (MODE == M_S0 && dz == 0) || ((MODE == M_S1 || MODE == M_S1O) && dz <= 1);
            // (1) U fragments of the next row (wraps to the first active row of the next step: step 1 starts at its dz = 1 rows)
            if (active) {
                const int jn = (j == 11 && MODE == M_S0) ? 4 : (j + 1) % 12, dzn = 2 - (jn >> 2), pyn = jn & 3;
#pragma unroll
                for (int px = 0; px < 4; ++px) Ub[(j + 1) & 1][px] = ldsr(ua + (unsigned)(((dzn * 4 + pyn) * 4 + px) * 1024));
            }
            // (2) the 16 MFMAs of this row: 4 independent accumulators, k-chained; dz = 0 opens a new output plane
#pragma unroll
            for (int kk = 0; kk < (active ? 4 : 0); ++kk)
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    // a new output plane starts from 0, except point (1,1) which enters all four outputs with weight +1
                    // and therefore carries the bias for free
                    const bool opens = dz == 0 || (MODE == M_S1O && dz == 1);
                    const f32x4 c = (opens && kk == 0) ? ((py == 1 && px == 1) ? bias4 : zero4) : acc[as][py * 4 + px];
                    acc[as][py * 4 + px] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ub[j & 1][px][kk], Vc[py * 4 + px][kk], c, 0, 0, 0);
                }
            // An fp32 MFMA and a VALU op of the same wave share the SIMD's FMA lanes (tools/ubench/mfma_valu.hip): every
            // MFMA -> VALU -> MFMA sandwich costs ~10 cycles of pipeline turn-around on top of ~3.7 cycles per VALU op, so
            // the VALU work of a slot is issued as ONE block behind its 16 MFMAs instead of being interleaved with them.
            __builtin_amdgcn_sched_barrier(0);
            // (3) everything else, spread over the slots.  Vc row r is last read by the MFMAs of slot 8+r, so the rows of
            //     the next plane are written in slots 9, 10, 11 and (row 3) slot 0 of the next step: no register copies.
            if (FIN) {
            } else if (j == 0) transform_y_row(Vc, Vn, 3);
            else if (j == 1) {
                const bool ok = (unsigned)(zb + 1 + s) < (unsigned)a.D;
                const __amdgpu_buffer_rsrc_t rp = make_rsrc((const void*)(ok ? in_pl : (unsigned long long)in_n), ok ? HWI : 0u);
                in_pl += HWI;
#pragma unroll
                for (int it = 0; it < ITEMS; ++it)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lds_ptr)(smem + slotW + (wave * 5 + it) * 1024), 16, (int)rel[it], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 16; ++i) Vn[i] = ldsr(ra[i] + slotN);
            } else if (j == 2) transform_x_rows(Vn, 0, 2);
            else if (j == 3) transform_x_rows(Vn, 2, 4);
            if (j == 4 || j == 5) {
                // residual (and partial sums of the previous cin groups, accumulated in place in `out`: same lane, same
                // address) of the two voxels of output row oy = j - 4.  (Requesting them in slot 0 instead was measured: no
                // change for the one-group layers, 8 % more cycles for the two-group ones.)
#pragma unroll
                for (int q = 2 * (j - 4); q < 2 * (j - 4) + 2; ++q) {
                    resv[q] = buf_load4(rres, rvo[q], 0);
                    if (PRE) prev[q] = buf_load4(rout, ovo[q], 0);
                }
            }
            if (j >= 5 && j <= 8) {
                // A^T along x on row r of the finished plane, accumulate A^T along y
                const int r = j - 5;
                const f32x4 m0 = acc_read(acc[AF][r * 4 + 0]), m1 = acc_read(acc[AF][r * 4 + 1]), m2 = acc_read(acc[AF][r * 4 + 2]), m3 = acc_read(acc[AF][r * 4 + 3]);
                const f32x4 r0 = add4(add4(m0, m1), m2), r1 = sub4(sub4(m1, m2), m3);
                if (r == 0) { S[0][0] = r0; S[0][1] = r1; }
                else if (r == 1) { S[0][0] = add4(S[0][0], r0); S[0][1] = add4(S[0][1], r1); S[1][0] = r0; S[1][1] = r1; }
                else if (r == 2) { S[0][0] = add4(S[0][0], r0); S[0][1] = add4(S[0][1], r1); S[1][0] = sub4(S[1][0], r0); S[1][1] = sub4(S[1][1], r1); }
                else { S[1][0] = sub4(S[1][0], r0); S[1][1] = sub4(S[1][1], r1); }
            }
            if (j == 8 || j == 9) {
                // epilogue of output row oy = j - 8 (complete after reduction row 2 resp. 3): ReLU, residual, clip, float4 stores
                // (the bias is already inside, see the dz = 0 rows)
#pragma unroll
                for (int q = 2 * (j - 8); q < 2 * (j - 8) + 2; ++q) {
                    f32x4 o = S[q >> 1][q & 1];
                    if (PRE) o = add4(o, prev[q]);
                    // one v_maximum3_f32 per element: fmaxf on the result of the inline-asm packed add costs a second v_max (NaN quieting)
                    if (RELU) o = __builtin_elementwise_maximum(o, zero4);
                    o = add4(o, resv[q]);     // zeros without PCC_CONV_ADD (zero-sized buffer)
                    if (CLIP) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] = fminf(fmaxf(o[c], 0.f), 1.f);
                    }
                    ost[q] = o;
                }
                // gfx950: a buffer_store_dwordx4 reads its data registers late; a VALU write to them in the next issue
                // slots corrupts the last dword of the last lanes (seen as out[q].w <- out[q+1].w).  The compiler only
                // guards the immediate-soffset form with one wait state, so the data registers are kept live (and
                // therefore unwritten) until the next slot.
#pragma unroll
                for (int q = 2 * (j - 8); q < 2 * (j - 8) + 2; ++q) buf_store4(rout, ost[q], ovo[q], 0);
            }
            if (j == 9 || j == 10) {
#pragma unroll
                for (int q = 2 * (j - 9); q < 2 * (j - 9) + 2; ++q) asm volatile("" ::"v"(ost[q]));
            }
            if (FIN) {
            } else if (j == 9) transform_y_row(Vc, Vn, 0);
            else if (j == 10) transform_y_row(Vc, Vn, 1);
            else if (j == 11) transform_y_row(Vc, Vn, 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        // The LDS-direct loads of plane s+2 (slot 1) must have landed before the barrier publishes them to the other
        // waves; the compiler only orders them against this wave's own LDS reads.  vmcnt counts in issue order: the 4
        // residual (+4 partial-sum) loads of slots 4 / 5 and the 4 stores of slots 8 / 9 were issued later and may stay in flight.
        if (!FIN) {
            __builtin_amdgcn_s_waitcnt(PRE ? 0x0F7C : 0x0F78);      // vmcnt(12 / 8) expcnt(7) lgkmcnt(15)
            __syncthreads();     // plane s+2 is published; nobody still reads plane s+1
        }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using P2 = std::integral_constant<int, 2>;
    using MAll = std::integral_constant<int, M_ALL>;
    using MFin = std::integral_constant<int, M_FIN>;

    if (first_zero) step(P1{}, 1, std::integral_constant<int, M_S1O>{});
    else {
        step(P0{}, 0, std::integral_constant<int, M_S0>{});
        step(P1{}, 1, std::integral_constant<int, M_S1>{});
    }
    const int nloop = nsteps - (last_zero ? 1 : 0);
    for (int s = 2; s < nloop; s += 3) {
        step(P2{}, s, MAll{});
        if (s + 1 < nloop) step(P0{}, s + 1, MAll{});
        if (s + 2 < nloop) step(P1{}, s + 2, MAll{});
    }
    if (last_zero) {
        const int sl = nsteps - 1, ph = sl % 3;
        if (ph == 0) step(P0{}, sl, MFin{});
        else if (ph == 1) step(P1{}, sl, MFin{});
        else step(P2{}, sl, MFin{});
    }
}



// ---------------------------------------------------------------------------------------------------------------------
// Cin = Cout = 16 G, G in {2, 4}: the cin groups INSIDE the z march (round 3).
//
// conv16_wino_kernel handles one 16-channel cin group per march; a G-group layer then needs G marches whose partial sums
// travel through `out` (written, re-read and re-written: 32 -> 32 @32^3 ran at 0.60-0.63 of the MFMA peak, 64 -> 64 @16^3 at
// 0.53, against 0.68-0.74 for the one-group layer).  Here the accumulators of the three output planes in flight (192
// AccVGPRs, one cout group) stay live across the cin groups: the march is a linear sequence of MICRO-STEPS m = (input
// plane s, cin group c), c fastest.  Micro-step m multiplies V(tile m) with U[c]; meanwhile tile m+1 is read from LDS and
// transformed, tile m+2 arrives global -> LDS.  Only the LAST cin group of a plane reduces / stores the finished output
// plane (A^T . A, bias, ReLU, residual); the other G-1 micro-steps carry the input transform as their only VALU work.
// No partial sum ever leaves the registers, one launch per layer, no read-modify-write of `out`.
//
// LDS (162816 B): ring of 3 tiles (18 x 18 x 16 ch; a tile is dead once its patches are in registers, so the ring is
// indexed by m, not by (s, c)) + two U buffers of 48 KB.  G = 2: both cin groups' U stay resident (the second one streams in
// during micro-step 0).  G = 4: U[c(m+1)] streams global(L2) -> LDS into the idle buffer during micro-step m, 12 pieces of
// 1 KB per wave in slots 0, 2..4 (48 KB per ~7k cycles and CU = 7 B/cycle, an eighth of the L2 rate).
// The tile ring phase (m mod 3) is independent of the accumulator phase (s mod 3, compile time): the 16 patch addresses
// advance by a wave-uniform delta per micro-step (16 v_add_u32) instead of being immediates.
//
// Summation order: per output element cin group 0's taps, then group 1's, ... in ONE fp32 accumulator chain, then A^T . A.
// (The per-group path reduces every group's accumulator separately and adds the reduced values: same tolerance against the
// oracle, not the same bits.)  Fixed order, independent of launch geometry: bit-deterministic.
template <bool RELU, int G>
__global__ void __launch_bounds__(NT, 1) conv16_wino_cin_kernel(WinoArgs a, int nwg) {
    static_assert(G == 2 || G == 4, "the two U buffers alternate with the cin group: G must be even");
    constexpr bool STREAM = G > 2;           // G = 2: both U resident from the prologue on; G = 4: U[c(m+1)] streams during m
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = lane & 15, g = lane >> 4;
    auto ldsr = [&](unsigned off) -> f32x4 { return *reinterpret_cast<const f32x4*>(smem + off); };

    int wg = xcd_remap(blockIdx.x, nwg);
    const int cog = wg % a.nco; wg /= a.nco;
    const int tx_ = wg % a.ntx; wg /= a.ntx;
    const int ty_ = wg % a.nty; wg /= a.nty;
    const int zs = wg % a.zsplit;
    const int n = wg / a.zsplit;
    const int X0 = tx_ * 16, Y0 = ty_ * 16, zb = zs * a.zlen;
    const int nsteps = a.zlen + 2;
    const size_t HW = (size_t)a.H * a.W;
    const unsigned HWI = (unsigned)(HW * a.ics * 4), HWR = (unsigned)(HW * a.rcs * 4), HWO = (unsigned)(HW * a.ocs * 4);
    const float* in_n = a.in + (size_t)n * a.D * HW * a.ics;
    typedef __attribute__((address_space(3))) void* lds_ptr;

    // ---- tile staging (global -> LDS directly; swizzle and x de-interleave on the global side, see conv16_wino_kernel)
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
    // tile of micro-step (sp, cg): input plane z = zb - 1 + sp, channels 16 cg .. 16 cg + 15; `addr` = address of that
    // plane's channel 16 cg (kept incrementally by the caller: no 64-bit multiply in the loop)
    auto stage_tile = [&](unsigned ring_off, int sp, int cg, unsigned long long addr) __attribute__((always_inline)) {
        const bool ok = (unsigned)(zb - 1 + sp) < (unsigned)a.D && sp < nsteps;
        const __amdgpu_buffer_rsrc_t rp = make_rsrc((const void*)(ok ? addr : (unsigned long long)in_n), ok ? HWI - (unsigned)(64 * cg) : 0u);
#pragma unroll
        for (int it = 0; it < ITEMS; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lds_ptr)(smem + ring_off + (wave * 5 + it) * 1024), 16, (int)rel[it], 0, 0, 0);
    };
    // one 1 KB piece of U[cg] (this cout group) -> U buffer at `ubase`; piece index k = 0 .. 11 of this wave
    auto stage_u = [&](unsigned ubase, int cg, int k) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t ru = make_rsrc(a.u + ((size_t)cg * a.nco + cog) * (U_BYTES / 4), (unsigned)U_BYTES);
        const int chunk = wave * 12 + k;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr)(smem + ubase + chunk * 1024), 16, (int)(lane * 16), chunk * 1024, 0, 0);
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
            ra[dy * 4 + dx] = (unsigned)(v * 64 + ((g ^ ((v >> 1) & 3)) << 4));
        }
    constexpr unsigned UTOG = (unsigned)(U_BASE ^ (U_BASE + U_BYTES));
    unsigned ua = (unsigned)(U_BASE + lane * 16);          // toggles between the two U buffers every micro-step

    // ---- epilogue addressing
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
    const f32x4 bias_l = (a.flags & PCC_CONV_BIAS) ? *reinterpret_cast<const f32x4*>(a.bias + 16 * cog + g * 4) : zero4;

    // A slab that starts at z = 0 does not march the padding plane below it: it starts at plane 1, whose dz = 1 rows then open
    // their accumulators (wave-uniform, decided here, outside the MFMA stream).
    const bool first_zero = zb == 0;
    const int s0 = first_zero ? 1 : 0;
    // ---- prologue: tiles 0, 1 (plane s0, groups 0, 1) -> ring slots 0, 1; U[0] -> buffer 0 (G = 2: and U[1] -> buffer 1)
    unsigned long long tile_pl = (unsigned long long)in_n + (unsigned long long)(long long)(zb - 1 + s0) * HWI;    // tile (plane s0, group 0)
    stage_tile(0, s0, 0, tile_pl);
    stage_tile(PLANE_BYTES, s0, 1, tile_pl + 64);
    tile_pl += 128;                                                                                          // tile m = 2
    if (G == 2) tile_pl += HWI - 128;
#pragma unroll
    for (int k = 0; k < 12; ++k) stage_u((unsigned)U_BASE, 0, k);
    if (!STREAM) {
#pragma unroll
        for (int k = 0; k < 12; ++k) stage_u((unsigned)(U_BASE + U_BYTES), 1, k);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
    __syncthreads();

    f32x4 Vc[16], Vn[16], Ub[4], acc[3][16], S[2][2], resv[4], ost[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) Vn[i] = ldsr(ra[i]);
    transform_x_rows(Vn, 0, 4);
    transform_y_row(Vc, Vn, 0);
    transform_y_row(Vc, Vn, 1);
    transform_y_row(Vc, Vn, 2);
#pragma unroll
    for (int px = 0; px < 4; ++px) Ub[px] = ldsr(ua + (unsigned)((s0 * 16 + px) * 1024));      // first active row: (dz 0, py 0) of plane 0, (dz 1, py 0) of plane 1
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][i] = zero4; acc[1][i] = zero4; acc[2][i] = zero4; }
#pragma unroll
    for (int i = 0; i < 16; ++i) ra[i] += (unsigned)PLANE_BYTES;      // -> the slot of tile 1

    // wave-uniform march state
    int c = 0, s = s0;                       // cin group and input plane of the current micro-step m
    int s2 = s0 + 2 / G, c2 = 2 % G;         // plane / cin group of tile m + 2 (its address: tile_pl)
    unsigned wr_off = 2u * PLANE_BYTES;      // ring slot of tile m + 2
    int rd_slot = 1;                         // ring slot of tile m + 1 (what ra[] points into)
    unsigned long long res_pl = (unsigned long long)res_n + (unsigned long long)(long long)(zb - 2 + s0) * HWR;     // plane zo of plane s0
    unsigned long long out_pl = (unsigned long long)out_n + (unsigned long long)(long long)(zb - 2 + s0) * HWO;

    // one micro-step.  PH = s mod 3 (accumulator rotation), FIRST / FIN = first / last cin group of the plane: all compile
    // time -- wave-uniform run-time branches around the slot pieces were measured at ~1000 cycles per micro-step (12 % of it:
    // every taken branch restarts the instruction fetch), more than the epilogue they skip.
    // The two head planes of a slab run only the MFMA rows that feed ITS output planes (ROWS below): plane 0 (zb - 1) its dz = 0 rows,
    // plane 1 its dz = 1 and dz = 0 rows (the rest of the step -- staging, input transform -- is unchanged): 10 -> 9 plane-equivalents
    // per 8-plane slab (64 -> 64 @16^3), 34 -> 33 (32 -> 32 @32^3).  The mirror image for the plane behind the slab (ROWS 3 / 5) is a
    // tail peeled off the loop, one variant per accumulator phase; earlier attempts that kept it inside the loop structure made the
    // allocator spill 90 - 780 registers, and a run-time row mask cost the 32-channel layer more (+7 % wave cycles, 12 scalar branches
    // per micro-step) than the plane saves.
    auto step = [&](auto ph_tag, auto first_tag, auto fin_tag, auto rows_tag) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_tag)::value;
        constexpr int AF = PH;
        constexpr bool first = decltype(first_tag)::value, fin = decltype(fin_tag)::value;
        // ROWS: 0 = all 12 MFMA rows; 1 = plane 0 of the slab (zb - 1): dz = 0 rows only; 2 = plane 1: dz = 1 and dz = 0 rows (its
        // dz = 2 rows would finish output plane zb - 1, which belongs to the slab below and is never stored); 3 = the plane behind a
        // slab that ends at z = D (zero padding): no matrix work, no input work -- only its LAST micro-step runs, to reduce and store
        // output plane D - 1
        constexpr int ROWS = decltype(rows_tag)::value;
        // 4 = plane 1 of a slab that starts at z = 0 (plane 0 skipped): as 2, and its dz = 1 rows OPEN their accumulators
        // 5 = the plane behind a slab INSIDE the volume (zb + zlen): only its dz = 2 rows feed this slab (output plane zb + zlen - 1)
        constexpr int J_OWN = ROWS == 1 ? 8 : (ROWS == 2 || ROWS == 4) ? 4 : 0;  // first active row of this plane's micro-steps
        constexpr int J_FIRST_NEXT = !fin ? J_OWN : ROWS == 1 ? 4 : 0;           // ... of the NEXT micro-step (plane 0 -> plane 1 -> full planes)
        const bool zo_ok = s >= 2;
        const __amdgpu_buffer_rsrc_t rres = make_rsrc((const void*)(zo_ok ? res_pl : (unsigned long long)res_n), zo_ok && has_res && fin ? HWR : 0u);
        const __amdgpu_buffer_rsrc_t rout = make_rsrc((const void*)(zo_ok ? out_pl : (unsigned long long)out_n), zo_ok && fin ? HWO : 0u);
        const int cn = c + 1 == G ? 0 : c + 1;                        // cin group of micro-step m + 1
        // STREAM: U[cn] -> the idle buffer during this micro-step (after the last micro-step: a harmless extra copy of U[0]);
        // its first row can only be read after the closing barrier.  Otherwise it is resident and prefetched across the barrier.
        const unsigned ub_next = (ua ^ UTOG) - (unsigned)(lane * 16);
        const int rd_delta = rd_slot == 2 ? -2 * PLANE_BYTES : PLANE_BYTES;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int dz = 2 - (j >> 2), py = j & 3;
            const int as = (PH + 2 - dz) % 3;
            // (1) + (2) the 16 MFMAs of this row in two halves of two points (px) each, k-chained and alternating between the two
            //     points (a dependent MFMA is two issue slots away: no stall on the 40-cycle latency).  The U fragments of a half
            //     are dead after its 8 MFMAs: the SAME registers receive the fragments of the next row right behind them, so every
            //     LDS read has >= 8 MFMAs (256 cycles) of cover in every micro-step -- most micro-steps have no VALU block behind the
            //     MFMAs of slots 4..8 that could cover it (in the one-group kernel every slot has one) -- at 16 registers for U.
            //     The dz = 0 rows open a new output plane in the FIRST cin group (from 0; point (1,1) enters all four outputs
            //     with weight +1 and carries the bias) and continue it in the others.
            const bool active = ROWS == 0 || (ROWS == 1 && dz == 0) || ((ROWS == 2 || ROWS == 4) && dz <= 1) || (ROWS == 5 && dz == 2);
            constexpr int J_LAST = ROWS == 5 ? 3 : 11;       // last active row: its U prefetch is the next micro-step's first row
            const int jn = j == J_LAST ? J_FIRST_NEXT : j + 1, dzn = 2 - (jn >> 2), pyn = jn & 3;
            const unsigned urow = (j == J_LAST ? (ua ^ UTOG) : ua) + (unsigned)((dzn * 4 + pyn) * 4 * 1024);
            const bool INIT = first && (dz == 0 || (ROWS == 4 && dz == 1));      // (folds: first is a constant, dz follows from the unrolled j)
            auto half = [&](auto h_tag) __attribute__((always_inline)) {
                constexpr int PX0 = decltype(h_tag)::value * 2;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int px = PX0; px < PX0 + 2; ++px) {
                        const f32x4 cc = (INIT && kk == 0) ? ((py == 1 && px == 1) ? bias_l : zero4) : acc[as][py * 4 + px];
                        acc[as][py * 4 + px] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ub[px][kk], Vc[py * 4 + px][kk], cc, 0, 0, 0);
                    }
            };
            auto next_u = [&](int px0) __attribute__((always_inline)) {
                if (j < J_LAST || !STREAM) {
                    Ub[px0] = ldsr(urow + (unsigned)(px0 * 1024));
                    Ub[px0 + 1] = ldsr(urow + (unsigned)((px0 + 1) * 1024));
                }
            };
            if (active) {
                half(std::integral_constant<int, 0>{});
                next_u(0);
                half(std::integral_constant<int, 1>{});
                next_u(2);
                if (j < J_LAST || !STREAM) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // (3) everything else, as one block behind the MFMAs of the slot
            if (ROWS == 3) {
            } else if (j == 0) transform_y_row(Vc, Vn, 3);
            else if (j == 1) {
                stage_tile(wr_off, s2, c2, tile_pl);
#pragma unroll
                for (int i = 0; i < 16; ++i) Vn[i] = ldsr(ra[i]);
            } else if (j == 2) transform_x_rows(Vn, 0, 2);
            else if (j == 3) transform_x_rows(Vn, 2, 4);
            if (STREAM && ROWS != 3) {        // U[cn] -> the idle buffer: pieces 0..11 of this wave in slots 0, 2..4
                // (all 12 pieces are on their way by slot 4: the last ones then have 8 slots to land before the closing barrier waits for them)
                if (j == 0) { stage_u(ub_next, cn, 0); stage_u(ub_next, cn, 1); stage_u(ub_next, cn, 2); }
                else if (j >= 2 && j <= 4) { stage_u(ub_next, cn, 3 * j - 3); stage_u(ub_next, cn, 3 * j - 2); stage_u(ub_next, cn, 3 * j - 1); }
            }
            if (fin) {
                if (j == 4 || j == 5) {
#pragma unroll
                    for (int q = 2 * (j - 4); q < 2 * (j - 4) + 2; ++q) resv[q] = buf_load4(rres, rvo[q], 0);
                }
                if (j >= 5 && j <= 8) {
                    const int r = j - 5;
                    const f32x4 m0 = acc_read(acc[AF][r * 4 + 0]), m1 = acc_read(acc[AF][r * 4 + 1]), m2 = acc_read(acc[AF][r * 4 + 2]), m3 = acc_read(acc[AF][r * 4 + 3]);
                    const f32x4 r0 = add4(add4(m0, m1), m2), r1 = sub4(sub4(m1, m2), m3);
                    if (r == 0) { S[0][0] = r0; S[0][1] = r1; }
                    else if (r == 1) { S[0][0] = add4(S[0][0], r0); S[0][1] = add4(S[0][1], r1); S[1][0] = r0; S[1][1] = r1; }
                    else if (r == 2) { S[0][0] = add4(S[0][0], r0); S[0][1] = add4(S[0][1], r1); S[1][0] = sub4(S[1][0], r0); S[1][1] = sub4(S[1][1], r1); }
                    else { S[1][0] = sub4(S[1][0], r0); S[1][1] = sub4(S[1][1], r1); }
                }
                if (j == 8 || j == 9) {
#pragma unroll
                    for (int q = 2 * (j - 8); q < 2 * (j - 8) + 2; ++q) {
                        f32x4 o = S[q >> 1][q & 1];
                        if (RELU) o = __builtin_elementwise_maximum(o, zero4);
                        ost[q] = add4(o, resv[q]);     // zeros without PCC_CONV_ADD (zero-sized buffer)
                    }
#pragma unroll
                    for (int q = 2 * (j - 8); q < 2 * (j - 8) + 2; ++q) buf_store4(rout, ost[q], ovo[q], 0);
                }
                if (j == 9 || j == 10) {      // store data registers stay unwritten for one more slot (see conv16_wino_kernel)
#pragma unroll
                    for (int q = 2 * (j - 9); q < 2 * (j - 9) + 2; ++q) asm volatile("" ::"v"(ost[q]));
                }
            }
            if (ROWS == 3) {
            } else if (j == 9) transform_y_row(Vc, Vn, 0);
            else if (j == 10) {
                transform_y_row(Vc, Vn, 1);
#pragma unroll
                for (int i = 0; i < 8; ++i) ra[i] += (unsigned)rd_delta;
            } else if (j == 11) {
                transform_y_row(Vc, Vn, 2);
#pragma unroll
                for (int i = 8; i < 16; ++i) ra[i] += (unsigned)rd_delta;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // tile m+2 (slot 1) and the U pieces (slots 0..7) must have landed before the barrier publishes them; the only younger
        // memory operations are the four stores of a final micro-step
        if (ROWS == 3) return;                            // (the last thing this workgroup does)
        if (fin) __builtin_amdgcn_s_waitcnt(0x0F74);      // vmcnt(4)
        else __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
        __syncthreads();
        // ---- advance (all wave-uniform)
        ua ^= UTOG;
        // opaque: for G = 2 the toggle is periodic in the loop body and the optimiser would otherwise hoist all 96 row addresses
        // (base + offset beyond the 16-bit DS immediate) out of the loop into spilled registers
        asm volatile("" : "+v"(ua));
        if (STREAM) {
            constexpr int dzf = 2 - (J_FIRST_NEXT >> 2);
#pragma unroll
            for (int px = 0; px < 4; ++px) Ub[px] = ldsr(ua + (unsigned)((dzf * 16 + px) * 1024));
        }
        rd_slot = rd_slot == 2 ? 0 : rd_slot + 1;
        wr_off = wr_off == 2u * PLANE_BYTES ? 0u : wr_off + (unsigned)PLANE_BYTES;
        tile_pl += 64;
        if (++c2 == G) { c2 = 0; ++s2; tile_pl += HWI - 64 * G; }
        if (++c == G) { c = 0; ++s; res_pl += HWR; out_pl += HWO; }
    };

    // one input plane = G micro-steps: first, (G - 2 middle ones: one body, looped), last
    auto plane = [&](auto ph_tag, auto rows_tag) __attribute__((always_inline)) {
        step(ph_tag, std::true_type{}, std::false_type{}, rows_tag);
        if (G > 2) {
#pragma nounroll
            for (int k = 0; k < G - 2; ++k) step(ph_tag, std::false_type{}, std::false_type{}, rows_tag);
        }
        step(ph_tag, std::false_type{}, std::true_type{}, rows_tag);
    };
    using R0 = std::integral_constant<int, 0>;
    if (first_zero) plane(std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{});      // plane s = 1 opens everything
    else {
        plane(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});      // plane s = 0 (z = zb - 1): dz = 0 rows only
        plane(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});      // plane s = 1: dz = 1, 0 rows
    }
    // The plane behind the slab is peeled off the loop: behind the volume (z = D, zero padding) only the reduction + store of output
    // plane D - 1 is left of it; inside the volume only its dz = 2 rows feed this slab.  Wave-uniform, decided outside the MFMA stream.
    const bool last_zero = zb + a.zlen == a.D;
    const int nloop = nsteps - 1;
#pragma nounroll
    for (int sp = 2; sp < nloop; sp += 3) {                                          // planes 2 .. nloop - 1, phase = s mod 3
        plane(std::integral_constant<int, 2>{}, R0{});
        if (sp + 1 < nloop) plane(std::integral_constant<int, 0>{}, R0{});
        if (sp + 2 < nloop) plane(std::integral_constant<int, 1>{}, R0{});
    }
    using R3 = std::integral_constant<int, 3>;
    using R5 = std::integral_constant<int, 5>;
    const int ph = (nsteps - 1) % 3;
    if (last_zero) {
        if (ph == 0) step(std::integral_constant<int, 0>{}, std::false_type{}, std::true_type{}, R3{});
        else if (ph == 1) step(std::integral_constant<int, 1>{}, std::false_type{}, std::true_type{}, R3{});
        else step(std::integral_constant<int, 2>{}, std::false_type{}, std::true_type{}, R3{});
    } else {
        if (ph == 0) plane(std::integral_constant<int, 0>{}, R5{});
        else if (ph == 1) plane(std::integral_constant<int, 1>{}, R5{});
        else plane(std::integral_constant<int, 2>{}, R5{});
    }
}

}  // namespace pccwino

using namespace pccwino;

bool pcc_wino_eligible(const pcc_conv_desc* d) {
    if (!pcc_wino_channels(d->Cin, d->Cout) || d->k != 3 || d->stride != 1) return false;
    if (d->W % 16 || d->H % 16) return false;
    const int ocs = d->out_cstride ? d->out_cstride : d->Cout;
    if (ocs % 4 || d->out_coffset % 4) return false;
    const double hw = (double)d->H * d->W;
    if (hw * d->Cin * 4.0 >= 2147483648.0 || hw * ocs * 4.0 >= 2147483648.0) return false;     // one z-plane per buffer descriptor
    return true;
}

// Cin = Cout = 16 g: g x g sub-convolutions of 16 -> 16 channels.  One launch per cin group covers all cout groups
// (grid dimension); the partial sums of the earlier cin groups are accumulated in place in `out` (template PRE), bias /
// ReLU / residual / clip are applied by the last launch only.
int pcc_conv_wino(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* u_packed, const float* bias,
                  const float* residual, float* out, hipStream_t st) {
    PCC_REQUIRE(pcc_wino_eligible(d), "pcc_conv_wino: shape not covered");
    WinoArgs a;
    a.in = in; a.bias = bias; a.res = residual; a.out = out;
    a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W;
    a.nty = d->H / 16; a.ntx = d->W / 16;
    a.ocs = d->out_cstride ? d->out_cstride : d->Cout;
    a.oco = d->out_coffset;
    const int G = d->Cin / 16;
    a.nco = G; a.ics = d->Cin; a.rcs = d->Cout;
    // split z so that every CU gets a workgroup (each split pays the 48 KB U load and two halo planes)
    const int base = d->N * a.nty * a.ntx * G;
    int zs = 1;
    const int min_zlen = G >= 2 ? 4 : 8;      // (multi-group layers: 32 -> 32 @16^3 x 32 gets 256 workgroups of 4 planes: 41 us; 128 of 8: 69 us)
    while (base * zs < ctx->num_cu && d->D % (zs * 2) == 0 && d->D / (zs * 2) >= min_zlen) zs *= 2;
    a.zsplit = zs; a.zlen = d->D / zs;
    const int nwg = base * zs;
    typedef void (*kern_t)(WinoArgs, int);
    static const kern_t kerns[8] = {conv16_wino_kernel<false, false, false>, conv16_wino_kernel<true, false, false>,
                                    conv16_wino_kernel<false, true, false>,  conv16_wino_kernel<true, true, false>,
                                    conv16_wino_kernel<false, false, true>,  conv16_wino_kernel<true, false, true>,
                                    conv16_wino_kernel<false, true, true>,   conv16_wino_kernel<true, true, true>};
    // multi-group layers take conv16_wino_cin_kernel unless they clip or PCC_WINO_PER_GROUP asks for one launch per cin group
    // (A/B runs; a test compares the two within the tolerance).  Measured (round 3, batch 32): 32 -> 32 @32^3 276 -> 251 us,
    // 64 -> 64 @16^3 172 -> 159 us.
    const bool per_group = ctx->num(PCC_NUM_WINO_PER_GROUP);
    // G = 2, 4: the cin groups inside the march, accumulators live across them (conv16_wino_cin_kernel), one launch
    if ((G == 2 || G == 4) && !(d->flags & PCC_CONV_CLIP01) && !per_group) {
        static const kern_t ck[4] = {conv16_wino_cin_kernel<false, 2>, conv16_wino_cin_kernel<true, 2>,
                                     conv16_wino_cin_kernel<false, 4>, conv16_wino_cin_kernel<true, 4>};
        a.u = u_packed; a.ico = 0; a.ncig = G; a.flags = d->flags;
        const kern_t ckern = ck[(G == 4 ? 2 : 0) + ((d->flags & PCC_CONV_RELU) ? 1 : 0)];
        { const int rc = pcc_enable_big_lds((const void*)ckern, LDS_BYTES_CIN); if (rc != PCC_OK) return rc; }
        hipLaunchKernelGGL(ckern, dim3((unsigned)nwg), dim3(NT), LDS_BYTES_CIN, st, a, nwg);
        PCC_CHECK_HIP(hipGetLastError());
        return PCC_OK;
    }
    a.ncig = 1;
    for (int ci = 0; ci < G; ++ci) {
        const bool last = ci == G - 1;
        a.u = u_packed + (size_t)ci * G * (U_BYTES / 4);
        a.ico = 16 * ci;
        a.flags = last ? d->flags : 0;
        const bool relu = last && (d->flags & PCC_CONV_RELU), clip = last && (d->flags & PCC_CONV_CLIP01);
        const kern_t kern = kerns[(relu ? 1 : 0) + (clip ? 2 : 0) + (ci > 0 ? 4 : 0)];
        { const int rc = pcc_enable_big_lds((const void*)kern, LDS_BYTES); if (rc != PCC_OK) return rc; }
        hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(NT), LDS_BYTES, st, a, nwg);
        PCC_CHECK_HIP(hipGetLastError());
    }
    return PCC_OK;
}
