// Conv3DTranspose, k = 3, stride 2, 32 -> 16 and 64 -> 32, marching along z, for the fp16 MODE (PCC_CONV_F16 | PCC_CONV_OUT16: BASELINE.json
// configs[4], never the headline precision) -- round 5.
//
// In that mode the first layer of a SynthesisBlock (/root/reference/src/model_transforms.py:78 inside :126-137) reads the fp32 output of the
// previous block and hands over in fp16 to the fp16-storage kernels (conv_f16.hip).  It ran on conv_tr2g_kernel<.., F16, EPI_F16> -- the
// tiled fp32 structure with the fragments rounded to fp16 at the matrix instruction: 251 us (32 -> 16 @64^3 -> 128^3 x 8) and 149 us
// (64 -> 32) per launch, 24 % of a configs[4] step, against an HBM floor of 128 / 64 us.  Here: the marching structure of
// conv_tr2m_bf16.hip with the operands it wants in this mode:
//   * v_mfma_f32_16x16x32_f16: K = 32 input channels per instruction, so a tap of the 32 -> 16 layer is ONE MFMA per row of 16 voxels
//     (the tiled kernel: two K = 16 ones after eight cvts; the split-bf16 march of the fp32 mode: six K-stacked ones);
//   * the tile of the next micro-step travels global (fp32) -> registers -> v_cvt_pk_f16_f32 (round to nearest even, the rounding
//     oracle/torch_oracle.run_transform_fp16 applies to the operands) -> LDS, 64 B per voxel at a 96-byte pitch (6 bank quads: the
//     8 + 8 voxels of a ds_read_b128 lane group fall on 16 different ones, brute-forced over the groups of MI355X_MICROARCH.md);
//     TWO tile buffers (30 KB each), so the overwrite needs no barrier of its own: one barrier per micro-step;
//   * the fp16 weights of the cout tile (27 fragments of 1 KB per 32-channel cin group) are converted from the tr2g-order fp32 image in
//     the prologue and stay LDS-resident; a 4-deep fragment ring (a tap is 4 MFMAs = 64 cycles) covers the LDS latency;
//   * parity decomposition, three accumulator sets (192 AccVGPRs), the finished planes leaving under the first taps of the next plane,
//     compile-time plane parity: conv_tr2m_kernel's (DESIGN_HISTORY.md 3.3b).  The epilogue rounds to fp16 and stores 8 bytes per lane.
// Summation order per output element: cin group of 32 -> tap -> the K = 32 chain of the instruction, fp32 accumulation, fixed:
// bit-deterministic, independent of batch and z split; NOT the order of conv_tr2g_kernel<F16>, so the dispatch is a function of the layer
// shape only (encoder and decoder must produce the same bits).
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

#include "common.h"

namespace pcctr2mh {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma_f16(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}
// 8 fp32 -> 8 fp16, round to nearest even
__device__ __forceinline__ u32x4 to_h8(const f32x4& lo, const f32x4& hi) {
    const f32x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(u32x4, __builtin_convertvector(v, h16x8));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
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
constexpr int TILE_SLOTS = LXY * LXY * 4;               // (voxel, 8-channel quad) items of one (plane, 32-channel cin group) tile: 1156
constexpr int ITEMS = 5;                                // items per thread: 5 x 256 = 1280 >= 1156
constexpr int VSB = 96;                                 // bytes per voxel of the fp16 tile: 4 quads x 16 B + 32 pad
constexpr int TILE_BYTES = ITEMS * 64 * VSB;            // 30720: 320 voxel slots (289 used; the items past the tile write zeros into the rest)
constexpr int W_BASE = 2 * TILE_BYTES;                  // two tile buffers, then the weights
constexpr int ROWB = LXY * VSB;

struct Args {
    const float* in;
    const float* w;      // conv_tr2g order, fp32: [cin group of 16][27 taps, class-major][cout tile][64 lanes][4]
    const float* bias;
    void* out;           // fp16 NDHWC
    int N, D, H, W;      // input dims (output = 2x)
    int nty, ntx, zsplit, zlen, nct;
    int flags, ocs, oco;
};

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

// NG = 32-channel cin groups (1: 32 -> 16, 2: 64 -> 32)
template <int NG, bool RELU>
__global__ void __launch_bounds__(NT, 1) conv_tr2m_f16_kernel(Args a, int nwg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int v = lane & 15, cq = lane >> 4;
    auto ldsu = [&](unsigned off) -> u32x4 { return *reinterpret_cast<const u32x4*>(smem + off); };

    int wg = xcd_remap(blockIdx.x, nwg);
    const int ct = wg % a.nct; wg /= a.nct;          // cout tile: neighbours in the grid share their input tiles in L2
    const int tx_ = wg % a.ntx; wg /= a.ntx;
    const int ty_ = wg % a.nty; wg /= a.nty;
    const int zs = wg % a.zsplit;
    const int n = wg / a.zsplit;
    const int X0 = tx_ * 16, Y0 = ty_ * 16, zb = zs * a.zlen;
    const int nsteps = a.zlen + 1;                   // input planes zb-1 .. zb+zlen-1 (the first one only opens output plane 2 zb)
    constexpr int CIN = NG * 32;
    const size_t HW = (size_t)a.H * a.W;
    const unsigned HWI = (unsigned)(HW * CIN * 4);
    const float* in_n = a.in + (size_t)n * a.D * HW * CIN;

    // ---- fp16 weights of this cout tile -> LDS (resident): fragment (G, sq) at W_BASE + (G * 27 + sq) KB; lane (cout = lane & 15, kq = lane >> 4)
    //      holds input channels 32 G + 8 kq .. + 7 = two float4 of the tr2g-order image (group 2 G + (kq >> 1), channel quads 2 (kq & 1), + 1)
    {
        const f32x4* w4 = reinterpret_cast<const f32x4*>(a.w);
        for (int p = wave; p < NG * 27; p += 4) {
            const int G = p / 27, sq = p - G * 27;
            const size_t frag = ((size_t)((2 * G + (cq >> 1)) * 27 + sq) * a.nct + ct) * 64;
            const f32x4 lo = w4[frag + v + 16 * (2 * (cq & 1))], hi = w4[frag + v + 16 * (2 * (cq & 1) + 1)];
            *reinterpret_cast<u32x4*>(smem + W_BASE + p * 1024 + lane * 16) = to_h8(lo, hi);
        }
    }
    // ---- tile staging: global -> registers (one micro-step ahead) -> fp16 -> LDS.  Item it of thread tid = (voxel u, quad q) =
    //      ((it * 256 + tid) >> 2, tid & 3): 8 channels = two 16-byte loads, one 16-byte LDS write at u * 96 + q * 16.  OOB items read zeros.
    // the four voxels of 16 consecutive lanes are taken in the order 0, 2, 1, 3: a ds_write_b128 is served in groups of 8 lanes on 32 banks; two
    // voxels 96 B apart overlap in 8 of them, 192 B apart they use the other 16 banks
    const int gperm = ((tid >> 2) & ~3) | (((tid >> 2) & 1) << 1) | (((tid >> 2) >> 1) & 1);
    unsigned rel[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int item = it * NT + tid;
        const int u = it * 64 + gperm, q = item & 3, ly = u / LXY, lx = u - ly * LXY;
        const int y = Y0 - 1 + ly, x = X0 - 1 + lx;
        const bool ok = item < TILE_SLOTS && y >= 0 && y < a.H && x >= 0 && x < a.W;
        rel[it] = ok ? (unsigned)(((y * a.W + x) * CIN + q * 8) * 4) : kOOB;
    }
    const unsigned cw = (unsigned)(gperm * VSB + (tid & 3) * 16);       // item it: cw + it * 64 * VSB (+ the tile buffer)
    f32x4 stg[ITEMS][2];
    // tile of (step sp, cin group cg): `addr` = address of channel 32 cg of input plane zb - 1 + sp (kept incrementally)
    auto fetch_tile = [&](int sp, int cg, unsigned long long addr) __attribute__((always_inline)) {
        const bool ok = (unsigned)(zb - 1 + sp) < (unsigned)a.D && sp < nsteps;
        const __amdgpu_buffer_rsrc_t rp = make_rsrc((const void*)(ok ? addr : (unsigned long long)in_n), ok ? HWI - (unsigned)(128 * cg) : 0u);
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            stg[it][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, (int)rel[it], 0, 0));
            stg[it][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, (int)rel[it], 16, 0));
        }
    };
    auto commit = [&](unsigned buf) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it)
            *reinterpret_cast<u32x4*>(smem + buf + cw + (unsigned)(it * 64 * VSB)) = to_h8(stg[it][0], stg[it][1]);
    };

    // ---- B operand addresses: lane (v, cq) reads voxel (row, lx = v + 1 - dx) of the tile, its channel quad cq; one base per x offset,
    //      rows / y offsets are immediates.  ba[dxi] points at tile row 4 * wave of buffer 0.
    unsigned ba[2];
#pragma unroll
    for (int dxi = 0; dxi < 2; ++dxi) ba[dxi] = (unsigned)((4 * wave * LXY + v + 1 - dxi) * VSB + cq * 16);
    unsigned wa = (unsigned)(W_BASE + lane * 16);      // + cin group * 27 KB (per micro-step), + tap * 1 KB (immediate)

    // ---- epilogue addressing (bytes, fp16 output): lane writes couts 4 cq .. 4 cq + 3 (of this cout tile) of output voxel (2 z + pz, 2 y + py, 2 x + px)
    const int OH = 2 * a.H, OW = 2 * a.W;
    const unsigned ob = (unsigned)((((2 * (Y0 + 4 * wave)) * OW + 2 * (X0 + v)) * a.ocs + a.oco + 16 * ct + 4 * cq) * 2);
    const unsigned PLANE_O = (unsigned)((size_t)OH * OW * a.ocs * 2);          // bytes per output plane
    unsigned char* out_n = (unsigned char*)a.out + (size_t)n * (2 * a.D) * OH * OW * a.ocs * 2;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bias_l = (a.flags & PCC_CONV_BIAS) ? *reinterpret_cast<const f32x4*>(a.bias + 16 * ct + 4 * cq) : zero4;

    // ---- prologue
    int s2 = 0, c2 = 0;                               // input plane step / cin group of the next tile to fetch (wave-uniform)
    unsigned long long tile_pl = (unsigned long long)in_n + (unsigned long long)(long long)(zb - 1) * HWI;     // tile of micro-step 0
    auto fetch_next = [&]() __attribute__((always_inline)) {
        fetch_tile(s2, c2, tile_pl);
        if (++c2 == NG) { c2 = 0; ++s2; tile_pl += HWI - 128 * (NG - 1); } else tile_pl += 128;
    };
    fetch_next();
    commit(0u);
    fetch_next();                                     // raw tile of micro-step 1 waits in registers
    __syncthreads();

    f32x4 E[2][4][4], O[4][4];                        // accumulators [set][class py * 2 + px][row]
    u32x4 b[2][2][4];                                 // fp16 B operands [dyi][dxi][row] of the current micro-step
    u32x4 wf[4];                                      // weight fragment ring, three taps ahead
    auto load_b = [&](unsigned buf) __attribute__((always_inline)) {
#pragma unroll
        for (int dyi = 0; dyi < 2; ++dyi)
#pragma unroll
            for (int dxi = 0; dxi < 2; ++dxi)
#pragma unroll
                for (int i = 0; i < 4; ++i) b[dyi][dxi][i] = ldsu(buf + ba[dxi] + (unsigned)((i + 1 - dyi) * ROWB));
    };
    auto load_w = [&](int slot, int sq) __attribute__((always_inline)) { wf[slot] = ldsu(wa + (unsigned)(sq * 1024)); };
    load_b(0u);
    load_w(0, tap_of(18).sq); load_w(1, tap_of(19).sq); load_w(2, tap_of(20).sq);      // step 0 is the halo plane: its first tap is 18

    int s = 0, c = 0;                                 // input plane step / cin group of the current micro-step
    unsigned long long out_pl = (unsigned long long)out_n + (unsigned long long)(long long)(2 * (zb - 2)) * PLANE_O;   // planes 2 (z - 1), 2 (z - 1) + 1 of step s = 0

    // epilogue item e of a finished plane pair: e < 16: odd set (pz = 1), class e >> 2, row e & 3; else the old even_cur (pz = 0)
    auto finish = [&](auto ph_tag, auto e_tag, const __amdgpu_buffer_rsrc_t& rout, u32x2& keep) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_tag)::value, e = decltype(e_tag)::value;
        constexpr int pz = e < 16 ? 1 : 0, cls = (e >> 2) & 3, i = e & 3, py = cls >> 1, px = cls & 1;
        f32x4 o = acc_read(pz ? O[cls][i] : E[PH ^ 1][cls][i]);
        if (RELU) o = __builtin_elementwise_maximum(o, zero4);
        keep = __builtin_bit_cast(u32x2, __builtin_convertvector(o, h16x4));
        const unsigned soff = (unsigned)pz * PLANE_O + (unsigned)(((2 * i + py) * OW + px) * a.ocs * 2);
        __builtin_amdgcn_raw_buffer_store_b64(keep, rout, (int)ob, (int)soff, 0);
    };

    // BUF = tile buffer of this micro-step (the other one receives the next tile).  HALO: the slab's first input plane (zb - 1) only
    // contributes its kz = 2 taps (to output plane 2 zb): 9 taps instead of 27
    auto micro = [&](auto ph_tag, auto first_tag, auto halo_tag, auto buf_tag) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_tag)::value, BUF = decltype(buf_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value, HALO = decltype(halo_tag)::value;
        constexpr int T0 = HALO ? 18 : 0;
        const bool prev_ok = s >= 2;
        const __amdgpu_buffer_rsrc_t rout = make_rsrc((const void*)(prev_ok ? out_pl : (unsigned long long)out_n), FIRST && prev_ok ? 2u * PLANE_O : 0u);
        u32x2 ost[2][3];
        static_for(std::make_integer_sequence<int, 27 - T0>{}, [&](auto t_tag) __attribute__((always_inline)) {
            constexpr int t = decltype(t_tag)::value + T0;
            constexpr Tap T = tap_of(t);
            if constexpr (t + 3 < 27) load_w((t + 3 - T0) % 4, tap_of(t + 3).sq);
            constexpr bool open = FIRST && T.opens && T.kz != 0;     // first tap of a class of the odd / even_next set in this plane
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4& acc = T.kz == 0 ? E[PH][T.cls][i] : T.kz == 1 ? O[T.cls][i] : E[PH ^ 1][T.cls][i];
                acc = mfma_f16(wf[(t - T0) % 4], b[T.dyi][T.dxi][i], open ? bias_l : acc);
            }
            __builtin_amdgcn_sched_barrier(0);
            // the tile of the NEXT micro-step (raw, in registers since the previous micro-step) goes into the OTHER buffer -- nobody reads that
            // one before the barrier at the end of this micro-step -- and the tile after next is requested
            if constexpr (t == 26) {
                commit((unsigned)((BUF ^ 1) * TILE_BYTES));
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
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): this wave's tile writes; the others': the barrier
        __syncthreads();
        if (++c == NG) { c = 0; ++s; out_pl += 2ull * PLANE_O; wa -= (unsigned)((NG - 1) * 27 * 1024); } else wa += 27u * 1024u;
        load_b((unsigned)((BUF ^ 1) * TILE_BYTES));
        if (s == 0) { load_w(0, tap_of(18).sq); load_w(1, tap_of(19).sq); load_w(2, tap_of(20).sq); }
        else { load_w(0, tap_of(0).sq); load_w(1, tap_of(1).sq); load_w(2, tap_of(2).sq); }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    // one input plane = NG micro-steps; tile buffers alternate per micro-step: NG = 1: buffer = plane parity, NG = 2: buffer = cin group
    auto plane = [&](auto ph_tag, auto halo_tag) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_tag)::value;
        if constexpr (NG == 1) {
            micro(ph_tag, std::true_type{}, halo_tag, std::integral_constant<int, PH>{});
        } else {
            micro(ph_tag, std::true_type{}, halo_tag, B0{});
            micro(ph_tag, std::false_type{}, halo_tag, B1{});
        }
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
        u32x2 keep[32];
        if (nsteps & 1) {      // last plane had PH = 0: its even_cur is E[0] = "E[PH ^ 1]" of a PH = 1 epilogue
            static_for(std::make_integer_sequence<int, 32>{}, [&](auto e_tag) __attribute__((always_inline)) {
                finish(std::integral_constant<int, 1>{}, e_tag, rout, keep[decltype(e_tag)::value]); });
        } else {
            static_for(std::make_integer_sequence<int, 32>{}, [&](auto e_tag) __attribute__((always_inline)) {
                finish(std::integral_constant<int, 0>{}, e_tag, rout, keep[decltype(e_tag)::value]); });
        }
#pragma unroll
        for (int e = 0; e < 32; ++e) asm volatile("" ::"v"(keep[e]));
    }
}

}  // namespace pcctr2mh

using namespace pcctr2mh;

// fp16 mode with fp16 hand-over only; shape-only rule (both layers always take it when eligible: no batch / grid dependence)
bool pcc_tr2m_f16_covers(const pcc_conv_desc* d) {
    if (!d->transposed || d->k != 3 || d->stride != 2) return false;
    if (!((d->Cin == 32 && d->Cout == 16) || (d->Cin == 64 && d->Cout == 32))) return false;
    if (d->H % 16 || d->W % 16) return false;
    if ((d->flags & (PCC_CONV_F16 | PCC_CONV_OUT16)) != (PCC_CONV_F16 | PCC_CONV_OUT16)) return false;
    if (d->flags & (PCC_CONV_ADD | PCC_CONV_CLIP01 | PCC_CONV_IN16 | PCC_CONV_RES16)) return false;
    const int ocs = d->out_cstride ? d->out_cstride : d->Cout;
    if (ocs % 4 || d->out_coffset % 4) return false;
    if ((double)d->H * d->W * d->Cin * 4.0 >= 2147483648.0) return false;                 // one input plane per descriptor
    if (2.0 * (2.0 * d->H) * (2.0 * d->W) * ocs * 2.0 >= 2147483648.0) return false;      // two output planes per descriptor
    return true;
}

int pcc_conv_tr2m_f16(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w_tr2g, const float* bias, void* out,
                      hipStream_t st) {
    PCC_REQUIRE(pcc_tr2m_f16_covers(d), "pcc_conv_tr2m_f16: shape not covered");
    Args a;
    a.in = in; a.w = w_tr2g; a.bias = bias; a.out = out;
    a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W;
    a.nty = d->H / 16; a.ntx = d->W / 16; a.nct = d->Cout / 16;
    a.flags = d->flags;
    a.ocs = d->out_cstride ? d->out_cstride : d->Cout;
    a.oco = d->out_coffset;
    const int base = d->N * a.nty * a.ntx * a.nct;
    int zs = 1;
    while (base * zs < ctx->num_cu && d->D % (zs * 2) == 0 && d->D / (zs * 2) >= 4) zs *= 2;
    a.zsplit = zs; a.zlen = d->D / zs;
    const int nwg = base * zs;
    const int NG = d->Cin / 32;
    const int lds = W_BASE + NG * 27 * 1024;
    typedef void (*kern_t)(Args, int);
    const bool relu = (d->flags & PCC_CONV_RELU) != 0;
    const kern_t kern = NG == 1 ? (relu ? (kern_t)conv_tr2m_f16_kernel<1, true> : (kern_t)conv_tr2m_f16_kernel<1, false>)
                                : (relu ? (kern_t)conv_tr2m_f16_kernel<2, true> : (kern_t)conv_tr2m_f16_kernel<2, false>);
    { const int rc = pcc_enable_big_lds((const void*)kern, lds); if (rc != PCC_OK) return rc; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(NT), lds, st, a, nwg);
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}
