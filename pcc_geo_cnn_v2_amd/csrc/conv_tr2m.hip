// Conv3DTranspose, k = 3, stride 2, fp32, marching along z (round 3) -- the first layer of a SynthesisBlock,
// /root/reference/src/model_transforms.py:78 (SynthesisTransformProgressiveV2: 64 -> 32 @16^3 -> 32^3, 32 -> 16 @32^3 -> 64^3).
//
// Parity decomposition (as conv_tr2g_kernel): out[2b + p] = sum over the taps k == p (mod 2) of W[k] . in[b + (p - k) / 2], per
// dimension: even outputs take tap 0 at b and tap 2 at b - 1, odd outputs take tap 1 at b.  27 taps per input voxel, no zero
// insertion, no wasted multiplies.  What differs from conv_tr2g_kernel (2 x 8 x 16 tiles with a z halo, two waves per SIMD,
// weights streamed from L2 per tap, 0.60-0.66 of the MFMA peak):
//   * a workgroup owns a 16 x 16 (x, y) column of one cout tile and MARCHES along z, one INPUT plane per step, one 16-channel
//     cin group per MICRO-STEP.  Input plane z feeds output planes 2z (kz = 0), 2z + 1 (kz = 1) and 2z + 2 (kz = 2): three
//     accumulator sets -- even_cur, odd, even_next; even_next becomes even_cur of the next plane -- of 4 parity classes (py, px)
//     x 4 rows = 48 float4 = 192 AccVGPRs stay live; no z halo is staged twice (the old tile re-read 3 planes per 2);
//   * ONE wave per SIMD, 512 registers: the 16 B-operand vectors of a micro-step (4 input offsets (dy, dx) x 4 rows) are read
//     from LDS once and feed all 27 taps x 4 k-slices = 432 MFMAs; the LDS image is swizzled (slot ^= 2 * bit 2 of x) so that
//     both x offsets read conflict-free (the old 80-byte voxel stride was 2-way conflicted);
//   * the weights of this cout tile (27 x NG KB) are LDS-resident for the life of the workgroup: no weight traffic in the loop;
//   * the finished planes are reduced in the shadow of the NEXT plane's first taps: the kz = 0 taps only touch even_cur, so the
//     odd set is read out (AccVGPR -> ReLU -> store) under taps 0..7 and the old even_cur under taps 0..15, before taps 9 /
//     18 re-open those registers (first tap of a class starts from the bias: srcC = bias, no init pass);
//   * everything that selects code is a compile-time constant (plane parity, first micro-step): no branches in the stream.
// Fixed summation order per output (cin group -> tap -> k-slice, one fp32 chain): bit-deterministic, batch invariant.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"

namespace pcctr2m {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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
constexpr int ITEMS = 5;                                // 1 KB chunks per wave (4 x 5 = 20 >= 1156 / 64)
constexpr int TILE_BYTES = 4 * ITEMS * 1024;            // 20480
constexpr int W_BASE = 2 * TILE_BYTES;                  // ring of two tiles, then the weights
constexpr int ROWB = LXY * 64;                          // bytes per tile row

struct Tr2mArgs {
    const float* in;
    const float* w;      // conv_tr2g order: [cin group][27 taps, class-major][cout tile][64 lanes][4]
    const float* bias;
    float* out;
    int N, D, H, W;      // input dims (output = 2x)
    int nty, ntx, zsplit, zlen, nct;
    int flags, ocs, oco;
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
__global__ void __launch_bounds__(NT, 1) conv_tr2m_kernel(Tr2mArgs a, int nwg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int v = lane & 15, cq = lane >> 4;
    auto ldsr = [&](unsigned off) -> f32x4 { return *reinterpret_cast<const f32x4*>(smem + off); };
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

    // ---- weights of this cout tile -> LDS (resident): NG x 27 fragments of 1 KB, fragment (g, sq) at W_BASE + (g * 27 + sq) KB
    {
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, (unsigned)(NG * 27 * a.nct) * 1024u);
        for (int p = wave; p < NG * 27; p += 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(smem + W_BASE + p * 1024), 16, lane * 16, (p * a.nct + ct) * 1024, 0, 0);
    }
    // ---- tile staging: global -> LDS directly.  LDS slot s = 4 * voxel + (channel quad ^ 2 * bit 2 of lx): the swizzle is
    //      applied on the global side (lane L of chunk c fetches what belongs into slot 64 c + L); OOB lanes write zeros.
    unsigned rel[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int slot = (wave * ITEMS + it) * 64 + lane;
        const int u = slot >> 2, ly = u / LXY, lx = u - ly * LXY;
        const int q = (slot & 3) ^ (((lx >> 2) & 1) << 1);
        const int y = Y0 - 1 + ly, x = X0 - 1 + lx;
        const bool ok = slot < TILE_SLOTS && y >= 0 && y < a.H && x >= 0 && x < a.W;
        rel[it] = ok ? (unsigned)(((y * a.W + x) * CIN + q * 4) * 4) : kOOB;
    }
    // tile of (step sp, cin group cg): `addr` = address of channel 16 cg of input plane zb - 1 + sp (kept incrementally)
    auto stage_tile = [&](unsigned ring_off, int sp, int cg, unsigned long long addr) __attribute__((always_inline)) {
        const bool ok = (unsigned)(zb - 1 + sp) < (unsigned)a.D && sp < nsteps;
        const __amdgpu_buffer_rsrc_t rp = make_rsrc((const void*)(ok ? addr : (unsigned long long)in_n), ok ? HWI - (unsigned)(64 * cg) : 0u);
#pragma unroll
        for (int it = 0; it < ITEMS; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lds_ptr)(smem + ring_off + (wave * ITEMS + it) * 1024), 16, (int)rel[it], 0, 0, 0);
    };

    // ---- B operand addresses: lane (v, cq) reads voxel (row, lx = v + 1 + dx) of the tile, its channel quad cq; one base per
    //      x offset (the swizzle depends on lx), rows / y offsets are immediates.  ba[dxi] points at tile row 4 * wave.
    unsigned ba[2];
#pragma unroll
    for (int dxi = 0; dxi < 2; ++dxi) {
        const int lx = v + 1 - dxi;
        ba[dxi] = (unsigned)((4 * wave * LXY + lx) * 64 + ((cq ^ (((lx >> 2) & 1) << 1)) << 4));
    }
    unsigned wa = (unsigned)(W_BASE + lane * 16);      // + cin group * 27 KB (per micro-step), + tap * 1 KB (immediate)

    // ---- epilogue addressing: lane writes couts 4 cq .. 4 cq + 3 (of this cout tile) of output voxel (2 z + pz, 2 y + py, 2 x + px)
    const int OH = 2 * a.H, OW = 2 * a.W;
    const unsigned ob = (unsigned)((((2 * (Y0 + 4 * wave)) * OW + 2 * (X0 + v)) * a.ocs + a.oco + 16 * ct + 4 * cq) * 4);
    const unsigned PLANE_O = (unsigned)((size_t)OH * OW * a.ocs * 4);          // bytes per output plane (launcher: two planes < 2^31)
    float* out_n = a.out + (size_t)n * (2 * a.D) * OH * OW * a.ocs;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bias_l = (a.flags & PCC_CONV_BIAS) ? *reinterpret_cast<const f32x4*>(a.bias + 16 * ct + 4 * cq) : zero4;

    // ---- prologue
    unsigned long long tile_pl = (unsigned long long)in_n + (unsigned long long)(long long)(zb - 1) * HWI;     // tile of micro-step 0
    stage_tile(0, 0, 0, tile_pl);
    tile_pl += 64;                                    // micro-step 1 = (plane 0, group 1): NG >= 2
    __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0)
    __syncthreads();

    f32x4 E[2][4][4], O[4][4];                        // accumulators [set][class py * 2 + px][row]
    f32x4 b[2][2][4];                                 // B operands [dyi][dxi][row] of the current micro-step
    f32x4 wf[2];                                      // weight fragment double buffer
    auto load_b = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int dyi = 0; dyi < 2; ++dyi)
#pragma unroll
            for (int dxi = 0; dxi < 2; ++dxi)
#pragma unroll
                for (int i = 0; i < 4; ++i) b[dyi][dxi][i] = ldsr(ba[dxi] + (unsigned)((i + 1 - dyi) * ROWB));
    };
    load_b();
    wf[0] = ldsr(wa + (unsigned)(tap_of(18).sq * 1024));      // step 0 is the halo plane: its first tap is 18

    // wave-uniform march state
    int s = 0, c = 0;                                 // input plane step / cin group of the current micro-step
    int s1 = 0, c1 = 1;                               // ... of the next micro-step (its tile address: tile_pl)
    unsigned cur_off = 0;                             // ring slot of the current tile
    unsigned long long out_pl = (unsigned long long)out_n + (unsigned long long)(long long)(2 * (zb - 2)) * PLANE_O;   // planes 2 (z - 1), 2 (z - 1) + 1 of step s = 0

    // epilogue item e of a finished plane pair: e < 16: odd set (pz = 1), class e >> 2, row e & 3; else the old even_cur (pz = 0)
    auto finish = [&](auto ph_tag, auto e_tag, const __amdgpu_buffer_rsrc_t& rout, f32x4& keep) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_tag)::value, e = decltype(e_tag)::value;
        constexpr int pz = e < 16 ? 1 : 0, cls = (e >> 2) & 3, i = e & 3, py = cls >> 1, px = cls & 1;
        f32x4 o = acc_read(pz ? O[cls][i] : E[PH ^ 1][cls][i]);
        if (RELU) o = __builtin_elementwise_maximum(o, zero4);
        keep = o;
        const unsigned soff = (unsigned)pz * PLANE_O + (unsigned)(((2 * i + py) * OW + px) * a.ocs * 4);
        buf_store4(rout, keep, ob, soff);
    };

    // HALO: the slab's first input plane (zb - 1) only contributes its kz = 2 taps (to output plane 2 zb): 9 taps instead of 27
    auto micro = [&](auto ph_tag, auto first_tag, auto halo_tag) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value, HALO = decltype(halo_tag)::value;
        constexpr int T0 = HALO ? 18 : 0;
        // tile of the next micro-step -> the other ring slot (its last reader passed the closing barrier of the previous micro-step)
        stage_tile(cur_off ^ (unsigned)TILE_BYTES, s1, c1, tile_pl);
        // FIRST: the planes finished by the previous input plane leave under the first taps (zero-sized: stores dropped, s < 2)
        const bool prev_ok = s >= 2;
        const __amdgpu_buffer_rsrc_t rout = make_rsrc((const void*)(prev_ok ? out_pl : (unsigned long long)out_n), FIRST && prev_ok ? 2u * PLANE_O : 0u);
        f32x4 ost[2][3];
        static_for(std::make_integer_sequence<int, 27 - T0>{}, [&](auto t_tag) __attribute__((always_inline)) {
            constexpr int t = decltype(t_tag)::value + T0;
            constexpr Tap T = tap_of(t);
            constexpr Tap Tn = tap_of(t + 1 < 27 ? t + 1 : 0);
            // weight fragment of the next tap (wraps to tap 0 of the next micro-step: loaded after the barrier instead)
            if constexpr (t + 1 < 27) wf[(t + 1) & 1] = ldsr(wa + (unsigned)(Tn.sq * 1024));
            constexpr bool open = FIRST && T.opens && T.kz != 0;     // first tap of a class of the odd / even_next set in this plane
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4& acc = T.kz == 0 ? E[PH][T.cls][i] : T.kz == 1 ? O[T.cls][i] : E[PH ^ 1][T.cls][i];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t & 1][j], b[T.dyi][T.dxi][i][j], (open && j == 0) ? bias_l : acc, 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
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
        // the next tile must have landed before the barrier publishes it; the only younger memory operations are the 32 stores
        // of a FIRST micro-step
        if (FIRST && !HALO) __builtin_amdgcn_s_waitcnt(0x8F70);      // vmcnt(32)
        else __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
        __syncthreads();
        // ---- advance (wave-uniform) and fetch the operands of the next micro-step
        cur_off ^= (unsigned)TILE_BYTES;
        const int d_ring = cur_off ? TILE_BYTES : -TILE_BYTES;
        ba[0] += (unsigned)d_ring; ba[1] += (unsigned)d_ring;
        if (++c == NG) { c = 0; ++s; out_pl += 2ull * PLANE_O; wa -= (unsigned)((NG - 1) * 27 * 1024); } else wa += 27u * 1024u;
        tile_pl += 64;
        if (++c1 == NG) { c1 = 0; ++s1; tile_pl += HWI - 64 * NG; }
        load_b();
        // first weight fragment of the next micro-step (buffer parity follows its first tap: 0, or 18 inside the halo plane)
        if (s == 0) wf[0] = ldsr(wa + (unsigned)(tap_of(18).sq * 1024));
        else wf[0] = ldsr(wa + (unsigned)(tap_of(0).sq * 1024));
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
    }
}

}  // namespace pcctr2m

using namespace pcctr2m;

// Eligible: k3 stride-2 transposed, (Cin, Cout) in {(32, 16), (64, 32)}, H and W multiples of 16, bias / ReLU epilogue only
// (no layer of the c* graphs adds a residual to, or clips, a stride-2 transposed conv), fp32.
bool pcc_tr2m_eligible(const pcc_conv_desc* d) {
    if (!d->transposed || d->k != 3 || d->stride != 2) return false;
    if (!((d->Cin == 32 && d->Cout == 16) || (d->Cin == 64 && d->Cout == 32))) return false;
    if (d->H % 16 || d->W % 16) return false;
    if (d->flags & (PCC_CONV_ADD | PCC_CONV_CLIP01 | PCC_CONV_F16 | PCC_CONV_OUT16 | PCC_CONV_IN16 | PCC_CONV_RES16)) return false;
    const int ocs = d->out_cstride ? d->out_cstride : d->Cout;
    if (ocs % 4 || d->out_coffset % 4) return false;
    if ((double)d->H * d->W * d->Cin * 4.0 >= 2147483648.0) return false;                 // one input plane per descriptor
    if (2.0 * (2.0 * d->H) * (2.0 * d->W) * ocs * 4.0 >= 2147483648.0) return false;      // two output planes per descriptor
    return true;
}

// z split: every CU gets a workgroup; every split pays one extra (halo) input plane of 9 taps
static int tr2m_zsplit(const pcc_ctx* ctx, const pcc_conv_desc* d) {
    const int base = d->N * (d->H / 16) * (d->W / 16) * (d->Cout / 16);
    int zs = 1;
    while (base * zs < ctx->num_cu && d->D % (zs * 2) == 0 && d->D / (zs * 2) >= 4) zs *= 2;
    return zs;
}

// AUTO dispatch: 32 -> 16 always; 64 -> 32 from 32 input planes up (16^3 x 32 blocks gives 4-plane slabs: the 9-tap halo plane and
// the 108 KB weight prologue per workgroup make the tiled conv_tr2g_kernel faster there: 127 vs 135 us).  The rule must not
// depend on the batch size: the two kernels sum in different orders, and encoder and decoder (which may chunk differently) have
// to produce the same bits (DESIGN_HISTORY.md section 4).
bool pcc_tr2m_preferred(const pcc_ctx* ctx, const pcc_conv_desc* d) {
    (void)ctx;
    if (!pcc_tr2m_eligible(d)) return false;
    return d->Cin == 32 || d->D >= 32;
}

int pcc_conv_tr2m(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w_tr2g, const float* bias, float* out,
                  hipStream_t st) {
    PCC_REQUIRE(pcc_tr2m_eligible(d), "pcc_conv_tr2m: shape not covered");
    Tr2mArgs a;
    a.in = in; a.w = w_tr2g; a.bias = bias; a.out = out;
    a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W;
    a.nty = d->H / 16; a.ntx = d->W / 16; a.nct = d->Cout / 16;
    a.flags = d->flags;
    a.ocs = d->out_cstride ? d->out_cstride : d->Cout;
    a.oco = d->out_coffset;
    const int zs = tr2m_zsplit(ctx, d);
    a.zsplit = zs; a.zlen = d->D / zs;
    const int nwg = d->N * a.nty * a.ntx * a.nct * zs;
    const int NG = d->Cin / 16;
    const int lds = W_BASE + NG * 27 * 1024;
    typedef void (*kern_t)(Tr2mArgs, int);
    static const kern_t kerns[4] = {conv_tr2m_kernel<2, false>, conv_tr2m_kernel<2, true>, conv_tr2m_kernel<4, false>, conv_tr2m_kernel<4, true>};
    const kern_t kern = kerns[(NG == 4 ? 2 : 0) + ((d->flags & PCC_CONV_RELU) ? 1 : 0)];
    { const int rc = pcc_enable_big_lds((const void*)kern, lds); if (rc != PCC_OK) return rc; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(NT), lds, st, a, nwg);
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}
