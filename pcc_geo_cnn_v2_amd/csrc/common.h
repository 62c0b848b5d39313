// Internal helpers shared by the translation units of libpcc_geo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/pcc_geo.h"

struct pcc_ctx {
    int device;
    int num_cu;
    hipDeviceProp_t prop;
    void* profile = nullptr;     // live kernel timing state of network.hip (pcc_profile_select / pcc_profile_read)
    // Device scratch owned by the context (grown on demand, freed with it): the partial sums of the 64-channel fp16 layers
    // (conv_f16.hip).  One tensor per context: launches that use it must be ordered on one stream.
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    // PCC_NUM_* switches in effect (include/pcc_geo.h, "codec numerics"): read from the environment ONCE, in pcc_ctx_create;
    // every dispatch decision that changes the bits of a layer tests this word, never getenv
    uint32_t numerics = 0;
    bool num(uint32_t bit) const { return (numerics & bit) != 0; }
    // per-block max |x| slots of the fp16-split kernels for callers that chain layers themselves (pcc_conv3d; pcc_network_forward keeps
    // its own in the workspace): grown on demand, one tensor per context like `scratch`
    unsigned* amax = nullptr;
    int amax_cap = 0;
};
int pcc_ctx_amax(pcc_ctx* ctx, int n, unsigned** ptr);
// Side channel of the fp16-split kernels (conv_wino_f16s.hip): in_amax[n] = fp32 bits of max |in| over block n as recorded by the
// layer that produced `in` (nullptr: the launcher computes it, pcc_block_amax); out_amax (zeroed by the caller, nullptr: not wanted)
// receives the same for `out` from every kernel that can record it -- pcc_conv_records_amax says which.
// A block's maximum is kept as PCC_AMAX_SLOTS partial maxima (row n = slots [n * PCC_AMAX_SLOTS, (n + 1) * PCC_AMAX_SLOTS)): thousands of
// device-scope atomicMax on ONE address serialise (measured: 16 k of them on 32 addresses cost the first layer 100 us); a recording
// wave picks its slot from its workgroup index, the reader takes the max over the 64 (one load per lane + a wave reduction).
// out_recorded: set by the launcher when the kernel it chose fills out_amax
constexpr int PCC_AMAX_SLOTS = 64;
struct pcc_conv_ext { const unsigned* in_amax; unsigned* out_amax; bool out_recorded; };
#ifdef __HIPCC__
// lane-local running max -> slot `spread` of block row `row` (all lanes of the wave call it; inf is recorded as FLT_MAX, NaNs never arrive)
__device__ __forceinline__ void pcc_amax_record(unsigned* row, float mx, int spread) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0) {
        const unsigned mb = __builtin_bit_cast(unsigned, mx);
        atomicMax(row + (spread & (PCC_AMAX_SLOTS - 1)), mb > 0x7f7fffffu ? 0x7f7fffffu : mb);
    }
}
// max over the slots of a block row, wave-uniform (non-negative fp32 bit patterns order like unsigned integers)
__device__ __forceinline__ unsigned pcc_amax_read(const unsigned* row) {
    unsigned m = row[threadIdx.x & (PCC_AMAX_SLOTS - 1)];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)m, off); m = o > m ? o : m; }
    return (unsigned)__builtin_amdgcn_readfirstlane((int)m);
}
#endif
int pcc_block_amax(pcc_ctx* ctx, const float* x, int N, size_t per_block, unsigned* amax, hipStream_t st);
int pcc_ctx_scratch(pcc_ctx* ctx, size_t bytes, void** ptr);
uint32_t pcc_numerics_from_env();
void pcc_profile_free(pcc_ctx* ctx);

void pcc_set_error(const char* fmt, ...);

#define PCC_API extern "C" __attribute__((visibility("default")))

#define PCC_CHECK_HIP(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            pcc_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,    \
                          __LINE__);                                                          \
            return PCC_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

#define PCC_REQUIRE(cond, ...)        \
    do {                              \
        if (!(cond)) {                \
            pcc_set_error(__VA_ARGS__); \
            return PCC_ERR_ARG;       \
        }                             \
    } while (0)

// TF `SAME` geometry (forward conv): out = ceil(n/s), pad_low = max((out-1)s + k - n, 0) / 2.
__host__ __device__ inline int pcc_same_out(int n, int s) { return (n + s - 1) / s; }
__host__ __device__ inline int pcc_same_pad_low(int n, int k, int s) {
    int tot = (pcc_same_out(n, s) - 1) * s + k - n;
    return tot > 0 ? tot / 2 : 0;
}

// entry points implemented per translation unit
int pcc_conv3d_generic(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w,
                       const float* bias, const float* residual, float* out, hipStream_t st);
int pcc_conv3d_mfma(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w_packed,
                    const float* bias, const float* residual, float* out, hipStream_t st);
// The fixed-threshold extraction folded into the last layer (16 -> 1 transposed conv, conv_cout1_mfma_kernel): the kernel also
// writes bit (z,y,x) of block n = (clip ? clamp01(x_hat) : x_hat) > thr[n] into `mask` (one bit per voxel, row-major, 32-bit
// words little-endian).  *fused tells the caller whether the layer took that path (else: pcc_threshold_compact on x_hat).
struct pcc_thr_fuse { const float* thr; int clip; uint32_t* mask; };
struct pcc_conv_ext;
int pcc_conv3d_mfma_thr(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w_packed, const float* bias,
                        const float* residual, float* out, const pcc_thr_fuse* fuse, bool* fused, pcc_conv_ext* ext, hipStream_t st);
bool pcc_conv_wants_amax(const pcc_ctx* ctx, const pcc_conv_desc* d);
int pcc_conv3d_ext(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w, const float* w_packed, const float* bias,
                   const float* residual, float* out, pcc_conv_ext* ext, void* stream);
// points from the bit mask (elementwise.hip): scratch = [B * D plane counts][B * D*H*W / 32 mask words]
uint32_t* pcc_threshold_mask_of(int32_t* scratch, int32_t B, int32_t D);
int pcc_threshold_from_mask(pcc_ctx* ctx, int32_t B, int32_t D, int32_t H, int32_t W, float* xyz, int32_t* counts, int64_t cap,
                            int32_t* scratch, hipStream_t st);
// fused element-wise steps of the codec graphs (elementwise.hip): quantiser + pack, scale fold + pack, unpack + dequantiser
int pcc_quantize_pack(pcc_ctx* ctx, const float* v, const float* medians, int32_t* sym, float* deq, int32_t N, int64_t vox,
                      int32_t C, int32_t mode, int32_t channels_first, void* dst, int32_t dst_bytes, int32_t* tile_max,
                      void* stream);
int pcc_index_pack(pcc_ctx* ctx, const float* sigma, const float* table, int32_t L, int32_t* idx, int32_t N, int64_t vox,
                   int32_t C, int32_t channels_first, void* dst, int32_t dst_bytes, void* stream);
int pcc_unpack_dequantize(pcc_ctx* ctx, const void* src, int32_t src_bytes, int32_t N, int64_t vox, int32_t C,
                          int32_t channels_first, int32_t* sym, const float* medians, float* deq, void* stream);
// Winograd F(2x2,3x3) (x,y) + direct z path for 16->16 and 32->32 k3 stride-1 layers (conv_wino.hip)
bool pcc_wino_eligible(const pcc_conv_desc* d);
int pcc_conv_wino(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* u_packed,
                  const float* bias, const float* residual, float* out, hipStream_t st);
// z-marching k3 stride-2 transposed conv for 32 -> 16 / 64 -> 32 (conv_tr2m.hip); weights in conv_tr2g_kernel's packed order
bool pcc_tr2m_eligible(const pcc_conv_desc* d);
bool pcc_tr2m_preferred(const pcc_ctx* ctx, const pcc_conv_desc* d);
int pcc_conv_tr2m(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w_tr2g, const float* bias, float* out,
                  hipStream_t st);
// the same march with split-bf16 operands for 32 -> 16 (conv_tr2m_bf16.hip)
size_t pcc_tr2m_bf16_packed_floats(int Cin, int Cout);
void pcc_tr2m_bf16_pack(int Cin, int Cout, const float* w_tr2g, float* out);
bool pcc_tr2m_bf16_covers(const pcc_conv_desc* d);
int pcc_conv_tr2m_bf16(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w_split, const float* bias, float* out,
                       pcc_conv_ext* ext, hipStream_t st);
// the same march with two fp16 pieces under a per-block pre-scale for 32 -> 16 (conv_tr2m_f16s.hip, round 6)
constexpr int PCC_TR2M_F16S_TAIL = 64;      // floats behind the fragments, [0] = the weight scale
size_t pcc_tr2m_f16s_packed_floats(int Cin, int Cout);
void pcc_tr2m_f16s_pack(int Cin, int Cout, const float* w_tr2g, float* out);
bool pcc_tr2m_f16s_covers(const pcc_conv_desc* d);
int pcc_conv_tr2m_f16s(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w_f16s, const float* bias, float* out,
                       pcc_conv_ext* ext, hipStream_t st);
// the march in the fp16 mode (conv_tr2m_f16.hip, round 5): fp32 input, fp16 MFMA, fp16 output (PCC_CONV_F16 | PCC_CONV_OUT16), 32 -> 16 and 64 -> 32
bool pcc_tr2m_f16_covers(const pcc_conv_desc* d);
int pcc_conv_tr2m_f16(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w_tr2g, const float* bias, void* out,
                      hipStream_t st);
// fp16-storage k3 stride-1 kernel for Cin = Cout in {16, 32} (conv_f16.hip), PCC_CONV_IN16 layers
bool pcc_f16_eligible(const pcc_conv_desc* d);
size_t pcc_f16_packed_bytes(int C);
void pcc_f16_pack(int C, const float* wlog, unsigned short* out);
int pcc_conv_f16(pcc_ctx* ctx, const pcc_conv_desc* d, const void* in, const void* w_packed, const float* bias,
                 const void* residual, void* out, bool out32, hipStream_t st);
constexpr int PCC_WINO_U_FLOATS = 48 * 64 * 4;   // per (cin group, cout group): [z tap][point][lane][cin quad member]
// split-bf16 Winograd path (conv_wino_bf16.hip): the same U as three bf16 pieces, two MFMA operands of 16 B per (row, lane)
constexpr int PCC_WINO_UB_FLOATS = 48 * 2 * 64 * 4;
bool pcc_wino_bf16_covers(const pcc_conv_desc* d);      // (given pcc_wino_eligible)
void pcc_wino_bf16_pack(int ngroups, const float* u_f32, float* out);
int pcc_conv_wino_bf16(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* ub_packed, const float* bias,
                       const float* residual, float* out, hipStream_t st);
// two-piece fp16 Winograd path (conv_wino_f16s.hip, round 6): U as two fp16 pieces under one power-of-two scale per layer
constexpr int PCC_WINO_UH_FLOATS = 64 * 776 / 4;        // per (cin group, cout group): 64 lanes x (12 slots x 4 points x 16 B + 8 B pad)
constexpr int PCC_WINO_UH_TAIL = 64;                    // [0] = the scale
bool pcc_wino_f16s_covers(const pcc_conv_desc* d);      // (given pcc_wino_eligible)
void pcc_wino_f16s_pack(int ngroups, const float* u_f32, float* out);
int pcc_conv_wino_f16s(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* uh_packed, const float* bias,
                       const float* residual, float* out, pcc_conv_ext* ext, hipStream_t st);
// direct k3 stride-1 convolution with split-bf16 operands for Cin = Cout in {32, 64} (conv_split.hip)
size_t pcc_split_packed_floats(int C);
void pcc_split_pack(int C, const float* wlog, float* out);
bool pcc_split_covers(const pcc_conv_desc* d);
bool pcc_split_preferred(const pcc_ctx* ctx, const pcc_conv_desc* d);
// Conv3DTranspose k3 stride 2, 64 -> 32 / 64 -> 64, split operands (conv_split.hip); weights = pcc_tr2m_bf16_pack(tr2g-order image)
bool pcc_tr2_split_covers(const pcc_conv_desc* d);
int pcc_conv_tr2_split(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w_split, const float* bias, float* out,
                       pcc_conv_ext* ext, hipStream_t st);
int pcc_conv_split(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w_split, const float* bias, const float* residual,
                   float* out, hipStream_t st);
// Kernels that use more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize, which is set per DEVICE:
// remember the (function, device) pairs this thread has configured.
#include <utility>
#include <vector>
inline int pcc_enable_big_lds(const void* kern, int lds_bytes) {
    if (lds_bytes <= 64 * 1024) return PCC_OK;
    static thread_local std::vector<std::pair<const void*, int>> done;
    int dev = 0;
    PCC_CHECK_HIP(hipGetDevice(&dev));
    for (const auto& e : done)
        if (e.first == kern && e.second == dev) return PCC_OK;
    PCC_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    done.emplace_back(kern, dev);
    return PCC_OK;
}

inline bool pcc_wino_channels(int cin, int cout) { return cin == cout && (cin == 16 || cin == 32 || cin == 64); }
