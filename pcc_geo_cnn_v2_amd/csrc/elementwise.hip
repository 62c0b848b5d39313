// HBM-bound element-wise / reduction / compaction kernels of the codec path (gfx950).
//   quantise, dequantise, scale->index, voxelise, threshold + order-preserving compaction,
//   focal-loss reduction.  All are deterministic (fixed reduction order, no float atomics).
#include "common.h"

namespace {

constexpr int kThreads = 256;

__host__ inline unsigned grid_for(size_t n, int per_thread, int num_cu) {
    size_t blocks = (n + (size_t)kThreads * per_thread - 1) / ((size_t)kThreads * per_thread);
    const size_t cap = (size_t)num_cu * 8;  // grid-stride beyond 8 blocks/CU
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    return (unsigned)blocks;
}

// ---- quantise -----------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) k_quantize(const float* __restrict__ v, const float* __restrict__ med,
                                                       int32_t* __restrict__ sym, float* __restrict__ deq,
                                                       size_t n, int C, int mode) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float m = med ? med[i % (size_t)C] : 0.f;
        float q;
        if (mode == PCC_ROUND_FLOOR_HALF) {
            const float hm = 0.5f - m;  // tfc 1.3: floor(inputs + (half - medians))
            q = floorf(v[i] + hm);
        } else {
            q = rintf(v[i] - m);        // half to even
        }
        if (sym) sym[i] = (int32_t)q;
        if (deq) deq[i] = q + m;
    }
}

__global__ void __launch_bounds__(kThreads) k_dequantize(const int32_t* __restrict__ sym, const float* __restrict__ med,
                                                         float* __restrict__ deq, size_t n, int C) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float m = med ? med[i % (size_t)C] : 0.f;
        deq[i] = (float)sym[i] + m;
    }
}

// ---- scale -> index ------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) k_scale_index(const float* __restrict__ sigma, const float* __restrict__ table,
                                                          int L, int32_t* __restrict__ idx, size_t n) {
    __shared__ float tab[256];
    for (int j = threadIdx.x; j < L; j += blockDim.x) tab[j] = table[j];
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = sigma[i];
        if (!(s >= tab[0])) s = tab[0];
        int id = L - 1;
        for (int j = 0; j < L - 1; ++j) id -= (s <= tab[j]) ? 1 : 0;
        idx[i] = id;
    }
}

// ---- symbols / CDF-row indexes <-> the host coder's stream order ------------------------------------------------------------
// The range-coded streams of one block are its tensor flattened in the reference's memory order (model_types.py:180,254,377:
// data_format 'channels_first' -> (C, D,H,W), channel-major); on the device everything is NDHWC int32.  These two kernels are the
// only thing between the quantisers and the PCIe copy: a 64-voxel x 64-channel tile goes through LDS (coalesced on both sides),
// the value is narrowed (int16 symbols, uint8 CDF rows: a third of the PCIe bytes) and every tile reports max|value| as a plain
// store (no atomics, nothing to zero) so that the host can tell afterwards whether the narrow type was enough.
constexpr int kPackT = 64;

template <typename T>
__device__ __forceinline__ T narrow(int32_t v) { return (T)v; }

template <typename T, int N>
struct alignas(sizeof(T) * N) PackVec { T v[N]; };

// What the pack kernel packs: int32 values that exist already (SRC_INT), or values it produces itself on the way -- the
// quantiser (SRC_QUANT: float -> symbol, also written as int32 + dequantised float in NDHWC) or the scale -> CDF-row fold
// (SRC_INDEX) -- so that the encoder / decoder graphs need one launch where they had two.
enum { SRC_INT = 0, SRC_QUANT = 1, SRC_INDEX = 2 };
struct PackSrc {
    const int32_t* isrc;      // SRC_INT
    const float* fsrc;        // SRC_QUANT: values; SRC_INDEX: sigma
    const float* med;         // SRC_QUANT: medians (C) or NULL
    int32_t* sym;             // SRC_QUANT / SRC_INDEX: int32 NDHWC output
    float* deq;               // SRC_QUANT: dequantised output (may be NULL)
    const float* table;       // SRC_INDEX: scale table
    int mode, L;
};

// idx = (L-1) - #{j < L-1 : sigma <= table[j]} (patch_gaussian_conditional.py:104-116).  For an ascending table -- what
// the reference builds -- that count is a lower bound, found in log2(L) steps; any other table takes the literal count.
__device__ __forceinline__ int scale_row(float sg, const float* tab, int L, bool ascending) {
    if (!(sg >= tab[0])) sg = tab[0];
    if (ascending) {
        int lo = 0, n = L - 1;                 // first j in [0, L-1) with tab[j] >= sg
        while (n > 0) {
            const int half = n >> 1;
            if (tab[lo + half] < sg) { lo += half + 1; n -= half + 1; } else n = half;
        }
        return lo;
    }
    int id = L - 1;
    for (int j = 0; j < L - 1; ++j) id -= (sg <= tab[j]) ? 1 : 0;
    return id;
}

template <int SRC>
__device__ __forceinline__ int32_t produce(const PackSrc& ps, size_t gi, int c, const float* tab, bool ascending) {
    if constexpr (SRC == SRC_INT) {
        return ps.isrc[gi];
    } else if constexpr (SRC == SRC_QUANT) {
        const float m = ps.med ? ps.med[c] : 0.f;
        const float q = ps.mode == PCC_ROUND_FLOOR_HALF ? floorf(ps.fsrc[gi] + (0.5f - m)) : rintf(ps.fsrc[gi] - m);   // as k_quantize
        ps.sym[gi] = (int32_t)q;
        if (ps.deq) ps.deq[gi] = q + m;
        return (int32_t)q;
    } else {
        const int id = scale_row(ps.fsrc[gi], tab, ps.L, ascending);                                                   // == k_scale_index
        ps.sym[gi] = id;
        return id;
    }
}

template <typename T, int SRC>
__global__ void __launch_bounds__(kThreads) k_symbols_pack(PackSrc ps, T* __restrict__ dst, int vox, int C,
                                                            int vtiles, int ctiles, int channels_first,
                                                            int32_t* __restrict__ tile_max) {
    __shared__ int32_t tile[kPackT][kPackT + 1];
    __shared__ int32_t wmax[kThreads / 64];
    __shared__ float tab[SRC == SRC_INDEX ? 256 : 1];
    bool ascending = false;
    if constexpr (SRC == SRC_INDEX) {
        for (int j = threadIdx.x; j < ps.L; j += kThreads) tab[j] = ps.table[j];
        __syncthreads();
        int ok = 1;
        for (int j = threadIdx.x; j + 1 < ps.L; j += kThreads) ok &= tab[j] <= tab[j + 1];
        ascending = __syncthreads_and(ok) != 0;
    }
    int t = blockIdx.x;
    const int ct = t % ctiles; t /= ctiles;
    const int vt = t % vtiles;
    const int n = t / vtiles;
    const int v0 = vt * kPackT, c0 = ct * kPackT;
    const int nv = min(kPackT, vox - v0), nc = min(kPackT, C - c0);
    const size_t g0 = ((size_t)n * vox + v0) * C + c0;        // NDHWC index of the tile's first element
    int32_t m = 0;
    const bool full = nv == kPackT && nc == kPackT && (C & 3) == 0 && (vox & 3) == 0;     // whole, 16-byte aligned tile
    if (channels_first && full) {
        // four 16-byte loads per thread in flight, then LDS; four consecutive voxels per store on the way out
        int4 x[4];
        if constexpr (SRC == SRC_INT) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = threadIdx.x + k * kThreads, v = i >> 4, q = i & 15;
                x[k] = *reinterpret_cast<const int4*>(ps.isrc + g0 + (size_t)v * C + q * 4);
            }
        } else {
            float4 f[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = threadIdx.x + k * kThreads, v = i >> 4, q = i & 15;
                f[k] = *reinterpret_cast<const float4*>(ps.fsrc + g0 + (size_t)v * C + q * 4);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = threadIdx.x + k * kThreads, v = i >> 4, q = i & 15;
                const size_t gi = g0 + (size_t)v * C + q * 4;
                const float fv[4] = {f[k].x, f[k].y, f[k].z, f[k].w};
                int32_t r[4];
                if constexpr (SRC == SRC_QUANT) {
                    float dq[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float md = ps.med ? ps.med[c0 + q * 4 + e] : 0.f;
                        const float qq = ps.mode == PCC_ROUND_FLOOR_HALF ? floorf(fv[e] + (0.5f - md)) : rintf(fv[e] - md);
                        r[e] = (int32_t)qq; dq[e] = qq + md;
                    }
                    if (ps.deq) *reinterpret_cast<float4*>(ps.deq + gi) = make_float4(dq[0], dq[1], dq[2], dq[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[e] = scale_row(fv[e], tab, ps.L, ascending);
                }
                x[k] = make_int4(r[0], r[1], r[2], r[3]);
                *reinterpret_cast<int4*>(ps.sym + gi) = x[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + k * kThreads, v = i >> 4, q = i & 15;
            tile[q * 4 + 0][v] = x[k].x; tile[q * 4 + 1][v] = x[k].y; tile[q * 4 + 2][v] = x[k].z; tile[q * 4 + 3][v] = x[k].w;
            m = max(max(m, abs(x[k].x)), max(max(abs(x[k].y), abs(x[k].z)), abs(x[k].w)));
        }
    } else if (channels_first) {
        for (int i = threadIdx.x; i < nv * kPackT; i += kThreads) {
            const int v = i / kPackT, c = i % kPackT;
            if (c < nc) {
                const int32_t x = produce<SRC>(ps, g0 + (size_t)v * C + c, c0 + c, tab, ascending);
                tile[c][v] = x;
                m = max(m, abs(x));
            }
        }
    }
    if (channels_first) {
        __syncthreads();
        T* db = dst + ((size_t)n * C + c0) * vox + v0;
        if (nv == kPackT && nc == kPackT && (vox & 3) == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = threadIdx.x + k * kThreads, c = i >> 4, vq = i & 15;
                PackVec<T, 4> o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o.v[e] = narrow<T>(tile[c][vq * 4 + e]);
                *reinterpret_cast<PackVec<T, 4>*>(db + (size_t)c * vox + vq * 4) = o;
            }
        } else {
            for (int i = threadIdx.x; i < nc * kPackT; i += kThreads) {
                const int c = i / kPackT, v = i % kPackT;
                if (v < nv) db[(size_t)c * vox + v] = narrow<T>(tile[c][v]);
            }
        }
    } else {
        T* db = dst + g0;
        for (int i = threadIdx.x; i < nv * kPackT; i += kThreads) {
            const int v = i / kPackT, c = i % kPackT;
            if (c < nc) {
                const int32_t x = produce<SRC>(ps, g0 + (size_t)v * C + c, c0 + c, tab, ascending);
                db[(size_t)v * C + c] = narrow<T>(x);
                m = max(m, abs(x));
            }
        }
    }
    if (tile_max) {
        for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) tile_max[blockIdx.x] = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
    }
}

// deq != NULL: the dequantiser rides along (deq = float(symbol) + medians[c], as k_dequantize)
template <typename T>
__global__ void __launch_bounds__(kThreads) k_symbols_unpack(const T* __restrict__ src, int32_t* __restrict__ dst, int vox, int C,
                                                              int vtiles, int ctiles, int channels_first,
                                                              const float* __restrict__ med, float* __restrict__ deq) {
    __shared__ int32_t tile[kPackT][kPackT + 1];
    int t = blockIdx.x;
    const int ct = t % ctiles; t /= ctiles;
    const int vt = t % vtiles;
    const int n = t / vtiles;
    const int v0 = vt * kPackT, c0 = ct * kPackT;
    const int nv = min(kPackT, vox - v0), nc = min(kPackT, C - c0);
    const size_t g0 = ((size_t)n * vox + v0) * C + c0;
    int32_t* db = dst + g0;
    const bool full = nv == kPackT && nc == kPackT && (C & 3) == 0 && (vox & 3) == 0;
    if (channels_first && full) {
        const T* sb = src + ((size_t)n * C + c0) * vox + v0;
        PackVec<T, 4> x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + k * kThreads, c = i >> 4, vq = i & 15;
            x[k] = *reinterpret_cast<const PackVec<T, 4>*>(sb + (size_t)c * vox + vq * 4);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + k * kThreads, c = i >> 4, vq = i & 15;
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[c][vq * 4 + e] = (int32_t)x[k].v[e];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + k * kThreads, v = i >> 4, q = i & 15;
            const int4 sv = make_int4(tile[q * 4 + 0][v], tile[q * 4 + 1][v], tile[q * 4 + 2][v], tile[q * 4 + 3][v]);
            *reinterpret_cast<int4*>(db + (size_t)v * C + q * 4) = sv;
            if (deq) {
                const float4 mv = med ? *reinterpret_cast<const float4*>(med + c0 + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(deq + g0 + (size_t)v * C + q * 4) =
                    make_float4((float)sv.x + mv.x, (float)sv.y + mv.y, (float)sv.z + mv.z, (float)sv.w + mv.w);
            }
        }
    } else if (channels_first) {
        const T* sb = src + ((size_t)n * C + c0) * vox + v0;
        for (int i = threadIdx.x; i < nc * kPackT; i += kThreads) {
            const int c = i / kPackT, v = i % kPackT;
            if (v < nv) tile[c][v] = (int32_t)sb[(size_t)c * vox + v];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nv * kPackT; i += kThreads) {
            const int v = i / kPackT, c = i % kPackT;
            if (c < nc) {
                db[(size_t)v * C + c] = tile[c][v];
                if (deq) deq[g0 + (size_t)v * C + c] = (float)tile[c][v] + (med ? med[c0 + c] : 0.f);
            }
        }
    } else {
        const T* sb = src + g0;
        for (int i = threadIdx.x; i < nv * kPackT; i += kThreads) {
            const int v = i / kPackT, c = i % kPackT;
            if (c < nc) {
                const int32_t x = (int32_t)sb[(size_t)v * C + c];
                db[(size_t)v * C + c] = x;
                if (deq) deq[g0 + (size_t)v * C + c] = (float)x + (med ? med[c0 + c] : 0.f);
            }
        }
    }
}

// ---- voxelise ------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) k_voxelize(const int32_t* __restrict__ pts, const int32_t* __restrict__ block_of,
                                                       long long npts, int B, int D, int H, int W,
                                                       float* __restrict__ dense) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npts; i += (long long)gridDim.x * blockDim.x) {
        const int x = pts[i * 3 + 0], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
        const int b = block_of ? block_of[i] : 0;
        if (b < 0 || b >= B || x < 0 || x >= D || y < 0 || y >= H || z < 0 || z >= W) continue;
        dense[(((size_t)b * D + x) * H + y) * W + z] = 1.0f;
    }
}

// ---- wave / block primitives ---------------------------------------------------------------
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// exclusive scan of one int per thread over a 256-thread block; returns exclusive prefix, *total
__device__ __forceinline__ int block_excl_scan(int v, int* total, int* lds4) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int incl = wave_incl_scan(v, lane);
    __syncthreads();  // lds4 reuse guard
    if (lane == 63) lds4[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int wv = 0; wv < kThreads / 64; ++wv) {
        const int s = lds4[wv];
        if (wv < wave) base += s;
        tot += s;
    }
    *total = tot;
    return base + incl - v;
}

// ---- threshold + compaction ----------------------------------------------------------------
constexpr int kChunk = 4096;  // voxels per workgroup (4 passes of 256 threads x 4 voxels)

__device__ __forceinline__ void load4(const float* __restrict__ xb, size_t idx, size_t nvox, bool vec, float (&v)[4]) {
    if (vec && idx + 3 < nvox) {
        const float4 t = *reinterpret_cast<const float4*>(xb + idx);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (idx + e < nvox) ? xb[idx + e] : -INFINITY;
    }
}

__device__ __forceinline__ bool hit(float v, float thr, int clip) {
    if (clip) v = fminf(fmaxf(v, 0.f), 1.f);
    return v > thr;  // float32 compare (numpy 1.18 value-based casting of the float64 threshold)
}

__global__ void __launch_bounds__(kThreads) k_thr_count(const float* __restrict__ x, const float* __restrict__ thr,
                                                        int clip, size_t nvox, int chunks, int32_t* __restrict__ chunk_cnt) {
    __shared__ int lds4[4];
    const int b = blockIdx.y, c = blockIdx.x;
    const float* xb = x + (size_t)b * nvox;
    const bool vec = (nvox % 4) == 0;
    const float t = thr[b];
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const size_t idx = (size_t)c * kChunk + ((size_t)j * kThreads + threadIdx.x) * 4;
        float v[4];
        load4(xb, idx, nvox, vec, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) cnt += hit(v[e], t, clip) ? 1 : 0;
    }
    int total;
    block_excl_scan(cnt, &total, lds4);
    if (threadIdx.x == 0) chunk_cnt[(size_t)b * chunks + c] = total;
}

__global__ void __launch_bounds__(kThreads) k_thr_write(const float* __restrict__ x, const float* __restrict__ thr,
                                                        int clip, size_t nvox, int chunks, int H, int W,
                                                        const int32_t* __restrict__ chunk_cnt, float* __restrict__ xyz,
                                                        int32_t* __restrict__ counts, long long cap) {
    __shared__ int lds4[4];
    const int b = blockIdx.y, c = blockIdx.x;
    const float* xb = x + (size_t)b * nvox;
    const bool vec = (nvox % 4) == 0;
    const float t = thr[b];
    // offset of this chunk = sum of the counts of the preceding chunks of the same block
    int part = 0;
    for (int j = threadIdx.x; j < c; j += kThreads) part += chunk_cnt[(size_t)b * chunks + j];
    int offset;
    block_excl_scan(part, &offset, lds4);
    float* ob = xyz + (size_t)b * (size_t)cap * 3;
    const int HW = H * W;
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
        const size_t idx = (size_t)c * kChunk + ((size_t)j * kThreads + threadIdx.x) * 4;
        float v[4];
        load4(xb, idx, nvox, vec, v);
        bool h[4];
        int cnt = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) { h[e] = hit(v[e], t, clip); cnt += h[e] ? 1 : 0; }
        int total;
        int pos = offset + block_excl_scan(cnt, &total, lds4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (h[e]) {
                if (pos < cap) {
                    const size_t vi = idx + e;
                    const int d = (int)(vi / HW);
                    const int r = (int)(vi - (size_t)d * HW);
                    ob[(size_t)pos * 3 + 0] = (float)d;
                    ob[(size_t)pos * 3 + 1] = (float)(r / W);
                    ob[(size_t)pos * 3 + 2] = (float)(r % W);
                }
                ++pos;
            }
        }
        offset += total;
    }
    if (c == chunks - 1 && threadIdx.x == 0) counts[b] = offset;
}

// ---- points from the occupancy bit mask the last layer wrote (pcc_thr_fuse) --------------------------------------------------
// One workgroup per (z plane, block): pass 1 counts the plane's set bits, pass 2 writes its points behind those of the planes
// before it (np.argwhere order: z, y, x ascending).  Together they read D*H*W / 8 bytes per block.
__global__ void __launch_bounds__(kThreads) k_mask_count(const uint32_t* __restrict__ mask, int wpp, int32_t* __restrict__ plane_cnt) {
    __shared__ int lds4[4];
    const uint32_t* mp = mask + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * wpp;
    int cnt = 0;
    for (int w = threadIdx.x; w < wpp; w += kThreads) cnt += __popc(mp[w]);
    int total;
    block_excl_scan(cnt, &total, lds4);
    if (threadIdx.x == 0) plane_cnt[blockIdx.y * gridDim.x + blockIdx.x] = total;
}

__global__ void __launch_bounds__(kThreads) k_mask_write(const uint32_t* __restrict__ mask, int wpp, int W,
                                                         const int32_t* __restrict__ plane_cnt, float* __restrict__ xyz,
                                                         int32_t* __restrict__ counts, long long cap) {
    __shared__ int lds4[4];
    const int b = blockIdx.y, z = blockIdx.x, D = gridDim.x;
    int part = 0;
    for (int j = threadIdx.x; j < z; j += kThreads) part += plane_cnt[b * D + j];
    int offset;
    block_excl_scan(part, &offset, lds4);
    const uint32_t* mp = mask + (size_t)(b * D + z) * wpp;
    float* ob = xyz + (size_t)b * (size_t)cap * 3;
    const float fz = (float)z;
#pragma unroll 1
    for (int base = 0; base < wpp; base += kThreads) {
        const int w = base + threadIdx.x;
        uint32_t bits = w < wpp ? mp[w] : 0u;
        int total;
        int pos = offset + block_excl_scan(__popc(bits), &total, lds4);
        while (bits) {
            const int vi = w * 32 + __ffs(bits) - 1;
            bits &= bits - 1;
            if (pos < cap) {
                ob[(size_t)pos * 3 + 0] = fz;
                ob[(size_t)pos * 3 + 1] = (float)(vi / W);
                ob[(size_t)pos * 3 + 2] = (float)(vi % W);
            }
            ++pos;
        }
        offset += total;
    }
    if (z == D - 1 && threadIdx.x == 0) counts[b] = offset;
}

// ---- focal loss ----------------------------------------------------------------------------
constexpr int kFocalBlocks = 1024;  // fixed: the reduction tree is identical on every device

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);  // butterfly: same value, same order in all lanes
    return v;
}

__global__ void __launch_bounds__(kThreads) k_focal_partial(const float* __restrict__ yt, const float* __restrict__ yp,
                                                            size_t n, float gamma, float alpha, float* __restrict__ partial) {
    __shared__ float lds[kThreads / 64];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float t = yt[i], p = yp[i];
        float pt1 = (t == 1.f) ? p : 1.f;
        float pt0 = (t == 0.f) ? p : 0.f;
        pt1 = fminf(fmaxf(pt1, 1e-3f), .999f);
        pt0 = fminf(fmaxf(pt0, 1e-3f), .999f);
        s += alpha * powf(1.f - pt1, gamma) * logf(pt1) + (1.f - alpha) * powf(pt0, gamma) * logf(1.f - pt0);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int wv = 0; wv < kThreads / 64; ++wv) tot += lds[wv];
        partial[blockIdx.x] = tot;
    }
}

__global__ void __launch_bounds__(kThreads) k_focal_final(const float* __restrict__ partial, int nparts, float* __restrict__ out) {
    __shared__ float lds[kThreads / 64];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += kThreads) s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int wv = 0; wv < kThreads / 64; ++wv) tot += lds[wv];
        out[0] = -tot;
    }
}

}  // namespace

PCC_API int pcc_quantize(pcc_ctx* ctx, const float* v, const float* medians, int32_t* sym, float* deq, size_t n,
                         int32_t C, int32_t mode, void* stream) {
    PCC_REQUIRE(ctx && v && (sym || deq), "pcc_quantize: NULL argument");
    PCC_REQUIRE(C > 0 && (mode == PCC_ROUND_FLOOR_HALF || mode == PCC_ROUND_HALF_EVEN), "pcc_quantize: bad C/mode");
    if (n == 0) return PCC_OK;
    PCC_CHECK_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_quantize, dim3(grid_for(n, 4, ctx->num_cu)), dim3(kThreads), 0, (hipStream_t)stream, v, medians,
                       sym, deq, n, C, mode);
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}

PCC_API int pcc_dequantize(pcc_ctx* ctx, const int32_t* sym, const float* medians, float* deq, size_t n, int32_t C,
                           void* stream) {
    PCC_REQUIRE(ctx && sym && deq && C > 0, "pcc_dequantize: bad argument");
    if (n == 0) return PCC_OK;
    PCC_CHECK_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_dequantize, dim3(grid_for(n, 4, ctx->num_cu)), dim3(kThreads), 0, (hipStream_t)stream, sym,
                       medians, deq, n, C);
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}

PCC_API int pcc_scale_to_index(pcc_ctx* ctx, const float* sigma, const float* table, int32_t L, int32_t* idx, size_t n,
                               void* stream) {
    PCC_REQUIRE(ctx && sigma && table && idx, "pcc_scale_to_index: NULL argument");
    PCC_REQUIRE(L >= 1 && L <= 256, "pcc_scale_to_index: table length %d not in [1,256]", L);
    if (n == 0) return PCC_OK;
    PCC_CHECK_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_scale_index, dim3(grid_for(n, 4, ctx->num_cu)), dim3(kThreads), 0, (hipStream_t)stream, sigma,
                       table, L, idx, n);
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}

PCC_API size_t pcc_symbols_tiles(int32_t N, int64_t vox, int32_t C) {
    if (N <= 0 || vox <= 0 || C <= 0) return 0;
    return (size_t)N * (size_t)((vox + kPackT - 1) / kPackT) * (size_t)((C + kPackT - 1) / kPackT);
}

template <int SRC>
static int launch_pack(pcc_ctx* ctx, const PackSrc& ps, int32_t N, int64_t vox, int32_t C, int32_t channels_first, void* dst,
                       int32_t dst_bytes, int32_t* tile_max, void* stream, const char* who) {
    PCC_REQUIRE(ctx && dst, "%s: NULL argument", who);
    PCC_REQUIRE(N > 0 && vox > 0 && vox < (1LL << 31) && C > 0, "%s: bad dimension", who);
    PCC_REQUIRE(dst_bytes == 1 || dst_bytes == 2 || dst_bytes == 4, "%s: dst_bytes must be 1, 2 or 4", who);
    PCC_CHECK_HIP(hipSetDevice(ctx->device));
    const int vt = (int)((vox + kPackT - 1) / kPackT), ct = (C + kPackT - 1) / kPackT;
    const size_t tiles = (size_t)N * vt * ct;
    PCC_REQUIRE(tiles < (1ull << 31), "%s: too many tiles for one launch", who);
    hipStream_t st = (hipStream_t)stream;
    if (dst_bytes == 1)
        hipLaunchKernelGGL((k_symbols_pack<uint8_t, SRC>), dim3((unsigned)tiles), dim3(kThreads), 0, st, ps, (uint8_t*)dst, (int)vox, C, vt, ct, channels_first, tile_max);
    else if (dst_bytes == 2)
        hipLaunchKernelGGL((k_symbols_pack<int16_t, SRC>), dim3((unsigned)tiles), dim3(kThreads), 0, st, ps, (int16_t*)dst, (int)vox, C, vt, ct, channels_first, tile_max);
    else
        hipLaunchKernelGGL((k_symbols_pack<int32_t, SRC>), dim3((unsigned)tiles), dim3(kThreads), 0, st, ps, (int32_t*)dst, (int)vox, C, vt, ct, channels_first, tile_max);
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}

PCC_API int pcc_symbols_pack(pcc_ctx* ctx, const int32_t* src, int32_t N, int64_t vox, int32_t C, int32_t channels_first,
                             void* dst, int32_t dst_bytes, int32_t* tile_max, void* stream) {
    PCC_REQUIRE(src, "pcc_symbols_pack: NULL argument");
    PackSrc ps = {};
    ps.isrc = src;
    return launch_pack<SRC_INT>(ctx, ps, N, vox, C, channels_first, dst, dst_bytes, tile_max, stream, "pcc_symbols_pack");
}

// quantiser + pack / scale fold + pack / unpack + dequantiser in one launch each (the codec graphs, network.hip)
int pcc_quantize_pack(pcc_ctx* ctx, const float* v, const float* medians, int32_t* sym, float* deq, int32_t N, int64_t vox,
                      int32_t C, int32_t mode, int32_t channels_first, void* dst, int32_t dst_bytes, int32_t* tile_max,
                      void* stream) {
    PCC_REQUIRE(v && sym && (mode == PCC_ROUND_FLOOR_HALF || mode == PCC_ROUND_HALF_EVEN), "pcc_quantize_pack: bad argument");
    PackSrc ps = {};
    ps.fsrc = v; ps.med = medians; ps.sym = sym; ps.deq = deq; ps.mode = mode;
    return launch_pack<SRC_QUANT>(ctx, ps, N, vox, C, channels_first, dst, dst_bytes, tile_max, stream, "pcc_quantize_pack");
}

int pcc_index_pack(pcc_ctx* ctx, const float* sigma, const float* table, int32_t L, int32_t* idx, int32_t N, int64_t vox,
                   int32_t C, int32_t channels_first, void* dst, int32_t dst_bytes, void* stream) {
    PCC_REQUIRE(sigma && table && idx && L >= 1 && L <= 256, "pcc_index_pack: bad argument");
    PackSrc ps = {};
    ps.fsrc = sigma; ps.table = table; ps.L = L; ps.sym = idx;
    return launch_pack<SRC_INDEX>(ctx, ps, N, vox, C, channels_first, dst, dst_bytes, nullptr, stream, "pcc_index_pack");
}

static int launch_unpack(pcc_ctx* ctx, const void* src, int32_t src_bytes, int32_t N, int64_t vox, int32_t C,
                         int32_t channels_first, int32_t* dst, const float* med, float* deq, void* stream) {
    PCC_REQUIRE(ctx && src && dst, "pcc_symbols_unpack: NULL argument");
    PCC_REQUIRE(N > 0 && vox > 0 && vox < (1LL << 31) && C > 0, "pcc_symbols_unpack: bad dimension");
    PCC_REQUIRE(src_bytes == 1 || src_bytes == 2 || src_bytes == 4, "pcc_symbols_unpack: src_bytes must be 1, 2 or 4");
    PCC_CHECK_HIP(hipSetDevice(ctx->device));
    const int vt = (int)((vox + kPackT - 1) / kPackT), ct = (C + kPackT - 1) / kPackT;
    const size_t tiles = (size_t)N * vt * ct;
    PCC_REQUIRE(tiles < (1ull << 31), "pcc_symbols_unpack: too many tiles for one launch");
    hipStream_t st = (hipStream_t)stream;
    if (src_bytes == 1)
        hipLaunchKernelGGL(k_symbols_unpack<uint8_t>, dim3((unsigned)tiles), dim3(kThreads), 0, st, (const uint8_t*)src, dst, (int)vox, C, vt, ct, channels_first, med, deq);
    else if (src_bytes == 2)
        hipLaunchKernelGGL(k_symbols_unpack<int16_t>, dim3((unsigned)tiles), dim3(kThreads), 0, st, (const int16_t*)src, dst, (int)vox, C, vt, ct, channels_first, med, deq);
    else
        hipLaunchKernelGGL(k_symbols_unpack<int32_t>, dim3((unsigned)tiles), dim3(kThreads), 0, st, (const int32_t*)src, dst, (int)vox, C, vt, ct, channels_first, med, deq);
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}

PCC_API int pcc_symbols_unpack(pcc_ctx* ctx, const void* src, int32_t src_bytes, int32_t N, int64_t vox, int32_t C,
                               int32_t channels_first, int32_t* dst, void* stream) {
    return launch_unpack(ctx, src, src_bytes, N, vox, C, channels_first, dst, nullptr, nullptr, stream);
}

int pcc_unpack_dequantize(pcc_ctx* ctx, const void* src, int32_t src_bytes, int32_t N, int64_t vox, int32_t C,
                          int32_t channels_first, int32_t* sym, const float* medians, float* deq, void* stream) {
    PCC_REQUIRE(deq, "pcc_unpack_dequantize: NULL argument");
    return launch_unpack(ctx, src, src_bytes, N, vox, C, channels_first, sym, medians, deq, stream);
}

PCC_API int pcc_voxelize(pcc_ctx* ctx, const int32_t* pts, const int32_t* block_of, int64_t npts, int32_t B, int32_t D,
                         int32_t H, int32_t W, float* dense, void* stream) {
    PCC_REQUIRE(ctx && dense && (pts || npts == 0), "pcc_voxelize: NULL argument");
    PCC_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && npts >= 0, "pcc_voxelize: bad dimension");
    if (npts == 0) return PCC_OK;
    PCC_CHECK_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_voxelize, dim3(grid_for((size_t)npts, 1, ctx->num_cu)), dim3(kThreads), 0, (hipStream_t)stream,
                       pts, block_of, (long long)npts, B, D, H, W, dense);
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}

// scratch of the thresholding calls: the chunk counts of pcc_threshold_compact, or -- when the last layer delivers the occupancy
// bits (pcc_thr_fuse) -- [B * D plane counts, padded to 4 ints][B * ceil(D*H*W / 32) mask words]; sized for either
PCC_API size_t pcc_threshold_scratch_ints(int32_t B, int32_t D, int32_t H, int32_t W) {
    const size_t nvox = (size_t)D * H * W;
    const size_t chunks = (size_t)B * ((nvox + kChunk - 1) / kChunk);
    const size_t fused = (((size_t)B * D + 3) & ~(size_t)3) + (size_t)B * ((nvox + 31) / 32);
    return chunks > fused ? chunks : fused;
}

uint32_t* pcc_threshold_mask_of(int32_t* scratch, int32_t B, int32_t D) {
    return reinterpret_cast<uint32_t*>(scratch + (((size_t)B * D + 3) & ~(size_t)3));
}

int pcc_threshold_from_mask(pcc_ctx* ctx, int32_t B, int32_t D, int32_t H, int32_t W, float* xyz, int32_t* counts, int64_t cap,
                            int32_t* scratch, hipStream_t st) {
    PCC_REQUIRE(ctx && xyz && counts && scratch && B > 0 && B <= 65535 && ((size_t)H * W) % 32 == 0 && cap >= 0,
                "pcc_threshold_from_mask: bad argument");
    const int wpp = (int)((size_t)H * W / 32);
    const uint32_t* mask = pcc_threshold_mask_of(scratch, B, D);
    hipLaunchKernelGGL(k_mask_count, dim3(D, B), dim3(kThreads), 0, st, mask, wpp, scratch);
    hipLaunchKernelGGL(k_mask_write, dim3(D, B), dim3(kThreads), 0, st, mask, wpp, W, scratch, xyz, counts, (long long)cap);
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}

PCC_API int pcc_threshold_compact(pcc_ctx* ctx, const float* x, int32_t B, int32_t D, int32_t H, int32_t W,
                                  const float* thr, int32_t clip, float* xyz, int32_t* counts, int64_t cap,
                                  int32_t* scratch, void* stream) {
    PCC_REQUIRE(ctx && x && thr && xyz && counts && scratch, "pcc_threshold_compact: NULL argument");
    PCC_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && cap >= 0, "pcc_threshold_compact: bad dimension");
    PCC_REQUIRE(B <= 65535, "pcc_threshold_compact: at most 65535 blocks per call");
    PCC_CHECK_HIP(hipSetDevice(ctx->device));
    const size_t nvox = (size_t)D * H * W;
    const int chunks = (int)((nvox + kChunk - 1) / kChunk);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_thr_count, dim3(chunks, B), dim3(kThreads), 0, st, x, thr, clip, nvox, chunks, scratch);
    hipLaunchKernelGGL(k_thr_write, dim3(chunks, B), dim3(kThreads), 0, st, x, thr, clip, nvox, chunks, H, W, scratch,
                       xyz, counts, (long long)cap);
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}

PCC_API size_t pcc_focal_scratch_floats(void) { return kFocalBlocks; }

PCC_API int pcc_focal_loss(pcc_ctx* ctx, const float* y_true, const float* y_pred, size_t n, float gamma, float alpha,
                           float* out, float* scratch, void* stream) {
    PCC_REQUIRE(ctx && y_true && y_pred && out && scratch, "pcc_focal_loss: NULL argument");
    PCC_CHECK_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_focal_partial, dim3(kFocalBlocks), dim3(kThreads), 0, st, y_true, y_pred, n, gamma, alpha, scratch);
    hipLaunchKernelGGL(k_focal_final, dim3(1), dim3(kThreads), 0, st, scratch, kFocalBlocks, out);
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}
