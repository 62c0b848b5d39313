// Exact per-block D1 statistics for EVERY threshold at once (the search of src/model_opt.py:21-77, which the
// reference does on the CPU with up to 255 KD-tree builds + queries per block, src/utils/pc_metric.py:76-138).
//
// For block b with original points A and reconstruction x_hat, let level k(v) = #{t : x_hat[v] > thr[t]}, so
// that the decoded set at threshold t is B_t = {v : k(v) > t} (nested sets).  All point coordinates are integers,
// hence all squared distances are integers and the sums below are exact:
//     n_B(t)  = |B_t|
//     S_BA(t) = sum_{v in B_t} min_{a in A} |v - a|^2      (one EDT of A + a level histogram)
//     S_AB(t) = sum_{a in A}  min_{v in B_t} |a - v|^2     (EDT of every level set, evaluated at the points of A)
// Squared Euclidean distance transforms are separable: a 1-D two-sweep pass along z, then min-plus passes
// along y and x with early termination (a candidate at axis distance d cannot win once d^2 >= best).
// The host turns these integers into the reference's d1_* metrics and applies its selection logic unchanged.
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace {

constexpr unsigned short kInf = 0xFFFF;   // no set voxel on the line / plane seen so far
constexpr int kT = 256;

__global__ void __launch_bounds__(256) k_levels(const float* __restrict__ x, const float* __restrict__ thr, int nthr,
                                                int clip, size_t nvox, unsigned char* __restrict__ lev,
                                                int* __restrict__ maxlev) {
    __shared__ float tab[kT];
    __shared__ int smax;
    for (int j = threadIdx.x; j < nthr; j += blockDim.x) tab[j] = thr[j];
    if (threadIdx.x == 0) smax = 0;
    __syncthreads();
    const int b = blockIdx.y;
    int m = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvox; i += (size_t)gridDim.x * blockDim.x) {
        float v = x[(size_t)b * nvox + i];
        if (clip) v = fminf(fmaxf(v, 0.f), 1.f);
        // thresholds are increasing: k = number of thresholds strictly below v (float32 compare, like the codec)
        int lo = 0, hi = nthr;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (v > tab[mid]) lo = mid + 1; else hi = mid; }
        lev[(size_t)b * nvox + i] = (unsigned char)min(lo, 255);
        m = max(m, lo);
    }
    atomicMax(&smax, m);
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&maxlev[b], smax);
}

__global__ void __launch_bounds__(256) k_fill_int(int* p, int v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void __launch_bounds__(256) k_occupancy(const int* __restrict__ pts, const int* __restrict__ block_of,
                                                   long long npts, int D, int H, int W, unsigned char* __restrict__ occ) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npts; i += (long long)gridDim.x * blockDim.x) {
        const int x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
        if (x < 0 || x >= D || y < 0 || y >= H || z < 0 || z >= W) continue;
        occ[(((size_t)block_of[i] * D + x) * H + y) * W + z] = 1;
    }
}

// 1-D squared distance along z (the contiguous axis) to the nearest voxel with level > t.
// thread <-> (line (x,y), t); grid.z = block.  out: [b][t][x][y][z] uint16.
__global__ void __launch_bounds__(256) k_edt_z(const unsigned char* __restrict__ lev, const int* __restrict__ tcount,
                                               int tmax, int t0, int lines, int W, unsigned short* __restrict__ out) {
    const int b = blockIdx.z, tl = blockIdx.y, t = t0 + tl;      // tl: slot inside the resident chunk of `tmax` thresholds
    if (t >= tcount[b]) return;
    const int line = blockIdx.x * blockDim.x + threadIdx.x;
    if (line >= lines) return;
    const unsigned char* l = lev + ((size_t)b * lines + line) * W;
    unsigned short* o = out + (((size_t)b * tmax + tl) * lines + line) * W;
    int last = -100000;
    for (int z = 0; z < W; ++z) {           // forward sweep: distance to the previous set voxel
        if (l[z] > t) last = z;
        const int d = z - last;
        o[z] = d < 256 ? (unsigned short)(d * d) : kInf;
    }
    last = 100000;
    for (int z = W - 1; z >= 0; --z) {      // backward sweep
        if (l[z] > t) last = z;
        const int d = last - z;
        if (d < 256) { const unsigned short q = (unsigned short)(d * d); if (q < o[z]) o[z] = q; }
    }
}

// min-plus pass along an axis with stride `astride` (in elements) and length L:
//   out[p] = min_{q on the same line} (p - q)^2 + in[q]
// thread <-> one output element; innermost (contiguous) index fastest so that accesses stay coalesced.
__global__ void __launch_bounds__(256) k_edt_axis(const unsigned short* __restrict__ in, const int* __restrict__ tcount,
                                                  int tmax, int t0, size_t nvox, int L, int astride, unsigned short* __restrict__ out) {
    const int b = blockIdx.z, tl = blockIdx.y, t = t0 + tl;
    if (t >= tcount[b]) return;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvox) return;
    const size_t base = ((size_t)b * tmax + tl) * nvox;
    const int p = (int)((i / astride) % L);
    const unsigned short* c = in + base + i;
    unsigned best = c[0];
    for (int d = 1; d < L; ++d) {
        const unsigned dd = (unsigned)(d * d);
        if (dd >= best) break;                       // farther candidates cannot win any more
        if (p - d >= 0) { const unsigned v = c[-(ptrdiff_t)d * astride]; if (v != kInf && v + dd < best) best = v + dd; }
        if (p + d < L) { const unsigned v = c[(ptrdiff_t)d * astride]; if (v != kInf && v + dd < best) best = v + dd; }
    }
    out[base + i] = (unsigned short)min(best, (unsigned)kInf);
}

// ---- z and y passes in ONE kernel, linear time per line (round 6).  The two kernels above cost 0.51 of the 0.57 s a 190-block cloud spends in
// the adaptive search: k_edt_z walks its lines with a stride of one line between lanes (165 GB/s), k_edt_axis visits ~40 candidates per
// output element (its early exit needs best < d^2).  Here one wave owns the plane (b, t, x): (1) every line y of the plane becomes a bit
// mask of its set voxels (level > t) in LDS -- the 1-D distance along z is then two bit scans, f(y', z) = dz(mask[y'], z)^2, never stored;
// (2) lane z builds the lower envelope of the parabolas f(y', z) + (y - y')^2 over y' (Felzenszwalb & Huttenlocher's linear-time 1-D
// transform with the stack in LDS, [entry][lane]) and evaluates it at every y.  All comparisons are exact integer cross-multiplications
// (no intersection abscissa is ever divided out): the outputs are the same integers as k_edt_z + k_edt_axis, which stay as the fall-back
// for shapes this kernel does not take (W % 16 != 0 or an edge > 128) and for the x pass of the originals' transform.
constexpr int kBigDist = 1 << 20;
template <int NW>
__device__ __forceinline__ int zdist(const unsigned long long* m, int z) {
    if constexpr (NW == 1) {
        const unsigned long long lo = m[0] & (~0ull >> (63 - z)), hi = m[0] >> z;
        const int dl = lo ? z - (63 - __clzll((long long)lo)) : kBigDist, dr = hi ? __ffsll((long long)hi) - 1 : kBigDist;
        return dl < dr ? dl : dr;
    } else {
        const int w = z >> 6, bit = z & 63;
        const unsigned long long lo = m[w] & (~0ull >> (63 - bit)), hi = m[w] >> bit;
        int dl = kBigDist, dr = kBigDist;
        if (lo) dl = bit - (63 - __clzll((long long)lo));
        else if (w == 1 && m[0]) dl = z - (63 - __clzll((long long)m[0]));
        if (hi) dr = __ffsll((long long)hi) - 1;
        else if (w == 0 && m[1]) dr = 64 + __ffsll((long long)m[1]) - 1 - z;
        return dl < dr ? dl : dr;
    }
}
template <int NW, int HM>      // NW = 64-bit words per line (W <= 64 NW), HM = stack depth (H <= HM): 12.5 KB of LDS per wave at 64^3, 25 KB at 128^3
__global__ void __launch_bounds__(64) k_edt_zy(const unsigned char* __restrict__ lev, const int* __restrict__ tcount, int tmax, int t0,
                                               int D, int H, int W, unsigned short* __restrict__ out) {
    const int x = blockIdx.x, tl = blockIdx.y, b = blockIdx.z, t = t0 + tl;
    if (t >= tcount[b]) return;
    __shared__ unsigned long long mask[HM][NW];
    __shared__ unsigned char sv[HM][64];         // envelope stack of lane z: positions y' ...
    __shared__ unsigned short sF[HM][64];        // ... and F = f + y'^2 (<= 2 * 127^2)
    const int lane = threadIdx.x;
    for (int y = lane; y < H; y += 64) {
        const unsigned char* l = lev + (((size_t)b * D + x) * H + y) * W;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            unsigned long long m = 0;
            for (int j = 0; j < 64 && w * 64 + j < W; j += 16) {
                const uint4 q = *reinterpret_cast<const uint4*>(l + w * 64 + j);
                const unsigned qs[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if ((int)((qs[u] >> (8 * c)) & 255u) > t) m |= 1ull << (j + 4 * u + c);
            }
            mask[y][w] = m;
        }
    }
    __syncthreads();
    for (int z = lane; z < W; z += 64) {
        int k = -1;
        for (int q = 0; q < H; ++q) {
            const int dz = zdist<NW>(mask[q], z);
            if (dz >= kBigDist) continue;                    // line q has no set voxel
            const int Fq = dz * dz + q * q;
            // the top entry leaves when its segment is empty: s(v[k-1], v[k]) >= s(v[k], q), cross-multiplied (all differences of positions > 0)
            while (k >= 1) {
                const int vk = sv[k][lane], Fk = sF[k][lane], vj = sv[k - 1][lane], Fj = sF[k - 1][lane];
                if ((Fk - Fj) * (q - vk) >= (Fq - Fk) * (vk - vj)) --k; else break;
            }
            ++k;
            sv[k][lane] = (unsigned char)q;
            sF[k][lane] = (unsigned short)Fq;
        }
        unsigned short* o = out + ((((size_t)b * tmax + tl) * D + x) * H) * W + z;
        if (k < 0) {
            for (int p = 0; p < H; ++p) o[(size_t)p * W] = kInf;
            continue;
        }
        int j = 0, vj = sv[0][lane], Fj = sF[0][lane];
        for (int p = 0; p < H; ++p) {
            while (j < k) {                                  // the next parabola takes over where it is at least as low
                const int vn = sv[j + 1][lane], Fn = sF[j + 1][lane];
                if (Fn - 2 * p * vn <= Fj - 2 * p * vj) { ++j; vj = vn; Fj = Fn; } else break;
            }
            const int val = Fj - 2 * p * vj + p * p;
            o[(size_t)p * W] = (unsigned short)(val < (int)kInf ? val : (int)kInf);
        }
    }
}
static bool edt_zy_takes(int D, int H, int W) { return W % 16 == 0 && W <= 128 && H <= 128 && D >= 1; }

// last pass (along x = the slowest axis) evaluated only at the points of A: S_AB[b][t] += min_x' (x_a-x')^2 + g[x'][y_a][z_a]
__global__ void __launch_bounds__(256) k_edt_points(const unsigned short* __restrict__ g, const int* __restrict__ tcount,
                                                    int tmax, int t0, const int* __restrict__ pts, const int* __restrict__ block_of,
                                                    long long npts, int D, int H, int W,
                                                    unsigned long long* __restrict__ s_ab) {
    const int tl = blockIdx.y, t = t0 + tl;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long val = 0;
    int b = -1;
    if (i < npts) {
        b = block_of[i];
        if (t < tcount[b]) {
            const int xa = pts[i * 3], ya = pts[i * 3 + 1], za = pts[i * 3 + 2];
            const size_t hw = (size_t)H * W;
            const unsigned short* c = g + ((size_t)b * tmax + tl) * D * hw + (size_t)ya * W + za;
            unsigned best = c[(size_t)xa * hw];
            for (int d = 1; d < D; ++d) {
                const unsigned dd = (unsigned)(d * d);
                if (dd >= best) break;
                if (xa - d >= 0) { const unsigned v = c[(size_t)(xa - d) * hw]; if (v != kInf && v + dd < best) best = v + dd; }
                if (xa + d < D) { const unsigned v = c[(size_t)(xa + d) * hw]; if (v != kInf && v + dd < best) best = v + dd; }
            }
            val = best;   // level set t is non-empty for t < tcount[b], so best is finite
        } else b = -1;
    }
    // points are grouped by block: reduce within the wave when the whole wave belongs to one block, else use atomics
    const int b0 = __shfl(b, 0, 64);
    const bool uniform = __all(b == b0 || b == -1) && b0 >= 0;
    if (uniform) {
        for (int o = 32; o > 0; o >>= 1) val += __shfl_xor(val, o, 64);
        if ((threadIdx.x & 63) == 0 && val) atomicAdd(&s_ab[(size_t)b0 * kT + t], val);
    } else if (b >= 0 && val) {
        atomicAdd(&s_ab[(size_t)b * kT + t], val);
    }
}

// S_BA / n_B histograms by level: hist[b][k] += edt_A[v] for every voxel of level k >= 1
__global__ void __launch_bounds__(256) k_level_hist(const unsigned char* __restrict__ lev, const unsigned short* __restrict__ edt_a,
                                                    size_t nvox, unsigned long long* __restrict__ hsum,
                                                    unsigned long long* __restrict__ hcnt) {
    __shared__ unsigned long long ssum[kT];
    __shared__ unsigned int scnt[kT];
    const int b = blockIdx.y;
    for (int j = threadIdx.x; j < kT; j += blockDim.x) { ssum[j] = 0; scnt[j] = 0; }
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvox; i += (size_t)gridDim.x * blockDim.x) {
        const int k = lev[(size_t)b * nvox + i];
        if (k) {
            atomicAdd(&ssum[k], (unsigned long long)edt_a[(size_t)b * nvox + i]);
            atomicAdd(&scnt[k], 1u);
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < kT; j += blockDim.x) {
        if (scnt[j]) { atomicAdd(&hsum[(size_t)b * kT + j], ssum[j]); atomicAdd(&hcnt[(size_t)b * kT + j], (unsigned long long)scnt[j]); }
    }
}

}  // namespace

// Thresholds resident at a time: the level-set EDTs are computed in chunks so that their two uint16 grids per
// (block, threshold) stay within ~1 GiB (32 blocks of 64^3 -> 32 thresholds per chunk; 8 blocks of 128^3 -> 16).
static int chunk_thresholds(int32_t B, size_t nvox) {
    const size_t per_t = (size_t)B * nvox * 2 * 2;
    size_t tc = ((size_t)1 << 30) / (per_t ? per_t : 1);
    if (tc < 8) tc = 8;
    if (tc > (size_t)kT) tc = kT;
    return (int)tc;
}

PCC_API size_t pcc_d1_search_workspace_bytes(int32_t B, int32_t D, int32_t H, int32_t W) {
    const size_t nvox = (size_t)D * H * W;
    // levels + occupancy (u8), EDT of A ping/pong (u16), level-set EDT ping/pong (u16 x chunk), per-block counters
    return (size_t)B * nvox * 2 + (size_t)B * nvox * 2 * 2 + (size_t)B * chunk_thresholds(B, nvox) * nvox * 2 * 2 + (size_t)B * 64 + 4096;
}

// x_hat: (B,D,H,W) float32; thr: 256 float32 thresholds (device); pts: (npts,3) int32 local coordinates grouped by
// block, block_of: (npts,) int32.  Outputs (device, zero-filled by this call): s_ab, hsum, hcnt: (B,256) uint64;
// tcount: (B,) int32 = number of thresholds with a non-empty decoded set.  hsum/hcnt are per-level histograms:
// S_BA(t) = sum_{k>t} hsum[k], n_B(t) = sum_{k>t} hcnt[k].
PCC_API int pcc_d1_threshold_stats(pcc_ctx* ctx, const float* x_hat, int32_t B, int32_t D, int32_t H, int32_t W,
                                   const float* thr, int32_t nthr, int32_t clip, const int32_t* pts,
                                   const int32_t* block_of, int64_t npts, void* workspace, uint64_t* s_ab,
                                   uint64_t* hsum, uint64_t* hcnt, int32_t* tcount, void* stream) {
    PCC_REQUIRE(ctx && x_hat && thr && workspace && s_ab && hsum && hcnt && tcount, "pcc_d1_threshold_stats: NULL argument");
    PCC_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && npts >= 0 && (pts || npts == 0), "pcc_d1_threshold_stats: bad dimension");
    PCC_REQUIRE(nthr >= 1 && nthr <= kT, "pcc_d1_threshold_stats: at most 256 thresholds");
    PCC_REQUIRE(D <= 128 && H <= 128 && W <= 128, "pcc_d1_threshold_stats: blocks up to 128^3 (uint16 squared distances)");
    PCC_REQUIRE(B <= 65535, "pcc_d1_threshold_stats: at most 65535 blocks per call");
    PCC_CHECK_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t nvox = (size_t)D * H * W;
    unsigned char* lev = (unsigned char*)workspace;
    unsigned char* occ = lev + (size_t)B * nvox;
    unsigned short* ea0 = (unsigned short*)(occ + (size_t)B * nvox);
    unsigned short* ea1 = ea0 + (size_t)B * nvox;
    unsigned short* g0 = ea1 + (size_t)B * nvox;
    const int TC = chunk_thresholds(B, nvox);
    unsigned short* g1 = g0 + (size_t)B * TC * nvox;
    int* one = (int*)(g1 + (size_t)B * TC * nvox);   // per-block "1 threshold" counter for the EDT of A

    PCC_CHECK_HIP(hipMemsetAsync(occ, 0, (size_t)B * nvox, st));
    PCC_CHECK_HIP(hipMemsetAsync(tcount, 0, (size_t)B * sizeof(int), st));
    PCC_CHECK_HIP(hipMemsetAsync(s_ab, 0, (size_t)B * kT * 8, st));
    PCC_CHECK_HIP(hipMemsetAsync(hsum, 0, (size_t)B * kT * 8, st));
    PCC_CHECK_HIP(hipMemsetAsync(hcnt, 0, (size_t)B * kT * 8, st));
    const int lines = D * H;
    const unsigned vox_blocks = (unsigned)((nvox + 255) / 256);
    hipLaunchKernelGGL(k_levels, dim3(256, B), dim3(256), 0, st, x_hat, thr, nthr, clip, nvox, lev, tcount);
    // ---- EDT of the original points A (one "threshold": occupancy > 0)
    if (npts > 0) {
        unsigned pblocks = (unsigned)((npts + 255) / 256);
        if (pblocks > 65535u * 16u) pblocks = 65535u * 16u;
        hipLaunchKernelGGL(k_occupancy, dim3(pblocks), dim3(256), 0, st, pts, block_of, (long long)npts, D, H, W, occ);
    }
    hipLaunchKernelGGL(k_fill_int, dim3((B + 255) / 256), dim3(256), 0, st, one, 1, B);
    const bool fused = edt_zy_takes(D, H, W) && !getenv("PCC_EDT_OLD");
    auto launch_zy = [&](const unsigned char* levels, const int* counts, int tmax_, int t0_, int nt_, unsigned short* dst) {
        if (W <= 64 && H <= 64) hipLaunchKernelGGL((k_edt_zy<1, 64>), dim3(D, nt_, B), dim3(64), 0, st, levels, counts, tmax_, t0_, D, H, W, dst);
        else if (W <= 64) hipLaunchKernelGGL((k_edt_zy<1, 128>), dim3(D, nt_, B), dim3(64), 0, st, levels, counts, tmax_, t0_, D, H, W, dst);
        else hipLaunchKernelGGL((k_edt_zy<2, 128>), dim3(D, nt_, B), dim3(64), 0, st, levels, counts, tmax_, t0_, D, H, W, dst);
    };
    if (fused) launch_zy(occ, one, 1, 0, 1, ea1);
    else {
        hipLaunchKernelGGL(k_edt_z, dim3((lines + 255) / 256, 1, B), dim3(256), 0, st, occ, one, 1, 0, lines, W, ea0);
        hipLaunchKernelGGL(k_edt_axis, dim3(vox_blocks, 1, B), dim3(256), 0, st, ea0, one, 1, 0, nvox, H, W, ea1);
    }
    hipLaunchKernelGGL(k_edt_axis, dim3(vox_blocks, 1, B), dim3(256), 0, st, ea1, one, 1, 0, nvox, D, H * W, ea0);
    hipLaunchKernelGGL(k_level_hist, dim3(64, B), dim3(256), 0, st, lev, ea0, nvox, (unsigned long long*)hsum,
                       (unsigned long long*)hcnt);
    // ---- EDT of every level set, evaluated at the points of A
    //      in chunks of TC thresholds (workspace bound); chunks beyond every block's tcount exit at once
    for (int t0 = 0; t0 < nthr; t0 += TC) {
        const int nt = nthr - t0 < TC ? nthr - t0 : TC;
        if (fused) launch_zy(lev, tcount, TC, t0, nt, g1);
        else {
            hipLaunchKernelGGL(k_edt_z, dim3((lines + 255) / 256, nt, B), dim3(256), 0, st, lev, tcount, TC, t0, lines, W, g0);
            hipLaunchKernelGGL(k_edt_axis, dim3(vox_blocks, nt, B), dim3(256), 0, st, g0, tcount, TC, t0, nvox, H, W, g1);
        }
        if (npts > 0) {
            const unsigned pblocks = (unsigned)((npts + 255) / 256);
            hipLaunchKernelGGL(k_edt_points, dim3(pblocks, nt), dim3(256), 0, st, g1, tcount, TC, t0, pts, block_of,
                               (long long)npts, D, H, W, (unsigned long long*)s_ab);
        }
    }
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}


// =====================================================================================================================
// D2 (point-to-plane) statistics on the GPU (round 4; src/utils/pc_metric.py:109-131 inside the search of src/model_opt.py:33-73).
//
// D2 needs WHICH point is nearest, not only how far it is: nearest-INDEX transforms replace the distance transforms.  Ties are
// the rule on a voxel grid and the reference's own numbers depend on the pick of scipy's KD-tree (pc_metric.py:114), so the rule
// here is stated and deterministic: among equidistant candidates the one with the LOWEST (x, y, z) in lexicographic order (= the
// lowest row-major voxel index, the order of np.argwhere).  Per pass of the separable transform that is "smaller coordinate
// wins a tie", which composes to the lexicographic rule (DESIGN_HISTORY.md 3.8).
//   B -> A (once per block): full-grid index transform of A -> a*(v); e(v) = ((v - a*) . n[a*])^2 in fp64;
//        D2_BA(t) = sum_{v : level(v) > t} e(v), one workgroup per (block, t), fixed summation order.
//   A -> B (per threshold):  z and y passes on the grid, x pass at the points of A -> b*(a, t);  the decoded point b* gets the MEAN
//        normal of the original points that chose it, summed in ascending point order like the reference's loop (pc_metric.py:16-18):
//        one stable radix sort of the (block, t, b*) keys of a whole chunk of thresholds groups them;
//        D2_AB(t) = sum_a ((a - b*) . mean_n(b*))^2, one workgroup per (block, t), fixed order.
// No float atomics anywhere: the results are bit-reproducible.
namespace {

constexpr unsigned char kNo8 = 0xFF;
constexpr unsigned short kNo16 = 0xFFFF;

// nearest set voxel along z (level > t), ties -> the smaller z.  thread <-> (line (x,y), t); out: [b][tl][x][y][z] uint8
__global__ void __launch_bounds__(256) k_ft_z(const unsigned char* __restrict__ lev, const int* __restrict__ tcount, int tmax, int t0,
                                              int lines, int W, unsigned char* __restrict__ out) {
    const int b = blockIdx.z, tl = blockIdx.y, t = t0 + tl;
    if (t >= tcount[b]) return;
    const int line = blockIdx.x * blockDim.x + threadIdx.x;
    if (line >= lines) return;
    const unsigned char* l = lev + ((size_t)b * lines + line) * W;
    unsigned char* o = out + (((size_t)b * tmax + tl) * lines + line) * W;
    int last = -1;
    for (int z = 0; z < W; ++z) {
        if (l[z] > t) last = z;
        o[z] = last < 0 ? kNo8 : (unsigned char)last;
    }
    int nxt = -1;
    for (int z = W - 1; z >= 0; --z) {
        if (l[z] > t) nxt = z;
        const int prev = o[z] == kNo8 ? -1 : (int)o[z];
        if (nxt >= 0 && (prev < 0 || nxt - z < z - prev)) o[z] = (unsigned char)nxt;        // strictly nearer only: a tie keeps the smaller z
    }
}

// y pass: (y*, z*) minimising (y - y')^2 + (z - z*(x, y', z))^2, ties -> the smaller y'.  thread <-> output voxel
__global__ void __launch_bounds__(256) k_ft_y(const unsigned char* __restrict__ in, const int* __restrict__ tcount, int tmax, int t0,
                                              size_t nvox, int H, int W, unsigned short* __restrict__ out) {
    const int b = blockIdx.z, tl = blockIdx.y, t = t0 + tl;
    if (t >= tcount[b]) return;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvox) return;
    const size_t base = ((size_t)b * tmax + tl) * nvox;
    const int z = (int)(i % W), y = (int)((i / W) % H);
    const unsigned char* c = in + base + i;
    unsigned best = 0xFFFFFFFFu, by = 0, bz = 0;
    if (c[0] != kNo8) { const int dz = z - c[0]; best = (unsigned)(dz * dz); by = (unsigned)y; bz = c[0]; }
    for (int d = 1; d < H; ++d) {
        const unsigned dd = (unsigned)(d * d);
        if (dd > best) break;                               // (>: a candidate at dd == best can still tie with a smaller y')
        if (y - d >= 0) {
            const unsigned char zs = c[-(ptrdiff_t)d * W];
            if (zs != kNo8) { const int dz = z - zs; const unsigned tot = dd + (unsigned)(dz * dz); if (tot <= best) { best = tot; by = (unsigned)(y - d); bz = zs; } }
        }
        if (y + d < H) {
            const unsigned char zs = c[(ptrdiff_t)d * W];
            if (zs != kNo8) { const int dz = z - zs; const unsigned tot = dd + (unsigned)(dz * dz); if (tot < best) { best = tot; by = (unsigned)(y + d); bz = zs; } }
        }
    }
    out[base + i] = best == 0xFFFFFFFFu ? kNo16 : (unsigned short)((by << 8) | bz);
}

// x pass over the full grid (index transform of A): out = x* << 16 | y* << 8 | z*, 0xFFFFFFFF = empty block
__global__ void __launch_bounds__(256) k_ft_x_full(const unsigned short* __restrict__ in, size_t nvox, int D, int H, int W,
                                                   unsigned* __restrict__ out) {
    const int b = blockIdx.z;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvox) return;
    const size_t hw = (size_t)H * W;
    const int z = (int)(i % W), y = (int)((i / W) % H), x = (int)(i / hw);
    const unsigned short* c = in + (size_t)b * nvox + i;
    unsigned best = 0xFFFFFFFFu, arg = 0xFFFFFFFFu;
    auto cand = [&](int xs, unsigned dd, bool tie_wins) {
        const unsigned short e = c[((ptrdiff_t)xs - x) * (ptrdiff_t)hw];
        if (e == kNo16) return;
        const int dy = y - (e >> 8), dz = z - (e & 255);
        const unsigned tot = dd + (unsigned)(dy * dy + dz * dz);
        if (tot < best || (tie_wins && tot == best)) { best = tot; arg = ((unsigned)xs << 16) | e; }
    };
    cand(x, 0, false);
    for (int d = 1; d < D; ++d) {
        const unsigned dd = (unsigned)(d * d);
        if (dd > best) break;
        if (x - d >= 0) cand(x - d, dd, true);
        if (x + d < D) cand(x + d, dd, false);
    }
    out[(size_t)b * nvox + i] = arg;
}

// point index of every occupied voxel (lowest index when a voxel holds several points)
__global__ void __launch_bounds__(256) k_index_grid(const int* __restrict__ pts, const int* __restrict__ block_of, long long npts,
                                                    int D, int H, int W, int* __restrict__ grid) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npts) return;
    const int x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    if (x < 0 || x >= D || y < 0 || y >= H || z < 0 || z >= W) return;
    atomicMin(&grid[(((size_t)block_of[i] * D + x) * H + y) * W + z], (int)i);
}

// e(v) = ((v - a*) . n[a*])^2 for the voxels that can be decoded at all (level >= 1)
__global__ void __launch_bounds__(256) k_plane_err_ba(const unsigned char* __restrict__ lev, const unsigned* __restrict__ fta,
                                                      const int* __restrict__ idxgrid, const float* __restrict__ normals,
                                                      size_t nvox, int D, int H, int W, double* __restrict__ e) {
    const int b = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvox) return;
    double val = 0.0;
    const unsigned a = fta[(size_t)b * nvox + i];
    if (lev[(size_t)b * nvox + i] && a != 0xFFFFFFFFu) {
        const int xs = (int)(a >> 16), ys = (int)((a >> 8) & 255), zs = (int)(a & 255);
        const size_t hw = (size_t)H * W;
        const int z = (int)(i % W), y = (int)((i / W) % H), x = (int)(i / hw);
        const int pi = idxgrid[(size_t)b * nvox + ((size_t)xs * H + ys) * W + zs];
        const double proj = (double)(x - xs) * (double)normals[(size_t)pi * 3] + (double)(y - ys) * (double)normals[(size_t)pi * 3 + 1] +
                            (double)(z - zs) * (double)normals[(size_t)pi * 3 + 2];
        val = proj * proj;
    }
    e[(size_t)b * nvox + i] = val;
}

// fixed-order block reduction of one double per thread (256 threads)
__device__ __forceinline__ double block_sum_256(double v, double* sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    return sh[0];
}

// D2_BA[b][t] = sum of e over the voxels with level > t.  One workgroup per (t, block); thread k adds voxels k, k + 256, ... in order
__global__ void __launch_bounds__(256) k_d2_ba(const unsigned char* __restrict__ lev, const double* __restrict__ e,
                                               const int* __restrict__ tcount, size_t nvox, double* __restrict__ d2_ba) {
    __shared__ double sh[256];
    const int t = blockIdx.x, b = blockIdx.y;
    if (t >= tcount[b]) return;
    double s = 0.0;
    for (size_t i = threadIdx.x; i < nvox; i += 256)
        if (lev[(size_t)b * nvox + i] > t) s += e[(size_t)b * nvox + i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) d2_ba[(size_t)b * kT + t] = s;
}

// x pass at the points of A: b*(a, t) (row-major voxel index) as the key (block, t, b*) of the grouping sort
__global__ void __launch_bounds__(256) k_ft_points(const unsigned short* __restrict__ g, const int* __restrict__ tcount, int tmax, int t0,
                                                   const int* __restrict__ pts, const int* __restrict__ block_of, long long npts,
                                                   int D, int H, int W, unsigned long long* __restrict__ keys, unsigned* __restrict__ vals) {
    const int tl = blockIdx.y, t = t0 + tl;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long key = ~0ull;
    if (i < npts) {
        const int b = block_of[i];
        if (t < tcount[b]) {
            const int xa = pts[i * 3], ya = pts[i * 3 + 1], za = pts[i * 3 + 2];
            const size_t hw = (size_t)H * W;
            const unsigned short* c = g + ((size_t)b * tmax + tl) * D * hw + (size_t)ya * W + za;
            unsigned best = 0xFFFFFFFFu, arg = 0;
            auto cand = [&](int xs, unsigned dd, bool tie_wins) {
                const unsigned short e = c[(size_t)xs * hw];
                if (e == kNo16) return;
                const int dy = ya - (e >> 8), dz = za - (e & 255);
                const unsigned tot = dd + (unsigned)(dy * dy + dz * dz);
                if (tot < best || (tie_wins && tot == best)) { best = tot; arg = (unsigned)(((size_t)xs * H + (e >> 8)) * W + (e & 255)); }
            };
            cand(xa, 0, false);
            for (int d = 1; d < D; ++d) {
                const unsigned dd = (unsigned)(d * d);
                if (dd > best) break;
                if (xa - d >= 0) cand(xa - d, dd, true);
                if (xa + d < D) cand(xa + d, dd, false);
            }
            key = ((unsigned long long)b << 32) | ((unsigned long long)tl << 24) | arg;      // arg < 2^21 (128^3), tl < 64
        }
        keys[(size_t)tl * npts + i] = key;
        vals[(size_t)tl * npts + i] = (unsigned)i;
    }
}

// sorted (block, t, b*) groups: the head of a group sums the normals of its members in ascending point order (the stable sort kept
// it), every member gets ((a - b*) . mean)^2 written at its own (t, point) slot
__global__ void __launch_bounds__(256) k_group_plane_err(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ vals,
                                                         size_t n, const int* __restrict__ pts, const float* __restrict__ normals,
                                                         long long npts, int H, int W, double* __restrict__ err) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const unsigned long long key = keys[j];
    if (key == ~0ull || (j > 0 && keys[j - 1] == key)) return;          // not a group head
    double sx = 0.0, sy = 0.0, sz = 0.0, cnt = 0.0;
    size_t end = j;
    for (; end < n && keys[end] == key; ++end) {
        const size_t pi = vals[end];
        sx += (double)normals[pi * 3]; sy += (double)normals[pi * 3 + 1]; sz += (double)normals[pi * 3 + 2]; cnt += 1.0;
    }
    sx /= cnt; sy /= cnt; sz /= cnt;
    const unsigned arg = (unsigned)(key & 0xFFFFFFu), tl = (unsigned)((key >> 24) & 0xFF);
    const int zs = (int)(arg % (unsigned)W), ys = (int)((arg / (unsigned)W) % (unsigned)H), xs = (int)(arg / (unsigned)(W * H));
    for (size_t m = j; m < end; ++m) {
        const size_t pi = vals[m];
        const double proj = (double)(pts[pi * 3] - xs) * sx + (double)(pts[pi * 3 + 1] - ys) * sy + (double)(pts[pi * 3 + 2] - zs) * sz;
        err[(size_t)tl * npts + pi] = proj * proj;
    }
}

// D2_AB[b][t0 + tl] = sum over the points of block b of err[tl][.], fixed order
__global__ void __launch_bounds__(256) k_d2_ab(const double* __restrict__ err, const int* __restrict__ block_start, const int* __restrict__ tcount,
                                               int t0, long long npts, double* __restrict__ d2_ab) {
    __shared__ double sh[256];
    const int tl = blockIdx.x, b = blockIdx.y, t = t0 + tl;
    if (t >= tcount[b]) return;
    double s = 0.0;
    for (long long i = block_start[b] + threadIdx.x; i < block_start[b + 1]; i += 256) s += err[(size_t)tl * npts + i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) d2_ab[(size_t)b * kT + t] = s;
}

struct D2Layout {
    size_t ftz, fty, fta, idx, e, keys0, keys1, vals0, vals1, err, sort_tmp, sort_tmp_bytes, total;
    int TC;
};
D2Layout d2_layout(int32_t B, size_t nvox, int64_t npts) {
    D2Layout l;
    // thresholds resident at a time: 3 bytes per voxel of index grids, 32 bytes per point of sort / error buffers -> ~1 GiB
    size_t tc = ((size_t)1 << 30) / ((size_t)B * nvox * 3 + (size_t)npts * 32 + 1);
    l.TC = (int)(tc < 4 ? 4 : tc > 64 ? 64 : tc);
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t o = 0;
    l.ftz = o; o += al((size_t)B * l.TC * nvox);
    l.fty = o; o += al((size_t)B * l.TC * nvox * 2);
    l.fta = o; o += al((size_t)B * nvox * 4);
    l.idx = o; o += al((size_t)B * nvox * 4);
    l.e = o; o += al((size_t)B * nvox * 8);
    const size_t pairs = (size_t)l.TC * (size_t)npts;
    l.keys0 = o; o += al(pairs * 8);
    l.keys1 = o; o += al(pairs * 8);
    l.vals0 = o; o += al(pairs * 4);
    l.vals1 = o; o += al(pairs * 4);
    l.err = o; o += al(pairs * 8);
    size_t tmp = 0;
    (void)hipcub::DeviceRadixSort::SortPairs((void*)nullptr, tmp, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (const unsigned*)nullptr,
                                       (unsigned*)nullptr, (int)(pairs ? pairs : 1), 0, 48, (hipStream_t)0);
    l.sort_tmp_bytes = tmp;
    l.sort_tmp = o; o += al(tmp + 256);
    l.total = o + 4096;
    return l;
}

}  // namespace

PCC_API size_t pcc_d12_search_workspace_bytes(int32_t B, int32_t D, int32_t H, int32_t W, int64_t npts) {
    return d2_layout(B, (size_t)D * H * W, npts).total;
}

// As pcc_d1_threshold_stats (same D1 outputs, computed by the same kernels), plus the D2 sums of every threshold:
//   normals: (npts, 3) float32 (device), block_start: (B + 1,) int32 offsets of the blocks' points in pts (device),
//   workspace: pcc_d1_search_workspace_bytes, workspace2: pcc_d12_search_workspace_bytes,
//   d2_ab, d2_ba: (B, 256) float64 (device), entry [b][t] valid for t < tcount[b].
PCC_API int pcc_d12_threshold_stats(pcc_ctx* ctx, const float* x_hat, int32_t B, int32_t D, int32_t H, int32_t W, const float* thr,
                                    int32_t nthr, int32_t clip, const int32_t* pts, const int32_t* block_of, const int32_t* block_start,
                                    int64_t npts, const float* normals, void* workspace, void* workspace2, uint64_t* s_ab, uint64_t* hsum,
                                    uint64_t* hcnt, int32_t* tcount, double* d2_ab, double* d2_ba, void* stream) {
    PCC_REQUIRE(normals && block_start && workspace2 && d2_ab && d2_ba && pts && npts > 0, "pcc_d12_threshold_stats: NULL argument");
    PCC_REQUIRE((size_t)npts * 64 < ((size_t)1 << 31), "pcc_d12_threshold_stats: too many points for one call");
    // D1 part (and the levels / tcount the D2 part builds on): unchanged kernels
    { const int rc = pcc_d1_threshold_stats(ctx, x_hat, B, D, H, W, thr, nthr, clip, pts, block_of, npts, workspace, s_ab, hsum, hcnt, tcount, stream);
      if (rc != PCC_OK) return rc; }
    hipStream_t st = (hipStream_t)stream;
    const size_t nvox = (size_t)D * H * W;
    const D2Layout l = d2_layout(B, nvox, npts);
    unsigned char* w2 = (unsigned char*)workspace2;
    unsigned char* ftz = w2 + l.ftz;
    unsigned short* fty = (unsigned short*)(w2 + l.fty);
    unsigned* fta = (unsigned*)(w2 + l.fta);
    int* idx = (int*)(w2 + l.idx);
    double* e = (double*)(w2 + l.e);
    unsigned long long *keys0 = (unsigned long long*)(w2 + l.keys0), *keys1 = (unsigned long long*)(w2 + l.keys1);
    unsigned *vals0 = (unsigned*)(w2 + l.vals0), *vals1 = (unsigned*)(w2 + l.vals1);
    double* err = (double*)(w2 + l.err);
    // buffers of the D1 call that are still valid: levels and occupancy at the start of `workspace`, and its "one" counter
    unsigned char* lev = (unsigned char*)workspace;
    unsigned char* occ = lev + (size_t)B * nvox;
    const int TC1 = chunk_thresholds(B, nvox);
    int* one = (int*)((unsigned short*)(occ + (size_t)B * nvox) + (size_t)B * nvox * 2 + (size_t)B * TC1 * nvox * 2);
    const int lines = D * H;
    const unsigned vox_blocks = (unsigned)((nvox + 255) / 256);
    const unsigned pblocks = (unsigned)((npts + 255) / 256);
    PCC_CHECK_HIP(hipMemsetAsync(d2_ab, 0, (size_t)B * kT * 8, st));
    PCC_CHECK_HIP(hipMemsetAsync(d2_ba, 0, (size_t)B * kT * 8, st));
    // ---- B -> A: index transform of the original points (occupancy as a one-threshold level set), plane error per voxel
    PCC_CHECK_HIP(hipMemsetAsync(idx, 0x7F, (size_t)B * nvox * 4, st));
    hipLaunchKernelGGL(k_index_grid, dim3(pblocks), dim3(256), 0, st, pts, block_of, (long long)npts, D, H, W, idx);
    hipLaunchKernelGGL(k_ft_z, dim3((lines + 255) / 256, 1, B), dim3(256), 0, st, occ, one, 1, 0, lines, W, ftz);
    hipLaunchKernelGGL(k_ft_y, dim3(vox_blocks, 1, B), dim3(256), 0, st, ftz, one, 1, 0, nvox, H, W, fty);
    hipLaunchKernelGGL(k_ft_x_full, dim3(vox_blocks, 1, B), dim3(256), 0, st, fty, nvox, D, H, W, fta);
    hipLaunchKernelGGL(k_plane_err_ba, dim3(vox_blocks, B), dim3(256), 0, st, lev, fta, idx, normals, nvox, D, H, W, e);
    hipLaunchKernelGGL(k_d2_ba, dim3(nthr, B), dim3(256), 0, st, lev, e, tcount, nvox, d2_ba);
    // ---- A -> B per chunk of thresholds (the D1 sums of this direction came from the D1 call above)
    for (int t0 = 0; t0 < nthr; t0 += l.TC) {
        const int nt = nthr - t0 < l.TC ? nthr - t0 : l.TC;
        const size_t pairs = (size_t)nt * (size_t)npts;
        hipLaunchKernelGGL(k_ft_z, dim3((lines + 255) / 256, nt, B), dim3(256), 0, st, lev, tcount, l.TC, t0, lines, W, ftz);
        hipLaunchKernelGGL(k_ft_y, dim3(vox_blocks, nt, B), dim3(256), 0, st, ftz, tcount, l.TC, t0, nvox, H, W, fty);
        hipLaunchKernelGGL(k_ft_points, dim3(pblocks, nt), dim3(256), 0, st, fty, tcount, l.TC, t0, pts, block_of, (long long)npts, D, H, W,
                           keys0, vals0);
        size_t tmp = l.sort_tmp_bytes;
        PCC_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs((void*)(w2 + l.sort_tmp), tmp, keys0, keys1, vals0, vals1, (int)pairs, 0, 48, st));
        hipLaunchKernelGGL(k_group_plane_err, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, keys1, vals1, pairs, pts, normals,
                           (long long)npts, H, W, err);
        hipLaunchKernelGGL(k_d2_ab, dim3(nt, B), dim3(256), 0, st, err, block_start, tcount, t0, (long long)npts, d2_ab);
    }
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}
