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
    hipLaunchKernelGGL(k_edt_z, dim3((lines + 255) / 256, 1, B), dim3(256), 0, st, occ, one, 1, 0, lines, W, ea0);
    hipLaunchKernelGGL(k_edt_axis, dim3(vox_blocks, 1, B), dim3(256), 0, st, ea0, one, 1, 0, nvox, H, W, ea1);
    hipLaunchKernelGGL(k_edt_axis, dim3(vox_blocks, 1, B), dim3(256), 0, st, ea1, one, 1, 0, nvox, D, H * W, ea0);
    hipLaunchKernelGGL(k_level_hist, dim3(64, B), dim3(256), 0, st, lev, ea0, nvox, (unsigned long long*)hsum,
                       (unsigned long long*)hcnt);
    // ---- EDT of every level set, evaluated at the points of A
    //      in chunks of TC thresholds (workspace bound); chunks beyond every block's tcount exit at once
    for (int t0 = 0; t0 < nthr; t0 += TC) {
        const int nt = nthr - t0 < TC ? nthr - t0 : TC;
        hipLaunchKernelGGL(k_edt_z, dim3((lines + 255) / 256, nt, B), dim3(256), 0, st, lev, tcount, TC, t0, lines, W, g0);
        hipLaunchKernelGGL(k_edt_axis, dim3(vox_blocks, nt, B), dim3(256), 0, st, g0, tcount, TC, t0, nvox, H, W, g1);
        if (npts > 0) {
            const unsigned pblocks = (unsigned)((npts + 255) / 256);
            hipLaunchKernelGGL(k_edt_points, dim3(pblocks, nt), dim3(256), 0, st, g1, tcount, TC, t0, pts, block_of,
                               (long long)npts, D, H, W, (unsigned long long*)s_ab);
        }
    }
    PCC_CHECK_HIP(hipGetLastError());
    return PCC_OK;
}
