// Host side of the octree blocking: Morton bucketing of a point cloud into blocks.
// Replaces the per-point Python loop of /root/reference/src/utils/octree_coding.py:82-108 (block id = pt // block_size,
// Morton key with x least significant, points appended to their block in input order) by one counting sort.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/pcc_geo.h"

extern void pcc_set_error(const char* fmt, ...);

extern "C" __attribute__((visibility("default")))
int64_t pcc_octree_bucket(const double* points, int64_t n, int32_t ncols, int32_t block_size, int32_t level,
                          int64_t* order, int64_t* bucket_count) {
    if (!points || !order || !bucket_count || n < 0 || ncols < 3 || block_size < 1 || level < 1 || level > 7) {
        pcc_set_error("pcc_octree_bucket: bad argument (level must be 1..7)");
        return PCC_ERR_ARG;
    }
    const int64_t nb = (int64_t)1 << (3 * level);
    const int64_t side = (int64_t)1 << level;
    std::vector<uint32_t> key((size_t)n);
    std::memset(bucket_count, 0, sizeof(int64_t) * (size_t)nb);
    for (int64_t i = 0; i < n; ++i) {
        const double* p = points + i * ncols;
        const int64_t bx = (int64_t)(p[0] / block_size), by = (int64_t)(p[1] / block_size), bz = (int64_t)(p[2] / block_size);
        if (p[0] < 0 || p[1] < 0 || p[2] < 0 || bx >= side || by >= side || bz >= side) {
            pcc_set_error("pcc_octree_bucket: point %lld lies outside the bounding box", (long long)i);
            return PCC_ERR_ARG;
        }
        uint32_t k = 0;
        for (int b = level - 1; b >= 0; --b)
            k = (k << 3) | (uint32_t)((((bz >> b) & 1) << 2) | (((by >> b) & 1) << 1) | ((bx >> b) & 1));
        key[(size_t)i] = k;
        ++bucket_count[k];
    }
    // exclusive prefix sums -> stable scatter (input order is kept inside a block)
    std::vector<int64_t> pos((size_t)nb);
    int64_t acc = 0, occupied = 0;
    for (int64_t b = 0; b < nb; ++b) { pos[(size_t)b] = acc; acc += bucket_count[b]; occupied += bucket_count[b] != 0; }
    for (int64_t i = 0; i < n; ++i) order[pos[key[(size_t)i]]++] = i;
    return occupied;
}
