// Index arithmetic of the radix-tree builder, written once for the kernels and for host-side checks.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MIPT_HD __host__ __device__ inline
#else
#define MIPT_HD inline
#endif

namespace mipt {

MIPT_HD int lbvh_clz32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clz(static_cast<int>(v));
#else
    return v ? __builtin_clz(v) : 32;
#endif
}

// spreads the low 10 bits of v to every third bit
MIPT_HD uint32_t lbvh_spread10(uint32_t v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// 30-bit Morton code of a point given as fractions of the scene box (clamped to [0, 1])
MIPT_HD uint32_t lbvh_morton30(float fx, float fy, float fz) {
    const float f[3] = {fx, fy, fz};
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {
        float v = f[a] * 1024.0f;
        v = v > 0.0f ? v : 0.0f;        // also catches NaN
        v = v < 1023.0f ? v : 1023.0f;
        q[a] = static_cast<uint32_t>(v);
    }
    return (lbvh_spread10(q[0]) << 2) | (lbvh_spread10(q[1]) << 1) | lbvh_spread10(q[2]);
}

// Length of the common prefix of the keys at sorted positions i and j; equal codes are told apart by their
// positions, so every key is distinct.  -1 when j lies outside [0, n).
MIPT_HD int lbvh_delta(const uint32_t* codes, int64_t n, int64_t i, int64_t j) {
    if (j < 0 || j >= n) return -1;
    const uint32_t ci = codes[i], cj = codes[j];
    if (ci != cj) return lbvh_clz32(ci ^ cj);
    return 32 + lbvh_clz32(static_cast<uint32_t>(i) ^ static_cast<uint32_t>(j));
}

// Internal node i (0 <= i < n - 1) of the radix tree over n sorted keys: the leaf range [first, last] it covers
// and the split position: its children cover [first, split] and [split + 1, last].
struct LbvhRange {
    int64_t first, last, split;
};
MIPT_HD LbvhRange lbvh_node_range(const uint32_t* codes, int64_t n, int64_t i) {
    const int d = lbvh_delta(codes, n, i, i + 1) - lbvh_delta(codes, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(codes, n, i, i - d);
    int64_t lmax = 2;
    while (lbvh_delta(codes, n, i, i + lmax * d) > dmin) lmax *= 2;
    int64_t l = 0;
    for (int64_t t = lmax / 2; t >= 1; t /= 2)
        if (lbvh_delta(codes, n, i, i + (l + t) * d) > dmin) l += t;
    const int64_t j = i + l * d;
    const int dnode = lbvh_delta(codes, n, i, j);
    int64_t s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (lbvh_delta(codes, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    LbvhRange r;
    r.first = d > 0 ? i : j;
    r.last = d > 0 ? j : i;
    r.split = i + s * d + (d < 0 ? -1 : 0);
    return r;
}

}   // namespace mipt
