// Host side of the bounding-sphere hierarchy (rt_hier.h): grouping the spheres and filling
// the float32 screening tables of the three levels.
#pragma once
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "rt_hier.h"

namespace rt {

struct HostHierarchy {
    std::vector<Group8> top, mid, leaf;
    std::vector<uint32_t> top_ref;
    std::vector<uint16_t> ids;
    int blocks = 0, loose = 0;
    std::vector<double> block_bound;   // [blocks][4] bounding spheres of the blocks
    double typical_radius = 0;         // median sphere radius
};

namespace hier_detail {

// Reorders idx[0..n) so that every aligned group of `unit` consecutive entries is spatially
// compact: median split along the longest axis of the centres, the left part a multiple of the
// group size, down to single groups.
inline void split(const double* geom, uint32_t* idx, size_t n, size_t unit) {
    if (n <= unit) return;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], geom[idx[i] * 4 + k]);
            hi[k] = std::max(hi[k], geom[idx[i] * 4 + k]);
        }
    int axis = 0;
    for (int k = 1; k < 3; ++k)
        if (hi[k] - lo[k] > hi[axis] - lo[axis]) axis = k;
    const size_t groups = (n + unit - 1) / unit;
    const size_t left = (groups / 2) * unit;   // >= unit because groups >= 2
    std::nth_element(idx, idx + left, idx + n, [&](uint32_t a, uint32_t b) {
        const double va = geom[a * 4 + axis], vb = geom[b * 4 + axis];
        return va < vb || (va == vb && a < b);
    });
    split(geom, idx, left, unit);
    split(geom, idx + left, n - left, unit);
}

// Bounding sphere (cx, cy, cz, r) of spheres given as (cx, cy, cz, r) quadruples: centre of the
// members' bounding box, radius the largest distance to a member's far side.
inline void bound(const double* members, size_t n, double* out) {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], members[i * 4 + k] - fabs(members[i * 4 + 3]));
            hi[k] = std::max(hi[k], members[i * 4 + k] + fabs(members[i * 4 + 3]));
        }
    for (int k = 0; k < 3; ++k) out[k] = 0.5 * (lo[k] + hi[k]);
    double r = 0;
    for (size_t i = 0; i < n; ++i) {
        const double dx = members[i * 4] - out[0], dy = members[i * 4 + 1] - out[1], dz = members[i * 4 + 2] - out[2];
        r = std::max(r, sqrt(dx * dx + dy * dy + dz * dz) + fabs(members[i * 4 + 3]));
    }
    out[3] = r * (1 + 1e-9) + 1e-300;
}

}  // namespace hier_detail

// geom: [S][4] centre and radius.  Spheres with a radius far above the typical one stay loose.
inline HostHierarchy build_hierarchy(const double* geom, size_t S) {
    using namespace hier_detail;
    HostHierarchy H;
    std::vector<double> radii(S);
    for (size_t i = 0; i < S; ++i) radii[i] = fabs(geom[i * 4 + 3]);
    std::vector<double> sorted = radii;
    std::nth_element(sorted.begin(), sorted.begin() + S / 2, sorted.end());
    const double typical = S ? sorted[S / 2] : 0.0;
    H.typical_radius = typical;
    std::vector<uint32_t> idx, loose;
    for (size_t i = 0; i < S; ++i) {
        const bool finite = std::isfinite(geom[i * 4]) && std::isfinite(geom[i * 4 + 1]) && std::isfinite(geom[i * 4 + 2]) && std::isfinite(radii[i]);
        ((finite && radii[i] <= 16 * typical) ? idx : loose).push_back((uint32_t)i);
    }
    // blocks of 64, clusters of 8 inside them
    const size_t n = idx.size();
    split(geom, idx.data(), n, 64);
    for (size_t b = 0; b < n; b += 64) split(geom, idx.data() + b, std::min<size_t>(64, n - b), 8);
    const size_t blocks = (n + 63) / 64;
    H.blocks = (int)blocks;
    H.loose = (int)loose.size();
    H.mid.resize(blocks);
    H.leaf.resize(blocks * 8);
    H.ids.assign(blocks * 64, HIER_NO_SPHERE);
    H.block_bound.assign(blocks * 4, 0.0);
    std::vector<double>& block_bound = H.block_bound;
    for (size_t b = 0; b < blocks; ++b) {
        double cluster_bound[8 * 4];
        size_t n_clusters = 0;
        for (size_t c = 0; c < 8; ++c) {
            const size_t first = b * 64 + c * 8;
            const size_t cnt = first < n ? std::min<size_t>(8, n - first) : 0;
            double members[8 * 4];
            for (size_t m = 0; m < 8; ++m) {
                Group8& group = H.leaf[b * 8 + c];
                if (m < cnt) {
                    const uint32_t s = idx[first + m];
                    for (int k = 0; k < 4; ++k) members[m * 4 + k] = geom[s * 4 + k];
                    group_entry(group, (int)m, geom + (size_t)s * 4);
                    H.ids[(b * 8 + c) * 8 + m] = (uint16_t)s;
                } else {
                    group_blank(group, (int)m);
                }
            }
            if (cnt) {
                bound(members, cnt, cluster_bound + n_clusters * 4);
                group_entry(H.mid[b], (int)c, cluster_bound + n_clusters * 4);
                ++n_clusters;
            } else {
                group_blank(H.mid[b], (int)c);
            }
        }
        bound(cluster_bound, n_clusters, block_bound.data() + b * 4);
    }
    // top list: blocks, loose spheres, padding to whole groups
    const size_t entries = blocks + loose.size();
    const size_t groups = (entries + 7) / 8;
    H.top.resize(groups);
    H.top_ref.assign(groups * 8, HIER_NONE);
    for (size_t e = 0; e < groups * 8; ++e) {
        Group8& group = H.top[e / 8];
        if (e < blocks) {
            group_entry(group, (int)(e & 7), block_bound.data() + e * 4);
            H.top_ref[e] = (uint32_t)e;
        } else if (e < entries) {
            const uint32_t s = loose[e - blocks];
            group_entry(group, (int)(e & 7), geom + (size_t)s * 4);
            H.top_ref[e] = HIER_LOOSE | s;
        } else {
            group_blank(group, (int)(e & 7));
        }
    }
    return H;
}

}  // namespace rt
