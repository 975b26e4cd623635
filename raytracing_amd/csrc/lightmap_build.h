// Occluders of a point light by direction (host side).
//
// Every shadow ray towards a point light lies on a line through the light's position, so the
// spheres such a line can touch are the ones whose cone of directions - as seen from the light -
// contains the line's direction (on either side of the light: the reference's shadow rays do not
// stop at the light, objects.py:274-285).  The directions are binned like the faces of a cube
// (face = largest component, two coordinates u, w in [-1, 1], LMAP_N x LMAP_N bins per face) and
// every sphere is listed in the bins its cone reaches; a shadow ray then screens the lists of two
// bins instead of walking the grid.  Spheres that contain the light, reach very many bins or are
// not in the grid are listed once as "always".
#pragma once
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace rt {

constexpr int LMAP_N = 128;                 // bins per face edge
constexpr int LMAP_MAX_REACH = 1024;        // a sphere in more bins than this is tested always
constexpr int LMAP_MAX_ALWAYS = 6;          // lights with more "always" spheres keep the grid walk
constexpr double LMAP_MARGIN = 1e-4;        // slack of the bin ranges, in units of u (a bin is 2 / LMAP_N wide)

struct HostLightMap {
    std::vector<uint32_t> start;    // [6 * N * N][2] first item and one past the last item of each bin
    std::vector<uint16_t> items;    // sphere indices, ascending inside a bin
    std::vector<uint16_t> always;   // ascending
    bool usable = false;
};

// Bin range [lo, hi] (inclusive, clamped) that the disk of radius r around (ca, cb) - ca measured
// along the face's outward axis - reaches in the coordinate cb / ca, or false if none.
inline bool lmap_axis_range(double ca, double cb, double r, int& lo, int& hi) {
    const double quarter = 0.7853981633974483 + LMAP_MARGIN;
    double a0 = -quarter, a1 = quarter;
    const double dist = sqrt(ca * ca + cb * cb);
    if (dist > r) {
        const double phi = atan2(cb, ca), delta = asin(std::min(1.0, r / dist)) + LMAP_MARGIN;
        if (phi + delta < -quarter || phi - delta > quarter) return false;
        a0 = std::max(a0, phi - delta);
        a1 = std::min(a1, phi + delta);
    }
    const double u0 = std::max(-1.0, tan(a0) - LMAP_MARGIN), u1 = std::min(1.0, tan(a1) + LMAP_MARGIN);
    lo = std::max(0, std::min(LMAP_N - 1, (int)floor((u0 * 0.5 + 0.5) * LMAP_N)));
    hi = std::max(0, std::min(LMAP_N - 1, (int)floor((u1 * 0.5 + 0.5) * LMAP_N)));
    return true;
}

// geom: [S][4] centre and radius; in_grid[i]: sphere i has a float32 screening entry in the grid's table
// (the others are tested always).
inline HostLightMap build_light_map(const double* geom, size_t S, const double* light, const std::vector<char>& in_grid) {
    HostLightMap M;
    const size_t bins = (size_t)6 * LMAP_N * LMAP_N;
    std::vector<std::vector<uint16_t>> lists(bins);
    for (size_t i = 0; i < S; ++i) {
        const double v[3] = {geom[i * 4] - light[0], geom[i * 4 + 1] - light[1], geom[i * 4 + 2] - light[2]};
        const double r = fabs(geom[i * 4 + 3]) * (1 + 1e-6) + 1e-9 * (fabs(v[0]) + fabs(v[1]) + fabs(v[2]));
        const double dist = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        bool always = !in_grid[i] || !(dist > r) || !std::isfinite(dist) || !std::isfinite(r);
        struct Rect { int face, u0, u1, w0, w1; };
        Rect rects[6];
        int n_rects = 0;
        size_t reach = 0;
        for (int face = 0; face < 6 && !always; ++face) {
            // face 2a: +axis a, face 2a + 1: -axis a; u along axis (a + 1) % 3, w along axis (a + 2) % 3
            const int a = face >> 1;
            const double s = (face & 1) ? -1.0 : 1.0;
            const double ca = s * v[a], cb = v[(a + 1) % 3], cc = v[(a + 2) % 3];
            Rect q{face, 0, 0, 0, 0};
            if (!lmap_axis_range(ca, cb, r, q.u0, q.u1) || !lmap_axis_range(ca, cc, r, q.w0, q.w1)) continue;
            rects[n_rects++] = q;
            reach += (size_t)(q.u1 - q.u0 + 1) * (q.w1 - q.w0 + 1);
        }
        if (reach > (size_t)LMAP_MAX_REACH) always = true;
        if (always) {
            M.always.push_back((uint16_t)i);
            continue;
        }
        for (int k = 0; k < n_rects; ++k)
            for (int w = rects[k].w0; w <= rects[k].w1; ++w)
                for (int u = rects[k].u0; u <= rects[k].u1; ++u)
                    lists[((size_t)rects[k].face * LMAP_N + w) * LMAP_N + u].push_back((uint16_t)i);
    }
    M.usable = M.always.size() <= (size_t)LMAP_MAX_ALWAYS;
    if (!M.usable) return M;
    M.start.resize(bins * 2);
    for (size_t b = 0; b < bins; ++b) {
        M.start[2 * b] = (uint32_t)M.items.size();
        M.items.insert(M.items.end(), lists[b].begin(), lists[b].end());   // ascending: spheres were visited in index order
        M.start[2 * b + 1] = (uint32_t)M.items.size();
    }
    if (M.items.empty()) M.items.push_back(0);
    return M;
}

}  // namespace rt
