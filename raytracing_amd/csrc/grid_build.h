// Host side of the uniform grid (rt_grid.h): choosing the cells and registering the spheres.
#pragma once
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "rt_grid.h"

namespace rt {

struct HostGrid {
    std::vector<uint32_t> start;    // [cells][2]: first item and one past the last item of each cell
    std::vector<uint16_t> items;    // sphere indices, an even number per cell (padded with the blank entry's index)
    std::vector<float> sph;         // [S + 1][4]: centre relative to `origin`, radius + error term; entry S is blank
    std::vector<uint16_t> loose;    // spheres kept out of the grid: always tested
    double origin[3] = {0, 0, 0}, hi[3] = {0, 0, 0};   // the box: [0, hi] relative to origin
    float h = 0, inv_h = 0;
    float margin = 0;               // what a sphere's radius is enlarged by when it is registered (plus its own rounding terms)
    int n[3] = {0, 0, 0};
    int cells() const { return n[0] * n[1] * n[2]; }
};

// geom: [S][4] centre and radius.  `density`: cells per registered sphere aimed at.  Spheres with a
// radius far above the typical one, or that would be registered in very many cells, stay loose.
// An empty `start` means "no grid" (no sphere left to register).
inline HostGrid build_grid(const double* geom, size_t S, double density) {
    HostGrid G;
    std::vector<double> radii(S);
    for (size_t i = 0; i < S; ++i) radii[i] = fabs(geom[i * 4 + 3]);
    std::vector<double> sorted = radii;
    if (S) std::nth_element(sorted.begin(), sorted.begin() + S / 2, sorted.end());
    const double typical = S ? sorted[S / 2] : 0.0;
    std::vector<uint32_t> in;
    for (size_t i = 0; i < S; ++i) {
        const bool finite = std::isfinite(geom[i * 4]) && std::isfinite(geom[i * 4 + 1]) && std::isfinite(geom[i * 4 + 2]) && std::isfinite(radii[i]);
        if (finite && radii[i] <= 16 * typical) in.push_back((uint32_t)i); else G.loose.push_back((uint16_t)i);
    }
    // blank entry: far away, never passes the screening
    G.sph.assign((S + 1) * 4, 0.0f);
    G.sph[S * 4 + 0] = 1.0e15f;
    if (in.empty()) return G;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i : in)
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], geom[i * 4 + k] - radii[i]);
            hi[k] = std::max(hi[k], geom[i * 4 + k] + radii[i]);
        }
    double ext[3], longest = 0;
    for (int k = 0; k < 3; ++k) longest = std::max(longest, hi[k] - lo[k]);
    longest = std::max(longest, 1e-30);
    for (int k = 0; k < 3; ++k) {
        lo[k] -= 1e-3 * longest;
        hi[k] += 1e-3 * longest;
        ext[k] = hi[k] - lo[k];
        G.origin[k] = lo[k];
    }
    // cell size: `density` cells per sphere, not below the typical radius, at most GRID_MAX_AXIS cells per axis
    double h = cbrt(ext[0] * ext[1] * ext[2] / std::max(1.0, density * (double)in.size()));
    h = std::max(h, typical);
    h = std::max(h, longest / GRID_MAX_AXIS * 1.0001);
    for (int k = 0; k < 3; ++k) {
        G.n[k] = std::max(1, std::min(GRID_MAX_AXIS, (int)ceil(ext[k] / h)));
        G.hi[k] = ext[k];
    }
    G.h = (float)h;
    h = (double)G.h;   // the device divides by this value
    G.inv_h = (float)(1.0 / h);
    G.margin = (float)(GRID_SLACK * h + GRID_DRIFT * longest);
    const size_t cells = (size_t)G.cells();
    // every sphere into the cells its enlarged ball reaches
    std::vector<std::vector<uint16_t>> lists(cells);
    auto cell_range = [&](double c, double R, int k, int& a, int& b) {
        a = std::max(0, std::min(G.n[k] - 1, (int)floor((c - R) / h)));
        b = std::max(0, std::min(G.n[k] - 1, (int)floor((c + R) / h)));
    };
    for (uint32_t i : in) {
        const double c[3] = {geom[i * 4] - G.origin[0], geom[i * 4 + 1] - G.origin[1], geom[i * 4 + 2] - G.origin[2]};
        const double l1 = fabs(c[0]) + fabs(c[1]) + fabs(c[2]);
        const double R = radii[i] * (1 + 4e-6) + G.margin + 1e-5 * l1;
        int a[3], b[3];
        for (int k = 0; k < 3; ++k) cell_range(c[k], R, k, a[k], b[k]);
        const size_t reach = (size_t)(b[0] - a[0] + 1) * (b[1] - a[1] + 1) * (b[2] - a[2] + 1);
        if (reach > 512) {
            G.loose.push_back((uint16_t)i);
        } else {
            for (int z = a[2]; z <= b[2]; ++z)
                for (int y = a[1]; y <= b[1]; ++y)
                    for (int x = a[0]; x <= b[0]; ++x) {
                        // distance from the centre to the cell's box
                        const int q[3] = {x, y, z};
                        double d2 = 0;
                        for (int k = 0; k < 3; ++k) {
                            const double blo = q[k] * h, bhi = (q[k] + 1) * h;
                            const double dk = c[k] < blo ? blo - c[k] : (c[k] > bhi ? c[k] - bhi : 0.0);
                            d2 += dk * dk;
                        }
                        if (d2 <= R * R) lists[((size_t)z * G.n[1] + y) * G.n[0] + x].push_back((uint16_t)i);
                    }
            // screening entry: as cull_entry, coordinates relative to the grid's origin
            G.sph[i * 4 + 0] = (float)c[0]; G.sph[i * 4 + 1] = (float)c[1]; G.sph[i * 4 + 2] = (float)c[2];
            G.sph[i * 4 + 3] = nextafterf((float)(radii[i] * 1.000002 + 1.5e-6 * l1 * 1.000001), INFINITY);
        }
    }
    std::sort(G.loose.begin(), G.loose.end());
    for (uint16_t i : G.loose) {   // not in any list; a blank entry all the same
        G.sph[(size_t)i * 4 + 0] = 1.0e15f; G.sph[(size_t)i * 4 + 1] = 0; G.sph[(size_t)i * 4 + 2] = 0; G.sph[(size_t)i * 4 + 3] = 0;
    }
    G.start.resize(cells * 2);
    for (size_t c = 0; c < cells; ++c) {
        G.start[2 * c] = (uint32_t)G.items.size();
        for (uint16_t i : lists[c]) G.items.push_back(i);        // ascending: spheres were visited in index order
        if (G.items.size() & 1) G.items.push_back((uint16_t)S);  // pad to a pair
        G.start[2 * c + 1] = (uint32_t)G.items.size();
    }
    if (G.items.empty()) G.items.push_back((uint16_t)S), G.items.push_back((uint16_t)S);
    return G;
}

}  // namespace rt
