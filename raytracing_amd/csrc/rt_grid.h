// Uniform grid over the spheres of a scene with many of them: which spheres can a ray touch?
//
// The moderately sized spheres are registered in every cell of a regular grid their (slightly
// enlarged) ball reaches; spheres much larger than the rest ("loose", e.g. a ground sphere) stay
// out of the grid and are always tested.  A ray walks the cells its line crosses, front to back
// (a three-dimensional DDA in float32), and screens the spheres of each cell with the float32
// test of cull_test (rt_math.h): "the line certainly misses the sphere".  What survives are the
// lane's candidates, and only they take the float64 test of the reference, which alone decides.
// A nearest-hit search ends at the first cell that begins beyond a hit it already knows.
//
// Why no sphere that the exact ray touches is lost.  All float32 coordinates are relative to the
// grid's own origin and the walk starts where the ray enters the grid's box (computed in float64),
// so the float32 line stays within a few 1e-7 of the box's extent of the exact one inside the box.
// The place of the next cell boundary along an axis is kept as a running sum (one float32 addition
// per step along that axis): after at most 3 * GRID_MAX_AXIS steps it is off by less than
// 768 * 2^-24 of the ray's length inside the box, i.e. below 8e-5 of the box's extent.  A walk whose
// boundaries are off by that much visits cells that cover the exact ray up to that distance (a
// corner may be passed on the other side).  Every sphere is therefore registered with its radius
// enlarged by the grid's margin - GRID_SLACK cell sizes plus GRID_DRIFT of the box's extent plus
// its own rounding terms - so a point of the exact ray inside the sphere lies in, or within the
// margin of, a visited cell that lists the sphere; the same margin is allowed for when a walk ends
// at a known hit.  The screening itself uses the error terms derived above cull_test.
#pragma once
#include <math.h>

#include "rt_math.h"
#include "rt_hier.h"

namespace rt {

constexpr double GRID_SLACK = 1e-3;          // registration slack, in cell sizes
constexpr double GRID_DRIFT = 2e-4;          // ... plus this fraction of the box's longest side (drift of the walk's running sums)
constexpr int GRID_MAX_AXIS = 256;           // cells along one axis at most
constexpr int GRID_LIST = 16;                // candidates a lane collects before the wave tests them
static_assert(GRID_LIST == HIER_CAP, "the walk uses the per-lane lists of the hierarchy");
constexpr int GRID_MAX_LOOSE = 8;            // scenes with more spheres outside the grid keep the hierarchy

// One lane's walk.  Places along the ray are measured from the point where the ray enters the
// grid's box (`enter`, in the ray's own parameter), in float32.
struct GridWalk {
    float px, py, pz;        // entry point relative to the grid's origin
    float dx, dy, dz;
    float ix_, iy_, iz_;     // 1 / direction (inf for a zero component)
    float e_ray;             // this ray's part of the screening's error bound
    float t_out;             // where the ray leaves the box (rounded up)
    float tmx, tmy, tmz;     // where the ray crosses the next cell boundary along each axis
    float dtx, dty, dtz;     // from one boundary to the next along each axis (inf for a zero component)
    int sx, sy, sz;          // direction of the steps
    int cx, cy, cz;          // current cell
    uint32_t cur, end;       // items of the current cell still to screen
    double enter;            // ray parameter of the entry point
    bool done;
};

// Starts the walk of the ray o + t d (float64, unit d); returns a finished walk if the ray misses the box.
__device__ __forceinline__ GridWalk grid_begin(const Dev& D, bool active, V3 o, V3 d) {
    GridWalk w;
    w.done = !active;
    w.cur = w.end = 0;
    w.cx = w.cy = w.cz = 0;
    w.enter = 0;
    w.t_out = 0;
    w.px = w.py = w.pz = 0; w.dx = 1; w.dy = w.dz = 0; w.ix_ = w.iy_ = w.iz_ = 0; w.e_ray = 0; w.tmx = w.tmy = w.tmz = 0;
    w.dtx = w.dty = w.dtz = 0; w.sx = w.sy = w.sz = 1;
    if (!active) return w;
    // the box, enlarged by the slack, against the ray in float64
    const double slack = GRID_SLACK * (double)D.grid_h;
    const double ox = o.x - D.grid_origin[0], oy = o.y - D.grid_origin[1], oz = o.z - D.grid_origin[2];
    double t_in = 0, t_out = __builtin_inf();
    const double oo[3] = {ox, oy, oz}, dd[3] = {d.x, d.y, d.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double lo = -slack, hi = D.grid_hi[k] + slack;
        if (dd[k] == 0) {
            if (oo[k] < lo || oo[k] > hi) t_out = -1;
        } else {
            const double a = (lo - oo[k]) / dd[k], b = (hi - oo[k]) / dd[k];
            t_in = fmax(t_in, fmin(a, b));
            t_out = fmin(t_out, fmax(a, b));
        }
    }
    if (!(t_out >= t_in)) { w.done = true; return w; }
    w.enter = t_in;
    const double ex = ox + t_in * d.x, ey = oy + t_in * d.y, ez = oz + t_in * d.z;
    w.px = (float)ex; w.py = (float)ey; w.pz = (float)ez;
    w.dx = (float)d.x; w.dy = (float)d.y; w.dz = (float)d.z;
    w.ix_ = 1.0f / w.dx; w.iy_ = 1.0f / w.dy; w.iz_ = 1.0f / w.dz;
    w.e_ray = 1.5e-6f * (float)((fabs(ex) + fabs(ey) + fabs(ez)) * 1.000001);
    w.t_out = (float)((t_out - t_in) * 1.000001) + 1e-30f;
    const float ih = D.grid_inv_h;
    w.cx = min(max((int)floorf(w.px * ih), 0), D.grid_n[0] - 1);
    w.cy = min(max((int)floorf(w.py * ih), 0), D.grid_n[1] - 1);
    w.cz = min(max((int)floorf(w.pz * ih), 0), D.grid_n[2] - 1);
    const float h = D.grid_h;
    w.tmx = w.dx == 0 ? __builtin_inff() : ((float)(w.cx + (w.dx > 0 ? 1 : 0)) * h - w.px) * w.ix_;
    w.tmy = w.dy == 0 ? __builtin_inff() : ((float)(w.cy + (w.dy > 0 ? 1 : 0)) * h - w.py) * w.iy_;
    w.tmz = w.dz == 0 ? __builtin_inff() : ((float)(w.cz + (w.dz > 0 ? 1 : 0)) * h - w.pz) * w.iz_;
    w.dtx = fabsf(h * w.ix_); w.dty = fabsf(h * w.iy_); w.dtz = fabsf(h * w.iz_);
    w.sx = w.dx > 0 ? 1 : -1; w.sy = w.dy > 0 ? 1 : -1; w.sz = w.dz > 0 ? 1 : -1;
    const uint32_t cell = ((uint32_t)w.cz * D.grid_n[1] + w.cy) * D.grid_n[0] + w.cx;
    const uint2 se = *reinterpret_cast<const uint2*>(D.grid_start + 2 * (size_t)cell);
    w.cur = se.x; w.end = se.y;
    return w;
}

// One step of the walk: the next two spheres of the current cell (if any are left) are screened,
// survivors handed to `push`; and when the cell's list is then used up the walk moves on to the next
// cell, so that a lane passes an empty or nearly empty cell per step (the wave executes both parts
// anyway as soon as its lanes differ).  `limit`: the walk ends at the first cell that begins beyond
// it (the box's far side, or a known hit; same measure as t_out).
template <typename Push>
__device__ __forceinline__ void grid_step(const Dev& D, GridWalk& w, float limit, Push push) {
    if (w.done) return;
    const bool has = w.cur < w.end;
    uint32_t two = 0;
    if (has) {
        two = *reinterpret_cast<const uint32_t*>(D.grid_items + w.cur);   // lists are padded to pairs
        w.cur += 2;
    }
    if (w.cur >= w.end) {
        // next cell: across the nearest boundary (its list is requested before the screening below);
        // without branches: one of the three axes takes the step
        const float t_next = fminf(w.tmx, fminf(w.tmy, w.tmz));
        const bool bx = w.tmx <= w.tmy && w.tmx <= w.tmz, by = !bx && w.tmy <= w.tmz, bz = !bx && !by;
        w.cx += bx ? w.sx : 0; w.cy += by ? w.sy : 0; w.cz += bz ? w.sz : 0;
        w.tmx += bx ? w.dtx : 0.0f; w.tmy += by ? w.dty : 0.0f; w.tmz += bz ? w.dtz : 0.0f;
        const bool out = !(t_next <= limit) || (unsigned)w.cx >= (unsigned)D.grid_n[0] || (unsigned)w.cy >= (unsigned)D.grid_n[1] ||
                         (unsigned)w.cz >= (unsigned)D.grid_n[2];
        if (out) {
            w.done = true;
        } else {
            const uint32_t cell = ((uint32_t)w.cz * D.grid_n[1] + w.cy) * D.grid_n[0] + w.cx;
            const uint2 se = *reinterpret_cast<const uint2*>(D.grid_start + 2 * (size_t)cell);
            w.cur = se.x; w.end = se.y;
        }
    }
    if (has) {
        const uint32_t i0 = two & 0xffffu, i1 = two >> 16;
        const float4 s0 = D.grid_sph[i0], s1 = D.grid_sph[i1];
        const v2f ocx = (v2f){s0.x, s1.x} - w.px, ocy = (v2f){s0.y, s1.y} - w.py, ocz = (v2f){s0.z, s1.z} - w.pz;
        const v2f b = vfma(ocz, w.dz, vfma(ocy, w.dy, ocx * w.dx));
        const v2f qx = vfma(-b, w.dx, ocx), qy = vfma(-b, w.dy, ocy), qz = vfma(-b, w.dz, ocz);
        const v2f p2 = vfma(qz, qz, vfma(qy, qy, qx * qx));
        const v2f t = (v2f){s0.w, s1.w} + w.e_ray;
        const v2f miss = vfma(-t, t, p2);          // > 0: the line certainly misses the sphere
        const v2f behind = -(b + t);               // > 0: it ends before the walk's start (callers want a positive far root)
        const v2f beyond = (b - t) - limit;        // > 0: it begins beyond the limit
        if (!(__builtin_fmaxf(__builtin_fmaxf(miss.x, behind.x), beyond.x) > 0)) push(i0);
        if (!(__builtin_fmaxf(__builtin_fmaxf(miss.y, behind.y), beyond.y) > 0)) push(i1);
    }
}

}  // namespace rt
