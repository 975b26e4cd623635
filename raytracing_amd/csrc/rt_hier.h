// Bounding-sphere hierarchy over the spheres of a scene with many of them: which spheres can a
// ray's line touch at all?
//
// Three levels, every level screened with the float32 test of cull_test (rt_math.h): groups of
// 64 spheres ("blocks", one bounding sphere each), inside a block 8 clusters of 8 spheres (one
// bounding sphere each), inside a cluster the spheres themselves.  Spheres much larger than the
// rest ("loose", e.g. a ground sphere) stay out of the groups and sit in the top list directly.
// A bounding sphere contains its members, so "the line certainly misses the bounding sphere"
// (what a positive cull_test value certifies) means it misses every member; what survives all
// three levels are the lane's candidates, and only they take the float64 test of the reference,
// which alone decides - in ascending sphere order, as a loop over all spheres would.
//
// The top list is the same for all lanes and is read through the scalar unit; below it every lane
// follows its own survivors.  Clusters and candidates go to two short per-lane lists in LDS; a lane
// whose candidate list overflows reports that, and the caller screens all spheres for it (the
// plain loop over the whole table).
#pragma once
#include <math.h>

#include "rt_math.h"

namespace rt {

constexpr int HIER_CAP = 16;                 // entries per lane and list
constexpr uint32_t HIER_LOOSE = 0x80000000u; // top entry is a single sphere (index in the low bits)
constexpr uint32_t HIER_NONE = 0xFFFFFFFFu;  // top entry is padding
constexpr uint16_t HIER_NO_SPHERE = 0xFFFFu; // cluster slot is padding

// One lane's list in LDS: element k of lane `tid` lives at base[k * BLOCK + tid].
struct LaneList {
    uint16_t* p;
    __device__ __forceinline__ uint16_t& operator[](int k) const { return p[k * BLOCK]; }
};

// Eight entries of one level (the blocks of a stretch of the top list, the clusters of a block,
// the spheres of a cluster), component by component: one 128-byte line, read with four-word loads.
// re = radius + the entry's own error term (cull_entry's r and e), rounded up.
struct alignas(128) Group8 {
    float cx[8], cy[8], cz[8], re[8];
};
inline void group_entry(Group8& g, int k, const double* sphere) {   // host
    CullPair p;
    cull_entry(p, 0, sphere);
    g.cx[k] = p.cx[0]; g.cy[k] = p.cy[0]; g.cz[k] = p.cz[0];
    g.re[k] = nextafterf((float)((double)p.r[0] + (double)p.e[0]), INFINITY);
}
inline void group_blank(Group8& g, int k) {   // host: far away (nothing is assumed about the test's outcome, see the padding marks)
    g.cx[k] = g.cy[k] = g.cz[k] = 1.0e15f;
    g.re[k] = 0.0f;
}

// cull_test plus the part of the line that matters, for entries 2j and 2j + 1 of a group.  With b
// the place of an entry's centre on the line (measured from (ox, oy, oz) along the direction) and
// t its radius plus the error terms, every point of the entry lies in [b - t, b + t].  Per entry:
//   .miss > 0    the line certainly misses it;
//   .behind > 0  it lies before s_from, the ray's origin (rounded down by more than its error):
//                it cannot hold a root the callers accept (they want the far root positive);
//   .near        b - t, the first place where it can be met.
struct CullRange {
    v2f miss, behind, near;
};
__device__ __forceinline__ CullRange cull_range(v2f cx, v2f cy, v2f cz, v2f re, float ox, float oy, float oz,
                                                float dx, float dy, float dz, float e_ray, float s_from) {
    const v2f ocx = cx - ox, ocy = cy - oy, ocz = cz - oz;
    const v2f b = vfma(ocz, dz, vfma(ocy, dy, ocx * dx));
    const v2f px = vfma(-b, dx, ocx), py = vfma(-b, dy, ocy), pz = vfma(-b, dz, ocz);
    const v2f p2 = vfma(pz, pz, vfma(py, py, px * px));
    const v2f t = re + e_ray;
    CullRange r;
    r.miss = vfma(-t, t, p2);
    r.behind = s_from - (b + t);
    r.near = b - t;
    return r;
}
// A group read by each lane for itself (four-word vector loads).
struct Group8Regs {
    float4 v[8];   // cx[0..3], cx[4..7], cy.., cz.., re..
    __device__ __forceinline__ v2f pair(int comp, int j) const {   // entries 2j, 2j + 1 of component comp
        const float4 q = v[comp * 2 + (j >> 1)];
        return (j & 1) ? (v2f){q.z, q.w} : (v2f){q.x, q.y};
    }
};
__device__ __forceinline__ Group8Regs load_group(const Group8* table, uint32_t index) {
    const float4* g = reinterpret_cast<const float4*>(table + index);
    Group8Regs r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = g[i];
    return r;
}

// The part of a ray a call of hier_candidates looks at (places on the line as in cull_range).
struct HierWindow {
    float s_from;   // the ray's origin
    float s_prev;   // top entries that begin at or before s_prev were handled by an earlier call (NaN: none) ...
    float s_to;     // ... this call handles those that begin in (s_prev, s_to]
    float s_best;   // nothing that begins after s_best can matter (a nearer hit is known); +inf: none
};

// Candidates of this lane's ray into C, ascending; returns their number, or -1 if the list
// overflowed; `more` tells whether top entries that can matter begin beyond the window.  The ray:
// the line through (ox, oy, oz) (a point of it near the reference point of the tables) with
// direction (dx, dy, dz); e_ray as for cull_test.  Inactive lanes get 0.  Wave-uniform control flow.
//
// The survivors of the top list are kept as a bit mask per 64 entries; every lane then walks its
// own blocks (their surviving clusters go to the list B), then its own clusters (candidates to C).
__device__ __forceinline__ int hier_candidates(const Dev& D, LaneList B, LaneList C, bool active, float ox, float oy, float oz,
                                               float dx, float dy, float dz, float e_ray, const HierWindow& win, bool& more) {
    typedef const __attribute__((address_space(4))) Group8* ConstGroups;   // uniform reads through the scalar unit
    ConstGroups top = (ConstGroups)(uintptr_t)D.hier_top;
    int nC = 0;
    bool overflow = false;
    auto push = [&](uint32_t v) {
        if (nC < HIER_CAP) C[nC++] = (uint16_t)v; else overflow = true;
    };
    const float s_far = fminf(win.s_to, win.s_best);
    more = false;
    // an entry below the top list: met by the line, not behind the origin, not beyond a known hit
    // (one three-way maximum; a NaN operand is ignored by it, so an entry with non-finite data stays)
    auto keep = [&](const CullRange& r, int w) {
        const v2f beyond = r.near - win.s_best;
        return !(__builtin_fmaxf(__builtin_fmaxf(r.miss[w], r.behind[w]), beyond[w]) > 0);
    };
    for (int chunk = 0; chunk < D.hier_top_groups; chunk += 8) {
        // 1. up to 64 entries of the top list (blocks, loose spheres, padding)
        unsigned long long m = 0;
        const int chunk_groups = min(8, D.hier_top_groups - chunk);
        for (int g = 0; g < chunk_groups; ++g) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const v2f cx = {top[chunk + g].cx[2 * j], top[chunk + g].cx[2 * j + 1]};
                const v2f cy = {top[chunk + g].cy[2 * j], top[chunk + g].cy[2 * j + 1]};
                const v2f cz = {top[chunk + g].cz[2 * j], top[chunk + g].cz[2 * j + 1]};
                const v2f re = {top[chunk + g].re[2 * j], top[chunk + g].re[2 * j + 1]};
                const CullRange r = cull_range(cx, cy, cz, re, ox, oy, oz, dx, dy, dz, e_ray, win.s_from);
#pragma unroll
                for (int w = 0; w < 2; ++w)
                    if (!(r.miss[w] > 0) && !(r.behind[w] > 0)) {
                        if (!(r.near[w] <= win.s_prev) && !(r.near[w] > s_far)) m |= 1ull << (8 * g + 2 * j + w);
                        if (r.near[w] > win.s_to && !(r.near[w] > win.s_best)) more = true;
                    }
            }
        }
        if (!active) m = 0;
        // 2. every lane takes its own surviving entries one by one: the clusters of a block that
        //    survive go to the lane's cluster list, a loose sphere is a candidate already.  When
        //    some lane's cluster list is nearly full, the wave works the lists off (step 3).
        int nB = 0;
        auto clusters_to_spheres = [&]() {
            // 3. the spheres of the listed clusters (as many steps as the longest list of the wave)
            for (int i = 0; __ballot(i < nB) != 0; ++i) {
                if (i < nB) {
                    const uint32_t cl = B[i];
                    const Group8Regs g = load_group(D.hier_leaf, cl);
                    const uint4 ids = reinterpret_cast<const uint4*>(D.hier_ids)[cl];   // 8 sphere indices of 16 bits
                    const uint32_t id2[4] = {ids.x, ids.y, ids.z, ids.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const CullRange r = cull_range(g.pair(0, k), g.pair(1, k), g.pair(2, k), g.pair(3, k), ox, oy, oz, dx, dy, dz, e_ray, win.s_from);
                        const uint32_t s0 = id2[k] & 0xffffu, s1 = id2[k] >> 16;
                        if (keep(r, 0) && s0 != HIER_NO_SPHERE) push(s0);
                        if (keep(r, 1) && s1 != HIER_NO_SPHERE) push(s1);
                    }
                }
            }
            nB = 0;
        };
        while (__ballot(m != 0) != 0) {
            if (m != 0) {
                const int e = __builtin_ctzll(m);
                m &= m - 1;
                // the blocks come first in the top list, in order; only other entries need the look-up
                uint32_t ref = 8 * chunk + e;
                if ((int)ref >= D.hier_blocks) ref = D.hier_top_ref[ref];
                if (ref == HIER_NONE) {
                    // padding (its table entry lies far away, but nothing is assumed about the test's outcome)
                } else if (ref & HIER_LOOSE) {
                    push(ref & 0xffffu);
                } else {
                    const Group8Regs g = load_group(D.hier_mid, ref);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const CullRange r = cull_range(g.pair(0, k), g.pair(1, k), g.pair(2, k), g.pair(3, k), ox, oy, oz, dx, dy, dz, e_ray, win.s_from);
#pragma unroll
                        for (int w = 0; w < 2; ++w)
                            if (keep(r, w)) B[nB++] = (uint16_t)(ref * 8 + 2 * k + w);   // room for 8 is kept free
                    }
                }
            }
            if (__ballot(nB > HIER_CAP - 8) != 0) clusters_to_spheres();
        }
        clusters_to_spheres();
    }
    if (overflow) return -1;
    // 3. ascending sphere order (the list is short: insertion sort)
    for (int i = 1; i < nC; ++i) {
        const uint16_t v = C[i];
        int j = i;
        while (j > 0 && C[j - 1] > v) { C[j] = C[j - 1]; --j; }
        C[j] = v;
    }
    return nC;
}

}  // namespace rt
