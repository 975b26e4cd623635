// HIP kernels of the level-synchronous wavefront ray tracer (gfx950, wave64).
//
// One frame is:  k_init, then for every bounce level  k_trace -> k_scan -> k_link,
// then  k_surface  once, then from the deepest level upwards  k_combine.
//
//  * A level's rays live in the arena in tiles of 64 (one wavefront), grouped by tree node
//    (a node = all rays that followed the same branch sequence, the batch of one
//    recursive call in reference raytrace.py:19).  Inside a node rays stay in
//    ascending pixel order, as boolean-mask selection keeps them in the reference.
//  * k_trace derives each ray from its parent's hit record, finds the nearest
//    sphere and stores the hit record; it counts, per tile, the children each
//    branch will get, the hits, the hits inside each spotlight's beam and the
//    hits whose surface colour will be needed.
//  * k_scan turns the counts into node sizes and slots.
//  * k_link is the ordered compaction (wave ballots + prefix): every child slot of
//    the next level learns its parent, and every hit that needs lighting is
//    appended to one dense list together with its rank inside its node.
//  * k_surface evaluates the lights for the dense list (all levels in one launch,
//    full waves whatever the fraction of rays that matter).
//  * k_combine adds a ray's surface colour and its children's colours (bottom-up,
//    the reference's return path).
//
// Compiled with -ffp-contract=off: the geometry follows the reference's rounding
// order operation by operation; only explicit fma() calls fuse.
#include "rt_internal.h"

#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 4   // waves per SIMD requested for k_trace
#endif
#ifndef RT_TRACE_WAVES_MANY
#define RT_TRACE_WAVES_MANY 4   // the same for the many-sphere variant of k_trace (4096 spheres at 4K, k_trace per frame: 3: 70.9 ms, 4: 63.3 with 7 spilled registers, 5: 76.6 with 64, 6: 91.7 with 110; without the prefetch of the next tile's parent record 67.0 at 4 and 63.5 at 5)
#endif
#ifndef RT_COMBINE_WAVES
#define RT_COMBINE_WAVES 4   // waves per SIMD requested for k_combine (measured: 4: 3.25 ms per step, 5: 4.8, 6: 6.6)
#endif
#ifndef RT_LINK_WAVES
#define RT_LINK_WAVES 8      // ... and k_link (6: 2.18 ms, 8: 2.06)
#endif
#ifndef RT_SURFACE_WAVES
#define RT_SURFACE_WAVES 2   // waves per SIMD requested for the float64 lighting kernel (few spheres)
#endif
#ifndef RT_PACK_WAVES
#define RT_PACK_WAVES 3   // waves per SIMD requested for k_pack (4K example_4 end to end: 2: 27.1 ms, 3: 27.0, 4: 26.7 but with 80 bytes of spilled registers per lane and 5 GB more traffic, 5: 28.3, 6: 31.7)
#endif
#ifndef RT_SURFACE_WAVES_MANY
#define RT_SURFACE_WAVES_MANY 3   // waves per SIMD requested for the many-sphere variant of k_surface (measured with the grid and the light maps, 4096 spheres at 4K: 2: 24.7 ms, 3: 24.6, 4: 27.0, 6: 41.8, 8: 54.6)
#endif
#ifndef RT_HIER_GROWTH
#define RT_HIER_GROWTH 4    // each stretch of a ray searched in the sphere hierarchy is this many times the one before
#endif
#ifndef RT_GRID_CHECK
#define RT_GRID_CHECK 4     // the candidates of a grid walk take the float64 test every this many steps (a power of two)
#endif
#ifndef RT_MIXED_WAVES
#define RT_MIXED_WAVES 2   // waves per SIMD requested for the mixed-precision surface kernel (4K example_4, with batches of 3 / 5 sample pairs: 1: 4.1 / 4.0 ms, 2: 2.8 / 3.5, 3: 3.3 / 4.3 with 156 / 279 spilled registers, 4: 4.8; batches of 2 at two waves: 2.9)
#endif

#include "rt_math.h"
#include "rt_hier.h"
#include "rt_grid.h"
#include "lightmap_build.h"
#include "rt_geometry.hip"
#include "rt_samples.hip"
#include "rt_lighting.hip"
#include "rt_image.hip"
#include "rt_debug.hip"
