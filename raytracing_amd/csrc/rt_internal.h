// Internal definitions shared by the device kernels and the host side of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/raytracing_amd.h"

namespace rt {

constexpr int TILE = 64;              // rays per tile = one wavefront
constexpr int BLOCK = 256;            // threads per workgroup
constexpr int WPB = BLOCK / TILE;     // waves per workgroup
#ifndef RT_NB
#define RT_NB 2048
#endif
constexpr int NB = RT_NB;              // workgroups per level launch (256 CUs x 8)
constexpr int NW = NB * WPB;          // waves per level launch; each owns a contiguous run of tiles
constexpr int NCNT = RT_NUM_COUNTERS; // per-tile counters: 3 child kinds, hits, in-beam per light, shade items
constexpr int MAX_LEVELS = 32;
constexpr int MAX_LEVEL_NODES = 1024; // widest level of the static ray tree we accept
constexpr int MAX_NODES = 65536;
constexpr int SPH_CHUNK_SMALL = 64;   // up to this many spheres the kernels keep all sphere constants in LDS
constexpr int SHADE_LDS_SPHERES = 128; // sphere and material tables of the lighting kernels kept in LDS up to this many

// counters
enum { C_T = 0, C_I = 1, C_R = 2, C_HIT = 3, C_BEAM0 = 4, C_SHADE = 4 + RT_MAX_LIGHTS,
       C_REC = 5 + RT_MAX_LIGHTS,    // hits whose record (hit point, direction) is stored
       C_LIVE = 6 + RT_MAX_LIGHTS,   // hits with non-zero weight in the pixel
       C_ANYLIVE = 7 + RT_MAX_LIGHTS }; // rays (hits and misses) with non-zero weight in the pixel

// bits of the per-ray info word
constexpr uint32_t I_SID_MASK = 0xFFFFu;   // sphere hit, I_NO_HIT for a miss
constexpr uint32_t I_NO_HIT = 0xFFFFu;
constexpr uint32_t I_LIVE = 1u << 16;      // the ray's colour reaches the pixel with non-zero weight
constexpr uint32_t I_INSIDE = 1u << 17;
constexpr uint32_t I_CHILD_T = 1u << 18;  // spawned a transmitted ray
constexpr uint32_t I_CHILD_I = 1u << 19;  // spawned a total-internal-reflection ray
constexpr uint32_t I_CHILD_R = 1u << 20;  // spawned a mirror-reflection ray
constexpr uint32_t I_SHADE = 1u << 21;    // the hit's surface colour reaches the pixel with non-zero weight
constexpr uint32_t I_LIVE_TI = 1u << 22;  // the ray's refraction slot reaches the pixel with non-zero weight
constexpr uint32_t I_LIVE_R = 1u << 23;   // ... and so does its reflection slot
constexpr int I_BEAM_SHIFT = 24;          // bit 24 + l: hit lies inside spotlight l's beam

// Float32 constants of a disk or spot light for the float32 sample loop (k_surface_mixed).
struct LightF32 {
    float xh[3], yh[3];   // the disk's frame times its radius
    float ps;             // intensity / n_points
    float extent;         // no light point is farther than this from the disk's centre
    float dcol[3], scol[3];
    float ps_grey;        // ps times the light's grey level (lights whose colour has three equal channels)
    float pad;
};

// Scene constants kept in device memory and read through scalar loads.
struct SceneConst {
    RtLight lights[RT_MAX_LIGHTS];
    LightF32 lightf[RT_MAX_LIGHTS];
    double ambient[3];
    double cam[3];
    double scr_pos[3], scr_x[3], scr_y[3];
    double scr_w, scr_h, pix_x, pix_y;
    double eps, air;
    double bg[3];
    int32_t small_num, width, height, n_lights;
};

// Everything a kernel needs, passed by value.
struct Group8;

struct Dev {
    // scene
    const double* geom;  // [S][4] centre, radius
    const double* mat;   // [S][18]
    const double* fres;  // [S][4] air/ior, ior/air, ((air - ior) / (air + ior))^2, unused
    const SceneConst* sc;
    int32_t S;
    int32_t n_lights;
    double cull_ref[3];  // a point in the middle of the (moderately sized) spheres, see the float32 screening
    const float* cull;   // [ceil(S / 8) * 4][10] float32 screening table, two spheres per entry (CullPair)
    // bounding-sphere hierarchy over the spheres (rt_hier.h), eight entries per Group8; hier_top_groups == 0: none
    const Group8* hier_top;        // [hier_top_groups]: blocks of 64 spheres, then loose spheres, then padding
    const uint32_t* hier_top_ref;  // [8 * hier_top_groups] block index, HIER_LOOSE | sphere, or HIER_NONE
    const Group8* hier_mid;        // [blocks]: the 8 clusters of each block
    const Group8* hier_leaf;       // [blocks * 8]: the 8 spheres of each cluster
    const uint16_t* hier_ids;      // [blocks * 8 * 8] sphere index of each cluster slot, or HIER_NO_SPHERE
    int32_t hier_top_groups;
    int32_t hier_blocks;           // the first hier_blocks entries of the top list are the blocks 0, 1, 2, ...
    float hier_window;             // length of the first stretch of a ray that k_trace searches (then x4, x16, ...)
    float hier_extent;             // from this length on a stretch is the whole rest of the ray
    // uniform grid over the spheres (rt_grid.h), used instead of the hierarchy when grid_cells != 0
    const uint32_t* grid_start;    // [grid_cells][2] first item and one past the last item of each cell
    const uint16_t* grid_items;    // sphere indices, an even number per cell (padding: index S, the blank entry)
    const float4* grid_sph;        // [S + 1] centre relative to grid_origin, radius + error term; entry S is blank
    const uint16_t* grid_loose;    // spheres that are not in the grid: always tested
    int32_t grid_n_loose;
    int32_t grid_cells;
    int32_t grid_n[3];             // cells along x, y, z
    float grid_h, grid_inv_h;      // cell size
    float grid_margin;             // what the spheres' radii were enlarged by at registration (rt_grid.h)
    double grid_origin[3];         // the grid's box is [0, grid_hi] relative to this point
    double grid_hi[3];
    // occluders of the point lights by direction (lightmap_build.h); bit l of lmap_mask: light l has a map
    const uint32_t* lmap_start;    // per light [6 * LMAP_N * LMAP_N][2] item range of each bin, at lmap_start_off[l]
    const uint16_t* lmap_items;    // sphere indices, ascending inside a bin
    const uint16_t* lmap_always;   // per light the spheres tested for every ray, at lmap_always_off[l]
    int32_t lmap_start_off[RT_MAX_LIGHTS], lmap_always_off[RT_MAX_LIGHTS], lmap_n_always[RT_MAX_LIGHTS];
    uint32_t lmap_mask;
    // static tree
    const int32_t* node_a;
    const int32_t* node_b;
    const int32_t* node_branch;
    const int32_t* node_child;   // [K][3]
    const int32_t* level_start;  // [n_levels + 1]
    int32_t K;
    int32_t n_levels;
    // per-frame node tables
    int32_t* node_n;        // local batch size
    int32_t* node_nglobal;  // whole-image batch sizes (== node_n on one GPU); use whole_image_batch()
    const int32_t* node_nguess;  // null, or the sizes the small-batch rule goes by instead (same layout, not written)
    int32_t batch_slot;     // this frame's place and the number of frames when several frames share a
    int32_t batch_slots;    // caller's buffer level by level (0 and 1: the buffer holds this frame alone)
    int32_t* node_hits;
    int32_t* node_beam;     // [K][RT_MAX_LIGHTS]
    int32_t* node_tile0;    // first tile of the node, relative to its level
    int32_t* node_ntiles;
    uint32_t* node_start;   // arena index of the node's first ray
    uint32_t* node_pre;     // [K][NCNT] level-wide running counter value at the node's first tile
    int32_t* lvl_ntiles;    // [n_levels]
    int32_t* lvl_tilebase;  // [n_levels] first tile of the level in the arena
    uint32_t* tile_run;     // [tiles][NCNT] running counter value inside the producing wave
    int nb;                 // workgroups of this frame's level launches (a multiple of 256, at most NB)
    uint32_t* blk_tot;      // [n_levels][NCNT][NB] per-workgroup totals, turned into prefixes by k_scan
    uint32_t* wave_off;     // [n_levels][NCNT][NW] counter value at a wave's first tile relative to its workgroup
    int32_t* flags;         // [0] overflow, [1] entries queued for float64 re-evaluation
    unsigned long long* work;  // [0] hits shaded, [1] light samples evaluated, [2] samples queued for float64,
                               // [3] hit records stored, [4] live hits, [5] certified float32 shadow decisions
                               // that differ from float64 (test mode); cleared by k_init
    // ray arena (structure of arrays, one slot per ray)
    double *px, *py, *pz;   // hit point
    double *dx, *dy, *dz;   // ray direction
    double *A;              // reflected fraction
    double *dist;           // path length inside the entered sphere (0: no attenuation)
    uint32_t* info;
    uint32_t* link;         // arena index of the parent ray
    int32_t col32;          // the arrays A, dist, cr..cb, sr..sb hold float32 values (mixed precision) instead of
                            // float64: smooth quantities no hit / miss decision depends on (load_smooth / store_smooth)
    double *cr, *cg, *cb;   // ray colour (after k_combine)
    double *sr, *sg, *sb;   // surface colour of shading-list entry e (after the lighting kernels): dense, so that
                            // the lighting kernels store and k_combine reads whole lines
    uint32_t cap_tiles;
    // dense list of the hits that need lighting, all levels one after the other
    int32_t* lvl_shade_base;   // [n_levels + 1] first list entry of each level; the last one is the total
    uint32_t* list_ray;        // arena index of the hit
    uint16_t* list_node;       // its tree node
    uint32_t* list_j[1 + RT_MAX_LIGHTS];  // [0] rank among the node's hits (disk lights), [1 + l] rank among
                                          // the node's hits inside spotlight l's beam; NULL when unused
    uint32_t spot_mask;        // bit l: light l is a spotlight
    uint32_t* redo_list;       // shading-list entries whose float32 lighting could not be certified
    // level-0 pixels
    const double* X;
    const double* Y;        // NULL: whole rows starting at row0
    int32_t row0;
    int32_t npix;
    // disk-light samples, packed per shading-list entry: for light l, sample s of entry e has its angle
    // uniform at packed[pack_base[l] + s * pack_E + e] and its radius uniform pack_plane doubles further on
    // (two planes, so that both the scatter that fills them and the readers touch whole cache lines)
    double* packed;
    int64_t pack_plane;
    // the same samples for the float32 pass, or null: the points of the unit disk the pairs select, as
    // floats, tile by tile of 64 shading-list entries; light l, tile t, samples 2p and 2p + 1 of entry
    // 64 t + i are the float4 (x0, x1, y0, y1) at index pack32_base[l] + (t * ceil(N / 2) + p) * 64 + i
    float* packed32;
    // *raw_values_at not null: `packed` has not been filled; value i of the stream is made of the generator's
    // raw words raw_values[2 i] and raw_values[2 i + 1] (GeneratedStream), positions as sample_column<true>
    // gives them.  Read from device memory when the kernel runs, not passed by value: a launch recorded in
    // a graph must follow the samples when they are moved into `packed` later (materialize_packed).
    const uint32_t* const* raw_values_at;
    int64_t pack32_base[RT_MAX_LIGHTS];
    int64_t pack_base[RT_MAX_LIGHTS];
    int64_t pack_E;             // entries of the shading list when the samples were packed
    const int64_t* sample_off;  // [K][RT_MAX_LIGHTS] offset of each (node, light) block in the caller's stream
    // optional, for samples generated on the device when this context renders a part of the
    // image: where the block sits in the whole image's stream ([3][K][RT_MAX_LIGHTS]: offset,
    // columns of the whole image, first column of this context), or null
    const int64_t* sample_global;
    // output
    double* image;  // [npix][3]
};

}  // namespace rt
