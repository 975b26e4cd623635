/*
 * raytracing_amd.h - C ABI of the MI355X (gfx950) sphere ray tracer.
 *
 * The reference renderer (JoarGunnarsson/Raytracing) is pure Python and has no
 * FFI of its own.  Its narrowest stable seams are
 *     raytrace(scene) -> (HEIGHT, WIDTH, 3) float64      reference raytrace.py:273-279
 *     get_pixel_color(X, Y, scene) -> (n, 3) float64     reference raytrace.py:6-16
 * and everything executed under them.  This library replaces that execution:
 * the caller flattens a scene into the plain structs below (what the
 * reference reads from its Sphere / Material / light / Camera / Screen
 * objects), and the entry points run the frame as a level-synchronous
 * wavefront of HIP kernels.  All pointers are plain host pointers unless a
 * name says "device"; all functions return 0 on success and a negative
 * RT_ERR_* code otherwise, with a message available from rt_last_error().
 *
 * One RtContext drives one GPU; calls on a context must be serialised by the
 * caller; different contexts are independent.
 */
#ifndef RAYTRACING_AMD_H
#define RAYTRACING_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_MAX_LIGHTS 8
#define RT_MAX_SPHERES 65535
#define RT_MAT_WIDTH 18
#define RT_NUM_COUNTERS (8 + RT_MAX_LIGHTS)

#define RT_OK 0
#define RT_ERR_INVALID (-1)   /* bad argument / unsupported scene            */
#define RT_ERR_HIP (-2)       /* a HIP runtime call failed                   */
#define RT_ERR_OVERFLOW (-3)  /* the ray arena was too small for this frame  */
#define RT_ERR_STATE (-4)     /* calls made in the wrong order               */

enum { RT_LIGHT_POINT = 0, RT_LIGHT_DISK = 1, RT_LIGHT_SPOT = 2 };
enum { RT_EASE_NONE = 0, RT_EASE_LINEAR = 1, RT_EASE_QUADRATIC = 2, RT_EASE_CUBIC = 3,
       RT_EASE_EXPONENTIAL = 4 };

/* One light source: the attributes the reference's PointSource / DiskSource /
 * DirectionalDiskSource read in compute_light_intensity (reference
 * objects.py:94-235).  x_hat / y_hat are the disk's in-plane frame as the
 * reference builds it (objects.py:124-130, not normalised); tan_angle and
 * tan_outer are tan(angle) and tan(angle + fall_off_angle) (objects.py:209-210). */
typedef struct RtLight {
    int32_t kind;      /* RT_LIGHT_*                                   */
    int32_t n_points;  /* disk samples per hit (1 for a point light)   */
    int32_t easing;    /* RT_EASE_*, spotlights only                   */
    int32_t reserved;
    double position[3];
    double intensity;
    double diffuse_color[3];
    double specular_color[3];
    double radius;
    double normal[3];
    double x_hat[3];
    double y_hat[3];
    double tan_angle;
    double tan_outer;
} RtLight;

/* A flattened scene.  sphere_geom is [n_spheres][4] = centre x, y, z, radius
 * (reference objects.py:54-58).  sphere_mat is [n_spheres][RT_MAT_WIDTH]:
 *   0-2 ambient colour, 3-5 diffuse colour * diffuse coefficient,
 *   6-8 specular colour * specular coefficient, 9 shininess,
 *   10 reflection coefficient, 11 transparency coefficient, 12 refractive index,
 *   13 smoothness, 14 attenuation coefficient, 15-17 absorption colour
 * (reference materials.py:4-29, raytrace.py:243-252).  ambient is
 * AmbientLight.intensity * color (reference raytrace.py:137-138).  The screen
 * fields are those read by Screen.index_to_position (reference objects.py:45-51). */
typedef struct RtScene {
    int32_t n_spheres;
    int32_t n_lights;
    const double *sphere_geom;
    const double *sphere_mat;
    const RtLight *lights;
    double ambient[3];
    double camera_position[3];
    double screen_position[3];
    double screen_x[3];
    double screen_y[3];
    double screen_width;
    double screen_height;
    double pixels_x;
    double pixels_y;
} RtScene;

/* Render settings: the reference's module constants (reference constants.py:2-19). */
typedef struct RtParams {
    int32_t width;
    int32_t height;
    int32_t max_reflection_depth;
    int32_t max_refraction_depth;
    int32_t small_num;
    int32_t reserved;
    double epsilon;
    double air_refractive_index;
    double background[3];
} RtParams;

/* Frame counters, valid after rt_trace_geometry (and rt_shade for shadow_rays). */
typedef struct RtCounters {
    int64_t pixels;          /* primary rays of this context's pixel set               */
    int64_t rays;            /* rays handed to the nearest-hit search (local batches in
                                nodes whose global batch is >= small_num)              */
    int64_t hits;            /* of those, rays that hit a sphere                       */
    int64_t shaded_hits;     /* hits whose surface colour was evaluated by rt_shade (hits
                                with zero weight in the pixel are skipped)             */
    int64_t shaded_samples;  /* light samples evaluated for them = shadow rays cast    */
    int64_t fallback_samples;/* light samples of the hits the float32 evaluation could not
                                certify and re-evaluated in float64 (mixed precision)  */
    int64_t stored_records;  /* hits whose record (hit point, direction) was stored: hits
                                with a child ray or a surface colour that matters      */
    int64_t live_hits;       /* hits with non-zero weight in their pixel               */
    int64_t arena_rays_used; /* ray slots used, including tile padding                 */
    int64_t arena_capacity;  /* ray slots allocated                                    */
    int32_t n_nodes;         /* nodes of the static ray tree                           */
    int32_t n_levels;        /* bounce levels (longest path + 1)                       */
    int32_t overflow;        /* non-zero: arena too small, results invalid             */
    int32_t certify_mismatches; /* RT_PRECISION_MIXED_CHECKED only: certified float32 shadow
                                decisions that differ from the float64 decision (must be 0) */
    int64_t generator_words; /* last rt_generate_samples: 32-bit words of the stream it made ... */
    int64_t generator_words_stored; /* ... and how many of them it stored for packing (the words
                                of values that shaded hits of this context read, in runs of 64) */
} RtCounters;

/* Accumulated device time per kernel kind, filled when profiling is on. */
enum { RT_K_TRACE = 0, RT_K_SCAN = 1, RT_K_LINK = 2, RT_K_SURFACE = 3, RT_K_COMBINE = 4, RT_K_INIT = 5,
       RT_K_COUNT = 6 };
typedef struct RtProfile {
    double ms[RT_K_COUNT];
    int64_t launches[RT_K_COUNT];
} RtProfile;

typedef struct RtContext RtContext;

/* Last error message of the calling thread ("" if none). */
const char *rt_last_error(void);

/* Number of GPUs visible to the HIP runtime (0 when there is none; never fails). */
int rt_device_count(void);

/* Static shape of the ray tree for the given depth limits: the recursion
 * pattern of reference raytrace.py:103-133 (branch 0 transmit (a, b-1) when
 * b != 0; branch 1 total internal reflection (a-1, b-1) when a != 0 and b != 0;
 * branch 2 reflect (a-1, b) when a != 0), numbered breadth first.  Pure host
 * code (usable without a GPU).  Pass NULL arrays to query sizes only.
 * a, b, parent, branch, level: [n_nodes]; child: [n_nodes][3]; level_start:
 * [n_levels + 1]. */
int rt_build_tree(int32_t max_reflection_depth, int32_t max_refraction_depth,
                  int32_t *n_nodes, int32_t *n_levels,
                  int32_t *a, int32_t *b, int32_t *parent, int32_t *branch, int32_t *child,
                  int32_t *level, int32_t *level_start);

/* The grouping rt_set_scene builds over scenes with more than 64 spheres so that the nearest-hit
 * and shadow searches (reference objects.py:288-299 and :274-285, loops over every sphere) only test
 * spheres a ray can touch: blocks of 64 spheres made of 8 clusters of 8, each with a bounding
 * sphere; spheres far larger than the typical one stay "loose".  Pure host code (usable without a
 * GPU), exported so that the grouping can be checked.  sphere_geom: [n][4] centre and radius.
 * Pass NULL arrays to query the counts only.
 * sphere_of_slot: [n_blocks * 64] sphere index of every cluster slot (slot = block * 64 + cluster
 * * 8 + place), 0xFFFF for padding; cluster_bound: [n_blocks * 8][4] and block_bound: [n_blocks][4]
 * centre and radius in float32 as the device tests them (radius = bounding radius + the error
 * allowance of the float32 test; padding clusters have radius 0 and lie far away);
 * loose_spheres: [n_loose]. */
int rt_build_hierarchy(const double *sphere_geom, int64_t n, int32_t *n_blocks, int32_t *n_loose,
                       uint16_t *sphere_of_slot, float *cluster_bound, float *block_bound, int32_t *loose_spheres);

/* Create a context on GPU `device`.  `stream` is a hipStream_t to launch on
 * (e.g. torch.cuda.current_stream().cuda_stream) or NULL for a private one. */
int rt_create(int32_t device, void *stream, RtContext **out);
void rt_destroy(RtContext *ctx);

/* Upload the scene and render settings (replaces everything the reference
 * reads from `scene` and from its constants inside raytrace(), reference
 * raytrace.py:273-279). */
int rt_set_scene(RtContext *ctx, const RtScene *scene, const RtParams *params);

/* Choose the pixels of the next frame and size the ray arena.
 * Either a block of whole rows [row0, row0 + n_rows) of the width x height
 * grid (X = NULL, Y = NULL; pixel order is row-major as in reference
 * raytrace.py:274-276), or n explicit pixel coordinates X[i], Y[i] (the
 * get_pixel_color form, reference raytrace.py:6-7).  arena_rays is the number
 * of ray slots to provide (0 = keep / pick a default of 24 per pixel). */
int rt_begin_frame(RtContext *ctx, int32_t row0, int32_t n_rows,
                   const double *X, const double *Y, int64_t n, int64_t arena_rays);

/* Geometry phase (reference raytrace.py:19-133 without the lighting): trace
 * all bounce levels, leaving per-ray hit records and per-node counts on the
 * device.  Does not synchronise. */
int rt_trace_geometry(RtContext *ctx);

/* The same phase one level at a time, for callers that split a frame over
 * several GPUs: after rt_trace_level(l) the local batch sizes of level l+1 are
 * final; the caller sums them over all GPUs and hands the global sizes back
 * with rt_set_global_batches before tracing level l+1, because the
 * small-batch rule (reference raytrace.py:26-27) looks at whole-image batches. */
int rt_trace_level(RtContext *ctx, int32_t level);
int rt_get_level_batches(RtContext *ctx, int32_t level, int64_t *n_local /* [nodes of level] */);
int rt_set_global_batches(RtContext *ctx, int32_t level, const int64_t *n_global);

/* Alternative to the two calls above that needs no host round trip: keep the
 * whole-image batch sizes in caller-owned DEVICE memory, int32[n_nodes] (for
 * example a torch tensor).  rt_trace_level(l) then leaves the LOCAL sizes of
 * level l+1 in device_batches[level_start[l+1] .. level_start[l+2]) and reads
 * the sizes of level l from the same array; between the two calls the caller
 * sums that slice over all GPUs in place with a stream-ordered collective
 * (RCCL all-reduce on the context's stream).  NULL switches back to the
 * context's own array.  Call after rt_set_scene. */
int rt_set_batch_buffer(RtContext *ctx, int32_t *device_batches);

/* The same for several frames (contexts with the same recursion depths on one
 * GPU) whose small-batch sums (reference raytrace.py:26-27) are to travel in ONE
 * collective per level: device_batches is int32[n_slots * n_nodes], laid out
 * level by level and, inside a level, frame by frame; the sizes of node k (a
 * node of level l) of the frame in `slot` sit at
 *   n_slots * level_start[l] + slot * (level_start[l+1] - level_start[l]) + (k - level_start[l]),
 * so [n_slots * level_start[l], n_slots * level_start[l+1]) holds every frame's
 * sizes of level l.  n_slots == 1 is rt_set_batch_buffer. */
int rt_set_batch_buffer_shared(RtContext *ctx, int32_t *device_batches, int32_t slot, int32_t n_slots);

/* Speculation for callers that re-render similar frames on several GPUs and want ONE sum of the
 * batch sizes per frame instead of one per level: device_guess (same layout as the buffer of
 * rt_set_batch_buffer / _shared, read only) holds what the caller expects the whole-image batch
 * sizes to be, e.g. those of the previous frame, and the small-batch rule (reference
 * raytrace.py:26-27) of the following geometry passes goes by it; the kernels still write this
 * context's own sizes into the batch buffer.  After rt_trace_geometry the caller sums the buffer
 * over its GPUs: if (guess < SMALL_NUM) == (sum < SMALL_NUM) for every node, the frame was traced
 * exactly as with level-by-level sums; otherwise it has to be traced again (level by level, or
 * with the sums as the new guess).  NULL switches the guess off.  No counterpart in the reference. */
int rt_set_batch_guess(RtContext *ctx, const int32_t *device_guess);

/* Per-node results of the geometry phase (synchronises):
 * batch[k] local rays of node k, hits[k] local hits, beam[k][l] local hits
 * inside spotlight l's beam.  Arrays may be NULL. */
int rt_get_node_counts(RtContext *ctx, int64_t *batch, int64_t *hits, int64_t *beam /* [n_nodes][RT_MAX_LIGHTS] */);

/* Disk-light samples (reference objects.py:132-133 draws them from NumPy's
 * global stream; the caller draws them the same way and uploads them).
 * rt_set_sample_layout gives, per (node, light), the offset of that block in
 * this context's sample buffer (-1: none) - layout inside a block:
 * u[s * h + j] is the angle uniform and u[N * h + s * h + j] the radius
 * uniform of sample s for the j-th (in-beam) hit of the node, h = this
 * context's (in-beam) hit count, N = n_points - and the total length. */
int rt_set_sample_layout(RtContext *ctx, const int64_t *offsets /* [n_nodes][RT_MAX_LIGHTS] */, int64_t total);
int rt_upload_samples(RtContext *ctx, int64_t dst_offset, const double *u, int64_t n);

/* rt_upload_samples for callers that must not allocate device memory (tight serving loops,
 * harnesses that forbid hipMalloc, processes whose memory pool is fixed at start-up).  No
 * counterpart in the reference.  Same arguments, meaning and result as rt_upload_samples, but the
 * piece of the stream is staged in an array that is part of the library's code object (one per
 * GPU, 8 MiB, held until the process ends) instead of a buffer that grows on demand: the call
 * makes no hipMalloc / hipFree.  Like rt_upload_samples it waits for the context's stream before
 * it overwrites the array and again before it returns (u is ordinary host memory and may be reused
 * at once), so it cannot be recorded into a HIP graph: a stream capture that is active on the
 * context's stream is invalidated by it.  n must not exceed RT_NOALLOC_MAX_SAMPLES; longer streams
 * are uploaded in several calls.  All contexts on one GPU share the array: calls on the same GPU
 * must not overlap in time. */
#define RT_NOALLOC_MAX_SAMPLES (1 << 20)
int rt_upload_samples_noalloc(RtContext *ctx, int64_t dst_offset, const double *u, int64_t n);

/* The same samples made on the device instead of drawn on the host and uploaded: the
 * whole stream of rt_set_sample_layout's `total` values is generated from a NumPy legacy
 * MT19937 state (numpy.random.get_state(): key[624], pos) exactly as
 * numpy.random.random(total) would produce it (reference objects.py:132-133 consumes that
 * stream), and key_out / pos_out receive the state those draws leave behind (for
 * numpy.random.set_state), so the caller's stream continues as in the reference.
 * A context that renders only part of the image (row blocks on several GPUs) first tells
 * where its columns sit in the whole image's stream: per (node, light) the block's offset
 * there, the whole image's number of (in-beam) hits and the rank of this context's first hit
 * among them (rt_set_sample_layout_global, after rt_set_sample_layout; `total` = length of
 * the whole stream).  Every context then generates the whole stream and keeps its columns,
 * and all of them return the same final state.  The generator's work buffer (8 bytes per value of
 * the stream) stays with the context between calls while it is at most 64 GB (RT_MT_RAW_KEEP_GB
 * in the environment; rt_trim releases it at once); longer streams, or ones that half of the free
 * device memory does not hold, are made in several passes through a buffer that is released again.
 * Words of values that no shaded hit of this context reads are not stored in it (RT_MT_STORE_ALL=1: all are).
 * In mixed precision, with a buffer that stays, the samples are kept as the disk points of the float32 pass plus
 * these raw words (the few float64 readers build their uniforms from them); rt_trim, an upload over generated
 * samples and RT_PRECISION_MIXED_CHECKED unpack them into the float64 array first (RT_PACK_UNIFORMS=1: always).
 * RT_MT_FUSED=1 (disk lights only): the values go straight to their places, no such buffer at all,
 * at about 1.6 times the kernel time. */
int rt_set_sample_layout_global(RtContext *ctx, const int64_t *offsets, const int64_t *columns,
                                const int64_t *first_column /* each [n_nodes][RT_MAX_LIGHTS] */, int64_t total);
int rt_generate_samples(RtContext *ctx, const uint32_t *key /* [624] */, int32_t pos,
                        uint32_t *key_out /* [624] */, int32_t *pos_out);

/* Host only (checks without a GPU): what rt_set_scene builds over a scene with more than 64 spheres
 * (the reference loops over all spheres for every ray, objects.py:288-299 and :274-285; these
 * structures only narrow down which spheres can matter, the float64 test of the reference decides).
 * rt_build_grid: the uniform grid.  dims[3] cells per axis (0: no grid), origin_cell[4] = the box's
 * lower corner and the cell size, the spheres kept out of the grid (always tested), and per cell
 * the spheres whose ball, enlarged by the registration slack, reaches it: cell_first[cells + 1],
 * items[*n_items], cell index = (z * dims[1] + y) * dims[0] + x.  Output pointers may be NULL (call
 * once for the sizes).  density: cells per sphere aimed at (<= 0: the default, 2).
 * rt_build_light_map: occluders of a point light at light[3] by direction as seen from the light,
 * 6 * N * N bins (face = largest component of the direction: +x, -x, +y, -y, +z, -z; bin =
 * (face * N + iw) * N + iu with u, w the next two components divided by the largest, mapped from
 * [-1, 1] to [0, N)); in_grid[n] (NULL: all) tells which spheres have a screening entry, the others
 * and those that contain the light or reach very many bins are listed in always[]
 * (*n_always = -1: too many, such a light gets no map). */
int rt_build_grid(const double *sphere_geom /* [n][4] */, int64_t n, double density, int32_t *dims, double *origin_cell,
                  int32_t *n_loose, int32_t *loose /* [n] */, int64_t *n_items, int32_t *cell_first, int32_t *items);
int rt_build_light_map(const double *sphere_geom /* [n][4] */, int64_t n, const double *light, const uint8_t *in_grid,
                       int32_t *bins_per_edge, int32_t *n_always, int32_t *always /* [n] */, int64_t *n_items,
                       int32_t *bin_first /* [6 N N + 1] */, int32_t *items);

/* Gives back the work buffers of the sample calls (the generator's raw words and stream states, the
 * staging buffer of rt_upload_samples): they are allocated again when next needed.  No counterpart in
 * the reference.  Waits for the context's stream.  Samples that live in the raw words are unpacked first
 * (one more kernel; RT_ERR_STATE under stream capture), and launches of rt_shade recorded in a graph
 * before the call stay valid. */
int rt_trim(RtContext *ctx);

/* Host only: the coefficients (bit i = coefficient of x^i, 624 words) of x^(2^k) mod the
 * characteristic polynomial of MT19937's one-word step - the jump-ahead polynomials
 * rt_generate_samples uses to start its parallel streams.  0 <= k < 48. */
int rt_mt19937_jump_poly(int32_t k, uint32_t *out624);

/* Arithmetic of the disk-light sample loop in rt_shade.
 * RT_PRECISION_F64: every sample in float64, the reference's operations.
 * RT_PRECISION_MIXED (default): samples in float32 with a rigorous error bound on
 * every shadow-ray discriminant; a sample with a decision inside its bound is
 * re-evaluated in float64, so shadow decisions are those of the float64 path and
 * colours differ by float32 rounding of continuous terms only (< 1e-5).
 * The float32 pass takes scenes of up to 64 spheres; larger ones are evaluated in float64.
 * RT_PRECISION_MIXED_CHECKED (tests): as MIXED, and every certified decision is compared
 * with the float64 decision on the device; RtCounters.certify_mismatches counts differences.
 * Geometry (hits, child rays, counts) is float64 in all modes. */
enum { RT_PRECISION_F64 = 0, RT_PRECISION_MIXED = 1, RT_PRECISION_MIXED_CHECKED = 2 };
int rt_set_precision(RtContext *ctx, int32_t mode);

/* Lighting, shading and bottom-up colour combination (reference
 * raytrace.py:135-170 and objects.py:94-235), deepest level first.  Leaves the
 * clipped pixel colours (reference raytrace.py:15) on the device.  Does not
 * synchronise.  With disk lights in the scene: RT_ERR_STATE unless the whole sample stream
 * announced by rt_set_sample_layout has been uploaded or generated.  The first call after new
 * samples also converts them for the float32 pass and must not be made under stream capture
 * (RT_ERR_STATE); later calls can be captured into a HIP graph. */
int rt_shade(RtContext *ctx);

/* Optional: have rt_shade write the colours into caller-owned DEVICE memory
 * ([n_pixels][3] float64, e.g. a torch tensor's data_ptr) instead of the
 * context's own buffer.  NULL switches back.  Call after rt_begin_frame. */
int rt_set_output(RtContext *ctx, void *device_rgb);

/* Copy the frame's colours, [n_pixels][3] float64 in pixel order (synchronises): the array that
 * reference raytrace.py:278 returns.  Into memory from rt_alloc_host the copy is one transfer at the
 * host link's rate; other host memory is filled through page-locked pieces of the context. */
int rt_read_image(RtContext *ctx, double *out_rgb);
/* Page-locked host memory for images (a current GPU is needed; no counterpart in the reference,
 * whose arrays are ordinary NumPy memory).  Free with rt_free_host. */
int rt_alloc_host(size_t bytes, void **out);
void rt_free_host(void *p);
/* The reference's output step (reference main.py:32, matplotlib imsave of the float
 * image): 8-bit channels, value = (uint8)(v * 255) of the clipped colour, [n_pixels][3].
 * rt_quantise_rgb8 converts the shaded frame on the device into caller-owned DEVICE
 * memory (NULL: a buffer of the context) without synchronising - the form a multi-GPU
 * caller gathers (3 instead of 24 bytes per pixel); rt_read_image_rgb8 converts into the context's buffer and copies it to the
 * host (synchronises).  Call after rt_shade. */
int rt_quantise_rgb8(RtContext *ctx, void *device_rgb8);
int rt_read_image_rgb8(RtContext *ctx, uint8_t *out_rgb8);

/* Device address of the same buffer (valid until the next rt_begin_frame). */
int rt_image_device_ptr(RtContext *ctx, void **out);

/* Wait for all work queued by this context. */
int rt_synchronize(RtContext *ctx);

int rt_get_counters(RtContext *ctx, RtCounters *out);

/* Per-kernel device timing with HIP events on the context's stream. */
int rt_profile_enable(RtContext *ctx, int32_t on);
int rt_get_profile(RtContext *ctx, RtProfile *out, int32_t reset);

/* Test hook: evaluate, on the GPU, one of the device functions the frame kernels are built from
 * for n caller-supplied input rows (host arrays, row-major doubles; synchronises).  Uses the scene
 * of the last rt_set_scene.  Rows in -> out:
 *   RT_DEBUG_SPHERE_ROOTS  o[3] d[3] sphere -> closest t, furthest t (-1: none)
 *                          reference objects.py:60-66 Sphere.intersection, :238-271 solve_quadratic
 *   RT_DEBUG_NEAREST       o[3] d[3] -> sphere (-1: miss), t (inf: miss); at most 64 spheres
 *                          reference objects.py:288-299 find_closest_intersected_object
 *   RT_DEBUG_SHADOW        o[3] d[3] -> RGB multiplier; at most 128 spheres
 *                          reference objects.py:274-285 shadow_objects_multipliers
 *   RT_DEBUG_OPTICS        N[3] d[3] n1 n2 -> T[3], transmitted (0/1), R, mirror direction[3]
 *                          reference raytrace.py:198-212 refract_vectors, :173-189 fresnel_multiplier
 *                          (R = 1 for totally reflected rays, raytrace.py:99), :192-195 reflect_vectors
 *   RT_DEBUG_EASE          mode x a d -> factor
 *                          reference objects.py:176-191 ease_fall_off_beam, :302-338
 *   RT_DEBUG_SCREEN        X Y -> screen point[3], unit ray direction[3]
 *                          reference objects.py:45-51 index_to_position, raytrace.py:7-10 */
enum { RT_DEBUG_SPHERE_ROOTS = 0, RT_DEBUG_NEAREST = 1, RT_DEBUG_SHADOW = 2, RT_DEBUG_OPTICS = 3, RT_DEBUG_EASE = 4,
       RT_DEBUG_SCREEN = 5 };
int rt_debug_eval(RtContext *ctx, int32_t what, const double *in, double *out, int64_t n);

/* Convenience: rt_begin_frame(whole image) + geometry + shade + read, for
 * scenes without disk lights (no samples needed). */
int rt_render(RtContext *ctx, double *out_rgb);

#ifdef __cplusplus
}
#endif
#endif /* RAYTRACING_AMD_H */
