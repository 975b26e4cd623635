// Host side of the C ABI declared in include/raytracing_amd.h: context, memory,
// launch sequence.  The kernels are compiled into the same translation unit.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "host_tree.h"
#include "mt_jump.h"
#include "rt_kernels.hip"
#include "hier_build.h"
#include "grid_build.h"

using namespace rt;

#ifndef RT_ALLOC_FILL_DEFAULT
#define RT_ALLOC_FILL_DEFAULT 1   // see alloc_treatment()
#endif

// Staging memory of rt_upload_samples_noalloc: one array per GPU, part of the loaded code object,
// so using it needs no allocation.  Host code must not use the array's name as an address (it is a
// host-side stand-in): upload_scratch_of() asks the runtime for the array's device address, once
// per device with that device current, and everything else uses that pointer.
static __device__ __align__(256) double g_upload_scratch[RT_NOALLOC_MAX_SAMPLES];

namespace {

thread_local std::string g_error;

constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
double* g_scratch_ptr[MAX_DEVICES] = {nullptr};

// Device address of g_upload_scratch on `device`, which must be the current device.
hipError_t upload_scratch_of(int device, double** out) {
    *out = nullptr;
    if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_upload_scratch));
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = (double*)p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(RT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// Breadth-first enumeration of the recursion pattern of reference raytrace.py:103-133 (host_tree.h).
static_assert(HOST_TREE_MAX_NODES == MAX_NODES, "the host tree and the device tables agree on the node limit");
int build_tree(int refl, int refr, HostTree& t) {
    const std::string problem = build_host_tree(refl, refr, t);
    if (!problem.empty()) return fail(RT_ERR_INVALID, "%s", problem.c_str());
    return RT_OK;
}

// Writes a buffer once from end to end: zeros, or (garbage != 0, an experiment's control) a word
// pattern with about half of its bits set.
__global__ __launch_bounds__(256) void k_fill(uint32_t* p, size_t words, uint32_t garbage) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) {
        uint32_t v = 0u;
        if (garbage) {
            v = (uint32_t)i * 2654435761u + garbage;
            v ^= v >> 15; v *= 2246822519u; v ^= v >> 13;
        }
        p[i] = v;
    }
}

// Stream of the context whose call is allocating on this thread (set by use_device): the first
// write of a new buffer is queued there.
thread_local hipStream_t g_alloc_stream = nullptr;

// What DevBuf::ensure does with a large, lasting buffer after hipMalloc (RT_ALLOC_FILL):
// bit 0: write it once from end to end (zeros); bit 1: give it back and ask again first;
// 4: write a word pattern instead of zeros (control of the measurement described in ensure()).
int alloc_treatment() {
    static const int mode = [] {
        const char* e = getenv("RT_ALLOC_FILL");
        return e ? atoi(e) : RT_ALLOC_FILL_DEFAULT;
    }();
    return mode;
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    bool transient = false;   // a work buffer of one call: released again, no treatment on allocation
    hipError_t ensure(size_t count) {
        if (count <= n && p) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
        const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
        hipError_t e = hipMalloc((void**)&p, bytes);
        // A large buffer that lasts is written once from end to end before its first use.  Measured on
        // MI355X (profiles/r3/alloc_experiment.txt, 150 steps of the 4K benchmark per process, six
        // processes in a row on a freshly started machine): k_trace needs 11.6 ms per step on buffers
        // that were never written as a whole (the arena is larger than what a frame fills, and hit
        // records are stored for some lanes of a tile only) and 11.1-11.2 ms after one pass of writes -
        // zeros or an arbitrary bit pattern alike, so it is not the data; in the machine's first
        // process and in later ones alike, so it is not the driver's clearing of reused memory.  Giving
        // the buffer back and asking again (which an earlier version did on top) adds nothing that
        // can be told from run-to-run spread, and costs seconds of set-up.  RT_ALLOC_FILL=0 turns the
        // pass off (2 and 3 bring the second allocation back, 4 writes the pattern).
        const int mode = alloc_treatment();
        const bool lasting = mode != 0 && !transient && bytes >= (1u << 20);
        if (e == hipSuccess && lasting && (mode & 2) && mode < 4 && bytes >= ((size_t)64 << 20)) {
            (void)hipFree(p);
            p = nullptr;
            e = hipMalloc((void**)&p, bytes);
        }
        if (e == hipSuccess && lasting && ((mode & 1) || mode >= 4)) {
            hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, g_alloc_stream, reinterpret_cast<uint32_t*>(p), bytes / 4,
                               mode >= 4 ? 0x9e3779b9u : 0u);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(g_alloc_stream);
        }
        if (e == hipSuccess) n = count;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

}  // namespace

struct RtContext {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // rt_generate_samples: the jump-ahead runs beside the hit mask and the store map
    hipStream_t side_stream = nullptr;
    hipEvent_t side_fork = nullptr, side_join = nullptr;

    bool have_scene = false, frame_begun = false, geometry_done = false, layout_set = false, shaded = false;
    int next_level = 0;
    bool sampled_lights = false;
    SceneConst sc_host{};
    RtParams params{};
    HostTree tree;
    int S = 0;

    DevBuf<double> geom, mat, fres;
    DevBuf<float> cull;
    double cull_ref[3] = {0, 0, 0};
    DevBuf<Group8> hier_top, hier_mid, hier_leaf;
    DevBuf<uint32_t> hier_top_ref;
    DevBuf<uint16_t> hier_ids;
    int hier_top_groups = 0, hier_blocks = 0;
    float hier_window = 0, hier_extent = 0;
    DevBuf<uint32_t> grid_start;
    DevBuf<uint16_t> grid_items, grid_loose;
    DevBuf<float> grid_sph;
    DevBuf<uint32_t> lmap_start;
    DevBuf<uint16_t> lmap_items, lmap_always;
    int lmap_start_off[RT_MAX_LIGHTS] = {0}, lmap_always_off[RT_MAX_LIGHTS] = {0}, lmap_n_always[RT_MAX_LIGHTS] = {0};
    uint32_t lmap_mask = 0;
    int grid_cells = 0, grid_n_loose = 0, grid_n[3] = {0, 0, 0};
    float grid_h = 0, grid_inv_h = 0, grid_margin = 0;
    double grid_origin[3] = {0, 0, 0}, grid_hi[3] = {0, 0, 0};
    DevBuf<SceneConst> sc;
    DevBuf<int32_t> node_a, node_b, node_branch, node_child, level_start;
    DevBuf<int32_t> node_n, node_nglobal, node_hits, node_beam, node_tile0, node_ntiles, lvl_ntiles, lvl_tilebase, flags;
    DevBuf<uint32_t> node_start, node_pre, tile_run, blk_tot, wave_off;
    DevBuf<double> px, py, pz, dx, dy, dz, A, dist, cr, cg, cb, sr, sg, sb;
    DevBuf<uint32_t> info, link, list_ray, redo_list, list_j[1 + RT_MAX_LIGHTS];
    DevBuf<uint16_t> list_node;
    DevBuf<int32_t> lvl_shade_base;
    uint32_t spot_mask = 0;
    bool disk_lights = false;
    DevBuf<double> X, Y, packed, staging, image;
    DevBuf<float> packed32;
    bool have_packed32 = false;
    bool disk_points_ready = false;   // packed32 holds the disk points of the current contents of packed
    // The generated samples exist as disk points (packed32) and as the generator's raw words, `packed` has not
    // been filled: value i of the stream is made of the words mt_raw[raw_value0 + 2 i], + 1 (materialize_packed fills it)
    bool samples_in_raw = false;
    long long raw_value0 = 0;
    DevBuf<const uint32_t*> raw_slot;          // what the kernels read: mt_raw + raw_value0, or null (Dev::raw_values_at)
    const uint32_t* raw_slot_host = nullptr;   // source of its updates
    int64_t pack32_base[RT_MAX_LIGHTS] = {0};
    bool simple_shading = false;   // see SampleSums in rt_lighting_mixed.hip: coefficients in [0, 1], grey lights
    int64_t pack_base[RT_MAX_LIGHTS] = {0};
    int64_t pack_E = 0;
    int64_t pack_plane = 0;   // doubles from a sample's angle uniform to its radius uniform in `packed`
    DevBuf<int64_t> sample_off, sample_global;
    std::vector<int64_t> sample_off_host, sample_global_host;   // what rt_set_sample_layout[_global] was given
    DevBuf<unsigned long long> hit_mask;    // work buffers of the fused generator (k_mt_emit)
    DevBuf<uint32_t> hit_pref, hit_word0;
    DevBuf<EmitBlock> emit_blocks;
    bool have_global_layout = false;
    int64_t global_total = 0;
    DevBuf<unsigned long long> work;
    double* user_image = nullptr;
    DevBuf<uint8_t> image8;
    char* bounce = nullptr;          // page-locked staging for reads into pageable host memory (2 x BOUNCE_BYTES)
    hipEvent_t bounce_done[2] = {nullptr, nullptr};
    DevBuf<uint32_t> mt_ring, mt_raw, mt_polys;
    DevBuf<long long> mt_end;
    DevBuf<unsigned long long> mt_need;     // store map of the generator (k_store_map)
    DevBuf<unsigned long long> mt_stat;
    DevBuf<uint32_t> mt_ring0;              // the streams' start windows, kept while a stream is made in passes of whole blocks
    DevBuf<long long> mt_end0;
    DevBuf<uint8_t> pass_of;                // [node][light]: the pass that holds the block
    std::vector<int32_t> node_hits_host, node_beam_host;   // the hits (in-beam hits) per node as rt_get_node_counts last read them ...
    bool node_hits_host_valid = false;      // ... and no level 0 has been traced since
    int mt_polys_ell = -1, mt_polys_n = 0;  // what mt_polys holds: the polynomials x^(2^(ell + b)), b < n
    long long generator_words = 0, generator_words_stored = 0;   // of the last rt_generate_samples
    uint8_t* user_image8 = nullptr;
    int32_t* user_batches = nullptr;
    const int32_t* batch_guess = nullptr;
    int batch_slot = 0, batch_slots = 1;

    // first element of a level's whole-image batch sizes (see whole_image_batch in rt_geometry.hip)
    int32_t* level_batches(int level) {
        const int a = tree.level_start[level], b = tree.level_start[level + 1];
        return (user_batches ? user_batches : node_nglobal.p) + a + (batch_slots - 1) * a + batch_slot * (b - a);
    }

    int64_t npix = 0;
    int row0 = 0;
    bool explicit_xy = false;
    int64_t arena_rays = 0;
    int nb = NB;   // workgroups per level launch of the current frame (see rt_begin_frame)
    int64_t sample_total = 0;
    int64_t samples_delivered = 0;   // values uploaded or generated since the layout was set
    double* upload_scratch = nullptr;   // this GPU's g_upload_scratch (see upload_scratch_of)

    int precision = RT_PRECISION_MIXED;
    bool frame_col32 = false;   // how the current frame's smooth arrays are stored (Dev::col32): chosen when its geometry starts
    bool profiling = false;
    struct Stamp { int kind; hipEvent_t t0, t1; };
    std::vector<Stamp> stamps;
    std::vector<hipEvent_t> free_events;
    RtProfile prof{};

    Dev dev() const {
        Dev D{};
        D.geom = geom.p; D.mat = mat.p; D.fres = fres.p; D.cull = cull.p; D.cull_ref[0] = cull_ref[0]; D.cull_ref[1] = cull_ref[1]; D.cull_ref[2] = cull_ref[2]; D.sc = sc.p; D.S = S; D.n_lights = sc_host.n_lights;
        D.hier_top = hier_top.p; D.hier_top_ref = hier_top_ref.p; D.hier_mid = hier_mid.p; D.hier_leaf = hier_leaf.p; D.hier_ids = hier_ids.p; D.hier_top_groups = hier_top_groups; D.hier_blocks = hier_blocks; D.hier_window = hier_window; D.hier_extent = hier_extent;
        D.grid_start = grid_start.p; D.grid_items = grid_items.p; D.grid_sph = reinterpret_cast<const float4*>(grid_sph.p); D.grid_loose = grid_loose.p;
        D.grid_n_loose = grid_n_loose; D.grid_cells = grid_cells; D.grid_h = grid_h; D.grid_inv_h = grid_inv_h; D.grid_margin = grid_margin;
        for (int k = 0; k < 3; ++k) { D.grid_n[k] = grid_n[k]; D.grid_origin[k] = grid_origin[k]; D.grid_hi[k] = grid_hi[k]; }
        D.lmap_start = lmap_start.p; D.lmap_items = lmap_items.p; D.lmap_always = lmap_always.p; D.lmap_mask = lmap_mask;
        for (int l = 0; l < RT_MAX_LIGHTS; ++l) { D.lmap_start_off[l] = lmap_start_off[l]; D.lmap_always_off[l] = lmap_always_off[l]; D.lmap_n_always[l] = lmap_n_always[l]; }
        D.node_a = node_a.p; D.node_b = node_b.p; D.node_branch = node_branch.p; D.node_child = node_child.p;
        D.level_start = level_start.p; D.K = tree.n_nodes(); D.n_levels = tree.n_levels();
        D.node_n = node_n.p; D.node_nglobal = user_batches ? user_batches : node_nglobal.p; D.nb = nb; D.node_nguess = batch_guess; D.batch_slot = batch_slot; D.batch_slots = batch_slots; D.node_hits = node_hits.p; D.node_beam = node_beam.p;
        D.node_tile0 = node_tile0.p; D.node_ntiles = node_ntiles.p; D.node_start = node_start.p; D.node_pre = node_pre.p;
        D.lvl_ntiles = lvl_ntiles.p; D.lvl_tilebase = lvl_tilebase.p; D.tile_run = tile_run.p; D.blk_tot = blk_tot.p; D.wave_off = wave_off.p;
        D.flags = flags.p; D.work = work.p;
        D.px = px.p; D.py = py.p; D.pz = pz.p; D.dx = dx.p; D.dy = dy.p; D.dz = dz.p; D.A = A.p; D.dist = dist.p;
        D.col32 = frame_col32 ? 1 : 0;
        D.info = info.p; D.link = link.p; D.cr = cr.p; D.cg = cg.p; D.cb = cb.p; D.sr = sr.p; D.sg = sg.p; D.sb = sb.p;
        D.cap_tiles = (uint32_t)(arena_rays / TILE);
        D.lvl_shade_base = lvl_shade_base.p; D.list_ray = list_ray.p; D.list_node = list_node.p; D.redo_list = redo_list.p;
        for (int l = 0; l <= RT_MAX_LIGHTS; ++l) D.list_j[l] = list_j[l].n ? list_j[l].p : nullptr;
        D.spot_mask = spot_mask;
        D.X = explicit_xy ? X.p : nullptr; D.Y = explicit_xy ? Y.p : nullptr;
        D.row0 = row0; D.npix = (int32_t)npix;
        D.packed = packed.p; D.pack_plane = pack_plane; D.packed32 = have_packed32 ? packed32.p : nullptr; D.raw_values_at = raw_slot.p; D.pack_E = pack_E; D.sample_off = sample_off.p; D.sample_global = have_global_layout ? sample_global.p : nullptr;
        for (int l = 0; l < RT_MAX_LIGHTS; ++l) { D.pack_base[l] = pack_base[l]; D.pack32_base[l] = pack32_base[l]; }
        D.image = user_image ? user_image : image.p;
        return D;
    }

    hipEvent_t get_event() {
        if (!free_events.empty()) {
            hipEvent_t e = free_events.back();
            free_events.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    void begin(int kind) {
        if (!profiling) return;
        Stamp s{kind, get_event(), get_event()};
        (void)hipEventRecord(s.t0, stream);
        stamps.push_back(s);
    }
    void end() {
        if (!profiling) return;
        (void)hipEventRecord(stamps.back().t1, stream);
    }
    void collect() {
        for (auto& s : stamps) {
            float ms = 0;
            if (hipEventSynchronize(s.t1) == hipSuccess && hipEventElapsedTime(&ms, s.t0, s.t1) == hipSuccess) {
                prof.ms[s.kind] += ms;
                prof.launches[s.kind] += 1;
            }
            free_events.push_back(s.t0);
            free_events.push_back(s.t1);
        }
        stamps.clear();
    }
};

namespace {

int use_device(RtContext* ctx) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    g_alloc_stream = ctx->stream;
    return RT_OK;
}

// Dynamic LDS of the bookkeeping kernels: the per-node tables of one level.
size_t level_table_bytes(RtContext* ctx, int level) {
    return (size_t)(ctx->tree.level_start[level + 1] - ctx->tree.level_start[level]) * LEVEL_TABLE_BYTES;
}

// Levels wider than ~680 nodes need more than the default 64 KiB of dynamic LDS.  The limit is a
// property of the kernel on the current device, so every context raises it on its own device.
hipError_t raise_dynamic_lds_limit() {
    const int most = MAX_LEVEL_NODES * LEVEL_TABLE_BYTES;
    if (hipError_t e = hipFuncSetAttribute((const void*)k_link, hipFuncAttributeMaxDynamicSharedMemorySize, most)) return e;
    if (hipError_t e = hipFuncSetAttribute((const void*)k_hitmask, hipFuncAttributeMaxDynamicSharedMemorySize, most)) return e;
    if (hipError_t e = hipFuncSetAttribute((const void*)k_combine<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, most)) return e;
    if (hipError_t e = hipFuncSetAttribute((const void*)k_combine<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, most)) return e;
    if (hipError_t e = hipFuncSetAttribute((const void*)k_combine<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, most)) return e;
    return hipFuncSetAttribute((const void*)k_combine<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, most);
}

int launch_level(RtContext* ctx, int level) {
    const Dev D = ctx->dev();
    ctx->begin(RT_K_TRACE);
    if (ctx->S <= SPH_CHUNK_SMALL) hipLaunchKernelGGL((k_trace<true>), dim3(ctx->nb), dim3(BLOCK), 0, ctx->stream, D, level);
    else if (ctx->grid_cells) hipLaunchKernelGGL((k_trace<false, true>), dim3(ctx->nb), dim3(BLOCK), 0, ctx->stream, D, level);
    else hipLaunchKernelGGL((k_trace<false>), dim3(ctx->nb), dim3(BLOCK), 0, ctx->stream, D, level);
    ctx->end();
    ctx->begin(RT_K_SCAN);
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, ctx->stream, D, level);
    ctx->end();
    ctx->begin(RT_K_LINK);
    hipLaunchKernelGGL(k_link, dim3(ctx->nb), dim3(BLOCK), level_table_bytes(ctx, level), ctx->stream, D, level);
    ctx->end();
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

// Device-to-host copy of a frame.  Into page-locked memory (rt_alloc_host) it is one transfer at the
// link's rate; into pageable memory the runtime's own staging is several times slower, so the
// data goes through two page-locked pieces of the context instead, one travelling while the other
// is copied out.  Waits for the stream.
constexpr size_t BOUNCE_BYTES = 8u << 20;
int copy_to_host(RtContext* ctx, void* dst, const void* src_device, size_t bytes) {
    hipPointerAttribute_t attr{};
    const bool pinned = hipPointerGetAttributes(&attr, dst) == hipSuccess && attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();   // pageable memory is reported as an error; it is none
    if (pinned || bytes <= BOUNCE_BYTES) {
        HIP_TRY(hipMemcpyAsync(dst, src_device, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return RT_OK;
    }
    if (!ctx->bounce) {
        HIP_TRY(hipHostMalloc((void**)&ctx->bounce, 2 * BOUNCE_BYTES, hipHostMallocDefault));
        HIP_TRY(hipEventCreateWithFlags(&ctx->bounce_done[0], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ctx->bounce_done[1], hipEventDisableTiming));
    }
    const size_t pieces = (bytes + BOUNCE_BYTES - 1) / BOUNCE_BYTES;
    for (size_t i = 0; i <= pieces; ++i) {
        if (i < pieces) {
            const size_t off = i * BOUNCE_BYTES, len = std::min(BOUNCE_BYTES, bytes - off);
            HIP_TRY(hipMemcpyAsync(ctx->bounce + (i & 1) * BOUNCE_BYTES, (const char*)src_device + off, len, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipEventRecord(ctx->bounce_done[i & 1], ctx->stream));
        }
        if (i > 0) {
            const size_t j = i - 1, off = j * BOUNCE_BYTES, len = std::min(BOUNCE_BYTES, bytes - off);
            HIP_TRY(hipEventSynchronize(ctx->bounce_done[j & 1]));
            memcpy((char*)dst + off, ctx->bounce + (j & 1) * BOUNCE_BYTES, len);
        }
    }
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_error.c_str(); }

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rt_build_tree(int32_t refl, int32_t refr, int32_t* n_nodes, int32_t* n_levels, int32_t* a, int32_t* b,
                  int32_t* parent, int32_t* branch, int32_t* child, int32_t* level, int32_t* level_start) {
    HostTree t;
    int rc = build_tree(refl, refr, t);
    if (rc) return rc;
    if (n_nodes) *n_nodes = t.n_nodes();
    if (n_levels) *n_levels = t.n_levels();
    const size_t K = t.a.size();
    if (a) memcpy(a, t.a.data(), K * 4);
    if (b) memcpy(b, t.b.data(), K * 4);
    if (parent) memcpy(parent, t.parent.data(), K * 4);
    if (branch) memcpy(branch, t.branch.data(), K * 4);
    if (child) memcpy(child, t.child.data(), K * 12);
    if (level) memcpy(level, t.level.data(), K * 4);
    if (level_start) memcpy(level_start, t.level_start.data(), t.level_start.size() * 4);
    return RT_OK;
}

int rt_build_hierarchy(const double* sphere_geom, int64_t n, int32_t* n_blocks, int32_t* n_loose, uint16_t* sphere_of_slot,
                       float* cluster_bound, float* block_bound, int32_t* loose_spheres) {
    if (n < 0 || n >= 0xFFFF || (n > 0 && !sphere_geom)) return fail(RT_ERR_INVALID, "sphere_geom must hold 0 <= n < 65535 spheres");
    const HostHierarchy H = build_hierarchy(sphere_geom, (size_t)n);
    if (n_blocks) *n_blocks = H.blocks;
    if (n_loose) *n_loose = H.loose;
    if (sphere_of_slot) memcpy(sphere_of_slot, H.ids.data(), H.ids.size() * sizeof(uint16_t));
    auto entry = [](const Group8& g, int k, float* out) { out[0] = g.cx[k]; out[1] = g.cy[k]; out[2] = g.cz[k]; out[3] = g.re[k]; };
    if (cluster_bound)
        for (int c = 0; c < H.blocks * 8; ++c) entry(H.mid[c / 8], c % 8, cluster_bound + (size_t)c * 4);
    if (block_bound)
        for (int b = 0; b < H.blocks; ++b) entry(H.top[b / 8], b % 8, block_bound + (size_t)b * 4);
    if (loose_spheres)
        for (size_t e = 0, i = 0; e < H.top_ref.size(); ++e)
            if (H.top_ref[e] != HIER_NONE && (H.top_ref[e] & HIER_LOOSE)) loose_spheres[i++] = (int32_t)(H.top_ref[e] & 0xffffu);
    return RT_OK;
}

int rt_build_grid(const double* sphere_geom, int64_t n, double density, int32_t* dims, double* origin_cell, int32_t* n_loose,
                  int32_t* loose, int64_t* n_items, int32_t* cell_first, int32_t* items) {
    if (n < 0 || n >= 0xFFFF || (n > 0 && !sphere_geom)) return fail(RT_ERR_INVALID, "sphere_geom must hold 0 <= n < 65535 spheres");
    const HostGrid G = build_grid(sphere_geom, (size_t)n, density > 0 ? density : 2.0);
    const size_t cells = G.start.empty() ? 0 : (size_t)G.cells();
    if (dims) for (int k = 0; k < 3; ++k) dims[k] = cells ? G.n[k] : 0;
    if (origin_cell) { for (int k = 0; k < 3; ++k) origin_cell[k] = G.origin[k]; origin_cell[3] = G.h; }
    if (n_loose) *n_loose = (int32_t)G.loose.size();
    if (loose) for (size_t i = 0; i < G.loose.size(); ++i) loose[i] = G.loose[i];
    // the lists without their padding
    int64_t total = 0;
    for (size_t c = 0; c < cells; ++c) {
        if (cell_first) cell_first[c] = (int32_t)total;
        for (uint32_t k = G.start[2 * c]; k < G.start[2 * c + 1]; ++k)
            if (G.items[k] != (uint16_t)n) {
                if (items) items[total] = G.items[k];
                ++total;
            }
    }
    if (cell_first) cell_first[cells] = (int32_t)total;
    if (n_items) *n_items = total;
    return RT_OK;
}

int rt_build_light_map(const double* sphere_geom, int64_t n, const double* light, const uint8_t* in_grid, int32_t* bins_per_edge,
                       int32_t* n_always, int32_t* always, int64_t* n_items, int32_t* bin_first, int32_t* items) {
    if (n < 0 || n >= 0xFFFF || (n > 0 && !sphere_geom) || !light) return fail(RT_ERR_INVALID, "bad arguments");
    std::vector<char> flags((size_t)n, 1);
    if (in_grid) for (int64_t i = 0; i < n; ++i) flags[i] = in_grid[i] ? 1 : 0;
    const HostLightMap M = build_light_map(sphere_geom, (size_t)n, light, flags);
    if (bins_per_edge) *bins_per_edge = LMAP_N;
    if (n_always) *n_always = M.usable ? (int32_t)M.always.size() : -1;   // -1: too many, no map is used for this light
    if (always && M.usable) for (size_t i = 0; i < M.always.size(); ++i) always[i] = M.always[i];
    const size_t bins = (size_t)6 * LMAP_N * LMAP_N;
    int64_t total = 0;
    if (M.usable)
        for (size_t b = 0; b < bins; ++b) {
            if (bin_first) bin_first[b] = (int32_t)total;
            for (uint32_t k = M.start[2 * b]; k < M.start[2 * b + 1]; ++k) {
                if (items) items[total] = M.items[k];
                ++total;
            }
        }
    if (bin_first && M.usable) bin_first[bins] = (int32_t)total;
    if (n_items) *n_items = total;
    return RT_OK;
}

int rt_create(int32_t device, void* stream, RtContext** out) {
    if (!out) return fail(RT_ERR_INVALID, "out is null");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(RT_ERR_INVALID, "device %d out of range (%d visible)", device, n);
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(raise_dynamic_lds_limit());
    double* scratch = nullptr;
    HIP_TRY(upload_scratch_of(device, &scratch));
    RtContext* ctx = new RtContext();
    ctx->device = device;
    ctx->upload_scratch = scratch;
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete ctx;
            return fail(RT_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
        }
        ctx->own_stream = true;
    }
    *out = ctx;
    return RT_OK;
}

void rt_destroy(RtContext* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->collect();
    for (auto e : ctx->free_events) (void)hipEventDestroy(e);
    ctx->geom.release(); ctx->mat.release(); ctx->fres.release(); ctx->cull.release(); ctx->hier_top.release(); ctx->hier_mid.release(); ctx->hier_leaf.release(); ctx->hier_top_ref.release(); ctx->hier_ids.release(); ctx->grid_start.release(); ctx->grid_items.release(); ctx->grid_loose.release(); ctx->grid_sph.release(); ctx->lmap_start.release(); ctx->lmap_items.release(); ctx->lmap_always.release(); ctx->sc.release();
    ctx->node_a.release(); ctx->node_b.release(); ctx->node_branch.release(); ctx->node_child.release();
    ctx->level_start.release(); ctx->node_n.release(); ctx->node_nglobal.release(); ctx->node_hits.release();
    ctx->node_beam.release(); ctx->node_tile0.release(); ctx->node_ntiles.release(); ctx->lvl_ntiles.release();
    ctx->lvl_tilebase.release(); ctx->flags.release(); ctx->node_start.release(); ctx->node_pre.release();
    ctx->tile_run.release(); ctx->blk_tot.release(); ctx->wave_off.release();
    ctx->px.release(); ctx->py.release(); ctx->pz.release(); ctx->dx.release(); ctx->dy.release(); ctx->dz.release();
    ctx->A.release(); ctx->dist.release(); ctx->cr.release(); ctx->cg.release(); ctx->cb.release(); ctx->sr.release(); ctx->sg.release(); ctx->sb.release();
    ctx->info.release(); ctx->link.release(); ctx->list_ray.release(); ctx->redo_list.release(); ctx->list_node.release(); ctx->lvl_shade_base.release();
    for (auto& b : ctx->list_j) b.release();
    ctx->X.release(); ctx->Y.release(); ctx->packed.release(); ctx->packed32.release(); ctx->staging.release();
    ctx->image.release(); ctx->image8.release(); ctx->mt_ring.release(); ctx->mt_raw.release(); ctx->mt_polys.release(); ctx->mt_end.release(); ctx->sample_off.release(); ctx->sample_global.release(); ctx->hit_mask.release(); ctx->hit_pref.release(); ctx->hit_word0.release(); ctx->emit_blocks.release(); ctx->work.release();
    ctx->mt_need.release(); ctx->mt_stat.release(); ctx->raw_slot.release();
    ctx->mt_ring0.release(); ctx->mt_end0.release(); ctx->pass_of.release();
    if (ctx->bounce) (void)hipHostFree(ctx->bounce);
    for (auto e : ctx->bounce_done) if (e) (void)hipEventDestroy(e);
    if (ctx->side_fork) (void)hipEventDestroy(ctx->side_fork);
    if (ctx->side_join) (void)hipEventDestroy(ctx->side_join);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int rt_set_scene(RtContext* ctx, const RtScene* scene, const RtParams* params) {
    if (int rc = use_device(ctx)) return rc;
    if (!scene || !params) return fail(RT_ERR_INVALID, "scene/params is null");
    if (scene->n_spheres < 0 || scene->n_spheres > RT_MAX_SPHERES)
        return fail(RT_ERR_INVALID, "n_spheres %d out of range", scene->n_spheres);
    if (scene->n_lights < 0 || scene->n_lights > RT_MAX_LIGHTS)
        return fail(RT_ERR_INVALID, "n_lights %d out of range (max %d)", scene->n_lights, RT_MAX_LIGHTS);
    if (params->width <= 0 || params->height <= 0) return fail(RT_ERR_INVALID, "image size must be positive");
    if (scene->n_spheres && (!scene->sphere_geom || !scene->sphere_mat)) return fail(RT_ERR_INVALID, "sphere arrays are null");
    if (scene->n_lights && !scene->lights) return fail(RT_ERR_INVALID, "lights is null");

    HostTree tree;
    if (int rc = build_tree(params->max_reflection_depth, params->max_refraction_depth, tree)) return rc;
    if (tree.n_levels() > MAX_LEVELS) return fail(RT_ERR_INVALID, "more than %d bounce levels", MAX_LEVELS);
    for (int l = 0; l < tree.n_levels(); ++l)
        if (tree.level_start[l + 1] - tree.level_start[l] > MAX_LEVEL_NODES)
            return fail(RT_ERR_INVALID, "ray tree level %d is wider than %d nodes", l, MAX_LEVEL_NODES);

    SceneConst sc{};
    bool sampled = false, disk = false;
    uint32_t spot_mask = 0;
    for (int l = 0; l < scene->n_lights; ++l) {
        const RtLight& L = scene->lights[l];
        if (L.kind < RT_LIGHT_POINT || L.kind > RT_LIGHT_SPOT) return fail(RT_ERR_INVALID, "light %d: unknown kind %d", l, L.kind);
        if (L.kind != RT_LIGHT_POINT) {
            if (L.n_points < 1) return fail(RT_ERR_INVALID, "light %d: n_points must be >= 1", l);
            sampled = true;
            if (L.kind == RT_LIGHT_SPOT) spot_mask |= 1u << l; else disk = true;
        }
        sc.lights[l] = L;
        if (L.kind != RT_LIGHT_POINT) {
            LightF32& F = sc.lightf[l];
            for (int k = 0; k < 3; ++k) {
                F.xh[k] = (float)(L.radius * L.x_hat[k]);
                F.yh[k] = (float)(L.radius * L.y_hat[k]);
                F.dcol[k] = (float)L.diffuse_color[k];
                F.scol[k] = (float)L.specular_color[k];
            }
            F.ps = (float)(L.intensity / L.n_points);
            F.ps_grey = (float)(L.intensity / L.n_points * L.diffuse_color[0]);
            // |radius (x x_hat + y y_hat)| over the unit disk is at most radius times the largest
            // singular value of the frame (x_hat, y_hat), whatever the frame
            const double a = L.x_hat[0] * L.x_hat[0] + L.x_hat[1] * L.x_hat[1] + L.x_hat[2] * L.x_hat[2];
            const double c = L.y_hat[0] * L.y_hat[0] + L.y_hat[1] * L.y_hat[1] + L.y_hat[2] * L.y_hat[2];
            const double b = L.x_hat[0] * L.y_hat[0] + L.x_hat[1] * L.y_hat[1] + L.x_hat[2] * L.y_hat[2];
            F.extent = (float)(fabs(L.radius) * sqrt(0.5 * (a + c + sqrt((a - c) * (a - c) + 4 * b * b))) * 1.000001);
        }
    }
    memcpy(sc.ambient, scene->ambient, 24);
    memcpy(sc.cam, scene->camera_position, 24);
    memcpy(sc.scr_pos, scene->screen_position, 24);
    memcpy(sc.scr_x, scene->screen_x, 24);
    memcpy(sc.scr_y, scene->screen_y, 24);
    sc.scr_w = scene->screen_width; sc.scr_h = scene->screen_height;
    sc.pix_x = scene->pixels_x; sc.pix_y = scene->pixels_y;
    sc.eps = params->epsilon; sc.air = params->air_refractive_index;
    memcpy(sc.bg, params->background, 24);
    sc.small_num = params->small_num; sc.width = params->width; sc.height = params->height;
    sc.n_lights = scene->n_lights;

    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const size_t S = scene->n_spheres, K = tree.a.size();
    HIP_TRY(ctx->geom.ensure(S * 4));
    HIP_TRY(ctx->mat.ensure(S * RT_MAT_WIDTH));
    HIP_TRY(ctx->fres.ensure(S * 4));
    HIP_TRY(ctx->sc.ensure(1));
    ctx->hier_top_groups = 0;
    if (S) {
        HIP_TRY(hipMemcpy(ctx->geom.p, scene->sphere_geom, S * 4 * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ctx->mat.p, scene->sphere_mat, S * RT_MAT_WIDTH * sizeof(double), hipMemcpyHostToDevice));
        // per-sphere refraction constants (reference raytrace.py:199-201, :237-239): the same
        // IEEE divisions the reference does per ray, done once
        std::vector<double> fres(S * 4);
        const double air = params->air_refractive_index;
        for (size_t i = 0; i < S; ++i) {
            const double ior = scene->sphere_mat[i * RT_MAT_WIDTH + 12];
            const double q = (air - ior) / (air + ior);
            fres[i * 4 + 0] = air / ior;
            fres[i * 4 + 1] = ior / air;
            fres[i * 4 + 2] = q * q;
            fres[i * 4 + 3] = 0;
        }
        HIP_TRY(hipMemcpy(ctx->fres.p, fres.data(), S * 4 * sizeof(double), hipMemcpyHostToDevice));
        // reference point of the float32 screening: centroid of the spheres that are not huge
        double sum[3] = {0, 0, 0};
        size_t cnt = 0;
        for (size_t i = 0; i < S; ++i) {
            const double* g = scene->sphere_geom + i * 4;
            if (fabs(g[3]) > 1e4) continue;
            sum[0] += g[0]; sum[1] += g[1]; sum[2] += g[2];
            ++cnt;
        }
        for (int k = 0; k < 3; ++k) ctx->cull_ref[k] = cnt ? sum[k] / (double)cnt : 0.0;
        // float32 screening table for the shadow tests of scenes with many spheres
        const size_t n_pairs = ((S + 7) / 8) * 4;
        std::vector<CullPair> cull(n_pairs);
        for (size_t i = 0; i < n_pairs * 2; ++i) {
            if (i < S) cull_entry(cull[i >> 1], (int)(i & 1), scene->sphere_geom + i * 4);
            else cull_blank(cull[i >> 1], (int)(i & 1));
        }
        HIP_TRY(ctx->cull.ensure(n_pairs * 10));
        HIP_TRY(hipMemcpy(ctx->cull.p, cull.data(), n_pairs * sizeof(CullPair), hipMemcpyHostToDevice));
        // Above that table (RT_HIERARCHY=0: neither, every sphere is screened for every ray): a uniform grid
        // over the spheres, or (RT_ACCEL=hier, or more than GRID_MAX_LOOSE spheres that do not fit a grid)
        // a bounding-sphere hierarchy.  RT_GRID_DENSITY: cells per sphere the grid aims at.
        ctx->hier_top_groups = 0;
        ctx->grid_cells = 0;
        ctx->lmap_mask = 0;
        const char* use_hier = getenv("RT_HIERARCHY");
        const char* accel = getenv("RT_ACCEL");
        const bool accelerate = S > SPH_CHUNK_SMALL && !(use_hier && use_hier[0] == '0');
        if (accelerate && !(accel && strcmp(accel, "hier") == 0)) {
            const char* dens = getenv("RT_GRID_DENSITY");
            const HostGrid G = build_grid(scene->sphere_geom, S, dens ? atof(dens) : 2.0);
            if (!G.start.empty() && G.loose.size() <= (size_t)GRID_MAX_LOOSE) {
                auto upload = [](auto& buf, const auto& v) -> hipError_t {
                    if (hipError_t e = buf.ensure(std::max<size_t>(2, v.size()))) return e;
                    return v.empty() ? hipSuccess : hipMemcpy(buf.p, v.data(), v.size() * sizeof(v[0]), hipMemcpyHostToDevice);
                };
                HIP_TRY(upload(ctx->grid_start, G.start));
                HIP_TRY(upload(ctx->grid_items, G.items));
                HIP_TRY(upload(ctx->grid_sph, G.sph));
                HIP_TRY(upload(ctx->grid_loose, G.loose));
                ctx->grid_cells = G.cells();
                ctx->grid_n_loose = (int)G.loose.size();
                ctx->grid_h = G.h; ctx->grid_inv_h = G.inv_h; ctx->grid_margin = G.margin;
                for (int k = 0; k < 3; ++k) { ctx->grid_n[k] = G.n[k]; ctx->grid_origin[k] = G.origin[k]; ctx->grid_hi[k] = G.hi[k]; }
                // point lights: their occluders by direction (RT_LIGHT_MAPS=0: shadow rays walk the grid)
                const char* use_maps = getenv("RT_LIGHT_MAPS");
                std::vector<char> in_grid(S, 1);
                for (uint16_t i : G.loose) in_grid[i] = 0;
                std::vector<uint32_t> all_start;
                std::vector<uint16_t> all_items, all_always;
                for (int l = 0; l < scene->n_lights && !(use_maps && use_maps[0] == '0'); ++l) {
                    if (scene->lights[l].kind != RT_LIGHT_POINT) continue;
                    const HostLightMap M = build_light_map(scene->sphere_geom, S, scene->lights[l].position, in_grid);
                    if (!M.usable) continue;
                    const uint32_t item_base = (uint32_t)all_items.size();
                    ctx->lmap_start_off[l] = (int)all_start.size();
                    ctx->lmap_always_off[l] = (int)all_always.size();
                    ctx->lmap_n_always[l] = (int)M.always.size();
                    for (uint32_t v : M.start) all_start.push_back(v + item_base);
                    all_items.insert(all_items.end(), M.items.begin(), M.items.end());
                    all_always.insert(all_always.end(), M.always.begin(), M.always.end());
                    ctx->lmap_mask |= 1u << l;
                }
                HIP_TRY(upload(ctx->lmap_start, all_start));
                HIP_TRY(upload(ctx->lmap_items, all_items));
                HIP_TRY(upload(ctx->lmap_always, all_always));
            }
        }
        if (accelerate && ctx->grid_cells == 0) {
            const HostHierarchy H = build_hierarchy(scene->sphere_geom, S);
            auto upload = [](auto& buf, const auto& v) -> hipError_t {
                if (hipError_t e = buf.ensure(std::max<size_t>(1, v.size()))) return e;
                return v.empty() ? hipSuccess : hipMemcpy(buf.p, v.data(), v.size() * sizeof(v[0]), hipMemcpyHostToDevice);
            };
            HIP_TRY(upload(ctx->hier_top, H.top));
            HIP_TRY(upload(ctx->hier_mid, H.mid));
            HIP_TRY(upload(ctx->hier_leaf, H.leaf));
            HIP_TRY(upload(ctx->hier_top_ref, H.top_ref));
            HIP_TRY(upload(ctx->hier_ids, H.ids));
            ctx->hier_top_groups = (int)H.top.size();
            ctx->hier_blocks = H.blocks;
            // the windows of k_trace's search: 32 typical radii first (measured best on the 4096-sphere scene; RT_HIER_WINDOW overrides); twice the distance from the
            // tables' reference point to the farthest grouped sphere covers every ray's way through them
            double reach = 0;
            for (int b = 0; b < H.blocks; ++b) {
                const double* q = H.block_bound.data() + (size_t)b * 4;
                const double dx = q[0] - ctx->cull_ref[0], dy = q[1] - ctx->cull_ref[1], dz = q[2] - ctx->cull_ref[2];
                reach = std::max(reach, sqrt(dx * dx + dy * dy + dz * dz) + q[3]);
            }
            ctx->hier_extent = (float)(2.5 * reach);
            const char* wenv = getenv("RT_HIER_WINDOW");
            ctx->hier_window = (float)std::max((wenv ? atof(wenv) : 32.0) * H.typical_radius, 2.5 * reach / 256);
        }
    }
    HIP_TRY(hipMemcpy(ctx->sc.p, &sc, sizeof sc, hipMemcpyHostToDevice));
    HIP_TRY(ctx->node_a.ensure(K)); HIP_TRY(ctx->node_b.ensure(K)); HIP_TRY(ctx->node_branch.ensure(K));
    HIP_TRY(ctx->node_child.ensure(K * 3)); HIP_TRY(ctx->level_start.ensure(tree.level_start.size()));
    HIP_TRY(hipMemcpy(ctx->node_a.p, tree.a.data(), K * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->node_b.p, tree.b.data(), K * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->node_branch.p, tree.branch.data(), K * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->node_child.p, tree.child.data(), K * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->level_start.p, tree.level_start.data(), tree.level_start.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(ctx->node_n.ensure(K)); HIP_TRY(ctx->node_nglobal.ensure(K)); HIP_TRY(ctx->node_hits.ensure(K));
    HIP_TRY(ctx->node_beam.ensure(K * RT_MAX_LIGHTS)); HIP_TRY(ctx->node_tile0.ensure(K)); HIP_TRY(ctx->node_ntiles.ensure(K));
    HIP_TRY(ctx->node_start.ensure(K)); HIP_TRY(ctx->node_pre.ensure(K * NCNT));
    HIP_TRY(ctx->lvl_ntiles.ensure(tree.n_levels())); HIP_TRY(ctx->lvl_tilebase.ensure(tree.n_levels()));
    HIP_TRY(ctx->lvl_shade_base.ensure(tree.n_levels() + 1));
    HIP_TRY(ctx->blk_tot.ensure((size_t)tree.n_levels() * NB * NCNT));
    HIP_TRY(ctx->wave_off.ensure((size_t)tree.n_levels() * NW * NCNT));
    HIP_TRY(ctx->flags.ensure(4));
    HIP_TRY(ctx->work.ensure(8));
    HIP_TRY(ctx->sample_off.ensure(K * RT_MAX_LIGHTS));

    bool simple = true;
    for (size_t i = 0; i < S && simple; ++i)
        for (int c = 3; c < 9; ++c) {
            const double k = scene->sphere_mat[i * RT_MAT_WIDTH + c];
            if (!(k >= 0.0 && k <= 1.0)) simple = false;
        }
    for (int l = 0; l < scene->n_lights && simple; ++l) {
        const RtLight& L = scene->lights[l];
        if (L.kind == RT_LIGHT_POINT) continue;
        if (memcmp(L.diffuse_color, L.specular_color, 24) != 0) simple = false;
        if (!(L.diffuse_color[0] == L.diffuse_color[1] && L.diffuse_color[0] == L.diffuse_color[2] && L.diffuse_color[0] >= 0)) simple = false;
        if (!(L.intensity >= 0)) simple = false;
    }
    ctx->simple_shading = simple;

    ctx->tree = std::move(tree);
    ctx->sc_host = sc;
    ctx->params = *params;
    ctx->S = (int)S;
    ctx->sampled_lights = sampled;
    ctx->disk_lights = disk;
    ctx->spot_mask = spot_mask;
    for (auto& b : ctx->list_j) b.release();   // sized per frame for the lights of this scene
    ctx->user_batches = nullptr;
    ctx->batch_guess = nullptr;
    ctx->batch_slot = 0; ctx->batch_slots = 1;
    ctx->have_scene = true;
    ctx->frame_begun = ctx->geometry_done = ctx->layout_set = ctx->shaded = false;
    return RT_OK;
}

int rt_begin_frame(RtContext* ctx, int32_t row0, int32_t n_rows, const double* X, const double* Y, int64_t n,
                   int64_t arena_rays) {
    if (int rc = use_device(ctx)) return rc;
    if (!ctx->have_scene) return fail(RT_ERR_STATE, "rt_set_scene has not been called");
    int64_t npix;
    if (X || Y) {
        if (!X || !Y || n <= 0) return fail(RT_ERR_INVALID, "X and Y must both be given with n > 0");
        npix = n;
    } else {
        if (row0 < 0 || n_rows <= 0 || row0 + n_rows > ctx->params.height)
            return fail(RT_ERR_INVALID, "rows [%d, %d) outside the image of height %d", row0, row0 + n_rows, ctx->params.height);
        npix = (int64_t)n_rows * ctx->params.width;
    }
    if (npix > 0x7FFFFFFF) return fail(RT_ERR_INVALID, "too many pixels");
    if (arena_rays <= 0) arena_rays = std::max<int64_t>(ctx->arena_rays, npix * 24);
    const int64_t min_rays = ((npix + TILE - 1) / TILE) * TILE;
    arena_rays = std::max(arena_rays, min_rays);
    arena_rays = ((arena_rays + TILE - 1) / TILE) * TILE;
    if (arena_rays >= (int64_t)0xFFFFFFFFu - 2 * TILE) return fail(RT_ERR_INVALID, "ray arena too large for 32-bit ray indices");

    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const size_t R = (size_t)arena_rays;
    DevBuf<double>* dbl[] = {&ctx->px, &ctx->py, &ctx->pz, &ctx->dx, &ctx->dy, &ctx->dz, &ctx->A, &ctx->dist, &ctx->cr, &ctx->cg, &ctx->cb, &ctx->sr, &ctx->sg, &ctx->sb};
    for (auto* buf : dbl) HIP_TRY(buf->ensure(R));
    HIP_TRY(ctx->info.ensure(R));
    HIP_TRY(ctx->link.ensure(R));
    HIP_TRY(ctx->list_ray.ensure(R));
    HIP_TRY(ctx->list_node.ensure(R));
    if (ctx->sampled_lights) HIP_TRY(ctx->redo_list.ensure(R));
    if (ctx->disk_lights) HIP_TRY(ctx->list_j[0].ensure(R));
    for (int l = 0; l < RT_MAX_LIGHTS; ++l)
        if ((ctx->spot_mask >> l) & 1u) HIP_TRY(ctx->list_j[1 + l].ensure(R));
    HIP_TRY(ctx->tile_run.ensure(R / TILE * NCNT));
    HIP_TRY(ctx->image.ensure((size_t)npix * 3));
    ctx->explicit_xy = X != nullptr;
    if (X) {
        HIP_TRY(ctx->X.ensure(npix));
        HIP_TRY(ctx->Y.ensure(npix));
        HIP_TRY(hipMemcpy(ctx->X.p, X, npix * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ctx->Y.p, Y, npix * sizeof(double), hipMemcpyHostToDevice));
    }
    ctx->user_image = nullptr;
    ctx->user_image8 = nullptr;
    // Workgroups per level launch: half as many for frames of up to five megapixels (measured: 1080p
    // frames 5.18 -> 5.00 ms, a 3840x540 strip 5.60 -> 5.39 ms with 1024 workgroups - k_scan walks fewer
    // totals, fewer workgroups find nothing to do; 4K frames are level: k_link 2.04 -> 2.40 ms, k_scan 0.69 -> 0.46).
    // RT_LEVEL_WORKGROUPS overrides (a multiple of 256).
    ctx->nb = npix <= 5000000 ? std::min(NB, 1024) : NB;
    if (const char* e = getenv("RT_LEVEL_WORKGROUPS")) ctx->nb = std::max(256, std::min(NB, atoi(e) / 256 * 256));
    ctx->npix = npix;
    ctx->row0 = X ? 0 : row0;
    ctx->arena_rays = arena_rays;
    ctx->frame_begun = true;
    ctx->geometry_done = ctx->layout_set = ctx->shaded = false;
    ctx->next_level = 0;
    return RT_OK;
}

int rt_trace_level(RtContext* ctx, int32_t level) {
    if (int rc = use_device(ctx)) return rc;
    if (!ctx->frame_begun) return fail(RT_ERR_STATE, "rt_begin_frame has not been called");
    if (level != ctx->next_level && level != 0)
        return fail(RT_ERR_STATE, "levels must be traced in order (expected %d or 0, got %d)", ctx->next_level, level);
    if (level < 0 || level >= ctx->tree.n_levels()) return fail(RT_ERR_INVALID, "level %d out of range", level);
    if (level == 0) {
        // (re)start the frame on the pixels chosen by the last rt_begin_frame
        ctx->geometry_done = false;
        ctx->node_hits_host_valid = false;
        ctx->frame_col32 = ctx->precision != RT_PRECISION_F64;   // k_trace, the lighting kernels and k_combine must agree
        ctx->begin(RT_K_INIT);
        hipLaunchKernelGGL(k_init, dim3(1), dim3(1024), 0, ctx->stream, ctx->dev());
        ctx->end();
    }
    if (int rc = launch_level(ctx, level)) return rc;
    ctx->next_level = level + 1;
    if (ctx->next_level == ctx->tree.n_levels()) ctx->geometry_done = true;
    ctx->shaded = false;
    return RT_OK;
}

int rt_trace_geometry(RtContext* ctx) {
    if (int rc = use_device(ctx)) return rc;
    if (!ctx->frame_begun) return fail(RT_ERR_STATE, "rt_begin_frame has not been called");
    ctx->next_level = 0;
    for (int l = 0; l < ctx->tree.n_levels(); ++l)
        if (int rc = rt_trace_level(ctx, l)) return rc;
    return RT_OK;
}

int rt_get_level_batches(RtContext* ctx, int32_t level, int64_t* n_local) {
    if (int rc = use_device(ctx)) return rc;
    if (!ctx->frame_begun || level < 0 || level >= ctx->tree.n_levels() || level > ctx->next_level)
        return fail(RT_ERR_STATE, "level %d has not been laid out yet", level);
    if (!n_local) return fail(RT_ERR_INVALID, "n_local is null");
    const int k0 = ctx->tree.level_start[level], nn = ctx->tree.level_start[level + 1] - k0;
    std::vector<int32_t> tmp(nn);
    HIP_TRY(hipMemcpyAsync(tmp.data(), ctx->node_n.p + k0, nn * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < nn; ++i) n_local[i] = tmp[i];
    return RT_OK;
}

int rt_set_global_batches(RtContext* ctx, int32_t level, const int64_t* n_global) {
    if (int rc = use_device(ctx)) return rc;
    if (!ctx->frame_begun || level < 0 || level >= ctx->tree.n_levels() || level != ctx->next_level)
        return fail(RT_ERR_STATE, "global batches of level %d must be set right before tracing it", level);
    if (!n_global) return fail(RT_ERR_INVALID, "n_global is null");
    const int k0 = ctx->tree.level_start[level], nn = ctx->tree.level_start[level + 1] - k0;
    std::vector<int32_t> tmp(nn);
    for (int i = 0; i < nn; ++i) {
        if (n_global[i] < 0 || n_global[i] > 0x7FFFFFFF) return fail(RT_ERR_INVALID, "batch size out of range");
        tmp[i] = (int32_t)n_global[i];
    }
    HIP_TRY(hipMemcpyAsync(ctx->level_batches(level), tmp.data(), nn * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_get_node_counts(RtContext* ctx, int64_t* batch, int64_t* hits, int64_t* beam) {
    if (int rc = use_device(ctx)) return rc;
    if (!ctx->geometry_done) return fail(RT_ERR_STATE, "geometry phase not finished");
    const int K = ctx->tree.n_nodes();
    std::vector<int32_t> n(K), h(K), bm((size_t)K * RT_MAX_LIGHTS);
    int32_t flag = 0;
    HIP_TRY(hipMemcpyAsync(n.data(), ctx->node_n.p, K * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(h.data(), ctx->node_hits.p, K * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(bm.data(), ctx->node_beam.p, (size_t)K * RT_MAX_LIGHTS * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&flag, ctx->flags.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (flag) return fail(RT_ERR_OVERFLOW, "ray arena of %lld rays is too small for this frame", (long long)ctx->arena_rays);
    ctx->node_hits_host = h;   // rt_generate_samples needs them again
    ctx->node_beam_host = bm;
    ctx->node_hits_host_valid = true;
    for (int k = 0; k < K; ++k) {
        if (batch) batch[k] = n[k];
        if (hits) hits[k] = h[k];
    }
    if (beam) for (size_t i = 0; i < bm.size(); ++i) beam[i] = bm[i];
    return RT_OK;
}

int rt_set_sample_layout(RtContext* ctx, const int64_t* offsets, int64_t total) {
    if (int rc = use_device(ctx)) return rc;
    if (!ctx->geometry_done) return fail(RT_ERR_STATE, "geometry phase not finished");
    if (!offsets || total < 0) return fail(RT_ERR_INVALID, "bad sample layout");
    const size_t n = (size_t)ctx->tree.n_nodes() * RT_MAX_LIGHTS;
    // the packed buffer has one slot per (sampled light, sample, shading-list entry)
    int32_t entries = 0, flag = 0;
    HIP_TRY(hipMemcpyAsync(&entries, ctx->lvl_shade_base.p + ctx->tree.n_levels(), 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&flag, ctx->flags.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (flag) return fail(RT_ERR_OVERFLOW, "ray arena of %lld rays is too small for this frame", (long long)ctx->arena_rays);
    int64_t slots = 0;
    for (int l = 0; l < ctx->sc_host.n_lights; ++l) {
        ctx->pack_base[l] = slots;
        if (ctx->sc_host.lights[l].kind != RT_LIGHT_POINT) slots += (int64_t)ctx->sc_host.lights[l].n_points * entries;
    }
    HIP_TRY(ctx->packed.ensure((size_t)slots * 2));
    ctx->pack_plane = slots;
    // scenes the float32 sample pass can take also keep the samples' disk points as floats (Dev::packed32)
    ctx->have_packed32 = ctx->S <= MIXED_MAX_SPHERES;
    if (ctx->have_packed32) {
        const int64_t tiles = ((int64_t)entries + TILE - 1) / TILE;
        int64_t quads = 0;
        for (int l = 0; l < ctx->sc_host.n_lights; ++l) {
            ctx->pack32_base[l] = quads;
            if (ctx->sc_host.lights[l].kind != RT_LIGHT_POINT) quads += tiles * ((ctx->sc_host.lights[l].n_points + 1) / 2) * TILE;
        }
        HIP_TRY(ctx->packed32.ensure((size_t)quads * 4));
    }
    ctx->disk_points_ready = false;
    ctx->samples_in_raw = false;
    HIP_TRY(ctx->raw_slot.ensure(1));
    HIP_TRY(hipMemsetAsync(ctx->raw_slot.p, 0, sizeof(const uint32_t*), ctx->stream));
    HIP_TRY(hipMemcpy(ctx->sample_off.p, offsets, n * sizeof(int64_t), hipMemcpyHostToDevice));
    ctx->sample_off_host.assign(offsets, offsets + n);
    ctx->pack_E = entries;
    ctx->sample_total = total;
    ctx->samples_delivered = 0;
    ctx->layout_set = true;
    ctx->have_global_layout = false;
    return RT_OK;
}

int rt_set_sample_layout_global(RtContext* ctx, const int64_t* offsets, const int64_t* columns, const int64_t* first_column,
                                int64_t total) {
    if (int rc = use_device(ctx)) return rc;
    if (!ctx->layout_set) return fail(RT_ERR_STATE, "rt_set_sample_layout has not been called");
    if (!offsets || !columns || !first_column || total < 0) return fail(RT_ERR_INVALID, "bad whole-image sample layout");
    const size_t n = (size_t)ctx->tree.n_nodes() * RT_MAX_LIGHTS;
    HIP_TRY(ctx->sample_global.ensure(3 * n));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(ctx->sample_global.p, offsets, n * sizeof(int64_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->sample_global.p + n, columns, n * sizeof(int64_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->sample_global.p + 2 * n, first_column, n * sizeof(int64_t), hipMemcpyHostToDevice));
    ctx->sample_global_host.assign(offsets, offsets + n);
    ctx->sample_global_host.insert(ctx->sample_global_host.end(), columns, columns + n);
    ctx->sample_global_host.insert(ctx->sample_global_host.end(), first_column, first_column + n);
    ctx->have_global_layout = true;
    ctx->global_total = total;
    return RT_OK;
}

// Generated samples that were only packed as disk points (rt_generate_samples): their uniforms into
// `packed` after all, from the raw words, for whoever reads that array or is about to change or release
// either.  Allocates nothing.  Not under stream capture (the flag would change without the kernel running).
static int materialize_packed(RtContext* ctx) {
    if (!ctx->samples_in_raw) return RT_OK;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone)
        return fail(RT_ERR_STATE, "the disk-light samples have to be unpacked for this call, which cannot be captured into a graph");
    if (ctx->pack_E > 0) {
        const long long total = ctx->have_global_layout ? ctx->global_total : ctx->sample_total;
        const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((ctx->pack_E + BLOCK - 1) / BLOCK, 4 * NB));
        hipLaunchKernelGGL((k_pack<GeneratedStream>), dim3((unsigned)blocks), dim3(BLOCK), 0, ctx->stream, ctx->dev(),
                           GeneratedStream{ctx->mt_raw.p + ctx->raw_value0}, (int64_t)0, (int64_t)total, 0, nullptr, 0);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemsetAsync(ctx->raw_slot.p, 0, sizeof(const uint32_t*), ctx->stream));
    ctx->samples_in_raw = false;
    return RT_OK;
}

int rt_upload_samples(RtContext* ctx, int64_t dst_offset, const double* u, int64_t n) {
    if (int rc = use_device(ctx)) return rc;
    if (!ctx->layout_set) return fail(RT_ERR_STATE, "rt_set_sample_layout has not been called");
    if (n < 0 || dst_offset < 0 || dst_offset + n > ctx->sample_total) return fail(RT_ERR_INVALID, "sample range out of bounds");
    if (n == 0) return RT_OK;
    if (!u) return fail(RT_ERR_INVALID, "u is null");
    if (int rc = materialize_packed(ctx)) return rc;
    // stage the piece of the stream, then scatter it into the packed layout
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx->staging.ensure((size_t)n));
    HIP_TRY(hipMemcpyAsync(ctx->staging.p, u, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (ctx->pack_E > 0) {
        const int64_t blocks = std::min<int64_t>((ctx->pack_E + BLOCK - 1) / BLOCK, 4 * NB);
        hipLaunchKernelGGL((k_pack<UploadedStream>), dim3((unsigned)blocks), dim3(BLOCK), 0, ctx->stream, ctx->dev(),
                           UploadedStream{ctx->staging.p}, dst_offset, n, 0, nullptr, 0);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));   // u may be reused by the caller
    ctx->samples_delivered += n;
    ctx->disk_points_ready = false;
    return RT_OK;
}

int rt_upload_samples_noalloc(RtContext* ctx, int64_t dst_offset, const double* u, int64_t n) {
    if (int rc = use_device(ctx)) return rc;
    if (!ctx->layout_set) return fail(RT_ERR_STATE, "rt_set_sample_layout has not been called");
    if (n < 0 || dst_offset < 0 || dst_offset + n > ctx->sample_total) return fail(RT_ERR_INVALID, "sample range out of bounds");
    if (n > (int64_t)RT_NOALLOC_MAX_SAMPLES)
        return fail(RT_ERR_INVALID, "rt_upload_samples_noalloc takes at most %lld values per call (got %lld): split the stream",
                    (long long)RT_NOALLOC_MAX_SAMPLES, (long long)n);
    if (n == 0) return RT_OK;
    if (!u) return fail(RT_ERR_INVALID, "u is null");
    if (int rc = materialize_packed(ctx)) return rc;
    // The staging array is shared by every call on this GPU and still holds the previous piece:
    // k_pack reads chunk[0 .. n) only, so nothing has to be cleared, but this context's previous
    // scatter must have finished reading before the array is overwritten.
    double* const staging = ctx->upload_scratch;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpyAsync(staging, u, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (ctx->pack_E > 0) {
        const int64_t blocks = std::min<int64_t>((ctx->pack_E + BLOCK - 1) / BLOCK, 4 * NB);
        hipLaunchKernelGGL((k_pack<UploadedStream>), dim3((unsigned)blocks), dim3(BLOCK), 0, ctx->stream, ctx->dev(),
                           UploadedStream{staging}, dst_offset, n, 0, nullptr, 0);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));   // u may be reused by the caller, the array by the next call
    ctx->samples_delivered += n;
    ctx->disk_points_ready = false;
    return RT_OK;
}

// The generator's raw words stay with the context between calls up to this size (RT_MT_RAW_KEEP_GB, default 64;
// 0: never; rt_trim gives them back at any time).
size_t mt_raw_keep_bytes() {
    static const size_t limit = [] {
        const char* e = getenv("RT_MT_RAW_KEEP_GB");
        return (size_t)(e ? std::max(0.0, atof(e)) : 64.0) << 30;
    }();
    return limit;
}
int rt_generate_samples(RtContext* ctx, const uint32_t* key, int32_t pos, uint32_t* key_out, int32_t* pos_out) {
    if (int rc = use_device(ctx)) return rc;
    if (!ctx->layout_set) return fail(RT_ERR_STATE, "rt_set_sample_layout has not been called");
    if (!key || !key_out || !pos_out) return fail(RT_ERR_INVALID, "null argument");
    if (pos < 0 || pos > MT_N) return fail(RT_ERR_INVALID, "MT19937 position %d out of range", pos);
    // doubles to draw (two 32-bit outputs each): the whole image's stream
    const long long total = ctx->have_global_layout ? ctx->global_total : ctx->sample_total;
    ctx->samples_delivered = 0;
    ctx->disk_points_ready = false;
    ctx->samples_in_raw = false;
    HIP_TRY(hipMemsetAsync(ctx->raw_slot.p, 0, sizeof(const uint32_t*), ctx->stream));
    if (total == 0) {
        ctx->samples_delivered = ctx->sample_total;
        memcpy(key_out, key, MT_N * sizeof(uint32_t));
        *pos_out = pos;
        return RT_OK;
    }
    // Absolute word indices: x[0..624) is the caller's key block, the first output consumed is
    // x[pos], the first one not consumed x[next].
    const long long next = (long long)pos + 2 * total;
    // streams of L = 2^ell words, at most 2048 of them: eight workgroups of k_mt_generate to a compute unit
    // (4K example_4, 12.5e9 words: 1490 streams 9.1 ms + 4.1 ms of jumps; 745: 11.7 + 3.3; 2980: 8.1 + 5.7)
    int ell = 16;
    long long max_streams = 2048;
    if (const char* e = getenv("RT_MT_MAX_STREAMS")) max_streams = std::max(1ll, atoll(e));
    while (((next + (1ll << ell) - 1) >> ell) > max_streams) ++ell;
    // A stream whose raw words go through the buffer in several passes (below) is cut finer, so that every
    // pass still has about max_streams workgroups: a pass of a few hundred leaves most of the GPU idle
    // (8K example_4, 50e9 words in five passes: 298 streams per pass 13.9 ms each, against 6.7 ms for 1490)
    {
        long long room = (long long)ctx->mt_raw.n;
        if (room < next + 2) {
            size_t free_bytes = 0, total_bytes = 0;
            HIP_TRY(hipMemGetInfo(&free_bytes, &total_bytes));
            room = std::max(room, std::max<long long>(1ll << 28, (long long)(free_bytes / 2 / sizeof(uint32_t))));
        }
        if (const char* e = getenv("RT_MT_RAW_WORDS")) room = std::max(1ll, atoll(e));
        const long long passes = std::min<long long>((next + room - 1) / room, 16);
        while (ell > 16 && ((next + (1ll << (ell - 1)) - 1) >> (ell - 1)) <= max_streams * passes) --ell;
    }
    const long long L = 1ll << ell;
    const int T = (int)((next + L - 1) >> ell);
    // words held in memory at a time (whole streams): all of them if a good part of the free device
    // memory holds them (then every stream is generated in one launch, four workgroups to a
    // compute unit); RT_MT_RAW_WORDS overrides for tests
    long long budget = (long long)ctx->mt_raw.n;            // what an earlier call kept is there already
    if (budget < (long long)T * L + 2) {
        size_t free_bytes = 0, total_bytes = 0;
        HIP_TRY(hipMemGetInfo(&free_bytes, &total_bytes));
        budget = std::max(budget, std::max<long long>(1ll << 28, (long long)(free_bytes / 2 / sizeof(uint32_t))));
    }
    if (const char* e = getenv("RT_MT_RAW_WORDS")) budget = std::max(1ll, atoll(e));
    const int per_pass = (int)std::max<long long>(1, std::min<long long>(T, budget >> ell));

    HIP_TRY(ctx->mt_ring.ensure((size_t)T * MT_RING));
    HIP_TRY(ctx->mt_end.ensure((size_t)T));
    std::vector<uint32_t> ring(MT_RING, 0u), first_ring(MT_RING, 0u);
    memcpy(first_ring.data(), key, MT_N * sizeof(uint32_t));
    const long long first_end = MT_N;
    HIP_TRY(hipMemcpyAsync(ctx->mt_ring.p, first_ring.data(), MT_RING * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->mt_end.p, &first_end, sizeof first_end, hipMemcpyHostToDevice, ctx->stream));

    bool jumping = false;
    if (T > 1) {
        // start windows of streams 1 .. T-1 by jump-ahead from the key block
        int n_polys = 1;                                   // bits of T - 1, the largest t
        while (((T - 1) >> n_polys) > 0) ++n_polys;
        // every polynomial as the list of its set coefficients, round by round (JumpPoly in rt_samples.hip);
        // what the last call uploaded is still there if it cut its stream the same way
        const size_t head_words = (n_polys * sizeof(JumpPoly) / 4 + 3) / 4 * 4;
        if (ctx->mt_polys_ell != ell || ctx->mt_polys_n != n_polys || !ctx->mt_polys.p) {
            std::vector<uint32_t> words(MT_N);
            std::vector<JumpPoly> heads(n_polys);
            std::vector<uint32_t> lists;
            for (int b = 0; b < n_polys; ++b) {
                if (!mtjump::pow2_words(ell + b, words.data()))
                    return fail(RT_ERR_INVALID, "MT19937 jump tables unavailable (stream too long?)");
                for (int r = 0; r < MTJ_ROUNDS; ++r) {
                    heads[b].first[r] = (int32_t)lists.size();
                    for (int i = r * MTJ_CHUNK; i < std::min((r + 1) * MTJ_CHUNK, 19937); ++i)
                        if ((words[i >> 5] >> (i & 31)) & 1u) lists.push_back(mtj_offset(i, r));
                    while (lists.size() % 16) lists.push_back(MTJ_PAD);
                }
                heads[b].first[MTJ_ROUNDS] = (int32_t)lists.size();
            }
            // (one buffer: the heads, then the lists, 16-byte aligned)
            std::vector<uint32_t> polys(head_words + lists.size() + 16, 0u);
            memcpy(polys.data(), heads.data(), n_polys * sizeof(JumpPoly));
            memcpy(polys.data() + head_words, lists.data(), lists.size() * 4);
            ctx->mt_polys_ell = -1;
            HIP_TRY(ctx->mt_polys.ensure(polys.size()));
            HIP_TRY(hipMemcpy(ctx->mt_polys.p, polys.data(), polys.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            ctx->mt_polys_ell = ell;
            ctx->mt_polys_n = n_polys;
        }
        const uint32_t* device_lists = ctx->mt_polys.p + head_words;
        // level b: the windows of streams 2^b .. min(2^(b+1), T) - 1 from those of streams 0 .. (one polynomial each),
        // summed into zeroed rings; on a stream of their own, beside the hit mask and the store map
        if (!ctx->side_stream) {
            HIP_TRY(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&ctx->side_join, hipEventDisableTiming));
        }
        HIP_TRY(hipEventRecord(ctx->side_fork, ctx->stream));
        HIP_TRY(hipStreamWaitEvent(ctx->side_stream, ctx->side_fork, 0));
        HIP_TRY(hipMemsetAsync(ctx->mt_ring.p + MT_RING, 0, (size_t)(T - 1) * MT_RING * sizeof(uint32_t), ctx->side_stream));
        for (int b = 0; b < n_polys; ++b) {
            const int count = std::min(1 << b, T - (1 << b));
            if (count > 0)
                hipLaunchKernelGGL(k_mt_jump, dim3(count * MTJ_ROUNDS), dim3(640), 0, ctx->side_stream, ctx->mt_ring.p, ctx->mt_end.p,
                                   reinterpret_cast<const JumpPoly*>(ctx->mt_polys.p), device_lists, b, ell);
        }
        HIP_TRY(hipEventRecord(ctx->side_join, ctx->side_stream));
        jumping = true;
    }
    // RT_MT_FUSED=1, for scenes whose sampled lights are all plain disk lights: the generator stores every
    // value straight at its place in the packed layout (k_mt_emit) and no raw words reach device memory -
    // 4K example_4: 17 GB written instead of 50 + 19 + 17 GB moved and no 50 GB work buffer, but 33 ms
    // instead of 12.6 + 8.2 ms of kernels (bound by the vector instructions of the placement, not by
    // memory).  Default: raw words first, k_pack afterwards.
    bool fused = false;
    if (const char* e = getenv("RT_MT_FUSED")) fused = e[0] == '1' && ctx->spot_mask == 0 && ctx->disk_lights && ctx->pack_E > 0;
    // Default path: raw words nobody will read are not stored (k_store_map; RT_MT_STORE_ALL=1 stores all).
    bool mapped = !fused && ctx->pack_E > 0 && !getenv("RT_MT_STORE_ALL");
    int n_emit_blocks = 0;
    // (sources of uploads that are queued on the stream: alive until the call's last synchronisation)
    std::vector<uint32_t> word0;
    std::vector<EmitBlock> blocks;
    if (fused || mapped) {
        const int K = ctx->tree.n_nodes(), Lv = ctx->tree.n_levels();
        std::vector<int32_t> hits(K), beam((size_t)K * RT_MAX_LIGHTS, 0), shade_base(Lv + 1, 0);
        std::vector<uint32_t> pre((size_t)K * NCNT, 0u);
        if (mapped && ctx->node_hits_host_valid && (int)ctx->node_hits_host.size() == K && ctx->node_beam_host.size() == beam.size()) {
            hits = ctx->node_hits_host;          // the map only needs the counts per node, and the host has read them
            beam = ctx->node_beam_host;
        } else {
            HIP_TRY(hipMemcpyAsync(hits.data(), ctx->node_hits.p, (size_t)K * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipMemcpyAsync(beam.data(), ctx->node_beam.p, beam.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipMemcpyAsync(shade_base.data(), ctx->lvl_shade_base.p, (size_t)(Lv + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipMemcpyAsync(pre.data(), ctx->node_pre.p, pre.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
        // The mask has one row of bits per way of numbering hits: row 0 all hits of a node (the columns of a disk
        // light's blocks), row 1 + l the hits inside spotlight l's beam (the columns of its blocks); node k's bits of
        // a row start at 64-bit word word0[row * (K + 1) + k].
        const int n_lights = ctx->sc_host.n_lights;
        const size_t row_len = (size_t)K + 1;
        word0.assign((1 + RT_MAX_LIGHTS) * row_len, 0u);
        uint32_t mask_words = 0;
        bool row_used[1 + RT_MAX_LIGHTS] = {ctx->disk_lights};
        for (int l = 0; l < n_lights; ++l) row_used[1 + l] = mapped && ((ctx->spot_mask >> l) & 1u);
        for (int row = 0; row <= RT_MAX_LIGHTS; ++row) {
            for (int k = 0; k < K; ++k) {
                word0[row * row_len + k] = mask_words;
                const int32_t count = row == 0 ? hits[k] : beam[(size_t)k * RT_MAX_LIGHTS + row - 1];
                if (row_used[row]) mask_words += (uint32_t)((count + 63) / 64);
            }
            word0[row * row_len + K] = mask_words;
        }
        const bool whole = ctx->have_global_layout;
        const size_t KL = (size_t)K * RT_MAX_LIGHTS;
        for (int k = 0; k < K; ++k)
            for (int l = 0; l < n_lights; ++l) {
                const int kind = ctx->sc_host.lights[l].kind;
                if (kind != RT_LIGHT_DISK && !(kind == RT_LIGHT_SPOT && mapped)) continue;
                const size_t at = (size_t)k * RT_MAX_LIGHTS + l;
                const int row = kind == RT_LIGHT_SPOT ? 1 + l : 0;
                const int64_t off = whole ? ctx->sample_global_host[at] : ctx->sample_off_host[at];
                const int32_t mine = row == 0 ? hits[k] : beam[at];
                const int64_t h = whole ? ctx->sample_global_host[KL + at] : mine;
                if (off < 0 || h <= 0 || (fused && mine <= 0)) continue;
                // shaded hits of the node: the level-wide running count at the next node's first tile minus its own
                const int lv = ctx->tree.level[k];
                const uint32_t upto = k + 1 < ctx->tree.level_start[lv + 1] ? pre[(size_t)(k + 1) * NCNT + C_SHADE]
                                                                             : (uint32_t)(shade_base[lv + 1] - shade_base[lv]);
                if (fused && upto == pre[(size_t)k * NCNT + C_SHADE]) continue;   // none: nobody reads this node's samples
                EmitBlock b{};
                b.off = off;
                b.rows = (uint32_t)ctx->sc_host.lights[l].n_points;
                b.end = off + 2 * (int64_t)b.rows * h;
                b.h = (uint32_t)h;
                b.inv_h = 1.0 / (double)h;
                b.first_col = whole ? (uint32_t)ctx->sample_global_host[2 * KL + at] : 0u;
                b.mine = (uint32_t)mine;
                b.word0 = word0[row * row_len + k];
                b.e_first = (uint32_t)shade_base[ctx->tree.level[k]] + pre[(size_t)k * NCNT + C_SHADE];   // (the fused generator's)
                b.light = (uint32_t)l;
                b.node = (uint32_t)k;
                blocks.push_back(b);
            }
        std::sort(blocks.begin(), blocks.end(), [](const EmitBlock& a, const EmitBlock& b) { return a.off < b.off; });
        n_emit_blocks = (int)blocks.size();
        fused = fused && n_emit_blocks > 0;
        mapped = mapped && n_emit_blocks > 0;
        if (fused || mapped) {
            ctx->hit_mask.transient = ctx->hit_pref.transient = true;
            HIP_TRY(ctx->hit_word0.ensure(word0.size()));
            HIP_TRY(ctx->emit_blocks.ensure(blocks.size()));
            HIP_TRY(ctx->hit_mask.ensure((size_t)mask_words + 2));
            if (fused) HIP_TRY(ctx->hit_pref.ensure((size_t)mask_words + 2));
            HIP_TRY(hipMemcpyAsync(ctx->hit_word0.p, word0.data(), word0.size() * 4, hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(hipMemcpyAsync(ctx->emit_blocks.p, blocks.data(), blocks.size() * sizeof(EmitBlock), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(hipMemsetAsync(ctx->hit_mask.p, 0, ((size_t)mask_words + 2) * 8, ctx->stream));
            const Dev Dm = ctx->dev();
            for (int row = 0; row <= RT_MAX_LIGHTS; ++row) {
                if (!row_used[row]) continue;
                for (int level = 0; level < Lv; ++level)
                    hipLaunchKernelGGL(k_hitmask, dim3(ctx->nb), dim3(BLOCK), level_table_bytes(ctx, level), ctx->stream, Dm, level,
                                       ctx->hit_word0.p + row * row_len, ctx->hit_mask.p, row == 0 ? (int)C_HIT : (int)C_BEAM0 + row - 1);
            }
            if (fused) hipLaunchKernelGGL(k_hitprefix, dim3(K), dim3(BLOCK), 0, ctx->stream, ctx->hit_mask.p, ctx->hit_word0.p, ctx->hit_pref.p);
            HIP_TRY(hipGetLastError());
        }
    }
    ctx->mt_raw.transient = true;
    if (!fused) HIP_TRY(ctx->mt_raw.ensure((size_t)per_pass * L + 2));
    // (tests of the store map: what the generator leaves out must not happen to hold an earlier call's words)
    if (!fused && getenv("RT_MT_POISON")) HIP_TRY(hipMemsetAsync(ctx->mt_raw.p, 0xA5, ctx->mt_raw.n * sizeof(uint32_t), ctx->stream));
    // The state NumPy would be left in: the key block that holds the last word consumed and
    // the position after it (a position of 624 means "regenerate at the next draw").
    const long long block = (next - 1) / MT_N;
    const int last_stream = T - 1;
    const Dev D = ctx->dev();
    const int64_t pack_blocks = std::max<int64_t>(1, std::min<int64_t>((ctx->pack_E + BLOCK - 1) / BLOCK, 4 * NB));
    if (jumping) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->side_join, 0));
    if (fused)
        hipLaunchKernelGGL(k_mt_emit, dim3(T), dim3(256), 0, ctx->stream, D, ctx->mt_ring.p, ctx->mt_end.p, ell, (long long)pos, total, next,
                           last_stream, (block + 1) * MT_N, ctx->emit_blocks.p, getenv("RT_EMIT_GENERATE_ONLY") ? -1 : n_emit_blocks, ctx->hit_mask.p, ctx->hit_pref.p);
    // steps (227 words) a stream makes after its window, and the map's bytes for them
    const int steps_per_stream = 2 * (int)((L + 2 - MT_N + 2 * MT_STEP - 1) / (2 * MT_STEP));
    const int need_words = (steps_per_stream + 8 * MAP_GROUP - 1) / (8 * MAP_GROUP) * MAP_GROUP;
    if (mapped) {
        const long long n_bytes = (long long)T * need_words * 8;
        ctx->mt_need.transient = true;
        HIP_TRY(ctx->mt_need.ensure((size_t)T * need_words));
        HIP_TRY(ctx->mt_stat.ensure(1));
        HIP_TRY(hipMemsetAsync(ctx->mt_stat.p, 0, sizeof(unsigned long long), ctx->stream));
        hipLaunchKernelGGL(k_store_map, dim3((unsigned)std::min<long long>((n_bytes + 255) / 256, 8192)), dim3(256), 0, ctx->stream, ctx->emit_blocks.p, n_emit_blocks, ctx->hit_mask.p,
                           (long long)pos, total, ell, steps_per_stream, need_words * 8, n_bytes, reinterpret_cast<uint8_t*>(ctx->mt_need.p), ctx->mt_stat.p);
    }
    bool points_made = false, raw_only = false;
    // Several passes, frame shaded in mixed precision: the passes are made of whole (node, light) blocks, so that
    // every block's samples are complete within its pass and k_pack makes their disk points on the way, as in
    // the single pass.  The streams at a pass's ends are generated again by its neighbour, from a copy of the start
    // windows.  (RT_MT_RANGE_PASSES=1, or a block that no pass holds: passes by ranges of the stream, uniforms
    // only, the points afterwards.)
    struct BlockPass { int t0, n_streams; };
    std::vector<BlockPass> block_passes;
    std::vector<uint8_t> pass_of;
    if (!fused && mapped && per_pass < T && ctx->pack_E > 0 && ctx->have_packed32 && ctx->precision != RT_PRECISION_F64 && !getenv("RT_MT_RANGE_PASSES")) {
        pass_of.assign((size_t)ctx->tree.n_nodes() * RT_MAX_LIGHTS, 0xFF);
        for (const EmitBlock& b : blocks) {
            const long long first_word = (long long)pos + 2 * b.off, last_word = (long long)pos + 2 * b.end - 1;
            const int ta = (int)std::max(0ll, (first_word - 1) >> ell), tb = (int)std::min<long long>(T - 1, last_word >> ell);
            if (block_passes.empty() || tb - block_passes.back().t0 + 1 > per_pass) block_passes.push_back({ta, 0});
            BlockPass& P = block_passes.back();
            P.n_streams = std::max(P.n_streams, tb - P.t0 + 1);
            if (P.n_streams > per_pass || block_passes.size() > 250) {   // a block larger than the buffer
                block_passes.clear();
                break;
            }
            pass_of[(size_t)b.node * RT_MAX_LIGHTS + b.light] = (uint8_t)(block_passes.size() - 1);
        }
    }
    if (!block_passes.empty()) {
        ctx->mt_ring0.transient = true;
        HIP_TRY(ctx->mt_ring0.ensure((size_t)T * MT_RING));
        HIP_TRY(ctx->mt_end0.ensure((size_t)T));
        HIP_TRY(ctx->pass_of.ensure(pass_of.size()));
        HIP_TRY(hipMemcpyAsync(ctx->pass_of.p, pass_of.data(), pass_of.size(), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(ctx->mt_ring0.p, ctx->mt_ring.p, (size_t)T * MT_RING * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(ctx->mt_end0.p, ctx->mt_end.p, (size_t)T * sizeof(long long), hipMemcpyDeviceToDevice, ctx->stream));
        for (size_t p = 0; p < block_passes.size(); ++p) {
            const int t0 = block_passes[p].t0, n_streams = block_passes[p].n_streams;
            if (p > 0) {   // (the first streams may have run in the pass before)
                HIP_TRY(hipMemcpyAsync(ctx->mt_ring.p + (size_t)t0 * MT_RING, ctx->mt_ring0.p + (size_t)t0 * MT_RING,
                                       (size_t)n_streams * MT_RING * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
                HIP_TRY(hipMemcpyAsync(ctx->mt_end.p + t0, ctx->mt_end0.p + t0, (size_t)n_streams * sizeof(long long), hipMemcpyDeviceToDevice, ctx->stream));
            }
            const long long w_lo = (long long)t0 << ell, w_first = w_lo + (t0 > 0 ? 1 : 0);
            const long long w_hi = std::min(((long long)(t0 + n_streams) << ell) + 2, next);
            hipLaunchKernelGGL(k_mt_generate, dim3(n_streams), dim3(256), 0, ctx->stream, ctx->mt_ring.p, ctx->mt_end.p, ctx->mt_raw.p,
                               t0, ell, w_lo, w_hi, last_stream, (block + 1) * MT_N, ctx->mt_need.p, need_words);
            const long long p_lo = w_first <= pos ? 0 : (w_first - pos + 1) / 2;
            const long long p_hi = (w_hi - 2 - pos) / 2;
            hipLaunchKernelGGL((k_pack<GeneratedStream>), dim3((unsigned)pack_blocks), dim3(BLOCK), 0, ctx->stream, D,
                               GeneratedStream{ctx->mt_raw.p + (pos + 2 * p_lo - w_lo)}, (int64_t)p_lo, (int64_t)(p_hi - p_lo + 1), 1,
                               ctx->pass_of.p, (int)p);
        }
        points_made = true;
    }
    for (int t0 = 0; t0 < T && !fused && block_passes.empty(); t0 += per_pass) {
        const int n_streams = std::min(per_pass, T - t0);
        const long long w_lo = (long long)t0 << ell;                                   // origin of the pass's buffer
        const long long w_first = w_lo + (t0 > 0 ? 1 : 0);                             // first word the pass delivers
        const long long w_hi = std::min(((long long)(t0 + n_streams) << ell) + 2, next); // one past its last word
        hipLaunchKernelGGL(k_mt_generate, dim3(n_streams), dim3(256), 0, ctx->stream, ctx->mt_ring.p, ctx->mt_end.p, ctx->mt_raw.p,
                           t0, ell, w_lo, w_hi, last_stream, (block + 1) * MT_N, mapped ? ctx->mt_need.p : nullptr, need_words);
        // values whose two words both lie in [w_first, w_hi): value p uses words pos + 2p, pos + 2p + 1
        // (a value that straddles two passes is whole in the earlier one: it reaches two words further)
        const long long p_lo = w_first <= pos ? 0 : (w_first - pos + 1) / 2;
        const long long p_hi = (w_hi - 2 - pos) / 2;                                   // inclusive
        // one pass over the whole stream: the float32 pass's disk points are made along the way
        const bool with_points = per_pass >= T && p_lo == 0 && p_hi + 1 == total && ctx->have_packed32 && ctx->precision != RT_PRECISION_F64;
        // ... and if the raw words stay with the context, the points are all that is written: the few hits
        // whose float32 lighting cannot be certified read their uniforms from the raw words (RT_PACK_UNIFORMS=1: always both)
        const bool points_only = with_points && ctx->precision == RT_PRECISION_MIXED && ctx->mt_raw.n * sizeof(uint32_t) <= mt_raw_keep_bytes() &&
                                 !getenv("RT_PACK_UNIFORMS");
        if (ctx->pack_E > 0 && p_hi >= p_lo && w_hi - 2 - pos >= 0) {
            hipLaunchKernelGGL((k_pack<GeneratedStream>), dim3((unsigned)pack_blocks), dim3(BLOCK), 0, ctx->stream, D,
                               GeneratedStream{ctx->mt_raw.p + (pos + 2 * p_lo - w_lo)}, (int64_t)p_lo, (int64_t)(p_hi - p_lo + 1),
                               points_only ? 2 : with_points ? 1 : 0, nullptr, 0);
            if (with_points) points_made = true;
            if (points_only) {
                raw_only = true;
                ctx->raw_value0 = pos - w_lo;
                ctx->raw_slot_host = ctx->mt_raw.p + ctx->raw_value0;
                HIP_TRY(hipMemcpyAsync(ctx->raw_slot.p, &ctx->raw_slot_host, sizeof(const uint32_t*), hipMemcpyHostToDevice, ctx->stream));
            }
        }
    }
    HIP_TRY(hipGetLastError());
    // the block may begin in the stream before the last one
    std::vector<uint32_t> prev_ring(MT_RING, 0u);
    HIP_TRY(hipMemcpyAsync(ring.data(), ctx->mt_ring.p + (size_t)last_stream * MT_RING, MT_RING * sizeof(uint32_t),
                           hipMemcpyDeviceToHost, ctx->stream));
    if (last_stream > 0)
        HIP_TRY(hipMemcpyAsync(prev_ring.data(), ctx->mt_ring.p + (size_t)(last_stream - 1) * MT_RING, MT_RING * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, ctx->stream));
    unsigned long long kept = 0;
    if (mapped) HIP_TRY(hipMemcpyAsync(&kept, ctx->mt_stat.p, sizeof kept, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const long long last_first = (long long)last_stream << ell;
    for (int i = 0; i < MT_N; ++i) {
        const long long n = block * MT_N + i;
        key_out[i] = (n > last_first ? ring : prev_ring)[n & (MT_RING - 1)];   // the last stream's own first word is not whole
    }
    *pos_out = (int32_t)(next - block * MT_N);
    ctx->samples_delivered = ctx->sample_total;   // the whole stream of this context has been made
    ctx->disk_points_ready = points_made;
    ctx->samples_in_raw = raw_only;
    ctx->generator_words = next;
    ctx->generator_words_stored = fused ? 0 : next;
    if (mapped) ctx->generator_words_stored = (long long)kept + (long long)T * MT_N;   // and every stream's window
    // The raw words are only needed while they are being packed, but getting device memory back that
    // some process has used before takes the driver tens of milliseconds per GB (measured: 1.3 s for
    // the 50 GB of a 4K example_4 frame, every call), so a buffer of moderate size stays with the context.
    if (ctx->mt_raw.n * sizeof(uint32_t) > mt_raw_keep_bytes()) ctx->mt_raw.release();
    return RT_OK;
}

int rt_trim(RtContext* ctx) {
    if (int rc = use_device(ctx)) return rc;
    if (int rc = materialize_packed(ctx)) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->mt_raw.release(); ctx->mt_ring.release(); ctx->mt_end.release(); ctx->mt_polys.release(); ctx->mt_need.release(); ctx->mt_stat.release();
    ctx->mt_ring0.release(); ctx->mt_end0.release(); ctx->pass_of.release();
    ctx->mt_polys_ell = -1;
    ctx->staging.release(); ctx->hit_mask.release(); ctx->hit_pref.release(); ctx->hit_word0.release(); ctx->emit_blocks.release();
    return RT_OK;
}

int rt_mt19937_jump_poly(int32_t k, uint32_t* out624) {
    if (!out624) return fail(RT_ERR_INVALID, "out624 is null");
    if (!mtjump::pow2_words(k, out624)) return fail(RT_ERR_INVALID, "no jump polynomial for 2^%d words", k);
    return RT_OK;
}

int rt_shade(RtContext* ctx) {
    if (int rc = use_device(ctx)) return rc;
    if (!ctx->geometry_done) return fail(RT_ERR_STATE, "geometry phase not finished");
    if (ctx->sampled_lights && !ctx->layout_set)
        return fail(RT_ERR_STATE, "the scene has disk lights: upload their samples before shading");
    if (ctx->sampled_lights && ctx->samples_delivered < ctx->sample_total)
        return fail(RT_ERR_STATE, "only %lld of the %lld disk-light sample values have been uploaded or generated",
                    (long long)ctx->samples_delivered, (long long)ctx->sample_total);
    const bool lds = ctx->S <= SHADE_LDS_SPHERES;
    // (the float32 sample pass stores float32 colours: only for frames whose geometry ran in mixed precision)
    const bool mixed = ctx->precision != RT_PRECISION_F64 && ctx->frame_col32 && ctx->sampled_lights && ctx->S <= MIXED_MAX_SPHERES && ctx->have_packed32;
    const bool check = ctx->precision == RT_PRECISION_MIXED_CHECKED;
    if (mixed && check)
        if (int rc = materialize_packed(ctx)) return rc;   // the checking pass reads every sample's uniforms
    const Dev D = ctx->dev();
    if (mixed && !ctx->disk_points_ready) {
        // once per sample stream (not per frame): the float32 pass reads disk points, see k_disk_points.
        // Not while the stream is being captured: the kernel would be recorded, not run, and a later
        // rt_shade outside the graph would find the flag set and the points missing.
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(ctx->stream, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone)
            return fail(RT_ERR_STATE, "the first rt_shade after new disk-light samples prepares them for the float32 pass and "
                                      "cannot be captured into a graph: call rt_shade once before the capture");
        for (int l = 0; l < ctx->sc_host.n_lights && ctx->pack_E > 0; ++l) {
            const RtLight& L = ctx->sc_host.lights[l];
            if (L.kind == RT_LIGHT_POINT) continue;
            const int64_t n = ((ctx->pack_E + TILE - 1) / TILE) * ((L.n_points + 1) / 2) * TILE;
            const int64_t blocks = std::min<int64_t>((n + BLOCK - 1) / BLOCK, 16 * NB);
            hipLaunchKernelGGL(k_disk_points, dim3((unsigned)blocks), dim3(BLOCK), 0, ctx->stream, ctx->packed.p + ctx->pack_base[l],
                               ctx->pack_plane, ctx->pack_E, L.n_points, reinterpret_cast<float4*>(ctx->packed32.p) + ctx->pack32_base[l]);
        }
        ctx->disk_points_ready = true;
    }
    ctx->begin(RT_K_SURFACE);
    if (mixed && check && ctx->simple_shading) hipLaunchKernelGGL((k_surface_mixed<true, true>), dim3(NB), dim3(BLOCK), 0, ctx->stream, D);
    else if (mixed && check) hipLaunchKernelGGL((k_surface_mixed<false, true>), dim3(NB), dim3(BLOCK), 0, ctx->stream, D);
    else if (mixed && ctx->simple_shading) hipLaunchKernelGGL((k_surface_mixed<true, false>), dim3(NB), dim3(BLOCK), 0, ctx->stream, D);
    else if (mixed) hipLaunchKernelGGL((k_surface_mixed<false, false>), dim3(NB), dim3(BLOCK), 0, ctx->stream, D);
    else {
        auto kernel = lds ? (D.col32 ? k_surface<true, true> : k_surface<true, false>)
                          : ctx->sampled_lights ? (D.col32 ? k_surface<false, true> : k_surface<false, false>)
                          : ctx->grid_cells     ? (D.col32 ? k_surface<false, true, true, true> : k_surface<false, false, true, true>)
                                                : (D.col32 ? k_surface<false, true, true> : k_surface<false, false, true>);
        hipLaunchKernelGGL(kernel, dim3(NB), dim3(BLOCK), 0, ctx->stream, D);
    }
    if (mixed) hipLaunchKernelGGL((k_surface_redo<true>), dim3(64), dim3(BLOCK), 0, ctx->stream, D);
    ctx->end();
    for (int level = D.n_levels - 1; level >= 0; --level) {
        ctx->begin(RT_K_COMBINE);
        const bool small = ctx->S <= SPH_CHUNK_SMALL;
        auto kernel = small ? (D.col32 ? k_combine<true, true> : k_combine<true, false>) : (D.col32 ? k_combine<false, true> : k_combine<false, false>);
        hipLaunchKernelGGL(kernel, dim3(ctx->nb), dim3(BLOCK), level_table_bytes(ctx, level), ctx->stream, D, level);
        ctx->end();
    }
    HIP_TRY(hipGetLastError());
    ctx->shaded = true;
    return RT_OK;
}

int rt_set_batch_buffer(RtContext* ctx, int32_t* device_batches) {
    if (int rc = use_device(ctx)) return rc;
    if (!ctx->have_scene) return fail(RT_ERR_STATE, "rt_set_scene has not been called");
    ctx->user_batches = device_batches;
    ctx->batch_slot = 0; ctx->batch_slots = 1;
    return RT_OK;
}

int rt_set_batch_buffer_shared(RtContext* ctx, int32_t* device_batches, int32_t slot, int32_t n_slots) {
    if (int rc = use_device(ctx)) return rc;
    if (!ctx->have_scene) return fail(RT_ERR_STATE, "rt_set_scene has not been called");
    if (!device_batches || n_slots < 1 || slot < 0 || slot >= n_slots)
        return fail(RT_ERR_INVALID, "slot %d of %d frames in a shared batch buffer", slot, n_slots);
    ctx->user_batches = device_batches;
    ctx->batch_slot = slot; ctx->batch_slots = n_slots;
    return RT_OK;
}

int rt_set_batch_guess(RtContext* ctx, const int32_t* device_guess) {
    if (int rc = use_device(ctx)) return rc;
    if (!ctx->have_scene) return fail(RT_ERR_STATE, "rt_set_scene has not been called");
    ctx->batch_guess = device_guess;
    return RT_OK;
}

int rt_set_output(RtContext* ctx, void* device_rgb) {
    if (int rc = use_device(ctx)) return rc;
    if (!ctx->frame_begun) return fail(RT_ERR_STATE, "rt_begin_frame has not been called");
    ctx->user_image = (double*)device_rgb;
    return RT_OK;
}

int rt_set_precision(RtContext* ctx, int32_t mode) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    if (mode != RT_PRECISION_F64 && mode != RT_PRECISION_MIXED && mode != RT_PRECISION_MIXED_CHECKED) return fail(RT_ERR_INVALID, "unknown precision mode %d", mode);
    ctx->precision = mode;
    return RT_OK;
}

int rt_synchronize(RtContext* ctx) {
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_read_image(RtContext* ctx, double* out_rgb) {
    if (int rc = use_device(ctx)) return rc;
    if (!ctx->shaded) return fail(RT_ERR_STATE, "rt_shade has not been called for this frame");
    if (!out_rgb) return fail(RT_ERR_INVALID, "out_rgb is null");
    int32_t flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, ctx->flags.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (int rc = copy_to_host(ctx, out_rgb, ctx->user_image ? ctx->user_image : ctx->image.p, (size_t)ctx->npix * 3 * sizeof(double))) return rc;
    if (flag) return fail(RT_ERR_OVERFLOW, "ray arena of %lld rays is too small for this frame", (long long)ctx->arena_rays);
    return RT_OK;
}

int rt_alloc_host(size_t bytes, void** out) {
    if (!out) return fail(RT_ERR_INVALID, "out is null");
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, std::max<size_t>(bytes, 1), hipHostMallocDefault));
    return RT_OK;
}

void rt_free_host(void* p) {
    if (p) (void)hipHostFree(p);
}

int rt_quantise_rgb8(RtContext* ctx, void* device_rgb8) {
    if (int rc = use_device(ctx)) return rc;
    if (!ctx->shaded) return fail(RT_ERR_STATE, "rt_shade has not been called for this frame");
    const int64_t n = (int64_t)ctx->npix * 3;
    if (!device_rgb8) {
        HIP_TRY(ctx->image8.ensure((size_t)n));
        device_rgb8 = ctx->image8.p;
    }
    ctx->user_image8 = (uint8_t*)device_rgb8;
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((n + BLOCK - 1) / BLOCK, 8 * NB));
    hipLaunchKernelGGL(k_quantise, dim3((unsigned)blocks), dim3(BLOCK), 0, ctx->stream,
                       ctx->user_image ? ctx->user_image : ctx->image.p, ctx->user_image8, n);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_read_image_rgb8(RtContext* ctx, uint8_t* out_rgb8) {
    if (int rc = use_device(ctx)) return rc;
    if (!out_rgb8) return fail(RT_ERR_INVALID, "out_rgb8 is null");
    if (int rc = rt_quantise_rgb8(ctx, nullptr)) return rc;
    int32_t flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, ctx->flags.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (int rc = copy_to_host(ctx, out_rgb8, ctx->user_image8, (size_t)ctx->npix * 3)) return rc;
    if (flag) return fail(RT_ERR_OVERFLOW, "ray arena of %lld rays is too small for this frame", (long long)ctx->arena_rays);
    return RT_OK;
}

int rt_image_device_ptr(RtContext* ctx, void** out) {
    if (!ctx || !out) return fail(RT_ERR_INVALID, "null argument");
    if (!ctx->frame_begun) return fail(RT_ERR_STATE, "rt_begin_frame has not been called");
    *out = ctx->user_image ? ctx->user_image : ctx->image.p;
    return RT_OK;
}

int rt_get_counters(RtContext* ctx, RtCounters* out) {
    if (int rc = use_device(ctx)) return rc;
    if (!out) return fail(RT_ERR_INVALID, "out is null");
    if (!ctx->geometry_done) return fail(RT_ERR_STATE, "geometry phase not finished");
    const int K = ctx->tree.n_nodes(), Lv = ctx->tree.n_levels();
    std::vector<int32_t> n(K), ng(K), h(K), nt(Lv), tb(Lv);
    int32_t flag = 0;
    unsigned long long work[6] = {0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(work, ctx->work.p, sizeof work, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(n.data(), ctx->node_n.p, K * 4, hipMemcpyDeviceToHost, ctx->stream));
    for (int l = 0; l < Lv; ++l) {
        const int a = ctx->tree.level_start[l], b = ctx->tree.level_start[l + 1];
        HIP_TRY(hipMemcpyAsync(ng.data() + a, ctx->level_batches(l), (size_t)(b - a) * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipMemcpyAsync(h.data(), ctx->node_hits.p, K * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(nt.data(), ctx->lvl_ntiles.p, Lv * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(tb.data(), ctx->lvl_tilebase.p, Lv * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&flag, ctx->flags.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    memset(out, 0, sizeof *out);
    out->pixels = ctx->npix;
    for (int k = 0; k < K; ++k) {
        if (ng[k] >= ctx->params.small_num) out->rays += n[k];
        out->hits += h[k];
    }
    out->shaded_hits = (int64_t)work[0];
    out->shaded_samples = (int64_t)work[1];
    out->fallback_samples = (int64_t)work[2];
    out->stored_records = (int64_t)work[3];
    out->live_hits = (int64_t)work[4];
    out->arena_rays_used = (int64_t)(tb[Lv - 1] + nt[Lv - 1]) * TILE;
    out->arena_capacity = ctx->arena_rays;
    out->n_nodes = K;
    out->n_levels = Lv;
    out->overflow = flag;
    out->certify_mismatches = (int32_t)std::min<unsigned long long>(work[5], 0x7FFFFFFFull);
    out->generator_words = ctx->generator_words;
    out->generator_words_stored = ctx->generator_words_stored;
    return RT_OK;
}

int rt_profile_enable(RtContext* ctx, int32_t on) {
    if (int rc = use_device(ctx)) return rc;
    if (!on && ctx->profiling) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->collect();
    }
    ctx->profiling = on != 0;
    return RT_OK;
}

int rt_get_profile(RtContext* ctx, RtProfile* out, int32_t reset) {
    if (int rc = use_device(ctx)) return rc;
    if (!out) return fail(RT_ERR_INVALID, "out is null");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->collect();
    *out = ctx->prof;
    if (reset) ctx->prof = RtProfile{};
    return RT_OK;
}

int rt_debug_eval(RtContext* ctx, int32_t what, const double* in, double* out, int64_t n) {
    if (int rc = use_device(ctx)) return rc;
    if (!ctx->have_scene) return fail(RT_ERR_STATE, "rt_set_scene has not been called");
    static const int widths[][2] = {{7, 2}, {6, 2}, {6, 3}, {8, 8}, {4, 1}, {2, 6}};   // per RT_DEBUG_*: in, out
    if (what < 0 || what > RT_DEBUG_SCREEN) return fail(RT_ERR_INVALID, "unknown debug function %d", what);
    if (n < 0 || (n > 0 && (!in || !out))) return fail(RT_ERR_INVALID, "bad debug arguments");
    if (what == RT_DEBUG_NEAREST && ctx->S > SPH_CHUNK_SMALL) return fail(RT_ERR_INVALID, "RT_DEBUG_NEAREST takes at most %d spheres", SPH_CHUNK_SMALL);
    if (what == RT_DEBUG_SHADOW && ctx->S > SHADE_LDS_SPHERES) return fail(RT_ERR_INVALID, "RT_DEBUG_SHADOW takes at most %d spheres", SHADE_LDS_SPHERES);
    if (n == 0) return RT_OK;
    const int in_w = widths[what][0], out_w = widths[what][1];
    if (what == RT_DEBUG_SPHERE_ROOTS)
        for (int64_t i = 0; i < n; ++i) {
            const double sph = in[i * in_w + 6];
            if (!(sph >= 0 && sph < ctx->S && sph == (double)(int)sph)) return fail(RT_ERR_INVALID, "row %lld: sphere index out of range", (long long)i);
        }
    DevBuf<double> d_in, d_out;
    d_in.transient = d_out.transient = true;
    HIP_TRY(d_in.ensure((size_t)n * in_w));
    hipError_t e = d_out.ensure((size_t)n * out_w);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in.p, in, (size_t)n * in_w * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((n + BLOCK - 1) / BLOCK, NB));
        hipLaunchKernelGGL(k_debug, dim3((unsigned)blocks), dim3(BLOCK), 0, ctx->stream, ctx->dev(), (int)what, d_in.p, in_w, d_out.p, out_w, (long long)n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out.p, (size_t)n * out_w * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    d_in.release();
    d_out.release();
    if (e != hipSuccess) return fail(RT_ERR_HIP, "rt_debug_eval: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_render(RtContext* ctx, double* out_rgb) {
    if (int rc = use_device(ctx)) return rc;
    if (!ctx->have_scene) return fail(RT_ERR_STATE, "rt_set_scene has not been called");
    if (ctx->sampled_lights) return fail(RT_ERR_STATE, "rt_render cannot draw disk-light samples; use the phased calls");
    if (int rc = rt_begin_frame(ctx, 0, ctx->params.height, nullptr, nullptr, 0, 0)) return rc;
    if (int rc = rt_trace_geometry(ctx)) return rc;
    if (int rc = rt_shade(ctx)) return rc;
    return rt_read_image(ctx, out_rgb);
}

}  // extern "C"
