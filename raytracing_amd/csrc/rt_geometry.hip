// Geometry kernels, one launch of each per bounce level: k_trace (nearest hit + hit record),
// k_scan (counts -> node sizes and slots), k_link (ordered compaction); k_init starts a frame.
#pragma once
#include "rt_math.h"

namespace rt {

// Whole-image batch size of node k (a node of `level`).  The sizes of `batch_slots` frames can
// share one buffer, laid out level by level and, inside a level, frame by frame, so that all
// frames' sizes of one level are one contiguous range (one sum over the GPUs per level serves
// every frame).  With one frame the index is k itself.
__device__ __forceinline__ int32_t& whole_image_batch(const Dev& D, int k, int level) {
    const int a = D.level_start[level], b = D.level_start[level + 1];
    return D.node_nglobal[k + (D.batch_slots - 1) * a + D.batch_slot * (b - a)];
}

// The size the small-batch rule (reference raytrace.py:26-27) compares with SMALL_NUM: the node's
// whole-image batch size, or, while the caller has not summed the sizes over its GPUs yet, the
// caller's guess for it (rt_set_batch_guess).
__device__ __forceinline__ int32_t batch_for_small_rule(const Dev& D, int k, int level) {
    if (!D.node_nguess) return whole_image_batch(D, k, level);
    const int a = D.level_start[level], b = D.level_start[level + 1];
    return D.node_nguess[k + (D.batch_slots - 1) * a + D.batch_slot * (b - a)];
}

// ---------------------------------------------------------------------------------------------
__global__ void k_init(Dev D) {
    const int tid = threadIdx.x;
    for (int level = 0; level < D.n_levels; ++level)
        for (int k = D.level_start[level] + tid; k < D.level_start[level + 1]; k += blockDim.x)
            whole_image_batch(D, k, level) = 0;
    for (int k = tid; k < D.K; k += blockDim.x) {
        D.node_n[k] = 0;
        D.node_hits[k] = 0;
        D.node_tile0[k] = 0;
        D.node_ntiles[k] = 0;
        D.node_start[k] = 0;
        for (int l = 0; l < RT_MAX_LIGHTS; ++l) D.node_beam[k * RT_MAX_LIGHTS + l] = 0;
        for (int c = 0; c < NCNT; ++c) D.node_pre[k * NCNT + c] = 0;
    }
    for (int l = tid; l < D.n_levels; l += blockDim.x) {
        D.lvl_ntiles[l] = 0;
        D.lvl_tilebase[l] = 0;
    }
    for (int l = tid; l <= D.n_levels; l += blockDim.x) D.lvl_shade_base[l] = 0;
    __syncthreads();
    if (tid == 0) {
        int nt = (D.npix + TILE - 1) / TILE;
        D.flags[0] = 0;
        D.flags[1] = 0;
        D.work[0] = 0;
        D.work[1] = 0;
        D.work[2] = 0;
        D.work[3] = 0;
        D.work[4] = 0;
        D.work[5] = 0;
        if ((uint32_t)nt > D.cap_tiles) {
            D.flags[0] = 1;
            nt = 0;
        }
        D.node_n[0] = nt ? D.npix : 0;
        whole_image_batch(D, 0, 0) = D.node_n[0];
        D.node_ntiles[0] = nt;
        D.lvl_ntiles[0] = nt;
    }
}

// ---------------------------------------------------------------------------------------------
// Pieces of k_trace that are also callable on their own (rt_debug_eval checks them against
// vectors recorded from the reference's functions).

// Primary ray of pixel (X, Y) (reference objects.py:45-51 and raytrace.py:7-12): point on the
// screen and unit direction from the camera.
__device__ __forceinline__ V3 screen_point(const SceneConst& sc, double X, double Y) {
    const double sx = X * sc.scr_w / sc.pix_x - sc.scr_w / 2;
    const double sy = (sc.pix_y - Y) * sc.scr_h / sc.pix_y - sc.scr_h / 2;
    return sx * ld3(sc.scr_x) + sy * ld3(sc.scr_y) + ld3(sc.scr_pos);
}

// Nearest positive root over a sphere table [S][4] = centre, radius^2 (reference objects.py:288-299):
// the float64 test for every sphere, two per iteration (two independent dependency chains in
// flight); ties go to the later sphere, as the reference's min_t == t update does.
__device__ __forceinline__ void nearest_of_table(const double (*sph)[4], int S, V3 o, V3 d, double d_o, double& best, int& sid) {
    auto test = [&](int i) {
        double lo, hi;
        const bool h = sphere_roots(o, d, d_o, sph[i][0], sph[i][1], sph[i][2], sph[i][3], lo, hi);
        return h ? (lo > 0 ? lo : hi) : -1.0;
    };
    int i = 0;
    for (; i + 1 < S; i += 2) {
        const double t0 = test(i), t1 = test(i + 1);
        if (t0 > 0 && t0 <= best) { best = t0; sid = i; }
        if (t1 > 0 && t1 <= best) { best = t1; sid = i + 1; }
    }
    if (i < S) {
        const double t0 = test(i);
        if (t0 > 0 && t0 <= best) { best = t0; sid = i; }
    }
}

// Refraction and Fresnel reflectance at a hit (reference raytrace.py:92-99, :173-212).  N faces
// along the ray (d.N >= 0), mu = n1 / n2, r0 = ((n1 - n2) / (n1 + n2))^2.
struct Optics {
    bool trans;   // a transmitted ray exists (no total internal reflection)
    double kq;    // 1 - mu^2 (1 - cos^2): squared length of the transmitted vector along N
    V3 T;         // transmitted direction (zero when !trans)
    double R;     // reflectance: Schlick's approximation, 1 under total internal reflection
};
__device__ __forceinline__ double refraction_discriminant(double mu, double cosi) { return 1 - mu * mu * (1 - cosi * cosi); }
__device__ __forceinline__ Optics hit_optics(V3 N, V3 d, double cosi, double mu, double r0, bool n2_ge_n1) {
    Optics o;
    o.kq = refraction_discriminant(mu, cosi);
    o.trans = o.kq >= 0;
    o.T = {0, 0, 0};
    if (o.trans) o.T = xsqrt(o.kq) * N + mu * (d - cosi * N);   // reference raytrace.py:207-211
    const double cosf = n2_ge_n1 ? cosi : dot(o.T, N);
    const double om = 1 - cosf;
    const double om2 = om * om;
    o.R = r0 + (1 - r0) * (om2 * om2 * om);
    if (!o.trans) o.R = 1;
    return o;
}
// Mirror direction about the surface (reference raytrace.py:192-195 with the normal -N).
__device__ __forceinline__ V3 mirror_direction(V3 d, V3 N, double cosi) { return d - cosi * (2.0 * N); }

// ---------------------------------------------------------------------------------------------
// Nearest hit + hit record for one level.  Every wave walks its own run of tiles.  All the
// small tables a tile needs (the level's nodes, the scene constants and, with few spheres,
// the sphere constants) are staged in LDS once per workgroup, so that the only vector-memory
// traffic inside the tile loop is the ray data itself: parent index, parent record, hit record.
struct NodeRec {
    int tile_end, tile0, n, bits;   // bits: a | b << 8 | branch << 16 | small-batch flag << 24
};
struct SphereRec {   // per-sphere constants of the hit record
    double mu_out, mu_in, r0, ior, floor_, transp, refl_max, pad;
};

// SMALL: at most SPH_CHUNK_SMALL spheres, all constants in LDS.  GRID (many spheres): the scene has a
// uniform grid; the other searches (hierarchy, every sphere) are left out of the kernel, which
// keeps its register demand down.
template <bool SMALL, bool GRID = false>
__global__ __launch_bounds__(BLOCK, SMALL ? RT_TRACE_WAVES : RT_TRACE_WAVES_MANY) void k_trace(Dev D, int level) {
    constexpr int SPH_CHUNK = SMALL ? SPH_CHUNK_SMALL : 1;
    __shared__ NodeRec s_node[MAX_LEVEL_NODES];
    __shared__ double s_sph[SPH_CHUNK][4];
    __shared__ SphereRec s_rec[SMALL ? SPH_CHUNK_SMALL : 1];
    // per lane: the spheres the screening kept (element k of thread t at [k][t])
    static_assert(CAND_MAX <= HIER_CAP, "the plain screening loop uses the same list");
    __shared__ uint16_t s_list[SMALL ? 1 : (GRID ? 1 : 2) * HIER_CAP][SMALL ? 1 : BLOCK];   // (the grid's walk uses one list per lane)
    __shared__ SceneConst s_sc;
    static_assert(sizeof(SceneConst) % 8 == 0, "SceneConst is copied in 8-byte words");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = D.level_start[level], nn = D.level_start[level + 1] - k0;
    const int nt = D.lvl_ntiles[level];
    const int tbase = D.lvl_tilebase[level];
    const int q = tiles_per_wave(D, nt);
    const int wid = blockIdx.x * WPB + wave;
    const int t_first = wid * q;
    const int S = D.S;
    if (blockIdx.x * WPB * q >= nt) {
        // no tile for this workgroup (most of the grid at a deep level): its counters are zero
        if (lane < NCNT) {
            D.wave_off[((size_t)level * NCNT + lane) * NW + wid] = 0;
            if (wave == WPB - 1) D.blk_tot[((size_t)level * NCNT + lane) * NB + blockIdx.x] = 0;
        }
        return;
    }

    for (int i = tid; i < (int)(sizeof(SceneConst) / 8); i += BLOCK)
        reinterpret_cast<double*>(&s_sc)[i] = reinterpret_cast<const double*>(D.sc)[i];
    {
        const int small_num = D.sc->small_num;
        for (int i = tid; i < nn; i += BLOCK) {
            const int k = k0 + i;
            const int t0 = D.node_tile0[k];
            NodeRec r;
            r.tile0 = t0;
            r.tile_end = t0 + D.node_ntiles[k];
            r.n = D.node_n[k];
            r.bits = (D.node_a[k] & 0xff) | ((D.node_b[k] & 0xff) << 8) | ((D.node_branch[k] & 0xff) << 16) |
                     ((batch_for_small_rule(D, k, level) < small_num) ? (1 << 24) : 0);   // reference raytrace.py:26-27
            s_node[i] = r;
        }
        if (SMALL) {
            for (int i = tid; i < S; i += BLOCK) {
                const double* g = D.geom + (size_t)i * 4;
                const double* m = D.mat + (size_t)i * RT_MAT_WIDTH;
                const double* f = D.fres + (size_t)i * 4;
                s_sph[i][0] = g[0]; s_sph[i][1] = g[1]; s_sph[i][2] = g[2]; s_sph[i][3] = g[3] * g[3];
                s_rec[i] = {f[0], f[1], f[2], m[12], m[10], m[11], m[13], 0.0};
            }
        }
    }
    __syncthreads();
    const SceneConst& sc = s_sc;
    const double eps = sc.eps, air = sc.air;
    unsigned run = 0;  // lanes 0..NCNT-1: running counters of this wave

    // Per-tile facts (node, batch, depths): all wave-uniform.
    struct Meta {
        bool ok, dead;
        int t, first, cnt, a, b, br;
    };
    int kk = 0;   // node cursor of the look-ahead
    if (t_first < nt) {
        int lo = 0, hi = nn - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_node[mid].tile_end > t_first) hi = mid; else lo = mid + 1;
        }
        kk = lo;
    }
    auto meta_of = [&](int step) {
        Meta m;
        m.t = t_first + step;
        m.ok = step < q && m.t < nt;
        m.dead = true;
        m.first = 0; m.cnt = 0; m.a = 0; m.b = 0; m.br = 0;
        if (m.ok) {
            while (s_node[kk].tile_end <= m.t) ++kk;
            kk = __builtin_amdgcn_readfirstlane(kk);
            const NodeRec r = s_node[kk];
            const int bits = __builtin_amdgcn_readfirstlane(r.bits);
            m.first = (m.t - __builtin_amdgcn_readfirstlane(r.tile0)) * TILE;
            m.cnt = min(TILE, __builtin_amdgcn_readfirstlane(r.n) - m.first);
            m.a = bits & 0xff; m.b = (bits >> 8) & 0xff; m.br = (bits >> 16) & 0xff;
            m.dead = (bits >> 24) & 1;
        }
        return m;
    };
    // What a ray reads of its parent's hit record.
    struct Parent {
        uint32_t info;
        V3 P, d;
    };
    auto load_link = [&](const Meta& m) -> uint32_t {
        return (level > 0 && lane < m.cnt) ? D.link[(uint32_t)(tbase + m.t) * TILE + lane] : 0u;
    };
    auto load_parent = [&](const Meta& m, uint32_t p) {
        Parent r;
        r.info = 0; r.P = {0, 0, 0}; r.d = {1, 0, 0};
        if (level > 0 && lane < m.cnt) {
            r.info = D.info[p];
            r.P = {D.px[p], D.py[p], D.pz[p]};
            r.d = {D.dx[p], D.dy[p], D.dz[p]};
        }
        return r;
    };
    auto sphere_rec = [&](int sid) -> SphereRec {
        if (SMALL) return s_rec[sid];
        const double* m = D.mat + (size_t)sid * RT_MAT_WIDTH;
        const double* f = D.fres + (size_t)sid * 4;
        return {f[0], f[1], f[2], m[12], m[10], m[11], m[13], 0.0};
    };
    auto sphere_centre = [&](int sid, double& r2) -> V3 {
        if (SMALL) { r2 = s_sph[sid][3]; return {s_sph[sid][0], s_sph[sid][1], s_sph[sid][2]}; }
        const double* g = D.geom + (size_t)sid * 4;
        r2 = g[3] * g[3];
        return {g[0], g[1], g[2]};
    };

    // ---- one tile ---------------------------------------------------------------------------
    auto process = [&](const Meta& m, const Parent& par) {
        const int t = m.t, first = m.first, a = m.a, b = m.b, br = m.br;
        const bool tile_ok = m.ok;
        const uint32_t idx = (uint32_t)(tbase + t) * TILE + lane;
        const bool valid = lane < m.cnt;
        const bool active = valid && !m.dead;

        // ---- derive the ray ------------------------------------------------------------------
        V3 o = {0, 0, 0}, d = {1, 0, 0};
        bool live = true;
        if (valid) {
            if (level == 0) {
                // reference objects.py:45-51 and raytrace.py:7-12
                double X, Y;
                const int i = first + lane;
                if (D.X) { X = D.X[i]; Y = D.Y[i]; }
                else { X = (double)(i % sc.width); Y = (double)(D.row0 + i / sc.width); }
                const V3 pix = screen_point(sc, X, Y);
                o = ld3(sc.cam);
                d = normalize(pix - o);
            } else {
                const uint32_t pinfo = par.info;
                const int psid = pinfo & I_SID_MASK;
                const bool pinside = pinfo & I_INSIDE;
                const V3 pd = par.d;
                double pr2;
                const V3 pc = sphere_centre(psid, pr2);
                const Surface sf = surface_at(par.P, pc, pinside, eps);
                const double cosi = dot(sf.N, pd);
                if (br == 0) {
                    // transmitted ray (reference raytrace.py:198-212, :104-107): the same
                    // operations on the same values as in the parent's own pass
                    const double mu = SMALL ? (pinside ? s_rec[psid].mu_in : s_rec[psid].mu_out)
                                            : D.fres[(size_t)psid * 4 + (pinside ? 1 : 0)];
                    const double kk_ = 1 - mu * mu * (1 - cosi * cosi);
                    const double sk = xsqrt(fmax(kk_, 0.0));
                    d = sk * sf.N + mu * (pd - cosi * sf.N);
                    o = sf.Pb + (2 * eps) * sf.N;
                    live = pinfo & I_LIVE_TI;
                } else {
                    // mirror direction about the surface (reference raytrace.py:192-195);
                    // total internal reflection and plain reflection share it
                    d = mirror_direction(pd, sf.N, cosi);
                    o = sf.Pb;
                    live = (br == 1) ? (pinfo & I_LIVE_TI) : (pinfo & I_LIVE_R);
                }
            }
        }

        // ---- nearest hit (reference objects.py:288-299) ------------------------------------
        const double d_o = dot(d, o);
        double best = __builtin_inf();
        int sid = -1;
        if (!SMALL && __ballot(active) != 0) {
            // Many spheres: float32 screening (rt_math.h: cull_test) first, the float64 test only
            // for the spheres it cannot rule out.  The screening looks at the ray's whole line, so
            // any point of the line can stand in for the origin: the one nearest to the middle of
            // the spheres keeps the float32 coordinates (and with them the error bound) small for
            // rays that start far away.
            const V3 ref = {D.cull_ref[0], D.cull_ref[1], D.cull_ref[2]};
            const V3 on = o + dot(ref - o, d) * d;
            const float ox = (float)on.x, oy = (float)on.y, oz = (float)on.z;
            const float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
            const float e_ray = cull_ray_margin(on);   // this ray's part of the error bound
            auto exact = [&](int i) {
                const double* g = D.geom + (size_t)i * 4;
                double lo, hi;
                if (sphere_roots(o, d, d_o, g[0], g[1], g[2], g[3] * g[3], lo, hi)) {
                    const double tt = lo > 0 ? lo : hi;
                    // ties go to the later sphere, as the reference's min_t == t update does in its
                    // loop over all spheres; written so that the order of the tests does not matter
                    if (tt > 0 && (tt < best || (tt == best && i > sid))) { best = tt; sid = i; }
                }
            };
            const LaneList list = {&s_list[0][SMALL ? 0 : tid]}, clusters = {&s_list[(SMALL || GRID) ? 0 : HIER_CAP][SMALL ? 0 : tid]};
            bool every_sphere = active;   // lanes that screen the whole table
            if (GRID || D.grid_cells) {
                // Uniform grid: the spheres outside it first (a hit there bounds the walk), then the
                // cells along the ray, front to back.  Candidates collect in the lane's list and take
                // the float64 test together, every few steps or when some lane's list is nearly full; a
                // lane's walk ends at the first cell that begins beyond the nearest hit it knows.
                typedef const __attribute__((address_space(4))) uint16_t* ConstIds;   // uniform reads through the scalar unit
                ConstIds loose = (ConstIds)(uintptr_t)D.grid_loose;
                for (int k = 0; k < D.grid_n_loose; ++k)
                    if (active) exact(loose[k]);
                GridWalk w = grid_begin(D, active, o, d);
                int n = 0;
                auto push = [&](uint32_t id) { list[n++] = (uint16_t)id; };   // two free places are kept
                const float slack = 2 * D.grid_margin;
                auto reach = [&]() -> float {
                    if (!(best < __builtin_inf())) return w.t_out;
                    return fminf(w.t_out, (float)((best - w.enter) * 1.000001) + slack);
                };
                float limit = reach();
                for (int step = 1; __ballot(!w.done) != 0; ++step) {
                    grid_step(D, w, limit, push);
                    if (__ballot(n > GRID_LIST - 2) != 0 || ((step & (RT_GRID_CHECK - 1)) == 0 && __ballot(n > 0) != 0)) {
                        for (int k = 0; __ballot(k < n) != 0; ++k)
                            if (k < n) exact(list[k]);
                        n = 0;
                        limit = reach();
                    }
                }
                for (int k = 0; __ballot(k < n) != 0; ++k)
                    if (k < n) exact(list[k]);
                every_sphere = false;
            } else if (!GRID && D.hier_top_groups) {
                // Bounding-sphere hierarchy: a few dozen tests instead of one per sphere.  The ray
                // is walked in windows that grow by a factor of 4: most rays of a scene full of
                // spheres end close to where they start (inside a sphere, at a neighbour), and a
                // hit inside the window ends the search - everything not yet looked at begins
                // beyond the window.  The ray starts at s_o on the line, measured from `on`.
                const double s_o = -dot(ref - o, d);
                HierWindow win;
                win.s_from = (float)(s_o - 1e-6 * fabs(s_o)) - e_ray;   // rounded down
                win.s_prev = __builtin_nanf("");
                float reach = D.hier_window;
                bool pending = active, overflowed = false;
                while (__ballot(pending) != 0) {
                    const bool last = !(reach < D.hier_extent);   // the rest of the ray
                    win.s_to = last ? __builtin_inff() : (float)(s_o + reach);
                    win.s_best = (best < __builtin_inf()) ? (float)((s_o + best) + 1e-6 * fabs(s_o + best)) + e_ray : __builtin_inff();
                    bool more;
                    const int n = hier_candidates(D, clusters, list, pending, ox, oy, oz, dx, dy, dz, e_ray, win, more);
                    for (int k = 0; __ballot(k < n) != 0; ++k)
                        if (k < n) exact(list[k]);
                    if (n < 0) { overflowed = true; pending = false; }   // the list overflowed
                    // nothing left beyond the window, or a hit before the window's end: whatever has
                    // not been looked at begins later
                    if (last || !more || (s_o + best) < (double)win.s_to - (double)e_ray - 1e-6 * fabs(s_o + best)) pending = false;
                    win.s_prev = win.s_to;
                    reach *= RT_HIER_GROWTH;
                }
                every_sphere = overflowed;
            }
            if (!GRID && every_sphere) {
                // All pairs of the table (read through the scalar unit, four pairs per step).  The
                // survivors of a lane go to its own short list first and are tested together
                // afterwards, in sphere order: lanes survive at different spheres, testing on the
                // spot would run the float64 code once per sphere that any lane of the wave keeps.
                typedef const __attribute__((address_space(4))) CullPair* ConstCull;
                ConstCull table = (ConstCull)(uintptr_t)D.cull;
                const LaneList mine = list;
                int n_mine = 0;
                auto flush = [&]() {
                    for (int k = 0; __ballot(k < n_mine) != 0; ++k)   // the longest list of the active lanes
                        if (k < n_mine) exact(mine[k]);
                    n_mine = 0;
                };
                const int np = ((S + 7) >> 3) << 2;   // the table is padded with blank entries
                for (int p = 0; p < np; p += 4) {
                    v2f t[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        CullPair cp;
#pragma unroll
                        for (int w = 0; w < 2; ++w) {
                            cp.cx[w] = table[p + k].cx[w]; cp.cy[w] = table[p + k].cy[w]; cp.cz[w] = table[p + k].cz[w];
                            cp.r[w] = table[p + k].r[w]; cp.e[w] = table[p + k].e[w];
                        }
                        t[k] = cull_test(cp, ox, oy, oz, dx, dy, dz, e_ray);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (!(t[k].x > 0) && 2 * (p + k) < S) mine[n_mine++] = (uint16_t)(2 * (p + k));
                        if (!(t[k].y > 0) && 2 * (p + k) + 1 < S) mine[n_mine++] = (uint16_t)(2 * (p + k) + 1);
                    }
                    if (__ballot(n_mine > CAND_MAX - 8) != 0) flush();   // some lane's list is nearly full
                }
                flush();
            }
        }
        if (SMALL && active) nearest_of_table(s_sph, S, o, d, d_o, best, sid);   // few spheres, all in LDS
        if (!tile_ok) return;

        // ---- hit record (reference raytrace.py:53-101) -------------------------------------
        uint32_t info = I_NO_HIT | (live && valid ? I_LIVE : 0u);
        if (active && sid >= 0) {
            double r2;
            const V3 centre = sphere_centre(sid, r2);
            const SphereRec sr = sphere_rec(sid);
            V3 P = o + best * d;
            V3 n = normalize(P - centre);
            const bool inside = dot(d, n) > 0;
            const double n1 = inside ? sr.ior : air, n2 = inside ? air : sr.ior;
            V3 N = inside ? n : -n;
            V3 Pb = P - eps * N;
            const double mu = inside ? sr.mu_in : sr.mu_out;   // n1 / n2
            const double cosi = dot(N, d);
            const bool trans = refraction_discriminant(mu, cosi) >= 0;
            // Reflected fraction and path length inside the sphere only matter for a ray whose
            // colour reaches the pixel (the others are traced for the batch and hit counts only).
            double A = 0, through = 0;
            if (live) {
                // sr.r0 = ((n1 - n2) / (n1 + n2))^2 is the same value from either side
                const Optics op = hit_optics(N, d, cosi, mu, sr.r0, n2 >= n1);
                const V3 T = op.T;
                const double floor_ = inside ? 0.0 : sr.floor_;
                A = floor_ + (sr.refl_max - floor_) * op.R;
                if (b != 0 && trans && !inside) {
                    // distance to the far side of the entered sphere (reference raytrace.py:117-118)
                    double lo, hi;
                    through = -1;
                    if (sphere_roots(Pb, T, dot(T, Pb), centre.x, centre.y, centre.z, r2, lo, hi))
                        through = hi > 0 ? hi : (lo > 0 ? lo : -1.0);
                }
            }
            info = (uint32_t)sid | (live ? I_LIVE : 0u);
            if (inside) info |= I_INSIDE;
            if (b != 0 && trans) info |= I_CHILD_T;
            if (a != 0 && b != 0 && !trans) info |= I_CHILD_I;
            if (a != 0) info |= I_CHILD_R;
            if (live) {
                // weights of reference raytrace.py:164-166; a zero weight is an exact zero there too
                if ((1 - sr.transp) * (1 - A) != 0) info |= I_SHADE;
                if (sr.transp * (1 - A) != 0) info |= I_LIVE_TI;
                if (A != 0) info |= I_LIVE_R;
            }
            for (int l = 0; l < D.n_lights; ++l) {
                if (!((D.spot_mask >> l) & 1u)) continue;
                // reference objects.py:204-213
                const RtLight& Lt = sc.lights[l];
                V3 rel = Pb - ld3(Lt.position);
                double x = dot(ld3(Lt.x_hat), rel), y = dot(ld3(Lt.y_hat), rel), z = dot(ld3(Lt.normal), rel);
                double rho = xsqrt(x * x + y * y);
                double edge = Lt.radius + Lt.tan_outer * fabs(z);
                if (rho <= edge) info |= 1u << (I_BEAM_SHIFT + l);
            }
            // the record is read by the children (hit point, direction), by k_surface (the
            // same) and by k_combine (A, path length): store only what somebody will read
            // (streaming stores: the record is read again only after the rest of the level has gone by;
            // measured -3.5 % on this kernel.  The scattered 4-byte stores of k_link and the colour
            // stores of k_combine and the lighting kernels did not gain from them.)
            if ((info & (I_CHILD_T | I_CHILD_I | I_CHILD_R | I_SHADE)) != 0) {
                stream_store(P.x, &D.px[idx]); stream_store(P.y, &D.py[idx]); stream_store(P.z, &D.pz[idx]);
                stream_store(d.x, &D.dx[idx]); stream_store(d.y, &D.dy[idx]); stream_store(d.z, &D.dz[idx]);
            }
            if (live) {
                stream_store_smooth(D.A, idx, A, D.col32);
                if (b != 0 && trans && !inside) stream_store_smooth(D.dist, idx, through, D.col32);   // k_combine reads it for these hits only
            }
        }
        if (valid) stream_store(info, &D.info[idx]);

        // ---- per-tile counters: lane c keeps counter c ---------------------------------------
        unsigned mine = 0;
#pragma unroll
        for (int c = 0; c < NCNT; ++c) {
            if (c >= C_BEAM0 && c < C_BEAM0 + RT_MAX_LIGHTS && !((D.spot_mask >> (c - C_BEAM0)) & 1u)) continue;
            const unsigned n_c = __popcll(__ballot(counted(info, c)));
            if (lane == c) mine = n_c;
        }
        if (lane < NCNT) {
            D.tile_run[(size_t)(tbase + t) * NCNT + lane] = run;
            run += mine;
        }
    };

    // Software pipeline over this wave's tiles, two tiles per iteration with two register
    // sets (A, B): while a tile is computed, the parent record of the next tile and the parent
    // index of the tile after it are in flight, and no loaded register is copied (a copy
    // would force the wave to wait for the load at once).
    const int steps = max(0, min(nt, t_first + q) - t_first);
    Meta mA = meta_of(0);
    uint32_t pA = load_link(mA);
    Meta mB = meta_of(1);
    uint32_t pB = load_link(mB);
    Parent parA = load_parent(mA, pA);
    for (int step = 0; step < steps; step += 2) {
        // explicit points where a parent index must have arrived (see below)
        asm volatile("" : "+v"(pB));
        const Parent parB = load_parent(mB, pB);
        const Meta mA2 = meta_of(step + 2);
        pA = load_link(mA2);
        process(mA, parA);
        // the parent index is first needed here: keeps the address arithmetic (and with it the
        // wait for the index load) from being scheduled ahead of the tile's work
        asm volatile("" : "+v"(pA));
        parA = load_parent(mA2, pA);
        const Meta mB2 = meta_of(step + 3);
        pB = load_link(mB2);
        if (step + 1 < steps) process(mB, parB);
        mA = mA2;
        mB = mB2;
    }

    // totals: per workgroup for k_scan, plus every wave's offset inside its workgroup
    __shared__ unsigned s_wtot[WPB][NCNT];
    if (lane < NCNT) s_wtot[wave][lane] = run;
    __syncthreads();
    if (lane < NCNT) {
        unsigned off = 0;
        for (int w = 0; w < wave; ++w) off += s_wtot[w][lane];
        D.wave_off[((size_t)level * NCNT + lane) * NW + wid] = off;
        if (wave == WPB - 1) D.blk_tot[((size_t)level * NCNT + lane) * NB + blockIdx.x] = off + run;
    }
}

// ---------------------------------------------------------------------------------------------
// Counts -> node sizes, slots and the next level's layout.  One workgroup.
__global__ __launch_bounds__(1024) void k_scan(Dev D, int level) {
    __shared__ unsigned s_grand[NCNT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = D.lvl_ntiles[level];
    const int tbase = D.lvl_tilebase[level];
    const int q = tiles_per_wave(D, nt);

    // 1. exclusive scan of the per-workgroup totals (stored counter-major), one wave per
    //    counter; every lane owns a run of consecutive entries and has all its loads in flight
    if (wave < NCNT) {
        // (D.nb workgroups ran, a multiple of 256: every lane owns D.nb / 64 entries, four to a load)
        constexpr int MAX_LOADS = NB / 64 / 4;
        const int per = D.nb / 64, loads = per / 4;
        uint32_t* base = D.blk_tot + ((size_t)level * NCNT + wave) * NB + lane * per;
        uint4 v[MAX_LOADS];
        unsigned sum = 0;
#pragma unroll
        for (int i = 0; i < MAX_LOADS; ++i) v[i] = i < loads ? reinterpret_cast<const uint4*>(base)[i] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MAX_LOADS; ++i) sum += v[i].x + v[i].y + v[i].z + v[i].w;
        unsigned incl = sum;
        for (int off = 1; off < 64; off <<= 1) {
            unsigned up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        unsigned running = incl - sum;
#pragma unroll
        for (int i = 0; i < MAX_LOADS; ++i) {
            uint4 o;
            o.x = running; running += v[i].x;
            o.y = running; running += v[i].y;
            o.z = running; running += v[i].z;
            o.w = running; running += v[i].w;
            if (i < loads) reinterpret_cast<uint4*>(base)[i] = o;
        }
        if (lane == 63) s_grand[wave] = incl;
    }
    __syncthreads();

    // 2. per-node prefix and totals: the running value of every counter at the node's first
    //    tile goes to LDS, so that a node's total is the difference to the next node's value
    const int k0 = D.level_start[level], nn = D.level_start[level + 1] - k0;
    __shared__ unsigned s_pre[(MAX_LEVEL_NODES + 1) * NCNT];
    for (int i = tid; i < nn * NCNT; i += 1024) {
        const int kk = i / NCNT, c = i % NCNT;
        const int T0 = D.node_tile0[k0 + kk];
        s_pre[i] = (T0 < nt) ? gscan(D, level, tbase, q, T0, c) : s_grand[c];
    }
    if (tid < NCNT) s_pre[nn * NCNT + tid] = s_grand[tid];
    __syncthreads();
    for (int i = tid; i < nn * NCNT; i += 1024) {
        const int kk = i / NCNT, c = i % NCNT, k = k0 + kk;
        const unsigned s = s_pre[i];
        const unsigned tot = s_pre[i + NCNT] - s;
        D.node_pre[(size_t)k * NCNT + c] = s;
        if (c < 3) {
            const int child = D.node_child[k * 3 + c];
            if (child >= 0) {
                D.node_n[child] = (int)tot;
                whole_image_batch(D, child, level + 1) = (int)tot;
            }
        } else if (c == C_HIT) {
            D.node_hits[k] = (int)tot;
        } else if (c >= C_BEAM0 && c < C_BEAM0 + RT_MAX_LIGHTS) {
            D.node_beam[k * RT_MAX_LIGHTS + (c - C_BEAM0)] = (int)tot;
        }
    }
    if (tid == 0) {
        D.lvl_shade_base[level + 1] = D.lvl_shade_base[level] + (int)s_grand[C_SHADE];
        D.work[3] += s_grand[C_REC];     // only this workgroup touches them between k_init and the host
        D.work[4] += s_grand[C_LIVE];
    }
    __syncthreads();

    // 3. layout of the next level: every node gets whole tiles (block-wide prefix over nodes)
    if (level + 1 < D.n_levels) {
        __shared__ int s_wave_tot[16];
        const int c0 = D.level_start[level + 1], c1 = D.level_start[level + 2];
        const int next_base = tbase + nt;
        const int k = c0 + tid;
        const int ntile = (k < c1) ? (D.node_n[k] + TILE - 1) / TILE : 0;
        int incl = ntile;
        for (int off = 1; off < 64; off <<= 1) {
            int up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (lane == 63) s_wave_tot[wave] = incl;
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) before += s_wave_tot[w];
            total += s_wave_tot[w];
        }
        const int running = before + incl - ntile;
        const bool overflow = (uint32_t)(next_base + total) > D.cap_tiles;
        if (k < c1) {
            if (overflow) {
                // arena too small: drop the deeper levels and tell the host
                D.node_n[k] = 0;
                whole_image_batch(D, k, level + 1) = 0;
                D.node_tile0[k] = 0;
                D.node_ntiles[k] = 0;
            } else {
                D.node_tile0[k] = running;
                D.node_ntiles[k] = ntile;
                D.node_start[k] = (uint32_t)(next_base + running) * TILE;
            }
        }
        if (tid == 0) {
            if (overflow) D.flags[0] = 1;
            D.lvl_ntiles[level + 1] = overflow ? 0 : total;
            D.lvl_tilebase[level + 1] = next_base;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Tiles a wave handles per loop iteration in the bookkeeping kernels: their loads are
// issued together so that their latencies overlap.
#ifndef RT_LINK_TILES
#define RT_LINK_TILES 4
#endif
constexpr int UNROLL_TILES = RT_LINK_TILES;

// Per-node tables of one level in (dynamic) LDS for the bookkeeping kernels: tile layout,
// first arena slot of each child node and the level-wide running value of every counter
// at the node's first tile.  LEVEL_TABLE_BYTES per node.
constexpr int LEVEL_TABLE_BYTES = 16 + 16 + 64;
struct LevelTables {
    const NodeRec* node;
    const uint32_t* child;   // [nn][4]
    const uint32_t* pre;     // [nn][16]; the shading list is level-wide, its entry is 0
};
__device__ __forceinline__ int find_node(const NodeRec* node, int nn, int t) {
    int lo = 0, hi = nn - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (node[mid].tile_end > t) hi = mid; else lo = mid + 1;
    }
    return lo;
}
// Stages the tables for the workgroup's tiles [t_first, t_last]: the tile layout of every node of
// the level (the node search needs it), child slots and counter values only of the nodes those
// tiles belong to (usually one or two of up to a few hundred; the rest of the arrays stays
// unwritten).  Contains a barrier; the caller synchronises once more before reading.
__device__ __forceinline__ LevelTables stage_level_tables(const Dev& D, int level, int* s_dyn, int tid, int t_first, int t_last) {
    const int k0 = D.level_start[level], nn = D.level_start[level + 1] - k0;
    NodeRec* node = reinterpret_cast<NodeRec*>(s_dyn);
    uint32_t* child = reinterpret_cast<uint32_t*>(node + nn);
    uint32_t* pre = child + 4 * nn;
    for (int i = tid; i < nn; i += BLOCK) {
        const int k = k0 + i;
        const int t0 = D.node_tile0[k];
        node[i] = {t0 + D.node_ntiles[k], t0, D.node_n[k], 0};
    }
    __syncthreads();
    const int n_first = find_node(node, nn, t_first), n_last = find_node(node, nn, t_last);
    const int span = n_last - n_first + 1;
    for (int j = tid; j < span * 4; j += BLOCK) {
        const int i = n_first * 4 + j, c = i & 3;
        const int ch = c < 3 ? D.node_child[(k0 + (i >> 2)) * 3 + c] : -1;
        child[i] = ch >= 0 ? D.node_start[ch] : 0u;
    }
    for (int j = tid; j < span * 16; j += BLOCK) {
        const int i = n_first * 16 + j, c = i & 15;
        pre[i] = (c < NCNT && c != C_SHADE) ? D.node_pre[(size_t)(k0 + (i >> 4)) * NCNT + c] : 0u;
    }
    return {node, child, pre};
}
static_assert(NCNT <= 16, "counter tables in LDS have 16 columns");
// Ordered compaction: every child slot of the next level learns its parent, and the
// hits that need lighting are appended to the dense shading list.
__global__ __launch_bounds__(BLOCK, RT_LINK_WAVES) void k_link(Dev D, int level) {
    extern __shared__ int s_dyn[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (D.flags[0]) return;  // arena overflow: child slots do not exist
    const int k0 = D.level_start[level], nn = D.level_start[level + 1] - k0;
    const int nt = D.lvl_ntiles[level];
    const int tbase = D.lvl_tilebase[level];
    const int q = tiles_per_wave(D, nt);
    const int wid = blockIdx.x * WPB + wave;
    const int t_begin = wid * q;
    const int t_end = min(nt, t_begin + q);
    if (blockIdx.x * WPB * q >= nt) return;   // nothing for this workgroup
    const LevelTables T = stage_level_tables(D, level, s_dyn, tid, blockIdx.x * WPB * q, min(nt, (int)(blockIdx.x + 1) * WPB * q) - 1);
    const bool has_children = level + 1 < D.n_levels;
    const int shade_base = D.lvl_shade_base[level];
    // counters whose ranks are needed: children, shade items, and what sample addressing uses
    uint32_t want = (has_children ? 7u : 0u) | (1u << C_SHADE);
    if (D.list_j[0]) want |= 1u << C_HIT;
    want |= D.spot_mask << C_BEAM0;
    // lane c: value of counter c at this wave's first tile (k_trace gave the same tiles to
    // the wave with this index)
    unsigned wave_base = 0;
    if (lane < NCNT)
        wave_base = D.blk_tot[((size_t)level * NCNT + lane) * NB + blockIdx.x] + D.wave_off[((size_t)level * NCNT + lane) * NW + wid];
    __syncthreads();
    int kk = (t_begin < t_end) ? find_node(T.node, nn, t_begin) : 0;
    struct Loaded {                      // what a group of UNROLL_TILES tiles needs from memory
        int kn[UNROLL_TILES];            // node of the tile, relative to the level
        uint32_t info[UNROLL_TILES];
        unsigned trun[UNROLL_TILES];     // lane c: counter c at the tile's start, relative to the wave's
    };
    // requests the info words and counters of the tiles t0 .. t0 + UNROLL_TILES - 1
    auto fetch = [&](int t0, Loaded& g) {
#pragma unroll
        for (int u = 0; u < UNROLL_TILES; ++u) {
            const int t = t0 + u;
            g.info[u] = I_NO_HIT;
            g.kn[u] = 0;
            g.trun[u] = 0;
            if (t < t_end) {
                while (T.node[kk].tile_end <= t) ++kk;
                kk = __builtin_amdgcn_readfirstlane(kk);
                g.kn[u] = kk;
                const NodeRec r = T.node[kk];
                const int first = (t - __builtin_amdgcn_readfirstlane(r.tile0)) * TILE;
                const int cnt = min(TILE, __builtin_amdgcn_readfirstlane(r.n) - first);
                if (lane < cnt) g.info[u] = D.info[(uint32_t)(tbase + t) * TILE + lane];
                if (lane < NCNT) g.trun[u] = D.tile_run[(size_t)(tbase + t) * NCNT + lane];
            }
        }
    };
    // ranks and stores of one group
    auto emit = [&](int t0, const Loaded& g) {
#pragma unroll
        for (int u = 0; u < UNROLL_TILES; ++u) {
            const int t = t0 + u;
            if (t >= t_end) break;
            const uint32_t idx = (uint32_t)(tbase + t) * TILE + lane;
            const uint32_t in = g.info[u];
            // lane c: rank offset of counter c inside the node (inside the level for the shading list)
            const int base = (int)(wave_base + g.trun[u] - T.pre[g.kn[u] * 16 + (lane & 15)]);
            int rank[NCNT];
#pragma unroll
            for (int c = 0; c < NCNT; ++c) {
                rank[c] = 0;
                if ((want >> c) & 1u) rank[c] = lanes_below(__ballot(counted(in, c))) + LANE(base, c);
            }
            if (has_children) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if ((in >> (18 + c)) & 1u) D.link[T.child[g.kn[u] * 4 + c] + rank[c]] = idx;
                }
            }
            if (in & I_SHADE) {
                const int e = shade_base + rank[C_SHADE];
                D.list_ray[e] = idx;
                D.list_node[e] = (uint16_t)(k0 + g.kn[u]);
                if (D.list_j[0]) D.list_j[0][e] = (uint32_t)rank[C_HIT];
#pragma unroll
                for (int l = 0; l < RT_MAX_LIGHTS; ++l)
                    if ((D.spot_mask >> l) & 1u) D.list_j[1 + l][e] = (uint32_t)rank[C_BEAM0 + l];
            }
        }
    };
    // Two groups in turn, each with its own registers: the loads of the next group are requested
    // before the current group's ranks and stores, and no register with a load in flight is copied.
    Loaded ga, gb;
    fetch(t_begin, ga);
    for (int t0 = t_begin; t0 < t_end; t0 += 2 * UNROLL_TILES) {
        fetch(t0 + UNROLL_TILES, gb);
        emit(t0, ga);
        fetch(t0 + 2 * UNROLL_TILES, ga);
        emit(t0 + UNROLL_TILES, gb);
    }
}

}  // namespace rt
