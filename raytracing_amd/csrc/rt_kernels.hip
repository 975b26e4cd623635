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
#ifndef RT_MIXED_WAVES
#define RT_MIXED_WAVES 2   // waves per SIMD requested for the mixed-precision surface kernel
#endif

namespace rt {

struct V3 {
    double x, y, z;
};
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 ld3(const double* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ double clip01(double v) { return fmin(fmax(v, 0.0), 1.0); }
__device__ __forceinline__ V3 clip01(V3 v) { return {clip01(v.x), clip01(v.y), clip01(v.z)}; }
// ---- correctly rounded float64 sqrt and division without the range scaling --------------
// The compiler's expansions of sqrt() and '/' wrap the Newton steps below in scaling code for
// operands near the ends of the exponent range (v_div_scale / v_div_fmas / v_div_fixup,
// v_ldexp), most of it quarter-rate instructions.  For operands well inside the range the
// scaling does nothing, so the bare steps give bit-identical (correctly rounded) results;
// anything else takes the compiler's path.
__device__ __forceinline__ bool mid_range(double x) {   // 2^-400 <= |x| < 2^400
    return (((uint32_t)__double2hiint(x) >> 20) & 0x7ffu) - 623u < 800u;
}
__device__ __forceinline__ double xsqrt(double x) {
    if (__builtin_expect(!(mid_range(x) || x == 0), 0)) return sqrt(x);
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    g = fma(fma(-g, g, x), h, g);
    g = fma(fma(-g, g, x), h, g);
    return x == 0 ? x : g;
}
// 1 / b to within the accuracy the final quotient step needs (two Newton steps).
__device__ __forceinline__ double refined_rcp(double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = fma(r, fma(-b, r, 1.0), r);
    r = fma(r, fma(-b, r, 1.0), r);
    return r;
}
// a / b given r = refined_rcp(b); a and b inside the middle of the exponent range (or a == 0).
__device__ __forceinline__ double quotient(double a, double b, double r) {
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}
__device__ __forceinline__ double xdiv(double a, double b) {
    if (__builtin_expect(!(mid_range(b) && (mid_range(a) || a == 0)), 0)) return a / b;
    return quotient(a, b, refined_rcp(b));
}
// v / |v| with the norm as sqrt of the left-to-right sum of squares and three
// divisions, the way numpy.linalg.norm + array division round.
__device__ __forceinline__ V3 normalize(V3 v) {
    const double len = xsqrt(dot(v, v));
    const bool fast = mid_range(len) && (mid_range(v.x) || v.x == 0) && (mid_range(v.y) || v.y == 0) &&
                      (mid_range(v.z) || v.z == 0);
    if (__builtin_expect(!fast, 0)) return {v.x / len, v.y / len, v.z / len};
    const double r = refined_rcp(len);
    return {quotient(v.x, len, r), quotient(v.y, len, r), quotient(v.z, len, r)};
}

// Pairs of floats: arithmetic on them compiles to the packed float32 instructions
// (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32), two results per lane and instruction.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f vfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f vfma(v2f a, float b, v2f c) { return __builtin_elementwise_fma(a, (v2f)(b), c); }
__device__ __forceinline__ v2f vfma(v2f a, float b, float c) { return __builtin_elementwise_fma(a, (v2f)(b), (v2f)(c)); }

// ---- float32 screening of ray-sphere pairs ------------------------------------------------
// With many spheres nearly every (ray, sphere) pair is a clear miss.  The distance of the
// sphere's centre from the ray's line is first evaluated in float32 for two spheres at a time:
// oc = c - o, b = oc.d, p = oc - b d, and the pair is a certain miss when |p| > r + E.  E bounds
// the error of the computed |p|: with u = 2^-24 and N = |o|_1 + |c|_1 (which also bounds |oc| and
// |b|), each component of oc is off by at most 2 u N (rounding of o, c and of the difference), b
// by 6.5 u N, b d_k by 7.5 u N, the fma that forms p_k adds 2 u N: 11.5 u N per component,
// 20 u N in length; the table uses 1.5e-6 N (25 u N) and a radius enlarged by 2e-6 r for the
// roundings of |p|^2, of the sum and of the comparison.  The reference's own float64 rounding of
// its discriminant corresponds to 1e-16 N^2 / r in this length, far inside E.  Only pairs that
// are not certain misses take the float64 test, which alone decides.
struct CullPair {   // two spheres, component-wise
    float cx[2], cy[2], cz[2], r[2], e[2];
};
__host__ __device__ __forceinline__ void cull_entry(CullPair& p, int k, const double* g) {
    p.cx[k] = (float)g[0]; p.cy[k] = (float)g[1]; p.cz[k] = (float)g[2];
    p.r[k] = (float)(fabs(g[3]) * 1.000002);
    p.e[k] = 1.5e-6f * (float)((fabs(g[0]) + fabs(g[1]) + fabs(g[2])) * 1.000001);
}
__host__ __device__ __forceinline__ void cull_blank(CullPair& p, int k) {   // never passes the screening
    p.cx[k] = p.cy[k] = p.cz[k] = 0.0f;
    p.cx[k] = 1.0e15f;   // far away; the caller also ignores indices beyond the table's end
    p.r[k] = 0.0f;
    p.e[k] = 0.0f;
}
__device__ __forceinline__ float cull_ray_margin(V3 o) {
    return 1.5e-6f * (float)((fabs(o.x) + fabs(o.y) + fabs(o.z)) * 1.000001);
}
// Positive: the ray's line certainly misses the sphere.  One value per sphere of the pair.
__device__ __forceinline__ v2f cull_test(const CullPair& p, float ox, float oy, float oz, float dx, float dy, float dz,
                                         float e_ray) {
    const v2f ocx = (v2f){p.cx[0], p.cx[1]} - ox, ocy = (v2f){p.cy[0], p.cy[1]} - oy, ocz = (v2f){p.cz[0], p.cz[1]} - oz;
    const v2f b = vfma(ocz, dz, vfma(ocy, dy, ocx * dx));
    const v2f px = vfma(-b, dx, ocx), py = vfma(-b, dy, ocy), pz = vfma(-b, dz, ocz);
    const v2f p2 = vfma(pz, pz, vfma(py, py, px * px));
    const v2f t = (v2f){p.r[0], p.r[1]} + ((v2f){p.e[0], p.e[1]} + e_ray);
    return vfma(-t, t, p2);
}

constexpr int CAND_MAX = 16;   // survivors a lane collects before the wave tests them
// Number of set bits of a wave ballot below the calling lane.
__device__ __forceinline__ int lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

// First node of the level whose tile range ends after tile t.
__device__ __forceinline__ int find_node(const int* tile_end, int nn, int t) {
    int lo = 0, hi = nn - 1;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (tile_end[mid] > t) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// Level-wide running value of counter c at the start of tile t (relative tile index).
__device__ __forceinline__ unsigned gscan(const Dev& D, int level, int tbase, int q, int t, int c) {
    const int w = t / q;   // the wave that produced the tile
    return D.blk_tot[((size_t)level * NCNT + c) * NB + w / WPB] + D.wave_off[((size_t)level * NCNT + c) * NW + w] +
           D.tile_run[(size_t)(tbase + t) * NCNT + c];
}

// The facts about a node that the tile loops need are fetched with ONE vector load per tile:
// lane i of the wave reads table i below for node k, and wave-uniform values are then taken
// with readlane.  (Many small scalar loads per tile are much slower: the scalar unit can only
// wait for all of its outstanding loads at once.)
enum { NL_TILE0 = 0, NL_N = 1, NL_NGLOBAL = 2, NL_A = 3, NL_B = 4, NL_BRANCH = 5, NL_CHILD0 = 6, NL_COUNT = 9 };
struct NodeLane {
    const int32_t* ptr;
    int mul, add;
};
__device__ __forceinline__ NodeLane node_lane(const Dev& D, int lane) {
    switch (lane) {
        case NL_TILE0: return {D.node_tile0, 1, 0};
        case NL_N: return {D.node_n, 1, 0};
        case NL_NGLOBAL: return {D.node_nglobal, 1, 0};
        case NL_A: return {D.node_a, 1, 0};
        case NL_B: return {D.node_b, 1, 0};
        case NL_BRANCH: return {D.node_branch, 1, 0};
        case NL_CHILD0: return {D.node_child, 3, 0};
        case NL_CHILD0 + 1: return {D.node_child, 3, 1};
        case NL_CHILD0 + 2: return {D.node_child, 3, 2};
        default: return {D.node_tile0, 0, 0};
    }
}
__device__ __forceinline__ int node_fetch(const NodeLane& nl, int k, int lane) {
    return lane < NL_COUNT ? nl.ptr[k * nl.mul + nl.add] : 0;
}
#define LANE(v, i) __builtin_amdgcn_readlane((v), (i))

// Lane c: level-wide running value of counter c at the start of tile t, relative to the
// first tile of node k (not for the shade counter, whose list is level-wide).
__device__ __forceinline__ int counter_base(const Dev& D, int level, int tbase, int q, int t, int k, int lane) {
    if (lane >= NCNT) return 0;
    const int w = t / q;
    int v = (int)(D.blk_tot[((size_t)level * NCNT + lane) * NB + w / WPB] + D.wave_off[((size_t)level * NCNT + lane) * NW + w] +
                  D.tile_run[(size_t)(tbase + t) * NCNT + lane]);
    if (lane != C_SHADE) v -= (int)D.node_pre[(size_t)k * NCNT + lane];
    return v;
}

__device__ __forceinline__ bool is_hit(uint32_t info) { return (info & I_SID_MASK) != I_NO_HIT; }

// Whether a ray with this info word is counted by counter c.
__device__ __forceinline__ bool counted(uint32_t info, int c) {
    if (c == C_HIT) return is_hit(info);
    if (c == C_REC) return is_hit(info) && (info & (I_CHILD_T | I_CHILD_I | I_CHILD_R | I_SHADE)) != 0;
    if (c == C_LIVE) return is_hit(info) && (info & I_LIVE) != 0;
    const int bit = c < 3 ? 18 + c : (c == C_SHADE ? 21 : I_BEAM_SHIFT + (c - C_BEAM0));
    return (info >> bit) & 1u;
}

// Roots of |o + t d - c|^2 = r^2 for unit d; returns false if there is none; lo <= hi.
//
// Every hit / miss decision of the frame hangs on the sign of this discriminant, and
// scenes do contain rays that are exactly tangent to a sphere, so the arithmetic
// follows reference objects.py:60-66 and :242-251 operation by operation:
//   B = 2 (sum(d*o) - dot(d, c)),  C = sum((c-o)^2) - r^2,  disc = B^2 - 4C,
//   roots -B/2 +- sqrt(disc)/2.
// sum() adds left to right; dot(d, c) is the BLAS product of the NumPy build the
// golden images were made with, which rounds as fma(d2, c2, fma(d0, c0, d1*c1)).
// d_o is sum(d*o), the same for every sphere.
__device__ __forceinline__ bool sphere_roots(V3 o, V3 d, double d_o, double cx, double cy, double cz, double r2,
                                             double& lo, double& hi) {
    const double d_c = fma(d.z, cz, fma(d.x, cx, d.y * cy));
    const double B = 2 * (d_o - d_c);
    const double fx = cx - o.x, fy = cy - o.y, fz = cz - o.z;
    const double C = (fx * fx + fy * fy + fz * fz) - r2;
    const double disc = B * B - 4 * C;
    if (!(0 <= disc)) return false;
    const double half_root = xsqrt(disc) / 2;
    const double mid = -B / 2;
    const double x1 = mid + half_root, x2 = mid - half_root;
    lo = fmin(x1, x2);
    hi = fmax(x1, x2);
    return true;
}

// What is recomputed from a stored hit record (hit point P, sphere, inside flag)
// with exactly the operations k_trace used when it made the record
// (reference raytrace.py:60-62, :86-89).
struct Surface {
    V3 n;   // outward unit normal
    V3 N;   // normal facing along the ray
    V3 Pb;  // hit point pulled back by epsilon to the side the ray came from
};
__device__ __forceinline__ Surface surface_at(V3 P, V3 centre, bool inside, double eps) {
    Surface s;
    s.n = normalize(P - centre);
    s.N = inside ? s.n : -s.n;
    s.Pb = P - eps * s.N;
    return s;
}

// ---------------------------------------------------------------------------------------------
__global__ void k_init(Dev D) {
    const int tid = threadIdx.x;
    for (int k = tid; k < D.K; k += blockDim.x) {
        D.node_n[k] = 0;
        D.node_nglobal[k] = 0;
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
        if ((uint32_t)nt > D.cap_tiles) {
            D.flags[0] = 1;
            nt = 0;
        }
        D.node_n[0] = nt ? D.npix : 0;
        D.node_nglobal[0] = D.node_n[0];
        D.node_ntiles[0] = nt;
        D.lvl_ntiles[0] = nt;
    }
}

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

template <bool SMALL>   // SMALL: at most SPH_CHUNK_SMALL spheres, all constants in LDS
__global__ __launch_bounds__(BLOCK, RT_TRACE_WAVES) void k_trace(Dev D, int level) {
    constexpr int SPH_CHUNK = SMALL ? SPH_CHUNK_SMALL : 1;
    __shared__ NodeRec s_node[MAX_LEVEL_NODES];
    __shared__ double s_sph[SPH_CHUNK][4];
    __shared__ SphereRec s_rec[SMALL ? SPH_CHUNK_SMALL : 1];
    __shared__ uint16_t s_cand[SMALL ? 1 : WPB][SMALL ? 1 : 64][CAND_MAX];   // per lane: spheres the screening kept
    __shared__ SceneConst s_sc;
    static_assert(sizeof(SceneConst) % 8 == 0, "SceneConst is copied in 8-byte words");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = D.level_start[level], nn = D.level_start[level + 1] - k0;
    const int nt = D.lvl_ntiles[level];
    const int tbase = D.lvl_tilebase[level];
    const int q = (nt + NW - 1) / NW;                  // tiles per wave (same for every wave)
    const int wid = blockIdx.x * WPB + wave;
    const int t_first = wid * q;
    const int S = D.S;
    const bool block_has_work = blockIdx.x * WPB * q < nt;

    for (int i = tid; i < (int)(sizeof(SceneConst) / 8); i += BLOCK)
        reinterpret_cast<double*>(&s_sc)[i] = reinterpret_cast<const double*>(D.sc)[i];
    if (block_has_work) {
        const int small_num = D.sc->small_num;
        for (int i = tid; i < nn; i += BLOCK) {
            const int k = k0 + i;
            const int t0 = D.node_tile0[k];
            NodeRec r;
            r.tile0 = t0;
            r.tile_end = t0 + D.node_ntiles[k];
            r.n = D.node_n[k];
            r.bits = (D.node_a[k] & 0xff) | ((D.node_b[k] & 0xff) << 8) | ((D.node_branch[k] & 0xff) << 16) |
                     ((D.node_nglobal[k] < small_num) ? (1 << 24) : 0);   // reference raytrace.py:26-27
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
                double sx = X * sc.scr_w / sc.pix_x - sc.scr_w / 2;
                double sy = (sc.pix_y - Y) * sc.scr_h / sc.pix_y - sc.scr_h / 2;
                V3 pix = sx * ld3(sc.scr_x) + sy * ld3(sc.scr_y) + ld3(sc.scr_pos);
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
                    d = pd - cosi * (2.0 * sf.N);
                    o = sf.Pb;
                    live = (br == 1) ? (pinfo & I_LIVE_TI) : (pinfo & I_LIVE_R);
                }
            }
        }

        // ---- nearest hit (reference objects.py:288-299) ------------------------------------
        const double d_o = dot(d, o);
        double best = __builtin_inf();
        int sid = -1;
        if (!SMALL && active) {
            // Many spheres: float32 screening of all pairs (table read through the scalar unit,
            // four pairs per step), the float64 test only for the pairs it cannot rule out.
            // The screening looks at the ray's whole line, so any point of the line can stand in
            // for the origin: the one nearest to the middle of the spheres keeps the float32
            // coordinates (and with them the error bound) small for rays that start far away.
            typedef const __attribute__((address_space(4))) CullPair* ConstCull;
            ConstCull table = (ConstCull)(uintptr_t)D.cull;
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
                    // ties go to the later sphere, as the reference's min_t == t update does
                    if (tt > 0 && tt <= best) { best = tt; sid = i; }
                }
            };
            // The survivors of a lane go to its own short list first and are tested together
            // afterwards, in sphere order: lanes survive at different spheres, testing on the
            // spot would run the float64 code once per sphere that any lane of the wave keeps.
            uint16_t* mine = s_cand[wave][lane];
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
        {
            const int c0 = 0, cn = S;
            if (SMALL && active) {
                // two spheres per iteration: two independent dependency chains in flight
                int i = 0;
                for (; i + 1 < cn; i += 2) {
                    double lo0, hi0, lo1, hi1;
                    const bool h0 = sphere_roots(o, d, d_o, s_sph[i][0], s_sph[i][1], s_sph[i][2], s_sph[i][3], lo0, hi0);
                    const bool h1 = sphere_roots(o, d, d_o, s_sph[i + 1][0], s_sph[i + 1][1], s_sph[i + 1][2], s_sph[i + 1][3], lo1, hi1);
                    // ties go to the later sphere, as the reference's min_t == t update does
                    if (h0) {
                        const double tt = lo0 > 0 ? lo0 : hi0;
                        if (tt > 0 && tt <= best) { best = tt; sid = c0 + i; }
                    }
                    if (h1) {
                        const double tt = lo1 > 0 ? lo1 : hi1;
                        if (tt > 0 && tt <= best) { best = tt; sid = c0 + i + 1; }
                    }
                }
                if (i < cn) {
                    double lo, hi;
                    if (sphere_roots(o, d, d_o, s_sph[i][0], s_sph[i][1], s_sph[i][2], s_sph[i][3], lo, hi)) {
                        const double tt = lo > 0 ? lo : hi;
                        if (tt > 0 && tt <= best) { best = tt; sid = c0 + i; }
                    }
                }
            }
        }
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
            const double kq = 1 - mu * mu * (1 - cosi * cosi);
            const bool trans = kq >= 0;
            // Reflected fraction and path length inside the sphere only matter for a ray whose
            // colour reaches the pixel (the others are traced for the batch and hit counts only).
            double A = 0, through = 0;
            if (live) {
                V3 T = {0, 0, 0};
                if (trans) T = xsqrt(kq) * N + mu * (d - cosi * N);  // reference raytrace.py:207-211
                const double r0 = sr.r0;   // ((n1 - n2) / (n1 + n2))^2, the same value from either side
                const double cosf = (n2 >= n1) ? cosi : dot(T, N);
                const double om = 1 - cosf;
                const double om2 = om * om;
                double R = r0 + (1 - r0) * (om2 * om2 * om);
                if (!trans) R = 1;
                const double floor_ = inside ? 0.0 : sr.floor_;
                A = floor_ + (sr.refl_max - floor_) * R;
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
            if ((info & (I_CHILD_T | I_CHILD_I | I_CHILD_R | I_SHADE)) != 0) {
                D.px[idx] = P.x; D.py[idx] = P.y; D.pz[idx] = P.z;
                D.dx[idx] = d.x; D.dy[idx] = d.y; D.dz[idx] = d.z;
            }
            if (live) {
                D.A[idx] = A;
                D.dist[idx] = through;
            }
        }
        if (valid) D.info[idx] = info;

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
    const int q = max(1, (nt + NW - 1) / NW);

    // 1. exclusive scan of the per-workgroup totals (stored counter-major), one wave per
    //    counter; every lane owns a run of consecutive entries and has all its loads in flight
    if (wave < NCNT) {
        constexpr int PER = NB / 64;
        uint32_t* base = D.blk_tot + ((size_t)level * NCNT + wave) * NB + lane * PER;
        uint4 v[PER / 4];
        unsigned sum = 0;
#pragma unroll
        for (int i = 0; i < PER / 4; ++i) v[i] = reinterpret_cast<const uint4*>(base)[i];
#pragma unroll
        for (int i = 0; i < PER / 4; ++i) sum += v[i].x + v[i].y + v[i].z + v[i].w;
        unsigned incl = sum;
        for (int off = 1; off < 64; off <<= 1) {
            unsigned up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        unsigned running = incl - sum;
#pragma unroll
        for (int i = 0; i < PER / 4; ++i) {
            uint4 o;
            o.x = running; running += v[i].x;
            o.y = running; running += v[i].y;
            o.z = running; running += v[i].z;
            o.w = running; running += v[i].w;
            reinterpret_cast<uint4*>(base)[i] = o;
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
                D.node_nglobal[child] = (int)tot;
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
                D.node_nglobal[k] = 0;
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
constexpr int UNROLL_TILES = 4;

// Per-node tables of one level in (dynamic) LDS for the bookkeeping kernels: tile layout,
// first arena slot of each child node and the level-wide running value of every counter
// at the node's first tile.  LEVEL_TABLE_BYTES per node.
constexpr int LEVEL_TABLE_BYTES = 16 + 16 + 64;
struct LevelTables {
    const NodeRec* node;
    const uint32_t* child;   // [nn][4]
    const uint32_t* pre;     // [nn][16]; the shading list is level-wide, its entry is 0
};
__device__ __forceinline__ LevelTables stage_level_tables(const Dev& D, int level, int* s_dyn, int tid) {
    const int k0 = D.level_start[level], nn = D.level_start[level + 1] - k0;
    NodeRec* node = reinterpret_cast<NodeRec*>(s_dyn);
    uint32_t* child = reinterpret_cast<uint32_t*>(node + nn);
    uint32_t* pre = child + 4 * nn;
    for (int i = tid; i < nn; i += BLOCK) {
        const int k = k0 + i;
        const int t0 = D.node_tile0[k];
        node[i] = {t0 + D.node_ntiles[k], t0, D.node_n[k], 0};
    }
    for (int i = tid; i < nn * 4; i += BLOCK) {
        const int c = i & 3;
        const int ch = c < 3 ? D.node_child[(k0 + (i >> 2)) * 3 + c] : -1;
        child[i] = ch >= 0 ? D.node_start[ch] : 0u;
    }
    for (int i = tid; i < nn * 16; i += BLOCK) {
        const int c = i & 15;
        pre[i] = (c < NCNT && c != C_SHADE) ? D.node_pre[(size_t)(k0 + (i >> 4)) * NCNT + c] : 0u;
    }
    return {node, child, pre};
}
static_assert(NCNT <= 16, "counter tables in LDS have 16 columns");
__device__ __forceinline__ int find_node(const NodeRec* node, int nn, int t) {
    int lo = 0, hi = nn - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (node[mid].tile_end > t) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// Ordered compaction: every child slot of the next level learns its parent, and the
// hits that need lighting are appended to the dense shading list.
__global__ __launch_bounds__(BLOCK) void k_link(Dev D, int level) {
    extern __shared__ int s_dyn[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (D.flags[0]) return;  // arena overflow: child slots do not exist
    const int k0 = D.level_start[level], nn = D.level_start[level + 1] - k0;
    const int nt = D.lvl_ntiles[level];
    const int tbase = D.lvl_tilebase[level];
    const int q = (nt + NW - 1) / NW;
    const int wid = blockIdx.x * WPB + wave;
    const int t_begin = wid * q;
    const int t_end = min(nt, t_begin + q);
    if (blockIdx.x * WPB * q >= nt) return;   // nothing for this workgroup
    const LevelTables T = stage_level_tables(D, level, s_dyn, tid);
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
    for (int t0 = t_begin; t0 < t_end; t0 += UNROLL_TILES) {
        int kn[UNROLL_TILES];            // node of the tile, relative to the level
        uint32_t info[UNROLL_TILES];
        unsigned trun[UNROLL_TILES];     // lane c: counter c at the tile's start, relative to the wave's
        // phase A: the loads of all tiles of this iteration
#pragma unroll
        for (int u = 0; u < UNROLL_TILES; ++u) {
            const int t = t0 + u;
            info[u] = I_NO_HIT;
            kn[u] = 0;
            trun[u] = 0;
            if (t < t_end) {
                while (T.node[kk].tile_end <= t) ++kk;
                kk = __builtin_amdgcn_readfirstlane(kk);
                kn[u] = kk;
                const NodeRec r = T.node[kk];
                const int first = (t - __builtin_amdgcn_readfirstlane(r.tile0)) * TILE;
                const int cnt = min(TILE, __builtin_amdgcn_readfirstlane(r.n) - first);
                if (lane < cnt) info[u] = D.info[(uint32_t)(tbase + t) * TILE + lane];
                if (lane < NCNT) trun[u] = D.tile_run[(size_t)(tbase + t) * NCNT + lane];
            }
        }
        // phase B: ranks and stores
#pragma unroll
        for (int u = 0; u < UNROLL_TILES; ++u) {
            const int t = t0 + u;
            if (t >= t_end) break;
            const uint32_t idx = (uint32_t)(tbase + t) * TILE + lane;
            // lane c: rank offset of counter c inside the node (inside the level for the shading list)
            const int base = (int)(wave_base + trun[u] - T.pre[kn[u] * 16 + (lane & 15)]);
            int rank[NCNT];
#pragma unroll
            for (int c = 0; c < NCNT; ++c) {
                rank[c] = 0;
                if ((want >> c) & 1u) rank[c] = lanes_below(__ballot(counted(info[u], c))) + LANE(base, c);
            }
            if (has_children) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if ((info[u] >> (18 + c)) & 1u) D.link[T.child[kn[u] * 4 + c] + rank[c]] = idx;
                }
            }
            if (info[u] & I_SHADE) {
                const int e = shade_base + rank[C_SHADE];
                D.list_ray[e] = idx;
                D.list_node[e] = (uint16_t)(k0 + kn[u]);
                if (D.list_j[0]) D.list_j[0][e] = (uint32_t)rank[C_HIT];
#pragma unroll
                for (int l = 0; l < RT_MAX_LIGHTS; ++l)
                    if ((D.spot_mask >> l) & 1u) D.list_j[1 + l][e] = (uint32_t)rank[C_BEAM0 + l];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Shading helpers.

struct SurfaceMaterial {  // what the surface colour needs of a sphere's material
    double ambient[3], kdiff[3], kspec[3], shininess;
};
__device__ __forceinline__ SurfaceMaterial surface_material(const double* m) {
    return {{m[0], m[1], m[2]}, {m[3], m[4], m[5]}, {m[6], m[7], m[8]}, m[9]};
}
struct ShadowSphere {  // what a shadow ray needs of a sphere
    double cx, cy, cz, r2, tr, kr, kg, kb;  // k = attenuation coefficient * absorption colour
};

__device__ __forceinline__ double pow_shininess(double x, double sh) {
    const int n = (int)sh;
    if ((double)n == sh && n >= 0 && n <= 65536) {
        double r = 1.0, p = x;
        int e = n;
        while (e) {
            if (e & 1) r *= p;
            p *= p;
            e >>= 1;
        }
        return r;
    }
    return pow(x, sh);
}

// RGB transmittance along a shadow ray (reference objects.py:274-285): every sphere
// with a positive far root multiplies by transparency * exp(-k * chord), whether or
// not it lies beyond the light.
template <bool LDS>
__device__ __forceinline__ V3 shadow_multiplier(const Dev& D, const ShadowSphere* s_sh, uint16_t* cand, V3 o, V3 d) {
    V3 mult = {1, 1, 1};
    const int S = D.S;
    const double d_o = dot(d, o);
    if (!LDS) {
        // many spheres: float32 screening first (see cull_test), the float64 test for the
        // survivors of this lane, in sphere order
        typedef const __attribute__((address_space(4))) CullPair* ConstCull;   // uniform reads through the scalar unit
        ConstCull table = (ConstCull)(uintptr_t)D.cull;
        const V3 ref = {D.cull_ref[0], D.cull_ref[1], D.cull_ref[2]};
        const V3 on = o + dot(ref - o, d) * d;
        const float ox = (float)on.x, oy = (float)on.y, oz = (float)on.z;
        const float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
        const float e_ray = cull_ray_margin(on);
        int n_mine = 0;
        bool opaque = false;
        auto flush = [&]() {
            for (int k = 0; __ballot(k < n_mine) != 0; ++k) {   // the longest list of the active lanes
                if (k < n_mine && !opaque) {
                    const int i = cand[k];
                    const double* g = D.geom + (size_t)i * 4;
                    double lo, hi;
                    if (sphere_roots(o, d, d_o, g[0], g[1], g[2], g[3] * g[3], lo, hi) && hi > 0) {
                        const double* m = D.mat + (size_t)i * RT_MAT_WIDTH;
                        const double tr = m[11];
                        if (tr == 0) {
                            opaque = true;
                        } else {
                            const double chord = lo > 0 ? hi - lo : 0.0;
                            mult.x *= tr * exp(-(m[14] * m[15]) * chord);
                            mult.y *= tr * exp(-(m[14] * m[16]) * chord);
                            mult.z *= tr * exp(-(m[14] * m[17]) * chord);
                        }
                    }
                }
            }
            n_mine = 0;
        };
        const int np = ((S + 7) >> 3) << 2;
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
                if (!(t[k].x > 0) && 2 * (p + k) < S) cand[n_mine++] = (uint16_t)(2 * (p + k));
                if (!(t[k].y > 0) && 2 * (p + k) + 1 < S) cand[n_mine++] = (uint16_t)(2 * (p + k) + 1);
            }
            if (__ballot(n_mine > CAND_MAX - 8) != 0) flush();
        }
        flush();
        if (opaque) return {0, 0, 0};
        return mult;
    }
    for (int i = 0; i < S; ++i) {
        ShadowSphere sp;
        if (LDS) {
            sp = s_sh[i];
        } else {
            const double* g = D.geom + (size_t)i * 4;
            const double* m = D.mat + (size_t)i * RT_MAT_WIDTH;
            sp = {g[0], g[1], g[2], g[3] * g[3], m[11], m[14] * m[15], m[14] * m[16], m[14] * m[17]};
        }
        double lo, hi;
        if (sphere_roots(o, d, d_o, sp.cx, sp.cy, sp.cz, sp.r2, lo, hi) && hi > 0) {
            if (sp.tr == 0) return {0, 0, 0};
            const double chord = lo > 0 ? hi - lo : 0.0;
            mult.x *= sp.tr * exp(-sp.kr * chord);
            mult.y *= sp.tr * exp(-sp.kg * chord);
            mult.z *= sp.tr * exp(-sp.kb * chord);
        }
    }
    return mult;
}

// Beam fall-off factor of a spotlight (reference objects.py:176-191 and :302-338).
__device__ __forceinline__ double ease(int mode, double x, double a, double d) {
    if (d == 0) return 1.0;
    switch (mode) {
        case RT_EASE_LINEAR:
            return fmin(fmax((a - x) / d + 1, 0.0), 1.0);
        case RT_EASE_QUADRATIC:
            if (x < a) return 1.0;
            if (x > a + d) return 0.0;
            return (x - a + d) * (1 / d - (x - a) / (d * d));
        case RT_EASE_CUBIC: {
            if (x < a) return 1.0;
            if (x > a + d) return 0.0;
            const double q = (4 * a + 3 * d) / (d * d * d);
            const double p = (1 - q * d * d) / (d * d * (d + 2 * a));
            return 1 - (x - a) * (p * (x * x - (a + d) * (a + d)) + q * (x - (a + d)) + 1 / d);
        }
        case RT_EASE_EXPONENTIAL:
            return 1 - 1 / (1 + exp(-10 / d * (x - a - d / 2)));
        default:
            return 1.0;
    }
}

// One disk-light sample in float64 (reference objects.py:132-138, :99-110 and
// raytrace.py:219-232): light intensities and BRDF terms of the sample.
struct SampleTerms {
    V3 Id, Is, kd, ks;
};
template <bool LDS>
__device__ SampleTerms disk_sample_f64(const Dev& D, const ShadowSphere* s_sh, uint16_t* cand, const RtLight& Lt, double u_theta,
                                       double u_rad, V3 Pb, V3 d, V3 n, V3 kdiff, V3 kspec, double shin) {
    const V3 lpos = ld3(Lt.position), xh = ld3(Lt.x_hat), yh = ld3(Lt.y_hat);
    const double theta = u_theta * 6.283185307179586;
    const double rad = sqrt(u_rad) * Lt.radius;
    double sn, cs;
    sincos(theta, &sn, &cs);
    V3 lp = lpos + rad * (cs * xh + sn * yh);
    V3 Lv = lp - Pb;
    const double dist = sqrt(dot(Lv, Lv));
    V3 Lh = {Lv.x / dist, Lv.y / dist, Lv.z / dist};
    V3 mult = shadow_multiplier<LDS>(D, s_sh, cand, Pb, Lh);
    const double fall = (Lt.intensity / Lt.n_points) / (dist * dist);
    SampleTerms o;
    o.Id = clip01((fall * ld3(Lt.diffuse_color)) * mult);
    o.Is = clip01((fall * ld3(Lt.specular_color)) * mult);
    const double nl = dot(n, Lh);
    V3 Rv = Lh - (2 * nl) * n;
    const double rd = fabs(dot(Rv, d));
    o.kd = clip01(nl * kdiff);
    o.ks = clip01(pow_shininess(rd, shin) * kspec);
    return o;
}

// ---- float32 evaluation of disk-light samples --------------------------------------------
struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 f3(V3 v) { return {(float)v.x, (float)v.y, (float)v.z}; }
__device__ __forceinline__ V3 v3(F3 v) { return {(double)v.x, (double)v.y, (double)v.z}; }
__device__ __forceinline__ float fdot(F3 a, F3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ float fclip01(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f); }
__device__ __forceinline__ F3 fclip01(F3 v) { return {fclip01(v.x), fclip01(v.y), fclip01(v.z)}; }
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

constexpr int SREG = 4;  // spheres whose hit-relative data the float32 path keeps in registers

// What a shadow ray from one hit needs of one sphere, relative to the hit: computed once
// per hit in float64, then rounded.  With |d| = 1 the ray hits iff b^2 - C >= 0 and one
// root is positive, b = oc.d, C = |oc|^2 - r^2.
struct HitSphere {
    F3 oc;      // Pb - centre
    float C;    // |oc|^2 - r^2, sign exact (origin inside the sphere iff C < 0)
    float len;  // |oc|
    float k0;   // error bound of C (float rounding + float64 cancellation noise)
};

// All samples of one disk light for one hit in float32, with every shadow decision
// certified: the discriminant of each (sample, sphere) test carries a rigorous error
// bound.  A sample with a decision inside its bound is only counted here (n_unsure); the
// caller then queues the whole hit for re-evaluation in float64 (k_surface_redo), so the
// set of occluders that reaches the image always equals the float64 path's.  Samples are
// fetched eight at a time so that their loads are in flight together.
constexpr int SAMPLE_BATCH = 8;
// Two samples per lane at a time: the straight-line float arithmetic runs on packed
// instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32), only the transcendental
// functions, the clamps and the decisions are per sample.
__device__ __forceinline__ v2f vclip01(v2f a) { return {fclip01(a.x), fclip01(a.y)}; }
struct F3x2 {
    v2f x, y, z;
};
__device__ __forceinline__ v2f vdot(F3 a, F3x2 b) { return vfma(b.z, a.z, vfma(b.y, a.y, b.x * a.x)); }

template <bool LDS>
__device__ void disk_light_mixed(const Dev& D, const ShadowSphere* s_sh, const RtLight& Lt, const float* pk,
                                 int64_t stride, V3 Pb, V3 d, V3 n, V3 kdiff, V3 kspec,
                                 double shin, V3& sumId, V3& sumIs, V3& sumKd, V3& sumKs, unsigned& n_unsure) {
    const int S = D.S;
    HitSphere hs[SREG];
#pragma unroll
    for (int i = 0; i < SREG; ++i) {
        if (i < S) {
            const ShadowSphere sp = LDS ? s_sh[i] : ShadowSphere{D.geom[i * 4], D.geom[i * 4 + 1], D.geom[i * 4 + 2],
                                                                   D.geom[i * 4 + 3] * D.geom[i * 4 + 3], 0, 0, 0, 0};
            const V3 oc = {Pb.x - sp.cx, Pb.y - sp.cy, Pb.z - sp.cz};
            const double l2 = dot(oc, oc);
            const double C = l2 - sp.r2;
            hs[i].oc = f3(oc);
            hs[i].C = (float)C;
            hs[i].len = (float)sqrt(l2) * 1.0000002f;
            hs[i].k0 = (float)(1.5e-7 * fabs(C) + 1e-15 * (l2 + sp.r2));
        } else {
            hs[i].oc = {0, 0, 0}; hs[i].C = 1; hs[i].len = 0; hs[i].k0 = 0;
        }
    }
    const F3 q = f3(ld3(Lt.position) - Pb);
    const F3 xh = f3(ld3(Lt.x_hat)), yh = f3(ld3(Lt.y_hat));
    const F3 nf = f3(n), df = f3(d), kdf = f3(kdiff), ksf = f3(kspec);
    const F3 dcol = f3(ld3(Lt.diffuse_color)), scol = f3(ld3(Lt.specular_color));
    const float Rf = (float)Lt.radius, per_sample = (float)(Lt.intensity / Lt.n_points), shf = (float)shin;
    const float frame = sqrtf(fmaxf(fdot(xh, xh), fdot(yh, yh)));  // the disk frame is not normalised
    const float eL_scale = 1.5e-6f * Rf * frame;
    const int N = Lt.n_points;
    F3x2 aId = {0, 0, 0}, aIs = {0, 0, 0}, aKd = {0, 0, 0}, aKs = {0, 0, 0};
    // Transmittance of the occluders of one sample (reference objects.py:274-285) as one
    // exponential per channel: the product over spheres of tr * exp(-k * chord) is
    // exp2(sum of log2(tr) - k * log2(e) * chord), chord = 2 sqrt(disc) for an origin outside
    // the sphere and 0 inside.  The per-sphere constants are the same in every lane.
    auto uniform = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
    float lt[SREG], k2r[SREG], k2g[SREG], k2b[SREG];
#pragma unroll
    for (int i = 0; i < SREG; ++i) {
        lt[i] = k2r[i] = k2g[i] = k2b[i] = 0;
        if (i < S) {
            const ShadowSphere sp = LDS ? s_sh[i]
                                        : ShadowSphere{0, 0, 0, 0, D.mat[i * RT_MAT_WIDTH + 11],
                                                       D.mat[i * RT_MAT_WIDTH + 14] * D.mat[i * RT_MAT_WIDTH + 15],
                                                       D.mat[i * RT_MAT_WIDTH + 14] * D.mat[i * RT_MAT_WIDTH + 16],
                                                       D.mat[i * RT_MAT_WIDTH + 14] * D.mat[i * RT_MAT_WIDTH + 17]};
            lt[i] = uniform(__builtin_amdgcn_logf((float)sp.tr));           // log2; -inf for an opaque sphere
            k2r[i] = uniform(-2.0f * 1.4426950408889634f * (float)sp.kr);
            k2g[i] = uniform(-2.0f * 1.4426950408889634f * (float)sp.kg);
            k2b[i] = uniform(-2.0f * 1.4426950408889634f * (float)sp.kb);
        }
    }
    auto occlude = [&](int i, float disc, F3& e) {
        const float sq = hs[i].C > 0 ? __builtin_amdgcn_sqrtf(disc) : 0.0f;
        e.x = fmaf(k2r[i], sq, e.x + lt[i]);
        e.y = fmaf(k2g[i], sq, e.y + lt[i]);
        e.z = fmaf(k2b[i], sq, e.z + lt[i]);
    };
    for (int s0 = 0; s0 < N; s0 += SAMPLE_BATCH) {
        float2 u[SAMPLE_BATCH];
#pragma unroll
        for (int k = 0; k < SAMPLE_BATCH; ++k)
            u[k] = *reinterpret_cast<const float2*>(pk + (int64_t)min(s0 + k, N - 1) * stride);
#pragma unroll
        for (int k = 0; k < SAMPLE_BATCH; k += 2) {
            if (s0 + k >= N) break;
            const bool both = s0 + k + 1 < N;    // with an odd sample count the last pair is half empty
            const v2f ut = {u[k].x, u[k + 1].x}, ur = {u[k].y, u[k + 1].y};
            // angle in revolutions
            const v2f sn = {__builtin_amdgcn_sinf(ut.x), __builtin_amdgcn_sinf(ut.y)};
            const v2f cs = {__builtin_amdgcn_cosf(ut.x), __builtin_amdgcn_cosf(ut.y)};
            const v2f rad = (v2f){__builtin_amdgcn_sqrtf(ur.x), __builtin_amdgcn_sqrtf(ur.y)} * Rf;
            const F3x2 Lv = {vfma(rad, vfma(cs, xh.x, sn * yh.x), (v2f)(q.x)), vfma(rad, vfma(cs, xh.y, sn * yh.y), (v2f)(q.y)),
                             vfma(rad, vfma(cs, xh.z, sn * yh.z), (v2f)(q.z))};
            const v2f d2 = vfma(Lv.z, Lv.z, vfma(Lv.y, Lv.y, Lv.x * Lv.x));
            const v2f inv = {__builtin_amdgcn_rsqf(d2.x), __builtin_amdgcn_rsqf(d2.y)};
            const F3x2 Lh = {Lv.x * inv, Lv.y * inv, Lv.z * inv};
            // bound of |Lh - exact unit light vector|: light point (sin/cos 1.3e-7, u -> float
            // 3.8e-7 rad, sqrt 1e-7, all times the disk radius) over the distance, plus rounding
            const v2f eL = vfma(inv, eL_scale, 6e-7f);
            F3 e0 = {0, 0, 0}, e1 = {0, 0, 0};   // log2 of the transmittance
            bool unsure0 = false, unsure1 = false, hit0 = false, hit1 = false;
#pragma unroll
            for (int i = 0; i < SREG; ++i) {
                if (i < S) {
                    const v2f b = vdot(hs[i].oc, Lh);
                    const v2f disc = vfma(b, b, (v2f)(-hs[i].C));
                    const v2f db = eL * hs[i].len;
                    const v2f ab2 = (v2f){fabsf(b.x), fabsf(b.y)} * 2.0f;
                    const v2f m = vfma(ab2, db, vfma(db, db, (v2f)(hs[i].k0)));
                    if (fabsf(disc.x) <= m.x) unsure0 = true;
                    else if (disc.x > 0 && (b.x < 0 || hs[i].C < 0)) { occlude(i, disc.x, e0); hit0 = true; }
                    if (fabsf(disc.y) <= m.y) unsure1 = true;
                    else if (disc.y > 0 && (b.y < 0 || hs[i].C < 0)) { occlude(i, disc.y, e1); hit1 = true; }
                }
            }
            if (unsure0) n_unsure += 1;
            if (unsure1 && both) n_unsure += 1;
            F3 mult0 = {1, 1, 1}, mult1 = {1, 1, 1};
            if (hit0) mult0 = {__builtin_amdgcn_exp2f(e0.x), __builtin_amdgcn_exp2f(e0.y), __builtin_amdgcn_exp2f(e0.z)};
            if (hit1) mult1 = {__builtin_amdgcn_exp2f(e1.x), __builtin_amdgcn_exp2f(e1.y), __builtin_amdgcn_exp2f(e1.z)};
            const F3x2 mult = {{mult0.x, mult1.x}, {mult0.y, mult1.y}, {mult0.z, mult1.z}};
            const v2f fall = inv * inv * per_sample;
            const v2f nl = vdot(nf, Lh);
            const v2f two_nl = nl * 2.0f;
            const F3x2 Rv = {vfma(-two_nl, nf.x, Lh.x), vfma(-two_nl, nf.y, Lh.y), vfma(-two_nl, nf.z, Lh.z)};
            const v2f rdv = vdot(df, Rv);
            v2f p = {1.0f, 1.0f};
            if (shf != 0)
                p = (v2f){__builtin_amdgcn_exp2f(shf * __builtin_amdgcn_logf(fabsf(rdv.x))),
                          __builtin_amdgcn_exp2f(shf * __builtin_amdgcn_logf(fabsf(rdv.y)))};
            F3x2 cId = {vclip01(fall * dcol.x * mult.x), vclip01(fall * dcol.y * mult.y), vclip01(fall * dcol.z * mult.z)};
            F3x2 cIs = {vclip01(fall * scol.x * mult.x), vclip01(fall * scol.y * mult.y), vclip01(fall * scol.z * mult.z)};
            F3x2 cKd = {vclip01(nl * kdf.x), vclip01(nl * kdf.y), vclip01(nl * kdf.z)};
            F3x2 cKs = {vclip01(p * ksf.x), vclip01(p * ksf.y), vclip01(p * ksf.z)};
            if (!both) {
                const v2f first = {1.0f, 0.0f};
                cId.x *= first; cId.y *= first; cId.z *= first; cIs.x *= first; cIs.y *= first; cIs.z *= first;
                cKd.x *= first; cKd.y *= first; cKd.z *= first; cKs.x *= first; cKs.y *= first; cKs.z *= first;
            }
            aId.x += cId.x; aId.y += cId.y; aId.z += cId.z;
            aIs.x += cIs.x; aIs.y += cIs.y; aIs.z += cIs.z;
            aKd.x += cKd.x; aKd.y += cKd.y; aKd.z += cKd.z;
            aKs.x += cKs.x; aKs.y += cKs.y; aKs.z += cKs.z;
            __builtin_amdgcn_sched_barrier(0);   // one pair at a time: interleaving pairs costs registers
        }
    }
    auto total = [](F3x2 v) -> V3 { return {(double)(v.x.x + v.x.y), (double)(v.y.x + v.y.y), (double)(v.z.x + v.z.y)}; };
    sumId = total(aId); sumIs = total(aIs); sumKd = total(aKd); sumKs = total(aKs);
}

// Surface colour of one hit under all lights (reference raytrace.py:136-152,
// :215-234 and objects.py:94-235).  entry: index of the hit in the shading list,
// which is also the index of its disk-light samples in the packed sample buffer.
template <bool LDS, bool MIXED>
__device__ V3 surface_color(const Dev& D, const SceneConst& sc, const ShadowSphere* s_sh, uint16_t* cand, int64_t entry, uint32_t info,
                            V3 Pb, V3 d, V3 n, const SurfaceMaterial& mat, unsigned& n_samples, unsigned& n_fallback) {
    const V3 kdiff = ld3(mat.kdiff), kspec = ld3(mat.kspec);
    const double shin = mat.shininess;
    V3 surf = {0, 0, 0};
    for (int l = 0; l < D.n_lights; ++l) {
        const RtLight& Lt = sc.lights[l];
        const V3 lpos = ld3(Lt.position);
        if (Lt.kind == RT_LIGHT_POINT) {
            const V3 dcol = ld3(Lt.diffuse_color), scol = ld3(Lt.specular_color);
            V3 Lv = lpos - Pb;
            const double d2 = dot(Lv, Lv);
            const double dist = sqrt(d2);
            V3 Lh = {Lv.x / dist, Lv.y / dist, Lv.z / dist};
            V3 mult = shadow_multiplier<LDS>(D, s_sh, cand, Pb, Lh);
            const double fall = Lt.intensity / (dist * dist);
            V3 Id = clip01((fall * dcol) * mult), Is = clip01((fall * scol) * mult);
            const double nl = dot(n, Lh);
            V3 Rv = Lh - (2 * nl) * n;
            const double rd = fabs(dot(Rv, d));
            V3 kd = clip01(nl * kdiff), ks = clip01(pow_shininess(rd, shin) * kspec);
            surf = surf + clip01(clip01(kd * Id) + clip01(ks * Is));
            n_samples += 1;
            continue;
        }
        // disk light, possibly restricted to a beam
        double factor = 1.0;
        if (Lt.kind == RT_LIGHT_SPOT) {
            if (!((info >> (I_BEAM_SHIFT + l)) & 1u)) continue;  // zero light vectors, zero intensity
            V3 rel = Pb - lpos;
            const double x = dot(ld3(Lt.x_hat), rel), y = dot(ld3(Lt.y_hat), rel), z = dot(ld3(Lt.normal), rel);
            const double rho = sqrt(x * x + y * y);
            const double allowed = Lt.radius + Lt.tan_angle * fabs(z);
            const double edge = Lt.radius + Lt.tan_outer * fabs(z);
            factor = ease(Lt.easing, rho, allowed, edge - allowed);
        }
        const int N = Lt.n_points;
        n_samples += N;
        const double* pk = D.packed + 2 * (D.pack_base[l] + entry);   // + 2 * pack_E per sample
        const int64_t stride = 2 * D.pack_E;
        V3 sumId = {0, 0, 0}, sumIs = {0, 0, 0}, sumKd = {0, 0, 0}, sumKs = {0, 0, 0};
        if constexpr (MIXED) {   // the host launches this variant only when S <= SREG
            disk_light_mixed<LDS>(D, s_sh, Lt, D.packed32 + 2 * (D.pack_base[l] + entry), stride, Pb, d, n, kdiff, kspec, shin, sumId, sumIs, sumKd, sumKs,
                                  n_fallback);
        } else {
            for (int s = 0; s < N; ++s) {
                const SampleTerms o = disk_sample_f64<LDS>(D, s_sh, cand, Lt, pk[s * stride], pk[s * stride + 1],
                                                           Pb, d, n, kdiff, kspec, shin);
                sumId = sumId + o.Id; sumIs = sumIs + o.Is; sumKd = sumKd + o.kd; sumKs = sumKs + o.ks;
            }
        }
        const double invN = 1.0 / N;
        V3 Id = clip01(invN * sumId), Is = clip01(invN * sumIs);
        if (Lt.kind == RT_LIGHT_SPOT) {
            Id = clip01(factor * Id);
            Is = clip01(factor * Is);
        }
        surf = surf + clip01(sumKd * Id + sumKs * Is);
    }
    surf = surf + clip01(ld3(mat.ambient) * ld3(sc.ambient));
    return clip01(surf);
}

// ---------------------------------------------------------------------------------------------
// Sample packing.  The caller's random stream is laid out as the reference consumes it:
// the block of (node, light) holds all angle uniforms and then all radius uniforms, each
// as n_points rows of h values (h = the node's hits, or in-beam hits for a spotlight),
// and the hit with rank j owns column j (reference objects.py:122, :132-133).  Only the
// hits on the shading list ever read their columns, so every uploaded piece of the
// stream, chunk[0..n) = stream[a..a+n), is scattered straight into the packed layout
// (entry-major within a sample, so k_surface reads it fully coalesced).
// Source of stream values for k_pack: doubles the caller uploaded ...
struct UploadedStream {
    static constexpr bool WHOLE_IMAGE = false;   // positions are in this context's own stream
    const double* chunk;   // chunk[i] = stream[a + i]
    __device__ __forceinline__ double operator()(int64_t i) const { return chunk[i]; }
};
// ... or raw MT19937 state words made on the device: value i of the piece is built from words
// 2i and 2i+1 exactly as NumPy's legacy random_sample does (53-bit resolution:
// (a >> 5) * 2^26 + (b >> 6), over 2^53, a and b tempered outputs).
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
struct GeneratedStream {
    static constexpr bool WHOLE_IMAGE = true;    // positions are in the whole image's stream when the layout says so
    const uint32_t* raw;   // raw[2 i], raw[2 i + 1]: untempered words of stream[a + i]
    __device__ __forceinline__ double operator()(int64_t i) const {
        const uint32_t hi = mt_temper(raw[2 * i]) >> 5, lo = mt_temper(raw[2 * i + 1]) >> 6;
        return ((double)hi * 67108864.0 + (double)lo) / 9007199254740992.0;
    }
};

template <typename Stream>
__global__ __launch_bounds__(BLOCK) void k_pack(Dev D, Stream chunk, int64_t a, int64_t n) {
    const SceneConst& sc = *D.sc;
    const int64_t b = a + n;
    const int64_t E = D.pack_E;
    for (int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x; e < E; e += (int64_t)gridDim.x * BLOCK) {
        const uint32_t info = D.info[D.list_ray[e]];
        const int node = D.list_node[e];
        for (int l = 0; l < D.n_lights; ++l) {
            const RtLight& Lt = sc.lights[l];
            if (Lt.kind == RT_LIGHT_POINT) continue;
            int64_t h, jj;
            if (Lt.kind == RT_LIGHT_SPOT) {
                if (!((info >> (I_BEAM_SHIFT + l)) & 1u)) continue;
                h = D.node_beam[node * RT_MAX_LIGHTS + l];
                jj = D.list_j[1 + l][e];
            } else {
                h = D.node_hits[node];
                jj = D.list_j[0][e];
            }
            const int N = Lt.n_points;
            int64_t off = D.sample_off[(size_t)node * RT_MAX_LIGHTS + l];
            if (off < 0) continue;
            if (Stream::WHOLE_IMAGE && D.sample_global) {
                // this context holds columns [first column, + h) of a block that is wider in the
                // whole image's stream
                const size_t KL = (size_t)D.K * RT_MAX_LIGHTS, at = (size_t)node * RT_MAX_LIGHTS + l;
                off = D.sample_global[at];
                h = D.sample_global[KL + at];
                jj += D.sample_global[2 * KL + at];
            }
            if (off + 2 * N * h <= a || off >= b) continue;
            double* dst = D.packed + 2 * (D.pack_base[l] + e);
            float* dst32 = D.packed32 ? D.packed32 + 2 * (D.pack_base[l] + e) : nullptr;
            for (int part = 0; part < 2; ++part) {
                const int64_t first = off + (int64_t)part * N * h + jj;     // stream position of row 0
                int64_t s0 = first >= a ? 0 : (a - first + h - 1) / h;
                for (int64_t s = s0; s < N; ++s) {
                    const int64_t pos = first + s * h;
                    if (pos >= b) break;
                    const double u = chunk(pos - a);
                    dst[2 * s * E + part] = u;
                    if (dst32) dst32[2 * s * E + part] = (float)u;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// MT19937 state sequence on the device (the generator behind NumPy's legacy global stream,
// numpy.random.random / RandomState).  x[n + 624] = x[n + 397] ^ twist(x[n], x[n + 1]).
// One workgroup: lane j of 227 computes x[g + j], g advancing by 227 per step - its own
// previous result is x[g + j - 227] = the x[n + 397] term, the twist operands are 624 back in
// an LDS ring and were written at least two steps earlier, so one barrier per two steps is
// enough (the ring of 2048 keeps writes clear of anything still being read).
//   ring[2048]  x[n] at slot n & 2047 for the last 2048 indices below *gen_end (in and out)
//   raw[i]      receives x[raw_first + i] for 0 <= i < raw_n (untempered)
//   generates until *gen_end >= max(raw_first + raw_n, gen_until)
//
// Streams: the word sequence is cut into streams of L = 2^ell words; stream t covers the
// absolute word indices [t L, (t+1) L] (one word of overlap, a value may be built from the
// last word of one stream and the first of the next) and has its own ring and counter.
// Stream 0 starts from the caller's key block, the others from windows k_mt_jump made.
constexpr int MT_N = 624, MT_M = 397, MT_STEP = MT_N - MT_M, MT_RING = 2048;
__device__ __forceinline__ uint32_t mt_twist(uint32_t lo, uint32_t hi) {
    const uint32_t y = (lo & 0x80000000u) | (hi & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
// grid: one workgroup per stream t = t0 + blockIdx.x.
//   raw[w - raw_origin] receives x[w] for the stream's words below w_end
//   the stream with index last_stream keeps generating until its counter reaches gen_until
__global__ __launch_bounds__(256) void k_mt_generate(uint32_t* rings, long long* gen_ends, uint32_t* raw, int t0, int ell,
                                                     long long raw_origin, long long w_end, int last_stream,
                                                     long long gen_until) {
    __shared__ uint32_t ring[MT_RING];
    const int tid = threadIdx.x;
    const int t = t0 + blockIdx.x;
    uint32_t* ring_g = rings + (size_t)t * MT_RING;
    for (int i = tid; i < MT_RING; i += 256) ring[i] = ring_g[i];
    const long long g0 = gen_ends[t];
    const long long raw_first = (long long)t << ell;
    long long raw_end = ((long long)(t + 1) << ell) + 1;
    if (raw_end > w_end) raw_end = w_end;
    if (!raw || raw_end < raw_first) raw_end = raw_first;
    const long long raw_n = raw_end - raw_first;
    long long target = raw_n > 0 ? raw_end : g0;
    if (t == last_stream && gen_until > target) target = gen_until;
    uint32_t* out = raw ? raw + (raw_first - raw_origin) : nullptr;
    __syncthreads();
    // words of the window that exist already
    for (long long n = raw_first + tid; n < raw_end && n < g0; n += 256) out[n - raw_first] = ring[n & (MT_RING - 1)];
    // two steps per iteration (all four twist operands were written before the previous
    // barrier); positions as 32-bit offsets from g0
    const long long todo = target > g0 ? target - g0 : 0;
    const int iters = (int)((todo + 2 * MT_STEP - 1) / (2 * MT_STEP));
    const bool lane_on = tid < MT_STEP;
    const int base = (int)(g0 & (MT_RING - 1));
    uint32_t prev = lane_on ? ring[(base + tid - MT_STEP) & (MT_RING - 1)] : 0u;
    // raw index of this lane's word in step 0, relative to the window (may be negative)
    long long rel = g0 + tid - raw_first;
    int slot = (base + tid) & (MT_RING - 1);
    for (int it = 0; it < iters; ++it) {
        if (lane_on) {
            const int s0 = slot, s1 = (slot + MT_STEP) & (MT_RING - 1);
            const uint32_t lo0 = ring[(s0 - MT_N) & (MT_RING - 1)], hi0 = ring[(s0 - MT_N + 1) & (MT_RING - 1)];
            const uint32_t lo1 = ring[(s1 - MT_N) & (MT_RING - 1)], hi1 = ring[(s1 - MT_N + 1) & (MT_RING - 1)];
            const uint32_t x0 = prev ^ mt_twist(lo0, hi0);
            const uint32_t x1 = x0 ^ mt_twist(lo1, hi1);
            ring[s0] = x0;
            ring[s1] = x1;
            if (rel >= 0 && rel < raw_n) out[rel] = x0;
            if (rel + MT_STEP >= 0 && rel + MT_STEP < raw_n) out[rel + MT_STEP] = x1;
            prev = x1;
        }
        slot = (slot + 2 * MT_STEP) & (MT_RING - 1);
        rel += 2 * MT_STEP;
        __syncthreads();
    }
    for (int i = tid; i < MT_RING; i += 256) ring_g[i] = ring[i];
    if (tid == 0) gen_ends[t] = g0 + (long long)iters * 2 * MT_STEP;
}

// Jump-ahead.  Workgroup blockIdx.x makes the start window of stream t = blockIdx.x + 1:
// the window at word index t L - 1, reached from the base window (index L - 1, taken from
// ring `base_ring` whose counter is base_end) by applying, for every set bit b of t - 1, the
// polynomial x^(L 2^b) mod phi:  new window word j = XOR over set coefficients i of s[i + j],
// s the sequence generated from the current window.  The low 31 bits of the new window's
// first word are not meaningful (nothing reads them: that word belongs to the previous
// stream).  polys: [n_polys][624] coefficient words, polys[b] = x^(L 2^b) mod phi.
constexpr int MTJ_RING = 4096, MTJ_CHUNK = 8 * MT_STEP;   // coefficients consumed per round
__global__ __launch_bounds__(640) void k_mt_jump(uint32_t* rings, long long* gen_ends, int base_ring, long long base_end,
                                                 const uint32_t* polys, int n_polys, int ell) {
    __shared__ uint32_t xs[MTJ_RING];     // s[i] at slot i & 4095
    __shared__ uint32_t coef[MT_N];
    __shared__ uint32_t win[MT_N];
    const int tid = threadIdx.x;
    const int t = blockIdx.x + 1;
    const long long L = 1ll << ell;
    // base window = x[L-1 .. L+622]
    for (int j = tid; j < MT_N; j += 640) win[j] = rings[(size_t)base_ring * MT_RING + ((L - 1 + j) & (MT_RING - 1))];
    (void)base_end;
    __syncthreads();
    for (int b = 0; b < n_polys; ++b) {
        if (!(((unsigned)(t - 1) >> b) & 1u)) continue;      // uniform
        for (int j = tid; j < MT_N; j += 640) {
            xs[j] = win[j];
            coef[j] = polys[(size_t)b * MT_N + j];
        }
        __syncthreads();
        uint32_t acc = 0;
        uint32_t prev = (tid < MT_STEP) ? xs[MT_N - MT_STEP + tid] : 0u;   // own-chain value s[gen - 227 + tid]
        int gen = MT_N;                                                     // s[0 .. gen) exist
        for (int c0 = 0; c0 < 19937; c0 += MTJ_CHUNK) {
            // make s available up to c0 + MTJ_CHUNK + 623
            while (gen < c0 + MTJ_CHUNK + MT_N) {
                if (tid < MT_STEP) {
                    const int n0 = gen + tid, n1 = n0 + MT_STEP;
                    const uint32_t x0 = prev ^ mt_twist(xs[(n0 - MT_N) & (MTJ_RING - 1)], xs[(n0 - MT_N + 1) & (MTJ_RING - 1)]);
                    const uint32_t x1 = x0 ^ mt_twist(xs[(n1 - MT_N) & (MTJ_RING - 1)], xs[(n1 - MT_N + 1) & (MTJ_RING - 1)]);
                    xs[n0 & (MTJ_RING - 1)] = x0;
                    xs[n1 & (MTJ_RING - 1)] = x1;
                    prev = x1;
                }
                gen += 2 * MT_STEP;
                __syncthreads();
            }
            // accumulate the coefficients c0 .. c0 + MTJ_CHUNK
            if (tid < MT_N) {
                const int c1 = min(c0 + MTJ_CHUNK, 19937);
                for (int w = c0 >> 5; w <= (c1 - 1) >> 5; ++w) {
                    uint32_t bits = __builtin_amdgcn_readfirstlane(coef[w]);
                    // keep only coefficients inside [c0, c1)
                    const int lo = w << 5;
                    if (lo < c0) bits &= ~0u << (c0 - lo);
                    if (lo + 32 > c1) bits &= ~0u >> (lo + 32 - c1);
                    while (bits) {
                        const int i = lo + __builtin_ctz(bits);
                        bits &= bits - 1;
                        acc ^= xs[(i + tid) & (MTJ_RING - 1)];
                    }
                }
            }
            __syncthreads();   // the next round overwrites ring slots this round read
        }
        if (tid < MT_N) win[tid] = acc;
        __syncthreads();
    }
    // the stream's ring: window word j is x[t L - 1 + j]
    const long long first = (long long)t * L - 1;
    for (int j = tid; j < MT_N; j += 640) rings[(size_t)t * MT_RING + ((first + j) & (MT_RING - 1))] = win[j];
    if (tid == 0) gen_ends[t] = first + MT_N;
}

// ---------------------------------------------------------------------------------------------
// Lighting for the dense list of hits whose surface colour matters (all levels).
template <bool LDS, bool MIXED>
__global__ __launch_bounds__(BLOCK, MIXED ? RT_MIXED_WAVES : 2) void k_surface(Dev D) {
    constexpr int TABLE = MIXED ? SREG : (LDS ? SHADE_LDS_SPHERES : 1);   // the float32 pass is for S <= SREG
    __shared__ ShadowSphere s_sh[TABLE];
    __shared__ SurfaceMaterial s_mat[TABLE];
    __shared__ double s_centre[TABLE][3];
    __shared__ SceneConst s_sc;
    __shared__ unsigned s_work[3];
    __shared__ uint16_t s_cand[LDS ? 1 : BLOCK][CAND_MAX];   // per lane: spheres the shadow screening kept
    const int tid = threadIdx.x;
    if (D.flags[0]) return;  // arena overflow: the frame is invalid, the host reports it
    if (tid < 3) s_work[tid] = 0;
    for (int i = tid; i < (int)(sizeof(SceneConst) / 8); i += BLOCK)
        reinterpret_cast<double*>(&s_sc)[i] = reinterpret_cast<const double*>(D.sc)[i];
    if (LDS) {
        for (int i = tid; i < D.S; i += BLOCK) {
            const double* g = D.geom + (size_t)i * 4;
            const double* m = D.mat + (size_t)i * RT_MAT_WIDTH;
            s_sh[i] = {g[0], g[1], g[2], g[3] * g[3], m[11], m[14] * m[15], m[14] * m[16], m[14] * m[17]};
            s_mat[i] = surface_material(m);
            s_centre[i][0] = g[0]; s_centre[i][1] = g[1]; s_centre[i][2] = g[2];
        }
    }
    __syncthreads();
    const SceneConst& sc = s_sc;
    const double eps = sc.eps;
    const int total = D.lvl_shade_base[D.n_levels];
    const int stride = gridDim.x * BLOCK;
    unsigned my_hits = 0, my_samples = 0, my_fallback = 0;
    int e = blockIdx.x * BLOCK + tid;
    uint32_t ray = e < total ? D.list_ray[e] : 0u;
    while (e < total) {
        // the next entry's ray index is in flight while this one is shaded
        const int e_next = e + stride;
        const uint32_t ray_next = e_next < total ? D.list_ray[e_next] : 0u;
        const uint32_t info = D.info[ray];
        V3 P = {D.px[ray], D.py[ray], D.pz[ray]};
        V3 d = {D.dx[ray], D.dy[ray], D.dz[ray]};
        const int sid = info & I_SID_MASK;
        const V3 centre = LDS ? V3{s_centre[sid][0], s_centre[sid][1], s_centre[sid][2]} : ld3(D.geom + (size_t)sid * 4);
        const SurfaceMaterial mat = LDS ? s_mat[sid] : surface_material(D.mat + (size_t)sid * RT_MAT_WIDTH);
        const Surface sf = surface_at(P, centre, info & I_INSIDE, eps);
        unsigned unsure = 0;
        V3 surf = surface_color<LDS, MIXED>(D, sc, s_sh, LDS ? nullptr : s_cand[tid], e, info, sf.Pb, d, sf.n, mat, my_samples, unsure);
        my_hits += 1;
        D.cr[ray] = surf.x; D.cg[ray] = surf.y; D.cb[ray] = surf.z;
        if (MIXED && unsure) {
            // some shadow decision of this hit could not be certified in float32
            my_fallback += unsure;
            D.redo_list[atomicAdd(&D.flags[1], 1)] = (uint32_t)e;
        }
        e = e_next;
        ray = ray_next;
    }
    if (my_hits) {
        atomicAdd(&s_work[0], my_hits);
        atomicAdd(&s_work[1], my_samples);
        if (my_fallback) atomicAdd(&s_work[2], my_fallback);
    }
    __syncthreads();
    if (tid < 3 && s_work[tid]) atomicAdd(&D.work[tid], (unsigned long long)s_work[tid]);
}

// Float64 re-evaluation of the hits the mixed-precision pass queued.
template <bool LDS>
__global__ __launch_bounds__(BLOCK) void k_surface_redo(Dev D) {
    __shared__ ShadowSphere s_sh[LDS ? SHADE_LDS_SPHERES : 1];
    __shared__ uint16_t s_cand[LDS ? 1 : BLOCK][CAND_MAX];
    const int tid = threadIdx.x;
    if (D.flags[0]) return;
    const int total = D.flags[1];
    if (total == 0) return;
    if (LDS) {
        for (int i = tid; i < D.S; i += BLOCK) {
            const double* g = D.geom + (size_t)i * 4;
            const double* m = D.mat + (size_t)i * RT_MAT_WIDTH;
            s_sh[i] = {g[0], g[1], g[2], g[3] * g[3], m[11], m[14] * m[15], m[14] * m[16], m[14] * m[17]};
        }
    }
    __syncthreads();
    const SceneConst& sc = *D.sc;
    const double eps = sc.eps;
    for (int i = blockIdx.x * BLOCK + tid; i < total; i += gridDim.x * BLOCK) {
        const int64_t e = D.redo_list[i];
        const uint32_t ray = D.list_ray[e];
        const uint32_t info = D.info[ray];
        const int sid = info & I_SID_MASK;
        V3 P = {D.px[ray], D.py[ray], D.pz[ray]};
        V3 d = {D.dx[ray], D.dy[ray], D.dz[ray]};
        const Surface sf = surface_at(P, ld3(D.geom + (size_t)sid * 4), info & I_INSIDE, eps);
        unsigned a = 0, b = 0;
        V3 surf = surface_color<LDS, false>(D, sc, s_sh, LDS ? nullptr : s_cand[tid], e, info, sf.Pb, d, sf.n,
                                            surface_material(D.mat + (size_t)sid * RT_MAT_WIDTH), a, b);
        D.cr[ray] = surf.x; D.cg[ray] = surf.y; D.cb[ray] = surf.z;
    }
}

// ---------------------------------------------------------------------------------------------
// Bottom-up colour combination for one level (reference raytrace.py:155-170); the
// deepest level runs first.  A ray's colour replaces its surface colour in place.
// Two tiles per loop iteration; per iteration the loads come in two rounds (the info words
// and counters, then everything the info words select), each round issued together.
constexpr int COMBINE_TILES = 2;
struct CombineSphere {
    double tr, kr, kg, kb;   // transparency, -attenuation * absorption colour
};
template <bool SMALL>   // SMALL: at most SPH_CHUNK_SMALL spheres, their constants in LDS
__global__ __launch_bounds__(BLOCK) void k_combine(Dev D, int level) {
    extern __shared__ int s_dyn[];
    __shared__ CombineSphere s_cs[SMALL ? SPH_CHUNK_SMALL : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (D.flags[0]) return;
    const int k0 = D.level_start[level], nn = D.level_start[level + 1] - k0;
    const int nt = D.lvl_ntiles[level];
    const int tbase = D.lvl_tilebase[level];
    const int q = (nt + NW - 1) / NW;
    const int wid = blockIdx.x * WPB + wave;
    const int t_begin = wid * q;
    const int t_end = min(nt, t_begin + q);
    if (blockIdx.x * WPB * q >= nt) return;   // nothing for this workgroup
    const LevelTables T = stage_level_tables(D, level, s_dyn, tid);
    auto sphere_consts = [&](int i) -> CombineSphere {
        const double* m = D.mat + (size_t)i * RT_MAT_WIDTH;
        return {m[11], -m[14] * m[15], -m[14] * m[16], -m[14] * m[17]};
    };
    if (SMALL)
        for (int i = tid; i < D.S; i += BLOCK) s_cs[i] = sphere_consts(i);
    const V3 bg = ld3(D.sc->bg);
    const bool has_children = level + 1 < D.n_levels;
    unsigned wave_base = 0;   // lane c < 3: children of kind c before this wave's first tile
    if (lane < 3)
        wave_base = D.blk_tot[((size_t)level * NCNT + lane) * NB + blockIdx.x] + D.wave_off[((size_t)level * NCNT + lane) * NW + wid];
    __syncthreads();

    int kk = (t_begin < t_end) ? find_node(T.node, nn, t_begin) : 0;
    for (int t0 = t_begin; t0 < t_end; t0 += COMBINE_TILES) {
        int kn[COMBINE_TILES], first[COMBINE_TILES];
        uint32_t info[COMBINE_TILES];
        unsigned trun[COMBINE_TILES];
        // round 1: info words and the running child counts of the tiles
#pragma unroll
        for (int u = 0; u < COMBINE_TILES; ++u) {
            const int t = t0 + u;
            info[u] = I_NO_HIT;
            kn[u] = 0; first[u] = 0; trun[u] = 0;
            if (t < t_end) {
                while (T.node[kk].tile_end <= t) ++kk;
                kk = __builtin_amdgcn_readfirstlane(kk);
                kn[u] = kk;
                const NodeRec r = T.node[kk];
                first[u] = (t - __builtin_amdgcn_readfirstlane(r.tile0)) * TILE;
                const int cnt = min(TILE, __builtin_amdgcn_readfirstlane(r.n) - first[u]);
                if (lane < cnt) info[u] = D.info[(uint32_t)(tbase + t) * TILE + lane];
                if (has_children && lane < 3) trun[u] = D.tile_run[(size_t)(tbase + t) * NCNT + lane];
            }
        }
        // round 2: what the info words select
        double A[COMBINE_TILES], through[COMBINE_TILES];
        V3 surf[COMBINE_TILES], refr[COMBINE_TILES], refl[COMBINE_TILES];
#pragma unroll
        for (int u = 0; u < COMBINE_TILES; ++u) {
            const uint32_t idx = (uint32_t)(tbase + t0 + u) * TILE + lane;
            const uint32_t in = info[u];
            A[u] = 0; through[u] = 0;
            surf[u] = {0, 0, 0}; refr[u] = {0, 0, 0}; refl[u] = {0, 0, 0};
            const bool want_ti = (in & I_LIVE_TI) && (in & (I_CHILD_T | I_CHILD_I));
            const bool want_r = (in & I_LIVE_R) && (in & I_CHILD_R);
            uint32_t slot[3] = {0, 0, 0};   // arena index of the ray's child of each kind
            if (has_children && __ballot(want_ti || want_r) != 0) {
                const int base = (int)(wave_base + trun[u] - T.pre[kn[u] * 16 + (lane & 15)]);
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    slot[c] = T.child[kn[u] * 4 + c] + LANE(base, c) + lanes_below(__ballot(((in >> (18 + c)) & 1u) != 0));
            }
            if ((in & I_LIVE) && is_hit(in)) {
                A[u] = D.A[idx];
                if (in & I_SHADE) surf[u] = {D.cr[idx], D.cg[idx], D.cb[idx]};
                if (want_ti) {
                    const uint32_t sl = (in & I_CHILD_T) ? slot[C_T] : slot[C_I];
                    refr[u] = {D.cr[sl], D.cg[sl], D.cb[sl]};
                    if ((in & I_CHILD_T) && !(in & I_INSIDE)) through[u] = D.dist[idx];
                }
                if (want_r) {
                    const uint32_t sl = slot[C_R];
                    refl[u] = {D.cr[sl], D.cg[sl], D.cb[sl]};
                }
            }
        }
        // arithmetic and stores
#pragma unroll
        for (int u = 0; u < COMBINE_TILES; ++u) {
            const uint32_t idx = (uint32_t)(tbase + t0 + u) * TILE + lane;
            const uint32_t in = info[u];
            if (in & I_LIVE) {   // nobody reads the colour of a ray whose weight in the pixel is zero
                V3 col = bg;
                if (is_hit(in)) {
                    const int sid = in & I_SID_MASK;
                    const CombineSphere cs = SMALL ? s_cs[sid] : sphere_consts(sid);
                    V3 rf = refr[u];
                    if ((in & I_LIVE_TI) && (in & I_CHILD_T) && !(in & I_INSIDE)) {
                        // absorption inside the entered sphere (reference raytrace.py:119-120)
                        rf.x *= exp(cs.kr * through[u]);
                        rf.y *= exp(cs.kg * through[u]);
                        rf.z *= exp(cs.kb * through[u]);
                    }
                    col = clip01(((1 - cs.tr) * (1 - A[u])) * surf[u] + (cs.tr * (1 - A[u])) * rf + A[u] * refl[u]);
                }
                D.cr[idx] = col.x; D.cg[idx] = col.y; D.cb[idx] = col.z;
                if (level == 0) {
                    double* out = D.image + (size_t)(first[u] + lane) * 3;
                    out[0] = col.x; out[1] = col.y; out[2] = col.z;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 8-bit image as the reference's output step stores it (reference main.py:32, plt.imsave
// of a float RGB array): channel = (uint8)(v * 255) for v already clipped to [0, 1].
__global__ __launch_bounds__(BLOCK) void k_quantise(const double* __restrict__ rgb, uint8_t* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK)
        out[i] = (uint8_t)(int)(fmin(fmax(rgb[i], 0.0), 1.0) * 255.0);
}

}  // namespace rt
