// Device helpers shared by all kernels: float64 vectors, correctly rounded sqrt / division without
// range scaling, packed float32 pairs, the float32 screening of ray-sphere pairs, the reference-order
// ray-sphere quadratic, info-word predicates.
#pragma once
#include "rt_internal.h"

namespace rt {

// Store of a value that nobody reads soon: goes to memory without displacing cached data that is still in use.
template <typename T>
__device__ __forceinline__ void stream_store(T v, T* p) { __builtin_nontemporal_store(v, p); }



struct V3 {
    double x, y, z;
};

// Colours, reflected fractions and in-sphere path lengths are smooth quantities: no hit / miss
// decision, count or random-stream address depends on them.  In mixed precision their arrays hold
// float32 values (half the bytes through the colour pass), in float64 mode doubles; the arithmetic on
// them is float64 either way.  `f32` is uniform over the launch.
template <bool F32> struct SmoothType { typedef double type; };
template <> struct SmoothType<true> { typedef float type; };
__device__ __forceinline__ double load_smooth(const double* base, uint32_t i, bool f32) {
    return f32 ? (double)reinterpret_cast<const float*>(base)[i] : base[i];
}
__device__ __forceinline__ void store_smooth(double* base, uint32_t i, double v, bool f32) {
    if (f32) reinterpret_cast<float*>(base)[i] = (float)v;
    else base[i] = v;
}
__device__ __forceinline__ void stream_store_smooth(double* base, uint32_t i, double v, bool f32) {
    if (f32) __builtin_nontemporal_store((float)v, reinterpret_cast<float*>(base) + i);
    else __builtin_nontemporal_store(v, base + i);
}
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
    // fast path: the norm and all three components in the middle of the exponent range (a
    // component that is exactly zero takes the general path too, which is merely slower)
    const uint32_t ex = (uint32_t)__double2hiint(v.x) & 0x7fffffffu, ey = (uint32_t)__double2hiint(v.y) & 0x7fffffffu,
                   ez = (uint32_t)__double2hiint(v.z) & 0x7fffffffu;
    const bool fast = mid_range(len) && min(min(ex, ey), ez) >= (623u << 20);   // components <= norm < 2^400
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

// Tiles per wave of a level with nt tiles: every wave of the fixed grid owns a contiguous run of this
// many tiles (the same number in every kernel that walks the level, so that a tile's running counters
// are found again).  A run is never shorter than RT_MIN_TILES_PER_WAVE: on a small level fewer waves
// work, each with a filled software pipeline and one staging of the tables for more tiles.
#ifndef RT_MIN_TILES_PER_WAVE
#define RT_MIN_TILES_PER_WAVE 8
#endif
// (the level kernels of a frame run with D.nb workgroups: fewer for small frames, see rt_begin_frame)
__device__ __forceinline__ int tiles_per_wave(const Dev& D, int nt) {
    const int waves = D.nb * WPB;
    return max((nt + waves - 1) / waves, RT_MIN_TILES_PER_WAVE);
}

constexpr int CAND_MAX = 16;   // survivors a lane collects before the wave tests them
// Number of set bits of a wave ballot below the calling lane.
__device__ __forceinline__ int lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

// Level-wide running value of counter c at the start of tile t (relative tile index).
__device__ __forceinline__ unsigned gscan(const Dev& D, int level, int tbase, int q, int t, int c) {
    const int w = t / q;   // the wave that produced the tile
    return D.blk_tot[((size_t)level * NCNT + c) * NB + w / WPB] + D.wave_off[((size_t)level * NCNT + c) * NW + w] +
           D.tile_run[(size_t)(tbase + t) * NCNT + c];
}

#define LANE(v, i) __builtin_amdgcn_readlane((v), (i))

__device__ __forceinline__ bool is_hit(uint32_t info) { return (info & I_SID_MASK) != I_NO_HIT; }

// Whether a ray with this info word is counted by counter c.
__device__ __forceinline__ bool counted(uint32_t info, int c) {
    if (c == C_HIT) return is_hit(info);
    if (c == C_REC) return is_hit(info) && (info & (I_CHILD_T | I_CHILD_I | I_CHILD_R | I_SHADE)) != 0;
    if (c == C_LIVE) return is_hit(info) && (info & I_LIVE) != 0;
    if (c == C_ANYLIVE) return (info & I_LIVE) != 0;
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

}  // namespace rt
