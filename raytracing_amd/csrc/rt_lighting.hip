// Lighting: shadow transmittance, point / disk / spot lights in float64 and the certified float32
// sample loop, k_surface over the dense shading list, k_surface_redo for uncertified hits.
#pragma once
#include "rt_geometry.hip"

namespace rt {

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
struct ShadowLists {   // per-lane lists in LDS for the screening of many spheres (rt_hier.h)
    LaneList clusters, spheres;
};
template <bool LDS>
__device__ __forceinline__ V3 shadow_multiplier(const Dev& D, const ShadowSphere* s_sh, const ShadowLists& lists, V3 o, V3 d) {
    V3 mult = {1, 1, 1};
    const int S = D.S;
    const double d_o = dot(d, o);
    if (!LDS) {
        // many spheres: float32 screening first (see cull_test), the float64 test for the
        // survivors of this lane, in sphere order (the order of the multiplications)
        const V3 ref = {D.cull_ref[0], D.cull_ref[1], D.cull_ref[2]};
        const V3 on = o + dot(ref - o, d) * d;
        const float ox = (float)on.x, oy = (float)on.y, oz = (float)on.z;
        const float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
        const float e_ray = cull_ray_margin(on);
        bool opaque = false;
        auto occluder = [&](int i) {
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
        };
        bool every_sphere = true;   // this lane screens the whole table
        if (D.hier_top_groups) {
            // the ray starts at -dot(ref - o, d) on the line, measured from `on`; rounded down
            const double s_o = -dot(ref - o, d);
            const float s_from = (float)(s_o - 1e-6 * fabs(s_o)) - e_ray;
            const HierWindow whole = {s_from, __builtin_nanf(""), __builtin_inff(), __builtin_inff()};   // every occluder counts
            bool more;
            const int n = hier_candidates(D, lists.clusters, lists.spheres, true, ox, oy, oz, dx, dy, dz, e_ray, whole, more);
            for (int k = 0; __ballot(k < n) != 0; ++k)
                if (k < n && !opaque) occluder(lists.spheres[k]);
            every_sphere = n < 0;   // a list overflowed
        }
        if (every_sphere) {
            typedef const __attribute__((address_space(4))) CullPair* ConstCull;   // uniform reads through the scalar unit
            ConstCull table = (ConstCull)(uintptr_t)D.cull;
            const LaneList cand = lists.spheres;
            int n_mine = 0;
            auto flush = [&]() {
                for (int k = 0; __ballot(k < n_mine) != 0; ++k)   // the longest list of the active lanes
                    if (k < n_mine && !opaque) occluder(cand[k]);
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
        }
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
__device__ SampleTerms disk_sample_f64(const Dev& D, const ShadowSphere* s_sh, const ShadowLists& cand, const RtLight& Lt, double u_theta,
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

// SIMPLE: every diffuse / specular coefficient of the scene lies in [0, 1] and every disk light
// has equal diffuse and specular colours (the host checks).  Then clip(nl * k) = k * clip(nl) and
// clip(p * k) = k * min(p, 1), so one sum of clip(nl) and one of min(p, 1) replace six channel
// sums, and the specular intensity sum equals the diffuse one.
template <bool LDS, bool SIMPLE>
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
            if constexpr (SIMPLE) {
                v2f cN = vclip01(nl), cP = {fminf(p.x, 1.0f), fminf(p.y, 1.0f)};
                if (!both) {
                    const v2f first = {1.0f, 0.0f};
                    cId.x *= first; cId.y *= first; cId.z *= first; cN *= first; cP *= first;
                }
                aId.x += cId.x; aId.y += cId.y; aId.z += cId.z;
                aKd.x += cN;    // sum of clip(n.l); the coefficients are applied after the loop
                aKs.x += cP;    // sum of min(|r.d|^shininess, 1)
            } else {
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
            }
            __builtin_amdgcn_sched_barrier(0);   // one pair at a time: interleaving pairs costs registers
        }
    }
    auto total = [](F3x2 v) -> V3 { return {(double)(v.x.x + v.x.y), (double)(v.y.x + v.y.y), (double)(v.z.x + v.z.y)}; };
    sumId = total(aId);
    if constexpr (SIMPLE) {
        const float sN = aKd.x.x + aKd.x.y, sP = aKs.x.x + aKs.x.y;
        sumIs = sumId;
        sumKd = {(double)(sN * kdf.x), (double)(sN * kdf.y), (double)(sN * kdf.z)};
        sumKs = {(double)(sP * ksf.x), (double)(sP * ksf.y), (double)(sP * ksf.z)};
    } else {
        sumIs = total(aIs); sumKd = total(aKd); sumKs = total(aKs);
    }
}

// Surface colour of one hit under all lights (reference raytrace.py:136-152,
// :215-234 and objects.py:94-235).  entry: index of the hit in the shading list,
// which is also the index of its disk-light samples in the packed sample buffer.
template <bool LDS, bool MIXED, bool SIMPLE = false>
__device__ V3 surface_color(const Dev& D, const SceneConst& sc, const ShadowSphere* s_sh, const ShadowLists& cand, int64_t entry, uint32_t info,
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
            disk_light_mixed<LDS, SIMPLE>(D, s_sh, Lt, D.packed32 + 2 * (D.pack_base[l] + entry), stride, Pb, d, n, kdiff, kspec, shin, sumId, sumIs, sumKd, sumKs,
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
// Lighting for the dense list of hits whose surface colour matters (all levels).
template <bool LDS, bool MIXED, bool SIMPLE = false>
__global__ __launch_bounds__(BLOCK, MIXED ? RT_MIXED_WAVES : (LDS ? 2 : RT_SURFACE_WAVES_MANY)) void k_surface(Dev D) {
    constexpr int TABLE = MIXED ? SREG : (LDS ? SHADE_LDS_SPHERES : 1);   // the float32 pass is for S <= SREG
    __shared__ ShadowSphere s_sh[TABLE];
    __shared__ SurfaceMaterial s_mat[TABLE];
    __shared__ double s_centre[TABLE][3];
    __shared__ SceneConst s_sc;
    __shared__ unsigned s_work[3];
    __shared__ uint16_t s_lists[LDS ? 1 : 2 * HIER_CAP][LDS ? 1 : BLOCK];   // per lane: spheres the shadow screening kept
    const int tid = threadIdx.x;
    const ShadowLists lists = {{&s_lists[LDS ? 0 : HIER_CAP][LDS ? 0 : tid]}, {&s_lists[0][LDS ? 0 : tid]}};
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
        V3 surf = surface_color<LDS, MIXED, SIMPLE>(D, sc, s_sh, lists, e, info, sf.Pb, d, sf.n, mat, my_samples, unsure);
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
    __shared__ uint16_t s_lists[LDS ? 1 : 2 * HIER_CAP][LDS ? 1 : BLOCK];
    const int tid = threadIdx.x;
    const ShadowLists lists = {{&s_lists[LDS ? 0 : HIER_CAP][LDS ? 0 : tid]}, {&s_lists[0][LDS ? 0 : tid]}};
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
        V3 surf = surface_color<LDS, false>(D, sc, s_sh, lists, e, info, sf.Pb, d, sf.n,
                                            surface_material(D.mat + (size_t)sid * RT_MAT_WIDTH), a, b);
        D.cr[ray] = surf.x; D.cg[ray] = surf.y; D.cb[ray] = surf.z;
    }
}

}  // namespace rt
