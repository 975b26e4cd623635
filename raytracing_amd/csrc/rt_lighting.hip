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
// `light`: index of the point light the ray goes to, or -1 (a sample of a disk light).
// GRID (many spheres): the scene has a uniform grid, the other searches are left out.
template <bool LDS, bool GRID = false>
__device__ __forceinline__ V3 shadow_multiplier(const Dev& D, const ShadowSphere* s_sh, const ShadowLists& lists, V3 o, V3 d, int light) {
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
        if (GRID || D.grid_cells) {
            // uniform grid: the spheres outside it and the survivors of the cells along the whole ray,
            // then ascending sphere order (the order of the multiplications), every sphere once
            const LaneList cand = lists.spheres;
            typedef const __attribute__((address_space(4))) uint16_t* ConstIds;
            ConstIds loose = (ConstIds)(uintptr_t)D.grid_loose;
            int n = 0;
            bool overflow = false;
            auto push = [&](uint32_t id) {
                // a sphere sits in every cell it reaches: the walk meets it in consecutive cells
                if ((n > 0 && cand[n - 1] == id) || (n > 1 && cand[n - 2] == id)) return;
                if (n < GRID_LIST) cand[n++] = (uint16_t)id; else overflow = true;
            };
            bool walk = true;
            if (light >= 0 && ((D.lmap_mask >> light) & 1u)) {
                // A point light with a map of its occluders by direction (lightmap_build.h): the ray lies on
                // a line through the light, so what it can touch is listed in the bin of the direction
                // from the light to the ray's origin and in the bin of the opposite direction, plus the
                // light's "always" spheres.  The listed spheres take the float32 screening of the grid.
                ConstIds always = (ConstIds)(uintptr_t)(D.lmap_always + D.lmap_always_off[light]);
                for (int k = 0; k < D.lmap_n_always[light]; ++k) push(always[k]);
                const double* lp = D.sc->lights[light].position;
                const float qx = (float)(o.x - lp[0]), qy = (float)(o.y - lp[1]), qz = (float)(o.z - lp[2]);
                const float ax = fabsf(qx), ay = fabsf(qy), az = fabsf(qz);
                int face;
                float qa, qb, qc;
                if (ax >= ay && ax >= az) { face = 0; qa = qx; qb = qy; qc = qz; }
                else if (ay >= az) { face = 2; qa = qy; qb = qz; qc = qx; }
                else { face = 4; qa = qz; qb = qx; qc = qy; }
                if (qa < 0) face |= 1;
                if (!(fabsf(qa) > 1e-20f)) overflow = true;   // the ray starts at the light itself: no direction, walk the grid
                const float inv = 1.0f / fabsf(qa);
                const float u = qb * inv, w_ = qc * inv;
                // the ray in the grid's coordinates, as the screening entries are
                const float px = (float)(o.x - D.grid_origin[0]), py = (float)(o.y - D.grid_origin[1]), pz = (float)(o.z - D.grid_origin[2]);
                const float fx = (float)d.x, fy = (float)d.y, fz = (float)d.z;
                const float e_ray = 1.5e-6f * ((fabsf(px) + fabsf(py) + fabsf(pz)) * 1.000001f);
                const uint32_t* start = D.lmap_start + D.lmap_start_off[light];
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const float su = side ? -u : u, sw = side ? -w_ : w_;
                    const int iu = min(max((int)floorf((su * 0.5f + 0.5f) * LMAP_N), 0), LMAP_N - 1);
                    const int iw = min(max((int)floorf((sw * 0.5f + 0.5f) * LMAP_N), 0), LMAP_N - 1);
                    const uint32_t bin = ((uint32_t)(face ^ side) * LMAP_N + iw) * LMAP_N + iu;
                    const uint2 se = *reinterpret_cast<const uint2*>(start + 2 * (size_t)bin);
                    for (uint32_t k = se.x; __ballot(k < se.y) != 0; ++k) {
                        if (k < se.y) {
                            const uint32_t id = D.lmap_items[k];
                            const float4 s = D.grid_sph[id];
                            const float ocx = s.x - px, ocy = s.y - py, ocz = s.z - pz;
                            const float b = fmaf(ocz, fz, fmaf(ocy, fy, ocx * fx));
                            const float rx = fmaf(-b, fx, ocx), ry = fmaf(-b, fy, ocy), rz = fmaf(-b, fz, ocz);
                            const float p2 = fmaf(rz, rz, fmaf(ry, ry, rx * rx));
                            const float t = s.w + e_ray;
                            // not a certain miss of the line, and not wholly behind the ray's origin
                            if (!(fmaf(-t, t, p2) > 0) && !(-(b + t) > 0)) push(id);
                        }
                    }
                }
                walk = overflow;
                if (overflow) { n = 0; overflow = false; }
            }
            if (walk) {
                for (int k = 0; k < D.grid_n_loose; ++k) push(loose[k]);
                GridWalk w = grid_begin(D, true, o, d);
                while (__ballot(!w.done) != 0) grid_step(D, w, w.t_out, push);
            }
            for (int i = 1; i < n; ++i) {   // the list is short: insertion sort
                const uint16_t v = cand[i];
                int j = i;
                while (j > 0 && cand[j - 1] > v) { cand[j] = cand[j - 1]; --j; }
                cand[j] = v;
            }
            if (!overflow) {
                for (int k = 0; __ballot(k < n) != 0; ++k)
                    if (k < n && !opaque && (k == 0 || cand[k] != cand[k - 1])) occluder(cand[k]);
            }
            every_sphere = overflow;
            if (overflow) { mult = {1, 1, 1}; opaque = false; }
        } else if (!GRID && D.hier_top_groups) {
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
template <bool LDS, bool GRID = false>
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
    V3 mult = shadow_multiplier<LDS, GRID>(D, s_sh, cand, Pb, Lh, -1);
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

// ---------------------------------------------------------------------------------------------
// Surface colour of one hit under all lights in float64 (reference raytrace.py:136-152,
// :215-234 and objects.py:94-235).  entry: index of the hit in the shading list,
// which is also the index of its disk-light samples in the packed sample buffer.
// POINT_ONLY: the scene has no sampled (disk, spot) light; their code is left out, which keeps the
// many-sphere kernel's register demand down.
template <bool LDS, bool POINT_ONLY = false, bool GRID = false>
__device__ V3 surface_color(const Dev& D, const SceneConst& sc, const ShadowSphere* s_sh, const ShadowLists& cand, int64_t entry, uint32_t info,
                            V3 Pb, V3 d, V3 n, const SurfaceMaterial& mat, unsigned& n_samples) {
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
            V3 mult = shadow_multiplier<LDS, GRID>(D, s_sh, cand, Pb, Lh, l);
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
        if constexpr (POINT_ONLY) continue;
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
        const double* pk = D.packed + D.pack_base[l] + entry;   // + pack_E per sample
        const int64_t stride = D.pack_E;
        // (samples that were generated on the device and only packed as disk points: from the raw words)
        const uint32_t* raw_values = D.raw_values_at ? *D.raw_values_at : nullptr;
        const GeneratedStream raw{raw_values};
        SampleColumn col{0, 0, false};
        if (raw_values) col = sample_column<true>(D, entry, D.list_node[entry], info, l, Lt);
        V3 sumId = {0, 0, 0}, sumIs = {0, 0, 0}, sumKd = {0, 0, 0}, sumKs = {0, 0, 0};
        for (int s = 0; s < N; ++s) {
            const double u_angle = raw_values ? raw(col.first + s * col.h) : pk[s * stride];
            const double u_radius = raw_values ? raw(col.first + ((int64_t)N + s) * col.h) : pk[D.pack_plane + s * stride];
            const SampleTerms o = disk_sample_f64<LDS, GRID>(D, s_sh, cand, Lt, u_angle, u_radius,
                                                       Pb, d, n, kdiff, kspec, shin);
            sumId = sumId + o.Id; sumIs = sumIs + o.Is; sumKd = sumKd + o.kd; sumKs = sumKs + o.ks;
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
// Lighting in float64 for the dense list of hits whose surface colour matters (all levels).
template <bool LDS, bool C32, bool POINT_ONLY = false, bool GRID = false>   // C32: surface colours stored as float32 (Dev::col32)
__global__ __launch_bounds__(BLOCK, LDS ? RT_SURFACE_WAVES : RT_SURFACE_WAVES_MANY) void k_surface(Dev D) {
    constexpr int TABLE = LDS ? SHADE_LDS_SPHERES : 1;
    __shared__ ShadowSphere s_sh[TABLE];
    __shared__ SurfaceMaterial s_mat[TABLE];
    __shared__ double s_centre[TABLE][3];
    __shared__ SceneConst s_sc;
    __shared__ unsigned s_work[2];
    __shared__ uint16_t s_lists[LDS ? 1 : 2 * HIER_CAP][LDS ? 1 : BLOCK];   // per lane: spheres the shadow screening kept
    const int tid = threadIdx.x;
    const ShadowLists lists = {{&s_lists[LDS ? 0 : HIER_CAP][LDS ? 0 : tid]}, {&s_lists[0][LDS ? 0 : tid]}};
    if (D.flags[0]) return;  // arena overflow: the frame is invalid, the host reports it
    if (tid < 2) s_work[tid] = 0;
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
    unsigned my_hits = 0, my_samples = 0;
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
        V3 surf = surface_color<LDS, POINT_ONLY, GRID>(D, sc, s_sh, lists, e, info, sf.Pb, d, sf.n, mat, my_samples);
        my_hits += 1;
        store_smooth(D.sr, e, surf.x, C32); store_smooth(D.sg, e, surf.y, C32); store_smooth(D.sb, e, surf.z, C32);
        e = e_next;
        ray = ray_next;
    }
    if (my_hits) {
        atomicAdd(&s_work[0], my_hits);
        atomicAdd(&s_work[1], my_samples);
    }
    __syncthreads();
    if (tid < 2 && s_work[tid]) atomicAdd(&D.work[tid], (unsigned long long)s_work[tid]);
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
        unsigned a = 0;
        V3 surf = surface_color<LDS>(D, sc, s_sh, lists, e, info, sf.Pb, d, sf.n,
                                     surface_material(D.mat + (size_t)sid * RT_MAT_WIDTH), a);
        store_smooth(D.sr, e, surf.x, true); store_smooth(D.sg, e, surf.y, true); store_smooth(D.sb, e, surf.z, true);   // follows the float32 sample pass: mixed precision
    }
}

}  // namespace rt

#include "rt_lighting_mixed.hip"
