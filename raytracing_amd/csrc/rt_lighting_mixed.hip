// Disk and spot lights with float32 samples and certified shadow decisions: k_surface_mixed walks
// the dense shading list like k_surface, one wave per tile of 64 hits; point lights of the same
// scene are evaluated in float64 as in k_surface.
#pragma once
#include "rt_lighting.hip"

namespace rt {

struct F3 {
    float x, y, z;
};
struct F3x2 {   // three components for two samples
    v2f x, y, z;
};
__device__ __forceinline__ F3 f3(V3 v) { return {(float)v.x, (float)v.y, (float)v.z}; }
__device__ __forceinline__ float fdot(F3 a, F3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ float fclip01(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f); }
__device__ __forceinline__ v2f vclip01(v2f a) { return {fclip01(a.x), fclip01(a.y)}; }
__device__ __forceinline__ v2f vdot(F3 a, F3x2 b) { return vfma(b.z, a.z, vfma(b.y, a.y, b.x * a.x)); }

constexpr int MIXED_MAX_SPHERES = 64;   // scenes up to this many spheres take the float32 sample pass (one 64-bit
                                        // mask of candidate spheres per hit)
constexpr int MIXED_T = 4;              // candidate spheres of a hit that the sample loop keeps in registers
#ifndef RT_PAIR_BATCH
#define RT_PAIR_BATCH 3
#endif
constexpr int PAIR_BATCH = RT_PAIR_BATCH;   // sample pairs whose loads are requested together, one batch ahead

// What a shadow ray from one hit needs of one candidate sphere, relative to the hit: computed once
// per hit in float64, then rounded.  With |L| = 1 the ray hits iff b^2 - C >= 0 and one root is
// positive, b = oc.L, C = |oc|^2 - r^2.
struct HitSphere {
    F3 oc;       // Pb - centre
    float C;     // |oc|^2 - r^2, sign exact (origin inside the sphere iff C < 0); +inf: not a candidate of this hit
    float len2;  // 2 (|oc| + allowance for the float64 rounding of b far from the origin)
    float k0;    // error bound of C (float rounding + float64 cancellation noise); +inf: the sign of C is in doubt
};
// What the sample loop needs of the hit itself.
struct DiskHit {
    F3 q;        // light centre - Pb
    F3 nf, df;   // outward normal, ray direction
    float nd2;   // -2 n.d:  r.d = L.d - 2 (n.L)(n.d)
    float sh;    // shininess
    float eLk;   // error of the computed light vector before the division by its length
    float ps;    // light intensity per sample (SIMPLE: times the light's grey level) times the transparency of
                 // every sphere the hit lies inside: such a sphere is met by every shadow ray of the hit with
                 // a zero-length path to its near side, so it multiplies by its transparency alone
                 // (reference objects.py:274-285 with min_t = max_t)
    F3 kdf, ksf; // diffuse / specular colour times coefficient (read by the general variant only)
};
// The light, the same for every lane (scalar registers).
struct DiskLightConst {
    F3 xh, yh;   // disk frame times the radius
    F3 dcol, scol;
};
// Sums over the samples of one hit and one light.
// SIMPLE (the host checks): every diffuse / specular coefficient of the scene lies in [0, 1] and every
// disk light has equal diffuse and specular colours with three equal, non-negative channels (the
// reference's disk lights are always white, objects.py:140-144).  Then clip(nl * k) = k * clip(nl),
// clip(p * k) = k * min(p, 1), the specular intensity equals the diffuse one, and an unoccluded
// sample has the same intensity in all channels: one sum of clip(n.l), one of min(p, 1), one of the
// unoccluded intensities and three channel sums for the occluded samples replace twelve channel sums.
template <bool SIMPLE>
struct SampleSums;
template <>
struct SampleSums<true> {
    v2f open = {0, 0}, nl = {0, 0}, pw = {0, 0};
    F3x2 shaded = {{0, 0}, {0, 0}, {0, 0}};
};
template <>
struct SampleSums<false> {
    F3x2 Id = {{0, 0}, {0, 0}, {0, 0}}, Is = {{0, 0}, {0, 0}, {0, 0}}, Kd = {{0, 0}, {0, 0}, {0, 0}}, Ks = {{0, 0}, {0, 0}, {0, 0}};
};

// Test mode: what the float64 re-evaluation of a sample's shadow decisions needs.
struct CertifyCheck {
    const Dev* D;
    const ShadowSphere* s_sh;
    uint64_t within;       // spheres the hit lies inside: met by every sample
    const double* pk64;    // the hit's first angle uniform in the float64 sample buffer
    int64_t stride64;      // doubles from one sample to the next
    int64_t plane64;       // doubles from an angle uniform to its radius uniform
    V3 Pb, lpos, xh, yh;
    double radius;
};
// Number of spheres whose certified float32 decision (hit: bit in `hit`, miss: no bit in `hit`;
// undecided spheres have their bit in `unsure` and are skipped) differs from the float64 decision of
// the reference's arithmetic for sample s.
__device__ __noinline__ unsigned certify_mismatches(const CertifyCheck& ck, int s, uint64_t hit, uint64_t unsure) {
    const double u_theta = ck.pk64[s * ck.stride64], u_rad = ck.pk64[ck.plane64 + s * ck.stride64];
    const double theta = u_theta * 6.283185307179586;
    const double rad = sqrt(u_rad) * ck.radius;
    double sn, cs;
    sincos(theta, &sn, &cs);
    const V3 lp = ck.lpos + rad * (cs * ck.xh + sn * ck.yh);
    const V3 Lv = lp - ck.Pb;
    const double dist = sqrt(dot(Lv, Lv));
    const V3 Lh = {Lv.x / dist, Lv.y / dist, Lv.z / dist};
    const double d_o = dot(Lh, ck.Pb);
    unsigned bad = 0;
    for (int i = 0; i < ck.D->S; ++i) {
        const ShadowSphere sp = ck.s_sh[i];
        double lo, hi;
        const bool hit64 = sphere_roots(ck.Pb, Lh, d_o, sp.cx, sp.cy, sp.cz, sp.r2, lo, hi) && hi > 0;
        if ((unsure >> i) & 1u) continue;
        if (hit64 != (bool)((hit >> i) & 1u)) bad += 1;
    }
    return bad;
}

// Two samples of one disk light for one hit in float32.  u = (x0, x1, y0, y1): the points of the
// unit disk the two samples select (k_disk_points).  w: weights 1 or 0 of the two samples (0: the
// slot lies beyond the light's sample count), on0 / on1 the same as flags; all three are the same
// in every lane.  The straight-line arithmetic runs on packed instructions (v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32); only rsq, log, exp, the clamps and the decisions are per sample.
//
// Every shadow decision is certified: the discriminant of each (sample, candidate sphere) test
// carries a rigorous error bound m.  With exact unit light vector L*, the computed one is within
//   eL = eLk / dist + 8e-7        (eLk: rounding of q, of the disk point and of the three fma chains
//                                   that form the light vector, 2.5e-7 (|q| + disk extent); the
//                                   constant: rsq, the products with it, the rounding of oc and of
//                                   the dot product b)
// of it, so b = oc.L is off by at most db = |oc| eL (len2 = 2 |oc|, plus what the float64 path's own
// rounding of b can be far from the origin), b^2 by 2 |b| db + db^2, and C by k0: |disc - disc*| <=
// m = |b| (2 db) + (2 db)^2 + k0.  A sample with |disc| <= m is unsure; otherwise the sign of disc is
// that of the exact discriminant, and for disc > m and C > 0 also |b| > db, so the sign of b (sphere
// ahead or behind) is exact as well.  The caller queues a hit with an unsure sample for re-evaluation
// in float64 (k_surface_redo), so the set of occluders that reaches the image is the float64 path's.
template <int T, bool SIMPLE, bool CHECK>
__device__ __forceinline__ void eval_pair(const float4 u, const v2f w, const bool on0, const bool on1, const DiskHit& H,
                                          const HitSphere (&hs)[MIXED_T], const int (&idx)[MIXED_T], const DiskLightConst& L,
                                          const ShadowSphere* s_sh, SampleSums<SIMPLE>& A, bool& unsure, const CertifyCheck& ck,
                                          int s_first, unsigned& mismatches) {
    const v2f px = {u.x, u.y}, py = {u.z, u.w};
    const F3x2 Lv = {vfma(px, L.xh.x, vfma(py, L.yh.x, (v2f)(H.q.x))), vfma(px, L.xh.y, vfma(py, L.yh.y, (v2f)(H.q.y))),
                     vfma(px, L.xh.z, vfma(py, L.yh.z, (v2f)(H.q.z)))};
    const v2f d2 = vfma(Lv.z, Lv.z, vfma(Lv.y, Lv.y, Lv.x * Lv.x));
    const v2f inv = {__builtin_amdgcn_rsqf(d2.x), __builtin_amdgcn_rsqf(d2.y)};
    const F3x2 Lh = {Lv.x * inv, Lv.y * inv, Lv.z * inv};
    bool un0 = false, un1 = false, hit0 = false, hit1 = false;
    bool h0[MIXED_T] = {false, false, false, false}, h1[MIXED_T] = {false, false, false, false};
    v2f disc[MIXED_T];
    uint64_t ck_hit0 = 0, ck_hit1 = 0, ck_un0 = 0, ck_un1 = 0;
    if constexpr (T > 0) {
        const v2f eL = vfma(inv, H.eLk, 8e-7f);
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const v2f b = vdot(hs[j].oc, Lh);
            disc[j] = vfma(b, b, (v2f)(-hs[j].C));
            const v2f db2 = eL * hs[j].len2;
            const v2f t = vfma(db2, db2, (v2f)(hs[j].k0));
            const float m0 = fmaf(fabsf(b.x), db2.x, t.x), m1 = fmaf(fabsf(b.y), db2.y, t.y);
            const bool inside = hs[j].C < 0;
            const bool u0 = fabsf(disc[j].x) <= m0, u1 = fabsf(disc[j].y) <= m1;
            h0[j] = disc[j].x > m0 && (b.x < 0 || inside);
            h1[j] = disc[j].y > m1 && (b.y < 0 || inside);
            un0 |= u0; un1 |= u1;
            hit0 |= h0[j]; hit1 |= h1[j];
            if constexpr (CHECK) {
                if (h0[j]) ck_hit0 |= 1ull << idx[j];
                if (h1[j]) ck_hit1 |= 1ull << idx[j];
                if (u0) ck_un0 |= 1ull << idx[j];
                if (u1) ck_un1 |= 1ull << idx[j];
            }
        }
        unsure |= (un0 && on0) || (un1 && on1);
    }
    if constexpr (CHECK) {
        if (on0) mismatches += certify_mismatches(ck, s_first, ck_hit0 | ck.within, ck_un0);
        if (on1) mismatches += certify_mismatches(ck, s_first + 1, ck_hit1 | ck.within, ck_un1);
    }
    // transmittance of the occluders (reference objects.py:274-285) as one exponential per channel:
    // the product over spheres of tr * exp(-k * chord) is exp2(sum of log2(tr) - k * log2(e) * chord),
    // chord = 2 sqrt(disc) for an origin outside the sphere and 0 inside.  Rare: most samples see the light.
    F3 mult0 = {1, 1, 1}, mult1 = {1, 1, 1};
    bool shaded = false;   // the same in every lane
    if constexpr (T > 0) {
        if (__ballot(hit0 || hit1) != 0) {
            shaded = true;
            F3 e0 = {0, 0, 0}, e1 = {0, 0, 0};
#pragma unroll
            for (int j = 0; j < T; ++j) {
                if (h0[j] || h1[j]) {
                    const ShadowSphere sp = s_sh[idx[j]];
                    const float lt = __builtin_amdgcn_logf((float)sp.tr);   // -inf for an opaque sphere
                    const float k2r = -2.0f * 1.4426950408889634f * (float)sp.kr, k2g = -2.0f * 1.4426950408889634f * (float)sp.kg,
                                k2b = -2.0f * 1.4426950408889634f * (float)sp.kb;
                    const bool outside = hs[j].C > 0;
                    if (h0[j]) {
                        const float sq = outside ? __builtin_amdgcn_sqrtf(disc[j].x) : 0.0f;
                        e0 = {fmaf(k2r, sq, e0.x + lt), fmaf(k2g, sq, e0.y + lt), fmaf(k2b, sq, e0.z + lt)};
                    }
                    if (h1[j]) {
                        const float sq = outside ? __builtin_amdgcn_sqrtf(disc[j].y) : 0.0f;
                        e1 = {fmaf(k2r, sq, e1.x + lt), fmaf(k2g, sq, e1.y + lt), fmaf(k2b, sq, e1.z + lt)};
                    }
                }
            }
            if (hit0) mult0 = {__builtin_amdgcn_exp2f(e0.x), __builtin_amdgcn_exp2f(e0.y), __builtin_amdgcn_exp2f(e0.z)};
            if (hit1) mult1 = {__builtin_amdgcn_exp2f(e1.x), __builtin_amdgcn_exp2f(e1.y), __builtin_amdgcn_exp2f(e1.z)};
        }
    }
    const v2f fall = inv * inv * H.ps;
    const v2f nl = vdot(H.nf, Lh);
    const v2f rd = vfma(nl, H.nd2, vdot(H.df, Lh));
    // |r.d|^shininess; |r.d| = 0 with shininess 0 gives NaN here, which fminf turns into 1 (0^0 = 1)
    const v2f p = {fminf(__builtin_amdgcn_exp2f(H.sh * __builtin_amdgcn_logf(fabsf(rd.x))), 1.0f),
                   fminf(__builtin_amdgcn_exp2f(H.sh * __builtin_amdgcn_logf(fabsf(rd.y))), 1.0f)};
    if constexpr (SIMPLE) {
        const v2f cf = vclip01(fall);
        A.nl = vfma(vclip01(nl), w, A.nl);
        A.pw = vfma(p, w, A.pw);
        if (shaded) {
            const v2f w_open = {hit0 ? 0.0f : w.x, hit1 ? 0.0f : w.y}, w_shaded = w - w_open;
            A.open = vfma(cf, w_open, A.open);
            A.shaded.x = vfma(vclip01(fall * (v2f){mult0.x, mult1.x}), w_shaded, A.shaded.x);
            A.shaded.y = vfma(vclip01(fall * (v2f){mult0.y, mult1.y}), w_shaded, A.shaded.y);
            A.shaded.z = vfma(vclip01(fall * (v2f){mult0.z, mult1.z}), w_shaded, A.shaded.z);
        } else {
            A.open = vfma(cf, w, A.open);
        }
    } else {
        const F3x2 mult = {{mult0.x, mult1.x}, {mult0.y, mult1.y}, {mult0.z, mult1.z}};
        A.Id.x = vfma(vclip01(fall * L.dcol.x * mult.x), w, A.Id.x);
        A.Id.y = vfma(vclip01(fall * L.dcol.y * mult.y), w, A.Id.y);
        A.Id.z = vfma(vclip01(fall * L.dcol.z * mult.z), w, A.Id.z);
        A.Is.x = vfma(vclip01(fall * L.scol.x * mult.x), w, A.Is.x);
        A.Is.y = vfma(vclip01(fall * L.scol.y * mult.y), w, A.Is.y);
        A.Is.z = vfma(vclip01(fall * L.scol.z * mult.z), w, A.Is.z);
        A.Kd.x = vfma(vclip01(nl * H.kdf.x), w, A.Kd.x);
        A.Kd.y = vfma(vclip01(nl * H.kdf.y), w, A.Kd.y);
        A.Kd.z = vfma(vclip01(nl * H.kdf.z), w, A.Kd.z);
        A.Ks.x = vfma(vclip01(p * H.ksf.x), w, A.Ks.x);
        A.Ks.y = vfma(vclip01(p * H.ksf.y), w, A.Ks.y);
        A.Ks.z = vfma(vclip01(p * H.ksf.z), w, A.Ks.z);
    }
}

// All samples of one disk light for one hit.  pk: this lane's slot of the tile's first pair in the
// float32 sample buffer; the tile's pairs follow each other TILE slots apart (k_disk_points), so a
// wave reads 1 KiB per pair, 16 bytes per lane.  The loads of the next batch of pairs are requested
// before the current batch is evaluated.  Slots beyond the sample count (an odd count, a batch that
// is not full) are evaluated with weight 0 on the data of the last pair.
// A pair of disk points: read exactly once per frame, so as a streaming load (measured -3 % on the kernel).
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 stream_load(const float4* p) {
    const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
template <int T, bool SIMPLE, bool CHECK>
__device__ __forceinline__ void disk_samples(const float4* __restrict__ pk, const int N, const DiskHit& H, const HitSphere (&hs)[MIXED_T],
                                             const int (&idx)[MIXED_T], const DiskLightConst& L, const ShadowSphere* s_sh,
                                             SampleSums<SIMPLE>& A, bool& unsure, const CertifyCheck& ck, unsigned& mismatches) {
    const int NP = (N + 1) >> 1;
    const int n_batches = (NP + PAIR_BATCH - 1) / PAIR_BATCH;
    float4 cur[PAIR_BATCH], nxt[PAIR_BATCH];
#pragma unroll
    for (int k = 0; k < PAIR_BATCH; ++k) cur[k] = stream_load(&pk[(int64_t)min(k, NP - 1) * TILE]);
    for (int bt = 0; bt < n_batches; ++bt) {
        // unconditional (the last batch requests the last pair again): a branch around the loads would
        // make the compiler wait for them before the batch at hand is evaluated
#pragma unroll
        for (int k = 0; k < PAIR_BATCH; ++k) nxt[k] = stream_load(&pk[(int64_t)min((bt + 1) * PAIR_BATCH + k, NP - 1) * TILE]);
#pragma unroll
        for (int k = 0; k < PAIR_BATCH; ++k) {
            const int s = 2 * (bt * PAIR_BATCH + k);
            const bool on0 = s < N, on1 = s + 1 < N;
            const v2f w = {on0 ? 1.0f : 0.0f, on1 ? 1.0f : 0.0f};
            eval_pair<T, SIMPLE, CHECK>(cur[k], w, on0, on1, H, hs, idx, L, s_sh, A, unsure, ck, s, mismatches);
            __builtin_amdgcn_sched_barrier(0);   // one pair at a time: interleaving pairs costs registers
        }
#pragma unroll
        for (int k = 0; k < PAIR_BATCH; ++k) cur[k] = nxt[k];
    }
}

// Lighting for the dense list of hits whose surface colour matters (all levels) in scenes with disk
// or spot lights and at most MIXED_MAX_SPHERES spheres (reference raytrace.py:136-152, :215-234 and
// objects.py:94-235).  One wave per tile of 64 list entries; entry e's samples are slot e % 64 of
// tile e / 64 in the float32 sample buffer.
//
// Per hit and sampled light, spheres no sample can touch are ruled out first: every unit light vector
// makes an angle of at most asin(extent / dc) with the direction to the disk's centre (the light points
// lie within `extent` of the centre, which is dc away), which bounds b = oc.L from both sides;
// a sphere the origin is outside of is missed by every sample if b stays positive (it lies
// behind every shadow ray) or b^2 stays below C.  The spheres a hit still needs are its candidates: up
// to MIXED_T of them, each lane its own, are tested per sample (the sample loop has as many slots as
// the hit of the wave with the most candidates needs; a hit with fewer leaves slots empty: C = +inf);
// a hit with more goes to the float64 re-evaluation.
template <bool SIMPLE, bool CHECK>
__global__ __launch_bounds__(BLOCK, RT_MIXED_WAVES) void k_surface_mixed(Dev D) {
    __shared__ ShadowSphere s_sh[MIXED_MAX_SPHERES];
    __shared__ SurfaceMaterial s_mat[MIXED_MAX_SPHERES];
    __shared__ unsigned s_work[4];
    const int tid = threadIdx.x, lane = tid & (TILE - 1);
    if (D.flags[0]) return;  // arena overflow: the frame is invalid, the host reports it
    if (tid < 4) s_work[tid] = 0;
    for (int i = tid; i < D.S; i += BLOCK) {
        const double* g = D.geom + (size_t)i * 4;
        const double* m = D.mat + (size_t)i * RT_MAT_WIDTH;
        s_sh[i] = {g[0], g[1], g[2], g[3] * g[3], m[11], m[14] * m[15], m[14] * m[16], m[14] * m[17]};
        s_mat[i] = surface_material(m);
    }
    __syncthreads();
    typedef const __attribute__((address_space(4))) SceneConst* ConstScene;   // the same in every lane: scalar loads
    ConstScene sc = (ConstScene)(uintptr_t)D.sc;
    const double eps = sc->eps;
    const int S = D.S;
    const int total = D.lvl_shade_base[D.n_levels];
    const int n_tiles = (total + TILE - 1) / TILE;
    const int n_waves = gridDim.x * WPB;
    const ShadowLists no_lists = {{nullptr}, {nullptr}};
    unsigned my_hits = 0, my_samples = 0, my_fallback = 0, my_mismatches = 0;
    for (int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + (tid >> 6)); tile < n_tiles; tile += n_waves) {
        const int e = tile * TILE + lane;
        const bool active = e < total;   // the last tile may be partly empty; its idle lanes shade the last entry again
        const uint32_t ray = D.list_ray[active ? e : total - 1];
        const uint32_t info = D.info[ray];
        const V3 P = {D.px[ray], D.py[ray], D.pz[ray]};
        const V3 d = {D.dx[ray], D.dy[ray], D.dz[ray]};
        const int sid = info & I_SID_MASK;
        const Surface sf = surface_at(P, V3{s_sh[sid].cx, s_sh[sid].cy, s_sh[sid].cz}, info & I_INSIDE, eps);
        const V3 Pb = sf.Pb, n = sf.n;
        const double pb1 = fabs(Pb.x) + fabs(Pb.y) + fabs(Pb.z);
        V3 surf = {0, 0, 0};
        bool redo = false;
        unsigned hit_samples = 0;   // light samples of this hit
        // point lights first, in float64 (the float32 pass is held to a looser tolerance anyway, so the order
        // in which the lights' contributions are added need not be the reference's): after this loop the
        // float64 normal and direction are no longer needed
        for (int l = 0; l < D.n_lights; ++l) {
            if (sc->lights[l].kind != RT_LIGHT_POINT) continue;
            const V3 lpos = {sc->lights[l].position[0], sc->lights[l].position[1], sc->lights[l].position[2]};
            const V3 kdiff = ld3(s_mat[sid].kdiff), kspec = ld3(s_mat[sid].kspec);
            const double shin = s_mat[sid].shininess;
            const V3 dcol = {sc->lights[l].diffuse_color[0], sc->lights[l].diffuse_color[1], sc->lights[l].diffuse_color[2]};
            const V3 scol = {sc->lights[l].specular_color[0], sc->lights[l].specular_color[1], sc->lights[l].specular_color[2]};
            const V3 Lv = lpos - Pb;
            const double dist = sqrt(dot(Lv, Lv));
            const V3 Lh = {Lv.x / dist, Lv.y / dist, Lv.z / dist};
            const V3 mult = shadow_multiplier<true>(D, s_sh, no_lists, Pb, Lh, -1);
            const double fall = sc->lights[l].intensity / (dist * dist);
            const V3 Id = clip01((fall * dcol) * mult), Is = clip01((fall * scol) * mult);
            const double nl = dot(n, Lh);
            const V3 Rv = Lh - (2 * nl) * n;
            const double rd = fabs(dot(Rv, d));
            const V3 kd = clip01(nl * kdiff), ks = clip01(pow_shininess(rd, shin) * kspec);
            surf = surf + clip01(clip01(kd * Id) + clip01(ks * Is));
            if (active) hit_samples += 1;
        }
        // what the sample loop needs of the hit, whatever the light
        DiskHit H;
        H.nf = f3(n); H.df = f3(d);
        H.nd2 = -2.0f * fdot(H.nf, H.df);
        H.sh = (float)s_mat[sid].shininess;
        if constexpr (!SIMPLE) {
            H.kdf = f3(ld3(s_mat[sid].kdiff)); H.ksf = f3(ld3(s_mat[sid].kspec));
        }
        for (int l = 0; l < D.n_lights; ++l) {
            const int kind = sc->lights[l].kind;
            if (kind == RT_LIGHT_POINT) continue;
            const V3 lpos = {sc->lights[l].position[0], sc->lights[l].position[1], sc->lights[l].position[2]};
            // disk light, possibly restricted to a beam (then hits outside the beam get no light from it)
            bool lit = active;
            double factor = 1.0;
            if (kind == RT_LIGHT_SPOT) {
                lit = lit && ((info >> (I_BEAM_SHIFT + l)) & 1u);
                const V3 rel = Pb - lpos;
                const V3 lx = {sc->lights[l].x_hat[0], sc->lights[l].x_hat[1], sc->lights[l].x_hat[2]};
                const V3 ly = {sc->lights[l].y_hat[0], sc->lights[l].y_hat[1], sc->lights[l].y_hat[2]};
                const V3 ln = {sc->lights[l].normal[0], sc->lights[l].normal[1], sc->lights[l].normal[2]};
                const double x = dot(lx, rel), y = dot(ly, rel), z = dot(ln, rel);
                const double rho = sqrt(x * x + y * y);
                const double allowed = sc->lights[l].radius + sc->lights[l].tan_angle * fabs(z);
                const double edge = sc->lights[l].radius + sc->lights[l].tan_outer * fabs(z);
                factor = ease(sc->lights[l].easing, rho, allowed, edge - allowed);
            }
            if (__ballot(lit) == 0) continue;
            const int N = sc->lights[l].n_points;
            if (lit) hit_samples += N;
            DiskLightConst L;
            L.xh = {sc->lightf[l].xh[0], sc->lightf[l].xh[1], sc->lightf[l].xh[2]};
            L.yh = {sc->lightf[l].yh[0], sc->lightf[l].yh[1], sc->lightf[l].yh[2]};
            L.dcol = {sc->lightf[l].dcol[0], sc->lightf[l].dcol[1], sc->lightf[l].dcol[2]};
            L.scol = {sc->lightf[l].scol[0], sc->lightf[l].scol[1], sc->lightf[l].scol[2]};
            const float extent = sc->lightf[l].extent;
            // what the sample loop needs of the hit
            const V3 qd = lpos - Pb;
            H.q = f3(qd);
            const float q2 = fdot(H.q, H.q);
            const float inv_dc = __builtin_amdgcn_rsqf(q2);
            H.eLk = 2.5e-7f * (extent + q2 * inv_dc);
            const F3 chat = {H.q.x * inv_dc, H.q.y * inv_dc, H.q.z * inv_dc};
            // sine of the largest angle between a unit light vector and chat (the disk lies in a ball of
            // radius extent around its centre, which is 1 / inv_dc away), with room for float rounding
            const float spread = extent * inv_dc * 1.000002f + 1e-7f;
            const bool narrow = spread <= 0.9f;
            // candidates: the spheres some sample of this hit may touch
            uint64_t need = 0, within = 0;
            float tr_within = 1.0f;
            for (int i = 0; i < S; ++i) {
                const V3 oc = {Pb.x - s_sh[i].cx, Pb.y - s_sh[i].cy, Pb.z - s_sh[i].cz};
                const double l2 = dot(oc, oc), r2 = s_sh[i].r2, C = l2 - r2;
                const float len = __builtin_amdgcn_sqrtf((float)l2) * 1.000001f;
                const double c1 = fabs(s_sh[i].cx) + fabs(s_sh[i].cy) + fabs(s_sh[i].cz);
                const double noise = 2e-15 * (l2 + r2) + 4e-15 * (pb1 + c1) * (double)len;   // float64 rounding of C
                // b = oc.L for a unit vector L at most that angle from chat: with bc = oc.chat and perp the
                // part of oc across chat, |b - bc cos| <= perp sin; the sphere lies behind every shadow ray
                // if the smallest b is positive, and no ray reaches it if the largest b^2 stays below C
                const float bc = fdot(f3(oc), chat);
                const float l2f = len * len;
                const float perp = __builtin_amdgcn_sqrtf(fmaxf(l2f - bc * bc, 0.0f) + 4e-6f * l2f) * spread;
                const float slack = 4e-6f * len;   // float rounding of chat, oc and bc
                const float bm = fabsf(bc) + perp + slack;
                const bool ruled_out = narrow && C > noise && (bc * (1.0f - spread * spread) - perp > slack || bm * bm < (float)C * 0.999998f);
                const bool inside = C < -noise;   // every sample meets this sphere: see DiskHit::ps
                if (inside) tr_within *= (float)s_sh[i].tr;
                if (CHECK && inside) within |= 1ull << i;
                if (lit && !ruled_out && !inside) need |= 1ull << i;
            }
            H.ps = (SIMPLE ? sc->lightf[l].ps_grey : sc->lightf[l].ps) * tr_within;
            int mine = __popcll(need);
            const bool too_many = mine > MIXED_T;
            if (too_many) {   // more candidates than the sample loop holds: this hit goes to the float64 re-evaluation
                redo = true;
                need = 0;
                mine = 0;
            }
            // slots of the sample loop: the largest number of candidates of any hit of the wave
            int Tn = 0;
#pragma unroll
            for (int k = 1; k <= MIXED_T; ++k)
                if (__ballot(mine >= k) != 0) Tn = k;
            HitSphere hs[MIXED_T];
            int idx[MIXED_T];
            {
                uint64_t rest = need;
#pragma unroll
                for (int j = 0; j < MIXED_T; ++j) {
                    idx[j] = 0;
                    hs[j] = {{0, 0, 0}, __builtin_inff(), 0, 0};
                    if (j < Tn) {   // the same in every lane; a hit with fewer candidates leaves the slot empty (C = +inf)
                        const bool used = rest != 0;
                        const int i = used ? __builtin_ctzll(rest) : 0;
                        rest &= rest - 1;
                        idx[j] = i;
                        const V3 oc = {Pb.x - s_sh[i].cx, Pb.y - s_sh[i].cy, Pb.z - s_sh[i].cz};
                        const double l2 = dot(oc, oc), r2 = s_sh[i].r2, C = l2 - r2;
                        const float len = __builtin_amdgcn_sqrtf((float)l2) * 1.000001f;
                        const double c1 = fabs(s_sh[i].cx) + fabs(s_sh[i].cy) + fabs(s_sh[i].cz);
                        const double noise = 2e-15 * (l2 + r2) + 4e-15 * (pb1 + c1) * (double)len;
                        hs[j].oc = f3(oc);
                        hs[j].len2 = 2.0f * (len + 1e-9f * (float)(pb1 + c1));
                        hs[j].k0 = fabs(C) <= noise ? __builtin_inff() : (float)(1.5e-7 * fabs(C) + noise);
                        hs[j].C = used ? (float)C : __builtin_inff();
                    }
                }
            }
            const int NP = (N + 1) >> 1;
            const float4* pk = reinterpret_cast<const float4*>(D.packed32) + D.pack32_base[l] + ((int64_t)tile * NP * TILE + lane);
            CertifyCheck ck{};
            if constexpr (CHECK) {
                ck.D = &D; ck.s_sh = s_sh; ck.within = within;
                ck.pk64 = D.packed + D.pack_base[l] + (active ? e : total - 1);
                ck.stride64 = D.pack_E;
                ck.plane64 = D.pack_plane;
                ck.Pb = Pb; ck.lpos = lpos;
                ck.xh = {sc->lights[l].x_hat[0], sc->lights[l].x_hat[1], sc->lights[l].x_hat[2]};
                ck.yh = {sc->lights[l].y_hat[0], sc->lights[l].y_hat[1], sc->lights[l].y_hat[2]};
                ck.radius = sc->lights[l].radius;
            }
            SampleSums<SIMPLE> A;
            bool unsure = false;
            unsigned mismatches = 0;
            switch (Tn) {
                case 0: disk_samples<0, SIMPLE, CHECK>(pk, N, H, hs, idx, L, s_sh, A, unsure, ck, mismatches); break;
                case 1: disk_samples<1, SIMPLE, CHECK>(pk, N, H, hs, idx, L, s_sh, A, unsure, ck, mismatches); break;
                case 2: disk_samples<2, SIMPLE, CHECK>(pk, N, H, hs, idx, L, s_sh, A, unsure, ck, mismatches); break;
                case 3: disk_samples<3, SIMPLE, CHECK>(pk, N, H, hs, idx, L, s_sh, A, unsure, ck, mismatches); break;
                default: disk_samples<4, SIMPLE, CHECK>(pk, N, H, hs, idx, L, s_sh, A, unsure, ck, mismatches); break;
            }
            if (lit) {
                redo = redo || unsure;
                if (CHECK && !unsure && !too_many) my_mismatches += mismatches;
            }
            V3 sumId, sumIs, sumKd, sumKs;
            if constexpr (SIMPLE) {
                const float open = A.open.x + A.open.y;
                sumId = {(double)(open + (A.shaded.x.x + A.shaded.x.y)), (double)(open + (A.shaded.y.x + A.shaded.y.y)),
                         (double)(open + (A.shaded.z.x + A.shaded.z.y))};
                sumIs = sumId;
                const double sN = (double)(A.nl.x + A.nl.y), sP = (double)(A.pw.x + A.pw.y);
                sumKd = sN * ld3(s_mat[sid].kdiff);
                sumKs = sP * ld3(s_mat[sid].kspec);
            } else {
                auto both = [](const F3x2& v) -> V3 { return {(double)(v.x.x + v.x.y), (double)(v.y.x + v.y.y), (double)(v.z.x + v.z.y)}; };
                sumId = both(A.Id); sumIs = both(A.Is); sumKd = both(A.Kd); sumKs = both(A.Ks);
            }
            const double invN = 1.0 / N;
            V3 Id = clip01(invN * sumId), Is = clip01(invN * sumIs);
            if (kind == RT_LIGHT_SPOT) {
                Id = clip01(factor * Id);
                Is = clip01(factor * Is);
            }
            const V3 add = clip01(sumKd * Id + sumKs * Is);
            if (lit) surf = surf + add;
        }
        const V3 amb = {sc->ambient[0], sc->ambient[1], sc->ambient[2]};
        surf = clip01(surf + clip01(ld3(s_mat[sid].ambient) * amb));
        if (active) {
            my_hits += 1;
            my_samples += hit_samples;
            store_smooth(D.sr, e, surf.x, true); store_smooth(D.sg, e, surf.y, true); store_smooth(D.sb, e, surf.z, true);   // this pass only runs in mixed precision (D.col32)
            if (redo) {
                // some shadow decision of this hit could not be certified in float32 (or the hit had more
                // candidate spheres than the sample loop holds): k_surface_redo evaluates the hit again
                my_fallback += hit_samples;
                D.redo_list[atomicAdd(&D.flags[1], 1)] = (uint32_t)e;
            }
        }
    }
    if (my_hits) {
        atomicAdd(&s_work[0], my_hits);
        atomicAdd(&s_work[1], my_samples);
        if (my_fallback) atomicAdd(&s_work[2], my_fallback);
        if (my_mismatches) atomicAdd(&s_work[3], my_mismatches);
    }
    __syncthreads();
    if (tid < 3 && s_work[tid]) atomicAdd(&D.work[tid], (unsigned long long)s_work[tid]);
    if (tid == 3 && s_work[3]) atomicAdd(&D.work[5], (unsigned long long)s_work[3]);
}

}  // namespace rt
