// HIP kernels of the level-synchronous wavefront ray tracer (gfx950, wave64).
//
// One frame is:  k_init, then for every bounce level  k_trace -> k_scan -> k_link,
// then from the deepest level upwards  k_shade.
//
//  * A level's rays live in the arena in tiles of 256, grouped by tree node
//    (a node = all rays that followed the same branch sequence, the batch of one
//    recursive call in reference raytrace.py:19).  Inside a node rays stay in
//    ascending pixel order, as boolean-mask selection keeps them in the reference.
//  * k_trace derives each ray from its parent's hit record, finds the nearest
//    sphere and stores the hit record; it counts, per tile, the children each
//    branch will get.  k_scan turns the counts into node sizes and slots,
//    k_link writes for every child slot the index of its parent (ordered
//    compaction with wave ballots), so the next level's rays are dense.
//  * k_shade evaluates lights and combines a ray's surface colour with its
//    children's colours (bottom-up, exactly the reference's return path).
#include "rt_internal.h"

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
// v / |v| with the norm as sqrt of the left-to-right sum of squares and three
// divisions, the way numpy.linalg.norm + array division round.
__device__ __forceinline__ V3 normalize(V3 v) {
    double len = sqrt(dot(v, v));
    return {v.x / len, v.y / len, v.z / len};
}

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
    return D.blk_tot[((size_t)level * NB + t / q) * NCNT + c] + D.tile_run[(size_t)(tbase + t) * NCNT + c];
}

// Roots of |o + t d - c|^2 = r^2 for unit d; returns false if there is none; lo <= hi.
//
// Every hit / miss decision of the frame hangs on the sign of this discriminant, and
// scenes do contain rays that are exactly tangent to a sphere, so the arithmetic
// follows reference objects.py:60-66 and :242-251 operation by operation (this file is
// compiled with -ffp-contract=off; only the explicit fma() calls fuse):
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
    const double half_root = sqrt(disc) / 2;
    const double mid = -B / 2;
    const double x1 = mid + half_root, x2 = mid - half_root;
    lo = fmin(x1, x2);
    hi = fmax(x1, x2);
    return true;
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
    __syncthreads();
    if (tid == 0) {
        int nt = (D.npix + TILE - 1) / TILE;
        D.flags[0] = 0;
        D.work[0] = 0;
        D.work[1] = 0;
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
// Nearest hit + hit record for one level.
__global__ __launch_bounds__(TILE) void k_trace(Dev D, int level) {
    __shared__ int s_tile_end[MAX_LEVEL_NODES];
    __shared__ int s_wcnt[4][NCNT];
    __shared__ double s_sph[SPH_CHUNK][4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = D.level_start[level], nn = D.level_start[level + 1] - k0;
    for (int i = tid; i < nn; i += TILE) s_tile_end[i] = D.node_tile0[k0 + i] + D.node_ntiles[k0 + i];

    const int nt = D.lvl_ntiles[level];
    const int tbase = D.lvl_tilebase[level];
    const int q = (nt + NB - 1) / NB;
    const int t_begin = blockIdx.x * q;
    const int t_end = min(nt, t_begin + q);
    unsigned run = 0;  // used by threads 0..NCNT-1

    const SceneConst& sc = *D.sc;
    const int S = D.S;
    const bool single_chunk = S <= SPH_CHUNK;
    if (single_chunk && t_begin < t_end) {
        for (int i = tid; i < S; i += TILE) {
            const double* g = D.geom + (size_t)i * 4;
            s_sph[i][0] = g[0]; s_sph[i][1] = g[1]; s_sph[i][2] = g[2]; s_sph[i][3] = g[3] * g[3];
        }
    }
    __syncthreads();

    const double eps = sc.eps, air = sc.air;

    for (int t = t_begin; t < t_end; ++t) {
        const int k = k0 + find_node(s_tile_end, nn, t);
        const int first = (t - D.node_tile0[k]) * TILE;
        const int cnt = min(TILE, D.node_n[k] - first);
        const bool dead = D.node_nglobal[k] < sc.small_num;
        const int a = D.node_a[k], b = D.node_b[k];
        const uint32_t idx = (uint32_t)(tbase + t) * TILE + tid;
        const bool valid = tid < cnt;
        bool active = valid && !dead;

        // ---- derive the ray ------------------------------------------------------------------
        V3 o = {0, 0, 0}, d = {1, 0, 0};
        bool live = true;
        if (valid) {
            if (level == 0) {
                // reference objects.py:45-51 and raytrace.py:7-12
                double X, Y;
                const int i = first + tid;
                if (D.X) { X = D.X[i]; Y = D.Y[i]; }
                else { X = (double)(i % sc.width); Y = (double)(D.row0 + i / sc.width); }
                double sx = X * sc.scr_w / sc.pix_x - sc.scr_w / 2;
                double sy = (sc.pix_y - Y) * sc.scr_h / sc.pix_y - sc.scr_h / 2;
                V3 pix = sx * ld3(sc.scr_x) + sy * ld3(sc.scr_y) + ld3(sc.scr_pos);
                o = ld3(sc.cam);
                d = normalize(pix - o);
            } else {
                const uint32_t p = D.link[idx];
                const uint32_t pinfo = D.info[p];
                const int psid = pinfo & I_SID_MASK;
                const bool pinside = pinfo & I_INSIDE;
                V3 Pb = {D.px[p], D.py[p], D.pz[p]};
                V3 pd = {D.dx[p], D.dy[p], D.dz[p]};
                V3 N = {D.nx[p], D.ny[p], D.nz[p]};
                double cosi = dot(N, pd);
                const int br = D.node_branch[k];
                if (br == 0) {
                    // transmitted ray (reference raytrace.py:198-212, :104-107); the same
                    // operations on the same stored values as in the parent's own pass
                    double ior = D.mat[(size_t)psid * RT_MAT_WIDTH + 12];
                    double mu = pinside ? ior / air : air / ior;
                    double kk = 1 - mu * mu * (1 - cosi * cosi);
                    double sk = sqrt(fmax(kk, 0.0));
                    d = sk * N + mu * (pd - cosi * N);
                    o = Pb + (2 * eps) * N;
                    live = pinfo & I_LIVE_TI;
                } else {
                    // mirror direction about the surface (reference raytrace.py:192-195);
                    // total internal reflection and plain reflection share it
                    d = pd - cosi * (2.0 * N);
                    o = Pb;
                    live = (br == 1) ? (pinfo & I_LIVE_TI) : (pinfo & I_LIVE_R);
                }
            }
        }

        // ---- nearest hit (reference objects.py:288-299) ------------------------------------
        const double d_o = dot(d, o);
        double best = __builtin_inf();
        int sid = -1;
        for (int c0 = 0; c0 < S; c0 += SPH_CHUNK) {
            const int cn = min(SPH_CHUNK, S - c0);
            if (!single_chunk) {
                __syncthreads();
                for (int i = tid; i < cn; i += TILE) {
                    const double* g = D.geom + (size_t)(c0 + i) * 4;
                    s_sph[i][0] = g[0]; s_sph[i][1] = g[1]; s_sph[i][2] = g[2]; s_sph[i][3] = g[3] * g[3];
                }
                __syncthreads();
            }
            if (active) {
                for (int i = 0; i < cn; ++i) {
                    double lo, hi;
                    if (sphere_roots(o, d, d_o, s_sph[i][0], s_sph[i][1], s_sph[i][2], s_sph[i][3], lo, hi)) {
                        double tt = lo > 0 ? lo : hi;
                        // ties go to the later sphere, as the reference's min_t == t update does
                        if (tt > 0 && tt <= best) { best = tt; sid = c0 + i; }
                    }
                }
            }
        }

        // ---- hit record (reference raytrace.py:53-101) -------------------------------------
        uint32_t info = 0;
        if (active && sid >= 0) {
            const double* g = D.geom + (size_t)sid * 4;
            const double* m = D.mat + (size_t)sid * RT_MAT_WIDTH;
            V3 P = o + best * d;
            V3 n = normalize(P - ld3(g));
            const bool inside = dot(d, n) > 0;
            const double ior = m[12];
            const double n1 = inside ? ior : air, n2 = inside ? air : ior;
            V3 N = inside ? n : -n;
            V3 Pb = P - eps * N;
            const double mu = n1 / n2;
            const double cosi = dot(N, d);
            const double kk = 1 - mu * mu * (1 - cosi * cosi);
            const bool trans = kk >= 0;
            V3 T = {0, 0, 0};
            if (trans) T = sqrt(kk) * N + mu * (d - cosi * N);  // reference raytrace.py:207-211
            double r0 = (n1 - n2) / (n1 + n2);
            r0 *= r0;
            const double cosf = (n2 >= n1) ? cosi : dot(T, N);
            const double om = 1 - cosf;
            const double om2 = om * om;
            double R = r0 + (1 - r0) * (om2 * om2 * om);
            if (!trans) R = 1;
            const double floor_ = inside ? 0.0 : m[10];
            const double A = floor_ + (m[13] - floor_) * R;

            double through = 0;
            if (b != 0 && trans && !inside) {
                // distance to the far side of the entered sphere (reference raytrace.py:117-118)
                double lo, hi;
                through = -1;
                if (sphere_roots(Pb, T, dot(T, Pb), g[0], g[1], g[2], g[3] * g[3], lo, hi))
                    through = hi > 0 ? hi : (lo > 0 ? lo : -1.0);
            }
            info = (uint32_t)sid | I_HIT;
            if (inside) info |= I_INSIDE;
            if (b != 0 && trans) info |= I_CHILD_T;
            if (a != 0 && b != 0 && !trans) info |= I_CHILD_I;
            if (a != 0) info |= I_CHILD_R;
            if (live) {
                info |= I_LIVE;
                if (m[11] * (1 - A) != 0) info |= I_LIVE_TI;
                if (A != 0) info |= I_LIVE_R;
            }
            for (int l = 0; l < D.n_lights; ++l) {
                const RtLight& Lt = sc.lights[l];
                if (Lt.kind == RT_LIGHT_SPOT) {
                    // reference objects.py:204-213
                    V3 rel = Pb - ld3(Lt.position);
                    double x = dot(ld3(Lt.x_hat), rel), y = dot(ld3(Lt.y_hat), rel), z = dot(ld3(Lt.normal), rel);
                    double rho = sqrt(x * x + y * y);
                    double edge = Lt.radius + Lt.tan_outer * fabs(z);
                    if (rho <= edge) info |= 1u << (I_BEAM_SHIFT + l);
                }
            }
            D.px[idx] = Pb.x; D.py[idx] = Pb.y; D.pz[idx] = Pb.z;
            D.dx[idx] = d.x; D.dy[idx] = d.y; D.dz[idx] = d.z;
            D.nx[idx] = N.x; D.ny[idx] = N.y; D.nz[idx] = N.z;
            D.A[idx] = A;
            D.dist[idx] = through;
        }
        if (valid) D.info[idx] = info;

        // ---- per-tile counters --------------------------------------------------------------
        {
            unsigned long long m_t = __ballot((info & I_CHILD_T) != 0);
            unsigned long long m_i = __ballot((info & I_CHILD_I) != 0);
            unsigned long long m_r = __ballot((info & I_CHILD_R) != 0);
            unsigned long long m_h = __ballot((info & I_HIT) != 0);
            if (lane == 0) {
                s_wcnt[wave][C_T] = __popcll(m_t);
                s_wcnt[wave][C_I] = __popcll(m_i);
                s_wcnt[wave][C_R] = __popcll(m_r);
                s_wcnt[wave][C_HIT] = __popcll(m_h);
            }
            for (int l = 0; l < RT_MAX_LIGHTS; ++l) {
                unsigned long long m_b = __ballot(((info >> (I_BEAM_SHIFT + l)) & 1u) != 0);
                if (lane == 0) s_wcnt[wave][C_BEAM0 + l] = __popcll(m_b);
            }
        }
        __syncthreads();
        if (tid < NCNT) {
            D.tile_run[(size_t)(tbase + t) * NCNT + tid] = run;
            run += s_wcnt[0][tid] + s_wcnt[1][tid] + s_wcnt[2][tid] + s_wcnt[3][tid];
        }
        __syncthreads();
    }
    if (tid < NCNT) D.blk_tot[((size_t)level * NB + blockIdx.x) * NCNT + tid] = run;
}

// ---------------------------------------------------------------------------------------------
// Counts -> node sizes, slots and the next level's layout.  One workgroup.
__global__ __launch_bounds__(1024) void k_scan(Dev D, int level) {
    __shared__ unsigned s_grand[NCNT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = D.lvl_ntiles[level];
    const int tbase = D.lvl_tilebase[level];
    const int q = max(1, (nt + NB - 1) / NB);

    // 1. exclusive scan of the workgroup totals, one wave per counter
    if (wave < NCNT) {
        unsigned carry = 0;
        for (int b0 = 0; b0 < NB; b0 += 64) {
            uint32_t* p = D.blk_tot + ((size_t)level * NB + b0 + lane) * NCNT + wave;
            unsigned v = *p;
            unsigned incl = v;
            for (int off = 1; off < 64; off <<= 1) {
                unsigned up = __shfl_up(incl, off);
                if (lane >= off) incl += up;
            }
            *p = carry + incl - v;
            carry += __shfl(incl, 63);
        }
        if (lane == 0) s_grand[wave] = carry;
    }
    __syncthreads();

    // 2. per-node prefix and totals
    const int k0 = D.level_start[level], nn = D.level_start[level + 1] - k0;
    for (int i = tid; i < nn * NCNT; i += 1024) {
        const int kk = i / NCNT, c = i % NCNT, k = k0 + kk;
        const int T0 = D.node_tile0[k];
        const unsigned s = (T0 < nt) ? gscan(D, level, tbase, q, T0, c) : s_grand[c];
        unsigned e = s_grand[c];
        if (kk + 1 < nn) {
            const int T1 = D.node_tile0[k + 1];
            if (T1 < nt) e = gscan(D, level, tbase, q, T1, c);
        }
        const unsigned tot = e - s;
        D.node_pre[(size_t)k * NCNT + c] = s;
        if (c < 3) {
            const int child = D.node_child[k * 3 + c];
            if (child >= 0) {
                D.node_n[child] = (int)tot;
                D.node_nglobal[child] = (int)tot;
            }
        } else if (c == C_HIT) {
            D.node_hits[k] = (int)tot;
        } else {
            D.node_beam[k * RT_MAX_LIGHTS + (c - C_BEAM0)] = (int)tot;
        }
    }
    __syncthreads();

    // 3. layout of the next level: every node gets whole tiles
    if (level + 1 < D.n_levels && tid == 0) {
        const int c0 = D.level_start[level + 1], c1 = D.level_start[level + 2];
        const int next_base = tbase + nt;
        int running = 0;
        for (int k = c0; k < c1; ++k) {
            const int ntile = (D.node_n[k] + TILE - 1) / TILE;
            D.node_tile0[k] = running;
            D.node_ntiles[k] = ntile;
            D.node_start[k] = (uint32_t)(next_base + running) * TILE;
            running += ntile;
        }
        if ((uint32_t)(next_base + running) > D.cap_tiles) {
            // arena too small: drop the deeper levels and tell the host
            D.flags[0] = 1;
            for (int k = c0; k < c1; ++k) {
                D.node_n[k] = 0;
                D.node_nglobal[k] = 0;
                D.node_tile0[k] = 0;
                D.node_ntiles[k] = 0;
            }
            running = 0;
        }
        D.lvl_ntiles[level + 1] = running;
        D.lvl_tilebase[level + 1] = next_base;
    }
}

// ---------------------------------------------------------------------------------------------
// Ordered compaction: every child slot of the next level learns its parent.
__global__ __launch_bounds__(TILE) void k_link(Dev D, int level) {
    __shared__ int s_tile_end[MAX_LEVEL_NODES];
    __shared__ int s_wcnt[4][3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (D.flags[0]) return;  // arena overflow: child slots do not exist
    const int k0 = D.level_start[level], nn = D.level_start[level + 1] - k0;
    for (int i = tid; i < nn; i += TILE) s_tile_end[i] = D.node_tile0[k0 + i] + D.node_ntiles[k0 + i];
    const int nt = D.lvl_ntiles[level];
    const int tbase = D.lvl_tilebase[level];
    const int q = (nt + NB - 1) / NB;
    const int t_begin = blockIdx.x * q;
    const int t_end = min(nt, t_begin + q);
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
        const int k = k0 + find_node(s_tile_end, nn, t);
        const int first = (t - D.node_tile0[k]) * TILE;
        const int cnt = min(TILE, D.node_n[k] - first);
        const uint32_t idx = (uint32_t)(tbase + t) * TILE + tid;
        const uint32_t info = (tid < cnt) ? D.info[idx] : 0u;
        unsigned long long m[3];
        for (int c = 0; c < 3; ++c) {
            m[c] = __ballot(((info >> (18 + c)) & 1u) != 0);
            if (lane == 0) s_wcnt[wave][c] = __popcll(m[c]);
        }
        __syncthreads();
        for (int c = 0; c < 3; ++c) {
            if ((info >> (18 + c)) & 1u) {
                int rank = lanes_below(m[c]);
                for (int w = 0; w < wave; ++w) rank += s_wcnt[w][c];
                const int child = D.node_child[k * 3 + c];
                const uint32_t slot = D.node_start[child] +
                                      (gscan(D, level, tbase, q, t, c) - D.node_pre[(size_t)k * NCNT + c]) + rank;
                D.link[slot] = idx;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Shading helpers.

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
__device__ __forceinline__ V3 shadow_multiplier(const Dev& D, const ShadowSphere* s_sh, V3 o, V3 d) {
    V3 mult = {1, 1, 1};
    const int S = D.S;
    const double d_o = dot(d, o);
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

// Surface colour of one hit under all lights (reference raytrace.py:136-152,
// :215-234 and objects.py:94-235).  j_hit / j_beam: rank of the hit among the
// node's hits / in-beam hits; h_hit / h_beam: those counts.
template <bool LDS>
__device__ V3 surface_color(const Dev& D, const ShadowSphere* s_sh, int node, uint32_t info, V3 Pb, V3 d, V3 n,
                            const double* m, int j_hit, int h_hit, const int* j_beam, unsigned& n_samples) {
    const SceneConst& sc = *D.sc;
    const V3 kdiff = ld3(m + 3), kspec = ld3(m + 6);
    const double shin = m[9];
    V3 surf = {0, 0, 0};
    for (int l = 0; l < D.n_lights; ++l) {
        const RtLight& Lt = sc.lights[l];
        const V3 lpos = ld3(Lt.position);
        const V3 dcol = ld3(Lt.diffuse_color), scol = ld3(Lt.specular_color);
        if (Lt.kind == RT_LIGHT_POINT) {
            V3 Lv = lpos - Pb;
            const double d2 = dot(Lv, Lv);
            const double dist = sqrt(d2);
            V3 Lh = {Lv.x / dist, Lv.y / dist, Lv.z / dist};
            V3 mult = shadow_multiplier<LDS>(D, s_sh, Pb, Lh);
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
        int j = j_hit, h = h_hit;
        double factor = 1.0;
        if (Lt.kind == RT_LIGHT_SPOT) {
            if (!((info >> (I_BEAM_SHIFT + l)) & 1u)) continue;  // zero light vectors, zero intensity
            j = j_beam[l];
            h = D.node_beam[node * RT_MAX_LIGHTS + l];
            V3 rel = Pb - lpos;
            const double x = dot(ld3(Lt.x_hat), rel), y = dot(ld3(Lt.y_hat), rel), z = dot(ld3(Lt.normal), rel);
            const double rho = sqrt(x * x + y * y);
            const double allowed = Lt.radius + Lt.tan_angle * fabs(z);
            const double edge = Lt.radius + Lt.tan_outer * fabs(z);
            factor = ease(Lt.easing, rho, allowed, edge - allowed);
        }
        const int N = Lt.n_points;
        n_samples += N;
        const int64_t off = D.sample_off[(size_t)node * RT_MAX_LIGHTS + l];
        const double* u_theta = D.samples + off + j;
        const double* u_rad = D.samples + off + (int64_t)N * h + j;
        const V3 xh = ld3(Lt.x_hat), yh = ld3(Lt.y_hat);
        const double per_sample = Lt.intensity / N;
        V3 sumId = {0, 0, 0}, sumIs = {0, 0, 0}, sumKd = {0, 0, 0}, sumKs = {0, 0, 0};
        for (int s = 0; s < N; ++s) {
            const double theta = u_theta[(int64_t)s * h] * 6.283185307179586;
            const double rad = sqrt(u_rad[(int64_t)s * h]) * Lt.radius;
            double sn, cs;
            sincos(theta, &sn, &cs);
            V3 lp = lpos + rad * (cs * xh + sn * yh);
            V3 Lv = lp - Pb;
            const double dist = sqrt(dot(Lv, Lv));
            V3 Lh = {Lv.x / dist, Lv.y / dist, Lv.z / dist};
            V3 mult = shadow_multiplier<LDS>(D, s_sh, Pb, Lh);
            const double fall = per_sample / (dist * dist);
            sumId = sumId + clip01((fall * dcol) * mult);
            sumIs = sumIs + clip01((fall * scol) * mult);
            const double nl = dot(n, Lh);
            V3 Rv = Lh - (2 * nl) * n;
            const double rd = fabs(dot(Rv, d));
            sumKd = sumKd + clip01(nl * kdiff);
            sumKs = sumKs + clip01(pow_shininess(rd, shin) * kspec);
        }
        const double invN = 1.0 / N;
        V3 Id = clip01(invN * sumId), Is = clip01(invN * sumIs);
        if (Lt.kind == RT_LIGHT_SPOT) {
            Id = clip01(factor * Id);
            Is = clip01(factor * Is);
        }
        surf = surf + clip01(sumKd * Id + sumKs * Is);
    }
    surf = surf + clip01(ld3(m) * ld3(sc.ambient));
    return clip01(surf);
}

// ---------------------------------------------------------------------------------------------
// Lighting + bottom-up combination for one level (deepest level first).
template <bool LDS>
__global__ __launch_bounds__(TILE) void k_shade(Dev D, int level) {
    __shared__ int s_tile_end[MAX_LEVEL_NODES];
    __shared__ int s_wcnt[4][NCNT];
    __shared__ ShadowSphere s_sh[LDS ? SHADE_LDS_SPHERES : 1];
    __shared__ unsigned s_work[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (D.flags[0]) return;  // arena overflow: the frame is invalid, the host reports it
    if (tid < 2) s_work[tid] = 0;
    unsigned my_hits = 0, my_samples = 0;
    const int k0 = D.level_start[level], nn = D.level_start[level + 1] - k0;
    for (int i = tid; i < nn; i += TILE) s_tile_end[i] = D.node_tile0[k0 + i] + D.node_ntiles[k0 + i];
    const int nt = D.lvl_ntiles[level];
    const int tbase = D.lvl_tilebase[level];
    const int q = (nt + NB - 1) / NB;
    const int t_begin = blockIdx.x * q;
    const int t_end = min(nt, t_begin + q);
    if (LDS && t_begin < t_end) {
        for (int i = tid; i < D.S; i += TILE) {
            const double* g = D.geom + (size_t)i * 4;
            const double* m = D.mat + (size_t)i * RT_MAT_WIDTH;
            s_sh[i] = {g[0], g[1], g[2], g[3] * g[3], m[11], m[14] * m[15], m[14] * m[16], m[14] * m[17]};
        }
    }
    __syncthreads();
    const SceneConst& sc = *D.sc;
    const V3 bg = ld3(sc.bg);

    for (int t = t_begin; t < t_end; ++t) {
        const int k = k0 + find_node(s_tile_end, nn, t);
        const int first = (t - D.node_tile0[k]) * TILE;
        const int cnt = min(TILE, D.node_n[k] - first);
        const uint32_t idx = (uint32_t)(tbase + t) * TILE + tid;
        const bool valid = tid < cnt;
        const uint32_t info = valid ? D.info[idx] : 0u;

        // ranks inside the tile for every counter
        unsigned long long masks[NCNT];
        for (int c = 0; c < NCNT; ++c) {
            const int bit = (c < 3) ? 18 + c : (c == C_HIT ? 16 : I_BEAM_SHIFT + (c - C_BEAM0));
            masks[c] = __ballot(((info >> bit) & 1u) != 0);
            if (lane == 0) s_wcnt[wave][c] = __popcll(masks[c]);
        }
        __syncthreads();

        if (valid) {
            V3 col = bg;
            if (info & I_HIT) {
                col = {0, 0, 0};
                if (info & I_LIVE) {
                    int rank[NCNT];
                    for (int c = 0; c < NCNT; ++c) {
                        int r = lanes_below(masks[c]);
                        for (int w = 0; w < wave; ++w) r += s_wcnt[w][c];
                        rank[c] = r + (int)(gscan(D, level, tbase, q, t, c) - D.node_pre[(size_t)k * NCNT + c]);
                    }
                    const int sid = info & I_SID_MASK;
                    const bool inside = info & I_INSIDE;
                    const double* g = D.geom + (size_t)sid * 4;
                    const double* m = D.mat + (size_t)sid * RT_MAT_WIDTH;
                    V3 Pb = {D.px[idx], D.py[idx], D.pz[idx]};
                    V3 d = {D.dx[idx], D.dy[idx], D.dz[idx]};
                    const double A = D.A[idx];
                    const double tr = m[11];
                    const double w_surf = (1 - tr) * (1 - A);
                    V3 surf = {0, 0, 0};
                    if (w_surf != 0) {
                        V3 N = {D.nx[idx], D.ny[idx], D.nz[idx]};
                        V3 n = inside ? N : -N;  // outward normal
                        surf = surface_color<LDS>(D, s_sh, k, info, Pb, d, n, m, rank[C_HIT], D.node_hits[k],
                                                  rank + C_BEAM0, my_samples);
                        my_hits += 1;
                    }
                    V3 refr = {0, 0, 0}, refl = {0, 0, 0};
                    if ((info & I_LIVE_TI) && (info & (I_CHILD_T | I_CHILD_I))) {
                        const int c = (info & I_CHILD_T) ? C_T : C_I;
                        const uint32_t slot = D.node_start[D.node_child[k * 3 + c]] + rank[c];
                        refr = {D.cr[slot], D.cg[slot], D.cb[slot]};
                        if ((info & I_CHILD_T) && !inside) {
                            // absorption inside the entered sphere (reference raytrace.py:119-120)
                            const double through = D.dist[idx];
                            const double att = m[14];
                            refr.x *= exp(-att * m[15] * through);
                            refr.y *= exp(-att * m[16] * through);
                            refr.z *= exp(-att * m[17] * through);
                        }
                    }
                    if ((info & I_LIVE_R) && (info & I_CHILD_R)) {
                        const uint32_t slot = D.node_start[D.node_child[k * 3 + C_R]] + rank[C_R];
                        refl = {D.cr[slot], D.cg[slot], D.cb[slot]};
                    }
                    // reference raytrace.py:155-170
                    col = clip01(w_surf * surf + (tr * (1 - A)) * refr + A * refl);
                }
            }
            D.cr[idx] = col.x; D.cg[idx] = col.y; D.cb[idx] = col.z;
            if (level == 0) {
                double* out = D.image + (size_t)(first + tid) * 3;
                out[0] = col.x; out[1] = col.y; out[2] = col.z;
            }
        }
        __syncthreads();
    }
    if (my_hits) {
        atomicAdd(&s_work[0], my_hits);
        atomicAdd(&s_work[1], my_samples);
    }
    __syncthreads();
    if (tid < 2 && s_work[tid]) atomicAdd(&D.work[tid], (unsigned long long)s_work[tid]);
}

template __global__ void k_shade<true>(Dev, int);
template __global__ void k_shade<false>(Dev, int);

}  // namespace rt
