// Bottom-up colour combination (k_combine) and the 8-bit output step (k_quantise).
#pragma once
#include "rt_geometry.hip"

namespace rt {

// ---------------------------------------------------------------------------------------------
// Bottom-up colour combination for one level (reference raytrace.py:155-170); the
// deepest level runs first.  A ray's colour replaces its surface colour in place.
// Two tiles per loop iteration; per iteration the loads come in two sets (the info words
// and counters, then everything the info words select), each set issued together.
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
    const LevelTables T = stage_level_tables(D, level, s_dyn, tid, blockIdx.x * WPB * q, min(nt, (int)(blockIdx.x + 1) * WPB * q) - 1);
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
        // first set of loads: info words and the running child counts of the tiles
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
        // second set of loads: what the info words select
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
                if (level == 0) {   // a primary ray's colour is the pixel; no parent reads it
                    double* out = D.image + (size_t)(first[u] + lane) * 3;
                    out[0] = col.x; out[1] = col.y; out[2] = col.z;
                } else {
                    D.cr[idx] = col.x; D.cg[idx] = col.y; D.cb[idx] = col.z;
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
