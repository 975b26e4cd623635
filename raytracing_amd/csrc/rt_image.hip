// Bottom-up colour combination (k_combine) and the 8-bit output step (k_quantise).
#pragma once
#include "rt_geometry.hip"

namespace rt {

// ---------------------------------------------------------------------------------------------
// Bottom-up colour combination for one level (reference raytrace.py:155-170); the
// deepest level runs first.  A ray's colour replaces its surface colour in place.
//
// Only rays whose colour reaches the pixel ("live") have anything to combine, and in most
// scenes most tiles hold none (every ray below an opaque, non-reflecting hit is traced for the
// batch and hit counts only).  A wave therefore first reads the running count of live hits that
// k_trace left at each of its tiles and visits only the tiles where it changes.  The tiles it
// visits go through a two-stage software pipeline, three tiles per stage: while the colours one
// group selected are in flight, the info words of the next group are requested already.
#ifndef RT_COMBINE_TILES
#define RT_COMBINE_TILES 3
#endif
constexpr int COMBINE_TILES = RT_COMBINE_TILES;
struct CombineSphere {
    double tr, kr, kg, kb;   // transparency, -attenuation * absorption colour
};
struct CombineTile {   // first stage of the pipeline: what is known / requested about a tile
    int t;             // tile (relative to the level), -1: none
    int kn, first;     // node of the tile (relative to the level), index of its first ray in the node
    uint32_t info;
    unsigned trun;     // lanes 0-2: running child counts at the tile's start
};
template <bool SMALL, bool C32>   // SMALL: at most SPH_CHUNK_SMALL spheres, their constants in LDS; C32: colours, A, dist kept as float32
__global__ __launch_bounds__(BLOCK, RT_COMBINE_WAVES) void k_combine(Dev D, int level) {
    extern __shared__ int s_dyn[];
    __shared__ CombineSphere s_cs[SMALL ? SPH_CHUNK_SMALL : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (D.flags[0]) return;
    const int k0 = D.level_start[level], nn = D.level_start[level + 1] - k0;
    const int nt = D.lvl_ntiles[level];
    const int tbase = D.lvl_tilebase[level];
    const int q = tiles_per_wave(D, nt);
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
    const int shade_base = D.lvl_shade_base[level];
    unsigned wave_base = 0;   // lane c < 3: children of kind c before this wave's first tile; lane C_SHADE: shading-list entries
    if (lane < 3 || lane == C_SHADE)
        wave_base = D.blk_tot[((size_t)level * NCNT + lane) * NB + blockIdx.x] + D.wave_off[((size_t)level * NCNT + lane) * NW + wid];
    __syncthreads();

    // ---- which tiles to visit ---------------------------------------------------------------
    // A live ray that misses everything still needs the background colour stored for its parent,
    // so the count that matters is that of live rays of either kind (column C_ANYLIVE).
    int chunk0 = t_begin;                      // first tile of the current chunk of 64 tiles
    unsigned long long todo = 0;               // tiles of the chunk still to visit
    auto next_chunk = [&]() {
        while (todo == 0 && chunk0 < t_end) {
            const int t = chunk0 + lane;
            bool visit = false;
            if (t < t_end) {
                const unsigned here = D.tile_run[(size_t)(tbase + t) * NCNT + C_ANYLIVE];
                // the running count is local to the wave that traced these tiles (the wave with this
                // index): the next tile's value tells this tile's count; the last tile is always visited
                visit = t + 1 >= t_end || D.tile_run[(size_t)(tbase + t + 1) * NCNT + C_ANYLIVE] != here;
            }
            todo = __ballot(visit);
            if (todo == 0) chunk0 += 64;
        }
    };
    int kk = (t_begin < t_end) ? find_node(T.node, nn, t_begin) : 0;
    auto fetch = [&]() -> CombineTile {   // takes the next tile to visit and requests its info words
        CombineTile c;
        c.t = -1; c.kn = 0; c.first = 0; c.info = I_NO_HIT; c.trun = 0;
        next_chunk();
        if (todo != 0) {
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int t = chunk0 + j;
            if (todo == 0) chunk0 += 64;
            c.t = t;
            while (T.node[kk].tile_end <= t) ++kk;
            kk = __builtin_amdgcn_readfirstlane(kk);
            c.kn = kk;
            const NodeRec r = T.node[kk];
            c.first = (t - __builtin_amdgcn_readfirstlane(r.tile0)) * TILE;
            const int cnt = min(TILE, __builtin_amdgcn_readfirstlane(r.n) - c.first);
            if (lane < cnt) c.info = D.info[(uint32_t)(tbase + t) * TILE + lane];
            if ((has_children && lane < 3) || lane == C_SHADE) c.trun = D.tile_run[(size_t)(tbase + t) * NCNT + lane];
        }
        return c;
    };

    // One pipeline step: second set of loads for the group `cur`, then the first set of the group
    // after it (into `nxt`), then arithmetic and stores for `cur`.
    // Loaded values stay in registers of the stored type until the arithmetic: a conversion at the
    // load would make the wave wait for that load at once.
    typedef typename SmoothType<C32>::type Stored;
    struct Stored3 { Stored x, y, z; };
    auto load3 = [](const double* bx, const double* by, const double* bz, uint32_t i) -> Stored3 {
        return {reinterpret_cast<const Stored*>(bx)[i], reinterpret_cast<const Stored*>(by)[i], reinterpret_cast<const Stored*>(bz)[i]};
    };
    auto widen = [](const Stored3& v) -> V3 { return {(double)v.x, (double)v.y, (double)v.z}; };
    auto step = [&](const CombineTile (&cur)[COMBINE_TILES], CombineTile (&nxt)[COMBINE_TILES]) {
        Stored A[COMBINE_TILES], through[COMBINE_TILES];
        Stored3 surf[COMBINE_TILES], refr[COMBINE_TILES], refl[COMBINE_TILES];
#pragma unroll
        for (int u = 0; u < COMBINE_TILES; ++u) {
            const uint32_t idx = (uint32_t)(tbase + max(cur[u].t, 0)) * TILE + lane;
            const uint32_t in = cur[u].info;
            A[u] = 0; through[u] = 0;
            surf[u] = {0, 0, 0}; refr[u] = {0, 0, 0}; refl[u] = {0, 0, 0};
            const bool want_ti = (in & I_LIVE_TI) && (in & (I_CHILD_T | I_CHILD_I));
            const bool want_r = (in & I_LIVE_R) && (in & I_CHILD_R);
            uint32_t slot[3] = {0, 0, 0};   // arena index of the ray's child of each kind
            if (has_children && __ballot(want_ti || want_r) != 0) {
                const int base = (int)(wave_base + cur[u].trun - T.pre[cur[u].kn * 16 + (lane & 15)]);
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    slot[c] = T.child[cur[u].kn * 4 + c] + LANE(base, c) + lanes_below(__ballot(((in >> (18 + c)) & 1u) != 0));
            }
            // the shading-list entry of a lit hit: the level's first entry + entries before this tile + lit hits in lower lanes
            const unsigned long long lit = __ballot((in & I_SHADE) != 0);
            uint32_t entry = 0;
            if (lit != 0)
                entry = (uint32_t)shade_base + LANE((int)(wave_base + cur[u].trun), C_SHADE) + lanes_below(lit);
            if ((in & I_LIVE) && is_hit(in)) {
                A[u] = reinterpret_cast<const Stored*>(D.A)[idx];
                if (in & I_SHADE) surf[u] = load3(D.sr, D.sg, D.sb, entry);
                if (want_ti) {
                    const uint32_t sl = (in & I_CHILD_T) ? slot[C_T] : slot[C_I];
                    refr[u] = load3(D.cr, D.cg, D.cb, sl);
                    if ((in & I_CHILD_T) && !(in & I_INSIDE)) through[u] = reinterpret_cast<const Stored*>(D.dist)[idx];
                }
                if (want_r) {
                    const uint32_t sl = slot[C_R];
                    refl[u] = load3(D.cr, D.cg, D.cb, sl);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < COMBINE_TILES; ++u) nxt[u] = fetch();
        // arithmetic and stores
#pragma unroll
        for (int u = 0; u < COMBINE_TILES; ++u) {
            if (cur[u].t < 0) continue;
            const uint32_t idx = (uint32_t)(tbase + cur[u].t) * TILE + lane;
            const uint32_t in = cur[u].info;
            if (in & I_LIVE) {   // nobody reads the colour of a ray whose weight in the pixel is zero
                V3 col = bg;
                if (is_hit(in)) {
                    const int sid = in & I_SID_MASK;
                    const CombineSphere cs = SMALL ? s_cs[sid] : sphere_consts(sid);
                    V3 rf = widen(refr[u]);
                    const double a = (double)A[u];
                    if ((in & I_LIVE_TI) && (in & I_CHILD_T) && !(in & I_INSIDE)) {
                        // absorption inside the entered sphere (reference raytrace.py:119-120)
                        const double len = (double)through[u];
                        rf.x *= exp(cs.kr * len);
                        rf.y *= exp(cs.kg * len);
                        rf.z *= exp(cs.kb * len);
                    }
                    col = clip01(((1 - cs.tr) * (1 - a)) * widen(surf[u]) + (cs.tr * (1 - a)) * rf + a * widen(refl[u]));
                }
                if (level == 0) {   // a primary ray's colour is the pixel; no parent reads it
                    double* out = D.image + (size_t)(cur[u].first + lane) * 3;
                    out[0] = col.x; out[1] = col.y; out[2] = col.z;
                } else {
                    store_smooth(D.cr, idx, col.x, C32); store_smooth(D.cg, idx, col.y, C32); store_smooth(D.cb, idx, col.z, C32);
                }
            }
        }
    };

    // two register sets, so that no register with a load in flight is ever copied
    CombineTile pa[COMBINE_TILES], pb[COMBINE_TILES];
#pragma unroll
    for (int u = 0; u < COMBINE_TILES; ++u) pa[u] = fetch();
    while (pa[0].t >= 0) {
        step(pa, pb);
        if (pb[0].t < 0) break;
        step(pb, pa);
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
