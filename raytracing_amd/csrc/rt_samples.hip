// Disk-light samples: packing of uploaded or generated stream values (k_pack) and NumPy's legacy
// MT19937 stream on the device (k_mt_generate, k_mt_jump).
#pragma once
#include "rt_math.h"

namespace rt {

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
    typedef double Raw;    // what is loaded for a value, and the value it becomes
    __device__ __forceinline__ Raw load(int64_t i) const { return chunk[i]; }
    static __device__ __forceinline__ double value(Raw r) { return r; }
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
    typedef uint2 Raw;
    __device__ __forceinline__ Raw load(int64_t i) const { return make_uint2(raw[2 * i], raw[2 * i + 1]); }
    static __device__ __forceinline__ double value(Raw r) {
        const uint32_t hi = mt_temper(r.x) >> 5, lo = mt_temper(r.y) >> 6;
        return ((double)hi * 67108864.0 + (double)lo) / 9007199254740992.0;
    }
    __device__ __forceinline__ double operator()(int64_t i) const { return value(load(i)); }
};

// The point of the unit disk a pair of uniforms selects, as the float32 sample pass reads it
// (reference objects.py:132-135): (sqrt(u_radius) cos(2 pi u_angle), sqrt(u_radius) sin(2 pi u_angle)),
// evaluated in float64 and rounded to float.
__device__ __forceinline__ void disk_point(double u_angle, double u_radius, float& x, float& y) {
    double sn, cs;
    sincospi(2.0 * u_angle, &sn, &cs);
    const double rad = sqrt(u_radius);
    x = (float)(rad * cs);
    y = (float)(rad * sn);
}

// Where the samples of shading-list entry e under the sampled light l sit in the stream: the angle
// uniform of sample s at first + s * h, its radius uniform n_points * h further on (h: the node's hits, or
// its in-beam hits for a spotlight; the hit's rank among them is its column).  WHOLE_IMAGE: positions in
// the whole image's stream where the context has such a layout.  any == false: the entry draws none.
struct SampleColumn {
    int64_t first, h;
    bool any;
};
template <bool WHOLE_IMAGE>
__device__ __forceinline__ SampleColumn sample_column(const Dev& D, int64_t e, int node, uint32_t info, int l, const RtLight& Lt) {
    SampleColumn c;
    int64_t jj;
    c.any = true;
    if (Lt.kind == RT_LIGHT_SPOT) {
        c.any = (info >> (I_BEAM_SHIFT + l)) & 1u;
        c.h = D.node_beam[node * RT_MAX_LIGHTS + l];
        jj = D.list_j[1 + l][e];
    } else {
        c.h = D.node_hits[node];
        jj = D.list_j[0][e];
    }
    int64_t off = D.sample_off[(size_t)node * RT_MAX_LIGHTS + l];
    c.any = c.any && off >= 0;
    if (WHOLE_IMAGE && D.sample_global) {
        // this context holds columns [first column, + h) of a block that is wider in the whole image's stream
        const size_t KL = (size_t)D.K * RT_MAX_LIGHTS, at = (size_t)node * RT_MAX_LIGHTS + l;
        off = D.sample_global[at];
        c.h = D.sample_global[KL + at];
        jj += D.sample_global[2 * KL + at];
    }
    c.first = off + jj;
    return c;
}

// with_points (the piece [a, a + n) is the whole stream and the scene's samples go through the float32
// pass): the entry's disk points are made here as well, from the values just read, and k_disk_points
// need not read the packed uniforms again.  with_points == 2: only the points - the float64 readers
// (the re-evaluation of uncertified hits) take their uniforms from the raw words, Dev::raw_values_at.
// pass_of (with_points only, or null): the piece holds whole (node, light) blocks, those with
// pass_of[node * RT_MAX_LIGHTS + light] == pass; the others are left to the launches of their pieces.
template <typename Stream>
__global__ __launch_bounds__(BLOCK, RT_PACK_WAVES) void k_pack(Dev D, Stream chunk, int64_t a, int64_t n, int with_points,
                                                               const uint8_t* __restrict__ pass_of, int pass) {
    const SceneConst& sc = *D.sc;
    const int64_t b = a + n;
    const int64_t E = D.pack_E;
    const int64_t E_points = with_points ? ((E + TILE - 1) / TILE) * TILE : E;   // the last tile's unused slots hold the disk's centre
    for (int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x; e < E_points; e += (int64_t)gridDim.x * BLOCK) {
        if (e >= E) {
            for (int l = 0; l < D.n_lights; ++l) {
                if (sc.lights[l].kind == RT_LIGHT_POINT) continue;
                const int NP = (sc.lights[l].n_points + 1) >> 1;
                float4* points = reinterpret_cast<float4*>(D.packed32) + D.pack32_base[l] + (e / TILE) * NP * TILE + (e % TILE);
                for (int p = 0; p < NP; ++p) points[(int64_t)p * TILE] = make_float4(0, 0, 0, 0);
            }
            continue;
        }
        const uint32_t info = D.info[D.list_ray[e]];
        const int node = D.list_node[e];
        for (int l = 0; l < D.n_lights; ++l) {
            const RtLight& Lt = sc.lights[l];
            if (Lt.kind == RT_LIGHT_POINT) continue;
            const SampleColumn c = sample_column<Stream::WHOLE_IMAGE>(D, e, node, info, l, Lt);
            const int N = Lt.n_points;
            double* dst = D.packed + D.pack_base[l] + e;
            if (with_points) {
                if (c.any && pass_of && pass_of[node * RT_MAX_LIGHTS + l] != pass) continue;
                // both uniforms of a sample together: packed as below, and the point they select
                const int NP = (N + 1) >> 1;
                float4* points = reinterpret_cast<float4*>(D.packed32) + D.pack32_base[l] + (e / TILE) * NP * TILE + (e % TILE);
                // (the rows of a column lie far apart in the stream: the loads of the next pair are in
                // flight while the current one goes through sincospi and sqrt; two register sets in turn,
                // no loaded register is copied)
                typedef typename Stream::Raw Raw;
                struct Pair { Raw angle[2], radius[2]; };
                auto fetch = [&](int p, Pair& q) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int s = 2 * p + k;
                        if (s < N && c.any) {
                            q.angle[k] = chunk.load(c.first + s * c.h - a);
                            q.radius[k] = chunk.load(c.first + ((int64_t)N + s) * c.h - a);
                        }
                    }
                };
                auto emit = [&](int p, const Pair& q) {
                    float x[2] = {0, 0}, y[2] = {0, 0};
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int s = 2 * p + k;
                        if (s >= N || !c.any) break;
                        const double u_angle = Stream::value(q.angle[k]), u_radius = Stream::value(q.radius[k]);
                        if (with_points != 2) {
                            dst[s * E] = u_angle;
                            dst[D.pack_plane + s * E] = u_radius;
                        }
                        disk_point(u_angle, u_radius, x[k], y[k]);
                    }
                    points[(int64_t)p * TILE] = make_float4(x[0], x[1], y[0], y[1]);
                };
                Pair qa{}, qb{};
                fetch(0, qa);
                for (int p = 0; p < NP; p += 2) {
                    if (p + 1 < NP) fetch(p + 1, qb);
                    emit(p, qa);
                    if (p + 2 < NP) fetch(p + 2, qa);
                    if (p + 1 < NP) emit(p + 1, qb);
                }
                continue;
            }
            if (!c.any || c.first >= b || c.first + (2 * (int64_t)N - 1) * c.h < a) continue;   // none of its values in the piece
            for (int part = 0; part < 2; ++part) {
                const int64_t first = c.first + (int64_t)part * N * c.h;     // stream position of row 0
                int64_t s0 = first >= a ? 0 : (a - first + c.h - 1) / c.h;
                for (int64_t s = s0; s < N; ++s) {
                    const int64_t pos = first + s * c.h;
                    if (pos >= b) break;
                    dst[part * D.pack_plane + s * E] = chunk(pos - a);
                }
            }
        }
    }
}

// The float32 sample pass does not start from the pair of uniforms but from the point of the unit
// disk they select: (sqrt(u_radius) cos(2 pi u_angle), sqrt(u_radius) sin(2 pi u_angle)) (reference
// objects.py:132-135), evaluated once in float64 when the sample stream is complete and rounded to
// float.  The light point of a sample is then position + radius * (x * x_hat + y * y_hat).
// Output layout: see Dev::packed32.  One thread per (tile, pair of samples, entry of the tile); slots
// beyond the shading list or the sample count hold the disk's centre.
__global__ __launch_bounds__(BLOCK) void k_disk_points(const double* __restrict__ angles /* the light's first angle uniform */,
                                                       int64_t plane /* doubles from an angle to its radius uniform */, int64_t E, int N,
                                                       float4* __restrict__ points /* the light's first float4 */) {
    const int NP = (N + 1) >> 1;
    const int64_t n = ((E + TILE - 1) / TILE) * NP * TILE;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const int lane = (int)(i & (TILE - 1));
        const int64_t tp = i / TILE;
        const int p = (int)(tp % NP);
        const int64_t e = (tp / NP) * TILE + lane;
        float x[2] = {0, 0}, y[2] = {0, 0};
        if (e < E) {
            for (int k = 0; k < 2; ++k) {
                const int s = 2 * p + k;
                if (s >= N) break;
                disk_point(angles[s * E + e], angles[plane + s * E + e], x[k], y[k]);
            }
        }
        points[i] = make_float4(x[0], x[1], y[0], y[1]);
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
// Streams: the word sequence is cut into streams of L = 2^ell words; stream t has its own ring
// and counter and starts from the window x[t L .. t L + 623]: the caller's key block for stream
// 0, windows k_mt_jump made for the others.  The first word of such a window is only good in
// its top bit, so stream t > 0 delivers the words [t L + 1, (t+1) L + 1] and stream 0 the words
// [0, L + 1]: two words of overlap, so that a value built from the last word of one stream and
// the first of the next is whole in one of them.
constexpr int MT_N = 624, MT_M = 397, MT_STEP = MT_N - MT_M, MT_RING = 2048;
constexpr int MAP_GROUP = 4;   // 64-bit words of the store map k_mt_generate reads at a time
__device__ __forceinline__ uint32_t mt_twist(uint32_t lo, uint32_t hi) {
    const uint32_t y = (lo & 0x80000000u) | (hi & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
// grid: one workgroup per stream t = t0 + blockIdx.x.
//   raw[w - raw_origin] receives x[w] for the stream's words below w_end
//   the stream with index last_stream keeps generating until its counter reaches gen_until
//   need (or null: every word is stored): the store map k_store_map made - most of the time goes into
//   the stores (the recurrence itself is a few instructions per word), and words nobody will read
//   are not stored.  Only valid for a stream that has not generated yet (its counter at t L + 624).
__global__ __launch_bounds__(256) void k_mt_generate(uint32_t* rings, long long* gen_ends, uint32_t* raw, int t0, int ell,
                                                     long long raw_origin, long long w_end, int last_stream,
                                                     long long gen_until, const unsigned long long* __restrict__ need, int need_words) {
    // x[n] at slot n & 2047 and again 2048 slots further up: a lane reads its four twist operands at
    // fixed distances from one address that never has to wrap
    __shared__ uint32_t ring[2 * MT_RING];
    typedef const __attribute__((address_space(4))) unsigned long long* ConstMap;   // uniform reads through the scalar unit
    const int tid = threadIdx.x;
    const int t = t0 + blockIdx.x;
    uint32_t* ring_g = rings + (size_t)t * MT_RING;
    for (int i = tid; i < MT_RING; i += 256) ring[i] = ring[i + MT_RING] = ring_g[i];
    const long long g0 = gen_ends[t];
    const long long raw_first = ((long long)t << ell) + (t > 0 ? 1 : 0);
    long long raw_end = ((long long)(t + 1) << ell) + 2;
    if (raw_end > w_end) raw_end = w_end;
    if (!raw || raw_end < raw_first) raw_end = raw_first;
    const long long raw_n = raw_end - raw_first;
    long long target = raw_n > 0 ? raw_end : g0;
    if (t == last_stream && gen_until > target) target = gen_until;
    uint32_t* out = raw ? raw + (raw_first - raw_origin) : nullptr;
    __syncthreads();
    // words of the window that exist already
    for (long long n = raw_first + tid; n < raw_end && n < g0; n += 256) out[n - raw_first] = ring[n & (MT_RING - 1)];
    // two steps per iteration (all four twist operands were written before the previous barrier)
    const long long todo = target > g0 ? target - g0 : 0;
    const int iters = (int)((todo + 2 * MT_STEP - 1) / (2 * MT_STEP));
    const bool lane_on = tid < MT_STEP;
    int base = (int)(g0 & (MT_RING - 1));                 // slot of the iteration's first word (uniform)
    uint32_t prev = lane_on ? ring[(base + tid - MT_STEP) & (MT_RING - 1)] : 0u;
    const long long rel_first = g0 - raw_first;           // place of iteration 0's first word in out[] (may be negative)
    // iterations all of whose words have a place in out[]: [full_lo, full_hi)
    const int full_lo = rel_first >= 0 ? 0 : (int)min((long long)iters, (-rel_first + 2 * MT_STEP - 1) / (2 * MT_STEP));
    const int full_hi = raw_n > rel_first ? (int)min((long long)iters, (raw_n - rel_first) / (2 * MT_STEP)) : 0;
    // store map: one byte per step (two per iteration), bit w of it = wave w stores; MAP_GROUP 64-bit
    // words - four iterations each - are read at a time
    ConstMap need_t = need ? (ConstMap)(uintptr_t)(need + (size_t)t * need_words) : nullptr;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t* lane_ring = ring + tid;
    // (the loop is bound by the scalar unit, one per compute unit and shared by every wave on it: the common
    // iteration - all of its words have a place in out[] - is kept free of 64-bit counters and range tests)
    char* const out_bytes = reinterpret_cast<char*>(out);
    uint32_t lane_out = (uint32_t)((rel_first + tid) * 4);   // byte offset of the lane's first word of the iteration in out[] (used while it is inside)
    // CHECKED: some of the iteration's words may have no place in out[] (a stream's first and last iterations)
    // whole: every lane of the wave computes (the first three waves; the fourth has 35 lanes that do)
    auto iteration = [&](auto checked, auto whole, int it, bool store0, bool store1) {
        if (decltype(whole)::value || lane_on) {
            const uint32_t* operands = lane_ring + ((base - MT_N) & (MT_RING - 1));
            const uint32_t lo0 = operands[0], hi0 = operands[1], lo1 = operands[MT_STEP], hi1 = operands[MT_STEP + 1];
            const uint32_t x0 = prev ^ mt_twist(lo0, hi0);
            const uint32_t x1 = x0 ^ mt_twist(lo1, hi1);
            const int s0 = (base + tid) & (MT_RING - 1), s1 = (base + tid + MT_STEP) & (MT_RING - 1);
            ring[s0] = x0; ring[s0 + MT_RING] = x0;
            ring[s1] = x1; ring[s1 + MT_RING] = x1;
            if constexpr (!decltype(checked)::value) {
                if (store0) *reinterpret_cast<uint32_t*>(out_bytes + lane_out) = x0;
                if (store1) *reinterpret_cast<uint32_t*>(out_bytes + lane_out + 4 * MT_STEP) = x1;
            } else {
                const long long rel = rel_first + (long long)it * (2 * MT_STEP) + tid;
                if (store0 && rel >= 0 && rel < raw_n) out[rel] = x0;
                if (store1 && rel + MT_STEP >= 0 && rel + MT_STEP < raw_n) out[rel + MT_STEP] = x1;
            }
            prev = x1;
        }
        lane_out += 4 * 2 * MT_STEP;
        base = (base + 2 * MT_STEP) & (MT_RING - 1);
        __syncthreads();
    };
    int it = 0;
    auto checked_until = [&](int end) {
        for (; it < end; ++it) {
            const unsigned long long word = need_t ? need_t[min(it >> 2, need_words - 1)] : ~0ull;
            const unsigned both = (unsigned)(word >> ((it & 3) * 16 + wave));
            iteration(std::true_type{}, std::false_type{}, it, both & 1u, both & 0x100u);
        }
    };
    // whole map words (four iterations each) of iterations that are inside: [main_lo, main_hi)
    const int main_lo = min(iters, (full_lo + 3) & ~3), main_hi = max(main_lo, full_hi & ~3);
    checked_until(main_lo);
    auto main_loop = [&](auto whole) {
        for (; it < main_hi; it += 4 * MAP_GROUP) {
            unsigned long long map[MAP_GROUP];
#pragma unroll
            for (int q = 0; q < MAP_GROUP; ++q) map[q] = need_t ? need_t[min(it / 4 + q, need_words - 1)] : ~0ull;
#pragma unroll
            for (int q = 0; q < MAP_GROUP; ++q) {
                if (it + 4 * q >= main_hi) break;
                const unsigned long long m = map[q] >> wave;
#pragma unroll
                for (int r = 0; r < 4; ++r) iteration(std::false_type{}, whole, 0, (m >> (16 * r)) & 1u, (m >> (16 * r + 8)) & 1u);
            }
        }
    };
    if (wave < MT_STEP / 64) main_loop(std::true_type{}); else main_loop(std::false_type{});   // (both take every barrier)
    it = main_hi;
    checked_until(iters);
    for (int i = tid; i < MT_RING; i += 256) ring_g[i] = ring[i];
    if (tid == 0) gen_ends[t] = g0 + (long long)iters * 2 * MT_STEP;
}

// ---------------------------------------------------------------------------------------------
// Generated values straight into the packed layout (disk lights): k_mt_emit is k_mt_generate that
// keeps its words in LDS and, every few steps, turns the new ones into doubles and stores each at the
// place of the (hit, sample) it belongs to - the raw words never reach device memory, and values of
// hits that are not shaded (two thirds of the stream in the example scenes) are dropped on the spot.
//
// Where a value belongs follows from its position p in the stream: the (node, light) block that
// holds p (EmitBlock, sorted by offset), there part (angles, then radii), sample s and column j -
// the j-th hit of the node in pixel order.  Whether that hit is on the shading list, and at which
// entry, is answered by one bit per hit (k_hitmask: bit j of the node's range is set for a shaded
// hit) and a running count per 64 bits (k_hitprefix): entries of a node are consecutive, in hit order.
struct EmitBlock {
    int64_t off, end;          // positions [off, end) of the stream: 2 * n_points rows of h values
    double inv_h;              // 1 / h
    uint32_t h;                // values per row (the node's hits in the whole image)
    uint32_t rows;             // n_points (the radius rows follow the angle rows)
    uint32_t first_col, mine;  // this context's hits are the columns [first_col, first_col + mine)
    uint32_t word0;            // first 64-bit word of the node's range in the hit mask
    uint32_t e_first;          // shading-list entry of the node's first shaded hit
    uint32_t light;
    uint32_t node;
};
static_assert(sizeof(EmitBlock) == 56, "EmitBlock is read as seven 8-byte words");

// One bit per hit, in node order then pixel order: set if the hit is on the shading list.  One launch
// per level, the grid and the tile runs of k_link.  mask must be zero on entry.  counter: C_HIT - all
// hits, the columns of a disk light's sample blocks - or C_BEAM0 + l: the hits inside spotlight l's beam,
// the columns of its blocks.
__global__ __launch_bounds__(BLOCK) void k_hitmask(Dev D, int level, const uint32_t* __restrict__ node_word0, unsigned long long* mask, int counter) {
    extern __shared__ int s_dyn[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (D.flags[0]) return;
    const int k0 = D.level_start[level], nn = D.level_start[level + 1] - k0;
    const int nt = D.lvl_ntiles[level];
    const int tbase = D.lvl_tilebase[level];
    const int q = tiles_per_wave(D, nt);
    const int wid = blockIdx.x * WPB + wave;
    const int t_begin = wid * q;
    const int t_end = min(nt, t_begin + q);
    if (blockIdx.x * WPB * q >= nt) return;
    const LevelTables T = stage_level_tables(D, level, s_dyn, tid, blockIdx.x * WPB * q, min(nt, (int)(blockIdx.x + 1) * WPB * q) - 1);
    const unsigned wave_base = D.blk_tot[((size_t)level * NCNT + counter) * NB + blockIdx.x] + D.wave_off[((size_t)level * NCNT + counter) * NW + wid];
    __syncthreads();
    int kk = (t_begin < t_end) ? find_node(T.node, nn, t_begin) : 0;
    for (int t = t_begin; t < t_end; ++t) {
        while (T.node[kk].tile_end <= t) ++kk;
        kk = __builtin_amdgcn_readfirstlane(kk);
        const NodeRec r = T.node[kk];
        const int first = (t - __builtin_amdgcn_readfirstlane(r.tile0)) * TILE;
        const int cnt = min(TILE, __builtin_amdgcn_readfirstlane(r.n) - first);
        const uint32_t info = lane < cnt ? D.info[(uint32_t)(tbase + t) * TILE + lane] : I_NO_HIT;
        const bool mine = counted(info, counter);
        const unsigned long long hits = __ballot(mine);
        if (hits == 0) continue;
        const unsigned trun = D.tile_run[(size_t)(tbase + t) * NCNT + counter];
        // shade bits in hit order: hit number i of the tile sends its bit to lane i
        const int nh = __popcll(hits);
        const int dest = mine ? lanes_below(hits) : 63;
        const int got = __builtin_amdgcn_ds_permute(dest * 4, (info & I_SHADE) ? 1 : 0);
        unsigned long long bits = __ballot(got != 0);
        if (nh < 64) bits &= (1ull << nh) - 1;
        // rank of the tile's first hit inside its node -> place in the mask
        const unsigned long long place = (unsigned long long)node_word0[k0 + kk] * 64 + (wave_base + trun - T.pre[kk * 16 + counter]);
        const unsigned sh = (unsigned)(place & 63);
        if (lane == 0 && bits != 0) {
            atomicOr(&mask[place >> 6], bits << sh);
            if (sh != 0 && (bits >> (64 - sh)) != 0) atomicOr(&mask[(place >> 6) + 1], bits >> (64 - sh));
        }
    }
}

// Set bits before each word of a node's range of the hit mask.  One workgroup per node.
__global__ __launch_bounds__(BLOCK) void k_hitprefix(const unsigned long long* __restrict__ mask, const uint32_t* __restrict__ node_word0, uint32_t* __restrict__ pref) {
    __shared__ unsigned s_wave[WPB];
    __shared__ unsigned s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t w0 = node_word0[blockIdx.x], w1 = node_word0[blockIdx.x + 1];
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = w0; base < w1; base += BLOCK) {
        const uint32_t w = base + tid;
        const unsigned c = w < w1 ? (unsigned)__popcll(mask[w]) : 0u;
        unsigned incl = c;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        unsigned before = s_carry;
        for (int v = 0; v < wave; ++v) before += s_wave[v];
        if (w < w1) pref[w] = before + incl - c;
        __syncthreads();
        if (tid == BLOCK - 1) s_carry = before + incl;
        __syncthreads();
    }
}

// Store map of k_mt_generate: which of the words it makes does k_pack read?  Only those of values
// that belong to a shaded hit of this context - a third of the stream in the example scenes, and in
// runs, since hits are numbered in pixel order and surfaces without a diffuse part cover areas.  The
// generator stores a step's 227 words wave by wave (64 + 64 + 64 + 35); the map has one byte per step,
// bit w set if wave w's words hold part of a value somebody reads (conservative: whenever that is
// not cheap to tell - a run of values across two (node, light) blocks, a block narrower than the run -
// the bit is set).  A spotlight's columns are the hits inside its beam, a disk light's all hits of the
// node: the mask has a row of bits for each numbering (k_hitmask's counter).  One thread per step; steps are counted from word t L + 624 of stream t, where an unused
// stream starts generating.
//   blocks      every (node, light) block with values, sorted by offset (light == EMIT_ALWAYS: a block without a mask)
//   need        [streams][bytes_per_stream] bytes (bytes_per_stream a multiple of 8)
//   words_kept  += the words the map lets the generator store
constexpr uint32_t EMIT_ALWAYS = 0xFFFFFFFFu;
__device__ __forceinline__ bool any_mask_bits(const unsigned long long* __restrict__ mask, unsigned long long lo, unsigned long long hi) {
    // bits [lo, hi), hi > lo
    const unsigned long long w0 = lo >> 6, w1 = (hi - 1) >> 6;
    for (unsigned long long w = w0; w <= w1; ++w) {
        unsigned long long m = mask[w];
        if (w == w0) m &= ~0ull << (lo & 63);
        if (w == w1 && (hi & 63)) m &= (1ull << (hi & 63)) - 1;
        if (m) return true;
    }
    return false;
}
// the byte of step i % bytes_per_stream of stream i / bytes_per_stream; block k's offset is offsets[k * STRIDE]
// (a copy in LDS, or the blocks themselves)
constexpr int MAP_LDS_BLOCKS = 2048;
static_assert(sizeof(EmitBlock) % sizeof(int64_t) == 0 && offsetof(EmitBlock, off) == 0, "the blocks as an array of offsets with a stride");
template <int STRIDE>
__device__ unsigned store_map_bits(const EmitBlock* __restrict__ blocks, const int64_t* offsets, int n_blocks, const unsigned long long* __restrict__ mask,
                                   long long pos, long long n_values, int ell, int steps_per_stream, int bytes_per_stream, long long i) {
    const long long t = i / bytes_per_stream;
    const int step = (int)(i - t * bytes_per_stream);
    if (step >= steps_per_stream) return 0;
    const long long w_first = (t << ell) + MT_N + (long long)step * MT_STEP;
    // the values with a word in [w_first, w_first + 227)
    long long v_first = w_first >= pos ? (w_first - pos) >> 1 : 0;
    long long v_last = w_first + MT_STEP - 1 >= pos ? (w_first + MT_STEP - 1 - pos) >> 1 : -1;
    if (v_last > n_values - 1) v_last = n_values - 1;
    if (v_first > v_last) return 0;
    // block of the first value: the last one that begins at or before it
    int lo = 0, hi = n_blocks - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (offsets[(size_t)mid * STRIDE] <= v_first) lo = mid; else hi = mid - 1;
    }
    const EmitBlock B = blocks[lo];
    if (B.light == EMIT_ALWAYS || v_first < B.off || v_last >= B.end) return 0xF;
    // column of the step's first value (one division for the step; the waves' ranges follow by addition)
    const long long h = B.h;
    const long long rel = v_first - B.off;
    long long col_first = rel - (long long)((double)rel * B.inv_h) * h;
    if (col_first < 0) col_first += h; else if (col_first >= h) col_first -= h;
    // does anyone read one of the n values from v on?  (columns [col, col + n), beyond h on the next row from column 0)
    auto read_by_someone = [&](long long v, long long n) -> bool {
        if (n >= h) return true;
        long long col = col_first + (v - v_first);
        while (col >= h) col -= h;
        for (int part = 0; part < 2; ++part) {
            long long c0 = part == 0 ? col : 0, c1 = part == 0 ? min(col + n, h) : col + n - h;
            c0 = max(c0, (long long)B.first_col) - B.first_col;
            c1 = min(c1, (long long)B.first_col + B.mine) - B.first_col;
            if (c0 < c1 && any_mask_bits(mask, (unsigned long long)B.word0 * 64 + c0, (unsigned long long)B.word0 * 64 + c1)) return true;
        }
        return false;
    };
    if (!read_by_someone(v_first, v_last - v_first + 1)) return 0;   // two steps in three
    unsigned bits = 0;
    for (int w = 0; w < 4; ++w) {
        const long long wa = w_first + 64 * w, wb = wa + (w == 3 ? MT_STEP - 192 : 64);   // [wa, wb)
        if (wb <= pos) continue;
        const long long va = wa >= pos ? (wa - pos) >> 1 : 0;
        const long long vb = min((wb - 1 - pos) >> 1, n_values - 1);
        if (va > vb) continue;
        if (read_by_someone(va, vb - va + 1)) bits |= 1u << w;
    }
    return bits;
}
__global__ __launch_bounds__(256) void k_store_map(const EmitBlock* __restrict__ blocks, int n_blocks, const unsigned long long* __restrict__ mask,
                                                   long long pos, long long n_values, int ell, int steps_per_stream, int bytes_per_stream,
                                                   long long n_bytes, uint8_t* __restrict__ need, unsigned long long* __restrict__ words_kept) {
    // the blocks' offsets for the search: in LDS if they fit (every workgroup goes through many steps)
    __shared__ int64_t s_off[MAP_LDS_BLOCKS];
    const bool staged = n_blocks <= MAP_LDS_BLOCKS;
    if (staged) {
        for (int k = threadIdx.x; k < n_blocks; k += 256) s_off[k] = blocks[k].off;
        __syncthreads();
    }
    unsigned long long words = 0;   // statistics: words the generator will store (64 per wave, 35 for the fourth)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_bytes; i += (long long)gridDim.x * 256) {
        unsigned bits;
        if (staged) bits = store_map_bits<1>(blocks, s_off, n_blocks, mask, pos, n_values, ell, steps_per_stream, bytes_per_stream, i);
        else bits = store_map_bits<sizeof(EmitBlock) / sizeof(int64_t)>(blocks, reinterpret_cast<const int64_t*>(blocks), n_blocks, mask, pos, n_values, ell,
                                                                       steps_per_stream, bytes_per_stream, i);
        need[i] = (uint8_t)bits;
        words += 64u * (unsigned)__popc(bits & 7u) + (unsigned)(MT_STEP - 192) * (bits >> 3);
    }
    for (int off = 32; off > 0; off >>= 1) words += __shfl_down(words, off);
    if ((threadIdx.x & 63) == 0 && words) atomicAdd(words_kept, words);
}

// grid: one workgroup per stream t = blockIdx.x (see k_mt_generate for rings, gen_ends, ell, last_stream,
// gen_until).  Value v of the stream (0 <= v < n_values) is built from the words pos + 2 v and pos + 2 v + 1
// (words below w_end exist) and goes to the packed layout if its hit is shaded.
constexpr int EMIT_RING = 8192;          // words kept in LDS
#ifndef RT_EMIT_STEPS
#define RT_EMIT_STEPS 14
#endif
constexpr int EMIT_STEPS = RT_EMIT_STEPS;   // double steps (454 words each) between two rounds of stores
#ifndef RT_EMIT_GROUP
#define RT_EMIT_GROUP 4
#endif
constexpr int EMIT_GROUP = RT_EMIT_GROUP;   // batches of 64 values whose loads a wave has in flight together
static_assert(EMIT_STEPS * 2 * MT_STEP + MT_N + 8 + 2 * 64 < EMIT_RING, "a round's words and the recurrence's reach fit the ring");
__global__ __launch_bounds__(256) void k_mt_emit(Dev D, uint32_t* rings, long long* gen_ends, int ell, long long pos, long long n_values, long long w_end,
                                                 int last_stream, long long gen_until, const EmitBlock* __restrict__ blocks, int n_blocks,
                                                 const unsigned long long* __restrict__ mask, const uint32_t* __restrict__ pref) {
    __shared__ uint32_t ring[EMIT_RING];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = blockIdx.x;
    uint32_t* ring_g = rings + (size_t)t * MT_RING;
    const long long g0 = gen_ends[t];
    for (int i = tid; i < MT_RING; i += 256) {
        // ring_g[n & 2047] holds x[n] for the 2048 indices below g0
        const long long n = g0 - MT_RING + i;
        ring[n & (EMIT_RING - 1)] = ring_g[n & (MT_RING - 1)];
    }
    const long long raw_first = ((long long)t << ell) + (t > 0 ? 1 : 0);
    long long raw_end = ((long long)(t + 1) << ell) + 2;
    if (raw_end > w_end) raw_end = w_end;
    if (raw_end < raw_first) raw_end = raw_first;
    long long target = raw_end > g0 ? raw_end : g0;
    if (t == last_stream && gen_until > target) target = gen_until;
    // values whose two words both lie in [raw_first, raw_end)
    long long v_next = raw_first <= pos ? 0 : (raw_first - pos + 1) >> 1;
    long long v_stop = raw_end - 2 - pos >= 0 ? ((raw_end - 2 - pos) >> 1) + 1 : 0;
    if (v_stop > n_values) v_stop = n_values;
    __syncthreads();

    const bool lane_on = tid < MT_STEP;
    long long g = g0;   // words below g exist
    uint32_t prev = lane_on ? ring[(g + tid - MT_STEP) & (EMIT_RING - 1)] : 0u;
    if (n_blocks < 0) n_blocks = 0;   // (measurement: generate only)
    EmitBlock cur = blocks[0];   // the block the wave's last batch began in
    struct {
        long long v0, v1;      // stream positions [v0, v1) of the row the wave's last batch began in (one sample of one part of a block)
        int64_t at;            // index in D.packed of the row's entry 0
        uint32_t first_col, mine, word0, e_first;
    } row = {0, 0, 0, 0, 0, 0, 0};
    struct { long long v0, v1; } gap = {0, 0};   // a range of positions without a block
    while (true) {
        // ---- generate: up to EMIT_STEPS double steps (all four twist operands were written before the previous barrier)
        for (int it = 0; it < EMIT_STEPS && g < target; ++it) {
            if (lane_on) {
                const int s0 = (int)((g + tid) & (EMIT_RING - 1)), s1 = (s0 + MT_STEP) & (EMIT_RING - 1);
                const uint32_t lo0 = ring[(s0 - MT_N) & (EMIT_RING - 1)], hi0 = ring[(s0 - MT_N + 1) & (EMIT_RING - 1)];
                const uint32_t lo1 = ring[(s1 - MT_N) & (EMIT_RING - 1)], hi1 = ring[(s1 - MT_N + 1) & (EMIT_RING - 1)];
                const uint32_t x0 = prev ^ mt_twist(lo0, hi0);
                const uint32_t x1 = x0 ^ mt_twist(lo1, hi1);
                ring[s0] = x0;
                ring[s1] = x1;
                prev = x1;
            }
            g += 2 * MT_STEP;
            __syncthreads();
        }
        // ---- store: the values whose words exist now, 64 consecutive ones per wave and batch (whole
        // batches only until the stream's end, so that a batch is always the same 64 positions).  Nearly
        // always the 64 positions lie in one row of one block: block, part, sample and first column are
        // then found once for the batch (the same arithmetic in every lane), and a lane adds its number
        // to the column.  The loads of EMIT_GROUP batches (mask word, count) are requested before the
        // first of them is used.
        const bool final_round = g >= target;
        long long v_avail = g - 2 - pos >= 0 ? ((g - 2 - pos) >> 1) + 1 : 0;
        if (v_avail > v_stop) v_avail = v_stop;
        const long long b_stop = final_round ? (v_stop + 63) >> 6 : v_avail >> 6;
        const long long v_to = final_round ? v_stop : (b_stop << 6);
        for (long long b0 = (v_next >> 6) + wave; b0 < b_stop && n_blocks > 0; b0 += (long long)EMIT_GROUP * 4) {
            struct Slot {
                double u;
                int64_t at;        // index in D.packed of entry 0 of the value's row
                uint32_t word;     // word of the hit mask with the hit's bit
                uint32_t bit, e_first;
                bool ok;
            } slot[EMIT_GROUP];
            unsigned long long m[EMIT_GROUP];
            uint32_t before[EMIT_GROUP];
#pragma unroll
            for (int k = 0; k < EMIT_GROUP; ++k) {
                Slot& S = slot[k];
                S.ok = false; S.u = 0; S.at = 0; S.word = 0; S.bit = 0; S.e_first = 0;
                const long long vb = (b0 + 4 * k) << 6;
                if (b0 + 4 * k >= b_stop) continue;   // uniform
                // lanes of a partial batch (the stream's first and last one) that have no value
                const int lane_lo = vb < v_next ? (int)(v_next - vb) : 0, lane_hi = vb + 64 > v_to ? (int)(v_to - vb) : 64;
                uint32_t col;
                if (vb >= gap.v0 && vb + 63 < gap.v1) continue;   // between two blocks: values of nodes without a shaded hit
                if (vb >= row.v0 && vb + 63 < row.v1) {
                    // the usual case: all 64 positions in the row of the batch before
                    col = (uint32_t)(vb - row.v0) + lane;
                } else {
                    // the batch's first position: block, part, sample, column (uniform)
                    if (!(vb >= cur.off && vb < cur.end)) {   // (kept in registers: a stream lies in a handful of blocks)
                        int a = 0, z = n_blocks - 1;
                        while (a < z) {
                            const int mid = (a + z) >> 1;
                            if (blocks[mid].end > vb) z = mid; else a = mid + 1;
                        }
                        cur = blocks[a];
                        if (vb < cur.off) {
                            // positions below this block that no block holds (the caller lists only nodes with shaded hits)
                            gap.v0 = a > 0 ? blocks[a - 1].end : 0;
                            gap.v1 = cur.off;
                            if (vb >= gap.v0 && vb + 63 < gap.v1) continue;
                        }
                    }
                    EmitBlock B = cur;
                    long long v = vb + lane;
                    unsigned long long r = (unsigned long long)(vb - B.off);
                    const unsigned long long half = (unsigned long long)B.rows * B.h;
                    int part = (vb >= B.off && r >= half) ? 1 : 0;
                    if (part) r -= half;
                    unsigned long long smp = (unsigned long long)((double)r * B.inv_h);
                    if (smp * B.h > r) --smp;
                    unsigned long long c0 = r - smp * B.h;
                    if (c0 >= B.h) { ++smp; c0 -= B.h; }
                    if (vb >= B.off && vb < B.end) {
                        // the row the batch begins in is the wave's row from now on
                        row.v0 = vb - (long long)c0;
                        row.v1 = row.v0 + B.h;
                        row.at = D.pack_base[B.light] + (int64_t)part * D.pack_plane + (int64_t)smp * D.pack_E;
                        row.first_col = B.first_col; row.mine = B.mine; row.word0 = B.word0; row.e_first = B.e_first;
                    }
                    if (vb >= row.v0 && vb + 63 < row.v1) {
                        col = (uint32_t)c0 + lane;
                    } else {
                        // a batch across the end of a row (or of a block): every lane finds its own place
                        int a = 0, z = n_blocks - 1;
                        while (a < z) {
                            const int mid = (a + z) >> 1;
                            if (blocks[mid].end > v) z = mid; else a = mid + 1;
                        }
                        B = blocks[a];
                        if (!(v >= B.off && v < B.end) || lane < lane_lo || lane >= lane_hi) continue;
                        r = (unsigned long long)(v - B.off);
                        const unsigned long long half_ = (unsigned long long)B.rows * B.h;
                        part = r >= half_ ? 1 : 0;
                        if (part) r -= half_;
                        smp = (unsigned long long)((double)r * B.inv_h);
                        if (smp * B.h > r) --smp;
                        c0 = r - smp * B.h;
                        if (c0 >= B.h) { ++smp; c0 -= B.h; }
                        if (c0 < B.first_col || c0 - B.first_col >= B.mine) continue;   // another context's column
                        const uint32_t jl = (uint32_t)(c0 - B.first_col);
                        const uint32_t w0 = (uint32_t)(pos + 2 * v);
                        const uint32_t hi = mt_temper(ring[w0 & (EMIT_RING - 1)]) >> 5, lo = mt_temper(ring[(w0 + 1) & (EMIT_RING - 1)]) >> 6;
                        S.u = ((double)hi * 67108864.0 + (double)lo) * 1.1102230246251565e-16;
                        S.at = D.pack_base[B.light] + (int64_t)part * D.pack_plane + (int64_t)smp * D.pack_E;
                        S.word = B.word0 + (jl >> 6);
                        S.bit = jl & 63;
                        S.e_first = B.e_first;
                        S.ok = true;
                        continue;
                    }
                }
                const uint32_t jl = col - row.first_col;
                if (!(jl < row.mine) || col < row.first_col || lane < lane_lo || lane >= lane_hi) continue;   // another context's column
                const uint32_t w0 = (uint32_t)(pos + 2 * vb) + 2 * lane;
                const uint32_t hi = mt_temper(ring[w0 & (EMIT_RING - 1)]) >> 5, lo = mt_temper(ring[(w0 + 1) & (EMIT_RING - 1)]) >> 6;
                S.u = ((double)hi * 67108864.0 + (double)lo) * 1.1102230246251565e-16;   // 2^-53, exact
                S.at = row.at;
                S.word = row.word0 + (jl >> 6);
                S.bit = jl & 63;
                S.e_first = row.e_first;
                S.ok = true;
            }
#pragma unroll
            for (int k = 0; k < EMIT_GROUP; ++k) {
                m[k] = slot[k].ok ? mask[slot[k].word] : 0ull;
                before[k] = slot[k].ok ? pref[slot[k].word] : 0u;
            }
#pragma unroll
            for (int k = 0; k < EMIT_GROUP; ++k) {
                const Slot& S = slot[k];
                if (S.ok && ((m[k] >> S.bit) & 1ull))   // a hit that is not shaded: nobody reads its samples
                    D.packed[S.at + (int64_t)(S.e_first + before[k] + (uint32_t)__popcll(m[k] & ((1ull << S.bit) - 1)))] = S.u;
            }
        }
        if (v_to > v_next) v_next = v_to;
        __syncthreads();   // the next round overwrites ring slots this round read
        if (g >= target) break;
    }
    // the last 2048 words, where k_mt_generate keeps them
    for (int i = tid; i < MT_RING; i += 256) {
        const long long n = g - MT_RING + i;
        ring_g[n & (MT_RING - 1)] = ring[n & (EMIT_RING - 1)];
    }
    if (tid == 0) gen_ends[t] = g;
}

// Jump-ahead.  The start window of stream t is the window at word index t L.  A window is moved
// J words ahead by the polynomial x^J mod phi:  new window word j = XOR over set coefficients i of
// s[i + j], s the sequence generated from the current window.  The low 31 bits of the new window's
// first word are not meaningful (nothing reads them: the previous stream delivers that word, and the
// recurrence itself only uses the top bit of a window's first word, so such a window can be moved
// on again).  Polynomial b is x^(L 2^b) mod phi, L = 2^ell words per stream; heads[b] and `lists`
// hold it as the list of its set coefficients (JumpPoly below).
//
// The windows are made level by level, one launch per level b = 0, 1, ...: the window of stream
// t = 2^b + i comes from the window of stream t - 2^b (stream 0: the caller's key block in ring 0,
// which must not have generated yet) with the one polynomial b - one application per stream instead
// of one per set bit of t.  An application is shared by MTJ_ROUNDS workgroups (blockIdx.x =
// i MTJ_ROUNDS + r): each takes the coefficients [r MTJ_CHUNK, (r + 1) MTJ_CHUNK) - it generates s
// as far as they reach, which is cheap next to the sums - and adds its part into the stream's ring
// with atomic XORs, so the rings of the streams 1.. must be zero before the first level.
constexpr int MTJ_RING = 4096, MTJ_CHUNK = 8 * MT_STEP;   // coefficients consumed per round
constexpr int MTJ_ROUNDS = (19937 + MTJ_CHUNK - 1) / MTJ_CHUNK;
// A polynomial as the list of its set coefficients (what k_mt_jump walks; pulling them out of the
// coefficient words bit by bit cost more scalar instructions than the sums cost vector ones), made
// so that a lane's read needs one addition: coefficient i of round r (i in [r MTJ_CHUNK,
// (r + 1) MTJ_CHUNK)) is stored as the byte offset of s[i] in the workgroup's array, which keeps s[n]
// at word (n - r MTJ_CHUNK + 624) & 4095 - so the words the round reads, s[i + j] for j < 624, lie
// at offset + 4 j without wrapping.  Ascending, one per 32-bit word (two per word measured the same kernel time: the
// kernel waits on its LDS reads, not on the scalar unit that unpacks), padded
// per round to a multiple of 16 with MTJ_PAD, the offset of 640 words that are kept zero.
// first[r] = first word of round r's list.
constexpr uint32_t MTJ_PAD = MTJ_RING * 4;
static_assert(MTJ_CHUNK + 2 * MT_N <= MTJ_RING, "the words a round reads lie in the array without wrapping");
struct JumpPoly {
    int32_t first[MTJ_ROUNDS + 1];
};
__host__ __device__ constexpr uint32_t mtj_offset(int i, int r) { return (uint32_t)(i - r * MTJ_CHUNK + MT_N) * 4u; }
__global__ __launch_bounds__(640) void k_mt_jump(uint32_t* rings, long long* gen_ends, const JumpPoly* heads, const uint32_t* lists, int level, int ell) {
    __shared__ uint32_t xs[MTJ_RING + 640];     // s[n] at slot (n - c0 + 624) & 4095; the last 640 words stay zero
    typedef const __attribute__((address_space(4))) uint32_t* ConstList;   // uniform reads through the scalar unit
    const int tid = threadIdx.x;
    const int t = (1 << level) + blockIdx.x / MTJ_ROUNDS;
    const int r = blockIdx.x % MTJ_ROUNDS;          // this workgroup's share of the polynomial
    const long long L = 1ll << ell;
    const int c0 = r * MTJ_CHUNK;
    const int shift = MT_N - c0;                    // slot of s[n]: (n + shift) & 4095
    xs[MTJ_RING + tid] = 0;
    // the source window: x[u L .. u L + 623] of stream u = t - 2^level, in that stream's ring
    {
        const int u = t - (1 << level);
        const long long from = (long long)u * L;
        for (int j = tid; j < MT_N; j += 640) xs[(j + shift) & (MTJ_RING - 1)] = rings[(size_t)u * MT_RING + ((from + j) & (MT_RING - 1))];
    }
    __syncthreads();
    // s up to the last word the share reads: s[c0 + MTJ_CHUNK + 623] (the array keeps the last 4096)
    {
        uint32_t prev = (tid < MT_STEP) ? xs[(MT_N - MT_STEP + tid + shift) & (MTJ_RING - 1)] : 0u;   // own-chain value s[gen - 227 + tid]
        for (int gen = MT_N; gen < c0 + MTJ_CHUNK + MT_N; gen += 2 * MT_STEP) {
            if (tid < MT_STEP) {
                const int n0 = gen + tid + shift, n1 = n0 + MT_STEP;
                const uint32_t x0 = prev ^ mt_twist(xs[(n0 - MT_N) & (MTJ_RING - 1)], xs[(n0 - MT_N + 1) & (MTJ_RING - 1)]);
                const uint32_t x1 = x0 ^ mt_twist(xs[(n1 - MT_N) & (MTJ_RING - 1)], xs[(n1 - MT_N + 1) & (MTJ_RING - 1)]);
                xs[n0 & (MTJ_RING - 1)] = x0;
                xs[n1 & (MTJ_RING - 1)] = x1;
                prev = x1;
            }
            __syncthreads();
        }
    }
    // the share's coefficients, sixteen at a time: their reads are in flight together
    if (tid < MT_N) {
        const JumpPoly& head = heads[level];
        const char* lane_words = reinterpret_cast<const char*>(xs + tid);
        uint32_t acc = 0;
        for (int w = head.first[r]; w < head.first[r + 1]; w += 16) {
            ConstList src = (ConstList)(uintptr_t)(lists + w);
            uint32_t offset[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) offset[k] = src[k];
            uint32_t v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = *reinterpret_cast<const uint32_t*>(lane_words + offset[k]);
#pragma unroll
            for (int k = 0; k < 16; ++k) acc ^= v[k];
        }
        // into the stream's ring (zero before the launch): window word j is x[t L + j]
        const long long first = (long long)t * L;
        if (acc) atomicXor(&rings[(size_t)t * MT_RING + ((first + tid) & (MT_RING - 1))], acc);
        if (r == 0 && tid == 0) gen_ends[t] = first + MT_N;
    }
}

}  // namespace rt
