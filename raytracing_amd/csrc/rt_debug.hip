// Test hook: k_debug evaluates, for caller-supplied inputs, the device functions the frame kernels
// are made of (the very functions, not copies), so that they can be held to vectors recorded
// from the reference's own functions.
#pragma once
#include "rt_lighting.hip"

namespace rt {

// rows of `in` -> rows of `out`, see rt_debug_eval in the header for the layouts
__global__ __launch_bounds__(BLOCK) void k_debug(Dev D, int what, const double* __restrict__ in, int in_w,
                                                 double* __restrict__ out, int out_w, long long n) {
    __shared__ ShadowSphere s_sh[SHADE_LDS_SPHERES];
    __shared__ double s_sph[SPH_CHUNK_SMALL][4];
    const int tid = threadIdx.x;
    for (int i = tid; i < D.S && i < SHADE_LDS_SPHERES; i += BLOCK) {
        const double* g = D.geom + (size_t)i * 4;
        const double* m = D.mat + (size_t)i * RT_MAT_WIDTH;
        s_sh[i] = {g[0], g[1], g[2], g[3] * g[3], m[11], m[14] * m[15], m[14] * m[16], m[14] * m[17]};
        if (i < SPH_CHUNK_SMALL) { s_sph[i][0] = g[0]; s_sph[i][1] = g[1]; s_sph[i][2] = g[2]; s_sph[i][3] = g[3] * g[3]; }
    }
    __syncthreads();
    const SceneConst& sc = *D.sc;
    const ShadowLists no_lists = {{nullptr}, {nullptr}};
    for (long long i = (long long)blockIdx.x * BLOCK + tid; i < n; i += (long long)gridDim.x * BLOCK) {
        const double* x = in + i * in_w;
        double* y = out + i * out_w;
        switch (what) {
            case RT_DEBUG_SPHERE_ROOTS: {   // reference objects.py:60-66, :238-271
                const V3 o = ld3(x), d = ld3(x + 3);
                const double* g = D.geom + (size_t)(int)x[6] * 4;
                double lo, hi;
                y[0] = y[1] = -1.0;
                if (sphere_roots(o, d, dot(d, o), g[0], g[1], g[2], g[3] * g[3], lo, hi)) {
                    y[0] = lo > 0 ? lo : (hi > 0 ? hi : -1.0);
                    y[1] = hi > 0 ? hi : (lo > 0 ? lo : -1.0);
                }
                break;
            }
            case RT_DEBUG_NEAREST: {        // reference objects.py:288-299
                const V3 o = ld3(x), d = ld3(x + 3);
                double best = __builtin_inf();
                int sid = -1;
                nearest_of_table(s_sph, D.S, o, d, dot(d, o), best, sid);
                y[0] = (double)sid;
                y[1] = best;
                break;
            }
            case RT_DEBUG_SHADOW: {         // reference objects.py:274-285
                const V3 m = shadow_multiplier<true>(D, s_sh, no_lists, ld3(x), ld3(x + 3), -1);
                y[0] = m.x; y[1] = m.y; y[2] = m.z;
                break;
            }
            case RT_DEBUG_OPTICS: {         // reference raytrace.py:173-212
                const V3 N = ld3(x), d = ld3(x + 3);
                const double n1 = x[6], n2 = x[7];
                const double mu = n1 / n2, q = (n1 - n2) / (n1 + n2);
                const double cosi = dot(N, d);
                const Optics op = hit_optics(N, d, cosi, mu, q * q, n2 >= n1);
                const V3 r = mirror_direction(d, N, cosi);
                y[0] = op.T.x; y[1] = op.T.y; y[2] = op.T.z;
                y[3] = op.trans ? 1.0 : 0.0;
                y[4] = op.R;
                y[5] = r.x; y[6] = r.y; y[7] = r.z;
                break;
            }
            case RT_DEBUG_EASE:             // reference objects.py:176-191, :302-338
                y[0] = ease((int)x[0], x[1], x[2], x[3]);
                break;
            case RT_DEBUG_SCREEN: {         // reference objects.py:45-51, raytrace.py:7-12
                const V3 p = screen_point(sc, x[0], x[1]);
                const V3 d = normalize(p - ld3(sc.cam));
                y[0] = p.x; y[1] = p.y; y[2] = p.z;
                y[3] = d.x; y[4] = d.y; y[5] = d.z;
                break;
            }
            default:
                break;
        }
    }
}

}  // namespace rt
