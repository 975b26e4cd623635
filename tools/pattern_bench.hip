// Memory pattern of k_trace in isolation: every "ray" reads a parent index, gathers 7 doubles and a word
// through it (indices ascend, about every second parent is used), and writes 8 doubles and a word.
// Compares the assignment of 64-ray tiles to waves: blocked (a wave owns one contiguous run of the level,
// as the kernels do) against segmented (the level is cut into segments that all waves work through
// together, a wave owning a short run in each), and plain against streaming stores.
//   hipcc -O3 --offload-arch=gfx950 pattern_bench.hip -o pattern_bench;  ./pattern_bench RAYS [dense]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int NB = 2048, BLOCK = 256, WPB = 4, TILE = 64;
struct Arrays { double *a[7], *o[8]; uint32_t *link, *info_in, *info_out; };
template <bool STREAM>   // STREAM: streaming (nontemporal) stores
__global__ __launch_bounds__(BLOCK, 4) void k(Arrays A, int nt, int segments) {
    const int lane = threadIdx.x & 63, wid = blockIdx.x * WPB + (threadIdx.x >> 6);
    const int W = NB * WPB;
    const int per_seg = (nt + segments - 1) / segments;            // tiles per segment
    const int q = (per_seg + W - 1) / W;                           // tiles of a wave per segment
    for (int s = 0; s < segments; ++s) {
        const int t0 = s * per_seg + wid * q, t1 = min(min(nt, (s + 1) * per_seg), t0 + q);
        for (int t = t0; t < t1; ++t) {
            const uint32_t i = (uint32_t)t * TILE + lane;
            const uint32_t p = A.link[i];
            double v[7];
#pragma unroll
            for (int k2 = 0; k2 < 7; ++k2) v[k2] = A.a[k2][p];
            const uint32_t in = A.info_in[p];
            double acc = (double)in;
#pragma unroll
            for (int k2 = 0; k2 < 7; ++k2) acc = acc * 1.0000001 + v[k2];
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) { if (STREAM) __builtin_nontemporal_store(acc + k2, &A.o[k2][i]); else A.o[k2][i] = acc + k2; }
            A.info_out[i] = in + 1;
        }
    }
}
int main(int argc, char** argv) {
    const size_t n = (argc > 1 ? atoll(argv[1]) : 25000000) / TILE * TILE;   // rays of the level
    const bool dense = argc > 2;                                              // any second argument: every parent is used
    const size_t np = 2 * n;                                                  // parents
    Arrays A;
    for (auto& p : A.a) { CK(hipMalloc(&p, np * 8)); CK(hipMemset(p, 0, np * 8)); }
    for (auto& p : A.o) { CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8)); }
    CK(hipMalloc(&A.link, n * 4)); CK(hipMalloc(&A.info_in, np * 4)); CK(hipMalloc(&A.info_out, n * 4));
    CK(hipMemset(A.info_in, 0, np * 4));
    std::vector<uint32_t> link(n);
    uint32_t x = 12345, p = 0;
    for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; p += dense ? 1 : 1 + ((x >> 16) % 3); if (p >= np) p = np - 1; link[i] = p; }
    CK(hipMemcpy(A.link, link.data(), n * 4, hipMemcpyHostToDevice));
    const double bytes = n * (4.0 + 7 * 8 + 4 + 8 * 8 + 4);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nt = (int)(n / TILE);
    for (int segments : {1, 2, 4, 8, 16, 32, 64, 128, 1, -1}) {
        const bool stream = segments < 0;      // last line: one segment, streaming stores
        if (stream) segments = 1;
        auto launch = [&]() {
            if (stream) hipLaunchKernelGGL(k<true>, dim3(NB), dim3(BLOCK), 0, 0, A, nt, segments);
            else hipLaunchKernelGGL(k<false>, dim3(NB), dim3(BLOCK), 0, 0, A, nt, segments);
        };
        for (int w = 0; w < 2; ++w) launch();
        CK(hipEventRecord(e0));
        const int reps = 10;
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("rays %zu %s segments %3d%s: %.3f ms  %.2f TB/s (algorithmic)\n", n, dense ? "dense" : "sparse", segments, stream ? " streaming stores" : "", ms / reps, bytes / (ms / reps * 1e-3) / 1e12);
    }
    return 0;
}
