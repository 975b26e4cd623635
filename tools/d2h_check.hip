// How a 199 MB device-to-host copy is best made: hipcc -O2 --offload-arch=gfx950 tools/d2h_check.hip -o tools/d2h_check
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = (size_t)3840 * 2160 * 3 * 8;
    void* d; hipMalloc(&d, n); hipMemset(d, 1, n);
    char* h = (char*)malloc(n); memset(h, 0, n);
    hipStream_t s; hipStreamCreate(&s);
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(); hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); double t1 = now();
        printf("pageable            %.2f ms\n", (t1 - t0) * 1e3);
    }
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        hipError_t e = hipHostRegister(h, n, hipHostRegisterDefault); double t1 = now();
        hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); double t2 = now();
        hipHostUnregister(h); double t3 = now();
        printf("register %.2f copy %.2f unregister %.2f total %.2f ms (%s)\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3, hipGetErrorString(e));
    }
    void* p; double t0 = now(); hipHostMalloc(&p, n, hipHostMallocDefault); double t1 = now();
    printf("hipHostMalloc       %.2f ms\n", (t1 - t0) * 1e3);
    for (int rep = 0; rep < 3; ++rep) {
        double a = now(); hipMemcpyAsync(p, d, n, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); double b = now();
        memcpy(h, p, n); double c = now();
        printf("pinned copy %.2f ms, memcpy to pageable %.2f ms\n", (b - a) * 1e3, (c - b) * 1e3);
    }
    // chunked: DMA into two pinned chunks, memcpy out while the next chunk travels
    const size_t chunk = 8 << 20;
    for (int rep = 0; rep < 3; ++rep) {
        double a = now();
        hipEvent_t ev[2]; hipEventCreate(&ev[0]); hipEventCreate(&ev[1]);
        size_t n_chunks = (n + chunk - 1) / chunk;
        for (size_t i = 0; i < n_chunks + 1; ++i) {
            if (i < n_chunks) { size_t off = i * chunk, len = n - off < chunk ? n - off : chunk;
                hipMemcpyAsync((char*)p + (i & 1) * chunk, (char*)d + off, len, hipMemcpyDeviceToHost, s); hipEventRecord(ev[i & 1], s); }
            if (i > 0) { size_t j = i - 1, off = j * chunk, len = n - off < chunk ? n - off : chunk;
                hipEventSynchronize(ev[j & 1]); memcpy(h + off, (char*)p + (j & 1) * chunk, len); }
        }
        double b = now();
        printf("chunked through pinned %.2f ms\n", (b - a) * 1e3);
    }
    return 0;
}
