// Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on the access widths the
// tracer's kernels use: streams N bytes with 4, 8 and 16 bytes per lane.
//   hipcc -O3 --offload-arch=gfx950 tools/pmc_calib.hip -o tools/pmc_calib
//   rocprofv3 --pmc FETCH_SIZE -d out -- tools/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename T>
__global__ void copy_w(const T* __restrict__ in, T* __restrict__ out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

int main() {
    const size_t bytes = (size_t)1 << 30;   // 1 GiB in, 1 GiB out: far beyond the 256 MiB on-die cache
    void *a, *b;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMemset(a, 1, bytes));
    CHECK(hipMemset(b, 0, bytes));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(copy_w<float>, dim3(8192), dim3(256), 0, 0, (const float*)a, (float*)b, bytes / 4);
        hipLaunchKernelGGL(copy_w<double>, dim3(8192), dim3(256), 0, 0, (const double*)a, (double*)b, bytes / 8);
        hipLaunchKernelGGL(copy_w<double2>, dim3(8192), dim3(256), 0, 0, (const double2*)a, (double2*)b, bytes / 16);
    }
    CHECK(hipDeviceSynchronize());
    printf("streamed %zu bytes in and out per kernel\n", bytes);
    return 0;
}
