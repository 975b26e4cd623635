// Accuracy of the gfx950 hardware sin/cos (revolutions input), rsq, sqrt, exp2, log2 against fp64.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    double es = 0, ec = 0, er = 0, eq = 0, ee = 0, el = 0;
    for (int j = i; j < n; j += gridDim.x * blockDim.x) {
        double u = (j + 0.37) / n;
        float uf = (float)u;
        float s = __builtin_amdgcn_sinf(uf), c = __builtin_amdgcn_cosf(uf);
        es = fmax(es, fabs((double)s - sin(6.283185307179586 * (double)uf)));
        ec = fmax(ec, fabs((double)c - cos(6.283185307179586 * (double)uf)));
        float x = (float)(u * 1000.0 + 1e-3);
        er = fmax(er, fabs((double)__builtin_amdgcn_rsqf(x) * sqrt((double)x) - 1.0));
        eq = fmax(eq, fabs((double)__builtin_amdgcn_sqrtf(x) / sqrt((double)x) - 1.0));
        float y = (float)(-u * 20.0);
        ee = fmax(ee, fabs((double)__builtin_amdgcn_exp2f(y) / exp2((double)y) - 1.0));
        float z = (float)(u + 1e-6);
        el = fmax(el, fabs((double)__builtin_amdgcn_logf(z) - log2((double)z)));
    }
    out[i * 6 + 0] = es; out[i * 6 + 1] = ec; out[i * 6 + 2] = er; out[i * 6 + 3] = eq; out[i * 6 + 4] = ee; out[i * 6 + 5] = el;
}
int main() {
    const int T = 256 * 256; double* d; hipMalloc(&d, T * 6 * 8);
    k<<<256, 256>>>(d, 1 << 26);
    static double h[T * 6]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    double m[6] = {0};
    for (int i = 0; i < T; ++i) for (int j = 0; j < 6; ++j) m[j] = fmax(m[j], h[i * 6 + j]);
    printf("max abs err sin %.3e cos %.3e | rel err rsq %.3e sqrt %.3e exp2 %.3e | abs err log2 %.3e\n", m[0], m[1], m[2], m[3], m[4], m[5]);
    return 0;
}
