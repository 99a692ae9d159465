// Accuracy of v_rcp_f64 on gfx950 and of the refinements the Frank-Wolfe scan could build on it:
// max relative error of rcp, rcp + 1 Newton, rcp + 2 Newton, rcp + 1 cubic (Halley-type) step against 1/x.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/rcp_probe.hip -o tools/_build/rcp_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void probe(const double *x, double *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = x[i];
    const double r0 = __builtin_amdgcn_rcp(d);
    const double r1 = __builtin_fma(__builtin_fma(-d, r0, 1.0), r0, r0);
    const double r2 = __builtin_fma(__builtin_fma(-d, r1, 1.0), r1, r1);
    const double e = __builtin_fma(-d, r0, 1.0);
    const double rc = __builtin_fma(r0, __builtin_fma(e, e, e), r0);
    out[4 * i + 0] = r0;
    out[4 * i + 1] = r1;
    out[4 * i + 2] = r2;
    out[4 * i + 3] = rc;
}

int main() {
    const int n = 1 << 22;
    std::vector<double> h(n);
    srand48(7);
    for (int i = 0; i < n; ++i) h[i] = std::ldexp(1.0 + drand48(), (int)(lrand48() % 80) - 60);
    double *dx, *dout;
    hipMalloc(&dx, n * sizeof(double));
    hipMalloc(&dout, 4 * n * sizeof(double));
    hipMemcpy(dx, h.data(), n * sizeof(double), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dout, n);
    std::vector<double> o(4 * (size_t)n);
    if (hipMemcpy(o.data(), dout, o.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    const char *names[4] = {"v_rcp_f64", "+ 1 Newton", "+ 2 Newton", "+ 1 cubic step"};
    for (int k = 0; k < 4; ++k) {
        long double worst = 0;
        for (int i = 0; i < n; ++i) {
            const long double exact = 1.0L / (long double)h[i];
            const long double rel = fabsl(((long double)o[4 * (size_t)i + k] - exact) / exact);
            if (rel > worst) worst = rel;
        }
        printf("%-16s max relative error %.3Le = 2^%.1Lf\n", names[k], worst, log2l(worst));
    }
    return 0;
}
