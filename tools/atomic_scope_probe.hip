// atomic_scope_probe.hip -- where do float64 atomic adds execute, and how fast?
//
// Scattered float64 adds into an m-entry table (the shape of colsum / acc / confusion accumulation):
//   A  agent scope (global_atomic_add_f64 ... sc1): performed at the memory side, one table;
//   W  workgroup scope (no sc1), one private copy of the table PER XCD, chosen by the XCC id the
//      wave runs on: all adders of a copy share that XCD's L2, so the L2 is their coherence point;
//      a second kernel sums the 8 copies.
// Checks that both give the same sums (W must not lose an add) and times them.
//   hipcc --offload-arch=gfx950 -O2 -o tools/_build/atomic_scope_probe tools/atomic_scope_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xF;
}

__global__ void add_agent(long n, const int *idx, const float *val, double *table) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride)
        (void)__hip_atomic_fetch_add(table + idx[t], (double)val[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void add_xcd(long n, const int *idx, const float *val, double *copies, long m, unsigned *seen) {
    const unsigned x = xcc_id();
    if (threadIdx.x == 0) atomicOr(seen, 1u << x);
    double *table = copies + (long)x * m;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride)
        (void)__hip_atomic_fetch_add(table + idx[t], (double)val[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ void sum_copies(long m, const double *copies, double *out) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    double s = 0.0;
    for (int x = 0; x < 8; ++x) s += copies[(long)x * m + j];
    out[j] = s;
}

int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 50000000L;
    const long m = argc > 2 ? atol(argv[2]) : 500000L;
    std::vector<int> hi(n);
    std::vector<float> hv(n);
    unsigned long long s = 88172645463325252ull;
    for (long t = 0; t < n; ++t) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        hi[t] = (int)(s % (unsigned long long)m);
        hv[t] = (float)((s >> 40) & 0xFFFF) / 65536.0f;
    }
    int *idx; float *val; double *ta, *copies, *tw; unsigned *seen;
    CHECK(hipMalloc(&idx, n * 4)); CHECK(hipMalloc(&val, n * 4));
    CHECK(hipMalloc(&ta, m * 8)); CHECK(hipMalloc(&copies, 8 * m * 8)); CHECK(hipMalloc(&tw, m * 8)); CHECK(hipMalloc(&seen, 4));
    CHECK(hipMemcpy(idx, hi.data(), n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(val, hv.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = 256 * 8;
    for (int rep = 0; rep < 3; ++rep) {
        float ms_a, ms_w, ms_s;
        CHECK(hipMemset(ta, 0, m * 8)); CHECK(hipMemset(copies, 0, 8 * m * 8)); CHECK(hipMemset(seen, 0, 4));
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(add_agent, dim3(blocks), dim3(256), 0, 0, n, idx, val, ta); CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_a, e0, e1));
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(add_xcd, dim3(blocks), dim3(256), 0, 0, n, idx, val, copies, m, seen); CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_w, e0, e1));
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(sum_copies, dim3((m + 255) / 256), dim3(256), 0, 0, m, copies, tw); CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_s, e0, e1));
        std::vector<double> ha(m), hw(m); unsigned hs;
        CHECK(hipMemcpy(ha.data(), ta, m * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hw.data(), tw, m * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(&hs, seen, 4, hipMemcpyDeviceToHost));
        double maxd = 0, tot_a = 0, tot_w = 0;
        for (long j = 0; j < m; ++j) { double d = ha[j] - hw[j]; if (d < 0) d = -d; if (d > maxd) maxd = d; tot_a += ha[j]; tot_w += hw[j]; }
        printf("n=%ld m=%ld rep %d: agent %.3f ms (%.1f G adds/s) | per-XCD workgroup-scope %.3f ms (%.1f G adds/s) + sum %.3f ms | "
               "max |diff| %.3e total %.6f vs %.6f | XCC ids seen mask 0x%x\n",
               n, m, rep, ms_a, n / ms_a / 1e6, ms_w, n / ms_w / 1e6, ms_s, maxd, tot_a, tot_w, hs);
    }
    return 0;
}
