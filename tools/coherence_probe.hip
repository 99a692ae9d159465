// coherence_probe.hip -- does a load see other XCDs' float64 atomics without a fence?
//
// 256 single-wave workgroups (spread over the 8 XCDs).  Every wave first reads x
// with a plain load (warming its CU's L1 and its XCD's L2), then all waves add
// 1.0 to x with a no-return float64 atomic, then every wave re-reads x three
// ways: plain load, sc1 (agent-scope relaxed atomic) load, returning atomic.
// Phases are separated by a relaxed counter barrier that issues NO cache
// invalidate, so what is counted is raw staleness.  Build:
//   hipcc --offload-arch=gfx950 -O2 -o tools/_build/coherence_probe tools/coherence_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

#define NB 256

__device__ __forceinline__ void barrier_relaxed(unsigned *ctr, unsigned target) {
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && spins < (1L << 26)) {
            __builtin_amdgcn_s_sleep(8);
            ++spins;
        }
    }
    __syncthreads();
}

__device__ __forceinline__ double plain_load(const double *p) {
    double v;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

__global__ void probe(double *x, unsigned *ctr, double *out, int rounds) {
    const int b = blockIdx.x;
    unsigned epoch = 0;
    for (int rnd = 0; rnd < rounds; ++rnd) {
        double warm = plain_load(x);
        double warm2 = __hip_atomic_load(x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        barrier_relaxed(ctr, (++epoch) * NB);
        if (threadIdx.x == 0) {
            (void)__hip_atomic_fetch_add(x, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        barrier_relaxed(ctr, (++epoch) * NB);
        double a = plain_load(x);
        double s = __hip_atomic_load(x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double r = __hip_atomic_fetch_add(x, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0) {
            double *o = out + ((size_t)rnd * NB + b) * 5;
            o[0] = warm; o[1] = warm2; o[2] = a; o[3] = s; o[4] = r;
        }
        barrier_relaxed(ctr, (++epoch) * NB);
    }
}

int main() {
    const int rounds = 20;
    double *x, *out;
    unsigned *ctr;
    hipMalloc(&x, 4096);
    hipMalloc(&ctr, 4096);
    hipMalloc(&out, sizeof(double) * 5 * NB * rounds);
    hipMemset(x, 0, 4096);
    hipMemset(ctr, 0, 4096);
    hipLaunchKernelGGL(probe, dim3(NB), dim3(64), 0, 0, x, ctr, out, rounds);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(e)); return 1; }
    double *h = (double *)malloc(sizeof(double) * 5 * NB * rounds);
    hipMemcpy(h, out, sizeof(double) * 5 * NB * rounds, hipMemcpyDeviceToHost);
    int stale_plain = 0, stale_sc1 = 0, stale_rmw = 0, total = 0;
    for (int rnd = 0; rnd < rounds; ++rnd) {
        const double expect = (double)(rnd + 1) * NB;
        for (int b = 0; b < NB; ++b) {
            const double *o = h + ((size_t)rnd * NB + b) * 5;
            stale_plain += o[2] != expect;
            stale_sc1 += o[3] != expect;
            stale_rmw += o[4] != expect;
            ++total;
        }
    }
    printf("coherence_probe: %d reads per method; stale: plain=%d sc1=%d returning_atomic=%d\n", total, stale_plain,
           stale_sc1, stale_rmw);
    const double *o = h + ((size_t)(rounds - 1) * NB + 17) * 5;
    printf("sample (last round, block 17): warm_plain=%.0f warm_sc1=%.0f plain=%.0f sc1=%.0f rmw=%.0f expect=%.0f\n",
           o[0], o[1], o[2], o[3], o[4], (double)rounds * NB);
    return 0;
}
