// Shader clock probe: a long chain of dependent FMAs on every CU, timed with
// HIP events (wall time) and with the shader cycle counter, gives the engine
// clock that kernels really run at.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void chain(float* out, long long* cycles, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a = fmaf(a, b, 1e-7f);
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    const int blocks = 1024, threads = 256, iters = 20000;
    float* out;
    long long* cyc;
    if (hipMalloc(&out, blocks * threads * sizeof(float)) != hipSuccess) return 1;
    if (hipMalloc(&cyc, blocks * sizeof(long long)) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(chain, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        long long h[4];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        // 1024 blocks of 4 waves on 256 CUs x 4 SIMDs: one wave per SIMD, one round
        const double fma_per_wave = 16.0 * iters;
        std::printf("wall %.3f ms, counter delta %lld, counter MHz %.0f, cycles per dependent FMA (at counter rate) %.2f, ns per FMA %.2f\n",
                    ms, h[0], h[0] / (ms * 1e3), static_cast<double>(h[0]) / fma_per_wave,
                    ms * 1e6 / fma_per_wave);
    }
    return 0;
}
