// clock_probe.hip -- what s_memtime counts, and the shader clock a lone workgroup gets against a full chip:
// a dependent chain of v_fma_f64 (5.5 cycles each at any clock) timed by s_memtime, s_getreg SHADER_CYCLES-free, and by HIP events.
// hipcc --offload-arch=gfx950 -O2 -o clock_probe.bin clock_probe.hip && ./clock_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(double *out, unsigned long long *ticks, int n) {
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-9, c = 0.1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) c = __builtin_fma(c, a, b);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
    double *out; unsigned long long *ticks;
    hipMalloc(&out, 4096 * 512 * 8); hipMalloc(&ticks, 4096 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 20000;
    for (int rep = 0; rep < 2; ++rep)
    for (int cfg = 0; cfg < 4; ++cfg) {
        const int grid = cfg == 0 ? 1 : (cfg == 1 ? 16 : (cfg == 2 ? 256 : 2048)), block = cfg == 3 ? 256 : 64;
        hipEventRecord(e0); chain<<<grid, block>>>(out, ticks, n); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        printf("grid %4d x %3d threads: %.3f ms, s_memtime ticks of workgroup 0: %llu -> %.1f ticks/us; %.2f ns per dependent v_fma_f64 (%.2f ticks)\n",
               grid, block, ms, t, t / (ms * 1e3), ms * 1e6 / (64.0 * n), (double)t / (64.0 * n));
    }
    return 0;
}
