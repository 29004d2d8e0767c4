// bigwg_probe.hip -- when does a workgroup that needs a WHOLE CU (512 threads, 150 KB of LDS) start while short full-chip
// launches (128-thread workgroups with 40 KB of LDS: four per CU) keep the GPU saturated from four streams?  (diagnostic for
// the straggler tail kernel: DESIGN.md section 5)
// usage: bigwg_probe prio(0 = normal, 1 = highest priority stream for the big workgroup)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void spin(unsigned long long ticks, unsigned long long *out, int jitter) {
    extern __shared__ double lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();        // 100 MHz constant clock
    // workgroups of a launch finish at different times, as the solve kernel's do (between 50 % and 150 % of `ticks`)
    const unsigned long long mine = jitter ? ticks / 2 + (ticks * ((blockIdx.x * 2654435761u) & 1023)) / 1024 : ticks;
    unsigned long long t = t0;
    while (t - t0 < mine) t = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { lds[0] = (double)t; if (out) { out[2 * (blockIdx.x & 1023)] = t0; out[2 * (blockIdx.x & 1023) + 1] = t; } }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const int prio = argc > 1 ? atoi(argv[1]) : 0;
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    std::vector<hipStream_t> cs(4);
    for (auto &s : cs) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipStream_t big;
    if (prio) CK(hipStreamCreateWithPriority(&big, hipStreamNonBlocking, greatest)); else CK(hipStreamCreateWithFlags(&big, hipStreamNonBlocking));
    CK(hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    unsigned long long *o_small, *o_big; CK(hipMalloc(&o_small, 2048 * 8)); CK(hipMalloc(&o_big, 2048 * 8));
    const unsigned long long per_ms = 100000;          // s_memrealtime: 100 MHz
    spin<<<1, 64, 1024>>>(1000, o_small, 0); CK(hipDeviceSynchronize());
    const int R = 12;                                   // launches per stream: 48 x 0.3 ms of full-chip work = ~15 ms
    // the big workgroup is enqueued after the first wave of small launches, i.e. onto a saturated GPU
    double t0 = now();
    unsigned long long *o_ref; CK(hipMalloc(&o_ref, 16)); spin<<<1, 64, 1024, big>>>(10, o_ref, 0); CK(hipStreamSynchronize(big));
    for (int r = 0; r < R; ++r) {
        for (auto &s : cs) spin<<<1024, 128, 40 * 1024, s>>>((unsigned long long)(0.3 * per_ms), o_small, 1);
        if (r == 2) spin<<<1, 512, 150 * 1024, big>>>((unsigned long long)(1.0 * per_ms), o_big, 0);
    }
    CK(hipDeviceSynchronize());
    const double wall = now() - t0;
    unsigned long long ref[2], b[2]; CK(hipMemcpy(ref, o_ref, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(b, o_big, 16, hipMemcpyDeviceToHost));
    printf("big workgroup on a %s stream, enqueued behind 12 of 48 saturating launches: started %.2f ms and ended %.2f ms after the reference launch; everything done after %.2f ms of wall time\n",
           prio ? "highest-priority" : "normal", (double)(b[0] - ref[1]) / per_ms, (double)(b[1] - ref[1]) / per_ms, wall * 1e3);
    return 0;
}
