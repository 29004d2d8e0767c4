// queue_probe.hip -- do long launches on extra streams hold up short launches on the caller's streams? (diagnostic)
// S "caller" streams run R short full-chip launches each; T extra streams (normal or high priority) each carry one long
// one-workgroup launch, enqueued first.  Prints the wall time of the short work alone and beside the long launches.
// usage: queue_probe S T prio(0 = normal, 1 = highest priority) [R]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void spin(unsigned long long cycles, unsigned long long *sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();        // 100 MHz constant clock
    unsigned long long t = t0;
    while (t - t0 < cycles) t = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && sink) sink[blockIdx.x & 1023] = t;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 4, T = argc > 2 ? atoi(argv[2]) : 4, prio = argc > 3 ? atoi(argv[3]) : 0;
    const int R = argc > 4 ? atoi(argv[4]) : 20;
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    std::vector<hipStream_t> cs(S), ts(T);
    for (auto &s : cs) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (auto &s : ts) {
        if (prio) CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest));
        else CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    }
    unsigned long long *sink; CK(hipMalloc(&sink, 1024 * 8));
    // calibrate the counter s_memtime reads (ticks per ms)
    spin<<<1, 64>>>(1000, sink); CK(hipDeviceSynchronize());
    double c0 = now(); spin<<<1, 64>>>(4000000, sink); CK(hipDeviceSynchronize());
    const double per_ms = 4000000.0 / ((now() - c0) * 1e3);
    const unsigned long long c_short = (unsigned long long)(0.2 * per_ms), c_long = (unsigned long long)(20.0 * per_ms);
    auto shorts = [&]() {
        for (int r = 0; r < R; ++r) for (auto &s : cs) spin<<<1024, 128, 0, s>>>(c_short, sink);      // 0.2 ms each, one workgroup per slot
    };
    shorts(); CK(hipDeviceSynchronize());
    double t0 = now(); shorts(); for (auto &s : cs) CK(hipStreamSynchronize(s)); const double alone = now() - t0;
    t0 = now();
    for (auto &s : ts) spin<<<1, 128, 0, s>>>(c_long, sink);            // 20 ms, one workgroup
    shorts();
    for (auto &s : cs) CK(hipStreamSynchronize(s));
    const double beside = now() - t0;
    CK(hipDeviceSynchronize());
    const double all = now() - t0;
    printf("%.0f ticks/ms; priority range [%d, %d]; S=%d caller streams x %d short launches, T=%d %s extra streams with one 20 ms launch each: "
           "short work alone %.2f ms, beside the long launches %.2f ms (everything done after %.2f ms)\n",
           per_ms, least, greatest, S, R, T, prio ? "high-priority" : "normal", alone * 1e3, beside * 1e3, all * 1e3);
    return 0;
}
