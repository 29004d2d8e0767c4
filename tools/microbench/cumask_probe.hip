// cumask_probe.hip -- do CU-masked streams (hipExtStreamCreateWithCUMask) isolate a whole-CU workgroup (512 threads, 150 KB of LDS:
// the straggler tail kernel) from short full-chip launches (128-thread workgroups with 40 KB of LDS, four per CU) on this pool?
// The saturating launches go to four streams masked to the COMPLEMENT of the reserved CUs (mode 1) or to plain streams (mode 0);
// the big workgroups go to a stream masked to the reserved CUs.  Reported: which (xcc, se, sh, cu) each class ran on, the start
// latency of the big workgroups and the wall time of the saturating launches (the cost of giving CUs away).
// usage: cumask_probe mode(0 plain main streams | 1 complement-masked main streams) reserved_cus [first_bit] [stride]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void spin(unsigned long long ticks, unsigned long long *out, int jitter, int slots) {
    extern __shared__ double lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();        // 100 MHz constant clock
    const unsigned long long mine = jitter ? ticks / 2 + (ticks * ((blockIdx.x * 2654435761u) & 1023)) / 1024 : ticks;
    unsigned long long t = t0;
    while (t - t0 < mine) t = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        lds[0] = (double)t;
        if (out) {
            const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));
            unsigned long long *o = out + 3 * (blockIdx.x % slots);
            o[0] = t0; o[1] = t; o[2] = ((unsigned long long)xcc << 32) | hw;
        }
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static unsigned where(unsigned long long w) {       // (xcc, se, sh, cu) packed
    const unsigned hw = (unsigned)w, xcc = (unsigned)(w >> 32) & 15;
    return (xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
}

int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 1, reserved = argc > 2 ? atoi(argv[2]) : 8;
    const int first = argc > 3 ? atoi(argv[3]) : 0, stride = argc > 4 ? atoi(argv[4]) : 1;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
    std::vector<uint32_t> mres(words, 0), mcomp(words, 0);
    for (int i = 0; i < ncu; ++i) mcomp[i / 32] |= 1u << (i % 32);
    for (int r = 0; r < reserved; ++r) { const int b = (first + r * stride) % ncu; mres[b / 32] |= 1u << (b % 32); mcomp[b / 32] &= ~(1u << (b % 32)); }
    std::vector<hipStream_t> cs(4);
    for (auto &s : cs) { if (mode) CK(hipExtStreamCreateWithCUMask(&s, words, mcomp.data())); else CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); }
    hipStream_t big; CK(hipExtStreamCreateWithCUMask(&big, words, mres.data()));
    std::vector<uint32_t> back(words, 0); CK(hipExtStreamGetCUMask(big, words, back.data()));
    CK(hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int SL = 4096, NB = 16;
    unsigned long long *o_small, *o_big, *o_ref; CK(hipMalloc(&o_small, SL * 24)); CK(hipMalloc(&o_big, NB * 24)); CK(hipMalloc(&o_ref, 24));
    CK(hipMemset(o_small, 0, SL * 24)); CK(hipMemset(o_big, 0, NB * 24));
    const unsigned long long per_ms = 100000;
    for (auto &s : cs) spin<<<1, 64, 1024, s>>>(1000, nullptr, 0, 1);
    spin<<<1, 64, 1024, big>>>(1000, nullptr, 0, 1); CK(hipDeviceSynchronize());
    const int R = 12;
    double t0 = now();
    spin<<<1, 64, 1024, big>>>(10, o_ref, 0, 1); CK(hipStreamSynchronize(big));
    for (int r = 0; r < R; ++r) {
        for (auto &s : cs) spin<<<1024, 128, 40 * 1024, s>>>((unsigned long long)(0.3 * per_ms), o_small, 1, SL);
        // onto a saturated GPU: NB big workgroups of 1 ms each (more than the reserved CUs hold at once when reserved < NB)
        if (r == 2) spin<<<NB, 512, 150 * 1024, big>>>((unsigned long long)(1.0 * per_ms), o_big, 0, NB);
    }
    for (auto &s : cs) CK(hipStreamSynchronize(s));
    const double wall_small = now() - t0;
    CK(hipDeviceSynchronize());
    const double wall = now() - t0;
    std::vector<unsigned long long> hs(SL * 3), hb(NB * 3), ref(3);
    CK(hipMemcpy(hs.data(), o_small, SL * 24, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), o_big, NB * 24, hipMemcpyDeviceToHost)); CK(hipMemcpy(ref.data(), o_ref, 24, hipMemcpyDeviceToHost));
    std::set<unsigned> cu_small, cu_big;
    for (int i = 0; i < SL; ++i) if (hs[3 * i]) cu_small.insert(where(hs[3 * i + 2]));
    double first_start = 1e9, last_start = 0, last_end = 0;
    for (int i = 0; i < NB; ++i) {
        cu_big.insert(where(hb[3 * i + 2]));
        const double st = (double)(hb[3 * i] - ref[1]) / per_ms, en = (double)(hb[3 * i + 1] - ref[1]) / per_ms;
        first_start = st < first_start ? st : first_start; last_start = st > last_start ? st : last_start; last_end = en > last_end ? en : last_end;
    }
    int shared = 0; for (unsigned c : cu_big) shared += cu_small.count(c);
    printf("mode %d (%s main streams), %d reserved CUs (bits %d + %d k), %d CUs on the device; mask read back %08x %08x ...\n", mode, mode ? "complement-masked" : "plain", reserved, first, stride, ncu, back[0], words > 1 ? back[1] : 0);
    printf("  small workgroups ran on %zu distinct CUs, the %d big ones on %zu; %d of the big ones' CUs also ran small workgroups\n", cu_small.size(), NB, cu_big.size(), shared);
    printf("  big workgroups (1 ms each, enqueued behind 12 of 48 saturating launches): first start %.2f ms, last start %.2f ms, last end %.2f ms after the reference\n", first_start, last_start, last_end);
    printf("  saturating launches done after %.2f ms of wall time, everything after %.2f ms\n", wall_small * 1e3, wall * 1e3);
    printf("  CUs of the big workgroups (xcc.se.sh.cu):");
    for (unsigned c : cu_big) printf(" %u.%u.%u.%u", c >> 12, (c >> 8) & 15, (c >> 4) & 15, c & 15);
    printf("\n");
    return 0;
}
