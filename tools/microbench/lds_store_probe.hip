// lds_store_probe.hip -- what does a 64-lane ds_write_b64 cost when only 8 lanes carry a result (one per element of a stage vector)?
//   0  owner lanes -> vector, the other 56 lanes -> a per-lane dump area (the "SINK" select of the sweeps: no exec masking)
//   1  every lane -> the element it holds a replica of (8 distinct addresses, 8 lanes each, identical data)
//   2  exec-masked: only the 8 owner lanes store
// 200 dependent-free stores per trip, one wavefront (and two wavefronts on one SIMD); cycles per store from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ void probe(unsigned long long *out, double seed) {
    __shared__ double vec[21 * 8], sink[64 + 21 * 8];
    const int ln = threadIdx.x & 63, r = ln >> 4, b = (ln >> 2) & 3, c = ln & 3;
    const int eB = 4 * (b & 1) + r;
    const bool owner = c == 0 && b < 2;
    double v = seed + eB;
    double *p0 = owner ? vec + eB : sink + ln;
    double *p1 = vec + eB;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int rep = 0; rep < 10; ++rep) {
#pragma unroll
        for (int k = 0; k < 20; ++k) {
            if (MODE == 0) p0[k * 8] = v;
            if (MODE == 1) p1[k * 8] = v;
            if (MODE == 2) { if (owner) p1[k * 8] = v; }
            v += 1.0;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (ln == 0) out[threadIdx.x >> 6] = t1 - t0;
    if (vec[ln & 7] == -1.0) out[7] = 1;
}

int main() {
    unsigned long long *d; CK(hipMalloc(&d, 64));
    unsigned long long h[8];
    const char *names[3] = {"owner lanes -> vector, others -> per-lane dump (SINK)", "all lanes -> their element (same-address replicas)", "exec-masked: owner lanes only"};
    for (int waves = 1; waves <= 8; waves *= 2) {
        for (int m = 0; m < 3; ++m) {
            for (int rep = 0; rep < 2; ++rep) {
                if (m == 0) probe<0><<<1, 64 * waves>>>(d, 1.0);
                if (m == 1) probe<1><<<1, 64 * waves>>>(d, 1.0);
                if (m == 2) probe<2><<<1, 64 * waves>>>(d, 1.0);
                CK(hipDeviceSynchronize());
            }
            CK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost));
            printf("%d wavefront(s) | %-58s | %.1f shader-clock cycles per ds_write_b64 (wave 0)\n", waves, names[m], (double)h[0] / 200.0);
        }
    }
    return 0;
}
