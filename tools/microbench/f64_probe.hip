// f64_probe.hip -- latency / issue cost of the f64 vector instructions and of LDS round trips on gfx950 (diagnostic).
// One workgroup; cycles by s_memtime around unrolled loops.  Usage: ./f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ void k(double *out, unsigned long long *cyc, int reps) {
    __shared__ double lds[8192];
    (void)0;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += blockDim.x) lds[i] = 1e-3 * (i % 97);
    __syncthreads();
    double a = 1.0 + 1e-9 * lane, b = 1e-9 * lane, c0 = 0.1, c1 = 0.2, c2 = 0.3, c3 = 0.4, c4 = 0.5, c5 = 0.6, c6 = 0.7, c7 = 0.8;
    int idx = lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0) {          // 64 dependent fma
#pragma unroll
            for (int i = 0; i < 64; ++i) c0 = __builtin_fma(c0, a, b);
        } else if (MODE == 1) {   // 64 fma in 8 independent chains
#pragma unroll
            for (int i = 0; i < 8; ++i) { c0 = __builtin_fma(c0, a, b); c1 = __builtin_fma(c1, a, b); c2 = __builtin_fma(c2, a, b); c3 = __builtin_fma(c3, a, b);
                                          c4 = __builtin_fma(c4, a, b); c5 = __builtin_fma(c5, a, b); c6 = __builtin_fma(c6, a, b); c7 = __builtin_fma(c7, a, b); }
        } else if (MODE == 2) {   // 64 dependent add
#pragma unroll
            for (int i = 0; i < 64; ++i) c0 = c0 + a;
        } else if (MODE == 3) {   // 16 dependent LDS round trips (pointer chase, b64)
#pragma unroll
            for (int i = 0; i < 16; ++i) { const double v = lds[idx & 8191]; idx = idx + (int)(v * 0.0) + 64; c0 += v; }
        } else if (MODE == 4) {   // 16 x (8 independent b64 loads, then use): one round trip per batch
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const double *p = lds + ((idx + 8 * i) & 4095);
                const double v0 = p[0], v1 = p[64], v2 = p[128], v3 = p[192], v4 = p[256], v5 = p[320], v6 = p[384], v7 = p[448];
                c0 += ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7));
            }
        } else if (MODE == 5) {   // 16 x (store, barrier-free wave fence, load back)
#pragma unroll
            for (int i = 0; i < 16; ++i) { lds[(lane ^ 1) + 64 * i] = c0; __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); c0 += lds[lane + 64 * i]; }
        } else if (MODE == 6) {   // 16 workgroup barriers
#pragma unroll
            for (int i = 0; i < 16; ++i) { __syncthreads(); c0 += 1.0; }
        } else if (MODE == 7) {   // 16 x (store, workgroup barrier, load from another wave's slot)
#pragma unroll
            for (int i = 0; i < 16; ++i) { lds[tid] = c0; __syncthreads(); c0 += lds[(tid + 64) % blockDim.x]; __syncthreads(); }
        } else if (MODE == 8) {   // 64 dependent v_mul_f64
#pragma unroll
            for (int i = 0; i < 64; ++i) c0 = c0 * a;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + tid] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + idx;
    if (lane == 0) cyc[tid >> 6] = t1 - t0;
}

template <int MODE>
static void run(const char *what, int per_rep, int threads) {
    double *out; unsigned long long *cyc;
    CK(hipMalloc(&out, 1024 * 8)); CK(hipMalloc(&cyc, 16 * 8));
    const int reps = 200;
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, out, cyc, reps); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, out, cyc, reps); CK(hipDeviceSynchronize());
    unsigned long long cs[16]; CK(hipMemcpy(cs, cyc, 8 * (threads / 64), hipMemcpyDeviceToHost));
    unsigned long long c = 0, c0 = cs[0]; for (int i = 0; i < threads / 64; ++i) c = cs[i] > c ? cs[i] : c;
    printf("%-78s %3d waves: %7.1f cycles each (slowest wave; wave 0: %.1f)\n", what, threads / 64, (double)c / reps / per_rep, (double)c0 / reps / per_rep);
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
    for (int th : {64, 128, 256, 512, 1024}) {
        if (th == 64) {
            run<0>("dependent v_fma_f64", 64, th); run<8>("dependent v_mul_f64", 64, th); run<2>("dependent v_add_f64", 64, th);
            run<1>("v_fma_f64, 8 independent chains (issue cost)", 64, th);
            run<3>("dependent LDS round trip (ds_read_b64 -> address)", 16, th);
            run<4>("batch of 8 independent ds_read_b64 + 7 adds", 16, th);
            run<5>("ds_write_b64 -> wave fence -> ds_read_b64", 16, th);
        } else {
            run<1>("v_fma_f64, 8 independent chains (per wave, waves share SIMDs beyond 4)", 64, th);
        }
        run<6>("__syncthreads()", 16, th);
        run<7>("ds_write, __syncthreads, ds_read, __syncthreads", 16, th);
    }
    return 0;
}
