// step_probe.hip -- latency of one chain step of the MFMA sweeps in several variants (diagnostic)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../autonomous-racing-lpv-mpp-mpc_amd/csrc/wave_ops.hpp"
using namespace lpvmpc;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ void k(double *out, unsigned long long *cyc, int reps) {
    __shared__ double lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 1e-3 * i;
    __syncthreads();
    double a0 = 1.0 + 1e-3 * lane, a1 = 0.5 - 1e-3 * lane;
    double Y = 1e-3 * lane, Dl = 0.0;
    const bool second = (lane >> 3) & 1;
    const double *src = lds + (lane >> 4) + 4 * ((lane >> 2) & 1);
    double *dst = lds + 2048 + lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            if (MODE == 0) {            // bare: two dependent MFMAs, Y feeds B of the first
                const double t = mfma4(a0, Y, 1.0); Y = mfma4(a1, Y, t);
            } else if (MODE == 1) {     // + swap_half on the second operand
                const double Ysw = swap_half(Y); const double t = mfma4(a0, Y, 1.0); Y = mfma4(a1, Ysw, t);
            } else if (MODE == 2) {     // + dup01 on the result
                const double Ysw = swap_half(Y); const double t = mfma4(a0, Y, 1.0); Y = dup01(mfma4(a1, Ysw, t));
            } else if (MODE == 3) {     // + C from LDS with the cndmask
                const double bl = src[p * 8]; const double bp = second ? 0.0 : bl;
                const double Ysw = swap_half(Y); const double t = mfma4(a0, Y, bp); Y = dup01(mfma4(a1, Ysw, t));
            } else if (MODE == 4) {     // + store of the result
                const double bl = src[p * 8]; const double bp = second ? 0.0 : bl;
                const double Ysw = swap_half(Y); const double t = mfma4(a0, Y, bp); const double D = mfma4(a1, Ysw, t);
                dst[p * 64] = D; Y = dup01(D);
            } else if (MODE == 5) {     // natural layout: no dup, no cndmask, C from LDS, store
                const double bl = src[p * 8];
                const double Ysw = swap_half(Y); const double t = mfma4(a0, Y, bl); Y = mfma4(a1, Ysw, t);
                dst[p * 64] = Y;
            } else if (MODE == 6) {     // natural layout + 2 independent MFMAs (pivot product) per step
                const double bl = src[p * 8];
                const double Ysw = swap_half(Y); const double t = mfma4(a0, Y, bl); const double u = mfma4(a1, Y, 0.0);
                const double Yn = mfma4(a1, Ysw, t); dst[p * 64] = mfma4(a0, Ysw, u); Y = Yn;
            } else if (MODE == 8) {     // as built, the store delayed by one step
                const double bl = src[p * 8]; const double bp = second ? 0.0 : bl;
                const double Ysw = swap_half(Y); const double t = mfma4(a0, Y, bp);
                dst[p * 64] = Dl;
                __builtin_amdgcn_sched_barrier(0);
                const double D = mfma4(a1, Ysw, t); Dl = D; Y = dup01(D);
            } else if (MODE == 9) {     // natural layout, store delayed
                const double bl = src[p * 8];
                const double Ysw = swap_half(Y); const double t = mfma4(a0, Y, bl);
                dst[p * 64] = Dl;
                __builtin_amdgcn_sched_barrier(0);
                Y = mfma4(a1, Ysw, t); Dl = Y;
            } else if (MODE == 10) {    // four-block chain: one MFMA + cross-block add (row_ror:8), C from LDS
                const double bl = src[p * 8]; const double bp = second ? 0.0 : bl;
                const double D = mfma4(a0, Y, bp);
                Y = D + dpp_mov<0x128>(D);
            } else if (MODE == 11) {    // four-block chain, alternating row_ror:8 / row_half_mirror, + independent pivot MFMA, add and store
                const double bl = src[p * 8]; const double bp = second ? 0.0 : bl;
                const double D = mfma4(a0, Y, bp);
                const double V = mfma4(a1, Y, 0.0);
                if (p & 1) { Y = D + dpp_mov<0x141>(D); dst[p * 64] = V + dpp_mov<0x141>(V); }
                else       { Y = D + dpp_mov<0x128>(D); dst[p * 64] = V + dpp_mov<0x128>(V); }
            } else if (MODE == 12) {    // as 11, the pivot product's add and store delayed by one step
                const double bl = src[p * 8]; const double bp = second ? 0.0 : bl;
                const double D = mfma4(a0, Y, bp);
                if (p & 1) dst[p * 64] = Dl + dpp_mov<0x128>(Dl); else dst[p * 64] = Dl + dpp_mov<0x141>(Dl);
                Dl = mfma4(a1, Y, 0.0);
                if (p & 1) Y = D + dpp_mov<0x141>(D); else Y = D + dpp_mov<0x128>(D);
            } else if (MODE == 7) {     // split accumulation: two independent MFMAs + f64 add
                const double bl = src[p * 8];
                const double Ysw = swap_half(Y); const double t = mfma4(a0, Y, bl); const double u = mfma4(a1, Ysw, 0.0);
                Y = t + u; dst[p * 64] = Y;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = Y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
static void run(const char *name, double *dout, unsigned long long *dcyc) {
    const int reps = 500;
    for (int wpb = 1; wpb <= 8; wpb *= 2) {
        if (wpb == 2) continue;
        for (int it = 0; it < 2; ++it) { hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * wpb), 0, 0, dout, dcyc, reps); CK(hipDeviceSynchronize()); }
        std::vector<unsigned long long> c(256);
        CK(hipMemcpy(c.data(), dcyc, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double s = 0; for (auto x : c) s += (double)x;
        printf("%-58s waves/CU=%d: %.1f cycles per step\n", name, wpb, s / 256 / reps / 8);
    }
}
int main() {
    double *dout; unsigned long long *dcyc;
    CK(hipMalloc(&dout, 256 * 512 * sizeof(double))); CK(hipMalloc(&dcyc, 1024 * sizeof(unsigned long long)));
    run<0>("two dependent MFMAs", dout, dcyc);
    run<1>("+ swap_half", dout, dcyc);
    run<2>("+ dup01", dout, dcyc);
    run<3>("+ C from LDS, cndmask", dout, dcyc);
    run<4>("+ store (= the step as built)", dout, dcyc);
    run<5>("natural layout: swap, C from LDS, store", dout, dcyc);
    run<6>("natural layout + 2 independent MFMAs", dout, dcyc);
    run<7>("two independent MFMAs + f64 add", dout, dcyc);
    run<8>("as built, store delayed one step", dout, dcyc);
    run<9>("natural layout, store delayed one step", dout, dcyc);
    run<10>("four-block chain: 1 MFMA + ror8 add, C from LDS", dout, dcyc);
    run<11>("four-block chain + pivot MFMA + add + store", dout, dcyc);
    run<12>("four-block chain + pivot MFMA, its add/store one step late", dout, dcyc);
    return 0;
}
