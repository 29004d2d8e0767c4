// mfma_f64_probe.hip -- layout and timing probe for v_mfma_f64_4x4x4_4b_f64 on gfx950 (diagnostic, not shipped).
// Build: hipcc -O3 --offload-arch=gfx950 mfma_f64_probe.hip -o mfma_f64_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ inline double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

// ---- layout: D = A*B + 0 with A[lane] = 1 + lane, B = one-hot at lane lb  -> out[lb][lane]
__global__ void layout_kernel(double *out) {
    const int lane = threadIdx.x;
    for (int lb = 0; lb < 64; ++lb) {
        const double a = 1.0 + lane, b = (lane == lb) ? 1.0 : 0.0;
        out[lb * 64 + lane] = mfma4(a, b, 0.0);
    }
}

__device__ inline void swap16(double v, double &x, double &y) {
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    x = __hiloint2double(b[0], a[0]); y = __hiloint2double(b[1], a[1]);
}

// ---- timing kernels: mode selects the instruction pattern; cycles via s_memtime
template <int MODE>
__global__ void timing_kernel(double *out, unsigned long long *cyc, int reps) {
    const int lane = threadIdx.x & 63;
    double a0 = 1.0 + 1e-3 * lane, a1 = 0.5 - 1e-3 * lane, a2 = 0.25 + 1e-4 * lane, a3 = 0.125;
    double v = 1e-3 * lane, w = 2e-3 * lane, u = 3e-3, t = 4e-3;
    double f0 = 1.0 + 1e-6 * lane, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0) {            // dependent through C: 4 MFMAs per rep
            v = mfma4(a0, w, v); v = mfma4(a1, w, v); v = mfma4(a2, w, v); v = mfma4(a3, w, v);
        } else if (MODE == 1) {     // dependent through B
            v = mfma4(a0, v, u); v = mfma4(a1, v, u); v = mfma4(a2, v, u); v = mfma4(a3, v, u);
        } else if (MODE == 2) {     // 4 independent accumulators
            v = mfma4(a0, u, v); w = mfma4(a1, u, w); t = mfma4(a2, u, t); f3 = mfma4(a3, u, f3);
        } else if (MODE == 3) {     // chain step as planned: swap16 -> mfma -> mfma (dep through C); 2 steps per rep
            double x, y;
            swap16(v, x, y); v = mfma4(a0, x, u); v = mfma4(a1, y, v);
            swap16(v, x, y); v = mfma4(a2, x, u); v = mfma4(a3, y, v);
        } else if (MODE == 4) {     // chain step + 2 off-chain MFMAs per step
            double x, y;
            swap16(v, x, y); v = mfma4(a0, x, u); v = mfma4(a1, y, v); w = mfma4(a2, x, w); w = mfma4(a3, y, w);
            swap16(v, x, y); v = mfma4(a2, x, u); v = mfma4(a3, y, v); t = mfma4(a0, x, t); t = mfma4(a1, y, t);
        } else if (MODE == 5) {     // 4 independent MFMAs + 8 independent f64 FMAs (overlap test)
            v = mfma4(a0, u, v); w = mfma4(a1, u, w); t = mfma4(a2, u, t); f3 = mfma4(a3, u, f3);
            f0 = fma(f0, 1.0000001, 1e-9); f1 = fma(f1, 1.0000001, 1e-9); f2 = fma(f2, 1.0000001, 1e-9); a3 = fma(a3, 1.0000001, 1e-9);
            f0 = fma(f0, 1.0000001, 1e-9); f1 = fma(f1, 1.0000001, 1e-9); f2 = fma(f2, 1.0000001, 1e-9); a3 = fma(a3, 1.0000001, 1e-9);
        } else if (MODE == 6) {     // 8 independent f64 FMAs only
            f0 = fma(f0, 1.0000001, 1e-9); f1 = fma(f1, 1.0000001, 1e-9); f2 = fma(f2, 1.0000001, 1e-9); f3 = fma(f3, 1.0000001, 1e-9);
            f0 = fma(f0, 1.0000001, 1e-9); f1 = fma(f1, 1.0000001, 1e-9); f2 = fma(f2, 1.0000001, 1e-9); f3 = fma(f3, 1.0000001, 1e-9);
        } else if (MODE == 7) {     // dependent f64 FMA chain (8 per rep)
            f0 = fma(f0, 1.0000001, 1e-9); f0 = fma(f0, 1.0000001, 1e-9); f0 = fma(f0, 1.0000001, 1e-9); f0 = fma(f0, 1.0000001, 1e-9);
            f0 = fma(f0, 1.0000001, 1e-9); f0 = fma(f0, 1.0000001, 1e-9); f0 = fma(f0, 1.0000001, 1e-9); f0 = fma(f0, 1.0000001, 1e-9);
        } else if (MODE == 8) {     // chain step with a DPP-based dup instead of permlane swap: v_mov dpp row_shr... emulate with __shfl (bpermute)
            double x = __shfl(v, lane & 47), y = __shfl(v, (lane & 47) | 16);
            v = mfma4(a0, x, u); v = mfma4(a1, y, v);
            x = __shfl(v, lane & 47); y = __shfl(v, (lane & 47) | 16);
            v = mfma4(a2, x, u); v = mfma4(a3, y, v);
        } else if (MODE == 9) {     // dependent 32-bit DPP mov + f64 add chain (the present reduction step), 4 per rep
            for (int q = 0; q < 4; ++q) {
                int lo = __double2loint(f0), hi = __double2hiint(f0);
                lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);
                hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
                f0 += __hiloint2double(hi, lo);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v + w + t + u + f0 + f1 + f2 + f3 + a3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run_timing(const char *name, int per_rep, double *dout, unsigned long long *dcyc) {
    const int reps = 2000;
    for (int wpb = 1; wpb <= 8; wpb *= 2) {      // waves per block (1 block per CU): 4 = one per SIMD, 8 = two per SIMD
        if (wpb == 2) continue;
        hipLaunchKernelGGL(timing_kernel<MODE>, dim3(256), dim3(64 * wpb), 0, 0, dout, dcyc, reps);
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(timing_kernel<MODE>, dim3(256), dim3(64 * wpb), 0, 0, dout, dcyc, reps);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> c(256);
        CK(hipMemcpy(c.data(), dcyc, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double s = 0; for (auto x : c) s += (double)x;
        printf("%-46s waves/CU=%d: %.1f cycles per rep (%d ops/rep -> %.2f cyc/op)\n", name, wpb, s / 256 / reps, per_rep, s / 256 / reps / per_rep);
    }
}

int main() {
    double *dout; unsigned long long *dcyc;
    CK(hipMalloc(&dout, 64 * 64 * 64 * sizeof(double)));
    CK(hipMalloc(&dcyc, 1024 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dout);
    CK(hipDeviceSynchronize());
    std::vector<double> h(64 * 64);
    CK(hipMemcpy(h.data(), dout, 64 * 64 * sizeof(double), hipMemcpyDeviceToHost));
    printf("layout: for B one-hot at lane lb, the D lanes that are non-zero and the A lane they picked\n");
    for (int lb = 0; lb < 64; ++lb) {
        printf("lb=%2d:", lb);
        for (int l = 0; l < 64; ++l) if (h[lb * 64 + l] != 0.0) printf(" D[%d]<-A[%d]", l, (int)h[lb * 64 + l] - 1);
        printf("\n");
    }
    // check hypothesis: block = lane/16; A[i][k]: lane = 16b + 4k + i;  B[k][j]: lane = 16b + 4k + j;  D[i][j]: lane = 16b + 4i + j
    int bad = 0;
    for (int lb = 0; lb < 64; ++lb) {
        const int b = lb / 16, k = (lb % 16) / 4, j = lb % 4;
        for (int l = 0; l < 64; ++l) {
            double expect = 0.0;
            if (l / 16 == b && l % 4 == j) { const int i = (l % 16) / 4; expect = 1.0 + (16 * b + 4 * k + i); }
            if (h[lb * 64 + l] != expect) ++bad;
        }
    }
    printf("hypothesis A: lane=16b+4k+i, B: lane=16b+4k+j, D: lane=16b+4i+j  -> %s (%d mismatches)\n", bad ? "WRONG" : "CONFIRMED", bad);
    run_timing<0>("mfma dep via C (4/rep)", 4, dout, dcyc);
    run_timing<1>("mfma dep via B (4/rep)", 4, dout, dcyc);
    run_timing<2>("mfma independent (4/rep)", 4, dout, dcyc);
    run_timing<3>("chain step swap16+2 mfma (2 steps/rep)", 2, dout, dcyc);
    run_timing<4>("chain step + 2 off-chain mfma (2 steps/rep)", 2, dout, dcyc);
    run_timing<5>("4 indep mfma + 8 indep f64 fma", 1, dout, dcyc);
    run_timing<6>("8 indep f64 fma", 8, dout, dcyc);
    run_timing<7>("8 dep f64 fma", 8, dout, dcyc);
    run_timing<8>("chain step bpermute dup + 2 mfma (2 steps/rep)", 2, dout, dcyc);
    run_timing<9>("dep dpp-mov x2 + f64 add (4/rep)", 4, dout, dcyc);
    return 0;
}
