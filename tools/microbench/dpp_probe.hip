// dpp_probe.hip -- checks the DPP permutations used by the MFMA sweeps against ds_bpermute shuffles (diagnostic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int CTRL, int BANK>
__device__ inline double dpp_into(double old, double src) {
    int ol = __double2loint(old), oh = __double2hiint(old), sl = __double2loint(src), sh = __double2hiint(src);
    ol = __builtin_amdgcn_update_dpp(ol, sl, CTRL, 0xF, BANK, false);
    oh = __builtin_amdgcn_update_dpp(oh, sh, CTRL, 0xF, BANK, false);
    return __hiloint2double(oh, ol);
}
__device__ inline double swap_half(double v) {          // lane ^ 4
    double t = dpp_into<0x104, 0x5>(v, v);               // row_shl:4 -> quads 0,2 take quads 1,3
    return dpp_into<0x114, 0xA>(t, v);                   // row_shr:4 -> quads 1,3 take quads 0,2
}
__device__ inline double dup01(double v) { return dpp_into<0x128, 0xC>(v, v); }     // row_ror:8 into quads 2,3
__device__ inline double take23(double v) { return dpp_into<0x128, 0x3>(v, v); }    // row_ror:8 into quads 0,1

__global__ void k(double *out) {
    const int lane = threadIdx.x;
    const double v = 100.0 + lane;
    out[lane] = swap_half(v);            out[64 + lane] = __shfl(v, lane ^ 4);
    out[128 + lane] = dup01(v);          out[192 + lane] = __shfl(v, lane & ~8);
    out[256 + lane] = take23(v);         out[320 + lane] = __shfl(v, lane | 8);
}
int main() {
    double *d, h[384];
    CK(hipMalloc(&d, sizeof(h)));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    const char *names[3] = {"swap_half (lane^4)", "dup01 (lane&~8)", "take23 (lane|8)"};
    for (int t = 0; t < 3; ++t) {
        int bad = 0;
        for (int l = 0; l < 64; ++l) if (h[128 * t + l] != h[128 * t + 64 + l]) ++bad;
        printf("%s: %s", names[t], bad ? "MISMATCH:" : "ok\n");
        if (bad) { for (int l = 0; l < 64; ++l) printf(" %d", (int)h[128 * t + l] - 100); printf("\n"); }
    }
    return 0;
}
