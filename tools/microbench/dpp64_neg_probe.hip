// dpp64_neg_probe.hip -- checks on gfx950 (a) that the NEG field of v_mfma_f64_4x4x4_4b_f64 (blgp bit 0) negates the A operand exactly,
// (b) that v_mov_b64_dpp row_newbcast:N under a bank mask is the two-dword broadcast of lane N of each 16-lane row.
// hipcc --offload-arch=gfx950 -O2 -o dpp64_neg_probe dpp64_neg_probe.hip && ./dpp64_neg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
template <int T>
__device__ double bcast_row_old(double v) {
    constexpr int q = T & 3, QP = q | (q << 2) | (q << 4) | (q << 6);
    constexpr int BM = (T >> 2) ? 0x5 : 0xA;
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, QP, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, QP, 0xF, 0xF, true);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xF, BM, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xF, BM, false);
    return __hiloint2double(hi, lo);
}
template <int T>
__device__ double bcast_row_new(double v) {
    long x = __double_as_longlong(v), y;
    y = __builtin_amdgcn_update_dpp(x, x, 0x150 + T, 0xF, 0x3, false);
    y = __builtin_amdgcn_update_dpp(y, x, 0x158 + T, 0xF, 0xC, false);
    return __longlong_as_double(y);
}
template <int T>
__device__ void one(const double *in, double *out, int &o) {
    const double v = in[threadIdx.x];
    out[o + threadIdx.x] = bcast_row_old<T>(v); out[o + 64 + threadIdx.x] = bcast_row_new<T>(v); o += 128;
}
__global__ void probe(const double *in, double *out) {
    int o = 0;
    one<0>(in, out, o); one<1>(in, out, o); one<2>(in, out, o); one<3>(in, out, o);
    one<4>(in, out, o); one<5>(in, out, o); one<6>(in, out, o); one<7>(in, out, o);
    const double a = in[64 + threadIdx.x], b = in[128 + threadIdx.x], c = in[192 + threadIdx.x];
    out[o + threadIdx.x] = __builtin_amdgcn_mfma_f64_4x4x4f64(-a, b, c, 0, 0, 0);
    out[o + 64 + threadIdx.x] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 1);
    out[o + 128 + threadIdx.x] = -__builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[o + 192 + threadIdx.x] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 1);
}
// (c) accuracy of v_rsq_f64 and of the refinements behind it: two Newton steps (the shipped inv_sqrt) against one third-order step
__global__ void rsq_probe(const double *x, double *y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = x[i];
    const double y0 = __builtin_amdgcn_rsq(d);
    double yn = y0 * (1.5 - 0.5 * d * y0 * y0); yn = yn * (1.5 - 0.5 * d * yn * yn);
    const double h = __builtin_fma(-(d * y0), y0, 1.0);                         // 1 - d y0^2
    const double yc = __builtin_fma(y0, h * __builtin_fma(0.375, h, 0.5), y0);  // y0 (1 + h/2 + 3 h^2 / 8)
    y[i] = y0; y[n + i] = yn; y[2 * n + i] = yc;
}
int main() {
    {
        const int n = 1 << 20;
        std::vector<double> hx(n), hy(3 * n);
        srand(11);
        for (int i = 0; i < n; ++i) hx[i] = exp2(-14.0 + 28.0 * (rand() / (double)RAND_MAX)) * (1.0 + rand() / (double)RAND_MAX * 1e-3);
        double *dx, *dy;
        hipMalloc(&dx, n * 8); hipMalloc(&dy, 3 * n * 8);
        hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice);
        rsq_probe<<<n / 256, 256>>>(dx, dy, n);
        hipMemcpy(hy.data(), dy, 3 * n * 8, hipMemcpyDeviceToHost);
        long double e0 = 0, en = 0, ec = 0; int same = 0;
        for (int i = 0; i < n; ++i) {
            const long double t = 1.0L / sqrtl((long double)hx[i]);
            e0 = fmaxl(e0, fabsl((hy[i] - t) / t)); en = fmaxl(en, fabsl((hy[n + i] - t) / t)); ec = fmaxl(ec, fabsl((hy[2 * n + i] - t) / t));
            same += hy[n + i] == hy[2 * n + i];
        }
        printf("v_rsq_f64 max relative error 2^%.1f; + two Newton steps 2^%.2f; + one third-order step 2^%.2f; the two agree bit for bit on %d of %d\n",
               (double)log2l(e0), (double)log2l(en), (double)log2l(ec), same, n);
    }
    std::vector<double> h(256), r(8 * 128 + 256);
    srand(7);
    for (auto &x : h) x = (rand() / (double)RAND_MAX - 0.5) * exp2((rand() % 40) - 20);
    double *din, *dout;
    hipMalloc(&din, h.size() * 8); hipMalloc(&dout, r.size() * 8);
    hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(din, dout);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
    hipMemcpy(r.data(), dout, r.size() * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 8; ++t) for (int l = 0; l < 64; ++l) {
        const double expect = h[(l & ~7) + t];
        if (memcmp(&r[t * 128 + l], &expect, 8) || memcmp(&r[t * 128 + 64 + l], &expect, 8)) ++bad;
    }
    printf("bcast_row old / new against lane T of the 8-lane group: %d mismatches of 1024\n", bad);
    int badn = 0;
    for (int l = 0; l < 64; ++l) { if (memcmp(&r[1024 + l], &r[1024 + 64 + l], 8)) ++badn; if (memcmp(&r[1024 + 128 + l], &r[1024 + 192 + l], 8)) ++badn; }
    printf("mfma neg:[1,0,0] against a negated A operand / a negated result: %d mismatches of 128\n", badn);
    return bad || badn;
}
