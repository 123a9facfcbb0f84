// Sustained chip-wide fp64 matrix and vector rates of the MI355X box at hand (gfx950): every SIMD runs W waves of independent
// v_mfma_f64_16x16x4_f64 chains (or v_fma_f64 chains) for ~20 ms, HIP events around the launch.  The guide's peak — 78.6 TFLOP/s — is
// 256 CUs x 4 SIMDs x 32 flop / clock x 2.4 GHz; what the chip SUSTAINS under an fp64 load is that times (sustained clock / 2.4 GHz), and it is
// the ceiling the IMU role's matrix-core phase can be priced against.  Also: the 20-MFMA dependency pattern of one IMU block.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_f64_chip tools/ubench/mfma_f64_chip.hip ; run on the MI355X box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double __attribute__((ext_vector_type(4))) d4;

template <int WPS> __global__ __launch_bounds__(64, WPS) void k_mfma(int iters, double* out) {
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
        }
    }
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
// the MFMA pattern of one IMU block (k_linearize.hip imu_blocks): two whitening chains of 4, then g01 (a chain of 4 on the whitened operands),
// then g00 / g11 (two chains of 4; g00 starts from the previous block's g11)
template <int WPS> __global__ __launch_bounds__(64, WPS) void k_imu_pattern(int blocks, double* out) {
    d4 chain = {0, 0, 0, 0};
    double s = 1.0 + threadIdx.x * 1e-9, acc = 0.0;
    for (int b = 0; b < blocks; ++b) {
        d4 y0 = {0, 0, 0, 0}, y1 = y0;
#pragma unroll
        for (int c = 0; c < 4; ++c) { y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(s, s + c, y0, 0, 0, 0); y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(s, s - c, y1, 0, 0, 0); }
        d4 g00 = chain, g01 = {0, 0, 0, 0}, g11 = g01;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            g00 = __builtin_amdgcn_mfma_f64_16x16x4f64(y0[c], y0[c], g00, 0, 0, 0);
            g01 = __builtin_amdgcn_mfma_f64_16x16x4f64(y0[c], y1[c], g01, 0, 0, 0);
            g11 = __builtin_amdgcn_mfma_f64_16x16x4f64(y1[c], y1[c], g11, 0, 0, 0);
        }
        chain = g11;
        acc += g00[0] + g01[1];
        s = 1.0 + acc * 1e-300;
    }
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = acc + chain[2];
}
template <int WPS> __global__ __launch_bounds__(64, WPS) void k_fma(int iters, double* out) {
    double a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = threadIdx.x * 1e-9 + k;
    const double x = 1.0000001, y = 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = __builtin_fma(a[k], x, y);
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += a[k];
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = s;
}
template <class F> static float timed(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    double* out; hipMalloc(&out, sizeof(double) * 64 * 8192);
    const int simds = 256 * 4;
    for (int wps = 1; wps <= 2; ++wps) {
        const int waves = simds * wps, iters = 20000 / wps;
        const float ms = wps == 1 ? timed([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(waves), dim3(64), 0, 0, iters, out); }) : timed([&] { hipLaunchKernelGGL(k_mfma<2>, dim3(waves), dim3(64), 0, 0, iters, out); });
        const double fl = (double)waves * iters * 32 * 2048.0;
        printf("independent v_mfma_f64_16x16x4 chains, %d wave(s) per SIMD: %.2f ms, %.1f TFLOP/s = %.3f of 78.6 -> sustained clock %.2f GHz (64 cycles per MFMA)\n", wps, ms, fl / ms / 1e9, fl / ms / 1e9 / 78.6, fl / ms / 1e9 / 78.6 * 2.4);
    }
    for (int wps = 1; wps <= 2; ++wps) {
        const int waves = simds * wps, blocks = 16 * 600 / wps;
        const float ms = wps == 1 ? timed([&] { hipLaunchKernelGGL(k_imu_pattern<1>, dim3(waves), dim3(64), 0, 0, blocks, out); }) : timed([&] { hipLaunchKernelGGL(k_imu_pattern<2>, dim3(waves), dim3(64), 0, 0, blocks, out); });
        const double fl = (double)waves * blocks * 20 * 2048.0;
        printf("the 20-MFMA pattern of an IMU block, %d wave(s) per SIMD: %.2f ms, %.1f TFLOP/s = %.3f of 78.6; per block and SIMD %.0f ns\n", wps, ms, fl / ms / 1e9, fl / ms / 1e9 / 78.6, ms * 1e6 / ((double)blocks * wps));
    }
    for (int wps = 1; wps <= 2; ++wps) {
        const int waves = simds * wps, iters = 100000 / wps;
        const float ms = wps == 1 ? timed([&] { hipLaunchKernelGGL(k_fma<1>, dim3(waves), dim3(64), 0, 0, iters, out); }) : timed([&] { hipLaunchKernelGGL(k_fma<2>, dim3(waves), dim3(64), 0, 0, iters, out); });
        const double fl = (double)waves * iters * 64.0 * 64 * 2;
        printf("independent v_fma_f64 chains, %d wave(s) per SIMD: %.2f ms, %.1f TFLOP/s = %.3f of 78.6\n", wps, ms, fl / ms / 1e9, fl / ms / 1e9 / 78.6);
    }
    return 0;
}
