// Latency micro-benchmarks on one wave (gfx950): dependent chains of the instructions the latency-bound kernels are made of.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/lat tools/ubench/lat.hip ; run on the MI355X box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 512
#define TICK(t, var) asm volatile("s_nop 0\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(var) :: "memory")
__global__ void k(double* out, long long* clk, double x0) {
    __shared__ double lds[1024];
    const int lane = threadIdx.x;
    lds[lane] = x0 + lane; lds[lane + 64] = 1.0;
    __syncthreads();
    double x = x0 + 1e-9 * lane, y = 1.0000001;
    long long t0, t1;
    // 1: dependent fma
    TICK(t0, x);
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_fma(x, y, 1e-9);
    TICK(t1, x); if (lane == 0) clk[0] = t1 - t0;
    // 2: dependent mul
    TICK(t0, x);
#pragma unroll
    for (int i = 0; i < N; ++i) x = x * y;
    TICK(t1, x); if (lane == 0) clk[1] = t1 - t0;
    // 3: dependent rsq
    TICK(t0, x);
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rsq(x) + 1.5;
    TICK(t1, x); if (lane == 0) clk[2] = t1 - t0;   // rsq + add
    // 4: dependent rcp
    TICK(t0, x);
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rcp(x) + 1.5;
    TICK(t1, x); if (lane == 0) clk[3] = t1 - t0;   // rcp + add
    // 5: dependent sqrt (IEEE, library expansion)
    TICK(t0, x);
#pragma unroll
    for (int i = 0; i < N; ++i) x = sqrt(x) + 1.5;
    TICK(t1, x); if (lane == 0) clk[4] = t1 - t0;
    // 6: dependent division
    TICK(t0, x);
#pragma unroll
    for (int i = 0; i < N; ++i) x = 1.5 / x + 1.5;
    TICK(t1, x); if (lane == 0) clk[5] = t1 - t0;
    // 7: dependent LDS round trip (address depends on the loaded value)
    int idx = lane;
    TICK(t0, idx);
#pragma unroll
    for (int i = 0; i < N; ++i) { const double v = lds[idx]; idx = ((int)v + i) & 63; }
    TICK(t1, idx); if (lane == 0) clk[6] = t1 - t0;
    // 8: readlane -> VALU dependent
    TICK(t0, x);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        int lo = __double2loint(x), hi = __double2hiint(x);
        lo = __builtin_amdgcn_readlane(lo, 3); hi = __builtin_amdgcn_readlane(hi, 3);
        x = __hiloint2double(hi, lo) * y + 1e-9 * lane;
    }
    TICK(t1, x); if (lane == 0) clk[7] = t1 - t0;   // 2 readlane + fma
    // 9: shfl_xor (DPP / bpermute) + add
    TICK(t0, x);
#pragma unroll
    for (int i = 0; i < N; ++i) x = x + __shfl_xor(x, 1 << (i % 6), 64);
    TICK(t1, x); if (lane == 0) clk[8] = t1 - t0;
    // 10: fp32 dependent fma
    float f = (float)x0;
    TICK(t0, f);
#pragma unroll
    for (int i = 0; i < N; ++i) f = __builtin_fmaf(f, 1.0000001f, 1e-9f);
    TICK(t1, f); if (lane == 0) clk[9] = t1 - t0;
    // 11: independent fma x4 (throughput of one wave)
    double a = x0, b = x0 + 1, c = x0 + 2, d = x0 + 3;
    TICK(t0, d);
#pragma unroll
    for (int i = 0; i < N; ++i) { a = __builtin_fma(a, y, 1e-9); b = __builtin_fma(b, y, 1e-9); c = __builtin_fma(c, y, 1e-9); d = __builtin_fma(d, y, 1e-9); }
    TICK(t1, d); if (lane == 0) clk[10] = t1 - t0;
    // 12: mfma f64 16x16x4 dependent
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 acc = {0, 0, 0, 0};
    double accx = acc[0]; TICK(t0, accx); acc[0] = accx;
#pragma unroll
    for (int i = 0; i < N; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[0], y, acc, 0, 0, 0);
    accx = acc[0]; TICK(t1, accx); acc[0] = accx; if (lane == 0) clk[11] = t1 - t0;
    // 13: s_memtime overhead
    TICK(t0, x); TICK(t1, x); if (lane == 0) clk[12] = t1 - t0;
    out[lane] = x + idx + f + a + b + c + d + acc[0];
}
int main() {
    double* out; long long* clk;
    hipMalloc(&out, 64 * 8); hipMalloc(&clk, 16 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, clk, 1.25);
    long long h[16];
    hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    const char* nm[] = {"fma f64", "mul f64", "rsq f64 + add", "rcp f64 + add", "sqrt() + add", "1.5/x + add", "LDS dependent read", "2 readlane + fma", "shfl_xor + add (f64)", "fma f32", "4 independent fma f64 (per group)", "mfma f64 16x16x4 dependent", "clock64 pair"};
    for (int i = 0; i < 13; ++i) printf("%-36s %8.1f ticks/iter\n", nm[i], i == 12 ? (double)h[i] : (double)h[i] / N);
    return 0;
}
