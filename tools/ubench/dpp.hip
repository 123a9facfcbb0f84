// Semantics + latency check of the DPP helpers used by the eigen-solver / wave reductions (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int ROWMASK = 0xF>
__device__ __forceinline__ double dpp64(double v) {   // lanes without a valid source (or outside ROWMASK) read 0
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWMASK, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWMASK, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sum(double v) {   // every lane: sum over its row of 16 lanes
    v += dpp64<0xB1>(v); v += dpp64<0x4E>(v); v += dpp64<0x141>(v); v += dpp64<0x140>(v);
    return v;
}
__device__ __forceinline__ double rdl(double v, int l) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_readlane(hi, l), __builtin_amdgcn_readlane(lo, l));
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v = row_sum(v);
    v += dpp64<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
    v += dpp64<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3
    return rdl(v, 63);
}
#define TICK(t, var) asm volatile("s_nop 0\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(var) :: "memory")
__global__ void k(double* out, long long* clk) {
    const int lane = threadIdx.x;
    double v = 1.0 + lane;                       // lane values 1..64
    out[lane] = row_sum(v);                      // rows: 136, 392, 648, 904
    out[64 + lane] = wave_sum_dpp(v);            // 2080
    out[128 + lane] = dpp64<0x101>(v);           // row_shl:1 -> lane i reads lane i+1 (0 at the row end)
    out[192 + lane] = dpp64<0x111>(v);           // row_shr:1 -> lane i reads lane i-1
    long long t0, t1;
    double x = v;
    TICK(t0, x);
#pragma unroll
    for (int i = 0; i < 64; ++i) x = row_sum(x) * 0.0625;
    TICK(t1, x);
    if (lane == 0) clk[0] = (t1 - t0) / 64;
    TICK(t0, x);
#pragma unroll
    for (int i = 0; i < 64; ++i) x = wave_sum_dpp(x) * (1.0 / 64) + lane;
    TICK(t1, x);
    if (lane == 0) clk[1] = (t1 - t0) / 64;
    out[256 + lane] = x;
}
int main() {
    double* out; long long* clk;
    (void)hipMalloc(&out, 320 * 8); (void)hipMalloc(&clk, 16 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, clk);
    double h[320]; long long c[2];
    (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost); (void)hipMemcpy(c, clk, sizeof c, hipMemcpyDeviceToHost);
    printf("row_sum lanes 0,15,16,63: %g %g %g %g (expect 136 136 392 904)\n", h[0], h[15], h[16], h[63]);
    printf("wave_sum lanes 0,63: %g %g (expect 2080)\n", h[64], h[127]);
    printf("row_shl:1 lanes 0,14,15,16: %g %g %g %g (expect 2 16 0 18)\n", h[128], h[142], h[143], h[144]);
    printf("row_shr:1 lanes 0,1,15,16: %g %g %g %g (expect 0 1 15 0)\n", h[192], h[193], h[207], h[208]);
    printf("row_sum %lld ticks, wave_sum_dpp %lld ticks\n", c[0], c[1]);
    return 0;
}
