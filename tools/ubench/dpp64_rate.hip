// Issue rate of v_fmac_f64_dpp row_newbcast under different operand patterns (gfx950): does the half rate seen in dpp64.hip (9.2 ticks
// against 4.8 for v_fma_f64) depend on the broadcast lane changing from one instruction to the next, on the source register, or on the
// accumulator?  15 accumulators x 32 rounds each; s_memtime ticks per instruction.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/dpp64_rate tools/ubench/dpp64_rate.hip ; run on the MI355X box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define TICK(t, var) asm volatile("s_nop 0\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(var) :: "memory")
template <int L> __device__ __forceinline__ void fnma_bc(double& acc, double src, double mul) {
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(L));
}
template <int B_, int E_, class F> __device__ __forceinline__ void sfor(F&& f) {
    if constexpr (B_ < E_) { f(std::integral_constant<int, B_>{}); sfor<B_ + 1, E_>(f); }
}
// PAT 0: lane changes every instruction, one source; 1: same lane, one source; 2: same lane, 15 sources (src = s[k]);
// 3: lane changes, 15 sources; 4: plain v_fmac_f64; 5: same lane, source = another accumulator set (as in the elimination)
template <int PAT>
__global__ void k_rate(double* out, long long* clk, double x0) {
    const int lane = threadIdx.x;
    double m = 1.0 + 1e-9 * lane;
    double a[15], s[15];
#pragma unroll
    for (int r = 0; r < 15; ++r) { a[r] = x0 + r; s[r] = x0 + 1e-3 * (lane + r); }
    double w = s[0];
    long long t0, t1;
    TICK(t0, w);
#pragma unroll 1
    for (int it = 0; it < 32; ++it) {
        sfor<0, 15>([&](auto K) {
            constexpr int k = std::remove_reference_t<decltype(K)>::value;
            if constexpr (PAT == 0) fnma_bc<k>(a[k], w, m);
            if constexpr (PAT == 1) fnma_bc<7>(a[k], w, m);
            if constexpr (PAT == 2) fnma_bc<7>(a[k], s[k], m);
            if constexpr (PAT == 3) fnma_bc<k>(a[k], s[k], m);
            if constexpr (PAT == 4) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[k]) : "v"(s[k]), "v"(m));
            if constexpr (PAT == 5) fnma_bc<7>(a[k], s[k], s[(k + 1) % 15]);
        });
    }
    double q = 0.0;
#pragma unroll
    for (int r = 0; r < 15; ++r) q += a[r];
    TICK(t1, q);
    out[lane] = q;
    if (lane == 0) clk[0] = t1 - t0;
}
int main() {
    double* out; long long* clk; long long h;
    (void)hipMalloc(&out, 8 * 64); (void)hipMalloc(&clk, 8);
    const char* names[6] = {"lane changes, one source", "same lane, one source", "same lane, 15 sources", "lane changes, 15 sources", "plain v_fmac_f64", "same lane, 15 sources, 15 multipliers"};
    for (int rep = 0; rep < 2; ++rep)
        for (int p = 0; p < 6; ++p) {
            if (p == 0) hipLaunchKernelGGL(k_rate<0>, dim3(1), dim3(64), 0, 0, out, clk, 1.0);
            if (p == 1) hipLaunchKernelGGL(k_rate<1>, dim3(1), dim3(64), 0, 0, out, clk, 1.0);
            if (p == 2) hipLaunchKernelGGL(k_rate<2>, dim3(1), dim3(64), 0, 0, out, clk, 1.0);
            if (p == 3) hipLaunchKernelGGL(k_rate<3>, dim3(1), dim3(64), 0, 0, out, clk, 1.0);
            if (p == 4) hipLaunchKernelGGL(k_rate<4>, dim3(1), dim3(64), 0, 0, out, clk, 1.0);
            if (p == 5) hipLaunchKernelGGL(k_rate<5>, dim3(1), dim3(64), 0, 0, out, clk, 1.0);
            (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
            printf("%-42s %7lld ticks = %5.2f per instruction\n", names[p], h, (double)h / (32 * 15));
        }
    return 0;
}
