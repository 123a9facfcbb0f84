// Do v_mfma_f64_16x16x4 and fp64 VALU (v_fma_f64, v_fmac_f64_dpp) instructions execute concurrently on a gfx950 SIMD?
// Three instruction streams timed with s_memtime on ONE work-group: MFMA only, VALU only, both interleaved — in one wave, and split over
// the two waves of a SIMD (work-group of 8 waves = 2 per SIMD: even waves MFMA, odd waves VALU).  If the interleaved time is the
// maximum of the two, the matrix pipe is a second fp64 resource next to the DP-ALU; if it is the sum, they share it.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_valu_f64 tools/ubench/mfma_valu_f64.hip ; run on the MI355X box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define TICK(t, var) asm volatile("s_nop 0\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(var) :: "memory")

constexpr int REP = 64;

// mode bit 0: MFMA stream (4 independent accumulators), bit 1: VALU stream (12 independent FMA chains per MFMA), bit 2: DPP-FMAs instead
template <int MODE>
__device__ __forceinline__ double body(double x, long long& dt) {
    d4 c0 = {x, x, x, x}, c1 = c0, c2 = c0, c3 = c0;
    double a[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) a[k] = x + k;
    double m = 1.0 + 1e-9 * x, w = x;
    long long t0, t1;
    TICK(t0, w);
#pragma unroll 1
    for (int it = 0; it < REP; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MODE & 1) {
                if (q == 0) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w, m, c0, 0, 0, 0);
                if (q == 1) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w, m, c1, 0, 0, 0);
                if (q == 2) c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(w, m, c2, 0, 0, 0);
                if (q == 3) c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(w, m, c3, 0, 0, 0);
            }
            if (MODE & 2) {
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    if (MODE & 4) asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[k]) : "v"(w), "v"(m));
                    else asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[k]) : "v"(w), "v"(m));
                }
            }
        }
    }
    double s = c0.x + c1.y + c2.z + c3.w;
#pragma unroll
    for (int k = 0; k < 12; ++k) s += a[k];
    TICK(t1, s);
    dt = t1 - t0;
    return s;
}

template <int MODE>
__global__ void k_one(double* out, long long* clk, double x0) {
    long long dt;
    const double s = body<MODE>(x0 + 1e-3 * threadIdx.x, dt);
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) clk[0] = dt;
}
// eight waves = two per SIMD; wave parity selects the stream
template <int MODE_EVEN, int MODE_ODD>
__global__ __launch_bounds__(512) void k_two(double* out, long long* clk, double x0) {
    const int wave = threadIdx.x >> 6;
    long long dt;
    double s;
    if ((wave >> 2) & 1) s = body<MODE_ODD>(x0 + 1e-3 * threadIdx.x, dt);     // waves 4..7 share SIMDs 0..3 with waves 0..3
    else s = body<MODE_EVEN>(x0 + 1e-3 * threadIdx.x, dt);
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[wave] = dt;
}

int main() {
    double* out; long long* clk;
    hipMalloc(&out, 8 * 512); hipMalloc(&clk, 8 * 8);
    long long h[8];
    auto show1 = [&](const char* name) {
        hipDeviceSynchronize(); hipMemcpy(h, clk, 8, hipMemcpyDeviceToHost);
        printf("%-46s %8lld ticks = %6.1f per MFMA slot (4 x %d slots)\n", name, h[0], (double)h[0] / (4 * REP), REP);
    };
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_one<1>, dim3(1), dim3(64), 0, 0, out, clk, 1.0); show1("one wave: MFMA only");
        hipLaunchKernelGGL(k_one<2>, dim3(1), dim3(64), 0, 0, out, clk, 1.0); show1("one wave: 12 v_fmac_f64 per slot only");
        hipLaunchKernelGGL(k_one<3>, dim3(1), dim3(64), 0, 0, out, clk, 1.0); show1("one wave: MFMA + 12 v_fmac_f64 per slot");
        hipLaunchKernelGGL(k_one<6>, dim3(1), dim3(64), 0, 0, out, clk, 1.0); show1("one wave: 12 v_fmac_f64_dpp per slot only");
        hipLaunchKernelGGL(k_one<7>, dim3(1), dim3(64), 0, 0, out, clk, 1.0); show1("one wave: MFMA + 12 v_fmac_f64_dpp per slot");
    }
    auto show2 = [&](const char* name) {
        hipDeviceSynchronize(); hipMemcpy(h, clk, 64, hipMemcpyDeviceToHost);
        printf("%-46s waves 0-3: %lld %lld %lld %lld | waves 4-7: %lld %lld %lld %lld ticks\n", name, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    };
    hipLaunchKernelGGL((k_two<1, 1>), dim3(1), dim3(512), 0, 0, out, clk, 1.0); show2("two waves per SIMD: MFMA | MFMA");
    hipLaunchKernelGGL((k_two<2, 2>), dim3(1), dim3(512), 0, 0, out, clk, 1.0); show2("two waves per SIMD: VALU | VALU");
    hipLaunchKernelGGL((k_two<1, 2>), dim3(1), dim3(512), 0, 0, out, clk, 1.0); show2("two waves per SIMD: MFMA | VALU");
    hipLaunchKernelGGL((k_two<1, 6>), dim3(1), dim3(512), 0, 0, out, clk, 1.0); show2("two waves per SIMD: MFMA | DPP-VALU");
    hipLaunchKernelGGL((k_two<3, 3>), dim3(1), dim3(512), 0, 0, out, clk, 1.0); show2("two waves per SIMD: both | both");
    return 0;
}
