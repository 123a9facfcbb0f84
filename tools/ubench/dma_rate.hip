// How fast does a CU stage global memory into LDS (gfx950)?  One work-group of W waves (1 .. 8, one CU), each wave stages N pieces of 1 KiB
// (64 lanes x 16 bytes) per round from a buffer that is either small (cache-resident: 64 KiB per wave, re-read every round) or streamed
// (distinct addresses, HBM), by  (a) global_load_lds_dwordx4 (LDS-DMA),  (b) global_load_dwordx4 into registers + ds_write_b128,
// (c) global_load_lds_dword (4-byte pieces).  s_memtime cycles per 1-KiB piece per wave and bytes per clock per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/dma_rate tools/ubench/dma_rate.hip ; run on the MI355X box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_t;
typedef double __attribute__((ext_vector_type(2))) dbl2;
constexpr int NP = 16, ROUNDS = 64;
template <int MODE>
__global__ __launch_bounds__(512) void k_dma(const double* src, long stride_round, double* out, long long* clk) {
    __shared__ double S[8 * NP * 128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* my = S + wave * NP * 128;
    const double* g = src + (size_t)wave * NP * 128 * ROUNDS + lane * 2;
    double acc = 0.0;
    __syncthreads();
    const long long t0 = clock64();
    for (int r = 0; r < ROUNDS; ++r) {
        const double* gr = g + (size_t)r * stride_round;
        if constexpr (MODE == 0) {
#pragma unroll
            for (int p = 0; p < NP; ++p) __builtin_amdgcn_global_load_lds(gr + p * 128, (lds_t)(my + p * 128), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if constexpr (MODE == 1) {
            dbl2 v[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) v[p] = *reinterpret_cast<const dbl2*>(gr + p * 128);
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<dbl2*>(my + p * 128 + lane * 2) = v[p];
        } else {
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(gr + p * 128) - lane * 3 + q * 64, (lds_t)(reinterpret_cast<float*>(my + p * 128) + q * 64), 4, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        acc += my[(lane * 7 + r) % (NP * 128)];
    }
    const long long t1 = clock64();
    out[threadIdx.x] = acc;
    if (lane == 0) clk[wave] = t1 - t0;
}
int main() {
    const size_t doubles = (size_t)8 * NP * 128 * ROUNDS;      // 12.6 MB: one distinct KiB per piece, wave and round
    double* src; double* out; long long* clk; long long h[8];
    (void)hipMalloc(&src, doubles * 8); (void)hipMemset(src, 0, doubles * 8);
    (void)hipMalloc(&out, 8 * 512); (void)hipMalloc(&clk, 64);
    const char* names[3] = {"global_load_lds_dwordx4", "global_load_dwordx4 + ds_write_b128", "global_load_lds_dword x4"};
    for (int mode = 0; mode < 3; ++mode)
        for (int streamed = 0; streamed < 2; ++streamed)
            for (int W : {1, 2, 4, 8}) {
                const long stride = streamed ? (long)NP * 128 : 0;
                for (int rep = 0; rep < 2; ++rep) {
                    if (mode == 0) hipLaunchKernelGGL(k_dma<0>, dim3(1), dim3(64 * W), 0, 0, src, stride, out, clk);
                    if (mode == 1) hipLaunchKernelGGL(k_dma<1>, dim3(1), dim3(64 * W), 0, 0, src, stride, out, clk);
                    if (mode == 2) hipLaunchKernelGGL(k_dma<2>, dim3(1), dim3(64 * W), 0, 0, src, stride, out, clk);
                    (void)hipDeviceSynchronize();
                }
                (void)hipMemcpy(h, clk, 64, hipMemcpyDeviceToHost);
                long long mx = 0;
                for (int w = 0; w < W; ++w) mx = h[w] > mx ? h[w] : mx;
                printf("%-38s %-9s %d wave(s): %7.1f cycles per 1-KiB piece per wave, %6.2f B/clk on the CU\n", names[mode], streamed ? "streamed" : "cached", W,
                       (double)mx / (ROUNDS * NP), (double)W * ROUNDS * NP * 1024 / (double)mx);
            }
    return 0;
}
