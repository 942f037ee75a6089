// depchain -- v_fma_f32 issue rate as a function of how many INDEPENDENT chains a wavefront interleaves (dependency distance),
// at 4 and 8 wavefronts per SIMD, no scalar instructions in the loop body.  Also: the same with the instruction forms the
// PnP scoring loop uses (v_fmac_f32, abs/neg modifiers, two compares and half a v_min3 per 19 arithmetic instructions).
// Build: hipcc -O3 --offload-arch=gfx950 depchain.hip -o depchain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)
constexpr int ITERS = 8192;
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define FMAC(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define FMAD(i, j) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[j]), "v"(b));
template <int C, int FORM>
__global__ void __launch_bounds__(256) chain_k(float* out, float seed)
{
    float a[16];
    for (int i = 0; i < 16; i++) a[i] = seed + threadIdx.x * 16 + i;
    float b = seed * 1.0001f, c = seed * 0.5f;
    for (int it = 0; it < ITERS; it++) {
        if constexpr (FORM == 0) {        // 16 v_fma_f32 per iteration over C chains
#pragma unroll
            for (int r = 0; r < 16 / C; r++) {
                if constexpr (C >= 1) FMA(0)
                if constexpr (C >= 2) FMA(1)
                if constexpr (C >= 4) { FMA(2) FMA(3) }
                if constexpr (C >= 8) { FMA(4) FMA(5) FMA(6) FMA(7) }
                if constexpr (C >= 16) { FMA(8) FMA(9) FMA(10) FMA(11) FMA(12) FMA(13) FMA(14) FMA(15) }
            }
        } else if constexpr (FORM == 1) { // the same with v_fmac_f32
#pragma unroll
            for (int r = 0; r < 16 / C; r++) {
                if constexpr (C >= 1) FMAC(0)
                if constexpr (C >= 2) FMAC(1)
                if constexpr (C >= 4) { FMAC(2) FMAC(3) }
                if constexpr (C >= 8) { FMAC(4) FMAC(5) FMAC(6) FMAC(7) }
                if constexpr (C >= 16) { FMAC(8) FMAC(9) FMAC(10) FMAC(11) FMAC(12) FMAC(13) FMAC(14) FMAC(15) }
            }
        } else if constexpr (FORM == 2) { // the shape of one point of the PnP loop: three chains of three, then A, B, e, s, m, two compares
            asm volatile(
                "v_fma_f32 %0, %8, %9, %10\n v_fma_f32 %1, %8, %9, %10\n v_fmac_f32 %0, %8, %10\n v_fma_f32 %2, %8, %9, %10\n"
                "v_fmac_f32 %1, %9, %10\n v_fmac_f32 %0, %9, %8\n v_fmac_f32 %2, %8, %10\n v_fmac_f32 %1, %10, %8\n v_fmac_f32 %2, %9, %8\n"
                "v_fmac_f32 %1, %9, %0\n v_mul_f32_e64 %3, %0, -%0\n v_fmac_f32 %2, %8, %0\n v_fmac_f32 %3, %1, %1\n v_fmac_f32 %3, %2, %2\n"
                "v_add_f32_e64 %2, |%2|, |%1|\n v_add_f32_e64 %2, |%0|, %2\n v_fma_f32 %1, %2, %9, %10\n v_fma_f32 %2, %2, %1, %8\n"
                "v_cmp_ngt_f32_e64 %4, |%3|, %2\n v_cmp_lt_f32_e64 %5, %3, -%2\n"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "=s"(*(unsigned long long*)&a[4]), "=s"(*(unsigned long long*)&a[6]), "+v"(a[8]), "+v"(a[9])
                : "v"(b), "v"(c), "v"(a[10]));
        }
    }
    float r = 0;
    for (int i = 0; i < 16; i++) r += a[i];
    if (r == 12345.678f) out[threadIdx.x] = r;
}
typedef void (*kern_t)(float*, float);
int main()
{
    float* out; CHECK(hipMalloc(&out, 4096));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct Case { const char* name; kern_t k; int per_iter; } cases[] = {
        {"v_fma_f32, 1 chain", chain_k<1, 0>, 16}, {"v_fma_f32, 2 chains", chain_k<2, 0>, 16}, {"v_fma_f32, 4 chains", chain_k<4, 0>, 16},
        {"v_fma_f32, 8 chains", chain_k<8, 0>, 16}, {"v_fma_f32, 16 chains", chain_k<16, 0>, 16},
        {"v_fmac_f32, 1 chain", chain_k<1, 1>, 16}, {"v_fmac_f32, 2 chains", chain_k<2, 1>, 16}, {"v_fmac_f32, 4 chains", chain_k<4, 1>, 16},
        {"v_fmac_f32, 16 chains", chain_k<16, 1>, 16}, {"PnP point shape (20)", chain_k<1, 2>, 20},
    };
    std::printf("cycles (at 2.4 GHz nominal) per vector instruction per SIMD; workgroups of 256, W per CU = W waves per SIMD\n%-26s", "waves/SIMD:");
    const int wl[] = {1, 2, 4, 8};
    for (int w : wl) std::printf(" %7d", w);
    std::printf("\n");
    for (auto& c : cases) {
        std::printf("%-26s", c.name);
        for (int w : wl) {
            hipLaunchKernelGGL(c.k, dim3(cus * w), dim3(256), 0, 0, out, 1.5f);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(c.k, dim3(cus * w), dim3(256), 0, 0, out, 1.5f + r);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::printf(" %7.2f", ms * 1e-3 / 5 * 2.4e9 / ((double)ITERS * c.per_iter * w));
        }
        std::printf("\n"); std::fflush(stdout);
    }
    return 0;
}
