// mfma_chain -- v_mfma_i32_32x32x32_i8: cycles per instruction for 1, 2, 4 independent accumulators (dependency distance), at
// 1, 2, 3 wavefronts per SIMD.  The Hamming matcher issues two chains of 8 dependent MFMAs per tile.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_chain.hip -o mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int ITERS = 2048;
template <int NACC>
__global__ void __launch_bounds__(256) chain_k(int* out, int seed)
{
    v4i a = { seed, seed + 1, seed + 2, seed + 3 }, b = { seed * 3, seed * 5, seed * 7, seed * 11 };
    v16i acc[4];
    for (int k = 0; k < 4; k++) for (int i = 0; i < 16; i++) acc[k][i] = k + i;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int r = 0; r < 8 / NACC; r++)
#pragma unroll
            for (int k = 0; k < NACC; k++) acc[k] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[k], 0, 0, 0);
    }
    int r = 0;
    for (int k = 0; k < 4; k++) for (int i = 0; i < 16; i++) r ^= acc[k][i];
    if (r == 0x7fffffff) out[threadIdx.x] = r;
}
typedef void (*kern_t)(int*, int);
int main()
{
    int* out; CHECK(hipMalloc(&out, 4096));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct { const char* n; kern_t k; } cases[] = { {"1 accumulator", chain_k<1>}, {"2 accumulators", chain_k<2>}, {"4 accumulators", chain_k<4>} };
    std::printf("v_mfma_i32_32x32x32_i8: cycles per instruction per SIMD (2.4 GHz nominal), W workgroups of 256 per CU = W waves per SIMD\n%-18s", "waves/SIMD:");
    for (int w = 1; w <= 3; w++) std::printf(" %8d", w);
    std::printf("\n");
    for (auto& c : cases) {
        std::printf("%-18s", c.n);
        for (int w = 1; w <= 3; w++) {
            hipLaunchKernelGGL(c.k, dim3(cus * w), dim3(256), 0, 0, out, 3);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(c.k, dim3(cus * w), dim3(256), 0, 0, out, 3 + r);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::printf(" %8.1f", ms * 1e-3 / 5 * 2.4e9 / ((double)ITERS * 8 * w));
        }
        std::printf("\n");
    }
    return 0;
}
