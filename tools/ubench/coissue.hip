// coissue -- do two KERNELS on different streams share a SIMD's issue slots the way two wavefronts of one kernel do?
// Kernel A runs a chain of one opcode class, kernel B another; each is timed alone (whole chip, 4 waves per SIMD) and then both
// together on two streams.  together ~ max(alone) means the second kernel rides in issue slots the first leaves; together ~ sum
// means they share one pipe.  Build: hipcc -O3 --offload-arch=gfx950 coissue.hip -o coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
template <int KIND>
__global__ void __launch_bounds__(256) chain_k(unsigned* out, int iters, unsigned seed)
{
    unsigned a[16]; double d[8];
    for (int i = 0; i < 16; i++) a[i] = seed + threadIdx.x * 16 + i;
    for (int i = 0; i < 8; i++) d[i] = (double)(seed + threadIdx.x + i) * 1.0000001;
    unsigned b = seed ^ 0x12345u, c = seed | 3u;
    double B = (double)seed * 1.0000003, Cc = 0.9999999;
    for (int it = 0; it < iters; it++) {
        if constexpr (KIND == 0) {          // complex integer: v_max_u32
#define S(i) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP16(S)
#undef S
        } else if constexpr (KIND == 1) {   // simple f32: v_fma_f32
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP16(S)
#undef S
        } else if constexpr (KIND == 2) {   // f64
#define S(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i & 7]) : "v"(B), "v"(Cc));
            REP16(S)
#undef S
        } else if constexpr (KIND == 3) {   // simple integer: v_add_u32
#define S(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP16(S)
#undef S
        } else if constexpr (KIND == 4) {   // SDWA byte op (FAST phase 1's class)
#define S(i) asm volatile("v_max_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2" : "+v"(a[i]) : "v"(b));
            REP16(S)
#undef S
        } else if constexpr (KIND == 5) {   // packed 16-bit (FAST phase 2's class)
#define S(i) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP16(S)
#undef S
        } else if constexpr (KIND == 6) {   // compare to a lane mask
#define S(i) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
            REP16(S)
#undef S
        }
    }
    unsigned r = 0;
    for (int i = 0; i < 16; i++) r ^= a[i];
    for (int i = 0; i < 8; i++) r ^= (unsigned)d[i];
    if (r == 0x7fffffffu) out[threadIdx.x] = r;
}
typedef void (*kern_t)(unsigned*, int, unsigned);
static const char* names[] = { "v_max_u32", "v_fma_f32", "v_fma_f64", "v_add_u32", "v_max_u32_sdwa", "v_pk_min_i16", "v_cmp_gt_u32" };
static kern_t kerns[] = { chain_k<0>, chain_k<1>, chain_k<2>, chain_k<3>, chain_k<4>, chain_k<5>, chain_k<6> };

int main()
{
    unsigned* out; CHECK(hipMalloc(&out, 4096));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    hipStream_t s0, s1; CHECK(hipStreamCreate(&s0)); CHECK(hipStreamCreate(&s1));
    hipEvent_t e0, e1, f1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&f1));
    const int iters = 20000, blocks = cus * 4;   // 4 workgroups of 256 per CU = 4 waves per SIMD per kernel (8 when two kernels run)
    auto run = [&](int ka, int kb) {
        // kb < 0: ka alone.  Both launches are enqueued before either can finish (20000 x 16 instructions per wave).
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, s0));
        CHECK(hipStreamWaitEvent(s1, e0, 0));
        hipLaunchKernelGGL(kerns[ka], dim3(blocks), dim3(256), 0, s0, out, iters, 7u);
        if (kb >= 0) hipLaunchKernelGGL(kerns[kb], dim3(blocks), dim3(256), 0, s1, out, iters, 9u);
        CHECK(hipEventRecord(f1, s1));
        CHECK(hipStreamWaitEvent(s0, f1, 0));
        CHECK(hipEventRecord(e1, s0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        return ms;
    };
    for (int k = 0; k < 7; k++) run(k, -1);   // warm-up / code load
    float alone[7];
    for (int k = 0; k < 7; k++) { alone[k] = run(k, -1); std::printf("alone %-16s %8.3f ms\n", names[k], alone[k]); }
    const int pairs[][2] = { {0, 1}, {0, 3}, {4, 1}, {5, 1}, {6, 1}, {2, 1}, {2, 0}, {2, 3}, {4, 3}, {5, 3}, {0, 0}, {1, 1}, {2, 2}, {1, 3}, {4, 2}, {5, 2} };
    for (auto& pr : pairs) {
        const float t = run(pr[0], pr[1]);
        std::printf("together %-16s + %-16s %8.3f ms   (sum %8.3f, max %8.3f)\n", names[pr[0]], names[pr[1]], t, alone[pr[0]] + alone[pr[1]],
                    alone[pr[0]] > alone[pr[1]] ? alone[pr[0]] : alone[pr[1]]);
        std::fflush(stdout);
    }
    return 0;
}
