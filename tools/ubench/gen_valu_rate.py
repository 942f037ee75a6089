#!/usr/bin/env python3
"""Generates valu_rate.hip: a per-opcode issue-rate probe for gfx950.

For every entry of OPS a kernel runs ITERS iterations of a block of N_CHAINS-way independent instructions; every wavefront stamps
s_memtime before and after its loop, and the host reports TRUE shader cycles per wave64 instruction per SIMD
(max end - min start over the waves of one workgroup, / instructions issued on one SIMD), for 1, 2, 4 and 8 waves per SIMD.
A "mix" entry interleaves several opcodes in one block (what a real kernel's inner loop looks like).

usage: python3 gen_valu_rate.py > valu_rate.hip && hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
"""
import sys

# {a}: 32-bit in/out accumulator of the chain, {b}/{c}: 32-bit loop-invariant inputs, {A}/{B}/{C}: the same as 64-bit pairs
S = {
    "add_u32": "v_add_u32 {a}, {a}, {b}", "sub_u32": "v_sub_u32 {a}, {a}, {b}", "and_b32": "v_and_b32 {a}, {a}, {b}",
    "or_b32": "v_or_b32 {a}, {a}, {b}", "xor_b32": "v_xor_b32 {a}, {a}, {b}", "lshl_b32": "v_lshlrev_b32 {a}, 3, {a}",
    "lshr_b32": "v_lshrrev_b32 {a}, 3, {a}", "mov_b32": "v_mov_b32 {a}, {b}", "max_u32": "v_max_u32 {a}, {a}, {b}",
    "min_u32": "v_min_u32 {a}, {a}, {b}", "max_i32": "v_max_i32 {a}, {a}, {b}",
    "add_f32": "v_add_f32 {a}, {a}, {b}", "mul_f32": "v_mul_f32 {a}, {a}, {b}", "fma_f32": "v_fma_f32 {a}, {a}, {b}, {c}",
    "max_f32": "v_max_f32 {a}, {a}, {b}",
    "add_u16": "v_add_u16 {a}, {a}, {b}", "sub_u16": "v_sub_u16 {a}, {a}, {b}", "max_u16": "v_max_u16 {a}, {a}, {b}",
    "min_u16": "v_min_u16 {a}, {a}, {b}", "max_i16": "v_max_i16 {a}, {a}, {b}", "min_i16": "v_min_i16 {a}, {a}, {b}",
    "mul_lo_u16": "v_mul_lo_u16 {a}, {a}, {b}", "lshl_b16": "v_lshlrev_b16 {a}, 3, {a}", "lshr_b16": "v_lshrrev_b16 {a}, 3, {a}",
    "mad_u16": "v_mad_u16 {a}, {a}, {b}, {c}", "add_f16": "v_add_f16 {a}, {a}, {b}", "max_f16": "v_max_f16 {a}, {a}, {b}",
    "fma_f16": "v_fma_f16 {a}, {a}, {b}, {c}",
    "max_u16_sdwa": "v_max_u16_sdwa {a}, {a}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2",
    "add_u16_sdwa": "v_add_u16_sdwa {a}, {a}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2",
    "add_f32_sdwa": "v_add_f32_sdwa {a}, {a}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD",
    "max_u32_sdwa": "v_max_u32_sdwa {a}, {a}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2",
    "add_u32_sdwa": "v_add_u32_sdwa {a}, {a}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2",
    "cmp_gt_u32": "v_cmp_gt_u32 vcc, {a}, {b}", "cmp_gt_u16": "v_cmp_gt_u16 vcc, {a}, {b}", "cmp_gt_f32": "v_cmp_gt_f32 vcc, {a}, {b}",
    "cmp_gt_u32_s": "v_cmp_gt_u32 s[20:21], {a}, {b}",
    "cndmask": "v_cndmask_b32 {a}, {a}, {b}, vcc",
    "pk_max_u16": "v_pk_max_u16 {a}, {a}, {b}", "pk_add_u16": "v_pk_add_u16 {a}, {a}, {b}", "pk_min_i16": "v_pk_min_i16 {a}, {a}, {b}",
    "pk_add_f16": "v_pk_add_f16 {a}, {a}, {b}", "pk_max_f16": "v_pk_max_f16 {a}, {a}, {b}",
    "perm": "v_perm_b32 {a}, {a}, {b}, {c}", "alignbyte": "v_alignbyte_b32 {a}, {a}, {b}, 3", "bfe": "v_bfe_u32 {a}, {a}, 8, 8",
    "and_or": "v_and_or_b32 {a}, {a}, {b}, {c}", "lshl_or": "v_lshl_or_b32 {a}, {a}, 3, {b}", "add3": "v_add3_u32 {a}, {a}, {b}, {c}",
    "max3_u32": "v_max3_u32 {a}, {a}, {b}, {c}", "med3_u32": "v_med3_u32 {a}, {a}, {b}, {c}", "mad_u24": "v_mad_u32_u24 {a}, {a}, {b}, {c}",
    "dot4_u8": "v_dot4_u32_u8 {a}, {a}, {b}, {c}", "sad_u8": "v_sad_u8 {a}, {a}, {b}, {c}", "bcnt": "v_bcnt_u32_b32 {a}, {b}, {a}",
    "mul_lo_u32": "v_mul_lo_u32 {a}, {a}, {b}", "cvt_f32_u32": "v_cvt_f32_u32 {a}, {a}", "cvt_f32_ub0": "v_cvt_f32_ubyte0 {a}, {a}",
    "cvt_u32_f32": "v_cvt_u32_f32 {a}, {a}", "rcp_f32": "v_rcp_f32 {a}, {a}",
    "fma_f64": "v_fma_f64 {A}, {A}, {B}, {C}", "add_f64": "v_add_f64 {A}, {A}, {B}", "mul_f64": "v_mul_f64 {A}, {A}, {B}",
    "pk_fma_f32": "v_pk_fma_f32 {A}, {A}, {B}, {C}", "pk_add_f32": "v_pk_add_f32 {A}, {A}, {B}", "pk_mul_f32": "v_pk_mul_f32 {A}, {A}, {B}",
    "cmp_gt_f64": "v_cmp_gt_f64 vcc, {A}, {B}", "rcp_f64": "v_rcp_f64 {A}, {A}", "max_f64": "v_max_f64 {A}, {A}, {B}",
    "s_and_b64": "s_and_b64 s[20:21], s[20:21], vcc", "s_nop": "s_nop 0",
    "ds_read_b32": None,
}

# (label, [opcode keys ...] cycled over the 16 slots of a block)
OPS = [(k, [k]) for k in S if S[k] is not None and not k.startswith("s_")]
# every opcode alternating with a "simple" one: does it share the SIMD with it (pair ~ 4.5 cycles) or not (pair ~ 6.75 / 9)?
OPS += [(k + " / add_u32", [k, "add_u32"]) for k in S if S[k] is not None and not k.startswith("s_") and k != "add_u32"]
OPS += [
    ("mix add_f32 / max_u32", ["add_f32", "max_u32"]),
    ("mix add_f32 / fma_f64", ["add_f32", "fma_f64"]),
    ("mix max_u16 / cmp_gt_u32", ["max_u16", "cmp_gt_u32"]),
    ("mix max_u16 / max_u32_sdwa", ["max_u16", "max_u32_sdwa"]),
    ("mix add_u32 / and_b32 / lshr", ["add_u32", "and_b32", "lshr_b32", "xor_b32"]),
    ("mix cmp_gt_u32 / s_and_b64", ["cmp_gt_u32", "s_and_b64"]),
    ("mix max_u32 / s_nop", ["max_u32", "s_nop"]),
    ("mix fma_f64 / s_nop", ["fma_f64", "s_nop"]),
    ("mix fma_f64 / mov_b32", ["fma_f64", "mov_b32"]),
    ("mix fma_f64 / add_u32", ["fma_f64", "add_u32"]),
    ("mix fma_f64 / max_u32", ["fma_f64", "max_u32"]),
]
N = 16


def block(keys):
    out, nv = [], 0
    for i in range(N):
        k = keys[i % len(keys)]
        t = S[k]
        if k.startswith("s_"):
            out.append('            asm volatile("%s" ::: "vcc", "scc", "s20", "s21");' % t)
            continue
        nv += 1
        if "{A}" in t:
            s = t.replace("{A}", "%0").replace("{B}", "%1").replace("{C}", "%2")
            out.append('            asm volatile("%s" : "+v"(d[%d]) : "v"(B), "v"(Cc) : "vcc", "scc", "s20", "s21");' % (s, i % 8))
        else:
            s = t.replace("{a}", "%0").replace("{b}", "%1").replace("{c}", "%2")
            out.append('            asm volatile("%s" : "+v"(a[%d]) : "v"(b), "v"(c) : "vcc", "scc", "s20", "s21");' % (s, i))
    return "\n".join(out), nv


def main():
    w = sys.stdout.write
    w('''// valu_rate.hip -- GENERATED by tools/ubench/gen_valu_rate.py (edit that, not this).
// True shader cycles per wave64 vector instruction per SIMD on gfx950, per opcode and for a few mixes, at 1/2/4/8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)
constexpr int ITERS = 1024;
struct Stamp { unsigned long long t0, t1, r0, r1; };
template <int KIND>
__global__ void __launch_bounds__(1024) rate_k(Stamp* st, unsigned* out, unsigned seed)
{
    unsigned a[16];
    double d[8];
    for (int i = 0; i < 16; i++) a[i] = seed + threadIdx.x * 16 + i;
    for (int i = 0; i < 8; i++) d[i] = (double)(seed + threadIdx.x + i) * 1.0000001;
    unsigned b = seed ^ 0x12345u, c = seed | 3u;
    double B = (double)seed * 1.0000003, Cc = 0.9999999;
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
''')
    counts = []
    for k, (label, keys) in enumerate(OPS):
        body, nv = block(keys)
        counts.append(nv)
        w("        if constexpr (KIND == %d) {\n%s\n        }\n" % (k, body))
    w('''    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) { Stamp s = { t0, t1, r0, r1 }; st[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = s; }
    unsigned r = 0;
    for (int i = 0; i < 16; i++) r ^= a[i];
    for (int i = 0; i < 8; i++) r ^= (unsigned)d[i];
    if (r == 0x7fffffffu) out[threadIdx.x] = r;
}
typedef void (*kern_t)(Stamp*, unsigned*, unsigned);
struct Case { const char* name; kern_t k; int slots; };
int main(int argc, char** argv)
{
    unsigned* out; CHECK(hipMalloc(&out, 4096));
    Stamp* st; CHECK(hipMalloc(&st, sizeof(Stamp) * 16 * 1024));
    std::vector<Stamp> h(16 * 1024);
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    std::vector<Case> cases = {
''')
    for k, (label, keys) in enumerate(OPS):
        w('        {"%s", rate_k<%d>, %d},\n' % (label, k, N))
    w('''    };
    const int wps_list[] = {1, 2, 4, 8};
    std::printf("true shader cycles per instruction SLOT per SIMD (16 slots per iteration, all CUs busy); last column: clock in MHz\\n");
    std::printf("%-28s", "waves/SIMD:");
    for (int w : wps_list) std::printf("  %7d", w);
    std::printf("\\n");
    for (int r = 0; r < 50; r++) hipLaunchKernelGGL(cases[0].k, dim3(cus), dim3(1024), 0, 0, st, out, 1u);
    CHECK(hipDeviceSynchronize());
    const int first = argc > 1 ? std::atoi(argv[1]) : 0;
    for (size_t ci = (size_t)first; ci < cases.size(); ci++) {
        const Case& c = cases[ci];
        const clock_t c0 = clock();
        std::printf("%-28s", c.name);
        double mhz = 0;
        for (int wps : wps_list) {
            const int threads = wps >= 4 ? 1024 : 256 * wps;
            const int per_cu = wps >= 8 ? 2 : 1;
            const int blocks = cus * per_cu;
            for (int r = 0; r < 3; r++) hipLaunchKernelGGL(c.k, dim3(blocks), dim3(threads), 0, 0, st, out, 12345u + r);
            CHECK(hipDeviceSynchronize());
            const int wpb = threads / 64;
            CHECK(hipMemcpy(h.data(), st, sizeof(Stamp) * blocks * wpb, hipMemcpyDeviceToHost));
            std::vector<double> cyc;
            for (int bl = 0; bl < blocks; bl++) {
                unsigned long long lo = ~0ull, hi = 0, rlo = ~0ull, rhi = 0;
                for (int wv = 0; wv < wpb; wv++) { const Stamp& s = h[bl * wpb + wv]; lo = std::min(lo, s.t0); hi = std::max(hi, s.t1); rlo = std::min(rlo, s.r0); rhi = std::max(rhi, s.r1); }
                cyc.push_back((double)(hi - lo));
                if (bl == 0) mhz = (double)(hi - lo) / (double)(rhi - rlo) * 100.0;
            }
            std::sort(cyc.begin(), cyc.end());
            // waves on one SIMD while this workgroup ran: wps (two co-resident workgroups at 8 overlap almost completely)
            std::printf("  %7.2f", cyc[cyc.size() / 2] / ((double)ITERS * c.slots * wps));
        }
        std::printf("  %6.0f  [%.2f s]\\n", mhz, (double)(clock() - c0) / CLOCKS_PER_SEC);
        std::fflush(stdout);
    }
    return 0;
}
''')


if __name__ == "__main__":
    main()
