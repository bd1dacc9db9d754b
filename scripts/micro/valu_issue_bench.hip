// Microbenchmark: what a wave64 VALU instruction costs on gfx950 (MI355X) -- issue cycles per SIMD for independent streams,
// latency of dependent chains, and the VALU -> SGPR -> branch round trip the composite kernels' visit contains
// (v_cmp -> s_and -> s_cbranch).  Settles the constant DESIGN.md 5 used for `valu_issue_frac` (round 2 assumed 4 cycles per
// wave64 instruction; /opt/skills/guides/MI355X_MICROARCH.md says 2).
//
// Build: hipcc --offload-arch=gfx950 -O3 valu_issue_bench.hip -o valu_issue_bench ; run on the GPU box.  Developer tool.
//
// Method: one workgroup of 256 x W threads per CU (W waves on each of the CU's 4 SIMDs), every wave runs ITERS trips of a
// 64-instruction body written in inline asm (the compiler cannot fold, reorder or vectorise it) between two s_memtime reads
// (clock64(): shader-clock cycles); all waves of a workgroup start together (barrier).  Reported per test and W:
//   cycles per instruction seen by ONE wave (= latency for a dependent chain),
//   instructions per cycle per SIMD (W waves x 64 x ITERS / cycles) and its inverse, the issue cost of the instruction.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

enum Test {
    FMA_INDEP, FMA_DEP, EXP_INDEP, EXP_DEP, RCP_INDEP, RCP_DEP, PKFMA_INDEP, PKFMA_DEP, CNDMASK_INDEP, MUL_INDEP, ADD_DEP,
    CMP_BRANCH, CMP_SALU, READFIRSTLANE_DEP, DSREAD_DEP, VALU_SALU_MIX, EXP_FMA_MIX, DPP_DEP, CMP_VCC, CMP_SGPR, CNDMASK_VCC, DSREAD128_INDEP, DSWRITE64, NUM_TESTS
};
static const char *kNames[NUM_TESTS] = {
    "v_fma_f32, 8 independent chains", "v_fma_f32, one dependent chain", "v_exp_f32, 8 independent", "v_exp_f32, dependent chain",
    "v_rcp_f32, 8 independent", "v_rcp_f32, dependent chain", "v_pk_fma_f32, 8 independent", "v_pk_fma_f32, dependent chain",
    "v_cndmask_b32 (SGPR mask), 8 independent", "v_mul_f32, 8 independent", "v_add_f32, one dependent chain",
    "v_cmp -> s_and_b64 -> s_cbranch_scc -> dependent v_add (per round trip)", "v_cmp -> s_and_b64 -> v_cndmask (no branch; per round trip)",
    "v_readfirstlane -> v_add (per round trip)", "ds_read_b32 dependent chain (per load)", "v_fma_f32 + s_add_u32 alternating (per pair)",
    "v_exp_f32 : v_fma_f32 1:3 independent", "v_add_f32 dpp quad_perm dependent chain",
    "v_cmp_gt_f32 -> vcc (e32), independent", "v_cmp_gt_f32 -> SGPR pair (e64), independent", "v_cndmask_b32 e32 (vcc), 8 independent",
    "ds_read_b128 uniform address, 8 in flight then wait (per load)", "ds_write_b64 lane-consecutive (per store)",
};
// instructions (or round trips) per body
static const int kPerBody[NUM_TESTS] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 16, 16, 16, 16, 32, 64, 64, 64, 64, 64, 64, 64};

template <int T>
__global__ void bench(long long *cycles, float *sink, int iters) {
    __shared__ float lds[1024];
    const int tid = threadIdx.x;
    for (int i = tid; i < 1024; i += blockDim.x) lds[i] = 0.f;          // ds chain: every word holds offset 0 -> loads address 0
    float a0 = tid * 1e-9f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float m = 0.999f, c = 1e-7f;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const v2 pm = {m, m}, pc = {c, c};
    unsigned long long mask = 0x5555555555555555ull;
    unsigned sacc = 0;
    unsigned addr = 0;
    __syncthreads();
    long long t0 = 0, t1 = 0;
    for (int pass = 0; pass < 2; ++pass) {          // pass 0 warms the instruction cache
        __syncthreads();
        t0 = clock64();
        for (int it = 0; it < iters; ++it) {
            if constexpr (T == FMA_INDEP) {
                asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                  "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if constexpr (T == FMA_DEP) {
                asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(m), "v"(c));
            } else if constexpr (T == EXP_INDEP) {
                asm volatile(REP8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                                  "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if constexpr (T == EXP_DEP) {
                asm volatile(REP64("v_exp_f32 %0, %0\n") : "+v"(a0));
            } else if constexpr (T == RCP_INDEP) {
                asm volatile(REP8("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                                  "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if constexpr (T == RCP_DEP) {
                asm volatile(REP64("v_rcp_f32 %0, %0\n") : "+v"(a0));
            } else if constexpr (T == PKFMA_INDEP) {
                asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                                  "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));
            } else if constexpr (T == PKFMA_DEP) {
                asm volatile(REP64("v_pk_fma_f32 %0, %0, %1, %2\n") : "+v"(p0) : "v"(pm), "v"(pc));
            } else if constexpr (T == CNDMASK_INDEP) {
                asm volatile(REP8("v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n"
                                  "v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "s"(mask));
            } else if constexpr (T == MUL_INDEP) {
                asm volatile(REP8("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                                  "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            } else if constexpr (T == ADD_DEP) {
                asm volatile(REP64("v_add_f32 %0, %0, %1\n") : "+v"(a0) : "v"(c));
            } else if constexpr (T == CMP_BRANCH) {
                // the visit's predicate path: compare -> mask in SGPRs -> scalar test -> branch -> dependent vector work
                asm volatile(REP16("v_cmp_gt_f32 vcc, %0, %2\n s_and_b64 vcc, vcc, %3\n s_cbranch_scc0 1f\n v_add_f32 %0, %0, %1\n1:\n")
                             : "+v"(a0) : "v"(c), "v"(-1.0f), "s"(mask) : "vcc", "scc");
            } else if constexpr (T == CMP_SALU) {
                asm volatile(REP16("v_cmp_gt_f32 vcc, %0, %2\n s_and_b64 vcc, vcc, %3\n v_cndmask_b32 %0, %0, %1, vcc\n")
                             : "+v"(a0) : "v"(c), "v"(-1.0f), "s"(mask) : "vcc", "scc");
            } else if constexpr (T == READFIRSTLANE_DEP) {
                asm volatile(REP16("v_readfirstlane_b32 s40, %0\n v_add_f32 %0, s40, %1\n") : "+v"(a0) : "v"(c) : "s40");
            } else if constexpr (T == DSREAD_DEP) {
                asm volatile(REP16("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(addr) : : "memory");
            } else if constexpr (T == VALU_SALU_MIX) {
                asm volatile(REP4("v_fma_f32 %0, %0, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %1, %1, %9, %10\n s_add_u32 %8, %8, 1\n"
                                  "v_fma_f32 %2, %2, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %3, %3, %9, %10\n s_add_u32 %8, %8, 1\n"
                                  "v_fma_f32 %4, %4, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %5, %5, %9, %10\n s_add_u32 %8, %8, 1\n"
                                  "v_fma_f32 %6, %6, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %7, %7, %9, %10\n s_add_u32 %8, %8, 1\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(sacc) : "v"(m), "v"(c) : "scc");
            } else if constexpr (T == EXP_FMA_MIX) {
                asm volatile(REP8("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                  "v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if constexpr (T == CMP_VCC) {
                asm volatile(REP8("v_cmp_gt_f32 vcc, %0, %8\n v_cmp_gt_f32 vcc, %1, %8\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_gt_f32 vcc, %3, %8\n"
                                  "v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %5, %8\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_gt_f32 vcc, %7, %8\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");
            } else if constexpr (T == CMP_SGPR) {
                asm volatile(REP8("v_cmp_gt_f32 s[40:41], %0, %8\n v_cmp_gt_f32 s[42:43], %1, %8\n v_cmp_gt_f32 s[44:45], %2, %8\n v_cmp_gt_f32 s[46:47], %3, %8\n"
                                  "v_cmp_gt_f32 s[40:41], %4, %8\n v_cmp_gt_f32 s[42:43], %5, %8\n v_cmp_gt_f32 s[44:45], %6, %8\n v_cmp_gt_f32 s[46:47], %7, %8\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m)
                             : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
            } else if constexpr (T == CNDMASK_VCC) {
                asm volatile(REP8("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                                  "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");
            } else if constexpr (T == DSREAD128_INDEP) {
                typedef float v4 __attribute__((ext_vector_type(4)));
                v4 q0, q1, q2, q3, q4, q5, q6, q7;
                asm volatile(REP8("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:48\n"
                                  "ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %8 offset:80\n ds_read_b128 %6, %8 offset:96\n ds_read_b128 %7, %8 offset:112\n"
                                  "s_waitcnt lgkmcnt(0)\n")
                             : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(q7) : "v"(addr) : "memory");
                a0 += q0.x + q7.w;
            } else if constexpr (T == DSWRITE64) {
                const unsigned wa = (unsigned)(tid & 127) * 8u;
                asm volatile(REP64("ds_write_b64 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(wa), "v"(p0) : "memory");
            } else if constexpr (T == DPP_DEP) {
                asm volatile(REP64("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n") : "+v"(a0));
            }
        }
        t1 = clock64();
    }
    if ((tid & 63) == 0) cycles[(size_t)blockIdx.x * (blockDim.x >> 6) + (tid >> 6)] = t1 - t0;
    const float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + (float)addr + (float)sacc + lds[tid & 1023];
    if (r == 123.456f) sink[0] = r;
}

template <int T>
static void run_one(int W, int iters, long long *d_cyc, float *d_sink, double clk_ratio) {
    // W <= 4: one workgroup of 256 W threads per CU; W = 8: two workgroups of 1024 threads per CU (the dispatcher fills the CUs
    // round robin; placement is not forced)
    const int blocks = W == 8 ? 512 : 256, threads = W == 8 ? 1024 : 256 * W;
    hipLaunchKernelGGL(bench<T>, dim3(blocks), dim3(threads), 0, 0, d_cyc, d_sink, iters);
    hipDeviceSynchronize();
    const int nw = blocks * threads / 64;
    std::vector<long long> h(nw);
    hipMemcpy(h.data(), d_cyc, sizeof(long long) * nw, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[nw / 2] * clk_ratio, lo = (double)h[nw / 20] * clk_ratio, hi = (double)h[nw - 1 - nw / 20] * clk_ratio;
    const double n = (double)iters * kPerBody[T];
    std::printf("| %-72s | %d | %8.2f | %8.2f | %8.2f | %6.3f | %6.2f |\n", kNames[T], W, lo / n, med / n, hi / n, (double)W * n / med, med / ((double)W * n));
}

template <int T>
static void run_all(int iters, long long *d_cyc, float *d_sink, double clk_ratio) {
    for (int W : {1, 2, 4, 8}) run_one<T>(W, iters, d_cyc, d_sink, clk_ratio);
    if constexpr (T + 1 < NUM_TESTS) run_all<T + 1>(iters, d_cyc, d_sink, clk_ratio);
}

// ---- which resources do VALU and SALU instructions of DIFFERENT waves share?  16 waves per CU (4 per SIMD; a workgroup's waves go to
// the SIMDs round robin, so waves with equal (index % 4) share a SIMD).  role(wave) picks the stream a wave runs:
//   0 idle, 1 VALU only (8 independent v_fma_f32 chains), 2 SALU only (4 independent s_add_u32 chains), 3 VALU + SALU alternating
// MODE 0: every wave SALU           MODE 1: only the waves of ONE SIMD run SALU      MODE 2: per SIMD two waves VALU, two waves SALU
// MODE 3: only the waves of ONE SIMD run the alternating stream                     MODE 4: per SIMD two waves VALU, two idle
// MODE 5: per SIMD one wave VALU, three waves SALU
template <int MODE>
__global__ void roles(long long *cycles, float *sink, int iters) {
    const int tid = threadIdx.x, wave = tid >> 6;
    int role = 0;
    if (MODE == 0) role = 2;
    if (MODE == 1) role = (wave & 3) == 0 ? 2 : 0;
    if (MODE == 2) role = ((wave >> 2) & 1) ? 2 : 1;
    if (MODE == 3) role = (wave & 3) == 0 ? 3 : 0;
    if (MODE == 4) role = ((wave >> 2) & 1) ? 0 : 1;
    if (MODE == 5) role = (wave >> 2) == 0 ? 1 : 2;
    role = __builtin_amdgcn_readfirstlane(role);
    float a0 = tid * 1e-9f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float m = 0.999f, c = 1e-7f;
    unsigned s0 = 0, s1 = 1, s2 = 2, s3 = 3;
    __syncthreads();
    long long t0 = 0, t1 = 0;
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
        t0 = clock64();
        if (role == 1) {
            for (int it = 0; it < iters; ++it)
                asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                  "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (role == 2) {
            for (int it = 0; it < iters; ++it)
                asm volatile(REP16("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n")
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        } else if (role == 3) {
            for (int it = 0; it < iters; ++it)
                asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n s_add_u32 %10, %10, 1\n v_fma_f32 %1, %1, %8, %9\n s_add_u32 %11, %11, 1\n"
                                  "v_fma_f32 %2, %2, %8, %9\n s_add_u32 %12, %12, 1\n v_fma_f32 %3, %3, %8, %9\n s_add_u32 %13, %13, 1\n"
                                  "v_fma_f32 %4, %4, %8, %9\n s_add_u32 %10, %10, 1\n v_fma_f32 %5, %5, %8, %9\n s_add_u32 %11, %11, 1\n"
                                  "v_fma_f32 %6, %6, %8, %9\n s_add_u32 %12, %12, 1\n v_fma_f32 %7, %7, %8, %9\n s_add_u32 %13, %13, 1\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c), "s"(s0), "s"(s1), "s"(s2), "s"(s3) : "scc");
        }
        t1 = clock64();
    }
    if ((tid & 63) == 0) cycles[(size_t)blockIdx.x * (blockDim.x >> 6) + wave] = role ? t1 - t0 : -1;
    const float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(s0 + s1 + s2 + s3);
    if (r == 123.456f) sink[0] = r;
}

template <int MODE>
static void run_roles(const char *what, int iters, long long *d_cyc, float *d_sink) {
    const int blocks = 256, threads = 1024;
    hipLaunchKernelGGL(roles<MODE>, dim3(blocks), dim3(threads), 0, 0, d_cyc, d_sink, iters);
    hipDeviceSynchronize();
    const int nw = blocks * 16;
    std::vector<long long> h(nw);
    (void)hipMemcpy(h.data(), d_cyc, sizeof(long long) * nw, hipMemcpyDeviceToHost);
    // per role group: waves (index >> 2) & 1 == 0 / 1 within a block, idle waves (-1) dropped
    for (int grp = 0; grp < 4; ++grp) {
        std::vector<long long> v;
        for (int b = 0; b < blocks; ++b)
            for (int w = 4 * grp; w < 4 * grp + 4; ++w)
                if (h[(size_t)b * 16 + w] >= 0) v.push_back(h[(size_t)b * 16 + w]);
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        std::printf("| %-78s | waves %2d-%2d | %8.2f cycles per 64-instruction body-instruction (median; p95 %.2f) |\n", what, 4 * grp, 4 * grp + 3,
                    (double)v[v.size() / 2] / (64.0 * iters), (double)v[v.size() - 1 - v.size() / 20] / (64.0 * iters));
    }
}

// shader clock vs the constant 100 MHz counter: what one clock64() tick is, in shader cycles at the reported clock
__global__ void clock_ratio_kernel(long long *out) {
    const long long w0 = wall_clock64(), c0 = clock64();
    float a = threadIdx.x;
    for (int i = 0; i < 200000; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0\n" : "+v"(a));
    const long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) { out[0] = w1 - w0; out[1] = c1 - c0; out[2] = (long long)a; }
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 200;
    long long *d_cyc;
    float *d_sink;
    hipMalloc(&d_cyc, sizeof(long long) * 512 * 16);
    hipMalloc(&d_sink, 64);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    int wall_khz = 0;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    hipLaunchKernelGGL(clock_ratio_kernel, dim3(1), dim3(64), 0, 0, d_cyc);
    hipDeviceSynchronize();
    long long h[3];
    hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double secs = (double)h[0] / ((double)wall_khz * 1e3);
    const double clock64_hz = (double)h[1] / secs;
    // 200 000 dependent v_fma_f32: if clock64 ticks at the shader clock, ticks per instruction is the dependent-issue latency
    std::printf("device %s, %d CUs, clockRate %d kHz, wall clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate, wall_khz);
    std::printf("clock64() ticks at %.1f MHz (measured against wall_clock64 over %.3f ms); 200k dependent v_fma_f32 took %.2f ticks each\n",
                clock64_hz / 1e6, secs * 1e3, (double)h[1] / 200000.0);
    // express everything in SHADER cycles at the device's reported engine clock if clock64 is not the shader clock
    const double shader_hz = (double)prop.clockRate * 1e3;
    const double ratio = (clock64_hz > 0.5 * shader_hz) ? 1.0 : shader_hz / clock64_hz;
    if (ratio != 1.0) std::printf("clock64() is not the shader clock: cycles below are ticks x %.2f (reported engine clock %.0f MHz)\n", ratio, shader_hz / 1e6);
    std::printf("\n| test | waves/SIMD | cyc/instr per wave p5 | median | p95 | instr/cyc/SIMD | issue cyc/instr |\n|---|---|---|---|---|---|---|\n");
    run_all<0>(iters, d_cyc, d_sink, ratio);
    std::printf("\nShared resources (16 waves per CU = 4 per SIMD; cycles per instruction as ONE wave sees them; an alternating body counts 64 = 32 pairs... see source):\n");
    run_roles<0>("every wave: SALU only (s_add_u32, 4 independent chains)", iters, d_cyc, d_sink);
    run_roles<1>("only the 4 waves of ONE SIMD: SALU only", iters, d_cyc, d_sink);
    run_roles<2>("per SIMD: waves 0-3, 8-11 VALU only (v_fma_f32), waves 4-7, 12-15 SALU only", iters, d_cyc, d_sink);
    run_roles<3>("only the 4 waves of ONE SIMD: v_fma_f32 + s_add_u32 alternating (128 instr per body)", iters, d_cyc, d_sink);
    run_roles<4>("per SIMD: two waves VALU only, two idle", iters, d_cyc, d_sink);
    run_roles<5>("per SIMD: one wave VALU only (waves 0-3), three waves SALU only", iters, d_cyc, d_sink);
    return 0;
}
