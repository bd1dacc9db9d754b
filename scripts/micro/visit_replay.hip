// Microbenchmark: the instruction stream of ONE VISIT of the backward composite's phase 1 (render.hip, K7), replayed as straight-line
// inline asm without its LDS traffic, at 2 / 4 waves per SIMD -- what the VECTOR PIPE ALONE allows per visit, and what each class of
// instruction in it costs (variants with one class removed or replaced).  Companion of valu_issue_bench.hip; developer tool.
// Build: hipcc --offload-arch=gfx950 -O3 visit_replay.hip -o visit_replay ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

// operands: %0 Tr  %1 R  %2 t0  %3 t1  %4 t2  %5 p2  %6 G  %7 al  %8 cd  %9 vv  %10 ww  %11 Gl
//           %12 mx %13 my %14 fpx %15 fpy %16 aA %17 aB %18 aC %19 op %20 f0 %21 f1 %22 f2 %23 f3 %24 d0 %25 d1 %26 d2 %27 d3 %28 e %29 last %30 amin(s)
#define GEO \
    "v_sub_f32 %2, %13, %15\n v_sub_f32 %3, %12, %14\n v_mul_f32 %4, %17, %2\n v_fmac_f32 %4, %16, %3\n v_mul_f32 %5, %3, %4\n" \
    "v_mul_f32 %4, %18, %2\n v_fmac_f32 %5, %2, %4\n"
#define EXPA "v_exp_f32 %6, %5\n v_mul_f32 %7, %19, %6\n v_min_f32 %7, 0x3f7d70a4, %7\n"
#define EXPA_NOTRANS "v_mul_f32 %6, %5, %5\n v_mul_f32 %7, %19, %6\n v_min_f32 %7, 0x3f7d70a4, %7\n"
#define LIVE \
    "v_cmp_le_i32 s[40:41], %28, %29\n v_cmp_ge_f32 vcc, 0, %5\n s_and_b64 s[40:41], vcc, s[40:41]\n v_cmp_le_f32 vcc, %30, %7\n" \
    "s_and_b64 vcc, s[40:41], vcc\n s_cmp_eq_u64 vcc, 0\n s_cbranch_scc1 1f\n"
#define LIVE_NOBRANCH \
    "v_cmp_le_i32 s[40:41], %28, %29\n v_cmp_ge_f32 vcc, 0, %5\n s_and_b64 s[40:41], vcc, s[40:41]\n v_cmp_le_f32 vcc, %30, %7\n" \
    "s_and_b64 vcc, s[40:41], vcc\n"
#define SEL_AL "v_cndmask_b32 %7, 0, %7, vcc\n"
#define SEL_G "v_cndmask_b32 %11, 0, %6, vcc\n"
#define SEL_AL64 "s_mov_b64 s[42:43], vcc\n v_cndmask_b32_e64 %7, 0, %7, s[42:43]\n"
#define SEL_G64 "v_cndmask_b32_e64 %11, 0, %6, s[42:43]\n"
#define REC \
    "v_sub_f32 %2, 1.0, %7\n v_mul_f32 %8, %24, %20\n v_rcp_f32 %2, %2\n v_fmac_f32 %8, %21, %25\n v_fmac_f32 %8, %22, %26\n" \
    "v_fmac_f32 %8, %23, %27\n v_mul_f32 %0, %0, %2\n v_mul_f32 %3, %1, %2\n"
#define REC_NOTRANS \
    "v_sub_f32 %2, 1.0, %7\n v_mul_f32 %8, %24, %20\n v_mul_f32 %2, %2, %2\n v_fmac_f32 %8, %21, %25\n v_fmac_f32 %8, %22, %26\n" \
    "v_fmac_f32 %8, %23, %27\n v_mul_f32 %0, %0, %2\n v_mul_f32 %3, %1, %2\n"
#define TAIL "v_fma_f32 %3, %8, %0, -%3\n v_mul_f32 %9, %11, %3\n v_mul_f32 %10, %7, %0\n v_fmac_f32 %1, %8, %10\n1:\n"
#define TAIL_NOG "v_fma_f32 %3, %8, %0, -%3\n v_mul_f32 %9, %6, %3\n v_mul_f32 %10, %7, %0\n v_fmac_f32 %1, %8, %10\n1:\n"

enum Variant { FULL, NO_BRANCH, NO_LIVE, NO_SELECT, SELECT_E64, NO_TRANS, ONLY_PLAIN, NUM_VARIANTS };
static const char *kNames[NUM_VARIANTS] = {
    "full visit: geometry, exp, 3 compares + 2 s_and + branch, 2 selects (vcc), rcp, recursion (31 VALU, 4 SALU)",
    "same without the scalar test + branch on the live mask",
    "same without compares, s_and, branch, selects (every lane live)",
    "compares + branch kept, the two selects dropped",
    "selects in their e64 form on an SGPR pair (copied from vcc by s_mov_b64)",
    "full, exp and rcp replaced by multiplies",
    "no compares / selects / branch, exp and rcp replaced by multiplies (plain VALU only, 24)",
};

template <int V>
__global__ void replay(long long *cycles, float *sink, int iters) {
    const int tid = threadIdx.x;
    float Tr = 1.f, R = 0.f, t0 = 0, t1 = 0, t2 = 0, p2 = 0, G = 0, al = 0, cd = 0, vv = 0, ww = 0, Gl = 0;
    const float mx = 3.2f, my = 4.1f, fpx = (float)(tid & 7), fpy = (float)((tid >> 3) & 7);
    const float aA = -0.05f, aB = 0.001f, aC = -0.04f, op = 1e-3f, f0 = 0.3f, f1 = 0.2f, f2 = 0.1f, f3 = 0.5f, d0 = 1e-3f, d1 = 2e-3f, d2 = -1e-3f, d3 = 5e-4f;
    const int e = 5, last = 100;
    const float amin = 1e-9f;           // every lane stays live (alpha ~ 1e-3: the recursion stays finite over the run)
    __syncthreads();
    long long tA = 0, tB = 0;
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
        tA = clock64();
        for (int it = 0; it < iters; ++it) {
#define BODY(x) asm volatile(REP8(x) : "+v"(Tr), "+v"(R), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(p2), "+v"(G), "+v"(al), "+v"(cd), "+v"(vv), "+v"(ww), "+v"(Gl) \
                             : "v"(mx), "v"(my), "v"(fpx), "v"(fpy), "v"(aA), "v"(aB), "v"(aC), "v"(op), "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(d0), "v"(d1), "v"(d2), "v"(d3), \
                               "v"(e), "v"(last), "s"(amin) : "vcc", "scc", "s40", "s41", "s42", "s43")
            if constexpr (V == FULL) BODY(GEO EXPA LIVE SEL_AL REC SEL_G TAIL);
            else if constexpr (V == NO_BRANCH) BODY(GEO EXPA LIVE_NOBRANCH SEL_AL REC SEL_G TAIL);
            else if constexpr (V == NO_LIVE) BODY(GEO EXPA REC TAIL_NOG);
            else if constexpr (V == NO_SELECT) BODY(GEO EXPA LIVE REC TAIL_NOG);
            else if constexpr (V == SELECT_E64) BODY(GEO EXPA LIVE SEL_AL64 REC SEL_G64 TAIL);
            else if constexpr (V == NO_TRANS) BODY(GEO EXPA_NOTRANS LIVE SEL_AL REC_NOTRANS SEL_G TAIL);
            else if constexpr (V == ONLY_PLAIN) BODY(GEO EXPA_NOTRANS REC_NOTRANS TAIL_NOG);
        }
        tB = clock64();
    }
    if ((tid & 63) == 0) cycles[(size_t)blockIdx.x * (blockDim.x >> 6) + (tid >> 6)] = tB - tA;
    const float r = Tr + R + t0 + t1 + t2 + p2 + G + al + cd + vv + ww + Gl;
    if (r == 123.456f) sink[0] = r;
}

template <int V>
static void run_all(int iters, long long *d_cyc, float *d_sink) {
    for (int W : {1, 2, 4}) {
        const int blocks = 256, threads = 256 * W;
        hipLaunchKernelGGL(replay<V>, dim3(blocks), dim3(threads), 0, 0, d_cyc, d_sink, iters);
        (void)hipDeviceSynchronize();
        const int nw = blocks * threads / 64;
        std::vector<long long> h(nw);
        (void)hipMemcpy(h.data(), d_cyc, sizeof(long long) * nw, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        // the SIMD's throughput: W waves finish 8 * iters visits each by the time the slowest is done
        const double slow = (double)h[nw - 1 - nw / 50], med = (double)h[nw / 2];
        std::printf("| %-108s | %d | %7.1f | %7.1f | %7.1f |\n", kNames[V], W, med / (8.0 * iters), slow / (8.0 * iters), slow / (8.0 * iters * W));
    }
    if constexpr (V + 1 < NUM_VARIANTS) run_all<V + 1>(iters, d_cyc, d_sink);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 400;
    long long *d_cyc;
    float *d_sink;
    (void)hipMalloc(&d_cyc, sizeof(long long) * 256 * 16);
    (void)hipMalloc(&d_sink, 64);
    std::printf("| visit variant | waves/SIMD | cycles per visit, one wave (median) | slowest wave | SIMD cycles per visit |\n|---|---|---|---|---|\n");
    run_all<0>(iters, d_cyc, d_sink);
    return 0;
}
