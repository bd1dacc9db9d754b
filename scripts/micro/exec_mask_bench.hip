// Microbenchmark: does a wave64 VALU instruction on gfx950 (MI355X) cost less when part of EXEC is zero?
//
// The composites' visits run ~21 % (backward) / ~30 % (forward) live lanes.  If the vector pipe skipped the passes of a wave64
// instruction whose lanes are all inactive (a 32-lane half or a 16-lane row), restricting EXEC to the rows that hold a live pixel
// would buy issue cycles without touching the arithmetic.  This measures it: the streams of valu_issue_bench.hip (8 independent
// v_fma_f32 chains; 8 independent v_exp_f32) under six EXEC masks, 1 / 4 / 8 waves per SIMD.
//
// Build: hipcc --offload-arch=gfx950 -O3 exec_mask_bench.hip -o exec_mask_bench ; run on the GPU box.  Developer tool.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

static const unsigned long long kMasks[] = {0xFFFFFFFFFFFFFFFFull, 0x00000000FFFFFFFFull, 0x000000000000FFFFull, 0x0000FFFF0000FFFFull,
                                            0x00000000000000FFull, 0x0000000000000001ull};
static const char *kMaskNames[] = {"all 64 lanes", "lanes 0-31", "lanes 0-15", "lanes 0-15 + 32-47", "lanes 0-7", "lane 0"};

template <int T>
__global__ void bench(long long *cycles, float *sink, int iters, unsigned long long mask) {
    const int tid = threadIdx.x;
    float a0 = tid * 1e-9f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float m = 0.999f, c = 1e-7f;
    __syncthreads();
    long long t0 = 0, t1 = 0;
    for (int pass = 0; pass < 2; ++pass) {          // pass 0 warms the instruction cache
        __syncthreads();
        t0 = clock64();
        unsigned long long saved;
        asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, %1\n" : "=&s"(saved) : "s"(mask));
        for (int it = 0; it < iters; ++it) {
            if constexpr (T == 0) {
                asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                  "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else {
                asm volatile(REP8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                                  "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        }
        asm volatile("s_mov_b64 exec, %0\n" : : "s"(saved));
        t1 = clock64();
    }
    if ((tid & 63) == 0) cycles[(size_t)blockIdx.x * (blockDim.x >> 6) + (tid >> 6)] = t1 - t0;
    const float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (r == 123.456f) sink[0] = r;
}

template <int T>
static void run(const char *what, int iters, long long *d_cyc, float *d_sink) {
    for (int mi = 0; mi < 6; ++mi)
        for (int W : {1, 4, 8}) {
            const int blocks = W == 8 ? 512 : 256, threads = W == 8 ? 1024 : 256 * W;
            hipLaunchKernelGGL(bench<T>, dim3(blocks), dim3(threads), 0, 0, d_cyc, d_sink, iters, kMasks[mi]);
            (void)hipDeviceSynchronize();
            const int nw = blocks * threads / 64;
            std::vector<long long> h(nw);
            (void)hipMemcpy(h.data(), d_cyc, sizeof(long long) * nw, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            const double n = 64.0 * iters, med = (double)h[nw / 2];
            std::printf("| %-28s | %-20s | %d | %8.2f | %6.2f |\n", what, kMaskNames[mi], W, med / n, med / (W * n));
        }
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 200;
    long long *d_cyc;
    float *d_sink;
    (void)hipMalloc(&d_cyc, sizeof(long long) * 512 * 16);
    (void)hipMalloc(&d_sink, 64);
    std::printf("| stream | EXEC | waves/SIMD | cycles per instruction, one wave (median) | issue cycles per instruction |\n|---|---|---|---|---|\n");
    run<0>("v_fma_f32, 8 indep. chains", iters, d_cyc, d_sink);
    run<1>("v_exp_f32, 8 independent", iters, d_cyc, d_sink);
    return 0;
}
