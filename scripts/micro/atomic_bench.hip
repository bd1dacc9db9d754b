// Microbenchmark: throughput of global float atomics (no return) into a [P][16]-float accumulator (one 64-byte line per
// row) for the access patterns the backward composite could use.  Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
// atomic_bench.hip -o atomic_bench ; run on the GPU box.  Developer tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// mode 0: 4 lanes (0,16,32,48) -> 4 consecutive floats of ONE random line per wave-iteration   (generation 3's publish)
// mode 1: 4 lanes -> 4 different random lines
// mode 2: 64 lanes -> 64 different random lines (1 float each)
// mode 3: 64 lanes -> 16 random lines x 4 consecutive floats
// mode 4: 64 lanes -> 64 different lines, but the same lane repeats 10 floats of its line over 10 instructions (a per-thread flush)
// mode 6: 12 lanes -> one line x 12 floats;  mode 7: 40 lanes -> 4 lines x 10 floats
// mode 5: 16 lanes (4 per row) -> 4 random lines x 4 floats (generation 4 direct publish)
template <int MODE>
__global__ void k(float *acc, int P, int iters, int local) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (int it = 0; it < iters; ++it) {
        const unsigned seed = wave * 7919u + (unsigned)it * 104729u;
        // `local`: ids drawn from a window of 4096 rows around the wave's own region (tile-coherent Gaussians), else uniform
        auto row = [&](unsigned s) { unsigned h = hash(s); return local ? (unsigned)(((unsigned long long)wave * 23u + (h & 4095u)) % (unsigned)P) : h % (unsigned)P; };
        if (MODE == 0) {
            if ((lane & 15) == 0) atomicAdd(acc + (size_t)row(seed) * 16 + (lane >> 4), 1.0f);
        } else if (MODE == 1) {
            if ((lane & 15) == 0) atomicAdd(acc + (size_t)row(seed + lane) * 16, 1.0f);
        } else if (MODE == 2) {
            atomicAdd(acc + (size_t)row(seed + lane) * 16, 1.0f);
        } else if (MODE == 3) {
            atomicAdd(acc + (size_t)row(seed + (lane >> 2)) * 16 + (lane & 3), 1.0f);
        } else if (MODE == 4) {
            float *d = acc + (size_t)row(seed + lane) * 16;
#pragma unroll
            for (int c = 0; c < 10; ++c) atomicAdd(d + c, 1.0f);
        } else if (MODE == 6) {
            // 12 lanes (3 per row) -> ONE line x 12 floats
            if ((lane & 15) < 3) atomicAdd(acc + (size_t)row(seed) * 16 + (lane >> 4) * 3 + (lane & 15), 1.0f);
        } else if (MODE == 7) {
            // 40 lanes (10 per row) -> 4 lines (one per row) x 10 floats
            if ((lane & 15) < 10) atomicAdd(acc + (size_t)row(seed + (lane >> 4)) * 16 + (lane & 15), 1.0f);
        } else if (MODE == 5) {
            if ((lane & 3) == 0) atomicAdd(acc + (size_t)row(seed + (lane >> 4)) * 16 + ((lane >> 2) & 3), 1.0f);
        }
    }
}

template <int MODE>
static void run(float *acc, int P, int waves, int iters, int local, double per_iter_atomics, const char *name) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<waves / 4, 256>>>(acc, P, 2, local);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<waves / 4, 256>>>(acc, P, iters, local);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double n = (double)waves * iters * per_iter_atomics;
    printf("%-58s local=%d  %8.3f ms  %7.1f lane-atomics/ns  %7.1f wave-instr/ns\n", name, local, ms, n / (ms * 1e6),
           (double)waves * iters * (MODE == 4 ? 10 : 1) / (ms * 1e6));
}

int main() {
    const int P = 300000, waves = 12900 / 4 * 4, iters = 200;
    float *acc;
    hipMalloc(&acc, sizeof(float) * 16 * (size_t)P);
    hipMemset(acc, 0, sizeof(float) * 16 * (size_t)P);
    for (int local = 0; local < 2; ++local) {
        run<0>(acc, P, waves, iters, local, 4, "0: 4 lanes -> one line x 4 floats");
        run<1>(acc, P, waves, iters, local, 4, "1: 4 lanes -> 4 lines");
        run<6>(acc, P, waves, iters, local, 12, "6: 12 lanes -> one line x 12 floats");
        run<7>(acc, P, waves, iters, local, 40, "7: 40 lanes -> 4 lines x 10 floats");
        run<5>(acc, P, waves, iters, local, 16, "5: 16 lanes -> 4 lines x 4 floats");
        run<3>(acc, P, waves, iters, local, 64, "3: 64 lanes -> 16 lines x 4 floats");
        run<2>(acc, P, waves, iters, local, 64, "2: 64 lanes -> 64 lines");
        run<4>(acc, P, waves, iters / 4, local, 640, "4: 64 lanes -> 64 lines, 10 floats each (10 instr)");
    }
    return 0;
}
