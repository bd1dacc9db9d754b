// Microbenchmark: rate of RETURNING 32-bit atomic adds on a few thousand hot counters (the bucket-slot atomics of the
// per-Gaussian kernel F1) by memory scope.  Device-scope atomics are served coherently across the 8 XCDs (~21 requests / ns,
// atomic_bench.hip); counters that only ONE XCD touches could be served by that XCD's L2 with workgroup-scope atomics.
//   mode 0: agent scope, one counter set shared by all XCDs (what F1 does)
//   mode 1: agent scope, one counter set per XCD (set = HW_REG_XCC_ID)
//   mode 2: workgroup scope, one counter set per XCD
// Every returned slot is recorded; the host checks that the slots of every (set, counter) are exactly 0 .. n-1 (a scope
// that is too weak shows up as duplicate slots).  Build: hipcc --offload-arch=gfx950 -O3 scope_bench.hip -o scope_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kCounters = 3225, kStride = 32, kSets = 8, kCap = 4096;

__device__ __forceinline__ unsigned hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned *counters, unsigned char *seen, int per_thread) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    unsigned xcc = 0;
    if (MODE != 0) xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;          // HW_REG_XCC_ID, 4 bits
    for (int it = 0; it < per_thread; ++it) {
        const unsigned c = hash(tid * 31u + (unsigned)it) % kCounters;
        unsigned *p = counters + ((size_t)xcc * kCounters + c) * kStride;
        unsigned slot;
        if (MODE == 2) slot = __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else slot = __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (slot < kCap) seen[((size_t)xcc * kCounters + c) * kCap + slot] += 1;      // (non-atomic: a duplicate slot may also lose an increment)
    }
}

template <int MODE>
static void run(unsigned *counters, unsigned char *seen, const char *name) {
    const int blocks = 1172, per_thread = 3;                // ~ 300 k lanes x 2.36 instances
    const size_t nc = (size_t)kSets * kCounters * kStride, ns = (size_t)kSets * kCounters * kCap;
    hipMemset(counters, 0, nc * 4);
    hipMemset(seen, 0, ns);
    k<MODE><<<blocks, 256>>>(counters, seen, 1);            // warm-up
    hipDeviceSynchronize();
    hipMemset(counters, 0, nc * 4);
    hipMemset(seen, 0, ns);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(counters, seen, per_thread);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned> hc(nc);
    std::vector<unsigned char> hs(ns);
    hipMemcpy(hc.data(), counters, nc * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hs.data(), seen, ns, hipMemcpyDeviceToHost);
    unsigned long long total = 0, bad = 0;
    int sets_used = 0;
    for (int s = 0; s < kSets; ++s) {
        unsigned long long in_set = 0;
        for (int c = 0; c < kCounters; ++c) {
            const unsigned n = hc[((size_t)s * kCounters + c) * kStride];
            in_set += n;
            for (unsigned j = 0; j < kCap; ++j) {
                const unsigned char v = hs[((size_t)s * kCounters + c) * kCap + j];
                if ((j < n) != (v == 1)) ++bad;
            }
        }
        total += in_set;
        sets_used += in_set > 0;
    }
    const double n = (double)blocks * 256 * per_thread;
    printf("%-52s %7.3f ms  %6.1f atomics/ns  total %llu of %.0f  sets used %d  slot errors %llu\n", name, ms, n / (ms * 1e6), total, n, sets_used, bad);
}

int main() {
    unsigned *counters;
    unsigned char *seen;
    hipMalloc(&counters, (size_t)kSets * kCounters * kStride * 4);
    hipMalloc(&seen, (size_t)kSets * kCounters * kCap);
    run<0>(counters, seen, "0: agent scope, shared counters");
    run<1>(counters, seen, "1: agent scope, per-XCD counters");
    run<2>(counters, seen, "2: workgroup scope, per-XCD counters");
    return 0;
}
