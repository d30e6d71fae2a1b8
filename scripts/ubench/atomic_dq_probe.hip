// Probe for a fused attention backward: what does the memory system sustain when every resident workgroup adds a 64 x 64 fp32
// dQ tile per step into its head's slab?  1024 workgroups (128 key blocks x 8 heads), 512 steps each = 8.6 GB of fp32 adds.
// modes: 0 agent-scope atomic add, 1 workgroup-scope atomic add (executes in the XCD's L2), 2 plain load+add+store (wrong
// under contention; traffic reference), 3 agent-scope with per-key-block staggered start, 4 wg-scope staggered
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* dq, int nkb, int nt, int spin) {
    const int id = blockIdx.x;
    // XCD-aware: consecutive block ids go round-robin over 8 XCDs; give XCD x the head x
    const int bh = id % 8, kb = id / 8;
    float* slab = dq + (size_t)bh * nt * 4096;
    const int tid = threadIdx.x;
    const bool stag = MODE >= 3;
    float v = 1.0f;
    for (int s = 0; s < nt; ++s) {
        const int t = stag ? (s + kb * (nt / nkb)) % nt : s;
        float* tile = slab + (size_t)t * 4096;
        // stand-in for the step's compute (dependent fma chain)
        for (int i = 0; i < spin; ++i) v = __builtin_fmaf(v, 1.0000001f, 1e-9f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float* p = tile + j * 512 + tid;
            if (MODE == 0 || MODE == 3) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (MODE == 1 || MODE == 4) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else *p += v;
        }
    }
}

int main(int argc, char** argv) {
    const int nbh = 8, nkb = 128, nt = 512;
    float* dq;
    const size_t n = (size_t)nbh * nt * 4096;
    CK(hipMalloc(&dq, n * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int spins[] = {0, 2000, 4000};
    for (int spin : spins)
    for (int mode = 0; mode < 5; ++mode) {
        float best = 1e9f;
        double sum = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(dq, 0, n * 4));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            switch (mode) {
                case 0: hipLaunchKernelGGL(probe<0>, dim3(nbh * nkb), dim3(512), 0, 0, dq, nkb, nt, spin); break;
                case 1: hipLaunchKernelGGL(probe<1>, dim3(nbh * nkb), dim3(512), 0, 0, dq, nkb, nt, spin); break;
                case 2: hipLaunchKernelGGL(probe<2>, dim3(nbh * nkb), dim3(512), 0, 0, dq, nkb, nt, spin); break;
                case 3: hipLaunchKernelGGL(probe<3>, dim3(nbh * nkb), dim3(512), 0, 0, dq, nkb, nt, spin); break;
                case 4: hipLaunchKernelGGL(probe<4>, dim3(nbh * nkb), dim3(512), 0, 0, dq, nkb, nt, spin); break;
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        std::vector<float> host(4096 * 4);
        CK(hipMemcpy(host.data(), dq + (size_t)3 * nt * 4096 + 77 * 4096, host.size() * 4, hipMemcpyDeviceToHost));
        for (float x : host) sum += x;
        const double gb = (double)n * 4 * nkb / 1e9;
        printf("spin %5d mode %d: %8.3f ms  %7.2f GB of adds -> %7.1f GB/s   (tile mean %.3f, expect ~%d when atomic)\n", spin, mode, best, gb, gb / best * 1e3,
               sum / host.size(), nkb);
    }
    return 0;
}
