// Build: hipcc --offload-arch=gfx950 -O3 -o probe/xcd_affinity scripts/xcd_affinity.hip ; run on the GPU box.
// Development probe (round 5): is the write rate of an XCD a function of WHICH part of an allocation it writes?  profiles/r05_pitch.md found the
// store-bound STFT 0.63-0.75 ms depending on where its 2.7 GB output landed.  If HBM were interleaved coarsely, an allocation's eighths would sit
// on different stacks and an XCD would write "near" eighths faster than "far" ones.  Workgroups are dealt to the XCDs round-robin (block b -> XCD b % 8),
// so a kernel whose other blocks return at once runs on ONE XCD; it writes one eighth of the buffer.  8 x 8 matrix of GB/s per allocation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// blocks with b % 8 == xcd write chunk b / 8 of [base, base + bytes): 256 threads x 16 B, strided by 4 KiB rows
__global__ __launch_bounds__(256) void write_one_xcd(char* base, size_t bytes, int xcd, size_t chunk) {
    const int b = blockIdx.x;
    if (b % 8 != xcd) return;
    const size_t j = b / 8;
    char* p = base + j * chunk;
    if (j * chunk >= bytes) return;
    const f4 v = {1.f, 2.f, 3.f, (float)b};
    for (size_t o = threadIdx.x * 16; o < chunk; o += 256 * 16) *reinterpret_cast<f4*>(p + o) = v;
}
// all XCDs at once: XCD x writes eighth perm[x]
struct Perm { int p[8]; };
__global__ __launch_bounds__(256) void write_all(char* base, size_t eighth, Perm perm, size_t chunk) {
    const int b = blockIdx.x, x = b % 8;
    const size_t j = b / 8;
    if (j * chunk >= eighth) return;
    char* p = base + (size_t)perm.p[x] * eighth + j * chunk;
    const f4 v = {1.f, 2.f, 3.f, (float)b};
    for (size_t o = threadIdx.x * 16; o < chunk; o += 256 * 16) *reinterpret_cast<f4*>(p + o) = v;
}

int main(int argc, char** argv) {
    const int n_alloc = argc > 1 ? atoi(argv[1]) : 4;
    const size_t total = (size_t)256 * 1292 * 8200;  // the STFT output of BASELINE configs[1]
    const size_t eighth = (total / 8) & ~(size_t)0xfffff;
    const size_t chunk = 1 << 20;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<char*> bufs;
    for (int a = 0; a < n_alloc; ++a) {
        char* d;
        CK(hipMalloc(&d, total + (a * 3 << 20)));
        bufs.push_back(d);
        CK(hipMemset(d, 0, total));
        const unsigned grid = (unsigned)(8 * ((eighth + chunk - 1) / chunk));
        // warm clocks
        Perm id; for (int i = 0; i < 8; ++i) id.p[i] = i;
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(write_all, dim3(grid), dim3(256), 0, 0, d, eighth, id, chunk);
        CK(hipDeviceSynchronize());
        printf("allocation %d at %p: GB/s of XCD x (row) writing eighth r (column) alone\n", a, (void*)d);
        double mat[8][8];
        for (int x = 0; x < 8; ++x) {
            for (int r = 0; r < 8; ++r) {
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipEventRecord(e0, 0));
                    hipLaunchKernelGGL(write_one_xcd, dim3(grid), dim3(256), 0, 0, d + (size_t)r * eighth, eighth, x, chunk);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    best = std::min(best, ms);
                }
                mat[x][r] = eighth / (best * 1e-3) / 1e9;
                printf(" %6.0f", mat[x][r]);
            }
            printf("\n");
        }
        // everything at once: identity, reversed, rotated assignments of eighths to XCDs
        const char* names[4] = {"identity", "reversed", "rotate+1", "rotate+4"};
        for (int m = 0; m < 4; ++m) {
            Perm pm;
            for (int i = 0; i < 8; ++i) pm.p[i] = m == 0 ? i : m == 1 ? 7 - i : m == 2 ? (i + 1) % 8 : (i + 4) % 8;
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(write_all, dim3(grid), dim3(256), 0, 0, d, eighth, pm, chunk);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms);
            }
            printf("  all XCDs, %-9s: %.3f ms  %.0f GB/s\n", names[m], best, 8.0 * eighth / (best * 1e-3) / 1e9);
        }
        // fine interleave: block b writes chunk b (every XCD everywhere)
        {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(write_one_xcd, dim3(1), dim3(256), 0, 0, d, 0, 0, chunk);  // (no-op spacer)
                CK(hipEventRecord(e0, 0));
                CK(hipMemsetAsync(d, 1, 8 * eighth, 0));
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms);
            }
            printf("  hipMemsetAsync of the same bytes: %.3f ms  %.0f GB/s\n", best, 8.0 * eighth / (best * 1e-3) / 1e9);
        }
    }
    return 0;
}
