// Build: hipcc --offload-arch=gfx950 -O3 -o probe/wave_placement scripts/wave_placement.hip ; run on the GPU box.
// Development probe (round 6): where do the waves of a small workgroup land?  The producer / consumer form of the fused mel
// kernel (lra_kernels_pc.h) wants every SIMD of a CU to hold two producer waves and one consumer wave, i.e. the consumer waves of the four resident
// 192-thread workgroups on four different SIMDs.  This kernel has that kernel's footprint (192 or 384 threads, ~34 KB of LDS per 192 threads, a
// 3-waves-per-SIMD register budget), stays resident long enough for a whole CU to fill, and records HW_ID / XCC_ID per wave.
// Output: per (workgroup size) the histogram of SIMD ids by wave index, and per CU how many "last waves" each SIMD received.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NT> __global__ __launch_bounds__(NT, 3) void where(unsigned* out, int spin) {
    extern __shared__ char smem[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // occupy registers like the real kernel (the allocation is what the dispatcher sees): 160 live values across the spin loop
    float r[160];
#pragma unroll
    for (int i = 0; i < 160; ++i) r[i] = (float)(threadIdx.x + i);
    const long long t0 = clock64();
    while (clock64() - t0 < spin) {
#pragma unroll
        for (int i = 0; i < 160; ++i) r[i] = r[i] * 1.0001f + 0.5f;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 160; ++i) s += r[i];
    if (s == 12345.f) smem[threadIdx.x] = 1;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (NT / 64) + threadIdx.x / 64;
        out[2 * w] = hw;
        out[2 * w + 1] = xcc;
    }
}

template <int NT> void run(int n_wg, int lds) {
    const int waves = n_wg * (NT / 64);
    unsigned* d;
    CK(hipMalloc(&d, waves * 8));
    CK(hipMemset(d, 0, waves * 8));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(where<NT>, dim3(n_wg), dim3(NT), lds, 0, d, 200000);
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned> h(2 * waves);
    CK(hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost));
    const int W = NT / 64;
    // histogram: wave index within the workgroup -> SIMD
    long long hist[8][4] = {};
    // per CU (xcc, se, sh, cu): count of last waves per SIMD, count of all waves per SIMD
    std::map<unsigned, std::vector<int>> last, all;
    int rr_ok = 0;  // workgroups whose waves sit on consecutive SIMDs (mod 4)
    for (int g = 0; g < n_wg; ++g) {
        bool consecutive = true;
        int prev = -1;
        for (int w = 0; w < W; ++w) {
            const unsigned hw = h[2 * (g * W + w)], xcc = h[2 * (g * W + w) + 1] & 0xf;
            const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            hist[w][simd]++;
            const unsigned key = (xcc << 16) | (se << 8) | (sh << 4) | cu;
            if (!all.count(key)) { all[key] = std::vector<int>(4, 0); last[key] = std::vector<int>(4, 0); }
            all[key][simd]++;
            if (w == W - 1) last[key][simd]++;
            if (prev >= 0 && simd != (prev + 1) % 4) consecutive = false;
            prev = simd;
        }
        rr_ok += consecutive;
    }
    printf("== %d-thread workgroups, %d KB LDS, %d workgroups: waves on consecutive SIMDs in %d of them\n", NT, lds / 1024, n_wg, rr_ok);
    for (int w = 0; w < W; ++w) printf("   wave %d -> SIMD 0..3: %lld %lld %lld %lld\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    // how evenly do the last waves spread inside a CU?
    int cus = 0, even = 0;
    std::map<std::string, int> patterns;
    for (auto& kv : last) {
        ++cus;
        const auto& l = kv.second;
        const auto& a = all[kv.first];
        char buf[96];
        snprintf(buf, sizeof buf, "last %d %d %d %d | all %d %d %d %d", l[0], l[1], l[2], l[3], a[0], a[1], a[2], a[3]);
        patterns[buf]++;
        int mx = 0, mn = 1 << 30;
        for (int s = 0; s < 4; ++s) { mx = l[s] > mx ? l[s] : mx; mn = l[s] < mn ? l[s] : mn; }
        even += (mx - mn <= 1);
    }
    printf("   CUs seen: %d, last waves spread evenly (max - min <= 1 over the SIMDs) on %d\n", cus, even);
    int shown = 0;
    for (auto& p : patterns) { if (shown++ < 12) printf("   %4d CUs: %s\n", p.second, p.first.c_str()); }
    CK(hipFree(d));
}

int main() {
    run<192>(1024, 34 * 1024);   // four workgroups per CU, one round
    run<192>(2048, 34 * 1024);   // two rounds (steady-state refill order)
    run<384>(512, 68 * 1024);
    run<256>(512, 40 * 1024);
    return 0;
}
