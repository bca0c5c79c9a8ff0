// Build: hipcc --offload-arch=gfx950 -O3 -o probe/storepat scripts/storepat.hip ; run on the GPU box.  Results of 2026-09 on MI355X are quoted in DESIGN.md (section 6).
// Store-pattern microbenchmark (development): how fast can the chip absorb the STFT output stream?
// Each wave64 writes `iters` consecutive rows of ROWB bytes of one "clip"; variants differ in how.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// mode 0: 16 x dwordx2, ascending k / descending M-k interleaved (the kernel's pattern), row pitch 1025 c64
// mode 1: 16 x dwordx2 ascending only
// mode 2: 8 x dwordx4 ascending (lane owns 2 adjacent bins), pitch 1025 (rows 16B-misaligned every other frame)
// mode 3: like 0 but pitch 1024 (aligned rows, no Nyquist)
// mode 4: like 2 but pitch 1024
template <int MODE, bool READ> __global__ __launch_bounds__(64) void k(f2* __restrict__ out, const float* __restrict__ in, int rows_per_clip, int iters, int wg_per_clip) {
    const int clip = blockIdx.x / wg_per_clip, part = blockIdx.x % wg_per_clip;
    const int tf = threadIdx.x;
    constexpr int PITCH = (MODE >= 3) ? 1024 : 1025;
    f2 v = {(float)tf, (float)part};
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int row = part * iters + it;
        if (row >= rows_per_clip) break;
        // read side: 512 new samples per frame (8 dwords per lane), as the kernel does
        const float* src = in + (size_t)clip * 661500 + (size_t)row * 512 + tf;
        if (READ) {
        #pragma unroll
        for (int c = 0; c < 8; ++c) acc += src[c * 64];
        }
        f2* r = out + ((size_t)clip * rows_per_clip + row) * PITCH;
        v.x += acc;
        if (MODE == 0 || MODE == 3) {
            f2* pk = r + tf; f2* pm = r + (1024 - tf);
            #pragma unroll
            for (int i = 0; i < 8; ++i) { pk[i * 64] = v; if (MODE == 0 || i > 0 || tf > 0) pm[-i * 64] = v; }
            if (MODE == 0 && tf == 0) r[512] = v;
        } else if (MODE == 1) {
            #pragma unroll
            for (int i = 0; i < 16; ++i) r[tf + i * 64] = v;
            if (tf == 0) r[1024] = v;
        } else {
            f4 w = {v.x, v.y, v.x, v.y};
            #pragma unroll
            for (int i = 0; i < 8; ++i) __builtin_memcpy(reinterpret_cast<char*>(r) + (size_t)(tf * 2 + i * 128) * 8, &w, 16);
            if (MODE == 2 && tf == 0) r[1024] = v;
        }
    }
}

template <int MODE, bool READ = true> float run(f2* out, const float* in, int batch, int rows, int iters) {
    const int wgpc = (rows + iters - 1) / iters;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<MODE, READ>), dim3(batch * wgpc), dim3(64), 0, 0, out, in, rows, iters, wgpc);
    hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((k<MODE, READ>), dim3(batch * wgpc), dim3(64), 0, 0, out, in, rows, iters, wgpc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

__global__ __launch_bounds__(256) void fillk(f4* __restrict__ out, size_t n4) {
    f4 w = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = w;
}
__global__ __launch_bounds__(256) void fillk_blocked(f4* __restrict__ out, size_t n4, size_t per_wg) {
    f4 w = {1.f, 2.f, 3.f, 4.f};
    const size_t b0 = (size_t)blockIdx.x * per_wg;
    for (size_t i = threadIdx.x; i < per_wg && b0 + i < n4; i += 256) out[b0 + i] = w;
}
__global__ __launch_bounds__(256) void readk(const f4* __restrict__ in, float* __restrict__ out, size_t n4) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += in[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
int main() {
    const int batch = 256, rows = 1292;
    f2* out; float* in;
    hipMalloc(&out, (size_t)batch * rows * 1025 * 8 + 4096);
    hipMalloc(&in, (size_t)batch * 661500 * 4 + (1 << 20));
    hipMemset(in, 0, (size_t)batch * 661500 * 4 + (1 << 20));
    const double bytes = (double)batch * rows * (8200 + 2048);
    for (int iters : {32, 81}) {
        printf("iters %d: ", iters);
        printf("m0 %.0f GB/s  ", bytes / run<0>(out, in, batch, rows, iters) / 1e6);
        printf("m1 %.0f GB/s  ", bytes / run<1>(out, in, batch, rows, iters) / 1e6);
        printf("m2 %.0f GB/s  ", bytes / run<2>(out, in, batch, rows, iters) / 1e6);
        printf("m3 %.0f GB/s  ", bytes / run<3>(out, in, batch, rows, iters) / 1e6);
        printf("m4 %.0f GB/s\n", bytes / run<4>(out, in, batch, rows, iters) / 1e6);
        const double wb = (double)batch * rows * 8200;
        printf("   write only: m0 %.0f  m1 %.0f  m2 %.0f  m4 %.0f GB/s\n", wb / run<0, false>(out, in, batch, rows, iters) / 1e6, wb / run<1, false>(out, in, batch, rows, iters) / 1e6,
               wb / run<2, false>(out, in, batch, rows, iters) / 1e6, (double)batch * rows * 8192 / run<4, false>(out, in, batch, rows, iters) / 1e6);
    }
    {
        const size_t n4 = (size_t)batch * rows * 1025 * 8 / 16;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int grid : {2048, 8192, 65536}) {
            for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(fillk, dim3(grid), dim3(256), 0, 0, (f4*)out, n4);
            hipEventRecord(e0);
            for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(fillk, dim3(grid), dim3(256), 0, 0, (f4*)out, n4);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("fill grid-stride grid %d: %.0f GB/s\n", grid, n4 * 16.0 / (ms / 10) / 1e6);
        }
        for (size_t per : {(size_t)4096, (size_t)65536}) {
            const int grid = (int)((n4 + per - 1) / per);
            for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(fillk_blocked, dim3(grid), dim3(256), 0, 0, (f4*)out, n4, per);
            hipEventRecord(e0);
            for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(fillk_blocked, dim3(grid), dim3(256), 0, 0, (f4*)out, n4, per);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("fill blocked %zu KB per WG: %.0f GB/s\n", per * 16 / 1024, n4 * 16.0 / (ms / 10) / 1e6);
        }
        for (int grid : {2048, 16384, 65536}) {
            for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(readk, dim3(grid), dim3(256), 0, 0, (const f4*)out, in, n4);
            hipEventRecord(e0);
            for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(readk, dim3(grid), dim3(256), 0, 0, (const f4*)out, in, n4);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float msr; hipEventElapsedTime(&msr, e0, e1);
            printf("read-only grid-stride grid %d: %.0f GB/s\n", grid, n4 * 16.0 / (msr / 10) / 1e6);
        }
        hipEventRecord(e0);
        for (int w = 0; w < 10; ++w) hipMemsetAsync(out, 0, n4 * 16, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("hipMemsetAsync: %.0f GB/s\n", n4 * 16.0 / (ms / 10) / 1e6);
    }
    return 0;
}
