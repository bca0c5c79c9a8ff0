// Microbenchmark (GPU box): LDS throughput of ds_read_b64 pairs vs ds_read2_b64, ds_write_b64 pairs vs ds_write2_b64, per CU.
//   hipcc --offload-arch=gfx950 -O3 scripts/lds_probe.hip -o probe/lds_probe && probe/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (float)i;
    __syncthreads();
    const unsigned a = threadIdx.x * 8;  // 8-byte slots, lane-contiguous: conflict-free
    float2 acc = make_float2(0.f, 0.f);
    for (int it = 0; it < iters; ++it) {
        float2 v0, v1, v2, v3, v4, v5, v6, v7;
        if (MODE == 0) {
            asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:2048\n ds_read_b64 %2, %8 offset:4096\n ds_read_b64 %3, %8 offset:6144\n"
                         "ds_read_b64 %4, %8 offset:8192\n ds_read_b64 %5, %8 offset:10240\n ds_read_b64 %6, %8 offset:12288\n ds_read_b64 %7, %8 offset:14336\n s_waitcnt lgkmcnt(0)"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a) : "memory");
        } else if (MODE == 1) {
            float4 w0, w1, w2, w3;
            asm volatile("ds_read2st64_b64 %0, %4 offset0:0 offset1:4\n ds_read2st64_b64 %1, %4 offset0:8 offset1:12\n ds_read2st64_b64 %2, %4 offset0:16 offset1:20\n ds_read2st64_b64 %3, %4 offset0:24 offset1:28\n s_waitcnt lgkmcnt(0)"
                         : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3) : "v"(a) : "memory");
            v0 = make_float2(w0.x, w0.y); v1 = make_float2(w0.z, w0.w); v2 = make_float2(w1.x, w1.y); v3 = make_float2(w1.z, w1.w);
            v4 = make_float2(w2.x, w2.y); v5 = make_float2(w2.z, w2.w); v6 = make_float2(w3.x, w3.y); v7 = make_float2(w3.z, w3.w);
        } else if (MODE == 2) {  // b128 reads of the same bytes (lane-contiguous 16-byte slots)
            float4 w0, w1, w2, w3;
            const unsigned b = threadIdx.x * 16;
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:4096\n ds_read_b128 %2, %4 offset:8192\n ds_read_b128 %3, %4 offset:12288\n s_waitcnt lgkmcnt(0)"
                         : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3) : "v"(b) : "memory");
            v0 = make_float2(w0.x, w0.y); v1 = make_float2(w0.z, w0.w); v2 = make_float2(w1.x, w1.y); v3 = make_float2(w1.z, w1.w);
            v4 = make_float2(w2.x, w2.y); v5 = make_float2(w2.z, w2.w); v6 = make_float2(w3.x, w3.y); v7 = make_float2(w3.z, w3.w);
        } else if (MODE == 3) {  // writes: 8 x ds_write_b64
            v0 = v1 = v2 = v3 = v4 = v5 = v6 = v7 = acc;
            asm volatile("ds_write_b64 %8, %0\n ds_write_b64 %8, %1 offset:2048\n ds_write_b64 %8, %2 offset:4096\n ds_write_b64 %8, %3 offset:6144\n"
                         "ds_write_b64 %8, %4 offset:8192\n ds_write_b64 %8, %5 offset:10240\n ds_write_b64 %8, %6 offset:12288\n ds_write_b64 %8, %7 offset:14336\n s_waitcnt lgkmcnt(0)"
                         :: "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7), "v"(a) : "memory");
        } else {  // writes: 4 x ds_write2st64_b64
            v0 = v1 = v2 = v3 = v4 = v5 = v6 = v7 = acc;
            asm volatile("ds_write2st64_b64 %8, %0, %1 offset0:0 offset1:4\n ds_write2st64_b64 %8, %2, %3 offset0:8 offset1:12\n ds_write2st64_b64 %8, %4, %5 offset0:16 offset1:20\n ds_write2st64_b64 %8, %6, %7 offset0:24 offset1:28\n s_waitcnt lgkmcnt(0)"
                         :: "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7), "v"(a) : "memory");
        }
        acc.x += v0.x + v1.x + v2.x + v3.x + v4.x + v5.x + v6.x + v7.x;
        acc.y += v0.y + v1.y + v2.y + v3.y + v4.y + v5.y + v6.y + v7.y;
    }
    if (acc.x == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + lds[threadIdx.x];
}
template <int MODE> void run(const char* name, int blocks_per_cu) {
    float* out; hipMalloc(&out, 256 * 256 * 16 * sizeof(float));
    const int iters = 20000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * 256 * 64.0 * iters;  // 64 bytes per thread and iteration
    printf("%-28s %d blocks/CU: %8.3f ms  %7.1f TB/s aggregate  (%.1f B/clk/CU at 2.4 GHz)\n", name, blocks_per_cu, ms, bytes / ms / 1e9, bytes / ms / 1e-3 / 256 / 2.4e9);
    hipFree(out);
}
int main() {
    for (int b : {1, 2}) {
        run<0>("8 x ds_read_b64", b);
        run<1>("4 x ds_read2st64_b64", b);
        run<2>("4 x ds_read_b128", b);
        run<3>("8 x ds_write_b64", b);
        run<4>("4 x ds_write2st64_b64", b);
    }
    return 0;
}
