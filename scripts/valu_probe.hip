// Microbenchmark (GPU box): issue cost of the VALU forms the FFT kernels are made of, wave64 on one SIMD, at 1 / 2 / 3 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_probe.hip -o probe/valu_probe && probe/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int MODE> __global__ __launch_bounds__(64) void k(float* out, int iters, long long* cyc) {
    f2 a0 = {1.f + threadIdx.x, 2.f}, a1 = {3.f, 4.f}, a2 = {5.f, 6.f}, a3 = {7.f, 8.f}, a4 = {1.5f, 2.5f}, a5 = {3.5f, 4.5f}, a6 = {5.5f, 6.5f}, a7 = {7.5f, 8.5f};
    const f2 c = {1.0001f, 0.9999f};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // v_pk_fma_f32, 8 independent chains
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                              "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (MODE == 1) {  // v_pk_add_f32
            REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                              "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (MODE == 2) {  // v_fma_f32 (one half)
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                              "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                              : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(c.x));)
        } else if (MODE == 3) {  // v_add_f32
            REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                              "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                              : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(c.x));)
        } else if (MODE == 4) {  // v_mov_b32
            REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0"
                              : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x));)
        } else if (MODE == 5) {  // v_mov_b64
            REP8(asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %4\n v_mov_b64 %4, %5\n v_mov_b64 %5, %6\n v_mov_b64 %6, %7\n v_mov_b64 %7, %0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 6) {  // v_cndmask_b32 (vcc)
            REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                              "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc"
                              : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) :: "vcc");)
        } else if (MODE == 7) {  // v_pk_mul_f32
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                              "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (MODE == 8) {  // dependent chain of v_pk_fma_f32 (latency)
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n"
                              "v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1"
                              : "+v"(a0) : "v"(c));)
        } else if (MODE == 9) {  // v_add_u32
            unsigned u0 = threadIdx.x, u1 = 1, u2 = 2, u3 = 3;
            REP8(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));)
            a0.x += (float)(u0 + u1 + u2 + u3);
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    float s = a0.x + a0.y + a1.x + a1.y + a2.x + a2.y + a3.x + a3.y + a4.x + a4.y + a5.x + a5.y + a6.x + a6.y + a7.x + a7.y;
    if (s == 123.456f) out[threadIdx.x] = s;
}
template <int MODE> void run(const char* name) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 8 * 4096);
    for (int wps : {1, 2, 3}) {
        const int iters = 2000, grid = 256 * 4 * wps;  // 64-thread blocks: one wave each; 4 * wps per CU
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, 10, cyc);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, iters, cyc);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        long long h[8]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        const double n = 64.0 * iters;
        printf("%-22s %d wave/SIMD: %.3f ms  s_memtime ticks/instr/wave %.2f  -> per SIMD %.2f ticks/instr; wall: %.2f ns/instr/SIMD\n", name, wps, ms, h[0] / n, h[0] / n / wps, ms * 1e6 / (n * wps));
    }
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<0>("v_pk_fma_f32"); run<1>("v_pk_add_f32"); run<7>("v_pk_mul_f32"); run<2>("v_fma_f32"); run<3>("v_add_f32"); run<4>("v_mov_b32"); run<5>("v_mov_b64"); run<6>("v_cndmask_b32");
    run<9>("v_add_u32"); run<8>("v_pk_fma_f32 dependent");
    return 0;
}
