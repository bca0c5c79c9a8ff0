// Microbenchmark (GPU box): the ISTFT's read stream without arithmetic.  One wave walks consecutive rows of 1025 complex64 (8 200 B,
// 8-byte aligned only) and reads each row either as 16 x 8-byte loads per lane in the kernel's order (bins tf + 64 i ascending,
// M - tf - 64 i descending) + bin M/2, or as 8 x 16-byte loads (lane l: bins 2 l, 2 l + 1 of each 128-bin block) + the last bin.
//   hipcc --offload-arch=gfx950 -O3 scripts/loadpat.hip -o probe/loadpat && probe/loadpat
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE, int AHEAD> __global__ __launch_bounds__(64) void k(const f2* __restrict__ D, float* out, int rows_per_strip, int n_strips) {
    const int strip = blockIdx.x;
    if (strip >= n_strips) return;
    const int l = threadIdx.x;
    const f2* row = D + (size_t)strip * rows_per_strip * 1025;
    f2 acc = {0.f, 0.f};
    if (MODE == 0) {
        f2 a[AHEAD][17];
        for (int r = 0; r < AHEAD; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { a[r][i] = row[r * 1025 + l + 64 * i]; a[r][8 + i] = row[r * 1025 + 1024 - l - 64 * i]; }
            a[r][16] = row[r * 1025 + 512];
        }
        for (int r = 0; r < rows_per_strip; ++r) {
            const int s = r % AHEAD;
#pragma unroll
            for (int i = 0; i < 17; ++i) acc += a[s][i];
            if (r + AHEAD < rows_per_strip) {
                const f2* p = row + (size_t)(r + AHEAD) * 1025;
#pragma unroll
                for (int i = 0; i < 8; ++i) { a[s][i] = p[l + 64 * i]; a[s][8 + i] = p[1024 - l - 64 * i]; }
                a[s][16] = p[512];
            }
            // stand-in for the frame's arithmetic: keep the loop from collapsing
            asm volatile("s_nop 0" ::: "memory");
        }
    } else {
        f4 a[AHEAD][8]; f2 e[AHEAD];
        for (int r = 0; r < AHEAD; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[r][i] = *reinterpret_cast<const f4*>(row + r * 1025 + 2 * l + 128 * i);
            e[r] = row[r * 1025 + 1024];
        }
        for (int r = 0; r < rows_per_strip; ++r) {
            const int s = r % AHEAD;
#pragma unroll
            for (int i = 0; i < 8; ++i) { acc.x += a[s][i].x + a[s][i].z; acc.y += a[s][i].y + a[s][i].w; }
            acc += e[s];
            if (r + AHEAD < rows_per_strip) {
                const f2* p = row + (size_t)(r + AHEAD) * 1025;
#pragma unroll
                for (int i = 0; i < 8; ++i) a[s][i] = *reinterpret_cast<const f4*>(p + 2 * l + 128 * i);
                e[s] = p[1024];
            }
            asm volatile("s_nop 0" ::: "memory");
        }
    }
    if (acc.x == 123.456f) out[blockIdx.x * 64 + l] = acc.x + acc.y;
}
template <int MODE, int AHEAD> void run(const char* name, const f2* D, float* out, int rows, int rows_per_strip) {
    const int n_strips = rows / rows_per_strip;
    hipLaunchKernelGGL((k<MODE, AHEAD>), dim3(n_strips), dim3(64), 0, 0, D, out, rows_per_strip, n_strips);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<MODE, AHEAD>), dim3(n_strips), dim3(64), 0, 0, D, out, rows_per_strip, n_strips);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10; if (ms < best) best = ms;
    }
    const double bytes = (double)n_strips * rows_per_strip * 8200.0;
    printf("%-34s rows/strip %4d (%5d strips): %.3f ms  %6.0f GB/s\n", name, rows_per_strip, n_strips, best, bytes / best / 1e6);
}
int main() {
    const int rows = 330752;
    f2* D; float* out;
    (void)hipMalloc(&D, (size_t)rows * 8200 + 4096); (void)hipMalloc(&out, 1 << 24);
    (void)hipMemset(D, 0, (size_t)rows * 8200);
    for (int rps : {81, 162, 40}) {
        run<0, 1>("8-byte loads, 1 row ahead", D, out, rows, rps);
        run<0, 2>("8-byte loads, 2 rows ahead", D, out, rows, rps);
        run<1, 1>("16-byte loads, 1 row ahead", D, out, rows, rps);
        run<1, 2>("16-byte loads, 2 rows ahead", D, out, rows, rps);
    }
    return 0;
}
