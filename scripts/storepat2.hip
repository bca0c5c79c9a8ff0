// Build: hipcc --offload-arch=gfx950 -O3 -o probe/storepat2 scripts/storepat2.hip ; run on the GPU box.
// Store-pattern microbenchmark, round 2 (development, not part of the product): which decomposition of the STFT
// output stream -- per frame 2 048 B of PCM read, one 8 200-byte row (1025 complex64) written -- does the chip
// absorb fastest?  No arithmetic, no LDS traffic; the loads of row r+1 are issued before the stores of row r
// (the kernels prefetch one frame ahead), so a store never waits for its own iteration's loads.
//
//   MODE 0  strip:   one wave walks `iters` consecutive rows; 16 x dwordx2 (bins k ascending / M-k descending) + mid bin
//   MODE 1  quad:    the W waves of a workgroup take W ADJACENT rows per step (one contiguous W x 8200-byte burst)
//   MODE 2  quadblk: as 1, but the W-row block is written by all threads as 16-byte-aligned dwordx4 chunks of 4 KB
//   MODE 3  strip16: one wave per strip, each row as 8 aligned 1-KB dwordx4 chunks (+ the 8..16-byte tail)
//   MODE 4  strip, dwordx2 ascending only
// Knobs: waves per workgroup, rows per strip, LDS pad (bounds the resident waves per CU), block -> XCD remap
// (consecutive strips of a clip on ONE XCD instead of round-robin), non-temporal stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int ROWC = 1025;      // complex64 per row
constexpr int ROWB = ROWC * 8;  // 8200
constexpr long long CLIP = 661500;

template <bool NT, class V> __device__ __forceinline__ void st(V* p, V v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

template <int MODE, bool NT, bool READ> __global__ __launch_bounds__(256) void k(char* __restrict__ out, const float* __restrict__ in, int rows_per_clip, int iters, int n_clips, int xcd_remap, int delay) {
    extern __shared__ char pad_lds[];
    const int W = blockDim.x / 64, wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    int b = blockIdx.x;
    const int nb = gridDim.x;
    if (xcd_remap && nb % 8 == 0) b = (b % 8) * (nb / 8) + b / 8;
    f2 v = {(float)lane, (float)b};
    if (MODE == 1 || MODE == 2) {
        const int rows_per_wg = iters * W;
        const int wg_per_clip = (rows_per_clip + rows_per_wg - 1) / rows_per_wg;
        const int clip = b / wg_per_clip, part = b % wg_per_clip;
        if (clip >= n_clips) return;
        const int row0 = part * rows_per_wg;
        // reads: W * 512 samples per step, as 2 x dwordx4 per thread when W == 4 (else 8 dwords per lane of each wave)
        f4 cur0 = {0, 0, 0, 0}, cur1 = {0, 0, 0, 0};
        for (int s = 0; s < iters; ++s) {
            const int r = row0 + s * W;
            if (r >= rows_per_clip) break;
            f4 n0 = cur0, n1 = cur1;
            if (READ) {
                const float* src = in + (size_t)clip * CLIP + (size_t)(r + W) * 512;
                n0 = *reinterpret_cast<const f4*>(src + 4 * threadIdx.x);
                n1 = *reinterpret_cast<const f4*>(src + 4 * (threadIdx.x + blockDim.x));
            }
            v.x += cur0.x + cur1.y;
            if (MODE == 1) {
                const int row = r + wave;
                if (row < rows_per_clip) {
                    f2* rp = reinterpret_cast<f2*>(out + ((size_t)clip * rows_per_clip + row) * ROWB);
                    f2* pk = rp + lane;
                    f2* pm = rp + (1024 - lane);
#pragma unroll
                    for (int i = 0; i < 8; ++i) { st<NT>(&pk[i * 64], v); st<NT>(&pm[-i * 64], v); }
                    if (lane == 0) st<NT>(&rp[512], v);
                }
            } else {
                const int nrows = (rows_per_clip - r) < W ? (rows_per_clip - r) : W;
                const size_t b0 = ((size_t)clip * rows_per_clip + r) * ROWB, b1 = b0 + (size_t)nrows * ROWB;
                const size_t a0 = b0 & ~(size_t)15;
                f4 w = {v.x, v.y, v.x, v.y};
                for (size_t o = a0 + 16 * (size_t)threadIdx.x; o < b1; o += 16 * (size_t)blockDim.x) st<NT>(reinterpret_cast<f4*>(out + o), w);
            }
            cur0 = n0;
            cur1 = n1;
        }
        return;
    }
    // strip modes: wave-private strips
    const int strips_per_clip = (rows_per_clip + iters - 1) / iters;
    const long long sid = (long long)b * W + wave;
    const int clip = (int)(sid / strips_per_clip), part = (int)(sid % strips_per_clip);
    if (clip >= n_clips) return;
    float cur[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const int row = part * iters + it;
        if (row >= rows_per_clip) break;
        float nx[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) nx[c] = cur[c];
        if (READ) {
            const float* src = in + (size_t)clip * CLIP + (size_t)(row + 1) * 512 + lane;
#pragma unroll
            for (int c = 0; c < 8; ++c) nx[c] = src[c * 64];
        }
        v.x += cur[0] + cur[7];
        // stand-in for the FFT of the frame: `delay` dependent FMAs (~4 cycles each for a lone wave)
        for (int d = 0; d < delay; ++d) v.y = __builtin_fmaf(v.y, 1.0000001f, 1e-9f);
        char* rb = out + ((size_t)clip * rows_per_clip + row) * ROWB;
        f2* rp = reinterpret_cast<f2*>(rb);
        if (MODE == 0) {
            f2* pk = rp + lane;
            f2* pm = rp + (1024 - lane);
#pragma unroll
            for (int i = 0; i < 8; ++i) { st<NT>(&pk[i * 64], v); st<NT>(&pm[-i * 64], v); }
            if (lane == 0) st<NT>(&rp[512], v);
        } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) st<NT>(&rp[lane + i * 64], v);
            if (lane == 0) st<NT>(&rp[1024], v);
        } else {  // MODE 3
            char* a0 = reinterpret_cast<char*>(reinterpret_cast<size_t>(rb) & ~(size_t)15);
            f4 w = {v.x, v.y, v.x, v.y};
#pragma unroll
            for (int i = 0; i < 8; ++i) st<NT>(reinterpret_cast<f4*>(a0 + 1024 * i + 16 * lane), w);
            if (lane == 0) st<NT>(reinterpret_cast<f4*>(a0 + 8192), w);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) cur[c] = nx[c];
    }
}

struct Cfg {
    int mode, W, iters, lds_pad, remap, nt, read, delay;
};

template <int MODE, bool NT, bool READ> float run_t(char* out, const float* in, const Cfg& c, int batch, int rows) {
    int grid;
    if (MODE == 1 || MODE == 2) {
        const int rows_per_wg = c.iters * c.W;
        grid = batch * ((rows + rows_per_wg - 1) / rows_per_wg);
    } else {
        const long long strips = (long long)batch * ((rows + c.iters - 1) / c.iters);
        grid = (int)((strips + c.W - 1) / c.W);
    }
    grid = (grid + 7) / 8 * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, NT, READ>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE, NT, READ>), dim3(grid), dim3(64 * c.W), c.lds_pad, 0, out, in, rows, c.iters, batch, c.remap, c.delay);
    hipEventRecord(e0);
    const int reps = 8;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<MODE, NT, READ>), dim3(grid), dim3(64 * c.W), c.lds_pad, 0, out, in, rows, c.iters, batch, c.remap, c.delay);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (hipGetLastError() != hipSuccess) return -1.f;
    return ms / reps;
}

template <int MODE> float run_m(char* out, const float* in, const Cfg& c, int batch, int rows) {
    if (c.nt) return c.read ? run_t<MODE, true, true>(out, in, c, batch, rows) : run_t<MODE, true, false>(out, in, c, batch, rows);
    return c.read ? run_t<MODE, false, true>(out, in, c, batch, rows) : run_t<MODE, false, false>(out, in, c, batch, rows);
}

float run(char* out, const float* in, const Cfg& c, int batch, int rows) {
    switch (c.mode) {
        case 0: return run_m<0>(out, in, c, batch, rows);
        case 1: return run_m<1>(out, in, c, batch, rows);
        case 2: return run_m<2>(out, in, c, batch, rows);
        case 3: return run_m<3>(out, in, c, batch, rows);
        default: return run_m<4>(out, in, c, batch, rows);
    }
}

int main(int argc, char** argv) {
    const int batch = 256, rows = 1292;
    char* out;
    float* in;
    hipMalloc(&out, (size_t)batch * rows * ROWB + (1 << 20));
    hipMalloc(&in, (size_t)batch * CLIP * 4 + (8 << 20));
    hipMemset(in, 0, (size_t)batch * CLIP * 4 + (8 << 20));
    const double bytes_rw = (double)batch * rows * (ROWB + 2048), bytes_w = (double)batch * rows * ROWB;
    const char* names[] = {"strip", "quad", "quadblk", "strip16", "strip-asc"};
    auto report = [&](const Cfg& c) {
        const float ms = run(out, in, c, batch, rows);
        const double by = c.read ? bytes_rw : bytes_w;
        printf("%-9s W=%d iters=%3d ldspad=%3dK remap=%d nt=%d read=%d delay=%4d : %.4f ms  %.0f GB/s\n", names[c.mode], c.W, c.iters, c.lds_pad / 1024, c.remap, c.nt, c.read, c.delay, ms, by / ms / 1e6);
        fflush(stdout);
    };
    if (argc > 1 && argv[1][0] == 'p') {  // "power [seconds] [memset]": the best strip form (or a plain hipMemset of the output) in a loop, for rocm-smi sampling from a second shell
        const double secs = argc > 2 ? atof(argv[2]) : 8.0;
        const bool ms_only = argc > 3;
        const Cfg c{0, 1, 162, (160 * 1024 / 12) & ~255, 1, 0, 1, 0};
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        float total = 0.f, last = 0.f;
        while (total < secs * 1e3f) {
            if (ms_only) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                for (int r = 0; r < 8; ++r) hipMemsetAsync(out, 0, (size_t)batch * rows * ROWB, 0);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&last, e0, e1); last /= 8;
                hipEventDestroy(e0); hipEventDestroy(e1);
            } else {
                last = run(out, in, c, batch, rows);
            }
            hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&total, a, b);
        }
        printf("%s: %.4f ms  %.0f GB/s\n", ms_only ? "hipMemsetAsync of the output" : "strip W=1 iters=162 12 waves/CU remap=1 read=1", last, (ms_only ? bytes_w : bytes_rw) / last / 1e6);
        return 0;
    }
    if (argc > 1) {  // "conc": the two store forms against resident waves per CU, rows per strip and a little compute between rows
        for (int mode : {0, 3})
            for (int nt : {0, 1}) {
                if (mode == 0 && nt) continue;
                for (int remap : {0, 1})
                    for (int wpc : {8, 12, 16, 24, 32})
                        for (int iters : {81, 162}) report({mode, 1, iters, wpc >= 32 ? 0 : ((160 * 1024 / wpc) & ~255), remap, nt, 1, 0});
            }
        for (int mode : {0, 3})
            for (int delay : {10, 30, 60, 100}) report({mode, 1, 108, (160 * 1024 / 12) & ~255, 1, mode == 3, 1, delay});
        return 0;
    }
    // 1. decomposition x rows per strip, default residency
    for (int read : {1, 0})
        for (int mode : {0, 3, 4}) {
            for (int W : {1, 4})
                for (int iters : {27, 81, 162, 323}) report({mode, W, iters, 0, 0, 0, read});
        }
    for (int read : {1, 0})
        for (int mode : {1, 2})
            for (int W : {4, 8, 16})
                for (int iters : {8, 21, 41, 81, 323}) {
                    if (iters * W > 1400 && iters != 8) continue;
                    report({mode, W, iters, 0, 0, 0, read});
                }
    // 2. residency: LDS pad so that 8 / 12 / 16 / 24 / 32 waves fit a CU
    for (int mode : {0, 3})
        for (int wpc : {4, 8, 12, 16, 24, 32}) report({mode, 1, 81, (160 * 1024 / wpc) & ~255, 0, 0, 1});
    for (int mode : {1, 2})
        for (int wgpc : {1, 2, 3, 4, 6, 8}) report({mode, 4, 21, (160 * 1024 / wgpc) & ~255, 0, 0, 1});
    // 3. XCD remap and non-temporal stores
    for (int mode : {0, 1, 2, 3})
        for (int remap : {0, 1})
            for (int nt : {0, 1}) report({mode, mode == 0 || mode == 3 ? 1 : 4, mode == 0 || mode == 3 ? 81 : 21, 0, remap, nt, 1});
    return 0;
}
