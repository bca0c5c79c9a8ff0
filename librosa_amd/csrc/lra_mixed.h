// lra_mixed.h -- fused forward transform for frame lengths that are not powers of two: n_fft = 2 M with M = 2^a 3^b 5^c 7^d
// (400 = the 25 ms frame of 16 kHz speech front ends, 320, 480, 640, 800, 960, 1200, 1600, ...).
//
// Reference semantics: librosa/core/spectrum.py:57-391 (stft: centre padding :287, framing :330-350, window multiply + rfft of the
// frames :380-390), :2920-3015 (_spectrogram: |X|^power), librosa/feature/spectral.py:2158-2160 (melspectrogram: basis . |X|^power).
//
// Round 3 served these sizes through frame_window_kernel -> rocFFT -> power_transpose_kernel -> mel_apply_kernel: the 4 x overlapped
// frames were materialised in HBM (0.9-1.1 TB/s of algorithmic traffic).  Here ONE launch reads the PCM, transforms in LDS and writes
// the complex spectrum, |X|^p or the mel bands; nothing else touches HBM.
//
// Structure (deliberately different from the power-of-two kernels, whose one-wave-per-frame register tiling has no counterpart for
// M = 200): a workgroup of 256 threads owns F consecutive frames of one clip at a time, resident in LDS as M complex points each
// (z[n] = x[2n] + i x[2n+1] times the window pair: the usual even/odd packing), and every stage hands its work items -- butterflies,
// split bins, mel bands -- to the threads by a flat index, so no stage cares how M factors against the wave size.  Stages, separated
// by workgroup barriers: (1) frame load + window (np.pad index fold for centred frames at the clip's edges), (2) one Stockham pass per
// radix in {8, 5, 4, 3, 2} ping-ponging between two LDS buffers, twiddles W_M^t from an LDS table, (3) Hermitian split with W_N^k and the
// epilogue.  All index arithmetic is on compile-time constants (N is a template parameter): divisions are multiplications.
//
// Self-contained (threadIdx / __shared__ / __syncthreads only): the whole kernel also runs in tests/hostsim/postsim.cpp, one OS thread
// per lane, against the oracle.
#pragma once

#ifndef LRA_POSTSIM
#include <hip/hip_runtime.h>
#endif

// wave priority per stage (s_setprio; experiment switch, profiles/r04_experiments.md 10): L = frame loads, P = passes, S = split, E = mel reduce; -1 = none
#ifndef LRA_MIXED_PRIO_L
#define LRA_MIXED_PRIO_L -1
#endif
#ifndef LRA_MIXED_PRIO_P
#define LRA_MIXED_PRIO_P -1
#endif
#ifndef LRA_MIXED_PRIO_S
#define LRA_MIXED_PRIO_S -1
#endif
#ifndef LRA_MIXED_PRIO_E
#define LRA_MIXED_PRIO_E -1
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define LRA_MIXED_SETPRIO(p) do { if ((p) >= 0) __builtin_amdgcn_s_setprio((p) >= 0 ? (p) : 0); } while (0)
#else
#define LRA_MIXED_SETPRIO(p) ((void)0)
#endif

namespace lra {
namespace mixed {

template <class T> struct alignas(2 * sizeof(T)) cpx {
    T x, y;
};
template <class T> __device__ __forceinline__ cpx<T> mkc(T x, T y) { cpx<T> r; r.x = x; r.y = y; return r; }
template <class T> __device__ __forceinline__ cpx<T> add(cpx<T> a, cpx<T> b) { return mkc<T>(a.x + b.x, a.y + b.y); }
template <class T> __device__ __forceinline__ cpx<T> sub(cpx<T> a, cpx<T> b) { return mkc<T>(a.x - b.x, a.y - b.y); }
// a b + c in one rounding, spelled out wherever a sum of two products could be contracted either way: the stages are unrolled over their work items
// (round 6), and a frame's bits must not depend on which copy of the body it lands in (a clip sharded by frames equals the unsharded call bit for bit)
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <class T> __device__ __forceinline__ cpx<T> mul(cpx<T> a, cpx<T> b) { return mkc<T>(fma_t(-a.y, b.y, a.x * b.x), fma_t(a.y, b.x, a.x * b.y)); }
template <class T> __device__ __forceinline__ cpx<T> mul_mi(cpx<T> a) { return mkc<T>(a.y, -a.x); }  // a * (-i)
template <class T> __device__ __forceinline__ cpx<T> scale(cpx<T> a, T k) { return mkc<T>(a.x * k, a.y * k); }

// ---- factorisation of M (compile time) ---------------------------------------------------------------------------------------
struct Radices {
    int r[12];
    int n;
};
constexpr Radices factor(int M) {
    Radices f{};
    f.n = 0;
    // radix-8 passes first, then one 4 or 2 for what is left of the power of two, then the fives and threes
    while (M % 8 == 0) { f.r[f.n++] = 8; M /= 8; }
    if (M % 4 == 0) { f.r[f.n++] = 4; M /= 4; }
    if (M % 2 == 0) { f.r[f.n++] = 2; M /= 2; }
    while (M % 5 == 0) { f.r[f.n++] = 5; M /= 5; }
    while (M % 3 == 0) { f.r[f.n++] = 3; M /= 3; }
    while (M % 7 == 0) { f.r[f.n++] = 7; M /= 7; }  // (round 6: 882 and 1764 samples = 20 / 40 ms at 44.1 kHz, M = 3^2 7^2 and 2 3^2 7^2)
    if (M != 1) f.n = 0;  // another prime: not served here
    return f;
}
constexpr bool supported(int N) { return N >= 12 && N % 2 == 0 && N / 2 <= 4096 && factor(N / 2).n > 0; }
constexpr int stride_before(const Radices& f, int p) {
    int s = 1;
    for (int q = 0; q < p; ++q) s *= f.r[q];
    return s;
}
// frames per workgroup: two ping-pong buffers of M complex points per frame + the W_M table in ~24 KB, i.e. six or seven workgroups per
// CU: every stage ends in a workgroup barrier and starts with loads, so what hides the latencies is OTHER workgroups in other stages.
// Measured on the MI355X, n_fft 400 / hop 160 / 80 mels, 256 x 30 s (profiles/r04_experiments.md section 5): 16 frames per workgroup (51 KB,
// three workgroups per CU) 1.39 ms, 12: 1.16, 8: 1.08, 6: 0.94, 4: 0.99; 64- or 128-thread workgroups 1.45-2.5 ms.
#ifndef LRA_MIXED_NT
#define LRA_MIXED_NT 256
#endif
#ifndef LRA_MIXED_FMAX
#define LRA_MIXED_FMAX 8
#endif
#ifndef LRA_MIXED_LDS_KB
#define LRA_MIXED_LDS_KB 24
#endif
// Among the frame counts the budget allows (and down to half of that), the one whose stages waste the fewest lanes: a stage of `items` work items per
// frame runs ceil(F items / NT) rounds of the whole workgroup, each costing about the same whether its last round is full or not -- at n_fft = 400 the
// budget's 7 frames put 280 radix-5 butterflies into 2 rounds of 256 lanes, 6 frames put 240 into one (round 6, after the global loads stopped being what
// the kernel waited for: the vector pipe is busy 0.76 of the time; measured 400 / 160: stft 0.518 -> 0.505 ms).  Weights ~ vector instructions per work item.
#ifndef LRA_MIXED_FTUNE
#define LRA_MIXED_FTUNE 1
#endif
constexpr int stage_rounds_cost(int M, int F, int nt) {
    const Radices f = factor(M);
    int cost = ((F * M + nt - 1) / nt) * 15 + ((F * (M / 2 + 1) + nt - 1) / nt) * 40;  // frame load, Hermitian split
    for (int p = 0; p < f.n; ++p) {
        const int r = f.r[p], w = r == 8 ? 100 : (r == 7 ? 110 : (r == 5 ? 75 : (r == 2 ? 20 : 40)));
        cost += ((F * (M / r) + nt - 1) / nt) * w;
    }
    return cost;
}
template <class T, int N> constexpr int frames_per_group() {
    constexpr int M = N / 2;
    int f = (int)((LRA_MIXED_LDS_KB * 1024 - M * 2 * (int)sizeof(T)) / (2 * M * 2 * (int)sizeof(T)));
    f = f < 1 ? 1 : (f > LRA_MIXED_FMAX ? LRA_MIXED_FMAX : f);
    if (LRA_MIXED_FTUNE) {
        int best = f;
        for (int c = f - 1; c >= 1 && 2 * c >= f; --c)
            if ((long long)stage_rounds_cost(M, c, LRA_MIXED_NT) * best < (long long)stage_rounds_cost(M, best, LRA_MIXED_NT) * c) best = c;  // cost per frame
        f = best;
    }
    return f;
}
template <class T, int N> constexpr int lds_bytes() { return (2 * frames_per_group<T, N>() * (N / 2) + N / 2) * 2 * (int)sizeof(T); }

// The mel stage's band table in LDS (round 6): a (band, frame) work item needs its band's (c0, len, off) and then its weights -- two dependent global round trips
// per round of work items, at the very end of a workgroup's life where nothing hides them.  Up to 256 bands and 1 536 weights (one and six loads per thread,
// issued with the frame stage's loads) go behind the W_M table for up to 9 KB; larger banks keep the cached reads.  Measured (256 x 30 s, mel): 400 / 160 / 80
// 0.70-0.73 -> 0.615 ms, 480 / 120 / 40 1.07 -> 0.93, 800 / 200 / 80 1.05 -> 0.94, 1200 / 300 / 128 1.49 -> 1.34; a cap of 768 misses the last two, 2 560 is 1 % slower.
#ifndef LRA_MIXED_MEL_WAVES
#define LRA_MIXED_MEL_WAVES 6   // waves per SIMD the forward kernels are compiled for (80 VGPRs): what the LDS budget keeps resident
#endif
#ifndef LRA_MIXED_MEL_NNZ
#define LRA_MIXED_MEL_NNZ 1536
#endif
constexpr int kMelLdsMaxBands = 256, kMelLdsMaxNnz = LRA_MIXED_MEL_NNZ;
template <class T> constexpr int mel_lds_bytes(int n_mels, int nnz) { return nnz * (int)sizeof(T) + 3 * n_mels * (int)sizeof(int); }
inline bool mel_lds_fits(int n_mels, int nnz) { return n_mels >= 1 && n_mels <= kMelLdsMaxBands && nnz >= 1 && nnz <= kMelLdsMaxNnz; }
enum { MIXED_COMPLEX = 0, MIXED_POWER = 1, MIXED_MEL = 2 };
constexpr int NT = LRA_MIXED_NT;  // threads per workgroup

template <class T> struct Args {
    const T* y;            // [batch][y_stride]
    long long y_stride, n;
    int n_frames, hop, pad, pad_mode;  // pad = n_fft / 2 when centred; pad_mode: 0 constant, 1 reflect, 2 edge, 3 symmetric (np.pad)
    const T* win;          // [N] window padded to n_fft (NOT halved)
    const cpx<T>* tw_m;    // [M]     W_M^t = exp(-2 pi i t / M)
    const cpx<T>* tw_n;    // [M + 1] W_N^k = exp(-2 pi i k / N)
    cpx<T>* D;             // MIXED_COMPLEX: [batch][n_frames][M + 1]
    T* S;                  // MIXED_POWER:   [batch][n_frames][M + 1]
    T* Mel;                // MIXED_MEL:     [batch][n_mels][n_frames]
    int power_mode;        // 1: |X|, 2: |X|^2, 0: |X|^power
    T power;
    const int* mel_c0;     // band m covers bins [c0, c0 + len), weights mel_val[off + i]
    const int* mel_len;
    const int* mel_off;
    const T* mel_val;
    int n_mels;
    int groups_per_clip;   // ceil(n_frames / F)
    int mel_nnz;           // MIXED_MEL: > 0 = the band table (3 n_mels descriptors + mel_nnz weights) is staged in LDS behind the W_M table (the host adds
                           // mel_lds_bytes<T>(n_mels, mel_nnz) to the launch's LDS); 0 = read through the caches
};

// sample g of a clip of n samples under np.pad's modes; matches lra::pad_index (lra_common.h) and np.pad incl. repeated reflection
template <class T> __device__ __forceinline__ T fetch(const T* __restrict__ yb, long long g, long long n, int mode) {
    if (g >= 0 && g < n) return yb[g];
    if (n <= 0 || mode == 0) return (T)0;
    if (mode == 2 || n == 1) return yb[g < 0 ? 0 : n - 1];
    const long long lo = mode == 1 ? 0 : -1, hi = mode == 1 ? 2 * (n - 1) : 2 * n - 1;
    while (g < 0 || g >= n) g = g < 0 ? lo - g : hi - g;
    return yb[g];
}

// ---- small forward DFTs, natural order in and out ---------------------------------------------------------------------------------
template <int R, class T> struct Dft;
template <class T> struct Dft<2, T> {
    static __device__ __forceinline__ void run(cpx<T>* v) {
        const cpx<T> a = v[0], b = v[1];
        v[0] = add(a, b);
        v[1] = sub(a, b);
    }
};
template <class T> struct Dft<3, T> {
    static __device__ __forceinline__ void run(cpx<T>* v) {
        const T s = (T)0.86602540378443864676;  // sin(2 pi / 3)
        const cpx<T> t1 = add(v[1], v[2]), d = scale(mul_mi(sub(v[1], v[2])), s);  // -i s (b - c)
        const cpx<T> m = mkc<T>(fma_t((T)-0.5, t1.x, v[0].x), fma_t((T)-0.5, t1.y, v[0].y));
        v[0] = add(v[0], t1);
        v[1] = add(m, d);
        v[2] = sub(m, d);
    }
};
template <class T> struct Dft<4, T> {
    static __device__ __forceinline__ void run(cpx<T>* v) {
        const cpx<T> a = add(v[0], v[2]), b = sub(v[0], v[2]), c = add(v[1], v[3]), d = mul_mi(sub(v[1], v[3]));
        v[0] = add(a, c);
        v[1] = add(b, d);
        v[2] = sub(a, c);
        v[3] = sub(b, d);
    }
};
template <class T> struct Dft<5, T> {
    static __device__ __forceinline__ void run(cpx<T>* v) {
        const T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410;  // cos(2 pi / 5), cos(4 pi / 5)
        const T s1 = (T)0.95105651629515357212, s2 = (T)0.58778525229247312917;   // sin(2 pi / 5), sin(4 pi / 5)
        const cpx<T> t1 = add(v[1], v[4]), t2 = add(v[2], v[3]), t3 = sub(v[1], v[4]), t4 = sub(v[2], v[3]);
        const cpx<T> m1 = mkc<T>(fma_t(c2, t2.x, fma_t(c1, t1.x, v[0].x)), fma_t(c2, t2.y, fma_t(c1, t1.y, v[0].y)));
        const cpx<T> m2 = mkc<T>(fma_t(c1, t2.x, fma_t(c2, t1.x, v[0].x)), fma_t(c1, t2.y, fma_t(c2, t1.y, v[0].y)));
        const cpx<T> n1 = mul_mi(mkc<T>(fma_t(s1, t3.x, s2 * t4.x), fma_t(s1, t3.y, s2 * t4.y)));    // -i (s1 t3 + s2 t4)
        const cpx<T> n2 = mul_mi(mkc<T>(fma_t(s2, t3.x, -(s1 * t4.x)), fma_t(s2, t3.y, -(s1 * t4.y))));  // -i (s2 t3 - s1 t4)
        v[0] = add(v[0], add(t1, t2));
        v[1] = add(m1, n1);
        v[4] = sub(m1, n1);
        v[2] = add(m2, n2);
        v[3] = sub(m2, n2);
    }
};
template <class T> struct Dft<7, T> {
    static __device__ __forceinline__ void run(cpx<T>* v) {
        // X[k] = a_k - i b_k, X[7 - k] = a_k + i b_k with a_k = v0 + sum_n cos(2 pi n k / 7) (v[n] + v[7 - n]), b_k = sum_n sin(2 pi n k / 7) (v[n] - v[7 - n]), n = 1 .. 3
        const T c1 = (T)0.62348980185873353053, c2 = (T)-0.22252093395631440429, c3 = (T)-0.90096886790241912624;  // cos(2 pi / 7), cos(4 pi / 7), cos(6 pi / 7)
        const T n1 = (T)0.78183148246802980871, n2 = (T)0.97492791218182360702, n3 = (T)0.43388373911755812048;   // sin(2 pi / 7), sin(4 pi / 7), sin(6 pi / 7)
        const cpx<T> t1 = add(v[1], v[6]), t2 = add(v[2], v[5]), t3 = add(v[3], v[4]), s1 = sub(v[1], v[6]), s2 = sub(v[2], v[5]), s3 = sub(v[3], v[4]);
        const cpx<T> a1 = mkc<T>(fma_t(c3, t3.x, fma_t(c2, t2.x, fma_t(c1, t1.x, v[0].x))), fma_t(c3, t3.y, fma_t(c2, t2.y, fma_t(c1, t1.y, v[0].y))));
        const cpx<T> a2 = mkc<T>(fma_t(c1, t3.x, fma_t(c3, t2.x, fma_t(c2, t1.x, v[0].x))), fma_t(c1, t3.y, fma_t(c3, t2.y, fma_t(c2, t1.y, v[0].y))));
        const cpx<T> a3 = mkc<T>(fma_t(c2, t3.x, fma_t(c1, t2.x, fma_t(c3, t1.x, v[0].x))), fma_t(c2, t3.y, fma_t(c1, t2.y, fma_t(c3, t1.y, v[0].y))));
        const cpx<T> b1 = mul_mi(mkc<T>(fma_t(n3, s3.x, fma_t(n2, s2.x, n1 * s1.x)), fma_t(n3, s3.y, fma_t(n2, s2.y, n1 * s1.y))));     // -i (n1 s1 + n2 s2 + n3 s3)
        const cpx<T> b2 = mul_mi(mkc<T>(fma_t(-n1, s3.x, fma_t(-n3, s2.x, n2 * s1.x)), fma_t(-n1, s3.y, fma_t(-n3, s2.y, n2 * s1.y))));  // -i (n2 s1 - n3 s2 - n1 s3)
        const cpx<T> b3 = mul_mi(mkc<T>(fma_t(n2, s3.x, fma_t(-n1, s2.x, n3 * s1.x)), fma_t(n2, s3.y, fma_t(-n1, s2.y, n3 * s1.y))));   // -i (n3 s1 - n1 s2 + n2 s3)
        v[0] = add(v[0], add(t1, add(t2, t3)));
        v[1] = add(a1, b1); v[6] = sub(a1, b1);
        v[2] = add(a2, b2); v[5] = sub(a2, b2);
        v[3] = add(a3, b3); v[4] = sub(a3, b3);
    }
};
template <class T> struct Dft<8, T> {
    static __device__ __forceinline__ void run(cpx<T>* v) {
        const T h = (T)0.70710678118654752440;
        cpx<T> e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
        Dft<4, T>::run(e);
        Dft<4, T>::run(o);
        const cpx<T> o1 = scale(mkc<T>(o[1].x + o[1].y, o[1].y - o[1].x), h);   // o1 * W8^1 = o1 (1 - i) / sqrt 2
        const cpx<T> o2 = mul_mi(o[2]);                                          // o2 * W8^2 = -i o2
        const cpx<T> o3 = scale(mkc<T>(o[3].y - o[3].x, -(o[3].x + o[3].y)), h);  // o3 * W8^3 = o3 (-1 - i) / sqrt 2
        v[0] = add(e[0], o[0]); v[4] = sub(e[0], o[0]);
        v[1] = add(e[1], o1);   v[5] = sub(e[1], o1);
        v[2] = add(e[2], o2);   v[6] = sub(e[2], o2);
        v[3] = add(e[3], o3);   v[7] = sub(e[3], o3);
    }
};

#ifdef LRA_POSTSIM
#define LRA_MIXED_DYN_LDS(ptr) char* ptr = reinterpret_cast<char*>(g_postsim_dyn_lds)
#else
#define LRA_MIXED_DYN_LDS(ptr) extern __shared__ __align__(16) char lra_mixed_dyn_lds[]; char* ptr = lra_mixed_dyn_lds
#endif

// One Stockham pass of radix R over the F frames in `src` (stride S = product of the earlier radices): butterfly b of a frame reads
// src[b + j M / R], multiplies input j by W_{S R}^{(b mod S) j} = W_M^{(b mod S) j M / (S R)} and writes output j to
// dst[(b - b mod S) R + (b mod S) + j S].  Work items (frame, butterfly) are dealt to the threads by a flat index.
// PITCH: complex slots between consecutive frames of a buffer (M, or M + 1 where the epilogue wants whole spectra of M + 1 bins per frame)
template <class T, int N, int R, int S, int F, int PITCH = N / 2> __device__ __forceinline__ void pass(const cpx<T>* src, cpx<T>* dst, const cpx<T>* twm, int frames) {
    constexpr int M = N / 2, NB = M / R, TWS = M / (S * R);
    // (trip count known at compile time, the partial group's bound as a predicate: unrolled, so that the LDS reads of one work item are in flight
    // while the previous one is in its butterfly -- round 6; the runtime-bounded loop ran its items strictly one after the other)
    constexpr int IT = (F * NB + NT - 1) / NT;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int w = (int)threadIdx.x + it * NT;
        if (w >= frames * NB) break;
        const int f = w / NB, b = w - f * NB, k = b % S;
        const cpx<T>* s = src + f * PITCH + b;
        cpx<T> v[R];
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = s[j * NB];
        if (S > 1) {
#pragma unroll
            for (int j = 1; j < R; ++j) v[j] = mul(v[j], twm[k * j * TWS]);
        }
        Dft<R, T>::run(v);
        cpx<T>* d = dst + f * PITCH + (b - k) * R + k;
#pragma unroll
        for (int j = 0; j < R; ++j) d[j * S] = v[j];
    }
}

// the passes of factor(M), unrolled at compile time; returns (through the pointer swap) the buffer that holds the result
template <class T, int N, int P, int F, int PITCH = N / 2> struct Passes {
    static __device__ __forceinline__ void run(cpx<T>*& a, cpx<T>*& b, const cpx<T>* twm, int frames) {
        constexpr Radices f = factor(N / 2);
        if constexpr (P < f.n) {
            pass<T, N, f.r[P], stride_before(f, P), F, PITCH>(a, b, twm, frames);
            __syncthreads();
            cpx<T>* t = a; a = b; b = t;
            Passes<T, N, P + 1, F, PITCH>::run(a, b, twm, frames);
        }
    }
};

template <class T> __device__ __forceinline__ T spec_pow(cpx<T> x, int mode, T p) {
    const T m2 = fma_t(x.y, x.y, x.x * x.x);
    if (mode == 2) return m2;
    const T m = sqrt(m2);
    if (mode == 1) return m;
    return pow(m, p);
}

// grid = batch * groups_per_clip workgroups of NT threads; dynamic LDS = lds_bytes<T, N>()
template <class T, int N, int MODE> __global__ __launch_bounds__(NT, (sizeof(T) == 4 && N <= 3200) ? LRA_MIXED_MEL_WAVES : 1) void mixed_stft_kernel(Args<T> a) {
    constexpr int M = N / 2, F = frames_per_group<T, N>();
    LRA_MIXED_DYN_LDS(lds);
    cpx<T>* buf0 = reinterpret_cast<cpx<T>*>(lds);
    cpx<T>* buf1 = buf0 + F * M;
    cpx<T>* twm = buf1 + F * M;
    const int clip = (int)(blockIdx.x / (unsigned)a.groups_per_clip), group = (int)(blockIdx.x % (unsigned)a.groups_per_clip);
    const int f0 = group * F;
    const int frames = a.n_frames - f0 < F ? a.n_frames - f0 : F;
    const T* __restrict__ yb = a.y + (long long)clip * a.y_stride;
    // (1) W_M table -> LDS; frames -> buf0 as window-multiplied sample pairs.  Every global load of the stage -- the table, the samples, the window
    // pairs and the split step's W_N^k (which depend on nothing) -- is issued before the first use: with runtime loop bounds hipcc ran the items one after
    // the other, a global round trip each (round 6: 6-10 of them per workgroup at n_fft 3200; scripts/mixed_counters.sh: waves waiting 0.71-0.75 of their
    // cycles, the vector pipe busy 0.39-0.59).
    LRA_MIXED_SETPRIO(LRA_MIXED_PRIO_L);
    constexpr int HP = M / 2 + 1;         // pairs (k, M - k) per frame
    constexpr int IT0 = (M + NT - 1) / NT, IT1 = (F * M + NT - 1) / NT, IT3 = (F * HP + NT - 1) / NT;
    const cpx<T>* __restrict__ win2 = reinterpret_cast<const cpx<T>*>(a.win);
    // sample pairs as ONE 8-byte load where every pair of the launch is aligned (even hop, padding and clip pitch, aligned base: uniform)
    const bool pair_aligned = ((a.hop | a.pad) & 1) == 0 && (a.y_stride & 1) == 0 && (reinterpret_cast<unsigned long long>(a.y) & (2 * sizeof(T) - 1)) == 0;
    cpx<T> tw0[IT0], xs[IT1], wv[IT1], twn[IT3];
#pragma unroll
    for (int it = 0; it < IT0; ++it) {
        const int t = (int)threadIdx.x + it * NT;
        tw0[it] = a.tw_m[t < M ? t : 0];
    }
#pragma unroll
    for (int it = 0; it < IT1; ++it) {
        const int w = (int)threadIdx.x + it * NT;
        if (w < frames * M) {
            const int f = w / M, i = w - f * M;
            const long long g = (long long)(f0 + f) * a.hop - a.pad + 2 * i;  // clip position of the pair's first sample
            if (g >= 0 && g + 1 < a.n) {
                if (pair_aligned) xs[it] = *reinterpret_cast<const cpx<T>*>(yb + g);
                else xs[it] = mkc<T>(yb[g], yb[g + 1]);
            } else {
                xs[it] = mkc<T>(fetch<T>(yb, g, a.n, a.pad_mode), fetch<T>(yb, g + 1, a.n, a.pad_mode));
            }
            wv[it] = win2[i];
        }
    }
#pragma unroll
    for (int it = 0; it < IT3; ++it) twn[it] = a.tw_n[((int)threadIdx.x + it * NT) % HP];
    // MIXED_MEL with the band table in LDS: weights [mel_nnz], then c0 / len / off [n_mels each], behind the W_M table
    T* lval = reinterpret_cast<T*>(twm + M);
    int* lc0 = reinterpret_cast<int*>(lval + a.mel_nnz);
    int* llen = lc0 + a.n_mels;
    int* loff = llen + a.n_mels;
    const bool mel_lds = MODE == MIXED_MEL && a.mel_nnz > 0;
    if (mel_lds) {
        constexpr int ITV = (kMelLdsMaxNnz + NT - 1) / NT, ITB = (kMelLdsMaxBands + NT - 1) / NT;
        T mv[ITV];
        int b0[ITB], b1[ITB], b2[ITB];
#pragma unroll
        for (int it = 0; it < ITV; ++it) {
            const int i = (int)threadIdx.x + it * NT;
            mv[it] = a.mel_val[i < a.mel_nnz ? i : 0];
        }
#pragma unroll
        for (int it = 0; it < ITB; ++it) {
            const int m = (int)threadIdx.x + it * NT, mm = m < a.n_mels ? m : 0;
            b0[it] = a.mel_c0[mm];
            b1[it] = a.mel_len[mm];
            b2[it] = a.mel_off[mm];
        }
#pragma unroll
        for (int it = 0; it < ITV; ++it) {
            const int i = (int)threadIdx.x + it * NT;
            if (i < a.mel_nnz) lval[i] = mv[it];
        }
#pragma unroll
        for (int it = 0; it < ITB; ++it) {
            const int m = (int)threadIdx.x + it * NT;
            if (m < a.n_mels) { lc0[m] = b0[it]; llen[m] = b1[it]; loff[m] = b2[it]; }
        }
    }
#pragma unroll
    for (int it = 0; it < IT0; ++it) {
        const int t = (int)threadIdx.x + it * NT;
        if (t < M) twm[t] = tw0[it];
    }
#pragma unroll
    for (int it = 0; it < IT1; ++it) {
        const int w = (int)threadIdx.x + it * NT;
        if (w < frames * M) buf0[w] = mkc<T>(xs[it].x * wv[it].x, xs[it].y * wv[it].y);
    }
    __syncthreads();
    // (2) M-point complex FFT of every frame
    LRA_MIXED_SETPRIO(LRA_MIXED_PRIO_P);
    cpx<T>* src = buf0;
    cpx<T>* dst = buf1;
    Passes<T, N, 0, F>::run(src, dst, twm, frames);
    LRA_MIXED_SETPRIO(LRA_MIXED_PRIO_S);
    // (3) Hermitian split.  With E = (Z[k] + conj Z[M-k]) / 2, O = (Z[k] - conj Z[M-k]) / 2 and P = W_N^k O:
    //       X[k] = E - i P,   X[M-k] = conj(E) - i conj(P)      (W_N^{M-k} = -conj W_N^k),
    // so one work item per pair (k, M - k), k = 0 .. M/2, reads its two points once and writes both bins (k = 0: the DC and Nyquist bins
    // from Z[0]; k = M/2, M even: one self-paired bin).
    T* prow = reinterpret_cast<T*>(dst);  // MIXED_MEL: the power rows go to the buffer the FFT no longer needs ((M + 1) values per frame <= 2 M)
#pragma unroll
    for (int it = 0; it < IT3; ++it) {
        const int w = (int)threadIdx.x + it * NT;
        if (w >= frames * HP) break;
        const int f = w / HP, k = w - f * HP;
        const cpx<T> zk = src[f * M + k], zr = src[f * M + (k == 0 ? 0 : M - k)];
        const cpx<T> e = mkc<T>((T)0.5 * (zk.x + zr.x), (T)0.5 * (zk.y - zr.y)), o = mkc<T>((T)0.5 * (zk.x - zr.x), (T)0.5 * (zk.y + zr.y));
        const cpx<T> pw = mul(o, twn[it]);
        cpx<T> xk = mkc<T>(e.x + pw.y, e.y - pw.x), xm = mkc<T>(e.x - pw.y, -e.y - pw.x);
        if (k == 0) { xk.y = (T)0; xm.y = (T)0; }  // DC and Nyquist: purely real by construction (rounding leaves -0 / +0 differences otherwise)
        const long long row = ((long long)clip * a.n_frames + f0 + f) * (M + 1);
        const bool two = 2 * k != M;  // (k = M/2 pairs with itself: xk == xm up to the sign of zero)
        if (MODE == MIXED_COMPLEX) {
            a.D[row + k] = xk;
            if (two) a.D[row + M - k] = xm;
        } else if (MODE == MIXED_POWER) {
            a.S[row + k] = spec_pow<T>(xk, a.power_mode, a.power);
            if (two) a.S[row + M - k] = spec_pow<T>(xm, a.power_mode, a.power);
        } else {
            prow[f * (M + 1) + k] = spec_pow<T>(xk, a.power_mode, a.power);
            if (two) prow[f * (M + 1) + M - k] = spec_pow<T>(xm, a.power_mode, a.power);
        }
    }
    if (MODE == MIXED_MEL) {
        __syncthreads();
        LRA_MIXED_SETPRIO(LRA_MIXED_PRIO_E);
        // mel[m][f] = sum_i val[off_m + i] P[f][c0_m + i]: a work item per (band, frame), frames fastest -- the F frames of a band are
        // consecutive in the output
        for (int w = (int)threadIdx.x; w < a.n_mels * F; w += NT) {
            const int m = w / F, f = w - m * F;
            if (f >= frames) continue;
            if (mel_lds) {  // band table in LDS: no global round trip at all
                const int c0 = lc0[m], len = llen[m];
                const T* val = lval + loff[m];
                const T* p = prow + f * (M + 1) + c0;
                T acc = (T)0;
                for (int i = 0; i < len; ++i) acc += val[i] * p[i];
                a.Mel[((long long)clip * a.n_mels + m) * a.n_frames + f0 + f] = acc;
                continue;
            }
            const int c0 = a.mel_c0[m], len = a.mel_len[m];
            const T* __restrict__ val = a.mel_val + a.mel_off[m];
            const T* p = prow + f * (M + 1) + c0;
            // the band's weights four at a time ahead of their use, the sum still in bin order (one dependent global load per bin was what this loop waited
            // for); whole batches with plain consecutive loads (hipcc merges them), then ONE partial batch that loads the band's last weight again instead and
            // skips the surplus products (clamped indices in every batch cost the merged loads: 3200 / 800 / 128 1.63 -> 2.76 ms)
            T acc = (T)0;
            int i = 0;
            for (; i + 4 <= len; i += 4) {
                T vv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) vv[q] = val[i + q];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc += vv[q] * p[i + q];
            }
            if (i < len) {
                T vv[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) vv[q] = val[i + q < len ? i + q : len - 1];
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (i + q < len) acc += vv[q] * p[i + q];
            }
            a.Mel[((long long)clip * a.n_mels + m) * a.n_frames + f0 + f] = acc;
        }
    }
}

// =====================================================================================================================================
// One octave of the constant-Q / variable-Q recursion in ONE launch: librosa/core/constantq.py:1197-1223 (__cqt_response: stft with a
// rectangular window, fft_basis.dot(D)), :1116-1118 (division by sqrt(filter length)), :1168-1194 (__trim_stack: the octave's rows of
// the stacked result, frames beyond the common count dropped).  The frame transform is the kernel above (n_fft is a power of two here:
// filters.wavelet pads to one); the spectrum of a frame never leaves LDS -- round 3 wrote it to HBM (85 MB per octave for 64 x 30 s
// clips) and read it back in cqt_project_kernel.  The projection follows that kernel operation by operation: CSR rows walked in order,
// one rounding per operation (no contraction), the scaling as a float64 multiplication by the reciprocal.
template <class T> struct CqtArgs {
    const T* y;
    long long y_stride, n;
    int hop, pad, pad_mode;
    const cpx<T>* tw_m;
    const cpx<T>* tw_n;
    const int* row_ptr;        // CSR over the octave's filters; rows row0 .. row0 + n_rows - 1 are used
    const int* col;
    const cpx<T>* val;
    const double* sqrt_len;    // [n_rows] or nullptr (scale=False)
    cpx<T>* out;               // [batch][n_frames][n_total], this octave at columns bin0 ...
    int n_frames, n_total, bin0, row0, n_rows;
    int groups_per_clip;
    unsigned int* nonfinite_flag;  // raised when a frame's DC bin is not finite (a non-finite sample in it), or nullptr
};
template <class T, int N> constexpr int cqt_frames_per_group() {
    constexpr int MP = N / 2 + 1;
    int f = (int)((LRA_MIXED_LDS_KB * 1024 - (N / 2) * 2 * (int)sizeof(T)) / (2 * MP * 2 * (int)sizeof(T)));
    return f < 1 ? 1 : (f > LRA_MIXED_FMAX ? LRA_MIXED_FMAX : f);
}
// (Staging the octave's CSR entries in LDS -- 12 KB for up to 1024 of them -- was measured and is slower: 62.9 against 54.7 us per octave of 64 x 30 s clips;
// the LDS it takes costs resident workgroups, which is what hides this kernel's latencies.)
template <class T, int N> constexpr int cqt_lds_bytes() { return (2 * cqt_frames_per_group<T, N>() * (N / 2 + 1) + N / 2) * 2 * (int)sizeof(T); }

#ifdef LRA_POSTSIM
#define LRA_MIXED_FLAG_OR(ptr) (*(ptr) |= 1u)
#else
#define LRA_MIXED_FLAG_OR(ptr) atomicOr((ptr), 1u)
#endif

#pragma clang fp contract(off)
template <class T> __device__ __forceinline__ cpx<T> cmadd_exact(cpx<T> acc, cpx<T> a, cpx<T> b) {  // acc + a b, as scipy.sparse's complex wrapper evaluates it
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LRA_MIXED_NO_PK_ASM)
    // float: the same six roundings as four packed instructions on the (re, im) register pairs -- hipcc's own packing of the lines below takes six
    // (a sum AND a difference of the two product pairs, then a move to splice their halves); this product is the octave kernel's inner loop
    if constexpr (sizeof(T) == 4) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 va = __builtin_bit_cast(f2, a), vb = __builtin_bit_cast(f2, b), vacc = __builtin_bit_cast(f2, acc);
        f2 p1, p2, sm, r;
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(p1) : "v"(va), "v"(vb));  // (a.x b.x, a.x b.y)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(p2) : "v"(va), "v"(vb));  // (a.y b.y, a.y b.x)
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(sm) : "v"(p1), "v"(p2));                   // (a.x b.x - a.y b.y, a.x b.y + a.y b.x)
        asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(vacc), "v"(sm));
        return __builtin_bit_cast(cpx<T>, r);
    }
#endif
    const T re = a.x * b.x - a.y * b.y;
    const T im = a.x * b.y + a.y * b.x;
    return mkc<T>(acc.x + re, acc.y + im);
}
#pragma clang fp contract(fast)

template <class T, int N> __device__ __forceinline__ void cqt_octave_body(const CqtArgs<T>& a, unsigned block, char* lds) {
    constexpr int M = N / 2, MP = M + 1, F = cqt_frames_per_group<T, N>(), HP = M / 2 + 1;
    cpx<T>* buf0 = reinterpret_cast<cpx<T>*>(lds);
    cpx<T>* buf1 = buf0 + F * MP;
    cpx<T>* twm = buf1 + F * MP;
    const int clip = (int)(block / (unsigned)a.groups_per_clip), group = (int)(block % (unsigned)a.groups_per_clip);
    const int f0 = group * F;
    const int frames = a.n_frames - f0 < F ? a.n_frames - f0 : F;
    const T* __restrict__ yb = a.y + (long long)clip * a.y_stride;
    // every global load of the frame stage (table, samples, the split step's W_N^k) ahead of its first use, as in mixed_stft_kernel above
    constexpr int IT0 = (M + NT - 1) / NT, IT1 = (F * M + NT - 1) / NT, IT3 = (F * HP + NT - 1) / NT;
    const bool pair_aligned = ((a.hop | a.pad) & 1) == 0 && (a.y_stride & 1) == 0 && (reinterpret_cast<unsigned long long>(a.y) & (2 * sizeof(T) - 1)) == 0;
    cpx<T> tw0[IT0], xs[IT1], twn[IT3];
#pragma unroll
    for (int it = 0; it < IT0; ++it) {
        const int t = (int)threadIdx.x + it * NT;
        tw0[it] = a.tw_m[t < M ? t : 0];
    }
    // frames -> buf0 (rectangular window: window="ones", constantq.py:1201)
#pragma unroll
    for (int it = 0; it < IT1; ++it) {
        const int w = (int)threadIdx.x + it * NT;
        if (w < frames * M) {
            const int f = w / M, i = w - f * M;
            const long long g = (long long)(f0 + f) * a.hop - a.pad + 2 * i;
            if (g >= 0 && g + 1 < a.n) {
                if (pair_aligned) xs[it] = *reinterpret_cast<const cpx<T>*>(yb + g);
                else xs[it] = mkc<T>(yb[g], yb[g + 1]);
            } else {
                xs[it] = mkc<T>(fetch<T>(yb, g, a.n, a.pad_mode), fetch<T>(yb, g + 1, a.n, a.pad_mode));
            }
        }
    }
#pragma unroll
    for (int it = 0; it < IT3; ++it) twn[it] = a.tw_n[((int)threadIdx.x + it * NT) % HP];
#pragma unroll
    for (int it = 0; it < IT0; ++it) {
        const int t = (int)threadIdx.x + it * NT;
        if (t < M) twm[t] = tw0[it];
    }
#pragma unroll
    for (int it = 0; it < IT1; ++it) {
        const int w = (int)threadIdx.x + it * NT;
        if (w < frames * M) {
            const int f = w / M, i = w - f * M;
            buf0[f * MP + i] = xs[it];
        }
    }
    __syncthreads();
    cpx<T>* src = buf0;
    cpx<T>* dst = buf1;
    Passes<T, N, 0, F, MP>::run(src, dst, twm, frames);
    // Hermitian split by pairs, whole spectra (M + 1 bins) -> dst
#pragma unroll
    for (int it = 0; it < IT3; ++it) {
        const int w = (int)threadIdx.x + it * NT;
        if (w >= frames * HP) break;
        const int f = w / HP, k = w - f * HP;
        const cpx<T> zk = src[f * MP + k], zr = src[f * MP + (k == 0 ? 0 : M - k)];
        const cpx<T> e = mkc<T>((T)0.5 * (zk.x + zr.x), (T)0.5 * (zk.y - zr.y)), o = mkc<T>((T)0.5 * (zk.x - zr.x), (T)0.5 * (zk.y + zr.y));
        const cpx<T> pw = mul(o, twn[it]);
        cpx<T> xk = mkc<T>(e.x + pw.y, e.y - pw.x), xm = mkc<T>(e.x - pw.y, -e.y - pw.x);
        if (k == 0) {
            xk.y = (T)0;
            xm.y = (T)0;
            if (a.nonfinite_flag && !(xk.x - xk.x == (T)0)) LRA_MIXED_FLAG_OR(a.nonfinite_flag);
        }
        dst[f * MP + k] = xk;
        if (2 * k != M) dst[f * MP + M - k] = xm;
    }
    __syncthreads();
    // projection: a work item per (frame, row), rows fastest (consecutive output elements).  A row's entries are fetched eight at a time (columns and
    // values: sixteen independent global loads, then eight LDS reads) ahead of the eight multiply-adds, which still run in CSR order with one rounding per
    // operation -- the walk of round 4 paid one global round trip per entry (~80 entries per row).
    for (int w = (int)threadIdx.x; w < frames * a.n_rows; w += NT) {
        const int f = w / a.n_rows, r = w - f * a.n_rows;
        const cpx<T>* d = dst + f * MP;
        cpx<T> acc = mkc<T>((T)0, (T)0);
        const int j1 = a.row_ptr[a.row0 + r + 1];
        int j = a.row_ptr[a.row0 + r];
        for (; j + 8 <= j1; j += 8) {
            int cc[8];
            cpx<T> vv[8], dd[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { cc[q] = a.col[j + q]; vv[q] = a.val[j + q]; }
#pragma unroll
            for (int q = 0; q < 8; ++q) dd[q] = d[cc[q]];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc = cmadd_exact(acc, vv[q], dd[q]);
        }
        if (j < j1) {  // ONE partial batch (it loads the row's last entry again instead and skips the surplus products): no entry-by-entry tail of dependent loads
            int cc[7];
            cpx<T> vv[7], dd[7];
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                const int jq = j + q < j1 ? j + q : j1 - 1;
                cc[q] = a.col[jq];
                vv[q] = a.val[jq];
            }
#pragma unroll
            for (int q = 0; q < 7; ++q) dd[q] = d[cc[q]];
#pragma unroll
            for (int q = 0; q < 7; ++q)
                if (j + q < j1) acc = cmadd_exact(acc, vv[q], dd[q]);
        }
        if (a.sqrt_len) {
            const double scl = 1.0 / a.sqrt_len[r];
            acc = mkc<T>((T)((double)acc.x * scl), (T)((double)acc.y * scl));
        }
        a.out[((long long)clip * a.n_frames + f0 + f) * a.n_total + a.bin0 + r] = acc;
    }
}

template <class T, int N> __global__ __launch_bounds__(NT) void mixed_cqt_kernel(CqtArgs<T> a) {
    LRA_MIXED_DYN_LDS(lds);
    cqt_octave_body<T, N>(a, blockIdx.x, lds);
}

// Several octaves of one transform in ONE launch (round 5): they share the frame length and the frame grid (every octave has n_frames frames, its hop halving with
// its signal), differ in signal, hop and basis rows, and are independent once their decimated signals exist.  As seven launches their ~50 us floors (a short
// latency-bound kernel each: global loads, five barrier-separated stages, a CSR walk) add up to 0.4 of the transform's 0.66 ms; side by side they overlap.
constexpr int kCqtMaxMerged = 8;
template <class T> struct CqtMultiArgs {
    CqtArgs<T> oct[kCqtMaxMerged];
    int n_oct;
    unsigned blocks_per_octave;  // batch x groups_per_clip (the same for every octave)
};
template <class T, int N> __global__ __launch_bounds__(NT) void mixed_cqt_multi_kernel(CqtMultiArgs<T> m) {
    LRA_MIXED_DYN_LDS(lds);
    const unsigned o = blockIdx.x / m.blocks_per_octave;
    if ((int)o >= m.n_oct) return;
    cqt_octave_body<T, N>(m.oct[o], blockIdx.x - o * m.blocks_per_octave, lds);
}

// =====================================================================================================================================
// Inverse: librosa/core/spectrum.py:394-626 (irfft of the columns :598, window, overlap-add in frame order :593-603 / :629-643,
// division by the window sum-square :606-624) for the same frame lengths, one launch.
//
// A workgroup owns the output samples of G consecutive hops of one clip -- padded positions [g G hop, (g + 1) G hop), the clip's last
// group also the tail up to the last frame's end -- and transforms every frame that reaches into them: its own G frames and the
// ceil(N / hop) - 1 frames before (recomputed by the neighbouring group too: no carry between workgroups, no atomics).  Per frame:
// Hermitian un-split of X[0 .. M] into conj Z' (pairs (k, M - k); the imaginary parts of X[0] and X[M] are ignored, as pocketfft's c2r
// does), the forward passes above (FFT of the conjugate = conjugate of the inverse), window / N; the windowed frames stay in LDS as N
// reals each.  Then every output sample gathers its <= ceil(N / hop) contributions in INCREASING frame order -- the reference's
// accumulation order -- and is multiplied by its normalisation factor (1 / wss where wss > tiny, else 1: lra_api.hip builds the table).
template <class T> struct InvArgs {
    const cpx<T>* D;            // [batch][d_batch_stride]: frames d_frame_stride apart, M + 1 bins each
    long long d_batch_stride, d_frame_stride;
    int n_used, hop, drop;      // frames [0, n_used) contribute; drop = n_fft / 2 when centred
    const T* win_scaled;        // [N] window / N
    const cpx<T>* tw_m;         // [M]     W_M^t
    const cpx<T>* tw_n;         // [M + 1] W_N^k
    const T* norm;              // [out_len] normalisation factors
    T* y;                       // [batch][y_stride]
    long long y_stride, out_len;
    int groups_per_clip, group_hops, halo;  // G = group_hops own frames per group, halo = ceil(N / hop) - 1 earlier frames
};
// Frames resident per workgroup (own + halo).  The halo frames are recomputed work, so the group must be long against the halo (the host
// takes this kernel only when own >= 2 halo, lra_api.hip), but LDS per workgroup is residency (measured, 400 / 160, 256 x 30 s: 24 KB 1.00 ms,
// 32 KB 1.02, 48 KB 1.21, 64 KB 1.55; 800 / 200 at 64 KB: 2.89 ms against 2.79 for the rocFFT path): 24 KB, or 32 KB where that is what holds
// nine frames (six own + the three halo frames of a 4 x overlap); larger frames fail the host's rule and take mixed_irfft_kernel + ola_gather_kernel
// (round 6: with this kernel's loads batched, 48 / 64 KB tiers were measured too -- 480 / 120 at 48 KB 1.73 ms, 800 / 200 at 64 KB 2.51 -- and lose to
// that pair: 1.62 / 1.63 ms).
template <class T, int N> constexpr int inv_frames_max() {
    constexpr int M = N / 2;
    constexpr int per_frame = 2 * M * 2 * (int)sizeof(T), table = M * 2 * (int)sizeof(T);
    int f = 0;
    for (int kb : {24, 32}) {
        f = (kb * 1024 - table) / per_frame;
        if (f >= 9) break;
    }
    return f < 2 ? 2 : (f > 24 ? 24 : f);
}
template <class T, int N> constexpr int inv_lds_bytes() { return (2 * inv_frames_max<T, N>() * (N / 2) + N / 2) * 2 * (int)sizeof(T); }

// Stage (1) of both inverse kernels: the W_M table -> LDS and the Hermitian un-split of `frames` spectra (X0: the first one, d_frame_stride apart, M + 1 bins each) into
// buf0[f][k] = conj Z'[k],  Z'[k] = E' + i O',  E' = X[k] + conj X[M-k],  O' = (X[k] - conj X[M-k]) conj W_N^k (the imaginary parts of X[0] and X[M] are ignored, as pocketfft's
// c2r does).  The table, both bins of every pair and its W_N^k are loaded ahead of the first use (compile-time trip counts, round 6: see mixed_stft_kernel).
template <class T, int N, int FR>
__device__ __forceinline__ void unsplit_stage(cpx<T>* buf0, cpx<T>* twm, const cpx<T>* __restrict__ X0, long long d_frame_stride, int frames, const cpx<T>* __restrict__ tw_m,
                                              const cpx<T>* __restrict__ tw_n) {
    constexpr int M = N / 2, HP = M / 2 + 1;
    constexpr int IT0 = (M + NT - 1) / NT, IT1 = (FR * HP + NT - 1) / NT;
    cpx<T> tw0[IT0], xk[IT1], xm[IT1], wc[IT1];
#pragma unroll
    for (int it = 0; it < IT0; ++it) {
        const int t = (int)threadIdx.x + it * NT;
        tw0[it] = tw_m[t < M ? t : 0];
    }
#pragma unroll
    for (int it = 0; it < IT1; ++it) {
        const int w = (int)threadIdx.x + it * NT;
        if (w < frames * HP) {
            const int f = w / HP, k = w - f * HP;
            const cpx<T>* __restrict__ X = X0 + (long long)f * d_frame_stride;
            xk[it] = X[k];
            xm[it] = X[M - k];
            wc[it] = tw_n[k];
        }
    }
#pragma unroll
    for (int it = 0; it < IT0; ++it) {
        const int t = (int)threadIdx.x + it * NT;
        if (t < M) twm[t] = tw0[it];
    }
#pragma unroll
    for (int it = 0; it < IT1; ++it) {
        const int w = (int)threadIdx.x + it * NT;
        if (w >= frames * HP) break;
        const int f = w / HP, k = w - f * HP;
        cpx<T> x0 = xk[it], x1 = xm[it];
        if (k == 0) { x0.y = (T)0; x1.y = (T)0; }
        const cpx<T> e = mkc<T>(x0.x + x1.x, x0.y - x1.y), d = mkc<T>(x0.x - x1.x, x0.y + x1.y);
        const cpx<T> o = mkc<T>(fma_t(d.y, wc[it].y, d.x * wc[it].x), fma_t(-d.x, wc[it].y, d.y * wc[it].x));  // d * conj(w)
        // conj Z'[k] = conj(E') - i conj(O') = (e.x - o.y, -e.y - o.x);  conj Z'[M-k] = E' - i O' = (e.x + o.y, e.y - o.x)
        buf0[f * M + k] = mkc<T>(e.x - o.y, -e.y - o.x);
        if (k > 0 && 2 * k != M) buf0[f * M + M - k] = mkc<T>(e.x + o.y, e.y - o.x);
    }
}

// The inverse real transform ALONE (round 6): frames too long for the gather kernel below (its own frames must outnumber the recomputed halo in 64 KB of LDS) used to go
// spec_pack_kernel -> rocFFT C2R -> ola_gather_kernel; this launch replaces the first two -- un-split, the forward passes on the conjugate, and every frame's N samples
// (N irfft, what rocFFT's unnormalised C2R leaves: the gather kernel multiplies by window / N) written once, 8 bytes per lane -- with the forward kernel's geometry.
template <class T> struct IrArgs {
    const cpx<T>* D;            // clip c of this launch at D + c * d_batch_stride; frames d_frame_stride apart, M + 1 bins each
    long long d_batch_stride, d_frame_stride;
    int n_used;
    const cpx<T>* tw_m;
    const cpx<T>* tw_n;
    T* frames;                  // [clip][n_used][N]
    int groups_per_clip;        // ceil(n_used / frames_per_group)
};
template <class T, int N> __global__ __launch_bounds__(NT) void mixed_irfft_kernel(IrArgs<T> a) {
    constexpr int M = N / 2, F = frames_per_group<T, N>();
    LRA_MIXED_DYN_LDS(lds);
    cpx<T>* buf0 = reinterpret_cast<cpx<T>*>(lds);
    cpx<T>* buf1 = buf0 + F * M;
    cpx<T>* twm = buf1 + F * M;
    const long long clip = (long long)(blockIdx.x / (unsigned)a.groups_per_clip);
    const int f0 = (int)(blockIdx.x % (unsigned)a.groups_per_clip) * F;
    const int frames = a.n_used - f0 < F ? a.n_used - f0 : F;
    unsplit_stage<T, N, F>(buf0, twm, a.D + clip * a.d_batch_stride + (long long)f0 * a.d_frame_stride, a.d_frame_stride, frames, a.tw_m, a.tw_n);
    __syncthreads();
    cpx<T>* src = buf0;
    cpx<T>* dst = buf1;
    Passes<T, N, 0, F>::run(src, dst, twm, frames);  // Y = FFT(conj Z') = conj(M z): sample pair m of a frame = (Y.x, -Y.y)
    cpx<T>* __restrict__ out = reinterpret_cast<cpx<T>*>(a.frames + (clip * a.n_used + f0) * (long long)N);
    constexpr int IT = (F * M + NT - 1) / NT;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int w = (int)threadIdx.x + it * NT;
        if (w >= frames * M) break;
        const cpx<T> v = src[w];
        out[w] = mkc<T>(v.x, -v.y);  // (frames of a group are consecutive: position w of the group is pair w of its first frame's row)
    }
}

template <class T, int N> __global__ __launch_bounds__(NT) void mixed_istft_kernel(InvArgs<T> a) {
    constexpr int M = N / 2, FMAX = inv_frames_max<T, N>();
    LRA_MIXED_DYN_LDS(lds);
    cpx<T>* buf0 = reinterpret_cast<cpx<T>*>(lds);
    cpx<T>* buf1 = buf0 + FMAX * M;
    cpx<T>* twm = buf1 + FMAX * M;
    const long long clip = (long long)(blockIdx.x / (unsigned)a.groups_per_clip);
    const int group = (int)(blockIdx.x % (unsigned)a.groups_per_clip);
    const bool last = group == a.groups_per_clip - 1;
    const int t_own = group * a.group_hops;                      // first frame whose hop block this group finalises
    const int t_first = t_own - a.halo < 0 ? 0 : t_own - a.halo;  // first frame transformed here
    int t_end = t_own + a.group_hops;                             // one past the last frame transformed here
    if (t_end > a.n_used || last) t_end = a.n_used;
    const int frames = t_end - t_first;                           // <= group_hops + halo <= FMAX (host)
    // (1) Hermitian un-split + W_M table (unsplit_stage above)
    unsplit_stage<T, N, FMAX>(buf0, twm, a.D + clip * a.d_batch_stride + (long long)t_first * a.d_frame_stride, a.d_frame_stride, frames, a.tw_m, a.tw_n);
    __syncthreads();
    // (2) forward passes: Y = FFT(conj Z') = conj(M z)
    cpx<T>* src = buf0;
    cpx<T>* dst = buf1;
    Passes<T, N, 0, FMAX>::run(src, dst, twm, frames);
    // (3) window / N in place: sample pair m of a frame = (Y.x ws[2m], -Y.y ws[2m+1])
    const cpx<T>* __restrict__ ws2 = reinterpret_cast<const cpx<T>*>(a.win_scaled);
    {
        constexpr int IT3 = (FMAX * M + NT - 1) / NT;
        cpx<T> wv[IT3];
#pragma unroll
        for (int it = 0; it < IT3; ++it) wv[it] = ws2[((int)threadIdx.x + it * NT) % M];
#pragma unroll
        for (int it = 0; it < IT3; ++it) {
            const int w = (int)threadIdx.x + it * NT;
            if (w >= frames * M) break;
            const cpx<T> v = src[w];
            src[w] = mkc<T>(v.x * wv[it].x, -v.y * wv[it].y);
        }
    }
    __syncthreads();
    // (4) overlap-add by gathering, increasing frame order; padded position p, output index s = p - drop
    const T* fr = reinterpret_cast<const T*>(src);  // frame f, sample u at fr[f * N + u]
    const long long p0 = (long long)t_own * a.hop;
    long long p1 = (long long)(t_own + a.group_hops) * a.hop;
    if (last) p1 = (long long)(a.n_used - 1) * a.hop + N > p1 ? (long long)(a.n_used - 1) * a.hop + N : p1;  // the clip's tail: up to the last frame's end
    // Positions relative to the first resident frame fit 32 bits (the 64-bit divisions of the absolute form were a third of this stage's instructions);
    // four positions per trip with their normalisation factors loaded together (round 6).
    const long long pbase = (long long)t_first * a.hop;
    const unsigned r0 = (unsigned)(p0 - pbase), r1 = (unsigned)(p1 - pbase), uhop = (unsigned)a.hop;
    const unsigned t_cap = (unsigned)(a.n_used - 1 - t_first);
    for (unsigned rb = r0 + threadIdx.x; rb < r1; rb += 4 * NT) {
        T nrm[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned r = rb + u * NT;
            const long long sidx = pbase + r - a.drop;
            ok[u] = r < r1 && sidx >= 0 && sidx < a.out_len;
            nrm[u] = a.norm[ok[u] ? sidx : 0];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!ok[u]) continue;
            const unsigned r = rb + u * NT;
            const unsigned tl = r + 1 <= (unsigned)N ? 0u : (r + 1 - (unsigned)N + uhop - 1) / uhop;
            unsigned th = r / uhop;
            if (th > t_cap) th = t_cap;
            T acc = (T)0;
            for (unsigned t = tl; t <= th; ++t) acc += fr[t * N + (r - t * uhop)];
            a.y[clip * a.y_stride + (pbase + r - a.drop)] = acc * nrm[u];
        }
    }
}

}  // namespace mixed
}  // namespace lra
