// lra_hpss.h -- harmonic / percussive separation of a spectrogram: librosa.decompose.hpss (librosa/decompose.py:371-528), the step
// between the two transforms of librosa.effects.hpss / harmonic / percussive (librosa/effects.py:70-301).
// Self-contained so that tests/hostsim/postsim.cpp can run the same kernel bodies on the host (-DLRA_POSTSIM).
//
//   mag  = |D|                                                          magnitude_kernel (np.abs; skipped for real input)
//   harm = median over `win_harm` frames around t   (same bin)          scipy.ndimage.median_filter(S, size=[1, .., win], mode="reflect")
//   perc = median over `win_perc` bins around k     (same frame)        scipy.ndimage.median_filter(S, size=[1, .., win, 1], mode="reflect")
//   mask_h = softmask(harm, perc * margin_h, power),  mask_p = softmask(perc, harm * margin_p, power)      (util/utils.py:1895-1932)
//   out_h = (S * mask_h) * phase,  out_p = (S * mask_p) * phase,  phase = D / |D| (1 where |D| == 0)        (decompose.py:528; core/spectrum.py:1347-1361)
// All of it in ONE pass over [clip][frame][bin] (the STFT kernel's layout, bins contiguous): a thread per element, lanes along the
// bins.  The time window is `win` coalesced row reads, the frequency window `win` neighbouring reads that the lanes of a wave share
// through L1; both medians come from a sorting network in registers (Batcher's odd-even merge sort over 32 or 64 slots, unused slots
// +inf; the median is a selection, so the result is exactly the value scipy selects), a counting selection for longer windows.
#pragma once

#ifndef LRA_POSTSIM
#include <hip/hip_runtime.h>
#endif
#include <cfloat>
#include <cmath>

namespace lra {

template <class T> struct HpssCplx { T x, y; };

template <class T> struct HpssLimits;
template <> struct HpssLimits<float> {
    static __device__ __forceinline__ float tiny() { return FLT_MIN; }
    static __device__ __forceinline__ float inf() { return __builtin_huge_valf(); }
};
template <> struct HpssLimits<double> {
    static __device__ __forceinline__ double tiny() { return DBL_MIN; }
    static __device__ __forceinline__ double inf() { return __builtin_huge_val(); }
};

// one rounding per operation, as NumPy's elementwise loops
#pragma clang fp contract(off)
template <class T> struct HpssOps {
    // np.abs of a complex value: float32 through an exactly rounded double sum (NumPy's own float32 hypot differs from it in the
    // last bit depending on the host's SIMD dispatch), float64 through hypot
    static __device__ __forceinline__ T mag(T a, T b);
    // x ** power for a scalar exponent as NumPy evaluates it: 1 -> x, 2 -> x * x, 0.5 -> sqrt(x), else pow
    static __device__ __forceinline__ T powr(T x, T power) {
        if (power == (T)2) return x * x;
        if (power == (T)1) return x;
        if (power == (T)0.5) return sizeof(T) == 4 ? (T)sqrtf((float)x) : (T)sqrt((double)x);
        return sizeof(T) == 4 ? (T)powf((float)x, (float)power) : (T)pow((double)x, (double)power);
    }
    // util.softmask(X, X_ref, power, split_zeros); hard != 0: power = inf, the mask is X > X_ref
    static __device__ __forceinline__ T softmask(T X, T Xr, T power, bool split_zeros, bool hard) {
        if (hard) return X > Xr ? (T)1 : (T)0;
        T Z = X > Xr ? X : Xr;
        const bool bad = Z < HpssLimits<T>::tiny();
        if (bad) return split_zeros ? (T)0.5 : (T)0;
        const T m = powr(X / Z, power), r = powr(Xr / Z, power);
        return m / (m + r);
    }
    static __device__ __forceinline__ T mul(T a, T b) { return a * b; }
    static __device__ __forceinline__ T div(T a, T b) { return a / b; }
    static __device__ __forceinline__ T add(T a, T b) { return a + b; }
};
template <> __device__ __forceinline__ float HpssOps<float>::mag(float a, float b) { return (float)sqrt((double)a * (double)a + (double)b * (double)b); }
template <> __device__ __forceinline__ double HpssOps<double>::mag(double a, double b) { return hypot(a, b); }
#pragma clang fp contract(fast)

template <class T> __global__ __launch_bounds__(256) void magnitude_kernel(const HpssCplx<T>* __restrict__ D, T* __restrict__ mag, long long count) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id < count) mag[id] = HpssOps<T>::mag(D[id].x, D[id].y);
}

// index of a half-sample symmetric ("reflect") extension of 0 .. n-1
__device__ __forceinline__ long long hpss_reflect(long long i, long long n) {
    const long long period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

// compare-exchange as v_min / v_max (2 issue cycles each).  Spelled `a < b ? a : b` / `a < b ? b : a` it compiled to v_cmp + two
// v_cndmask_b32_e32 back to back -- and on gfx950 a VOP2 select directly behind another stalls the SIMD's vector pipe ~16 cycles
// (profiles/r03_experiments.md): 382 exchanges per element spent most of the kernel's time in that stall.  The windows hold
// magnitudes (and +inf in unused slots): no negative zero, so min / max select exactly the values the comparisons would -- FOR FINITE INPUT.
// Non-finite input is outside the contract: v_min / v_max return the other operand for a NaN, so a NaN bin is skipped by the medians
// around it (scipy's median_filter would carry it through); the NumPy entry point rejects such input (decompose.py), a device tensor is
// the caller's responsibility like the non-negativity of a real S.
__device__ __forceinline__ float hpss_min(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float hpss_max(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double hpss_min(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ double hpss_max(double a, double b) { return __builtin_fmax(a, b); }

// Batcher's odd-even merge sort over CAP register slots (CAP a power of two; every index is a compile-time constant after unrolling)
template <class T, int CAP> __device__ __forceinline__ void hpss_sort(T (&a)[CAP]) {
#pragma unroll
    for (int p = 1; p < CAP; p <<= 1)
#pragma unroll
        for (int k = p; k >= 1; k >>= 1)
#pragma unroll
            for (int j = k % p; j + k < CAP; j += 2 * k)
#pragma unroll
                for (int i = 0; i < k; ++i)
                    if (i + j + k < CAP && (i + j) / (2 * p) == (i + j + k) / (2 * p)) {
                        const T lo = hpss_min(a[i + j], a[i + j + k]);
                        const T hi = hpss_max(a[i + j], a[i + j + k]);
                        a[i + j] = lo;
                        a[i + j + k] = hi;
                    }
}

// median (rank win / 2 of the sorted window, scipy's rank filter) of `win` values base[reflect(first + j) * stride], j < win
template <class T, int CAP> __device__ __forceinline__ T hpss_median(const T* __restrict__ base, long long first, long long n, long long stride, int win) {
    if constexpr (CAP > 0) {
        T a[CAP];
        if (first >= 0 && first + win <= n) {
            // the window lies inside the axis (all but the first / last win / 2 positions): a walking pointer, no index fold -- the
            // reflect() of the general path is a 64-bit modulo per tap (~30 instructions), which was most of the kernel's work
            const T* __restrict__ p = base + first * stride;
#pragma unroll
            for (int j = 0; j < CAP; ++j) {
                a[j] = j < win ? *p : HpssLimits<T>::inf();
                if (j + 1 < win) p += stride;
            }
        } else {
#pragma unroll
            for (int j = 0; j < CAP; ++j) a[j] = j < win ? base[hpss_reflect(first + j, n) * stride] : HpssLimits<T>::inf();
        }
        hpss_sort<T, CAP>(a);
        T v = a[0];
#pragma unroll
        for (int j = 1; j < CAP; ++j) v = j == win / 2 ? a[j] : v;
        return v;
    } else {
        // long windows: the value with `rank` smaller-or-equal predecessors, by counting (ties share a value, any of them is the answer)
        const int rank = win / 2;
        for (int c = 0; c < win; ++c) {
            const T v = base[hpss_reflect(first + c, n) * stride];
            int less = 0, equal = 0;
            for (int j = 0; j < win; ++j) {
                const T w = base[hpss_reflect(first + j, n) * stride];
                less += w < v;
                equal += w == v;
            }
            if (less <= rank && rank < less + equal) return v;
        }
        return base[hpss_reflect(first + rank, n) * stride];  // unreachable for NaN-free input
    }
}

struct HpssArgs {
    long long batch, n_frames;
    int n_bins, win_harm, win_perc;
    double power, margin_harm, margin_perc;
    int hard;       // power = inf: hard masks
    int want_mask;  // outputs are the masks themselves
};

// mag: [clip][frame][bin] real; D: the complex spectrogram in the same layout, or nullptr for a real-valued S (phase = 1)
// out_h / out_p: masks (real) | masked real spectrogram (real) | masked complex spectrogram (complex, D != nullptr).
// CAP: slots of the sorting network (32 or 64, >= both windows) or 0 for the counting selection.
template <class T, int CAP>
__global__ __launch_bounds__(256) void hpss_kernel(const T* __restrict__ mag, const HpssCplx<T>* __restrict__ D, void* __restrict__ out_h, void* __restrict__ out_p, HpssArgs a) {
    using Op = HpssOps<T>;
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per_clip = a.n_frames * a.n_bins;
    if (id >= a.batch * per_clip) return;
    const long long clip = id / per_clip, t = (id % per_clip) / a.n_bins;
    const int k = (int)(id % a.n_bins);
    const T* __restrict__ m0 = mag + clip * per_clip;
    const T harm = hpss_median<T, CAP>(m0 + k, t - a.win_harm / 2, a.n_frames, a.n_bins, a.win_harm);
    const T perc = hpss_median<T, CAP>(m0 + t * a.n_bins, (long long)k - a.win_perc / 2, a.n_bins, 1, a.win_perc);
    const bool split_zeros = a.margin_harm == 1.0 && a.margin_perc == 1.0;
    const T mask_h = Op::softmask(harm, Op::mul(perc, (T)a.margin_harm), (T)a.power, split_zeros, a.hard != 0);
    const T mask_p = Op::softmask(perc, Op::mul(harm, (T)a.margin_perc), (T)a.power, split_zeros, a.hard != 0);
    if (a.want_mask) {
        ((T*)out_h)[id] = mask_h;
        ((T*)out_p)[id] = mask_p;
        return;
    }
    const T s = m0[(long long)t * a.n_bins + k];
    const T sh = Op::mul(s, mask_h), sp = Op::mul(s, mask_p);
    if (!D) {
        ((T*)out_h)[id] = sh;
        ((T*)out_p)[id] = sp;
        return;
    }
    // phase = D / |D| with 1 + 0j where |D| == 0 (magphase): real = D.re / (|D| + z) + z, imag = D.im / (|D| + z), z = (|D| == 0)
    const HpssCplx<T> d = D[id];
    const T z = s == (T)0 ? (T)1 : (T)0;
    const T nz = Op::add(s, z);
    const T pr = Op::add(Op::div(d.x, nz), z), pi = Op::div(d.y, nz);
    HpssCplx<T> oh, op;
    oh.x = Op::mul(sh, pr);
    oh.y = Op::mul(sh, pi);
    op.x = Op::mul(sp, pr);
    op.y = Op::mul(sp, pi);
    ((HpssCplx<T>*)out_h)[id] = oh;
    ((HpssCplx<T>*)out_p)[id] = op;
}

}  // namespace lra
