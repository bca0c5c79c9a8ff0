// lra_hpss.h -- harmonic / percussive separation of a spectrogram: librosa.decompose.hpss (librosa/decompose.py:371-528), the step
// between the two transforms of librosa.effects.hpss / harmonic / percussive (librosa/effects.py:70-301).
// Self-contained so that tests/hostsim/postsim.cpp can run the same kernel bodies on the host (-DLRA_POSTSIM).
//
//   mag  = |D|                                                          magnitude_kernel (np.abs; skipped for real input)
//   harm = median over `win_harm` frames around t   (same bin)          scipy.ndimage.median_filter(S, size=[1, .., win], mode="reflect")
//   perc = median over `win_perc` bins around k     (same frame)        scipy.ndimage.median_filter(S, size=[1, .., win, 1], mode="reflect")
//   mask_h = softmask(harm, perc * margin_h, power),  mask_p = softmask(perc, harm * margin_p, power)      (util/utils.py:1895-1932)
//   out_h = (S * mask_h) * phase,  out_p = (S * mask_p) * phase,  phase = D / |D| (1 where |D| == 0)        (decompose.py:528; core/spectrum.py:1347-1361)
// All of it in ONE pass over [clip][frame][bin] (the STFT kernel's layout, bins contiguous).  Both medians come from a sorting network
// in registers (Batcher's odd-even merge sort over 32 or 64 slots; the median is a selection, so the result is exactly the value scipy
// selects).  Two kernels:
//   hpss_tile_kernel   a thread per 4 x 4 tile of elements; four neighbouring windows of an axis share ONE sorted core (hpss_median_quad):
//                      8 networks per 16 elements instead of 32.  Windows of 6 .. 33 (float32: .. 65) on axes of at least win + 4.
//   hpss_kernel        a thread per element (two networks each), a counting selection for longer windows: everything else.
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
        // Z is X or Xr, so one of the two quotients is exactly 1 (and 1 ** power = 1): ONE correctly rounded division of the smaller by
        // the larger gives both terms, bit for bit what X / Z and Xr / Z give (an IEEE float32 division is ~11 vector instructions)
        const T q = powr((X > Xr ? Xr : X) / Z, power);
        const T m = X > Xr ? (T)1 : q, r = X > Xr ? q : (T)1;
        return m / (m + r);
    }
    // both masks of an element, softmask(h, p * margin_h) and softmask(p, h * margin_p).  With both margins 1 (split_zeros: the default)
    // they share Z = max(h, p), the quotient and the sum m + r (the same two terms in either order): three divisions instead of six.
    static __device__ __forceinline__ void softmask2(T h, T p, T margin_h, T margin_p, T power, bool split_zeros, bool hard, T& mask_h, T& mask_p) {
        if (!split_zeros || hard) {
            mask_h = softmask(h, p * margin_h, power, split_zeros, hard);
            mask_p = softmask(p, h * margin_p, power, split_zeros, hard);
            return;
        }
        const bool h_larger = h > p;
        const T Z = h_larger ? h : p;
        if (Z < HpssLimits<T>::tiny()) {
            mask_h = mask_p = (T)0.5;
            return;
        }
        const T q = powr((h_larger ? p : h) / Z, power), sum = h_larger ? (T)1 + q : q + (T)1;
        const T one = (T)1 / sum, part = q / sum;
        mask_h = h_larger ? one : part;
        mask_p = h_larger ? part : one;
    }
    static __device__ __forceinline__ T mul(T a, T b) { return a * b; }
    static __device__ __forceinline__ T div(T a, T b) { return a / b; }
    static __device__ __forceinline__ T add(T a, T b) { return a + b; }
};
template <> __device__ __forceinline__ float HpssOps<float>::mag(float a, float b) { return (float)sqrt((double)a * (double)a + (double)b * (double)b); }
template <> __device__ __forceinline__ double HpssOps<double>::mag(double a, double b) { return hypot(a, b); }
// librosa.magphase (core/spectrum.py:1296-1361): mag = |D| ** power, phase = D / |D| with 1 + 0j where |D| = 0 (real and imaginary parts
// divided separately, as the reference does for the sake of denormals).  D complex, or real (its imaginary part is zero).  x ** power as
// NumPy evaluates a scalar exponent on an array: 1, 2, 0.5, -1, 0 take its fast paths (copy, square, sqrt, reciprocal, ones), else pow.
template <class T> __device__ __forceinline__ T magphase_pow(T m, T power) {
    if (power == (T)1) return m;
    if (power == (T)2) return m * m;
    if (power == (T)0.5) return sizeof(T) == 4 ? (T)sqrtf((float)m) : (T)sqrt((double)m);
    if (power == (T)-1) return (T)1 / m;
    if (power == (T)0) return (T)1;
    return sizeof(T) == 4 ? (T)powf((float)m, (float)power) : (T)pow((double)m, (double)power);
}
template <class T>
__global__ __launch_bounds__(256) void magphase_kernel(const void* __restrict__ D, int is_complex, T* __restrict__ mag, HpssCplx<T>* __restrict__ phase, long long count, T power) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= count) return;
    T re, im, m;
    if (is_complex) {
        const HpssCplx<T> d = ((const HpssCplx<T>*)D)[id];
        re = d.x;
        im = d.y;
        m = HpssOps<T>::mag(re, im);
    } else {
        re = ((const T*)D)[id];
        im = (T)0;
        m = re < (T)0 ? -re : re;   // np.abs of a real: the sign bit cleared (-0.0 -> 0.0)
        if (m == (T)0) m = (T)0;
    }
    const T z = m == (T)0 ? (T)1 : (T)0;
    const T nz = HpssOps<T>::add(m, z);
    HpssCplx<T> ph;
    ph.x = HpssOps<T>::add(HpssOps<T>::div(re, nz), z);
    ph.y = HpssOps<T>::div(im, nz);
    phase[id] = ph;
    mag[id] = magphase_pow<T>(m, power);
}
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
// Slots below LO hold -inf and slots from HI on +inf when the window length is a compile-time constant (hpss_quad_load): an exchange
// with such a slot moves nothing (the lower index already has the smaller value), so it is skipped -- the network shrinks to the live slots.
template <class T, int CAP, int LO = 0, int HI = CAP> __device__ __forceinline__ void hpss_sort(T (&a)[CAP]) {
#pragma unroll
    for (int p = 1; p < CAP; p <<= 1)
#pragma unroll
        for (int k = p; k >= 1; k >>= 1)
#pragma unroll
            for (int j = k % p; j + k < CAP; j += 2 * k)
#pragma unroll
                for (int i = 0; i < k; ++i)
                    if (i + j + k < CAP && (i + j) / (2 * p) == (i + j + k) / (2 * p) && i + j >= LO && i + j + k < HI) {
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

// FOUR NEIGHBOURING windows at once (round 4).  The windows of positions p .. p + 3 of an axis share win - 3 of their win values (the core:
// x[3] .. x[win - 1] of x[i] = value at p - win / 2 + i); window q adds the three extras x[q] .. x[2], x[win] .. x[win + q - 1].  With the
// core SORTED once (A[0] <= A[1] <= ...) and the three extras sorted (b0 <= b1 <= b2), the value of rank r = win / 2 of their union is
//     min(A[r], max(A[r - 1], b0), max(A[r - 2], b1), max(A[r - 3], b2))
// (the k-th smallest of two sorted lists is the least max(A_i, B_j) over i + j = k; A[< 0] = -inf, A[>= size] = +inf).  One sorting network
// -- the kernel's cost: 191 compare-exchanges over 32 slots -- then serves four outputs at 3 + 3 exchanges each, and the result is still a
// SELECTION (no arithmetic): exactly the value scipy's median_filter picks.  So that A[r - 3 .. r] sit in compile-time slots whatever
// the window length, the core is loaded behind L = CAP / 2 - r slots of -inf (and in front of +inf): A[r] is slot CAP / 2 after sorting; the
// network's exchanges that feed none of the four slots read are dead code to the compiler.  Serves 6 <= win <= CAP + 1 on an axis of at
// least win + 4 positions (one fold of the reflection then reaches every index: hpss_tiles()).
// Addressing: inside the axis (FOLD = false) a lane's pointer to x[0] plus WAVE-UNIFORM offsets (the scalar unit multiplies by the stride);
// at the ends the folded position times the stride as a 32-bit product (hpss_plan: a clip has fewer than 2^31 elements).  Spelled
// base[(first + i) * stride] both ways, every load paid two quarter-rate 64-bit vector multiplies -- as many cycles as its slot's share of the sort.
template <bool FOLD> __device__ __forceinline__ int hpss_pos(int i, int n) {
    i = i < 0 ? -1 - i : i;
    i = i >= n ? 2 * n - 1 - i : i;
    return i;
}
template <class T, bool FOLD> __device__ __forceinline__ T hpss_at(const T* __restrict__ base, const T* __restrict__ p0, int first, int i, int n, long long stride) {
    if (FOLD) return base[hpss_pos<FOLD>(first + i, n) * (int)stride];
    return p0[(long long)i * stride];
}

// a wave-uniform value the optimiser may not look through: it stays in scalar registers (s_cselect_b32 feeding v_min / v_max as the scalar
// operand).  Left visible, `max(v, cond ? -inf : +inf)` is rewritten into `cond ? v : +inf` -- the select the min / max form is there to avoid.
// (selected as integer bit patterns: a uniform float select is still lowered to a vector one.)
template <class T> struct HpssPad;
template <> struct HpssPad<float> {
    static __device__ __forceinline__ float pick(bool minus) {
        unsigned bits = minus ? 0xff800000u : 0x7f800000u;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+s"(bits));
#endif
        return __builtin_bit_cast(float, bits);
    }
};
template <> struct HpssPad<double> {
    static __device__ __forceinline__ double pick(bool minus) {
        unsigned long long bits = minus ? 0xfff0000000000000ull : 0x7ff0000000000000ull;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+s"(bits));
#endif
        return __builtin_bit_cast(double, bits);
    }
};

// WIN: the window length as a compile-time constant (0: `win` at run time).  Then the padding slots are constants, the core's offsets
// fold into the addressing and the network is pruned to the live slots (hpss_sort).
template <class T, int CAP, bool FOLD, int WIN>
__device__ __forceinline__ void hpss_quad_load(const T* __restrict__ base, int first, int n, long long stride, int win_rt, T (&a)[CAP], T (&x)[6]) {
    const int win = WIN ? WIN : win_rt;
    const int c = win - 3, L = CAP / 2 - win / 2;
    const T* __restrict__ p0 = base + (long long)first * stride;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        x[i] = hpss_at<T, FOLD>(base, p0, first, i, n, stride);
        x[3 + i] = hpss_at<T, FOLD>(base, p0, first, win + i, n, stride);
    }
    // Slots below CAP / 2 can only be -inf padding, slots from CAP / 2 on only +inf padding (win >= 6: hpss_plan).  The padding is applied as a
    // min / max with a wave-uniform +-inf, not as a select on the loaded value: back-to-back v_cndmask stall the vector pipe (above), and the
    // loads of padding slots just repeat a core address (cache hits).
#pragma unroll
    for (int s = 0; s < CAP; ++s) {
        if (WIN) {
            if (s < L) a[s] = -HpssLimits<T>::inf();
            else if (s - L < c) a[s] = hpss_at<T, FOLD>(base, p0, first, 3 + s - L, n, stride);
            else a[s] = HpssLimits<T>::inf();
        } else if (s < CAP / 2) {
            const int idx = s - L < 0 ? 0 : s - L;
            const T v = hpss_at<T, FOLD>(base, p0, first, 3 + idx, n, stride);
            a[s] = hpss_min(v, HpssPad<T>::pick(s < L));
        } else {
            const int idx = s - L > c - 1 ? c - 1 : s - L;
            const T v = hpss_at<T, FOLD>(base, p0, first, 3 + idx, n, stride);
            a[s] = hpss_max(v, HpssPad<T>::pick(s - L < c));
        }
    }
}

template <class T> __device__ __forceinline__ void hpss_cx(T& lo, T& hi) {
    const T l = hpss_min(lo, hi), h = hpss_max(lo, hi);
    lo = l;
    hi = h;
}

template <class T, int CAP, int WIN> __device__ __forceinline__ void hpss_median_quad(const T* __restrict__ base, int first, int n, long long stride, int win_rt, T (&m)[4]) {
    static_assert(CAP >= 8, "sorting-network form");
    static_assert(WIN == 0 || (WIN >= 6 && WIN <= CAP + 1), "window the network serves");
    const int win = WIN ? WIN : win_rt;
    T a[CAP], x[6];
    if (first >= 0 && first + win + 3 <= n) hpss_quad_load<T, CAP, false, WIN>(base, first, n, stride, win, a, x);
    else hpss_quad_load<T, CAP, true, WIN>(base, first, n, stride, win, a, x);
    constexpr int P = CAP / 2;
    hpss_sort<T, CAP, WIN ? P - WIN / 2 : 0, WIN ? P - WIN / 2 + WIN - 3 : CAP>(a);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        T b0 = x[q], b1 = x[q + 1], b2 = x[q + 2];
        hpss_cx(b0, b1);
        hpss_cx(b1, b2);
        hpss_cx(b0, b1);
        m[q] = hpss_min(hpss_min(a[P], hpss_max(a[P - 1], b0)), hpss_min(hpss_max(a[P - 2], b1), hpss_max(a[P - 3], b2)));
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
    T mask_h, mask_p;
    Op::softmask2(harm, perc, (T)a.margin_harm, (T)a.margin_perc, (T)a.power, split_zeros, a.hard != 0, mask_h, mask_p);
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

// four consecutive elements of a row as ONE access: the struct is aligned like its scalar, so the address may be any element's
template <class T, class E> struct __attribute__((packed, aligned(sizeof(T)))) HpssRow4 {
    E v[4];
};
template <class T, class E> __device__ __forceinline__ void hpss_row_load(const E* __restrict__ p, E (&v)[4], bool whole, int have) {
    if (whole) {
        const HpssRow4<T, E> r = *(const HpssRow4<T, E>*)p;
#pragma unroll
        for (int b = 0; b < 4; ++b) v[b] = r.v[b];
    } else {
#pragma unroll
        for (int b = 0; b < 4; ++b) v[b] = p[b < have ? b : 0];
    }
}
template <class T, class E> __device__ __forceinline__ void hpss_row_store(E* __restrict__ p, const E (&v)[4], bool whole, int have) {
    if (whole) {
        HpssRow4<T, E> r;
#pragma unroll
        for (int b = 0; b < 4; ++b) r.v[b] = v[b];
        *(HpssRow4<T, E>*)p = r;
    } else {
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if (b < have) p[b] = v[b];
    }
}

// Which kernel serves a problem (shared by hpss_launch in lra_api.hip and its simulator mirror): sorting-network slots for the 4 x 4-tile
// kernel (0: not served), else for the element kernel (0: counting selection).  64 float64 slots would not fit the register file.
constexpr int kHpssFixedWin = 31;  // librosa's default kernel_size: its own instantiation of the tile kernel (compile-time window)
struct HpssPlan {
    int tile_slots, element_slots;
    bool fixed_win;  // both windows are kHpssFixedWin long
};
inline HpssPlan hpss_plan(const HpssArgs& a, int elem_bytes) {
    const int widest = a.win_harm > a.win_perc ? a.win_harm : a.win_perc, narrowest = a.win_harm < a.win_perc ? a.win_harm : a.win_perc;
    HpssPlan p;
    p.element_slots = widest <= 32 ? 32 : (widest <= 64 && elem_bytes == 4 ? 64 : 0);
    p.tile_slots = widest <= 33 ? 32 : (widest <= 65 && elem_bytes == 4 ? 64 : 0);
    p.fixed_win = a.win_harm == kHpssFixedWin && a.win_perc == kHpssFixedWin;
    if (narrowest < 6 || a.n_frames < a.win_harm + 4 || a.n_bins < a.win_perc + 4 || a.n_frames * a.n_bins > 0x7fffffffLL) p.tile_slots = 0;
    return p;
}
inline long long hpss_tiles(const HpssArgs& a) { return a.batch * ((a.n_frames + 3) / 4) * ((a.n_bins + 3) / 4); }

// The same separation with a thread per 4 x 4 tile of elements (frames t0 .. t0 + 3 x bins k0 .. k0 + 3): four harmonic quads (one per bin,
// along time) and four percussive ones (one per frame, along the bins) -- eight sorting networks for sixteen elements instead of
// thirty-two.  ONE network body in a loop of eight rounds (unrolled, the kernel was 27 000 instructions -- twice the instruction cache a CU
// pair shares -- and kept four windows in registers at once: 320 VGPRs).  The workgroup's 256 tiles are consecutive (bin quads of a frame
// quad, then the next one's), and the harmonic rounds deal their 1 024 (tile, bin) columns out so that consecutive LANES take consecutive
// BINS (round u: column 256 u + lane): every load of a window row is one dense 256-byte run per wave.  With each thread walking its own
// tile's four bins instead (lanes 16 bytes apart) a wave's load touched eight lines for the bytes of two, and the kernel was bound by the
// L2 feeding them.  The medians cross to the tile's owner through LDS; the percussive rounds read along the rows, where the windows of
// consecutive steps overlap (L1 hits).  Element arithmetic identical to hpss_kernel's.
template <class T, int CAP, int WIN>
__global__ __launch_bounds__(256) void hpss_tile_kernel(const T* __restrict__ mag, const HpssCplx<T>* __restrict__ D, void* __restrict__ out_h, void* __restrict__ out_p, HpssArgs a) {
    using Op = HpssOps<T>;
    __shared__ T harm[4 * 1024];  // [frame of the tile][column of the workgroup = tile * 4 + bin]
    const long long id0 = (long long)blockIdx.x * 256;
    const int kq = (a.n_bins + 3) / 4;                    // bin quads per frame
    const long long tq = (a.n_frames + 3) / 4;            // frame quads per clip
    const long long total = a.batch * tq * kq, per_clip = a.n_frames * a.n_bins;
    const bool split_zeros = a.margin_harm == 1.0 && a.margin_perc == 1.0;
#pragma unroll 1
    for (int u = 0; u < 8; ++u) {
        if (u == 4) __syncthreads();
        const bool along_time = u < 4;
        const int f = u - 4, col = along_time ? u * 256 + (int)threadIdx.x : 4 * (int)threadIdx.x;
        const long long id = id0 + (col >> 2);
        if (id >= total) continue;
        const long long clip = id / (tq * kq);
        const int t0 = 4 * (int)((id / kq) % tq), k0 = 4 * (int)(id % kq);
        if (!along_time && t0 + f >= a.n_frames) continue;
        const T* __restrict__ m0 = mag + clip * per_clip;
        const int k = k0 + (col & 3) < a.n_bins ? k0 + (col & 3) : a.n_bins - 1;  // a bin past the end repeats the last one (never used)
        const T* __restrict__ base = along_time ? m0 + k : m0 + (long long)(t0 + f) * a.n_bins;
        T m[4];
        hpss_median_quad<T, CAP, WIN>(base, along_time ? t0 - a.win_harm / 2 : k0 - a.win_perc / 2, along_time ? (int)a.n_frames : a.n_bins, along_time ? (long long)a.n_bins : 1LL,
                                 along_time ? a.win_harm : a.win_perc, m);
        if (along_time) {
#pragma unroll
            for (int q = 0; q < 4; ++q) harm[q * 1024 + col] = m[q];
            continue;
        }
        // the tile's row of this frame: four consecutive elements, moved as whole 16-byte accesses where the row has all four (rows start at
        // any multiple of the element size: global accesses need no more).  One element at a time, lanes 16 / 32 bytes apart made four
        // partial-line requests of every line: the complex output ran at the speed of a kernel twice its size.
        const long long e0 = clip * per_clip + (long long)(t0 + f) * a.n_bins + k0;
        const bool whole = k0 + 4 <= a.n_bins;
        T mask_h[4], mask_p[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const T h = harm[f * 1024 + col + b], p = m[b];
            Op::softmask2(h, p, (T)a.margin_harm, (T)a.margin_perc, (T)a.power, split_zeros, a.hard != 0, mask_h[b], mask_p[b]);
        }
        if (a.want_mask) {
            hpss_row_store<T, T>((T*)out_h + e0, mask_h, whole, a.n_bins - k0);
            hpss_row_store<T, T>((T*)out_p + e0, mask_p, whole, a.n_bins - k0);
            continue;
        }
        T s[4], sh[4], sp[4];
        hpss_row_load<T, T>(mag + e0, s, whole, a.n_bins - k0);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            sh[b] = Op::mul(s[b], mask_h[b]);
            sp[b] = Op::mul(s[b], mask_p[b]);
        }
        if (!D) {
            hpss_row_store<T, T>((T*)out_h + e0, sh, whole, a.n_bins - k0);
            hpss_row_store<T, T>((T*)out_p + e0, sp, whole, a.n_bins - k0);
            continue;
        }
        HpssCplx<T> d[4], oh[4], op[4];
        hpss_row_load<T, HpssCplx<T>>(D + e0, d, whole, a.n_bins - k0);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const T z = s[b] == (T)0 ? (T)1 : (T)0;
            const T nz = Op::add(s[b], z);
            const T pr = Op::add(Op::div(d[b].x, nz), z), pi = Op::div(d[b].y, nz);
            oh[b].x = Op::mul(sh[b], pr);
            oh[b].y = Op::mul(sh[b], pi);
            op[b].x = Op::mul(sp[b], pr);
            op[b].y = Op::mul(sp[b], pi);
        }
        hpss_row_store<T, HpssCplx<T>>((HpssCplx<T>*)out_h + e0, oh, whole, a.n_bins - k0);
        hpss_row_store<T, HpssCplx<T>>((HpssCplx<T>*)out_p + e0, op, whole, a.n_bins - k0);
    }
}

}  // namespace lra
