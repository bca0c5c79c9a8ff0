// lra_cqt.h -- the two kernels the constant-Q / variable-Q octave recursion adds to the STFT (SURVEY.md 8f rank 4;
// librosa/core/constantq.py:1054-1099): the FIR decimator between octaves and the sparse filter-basis projection of every frame.
// Self-contained so that tests/hostsim/postsim.cpp can run the same kernel bodies on host threads (-DLRA_POSTSIM).
//
//   per octave:  D = stft(y, n_fft, hop, window="ones")                    the existing forward kernel ([clip][frame][bin])
//                C[clip][frame][bin0 + r] = (basis[r] . D[clip][frame]) / sqrt(length[r])       cqt_project_kernel
//                y = decimate(y, 2) * sqrt(2)                                                  fir_decimate_kernel
#pragma once

#ifndef LRA_POSTSIM
#include <hip/hip_runtime.h>
#endif

namespace lra {

template <class T> struct CqtCplx { T x, y; };

// one rounding per operation, sums in the order of the host libraries (scipy's upfirdn and CSR products are plain C loops)
#pragma clang fp contract(off)
template <class T> struct CqtOps {
    static __device__ __forceinline__ T madd(T acc, T a, T b) { return acc + a * b; }
    // acc + a * b for complex a, b: (ar br - ai bi, ar bi + ai br), as scipy.sparse's complex wrapper multiplies
    static __device__ __forceinline__ CqtCplx<T> cmadd(CqtCplx<T> acc, CqtCplx<T> a, CqtCplx<T> b) {
        const T re = a.x * b.x - a.y * b.y;
        const T im = a.x * b.y + a.y * b.x;
        CqtCplx<T> r;
        r.x = acc.x + re;
        r.y = acc.y + im;
        return r;
    }
};
// two running sums at once: acc + v h per half, product and sum rounded separately (as madd above); float on the device: v_pk_mul_f32 + v_pk_add_f32
template <class T> struct Pair2 {
    T lo, hi;
};
template <class T> __device__ __forceinline__ Pair2<T> fma2_exact(Pair2<T> acc, Pair2<T> v, T h) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (sizeof(T) == 4) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 hv = {h, h};
        const f2 prod = __builtin_bit_cast(f2, v) * hv;
        const f2 r = __builtin_bit_cast(f2, acc) + prod;
        return __builtin_bit_cast(Pair2<T>, r);
    }
#endif
    acc.lo = acc.lo + v.lo * h;
    acc.hi = acc.hi + v.hi * h;
    return acc;
}
#pragma clang fp contract(fast)

// ---- FIR decimator: scipy.signal.resample_poly(x, 1, down) with zero padding (librosa/core/audio.py:676-693, the resampler behind
// constantq.py:1095-1098 and :1254-1256 for res_type="polyphase"; also the carrier of this library's own half-band design for the
// resamplers that are not in the image) ---------------------------------------------------------------------------------------------
//   out[clip][n] = (sum_k h[k] x[clip][(n + first) * down - k]) / div,   x = 0 outside [0, n_in),   0 <= n < n_out
// h: the padded filter scipy hands to upfirdn (n_pre_pad zeros in front), `first` = n_pre_remove, the sum runs over ascending input
// index like upfirdn's inner loop; div = sqrt(ratio) from resample(scale=True) (:719-720), applied in float64 as NumPy does for a
// float64 scalar divisor (mul likewise).
// Fallback for spans that do not fit the LDS (large early-downsampling factors): a thread per output sample reading the signal
// directly; neighbouring threads read overlapping windows through L1.  Same sum, same order.
template <class T>
__global__ __launch_bounds__(256) void fir_decimate_direct_kernel(const T* __restrict__ x, T* __restrict__ out, const T* __restrict__ h, long long batch, long long n_in, long long n_out,
                                                                  int n_taps, int down, int first, double div, double mul) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= batch * n_out) return;
    const long long clip = id / n_out, n = id % n_out;
    const T* __restrict__ xc = x + clip * n_in;
    const long long top = (n + first) * (long long)down;  // input index under tap 0
    long long lo = top - (n_taps - 1);
    if (lo < 0) lo = 0;
    const long long hi = top < n_in - 1 ? top : n_in - 1;
    T acc = (T)0;
    for (long long i = lo; i <= hi; ++i) acc = CqtOps<T>::madd(acc, xc[i], h[top - i]);
    const T scaled = (T)((double)acc / div);
    out[id] = mul == 1.0 ? scaled : (T)((double)scaled * mul);
}

// ---- general rational resampling, scipy.signal.resample_poly(x, up, down) (librosa.resample(res_type="polyphase"), core/audio.py:676-693):
// the signal with up - 1 zeros between samples, filtered by h, every down-th sample kept (scipy's upfirdn).  Output n sits at position
// (n + first) * down of the zero-stuffed signal; only the taps over real samples are summed, over ascending input index like upfirdn's
// inner loop: out[n] = sum_k x[k] h[(n + first) down - k up].  ceil(n_taps / up) products per output (the filter has 20 max(up, down) + 1
// taps: about 20 down / up + 1 of them), a thread per output, neighbouring threads' windows overlap (L1).
template <class T>
__global__ __launch_bounds__(256) void resample_poly_kernel(const T* __restrict__ x, T* __restrict__ out, const T* __restrict__ h, long long batch, long long n_in, long long n_out, int n_taps,
                                                            int up, int down, int first, double div, double mul) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= batch * n_out) return;
    const long long clip = id / n_out, n = id % n_out;
    const T* __restrict__ xc = x + clip * n_in;
    const long long top = (n + first) * (long long)down;  // position of tap 0 in the zero-stuffed signal
    long long k_hi = top / up;                              // last real sample at or before it
    long long k_lo = top - (n_taps - 1) <= 0 ? 0 : (top - (n_taps - 1) + up - 1) / up;
    if (k_hi > n_in - 1) k_hi = n_in - 1;
    T acc = (T)0;
    for (long long k = k_lo; k <= k_hi; ++k) acc = CqtOps<T>::madd(acc, xc[k], h[top - k * up]);
    const T scaled = (T)((double)acc / div);
    out[id] = mul == 1.0 ? scaled : (T)((double)scaled * mul);
}

// ---- Fourier-domain resampling, scipy.signal.resample(x, num) for real x (librosa.resample(res_type="fft" / "scipy"), core/audio.py:672-675):
// rfft, the lower min(num, Nx) / 2 + 1 bins copied (the rest zero), the shared Nyquist bin doubled when shortening / halved when
// lengthening an even length, irfft, times num / Nx.  This kernel is the middle step: Y[clip][k] of bins_out bins from X[clip][k] of
// bins_in; `gain` carries the normalisation of the unnormalised inverse transform (1 / Nx) and the caller's scaling.  The imaginary
// parts of bin 0 and of the output's own Nyquist bin are cleared (irfft ignores them).
template <class T>
__global__ __launch_bounds__(256) void resample_spectrum_kernel(const CqtCplx<T>* __restrict__ X, CqtCplx<T>* __restrict__ Y, long long batch, long long bins_in, long long bins_out,
                                                                long long n_copy, long long shared_nyquist, T nyquist_factor, long long real_last, T gain) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= batch * bins_out) return;
    const long long clip = id / bins_out, k = id % bins_out;
    CqtCplx<T> v;
    v.x = v.y = (T)0;
    if (k < n_copy) {
        v = X[clip * bins_in + k];
        if (k == shared_nyquist) {
            v.x *= nyquist_factor;
            v.y *= nyquist_factor;
        }
        if (k == 0 || k == real_last) v.y = (T)0;
        v.x *= gain;
        v.y *= gain;
    }
    Y[id] = v;
}

// ---- band-limited resampling by a rational ratio in the Fourier domain (the library's own converter behind the soxr / kaiser / sinc names
// when the ratio is not a plain decimation): the clip zero-padded to n_fft_in = g down samples (g down >= n_in + the filter's length, so the
// circular convolution IS the linear one), rfft, every bin times the low-pass H(k) = erfc((k - k_mid) / k_sigma) / 2 (flat to 6e-7 up to 0.913
// of the lower Nyquist, 6e-7 = -125 dB at the lower Nyquist, a Gaussian-decaying impulse response: ~100 / f_c samples to 1e-7), the bins
// above the lower Nyquist dropped, irfft of n_fft_out = g up samples, the first n_out kept.  Same band edges as the polyphase design it
// replaces for these ratios (258 products per output at 22 050 -> 16 000 Hz), at the cost of two transforms.
template <class T>
__global__ __launch_bounds__(256) void resample_shaped_spectrum_kernel(const CqtCplx<T>* __restrict__ X, CqtCplx<T>* __restrict__ Y, long long batch, long long bins_in, long long bins_out,
                                                                       long long n_copy, long long real_last, double k_mid, double inv_sigma, T gain) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= batch * bins_out) return;
    const long long clip = id / bins_out, k = id % bins_out;
    CqtCplx<T> v;
    v.x = v.y = (T)0;
    if (k < n_copy) {
        v = X[clip * bins_in + k];
        const double u = ((double)k - k_mid) * inv_sigma;
        const T h = u < -6.0 ? gain : (T)(0.5 * erfc(u)) * gain;
        if (k == 0 || k == real_last) v.y = (T)0;
        v.x *= h;
        v.y *= h;
    }
    Y[id] = v;
}

// rows of `width` elements from pitch src_pitch to pitch dst_pitch (elements), the rest of each destination row zero (dst_pitch >= width
// pads, dst_pitch < src_pitch with width = dst_pitch crops)
template <class T> __global__ __launch_bounds__(256) void repitch_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, long long rows, long long width, long long src_pitch, long long dst_pitch) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= rows * dst_pitch) return;
    const long long r = id / dst_pitch, c = id % dst_pitch;
    dst[id] = c < width ? src[r * src_pitch + c] : (T)0;
}

// A workgroup produces 256 consecutive outputs of one clip: their common input span (255 down + n_taps samples) is staged in LDS
// once (zeros outside the signal: adding 0 * h leaves the running sum unchanged), every thread then walks its n_taps-long window of
// it; the tap index is the same for all lanes in each step, so the taps come through the scalar cache.
#ifdef LRA_POSTSIM
#define LRA_DYN_LDS(T, name) T* name = reinterpret_cast<T*>(g_postsim_dyn_lds)
#else
#define LRA_DYN_LDS(T, name)                         \
    extern __shared__ __attribute__((aligned(16))) unsigned char lra_dyn_lds_raw[]; \
    T* name = reinterpret_cast<T*>(lra_dyn_lds_raw)
#endif

// OPT outputs per thread, 256 apart: the same sums in the same order, OPT independent chains in flight per thread (the one-output form is a
// chain of n_taps dependent LDS reads + multiply-adds, latency-bound) and a quarter as many workgroups.
template <class T, int OPT>
__global__ __launch_bounds__(256) void fir_decimate_kernel(const T* __restrict__ x, T* __restrict__ out, const T* __restrict__ h, long long n_in, long long n_out, int blocks_per_clip, int n_taps,
                                                           int down, int first, double div, double mul) {
    LRA_DYN_LDS(T, xs);
    const long long clip = blockIdx.x / blocks_per_clip;
    const long long n0 = (long long)(blockIdx.x % blocks_per_clip) * (256 * OPT);
    const int t = threadIdx.x;
    const T* __restrict__ xc = x + clip * n_in;
    const long long base = (n0 + first) * (long long)down - (n_taps - 1);  // input index of xs[0]
    const int span = (256 * OPT - 1) * down + n_taps;
    for (int s = t; s < span; s += 256) {
        const long long gi = base + s;
        xs[s] = (gi >= 0 && gi < n_in) ? xc[gi] : (T)0;
    }
    __syncthreads();
    T acc[OPT];
#pragma unroll
    for (int q = 0; q < OPT; ++q) acc[q] = (T)0;
    for (int j = 0; j < n_taps; ++j) {
        const T hj = h[n_taps - 1 - j];
#pragma unroll
        for (int q = 0; q < OPT; ++q) acc[q] = CqtOps<T>::madd(acc[q], xs[(t + 256 * q) * down + j], hj);  // output n0 + t + 256 q: its window, oldest sample first
    }
#pragma unroll
    for (int q = 0; q < OPT; ++q) {
        const long long n = n0 + t + 256 * q;
        if (n < n_out) {
            const T scaled = (T)((double)acc[q] / div);
            out[clip * n_out + n] = mul == 1.0 ? scaled : (T)((double)scaled * mul);  // `y *= sqrt(factor)` of the unscaled transform (constantq.py:1263-1264)
        }
    }
}

// Halving (down = 2: every step of the octave recursion), four CONSECUTIVE outputs per thread.  Output n takes inputs 2 (n + first) - k, so
// neighbouring outputs share all but two of their n_taps inputs: a thread that walks its n_taps + 6 input samples once feeds each of them to up
// to four running sums -- the same products added in the same (ascending-input) order per output, a quarter of the LDS reads.  The strided
// form above is bound by exactly those reads (41 per output for the Kaiser half-band, 375 for this library's own design).  A thread's window
// starts eight samples after its neighbour's; the staged span is laid out with one pad slot per eight samples (sample i at slot i + i / 8), so
// that the lanes' reads of "their j-th sample" are nine slots apart: conflict-free, and the pad of the j-th sample is a scalar.
template <class T>
__global__ __launch_bounds__(256) void fir_halve4_kernel(const T* __restrict__ x, T* __restrict__ out, const T* __restrict__ h, long long n_in, long long n_out, int blocks_per_clip, int n_taps,
                                                         int first, double div, double mul) {
    LRA_DYN_LDS(Pair2<T>, xs);  // entry e(s) = s + s / 8 holds the pair (x[s], x[s + 2]) of staged samples
    const long long clip = blockIdx.x / blocks_per_clip;
    const long long n0 = (long long)(blockIdx.x % blocks_per_clip) * 1024;
    const int t = threadIdx.x;
    const T* __restrict__ xc = x + clip * n_in;
    const long long base = (n0 + first) * 2 - (n_taps - 1);  // input index of staged sample 0
    const int span = 1023 * 2 + n_taps;
    // Staging: sample s is the first half of pair s and the second half of pair s - 2 (every sample is written twice, so that the walk below reads a
    // (sample, sample + 2) pair as ONE aligned 8-byte LDS read straight into a register pair).  The span's first 2 560 samples as ten loads per thread
    // issued together (index clamped into the clip, the value masked: nothing for a load to wait on), then the LDS writes -- with its runtime bound the
    // staging loop ran as nine dependent global round trips; longer filters finish in the loop below.
    auto put = [&](int s, T v) {
        xs[s + (s >> 3)].lo = v;
        if (s >= 2) xs[s - 2 + ((s - 2) >> 3)].hi = v;
    };
    {
        constexpr int IT = 10;
        T st[IT];
        const long long last = n_in > 0 ? n_in - 1 : 0;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const long long gi = base + t + it * 256;
            st[it] = n_in > 0 ? xc[gi < 0 ? 0 : (gi > last ? last : gi)] : (T)0;
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int s = t + it * 256;
            const long long gi = base + s;
            if (s < span) put(s, (gi >= 0 && gi < n_in) ? st[it] : (T)0);
        }
        for (int s = t + IT * 256; s < span; s += 256) {
            const long long gi = base + s;
            put(s, (gi >= 0 && gi < n_in) ? xc[gi] : (T)0);
        }
    }
    __syncthreads();
    const Pair2<T>* __restrict__ w = xs + t * 9;  // this thread's window: the pair (sample j, sample j + 2) of it at w[j + j / 8]
    // Output q of the thread takes window samples 2 q .. 2 q + n_taps - 1, sample 2 q + k with tap h[n_taps - 1 - k]: step k of ONE walk over the taps
    // feeds (v[k], v[k + 2], v[k + 4], v[k + 6]) h[n_taps - 1 - k] to the four running sums -- per sum the same products in the same (ascending-input)
    // order as scipy's upfirdn, one rounding per operation, and no edge steps.  As two (sample, sample + 2) pairs times one tap the step is two packed
    // multiplies and two packed adds (round 6: the halvings of a transform ran at the vector pipe's limit, eight instructions per step); the pair
    // (v[k + 4], v[k + 6]) of step k is step k + 4's first pair, so every step reads ONE new pair from the staged span (which holds every sample twice,
    // as the first half of its own pair and the second half of the pair two samples back: left to itself hipcc reads each sample once and splices the
    // pairs with moves, which cost what the packing saves).
    auto pair_at = [&](int j) { return w[j + (j >> 3)]; };
    Pair2<T> a01 = {(T)0, (T)0}, a23 = {(T)0, (T)0};
    Pair2<T> p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = pair_at(i);
    int k = 0;
    for (; k + 8 <= n_taps; k += 8) {
        const T* __restrict__ hp = h + (n_taps - 8 - k);  // taps of steps k + 7 (lowest index) ... k: the same for all lanes, scalar loads
        T hh[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) hh[i] = hp[i];
        Pair2<T> nw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) nw[u] = pair_at(k + u + 4);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a01 = fma2_exact(a01, u < 4 ? p[u] : nw[u - 4], hh[7 - u]);
            a23 = fma2_exact(a23, nw[u], hh[7 - u]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = nw[4 + i];
    }
    for (; k < n_taps; ++k) {
        const T hk = h[n_taps - 1 - k];
        const Pair2<T> nwp = pair_at(k + 4);
        a01 = fma2_exact(a01, p[0], hk);
        a23 = fma2_exact(a23, nwp, hk);
        p[0] = p[1]; p[1] = p[2]; p[2] = p[3]; p[3] = nwp;
    }
    const T a0 = a01.lo, a1 = a01.hi, a2 = a23.lo, a3 = a23.hi;
    const T acc[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long long n = n0 + (long long)t * 4 + q;
        if (n < n_out) {
            const T scaled = (T)((double)acc[q] / div);
            out[clip * n_out + n] = mul == 1.0 ? scaled : (T)((double)scaled * mul);
        }
    }
}

// ---- sparse basis projection: fft_basis.dot(D) of __cqt_response (constantq.py:1213-1218) and the length scaling (:1116-1118),
// written straight into the octave's rows of the stacked result (__trim_stack, :1168-1194) ------------------------------------------
// basis: CSR over the octave's filters (row_ptr[n_rows + 1], col[], val[]: complex, already times sqrt(sr / my_sr)); rows row0 ..
// row0 + n_rows - 1 are used (the lowest octave may keep only its highest filters).  D: [clip][frames_in][n_bins];  out:
// [clip][n_frames][n_total], this octave at columns bin0 ...  sqrt_len: sqrt(length) per used row (float64) or nullptr
// (scale=False); the reference divides the complex64 result by that float64 array, which NumPy evaluates in complex128 as a
// multiplication by the reciprocal (Smith's division with a zero imaginary divisor) before rounding back.
// A thread per (clip, frame, row), rows fastest: the lanes of a wave share a handful of D rows (L1) and write contiguous outputs.
template <class T>
__global__ __launch_bounds__(256) void cqt_project_kernel(const CqtCplx<T>* __restrict__ D, CqtCplx<T>* __restrict__ out, const int* __restrict__ row_ptr, const int* __restrict__ col,
                                                          const CqtCplx<T>* __restrict__ val, const double* __restrict__ sqrt_len, long long batch, long long frames_in, int n_bins,
                                                          long long n_frames, int n_total, int bin0, int row0, int n_rows) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= batch * n_frames * n_rows) return;
    const int r = (int)(id % n_rows);
    const long long t = (id / n_rows) % n_frames;
    const long long clip = id / ((long long)n_rows * n_frames);
    const CqtCplx<T>* __restrict__ d = D + (clip * frames_in + t) * n_bins;
    CqtCplx<T> acc;
    acc.x = (T)0;
    acc.y = (T)0;
    const int j1 = row_ptr[row0 + r + 1];
    for (int j = row_ptr[row0 + r]; j < j1; ++j) acc = CqtOps<T>::cmadd(acc, val[j], d[col[j]]);
    if (sqrt_len) {
        const double scl = 1.0 / sqrt_len[r];
        acc.x = (T)((double)acc.x * scl);
        acc.y = (T)((double)acc.y * scl);
    }
    out[(clip * n_frames + t) * n_total + bin0 + r] = acc;
}

}  // namespace lra
