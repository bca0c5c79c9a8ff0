// lra_pcen.h -- per-channel energy normalisation and its band max-filter (the consumer of the streaming STFT,
// docs/examples/plot_pcen_stream.py:71-80).  Self-contained so that tests/hostsim/postsim.cpp can run the same kernel bodies on
// host threads (-DLRA_POSTSIM: one OS thread per lane, __syncthreads = a barrier).
#pragma once

#ifndef LRA_POSTSIM
#include <hip/hip_runtime.h>
#endif

namespace lra {

// one rounding per operation (hipcc contracts a * b + c into fma by default; scipy's lfilter loop and NumPy's ufuncs do not)
#pragma clang fp contract(off)
struct PcenOps {
    static __device__ __forceinline__ double mul(double a, double b) { return a * b; }
    static __device__ __forceinline__ double sub(double a, double b) { return a - b; }
    static __device__ __forceinline__ double add(double a, double b) { return a + b; }
};
#pragma clang fp contract(fast)

// ---- PCEN, per-channel energy normalisation: librosa/core/spectrum.py:2396-2666 (the consumer of the
// streaming stft in docs/examples/plot_pcen_stream.py:71-80) ----------------------------------------------------------------
//   M      = lfilter([b], [1, b - 1], ref, zi, axis=time)                        first-order smoother, float64 state (:2649-2655)
//   smooth = exp(-gain * (log(eps) + log1p(M / eps)))                             (:2658)
//   out    = log1p(S * smooth)                                   power == 0       (:2661)
//          = exp(power * (log(S) + log(smooth)))                 bias == 0        (:2663; log(S) in S's own precision, as NumPy does)
//          = bias**power * expm1(power * log1p(S * smooth / bias))  otherwise     (:2665)
// The reference's result is float64 whatever S is (its zi is float64, which promotes the filter and everything after it).
// Rows ([clip x band], time contiguous) are independent; the recurrence along time is sequential.  One wave owns kPcenRows rows:
// a tile of kPcenTile frames is loaded with lanes along time (coalesced 256-byte row segments), widened to float64 in LDS, the
// first kPcenRows lanes run the recurrence in place (scipy's transposed direct form II, one rounding per operation: y = z + b x,
// z = 0 x - a1 y), and the elementwise part runs with lanes along time again.  The float64 transcendentals dominate
// (~5 per element), so the sequential step costs a few percent.
constexpr int kPcenRows = 16, kPcenTile = 64;

struct PcenArgs {
    double b, a1;        // numerator / second denominator coefficient (b - 1)
    double zi_scalar;    // initial state when no per-row state is given: scipy.signal.lfilter_zi([b], [1, b - 1]), from the host
    double neg_gain, log_eps, eps, power, bias, bias_pow;  // bias_pow = bias**power (host)
    int mode;            // 0: power == 0, 1: bias == 0, 2: general
};

template <class T> __device__ __forceinline__ double pcen_log_in_precision(T s);
template <> __device__ __forceinline__ double pcen_log_in_precision<float>(float s) { return (double)logf(s); }
template <> __device__ __forceinline__ double pcen_log_in_precision<double>(double s) { return log(s); }

template <class T>
__global__ __launch_bounds__(64) void pcen_kernel(const T* __restrict__ S, const T* __restrict__ ref, double* __restrict__ out, long long rows, long long n_frames, PcenArgs p,
                                                  const double* __restrict__ zi /* [rows] or nullptr */, double* __restrict__ zf /* [rows] or nullptr */) {
    using R = PcenOps;
    __shared__ double sm[kPcenRows][kPcenTile + 1];  // row pitch 65 doubles: the 16 lanes of the recurrence hit distinct banks
    const int lane = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * kPcenRows;
    const int nrows = (int)(rows - row0 < kPcenRows ? rows - row0 : kPcenRows);
    double z = 0;
    if (lane < nrows) z = zi ? zi[row0 + lane] : p.zi_scalar;
    for (long long f0 = 0; f0 < n_frames; f0 += kPcenTile) {
        const int nf = (int)(n_frames - f0 < kPcenTile ? n_frames - f0 : kPcenTile);
        if (lane < nf)
            for (int r = 0; r < nrows; ++r) sm[r][lane] = (double)ref[(row0 + r) * n_frames + f0 + lane];
        __syncthreads();
        if (lane < nrows) {
#pragma unroll 8
            for (int i = 0; i < nf; ++i) {
                const double x = sm[lane][i];
                const double y = R::add(z, R::mul(p.b, x));
                z = R::sub(R::mul(x, 0.0), R::mul(y, p.a1));
                sm[lane][i] = y;
            }
        }
        __syncthreads();
        if (lane < nf)
            for (int r = 0; r < nrows; ++r) {
                const long long at = (row0 + r) * n_frames + f0 + lane;
                const T s_in = S[at];
                const double s = (double)s_in;
                const double m = sm[r][lane];
                const double lsm = R::mul(p.neg_gain, R::add(p.log_eps, log1p(m / p.eps)));
                const double smooth = exp(lsm);
                double v;
                if (p.mode == 0)
                    v = log1p(R::mul(s, smooth));
                else if (p.mode == 1)
                    v = exp(R::mul(p.power, R::add(pcen_log_in_precision<T>(s_in), log(smooth))));
                else
                    v = R::mul(p.bias_pow, expm1(R::mul(p.power, log1p(R::mul(s, smooth) / p.bias))));
                out[at] = v;
            }
        __syncthreads();
    }
    if (zf && lane < nrows) zf[row0 + lane] = z;
}

// scipy.ndimage.maximum_filter1d(S, size, axis=band axis) with its default mode="reflect" (half-sample symmetric: d c b a | a b c d | d c b a)
// and origin 0: out[m] = max(S[m - size // 2 .. m - size // 2 + size - 1]) (librosa/core/spectrum.py:2640-2642).  [outer][n_bands][inner],
// inner contiguous; a thread per element, lanes along inner.
template <class T>
__global__ __launch_bounds__(256) void maxfilter_bands_kernel(const T* __restrict__ S, T* __restrict__ out, long long outer, int n_bands, long long inner, int size) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= outer * n_bands * inner) return;
    const long long t = id % inner;
    const int m = (int)((id / inner) % n_bands);
    const long long o = id / (inner * n_bands);
    const T* __restrict__ base = S + o * n_bands * inner + t;
    const int period = 2 * n_bands;
    int idx = (m - size / 2) % period;
    if (idx < 0) idx += period;
    T v = base[(long long)(idx < n_bands ? idx : period - 1 - idx) * inner];
    for (int j = 1; j < size; ++j) {
        idx = idx + 1 == period ? 0 : idx + 1;
        const T c = base[(long long)(idx < n_bands ? idx : period - 1 - idx) * inner];
        v = c > v ? c : v;
    }
    out[id] = v;
}

}  // namespace lra
