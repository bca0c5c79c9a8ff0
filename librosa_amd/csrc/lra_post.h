// lra_post.h -- the consumers right after the mel reduce: decibel scaling and MFCC.
//
//   power_to_db / amplitude_to_db   librosa/core/spectrum.py:1735-1883, 1946-2038
//       log_spec = 10 log10(max(amin, S)) - 10 log10(max(amin, ref)),  then  max(log_spec, log_spec.max(axes) - top_db)
//   db_to_power / db_to_amplitude   librosa/core/spectrum.py:1898-1925, 2054-2082
//   mfcc                            librosa/feature/spectral.py:1843-2019:  scipy.fft.dct(S_db, axis=-2, type, norm)[..., :n_mfcc, :] (+ lifter)
//
// All of them are HBM-bound elementwise / small-contraction work over [batch][per_item] or [batch][n_in][n_frames] arrays:
// lanes run along the contiguous axis (16-byte accesses where the layout allows), reductions are per item (the
// reference's axes="auto": the last two axes), the DCT basis is a host-built table (float64 recipe, exactly scipy's
// transform applied to the identity) that every lane reads through the scalar cache.
#pragma once

#include <hip/hip_runtime.h>

namespace lra {

// ---- per-item maximum of |x| (the reduction behind ref=np.max and top_db) --------------------------------------------
// Non-negative IEEE values order like their bit patterns, so the cross-workgroup combine is an integer atomicMax.
template <class T> struct MaxBits;
template <> struct MaxBits<float> {
    using type = unsigned int;
    static __device__ type bits(float v) { return __float_as_uint(v); }
};
template <> struct MaxBits<double> {
    using type = unsigned long long;
    static __device__ type bits(double v) { return (unsigned long long)__double_as_longlong(v); }
};

// ABS: max |x| (amplitude_to_db takes np.abs of every input, core/spectrum.py:2011); otherwise max(0, max x): power_to_db leaves real
// input signed (:1855-1859), and everything downstream takes max(amin, .) of the maximum with amin > 0, so clamping it at zero is
// exact and keeps the bit-pattern ordering of the combine.
template <class T, bool ABS> __global__ __launch_bounds__(256) void item_absmax_kernel(const T* __restrict__ x, long long per_item, int chunks_per_item, typename MaxBits<T>::type* __restrict__ out) {
    const long long item = blockIdx.x / chunks_per_item;
    const int chunk = blockIdx.x % chunks_per_item;
    const T* __restrict__ xi = x + item * per_item;
    const long long per_chunk = (per_item + chunks_per_item - 1) / chunks_per_item;
    const long long lo = chunk * per_chunk, hi = lo + per_chunk < per_item ? lo + per_chunk : per_item;
    T m = (T)0;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const T v = xi[i] < (T)0 ? (ABS ? -xi[i] : (T)0) : xi[i];
        m = v > m ? v : m;  // NaN never wins: the reference's np.max would propagate it, but NaN power is rejected upstream
    }
    __shared__ T red[256];
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = red[threadIdx.x + s] > red[threadIdx.x] ? red[threadIdx.x + s] : red[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax(&out[item], MaxBits<T>::bits(red[0]));
}

// ---- decibel scaling ---------------------------------------------------------------------------------------------------
// AMP: the input is an amplitude (amplitude_to_db squares it first, in the input precision, exactly as the reference does:
// power_to_db(S**2, ref=ref**2, amin=amin**2), core/spectrum.py:2030-2037)
template <class T> __device__ __forceinline__ T ten_log10(T v) { return (T)10 * log10(v); }
template <> __device__ __forceinline__ float ten_log10<float>(float v) { return 10.0f * log10f(v); }

template <class T> struct DbArgs {
    T amin;               // power domain (the host squares amplitude_to_db's amin)
    T ref_scalar;         // |ref| when ref_items == nullptr (input domain: squared here for amplitudes)
    const T* ref_items;   // per-item |reference| values (input domain), or nullptr
    const T* item_max;    // per-item max of |x| (input domain), or nullptr when top_db is None
    T top_db;
};

template <class T> __device__ __forceinline__ T db_of(T mag, T amin, T ref_db) { return ten_log10<T>(mag > amin ? mag : amin) - ref_db; }

template <class T> __device__ __forceinline__ void db_item_constants(const DbArgs<T>& d, long long item, bool amp, T& ref_db, T& floor_db) {
    T ref = d.ref_items ? d.ref_items[item] : d.ref_scalar;
    if (amp) ref = ref * ref;
    ref_db = ten_log10<T>(ref > d.amin ? ref : d.amin);
    floor_db = -INFINITY;
    if (d.item_max) {
        T mx = d.item_max[item];
        if (amp) mx = mx * mx;
        floor_db = db_of<T>(mx, d.amin, ref_db) - d.top_db;  // log10 is monotone: max of the logs = log of the max
    }
}

template <class T, bool AMP> __global__ __launch_bounds__(256) void to_db_kernel(const T* __restrict__ x, T* __restrict__ out, long long per_item, int chunks_per_item, DbArgs<T> d) {
    const long long item = blockIdx.x / chunks_per_item;
    const int chunk = blockIdx.x % chunks_per_item;
    T ref_db, floor_db;
    db_item_constants<T>(d, item, AMP, ref_db, floor_db);
    const long long per_chunk = (per_item + chunks_per_item - 1) / chunks_per_item;
    const long long lo = chunk * per_chunk, hi = lo + per_chunk < per_item ? lo + per_chunk : per_item;
    const T* __restrict__ xi = x + item * per_item;
    T* __restrict__ oi = out + item * per_item;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        T v = xi[i];
        if (AMP) v = v * v;  // (power_to_db does NOT take |x| of real input: negative values floor at amin, core/spectrum.py:1855-1873)
        const T db = db_of<T>(v, d.amin, ref_db);
        oi[i] = db > floor_db ? db : floor_db;
    }
}

// db_to_power: ref * 10^(0.1 x); db_to_amplitude: db_to_power(x, ref = ref^2) ^ 0.5 (core/spectrum.py:1925, 2082)
template <class T, bool AMP> __global__ __launch_bounds__(256) void from_db_kernel(const T* __restrict__ x, T* __restrict__ out, long long count, T ref) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
        const T p = (AMP ? ref * ref : ref) * pow((T)10, x[i] * (T)0.1);
        out[i] = AMP ? sqrt(p) : p;
    }
}

// ---- DCT over the band axis (MFCC) -------------------------------------------------------------------------------------
// out[b][k][t] = lift[k] * sum_m C[k][m] * f(S[b][m][t]),  k < n_out <= NOUT,  f = identity or the decibel scaling above
// (DB: the fused mfcc(y=...) path reads the mel POWER spectrogram once and never materialises power_to_db's output).
// One thread per frame t (lanes along the contiguous axis: every S row access is a 256-byte coalesced wave load), NOUT
// accumulators in registers, the basis C through uniform (scalar-cache) loads: 2 n_in NOUT flops per 4 n_in bytes read.  The
// basis is stored band-major ([n_in][ldc]) so that the NOUT coefficients of one band are contiguous (s_load_dwordx8/x16);
// coefficient-major storage made every one of them its own scalar load.
template <class T, int NOUT, bool DB>
__global__ __launch_bounds__(256) void dct_rows_kernel(const T* __restrict__ S, T* __restrict__ out, long long n_frames, int n_in, int n_out /* rows of this launch */, int out_rows /* rows of `out` per item */,
                                                       const T* __restrict__ C /* [n_in][ldc]: column k of row m = basis[k][m], zero columns beyond n_out */, int ldc,
                                                       const T* __restrict__ lift /* [n_out] */, DbArgs<T> d) {
    const long long tblocks = (n_frames + 255) / 256;
    const long long b = blockIdx.x / tblocks;
    const long long t = (blockIdx.x % tblocks) * 256 + threadIdx.x;
    T ref_db = (T)0, floor_db = -INFINITY;
    if (DB) db_item_constants<T>(d, b, false, ref_db, floor_db);
    if (t >= n_frames) return;
    const T* __restrict__ s = S + b * (long long)n_in * n_frames + t;
    T acc[NOUT];
#pragma unroll
    for (int k = 0; k < NOUT; ++k) acc[k] = (T)0;
    for (int m = 0; m < n_in; ++m) {
        T v = s[(long long)m * n_frames];
        if (DB) {
            const T db = db_of<T>(v, d.amin, ref_db);
            v = db > floor_db ? db : floor_db;
        }
#pragma unroll
        for (int k = 0; k < NOUT; ++k) acc[k] += C[(long long)m * ldc + k] * v;  // NOUT contiguous coefficients per band: wide scalar loads
    }
    T* __restrict__ o = out + b * (long long)out_rows * n_frames + t;
#pragma unroll
    for (int k = 0; k < NOUT; ++k)
        if (k < n_out) o[(long long)k * n_frames] = acc[k] * lift[k];
}

// ---- Griffin-Lim phase update -----------------------------------------------------------------------
//   librosa/core/spectrum.py:2896-2902:  angles = rebuilt - momentum/(1+momentum) tprev;  angles /= |angles| + eps;  angles *= S
// Elementwise over `count` complex values in any common layout (the arrays here are [batch][frame][bin]); 28 bytes per value.
// The arithmetic follows NumPy's complex64 / complex128 loops operation by operation (no contraction into fma, which would
// change the rounding): coef * tprev per component, the subtraction, |.| as the correctly rounded hypot (double-width square
// root for f32; NumPy's own float32 hypot differs from it in the last bit depending on the host's SIMD dispatch), complex /
// real as multiplication by the reciprocal 1 / (|.| + eps) (NumPy's Smith division with a zero imaginary divisor), then the
// two products with S.
// hipcc contracts a * b - c into fma by default (-ffp-contract=fast): the operations below are emitted with contraction off,
// and only they.
#pragma clang fp contract(off)
template <class T> struct Cplx2 { T x, y; };
template <class T> __device__ __forceinline__ T gl_hypot(T a, T b);
template <> __device__ __forceinline__ float gl_hypot<float>(float a, float b) { return (float)sqrt((double)a * (double)a + (double)b * (double)b); }  // exact products, one rounding in the sum
template <> __device__ __forceinline__ double gl_hypot<double>(double a, double b) { return hypot(a, b); }
// one rounding per operation: plain operators compiled with contraction off (ROCm's __fmul_rn / __fsub_rn are plain operators
// in a header compiled with contraction ON, so they fuse)
template <class T> struct RnOps {
    static __device__ __forceinline__ T mul(T a, T b) { return a * b; }
    static __device__ __forceinline__ T sub(T a, T b) { return a - b; }
    static __device__ __forceinline__ T add(T a, T b) { return a + b; }
    static __device__ __forceinline__ T div(T a, T b) { return a / b; }
};
#pragma clang fp contract(fast)

// NORMALIZE = false: angles = rebuilt * S only (the initial "absorb magnitudes into angles", :2847)
template <class T, bool NORMALIZE>
__global__ __launch_bounds__(256) void griffinlim_update_kernel(const Cplx2<T>* __restrict__ rebuilt, const Cplx2<T>* __restrict__ tprev /* or nullptr */, const T* __restrict__ S,
                                                                Cplx2<T>* angles /* may alias rebuilt */, long long count, T coef, T eps) {
    using R = RnOps<T>;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
        Cplx2<T> a = rebuilt[i];
        if (NORMALIZE) {
            if (tprev) {
                const Cplx2<T> t = tprev[i];
                a.x = R::sub(a.x, R::mul(coef, t.x));
                a.y = R::sub(a.y, R::mul(coef, t.y));
            }
            const T scl = R::div((T)1, R::add(gl_hypot<T>(a.x, a.y), eps));
            a.x = R::mul(a.x, scl);
            a.y = R::mul(a.y, scl);
        }
        const T s = S[i];
        a.x = R::mul(a.x, s);
        a.y = R::mul(a.y, s);
        angles[i] = a;
    }
}

// Initial phase estimate (:2832-2847): angles = S * (cos, sin)(2 pi u), u = the caller's uniform draws in [0, 1) (float64, drawn
// on the host so that `rng` reproduces the reference's stream).  The phasor is evaluated in float64 like util.phasor
// (util/utils.py:2629-2637) and rounded once to the working precision.
template <class T> __global__ __launch_bounds__(256) void griffinlim_init_kernel(const double* __restrict__ u, const T* __restrict__ S, Cplx2<T>* __restrict__ angles, long long count) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
        const double a = 6.283185307179586 * u[i];  // 2 * np.pi * rng.random(...)
        double sn, cs;
        sincos(a, &sn, &cs);
        const T s = S[i];
        Cplx2<T> z;
        z.x = (T)cs * s;
        z.y = (T)sn * s;
        angles[i] = z;
    }
}

// ---- phase vocoder (librosa/core/spectrum.py:1459-1519) ----------------------------------------------
// One thread per (clip, bin), lanes along the contiguous bin axis of the [batch][frame][bin] layout: every frame access is a
// coalesced wave load / store.  The thread walks the output frames in order, because the phase is a running sum along time
// (np.cumsum in the spectrum's real precision, :1507): phase_0 = angle(D[i0_0]); phase_t = phase_{t-1} + angle(D[i1_{t-1}]) -
// angle(D[i0_{t-1}]) with i0 = floor(t_out), i1 = min(i0 + 1, n - 1) (:1491-1504).  The magnitude is scipy's linear interp1d of
// |D| (:1507-1515): between the frames lo = clip(searchsorted(t), 1, n - 1) - 1 and lo + 1, evaluated in float64 as
// slope * (t - lo) + |D[lo]|, and the phasor multiply is NumPy's complex64 *= float64 (in double, rounded once).
template <class T> struct VocoderStep {
    int i0, i1, lo;    // phase frames; lower interpolation frame
    double frac;       // t - lo
};

template <class T>
__global__ __launch_bounds__(256) void phase_vocoder_kernel(const Cplx2<T>* __restrict__ D, Cplx2<T>* __restrict__ out, const VocoderStep<T>* __restrict__ steps, long long n_in, long long n_out,
                                                            int n_bins, long long batch) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= batch * n_bins) return;
    const long long b = id / n_bins;
    const int k = (int)(id % n_bins);
    const Cplx2<T>* __restrict__ d = D + b * n_in * n_bins + k;
    Cplx2<T>* __restrict__ o = out + b * n_out * n_bins + k;
    T phase = (T)0;
    for (long long t = 0; t < n_out; ++t) {
        const VocoderStep<T> st = steps[t];  // uniform: scalar loads
        const Cplx2<T> a = d[(long long)st.i0 * n_bins], c = d[(long long)st.i1 * n_bins];
        const T pa = atan2(a.y, a.x), pc = atan2(c.y, c.x);
        if (t == 0) phase = pa;
        const Cplx2<T> l = d[(long long)st.lo * n_bins], h = d[(long long)(st.lo + 1) * n_bins];
        const double ml = (double)gl_hypot<T>(l.x, l.y), mh = (double)gl_hypot<T>(h.x, h.y);
        const double mag = (mh - ml) * st.frac + ml;
        const T sn = sin(phase), cs = cos(phase);
        Cplx2<T> z;
        z.x = (T)((double)cs * mag);
        z.y = (T)((double)sn * mag);
        o[t * n_bins] = z;
        phase += pc - pa;  // consumed by the next frame (phase[1:] = diff[:-1], :1504)
    }
}

}  // namespace lra
