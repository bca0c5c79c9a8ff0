// lra_rng.h -- NumPy's default generator on the device: PCG64 (PCG XSL RR 128/64) with jump-ahead, bit for bit.
//
// librosa.griffinlim draws its initial phases as `rng.random(size=S.shape)` (librosa/core/spectrum.py:2832-2847) with
// rng = np.random.default_rng(seed): one float64 per spectrogram element, in the element order of S (C order, (..., bin, frame)).  For 32 clips
// of 30 s that is 42 M draws: 92 ms on a host core plus their upload, against 13 ms for all 32 device-resident iterations.
// The generator is a 128-bit linear congruential generator -- state' = state * M + inc (mod 2^128), output = rotr64(hi ^ lo, hi >> 58) of the NEW
// state, double = (output >> 11) * 2^-53 (numpy/random/src/pcg64/pcg64.h: pcg64_next64, pcg64_next_double) -- and an LCG can be advanced by n steps
// in O(log n) (Brown, "Random number generation with arbitrary strides"; pcg_advance_lcg_128): every thread jumps to the first draw of its own
// run of consecutive draws and then steps.  Integer arithmetic only, so the device stream is THE NumPy stream, bit for bit
// (tests: np.array_equal against np.random.default_rng(seed).random(n) in the CPU suite through tests/hostsim/postsim.cpp and on the device).
//
// Self-contained (threadIdx / blockIdx only): also compiled into the CPU thread simulator.
#pragma once

#include <cstdint>

namespace lra {
namespace rng {

typedef unsigned __int128 u128;

struct Pcg64 {
    uint64_t state_hi, state_lo, inc_hi, inc_lo;  // np.random.Generator(PCG64).bit_generator.state["state"]: {"state", "inc"} as 128-bit integers
};

__device__ __forceinline__ u128 make128(uint64_t hi, uint64_t lo) { return ((u128)hi << 64) | (u128)lo; }
// PCG_DEFAULT_MULTIPLIER_128
__device__ __forceinline__ u128 pcg_mult() { return make128(2549297995355413924ULL, 4865540595714422341ULL); }

// state after `delta` further steps (pcg_advance_lcg_128)
__device__ __forceinline__ u128 pcg_advance(u128 state, u128 inc, uint64_t delta) {
    u128 cur_mult = pcg_mult(), cur_plus = inc, acc_mult = 1, acc_plus = 0;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    return acc_mult * state + acc_plus;
}

// one draw: step, then the XSL-RR output of the new state, as NumPy's next_double
__device__ __forceinline__ double pcg_next_double(u128& state, u128 inc) {
    state = state * pcg_mult() + inc;
    const uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state;
    const uint64_t x = hi ^ lo;
    const unsigned rot = (unsigned)(hi >> 58);
    const uint64_t out = (x >> rot) | (x << ((64u - rot) & 63u));
    return (double)(out >> 11) * (1.0 / 9007199254740992.0);
}

constexpr int kRunLength = 128;  // consecutive draws per thread: the jump costs ~2 x 27 128-bit products for offsets of 10^8, a step one

// out[i] = draw number offset + i of the generator (i.e. rng.random(offset + count)[offset:]), 0 <= i < count
__global__ __launch_bounds__(256) void pcg64_uniform_kernel(Pcg64 g, uint64_t offset, double* __restrict__ out, long long count) {
    const long long run = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long first = run * kRunLength;
    if (first >= count) return;
    const u128 inc = make128(g.inc_hi, g.inc_lo);
    u128 state = pcg_advance(make128(g.state_hi, g.state_lo), inc, offset + (uint64_t)first);
    const long long last = first + kRunLength < count ? first + kRunLength : count;
    for (long long i = first; i < last; ++i) out[i] = pcg_next_double(state, inc);
}

template <class T> struct RngCplx {
    T x, y;
};

// Griffin-Lim's initial estimate in one pass (librosa/core/spectrum.py:2832-2847): angles = S * exp(2 pi i u), u = rng.random(S.shape) in S's own
// element order (clip, bin, frame), S and angles in the device layout [clip][frame][bin].  A thread owns one (clip, bin) row segment of `seg`
// consecutive frames = consecutive draws; lanes run along the bins, so every load of S and store of angles is a coalesced row piece.  The phasor is
// evaluated in float64 like util.phasor (util/utils.py:2629-2637) and rounded once -- the arithmetic of griffinlim_init_kernel (lra_post.h).
template <class T>
__global__ __launch_bounds__(256) void griffinlim_init_pcg64_kernel(Pcg64 g, const T* __restrict__ S, RngCplx<T>* __restrict__ angles, long long batch, int n_bins, long long n_frames, int seg,
                                                                    int bin_blocks) {
    // block -> (clip, frame segment, block of 256 bins)
    long long b = blockIdx.x;
    const int kb = (int)(b % bin_blocks);
    b /= bin_blocks;
    const long long segs = (n_frames + seg - 1) / seg;
    const long long sg = b % segs;
    const long long clip = b / segs;
    const int k = kb * 256 + (int)threadIdx.x;
    if (clip >= batch || k >= n_bins) return;
    const long long t0 = sg * seg;
    const long long t1 = t0 + seg < n_frames ? t0 + seg : n_frames;
    const u128 inc = make128(g.inc_hi, g.inc_lo);
    u128 state = pcg_advance(make128(g.state_hi, g.state_lo), inc, (uint64_t)((clip * n_bins + k) * n_frames + t0));
    for (long long t = t0; t < t1; ++t) {
        const double a = 6.283185307179586 * pcg_next_double(state, inc);  // 2 * np.pi * rng.random(...)
        double sn, cs;
        sincos(a, &sn, &cs);
        const long long i = (clip * n_frames + t) * n_bins + k;
        const T s = S[i];
        RngCplx<T> z;
        z.x = (T)cs * s;
        z.y = (T)sn * s;
        angles[i] = z;
    }
}

}  // namespace rng
}  // namespace lra
