// lra_fft.h -- power-of-two complex FFT machinery for one frame held in LDS.
//
// A real frame of N = 2M samples is transformed through an M-point complex FFT of
// z[n] = x[2n] + i x[2n+1] (the standard even/odd packing), followed by a split step.  The
// M-point FFT is a Stockham autosort FFT with P passes of radix 2^logr(p) <= R; every thread
// keeps R complex points in registers, so one frame occupies TF = M/R threads and a workgroup
// of NT threads transforms FPB = NT/TF frames at once (one wave64 per frame at n_fft = 2048,
// R = 16).  Between passes the data round-trips through LDS; the first pass reads straight from
// global memory (window multiply fused), so per frame the LDS sees P writes + P reads of 8M bytes.
//
// LDS layout of one frame: complex element i lives at slot phys(i) = i + (i >> logr(0)).  The
// one-slot pad every r0 elements makes the first pass's stride-r0 stores conflict-free
// (stride r0+1 slots = 2(r0+1) dwords: the 16 lanes of a ds_write_b64 group hit 16 distinct
// even banks) while keeping the later unit-stride accesses essentially contiguous.
//
// Reference semantics implemented by the callers (lra_kernels.h): librosa/core/spectrum.py:
// 380-390 (rfft of windowed frames) and :598 (irfft of spectrogram columns).
#pragma once

#include <cmath>

#include "lra_common.h"

namespace lra {

// MINW_: occupancy target handed to __launch_bounds__ (waves per SIMD; 0 = default).  HOIST_: let
// the compiler keep the per-thread window/twiddle values in registers across the frame loop (they
// are loop-invariant); when false the table pointers are laundered every iteration so the tables
// are re-read from L1/L2 and the register budget stays small.
// REV_: the radices in ascending order (8, 8, 16 instead of 16, 8, 8 at M = 1024): the inverse kernel wants the
// two-butterfly pass FIRST (its Hermitian step is fused into the first pass, see istft_unsplit_pass0) and the one-butterfly
// radix-16 pass last (fused with the overlap-add).
// PLAN_: how LOGM is dealt to the passes.  0: balanced (16, 8, 8 at M = 1024).  1: largest radix first, the remainder last (16, 16, 4 at M = 1024): the
// last pass then has FOUR butterflies per thread, and a thread that takes butterflies {2t, 2t + 1} and their mirrors {s - 2t - 1, s - 2t} ends up with
// the bins 2t, 2t + 1 (+ j s) of every mirrored pair side by side in its registers -- 16-byte pieces of the spectrum row without any exchange (lra_kernels3.h).
template <int LOGM_, int LOGR_, class T_, int NTMIN_ = 256, int MINW_ = 0, bool HOIST_ = true, bool REV_ = false, int PLAN_ = 0> struct FftCfg {
    using real = T_;
    using cplx = cx<T_>;
    static constexpr int LOGM = LOGM_;
    static constexpr int M = 1 << LOGM_;  // complex points per frame
    static constexpr int N = 2 * M;       // n_fft
    static constexpr int LOGR = LOGR_ < LOGM_ ? LOGR_ : LOGM_;
    static constexpr int R = 1 << LOGR;  // complex points per thread
    static constexpr int TF = M / R;     // threads per frame
    static constexpr int NT = TF > NTMIN_ ? TF : NTMIN_;
    static constexpr int FPB = NT / TF;  // frames per workgroup iteration
    static constexpr int P = (LOGM + LOGR - 1) / LOGR;
    static constexpr bool REV = REV_;
    static constexpr int PLAN = PLAN_;
    static constexpr int logr(int p) {
        if (PLAN_ == 1 && REV_) return p == 0 ? LOGM - (P - 1) * LOGR : LOGR;   // ascending: the remainder first (4, 16, 16 at M = 1024: the inverse kernel's form)
        if (PLAN_ == 1) return (p + 1) * LOGR <= LOGM ? LOGR : LOGM - p * LOGR;
        return LOGM / P + ((REV_ ? p >= P - LOGM % P : p < LOGM % P) ? 1 : 0);
    }
    static constexpr int logs(int p) {
        int s = 0;
        for (int q = 0; q < p; ++q) s += logr(q);
        return s;
    }
    // offset (complex elements) of pass p's twiddle block; pass 0 has none
    static constexpr int tw_off(int p) {
        int o = 0;
        for (int q = 1; q < p; ++q) o += ((1 << logr(q)) - 1) << logs(q);
        return o;
    }
    static constexpr int TW_TOTAL = tw_off(P) > 0 ? tw_off(P) : 1;
    // per-thread register copies of the pass twiddles (HOIST).  Only the power-of-two powers
    // w^1, w^2, w^4, ... of each butterfly's base twiddle are kept (log2 r values instead of r - 1);
    // the others are one complex multiply away: w^j = w^(j - msb j) * w^(msb j)
    static constexpr int treg_off(int p) {
        int o = 0;
        for (int q = 1; q < p; ++q) o += (R >> logr(q)) * logr(q);
        return o;
    }
    static constexpr int TREG_TOTAL = treg_off(P) > 0 ? treg_off(P) : 1;
    static constexpr int PADSHIFT = logr(0);
    static constexpr int FRAME_ELEMS = M + (M >> PADSHIFT) + 1;
    static constexpr int FRAME_BYTES = ((FRAME_ELEMS * (int)sizeof(cplx) + 15) / 16) * 16;
    static constexpr int phys(int i) { return i + (i >> PADSHIFT); }
    static constexpr int MIN_WAVES = MINW_ > 0 ? MINW_ : 2;
    static constexpr bool HOIST = HOIST_;
    // same transform, different workgroup size (the mel kernel shares its filter tables across slots)
    template <int NT2> using with_nt = FftCfg<LOGM_, LOGR_, T_, NT2, MINW_, HOIST_, REV_, PLAN_>;
    // a frame slot (TF threads) never spans two waves: slot-private LDS traffic needs no s_barrier
    static constexpr bool WAVE_SYNC = TF <= 64;
};

// ----------------------------------------------------------------------------- small DFTs
// W16^idx = exp(-2 pi i idx / 16), idx in [0, 8)
template <class T> LRA_HD cx<T> mul_w16(cx<T> a, int idx) {
    const T h = (T)0.70710678118654752440;
    const T c1 = (T)0.92387953251128675613, s1 = (T)0.38268343236508977173;
    switch (idx) {
        case 0: return a;
        case 4: return cmul_mi(a);
        case 2: return mk<T>((a.x + a.y) * h, (a.y - a.x) * h);
        case 6: return mk<T>((a.y - a.x) * h, -(a.x + a.y) * h);
        case 1: return mk<T>(a.x * c1 + a.y * s1, a.y * c1 - a.x * s1);
        case 3: return mk<T>(a.x * s1 + a.y * c1, a.y * s1 - a.x * c1);
        case 5: return mk<T>(a.y * c1 - a.x * s1, -(a.x * c1 + a.y * s1));
        default: /* 7 */ return mk<T>(a.y * s1 - a.x * c1, -(a.x * s1 + a.y * c1));
    }
}

// One radix-2 butterfly of the recursion below with the twiddle W16^idx folded in:
// p = e + W16^idx o,  m = e - W16^idx o.  Every case is 2..4 packed instructions (lra_common.h).
template <int idx, class T> LRA_HD void bfly_w16(cx<T> e, cx<T> o, cx<T>& p, cx<T>& m) {
    const T h = (T)0.70710678118654752440;
    const T c1 = (T)0.92387953251128675613, s1 = (T)0.38268343236508977173;
    if constexpr (idx == 0) {
        p = cadd(e, o);
        m = csub(e, o);
    } else if constexpr (idx == 4) {  // W = -i
        p = add_mi(e, o);
        m = sub_mi(e, o);
    } else if constexpr (idx == 2) {  // W = h (1 - i):  W o = h u,  u = o - i o
        const cx<T> u = add_mi(o, o);
        p = axpy(h, u, e);
        m = axpy(-h, u, e);
    } else if constexpr (idx == 6) {  // W = -i h (1 - i):  W o = h (-i u)
        const cx<T> u = add_mi(o, o);
        p = axpy_mi(h, u, e);
        m = axpy_pi(h, u, e);
    } else if constexpr (idx == 1 || idx == 3) {  // W = c - i s:  W o = c o + s (-i o)
        const T c = idx == 1 ? c1 : s1, s = idx == 1 ? s1 : c1;
        p = axpy_mi(s, o, axpy(c, o, e));
        m = axpy_pi(s, o, axpy(-c, o, e));
    } else {  // idx 5, 7:  W = -i (c - i s):  W o = c (-i o) - s o
        const T c = idx == 5 ? c1 : s1, s = idx == 5 ? s1 : c1;
        p = axpy_mi(c, o, axpy(-s, o, e));
        m = axpy_pi(c, o, axpy(s, o, e));
    }
}

// In-place forward DFT of r points (natural order in, natural order out), r in {1,2,4,8,16}.
template <int r, class T> struct Dft {
    template <int k> static LRA_HD void stage(cx<T>* v, const cx<T>* e, const cx<T>* o) {
        if constexpr (k < r / 2) {
            bfly_w16<k * (16 / r), T>(e[k], o[k], v[k], v[k + r / 2]);
            stage<k + 1>(v, e, o);
        }
    }
    static LRA_HD void run(cx<T>* v) {
        cx<T> e[r / 2], o[r / 2];
        LRA_UNROLL
        for (int k = 0; k < r / 2; ++k) { e[k] = v[2 * k]; o[k] = v[2 * k + 1]; }
        Dft<r / 2, T>::run(e);
        Dft<r / 2, T>::run(o);
        stage<0>(v, e, o);
    }
};
template <class T> struct Dft<1, T> {
    static LRA_HD void run(cx<T>*) {}
};
template <class T> struct Dft<2, T> {
    static LRA_HD void run(cx<T>* v) {
        const cx<T> a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};

// ----------------------------------------------------------------------------- Stockham passes
// Pass p: butterfly b in [0, M/r) takes inputs at b + j*(M/r), multiplies input j by
// W_{s r}^{(b mod s) j}, and writes output j to (b - b mod s) r + (b mod s) + j s, where s is the
// product of the earlier radices.  Thread tf owns butterflies b = tf + i*TF.
// LDS addressing.  phys(i) = i + (i >> PADSHIFT) is affine over any step that is a multiple of
// Q = 2^PADSHIFT: phys(x + c) = phys(x) + c + c/Q.  Every access pattern of the passes is
// "per-thread base + compile-time multiples of TF / stride", so when TF is a multiple of Q the
// whole pass needs ONE runtime address and immediate DS offsets; spelling it out that way keeps
// hipcc from materialising (and hoisting) one address VGPR per access.
template <class Cfg> constexpr int pstride(int c) { return c + (c >> Cfg::PADSHIFT); }
template <class Cfg> constexpr bool affine_tf() { return Cfg::TF % (1 << Cfg::PADSHIFT) == 0; }

template <class Cfg, int p> LRA_HD void pass_read(typename Cfg::cplx* v, Lds fr, int tf) {
    using C = typename Cfg::cplx;
    constexpr int lr = Cfg::logr(p), r = 1 << lr, nb = Cfg::R >> lr, sin = Cfg::M >> lr, Q = 1 << Cfg::PADSHIFT;
    if constexpr (affine_tf<Cfg>() && sin % Q == 0) {
        const int base = Cfg::phys(tf) * (int)sizeof(C);
        LRA_UNROLL
        for (int i = 0; i < nb; ++i) {
            LRA_UNROLL
            for (int j = 0; j < r; ++j) v[i * r + j] = lds_ld<C>(fr, base + (i * pstride<Cfg>(Cfg::TF) + j * pstride<Cfg>(sin)) * (int)sizeof(C));
        }
    } else {
        LRA_UNROLL
        for (int i = 0; i < nb; ++i) {
            const int b = tf + i * Cfg::TF;
            LRA_UNROLL
            for (int j = 0; j < r; ++j) v[i * r + j] = lds_ld<C>(fr, Cfg::phys(b + j * sin) * (int)sizeof(C));
        }
    }
}

template <class Cfg, int p> LRA_HD void pass_twiddle_dft(typename Cfg::cplx* v, int tf, const typename Cfg::cplx* __restrict__ tw) {
    using T = typename Cfg::real;
    constexpr int lr = Cfg::logr(p), r = 1 << lr, nb = Cfg::R >> lr, s = 1 << Cfg::logs(p);
    LRA_UNROLL
    for (int i = 0; i < nb; ++i) {
        const int b = tf + i * Cfg::TF;
        if (p > 0) {
            const int k = b & (s - 1);
            typename Cfg::cplx q[r];
            LRA_UNROLL
            for (int j = 1; j < r; ++j) q[j] = cmul_p(v[i * r + j], tw[Cfg::tw_off(p) + (j - 1) * s + k]);
            LRA_UNROLL
            for (int j = 1; j < r; ++j) v[i * r + j] = cmul_f(v[i * r + j], tw[Cfg::tw_off(p) + (j - 1) * s + k], q[j]);
        }
        Dft<r, T>::run(v + i * r);
    }
}

// Same, with this thread's twiddles already sitting in registers (see load_pass_twiddles).
constexpr int popcount_c(int j) { int c = 0; for (; j; j >>= 1) c += j & 1; return c; }
constexpr int msb_c(int j) { int m = 0; for (int t = 0; t < 16; ++t) if ((j >> t) & 1) m = t; return m; }

template <class Cfg, int p> LRA_HD void pass_twiddle_dft_reg(typename Cfg::cplx* v, const typename Cfg::cplx* treg) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int lr = Cfg::logr(p), r = 1 << lr, nb = Cfg::R >> lr;
    LRA_UNROLL
    for (int i = 0; i < nb; ++i) {
        if (p > 0) {
            // w[j] = W^(k j) from the stored powers of two: w^j = w^(j - msb j) * w^(msb j), level by level
            // (number of set bits of j), each level as a batch of independent products; then the r - 1
            // twiddle multiplies as another batch
            C w[r], q[r];
            LRA_UNROLL
            for (int j = 1; j < r; ++j)
                if (popcount_c(j) == 1) w[j] = treg[Cfg::treg_off(p) + i * lr + msb_c(j)];
            LRA_UNROLL
            for (int level = 2; level <= lr; ++level) {
                LRA_UNROLL
                for (int j = 1; j < r; ++j)
                    if (popcount_c(j) == level) q[j] = cmul_p(w[j - (1 << msb_c(j))], treg[Cfg::treg_off(p) + i * lr + msb_c(j)]);
                LRA_UNROLL
                for (int j = 1; j < r; ++j)
                    if (popcount_c(j) == level) w[j] = cmul_f(w[j - (1 << msb_c(j))], treg[Cfg::treg_off(p) + i * lr + msb_c(j)], q[j]);
            }
            LRA_UNROLL
            for (int j = 1; j < r; ++j) q[j] = cmul_p(v[i * r + j], w[j]);
            LRA_UNROLL
            for (int j = 1; j < r; ++j) v[i * r + j] = cmul_f(v[i * r + j], w[j], q[j]);
        }
        Dft<r, T>::run(v + i * r);
    }
}

// Copies this thread's base twiddles of pass p (they depend only on tf) from the global table to
// registers: W^(k 2^t), t = 0 .. log2(r) - 1.
template <class Cfg, int p> LRA_HD void load_pass_twiddles(typename Cfg::cplx* treg, int tf, const typename Cfg::cplx* __restrict__ tw) {
    constexpr int lr = Cfg::logr(p), nb = Cfg::R >> lr, s = 1 << Cfg::logs(p);
    LRA_UNROLL
    for (int i = 0; i < nb; ++i) {
        const int k = (tf + i * Cfg::TF) & (s - 1);
        LRA_UNROLL
        for (int t = 0; t < lr; ++t) treg[Cfg::treg_off(p) + i * lr + t] = tw[Cfg::tw_off(p) + ((1 << t) - 1) * s + k];
    }
}

template <class Cfg, int p> LRA_HD void pass_write(const typename Cfg::cplx* v, Lds fr, int tf) {
    using C = typename Cfg::cplx;
    constexpr int lr = Cfg::logr(p), r = 1 << lr, nb = Cfg::R >> lr, s = 1 << Cfg::logs(p);
    constexpr bool last = p == Cfg::P - 1;  // then b < s, so k = b and the output position is b + j s
    if constexpr (affine_tf<Cfg>() && p == 0) {
        // s = 1, k = 0: position b r + j with b r a multiple of Q = r: phys = b (r + 1) + j
        const int base = tf * (r + 1) * (int)sizeof(C);
        LRA_UNROLL
        for (int i = 0; i < nb; ++i) {
            LRA_UNROLL
            for (int j = 0; j < r; ++j) lds_st<C>(fr, base + (i * Cfg::TF * (r + 1) + j) * (int)sizeof(C), v[i * r + j]);
        }
    } else if constexpr (affine_tf<Cfg>() && p > 0 && (last || Cfg::TF % s == 0)) {
        int base;
        if (last) {
            base = Cfg::phys(tf) * (int)sizeof(C);  // position tf + i TF + j s
        } else {
            const int k = tf & (s - 1);  // TF % s == 0: k does not depend on i
            base = Cfg::phys(((tf - k) << lr) + k) * (int)sizeof(C);
        }
        constexpr int istep = last ? pstride<Cfg>(Cfg::TF) : pstride<Cfg>(Cfg::TF << lr);
        LRA_UNROLL
        for (int i = 0; i < nb; ++i) {
            LRA_UNROLL
            for (int j = 0; j < r; ++j) lds_st<C>(fr, base + (i * istep + j * pstride<Cfg>(s)) * (int)sizeof(C), v[i * r + j]);
        }
    } else {
        LRA_UNROLL
        for (int i = 0; i < nb; ++i) {
            const int b = tf + i * Cfg::TF;
            const int k = b & (s - 1);
            const int base = ((b - k) << lr) + k;
            LRA_UNROLL
            for (int j = 0; j < r; ++j) lds_st<C>(fr, Cfg::phys(base + j * s) * (int)sizeof(C), v[i * r + j]);
        }
    }
}

// Output position (natural order) of register slot (i, j) after the LAST pass.
template <class Cfg> LRA_HD int last_pass_pos(int tf, int i, int j) {
    constexpr int p = Cfg::P - 1, s = 1 << Cfg::logs(p);
    return (tf + i * Cfg::TF) + j * s;
}

// Host-side table builders (double precision, rounded once to T).
template <class Cfg> inline void build_pass_twiddles(typename Cfg::cplx* out) {
    const double two_pi = 6.283185307179586476925286766559;
    for (int p = 1; p < Cfg::P; ++p) {
        const int r = 1 << Cfg::logr(p), s = 1 << Cfg::logs(p);
        for (int j = 1; j < r; ++j)
            for (int k = 0; k < s; ++k) {
                const double ang = -two_pi * (double)k * (double)j / ((double)s * (double)r);
                out[Cfg::tw_off(p) + (j - 1) * s + k] = mk<typename Cfg::real>((typename Cfg::real)std::cos(ang), (typename Cfg::real)std::sin(ang));
            }
    }
}
// W_N^k = exp(-2 pi i k / N), k = 0 .. M  (SPLIT_TW_COUNT entries; the first-generation kernels read k <= M/2 only)
template <class Cfg> constexpr int split_tw_count() { return Cfg::M + 1; }
template <class Cfg> inline void build_split_twiddles(typename Cfg::cplx* out) {
    const double two_pi = 6.283185307179586476925286766559;
    for (int k = 0; k <= Cfg::M; ++k) {
        const double ang = -two_pi * (double)k / (double)Cfg::N;
        out[k] = mk<typename Cfg::real>((typename Cfg::real)std::cos(ang), (typename Cfg::real)std::sin(ang));
    }
}

}  // namespace lra
