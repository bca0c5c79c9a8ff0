// lra_kernels.h -- workgroup bodies of the fused power-of-two kernels:
//
//   stft_block<Cfg, MODE>   frame gather (+ np.pad index mapping at the clip edges) + window
//                           + real FFT, then one of three epilogues:
//       OUT_COMPLEX  D[b][t][k] complex          (librosa.stft,           core/spectrum.py:380-390)
//       OUT_POWER    S[b][t][k] = |X|^power      (_spectrogram,           core/spectrum.py:3000-3013)
//       OUT_MEL      M[b][m][t] = sum_k B[m,k] |X[k]|^power, the 1025-bin spectrum never leaves
//                    the CU (feature.melspectrogram, feature/spectral.py:2145-2161)
//   istft_block<Cfg>        Hermitian split + inverse FFT + window, overlap-add in LDS with a
//                           carry between frame groups (frame order = the reference's
//                           accumulation order, core/spectrum.py:593-603, 629-643), divide by the
//                           window sum-square where it exceeds tiny (:606-624).
//
// Device layouts (row-major, last index fastest):
//   PCM      y[b][n]                       n samples per clip, batch stride y_stride
//   spectrum D[b][t][k], k = 0..M          (the reference's F-ordered (..., 1+n_fft/2, T) array is
//                                           a transposed VIEW of this buffer, core/spectrum.py:356)
//   mel      M[b][m][t]                    C order, exactly the reference's einsum output
// Written once; compiled by hipcc for gfx950 and by g++ -DLRA_HOSTSIM for the CPU thread
// simulator used by tests (see lra_common.h).
#pragma once

#include <cmath>
#include <limits>

#include "lra_fft.h"

namespace lra {

enum OutMode : int { OUT_COMPLEX = 0, OUT_POWER = 1, OUT_MEL = 2 };
enum PowerMode : int { POW_ONE = 1, POW_TWO = 2, POW_GENERAL = 3 };

template <class T> struct StftArgs {
    // input
    const T* y;
    long long y_stride;  // elements between clips
    long long n;         // samples per clip
    int n_frames;        // T
    int hop;
    int pad;       // n_fft/2 when centred, else 0
    int pad_mode;  // PadMode
    // tables
    const T* win;       // [N]   window padded to n_fft
    const cx<T>* tw;    // pass twiddles (FftCfg::tw_off layout)
    const cx<T>* twr;   // split twiddles W_N^k, k = 0..M/2
    // work decomposition
    int frames_per_wg;  // multiple of FPB
    int wg_per_clip;
    // outputs (one of)
    cx<T>* D;  // [batch][T][M+1]
    T* S;      // [batch][T][M+1]
    T* Mel;    // [batch][n_mels][T]
    int power_mode;
    T power;
    // mel filterbank in band form: row m covers columns [c0, c0+len), values at mel_val[off + i]
    const int* mel_c0;
    const int* mel_len;
    const int* mel_off;
    const T* mel_val;
    int n_mels;
    // set to 1 when a frame's DC bin is not finite, i.e. (barring overflow) when some sample of the
    // frame is NaN/Inf: the device-side half of util.valid_audio (util/utils.py:305)
    unsigned int* nonfinite_flag;
};

template <class Cfg> struct FftRegs {
    typename Cfg::cplx v[Cfg::R];
    typename Cfg::cplx mid;
};

template <class T> LRA_HD T spec_power(cx<T> x, int power_mode, T power) {
    const T p2 = x.x * x.x + x.y * x.y;
    if (power_mode == POW_TWO) return p2;
    const T mag = std::sqrt(p2);
    if (power_mode == POW_ONE) return mag;
    return std::pow(mag, power);
}

// ---- phase (edge groups only): stage a frame that touches the np.pad region into LDS ------------
// The frame's N real samples (pad mode applied, no window yet) go to the start of its own LDS
// region; keeping the 64-bit index arithmetic of pad_index() out of the unrolled hot path.
template <class Cfg> LRA_HD void stft_stage_edge(const StftArgs<typename Cfg::real>& a, int clip, int frame, bool valid, int tf, Lds fr) {
    using T = typename Cfg::real;
    const T* __restrict__ yb = a.y + (long long)clip * a.y_stride;
    const long long base = (long long)frame * a.hop - a.pad;
    for (int q = tf; q < Cfg::N; q += Cfg::TF) {
        T x = (T)0;
        if (valid) {
            const long long i0 = pad_index(base + q, a.n, a.pad_mode);
            if (i0 >= 0) x = yb[i0];
        }
        lds_st<T>(fr, q * (int)sizeof(T), x);
    }
}

// frame [base, base+N) lies entirely inside the clip: the hot path may read it straight from HBM
template <class Cfg> LRA_HD bool frame_interior(const StftArgs<typename Cfg::real>& a, int frame) {
    const long long base = (long long)frame * a.hop - a.pad;
    return frame < a.n_frames && base >= 0 && base + Cfg::N <= a.n;
}

// ---- phase: gather + window (+ pass 0 butterflies when fused) ----------------------------------
// FROM_LDS: samples come from the staged copy (edge groups), else directly from global memory.
template <class Cfg, bool FROM_LDS> LRA_HD void stft_load(const StftArgs<typename Cfg::real>& a, int clip, int frame, int tf, typename Cfg::cplx* v, Lds fr) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int lr = Cfg::logr(0), r = 1 << lr, nb = Cfg::R >> lr, sin = Cfg::M >> lr;
    const C* __restrict__ win2 = reinterpret_cast<const C*>(a.win);
    if (FROM_LDS) {
        LRA_UNROLL
        for (int i = 0; i < nb; ++i) {
            LRA_UNROLL
            for (int j = 0; j < r; ++j) {
                const int q = tf + i * Cfg::TF + j * sin;  // complex index: samples 2q, 2q+1
                const C x = lds_ld<C>(fr, q * (int)sizeof(C));
                const C w = win2[q];
                v[i * r + j] = mk<T>(x.x * w.x, x.y * w.y);
            }
        }
        return;
    }
    const T* __restrict__ yf = a.y + (long long)clip * a.y_stride + ((long long)frame * a.hop - a.pad);
    if ((reinterpret_cast<uintptr_t>(yf) & (sizeof(C) - 1)) == 0) {
        const C* __restrict__ y2 = reinterpret_cast<const C*>(yf);
        LRA_UNROLL
        for (int i = 0; i < nb; ++i) {
            LRA_UNROLL
            for (int j = 0; j < r; ++j) {
                const int q = tf + i * Cfg::TF + j * sin;
                const C x = y2[q];
                const C w = win2[q];
                v[i * r + j] = mk<T>(x.x * w.x, x.y * w.y);
            }
        }
    } else {
        LRA_UNROLL
        for (int i = 0; i < nb; ++i) {
            LRA_UNROLL
            for (int j = 0; j < r; ++j) {
                const int q = tf + i * Cfg::TF + j * sin;
                const C w = win2[q];
                v[i * r + j] = mk<T>(yf[2 * q] * w.x, yf[2 * q + 1] * w.y);
            }
        }
    }
}

template <class Cfg> LRA_HD void stft_pass0(const StftArgs<typename Cfg::real>& a, int tf, typename Cfg::cplx* v, Lds fr) {
    pass_twiddle_dft<Cfg, 0>(v, tf, a.tw);
    pass_write<Cfg, 0>(v, fr, tf);
}

// ---- phase: read Z[k], Z[M-k] pairs for the split step ----------------------------------------
template <class Cfg> LRA_HD void split_read(FftRegs<Cfg>& rg, Lds fr, int tf) {
    using C = typename Cfg::cplx;
    LRA_UNROLL
    for (int i = 0; i < Cfg::R / 2; ++i) {
        const int k = tf + i * Cfg::TF;
        const int km = (Cfg::M - k) & (Cfg::M - 1);
        rg.v[2 * i] = lds_ld<C>(fr, Cfg::phys(k) * (int)sizeof(C));
        rg.v[2 * i + 1] = lds_ld<C>(fr, Cfg::phys(km) * (int)sizeof(C));
    }
    if (tf == 0) rg.mid = lds_ld<C>(fr, Cfg::phys(Cfg::M / 2) * (int)sizeof(C));
}

// X[k] and X[M-k] from Z[k], Z[M-k]:  A = (Zk + conj Zm)/2, B = W_N^k (Zk - conj Zm)/(2i),
// X[k] = A + B, X[M-k] = conj(A - B).
template <class T> LRA_HD void split_pair(cx<T> zk, cx<T> zm, cx<T> w, cx<T>& xk, cx<T>& xm) {
    const T h = (T)0.5;
    const cx<T> A = mk<T>((zk.x + zm.x) * h, (zk.y - zm.y) * h);
    const cx<T> O = mk<T>((zk.y + zm.y) * h, (zm.x - zk.x) * h);
    const cx<T> B = cmul(w, O);
    xk = cadd(A, B);
    xm = cconj(csub(A, B));
}

// ---- phase: split + epilogue store (complex / power) or power -> LDS (mel) --------------------
template <class Cfg, int MODE> LRA_HD void stft_split_store(const StftArgs<typename Cfg::real>& a, int clip, int frame, bool valid, int tf,
                                                            FftRegs<Cfg>& rg, Lds fr) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int M = Cfg::M;
    const long long row = ((long long)clip * a.n_frames + frame) * (M + 1);
    LRA_UNROLL
    for (int i = 0; i < Cfg::R / 2; ++i) {
        const int k = tf + i * Cfg::TF;
        C xk, xm;
        int km;
        if (k == 0) {
            const C z0 = rg.v[2 * i];
            xk = mk<T>(z0.x + z0.y, (T)0);
            xm = mk<T>(z0.x - z0.y, (T)0);
            km = M;
            if (valid && a.nonfinite_flag && !(std::fabs(xk.x) <= std::numeric_limits<T>::max())) LRA_ATOMIC_OR(a.nonfinite_flag, 1u);
        } else {
            split_pair<T>(rg.v[2 * i], rg.v[2 * i + 1], a.twr[k], xk, xm);
            km = M - k;
        }
        if (MODE == OUT_COMPLEX) {
            if (valid) { a.D[row + k] = xk; a.D[row + km] = xm; }
        } else {
            const T pk = spec_power<T>(xk, a.power_mode, a.power), pm = spec_power<T>(xm, a.power_mode, a.power);
            if (MODE == OUT_POWER) {
                if (valid) { a.S[row + k] = pk; a.S[row + km] = pm; }
            } else {
                lds_st<T>(fr, k * (int)sizeof(T), pk);
                lds_st<T>(fr, km * (int)sizeof(T), pm);
            }
        }
    }
    if (tf == 0) {
        const C xmid = cconj(rg.mid);  // X[M/2] = conj(Z[M/2])
        if (MODE == OUT_COMPLEX) {
            if (valid) a.D[row + M / 2] = xmid;
        } else {
            const T pmid = spec_power<T>(xmid, a.power_mode, a.power);
            if (MODE == OUT_POWER) {
                if (valid) a.S[row + M / 2] = pmid;
            } else {
                lds_st<T>(fr, (M / 2) * (int)sizeof(T), pmid);
            }
        }
    }
}

// ---- phase: sparse (banded) mel reduce over the power spectra held in LDS ----------------------
template <class Cfg> LRA_HD void mel_reduce_store(const StftArgs<typename Cfg::real>& a, int clip, int frame0, int tid, Lds lds) {
    using T = typename Cfg::real;
    const int total = a.n_mels * Cfg::FPB;
    for (int idx = tid; idx < total; idx += Cfg::NT) {
        const int slot = idx % Cfg::FPB, m = idx / Cfg::FPB;
        const int frame = frame0 + slot;
        if (frame >= a.n_frames) continue;
        const int c0 = a.mel_c0[m], len = a.mel_len[m];
        const T* __restrict__ w = a.mel_val + a.mel_off[m];
        const Lds fr = lds_sub(lds, slot * Cfg::FRAME_BYTES);
        T acc = (T)0;
        for (int i = 0; i < len; ++i) acc += w[i] * lds_ld<T>(fr, (c0 + i) * (int)sizeof(T));
        a.Mel[((long long)clip * a.n_mels + m) * a.n_frames + frame] = acc;
    }
}

#define LRA_MID_PASS(Cfg, p, rg, lds, tw)                                                                 \
    if (Cfg::P > p) {                                                                                     \
        LRA_PHASE(Cfg::NT, tid) {                                                                         \
            pass_read<Cfg, (p < Cfg::P ? p : 0)>(LRA_R(rg).v, lds_sub(lds, (tid / Cfg::TF) * Cfg::FRAME_BYTES), tid % Cfg::TF); \
        } LRA_PHASE_END                                                                                   \
        LRA_PHASE(Cfg::NT, tid) {                                                                         \
            pass_twiddle_dft<Cfg, (p < Cfg::P ? p : 0)>(LRA_R(rg).v, tid % Cfg::TF, tw);                \
            pass_write<Cfg, (p < Cfg::P ? p : 0)>(LRA_R(rg).v, lds_sub(lds, (tid / Cfg::TF) * Cfg::FRAME_BYTES), tid % Cfg::TF); \
        } LRA_PHASE_END                                                                                   \
    }

// One workgroup: clip = blk / wg_per_clip, frames [wg*frames_per_wg, +frames_per_wg).
template <class Cfg, int MODE> LRA_HD void stft_block(const StftArgs<typename Cfg::real>& a_in, const int blk, Lds lds) {
    StftArgs<typename Cfg::real> a = a_in;
    const int clip = blk / a.wg_per_clip;
    const int f_first = (blk % a.wg_per_clip) * a.frames_per_wg;
    const int iters = a.frames_per_wg / Cfg::FPB;
    LRA_REGS(FftRegs<Cfg>, rg, Cfg::NT);
    for (int it = 0; it < iters; ++it) {
        if (!Cfg::HOIST) { LRA_LAUNDER(a.win); LRA_LAUNDER(a.tw); LRA_LAUNDER(a.twr); }
        const int frame0 = f_first + it * Cfg::FPB;
        if (frame0 >= a.n_frames) break;  // uniform across the workgroup
        // a group is "interior" when all of its FPB frames exist and lie inside the clip
        const bool group_interior = frame_interior<Cfg>(a, frame0) && frame_interior<Cfg>(a, frame0 + Cfg::FPB - 1);
        if (group_interior) {
            LRA_PHASE(Cfg::NT, tid) {
                const int slot = tid / Cfg::TF, tf = tid % Cfg::TF;
                const Lds fr = lds_sub(lds, slot * Cfg::FRAME_BYTES);
                stft_load<Cfg, false>(a, clip, frame0 + slot, tf, LRA_R(rg).v, fr);
                stft_pass0<Cfg>(a, tf, LRA_R(rg).v, fr);
            } LRA_PHASE_END
        } else {
            LRA_PHASE(Cfg::NT, tid) {
                const int slot = tid / Cfg::TF, tf = tid % Cfg::TF, frame = frame0 + slot;
                stft_stage_edge<Cfg>(a, clip, frame, frame < a.n_frames, tf, lds_sub(lds, slot * Cfg::FRAME_BYTES));
            } LRA_PHASE_END
            LRA_PHASE(Cfg::NT, tid) {
                const int slot = tid / Cfg::TF, tf = tid % Cfg::TF;
                stft_load<Cfg, true>(a, clip, frame0 + slot, tf, LRA_R(rg).v, lds_sub(lds, slot * Cfg::FRAME_BYTES));
            } LRA_PHASE_END
            LRA_PHASE(Cfg::NT, tid) {
                const int slot = tid / Cfg::TF, tf = tid % Cfg::TF;
                stft_pass0<Cfg>(a, tf, LRA_R(rg).v, lds_sub(lds, slot * Cfg::FRAME_BYTES));
            } LRA_PHASE_END
        }
        LRA_MID_PASS(Cfg, 1, rg, lds, a.tw)
        LRA_MID_PASS(Cfg, 2, rg, lds, a.tw)
        LRA_MID_PASS(Cfg, 3, rg, lds, a.tw)
        LRA_PHASE(Cfg::NT, tid) {
            split_read<Cfg>(LRA_R(rg), lds_sub(lds, (tid / Cfg::TF) * Cfg::FRAME_BYTES), tid % Cfg::TF);
        } LRA_PHASE_END
        LRA_PHASE(Cfg::NT, tid) {
            const int slot = tid / Cfg::TF, tf = tid % Cfg::TF, frame = frame0 + slot;
            stft_split_store<Cfg, MODE>(a, clip, frame, frame < a.n_frames, tf, LRA_R(rg), lds_sub(lds, slot * Cfg::FRAME_BYTES));
        } LRA_PHASE_END
        if (MODE == OUT_MEL) {
            LRA_PHASE(Cfg::NT, tid) { mel_reduce_store<Cfg>(a, clip, frame0, tid, lds); } LRA_PHASE_END
        }
    }
}

template <class Cfg> constexpr int stft_lds_bytes() { return Cfg::FPB * Cfg::FRAME_BYTES; }

// =================================================================================================
// ISTFT
// =================================================================================================
template <class T> struct IstftArgs {
    const cx<T>* D;             // [batch][*][M+1]
    long long d_batch_stride;   // elements between clips
    long long d_frame_stride;   // elements between frames (>= M+1)
    int n_used;                 // frames [0, n_used) contribute
    int hop;
    int drop;                   // n_fft/2 when centred, else 0
    const T* win_scaled;        // [N] window / N
    const cx<T>* tw;
    const cx<T>* twr;
    const T* wss;               // [out_len] window sum-square already sliced/fixed to the output
    T tiny;
    T* y;                       // [batch][out_len], pre-zeroed by the host wrapper
    long long y_stride;
    long long out_len;
    int strip_frames;           // frames finalised per workgroup, multiple of FPB
    int strips_per_clip;
    int warm_groups;            // groups of FPB frames replayed before the strip
    int drain_groups;           // extra groups the last strip runs to flush the carry
};

template <class Cfg> constexpr int istft_carry_elems() { return Cfg::N; }  // >= N - hop for any hop >= 1
template <class Cfg> constexpr int istft_lds_bytes() { return Cfg::FPB * Cfg::FRAME_BYTES + 2 * istft_carry_elems<Cfg>() * (int)sizeof(typename Cfg::real); }

// ---- phase: Hermitian split of X[0..M] into conj(Z'[0..M-1]) in LDS ---------------------------
// Z'[k] = E' + i O',  E' = X[k] + conj(X[M-k]),  O' = (X[k] - conj(X[M-k])) conj(W_N^k); the
// imaginary parts of X[0] and X[M] are ignored, as pocketfft's c2r does.
template <class Cfg> LRA_HD void istft_split_write(const IstftArgs<typename Cfg::real>& a, int clip, int frame, bool valid, int tf, Lds fr) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int M = Cfg::M;
    const C* __restrict__ X = a.D + (long long)clip * a.d_batch_stride + (long long)frame * a.d_frame_stride;
    const C zero = mk<T>((T)0, (T)0);
    LRA_UNROLL
    for (int i = 0; i < Cfg::R / 2; ++i) {
        const int k = tf + i * Cfg::TF;
        if (k == 0) {
            const C x0 = valid ? X[0] : zero, xM = valid ? X[M] : zero;
            lds_st<C>(fr, Cfg::phys(0) * (int)sizeof(C), mk<T>(x0.x + xM.x, -(x0.x - xM.x)));
        } else {
            const C xk = valid ? X[k] : zero, xm = valid ? X[M - k] : zero;
            const C E = mk<T>(xk.x + xm.x, xk.y - xm.y);
            const C Dif = mk<T>(xk.x - xm.x, xk.y + xm.y);
            const C O = cmul(Dif, cconj(a.twr[k]));
            // Z'[k] = (E.x - O.y, E.y + O.x);  Z'[M-k] = (E.x + O.y, O.x - E.y); store conjugates
            lds_st<C>(fr, Cfg::phys(k) * (int)sizeof(C), mk<T>(E.x - O.y, -(E.y + O.x)));
            lds_st<C>(fr, Cfg::phys(M - k) * (int)sizeof(C), mk<T>(E.x + O.y, -(O.x - E.y)));
        }
    }
    if (tf == 0) {
        const C xh = valid ? X[M / 2] : zero;  // Z'[M/2] = 2 conj(X[M/2]); store its conjugate
        lds_st<C>(fr, Cfg::phys(M / 2) * (int)sizeof(C), mk<T>((T)2 * xh.x, (T)2 * xh.y));
    }
}

// ---- phase: last pass butterflies, then windowed time-domain frame -> LDS (natural order) ------
template <class Cfg> LRA_HD void istft_last_write(const IstftArgs<typename Cfg::real>& a, typename Cfg::cplx* v, int tf, Lds fr) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int p = Cfg::P - 1, lr = Cfg::logr(p), r = 1 << lr, nb = Cfg::R >> lr;
    pass_twiddle_dft<Cfg, p>(v, tf, a.tw);
    LRA_UNROLL
    for (int i = 0; i < nb; ++i) {
        LRA_UNROLL
        for (int j = 0; j < r; ++j) {
            const int q = last_pass_pos<Cfg>(tf, i, j);
            const C z = v[i * r + j];  // = conj(z'[q]); x[2q] = Re z', x[2q+1] = Im z'
            lds_st<C>(fr, q * (int)sizeof(C), mk<T>(z.x * a.win_scaled[2 * q], -z.y * a.win_scaled[2 * q + 1]));
        }
    }
}

// ---- phase: overlap-add of the FPB frames of this group + carry, finalise FPB*hop samples ------
template <class Cfg> LRA_HD void istft_ola(const IstftArgs<typename Cfg::real>& a, int clip, int group_first_frame, long long write_lo,
                                           long long write_hi, int parity, int tid, Lds lds) {
    using T = typename Cfg::real;
    constexpr int N = Cfg::N, FPB = Cfg::FPB;
    const int H = a.hop;
    const int CL = N > H ? N - H : 0;                    // carry length
    const long long L = (long long)(FPB - 1) * H + N;    // positions touched by this group
    const long long pa = (long long)group_first_frame * H;  // padded position of u = 0
    const long long fin = (long long)FPB * H;            // positions finalised by this group
    const Lds carry_in = lds_sub(lds, FPB * Cfg::FRAME_BYTES + parity * istft_carry_elems<Cfg>() * (int)sizeof(T));
    const Lds carry_out = lds_sub(lds, FPB * Cfg::FRAME_BYTES + (1 - parity) * istft_carry_elems<Cfg>() * (int)sizeof(T));
    const long long span = L > fin ? L : fin;
    for (long long u = tid; u < span; u += Cfg::NT) {
        T val = (T)0;
        if (u < CL) val = lds_ld<T>(carry_in, (int)u * (int)sizeof(T));
        LRA_UNROLL
        for (int slot = 0; slot < FPB; ++slot) {
            const int t = group_first_frame + slot;
            const long long off = u - (long long)slot * H;
            if (t >= 0 && t < a.n_used && off >= 0 && off < N) val += lds_ld<T>(lds_sub(lds, slot * Cfg::FRAME_BYTES), (int)off * (int)sizeof(T));
        }
        if (u < fin) {
            const long long sp = pa + u;  // padded position
            const long long s = sp - a.drop;
            if (sp >= write_lo && sp < write_hi && s >= 0 && s < a.out_len) {
                const T w = a.wss[s];
                a.y[(long long)clip * a.y_stride + s] = (w > a.tiny) ? val / w : val;
            }
        } else if (u - fin < CL) {
            lds_st<T>(carry_out, (int)(u - fin) * (int)sizeof(T), val);
        }
    }
}

template <class Cfg> LRA_HD void istft_block(const IstftArgs<typename Cfg::real>& a_in, const int blk, Lds lds) {
    IstftArgs<typename Cfg::real> a = a_in;
    using T = typename Cfg::real;
    constexpr int FPB = Cfg::FPB;
    const int clip = blk / a.strips_per_clip;
    const int strip = blk % a.strips_per_clip;
    const bool last = strip == a.strips_per_clip - 1;
    const int t0 = strip * a.strip_frames;
    int t1 = t0 + a.strip_frames;
    if (t1 > a.n_used) t1 = a.n_used;
    const int ta = t0 - a.warm_groups * FPB;
    int groups = a.warm_groups + (t1 - t0 + FPB - 1) / FPB;
    if (last) groups += a.drain_groups;
    const long long write_lo = (long long)t0 * a.hop;
    const long long write_hi = last ? (long long)0x7fffffffffffffffLL : (long long)t1 * a.hop;
    LRA_REGS(FftRegs<Cfg>, rg, Cfg::NT);
    // the first group's carry-in is all zeros
    LRA_PHASE(Cfg::NT, tid) {
        const Lds c0 = lds_sub(lds, FPB * Cfg::FRAME_BYTES);
        for (int u = tid; u < istft_carry_elems<Cfg>(); u += Cfg::NT) lds_st<T>(c0, u * (int)sizeof(T), (T)0);
    } LRA_PHASE_END
    for (int g = 0; g < groups; ++g) {
        if (!Cfg::HOIST) { LRA_LAUNDER(a.win_scaled); LRA_LAUNDER(a.tw); LRA_LAUNDER(a.twr); }
        const int gf = ta + g * FPB;
        LRA_PHASE(Cfg::NT, tid) {
            const int slot = tid / Cfg::TF, tf = tid % Cfg::TF, frame = gf + slot;
            istft_split_write<Cfg>(a, clip, frame, frame >= 0 && frame < a.n_used, tf, lds_sub(lds, slot * Cfg::FRAME_BYTES));
        } LRA_PHASE_END
        // pass 0 (no twiddles) reads from LDS here, unlike the forward kernel
        if (Cfg::P > 1) {
            LRA_PHASE(Cfg::NT, tid) {
                pass_read<Cfg, 0>(LRA_R(rg).v, lds_sub(lds, (tid / Cfg::TF) * Cfg::FRAME_BYTES), tid % Cfg::TF);
            } LRA_PHASE_END
            LRA_PHASE(Cfg::NT, tid) {
                pass_twiddle_dft<Cfg, 0>(LRA_R(rg).v, tid % Cfg::TF, a.tw);
                pass_write<Cfg, 0>(LRA_R(rg).v, lds_sub(lds, (tid / Cfg::TF) * Cfg::FRAME_BYTES), tid % Cfg::TF);
            } LRA_PHASE_END
        }
        if (Cfg::P > 2) { LRA_MID_PASS(Cfg, 1, rg, lds, a.tw) }
        if (Cfg::P > 3) { LRA_MID_PASS(Cfg, 2, rg, lds, a.tw) }
        LRA_PHASE(Cfg::NT, tid) {
            pass_read<Cfg, Cfg::P - 1>(LRA_R(rg).v, lds_sub(lds, (tid / Cfg::TF) * Cfg::FRAME_BYTES), tid % Cfg::TF);
        } LRA_PHASE_END
        LRA_PHASE(Cfg::NT, tid) {
            istft_last_write<Cfg>(a, LRA_R(rg).v, tid % Cfg::TF, lds_sub(lds, (tid / Cfg::TF) * Cfg::FRAME_BYTES));
        } LRA_PHASE_END
        LRA_PHASE(Cfg::NT, tid) { istft_ola<Cfg>(a, clip, gf, write_lo, write_hi, g & 1, tid, lds); } LRA_PHASE_END
    }
}

}  // namespace lra
