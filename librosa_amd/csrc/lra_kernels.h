// lra_kernels.h -- workgroup bodies of the fused power-of-two kernels:
//
//   stft_block<Cfg, MODE>   frame gather (+ np.pad index mapping at the clip edges) + window
//                           + real FFT, then one of three epilogues:
//       OUT_COMPLEX  D[b][t][k] complex          (librosa.stft,           core/spectrum.py:380-390)
//       OUT_POWER    S[b][t][k] = |X|^power      (_spectrogram,           core/spectrum.py:3000-3013)
//       OUT_MEL*     M[b][m][t] = sum_k B[m,k] |X[k]|^power, the 1025-bin spectrum never leaves
//                    the CU (feature.melspectrogram, feature/spectral.py:2145-2161): OUT_MELR / OUT_MEL2 for
//                    triangular banks (run-ordered / masked two-slope forms, lra_mel.h), OUT_MEL for any banded basis
//   istft_block<Cfg, HC>    Hermitian split + inverse FFT + window, overlap-add in LDS with a
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

#ifndef LRA_ABLATE
#define LRA_ABLATE 0  // kernel experiments (scripts/gpu_probe.py builds), 0 in the product
#endif

#include <cmath>
#include <limits>

#include "lra_fft.h"

// Phase timer (experiments only, -DLRA_PHASE_TIMER in a single-TU probe build): every wave adds the shader-clock
// ticks it spends between consecutive LRA_TICK points of the forward kernel's frame loop to lra_phase_ticks[i].
// s_memtime needs its own lgkmcnt(0) wait, which also drains the wave's LDS queue at that point: the numbers show
// where the time goes, at ~10 % overhead.
#if defined(LRA_PHASE_TIMER) && !defined(LRA_HOSTSIM)
__device__ unsigned long long lra_phase_ticks[16];
#define LRA_TICK(i)                                                    \
    do {                                                               \
        const unsigned long long now__ = __builtin_amdgcn_s_memtime(); \
        tick_acc[i] += now__ - tick_prev;                              \
        tick_prev = now__;                                             \
    } while (0)
#define LRA_TICK_DECL unsigned long long tick_acc[16] = {}, tick_prev = __builtin_amdgcn_s_memtime()
#define LRA_TICK_FLUSH                                                                    \
    do {                                                                                  \
        if ((threadIdx.x & 63) == 0)                                                      \
            for (int i__ = 0; i__ < 16; ++i__) atomicAdd(&lra_phase_ticks[i__], tick_acc[i__]); \
    } while (0)
#else
#define LRA_TICK(i) ((void)0)
#define LRA_TICK_DECL ((void)0)
#define LRA_TICK_FLUSH ((void)0)
#endif

namespace lra {

enum OutMode : int { OUT_COMPLEX = 0, OUT_POWER = 1, OUT_MEL = 2, OUT_MEL2 = 3, OUT_MELR = 4 };  // MEL2: two-slope filterbank, MELR: its run-ordered form (lra_mel.h)
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
    const T* win;       // [N]   0.5 * window padded to n_fft (the 1/2 of the real-FFT split step, folded in)
    const cx<T>* tw;    // pass twiddles (FftCfg::tw_off layout)
    const cx<T>* twr;   // split twiddles W_N^k, k = 0..M/2
    // work decomposition
    int frames_per_wg;  // FPB * iters: slot s owns frames [s*iters, (s+1)*iters) of the workgroup's range
    int wg_per_clip;
    int slot_bytes;     // LDS bytes per frame slot (frame area + PCM ring + mel staging tile)
    int mel_tile;       // frames staged per mel row before a flush
    // outputs (one of)
    cx<T>* D;  // [batch][T][row_pitch], M + 1 bins of each row used
    T* S;      // [batch][T][row_pitch]
    long long row_pitch;  // ELEMENTS between the rows of consecutive frames: M + 1 = packed (the layout core/spectrum.py:356 allocates); larger = rows padded so that
                          // each starts on a cache-line boundary, the caller presenting the result as a strided view (lra_stft_exec_strided)
    T* Mel;    // [batch][n_mels][T]
    int power_mode;
    T power;
    // mel filterbank in band form: row m covers columns [c0, c0+len), values at mel_val[off + i]
    const int* mel_c0;
    const int* mel_len;
    const int* mel_off;
    const T* mel_val;
    int n_mels;
    // two-slope form (OUT_MEL2, lra_mel.h): interleaved per-bin weights (wA, wB) and the packed pair
    // segments, copied to a workgroup-shared LDS region at byte offset shared_off once per workgroup
    const T* mel_wAB;
    const int* mel_run;   // [TF] run descriptors (lra_mel.h)
    const int* mel_segd;  // [n_mels + 1] segment -> pieces
    int mel_nyq;          // piece id of the Nyquist bin
    // run-ordered two-slope form (OUT_MELR, lra_mel.h MelRuns): also copied to the shared LDS region
    const T* melr_w;        // [(M + 1)] (wA, wB) pairs: bins 0..M/2-1, then M, M-1, .., M/2+1, then M/2
    const T* melr_keep;     // [R][TF]: 0 where a thread's running sum restarts
    const int* melr_addr;   // [2 pmax][n_mels] LDS byte addresses of the piece totals (entries 0..pmax-1: B list, then the A list)
    int rot_uniform;          // row-aligned ring, several slots per wave: frames per slot is a multiple of n_fft/hop, so all slots of a wave share the ring rotation
    int melr_pmax;            // list length = pieces of the widest pair segment, at least MELR_PHOIST (<= MELR_PMAX)
    int melr_zero, melr_mid;  // byte addresses (inside the slot's running-sum area) of the zero slot and of bin M/2's slot
    int shared_off;
    // launch geometry: the grid is n_blocks rounded up to 8 * xcd_chunk workgroups (xcd_chunk = 0: no remap), see xcd_block
    int n_blocks, xcd_chunk;
    // set to 1 when a frame's DC bin is not finite, i.e. (barring overflow) when some sample of the
    // frame is NaN/Inf: the device-side half of util.valid_audio (util/utils.py:305)
    unsigned int* nonfinite_flag;
};

// Workgroup -> work item, XCD-aware.  The hardware deals workgroups round-robin to the 8 XCDs (workgroup b runs on XCD
// b mod 8, MI355X_MICROARCH.md), so with the identity map the consecutive strips of a clip -- neighbours in the output
// and sharing their PCM halo -- land on 8 different L2s.  Mapping b -> (b mod 8) * chunk + b / 8 gives every XCD one
// contiguous eighth of the (clip, strip) list instead: measured on the store stream of the STFT (scripts/storepat2.hip)
// +6 % on its own and +15 % together with aligned non-temporal stores.  chunk = ceil(n_blocks / 8); 0 = identity.
LRA_HD int xcd_block(int b, int chunk) { return chunk > 0 ? (b & 7) * chunk + (b >> 3) : b; }

// Thread -> (frame slot, thread within the slot).  The slot index is wave-uniform whenever a slot is one or more
// whole waves (TF >= 64): fetching it through readfirstlane keeps everything derived from it -- frame number,
// row bases of the stores, prefetch pointers and their 64-bit bound checks -- in SGPRs and on the scalar unit.
template <class Cfg> LRA_HD int slot_of(int tid) {
    if (Cfg::FPB == 1) return 0;
    if (Cfg::TF >= 64) return LRA_UNIFORM((int)((unsigned)tid / (unsigned)Cfg::TF));
    return (int)((unsigned)tid / (unsigned)Cfg::TF);
}
template <class Cfg> LRA_HD int lane_of(int tid) { return (int)((unsigned)tid % (unsigned)Cfg::TF); }

// OUT_MELR register budget per thread: NB mel bands (NB tf .. NB tf + NB - 1), the first PH entries of each band's two piece lists, and the last
// TILE frames' values of each band.  One wave per frame (TF >= 64) serves 128 bands with two per thread; small frames share a wave (TF = 32 at
// n_fft 1024, 16 at 512) and their threads take 4 / 8 bands each -- round 5: with two, 96 of 128 bands of an n_fft = 512 frame went through the
// table path (two dependent LDS reads per piece, one 4-byte store per band and frame): 1.84 ms against 0.31 ms for the transform itself.  Their
// pair segments are a few bins wide, so two hoisted pieces per list cover them (longer lists take the rest from the table).
#ifndef LRA_MELR_NB_SMALL
#define LRA_MELR_NB_SMALL 4  // bands per thread where TF <= 16 (8 was measured: the n_fft = 512 kernel then needs 276 VGPRs, and 20 spilled registers cost more than the table path)
#endif
#ifndef LRA_MELR_PH_SMALL
#define LRA_MELR_PH_SMALL 3  // hoisted pieces per list there (same box, n_fft 512 / hop 128 / 80 bands: 2 -> 0.85, 3 -> 0.75, 4 -> 0.77 ms; two bands x four pieces before: 0.95)
#endif
// (TF = 32, n_fft = 1024: four bands per thread with two hoisted pieces measured 0.79 against 0.83 ms at 128 bands but 0.78 against 0.68 ms at 80, whose
// wider segments need the longer hoisted lists: it keeps two bands per thread)
// `many`: the 128-thread workgroup shape of lra_dispatch.h (MelManyCfgOf): eight bands, one hoisted piece per list
constexpr int melr_nb_of_tf(int tf, bool many = false) { return tf >= 32 ? 2 : (many ? 8 : LRA_MELR_NB_SMALL); }
constexpr int melr_ph_of_tf(int tf, bool many = false) { return tf >= 32 ? 4 : (many ? 1 : LRA_MELR_PH_SMALL); }  // (also the minimum list length the host builds: lra_api.hip)
template <class Cfg> constexpr bool melr_many() { return Cfg::TF == 16 && Cfg::NT == 128; }
template <class Cfg> constexpr int melr_nb() { return melr_nb_of_tf(Cfg::TF, melr_many<Cfg>()); }
template <class Cfg> constexpr int melr_ph() { return melr_ph_of_tf(Cfg::TF, melr_many<Cfg>()); }
template <class Cfg> constexpr int melr_tile_frames() { return melr_nb<Cfg>() == 2 ? 8 : 4; }  // (NB x TILE = 16 registers, or 32 at eight bands per thread)

template <class Cfg> struct FftRegs {
    typename Cfg::cplx v[Cfg::R];
    typename Cfg::cplx mid;
    // next frame's new hop samples, in flight while the current frame is transformed (hop <= n_fft/4 ...
    static constexpr int NPF = (Cfg::N / 4) / Cfg::TF > 0 ? (Cfg::N / 4) / Cfg::TF : 1;
    // ... up to hop = n_fft/2 in the kernels with general ring addressing (the row-aligned ones use the first NPF only)
    static constexpr int NPFX = 2 * NPF;
    typename Cfg::real pf[NPFX];
    typename Cfg::cplx nxt[Cfg::R];  // direct framing (hop >= n_fft, no ring): the next frame's sample pairs in pass-0 register order, in flight during the current frame
    typename Cfg::cplx raw[Cfg::R];  // register ring (hop = n_fft / HD, RAM = 3..6): this frame's sample pairs in pass-0 order; nxt[0 .. R/HD) then holds the next frame's new pairs
    // HOIST configurations: this thread's table values, loaded once before the frame loop.  (hipcc
    // does not hoist them by itself across the per-phase fences; re-reading ~52 table values per
    // frame from L2 left the waves 65 % of their time in s_waitcnt.)
    // ISTFT: next frame's spectrum bins (X[k], X[M-k] pairs and X[M/2]) in flight during the current
    // frame, and the current hop's finished output samples held back until that prefetch has landed
    typename Cfg::cplx xk[Cfg::R / 2 > 0 ? Cfg::R / 2 : 1], xm[Cfg::R / 2 > 0 ? Cfg::R / 2 : 1], xmid;
    typename Cfg::real out[NPFX];  // ISTFT hold-back (hop <= n_fft/2 in the row-aligned kernels, <= n_fft/4 otherwise)
    // OUT_MELR: this thread's restart factors and the first MELR_PHOIST entries of the piece lists of its (up to two) mel bands
    static constexpr int MELR_PHOIST = melr_ph<Cfg>(), MELR_NB = melr_nb<Cfg>();
    typename Cfg::real keep[Cfg::R];
    int mad[MELR_NB][2 * MELR_PHOIST];
    // ... and the last MELR_TILE frames' values of those two bands: stored as one burst per band every MELR_TILE
    // frames, so that the L2 merges them into whole 32-byte sectors (single 4-byte stores 5 us apart do not merge:
    // 865 MB of HBM writes per launch for 169 MB of output)
    static constexpr int MELR_TILE = melr_tile_frames<Cfg>();
    typename Cfg::real mt[MELR_NB][MELR_TILE];
    typename Cfg::real wv[NPFX];  // ISTFT (row-aligned): window sum-square values of the held-back samples, loaded a frame ahead
    typename Cfg::cplx cry[Cfg::R];  // ISTFT (row-aligned, istft_reg_carry): the overlap-add carry, pair c of this thread's R - HC rows that outlive a frame
    static constexpr int NH = Cfg::HOIST ? 1 : 0;
    typename Cfg::cplx win2[NH * Cfg::R + 1 - NH];       // window pairs in pass-0 register order
    typename Cfg::cplx treg[NH * Cfg::TREG_TOTAL + 1 - NH];
    typename Cfg::cplx twr[NH * (Cfg::R / 2) + 1 - NH];  // split twiddles W_N^k, k = tf + i*TF
};

// prologue of HOIST kernels: window, pass twiddles and split twiddles -> registers
template <class Cfg> LRA_HD void hoist_tables(FftRegs<Cfg>& rg, int tf, const typename Cfg::real* __restrict__ win, const typename Cfg::cplx* __restrict__ tw,
                                              const typename Cfg::cplx* __restrict__ twr, bool window_in_last_pass_order, bool split_run_order = false) {
    using C = typename Cfg::cplx;
    if (!Cfg::HOIST) return;
    const C* __restrict__ win2 = reinterpret_cast<const C*>(win);
    if (window_in_last_pass_order) {
        constexpr int p = Cfg::P - 1, lr = Cfg::logr(p), r = 1 << lr, nb = Cfg::R >> lr;
        LRA_UNROLL
        for (int i = 0; i < nb; ++i) {
            LRA_UNROLL
            for (int j = 0; j < r; ++j) rg.win2[i * r + j] = win2[last_pass_pos<Cfg>(tf, i, j)];
        }
    } else {
        constexpr int lr = Cfg::logr(0), r = 1 << lr, nb = Cfg::R >> lr, sin = Cfg::M >> lr;
        LRA_UNROLL
        for (int i = 0; i < nb; ++i) {
            LRA_UNROLL
            for (int j = 0; j < r; ++j) rg.win2[i * r + j] = win2[tf + i * Cfg::TF + j * sin];
        }
    }
    if (Cfg::P > 1) load_pass_twiddles<Cfg, (1 < Cfg::P ? 1 : 0)>(rg.treg, tf, tw);
    if (Cfg::P > 2) load_pass_twiddles<Cfg, (2 < Cfg::P ? 2 : 0)>(rg.treg, tf, tw);
    if (Cfg::P > 3) load_pass_twiddles<Cfg, (3 < Cfg::P ? 3 : 0)>(rg.treg, tf, tw);
    LRA_UNROLL
    for (int i = 0; i < Cfg::R / 2; ++i) rg.twr[i] = split_run_order ? twr[(Cfg::R / 2) * tf + i] : twr[tf + i * Cfg::TF];
}

// twiddle + butterflies of pass p from whichever source the configuration uses
template <class Cfg, int p> LRA_HD void pass_dft(FftRegs<Cfg>& rg, int tf, const typename Cfg::cplx* __restrict__ tw) {
    if (Cfg::HOIST) pass_twiddle_dft_reg<Cfg, p>(rg.v, rg.treg);
    else pass_twiddle_dft<Cfg, p>(rg.v, tf, tw);
}

// The power mode is a compile-time parameter of the kernels (PM in {POW_TWO, POW_ONE, POW_GENERAL}): with a
// run-time switch hipcc if-converts the uniform branch and executes the whole software pow() for every
// bin even when power == 2 (it cost the fused mel kernel 25 % of its time).
template <class T, int PM> LRA_HD T spec_power(cx<T> x, T power) {
    const T p2 = x.x * x.x + x.y * x.y;
    if (PM == POW_TWO) return p2;
    const T mag = std::sqrt(p2);
    if (PM == POW_ONE) return mag;
    return std::pow(mag, power);
}

// slot of bin k in the (wA P, wB P) array of the mel epilogue: one pad slot per thread run of R bins
template <class Cfg> LRA_HD int ab_slot(int k) { return k + (k >> Cfg::LOGR); }

// ---- PCM ring -------------------------------------------------------------------------------------
// Each frame slot keeps the N samples of its current frame in an LDS ring indexed by the padded
// sample position modulo N.  A slot walks CONSECUTIVE frames, so going from frame t to t+1 only the
// last min(hop, N) samples are new: every PCM sample is fetched from HBM once per slot (instead of
// n_fft/hop = 4 times), and because the new samples are few (n_fft/4 per frame = R/2 registers per
// thread) they are prefetched into registers one whole frame ahead, hiding the HBM latency behind
// the FFT of the current frame.  np.pad semantics (core/spectrum.py:287) are applied at fetch time.
template <class T> LRA_HD T fetch_sample(const T* __restrict__ yb, long long p, int pad, long long n, int pad_mode) {
    const long long g = p - pad;
    if (g >= 0 && g < n) return yb[g];
    const long long idx = pad_index(g, n, pad_mode);
    return idx >= 0 ? yb[idx] : (T)0;
}

// Ring maintenance is split in three so that the wave never has to wait for its own spectrum stores
// (vmcnt completes in order: a wait for a prefetch load also waits for every older store):
//   ring_fill      once per slot: all N samples of its first frame;
//   ring_prefetch  at the START of frame t: issue the loads of frame t+1's new samples (registers);
//   ring_advance   at the END of frame t, BEFORE frame t's stores are issued: wait for those loads
//                  and write them into the ring (frame t's ring reads are long done).  The only
//                  older stores still in flight at that wait are frame t-1's, a whole FFT old.
template <class Cfg> LRA_HD void stft_ring_fill(const StftArgs<typename Cfg::real>& a, int clip, int frame, int tf, Lds ring) {
    using T = typename Cfg::real;
    constexpr int N = Cfg::N;
    if (frame >= a.n_frames) return;
    const T* __restrict__ yb = a.y + (long long)clip * a.y_stride;
    const long long p0 = (long long)frame * a.hop;  // padded position of the frame's first sample
    for (int i = tf; i < N; i += Cfg::TF) lds_st<T>(ring, (int)((p0 + i) & (N - 1)) * (int)sizeof(T), fetch_sample<T>(yb, p0 + i, a.pad, a.n, a.pad_mode));
}

// new samples of frame `next` = frame + 1: padded positions [next*H + N - Hn, next*H + N).
// Only blocks that lie entirely inside the clip are prefetched (one simple, branch-free batch of
// loads); the few blocks that touch the np.pad region are fetched directly in ring_advance.
template <class Cfg> LRA_HD bool ring_block_prefetchable(const StftArgs<typename Cfg::real>& a, int next) {
    constexpr int N = Cfg::N;
    const int H = a.hop, Hn = H < N ? H : N;
    if (2 * Hn > N || next >= a.n_frames) return false;
    const long long g1 = (long long)next * H + (N - Hn) - a.pad;
    return g1 >= 0 && g1 + Hn <= a.n;
}

template <class Cfg, bool RA> LRA_HD void stft_ring_prefetch(const StftArgs<typename Cfg::real>& a, int clip, int next, int tf, FftRegs<Cfg>& rg) {
    using T = typename Cfg::real;
    constexpr int N = Cfg::N, NPF = FftRegs<Cfg>::NPF;
    if (!ring_block_prefetchable<Cfg>(a, next)) return;
    const int H = a.hop, Hn = H < N ? H : N;
    const T* __restrict__ src = a.y + (long long)clip * a.y_stride + ((long long)next * H + (N - Hn) - a.pad);
    if (RA) {  // hop = n_fft / 4 = NPF rows of TF samples: one per-thread base, immediate offsets
        const T* __restrict__ st = src + tf;
        LRA_UNROLL
        for (int c = 0; c < NPF; ++c) rg.pf[c] = st[c * Cfg::TF];
        return;
    }
    // two batches of NPF loads: the second only for hops above n_fft/4 (uniform branch)
    LRA_UNROLL
    for (int c = 0; c < NPF; ++c) {
        const int e = tf + c * Cfg::TF;
        // lanes beyond the block re-read its last sample (in bounds) instead of branching
        rg.pf[c] = src[e < Hn ? e : Hn - 1];
    }
    if (Hn > NPF * Cfg::TF) {
        LRA_UNROLL
        for (int c = NPF; c < FftRegs<Cfg>::NPFX; ++c) {
            const int e = tf + c * Cfg::TF;
            rg.pf[c] = src[e < Hn ? e : Hn - 1];
        }
    }
}

// Row-aligned ring addressing (RA kernels: hop = n_fft / 4, librosa's default).  View the ring as R rows
// of TF complex (= 2 TF real) samples.  The hop is R/4 whole rows, so every frame starts on a row
// boundary: thread tf's R sample pairs of a frame sit at byte  ((rot + row) mod R) * ROWB + 8 tf  with
// ONE wave-uniform rotation rot in {0, R/4, R/2, 3R/4}; a scalar switch on rot turns every ring access
// into "constant per-thread base + immediate offset" (and the pair reads into single ds_read_b64s)
// instead of R masked address computations per frame.  Other hops use the general addressing.
template <class Cfg> LRA_HD bool ring_rows_aligned(int hop) { return 4 * hop == Cfg::N && Cfg::R >= 4; }

template <class Cfg, int ROT> LRA_HD void ring_rows_load(int rot, int tf, typename Cfg::cplx* v, Lds ring) {
    using C = typename Cfg::cplx;
    constexpr int lr = Cfg::logr(0), r = 1 << lr, nb = Cfg::R >> lr, ROWB = Cfg::TF * (int)sizeof(C);
    if (rot == ROT) {
        LRA_UNROLL
        for (int i = 0; i < nb; ++i) {
            LRA_UNROLL
            for (int j = 0; j < r; ++j) v[i * r + j] = lds_ld<C>(ring, tf * (int)sizeof(C) + ((ROT + i + j * nb) & (Cfg::R - 1)) * ROWB);
        }
    } else if constexpr (ROT + Cfg::R / 4 < Cfg::R) {
        ring_rows_load<Cfg, ROT + Cfg::R / 4>(rot, tf, v, ring);
    }
}

template <class Cfg, int ROT> LRA_HD void ring_rows_store(int rot, int tf, const FftRegs<Cfg>& rg, Lds ring) {
    using T = typename Cfg::real;
    constexpr int NPF = FftRegs<Cfg>::NPF, ROWB = 2 * Cfg::TF * (int)sizeof(T);
    if (rot == ROT) {
        LRA_UNROLL
        for (int c = 0; c < NPF; ++c)
            lds_st<T>(ring, tf * (int)sizeof(T) + ((ROT + (c >> 1)) & (Cfg::R - 1)) * ROWB + (c & 1) * Cfg::TF * (int)sizeof(T), rg.pf[c]);
    } else if constexpr (ROT + Cfg::R / 4 < Cfg::R) {
        ring_rows_store<Cfg, ROT + Cfg::R / 4>(rot, tf, rg, ring);
    }
}

template <class Cfg, bool RA> LRA_HD void stft_ring_advance(const StftArgs<typename Cfg::real>& a, int clip, int next, int tf, const FftRegs<Cfg>& rg, Lds ring) {
    using T = typename Cfg::real;
    constexpr int N = Cfg::N, NPF = FftRegs<Cfg>::NPF;
    const int H = a.hop, Hn = H < N ? H : N;
    if (next >= a.n_frames) return;
    const long long p1 = (long long)next * H + (N - Hn);
    if (ring_block_prefetchable<Cfg>(a, next)) {
        if (RA) {
            int rot = (int)((p1 & (N - 1)) / (2 * Cfg::TF));
            if (Cfg::TF >= 64 || a.rot_uniform) rot = LRA_UNIFORM(rot);  // wave-uniform: a wave never spans two slots, or all its slots are in step
            ring_rows_store<Cfg, 0>(rot, tf, rg, ring);
            return;
        }
        LRA_UNROLL
        for (int c = 0; c < FftRegs<Cfg>::NPFX; ++c) {
            const int e = tf + c * Cfg::TF;
            if (e < Hn) lds_st<T>(ring, (int)((p1 + e) & (N - 1)) * (int)sizeof(T), rg.pf[c]);
        }
    } else {
        const T* __restrict__ yb = a.y + (long long)clip * a.y_stride;
        for (int e = tf; e < Hn; e += Cfg::TF) lds_st<T>(ring, (int)((p1 + e) & (N - 1)) * (int)sizeof(T), fetch_sample<T>(yb, p1 + e, a.pad, a.n, a.pad_mode));
    }
}

// phase B: frame samples (ring) x window -> registers -> pass-0 butterflies -> frame area
template <class Cfg, bool RA> LRA_HD void stft_ring_load_pass0(const StftArgs<typename Cfg::real>& a, int frame, int tf, FftRegs<Cfg>& rg, Lds ring, Lds fr) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int lr = Cfg::logr(0), r = 1 << lr, nb = Cfg::R >> lr, sin = Cfg::M >> lr, N = Cfg::N;
    const C* __restrict__ win2 = reinterpret_cast<const C*>(a.win);
    C* v = rg.v;
    const int base = (int)(((long long)frame * a.hop) & (N - 1));
    if (Cfg::FPB > 1 && frame >= a.n_frames) {  // (with one slot per workgroup the frame loop has already exited)
        LRA_UNROLL
        for (int i = 0; i < Cfg::R; ++i) v[i] = mk<T>((T)0, (T)0);
    } else if (RA) {
        int rot = base / (2 * Cfg::TF);
        if (Cfg::TF >= 64 || a.rot_uniform) rot = LRA_UNIFORM(rot);  // wave-uniform: a wave never spans two slots, or all its slots are in step
        ring_rows_load<Cfg, 0>(rot, tf, v, ring);
        LRA_UNROLL
        for (int i = 0; i < nb; ++i) {
            LRA_UNROLL
            for (int j = 0; j < r; ++j) {
                const C w = Cfg::HOIST ? rg.win2[i * r + j] : win2[tf + i * Cfg::TF + j * sin];
                v[i * r + j] = mk<T>(v[i * r + j].x * w.x, v[i * r + j].y * w.y);
            }
        }
    } else if ((base & 1) == 0) {
        LRA_UNROLL
        for (int i = 0; i < nb; ++i) {
            LRA_UNROLL
            for (int j = 0; j < r; ++j) {
                const int q = tf + i * Cfg::TF + j * sin;  // complex index: samples 2q, 2q+1 of the frame
                const C x = lds_ld<C>(ring, ((base + 2 * q) & (N - 1)) * (int)sizeof(T));
                const C w = Cfg::HOIST ? rg.win2[i * r + j] : win2[q];
                v[i * r + j] = mk<T>(x.x * w.x, x.y * w.y);
            }
        }
    } else {  // odd hop: the sample pair is not 8-byte aligned in the ring
        LRA_UNROLL
        for (int i = 0; i < nb; ++i) {
            LRA_UNROLL
            for (int j = 0; j < r; ++j) {
                const int q = tf + i * Cfg::TF + j * sin;
                const T x0 = lds_ld<T>(ring, ((base + 2 * q) & (N - 1)) * (int)sizeof(T));
                const T x1 = lds_ld<T>(ring, ((base + 2 * q + 1) & (N - 1)) * (int)sizeof(T));
                const C w = Cfg::HOIST ? rg.win2[i * r + j] : win2[q];
                v[i * r + j] = mk<T>(x0 * w.x, x1 * w.y);
            }
        }
    }
#if LRA_ABLATE != 2
    pass_dft<Cfg, 0>(rg, tf, a.tw);
#endif
    pass_write<Cfg, 0>(v, fr, tf);
}

// ---- direct framing (stft_kernel<..., RA = 2>): hop >= n_fft --------------------------------------------------------------
// Frames do not overlap, so there is nothing for a ring to keep: frame t+1's R sample pairs per thread are loaded straight
// into registers (pass-0 order: element i r + j is complex index tf + i TF + j sin) at the START of frame t -- before frame
// t's stores are issued, so that the wait for them one frame later never covers fresh stores (vmcnt completes in order) --
// and the slot's LDS is the frame area only.  Frames that touch the np.pad region (core/spectrum.py:287) take the index fold.
// (The ring kernels fetched such hops with one dependent load per sample: n_fft = 512, hop = 512 ran at 1.3 TB/s.)
template <class Cfg> LRA_HD void stft_direct_fetch(const StftArgs<typename Cfg::real>& a, int clip, int frame, int tf, FftRegs<Cfg>& rg) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int lr = Cfg::logr(0), r = 1 << lr, nb = Cfg::R >> lr, sin = Cfg::M >> lr, N = Cfg::N;
    if (frame >= a.n_frames) return;
    const T* __restrict__ yb = a.y + (long long)clip * a.y_stride;
    const long long p0 = (long long)frame * a.hop, g0 = p0 - a.pad;
    if (g0 >= 0 && g0 + N <= a.n) {
        const T* __restrict__ src = yb + g0 + 2 * tf;
        if ((reinterpret_cast<size_t>(src) & (2 * sizeof(T) - 1)) == 0) {
            LRA_UNROLL
            for (int i = 0; i < nb; ++i) {
                LRA_UNROLL
                for (int j = 0; j < r; ++j) rg.nxt[i * r + j] = *reinterpret_cast<const C*>(src + 2 * (i * Cfg::TF + j * sin));
            }
        } else {
            LRA_UNROLL
            for (int i = 0; i < nb; ++i) {
                LRA_UNROLL
                for (int j = 0; j < r; ++j) {
                    const T* __restrict__ s2 = src + 2 * (i * Cfg::TF + j * sin);
                    rg.nxt[i * r + j] = mk<T>(s2[0], s2[1]);
                }
            }
        }
    } else {
        LRA_UNROLL
        for (int i = 0; i < nb; ++i) {
            LRA_UNROLL
            for (int j = 0; j < r; ++j) {
                const long long pp = p0 + 2 * (tf + i * Cfg::TF + j * sin);
                rg.nxt[i * r + j] = mk<T>(fetch_sample<T>(yb, pp, a.pad, a.n, a.pad_mode), fetch_sample<T>(yb, pp + 1, a.pad, a.n, a.pad_mode));
            }
        }
    }
}

// phase B of a directly framed kernel: the pairs loaded a frame ago x window -> pass-0 butterflies -> frame area; then the
// next frame's loads are issued
template <class Cfg> LRA_HD void stft_direct_pass0(const StftArgs<typename Cfg::real>& a, int clip, int frame, bool more, int tf, FftRegs<Cfg>& rg, Lds fr) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int lr = Cfg::logr(0), r = 1 << lr, nb = Cfg::R >> lr, sin = Cfg::M >> lr;
    const C* __restrict__ win2 = reinterpret_cast<const C*>(a.win);
    // (a frame beyond n_frames is transformed like any other -- stale or zero samples (the prologue zero-fills rg.nxt), nothing of it is
    // stored: no zeroing selects, which hipcc emits as back-to-back VOP2 v_cndmask pairs; see sel_mask in lra_common.h)
    LRA_UNROLL
    for (int i = 0; i < nb; ++i) {
        LRA_UNROLL
        for (int j = 0; j < r; ++j) {
            const C w = Cfg::HOIST ? rg.win2[i * r + j] : win2[tf + i * Cfg::TF + j * sin];
            const C x = rg.nxt[i * r + j];
            rg.v[i * r + j] = mk<T>(x.x * w.x, x.y * w.y);
        }
    }
    if (more) stft_direct_fetch<Cfg>(a, clip, frame + 1, tf, rg);
    pass_dft<Cfg, 0>(rg, tf, a.tw);
    pass_write<Cfg, 0>(rg.v, fr, tf);
}

// ---- register ring (stft_kernel<..., RA = 2 + log2 HD>, HD = 2 .. 16): hop = n_fft / HD with the hop a whole number of pass-0 rows --
// The idea of the second-generation kernel (lra_kernels2.h) for the configurations whose last pass it cannot mirror (n_fft >=
// 8192: 16 . 16 . 16): frame t+1's pass-0 element (tf, i, j) is frame t's element (tf, i, j + r0/HD), so the samples live in
// 2R registers per thread, advanced by register moves, and only the R/HD new pairs per frame are loaded (one frame ahead,
// before the frame's stores).  No ring in LDS: at n_fft = 8192 the slot shrinks from 67 KB to 35 KB (4 workgroups of four waves
// per CU instead of 2) and the general ring addressing (a masked address per element and frame) disappears.
template <class Cfg, int HD> struct RegRing {
    static constexpr int lr = Cfg::logr(0), r0 = 1 << lr, nb0 = Cfg::R >> lr, sin0 = Cfg::M >> lr;
    static constexpr int SJ = r0 / HD;       // pass-0 rows a frame advances by
    static constexpr int NEW = Cfg::R / HD;  // new sample pairs per thread and frame
    static constexpr bool ok = HD >= 2 && r0 % HD == 0 && Cfg::R % HD == 0;
    static LRA_HD int q_of(int tf, int e) { return tf + (e / r0) * Cfg::TF + (e % r0) * sin0; }  // complex index of register element e
    static LRA_HD int elem_of_new(int n) { return (n / SJ) * r0 + (r0 - SJ) + (n % SJ); }         // register element of new pair n
};
template <class Cfg> LRA_HD int regring_hd(int hop) {
    for (int hd = 2; hd <= 16; hd *= 2)
        if ((long long)hop * hd == Cfg::N && (1 << Cfg::logr(0)) % hd == 0 && Cfg::R % hd == 0) return hd;
    return 0;
}

// all R pairs of a slot's first frame -> rg.raw
template <class Cfg, int HD> LRA_HD void regring_fill(const StftArgs<typename Cfg::real>& a, int clip, int frame, int tf, FftRegs<Cfg>& rg) {
    using T = typename Cfg::real;
    const T* __restrict__ yb = a.y + (long long)clip * a.y_stride;
    const long long p0 = (long long)frame * a.hop;
    const bool live = frame < a.n_frames;
    LRA_UNROLL
    for (int e = 0; e < Cfg::R; ++e) {
        const long long p = p0 + 2 * RegRing<Cfg, HD>::q_of(tf, e);
        rg.raw[e] = live ? mk<T>(fetch_sample<T>(yb, p, a.pad, a.n, a.pad_mode), fetch_sample<T>(yb, p + 1, a.pad, a.n, a.pad_mode)) : mk<T>((T)0, (T)0);
    }
}

// the NEW pairs of frame `next` -> rg.nxt (issued one frame ahead)
template <class Cfg, int HD> LRA_HD void regring_issue(const StftArgs<typename Cfg::real>& a, int clip, int next, int tf, FftRegs<Cfg>& rg) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    using RR = RegRing<Cfg, HD>;
    if (next >= a.n_frames) return;
    const T* __restrict__ yb = a.y + (long long)clip * a.y_stride;
    const long long p0 = (long long)next * a.hop;
    const long long g0 = p0 + (Cfg::N - a.hop) - a.pad;  // clip position of the first NEW sample
    if (g0 >= 0 && g0 + a.hop <= a.n) {
        const T* __restrict__ src = yb + (p0 - a.pad) + 2 * tf;
        if ((reinterpret_cast<size_t>(src) & (2 * sizeof(T) - 1)) == 0) {
            LRA_UNROLL
            for (int n = 0; n < RR::NEW; ++n) {
                const int e = RR::elem_of_new(n);
                rg.nxt[n] = *reinterpret_cast<const C*>(src + 2 * ((e / RR::r0) * Cfg::TF + (e % RR::r0) * RR::sin0));
            }
        } else {
            LRA_UNROLL
            for (int n = 0; n < RR::NEW; ++n) {
                const int e = RR::elem_of_new(n);
                const T* __restrict__ s2 = src + 2 * ((e / RR::r0) * Cfg::TF + (e % RR::r0) * RR::sin0);
                rg.nxt[n] = mk<T>(s2[0], s2[1]);
            }
        }
    } else {
        LRA_UNROLL
        for (int n = 0; n < RR::NEW; ++n) {
            const long long p = p0 + 2 * RR::q_of(tf, RR::elem_of_new(n));
            rg.nxt[n] = mk<T>(fetch_sample<T>(yb, p, a.pad, a.n, a.pad_mode), fetch_sample<T>(yb, p + 1, a.pad, a.n, a.pad_mode));
        }
    }
}

// phase B: advance the register ring (not for a slot's first frame), start the next frame's loads, window, pass 0
template <class Cfg, int HD> LRA_HD void regring_pass0(const StftArgs<typename Cfg::real>& a, int clip, int frame, bool first, bool more, int tf, FftRegs<Cfg>& rg, Lds fr) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    using RR = RegRing<Cfg, HD>;
    const C* __restrict__ win2 = reinterpret_cast<const C*>(a.win);
    if (!first) {
        LRA_UNROLL
        for (int i = 0; i < RR::nb0; ++i) {
            LRA_UNROLL
            for (int j = 0; j < RR::r0; ++j) rg.raw[i * RR::r0 + j] = j + RR::SJ < RR::r0 ? rg.raw[i * RR::r0 + j + RR::SJ] : rg.nxt[i * RR::SJ + (j + RR::SJ - RR::r0)];
        }
    }
    if (more) regring_issue<Cfg, HD>(a, clip, frame + 1, tf, rg);
    LRA_UNROLL
    for (int e = 0; e < Cfg::R; ++e) {  // (no zeroing of frames beyond n_frames: see stft_direct_pass0)
        const C w = Cfg::HOIST ? rg.win2[e] : win2[RR::q_of(tf, e)];
        rg.v[e] = mk<T>(rg.raw[e].x * w.x, rg.raw[e].y * w.y);
    }
    pass_dft<Cfg, 0>(rg, tf, a.tw);
    pass_write<Cfg, 0>(rg.v, fr, tf);
}

// ---- phase: read Z[k], Z[M-k] pairs for the split step ----------------------------------------
template <class Cfg> LRA_HD void split_read(FftRegs<Cfg>& rg, Lds fr, int tf) {
    using C = typename Cfg::cplx;
    if constexpr (affine_tf<Cfg>()) {
        // k = tf + i TF and M - k: two runtime bases, immediate offsets (see lra_fft.h)
        const int bk = Cfg::phys(tf) * (int)sizeof(C);
        const int bm0 = Cfg::phys((Cfg::M - tf) & (Cfg::M - 1)) * (int)sizeof(C);  // i = 0 (wraps to 0 for tf = 0)
        const int bm = (Cfg::M - tf + ((Cfg::M - tf) >> Cfg::PADSHIFT)) * (int)sizeof(C);
        LRA_UNROLL
        for (int i = 0; i < Cfg::R / 2; ++i) {
            rg.v[2 * i] = lds_ld<C>(fr, bk + i * pstride<Cfg>(Cfg::TF) * (int)sizeof(C));
            rg.v[2 * i + 1] = lds_ld<C>(fr, i == 0 ? bm0 : bm - i * pstride<Cfg>(Cfg::TF) * (int)sizeof(C));
        }
    } else {
        LRA_UNROLL
        for (int i = 0; i < Cfg::R / 2; ++i) {
            const int k = tf + i * Cfg::TF;
            const int km = (Cfg::M - k) & (Cfg::M - 1);
            rg.v[2 * i] = lds_ld<C>(fr, Cfg::phys(k) * (int)sizeof(C));
            rg.v[2 * i + 1] = lds_ld<C>(fr, Cfg::phys(km) * (int)sizeof(C));
        }
    }
    if (tf == 0) rg.mid = lds_ld<C>(fr, Cfg::phys(Cfg::M / 2) * (int)sizeof(C));
}

// X[k] and X[M-k] from Z[k], Z[M-k]:  A = (Zk + conj Zm)/2, B = W_N^k (Zk - conj Zm)/(2i),
// X[k] = A + B, X[M-k] = conj(A - B).  The factor 1/2 is folded into the window table of the fused
// kernels (StftArgs::win holds 0.5 w, an exact scaling), so Z arrives pre-halved here.
template <class T> LRA_HD void split_pair(cx<T> zk, cx<T> zm, cx<T> w, cx<T>& xk, cx<T>& xm) {
    const cx<T> A = add_conj(zk, zm);
    const cx<T> B = cmul2_mi(sub_conj(zk, zm), w);  // W (-i)(Zk - conj Zm)
    xk = cadd(A, B);
    xm = conj_sub(A, B);
}

// ---- phase: split + epilogue store (complex / power) or power -> LDS (mel) --------------------
template <class Cfg, int MODE, int PM> LRA_HD void stft_split_store(const StftArgs<typename Cfg::real>& a, int clip, int frame, bool valid, int tf,
                                                            FftRegs<Cfg>& rg, Lds fr, Lds sh) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int M = Cfg::M;
    const long long row = ((long long)clip * a.n_frames + frame) * a.row_pitch;
    // bins k = tf + i TF ascend from one per-thread pointer, the mirrored bins M - k descend from another:
    // two 64-bit bases per frame and immediate offsets for all R stores
    C* __restrict__ const Dk = MODE == OUT_COMPLEX ? a.D + row + tf : nullptr;
    C* __restrict__ const Dm = MODE == OUT_COMPLEX ? a.D + row + (M - tf) : nullptr;
    T* __restrict__ const Sk = MODE == OUT_POWER ? a.S + row + tf : nullptr;
    T* __restrict__ const Sm = MODE == OUT_POWER ? a.S + row + (M - tf) : nullptr;
    LRA_UNROLL
    for (int i = 0; i < Cfg::R / 2; ++i) {
        const int k = tf + i * Cfg::TF;
        C xk, xm;
        int km;
        if (k == 0) {
            const C z0 = rg.v[2 * i];
            xk = mk<T>((T)2 * (z0.x + z0.y), (T)0);  // Z is pre-halved (see split_pair)
            xm = mk<T>((T)2 * (z0.x - z0.y), (T)0);
            km = M;
            if (valid && a.nonfinite_flag && !(std::fabs(xk.x) <= std::numeric_limits<T>::max())) LRA_ATOMIC_OR(a.nonfinite_flag, 1u);
        } else {
#if LRA_ABLATE == 2  // experiment: stores only
            xk = rg.v[2 * i]; xm = rg.v[2 * i + 1];
#else
            split_pair<T>(rg.v[2 * i], rg.v[2 * i + 1], Cfg::HOIST ? rg.twr[i] : a.twr[k], xk, xm);
#endif
            km = M - k;
        }
        if (MODE == OUT_COMPLEX) {
#if LRA_ABLATE == 1 || LRA_ABLATE == 4  // experiment: compute everything, store (practically) nothing
            if (valid && xk.x == (T)12345.678) { Dk[i * Cfg::TF] = xk; Dm[-i * Cfg::TF] = xm; }
#else
            if (valid) { stream_store(&Dk[i * Cfg::TF], xk); stream_store(&Dm[-i * Cfg::TF], xm); }
#endif
        } else {
            const T pk = spec_power<T, PM>(xk, a.power), pm = spec_power<T, PM>(xm, a.power);
            if (MODE == OUT_POWER) {
                if (valid) { Sk[i * Cfg::TF] = pk; Sm[-i * Cfg::TF] = pm; }
            } else if (MODE == OUT_MEL2) {
                const C wk = lds_ld<C>(sh, k * (int)sizeof(C)), wm = lds_ld<C>(sh, km * (int)sizeof(C));
                // AB[k] sits at the padded slot k + k/R: a thread's run of R consecutive bins then starts
                // R+1 slots after its neighbour's, which makes the run reads of mel2_gather conflict-free
#if LRA_ABLATE == 13  // experiment: no mel epilogue at all (keep the power computation alive)
                if (wk.x * pk + wm.x * pm == (T)12345.678) a.Mel[k] = pk;
#else
                lds_st<C>(fr, ab_slot<Cfg>(k) * (int)sizeof(C), mk<T>(wk.x * pk, wk.y * pk));
                lds_st<C>(fr, ab_slot<Cfg>(km) * (int)sizeof(C), mk<T>(wm.x * pm, wm.y * pm));
#endif
            } else {
                lds_st<T>(fr, k * (int)sizeof(T), pk);
                lds_st<T>(fr, km * (int)sizeof(T), pm);
            }
        }
    }
    if (tf == 0) {
        const C xmid = mk<T>((T)2 * rg.mid.x, (T)-2 * rg.mid.y);  // X[M/2] = conj(Z[M/2]), Z pre-halved
        if (MODE == OUT_COMPLEX) {
            if (valid) a.D[row + M / 2] = xmid;
        } else {
            const T pmid = spec_power<T, PM>(xmid, a.power);
            if (MODE == OUT_POWER) {
                if (valid) a.S[row + M / 2] = pmid;
            } else if (MODE == OUT_MEL2) {
                const C wq = lds_ld<C>(sh, (M / 2) * (int)sizeof(C));
                lds_st<C>(fr, ab_slot<Cfg>(M / 2) * (int)sizeof(C), mk<T>(wq.x * pmid, wq.y * pmid));
            } else {
                lds_st<T>(fr, (M / 2) * (int)sizeof(T), pmid);
            }
        }
    }
}

// ---- last pass mirrored across the halves of a wave (complex / power epilogues without an LDS ring; n_fft = 8192) -------------------
// The second-generation kernel's idea (lra_kernels2.h: Z[k] and Z[M-k] come out of the last pass in ONE thread, so the split step
// needs no LDS round trip) for a last pass of ONE radix-16 butterfly per thread, where the mirror s - b of butterfly b cannot sit in
// the same thread.  It sits in the same WAVE instead: lane l < 32 of wave w takes butterfly pi = 32 w + l, lane l + 32 takes s - pi
// (s = M / 16 = TF), and after the pass the two lanes exchange their upper eight outputs (v_permlane32_swap_b32, half_swap8): lane
// (b) then holds Z[b + j s], j < 8, and in v[15 - j] their mirrors Z[(s - b) + (15 - j) s] = Z[M - (b + j s)].  Pair pi = 0 is the
// two self-mirrored butterflies 0 and s/2 (thread 0 and thread 32): those two lanes stay out of the exchange; butterfly s/2 pairs
// j with 15 - j by itself, butterfly 0 pairs j with 16 - j, which thread 0 absorbs by acting as b = s with its registers shifted
// by one (eight lane-0 selects; its slot 7 is X[M/2], its v[0] gives X[0] and X[M]).
// Per frame and thread: 16 LDS writes + 16 reads and two workgroup barriers less than the split through LDS, for 16 swaps.
#ifndef LRA_MIRROR_TWR_BASE
#define LRA_MIRROR_TWR_BASE 1
#endif
template <class Cfg> constexpr bool mirror32_cfg_ok() {
    constexpr int pl = Cfg::P - 1;
    return Cfg::P >= 2 && Cfg::R == 16 && Cfg::HOIST && sizeof(typename Cfg::real) == 4 && Cfg::logr(pl) == 4 && Cfg::TF % 64 == 0 && Cfg::TF == (Cfg::M >> 4) &&
           Cfg::TF == Cfg::NT && affine_tf<Cfg>();
}
template <class Cfg> LRA_HD int mirror32_bfly(int tf) {
    const int pi = (tf >> 6) * 32 + (tf & 31);
    return (tf & 32) == 0 ? pi : (pi == 0 ? Cfg::TF / 2 : Cfg::TF - pi);
}
// first bin of the thread's eight pair slots (slot q: bins kb + q s and M - kb - q s)
template <class Cfg> LRA_HD int mirror32_kbase(int tf) { return tf == 0 ? Cfg::TF : mirror32_bfly<Cfg>(tf); }
template <class Cfg> LRA_HD bool mirror32_swaps(int tf) { return tf != 0 && tf != 32; }

// prologue: the last pass's twiddles of butterfly mirror32_bfly(tf) and the split twiddles of the thread's pair slots replace
// what hoist_tables loaded for butterfly tf
template <class Cfg> LRA_HD void mirror32_hoist(FftRegs<Cfg>& rg, int tf, const typename Cfg::cplx* __restrict__ tw, const typename Cfg::cplx* __restrict__ twr) {
    if constexpr (mirror32_cfg_ok<Cfg>()) {
        load_pass_twiddles<Cfg, Cfg::P - 1>(rg.treg, mirror32_bfly<Cfg>(tf), tw);
        const int kb = mirror32_kbase<Cfg>(tf);
        LRA_UNROLL
        for (int q = 0; q < (LRA_MIRROR_TWR_BASE ? 1 : Cfg::R / 2); ++q) rg.twr[q] = twr[kb + q * Cfg::TF];
    }
}
// split twiddle of pair slot q: W_N^(kb + q s) = W_N^kb W_32^q (s = N / 32).  Only W_N^kb is kept in registers -- with all eight the
// kernel needs 182 VGPRs, two workgroups per CU instead of three; seven complex products per frame are the cheaper side.
template <class Cfg> LRA_HD typename Cfg::cplx mirror32_twr(const FftRegs<Cfg>& rg, int q) {
    using T = typename Cfg::real;
    if (!LRA_MIRROR_TWR_BASE) return rg.twr[q];
    constexpr double c32[8] = {1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708, 0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173, 0.19509032201612826785};
    if (q == 0) return rg.twr[0];
    const cx<T> c = mk<T>((T)c32[q], (T)-c32[8 - q]);  // exp(-2 pi i q / 32): cos = c32[q], sin = c32[8 - q]
    return cmul_f(rg.twr[0], c, cmul_p(rg.twr[0], c));
}

// split + store from the exchanged butterfly registers
template <class Cfg, int MODE, int PM> LRA_HD void mirror32_split_store(const StftArgs<typename Cfg::real>& a, int clip, int frame, bool valid, int tf, FftRegs<Cfg>& rg) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int M = Cfg::M, S = Cfg::TF;
    const bool l0 = tf == 0;
    const LaneMask l0m = lane_mask(l0);
    const long long row = ((long long)clip * a.n_frames + frame) * a.row_pitch;
    const int kb = mirror32_kbase<Cfg>(tf);
    C* __restrict__ const Dk = MODE == OUT_COMPLEX ? a.D + row + kb : nullptr;
    C* __restrict__ const Dm = MODE == OUT_COMPLEX ? a.D + row + (M - kb) : nullptr;
    T* __restrict__ const Sk = MODE == OUT_POWER ? a.S + row + kb : nullptr;
    T* __restrict__ const Sm = MODE == OUT_POWER ? a.S + row + (M - kb) : nullptr;
    LRA_UNROLL
    for (int q = 0; q < Cfg::R / 2; ++q) {
        const C zk = sel_mask(l0m, l0, rg.v[q + 1], rg.v[q]);
        C xk, xm;
        split_pair<T>(zk, rg.v[Cfg::R - 1 - q], mirror32_twr<Cfg>(rg, q), xk, xm);
        const bool both = q + 1 < Cfg::R / 2 || !l0;  // thread 0's last slot is the single bin M/2
        if (q + 1 == Cfg::R / 2) xk = sel_mask(l0m, l0, mk<T>((T)2 * rg.v[Cfg::R / 2].x, (T)-2 * rg.v[Cfg::R / 2].y), xk);  // X[M/2] = conj(Z[M/2]), Z pre-halved
        if (MODE == OUT_COMPLEX) {
            if (valid) {
                stream_store(&Dk[q * S], xk);
                if (both) stream_store(&Dm[-q * S], xm);
            }
        } else {
            const T pk = spec_power<T, PM>(xk, a.power), pm = spec_power<T, PM>(xm, a.power);
            if (valid) {
                Sk[q * S] = pk;
                if (both) Sm[-q * S] = pm;
            }
        }
    }
    if (l0) {  // X[0], X[M] from Z[0]
        const C z0 = rg.v[0];
        const C x0 = mk<T>((T)2 * (z0.x + z0.y), (T)0), xn = mk<T>((T)2 * (z0.x - z0.y), (T)0);
        if (valid && a.nonfinite_flag && !(std::fabs(x0.x) <= std::numeric_limits<T>::max())) LRA_ATOMIC_OR(a.nonfinite_flag, 1u);
        if (valid) {
            if (MODE == OUT_COMPLEX) { a.D[row] = x0; a.D[row + M] = xn; }
            else { a.S[row] = spec_power<T, PM>(x0, a.power); a.S[row + M] = spec_power<T, PM>(xn, a.power); }
        }
    }
}

// ---- phase: banded mel reduce of ONE frame (its power spectrum sits in the slot's frame area) ----
// Thread tf of the slot owns mel rows tf, tf+TF, ...; results go to the slot's staging tile
// stage[m][it] so that the flush after the frame loop writes rows of `iters` consecutive frames.
template <class Cfg> LRA_HD void mel_reduce_slot(const StftArgs<typename Cfg::real>& a, int tf, int it, int iters, Lds fr, Lds stage) {
    using T = typename Cfg::real;
    for (int m = tf; m < a.n_mels; m += Cfg::TF) {
        const int c0 = a.mel_c0[m], len = a.mel_len[m];
        const T* __restrict__ w = a.mel_val + a.mel_off[m];
        T acc = (T)0;
        for (int i = 0; i < len; ++i) acc += w[i] * lds_ld<T>(fr, (c0 + i) * (int)sizeof(T));
        lds_st<T>(stage, (m * iters + it) * (int)sizeof(T), acc);
    }
}

// ---- phase: flush the slot's staging tile: M[clip][m][f0 .. f0+nv) ------------------------------
template <class Cfg> LRA_HD void mel_flush_slot(const StftArgs<typename Cfg::real>& a, int clip, int f0, int nv, int tf, int iters, Lds stage) {
    using T = typename Cfg::real;
    const int total = a.n_mels * iters;
    for (int idx = tf; idx < total; idx += Cfg::TF) {
        const int m = idx / iters, i = idx - m * iters;
        if (i < nv) a.Mel[((long long)clip * a.n_mels + m) * a.n_frames + f0 + i] = lds_ld<T>(stage, idx * (int)sizeof(T));
    }
}

// flush of the staging tile that iteration `it` of a slot (first frame f_slot) belongs to: the frames
// staged so far in that tile, clipped to the clip's frame count
template <class Cfg> LRA_HD void mel_flush_tile(const StftArgs<typename Cfg::real>& a, int clip, int f_slot, int it, int tile, int tf, Lds stage) {
#if LRA_ABLATE == 14
    if (a.n_mels != 12345) return;
#endif
    const int f0 = f_slot + (it / tile) * tile;  // first frame of the staged tile
    int nv = a.n_frames - f0;
    const int staged = it % tile + 1;
    nv = nv < 0 ? 0 : (nv > staged ? staged : nv);
    mel_flush_slot<Cfg>(a, clip, f0, nv, tf, tile, stage);
}

// ---- OUT_MEL2: two-slope mel reduce (lra_mel.h) ---------------------------------------------------
// Shared region (per workgroup): wAB[M+1] (float2) | run_desc[TF] (int) | seg_desc[n_mels+1] (int).
// Per slot: the frame area holds AB[k] = (wA[k] P[k], wB[k] P[k]) after the split step; the piece sums
// live in a small extra region at the end of the slot.
template <class Cfg> LRA_HD int mel2_shared_bytes(int n_mels) {
    return ((2 * (Cfg::M + 1) * (int)sizeof(typename Cfg::real) + (Cfg::TF + n_mels + 1) * (int)sizeof(int) + 15) / 16) * 16;
}
template <class Cfg> LRA_HD int mel2_psum_bytes(int n_mels) { return ((2 * (Cfg::TF + n_mels + 2) * (int)sizeof(typename Cfg::real) + 15) / 16) * 16; }
template <class Cfg> inline bool mel2_fits(int) {
    return (Cfg::M + (Cfg::M >> Cfg::LOGR) + 1) * 2 * (int)sizeof(typename Cfg::real) <= Cfg::FRAME_BYTES && (Cfg::R == 8 || Cfg::R == 16);
}

template <class Cfg> LRA_HD void mel2_tables_to_lds(const StftArgs<typename Cfg::real>& a, int tid, Lds sh) {
    using C = typename Cfg::cplx;
    constexpr int NB = Cfg::M + 1;
    const C* __restrict__ w2 = reinterpret_cast<const C*>(a.mel_wAB);
    for (int k = tid; k < NB; k += Cfg::NT) lds_st<C>(sh, k * (int)sizeof(C), w2[k]);
    for (int i = tid; i < Cfg::TF; i += Cfg::NT) lds_st<int>(sh, NB * (int)sizeof(C) + i * (int)sizeof(int), a.mel_run[i]);
    for (int i = tid; i <= a.n_mels; i += Cfg::NT) lds_st<int>(sh, NB * (int)sizeof(C) + (Cfg::TF + i) * (int)sizeof(int), a.mel_segd[i]);
}

// piece sums: thread tf reads its run of R consecutive bins (contiguous 8R bytes, all reads issued at
// once) and emits one (sum A, sum B) pair per segment that the run touches
template <class Cfg> LRA_HD void mel2_gather(const StftArgs<typename Cfg::real>& a, int tf, Lds fr, Lds sh, Lds psum) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int NB = Cfg::M + 1, BPL = Cfg::R;
    const int d = lds_ld<int>(sh, NB * (int)sizeof(C) + tf * (int)sizeof(int));
    int pid = d & 0xfff;
    const int mask = d >> 12;
    C v[BPL];
    LRA_UNROLL
    for (int j = 0; j < BPL; ++j) v[j] = lds_ld<C>(fr, (tf * (BPL + 1) + j) * (int)sizeof(C));  // = ab_slot(tf*BPL + j)
    C acc = mk<T>((T)0, (T)0);
    LRA_UNROLL
    for (int j = 0; j < BPL; ++j) {
        acc = cadd(acc, v[j]);
        if (j == BPL - 1 || ((mask >> j) & 1)) {
            lds_st<C>(psum, pid * (int)sizeof(C), acc);
            ++pid;
            acc = mk<T>((T)0, (T)0);
        }
    }
    if (tf == 0) lds_st<C>(psum, a.mel_nyq * (int)sizeof(C), lds_ld<C>(fr, ab_slot<Cfg>(Cfg::M) * (int)sizeof(C)));
}

// mel[m] = rising part (B over the pieces of segment m) + falling part (A over the pieces of segment
// m+1), pieces added in ascending bin order.  tile == 1: straight to M[clip][m][frame] (the rows of
// consecutive frames are merged by the L2); else into the staging tile.
template <class Cfg> LRA_HD void mel2_combine(const StftArgs<typename Cfg::real>& a, int clip, int frame, int tf, int it, int tile, Lds sh, Lds psum, Lds stage) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int NB = Cfg::M + 1;
#if LRA_ABLATE == 15
    if (a.n_mels != 12345) return;
#endif
    for (int m = tf; m < a.n_mels; m += Cfg::TF) {
        // both descriptors, then the first UNR pieces of both halves as ONE batch of (predicated) loads: a
        // single LDS round trip per mel band instead of one per chunk of each half; wider segments (rare:
        // few mel bands over a large n_fft) finish in the loop
        constexpr int UNR = 6;
        int first[2], cnt[2];
        LRA_UNROLL
        for (int h = 0; h < 2; ++h) {  // h = 0: B over segment m; h = 1: A over segment m + 1
            const int d = lds_ld<int>(sh, NB * (int)sizeof(C) + (Cfg::TF + m + h) * (int)sizeof(int));
            first[h] = d & 0xfff;
            cnt[h] = d >> 12;
        }
        T x[2][UNR];
        LRA_UNROLL
        for (int h = 0; h < 2; ++h) {
            LRA_UNROLL
            for (int q = 0; q < UNR; ++q) {
                const bool ok = q < cnt[h];
                x[h][q] = lds_ld<T>(psum, ((ok ? first[h] + q : 0) * 2 + (1 - h)) * (int)sizeof(T));  // piece 0 always exists
                if (!ok) x[h][q] = (T)0;
            }
        }
        T part[2];
        LRA_UNROLL
        for (int h = 0; h < 2; ++h) {
            T acc = (T)0;
            LRA_UNROLL
            for (int q = 0; q < UNR; ++q) acc += x[h][q];
            for (int q = UNR; q < cnt[h]; ++q) acc += lds_ld<T>(psum, ((first[h] + q) * 2 + (1 - h)) * (int)sizeof(T));
            part[h] = acc;
        }
        const T v = part[0] + part[1];
        if (tile == 1) a.Mel[((long long)clip * a.n_mels + m) * a.n_frames + frame] = v;
        else lds_st<T>(stage, (m * tile + (it % tile)) * (int)sizeof(T), v);
    }
}

// ---- OUT_MELR: run-ordered two-slope mel epilogue (lra_mel.h, MelRuns) -------------------------------------
// Shared region (per workgroup): w[(M+1) + pads] pairs | addr[2 pmax][n_mels] ints.
// Per slot: the frame area is reused as the running-sum area rs[R][TF] pairs (+ zero slot, + bin M/2's slot).
constexpr int MELR_PMAX = 16;  // longest piece list supported (the host falls back to OUT_MEL2 beyond); the lists are stored with their actual length
// the weight pairs are stored with two pad pairs per run of R/2 (index i -> i + 2 (i / (R/2))): a thread reads ITS run, so lanes
// are R/2 + 2 pairs = 80 bytes apart -- every run starts on a 16-byte boundary (two pairs per ds_read_b128) and the 16 lanes of
// a read group start 20 dwords apart, i.e. on 16 distinct 4-bank groups: conflict-free for 8- and 16-byte reads alike
template <class Cfg> LRA_HD int melr_w_slot(int i) { return i + 2 * (i / (Cfg::R / 2)); }
template <class Cfg> LRA_HD int melr_keep_off() { return ((melr_w_slot<Cfg>(Cfg::M) + 1) * 2 * (int)sizeof(typename Cfg::real) + 15) / 16 * 16; }
template <class Cfg> LRA_HD int melr_addr_off() { return melr_keep_off<Cfg>(); }  // (the restart factors live in registers only, melr_hoist)
// the address table is needed in LDS only for lists longer than the hoisted prefix or more than two bands per thread
constexpr int MELR_PHOIST_N = 4;  // piece-list entries per band kept in registers at two bands per thread (melr_ph)
template <class Cfg> LRA_HD bool melr_needs_table(int n_mels, int pmax) { return pmax > melr_ph<Cfg>() || n_mels > melr_nb<Cfg>() * Cfg::TF; }
template <class Cfg> LRA_HD int melr_shared_bytes(int n_mels, int pmax) {
    return ((melr_addr_off<Cfg>() + (melr_needs_table<Cfg>(n_mels, pmax) ? n_mels * 2 * pmax * (int)sizeof(int) : 0) + 15) / 16) * 16;
}
template <class Cfg> constexpr bool melr_fits() {
    return Cfg::R == 16 && (1 << Cfg::PADSHIFT) % (Cfg::R / 2) == 0 && (Cfg::R * Cfg::TF + 2) * 2 * (int)sizeof(typename Cfg::real) <= Cfg::FRAME_BYTES;
}

template <class Cfg> LRA_HD void melr_tables_to_lds(const StftArgs<typename Cfg::real>& a, int tid, Lds sh) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    const C* __restrict__ w2 = reinterpret_cast<const C*>(a.melr_w);
    for (int k = tid; k < Cfg::M + 1; k += Cfg::NT) lds_st<C>(sh, melr_w_slot<Cfg>(k) * (int)sizeof(C), w2[k]);
    if (melr_needs_table<Cfg>(a.n_mels, a.melr_pmax))
        for (int i = tid; i < a.n_mels * 2 * a.melr_pmax; i += Cfg::NT) lds_st<int>(sh, melr_addr_off<Cfg>() + i * (int)sizeof(int), a.melr_addr[i]);
}

// phase: Z[k], Z[M-k] for this thread's RUN of R/2 consecutive bins k = (R/2) tf + j  (v[2j], v[2j+1])
template <class Cfg> LRA_HD void split_read_runs(FftRegs<Cfg>& rg, Lds fr, int tf) {
    using C = typename Cfg::cplx;
    constexpr int BPL = Cfg::R / 2, M = Cfg::M;
    // a run of BPL <= 16 consecutive elements never straddles a pad slot (phys(i) = i + i/16): base + immediates
    const int ba = Cfg::phys(BPL * tf) * (int)sizeof(C);
    const int bm = Cfg::phys(M - BPL * tf - 1) * (int)sizeof(C);                 // j = 1 .. BPL-1 descend from here
    const int b0 = Cfg::phys((M - BPL * tf) & (M - 1)) * (int)sizeof(C);         // j = 0 (wraps to Z[0] for tf = 0)
    LRA_UNROLL
    for (int j = 0; j < BPL; ++j) {
        rg.v[2 * j] = lds_ld<C>(fr, ba + j * (int)sizeof(C));
        rg.v[2 * j + 1] = lds_ld<C>(fr, j == 0 ? b0 : bm - (j - 1) * (int)sizeof(C));
    }
    if (tf == 0) rg.mid = lds_ld<C>(fr, Cfg::phys(M / 2) * (int)sizeof(C));
}

// phase: split + power + (wA, wB) + running sums along both runs -> rs[jj][tf]
template <class Cfg, int PM> LRA_HD void melr_split_accumulate(const StftArgs<typename Cfg::real>& a, bool valid, int tf, FftRegs<Cfg>& rg, Lds rs, Lds sh) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int BPL = Cfg::R / 2, M = Cfg::M, TF = Cfg::TF;
    C ab[2 * BPL];
    LRA_UNROLL
    for (int j = 0; j < BPL; ++j) {
        const int k = BPL * tf + j;
        C xk, xm;
        if (k == 0) {
            const C z0 = rg.v[0];
            xk = mk<T>((T)2 * (z0.x + z0.y), (T)0);  // Z is pre-halved (see split_pair)
            xm = mk<T>((T)2 * (z0.x - z0.y), (T)0);
            if (valid && a.nonfinite_flag && !(std::fabs(xk.x) <= std::numeric_limits<T>::max())) LRA_ATOMIC_OR(a.nonfinite_flag, 1u);
        } else {
            split_pair<T>(rg.v[2 * j], rg.v[2 * j + 1], Cfg::HOIST ? rg.twr[j] : a.twr[k], xk, xm);
        }
        const T pk = spec_power<T, PM>(xk, a.power), pm = spec_power<T, PM>(xm, a.power);
        // (slot of pair i: i + i / BPL; both halves start on a run boundary, so the thread's pairs are base + j)
        const C wk = lds_ld<C>(sh, ((BPL + 2) * tf + j) * (int)sizeof(C)), wm = lds_ld<C>(sh, ((BPL + 2) * (TF + tf) + j) * (int)sizeof(C));
        ab[j] = mk<T>(wk.x * pk, wk.y * pk);
        ab[BPL + j] = mk<T>(wm.x * pm, wm.y * pm);
    }
    // running sums: acc = keep * acc + ab (keep = 0 restarts the sum at the first bin of a pair segment / of the run)
    LRA_UNROLL
    for (int run = 0; run < 2; ++run) {
        C acc = mk<T>((T)0, (T)0);
        LRA_UNROLL
        for (int j = 0; j < BPL; ++j) {
            const int jj = run * BPL + j;
            const T keep = rg.keep[jj];
            acc = mk<T>(acc.x * keep + ab[jj].x, acc.y * keep + ab[jj].y);
            lds_st<C>(rs, (jj * TF + tf) * (int)sizeof(C), acc);
        }
    }
    if (tf == 0) {
        const C xmid = mk<T>((T)2 * rg.mid.x, (T)-2 * rg.mid.y);  // X[M/2] = conj(Z[M/2]), Z pre-halved
        const T pmid = spec_power<T, PM>(xmid, a.power);
        const C wq = lds_ld<C>(sh, melr_w_slot<Cfg>(M) * (int)sizeof(C));
        lds_st<C>(rs, a.melr_mid, mk<T>(wq.x * pmid, wq.y * pmid));
        lds_st<C>(rs, a.melr_zero, mk<T>((T)0, (T)0));
    }
}

#ifndef LRA_MEL_PAIR_READS
#define LRA_MEL_PAIR_READS 1
#endif
// prologue: per-thread constants of the run-ordered epilogue -> registers (restart factors; the first MELR_PHOIST
// entries of both piece lists of mel bands tf and tf + TF)
// `base`: byte offset added to every hoisted address (the second-generation kernel passes its slot's offset, so that the 16
// piece reads per frame need no address arithmetic at all)
template <class Cfg, class RG> LRA_HD void melr_hoist(const StftArgs<typename Cfg::real>& a, int tf, RG& rg, int base = 0) {
    constexpr int PH = RG::MELR_PHOIST;
    LRA_UNROLL
    for (int jj = 0; jj < Cfg::R; ++jj) rg.keep[jj] = a.melr_keep[jj * Cfg::TF + tf];
    LRA_UNROLL
    for (int b = 0; b < RG::MELR_NB; ++b) {
        const int m = RG::MELR_NB * tf + b;  // bands NB tf .. NB tf + NB - 1 (see melr_combine)
        LRA_UNROLL
        for (int h = 0; h < 2; ++h) {
            LRA_UNROLL
            for (int q = 0; q < PH; ++q) {
                const int ad = base + (m < a.n_mels ? a.melr_addr[(h * a.melr_pmax + q) * a.n_mels + m] : a.melr_zero);
                // (the table addresses the B half of a pair for list 0; melr_combine reads whole pairs: keep the pair's address)
                rg.mad[b][h * PH + q] = (LRA_MEL_PAIR_READS && sizeof(typename Cfg::real) == 4) ? (ad & ~(2 * (int)sizeof(typename Cfg::real) - 1)) : ad;
            }
        }
    }
}

// ---- the register tile of the run-ordered mel epilogue: slot of a frame, and the burst that stores a band's tile ------------------
// The tile position is tied to the ABSOLUTE element index in the output (row starts are n_frames elements apart, not a multiple of
// MT), so that every full burst is one aligned MT x sizeof(T) = 32-byte piece: unaligned 32-byte bursts each dirtied two HBM sectors
// (306 MB written per launch for 169 MB of output).  A slot's first and last bursts are partial.
#ifndef LRA_MEL_ALIGNED_BURSTS
#define LRA_MEL_ALIGNED_BURSTS 1
#endif
#ifndef LRA_MEL_BURST16  // a whole tile as two 16-byte stores instead of eight predicated 4-byte ones
#define LRA_MEL_BURST16 1
#endif
template <class Cfg, class RG> LRA_HD int melr_tile_slot(const StftArgs<typename Cfg::real>& a, int clip, int frame, int it, int b, long long row0) {
    constexpr int MT = RG::MELR_TILE;
    if (!LRA_MEL_ALIGNED_BURSTS) return it & (MT - 1);
    // rows start 0 or 16 bytes into a 32-byte piece, alternating with the band's parity: with bands 2 tf and 2 tf + 1
    // per thread the phase is (clip, frame, b) only -- scalar, so the tile select and the flush branch stay scalar
    // (a per-lane phase made the flush run twice per MT frames under half masks: +2.8 % kernel time)
    if ((a.n_frames & 3) == 0 && Cfg::TF >= 64) return (int)(((long long)clip * a.n_mels * a.n_frames + (long long)b * a.n_frames + frame) & (MT - 1));
    return (int)((row0 + frame) & (MT - 1));
}
// stores slots max(0, s8 - it) .. s8 of band slot b's tile (the frames this slot has produced so far); a whole tile leaves as two 16-byte stores
template <class Cfg, class RG> LRA_HD void melr_burst(const StftArgs<typename Cfg::real>& a, long long row0, int frame, int s8, int it, int b, RG& rg) {
    using T = typename Cfg::real;
    constexpr int MT = RG::MELR_TILE;
    T* __restrict__ row = a.Mel + (row0 + frame - s8);
    if constexpr (sizeof(T) == 4 && (MT == 8 || MT == 4)) {
        if (LRA_MEL_BURST16 && s8 == MT - 1 && it >= MT - 1) {
            store4_unaligned(row, rg.mt[b][0], rg.mt[b][1], rg.mt[b][2], rg.mt[b][3]);
            if constexpr (MT == 8) store4_unaligned(row + 4, rg.mt[b][4], rg.mt[b][5], rg.mt[b][6], rg.mt[b][7]);
            return;
        }
    }
    LRA_UNROLL
    for (int k = 0; k < MT; ++k)
        if (k <= s8 && k >= s8 - it) row[k] = rg.mt[b][k];
}

// phase: mel[m] = sum of the B totals of segment m's pieces + sum of the A totals of segment m+1's pieces
// (ascending bins).  Bands tf and tf + TF use the hoisted address lists; longer lists / further bands read
// theirs from the shared table.
// rs_hoisted: the region the hoisted addresses (rg.mad) are relative to (see melr_hoist's `base`)
template <class Cfg, class RG> LRA_HD void melr_combine(const StftArgs<typename Cfg::real>& a, int clip, int frame, int tf, int it, int tile, bool last_of_slot, RG& rg, Lds sh,
                                                        Lds rs, Lds stage, Lds rs_hoisted) {
    using T = typename Cfg::real;
    constexpr int PH = RG::MELR_PHOIST, TF = Cfg::TF, NB = RG::MELR_NB;
    const bool more = a.melr_pmax > PH;  // uniform
    // The hoisted piece totals are read as whole (A, B) pairs and the wanted half is picked in registers (list h = 0: the B totals of
    // segment m, h = 1: the A totals of segment m + 1): an 8-byte read spreads 32 lanes over 64 banks, a 4-byte read of one half of
    // 8-byte slots over 16 -- a built-in 2-way conflict that was a third of the epilogue's LDS cycles (scripts/lds_model.py).
    // (sixteen reads in flight at a time: the bands are taken in groups of GB, so that eight bands per thread do not keep 32 totals live at once)
    constexpr int GB = NB * 2 * PH > 16 ? 16 / (2 * PH) : NB;
    T x[NB][2 * PH];
    LRA_UNROLL
    for (int b = 0; b < NB; ++b) {
      if (b % GB == 0) {
        LRA_UNROLL
        for (int bb = b; bb < b + GB; ++bb) {
            LRA_UNROLL
            for (int q = 0; q < 2 * PH; ++q) {
                if constexpr (LRA_MEL_PAIR_READS && sizeof(T) == 4) {
                    const cx<T> pr = lds_ld<cx<T>>(rs_hoisted, rg.mad[bb][q]);  // (melr_hoist keeps the pair's address)
                    x[bb][q] = q < PH ? pr.y : pr.x;
                } else {
                    x[bb][q] = lds_ld<T>(rs_hoisted, rg.mad[bb][q]);
                }
            }
        }
      }
        // Bands NB tf .. NB tf + NB - 1.  (NB = 2: a thread's first band is even and its second odd for EVERY lane, which makes the burst phase
        // below one value per wave and band slot wherever rows start 0 or 16 bytes into a 32-byte piece -- n_frames a multiple of 4.)
        const int m = NB * tf + b;
        if (m >= a.n_mels) break;
        T part[2];
        LRA_UNROLL
        for (int h = 0; h < 2; ++h) {
            T acc = (T)0;
            LRA_UNROLL
            for (int q = 0; q < PH; ++q) acc += x[b][h * PH + q];
            if (LRA_UNLIKELY(more)) {
                for (int q = PH; q < a.melr_pmax; ++q)
                    acc += lds_ld<T>(rs, lds_ld<int>(sh, melr_addr_off<Cfg>() + ((h * a.melr_pmax + q) * a.n_mels + m) * (int)sizeof(int)));
            }
            part[h] = acc;
        }
        const T v = part[0] + part[1];
        if (tile == 1) {
            // Register tile: the last MT frames of this band's row, stored as one burst (melr_tile_slot, melr_burst)
            constexpr int MT = RG::MELR_TILE;
            const long long row0 = ((long long)clip * a.n_mels + m) * a.n_frames;  // element index of this band's row in the output
            const int s8 = melr_tile_slot<Cfg, RG>(a, clip, frame, it, b, row0);
            LRA_UNROLL
            for (int k = 0; k < MT; ++k) rg.mt[b][k] = k == s8 ? v : rg.mt[b][k];  // (a select per slot: a conditional store here ends up as an indexed store to scratch memory)
            // (issued one frame later, behind the next frame's sample loads and a whole frame ahead of the loop top's s_waitcnt vmcnt(0), the
            // bursts are 1-2 % SLOWER: profiles/r03_experiments.md section 11)
            if (last_of_slot || s8 == MT - 1) melr_burst<Cfg, RG>(a, row0, frame, s8, it, b, rg);
        } else {
            lds_st<T>(stage, (m * tile + (it % tile)) * (int)sizeof(T), v);
        }
    }
    for (int m = tf + NB * TF; m < a.n_mels; m += TF) {  // more than NB bands per thread: everything from the table
        T part[2];
        LRA_UNROLL
        for (int h = 0; h < 2; ++h) {
            T acc = (T)0;
            for (int q = 0; q < a.melr_pmax; ++q)
                acc += lds_ld<T>(rs, lds_ld<int>(sh, melr_addr_off<Cfg>() + ((h * a.melr_pmax + q) * a.n_mels + m) * (int)sizeof(int)));
            part[h] = acc;
        }
        const T v = part[0] + part[1];
        if (tile == 1) a.Mel[((long long)clip * a.n_mels + m) * a.n_frames + frame] = v;
        else lds_st<T>(stage, (m * tile + (it % tile)) * (int)sizeof(T), v);
    }
}

#define LRA_MID_PASS(Cfg, p, rg, lds, tw, slot_bytes) LRA_MID_PASS_TO(Cfg, Cfg::P, p, rg, lds, tw, slot_bytes)
#define LRA_MID_PASS_TO(Cfg, END, p, rg, lds, tw, slot_bytes)                                             \
    if (END > p) {                                                                                        \
        LRA_PHASE(Cfg::NT, tid) {                                                                         \
            pass_read<Cfg, (p < Cfg::P ? p : 0)>(LRA_R(rg).v, lds_sub(lds, (slot_of<Cfg>(tid)) * (slot_bytes)), lane_of<Cfg>(tid)); \
        } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)                                                              \
        LRA_TICK(2 * p);                                                                                  \
        LRA_PHASE(Cfg::NT, tid) {                                                                         \
            pass_dft<Cfg, (p < Cfg::P ? p : 0)>(LRA_R(rg), lane_of<Cfg>(tid), tw);                          \
            pass_write<Cfg, (p < Cfg::P ? p : 0)>(LRA_R(rg).v, lds_sub(lds, (slot_of<Cfg>(tid)) * (slot_bytes)), lane_of<Cfg>(tid)); \
        } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)                                                              \
        LRA_TICK(2 * p + 1);                                                                              \
    }

// bytes of LDS one frame slot needs (frame area + mel staging tile), multiple of 16
// slot layout: [frame area | PCM ring of N reals | mel staging tile n_mels x tile]
template <class Cfg> constexpr int stft_ring_off() { return Cfg::FRAME_BYTES; }
template <class Cfg> constexpr int stft_tile_off() { return Cfg::FRAME_BYTES + Cfg::N * (int)sizeof(typename Cfg::real); }
template <class Cfg> inline int stft_slot_bytes(int mode, int n_mels, int tile) {
    int b = stft_tile_off<Cfg>();
    if (mode == OUT_MEL || ((mode == OUT_MEL2 || mode == OUT_MELR) && tile > 1)) b += ((n_mels * tile * (int)sizeof(typename Cfg::real) + 15) / 16) * 16;
    if (mode == OUT_MEL2) b += mel2_psum_bytes<Cfg>(n_mels);
    return b;
}

// One workgroup = FPB frame slots; slot s transforms the `iters` CONSECUTIVE frames
// f_first + s*iters + it of clip blk / wg_per_clip, as a private pipeline: all LDS traffic of a slot
// stays inside the slot (and, when TF <= 64, inside one wave: no s_barrier anywhere).  The mel
// epilogue stages `mel_tile` frames per row before flushing them as contiguous runs.
// RAM: ring addressing mode: 0 general, 1 row-aligned (hop = n_fft/4), 2 no ring at all (direct framing, hop >= n_fft; complex / power epilogues)
template <class Cfg, int MODE, int PM = POW_TWO, int RAM = 0> LRA_HD void stft_block(const StftArgs<typename Cfg::real>& a_in, const int blk, Lds lds) {
    constexpr bool RA = RAM == 1, DIRECT = RAM >= 2;        // DIRECT: no ring in LDS
    constexpr int RHD = 1 << (RAM >= 3 ? RAM - 2 : 0);      // RAM = 3 .. 6: register ring with hop = n_fft / RHD
    static_assert(RAM <= 2 || RegRing<Cfg, RHD>::ok, "register ring: the hop must be a whole number of pass-0 rows");
    static_assert(!DIRECT || MODE == OUT_COMPLEX || MODE == OUT_POWER, "direct framing serves the complex / power epilogues");
    StftArgs<typename Cfg::real> a = a_in;
    const int clip = blk / a.wg_per_clip;
    const int f_first = (blk % a.wg_per_clip) * a.frames_per_wg;
    const int iters = a.frames_per_wg / Cfg::FPB;
    const int slot_bytes = a.slot_bytes;
    const int tile = a.mel_tile;
    // OUT_MEL2 defers the last two steps of frame t's mel epilogue into frame t+1's phases (their LDS
    // regions -- piece sums, staging tile -- are not touched by the FFT): combine(t) runs next to the ring
    // loads / pass 0 of frame t+1 and flush(t) next to its split reads, which saves two workgroup
    // barriers per frame and lets the loads of both overlap.  The last frame's pair runs after the loop.
    constexpr bool DEFER = MODE == OUT_MEL2 && !(LRA_ABLATE >= 11 && LRA_ABLATE <= 13);
    // The mel kernel has no spectrum stores to keep clear of (see the ring notes above), so it fetches the
    // next frame's samples late -- issued with the split reads, consumed after the piece sums -- which
    // keeps the NPF prefetch registers out of the FFT passes, where the register pressure peaks.
#ifndef LRA_MEL_LATE_PF
#define LRA_MEL_LATE_PF 1
#endif
    constexpr bool LATE_PF = DEFER && LRA_MEL_LATE_PF;
    // Vector memory operations of a wave complete IN ORDER (one vmcnt for loads and stores): a wait for the prefetched
    // samples also waits for every store issued before those loads, and a spectrum store is acknowledged only ~1.5 us
    // after issue (a lone wave loses 1 400 of 5 200 cycles per frame to that wait).  So the loads of frame t+2 are issued
    // at the END of frame t, right after the ring took frame t+1's samples and BEFORE frame t's stores: the wait one
    // frame later is then s_waitcnt vmcnt(<stores of frame t+1>) and only covers stores that are two frames old.
#ifndef LRA_PF_EARLY
#define LRA_PF_EARLY 1
#endif
    constexpr bool PF_EARLY = !LATE_PF && LRA_PF_EARLY;
    // Complex / power epilogues: the split step works on registers and writes HBM only, so the frame needs ONE workgroup barrier
    // behind its split reads ("every wave has read the frame area"), and that one sits behind the stores: a wave that waits there
    // has its row already in flight.  (Frames of more than one wave only; 7 -> 6 barriers per frame at n_fft = 8192.)
#ifndef LRA_SPLIT_ONE_BARRIER
#define LRA_SPLIT_ONE_BARRIER 1
#endif
    constexpr bool SPLIT_NO_LDS = LRA_SPLIT_ONE_BARRIER && (MODE == OUT_COMPLEX || MODE == OUT_POWER);
    // ... and where the last pass is one radix-16 butterfly per thread (n_fft = 8192), the split step's LDS round trip goes as well
#ifndef LRA_MIRROR32
#define LRA_MIRROR32 1
#endif
    // (measured per ring form on the 256 x 30 s batch, n_fft = 8192, profiles/r03_experiments.md section 8: hop = n_fft / 16 -1 %, n_fft / 2 -9 %;
    // n_fft / 4 unchanged, n_fft / 8 and direct framing +3 ... +6 % -- those keep the split through LDS)
    constexpr bool MIRROR = LRA_MIRROR32 && (RAM == 3 || RAM == 6 || LRA_MIRROR32 > 1) && mirror32_cfg_ok<Cfg>();
    constexpr int MID_END = Cfg::P - (MIRROR ? 1 : 0);  // passes 1 .. MID_END - 1 go through LRA_MID_PASS
    LRA_REGS(FftRegs<Cfg>, rg, Cfg::NT);
    LRA_PHASE(Cfg::NT, tid) {
        hoist_tables<Cfg>(LRA_R(rg), lane_of<Cfg>(tid), a.win, a.tw, a.twr, false, MODE == OUT_MELR);
        if (MIRROR) mirror32_hoist<Cfg>(LRA_R(rg), lane_of<Cfg>(tid), a.tw, a.twr);
        if (MODE == OUT_MEL2) mel2_tables_to_lds<Cfg>(a, tid, lds_sub(lds, a.shared_off));
        if (MODE == OUT_MELR) {
            melr_tables_to_lds<Cfg>(a, tid, lds_sub(lds, a.shared_off));
            melr_hoist<Cfg>(a, lane_of<Cfg>(tid), LRA_R(rg));
        }
    } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC && MODE != OUT_MEL2 && MODE != OUT_MELR)  // the shared tables need a workgroup barrier, once
    LRA_PHASE(Cfg::NT, tid) {
        const int slot = slot_of<Cfg>(tid), tf = lane_of<Cfg>(tid);
        if (RHD > 1 || DIRECT) {
            // prefetch registers of a slot whose frames lie past the clip are never loaded (the fetches return early): defined zeros, once,
            // outside the frame loop -- such frames are transformed like any other and nothing of them is stored
            LRA_UNROLL
            for (int e = 0; e < Cfg::R; ++e) LRA_R(rg).nxt[e] = mk<typename Cfg::real>((typename Cfg::real)0, (typename Cfg::real)0);
        }
        if (RHD > 1) {
            regring_fill<Cfg, RHD>(a, clip, f_first + slot * iters, tf, LRA_R(rg));
        } else if (DIRECT) {
            stft_direct_fetch<Cfg>(a, clip, f_first + slot * iters, tf, LRA_R(rg));
        } else {
            stft_ring_fill<Cfg>(a, clip, f_first + slot * iters, tf, lds_sub(lds, slot * slot_bytes + stft_ring_off<Cfg>()));
            if (PF_EARLY && iters > 1) stft_ring_prefetch<Cfg, RA>(a, clip, f_first + slot * iters + 1, tf, LRA_R(rg));
        }
    } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
    int done = 0;  // frames of this workgroup's slots processed so far (uniform)
    LRA_TICK_DECL;
    for (int it = 0; it < iters; ++it) {
        if (f_first + it >= a.n_frames) break;  // slot 0 has the smallest frame index: uniform exit
        LRA_TICK(0);
        if (!Cfg::HOIST) { LRA_LAUNDER(a.win); LRA_LAUNDER(a.tw); LRA_LAUNDER(a.twr); }
        if (MODE == OUT_MELR) lra_setprio<LRA_V1_PRIO_A>();  // wave priority per phase, as in stft2_kernel (lra_kernels2.h): transform / split / epilogue / band combine
        else if (MODE == OUT_COMPLEX || MODE == OUT_POWER) lra_setprio<LRA_V1_PRIO_CA>();  // (complex / power epilogues: transform, then un-split + stores)
        LRA_PHASE(Cfg::NT, tid) {
            const int slot = slot_of<Cfg>(tid), tf = lane_of<Cfg>(tid), frame = f_first + slot * iters + it;
            const Lds sl = lds_sub(lds, slot * slot_bytes);
#if LRA_ABLATE != 3 && LRA_ABLATE != 4  // experiments 3 / 4: no PCM loads in the frame loop (4: and no spectrum stores)
            if (!DIRECT && !LATE_PF && !PF_EARLY && it + 1 < iters) stft_ring_prefetch<Cfg, RA>(a, clip, frame + 1, tf, LRA_R(rg));
#endif
            if (DEFER && it > 0 && frame - 1 < a.n_frames)
                mel2_combine<Cfg>(a, clip, frame - 1, tf, (it - 1) % tile, tile, lds_sub(lds, a.shared_off), lds_sub(sl, slot_bytes - mel2_psum_bytes<Cfg>(a.n_mels)), lds_sub(sl, stft_tile_off<Cfg>()));
            if (RHD > 1) regring_pass0<Cfg, RHD>(a, clip, frame, it == 0, it + 1 < iters, tf, LRA_R(rg), sl);
            else if (DIRECT) stft_direct_pass0<Cfg>(a, clip, frame, it + 1 < iters, tf, LRA_R(rg), sl);
            else stft_ring_load_pass0<Cfg, RA>(a, frame, tf, LRA_R(rg), lds_sub(sl, stft_ring_off<Cfg>()), sl);
        } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
        LRA_TICK(1);
#if LRA_ABLATE != 2
        LRA_MID_PASS_TO(Cfg, MID_END, 1, rg, lds, a.tw, slot_bytes)
        LRA_MID_PASS_TO(Cfg, MID_END, 2, rg, lds, a.tw, slot_bytes)
        LRA_MID_PASS_TO(Cfg, MID_END, 3, rg, lds, a.tw, slot_bytes)
#endif
        if constexpr (MIRROR) {
            // last pass (butterfly mirror32_bfly), exchange across the wave's halves, split + stores from the registers; the ONE
            // workgroup barrier ("every wave has read the frame area") sits behind the stores
            LRA_PHASE(Cfg::NT, tid) {
                const int tf = lane_of<Cfg>(tid);
                pass_read<Cfg, Cfg::P - 1>(LRA_R(rg).v, lds_sub(lds, slot_of<Cfg>(tid) * slot_bytes), mirror32_bfly<Cfg>(tf));
                pass_twiddle_dft_reg<Cfg, Cfg::P - 1>(LRA_R(rg).v, LRA_R(rg).treg);
            } LRA_PHASE_END_SYNC(true)
            LRA_PHASE(Cfg::NT, tid) {
                LRA_HALF_SWAP8(rg, tid, Cfg::R / 2, mirror32_swaps<Cfg>(lane_of<Cfg>(tid)));
            } LRA_PHASE_END_SYNC(true)
            if (MODE == OUT_COMPLEX || MODE == OUT_POWER) lra_setprio<LRA_V1_PRIO_CS>();
            LRA_PHASE(Cfg::NT, tid) {
                const int slot = slot_of<Cfg>(tid), tf = lane_of<Cfg>(tid), frame = f_first + slot * iters + it;
                mirror32_split_store<Cfg, MODE, PM>(a, clip, frame, frame < a.n_frames, tf, LRA_R(rg));
            } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
            LRA_TICK(10);
            done = it + 1;
            continue;
        }
        LRA_PHASE(Cfg::NT, tid) {
            const int slot = slot_of<Cfg>(tid), tf = lane_of<Cfg>(tid), frame = f_first + slot * iters + it;
            const Lds sl = lds_sub(lds, slot * slot_bytes);
            if (LATE_PF && it + 1 < iters) stft_ring_prefetch<Cfg, RA>(a, clip, frame + 1, tf, LRA_R(rg));
            if constexpr (MODE == OUT_MELR) split_read_runs<Cfg>(LRA_R(rg), sl, tf);
            else split_read<Cfg>(LRA_R(rg), sl, tf);
#if LRA_ABLATE != 3 && LRA_ABLATE != 4
            if (!DIRECT && !LATE_PF && it + 1 < iters) stft_ring_advance<Cfg, RA>(a, clip, frame + 1, tf, LRA_R(rg), lds_sub(sl, stft_ring_off<Cfg>()));
            if (!DIRECT && PF_EARLY && it + 2 < iters) stft_ring_prefetch<Cfg, RA>(a, clip, frame + 2, tf, LRA_R(rg));
#endif
            if (DEFER && tile > 1 && it > 0 && it % tile == 0)  // the tile that frame it-1 completed
                mel_flush_tile<Cfg>(a, clip, f_first + slot * iters, it - 1, tile, tf, lds_sub(sl, stft_tile_off<Cfg>()));
        } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC || SPLIT_NO_LDS)
        LRA_TICK(8);
        if (MODE == OUT_MELR) lra_setprio<LRA_V1_PRIO_B>();
        else if (MODE == OUT_COMPLEX || MODE == OUT_POWER) lra_setprio<LRA_V1_PRIO_CS>();
        LRA_PHASE(Cfg::NT, tid) {
            const int slot = slot_of<Cfg>(tid), tf = lane_of<Cfg>(tid), frame = f_first + slot * iters + it;
            if constexpr (MODE == OUT_MELR) melr_split_accumulate<Cfg, PM>(a, frame < a.n_frames, tf, LRA_R(rg), lds_sub(lds, slot * slot_bytes), lds_sub(lds, a.shared_off));
            else stft_split_store<Cfg, MODE, PM>(a, clip, frame, frame < a.n_frames, tf, LRA_R(rg), lds_sub(lds, slot * slot_bytes), lds_sub(lds, a.shared_off));
        } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
        LRA_TICK(9);
        if (MODE == OUT_MEL2 && LRA_ABLATE != 12 && LRA_ABLATE != 13) {
            LRA_PHASE(Cfg::NT, tid) {
                const int slot = slot_of<Cfg>(tid), tf = lane_of<Cfg>(tid), frame = f_first + slot * iters + it;
                const Lds sl = lds_sub(lds, slot * slot_bytes);
                if (frame < a.n_frames) mel2_gather<Cfg>(a, tf, sl, lds_sub(lds, a.shared_off), lds_sub(sl, slot_bytes - mel2_psum_bytes<Cfg>(a.n_mels)));
                if (LATE_PF && it + 1 < iters) stft_ring_advance<Cfg, RA>(a, clip, frame + 1, tf, LRA_R(rg), lds_sub(sl, stft_ring_off<Cfg>()));
            } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
        }
        if (MODE == OUT_MEL || MODE == OUT_MELR) {
            if (MODE == OUT_MELR) lra_setprio<LRA_V1_PRIO_B3>();
            LRA_PHASE(Cfg::NT, tid) {
                const int slot = slot_of<Cfg>(tid), tf = lane_of<Cfg>(tid), frame = f_first + slot * iters + it;
                const Lds sl = lds_sub(lds, slot * slot_bytes);
                if (frame < a.n_frames) {
                    if constexpr (MODE == OUT_MELR) melr_combine<Cfg>(a, clip, frame, tf, it, tile, it + 1 == iters || frame + 1 >= a.n_frames, LRA_R(rg), lds_sub(lds, a.shared_off), sl, lds_sub(sl, stft_tile_off<Cfg>()), sl);
                    else mel_reduce_slot<Cfg>(a, tf, it % tile, tile, sl, lds_sub(sl, stft_tile_off<Cfg>()));
                }
            } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
            if (!(MODE == OUT_MELR && tile == 1) && ((it + 1) % tile == 0 || it + 1 == iters || f_first + it + 1 >= a.n_frames)) {
                LRA_PHASE(Cfg::NT, tid) {
                    const int slot = slot_of<Cfg>(tid), tf = lane_of<Cfg>(tid);
                    mel_flush_tile<Cfg>(a, clip, f_first + slot * iters, it, tile, tf, lds_sub(lds, slot * slot_bytes + stft_tile_off<Cfg>()));
                } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
            }
        }
        LRA_TICK(10);
        done = it + 1;
    }
    LRA_TICK_FLUSH;
    if (DEFER && done > 0) {  // epilogue of the last frame
        const int it = done - 1;
        LRA_PHASE(Cfg::NT, tid) {
            const int slot = slot_of<Cfg>(tid), tf = lane_of<Cfg>(tid), frame = f_first + slot * iters + it;
            const Lds sl = lds_sub(lds, slot * slot_bytes);
            if (frame < a.n_frames)
                mel2_combine<Cfg>(a, clip, frame, tf, it % tile, tile, lds_sub(lds, a.shared_off), lds_sub(sl, slot_bytes - mel2_psum_bytes<Cfg>(a.n_mels)), lds_sub(sl, stft_tile_off<Cfg>()));
        } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
        if (tile > 1) {
            LRA_PHASE(Cfg::NT, tid) {
                const int slot = slot_of<Cfg>(tid), tf = lane_of<Cfg>(tid);
                mel_flush_tile<Cfg>(a, clip, f_first + slot * iters, it, tile, tf, lds_sub(lds, slot * slot_bytes + stft_tile_off<Cfg>()));
            } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
        }
    }
}

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
    const T* wss;               // [out_len] NORMALISATION FACTORS of the output samples: 1 / wss[s] where the window sum-square (already sliced / fixed to the
                                // output) exceeds `tiny`, else 1 (core/spectrum.py:606-624 as a multiplication; the host wrapper builds the table: lra_api.hip)
    T tiny;
    T* y;                       // [batch][out_len]; the kernel writes every sample below istft_written_end(), the host wrapper zeroes the rest
    long long y_stride;
    long long out_len;
    long long batch;
    int strip_frames;           // frames whose hop-blocks one slot finalises
    int strips_per_clip;
    int warm_frames;            // frames replayed before the strip: ceil(N/hop) - 1
    int drain_steps;            // extra steps the last strip of a clip runs to flush the carry
    int n_blocks, xcd_chunk;    // launch geometry, see xcd_block
};

// First output sample of a clip that NO step of the fused kernel stores: the strips of a clip walk frames 0 .. n_used - 1 and the last
// strip runs drain_steps more, each step finalising one hop block of padded positions [t hop, (t + 1) hop) -- gaps between frames
// (hop > n_fft) included, they leave as the zero carry -- so samples [0, min(out_len, end)) are all written and the host wrapper
// only zeroes [end, out_len): the part of `length` beyond the frames' reach (core/spectrum.py:553-555, 606-624), normally nothing.
LRA_HD long long istft_written_end(int n_fft, int hop, long long n_used, int drop) {
    const long long drain = n_fft > hop ? (n_fft - hop + hop - 1) / hop : 0;
    const long long end = (n_used + drain) * (long long)hop - drop;
    return end < 0 ? 0 : end;
}

// per slot: frame area + double-buffered carry of N reals (>= N - hop for any hop >= 1)
// HC > 0 (row-aligned): one in-place carry of the R - HC rows that outlive a frame, (R - HC) TF sample pairs
// Row-aligned hops shift a thread's carry pairs onto ITS OWN pairs (istft_last_ola_rows), so the carry is thread-private: where the
// register budget allows (the ascending-radix float configuration: n_fft = 2048) it lives in R - HC register pairs instead of LDS --
// 2 (R - HC) LDS instructions per frame and thread less (24 of 54 at hop = n_fft / 4), for no instruction more: the frame's sums ARE
// the next frame's carry, one hop of rows down.
#ifndef LRA_ISTFT_REG_CARRY
#define LRA_ISTFT_REG_CARRY 1
#endif
template <class Cfg, int HC> constexpr bool istft_reg_carry() { return LRA_ISTFT_REG_CARRY && HC > 0 && Cfg::REV && Cfg::HOIST && sizeof(typename Cfg::real) == 4 && Cfg::R == 16; }
template <class Cfg, int HC = 0> constexpr int istft_carry_reals() { return istft_reg_carry<Cfg, HC>() ? 0 : (HC > 0 ? (Cfg::R - HC) * Cfg::TF * 2 : 2 * Cfg::N); }
template <class Cfg, int HC = 0> constexpr int istft_slot_bytes() { return Cfg::FRAME_BYTES + ((istft_carry_reals<Cfg, HC>() * (int)sizeof(typename Cfg::real) + 15) / 16) * 16; }
template <class Cfg, int HC = 0> constexpr int istft_lds_bytes() { return Cfg::FPB * istft_slot_bytes<Cfg, HC>(); }

// ---- spectrum prefetch: X[k], X[M-k] (k = tf + i TF) and X[M/2] of one frame -> registers ------------
template <class Cfg> LRA_HD void istft_spec_load(const IstftArgs<typename Cfg::real>& a, long long clip, int frame, bool valid, int tf, FftRegs<Cfg>& rg) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int M = Cfg::M;
    const C zero = mk<T>((T)0, (T)0);
    if (!valid) {
        LRA_UNROLL
        for (int i = 0; i < Cfg::R / 2; ++i) { rg.xk[i] = zero; rg.xm[i] = zero; }
        rg.xmid = zero;
        return;
    }
    const C* __restrict__ X = a.D + clip * a.d_batch_stride + (long long)frame * a.d_frame_stride;
    const C* __restrict__ Xk = X + tf;        // ascending bins: one base, immediate offsets
    const C* __restrict__ Xm = X + (M - tf);  // mirrored bins descend from the other
    LRA_UNROLL
    for (int i = 0; i < Cfg::R / 2; ++i) {
        rg.xk[i] = Xk[i * Cfg::TF];
        rg.xm[i] = Xm[-i * Cfg::TF];  // tf = 0, i = 0: X[M], the Nyquist bin
    }
    rg.xmid = X[M / 2];
}

// ---- phase: Hermitian split of X[0..M] into conj(Z'[0..M-1]) in LDS ---------------------------
// Z'[k] = E' + i O',  E' = X[k] + conj(X[M-k]),  O' = (X[k] - conj(X[M-k])) conj(W_N^k); the
// imaginary parts of X[0] and X[M] are ignored, as pocketfft's c2r does.
template <class Cfg> LRA_HD void istft_split_write(const IstftArgs<typename Cfg::real>& a, int tf, const FftRegs<Cfg>& rg, Lds fr) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int M = Cfg::M;
    LRA_UNROLL
    for (int i = 0; i < Cfg::R / 2; ++i) {
        const int k = tf + i * Cfg::TF;
        const C xk = rg.xk[i], xm = rg.xm[i];
        if (k == 0) {
            lds_st<C>(fr, Cfg::phys(0) * (int)sizeof(C), mk<T>(xk.x + xm.x, -(xk.x - xm.x)));
        } else {
            const C E = add_conj(xk, xm);
            const C O = cmul2_conj(sub_conj(xk, xm), Cfg::HOIST ? rg.twr[i] : a.twr[k]);
            // Z'[k] = (E.x - O.y, E.y + O.x);  Z'[M-k] = (E.x + O.y, O.x - E.y); store conjugates
            lds_st<C>(fr, Cfg::phys(k) * (int)sizeof(C), conj_add_mi_neg(E, O));
            lds_st<C>(fr, Cfg::phys(M - k) * (int)sizeof(C), add_mi(E, O));
        }
    }
    // (consumed by every lane, not only tf = 0: a load left pending behind a skipped branch costs a vmcnt(0) later)
    C mid2 = mk<T>((T)2 * rg.xmid.x, (T)2 * rg.xmid.y);  // conj(2 conj X[M/2])
    LRA_KEEP(mid2.x);
    LRA_KEEP(mid2.y);
    if (tf == 0) lds_st<C>(fr, Cfg::phys(M / 2) * (int)sizeof(C), mid2);
}

// ---- ascending-radix configurations (FftCfg REV): Hermitian step fused into the first pass --------------------------------
// The first pass (radix r0, butterfly b reads Z'[b + j s], s = M / r0) then has two butterflies per thread, and as in the
// second-generation forward kernel (lra_kernels2.h) thread tf takes butterflies tf and s - tf (thread 0: 0 and s/2): both
// members of every pair (Z'[k], Z'[M-k]) are inputs of ONE thread.  X[k], X[M-k] are loaded in that order (pair slot q:
// k = tf + q s), turned into the pair in registers and fed straight to the butterflies: the split phase's LDS write and the
// first pass's LDS read (32 of the frame's 80 LDS instructions) disappear.
template <class Cfg> constexpr bool istft_mir_ok() {
    return Cfg::REV && Cfg::HOIST && sizeof(typename Cfg::real) == 4 && Cfg::R == 16 && Cfg::P >= 2 && (Cfg::R >> Cfg::logr(0)) == 2 && 2 * Cfg::TF == (Cfg::M >> Cfg::logr(0)) &&
           affine_tf<Cfg>();
}
template <class Cfg> LRA_HD int mir_bfly(int tf) { return tf == 0 ? Cfg::TF : 2 * Cfg::TF - tf; }
// bins of pair slots q >= r0/2 are tf + q s, except for lane 0 (whose second butterfly is s/2): s/2 + (q - r0/2) s
template <class Cfg> LRA_HD int mir_tf_hi(int tf) {
    constexpr int r0 = 1 << Cfg::logr(0), s = 2 * Cfg::TF;
    return tf == 0 ? (s / 2) * (1 - r0) : tf;
}
template <class Cfg> LRA_HD int mir_slot_bin(int tf, int q) {
    constexpr int r0 = 1 << Cfg::logr(0), s = 2 * Cfg::TF;
    return (q < r0 / 2 ? tf : mir_tf_hi<Cfg>(tf)) + q * s;
}

template <class Cfg> LRA_HD void istft_spec_load_mir(const IstftArgs<typename Cfg::real>& a, long long clip, int frame, bool valid, int tf, FftRegs<Cfg>& rg) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int M = Cfg::M, r0 = 1 << Cfg::logr(0), s = 2 * Cfg::TF;
    static_assert(r0 == Cfg::R / 2, "one pair slot per first-pass input");
    const C zero = mk<T>((T)0, (T)0);
    if (!valid) {
        LRA_UNROLL
        for (int q = 0; q < r0; ++q) { rg.xk[q] = zero; rg.xm[q] = zero; }
        rg.xmid = zero;
        return;
    }
    const C* __restrict__ X = a.D + clip * a.d_batch_stride + (long long)frame * a.d_frame_stride;
    const C* __restrict__ Xlo = X + tf;                     // slots q < r0/2: ascending from tf, mirrored from M - tf
    const C* __restrict__ Xhi = X + mir_tf_hi<Cfg>(tf);
    LRA_UNROLL
    for (int q = 0; q < r0; ++q) {
        const C* __restrict__ Xq = q < r0 / 2 ? Xlo : Xhi;
        rg.xk[q] = stream_load(&Xq[q * s]);
        rg.xm[q] = stream_load(&X[M - (q < r0 / 2 ? tf : mir_tf_hi<Cfg>(tf)) - q * s]);
    }
    rg.xmid = X[M / 2];
}

// The two halves of istft_unsplit_pass0 below: the register-only part (pairs -> conj Z' in
// first-pass order, first-pass butterflies) and the first LDS write of the frame.
template <class Cfg> LRA_HD void istft_unsplit_pass0_arith(int tf, FftRegs<Cfg>& rg) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int r0 = 1 << Cfg::logr(0);
    const bool l0 = tf == 0;
    C zk[r0], zm[r0];
    LRA_UNROLL
    for (int q = 0; q < r0; ++q) {
        const C xk = rg.xk[q], xm = rg.xm[q];
        const C E = add_conj(xk, xm);
        const C O = cmul2_conj(sub_conj(xk, xm), rg.twr[q]);
        zk[q] = conj_add_mi_neg(E, O);
        zm[q] = add_mi(E, O);
    }
    const C z0 = mk<T>(rg.xk[0].x + rg.xm[0].x, -(rg.xk[0].x - rg.xm[0].x));
    const C mid2 = mk<T>((T)2 * rg.xmid.x, (T)2 * rg.xmid.y);
    C* A = rg.v;
    C* B = rg.v + r0;
    LRA_UNROLL
    for (int j = 0; j < r0; ++j) {
        C a0;
        if (j == 0) a0 = z0;
        else if (j < r0 / 2) a0 = zk[j];
        else if (j == r0 / 2) a0 = mid2;
        else a0 = zm[r0 - j];
        const C b0 = j < r0 / 2 ? zk[r0 / 2 + j] : zm[3 * r0 / 2 - 1 - j];
        A[j] = l0 ? a0 : zk[j];
        B[j] = l0 ? b0 : zm[r0 - 1 - j];
    }
    Dft<r0, T>::run(A);
    Dft<r0, T>::run(B);
}
template <class Cfg> LRA_HD void istft_unsplit_pass0_write(int tf, FftRegs<Cfg>& rg, Lds fr) {
    using C = typename Cfg::cplx;
    constexpr int r0 = 1 << Cfg::logr(0);
    const C* A = rg.v;
    const C* B = rg.v + r0;
    const int baseA = tf * (r0 + 1) * (int)sizeof(C), baseB = mir_bfly<Cfg>(tf) * (r0 + 1) * (int)sizeof(C);
    LRA_UNROLL
    for (int j = 0; j < r0; ++j) {
        lds_st<C>(fr, baseA + j * (int)sizeof(C), A[j]);
        lds_st<C>(fr, baseB + j * (int)sizeof(C), B[j]);
    }
}

// phase: pairs -> conj Z' in first-pass order, first-pass butterflies, first LDS write of the frame
template <class Cfg> LRA_HD void istft_unsplit_pass0(int tf, FftRegs<Cfg>& rg, Lds fr) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int r0 = 1 << Cfg::logr(0);
    const bool l0 = tf == 0;
    C zk[r0], zm[r0];
    LRA_UNROLL
    for (int q = 0; q < r0; ++q) {
        const C xk = rg.xk[q], xm = rg.xm[q];
        const C E = add_conj(xk, xm);
        const C O = cmul2_conj(sub_conj(xk, xm), rg.twr[q]);
        zk[q] = conj_add_mi_neg(E, O);  // conj Z'[k]
        zm[q] = add_mi(E, O);           // conj Z'[M - k]
    }
    // lane 0, slot 0: (X[0], X[M]) -> conj Z'[0] (their imaginary parts are ignored, as pocketfft's c2r does); bin M/2 -> conj Z'[M/2]
    const C z0 = mk<T>(rg.xk[0].x + rg.xm[0].x, -(rg.xk[0].x - rg.xm[0].x));
    const C mid2 = mk<T>((T)2 * rg.xmid.x, (T)2 * rg.xmid.y);
    C* A = rg.v;        // butterfly tf:      inputs Z'[tf + j s]        lanes 1..: zk[j];            lane 0: Z'[j s]
    C* B = rg.v + r0;   // butterfly s - tf:  inputs Z'[(s - tf) + j s]  lanes 1..: zm[r0 - 1 - j];   lane 0: Z'[s/2 + j s]
    LRA_UNROLL
    for (int j = 0; j < r0; ++j) {
        C a0;  // lane 0's A[j]
        if (j == 0) a0 = z0;
        else if (j < r0 / 2) a0 = zk[j];
        else if (j == r0 / 2) a0 = mid2;
        else a0 = zm[r0 - j];
        const C b0 = j < r0 / 2 ? zk[r0 / 2 + j] : zm[3 * r0 / 2 - 1 - j];  // lane 0's B[j]
        A[j] = l0 ? a0 : zk[j];
        B[j] = l0 ? b0 : zm[r0 - 1 - j];
    }
    Dft<r0, T>::run(A);
    Dft<r0, T>::run(B);
    // pass 0 (s = 1): output j of butterfly b at position b r0 + j, i.e. padded slot b (r0 + 1) + j
    const int baseA = tf * (r0 + 1) * (int)sizeof(C), baseB = mir_bfly<Cfg>(tf) * (r0 + 1) * (int)sizeof(C);
    LRA_UNROLL
    for (int j = 0; j < r0; ++j) {
        lds_st<C>(fr, baseA + j * (int)sizeof(C), A[j]);
        lds_st<C>(fr, baseB + j * (int)sizeof(C), B[j]);
    }
}

// ---- the same with FOUR first-pass butterflies per thread (FftCfg REV + PLAN 1: radices 4, 16, 16 at M = 1024; round 6) --------------------------
// The inverse of the forward kernels' third form (lra_kernels2.h): thread t takes first-pass butterflies 2t, 2t + 1 and their mirrors 255 - 2t,
// 256 - 2t (thread 0: 128), whose inputs Z'[b + 256 j] are the un-split of the bin pairs (k, M - k), k = 2t + 256 q (E slots) and 2t + 1 + 256 q
// (O slots): bins 2t, 2t + 1 and M - 2t - 1, M - 2t are NEIGHBOURS, so the spectrum row is read as 4 + 4 global_load_dwordx4 per thread (1 KiB per wave
// instruction) instead of 16 x dwordx2 -- the inverse stream gains 3.5 % from 16-byte pieces (profiles/r05_pitch.md 3b).  Thread 0's E slots 2 and 3 hold
// butterfly 128's pairs (bins 128 / 896 and 384 / 640): those four bins are read by every lane (one address per wave: a broadcast) and selected in.
// Pass 0 -> pass 1 hand-over: element i at slot i + (i >> 3) (a thread's eight outputs of E and O are contiguous: lanes nine slots apart, conflict-free).
template <class Cfg> constexpr bool istft_mir4_ok() {
    return Cfg::REV && Cfg::PLAN == 1 && Cfg::HOIST && sizeof(typename Cfg::real) == 4 && Cfg::R == 16 && Cfg::P == 3 && Cfg::logr(0) == 2 && Cfg::logr(1) == 4 && Cfg::TF == 64 &&
           affine_tf<Cfg>() && (Cfg::M + (Cfg::M >> 3)) * (int)sizeof(typename Cfg::cplx) <= Cfg::FRAME_BYTES;
}
// bin of pair slot q (q < 4: E slots, k = 2 tf + 256 q -- thread 0's slots 2, 3: 128, 384 --; q >= 4: O slots, k = 2 tf + 1 + 256 (q - 4))
template <class Cfg> LRA_HD int mir4_slot_bin(int tf, int q) {
    constexpr int s = 4 * Cfg::TF;
    if (q >= 4) return 2 * tf + 1 + (q - 4) * s;
    if (tf == 0 && q >= 2) return s / 2 + (q - 2) * s;
    return 2 * tf + q * s;
}
template <class C> LRA_HD void load_pair(const C* __restrict__ p, C& lo, C& hi) {
#if !defined(LRA_HOSTSIM)
    if constexpr (sizeof(C) == 8) {
        typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));
        const f4u v = *reinterpret_cast<const f4u*>(p);
        lo = __builtin_bit_cast(C, __builtin_shufflevector(v, v, 0, 1));
        hi = __builtin_bit_cast(C, __builtin_shufflevector(v, v, 2, 3));
        return;
    }
#endif
    lo = p[0];
    hi = p[1];
}
// xk[q] / xm[q]: q < 4 the E slots, q >= 4 the O slots (general addressing for every lane); nxt[0 .. 3] = X[128], X[896], X[384], X[640] (butterfly 128's bins)
template <class Cfg> LRA_HD void istft_spec_load_mir4(const IstftArgs<typename Cfg::real>& a, long long clip, int frame, bool valid, int tf, FftRegs<Cfg>& rg) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int M = Cfg::M, s = 4 * Cfg::TF;
    const C zero = mk<T>((T)0, (T)0);
    if (!valid) {
        LRA_UNROLL
        for (int q = 0; q < 8; ++q) { rg.xk[q] = zero; rg.xm[q] = zero; }
        LRA_UNROLL
        for (int q = 0; q < 4; ++q) rg.nxt[q] = zero;
        return;
    }
    const C* __restrict__ X = a.D + clip * a.d_batch_stride + (long long)frame * a.d_frame_stride;
    const C* __restrict__ pa = X + 2 * tf;
    const C* __restrict__ pm = X + (M - 1 - 2 * tf);
    LRA_UNROLL
    for (int q = 0; q < 4; ++q) {
        load_pair<C>(pa + q * s, rg.xk[q], rg.xk[4 + q]);   // X[2 tf + q s], X[2 tf + 1 + q s]
        load_pair<C>(pm - q * s, rg.xm[4 + q], rg.xm[q]);   // X[M - 2 tf - 1 - q s], X[M - 2 tf - q s]
    }
    rg.nxt[0] = X[s / 2];
    rg.nxt[1] = X[M - s / 2];
    rg.nxt[2] = X[s / 2 + s];
    rg.nxt[3] = X[M - s / 2 - s];
}
// phase: pairs -> conj Z' in first-pass order, the four first-pass butterflies, first LDS write of the frame
template <class Cfg> LRA_HD void istft_unsplit_pass0_mir4(int tf, FftRegs<Cfg>& rg, Lds fr) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    const bool l0 = tf == 0;
    const LaneMask l0m = lane_mask(l0);
    // thread 0: bin M/2 came in as the E half of ascending piece 2; its E slots 2 and 3 take butterfly 128's pairs
    const C mid2 = mk<T>((T)2 * rg.xk[2].x, (T)2 * rg.xk[2].y);
    C xk[8], xm[8];
    LRA_UNROLL
    for (int q = 0; q < 8; ++q) { xk[q] = rg.xk[q]; xm[q] = rg.xm[q]; }
    xk[2] = sel_mask(l0m, l0, rg.nxt[0], xk[2]);
    xm[2] = sel_mask(l0m, l0, rg.nxt[1], xm[2]);
    xk[3] = sel_mask(l0m, l0, rg.nxt[2], xk[3]);
    xm[3] = sel_mask(l0m, l0, rg.nxt[3], xm[3]);
    C zk[8], zm[8];
    LRA_UNROLL
    for (int q = 0; q < 8; ++q) {
        const C E = add_conj(xk[q], xm[q]);
        const C O = cmul2_conj(sub_conj(xk[q], xm[q]), rg.twr[q]);
        zk[q] = conj_add_mi_neg(E, O);  // conj Z'[k]
        zm[q] = add_mi(E, O);           // conj Z'[M - k]
    }
    // thread 0, slot 0: (X[0], X[M]) -> conj Z'[0] (their imaginary parts are ignored, as pocketfft's c2r does)
    const C z0 = mk<T>(xk[0].x + xm[0].x, -(xk[0].x - xm[0].x));
    C* E = rg.v;        // butterfly 2 tf:        inputs Z'[2 tf + j s] = zk[j]                     thread 0: Z'[0], Z'[s], Z'[2 s] (= M/2), Z'[3 s]
    C* O = rg.v + 4;    // butterfly 2 tf + 1:    inputs zk[4 + j]
    C* Om = rg.v + 8;   // butterfly s - 1 - 2 tf: inputs Z'[M - (2 tf + 1 + (3 - j) s)] = zm[4 + 3 - j]
    C* Em = rg.v + 12;  // butterfly s - 2 tf:    inputs zm[3 - j]                                   thread 0 (butterfly s/2): Z'[s/2], Z'[s/2 + s], Z'[M - s/2 - s], Z'[M - s/2]
    E[0] = sel_mask(l0m, l0, z0, zk[0]);
    E[1] = zk[1];
    E[2] = sel_mask(l0m, l0, mid2, zk[2]);
    E[3] = sel_mask(l0m, l0, zm[1], zk[3]);
    LRA_UNROLL
    for (int j = 0; j < 4; ++j) { O[j] = zk[4 + j]; Om[j] = zm[7 - j]; }
    Em[0] = sel_mask(l0m, l0, zk[2], zm[3]);
    Em[1] = sel_mask(l0m, l0, zk[3], zm[2]);
    Em[2] = sel_mask(l0m, l0, zm[3], zm[1]);
    Em[3] = sel_mask(l0m, l0, zm[2], zm[0]);
    LRA_UNROLL
    for (int i = 0; i < 4; ++i) Dft<4, T>::run(rg.v + 4 * i);
    // pass 0 (s = 1): output j of butterfly b is element 4 b + j, at slot i + (i >> 3): E and O fill elements 8 tf .. 8 tf + 7, Om and Em elements
    // 1020 - 8 tf .. 1027 - 8 tf (thread 0's second half: butterfly 128, elements 512 .. 515)
    constexpr int sz = (int)sizeof(C), M = Cfg::M;
    const int bA = 9 * tf * sz;
    const int bOm = (M - 4 - 8 * tf + ((M - 4 - 8 * tf) >> 3)) * sz;
    const int em0 = l0 ? M / 2 : M - 8 * tf;
    const int bEm = (em0 + (em0 >> 3)) * sz;
    LRA_UNROLL
    for (int j = 0; j < 4; ++j) {
        lds_st<C>(fr, bA + j * sz, E[j]);
        lds_st<C>(fr, bA + (4 + j) * sz, O[j]);
        lds_st<C>(fr, bOm + j * sz, Om[j]);
        lds_st<C>(fr, bEm + j * sz, Em[j]);
    }
}
// phase: inputs of the middle pass (radix 16, butterfly tf: elements tf + 64 j) from that layout
template <class Cfg> LRA_HD void istft_mir4_mid_read(FftRegs<Cfg>& rg, Lds fr, int tf) {
    using C = typename Cfg::cplx;
    constexpr int sin = Cfg::M >> Cfg::logr(1);
    const int base = (tf + (tf >> 3)) * (int)sizeof(C);
    LRA_UNROLL
    for (int j = 0; j < 16; ++j) rg.v[j] = lds_ld<C>(fr, base + j * (sin + (sin >> 3)) * (int)sizeof(C));
}

// ---- phase: last pass butterflies, then windowed time-domain frame -> LDS (natural order) ------
template <class Cfg> LRA_HD void istft_last_write(const IstftArgs<typename Cfg::real>& a, FftRegs<Cfg>& rg, int tf, Lds fr) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int p = Cfg::P - 1, lr = Cfg::logr(p), r = 1 << lr, nb = Cfg::R >> lr;
    const C* __restrict__ ws2 = reinterpret_cast<const C*>(a.win_scaled);
    C* v = rg.v;
    pass_dft<Cfg, p>(rg, tf, a.tw);
    LRA_UNROLL
    for (int i = 0; i < nb; ++i) {
        LRA_UNROLL
        for (int j = 0; j < r; ++j) {
            const int q = last_pass_pos<Cfg>(tf, i, j);
            const C z = v[i * r + j];  // = conj(z'[q]); x[2q] = Re z', x[2q+1] = Im z'
            const C w = Cfg::HOIST ? rg.win2[i * r + j] : ws2[q];
            lds_st<C>(fr, q * (int)sizeof(C), mk<T>(z.x * w.x, -z.y * w.y));
        }
    }
}

// ---- phase: overlap-add of ONE frame into the slot's carry; finalise one hop of output -----------
// Padded positions [t*hop, t*hop + max(N, hop)): value = carry_in[u] + frame[u]; the first `hop`
// positions are final after frame t (later frames start beyond them) and are normalised and stored,
// the rest become the carry for frame t+1.  Contributions are therefore added in increasing frame
// order, the reference's accumulation order (core/spectrum.py:593-603, 629-643).
template <class Cfg> LRA_HD void istft_store_sample(const IstftArgs<typename Cfg::real>& a, long long clip, long long sp, long long write_lo, long long write_hi,
                                                    typename Cfg::real val) {
    using T = typename Cfg::real;
    const long long s = sp - a.drop;
    if (sp >= write_lo && sp < write_hi && s >= 0 && s < a.out_len) {
        a.y[clip * a.y_stride + s] = val * a.wss[s];
    }
}

// DEFER: the finished samples of this hop stay in rg.out (hop <= n_fft/4) and are stored by
// istft_flush_out after the next frame's prefetch has been consumed.
template <class Cfg, bool DEFER> LRA_HD void istft_ola_step(const IstftArgs<typename Cfg::real>& a, long long clip, int t, bool contribute, long long write_lo,
                                                            long long write_hi, int parity, int tf, FftRegs<Cfg>& rg, Lds slot_lds) {
    using T = typename Cfg::real;
    constexpr int N = Cfg::N, NPF = FftRegs<Cfg>::NPF;
    const int H = a.hop;
    const int CL = N > H ? N - H : 0;
    const int span = N > H ? N : H;
    const Lds fr = slot_lds;
    const Lds carry_in = lds_sub(slot_lds, Cfg::FRAME_BYTES + parity * N * (int)sizeof(T));
    const Lds carry_out = lds_sub(slot_lds, Cfg::FRAME_BYTES + (1 - parity) * N * (int)sizeof(T));
    const long long pa = (long long)t * H;
    if (DEFER) {
        // u = tf + c TF < hop: final samples -> registers
        LRA_UNROLL
        for (int c = 0; c < NPF; ++c) {
            const int u = tf + c * Cfg::TF;
            T val = (T)0;
            if (u < H) {
                if (u < CL) val = lds_ld<T>(carry_in, u * (int)sizeof(T));
                if (contribute && u < N) val += lds_ld<T>(fr, u * (int)sizeof(T));
            }
            rg.out[c] = val;
        }
        // the rest becomes the carry for frame t + 1; eight positions per trip with all LDS reads issued
        // before the first dependent add/store (a one-position loop is a chain of LDS round trips)
        for (int u0 = tf + ((H - tf + Cfg::TF - 1) / Cfg::TF) * Cfg::TF; u0 < span; u0 += 8 * Cfg::TF) {
            T cv[8], fv[8];
            LRA_UNROLL
            for (int q = 0; q < 8; ++q) {
                const int u = u0 + q * Cfg::TF;
                cv[q] = u < CL ? lds_ld<T>(carry_in, u * (int)sizeof(T)) : (T)0;
                fv[q] = (contribute && u < N) ? lds_ld<T>(fr, u * (int)sizeof(T)) : (T)0;
            }
            LRA_UNROLL
            for (int q = 0; q < 8; ++q) {
                const int u = u0 + q * Cfg::TF;
                if (u < span && u - H < CL) lds_st<T>(carry_out, (u - H) * (int)sizeof(T), cv[q] + fv[q]);
            }
        }
        return;
    }
    for (int u0 = tf; u0 < span; u0 += 8 * Cfg::TF) {
        T cv[8], fv[8];
        LRA_UNROLL
        for (int q = 0; q < 8; ++q) {
            const int u = u0 + q * Cfg::TF;
            cv[q] = u < CL ? lds_ld<T>(carry_in, u * (int)sizeof(T)) : (T)0;
            fv[q] = (contribute && u < N) ? lds_ld<T>(fr, u * (int)sizeof(T)) : (T)0;
        }
        LRA_UNROLL
        for (int q = 0; q < 8; ++q) {
            const int u = u0 + q * Cfg::TF;
            if (u >= span) continue;
            const T val = cv[q] + fv[q];
            if (u < H) istft_store_sample<Cfg>(a, clip, pa + u, write_lo, write_hi, val);
            else if (u - H < CL) lds_st<T>(carry_out, (u - H) * (int)sizeof(T), val);
        }
    }
}

// ---- row-aligned overlap-add (hop = n_fft/2, n_fft/4 or n_fft/8) ------------------------------------------
// Thread tf owns the sample PAIRS p = tf + c TF (c < R) of the frame span, i.e. one residue class mod TF.
// Because the hop is a whole number hc of such rows, (a) "final vs. carry" is a wave-uniform property of
// c -- no per-sample bound tests --, (b) the carry shift by one hop maps a thread's pairs onto its own
// pairs, so the carry is updated IN PLACE (reads of a chunk before its writes; chunks ascend), and
// (c) every LDS access is one per-thread base plus an immediate, as 8-byte reads and writes.
// The row count per hop, hc = hop / (2 TF), is a compile-time parameter HC of the kernel (all "final or
// carry?" decisions fold away); instantiated for hop = n_fft/2, n_fft/4 and n_fft/8 (HC = R/2, R/4, R/8).
template <class Cfg> LRA_HD int istft_rows_hc(int hop) {
    if (Cfg::R >= 2 && 2 * hop == Cfg::N) return Cfg::R / 2;
    if (Cfg::R >= 4 && 4 * hop == Cfg::N) return Cfg::R / 4;
    if (Cfg::R >= 8 && 8 * hop == Cfg::N) return Cfg::R / 8;
    if (Cfg::R >= 16 && 16 * hop == Cfg::N) return Cfg::R / 16;  // (instantiated for the 16-point configurations: n_fft = 8192 at hop 512 ran 9 ms on the general path)
    return 0;
}

// Last FFT pass fused with the row-aligned overlap-add: the last pass leaves thread tf with exactly the sample
// pairs it owns in the overlap-add (output position tf + i TF + j s = pair c = i + nb j of its residue class), so
// the windowed frame never goes through LDS: sum = carry[c] + frame[c] straight from the butterfly registers.
template <class Cfg, int HC> LRA_HD void istft_last_ola_rows(const IstftArgs<typename Cfg::real>& a, bool contribute, int tf, FftRegs<Cfg>& rg, Lds slot_lds) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int p = Cfg::P - 1, lr = Cfg::logr(p), r = 1 << lr, nb = Cfg::R >> lr;
    // float: ONE chunk -- all carry pairs are read before the butterflies below (their latency hides behind ~100 packed instructions),
    // the frame is added, everything is written.  (In chunks of 8 the second chunk's reads sat behind the first chunk's writes -- the
    // compiler cannot reorder LDS accesses it cannot tell apart -- i.e. two more dependent LDS round trips per frame.)
    constexpr int R = Cfg::R, TF = Cfg::TF, CH = sizeof(T) == 4 ? R : (R < 8 ? R : 8);
    constexpr int hc = HC, clc = R - hc;
    static_assert((1 << Cfg::logs(p)) == nb * TF, "last-pass output stride must be nb rows");
    const C* __restrict__ ws2 = reinterpret_cast<const C*>(a.win_scaled);
    const Lds carry = lds_sub(slot_lds, Cfg::FRAME_BYTES);
    const C zero = mk<T>((T)0, (T)0);
    const int rbase = tf * (int)sizeof(C);
    const int wbase = (tf - hc * TF) * (int)sizeof(C);  // pair c lands on pair c - hc
    constexpr bool REGC = istft_reg_carry<Cfg, HC>();
    static_assert(!REGC || CH == R, "register carry: one chunk");
    // A frame that adds nothing (warm-up before the clip, drain behind it, an inactive slot's) was LOADED as an all-zero spectrum
    // (istft_spec_load / _mir under the same condition as `contribute`), so its transform is exactly zero and `carry + frame` is the
    // carry: no `contribute ? sum : carry` select per value (2 R v_cndmask per frame), and no second arm -- as a scalar branch around two
    // bodies hipcc resolved the carry's PHIs with 48 v_mov per frame behind the join.
    (void)contribute;
    C cv0[CH];
    if (CH == R) {
        LRA_UNROLL
        for (int q = 0; q < CH; ++q) cv0[q] = q < clc ? (REGC ? rg.cry[q] : lds_ld<C>(carry, rbase + q * TF * (int)sizeof(C))) : zero;
    }
    pass_dft<Cfg, p>(rg, tf, a.tw);
    LRA_UNROLL
    for (int c0 = 0; c0 < R; c0 += CH) {
        C cv[CH];
        LRA_UNROLL
        for (int q = 0; q < CH; ++q) cv[q] = CH == R ? cv0[q] : ((c0 + q) < clc ? lds_ld<C>(carry, rbase + (c0 + q) * TF * (int)sizeof(C)) : zero);
        LRA_UNROLL
        for (int q = 0; q < CH; ++q) {
            const int c = c0 + q, i = c % nb, j = c / nb;
            const C z = rg.v[i * r + j];  // = conj(z'[pos]); x[2 pos] = Re z', x[2 pos + 1] = Im z'
            const C w = Cfg::HOIST ? rg.win2[i * r + j] : ws2[last_pass_pos<Cfg>(tf, i, j)];
            const C val = fma_conj_win(z, w, cv[q]);  // (cv.x + z.x w.x, cv.y - z.y w.y): one packed FMA
            if (c < hc) {
                if (2 * c + 1 < FftRegs<Cfg>::NPFX) { rg.out[2 * c] = val.x; rg.out[2 * c + 1] = val.y; }
            } else if (REGC) {
                rg.cry[c - hc] = val;
            } else {
                lds_st<C>(carry, wbase + c * TF * (int)sizeof(C), val);
            }
        }
    }
}

// held-back samples of frame t -> y: pair c of thread tf is padded positions t hop + 2 (tf + c TF) + {0, 1}.
// istft_wss_rows loads their window sum-square values while frame t is still being transformed (one whole
// frame ahead of the stores); lanes that will store nothing read wss[0].
// A frame's held-back samples are stored iff the frame lies in the slot's strip (wave-uniform: the
// [write_lo, write_hi) test of the general path, taken over whole hops) and the sample index falls inside
// the output: per thread two 32-bit offset bounds, worked out once per frame.
template <class Cfg> struct IstftRowWin {
    long long s0;    // output index of this thread's pair 0, sample 0
    int lo, hi;      // offsets off with lo <= off < hi are inside [0, out_len)
};
template <class Cfg> LRA_HD IstftRowWin<Cfg> istft_row_window(const IstftArgs<typename Cfg::real>& a, int t, bool in_strip, int tf) {
    IstftRowWin<Cfg> w;
    w.s0 = (long long)t * a.hop + 2 * tf - a.drop;
    const long long big = 1 << 30;
    long long lo = -w.s0, hi = a.out_len - w.s0;
    lo = lo < -big ? -big : (lo > big ? big : lo);
    hi = hi < -big ? -big : (hi > big ? big : hi);
    w.lo = (int)lo;
    w.hi = in_strip ? (int)hi : w.lo;  // empty window when the frame is outside the strip
    return w;
}
// The usual frame: its whole hop block lies inside the output and is stored by this slot, and the block starts on an 8-byte
// boundary (even hop, even clip stride).  Wave-uniform, so the fast paths below are behind ONE scalar branch: unconditional 8-byte
// loads / stores of whole sample pairs.  (Per-sample bound tests compiled to a dozen exec-mask branches and 4-byte accesses per frame.)
// Uniform only where a slot is one or more whole waves (TF >= 64); narrower slots take the same paths under a per-lane condition.
template <class Cfg, int HC> LRA_HD bool istft_rows_interior(const IstftArgs<typename Cfg::real>& a, long long clip, int t, bool in_strip) {
    using T = typename Cfg::real;
    const long long s_first = (long long)t * a.hop - a.drop;  // output index of the block's first sample (thread 0, pair 0)
    const bool inside = in_strip && s_first >= 0 && s_first + 2 * HC * Cfg::TF <= a.out_len;
    const bool aligned = ((reinterpret_cast<size_t>(a.y + clip * a.y_stride + s_first) | reinterpret_cast<size_t>(a.wss + s_first)) & (2 * sizeof(T) - 1)) == 0;
    return inside && aligned;
}
template <class Cfg, int HC> LRA_HD void istft_wss_rows(const IstftArgs<typename Cfg::real>& a, long long clip, int t, bool in_strip, int tf, FftRegs<Cfg>& rg) {
    using T = typename Cfg::real;
    constexpr int TF = Cfg::TF;
    const IstftRowWin<Cfg> w = istft_row_window<Cfg>(a, t, in_strip, tf);
    const T* __restrict__ wb = a.wss + w.s0;
    const bool interior = istft_rows_interior<Cfg, HC>(a, clip, t, in_strip);
    if (Cfg::TF >= 64 ? (bool)LRA_UNIFORM(interior) : interior) {  // (a slot narrower than a wave shares it with other slots: per-lane there)
        LRA_UNROLL
        for (int c = 0; c < HC; ++c) {
            const cx<T> p2 = *reinterpret_cast<const cx<T>*>(wb + 2 * c * TF);
            rg.wv[2 * c] = p2.x;
            rg.wv[2 * c + 1] = p2.y;
        }
        LRA_ARM_END("wss rows: interior frame");
        return;
    }
    LRA_UNROLL
    for (int i = 0; i < 2 * HC; ++i) {
        const int off = 2 * (i >> 1) * TF + (i & 1);
        rg.wv[i] = (off >= w.lo && off < w.hi) ? wb[off] : a.wss[0];
    }
}
template <class Cfg, int HC> LRA_HD void istft_flush_rows(const IstftArgs<typename Cfg::real>& a, long long clip, int t, bool in_strip, int tf, const FftRegs<Cfg>& rg) {
    using T = typename Cfg::real;
    constexpr int TF = Cfg::TF;
    const IstftRowWin<Cfg> w = istft_row_window<Cfg>(a, t, in_strip, tf);
    T* __restrict__ yb = a.y + clip * a.y_stride + w.s0;
    // normalise unconditionally (every load is consumed on every control path; see istft_flush_out): a multiplication by the
    // host-built factor 1 / wss (1 where the envelope is tiny), within 1 ulp of the reference's division (core/spectrum.py:624) -- one
    // packed multiply per sample pair instead of v_rcp + v_mul + v_cmp + v_cndmask per sample
    T val[2 * HC];
    LRA_UNROLL
    for (int c = 0; c < HC; ++c) {
        const cx<T> pr = pmul(mk<T>(rg.out[2 * c], rg.out[2 * c + 1]), mk<T>(rg.wv[2 * c], rg.wv[2 * c + 1]));
        val[2 * c] = pr.x;
        val[2 * c + 1] = pr.y;
        LRA_KEEP(val[2 * c]);
        LRA_KEEP(val[2 * c + 1]);
    }
    const bool interior = istft_rows_interior<Cfg, HC>(a, clip, t, in_strip);
    if (Cfg::TF >= 64 ? (bool)LRA_UNIFORM(interior) : interior) {  // (a slot narrower than a wave shares it with other slots: per-lane there)
        LRA_UNROLL
        for (int c = 0; c < HC; ++c) *reinterpret_cast<cx<T>*>(yb + 2 * c * TF) = mk<T>(val[2 * c], val[2 * c + 1]);
        LRA_ARM_END("flush rows: interior frame");
        return;
    }
    LRA_UNROLL
    for (int i = 0; i < 2 * HC; ++i) {
        const int off = 2 * (i >> 1) * TF + (i & 1);
        if (off >= w.lo && off < w.hi) yb[off] = val[i];
    }
}

template <class Cfg> LRA_HD void istft_flush_out(const IstftArgs<typename Cfg::real>& a, long long clip, int t, long long write_lo, long long write_hi, int tf,
                                                 const FftRegs<Cfg>& rg) {
    using T = typename Cfg::real;
    constexpr int NPF = FftRegs<Cfg>::NPF;
    const long long pa = (long long)t * a.hop;
    // all window-sum-square loads first (clamped index for the lanes that store nothing), then the
    // divides and stores: one memory round trip per hop instead of one per sample
    T w[NPF];
    long long s[NPF];
    bool ok[NPF];
    LRA_UNROLL
    for (int c = 0; c < NPF; ++c) {
        const int u = tf + c * Cfg::TF;
        const long long sp = pa + u;
        s[c] = sp - a.drop;
        ok[c] = u < a.hop && sp >= write_lo && sp < write_hi && s[c] >= 0 && s[c] < a.out_len;
        w[c] = a.wss[ok[c] ? s[c] : 0];
    }
    // normalise unconditionally (lanes that store nothing divide by wss[0]): every load is then consumed
    // on every control path -- a load left pending behind a skipped branch makes hipcc wait for vmcnt(0),
    // i.e. for the previous stores too, the next time its destination register is reused
    T val[NPF];
    LRA_UNROLL
    for (int c = 0; c < NPF; ++c) {
        val[c] = rg.out[c] * w[c];
        LRA_KEEP(val[c]);
    }
    LRA_UNROLL
    for (int c = 0; c < NPF; ++c)
        if (ok[c]) a.y[clip * a.y_stride + s[c]] = val[c];
}

// One workgroup = FPB slots; slot s owns strip (blk*FPB + s) of the (clip, strip) grid and walks its
// frames one at a time: warm-up frames (their contribution to the strip's positions), the strip's
// own frames, and -- for the last strip of a clip -- drain steps that flush the carry.
template <class Cfg> struct IstftSlot {
    long long clip;
    int t0, t1;
    long long write_lo, write_hi;
    bool active;
    bool last;  // last strip of its clip: it also flushes the tail beyond its frames
};
template <class Cfg> LRA_HD IstftSlot<Cfg> istft_slot(const IstftArgs<typename Cfg::real>& a, int blk, int slot) {
    IstftSlot<Cfg> s;
    const long long sid = (long long)blk * Cfg::FPB + slot;
    s.clip = sid / a.strips_per_clip;
    const int strip = (int)(sid % a.strips_per_clip);
    s.t0 = strip * a.strip_frames;
    s.t1 = s.t0 + a.strip_frames;
    if (s.t1 > a.n_used) s.t1 = a.n_used;
    s.write_lo = (long long)s.t0 * a.hop;
    s.write_hi = strip == a.strips_per_clip - 1 ? (long long)0x7fffffffffffffffLL : (long long)s.t1 * a.hop;
    s.last = strip == a.strips_per_clip - 1;
    s.active = s.clip < a.batch;
    if (!s.active) s.clip = 0;
    return s;
}

// HC > 0: istft_rows_hc(hop), chosen by the host: only the row-aligned overlap-add is compiled in.
template <class Cfg, int HC = 0> LRA_HD void istft_block(const IstftArgs<typename Cfg::real>& a_in, const int blk, Lds lds) {
    IstftArgs<typename Cfg::real> a = a_in;
    using T = typename Cfg::real;
    constexpr int FPB = Cfg::FPB;
    constexpr bool ROWS = HC > 0;
    constexpr bool MIR4 = ROWS && istft_mir4_ok<Cfg>();                // ... with four first-pass butterflies per thread and 16-byte spectrum loads (radices 4, 16, 16)
    constexpr bool MIR = (ROWS && istft_mir_ok<Cfg>()) || MIR4;      // Hermitian step in registers, fused into the first pass
    constexpr int SB = istft_slot_bytes<Cfg, HC>();
    // uniform step count: drain steps only when one of this workgroup's slots owns a clip's last strip
    bool has_last = false;
    for (int s = 0; s < FPB; ++s) has_last = has_last || (((long long)blk * FPB + s) % a.strips_per_clip) == a.strips_per_clip - 1;
    const int steps = a.warm_frames + a.strip_frames + (has_last ? a.drain_steps : 0);
    const bool defer = ROWS || 4 * a.hop <= Cfg::N;  // finished samples per hop fit the hold-back registers
    constexpr bool rows = ROWS;
    LRA_TICK_DECL;  // (the pass macro shared with the forward kernel ticks; the inverse kernel does not report)
    LRA_REGS(FftRegs<Cfg>, rg, Cfg::NT);
    // the slot's strip (64-bit divisions) is worked out once, from the un-laundered thread index
    LRA_REGS(IstftSlot<Cfg>, sl, Cfg::NT);
    LRA_PHASE(Cfg::NT, tid) {
        const int slot = slot_of<Cfg>(tid), tf = lane_of<Cfg>(tid);
        hoist_tables<Cfg>(LRA_R(rg), tf, a.win_scaled, a.tw, a.twr, true);
        if constexpr (MIR4) {
            LRA_UNROLL
            for (int q = 0; q < Cfg::R / 2; ++q) LRA_R(rg).twr[q] = a.twr[mir4_slot_bin<Cfg>(tf, q)];
        } else if constexpr (MIR) {
            LRA_UNROLL
            for (int q = 0; q < Cfg::R / 2; ++q) LRA_R(rg).twr[q] = a.twr[mir_slot_bin<Cfg>(tf, q)];
        }
        const Lds c0 = lds_sub(lds, slot * SB + Cfg::FRAME_BYTES);
        for (int u = tf; u < (HC > 0 ? istft_carry_reals<Cfg, HC>() : Cfg::N); u += Cfg::TF) lds_st<T>(c0, u * (int)sizeof(T), (T)0);
        if constexpr (istft_reg_carry<Cfg, HC>()) {
            LRA_UNROLL
            for (int q = 0; q < Cfg::R; ++q) LRA_R(rg).cry[q] = mk<T>((T)0, (T)0);
        }
        LRA_R(sl) = istft_slot<Cfg>(a, blk, slot_of<Cfg>(LRA_RAW_TID(tid)));
        const IstftSlot<Cfg> s = LRA_R(sl);
        const int t = s.t0 - a.warm_frames;
        if constexpr (MIR4) istft_spec_load_mir4<Cfg>(a, s.clip, t, s.active && t >= 0 && t < s.t1, tf, LRA_R(rg));
        else if constexpr (MIR) istft_spec_load_mir<Cfg>(a, s.clip, t, s.active && t >= 0 && t < s.t1, tf, LRA_R(rg));
        else istft_spec_load<Cfg>(a, s.clip, t, s.active && t >= 0 && t < s.t1, tf, LRA_R(rg));
    } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
    for (int j = 0; j < steps; ++j) {
        if (!Cfg::HOIST) { LRA_LAUNDER(a.win_scaled); LRA_LAUNDER(a.tw); LRA_LAUNDER(a.twr); }
        lra_setprio<LRA_I_PRIO_A>();   // wave priority per phase (s_setprio, -1 = none): A = un-split + loads + held-back stores, B = the passes + overlap-add
        // (a) split the prefetched spectrum of frame j into LDS, (b) only now issue the held-back output
        // stores of frame j-1, (c) start the prefetch of frame j+1: the wait in (a) never covers (b)
        LRA_PHASE(Cfg::NT, tid) {
            const int slot = slot_of<Cfg>(tid), tf = lane_of<Cfg>(tid);
            const IstftSlot<Cfg> s = LRA_R(sl);
            const int t = s.t0 - a.warm_frames + j;
#if defined(__HIP_DEVICE_COMPILE__) && defined(LRA_ISTFT_ABLATE)
            if constexpr ((LRA_ISTFT_ABLATE & 4) != 0 && sizeof(T) == 4) {  // (the experiment's load targets stay allocated while their loads are in flight)
                LRA_UNROLL
                for (int q = 0; q < Cfg::R; ++q) { LRA_KEEP(LRA_R(rg).nxt[q].x); LRA_KEEP(LRA_R(rg).nxt[q].y); }
            }
#endif
            if constexpr (MIR4) istft_unsplit_pass0_mir4<Cfg>(tf, LRA_R(rg), lds_sub(lds, slot * SB));
            else if constexpr (MIR) istft_unsplit_pass0<Cfg>(tf, LRA_R(rg), lds_sub(lds, slot * SB));
            else istft_split_write<Cfg>(a, tf, LRA_R(rg), lds_sub(lds, slot * SB));
#ifndef LRA_ISTFT_ABLATE  // timing experiments (scripts/ab_run.sh): bit 0 = no spectrum loads in the frame loop, bit 1 = no output stores
#define LRA_ISTFT_ABLATE 0
#endif
            const bool ablate_never = LRA_ISTFT_ABLATE != 0 && a.tiny == (T)12345.678;  // never true at run time
#ifndef LRA_ISTFT_LOADS_FIRST
#define LRA_ISTFT_LOADS_FIRST 1
#endif
            auto prefetch = [&]() {
                if (j + 1 < steps && (!(LRA_ISTFT_ABLATE & 1) || ablate_never)) {
                    if constexpr (MIR4) istft_spec_load_mir4<Cfg>(a, s.clip, t + 1, s.active && t + 1 >= 0 && t + 1 < s.t1, tf, LRA_R(rg));
                    else if constexpr (MIR) istft_spec_load_mir<Cfg>(a, s.clip, t + 1, s.active && t + 1 >= 0 && t + 1 < s.t1, tf, LRA_R(rg));
                    else istft_spec_load<Cfg>(a, s.clip, t + 1, s.active && t + 1 >= 0 && t + 1 < s.t1, tf, LRA_R(rg));
                }
            };
            // the next frame's spectrum loads go out BEFORE the held-back stores: vector-memory operations retire in order, so loads
            // queued behind stores are only seen complete once those stores have been acknowledged
            if (LRA_ISTFT_LOADS_FIRST) prefetch();
            if (defer && j > 0 && s.active && (!(LRA_ISTFT_ABLATE & 2) || ablate_never)) {
                if constexpr (rows) istft_flush_rows<Cfg, HC>(a, s.clip, t - 1, t - 1 >= s.t0 && (s.last || t - 1 < s.t1), tf, LRA_R(rg));
                else istft_flush_out<Cfg>(a, s.clip, t - 1, s.write_lo, s.write_hi, tf, LRA_R(rg));
            }
            if (!LRA_ISTFT_LOADS_FIRST) prefetch();
            if constexpr (rows) {  // for the flush of THIS frame, one iteration from now
                if (s.active && (!(LRA_ISTFT_ABLATE & 8) || ablate_never)) istft_wss_rows<Cfg, HC>(a, s.clip, t, t >= s.t0 && (s.last || t < s.t1), tf, LRA_R(rg));
            }
#if defined(__HIP_DEVICE_COMPILE__) && (LRA_ISTFT_ABLATE & 4)
            // experiment: the next frame's spectrum is read (16 loads the compiler does not know about, into registers nothing else uses) and never
            // waited for -- what the read traffic costs the arithmetic when nothing waits on it
            if constexpr (MIR && sizeof(T) == 4) {
                if (j + 1 < steps && s.active && t + 1 >= 0 && t + 1 < s.t1) {
                    const typename Cfg::cplx* X = a.D + s.clip * a.d_batch_stride + (long long)(t + 1) * a.d_frame_stride;
                    LRA_UNROLL
                    for (int q = 0; q < Cfg::R / 2; ++q) {
                        const typename Cfg::cplx* pk = X + tf + q * 2 * Cfg::TF;
                        const typename Cfg::cplx* pm = X + Cfg::M - tf - q * 2 * Cfg::TF;
                        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(LRA_R(rg).nxt[2 * q]) : "v"(pk));
                        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(LRA_R(rg).nxt[2 * q + 1]) : "v"(pm));
                    }
                }
                LRA_UNROLL
                for (int q = 0; q < Cfg::R; ++q) { LRA_KEEP(LRA_R(rg).nxt[q].x); LRA_KEEP(LRA_R(rg).nxt[q].y); }
            }
#endif
        } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
        lra_setprio<LRA_I_PRIO_B>();
        // pass 0 (no twiddles) reads from LDS here, unlike the forward kernel (unless it was fused into the Hermitian step)
        if (Cfg::P > 1 && !MIR) {
            LRA_PHASE(Cfg::NT, tid) {
                pass_read<Cfg, 0>(LRA_R(rg).v, lds_sub(lds, (slot_of<Cfg>(tid)) * SB), lane_of<Cfg>(tid));
            } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
            LRA_PHASE(Cfg::NT, tid) {
                pass_dft<Cfg, 0>(LRA_R(rg), lane_of<Cfg>(tid), a.tw);
                pass_write<Cfg, 0>(LRA_R(rg).v, lds_sub(lds, (slot_of<Cfg>(tid)) * SB), lane_of<Cfg>(tid));
            } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
        }
        if constexpr (MIR4) {  // the middle pass reads the first pass's own layout (istft_unsplit_pass0_mir4)
            LRA_PHASE(Cfg::NT, tid) {
                istft_mir4_mid_read<Cfg>(LRA_R(rg), lds_sub(lds, (slot_of<Cfg>(tid)) * SB), lane_of<Cfg>(tid));
            } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
            LRA_PHASE(Cfg::NT, tid) {
                pass_dft<Cfg, 1>(LRA_R(rg), lane_of<Cfg>(tid), a.tw);
                pass_write<Cfg, 1>(LRA_R(rg).v, lds_sub(lds, (slot_of<Cfg>(tid)) * SB), lane_of<Cfg>(tid));
            } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
        } else {
            if (Cfg::P > 2) { LRA_MID_PASS(Cfg, 1, rg, lds, a.tw, SB) }
            if (Cfg::P > 3) { LRA_MID_PASS(Cfg, 2, rg, lds, a.tw, SB) }
        }
        LRA_PHASE(Cfg::NT, tid) {
            pass_read<Cfg, Cfg::P - 1>(LRA_R(rg).v, lds_sub(lds, (slot_of<Cfg>(tid)) * SB), lane_of<Cfg>(tid));
        } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
        if constexpr (rows) {  // last pass + window + overlap-add in one phase (an inactive slot still runs the butterflies: harmless)
            LRA_PHASE(Cfg::NT, tid) {
                const int slot = slot_of<Cfg>(tid), tf = lane_of<Cfg>(tid);
                const IstftSlot<Cfg> s = LRA_R(sl);
                const int t = s.t0 - a.warm_frames + j;
                if (s.active) istft_last_ola_rows<Cfg, HC>(a, t >= 0 && t < s.t1, tf, LRA_R(rg), lds_sub(lds, slot * SB));
            } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
            continue;
        }
        LRA_PHASE(Cfg::NT, tid) {
            istft_last_write<Cfg>(a, LRA_R(rg), lane_of<Cfg>(tid), lds_sub(lds, (slot_of<Cfg>(tid)) * SB));
        } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
        LRA_PHASE(Cfg::NT, tid) {
            const int slot = slot_of<Cfg>(tid), tf = lane_of<Cfg>(tid);
            const IstftSlot<Cfg> s = LRA_R(sl);
            const int t = s.t0 - a.warm_frames + j;
            if (s.active) {
                if (defer) istft_ola_step<Cfg, true>(a, s.clip, t, t >= 0 && t < s.t1, s.write_lo, s.write_hi, j & 1, tf, LRA_R(rg), lds_sub(lds, slot * SB));
                else istft_ola_step<Cfg, false>(a, s.clip, t, t >= 0 && t < s.t1, s.write_lo, s.write_hi, j & 1, tf, LRA_R(rg), lds_sub(lds, slot * SB));
            }
        } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
    }
    if (defer && steps > 0) {
        LRA_PHASE(Cfg::NT, tid) {
            const int slot = slot_of<Cfg>(tid), tf = lane_of<Cfg>(tid);
            const IstftSlot<Cfg> s = LRA_R(sl);
            if (s.active) {
                const int tl = s.t0 - a.warm_frames + steps - 1;
                if constexpr (rows) istft_flush_rows<Cfg, HC>(a, s.clip, tl, tl >= s.t0 && (s.last || tl < s.t1), tf, LRA_R(rg));
                else istft_flush_out<Cfg>(a, s.clip, s.t0 - a.warm_frames + steps - 1, s.write_lo, s.write_hi, tf, LRA_R(rg));
            }
        } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
    }
}

}  // namespace lra
