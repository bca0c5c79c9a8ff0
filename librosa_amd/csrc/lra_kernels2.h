// lra_kernels2.h -- second-generation forward kernel body (stft_block2): same results as stft_block (lra_kernels.h),
// half of its LDS traffic.  Reference semantics: librosa/core/spectrum.py:380-390 (rfft of the windowed frames),
// :3000-3013 (|X|^power).
//
// Measured on MI355X (profiles/r02_*): the first-generation kernel is bound by the LDS -- 470 LDS-array cycles per
// frame against one frame per ~930 cycles per CU with every store and load removed -- not by the VALU (23 % busy)
// and not by HBM.  Two of its four LDS round trips per frame are not arithmetic at all:
//
//   * the PCM ring.  A slot walks CONSECUTIVE frames, and with the hop a whole number of pass-0 rows (hop = n_fft/HD,
//     HD in {1, 2, 4, 8}) frame t+1's pass-0 input element (tf, j) IS frame t's element (tf, j + r0/HD): the samples a
//     thread needs next time are already in ITS OWN registers.  The ring therefore lives in 2 R registers per thread
//     (`raw`), shifted by register moves; only the R/HD new sample pairs per frame are loaded (8-byte loads, one frame
//     ahead, issued before the frame's stores so that the wait for them never covers those stores).  No ring in LDS:
//     16 reads + 8 writes per frame gone, and the slot shrinks from 17 KB to 8.7 KB (12 instead of 9 waves per CU).
//
//   * the last pass's write + the split step's read.  The split needs Z[k] and Z[M-k] in one thread.  In the last
//     Stockham pass (radix r, s = M/r butterflies, output of butterfly b at b + j s) the mirror of an output of
//     butterfly b belongs to butterfly s - b.  Nothing ties butterfly b to thread b: thread tf takes butterflies tf and
//     s - tf (thread 0: 0 and s/2, the two self-mirrored ones), and after the pass every mirrored pair sits in ONE
//     thread's registers: X[k], X[M-k] come straight out of the butterflies and go to HBM (or to |X|^p) without the
//     16 ds_write_b64 + 16 ds_read_b64 of a third round trip.  Lane 0's pairs are arranged differently (both of its
//     butterflies are self-mirrored); a handful of lane-0 selects and a second pair of store bases absorb that.
//
// Applies to configurations whose last pass has two butterflies per thread (R / r_last == 2): n_fft 1024, 2048, 4096
// at 16 points per thread.  Everything else keeps stft_block.
#pragma once

#include "lra_kernels.h"

namespace lra {

template <class Cfg> constexpr bool v2_cfg_ok() {
    constexpr int pl = Cfg::P - 1;
    return Cfg::P >= 2 && Cfg::R == 16 && Cfg::HOIST && sizeof(typename Cfg::real) == 4 && affine_tf<Cfg>() && (Cfg::R >> Cfg::logr(pl)) == 2 &&
           2 * Cfg::TF == (Cfg::M >> Cfg::logr(pl)) && (Cfg::M >> Cfg::logr(pl)) % (1 << Cfg::PADSHIFT) == 0;
}
// hop = n_fft / HD with the hop a whole number of pass-0 rows
template <class Cfg> constexpr bool v2_hd_ok(int hd) { return (hd == 1 || hd == 2 || hd == 4 || hd == 8) && (1 << Cfg::logr(0)) % hd == 0; }
template <class Cfg> LRA_HD int v2_hop_divisor(int hop) {
    for (int hd = 1; hd <= 8; hd *= 2)
        if (hop * hd == Cfg::N && v2_hd_ok<Cfg>(hd)) return hd;
    return 0;
}

#ifndef LRA_V2_ROTATE_MEL
#define LRA_V2_ROTATE_MEL 0  // the rotating register ring for the mel epilogue too (measured +1 % without wave priorities, see the frame loop)
#endif
// s_setprio at the phase boundaries of the frame loop (see there; lra_common.h, lra_setprio).  Mel kernel: A = window + transform passes (A1: the
// passes apart from pass 0), S = un-split + power row, B / B2 / B3 = the epilogue's run read / accumulate / band combine; complex / power kernels:
// CA = transform, CS = un-split + stores.  -1: no instruction.  Measured on the 256 x 30 s batch (same box, alternating; product without: 0.604 ms):
// 3 / 2 / 0 / combine 1: 0.564-0.568; 3 / 2 / 0: 0.570-0.574; 3 / 2 / 1: 0.570; 3 / 3 / 0: 0.573-0.576; 2 / 3 / 0: 0.578; 3 / 0 / 0: 0.603; the
// reverse order (0 / - / 3): 0.583-0.587.  The store-bound complex kernel: CA 3 / CS 0 0.744 against 0.747 (left alone), the reverse 0.757.
#ifndef LRA_V2_STAGGER_ALL
#define LRA_V2_STAGGER_ALL 0
#endif
#ifndef LRA_V2_PRIO_A
#define LRA_V2_PRIO_A 3
#endif
#ifndef LRA_V2_PRIO_S
#define LRA_V2_PRIO_S 2
#endif
#ifndef LRA_V2_PRIO_B
#define LRA_V2_PRIO_B 0
#endif
#ifndef LRA_V2_PRIO_B2
#define LRA_V2_PRIO_B2 -1
#endif
#ifndef LRA_V2_PRIO_B3
#define LRA_V2_PRIO_B3 1
#endif
#ifndef LRA_V2_PRIO_A1
#define LRA_V2_PRIO_A1 -1
#endif
// (round 5: 3 / 2 for the complex kernel too -- over seven allocations per process and two rounds its slow level reads 0.737-0.741 / 0.732-0.735 ms against
// 0.742-0.746 / 0.739-0.745 without, profiles/r05_raw/aa_*: under 1 %, but the same sign every time; the inverse kernel's priorities (LRA_I_PRIO_*) move nothing)
#ifndef LRA_V2_PRIO_CA
#define LRA_V2_PRIO_CA 3
#endif
// |X|^p epilogue (round 5): PA = window + transform passes, PS = un-split + power stores.  Same box, alternating, 256 x 30 s:
// none 0.518-0.521 ms, 3 / 2 0.499-0.501, 3 / 0 0.498-0.499 (profiles/r05_raw/z_prio_power.txt); the complex kernel with the same pair: inside its
// allocation-to-allocation spread, left alone.
#ifndef LRA_V2_PRIO_PA
#define LRA_V2_PRIO_PA 3
#endif
#ifndef LRA_V2_PRIO_PS
#define LRA_V2_PRIO_PS 2
#endif
#ifndef LRA_V2_PRIO_CS
#define LRA_V2_PRIO_CS 2
#endif
template <int P> LRA_HD void v2_setprio() { lra_setprio<P>(); }

template <class Cfg, int HD> struct Regs2 {
    using C = typename Cfg::cplx;
    static constexpr int R = Cfg::R;
    static constexpr int r0 = 1 << Cfg::logr(0), nb0 = R >> Cfg::logr(0), sin0 = Cfg::M >> Cfg::logr(0);
    static constexpr int SJ = r0 / HD;   // pass-0 rows a frame advances by
    static constexpr int NEW = R / HD;   // new sample pairs per thread and frame
    static constexpr int rl = 1 << Cfg::logr(Cfg::P - 1);  // last-pass radix
    C v[R];            // butterfly registers
    C raw[R];          // this frame's sample pairs in pass-0 register order: element e = i r0 + j is complex index tf + i TF + j sin0
    C pf[NEW];         // next frame's new sample pairs, in flight during the current frame
    C win2[R];         // window pairs (x 1/2, see split_pair), pass-0 order
    C treg[Cfg::TREG_TOTAL];
    C twr[R / 2];      // split twiddles W_N^k of this thread's R / 2 pair slots (= rl with a two-butterfly last pass)
    // OUT_MELR (run-ordered two-slope mel epilogue, lra_mel.h layout 1): this thread's two runs of R/2 power values, the
    // restart factors of its running sums, the first MELR_PHOIST piece addresses of its two mel bands and the last
    // MELR_TILE frames' values of those bands (stored as one burst per band, see FftRegs)
    static constexpr int MELR_PHOIST = melr_ph<Cfg>(), MELR_TILE = melr_tile_frames<Cfg>(), MELR_NB = melr_nb<Cfg>();  // (two bands per thread at one wave per frame; see FftRegs)
    typename Cfg::real pw[R], pw_extra;
    typename Cfg::real keep[R];
    int mad[MELR_NB][2 * MELR_PHOIST];
    typename Cfg::real mt[MELR_NB][MELR_TILE];
    // complex index (within a frame) of pass-0 register element e
    static LRA_HD int q_of(int tf, int e) { return tf + (e / r0) * Cfg::TF + (e % r0) * sin0; }
    // pass-0 register element of new pair n
    static LRA_HD int elem_of_new(int n) { return (n / SJ) * r0 + (r0 - SJ) + (n % SJ); }
};

// second butterfly of thread tf in the last pass (the first is tf): the mirror s - tf; thread 0 takes s/2
template <class Cfg> LRA_HD int v2_mirror_bfly(int tf) { return tf == 0 ? Cfg::TF : 2 * Cfg::TF - tf; }
// "virtual thread index" of the store / twiddle addressing of pair slots q >= rl/2: tf, except for lane 0 whose slots
// there hold the pairs of butterfly s/2: bins s/2 + (q - rl/2) s = q s + (s/2)(1 - rl)
template <class Cfg> LRA_HD int v2_tf_hi(int tf) {
    constexpr int rl = 1 << Cfg::logr(Cfg::P - 1), s = 2 * Cfg::TF;
    return tf == 0 ? (s / 2) * (1 - rl) : tf;
}

template <class Cfg, int HD> LRA_HD void v2_hoist(Regs2<Cfg, HD>& rg, int tf, const typename Cfg::real* __restrict__ win, const typename Cfg::cplx* __restrict__ tw,
                                                   const typename Cfg::cplx* __restrict__ twr_full) {
    using C = typename Cfg::cplx;
    using RG = Regs2<Cfg, HD>;
    const C* __restrict__ win2 = reinterpret_cast<const C*>(win);
    LRA_UNROLL
    for (int e = 0; e < Cfg::R; ++e) rg.win2[e] = win2[RG::q_of(tf, e)];
    // middle passes: the usual butterflies tf + i TF
    if (Cfg::P > 2) load_pass_twiddles<Cfg, (1 < Cfg::P - 1 ? 1 : 0)>(rg.treg, tf, tw);
    if (Cfg::P > 3) load_pass_twiddles<Cfg, (2 < Cfg::P - 1 ? 2 : 0)>(rg.treg, tf, tw);
    // last pass: butterflies tf and its mirror
    {
        constexpr int p = Cfg::P - 1, lr = Cfg::logr(p), s = 1 << Cfg::logs(p);
        LRA_UNROLL
        for (int i = 0; i < 2; ++i) {
            const int k = i == 0 ? tf : v2_mirror_bfly<Cfg>(tf);
            LRA_UNROLL
            for (int t = 0; t < lr; ++t) rg.treg[Cfg::treg_off(p) + i * lr + t] = tw[Cfg::tw_off(p) + ((1 << t) - 1) * s + k];
        }
    }
    // split twiddles of the pair slots: slot q pairs bin k_q = tfo + q s with bin M - k_q
    {
        constexpr int rl = RG::rl, s = 2 * Cfg::TF;
        const int tfh = v2_tf_hi<Cfg>(tf);
        LRA_UNROLL
        for (int q = 0; q < rl; ++q) rg.twr[q] = twr_full[(q < rl / 2 ? tf : tfh) + q * s];
    }
}

// new sample pairs of frame `next` -> rg.pf (issued one frame ahead).  Blocks that lie inside the clip take plain
// 8-byte loads (4-byte loads when the clip is not 8-byte aligned); the few that touch the np.pad region
// (core/spectrum.py:287) go through the index fold.
template <class Cfg, int HD> LRA_HD void v2_issue_loads(const StftArgs<typename Cfg::real>& a, int clip, int next, int tf, Regs2<Cfg, HD>& rg) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    using RG = Regs2<Cfg, HD>;
    if (next >= a.n_frames) return;
    const T* __restrict__ yb = a.y + (long long)clip * a.y_stride;
    const long long p0 = (long long)next * a.hop;               // padded position of the frame's first sample
    const long long g0 = p0 + (Cfg::N - a.hop) - a.pad;         // clip position of the first NEW sample
    if (LRA_LIKELY(g0 >= 0 && g0 + a.hop <= a.n)) {
        const T* __restrict__ src = yb + (p0 - a.pad) + 2 * tf;  // this thread's pair 0 of the frame (only the new pairs are dereferenced)
        if (LRA_LIKELY((reinterpret_cast<size_t>(src) & (2 * sizeof(T) - 1)) == 0)) {
            LRA_UNROLL
            for (int n = 0; n < RG::NEW; ++n) {
                const int e = RG::elem_of_new(n);
                rg.pf[n] = *reinterpret_cast<const C*>(src + 2 * ((e / RG::r0) * Cfg::TF + (e % RG::r0) * RG::sin0));
            }
        } else {
            LRA_UNROLL
            for (int n = 0; n < RG::NEW; ++n) {
                const int e = RG::elem_of_new(n);
                const T* __restrict__ s2 = src + 2 * ((e / RG::r0) * Cfg::TF + (e % RG::r0) * RG::sin0);
                rg.pf[n] = mk<T>(s2[0], s2[1]);
            }
        }
    } else {
        LRA_UNROLL
        for (int n = 0; n < RG::NEW; ++n) {
            const long long p = p0 + 2 * RG::q_of(tf, RG::elem_of_new(n));
            rg.pf[n] = mk<T>(fetch_sample<T>(yb, p, a.pad, a.n, a.pad_mode), fetch_sample<T>(yb, p + 1, a.pad, a.n, a.pad_mode));
        }
    }
}

// all R sample pairs of a slot's first frame
template <class Cfg, int HD> LRA_HD void v2_fill(const StftArgs<typename Cfg::real>& a, int clip, int frame, int tf, Regs2<Cfg, HD>& rg) {
    using T = typename Cfg::real;
    using RG = Regs2<Cfg, HD>;
    const T* __restrict__ yb = a.y + (long long)clip * a.y_stride;
    const long long p0 = (long long)frame * a.hop;
    LRA_UNROLL
    for (int e = 0; e < Cfg::R; ++e) {
        const long long p = p0 + 2 * RG::q_of(tf, e);
        const bool live = frame < a.n_frames;
        rg.raw[e] = live ? mk<T>(fetch_sample<T>(yb, p, a.pad, a.n, a.pad_mode), fetch_sample<T>(yb, p + 1, a.pad, a.n, a.pad_mode)) : mk<T>((T)0, (T)0);
    }
}

// frame t -> t + 1: rows move down by SJ, the prefetched pairs become the last SJ rows
template <class Cfg, int HD> LRA_HD void v2_shift(Regs2<Cfg, HD>& rg) {
    using RG = Regs2<Cfg, HD>;
    LRA_UNROLL
    for (int i = 0; i < RG::nb0; ++i) {
        LRA_UNROLL
        for (int j = 0; j < RG::r0; ++j) rg.raw[i * RG::r0 + j] = j + RG::SJ < RG::r0 ? rg.raw[i * RG::r0 + j + RG::SJ] : rg.pf[i * RG::SJ + (j + RG::SJ - RG::r0)];
    }
}

// The register ring WITHOUT the shift (LRA_V2_ROTATE): the R pairs stay where they are and the frame's VIEW of them rotates.  Logical
// pass-0 row j of the slot's frame number `it` lives in physical row (j + SJ it) mod r0, so the only data that moves per frame are the
// SJ new rows per butterfly (R / HD pairs: 4 at hop = n_fft / 4) from their landing registers into the rows the previous frame read
// first; the window multiply picks its operands by the rotation, which is wave-uniform (a scalar switch with HD arms of R packed
// multiplies each -- the rest of the frame is rotation-free).  v2_shift moved all R pairs every frame: 16 v_mov_b64 at n_fft = 2048
// (+ the copies hipcc adds at the loop's back edge), 3.4 % of the kernel by ablation (profiles/r03_experiments.md section 1).
#ifndef LRA_V2_ROTATE
#define LRA_V2_ROTATE 1
#endif
// The C++ spelling of the rotating view (what the host simulator runs, and the definition of the assembly below).
template <class Cfg, int HD> LRA_HD void v2_window_rotating_ref(int it, Regs2<Cfg, HD>& rg) {
    using T = typename Cfg::real;
    using RG = Regs2<Cfg, HD>;
    constexpr int r0 = RG::r0, nb0 = RG::nb0, SJ = RG::SJ;
    const int rot = it & (HD - 1);
    for (int i = 0; i < nb0; ++i) {
        if (it > 0)
            for (int m = 0; m < SJ; ++m) rg.raw[i * r0 + (SJ * ((rot + HD - 1) % HD) + m) % r0] = rg.pf[i * SJ + m];
        for (int j = 0; j < r0; ++j) {
            const typename Cfg::cplx x = rg.raw[i * r0 + (j + SJ * rot) % r0], w = rg.win2[i * r0 + j];
            rg.v[i * r0 + j] = mk<T>(x.x * w.x, x.y * w.y);
        }
    }
}
template <class Cfg, int HD> constexpr bool v2_rotate_asm_ok() {
    return LRA_V2_ROTATE && HD == 4 && Regs2<Cfg, HD>::r0 == 16 && Regs2<Cfg, HD>::nb0 == 1 && sizeof(typename Cfg::real) == 4;
}
#if defined(LRA_PK_ASM)
// One residue class of rows (c, c + 4, c + 8, c + 12: the rows a frame's hop of four rows maps onto each other), n_fft / hop = 4, sixteen rows:
// v[c + 4 u] = P[c + 4 ((u + rot) mod 4)] * win[c + 4 u], the new pair `pf` going into P[c + 4 ((rot + 3) mod 4)] (and straight into the
// product of logical row c + 12).  The rotation is wave-uniform (SGPR): a scalar switch INSIDE the block, so that the ring's sixteen register
// pairs are tied operands of straight-line code as far as hipcc is concerned -- they never move and never pass through a PHI.  (Written as a
// C++ switch hipcc either turned the ring into a run-time-indexed array in scratch memory or copied 12-16 pairs per frame between the arms.)
__device__ __forceinline__ void v2_rot_class(pk::f2& p0, pk::f2& p1, pk::f2& p2, pk::f2& p3, pk::f2 pf, pk::f2 w0, pk::f2 w1, pk::f2 w2, pk::f2 w3, pk::f2& v0, pk::f2& v1,
                                              pk::f2& v2, pk::f2& v3, int rot, int ins) {
    asm("s_cmp_lg_u32 %13, 0\n\t"
        "s_cbranch_scc1 .Lrot1_%=\n\t"
        "s_cmp_eq_u32 %14, 0\n\t"
        "s_cbranch_scc1 .Lrot0_%=\n\t"
        "v_mov_b64 %7, %8\n"
        ".Lrot0_%=:\n\t"
        "v_pk_mul_f32 %0, %4, %9\n\t"
        "v_pk_mul_f32 %1, %5, %10\n\t"
        "v_pk_mul_f32 %2, %6, %11\n\t"
        "v_pk_mul_f32 %3, %7, %12\n\t"
        "s_branch .Lrotend_%=\n"
        ".Lrot1_%=:\n\t"
        "s_cmp_lg_u32 %13, 1\n\t"
        "s_cbranch_scc1 .Lrot2_%=\n\t"
        "v_pk_mul_f32 %0, %5, %9\n\t"
        "v_pk_mul_f32 %1, %6, %10\n\t"
        "v_pk_mul_f32 %2, %7, %11\n\t"
        "v_pk_mul_f32 %3, %8, %12\n\t"
        "v_mov_b64 %4, %8\n\t"
        "s_branch .Lrotend_%=\n"
        ".Lrot2_%=:\n\t"
        "s_cmp_lg_u32 %13, 2\n\t"
        "s_cbranch_scc1 .Lrot3_%=\n\t"
        "v_pk_mul_f32 %0, %6, %9\n\t"
        "v_pk_mul_f32 %1, %7, %10\n\t"
        "v_pk_mul_f32 %2, %4, %11\n\t"
        "v_pk_mul_f32 %3, %8, %12\n\t"
        "v_mov_b64 %5, %8\n\t"
        "s_branch .Lrotend_%=\n"
        ".Lrot3_%=:\n\t"
        "v_pk_mul_f32 %0, %7, %9\n\t"
        "v_pk_mul_f32 %1, %4, %10\n\t"
        "v_pk_mul_f32 %2, %5, %11\n\t"
        "v_pk_mul_f32 %3, %8, %12\n\t"
        "v_mov_b64 %6, %8\n"
        ".Lrotend_%=:\n\t"
        "s_nop 0"
        : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
        : "v"(pf), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "s"(rot), "s"(ins)
        : "scc");
}
#endif
// `it` is the slot's frame counter (wave-uniform: every slot of the workgroup is at the same count)
template <class Cfg, int HD> LRA_HD void v2_window_rotating(int it, Regs2<Cfg, HD>& rg) {
#if defined(LRA_PK_ASM)
    if constexpr (v2_rotate_asm_ok<Cfg, HD>()) {
        const int rot = it & 3, ins = it;  // `ins`: zero for the slot's first frame, whose ring v2_fill loaded whole (the loop counter is scalar already)
        pk::f2 p[16], o[16];
        LRA_UNROLL
        for (int e = 0; e < 16; ++e) p[e] = pk::v(rg.raw[e]);
        LRA_UNROLL
        for (int c = 0; c < 4; ++c)
            v2_rot_class(p[c], p[c + 4], p[c + 8], p[c + 12], pk::v(rg.pf[c]), pk::v(rg.win2[c]), pk::v(rg.win2[c + 4]), pk::v(rg.win2[c + 8]), pk::v(rg.win2[c + 12]), o[c],
                         o[c + 4], o[c + 8], o[c + 12], rot, ins);
        LRA_UNROLL
        for (int e = 0; e < 16; ++e) { rg.raw[e] = pk::c(p[e]); rg.v[e] = pk::c(o[e]); }
        return;
    }
#endif
    v2_window_rotating_ref<Cfg, HD>(it, rg);
}
// pass-0 butterflies on the windowed frame, first LDS write of the frame (second half of v2_pass0)
template <class Cfg, int HD> LRA_HD void v2_pass0_dft(int tf, Regs2<Cfg, HD>& rg, Lds fr) {
    using T = typename Cfg::real;
    constexpr int r0 = Regs2<Cfg, HD>::r0, nb0 = Regs2<Cfg, HD>::nb0;
    LRA_UNROLL
    for (int i = 0; i < nb0; ++i) Dft<r0, T>::run(rg.v + i * r0);
    pass_write<Cfg, 0>(rg.v, fr, tf);
}

// phase: window, pass-0 butterflies, first LDS write of the frame
template <class Cfg, int HD> LRA_HD void v2_pass0(bool live, int tf, Regs2<Cfg, HD>& rg, Lds fr) {
    using T = typename Cfg::real;
    constexpr int r0 = Regs2<Cfg, HD>::r0, nb0 = Regs2<Cfg, HD>::nb0;
    // (a frame beyond n_frames is transformed like any other -- its ring holds stale samples or the prologue's zeros and nothing of it is stored;
    // zeroing it under `live` cost 12 selects per frame, ten of them back to back: see sel_mask in lra_common.h)
    (void)live;
    LRA_UNROLL
    for (int e = 0; e < Cfg::R; ++e) rg.v[e] = mk<T>(rg.raw[e].x * rg.win2[e].x, rg.raw[e].y * rg.win2[e].y);
    LRA_UNROLL
    for (int i = 0; i < nb0; ++i) Dft<r0, T>::run(rg.v + i * r0);
    pass_write<Cfg, 0>(rg.v, fr, tf);
}

// phase: inputs of the last pass for butterflies tf (v[0 .. rl)) and its mirror (v[rl .. 2 rl))
template <class Cfg, int HD> LRA_HD void v2_last_read(Regs2<Cfg, HD>& rg, Lds fr, int tf) {
    using C = typename Cfg::cplx;
    constexpr int p = Cfg::P - 1, lr = Cfg::logr(p), r = 1 << lr, sin = Cfg::M >> lr;
    const int bA = Cfg::phys(tf) * (int)sizeof(C);
    const int bB = Cfg::phys(v2_mirror_bfly<Cfg>(tf)) * (int)sizeof(C);
    LRA_UNROLL
    for (int j = 0; j < r; ++j) {
        rg.v[j] = lds_ld<C>(fr, bA + j * pstride<Cfg>(sin) * (int)sizeof(C));
        rg.v[r + j] = lds_ld<C>(fr, bB + j * pstride<Cfg>(sin) * (int)sizeof(C));
    }
}


// staged complex epilogue (v2_store_row below): 16-byte pieces, and where a row starts relative to them
template <class T> struct alignas(16) V4 {
    T a, b, c, d;
};
template <class Cfg> LRA_HD int v2_row_shift(const StftArgs<typename Cfg::real>& a, int clip, int frame) {
    const long long row = ((long long)clip * a.n_frames + frame) * a.row_pitch;
    return (int)(reinterpret_cast<size_t>(a.D + row) & (2 * sizeof(typename Cfg::real)));  // 0 or sizeof(cplx): the row starts on / half-way into a 16-byte piece
}

// OUT_MELR: float index of bin k in the power row.  Runs of 8 bins back to back, the two 16-byte halves of a run swapped where bit 7 of the
// bin is set: the butterfly-order writes (lanes = 32 consecutive bins, ds_write_b32) then stay within a handful of distinct banks per
// lane group, and the run reads (ds_read_b128, thread tf takes run tf: sixteen lanes 32 bytes apart) find the halves of neighbouring
// lane groups on disjoint bank quarters.  scripts/lds_model.py: 36 + 16 LDS cycles per frame for the 16 writes + 4 reads against 64 + 16
// for the round-3 layout (runs 48 bytes apart, every 32-lane write 2-way conflicted).  Linear over steps of 256 bins: idx(k + 256) = idx(k) + 256.
#ifndef LRA_MEL_PW_SWZ
#define LRA_MEL_PW_SWZ 1
#endif
// (the swizzled form needs pair slots a whole number of 128-bin blocks apart: n_fft >= 2048 at sixteen points per thread; smaller frames keep the padded runs)
template <class Cfg> constexpr bool v2_pw_swz() { return LRA_MEL_PW_SWZ && (2 * Cfg::TF) % 128 == 0; }
template <bool SWZ> LRA_HD int v2_pw_index_t(int k) {
    if (SWZ) return (k >> 3) * 8 + ((((k >> 2) ^ (k >> 7)) & 1) << 2) + (k & 3);
    return (k >> 3) * 12 + (k & 7);
}
template <class Cfg> LRA_HD int v2_pw_index(int k) { return v2_pw_index_t<v2_pw_swz<Cfg>()>(k); }
template <class Cfg> constexpr int v2_pw_bytes() { return (v2_pw_swz<Cfg>() ? Cfg::M + 8 : (Cfg::M >> 3) * 12 + 4) * (int)sizeof(typename Cfg::real); }

// phase: last-pass butterflies, Hermitian split in registers, epilogue (complex spectrum or |X|^power) to HBM
template <class Cfg, int HD, int MODE, int PM, bool STAGED, bool PIN = true>
LRA_HD void v2_last_split_store(const StftArgs<typename Cfg::real>& a, int clip, int frame, bool valid, int tf, Regs2<Cfg, HD>& rg, Lds stage) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int p = Cfg::P - 1, r = 1 << Cfg::logr(p), s = 2 * Cfg::TF, M = Cfg::M;
    static_assert(r * s == M, "last pass: r butterfly outputs s apart");
    pass_twiddle_dft_reg<Cfg, p>(rg.v, rg.treg);
    const C* A = rg.v;       // A[j] = Z[tf + j s]          (lane 0: Z[j s])
    const C* B = rg.v + r;   // B[j] = Z[(s - tf) + j s]    (lane 0: Z[s/2 + j s])
    const bool l0 = tf == 0;
    const LaneMask l0m = lane_mask(l0);
    const int tfh = v2_tf_hi<Cfg>(tf);
    const long long row = ((long long)clip * a.n_frames + frame) * a.row_pitch;
    C* __restrict__ const D = MODE == OUT_COMPLEX ? a.D + row : nullptr;
    T* __restrict__ const S = MODE == OUT_POWER ? a.S + row : nullptr;
    const int sh = (MODE == OUT_COMPLEX && STAGED) ? v2_row_shift<Cfg>(a, clip, frame) : 0;
    // OUT_MELR: byte addresses of the power-row slots of this thread's four bin families (k = tf + q s and M - k, with lane 0's
    // alternative bases), pinned in registers: left to itself hipcc re-derives (k >> 3) * 12 + (k & 7) for each of the 16 stores
    // (4 VALU instructions each) rather than keep four values live
    // ([family][parity of the pair slot]: the swizzled row is linear over steps of TWO slots (256 bins), so each family has two bases)
    int pwk[2][2] = {{0, 0}, {0, 0}}, pwm[2][2] = {{0, 0}, {0, 0}};
    if (MODE == OUT_MELR) {
        LRA_UNROLL
        for (int par = 0; par < 2; ++par) {
            pwk[0][par] = v2_pw_index<Cfg>(tf + par * s) * (int)sizeof(T);
            pwk[1][par] = v2_pw_index<Cfg>(tfh + par * s) * (int)sizeof(T);
            pwm[0][par] = v2_pw_index<Cfg>(M - tf - par * s) * (int)sizeof(T);
            pwm[1][par] = v2_pw_index<Cfg>(M - tfh - par * s) * (int)sizeof(T);
            // (PIN = false: the producer / consumer kernel, whose 168-VGPR producer spills a window pair with the pins and needs no more instructions without them)
            if (PIN) { LRA_KEEP(pwk[0][par]); LRA_KEEP(pwk[1][par]); LRA_KEEP(pwm[0][par]); LRA_KEEP(pwm[1][par]); }
        }
    }
    LRA_UNROLL
    for (int q = 0; q < r; ++q) {
        // pair slot q.  Lanes 1..: (A[q], B[r-1-q]) = (Z[k], Z[M-k]), k = tf + q s.  Lane 0: q < r/2: (A[q], A[r-q]), k = q s;
        // q >= r/2: (B[q - r/2], B[3r/2 - 1 - q]), k = s/2 + (q - r/2) s; slot 0 of lane 0 is the DC / Nyquist pair.
        const C zk = q >= r / 2 ? sel_mask(l0m, l0, B[q - r / 2], A[q]) : A[q];
        C zm;
        if (q == 0) zm = B[r - 1];
        else if (q < r / 2) zm = sel_mask(l0m, l0, A[r - q], B[r - 1 - q]);
        else zm = sel_mask(l0m, l0, B[3 * r / 2 - 1 - q], B[r - 1 - q]);
        C xk, xm;
        split_pair<T>(zk, zm, rg.twr[q], xk, xm);
        if (q == 0) {
            const C z0 = A[0];  // Z is pre-halved (see split_pair)
            const C dc = mk<T>((T)2 * (z0.x + z0.y), (T)0), ny = mk<T>((T)2 * (z0.x - z0.y), (T)0);
            xk = sel_mask(l0m, l0, dc, xk);
            xm = sel_mask(l0m, l0, ny, xm);
            if (LRA_UNLIKELY(l0 && valid && a.nonfinite_flag && !(std::fabs(dc.x) <= std::numeric_limits<T>::max()))) LRA_ATOMIC_OR(a.nonfinite_flag, 1u);
        }
        const int k = (q < r / 2 ? tf : tfh) + q * s;
        if (MODE == OUT_MELR) {
            // power row -> LDS (the frame area is free: every Z is in registers), bin k at float (k / 8) 12 + k % 8: runs of 8
            // bins 48 bytes apart, so that the 16-byte run reads of v2_mel_runs_read hit disjoint banks.  Per-thread bases + immediates.
            if (v2_pw_swz<Cfg>()) {
                lds_st<T>(stage, pwk[q < r / 2 ? 0 : 1][q & 1] + (q >> 1) * 2 * s * (int)sizeof(T), spec_power<T, PM>(xk, a.power));
                lds_st<T>(stage, pwm[q < r / 2 ? 0 : 1][q & 1] - (q >> 1) * 2 * s * (int)sizeof(T), spec_power<T, PM>(xm, a.power));
            } else {
                lds_st<T>(stage, pwk[q < r / 2 ? 0 : 1][0] + 24 * q * (s / 16) * (int)sizeof(T), spec_power<T, PM>(xk, a.power));
                lds_st<T>(stage, pwm[q < r / 2 ? 0 : 1][0] - 24 * q * (s / 16) * (int)sizeof(T), spec_power<T, PM>(xm, a.power));
            }
        } else if (MODE == OUT_COMPLEX && STAGED) {
            // the row goes to LDS in bin order (the frame area is free: every Z is in registers), shifted so that LDS and
            // global addresses agree modulo 16; v2_store_row then writes it as aligned 16-byte pieces.  Of the two bins that
            // a row of 1025 has in excess of whole 16-byte pieces one is stored here: bin 0 or bin M, both lane 0's slot 0.
            lds_st<C>(stage, sh + k * (int)sizeof(C), xk);
            lds_st<C>(stage, sh + (M - k) * (int)sizeof(C), xm);
            if (q == 0 && l0 && valid) D[sh ? 0 : M] = sh ? xk : xm;
        } else if (MODE == OUT_COMPLEX) {
#if LRA_ABLATE == 1
            if (valid && xk.x == (T)12345.678) { stream_store(&D[k], xk); stream_store(&D[M - k], xm); }
#else
            if (valid) { stream_store(&D[k], xk); stream_store(&D[M - k], xm); }
#endif
        } else {
            const T pk = spec_power<T, PM>(xk, a.power), pm = spec_power<T, PM>(xm, a.power);
            if (valid) { S[k] = pk; S[M - k] = pm; }
        }
    }
    // X[M/2] = conj(Z[M/2]) (Z pre-halved): lane 0's A[r/2]
    const C zmid = A[r / 2];
    const C xmid = mk<T>((T)2 * zmid.x, (T)-2 * zmid.y);
    if (MODE == OUT_MELR) {
        if (l0) lds_st<T>(stage, v2_pw_index<Cfg>(M / 2) * (int)sizeof(T), spec_power<T, PM>(xmid, a.power));
    } else if (MODE == OUT_COMPLEX && STAGED) {
        if (l0) lds_st<C>(stage, sh + (M / 2) * (int)sizeof(C), xmid);
    } else if (l0 && valid) {
        if (MODE == OUT_COMPLEX) D[M / 2] = xmid;
        else S[M / 2] = spec_power<T, PM>(xmid, a.power);
    }
}

// Staged complex epilogue, second half: the row sits in LDS (v2_last_split_store) and leaves as whole 16-byte pieces:
// 8 M bytes = M / (2 TF) store instructions of TF x 16 contiguous bytes each, starting at the row's first 16-byte
// boundary.  Measured on the bare store stream (scripts/storepat2.hip): 8-byte pieces in butterfly order 5.0 TB/s,
// aligned 16-byte pieces 5.4, and only these take non-temporal stores and the XCD-aware workgroup map well (6.2).
template <class Cfg> LRA_HD void v2_store_row(const StftArgs<typename Cfg::real>& a, int clip, int frame, bool valid, int tf, Lds stage) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int M = Cfg::M, PIECES = M / (2 * Cfg::TF);
    static_assert(sizeof(T) == 4, "16-byte pieces of complex64");
    if (!valid) return;
    const long long row = ((long long)clip * a.n_frames + frame) * a.row_pitch;
    const int sh = v2_row_shift<Cfg>(a, clip, frame);
    char* __restrict__ const g = reinterpret_cast<char*>(a.D + row) + sh + 16 * tf;  // first aligned piece of this thread
    V4<T> w[PIECES];
    LRA_UNROLL
    for (int c = 0; c < PIECES; ++c) w[c] = lds_ld<V4<T>>(stage, 2 * sh + c * 16 * Cfg::TF + 16 * tf);
    LRA_UNROLL
    for (int c = 0; c < PIECES; ++c) {
#if LRA_ABLATE == 1  // experiment: everything but the spectrum stores
        if (w[c].a != (T)12345.678) continue;
#endif
        stream_store16(reinterpret_cast<V4<T>*>(g + c * 16 * Cfg::TF), w[c]);
    }
}

// ---- OUT_MELR on the second-generation core ------------------------------------------------------------------------

// phase: this thread's runs (bins 8 tf .. 8 tf + 7 and M/2 + 8 tf ..) from the power row; thread 0 also takes bin M
template <class Cfg, int HD> LRA_HD void v2_mel_runs_read(Regs2<Cfg, HD>& rg, Lds pwr, int tf) {
    using T = typename Cfg::real;
    constexpr int BPL = Cfg::R / 2;
    static_assert(BPL == 8, "runs of 8 bins");
    LRA_UNROLL
    for (int run = 0; run < 2; ++run) {
        const int first = 8 * (run * Cfg::TF + tf);  // first bin of the run (padded layout: 12 (run TF + tf) floats, the second half 16 bytes on)
        const V4<T> lo = lds_ld<V4<T>>(pwr, v2_pw_index<Cfg>(first) * (int)sizeof(T)), hi = lds_ld<V4<T>>(pwr, v2_pw_index<Cfg>(first + 4) * (int)sizeof(T));
        T* d = rg.pw + run * BPL;
        d[0] = lo.a; d[1] = lo.b; d[2] = lo.c; d[3] = lo.d; d[4] = hi.a; d[5] = hi.b; d[6] = hi.c; d[7] = hi.d;
    }
    rg.pw_extra = lds_ld<T>(pwr, (tf == 0 ? v2_pw_index<Cfg>(Cfg::M) : 0) * (int)sizeof(T));  // consumed by thread 0 only
}

// same phase: this thread's 2 x 8 weight pairs (wA, wB) from the workgroup's table -> the butterfly registers (dead between the split
// and the next frame's window multiply), two pairs per 16-byte read.  They depend on nothing in the frame: issued here, behind the
// run reads, their latency is covered by the same wait instead of opening the accumulate phase with a second LDS round trip.
template <class Cfg, int HD> LRA_HD void v2_mel_weights_read(Regs2<Cfg, HD>& rg, Lds sh, int tf) {
    using T = typename Cfg::real;
    constexpr int BPL = Cfg::R / 2, TF = Cfg::TF;
    LRA_UNROLL
    for (int run = 0; run < 2; ++run) {
        LRA_UNROLL
        for (int j = 0; j < BPL; j += 2) {
            const V4<T> w2 = lds_ld<V4<T>>(sh, ((BPL + 2) * (run * TF + tf) + j) * 2 * (int)sizeof(T));
            rg.v[run * BPL + j] = mk<T>(w2.a, w2.b);
            rg.v[run * BPL + j + 1] = mk<T>(w2.c, w2.d);
        }
    }
}

// phase: (wA, wB) x power, running sums along both runs (restart factors in registers) -> rs[jj][tf] (as melr_split_accumulate)
template <class Cfg, int HD> LRA_HD void v2_mel_accumulate(const StftArgs<typename Cfg::real>& a, int tf, Regs2<Cfg, HD>& rg, Lds rs, Lds sh) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int BPL = Cfg::R / 2, TF = Cfg::TF;
    // The 16 weight pairs were read in the previous phase (v2_mel_weights_read: rg.v).  (Round 2 read weight j + 1 behind the store
    // of running sum j; the compiler must keep that order -- it cannot see that `sh` and `rs` never overlap -- and the frame ran 16
    // dependent LDS round trips here, each behind an s_waitcnt lgkmcnt(0): 0.664 -> 0.635 ms for the 256 x 30 s batch once batched.)
    const C* w = rg.v;
    LRA_UNROLL
    for (int run = 0; run < 2; ++run) {
        C acc = mk<T>((T)0, (T)0);
        LRA_UNROLL
        for (int j = 0; j < BPL; ++j) {
            const int jj = run * BPL + j;
            const T p = rg.pw[jj], keep = rg.keep[jj];
            acc = mk<T>(acc.x * keep + w[jj].x * p, acc.y * keep + w[jj].y * p);
            lds_st<C>(rs, (jj * mel_runs_pitch(TF, 1) + tf) * (int)sizeof(C), acc);  // (pitch: lra_mel.h, mel_runs_pitch)
        }
    }
    if (tf == 0) {
        const C wq = lds_ld<C>(sh, melr_w_slot<Cfg>(Cfg::M) * (int)sizeof(C));
        lds_st<C>(rs, a.melr_mid, mk<T>(wq.x * rg.pw_extra, wq.y * rg.pw_extra));
        lds_st<C>(rs, a.melr_zero, mk<T>((T)0, (T)0));
    }
}


// ---- third form: radices 16, 16, 4 -- adjacent bins side by side in one thread (round 6) ---------------------------------------
// FftCfg<..., PLAN = 1> at M = 1024.  The last pass has s = 256 butterflies of radix 4, four per thread: thread t takes
//     E = 2t,   O = 2t + 1,   Om = s - 1 - 2t (the mirror of O),   Em = s - 2t (the mirror of E; thread 0: s / 2, the other self-mirrored one),
// outputs of butterfly b at b + j s.  Every mirrored pair (Z[k], Z[M - k]) is still in ONE thread -- (E[j], Em[3 - j]) is k = 2t + j s and
// (O[j], Om[3 - j]) is k = 2t + 1 + j s -- AND bins 2t, 2t + 1 (+ j s) are neighbours, as are M - 2t - 1, M - 2t (- j s): the spectrum row leaves as
// 4 + 4 global_store_dwordx4 per thread, a wave instruction covering 1 KiB of the row, with no lane exchange and no staging (the lane-pair
// exchange of round 5 cost ~85 instructions per frame, profiles/r05_pitch.md section 4).  Against the two-butterfly form: 22 instead of 48 lane-0
// selects, 8 instead of 16 row stores; the transform's own arithmetic is the same within two instructions (42 twiddle products against 44,
// 114 butterfly instructions against 112).
// LDS hand-over middle pass -> last pass: element i at slot i + 16 (i >> 8) (blocks of 256 + 16 slots): the middle pass's stores (lanes 16 a + k
// -> 256 a + k + 16 j) are conflict-free, the last pass reads (E, O)[j] as ONE aligned 16-byte piece at 16 t + 2176 j, Om / Em as 8-byte reads.
#ifndef LRA_V3_DERIVE_TWR
#define LRA_V3_DERIVE_TWR 1
#endif
template <class Cfg> constexpr bool v3_cfg_ok() {
    return Cfg::PLAN == 1 && Cfg::P == 3 && Cfg::R == 16 && Cfg::HOIST && sizeof(typename Cfg::real) == 4 && Cfg::logr(0) == 4 && Cfg::logr(1) == 4 && Cfg::logr(2) == 2 &&
           Cfg::TF == 64 && affine_tf<Cfg>();
}
template <class Cfg> constexpr bool v23_cfg_ok() { return v2_cfg_ok<Cfg>() || v3_cfg_ok<Cfg>(); }
constexpr int V3_BLOCK = 256 + 16;  // slots per block of 256 elements in the middle -> last hand-over
// last-pass butterfly i of thread tf (i = 0: E, 1: O, 2: Om, 3: Em)
template <class Cfg> LRA_HD int v3_bfly(int tf, int i) {
    constexpr int s = 4 * Cfg::TF;
    return i == 0 ? 2 * tf : (i == 1 ? 2 * tf + 1 : (i == 2 ? s - 1 - 2 * tf : (tf == 0 ? s / 2 : s - 2 * tf)));
}
// bin of pair slot q: q < 4: the E slots (k = 2 tf + q s; thread 0's slots 2 and 3 hold butterfly s/2's pairs, bins s/2 and s/2 + s), q >= 4: the O slots
template <class Cfg> LRA_HD int v3_slot_bin(int tf, int q) {
    constexpr int s = 4 * Cfg::TF;
    if (q >= 4) return 2 * tf + 1 + (q - 4) * s;
    if (tf == 0 && q >= 2) return s / 2 + (q - 2) * s;
    return 2 * tf + q * s;
}
template <class Cfg, int HD> LRA_HD void v3_hoist(Regs2<Cfg, HD>& rg, int tf, const typename Cfg::real* __restrict__ win, const typename Cfg::cplx* __restrict__ tw,
                                                   const typename Cfg::cplx* __restrict__ twr_full) {
    using C = typename Cfg::cplx;
    using RG = Regs2<Cfg, HD>;
    const C* __restrict__ win2 = reinterpret_cast<const C*>(win);
    LRA_UNROLL
    for (int e = 0; e < Cfg::R; ++e) rg.win2[e] = win2[RG::q_of(tf, e)];
    load_pass_twiddles<Cfg, 1>(rg.treg, tf, tw);
    {
        constexpr int p = 2, lr = 2, s = 1 << Cfg::logs(p);
        LRA_UNROLL
        for (int i = 0; i < 4; ++i) {
            const int k = v3_bfly<Cfg>(tf, i);
            LRA_UNROLL
            for (int t = 0; t < lr; ++t) rg.treg[Cfg::treg_off(p) + i * lr + t] = tw[Cfg::tw_off(p) + ((1 << t) - 1) * s + k];
        }
    }
    // (the O slots' twiddles W_N^(2 tf + 1 + j s), j = 1 .. 3, are W_N^(2 tf + 1) times e^(-i pi j / 4): three or four instructions per frame in
    // v3_last_split_store instead of six registers -- with all eight held the complex kernel needs 174 VGPRs, six more than three waves per SIMD allow)
    LRA_UNROLL
    for (int q = 0; q < (LRA_V3_DERIVE_TWR ? 5 : 8); ++q) rg.twr[q] = twr_full[v3_slot_bin<Cfg>(tf, q)];
}
// middle pass twiddles, register-lean.  pass_twiddle_dft_reg derives all fifteen W^(k j) from the four stored powers first (w[16] + q[16] live next to
// v[16]: the complex kernel then needs ~206 VGPRs and spills 38 under the three-wave budget).  Here W^(k (8 + j)) v = W^(k j) (W^(8 k) v): the upper eight
// inputs take W^(8 k) first, then each W^(k j), j = 1 .. 7, multiplies v[j] AND v[8 + j] and is dead -- the same 26 complex products, three twiddles live.
template <class Cfg> LRA_HD void v3_mid_twiddle_dft(typename Cfg::cplx* v, const typename Cfg::cplx* treg) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int o = Cfg::treg_off(1);
    const C w1 = treg[o], w2 = treg[o + 1], w4 = treg[o + 2], w8 = treg[o + 3];
    LRA_UNROLL
    for (int h = 0; h < 2; ++h) {
        C q[4];
        LRA_UNROLL
        for (int j = 0; j < 4; ++j) q[j] = cmul_p(v[8 + 4 * h + j], w8);
        LRA_UNROLL
        for (int j = 0; j < 4; ++j) v[8 + 4 * h + j] = cmul_f(v[8 + 4 * h + j], w8, q[j]);
    }
    // j = 1, 2, 4, 8 are the stored powers; 3 = 1 + 2, 7 = 3 + 4, 5 = 1 + 4, 6 = 2 + 4 are derived one at a time, each started (cmul_p) alongside the
    // data products of the previous one and dead after its own two: at most two derived twiddles and five half-products live at any point
#define LRA_V3_TW2(j, w)                                                    \
    const C qa##j = cmul_p(v[j], w), qb##j = cmul_p(v[8 + j], w);
#define LRA_V3_TW2F(j, w)                                                   \
    v[j] = cmul_f(v[j], w, qa##j);                                          \
    v[8 + j] = cmul_f(v[8 + j], w, qb##j);
    const C q3 = cmul_p(w1, w2);
    LRA_V3_TW2(1, w1) LRA_V3_TW2(2, w2)
    const C w3 = cmul_f(w1, w2, q3);
    LRA_V3_TW2F(1, w1) LRA_V3_TW2F(2, w2)
    const C q7 = cmul_p(w3, w4);
    LRA_V3_TW2(3, w3) LRA_V3_TW2(4, w4)
    const C w7 = cmul_f(w3, w4, q7);
    LRA_V3_TW2F(3, w3) LRA_V3_TW2F(4, w4)
    const C q5 = cmul_p(w1, w4);
    LRA_V3_TW2(7, w7)
    const C w5 = cmul_f(w1, w4, q5);
    LRA_V3_TW2F(7, w7)
    const C q6 = cmul_p(w2, w4);
    LRA_V3_TW2(5, w5)
    const C w6 = cmul_f(w2, w4, q6);
    LRA_V3_TW2F(5, w5)
    LRA_V3_TW2(6, w6)
    LRA_V3_TW2F(6, w6)
#undef LRA_V3_TW2
#undef LRA_V3_TW2F
    Dft<16, T>::run(v);
}
// middle pass (radix 16, one butterfly per thread): output j of butterfly tf = 16 a + k -> element 256 a + k + 16 j
template <class Cfg> LRA_HD void v3_mid_write(const typename Cfg::cplx* v, Lds fr, int tf) {
    using C = typename Cfg::cplx;
    const int base = (V3_BLOCK * (tf >> 4) + (tf & 15)) * (int)sizeof(C);
    LRA_UNROLL
    for (int j = 0; j < 16; ++j) lds_st<C>(fr, base + j * 16 * (int)sizeof(C), v[j]);
}
// inputs of the four last-pass butterflies: v[4 i + j] = input j of butterfly i (element b_i + 256 j)
template <class Cfg, int HD> LRA_HD void v3_last_read(Regs2<Cfg, HD>& rg, Lds fr, int tf) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int BB = V3_BLOCK * (int)sizeof(C);
    LRA_UNROLL
    for (int j = 0; j < 4; ++j) {
        const V4<T> eo = lds_ld<V4<T>>(fr, 2 * (int)sizeof(C) * tf + j * BB);
        rg.v[j] = mk<T>(eo.a, eo.b);
        rg.v[4 + j] = mk<T>(eo.c, eo.d);
    }
    const int bo = v3_bfly<Cfg>(tf, 2) * (int)sizeof(C), be = v3_bfly<Cfg>(tf, 3) * (int)sizeof(C);
    LRA_UNROLL
    for (int j = 0; j < 4; ++j) {
        rg.v[8 + j] = lds_ld<C>(fr, bo + j * BB);
        rg.v[12 + j] = lds_ld<C>(fr, be + j * BB);
    }
}
// two neighbouring elements as one store (global_store_dwordx4 / dwordx2: the pointer need only be element-aligned)
#ifndef LRA_V3_NT
#define LRA_V3_NT 0  // non-temporal 16-byte row pieces (experiment)
#endif
template <class V> LRA_HD void store_pair(V* p, V lo, V hi) {
#if !defined(LRA_HOSTSIM)
    if constexpr (sizeof(V) == 8) {
        typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));
        const cx<float> l = __builtin_bit_cast(cx<float>, lo), h = __builtin_bit_cast(cx<float>, hi);
        f4u v;
        v.x = l.x; v.y = l.y; v.z = h.x; v.w = h.y;
#if LRA_V3_NT
        __builtin_nontemporal_store(v, reinterpret_cast<f4u*>(p));
#else
        *reinterpret_cast<f4u*>(p) = v;
#endif
        return;
    } else if constexpr (sizeof(V) == 4) {
        typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
        f2u v;
        v.x = __builtin_bit_cast(float, lo); v.y = __builtin_bit_cast(float, hi);
        *reinterpret_cast<f2u*>(p) = v;
        return;
    }
#endif
    p[0] = lo;
    p[1] = hi;
}
// phase: last-pass butterflies, Hermitian split in registers, epilogue
template <class Cfg, int HD, int MODE, int PM, bool PIN = true>
LRA_HD void v3_last_split_store(const StftArgs<typename Cfg::real>& a, int clip, int frame, bool valid, int tf, Regs2<Cfg, HD>& rg, Lds stage) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int M = Cfg::M, s = 4 * Cfg::TF;
    static_assert(4 * s == M, "last pass: four butterfly outputs s apart");
    pass_twiddle_dft_reg<Cfg, 2>(rg.v, rg.treg);
    const C* E = rg.v;        // E[j]  = Z[2 tf + j s]            (thread 0: Z[j s])
    const C* O = rg.v + 4;    // O[j]  = Z[2 tf + 1 + j s]
    const C* Om = rg.v + 8;   // Om[j] = Z[s - 1 - 2 tf + j s]
    const C* Em = rg.v + 12;  // Em[j] = Z[s - 2 tf + j s]        (thread 0: Z[s / 2 + j s])
    const bool l0 = tf == 0;
    const LaneMask l0m = lane_mask(l0);
    const C xmid = mk<T>((T)2 * E[2].x, (T)-2 * E[2].y);  // X[M/2] = conj(Z[M/2]) (Z pre-halved): thread 0's E[2]
    const long long row = ((long long)clip * a.n_frames + frame) * a.row_pitch;
    C* __restrict__ const D = MODE == OUT_COMPLEX ? a.D + row : nullptr;
    T* __restrict__ const S = MODE == OUT_POWER ? a.S + row : nullptr;
    int ia = 0, im1 = 0, im0 = 0;
    if (MODE == OUT_MELR) {
        // the power row (v2_pw_index: linear over steps of 256 bins, neighbours 2 t, 2 t + 1 in one 8-byte slot)
        ia = v2_pw_index<Cfg>(2 * tf) * (int)sizeof(T); im1 = v2_pw_index<Cfg>(M - 1 - 2 * tf) * (int)sizeof(T); im0 = v2_pw_index<Cfg>(M - 2 * tf) * (int)sizeof(T);
        if (PIN) { LRA_KEEP(ia); LRA_KEEP(im1); LRA_KEEP(im0); }
    }
    // Row pieces.  Ascending piece j = bins (2 tf + j s, + 1) = (X of E slot j, X of O slot j); mirrored piece j = bins (M - 2 tf - 1 - j s, + 1) =
    // (X' of O slot j, X' of E slot j).  Thread 0's E slots: 0 is the DC / Nyquist pair; 1 pairs butterfly 0 with itself (Z[s], Z[3 s]); 2 and 3 take
    // butterfly s/2's pairs (Z[s/2], Z[s/2 + 3 s]) and (Z[s/2 + s], Z[s/2 + 2 s]), whose four bins leave as single stores, while the E halves of its
    // pieces 2 and 3 carry X[2 s] = xmid and X[3 s] / X[s] = slot 1's results -- bins s, 2 s, 3 s are written twice (ascending and mirrored piece) from
    // the SAME register.  Slot 1 goes first because pieces 3 need it.
    C x1k = mk<T>((T)0, (T)0), x1m = x1k;
    LRA_UNROLL
    for (int jo = 0; jo < 4; ++jo) {
        const int j = jo == 0 ? 1 : (jo == 1 ? 0 : jo);  // 1, 0, 2, 3
        C xkO, xmO, xkE, xmE;
        C wO = rg.twr[LRA_V3_DERIVE_TWR ? 4 : 4 + j];
        if (LRA_V3_DERIVE_TWR && j > 0) {
            const T h = (T)0.70710678118654752440;
            const C t = rg.twr[4];
            if (j == 1) { const C u = add_mi(t, t); wO = mk<T>(h * u.x, h * u.y); }          // e^(-i pi / 4) = h (1 - i)
            else if (j == 2) wO = cmul_mi(t);                                               // e^(-i pi / 2) = -i
            else { const C u = sub_mi(t, t); wO = mk<T>(-h * u.x, -h * u.y); }              // e^(-3 i pi / 4) = -h (1 + i)
        }
        split_pair<T>(O[j], Om[3 - j], wO, xkO, xmO);
        C zk = E[j], zm = Em[3 - j];
        if (j == 1) zm = sel_mask(l0m, l0, E[3], Em[2]);
        if (j == 2) { zk = sel_mask(l0m, l0, Em[0], E[2]); zm = sel_mask(l0m, l0, Em[3], Em[1]); }
        if (j == 3) { zk = sel_mask(l0m, l0, Em[1], E[3]); zm = sel_mask(l0m, l0, Em[2], Em[0]); }
        split_pair<T>(zk, zm, rg.twr[j], xkE, xmE);
        if (j == 0) {
            const C z0 = E[0];  // Z is pre-halved (see split_pair)
            const C dc = mk<T>((T)2 * (z0.x + z0.y), (T)0), ny = mk<T>((T)2 * (z0.x - z0.y), (T)0);
            xkE = sel_mask(l0m, l0, dc, xkE);
            xmE = sel_mask(l0m, l0, ny, xmE);
            if (LRA_UNLIKELY(l0 && valid && a.nonfinite_flag && !(std::fabs(dc.x) <= std::numeric_limits<T>::max()))) LRA_ATOMIC_OR(a.nonfinite_flag, 1u);
        }
        if (j == 1) { x1k = xkE; x1m = xmE; }
        C ascE = xkE, mirE = xmE;  // the E halves of the two pieces
        if (j == 2) { ascE = sel_mask(l0m, l0, xmid, xkE); mirE = sel_mask(l0m, l0, xmid, xmE); }
        if (j == 3) { ascE = sel_mask(l0m, l0, x1m, xkE); mirE = sel_mask(l0m, l0, x1k, xmE); }
        if (MODE == OUT_COMPLEX) {
            if (valid) {
                store_pair<C>(D + 2 * tf + j * s, ascE, xkO);
                store_pair<C>(D + (M - 1 - 2 * tf) - j * s, xmO, mirE);
                if (j >= 2 && l0) { D[s / 2 + (j - 2) * s] = xkE; D[M - s / 2 - (j - 2) * s] = xmE; }
            }
        } else if (MODE == OUT_POWER) {
            const T pa0 = spec_power<T, PM>(ascE, a.power), pa1 = spec_power<T, PM>(xkO, a.power), pm0 = spec_power<T, PM>(xmO, a.power), pm1 = spec_power<T, PM>(mirE, a.power);
            T q0 = (T)0, q1 = (T)0;
            if (j >= 2) { q0 = spec_power<T, PM>(xkE, a.power); q1 = spec_power<T, PM>(xmE, a.power); }
            if (valid) {
                store_pair<T>(S + 2 * tf + j * s, pa0, pa1);
                store_pair<T>(S + (M - 1 - 2 * tf) - j * s, pm0, pm1);
                if (j >= 2 && l0) { S[s / 2 + (j - 2) * s] = q0; S[M - s / 2 - (j - 2) * s] = q1; }
            }
        } else {
            static_assert(MODE != OUT_MELR || v2_pw_swz<Cfg>(), "swizzled power row");
            lds_st<C>(stage, ia + j * s * (int)sizeof(T), mk<T>(spec_power<T, PM>(ascE, a.power), spec_power<T, PM>(xkO, a.power)));
            lds_st<T>(stage, im1 - j * s * (int)sizeof(T), spec_power<T, PM>(xmO, a.power));
            lds_st<T>(stage, im0 - j * s * (int)sizeof(T), spec_power<T, PM>(mirE, a.power));
            if (j >= 2 && l0) {
                lds_st<T>(stage, v2_pw_index<Cfg>(s / 2 + (j - 2) * s) * (int)sizeof(T), spec_power<T, PM>(xkE, a.power));
                lds_st<T>(stage, v2_pw_index<Cfg>(M - s / 2 - (j - 2) * s) * (int)sizeof(T), spec_power<T, PM>(xmE, a.power));
            }
        }
    }
}

// bytes of LDS one frame slot needs: the frame area only
template <class Cfg> constexpr int stft2_slot_bytes() { return Cfg::FRAME_BYTES; }

// One workgroup = FPB frame slots, slot s transforms frames f_first + s iters + it, it = 0 .. iters-1 (as stft_block).
template <class Cfg, int HD, int MODE, int PM = POW_TWO> LRA_HD void stft_block2(const StftArgs<typename Cfg::real>& a_in, const int blk, Lds lds) {
    static_assert(v23_cfg_ok<Cfg>(), "configuration has no mirrored last pass");
    static_assert(MODE == OUT_COMPLEX || MODE == OUT_POWER || MODE == OUT_MELR, "epilogues of the second-generation kernel");
    static_assert(MODE != OUT_MELR || (melr_fits<Cfg>() && v2_pw_bytes<Cfg>() <= Cfg::FRAME_BYTES && (Cfg::R * (Cfg::TF + LRA_MEL_RS_PITCH_EXTRA) + 2) * 2 * (int)sizeof(typename Cfg::real) <= Cfg::FRAME_BYTES),
                  "mel epilogue: running sums (pitch TF + 1) and power row live in the frame area");
    StftArgs<typename Cfg::real> a = a_in;
    using RG = Regs2<Cfg, HD>;
    const int clip = blk / a.wg_per_clip;
    const int f_first = (blk % a.wg_per_clip) * a.frames_per_wg;
    const int iters = a.frames_per_wg / Cfg::FPB;
    constexpr int SB = stft2_slot_bytes<Cfg>();
    // Complex epilogue form.  0 (product): 8-byte stores straight from the butterfly registers.  1 (experiment): the row is
    // staged in LDS and leaves as aligned 16-byte pieces (v2_store_row).  On the bare store stream aligned pieces are up to
    // 6 % faster, but a kernel with 12 resident waves per CU is bound by the bytes it keeps in flight, not by the piece
    // size: measured 0.66 ms direct vs 0.72-0.74 ms staged (with or without non-temporal stores) for the 256 x 30 s batch,
    // the staging round trip costing more LDS time than the wider stores save (profiles/r02_store_stream.md).
#ifndef LRA_V2_STAGED
#define LRA_V2_STAGED 0
#endif
    constexpr bool STAGED = Cfg::PLAN == 0 && MODE == OUT_COMPLEX && LRA_V2_STAGED && sizeof(typename Cfg::real) == 4 && (Cfg::M + 3) * (int)sizeof(typename Cfg::cplx) <= SB;
    LRA_REGS(RG, rg, Cfg::NT);
    LRA_PHASE(Cfg::NT, tid) {
        const int slot = slot_of<Cfg>(tid), tf = lane_of<Cfg>(tid);
        if constexpr (Cfg::PLAN == 1) v3_hoist<Cfg, HD>(LRA_R(rg), tf, a.win, a.tw, a.twr);
        else v2_hoist<Cfg, HD>(LRA_R(rg), tf, a.win, a.tw, a.twr);
        v2_fill<Cfg, HD>(a, clip, f_first + slot * iters, tf, LRA_R(rg));
        if constexpr (v2_rotate_asm_ok<Cfg, HD>() && MODE != OUT_MELR) {
            // the fill's loads are waited for HERE, once: the frame loop's first consumer of the ring is an asm block whose operands are the ring
            // itself, and a wait placed there is a wait in every iteration (s_waitcnt vmcnt(0) at the loop's top: for the frame's fresh stores too)
            LRA_UNROLL
            for (int e = 0; e < Cfg::R; ++e) { LRA_KEEP(LRA_R(rg).raw[e].x); LRA_KEEP(LRA_R(rg).raw[e].y); }
        }
        // (v2_issue_loads returns early for frames past the clip: the prefetch registers then keep these zeros or an earlier frame's samples)
        LRA_UNROLL
        for (int e = 0; e < RG::NEW; ++e) LRA_R(rg).pf[e] = mk<typename Cfg::real>((typename Cfg::real)0, (typename Cfg::real)0);
        if (MODE == OUT_MELR) {
            melr_tables_to_lds<Cfg>(a, tid, lds_sub(lds, a.shared_off));
            melr_hoist<Cfg>(a, tf, LRA_R(rg), slot * SB);  // (addresses relative to the workgroup's LDS, not to the slot)
        }
    } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC && MODE != OUT_MELR)  // the shared mel tables need a workgroup barrier, once
#if defined(LRA_V2_STAGGER) && !defined(LRA_HOSTSIM)
    // experiment (round 5): the two waves that share a SIMD start LRA_V2_STAGGER x 64 cycles apart -- the wave in an odd
    // hardware wave slot sleeps before its first frame (HW_ID bits 3:0 = wave slot within the SIMD) -- instead of drifting there over the first frames
    if (MODE == OUT_MELR || LRA_V2_STAGGER_ALL) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        if (hw & 1u) __builtin_amdgcn_s_sleep(LRA_V2_STAGGER);
    }
#endif
    for (int it = 0; it < iters; ++it) {
        if (f_first + it >= a.n_frames) break;  // slot 0 has the smallest frame index: uniform exit
        // Wave priority per phase (s_setprio; LRA_V2_PRIO_A: window + transform passes, _S: un-split + stores / power row, _B: mel epilogue;
        // -1 = leave alone).  profiles/r04_experiments.md 10.
        v2_setprio<MODE == OUT_MELR ? LRA_V2_PRIO_A : (MODE == OUT_POWER ? LRA_V2_PRIO_PA : LRA_V2_PRIO_CA)>();
        LRA_PHASE(Cfg::NT, tid) {
            const int slot = slot_of<Cfg>(tid), tf = lane_of<Cfg>(tid), frame = f_first + slot * iters + it;
            RG& r = LRA_R(rg);
            // (measured, profiles/r04_experiments.md: 21 vector instructions per frame fewer and the same time for the complex / power
            // epilogues; 1 % SLOWER with the mel epilogue, whose issue-bound loop pays for the switches' taken branches -- it keeps the shift)
            if (v2_rotate_asm_ok<Cfg, HD>() && (MODE != OUT_MELR || LRA_V2_ROTATE_MEL)) {
                v2_window_rotating<Cfg, HD>(it, r);                                      // consumes the pairs loaded during the previous frame (4 moves, not 16)
                if (it + 1 < iters) v2_issue_loads<Cfg, HD>(a, clip, frame + 1, tf, r);  // ... and starts the next batch, BEFORE this frame's stores
                v2_pass0_dft<Cfg, HD>(tf, r, lds_sub(lds, slot * SB));
            } else {
                if (it > 0) v2_shift<Cfg, HD>(r);                                       // consumes the pairs loaded during the previous frame
                if (it + 1 < iters) v2_issue_loads<Cfg, HD>(a, clip, frame + 1, tf, r);  // ... and starts the next batch, BEFORE this frame's stores
                v2_pass0<Cfg, HD>(frame < a.n_frames, tf, r, lds_sub(lds, slot * SB));
            }
        } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
        if (MODE == OUT_MELR) v2_setprio<LRA_V2_PRIO_A1>();   // (middle + last passes apart from window / pass 0)
#define LRA_MID_PASS2(p)                                                                                                  \
        if (Cfg::P - 1 > p) {                                                                                             \
            LRA_PHASE(Cfg::NT, tid) {                                                                                     \
                pass_read<Cfg, (p < Cfg::P ? p : 0)>(LRA_R(rg).v, lds_sub(lds, slot_of<Cfg>(tid) * SB), lane_of<Cfg>(tid)); \
            } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)                                                                          \
            LRA_PHASE(Cfg::NT, tid) {                                                                                     \
                if constexpr (Cfg::PLAN == 1) {                                                                           \
                    v3_mid_twiddle_dft<Cfg>(LRA_R(rg).v, LRA_R(rg).treg);                                                 \
                    v3_mid_write<Cfg>(LRA_R(rg).v, lds_sub(lds, slot_of<Cfg>(tid) * SB), lane_of<Cfg>(tid));              \
                } else {                                                                                                  \
                    pass_twiddle_dft_reg<Cfg, (p < Cfg::P ? p : 0)>(LRA_R(rg).v, LRA_R(rg).treg);                         \
                    pass_write<Cfg, (p < Cfg::P ? p : 0)>(LRA_R(rg).v, lds_sub(lds, slot_of<Cfg>(tid) * SB), lane_of<Cfg>(tid)); \
                }                                                                                                         \
            } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)                                                                          \
        }
        LRA_MID_PASS2(1)
        LRA_MID_PASS2(2)
#undef LRA_MID_PASS2
        LRA_PHASE(Cfg::NT, tid) {
            if constexpr (Cfg::PLAN == 1) v3_last_read<Cfg, HD>(LRA_R(rg), lds_sub(lds, slot_of<Cfg>(tid) * SB), lane_of<Cfg>(tid));
            else v2_last_read<Cfg, HD>(LRA_R(rg), lds_sub(lds, slot_of<Cfg>(tid) * SB), lane_of<Cfg>(tid));
        } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC || (LRA_SPLIT_ONE_BARRIER && MODE != OUT_MELR && !STAGED))  // (registers -> HBM next: see stft_block)
        v2_setprio<MODE == OUT_MELR ? LRA_V2_PRIO_S : (MODE == OUT_POWER ? LRA_V2_PRIO_PS : LRA_V2_PRIO_CS)>();
        LRA_PHASE(Cfg::NT, tid) {
            const int slot = slot_of<Cfg>(tid), tf = lane_of<Cfg>(tid), frame = f_first + slot * iters + it;
            if constexpr (Cfg::PLAN == 1) v3_last_split_store<Cfg, HD, MODE, PM>(a, clip, frame, frame < a.n_frames, tf, LRA_R(rg), lds_sub(lds, slot * SB));
            else v2_last_split_store<Cfg, HD, MODE, PM, STAGED>(a, clip, frame, frame < a.n_frames, tf, LRA_R(rg), lds_sub(lds, slot * SB));
        } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
        if (STAGED) {
            LRA_PHASE(Cfg::NT, tid) {
                const int slot = slot_of<Cfg>(tid), tf = lane_of<Cfg>(tid), frame = f_first + slot * iters + it;
                v2_store_row<Cfg>(a, clip, frame, frame < a.n_frames, tf, lds_sub(lds, slot * SB));
            } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
        }
        if (MODE == OUT_MELR) {
            v2_setprio<LRA_V2_PRIO_B>();
            // power row -> runs in registers; then (only then: the running sums reuse the row's bytes) weights, running sums
            // -> rs; then every band adds its piece totals (melr_combine, shared with the first-generation kernel)
            // LRA_MEL_ABLATE = n (timing experiments, scripts/ab_run.sh): the last n of the three epilogue phases sit behind a
            // condition that is never true at run time, so the code, its registers and everything upstream stay as they are
#ifndef LRA_MEL_ABLATE
#define LRA_MEL_ABLATE 0
#endif
            const bool ablate_never = LRA_MEL_ABLATE > 0 && a.power == (typename Cfg::real)12345.678;
            if (LRA_MEL_ABLATE < 3 || ablate_never) {
            LRA_PHASE(Cfg::NT, tid) {
                v2_mel_runs_read<Cfg, HD>(LRA_R(rg), lds_sub(lds, slot_of<Cfg>(tid) * SB), lane_of<Cfg>(tid));
                v2_mel_weights_read<Cfg, HD>(LRA_R(rg), lds_sub(lds, a.shared_off), lane_of<Cfg>(tid));
            } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
            }
            v2_setprio<LRA_V2_PRIO_B2>();
            if (LRA_MEL_ABLATE < 2 || ablate_never) {
            LRA_PHASE(Cfg::NT, tid) {
                v2_mel_accumulate<Cfg, HD>(a, lane_of<Cfg>(tid), LRA_R(rg), lds_sub(lds, slot_of<Cfg>(tid) * SB), lds_sub(lds, a.shared_off));
            } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
            }
            v2_setprio<LRA_V2_PRIO_B3>();
            if (LRA_MEL_ABLATE < 1 || ablate_never) {
            LRA_PHASE(Cfg::NT, tid) {
                const int slot = slot_of<Cfg>(tid), tf = lane_of<Cfg>(tid), frame = f_first + slot * iters + it;
                const Lds sl = lds_sub(lds, slot * SB);
                if (frame < a.n_frames)
                    melr_combine<Cfg>(a, clip, frame, tf, it, 1, it + 1 == iters || frame + 1 >= a.n_frames, LRA_R(rg), lds_sub(lds, a.shared_off), sl, sl, lds);
            } LRA_PHASE_END_SYNC(Cfg::WAVE_SYNC)
            }
        }
    }
}

}  // namespace lra
