// lra_kernels_pc.h -- producer / consumer form of the fused mel kernel (round 6).
// Reference semantics: librosa/feature/spectral.py:2158-2160 (mel_basis . |stft|^power), core/spectrum.py:380-390, :3000-3013.
//
// Why.  stft2_kernel<OUT_MELR> keeps the transform's state (butterfly registers, sample ring, window, twiddles: ~150 VGPRs) AND the mel
// epilogue's (restart factors, piece addresses, the eight-frame output tile of two bands, the power runs: ~90) live in every wave: 240 VGPRs, two
// waves per SIMD, and the vector pipe idle 39 % of the time with both of them stalled.  Every attempt to fit both states under the
// 168-VGPR budget of three waves paid for it with per-frame table re-reads and lost.  Here the two states live in DIFFERENT waves:
//
//   workgroup = 192 threads = [P, P, C].  A producer wave runs window + FFT + un-split + |X|^p exactly as the complex kernel does (its register
//   budget: 168) and writes the power row -- the bytes the epilogue already read from LDS -- into a row buffer; the consumer wave serves both
//   producers' rows with the run-ordered two-slope accumulate, the band combine and the aligned eight-frame bursts (lra_mel.h, lra_kernels.h:
//   the same arithmetic in the same order as stft2_kernel<OUT_MELR>, so the two kernels agree bit for bit).  The consumer keeps the (wA, wB)
//   pairs of its two runs in registers (it has no transform state), so the workgroup has no shared weight table: 33.9 KB of LDS, four
//   workgroups = twelve waves per CU, three per SIMD.  scripts/wave_placement.hip (profiles/r06_raw/a_wave_placement.txt): with one round of
//   192-thread workgroups the dispatcher puts exactly one third wave (the consumer) on every SIMD in 234 of 256 CUs.
//
// Hand-over.  No s_barrier in the frame loop -- a barrier would march two producers on different SIMDs in step.  Two LDS words per producer:
//   ready[s]    rows published by producer s      (written by P_s behind its row writes, polled by C)
//   consumed[s] rows C has taken into registers   (written by C behind its run reads, read by P_s one frame later: never a wait in practice)
// A wave's DS instructions execute in order and the CU has one LDS pipeline, so "flag write issued behind the row writes" / "row reads issued
// behind the flag read" is all the ordering needed; the compiler is held to that order by wave-level fences.  Every poll loop is bounded
// (a protocol bug must produce wrong numbers in a test, not a hung GPU).
#pragma once

#include "lra_kernels2.h"

namespace lra {

#ifndef LRA_PC_PRIO_PA   // producer: window + transform passes
#define LRA_PC_PRIO_PA 3
#endif
#ifndef LRA_PC_PRIO_PS   // producer: un-split + power row
#define LRA_PC_PRIO_PS 2
#endif
#ifndef LRA_PC_PRIO_CA   // consumer: run read + accumulate
#define LRA_PC_PRIO_CA 0
#endif
#ifndef LRA_PC_PRIO_CB   // consumer: band combine + bursts
#define LRA_PC_PRIO_CB 1
#endif
#ifndef LRA_PC_ROTATE    // producer's register ring: 1 = rotating view (the complex kernel's form), 0 = shifted every frame (the mel kernel's form)
#define LRA_PC_ROTATE 1
#endif
#ifndef LRA_PC_SPIN_LIMIT
#define LRA_PC_SPIN_LIMIT (1 << 22)
#endif

template <class Cfg> struct PcLayout {
    using T = typename Cfg::real;
    static constexpr int NP = 2;                     // producer waves = frame slots per workgroup
    static constexpr int NT = (NP + 1) * Cfg::TF;    // 192 threads at one wave per frame
    static constexpr int FRAME = Cfg::FRAME_BYTES;
    static constexpr int PW = (v2_pw_bytes<Cfg>() + 15) / 16 * 16;                                                       // one power row
    static constexpr int RS = ((Cfg::R * (Cfg::TF + LRA_MEL_RS_PITCH_EXTRA) + 2) * 2 * (int)sizeof(T) + 15) / 16 * 16;     // running sums + zero slot + extra bin's slot
    static constexpr int frame_off(int s) { return s * FRAME; }
    static constexpr int pw_off(int s) { return NP * FRAME + s * PW; }
    static constexpr int rs_off() { return NP * (FRAME + PW); }
    static constexpr int flags_off() { return rs_off() + RS; }
    static constexpr int ready_off(int s) { return flags_off() + 4 * s; }
    static constexpr int consumed_off(int s) { return flags_off() + 4 * (NP + s); }
    static constexpr int BYTES = flags_off() + 16;
};

// one wave per frame, sixteen points per thread, mirrored two-butterfly last pass, swizzled power row: n_fft = 2048 in float32
template <class Cfg> constexpr bool pc_cfg_ok() { return v23_cfg_ok<Cfg>() && Cfg::TF == 64 && v2_pw_swz<Cfg>() && melr_fits<Cfg>() && sizeof(typename Cfg::real) == 4; }
// what the consumer holds in registers covers the bank: two bands per thread, four hoisted pieces per list, no table path
template <class Cfg> LRA_HD bool pc_bank_ok(int n_mels, int pmax) { return n_mels >= 1 && n_mels <= 2 * Cfg::TF && pmax <= 4; }

// (hop, power) combinations whose producer fits the 168-VGPR budget of three waves per SIMD without spilling (hipcc 7.2, -Rpass-analysis=kernel-resource-usage:
// hop = n_fft / 4: |X|^2 only -- the square root costs two registers too many, pow() six; n_fft / 8: |X| and |X|^2; n_fft / 2 keeps eight sample pairs in
// flight and spills 8-14).  Everything else stays with stft2_kernel<OUT_MELR>.
// (on the radix 16-16-4 core -- FftCfg PLAN = 1, its O-slot split twiddles derived -- |X| fits at hop = n_fft / 4 too)
template <class Cfg> LRA_HD bool pc_fits_budget(int hd, int power_mode) {
    if (power_mode != POW_TWO && power_mode != POW_ONE) return false;
    if (hd == 8) return true;
    return hd == 4 && (power_mode == POW_TWO || Cfg::PLAN == 1);
}

// consumer state
template <class Cfg> struct PcRegs {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    static constexpr int NB = 2, PH = 4, TILE = 8, NP = PcLayout<Cfg>::NP;
    C w[Cfg::R];        // (wA, wB) of this thread's two runs of eight bins (layout 1: bins 8 tf + j and M/2 + 8 tf + j)
    C wq;               // ... and of bin M (consumed by thread 0)
    T keep[Cfg::R];     // 0 where a running sum restarts
    T pw[Cfg::R], pw_extra;
    int mad[NB][2 * PH];       // byte addresses (workgroup LDS) of the piece totals of bands 2 tf and 2 tf + 1: B list, then A list
    T mt[NP][NB][TILE];        // per producer: the last eight frames of the two bands
};
// what melr_tile_slot / melr_burst (lra_kernels.h) need of a register struct: one producer's tile
template <class Cfg> struct PcTile {
    static constexpr int MELR_TILE = PcRegs<Cfg>::TILE;
    typename Cfg::real (&mt)[PcRegs<Cfg>::NB][PcRegs<Cfg>::TILE];
};

// ---- flags ---------------------------------------------------------------------------------------------------------------------
#ifdef LRA_HOSTSIM
LRA_HD int pc_flag_load(Lds l, int off) { return lds_ld<int>(l, off); }
LRA_HD void pc_flag_store(Lds l, int off, int v) { lds_st<int>(l, off, v); }
LRA_HD void pc_wait(Lds, int, int, unsigned int*) {}
LRA_HD void pc_fence() {}
#else
__device__ __forceinline__ void pc_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int pc_flag_load(Lds l, int off) { return *(const volatile __attribute__((address_space(3))) int*)(l.base + off); }
__device__ __forceinline__ void pc_flag_store(Lds l, int off, int v) { *(volatile __attribute__((address_space(3))) int*)(l.base + off) = v; }
// until flag >= want (every lane reads the same word: a broadcast; the value is wave-uniform, so the loop is scalar)
// A wait that gives up (a protocol bug, a partner wave that never came) raises bit 1 of the context's sticky device flag: the host turns that into an error
// (lra_ctx_nonfinite_read) instead of returning whatever the row buffers held.
__device__ __forceinline__ void pc_wait(Lds l, int off, int want, unsigned int* sticky) {
    int spins = 0;
    while (true) {
        const int v = LRA_UNIFORM(pc_flag_load(l, off));
        if (LRA_LIKELY(v >= want)) break;
        if (LRA_UNLIKELY(++spins > LRA_PC_SPIN_LIMIT)) {
            if (sticky && (threadIdx.x & 63) == 0) atomicOr(sticky, 2u);
            break;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    pc_fence();
}
#endif

// ---- producer --------------------------------------------------------------------------------------------------------------------
template <class Cfg, int HD> LRA_HD void pc_producer_prologue(const StftArgs<typename Cfg::real>& a, int clip, int frame0, int tf, Regs2<Cfg, HD>& rg) {
    using T = typename Cfg::real;
    if constexpr (Cfg::PLAN == 1) v3_hoist<Cfg, HD>(rg, tf, a.win, a.tw, a.twr);  // (radices 16, 16, 4: lra_kernels2.h, third form)
    else v2_hoist<Cfg, HD>(rg, tf, a.win, a.tw, a.twr);
    v2_fill<Cfg, HD>(a, clip, frame0, tf, rg);
    if constexpr (v2_rotate_asm_ok<Cfg, HD>() && LRA_PC_ROTATE) {
        LRA_UNROLL
        for (int e = 0; e < Cfg::R; ++e) { LRA_KEEP(rg.raw[e].x); LRA_KEEP(rg.raw[e].y); }  // (the fill's loads are waited for once, here: see stft_block2)
    }
    LRA_UNROLL
    for (int e = 0; e < Regs2<Cfg, HD>::NEW; ++e) rg.pf[e] = mk<T>((T)0, (T)0);
}
// window + pass-0 butterflies + first LDS write of frame `frame` (the slot's frame number `it`); starts the next frame's sample loads
template <class Cfg, int HD> LRA_HD void pc_producer_pass0(const StftArgs<typename Cfg::real>& a, int clip, int frame, int it, bool more, int tf, Regs2<Cfg, HD>& rg, Lds fr) {
    if (v2_rotate_asm_ok<Cfg, HD>() && LRA_PC_ROTATE) {
        v2_window_rotating<Cfg, HD>(it, rg);
        if (more) v2_issue_loads<Cfg, HD>(a, clip, frame + 1, tf, rg);
        v2_pass0_dft<Cfg, HD>(tf, rg, fr);
    } else {
        if (it > 0) v2_shift<Cfg, HD>(rg);
        if (more) v2_issue_loads<Cfg, HD>(a, clip, frame + 1, tf, rg);
        v2_pass0<Cfg, HD>(frame < a.n_frames, tf, rg, fr);
    }
}

// the middle pass's butterflies + LDS write, the last pass's reads, and the last pass + un-split + power row, on either transform plan
template <class Cfg, int p> LRA_HD void pc_mid_dft_write(typename Cfg::cplx* v, const typename Cfg::cplx* treg, Lds fr, int tf) {
    if constexpr (Cfg::PLAN == 1) {
        v3_mid_twiddle_dft<Cfg>(v, treg);
        v3_mid_write<Cfg>(v, fr, tf);
    } else {
        pass_twiddle_dft_reg<Cfg, p>(v, treg);
        pass_write<Cfg, p>(v, fr, tf);
    }
}
template <class Cfg, int HD> LRA_HD void pc_last_read(Regs2<Cfg, HD>& rg, Lds fr, int tf) {
    if constexpr (Cfg::PLAN == 1) v3_last_read<Cfg, HD>(rg, fr, tf);
    else v2_last_read<Cfg, HD>(rg, fr, tf);
}
template <class Cfg, int HD, int PM> LRA_HD void pc_last_power_row(const StftArgs<typename Cfg::real>& a, int clip, int frame, int tf, Regs2<Cfg, HD>& rg, Lds pwr) {
    if constexpr (Cfg::PLAN == 1) v3_last_split_store<Cfg, HD, OUT_MELR, PM, false>(a, clip, frame, frame < a.n_frames, tf, rg, pwr);
    else v2_last_split_store<Cfg, HD, OUT_MELR, PM, false, false>(a, clip, frame, frame < a.n_frames, tf, rg, pwr);
}

// ---- consumer --------------------------------------------------------------------------------------------------------------------
template <class Cfg> LRA_HD void pc_consumer_prologue(const StftArgs<typename Cfg::real>& a, int tf, PcRegs<Cfg>& rg) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    using RG = PcRegs<Cfg>;
    constexpr int BPL = Cfg::R / 2, TF = Cfg::TF;
    const C* __restrict__ w2 = reinterpret_cast<const C*>(a.melr_w);
    LRA_UNROLL
    for (int run = 0; run < 2; ++run) {
        LRA_UNROLL
        for (int j = 0; j < BPL; ++j) rg.w[run * BPL + j] = w2[BPL * (run * TF + tf) + j];
    }
    rg.wq = w2[Cfg::M];
    LRA_UNROLL
    for (int jj = 0; jj < Cfg::R; ++jj) rg.keep[jj] = a.melr_keep[jj * TF + tf];
    LRA_UNROLL
    for (int b = 0; b < RG::NB; ++b) {
        const int m = RG::NB * tf + b;
        LRA_UNROLL
        for (int h = 0; h < 2; ++h) {
            LRA_UNROLL
            for (int q = 0; q < RG::PH; ++q) {
                const int ad = PcLayout<Cfg>::rs_off() + (m < a.n_mels ? a.melr_addr[(h * a.melr_pmax + q) * a.n_mels + m] : a.melr_zero);
                rg.mad[b][h * RG::PH + q] = ad & ~(2 * (int)sizeof(T) - 1);  // (the B list addresses the B half of a pair; whole pairs are read)
            }
        }
    }
    LRA_UNROLL
    for (int s = 0; s < RG::NP; ++s) {
        LRA_UNROLL
        for (int b = 0; b < RG::NB; ++b) {
            LRA_UNROLL
            for (int k = 0; k < RG::TILE; ++k) rg.mt[s][b][k] = (T)0;
        }
    }
    rg.pw_extra = (T)0;
}
// a producer's power row -> this thread's two runs (as v2_mel_runs_read)
template <class Cfg> LRA_HD void pc_runs_read(PcRegs<Cfg>& rg, Lds pwr, int tf) {
    using T = typename Cfg::real;
    constexpr int BPL = Cfg::R / 2;
    static_assert(BPL == 8, "runs of 8 bins");
    LRA_UNROLL
    for (int run = 0; run < 2; ++run) {
        const int first = 8 * (run * Cfg::TF + tf);
        const V4<T> lo = lds_ld<V4<T>>(pwr, v2_pw_index<Cfg>(first) * (int)sizeof(T)), hi = lds_ld<V4<T>>(pwr, v2_pw_index<Cfg>(first + 4) * (int)sizeof(T));
        T* d = rg.pw + run * BPL;
        d[0] = lo.a; d[1] = lo.b; d[2] = lo.c; d[3] = lo.d; d[4] = hi.a; d[5] = hi.b; d[6] = hi.c; d[7] = hi.d;
    }
    rg.pw_extra = lds_ld<T>(pwr, (tf == 0 ? v2_pw_index<Cfg>(Cfg::M) : 0) * (int)sizeof(T));  // consumed by thread 0 only
}
// (wA, wB) x power, running sums along both runs -> rs[jj][tf] (as v2_mel_accumulate: same operations, same order)
template <class Cfg> LRA_HD void pc_accumulate(const StftArgs<typename Cfg::real>& a, int tf, PcRegs<Cfg>& rg, Lds rs) {
    using T = typename Cfg::real;
    using C = typename Cfg::cplx;
    constexpr int BPL = Cfg::R / 2, TF = Cfg::TF;
    LRA_UNROLL
    for (int run = 0; run < 2; ++run) {
        C acc = mk<T>((T)0, (T)0);
        LRA_UNROLL
        for (int j = 0; j < BPL; ++j) {
            const int jj = run * BPL + j;
            const T p = rg.pw[jj], keep = rg.keep[jj];
            acc = mk<T>(acc.x * keep + rg.w[jj].x * p, acc.y * keep + rg.w[jj].y * p);
            lds_st<C>(rs, (jj * mel_runs_pitch(TF, 1) + tf) * (int)sizeof(C), acc);
        }
    }
    if (tf == 0) {
        lds_st<C>(rs, a.melr_mid, mk<T>(rg.wq.x * rg.pw_extra, rg.wq.y * rg.pw_extra));
        lds_st<C>(rs, a.melr_zero, mk<T>((T)0, (T)0));
    }
}
// mel[m] of bands 2 tf, 2 tf + 1 for producer S's frame (as melr_combine with the register tile; `wg`: the workgroup's LDS, which rg.mad addresses)
template <class Cfg, int S> LRA_HD void pc_combine(const StftArgs<typename Cfg::real>& a, int clip, int frame, int tf, int it, bool last_of_slot, PcRegs<Cfg>& rg, Lds wg) {
    using T = typename Cfg::real;
    using RG = PcRegs<Cfg>;
    constexpr int PH = RG::PH, NB = RG::NB, MT = RG::TILE;
    T x[NB][2 * PH];
    LRA_UNROLL
    for (int b = 0; b < NB; ++b) {
        LRA_UNROLL
        for (int q = 0; q < 2 * PH; ++q) {
            const cx<T> pr = lds_ld<cx<T>>(wg, rg.mad[b][q]);
            x[b][q] = q < PH ? pr.y : pr.x;
        }
    }
    PcTile<Cfg> tile{rg.mt[S]};
    LRA_UNROLL
    for (int b = 0; b < NB; ++b) {
        const int m = NB * tf + b;
        if (m >= a.n_mels) break;
        T part[2];
        LRA_UNROLL
        for (int h = 0; h < 2; ++h) {
            T acc = (T)0;
            LRA_UNROLL
            for (int q = 0; q < PH; ++q) acc += x[b][h * PH + q];
            part[h] = acc;
        }
        const T v = part[0] + part[1];
        const long long row0 = ((long long)clip * a.n_mels + m) * a.n_frames;
        const int s8 = melr_tile_slot<Cfg, PcTile<Cfg>>(a, clip, frame, it, b, row0);
        LRA_UNROLL
        for (int k = 0; k < MT; ++k) rg.mt[S][b][k] = k == s8 ? v : rg.mt[S][b][k];
        if (last_of_slot || s8 == MT - 1) melr_burst<Cfg, PcTile<Cfg>>(a, row0, frame, s8, it, b, tile);
    }
}

// One workgroup = NP producer slots + one consumer; slot s transforms frames f_first + s iters + it, it = 0 .. iters - 1 (as stft_block2 with FPB = NP).
template <class Cfg, int HD, int PM = POW_TWO> LRA_HD void stft_pc_block(const StftArgs<typename Cfg::real>& a_in, const int blk, Lds lds) {
    static_assert(pc_cfg_ok<Cfg>(), "producer / consumer mel kernel: one wave per frame, mirrored last pass, swizzled power row");
    using T = typename Cfg::real;
    using L = PcLayout<Cfg>;
    using RG = Regs2<Cfg, HD>;
    constexpr int NP = L::NP, TF = Cfg::TF;
    StftArgs<T> a = a_in;
    const int clip = blk / a.wg_per_clip;
    const int f_first = (blk % a.wg_per_clip) * a.frames_per_wg;
    const int iters = a.frames_per_wg / NP;
    const int left = a.n_frames - f_first;        // >= 1: slot 0 has the smallest frame index
    const int n_it = left < iters ? left : iters;  // frames beyond the clip (slot 1's tail) are transformed like any other and nothing of them is stored
#ifdef LRA_HOSTSIM
    // The simulator runs the same phase bodies in program order: producers' phases, hand-over (a workgroup-level boundary stands in for the
    // flags), consumer's phases.
    LRA_REGS(RG, prg, L::NT);
    LRA_REGS(PcRegs<Cfg>, crg, L::NT);
    LRA_PHASE(L::NT, tid) {
        const int w = tid / TF, tf = tid % TF;
        if (w < NP) pc_producer_prologue<Cfg, HD>(a, clip, f_first + w * iters, tf, LRA_R(prg));
        else pc_consumer_prologue<Cfg>(a, tf, LRA_R(crg));
        if (tid < 2 * NP) pc_flag_store(lds, L::flags_off() + 4 * tid, 0);
    } LRA_PHASE_END
    for (int it = 0; it < n_it; ++it) {
        LRA_PHASE(L::NT, tid) {
            const int w = tid / TF, tf = tid % TF;
            if (w < NP) pc_producer_pass0<Cfg, HD>(a, clip, f_first + w * iters + it, it, it + 1 < iters, tf, LRA_R(prg), lds_sub(lds, L::frame_off(w)));
        } LRA_PHASE_END_SYNC(true)
#define LRA_PC_MID(p)                                                                                                                   \
        if (Cfg::P - 1 > p) {                                                                                                           \
            LRA_PHASE(L::NT, tid) {                                                                                                     \
                if (tid / TF < NP) pass_read<Cfg, (p < Cfg::P ? p : 0)>(LRA_R(prg).v, lds_sub(lds, L::frame_off(tid / TF)), tid % TF);  \
            } LRA_PHASE_END_SYNC(true)                                                                                                  \
            LRA_PHASE(L::NT, tid) {                                                                                                     \
                if (tid / TF < NP) pc_mid_dft_write<Cfg, (p < Cfg::P ? p : 0)>(LRA_R(prg).v, LRA_R(prg).treg, lds_sub(lds, L::frame_off(tid / TF)), tid % TF); \
            } LRA_PHASE_END_SYNC(true)                                                                                                  \
        }
        LRA_PC_MID(1)
        LRA_PC_MID(2)
#undef LRA_PC_MID
        LRA_PHASE(L::NT, tid) {
            if (tid / TF < NP) pc_last_read<Cfg, HD>(LRA_R(prg), lds_sub(lds, L::frame_off(tid / TF)), tid % TF);
        } LRA_PHASE_END_SYNC(true)
        LRA_PHASE(L::NT, tid) {
            const int w = tid / TF, tf = tid % TF, frame = f_first + w * iters + it;
            if (w < NP) pc_last_power_row<Cfg, HD, PM>(a, clip, frame, tf, LRA_R(prg), lds_sub(lds, L::pw_off(w)));
        } LRA_PHASE_END  // (ready[s] published, C has seen it)
        for (int s = 0; s < NP; ++s) {
            LRA_PHASE(L::NT, tid) {
                if (tid / TF == NP) pc_runs_read<Cfg>(LRA_R(crg), lds_sub(lds, L::pw_off(s)), tid % TF);
            } LRA_PHASE_END_SYNC(true)
            LRA_PHASE(L::NT, tid) {
                if (tid / TF == NP) pc_accumulate<Cfg>(a, tid % TF, LRA_R(crg), lds_sub(lds, L::rs_off()));
            } LRA_PHASE_END_SYNC(true)
            LRA_PHASE(L::NT, tid) {
                const int tf = tid % TF, frame = f_first + s * iters + it;
                if (tid / TF == NP && frame < a.n_frames) {
                    const bool last = it + 1 == iters || frame + 1 >= a.n_frames;
                    if (s == 0) pc_combine<Cfg, 0>(a, clip, frame, tf, it, last, LRA_R(crg), lds);
                    else pc_combine<Cfg, 1>(a, clip, frame, tf, it, last, LRA_R(crg), lds);
                }
            } LRA_PHASE_END_SYNC(true)
        }
        sim::state().barrier();  // (consumed[s] published, P_s has seen it)
    }
#else
    // ---- device: the three waves run their own loops
    if (threadIdx.x < 2 * NP) pc_flag_store(lds, L::flags_off() + 4 * (int)threadIdx.x, 0);
    phase_sync<false>();  // the only workgroup barrier of the kernel
    const int wave = LRA_UNIFORM((int)(threadIdx.x / TF));
    if (wave < NP) {
        RG rg;
        const Lds fr = lds_sub(lds, L::frame_off(wave)), pwr = lds_sub(lds, L::pw_off(wave));
        {
            const int tf = phase_tid() % TF;
            pc_producer_prologue<Cfg, HD>(a, clip, f_first + wave * iters, tf, rg);
        }
        pc_fence();
        for (int it = 0; it < n_it; ++it) {
            const int frame = f_first + wave * iters + it;
            v2_setprio<LRA_PC_PRIO_PA>();
            {
                const int tf = phase_tid() % TF;
                pc_producer_pass0<Cfg, HD>(a, clip, frame, it, it + 1 < iters, tf, rg, fr);
            }
            pc_fence();
#define LRA_PC_MID(p)                                                                                    \
            if (Cfg::P - 1 > p) {                                                                        \
                { const int tf = phase_tid() % TF; pass_read<Cfg, (p < Cfg::P ? p : 0)>(rg.v, fr, tf); } \
                pc_fence();                                                                              \
                { const int tf = phase_tid() % TF; pc_mid_dft_write<Cfg, (p < Cfg::P ? p : 0)>(rg.v, rg.treg, fr, tf); } \
                pc_fence();                                                                              \
            }
            LRA_PC_MID(1)
            LRA_PC_MID(2)
#undef LRA_PC_MID
            int taken;
            {
                const int tf = phase_tid() % TF;
                pc_last_read<Cfg, HD>(rg, fr, tf);
                taken = pc_flag_load(lds, L::consumed_off(wave));  // rides on the same wait as the last pass's inputs
            }
            pc_fence();
            v2_setprio<LRA_PC_PRIO_PS>();
            if (LRA_UNLIKELY(LRA_UNIFORM(taken) < it)) pc_wait(lds, L::consumed_off(wave), it, a.nonfinite_flag);  // row it - 1 still unread (it was handed over a whole frame ago)
            {
                const int tf = phase_tid() % TF;
                pc_last_power_row<Cfg, HD, PM>(a, clip, frame, tf, rg, pwr);
            }
            pc_fence();
            if (phase_tid() % TF == 0) pc_flag_store(lds, L::ready_off(wave), it + 1);  // behind the row's writes in this wave's DS queue
            pc_fence();
        }
    } else {
        PcRegs<Cfg> rg;
        const Lds rs = lds_sub(lds, L::rs_off());
        {
            const int tf = phase_tid() % TF;
            pc_consumer_prologue<Cfg>(a, tf, rg);
        }
        pc_fence();
        for (int it = 0; it < n_it; ++it) {
#define LRA_PC_SERVE(S)                                                                                                        \
            {                                                                                                                  \
                const int frame = f_first + S * iters + it;                                                                   \
                v2_setprio<LRA_PC_PRIO_CA>();                                                                                  \
                pc_wait(lds, L::ready_off(S), it + 1, a.nonfinite_flag);                                                                        \
                { const int tf = phase_tid() % TF; pc_runs_read<Cfg>(rg, lds_sub(lds, L::pw_off(S)), tf); }                    \
                pc_fence();                                                                                                    \
                if (phase_tid() % TF == 0) pc_flag_store(lds, L::consumed_off(S), it + 1); /* behind the run reads */          \
                pc_fence();                                                                                                    \
                { const int tf = phase_tid() % TF; pc_accumulate<Cfg>(a, tf, rg, rs); }                                        \
                pc_fence();                                                                                                    \
                v2_setprio<LRA_PC_PRIO_CB>();                                                                                  \
                if (frame < a.n_frames) {                                                                                      \
                    const int tf = phase_tid() % TF;                                                                           \
                    pc_combine<Cfg, S>(a, clip, frame, tf, it, it + 1 == iters || frame + 1 >= a.n_frames, rg, lds);           \
                }                                                                                                              \
                pc_fence();                                                                                                    \
            }
            LRA_PC_SERVE(0)
            LRA_PC_SERVE(1)
#undef LRA_PC_SERVE
        }
    }
#endif
}

}  // namespace lra
