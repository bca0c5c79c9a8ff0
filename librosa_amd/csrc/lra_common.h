// lra_common.h -- shared definitions for the gfx950 kernels of the STFT -> mel / ISTFT path.
//
// The kernel bodies in lra_fft.h / lra_kernels.h are written once and compiled two ways:
//   * by hipcc for gfx950 (the product), where a "phase" is straight-line code executed by every
//     thread of the workgroup followed by __syncthreads();
//   * by g++ with -DLRA_HOSTSIM (tests/hostsim, test infrastructure only), where a phase is a
//     loop over the workgroup's threads and every LDS access goes through a shadow that reports
//     cross-thread races inside a phase and reads of never-written LDS.  This is how index math
//     and barrier placement are checked in the build container, which has no GPU.
// The host simulation is never linked into the product library.
#pragma once

#include <cstddef>
#include <cstdint>

#ifdef LRA_HOSTSIM
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define LRA_HD inline
#define LRA_UNROLL _Pragma("GCC unroll 64")
#else
#include <hip/hip_runtime.h>
#define LRA_HD __host__ __device__ __forceinline__
#define LRA_UNROLL _Pragma("unroll")
#endif

// branch-probability hints: hipcc's block placement then keeps the rare paths (np.pad folds at a clip's edges, the non-finite flag,
// piece lists beyond the hoisted prefix) out of the frame loop's fall-through chain
#ifndef LRA_NO_EXPECT
#define LRA_LIKELY(x) __builtin_expect(!!(x), 1)
#define LRA_UNLIKELY(x) __builtin_expect(!!(x), 0)
#else
#define LRA_LIKELY(x) (x)
#define LRA_UNLIKELY(x) (x)
#endif

namespace lra {

// ----------------------------------------------------------------------------- complex helpers
template <class T> struct alignas(2 * sizeof(T)) cx {
    T x, y;
};
template <class T> LRA_HD cx<T> mk(T x, T y) { cx<T> r; r.x = x; r.y = y; return r; }
template <class T> LRA_HD cx<T> cadd(cx<T> a, cx<T> b) { return mk<T>(a.x + b.x, a.y + b.y); }
template <class T> LRA_HD cx<T> csub(cx<T> a, cx<T> b) { return mk<T>(a.x - b.x, a.y - b.y); }
template <class T> LRA_HD cx<T> cmul(cx<T> a, cx<T> b) { return mk<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
template <class T> LRA_HD cx<T> cconj(cx<T> a) { return mk<T>(a.x, -a.y); }
// multiply by -i (forward-transform quarter turn)
template <class T> LRA_HD cx<T> cmul_mi(cx<T> a) { return mk<T>(a.y, -a.x); }

// ----------------------------------------------------------------------------- packed-f32 complex arithmetic
// A complex float lives in an aligned VGPR pair, and gfx950's packed-f32 VALU ops (v_pk_add/mul/fma_f32)
// can read either half of each source for either half of the result (op_sel / op_sel_hi) and negate per
// half (neg_lo / neg_hi).  Every rotation-and-add of a butterfly -- a -+ i b, conj, a + k (-i u), the
// complex multiply -- is therefore ONE or TWO instructions on whole pairs.  hipcc only finds the
// symmetric forms (both halves negated); for the others it computes two packed results and splices their
// halves with v_mov (28 % of the FFT phases' VALU instructions were such moves), so the asymmetric
// forms are spelled out here.  The plain C++ bodies are the definition (used for double, for the host
// pass and by the CPU simulator); the asm is the same arithmetic, IEEE-identical per lane.
// The inline assembly below (v_pk_*_f32 with op_sel / neg modifiers, v_cndmask_b32_e64 under a 64-bit ballot, v_permlane32_swap_b32 with
// hand-placed s_nop hazards) is gfx950 wave64 code and nothing else: a build for any other target must fail here, not miscompile.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "librosa_amd kernels are written for gfx950 (MI355X, CDNA4) only: build with --offload-arch=gfx950"
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(__AMDGCN_WAVEFRONT_SIZE) && __AMDGCN_WAVEFRONT_SIZE != 64
#error "librosa_amd kernels assume 64-wide wavefronts"
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LRA_NO_PK_ASM)
#define LRA_PK_ASM 1
namespace pk {
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 v(cx<float> a) { return __builtin_bit_cast(f2, a); }
__device__ __forceinline__ cx<float> c(f2 a) { return __builtin_bit_cast(cx<float>, a); }
// r.lo = (+-a.lo) + (+-b[S1L]);  r.hi = (+-a.hi) + (+-b[S1H])      (N..: 1 = negate that operand half)
template <int S1L, int S1H, int N0L, int N0H, int N1L, int N1H> __device__ __forceinline__ f2 add(f2 a, f2 b) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,%3] op_sel_hi:[1,%4] neg_lo:[%5,%6] neg_hi:[%7,%8]"
        : "=v"(r) : "v"(a), "v"(b), "n"(S1L), "n"(S1H), "n"(N0L), "n"(N1L), "n"(N0H), "n"(N1H));
    return r;
}
// r.lo = a[S0L] * b[S1L];  r.hi = a[S0H] * b[S1H]
template <int S0L, int S0H, int S1L, int S1H> __device__ __forceinline__ f2 mul(f2 a, f2 b) {
    f2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[%3,%4] op_sel_hi:[%5,%6]" : "=v"(r) : "v"(a), "v"(b), "n"(S0L), "n"(S1L), "n"(S0H), "n"(S1H));
    return r;
}
// r.lo = (+-a[S0L]) * b[S1L] + c.lo;  r.hi = (+-a[S0H]) * b[S1H] + c.hi
template <int S0L, int S0H, int S1L, int S1H, int NL, int NH> __device__ __forceinline__ f2 fma(f2 a, f2 b, f2 c) {
    f2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[%4,%5,0] op_sel_hi:[%6,%7,1] neg_lo:[%8,0,0] neg_hi:[%9,0,0]"
        : "=v"(r) : "v"(a), "v"(b), "v"(c), "n"(S0L), "n"(S1L), "n"(S0H), "n"(S1H), "n"(NL), "n"(NH));
    return r;
}
}  // namespace pk
#endif

// a - i b  and  a + i b
template <class T> LRA_HD cx<T> add_mi(cx<T> a, cx<T> b) {
#ifdef LRA_PK_ASM
    if constexpr (sizeof(T) == 4) return pk::c(pk::add<1, 0, 0, 0, 0, 1>(pk::v(a), pk::v(b)));
#endif
    return mk<T>(a.x + b.y, a.y - b.x);
}
template <class T> LRA_HD cx<T> sub_mi(cx<T> a, cx<T> b) {
#ifdef LRA_PK_ASM
    if constexpr (sizeof(T) == 4) return pk::c(pk::add<1, 0, 0, 0, 1, 0>(pk::v(a), pk::v(b)));
#endif
    return mk<T>(a.x - b.y, a.y + b.x);
}
// a + conj(b)  and  a - conj(b)
template <class T> LRA_HD cx<T> add_conj(cx<T> a, cx<T> b) {
#ifdef LRA_PK_ASM
    if constexpr (sizeof(T) == 4) return pk::c(pk::add<0, 1, 0, 0, 0, 1>(pk::v(a), pk::v(b)));
#endif
    return mk<T>(a.x + b.x, a.y - b.y);
}
template <class T> LRA_HD cx<T> sub_conj(cx<T> a, cx<T> b) {
#ifdef LRA_PK_ASM
    if constexpr (sizeof(T) == 4) return pk::c(pk::add<0, 1, 0, 0, 1, 0>(pk::v(a), pk::v(b)));
#endif
    return mk<T>(a.x - b.x, a.y + b.y);
}
// conj(a - b)  and  -conj(a + i b) = (-(a.x - b.y), a.y + b.x) ... spelled by their components below
template <class T> LRA_HD cx<T> conj_sub(cx<T> a, cx<T> b) {  // (a.x - b.x, b.y - a.y)
#ifdef LRA_PK_ASM
    if constexpr (sizeof(T) == 4) return pk::c(pk::add<0, 1, 0, 1, 1, 0>(pk::v(a), pk::v(b)));
#endif
    return mk<T>(a.x - b.x, b.y - a.y);
}
template <class T> LRA_HD cx<T> conj_add_mi_neg(cx<T> a, cx<T> b) {  // (a.x - b.y, -a.y - b.x) = conj(a + i b)
#ifdef LRA_PK_ASM
    if constexpr (sizeof(T) == 4) return pk::c(pk::add<1, 0, 0, 1, 1, 1>(pk::v(a), pk::v(b)));
#endif
    return mk<T>(a.x - b.y, -a.y - b.x);
}
// a * w  and  a * conj(w): two instructions
template <class T> LRA_HD cx<T> cmul2(cx<T> a, cx<T> w) {
#ifdef LRA_PK_ASM
    if constexpr (sizeof(T) == 4) {
        const pk::f2 p = pk::mul<0, 1, 0, 0>(pk::v(a), pk::v(w));                  // (a.x w.x, a.y w.x)
        return pk::c(pk::fma<1, 0, 1, 1, 1, 0>(pk::v(a), pk::v(w), p));            // (-a.y w.y + ., a.x w.y + .)
    }
#endif
    return cmul(a, w);
}
// the same product as two separately schedulable halves, p = cmul_p(a, w) then cmul_f(a, w, p): callers
// with several independent products issue all the p's first (a packed result cannot feed the very next
// instruction without a wait state)
template <class T> LRA_HD cx<T> cmul_p(cx<T> a, cx<T> w) {
#ifdef LRA_PK_ASM
    if constexpr (sizeof(T) == 4) return pk::c(pk::mul<0, 1, 0, 0>(pk::v(a), pk::v(w)));
#endif
    return mk<T>(a.x * w.x, a.y * w.x);
}
template <class T> LRA_HD cx<T> cmul_f(cx<T> a, cx<T> w, cx<T> p) {
#ifdef LRA_PK_ASM
    if constexpr (sizeof(T) == 4) return pk::c(pk::fma<1, 0, 1, 1, 1, 0>(pk::v(a), pk::v(w), pk::v(p)));
#endif
    return mk<T>(p.x - a.y * w.y, p.y + a.x * w.y);
}
template <class T> LRA_HD cx<T> cmul2_conj(cx<T> a, cx<T> w) {
#ifdef LRA_PK_ASM
    if constexpr (sizeof(T) == 4) {
        const pk::f2 p = pk::mul<0, 1, 0, 0>(pk::v(a), pk::v(w));                  // (a.x w.x, a.y w.x)
        return pk::c(pk::fma<1, 0, 1, 1, 0, 1>(pk::v(a), pk::v(w), p));            // (a.y w.y + ., -a.x w.y + .)
    }
#endif
    return mk<T>(a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y);
}
// (-i a) * w = (a.y w.x + a.x w.y, a.y w.y - a.x w.x): two instructions
template <class T> LRA_HD cx<T> cmul2_mi(cx<T> a, cx<T> w) {
#ifdef LRA_PK_ASM
    if constexpr (sizeof(T) == 4) {
        const pk::f2 p = pk::mul<0, 1, 1, 1>(pk::v(a), pk::v(w));                  // (a.x w.y, a.y w.y)
        return pk::c(pk::fma<1, 0, 0, 0, 0, 1>(pk::v(a), pk::v(w), p));            // (a.y w.x + ., -a.x w.x + .)
    }
#endif
    return mk<T>(a.y * w.x + a.x * w.y, a.y * w.y - a.x * w.x);
}
// acc + k (-i u) = (acc.x + k u.y, acc.y - k u.x)  and  acc - k (-i u); k real, passed as the pair (k, k)
template <class T> LRA_HD cx<T> axpy_mi(T k, cx<T> u, cx<T> acc) {
#ifdef LRA_PK_ASM
    if constexpr (sizeof(T) == 4) return pk::c(pk::fma<1, 0, 0, 0, 0, 1>(pk::v(u), pk::v(mk<T>(k, k)), pk::v(acc)));
#endif
    return mk<T>(acc.x + k * u.y, acc.y - k * u.x);
}
template <class T> LRA_HD cx<T> axpy_pi(T k, cx<T> u, cx<T> acc) {
#ifdef LRA_PK_ASM
    if constexpr (sizeof(T) == 4) return pk::c(pk::fma<1, 0, 0, 0, 1, 0>(pk::v(u), pk::v(mk<T>(k, k)), pk::v(acc)));
#endif
    return mk<T>(acc.x - k * u.y, acc.y + k * u.x);
}
// acc + k u, k real
template <class T> LRA_HD cx<T> axpy(T k, cx<T> u, cx<T> acc) { return mk<T>(acc.x + k * u.x, acc.y + k * u.y); }
// acc + (z.x w.x, -z.y w.y): the inverse transform's window multiply of a conjugated value fused with the overlap-add, ONE instruction
// (hipcc builds it from two packed FMAs on re-paired halves: seven v_mov per pair)
template <class T> LRA_HD cx<T> fma_conj_win(cx<T> z, cx<T> w, cx<T> acc) {
#ifdef LRA_PK_ASM
    if constexpr (sizeof(T) == 4) return pk::c(pk::fma<0, 1, 0, 1, 0, 1>(pk::v(z), pk::v(w), pk::v(acc)));
#endif
    return mk<T>(acc.x + z.x * w.x, acc.y - z.y * w.y);
}
// element-wise product of two pairs
template <class T> LRA_HD cx<T> pmul(cx<T> a, cx<T> b) { return mk<T>(a.x * b.x, a.y * b.y); }
// Keeps hipcc from merging the tail of one arm of a branch with the other arm's (SimplifyCFG sinks "common" loads / stores behind the join
// and selects their ADDRESSES per arm: the fast path's four 8-byte accesses with immediate offsets became eight 4-byte accesses through
// seven 64-bit address computations).  An inline-asm statement is never sunk, and the scan for sinkable code stops at it.
#if defined(__HIP_DEVICE_COMPILE__)
#define LRA_ARM_END(tag) asm volatile("; arm " tag)
#else
#define LRA_ARM_END(tag) ((void)0)
#endif

// Wave priority at the phase boundaries of the frame loops (s_setprio; P < 0: no instruction; nothing in the host simulator).  With two
// waves per SIMD the issue arbiter otherwise serves the older wave first whatever it is doing; a wave in its transform passes (dense
// vector work) ahead of one in its epilogue (short dependent LDS round trips) is worth 5-6 % of the fused mel kernel
// (profiles/r04_experiments.md 10).
template <int P> LRA_HD void lra_setprio() {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (P >= 0) __builtin_amdgcn_s_setprio(P);
#endif
}
#ifndef LRA_V1_PRIO_A   // first-generation forward kernel, mel epilogue (stft_block): transform 3 / split + accumulate 0 / band combine 1 (n_fft 512, 80 mels, 256 x 30 s:
                        // hop 128 0.994 -> 0.946 ms, hop 160 0.849 -> 0.821 ms; profiles/r04_experiments.md 10)
#define LRA_V1_PRIO_A 3
#endif
#ifndef LRA_V1_PRIO_B
#define LRA_V1_PRIO_B 0
#endif
#ifndef LRA_V1_PRIO_B3
#define LRA_V1_PRIO_B3 1
#endif
#ifndef LRA_V1_PRIO_CA  // first-generation kernels, complex / power epilogues (experiment hooks; -1 = no instruction)
#define LRA_V1_PRIO_CA -1
#endif
#ifndef LRA_V1_PRIO_CS
#define LRA_V1_PRIO_CS -1
#endif
#ifndef LRA_I_PRIO_A
#define LRA_I_PRIO_A -1
#endif
#ifndef LRA_I_PRIO_B
#define LRA_I_PRIO_B -1
#endif

// Per-lane select under a mask that is held in an SGPR pair: c ? a : b as v_cndmask_b32_e64 (VOP3).  hipcc emits the VOP2 form with
// the implicit VCC operand for most selects, and on gfx950 a v_cndmask_b32_e32 issued directly behind another one stalls the
// SIMD's vector pipe for ~16 cycles (scripts/valu_probe2.hip: 9.7 ns per instruction back to back at one or two waves per SIMD
// against 2.0 ns for the VOP3 form and 1.1 ns when another VALU instruction sits in between) -- the lane-0 selects of the
// mirrored split come in pairs (real / imaginary half), i.e. every second one paid that.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LRA_NO_SEL_ASM)
#define LRA_SEL_ASM 1
using LaneMask = unsigned long long;
__device__ __forceinline__ LaneMask lane_mask(bool c) { return __builtin_amdgcn_ballot_w64(c); }
__device__ __forceinline__ float sel_mask(LaneMask m, bool, float a, float b) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
__device__ __forceinline__ double sel_mask(LaneMask, bool c, double a, double b) { return c ? a : b; }
#else
using LaneMask = bool;
LRA_HD LaneMask lane_mask(bool c) { return c; }
template <class T> LRA_HD T sel_mask(LaneMask, bool c, T a, T b) { return c ? a : b; }
#endif
template <class T> LRA_HD cx<T> sel_mask(LaneMask m, bool c, cx<T> a, cx<T> b) { return mk<T>(sel_mask(m, c, a.x, b.x), sel_mask(m, c, a.y, b.y)); }

// Pitch (in (A, B) pairs) of the run-ordered mel epilogue's running-sum area, register slot to register slot (lra_mel.h builds the piece
// addresses with it, the kernels store with it).  Layout 0: first-generation kernels, pitch TF.  Layout 1 (second generation):
// Round 4 reads the piece totals as whole (A, B) pairs (ds_read_b64: 64 banks; a 4-byte read of one half of an 8-byte slot can only ever
// land on 16 of its 32 banks, a built-in 2-way conflict), for which the plain pitch TF models best (scripts/lds_model.py: 57 LDS cycles per
// frame for the 16 reads against 90 with 4-byte reads at pitch TF + 1).
#ifndef LRA_MEL_RS_PITCH_EXTRA
#define LRA_MEL_RS_PITCH_EXTRA 0
#endif
LRA_HD int mel_runs_pitch(int tf_count, int layout) { return layout == 1 ? tf_count + LRA_MEL_RS_PITCH_EXTRA : tf_count; }

// pad modes for centred framing (np.pad modes the reference forwards, core/spectrum.py:287)
enum PadMode : int { PAD_CONSTANT = 0, PAD_REFLECT = 1, PAD_EDGE = 2, PAD_SYMMETRIC = 3 };

// Index into a signal of length n for a (possibly out-of-range) position g under a pad mode.
// Returns -1 for "zero" (constant mode or empty signal).  Matches np.pad, including repeated
// reflection when the pad is longer than the signal.
LRA_HD long long pad_index(long long g, long long n, int mode) {
    if (g >= 0 && g < n) return g;
    if (n <= 0 || mode == PAD_CONSTANT) return -1;
    if (mode == PAD_EDGE || n == 1) return g < 0 ? 0 : n - 1;
    // reflect / symmetric: fold back into [0, n).  A loop instead of a modulo: one trip unless the
    // pad is longer than the signal, and no 64-bit division in the kernels.
    const long long lo = mode == PAD_REFLECT ? 0 : -1;               // g < 0  ->  lo - g
    const long long hi = mode == PAD_REFLECT ? 2 * (n - 1) : 2 * n - 1;  // g >= n ->  hi - g
    while (g < 0 || g >= n) g = g < 0 ? lo - g : hi - g;
    return g;
}

// ----------------------------------------------------------------------------- LDS access layer
#ifdef LRA_HOSTSIM
namespace sim {
struct State {
    std::vector<unsigned char> mem;
    std::vector<int> w_epoch, w_tid, r_epoch, r_tid;  // per 4-byte word
    std::vector<int> w_wg, r_wg, r_wave;                // workgroup-barrier epoch of last write / read
    std::vector<unsigned char> ever;
    int epoch = 1;     // advances at every phase end (wave-level or workgroup-level sync)
    int wg_epoch = 1;  // advances only at workgroup barriers
    int cur_tid = 0;
    long long races = 0, uninit = 0;
    void resize(size_t bytes) {
        mem.assign(bytes, 0xCD);
        size_t w = (bytes + 3) / 4;
        w_epoch.assign(w, 0); w_tid.assign(w, -1); r_epoch.assign(w, 0); r_tid.assign(w, -1); ever.assign(w, 0);
        w_wg.assign(w, 0); r_wg.assign(w, 0); r_wave.assign(w, -1);
    }
    // wave_only: the phase boundary is a wave-level fence, not __syncthreads(): data may only flow
    // between lanes of the same wave64 across it
    void barrier(bool wave_only = false) { ++epoch; if (!wave_only) ++wg_epoch; }
    void note_read(size_t off, size_t bytes) {
        if (off + bytes > mem.size()) { std::fprintf(stderr, "hostsim: LDS read out of bounds off=%zu\n", off); std::abort(); }
        for (size_t w = off / 4; w < (off + bytes + 3) / 4; ++w) {
            if (!ever[w]) { if (uninit < 4 && std::getenv("LRA_SIM_DEBUG")) std::fprintf(stderr, "hostsim: uninit read word %zu tid %d epoch %d\n", w, cur_tid, epoch); ++uninit; }
            if (w_epoch[w] == epoch && w_tid[w] != cur_tid) ++races;
            // written by another wave with no workgroup barrier in between
            if (ever[w] && w_wg[w] == wg_epoch && w_tid[w] / 64 != cur_tid / 64) ++races;
            if (r_epoch[w] != epoch) { r_epoch[w] = epoch; r_tid[w] = cur_tid; }
            else if (r_tid[w] != cur_tid) r_tid[w] = -2;  // several readers this phase
            if (r_wg[w] != wg_epoch) { r_wg[w] = wg_epoch; r_wave[w] = cur_tid / 64; }
            else if (r_wave[w] != cur_tid / 64) r_wave[w] = -2;  // several waves since the last barrier
        }
    }
    void note_write(size_t off, size_t bytes) {
        if (off + bytes > mem.size()) { std::fprintf(stderr, "hostsim: LDS write out of bounds off=%zu\n", off); std::abort(); }
        for (size_t w = off / 4; w < (off + bytes + 3) / 4; ++w) {
            if (w_epoch[w] == epoch && w_tid[w] != cur_tid) ++races;
            if (r_epoch[w] == epoch && r_tid[w] != cur_tid) ++races;
            if (ever[w] && w_wg[w] == wg_epoch && w_tid[w] / 64 != cur_tid / 64) ++races;   // cross-wave WAW
            if (r_wg[w] == wg_epoch && r_wave[w] != cur_tid / 64) ++races;                    // cross-wave WAR
            w_epoch[w] = epoch; w_tid[w] = cur_tid; w_wg[w] = wg_epoch; ever[w] = 1;
        }
    }
};
inline State& state() { static thread_local State s; return s; }
}  // namespace sim

struct Lds {
    size_t base;  // byte offset of this view inside the simulated LDS
};
template <class V> inline V lds_ld(Lds l, int byte_off) {
    auto& s = sim::state();
    s.note_read(l.base + byte_off, sizeof(V));
    V v; std::memcpy(&v, s.mem.data() + l.base + byte_off, sizeof(V)); return v;
}
template <class V> inline void lds_st(Lds l, int byte_off, V v) {
    auto& s = sim::state();
    s.note_write(l.base + byte_off, sizeof(V));
    std::memcpy(s.mem.data() + l.base + byte_off, &v, sizeof(V));
}
inline Lds lds_sub(Lds l, int byte_off) { Lds r; r.base = l.base + byte_off; return r; }

#define LRA_LAUNDER(p) ((void)0)
#define LRA_ATOMIC_OR(ptr, v) (*(ptr) |= (v))
#define LRA_UNIFORM(x) (x)
#define LRA_KEEP(x) ((void)0)
#define LRA_RAW_TID(tid) (tid)
template <class T> inline T fast_div(T x, T w) { return x / w; }
template <class V> inline void stream_store(V* p, V v) { *p = v; }
template <class V> inline void stream_store16(V* p, V v) { *p = v; }
template <class V> inline V stream_load(const V* p) { return *p; }
template <class T> inline void store4_unaligned(T* p, T a, T b, T c, T d) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }
#define LRA_PHASE(NT, tid) for (int tid = 0; tid < (NT); ++tid) { ::lra::sim::state().cur_tid = tid;
#define LRA_PHASE_END } ::lra::sim::state().barrier();
#define LRA_PHASE_END_SYNC(WAVE) } ::lra::sim::state().barrier(WAVE);
#define LRA_REGS(Type, name, NT) std::vector<Type> name##_all(NT)
#define LRA_R(name) name##_all[tid]
// exchange of `name`.v[first .. first + 8) between lane l < 32 and lane l + 32 of every wave64 (a phase of its own); `active` must be
// the same for both lanes of a pair: an inactive pair keeps its values
#define LRA_HALF_SWAP8(name, tid, first, active)                                                                  \
    if ((tid & 63) < 32 && (active)) {                                                                            \
        for (int m_ = 0; m_ < 8; ++m_) std::swap(name##_all[tid].v[(first) + m_], name##_all[tid + 32].v[(first) + m_]); \
    }

#else  // device build

struct Lds {
    char* base;
};
// LRA_LDS_NOMERGE (experiment): 8-byte LDS reads are issued as volatile loads, which keeps hipcc from fusing neighbouring pairs
// into ds_read2_b64 / ds_read2st64_b64.  /opt/skills/guides/MI355X_MICROARCH.md (LDS table) prices ds_read2_b64 at 8 LDS cycles per
// wave-instruction against 2 for ds_read_b64, i.e. the fused form at HALF the rate of the two reads it replaces.
#ifndef LRA_LDS_NOMERGE
#define LRA_LDS_NOMERGE 1
#endif
template <class V> LRA_HD V lds_ld(Lds l, int byte_off) {
#if LRA_LDS_NOMERGE
    if constexpr (sizeof(V) == 8) {
        typedef float f2v __attribute__((ext_vector_type(2)));
        // (explicit LDS address space: address-space inference leaves volatile accesses alone, and a flat volatile load is no DS op)
        const f2v t = *(const volatile __attribute__((address_space(3))) f2v*)(l.base + byte_off);
        return __builtin_bit_cast(V, t);
    }
#endif
    return *reinterpret_cast<const V*>(l.base + byte_off);
}
template <class V> LRA_HD void lds_st(Lds l, int byte_off, V v) { *reinterpret_cast<V*>(l.base + byte_off) = v; }
LRA_HD Lds lds_sub(Lds l, int byte_off) { Lds r; r.base = l.base + byte_off; return r; }

// makes the compiler forget what it knows about a (uniform) pointer: loads through it cannot be
// hoisted out of the enclosing loop
#define LRA_LAUNDER(p) asm volatile("" : "+s"(p))
#define LRA_ATOMIC_OR(ptr, v) atomicOr((ptr), (v))
// value known to be the same in every lane of the wave: keep it in an SGPR so that branches on it stay scalar
#define LRA_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
// pins a value's computation where it is written (hipcc may not sink it into a later branch)
#define LRA_KEEP(x) asm volatile("" : "+v"(x))
// the hardware thread index without the per-phase launder (for values that SHOULD be hoisted)
#define LRA_RAW_TID(tid) ((int)threadIdx.x)
// x / w for normal w: v_rcp_f32 + v_mul (<= 1 ulp) instead of the ~12-instruction IEEE division sequence
__device__ __forceinline__ float fast_div(float x, float w) { return x * __builtin_amdgcn_rcpf(w); }
__device__ __forceinline__ double fast_div(double x, double w) { return x / w; }
// Output that is written once and not read again by this kernel: a non-temporal (streaming) store.
#ifndef LRA_NT_STORE
#define LRA_NT_STORE 0  // measured on MI355X: no gain for the spectrum rows (the L2 write-back path already streams them)
#endif
template <class V> __device__ __forceinline__ void stream_store(V* p, V v) {
#if LRA_NT_STORE
    if constexpr (sizeof(V) == 8) {
        typedef float f2v __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(__builtin_bit_cast(f2v, v), reinterpret_cast<f2v*>(p));
    } else if constexpr (sizeof(V) == 16) {
        typedef double d2v __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(__builtin_bit_cast(d2v, v), reinterpret_cast<d2v*>(p));
    } else {
        __builtin_nontemporal_store(v, p);
    }
#else
    *p = v;
#endif
}
// four consecutive words as ONE store where the element type allows it (global_store_dwordx4; the pointer need only be element-aligned:
// the hardware takes unaligned vector stores, and the type says so to the compiler)
template <class T> __device__ __forceinline__ void store4_unaligned(T* p, T a, T b, T c, T d) {
    if constexpr (sizeof(T) == 4) {
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        f4u v;
        v.x = a; v.y = b; v.z = c; v.w = d;
        *reinterpret_cast<f4u*>(p) = v;
    } else {
        p[0] = a; p[1] = b; p[2] = c; p[3] = d;
    }
}
// Input that is read once: optionally a non-temporal load (experiment, LRA_NT_LOAD)
#ifndef LRA_NT_LOAD
#define LRA_NT_LOAD 0
#endif
template <class V> __device__ __forceinline__ V stream_load(const V* p) {
#if LRA_NT_LOAD
    if constexpr (sizeof(V) == 8) {
        typedef float f2v __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(V, __builtin_nontemporal_load(reinterpret_cast<const f2v*>(p)));
    }
#endif
    return *p;
}
// 16-byte piece of a row that is written once, whole 16-byte pieces only, neighbours back to back: non-temporal
// (measured on the store stream of the STFT: +8 % alone, +16 % with the XCD-aware workgroup map)
#ifndef LRA_V2_NT
#define LRA_V2_NT 1
#endif
template <class V> __device__ __forceinline__ void stream_store16(V* p, V v) {
    static_assert(sizeof(V) == 16, "16-byte pieces");
    typedef float f4v __attribute__((ext_vector_type(4)));
#if LRA_V2_NT
    __builtin_nontemporal_store(__builtin_bit_cast(f4v, v), reinterpret_cast<f4v*>(p));
#else
    *reinterpret_cast<f4v*>(p) = __builtin_bit_cast(f4v, v);
#endif
}
// Phase boundary.  WAVE = true: every lane that exchanges data through LDS across this boundary is
// in the same wave64 (a wave's DS instructions execute in order), so a compiler-level fence is
// enough and the waves of the workgroup are free to drift apart; otherwise a workgroup barrier.
template <bool WAVE> __device__ __forceinline__ void phase_sync() {
    if constexpr (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        // Workgroup barrier: the phases exchange data through LDS and nothing else, so all that is needed is
        // "my DS writes have retired" + s_barrier.  That is also what __syncthreads() compiles to on gfx950 today
        // (s_waitcnt lgkmcnt(0); s_barrier -- no vmcnt drain, checked in the ISA); it is spelled out so that the
        // in-flight prefetch loads and output stores can never be made to wait here by a stronger fence.  The
        // "memory" clobber keeps the compiler from moving LDS accesses across the barrier.
#ifdef LRA_FULL_BARRIER  // experiments only
        __syncthreads();
#else
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    }
}
// The thread index is re-read through an opaque move at the top of every phase.  Without it hipcc hoists
// every tid-derived LDS / global address of every phase out of the frame loop (dozens of VGPRs that stay
// live for the whole kernel) and then spills the values that matter, e.g. the prefetch registers.
__device__ __forceinline__ int phase_tid() {
    int t = (int)threadIdx.x;
#ifndef LRA_NO_TID_LAUNDER
    asm volatile("" : "+v"(t));
#endif
    return t;
}
#define LRA_PHASE(NT, tid) { const int tid = ::lra::phase_tid();
#define LRA_PHASE_END } ::lra::phase_sync<false>();
#define LRA_PHASE_END_SYNC(WAVE) } ::lra::phase_sync<(WAVE)>();
#define LRA_REGS(Type, name, NT) Type name
#define LRA_R(name) name
// v[0 .. 8) of lane l < 32 <-> v[0 .. 8) of lane l + 32, in place: v_permlane32_swap_b32 with one register on both sides swaps the
// register's two halves (scripts/permlane_probe.hip: so it does, and lanes switched off in EXEC keep their values together with
// their partners).  Raw instructions are invisible to hipcc's hazard recogniser, hence the wait states on both sides.
__device__ __forceinline__ void half_swap8(cx<float>* v, bool active) {
    if (active) {
        asm volatile(
            "s_nop 1\n\t"
            "v_permlane32_swap_b32 %0, %0\n\tv_permlane32_swap_b32 %1, %1\n\tv_permlane32_swap_b32 %2, %2\n\tv_permlane32_swap_b32 %3, %3\n\t"
            "v_permlane32_swap_b32 %4, %4\n\tv_permlane32_swap_b32 %5, %5\n\tv_permlane32_swap_b32 %6, %6\n\tv_permlane32_swap_b32 %7, %7\n\t"
            "v_permlane32_swap_b32 %8, %8\n\tv_permlane32_swap_b32 %9, %9\n\tv_permlane32_swap_b32 %10, %10\n\tv_permlane32_swap_b32 %11, %11\n\t"
            "v_permlane32_swap_b32 %12, %12\n\tv_permlane32_swap_b32 %13, %13\n\tv_permlane32_swap_b32 %14, %14\n\tv_permlane32_swap_b32 %15, %15\n\t"
            "s_nop 1"
            : "+v"(v[0].x), "+v"(v[0].y), "+v"(v[1].x), "+v"(v[1].y), "+v"(v[2].x), "+v"(v[2].y), "+v"(v[3].x), "+v"(v[3].y),
              "+v"(v[4].x), "+v"(v[4].y), "+v"(v[5].x), "+v"(v[5].y), "+v"(v[6].x), "+v"(v[6].y), "+v"(v[7].x), "+v"(v[7].y));
    }
}
__device__ __forceinline__ void half_swap8(cx<double>*, bool) {}  // (no double-precision configuration uses it)
#define LRA_HALF_SWAP8(name, tid, first, active) ::lra::half_swap8(LRA_R(name).v + (first), (active))

#endif

}  // namespace lra
