// hostsim.cpp -- CPU thread simulator for the kernel bodies in librosa_amd/csrc/lra_kernels.h.
//
// TEST INFRASTRUCTURE ONLY.  Built by tests (g++ -DLRA_HOSTSIM) into tests/hostsim/_hostsim.so and
// loaded with ctypes by tests/test_hostsim.py.  It executes the SAME templated workgroup bodies the
// gfx950 library launches, one phase at a time over all threads of a workgroup, with an LDS shadow
// that counts cross-thread races inside a phase and reads of never-written LDS.  It is never
// linked into, imported by, or used as a fallback for the product library.
#define LRA_HOSTSIM 1
#include "../../librosa_amd/csrc/lra_dispatch.h"
#include "../../librosa_amd/csrc/lra_kernels_pc.h"
#include "../../librosa_amd/csrc/lra_mel.h"

#include <cstdlib>
#include <vector>

using namespace lra;

namespace {

template <class T> struct StftSim {
    StftArgs<T> a;
    int mode;
    long long blocks;
    long long* diag;
    const T* dense_basis = nullptr;  // mode 3: dense [n_mels][M+1]
    bool use_v2 = true;
    bool mel_v2 = false;
    template <class Cfg, int MODE> void run(int iters, int shared_bytes) {
        std::vector<cx<T>> tw(Cfg::TW_TOTAL), twr(split_tw_count<Cfg>());
        build_pass_twiddles<Cfg>(tw.data());
        build_split_twiddles<Cfg>(twr.data());
        a.tw = tw.data();
        a.twr = twr.data();
        a.frames_per_wg = iters * Cfg::FPB;
        a.rot_uniform = (Cfg::FPB > 1 && Cfg::TF < 64 && iters % 4 == 0) ? 1 : 0;  // as StftLaunch::launch (no effect on the simulator: LRA_UNIFORM is the identity)
        a.wg_per_clip = (a.n_frames + a.frames_per_wg - 1) / a.frames_per_wg;
        a.mel_tile = iters < 3 ? iters : (iters == 5 ? 1 : 3);  // odd: exercises partial tiles; iters 5 -> direct stores
        a.slot_bytes = stft_slot_bytes<Cfg>(MODE, a.n_mels, a.mel_tile);
        a.shared_off = Cfg::FPB * a.slot_bytes;
        auto& st = sim::state();
        const long long nblk = blocks * a.wg_per_clip;
        for (long long blk = 0; blk < nblk; ++blk) {
            st.resize(Cfg::FPB * a.slot_bytes + shared_bytes);
            Lds lds; lds.base = 0;
            // same kernel selection as StftLaunch::launch (lra_api.hip): row-aligned hops take the fast ring path
            bool ra = false;
            if constexpr (sizeof(typename Cfg::real) == 4) ra = ring_rows_aligned<Cfg>(a.hop) && !std::getenv("LRA_SIM_NO_RA");
            // second-generation kernel body (lra_kernels2.h), same selection as StftLaunch::launch
            bool v2 = false;
            if constexpr (v23_cfg_ok<Cfg>() && (MODE == OUT_COMPLEX || MODE == OUT_POWER || MODE == OUT_MELR)) {
                const int hd = (use_v2 && (MODE != OUT_MELR || mel_v2)) ? v2_hop_divisor<Cfg>(a.hop) : 0;
                if (hd) {
                    v2 = true;
                    a.slot_bytes = stft2_slot_bytes<Cfg>();
                    a.shared_off = Cfg::FPB * a.slot_bytes;
                    st.resize(Cfg::FPB * stft2_slot_bytes<Cfg>() + shared_bytes);
#define SIM_V2(HD)                                                                                             \
    if (a.power_mode == POW_TWO) stft_block2<Cfg, HD, MODE, POW_TWO>(a, (int)blk, lds);                        \
    else if (a.power_mode == POW_ONE) stft_block2<Cfg, HD, MODE, POW_ONE>(a, (int)blk, lds);                   \
    else stft_block2<Cfg, HD, MODE, POW_GENERAL>(a, (int)blk, lds);
                    if (hd == 1) { SIM_V2(1) } else if (hd == 2) { SIM_V2(2) } else if (hd == 4) { SIM_V2(4) } else { SIM_V2(8) }
#undef SIM_V2
                }
            }
            diag[10] = v2;
            // direct framing (hop >= n_fft: no ring), same selection as StftLaunch::launch
            bool direct = false;
            if constexpr (MODE == OUT_COMPLEX || MODE == OUT_POWER) direct = !v2 && a.hop >= Cfg::N && !std::getenv("LRA_SIM_NO_DIRECT");
            int rhd = 0;
            if constexpr ((MODE == OUT_COMPLEX || MODE == OUT_POWER) && sizeof(typename Cfg::real) == 4 && (Cfg::LOGM >= 12 || Cfg::LOGM == 7 || Cfg::LOGM == 8)) rhd = (!v2 && !std::getenv("LRA_SIM_NO_DIRECT")) ? regring_hd<Cfg>(a.hop) : 0;
            if (v2) {
            } else if (rhd) {
                if constexpr ((MODE == OUT_COMPLEX || MODE == OUT_POWER) && sizeof(typename Cfg::real) == 4 && (Cfg::LOGM >= 12 || Cfg::LOGM == 7 || Cfg::LOGM == 8)) {
                    a.slot_bytes = Cfg::FRAME_BYTES;
                    a.shared_off = Cfg::FPB * a.slot_bytes;
                    st.resize(Cfg::FPB * a.slot_bytes + shared_bytes);
#define SIM_RR(RAM)                                                                                     \
    if (a.power_mode == POW_TWO) stft_block<Cfg, MODE, POW_TWO, RAM>(a, (int)blk, lds);                 \
    else if (a.power_mode == POW_ONE) stft_block<Cfg, MODE, POW_ONE, RAM>(a, (int)blk, lds);            \
    else stft_block<Cfg, MODE, POW_GENERAL, RAM>(a, (int)blk, lds);
                    if (rhd == 2) { SIM_RR(3) } else if (rhd == 4) { SIM_RR(4) } else if (rhd == 8) { SIM_RR(5) } else { SIM_RR(6) }
#undef SIM_RR
                }
            } else if (direct) {
                if constexpr (MODE == OUT_COMPLEX || MODE == OUT_POWER) {
                    a.slot_bytes = Cfg::FRAME_BYTES;
                    a.shared_off = Cfg::FPB * a.slot_bytes;
                    st.resize(Cfg::FPB * a.slot_bytes + shared_bytes);
                    if (a.power_mode == POW_TWO) stft_block<Cfg, MODE, POW_TWO, 2>(a, (int)blk, lds);
                    else if (a.power_mode == POW_ONE) stft_block<Cfg, MODE, POW_ONE, 2>(a, (int)blk, lds);
                    else stft_block<Cfg, MODE, POW_GENERAL, 2>(a, (int)blk, lds);
                }
            } else if (ra) {
                if constexpr (sizeof(typename Cfg::real) == 4) {
                    if (a.power_mode == POW_TWO) stft_block<Cfg, MODE, POW_TWO, 1>(a, (int)blk, lds);
                    else if (a.power_mode == POW_ONE) stft_block<Cfg, MODE, POW_ONE, 1>(a, (int)blk, lds);
                    else stft_block<Cfg, MODE, POW_GENERAL, 1>(a, (int)blk, lds);
                }
            } else if (a.power_mode == POW_TWO) stft_block<Cfg, MODE, POW_TWO, 0>(a, (int)blk, lds);
            else if (a.power_mode == POW_ONE) stft_block<Cfg, MODE, POW_ONE, 0>(a, (int)blk, lds);
            else stft_block<Cfg, MODE, POW_GENERAL, 0>(a, (int)blk, lds);
            diag[8] = ra;
            diag[0] += st.races; diag[1] += st.uninit;
            st.races = st.uninit = 0;
        }
        diag[2] = Cfg::NT; diag[3] = Cfg::FPB; diag[4] = Cfg::P; diag[5] = Cfg::FPB * a.slot_bytes + shared_bytes; diag[6] = Cfg::WAVE_SYNC;
    }
    // producer / consumer mel kernel body (lra_kernels_pc.h), selected as StftLaunch::run does (LRA_SIM_PC = the ctx option "mel_pc")
    template <class Cfg> void run_pc(int iters, int hd) {
        using PL = PcLayout<Cfg>;
        std::vector<cx<T>> tw(Cfg::TW_TOTAL), twr(split_tw_count<Cfg>());
        build_pass_twiddles<Cfg>(tw.data());
        build_split_twiddles<Cfg>(twr.data());
        a.tw = tw.data();
        a.twr = twr.data();
        a.frames_per_wg = iters * PL::NP;
        a.wg_per_clip = (a.n_frames + a.frames_per_wg - 1) / a.frames_per_wg;
        a.mel_tile = 1;
        a.slot_bytes = PL::FRAME;
        a.shared_off = 0;
        auto& st = sim::state();
        const long long nblk = blocks * a.wg_per_clip;
        for (long long blk = 0; blk < nblk; ++blk) {
            st.resize(PL::BYTES);
            Lds lds; lds.base = 0;
#define SIM_PC(HD)                                                                                 \
    if (a.power_mode == POW_TWO) stft_pc_block<Cfg, HD, POW_TWO>(a, (int)blk, lds);                \
    else if (a.power_mode == POW_ONE) stft_pc_block<Cfg, HD, POW_ONE>(a, (int)blk, lds);           \
    else stft_pc_block<Cfg, HD, POW_GENERAL>(a, (int)blk, lds);
            if (hd == 4) { SIM_PC(4) } else { SIM_PC(8) }
#undef SIM_PC
            diag[0] += st.races; diag[1] += st.uninit;
            st.races = st.uninit = 0;
        }
        diag[2] = PL::NT; diag[3] = PL::NP; diag[4] = Cfg::P; diag[5] = PL::BYTES; diag[6] = 1; diag[10] = 2;  // (v2 = 2: the producer / consumer body ran)
    }
    template <class Cfg> void operator()() {
        const int iters = a.frames_per_wg;  // caller passes the iteration count
        if (mode == OUT_MELR && std::getenv("LRA_SIM_PC")) {
            if constexpr (pc_cfg_ok<Cfg>()) {
                TwoSlope<T> ts = build_two_slope<T>(dense_basis, a.n_mels, Cfg::M + 1);
                const int hd = v2_hop_divisor<Cfg>(a.hop);
                if (ts.ok && pc_fits_budget<Cfg>(hd, a.power_mode)) {
                    MelRuns<T> mr = build_mel_runs<T>(ts, Cfg::TF, Cfg::R / 2, MELR_PMAX, 4, 1);
                    if (mr.ok && pc_bank_ok<Cfg>(a.n_mels, mr.pmax)) {
                        a.melr_w = mr.w.data(); a.melr_keep = mr.keep.data(); a.melr_addr = mr.addr.data(); a.melr_zero = mr.zero_addr; a.melr_mid = mr.mid_addr; a.melr_pmax = mr.pmax;
                        diag[9] = mr.max_pieces;
                        run_pc<Cfg>(iters, hd);
                        return;
                    }
                }
            }
            diag[7] = 3;  // not applicable: the library keeps stft2_kernel<OUT_MELR>
            return;
        }
        if (mode == OUT_MEL2) {
            using MC = typename MelCfgOf<Cfg>::type;
            TwoSlope<T> ts = build_two_slope<T>(dense_basis, a.n_mels, Cfg::M + 1);
            if (!ts.ok || !mel2_fits<MC>(a.n_mels)) { diag[7] = 1; return; }
            MelPieces mp = build_mel_pieces<T>(ts, MC::TF, MC::R);
            if (mp.n_pieces <= 0 || mp.n_pieces > MC::TF + a.n_mels + 2) { diag[7] = 2; return; }
            a.mel_wAB = ts.wAB.data(); a.mel_run = mp.run_desc.data(); a.mel_segd = mp.seg_desc.data(); a.mel_nyq = mp.nyquist_piece;
            run<MC, OUT_MEL2>(iters, mel2_shared_bytes<MC>(a.n_mels));
        } else if (mode == OUT_MELR) {
            using MC = typename MelCfgOf<Cfg>::type;
            TwoSlope<T> ts = build_two_slope<T>(dense_basis, a.n_mels, Cfg::M + 1);
            if constexpr (mel_many_applies<Cfg>()) {  // n_fft = 512 with more than 64 bands: the eight-bands-per-thread shape, as lra_api.hip selects it
                using MM = typename MelManyCfgOf<Cfg>::type;
                if (ts.ok && melr_fits<MM>() && a.n_mels > 100 && !std::getenv("LRA_SIM_NO_MELMANY")) {
                    MelRuns<T> mr = build_mel_runs<T>(ts, MM::TF, MM::R / 2, MELR_PMAX, FftRegs<MM>::MELR_PHOIST, 0);
                    if (!mr.ok) { diag[7] = 2; return; }
                    a.melr_w = mr.w.data(); a.melr_keep = mr.keep.data(); a.melr_addr = mr.addr.data(); a.melr_zero = mr.zero_addr; a.melr_mid = mr.mid_addr; a.melr_pmax = mr.pmax;
                    diag[9] = mr.max_pieces;
                    diag[11] = 1;
                    run<MM, OUT_MELR>(iters, melr_shared_bytes<MM>(a.n_mels, mr.pmax));
                    return;
                }
            }
            if (!ts.ok || !melr_fits<MC>()) { diag[7] = 1; return; }
            bool v2m = false;
            if constexpr (v23_cfg_ok<MC>()) v2m = use_v2 && v2_hop_divisor<MC>(a.hop) > 0;
            mel_v2 = v2m;
            MelRuns<T> mr = build_mel_runs<T>(ts, MC::TF, MC::R / 2, MELR_PMAX, FftRegs<MC>::MELR_PHOIST, v2m ? 1 : 0);
            if (!mr.ok) { diag[7] = 2; return; }
            a.melr_w = mr.w.data(); a.melr_keep = mr.keep.data(); a.melr_addr = mr.addr.data(); a.melr_zero = mr.zero_addr; a.melr_mid = mr.mid_addr; a.melr_pmax = mr.pmax;
            diag[9] = mr.max_pieces;
            run<MC, OUT_MELR>(iters, melr_shared_bytes<MC>(a.n_mels, mr.pmax));
        } else if (mode == OUT_COMPLEX) run<Cfg, OUT_COMPLEX>(iters, 0);
        else if (mode == OUT_POWER) run<Cfg, OUT_POWER>(iters, 0);
        else run<Cfg, OUT_MEL>(iters, 0);
    }
};

template <class T> struct IstftSim {
    IstftArgs<T> a;
    long long batch;
    long long* diag;
    int strip_groups;
    template <class Cfg> void operator()() {
        std::vector<cx<T>> tw(Cfg::TW_TOTAL), twr(split_tw_count<Cfg>());
        build_pass_twiddles<Cfg>(tw.data());
        build_split_twiddles<Cfg>(twr.data());
        a.tw = tw.data();
        a.twr = twr.data();
        const int FPB = Cfg::FPB, N = Cfg::N, H = a.hop;
        a.strip_frames = strip_groups;  // frames per strip
        a.strips_per_clip = (a.n_used + a.strip_frames - 1) / a.strip_frames;
        a.warm_frames = (N + H - 1) / H - 1;
        a.drain_steps = N > H ? (N - H + H - 1) / H : 0;
        a.batch = batch;
        auto& st = sim::state();
        const long long nblk = (batch * a.strips_per_clip + FPB - 1) / FPB;
        for (long long blk = 0; blk < nblk; ++blk) {
            int hc = 0;  // same kernel selection as IstftLaunch (lra_api.hip)
            if constexpr (sizeof(typename Cfg::real) == 4) hc = std::getenv("LRA_SIM_NO_RA") ? 0 : istft_rows_hc<Cfg>(a.hop);
            const bool rows = hc > 0;
            Lds lds; lds.base = 0;
            bool ran = false;
            if constexpr (sizeof(typename Cfg::real) == 4 && Cfg::R >= 4) {
                if (hc == Cfg::R / 2) { st.resize(istft_lds_bytes<Cfg, Cfg::R / 2>()); istft_block<Cfg, Cfg::R / 2>(a, (int)blk, lds); ran = true; }
                if (!ran && hc == Cfg::R / 4) { st.resize(istft_lds_bytes<Cfg, Cfg::R / 4>()); istft_block<Cfg, Cfg::R / 4>(a, (int)blk, lds); ran = true; }
            }
            if constexpr (sizeof(typename Cfg::real) == 4 && Cfg::R >= 8) {
                if (!ran && hc == Cfg::R / 8) { st.resize(istft_lds_bytes<Cfg, Cfg::R / 8>()); istft_block<Cfg, Cfg::R / 8>(a, (int)blk, lds); ran = true; }
            }
            if (!ran) st.resize(istft_lds_bytes<Cfg, 0>());
            if constexpr (sizeof(typename Cfg::real) == 4 && Cfg::R >= 16) {
                if (!ran && hc == Cfg::R / 16) { st.resize(istft_lds_bytes<Cfg, Cfg::R / 16>()); istft_block<Cfg, Cfg::R / 16>(a, (int)blk, lds); ran = true; }
            }
            if (!ran) istft_block<Cfg, 0>(a, (int)blk, lds);
            diag[8] = rows;
            diag[0] += st.races; diag[1] += st.uninit;
            st.races = st.uninit = 0;
        }
        diag[2] = Cfg::NT; diag[3] = Cfg::FPB; diag[4] = Cfg::P; diag[5] = istft_lds_bytes<Cfg, 0>(); diag[6] = Cfg::WAVE_SYNC;
    }
};

template <class T>
int run_stft(int n_fft, int mode, const T* y, long long n, long long batch, int n_frames, int hop, int center, int pad_mode, const T* win,
             int iters_per_wg, void* out, int power_mode, double power, const int* mel_c0, const int* mel_len, const int* mel_off, const T* mel_val,
             int n_mels, int variant, const T* dense_basis, long long* diag) {
    if (!pow2_supported(n_fft, sizeof(T) == 8)) return 1;
    StftSim<T> s;
    s.a = StftArgs<T>();
    s.a.y = y; s.a.y_stride = n; s.a.n = n; s.a.n_frames = n_frames; s.a.hop = hop; s.a.pad = center ? n_fft / 2 : 0; s.a.pad_mode = pad_mode;
    s.a.win = win; s.a.frames_per_wg = iters_per_wg;
    s.a.D = (cx<T>*)out; s.a.S = (T*)out; s.a.Mel = (T*)out;
    // rows of the complex / power result: packed, or (LRA_SIM_ROW_PAD = extra elements per row; the test reads the padded buffer) padded
    s.a.row_pitch = n_fft / 2 + 1 + (std::getenv("LRA_SIM_ROW_PAD") ? std::atoi(std::getenv("LRA_SIM_ROW_PAD")) : 0);
    s.a.power_mode = power_mode; s.a.power = (T)power;
    s.a.mel_c0 = mel_c0; s.a.mel_len = mel_len; s.a.mel_off = mel_off; s.a.mel_val = mel_val; s.a.n_mels = n_mels;
    s.mode = mode; s.blocks = batch; s.diag = diag; s.dense_basis = dense_basis;
    s.use_v2 = !std::getenv("LRA_SIM_NO_V2");
    for (int i = 0; i < 12; ++i) diag[i] = 0;
    return dispatch_logm<T>(log2_exact(n_fft) - 1, variant, s) ? 0 : 1;
}

template <class T>
int run_istft(int n_fft, const T* D /* interleaved complex [batch][T][M+1] */, long long batch, int n_frames_total, int n_used, int hop, int center,
              const T* win_scaled, const T* wss, double tiny, T* y, long long out_len, int strip_groups, int variant, long long* diag) {
    if (!pow2_supported(n_fft, sizeof(T) == 8)) return 1;
    IstftSim<T> s;
    s.a = IstftArgs<T>();
    const int M = n_fft / 2;
    s.a.D = (const cx<T>*)D; s.a.d_frame_stride = M + 1; s.a.d_batch_stride = (long long)n_frames_total * (M + 1);
    s.a.n_used = n_used; s.a.hop = hop; s.a.drop = center ? n_fft / 2 : 0; s.a.win_scaled = win_scaled; s.a.wss = wss; s.a.tiny = (T)tiny;
    s.a.y = y; s.a.y_stride = out_len; s.a.out_len = out_len;
    s.batch = batch; s.diag = diag; s.strip_groups = strip_groups;
    for (int i = 0; i < 12; ++i) diag[i] = 0;
    return dispatch_logm<T>(log2_exact(n_fft) - 1, variant, s) ? 0 : 1;
}

}  // namespace

// The four entry points are compiled as four objects side by side (HOSTSIM_PART = 1..4, tests/hostsim_util.py):
// each instantiates every n_fft configuration of one (transform, dtype) pair.
#ifndef HOSTSIM_PART
#define HOSTSIM_PART 0  // 0 = everything in one object
#endif
extern "C" {
#if HOSTSIM_PART == 0 || HOSTSIM_PART == 1
int hostsim_stft_f32(int n_fft, int mode, const float* y, long long n, long long batch, int n_frames, int hop, int center, int pad_mode,
                     const float* win, int iters_per_wg, void* out, int power_mode, double power, const int* mel_c0, const int* mel_len,
                     const int* mel_off, const float* mel_val, int n_mels, int variant, const float* dense_basis, long long* diag) {
    return run_stft<float>(n_fft, mode, y, n, batch, n_frames, hop, center, pad_mode, win, iters_per_wg, out, power_mode, power, mel_c0, mel_len, mel_off, mel_val, n_mels, variant, dense_basis, diag);
}
#endif
#if HOSTSIM_PART == 0 || HOSTSIM_PART == 2
int hostsim_stft_f64(int n_fft, int mode, const double* y, long long n, long long batch, int n_frames, int hop, int center, int pad_mode,
                     const double* win, int iters_per_wg, void* out, int power_mode, double power, const int* mel_c0, const int* mel_len,
                     const int* mel_off, const double* mel_val, int n_mels, int variant, const double* dense_basis, long long* diag) {
    return run_stft<double>(n_fft, mode, y, n, batch, n_frames, hop, center, pad_mode, win, iters_per_wg, out, power_mode, power, mel_c0, mel_len, mel_off, mel_val, n_mels, variant, dense_basis, diag);
}
#endif
#if HOSTSIM_PART == 0 || HOSTSIM_PART == 3
int hostsim_istft_f32(int n_fft, const float* D, long long batch, int n_frames_total, int n_used, int hop, int center, const float* win_scaled,
                      const float* wss, double tiny, float* y, long long out_len, int strip_groups, int variant, long long* diag) {
    return run_istft<float>(n_fft, D, batch, n_frames_total, n_used, hop, center, win_scaled, wss, tiny, y, out_len, strip_groups, variant, diag);
}
#endif
#if HOSTSIM_PART == 0 || HOSTSIM_PART == 4
int hostsim_istft_f64(int n_fft, const double* D, long long batch, int n_frames_total, int n_used, int hop, int center, const double* win_scaled,
                      const double* wss, double tiny, double* y, long long out_len, int strip_groups, int variant, long long* diag) {
    return run_istft<double>(n_fft, D, batch, n_frames_total, n_used, hop, center, win_scaled, wss, tiny, y, out_len, strip_groups, variant, diag);
}
long long hostsim_pad_index(long long g, long long n, int mode) { return pad_index(g, n, mode); }
long long hostsim_istft_written_end(int n_fft, int hop, long long n_used, int drop) { return istft_written_end(n_fft, hop, n_used, drop); }
#endif
}
