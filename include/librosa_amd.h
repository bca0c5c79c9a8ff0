/* librosa_amd.h -- C ABI of the MI355X (gfx950) implementation of librosa's STFT -> mel (+ ISTFT)
 * hot path.
 *
 * The reference (librosa, pure Python) exposes no C ABI / FFI for this path; its only swap point
 * is scipy.fft's uarray backend at the rfft/irfft call sites (librosa/core/spectrum.py:372,376,
 * 388,566,598), which hands over host float64 blocks of <= 32 frames and is the wrong granularity
 * for a GPU.  The drop-in boundary is therefore librosa's Python function level,
 * and this header is what the Python shim (librosa_amd/, ctypes) binds underneath it.  Every entry
 * point cites the reference code whose arithmetic it replaces.
 *
 * Conventions
 *   - plain C types only; every function returns 0 (LRA_OK) or a negative LRA_E* code, and
 *     lra_last_error() returns the message of the calling thread's last failure;
 *   - every data pointer is a DEVICE pointer unless the parameter is documented as "host";
 *     the caller owns all buffers; the library never frees or reallocates them;
 *   - all work is enqueued on the context's stream (its own, or one adopted with
 *     lra_ctx_set_stream, e.g. PyTorch's current stream); calls are asynchronous with respect to
 *     the host unless documented otherwise;
 *   - device layouts are row-major:  PCM y[batch][n];  spectrum D[batch][n_frames][n_bins]
 *     (n_bins = 1 + n_fft/2, interleaved re/im) -- librosa's (..., n_bins, n_frames) array is a
 *     transposed view of this buffer, exactly like the Fortran-ordered array the reference
 *     allocates (core/spectrum.py:356);  mel M[batch][n_mels][n_frames] (C order, as the
 *     reference's einsum returns, feature/spectral.py:2160).
 */
#ifndef LIBROSA_AMD_H
#define LIBROSA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LRA_OK 0
#define LRA_EINVAL (-1)   /* bad argument (maps to librosa.ParameterError in the shim) */
#define LRA_EHIP (-2)     /* HIP runtime error */
#define LRA_EROCFFT (-3)  /* rocFFT error */
#define LRA_ENODEV (-4)   /* no usable gfx950 device */
#define LRA_ENOMEM (-5)

#define LRA_F32 0 /* float32 PCM  <-> complex64  spectrum (util.dtype_r2c, util/utils.py:2404-2416) */
#define LRA_F64 1 /* float64 PCM  <-> complex128 spectrum */

/* np.pad modes that depend only on edge values (the ones stft supports natively on the device;
 * the shim pre-pads for the exotic ones; "wrap/maximum/mean/median/minimum" are rejected as in
 * core/spectrum.py:253-265). */
#define LRA_PAD_CONSTANT 0
#define LRA_PAD_REFLECT 1
#define LRA_PAD_EDGE 2
#define LRA_PAD_SYMMETRIC 3

typedef struct lra_ctx lra_ctx;
typedef struct lra_event lra_event;
typedef struct lra_stft_plan lra_stft_plan;
typedef struct lra_mel_plan lra_mel_plan;
typedef struct lra_istft_plan lra_istft_plan;

/* ---- library / context ------------------------------------------------------------------ */
const char* lra_last_error(void);
const char* lra_version(void);
int lra_device_count(int* count);
/* One context = one device + one stream + error state.  Fails with LRA_ENODEV when no GPU. */
int lra_ctx_create(int device, lra_ctx** out);
void lra_ctx_destroy(lra_ctx* ctx);
/* Enqueue on an existing hipStream_t, e.g. PyTorch's current stream.  NULL means HIP's default
 * (null) stream -- which is what torch.cuda.current_stream() is unless the caller changed it. */
int lra_ctx_set_stream(lra_ctx* ctx, void* hip_stream);
/* Go back to the context's own (non-blocking) stream. */
int lra_ctx_use_own_stream(lra_ctx* ctx);
/* A second stream of the context for work that may overlap the main chain (the constant-Q recursion: the octave transforms run beside
 * the chain of halvings, librosa/core/constantq.py:1054-1099).  LRA_SIDE_FORK: calls from now on are enqueued on the side stream, behind
 * everything enqueued on the main stream so far; LRA_SIDE_BACK: back to the main stream, which does NOT wait for the side stream;
 * LRA_SIDE_JOIN (from the main stream): the main stream waits for everything enqueued on the side stream.  Buffers used on the side
 * stream must stay alive until a join; LRA_SIDE_END = back (if forked) + join, for error paths.  The two directions use separate
 * events, and every fork its own event out of a ring whose slots are reused only once the side stream is past its wait on them
 * (re-recording one event that the other stream still waits on let that wait slip on ROCm 7.0: a use-after-free in the
 * caller's buffers).  lra_ctx_set_stream / lra_ctx_use_own_stream forget any fork. */
#define LRA_SIDE_FORK 1
#define LRA_SIDE_BACK 2
#define LRA_SIDE_JOIN 3
#define LRA_SIDE_END 4
int lra_ctx_side(lra_ctx* ctx, int mode);
int lra_ctx_sync(lra_ctx* ctx);
/* Tuning knobs: "stft_iters" (frames per slot, 0 = auto), "istft_strip_groups" (frames per strip, 0 = auto),
   "variant" (-1 = auto), "autotune" (1/0).  Kernel-form switches kept for same-box A/B runs (defaults in brackets; DESIGN.md 4.2 / 4.4b / 4.6d say what
   each form is): "v2" [1], "v3" [1: radices 16-16-4 for the complex epilogue; 2: for |X|^p and mel too], "mel_pc" [1: producer / consumer mel kernel; 2: on the
   16-16-4 core], "istft16" [0], "direct" [1], "mixed" [1: fused mixed-radix kernels; 2: the same with the mel band table read through the caches; 0: rocFFT path],
   "mixed_inv_pow2" [1], "mixed_irfft" [1: inverse transform of long mixed-radix frames as one launch in front of the gather kernel; 0: spec_pack + rocFFT C2R],
   "ola4" [1: four samples per thread in the gather kernel], "mixed_pow2_mel" [1: n_fft 128 / 256 with many bands on the flat-index kernel],
   "cqt_merge" [1: octaves 1-2 beside the later halvings; 2: all octaves behind the chain; 0: one launch per octave], "mel_many" [1], "hpss_tile" [1],
   "placement_retry" [4; 0 = off], "pipe_chunk_mb" [128], "pipe_threads" [8], "xcd_remap" [1].  Unknown keys are an error.  The same keys can be preset for
   every context of a process through the environment: LRA_CTX_OPTIONS="key=value,key=value". */
int lra_ctx_set_option(lra_ctx* ctx, const char* key, int value);
int lra_ctx_device_name(lra_ctx* ctx, char* buf, size_t buflen);
/* Device-side half of util.valid_audio (librosa/util/utils.py:294-306): the fused power-of-two STFT
 * kernels raise a sticky flag when a frame's DC bin is not finite, which (barring overflow) happens
 * iff some sample of the frame is NaN/Inf.  reset clears it (async); read syncs the stream. */
int lra_ctx_nonfinite_reset(lra_ctx* ctx);
int lra_ctx_nonfinite_read(lra_ctx* ctx, int* flag);
/* 1 when the plan runs the fused power-of-two kernels (and therefore feeds the flag above). */
int lra_stft_plan_is_fused(const lra_stft_plan* plan);
/* Kernel variant the plan settled on for an epilogue (mode 0 = stft, 1 = spectrogram, 2 = melspectrogram), after the
   first large call timed the candidates (ctx option "autotune", default on; "variant" >= 0 pins one instead):
   -1 = not measured (yet).  Reporting only (bench.py); the reference has no counterpart. */
int lra_stft_plan_tuned_variant(const lra_stft_plan* plan, int mode);
int lra_istft_plan_tuned_variant(const lra_istft_plan* plan);

/* device memory helpers for hosts that do not bring their own allocator (NumPy path of the shim) */
int lra_malloc(lra_ctx* ctx, size_t bytes, void** dptr);
int lra_free(lra_ctx* ctx, void* dptr);
/* Placement-aware allocation of a LARGE, long-lived result buffer of rows (a spectrum: the Fortran-ordered array of core/spectrum.py:356 seen
 * frame-major, `row_bytes` per frame).  Where such a buffer lands in HBM moves the store-bound transforms by up to 15 % on some boxes
 * (profiles/r05_pitch.md); this call builds up to `tries` candidates (ctx option "placement_retry" overrides; 1..8) -- alternately an ordinary hipMalloc
 * block and a range assembled from 64 MiB physical handles mapped in order: which kind lands better differs from box to box -- times the kernels' write
 * stream on each (~1 ms per candidate), keeps the best and releases the rest.  It stops early where two
 * candidates agree within 1.5 % (no lottery on this box) or one matches the best this context has seen.  rows_per_item: rows of one clip (its
 * frames; 0 = no item structure) -- the probe cuts the buffer into per-clip strips of rows exactly as the kernels do, because WHICH rows are written
 * at the same time is what the placement levels depend on.  probe_ms / tried: optional outputs.
 * Free with lra_free_placed only.  bytes >= 4096 rows; row_bytes a multiple of 8, >= 512.
 * Address ranges are never re-used or freed while the context lives (on ROCm 7.0 / gfx950 that leaves other live ranges with stale translations:
 * scripts/vmm_coherence.hip); a context spends at most 4 TiB of address space this way, then the call fails with LRA_ENOMEM and callers allocate normally. */
int lra_malloc_placed(lra_ctx* ctx, size_t bytes, int row_bytes, int64_t rows_per_item, int tries, void** dptr, float* probe_ms, int* tried);
int lra_free_placed(lra_ctx* ctx, void* dptr);
int lra_memset(lra_ctx* ctx, void* dptr, int value, size_t bytes);
int lra_memcpy_h2d(lra_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes); /* synchronous */
int lra_memcpy_d2h(lra_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes); /* synchronous */

/* HIP events on the context's stream (bench.py times the kernels with these) */
int lra_event_create(lra_ctx* ctx, lra_event** out);
void lra_event_destroy(lra_event* ev);
int lra_event_record(lra_event* ev);
int lra_event_elapsed_ms(lra_event* start, lra_event* stop, float* ms); /* syncs on stop */

/* ---- STFT: librosa.stft, librosa/core/spectrum.py:57-391 ------------------------------- */
/* window: host pointer, n_fft reals of `dtype` = pad_center(get_window(window, win_length), n_fft)
 * (core/spectrum.py:243-246), built by the shim with the reference's own scipy call.
 * center != 0 pads n_fft/2 samples each side with pad_mode (core/spectrum.py:252-328).  Power-of-two
 * n_fft in [32, 16384] (f64: 8192) runs the fused LDS-FFT kernels; any other n_fft runs a framing
 * kernel + rocFFT batched R2C. */
int lra_stft_plan_create(lra_ctx* ctx, int n_fft, int hop_length, const void* window_host, int center, int pad_mode, int dtype,
                         lra_stft_plan** out);
void lra_stft_plan_destroy(lra_stft_plan* plan);
/* n_frames = 1 + (n + 2*(n_fft/2)*center - n_fft) / hop  (core/spectrum.py:344-353); LRA_EINVAL when
 * n_fft > n and not centred (core/spectrum.py:330-333). */
int lra_stft_num_frames(const lra_stft_plan* plan, int64_t n, int64_t* n_frames);
/* D[batch][n_frames][n_bins] = rfft(window * frames)  (core/spectrum.py:380-390). */
int lra_stft_exec(lra_stft_plan* plan, const void* y, int64_t batch, int64_t n, int64_t y_stride, void* D);
/* S[batch][n_frames][n_bins] = |stft|**power  (_spectrogram, core/spectrum.py:3000-3013). */
int lra_spectrogram_exec(lra_stft_plan* plan, const void* y, int64_t batch, int64_t n, int64_t y_stride, double power, void* S);
/* The same two results with the rows of consecutive frames `out_frame_stride` ELEMENTS apart (>= n_bins; 0 = n_bins = the calls above):
 * kind 0 = lra_stft_exec, kind 1 = lra_spectrogram_exec.  The array librosa.stft returns is itself a strided view -- (..., n_bins, n_frames)
 * over storage whose fastest axis is the bin axis (core/spectrum.py:356, order="F") -- so a caller may pad each frame's row to a whole
 * number of 128-byte cache lines (1040 complex64 for n_fft = 2048) and present out[..., :n_bins] transposed: every wave-wide store of the
 * kernels then covers whole lines (round 5, profiles/r05_pitch.md).  Only the elements [0, n_bins) of a row are written.  Fused power-of-two
 * plans only (lra_stft_plan_is_fused); LRA_EINVAL otherwise. */
int lra_stft_exec_strided(lra_stft_plan* plan, int kind, const void* y, int64_t batch, int64_t n, int64_t y_stride, double power, void* out, int64_t out_frame_stride);

/* Host-buffer form of the three forward calls (what librosa.stft / _spectrogram / feature.melspectrogram take: NumPy arrays,
 * core/spectrum.py:57-391, :2920-3015, feature/spectral.py:2145-2160).  y_host: [batch] rows of n reals, y_stride elements
 * apart, pageable host memory; out_host: [batch] items of [n_frames][n_bins] complex (kind 0), [n_frames][n_bins] reals
 * (kind 1, |X|**power) or [n_mels][n_frames] reals (kind 2; mel required), out_item_stride reals apart (0 = packed).
 * Clips are moved in stages of ~pipe_chunk_mb (ctx option): host threads copy to / from pinned staging while the previous
 * stage's upload, kernel and download run on their own streams, so the call runs at link rate instead of at the rate of
 * a pageable copy.  Staging and device buffers persist in the context: a streaming caller (librosa.stream blocks ->
 * stft(center=False, out=D), docs/examples/plot_pcen_stream.py:72-74) allocates nothing after its first block.
 * nonfinite (optional): the staging threads also apply util.valid_audio's test (util/utils.py:305: every sample finite) to
 * the samples they copy; *nonfinite != 0 when it failed (the caller raises, the output is then meaningless).
 * Synchronous: out_host is complete on return. */
int lra_stft_exec_host(lra_stft_plan* plan, lra_mel_plan* mel, int kind, const void* y_host, int64_t batch, int64_t n, int64_t y_stride, double power,
                       void* out_host, int64_t out_item_stride, int* nonfinite);

/* ---- mel: librosa.filters.mel (filters.py:116-251) applied as in feature/spectral.py:2158-2160 */
/* basis: host pointer, dense [n_mels][n_bins] reals of `dtype`, built by the shim on the host with
 * the reference's float64 recipe (so it is bit-identical to filters.mel); stored on the device in
 * band form (first/last non-zero column per row; 2018 of 131200 entries at the default config). */
int lra_mel_plan_create(lra_ctx* ctx, int n_mels, int n_bins, const void* basis_host, int dtype, lra_mel_plan** out);
void lra_mel_plan_destroy(lra_mel_plan* plan);
/* Fused feature.melspectrogram(y=...): M[batch][n_mels][n_frames]; the complex spectrum never
 * reaches HBM (feature/spectral.py:2145-2160). */
int lra_melspectrogram_exec(lra_stft_plan* stft, lra_mel_plan* mel, const void* y, int64_t batch, int64_t n, int64_t y_stride, double power,
                            void* M);
/* feature.melspectrogram(S=...): M[b][m][t] = sum_f basis[m][f] * S[b*batch_stride + f*bin_stride + t*frame_stride]. */
int lra_mel_apply_exec(lra_mel_plan* mel, const void* S, int64_t batch, int64_t n_frames, int64_t batch_stride, int64_t bin_stride,
                       int64_t frame_stride, void* M);

/* ---- ISTFT: librosa.istft, librosa/core/spectrum.py:394-626 (+ __overlap_add :629-643) ---- */
int lra_istft_plan_create(lra_ctx* ctx, int n_fft, int hop_length, const void* window_host, int center, int dtype, lra_istft_plan** out);
void lra_istft_plan_destroy(lra_istft_plan* plan);
/* D: [batch] x frames x n_bins with the given element strides (d_frame_stride >= n_bins);
 * frames [0, n_used) are inverse-transformed, windowed and overlap-added in frame order; sample s of
 * the output is padded position s + n_fft/2 (centred) or s; wss (device, [out_len]) is
 * fix_length(window_sumsquare(...)[start:], out_len) computed by the shim exactly as the reference
 * does (core/spectrum.py:606-620); y[batch][out_len] = ola / wss where wss > tiny (:622-624). */
int lra_istft_exec(lra_istft_plan* plan, const void* D, int64_t batch, int64_t d_batch_stride, int64_t d_frame_stride, int64_t n_used,
                   const void* wss, void* y, int64_t out_len, int64_t y_stride);

/* The same transform with the normalisation handed over as FACTORS: norm[s] = 1 / wss[s] where wss[s] > tiny, else 1 (device, [out_len]) --
 * the form the fused kernels consume (y = ola * norm: core/spectrum.py:622-624 as a multiplication, within 1 ulp of the division).  A caller
 * that keeps the envelope of a (window, hop, length) combination around (the Python shim memoises it per context) builds the factors once;
 * lra_istft_exec derives them from `wss` with one small launch per call. */
int lra_istft_exec_norm(lra_istft_plan* plan, const void* D, int64_t batch, int64_t d_batch_stride, int64_t d_frame_stride, int64_t n_used,
                        const void* norm, void* y, int64_t out_len, int64_t y_stride);

/* Host-buffer form (librosa.istft on np.ndarrays): D_host [batch][n_frames][n_bins] complex, packed, pageable; wss_host
 * [out_len] reals; y_host [batch] rows of out_len reals, y_stride apart.  Same staged, overlapped transfer as
 * lra_stft_exec_host. */
int lra_istft_exec_host(lra_istft_plan* plan, const void* D_host, int64_t batch, int64_t n_frames, int64_t n_used, const void* wss_host, void* y_host,
                        int64_t out_len, int64_t y_stride);

/* ---- decibel scaling: librosa.power_to_db / amplitude_to_db, librosa/core/spectrum.py:1735-1883, 1946-2038 ---- */
/* Arrays are [batch][per_item] (the shim flattens the reduced axes -- "auto" = the last two -- into per_item).
 * out_max[b] = max_i |x[b][i]|: the reduction behind ref=np.max and top_db (log_spec.max(axes), :1877-1881). */
int lra_item_absmax_exec(lra_ctx* ctx, const void* x, int64_t batch, int64_t per_item, int dtype, void* out_max);
/* absolute != 0: as above (amplitude_to_db takes np.abs of every input, :2011).  absolute == 0: out_max[b] = max(0, max_i x[b][i]) --
 * power_to_db leaves real input signed (:1855-1859), and every use of the maximum is max(amin, .) with amin > 0. */
int lra_item_max_exec(lra_ctx* ctx, const void* x, int64_t batch, int64_t per_item, int dtype, int absolute, void* out_max);
/* out = 10 log10(max(amin, mag)) - 10 log10(max(amin, ref)), floored at its per-item maximum - top_db (:1873-1881).
 * amplitude != 0: mag = x**2, ref -> ref**2 (amplitude_to_db, :2030-2037; pass amin already squared), else mag = x as it is
 * (complex input has its modulus taken by the caller, :1855-1859; negative real values floor at amin).
 * ref_items (device, [batch]) overrides ref_scalar; item_max (device, [batch], input domain) is read when use_top_db. */
int lra_to_db_exec(lra_ctx* ctx, const void* x, void* out, int64_t batch, int64_t per_item, int dtype, int amplitude, double amin, double ref_scalar,
                   const void* ref_items, const void* item_max, int use_top_db, double top_db);
/* db_to_power: ref * 10**(0.1 x) (:1925); amplitude != 0: db_to_amplitude = db_to_power(x, ref**2)**0.5 (:2082). */
int lra_from_db_exec(lra_ctx* ctx, const void* x, void* out, int64_t count, int dtype, int amplitude, double ref);

/* ---- MFCC: librosa.feature.mfcc, librosa/feature/spectral.py:1843-2019 --------------------------------------------- */
/* out[b][k][t] = lift[k] * sum_m basis[m][k] * f(S[b][m][t]), k < n_out.  basis (device): band-major [n_in][ldc],
 * basis[m][k] = scipy.fft.dct(eye(n_in), axis=0, type, norm)[k][m] (:2005), ldc = n_out rounded up to a multiple of 128, zeros
 * beyond n_out;
 * lift (device, [n_out]): 1 + (lifter/2) sin(pi (k+1) / lifter) or ones (:2008-2015).  fuse_db != 0: f is the decibel
 * scaling of lra_to_db_exec (power domain), so that mfcc(y=...) reads the mel power spectrogram once (:2001). */
int lra_dct_exec(lra_ctx* ctx, const void* S, void* out, int64_t batch, int n_in, int n_out, int64_t n_frames, int dtype, const void* basis, const void* lift,
                 int fuse_db, double amin, double ref_scalar, const void* ref_items, const void* item_max, int use_top_db, double top_db);

/* ---- Griffin-Lim: librosa.griffinlim, librosa/core/spectrum.py:2669-2917 ------------------------------------------- */
/* One phase update over `count` complex values (:2896-2902):
 *   angles = rebuilt - coef * tprev (tprev may be NULL: first iteration);  angles /= |angles| + eps;  angles *= S
 * coef = momentum / (1 + momentum), eps = tiny(angles) (:2830).  normalize == 0: angles = rebuilt * S only (:2847).
 * rebuilt / tprev / angles: complex of `dtype`'s precision, S: real, all in the same element order ([batch][frame][bin] in
 * the shim's loop: lra_istft_exec -> lra_stft_exec -> this, device-resident across iterations); angles may alias rebuilt. */
int lra_griffinlim_update(lra_ctx* ctx, const void* rebuilt, const void* tprev, const void* S, void* angles, int64_t count, int dtype, double coef, double eps,
                          int normalize);

/* Initial estimate (:2832-2847, init="random"): angles = S * exp(2 pi i u), u: device array of float64 uniform draws (the
 * caller's rng.random(S.shape), so that a seed reproduces the reference's stream), in the element order of S / angles. */
int lra_griffinlim_init(lra_ctx* ctx, const void* u, const void* S, void* angles, int64_t count, int dtype);

/* NumPy's default generator on the device (round 5): PCG64 = PCG XSL RR 128/64 with O(log n) jump-ahead per thread, integer arithmetic only, so
 * the stream is np.random.default_rng's bit for bit.  state4 (host) = {state_hi, state_lo, inc_hi, inc_lo}: the two 128-bit integers of
 * Generator.bit_generator.state["state"].  out[i] (device, float64) = draw number offset + i, i.e. rng.random(offset + count)[offset:]. */
int lra_pcg64_random_exec(lra_ctx* ctx, const uint64_t* state4, uint64_t offset, void* out, int64_t count);
/* lra_griffinlim_init with the draws made in place: angles = S * exp(2 pi i u), u = rng.random((batch, n_bins, n_frames)) in THAT element order
 * (the reference's `rng.random(size=S.shape)`, librosa/core/spectrum.py:2832-2847), S / angles in the device layout [batch][frame][bin].
 * The caller advances its host generator by batch * n_bins * n_frames draws (bit_generator.advance) to leave it where the reference would. */
int lra_griffinlim_init_pcg64(lra_ctx* ctx, const uint64_t* state4, const void* S, void* angles, int64_t batch, int n_bins, int64_t n_frames, int dtype);

/* ---- phase vocoder: librosa.phase_vocoder, librosa/core/spectrum.py:1364-1519 (effects.time_stretch, effects.py:464-484) ---- */
/* D: [batch][n_frames][n_bins] complex (device), out: [batch][n_out][n_bins]; t_out_host: n_out fractional input frame times in
 * [0, n_frames) (np.arange(0, n_frames, rate) for a constant rate, :1488).  Phase: running sum of the phase differences of the
 * two input frames around each time (:1491-1507); magnitude: scipy interp1d(kind="linear") of |D| (:1507-1515). */
int lra_phase_vocoder_exec(lra_ctx* ctx, const void* D, void* out, int64_t batch, int64_t n_frames, int n_bins, const double* t_out_host, int64_t n_out, int dtype);

/* ---- PCEN: librosa.pcen, librosa/core/spectrum.py:2396-2666 (the consumer of the streaming STFT in
 * docs/examples/plot_pcen_stream.py:71-80) ---------------------------------------------------------------------------------- */
/* S, ref: [rows][n_frames] real of `dtype` (device), time on the last axis; ref = the array the smoother runs over (NULL: S itself,
 * i.e. max_size == 1 and no ref=, :2628-2630).  out: [rows][n_frames] FLOAT64 whatever `dtype` is -- the reference's filter state
 * is float64 and promotes everything after it (:2649-2665).
 *   M = scipy.signal.lfilter([b], [1, b - 1], ref, zi, axis=time)   smooth = exp(-gain (log eps + log1p(M / eps)))
 *   out = log1p(S smooth) | exp(power (log S + log smooth)) | bias**power expm1(power log1p(S smooth / bias))   (power == 0 | bias == 0 | else)
 * zi: device float64 [rows] initial filter state (a previous call's zf) or NULL = zi_scalar for every row (the caller's
 * scipy.signal.lfilter_zi([b], [1, b - 1]), :2649-2652).  zf: device float64 [rows] final state or NULL (return_zf, :2655). */
int lra_pcen_exec(lra_ctx* ctx, const void* S, const void* ref, void* out, int64_t rows, int64_t n_frames, int dtype, double b, double gain, double bias, double power, double eps,
                  const void* zi, double zi_scalar, void* zf);

/* scipy.ndimage.maximum_filter1d(S, size, axis) with mode="reflect", origin 0 -- pcen's max_size > 1 (:2640-2642).  S, out:
 * [outer][n_bands][inner] of `dtype` (device), the filter runs over n_bands: out[m] = max(S[m - size/2 .. m - size/2 + size - 1]). */
int lra_maxfilter_exec(lra_ctx* ctx, const void* S, void* out, int64_t outer, int n_bands, int64_t inner, int size, int dtype);

/* ---- constant-Q / variable-Q transform: librosa.cqt / librosa.vqt, librosa/core/constantq.py:42-225, 820-1122 ---------------------
 * The octave recursion (:1054-1099) is, per octave: lra_stft_exec with a rectangular window (__cqt_response, :1202-1204), then
 * lra_cqt_project_exec (the sparse filter basis applied to every frame, :1213-1218, with the length scaling :1116-1118 and the
 * stacking of __trim_stack :1168-1194 folded in), then lra_fir_decimate_exec (audio.resample by 2, :1095-1098). */

/* out[clip][n] = (sum_k taps[k] x[clip][(n + first) down - k]) / div * mul, x = 0 outside [0, n_in), 0 <= n < n_out.
 * With taps = the zero-prefixed filter, first = n_pre_remove and n_out = ceil(n_in / down) this is
 * scipy.signal.resample_poly(x, 1, down) (librosa/core/audio.py:676-693), summed in its order; div = sqrt(1 / down) is
 * resample(scale=True) (:719-720).  x, out, taps: real of `dtype` (device). */
int lra_fir_decimate_exec(lra_ctx* ctx, const void* x, void* out, int64_t batch, int64_t n_in, int64_t n_out, const void* taps, int n_taps, int down, int first, double div, double mul,
                          int dtype);

/* librosa.resample(res_type="polyphase") for any rational ratio, librosa/core/audio.py:676-693 = scipy.signal.resample_poly(x, up, down):
 * out[clip][n] = (sum_k x[clip][k] taps[(n + first) down - k up]) / div * mul over the real samples k under the filter (the
 * zero-stuffed signal filtered and decimated: scipy's upfirdn, summed in its order).  taps = scipy's design times `up`, zero-prefixed;
 * first = n_pre_remove; n_out = ceil(n_in up / down).  up == 1 is lra_fir_decimate_exec.  x, out, taps: real of `dtype` (device). */
int lra_resample_poly_exec(lra_ctx* ctx, const void* x, void* out, int64_t batch, int64_t n_in, int64_t n_out, const void* taps, int n_taps, int up, int down, int first, double div,
                           double mul, int dtype);

/* librosa.resample(res_type="fft" / "scipy"), librosa/core/audio.py:672-675 = scipy.signal.resample(x, n_out) along the last axis for
 * real x: out[clip] = irfft(Y, n_out) * (n_out / n_in) * gain, Y = the lower min(n_in, n_out) / 2 + 1 bins of rfft(x[clip]) (zero above),
 * the shared Nyquist bin doubled when shortening / halved when lengthening an even length.  Whole-signal transforms (rocFFT, any
 * length < 2^31; plans kept per length, least recently used evicted); gain: the caller's scaling (1 / sqrt(ratio) of
 * resample(scale=True), :719-720).  x: [batch][n_in], out: [batch][n_out] real of `dtype` (device). */
int lra_resample_fft_exec(lra_ctx* ctx, const void* x, void* out, int64_t batch, int64_t n_in, int64_t n_out, double gain, int dtype);

/* Band-limited resampling by a rational ratio in the Fourier domain: the library's own converter behind librosa.resample's soxr_* /
 * kaiser_* / sinc_* names (librosa/core/audio.py:1146-1170; packages that are not in the build image) when the ratio is not a plain
 * decimation.  Each clip is zero-padded to fft_in samples, transformed, multiplied by the low-pass erfc((k - k_mid) / k_sigma) / 2 (bins of
 * the forward transform), cut at the lower Nyquist, inverted at fft_out samples; out[clip] = the first n_out samples * gain * fft_out /
 * fft_in.  The caller chooses fft_in = g down >= n_in + the filter's length and fft_out = g up (then the circular convolution is the linear
 * one and the output grid is exact).  x: [batch][n_in], out: [batch][n_out] real of `dtype` (device). */
int lra_resample_band_exec(lra_ctx* ctx, const void* x, void* out, int64_t batch, int64_t n_in, int64_t n_out, int64_t fft_in, int64_t fft_out, double k_mid, double k_sigma, double gain,
                           int dtype);

/* out[clip][t][bin0 + r] = (sum_j val[j] D[clip][t][col[j]], j in row row0 + r of the CSR basis) / sqrt_len[r], 0 <= r < n_rows,
 * 0 <= t < n_frames.  D: [clip][frames_in][n_bins] complex (lra_stft_exec's layout), out: [clip][n_frames][n_total] complex, both of
 * `dtype`'s precision; row_ptr / col: int32, val: complex (device); sqrt_len: float64 [n_rows] (device) or NULL (scale=False). */
int lra_cqt_project_exec(lra_ctx* ctx, const void* D, void* out, const void* row_ptr, const void* col, const void* val, const void* sqrt_len, int64_t batch, int64_t frames_in, int n_bins,
                         int64_t n_frames, int n_total, int bin0, int row0, int n_rows, int dtype);

/* One octave of the recursion in one launch (round 4): the rectangular-window STFT of centred frames (n_fft a power of two in [32, 4096]:
 * lra_cqt_octave_supported; constantq.py:1197-1212), the sparse basis projection (:1213-1218), the length scaling (:1116-1118) and the stacking
 * (:1168-1194) -- arguments as lra_cqt_project_exec, y [batch][y_stride] instead of D.  The frame spectra stay in LDS.  Raises the context's
 * non-finite flag like the forward kernels (a frame's DC bin is non-finite iff one of its samples is). */
int lra_cqt_octave_supported(int n_fft);
/* The whole octave recursion of one cqt / vqt call in ONE native call (round 5; constantq.py:1054-1099): for every octave lra_cqt_octave_exec on the
 * current signal and -- where `halve` -- lra_fir_decimate_exec by two (taps / first as there, div = sqrt(1/2): audio.resample(scale=True)) into the
 * next slice of `scratch` (>= the sum of the 256-byte-rounded halved signals).  overlap != 0: the octave transforms go to the context's side stream
 * beside the chain of halvings (lra_ctx_side) and are joined before returning.  What the Python loop did with ~40 calls through ctypes per transform. */
typedef struct lra_cqt_octave {
    int n_fft, hop;             /* frame length and hop of the octave's transform */
    int bin0, row0, n_rows;     /* its columns of the stacked result, its rows of the CSR basis */
    int halve;                  /* != 0: the signal is halved behind this octave */
    int64_t n;                  /* samples per clip of the octave's signal */
    const void* row_ptr;        /* CSR basis of the octave (device): int32 row_ptr / col, complex val */
    const void* col;
    const void* val;
} lra_cqt_octave;
int lra_cqt_recursion_exec(lra_ctx* ctx, const void* y, int64_t batch, const lra_cqt_octave* octaves, int n_octaves, int pad_mode, const void* sqrt_len, void* out, int64_t n_frames,
                           int n_total, const void* taps, int n_taps, int first, void* scratch, int64_t scratch_bytes, int overlap, int dtype);
int lra_cqt_octave_exec(lra_ctx* ctx, const void* y, int64_t batch, int64_t n, int64_t y_stride, int n_fft, int hop, int pad_mode, const void* row_ptr, const void* col, const void* val,
                        const void* sqrt_len, void* out, int64_t n_frames, int n_total, int bin0, int row0, int n_rows, int dtype);

/* ---- harmonic / percussive separation: librosa.decompose.hpss, librosa/decompose.py:371-528 (between the stft and the two istft of
 * librosa.effects.hpss / harmonic / percussive, librosa/effects.py:70-301) --------------------------------------------------------- */
/* mag[i] = |D[i]| (np.abs of the complex spectrogram, core/spectrum.py:1347); D complex, mag real of `dtype`'s precision (device). */
int lra_magnitude_exec(lra_ctx* ctx, const void* D, void* mag, int64_t count, int dtype);

/* librosa.magphase, librosa/core/spectrum.py:1296-1361: mag[i] = |D[i]| ** power, phase[i] = D[i] / |D[i]| (1 + 0j where |D[i]| = 0; real and
 * imaginary parts divided separately, :1353-1357).  D: complex (is_complex != 0) or real of `dtype`'s precision; mag real, phase complex
 * (device).  The exponent as NumPy evaluates a scalar one (1, 2, 0.5, -1, 0: copy, square, sqrt, reciprocal, ones; else pow). */
int lra_magphase_exec(lra_ctx* ctx, const void* D, int is_complex, void* mag, void* phase, int64_t count, double power, int dtype);

/* mag: [clip][frame][bin] real (device; |D|, or any non-negative real spectrogram S).  One pass: harm / perc = medians over win_harm
 * frames / win_perc bins (scipy.ndimage.median_filter, mode="reflect", :499-510), the two soft masks (util.softmask with `power`,
 * margins, split_zeros = both margins 1, :512-520; power = inf: hard masks), and
 *   want_mask != 0:  out_h / out_p = the masks (real)                                                          (:522-523)
 *   D == NULL:       out_h / out_p = S * mask (real)                                                            (:528, phase = 1)
 *   else:            out_h / out_p = (|D| * mask) * D / |D| (complex, same layout; phase 1 + 0j where |D| == 0)  (:528, :471-472) */
int lra_hpss_exec(lra_ctx* ctx, const void* mag, const void* D, void* out_h, void* out_p, int64_t batch, int64_t n_frames, int n_bins, int win_harm, int win_perc, double power,
                  double margin_harm, double margin_perc, int want_mask, int dtype);

/* ---- multi-GPU: the trivial gather of the sharded result (the reference has no counterpart) ----------------- */
/* One process per GPU.  Clips shard by contiguous ranges with no collective on the data path; these entry points gather the
 * per-rank results over RCCL (xGMI) for hosts that do not use torch.distributed.  RCCL is bound at run time (dlopen): the
 * library does not depend on it.  Rank 0 obtains an id (LRA_COMM_ID_BYTES bytes) and the host program hands it to the other
 * ranks (MPI, a file, a socket); every rank then creates its communicator on its own context (= device + stream). */
#define LRA_COMM_ID_BYTES 128
typedef struct lra_comm lra_comm;
int lra_comm_unique_id(void* id_out);
int lra_comm_init(lra_ctx* ctx, int rank, int n_ranks, const void* id, lra_comm** out);
/* recv_dev[r * bytes_per_rank ...] = rank r's send_dev[0 .. bytes_per_rank), enqueued on the context's stream (stream-ordered
 * after the kernels that produced send_dev); equal shard sizes -- the shim gathers unequal shards piecewise. */
int lra_comm_allgather(lra_comm* comm, const void* send_dev, void* recv_dev, size_t bytes_per_rank);
/* The same for UNEQUAL shards (clips % ranks != 0: librosa_amd.distributed.shard_range): rank r's bytes_per_rank[r] bytes land at recv_dev + recv_offsets[r] on
 * every rank -- one ncclBroadcast per rank inside one ncclGroupStart / End, no padding, no staging copy.  Both tables are host arrays of n_ranks entries, the same
 * on every rank. */
int lra_comm_allgatherv(lra_comm* comm, const void* send_dev, void* recv_dev, const size_t* bytes_per_rank, const size_t* recv_offsets);
void lra_comm_destroy(lra_comm* comm);

/* ---- measurement aid (no reference counterpart): the transform's access stream without its arithmetic ------------------
 * direction 0: per row read `hop` float32 of PCM and write one row of n_fft/2 + 1 complex64 (the shape of the array
 * librosa/core/spectrum.py:356 allocates); direction 1: read a row, write `hop` samples (the columns :598 reads).  Same
 * decomposition as the fused kernels (one wave64 per strip of `strip_rows` consecutive rows, 8-byte pieces, next row's loads
 * ahead of this row's stores), `waves_per_cu` resident waves per CU (0 = 12, the forward kernel's residency).  bench.py
 * reports it as `stream_ceiling`: what this mix of reads and 8-byte-aligned row writes reaches on the chip at hand
 * (SURVEY.md 8d prices the kernels against 8 TB/s).  n_fft in {256, 512, ..., 2048}, hop a multiple of 128. */
int lra_probe_stream(lra_ctx* ctx, int direction, const void* in, void* out, int64_t batch, int64_t rows_per_clip, int n_fft, int hop, int64_t clip_samples,
                     int strip_rows, int waves_per_cu);
/* ... with the rows `row_pitch_bytes` apart (0 = packed: (n_fft/2 + 1) * 8; a multiple of 8) and the row side moved in pieces of
 * `piece_bytes`: 8 = the kernels' own butterfly order (bins k ascending from one base, M - k descending from the other), 16 = wave-wide
 * 1 KiB pieces in address order + bin M (needs a pitch that is a multiple of 16). */
int lra_probe_stream_pitched(lra_ctx* ctx, int direction, const void* in, void* out, int64_t batch, int64_t rows_per_clip, int n_fft, int hop, int64_t clip_samples,
                             int strip_rows, int waves_per_cu, int64_t row_pitch_bytes, int piece_bytes);

/* ... the forward stream with strips of `strip_rows` rows dealt out in address order to n_cu x waves_per_cu persistent waves (all of them writing
 * inside one moving window of the result): the locality experiment of profiles/r05_pitch.md. */
int lra_probe_stream_window(lra_ctx* ctx, const void* in, void* out, int64_t batch, int64_t rows_per_clip, int n_fft, int hop, int64_t clip_samples, int strip_rows, int waves_per_cu);

/* ---- layout helper: dst[b][c][r] = src[b][r][c], elem_bytes in {4, 8, 16} ------------------ */
int lra_transpose(lra_ctx* ctx, const void* src, void* dst, int64_t batch, int64_t rows, int64_t cols, int elem_bytes);

#ifdef __cplusplus
}
#endif
#endif /* LIBROSA_AMD_H */
