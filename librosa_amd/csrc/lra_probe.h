// lra_probe.h -- the access stream of the forward / inverse transform WITHOUT its arithmetic: a measurement aid, not a transform.
//
// SURVEY.md 8(d) prices the complex STFT at 2 048 B of PCM read + one 8 200-byte row (1025 complex64, 8-byte aligned only) written
// per frame, and the inverse at the same bytes in the other direction.  What fraction of the 8 TB/s HBM peak that mix of reads and
// 8-byte-aligned row writes can reach at all on a given chip -- no FFT, no LDS traffic, the loads of row r + 1 issued before the
// stores of row r exactly as the kernels prefetch one frame ahead -- is what these kernels measure, with the kernels' own
// decomposition: one wave64 walks a strip of consecutive rows, a row leaves as 16 x global_store_dwordx2 (bins k ascending from one
// base, M - k descending from the other) + the middle bin, twelve waves per CU (the forward kernel's residency: bounded here by an
// LDS pad), strips of one clip on one XCD.  Round 2 found this "strip" form the fastest of six decompositions
// (profiles/r02_store_stream.md); bench.py reports it next to the transform as `stream_ceiling`, on the same box, batch and clock ramp.
//
// Reference semantics: none -- librosa has no counterpart; the shapes are those of librosa/core/spectrum.py:356 (the STFT's output
// array) and :598 (the columns istft reads).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace lra {

struct ProbeArgs {
    const char* in;          // forward: PCM [batch][clip_in_bytes]; inverse: rows [batch][rows][row_bytes]
    char* out;               // forward: rows; inverse: PCM
    long long clip_in_bytes;   // bytes between clips on the PCM side
    int rows_per_clip, strip_rows, n_clips;
    int pcm_rows;            // rows whose hop_bytes of PCM lie inside the clip (the transform pads the rest; the probe skips it)
    int bins;                // complex64 per row (n_fft / 2 + 1)
    int hop_bytes;           // PCM bytes per row (hop * 4)
    int xcd_chunk;           // > 0: block b -> (b % 8) * xcd_chunk + b / 8
    long long row_pitch;     // bytes between consecutive rows (bins * 8 = packed; round 5: a multiple of 128 puts every row on a cache-line boundary)
};

// DIR 0: forward stream (read hop_bytes, write one row); DIR 1: inverse stream (read one row, write hop_bytes).
// One wave per workgroup; R8 = 8-byte pieces per lane on the PCM side (hop_bytes / 512), at most 8.
// P16 (round 5, needs a row pitch that is a multiple of 16): the row side moves as M / 128 wave-wide 16-byte pieces of 1 KiB each in
// address order + bin M as lane 0's 8-byte tail (global_store_dwordx4 / global_load_dwordx4) instead of the kernels' 2 x M / 128
// 8-byte pieces in butterfly order -- what a transform could do after exchanging neighbouring bins between lane pairs.
template <int DIR, bool P16> __global__ __launch_bounds__(64) void stream_probe_kernel(ProbeArgs a) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    extern __shared__ char probe_pad[];  // (residency bound only)
    const int lane = threadIdx.x;
    long long b = blockIdx.x;
    if (a.xcd_chunk > 0) b = (b % 8) * (long long)a.xcd_chunk + b / 8;
    const int strips_per_clip = (a.rows_per_clip + a.strip_rows - 1) / a.strip_rows;
    const long long clip = b / strips_per_clip;
    const int part = (int)(b % strips_per_clip);
    if (clip >= a.n_clips) return;
    const int M = a.bins - 1;                 // bins k and M - k pair up, bin M / 2 is lane 0's extra piece
    const int pieces = M / 128;                // 8-byte pieces per lane on each side (8 at n_fft = 2048); the host checks M % 128 == 0, M <= 1024
    const int pcm8 = a.hop_bytes / 512;        // 8-byte pieces per lane of the PCM side (4 at hop 512)
    const long long row_bytes = a.row_pitch;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f2 v = {(float)lane, (float)b};
    if (DIR == 0) {
        f2 cur[8], nx[8];
        for (int c = 0; c < 8; ++c) cur[c] = v;
        for (int it = 0; it < a.strip_rows; ++it) {
            const int row = part * a.strip_rows + it;
            if (row >= a.rows_per_clip) break;
            for (int c = 0; c < 8; ++c) nx[c] = cur[c];
            if (row + 1 < a.pcm_rows) {
                const f2* src = reinterpret_cast<const f2*>(a.in + clip * a.clip_in_bytes + (long long)(row + 1) * a.hop_bytes) + lane;
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (c < pcm8) nx[c] = src[c * 64];
            }
            v.x += cur[0].x + cur[7].y;
            f2* rp = reinterpret_cast<f2*>(a.out + (clip * a.rows_per_clip + row) * row_bytes);
            if (P16) {
                f4* p4 = reinterpret_cast<f4*>(rp) + lane;
                const f4 w = {v.x, v.y, v.x, v.y};
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (i < pieces) p4[i * 64] = w;
                if (lane == 0) rp[M] = v;
            } else {
                f2* pk = rp + lane;
                f2* pm = rp + (M - lane);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (i < pieces) { pk[i * 64] = v; pm[-i * 64] = v; }  // bins lane + 64 i < M / 2 and their mirrors M - lane - 64 i > M / 2
                if (lane == 0) rp[M / 2] = v;
            }
            for (int c = 0; c < 8; ++c) cur[c] = nx[c];
        }
    } else {
        f2 acc = v;
        f2 cur[17], nx[17];
        for (int c = 0; c < 17; ++c) cur[c] = v;
        for (int it = 0; it < a.strip_rows; ++it) {
            const int row = part * a.strip_rows + it;
            if (row >= a.rows_per_clip) break;
            for (int c = 0; c < 17; ++c) nx[c] = cur[c];
            if (row + 1 < a.rows_per_clip) {
                const f2* rp = reinterpret_cast<const f2*>(a.in + (clip * a.rows_per_clip + row + 1) * row_bytes);
                if (P16) {
                    const f4* p4 = reinterpret_cast<const f4*>(rp) + lane;
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (i < pieces) { const f4 w = p4[i * 64]; nx[2 * i] = f2{w.x, w.y}; nx[2 * i + 1] = f2{w.z, w.w}; }
                    nx[16] = rp[M];
                } else {
                    const f2* pk = rp + lane;
                    const f2* pm = rp + (M - lane);
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (i < pieces) { nx[2 * i] = pk[i * 64]; nx[2 * i + 1] = pm[-i * 64]; }
                    nx[16] = rp[M / 2];
                }
            }
#pragma unroll
            for (int c = 0; c < 17; ++c) { acc.x += cur[c].x; acc.y += cur[c].y; }
            f2* dst = reinterpret_cast<f2*>(a.out + clip * a.clip_in_bytes + (long long)row * a.hop_bytes) + lane;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < pcm8 && row < a.pcm_rows) dst[c * 64] = acc;
            for (int c = 0; c < 17; ++c) cur[c] = nx[c];
        }
    }
}

// Round 5: the forward stream with the strips dealt out in ADDRESS ORDER to a fixed set of persistent waves -- worker w takes strips w, w + W, w + 2 W ... of
// `strip_rows` consecutive rows over the flattened (clip, row) range -- so that at any moment all W waves write inside one moving window of
// W x strip_rows x row_pitch bytes instead of 2 048 strips spread over the whole 2.7 GB result.  Question behind it (profiles/r05_pitch.md): is what makes
// some allocations slow the number of distinct pages / DRAM rows the concurrent strips touch?
__global__ __launch_bounds__(64) void stream_probe_window_kernel(ProbeArgs a, long long total_rows, int n_workers) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    extern __shared__ char probe_pad2[];
    const int lane = threadIdx.x;
    const int M = a.bins - 1, pieces = M / 128, pcm8 = a.hop_bytes / 512;
    f2 v = {(float)lane, (float)blockIdx.x};
    const long long n_strips = (total_rows + a.strip_rows - 1) / a.strip_rows;
    for (long long s = blockIdx.x; s < n_strips; s += n_workers) {
        const long long r0 = s * a.strip_rows;
        const long long r1 = r0 + a.strip_rows < total_rows ? r0 + a.strip_rows : total_rows;
        f2 cur[8], nx[8];
        for (int c = 0; c < 8; ++c) cur[c] = v;
        for (long long g = r0; g < r1; ++g) {
            const long long clip = g / a.rows_per_clip;
            const int row = (int)(g % a.rows_per_clip);
            for (int c = 0; c < 8; ++c) nx[c] = cur[c];
            if (row + 1 < a.pcm_rows) {
                const f2* src = reinterpret_cast<const f2*>(a.in + clip * a.clip_in_bytes + (long long)(row + 1) * a.hop_bytes) + lane;
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (c < pcm8) nx[c] = src[c * 64];
            }
            v.x += cur[0].x + cur[7].y;
            f2* rp = reinterpret_cast<f2*>(a.out + g * a.row_pitch);
            f2* pk = rp + lane;
            f2* pm = rp + (M - lane);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < pieces) { pk[i * 64] = v; pm[-i * 64] = v; }
            if (lane == 0) rp[M / 2] = v;
            for (int c = 0; c < 8; ++c) cur[c] = nx[c];
        }
    }
}

}  // namespace lra
