// lra_mel.h -- host-side analysis of a mel filterbank (librosa/filters.py:116-251).
//
// A triangular bank has at most two non-zero filters per FFT bin: bin k sits on the FALLING slope
// of one filter (call its weight wA[k]) and on the RISING slope of the next (wB[k]).  The bins between
// two consecutive mel points form one "pair segment" p: they are exactly the falling part of filter
// p-1 and the rising part of filter p.  Hence, with A[k] = wA[k] P[k] and B[k] = wB[k] P[k],
//     mel[m] = sum_{k in seg(m)} B[k]  +  sum_{k in seg(m+1)} A[k],
// i.e. ONE pass over the 1025 bins producing two running sums per segment, instead of a 128 x 1025
// product that is 98.5 % zeros.  The weights are the reference's values bit for bit; only the storage
// is sparse.  Any basis without this structure (three filters on a bin, non-adjacent overlaps,
// non-contiguous supports) is served by the generic banded path instead.
//
// Device work split ("pieces"): every thread of a frame slot sums a RUN of `bpl` consecutive bins held
// in registers and emits one partial sum per segment that its run touches; a piece is a maximal
// interval of bins inside one run and one segment.  Pieces are numbered in bin order, so the pieces of
// a segment are consecutive and mel[m] is the sum of a short, fixed list of partials (deterministic
// order: ascending bins).
#pragma once

#include <algorithm>
#include <vector>

#include "lra_common.h"

namespace lra {

template <class T> struct TwoSlope {
    bool ok = false;
    int n_bins = 0, n_mels = 0;
    std::vector<T> wAB;      // [n_bins][2]: (wA, wB) interleaved
    std::vector<int> owner;  // [n_bins]: pair segment of the bin, -1 if the bin is in no filter
};

// per-configuration piece tables; `runs` = threads per frame slot (TF), bpl = bins per thread run.
// The last bin (index runs*bpl == n_bins-1, the Nyquist bin) is an extra one-bin run of thread 0.
struct MelPieces {
    int bpl = 0, runs = 0, n_pieces = 0, max_per_seg = 0;
    std::vector<int> run_desc;  // [runs]: first piece id | (mask of "piece ends after bin j", j < bpl-1) << 12
    std::vector<int> seg_desc;  // [n_mels + 1]: first piece id | count << 12
    int nyquist_piece = 0;
};

template <class T> inline TwoSlope<T> build_two_slope(const T* B, int n_mels, int n_bins) {
    TwoSlope<T> ts;
    ts.n_bins = n_bins;
    ts.n_mels = n_mels;
    if (n_bins > 4097 || n_mels < 1) return ts;  // packing limit (n_fft <= 8192); larger sizes use the generic path
    std::vector<int> rowA(n_bins, -1), rowB(n_bins, -1), peak(n_mels, 0);
    for (int m = 0; m < n_mels; ++m) {
        int best = 0;
        for (int k = 1; k < n_bins; ++k)
            if (B[(size_t)m * n_bins + k] > B[(size_t)m * n_bins + best]) best = k;
        peak[m] = best;
    }
    ts.wAB.assign(2 * (size_t)n_bins, (T)0);
    for (int k = 0; k < n_bins; ++k) {
        int rows[2], cnt = 0;
        for (int m = 0; m < n_mels; ++m)
            if (B[(size_t)m * n_bins + k] != (T)0) {
                if (cnt == 2) return ts;  // three filters on one bin
                rows[cnt++] = m;
            }
        if (cnt == 2) {
            if (rows[1] != rows[0] + 1) return ts;  // overlapping filters must be neighbours
            rowA[k] = rows[0];
            rowB[k] = rows[1];
        } else if (cnt == 1) {
            if (k <= peak[rows[0]]) rowB[k] = rows[0];
            else rowA[k] = rows[0];
        }
        if (rowA[k] >= 0) ts.wAB[2 * (size_t)k] = B[(size_t)rowA[k] * n_bins + k];
        if (rowB[k] >= 0) ts.wAB[2 * (size_t)k + 1] = B[(size_t)rowB[k] * n_bins + k];
    }
    // pair segment p = bins whose rising filter is p or whose falling filter is p-1
    ts.owner.assign(n_bins, -1);
    for (int k = 0; k < n_bins; ++k) {
        const int pb = rowB[k] >= 0 ? rowB[k] : -1, pa = rowA[k] >= 0 ? rowA[k] + 1 : -1;
        if (pb >= 0 && pa >= 0 && pa != pb) return ts;
        ts.owner[k] = pb >= 0 ? pb : pa;
    }
    for (int p = 0; p <= n_mels; ++p) {  // every segment must be one contiguous run of bins
        int first = -1, last = -1, count = 0;
        for (int k = 0; k < n_bins; ++k)
            if (ts.owner[k] == p) {
                if (first < 0) first = k;
                last = k;
                ++count;
            }
        if (count && last - first + 1 != count) return ts;
    }
    ts.ok = true;
    return ts;
}

template <class T> inline MelPieces build_mel_pieces(const TwoSlope<T>& ts, int runs, int bpl) {
    MelPieces mp;
    mp.bpl = bpl;
    mp.runs = runs;
    if (!ts.ok || runs * bpl + 1 != ts.n_bins || bpl < 1 || bpl > 16) return mp;
    mp.run_desc.assign(runs, 0);
    mp.seg_desc.assign((size_t)ts.n_mels + 1, 0);
    std::vector<int> seg_first(ts.n_mels + 1, -1), seg_count(ts.n_mels + 1, 0);
    int piece = 0;
    auto account = [&](int owner) {
        if (owner >= 0) {
            if (seg_first[owner] < 0) seg_first[owner] = piece;
            ++seg_count[owner];
        }
    };
    for (int r = 0; r < runs; ++r) {
        const int first_piece = piece;
        int mask = 0;
        for (int j = 0; j < bpl; ++j) {
            const int k = r * bpl + j;
            const bool end = j == bpl - 1 || ts.owner[k + 1] != ts.owner[k];
            if (end) {
                if (j < bpl - 1) mask |= 1 << j;
                account(ts.owner[k]);
                ++piece;
            }
        }
        if (first_piece > 0xfff) return MelPieces();
        mp.run_desc[r] = first_piece | (mask << 12);
    }
    mp.nyquist_piece = piece;
    account(ts.owner[ts.n_bins - 1]);
    ++piece;
    mp.n_pieces = piece;
    for (int p = 0; p <= ts.n_mels; ++p) {
        // pieces of a segment are consecutive in bin order, EXCEPT that the Nyquist piece comes last:
        // a segment that contains the Nyquist bin and earlier bins still has consecutive ids because all
        // bins before the Nyquist bin that belong to it are the final pieces of the last runs
        const int first = seg_first[p] < 0 ? 0 : seg_first[p];
        if (seg_count[p] > 15 || first > 0xfff) return MelPieces();
        mp.seg_desc[p] = first | (seg_count[p] << 12);
        if (seg_count[p] > mp.max_per_seg) mp.max_per_seg = seg_count[p];
    }
    return mp;
}

// ---- run-ordered epilogue (OUT_MELR) ------------------------------------------------------------------------
// With `bpl` = R/2 and TF threads per frame, thread tf owns, in REGISTERS right after the Hermitian split,
//     run A: bins  bpl tf + j           (ascending,  j = 0 .. bpl-1)   -- register slot jj = j
//     run B: bins  M - bpl tf - j       (descending, j = 0 .. bpl-1)   -- register slot jj = bpl + j
// (M = n_bins - 1; bin M/2 is an extra bin of thread 0).  It multiplies by (wA, wB), keeps a running sum
// along each run that restarts where the run enters another pair segment, and writes ALL 2 bpl running sums
// to LDS slot (jj TF + tf); a piece's total is the running sum at the piece's last slot.  mel[m] then adds
// the B components of the pieces of segment m and the A components of the pieces of segment m+1: two fixed
// lists of at most PMAX LDS addresses per mel band (unused entries point at a slot that holds zero).
// Nothing in the device code depends on per-lane control flow.
template <class T> struct MelRuns {
    bool ok = false;
    int tf = 0, bpl = 0, pmax = 0, max_pieces = 0;
    std::vector<T> w;        // [2][M/2] pairs: (wA, wB)[i] for i < M/2, then (wA, wB)[M - i] for i < M/2; then the pair of bin M/2
    std::vector<T> keep;     // [2 bpl][TF]: 0 where the running sum restarts, 1 elsewhere
    std::vector<int> addr;   // [2 pmax][n_mels]: byte addresses (entries 0..pmax-1: B components, then the A components)
    int zero_addr = 0, mid_addr = 0;  // byte addresses of the always-zero slot and of bin M/2's (A, B) slot
};

// layout 0 (first-generation kernels): run A = bins bpl t + j, run B = bins M - bpl t - j (mirrored), extra bin M/2;
// layout 1 (second-generation kernel, which reads its runs from a power row in LDS): run A = bins bpl t + j, run B = bins
// M/2 + bpl t + j (both ascending), extra bin M; the weight table is then in plain bin order.  Layout 1 also stores the running
// sums with a pitch of TF + 1 pairs per register slot: with a pitch of TF all 2 bpl sums of one thread share one pair of LDS banks,
// and the bands of the low mel range -- whose pieces all come from the first few threads -- gathered them with multi-way conflicts.
// (mel_runs_pitch: lra_common.h -- the kernels and this table builder share it)
template <class T> inline MelRuns<T> build_mel_runs(const TwoSlope<T>& ts, int tf_count, int bpl, int pmax, int min_len, int layout = 0) {
    MelRuns<T> mr;
    mr.tf = tf_count;
    mr.bpl = bpl;
    mr.pmax = pmax;
    const int M = ts.n_bins - 1;
    if (!ts.ok || bpl < 1 || 2 * bpl * tf_count != M || pmax < 1) return mr;
    const int pitch = mel_runs_pitch(tf_count, layout);
    const int half = M / 2, slots = 2 * bpl * pitch, pair_bytes = 2 * (int)sizeof(T);
    mr.zero_addr = slots * pair_bytes;
    mr.mid_addr = (slots + 1) * pair_bytes;
    if (mr.mid_addr + pair_bytes > 65535) return mr;
    mr.w.assign(2 * (size_t)(2 * half + 1), (T)0);
    const int extra_bin = layout == 0 ? half : M;
    for (int i = 0; i < half; ++i) {
        const int hi = layout == 0 ? M - i : half + i;  // bin behind weight-table entry half + i
        mr.w[2 * (size_t)i] = ts.wAB[2 * (size_t)i];
        mr.w[2 * (size_t)i + 1] = ts.wAB[2 * (size_t)i + 1];
        mr.w[2 * (size_t)(half + i)] = ts.wAB[2 * (size_t)hi];
        mr.w[2 * (size_t)(half + i) + 1] = ts.wAB[2 * (size_t)hi + 1];
    }
    mr.w[2 * (size_t)(2 * half)] = ts.wAB[2 * (size_t)extra_bin];
    mr.w[2 * (size_t)(2 * half) + 1] = ts.wAB[2 * (size_t)extra_bin + 1];
    // bin -> register slot; effective segment of a bin (bins outside every filter carry zero weights and simply
    // extend the neighbouring stretch)
    auto bin_of = [&](int t, int jj) {
        if (jj < bpl) return bpl * t + jj;
        return layout == 0 ? M - bpl * t - (jj - bpl) : half + bpl * t + (jj - bpl);
    };
    mr.keep.assign((size_t)2 * bpl * tf_count, (T)1);
    // pieces: (segment, byte address of the slot that holds the piece's total, lowest bin) in bin order per segment
    struct Piece { int seg, addr, lowbin; };
    std::vector<Piece> pieces;
    for (int t = 0; t < tf_count; ++t) {
        for (int run = 0; run < 2; ++run) {
            int cur = -2;  // segment of the current stretch (-2: none yet)
            for (int j = 0; j < bpl; ++j) {
                const int jj = run * bpl + j, k = bin_of(t, jj);
                const int seg = ts.owner[k];
                const bool restart = j == 0 || (seg >= 0 && cur >= 0 && seg != cur);
                if (restart) {
                    mr.keep[(size_t)jj * tf_count + t] = (T)0;
                    if (j > 0 && cur >= 0) {  // the previous slot closed a piece of segment cur
                        const int pj = jj - 1;
                        int lo = bin_of(t, pj), hi = lo;
                        pieces.push_back({cur, (pj * pitch + t) * pair_bytes, 0});
                        (void)lo; (void)hi;
                    }
                    cur = seg >= 0 ? seg : -2;
                } else if (seg >= 0 && cur < 0) {
                    cur = seg;  // first weighted bin of a stretch that began with unweighted bins
                }
            }
            if (cur >= 0) pieces.push_back({cur, ((run * bpl + bpl - 1) * pitch + t) * pair_bytes, 0});
        }
    }
    if (ts.owner[extra_bin] >= 0) pieces.push_back({ts.owner[extra_bin], mr.mid_addr, 0});
    // lowest bin of each piece (for ordering): recompute from the address
    for (auto& pc : pieces) {
        if (pc.addr == mr.mid_addr) { pc.lowbin = extra_bin; continue; }
        const int slot = pc.addr / pair_bytes, jj = slot / pitch, t = slot % pitch;
        pc.lowbin = bin_of(t, jj);  // run A: the last (highest) bin; run B: the last slot is the LOWEST bin -- either orders pieces consistently
    }
    std::vector<std::vector<Piece>> by_seg((size_t)ts.n_mels + 1);
    for (const auto& pc : pieces) by_seg[(size_t)pc.seg].push_back(pc);
    for (auto& v : by_seg) {
        std::sort(v.begin(), v.end(), [](const Piece& a, const Piece& b) { return a.lowbin < b.lowbin; });
        if ((int)v.size() > mr.max_pieces) mr.max_pieces = (int)v.size();
    }
    if (mr.max_pieces > pmax) return mr;
    pmax = mr.max_pieces < min_len ? min_len : mr.max_pieces;  // lists are stored with their actual length (>= the hoisted prefix)
    mr.pmax = pmax;
    mr.addr.assign((size_t)ts.n_mels * 2 * pmax, mr.zero_addr);  // [entry][mel]
    for (int m = 0; m < ts.n_mels; ++m) {
        int q = 0;
        for (const auto& pc : by_seg[(size_t)m]) mr.addr[(size_t)(q++) * ts.n_mels + m] = pc.addr + (int)sizeof(T);              // B component
        q = 0;
        for (const auto& pc : by_seg[(size_t)m + 1]) mr.addr[(size_t)(pmax + q++) * ts.n_mels + m] = pc.addr;                    // A component
    }
    mr.ok = true;
    return mr;
}

}  // namespace lra
