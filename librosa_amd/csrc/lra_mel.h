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

#include <vector>

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

}  // namespace lra
