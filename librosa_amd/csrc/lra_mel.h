// lra_mel.h -- host-side analysis of a mel filterbank (librosa/filters.py:116-251).
//
// A triangular bank has at most two non-zero filters per FFT bin: bin k sits on the FALLING slope
// of one filter and on the RISING slope of the next.  "Two-slope form" stores exactly that:
//     wA[k] = weight of bin k in the filter it falls in,   wB[k] = weight in the filter it rises in,
// and, per filter m, the contiguous bin ranges of its rising part (taken from wB) and falling part
// (taken from wA).  mel[m] = sum_{k in rise(m)} wB[k] p[k] + sum_{k in fall(m)} wA[k] p[k].
// The weights are the reference's values bit for bit; only the storage is sparse (2 n_bins floats
// instead of n_mels x n_bins).  Any basis that does not have this structure (more than two filters on
// a bin, non-contiguous supports) is served by the generic banded path instead.
#pragma once

#include <vector>

namespace lra {

template <class T> struct TwoSlope {
    bool ok = false;
    std::vector<T> wA, wB;   // [n_bins]
    std::vector<int> rng;    // [2 n_mels]: id 2m = rising part (array B), 2m+1 = falling part (array A);
                             // packed start | len << 12 | array << 24 (array 1 = B)
};

template <class T> inline TwoSlope<T> build_two_slope(const T* B, int n_mels, int n_bins) {
    TwoSlope<T> ts;
    if (n_bins > 4095 + 1) return ts;  // packing limit (n_fft <= 8190); larger sizes use the generic path
    ts.wA.assign(n_bins, (T)0);
    ts.wB.assign(n_bins, (T)0);
    std::vector<int> rowA(n_bins, -1), rowB(n_bins, -1), peak(n_mels, 0);
    for (int m = 0; m < n_mels; ++m) {
        int best = 0;
        for (int k = 1; k < n_bins; ++k)
            if (B[(size_t)m * n_bins + k] > B[(size_t)m * n_bins + best]) best = k;
        peak[m] = best;
    }
    for (int k = 0; k < n_bins; ++k) {
        int rows[3], cnt = 0;
        for (int m = 0; m < n_mels; ++m)
            if (B[(size_t)m * n_bins + k] != (T)0) {
                if (cnt == 2) return ts;  // three filters on one bin
                rows[cnt++] = m;
            }
        if (cnt == 2) {
            rowA[k] = rows[0];
            rowB[k] = rows[1];
        } else if (cnt == 1) {
            if (k <= peak[rows[0]]) rowB[k] = rows[0];
            else rowA[k] = rows[0];
        }
        if (rowA[k] >= 0) ts.wA[k] = B[(size_t)rowA[k] * n_bins + k];
        if (rowB[k] >= 0) ts.wB[k] = B[(size_t)rowB[k] * n_bins + k];
    }
    ts.rng.assign(2 * (size_t)n_mels, 0);
    for (int m = 0; m < n_mels; ++m) {
        for (int part = 0; part < 2; ++part) {  // 0: rising (B), 1: falling (A)
            const std::vector<int>& owner = part == 0 ? rowB : rowA;
            int first = -1, last = -1, count = 0;
            for (int k = 0; k < n_bins; ++k)
                if (owner[k] == m) {
                    if (first < 0) first = k;
                    last = k;
                    ++count;
                }
            if (count && last - first + 1 != count) return ts;  // support is not one contiguous run
            const int start = count ? first : 0, len = count;
            ts.rng[2 * m + part] = start | (len << 12) | ((part == 0 ? 1 : 0) << 24);
        }
    }
    ts.ok = true;
    return ts;
}

}  // namespace lra
