// scipy.optimize.linear_sum_assignment for up to 4 x 4 costs, restated with its tie rules (see stitch.hip, "Sequential
// permutation scan"): the permutation solver of training/losses.py:43, shared by the stitching scan (host and device)
// and the validation loss.
#pragma once
#include <hip/hip_runtime.h>

namespace css {

constexpr int SMAX = 4;

__host__ __device__ inline void lsap_small(const double c[SMAX][SMAX], int n, int col4row[SMAX]) {
    const double inf = 1.0 / 0.0;
    double u[SMAX], v[SMAX], spc[SMAX];
    int path[SMAX], row4col[SMAX], remaining[SMAX];
    bool sr[SMAX], sc[SMAX];
    for (int k = 0; k < SMAX; ++k) { u[k] = 0.0; v[k] = 0.0; path[k] = -1; row4col[k] = -1; col4row[k] = -1; }
    for (int cur = 0; cur < n; ++cur) {
        int num_remaining = n;
        for (int it = 0; it < n; ++it) { remaining[it] = n - it - 1; sr[it] = false; sc[it] = false; spc[it] = inf; }
        double min_val = 0.0;
        int i = cur, sink = -1;
        for (int step = 0; step < n && sink == -1; ++step) {   // (every step retires one column: at most n)
            int index = -1;
            double lowest = inf;
            sr[i] = true;
            for (int it = 0; it < num_remaining; ++it) {
                const int j = remaining[it];
                const double r = min_val + c[i][j] - u[i] - v[j];
                if (r < spc[j]) { path[j] = i; spc[j] = r; }
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
            }
            min_val = lowest;
            if (index < 0 || !(lowest < inf)) {   // non-finite costs ("infeasible": scipy raises): keep the rows' own columns
                for (int k = 0; k < n; ++k) col4row[k] = k;
                return;
            }
            const int j = remaining[index];
            if (row4col[j] == -1) sink = j;
            else i = row4col[j];
            sc[j] = true;
            remaining[index] = remaining[--num_remaining];
        }
        if (sink == -1) {   // cannot happen with finite costs; never loop on the device
            for (int k = 0; k < n; ++k) col4row[k] = k;
            return;
        }
        u[cur] += min_val;
        for (int k = 0; k < n; ++k)
            if (sr[k] && k != cur) u[k] += min_val - spc[col4row[k]];
        for (int k = 0; k < n; ++k)
            if (sc[k]) v[k] -= min_val - spc[k];
        int j = sink;
        for (int step = 0; step < n; ++step) {   // (the alternating path back to `cur` visits a row at most once)
            const int r = path[j];
            row4col[j] = r;
            const int t = col4row[r]; col4row[r] = j; j = t;
            if (r == cur) break;
        }
    }
}

}  // namespace css
