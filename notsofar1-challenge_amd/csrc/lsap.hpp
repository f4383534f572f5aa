// scipy.optimize.linear_sum_assignment for up to 4 x 4 costs, restated with its tie rules (see stitch.hip, "Sequential
// permutation scan"): the permutation solver of training/losses.py:43, shared by the stitching scan (host and device)
// and the validation loss.
#pragma once
#include <hip/hip_runtime.h>

namespace css {

constexpr int SMAX = 4;

// four values addressed by a RUN-TIME index through selects: the algorithm below indexes its work arrays by data, and
// arrays indexed that way would live in private scratch on the device
template <typename T>
struct V4 {
    T a, b, c, d;
    __host__ __device__ __forceinline__ T get(int i) const { return i == 0 ? a : (i == 1 ? b : (i == 2 ? c : d)); }
    __host__ __device__ __forceinline__ void set(int i, T v) {
        a = i == 0 ? v : a; b = i == 1 ? v : b; c = i == 2 ? v : c; d = i == 3 ? v : d;
    }
    __host__ __device__ __forceinline__ void fill(T v) { a = b = c = d = v; }
};

// The shortest-augmenting-path algorithm of scipy's rectangular_lsap (Crouse 2016), step by step: rows are assigned in
// order; each search scans the remaining columns in DESCENDING order (the list starts reversed and a retired entry is
// replaced by the last one), a column on a strictly shorter path wins, and among equally short ones an UNASSIGNED column
// wins -- so on exact ties it returns the assignment scipy returns.  c.get(a).get(b) = cost of row a -> column b;
// out: col4row.  Non-finite costs ("infeasible": scipy raises) keep the rows' own columns; every loop is bounded by the
// compile-time size N, so the device code is straight-line selects: no scratch, nothing to spin on.
template <int N>
__host__ __device__ __forceinline__ void lsap_fixed(const V4<V4<double>>& c, V4<int>& col4row) {
    const double inf = 1.0 / 0.0;
    V4<double> u, v, spc;
    V4<int> path, row4col, remaining;
    V4<bool> sr, sc;
    u.fill(0.0); v.fill(0.0); path.fill(-1); row4col.fill(-1); col4row.fill(-1);
    bool failed = false;
#pragma unroll
    for (int cur = 0; cur < N; ++cur) {
        int num_remaining = N;
        remaining = V4<int>{N - 1, N - 2, N - 3, N - 4};
        sr.fill(false); sc.fill(false); spc.fill(inf);
        double min_val = 0.0;
        int i = cur, sink = -1;
#pragma unroll
        for (int step = 0; step < N; ++step) {   // (every step retires one column: at most N)
            if (sink != -1 || failed) continue;
            int index = -1;
            double lowest = inf;
            sr.set(i, true);
            // (row i element by element: a select between whole rows would be lowered to an indexed stack array)
            const V4<double> ci{i == 0 ? c.a.a : (i == 1 ? c.b.a : (i == 2 ? c.c.a : c.d.a)), i == 0 ? c.a.b : (i == 1 ? c.b.b : (i == 2 ? c.c.b : c.d.b)),
                                i == 0 ? c.a.c : (i == 1 ? c.b.c : (i == 2 ? c.c.c : c.d.c)), i == 0 ? c.a.d : (i == 1 ? c.b.d : (i == 2 ? c.c.d : c.d.d))};
            const double ui = u.get(i);
#pragma unroll
            for (int it = 0; it < N; ++it) {
                if (it >= num_remaining) continue;
                const int j = remaining.get(it);
                const double r = min_val + ci.get(j) - ui - v.get(j);
                if (r < spc.get(j)) { path.set(j, i); spc.set(j, r); }
                const double sj = spc.get(j);
                if (sj < lowest || (sj == lowest && row4col.get(j) == -1)) { lowest = sj; index = it; }
            }
            min_val = lowest;
            if (index < 0 || !(lowest < inf)) { failed = true; continue; }
            const int j = remaining.get(index);
            if (row4col.get(j) == -1) sink = j;
            else i = row4col.get(j);
            sc.set(j, true);
            --num_remaining;
            remaining.set(index, remaining.get(num_remaining));
        }
        if (sink == -1) failed = true;
        if (failed) continue;
        u.set(cur, u.get(cur) + min_val);
#pragma unroll
        for (int k = 0; k < N; ++k)
            if (sr.get(k) && k != cur) u.set(k, u.get(k) + (min_val - spc.get(col4row.get(k))));   // (scipy: u[i] += minVal - spc[...]: this association)
#pragma unroll
        for (int k = 0; k < N; ++k)
            if (sc.get(k)) v.set(k, v.get(k) - (min_val - spc.get(k)));
        int j = sink;
        bool walking = true;
#pragma unroll
        for (int step = 0; step < N; ++step) {   // (the alternating path back to `cur` visits a row at most once)
            if (!walking) continue;
            const int r = path.get(j);
            row4col.set(j, r);
            const int t = col4row.get(r);
            col4row.set(r, j);
            j = t;
            walking = r != cur;
        }
    }
    if (failed) col4row = V4<int>{0, 1, 2, 3};
}

__host__ __device__ __forceinline__ void lsap_small(const V4<V4<double>>& c, int n, V4<int>& col4row) {
    switch (n) {
        case 1: col4row = V4<int>{0, 1, 2, 3}; break;
        case 2: lsap_fixed<2>(c, col4row); break;
        case 3: lsap_fixed<3>(c, col4row); break;
        default: lsap_fixed<4>(c, col4row); break;
    }
}

}  // namespace css
