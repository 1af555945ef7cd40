/*
 * TEST INFRASTRUCTURE ONLY — CPU oracle for Monotonic Alignment Search (MAS).
 *
 * This file is a plain-C restatement of the reference algorithm and may only be
 * used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
 * checker / reported CPU baseline.  The product path (glow-tts_amd/) never links,
 * imports or calls it.
 *
 * Restates:
 *   reference monotonic_align/core.pyx:9-35   (maximum_path_each)
 *   reference monotonic_align/core.pyx:38-45  (maximum_path_c, batch loop; the
 *     reference's prange is serial because monotonic_align/setup.py:5-9 passes no
 *     OpenMP flags)
 *   reference monotonic_align/__init__.py:6-21 (value*mask, lengths from the mask)
 *
 * Parity status: PINNED — checked bit-exact against oracle/_ref (the reference
 * core.pyx itself, cythonized + compiled by oracle/Makefile) in
 * tests/test_mas_oracle.py, and against tests/golden/mas_golden.npz which was
 * produced by that same reference build (tests/golden/make_mas_golden.py).
 *
 * Arithmetic notes (from the Cython-generated C of core.pyx):
 *   max(v_cur, v_prev) lowers to  (v_prev > v_cur) ? v_prev : v_cur
 *   all arithmetic is IEEE binary32, one add per cell, no FMA.
 * Build with -ffp-contract=off (there is nothing to contract, but keep it exact).
 */
#include <stdint.h>
#include <stddef.h>

/* core.pyx:9-35.  value: [t_x_stride rows][t_y_stride cols] fp32, mutated in place
 * (it becomes the running maximum Q).  path: same shape int32, pre-zeroed. */
void mas_oracle_each(int32_t *path, float *value, int t_x, int t_y,
                     int64_t row_stride, float max_neg_val)
{
    int x, y;
    float v_prev, v_cur;
    int index = t_x - 1;

    for (y = 0; y < t_y; ++y) {
        int lo = t_x + y - t_y; if (lo < 0) lo = 0;          /* max(0, t_x+y-t_y) */
        int hi = (t_x < y + 1) ? t_x : (y + 1);              /* min(t_x, y+1)     */
        for (x = lo; x < hi; ++x) {
            if (x == y) v_cur = max_neg_val;
            else        v_cur = value[(int64_t)x * row_stride + (y - 1)];
            if (x == 0) v_prev = (y == 0) ? 0.0f : max_neg_val;
            else        v_prev = value[(int64_t)(x - 1) * row_stride + (y - 1)];
            {
                float m = (v_prev > v_cur) ? v_prev : v_cur;  /* Cython max(v_cur, v_prev) */
                value[(int64_t)x * row_stride + y] = m + value[(int64_t)x * row_stride + y];
            }
        }
    }
    if (t_x <= 0) return;   /* reference would write path[-1,y]; out of contract */
    for (y = t_y - 1; y >= 0; --y) {
        path[(int64_t)index * row_stride + y] = 1;
        if (index != 0 &&
            (index == y ||
             value[(int64_t)index * row_stride + (y - 1)] <
             value[(int64_t)(index - 1) * row_stride + (y - 1)]))
            index = index - 1;
    }
}

/* core.pyx:38-45 — serial batch loop. */
void mas_oracle_batch(int32_t *paths, float *values, const int32_t *t_xs,
                      const int32_t *t_ys, int b, int T_x, int T_y)
{
    int i;
    for (i = 0; i < b; ++i)
        mas_oracle_each(paths + (int64_t)i * T_x * T_y, values + (int64_t)i * T_x * T_y,
                        t_xs[i], t_ys[i], (int64_t)T_y, -1e9f);
}

/* The batch loop with one utterance per OpenMP thread: what the reference's `prange` would be had monotonic_align/setup.py
 * passed -fopenmp (it does not: core.pyx:38-45 runs serially).  bench.py's "all cores" CPU baseline; returns the threads used. */
#ifdef _OPENMP
#include <omp.h>
#endif
int mas_oracle_batch_omp(int32_t *paths, float *values, const int32_t *t_xs,
                         const int32_t *t_ys, int b, int T_x, int T_y, int n_threads)
{
    int i, used = 1;
#ifdef _OPENMP
    used = n_threads > 0 ? n_threads : omp_get_max_threads();     /* (num_threads clause only: the process-wide setting stays as it is) */
    if (used > b) used = b;
#pragma omp parallel for schedule(dynamic, 1) num_threads(used)
#endif
    for (i = 0; i < b; ++i)
        mas_oracle_each(paths + (int64_t)i * T_x * T_y, values + (int64_t)i * T_x * T_y,
                        t_xs[i], t_ys[i], (int64_t)T_y, -1e9f);
    return used;
}

/* Same as mas_oracle_batch for the utterances [i0, i1) (a caller that brings its own threads). */
void mas_oracle_range(int32_t *paths, float *values, const int32_t *t_xs,
                      const int32_t *t_ys, int i0, int i1, int T_x, int T_y)
{
    int i;
    for (i = i0; i < i1; ++i)
        mas_oracle_each(paths + (int64_t)i * T_x * T_y, values + (int64_t)i * T_x * T_y,
                        t_xs[i], t_ys[i], (int64_t)T_y, -1e9f);
}
