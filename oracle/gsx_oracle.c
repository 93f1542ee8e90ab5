/*
 * gsx_oracle.c -- CPU restatement (plain C) of the arithmetic on the
 * 3dgsconverter point-cloud filtering hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.
 *
 * Each function cites the reference lines (into /root/reference) or the
 * third-party algorithm it restates.  Build: see oracle/Makefile
 * (gcc -O2 -ffp-contract=off: no FMA contraction, the reference's numpy /
 * scipy wheels are baseline x86-64 builds without FMA).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ *
 * numpy pairwise summation, float32.
 * Third party: numpy (un-pinned in requirements.txt:3; 2.2.6 here),
 * numpy/_core/src/umath/loops_utils.h.src  @TYPE@_pairwise_sum:
 *   n < 8            : sequential
 *   n <= 128         : 8 accumulators, ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)),
 *                      then the n%8 tail sequentially
 *   else             : split at n/2 rounded down to a multiple of 8
 * Reached from data_processor.py:176-177 (np.mean / np.std of the f32
 * mean-distance array) and gpu_ops.py:259-260.
 * ------------------------------------------------------------------ */
static float pw_sum_f32(const float *a, int64_t n)
{
    if (n < 8) {
        float res = 0.0f;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        int64_t i;
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pw_sum_f32(a, n2) + pw_sum_f32(a + n2, n - n2);
    }
}

float gsxo_pairwise_sum_f32(const float *a, int64_t n) { return pw_sum_f32(a, n); }

/*
 * np.add.reduce over a 1-D contiguous float32 array as numpy 2.2.6 executes it:
 * the reduction iterator hands the inner loop buffer-sized pieces
 * (np.getbufsize() == 8192 elements) and the FLOAT_add reduce loop does
 *     *out += pairwise_sum(piece)
 * so the result is a SEQUENTIAL f32 accumulation of per-8192-element pairwise
 * sums.  [probed in the build container: 100/100 random lengths in
 * 60000..70000 match this and only 63/100 match a single whole-array tree.]
 */
#define NP_BUFSIZE 8192
float gsxo_np_sum_f32(const float *a, int64_t n)
{
    float acc = 0.0f;
    for (int64_t i = 0; i < n; i += NP_BUFSIZE) {
        int64_t m = n - i < NP_BUFSIZE ? n - i : NP_BUFSIZE;
        acc += pw_sum_f32(a + i, m);
    }
    return acc;
}

/* Same tree, summing (a[i]-m)^2 computed in f32 (numpy _var: x = arr - arrmean;
 * x = x*x; umr_sum(x)) without materialising x. */
static float pw_sumsq_f32(const float *a, int64_t n, float m)
{
#define SQ(v) (((v) - m) * ((v) - m))
    if (n < 8) {
        float res = 0.0f;
        for (int64_t i = 0; i < n; ++i) { float d = a[i] - m; res += d * d; }
        return res;
    } else if (n <= 128) {
        float r[8];
        int64_t i;
        for (int j = 0; j < 8; ++j) { float d = a[j] - m; r[j] = d * d; }
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) { float d = a[i + j] - m; r[j] += d * d; }
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) { float d = a[i] - m; res += d * d; }
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pw_sumsq_f32(a, n2, m) + pw_sumsq_f32(a + n2, n - n2, m);
    }
#undef SQ
}

/*
 * np.mean / np.std (ddof=0) of a 1-D float32 array and the SOR threshold.
 * Restates data_processor.py:176-178 (and gpu_ops.py:259-261):
 *     global_mean = np.mean(md); global_std = np.std(md)
 *     threshold   = global_mean + threshold_factor * global_std
 * numpy/_core/_methods.py _mean/_var: the f32 pairwise sum is divided by the
 * np.intp count in FLOAT64 and the quotient is cast back to f32; the python
 * float threshold_factor is a weak scalar => rounded to f32, f32 mul, f32 add.
 * out[0]=mean, out[1]=std, out[2]=threshold.
 */
void gsxo_sor_stats_f32(const float *md, int64_t n, double threshold_factor, float *out)
{
    float s = gsxo_np_sum_f32(md, n);
    float mean = (float)((double)s / (double)n);
    float ss = 0.0f;
    for (int64_t i = 0; i < n; i += NP_BUFSIZE) {
        int64_t m = n - i < NP_BUFSIZE ? n - i : NP_BUFSIZE;
        ss += pw_sumsq_f32(md + i, m, mean);
    }
    float var = (float)((double)ss / (double)n);
    float sd = sqrtf(var);
    float tf = (float)threshold_factor;
    float prod = tf * sd;
    out[0] = mean;
    out[1] = sd;
    out[2] = mean + prod;
}

/* ------------------------------------------------------------------ *
 * Row mean in numpy order (np.mean(dists[:, 1:], axis=1), float64),
 * data_processor.py:172.  d[0..k] ascending; d[0] (self) is dropped.
 * ------------------------------------------------------------------ */
static double pw_sum_f64(const double *a, int64_t n)
{
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        int64_t i;
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pw_sum_f64(a, n2) + pw_sum_f64(a + n2, n - n2);
    }
}

double gsxo_pairwise_sum_f64(const double *a, int64_t n) { return pw_sum_f64(a, n); }

/* ------------------------------------------------------------------ *
 * Exact brute-force KNN mean distance (O(N^2), scalar) -- restates
 * data_processor.py:160-173 with the arithmetic scipy.spatial.cKDTree
 * (un-pinned; 1.15.3 here) performs for p=2, m=3:
 *   ckdtree/src/distance.h sqeuclidean_distance_double:
 *       s = ((0 + dx*dx) + dy*dy) + dz*dz   in float64, inputs widened f32
 *   query.cxx: k+1 smallest by s, result sqrt(s), ascending.
 * mean_out[i] = (float) ( pairwise_sum(d[1..k]) / k ).
 * Missing neighbours (n < k+1) are +inf like cKDTree.query.
 * xyz is (n,3) row-major float32.
 * ------------------------------------------------------------------ */
void gsxo_sor_mean_dists_brute(const float *xyz, int64_t n, int k, float *mean_out)
{
    int kk = k + 1;
    double *best = (double *)malloc(sizeof(double) * (size_t)kk);
    for (int64_t i = 0; i < n; ++i) {
        double qx = xyz[3 * i], qy = xyz[3 * i + 1], qz = xyz[3 * i + 2];
        for (int t = 0; t < kk; ++t) best[t] = INFINITY;
        for (int64_t j = 0; j < n; ++j) {
            double dx = qx - (double)xyz[3 * j];
            double dy = qy - (double)xyz[3 * j + 1];
            double dz = qz - (double)xyz[3 * j + 2];
            double s = 0.0;
            s += dx * dx;
            s += dy * dy;
            s += dz * dz;
            if (s < best[kk - 1]) {
                int p = kk - 1;
                while (p > 0 && best[p - 1] > s) { best[p] = best[p - 1]; --p; }
                best[p] = s;
            }
        }
        for (int t = 0; t < kk; ++t) best[t] = sqrt(best[t]);
        double sum = pw_sum_f64(best + 1, k);
        mean_out[i] = (float)(sum / (double)k);
    }
    free(best);
}

/* Same, restricted to a list of query indices (used to spot-check big clouds). */
void gsxo_sor_mean_dists_brute_subset(const float *xyz, int64_t n, int k,
                                      const int64_t *qidx, int64_t nq, float *mean_out)
{
    int kk = k + 1;
    double *best = (double *)malloc(sizeof(double) * (size_t)kk);
    for (int64_t qi = 0; qi < nq; ++qi) {
        int64_t i = qidx[qi];
        double qx = xyz[3 * i], qy = xyz[3 * i + 1], qz = xyz[3 * i + 2];
        for (int t = 0; t < kk; ++t) best[t] = INFINITY;
        for (int64_t j = 0; j < n; ++j) {
            double dx = qx - (double)xyz[3 * j];
            double dy = qy - (double)xyz[3 * j + 1];
            double dz = qz - (double)xyz[3 * j + 2];
            double s = 0.0;
            s += dx * dx;
            s += dy * dy;
            s += dz * dz;
            if (s < best[kk - 1]) {
                int p = kk - 1;
                while (p > 0 && best[p - 1] > s) { best[p] = best[p - 1]; --p; }
                best[p] = s;
            }
        }
        for (int t = 0; t < kk; ++t) best[t] = sqrt(best[t]);
        double sum = pw_sum_f64(best + 1, k);
        mean_out[qi] = (float)(sum / (double)k);
    }
    free(best);
}

/* ------------------------------------------------------------------ *
 * Voxel key of the density filter, data_processor.py:38-39:
 *     np.floor(coords / voxel_size).astype(np.int64)
 * coords is float32 and voxel_size a python float (weak) => the divide is an
 * IEEE float32 divide by (float)voxel_size, floor in f32, then widened.
 * ------------------------------------------------------------------ */
void gsxo_voxel_keys(const float *xyz, int64_t n, double voxel_size, int64_t *keys)
{
    float v = (float)voxel_size;
    for (int64_t i = 0; i < 3 * n; ++i) keys[i] = (int64_t)floorf(xyz[i] / v);
}

/* ------------------------------------------------------------------ *
 * Lloyd iteration of the reference's Taichi kernels with an injected init
 * (gpu_ops.py:57-96, 178-191).  PINNED: gsxo_kmeans_assign + gsxo_kmeans_update_f32seq
 * reproduce, bit for bit, what the reference's own kernels return when executed
 * through oracle/taichi_shim.py (tests/golden/kmeans_ref.npz):
 *   assign: dist accumulated over dims in f32, strict '<' => lowest index wins,
 *           min_dist starts at 1e20f;
 *   update: centroids zeroed, sum in point order (the reference uses f32
 *           atomics in arbitrary order), divide by count, empty cluster => 0.
 * ------------------------------------------------------------------ */
void gsxo_kmeans_assign(const float *data, int64_t n, int d, const float *cent, int k, int32_t *labels)
{
    for (int64_t i = 0; i < n; ++i) {
        float min_dist = 1e20f;
        int best = -1;
        for (int c = 0; c < k; ++c) {
            float dist = 0.0f;
            for (int t = 0; t < d; ++t) {
                float diff = data[i * d + t] - cent[(int64_t)c * d + t];
                dist += diff * diff;
            }
            if (dist < min_dist) { min_dist = dist; best = c; }
        }
        labels[i] = best;
    }
}

void gsxo_kmeans_update(const float *data, int64_t n, int d, const int32_t *labels, int k,
                        float *cent, int32_t *counts)
{
    /* accumulate in double then round once: the order-independent centre of the
     * f32-atomic results the reference can produce */
    double *acc = (double *)calloc((size_t)k * (size_t)d, sizeof(double));
    memset(counts, 0, sizeof(int32_t) * (size_t)k);
    for (int64_t i = 0; i < n; ++i) {
        int l = labels[i];
        for (int t = 0; t < d; ++t) acc[(int64_t)l * d + t] += (double)data[i * d + t];
        counts[l] += 1;
    }
    for (int c = 0; c < k; ++c) {
        if (counts[c] > 0) {
            float inv = 1.0f / (float)counts[c];
            for (int t = 0; t < d; ++t) cent[(int64_t)c * d + t] = (float)acc[(int64_t)c * d + t] * inv;
        } else {
            for (int t = 0; t < d; ++t) cent[(int64_t)c * d + t] = 0.0f;
        }
    }
    free(acc);
}

/* gpu_ops.py:75-96 (k_means_update) executed in index order with binary32 accumulation: reset,
 * centroids[l] += data[i] for i = 0..n-1 (one rounding per add), then centroids[c] *= 1/count.
 * This is the order oracle/taichi_shim.py runs the reference's kernel in -- ONE of the orders its
 * f32 atomics can take -- so it reproduces tests/golden/kmeans_ref.npz bit for bit. */
void gsxo_kmeans_update_f32seq(const float *data, int64_t n, int d, const int32_t *labels, int k,
                               float *cent, int32_t *counts)
{
    for (int c = 0; c < k; ++c) {
        for (int t = 0; t < d; ++t) cent[(int64_t)c * d + t] = 0.0f;
        counts[c] = 0;
    }
    for (int64_t i = 0; i < n; ++i) {
        int l = labels[i];
        if (l < 0) l += k; /* numpy's negative index: what the shimmed reference does with best_k = -1 */
        for (int t = 0; t < d; ++t) cent[(int64_t)l * d + t] += data[i * d + t];
        counts[l] += 1;
    }
    for (int c = 0; c < k; ++c) {
        if (counts[c] > 0) {
            float inv = 1.0f / (float)counts[c];
            for (int t = 0; t < d; ++t) cent[(int64_t)c * d + t] *= inv;
        }
    }
}
