/* TEST INFRASTRUCTURE - CPU oracle of the radius-graph construction.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product path never does.
 *
 * Restates, in the reference's arithmetic,
 *     pwd = sklearn.metrics.pairwise_distances(X, Y)            (float64, metric = euclidean)
 *     edge_index = np.vstack(np.where(pwd <= r))
 * as called by SquareMeshGenerator.ball_connectivity (/root/reference/graph-neural-operator/utilities.py:250-255)
 * and RandomMultiMeshGenerator.ball_connectivity (/root/reference/multipole-graph-neural-operator/utilities.py:602-640).
 * scikit-learn (1.7.2 in this image; the reference pins no version) evaluates the distance by the dot-product
 * expansion   d2 = ((-2 * <x, y>) + |x|^2) + |y|^2 ;  d2 = max(d2, 0) ;  diagonal := 0 when Y is X ;  d = sqrt(d2)
 * with <x, y> from BLAS dgemm (an FMA chain over k starting at 0) and the squared norms from
 * sklearn.utils.extmath.row_norms (einsum: rounded products, summed).  Pairs at exactly distance r can fall on
 * either side of `<= r` under this rounding, which makes the reference's graphs slightly asymmetric (SURVEY.md §8a:
 * s = 61, r = 0.10 gives 376,471 edges, the exact count is 386,221).  This arithmetic was identified by comparing
 * candidates against the reference's own generator (tests/golden/make_golden.py case 9 -> tests/golden/mesh_ties.npz:
 * identical edge lists at s = 16, 31, 61); `exact` != 0 switches to the symmetric sum-of-squares test
 * sum_k (x_k - y_k)^2 <= r^2 the product uses by default.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC -o libradius_oracle.so radius_oracle.c -lm   (oracle/build_oracle.py)
 */
#include <math.h>
#include <stdint.h>

/* edges (i in X) -> (j in Y) in row-major np.where order; returns the count, writes at most `cap` pairs */
long gpde_oracle_radius_edges(const double* X, long nx, const double* Y, long ny, int dim, double r, int same_set,
                              int exact, int64_t* src, int64_t* dst, long cap) {
    long c = 0;
    for (long i = 0; i < nx; ++i) {
        double xx = 0.0;
        for (int k = 0; k < dim; ++k) xx = xx + X[i * dim + k] * X[i * dim + k];
        for (long j = 0; j < ny; ++j) {
            int hit;
            if (exact) {
                double d2 = 0.0;
                for (int k = 0; k < dim; ++k) { const double d = Y[j * dim + k] - X[i * dim + k]; d2 = d2 + d * d; }
                hit = d2 <= r * r;
            } else {
                double yy = 0.0, dot = 0.0;
                for (int k = 0; k < dim; ++k) yy = yy + Y[j * dim + k] * Y[j * dim + k];
                for (int k = 0; k < dim; ++k) dot = fma(X[i * dim + k], Y[j * dim + k], dot);
                double d2 = -2.0 * dot;
                d2 = d2 + xx;
                d2 = d2 + yy;
                if (d2 < 0.0) d2 = 0.0;
                if (same_set && i == j) d2 = 0.0;
                hit = sqrt(d2) <= r;
            }
            if (hit) {
                if (c < cap) { src[c] = i; dst[c] = j; }
                ++c;
            }
        }
    }
    return c;
}
