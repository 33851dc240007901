// Host-side index-map builders of lib_new/coarsening.py, natively (no GPU involved).
// The reference runs these as pure-Python loops (O(nnz) per level for the matching,
// O(M^2) for compute_perm through np.where per vertex); here they are linear time.
// Built with -ffp-contract=off: the matching weights must round exactly like NumPy's.
#include <stdint.h>
#include <vector>

#include "status.h"

namespace {

// coarsening.py:120-166.  Row extents come from the reference's counting loop, whose
// boundary test runs after the increment: the first run is one entry too long (it also
// sees the first entry of the second run), the last run one entry short; runs are
// numbered in order of appearance.  Ties keep the first maximum (strict >).
// T = storage type of the weights, S = type the score is evaluated in (S = double with T = float is
// NumPy 1.x's promotion of `python float / float32 scalar`; the conversions float -> double are exact).
template <typename T, typename S = T>
int metis_one_level(int64_t nnz, const int64_t* rr, const int64_t* cc, const T* vv, const int64_t* rid,
                    const T* weights, int64_t N, int32_t* cluster_id) {
    using chebgcn::fail;
    if (nnz <= 0 || !rr || !cc || !vv || !rid || !weights || !cluster_id || N <= 0)
        return fail(CHEBGCN_EINVAL, "metis_one_level: bad arguments");
    if (rr[nnz - 1] + 1 != N) return fail(CHEBGCN_EINVAL, "metis_one_level: N must equal rr[nnz-1]+1");
    std::vector<int64_t> rowstart(N, 0), rowlength(N, 0);
    std::vector<char> marked(N, 0);
    int64_t oldval = rr[0], count = 0;
    for (int64_t ii = 0; ii < nnz; ++ii) {
        rowlength[count] += 1;
        if (rr[ii] > oldval) {
            if (count + 1 >= N) return fail(CHEBGCN_EINVAL, "metis_one_level: rr is not sorted");
            oldval = rr[ii];
            rowstart[count + 1] = ii;
            count += 1;
        }
    }
    for (int64_t i = 0; i < N; ++i) cluster_id[i] = 0;
    int32_t clustercount = 0;
    for (int64_t ii = 0; ii < N; ++ii) {
        const int64_t tid = rid[ii];
        if (tid < 0 || tid >= N) return fail(CHEBGCN_EINVAL, "metis_one_level: rid[%lld] out of range", (long long)ii);
        if (marked[tid]) continue;
        S wmax = 0;
        const int64_t rs = rowstart[tid];
        marked[tid] = 1;
        int64_t best = -1;
        for (int64_t jj = 0; jj < rowlength[tid]; ++jj) {
            if (rs + jj >= nnz) return fail(CHEBGCN_EINVAL, "metis_one_level: row extent past nnz");
            const int64_t nid = cc[rs + jj];
            if (nid < 0 || nid >= N) return fail(CHEBGCN_EINVAL, "metis_one_level: column out of range");
            S tval;
            if (marked[nid]) {
                tval = 0;
            } else {
                const S a = S(1) / S(weights[tid]);
                const S b = S(1) / S(weights[nid]);
                const S s = a + b;
                tval = S(vv[rs + jj]) * s;
            }
            if (tval > wmax) {
                wmax = tval;
                best = nid;
            }
        }
        cluster_id[tid] = clustercount;
        if (best > -1) {
            cluster_id[best] = clustercount;
            marked[best] = 1;
        }
        clustercount += 1;
    }
    return CHEBGCN_OK;
}

}  // namespace

extern "C" int chebgcn_metis_one_level_f32(int64_t nnz, const int64_t* rr, const int64_t* cc, const float* vv,
                                           const int64_t* rid, const float* weights, int64_t N,
                                           int32_t* cluster_id) {
    return metis_one_level<float>(nnz, rr, cc, vv, rid, weights, N, cluster_id);
}

extern "C" int chebgcn_metis_one_level_f32p(int64_t nnz, const int64_t* rr, const int64_t* cc, const float* vv,
                                            const int64_t* rid, const float* weights, int64_t N,
                                            int32_t* cluster_id) {
    return metis_one_level<float, double>(nnz, rr, cc, vv, rid, weights, N, cluster_id);
}

extern "C" int chebgcn_metis_one_level_f64(int64_t nnz, const int64_t* rr, const int64_t* cc, const double* vv,
                                           const int64_t* rid, const double* weights, int64_t N,
                                           int32_t* cluster_id) {
    return metis_one_level<double>(nnz, rr, cc, vv, rid, weights, N, cluster_id);
}

// One level of coarsening.py:168-215: for every entry of `order` (a vertex id of the
// coarser level, possibly a fake one) emit its two children among the finer vertices:
// real children in ascending index, missing ones replaced by fresh fake ids handed out
// consecutively from n_fine.
extern "C" int chebgcn_compute_perm_level(const int32_t* parent, int64_t n_fine, const int64_t* order,
                                          int64_t n_order, int64_t* out) {
    using chebgcn::fail;
    if (!parent || !order || !out || n_fine <= 0 || n_order < 0) return fail(CHEBGCN_EINVAL, "compute_perm_level: bad arguments");
    int64_t nvals = 0;
    for (int64_t i = 0; i < n_fine; ++i) {
        if (parent[i] < 0) return fail(CHEBGCN_EINVAL, "compute_perm_level: negative parent");
        if (parent[i] + 1 > nvals) nvals = parent[i] + 1;
    }
    std::vector<int64_t> kid0(nvals, -1), kid1(nvals, -1);
    for (int64_t i = 0; i < n_fine; ++i) {
        const int32_t p = parent[i];
        if (kid0[p] < 0) kid0[p] = i;
        else if (kid1[p] < 0) kid1[p] = i;
        else return fail(CHEBGCN_EINVAL, "compute_perm_level: vertex %d has more than two children", (int)p);
    }
    int64_t next_fake = n_fine;
    for (int64_t j = 0; j < n_order; ++j) {
        const int64_t node = order[j];
        int64_t a = -1, b = -1;
        if (node >= 0 && node < nvals) { a = kid0[node]; b = kid1[node]; }
        if (a < 0) a = next_fake++;
        if (b < 0) b = next_fake++;
        out[2 * j] = a;
        out[2 * j + 1] = b;
    }
    return CHEBGCN_OK;
}
