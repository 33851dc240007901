// Graph handle: uploads the rescaled Laplacian L~ and its transpose as length-sorted
// sliced-ELL images (for the on-chip recurrence) plus plain CSR (fallback path).
// Replaces the constant tf.SparseTensor of lib_new/models_gcn.py:593-596.
#include <algorithm>
#include <numeric>
#include <vector>
#include <new>
#include <string.h>

#include "common.h"

namespace chebgcn {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

template <typename T>
static int upload(T** dst, const std::vector<T>& src) {
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    CG_HIP(hipMalloc((void**)dst, bytes));
    if (!src.empty()) CG_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return CHEBGCN_OK;
}

static void free_ell(Ell& e) {
    void* ptrs[] = {e.goff, e.col16, e.val, e.rowid, e.rowptr, e.col32, e.cval};
    for (void* p : ptrs) (void)hipFree(p);
    e = Ell();
}

// CSR (host) -> device Ell.  Entry order inside a row is preserved, so the on-chip
// kernel sums a row in the order the caller gave (ascending column after
// tf.sparse_reorder in the reference).
static int build_ell(int M, const std::vector<int32_t>& rowptr, const std::vector<int32_t>& col,
                     const std::vector<float>& val, Ell* out) {
    std::vector<int32_t> order(M);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        return (rowptr[a + 1] - rowptr[a]) > (rowptr[b + 1] - rowptr[b]);
    });
    const int ngroups = (M + 63) / 64;
    std::vector<int32_t> goff(ngroups + 1, 0), rowid((size_t)ngroups * 64, -1);
    int max_len = 0;
    for (int g = 0; g < ngroups; ++g) {
        int r0 = order[g * 64];                       // longest row of the group
        int len = rowptr[r0 + 1] - rowptr[r0];
        max_len = std::max(max_len, len);
        goff[g + 1] = goff[g] + len;
    }
    const int64_t nslots = goff[ngroups];
    std::vector<uint16_t> col16;
    std::vector<float> eval((size_t)nslots * 64, 0.0f);
    const bool small = M < 65535;
    if (small) col16.assign((size_t)nslots * 64, (uint16_t)M);
    for (int r = 0; r < M; ++r) {
        int row = order[r], g = r / 64, lane = r % 64;
        rowid[r] = row;
        size_t base = (size_t)goff[g] * 64 + lane;
        for (int e = rowptr[row], s = 0; e < rowptr[row + 1]; ++e, ++s) {
            if (small) col16[base + (size_t)s * 64] = (uint16_t)col[e];
            eval[base + (size_t)s * 64] = val[e];
        }
    }
    out->ngroups = ngroups;
    out->max_len = max_len;
    out->nslots = nslots;
    int rc;
    if ((rc = upload(&out->goff, goff))) return rc;
    if ((rc = upload(&out->col16, col16))) return rc;
    if ((rc = upload(&out->val, eval))) return rc;
    if ((rc = upload(&out->rowid, rowid))) return rc;
    if ((rc = upload(&out->rowptr, rowptr))) return rc;
    if ((rc = upload(&out->col32, col))) return rc;
    if ((rc = upload(&out->cval, val))) return rc;
    return CHEBGCN_OK;
}

}  // namespace chebgcn

using namespace chebgcn;

extern "C" int chebgcn_version(void) { return CHEBGCN_VERSION; }
extern "C" const char* chebgcn_last_error(void) { return err_buf(); }
extern "C" int chebgcn_plane_stride(int M) { return plane_stride(M); }

extern "C" int chebgcn_graph_create(int M, int64_t nnz, const int32_t* rowptr, const int32_t* colidx,
                                    const float* vals, chebgcn_graph** out) {
    CG_REQUIRE(out != nullptr, "graph_create: out is NULL");
    *out = nullptr;
    CG_REQUIRE(M > 0 && nnz >= 0 && rowptr && (nnz == 0 || (colidx && vals)), "graph_create: bad arguments");
    CG_REQUIRE(rowptr[0] == 0 && rowptr[M] == nnz, "graph_create: rowptr[0] != 0 or rowptr[M] != nnz");
    for (int r = 0; r < M; ++r) CG_REQUIRE(rowptr[r + 1] >= rowptr[r], "graph_create: rowptr not monotone at row %d", r);
    for (int64_t e = 0; e < nnz; ++e)
        CG_REQUIRE(colidx[e] >= 0 && colidx[e] < M, "graph_create: column %d out of range at entry %lld", colidx[e], (long long)e);

    std::vector<int32_t> rp(rowptr, rowptr + M + 1), ci(colidx, colidx + nnz);
    std::vector<float> va(vals, vals + nnz);
    // transpose (counting sort by column keeps ascending-row order inside a column)
    std::vector<int32_t> trp(M + 1, 0), tci(nnz);
    std::vector<float> tva(nnz);
    for (int64_t e = 0; e < nnz; ++e) trp[ci[e] + 1]++;
    for (int r = 0; r < M; ++r) trp[r + 1] += trp[r];
    {
        std::vector<int32_t> cursor(trp.begin(), trp.end() - 1);
        for (int r = 0; r < M; ++r)
            for (int e = rp[r]; e < rp[r + 1]; ++e) {
                int dst = cursor[ci[e]]++;
                tci[dst] = r;
                tva[dst] = va[e];
            }
    }
    chebgcn_graph* g = new (std::nothrow) chebgcn_graph();
    if (!g) return fail(CHEBGCN_ENOMEM, "graph_create: out of host memory");
    g->M = M;
    g->Mp = plane_stride(M);
    g->nnz = nnz;
    hipDeviceProp_t prop;
    if (hipGetDevice(&g->device) != hipSuccess || hipGetDeviceProperties(&prop, g->device) != hipSuccess) {
        delete g;
        return fail(CHEBGCN_EHIP, "graph_create: no HIP device");
    }
    g->num_cus = prop.multiProcessorCount;
    // LDS image: (Mp + 1) float2 entries; 160 KiB per workgroup on gfx950
    size_t lds_need = (size_t)(g->Mp + 4) * 8;
    g->lds_ok = (M < 65535 && lds_need <= (size_t)prop.maxSharedMemoryPerMultiProcessor && lds_need <= 160 * 1024) ? 1 : 0;
    int rc = build_ell(M, rp, ci, va, &g->fwd);
    if (rc == CHEBGCN_OK) rc = build_ell(M, trp, tci, tva, &g->adj);
    if (rc != CHEBGCN_OK) {
        chebgcn_graph_destroy(g);
        return rc;
    }
    *out = g;
    return CHEBGCN_OK;
}

extern "C" void chebgcn_graph_destroy(chebgcn_graph* g) {
    if (!g) return;
    free_ell(g->fwd);
    free_ell(g->adj);
    delete g;
}

extern "C" int chebgcn_graph_query(const chebgcn_graph* g, int what, int64_t* value) {
    CG_REQUIRE(g && value, "graph_query: NULL argument");
    switch (what) {
        case 0: *value = g->M; break;
        case 1: *value = g->nnz; break;
        case 2: *value = g->Mp; break;
        case 3: *value = g->lds_ok; break;
        case 4: *value = g->fwd.nslots; break;
        case 5: *value = g->fwd.max_len; break;
        default: return fail(CHEBGCN_EINVAL, "graph_query: unknown item %d", what);
    }
    return CHEBGCN_OK;
}
