// Graph handle: uploads the rescaled Laplacian L~ and its transpose as length-sorted,
// bank-coloured sliced-ELL images (for the on-chip recurrence) plus plain CSR (fallback).
// Replaces the constant tf.SparseTensor of lib_new/models_gcn.py:593-596.
#include <algorithm>
#include <new>
#include <numeric>
#include <string.h>
#include <vector>

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
    void* ptrs[] = {e.ginfo, e.colq, e.valq, e.rowslot, e.nodeslot, e.rowptr, e.col32, e.cval};
    for (void* p : ptrs) (void)hipFree(p);
    e = Ell();
}

constexpr int kColours = 32;              // ds_read_b64: 32 bank pairs per half-wave group
constexpr int kLdsBytes = 160 * 1024;     // LDS per workgroup on gfx950
constexpr int kMaxEntries = kLdsBytes / 8;

// Can the LDS image of an M-vertex graph fit at all (perfectly balanced colouring)?
static bool image_can_fit(int M) {
    const int P = (M + kColours - 1) / kColours;
    return M < 65535 && (kColours * P + 4) <= kMaxEntries;
}

// ---------------------------------------------------------------------------------------
// Bank-aware vertex colouring.
// A half-wave ds_read_b64 gather costs one LDS cycle per distinct address on its busiest
// bank pair; the bank pair of slot s is s % 32.  Every (group, entry, half-wave) defines a set
// of up to 32 vertices read together; we pick colour[v] in [0, 32) to minimise same-colour
// pairs inside the sets (a smooth proxy for the max), subject to a per-colour capacity so
// that the image (32 * max population entries) still fits the 160 KiB LDS.  Deterministic.
// ---------------------------------------------------------------------------------------
struct Colouring {
    std::vector<int> colour;     // per vertex
    double cycles = 0, cycles_naive = 0;
};

static double mean_max(const std::vector<std::vector<int>>& sets, const std::vector<char>& has_pad,
                       const std::vector<int>& colour) {
    if (sets.empty()) return 1.0;
    double total = 0;
    int cnt[kColours];
    for (size_t s = 0; s < sets.size(); ++s) {
        memset(cnt, 0, sizeof(cnt));
        if (has_pad[s]) cnt[0] = 1;                    // the zero slot lives on bank pair 0
        int mx = has_pad[s] ? 1 : 0;
        for (int v : sets[s]) mx = std::max(mx, ++cnt[colour[v]]);
        total += std::max(mx, 1);
    }
    return total / sets.size();
}

static Colouring colour_vertices(int M, const std::vector<std::vector<int>>& sets, const std::vector<char>& has_pad) {
    Colouring out;
    out.colour.resize(M);
    for (int v = 0; v < M; ++v) out.colour[v] = v % kColours;
    out.cycles_naive = mean_max(sets, has_pad, out.colour);
    const int even = (M + kColours - 1) / kColours;
    const int cap = std::min((kMaxEntries - 4) / kColours, std::max(even * 3 / 2, even + 1));
    std::vector<std::vector<int>> member(M);
    for (size_t s = 0; s < sets.size(); ++s)
        for (int v : sets[s]) member[v].push_back((int)s);
    std::vector<int> cnt(sets.size() * kColours, 0), pop(kColours, 0);
    for (size_t s = 0; s < sets.size(); ++s) {
        if (has_pad[s]) cnt[s * kColours] += 1;
        for (int v : sets[s]) cnt[s * kColours + out.colour[v]] += 1;
    }
    for (int v = 0; v < M; ++v) pop[out.colour[v]]++;
    std::vector<int> order(M);
    std::iota(order.begin(), order.end(), 0);
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    for (int sweep = 0; sweep < 8; ++sweep) {
        for (int i = M - 1; i > 0; --i) std::swap(order[i], order[next() % (uint64_t)(i + 1)]);
        int moved = 0;
        for (int v : order) {
            if (member[v].empty()) continue;
            const int cur = out.colour[v];
            int score[kColours] = {0};
            for (int s : member[v]) {
                const int* c = &cnt[(size_t)s * kColours];
                for (int k = 0; k < kColours; ++k) score[k] += c[k];
            }
            score[cur] -= (int)member[v].size();      // do not count v itself
            int best = cur;
            for (int k = 0; k < kColours; ++k) {
                if (k == cur || pop[k] >= cap) continue;
                if (score[k] < score[best] || (best != cur && score[k] == score[best] && pop[k] < pop[best])) best = k;
            }
            if (best != cur) {
                for (int s : member[v]) { cnt[(size_t)s * kColours + cur]--; cnt[(size_t)s * kColours + best]++; }
                pop[cur]--; pop[best]++;
                out.colour[v] = best;
                ++moved;
            }
        }
        if (moved == 0) break;
    }
    // vertices that are never gathered go to the emptiest colours
    for (int v = 0; v < M; ++v)
        if (member[v].empty()) {
            pop[out.colour[v]]--;
            const int best = (int)(std::min_element(pop.begin(), pop.end()) - pop.begin());
            pop[best]++;
            out.colour[v] = best;
        }
    out.cycles = mean_max(sets, has_pad, out.colour);
    return out;
}

// CSR (host) -> device Ell.  Entry order inside a row is preserved, so the on-chip kernel
// sums a row in the order the caller gave (ascending column after tf.sparse_reorder in the
// reference).
static int build_ell(int M, int Mp, bool on_chip, const std::vector<int32_t>& rowptr, const std::vector<int32_t>& col,
                     const std::vector<float>& val, Ell* out) {
    int rc;
    if ((rc = upload(&out->rowptr, rowptr))) return rc;
    if ((rc = upload(&out->col32, col))) return rc;
    if ((rc = upload(&out->cval, val))) return rc;
    if (!on_chip) return CHEBGCN_OK;

    auto rlen = [&](int r) { return rowptr[r + 1] - rowptr[r]; };
    std::vector<int32_t> order(M);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return rlen(a) > rlen(b); });
    const int ngroups = (M + 63) / 64;
    std::vector<int2> ginfo(ngroups);
    int max_len = 0;
    int64_t nquads = 0;
    for (int g = 0; g < ngroups; ++g) {
        const int len = rlen(order[g * 64]);              // longest row of the group
        max_len = std::max(max_len, len);
        ginfo[g] = make_int2((int)nquads, (len + 1) & ~1);
        nquads += (len + 3) / 4;
    }
    // gather sets for the colouring
    std::vector<std::vector<int>> sets;
    std::vector<char> has_pad;
    for (int g = 0; g < ngroups; ++g) {
        const int len = rlen(order[g * 64]);
        for (int s = 0; s < ((len + 1) & ~1); ++s)
            for (int h = 0; h < 2; ++h) {
                std::vector<int> nodes;
                bool pad = false;
                for (int lane = 32 * h; lane < 32 * h + 32; ++lane) {
                    const int r = g * 64 + lane;
                    if (r < M && s < rlen(order[r])) nodes.push_back(col[rowptr[order[r]] + s]);
                    else pad = true;
                }
                std::sort(nodes.begin(), nodes.end());
                nodes.erase(std::unique(nodes.begin(), nodes.end()), nodes.end());
                sets.push_back(std::move(nodes));
                has_pad.push_back(pad);
            }
    }
    Colouring cl = colour_vertices(M, sets, has_pad);
    // slots: colour + 32 * index (index = rank of the vertex inside its colour)
    std::vector<int> pop(kColours, 0);
    std::vector<uint16_t> nodeslot((size_t)Mp + 4, 0xFFFF);
    for (int v = 0; v < M; ++v) nodeslot[v] = (uint16_t)(cl.colour[v] + kColours * pop[cl.colour[v]]++);
    const int P = *std::max_element(pop.begin(), pop.end());
    const int zero_slot = kColours * P;                    // bank pair 0, one past the last index
    const int lds_entries = (zero_slot + 1 + 3) & ~3;
    if (lds_entries > kMaxEntries) return fail(CHEBGCN_EUNSUPPORTED, "graph_create: LDS image needs %d entries", lds_entries);

    const uint32_t zz = (uint32_t)zero_slot | ((uint32_t)zero_slot << 16);
    std::vector<uint2> colq((size_t)nquads * 64, make_uint2(zz, zz));
    std::vector<float4> valq((size_t)nquads * 64, make_float4(0.f, 0.f, 0.f, 0.f));
    std::vector<uint16_t> rowslot((size_t)ngroups * 64, 0xFFFF);
    for (int r = 0; r < M; ++r) {
        const int row = order[r], g = r / 64, lane = r % 64;
        rowslot[r] = nodeslot[row];
        for (int e = rowptr[row], s = 0; e < rowptr[row + 1]; ++e, ++s) {
            const size_t at = ((size_t)ginfo[g].x + s / 4) * 64 + lane;
            const uint32_t slot = nodeslot[col[e]];
            uint32_t* w = (s & 2) ? &colq[at].y : &colq[at].x;
            *w = (s & 1) ? ((*w & 0x0000FFFFu) | (slot << 16)) : ((*w & 0xFFFF0000u) | slot);
            (&valq[at].x)[s & 3] = val[e];
        }
    }
    out->ngroups = ngroups;
    out->max_len = max_len;
    out->nquads = nquads;
    out->lds_entries = lds_entries;
    out->zero_slot = zero_slot;
    out->est_cycles = cl.cycles;
    out->est_cycles_naive = cl.cycles_naive;
    if ((rc = upload(&out->ginfo, ginfo))) return rc;
    if ((rc = upload(&out->colq, colq))) return rc;
    if ((rc = upload(&out->valq, valq))) return rc;
    if ((rc = upload(&out->rowslot, rowslot))) return rc;
    if ((rc = upload(&out->nodeslot, nodeslot))) return rc;
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
    g->lds_ok = (image_can_fit(M) && (size_t)kLdsBytes <= (size_t)prop.maxSharedMemoryPerMultiProcessor) ? 1 : 0;
    int rc = build_ell(M, g->Mp, g->lds_ok != 0, rp, ci, va, &g->fwd);
    if (rc == CHEBGCN_OK) rc = build_ell(M, g->Mp, g->lds_ok != 0, trp, tci, tva, &g->adj);
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
        case 4: *value = g->fwd.nquads * 4; break;
        case 5: *value = g->fwd.max_len; break;
        case 6: *value = (int64_t)(g->fwd.est_cycles * 1000); break;         // milli-cycles per half-wave gather
        case 7: *value = (int64_t)(g->fwd.est_cycles_naive * 1000); break;
        case 8: *value = (int64_t)g->fwd.lds_entries * 8; break;             // LDS bytes of the image
        default: return fail(CHEBGCN_EINVAL, "graph_query: unknown item %d", what);
    }
    return CHEBGCN_OK;
}
