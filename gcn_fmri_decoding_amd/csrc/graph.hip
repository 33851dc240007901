// Graph handle: uploads the rescaled Laplacian L~ and its transpose as length-sorted
// sliced-ELL images (for the on-chip recurrence) plus plain CSR (fallback path).
// Replaces the constant tf.SparseTensor of lib_new/models_gcn.py:593-596.
#include <algorithm>
#include <cstring>
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

namespace {
struct DispatchRecord {
    const char* name[6];
    int n;
    char joined[512];
};
DispatchRecord& dispatch_record() {
    static thread_local DispatchRecord r = {{nullptr}, 0, {0}};
    return r;
}
}  // namespace

void note_dispatch(const char* name) {
    DispatchRecord& r = dispatch_record();
    r.name[0] = name;
    r.n = 1;
}

void note_dispatch_more(const char* name) {
    DispatchRecord& r = dispatch_record();
    if (r.n < 6) r.name[r.n++] = name;
}

template <typename T>
static int upload(T** dst, const std::vector<T>& src) {
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    CG_HIP(hipMalloc((void**)dst, bytes));
    if (!src.empty()) CG_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return CHEBGCN_OK;
}

static void free_ell(Ell& e) {
    void* ptrs[] = {e.ginfo, e.colq, e.colo, e.valq, e.valp, e.rowslot, e.nodeslot, e.rowptr, e.col32, e.cval, e.uval, e.uids, e.blkmap, e.fs_rec};
    for (void* p : ptrs) (void)hipFree(p);
    e = Ell();
}

constexpr int kLdsBytes = 160 * 1024;     // LDS per workgroup on gfx950

// Planes per workgroup for an image of n vertices (+ zero and trash slot, rounded to 4 entries):
// 4 if 16 B per vertex fit the LDS, else 2 if 8 B fit, else 0 (no on-chip path).
static int planes_for(int n) {
    const size_t entries = ((size_t)n + 2 + 3) & ~(size_t)3;
    if (n >= 65535) return 0;
    if (entries * 16 <= (size_t)kLdsBytes) return 4;
    if (entries * 8 <= (size_t)kLdsBytes) return 2;
    return 0;
}

// Entry order inside a row is free (a sum).  The 64 rows of a group gather entry position e with one LDS instruction, which
// the hardware serves in fixed lane sets (two of 32 lanes for the 8-byte reads of P = 2, four of 16 for the 16-byte reads of
// P = 4, MI355X_MICROARCH.md "LDS"); inside a set, distinct entries on one bank serialise.  The positions of each row's
// entries are therefore chosen greedily, position by position, so that the rows of a lane set hit different banks; rows
// shorter than the group may leave holes (zero-slot padding).  rows[lane] = the vertex whose row lane gathers (-1: none),
// q0 / L = the group's first quad and (even) length, slot_of(v) = the LDS slot of vertex v.
template <class SlotOf>
static void place_group(int planes, const int (&rows)[64], int q0, int L, const std::vector<int32_t>& rowptr,
                        const std::vector<int32_t>& col, const std::vector<float>& val, SlotOf slot_of, std::vector<uint2>& colq,
                        std::vector<float4>& valq, int64_t* cost_before, int64_t* cost_after, int64_t* cost_ideal) {
    const int nbanks = planes == 4 ? 16 : 32;            // distinct entry-sized bank groups
    auto lane_set = [&](int lane) {
        if (planes != 4) return lane >> 5;
        static const int blk[16] = {0, 1, 1, 0, 1, 0, 0, 1, 2, 3, 3, 2, 3, 2, 2, 3};
        return blk[lane >> 2];
    };
    // positions the kernel visits (recurrence*.hip); a third quad of at most two entries is
    // gathered as a pair (its ids travel in the value record, see valp below)
    const int npos = L <= 8 ? 8 : L <= 10 ? 10 : 4 * ((L + 3) / 4);
    for (int set = 0; set < (planes == 4 ? 4 : 2); ++set) {
        std::vector<int> lanes;
        for (int lane = 0; lane < 64; ++lane)
            if (lane_set(lane) == set && rows[lane] >= 0) lanes.push_back(lane);
        const int nr = (int)lanes.size();
        std::vector<std::vector<int>> rem(nr);          // remaining CSR entry ids per row
        for (int i = 0; i < nr; ++i) {
            const int row = rows[lanes[i]];
            for (int e = rowptr[row]; e < rowptr[row + 1]; ++e) rem[i].push_back(e);
        }
        // cost of the caller's order, for the statistics
        for (int pos = 0; pos < npos; ++pos) {
            std::vector<std::vector<uint32_t>> seen(nbanks);
            int worst = 1;
            for (int i = 0; i < nr; ++i) {
                if (pos >= (int)rem[i].size()) continue;
                const uint32_t sl = slot_of(col[rem[i][pos]]);
                auto& v = seen[sl % nbanks];
                if (std::find(v.begin(), v.end(), sl) == v.end()) v.push_back(sl);
                worst = std::max(worst, (int)v.size());
            }
            *cost_before += worst;
        }
        std::vector<int> idx(nr);
        for (int pos = 0; pos < npos; ++pos) {
            for (int i = 0; i < nr; ++i) idx[i] = i;
            // rows that may not leave a hole any more go first, then the longer ones
            std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
                const int sa = (npos - pos) - (int)rem[a].size(), sb = (npos - pos) - (int)rem[b].size();
                if ((sa <= 0) != (sb <= 0)) return sa <= 0;
                return rem[a].size() > rem[b].size();
            });
            std::vector<std::vector<uint32_t>> seen(nbanks);
            int worst = 1;
            for (int i : idx) {
                if (rem[i].empty()) continue;
                const int slack = (npos - pos) - (int)rem[i].size();
                int best = -1, best_load = 1 << 30;
                for (int k = 0; k < (int)rem[i].size(); ++k) {
                    const uint32_t sl = slot_of(col[rem[i][k]]);
                    const auto& v = seen[sl % nbanks];
                    const int load = std::find(v.begin(), v.end(), sl) != v.end() ? 0 : (int)v.size();
                    if (load < best_load) { best_load = load; best = k; }
                }
                if (best_load > 0 && slack > 0) continue;       // leave a hole, try a later position
                const int e = rem[i][best];
                rem[i].erase(rem[i].begin() + best);
                const uint32_t slot = slot_of(col[e]);
                auto& v = seen[slot % nbanks];
                if (std::find(v.begin(), v.end(), slot) == v.end()) v.push_back(slot);
                worst = std::max(worst, (int)v.size());
                const size_t at = ((size_t)q0 + pos / 4) * 64 + lanes[i];
                uint32_t* w = (pos & 2) ? &colq[at].y : &colq[at].x;
                *w = (pos & 1) ? ((*w & 0x0000FFFFu) | (slot << 16)) : ((*w & 0xFFFF0000u) | slot);
                (&valq[at].x)[pos & 3] = val[e];
            }
            *cost_after += worst;
            *cost_ideal += 1;
        }
    }
}

// per-vertex operator records of the fused atlas-size layer (common.h Ell::fs_rec); rows of more than 20 entries: none
static int build_fs_records(int M, int Mp, const std::vector<int32_t>& rowptr, const std::vector<int32_t>& col,
                            const std::vector<float>& val, Ell* out) {
    std::vector<uint32_t> rec((size_t)Mp * 32, 0u);
    for (int v = 0; v < M; ++v) {
        const int len = rowptr[v + 1] - rowptr[v];
        if (len > 20) return CHEBGCN_OK;
        uint32_t* r = &rec[(size_t)v * 32];
        for (int e = 0; e < len; ++e) {
            const uint32_t c = (uint32_t)col[rowptr[v] + e];
            r[e >> 1] |= (e & 1) ? c << 16 : c;
            const float w = val[rowptr[v] + e];
            std::memcpy(&r[12 + e], &w, 4);
        }
        r[10] = (uint32_t)len;
    }
    return upload(&out->fs_rec, rec);
}

// CSR (host) -> device Ell.  Entry order inside a row is NOT the caller's: positions are chosen
// below so that the 64 rows of a group hit different LDS banks (a row sum has <= ~17 terms; the
// order is fixed per graph, so results are deterministic).  `active[v]` marks vertices whose row
// or column is non-empty.
static int build_ell(int M, int Mp, int planes, const std::vector<char>& active, const std::vector<int32_t>& rowptr,
                     const std::vector<int32_t>& col, const std::vector<float>& val, Ell* out) {
    int rc;
    if ((rc = upload(&out->rowptr, rowptr))) return rc;
    if ((rc = upload(&out->col32, col))) return rc;
    if ((rc = upload(&out->cval, val))) return rc;
    out->planes = planes;
    if (planes == 0) return CHEBGCN_OK;

    auto rlen = [&](int r) { return rowptr[r + 1] - rowptr[r]; };
    // LDS slots: P = 4 keeps only the active vertices, P = 2 all of them.  Slots are handed out
    // component-major -- all vertices 4q, then all 4q+1, ... -- because the linear phases of the
    // kernel give lane l the four vertices 4(q0+l)..+3: for a fixed component the lanes of a wave
    // then touch consecutive slots (no bank conflicts), where slot = vertex id would be 4-way
    // (8-byte entries) or 8-way (16-byte entries) conflicted.
    std::vector<uint16_t> nodeslot((size_t)Mp + 4, 0xFFFF);
    std::vector<int32_t> order;
    int nslot = 0;
    for (int i = 0; i < 4; ++i)
        for (int v = i; v < M; v += 4)
            if (planes == 2 || active[v]) {
                nodeslot[v] = (uint16_t)nslot++;
                order.push_back(v);
            }
    const int zero_slot = nslot;
    const int lds_entries = (nslot + 2 + 3) & ~3;          // + zero slot + trash slot (zero_slot + 1)
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return rlen(a) > rlen(b); });
    // Rows of equal length may take any lane.  The rotate phase of the kernels reads and writes
    // the OWN row's entry of every lane with one LDS instruction; giving lane l a row whose slot
    // is congruent to l modulo the number of entry-sized bank groups makes those accesses
    // conflict-free (every lane set of MI355X_MICROARCH.md "LDS" then covers each bank once).
    {
        const int nb = planes == 4 ? 16 : 32;
        std::vector<int32_t> placed;
        placed.reserve(order.size());
        size_t a = 0;
        while (a < order.size()) {
            size_t b = a;
            while (b < order.size() && rlen(order[b]) == rlen(order[a])) ++b;       // one length class
            std::vector<std::vector<int32_t>> bucket(nb);
            for (size_t i = b; i-- > a;) bucket[nodeslot[order[i]] % nb].push_back(order[i]);   // pop_back = ascending
            for (size_t i = a; i < b; ++i) {
                int want = (int)(placed.size() % 64) % nb;
                if (bucket[want].empty()) {                      // residue used up: take from the fullest
                    want = 0;
                    for (int k = 1; k < nb; ++k)
                        if (bucket[k].size() > bucket[want].size()) want = k;
                }
                placed.push_back(bucket[want].back());
                bucket[want].pop_back();
            }
            a = b;
        }
        order.swap(placed);
    }
    const int nranked = (int)order.size();
    const int ngroups = (nranked + 63) / 64;
    std::vector<int2> ginfo(ngroups);
    int max_len = 0;
    int64_t nquads = 0, nslots = 0;
    for (int g = 0; g < ngroups; ++g) {
        const int len = rlen(order[g * 64]);              // longest row of the group
        max_len = std::max(max_len, len);
        ginfo[g] = make_int2((int)nquads, (len + 1) & ~1);   // even: the kernel gathers in pairs
        // at least kQuadMin quads (recurrence4.hip requests that many unconditionally), an even
        // number of them: two quads share one 16-byte record of slot ids (colo below)
        nquads += (std::max((len + 3) / 4, kQuadMin) + 1) & ~1;
        nslots += (len + 1) & ~1;
    }
    const uint32_t zz = (uint32_t)zero_slot | ((uint32_t)zero_slot << 16);
    std::vector<uint2> colq((size_t)(nquads + kQuadPad) * 64, make_uint2(zz, zz));
    std::vector<float4> valq((size_t)(nquads + kQuadPad) * 64, make_float4(0.f, 0.f, 0.f, 0.f));
    std::vector<uint16_t> rowslot((size_t)ngroups * 64, 0xFFFF);
    int64_t cost_before = 0, cost_after = 0, cost_ideal = 0;
    for (int g = 0; g < ngroups; ++g) {
        int rows[64];
        for (int lane = 0; lane < 64; ++lane) {
            rows[lane] = g * 64 + lane < nranked ? order[g * 64 + lane] : -1;
            if (rows[lane] >= 0) rowslot[g * 64 + lane] = nodeslot[rows[lane]];
        }
        place_group(planes, rows, ginfo[g].x, ginfo[g].y, rowptr, col, val, [&](int v) { return (uint32_t)nodeslot[v]; }, colq, valq,
                    &cost_before, &cost_after, &cost_ideal);
    }
    out->cost_before = cost_before;
    out->cost_after = cost_after;
    out->cost_ideal = cost_ideal;
    out->ngroups = ngroups;
    out->max_len = max_len;
    out->nranked = nranked;
    out->lds_entries = lds_entries;
    out->zero_slot = zero_slot;
    out->nslots = nslots;
    out->nquads = nquads;
    if ((rc = upload(&out->ginfo, ginfo))) return rc;
    // the slot ids once more, eight per lane and record: one 16-byte load serves two quads (the
    // vector-memory path costs per wave instruction, not per byte)
    std::vector<uint4> colo(colq.size() / 2);
    for (size_t o = 0; o < colo.size() / 64; ++o)
        for (int lane = 0; lane < 64; ++lane) {
            const uint2 a = colq[(2 * o) * 64 + lane], b = colq[(2 * o + 1) * 64 + lane];
            colo[o * 64 + lane] = make_uint4(a.x, a.y, b.x, b.y);
        }
    if ((rc = upload(&out->colo, colo))) return rc;
    if ((rc = upload(&out->colq, colq))) return rc;
    if ((rc = upload(&out->valq, valq))) return rc;
    // values as the generic kernel reads them: where the third quad of a group holds at most two
    // entries (length 9..10), its two ids ride in the unused half of the 16-byte value record, so
    // that such a group needs no second id record (one vector-memory instruction less per step)
    std::vector<float4> valp(valq);
    for (int g = 0; g < ngroups; ++g) {
        if (ginfo[g].y <= 8 || ginfo[g].y > 10) continue;
        const size_t q2 = (size_t)ginfo[g].x + 2;
        for (int lane = 0; lane < 64; ++lane) {
            float4& v = valp[q2 * 64 + lane];
            const uint32_t ids = colq[q2 * 64 + lane].x;
            memcpy(&v.z, &ids, 4);
            memcpy(&v.w, &zz, 4);
        }
    }
    if ((rc = upload(&out->valp, valp))) return rc;
    if (planes == 4) {
        std::vector<int> per_thread(512, 0);
        for (int v = 0; v < M; ++v)
            if (nodeslot[v] == 0xFFFF) per_thread[(v / 4) % 512]++;
        out->iso_max512 = *std::max_element(per_thread.begin(), per_thread.end());
        // fixed-stride image of entries 0..11 (common.h): every group has at least four stored quads
        // a kernel shape requests the records of every slot of every wave (up to 20 slots x 8 waves), also those beyond
        // the last group: padded with empty groups (values 0, ids = the zero slot)
        const int ngpad = std::max(((ngroups + 7) / 8) * 8, 160);
        std::vector<float4> uval((size_t)ngpad * 4 * 64, make_float4(0.f, 0.f, 0.f, 0.f));
        std::vector<uint4> uids((size_t)ngpad * 64, make_uint4(zz, zz, zz, zz));
        for (size_t i = (size_t)ngroups * 4 * 64 + 2 * 64; i < uval.size(); i += 4 * 64)      // record 2 of a padding group: its two id words
            for (int lane = 0; lane < 64; ++lane) {
                memcpy(&uval[i + lane].z, &zz, 4);
                memcpy(&uval[i + lane].w, &zz, 4);
            }
        for (int g = 0; g < ngroups; ++g) {
            const size_t q0 = (size_t)ginfo[g].x;
            for (int lane = 0; lane < 64; ++lane) {
                const uint2 c0 = colq[q0 * 64 + lane], c1 = colq[(q0 + 1) * 64 + lane], c2 = colq[(q0 + 2) * 64 + lane];
                const float4 v2 = valq[(q0 + 2) * 64 + lane];
                uids[(size_t)g * 64 + lane] = make_uint4(c0.x, c0.y, c1.x, c1.y);
                uval[((size_t)g * 4 + 0) * 64 + lane] = valq[q0 * 64 + lane];
                uval[((size_t)g * 4 + 1) * 64 + lane] = valq[(q0 + 1) * 64 + lane];
                float4 r2 = make_float4(v2.x, v2.y, 0.f, 0.f);
                memcpy(&r2.z, &c2.x, 4);
                memcpy(&r2.w, &c2.y, 4);
                uval[((size_t)g * 4 + 2) * 64 + lane] = r2;
                uval[((size_t)g * 4 + 3) * 64 + lane] = make_float4(v2.z, v2.w, 0.f, 0.f);
            }
        }
        if ((rc = upload(&out->uval, uval))) return rc;
        if ((rc = upload(&out->uids, uids))) return rc;
    }
    if ((rc = upload(&out->rowslot, rowslot))) return rc;
    if ((rc = upload(&out->nodeslot, nodeslot))) return rc;
    return CHEBGCN_OK;
}

// Ordered image (common.h, recurrence_ord.hip): the operator of a graph whose rows are sorted by descending length, laid
// out for threads that own whole vertex quads.  Same record formats as above (fixed-stride image of the first 12 entries of
// every row group + the variable-stride quads for longer rows), same bank-aware placement of the entries of a row.
static int build_ell_ordered(int M, int NT, int NQ, int NG, int SQ, int planes, const std::vector<int32_t>& rowptr,
                             const std::vector<int32_t>& col, const std::vector<float>& val, Ell* out) {
    int rc;
    const int NW = NT / 64, NJ = 4 * NG, ngroups = NJ * NW;
    out->planes = planes;
    out->ord_NT = NT; out->ord_NQ = NQ; out->ord_NG = NG; out->ord_SQ = SQ;
    auto slot_of_vertex = [&](int v) { return (uint32_t)((v & 3) * SQ + (v >> 2)); };
    const uint32_t zero_slot = (uint32_t)(4 * SQ);
    auto rlen = [&](int r) { return r < 0 ? 0 : rowptr[r + 1] - rowptr[r]; };
    // ---- blocks of 64 quads -> (wave, level).  Cost of a block = the entry positions its four slices visit (8 / 10 / 12 /
    // whole quads beyond: what recurrence_ord.hip gathers for the longest row of a slice).  Longest-processing-time first
    // onto the least loaded wave that still has a free level, then pairwise swaps while the heaviest wave gets lighter.
    const int nblk_rows = (SQ + 63) / 64;              // blocks that hold rows: levels 0..NG-1 of the waves
    std::vector<int> bcost(NG * NW, 0);
    for (int b = 0; b < nblk_rows; ++b)
        for (int i = 0; i < 4; ++i) {
            int len = 0;
            for (int lane = 0; lane < 64; ++lane) {
                const int q = b * 64 + lane, v = 4 * q + i;
                if (q < SQ && v < M) len = std::max(len, rlen(v));
            }
            len = (len + 1) & ~1;
            bcost[b] += len <= 8 ? 8 : len <= 10 ? 10 : len <= 12 ? 12 : 4 * ((len + 3) / 4);
        }
    // (a block that is not full of rows -- the last one with rows, and empty ones behind it: fewer than NW + 1 -- must be the
    // LAST level of its wave: the kernel treats the levels below NG - 1 as "every lane has a slot")
    auto full = [&](int b) { return 64 * (b + 1) <= SQ; };
    std::vector<std::vector<int>> mine(NW);
    std::vector<int> load(NW, 0), partial(NW, 0);
    {
        std::vector<int> idx(NG * NW);
        std::iota(idx.begin(), idx.end(), 0);
        // the blocks that are not full first (at most one per wave), then the full ones by descending cost
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
            if (full(a) != full(b)) return !full(a);
            return bcost[a] > bcost[b];
        });
        for (int b : idx) {
            int best = -1;
            for (int w = 0; w < NW; ++w)
                if ((int)mine[w].size() < NG && (full(b) || !partial[w]) && (best < 0 || load[w] < load[best])) best = w;
            if (best < 0) return fail(CHEBGCN_EUNSUPPORTED, "graph_create: no block assignment for the ordered image");
            mine[best].push_back(b);
            load[best] += bcost[b];
            partial[best] += !full(b);
        }
        for (bool moved = true; moved;) {
            moved = false;
            const int hw = (int)(std::max_element(load.begin(), load.end()) - load.begin());
            for (int w = 0; w < NW && !moved; ++w)
                for (size_t x = 0; x < mine[hw].size() && !moved; ++x)
                    for (size_t y = 0; y < mine[w].size() && !moved; ++y) {
                        const int d = bcost[mine[hw][x]] - bcost[mine[w][y]];
                        if (w != hw && d > 0 && load[w] + d < load[hw] && full(mine[hw][x]) && full(mine[w][y])) {
                            std::swap(mine[hw][x], mine[w][y]);
                            load[hw] -= d;
                            load[w] += d;
                            moved = true;
                        }
                    }
        }
    }
    std::vector<int32_t> blkmap((size_t)NW * NQ);
    for (int w = 0; w < NW; ++w) {
        std::sort(mine[w].begin(), mine[w].end());     // ascending: a wave's slices stay sorted by descending length
        for (int u = 0; u < NQ; ++u) blkmap[(size_t)w * NQ + u] = u < NG ? mine[w][u] : u * NW + w;
    }
    auto row_of = [&](int g, int lane) {               // vertex of a rank, -1 = none
        const int j = g / NW, w = g % NW, u = j >> 2, i = j & 3;
        const int q = blkmap[(size_t)w * NQ + u] * 64 + lane, v = 4 * q + i;
        return (q < SQ && v < M) ? v : -1;
    };
    std::vector<int2> ginfo(ngroups);
    int max_len = 0;
    int64_t nquads = 0, nslots = 0;
    for (int g = 0; g < ngroups; ++g) {
        int len = 0;
        for (int lane = 0; lane < 64; ++lane) len = std::max(len, rlen(row_of(g, lane)));
        max_len = std::max(max_len, len);
        ginfo[g] = make_int2((int)nquads, (len + 1) & ~1);
        nquads += (std::max((len + 3) / 4, kQuadMin) + 1) & ~1;
        nslots += (len + 1) & ~1;
    }
    const uint32_t zz = zero_slot | (zero_slot << 16);
    std::vector<uint2> colq((size_t)(nquads + kQuadPad) * 64, make_uint2(zz, zz));
    std::vector<float4> valq((size_t)(nquads + kQuadPad) * 64, make_float4(0.f, 0.f, 0.f, 0.f));
    for (int g = 0; g < ngroups; ++g) {
        int rows[64];
        for (int lane = 0; lane < 64; ++lane) rows[lane] = row_of(g, lane);
        place_group(planes, rows, ginfo[g].x, ginfo[g].y, rowptr, col, val, slot_of_vertex, colq, valq, &out->cost_before,
                    &out->cost_after, &out->cost_ideal);
    }
    out->ngroups = ngroups;
    out->max_len = max_len;
    out->nranked = ngroups * 64;
    out->lds_entries = (4 * SQ + 2 + 3) & ~3;            // (of `planes` floats each)
    out->zero_slot = (int)zero_slot;
    out->nslots = nslots;
    out->nquads = nquads;
    if ((rc = upload(&out->ginfo, ginfo))) return rc;
    std::vector<uint4> colo(colq.size() / 2);
    for (size_t o = 0; o < colo.size() / 64; ++o)
        for (int lane = 0; lane < 64; ++lane) {
            const uint2 a = colq[(2 * o) * 64 + lane], b = colq[(2 * o + 1) * 64 + lane];
            colo[o * 64 + lane] = make_uint4(a.x, a.y, b.x, b.y);
        }
    if ((rc = upload(&out->colo, colo))) return rc;
    if ((rc = upload(&out->valq, valq))) return rc;
    // fixed-stride image of entries 0..11; two spare groups behind the last one (the ring requests two groups ahead)
    std::vector<float4> uval((size_t)(ngroups + 2 * NW) * 4 * 64, make_float4(0.f, 0.f, 0.f, 0.f));
    std::vector<uint4> uids((size_t)(ngroups + 2 * NW) * 64, make_uint4(zz, zz, zz, zz));
    for (size_t i = 2 * 64; i < uval.size(); i += 4 * 64)          // record 2 of every group: its two id words default to the zero slot
        for (int lane = 0; lane < 64; ++lane) {
            memcpy(&uval[i + lane].z, &zz, 4);
            memcpy(&uval[i + lane].w, &zz, 4);
        }
    for (int g = 0; g < ngroups; ++g) {
        const size_t q0 = (size_t)ginfo[g].x;
        for (int lane = 0; lane < 64; ++lane) {
            const uint2 c0 = colq[q0 * 64 + lane], c1 = colq[(q0 + 1) * 64 + lane], c2 = colq[(q0 + 2) * 64 + lane];
            const float4 v2 = valq[(q0 + 2) * 64 + lane];
            uids[(size_t)g * 64 + lane] = make_uint4(c0.x, c0.y, c1.x, c1.y);
            uval[((size_t)g * 4 + 0) * 64 + lane] = valq[q0 * 64 + lane];
            uval[((size_t)g * 4 + 1) * 64 + lane] = valq[(q0 + 1) * 64 + lane];
            float4 r2 = make_float4(v2.x, v2.y, 0.f, 0.f);
            memcpy(&r2.z, &c2.x, 4);
            memcpy(&r2.w, &c2.y, 4);
            uval[((size_t)g * 4 + 2) * 64 + lane] = r2;
            uval[((size_t)g * 4 + 3) * 64 + lane] = make_float4(v2.z, v2.w, 0.f, 0.f);
        }
    }
    if ((rc = upload(&out->uval, uval))) return rc;
    if ((rc = upload(&out->uids, uids))) return rc;
    if ((rc = upload(&out->blkmap, blkmap))) return rc;
    return CHEBGCN_OK;
}

// rows sorted by descending length, every empty row behind every non-empty one: the number of non-empty rows, else -1
static int sorted_rows(int M, const std::vector<int32_t>& rp) {
    int n = 0;
    for (int r = 0; r < M; ++r) {
        const int len = rp[r + 1] - rp[r];
        if (r > 0 && len > rp[r] - rp[r - 1]) return -1;
        n += len > 0;
    }
    return n;
}

}  // namespace chebgcn

using namespace chebgcn;

extern "C" int chebgcn_version(void) { return CHEBGCN_VERSION; }
extern "C" const char* chebgcn_last_error(void) { return err_buf(); }
extern "C" const char* chebgcn_last_dispatch(void) {
    DispatchRecord& r = dispatch_record();
    size_t at = 0;
    r.joined[0] = 0;
    for (int i = 0; i < r.n; ++i) {
        const int w = snprintf(r.joined + at, sizeof(r.joined) - at, "%s%s", i ? " + " : "", r.name[i]);
        if (w < 0 || (size_t)w >= sizeof(r.joined) - at) break;
        at += (size_t)w;
    }
    return r.joined;
}
extern "C" int chebgcn_plane_stride(int M) { return plane_stride(M); }

extern "C" int chebgcn_graph_create(int M, int64_t nnz, const int32_t* rowptr, const int32_t* colidx,
                                    const float* vals, chebgcn_graph** out) {
    return chebgcn_graph_create_planes(M, nnz, rowptr, colidx, vals, 0, out);
}

extern "C" int chebgcn_graph_create_planes(int M, int64_t nnz, const int32_t* rowptr, const int32_t* colidx,
                                           const float* vals, int want_planes, chebgcn_graph** out) {
    CG_REQUIRE(out != nullptr, "graph_create: out is NULL");
    CG_REQUIRE(want_planes == 0 || want_planes == 2 || want_planes == 4, "graph_create: planes must be 0, 2 or 4");
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
    std::vector<char> active(M, 0);
    int nactive = 0;
    for (int v = 0; v < M; ++v) {
        active[v] = (rp[v + 1] > rp[v]) || (trp[v + 1] > trp[v]);
        nactive += active[v];
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
    int planes = 0;
    if ((size_t)kLdsBytes <= (size_t)prop.maxSharedMemoryPerMultiProcessor) {
        // 2 planes per workgroup keep every vertex on chip; 4 planes (only the active vertices on
        // chip, 16 bytes each) halve the operator stream per plane.  Measured on MI355X (same box, A/B):
        // up to 2048 rows (the sizes of real brain atlases, 246..1000 nodes) the generic kernel with 4
        // planes is 1.1-2x faster; beyond, the kernel of recurrence4.hip is 1.07x (adjoint) to 1.19x
        // (forward) faster at the north-star shape of the benchmark graph (10000 active vertices).
        // Automatic choice: 4 planes wherever a four-plane kernel shape exists.
        planes = planes_for(M) >= 2 ? 2 : 0;
        const int rows = ((nactive + 63) / 64) * 64, entries = (nactive + 2 + 3) & ~3;
        const bool small4 = planes_for(nactive) == 4 && generic4_fits(rows, g->Mp / 4);
        const bool big4 = planes_for(nactive) == 4 && onchip4_fits(entries, rows, g->Mp / 4);
        if (want_planes == 4 && !(small4 || big4)) {
            delete g;
            return fail(CHEBGCN_EUNSUPPORTED, "graph_create: no four-plane kernel for %d active of %d vertices", nactive, M);
        }
        if (want_planes == 4 || (want_planes == 0 && (small4 || big4))) planes = 4;
    }
    g->lds_ok = planes != 0;
    int rc = build_ell(M, g->Mp, planes, active, rp, ci, va, &g->fwd);
    if (rc == CHEBGCN_OK) rc = build_ell(M, g->Mp, planes, active, trp, tci, tva, &g->adj);
    if (rc == CHEBGCN_OK && want_planes == 0 && planes == 4 && !generic4_fits(((nactive + 63) / 64) * 64, g->Mp / 4) &&
        planes_for(M) >= 2) {
        // big graph, automatic choice: small launches take the two-plane kernel (pick_ell, common.h)
        rc = build_ell(M, g->Mp, 2, active, rp, ci, va, &g->fwd2);
        if (rc == CHEBGCN_OK) rc = build_ell(M, g->Mp, 2, active, trp, tci, tva, &g->adj2);
        g->has_alt2 = rc == CHEBGCN_OK;
    }
    if (rc == CHEBGCN_OK && g->Mp <= 384) {
        rc = build_fs_records(M, g->Mp, rp, ci, va, &g->fwd);
        if (rc == CHEBGCN_OK) rc = build_fs_records(M, g->Mp, trp, tci, tva, &g->adj);
    }
    // rows of L~ and of L~^T sorted by descending length (the caller relabelled the vertices: graph.length_order in the
    // Python host), isolated vertices last: the ordered images
    // (also where the regular images do not fit the LDS -- more than 20478 vertices of which at most 20476 are active)
    if (rc == CHEBGCN_OK && want_planes == 0) {
        const int nf = sorted_rows(M, rp), na = sorted_rows(M, trp);
        int NT = 0, NQ = 0, NG = 0, PL = 0;
        if (nf >= 0 && nf == na && nf == nactive && ordered_shape(g->Mp / 4, (nactive + 3) / 4, &NT, &NQ, &NG, &PL)) {
            const int SQ = (nactive + 3) / 4;
            rc = build_ell_ordered(M, NT, NQ, NG, SQ, PL, rp, ci, va, &g->ofwd);
            if (rc == CHEBGCN_OK) rc = build_ell_ordered(M, NT, NQ, NG, SQ, PL, trp, tci, tva, &g->oadj);
            g->ord_ok = rc == CHEBGCN_OK;
            // quad levels the kernel shape does not reach (ordered_shape caps NQ at NG + 1): the streamed tail
            g->ofwd.ord_tail = g->oadj.ord_tail = 4 * NT * NQ < g->Mp ? 4 * NT * NQ : 0;
        }
    }
    if (rc != CHEBGCN_OK) {
        chebgcn_graph_destroy(g);
        return rc;
    }
    *out = g;
    return CHEBGCN_OK;
}

// Relabelling of a length-sorted graph INSIDE its classes of equal row length that spreads the gather's LDS reads over the banks
// (host only, no device needed).  The ordered kernel gives vertex v the LDS slot (v & 3)*SQ + (v >> 2) and its row the lane
// (v >> 2) & 63 of block (v >> 2) >> 6, slice v & 3: the label decides both which bank group v's entry lives in and which lane set
// (the 16 -- two planes: 32 -- lanes the LDS serves together) v's row gathers with.  A lane set reads entry position e of its
// rows in as many passes as the fullest bank group holds DISTINCT entries; place_group() chooses the positions, but with the
// neighbours' bank groups as random as a graph leaves them the fullest of 16 groups holds ~1.6x the average, and no placement
// gets below that (round 5: 46 % of the LDS-active cycles of the recurrence were conflict cycles).  Here the labels themselves
// are chosen: vertices of ONE lane set (whose own slots always cover every bank group once) swap labels pairwise -- their rows
// stay in the set, only their bank groups as gather TARGETS change -- whenever that lowers the sum over all lane sets of the
// squared bank-group counts of the entries they gather.  Deterministic (fixed sweep order), rows stay sorted by length.
// perm_out[new label] = old label.  stats (optional, 3 values): sum over the lane sets of the fullest bank group's count
// before, after, and the number of swaps.
extern "C" int chebgcn_bank_order(int M, const int32_t* rowptr, const int32_t* colidx, int sweeps, int32_t* perm_out,
                                  int64_t* stats) {
    CG_REQUIRE(M > 0 && rowptr && perm_out && sweeps >= 0, "bank_order: bad argument");
    const int64_t nnz = rowptr[M];
    CG_REQUIRE(nnz == 0 || colidx, "bank_order: colidx is NULL");
    for (int v = 0; v < M; ++v) perm_out[v] = v;
    if (stats) stats[0] = stats[1] = stats[2] = 0;
    std::vector<int32_t> rp(rowptr, rowptr + M + 1);
    const int nactive = sorted_rows(M, rp);
    if (nactive <= 0) return CHEBGCN_OK;                   // not sorted by descending length: nothing to refine
    int NT = 0, NQ = 0, NG = 0, PL = 0;
    const int SQ = (nactive + 3) / 4;
    if (!ordered_shape(plane_stride(M) / 4, SQ, &NT, &NQ, &NG, &PL)) return CHEBGCN_OK;     // no ordered kernel: nothing to gain
    const int nb = PL == 4 ? 16 : 32, nsets = PL == 4 ? 4 : 2;
    auto lane_set = [&](int lane) {
        if (PL != 4) return lane >> 5;
        static const int blk[16] = {0, 1, 1, 0, 1, 0, 0, 1, 2, 3, 3, 2, 3, 2, 2, 3};
        return blk[lane >> 2];
    };
    auto cls_of = [&](int label) { return (int)(((int64_t)(label & 3) * SQ + (label >> 2)) % nb); };
    auto set_of = [&](int label) { const int q = label >> 2; return (((q >> 6) * 4 + (label & 3)) * nsets) + lane_set(q & 63); };
    const int nset_ids = ((SQ + 63) / 64) * 4 * nsets;
    // who gathers from u: the rows r with u in row r (the transpose's structure)
    std::vector<int32_t> tp(M + 1, 0), tr(nnz);
    for (int64_t e = 0; e < nnz; ++e) {
        CG_REQUIRE(colidx[e] >= 0 && colidx[e] < M, "bank_order: column out of range");
        tp[colidx[e] + 1]++;
    }
    for (int v = 0; v < M; ++v) tp[v + 1] += tp[v];
    {
        std::vector<int32_t> cur(tp.begin(), tp.end() - 1);
        for (int r = 0; r < M; ++r)
            for (int e = rp[r]; e < rp[r + 1]; ++e) tr[cur[colidx[e]]++] = r;
    }
    std::vector<int32_t> lab(M), at(M);                    // vertex -> label, label -> vertex
    for (int v = 0; v < M; ++v) lab[v] = at[v] = v;
    std::vector<int32_t> H((size_t)nset_ids * nb, 0);      // entries of lane set s that point into bank group k
    for (int r = 0; r < nactive; ++r)
        for (int e = rp[r]; e < rp[r + 1]; ++e)
            if (colidx[e] < nactive) H[(size_t)set_of(r) * nb + cls_of(colidx[e])]++;
    auto sum_max = [&]() {
        int64_t t = 0;
        for (int s = 0; s < nset_ids; ++s) t += *std::max_element(H.begin() + (size_t)s * nb, H.begin() + (size_t)(s + 1) * nb);
        return t;
    };
    if (stats) stats[0] = sum_max();
    // the labels of every lane set
    std::vector<std::vector<int32_t>> members(nset_ids);
    for (int l = 0; l < nactive; ++l) members[set_of(l)].push_back(l);
    auto len_of = [&](int v) { return rp[v + 1] - rp[v]; };
    int64_t swaps = 0;
    for (int sweep = 0; sweep < sweeps; ++sweep) {
        int64_t moved = 0;
        for (int s = 0; s < nset_ids; ++s) {
            const auto& mem = members[s];
            for (size_t a = 0; a < mem.size(); ++a)
                for (size_t b = a + 1; b < mem.size(); ++b) {
                    const int la = mem[a], lb = mem[b];
                    const int x = at[la], y = at[lb];
                    if (len_of(x) != len_of(y)) continue;
                    const int ca = cls_of(la), cb = cls_of(lb);
                    if (ca == cb) continue;
                    // rows that gather from x lose an entry in group ca and gain one in cb; those that gather from y the reverse
                    int64_t d = 0;
                    for (int e = tp[x]; e < tp[x + 1]; ++e) {
                        int32_t* h = &H[(size_t)set_of(lab[tr[e]]) * nb];
                        d += 2 * (h[cb] - h[ca]) + 2;
                        h[ca]--; h[cb]++;
                    }
                    for (int e = tp[y]; e < tp[y + 1]; ++e) {
                        int32_t* h = &H[(size_t)set_of(lab[tr[e]]) * nb];
                        d += 2 * (h[ca] - h[cb]) + 2;
                        h[cb]--; h[ca]++;
                    }
                    if (d < 0) {
                        lab[x] = lb; lab[y] = la;
                        at[la] = y; at[lb] = x;
                        ++moved;
                    } else {
                        for (int e = tp[x]; e < tp[x + 1]; ++e) {
                            int32_t* h = &H[(size_t)set_of(lab[tr[e]]) * nb];
                            h[ca]++; h[cb]--;
                        }
                        for (int e = tp[y]; e < tp[y + 1]; ++e) {
                            int32_t* h = &H[(size_t)set_of(lab[tr[e]]) * nb];
                            h[cb]++; h[ca]--;
                        }
                    }
                }
        }
        swaps += moved;
        if (moved == 0) break;
    }
    for (int l = 0; l < M; ++l) perm_out[l] = at[l];
    if (stats) { stats[1] = sum_max(); stats[2] = swaps; }
    return CHEBGCN_OK;
}

extern "C" void chebgcn_graph_destroy(chebgcn_graph* g) {
    if (!g) return;
    free_ell(g->fwd);
    free_ell(g->adj);
    free_ell(g->fwd2);
    free_ell(g->adj2);
    free_ell(g->ofwd);
    free_ell(g->oadj);
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
        case 9: *value = g->fwd.cost_before; break;     // LDS cycles units of one gather pass, caller's entry order
        case 10: *value = g->fwd.cost_after; break;     // ... after the bank-aware placement
        case 11: *value = g->fwd.cost_ideal; break;     // ... without any conflict
        case 5: *value = g->fwd.max_len; break;
        case 6: *value = g->fwd.planes; break;                               // planes per workgroup (0, 2, 4)
        case 7: *value = g->fwd.nranked; break;                              // rows in the LDS image
        case 8: *value = (int64_t)g->fwd.lds_entries * 4 * g->fwd.planes; break;   // LDS bytes of the image
        case 12: *value = g->ord_ok; break;                                  // 1: rows sorted by length, ordered kernels in use
        case 13: *value = g->ord_ok ? g->ofwd.cost_before : 0; break;        // items 9..11 of the ordered image
        case 14: *value = g->ord_ok ? g->ofwd.cost_after : 0; break;
        case 15: *value = g->ord_ok ? g->ofwd.cost_ideal : 0; break;
        case 16: *value = g->ord_ok ? g->ofwd.planes : 0; break;
        case 17: *value = g->ord_ok ? g->ofwd.ord_tail : 0; break;             // first vertex of the streamed tail of the ordered image, 0 = none
        default: return fail(CHEBGCN_EINVAL, "graph_query: unknown item %d", what);
    }
    return CHEBGCN_OK;
}
