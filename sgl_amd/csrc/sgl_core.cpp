// Host-only part of libsgl_hip.so: error text, tuning knobs, and the SpMM execution-plan builder.
// No device code here, so these entry points work (and are unit-tested) on a machine without a GPU.
#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <thread>

#include "sgl_common.h"

namespace sgl {

static thread_local std::string g_err;

void set_error(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}
const char *get_error() { return g_err.c_str(); }

static std::mutex g_tune_mu;
static std::map<std::string, int64_t> &tune_map() {
    static std::map<std::string, int64_t> m = {
        {"spmm_unroll", 0},      // 0 = auto; 1 / 3 / 2 = low / mid / high number of gathers in flight per lane
        {"spmm_vec", 0},         // 0 = widest legal lane access; 2 / 1 = cap at 8 / 4 bytes per lane
        {"spmm_nt", 0},          // 1 = non-temporal loads for the CSR stream / stores of Y
        {"spmm_group", 0},       // 0 = auto; force lanes-per-feature-row (8/16/32/64)
        {"spmm_waves", 0},       // 0 = default (4 waves per workgroup)
        {"spmm_xcd_remap", 1},   // contiguous row ranges per XCD
        {"spmm_heavy_first", 1}, // plan time: work items that end in a heavy row are issued first inside every XCD range
        {"agg_blocks", 0},       // 0 = one 16-byte element per thread (grid capped at 2^22 blocks); > 0 caps the grid of the streaming aggregators (grid-stride)
        {"nafs_fused", 1},       // 0 = force the two-pass NAFS path
        {"row_lpr32x2", 1},      // row-wise kernels, 128 < d <= 256: 32 lanes x 2 chunks per row (2 rows per wavefront)
        {"row_whole_lines", 1},    // gate / NAFS / concat outputs: the pad columns of a row pitch (< one line beyond d) are written as zeros
        {"row_narrow_groups", 1},  // row-wise kernels: 16 lanes x 3 / 8 lanes x 5 chunks when that at least halves the idle lane slots
                                   // (0 = off, 1 = both, 2 = 16 x 3 only, 3 = 8 x 5 only)
        {"gather_lpr", 0},       // row gather / scatter: 0 = the lane group that idles the fewest slots; 8 / 16 / 32 / 64 force one
        {"gather_hops_grid", 1},         // sgl_gather_hops_padded_f32: 1 = hop in blockIdx.y (default), 0 = hop loop inside the thread
        {"gather_rows_per_thread", 0},   // 0 = sized so that the grid is about one round of the chip (1 ... 16); else 1 / 2 / 4 / 8 / 16
        {"concat_flat_read", 1}, // whole-rows concat: read whole pitches contiguously across the block's rows (0 = row by row, hop by hop)
        {"concat_lds", 1},       // any-width concat of rows >= 256 floats: assemble the output row in LDS (1 = auto: whole rows per block
                                 // where 1024-float tiles would be < 85 % full, 2 = always tiles, 3 = always whole rows, 0 = funnel-select kernel)
    };
    return m;
}

int64_t tuning(const char *key, int64_t dflt) {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    auto &m = tune_map();
    auto it = m.find(key);
    if (it == m.end()) return dflt;
    return it->second;
}

// Issue order of the work items.  An item closes when it reaches item_nnz non-zeros, so one that ends in a heavy row holds up
// to item_nnz + long_row_nnz of them: several times the usual wavefront lifetime.  Where such an item is issued late, the launch
// ends with a few wavefronts on an otherwise empty chip -- one per cent of the single-GPU launch, but a rank's launch of the
// 8-rank job is only ~3.5 rounds of resident wavefronts (profiles/r03_probe_small_launch.log: -3.5 %).  Inside every XCD's range
// of the item list (the ranges spmm_kernel walks with 4 wavefronts per block) the items holding >= 2 x item_nnz non-zeros
// therefore go first, longest first; the others keep their row order.  Items are whole rows: results do not change.
static void heavy_items_first(Plan &plan, const int64_t *rowptr, int32_t item_nnz) {
    const int64_t n = (int64_t)plan.items.size() / 2;
    if (n < 64) return;
    const int64_t range = (((n + 3) / 4 + 7) / 8) * 4;
    const int64_t heavy = 2 * (int64_t)std::max(item_nnz, 1);
    struct It {
        int32_t b, e;
        int64_t nnz;
    };
    std::vector<It> seg;
    for (int64_t lo = 0; lo < n; lo += range) {
        const int64_t hi = std::min(n, lo + range);
        seg.clear();
        for (int64_t i = lo; i < hi; ++i) {
            const int32_t b = plan.items[2 * i], e = plan.items[2 * i + 1];
            seg.push_back(It{b, e, rowptr[e] - rowptr[b]});
        }
        auto mid = std::stable_partition(seg.begin(), seg.end(), [&](const It &t) { return t.nnz >= heavy; });
        std::stable_sort(seg.begin(), mid, [](const It &x, const It &y) { return x.nnz > y.nnz; });
        for (int64_t i = lo; i < hi; ++i) {
            plan.items[2 * i] = seg[(size_t)(i - lo)].b;
            plan.items[2 * i + 1] = seg[(size_t)(i - lo)].e;
        }
    }
}

// Greedy partition of the rows [r0, r1) into work items (see include/sgl_hip.h, "execution plan"); a fresh item starts at r0.
static int build_segment(Plan &plan, const int64_t *rowptr, int64_t r0, int64_t r1, int32_t item_nnz, int32_t long_row_nnz) {
    const bool split = long_row_nnz > 0;
    int64_t cur_begin = r0, cur_nnz = 0;
    auto close_item = [&](int64_t end) {
        if (end > cur_begin) {
            plan.items.push_back((int32_t)cur_begin);
            plan.items.push_back((int32_t)end);
            plan.max_item_rows = std::max<int64_t>(plan.max_item_rows, end - cur_begin);
            plan.max_item_nnz = std::max<int64_t>(plan.max_item_nnz, cur_nnz);
        }
        cur_begin = end;
        cur_nnz = 0;
    };
    for (int64_t r = r0; r < r1; ++r) {
        const int64_t b = rowptr[r], e = rowptr[r + 1];
        if (e < b) return fail(SGL_ERR_INVALID, "build_plan: row pointers decrease at row %lld", (long long)r);
        const int64_t deg = e - b;
        if (!split && deg >= (int64_t)INT32_MAX)
            return fail(SGL_ERR_UNSUPPORTED, "build_plan: a row with >= 2^31 non-zeros needs long_row_nnz > 0");
        if (split && deg > long_row_nnz) {
            close_item(r);  // rows before the long one
            plan.long_row.push_back((int32_t)r);
            for (int64_t p = b; p < e; p += long_row_nnz) {
                Piece pc;
                pc.begin = p;
                pc.len = (int32_t)std::min<int64_t>(long_row_nnz, e - p);
                pc.row = (int32_t)r;
                plan.pieces.push_back(pc);
            }
            plan.long_first.push_back((int32_t)plan.pieces.size());
            cur_begin = r + 1;
            cur_nnz = 0;
            continue;
        }
        // a row that would push the item past 2^31-1 relative offsets cannot happen when split is on
        // (deg <= long_row_nnz); without splitting close the item first so offsets stay 32-bit.
        if (cur_nnz > 0 && cur_nnz + deg >= (int64_t)INT32_MAX) close_item(r);
        cur_nnz += deg;
        if (cur_nnz >= item_nnz || (r + 1 - cur_begin) >= kMaxItemRows) close_item(r + 1);
    }
    close_item(r1);
    return SGL_OK;
}

// Rows are cut into segments of kPlanSegmentRows; every segment is partitioned on its own (a fresh item starts at its first row:
// at most one short item per 2^20 rows) by a team of host threads and the pieces are laid end to end.  The plan depends on the
// segment size only, never on the number of threads; matrices of up to two segments -- everything but the 10^7 ... 10^8-row
// blocks of the papers100M-sized jobs, where the serial loop was hundreds of milliseconds of setup -- are one segment: the plan
// they always had.  Items are whole rows and long rows are cut per row: results do not depend on any of this.
constexpr int64_t kPlanSegmentRows = (int64_t)1 << 20;

int build_plan(Plan &plan, const int64_t *rowptr, int64_t n_rows, int32_t item_nnz, int32_t long_row_nnz) {
    if (n_rows < 0 || (n_rows > 0 && rowptr == nullptr)) return fail(SGL_ERR_INVALID, "build_plan: bad arguments");
    if (n_rows >= (int64_t)INT32_MAX) return fail(SGL_ERR_UNSUPPORTED, "build_plan: n_rows >= 2^31 (shard the matrix)");
    if (item_nnz <= 0) item_nnz = kDefaultItemNnz;
    plan = Plan();
    plan.n_rows = n_rows;
    plan.long_first.push_back(0);
    if (n_rows <= 2 * kPlanSegmentRows) {
        const int rc = build_segment(plan, rowptr, 0, n_rows, item_nnz, long_row_nnz);
        if (rc != SGL_OK) return rc;
    } else {
        const int64_t n_seg = (n_rows + kPlanSegmentRows - 1) / kPlanSegmentRows;
        std::vector<Plan> segs((size_t)n_seg);
        std::vector<int> rcs((size_t)n_seg, SGL_OK);
        std::vector<std::string> errs((size_t)n_seg);
        const int n_thr = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), (int64_t)16, n_seg}));
        std::atomic<int64_t> next{0};
        auto work = [&]() {
            for (int64_t sidx = next.fetch_add(1); sidx < n_seg; sidx = next.fetch_add(1)) {
                const int64_t r0 = sidx * kPlanSegmentRows, r1 = std::min(n_rows, r0 + kPlanSegmentRows);
                rcs[(size_t)sidx] = build_segment(segs[(size_t)sidx], rowptr, r0, r1, item_nnz, long_row_nnz);
                if (rcs[(size_t)sidx] != SGL_OK) errs[(size_t)sidx] = get_error();      // the error text is per thread
            }
        };
        std::vector<std::thread> team;
        for (int t = 1; t < n_thr; ++t) team.emplace_back(work);
        work();
        for (auto &t : team) t.join();
        size_t n_items = 0, n_pieces = 0, n_long = 0;
        for (int64_t sidx = 0; sidx < n_seg; ++sidx) {
            if (rcs[(size_t)sidx] != SGL_OK) return fail(rcs[(size_t)sidx], "%s", errs[(size_t)sidx].c_str());
            n_items += segs[(size_t)sidx].items.size();
            n_pieces += segs[(size_t)sidx].pieces.size();
            n_long += segs[(size_t)sidx].long_row.size();
        }
        plan.items.reserve(n_items);
        plan.pieces.reserve(n_pieces);
        plan.long_row.reserve(n_long);
        plan.long_first.reserve(n_long + 1);
        for (auto &sg : segs) {
            const int32_t piece0 = (int32_t)plan.pieces.size();
            plan.items.insert(plan.items.end(), sg.items.begin(), sg.items.end());
            plan.pieces.insert(plan.pieces.end(), sg.pieces.begin(), sg.pieces.end());
            plan.long_row.insert(plan.long_row.end(), sg.long_row.begin(), sg.long_row.end());
            for (int32_t lf : sg.long_first) plan.long_first.push_back(piece0 + lf);   // (a segment's list has no leading 0)
            plan.max_item_rows = std::max(plan.max_item_rows, sg.max_item_rows);
            plan.max_item_nnz = std::max(plan.max_item_nnz, sg.max_item_nnz);
            sg = Plan();
        }
    }
    if (tuning("spmm_heavy_first", 1) != 0) heavy_items_first(plan, rowptr, item_nnz);
    return SGL_OK;
}

}  // namespace sgl

SGL_EXPORT int sgl_version(void) { return 100; }

SGL_EXPORT const char *sgl_last_error(void) { return sgl::get_error(); }

SGL_EXPORT int sgl_device_count(int *count) {
    if (!count) return sgl::fail(SGL_ERR_INVALID, "sgl_device_count: NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return SGL_OK;
}

SGL_EXPORT int sgl_set_tuning(const char *key, int64_t value) {
    if (!key) return sgl::fail(SGL_ERR_INVALID, "sgl_set_tuning: NULL key");
    std::lock_guard<std::mutex> lk(sgl::g_tune_mu);
    auto &m = sgl::tune_map();
    auto it = m.find(key);
    if (it == m.end()) return sgl::fail(SGL_ERR_INVALID, "sgl_set_tuning: unknown key '%s'", key);
    it->second = value;
    return SGL_OK;
}

SGL_EXPORT int sgl_get_tuning(const char *key, int64_t *value) {
    if (!key || !value) return sgl::fail(SGL_ERR_INVALID, "sgl_get_tuning: NULL");
    std::lock_guard<std::mutex> lk(sgl::g_tune_mu);
    auto &m = sgl::tune_map();
    auto it = m.find(key);
    if (it == m.end()) return sgl::fail(SGL_ERR_INVALID, "sgl_get_tuning: unknown key '%s'", key);
    *value = it->second;
    return SGL_OK;
}

SGL_EXPORT int sgl_plan_build(sgl_plan_t **out, const int64_t *h_rowptr, int64_t n_rows, int32_t item_nnz,
                              int32_t long_row_nnz) {
    if (!out) return sgl::fail(SGL_ERR_INVALID, "sgl_plan_build: NULL out");
    *out = nullptr;
    sgl_plan_t *p = new (std::nothrow) sgl_plan_t();
    if (!p) return sgl::fail(SGL_ERR_ALLOC, "sgl_plan_build: out of memory");
    int rc = sgl::build_plan(p->p, h_rowptr, n_rows, item_nnz, long_row_nnz);
    if (rc != SGL_OK) {
        delete p;
        return rc;
    }
    *out = p;
    return SGL_OK;
}

SGL_EXPORT int sgl_plan_counts(const sgl_plan_t *plan, int64_t counts[8]) {
    if (!plan || !counts) return sgl::fail(SGL_ERR_INVALID, "sgl_plan_counts: NULL");
    memset(counts, 0, 8 * sizeof(int64_t));
    counts[0] = (int64_t)plan->p.items.size() / 2;
    counts[1] = (int64_t)plan->p.pieces.size();
    counts[2] = (int64_t)plan->p.long_row.size();
    counts[3] = plan->p.max_item_rows;
    counts[4] = plan->p.max_item_nnz;
    counts[5] = plan->p.n_rows;
    return SGL_OK;
}

SGL_EXPORT int sgl_plan_export(const sgl_plan_t *plan, int32_t *h_items, int64_t *h_piece_begin, int32_t *h_piece_len,
                               int32_t *h_piece_row, int32_t *h_long_row, int32_t *h_long_first) {
    if (!plan) return sgl::fail(SGL_ERR_INVALID, "sgl_plan_export: NULL plan");
    const sgl::Plan &p = plan->p;
    if (h_items && !p.items.empty()) memcpy(h_items, p.items.data(), p.items.size() * sizeof(int32_t));
    for (size_t i = 0; i < p.pieces.size(); ++i) {
        if (h_piece_begin) h_piece_begin[i] = p.pieces[i].begin;
        if (h_piece_len) h_piece_len[i] = p.pieces[i].len;
        if (h_piece_row) h_piece_row[i] = p.pieces[i].row;
    }
    if (h_long_row && !p.long_row.empty()) memcpy(h_long_row, p.long_row.data(), p.long_row.size() * sizeof(int32_t));
    if (h_long_first) memcpy(h_long_first, p.long_first.data(), p.long_first.size() * sizeof(int32_t));
    return SGL_OK;
}

SGL_EXPORT void sgl_plan_destroy(sgl_plan_t *plan) { delete plan; }
