// Reference-signature host shims: a ctypes caller written against the reference's libmatmul.so /
// libcudamatmul.so (sgl/operators/utils.py:10-73) can load libsgl_hip.so instead and run on the MI355X.
// Host pointers in, host pointers out: upload -> device SpMM -> download.  Synchronous by construction
// (the reference calls are, too).  The device-resident API (sgl_csr_create / sgl_spmm_f32) is the fast path.
#include "sgl_common.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <mutex>
#include <thread>

#include <sys/mman.h>

// The unmodified reference calls the CPU symbol K times per propagate() with the SAME adjacency (base_op.py:29-35) and fresh
// `answer` / `mat` arrays (utils.py:31-35).  So: the uploaded CSR, its execution plan and every device / pinned buffer are
// kept between calls and re-used when the adjacency is bit-for-bit the one of the previous call (pointers for indptr /
// indices, full multi-threaded checksums for indptr, indices AND data -- utils.py:32 makes a new float32 copy of `data`
// on every call, so its address never repeats); the dense operands travel through pinned staging buffers filled by a
// team of host threads, each with its own stream, instead of one pageable hipMemcpy.

namespace {

constexpr size_t kChunk = (size_t)16 << 20;   // staging granularity
constexpr int kMaxThreads = 32;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    int reserve(size_t bytes) {
        if (bytes == 0) bytes = 4;
        if (bytes <= cap) return SGL_OK;
        release();
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return sgl::fail((int)e, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        cap = bytes;
        return SGL_OK;
    }
};

int team_size() {
    unsigned hc = std::thread::hardware_concurrency();
    if (hc == 0) hc = 4;
    return (int)std::min<unsigned>(kMaxThreads, std::max<unsigned>(1, hc / 2));
}

// SGL_SHIM_TRACE=1: wall time of every phase of a shim call on stderr (where the PCIe-inclusive time goes)
struct Phases {
    bool on = std::getenv("SGL_SHIM_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    std::string log;
    void mark(const char *name) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        char buf[96];
        snprintf(buf, sizeof(buf), " %s=%.2fms", name, std::chrono::duration<double, std::milli>(t1 - t0).count());
        log += buf;
        t0 = t1;
    }
    ~Phases() {
        if (on) fprintf(stderr, "[sgl shim]%s\n", log.c_str());
    }
};

// A freshly allocated host destination is faulted in by the copy itself: ask for huge pages on its 2 MiB-aligned interior
// (500 x fewer faults where transparent huge pages are in "madvise" / "always" mode; a no-op elsewhere).
void want_huge_pages(void *p, size_t bytes) {
    if (bytes < ((size_t)4 << 20)) return;
    const uintptr_t two_mb = (uintptr_t)2 << 20;
    const uintptr_t a = ((uintptr_t)p + two_mb - 1) & ~(two_mb - 1), b = ((uintptr_t)p + bytes) & ~(two_mb - 1);
    if (b > a) (void)madvise(reinterpret_cast<void *>(a), b - a, MADV_HUGEPAGE);
}

// A team of host threads; the caller's thread is member 0.  The members are joined on every path out (an exception thrown by
// member 0 -- bad_alloc in a hash -- must not leave joinable threads behind: std::terminate).
template <typename F>
void run_team(int threads, F fn) {
    std::vector<std::thread> team;
    struct Join {
        std::vector<std::thread> &t;
        ~Join() {
            for (auto &th : t)
                if (th.joinable()) th.join();
        }
    } join{team};
    for (int t = 1; t < threads; ++t) team.emplace_back(fn, t);
    fn(0);
}

struct JoinThread {   // RAII join of one helper thread
    std::thread &t;
    ~JoinThread() {
        if (t.joinable()) t.join();
    }
};

// order-sensitive 64-bit content hash, computed by a team of threads over fixed 1 MiB blocks (so the result does not
// depend on the number of threads)
uint64_t content_hash(const void *p, size_t bytes, int threads) {
    constexpr size_t kBlock = (size_t)1 << 20;
    const size_t blocks = (bytes + kBlock - 1) / kBlock;
    std::vector<uint64_t> part(blocks, 0);
    std::atomic<size_t> next(0);
    run_team(threads, [&](int) {
        for (;;) {
            const size_t b = next.fetch_add(1);
            if (b >= blocks) break;
            const unsigned char *q = (const unsigned char *)p + b * kBlock;
            const size_t len = std::min(kBlock, bytes - b * kBlock);
            uint64_t h0 = 0x9E3779B97F4A7C15ull ^ b, h1 = 0xC2B2AE3D27D4EB4Full, h2 = 0x165667B19E3779F9ull, h3 = 0x27D4EB2F165667C5ull;
            size_t i = 0;
            for (; i + 32 <= len; i += 32) {
                uint64_t w[4];
                memcpy(w, q + i, 32);
                h0 = (h0 ^ w[0]) * 0x100000001B3ull;
                h1 = (h1 ^ w[1]) * 0x100000001B3ull;
                h2 = (h2 ^ w[2]) * 0x100000001B3ull;
                h3 = (h3 ^ w[3]) * 0x100000001B3ull;
            }
            for (; i < len; ++i) h0 = (h0 ^ q[i]) * 0x100000001B3ull;
            uint64_t h = h0 ^ (h1 << 1 | h1 >> 63) ^ (h2 << 2 | h2 >> 62) ^ (h3 << 3 | h3 >> 61);
            h ^= h >> 29;
            part[b] = h * 0xBF58476D1CE4E5B9ull;
        }
    });
    uint64_t h = 0xCBF29CE484222325ull ^ (uint64_t)bytes;
    for (size_t b = 0; b < blocks; ++b) h = (h ^ part[b]) * 0x100000001B3ull;
    return h;
}

bool all_zero(const void *p, size_t bytes, int threads) {
    constexpr size_t kBlock = (size_t)4 << 20;
    const size_t blocks = (bytes + kBlock - 1) / kBlock;
    std::atomic<size_t> next(0);
    std::atomic<bool> nonzero(false);
    run_team(threads, [&](int) {
        for (;;) {
            const size_t b = next.fetch_add(1);
            if (b >= blocks || nonzero.load(std::memory_order_relaxed)) break;
            const unsigned char *q = (const unsigned char *)p + b * kBlock;
            const size_t len = std::min(kBlock, bytes - b * kBlock);
            uint64_t acc = 0;
            size_t i = 0;
            for (; i + 8 <= len; i += 8) {
                uint64_t w;
                memcpy(&w, q + i, 8);
                acc |= w;                       // -0.0f has a bit set: it is NOT "zero" here, the upload then happens
            }
            for (; i < len; ++i) acc |= q[i];
            if (acc) nonzero.store(true, std::memory_order_relaxed);
        }
    });
    return !nonzero.load();
}

struct Shim {
    std::mutex mu;
    int device = 0;                        // every buffer, stream and handle below lives on this device (one Shim per device)
    // cached adjacency
    const void *indptr_p = nullptr, *indices_p = nullptr;
    int64_t n = -1, nnz = -1;
    uint64_t h_indptr = 0, h_indices = 0, h_data = 0;
    sgl_csr_t *handle = nullptr;          // == pieces[0] while a graph is cached
    std::vector<sgl_csr_t *> pieces;       // row pieces of the cached adjacency, each with its own plan
    std::vector<int64_t> piece_rows;       // [pieces + 1] row boundaries
    std::vector<hipEvent_t> piece_done;
    DevBuf d_rp, d_col, d_val, d_x, d_y, d_rp_local;
    // staging
    void *pinned[kMaxThreads] = {};
    hipStream_t streams[kMaxThreads] = {};
    int staged_threads = 0;
    int64_t hits = 0, misses = 0;

    void drop_graph() {
        for (sgl_csr_t *h : pieces) sgl_csr_destroy(h);
        pieces.clear();
        piece_rows.clear();
        handle = nullptr;
        indptr_p = indices_p = nullptr;
        n = nnz = -1;
    }
    // nnz-balanced row pieces (at most kPieces, none for tiny graphs), local row pointers behind one another in d_rp_local
    int build_pieces(const std::vector<int64_t> &rp64, int64_t n_rows, int64_t n_nz) {
        constexpr int kPieces = 8;
        const int want = (n_nz >= ((int64_t)1 << 22)) ? kPieces : 1;
        piece_rows.assign(1, 0);
        for (int p = 1; p < want; ++p) {
            const int64_t target = n_nz * p / want;
            int64_t r = std::lower_bound(rp64.begin(), rp64.end(), target) - rp64.begin();
            r = std::min<int64_t>(std::max<int64_t>(r, piece_rows.back()), n_rows);
            if (r > piece_rows.back() && r < n_rows) piece_rows.push_back(r);
        }
        piece_rows.push_back(n_rows);
        const size_t np = piece_rows.size() - 1;
        std::vector<int64_t> local;
        std::vector<size_t> at(np);
        for (size_t p = 0; p < np; ++p) {
            at[p] = local.size();
            const int64_t base = rp64[piece_rows[p]];
            for (int64_t r = piece_rows[p]; r <= piece_rows[p + 1]; ++r) local.push_back(rp64[r] - base);
        }
        int rc = d_rp_local.reserve(local.size() * sizeof(int64_t));
        if (rc != SGL_OK) return rc;
        SGL_HIP_CHECK(hipMemcpy(d_rp_local.p, local.data(), local.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        for (size_t p = 0; p < np; ++p) {
            const int64_t base = rp64[piece_rows[p]], rows = piece_rows[p + 1] - piece_rows[p];
            const int64_t pnz = rp64[piece_rows[p + 1]] - base;
            sgl_csr_t *h = nullptr;
            rc = sgl_csr_create(&h, rows, n_rows, pnz, (const int64_t *)d_rp_local.p + at[p], (const int32_t *)d_col.p + base,
                                (const float *)d_val.p + base, SGL_CSR_STRICT_ORDER, 0, 0, nullptr);
            if (rc != SGL_OK) return rc;
            pieces.push_back(h);
        }
        while (piece_done.size() < np) {
            hipEvent_t e;
            SGL_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            piece_done.push_back(e);
        }
        handle = pieces[0];
        return SGL_OK;
    }
    int ensure_staging(int threads) {
        for (int t = staged_threads; t < threads; ++t) {
            SGL_HIP_CHECK(hipHostMalloc(&pinned[t], kChunk, hipHostMallocDefault));
            SGL_HIP_CHECK(hipStreamCreateWithFlags(&streams[t], hipStreamNonBlocking));
            staged_threads = t + 1;
        }
        return SGL_OK;
    }
    // host <-> device through the pinned buffers: chunk c is handled end to end by thread c % threads on its own stream
    int copy(void *dev, void *host, size_t bytes, bool to_device, int threads) {
        if (bytes == 0) return SGL_OK;
        int rc = ensure_staging(threads);
        if (rc != SGL_OK) return rc;
        const size_t chunks = (bytes + kChunk - 1) / kChunk;
        std::atomic<int> err((int)hipSuccess);
        run_team(std::min<size_t>(threads, chunks), [&](int t) {
            if (t != 0 && hipSetDevice(device) != hipSuccess) {   // a new thread starts on device 0: bind it to the shim's device
                err.store((int)hipErrorInvalidDevice);
                return;
            }
            for (size_t c = t; c < chunks; c += threads) {
                const size_t off = c * kChunk, len = std::min(kChunk, bytes - off);
                hipError_t e;
                if (to_device) {
                    memcpy(pinned[t], (const char *)host + off, len);
                    e = hipMemcpyAsync((char *)dev + off, pinned[t], len, hipMemcpyHostToDevice, streams[t]);
                    if (e == hipSuccess) e = hipStreamSynchronize(streams[t]);     // the buffer is refilled next
                } else {
                    e = hipMemcpyAsync(pinned[t], (const char *)dev + off, len, hipMemcpyDeviceToHost, streams[t]);
                    if (e == hipSuccess) e = hipStreamSynchronize(streams[t]);
                    if (e == hipSuccess) memcpy((char *)host + off, pinned[t], len);
                }
                if (e != hipSuccess) {
                    err.store((int)e);
                    return;
                }
            }
        });
        if (err.load() != (int)hipSuccess)
            return sgl::fail(err.load(), "staged copy failed: %s", hipGetErrorString((hipError_t)err.load()));
        return SGL_OK;
    }
};

// One Shim per device, created on first use with that device current and kept for the life of the process (the cached device
// memory is released with the context): a process that drives several GPUs never sends one device's memory through another
// device's streams.
Shim &shim(int device) {
    static std::mutex mu;
    static std::vector<Shim *> per_device;
    std::lock_guard<std::mutex> lk(mu);
    if ((size_t)device >= per_device.size()) per_device.resize((size_t)device + 1, nullptr);
    if (!per_device[device]) {
        per_device[device] = new Shim();
        per_device[device]->device = device;
    }
    return *per_device[device];
}

int current_device(int *device) {
    SGL_HIP_CHECK(hipGetDevice(device));
    SGL_REQUIRE(*device >= 0, "no current HIP device");
    return SGL_OK;
}

int host_spmm(float *answer, const float *data, const int *indices, const int *indptr, const float *mat,
              int mat_row, int mat_col, int accumulate) {
    SGL_REQUIRE(mat_row >= 0 && mat_col >= 0, "FloatCSRMulDense*: negative size");
    if (mat_row == 0 || mat_col == 0) return SGL_OK;
    SGL_REQUIRE(answer && indptr && mat, "FloatCSRMulDense*: NULL argument");
    int ndev = 0;
    sgl_device_count(&ndev);
    if (ndev <= 0) return sgl::fail(SGL_ERR_NO_DEVICE, "FloatCSRMulDense*: no HIP device available");
    // The reference passes the SAME n for rows of A, columns of A and rows of `mat` (square adjacency,
    // utils.py:36: mat_row, mat_col = feature.shape).
    const int64_t n = mat_row, d = mat_col;
    const int64_t nnz = indptr[n];
    SGL_REQUIRE(nnz >= 0 && indptr[0] == 0, "FloatCSRMulDense*: bad indptr");
    SGL_REQUIRE(nnz == 0 || (data && indices), "FloatCSRMulDense*: NULL data/indices");
    int device = 0, rc;
    if ((rc = current_device(&device)) != SGL_OK) return rc;
    Shim &S = shim(device);
    std::lock_guard<std::mutex> lk(S.mu);
    const int threads = team_size();
    Phases ph;

    // ---- the dense input starts travelling at once: it does not depend on whether the adjacency is cached -----------------
    const size_t dense = (size_t)n * d * sizeof(float);
    if ((rc = S.d_x.reserve(dense)) != SGL_OK) return rc;
    if ((rc = S.d_y.reserve(dense)) != SGL_OK) return rc;
    int rc_x = SGL_OK;
    std::string err_x;
    if ((rc = S.ensure_staging(threads)) != SGL_OK) return rc;   // pinned buffers and streams are created by THIS thread, on `device`
    std::thread up_x([&] {
        if (hipSetDevice(device) != hipSuccess) {
            rc_x = sgl::fail(SGL_ERR_NO_DEVICE, "FloatCSRMulDense*: the upload thread could not select device %d", device);
            err_x = sgl::get_error();
            return;
        }
        rc_x = S.copy(S.d_x.p, const_cast<float *>(mat), dense, true, threads);
        if (rc_x != SGL_OK) err_x = sgl::get_error();     // the error text is thread-local: carry it over
    });
    JoinThread join_up_x{up_x};                            // joined on every path out, also when a hash below throws

    // ---- meanwhile: is the adjacency bit-for-bit the previous call's?  is `answer` all zeros? -----------------------------
    const uint64_t hp = content_hash(indptr, ((size_t)n + 1) * sizeof(int), threads);
    const uint64_t hi = content_hash(indices, (size_t)nnz * sizeof(int), threads);
    const uint64_t hd = content_hash(data, (size_t)nnz * sizeof(float), threads);
    const bool hit = S.handle && S.indptr_p == indptr && S.indices_p == indices && S.n == n && S.nnz == nnz &&
                     S.h_indptr == hp && S.h_indices == hi && S.h_data == hd;
    int acc = accumulate;
    if (acc && all_zero(answer, dense, threads)) acc = 0;   // the reference pre-zeroes `answer` (utils.py:31): 0 + A.X = A.X
    ph.mark("hash+zero_scan");
    up_x.join();
    ph.mark("x_upload_rest");
    if (rc_x != SGL_OK) return sgl::fail(rc_x, "%s", err_x.c_str());

    if (!hit) {
        ++S.misses;
        S.drop_graph();
        std::vector<int64_t> rp64((size_t)n + 1);
        for (int64_t i = 0; i <= n; ++i) rp64[i] = indptr[i];
        if ((rc = S.d_rp.reserve(rp64.size() * sizeof(int64_t))) != SGL_OK) return rc;
        if ((rc = S.d_col.reserve((size_t)nnz * sizeof(int32_t))) != SGL_OK) return rc;
        if ((rc = S.d_val.reserve((size_t)nnz * sizeof(float))) != SGL_OK) return rc;
        if ((rc = S.copy(S.d_rp.p, rp64.data(), rp64.size() * sizeof(int64_t), true, threads)) != SGL_OK) return rc;
        if ((rc = S.copy(S.d_col.p, const_cast<int *>(indices), (size_t)nnz * sizeof(int32_t), true, threads)) != SGL_OK) return rc;
        if ((rc = S.copy(S.d_val.p, const_cast<float *>(data), (size_t)nnz * sizeof(float), true, threads)) != SGL_OK) return rc;
        // strict order: the shim promises the reference's exact per-row fmaf chain.  The rows are cut into pieces with
        // their own plans so that the download of a finished piece overlaps the computation of the next ones.
        if ((rc = S.build_pieces(rp64, n, nnz)) != SGL_OK) {
            S.drop_graph();
            return rc;
        }
        S.indptr_p = indptr;
        S.indices_p = indices;
        S.n = n;
        S.nnz = nnz;
        S.h_indptr = hp;
        S.h_indices = hi;
        S.h_data = hd;
        ph.mark("adjacency_upload+plan");
    } else {
        ++S.hits;
    }

    if (acc && (rc = S.copy(S.d_y.p, answer, dense, true, threads)) != SGL_OK) return rc;
    if (acc) ph.mark("answer_upload");
    want_huge_pages(answer, dense);     // utils.py:31 hands over a fresh np.zeros: its pages are faulted in by the download
    // all pieces are queued at once; each piece's rows are downloaded as soon as its event has fired
    const size_t np = S.pieces.size();
    for (size_t p = 0; p < np; ++p) {
        const int64_t r0 = S.piece_rows[p], r1 = S.piece_rows[p + 1];
        rc = sgl_spmm_f32(S.pieces[p], (const float *)S.d_x.p, d, (float *)S.d_y.p + r0 * d, d, d, acc, nullptr);
        if (rc != SGL_OK) return rc;
        SGL_HIP_CHECK(hipEventRecord(S.piece_done[p], nullptr));
        (void)r1;
    }
    for (size_t p = 0; p < np; ++p) {
        const int64_t r0 = S.piece_rows[p], r1 = S.piece_rows[p + 1];
        SGL_HIP_CHECK(hipEventSynchronize(S.piece_done[p]));
        if (p == 0) ph.mark("first_piece");
        if ((rc = S.copy((float *)S.d_y.p + r0 * d, answer + r0 * d, (size_t)(r1 - r0) * d * sizeof(float), false, threads)) != SGL_OK)
            return rc;
    }
    ph.mark("remaining_compute+download");
    return SGL_OK;
}

}  // namespace

// matmul.h:5 / matmul.c:23-40: answer += A . mat
SGL_EXPORT void FloatCSRMulDenseOMP(float answer[], float data[], int indices[], int indptr[], float mat[], int mat_row,
                                    int mat_col) {
    sgl::set_error("%s", "");
    (void)host_spmm(answer, data, indices, indptr, mat, mat_row, mat_col, /*accumulate=*/1);
}

// cudamatmul.c:28-146: answer = A . mat (alpha = 1, beta = 0); EXIT_SUCCESS / EXIT_FAILURE
SGL_EXPORT int FloatCSRMulDense(float answer[], int data_nnz, float data[], int indices[], int indptr[], float mat[],
                                int mat_row, int mat_col) {
    sgl::set_error("%s", "");
    if (indptr && mat_row >= 0 && indptr[mat_row] != data_nnz) {
        sgl::set_error("FloatCSRMulDense: data_nnz=%d but indptr[mat_row]=%d", data_nnz, indptr[mat_row]);
        return 1;
    }
    return host_spmm(answer, data, indices, indptr, mat, mat_row, mat_col, /*accumulate=*/0) == SGL_OK ? 0 : 1;
}

// number of calls served from / not served from the cached adjacency (tests, INTEGRATION.md figures)
// Device memory -> pageable host memory through the same team of threads and pinned staging buffers the shims use: for
// callers that must receive ordinary host arrays -- the reference contract's CPU hop tensors (base_op.py:36).  Synchronous.
// (The other direction gains nothing over torch's own pageable upload, measured; there is no sgl_upload.)
SGL_EXPORT int sgl_download(void *h_dst, const void *d_src, int64_t bytes, void *stream) {
    SGL_REQUIRE(bytes >= 0 && (bytes == 0 || (h_dst && d_src)), "sgl_download: bad arguments");
    SGL_HIP_CHECK(hipStreamSynchronize(sgl::as_stream(stream)));   // the producer's work is complete before the copy starts
    want_huge_pages(h_dst, (size_t)bytes);
    int device = 0, rc;
    if ((rc = current_device(&device)) != SGL_OK) return rc;
    Shim &S = shim(device);
    std::lock_guard<std::mutex> lk(S.mu);
    return S.copy(const_cast<void *>(d_src), h_dst, (size_t)bytes, false, team_size());
}

SGL_EXPORT int sgl_shim_cache_stats(int64_t *hits, int64_t *misses) {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0) device = 0;    // no GPU: the (empty) statistics of device 0
    Shim &S = shim(device);
    std::lock_guard<std::mutex> lk(S.mu);
    if (hits) *hits = S.hits;
    if (misses) *misses = S.misses;
    return SGL_OK;
}

// Order-sensitive 64-bit hash of a HOST buffer, computed by the shims' team of threads (the one they key their cached adjacency
// on): lets the Python operator layer fingerprint a scipy matrix's index / value arrays in full at memory speed, without an
// optional dependency.  Host-only: needs no GPU.
SGL_EXPORT int sgl_content_hash(const void *h_ptr, int64_t bytes, uint64_t *out) {
    SGL_REQUIRE(out && bytes >= 0 && (bytes == 0 || h_ptr), "sgl_content_hash: bad arguments");
    try {
        *out = content_hash(h_ptr, (size_t)bytes, team_size());
    } catch (const std::exception &e) {
        return sgl::fail(SGL_ERR_ALLOC, "sgl_content_hash: %s", e.what());
    }
    return SGL_OK;
}
