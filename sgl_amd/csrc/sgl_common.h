// Internal helpers shared by the translation units of libsgl_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sgl_hip.h"

#define SGL_EXPORT extern "C" __attribute__((visibility("default")))

namespace sgl {

// thread-local last-error text (sgl_last_error)
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
const char *get_error();

inline int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
inline int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    set_error("%s", buf);
    return code;
}

#define SGL_HIP_CHECK(expr)                                                                          \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return ::sgl::fail((int)_e, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                               __LINE__);                                                            \
    } while (0)

#define SGL_REQUIRE(cond, ...)                                     \
    do {                                                           \
        if (!(cond)) return ::sgl::fail(SGL_ERR_INVALID, __VA_ARGS__); \
    } while (0)

// ---- execution plan (host) ---------------------------------------------------------------------------------
struct Piece {
    int64_t begin;  // first non-zero (absolute)
    int32_t len;    // number of non-zeros
    int32_t row;    // output row
};

struct Plan {
    int64_t n_rows = 0;
    std::vector<int32_t> items;        // (row_begin, row_end) pairs
    std::vector<Piece> pieces;         // pieces of long rows, in row / storage order
    std::vector<int32_t> long_row;     // rows that were split
    std::vector<int32_t> long_first;   // [n_long+1] first piece of each long row
    int64_t max_item_rows = 0, max_item_nnz = 0;
};

constexpr int kMaxItemRows = 63;        // row-pointer window of one wavefront: 64 lanes hold rows+1 offsets
constexpr int kDefaultItemNnz = 512;
constexpr int kDefaultLongRowNnz = 2048;
constexpr int64_t kSmallLaunchNnz = 100000000;   // below this a launch gets 256-nnz work items (sgl_csr_create)

int build_plan(Plan &plan, const int64_t *rowptr, int64_t n_rows, int32_t item_nnz, int32_t long_row_nnz);

// tuning knobs (sgl_set_tuning)
int64_t tuning(const char *key, int64_t dflt);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// A HIP launch carries at most 2^32 - 1 threads per grid dimension; beyond that the launch is silently truncated on this
// stack (found the hard way at papers100M size).  One-thread-per-element launchers check this, the rest stride.
inline bool launch_fits(int64_t blocks, int64_t threads_per_block) {
    return blocks >= 0 && blocks * threads_per_block < ((int64_t)1 << 32) && blocks < (int64_t)INT32_MAX;
}

}  // namespace sgl

struct sgl_plan {
    sgl::Plan p;
};
