// Host side of the C ABI under AddressSanitizer + UBSan ("race detection / sanitizers"): the
// execution-plan builder, its export, the tuning table and the error path, driven with adversarial row-pointer vectors.
// Built by tests/test_host_cpu.py from sgl_amd/csrc/sgl_core.cpp with g++ -fsanitize=address,undefined and again with
// -fsanitize=thread; any heap overflow, use-after-free, leak, undefined behaviour or data race makes the process exit non-zero.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "../../include/sgl_hip.h"

#define CHECK(c)                                                     \
    do {                                                             \
        if (!(c)) {                                                  \
            fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #c); \
            exit(2);                                                 \
        }                                                            \
    } while (0)

static void one_plan(const std::vector<int64_t> &rp, int item_nnz, int long_nnz) {
    const int64_t n = (int64_t)rp.size() - 1;
    sgl_plan_t *p = nullptr;
    CHECK(sgl_plan_build(&p, rp.data(), n, item_nnz, long_nnz) == SGL_OK && p);
    int64_t c[8];
    CHECK(sgl_plan_counts(p, c) == SGL_OK && c[5] == n);
    std::vector<int32_t> items(2 * c[0]), plen(c[1]), prow(c[1]), lrow(c[2]), lfirst(c[2] + 1);
    std::vector<int64_t> pbeg(c[1]);
    CHECK(sgl_plan_export(p, items.data(), pbeg.data(), plen.data(), prow.data(), lrow.data(), lfirst.data()) == SGL_OK);
    // every row exactly once: in an item, or as a long row whose pieces tile its non-zeros
    std::vector<char> seen(n, 0);
    for (int64_t i = 0; i < c[0]; ++i) {
        CHECK(items[2 * i] < items[2 * i + 1] && items[2 * i + 1] - items[2 * i] <= 63);
        for (int32_t r = items[2 * i]; r < items[2 * i + 1]; ++r) CHECK(!seen[r]++);
    }
    for (int64_t l = 0; l < c[2]; ++l) {
        CHECK(!seen[lrow[l]]++);
        int64_t at = rp[lrow[l]];
        for (int32_t q = lfirst[l]; q < lfirst[l + 1]; ++q) {
            CHECK(prow[q] == lrow[l] && pbeg[q] == at && plen[q] > 0 && (long_nnz <= 0 || plen[q] <= long_nnz));
            at += plen[q];
        }
        CHECK(at == rp[lrow[l] + 1]);
    }
    for (int64_t r = 0; r < n; ++r) CHECK(seen[r] == 1);
    // NULL outputs are allowed
    CHECK(sgl_plan_export(p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == SGL_OK);
    sgl_plan_destroy(p);
}

int main() {
    std::mt19937_64 rng(7);
    for (int trial = 0; trial < 60; ++trial) {
        const int64_t n = 1 + rng() % 3000;
        std::vector<int64_t> rp(n + 1, 0);
        for (int64_t i = 0; i < n; ++i) {
            int64_t deg = (rng() % 10 == 0) ? 0 : (int64_t)(rng() % 40);
            if (rng() % 97 == 0) deg = 1000 + rng() % 9000;          // long rows
            rp[i + 1] = rp[i] + deg;
        }
        const int item_nnz[] = {0, 1, 8, 64, 512, 100000};
        const int long_nnz[] = {0, -1, 1, 100, 2048};
        one_plan(rp, item_nnz[rng() % 6], long_nnz[rng() % 5]);
    }
    one_plan({0}, 0, 0);                                              // empty matrix
    one_plan({0, 0, 0, 0}, 16, 4);                                    // only empty rows
    // error paths: messages are thread-local and never NULL
    sgl_plan_t *p = nullptr;
    std::vector<int64_t> bad = {0, 5, 3};
    CHECK(sgl_plan_build(&p, bad.data(), 2, 0, 0) != SGL_OK && p == nullptr && strlen(sgl_last_error()) > 0);
    CHECK(sgl_plan_build(nullptr, bad.data(), 2, 0, 0) != SGL_OK);
    CHECK(sgl_plan_build(&p, nullptr, 3, 0, 0) != SGL_OK);
    int64_t c[8];
    CHECK(sgl_plan_counts(nullptr, c) != SGL_OK);
    sgl_plan_destroy(nullptr);
    // tuning table from several threads (it is mutex-protected)
    std::vector<std::thread> th;
    for (int t = 0; t < 8; ++t)
        th.emplace_back([t] {
            for (int i = 0; i < 2000; ++i) {
                int64_t v = -1;
                CHECK(sgl_set_tuning("spmm_unroll", (t + i) % 3) == SGL_OK);
                CHECK(sgl_get_tuning("spmm_unroll", &v) == SGL_OK && v >= 0 && v <= 2);
                CHECK(sgl_set_tuning("no_such_knob", 1) != SGL_OK && strlen(sgl_last_error()) > 0);
            }
        });
    for (auto &x : th) x.join();
    CHECK(sgl_set_tuning("spmm_unroll", 0) == SGL_OK);
    // plans built, exported and destroyed from several threads at once, good and bad inputs interleaved: the builder shares no
    // state, the error message is per thread (under -fsanitize=thread any race between these fails the run)
    th.clear();
    for (int t = 0; t < 6; ++t)
        th.emplace_back([t] {
            std::mt19937_64 r2(100 + t);
            for (int i = 0; i < 40; ++i) {
                const int64_t n = 1 + r2() % 800;
                std::vector<int64_t> rp(n + 1, 0);
                for (int64_t k = 0; k < n; ++k) rp[k + 1] = rp[k] + (int64_t)(r2() % 30) + ((r2() % 211 == 0) ? 5000 : 0);
                one_plan(rp, (int)(r2() % 300), (i & 1) ? 64 : 0);
                sgl_plan_t *q = nullptr;
                std::vector<int64_t> bad2 = {0, 9, 2};
                CHECK(sgl_plan_build(&q, bad2.data(), 2, 0, 0) != SGL_OK && q == nullptr && strstr(sgl_last_error(), "row") != nullptr);
            }
        });
    for (auto &x : th) x.join();
    CHECK(sgl_version() >= 100);
    printf("plan_asan: OK\n");
    return 0;
}
