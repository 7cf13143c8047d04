// The multi-peer exchange of the C ABI (sgl_allgather_rows, sgl_exchange_rows) executed with 2..8 ranks and NO GPU.
//
// The library resolves ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd from the host process at run time
// (dlsym(RTLD_DEFAULT), sgl_exchange.hip).  This program IS that host: it exports a mock RCCL -- a communicator is a rank in a
// shared mailbox world, a group collects its operations and executes them at ncclGroupEnd (sends are buffered into the
// (src, dst) mailbox, receives block until their message is there), memcpy on host memory -- loads libsgl_hip.so and drives
// one host thread per rank, the way a one-thread-per-GPU host would.  Checked: every replica / ghost range equals the
// reference all-gather, unequal and empty blocks, the exact set of operations (peers, element counts, no zero-size op,
// dtype float32, the caller's stream handed through, everything inside ONE group), and an injected RCCL failure coming back
// as an error code with the mock's message.
//
//   g++ -std=c++17 -O1 -rdynamic exchange_mock.cpp -ldl -pthread -o exchange_mock && ./exchange_mock path/to/libsgl_hip.so
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#define CHECK(c)                                                         \
    do {                                                                 \
        if (!(c)) {                                                      \
            fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #c); \
            exit(2);                                                     \
        }                                                                \
    } while (0)

namespace {

struct World {
    std::mutex mu;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::deque<std::vector<float>>> box;   // (src, dst) -> messages in order
};

struct Op {
    bool send;
    const void *sbuf;
    void *rbuf;
    size_t count;
    int peer;
};

struct Comm {
    int rank, world;
    World *w;
    int fail_on_send_to = -1;              // error injection
    // accounting (per communicator = per thread: no locking needed)
    std::vector<Op> log;
    int groups = 0, ops_outside_group = 0, bad_dtype = 0, bad_stream = 0;
};

thread_local int g_depth = 0;
thread_local std::vector<std::pair<Comm *, Op>> g_pending;
void *const kStreamTag = reinterpret_cast<void *>(0x5151);

void execute(Comm *c, const Op &op) {
    World &w = *c->w;
    if (op.send) {
        std::vector<float> msg(op.count);
        memcpy(msg.data(), op.sbuf, op.count * sizeof(float));
        std::lock_guard<std::mutex> lk(w.mu);
        w.box[{c->rank, op.peer}].push_back(std::move(msg));
        w.cv.notify_all();
    } else {
        std::unique_lock<std::mutex> lk(w.mu);
        auto &q = w.box[{op.peer, c->rank}];
        w.cv.wait(lk, [&] { return !q.empty(); });
        CHECK(q.front().size() == op.count);          // sender and receiver agree on every message size
        memcpy(op.rbuf, q.front().data(), op.count * sizeof(float));
        q.pop_front();
    }
}

}  // namespace

// ---- the mock RCCL the library finds through dlsym(RTLD_DEFAULT) (this executable is linked with -rdynamic) ----------
extern "C" {
int ncclGroupStart() {
    ++g_depth;
    return 0;
}
int ncclGroupEnd() {
    if (--g_depth > 0) return 0;
    for (auto &po : g_pending)
        if (po.second.send) execute(po.first, po.second);   // all sends are buffered first: no ordering between peers
    for (auto &po : g_pending)
        if (!po.second.send) execute(po.first, po.second);
    if (!g_pending.empty()) g_pending.front().first->groups++;
    g_pending.clear();
    return 0;
}
static int post(Comm *c, Op op, int dtype, void *stream) {
    if (dtype != 7) c->bad_dtype++;
    if (stream != kStreamTag) c->bad_stream++;
    if (op.send && op.peer == c->fail_on_send_to) return 5;   // "ncclInvalidUsage"-like
    c->log.push_back(op);
    if (g_depth == 0) {
        c->ops_outside_group++;
        execute(c, op);
    } else {
        g_pending.push_back({c, op});
    }
    return 0;
}
int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, void *stream) {
    return post(static_cast<Comm *>(comm), Op{true, buf, nullptr, count, peer}, dtype, stream);
}
int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, void *stream) {
    return post(static_cast<Comm *>(comm), Op{false, nullptr, buf, count, peer}, dtype, stream);
}
const char *ncclGetErrorString(int rc) { return rc == 5 ? "mock failure injected" : "mock: unknown"; }
}

typedef int (*allgather_t)(void *, int, int, const int64_t *, float *, int64_t, void *);
typedef int (*exchange_t)(void *, int, int, const float *, const int64_t *, float *, const int64_t *, int64_t, void *);
typedef const char *(*cstr_t)(void);

static float cell(int64_t row, int64_t col, int salt) { return (float)((row * 131 + col * 7 + salt) % 100003) * 0.25f; }

int main(int argc, char **argv) {
    CHECK(argc >= 2);
    void *h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        fprintf(stderr, "dlopen failed: %s\n", dlerror());
        return 3;
    }
    auto allgather = (allgather_t)dlsym(h, "sgl_allgather_rows");
    auto exchange = (exchange_t)dlsym(h, "sgl_exchange_rows");
    auto last_error = (cstr_t)dlsym(h, "sgl_last_error");
    auto backend = (cstr_t)dlsym(h, "sgl_exchange_backend");
    CHECK(allgather && exchange && last_error && backend);

    std::mt19937_64 rng(11);
    int cases = 0;
    for (int world = 2; world <= 8; ++world) {
        for (int variant = 0; variant < 3; ++variant) {
            // ---- row bounds: balanced / ragged with empty blocks / everything on one rank ----
            const int64_t n = 37 + 11 * world, ld = 5 + variant;
            std::vector<int64_t> bounds(world + 1, 0);
            if (variant == 0) {
                for (int q = 0; q <= world; ++q) bounds[q] = n * q / world;
            } else if (variant == 1) {
                std::vector<int64_t> cuts;
                for (int q = 1; q < world; ++q) cuts.push_back((int64_t)(rng() % (uint64_t)(n + 1)));
                std::sort(cuts.begin(), cuts.end());
                if (world > 2) cuts[1] = cuts[0];                 // at least one empty block
                for (int q = 1; q < world; ++q) bounds[q] = cuts[q - 1];
                bounds[world] = n;
            } else {
                for (int q = 1; q <= world; ++q) bounds[q] = n;   // rank 0 owns all rows
            }
            // ================= sgl_allgather_rows =================
            {
                World w;
                std::vector<Comm> comms(world);
                std::vector<std::vector<float>> x(world, std::vector<float>(n * ld, -1.0f));
                for (int r = 0; r < world; ++r) {
                    comms[r] = Comm{r, world, &w};
                    for (int64_t i = bounds[r]; i < bounds[r + 1]; ++i)
                        for (int64_t k = 0; k < ld; ++k) x[r][i * ld + k] = cell(i, k, 1);
                }
                std::vector<int> rcs(world, -1);
                std::vector<std::thread> th;
                for (int r = 0; r < world; ++r)
                    th.emplace_back([&, r] { rcs[r] = allgather(&comms[r], r, world, bounds.data(), x[r].data(), ld, kStreamTag); });
                for (auto &t : th) t.join();
                for (int r = 0; r < world; ++r) {
                    CHECK(rcs[r] == 0);
                    for (int64_t i = 0; i < n; ++i)
                        for (int64_t k = 0; k < ld; ++k) CHECK(x[r][i * ld + k] == cell(i, k, 1));
                    const Comm &c = comms[r];
                    CHECK(c.bad_dtype == 0 && c.bad_stream == 0 && c.ops_outside_group == 0);
                    int64_t mine = bounds[r + 1] - bounds[r];
                    size_t want_sends = mine ? (size_t)(world - 1) : 0, want_recvs = 0;
                    for (int q = 0; q < world; ++q) want_recvs += (q != r && bounds[q + 1] > bounds[q]);
                    size_t sends = 0, recvs = 0;
                    for (const Op &op : c.log) {
                        CHECK(op.count > 0 && op.peer != r && op.peer >= 0 && op.peer < world);
                        if (op.send) {
                            ++sends;
                            CHECK(op.count == (size_t)(mine * ld) && op.sbuf == x[r].data() + bounds[r] * ld);
                        } else {
                            ++recvs;
                            CHECK(op.count == (size_t)((bounds[op.peer + 1] - bounds[op.peer]) * ld));
                            CHECK(op.rbuf == x[r].data() + bounds[op.peer] * ld);      // lands in place
                        }
                    }
                    CHECK(sends == want_sends && recvs == want_recvs && c.groups == ((sends + recvs) ? 1 : 0));
                }
                for (auto &kv : w.box) CHECK(kv.second.empty());     // nothing sent that nobody received
                ++cases;
            }
            // ================= sgl_exchange_rows (need-aware, packed) =================
            {
                World w;
                std::vector<Comm> comms(world);
                // need[r][q]: sorted rows of q's block that r gathers (random subsets; some empty)
                std::vector<std::vector<std::vector<int64_t>>> need(world, std::vector<std::vector<int64_t>>(world));
                for (int r = 0; r < world; ++r)
                    for (int q = 0; q < world; ++q) {
                        if (q == r) continue;
                        const uint64_t keep = rng() % 4;              // 0: nothing, else 1/keep of the rows
                        for (int64_t i = bounds[q]; i < bounds[q + 1]; ++i)
                            if (keep && rng() % keep == 0) need[r][q].push_back(i);
                    }
                std::vector<std::vector<int64_t>> send_off(world, std::vector<int64_t>(world + 1, 0)),
                    recv_off(world, std::vector<int64_t>(world + 1, 0));
                std::vector<std::vector<float>> sendbuf(world), table(world);
                for (int r = 0; r < world; ++r) {
                    comms[r] = Comm{r, world, &w};
                    const int64_t n_own = bounds[r + 1] - bounds[r];
                    recv_off[r][0] = n_own;                              // ghosts follow the own rows in the compact table
                    for (int q = 0; q < world; ++q) {
                        send_off[r][q + 1] = send_off[r][q] + (int64_t)need[q][r].size();
                        recv_off[r][q + 1] = recv_off[r][q] + (int64_t)need[r][q].size();
                    }
                    sendbuf[r].assign((size_t)(send_off[r][world] * ld) + 1, -2.0f);
                    table[r].assign((size_t)(recv_off[r][world] * ld) + 1, -3.0f);
                    for (int q = 0; q < world; ++q)                      // the pack step (sgl_gather_rows_f32 on the GPU)
                        for (size_t j = 0; j < need[q][r].size(); ++j)
                            for (int64_t k = 0; k < ld; ++k)
                                sendbuf[r][(send_off[r][q] + (int64_t)j) * ld + k] = cell(need[q][r][j], k, 2);
                }
                std::vector<int> rcs(world, -1);
                std::vector<std::thread> th;
                for (int r = 0; r < world; ++r)
                    th.emplace_back([&, r] {
                        rcs[r] = exchange(&comms[r], r, world, sendbuf[r].data(), send_off[r].data(), table[r].data(),
                                          recv_off[r].data(), ld, kStreamTag);
                    });
                for (auto &t : th) t.join();
                for (int r = 0; r < world; ++r) {
                    CHECK(rcs[r] == 0);
                    const int64_t n_own = bounds[r + 1] - bounds[r];
                    for (int64_t i = 0; i < n_own * ld; ++i) CHECK(table[r][i] == -3.0f);       // own rows untouched
                    for (int q = 0; q < world; ++q)
                        for (size_t j = 0; j < need[r][q].size(); ++j)
                            for (int64_t k = 0; k < ld; ++k)
                                CHECK(table[r][(recv_off[r][q] + (int64_t)j) * ld + k] == cell(need[r][q][j], k, 2));
                    const Comm &c = comms[r];
                    CHECK(c.bad_dtype == 0 && c.bad_stream == 0 && c.ops_outside_group == 0);
                    size_t sends = 0, recvs = 0, want_s = 0, want_r = 0;
                    for (int q = 0; q < world; ++q) {
                        want_s += !need[q][r].empty();
                        want_r += !need[r][q].empty();
                    }
                    for (const Op &op : c.log) {
                        CHECK(op.count > 0 && op.peer != r);
                        if (op.send) {
                            ++sends;
                            CHECK(op.count == need[op.peer][r].size() * (size_t)ld);
                        } else {
                            ++recvs;
                            CHECK(op.count == need[r][op.peer].size() * (size_t)ld);
                        }
                    }
                    CHECK(sends == want_s && recvs == want_r);
                }
                for (auto &kv : w.box) CHECK(kv.second.empty());
                ++cases;
            }
        }
    }
    // ---- argument contract and an RCCL failure: error code + the mock's message, no abort ----
    {
        World w;
        Comm c{0, 2, &w};
        c.fail_on_send_to = 1;
        int64_t b[3] = {0, 4, 9};
        std::vector<float> x(9 * 3, 0.f);
        int rc = allgather(&c, 0, 2, b, x.data(), 3, kStreamTag);
        CHECK(rc != 0 && strstr(last_error(), "mock failure injected"));
        int64_t so[3] = {0, 0, 2}, ro[3] = {4, 4, 4};
        rc = exchange(&c, 0, 2, x.data(), so, x.data(), ro, 3, kStreamTag);
        CHECK(rc != 0 && strstr(last_error(), "mock failure injected"));
        int64_t bad[3] = {0, 2, 1};
        CHECK(exchange(&c, 0, 2, x.data(), bad, x.data(), ro, 3, kStreamTag) != 0 && strstr(last_error(), "offsets"));
        int64_t self[3] = {0, 1, 2};
        CHECK(exchange(&c, 0, 2, x.data(), self, x.data(), ro, 3, kStreamTag) != 0 && strstr(last_error(), "itself"));
        CHECK(std::string(backend()) == "process");      // resolved from the host process, not from a librccl.so on disk
    }
    // ---- sgl_exchange_selftest: the loop-back pairs go through the same post_group() as the exchanges (one group, float32, the
    // caller's stream, peer = the rank itself), on a one-rank world and as rank 3 of a larger one; overlap and bad counts refused ----
    {
        typedef int (*selftest_t)(void *, int, const float *, float *, int64_t, int, void *);
        auto selftest = (selftest_t)dlsym(h, "sgl_exchange_selftest");
        CHECK(selftest);
        for (int rank : {0, 3}) {
            World w;
            Comm c{rank, rank + 1, &w};
            std::vector<float> src(1000), dst(1000, -1.f);
            for (int i = 0; i < 1000; ++i) src[i] = cell(i, 3, rank);
            CHECK(selftest(&c, rank, src.data(), dst.data(), 1000, 3, kStreamTag) == 0);
            CHECK(src == dst && c.groups == 1 && c.ops_outside_group == 0 && c.bad_dtype == 0 && c.bad_stream == 0 && c.log.size() == 6);
            size_t moved = 0;
            for (auto &op : c.log) {
                CHECK(op.peer == rank && op.count > 0);
                if (op.send) moved += op.count;
            }
            CHECK(moved == 1000);
            for (auto &kv : w.box) CHECK(kv.second.empty());
            CHECK(selftest(&c, rank, src.data(), src.data() + 10, 100, 1, kStreamTag) != 0 && strstr(last_error(), "overlap"));
            CHECK(selftest(&c, rank, src.data(), dst.data(), 10, 0, kStreamTag) != 0);
            c.fail_on_send_to = rank;
            CHECK(selftest(&c, rank, src.data(), dst.data(), 10, 1, kStreamTag) != 0 && strstr(last_error(), "mock failure injected"));
        }
    }
    printf("exchange_mock: OK (%d multi-rank cases, worlds 2..8)\n", cases);
    return 0;
}
