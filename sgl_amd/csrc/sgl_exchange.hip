// The exchange step of the row-sharded layout behind the C ABI (SURVEY.md section 8(b) export set, 8(e)): between hops
// every rank owns a contiguous block of rows of the next feature replica and needs everybody else's.  RCCL is used through
// the CALLER's communicator and is resolved at run time from whatever RCCL the host process has loaded (dlsym), so the
// library neither links a second copy nor constrains the host's choice; a host without RCCL simply gets an error here.
//
// Direct all-gather(v): one grouped batch of ncclSend / ncclRecv to / from every peer -- xGMI is a point-to-point mesh, so
// all seven links carry traffic at once and unequal row blocks need no padding (a ring all-gather would be bound by one
// link).  Received rows land in place in the replica.  Stream-ordered: nothing synchronises the host.
#include <dlfcn.h>

#include <mutex>
#include <vector>

#include "sgl_common.h"

namespace {

typedef int (*nccl_send_t)(const void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_recv_t)(void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_group_t)(void);
typedef const char *(*nccl_errstr_t)(int);

struct Rccl {
    nccl_send_t send = nullptr;
    nccl_recv_t recv = nullptr;
    nccl_group_t group_start = nullptr, group_end = nullptr;
    nccl_errstr_t errstr = nullptr;
    bool tried = false;
    const char *origin = "";
};

void resolve_rccl(Rccl &r) {
    r.tried = true;
    void *h = RTLD_DEFAULT;                       // the host's own RCCL (an application linked against it, LD_PRELOAD, ...)
    r.origin = "process";
    if (!dlsym(h, "ncclSend")) {
        h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        r.origin = "librccl.so";
        if (!h) return;
    }
    r.send = (nccl_send_t)dlsym(h, "ncclSend");
    r.recv = (nccl_recv_t)dlsym(h, "ncclRecv");
    r.group_start = (nccl_group_t)dlsym(h, "ncclGroupStart");
    r.group_end = (nccl_group_t)dlsym(h, "ncclGroupEnd");
    r.errstr = (nccl_errstr_t)dlsym(h, "ncclGetErrorString");
}

Rccl &rccl() {   // resolved once, also when several host threads (one per GPU) make their first call together
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { resolve_rccl(r); });
    return r;
}

constexpr int kNcclFloat32 = 7;   // ncclFloat32 (rccl.h: ncclDataType_t)

// One transfer pair of a grouped batch: `n_send` floats to rank `dst`, `n_recv` floats from rank `src` (either may be 0).
struct Xfer {
    const float *send;
    size_t n_send;
    int dst;
    float *recv;
    size_t n_recv;
    int src;
};

// THE place where RCCL is called: ncclGroupStart, the sends / receives in the order given, ncclGroupEnd -- through the
// run-time resolved table, with kNcclFloat32, on the caller's communicator and stream.  sgl_allgather_rows, sgl_exchange_rows
// and sgl_exchange_selftest all end here, so a self-test that passes has exercised the exact calls of the multi-GPU exchange.
int post_group(void *comm, hipStream_t st, const Xfer *x, int n, const char *who) {
    Rccl &r = rccl();
    if (!r.send || !r.recv || !r.group_start || !r.group_end)
        return sgl::fail(SGL_ERR_UNSUPPORTED, "%s: no RCCL in this process (ncclSend / ncclRecv not found)", who);
    auto fail = [&](int rc, const char *what) {
        return sgl::fail(rc, "%s: %s failed: %s", who, what, r.errstr ? r.errstr(rc) : "RCCL error");
    };
    int rc = r.group_start();
    if (rc != 0) return fail(rc, "ncclGroupStart");
    for (int i = 0; i < n && rc == 0; ++i) {
        if (x[i].n_send) rc = r.send(x[i].send, x[i].n_send, kNcclFloat32, x[i].dst, comm, st);
        if (rc == 0 && x[i].n_recv) rc = r.recv(x[i].recv, x[i].n_recv, kNcclFloat32, x[i].src, comm, st);
    }
    const int rc_end = r.group_end();
    if (rc != 0) return fail(rc, "ncclSend / ncclRecv");
    if (rc_end != 0) return fail(rc_end, "ncclGroupEnd");
    return SGL_OK;
}

}  // namespace

SGL_EXPORT int sgl_allgather_rows(void *nccl_comm, int rank, int world, const int64_t *h_bounds, float *d_x, int64_t ldx,
                                  void *stream) {
    SGL_REQUIRE(world >= 1 && rank >= 0 && rank < world && h_bounds, "sgl_allgather_rows: bad rank / world / bounds");
    SGL_REQUIRE(ldx >= 0, "sgl_allgather_rows: bad leading dimension");
    for (int q = 0; q < world; ++q)
        SGL_REQUIRE(h_bounds[q] <= h_bounds[q + 1] && h_bounds[0] >= 0, "sgl_allgather_rows: row bounds must not decrease");
    if (world == 1 || ldx == 0) return SGL_OK;
    SGL_REQUIRE(d_x != nullptr && nccl_comm != nullptr, "sgl_allgather_rows: NULL matrix or communicator");
    const size_t mine = (size_t)(h_bounds[rank + 1] - h_bounds[rank]) * (size_t)ldx;
    std::vector<Xfer> xs;
    xs.reserve((size_t)world);
    // stagger the peer order per rank so that at any moment every link carries one transfer in each direction
    for (int k = 1; k < world; ++k) {
        const int dst = (rank + k) % world, src = (rank - k + world) % world;
        const size_t theirs = (size_t)(h_bounds[src + 1] - h_bounds[src]) * (size_t)ldx;
        xs.push_back(Xfer{d_x + h_bounds[rank] * ldx, mine, dst, d_x + h_bounds[src] * ldx, theirs, src});
    }
    return post_group(nccl_comm, sgl::as_stream(stream), xs.data(), (int)xs.size(), "sgl_allgather_rows");
}

// Need-aware form of the same exchange (sgl_amd/dist/halo.py is the plan that produces the offsets): the rows of this rank
// that peer q gathers were packed into d_send (sgl_gather_rows_f32 with the rank's send list), q's share at rows
// [h_send_off[q], h_send_off[q+1]); the rows of peer q this rank gathers land packed at rows [h_recv_off[q], h_recv_off[q+1])
// of d_recv -- the ghost range of the rank's compact table.  One grouped batch, every link busy in both directions.
SGL_EXPORT int sgl_exchange_rows(void *nccl_comm, int rank, int world, const float *d_send, const int64_t *h_send_off,
                                 float *d_recv, const int64_t *h_recv_off, int64_t ld, void *stream) {
    SGL_REQUIRE(world >= 1 && rank >= 0 && rank < world && h_send_off && h_recv_off, "sgl_exchange_rows: bad rank / world / offsets");
    SGL_REQUIRE(ld >= 0, "sgl_exchange_rows: bad leading dimension");
    for (int q = 0; q < world; ++q)
        SGL_REQUIRE(h_send_off[q] <= h_send_off[q + 1] && h_recv_off[q] <= h_recv_off[q + 1] && h_send_off[0] >= 0 && h_recv_off[0] >= 0,
                    "sgl_exchange_rows: offsets must not decrease");
    SGL_REQUIRE(h_send_off[rank] == h_send_off[rank + 1] && h_recv_off[rank] == h_recv_off[rank + 1],
                "sgl_exchange_rows: a rank exchanges nothing with itself");
    if (world == 1 || ld == 0) return SGL_OK;
    const bool any_send = h_send_off[world] > h_send_off[0], any_recv = h_recv_off[world] > h_recv_off[0];
    SGL_REQUIRE(nccl_comm != nullptr && (!any_send || d_send) && (!any_recv || d_recv), "sgl_exchange_rows: NULL buffer or communicator");
    std::vector<Xfer> xs;
    xs.reserve((size_t)world);
    for (int k = 1; k < world; ++k) {
        const int dst = (rank + k) % world, src = (rank - k + world) % world;
        const size_t out = (size_t)(h_send_off[dst + 1] - h_send_off[dst]) * (size_t)ld;
        const size_t in = (size_t)(h_recv_off[src + 1] - h_recv_off[src]) * (size_t)ld;
        xs.push_back(Xfer{d_send + h_send_off[dst] * ld, out, dst, d_recv + h_recv_off[src] * ld, in, src});
    }
    return post_group(nccl_comm, sgl::as_stream(stream), xs.data(), (int)xs.size(), "sgl_exchange_rows");
}

// Loop-back check of the RCCL binding on the CALLER's communicator: `n` floats travel from d_src to d_dst as a grouped
// ncclSend / ncclRecv pair whose peer is this very rank -- the same post_group(), function table and data-type constant the two
// exchanges above use, so it proves on ONE GPU (or on every rank of a real job, before the first hop) that the library found a
// working RCCL, that its idea of ncclFloat32 and of the argument order matches that RCCL's, and that the communicator and the
// stream are accepted.  `n_ops` pairs are posted in the one group (the shape of a world of n_ops + 1 ranks), each moving its
// share of the n floats.  d_src and d_dst must not overlap.  Stream-ordered.
SGL_EXPORT int sgl_exchange_selftest(void *nccl_comm, int rank, const float *d_src, float *d_dst, int64_t n, int n_ops, void *stream) {
    SGL_REQUIRE(nccl_comm != nullptr && rank >= 0, "sgl_exchange_selftest: NULL communicator or negative rank");
    SGL_REQUIRE(n >= 0 && n_ops >= 1 && n_ops <= 64, "sgl_exchange_selftest: n >= 0 and 1 <= n_ops <= 64");
    SGL_REQUIRE(n == 0 || (d_src && d_dst), "sgl_exchange_selftest: NULL buffer");
    SGL_REQUIRE(n == 0 || d_src + n <= d_dst || d_dst + n <= d_src, "sgl_exchange_selftest: source and destination overlap");
    if (n == 0) return SGL_OK;
    std::vector<Xfer> xs;
    for (int i = 0; i < n_ops; ++i) {
        const int64_t a = n * i / n_ops, b = n * (i + 1) / n_ops;
        xs.push_back(Xfer{d_src + a, (size_t)(b - a), rank, d_dst + a, (size_t)(b - a), rank});
    }
    return post_group(nccl_comm, sgl::as_stream(stream), xs.data(), (int)xs.size(), "sgl_exchange_selftest");
}

// which RCCL the exchange resolved to: "process" (symbols the host already had), "librccl.so" (loaded here) or "" (none)
SGL_EXPORT const char *sgl_exchange_backend(void) {
    Rccl &r = rccl();
    return (r.send && r.recv) ? r.origin : "";
}
