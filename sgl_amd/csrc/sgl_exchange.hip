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

}  // namespace

SGL_EXPORT int sgl_allgather_rows(void *nccl_comm, int rank, int world, const int64_t *h_bounds, float *d_x, int64_t ldx,
                                  void *stream) {
    SGL_REQUIRE(world >= 1 && rank >= 0 && rank < world && h_bounds, "sgl_allgather_rows: bad rank / world / bounds");
    SGL_REQUIRE(ldx >= 0, "sgl_allgather_rows: bad leading dimension");
    for (int q = 0; q < world; ++q)
        SGL_REQUIRE(h_bounds[q] <= h_bounds[q + 1] && h_bounds[0] >= 0, "sgl_allgather_rows: row bounds must not decrease");
    if (world == 1 || ldx == 0) return SGL_OK;
    SGL_REQUIRE(d_x != nullptr && nccl_comm != nullptr, "sgl_allgather_rows: NULL matrix or communicator");
    Rccl &r = rccl();
    if (!r.send || !r.recv || !r.group_start || !r.group_end)
        return sgl::fail(SGL_ERR_UNSUPPORTED, "sgl_allgather_rows: no RCCL in this process (ncclSend / ncclRecv not found)");
    hipStream_t st = sgl::as_stream(stream);
    auto fail = [&](int rc, const char *what) {
        return sgl::fail(rc, "sgl_allgather_rows: %s failed: %s", what, r.errstr ? r.errstr(rc) : "RCCL error");
    };
    const size_t mine = (size_t)(h_bounds[rank + 1] - h_bounds[rank]) * (size_t)ldx;
    int rc = r.group_start();
    if (rc != 0) return fail(rc, "ncclGroupStart");
    // stagger the peer order per rank so that at any moment every link carries one transfer in each direction
    for (int k = 1; k < world && rc == 0; ++k) {
        const int dst = (rank + k) % world, src = (rank - k + world) % world;
        if (mine) rc = r.send(d_x + h_bounds[rank] * ldx, mine, kNcclFloat32, dst, nccl_comm, st);
        const size_t theirs = (size_t)(h_bounds[src + 1] - h_bounds[src]) * (size_t)ldx;
        if (rc == 0 && theirs) rc = r.recv(d_x + h_bounds[src] * ldx, theirs, kNcclFloat32, src, nccl_comm, st);
    }
    const int rc_end = r.group_end();
    if (rc != 0) return fail(rc, "ncclSend / ncclRecv");
    if (rc_end != 0) return fail(rc_end, "ncclGroupEnd");
    return SGL_OK;
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
    Rccl &r = rccl();
    if (!r.send || !r.recv || !r.group_start || !r.group_end)
        return sgl::fail(SGL_ERR_UNSUPPORTED, "sgl_exchange_rows: no RCCL in this process (ncclSend / ncclRecv not found)");
    hipStream_t st = sgl::as_stream(stream);
    auto fail = [&](int rc, const char *what) {
        return sgl::fail(rc, "sgl_exchange_rows: %s failed: %s", what, r.errstr ? r.errstr(rc) : "RCCL error");
    };
    int rc = r.group_start();
    if (rc != 0) return fail(rc, "ncclGroupStart");
    for (int k = 1; k < world && rc == 0; ++k) {
        const int dst = (rank + k) % world, src = (rank - k + world) % world;
        const size_t out = (size_t)(h_send_off[dst + 1] - h_send_off[dst]) * (size_t)ld;
        const size_t in = (size_t)(h_recv_off[src + 1] - h_recv_off[src]) * (size_t)ld;
        if (out) rc = r.send(d_send + h_send_off[dst] * ld, out, kNcclFloat32, dst, nccl_comm, st);
        if (rc == 0 && in) rc = r.recv(d_recv + h_recv_off[src] * ld, in, kNcclFloat32, src, nccl_comm, st);
    }
    const int rc_end = r.group_end();
    if (rc != 0) return fail(rc, "ncclSend / ncclRecv");
    if (rc_end != 0) return fail(rc_end, "ncclGroupEnd");
    return SGL_OK;
}

// which RCCL the exchange resolved to: "process" (symbols the host already had), "librccl.so" (loaded here) or "" (none)
SGL_EXPORT const char *sgl_exchange_backend(void) {
    Rccl &r = rccl();
    return (r.send && r.recv) ? r.origin : "";
}
