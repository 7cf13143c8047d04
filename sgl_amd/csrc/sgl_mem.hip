// Device allocations with a stated physical placement, for the tables the SpMM gathers from.
//
// The k-step A_hat * X of the papers100M-sized jobs gathers 512-byte rows at random from a 57 GB replica: every gather
// is an address translation the L1 / L2 translation caches (UTCL1 / UTCL2) cannot hold at the 2 MB granularity an ordinary
// hipMalloc of that size ends up with (the driver writes a larger PTE "fragment" only where virtual AND physical ranges are
// contiguous and equally aligned, which depends on what the VRAM allocator had free at that moment: the +-3 % run-to-run
// spread of that regime, profiles/r02_placement.log).  These entry points let the caller ask for the placement:
//   SGL_MEM_DEFAULT      hipMalloc
//   SGL_MEM_CONTIGUOUS   hipExtMallocWithFlags(hipDeviceMallocContiguous): one physically contiguous range
//   SGL_MEM_VMM          hipMemCreate / hipMemAddressReserve / hipMemMap: physical chunks of `chunk_bytes` (0 = one chunk),
//                        each mapped at a virtual address aligned to its own size
// sgl_mem_free takes the pointer back whatever mode produced it.
#include <mutex>
#include <unordered_map>

#include "sgl_common.h"

#include "../../include/sgl_probe.h"

namespace {

struct Block {
    int mode;
    size_t bytes;                                     // reserved / allocated size
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<size_t> sizes;
};

std::mutex g_mu;
std::unordered_map<void *, Block> g_blocks;

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

SGL_EXPORT int sgl_mem_alloc(void **out, int64_t bytes, int mode, int64_t chunk_bytes) {
    SGL_REQUIRE(out && bytes > 0, "sgl_mem_alloc: bad arguments");
    *out = nullptr;
    Block b;
    b.mode = mode;
    void *p = nullptr;
    if (mode == SGL_MEM_DEFAULT) {
        SGL_HIP_CHECK(hipMalloc(&p, (size_t)bytes));
        b.bytes = (size_t)bytes;
    } else if (mode == SGL_MEM_CONTIGUOUS) {
        SGL_HIP_CHECK(hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocContiguous));
        b.bytes = (size_t)bytes;
    } else if (mode == SGL_MEM_VMM) {
        int dev = 0;
        SGL_HIP_CHECK(hipGetDevice(&dev));
        hipMemAllocationProp prop;
        memset(&prop, 0, sizeof(prop));
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = dev;
        size_t gran = 0;
        SGL_HIP_CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        if (gran == 0) gran = (size_t)2 << 20;
        const size_t chunk = chunk_bytes > 0 ? round_up((size_t)chunk_bytes, gran) : round_up((size_t)bytes, gran);
        const size_t total = round_up((size_t)bytes, chunk);
        // the virtual range is aligned to the chunk size (capped: an alignment of tens of GB is neither needed nor granted)
        const size_t va_align = std::min(chunk, (size_t)1 << 30);
        SGL_HIP_CHECK(hipMemAddressReserve(&p, total, va_align, nullptr, 0));
        hipMemAccessDesc acc;
        memset(&acc, 0, sizeof(acc));
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = dev;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        hipError_t e = hipSuccess;
        for (size_t off = 0; off < total && e == hipSuccess; off += chunk) {
            hipMemGenericAllocationHandle_t h;
            e = hipMemCreate(&h, chunk, &prop, 0);
            if (e != hipSuccess) break;
            b.handles.push_back(h);
            b.sizes.push_back(chunk);
            e = hipMemMap(static_cast<char *>(p) + off, chunk, 0, h, 0);
        }
        if (e == hipSuccess) e = hipMemSetAccess(p, total, &acc, 1);
        if (e != hipSuccess) {
            size_t off = 0;
            for (size_t i = 0; i < b.handles.size(); ++i) {
                (void)hipMemUnmap(static_cast<char *>(p) + off, b.sizes[i]);
                (void)hipMemRelease(b.handles[i]);
                off += b.sizes[i];
            }
            (void)hipMemAddressFree(p, total);
            return sgl::fail((int)e, "sgl_mem_alloc: virtual-memory path failed: %s", hipGetErrorString(e));
        }
        b.bytes = total;
    } else {
        return sgl::fail(SGL_ERR_INVALID, "sgl_mem_alloc: unknown mode %d", mode);
    }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_blocks.emplace(p, std::move(b));
    }
    *out = p;
    return SGL_OK;
}

SGL_EXPORT int sgl_mem_free(void *d_ptr) {
    if (!d_ptr) return SGL_OK;
    Block b;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_blocks.find(d_ptr);
        if (it == g_blocks.end()) return sgl::fail(SGL_ERR_INVALID, "sgl_mem_free: pointer was not returned by sgl_mem_alloc");
        b = std::move(it->second);
        g_blocks.erase(it);
    }
    if (b.mode != SGL_MEM_VMM) {
        SGL_HIP_CHECK(hipFree(d_ptr));
        return SGL_OK;
    }
    SGL_HIP_CHECK(hipDeviceSynchronize());
    size_t off = 0;
    for (size_t i = 0; i < b.handles.size(); ++i) {
        SGL_HIP_CHECK(hipMemUnmap(static_cast<char *>(d_ptr) + off, b.sizes[i]));
        SGL_HIP_CHECK(hipMemRelease(b.handles[i]));
        off += b.sizes[i];
    }
    SGL_HIP_CHECK(hipMemAddressFree(d_ptr, b.bytes));
    return SGL_OK;
}
