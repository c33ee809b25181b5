// Device-memory cache of libdesc_amd.so.  hipFree of the large per-call arrays (sampled cycles, weights, packed words: hundreds of
// megabytes to gigabytes) costs 5-15 ms per solve in page-table work, hipMalloc of the same sizes again on the next call; a
// MATLAB / Python session calls DESC_PGD many times on problems of the same shape.  Blocks are therefore parked on release and
// handed out again (best fit, at most 25 % larger than asked) instead of going back to the driver, up to DESC_CACHE_MB
// (default 8192, at most a quarter of the device memory; 0 = off) per process; desc_trim_memory() returns everything parked to the driver
// (the MEX shims register it with mexAtExit).
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "device_utils.h"
#include "hostmem.h"

namespace desc {
namespace {
struct Block { void* p; size_t bytes; int dev; bool uc; };      // uc: uncached device memory (hipDeviceMallocUncached): only ever handed out as such
std::mutex g_mu;
std::vector<Block> g_parked;                         // released blocks waiting for reuse
std::unordered_map<void*, Block> g_live;             // blocks handed out
size_t g_parked_bytes = 0;
// DESC_CACHE_MB if set (0 = park nothing); else 8 GiB but never more than a quarter of the device's memory: the card is shared with
// the other libraries of the process (PyTorch in the sharded driver, other toolboxes in a MATLAB session) and with other ranks' processes.
size_t cache_cap() {
    static size_t cap = [] {
        if (const char* e = std::getenv("DESC_CACHE_MB")) return (size_t)std::atoll(e) << 20;
        size_t fr = 0, tot = 0;
        size_t c = (size_t)8192 << 20;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && tot > 0) c = std::min(c, tot / 4); else (void)hipGetLastError();
        return c;
    }();
    return cap;
}
}  // namespace

static hipError_t dev_alloc_impl(void** out, size_t bytes, bool uc);
hipError_t dev_alloc(void** out, size_t bytes) { return dev_alloc_impl(out, bytes, false); }
// Uncached device memory (MTYPE UC): loads and stores of such a block do not leave their lines in the L2.  For arrays a kernel streams
// through exactly once per launch next to data it gathers from repeatedly (pgd.hip: S0 and the packed words of the band sweep).
hipError_t dev_alloc_uncached(void** out, size_t bytes) { return dev_alloc_impl(out, bytes, true); }
static hipError_t dev_alloc_impl(void** out, size_t bytes, bool uc) {
    *out = nullptr;
    if (bytes == 0) bytes = 8;
    bytes = (bytes + 255) & ~(size_t)255;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        size_t best = (size_t)-1; size_t bi = 0;
        for (size_t i = 0; i < g_parked.size(); ++i) {
            const Block& b = g_parked[i];
            if (b.dev == dev && b.uc == uc && b.bytes >= bytes && b.bytes <= bytes + bytes / 4 + 4096 && b.bytes < best) { best = b.bytes; bi = i; }
        }
        if (best != (size_t)-1) {
            const Block b = g_parked[bi];
            g_parked[bi] = g_parked.back(); g_parked.pop_back();
            g_parked_bytes -= b.bytes;
            g_live[b.p] = b;
            *out = b.p;
            return hipSuccess;
        }
    }
    void* p = nullptr;
    auto raw = [&]() { return uc ? hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) : hipMalloc(&p, bytes); };
    e = raw();
    if (e != hipSuccess) {                            // make room: give the parked blocks back and try once more
        (void)hipGetLastError();
        desc_trim_memory();
        e = raw();
        if (e != hipSuccess && uc) {                  // no uncached memory to be had: an ordinary block does the same job, slower
            (void)hipGetLastError();
            uc = false;
            e = hipMalloc(&p, bytes);
        }
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    g_live[p] = Block{p, bytes, dev, uc};
    *out = p;
    return hipSuccess;
}

void dev_free(void* p) {
    if (!p) return;
    // hipFree waits for the device before it releases a block; a parked block may be handed out again at once, so the same
    // guarantee is kept here (idle device: microseconds) -- on the BLOCK's device, which need not be the thread's current one
    int cur = 0, bdev = -1;
    (void)hipGetDevice(&cur);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_live.find(p);
        if (it != g_live.end()) bdev = it->second.dev;
    }
    if (bdev >= 0 && bdev != cur) (void)hipSetDevice(bdev);
    (void)hipDeviceSynchronize();
    if (bdev >= 0 && bdev != cur) (void)hipSetDevice(cur);
    dev_free_idle(p);
}
void dev_free_idle(void* p) {
    if (!p) return;
    Block b{};
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_live.find(p);
        if (it == g_live.end()) { (void)hipFree(p); return; }         // not ours (should not happen)
        b = it->second;
        g_live.erase(it);
        if (g_parked_bytes + b.bytes <= cache_cap()) {
            g_parked.push_back(b); g_parked_bytes += b.bytes;
            return;
        }
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != b.dev) (void)hipSetDevice(b.dev);
    (void)hipFree(p);
    if (cur != b.dev) (void)hipSetDevice(cur);
}


// ---- streams
namespace {
struct PooledStream { int dev; hipStream_t s; };
std::vector<PooledStream> g_streams;                 // idle non-blocking streams (at most 8 are kept)
}  // namespace
hipError_t stream_acquire(hipStream_t* out) {
    *out = nullptr;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (size_t i = 0; i < g_streams.size(); ++i)
            if (g_streams[i].dev == dev) { *out = g_streams[i].s; g_streams[i] = g_streams.back(); g_streams.pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
void stream_release(hipStream_t s) {
    if (!s) return;
    int dev = hipGetStreamDeviceId(s);                // the stream's own device, not the thread's current one
    if (dev >= 0) {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_streams.size() < 8) { g_streams.push_back(PooledStream{dev, s}); return; }
    } else (void)hipGetLastError();
    (void)hipStreamDestroy(s);
}

// ---- host blocks (hostmem.h)
namespace {
constexpr size_t HOST_POOL_MIN = 256 << 10;
struct HBlock { void* p; size_t bytes; };
std::mutex g_hmu;
std::vector<HBlock> g_hparked;
std::unordered_map<void*, size_t> g_hlive;            // pooled blocks handed out -> their real size
size_t g_hparked_bytes = 0;
size_t host_cache_cap() {
    static size_t cap = [] { const char* e = std::getenv("DESC_HOST_CACHE_MB"); return (size_t)(e ? std::atoll(e) : 1024) << 20; }();
    return cap;
}
}  // namespace

void* host_block_alloc(size_t bytes) {
    if (bytes < HOST_POOL_MIN) return ::operator new(bytes);
    {
        std::lock_guard<std::mutex> lk(g_hmu);
        size_t best = (size_t)-1, bi = 0;
        for (size_t i = 0; i < g_hparked.size(); ++i) {
            const HBlock& b = g_hparked[i];
            if (b.bytes >= bytes && b.bytes <= bytes + bytes / 4 && b.bytes < best) { best = b.bytes; bi = i; }
        }
        if (best != (size_t)-1) {
            const HBlock b = g_hparked[bi];
            g_hparked[bi] = g_hparked.back(); g_hparked.pop_back();
            g_hparked_bytes -= b.bytes;
            g_hlive[b.p] = b.bytes;
            return b.p;
        }
    }
    void* p = ::operator new(bytes);
    std::lock_guard<std::mutex> lk(g_hmu);
    g_hlive[p] = bytes;
    return p;
}
void host_block_free(void* p, size_t bytes) {
    if (!p) return;
    if (bytes < HOST_POOL_MIN) { ::operator delete(p); return; }
    {
        std::lock_guard<std::mutex> lk(g_hmu);
        auto it = g_hlive.find(p);
        if (it != g_hlive.end()) {
            const size_t real = it->second;
            g_hlive.erase(it);
            if (g_hparked_bytes + real <= host_cache_cap()) { g_hparked.push_back(HBlock{p, real}); g_hparked_bytes += real; return; }
        }
    }
    ::operator delete(p);
}
size_t host_block_trim() {
    std::vector<HBlock> blocks;
    {
        std::lock_guard<std::mutex> lk(g_hmu);
        blocks.swap(g_hparked);
        g_hparked_bytes = 0;
    }
    size_t freed = 0;
    for (const HBlock& b : blocks) { ::operator delete(b.p); freed += b.bytes; }
    return freed;
}

}  // namespace desc

extern "C" int64_t desc_trim_memory(void) {
    (void)desc::host_block_trim();
    {
        std::vector<desc::PooledStream> ss;
        { std::lock_guard<std::mutex> lk(desc::g_mu); ss.swap(desc::g_streams); }
        int cur = 0;
        (void)hipGetDevice(&cur);
        for (const desc::PooledStream& q : ss) { (void)hipSetDevice(q.dev); (void)hipStreamDestroy(q.s); }
        (void)hipSetDevice(cur);
    }
    std::vector<desc::Block> blocks;
    {
        std::lock_guard<std::mutex> lk(desc::g_mu);
        blocks.swap(desc::g_parked);
        desc::g_parked_bytes = 0;
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    int64_t freed = 0;
    for (const desc::Block& b : blocks) {
        (void)hipSetDevice(b.dev);
        (void)hipFree(b.p);
        freed += (int64_t)b.bytes;
    }
    (void)hipSetDevice(cur);
    return freed;
}
