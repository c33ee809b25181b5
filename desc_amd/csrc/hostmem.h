// Host-side block cache for the library's large temporary vectors (index arrays of the structure builder and of the solver's
// setup: ~200 MB at C4).  Fresh blocks of that size cost a page fault per 4 KiB on first touch and a munmap on release -- 20+ ms
// per solve; a session calls DESC_PGD many times on problems of the same shape.  Blocks of >= 256 KiB are therefore parked on release
// and handed out again (best fit, at most 25 % larger than asked), up to DESC_HOST_CACHE_MB (default 1024) per process;
// desc_trim_memory() returns them.  Smaller requests go straight to operator new.
#pragma once
#include <cstddef>
#include <new>
#include <vector>

namespace desc {

void* host_block_alloc(size_t bytes);
void host_block_free(void* p, size_t bytes);
size_t host_block_trim();

template <class T>
struct pool_alloc {
    using value_type = T;
    pool_alloc() = default;
    template <class U> pool_alloc(const pool_alloc<U>&) {}
    T* allocate(size_t n) { return (T*)host_block_alloc(n * sizeof(T)); }
    void deallocate(T* p, size_t n) { host_block_free(p, n * sizeof(T)); }
    template <class U> bool operator==(const pool_alloc<U>&) const { return true; }
    template <class U> bool operator!=(const pool_alloc<U>&) const { return false; }
};
template <class T> using hvec = std::vector<T, pool_alloc<T>>;

}  // namespace desc
