// TEST INFRASTRUCTURE (oracle/ref_shim): mimalloc is absent from this image; the two calls
// aligator/core/mimalloc-resource.cpp makes, on the C library's aligned allocator.
#pragma once
#include <cstddef>
#include <cstdlib>
static inline void *mi_malloc_aligned(std::size_t bytes, std::size_t alignment) {
  if (alignment < sizeof(void *))
    alignment = sizeof(void *);
  return std::aligned_alloc(alignment, (bytes + alignment - 1) / alignment * alignment);
}
static inline void mi_free_aligned(void *p, std::size_t) { std::free(p); }
