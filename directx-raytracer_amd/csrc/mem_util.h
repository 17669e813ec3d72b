// memcpy that tolerates the (nullptr, 0) pair an empty std::vector hands out: passing a null pointer to memcpy is undefined
// behaviour even for zero bytes (found by the UBSan run of tools/asan_host.sh).
#pragma once

#include <cstddef>
#include <cstring>

namespace crt {
inline void copyBytes(void* dst, const void* src, std::size_t n)
{
    if (n) std::memcpy(dst, src, n);
}
} // namespace crt
