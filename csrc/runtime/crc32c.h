// CRC-32C (Castagnoli) and the "masked" form TensorFlow stores in TFRecord frames,
// table blocks and TensorBundle entries.
#pragma once
#include <cstddef>
#include <cstdint>

namespace dtf {

uint32_t crc32c_extend(uint32_t crc, const void* data, size_t n);

inline uint32_t crc32c(const void* data, size_t n) { return crc32c_extend(0, data, n); }

// mask = rotate right by 15, add a constant (so a CRC of data containing CRCs stays well distributed)
inline uint32_t crc32c_mask(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + 0xa282ead8u; }

inline uint32_t crc32c_unmask(uint32_t masked) {
    uint32_t rot = masked - 0xa282ead8u;
    return (rot >> 17) | (rot << 15);
}

}  // namespace dtf

#define DTF_EXPORT extern "C" __attribute__((visibility("default")))
