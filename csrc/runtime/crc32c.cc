// Software CRC-32C, slice-by-8 (polynomial 0x1EDC6F41, reflected 0x82F63B78).
#include "crc32c.h"

#include <cstring>

namespace dtf {
namespace {

struct Tables {
    uint32_t t[8][256];
    Tables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xff];
    }
};

const Tables& tables() {
    static const Tables tb;
    return tb;
}

}  // namespace

uint32_t crc32c_extend(uint32_t crc, const void* data, size_t n) {
    const Tables& tb = tables();
    const uint8_t* p = static_cast<const uint8_t*>(data);
    uint32_t c = crc ^ 0xffffffffu;
    while (n && (reinterpret_cast<uintptr_t>(p) & 7)) {
        c = tb.t[0][(c ^ *p++) & 0xff] ^ (c >> 8);
        --n;
    }
    while (n >= 8) {
        uint64_t w;
        std::memcpy(&w, p, 8);
        w ^= c;
        c = tb.t[7][w & 0xff] ^ tb.t[6][(w >> 8) & 0xff] ^ tb.t[5][(w >> 16) & 0xff] ^ tb.t[4][(w >> 24) & 0xff] ^
            tb.t[3][(w >> 32) & 0xff] ^ tb.t[2][(w >> 40) & 0xff] ^ tb.t[1][(w >> 48) & 0xff] ^ tb.t[0][w >> 56];
        p += 8;
        n -= 8;
    }
    while (n--) c = tb.t[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return c ^ 0xffffffffu;
}

}  // namespace dtf

DTF_EXPORT uint32_t dtf_crc32c(const void* data, size_t n) { return dtf::crc32c(data, n); }
DTF_EXPORT uint32_t dtf_crc32c_extend(uint32_t crc, const void* data, size_t n) { return dtf::crc32c_extend(crc, data, n); }
DTF_EXPORT uint32_t dtf_crc32c_masked(const void* data, size_t n) { return dtf::crc32c_mask(dtf::crc32c(data, n)); }
