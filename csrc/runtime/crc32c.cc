// CRC-32C (polynomial 0x1EDC6F41, reflected 0x82F63B78): the SSE4.2 crc32 instruction, three interleaved streams
// per block so the 3-cycle instruction latency is hidden, when the CPU has it; slice-by-8 tables otherwise.
#include "crc32c.h"

#include <cstring>

#if defined(__x86_64__)
#include <nmmintrin.h>
#define DTF_CRC_HW 1
#endif

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

#ifdef DTF_CRC_HW
// crc of (a || b) from crc(a) with the bytes of b still to come: shifting crc(a) over len zero bytes is a linear map
// over GF(2); `shift_table` holds that map for one fixed block length as 4 x 256 lookups.
struct ShiftTable {
    uint32_t t[4][256];
};

uint32_t gf2_times(const uint32_t* mat, uint32_t vec) {
    uint32_t sum = 0;
    while (vec) {
        if (vec & 1) sum ^= *mat;
        vec >>= 1;
        ++mat;
    }
    return sum;
}

void gf2_square(uint32_t* sq, const uint32_t* mat) {
    for (int n = 0; n < 32; ++n) sq[n] = gf2_times(mat, mat[n]);
}

// operator that advances a (raw, un-inverted) crc register over `len` zero bytes
void zeros_operator(uint32_t* even, size_t len) {
    uint32_t odd[32];
    odd[0] = 0x82F63B78u;
    uint32_t row = 1;
    for (int n = 1; n < 32; ++n) {
        odd[n] = row;
        row <<= 1;
    }
    gf2_square(even, odd);   // 2 zero bits
    gf2_square(odd, even);   // 4 zero bits
    do {                     // first square: 8 bits = one zero byte
        gf2_square(even, odd);
        len >>= 1;
        if (len == 0) return;
        gf2_square(odd, even);
        len >>= 1;
    } while (len);
    for (int n = 0; n < 32; ++n) even[n] = odd[n];
}

constexpr size_t BLOCK = 4096;   // bytes per stream; three streams per round

const ShiftTable& shift_table() {
    static const ShiftTable st = [] {
        ShiftTable s;
        uint32_t op[32];
        zeros_operator(op, BLOCK);
        for (uint32_t n = 0; n < 256; ++n) {
            s.t[0][n] = gf2_times(op, n);
            s.t[1][n] = gf2_times(op, n << 8);
            s.t[2][n] = gf2_times(op, n << 16);
            s.t[3][n] = gf2_times(op, n << 24);
        }
        return s;
    }();
    return st;
}

inline uint32_t shift_block(const ShiftTable& s, uint32_t c) {
    return s.t[0][c & 0xff] ^ s.t[1][(c >> 8) & 0xff] ^ s.t[2][(c >> 16) & 0xff] ^ s.t[3][c >> 24];
}

__attribute__((target("sse4.2"))) uint32_t crc_hw(uint32_t c, const uint8_t* p, size_t n) {
    uint64_t c0 = c;
    while (n && (reinterpret_cast<uintptr_t>(p) & 7)) {
        c0 = _mm_crc32_u8(static_cast<uint32_t>(c0), *p++);
        --n;
    }
    const ShiftTable& st = shift_table();
    while (n >= 3 * BLOCK) {
        uint64_t c1 = 0, c2 = 0;
        const uint8_t* end = p + BLOCK;
        do {
            uint64_t a, b, d;
            std::memcpy(&a, p, 8);
            std::memcpy(&b, p + BLOCK, 8);
            std::memcpy(&d, p + 2 * BLOCK, 8);
            c0 = _mm_crc32_u64(c0, a);
            c1 = _mm_crc32_u64(c1, b);
            c2 = _mm_crc32_u64(c2, d);
            p += 8;
        } while (p < end);
        c0 = shift_block(st, static_cast<uint32_t>(c0)) ^ c1;
        c0 = shift_block(st, static_cast<uint32_t>(c0)) ^ c2;
        p += 2 * BLOCK;
        n -= 3 * BLOCK;
    }
    while (n >= 8) {
        uint64_t a;
        std::memcpy(&a, p, 8);
        c0 = _mm_crc32_u64(c0, a);
        p += 8;
        n -= 8;
    }
    while (n--) c0 = _mm_crc32_u8(static_cast<uint32_t>(c0), *p++);
    return static_cast<uint32_t>(c0);
}

bool have_hw() {
    static const bool ok = __builtin_cpu_supports("sse4.2");
    return ok;
}
#endif

}  // namespace

uint32_t crc32c_extend(uint32_t crc, const void* data, size_t n) {
    const uint8_t* p = static_cast<const uint8_t*>(data);
    uint32_t c = crc ^ 0xffffffffu;
#ifdef DTF_CRC_HW
    if (have_hw()) return crc_hw(c, p, n) ^ 0xffffffffu;
#endif
    const Tables& tb = tables();
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
