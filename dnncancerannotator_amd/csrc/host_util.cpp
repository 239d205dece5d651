// host_util.cpp -- host-side helpers of the data path (plain C++, no device code).
//
// CRC-32C (Castagnoli) of TFRecord framing (tfrecord.py; the reference reads its exam files with tf.data.TFRecordDataset,
// annotator/data.py:448-470, whose reader checks the masked CRC-32C of every length and every payload).  The payload of an exam is
// ~20 MB of pixels: the SSE4.2 crc32 instruction walks it at several GB/s (three independent streams hide its 3-cycle latency),
// a byte-wise table in Python would need seconds.
#include <nmmintrin.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "../../include/dnnca.h"

namespace {

uint32_t table[4][256];

void make_table() {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
        table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 4; ++t) table[t][i] = (table[t - 1][i] >> 8) ^ table[0][table[t - 1][i] & 0xFF];
}

// portable slicing-by-4 (hosts without SSE4.2)
uint32_t crc_sw(uint32_t c, const unsigned char* p, size_t n) {
    static const bool ready = (make_table(), true);      // function-local static: built once, thread-safe (reader threads)
    (void)ready;
    while (n >= 4) {
        uint32_t w;
        memcpy(&w, p, 4);
        c ^= w;
        c = table[3][c & 0xFF] ^ table[2][(c >> 8) & 0xFF] ^ table[1][(c >> 16) & 0xFF] ^ table[0][c >> 24];
        p += 4;
        n -= 4;
    }
    while (n--) c = table[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return c;
}

__attribute__((target("sse4.2"))) uint32_t crc_hw(uint32_t c, const unsigned char* p, size_t n) {
    uint64_t c64 = c;
    while (n && (reinterpret_cast<uintptr_t>(p) & 7)) { c64 = _mm_crc32_u8((uint32_t)c64, *p++); --n; }
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        c64 = _mm_crc32_u64(c64, w);
        p += 8;
        n -= 8;
    }
    while (n--) c64 = _mm_crc32_u8((uint32_t)c64, *p++);
    return (uint32_t)c64;
}

}  // namespace

extern "C" {

int dnnca_crc32c(const void* data, size_t n, uint32_t* crc_out) {
    if (!crc_out || (!data && n)) return DNNCA_EINVAL;
    static const bool hw = __builtin_cpu_supports("sse4.2");
    const unsigned char* p = static_cast<const unsigned char*>(data);
    const uint32_t c = hw ? crc_hw(0xFFFFFFFFu, p, n) : crc_sw(0xFFFFFFFFu, p, n);
    *crc_out = c ^ 0xFFFFFFFFu;
    return DNNCA_OK;
}

}  // extern "C"
