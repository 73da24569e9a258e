// The "S24" form of a bit matrix: what v_smfmac_i32_16x16x128_i8 takes as its 2:4 structured-sparse A operand (xf_bits_i8s.hip).
// Plain C, host and device: tests/ compile the encoder with gcc and check it against a model of the instruction.
//
// Reduction indices come in 512-blocks = 16 words of the bit matrix; word (g, t) = word 4 g + t of the block, g = lane group of the
// dense kernel, t = stage (the order of bmf_panel_pos_i8).  A GROUP OF FOUR is the four bits {s, s + 8, s + 16, s + 24} of one word,
// s = 0..7 (their digit-plane bytes are the four consecutive bytes 4 (s & 3) .. + 3 of 16-byte chunk (s >> 2) * 4 + g of stage t).
// Per group the form keeps its first two ones (lowest bit positions): two value bits and two 2-bit positions p0 < p1; the ones that
// do not fit are the caller's "overflow".
//
// Per (row, 512-block) the form is, for each A lane group a = 0..3 of the instruction:
//   idx[a][t]  (t = 0..3)   one dword per stage: slot sigma = 0..15 at bits [2 sigma, 2 sigma + 2); the slots of lane group a are the
//                           groups s = 4 (a >> 1) + q, q = 0..3, of words g = 2 (a & 1) (sigma = 2 q, 2 q + 1) and g + 1 (sigma = 8 + 2 q, + 1)
//   val[a][tp] (tp = 0, 1)  one dword per stage PAIR: the value bit of slot sigma in stage t = 2 tp + u sits at bit
//                           4 u + (sigma >> 2) + 8 (sigma & 3), so that (val >> (4 u + e)) & 0x01010101 is dword e of the operand
// A (256-row tile, 512-block) is one contiguous block of BMF_S24_GROUP_BYTES: [256 rows][4 a][4 t] idx dwords (16 KiB), then
// [256 rows][4 a][2 tp] val dwords (8 KiB); blocks in (row tile, 512-block) order.
#pragma once
#include <stdint.h>

#define BMF_S24_GROUP_BYTES 24576
#define BMF_S24_IDX_BYTES 16384

#if defined(__HIPCC__) || defined(__CUDACC__)
#define BMF_S24_HD __host__ __device__ static inline
#else
#define BMF_S24_HD static inline
#endif

// nibble m (bit b = the group's cell at bit position s + 8 b) -> p0 | p1 << 2 | v0 << 4 | v1 << 5 | extra << 6
// (extra = ones that do not fit: popcount - 2, or 0)
BMF_S24_HD unsigned bmf_s24_code(unsigned m) {
    m &= 15u;
    unsigned p[4], n = 0;
    for (unsigned b = 0; b < 4; ++b)
        if ((m >> b) & 1u) p[n++] = b;
    if (n == 0) return 0u | (1u << 2);                               // positions (0, 1), both values 0
    if (n == 1) return p[0] == 3u ? (0u | (3u << 2) | (1u << 5))     // the one at position 3 goes second: (0, 3), values (0, 1)
                                  : (p[0] | (3u << 2) | (1u << 4));  // else first: (p, 3), values (1, 0)
    return p[0] | (p[1] << 2) | (3u << 4) | ((n - 2u) << 6);
}

// The eight words of word pair h (g = 2 h, 2 h + 1; w[gi][t]) -> the idx / val dwords of lane groups a = h (s = 0..3) and a = h + 2
// (s = 4..7): idx[ai][t], val[ai][tp] with ai = 0 for a = h, 1 for a = h + 2.  Returns the number of overflow ones; kept[gi][t] = the
// words with the overflow ones cleared.
BMF_S24_HD unsigned bmf_s24_encode_pair(const uint32_t w[2][4], uint32_t idx[2][4], uint32_t val[2][2], uint32_t kept[2][4]) {
    unsigned extra = 0;
    for (int ai = 0; ai < 2; ++ai) {
        val[ai][0] = val[ai][1] = 0u;
        for (int t = 0; t < 4; ++t) idx[ai][t] = 0u;
    }
    for (int gi = 0; gi < 2; ++gi)
        for (int t = 0; t < 4; ++t) {
            const uint32_t x = w[gi][t];
            uint32_t k = 0u;
            for (int s = 0; s < 8; ++s) {
                const unsigned m = ((x >> s) & 1u) | (((x >> (s + 8)) & 1u) << 1) | (((x >> (s + 16)) & 1u) << 2) | (((x >> (s + 24)) & 1u) << 3);
                const unsigned c = bmf_s24_code(m);
                const unsigned p0 = c & 3u, p1 = (c >> 2) & 3u, v0 = (c >> 4) & 1u, v1 = (c >> 5) & 1u;
                extra += c >> 6;
                const int ai = s >> 2, q = s & 3;
                const int sg0 = 8 * gi + 2 * q, sg1 = sg0 + 1;
                idx[ai][t] |= (p0 << (2 * sg0)) | (p1 << (2 * sg1));
                const int u = t & 1;
                val[ai][t >> 1] |= (v0 << (4 * u + (sg0 >> 2) + 8 * (sg0 & 3))) | (v1 << (4 * u + (sg1 >> 2) + 8 * (sg1 & 3)));
                k |= (v0 << (s + 8 * p0)) | (v1 << (s + 8 * p1));
            }
            kept[gi][t] = k;
        }
    return extra;
}
