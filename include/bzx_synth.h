/*
 * bzx_synth.h -- deterministic synthetic inputs of SURVEY.md section 8(d) (header only).
 * Shared by libbzx.so (bzx_synth_* exports used by bench.py and the tests' GPU side) and by
 * the oracle library, so both sides generate identical bytes from a seed without shipping data.
 */
#ifndef BZX_SYNTH_H
#define BZX_SYNTH_H
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define BZX_SEED_TEXT 0x9E3779B97F4A7C15ull
#define BZX_SEED_RANDOM 0xD1B54A32D192ED03ull

static inline uint64_t bzx_xs64(uint64_t *s)
{
    uint64_t x = *s;
    x ^= x >> 12;
    x ^= x << 25;
    x ^= x >> 27;
    *s = x;
    return x * 0x2545F4914F6CDD1Dull;
}

/*
 * synthtext: xorshift64* PRNG; vocabulary of 8192 words, word length 2 + r%9, letters a..z uniform
 * with rejection of a 3rd equal consecutive letter; text = words picked with index
 * ((r1 % 8192) * (r2 % 8192)) >> 13 (skewed to low indices), separated by ' ', every 13th separator
 * '\n'; truncated to nbytes.  No byte run of length >= 4 occurs, so RLE1 is the identity on it.
 */
static inline void bzx_synth_text_impl(uint64_t seed, uint8_t *out, size_t nbytes)
{
    enum { NW = 8192, MAXL = 10 };
    static __thread uint8_t vocab[NW][MAXL];
    static __thread uint8_t vlen[NW];
    uint64_t s = seed ? seed : BZX_SEED_TEXT;
    for (int w = 0; w < NW; w++) {
        int L = 2 + (int)(bzx_xs64(&s) % 9);
        vlen[w] = (uint8_t)L;
        for (int k = 0; k < L; k++) {
            uint8_t c;
            do {
                c = (uint8_t)('a' + bzx_xs64(&s) % 26);
            } while (k >= 2 && vocab[w][k - 1] == c && vocab[w][k - 2] == c);
            vocab[w][k] = c;
        }
    }
    size_t p = 0;
    uint64_t nword = 0;
    while (p < nbytes) {
        uint64_t r1 = bzx_xs64(&s), r2 = bzx_xs64(&s);
        uint32_t w = (uint32_t)(((r1 % NW) * (r2 % NW)) >> 13);
        for (int k = 0; k < vlen[w] && p < nbytes; k++) out[p++] = vocab[w][k];
        nword++;
        if (p < nbytes) out[p++] = (nword % 13 == 0) ? '\n' : ' ';
    }
}

/* pseudo-random bytes (stand-in for /dev/urandom, reproducible) */
static inline void bzx_synth_random_impl(uint64_t seed, uint8_t *out, size_t nbytes)
{
    uint64_t s = seed ? seed : BZX_SEED_RANDOM;
    size_t p = 0;
    while (p + 8 <= nbytes) {
        uint64_t r = bzx_xs64(&s);
        memcpy(out + p, &r, 8);
        p += 8;
    }
    if (p < nbytes) {
        uint64_t r = bzx_xs64(&s);
        memcpy(out + p, &r, nbytes - p);
    }
}
#endif
