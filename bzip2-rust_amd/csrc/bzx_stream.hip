// bzx_stream.hip -- host-side stream assembler of the C ABI (bzx_stream_*).
//
// Replaces BitWriter (reference src/bitstream/bitwriter.rs:42-172) for callers that keep the
// reference's structure (compress_block per block, ordered writer thread, compress.rs:74-122):
// "BZh<level>" header (bitwriter.rs:67-72), block images appended bit-granularly without their
// padding (bitwriter.rs:94-100), combined CRC folded from bytes 6..10 of every image
// (bitwriter.rs:89-91, crc.rs:25-27), footer magic + CRC + zero padding (bitwriter.rs:103-114,158-172).
// Pure host bookkeeping, no compute; the batched device path (bzx_compress_device) does this
// layout on the GPU instead (bzx_emit.hip).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>
#include "../../include/bzx.h"

struct bzx_stream {
    std::vector<uint8_t> out;
    uint64_t queue = 0;
    int q_bits = 0;
    uint32_t crc = 0;
    int level = 9;
    bool started = false;
    bool finished = false;

    void put(int nbits, uint32_t v)
    {
        queue = (queue << nbits) | (uint64_t)v;
        q_bits += nbits;
        while (q_bits >= 8) {
            out.push_back((uint8_t)(queue >> (q_bits - 8)));
            q_bits -= 8;
        }
        queue &= (1ull << q_bits) - 1ull;
    }
    void header()
    {
        put(8, 'B');
        put(8, 'Z');
        put(8, 'h');
        put(8, (uint32_t)('0' + level));
        started = true;
    }
};

extern "C" int bzx_stream_begin(int level, bzx_stream **out)
{
    if (!out || level < 1 || level > 9) return BZX_E_PARAM;
    bzx_stream *s = new (std::nothrow) bzx_stream();
    if (!s) return BZX_E_NOMEM;
    s->level = level;
    *out = s;
    return BZX_OK;
}

extern "C" int bzx_stream_append_block(bzx_stream *s, const uint8_t *data, size_t len, uint8_t pad_bits)
{
    if (!s || !data || len < 10 || pad_bits > 7) return BZX_E_PARAM;
    if (s->finished) return BZX_E_STATE;
    try {
        if (!s->started) s->header();
        const uint32_t bc = ((uint32_t)data[6] << 24) | ((uint32_t)data[7] << 16) | ((uint32_t)data[8] << 8) | data[9];
        s->crc = ((s->crc << 1) | (s->crc >> 31)) ^ bc;
        s->out.reserve(s->out.size() + len + 16);
        if (s->q_bits == 0) {
            s->out.insert(s->out.end(), data, data + len - 1);
        } else {
            for (size_t i = 0; i + 1 < len; i++) s->put(8, data[i]);
        }
        s->put(8 - pad_bits, (uint32_t)data[len - 1] >> pad_bits);
    } catch (const std::bad_alloc &) {
        return BZX_E_NOMEM;
    }
    return BZX_OK;
}

extern "C" int bzx_stream_finish(bzx_stream *s, const uint8_t **data, size_t *len)
{
    if (!s || !data || !len) return BZX_E_PARAM;
    try {
        if (!s->finished) {
            if (!s->started) s->header();
            s->put(24, 0x177245u);
            s->put(24, 0x385090u);
            s->put(16, s->crc >> 16);
            s->put(16, s->crc & 0xffffu);
            if (s->q_bits) s->put(8 - s->q_bits, 0);
            s->finished = true;
        }
    } catch (const std::bad_alloc &) {
        return BZX_E_NOMEM;
    }
    *data = s->out.data();
    *len = s->out.size();
    return BZX_OK;
}

extern "C" void bzx_stream_free(bzx_stream *s) { delete s; }

// Synthetic inputs for bench.py and the GPU tests: host-side generators, not part of the hot path.
#include "../../include/bzx_synth.h"
extern "C" void bzx_synth_text(uint64_t seed, uint8_t *out, size_t nbytes) { bzx_synth_text_impl(seed, out, nbytes); }
extern "C" void bzx_synth_random(uint64_t seed, uint8_t *out, size_t nbytes) { bzx_synth_random_impl(seed, out, nbytes); }
