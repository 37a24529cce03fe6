// bzx_api.hip -- C ABI (include/bzx.h), context and batch orchestration of the stage kernels.
//
// Host side of the boundary described in include/bzx.h.  Mirrors the reference's driver
// (src/compression/compress.rs:40-136) but batch-shaped: every stage kernel runs once over
// all blocks of the batch, one workgroup per block, on one HIP stream; HIP events around
// each stage feed bzx_stats.  No CPU implementation of any stage exists here.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <algorithm>
#include <new>
#include <string>
#include <vector>
#include "../../include/bzx.h"
#include "bzx_device.h"

void bzx_launch_bwt(const BzxBatch &B, uint32_t grid, hipStream_t stream);
void bzx_launch_bsplit(const BzxBatch &B, uint32_t grid, uint32_t grid_deep, hipStream_t stream);
void bzx_launch_bsort(const BzxBatch &B, uint32_t grid, hipStream_t stream);
void bzx_launch_brank(const BzxBatch &B, uint32_t grid, hipStream_t stream);
void bzx_launch_bgiant(const BzxBatch &B, uint32_t grid, hipStream_t stream);
void bzx_launch_pack_max(const BzxBatch &B, uint32_t world, uint64_t *d_out, hipStream_t stream);
void bzx_launch_periodic(const BzxBatch &B, uint32_t grid, hipStream_t stream);
void bzx_launch_mtf(const BzxBatch &B, uint32_t grid, hipStream_t stream);
void bzx_launch_huffman(const BzxBatch &B, uint32_t grid, hipStream_t stream);
void bzx_launch_emit(const BzxBatch &B, uint32_t grid, hipStream_t stream);
void bzx_launch_layout(const BzxBatch &B, uint64_t first_bit, uint64_t stride_bits, uint64_t *d_total_bits,
                       hipStream_t stream, uint64_t *d_phase = nullptr);
void bzx_launch_stream_frame(const BzxBatch &B, int level, const uint64_t *d_total_bits, uint64_t *d_out_bytes,
                             hipStream_t stream);
int bzx_split_launch_boundaries(struct bzx_ctx *ctx, const uint8_t *d_raw, size_t len, int level, uint32_t max_blocks,
                                BzxSplitWs *ws_out);
uint64_t bzx_split_tiles_per_rank(size_t len, uint32_t world);
int bzx_split_shard_runs(struct bzx_ctx *ctx, const uint8_t *d_raw, size_t len, uint32_t rank, uint32_t world, uint64_t *tiles);
int bzx_split_shard_counts(struct bzx_ctx *ctx, const uint8_t *d_raw, size_t len, uint32_t rank, uint32_t world, uint64_t *tiles);
int bzx_split_shard_boundaries(struct bzx_ctx *ctx, const uint8_t *d_raw, size_t len, int level, uint32_t max_blocks,
                               uint32_t world, uint64_t *tiles, BzxSplitWs *ws_out);
void bzx_split_launch_scatter(struct bzx_ctx *ctx, const uint8_t *d_raw, size_t len, const BzxSplitWs &ws,
                              uint32_t nblk, uint8_t *d_slabs, BzxBlock *d_blk, uint32_t own_first, uint32_t own_step);
void bzx_launch_dc_scan(const uint8_t *z, uint64_t nbytes, uint64_t *found, uint32_t *n_found, uint32_t cap, uint32_t grid,
                        hipStream_t stream);
void bzx_launch_dc_decode(const BzxBatch &B, const uint8_t *z, uint64_t nbytes, const uint64_t *starts, uint32_t max_n,
                          hipStream_t stream);
void bzx_launch_dc_ibwt(const BzxBatch &B, uint8_t *img_slabs, hipStream_t stream);
void bzx_launch_dc_expand(const BzxBatch &B, const uint8_t *img_slabs, const uint64_t *off, uint8_t *out, uint64_t cap,
                          hipStream_t stream);
void bzx_launch_block_crcs(bzx_ctx *ctx, const uint8_t *d_raw, const uint64_t *d_bounds, uint32_t *d_nblk, BzxBlock *d_blk,
                           uint32_t nblk);
uint32_t bzx_bwt_max_blocks_per_cu();
uint32_t bzx_bsort_blocks_per_cu();
void bzx_launch_bits_export(const BzxBatch &B, long long *bits, hipStream_t stream);
void bzx_launch_bits_import(const BzxBatch &B, const long long *bits, hipStream_t stream);
void bzx_launch_pack_layout(const BzxBatch &B, uint32_t first, uint32_t step, uint32_t nown, uint64_t *d_total,
                            hipStream_t stream);
void bzx_launch_zero_edges(const BzxBatch &B, const uint64_t *d_total, hipStream_t stream);
void bzx_launch_unpack(const BzxBatch &B, const uint32_t *packed, uint32_t first, uint32_t step, uint32_t nown,
                       uint32_t grid, hipStream_t stream);

struct BlockReq {
    const uint8_t *blk;
    size_t n;
    uint32_t crc;
    uint8_t *out;
    size_t cap;
    size_t out_len = 0;
    uint8_t pad = 0;
    int rc = 0;
    bool done = false;
};

struct bzx_ctx {
    int device = 0;
    int n_cu = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // Calls on one context from several host threads are serialised (api_mu); bzx_compress_block calls that arrive
    // together (the reference's rayon workers, compress.rs:125-132) are collected into one device batch (bq_*).
    std::recursive_mutex api_mu;
    std::mutex bq_mu;
    std::condition_variable bq_cv;
    std::vector<struct BlockReq *> bq_pending;
    bool bq_leader = false;
    // second stream: MTF of finished blocks runs beside the last (partial) round of the sort
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_bend = nullptr;
    hipEvent_t ev_b3 = nullptr, ev_b4 = nullptr;     // ... around the rank rounds
    hipEvent_t ev_b1 = nullptr, ev_b2 = nullptr;     // bucket sorter: after the split kernel, after the sort kernel
    bool bsort_used = false;
    uint32_t *h_counters = nullptr;                   // pinned copy of d_counters after a run
    bool overlap_used = false;       // the last run launched the overlapped pair
    bool overlap_off = false;        // ... and it turned out to be serialised on this device/runtime: do not try again
    uint32_t bwt_launches = 1;       // BWT kernel launches of the last run (telemetry)
    bool use_bsort = true;           // bucket sorter (bzx_bsort.hip) first, general sorter for what it hands over;
                                     // BZX_SORTER=general in the environment selects the general sorter alone
    std::string err;

    uint32_t cap_blocks = 0;   // block descriptor capacity (global block numbers)
    uint32_t cap_slabs = 0;    // per-block slab capacity (owned blocks)
    std::vector<void *> descs; // everything hipMalloc'ed for cap_blocks
    uint32_t n_slots = 0;      // per-workgroup scratch slots
    BzxBatch B;                // device pointers (by value into kernels)
    std::vector<void *> slabs; // everything hipMalloc'ed for cap_blocks
    std::vector<void *> slot_allocs;
    uint8_t *d_in = nullptr;   // block slab buffer owned by the context
    uint32_t *d_outbuf = nullptr;   // per-block output slabs (per-block entry points)
    uint32_t *d_counters = nullptr;
    uint64_t *d_scalars = nullptr;   // [0] total bits, [1] out bytes
    unsigned long long *d_dbg = nullptr;   // [64] phase timers, only when bzx_dbg_phase_timers(ctx, 1)
    BzxBlock *h_blk = nullptr;       // pinned mirror
    uint64_t *h_scalars = nullptr;   // pinned
    hipEvent_t ev[8];
    bzx_stats stats;

    // sharded run state (bzx_shard_prepare -> bzx_shard_emit)
    uint32_t shard_total = 0, shard_rank = 0, shard_world = 1;
    int shard_level = 0;
    uint64_t shard_packed_max = 0;   // bytes of the longest packed buffer of any rank (known after bzx_shard_emit_packed)
    size_t shard_len = 0;

    struct bzx_cstream *cs = nullptr;        // chunked stream compressor kept for bzx_compress_buffer
    std::vector<uint8_t> split_carry;        // bzx_split_rle1_chunk: raw bytes of the withheld block

    // device split scratch (bzx_rle1.hip)
    void *split_ws = nullptr;
    size_t split_ws_bytes = 0;
};

#define HIP_TRY(ctx, expr)                                                                       \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                      \
            return BZX_E_HIP;                                                                    \
        }                                                                                        \
    } while (0)

extern "C" const char *bzx_version(void) { return "bzx 0.1 (gfx950)"; }

extern "C" const char *bzx_strerror(int code)
{
    switch (code) {
    case BZX_OK: return "ok";
    case BZX_E_NODEVICE: return "no HIP device";
    case BZX_E_PARAM: return "bad parameter";
    case BZX_E_NOMEM: return "out of memory";
    case BZX_E_OUTBUF: return "output buffer too small";
    case BZX_E_HIP: return "HIP runtime error";
    case BZX_E_STATE: return "bad call sequence";
    case BZX_E_DATA: return "damaged or invalid bzip2 data";
    default: return "unknown error";
    }
}

extern "C" const char *bzx_last_error(const bzx_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }

template <typename T> static int dev_alloc(bzx_ctx *ctx, std::vector<void *> &owner, T **p, size_t count)
{
    void *v = nullptr;
    hipError_t e = hipMalloc(&v, count * sizeof(T));
    if (e != hipSuccess) {
        ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return BZX_E_NOMEM;
    }
    owner.push_back(v);
    *p = (T *)v;
    return BZX_OK;
}

static void free_all(std::vector<void *> &v)
{
    for (void *p : v) (void)hipFree(p);
    v.clear();
}

// Block descriptors for `nblk` blocks (global block numbers) and per-block slabs for `nslab` of them (the blocks
// this context owns: all of them, or every world-th one of a sharded run -- BZX_SLAB in bzx_device.h).
static int ensure_blocks(bzx_ctx *ctx, uint32_t nblk, uint32_t nslab = 0)
{
    if (nslab == 0 || nslab > nblk) nslab = nblk;
    BzxBatch &B = ctx->B;
    int rc;
    if (nblk > ctx->cap_blocks) {
        free_all(ctx->descs);
        ctx->cap_blocks = 0;
        const uint32_t cap = nblk < 16 ? 16 : nblk;
        if ((rc = dev_alloc(ctx, ctx->descs, &B.blk, cap))) return rc;
        if ((rc = dev_alloc(ctx, ctx->descs, &B.plist, (size_t)cap))) return rc;
        if ((rc = dev_alloc(ctx, ctx->descs, &B.redo_list, (size_t)cap))) return rc;
        if ((rc = dev_alloc(ctx, ctx->descs, &B.resume_list, (size_t)cap))) return rc;
        if (ctx->h_blk) (void)hipHostFree(ctx->h_blk);
        ctx->h_blk = nullptr;
        if (hipHostMalloc((void **)&ctx->h_blk, (size_t)cap * sizeof(BzxBlock), 0) != hipSuccess) return BZX_E_NOMEM;
        ctx->cap_blocks = cap;
    }
    if (nslab <= ctx->cap_slabs) return BZX_OK;
    free_all(ctx->slabs);
    ctx->cap_slabs = 0;
    const uint32_t cap = nslab < 16 ? 16 : nslab;
    if ((rc = dev_alloc(ctx, ctx->slabs, &ctx->d_in, (size_t)cap * BZX_BLK_STRIDE))) return rc;
    if ((rc = dev_alloc(ctx, ctx->slabs, &B.bwt, (size_t)cap * BZX_BLK_STRIDE))) return rc;
    if ((rc = dev_alloc(ctx, ctx->slabs, &B.rank, (size_t)cap * BZX_BLK_STRIDE))) return rc;
    if ((rc = dev_alloc(ctx, ctx->slabs, &B.mtfv, (size_t)cap * BZX_BLK_STRIDE))) return rc;
    if ((rc = dev_alloc(ctx, ctx->slabs, &B.freq, (size_t)cap * 260))) return rc;
    if ((rc = dev_alloc(ctx, ctx->slabs, &B.in_use, (size_t)cap * 256))) return rc;
    if ((rc = dev_alloc(ctx, ctx->slabs, &B.len, (size_t)cap * 6 * 260))) return rc;
    if ((rc = dev_alloc(ctx, ctx->slabs, &B.code, (size_t)cap * 6 * 260))) return rc;
    if ((rc = dev_alloc(ctx, ctx->slabs, &B.selector, (size_t)cap * BZX_SEL_STRIDE))) return rc;
    if ((rc = dev_alloc(ctx, ctx->slabs, &B.selector_mtf, (size_t)cap * BZX_SEL_STRIDE))) return rc;
    if ((rc = dev_alloc(ctx, ctx->slabs, &B.gbits, (size_t)cap * BZX_SEL_STRIDE))) return rc;
    if (ctx->use_bsort) {
        if ((rc = dev_alloc(ctx, ctx->slabs, &B.pk, (size_t)cap * BZX_PK_STRIDE))) return rc;
        if ((rc = dev_alloc(ctx, ctx->slabs, &B.rec_a, (size_t)cap * BZX_MAX_N))) return rc;
        if ((rc = dev_alloc(ctx, ctx->slabs, &B.rec_b, (size_t)cap * BZX_MAX_N))) return rc;
        if ((rc = dev_alloc(ctx, ctx->slabs, &B.bk_list, (size_t)cap * BZX_BK_PER_BLOCK))) return rc;
        B.bk_cap = cap * BZX_BK_PER_BLOCK;
        if ((rc = dev_alloc(ctx, ctx->slabs, &B.rk_list, (size_t)B.bk_cap * 2))) return rc;
        if ((rc = dev_alloc(ctx, ctx->slabs, &B.deep_list, (size_t)cap * BZX_DEEP_PER_BLOCK * 4 * 3))) return rc;
        if ((rc = dev_alloc(ctx, ctx->slabs, &B.isa2, (size_t)cap * 2 * BZX_MAX_N))) return rc;
        B.rk_blocks = cap;
    }
    // MTF stage: positions of the run heads, 4 B each and one more.  With the bucket sorter they live in the block's
    // record slab (8 B per rotation, dead once the BWT of the block is done); else in a slab of their own.
    if (ctx->use_bsort) {
        B.hpos = reinterpret_cast<uint32_t *>(B.rec_a);
        B.hpos_stride = 2 * BZX_MAX_N;
    } else {
        if ((rc = dev_alloc(ctx, ctx->slabs, &B.hpos, (size_t)cap * (BZX_MAX_N + 64)))) return rc;
        B.hpos_stride = BZX_MAX_N + 64;
    }
    if ((rc = dev_alloc(ctx, ctx->slabs, &ctx->d_outbuf, (size_t)cap * (BZX_OUT_STRIDE / 4)))) return rc;
    ctx->cap_slabs = cap;
    return BZX_OK;
}

static int ensure_slots(bzx_ctx *ctx, uint32_t n_slots)
{
    if (n_slots <= ctx->n_slots) return BZX_OK;
    free_all(ctx->slot_allocs);
    ctx->n_slots = 0;
    std::vector<BzxSortWs> h(n_slots);
    int rc;
    // one slab per array kind, carved per slot (a few large allocations instead of thousands)
    uint64_t *u = nullptr;
    uint32_t *w = nullptr;
    if ((rc = dev_alloc(ctx, ctx->slot_allocs, &u, (size_t)n_slots * 2 * BZX_MAX_N))) return rc;
    if ((rc = dev_alloc(ctx, ctx->slot_allocs, &w, (size_t)n_slots * 4 * BZX_MAX_N))) return rc;
    for (uint32_t i = 0; i < n_slots; i++) {
        h[i].u0 = u + (size_t)i * 2 * BZX_MAX_N;
        h[i].u1 = h[i].u0 + BZX_MAX_N;
        h[i].s0 = w + (size_t)i * 4 * BZX_MAX_N;
        h[i].s1 = h[i].s0 + BZX_MAX_N;
        h[i].isa = h[i].s1 + BZX_MAX_N;
        h[i].sa = h[i].isa + BZX_MAX_N;
    }
    if ((rc = dev_alloc(ctx, ctx->slot_allocs, &ctx->B.sort_ws, (size_t)n_slots))) return rc;
    HIP_TRY(ctx, hipMemcpy(ctx->B.sort_ws, h.data(), n_slots * sizeof(BzxSortWs), hipMemcpyHostToDevice));
    ctx->n_slots = n_slots;
    ctx->B.n_slots = n_slots;
    return BZX_OK;
}

extern "C" int bzx_ctx_create(int device, uint32_t max_blocks, bzx_ctx **out)
{
    if (!out) return BZX_E_PARAM;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return BZX_E_NODEVICE;
    if (hipSetDevice(device) != hipSuccess) return BZX_E_NODEVICE;
    bzx_ctx *ctx = new (std::nothrow) bzx_ctx();
    if (!ctx) return BZX_E_NOMEM;
    memset(&ctx->B, 0, sizeof(ctx->B));
    memset(&ctx->stats, 0, sizeof(ctx->stats));
    ctx->device = device;
    {
        const char *e = getenv("BZX_SORTER");
        if (e && !strcmp(e, "general")) ctx->use_bsort = false;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        delete ctx;
        return BZX_E_NODEVICE;
    }
    ctx->n_cu = prop.multiProcessorCount;
    if (hipStreamCreate(&ctx->stream) != hipSuccess) {
        delete ctx;
        return BZX_E_NODEVICE;
    }
    ctx->own_stream = true;
    for (int i = 0; i < 8; i++) (void)hipEventCreate(&ctx->ev[i]);
    // A stream of another priority gets a hardware queue of its own; with the default priority HIP may map it onto
    // the queue of the caller's stream (it does once RCCL has created its streams), which would serialise the
    // overlapped MTF launch behind the sort instead of running it beside it.
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if (hipStreamCreateWithPriority(&ctx->aux, hipStreamNonBlocking, prio_greatest) != hipSuccess ||
        hipEventCreate(&ctx->ev_fork) != hipSuccess || hipEventCreate(&ctx->ev_join) != hipSuccess ||
        hipEventCreate(&ctx->ev_bend) != hipSuccess || hipEventCreate(&ctx->ev_b1) != hipSuccess ||
        hipEventCreate(&ctx->ev_b2) != hipSuccess || hipEventCreate(&ctx->ev_b3) != hipSuccess ||
        hipEventCreate(&ctx->ev_b4) != hipSuccess) {
        bzx_ctx_destroy(ctx);
        return BZX_E_HIP;
    }
    bool ok = hipMalloc((void **)&ctx->d_counters, BZX_N_COUNTERS * sizeof(uint32_t)) == hipSuccess &&
              hipMalloc((void **)&ctx->d_scalars, 8 * sizeof(uint64_t)) == hipSuccess &&
              hipHostMalloc((void **)&ctx->h_scalars, 8 * sizeof(uint64_t), 0) == hipSuccess &&
              hipHostMalloc((void **)&ctx->h_counters, 64 * sizeof(uint32_t), 0) == hipSuccess;
    if (!ok || ensure_blocks(ctx, max_blocks ? max_blocks : 16) != BZX_OK) {
        bzx_ctx_destroy(ctx);
        return BZX_E_NOMEM;
    }
    *out = ctx;
    return BZX_OK;
}

extern "C" void bzx_cstream_end(struct bzx_cstream *s);

extern "C" void bzx_ctx_destroy(bzx_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->cs) {
        bzx_cstream_end(ctx->cs);
        ctx->cs = nullptr;
    }
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->aux) (void)hipStreamSynchronize(ctx->aux);
    free_all(ctx->slabs);
    free_all(ctx->descs);
    free_all(ctx->slot_allocs);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->d_scalars) (void)hipFree(ctx->d_scalars);
    if (ctx->split_ws) (void)hipFree(ctx->split_ws);
    if (ctx->h_blk) (void)hipHostFree(ctx->h_blk);
    if (ctx->h_scalars) (void)hipHostFree(ctx->h_scalars);
    if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
    if (ctx->ev_b1) (void)hipEventDestroy(ctx->ev_b1);
    if (ctx->ev_b2) (void)hipEventDestroy(ctx->ev_b2);
    if (ctx->ev_b3) (void)hipEventDestroy(ctx->ev_b3);
    if (ctx->ev_b4) (void)hipEventDestroy(ctx->ev_b4);
    for (int i = 0; i < 8; i++) (void)hipEventDestroy(ctx->ev[i]);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->ev_bend) (void)hipEventDestroy(ctx->ev_bend);
    if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int bzx_ctx_set_stream(bzx_ctx *ctx, void *hip_stream)
{
    if (!ctx) return BZX_E_PARAM;
    std::unique_lock<std::recursive_mutex> api_lock_(ctx->api_mu);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->aux) (void)hipStreamSynchronize(ctx->aux);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return BZX_OK;
}

extern "C" int bzx_get_stats(const bzx_ctx *ctx, bzx_stats *out)
{
    if (!ctx || !out) return BZX_E_PARAM;
    std::unique_lock<std::recursive_mutex> api_lock_(const_cast<bzx_ctx *>(ctx)->api_mu);
    *out = ctx->stats;
    return BZX_OK;
}

extern "C" int bzx_get_block_info(const bzx_ctx *ctx, uint32_t block, bzx_block_info *out)
{
    if (!ctx || !out) return BZX_E_PARAM;
    std::unique_lock<std::recursive_mutex> api_lock_(const_cast<bzx_ctx *>(ctx)->api_mu);
    if (!ctx->h_blk || block >= ctx->stats.nblk || block >= ctx->cap_blocks) return BZX_E_PARAM;
    const BzxBlock &d = ctx->h_blk[block];
    out->n = d.n;
    out->crc = d.crc;
    out->orig_ptr = d.orig_ptr;
    out->periodic = (d.status & BZX_ST_PERIODIC) ? 1u : 0u;
    out->n_in_use = d.n_in_use;
    out->n_mtf = d.n_mtf;
    out->n_tables = d.n_groups;
    out->n_selectors = d.n_selectors;
    out->bits_selectors = d.sec_bits[0];
    out->bits_tables = d.sec_bits[1];
    out->bits_payload = d.sec_bits[2];
    out->bits_symbol_map = d.sec_bits[3];
    out->bits = d.bits;
    return BZX_OK;
}

// Number of workgroups for a one-workgroup-per-block kernel over nblk blocks.
static uint32_t grid_for(const bzx_ctx *ctx, uint32_t nblk, uint32_t per_cu)
{
    uint32_t g = (uint32_t)ctx->n_cu * per_cu;
    return nblk < g ? nblk : g;
}

enum { STG_BWT = 1, STG_MTF = 2, STG_HUF = 4, STG_EMIT = 8, STG_ALL = 15 };

// Runs the stage kernels over blocks [0,nblk) whose descriptors (in_off,n,crc) are already on the device.
// STG_EMIT: out_level == 0 -> every block image byte-aligned in its own slab of ctx->d_outbuf;
//           out_level 1..9 -> one .bz2 stream in d_stream_out (cap bytes, device memory).
//           out_level -1   -> one CHUNK of a stream (bzx_cstream_*): the block images back to back in d_stream_out,
//                             starting at the bit phase d_phase[0] (kept on the device from chunk to chunk); no header,
//                             no footer, no host synchronisation
static int run_stages(bzx_ctx *ctx, uint32_t nblk, int stages, int out_level = 0, void *d_stream_out = nullptr,
                      size_t stream_cap = 0, uint64_t *d_phase = nullptr)
{
    BzxBatch &B = ctx->B;
    B.nblk = nblk;
    B.counters = ctx->d_counters;
    if (B.blk_step == 0) B.blk_step = 1;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters, 0, BZX_N_COUNTERS * sizeof(uint32_t), ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    B.ctr_bwt = 0;
    B.ctr_mtf = 1;
    ctx->bwt_launches = 1;
    bool mtf_done = false;
    ctx->bsort_used = false;
    if ((stages & STG_BWT) && nblk && ctx->use_bsort) {
        // bucket sorter: split every block into LDS-sized buckets, sort the buckets (any workgroup, any block), then
        // the general sorter takes the blocks that were handed over (deep repeats, periodic blocks); usually none
        const uint32_t ncu = (uint32_t)ctx->n_cu;
        uint32_t per_cu = bzx_bwt_max_blocks_per_cu();
        int rc = ensure_slots(ctx, ncu * per_cu);
        if (rc) return rc;
        // this launch's share of the bucket work lists: eight lists of items / 8 (cap_slabs >= 16, so at least 4096 each);
        // ALL of it is zeroed, so an item the split kernel reserved but did not write is an empty one
        const size_t cap_all = (size_t)ctx->cap_slabs * BZX_BK_PER_BLOCK;
        const size_t want = (size_t)(nblk < 16 ? 16 : nblk) * BZX_BK_PER_BLOCK;       // (a lone repetitive block emits thousands of one-bucket splits)
        const size_t items = (want < cap_all ? want : cap_all) & ~(size_t)7;
        B.bk_cap = (uint32_t)items;
        // few blocks: their buckets are dealt over all eight lists, so that every compute unit gets work; many: a list
        // holds whole blocks (one L2 per block)
        B.bk_affine = nblk >= 64 ? 1u : 0u;
        HIP_TRY(ctx, hipMemsetAsync(B.bk_list, 0, items * sizeof(BzxBucket), ctx->stream));
        // the two lists of oversized bins (deeper split levels): zeroed, an item reserved but not written is an empty one
        const size_t deep_all = (size_t)ctx->cap_slabs * BZX_DEEP_PER_BLOCK;
        B.deep_cap = (uint32_t)((size_t)nblk * BZX_DEEP_PER_BLOCK < deep_all ? (size_t)nblk * BZX_DEEP_PER_BLOCK : deep_all);
        HIP_TRY(ctx, hipMemsetAsync(B.deep_list, 0, (size_t)B.deep_cap * 2 * 16, ctx->stream));
        bzx_launch_bsplit(B, nblk < ncu ? nblk : ncu, ncu, ctx->stream);
        HIP_TRY(ctx, hipEventRecord(ctx->ev_b1, ctx->stream));
        // Blocks the split kernel refused (oversized bins beyond its depth / split limits: a handful in real data)
        // are sorted from scratch by the general sorter, one workgroup each, 10-60 ms: started right away on the
        // high-priority side stream, in sort slots of their own, beside the bucket sort of everything else.
        const bool early = ctx->n_slots >= 64;
        const uint32_t n_early = 32;
        B.rk_slot0 = 0;
        B.rk_blocks = ctx->cap_slabs;                               // every block in resume state gets its two rank arrays
#ifdef BZX_STRESS_FEW_RANK_ARRAYS                                   // (stress builds: only eight blocks get rank arrays)
        B.rk_blocks = 8;
#endif
        B.slot_base = 0;
        B.redo_once = 0;
        if (early) {
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_b1, 0));
            BzxBatch Be = B;
            Be.redo = 1;
            Be.ctr_bwt = BZX_CTR_REDO_FETCH;
            Be.redo_once = 1;
            bzx_launch_bwt(Be, n_early, ctx->aux);
            HIP_TRY(ctx, hipEventRecord(ctx->ev_join, ctx->aux));
        }
        B.bsort_mode = 0;
        bzx_launch_bsort(B, bzx_bsort_blocks_per_cu() * ncu, ctx->stream);
        HIP_TRY(ctx, hipEventRecord(ctx->ev_b2, ctx->stream));
        ctx->bsort_used = true;
        // the blocks in which a bucket gave up (deep repeats): the fill pass writes the order of their finished
        // buckets and enters their ranks into the block's two rank arrays, the regrouping pass turns oversized groups
        // into ordinary items, prefix-tripling rank rounds over the open buckets finish the leftover groups -- any
        // workgroup on any bucket, each launch exits at once when nothing is open.  Then the general sorter takes what is
        // left: refused blocks beyond the early launch's 32, and resume blocks still open after the rank rounds (periodic
        // blocks, oversized groups the regrouping pass could not dissolve and the groups that read their coarse ranks,
        // stress builds: blocks without rank arrays).
        BzxBatch Bf = B;
        Bf.bsort_mode = 1;
        bzx_launch_bsort(Bf, bzx_bsort_blocks_per_cu() * ncu, ctx->stream);
        HIP_TRY(ctx, hipEventRecord(ctx->ev_b3, ctx->stream));
        bzx_launch_bgiant(B, ncu, ctx->stream);                      // (a no-op launch unless the split left an oversized group)
        bzx_launch_brank(B, bzx_bsort_blocks_per_cu() * ncu, ctx->stream);
        HIP_TRY(ctx, hipEventRecord(ctx->ev_b4, ctx->stream));
        if (early) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        BzxBatch Br = B;
        Br.redo = 1;
        Br.ctr_bwt = BZX_CTR_REDO_FETCH;
        bzx_launch_bwt(Br, grid_for(ctx, nblk, per_cu), ctx->stream);
        Br.redo = 2;
        Br.ctr_bwt = BZX_CTR_RESUME_FETCH;
        bzx_launch_bwt(Br, grid_for(ctx, nblk, per_cu), ctx->stream);
        bzx_launch_periodic(B, ctx->n_slots < 64 ? ctx->n_slots : 64, ctx->stream);
    } else if ((stages & STG_BWT) && nblk) {
        uint32_t per_cu = bzx_bwt_max_blocks_per_cu();
        uint32_t grid = grid_for(ctx, nblk, per_cu);
        int rc = ensure_slots(ctx, (uint32_t)ctx->n_cu * per_cu);
        if (rc) return rc;
        // The sort runs one workgroup per compute unit, so its last round leaves (n_cu - nblk % n_cu) units idle
        // for a whole block time.  With enough blocks, sort the full rounds first, then run the partial round
        // beside the MTF stage of blocks that are already sorted (second stream, exactly the idle units).
        const uint32_t ncu = (uint32_t)ctx->n_cu;
        const uint32_t rem = nblk % ncu, idle = ncu - rem;
        ctx->overlap_used = false;
        if ((stages & STG_MTF) && per_cu == 1 && nblk > ncu && rem != 0 && idle * 8 >= ncu && !ctx->overlap_off) {
            const uint32_t full = nblk - rem;
            const uint32_t n_a1 = full < idle * 6 ? full : idle * 6;      // a block sorts in roughly 6 MTF times
            BzxBatch Ba = B;
            Ba.nblk = full;
            bzx_launch_bwt(Ba, ncu, ctx->stream);
            HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
            BzxBatch Bb = B;
            Bb.blk_first = B.blk_first + full * B.blk_step;
            Bb.nblk = rem;
            Bb.ctr_bwt = 6;
            bzx_launch_bwt(Bb, rem, ctx->stream);
            HIP_TRY(ctx, hipEventRecord(ctx->ev_bend, ctx->stream));
            ctx->overlap_used = true;
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
            BzxBatch Bm = B;
            Bm.nblk = n_a1;
            Bm.ctr_mtf = 7;
            bzx_launch_mtf(Bm, idle < n_a1 ? idle : n_a1, ctx->aux);
            HIP_TRY(ctx, hipEventRecord(ctx->ev_join, ctx->aux));
            bzx_launch_periodic(B, 16, ctx->stream);
            HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
            if (nblk > n_a1) {
                BzxBatch Br = B;
                Br.blk_first = B.blk_first + n_a1 * B.blk_step;
                Br.nblk = nblk - n_a1;
                bzx_launch_mtf(Br, grid_for(ctx, Br.nblk, 1), ctx->stream);
            }
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
            ctx->bwt_launches = 2;
            mtf_done = true;
        } else {
            bzx_launch_bwt(B, grid, ctx->stream);
            bzx_launch_periodic(B, grid < 16 ? grid : 16, ctx->stream);   // no-op unless blocks were flagged periodic
        }
    }
    if (!mtf_done) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
        if ((stages & STG_MTF) && nblk) bzx_launch_mtf(B, grid_for(ctx, nblk, 1), ctx->stream);
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    if ((stages & STG_HUF) && nblk) bzx_launch_huffman(B, grid_for(ctx, nblk, 3), ctx->stream);
    HIP_TRY(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    if (stages & STG_EMIT) {
        if (out_level < 0) {
            B.out = (uint32_t *)d_stream_out;
            bzx_launch_layout(B, 0, 0, ctx->d_scalars, ctx->stream, d_phase);
            HIP_TRY(ctx, hipMemsetAsync(d_stream_out, 0, stream_cap, ctx->stream));
            if (nblk) bzx_launch_emit(B, grid_for(ctx, nblk, 2), ctx->stream);
        } else if (out_level == 0) {
            B.out = ctx->d_outbuf;
            bzx_launch_layout(B, 0, (uint64_t)BZX_OUT_STRIDE * 8, ctx->d_scalars, ctx->stream);
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_outbuf, 0, (size_t)nblk * BZX_OUT_STRIDE, ctx->stream));
            if (nblk) bzx_launch_emit(B, grid_for(ctx, nblk, 2), ctx->stream);
        } else {
            // sizes are known after the Huffman stage: lay the stream out, check it fits, then emit
            B.out = (uint32_t *)d_stream_out;
            bzx_launch_layout(B, 32, 0, ctx->d_scalars, ctx->stream);
            HIP_TRY(ctx, hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, sizeof(uint64_t), hipMemcpyDeviceToHost,
                                        ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            const uint64_t out_bytes = (ctx->h_scalars[0] + 80 + 7) >> 3;
            const uint64_t need = (out_bytes + 3) & ~3ull;
            ctx->h_scalars[1] = out_bytes;
            if (need > stream_cap) {
                ctx->err = "output buffer too small for the compressed stream";
                return BZX_E_OUTBUF;
            }
            HIP_TRY(ctx, hipMemsetAsync(d_stream_out, 0, need, ctx->stream));
            if (nblk) bzx_launch_emit(B, grid_for(ctx, nblk, 2), ctx->stream);
            bzx_launch_stream_frame(B, out_level, ctx->d_scalars, ctx->d_scalars + 1, ctx->stream);
        }
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_counters, ctx->d_counters, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipGetLastError());
    return BZX_OK;
}

int bzx_ctx_split_scratch(bzx_ctx *ctx, size_t bytes, void **p)
{
    if (bytes > ctx->split_ws_bytes) {
        if (ctx->split_ws) (void)hipFree(ctx->split_ws);
        ctx->split_ws = nullptr;
        ctx->split_ws_bytes = 0;
        if (hipMalloc(&ctx->split_ws, bytes) != hipSuccess) {
            ctx->err = "hipMalloc(split scratch) failed";
            return BZX_E_NOMEM;
        }
        ctx->split_ws_bytes = bytes;
    }
    *p = ctx->split_ws;
    return BZX_OK;
}
hipStream_t bzx_ctx_stream(bzx_ctx *ctx) { return ctx->stream; }
int bzx_ctx_ncu(bzx_ctx *ctx) { return ctx->n_cu; }

static void collect_stage_times(bzx_ctx *ctx)
{
    // Did the overlapped MTF launch really run beside the partial sort round?  If the runtime put both streams on
    // one hardware queue the MTF launch finished a whole launch time after the sort round instead of with it:
    // then the split costs time, and later runs on this context use the plain order.
    if (ctx->overlap_used) {
        float t_sort = 0.f, t_join = 0.f;
        if (hipEventElapsedTime(&t_sort, ctx->ev_fork, ctx->ev_bend) == hipSuccess &&
            hipEventElapsedTime(&t_join, ctx->ev_fork, ctx->ev_join) == hipSuccess && t_sort > 0.f &&
            t_join - t_sort > 0.6f * t_sort)
            ctx->overlap_off = true;
        ctx->overlap_used = false;
    }
    float ms[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; i++) (void)hipEventElapsedTime(&ms[i], ctx->ev[i], ctx->ev[i + 1]);
    ctx->stats.ms_bwt = ms[0];
    ctx->stats.n_redo = 0;
    ctx->stats.n_buckets = 0;
    ctx->stats.ms_bwt_split = ctx->stats.ms_bwt_sort = ctx->stats.ms_bwt_general = ctx->stats.ms_bwt_rank = 0.f;
    ctx->stats.n_open_buckets = ctx->stats.n_open_left = ctx->stats.n_resume_left = ctx->stats.n_from_scratch = ctx->stats.n_unsorted = 0;
    if (ctx->bsort_used) {
        ctx->stats.n_redo = ctx->h_counters[BZX_CTR_REDO] + ctx->h_counters[BZX_CTR_RESUME];
        ctx->stats.n_buckets = 0;
        for (int x = 0; x < 8; x++) ctx->stats.n_buckets += ctx->h_counters[BZX_CTR_BK_LIST0 + x];
        (void)hipEventElapsedTime(&ctx->stats.ms_bwt_split, ctx->ev[0], ctx->ev_b1);
        (void)hipEventElapsedTime(&ctx->stats.ms_bwt_sort, ctx->ev_b1, ctx->ev_b2);
        (void)hipEventElapsedTime(&ctx->stats.ms_bwt_general, ctx->ev_b2, ctx->ev[1]);
        (void)hipEventElapsedTime(&ctx->stats.ms_bwt_rank, ctx->ev_b3, ctx->ev_b4);
        ctx->stats.n_open_buckets = ctx->h_counters[BZX_CTR_RK_ITEMS];
        ctx->stats.n_open_left = ctx->h_counters[BZX_CTR_RK_OPEN];
        ctx->stats.n_resume_left = ctx->h_counters[BZX_CTR_RESUME_LEFT];
        ctx->stats.n_from_scratch = ctx->h_counters[BZX_CTR_REDO];
        ctx->stats.n_unsorted = ctx->h_counters[BZX_CTR_STAT0 + 15];

    }
    ctx->stats.bwt_launches = ctx->bwt_launches;
    ctx->stats.ms_mtf = ms[1];
    ctx->stats.ms_huffman = ms[2];
    ctx->stats.ms_emit = ms[3];
}

static int check_blk_args(const uint8_t *p, size_t n)
{
    if (!p || n == 0 || n > BZX_MAX_BLOCK) return BZX_E_PARAM;
    return BZX_OK;
}

extern "C" int bzx_stage_bwt(bzx_ctx *ctx, const uint8_t *blk, size_t n, uint8_t *bwt_out, uint32_t *orig_ptr,
                             uint32_t *status)
{
    std::unique_lock<std::recursive_mutex> api_lock_;
    if (ctx) api_lock_ = std::unique_lock<std::recursive_mutex>(ctx->api_mu);
    if (!ctx || !bwt_out || !orig_ptr || check_blk_args(blk, n)) return BZX_E_PARAM;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_blocks(ctx, 1);
    if (rc) return rc;
    memset(&ctx->h_blk[0], 0, sizeof(BzxBlock));
    ctx->h_blk[0].in_off = 0;
    ctx->h_blk[0].n = (uint32_t)n;
    ctx->B.in = ctx->d_in;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_in, blk, n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->B.blk, ctx->h_blk, sizeof(BzxBlock), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = run_stages(ctx, 1, STG_BWT))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(bwt_out, ctx->B.bwt, n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_blk, ctx->B.blk, sizeof(BzxBlock), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *orig_ptr = ctx->h_blk[0].orig_ptr;
    if (status) *status = ctx->h_blk[0].status & BZX_ST_PERIODIC;
    return BZX_OK;
}

#ifdef BZX_DIAG
// Debug/bench helper (not part of include/bzx.h): replicate one block `reps` times as a batch, run the
// given stages, return the HIP-event time of each stage in ms[4] (bwt, mtf, huffman, emit).
extern "C" int bzx_dbg_time_stages(bzx_ctx *ctx, const uint8_t *blk, size_t n, uint32_t reps, int stages, float ms[4])
{
    std::unique_lock<std::recursive_mutex> api_lock_;
    if (ctx) api_lock_ = std::unique_lock<std::recursive_mutex>(ctx->api_mu);
    if (!ctx || check_blk_args(blk, n) || reps == 0) return BZX_E_PARAM;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_blocks(ctx, reps);
    if (rc) return rc;
    ctx->B.in = ctx->d_in;
    for (uint32_t b = 0; b < reps; b++) {
        memset(&ctx->h_blk[b], 0, sizeof(BzxBlock));
        ctx->h_blk[b].in_off = (uint64_t)b * BZX_BLK_STRIDE;
        ctx->h_blk[b].n = (uint32_t)n;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_in + (size_t)b * BZX_BLK_STRIDE, blk, n, hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->B.blk, ctx->h_blk, reps * sizeof(BzxBlock), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = run_stages(ctx, reps, stages))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 4; i++) {
        ms[i] = 0.f;
        (void)hipEventElapsedTime(&ms[i], ctx->ev[i], ctx->ev[i + 1]);
    }
    return BZX_OK;
}
#endif   // BZX_DIAG

extern "C" int bzx_stage_mtf(bzx_ctx *ctx, const uint8_t *bwt, size_t n, uint16_t *mtfv_out, uint32_t *n_mtf,
                             uint32_t freq_out[258], uint8_t in_use_out[256])
{
    std::unique_lock<std::recursive_mutex> api_lock_;
    if (ctx) api_lock_ = std::unique_lock<std::recursive_mutex>(ctx->api_mu);
    if (!ctx || !mtfv_out || !n_mtf || !freq_out || !in_use_out || check_blk_args(bwt, n)) return BZX_E_PARAM;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_blocks(ctx, 1);
    if (rc) return rc;
    memset(&ctx->h_blk[0], 0, sizeof(BzxBlock));
    ctx->h_blk[0].n = (uint32_t)n;
    ctx->B.in = ctx->d_in;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->B.bwt, bwt, n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->B.blk, ctx->h_blk, sizeof(BzxBlock), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = run_stages(ctx, 1, STG_MTF))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_blk, ctx->B.blk, sizeof(BzxBlock), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t m = ctx->h_blk[0].n_mtf;
    if (m == 0 || m > n + 1) {
        ctx->err = "device MTF produced an impossible symbol count";
        return BZX_E_HIP;
    }
    *n_mtf = m;
    HIP_TRY(ctx, hipMemcpy(mtfv_out, ctx->B.mtfv, (size_t)m * 2, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(freq_out, ctx->B.freq, 258 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(in_use_out, ctx->B.in_use, 256, hipMemcpyDeviceToHost));
    return BZX_OK;
}

extern "C" int bzx_stage_huffman(bzx_ctx *ctx, const uint16_t *mtfv, uint32_t n_mtf, const uint32_t freq[258],
                                 uint32_t alpha_size, uint32_t *n_groups, uint32_t *n_selectors, uint8_t *selectors,
                                 uint8_t len_out[6][258], uint32_t code_out[6][258])
{
    std::unique_lock<std::recursive_mutex> api_lock_;
    if (ctx) api_lock_ = std::unique_lock<std::recursive_mutex>(ctx->api_mu);
    if (!ctx || !mtfv || !freq || !n_groups || !n_selectors || !selectors || !len_out || !code_out) return BZX_E_PARAM;
    if (n_mtf == 0 || n_mtf > BZX_MAX_BLOCK + 1 || alpha_size < 3 || alpha_size > BZX_MAX_ALPHA) return BZX_E_PARAM;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_blocks(ctx, 1);
    if (rc) return rc;
    memset(&ctx->h_blk[0], 0, sizeof(BzxBlock));
    ctx->h_blk[0].n = n_mtf - 1;
    ctx->h_blk[0].n_mtf = n_mtf;
    ctx->h_blk[0].n_in_use = alpha_size - 2;
    uint32_t f260[260];
    memset(f260, 0, sizeof(f260));
    memcpy(f260, freq, 258 * sizeof(uint32_t));
    HIP_TRY(ctx, hipMemcpy(ctx->B.mtfv, mtfv, (size_t)n_mtf * 2, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->B.freq, f260, sizeof(f260), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemset(ctx->B.in_use, 0, 256));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->B.blk, ctx->h_blk, sizeof(BzxBlock), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = run_stages(ctx, 1, STG_HUF))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_blk, ctx->B.blk, sizeof(BzxBlock), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *n_groups = ctx->h_blk[0].n_groups;
    *n_selectors = ctx->h_blk[0].n_selectors;
    if (*n_selectors > BZX_MAX_SEL) {
        ctx->err = "device Huffman stage produced an impossible selector count";
        return BZX_E_HIP;
    }
    HIP_TRY(ctx, hipMemcpy(selectors, ctx->B.selector, *n_selectors, hipMemcpyDeviceToHost));
    static thread_local uint8_t hl[6 * 260];
    static thread_local uint32_t hc[6 * 260];
    HIP_TRY(ctx, hipMemcpy(hl, ctx->B.len, sizeof(hl), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(hc, ctx->B.code, sizeof(hc), hipMemcpyDeviceToHost));
    for (int t = 0; t < 6; t++)
        for (int v = 0; v < 258; v++) {
            len_out[t][v] = hl[t * 260 + v];
            code_out[t][v] = hc[t * 260 + v];
        }
    return BZX_OK;
}

extern "C" int bzx_compress_blocks(bzx_ctx *ctx, uint32_t nblk, const uint8_t *const *blks, const size_t *ns,
                                   const uint32_t *crcs, uint8_t *const *outs, const size_t *caps, size_t *out_lens,
                                   uint8_t *pads)
{
    std::unique_lock<std::recursive_mutex> api_lock_;
    if (ctx) api_lock_ = std::unique_lock<std::recursive_mutex>(ctx->api_mu);
    if (!ctx || !blks || !ns || !crcs || !outs || !caps || !out_lens || !pads) return BZX_E_PARAM;
    if (nblk == 0) return BZX_OK;
    for (uint32_t b = 0; b < nblk; b++)
        if (check_blk_args(blks[b], ns[b]) || !outs[b]) return BZX_E_PARAM;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_blocks(ctx, nblk);
    if (rc) return rc;
    ctx->B.in = ctx->d_in;
    uint64_t rle1 = 0;
    for (uint32_t b = 0; b < nblk; b++) {
        memset(&ctx->h_blk[b], 0, sizeof(BzxBlock));
        ctx->h_blk[b].in_off = (uint64_t)b * BZX_BLK_STRIDE;
        ctx->h_blk[b].n = (uint32_t)ns[b];
        ctx->h_blk[b].crc = crcs[b];
        rle1 += ns[b];
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_in + (size_t)b * BZX_BLK_STRIDE, blks[b], ns[b], hipMemcpyHostToDevice,
                                    ctx->stream));
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->B.blk, ctx->h_blk, nblk * sizeof(BzxBlock), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = run_stages(ctx, nblk, STG_ALL, 0))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_blk, ctx->B.blk, nblk * sizeof(BzxBlock), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    collect_stage_times(ctx);
    bzx_stats &st = ctx->stats;
    st.nblk = nblk;
    st.n_periodic = 0;
    st.raw_bytes = 0;
    st.rle1_bytes = rle1;
    st.mtf_symbols = 0;
    st.out_bits = 0;
    st.ms_split = 0;
    int ret = BZX_OK;
    for (uint32_t b = 0; b < nblk; b++) {
        const BzxBlock &d = ctx->h_blk[b];
        const size_t bytes = (size_t)((d.bits + 7) >> 3);
        st.n_periodic += (d.status & BZX_ST_PERIODIC) ? 1 : 0;
        st.mtf_symbols += d.n_mtf;
        st.out_bits += d.bits;
        if (bytes > BZX_OUT_STRIDE) {
            // cannot happen: an image is at most n * 17/8 + tables, and the emit kernel clips at the slab end
            ctx->err = "block image larger than its device slab";
            (void)hipStreamSynchronize(ctx->stream);       // copies into earlier callers' buffers are in flight
            return BZX_E_HIP;
        }
        if (bytes > caps[b]) {
            ret = BZX_E_OUTBUF;
            out_lens[b] = bytes;
            continue;
        }
        if (hipMemcpyAsync(outs[b], (const uint8_t *)ctx->d_outbuf + (size_t)b * BZX_OUT_STRIDE, bytes, hipMemcpyDeviceToHost,
                           ctx->stream) != hipSuccess) {
            ctx->err = "hipMemcpyAsync(block image) failed";
            (void)hipStreamSynchronize(ctx->stream);       // earlier copies into caller buffers are in flight
            return BZX_E_HIP;
        }
        out_lens[b] = bytes;
        pads[b] = (uint8_t)((8 - (d.bits & 7)) & 7);
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    st.ms_total = st.ms_bwt + st.ms_mtf + st.ms_huffman + st.ms_emit;
    return ret;
}

// compress_block.rs:24.  Thread-safe: the reference calls compress_block from every rayon worker at once
// (compress.rs:125-132).  The first caller becomes the batch leader, waits a moment for the others, runs all pending
// blocks as ONE device batch and hands the results back; callers that arrive meanwhile form the next batch.
extern "C" int bzx_compress_block(bzx_ctx *ctx, const uint8_t *blk, size_t n, uint32_t crc, uint8_t *out, size_t cap,
                                  size_t *out_len, uint8_t *pad_bits)
{
    if (!ctx || !out || !out_len || !pad_bits || check_blk_args(blk, n)) return BZX_E_PARAM;
    BlockReq rq;
    rq.blk = blk;
    rq.n = n;
    rq.crc = crc;
    rq.out = out;
    rq.cap = cap;
    std::unique_lock<std::mutex> lk(ctx->bq_mu);
    try {
        ctx->bq_pending.push_back(&rq);
    } catch (const std::bad_alloc &) {
        return BZX_E_NOMEM;                      // (nothing may unwind across the C ABI)
    }
    while (!rq.done) {
        if (ctx->bq_leader) {
            ctx->bq_cv.wait(lk);
            continue;
        }
        ctx->bq_leader = true;
        ctx->bq_cv.wait_for(lk, std::chrono::microseconds(300));          // collection window (lock released)
        std::vector<BlockReq *> batch;
        batch.swap(ctx->bq_pending);                                      // (swap does not allocate)
        lk.unlock();
        const uint32_t nb = (uint32_t)batch.size();
        int rc = BZX_OK;
        std::vector<size_t> lens;
        try {
            std::vector<const uint8_t *> blks(nb);
            std::vector<size_t> ns(nb), caps(nb);
            std::vector<uint32_t> crcs(nb);
            std::vector<uint8_t *> outs(nb);
            std::vector<uint8_t> pads(nb, 0);
            lens.assign(nb, 0);
            for (uint32_t i = 0; i < nb; i++) {
                blks[i] = batch[i]->blk;
                ns[i] = batch[i]->n;
                crcs[i] = batch[i]->crc;
                outs[i] = batch[i]->out;
                caps[i] = batch[i]->cap;
            }
            rc = nb ? bzx_compress_blocks(ctx, nb, blks.data(), ns.data(), crcs.data(), outs.data(), caps.data(),
                                          lens.data(), pads.data())
                    : BZX_OK;
            for (uint32_t i = 0; i < nb; i++) batch[i]->pad = pads[i];
        } catch (const std::bad_alloc &) {
            rc = BZX_E_NOMEM;                    // every request of the batch fails; nobody is left waiting
        }
        lk.lock();
        for (uint32_t i = 0; i < nb; i++) {
            BlockReq *q = batch[i];
            q->out_len = i < lens.size() ? lens[i] : 0;
            // a block whose image did not fit reports that; its neighbours in the batch are fine
            q->rc = rc == BZX_E_OUTBUF ? (q->out_len > q->cap ? BZX_E_OUTBUF : BZX_OK) : rc;
            q->done = true;
        }
        ctx->bq_leader = false;
        ctx->bq_cv.notify_all();
    }
    *out_len = rq.out_len;
    *pad_bits = rq.pad;
    return rq.rc;
}

static int level_ok(int level) { return level >= 1 && level <= 9; }

// Device split: raw (device) -> block slabs + descriptors (n, crc, in_off).  Returns the block count.
static int split_on_device(bzx_ctx *ctx, const uint8_t *d_raw, size_t len, int level, uint32_t *nblk_out,
                           uint32_t own_first = 0, uint32_t own_step = 1, uint64_t *last_raw_start = nullptr,
                           uint64_t *gathered_tiles = nullptr)
{
    *nblk_out = 0;
    if (len == 0) return BZX_OK;
    const size_t nmax = (size_t)100000 * level - 19;
    const size_t max_blocks_sz = (len + len / 4) / nmax + 2;
    if (max_blocks_sz > 0x7fffffffu) return BZX_E_PARAM;
    const uint32_t max_blocks = (uint32_t)max_blocks_sz;
    int rc = ensure_blocks(ctx, max_blocks, (max_blocks + own_step - 1) / own_step + 1);
    if (rc) return rc;
    BzxSplitWs ws;
    // (gathered_tiles: the per-byte scans were done rank by rank, bzx_shard_scan_*; only the chain of boundaries is left)
    if (gathered_tiles) rc = bzx_split_shard_boundaries(ctx, d_raw, len, level, max_blocks, own_step, gathered_tiles, &ws);
    else rc = bzx_split_launch_boundaries(ctx, d_raw, len, level, max_blocks, &ws);
    if (rc) return rc;
    uint32_t *h_n = (uint32_t *)(ctx->h_scalars + 4);
    HIP_TRY(ctx, hipMemcpyAsync(h_n, ws.nblk, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t nblk = *h_n;
    if (nblk == 0 || nblk > max_blocks) {
        ctx->err = "device block splitter produced an impossible block count";
        return BZX_E_HIP;
    }
    if (last_raw_start) {
        // raw position where the last block starts: a chunked caller restarts the split there (the splitter's state
        // is clean at a block start: a block is a whole number of run pieces, bzx_rle1.hip)
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_scalars + 5, ws.blk_raw + (nblk - 1), sizeof(uint64_t), hipMemcpyDeviceToHost,
                                    ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        *last_raw_start = ctx->h_scalars[5];
    }
    bzx_split_launch_scatter(ctx, d_raw, len, ws, nblk, ctx->d_in, ctx->B.blk, own_first, own_step);
    HIP_TRY(ctx, hipGetLastError());
    ctx->B.in = ctx->d_in;
    ctx->B.raw = d_raw;
    *nblk_out = nblk;
    return BZX_OK;
}

static void fill_stats_from_blocks(bzx_ctx *ctx, uint32_t nblk, uint64_t raw_bytes)
{
    bzx_stats &st = ctx->stats;
    st.nblk = nblk;
    st.n_periodic = 0;
    st.raw_bytes = raw_bytes;
    st.rle1_bytes = 0;
    st.mtf_symbols = 0;
    for (uint32_t b = 0; b < nblk; b++) {
        const BzxBlock &d = ctx->h_blk[b];
        st.n_periodic += (d.status & BZX_ST_PERIODIC) ? 1 : 0;
        st.rle1_bytes += d.n;
        st.mtf_symbols += d.n_mtf;
    }
}

extern "C" int bzx_compress_device(bzx_ctx *ctx, const void *d_raw, size_t len, int level, void *d_out, size_t cap,
                                   size_t *out_len)
{
    std::unique_lock<std::recursive_mutex> api_lock_;
    if (ctx) api_lock_ = std::unique_lock<std::recursive_mutex>(ctx->api_mu);
    if (!ctx || !d_out || !out_len || !level_ok(level) || (len && !d_raw)) return BZX_E_PARAM;
    if (((uintptr_t)d_raw & 15u) || ((uintptr_t)d_out & 3u) || cap < 16) {
        ctx->err = "bzx_compress_device: d_raw must be 16-byte aligned, d_out 4-byte aligned, cap >= 16";
        return BZX_E_PARAM;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
    uint32_t nblk = 0;
    int rc = split_on_device(ctx, (const uint8_t *)d_raw, len, level, &nblk);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
    if ((rc = run_stages(ctx, nblk, STG_ALL, level, d_out, cap & ~(size_t)3))) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev[7], ctx->stream));
    if (nblk) HIP_TRY(ctx, hipMemcpyAsync(ctx->h_blk, ctx->B.blk, nblk * sizeof(BzxBlock), hipMemcpyDeviceToHost,
                                          ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *out_len = (size_t)ctx->h_scalars[1];
    collect_stage_times(ctx);
    fill_stats_from_blocks(ctx, nblk, len);
    ctx->stats.out_bits = (uint64_t)*out_len * 8;
    (void)hipEventElapsedTime(&ctx->stats.ms_split, ctx->ev[5], ctx->ev[6]);
    (void)hipEventElapsedTime(&ctx->stats.ms_total, ctx->ev[5], ctx->ev[7]);
    return BZX_OK;
}

extern "C" int bzx_cstream_begin(bzx_ctx *ctx, int level, size_t max_chunk, struct bzx_cstream **out);
extern "C" int bzx_cstream_feed(struct bzx_cstream *s, const uint8_t *raw, size_t len, int final, uint8_t *out, size_t cap,
                                size_t *produced);
extern "C" void bzx_cstream_end(struct bzx_cstream *s);
static int cstream_reset(struct bzx_cstream *s, int level);
static int cstream_collect_finish(struct bzx_cstream *s);
static size_t cstream_chunk_of(const struct bzx_cstream *s);
static size_t cstream_need_hint(const struct bzx_cstream *s);

// Host buffer -> host buffer: the chunked stream compressor over the whole input (H2D of chunk k+1, compression of
// chunk k and D2H of chunk k-1 overlap; no device allocation per call: the stream object is kept in the context).
// Pinned caller buffers (hipHostMalloc / hipHostRegister / bzx_host_alloc) make the copies truly asynchronous.
extern "C" int bzx_compress_buffer(bzx_ctx *ctx, const uint8_t *raw, size_t len, int level, uint8_t *out, size_t cap,
                                   size_t *out_len)
{
    std::unique_lock<std::recursive_mutex> api_lock_;
    if (ctx) api_lock_ = std::unique_lock<std::recursive_mutex>(ctx->api_mu);
    if (!ctx || !out || !out_len || !level_ok(level) || (len && !raw) || cap < 16) return BZX_E_PARAM;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // chunk: 16 MiB doubling up to 128 MiB, then one block per compute unit (256 x 900,000 B on MI355X): the kernels that
    // give a block one workgroup then run whole rounds (299 blocks of a 256 MiB chunk were 1.17 rounds, paid as two)
    size_t chunk = (size_t)16 << 20;
    while (chunk < len && chunk < ((size_t)128 << 20)) chunk <<= 1;
    if (chunk < len) chunk = (size_t)(ctx->n_cu > 0 ? ctx->n_cu : 256) * 900000u;
    int rc;
    if (ctx->cs && cstream_chunk_of(ctx->cs) < chunk) {
        bzx_cstream_end(ctx->cs);
        ctx->cs = nullptr;
    }
    if (!ctx->cs) {
        if ((rc = bzx_cstream_begin(ctx, level, chunk, &ctx->cs))) return rc;
    } else if ((rc = cstream_reset(ctx->cs, level))) {
        return rc;
    }
    chunk = cstream_chunk_of(ctx->cs);
    hipEvent_t e0 = ctx->ev[5], e1 = ctx->ev[7];
    HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
    size_t off = 0, produced = 0;
    do {
        const size_t n = len - off < chunk ? len - off : chunk;
        const int fin = off + n == len;
        if ((rc = bzx_cstream_feed(ctx->cs, raw + off, n, fin, out, cap, &produced))) {
            if (rc == BZX_E_OUTBUF) *out_len = cstream_need_hint(ctx->cs);      // (a lower bound when chunks remain)
            return rc;
        }
        off += n;
    } while (off < len);
    HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(e1));
    (void)hipEventElapsedTime(&ctx->stats.ms_total, e0, e1);
    ctx->stats.raw_bytes = len;
    *out_len = produced;
    return BZX_OK;
}

extern "C" int bzx_split_rle1(bzx_ctx *ctx, const uint8_t *raw, size_t len, int level, uint8_t *blocks_out,
                              uint32_t nblk_cap, uint32_t *ns, uint32_t *crcs, uint32_t *nblk_out)
{
    std::unique_lock<std::recursive_mutex> api_lock_;
    if (ctx) api_lock_ = std::unique_lock<std::recursive_mutex>(ctx->api_mu);
    if (!ctx || !blocks_out || !ns || !crcs || !nblk_out || !level_ok(level) || (len && !raw)) return BZX_E_PARAM;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *nblk_out = 0;
    if (len == 0) return BZX_OK;
    void *d_raw = nullptr;
    if (hipMalloc(&d_raw, len) != hipSuccess) return BZX_E_NOMEM;
    int rc = BZX_OK;
    uint32_t nblk = 0;
    if (hipMemcpyAsync(d_raw, raw, len, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = BZX_E_HIP;
    if (!rc) rc = split_on_device(ctx, (const uint8_t *)d_raw, len, level, &nblk);
    if (!rc && nblk > nblk_cap) rc = BZX_E_OUTBUF;
    if (!rc) {
        if (hipMemcpyAsync(ctx->h_blk, ctx->B.blk, nblk * sizeof(BzxBlock), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            rc = BZX_E_HIP;
    }
    for (uint32_t b = 0; !rc && b < nblk; b++) {
        ns[b] = ctx->h_blk[b].n;
        crcs[b] = ctx->h_blk[b].crc;
        if (ns[b] == 0 || ns[b] > BZX_MAX_BLOCK) {
            ctx->err = "device block splitter produced an impossible block length";
            rc = BZX_E_HIP;
            break;
        }
        const uint64_t off = ctx->h_blk[b].in_off;
        const uint8_t *src = (off & BZX_IN_RAW) ? (const uint8_t *)d_raw + (off & ~BZX_IN_RAW) : ctx->d_in + off;
        if (hipMemcpy(blocks_out + (size_t)b * BZX_MAX_BLOCK, src, ns[b], hipMemcpyDeviceToHost) != hipSuccess)
            rc = BZX_E_HIP;
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_raw);
    if (!rc) *nblk_out = nblk;
    return rc;
}

// ---- multi-GPU sharding: block i belongs to rank i mod world; no collective in here.
// The caller all-reduces (sum) d_bits between the two calls and sums the partial streams afterwards
// (torch.distributed / RCCL; see bench.py).  Declared in include/bzx.h.
static int shard_prepare(bzx_ctx *ctx, const void *d_raw, size_t len, int level, uint32_t rank, uint32_t world,
                         uint32_t *nblk_total, long long *d_bits, size_t bits_cap, uint64_t *gathered_tiles)
{
    std::unique_lock<std::recursive_mutex> api_lock_;
    if (ctx) api_lock_ = std::unique_lock<std::recursive_mutex>(ctx->api_mu);
    if (!ctx || !nblk_total || !d_bits || !level_ok(level) || world == 0 || rank >= world || (len && !d_raw)) return BZX_E_PARAM;
    if ((uintptr_t)d_raw & 15u) {
        ctx->err = "bzx_shard_prepare: d_raw must be 16-byte aligned";
        return BZX_E_PARAM;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!gathered_tiles) HIP_TRY(ctx, hipEventRecord(ctx->ev[5], ctx->stream));     // (else: bzx_shard_scan_runs did)
    uint32_t nblk = 0;
    ctx->B.blk_first = 0;
    ctx->B.blk_step = 1;
    int rc = split_on_device(ctx, (const uint8_t *)d_raw, len, level, &nblk, rank, world, nullptr, gathered_tiles);
    if (rc) return rc;
    if (nblk > bits_cap) return BZX_E_OUTBUF;
    HIP_TRY(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
    const uint32_t mine = nblk > rank ? (nblk - rank + world - 1) / world : 0;
    ctx->B.blk_first = rank;
    ctx->B.blk_step = world;
    rc = run_stages(ctx, mine, STG_BWT | STG_MTF | STG_HUF);
    if (!rc && mine) bzx_launch_bits_export(ctx->B, d_bits, ctx->stream);
    ctx->B.blk_first = 0;
    ctx->B.blk_step = 1;
    if (rc) return rc;
    ctx->shard_total = nblk;
    ctx->shard_rank = rank;
    ctx->shard_world = world;
    ctx->shard_level = level;
    ctx->shard_packed_max = 0;
    ctx->shard_len = len;
    *nblk_total = nblk;
    return BZX_OK;
}

extern "C" int bzx_shard_prepare(bzx_ctx *ctx, const void *d_raw, size_t len, int level, uint32_t rank, uint32_t world,
                                 uint32_t *nblk_total, long long *d_bits, size_t bits_cap)
{
    return shard_prepare(ctx, d_raw, len, level, rank, world, nblk_total, d_bits, bits_cap, nullptr);
}

// ---- sharded split analysis: the two per-byte passes of the block splitter run on 1/world of the
// input per rank; what the ranks exchange is 24 bytes per 8 KiB tile (the caller's all-gathers).  include/bzx.h.
static int shard_scan_args(bzx_ctx *ctx, const void *d_raw, size_t len, uint32_t rank, uint32_t world, long long *d_tiles)
{
    if (!ctx || !d_tiles || world == 0 || rank >= world || !len || !d_raw) return BZX_E_PARAM;
    if ((uintptr_t)d_raw & 15u) {
        ctx->err = "bzx_shard_scan_*: d_raw must be 16-byte aligned";
        return BZX_E_PARAM;
    }
    return BZX_OK;
}

extern "C" size_t bzx_shard_scan_entries(size_t len, uint32_t world)
{
    return world ? (size_t)bzx_split_tiles_per_rank(len, world) : 0;
}

extern "C" int bzx_shard_scan_runs(bzx_ctx *ctx, const void *d_raw, size_t len, uint32_t rank, uint32_t world, long long *d_tiles)
{
    std::unique_lock<std::recursive_mutex> api_lock_;
    if (ctx) api_lock_ = std::unique_lock<std::recursive_mutex>(ctx->api_mu);
    int rc = shard_scan_args(ctx, d_raw, len, rank, world, d_tiles);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
    rc = bzx_split_shard_runs(ctx, (const uint8_t *)d_raw, len, rank, world, (uint64_t *)d_tiles);
    HIP_TRY(ctx, hipGetLastError());
    return rc;
}

extern "C" int bzx_shard_scan_counts(bzx_ctx *ctx, const void *d_raw, size_t len, uint32_t rank, uint32_t world, long long *d_tiles)
{
    std::unique_lock<std::recursive_mutex> api_lock_;
    if (ctx) api_lock_ = std::unique_lock<std::recursive_mutex>(ctx->api_mu);
    int rc = shard_scan_args(ctx, d_raw, len, rank, world, d_tiles);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    rc = bzx_split_shard_counts(ctx, (const uint8_t *)d_raw, len, rank, world, (uint64_t *)d_tiles);
    HIP_TRY(ctx, hipGetLastError());
    return rc;
}

extern "C" int bzx_shard_prepare_scanned(bzx_ctx *ctx, const void *d_raw, size_t len, int level, uint32_t rank, uint32_t world,
                                         long long *d_tiles, uint32_t *nblk_total, long long *d_bits, size_t bits_cap)
{
    if (!d_tiles || !len) return BZX_E_PARAM;
    return shard_prepare(ctx, d_raw, len, level, rank, world, nblk_total, d_bits, bits_cap, (uint64_t *)d_tiles);
}

extern "C" int bzx_ctx_sync(bzx_ctx *ctx)
{
    std::unique_lock<std::recursive_mutex> api_lock_;
    if (ctx) api_lock_ = std::unique_lock<std::recursive_mutex>(ctx->api_mu);
    if (!ctx) return BZX_E_PARAM;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return BZX_OK;
}

static uint32_t shard_count(uint32_t nblk, uint32_t rank, uint32_t world)
{
    return nblk > rank ? (nblk - rank + world - 1) / world : 0;
}

extern "C" int bzx_shard_packed_max(bzx_ctx *ctx, size_t *max_len)
{
    std::unique_lock<std::recursive_mutex> api_lock_;
    if (ctx) api_lock_ = std::unique_lock<std::recursive_mutex>(ctx->api_mu);
    if (!ctx || !max_len) return BZX_E_PARAM;
    if (ctx->shard_level == 0 || ctx->shard_packed_max == 0) return BZX_E_STATE;
    *max_len = (size_t)ctx->shard_packed_max;
    return BZX_OK;
}

extern "C" int bzx_shard_emit_packed(bzx_ctx *ctx, const long long *d_bits_all, void *d_packed, size_t cap,
                                     size_t *packed_len, size_t *stream_len)
{
    std::unique_lock<std::recursive_mutex> api_lock_;
    if (ctx) api_lock_ = std::unique_lock<std::recursive_mutex>(ctx->api_mu);
    if (!ctx || !d_bits_all || !d_packed || !packed_len || !stream_len || ((uintptr_t)d_packed & 3u)) return BZX_E_PARAM;
    if (ctx->shard_level == 0) return BZX_E_STATE;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    BzxBatch &B = ctx->B;
    const uint32_t nblk = ctx->shard_total, rank = ctx->shard_rank, world = ctx->shard_world;
    const uint32_t mine = shard_count(nblk, rank, world);
    B.nblk = nblk;
    B.blk_first = 0;
    B.blk_step = 1;
    B.packed = 0;
    if (nblk) bzx_launch_bits_import(B, d_bits_all, ctx->stream);
    bzx_launch_layout(B, 32, 0, ctx->d_scalars, ctx->stream);                              // final positions of ALL blocks
    bzx_launch_pack_layout(B, rank, world, mine, ctx->d_scalars + 2, ctx->stream);         // my packed positions
    if (world <= 64) bzx_launch_pack_max(B, world, ctx->d_scalars + 3, ctx->stream);       // ... and the longest of any rank
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->shard_packed_max = world <= 64 ? ctx->h_scalars[3] * 4 : 0;
    const uint64_t out_bytes = (ctx->h_scalars[0] + 80 + 7) >> 3;
    const uint64_t need = ctx->h_scalars[2] * 4;
    if (need > cap) {
        ctx->err = "packed buffer too small for this rank's blocks";
        return BZX_E_OUTBUF;
    }
    HIP_TRY(ctx, hipMemsetAsync(d_packed, 0, need, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters, 0, 64 * sizeof(uint32_t), ctx->stream));
    B.out = (uint32_t *)d_packed;
    B.nblk = mine;
    B.blk_first = rank;
    B.blk_step = world;
    B.packed = 1;
    if (mine) bzx_launch_emit(B, grid_for(ctx, mine, 2), ctx->stream);
    B.packed = 0;
    HIP_TRY(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev[7], ctx->stream));
    if (nblk) HIP_TRY(ctx, hipMemcpyAsync(ctx->h_blk, B.blk, nblk * sizeof(BzxBlock), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipGetLastError());
    B.nblk = nblk;
    B.blk_first = 0;
    B.blk_step = 1;
    *packed_len = (size_t)need;
    *stream_len = (size_t)out_bytes;
    // stats over my blocks
    bzx_stats &st = ctx->stats;
    collect_stage_times(ctx);
    st.nblk = mine;
    st.n_periodic = 0;
    st.raw_bytes = ctx->shard_len / world;
    st.rle1_bytes = 0;
    st.mtf_symbols = 0;
    for (uint32_t b = rank; b < nblk; b += world) {
        const BzxBlock &d = ctx->h_blk[b];
        st.n_periodic += (d.status & BZX_ST_PERIODIC) ? 1 : 0;
        st.rle1_bytes += d.n;
        st.mtf_symbols += d.n_mtf;
    }
    st.out_bits = out_bytes * 8;
    (void)hipEventElapsedTime(&st.ms_split, ctx->ev[5], ctx->ev[6]);
    (void)hipEventElapsedTime(&st.ms_total, ctx->ev[5], ctx->ev[7]);
    return BZX_OK;
}

extern "C" int bzx_shard_assemble_begin(bzx_ctx *ctx, void *d_out, size_t cap, size_t *stream_len)
{
    std::unique_lock<std::recursive_mutex> api_lock_;
    if (ctx) api_lock_ = std::unique_lock<std::recursive_mutex>(ctx->api_mu);
    if (!ctx || !d_out || ((uintptr_t)d_out & 3u)) return BZX_E_PARAM;
    if (ctx->shard_level == 0) return BZX_E_STATE;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    BzxBatch &B = ctx->B;
    B.nblk = ctx->shard_total;
    B.blk_first = 0;
    B.blk_step = 1;
    B.out = (uint32_t *)d_out;
    const uint64_t out_bytes = (ctx->h_scalars[0] + 80 + 7) >> 3;      // total bits from bzx_shard_emit_packed
    const uint64_t need = (out_bytes + 3) & ~3ull;
    if (need > (cap & ~(size_t)3)) {
        ctx->err = "output buffer too small for the compressed stream";
        return BZX_E_OUTBUF;
    }
    bzx_launch_zero_edges(B, ctx->d_scalars, ctx->stream);       // only the words that are OR-merged, not the whole stream
    bzx_launch_stream_frame(B, ctx->shard_level, ctx->d_scalars, ctx->d_scalars + 1, ctx->stream);
    if (stream_len) *stream_len = (size_t)out_bytes;
    return BZX_OK;
}

extern "C" int bzx_shard_assemble_rank(bzx_ctx *ctx, const void *d_packed_r, uint32_t r, void *d_out)
{
    std::unique_lock<std::recursive_mutex> api_lock_;
    if (ctx) api_lock_ = std::unique_lock<std::recursive_mutex>(ctx->api_mu);
    if (!ctx || !d_packed_r || !d_out || ((uintptr_t)d_packed_r & 3u) || r >= ctx->shard_world) return BZX_E_PARAM;
    if (ctx->shard_level == 0) return BZX_E_STATE;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    BzxBatch &B = ctx->B;
    const uint32_t nblk = ctx->shard_total, world = ctx->shard_world;
    const uint32_t nown = shard_count(nblk, r, world);
    B.nblk = nblk;
    B.out = (uint32_t *)d_out;
    if (nown == 0) return BZX_OK;
    bzx_launch_pack_layout(B, r, world, nown, ctx->d_scalars + 3, ctx->stream);            // rank r's packed positions
    bzx_launch_unpack(B, (const uint32_t *)d_packed_r, r, world, nown, grid_for(ctx, nown, 4), ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    return BZX_OK;
}

// ---- decompression (include/bzx.h: bzx_decompress_*; kernels in bzx_decomp.hip) ----------------------------------
#define DC_MAX_FOUND 262144u

// One bzip2 stream at the start of d_bz2[0..len); *consumed = bytes up to and including its footer.
static int decompress_one(bzx_ctx *ctx, const void *d_bz2, size_t len, void *d_out, size_t cap, size_t *out_len, size_t *consumed)
{
    *consumed = 0;
    if (!ctx->use_bsort) {
        ctx->err = "decompression needs the bucket sorter's buffers (BZX_SORTER=general is set)";
        return BZX_E_STATE;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *out_len = 0;
    const uint8_t *z = (const uint8_t *)d_bz2;
    uint8_t head[4] = {0, 0, 0, 0};
    if (len < 14) {
        ctx->err = "shorter than the smallest bzip2 stream";
        return BZX_E_DATA;
    }
    HIP_TRY(ctx, hipMemcpy(head, z, 4, hipMemcpyDeviceToHost));
    if (head[0] != 'B' || head[1] != 'Z' || head[2] != 'h' || head[3] < '1' || head[3] > '9') {
        ctx->err = "no BZh1..BZh9 header";
        return BZX_E_DATA;
    }
    const uint32_t max_n = 100000u * (uint32_t)(head[3] - '0');
    // ---- scan for block / end-of-stream magics at every bit offset
    void *scratch = nullptr;
    int rc = bzx_ctx_split_scratch(ctx, (size_t)DC_MAX_FOUND * 8 * 3 + 4096, &scratch);
    if (rc) return rc;
    uint64_t *d_found = (uint64_t *)scratch;
    uint64_t *d_starts = d_found + DC_MAX_FOUND;
    uint64_t *d_off = d_starts + DC_MAX_FOUND;
    uint32_t *d_nfound = (uint32_t *)(d_off + DC_MAX_FOUND);
    HIP_TRY(ctx, hipMemsetAsync(d_nfound, 0, 64, ctx->stream));
    bzx_launch_dc_scan(z, len, d_found, d_nfound, DC_MAX_FOUND, (uint32_t)ctx->n_cu * 8, ctx->stream);
    uint32_t nfound = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&nfound, d_nfound, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (nfound > DC_MAX_FOUND) {
        ctx->err = "too many block-magic candidates";
        return BZX_E_DATA;
    }
    std::vector<uint64_t> found, starts;
    try {
        found.resize(nfound);
        if (nfound) HIP_TRY(ctx, hipMemcpy(found.data(), d_found, (size_t)nfound * 8, hipMemcpyDeviceToHost));
        std::sort(found.begin(), found.end());
        for (uint64_t f : found)
            if (!(f & 1u)) starts.push_back(f >> 1);
    } catch (const std::bad_alloc &) {
        return BZX_E_NOMEM;
    }
    auto is_eos = [&](uint64_t bit) { return std::binary_search(found.begin(), found.end(), (bit << 1) | 1u); };
    // ---- decode every candidate; keep the chain that starts at bit 32 (a chance match of the magic inside compressed
    // data does not continue the chain: drop it and decode again without it)
    uint64_t end_bit = 32;
    uint32_t nblk = 0;
    for (int attempt = 0;; attempt++) {
        nblk = (uint32_t)starts.size();
        end_bit = 32;
        if (nblk == 0) break;
        if ((rc = ensure_blocks(ctx, nblk))) return rc;
        BzxBatch &B = ctx->B;
        B.nblk = nblk;
        B.blk_first = 0;
        B.blk_step = 1;
        HIP_TRY(ctx, hipMemcpyAsync(d_starts, starts.data(), (size_t)nblk * 8, hipMemcpyHostToDevice, ctx->stream));
        bzx_launch_dc_decode(B, z, len, d_starts, max_n, ctx->stream);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_blk, B.blk, (size_t)nblk * sizeof(BzxBlock), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<uint64_t> chain;
        bool clean = true;
        uint32_t i = 0;
        while (i < nblk) {
            if (starts[i] != end_bit) {             // not where the previous block ended: a chance match
                clean = false;
                i++;
                continue;
            }
            if (ctx->h_blk[i].status) {
                ctx->err = "damaged block in the bzip2 stream";
                return BZX_E_DATA;
            }
            chain.push_back(starts[i]);
            end_bit = ctx->h_blk[i].bits;
            i++;
        }
        if (clean) break;
        if (attempt >= 3) {
            ctx->err = "cannot follow the chain of blocks";
            return BZX_E_DATA;
        }
        starts.swap(chain);
    }
    if (!is_eos(end_bit)) {
        ctx->err = "blocks do not end at an end-of-stream marker";
        return BZX_E_DATA;
    }
    uint8_t foot[16] = {0};
    {
        const size_t fb = (size_t)((end_bit + 48) >> 3);
        const size_t nfb = len - fb < 5 ? len - fb : 5;
        if ((end_bit + 80 + 7) / 8 > len) {
            ctx->err = "truncated after the end-of-stream marker";
            return BZX_E_DATA;
        }
        HIP_TRY(ctx, hipMemcpy(foot, z + fb, nfb, hipMemcpyDeviceToHost));
    }
    uint64_t fv = 0;
    for (int i = 0; i < 5; i++) fv = (fv << 8) | foot[i];
    const uint32_t stream_crc = (uint32_t)((fv << ((end_bit + 48) & 7u)) >> 8);
    uint64_t total = 0;
    uint32_t comb = 0;
    if (nblk) {
        BzxBatch &B = ctx->B;
        // ---- inverse BWT, expanded sizes, offsets
        bzx_launch_dc_ibwt(B, ctx->d_in, ctx->stream);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_blk, B.blk, (size_t)nblk * sizeof(BzxBlock), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<uint64_t> off;
        try {
            off.resize((size_t)nblk + 1);
        } catch (const std::bad_alloc &) {
            return BZX_E_NOMEM;
        }
        for (uint32_t b = 0; b < nblk; b++) {
            if (ctx->h_blk[b].status) {
                ctx->err = "damaged block in the bzip2 stream (inverse BWT)";
                return BZX_E_DATA;
            }
            off[b] = total;
            total += ctx->h_blk[b].pack_word;
            comb = ((comb << 1) | (comb >> 31)) ^ ctx->h_blk[b].crc;       // stored CRCs (crc.rs:25-27)
        }
        off[nblk] = total;
        *out_len = (size_t)total;
        if (total > cap) {
            ctx->err = "output buffer too small for the decompressed data";
            return BZX_E_OUTBUF;
        }
        HIP_TRY(ctx, hipMemcpyAsync(d_off, off.data(), ((size_t)nblk + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        bzx_launch_dc_expand(B, ctx->d_in, d_off, (uint8_t *)d_out, total, ctx->stream);
        // ---- block CRCs of the output (the compressor's CRC kernel), against the stored ones
        std::vector<uint32_t> stored(nblk);
        for (uint32_t b = 0; b < nblk; b++) stored[b] = ctx->h_blk[b].crc;
        HIP_TRY(ctx, hipMemcpyAsync(d_nfound, &nblk, 4, hipMemcpyHostToDevice, ctx->stream));
        bzx_launch_block_crcs(ctx, (const uint8_t *)d_out, d_off, d_nfound, B.blk, nblk);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_blk, B.blk, (size_t)nblk * sizeof(BzxBlock), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipGetLastError());
        for (uint32_t b = 0; b < nblk; b++) {
            if (ctx->h_blk[b].crc != stored[b]) {
                ctx->err = "block CRC mismatch in block " + std::to_string(b);
                return BZX_E_DATA;
            }
        }
    }
    if (comb != stream_crc) {
        ctx->err = "combined CRC mismatch";
        return BZX_E_DATA;
    }
    *out_len = (size_t)total;
    *consumed = (size_t)((end_bit + 80 + 7) / 8);
    ctx->stats.nblk = nblk;
    ctx->stats.raw_bytes = total;
    return BZX_OK;
}

// true when another stream header (BZh1..BZh9) starts at d_bz2[at]
static bool stream_follows(bzx_ctx *ctx, const void *d_bz2, size_t len, size_t at)
{
    uint8_t h[4] = {0, 0, 0, 0};
    if (at + 14 > len) return false;
    if (hipMemcpy(h, (const uint8_t *)d_bz2 + at, 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
    return h[0] == 'B' && h[1] == 'Z' && h[2] == 'h' && h[3] >= '1' && h[3] <= '9';
}

// Device buffer -> device buffer: ONE stream (the reference's decompress() also stops at the first footer,
// decompress.rs:81-95).  Bytes behind the footer that are not another stream are ignored, as bzip2 does ("trailing
// garbage"); a concatenated .bz2 (pbzip2 output, cat a.bz2 b.bz2) is refused here rather than decoded in part --
// bzx_decompress_buffer decodes every stream of it.
extern "C" int bzx_decompress_device(bzx_ctx *ctx, const void *d_bz2, size_t len, void *d_out, size_t cap, size_t *out_len)
{
    std::unique_lock<std::recursive_mutex> api_lock_;
    if (ctx) api_lock_ = std::unique_lock<std::recursive_mutex>(ctx->api_mu);
    if (!ctx || !d_bz2 || !out_len || (cap && !d_out) || ((uintptr_t)d_out & 15u)) return BZX_E_PARAM;
    size_t used = 0;
    const int rc = decompress_one(ctx, d_bz2, len, d_out, cap, out_len, &used);
    if (rc == BZX_OK && stream_follows(ctx, d_bz2, len, used)) {
        ctx->err = "another bzip2 stream follows the first (concatenated .bz2): bzx_decompress_buffer decodes all of them";
        return BZX_E_DATA;
    }
    return rc;
}

extern "C" int bzx_decompress_buffer(bzx_ctx *ctx, const uint8_t *bz2, size_t len, uint8_t *out, size_t cap, size_t *out_len)
{
    std::unique_lock<std::recursive_mutex> api_lock_;
    if (ctx) api_lock_ = std::unique_lock<std::recursive_mutex>(ctx->api_mu);
    if (!ctx || !bz2 || !out_len || (cap && !out)) return BZX_E_PARAM;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void *d_z = nullptr, *d_o = nullptr;
    if (hipMalloc(&d_z, len + 64) != hipSuccess) return BZX_E_NOMEM;
    if (hipMalloc(&d_o, cap + 64) != hipSuccess) {
        (void)hipFree(d_z);
        return BZX_E_NOMEM;
    }
    int rc = hipMemcpyAsync(d_z, bz2, len, hipMemcpyHostToDevice, ctx->stream) == hipSuccess ? BZX_OK : BZX_E_HIP;
    // every stream of a concatenated .bz2, one after the other (each stream starts on a byte boundary)
    size_t at = 0, total = 0;
    uint32_t nblk_all = 0;
    *out_len = 0;
    void *d_z2 = nullptr;                          // a later stream, moved to an aligned start (the kernels read words)
    while (!rc) {
        size_t n = 0, used = 0;
        const void *src = d_z;
        if (at) {
            if (!d_z2 && hipMalloc(&d_z2, len + 64) != hipSuccess) {
                rc = BZX_E_NOMEM;
                break;
            }
            if (hipMemcpyAsync(d_z2, (const uint8_t *)d_z + at, len - at, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
                rc = BZX_E_HIP;
                break;
            }
            src = d_z2;
        }
        rc = decompress_one(ctx, src, len - at, d_o, cap - total, &n, &used);
        if (rc == BZX_E_OUTBUF) *out_len = total + n;          // (a lower bound when streams remain)
        if (rc) break;
        if (n && hipMemcpy(out + total, d_o, n, hipMemcpyDeviceToHost) != hipSuccess) rc = BZX_E_HIP;
        total += n;
        nblk_all += ctx->stats.nblk;
        at += used;
        *out_len = total;
        if (!stream_follows(ctx, d_z, len, at)) break;
    }
    if (!rc) {
        ctx->stats.nblk = nblk_all;
        ctx->stats.raw_bytes = total;
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_z);
    if (d_z2) (void)hipFree(d_z2);
    (void)hipFree(d_o);
    return rc;
}

// RLE1Block::new(source, block_size) + Iterator::next (rle1.rs:49-85,245-263) for a source that arrives in pieces:
// every call returns the blocks that are complete with the bytes seen so far; the last, unfinished block is withheld
// (its raw bytes are kept in the context) and comes out of a later call, or of the call with final != 0.
extern "C" int bzx_split_rle1_chunk(bzx_ctx *ctx, const uint8_t *raw, size_t len, int level, int final, uint8_t *blocks_out,
                                    uint32_t nblk_cap, uint32_t *ns, uint32_t *crcs, uint32_t *nblk_out)
{
    std::unique_lock<std::recursive_mutex> api_lock_;
    if (ctx) api_lock_ = std::unique_lock<std::recursive_mutex>(ctx->api_mu);
    if (!ctx || !blocks_out || !ns || !crcs || !nblk_out || !level_ok(level) || (len && !raw)) return BZX_E_PARAM;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *nblk_out = 0;
    int rc = BZX_OK;
    std::vector<uint8_t> &carry = ctx->split_carry;
    const size_t total = carry.size() + len;
    if (total == 0) return BZX_OK;
    void *d_raw = nullptr;
    if (hipMalloc(&d_raw, total) != hipSuccess) return BZX_E_NOMEM;
    uint32_t nblk = 0, use = 0;
    uint64_t last_start = 0;
    if (!carry.empty() && hipMemcpyAsync(d_raw, carry.data(), carry.size(), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = BZX_E_HIP;
    if (!rc && len && hipMemcpyAsync((uint8_t *)d_raw + carry.size(), raw, len, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = BZX_E_HIP;
    if (!rc) rc = split_on_device(ctx, (const uint8_t *)d_raw, total, level, &nblk, 0, 1, &last_start);
    if (!rc) {
        use = final ? nblk : nblk - 1;
        if (use > nblk_cap) rc = BZX_E_OUTBUF;
    }
    if (!rc && use) {
        if (hipMemcpyAsync(ctx->h_blk, ctx->B.blk, use * sizeof(BzxBlock), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            rc = BZX_E_HIP;
    }
    for (uint32_t b = 0; !rc && b < use; b++) {
        ns[b] = ctx->h_blk[b].n;
        crcs[b] = ctx->h_blk[b].crc;
        const uint64_t off = ctx->h_blk[b].in_off;
        const uint8_t *src = (off & BZX_IN_RAW) ? (const uint8_t *)d_raw + (off & ~BZX_IN_RAW) : ctx->d_in + off;
        if (ns[b] == 0 || ns[b] > BZX_MAX_BLOCK ||
            hipMemcpy(blocks_out + (size_t)b * BZX_MAX_BLOCK, src, ns[b], hipMemcpyDeviceToHost) != hipSuccess)
            rc = BZX_E_HIP;
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_raw);
    if (rc) return rc;
    // the withheld block's raw bytes: [last_start, total) of (carry | raw)
    std::vector<uint8_t> next;
    if (!final) {
        const size_t ls = (size_t)last_start, cs = carry.size();
        try {
            next.reserve(total - ls);
            if (ls < cs) next.insert(next.end(), carry.begin() + ls, carry.end());
            const size_t from = ls > cs ? ls - cs : 0;
            if (len > from) next.insert(next.end(), raw + from, raw + len);
        } catch (const std::bad_alloc &) {
            return BZX_E_NOMEM;
        }
    }
    carry.swap(next);
    *nblk_out = use;
    return BZX_OK;
}

// Page-locked host memory for the callers' buffers (copies from/to pageable memory are staged by the runtime and
// cannot overlap the kernels).
extern "C" void *bzx_host_alloc(size_t bytes)
{
    void *p = nullptr;
    return hipHostMalloc(&p, bytes ? bytes : 1, 0) == hipSuccess ? p : nullptr;
}
extern "C" void bzx_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

// ---- chunked stream compressor (include/bzx.h: bzx_cstream_*) -----------------------------------------------------
// The reference's driver reads the input incrementally (RLE1Block<R: Read>, rle1.rs:49-85,245-263), overlaps block
// production, compression and an ordered writer thread (compress.rs:66-132, bitwriter.rs:77-132).  Here the input
// arrives in CHUNKS: chunk k is copied to the device while chunk k-1 is compressed and the output of chunk k-2..k-1
// travels back, on three HIP streams with double buffers.  Block boundaries depend on the whole stream before them
// (SURVEY.md D1); a chunk is therefore split as "the raw bytes of the last, unfinished block of the previous chunk
// + the new bytes": the splitter's state is clean at a block start (a block is a whole number of run pieces), so
// restarting there reproduces exactly the blocks a one-shot split would cut.  All blocks but the last of a chunk
// are compressed; the last one is withheld until more input (or `final`) arrives.  Chunk outputs are bit-contiguous:
// the bit phase travels on the device (bzx_layout_kernel), the shared boundary word is OR-merged on the host, header
// and footer (+ combined CRC, crc.rs:25-27) are written by the host.
struct bzx_cstream {
    bzx_ctx *ctx = nullptr;
    int level = 9;
    size_t max_chunk = 0, in_cap = 0, out_cap = 0;
    uint8_t *d_in[2] = {nullptr, nullptr};
    uint32_t *d_out[2] = {nullptr, nullptr};
    uint64_t *d_phase = nullptr;            // [0] bit phase of the next chunk, [1] bits of the last laid-out chunk
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr}, ev_d2h = nullptr;
    uint64_t *h_info[2] = {nullptr, nullptr};     // pinned: {phase in, bits} of the chunk emitted into d_out[slot]
    uint32_t *h_w0 = nullptr;                     // pinned: first word of a chunk's output (shared with its predecessor)
    BzxBlock *h_blk[2] = {nullptr, nullptr};      // pinned: descriptors of the chunk's blocks (CRCs)
    uint32_t blk_cap = 0;
    uint32_t k = 0;                         // chunks fed
    size_t carry_len = 0, carry_start = 0;  // raw bytes of the withheld block inside d_in[(k-1)&1]
    uint64_t bits = 32;                     // stream bits accounted for so far (header included)
    uint32_t crc_comb = 0;
    uint64_t nblk_total = 0, st_rle1 = 0, st_mtf = 0, st_raw = 0;
    uint32_t st_per = 0;
    bool pend = false;                      // a chunk's output still sits in d_out[pend_slot]
    bool coll_issued = false;               // ... and its copy-back has been enqueued (cstream_collect), not yet awaited
    uint32_t pend_slot = 0, pend_nblk = 0;
    bool finished = false;
    uint8_t *out = nullptr;
    size_t cap = 0;
    size_t need_hint = 0;                   // after BZX_E_OUTBUF: bytes the output needs at least
};

extern "C" void bzx_cstream_end(bzx_cstream *s)
{
    if (!s) return;
    if (s->ctx) (void)hipSetDevice(s->ctx->device);
    if (s->ctx) (void)hipStreamSynchronize(s->ctx->stream);
    if (s->s_h2d) (void)hipStreamSynchronize(s->s_h2d);
    if (s->s_d2h) (void)hipStreamSynchronize(s->s_d2h);
    for (int i = 0; i < 2; i++) {
        if (s->d_in[i]) (void)hipFree(s->d_in[i]);
        if (s->d_out[i]) (void)hipFree(s->d_out[i]);
        if (s->ev_h2d[i]) (void)hipEventDestroy(s->ev_h2d[i]);
        if (s->ev_done[i]) (void)hipEventDestroy(s->ev_done[i]);
        if (s->h_info[i]) (void)hipHostFree(s->h_info[i]);
        if (s->h_blk[i]) (void)hipHostFree(s->h_blk[i]);
    }
    if (s->ev_d2h) (void)hipEventDestroy(s->ev_d2h);
    if (s->d_phase) (void)hipFree(s->d_phase);
    if (s->h_w0) (void)hipHostFree(s->h_w0);
    if (s->s_h2d) (void)hipStreamDestroy(s->s_h2d);
    if (s->s_d2h) (void)hipStreamDestroy(s->s_d2h);
    delete s;
}

// A block covers at most nblockMAX RLE1 bytes = nblockMAX / 5 runs of 255: the withheld raw tail never exceeds this.
static size_t cstream_max_carry(int level) { return ((size_t)100000 * level / 5 + 2) * 255 + 4096; }

extern "C" int bzx_cstream_begin(bzx_ctx *ctx, int level, size_t max_chunk, bzx_cstream **out)
{
    if (!ctx || !out || !level_ok(level)) return BZX_E_PARAM;
    *out = nullptr;
    std::unique_lock<std::recursive_mutex> api_lock_(ctx->api_mu);
    if (max_chunk == 0) max_chunk = (size_t)256 << 20;
    max_chunk = (max_chunk + 15) & ~(size_t)15;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    bzx_cstream *s = new (std::nothrow) bzx_cstream();
    if (!s) return BZX_E_NOMEM;
    s->ctx = ctx;
    s->level = level;
    s->max_chunk = max_chunk;
    // Sized for EVERY level, not the one given here: bzx_compress_buffer keeps the stream object in the context and
    // starts the next stream on it at whatever level its caller asks for (cstream_reset) -- the withheld raw tail is
    // longest at level 9, the blocks of a chunk are most numerous at level 1.
    s->in_cap = max_chunk + cstream_max_carry(9) + 256;
    s->out_cap = (s->in_cap + s->in_cap / 50 + 65536) & ~(size_t)255;       // RLE1 +25 % never survives coding: 2 % + slack
    s->out_cap += s->in_cap / 4;
    s->blk_cap = (uint32_t)((s->in_cap + s->in_cap / 4) / ((size_t)100000 * 1 - 19) + 4);
    bool ok = true;
    for (int i = 0; i < 2 && ok; i++) {
        ok = hipMalloc((void **)&s->d_in[i], s->in_cap) == hipSuccess && hipMalloc((void **)&s->d_out[i], s->out_cap) == hipSuccess &&
             hipEventCreateWithFlags(&s->ev_h2d[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&s->ev_done[i], hipEventDisableTiming) == hipSuccess &&
             hipHostMalloc((void **)&s->h_info[i], 4 * sizeof(uint64_t), 0) == hipSuccess &&
             hipHostMalloc((void **)&s->h_blk[i], (size_t)s->blk_cap * sizeof(BzxBlock), 0) == hipSuccess;
    }
    ok = ok && hipEventCreateWithFlags(&s->ev_d2h, hipEventDisableTiming) == hipSuccess &&
         hipMalloc((void **)&s->d_phase, 4 * sizeof(uint64_t)) == hipSuccess &&
         hipHostMalloc((void **)&s->h_w0, 16, 0) == hipSuccess &&
         hipStreamCreateWithFlags(&s->s_h2d, hipStreamNonBlocking) == hipSuccess &&
         hipStreamCreateWithFlags(&s->s_d2h, hipStreamNonBlocking) == hipSuccess &&
         hipMemsetAsync(s->d_phase, 0, 4 * sizeof(uint64_t), ctx->stream) == hipSuccess;
    if (!ok) {
        ctx->err = "bzx_cstream_begin: device or pinned allocation failed";
        bzx_cstream_end(s);
        return BZX_E_NOMEM;
    }
    *out = s;
    return BZX_OK;
}

// Brings the output of the chunk parked in d_out[pend_slot] to the caller's buffer (async on the copy-back stream)
// and accounts for its bits and block CRCs.  The chunk's layout has completed when this is called.
static int cstream_collect(bzx_cstream *s)
{
    bzx_ctx *ctx = s->ctx;
    if (!s->pend) return BZX_OK;
    const uint32_t slot = s->pend_slot;
    // h_info = {phase the NEXT chunk starts with, bits of this chunk}; this chunk started at the phase the host
    // accounting says
    const uint64_t phase = s->bits & 31u, cbits = s->h_info[slot][1];
    if (s->h_info[slot][0] != ((phase + cbits) & 31u)) {
        ctx->err = "chunked stream: bit phase out of step";
        return BZX_E_STATE;
    }
    const uint64_t nwords = (phase + cbits + 31) >> 5;
    const size_t off = (size_t)(s->bits >> 5) * 4;
    if (off + nwords * 4 > s->cap) {
        ctx->err = "output buffer too small for the compressed stream";
        s->need_hint = off + (size_t)((phase + cbits + 80 + 7) >> 3);
        return BZX_E_OUTBUF;
    }
    if (nwords * 4 > s->out_cap) {
        ctx->err = "chunk output larger than its device buffer";
        return BZX_E_HIP;
    }
    HIP_TRY(ctx, hipStreamWaitEvent(s->s_d2h, s->ev_done[slot], 0));
    if (nwords) {
        HIP_TRY(ctx, hipMemcpyAsync(s->h_w0, s->d_out[slot], 4, hipMemcpyDeviceToHost, s->s_d2h));
        if (nwords > 1)
            HIP_TRY(ctx, hipMemcpyAsync(s->out + off + 4, s->d_out[slot] + 1, (nwords - 1) * 4, hipMemcpyDeviceToHost, s->s_d2h));
    }
    HIP_TRY(ctx, hipEventRecord(s->ev_d2h, s->s_d2h));
    s->coll_issued = true;
    return BZX_OK;
}

// Second half: waits for the copy-back issued by cstream_collect, merges the word the chunk shares with its
// predecessor and folds its block CRCs.  Called AFTER the next chunk's stages have been enqueued, so the copy-back of
// chunk k-1 runs beside the compression of chunk k.
static int cstream_collect_finish(bzx_cstream *s)
{
    bzx_ctx *ctx = s->ctx;
    if (!s->pend || !s->coll_issued) return BZX_OK;
    s->coll_issued = false;
    const uint32_t slot = s->pend_slot;
    const uint64_t phase = s->bits & 31u, cbits = s->h_info[slot][1];
    const uint64_t nwords = (phase + cbits + 31) >> 5;
    const size_t off = (size_t)(s->bits >> 5) * 4;
    HIP_TRY(ctx, hipEventSynchronize(s->ev_d2h));
    if (nwords) {
        // the first word is shared with the predecessor (or with nothing: then the bytes there are still zero)
        uint8_t w[4];
        memcpy(w, s->h_w0, 4);
        if (phase == 0) memcpy(s->out + off, w, 4);
        else for (int i = 0; i < 4; i++) s->out[off + i] |= w[i];
    }
    for (uint32_t b = 0; b < s->pend_nblk; b++) {
        const BzxBlock &d = s->h_blk[slot][b];
        s->crc_comb = ((s->crc_comb << 1) | (s->crc_comb >> 31)) ^ d.crc;
        s->st_rle1 += d.n;
        s->st_mtf += d.n_mtf;
        s->st_per += (d.status & BZX_ST_PERIODIC) ? 1u : 0u;
        // (bzx_get_block_info: the stream's descriptors in order, as far as the context's descriptor table reaches)
        if (ctx->h_blk && s->nblk_total + b < ctx->cap_blocks) ctx->h_blk[s->nblk_total + b] = d;
    }
    s->nblk_total += s->pend_nblk;
    s->bits += cbits;
    s->pend = false;
    return BZX_OK;
}

// Back to the state after bzx_cstream_begin (buffers kept): a new stream on the same object.
static size_t cstream_chunk_of(const bzx_cstream *s) { return s->max_chunk; }
static size_t cstream_need_hint(const bzx_cstream *s) { return s->need_hint; }

static int cstream_reset(bzx_cstream *s, int level)
{
    bzx_ctx *ctx = s->ctx;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(s->s_h2d));
    HIP_TRY(ctx, hipStreamSynchronize(s->s_d2h));
    HIP_TRY(ctx, hipMemsetAsync(s->d_phase, 0, 4 * sizeof(uint64_t), ctx->stream));
    s->level = level;
    s->k = 0;
    s->carry_len = s->carry_start = 0;
    s->bits = 32;
    s->crc_comb = 0;
    s->nblk_total = s->st_rle1 = s->st_mtf = s->st_raw = 0;
    s->st_per = 0;
    s->pend = false;
    s->coll_issued = false;
    s->finished = false;
    return BZX_OK;
}

extern "C" int bzx_cstream_feed(bzx_cstream *s, const uint8_t *raw, size_t len, int final, uint8_t *out, size_t cap,
                                size_t *produced)
{
    if (!s || !s->ctx || !out || !produced || (len && !raw) || len > s->max_chunk || cap < 16) return BZX_E_PARAM;
    bzx_ctx *ctx = s->ctx;
    std::unique_lock<std::recursive_mutex> api_lock_(ctx->api_mu);
    if (s->finished) return BZX_E_STATE;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (s->k == 0) {
        memset(out, 0, cap < 64 ? cap : 64);
        out[0] = 'B'; out[1] = 'Z'; out[2] = 'h'; out[3] = (uint8_t)('0' + s->level);
    }
    s->out = out;
    s->cap = cap;
    const uint32_t slot = s->k & 1u;
    const size_t total = s->carry_len + len;
    if (total > s->in_cap) {                     // (cannot happen with the provisioning above; never write past d_in)
        ctx->err = "chunked stream: withheld bytes + chunk exceed the device input buffer";
        return BZX_E_STATE;
    }
    // the device buffer of this slot was last read by chunk k-2; its kernels are long done when k-1's results were
    // collected, but the copy stream does not know that: make it wait
    if (s->k >= 2) HIP_TRY(ctx, hipStreamWaitEvent(s->s_h2d, s->ev_done[slot], 0));
    if (len) {
        HIP_TRY(ctx, hipMemcpyAsync(s->d_in[slot] + s->carry_len, raw, len, hipMemcpyHostToDevice, s->s_h2d));
    }
    HIP_TRY(ctx, hipEventRecord(s->ev_h2d[slot], s->s_h2d));
    if (s->carry_len)     // the withheld block's raw bytes move to the front of this chunk (after the kernels that read them)
        HIP_TRY(ctx, hipMemcpyAsync(s->d_in[slot], s->d_in[slot ^ 1u] + s->carry_start, s->carry_len, hipMemcpyDeviceToDevice,
                                    ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, s->ev_h2d[slot], 0));
    uint32_t nblk = 0, use = 0;
    uint64_t last_start = 0;
    int rc = BZX_OK;
    if (total) {
        ctx->B.blk_first = 0;
        ctx->B.blk_step = 1;
        rc = split_on_device(ctx, s->d_in[slot], total, s->level, &nblk, 0, 1, &last_start);     // (synchronises)
        if (rc) return rc;
        use = final ? nblk : nblk - 1;
        if (use > s->blk_cap) {
            ctx->err = "chunked stream: more blocks in a chunk than provisioned";
            return BZX_E_HIP;
        }
    }
    // the previous chunk was laid out before this chunk's split ran: its sizes are on the host now
    if (!total) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // Chunk k's stages go into the queue FIRST; then the copy-back of chunk k-1 is issued on its own stream and awaited:
    // it runs beside the compression of chunk k (with a pageable destination the runtime stages the copy and blocks the
    // host while it lasts -- the device has its work by then).
    if (use) {
        if ((rc = run_stages(ctx, use, STG_ALL, -1, s->d_out[slot], s->out_cap, s->d_phase))) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(s->h_info[slot], s->d_phase, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(s->h_blk[slot], ctx->B.blk, (size_t)use * sizeof(BzxBlock), hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(ctx, hipEventRecord(s->ev_done[slot], ctx->stream));
    if ((rc = cstream_collect(s))) return rc;
    if ((rc = cstream_collect_finish(s))) return rc;
    if (use) {
        s->pend = true;
        s->pend_slot = slot;
        s->pend_nblk = use;
    }
    if (!final && total) {
        s->carry_start = (size_t)last_start;
        s->carry_len = total - (size_t)last_start;
    } else {
        s->carry_len = 0;
        s->carry_start = 0;
    }
    s->k++;
    if (final) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if ((rc = cstream_collect(s))) return rc;
        if ((rc = cstream_collect_finish(s))) return rc;
        collect_stage_times(ctx);
        // footer: magic, combined CRC (crc.rs:25-27), zero padding to a byte (bitwriter.rs:103-114,158-172)
        const uint64_t end = s->bits;
        const size_t need = (size_t)((end + 80 + 7) >> 3);
        if (need > cap) {
            ctx->err = "output buffer too small for the compressed stream";
            s->need_hint = need;
            return BZX_E_OUTBUF;
        }
        const uint8_t foot[10] = {0x17, 0x72, 0x45, 0x38, 0x50, 0x90, (uint8_t)(s->crc_comb >> 24), (uint8_t)(s->crc_comb >> 16),
                                  (uint8_t)(s->crc_comb >> 8), (uint8_t)s->crc_comb};
        const size_t ebyte = (size_t)(end >> 3);
        const uint32_t sh = (uint32_t)(end & 7u);
        // bytes from the end of the last word written on are untouched so far: clear, then OR the shifted footer in
        const size_t clear_from = (size_t)((end + 31) >> 5) * 4;
        for (size_t i = clear_from; i < need; i++) out[i] = 0;
        for (int i = 0; i < 10; i++) {
            out[ebyte + i] |= (uint8_t)(foot[i] >> sh);
            if (sh) out[ebyte + i + 1] |= (uint8_t)(foot[i] << (8 - sh));
        }
        *produced = need;
        s->finished = true;
        ctx->stats.nblk = (uint32_t)s->nblk_total;
        ctx->stats.n_periodic = s->st_per;
        ctx->stats.rle1_bytes = s->st_rle1;
        ctx->stats.mtf_symbols = s->st_mtf;
        ctx->stats.raw_bytes = s->st_raw + len;
        ctx->stats.out_bits = (uint64_t)need * 8;
        return BZX_OK;
    }
    s->st_raw += len;
    // bytes that can no longer change: everything before the word the next chunk starts in
    *produced = (size_t)(s->bits >> 5) * 4;
    return BZX_OK;
}

#ifdef BZX_DIAG
// Diagnostic build only (libbzx_diag.so, -DBZX_DIAG; never in libbzx.so): phase timers of the sort kernels.
// Debug helper (not in include/bzx.h): enable/read the BWT kernel's phase timers (100 MHz wall-clock ticks summed over blocks).
extern "C" int bzx_dbg_set_stop(bzx_ctx *ctx, uint32_t k)
{
    if (!ctx) return BZX_E_PARAM;
    ctx->B.dbg_stop = k;
    return BZX_OK;
}

extern "C" int bzx_dbg_phase_timers(bzx_ctx *ctx, int enable, unsigned long long out[128])
{
    if (!ctx) return BZX_E_PARAM;
    if (enable && !ctx->d_dbg) {
        if (hipMalloc((void **)&ctx->d_dbg, 128 * sizeof(unsigned long long)) != hipSuccess) return BZX_E_NOMEM;
    }
    if (out && ctx->d_dbg) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipMemcpy(out, ctx->d_dbg, 128 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    if (ctx->d_dbg) HIP_TRY(ctx, hipMemset(ctx->d_dbg, 0, 128 * sizeof(unsigned long long)));
    ctx->B.dbg = enable ? ctx->d_dbg : nullptr;
    return BZX_OK;
}

// Debug helper: per-block microseconds spent in the BWT kernel during the last run with phase timers enabled.
extern "C" int bzx_dbg_block_times(bzx_ctx *ctx, uint32_t nblk, uint32_t *us_out, uint32_t *n_out, uint32_t *inuse_out)
{
    if (!ctx || !us_out || nblk > ctx->cap_blocks) return BZX_E_PARAM;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(ctx->h_blk, ctx->B.blk, nblk * sizeof(BzxBlock), hipMemcpyDeviceToHost));
    for (uint32_t b = 0; b < nblk; b++) {
        us_out[b] = ctx->h_blk[b].pad_[1];
        if (n_out) n_out[b] = ctx->h_blk[b].n;
        if (inuse_out) inuse_out[b] = ctx->h_blk[b].n_in_use;
    }
    return BZX_OK;
}
#endif   // BZX_DIAG
