/*
 * oracle_driver.c -- TEST INFRASTRUCTURE (see bzx_oracle.h).
 *
 * Multi-threaded whole-buffer driver shaped like the reference's compress()
 * (src/compression/compress.rs:125-132: RLE1 blocks produced serially, one block per worker,
 * results re-ordered by sequence number, compress.rs:74-122) and the synthetic input
 * generators of SURVEY.md section 8(d).  Used as bench.py's reported-only cpu_baseline.
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include "bzx_oracle.h"

typedef struct {
    uint8_t *blk;
    size_t n;
    uint32_t crc;
    uint8_t *z;
    size_t zcap, zlen;
    uint8_t pad;
    int rc;
} job_t;

typedef struct {
    job_t *jobs;
    int32_t njobs;
    int32_t next;
    pthread_mutex_t mu;
} pool_t;

static void *worker(void *arg)
{
    pool_t *p = (pool_t *)arg;
    for (;;) {
        pthread_mutex_lock(&p->mu);
        int32_t i = p->next++;
        pthread_mutex_unlock(&p->mu);
        if (i >= p->njobs) break;
        job_t *j = &p->jobs[i];
        j->z = (uint8_t *)malloc(j->zcap);
        j->rc = bzo_compress_block(j->blk, j->n, j->crc, j->z, j->zcap, &j->zlen, &j->pad);
    }
    return NULL;
}

size_t bzo_compress_buffer_mt(const uint8_t *raw, size_t len, int level, int nthreads, uint8_t *out, size_t cap,
                              int32_t *nblocks_out)
{
    if (nthreads < 1) nthreads = 1;
    size_t bcap = (size_t)100000 * level;
    size_t max_jobs = len / (bcap - 19) * 52 / 50 + 8; /* RLE1 never expands a block's raw coverage below nmax/1.25 */
    /* conservative: RLE1 output is at most 5/4 of the raw bytes */
    max_jobs = (len + len / 4) / (bcap - 19) + 8;
    job_t *jobs = (job_t *)calloc(max_jobs, sizeof(job_t));
    uint32_t st[2] = {256, 0};
    size_t pos = 0;
    int32_t nb = 0;

    /* serial RLE1 pass, as under par_bridge's iterator lock (compress.rs:125-128) */
    while (pos < len || st[0] < 256) {
        job_t *j = &jobs[nb];
        j->blk = (uint8_t *)malloc(bcap + 8);
        j->n = bzo_rle1_block(raw, len, &pos, level, st, j->blk, &j->crc);
        if (j->n == 0) {
            free(j->blk);
            break;
        }
        j->zcap = j->n + j->n / 50 + 1024;
        nb++;
    }

    pool_t pool;
    pool.jobs = jobs;
    pool.njobs = nb;
    pool.next = 0;
    pthread_mutex_init(&pool.mu, NULL);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, worker, &pool);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th);
    pthread_mutex_destroy(&pool.mu);

    bzo_stream s;
    bzo_stream_begin(&s, out, cap, level);
    for (int32_t i = 0; i < nb; i++) {
        if (jobs[i].rc != 0) s.overflow = 1;
        else bzo_stream_add_block(&s, jobs[i].z, jobs[i].zlen, jobs[i].pad);
        free(jobs[i].blk);
        free(jobs[i].z);
    }
    size_t total = bzo_stream_finish(&s);
    free(jobs);
    if (nblocks_out) *nblocks_out = nb;
    return total;
}

/* ------------------------------------------------------------------ synthetic inputs */
#include "../include/bzx_synth.h"

void bzo_synthtext(uint64_t seed, uint8_t *out, size_t nbytes) { bzx_synth_text_impl(seed, out, nbytes); }
void bzo_xorshift_bytes(uint64_t seed, uint8_t *out, size_t nbytes) { bzx_synth_random_impl(seed, out, nbytes); }
