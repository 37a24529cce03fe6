// bzx_periodic.hip -- origPtr of PERIODIC blocks on gfx950.
//
// A block u^k (k > 1) has k identical copies of every rotation.  The last column L does not depend
// on how ties are ordered, but the 24-bit origPtr (row of rotation 0) does, and the metric is
// bit-exactness with C bzip2 1.0.8, whose tie order falls out of its block sorter's internals
// (the reference leaves it unspecified: src/bwt_algorithms/bwt_sort.rs:40-42 sort_unstable).
// bzx_bwt.hip flags such blocks (BZX_ST_PERIODIC).  For them only, this kernel replays the
// published libbz2 1.0.8 block-sorting algorithm literally -- "main" sort with its work budget of
// 9*nblock, "fallback" sort (bucket + doubling with LCG-pivot 3-way quicksort) when the budget runs
// out or nblock < 10000 -- on ONE lane per block, entirely on the device, and takes origPtr from it.
// Exactly periodic blocks are vanishingly rare outside all-equal-byte inputs (config 5a: 5 blocks),
// so this path is about exactness, not speed; it is the only serial code in the pipeline.
#include <hip/hip_runtime.h>
#include "bzx_device.h"
#include "bzx_wg.h"

#define N_RADIX 2
#define N_QSORT 12
#define N_SHELL 18
#define N_OVERSHOOT (N_RADIX + N_QSORT + N_SHELL + 2)

/* ------------------------------------------------------------------ fallback sort */

__device__ static void fb_simple_sort(uint32_t *fmap, const uint32_t *eclass, int32_t lo, int32_t hi)
{
    int32_t i, j;
    uint32_t tmp, ec;
    if (lo == hi) return;
    if (hi - lo > 3) {
        for (i = hi - 4; i >= lo; i--) {
            tmp = fmap[i];
            ec = eclass[tmp];
            for (j = i + 4; j <= hi && ec > eclass[fmap[j]]; j += 4) fmap[j - 4] = fmap[j];
            fmap[j - 4] = tmp;
        }
    }
    for (i = hi - 1; i >= lo; i--) {
        tmp = fmap[i];
        ec = eclass[tmp];
        for (j = i + 1; j <= hi && ec > eclass[fmap[j]]; j++) fmap[j - 1] = fmap[j];
        fmap[j - 1] = tmp;
    }
}

__device__ static inline void swap_u32(uint32_t *a, uint32_t *b)
{
    uint32_t t = *a;
    *a = *b;
    *b = t;
}

__device__ static void vswap(uint32_t *p, int32_t a, int32_t b, int32_t n)
{
    while (n > 0) {
        swap_u32(&p[a], &p[b]);
        a++;
        b++;
        n--;
    }
}

#define FB_SMALL 10
#define STACK_SZ 100

__device__ static void fb_qsort3(uint32_t *fmap, const uint32_t *eclass, int32_t lo_st, int32_t hi_st)
{
    int32_t un_lo, un_hi, lt_lo, gt_hi, n, m, sp, lo, hi;
    uint32_t med, r, r3;
    int32_t stack_lo[STACK_SZ], stack_hi[STACK_SZ];

    r = 0;
    sp = 0;
    stack_lo[sp] = lo_st;
    stack_hi[sp] = hi_st;
    sp++;

    while (sp > 0) {
        sp--;
        lo = stack_lo[sp];
        hi = stack_hi[sp];
        if (hi - lo < FB_SMALL) {
            fb_simple_sort(fmap, eclass, lo, hi);
            continue;
        }
        r = ((r * 7621) + 1) % 32768;
        r3 = r % 3;
        if (r3 == 0)
            med = eclass[fmap[lo]];
        else if (r3 == 1)
            med = eclass[fmap[(lo + hi) >> 1]];
        else
            med = eclass[fmap[hi]];

        un_lo = lt_lo = lo;
        un_hi = gt_hi = hi;
        for (;;) {
            for (;;) {
                if (un_lo > un_hi) break;
                n = (int32_t)eclass[fmap[un_lo]] - (int32_t)med;
                if (n == 0) {
                    swap_u32(&fmap[un_lo], &fmap[lt_lo]);
                    lt_lo++;
                    un_lo++;
                    continue;
                }
                if (n > 0) break;
                un_lo++;
            }
            for (;;) {
                if (un_lo > un_hi) break;
                n = (int32_t)eclass[fmap[un_hi]] - (int32_t)med;
                if (n == 0) {
                    swap_u32(&fmap[un_hi], &fmap[gt_hi]);
                    gt_hi--;
                    un_hi--;
                    continue;
                }
                if (n < 0) break;
                un_hi--;
            }
            if (un_lo > un_hi) break;
            swap_u32(&fmap[un_lo], &fmap[un_hi]);
            un_lo++;
            un_hi--;
        }
        if (gt_hi < lt_lo) continue;

        n = (lt_lo - lo) < (un_lo - lt_lo) ? (lt_lo - lo) : (un_lo - lt_lo);
        vswap(fmap, lo, un_lo - n, n);
        m = (hi - gt_hi) < (gt_hi - un_hi) ? (hi - gt_hi) : (gt_hi - un_hi);
        vswap(fmap, un_lo, hi - m + 1, m);

        n = lo + un_lo - lt_lo - 1;
        m = hi - (gt_hi - un_hi) + 1;

        if (n - lo > hi - m) {
            stack_lo[sp] = lo; stack_hi[sp] = n; sp++;
            stack_lo[sp] = m; stack_hi[sp] = hi; sp++;
        } else {
            stack_lo[sp] = m; stack_hi[sp] = hi; sp++;
            stack_lo[sp] = lo; stack_hi[sp] = n; sp++;
        }
    }
}

#define SET_BH(z) bhtab[(z) >> 5] |= ((uint32_t)1 << ((z) & 31))
#define CLEAR_BH(z) bhtab[(z) >> 5] &= ~((uint32_t)1 << ((z) & 31))
#define ISSET_BH(z) (bhtab[(z) >> 5] & ((uint32_t)1 << ((z) & 31)))
#define WORD_BH(z) bhtab[(z) >> 5]
#define UNALIGNED_BH(z) ((z) & 0x1f)

/* ------------------------------------------------------------------ main sort */

// Work-budget shortcut (exact): while no quadrant descriptor has been written yet (every quadrant is 0) two
// IDENTICAL rotations (i1 == i2 mod period) compare equal on every byte and every quadrant, so libbz2's loop
// runs its full (nblock+8)/8 + 1 iterations, charges that many budget units and reports "not greater".
struct PeriodInfo {
    uint32_t period;        // block = u^k, period = |u|
    int quadrant_clean;     // no quadrant[] entry has been set to a non-zero value yet
};

__device__ static inline int main_gtu(uint32_t i1, uint32_t i2, const uint8_t *block, const uint16_t *quadrant,
                           uint32_t nblock, int32_t *budget, const PeriodInfo *pi)
{
    int32_t k;
    uint8_t c1, c2;
    uint16_t s1, s2;
    int t;

    if (pi->quadrant_clean && (i1 % pi->period) == (i2 % pi->period)) {
        *budget -= (int32_t)((nblock + 8) / 8 + 1);
        return 0;
    }
    for (t = 0; t < 12; t++) {
        c1 = block[i1];
        c2 = block[i2];
        if (c1 != c2) return c1 > c2;
        i1++;
        i2++;
    }
    k = (int32_t)nblock + 8;
    do {
        for (t = 0; t < 8; t++) {
            c1 = block[i1];
            c2 = block[i2];
            if (c1 != c2) return c1 > c2;
            s1 = quadrant[i1];
            s2 = quadrant[i2];
            if (s1 != s2) return s1 > s2;
            i1++;
            i2++;
        }
        if (i1 >= nblock) i1 -= nblock;
        if (i2 >= nblock) i2 -= nblock;
        k -= 8;
        (*budget)--;
    } while (k >= 0);
    return 0;
}

__device__ static const int32_t shell_incs[14] = {1, 4, 13, 40, 121, 364, 1093, 3280, 9841, 29524, 88573, 265720, 797161, 2391484};

__device__ static void main_simple_sort(uint32_t *ptr, const uint8_t *block, const uint16_t *quadrant, int32_t nblock,
                             int32_t lo, int32_t hi, int32_t d, int32_t *budget, const PeriodInfo *pi)
{
    int32_t i, j, h, big_n, hp, rep;
    uint32_t v;

    big_n = hi - lo + 1;
    if (big_n < 2) return;
    hp = 0;
    while (shell_incs[hp] < big_n) hp++;
    hp--;

    for (; hp >= 0; hp--) {
        h = shell_incs[hp];
        i = lo + h;
        for (;;) {
            /* three insertions between budget checks */
            for (rep = 0; rep < 3; rep++) {
                if (i > hi) break;
                v = ptr[i];
                j = i;
                while (main_gtu(ptr[j - h] + d, v + d, block, quadrant, (uint32_t)nblock, budget, pi)) {
                    ptr[j] = ptr[j - h];
                    j = j - h;
                    if (j <= (lo + h - 1)) break;
                }
                ptr[j] = v;
                i++;
            }
            if (rep < 3) break;
            if (*budget < 0) return;
        }
    }
}

__device__ static inline uint8_t med3(uint8_t a, uint8_t b, uint8_t c)
{
    uint8_t t;
    if (a > b) { t = a; a = b; b = t; }
    if (b > c) {
        b = c;
        if (a > b) b = a;
    }
    return b;
}

#define MAIN_SMALL 20
#define MAIN_DEPTH (N_RADIX + N_QSORT)

__device__ static void main_qsort3(uint32_t *ptr, const uint8_t *block, const uint16_t *quadrant, int32_t nblock,
                        int32_t lo_st, int32_t hi_st, int32_t d_st, int32_t *budget, const PeriodInfo *pi)
{
    int32_t un_lo, un_hi, lt_lo, gt_hi, n, m, med, sp, lo, hi, d;
    int32_t stack_lo[STACK_SZ], stack_hi[STACK_SZ], stack_d[STACK_SZ];
    int32_t next_lo[3], next_hi[3], next_d[3], tz;

    sp = 0;
    stack_lo[sp] = lo_st; stack_hi[sp] = hi_st; stack_d[sp] = d_st; sp++;

    while (sp > 0) {
        sp--;
        lo = stack_lo[sp]; hi = stack_hi[sp]; d = stack_d[sp];
        if (hi - lo < MAIN_SMALL || d > MAIN_DEPTH) {
            main_simple_sort(ptr, block, quadrant, nblock, lo, hi, d, budget, pi);
            if (*budget < 0) return;
            continue;
        }
        med = (int32_t)med3(block[ptr[lo] + d], block[ptr[hi] + d], block[ptr[(lo + hi) >> 1] + d]);

        un_lo = lt_lo = lo;
        un_hi = gt_hi = hi;
        for (;;) {
            for (;;) {
                if (un_lo > un_hi) break;
                n = ((int32_t)block[ptr[un_lo] + d]) - med;
                if (n == 0) {
                    swap_u32(&ptr[un_lo], &ptr[lt_lo]);
                    lt_lo++;
                    un_lo++;
                    continue;
                }
                if (n > 0) break;
                un_lo++;
            }
            for (;;) {
                if (un_lo > un_hi) break;
                n = ((int32_t)block[ptr[un_hi] + d]) - med;
                if (n == 0) {
                    swap_u32(&ptr[un_hi], &ptr[gt_hi]);
                    gt_hi--;
                    un_hi--;
                    continue;
                }
                if (n < 0) break;
                un_hi--;
            }
            if (un_lo > un_hi) break;
            swap_u32(&ptr[un_lo], &ptr[un_hi]);
            un_lo++;
            un_hi--;
        }
        if (gt_hi < lt_lo) {
            stack_lo[sp] = lo; stack_hi[sp] = hi; stack_d[sp] = d + 1; sp++;
            continue;
        }
        n = (lt_lo - lo) < (un_lo - lt_lo) ? (lt_lo - lo) : (un_lo - lt_lo);
        vswap(ptr, lo, un_lo - n, n);
        m = (hi - gt_hi) < (gt_hi - un_hi) ? (hi - gt_hi) : (gt_hi - un_hi);
        vswap(ptr, un_lo, hi - m + 1, m);

        n = lo + un_lo - lt_lo - 1;
        m = hi - (gt_hi - un_hi) + 1;

        next_lo[0] = lo;    next_hi[0] = n;     next_d[0] = d;
        next_lo[1] = m;     next_hi[1] = hi;    next_d[1] = d;
        next_lo[2] = n + 1; next_hi[2] = m - 1; next_d[2] = d + 1;

#define NSIZE(a) (next_hi[a] - next_lo[a])
#define NSWAP(a, b) { tz = next_lo[a]; next_lo[a] = next_lo[b]; next_lo[b] = tz; \
                      tz = next_hi[a]; next_hi[a] = next_hi[b]; next_hi[b] = tz; \
                      tz = next_d[a];  next_d[a] = next_d[b];   next_d[b] = tz; }
        if (NSIZE(0) < NSIZE(1)) NSWAP(0, 1);
        if (NSIZE(1) < NSIZE(2)) NSWAP(1, 2);
        if (NSIZE(0) < NSIZE(1)) NSWAP(0, 1);
#undef NSIZE
#undef NSWAP
        stack_lo[sp] = next_lo[0]; stack_hi[sp] = next_hi[0]; stack_d[sp] = next_d[0]; sp++;
        stack_lo[sp] = next_lo[1]; stack_hi[sp] = next_hi[1]; stack_d[sp] = next_d[1]; sp++;
        stack_lo[sp] = next_lo[2]; stack_hi[sp] = next_hi[2]; stack_d[sp] = next_d[2]; sp++;
    }
}

#define SETMASK (1u << 21)
#define CLEARMASK (~SETMASK)
#define BIGFREQ(b) (ftab[((b) + 1) << 8] - ftab[(b) << 8])

/* block must have N_OVERSHOOT writable bytes after nblock; quadrant nblock+N_OVERSHOOT entries; ftab 65537.
   Everything after libbz2's set-up loops: bucket order, bucket sorts under the work budget, copying, quadrants. */
__device__ static void main_sort(uint32_t *ptr, uint8_t *block, uint16_t *quadrant, uint32_t *ftab, int32_t nblock,
                      int32_t *budget, PeriodInfo *pi)
{
    int32_t i, j, k, ss, sb;
    int32_t running_order[256], copy_start[256], copy_end[256];
    uint8_t big_done[256];
    uint8_t c1;
    uint16_t s;

    // (ftab, ptr, quadrant and the overshoot bytes were set up by main_setup_coop, all lanes)
    for (i = 0; i <= 255; i++) {
        big_done[i] = 0;
        running_order[i] = i;
    }
    {
        int32_t vv, h = 1;
        do h = 3 * h + 1; while (h <= 256);
        do {
            h = h / 3;
            for (i = h; i <= 255; i++) {
                vv = running_order[i];
                j = i;
                while (BIGFREQ(running_order[j - h]) > BIGFREQ(vv)) {
                    running_order[j] = running_order[j - h];
                    j = j - h;
                    if (j <= (h - 1)) break;
                }
                running_order[j] = vv;
            }
        } while (h != 1);
    }

    for (i = 0; i <= 255; i++) {
        ss = running_order[i];

        for (j = 0; j <= 255; j++) {
            if (j != ss) {
                sb = (ss << 8) + j;
                if (!(ftab[sb] & SETMASK)) {
                    int32_t lo = (int32_t)(ftab[sb] & CLEARMASK);
                    int32_t hi = (int32_t)(ftab[sb + 1] & CLEARMASK) - 1;
                    if (hi > lo) {
                        main_qsort3(ptr, block, quadrant, nblock, lo, hi, N_RADIX, budget, pi);
                        if (*budget < 0) return;
                    }
                }
                ftab[sb] |= SETMASK;
            }
        }

        for (j = 0; j <= 255; j++) {
            copy_start[j] = (int32_t)(ftab[(j << 8) + ss] & CLEARMASK);
            copy_end[j] = (int32_t)(ftab[(j << 8) + ss + 1] & CLEARMASK) - 1;
        }
        for (j = (int32_t)(ftab[ss << 8] & CLEARMASK); j < copy_start[ss]; j++) {
            k = (int32_t)ptr[j] - 1;
            if (k < 0) k += nblock;
            c1 = block[k];
            if (!big_done[c1]) ptr[copy_start[c1]++] = (uint32_t)k;
        }
        for (j = (int32_t)(ftab[(ss + 1) << 8] & CLEARMASK) - 1; j > copy_end[ss]; j--) {
            k = (int32_t)ptr[j] - 1;
            if (k < 0) k += nblock;
            c1 = block[k];
            if (!big_done[c1]) ptr[copy_end[c1]--] = (uint32_t)k;
        }

        for (j = 0; j <= 255; j++) ftab[(j << 8) + ss] |= SETMASK;

        big_done[ss] = 1;

        if (i < 255) {
            int32_t bb_start = (int32_t)(ftab[ss << 8] & CLEARMASK);
            int32_t bb_size = (int32_t)(ftab[(ss + 1) << 8] & CLEARMASK) - bb_start;
            int32_t shifts = 0;
            while ((bb_size >> shifts) > 65534) shifts++;
            for (j = bb_size - 1; j >= 0; j--) {
                int32_t a2update = (int32_t)ptr[bb_start + j];
                uint16_t qval = (uint16_t)(j >> shifts);
                if (qval) pi->quadrant_clean = 0;
                quadrant[a2update] = qval;
                if (a2update < N_OVERSHOOT) quadrant[a2update + nblock] = qval;
            }
        }
    }
}



// ---- cooperative set-up of the main sort --------------------------------------------------------------------------
// libbz2's mainSort starts with three loops over the block on one thread: a 65,536-bin histogram of the 2-byte
// prefixes key(i) = block[i] << 8 | block[i+1], its prefix sums, and the pointer fill "for i = n-1 .. 0:
// ptr[--ftab[key(i)]] = i", which leaves every 2-byte bucket holding its rotations in ASCENDING order and ftab[s]
// at the bucket's start.  The same result, by all lanes: histogram with atomics, an exclusive prefix over the bins,
// and a stable counting sort of i by key(i) as two stable 8-bit passes (low byte, then high byte), each with
// per-lane counts over contiguous chunks.  On one lane these loops were ~1.8 M dependent global read-modify-writes,
// most of the two seconds a periodic block used to take.
#define PER_NT 256
__device__ static void main_setup_coop(uint32_t *ptr, uint32_t *tmp, uint8_t *block, uint16_t *quadrant, uint32_t *ftab,
                                       uint32_t *cnt /* [PER_NT][256] */, int32_t nblock)
{
    __shared__ uint32_t s_tot[256];
    __shared__ uint32_t s_base[256];
    const uint32_t tid = threadIdx.x;
    const int32_t chunk = (nblock + PER_NT - 1) / PER_NT;
    const int32_t ca = (int32_t)tid * chunk < nblock ? (int32_t)tid * chunk : nblock;
    const int32_t cb = ca + chunk < nblock ? ca + chunk : nblock;
    for (int32_t i = (int32_t)tid; i <= 65536; i += PER_NT) ftab[i] = 0;
    for (int32_t i = (int32_t)tid; i < nblock + N_OVERSHOOT; i += PER_NT) quadrant[i] = 0;
    if (tid < N_OVERSHOOT) block[nblock + tid] = block[tid];
    __syncthreads();
    // histogram (consecutive equal keys of a lane are added at once: periodic blocks have few distinct keys)
    {
        uint32_t run_key = 0, run = 0;
        for (int32_t i = ca; i < cb; i++) {
            const uint32_t key = ((uint32_t)block[i] << 8) | block[i + 1];      // block[nblock] = block[0]
            if (run && key != run_key) {
                atomicAdd(&ftab[run_key], run);
                run = 0;
            }
            run_key = key;
            run++;
        }
        if (run) atomicAdd(&ftab[run_key], run);
    }
    __syncthreads();
    // exclusive prefix over the 65,536 bins: lane t owns bins [256 t, 256 t + 256)
    {
        uint32_t sum = 0;
        for (uint32_t k = 0; k < 256; k++) sum += ftab[tid * 256 + k];
        s_tot[tid] = sum;
        __syncthreads();
        uint32_t pre = 0;
        for (uint32_t t = 0; t < tid; t++) pre += s_tot[t];
        for (uint32_t k = 0; k < 256; k++) {
            const uint32_t c = ftab[tid * 256 + k];
            ftab[tid * 256 + k] = pre;
            pre += c;
        }
        if (tid == 0) ftab[65536] = (uint32_t)nblock;
    }
    __syncthreads();
    // stable counting sort of i by key(i): pass 0 by block[i+1] (identity -> tmp), pass 1 by block[i] (tmp -> ptr)
    for (int pass = 0; pass < 2; pass++) {
        for (int32_t c = 0; c < 256; c++) cnt[tid * 256 + c] = 0;
        for (int32_t x = ca; x < cb; x++) {
            const uint32_t i = pass ? tmp[x] : (uint32_t)x;
            cnt[tid * 256 + block[i + (pass ? 0 : 1)]]++;
        }
        __syncthreads();
        {
            uint32_t run = 0;                       // column tid: lanes in order
            for (uint32_t t = 0; t < PER_NT; t++) {
                const uint32_t v = cnt[t * 256 + tid];
                cnt[t * 256 + tid] = run;
                run += v;
            }
            s_tot[tid] = run;
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t acc = 0;
            for (int32_t c = 0; c < 256; c++) {
                s_base[c] = acc;
                acc += s_tot[c];
            }
        }
        __syncthreads();
        uint32_t *dst = pass ? ptr : tmp;
        for (int32_t x = ca; x < cb; x++) {
            const uint32_t i = pass ? tmp[x] : (uint32_t)x;
            const uint32_t c = block[i + (pass ? 0 : 1)];
            dst[s_base[c] + cnt[tid * 256 + c]++] = i;
        }
        __syncthreads();
    }
}

// ---- cooperative form of the fallback sort ---------------------------------------------------------------
// Same algorithm and, step for step, the same permutation as fallback_sort above (libbz2 fallbackSort), run by a
// whole workgroup: the position-parallel parts (bucket fill, equivalence-class assignment, header bits) are
// spread over the lanes, lane 0 walks the buckets and runs the 3-way quicksort, and large buckets whose keys
// are all equal -- the common case in a periodic block once the doubling depth passes the period; libbz2's
// quicksort provably leaves such a bucket untouched -- are recognised in parallel and skipped.
#define PER_COOP_MIN 4096

__device__ static void fallback_sort_coop(const uint8_t *__restrict__ T, uint32_t *fmap, uint32_t *eclass,
                                          uint32_t *bhtab, uint32_t *cnt /* [PER_NT][256] */, int32_t nblock)
{
    __shared__ int32_t s_start[257];
    __shared__ uint32_t s_scan[PER_NT];
    __shared__ int32_t s_cmd[4];       // [0] 1 = big bucket, 0 = round finished; [1] l; [2] r; [3] n_not_done
    const uint32_t tid = threadIdx.x;
    const int32_t chunk = (nblock + PER_NT - 1) / PER_NT;
    const int32_t ca = (int32_t)tid * chunk < nblock ? (int32_t)tid * chunk : nblock;
    const int32_t cb = ca + chunk < nblock ? ca + chunk : nblock;

    // ---- initial bucket sort by first byte; within a bucket the rotation indices are DESCENDING, as libbz2 leaves them
    for (int32_t c = 0; c < 256; c++) cnt[tid * 256 + c] = 0;
    for (int32_t i = ca; i < cb; i++) cnt[tid * 256 + T[i]]++;
    __syncthreads();
    {
        uint32_t run = 0;
        for (uint32_t t = 0; t < PER_NT; t++) {
            const uint32_t v = cnt[t * 256 + tid];
            cnt[t * 256 + tid] = run;
            run += v;
        }
        s_scan[tid] = run;      // occurrences of byte value tid
    }
    __syncthreads();
    if (tid == 0) {
        int32_t acc = 0;
        for (int32_t c = 0; c < 256; c++) {
            s_start[c] = acc;
            acc += (int32_t)s_scan[c];
        }
        s_start[256] = acc;
    }
    __syncthreads();
    for (int32_t i = ca; i < cb; i++) {
        const uint32_t c = T[i];
        const uint32_t rnk = cnt[tid * 256 + c]++;        // number of earlier positions holding c
        fmap[s_start[c + 1] - 1 - (int32_t)rnk] = (uint32_t)i;
    }
    const int32_t n_bh = nblock / 32 + 8;
    for (int32_t i = (int32_t)tid; i < n_bh; i += PER_NT) bhtab[i] = 0;
    __syncthreads();
    if (tid == 0) {
        for (int32_t c = 0; c < 256; c++) SET_BH(s_start[c]);
        for (int32_t i = 0; i < 32; i++) {
            SET_BH(nblock + 2 * i);
            CLEAR_BH(nblock + 2 * i + 1);
        }
    }
    __syncthreads();

    int32_t H = 1;
    // lane 0's bucket walk state
    int32_t k = 0, l = 0, r = -1, n_not_done = 0;
    for (;;) {
        // ---- equivalence classes: eclass[fmap[i] - H] = position of the bucket header at or before i
        {
            uint32_t last = 0;      // 1 + last header position inside my chunk
            for (int32_t i = ca; i < cb; i++)
                if (ISSET_BH(i)) last = (uint32_t)i + 1;
            s_scan[tid] = last;
            __syncthreads();
            uint32_t jin = 0;
            for (uint32_t t = 0; t < tid; t++)
                if (s_scan[t] > jin) jin = s_scan[t];
            int32_t j = jin ? (int32_t)jin - 1 : 0;
            for (int32_t i = ca; i < cb; i++) {
                if (ISSET_BH(i)) j = i;
                int32_t kk = (int32_t)fmap[i] - H;
                if (kk < 0) kk += nblock;
                eclass[kk] = (uint32_t)j;
            }
        }
        __syncthreads();

        // ---- buckets: lane 0 enumerates them exactly as libbz2 does; big ones are handled by everybody
        if (tid == 0) {
            n_not_done = 0;
            r = -1;
        }
        for (;;) {
            if (tid == 0) {
                s_cmd[0] = 0;
                for (;;) {
                    k = r + 1;
                    while (ISSET_BH(k) && UNALIGNED_BH(k)) k++;
                    if (ISSET_BH(k)) {
                        while (WORD_BH(k) == 0xffffffffu) k += 32;
                        while (ISSET_BH(k)) k++;
                    }
                    l = k - 1;
                    if (l >= nblock) break;
                    while (!ISSET_BH(k) && UNALIGNED_BH(k)) k++;
                    if (!ISSET_BH(k)) {
                        while (WORD_BH(k) == 0x00000000u) k += 32;
                        while (!ISSET_BH(k)) k++;
                    }
                    r = k - 1;
                    if (r >= nblock) break;
                    if (r > l) {
                        n_not_done += (r - l + 1);
                        if (r - l + 1 >= PER_COOP_MIN) {
                            s_cmd[0] = 1;
                            s_cmd[1] = l;
                            s_cmd[2] = r;
                            break;
                        }
                        fb_qsort3(fmap, eclass, l, r);
                        int32_t cc = -1;
                        for (int32_t i = l; i <= r; i++) {
                            const int32_t cc1 = (int32_t)eclass[fmap[i]];
                            if (cc != cc1) {
                                SET_BH(i);
                                cc = cc1;
                            }
                        }
                    }
                }
                s_cmd[3] = n_not_done;
            }
            __syncthreads();
            if (s_cmd[0] == 0) break;
            const int32_t bl = s_cmd[1], br = s_cmd[2];
            const uint32_t key0 = eclass[fmap[bl]];
            int same = 1;
            for (int32_t i = bl + (int32_t)tid; i <= br; i += PER_NT)
                if (eclass[fmap[i]] != key0) same = 0;
            const int all_same = __syncthreads_and(same);
            if (!all_same) {
                if (tid == 0) fb_qsort3(fmap, eclass, bl, br);
                __syncthreads();
                for (int32_t i = bl + 1 + (int32_t)tid; i <= br; i += PER_NT)
                    if (eclass[fmap[i]] != eclass[fmap[i - 1]]) atomicOr(&bhtab[i >> 5], (uint32_t)1 << (i & 31));
            }
            __syncthreads();
        }
        const int32_t nnd = s_cmd[3];
        __syncthreads();
        H *= 2;
        if (H > nblock || nnd == 0) break;
    }
}

// Workspace layout inside a sort slot (BzxSortWs): ptr = sa[], eclass = isa[], block copy + quadrant in u0,
// ftab / bhtab / per-lane byte counts in u1.
__global__ __launch_bounds__(PER_NT) void bzx_periodic_kernel(BzxBatch B)
{
    __shared__ int s_need_fallback;
    const BzxSortWs ws = B.sort_ws[blockIdx.x];
    const uint32_t n_per = B.counters[5];
    for (uint32_t li = blockIdx.x; li < n_per; li += gridDim.x) {
        const uint32_t b = B.plist[li];
        const int32_t n = (int32_t)B.blk[b].n;
        const uint8_t *T = BZX_BLOCK_PTR(B, B.blk[b]);
        uint32_t *ptr = ws.sa;
        uint32_t *eclass = ws.isa;
        uint8_t *block = (uint8_t *)ws.u0;                                  // n + N_OVERSHOOT bytes
        uint16_t *quadrant = (uint16_t *)(block + ((n + N_OVERSHOOT + 15) & ~15));   // n + N_OVERSHOOT entries
        uint32_t *ftab = (uint32_t *)ws.u1;                                 // 65537 words
        uint32_t *bhtab = ftab + 65600;                                     // n/32 + 8 words
        uint32_t *cnt = bhtab + 32768;                                      // PER_NT * 256 words
        if (threadIdx.x == 0) s_need_fallback = 1;
        __syncthreads();
        if (n >= 10000) {
            for (int32_t i = (int32_t)threadIdx.x; i < n; i += PER_NT) block[i] = T[i];
            __syncthreads();
            main_setup_coop(ptr, eclass, block, quadrant, ftab, cnt, n);
            if (threadIdx.x == 0) {
                int32_t budget = n * ((30 - 1) / 3);
                PeriodInfo pinfo;
                const uint32_t copies = B.blk[b].pad_[0];
                pinfo.period = (copies > 1 && (uint32_t)n % copies == 0) ? (uint32_t)n / copies : (uint32_t)n;
                pinfo.quadrant_clean = 1;
                main_sort(ptr, block, quadrant, ftab, n, &budget, &pinfo);
                s_need_fallback = budget < 0;
            }
            __syncthreads();
        }
        if (s_need_fallback) fallback_sort_coop(T, ptr, eclass, bhtab, cnt, n);
        __syncthreads();
        for (int32_t i = (int32_t)threadIdx.x; i < n; i += PER_NT)
            if (ptr[i] == 0) B.blk[b].orig_ptr = (uint32_t)i;
        __syncthreads();
    }
}

void bzx_launch_periodic(const BzxBatch &B, uint32_t grid, hipStream_t stream)
{
    hipLaunchKernelGGL(bzx_periodic_kernel, dim3(grid), dim3(PER_NT), 0, stream, B);
}
