// bzx_periodic.hip -- origPtr of PERIODIC blocks on gfx950.
//
// A block u^k (k > 1) has k identical copies of every rotation.  The last column L does not depend
// on how ties are ordered, but the 24-bit origPtr (row of rotation 0) does, and the metric is
// bit-exactness with C bzip2 1.0.8, whose tie order falls out of its block sorter's internals
// (the reference leaves it unspecified: src/bwt_algorithms/bwt_sort.rs:40-42 sort_unstable).
// The sorters flag such blocks (BZX_ST_PERIODIC).  For them only, this kernel reproduces what the published
// libbz2 1.0.8 block sorter does to tied rotations -- "main" sort with its work budget of 9*nblock, "fallback" sort
// (bucket + doubling with LCG-pivot 3-way quicksort) when the budget runs out or nblock < 10000 -- one workgroup
// per block, and takes origPtr from the result.  The permutation is libbz2's; the execution is not:
//   * the main sort's set-up (65,536-bin histogram, pointer fill) is a parallel counting sort (main_setup_coop);
//   * the main sort proper runs on one wave in lockstep, comparing two rotations 64 positions per step with the
//     budget charged as the serial loop would (main_gtu), and skipping depths at which a large range is uniform;
//   * the fallback sort spreads class assignment and header bits over the lanes, partitions large buckets in
//     parallel with the serial partition's exact permutation (fb_partition_coop: pairing by counts, the "equal"
//     swaps replayed as a queue on a tape and resolved by pointer jumping), and stops at the first doubling round
//     that splits nothing, which provably is the final state.
// Small ranges (< PER_COOP_MIN) and the pivot-choice sequence stay serial on lane 0.  Exactly periodic blocks are
// vanishingly rare outside all-equal-byte inputs (config 5a: 6 blocks); an all-zero 900 KB block takes ~90 ms.
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

// r: state of libbz2's pivot-choice generator when the range is entered (0 at the start of a bucket); returns the
// state it is left in, so that a caller that has split off the top of the recursion can carry it on.
__device__ static uint32_t fb_qsort3(uint32_t *fmap, const uint32_t *eclass, int32_t lo_st, int32_t hi_st, uint32_t r = 0)
{
    int32_t un_lo, un_hi, lt_lo, gt_hi, n, m, sp, lo, hi;
    uint32_t med, r3;
    int32_t stack_lo[STACK_SZ], stack_hi[STACK_SZ];

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
    return r;
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

// The main sort below is executed by ONE WAVE in lockstep: every lane runs the same (uniform) control flow and
// stores the same values, so the sorter's logic reads like the serial algorithm, and the one expensive primitive --
// "is rotation i1 greater than rotation i2" -- is done 64 positions at a time.  libbz2's comparison looks at bytes
// 0..11, then at (byte, quadrant) pairs in steps of eight, charging one budget unit per completed step of eight and
// giving up after (nblock + 8) / 8 + 1 of them; the first difference in that order decides.  Lane t looks at
// offset 12 + 64 c + t in chunk c; the lowest lane that sees a difference (byte before quadrant) decides, and the
// charge is the number of steps of eight completed before it.  Positions are taken modulo nblock: libbz2 reads up
// to 33 entries past the end instead, where both arrays mirror their start.
__device__ static inline int main_gtu(uint32_t i1, uint32_t i2, const uint8_t *block, const uint16_t *quadrant,
                           uint32_t nblock, int32_t *budget, const PeriodInfo *pi)
{
    if (pi->quadrant_clean && (i1 % pi->period) == (i2 % pi->period)) {
        *budget -= (int32_t)((nblock + 8) / 8 + 1);
        return 0;
    }
    const uint32_t lane = bzx_lane();
    {
        const bool df = lane < 12 && block[i1 + lane] != block[i2 + lane];
        const uint64_t m = __ballot(df);
        if (m) {
            const uint32_t f = (uint32_t)__builtin_ctzll(m);
            return block[i1 + f] > block[i2 + f];
        }
    }
    i1 += 12;
    i2 += 12;
    if (i1 >= nblock) i1 -= nblock;
    if (i2 >= nblock) i2 -= nblock;
    const uint32_t total = (nblock + 8) / 8 + 1;
    for (uint32_t c = 0;; c++) {
        const uint32_t step = 8 * c + (lane >> 3);
        uint32_t a = i1 + 64 * c + lane, b = i2 + 64 * c + lane;
        if (a >= nblock) a -= nblock;
        if (a >= nblock) a -= nblock;
        if (b >= nblock) b -= nblock;
        if (b >= nblock) b -= nblock;
        uint32_t c1 = 0, c2 = 0, s1 = 0, s2 = 0;
        if (step < total) {
            c1 = block[a];
            c2 = block[b];
            s1 = quadrant[a];
            s2 = quadrant[b];
        }
        const uint64_t m = __ballot(c1 != c2 || s1 != s2);
        if (m) {
            const int f = __builtin_ctzll(m);
            const uint32_t fc1 = __shfl(c1, f), fc2 = __shfl(c2, f), fs1 = __shfl(s1, f), fs2 = __shfl(s2, f);
            *budget -= (int32_t)(8 * c + ((uint32_t)f >> 3));
            return fc1 != fc2 ? fc1 > fc2 : fs1 > fs2;
        }
        if (8 * c + 8 >= total) break;
    }
    *budget -= (int32_t)total;
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
#define MAIN_WIDE 2048               // ranges from this size on are checked for uniform bytes by the whole wave

__device__ static void main_qsort3(uint32_t *ptr, const uint8_t *block, const uint16_t *quadrant, int32_t nblock,
                        int32_t lo_st, int32_t hi_st, int32_t d_st, int32_t *budget, const PeriodInfo *pi)
{
    int32_t un_lo, un_hi, lt_lo, gt_hi, n, m, med, sp, lo, hi, d;
    __shared__ int32_t stack_lo[STACK_SZ], stack_hi[STACK_SZ], stack_d[STACK_SZ];    // (one wave runs this)
    int32_t next_lo[3], next_hi[3], next_d[3], tz;

    sp = 0;
    stack_lo[sp] = lo_st; stack_hi[sp] = hi_st; stack_d[sp] = d_st; sp++;

    while (sp > 0) {
        sp--;
        lo = stack_lo[sp]; hi = stack_hi[sp]; d = stack_d[sp];
        if (hi - lo >= MAIN_WIDE && d <= MAIN_DEPTH) {
            // A range whose rotations all carry the same byte at depth d is left exactly as it is by the partition
            // below and comes back at depth d + 1: find, 64 rotations at a time, the first depth at which the range
            // is not uniform and go there directly (periodic blocks have ranges of 10^5 identical rotations).
            const uint32_t lane = bzx_lane();
            const uint32_t p0 = ptr[lo];
            int32_t first = MAIN_DEPTH + 1;
            for (int32_t i = lo + (int32_t)lane; i <= hi; i += 64) {
                const uint32_t pp = ptr[i];
                for (int32_t dd = d; dd < first; dd++)
                    if (block[pp + dd] != block[p0 + dd]) first = dd;
            }
            for (int sh = 32; sh > 0; sh >>= 1) {
                const int32_t o = __shfl_xor(first, sh);
                first = o < first ? o : first;
            }
            d = first;
        }
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
    __shared__ int32_t running_order[256], copy_start[256], copy_end[256];            // (one wave runs this)
    __shared__ uint8_t big_done[256];
    uint8_t c1;

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
#ifdef BZX_DIAG
#define PSTAMP(slot)                                                          \
    do {                                                                      \
        if (dbg && threadIdx.x == 0) {                                        \
            const unsigned long long now_ = wall_clock64();                   \
            atomicAdd(&dbg[slot], now_ - t_last_);                            \
            t_last_ = now_;                                                   \
        }                                                                     \
    } while (0)
#define PSTAMP_DECL unsigned long long t_last_ = wall_clock64()
#else
#define PSTAMP(slot) do {} while (0)
#define PSTAMP_DECL do {} while (0)
#endif

// Work arrays of the parallel partition (all inside the block's sort slot, dead while the fallback sort runs).
struct FbWork {
    uint8_t *cls;       // class of every element of the range: 0 below the pivot, 1 equal, 2 above
    uint32_t *pg, *pl;  // positions of the out-of-place "above" / "below" elements that the scans exchange
    uint32_t *nxt;      // queue replay: where the element that ends at a position comes from
    uint32_t *out;      // the permuted range
};

// Exclusive prefix over one value per lane (PER_NT lanes); total returned in `total`.
__device__ static inline uint32_t fb_lane_prefix(uint32_t v, uint32_t *s_scan, uint32_t &total)
{
    const uint32_t tid = threadIdx.x;
    __syncthreads();
    s_scan[tid] = v;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
    for (uint32_t t = 0; t < PER_NT; t++) {
        const uint32_t x = s_scan[t];
        pre += t < tid ? x : 0u;
        tot += x;
    }
    total = tot;
    return pre;
}

// libbz2's three-way partition (fallbackQSort3's inner loops) of fmap[lo..hi] around `med`, by the whole workgroup,
// with exactly the permutation the serial scans produce.  What the serial code does, seen as data movement:
//   * the left scan stops at elements above the pivot, the right scan at elements below it, and the j-th such
//     stop of one side is exchanged with the j-th of the other as long as they have not crossed -- a pairing
//     that only depends on the counts of "above" elements to the left and "below" elements to the right;
//   * after these exchanges the scans meet at X, the first position holding an "above" element; the left scan has
//     then seen only "below"/"equal" elements in [lo, X), the right scan only "above"/"equal" ones in [X, hi];
//   * an "equal" element met by the left scan is swapped with the FIRST element of the "below" zone collected so
//     far, which thereby moves to the zone's end: the zone is a queue, "below" = push, "equal" = pop + push.
//     Replaying the queue on a tape (every operation appends one entry, a pop + push re-appends the entry at the
//     read head = number of pops so far) makes each final position either an original element or a copy of an
//     earlier tape entry; pointer jumping resolves the copies.  The "equal" elements end up in arrival order at the
//     range's left end.  The right scan mirrors all of this.
// Returns the sizes of the four zones; nothing is written when every element equals the pivot.
__device__ static void fb_partition_coop(uint32_t *fmap, const uint32_t *eclass, int32_t lo, int32_t hi, uint32_t med,
                                         const FbWork &W, uint32_t *s_scan, uint32_t *s_red, uint32_t &n_el,
                                         uint32_t &n_l, uint32_t &n_g, uint32_t &n_er)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t s = (uint32_t)(hi - lo + 1);
    const uint32_t ch = (s + PER_NT - 1) / PER_NT;
    const uint32_t ca = tid * ch < s ? tid * ch : s, cb = ca + ch < s ? ca + ch : s;
    uint32_t *f = fmap + lo;

    // classes and their counts per chunk
    uint32_t cl = 0, ce = 0, cg = 0;
    for (uint32_t i = ca; i < cb; i++) {
        const uint32_t key = eclass[f[i]];
        const uint32_t c = key < med ? 0u : key == med ? 1u : 2u;
        W.cls[i] = (uint8_t)c;
        cl += c == 0;
        ce += c == 1;
        cg += c == 2;
    }
    uint32_t tot_l, tot_e, tot_g;
    const uint32_t pre_l = fb_lane_prefix(cl, s_scan, tot_l);
    const uint32_t pre_e = fb_lane_prefix(ce, s_scan, tot_e);
    const uint32_t pre_g = fb_lane_prefix(cg, s_scan, tot_g);
    if (tot_e == s) {
        n_el = s;
        n_l = n_g = n_er = 0;
        return;
    }
    // exchanges: the j-th "above" from the left with the j-th "below" from the right while the former lies left
    if (tid == 0) s_red[0] = 0;
    __syncthreads();
    {
        uint32_t gl = pre_g, ll = pre_l, mine = 0;              // "above" / "below" elements left of i
        for (uint32_t i = ca; i < cb; i++) {
            const uint32_t c = W.cls[i];
            if (c == 2) {
                if (tot_l - ll >= gl + 1) {                      // enough "below" elements to my right
                    W.pg[gl] = i;
                    mine++;
                }
                gl++;
            } else if (c == 0) {
                const uint32_t j = tot_l - ll;                   // my rank from the right, 1-based
                if (gl >= j) W.pl[j - 1] = i;
                ll++;
            }
        }
        if (mine) atomicAdd(&s_red[0], mine);
    }
    __syncthreads();
    const uint32_t n_x = s_red[0];
    for (uint32_t j = tid; j < n_x; j += PER_NT) {
        const uint32_t a = W.pg[j], b = W.pl[j];
        const uint32_t t = f[a];
        f[a] = f[b];
        f[b] = t;
        W.cls[a] = 0;
        W.cls[b] = 2;
    }
    if (tid == 0) {
        s_red[1] = s;          // X: first "above" element
        s_red[2] = s;          // first "below" element
        s_red[3] = 0;          // 1 + last "above" element
    }
    __syncthreads();
    {
        uint32_t fx = s, fl = s, lg = 0;
        for (uint32_t i = ca; i < cb; i++) {
            const uint32_t c = W.cls[i];
            if (c == 2) {
                if (fx == s) fx = i;
                lg = i + 1;
            } else if (c == 0 && fl == s) {
                fl = i;
            }
        }
        if (fx < s) atomicMin(&s_red[1], fx);
        if (fl < s) atomicMin(&s_red[2], fl);
        if (lg) atomicMax(&s_red[3], lg);
    }
    __syncthreads();
    const uint32_t X = s_red[1], a0 = s_red[2];
    const uint32_t b0 = s_red[3] ? s_red[3] - 1 : 0;            // (only used when an "above" element exists)
    const uint32_t e_left_zone = X - tot_l;                      // "equal" elements the left scan collects
    const uint32_t e_right_zone = tot_e - e_left_zone;
    // "equal" elements: arrival order at the two ends; queue entries: source pointers
    {
        uint32_t el = pre_e;                                     // "equal" elements left of i
        for (uint32_t i = ca; i < cb; i++) {
            const uint32_t c = W.cls[i];
            if (c == 1) {
                if (i < X) {
                    W.out[el] = f[i];
                    W.nxt[i] = i < a0 ? i : a0 + (el - a0);      // pops before me = "equal" elements in [a0, i)
                } else {
                    W.out[s - (tot_e - el)] = f[i];              // rank from the right: tot_e - el - 1
                    // "equal" elements in (i, b0]: all of the tot_e - el - 1 to my right minus the s - 1 - b0 above b0
                    W.nxt[i] = i > b0 ? i : b0 - ((tot_e - el - 1) - (s - 1 - b0));
                }
                el++;
            } else {
                W.nxt[i] = i;
            }
        }
    }
    // resolve the copies (a chain ends at an entry that points to itself)
    for (;;) {
        __syncthreads();
        int changed = 0;
        for (uint32_t i = tid; i < s; i += PER_NT) {
            const uint32_t v = W.nxt[i];
            if (v != i) {
                const uint32_t w = W.nxt[v];
                if (w != v) {
                    W.nxt[i] = w;
                    changed = 1;
                }
            }
        }
        if (!__syncthreads_or(changed)) break;
    }
    // the two queues' final contents: [e_left_zone, X) and [X, s - e_right_zone)
    for (uint32_t i = e_left_zone + tid; i < s - e_right_zone; i += PER_NT) W.out[i] = f[W.nxt[i]];
    __syncthreads();
    for (uint32_t i = tid; i < s; i += PER_NT) f[i] = W.out[i];
    __syncthreads();
    n_el = e_left_zone;
    n_l = tot_l;
    n_g = tot_g;
    n_er = e_right_zone;
}

// libbz2's fallbackQSort3 on fmap[lo_st..hi_st] by the whole workgroup: ranges of PER_COOP_MIN elements and more are
// partitioned in parallel, smaller ones (with everything below them) by lane 0, in the serial algorithm's stack order
// and with its pivot-choice sequence carried through.
__device__ static void fb_qsort3_coop(uint32_t *fmap, const uint32_t *eclass, int32_t lo_st, int32_t hi_st, const FbWork &W,
                                      uint32_t *s_scan, uint32_t *s_red)
{
    __shared__ int32_t q_lo[STACK_SZ], q_hi[STACK_SZ];
    __shared__ uint32_t q_r;
    const uint32_t tid = threadIdx.x;
    int32_t sp = 0;
    uint32_t r = 0;
    __syncthreads();
    if (tid == 0) {
        q_lo[0] = lo_st;
        q_hi[0] = hi_st;
    }
    sp = 1;
    while (sp > 0) {
        __syncthreads();
        sp--;
        const int32_t lo = q_lo[sp], hi = q_hi[sp];
        if (hi - lo + 1 < PER_COOP_MIN) {
            if (tid == 0) q_r = fb_qsort3(fmap, eclass, lo, hi, r);
            __syncthreads();
            r = q_r;
            continue;
        }
        r = ((r * 7621) + 1) % 32768;
        const uint32_t r3 = r % 3;
        const uint32_t med = eclass[fmap[r3 == 0 ? lo : r3 == 1 ? (lo + hi) >> 1 : hi]];
        uint32_t n_el, n_l, n_g, n_er;
        fb_partition_coop(fmap, eclass, lo, hi, med, W, s_scan, s_red, n_el, n_l, n_g, n_er);
        if (n_l + n_g == 0) continue;                              // every element equals the pivot
        // "equal" zones to the middle (the two exchanged stretches never overlap)
        const int32_t un_lo = lo + (int32_t)(n_el + n_l);
        {
            const uint32_t k = n_el < n_l ? n_el : n_l;
            for (uint32_t t = tid; t < k; t += PER_NT) swap_u32(&fmap[lo + (int32_t)t], &fmap[un_lo - (int32_t)k + (int32_t)t]);
            const uint32_t m = n_er < n_g ? n_er : n_g;
            for (uint32_t t = tid; t < m; t += PER_NT) swap_u32(&fmap[un_lo + (int32_t)t], &fmap[hi - (int32_t)m + 1 + (int32_t)t]);
        }
        const int32_t n = lo + (int32_t)n_l - 1, m = hi - (int32_t)n_g + 1;
        __syncthreads();
        if (tid == 0) {
            if (n - lo > hi - m) {
                q_lo[sp] = lo; q_hi[sp] = n;
                q_lo[sp + 1] = m; q_hi[sp + 1] = hi;
            } else {
                q_lo[sp] = m; q_hi[sp] = hi;
                q_lo[sp + 1] = lo; q_hi[sp + 1] = n;
            }
        }
        sp += 2;
    }
    __syncthreads();
}

__device__ static void fallback_sort_coop(const uint8_t *__restrict__ T, uint32_t *fmap, uint32_t *eclass,
                                          uint32_t *bhtab, uint32_t *cnt /* [PER_NT][256] */, int32_t nblock,
                                          const FbWork &W, unsigned long long *dbg)
{
    PSTAMP_DECL;
    __shared__ int32_t s_start[257];
    __shared__ uint32_t s_scan[PER_NT];
    __shared__ uint32_t s_red[4];
    __shared__ int32_t s_cmd[5];       // [0] 1 = big bucket, 0 = round finished; [1] l; [2] r; [3] n_not_done; [4] a bucket was split
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

    PSTAMP(112);
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
        PSTAMP(113);

        // ---- buckets: lane 0 enumerates them exactly as libbz2 does; big ones are handled by everybody
        if (tid == 0) {
            n_not_done = 0;
            r = -1;
            s_cmd[4] = 0;
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
                                if (i > l) s_cmd[4] = 1;
                                SET_BH(i);
                                cc = cc1;
                            }
                        }
                    }
                }
                s_cmd[3] = n_not_done;
            }
            __syncthreads();
            PSTAMP(114);
            if (s_cmd[0] == 0) break;
            const int32_t bl = s_cmd[1], br = s_cmd[2];
            fb_qsort3_coop(fmap, eclass, bl, br, W, s_scan, s_red);
            PSTAMP(116);
            int split = 0;
            for (int32_t i = bl + 1 + (int32_t)tid; i <= br; i += PER_NT)
                if (eclass[fmap[i]] != eclass[fmap[i - 1]]) {
                    atomicOr(&bhtab[i >> 5], (uint32_t)1 << (i & 31));
                    split = 1;
                }
            if (__syncthreads_or(split) && tid == 0) s_cmd[4] = 1;
            __syncthreads();
            PSTAMP(115);
        }
        const int32_t nnd = s_cmd[3], any_split = s_cmd[4];
        __syncthreads();
        H *= 2;
        if (H > nblock || nnd == 0) break;
        // A round that splits no bucket is the last one that could have: the classes by 2H symbols equal those by H
        // symbols, hence (induction over x ~2H y <=> x ~H y and x+H ~H y+H) by any number of symbols, and sorting a
        // bucket whose keys all agree moves nothing (the insertion sort compares strictly, the partition swaps every
        // element with itself).  libbz2 would go on doubling H up to nblock without changing fmap.
        if (!any_split) break;
    }
}

// Workspace layout inside a sort slot (BzxSortWs): ptr = sa[], eclass = isa[], block copy + quadrant in u0,
// ftab / bhtab / per-lane byte counts in u1.
__global__ __launch_bounds__(PER_NT) void bzx_periodic_kernel(BzxBatch B)
{
    __shared__ int s_need_fallback;
    const BzxSortWs ws = B.sort_ws[blockIdx.x];
    const uint32_t n_per = B.counters[5];
#ifdef BZX_DIAG
    unsigned long long *dbg = B.dbg;
#else
    unsigned long long *dbg = nullptr;
#endif
    PSTAMP_DECL;
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
            PSTAMP(110);
            if (threadIdx.x < 64) {                                           // one wave, in lockstep (see main_gtu)
                int32_t budget = n * ((30 - 1) / 3);
                PeriodInfo pinfo;
                const uint32_t copies = B.blk[b].pad_[0];
                pinfo.period = (copies > 1 && (uint32_t)n % copies == 0) ? (uint32_t)n / copies : (uint32_t)n;
                pinfo.quadrant_clean = 1;
                main_sort(ptr, block, quadrant, ftab, n, &budget, &pinfo);
                s_need_fallback = budget < 0;
            }
            __syncthreads();
            PSTAMP(111);
        }
        if (s_need_fallback) {
            FbWork W;                                                       // (block copy, quadrants, ftab: dead by now)
            W.pg = (uint32_t *)ws.u0;
            W.pl = W.pg + BZX_MAX_N;
            W.nxt = ws.s0;
            W.out = ws.s1;
            W.cls = (uint8_t *)(cnt + PER_NT * 256);
            fallback_sort_coop(T, ptr, eclass, bhtab, cnt, n, W, dbg);
        }
        __syncthreads();
        PSTAMP(117);
        for (int32_t i = (int32_t)threadIdx.x; i < n; i += PER_NT)
            if (ptr[i] == 0) B.blk[b].orig_ptr = (uint32_t)i;
        __syncthreads();
    }
}

void bzx_launch_periodic(const BzxBatch &B, uint32_t grid, hipStream_t stream)
{
    hipLaunchKernelGGL(bzx_periodic_kernel, dim3(grid), dim3(PER_NT), 0, stream, B);
}
