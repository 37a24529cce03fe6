// bzx -- thin command line over libbzx.so, mirroring the reference's CLI (src/tools/cli.rs:113-303, src/main.rs:16-38)
// and C bzip2's conventions: -z / -d / -t, -1..-9, -c, -k, -f, -q, -v.  SURVEY.md 8f N4.  Host glue only: every byte of
// compression and decompression work happens on the device behind include/bzx.h; without a HIP device the tool fails.
// Compression streams the input through bzx_cstream_feed in chunks (files larger than device memory are fine and the
// output is written while the next chunk is compressed); decompression reads the whole .bz2.
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <string>
#include <vector>
#include "../include/bzx.h"

enum Mode { ZIP, UNZIP, TEST };
struct Opts {
    Mode mode = ZIP;
    int level = 9;
    bool to_stdout = false, keep = false, force = false;
    int verbose = 0;
    bool quiet = false;
    std::vector<std::string> files;
};

static void help()
{
    puts("usage: bzx [flags] [files ...]\n"
         "  -z --compress     compress (default)        -d --decompress   decompress\n"
         "  -t --test         check integrity           -c --stdout       write to standard output\n"
         "  -k --keep         keep input files          -f --force        overwrite output files\n"
         "  -1 .. -9          block size 100k .. 900k   --fast = -1, --best = -9 (default)\n"
         "  -q --quiet        no warnings               -v --verbose      statistics (-vv more)\n"
         "  -s --small        accepted, ignored         -h --help  -V --version  -L --license\n"
         "With no file, or when a file is -, reads standard input and writes standard output.");
}

static bool read_all(FILE *f, std::vector<uint8_t> &v)
{
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
    return !ferror(f);
}

static int fail(const Opts &o, const char *what, const char *name, bzx_ctx *ctx, int rc)
{
    if (!o.quiet) fprintf(stderr, "bzx: %s: %s: %s%s%s\n", name, what, bzx_strerror(rc), ctx && bzx_last_error(ctx)[0] ? ": " : "",
                          ctx ? bzx_last_error(ctx) : "");
    return 1;
}

// input stream -> .bz2 on `out`, chunk by chunk
static int do_zip(const Opts &o, bzx_ctx *ctx, FILE *in, FILE *out, const char *name)
{
    const size_t CH = (size_t)64 << 20;
    bzx_cstream *cs = nullptr;
    int rc = bzx_cstream_begin(ctx, o.level, CH, &cs);
    if (rc) return fail(o, "cannot start", name, ctx, rc);
    uint8_t *buf[2] = {(uint8_t *)bzx_host_alloc(CH), (uint8_t *)bzx_host_alloc(CH)};
    size_t cap = CH + CH / 50 + (1 << 20), total_in = 0, flushed = 0, produced = 0;
    uint8_t *obuf = (uint8_t *)malloc(cap);
    int ret = 0;
    if (!buf[0] || !buf[1] || !obuf) ret = fail(o, "out of memory", name, nullptr, BZX_E_NOMEM);
    size_t have = ret ? 0 : fread(buf[0], 1, CH, in);
    for (int k = 0; !ret; k++) {
        // read ahead to know whether this chunk is the last one
        const size_t next = have == CH ? fread(buf[(k + 1) & 1], 1, CH, in) : 0;
        const int fin = next == 0;
        // the whole stream must fit the output buffer the library writes into: grow it as the input grows
        const size_t need = total_in + have + (total_in + have) / 50 + (1 << 20);
        if (need > cap) {
            uint8_t *nb = (uint8_t *)realloc(obuf, need * 2);
            if (!nb) {
                ret = fail(o, "out of memory", name, nullptr, BZX_E_NOMEM);
                break;
            }
            memset(nb + cap, 0, need * 2 - cap);
            obuf = nb;
            cap = need * 2;
        }
        rc = bzx_cstream_feed(cs, buf[k & 1], have, fin, obuf, cap, &produced);
        if (rc) {
            ret = fail(o, "compression failed", name, ctx, rc);
            break;
        }
        total_in += have;
        if (produced > flushed) {
            if (fwrite(obuf + flushed, 1, produced - flushed, out) != produced - flushed) {
                ret = fail(o, strerror(errno), name, nullptr, BZX_OK);
                break;
            }
            flushed = produced;
        }
        if (fin) break;
        have = next;
    }
    if (!ret && o.verbose) {
        bzx_stats st;
        bzx_get_stats(ctx, &st);
        fprintf(stderr, "  %s: %zu -> %zu bytes, %.3f:1, %u blocks (%u periodic)\n", name, total_in, produced,
                produced ? (double)total_in / (double)produced : 0.0, st.nblk, st.n_periodic);
        // -vv: one line per block, the figures the reference logs at -vvv (compress_block.rs:58-63, huffman.rs:176-181)
        bzx_block_info bi;
        for (uint32_t b = 0; o.verbose > 1 && b < st.nblk && bzx_get_block_info(ctx, b, &bi) == BZX_OK; b++)
            fprintf(stderr, "    block %u: crc = 0x%08x, %u in block, origPtr %u%s, %u values in use, %u mtf symbols, "
                            "%u coding tables, %u selectors; bits: map %u + selectors %u + tables %u + codes %u -> %llu\n",
                    b + 1, bi.crc, bi.n, bi.orig_ptr, bi.periodic ? " (periodic)" : "", bi.n_in_use, bi.n_mtf, bi.n_tables,
                    bi.n_selectors, bi.bits_symbol_map, bi.bits_selectors, bi.bits_tables, bi.bits_payload,
                    (unsigned long long)bi.bits);
    }
    bzx_cstream_end(cs);
    bzx_host_free(buf[0]);
    bzx_host_free(buf[1]);
    free(obuf);
    return ret;
}

static int do_unzip(const Opts &o, bzx_ctx *ctx, FILE *in, FILE *out, const char *name)
{
    std::vector<uint8_t> z;
    if (!read_all(in, z)) return fail(o, strerror(errno), name, nullptr, BZX_OK);
    size_t cap = z.size() * 6 + (1 << 20), n = 0;
    std::vector<uint8_t> raw;
    for (;;) {
        raw.resize(cap);
        const int rc = bzx_decompress_buffer(ctx, z.data(), z.size(), raw.data(), cap, &n);
        if (rc == BZX_E_OUTBUF && n > cap) {
            cap = n;
            continue;
        }
        if (rc) return fail(o, o.mode == TEST ? "integrity check failed" : "decompression failed", name, ctx, rc);
        break;
    }
    if (out && fwrite(raw.data(), 1, n, out) != n) return fail(o, strerror(errno), name, nullptr, BZX_OK);
    if (o.verbose) fprintf(stderr, "  %s: %s, %zu -> %zu bytes\n", name, o.mode == TEST ? "ok" : "done", z.size(), n);
    return 0;
}

int main(int argc, char **argv)
{
    Opts o;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "-" || a[0] != '-') {
            o.files.push_back(a);
        } else if (a.rfind("--", 0) == 0) {
            if (a == "--help") { help(); return 0; }
            else if (a == "--version" || a == "--license") { printf("bzx, bzip2 block compression on MI355X; %s\n", bzx_version()); return 0; }
            else if (a == "--decompress") o.mode = UNZIP;
            else if (a == "--compress") o.mode = ZIP;
            else if (a == "--test") o.mode = TEST;
            else if (a == "--stdout") o.to_stdout = true;
            else if (a == "--keep") o.keep = true;
            else if (a == "--force") o.force = true;
            else if (a == "--quiet") o.quiet = true;
            else if (a == "--verbose") o.verbose++;
            else if (a == "--small") {}
            else if (a == "--fast") o.level = 1;
            else if (a == "--best") o.level = 9;
            else { fprintf(stderr, "bzx: unexpected argument %s\n", a.c_str()); return 1; }
        } else {
            for (size_t k = 1; k < a.size(); k++) {
                const char c = a[k];
                if (c >= '1' && c <= '9') o.level = c - '0';
                else if (c == 'd') o.mode = UNZIP;
                else if (c == 'z') o.mode = ZIP;
                else if (c == 't') o.mode = TEST;
                else if (c == 'c') o.to_stdout = true;
                else if (c == 'k') o.keep = true;
                else if (c == 'f') o.force = true;
                else if (c == 'q') o.quiet = true;
                else if (c == 'v') o.verbose++;
                else if (c == 's') {}
                else if (c == 'h') { help(); return 0; }
                else if (c == 'V' || c == 'L') { printf("bzx, bzip2 block compression on MI355X; %s\n", bzx_version()); return 0; }
                else { fprintf(stderr, "bzx: unexpected flag -%c\n", c); return 1; }
            }
        }
    }
    bzx_ctx *ctx = nullptr;
    int rc = bzx_ctx_create(0, 0, &ctx);
    if (rc) {
        fprintf(stderr, "bzx: %s (the product has no CPU path)\n", bzx_strerror(rc));
        return 2;
    }
    if (o.files.empty()) o.files.push_back("-");
    int ret = 0;
    for (const std::string &f : o.files) {
        const bool std_in = f == "-";
        FILE *in = std_in ? stdin : fopen(f.c_str(), "rb");
        if (!in) {
            if (!o.quiet) fprintf(stderr, "bzx: %s: %s\n", f.c_str(), strerror(errno));
            ret = 1;
            continue;
        }
        std::string oname;
        FILE *out = nullptr;
        if (o.mode != TEST) {
            if (std_in || o.to_stdout) {
                out = stdout;
            } else {
                if (o.mode == ZIP) oname = f + ".bz2";
                else if (f.size() > 4 && f.compare(f.size() - 4, 4, ".bz2") == 0) oname = f.substr(0, f.size() - 4);
                else oname = f + ".out";
                struct stat sb;
                if (!o.force && stat(oname.c_str(), &sb) == 0) {
                    if (!o.quiet) fprintf(stderr, "bzx: %s already exists (use -f)\n", oname.c_str());
                    fclose(in);
                    ret = 1;
                    continue;
                }
                out = fopen(oname.c_str(), "wb");
                if (!out) {
                    if (!o.quiet) fprintf(stderr, "bzx: %s: %s\n", oname.c_str(), strerror(errno));
                    fclose(in);
                    ret = 1;
                    continue;
                }
            }
        }
        const int r = o.mode == ZIP ? do_zip(o, ctx, in, out, f.c_str()) : do_unzip(o, ctx, in, out, f.c_str());
        if (!std_in) fclose(in);
        int wr = 0;                                  // the output is complete on disk only when flush and close succeed
        if (out && out != stdout) {
            if (fflush(out) != 0 || ferror(out)) wr = 1;
            if (fclose(out) != 0) wr = 1;
            if (wr && !r && !o.quiet) fprintf(stderr, "bzx: %s: %s\n", oname.c_str(), strerror(errno));
            if (r || wr) unlink(oname.c_str());      // never leave a partial output ...
            else if (!o.keep) unlink(f.c_str());     // ... and never remove the input unless the output is whole
        } else if (out) {
            if (fflush(out) != 0 || ferror(out)) wr = 1;
        }
        ret |= r | wr;
    }
    bzx_ctx_destroy(ctx);
    return ret;
}
