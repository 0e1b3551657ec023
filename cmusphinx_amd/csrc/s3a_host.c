/*
 * s3a_host.c -- host C side of libcmusphinx_amd: error reporting, the S3
 * binary file envelope, integer log-domain arithmetic, acoustic-model
 * loading + precomputation and transition-matrix conversion.
 *
 * Behavioural parity with the reference (paths relative to cjac/cmusphinx):
 *   envelope        sphinxbase/src/libsphinxbase/util/bio.c:187-262, 265-296, 491-504
 *   logmath         sphinxbase/src/libsphinxbase/util/logmath.c:61-161, 391-483
 *   logs3           sphinx3/src/libs3decoder/libcommon/logs3.c:101-119
 *   means/variances sphinx3/src/libs3decoder/libam/cont_mgau.c:148-429
 *   mixture weights cont_mgau.c:507-683
 *   compaction/floor/precompute   cont_mgau.c:700-894, order of cont_mgau.c:938-951
 *   tmat            sphinx3/src/libs3decoder/libam/tmat.c:155-270
 *
 * libm's log() is used only here, at initialisation, exactly where the
 * reference uses it, so table entries and lrd terms carry the same bits.
 * This file must be compiled without -ffast-math and with -ffp-contract=off.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <math.h>

#include "s3a_internal.h"

/* ------------------------------------------------------------------ */
/* errors                                                              */
/* ------------------------------------------------------------------ */
static __thread char g_err[512];

void
s3a_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

/* kernel variants (include/cmusphinx_amd.h): process-wide, set by the host program, never read from the environment */
static s3a_variants_t g_variants;
void s3a_variants_default(s3a_variants_t *v) { if (v) memset(v, 0, sizeof *v); }
int32_t
s3a_set_variants(const s3a_variants_t *v)
{
    if (!v) return S3A_EINVAL;
    if (v->score_nt != 0 && v->score_nt != 256 && v->score_nt != 512 && v->score_nt != 1024) { s3a_set_error("s3a_set_variants: score_nt 0 / 256 / 512 / 1024"); return S3A_EINVAL; }
    g_variants = *v;
    return S3A_OK;
}
const s3a_variants_t *s3a_variants(void) { return &g_variants; }
void s3a_get_variants(s3a_variants_t *v) { if (v) *v = g_variants; }

const char *
s3a_last_error(void)
{
    return g_err;
}

const char *
s3a_version(void)
{
    return "cmusphinx_amd 0.1 (gfx950)";
}

/* ------------------------------------------------------------------ */
/* S3 binary envelope                                                  */
/* ------------------------------------------------------------------ */
static uint32_t
bswap32(uint32_t v)
{
    return (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24);
}

int32_t
s3a_bio_read(const char *path, const char *expect_version, uint32_t **words, size_t *n_words)
{
    FILE *fp;
    long fsize;
    unsigned char *buf = NULL;
    size_t pos, hdr_end = 0, n, i;
    int have_chksum = 0, swap;
    uint32_t magic, *w;

    *words = NULL;
    *n_words = 0;
    if ((fp = fopen(path, "rb")) == NULL) {
        s3a_set_error("cannot open %s", path);
        return S3A_EIO;
    }
    fseek(fp, 0, SEEK_END);
    fsize = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    if (fsize < 12 || (buf = (unsigned char *)malloc((size_t)fsize + 1)) == NULL) {
        fclose(fp);
        s3a_set_error("%s: empty file or out of memory", path);
        return fsize < 12 ? S3A_EIO : S3A_ENOMEM;
    }
    if (fread(buf, 1, (size_t)fsize, fp) != (size_t)fsize) {
        fclose(fp);
        free(buf);
        s3a_set_error("%s: short read", path);
        return S3A_EIO;
    }
    fclose(fp);
    buf[fsize] = 0;

    /* header: "s3\n" then "name value\n" lines up to "endhdr\n" */
    if (memcmp(buf, "s3\n", 3) != 0) {
        free(buf);
        s3a_set_error("%s: not an s3 binary file (old headerless format unsupported)", path);
        return S3A_EIO;
    }
    pos = 3;
    for (;;) {
        char *line = (char *)buf + pos, *nl = memchr(line, '\n', (size_t)fsize - pos);
        char name[256] = "", val[256] = "";
        if (nl == NULL) {
            free(buf);
            s3a_set_error("%s: premature EOF in header", path);
            return S3A_EIO;
        }
        *nl = 0;
        pos = (size_t)(nl - (char *)buf) + 1;
        if (sscanf(line, "%255s %255s", name, val) < 1)
            continue;
        if (strcmp(name, "endhdr") == 0) {
            hdr_end = pos;
            break;
        }
        if (name[0] == '#')
            continue;
        if (strcmp(name, "chksum0") == 0)
            have_chksum = 1;
        else if (strcmp(name, "version") == 0 && expect_version
                 && strcmp(val, expect_version) != 0)
            fprintf(stderr, "WARNING: %s: version %s, expecting %s\n", path, val,
                    expect_version);
    }
    if ((size_t)fsize < hdr_end + 4 || (((size_t)fsize - hdr_end) & 3) != 0) {
        free(buf);
        s3a_set_error("%s: payload is not a whole number of 32-bit words", path);
        return S3A_EIO;
    }
    memcpy(&magic, buf + hdr_end, 4);
    if (magic == 0x11223344u)
        swap = 0;
    else if (bswap32(magic) == 0x11223344u)
        swap = 1;
    else {
        free(buf);
        s3a_set_error("%s: bad byte-order magic %08x", path, magic);
        return S3A_EIO;
    }
    n = ((size_t)fsize - hdr_end - 4) / 4;
    if ((w = (uint32_t *)malloc(4 * (n ? n : 1))) == NULL) {
        free(buf);
        return S3A_ENOMEM;
    }
    memcpy(w, buf + hdr_end + 4, 4 * n);
    free(buf);
    if (swap)
        for (i = 0; i < n; i++)
            w[i] = bswap32(w[i]);
    if (have_chksum) {
        uint32_t sum = 0;
        if (n < 1) {
            free(w);
            s3a_set_error("%s: checksum announced but missing", path);
            return S3A_EIO;
        }
        n -= 1;
        for (i = 0; i < n; i++)
            sum = ((sum << 20) | (sum >> 12)) + w[i];
        if (sum != w[n]) {
            s3a_set_error("%s: checksum error; file-checksum %08x, computed %08x", path,
                          w[n], sum);
            free(w);
            return S3A_EIO;
        }
    }
    *words = w;
    *n_words = n;
    return S3A_OK;
}

/* ------------------------------------------------------------------ */
/* logmath                                                             */
/* ------------------------------------------------------------------ */
static int32_t
addtab_entry(const s3a_logmath_t *lm, double byx)
{
    /* round(log_base(1 + base^-d)) at the table's resolution */
    double lobyx = log(1.0 + byx) * lm->inv_log_of_base;
    return (int32_t)(lobyx + 0.5 * (1 << lm->shift)) >> lm->shift;
}

s3a_logmath_t *
s3a_logmath_init(double base, int32_t shift, int32_t use_table)
{
    s3a_logmath_t *lm;
    uint32_t maxyx, cap, n_raw, n, i;
    int32_t *raw;
    double byx;

    if (base <= 1.0) {
        s3a_set_error("logmath base must be greater than 1.0");
        return NULL;
    }
    if ((lm = (s3a_logmath_t *)calloc(1, sizeof *lm)) == NULL)
        return NULL;
    lm->base = base;
    lm->log_of_base = log(base);
    lm->log10_of_base = log10(base);
    lm->inv_log_of_base = 1.0 / lm->log_of_base;
    lm->inv_log10_of_base = 1.0 / lm->log10_of_base;
    lm->shift = shift;
    lm->zero = S3A_MAX_NEG_INT32 >> (shift + 2);
    if (!use_table)
        return lm;

    maxyx = (uint32_t)(log(2.0) / log(base) + 0.5) >> shift;
    lm->width = (maxyx < 256) ? 1 : (maxyx < 65536) ? 2 : 4;

    /* generate the un-shifted sequence k(d), d = 0,1,... until it hits 0,
     * dividing by base each step exactly as the reference does */
    cap = 1u << 16;
    raw = (int32_t *)malloc(sizeof(int32_t) * cap);
    byx = 1.0;
    for (n_raw = 0;; n_raw++) {
        if (n_raw == cap) {
            cap *= 2;
            raw = (int32_t *)realloc(raw, sizeof(int32_t) * cap);
        }
        raw[n_raw] = addtab_entry(lm, byx);
        if (raw[n_raw] <= 0)
            break;
        byx /= base;
    }
    /* n_raw = index of the first non-positive entry */
    n = n_raw >> shift;
    if (n < 255)
        n = 255;
    lm->table_size = n + 1;
    lm->table = (uint32_t *)calloc(lm->table_size, sizeof(uint32_t));
    /* slot j keeps the first value that maps to it unless that value is 0 */
    for (i = 0; i <= n_raw; i++) {
        uint32_t j = i >> shift, v = (uint32_t)raw[i];
        if (lm->width == 1) v &= 0xffu;
        else if (lm->width == 2) v &= 0xffffu;
        if (lm->table[j] == 0)
            lm->table[j] = v;
    }
    free(raw);
    return lm;
}

s3a_logmath_t *
s3a_logs3_init(double base, int32_t breport, int32_t blogtable)
{
    s3a_logmath_t *lm = s3a_logmath_init(base, 0, blogtable);
    if (lm && breport)
        fprintf(stderr, "INFO: Log-Add table size = %u x %d >> %d\n", lm->table_size,
                lm->width, lm->shift);
    return lm;
}

void
s3a_logmath_free(s3a_logmath_t *lm)
{
    if (lm) {
        free(lm->table);
        free(lm);
    }
}

int32_t
s3a_logmath_log(const s3a_logmath_t *lm, double p)
{
    if (p <= 0)
        return lm->zero;
    return (int32_t)(log(p) * lm->inv_log_of_base) >> lm->shift;
}

double
s3a_logmath_exp(const s3a_logmath_t *lm, int32_t logb_p)
{
    return pow(lm->base, (double)(int32_t)((uint32_t)logb_p << lm->shift));
}

int32_t
s3a_logmath_add(const s3a_logmath_t *lm, int32_t x, int32_t y)
{
    int32_t hi, lo, d;
    if (x <= lm->zero)
        return y;
    if (y <= lm->zero)
        return x;
    if (lm->table == NULL)
        return s3a_logmath_log(lm, s3a_logmath_exp(lm, x) + s3a_logmath_exp(lm, y));
    hi = x > y ? x : y;
    lo = x > y ? y : x;
    d = (int32_t)((uint32_t)hi - (uint32_t)lo);
    if (d < 0 || (uint32_t)d >= lm->table_size)
        return hi;
    return hi + (int32_t)lm->table[d];
}

int32_t
s3a_logmath_ln_to_log(const s3a_logmath_t *lm, double log_p)
{
    return (int32_t)(log_p * lm->inv_log_of_base) >> lm->shift;
}

double
s3a_logmath_log_to_ln(const s3a_logmath_t *lm, int32_t logb_p)
{
    return (double)(int32_t)((uint32_t)logb_p << lm->shift) * lm->log_of_base;
}

int32_t
s3a_logmath_log10_to_log(const s3a_logmath_t *lm, double log_p)
{
    return (int32_t)(log_p * lm->inv_log10_of_base) >> lm->shift;
}

double
s3a_logmath_get_base(const s3a_logmath_t *lm)
{
    return lm->base;
}

int32_t
s3a_logmath_get_zero(const s3a_logmath_t *lm)
{
    return lm->zero;
}

int32_t
s3a_logmath_get_table_shape(const s3a_logmath_t *lm, uint32_t *out_size, uint32_t *out_width,
                            uint32_t *out_shift)
{
    if (lm->table == NULL)
        return S3A_EINVAL;
    if (out_size) *out_size = lm->table_size;
    if (out_width) *out_width = (uint32_t)lm->width;
    if (out_shift) *out_shift = (uint32_t)lm->shift;
    return S3A_OK;
}

int32_t
s3a_logmath_copy_table(const s3a_logmath_t *lm, uint32_t *out, uint32_t size)
{
    if (lm->table == NULL || size < lm->table_size)
        return S3A_EINVAL;
    memcpy(out, lm->table, sizeof(uint32_t) * lm->table_size);
    return S3A_OK;
}

int32_t
s3a_logs3(const s3a_logmath_t *lm, double p)
{
    if (p <= 0.0)
        return S3A_LOGPROB_ZERO;
    return s3a_logmath_log(lm, p);
}

/* ------------------------------------------------------------------ */
/* acoustic model: host half of mgau_init                              */
/* ------------------------------------------------------------------ */
static int
all_zero(const float *v, int32_t n)
{
    int32_t i;
    for (i = 0; i < n; i++)
        if (v[i] != 0.0)
            return 0;
    return 1;
}

static int
any_nan(const float *v, int32_t n)
{
    int32_t i;
    for (i = 0; i < n; i++)
        if (isnan(v[i]))
            return 1;
    return 0;
}

/* sum-normalise in double, store back as float (vector.c:105-123) */
static double
normalise(float *v, int32_t n)
{
    double sum = 0.0, f;
    int32_t i;
    for (i = 0; i < n; i++)
        sum += v[i];
    if (sum != 0.0) {
        f = 1.0 / sum;
        for (i = 0; i < n; i++)
            v[i] = (float)((double)v[i] * f);
    }
    return sum;
}

/* floor the non-zero entries (vector.c:138-145) */
static void
floor_nonzero(float *v, int32_t n, double flr)
{
    int32_t i;
    for (i = 0; i < n; i++)
        if (v[i] != 0.0 && v[i] < flr)
            v[i] = (float)flr;
}

s3a_mgau_model_t *
s3a_mgau_host_init(const float *mean, const float *var, const float *mixw, int32_t n_mgau,
                   int32_t n_density, int32_t veclen, double varfloor, double mixwfloor,
                   int32_t precomp, s3a_logmath_t *lm)
{
    s3a_mgau_model_t *g;
    size_t ng = (size_t)n_mgau * n_density, nv = ng * veclen;
    float *w;
    int32_t m, c, i;

    if (n_mgau <= 0 || n_density <= 0 || veclen <= 0 || !lm || !mean || !var || !mixw) {
        s3a_set_error("s3a_mgau_init: bad arguments");
        return NULL;
    }
    g = (s3a_mgau_model_t *)calloc(1, sizeof *g);
    g->n_mgau = n_mgau;
    g->max_comp = n_density;
    g->veclen = veclen;
    g->lm = lm;
    g->n_comp = (int32_t *)calloc(n_mgau, sizeof(int32_t));
    g->mean = (float *)calloc(nv, sizeof(float));
    g->prec = (float *)calloc(nv, sizeof(float));
    g->lrd = (float *)calloc(ng, sizeof(float));
    g->mixw = (int32_t *)calloc(ng, sizeof(int32_t));
    w = (float *)malloc(sizeof(float) * n_density);
    if (!g->n_comp || !g->mean || !g->prec || !g->lrd || !g->mixw || !w) {
        free(w);
        s3a_mgau_host_free(g);
        s3a_set_error("s3a_mgau_init: out of memory");
        return NULL;
    }

    for (m = 0; m < n_mgau; m++) {
        const float *m_in = mean + (size_t)m * n_density * veclen;
        const float *v_in = var + (size_t)m * n_density * veclen;
        int32_t *mixw_m = g->mixw + (size_t)m * n_density;
        int32_t kept = 0;

        /* 1. mixture weights of this senone -> logs3 (before compaction, as
         *    mgau_mixw_read runs before mgau_uninit_compact) */
        memcpy(w, mixw + (size_t)m * n_density, sizeof(float) * n_density);
        if (all_zero(w, n_density)) {
            for (c = 0; c < n_density; c++)
                w[c] = 0.0f;
        }
        else {
            floor_nonzero(w, n_density, mixwfloor);
            normalise(w, n_density);
        }

        /* 2. keep initialised components only, in order */
        for (c = 0; c < n_density; c++) {
            const float *mc = m_in + (size_t)c * veclen, *vc = v_in + (size_t)c * veclen;
            float *m_out, *p_out;
            double lrd;
            if (any_nan(mc, veclen) || any_nan(vc, veclen) || all_zero(vc, veclen))
                continue;
            m_out = g->mean + ((size_t)m * n_density + kept) * veclen;
            p_out = g->prec + ((size_t)m * n_density + kept) * veclen;
            memcpy(m_out, mc, sizeof(float) * veclen);
            memcpy(p_out, vc, sizeof(float) * veclen);
            mixw_m[kept] = (w[c] != 0.0) ? s3a_logs3(lm, w[c]) : S3A_LOGPROB_ZERO;

            /* 3. variance floor, then 4. precompute, per kept component */
            if (varfloor > 0.0)
                for (i = 0; i < veclen; i++)
                    if (p_out[i] < varfloor)
                        p_out[i] = (float)varfloor;
            if (precomp) {
                lrd = 0.0;
                for (i = 0; i < veclen; i++) {
                    lrd += log(p_out[i]);
                    p_out[i] = (float)(1.0 / (p_out[i] * 2.0));
                }
                lrd += veclen * log(2.0 * M_PI);
                g->lrd[(size_t)m * n_density + kept] = (float)(-0.5 * lrd);
            }
            kept++;
        }
        g->n_comp[m] = kept;
        for (c = kept; c < n_density; c++)
            mixw_m[c] = S3A_LOGPROB_ZERO;
    }
    free(w);

    g->distfloor = s3a_logmath_log_to_ln(lm, S3A_LOGPROB_ZERO);
    g->f = 1.0 / log(lm->base);
    g->precision = S3A_GMM_EXACT;
    return g;
}

void
s3a_mgau_host_free(s3a_mgau_model_t *g)
{
    if (!g)
        return;
    free(g->n_comp);
    free(g->mean);
    free(g->prec);
    free(g->lrd);
    free(g->mixw);
    free(g);
}

/* parse a means / variances payload; returns pointer into words */
static int32_t
parse_gau(const char *path, const uint32_t *w, size_t nw, int32_t *n_mgau, int32_t *n_density,
          int32_t *veclen, const float **data)
{
    int32_t n_feat, n;
    if (nw < 5) {
        s3a_set_error("%s: truncated header", path);
        return S3A_EIO;
    }
    *n_mgau = (int32_t)w[0];
    n_feat = (int32_t)w[1];
    *n_density = (int32_t)w[2];
    if (n_feat != 1) {
        s3a_set_error("%s: #feature streams(%d) != 1 in continuous HMM", path, n_feat);
        return S3A_EUNSUP;
    }
    *veclen = (int32_t)w[3];
    n = (int32_t)w[4];
    if (*n_mgau <= 0 || *n_density <= 0 || *veclen <= 0) {
        s3a_set_error("%s: bad dimensions", path);
        return S3A_EIO;
    }
    if ((int64_t)n == (int64_t)*n_mgau * *n_density * *veclen * *veclen && *veclen > 1) {
        s3a_set_error("%s: full covariance matrices are not supported", path);
        return S3A_EUNSUP;
    }
    if ((int64_t)n != (int64_t)*n_mgau * *n_density * *veclen) {
        s3a_set_error("%s: #float32s(%d) doesn't match dimensions: %d x %d x %d", path, n,
                      *n_mgau, *n_density, *veclen);
        return S3A_EIO;
    }
    if (nw != 5 + (size_t)n) {
        s3a_set_error("%s: %s data than expected", path, nw > 5 + (size_t)n ? "more" : "less");
        return S3A_EIO;
    }
    *data = (const float *)(w + 5);
    return S3A_OK;
}

s3a_mgau_model_t *
s3a_mgau_init_arrays(const float *mean, const float *var, const float *mixw, int32_t n_mgau,
                     int32_t n_density, int32_t veclen, double varfloor, double mixwfloor,
                     int32_t precomp, s3a_logmath_t *logmath)
{
    s3a_mgau_model_t *g = s3a_mgau_host_init(mean, var, mixw, n_mgau, n_density, veclen,
                                             varfloor, mixwfloor, precomp, logmath);
    if (g && s3a_mgau_dev_create(g) != S3A_OK) {
        s3a_mgau_free(g);           /* device half (whatever dev_create got to) + host half */
        return NULL;
    }
    return g;
}

static s3a_mgau_model_t *
mgau_from_files(const char *meanfile, const char *varfile, double varfloor, const char *mixwfile,
                double mixwfloor, int32_t precomp, s3a_logmath_t *logmath, int upload)
{
    uint32_t *wm = NULL, *wv = NULL, *ww = NULL;
    size_t nm, nv, nw;
    int32_t S, C, D, S2, C2, D2, S3, n_feat, C3, n;
    const float *mean, *var;
    s3a_mgau_model_t *g = NULL;

    if (!meanfile || !varfile || !mixwfile || !logmath || varfloor < 0.0 || mixwfloor < 0.0) {
        s3a_set_error("s3a_mgau_init: bad arguments");
        return NULL;
    }
    if (s3a_bio_read(meanfile, "1.0", &wm, &nm) != S3A_OK
        || s3a_bio_read(varfile, "1.0", &wv, &nv) != S3A_OK
        || s3a_bio_read(mixwfile, "1.0", &ww, &nw) != S3A_OK)
        goto done;
    if (parse_gau(meanfile, wm, nm, &S, &C, &D, &mean) != S3A_OK
        || parse_gau(varfile, wv, nv, &S2, &C2, &D2, &var) != S3A_OK)
        goto done;
    if (S2 != S || C2 != C || D2 != D) {
        s3a_set_error("%s: dimensions %dx%dx%d don't match those of means %dx%dx%d", varfile,
                      S2, C2, D2, S, C, D);
        goto done;
    }
    if (nw < 4) {
        s3a_set_error("%s: truncated header", mixwfile);
        goto done;
    }
    S3 = (int32_t)ww[0]; n_feat = (int32_t)ww[1]; C3 = (int32_t)ww[2]; n = (int32_t)ww[3];
    if (n_feat != 1) {
        s3a_set_error("%s: #feature streams(%d) != 1 in continuous HMM", mixwfile, n_feat);
        goto done;
    }
    if (S3 != S || C3 != C || (int64_t)n != (int64_t)S * C || nw != 4 + (size_t)n) {
        s3a_set_error("%s: %d x %d mixture weights don't match mean/var parameters %d x %d",
                      mixwfile, S3, C3, S, C);
        goto done;
    }
    if (upload)
        g = s3a_mgau_init_arrays(mean, var, (const float *)(ww + 4), S, C, D, varfloor,
                                 mixwfloor, precomp, logmath);
    else
        g = s3a_mgau_host_init(mean, var, (const float *)(ww + 4), S, C, D, varfloor, mixwfloor,
                               precomp, logmath);
done:
    free(wm);
    free(wv);
    free(ww);
    return g;
}

s3a_mgau_model_t *
s3a_mgau_init(const char *meanfile, const char *varfile, double varfloor, const char *mixwfile,
              double mixwfloor, int32_t precomp, const char *senmgau, int32_t comp_type,
              s3a_logmath_t *logmath)
{
    if (!senmgau || strcmp(senmgau, ".cont.") != 0) {
        s3a_set_error("s3a_mgau_init: only -senmgau .cont. is supported (got %s)",
                      senmgau ? senmgau : "NULL");
        return NULL;
    }
    if (comp_type != S3A_MIX_INT_FLOAT_COMP) {
        s3a_set_error("s3a_mgau_init: only MIX_INT_FLOAT_COMP is supported");
        return NULL;
    }
    return mgau_from_files(meanfile, varfile, varfloor, mixwfile, mixwfloor, precomp, logmath, 1);
}

s3a_mgau_model_t *
s3a_mgau_load_host(const char *meanfile, const char *varfile, double varfloor,
                   const char *mixwfile, double mixwfloor, int32_t precomp,
                   s3a_logmath_t *logmath)
{
    return mgau_from_files(meanfile, varfile, varfloor, mixwfile, mixwfloor, precomp, logmath, 0);
}

void
s3a_mgau_free(s3a_mgau_model_t *g)
{
    if (!g)
        return;
    if (g->dev)
        s3a_mgau_dev_destroy(g);
    s3a_mgau_host_free(g);
}

int32_t s3a_mgau_n_mgau(const s3a_mgau_model_t *g) { return g->n_mgau; }
int32_t s3a_mgau_max_comp(const s3a_mgau_model_t *g) { return g->max_comp; }
int32_t s3a_mgau_veclen(const s3a_mgau_model_t *g) { return g->veclen; }
int32_t s3a_mgau_n_comp(const s3a_mgau_model_t *g, int32_t m)
{
    return (m >= 0 && m < g->n_mgau) ? g->n_comp[m] : S3A_EINVAL;
}
double s3a_mgau_distfloor(const s3a_mgau_model_t *g) { return g->distfloor; }

int32_t
s3a_mgau_get_params(const s3a_mgau_model_t *g, float *mean, float *prec, float *lrd,
                    int32_t *mixw, int32_t *n_comp)
{
    size_t ng = (size_t)g->n_mgau * g->max_comp;
    if (mean) memcpy(mean, g->mean, sizeof(float) * ng * g->veclen);
    if (prec) memcpy(prec, g->prec, sizeof(float) * ng * g->veclen);
    if (lrd) memcpy(lrd, g->lrd, sizeof(float) * ng);
    if (mixw) memcpy(mixw, g->mixw, sizeof(int32_t) * ng);
    if (n_comp) memcpy(n_comp, g->n_comp, sizeof(int32_t) * g->n_mgau);
    return S3A_OK;
}

/* ------------------------------------------------------------------ */
/* multi-stream scorer: host half of ms_mgau_init                      */
/* gauden_init + gauden_dist_precompute (ms_gauden.c:330-476),         */
/* senone_init / senone_mixw_read (ms_senone.c:212-417),               */
/* ms_mgau_init (ms_mgau.c:149-227)                                    */
/* ------------------------------------------------------------------ */
static size_t
ms_off(const s3a_ms_mgau_t *ms, int32_t m, int32_t f, int32_t d)
{
    return (size_t)m * ms->n_density * ms->veclen + (size_t)ms->n_density * ms->featoff[f]
        + (size_t)d * ms->featlen[f];
}

s3a_ms_mgau_t *
s3a_ms_host_init(const float *mean, const float *var, const float *mixw, int32_t n_mgau, int32_t n_feat,
                 int32_t n_density, const int32_t *featlen, int32_t n_sen, const int32_t *sen2mgau,
                 double varfloor_d, double mixwfloor, int32_t topn, s3a_logmath_t *lm)
{
    s3a_ms_mgau_t *ms;
    float varfloor = (float)varfloor_d;         /* gauden_init(..., float32 varfloor, ...) */
    float *row;
    int32_t m, f, d, i, s, c;
    size_t n;

    if (!mean || !var || !mixw || !featlen || !lm || n_mgau <= 0 || n_feat <= 0 || n_density <= 0
        || n_sen <= 1 || !(varfloor_d > 0.0) || !(mixwfloor > 0.0 && mixwfloor < 1.0)) {
        s3a_set_error("s3a_ms_mgau_init: bad arguments");
        return NULL;
    }
    if ((ms = (s3a_ms_mgau_t *)calloc(1, sizeof *ms)) == NULL) return NULL;
    ms->n_mgau = n_mgau; ms->n_feat = n_feat; ms->n_density = n_density; ms->n_sen = n_sen; ms->lm = lm;
    ms->featlen = (int32_t *)malloc(sizeof(int32_t) * n_feat);
    ms->featoff = (int32_t *)malloc(sizeof(int32_t) * (n_feat + 1));
    for (f = 0; f < n_feat; f++) {
        if (featlen[f] <= 0) { s3a_set_error("s3a_ms_mgau_init: bad stream length"); s3a_ms_host_free(ms); return NULL; }
        ms->featlen[f] = featlen[f];
        ms->featoff[f] = ms->veclen;
        ms->veclen += featlen[f];
    }
    ms->featoff[n_feat] = ms->veclen;
    n = (size_t)n_mgau * n_density * ms->veclen;
    ms->mean = (float *)malloc(sizeof(float) * n);
    ms->prec = (float *)malloc(sizeof(float) * n);
    ms->det = (float *)calloc((size_t)n_mgau * n_feat * n_density, sizeof(float));
    ms->pdf = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_sen * n_feat * n_density);
    ms->mgau = (int32_t *)malloc(sizeof(int32_t) * n_sen);
    row = (float *)malloc(sizeof(float) * n_density);
    if (!ms->mean || !ms->prec || !ms->det || !ms->pdf || !ms->mgau || !row) {
        free(row); s3a_ms_host_free(ms); s3a_set_error("s3a_ms_mgau_init: out of memory"); return NULL;
    }
    memcpy(ms->mean, mean, sizeof(float) * n);
    memcpy(ms->prec, var, sizeof(float) * n);
    /* determinant term accumulated in float32, precision = (float32)(1 / (2 var)) */
    for (m = 0; m < n_mgau; m++)
        for (f = 0; f < n_feat; f++)
            for (d = 0; d < n_density; d++) {
                float *varp = ms->prec + ms_off(ms, m, f, d);
                float *detp = &ms->det[((size_t)m * n_feat + f) * n_density + d];
                *detp = (float)0.0;
                for (i = 0; i < featlen[f]; i++, varp++) {
                    if (*varp < varfloor)
                        *varp = varfloor;
                    *detp += (float)(log(*varp));
                    *varp = (float)(1.0 / (*varp * 2.0));
                }
                *detp += (float)(featlen[f] * log(2.0 * M_PI));
                *detp *= (float)0.5;
            }
    ms->min_density = s3a_logmath_log_to_ln(lm, S3A_LOGPROB_ZERO);
    /* senone weights: normalise, floor, normalise, -logs3 (TRUNCATE_LOGPDF is not defined) */
    for (s = 0; s < n_sen; s++)
        for (f = 0; f < n_feat; f++) {
            memcpy(row, mixw + ((size_t)s * n_feat + f) * n_density, sizeof(float) * n_density);
            normalise(row, n_density);
            for (c = 0; c < n_density; c++)
                if (row[c] < mixwfloor) row[c] = (float)mixwfloor;
            normalise(row, n_density);
            for (c = 0; c < n_density; c++)
                ms->pdf[((size_t)s * n_feat + f) * n_density + c] = -(s3a_logs3(lm, row[c]));
        }
    free(row);
    ms->one_to_one = (sen2mgau == NULL);
    for (s = 0; s < n_sen; s++) {
        ms->mgau[s] = sen2mgau ? sen2mgau[s] : s;
        if (ms->mgau[s] < 0 || ms->mgau[s] >= n_mgau) {
            s3a_set_error("s3a_ms_mgau_init: senone %d needs codebook %d of %d", s, ms->mgau[s], n_mgau);
            s3a_ms_host_free(ms);
            return NULL;
        }
    }
    ms->topn = (topn <= 0 || topn > n_density) ? n_density : topn;     /* ms_mgau.c:214-219 */
    return ms;
}

void
s3a_ms_host_free(s3a_ms_mgau_t *ms)
{
    if (!ms) return;
    free(ms->featlen); free(ms->featoff); free(ms->mean); free(ms->prec); free(ms->det);
    free(ms->pdf); free(ms->mgau);
    free(ms);
}

s3a_ms_mgau_t *
s3a_ms_mgau_init_arrays(const float *mean, const float *var, const float *mixw, int32_t n_mgau,
                        int32_t n_feat, int32_t n_density, const int32_t *featlen, int32_t n_sen,
                        const int32_t *sen2mgau, double varfloor, double mixwfloor, int32_t topn,
                        s3a_logmath_t *lm)
{
    s3a_ms_mgau_t *ms = s3a_ms_host_init(mean, var, mixw, n_mgau, n_feat, n_density, featlen, n_sen,
                                         sen2mgau, varfloor, mixwfloor, topn, lm);
    if (ms && s3a_ms_dev_create(ms) != S3A_OK) {
        s3a_ms_host_free(ms);
        return NULL;
    }
    return ms;
}

/* multi-stream means / variances payload (gauden_param_read, ms_gauden.c:205-312) */
static int32_t
parse_gau_streams(const char *path, const uint32_t *w, size_t nw, int32_t *n_mgau, int32_t *n_feat,
                  int32_t *n_density, const int32_t **featlen, const float **data)
{
    int64_t tot = 0;
    int32_t f;
    if (nw < 4 || (int32_t)w[1] <= 0 || nw < 4 + (size_t)w[1]) { s3a_set_error("%s: truncated header", path); return S3A_EIO; }
    *n_mgau = (int32_t)w[0]; *n_feat = (int32_t)w[1]; *n_density = (int32_t)w[2];
    *featlen = (const int32_t *)(w + 3);
    for (f = 0; f < *n_feat; f++) {
        if ((*featlen)[f] <= 0) { s3a_set_error("%s: stream %d has length %d", path, f, (*featlen)[f]); return S3A_EIO; }
        tot += (*featlen)[f];
    }
    if (*n_mgau <= 0 || *n_density <= 0 || tot <= 0
        || (int64_t)(int32_t)w[3 + *n_feat] != (int64_t)*n_mgau * *n_density * tot
        || nw != 4 + (size_t)*n_feat + (size_t)w[3 + *n_feat]) {
        s3a_set_error("%s: #float32s doesn't match dimensions", path);
        return S3A_EIO;
    }
    *data = (const float *)(w + 4 + *n_feat);
    return S3A_OK;
}

s3a_ms_mgau_t *
s3a_ms_mgau_init(const char *meanfile, const char *varfile, double varfloor, const char *mixwfile,
                 double mixwfloor, int32_t precomp, const char *senmgau, const char *lambdafile,
                 int32_t topn, s3a_logmath_t *lm)
{
    uint32_t *wm = NULL, *wv = NULL, *ww = NULL;
    size_t nm, nv, nw;
    int32_t M, F, C, M2, F2, C2, S, f, semi;
    const int32_t *fl, *fl2;
    const float *mean, *var;
    int32_t *map = NULL;
    s3a_ms_mgau_t *ms = NULL;

    if (!meanfile || !varfile || !mixwfile || !senmgau || !lm) { s3a_set_error("s3a_ms_mgau_init: bad arguments"); return NULL; }
    if (lambdafile) { s3a_set_error("s3a_ms_mgau_init: CD/CI interpolation weights (-lambda) are not supported"); return NULL; }
    if (!precomp) { s3a_set_error("s3a_ms_mgau_init: precomp must be 1"); return NULL; }
    semi = strcmp(senmgau, ".semi.") == 0;
    if (!semi && strcmp(senmgau, ".s3cont.") != 0 && strcmp(senmgau, ".cont.") != 0) {
        s3a_set_error("s3a_ms_mgau_init: -senmgau %s: senone-codebook mapping FILES are not supported", senmgau);
        return NULL;
    }
    if (s3a_bio_read(meanfile, "1.0", &wm, &nm) != S3A_OK || s3a_bio_read(varfile, "1.0", &wv, &nv) != S3A_OK
        || s3a_bio_read(mixwfile, "1.0", &ww, &nw) != S3A_OK)
        goto done;
    if (parse_gau_streams(meanfile, wm, nm, &M, &F, &C, &fl, &mean) != S3A_OK
        || parse_gau_streams(varfile, wv, nv, &M2, &F2, &C2, &fl2, &var) != S3A_OK)
        goto done;
    if (M2 != M || F2 != F || C2 != C || memcmp(fl, fl2, sizeof(int32_t) * F) != 0) {
        s3a_set_error("%s: dimensions don't match those of %s", varfile, meanfile);
        goto done;
    }
    if (nw < 4 || (int64_t)(int32_t)ww[3] != (int64_t)(int32_t)ww[0] * (int32_t)ww[1] * (int32_t)ww[2]
        || nw != 4 + (size_t)ww[3]) {
        s3a_set_error("%s: #float32s doesn't match dimensions", mixwfile);
        goto done;
    }
    S = (int32_t)ww[0];
    if ((int32_t)ww[1] != F || (int32_t)ww[2] != C) {
        s3a_set_error("%s: %d streams x %d codewords don't match the codebooks' %d x %d", mixwfile,
                      (int32_t)ww[1], (int32_t)ww[2], F, C);
        goto done;
    }
    if (semi) {                             /* all senones share codebook 0 (ms_senone.c:390-397) */
        map = (int32_t *)calloc(S, sizeof(int32_t));
    }
    else if (S > M) {
        s3a_set_error("Senones need more codebooks (%d) than present (%d)", S, M);
        goto done;
    }
    (void)f;
    ms = s3a_ms_mgau_init_arrays(mean, var, (const float *)(ww + 4), M, F, C, fl, S, map, varfloor, mixwfloor,
                                 topn, lm);
done:
    free(map); free(wm); free(wv); free(ww);
    return ms;
}

/* ------------------------------------------------------------------ */
/* pocketsphinx's continuous scorer: host half of ms_mgau_init          */
/* (pocketsphinx/src/libpocketsphinx/ms_mgau.c:75-138, ms_gauden.c:307-398, ms_senone.c:150-352) */
/* ------------------------------------------------------------------ */
#define PS_SENSCR_SHIFT 10

static void
s3a_ps_host_free(s3a_ps_mgau_t *ps)
{
    if (!ps) return;
    free(ps->featlen); free(ps->featoff); free(ps->mean); free(ps->prec); free(ps->det); free(ps->pdf);
    free(ps->mgau); free(ps->flags);
    s3a_logmath_free(ps->lm); s3a_logmath_free(ps->lm8);
    free(ps);
}

s3a_ps_mgau_t *
s3a_ps_ms_mgau_init_arrays(const float *mean, const float *var, const float *mixw, int32_t n_mgau,
                           int32_t n_feat, int32_t n_density, const int32_t *featlen, int32_t n_sen,
                           const int32_t *sen2mgau, double varfloor_d, double mixwfloor, int32_t topn,
                           int32_t aw, double logbase)
{
    s3a_ps_mgau_t *ps;
    float varfloor = (float)varfloor_d, *row;
    int32_t m, f, d, i, s, c;
    size_t n;
    if (!mean || !var || !mixw || !featlen || n_mgau <= 0 || n_feat <= 0 || n_density <= 0 || n_sen <= 1
        || !(varfloor_d > 0.0) || !(mixwfloor > 0.0 && mixwfloor < 1.0) || aw == 0 || !(logbase > 1.0)) {
        s3a_set_error("s3a_ps_ms_mgau_init: bad arguments");
        return NULL;
    }
    for (f = 0; f < n_feat; f++)
        if (featlen[f] <= 0) { s3a_set_error("s3a_ps_ms_mgau_init: stream %d has length %d", f, featlen[f]); return NULL; }
    if ((ps = (s3a_ps_mgau_t *)calloc(1, sizeof *ps)) == NULL) return NULL;
    ps->n_mgau = n_mgau; ps->n_feat = n_feat; ps->n_density = n_density; ps->n_sen = n_sen; ps->aw = aw;
    ps->lm = s3a_logmath_init(logbase, 0, 0);                   /* acmod.c: logmath_init(-logbase, 0, FALSE) */
    ps->lm8 = s3a_logmath_init(logbase, PS_SENSCR_SHIFT, 1);    /* ms_senone.c:293 */
    ps->featlen = (int32_t *)malloc(sizeof(int32_t) * n_feat);
    ps->featoff = (int32_t *)malloc(sizeof(int32_t) * (n_feat + 1));
    for (f = 0; f < n_feat; f++) { ps->featlen[f] = featlen[f]; ps->featoff[f] = ps->veclen; ps->veclen += featlen[f]; }
    ps->featoff[n_feat] = ps->veclen;
    n = (size_t)n_mgau * n_density * ps->veclen;
    ps->mean = (float *)malloc(sizeof(float) * n);
    ps->prec = (float *)malloc(sizeof(float) * n);
    ps->det = (float *)calloc((size_t)n_mgau * n_feat * n_density, sizeof(float));
    ps->pdf = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_sen * n_feat * n_density);
    ps->mgau = (int32_t *)malloc(sizeof(int32_t) * n_sen);
    ps->flags = (uint8_t *)calloc(n_sen, 1);
    row = (float *)malloc(sizeof(float) * n_density);
    if (!ps->lm || !ps->lm8 || !ps->mean || !ps->prec || !ps->det || !ps->pdf || !ps->mgau || !ps->flags || !row) {
        free(row); s3a_ps_host_free(ps); s3a_set_error("s3a_ps_ms_mgau_init: out of memory"); return NULL;
    }
    memcpy(ps->mean, mean, sizeof(float) * n);
    memcpy(ps->prec, var, sizeof(float) * n);
    for (m = 0; m < n_mgau; m++)
        for (f = 0; f < n_feat; f++)
            for (d = 0; d < n_density; d++) {
                float *varp = ps->prec + (size_t)m * n_density * ps->veclen + (size_t)n_density * ps->featoff[f]
                    + (size_t)d * featlen[f];
                float *detp = &ps->det[((size_t)m * n_feat + f) * n_density + d];
                *detp = 0;
                for (i = 0; i < featlen[f]; i++, varp++) {
                    if (*varp < varfloor) *varp = varfloor;
                    *detp += (float)s3a_logmath_log(ps->lm, 1.0 / sqrt(*varp * 2.0 * M_PI));
                    *varp = (float)s3a_logmath_ln_to_log(ps->lm, (1.0 / (*varp * 2.0)));
                }
            }
    for (s = 0; s < n_sen; s++)
        for (f = 0; f < n_feat; f++) {
            memcpy(row, mixw + ((size_t)s * n_feat + f) * n_density, sizeof(float) * n_density);
            normalise(row, n_density);
            for (c = 0; c < n_density; c++) if (row[c] < mixwfloor) row[c] = (float)mixwfloor;
            normalise(row, n_density);
            for (c = 0; c < n_density; c++) {
                int32_t p = -(s3a_logmath_log(ps->lm, row[c]));
                p += (1 << (PS_SENSCR_SHIFT - 1)) - 1;          /* rounding before truncation */
                ps->pdf[((size_t)s * n_feat + f) * n_density + c] = (p < (255 << PS_SENSCR_SHIFT)) ? (p >> PS_SENSCR_SHIFT) : 255;
            }
        }
    free(row);
    ps->one_to_one = (sen2mgau == NULL);
    for (s = 0; s < n_sen; s++) {
        ps->mgau[s] = sen2mgau ? sen2mgau[s] : s;
        if (ps->mgau[s] < 0 || ps->mgau[s] >= n_mgau) {
            s3a_set_error("s3a_ps_ms_mgau_init: senone %d needs codebook %d of %d", s, ps->mgau[s], n_mgau);
            s3a_ps_host_free(ps);
            return NULL;
        }
    }
    ps->topn = (topn <= 0 || topn > n_density) ? n_density : topn;
    if (s3a_ps_dev_create(ps) != S3A_OK) { s3a_ps_host_free(ps); return NULL; }
    return ps;
}

s3a_ps_mgau_t *
s3a_ps_ms_mgau_init(const char *meanfile, const char *varfile, double varfloor, const char *mixwfile,
                    double mixwfloor, const char *senmgau, int32_t topn, int32_t aw, double logbase)
{
    uint32_t *wm = NULL, *wv = NULL, *ww = NULL;
    size_t nm, nv, nw;
    int32_t M, F, C, M2, F2, C2, S, semi;
    const int32_t *fl, *fl2;
    const float *mean, *var;
    int32_t *map = NULL;
    s3a_ps_mgau_t *ps = NULL;
    if (!meanfile || !varfile || !mixwfile) { s3a_set_error("s3a_ps_ms_mgau_init: bad arguments"); return NULL; }
    if (s3a_bio_read(meanfile, "1.0", &wm, &nm) != S3A_OK || s3a_bio_read(varfile, "1.0", &wv, &nv) != S3A_OK
        || s3a_bio_read(mixwfile, "1.0", &ww, &nw) != S3A_OK)
        goto done;
    if (parse_gau_streams(meanfile, wm, nm, &M, &F, &C, &fl, &mean) != S3A_OK
        || parse_gau_streams(varfile, wv, nv, &M2, &F2, &C2, &fl2, &var) != S3A_OK)
        goto done;
    if (M2 != M || F2 != F || C2 != C || memcmp(fl, fl2, sizeof(int32_t) * F) != 0) {
        s3a_set_error("Mixture-gaussians dimensions for means and variances differ");
        goto done;
    }
    if (nw < 4 || (int64_t)(int32_t)ww[3] != (int64_t)(int32_t)ww[0] * (int32_t)ww[1] * (int32_t)ww[2]
        || nw != 4 + (size_t)ww[3] || (int32_t)ww[1] != F || (int32_t)ww[2] != C) {
        s3a_set_error("%s: mixture weights don't match the codebooks", mixwfile);
        goto done;
    }
    S = (int32_t)ww[0];
    /* senone -> codebook mapping as senone_init decides it (ms_senone.c:296-333) */
    if (senmgau && strcmp(senmgau, ".semi.") && strcmp(senmgau, ".cont.") && strcmp(senmgau, ".s3cont.")) {
        s3a_set_error("s3a_ps_ms_mgau_init: -senmgau %s: mapping files and .ptm. are not supported", senmgau);
        goto done;
    }
    semi = senmgau ? strcmp(senmgau, ".semi.") == 0 : M == 1;
    if (semi) map = (int32_t *)calloc(S, sizeof(int32_t));
    else if (S > M) { s3a_set_error("Senones need more codebooks (%d) than present (%d)", S, M); goto done; }
    ps = s3a_ps_ms_mgau_init_arrays(mean, var, (const float *)(ww + 4), M, F, C, fl, S, map, varfloor, mixwfloor,
                                    topn, aw, logbase);
done:
    free(map); free(wm); free(wv); free(ww);
    return ps;
}

void
s3a_ps_ms_mgau_free(s3a_ps_mgau_t *ps)
{
    if (!ps) return;
    s3a_ps_dev_destroy(ps);
    s3a_ps_host_free(ps);
}

int32_t s3a_ps_ms_mgau_n_sen(const s3a_ps_mgau_t *ps) { return ps->n_sen; }
int32_t s3a_ps_ms_mgau_veclen(const s3a_ps_mgau_t *ps) { return ps->veclen; }

void
s3a_ms_mgau_free(s3a_ms_mgau_t *ms)
{
    if (!ms) return;
    s3a_ms_dev_destroy(ms);
    s3a_ms_host_free(ms);
}

int32_t s3a_ms_mgau_n_sen(const s3a_ms_mgau_t *ms) { return ms->n_sen; }
int32_t s3a_ms_mgau_topn(const s3a_ms_mgau_t *ms) { return ms->topn; }
int32_t s3a_ms_mgau_veclen(const s3a_ms_mgau_t *ms) { return ms->veclen; }


/* ------------------------------------------------------------------ */
/* transition matrices                                                 */
/* ------------------------------------------------------------------ */
s3a_tmat_t *
s3a_tmat_init_arrays(const float *tp, int32_t n_tmat, int32_t n_state, double tpfloor,
                     s3a_logmath_t *lm)
{
    s3a_tmat_t *t;
    int32_t n_dst = n_state + 1, i, j, k;
    float *row;

    if (!tp || n_tmat <= 0 || n_state <= 0 || !lm) {
        s3a_set_error("s3a_tmat_init: bad arguments");
        return NULL;
    }
    t = (s3a_tmat_t *)calloc(1, sizeof *t);
    t->n_tmat = n_tmat;
    t->n_state = n_state;
    t->tp = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_tmat * n_state * n_dst);
    row = (float *)malloc(sizeof(float) * n_dst);
    for (i = 0; i < n_tmat; i++)
        for (j = 0; j < n_state; j++) {
            int32_t *out = t->tp + ((size_t)i * n_state + j) * n_dst;
            memcpy(row, tp + ((size_t)i * n_state + j) * n_dst, sizeof(float) * n_dst);
            if (normalise(row, n_dst) == 0.0)
                fprintf(stderr, "WARNING: Normalization failed for tmat %d from state %d\n", i, j);
            floor_nonzero(row, n_dst, tpfloor);
            normalise(row, n_dst);
            for (k = 0; k < n_dst; k++)
                out[k] = (row[k] == 0.0) ? S3A_LOGPROB_ZERO : s3a_logs3(lm, row[k]);
        }
    free(row);
    /* tmat_chk_uppertri, tmat.c:126-140 */
    for (i = 0; i < n_tmat; i++)
        for (j = 1; j < n_state; j++)
            for (k = 0; k < j; k++)
                if (t->tp[((size_t)i * n_state + j) * n_dst + k] > S3A_LOGPROB_ZERO) {
                    s3a_set_error("tmat %d not upper triangular: [%d][%d]", i, j, k);
                    s3a_tmat_free(t);
                    return NULL;
                }
    return t;
}

s3a_tmat_t *
s3a_tmat_init_logs3(const int32_t *tp, int32_t n_tmat, int32_t n_state)
{
    s3a_tmat_t *t;
    size_t n;
    if (!tp || n_tmat <= 0 || n_state <= 0) {
        s3a_set_error("s3a_tmat_init_logs3: bad arguments");
        return NULL;
    }
    n = (size_t)n_tmat * n_state * (n_state + 1);
    t = (s3a_tmat_t *)calloc(1, sizeof *t);
    t->n_tmat = n_tmat;
    t->n_state = n_state;
    t->tp = (int32_t *)malloc(sizeof(int32_t) * n);
    memcpy(t->tp, tp, sizeof(int32_t) * n);
    return t;
}

s3a_tmat_t *
s3a_tmat_init(const char *tmatfile, double tpfloor, int32_t breport, s3a_logmath_t *lm)
{
    uint32_t *w = NULL;
    size_t nw;
    s3a_tmat_t *t = NULL;
    int32_t n_tmat, n_src, n_dst, n;

    if (breport)
        fprintf(stderr, "INFO: Reading HMM transition probability matrices: %s\n", tmatfile);
    if (s3a_bio_read(tmatfile, "1.0", &w, &nw) != S3A_OK)
        return NULL;
    if (nw < 4) {
        s3a_set_error("%s: truncated header", tmatfile);
        goto done;
    }
    n_tmat = (int32_t)w[0]; n_src = (int32_t)w[1]; n_dst = (int32_t)w[2]; n = (int32_t)w[3];
    if (n_dst != n_src + 1) {
        s3a_set_error("%s: #from-states(%d) != #to-states(%d)-1", tmatfile, n_src, n_dst);
        goto done;
    }
    if ((int64_t)n != (int64_t)n_tmat * n_src * n_dst || nw != 4 + (size_t)n) {
        s3a_set_error("%s: #float32s(%d) doesn't match dimensions: %d x %d x %d", tmatfile, n,
                      n_tmat, n_src, n_dst);
        goto done;
    }
    t = s3a_tmat_init_arrays((const float *)(w + 4), n_tmat, n_src, tpfloor, lm);
done:
    free(w);
    return t;
}

void
s3a_tmat_free(s3a_tmat_t *t)
{
    if (t) {
        free(t->tp);
        free(t);
    }
}

int32_t s3a_tmat_n_tmat(const s3a_tmat_t *t) { return t->n_tmat; }
int32_t s3a_tmat_n_state(const s3a_tmat_t *t) { return t->n_state; }

int32_t
s3a_tmat_get_tp(const s3a_tmat_t *t, int32_t *tp)
{
    memcpy(tp, t->tp, sizeof(int32_t) * (size_t)t->n_tmat * t->n_state * (t->n_state + 1));
    return S3A_OK;
}

/* ------------------------------------------------------------------ */
/* pocketsphinx's senone score dump (-senlogdir): the cross-decoder    */
/* score interchange format                         */
/* ------------------------------------------------------------------ */
/*
 * File = an s3 header (bio_writehdr, sphinxbase util/bio.c:154-183: "s3", the pairs version 0.1 /
 * mdef_file / n_sen / logbase, "endhdr", the 32-bit byte-order word) followed by frames
 * (acmod_write_scores, pocketsphinx acmod.c:885-923): int16 n_active; then, when every senone is
 * active, n_active int16 scores; otherwise n_active delta bytes (the list ps_mgau_frame_eval takes)
 * and the n_active scores of those senones.  Reading (acmod_read_senfh_header :805-836,
 * acmod_read_scores_internal :928-985) fills the senones that are not listed with SENSCR_DUMMY.
 */
#define S3A_SENSCR_DUMMY 0x7fff         /* acmod.h:76 */

struct s3a_senlog_s {
    FILE *fp;
    int32_t n_sen, swap, writing;
    double logbase;
};

s3a_senlog_t *
s3a_senlog_open_write(const char *path, const char *mdef_file, int32_t n_sen, double logbase)
{
    s3a_senlog_t *s;
    uint32_t magic = 0x11223344u;
    if (!path || n_sen <= 0 || n_sen > 32767) {     /* (n_active is an int16 in the file) */
        s3a_set_error("s3a_senlog_open_write: bad arguments (1..32767 senones)");
        return NULL;
    }
    if ((s = (s3a_senlog_t *)calloc(1, sizeof *s)) == NULL) return NULL;
    if ((s->fp = fopen(path, "wb")) == NULL) {
        s3a_set_error("cannot create %s", path);
        free(s);
        return NULL;
    }
    s->n_sen = n_sen; s->logbase = logbase; s->writing = 1;
    /* acmod_write_senfh_header (acmod.c:349-361): "%d" and "%f" */
    fprintf(s->fp, "s3\nversion 0.1\nmdef_file %s\nn_sen %d\nlogbase %f\nendhdr\n", mdef_file ? mdef_file : "(null)",
            n_sen, logbase);
    if (fwrite(&magic, 4, 1, s->fp) != 1) {
        s3a_set_error("%s: write failed", path);
        s3a_senlog_close(s);
        return NULL;
    }
    return s;
}

int32_t
s3a_senlog_write_frame(s3a_senlog_t *s, int32_t n_active, const uint8_t *active, const int16_t *senscr)
{
    int16_t n16 = (int16_t)n_active;
    int32_t i, n;
    if (!s || !s->writing || !senscr || n_active < 0 || n_active > s->n_sen || (n_active < s->n_sen && n_active > 0 && !active))
        return S3A_EINVAL;
    if (fwrite(&n16, 2, 1, s->fp) != 1) goto fail;
    if (n_active == s->n_sen) {
        if (fwrite(senscr, 2, (size_t)n_active, s->fp) != (size_t)n_active) goto fail;
    }
    else {
        if (n_active && fwrite(active, 1, (size_t)n_active, s->fp) != (size_t)n_active) goto fail;
        for (i = n = 0; i < n_active; ++i) {
            n += active[i];
            if (n >= s->n_sen) { s3a_set_error("s3a_senlog_write_frame: the active list runs past the last senone"); return S3A_EINVAL; }
            if (fwrite(senscr + n, 2, 1, s->fp) != 1) goto fail;
        }
    }
    return S3A_OK;
fail:
    s3a_set_error("s3a_senlog_write_frame: write failed");
    return S3A_EIO;
}

s3a_senlog_t *
s3a_senlog_open_read(const char *path, int32_t *n_sen, double *logbase)
{
    s3a_senlog_t *s;
    char line[16384], name[256], val[4096];
    uint32_t magic;
    if (!path) return NULL;
    if ((s = (s3a_senlog_t *)calloc(1, sizeof *s)) == NULL) return NULL;
    if ((s->fp = fopen(path, "rb")) == NULL) {
        s3a_set_error("cannot open %s", path);
        free(s);
        return NULL;
    }
    if (fgets(line, sizeof line, s->fp) == NULL || strcmp(line, "s3\n") != 0) goto bad;
    for (;;) {
        if (fgets(line, sizeof line, s->fp) == NULL) goto bad;
        name[0] = val[0] = 0;
        if (sscanf(line, "%255s %4095s", name, val) < 1) continue;
        if (strcmp(name, "endhdr") == 0) break;
        if (strcmp(name, "n_sen") == 0) s->n_sen = atoi(val);
        else if (strcmp(name, "logbase") == 0) s->logbase = atof(val);
    }
    if (fread(&magic, 4, 1, s->fp) != 1) goto bad;
    if (magic == 0x11223344u) s->swap = 0;
    else if (bswap32(magic) == 0x11223344u) s->swap = 1;
    else goto bad;
    if (s->n_sen <= 0 || s->n_sen > 32767) goto bad;
    if (n_sen) *n_sen = s->n_sen;
    if (logbase) *logbase = s->logbase;
    return s;
bad:
    s3a_set_error("%s: not a senone score dump", path);
    s3a_senlog_close(s);
    return NULL;
}

static int16_t
senlog_swap16(const s3a_senlog_t *s, int16_t v)
{
    uint16_t u = (uint16_t)v;
    return s->swap ? (int16_t)(uint16_t)((u >> 8) | (u << 8)) : v;
}

/* returns 1 and fills senscr[n_sen] (+ active[n_active], *n_active when given), 0 at the end of the file */
int32_t
s3a_senlog_read_frame(s3a_senlog_t *s, int16_t *senscr, uint8_t *active, int32_t *n_active)
{
    int16_t n16;
    int32_t i, n, na;
    if (!s || s->writing || !senscr) return S3A_EINVAL;
    if (fread(&n16, 2, 1, s->fp) != 1) return 0;
    na = senlog_swap16(s, n16);
    if (na < 0 || na > s->n_sen) { s3a_set_error("senone score dump: %d active of %d senones", na, s->n_sen); return S3A_EIO; }
    if (n_active) *n_active = na;
    if (na == s->n_sen) {
        if (fread(senscr, 2, (size_t)na, s->fp) != (size_t)na) return 0;
        for (i = 0; i < na; i++) senscr[i] = senlog_swap16(s, senscr[i]);
        return 1;
    }
    {
        uint8_t *lst = active ? active : (uint8_t *)malloc((size_t)(na ? na : 1));
        int32_t rc = 1;
        if (na && fread(lst, 1, (size_t)na, s->fp) != (size_t)na) rc = 0;
        /* (acmod_read_scores_internal leaves senone 0 untouched unless it is listed; a dummy is what a
         * reader of a fresh buffer needs, and listed senones overwrite it) */
        for (i = 0; i < s->n_sen; i++) senscr[i] = (int16_t)S3A_SENSCR_DUMMY;
        for (i = 0, n = 0; rc == 1 && i < na; ++i) {
            int16_t v;
            n += lst[i];
            if (n >= s->n_sen) { s3a_set_error("senone score dump: active list runs past the last senone"); rc = S3A_EIO; break; }
            if (fread(&v, 2, 1, s->fp) != 1) { rc = 0; break; }
            senscr[n] = senlog_swap16(s, v);
        }
        if (!active) free(lst);
        return rc;
    }
}

void
s3a_senlog_close(s3a_senlog_t *s)
{
    if (!s) return;
    if (s->fp) fclose(s->fp);
    free(s);
}

/* ------------------------------------------------------------------ */
/* lattice files from the device's lattice (dag_write, dag_write_htk: sphinx3 libsearch/dag.c:731-897) */
/* ------------------------------------------------------------------ */
typedef struct { char *buf; int64_t cap, len; } lat_out_t;
static void
lat_put(lat_out_t *o, const char *fmt, ...)
{
    va_list ap;
    int n;
    char tmp[1];
    const int64_t room = o->cap > o->len ? o->cap - o->len : 0;
    va_start(ap, fmt);
    n = vsnprintf(room > 0 ? o->buf + o->len : tmp, room > 0 ? (size_t)room : 0, fmt, ap);
    va_end(ap);
    if (n > 0) o->len += n;
}

/* a lattice's indices before they are used: initial / final node, every link's ends, word ids not negative (the caller's wordstr /
 * basewid arrays are as long as its dictionary: that bound is the caller's) */
static int32_t
lat_check(const s3a_lat_info_t *info, const s3a_lat_node_t *nodes, const s3a_lat_link_t *links, const char *who)
{
    int32_t i;
    if (info->n_nodes < 0 || info->n_links < 0 || (info->n_nodes > 0 && (info->initial < 0 || info->initial >= info->n_nodes || info->final < 0 || info->final >= info->n_nodes))) {
        s3a_set_error("%s: initial / final node outside the lattice's %d nodes", who, info->n_nodes);
        return S3A_EINVAL;
    }
    for (i = 0; i < info->n_nodes; i++)
        if (nodes[i].wid < 0) { s3a_set_error("%s: node %d has word id %d", who, i, nodes[i].wid); return S3A_EINVAL; }
    for (i = 0; i < info->n_links; i++)
        if (links[i].from < 0 || links[i].from >= info->n_nodes || links[i].to < 0 || links[i].to >= info->n_nodes) {
            s3a_set_error("%s: link %d joins nodes %d -> %d of %d", who, i, links[i].from, links[i].to, info->n_nodes);
            return S3A_EINVAL;
        }
    return S3A_OK;
}

int64_t
s3a_lattice_format_s3(const char *header, const s3a_lat_info_t *info, const s3a_lat_node_t *nodes,
                      const s3a_lat_link_t *links, const char *const *wordstr, char *buf, int64_t cap)
{
    lat_out_t o;
    int32_t i;
    if (!info || !nodes || (!links && info->n_links > 0) || !wordstr || cap < 0 || (cap > 0 && !buf)) { s3a_set_error("s3a_lattice_format_s3: bad arguments"); return S3A_EINVAL; }
    if (lat_check(info, nodes, links, "s3a_lattice_format_s3") != S3A_OK) return S3A_EINVAL;
    o.buf = buf; o.cap = cap; o.len = 0;
    if (header) lat_put(&o, "%s", header);
    lat_put(&o, "Frames %d\n#\n", info->n_frames);
    lat_put(&o, "Nodes %d (NODEID WORD STARTFRAME FIRST-ENDFRAME LAST-ENDFRAME)\n", info->n_nodes);
    for (i = 0; i < info->n_nodes; i++)
        lat_put(&o, "%d %s %d %d %d\n", i, wordstr[nodes[i].wid], nodes[i].sf, nodes[i].fef, nodes[i].lef);
    lat_put(&o, "#\nInitial %d\nFinal %d\n", info->initial, info->final);
    lat_put(&o, "BestSegAscr 0 (NODEID ENDFRAME ASCORE)\n#\n");
    lat_put(&o, "Edges (FROM-NODEID TO-NODEID ASCORE)\n");
    for (i = 0; i < info->n_links; i++) lat_put(&o, "%d %d %d\n", links[i].from, links[i].to, links[i].ascr);
    lat_put(&o, "End\n");
    return o.len;
}

int64_t
s3a_lattice_format_htk(const char *header, const s3a_htk_opts_t *h, const s3a_lat_info_t *info, const s3a_lat_node_t *nodes,
                       const s3a_lat_link_t *links, const char *const *wordstr, char *buf, int64_t cap)
{
    lat_out_t o;
    int32_t i, j, *first = NULL, *order = NULL;
    float fps;
    if (!h || !info || !nodes || (!links && info->n_links > 0) || !wordstr || !h->basewid || !h->n_alt || cap < 0 || (cap > 0 && !buf)) {
        s3a_set_error("s3a_lattice_format_htk: bad arguments");
        return S3A_EINVAL;
    }
    if (lat_check(info, nodes, links, "s3a_lattice_format_htk") != S3A_OK) return S3A_EINVAL;
    o.buf = buf; o.cap = cap; o.len = 0;
    lat_put(&o, "# Lattice generated by Sphinx-III\n");
    if (header) lat_put(&o, "%s", header);
    lat_put(&o, "VERSION=1.0\nUTTERANCE=%s\n", h->uttid ? h->uttid : "");
    if (h->have_lm) {
        if (h->lmname) lat_put(&o, "lmname=%s\n", h->lmname);
        lat_put(&o, "lmscale=%f\n", h->opt_lw);
        lat_put(&o, "wdpenalty=%f\n", h->opt_wip);
    }
    lat_put(&o, "N=%d\tL=%d\n", info->n_nodes + 1, info->n_links + 1);
    fps = h->frate > 0 ? (float)h->frate : 100.0f;
    lat_put(&o, "I=%-5d t=%-10.2f\n", 0, (float)info->n_frames / fps);
    for (i = 0; i < info->n_nodes; i++) lat_put(&o, "I=%-5d t=%-10.2f\n", i + 1, (float)nodes[i].sf / fps);
    lat_put(&o, "J=%-10d S=%-5d E=%-5d W=%-20s a=%-10.2f v=%-5d l=%-10.2f\n", 0, info->final + 1, 0, wordstr[nodes[info->final].wid], 0.0, 1, 0.0);
    /* a node's predlist = its incoming links by source id ascending: a stable counting sort of the links by destination */
    first = (int32_t *)calloc((size_t)info->n_nodes + 2, sizeof(int32_t));
    order = (int32_t *)malloc(((size_t)info->n_links + 1) * sizeof(int32_t));
    if (!first || !order) { free(first); free(order); s3a_set_error("s3a_lattice_format_htk: out of memory"); return S3A_ENOMEM; }
    for (i = 0; i < info->n_links; i++) first[links[i].to + 1]++;
    for (i = 0; i < info->n_nodes; i++) first[i + 1] += first[i];
    for (i = 0; i < info->n_links; i++) order[first[links[i].to]++] = i;
    for (j = 0; j < info->n_links; j++) {
        const s3a_lat_link_t *l = &links[order[j]];
        const int32_t b = h->basewid[nodes[l->from].wid];
        int32_t ls = l->lscr;
        if (h->have_lm) { ls -= h->lm_wip; ls = (int32_t)((float)ls / h->lm_lw); }      /* lm_rawscore, lm.c:2172-2178 */
        lat_put(&o, "J=%-10d S=%-5d E=%-5d W=%-20s a=%-10.2f v=%-5d l=%-10.2f\n", j + 1, l->from + 1, l->to + 1, wordstr[b],
                (double)(l->ascr << h->log_shift) * h->log_of_base, h->n_alt[b], (double)(ls << h->log_shift) * h->log_of_base);
    }
    free(first); free(order);
    return o.len;
}
