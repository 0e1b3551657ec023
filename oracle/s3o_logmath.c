/*
 * oracle/s3o_logmath.c -- CPU ORACLE (test infrastructure only; see s3o.h).
 *
 * Integer log-domain arithmetic, restating
 *   sphinxbase/src/libsphinxbase/util/logmath.c:61-161 (logmath_init)
 *   sphinxbase/src/libsphinxbase/util/logmath.c:391-436 (logmath_add)
 *   sphinxbase/src/libsphinxbase/util/logmath.c:445-483 (log/exp/conversions)
 *   sphinx3/src/libs3decoder/libcommon/logs3.c:111-119  (logs3)
 */
#include <math.h>
#include <stdlib.h>
#include "s3o.h"

s3o_logmath_t *
s3o_logmath_init(double base, int shift, int use_table)
{
    s3o_logmath_t *lm;
    uint32_t maxyx, i;
    double byx;

    if (base <= 1.0)
        return NULL;
    lm = (s3o_logmath_t *)calloc(1, sizeof(*lm));
    lm->base = base;
    lm->log_of_base = log(base);
    lm->log10_of_base = log10(base);
    lm->inv_log_of_base = 1.0 / lm->log_of_base;
    lm->inv_log10_of_base = 1.0 / lm->log10_of_base;
    lm->shift = shift;
    lm->zero = S3O_MAX_NEG_INT32 >> (shift + 2);
    if (!use_table)
        return lm;

    /* entry width follows the largest value the table can hold */
    maxyx = (uint32_t)(log(2.0) / log(base) + 0.5) >> shift;
    lm->width = (maxyx < 256) ? 1 : (maxyx < 65536) ? 2 : 4;

    /* pass 1: how many entries until log_b(1 + b^-d) rounds to zero */
    byx = 1.0;
    for (i = 0;; ++i) {
        double lobyx = log(1.0 + byx) * lm->inv_log_of_base;
        int32_t k = (int32_t)(lobyx + 0.5 * (1 << shift)) >> shift;
        if (k <= 0)
            break;
        byx /= base;
    }
    i >>= shift;
    if (i < 255)
        i = 255;
    lm->table_size = i + 1;
    lm->table = (uint32_t *)calloc(lm->table_size, sizeof(uint32_t));

    /* pass 2: fill; with a shift only the first (highest) value per slot is kept */
    byx = 1.0;
    for (i = 0;; ++i) {
        double lobyx = log(1.0 + byx) * lm->inv_log_of_base;
        int32_t k = (int32_t)(lobyx + 0.5 * (1 << shift)) >> shift;
        if (lm->table[i >> shift] == 0) {
            uint32_t v = (uint32_t)k;
            if (lm->width == 1) v = (uint8_t)k;
            else if (lm->width == 2) v = (uint16_t)k;
            lm->table[i >> shift] = v;
        }
        if (k <= 0)
            break;
        byx /= base;
    }
    return lm;
}

void
s3o_logmath_free(s3o_logmath_t *lm)
{
    if (!lm) return;
    free(lm->table);
    free(lm);
}

int
s3o_logmath_log(const s3o_logmath_t *lm, double p)
{
    if (p <= 0)
        return lm->zero;
    return (int)(log(p) * lm->inv_log_of_base) >> lm->shift;
}

double
s3o_logmath_exp(const s3o_logmath_t *lm, int logb_p)
{
    return pow(lm->base, (double)(logb_p << lm->shift));
}

int
s3o_logmath_add(const s3o_logmath_t *lm, int x, int y)
{
    int d, r;

    if (x <= lm->zero)
        return y;
    if (y <= lm->zero)
        return x;
    if (lm->table == NULL)      /* logmath_add_exact, logmath.c:438-443 */
        return s3o_logmath_log(lm, s3o_logmath_exp(lm, x) + s3o_logmath_exp(lm, y));
    if (x > y) { d = x - y; r = x; }
    else       { d = y - x; r = y; }
    if (d < 0)
        return r;               /* overflow: fail gracefully */
    if ((size_t)d >= lm->table_size)
        return r;
    return r + (int)lm->table[d];
}

double
s3o_logmath_log_to_ln(const s3o_logmath_t *lm, int logb_p)
{
    return (double)(logb_p << lm->shift) * lm->log_of_base;
}

int
s3o_logmath_ln_to_log(const s3o_logmath_t *lm, double log_p)
{
    return (int)(log_p * lm->inv_log_of_base) >> lm->shift;
}

int
s3o_logmath_log10_to_log(const s3o_logmath_t *lm, double log_p)
{
    return (int)(log_p * lm->inv_log10_of_base) >> lm->shift;
}

int32_t
s3o_logs3(const s3o_logmath_t *lm, double p)
{
    if (p <= 0.0)
        return S3O_LOGPROB_ZERO;
    return s3o_logmath_log(lm, p);
}
