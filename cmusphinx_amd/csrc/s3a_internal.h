/*
 * s3a_internal.h -- structures shared between the host C side
 * (s3a_host.c: file formats, logmath, model precomputation) and the HIP side
 * (s3a_device.hip: device layouts, kernels, launchers).
 */
#ifndef S3A_INTERNAL_H
#define S3A_INTERNAL_H

#include "cmusphinx_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

void s3a_set_error(const char *fmt, ...);
const s3a_variants_t *s3a_variants(void);      /* s3a_host.c: what s3a_set_variants stored */

struct s3a_logmath_s {
    double base, log_of_base, log10_of_base, inv_log_of_base, inv_log10_of_base;
    int32_t shift;
    int32_t zero;
    int32_t width;          /* bytes per table entry in the reference: 1, 2 or 4 */
    uint32_t table_size;    /* 0: no table (logmath_add_exact) */
    uint32_t *table;        /* widened */
};

/* device side of a model; defined in s3a_device.hip */
struct s3a_mgau_dev_s;

struct s3a_mgau_model_s {
    int32_t n_mgau, max_comp, veclen;
    int32_t *n_comp;        /* [n_mgau] */
    float *mean;            /* [n_mgau][max_comp][veclen] host AoS (compacted) */
    float *prec;            /* 1/(2 sigma^2) after precomp */
    float *lrd;             /* [n_mgau][max_comp] */
    int32_t *mixw;          /* [n_mgau][max_comp] */
    double distfloor;
    double f;               /* 1.0 / log(base): cont_mgau.c:1042 */
    s3a_logmath_t *lm;      /* borrowed */
    int32_t precision;      /* S3A_GMM_EXACT / S3A_GMM_FAST */
    struct s3a_mgau_dev_s *dev;
};

/* multi-stream scorer (-senmgau .s3cont. / .semi.): gauden_t + senone_t of ms_mgau_model_t */
struct s3a_ms_dev_s;
struct s3a_ms_mgau_s {
    int32_t n_mgau, n_feat, n_density, n_sen, topn, veclen;
    int32_t one_to_one;     /* ".s3cont.": senone s uses codebook s */
    int32_t *featlen, *featoff;         /* [n_feat], [n_feat + 1] */
    float *mean, *prec, *det;           /* file order [m][f][d][featlen f]; det [m][f][d] */
    int32_t *pdf;                       /* [n_sen][n_feat][n_density]  -logs3(weight) */
    int32_t *mgau;                      /* [n_sen] */
    double min_density;
    s3a_logmath_t *lm;                  /* borrowed */
    struct s3a_ms_dev_s *dev;
};

/* pocketsphinx's continuous scorer (ps_mgaufuncs_t "ms"): float32, log-domain precisions, 8-bit weights */
struct s3a_ps_dev_s;
struct s3a_ps_mgau_s {
    int32_t n_mgau, n_feat, n_density, n_sen, topn, veclen, aw;
    int32_t one_to_one;
    int32_t *featlen, *featoff;
    float *mean, *prec, *det;           /* file order; prec = (float) logmath_ln_to_log(1 / (2 var)) */
    int32_t *pdf;                       /* [n_sen][n_feat][n_density]: the 8-bit weights, widened */
    int32_t *mgau;
    s3a_logmath_t *lm, *lm8;            /* owned: unshifted (no table) and shifted by SENSCR_SHIFT (table) */
    uint8_t *flags;                     /* host scratch [n_sen] */
    struct s3a_ps_dev_s *dev;
};

struct s3a_tmat_s {
    int32_t n_tmat, n_state;
    int32_t *tp;            /* [n_tmat][n_state][n_state+1] logs3 */
};

/* ---- host side (s3a_host.c) ---- */
/* S3 binary envelope reader: returns malloc'd payload words (host byte order)
 * after verifying the checksum when the header announces one. */
int32_t s3a_bio_read(const char *path, const char *expect_version, uint32_t **words,
                     size_t *n_words);
/* host half of ms_mgau_init on raw arrays; leaves msg->dev NULL (s3a_host.c) */
struct s3a_ms_mgau_s *s3a_ms_host_init(const float *mean, const float *var, const float *mixw,
                                       int32_t n_mgau, int32_t n_feat, int32_t n_density,
                                       const int32_t *featlen, int32_t n_sen, const int32_t *sen2mgau,
                                       double varfloor, double mixwfloor, int32_t topn,
                                       s3a_logmath_t *lm);
void s3a_ms_host_free(struct s3a_ms_mgau_s *msg);
int32_t s3a_ps_dev_create(struct s3a_ps_mgau_s *ps);       /* s3a_psms.hip */
void s3a_ps_dev_destroy(struct s3a_ps_mgau_s *ps);
/* all senones of many frames, before normalisation (s3a_psms.hip; used by s3a_psfwd.hip) */
int32_t s3a_ps_score_slots_dev(struct s3a_ps_mgau_s *ps, const float *feat_dev, const int32_t *slot_row_dev,
                               int32_t n_slots, int16_t *raw_dev, void *stream);
int32_t s3a_ms_dev_create(struct s3a_ms_mgau_s *msg);      /* s3a_ms.hip */
void s3a_ms_dev_destroy(struct s3a_ms_mgau_s *msg);
/* host half of mgau_init on raw arrays; leaves g->dev NULL */
s3a_mgau_model_t *s3a_mgau_host_init(const float *mean, const float *var, const float *mixw,
                                     int32_t n_mgau, int32_t n_density, int32_t veclen,
                                     double varfloor, double mixwfloor, int32_t precomp,
                                     s3a_logmath_t *lm);
void s3a_mgau_host_free(s3a_mgau_model_t *g);

/* ---- device side (s3a_device.hip) ---- */
int32_t s3a_mgau_dev_create(s3a_mgau_model_t *g);
void s3a_mgau_dev_destroy(s3a_mgau_model_t *g);

#ifdef __cplusplus
}
#endif
#endif
