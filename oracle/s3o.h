/*
 * oracle/s3o.h -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's algorithm for the
 * GMM-senone-scoring + HMM-Viterbi hot path.  Nothing
 * in the product (cmusphinx_amd/, include/) may include, link or call this;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do,
 * and only as the checker.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose behaviour it restates.  The oracle works on plain arrays (the tests
 * parse the S3 model files with numpy), so it shares no loader code with the
 * product's host C.
 *
 * Parity status: PINNED.  tests/test_oracle_*.py check it (a) against the
 * reference's own known-answer tests (sphinxbase test_logmath, sphinx3
 * test_logs3 = 79150, test_hmm/_testhmm_tidigits.res) and (b) against
 * outputs of the unmodified reference built in this container
 * (oracle/_ref/ref_dump -> tests/golden/, generator: tests/golden/make_golden.py).
 */
#ifndef S3O_H
#define S3O_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* sphinx3/include/s3types.h:192 */
#define S3O_LOGPROB_ZERO ((int32_t)0xc8000000)
/* sphinxbase/include/sphinxbase/prim_type.h:153 */
#define S3O_MAX_NEG_INT32 ((int32_t)0x80000000)
/* sphinx3/include/cont_mgau.h:121-123 */
#define S3O_NO_BSTIDX (-1)
#define S3O_NOT_UPDATED (-100)

/* ------------------------------------------------------------------ */
/* integer log-domain arithmetic                                       */
/* sphinxbase/src/libsphinxbase/util/logmath.c:61-161, 391-471         */
/* ------------------------------------------------------------------ */
typedef struct s3o_logmath_s {
    double base, log_of_base, log10_of_base, inv_log_of_base, inv_log10_of_base;
    int shift;
    int32_t zero;
    int width;              /* 1, 2 or 4 bytes per entry in the reference */
    uint32_t table_size;    /* 0 when built without a table */
    uint32_t *table;        /* widened to u32; values already truncated to `width` */
} s3o_logmath_t;

s3o_logmath_t *s3o_logmath_init(double base, int shift, int use_table);
void s3o_logmath_free(s3o_logmath_t *lm);
int s3o_logmath_add(const s3o_logmath_t *lm, int x, int y);
int s3o_logmath_log(const s3o_logmath_t *lm, double p);
double s3o_logmath_exp(const s3o_logmath_t *lm, int logb_p);
double s3o_logmath_log_to_ln(const s3o_logmath_t *lm, int logb_p);
int s3o_logmath_ln_to_log(const s3o_logmath_t *lm, double log_p);
int s3o_logmath_log10_to_log(const s3o_logmath_t *lm, double log_p);
/* sphinx3/src/libs3decoder/libcommon/logs3.c:111-119 */
int32_t s3o_logs3(const s3o_logmath_t *lm, double p);

/* ------------------------------------------------------------------ */
/* continuous-density mixture Gaussians                                */
/* sphinx3/include/cont_mgau.h:170-226, libam/cont_mgau.c              */
/* ------------------------------------------------------------------ */
typedef struct s3o_mgau_s {
    int32_t n_mgau, max_comp, veclen;
    int32_t *n_comp;        /* [n_mgau]  after mgau_uninit_compact */
    float *mean;            /* [n_mgau][max_comp][veclen] (compacted in place) */
    float *var;             /* same shape; after precomp holds 1/(2 sigma^2) */
    float *lrd;             /* [n_mgau][max_comp] */
    int32_t *mixw;          /* [n_mgau][max_comp] logs3 */
    int32_t *bstidx, *bstscr, *updatetime;  /* [n_mgau] mutable per-senone state */
    double distfloor;
    const s3o_logmath_t *lm;
    /* per-frame counters, cont_mgau.h:214-218 */
    int32_t frm_sen_eval, frm_gau_eval, frm_ci_sen_eval, frm_ci_gau_eval;
} s3o_mgau_t;

/* mgau_init (cont_mgau.c:901-956) minus the file parsing: takes the raw
 * float arrays of the means / variances / mixture_weights files. */
s3o_mgau_t *s3o_mgau_init(const float *mean, const float *var, const float *mixw,
                          int32_t n_mgau, int32_t n_density, int32_t veclen,
                          double varfloor, double mixwfloor, int precomp,
                          const s3o_logmath_t *lm);
void s3o_mgau_free(s3o_mgau_t *g);
/* mgau_eval (cont_mgau.c:1174-1205) */
int32_t s3o_mgau_eval(s3o_mgau_t *g, int32_t m, const int32_t *active,
                      const float *x, int32_t fr, int32_t update_best_id);
/* reset bstidx/bstscr/updatetime of every senone, as srch_TST_begin does
 * (srch_time_switch_tree.c:485-490) */
void s3o_mgau_reset_state(s3o_mgau_t *g);

/* fast_gmm_t subset that matters without GS/SVQ (fast_algo_struct.h) */
typedef struct s3o_fastgmm_s {
    int32_t ds_ratio;        /* -ds */
    int32_t cond_ds;         /* -cond_ds (needs a Gaussian selector: must be 0) */
    int32_t ci_pbeam;        /* logs3(-ci_pbeam) */
    int32_t max_cd;          /* -maxcdsenpf */
    float tighten_factor;    /* -tighten_factor */
    int32_t dyn_ci_pbeam;    /* out: last beam used */
    int32_t skip_count;
} s3o_fastgmm_t;

/* approx_cont_mgau_ci_eval (approx_cont_mgau.c:367-428) */
void s3o_approx_cont_mgau_ci_eval(s3o_mgau_t *g, const int16_t *cd2cisen, int32_t n_sen,
                                  const float *feat, int32_t *ci_senscr,
                                  int32_t *best_score, int32_t fr);
/* approx_cont_mgau_frame_eval (approx_cont_mgau.c:434-616) */
int32_t s3o_approx_cont_mgau_frame_eval(s3o_mgau_t *g, s3o_fastgmm_t *fg,
                                        const int16_t *cd2cisen, int32_t n_ci_sen,
                                        uint8_t *sen_active, uint8_t *rec_sen_active,
                                        int32_t *senscr, const float *feat, int32_t frame,
                                        const int32_t *cache_ci_senscr);

/* dict2pid_comsenscr (libsearch/dict2pid.c:1029-1048); comstate is the
 * ragged list flattened: comstate_off[i]..comstate_off[i+1] */
void s3o_dict2pid_comsenscr(int32_t n_comstate, const int32_t *comstate_off,
                            const int16_t *comstate, const int32_t *comwt,
                            const int32_t *senscr, int32_t *comsenscr);

/* ------------------------------------------------------------------ */
/* transition matrices + HMM Viterbi                                   */
/* libam/tmat.c:155-270, include/hmm.h:156-197, libam/hmm.c            */
/* ------------------------------------------------------------------ */
#define S3O_MAX_HMM_NSTATE 5
typedef struct s3o_hmm_s {
    int32_t score[S3O_MAX_HMM_NSTATE];
    int64_t history[S3O_MAX_HMM_NSTATE];    /* union {long id; void *ptr} */
    int32_t out_score;
    int64_t out_history;
    int32_t ssid;                           /* non-mpx */
    int32_t mpx_ssid[S3O_MAX_HMM_NSTATE];   /* mpx */
    int32_t bestscore;
    int32_t tmatid;
    int32_t frame;
    uint8_t mpx;
} s3o_hmm_t;

typedef struct s3o_hmm_ctx_s {
    int32_t n_emit_state;
    const int32_t *tp;          /* [n_tmat][n_emit][n_emit+1] logs3 */
    const int32_t *senscore;    /* [n_sen] */
    const int16_t *sseq;        /* [n_sseq][n_emit] */
} s3o_hmm_ctx_t;

/* tmat_init's float->logs3 conversion (tmat.c:232-247): in place on a copy */
void s3o_tmat_logs3(const float *tp_in, int32_t n_tmat, int32_t n_src,
                    double tpfloor, const s3o_logmath_t *lm, int32_t *tp_out);
void s3o_hmm_clear(const s3o_hmm_ctx_t *ctx, s3o_hmm_t *h);                 /* hmm.c:225-241 */
void s3o_hmm_init(const s3o_hmm_ctx_t *ctx, s3o_hmm_t *h, int mpx,
                  int32_t ssid, int32_t tmatid);                             /* hmm.c:130-147 */
void s3o_hmm_enter(s3o_hmm_t *h, int32_t score, int64_t histid, int32_t frame); /* hmm.c:244-250 */
void s3o_hmm_normalize(const s3o_hmm_ctx_t *ctx, s3o_hmm_t *h, int32_t bestscr); /* hmm.c:260-271 */
int32_t s3o_hmm_vit_eval(const s3o_hmm_ctx_t *ctx, s3o_hmm_t *h);            /* hmm.c:855-873 */

/* ------------------------------------------------------------------ */
/* lexical-tree search, per-frame operations on a flattened lextree     */
/* libsearch/lextree.c:910-961, 1093-1663 (see s3o_lextree.c)           */
/* ------------------------------------------------------------------ */
typedef struct s3o_lextree_s {
    int32_t n_node;
    const int32_t *ssid, *tmatid, *wid, *prob;  /* wid < 0: not a leaf */
    const uint8_t *composite;
    const int32_t *child_off, *child;           /* CSR child lists, glist order */
    int32_t n_lc;                               /* 0: enter through root[] (filler trees) */
    const int16_t *lc;
    const int32_t *lcroot_off, *lcroot;         /* per left context: root node list, glist order */
    int32_t n_root;
    const int32_t *root;
    s3o_hmm_ctx_t ctx, comctx;
    s3o_hmm_t *hmm;                             /* [n_node] */
    int32_t *active, *next_active;
    int32_t n_active, n_next_active;
    int32_t best, wbest;
} s3o_lextree_t;

s3o_lextree_t *s3o_lextree_init(int32_t n_node, const int32_t *ssid, const int32_t *tmatid,
                                const uint8_t *composite, const int32_t *wid, const int32_t *prob,
                                const int32_t *child_off, const int32_t *child,
                                int32_t n_lc, const int16_t *lc, const int32_t *lcroot_off,
                                const int32_t *lcroot, int32_t n_root, const int32_t *root,
                                int32_t n_emit, const int32_t *tp, const int16_t *sseq,
                                const int16_t *comsseq);
void s3o_lextree_free(s3o_lextree_t *lt);
void s3o_lextree_enter(s3o_lextree_t *lt, int32_t lc, int32_t cf, int32_t inscore, int32_t inhist,
                       int32_t thresh);
void s3o_lextree_active_swap(s3o_lextree_t *lt);
int32_t s3o_lextree_hmm_eval(s3o_lextree_t *lt, const int32_t *senscr, const int32_t *comsen,
                             int32_t frm);
void s3o_lextree_hmm_histbin(s3o_lextree_t *lt, int32_t bestscr, int32_t *bin, int32_t nbin,
                             int32_t bw);
void s3o_lextree_hmm_propagate_non_leaves(s3o_lextree_t *lt, int32_t cf, int32_t th, int32_t pth,
                                          int32_t wth);
/* returns #word exits (or -1 on out.history == -1); fills at most max_out */
int32_t s3o_lextree_hmm_propagate_leaves(const s3o_lextree_t *lt, int32_t wth, int32_t *out_wid,
                                         int32_t *out_score, int32_t *out_hist, int32_t max_out);
void s3o_lextree_ssid_active(const s3o_lextree_t *lt, uint8_t *ssid, uint8_t *comssid);
void s3o_sseq2sen_active(const int16_t *sseq, int32_t n_sseq, int32_t n_emit, const uint8_t *ssid,
                         uint8_t *sen);
void s3o_comsseq2sen_active(const int16_t *comsseq, int32_t n_comsseq, int32_t n_emit,
                            const int32_t *comstate_off, const int16_t *comstate,
                            const uint8_t *comssid, uint8_t *sen);
void s3o_lextree_utt_end(s3o_lextree_t *lt);

/* ------------------------------------------------------------------ */
/* the word level of mode 4 (s3o_wordlevel.c): trigram look-up,         */
/* Viterbi history, word transitions                                    */
/* liblm/lm.c:983-1833, libsearch/vithist.c:300-860,1066-1100,          */
/* libsearch/srch_time_switch_tree.c:1086-1179, sphinxbase util/heap.c  */
/* ------------------------------------------------------------------ */
/* lm_t flattened: probabilities / back-off weights already dereferenced (bgprob[probid] ...)
 * and language-weighted as lm_set_param left them; bigrams of unigram w = [ug_firstbg[w],
 * ug_firstbg[w+1]), trigrams of bigram b = [bg_firsttg[b], bg_firsttg[b+1]) (absolute). */
typedef struct s3o_lm3g_s {
    int32_t n_ug, n_bg, n_tg;
    const int32_t *ug_prob, *ug_bowt, *ug_firstbg;              /* [n_ug], [n_ug], [n_ug + 1] */
    const int32_t *bg_wid, *bg_prob, *bg_bowt, *bg_firsttg;      /* [n_bg] x 3, [n_bg + 1] */
    const int32_t *tg_wid, *tg_prob;                            /* [n_tg] */
    const int32_t *inclass;                                     /* per dictionary word, or NULL */
} s3o_lm3g_t;
int32_t s3o_lm_bg_score(const s3o_lm3g_t *lm, int32_t lw1, int32_t lw2, int32_t wid);
int32_t s3o_lm_tg_score(const s3o_lm3g_t *lm, int32_t lw1, int32_t lw2, int32_t lw3, int32_t wid);

/* what the word level reads of dict_t / fillpen_t / mdef_t, per dictionary word id */
typedef struct s3o_wdict_s {
    int32_t n_word, n_ci;
    const int32_t *lwid;        /* lm->dict2lmwid[w] (negative: none) */
    const uint8_t *is_filler;   /* dict_filler_word */
    const int32_t *fillpen;     /* fillpen(kbcore_fillpen, w) for filler words */
    const int32_t *last_ci;     /* dict_last_phone, filler phones mapped to mdef_silphone */
    int32_t startwid, finishwid, silwid, start_lwid, finish_lwid;
} s3o_wdict_t;

typedef struct s3o_vithist_s {
    int32_t cap, max_frames, n_entry, n_frm, wbeam, bghist, overflow;
    int32_t *score, *pred, *lw0, *lw1, *wid, *sf, *ef, *ascr, *lscr, *type;
    uint8_t *valid;
    int32_t *frame_start, *bestscore, *bestvh;
} s3o_vithist_t;

s3o_vithist_t *s3o_vithist_init(int32_t cap, int32_t max_frames, int32_t wbeam, int32_t bghist);
void s3o_vithist_free(s3o_vithist_t *vh);
void s3o_vithist_utt_begin(s3o_vithist_t *vh, int32_t startwid, int32_t start_lwid);
int32_t s3o_vithist_rescore(s3o_vithist_t *vh, const s3o_lm3g_t *lm, const s3o_wdict_t *d, int32_t wid,
                            int32_t ef, int32_t score, int32_t pred, int32_t type);
void s3o_vithist_prune(s3o_vithist_t *vh, const s3o_wdict_t *d, int32_t frm, int32_t maxwpf,
                       int32_t maxhist, int32_t beam, int32_t *order_out);
void s3o_vithist_frame_windup(s3o_vithist_t *vh, int32_t frm);
int32_t s3o_word_trans(const s3o_vithist_t *vh, const s3o_wdict_t *d, int32_t cf, int32_t wordend_beam,
                       int32_t *lc, int32_t *scr, int32_t *hist, int32_t *fill_scr, int32_t *fill_hist);
int32_t s3o_vithist_utt_end(s3o_vithist_t *vh, const s3o_lm3g_t *lm, const s3o_wdict_t *d);
int32_t s3o_vithist_backtrace(const s3o_vithist_t *vh, int32_t id, int32_t *ids, int32_t max_ids);

/* ------------------------------------------------------------------ */
/* the second pass (s3o_dag.c): word lattice from the history table,    */
/* filler bypass, best path under the trigram                           */
/* libsearch/vithist.c:1100-1311, dag.c:186-300, 397-484, 590-671,      */
/* 893-965, 1037-1075, srch_time_switch_tree.c:1391-1440                */
/* ------------------------------------------------------------------ */
typedef struct s3o_dagcfg_s {
    int32_t n_word;
    const int32_t *basewid;     /* dict_basewid */
    const uint8_t *is_filler;   /* dict_filler_word (0 for <s> / </s>) */
    const int32_t *lwid;        /* lm->dict2lmwid[w] */
    const int32_t *fillpen;     /* fillpen(fpen, w) of the filler words */
    int32_t startwid, finishwid, start_lwid, finish_lwid;
    int32_t wip;                /* logs3(fpen->wip) */
    double lwf;                 /* -bestpathlw / -lw (1.0 when -bestpathlw is 0) */
    int32_t min_endfr, maxedge, maxlmop, maxlpf;
} s3o_dagcfg_t;
/* history table incl. the final </s> entry (endid); hyp_* = the first pass's words (kept whatever their duration);
 * out = [5][max_out]: wid sf ef ascr lscr; returns the number of words, -1: no path / limits, -2: out too small;
 * stats[5] (optional): nodes kept, links built, links after the bypass, bypass links, LM operations */
int32_t s3o_dag_bestpath(const s3o_dagcfg_t *c, const s3o_lm3g_t *lm, int32_t n_entry, const int32_t *wid,
                         const int32_t *sf, const int32_t *ef, const int32_t *ascr, const int32_t *lscr,
                         const int32_t *score, const uint8_t *valid, int32_t n_frm, int32_t endid, int32_t n_hyp,
                         const int32_t *hyp_wid, const int32_t *hyp_sf, int32_t *out, int32_t max_out, int32_t *stats);

/* ------------------------------------------------------------------ */
/* multi-stream senone scorer (-senmgau .s3cont. / .semi.)             */
/* sphinx3 libam/ms_gauden.c, ms_senone.c, ms_mgau.c                   */
/* ------------------------------------------------------------------ */
typedef struct s3o_ms_s {
    int32_t n_mgau, n_feat, n_density, n_sen, topn, veclen;
    int32_t *featlen, *featoff;     /* [n_feat], [n_feat + 1] */
    float *mean, *var, *det;        /* file order [m][f][d][featlen f]; var = 1/(2 sigma^2); det [m][f][d] */
    int32_t *pdf;                   /* [n_sen][n_feat][n_density]  -logs3(mixture weight) */
    int32_t *mgau;                  /* [n_sen] codebook of each senone */
    double min_density;
    int32_t *dist_id, *dist;        /* [n_mgau][n_feat][topn] top-N of the last frame */
    uint8_t *mgau_active;
    const s3o_logmath_t *lm;
} s3o_ms_t;

/* ms_mgau_init (ms_mgau.c:149-227) minus the file parsing; sen2mgau == NULL is ".s3cont." */
s3o_ms_t *s3o_ms_init(const float *mean, const float *var, const float *mixw, int32_t n_mgau,
                      int32_t n_feat, int32_t n_density, const int32_t *featlen, int32_t n_sen,
                      const int32_t *sen2mgau, double varfloor, double mixwfloor, int32_t topn,
                      const s3o_logmath_t *lm);
void s3o_ms_free(s3o_ms_t *ms);
/* ms_cont_mgau_frame_eval (ms_mgau.c:242-329); feat = the streams concatenated */
int32_t s3o_ms_cont_mgau_frame_eval(s3o_ms_t *ms, const uint8_t *sen_active, int32_t *senscr,
                                    const float *feat);

/* ------------------------------------------------------------------ */
/* pocketsphinx's continuous scorer (ps_mgaufuncs_t "ms"), see s3o_psms.c */
/* ------------------------------------------------------------------ */
typedef struct s3o_psms_s {
    int32_t n_mgau, n_feat, n_density, n_sen, topn, veclen, aw;
    int32_t *featlen, *featoff;
    float *mean, *var, *det;        /* var = log-domain precision (float32 of an int), det [m][f][d] */
    uint8_t *pdf;                   /* [n_sen][n_feat][n_density] 8-bit -log weights */
    int32_t *mgau;
    float *dist; int32_t *dist_id;  /* top-N of the last frame [n_mgau][n_feat][topn] */
    uint8_t *mgau_active;
    s3o_logmath_t *lm, *lm8;
} s3o_psms_t;
s3o_psms_t *s3o_psms_init(const float *mean, const float *var, const float *mixw, int32_t n_mgau,
                          int32_t n_feat, int32_t n_density, const int32_t *featlen, int32_t n_sen,
                          const int32_t *sen2mgau, double varfloor, double mixwfloor, int32_t topn,
                          int32_t aw, double logbase);
void s3o_psms_free(s3o_psms_t *ms);
void s3o_psms_frame_eval(s3o_psms_t *ms, int16_t *senscr, const uint8_t *senone_active,
                         int32_t n_senone_active, const float *feat, int32_t compallsen);

/* feat_compute_utt for the stream type "1s_c_d_dd" (sphinxbase feat.c:1111-1123, :726-769; cmn.c; agc.c):
 * cep [n][cepsize] -> feat [n][3 * cepsize] */
void s3o_feat_1s_c_d_dd(const float *cep, int32_t n_frames, int32_t cepsize, int32_t cmn_current,
                        int32_t varnorm, int32_t agc_max, float *feat);

/* ---- the MFCC front end (s3o_fe.c; sphinxbase fe_interface.c / fe_sigproc.c) ---- */
typedef struct {
    float samprate;         /* -samprate */
    int32_t frate;          /* -frate */
    float wlen;             /* -wlen */
    float alpha;            /* -alpha */
    int32_t ncep, nfft, nfilt;
    float lowerf, upperf;
    int32_t transform;      /* 0 legacy, 1 dct, 2 htk */
    int32_t lifter, remove_dc, round_filters, unit_area, doublebw;
    int32_t logspec;        /* 0 cepstra, 1 -logspec, 2 -smoothspec */
} s3o_fe_params_t;

typedef struct {
    s3o_fe_params_t p;
    int32_t fft_order, frame_shift, frame_size, feature_dimension, n_coeffs;
    double *hamming, *ccc, *sss;
    int16_t *spec_start, *filt_start, *filt_width;
    float *filt_coeffs, *mel_cosine, *lifter;
    float sqrt_inv_n, sqrt_inv_2n;
} s3o_fe_t;

s3o_fe_t *s3o_fe_init(const s3o_fe_params_t *p);
void s3o_fe_free(s3o_fe_t *fe);
int32_t s3o_fe_n_frames(const s3o_fe_t *fe, int64_t nsamps);
int32_t s3o_fe_process_utt(const s3o_fe_t *fe, const int16_t *spch, int64_t nsamps, float *cep);

#ifdef __cplusplus
}
#endif
#endif
