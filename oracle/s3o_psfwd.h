/*
 * oracle/s3o_psfwd.h -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see s3o.h).
 *
 * pocketsphinx's first pass (ngram_search_fwdtree.c) restated sequentially on flat arrays.
 * The descriptor has the same fields as the product's s3a_psfwd_desc_t (both are filled by
 * integration/pocketsphinx/psamd_export.h, which is written against the field NAMES), but is declared here
 * independently: the oracle includes nothing of the product.
 *
 * Parity status: PINNED on the unmodified pocketsphinx (oracle/_ref/libpsref.so) by tests/test_oracle_psfwd.py:
 * oracle/_ref/ref_ps_fwd runs the reference decoder with this restatement behind ps_searchfuncs_t
 * {start, step, finish} and the backpointer tables / hypotheses must equal the unmodified decoder's.
 */
#ifndef S3O_PSFWD_H
#define S3O_PSFWD_H
#include <stdint.h>

#define S3O_PS_WORST_SCORE ((int32_t)0xE0000000)    /* hmm.h:74 */
#define S3O_PS_TMAT_WORST (-255)                    /* hmm.h:80 */
#define S3O_PS_SENSCR_SHIFT 10                      /* hmm.h:63 */
#define S3O_PS_BAD_SSID 0xffff
#define S3O_PS_NO_BP (-1)

typedef struct s3o_psfwd_desc_s {
    int32_t n_ci, sil_ci, n_emit, n_sen, n_sseq, n_tmat;
    const uint16_t *sseq;
    const uint8_t *tp;
    int32_t n_words, start_wid, finish_wid, silence_wid;
    const int32_t *w_basewid, *w_lmwid;
    const int16_t *w_first_ci, *w_last_ci, *w_last2_ci;
    const uint8_t *w_flags;
    const int32_t *w_rc_off;
    const uint16_t *rc_ssid;
    const int32_t *w_rc_row;
    int32_t n_rc_rows;
    const int16_t *rc_cimap;
    const int16_t *w_rc_tmat;
    int32_t n_root, n_nonroot;
    const int16_t *root_ci, *root_ci2, *root_tmat;
    const uint16_t *root_ssid0;
    const uint16_t *root_lc_ssid;
    const int32_t *ch_child_off, *ch_child;
    const int32_t *ch_pen_off, *ch_pen_wid;
    const uint16_t *nr_ssid;
    const int16_t *nr_tmat, *nr_ci;
    int32_t n_1ph, n_1ph_lm;
    const int32_t *sp_wid;
    const uint16_t *sp_ssid0, *sp_lc_ssid;
    const int16_t *sp_tmat, *sp_ci;
    int32_t n_fill;
    const int32_t *fill_sp;
    int32_t lm_order, lm_n_ug, lm_n_bg, lm_n_tg, lm_zero;
    const int32_t *ug_prob, *ug_bowt, *ug_firstbg;
    const int32_t *bg_wid, *bg_prob, *bg_bowt, *bg_firsttg;
    const int32_t *tg_wid, *tg_prob;
    int32_t beam, pbeam, wbeam, lpbeam, lponlybeam, fillpen, silpen, nwpen, pip, maxwpf, maxhmmpf;
    int32_t pl_window, pl_beam, pl_pbeam, pl_pip;   /* phone loop look-ahead (phone_loop_search.c); 0: off */
    const uint16_t *ci_ssid;
    const int16_t *ci_tmat;
    const int32_t *w_lmcw;          /* [n_words] in-class weights of class-based LMs (ngram_class_prob; 1: not in the class); NULL: none */
} s3o_psfwd_desc_t;

/* hmm_t (hmm.h:156-173) */
typedef struct {
    int32_t score[5], hist[5], out_score, out_hist, bestscore, frame;
    uint16_t senid[5];      /* senone ids (plain) or senone-sequence ids (multiplexed) */
    uint16_t ssid;
    int16_t tmatid;
    uint8_t mpx, alloc;
} s3o_pshmm_t;

typedef struct { int32_t wid, score, bp, next; } s3o_pscand_t;          /* lastphn_cand_t */
typedef struct { int32_t sf, dscr, bp; } s3o_psltrans_t;                /* last_ltrans_t */
typedef struct { int32_t score, path, lc; } s3o_psbestrc_t;             /* bestbp_rc_t */

typedef struct s3o_psfwd_s {
    s3o_psfwd_desc_t d;             /* arrays are borrowed from the caller */
    int32_t n_ch, sp_base, rc_base, n_hmm;
    s3o_pshmm_t *hmm;               /* roots, interior, single-phone words, right-context channels */
    int32_t *w_sp;                  /* [n_words] index into sp_*, -1 */
    int32_t *acl[2], n_acl[2];      /* active_chan_list */
    int32_t *awl[2], n_awl[2];      /* active_word_list */
    uint8_t *word_active;
    s3o_pscand_t *cand; int32_t n_cand;
    s3o_psltrans_t *ltrans;
    int32_t *cand_sf_ef, *cand_sf_cand, cand_sf_alloc;
    s3o_psbestrc_t *bestrc;
    int32_t *word_lat_idx;
    /* bptbl_t by field */
    int32_t *bp_frame, *bp_wid, *bp_bp, *bp_score, *bp_sidx, *bp_realwid;
    uint8_t *bp_valid;
    int32_t bp_cap, bpidx;
    int32_t *bss; int32_t bss_cap, bss_head;
    int32_t *bp_table_idx; int32_t n_frame_alloc;
    int32_t n_frame, best_score, last_phone_best_score, renormalized, dynamic_beam;
    int32_t st_n_root_chan_eval, st_n_nonroot_chan_eval, st_n_last_chan_eval, st_n_word_lastchan_eval,
        st_n_lastphn_cand_utt, st_n_senone_active_utt;
    const int16_t *senscr;
    int32_t *pl;                    /* [n_ci] phone_loop_search_score of the frame (zeros without -pl_window) */
} s3o_psfwd_t;

s3o_psfwd_t *s3o_psfwd_init(const s3o_psfwd_desc_t *d);
void s3o_psfwd_free(s3o_psfwd_t *s);
void s3o_psfwd_set_lookahead(s3o_psfwd_t *s, const int32_t *pl);
void s3o_psfwd_reset(s3o_psfwd_t *s);
void s3o_psfwd_start(s3o_psfwd_t *s);
/* compute_sen_active: flags[n_sen] set to 0/1; returns the number of active senones */
int32_t s3o_psfwd_sen_active(s3o_psfwd_t *s, int32_t frame_idx, uint8_t *flags);
int32_t s3o_psfwd_step(s3o_psfwd_t *s, const int16_t *senscr, int32_t frame_idx, int32_t n_senone_active);
void s3o_psfwd_finish(s3o_psfwd_t *s, int32_t cf);
int32_t s3o_psfwd_find_exit(const s3o_psfwd_t *s, int32_t frame_idx, int32_t *out_best_score);
int32_t s3o_psfwd_exit_score(const s3o_psfwd_t *s, int32_t bp, int32_t rcphone);
int32_t s3o_psfwd_tg_score(const s3o_psfwd_t *s, int32_t w3, int32_t w2, int32_t w1);
/* ngram_search_bp2itor for every entry of the backtrace from bp (utterance order); returns the count */
int32_t s3o_psfwd_backtrace(const s3o_psfwd_t *s, int32_t bp, int32_t *wid, int32_t *sf, int32_t *ef,
                            int32_t *ascr, int32_t *lscr, int32_t *bps, int32_t max);
void s3o_psfwd_scalars(const s3o_psfwd_t *s, int32_t *out);
const int32_t *s3o_psfwd_array(const s3o_psfwd_t *s, int32_t which);
const uint8_t *s3o_psfwd_valid(const s3o_psfwd_t *s);
/* hmm_vit_eval on one HMM given plain arrays (unit tests of the device's evaluation) */
int32_t s3o_ps_hmm_vit_eval(s3o_pshmm_t *h, int32_t n_emit, const uint8_t *tp, const uint16_t *sseq,
                            const int16_t *senscr);
#endif
