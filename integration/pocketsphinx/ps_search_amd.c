/*
 * ps_search_amd.c -- what a pocketsphinx maintainer adds: the first pass (ngram_search_fwdtree.c) served by
 * libcmusphinx_amd behind the decoder's own ps_searchfuncs_t (pocketsphinx_internal.h:68-81).
 *
 *   ps_amd_search_install(ps, n_lanes)   after ps_init(): flattens the search's structures (psamd_export.h),
 *       builds the device engine (s3a_psfwd_init) and re-points start / step / finish (and free / reinit) of the
 *       decoder's N-gram search at the functions below.  Everything else of ps_decoder_t -- ps_start_utt,
 *       ps_process_raw / _cep, ps_end_utt, ps_get_hyp, ps_seg_iter, ps_get_lattice, fwdflat, bestpath -- is
 *       unchanged: finish copies the device's backpointer table into the ngram_search_t and then runs the
 *       reference's own finish, so the later passes and the result accessors work on it as they always do.
 *         start   ngram_search_start  (ngram_search.c:676)  -> s3a_psfwd_start
 *         step    ngram_search_step   (:691): compute_sen_active -> s3a_psfwd_sen_active, acmod_score (whatever
 *                 ps_mgau_t the decoder has: semi-continuous, PTM, or the device scorer of ps_mgau_amd.c),
 *                 ngram_fwdtree_search -> s3a_psfwd_step
 *         finish  ngram_search_finish (:722) -> s3a_psfwd_finish, s3a_psfwd_table
 *   ps_amd_decode_cep_batch(ps, ...)     whole utterances as LANES of the engine: features by the decoder's own
 *       feat_t, scoring (continuous models, "-senmgau .cont." / ms_mgau) and search on the device, hypotheses
 *       and segmentations made on the device (s3a_psfwd_decode / _hyp); what ps_decode_raw / ps_process_cep +
 *       ps_get_hyp + ps_seg_iter return for -fwdflat no -bestpath no, for n utterances at a time.
 *
 * -pl_window > 0 (round 6): step hands the decoder's own phone loop scores over (s3a_psfwd_set_lookahead); in whole-utterance
 * mode the loop runs on the device inside the lane's launch, pl_window frames ahead of the lane's search.
 * Refused (loudly, at install): interpolated or class-based LM sets, topologies other than 3 or 5
 * emitting states.  -lmname switching and ps_load_dict go through reinit, which re-exports.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sphinxbase/ckd_alloc.h>
#include <sphinxbase/err.h>
#include <sphinxbase/feat.h>
#include "pocketsphinx_internal.h"
#include "ngram_search.h"
#include "phone_loop_search.h"
#include "cmusphinx_amd.h"

#define PSAMD_DESC_T s3a_psfwd_desc_t
#include "psamd_export.h"

#define PSAMD_MAX_FRAMES 8192       /* frames per utterance the lanes are sized for unless PSAMD_MAX_FRAMES says otherwise (the reference grows
                                     * its tables up to 32 767 frames: an utterance beyond the lanes' size ends with an error from step, never
                                     * silently) */

typedef struct {
    ps_searchfuncs_t vt;            /* the decoder's table with the slots re-pointed; MUST be first */
    ps_searchfuncs_t *orig;
    ps_decoder_t *ps;
    s3a_psfwd_t *e;
    s3a_psfwd_desc_t desc;
    psamd_pool_t pool;
    s3a_ps_mgau_t *scorer;          /* whole-utterance mode, made on first use */
    int n_lanes;
    int32 max_frames;
    uint8 *flags;
    uint16 *sp_ssid;
    int32 *pl;                      /* [n_ci] -pl_window: the phone loop's scores of the frame */
} amd_search_t;

#define BINDING(search) ((amd_search_t *)(search)->vt)

static int
amd_build(amd_search_t *b, ngram_search_t *ngs)
{
    if (b->e) { s3a_psfwd_free(b->e); b->e = NULL; }
    psamd_pool_free(&b->pool);
    ckd_free(b->flags); ckd_free(b->sp_ssid); ckd_free(b->pl);
    b->flags = NULL; b->sp_ssid = NULL; b->pl = NULL;
    if (psamd_export(b->ps, ngs, &b->desc, &b->pool) < 0) return -1;
    {
        /* the lanes' capacities from the configuration: frames per utterance (this program's PSAMD_MAX_FRAMES variable, up to the
         * reference's own limit), backpointer entries per frame from -maxwpf (0: the library's default) */
        const char *mf = getenv("PSAMD_MAX_FRAMES");
        int32 max_frames = mf ? atoi(mf) : PSAMD_MAX_FRAMES;
        if (max_frames < 16) max_frames = 16;
        if (max_frames > 32767) max_frames = 32767;
        b->max_frames = max_frames;
    }
    if ((b->e = s3a_psfwd_init(&b->desc, b->n_lanes, b->max_frames, 0, 0)) == NULL) {
        E_ERROR("s3a_psfwd_init: %s\n", s3a_last_error());
        return -1;
    }
    b->flags = ckd_calloc(b->desc.n_sen, 1);
    b->sp_ssid = ckd_calloc((size_t)b->desc.n_1ph * b->desc.n_emit + 1, sizeof(uint16));
    b->pl = ckd_calloc(b->desc.n_ci + 1, sizeof(int32));
    return 0;
}

static int
amd_start(ps_search_t *search)
{
    ngram_search_t *ngs = (ngram_search_t *)search;
    amd_search_t *b = BINDING(search);
    int32 i, k, n_words = ps_search_n_words(ngs);
    ngs->done = FALSE;
    ngram_model_flush(ngs->lmset);
    ckd_free(search->hyp_str);
    search->hyp_str = NULL;
    /* what ngram_fwdtree_start resets of the decoder's own structures (ngram_search_fwdtree.c:472-494): a partial result asked for
     * before this utterance's table has come back must not see the last utterance's */
    memset(&ngs->st, 0, sizeof(ngs->st));
    ngs->bpidx = 0; ngs->bss_head = 0;
    for (i = 0; i < n_words; ++i) ngs->word_lat_idx[i] = NO_BP;
    ngs->n_active_chan[0] = ngs->n_active_chan[1] = 0;
    ngs->n_active_word[0] = ngs->n_active_word[1] = 0;
    ngs->best_score = 0; ngs->renormalized = 0; ngs->n_frame = 0;
    /* the single-phone words' channels are shared with the host's fwdflat pass, which leaves its ids in them */
    for (i = 0; i < b->desc.n_1ph; i++) {
        root_chan_t *r = (root_chan_t *)ngs->word_chan[b->desc.sp_wid[i]];
        for (k = 0; k < b->desc.n_emit; k++) b->sp_ssid[i * b->desc.n_emit + k] = r->hmm.senid[k];
    }
    if (s3a_psfwd_set_sp_ssid(b->e, 0, b->sp_ssid) != S3A_OK || s3a_psfwd_start(b->e, 0) != S3A_OK) {
        E_ERROR("s3a_psfwd_start: %s\n", s3a_last_error());
        return -1;
    }
    return 0;
}

static int
amd_step(ps_search_t *search, int frame_idx)
{
    amd_search_t *b = BINDING(search);
    acmod_t *acmod = ps_search_acmod(search);
    int16 const *senscr;
    int rv;
    if (!acmod->compallsen) {
        int32 s;
        if (s3a_psfwd_sen_active(b->e, 0, frame_idx, b->flags) != S3A_OK) { E_ERROR("s3a_psfwd_sen_active: %s\n", s3a_last_error()); return -1; }
        acmod_clear_active(acmod);
        for (s = 0; s < b->desc.n_sen; s++)
            if (b->flags[s]) acmod_activate_sen(acmod, s);
    }
    if ((senscr = acmod_score(acmod, &frame_idx)) == NULL) return 0;
    if (ps_search_lookahead(search)) {
        /* -pl_window: the decoder's phone loop (stepped by ps_search_forward in front of this search, pocketsphinx.c:704-712) says
         * what every transition of this frame adds, phone by phone (phone_loop_search.h:103-105) */
        phone_loop_search_t *pls = (phone_loop_search_t *)ps_search_lookahead(search);
        int32 ci;
        for (ci = 0; ci < b->desc.n_ci; ci++) b->pl[ci] = phone_loop_search_score(pls, ci);
        if (s3a_psfwd_set_lookahead(b->e, 0, b->pl) != S3A_OK) { E_ERROR("s3a_psfwd_set_lookahead: %s\n", s3a_last_error()); return -1; }
    }
    if ((rv = s3a_psfwd_step(b->e, 0, senscr, frame_idx, acmod->n_senone_active)) < 0) E_ERROR("s3a_psfwd_step: %s\n", s3a_last_error());
    return rv;
}

/* a lane's backpointer table into the reference's structures */
static int
amd_table_to_ngs(amd_search_t *b, ngram_search_t *ngs, int lane, int32 cf)
{
    s3a_psfwd_table_t t;
    int32 i;
    if (s3a_psfwd_table(b->e, lane, &t) != S3A_OK) { E_ERROR("s3a_psfwd_table: %s\n", s3a_last_error()); return -1; }
    while (ngs->bp_table_size <= t.bpidx) {
        ngs->bp_table_size *= 2;
        ngs->bp_table = ckd_realloc(ngs->bp_table, ngs->bp_table_size * sizeof(*ngs->bp_table));
    }
    while (ngs->bscore_stack_size <= t.bss_head + b->desc.n_ci) {
        ngs->bscore_stack_size *= 2;
        ngs->bscore_stack = ckd_realloc(ngs->bscore_stack, ngs->bscore_stack_size * sizeof(*ngs->bscore_stack));
    }
    for (i = 0; i <= cf; i++) ngram_search_mark_bptable(ngs, i);       /* grows bp_table_idx as the reference does */
    for (i = 0; i < t.bpidx; i++) {
        bptbl_t *be = &ngs->bp_table[i];
        const int32 w = t.wid[i];
        be->frame = t.frame[i]; be->valid = t.valid[i]; be->refcnt = 0; be->wid = w; be->bp = t.bp[i];
        be->score = t.score[i]; be->s_idx = t.s_idx[i]; be->real_wid = t.real_wid[i];
        be->last_phone = b->desc.w_last_ci[w]; be->last2_phone = b->desc.w_last2_ci[w];
    }
    memcpy(ngs->bscore_stack, t.bscore_stack, sizeof(int32) * t.bss_head);
    for (i = 0; i <= cf; i++) ngs->bp_table_idx[i] = t.bp_table_idx[i];
    ngs->bpidx = t.bpidx; ngs->bss_head = t.bss_head; ngs->n_frame = t.n_frame;
    ngs->best_score = t.best_score; ngs->last_phone_best_score = t.last_phone_best_score;
    ngs->renormalized = t.renormalized;
    ngs->st.n_root_chan_eval = t.st[1]; ngs->st.n_nonroot_chan_eval = t.st[2]; ngs->st.n_last_chan_eval = t.st[3];
    ngs->st.n_word_lastchan_eval = t.st[4]; ngs->st.n_lastphn_cand_utt = t.st[5]; ngs->st.n_senone_active_utt = t.st[6];
    ngs->n_active_chan[0] = ngs->n_active_chan[1] = 0;
    ngs->n_active_word[0] = ngs->n_active_word[1] = 0;
    return 0;
}

static int
amd_finish(ps_search_t *search)
{
    ngram_search_t *ngs = (ngram_search_t *)search;
    amd_search_t *b = BINDING(search);
    int32 cf = ps_search_acmod(search)->output_frame, i, k;
    if (s3a_psfwd_finish(b->e, 0, cf) != S3A_OK) { E_ERROR("s3a_psfwd_finish: %s\n", s3a_last_error()); return -1; }
    if (amd_table_to_ngs(b, ngs, 0, cf) < 0) return -1;
    if (s3a_psfwd_get_sp_ssid(b->e, 0, b->sp_ssid) != S3A_OK) return -1;
    for (i = 0; i < b->desc.n_1ph; i++) {
        root_chan_t *r = (root_chan_t *)ngs->word_chan[b->desc.sp_wid[i]];
        for (k = 0; k < b->desc.n_emit; k++) r->hmm.senid[k] = b->sp_ssid[i * b->desc.n_emit + k];
    }
    /* ngram_search_finish: ngram_fwdtree_finish (a no-op on the host's empty lists), fwdflat if enabled, done */
    return b->orig->finish(search);
}

/* partial results (ps_get_hyp / ps_seg_iter / ps_get_prob between ps_start_utt and ps_end_utt: normal live use): the device's table
 * as it stands comes into the decoder's structures first, then the reference's own code reads it (ngram_search.c:770-) */
static int
amd_pull_partial(ps_search_t *search)
{
    ngram_search_t *ngs = (ngram_search_t *)search;
    amd_search_t *b = BINDING(search);
    const int32 cf = ps_search_acmod(search)->output_frame;
    if (ngs->done || cf <= 0) return 0;
    return amd_table_to_ngs(b, ngs, 0, cf);
}

static char const *
amd_hyp(ps_search_t *search, int32 *out_score)
{
    if (amd_pull_partial(search) < 0) return NULL;
    return BINDING(search)->orig->hyp(search, out_score);
}

static int32
amd_prob(ps_search_t *search)
{
    if (amd_pull_partial(search) < 0) return 0;
    return BINDING(search)->orig->prob(search);
}

static ps_seg_t *
amd_seg_iter(ps_search_t *search, int32 *out_score)
{
    if (amd_pull_partial(search) < 0) return NULL;
    return BINDING(search)->orig->seg_iter(search, out_score);
}

static int
amd_reinit(ps_search_t *search, dict_t *dict, dict2pid_t *d2p)
{
    amd_search_t *b = BINDING(search);
    int rv = b->orig->reinit(search, dict, d2p);
    if (rv < 0) return rv;
    return amd_build(b, (ngram_search_t *)search);
}

static void
amd_free(ps_search_t *search)
{
    amd_search_t *b = BINDING(search);
    ps_searchfuncs_t *orig = b->orig;
    search->vt = orig;
    if (b->e) s3a_psfwd_free(b->e);
    if (b->scorer) s3a_ps_ms_mgau_free(b->scorer);
    psamd_pool_free(&b->pool);
    ckd_free(b->flags); ckd_free(b->sp_ssid); ckd_free(b->pl);
    ckd_free(b);
    orig->free(search);
}

int
ps_amd_search_install(ps_decoder_t *ps, int n_lanes)
{
    amd_search_t *b;
    if (ps->search == NULL || strcmp(ps_search_name(ps->search), "ngram") != 0) {
        E_ERROR("ps_amd_search_install: the decoder's search is not the N-gram search\n");
        return -1;
    }
    b = ckd_calloc(1, sizeof(*b));
    b->ps = ps;
    b->n_lanes = n_lanes > 0 ? n_lanes : 1;
    if (getenv("PSAMD_OVERLAP")) {     /* (this program's switch; the library takes it as a field of s3a_variants_t) */
        s3a_variants_t v;
        s3a_get_variants(&v);           /* (the other variants stay as the host set them) */
        v.ps_overlap = 1;
        s3a_set_variants(&v);
    }
    if (amd_build(b, (ngram_search_t *)ps->search) < 0) { ckd_free(b); return -1; }
    b->orig = ps->search->vt;
    b->vt = *b->orig;
    b->vt.start = amd_start; b->vt.step = amd_step; b->vt.finish = amd_finish; b->vt.reinit = amd_reinit; b->vt.free = amd_free;
    b->vt.hyp = amd_hyp; b->vt.prob = amd_prob; b->vt.seg_iter = amd_seg_iter;
    ps->search->vt = &b->vt;
    E_INFO("first pass served by %s: %d lanes, %d roots, %d interior channels, %d single-phone words\n", s3a_version(), b->n_lanes,
           b->desc.n_root, b->desc.n_nonroot, b->desc.n_1ph);
    return 0;
}

/* ---------------------------------------------------------------------------------------------------------- */
/* whole utterances as lanes                                                                                    */
/* ---------------------------------------------------------------------------------------------------------- */
static void
amd_dump_table(FILE *fh, amd_search_t *b, int lane, const char *uttid, int32 cf)
{
    /* the record format of oracle/ref_ps_fwd.c's -bpdump, from the device's table */
    s3a_psfwd_table_t t;
    int32 hdr[16], i, len = (int32)strlen(uttid);
    if (s3a_psfwd_table(b->e, lane, &t) != S3A_OK) E_FATAL("s3a_psfwd_table: %s\n", s3a_last_error());
    memset(hdr, 0, sizeof hdr);
    hdr[0] = 0x50534250; hdr[1] = len; hdr[2] = t.n_frame; hdr[3] = t.bpidx; hdr[4] = t.bss_head; hdr[5] = t.best_score;
    hdr[6] = t.last_phone_best_score; hdr[7] = t.renormalized;
    for (i = 0; i < 6; i++) hdr[8 + i] = t.st[1 + i];
    hdr[14] = cf;
    fwrite(hdr, 4, 16, fh);
    fwrite(uttid, 1, len, fh);
    for (i = 0; i < t.bpidx; i++) {
        int32 rec[7] = { t.frame[i], t.wid[i], t.bp[i], t.score[i], t.s_idx[i], t.real_wid[i], t.valid[i] };
        fwrite(rec, 4, 7, fh);
    }
    fwrite(t.bscore_stack, 4, t.bss_head, fh);
    fwrite(t.bp_table_idx, 4, cf + 1, fh);
}

int
ps_amd_decode_cep_batch(ps_decoder_t *ps, int n_utt, mfcc_t ***cep, const int *n_frames, int fresh, char **out_hyp,
                        int32 *out_score, FILE *segfh, char **uttids, FILE *bpfh)
{
    amd_search_t *b;
    cmd_ln_t *config = ps->config;
    feat_t *fcb = ps->acmod->fcb;
    const float **rows;
    mfcc_t ****feats;
    int32 *nfr;
    int z, rv = -1;
    if (ps->search == NULL || ps->search->vt->start != amd_start) { E_ERROR("ps_amd_decode_cep_batch: ps_amd_search_install first\n"); return -1; }
    b = BINDING(ps->search);
    /* more utterances than lanes: the lanes take them from a queue (s3a_psfwd_decode_queue; every utterance from a new
     * decoder's state; no backpointer-table dump) */
    const int queue = n_utt > b->n_lanes;
    if (queue && (bpfh || !fresh)) { E_ERROR("ps_amd_decode_cep_batch: %d utterances on %d lanes go through the queue: -fresh yes, no -bpdump\n", n_utt, b->n_lanes); return -1; }
    if (feat_dimension1(fcb) != 1) { E_ERROR("ps_amd_decode_cep_batch: one feature stream served\n"); return -1; }
    if (b->scorer == NULL) {
        if (strcmp(ps->acmod->mgau->vt->name, "ms") != 0) {
            E_ERROR("ps_amd_decode_cep_batch: the acoustic model is not served by the multi-stream (continuous) scorer\n");
            return -1;
        }
        /* the values ms_mgau_init reads from the configuration (ms_mgau.c:88-103) */
        b->scorer = s3a_ps_ms_mgau_init(cmd_ln_str_r(config, "-mean"), cmd_ln_str_r(config, "-var"),
                                        cmd_ln_float32_r(config, "-varfloor"), cmd_ln_str_r(config, "-mixw"),
                                        cmd_ln_float32_r(config, "-mixwfloor"), cmd_ln_str_r(config, "-senmgau"),
                                        cmd_ln_int32_r(config, "-topn"), cmd_ln_int32_r(config, "-aw"),
                                        cmd_ln_float32_r(config, "-logbase"));
        if (b->scorer == NULL) { E_ERROR("s3a_ps_ms_mgau_init: %s\n", s3a_last_error()); return -1; }
    }
    rows = ckd_calloc(n_utt, sizeof(*rows));
    feats = ckd_calloc(n_utt, sizeof(*feats));
    nfr = ckd_calloc(n_utt, sizeof(*nfr));
    for (z = 0; z < n_utt; z++) {
        /* acmod_process_full_cep (acmod.c:620-660): the whole utterance through feat_s2mfc2feat_live */
        int32 n = n_frames[z];
        feats[z] = feat_array_alloc(fcb, n > 0 ? n : 1);
        nfr[z] = n > 0 ? feat_s2mfc2feat_live(fcb, cep[z], &n, TRUE, TRUE, feats[z]) : 0;
        rows[z] = (const float *)feats[z][0][0];
    }
    if ((queue ? s3a_psfwd_decode_queue(b->e, b->scorer, n_utt, rows, nfr, cmd_ln_boolean_r(config, "-compallsen"))
               : s3a_psfwd_decode(b->e, b->scorer, n_utt, rows, nfr, cmd_ln_boolean_r(config, "-compallsen"), fresh)) != S3A_OK) {
        E_ERROR("s3a_psfwd_decode%s: %s\n", queue ? "_queue" : "", s3a_last_error());
        goto done;
    }
    {
        int32 tot = 0;
        double ms = s3a_psfwd_last_decode_ms(b->e);
        for (z = 0; z < n_utt; z++) tot += nfr[z];
        E_INFO("batch of %d utterances, %d frames: %.2f ms on the device (scoring + search; %.0f frames/s)%s\n", n_utt, tot, ms,
               ms > 0 ? tot / (ms * 1e-3) : 0.0, queue ? " [queue]" : "");
        if (queue) E_INFO("of which scoring %.2f ms\n", s3a_psfwd_last_score_ms(b->e));
    }
    for (z = 0; z < n_utt; z++) {
        static s3a_psfwd_seg_t seg[4096];
        int32 score = 0, n, i;
        size_t len = 0;
        char *c;
        if ((n = queue ? s3a_psfwd_queue_hyp(b->e, z, &score, seg, 4096) : s3a_psfwd_hyp(b->e, z, &score, seg, 4096)) < 0) { E_ERROR("s3a_psfwd_hyp: %s\n", s3a_last_error()); goto done; }
        out_score[z] = score;
        /* ngram_search_bp_hyp (ngram_search.c:486-539): the real words' base strings */
        for (i = 0; i < n; i++)
            if (dict_real_word(ps->dict, seg[i].wid)) len += strlen(dict_basestr(ps->dict, seg[i].wid)) + 1;
        out_hyp[z] = NULL;
        if (len > 0) {
            c = out_hyp[z] = ckd_calloc(1, len);
            for (i = 0; i < n; i++)
                if (dict_real_word(ps->dict, seg[i].wid)) {
                    const char *s = dict_basestr(ps->dict, seg[i].wid);
                    if (c > out_hyp[z]) *c++ = ' ';
                    memcpy(c, s, strlen(s)); c += strlen(s);
                }
        }
        if (segfh) {
            fprintf(segfh, "%s", uttids[z]);
            for (i = 0; i < n; i++)
                fprintf(segfh, " %s %d %d %d %d", dict_wordstr(ps->dict, seg[i].wid), seg[i].sf, seg[i].ef, seg[i].ascr, seg[i].lscr);
            fprintf(segfh, "\n");
        }
        if (bpfh) amd_dump_table(bpfh, b, z, uttids[z], nfr[z]);
    }
    rv = 0;
done:
    for (z = 0; z < n_utt; z++) feat_array_free(feats[z]);
    ckd_free(rows); ckd_free(feats); ckd_free(nfr);
    return rv;
}
