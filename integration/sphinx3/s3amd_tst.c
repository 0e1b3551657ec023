/*
 * integration/sphinx3/s3amd_tst.c -- the sphinx3 side of the drop-in: sphinx3_decode (mode 4,
 * "fwdtree") with the WHOLE per-frame hot path served by a replacement backend through the
 * reference's own srch_funcs_t table (sphinx3/include/srch.h:528-701).  This is the file a
 * sphinx3 maintainer adds; it is compiled against the unmodified reference
 * where that lies (oracle/Makefile -> oracle/_ref/, never committed) and run as a test.
 *
 * Built by oracle/Makefile into oracle/_ref/ref_s3amd_tst_decode: the MI355X backend (include/cmusphinx_amd.h).
 *        S3A_UTT=L       whole utterances on the device, L at a time: senone scoring, lextree search AND the
 *                        word level (trigram look-ups, Viterbi history, pruning, word transitions) run as
 *                        kernels with no host synchronisation inside an utterance (the `decode` slot,
 *                        srch.h:552-555, srch.c:673-675); the host reads the finished history table back,
 *                        hands it to the reference's own vithist_utt_end / backtrace / output code.
 *        (otherwise)     frame-synchronous: scoring + lextree search on the device, the reference's own
 *                        vithist / LM on the host (one synchronisation per frame); S3A_STREAMS / S3A_BATCH.
 * (The programs that serve the same slots from the CPU restatement, and so pin it, are oracle/ref_s3o_tst_decode.c.)
 *
 * What stays the reference's: kb_init (models, dictionary, LM, lextree_build, dict2pid), feature
 * computation, the utterance API, hypothesis output (and, frame-synchronous mode, vithist_* / lm_*).
 * The lextrees the reference built are FLATTENED once (flatten_tree) into the node/CSR arrays the
 * backends take, its lm_t / dict_t into the plain arrays of s3a_lm3g_init / s3a_wordlevel_init
 * (flatten_lm).  The replaced slots restate the control flow of srch_time_switch_tree.c:457-560
 * (begin/end), :776-907 (hmm_compute_lv2), :923-1007 (propagate_graph_ph_lv2), :1010-1210 (rescoring,
 * word transitions), :1213-1237 (frame_windup), :1262-1324 (select_active_gmm).
 */
#define main sphinx3_decode_reference_main
#include "main_decode.c"        /* the reference's file, in place (for its arg table) */
#undef main

/* srch_TST_graph_t (the mode's private graph structure) is defined INSIDE the reference's
 * srch_time_switch_tree.c:233-257, not in a header.  A maintainer would add these slots in
 * that file; here the file is #included in place (not copied) to obtain the definition. */
#include "srch_time_switch_tree.c"

#include <string.h>
#include <math.h>
#include "pio.h"
#include "genrand.h"
#include "cmn.h"
#include "fixpoint.h"
#include "logs3.h"
#include "byteorder.h"
#include "s3_decode.h"
#include "srch.h"
#include "gmm_wrap.h"
#include "dict2pid.h"
#include "lextree.h"
#include "vithist.h"
#include "dag.h"

#include "cmusphinx_amd.h"

#include "s3amd_flatten.h"

/* ------------------------------------------------------------------ */
/* backend                                                             */
/* ------------------------------------------------------------------ */
static __thread int32 g_ntree;           /* 2 * n_lextree: unigram trees then filler trees */
static __thread flat_t **g_flat;
static __thread int32 *g_tp_flat;        /* tmat->tp flattened */
static __thread int16 *g_sseq_flat, *g_comsseq_flat, *g_comstate;
static __thread int32 *g_comstate_off, g_n_comstate;
static __thread int32 g_max_node;
static __thread int32 *g_best, *g_wbest, *g_nact;
static __thread int32 *g_exit_n, *g_exit_wid, *g_exit_scr, *g_exit_hist;
static __thread long g_frames, g_histframes;
#include <time.h>
static __thread double g_t_score, g_t_search, g_t_word, g_t_utt;
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static __thread s3a_logmath_t *g_lm;
static __thread s3a_mgau_model_t *g_gm;
static __thread s3a_scorer_t *g_sc;
static __thread s3a_comsen_t *g_cs;
static __thread s3a_tmat_t *g_tm;
static __thread s3a_lexsearch_t *g_ls;
/* S3A_BATCH=1: all decoder threads share every kernel launch through one s3a_batch_t */
#define MAX_GROUPS 8
static s3a_batch_t *g_batches[MAX_GROUPS];       /* S3A_BATCH=G: G engines, decoder i steps with group i % G, so
                                                 * one group's host phase overlaps the others' device phase */
static int g_n_groups;
static __thread s3a_batch_t *g_batch;
static __thread int g_worker_id;
static s3a_logmath_t *g_lm_shared[MAX_GROUPS];   /* batch mode: ONE model on the device per group of decoders */
static s3a_mgau_model_t *g_gm_shared[MAX_GROUPS];
static s3a_lexsearch_t *g_ls_shared[MAX_GROUPS];  /* ... and one copy of the static lextree arrays */
static __thread int32 g_slot;
static __thread float32 g_featbuf[64];
static __thread int32 g_feat_idx;
/* lextree_enter calls of the current frame, waiting for the swap (see be_enter) */
#define PEND_MAX 1024
static __thread int32 g_pend_tree[2], g_pend_n[2], g_pend_cf, g_pend_thresh;
static __thread int32 g_pend_lc[2][PEND_MAX], g_pend_scr[2][PEND_MAX], g_pend_hist[2][PEND_MAX];
static void die(const char *w) { E_FATAL("tst shim: %s: %s\n", w, s3a_last_error()); }


/* -mllr / -ctl_mllr: kb_setmllr (kb.c:335-365 -> adapt_set_mllr, libam/adaptor.c:106-170) has rewritten the host model
 * -- means and variances reloaded, mllr_norm_mgau, variance floor, mgau_precomp --; the device model takes the result
 * (s3a_mgau_set_params).  g_mllr_cur: the regression matrix file the device models hold ("": none). */
static char g_mllr_cur[4096];
static void
adapt_upload(kb_t *kb, s3a_mgau_model_t *gm)
{
    mgau_model_t *g = kbcore_mgau(kb->kbcore);
    const int32 S = mgau_n_mgau(g), C = mgau_max_comp(g), D = mgau_veclen(g);
    float *mean, *prec, *lrd;
    int32 m, c;
    if (S != s3a_mgau_n_mgau(gm) || C != s3a_mgau_max_comp(gm) || D != s3a_mgau_veclen(gm))
        E_FATAL("tst shim: the adapted model's shape differs from the device model's\n");
    mean = ckd_calloc((size_t)S * C * D, sizeof(float)); prec = ckd_calloc((size_t)S * C * D, sizeof(float));
    lrd = ckd_calloc((size_t)S * C, sizeof(float));
    for (m = 0; m < S; m++) {
        if (mgau_n_comp(g, m) != s3a_mgau_n_comp(gm, m))
            E_FATAL("tst shim: senone %d has %d components after adaptation, the device model %d\n", m, mgau_n_comp(g, m), s3a_mgau_n_comp(gm, m));
        for (c = 0; c < mgau_n_comp(g, m); c++) {
            memcpy(mean + ((size_t)m * C + c) * D, mgau_mean(g, m, c), D * sizeof(float));
            memcpy(prec + ((size_t)m * C + c) * D, mgau_var(g, m, c), D * sizeof(float));
            lrd[(size_t)m * C + c] = mgau_lrd(g, m, c);
        }
    }
    if (s3a_mgau_set_params(gm, mean, prec, lrd) != S3A_OK) die("s3a_mgau_set_params");
    ckd_free(mean); ckd_free(prec); ckd_free(lrd);
}

/* the device model(s) follow the host model: gms[0 .. n) all take it when the regression matrix file has changed */
static void
adapt_sync(kb_t *kb, s3a_mgau_model_t **gms, int32 n)
{
    const char *now = (kb->adapt_am && kb->adapt_am->prevmllrfn) ? kb->adapt_am->prevmllrfn : "";
    int32 e;
    if (strcmp(now, g_mllr_cur) == 0) return;
    if (strlen(now) >= sizeof g_mllr_cur) E_FATAL("tst shim: MLLR file name too long\n");
    for (e = 0; e < n; e++) adapt_upload(kb, gms[e]);
    strcpy(g_mllr_cur, now);
    E_INFO("tst shim: the device model%s now hold%s the model adapted with %s\n", n > 1 ? "s" : "", n > 1 ? "" : "s", now);
}

/* ctl_process callback of the frame-synchronous drivers: utt_decode (libAPI/utt.c:185) behind the model switch it would
 * make itself (:245-246), so that the device model has switched too before the first frame is scored */
static void
utt_decode_adapt(void *data, utt_res_t *ur, int32 sf, int32 ef, char *uttid)
{
    kb_t *kb = data;
    if (ur->regmatname != NULL) {
        if (g_n_groups) E_FATAL("tst shim: -ctl_mllr with S3A_BATCH (decoders share one device model) is not supported\n");
        kb_setmllr(ur->regmatname, ur->cb2mllrname, kb);
        adapt_sync(kb, &g_gm, 1);
    }
    utt_decode(data, ur, sf, ef, uttid);
}

/* the lextrees the search walks now (the CURRENT LM's unigram trees + the filler trees), flattened */
static void
flatten_current_trees(srch_TST_graph_t *tstg)
{
    int32 i;
    g_ntree = 2 * tstg->n_lextree;
    g_flat = ckd_calloc(g_ntree, sizeof(*g_flat));
    for (i = 0; i < g_ntree; i++) {
        lextree_t *lt = (i < tstg->n_lextree) ? tstg->curugtree[i] : tstg->fillertree[i - tstg->n_lextree];
        g_flat[i] = flatten_tree(lt);
        if (g_flat[i]->n_node > g_max_node) g_max_node = g_flat[i]->n_node;
    }
}

/* ... and the device search space made of them (g_tm, the senone-sequence tables and the model's stream exist) */
static s3a_lexsearch_t *
make_lexsearch(mdef_t *mdef, dict2pid_t *d2p)
{
    const int32 **ssid = ckd_calloc(g_ntree, sizeof(void *)), **tm = ckd_calloc(g_ntree, sizeof(void *));
    const int32 **wid = ckd_calloc(g_ntree, sizeof(void *)), **prob = ckd_calloc(g_ntree, sizeof(void *));
    const int32 **coff = ckd_calloc(g_ntree, sizeof(void *)), **ch = ckd_calloc(g_ntree, sizeof(void *));
    const int32 **lro = ckd_calloc(g_ntree, sizeof(void *)), **lr = ckd_calloc(g_ntree, sizeof(void *));
    const int32 **root = ckd_calloc(g_ntree, sizeof(void *));
    const uint8 **comp = ckd_calloc(g_ntree, sizeof(void *));
    const int16 **lc = ckd_calloc(g_ntree, sizeof(void *));
    int32 *nn = ckd_calloc(g_ntree, 4), *nlc = ckd_calloc(g_ntree, 4), *nroot = ckd_calloc(g_ntree, 4), i;
    s3a_lexsearch_t *ls;
    for (i = 0; i < g_ntree; i++) {
        flat_t *f = g_flat[i];
        nn[i] = f->n_node; ssid[i] = f->ssid; tm[i] = f->tmatid; comp[i] = f->composite;
        wid[i] = f->wid; prob[i] = f->prob; coff[i] = f->child_off; ch[i] = f->child;
        nlc[i] = f->n_lc; lc[i] = f->lc; lro[i] = f->lcroot_off; lr[i] = f->lcroot;
        nroot[i] = f->n_root; root[i] = f->root;
    }
    ls = s3a_lexsearch_init(g_ntree, nn, ssid, tm, comp, wid, prob, coff, ch, nlc, lc, lro, lr,
                            nroot, root, g_tm, g_sseq_flat, mdef_n_sseq(mdef), g_comsseq_flat,
                            d2p->n_comsseq, g_n_comstate, g_comstate_off, g_comstate,
                            s3a_mgau_stream(g_gm));
    ckd_free(ssid); ckd_free(tm); ckd_free(wid); ckd_free(prob); ckd_free(coff); ckd_free(ch); ckd_free(lro); ckd_free(lr);
    ckd_free(root); ckd_free(comp); ckd_free(lc); ckd_free(nn); ckd_free(nlc); ckd_free(nroot);
    return ls;
}

static void
backend_init(kb_t *kb, srch_TST_graph_t *tstg)
{
    kbcore_t *kbc = kb->kbcore;
    mdef_t *mdef = kbcore_mdef(kbc);
    dict2pid_t *d2p = kbcore_dict2pid(kbc);
    tmat_t *tmat = kbcore_tmat(kbc);
    int32 ne = mdef_n_emit_state(mdef), i, j, k, n;

    if (!dict2pid_is_composite(d2p))
        E_FATAL("tst shim: full cross-word triphone expansion grows the lextree during search; "
                "only composite triphones (the reference's default) are supported\n");
    if (kb->pl->pheurtype != 0 && !getenv("S3A_UTT"))
        E_FATAL("tst shim: -pheurtype > 0 is served by the whole-utterance engine only (S3A_UTT=lanes)\n");
    /* several LMs (-lmctlfn; every LM has unigram lextrees of its own, srch_time_switch_tree.c:260-330): the whole-utterance
     * mode keeps a search space and engines per LM and switches between utterances (s3amd_uttmode.h) */
    if (kbcore_lmset(kbc)->n_lm != 1 && !getenv("S3A_UTT"))
        E_FATAL("tst shim: several LMs (-lmctlfn) are served by the whole-utterance engine only (S3A_UTT=lanes)\n");

    flatten_current_trees(tstg);
    g_tp_flat = ckd_calloc(tmat->n_tmat * ne * (ne + 1), 4);
    for (i = 0; i < tmat->n_tmat; i++)
        for (j = 0; j < ne; j++)
            for (k = 0; k <= ne; k++)
                g_tp_flat[(i * ne + j) * (ne + 1) + k] = tmat->tp[i][j][k];
    g_sseq_flat = ckd_calloc(mdef_n_sseq(mdef) * ne, 2);
    for (i = 0; i < mdef_n_sseq(mdef); i++)
        for (j = 0; j < ne; j++)
            g_sseq_flat[i * ne + j] = mdef->sseq[i][j];
    g_comsseq_flat = ckd_calloc(d2p->n_comsseq * ne + 1, 2);
    for (i = 0; i < d2p->n_comsseq; i++)
        for (j = 0; j < ne; j++)
            g_comsseq_flat[i * ne + j] = d2p->comsseq[i][j];
    for (i = 0, n = 0; i < d2p->n_comstate; i++)
        for (j = 0; IS_S3SENID(d2p->comstate[i][j]); j++)
            n++;
    g_n_comstate = d2p->n_comstate;
    g_comstate_off = ckd_calloc(g_n_comstate + 1, 4);
    g_comstate = ckd_calloc(n + 1, 2);
    for (i = 0, n = 0; i < g_n_comstate; i++) {
        g_comstate_off[i] = n;
        for (j = 0; IS_S3SENID(d2p->comstate[i][j]); j++)
            g_comstate[n++] = d2p->comstate[i][j];
    }
    g_comstate_off[g_n_comstate] = n;

    g_best = ckd_calloc(g_ntree, 4); g_wbest = ckd_calloc(g_ntree, 4); g_nact = ckd_calloc(g_ntree, 4);
    g_exit_n = ckd_calloc(g_ntree, 4);
    g_exit_wid = ckd_calloc(g_ntree * g_max_node, 4);
    g_exit_scr = ckd_calloc(g_ntree * g_max_node, 4);
    g_exit_hist = ckd_calloc(g_ntree * g_max_node, 4);

    {
        cmd_ln_t *config = kbcore_config(kbc);
        if (s3a_device_count() < 1)
            E_FATAL("tst shim: no GPU; libcmusphinx_amd has no CPU fallback\n");
        if (kbcore_svq(kbc) || kbcore_gs(kbc) || !kbcore_mgau(kbc))
            E_FATAL("tst shim: only plain -senmgau .cont. scoring is supported\n");
        g_batch = g_n_groups ? g_batches[g_worker_id % g_n_groups] : NULL;
        if (g_batch && g_gm_shared[g_worker_id % g_n_groups]) {         /* (called under g_init_lock) */
            g_lm = g_lm_shared[g_worker_id % g_n_groups];
            g_gm = g_gm_shared[g_worker_id % g_n_groups];
        }
        else {
            g_lm = s3a_logs3_init(cmd_ln_float64_r(config, "-logbase"), 0, 1);
            g_gm = s3a_mgau_init(cmd_ln_str_r(config, "-mean"), cmd_ln_str_r(config, "-var"),
                                 cmd_ln_float32_r(config, "-varfloor"), cmd_ln_str_r(config, "-mixw"),
                                 cmd_ln_float32_r(config, "-mixwfloor"), 1, ".cont.",
                                 S3A_MIX_INT_FLOAT_COMP, g_lm);
            if (!g_gm) die("s3a_mgau_init");
            if (g_batch) { g_lm_shared[g_worker_id % g_n_groups] = g_lm; g_gm_shared[g_worker_id % g_n_groups] = g_gm; }
        }
        g_sc = (g_batch ? s3a_scorer_init_private : s3a_scorer_init)(
                               g_gm, mdef->cd2cisen, mdef_n_sen(mdef), mdef->n_ci_sen,
                               cmd_ln_int32_r(config, "-ds"), cmd_ln_int32_r(config, "-cond_ds"),
                               cmd_ln_float64_r(config, "-ci_pbeam"),
                               cmd_ln_float32_r(config, "-tighten_factor"),
                               cmd_ln_int32_r(config, "-maxcdsenpf"));
        if (!g_sc) die("s3a_scorer_init");
        g_cs = s3a_comsen_init(g_n_comstate, g_comstate_off, g_comstate, d2p->comwt);
        if (!g_cs) die("s3a_comsen_init");
        g_tm = s3a_tmat_init_logs3(g_tp_flat, tmat->n_tmat, ne);
        if (g_batch && g_ls_shared[g_worker_id % g_n_groups])
            g_ls = s3a_lexsearch_clone(g_ls_shared[g_worker_id % g_n_groups], s3a_mgau_stream(g_gm));
        else {
            g_ls = make_lexsearch(mdef, d2p);
            if (g_batch) g_ls_shared[g_worker_id % g_n_groups] = g_ls;
        }
        if (!g_ls) die("s3a_lexsearch_init");
        if (kb->adapt_am && kb->adapt_am->prevmllrfn && kb->adapt_am->prevmllrfn[0]) {     /* -mllr: kb_init adapted the host model */
            if (g_n_groups) E_FATAL("tst shim: -mllr with S3A_BATCH is not supported\n");
            adapt_sync(kb, &g_gm, 1);
        }
    }
    E_INFO("tst shim: %d lextrees flattened (largest %d nodes), backend %s\n", g_ntree, g_max_node,
           s3a_version()
        );
}

/* one batch of lextree_enter calls into tree t */
static void
be_enter(int32 t, int32 n, int32 *lc, int32 *scr, int32 *hist, int32 cf, int32 thresh)
{
    /* fused device frame: the calls of a frame (one unigram tree, then one filler tree) are
     * collected and issued together with the swap by be_swap -> s3a_decoder_transition */
    int32 g = (t >= g_ntree / 2), c;
    if (n == 0) return;
    if (g_pend_n[g] || n > PEND_MAX) E_FATAL("tst shim: unexpected lextree_enter pattern\n");
    g_pend_tree[g] = t; g_pend_n[g] = n; g_pend_cf = cf; g_pend_thresh = thresh;
    for (c = 0; c < n; c++) { g_pend_lc[g][c] = lc[c]; g_pend_scr[g][c] = scr[c]; g_pend_hist[g][c] = hist[c]; }
}

static void
be_swap(int32 cf)
{
    /* the unigram-tree batch may be empty while the filler batch is not: keep them apart by slot */
    if (g_batch) {
        if (s3a_batch_transition(g_batch, g_slot, g_pend_n[0] || g_pend_n[1] ? g_pend_cf : cf, g_pend_thresh,
                                 g_pend_tree[0], g_pend_n[0], g_pend_lc[0], g_pend_scr[0], g_pend_hist[0],
                                 g_pend_tree[1], g_pend_n[1], g_pend_lc[1], g_pend_scr[1], g_pend_hist[1]) != S3A_OK)
            die("batch transition");
    }
    else if (s3a_decoder_transition(g_ls, g_sc, g_cs, g_pend_n[0] || g_pend_n[1] ? g_pend_cf : cf,
                               g_pend_thresh, g_pend_tree[0], g_pend_n[0], g_pend_lc[0], g_pend_scr[0],
                               g_pend_hist[0], g_pend_tree[1], g_pend_n[1], g_pend_lc[1], g_pend_scr[1],
                               g_pend_hist[1]) != S3A_OK) die("transition");
    g_pend_n[0] = g_pend_n[1] = 0;
}

/* ------------------------------------------------------------------ */
/* replaced srch_funcs_t slots                                         */
/* ------------------------------------------------------------------ */
static int
tst_begin(void *srch)
{
    srch_t *s = srch;
    srch_TST_graph_t *tstg = s->grh->graph_struct;
    kbcore_t *kbc = s->kbc;
    mgau_model_t *g = kbc->mgau;
    int32 pred, i, lc, zero = 0;

    g_t_utt -= now_s();
    vithist_utt_reset(tstg->vithist);
    histprune_zero_histbin(tstg->histprune);
    pred = vithist_utt_begin(tstg->vithist, kbc);
    if (g)
        for (i = 0; i < g->n_mgau; i++) { g->mgau[i].bstidx = NO_BSTIDX; g->mgau[i].updatetime = NOT_UPDATED; }
    if ((g_batch ? s3a_batch_utt_begin(g_batch, g_slot) : s3a_decoder_utt_begin(g_ls, g_sc)) != S3A_OK)
        die("decoder_utt_begin");
    g_pend_n[0] = g_pend_n[1] = 0;
    lc = mdef_silphone(kbc->mdef);
    be_enter(0, 1, &lc, &zero, &pred, -1, s->beam->hmm);
    lc = BAD_S3CIPID;
    be_enter(tstg->n_lextree, 1, &lc, &zero, &pred, -1, s->beam->hmm);
    tstg->n_lextrans = 1;
    be_swap(-1);
    return SRCH_SUCCESS;
}

static int
tst_end(void *srch)
{
    srch_t *s = srch;
    srch_TST_graph_t *tstg = s->grh->graph_struct;
    int32 t;
    g_t_utt += now_s();
    s->exit_id = vithist_utt_end(tstg->vithist, s->kbc);
    s->stat->utt_wd_exit = vithist_n_entry(tstg->vithist);
    histprune_showhistbin(tstg->histprune, s->stat->nfr, s->uttid);
    (void)t;
    if ((g_batch ? s3a_batch_utt_end(g_batch, g_slot) : s3a_lexsearch_utt_end(g_ls)) != S3A_OK) die("utt_end");
    lm_cache_stats_dump(kbcore_lm(s->kbc));
    lm_cache_reset(kbcore_lm(s->kbc));
    return (s->exit_id >= 0) ? SRCH_SUCCESS : SRCH_FAILURE;
}

static __thread int32 g_ascale_idx;

/* the CI senones are scored on the device inside gmm_compute_lv2 (no host cache needed) */
static int
tst_gmm_lv1(void *srch, float32 *feat, int32 cache_idx, int32 wav_idx)
{
    return SRCH_SUCCESS;
}

static int
tst_select_active(void *srch)
{
    /* the senones of the coming frame were marked by the previous s3a_decoder_transition */
    return SRCH_SUCCESS;
}

/* enqueue only: CI gate + CD senones (raw scores; the search kernels subtract the frame's
 * best); nothing read back.  srch.c:752 copies s->senscale into ascale[] right after this
 * slot returns; the real value arrives with the frame's single read-back and is patched in. */
static int
tst_gmm_lv2(void *srch, float32 **feat, int32 wav_idx)
{
    srch_t *s = srch;
    if (g_batch) {          /* scored inside the batched step (propagate_graph_wd_lv2 slot) */
        memcpy(g_featbuf, feat[0], sizeof(float32) * s3a_mgau_veclen(g_gm));
        g_feat_idx = wav_idx;
    }
    else if (s3a_decoder_score(g_sc, feat[0], wav_idx) != S3A_OK)
        die("lv2");
    g_ascale_idx = s->num_frm + wav_idx;
    s->senscale = 0;
    return SRCH_SUCCESS;
}

static int
tst_hmm_compute_lv2(void *srch, int32 frmno)
{
    srch_t *s = srch;
    srch_TST_graph_t *tstg = s->grh->graph_struct;
    histprune_t *hp = tstg->histprune;
    beam_t *bm = s->beam;
    int32 besthmmscr = MAX_NEG_INT32, bestwordscr = MAX_NEG_INT32, frm_nhmm = 0, t, hb, pb, wb;

    /* device backend: evaluation, thresholds, propagation and word exits are ONE enqueue
     * with one read-back, issued from the propagate_graph_wd_lv2 slot */
    (void)besthmmscr; (void)bestwordscr; (void)frm_nhmm; (void)t; (void)hb; (void)pb; (void)wb;
    (void)hp; (void)bm;
    g_frames++;
    return SRCH_SUCCESS;
}

static int
tst_propagate_ph_lv2(void *srch, int32 frmno)
{
    srch_t *s = srch;
    beam_t *bm = s->beam;
    int32 pth = bm->phone_thres;
    if (bm->ptranskip != 0 && (frmno % bm->ptranskip) == 0)
        pth = bm->word_thres;           /* srch_time_switch_tree.c:975-1003 */
    (void)pth;      /* done inside s3a_lexsearch_frame_search (propagate_graph_wd_lv2 slot) */
    return SRCH_SUCCESS;
}

/* srch_utt_word_trans, srch_time_switch_tree.c:1087-1179 */
static void
tst_word_trans(srch_t *s, int32 cf)
{
    srch_TST_graph_t *tstg = s->grh->graph_struct;
    vithist_t *vh = tstg->vithist;
    beam_t *bm = s->beam;
    dict_t *dict = kbcore_dict(s->kbc);
    mdef_t *mdef = kbcore_mdef(s->kbc);
    int32 n_ci = mdef_n_ciphone(mdef), th = bm->bestscore + bm->hmm;
    int32 *bs = bm->wordbestscores, *bv = bm->wordbestexits;
    int32 p, vhid, le, k, n, maxpscore = MAX_NEG_INT32;
    static __thread int32 *c_lc, *c_scr, *c_hist;

    if (vh->bestvh[cf] < 0)
        return;
    if (!c_lc) { c_lc = ckd_calloc(n_ci + 1, 4); c_scr = ckd_calloc(n_ci + 1, 4); c_hist = ckd_calloc(n_ci + 1, 4); }
    for (p = 0; p < n_ci; p++) { bs[p] = MAX_NEG_INT32; bv[p] = -1; }
    vhid = vithist_first_entry(vh, cf);
    le = vithist_n_entry(vh) - 1;
    for (; vhid <= le; vhid++) {
        vithist_entry_t *ve = vithist_id2entry(vh, vhid);
        int32 score;
        if (!vithist_entry_valid(ve))
            continue;
        p = dict_last_phone(dict, vithist_entry_wid(ve));
        if (mdef_is_fillerphone(mdef, p))
            p = mdef_silphone(mdef);
        score = vithist_entry_score(ve);
        if (score > bs[p]) {
            bs[p] = score;
            bv[p] = vhid;
            if (maxpscore < score) maxpscore = score;
        }
    }
    k = tstg->n_lextrans++;
    k = (k % (tstg->n_lextree * tstg->epl)) / tstg->epl;
    for (p = 0, n = 0; p < n_ci; p++)
        if (bv[p] >= 0 && (bm->wordend == 0 || bs[p] > bm->wordend + maxpscore)) {
            c_lc[n] = p; c_scr[n] = bs[p]; c_hist[n] = bv[p]; n++;
        }
    be_enter(k, n, c_lc, c_scr, c_hist, cf, th);
    c_lc[0] = BAD_S3CIPID; c_scr[0] = vh->bestscore[cf]; c_hist[0] = vh->bestvh[cf];
    be_enter(tstg->n_lextree + k, 1, c_lc, c_scr, c_hist, cf, th);
}


static int
tst_propagate_wd_lv2(void *srch, int32 frmno)
{
    srch_t *s = srch;
    srch_TST_graph_t *tstg = s->grh->graph_struct;
    histprune_t *hp = tstg->histprune;
    vithist_t *vh = tstg->vithist;
    int32 t, i;

    /* srch_TST_rescoring: word exits of every tree, in tree then active-list order */
    {
        s3a_frame_result_t r;
        beam_t *bm = s->beam;
        int32 k = 0;
        int32 wbeam_phone = (bm->ptranskip != 0 && (frmno % bm->ptranskip) == 0);
        double t0 = now_s();
        int32 rc = g_batch
            ? s3a_batch_step(g_batch, g_slot, g_featbuf, g_feat_idx, frmno, bm->hmm, bm->ptrans, bm->word,
                             wbeam_phone, hp->maxhmmpf, &r, g_exit_n, g_exit_wid, g_exit_scr, g_exit_hist,
                             g_ntree * g_max_node)
            : s3a_decoder_search(g_ls, g_sc, g_cs, frmno, bm->hmm, bm->ptrans, bm->word, wbeam_phone,
                                 hp->maxhmmpf, &r, g_exit_n, g_exit_wid, g_exit_scr, g_exit_hist,
                                 g_ntree * g_max_node);
        if (rc == S3A_EUNSUP)           /* a configuration the device path refuses: never a silent difference */
            E_FATAL("tst shim: %s\n", s3a_last_error());
        if (rc != S3A_OK) {
            E_ERROR("%s\n", s3a_last_error());
            return SRCH_FAILURE;
        }
        g_t_search += now_s() - t0;
        if (r.need_histprune)
            g_histframes++;     /* lextree_hmm_histbin + the bin scan ran on the device */
        /* what srch_TST_hmm_compute_lv2 leaves in beam_t / stat_t / histprune_t */
        bm->bestscore = r.best_hmm; bm->bestwordscore = r.best_word;
        bm->thres = r.thres; bm->phone_thres = r.phone_thres; bm->word_thres = r.word_thres;
        if (r.best_hmm > 0)
            E_ERROR("***ERROR*** Fr %d, best HMM score > 0 (%d); int32 wraparound?\n", frmno, r.best_hmm);
        s->stat->utt_hmm_eval += r.n_hmm;
        if (r.n_hmm / hp->hmm_hist_binsize > hp->hmm_hist_bins - 1) hp->hmm_hist[hp->hmm_hist_bins - 1]++;
        else hp->hmm_hist[r.n_hmm / hp->hmm_hist_binsize]++;
        /* what gmm_compute_lv2 leaves: senscale -> ascale[], evaluation counters */
        s->senscale = r.extra[6];
        s->ascale[g_ascale_idx] = r.extra[6];
        s->stat->utt_sen_eval += r.extra[1]; s->stat->utt_gau_eval += r.extra[2];
        s->stat->utt_cisen_eval += r.extra[3]; s->stat->utt_cigau_eval += r.extra[4];
        for (t = 0; t < g_ntree; t++)
            for (i = 0; i < g_exit_n[t]; i++, k++)
                vithist_rescore(vh, s->kbc, g_exit_wid[k], frmno, g_exit_scr[k], g_exit_hist[k],
                                g_flat[t]->type, -1);
    }
    {
        double t1 = now_s();
        vithist_prune(vh, kbcore_dict(s->kbc), frmno, hp->maxwpf, hp->maxhistpf,
                      s->beam->word_thres - s->beam->bestwordscore);
        tst_word_trans(s, frmno);
        g_t_word += now_s() - t1;
    }
    return SRCH_SUCCESS;
}

static int
tst_frame_windup(void *srch, int32 frmno)
{
    srch_t *s = srch;
    srch_TST_graph_t *tstg = s->grh->graph_struct;
    vithist_frame_windup(tstg->vithist, frmno, NULL, s->kbc);
    be_swap(frmno);
    return SRCH_SUCCESS;
}

/* ------------------------------------------------------------------ */
/* driver                                                              */
/* ------------------------------------------------------------------ */
#include <pthread.h>

static void
install_slots(srch_t *s)
{
    s->funcs->utt_begin = tst_begin;
    s->funcs->utt_end = tst_end;
    s->funcs->select_active_gmm = tst_select_active;
    s->funcs->hmm_compute_lv2 = tst_hmm_compute_lv2;
    s->funcs->propagate_graph_ph_lv2 = tst_propagate_ph_lv2;
    s->funcs->propagate_graph_wd_lv2 = tst_propagate_wd_lv2;
    s->funcs->frame_windup = tst_frame_windup;
    s->funcs->gmm_compute_lv1 = tst_gmm_lv1;
    s->funcs->gmm_compute_lv2 = tst_gmm_lv2;
}

/*
 * One decoder = one kb_t + one set of device objects + one HIP stream.  With
 * S3A_STREAMS=N (device build) N decoders run in N host threads of ONE process, each on
 * its contiguous shard of the control file (-ctloffset/-ctlcount, the reference's own
 * sharding device), sharing the GPU through their streams: utterances are independent, so
 * there is no cross-stream communication; the per-shard -hyp/-hypseg files are
 * concatenated in control-file order at the end.
 */
typedef struct {
    int id, n, argc;
    char **argv;
    int32 off, cnt;
    long frames, histframes;
    double t_utt, t_search, t_word;
} worker_t;

static pthread_mutex_t g_init_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_barrier_t g_start;
static double g_t_start;
static char g_hyp[2][4096];

static void *
worker_main(void *vp)
{
    worker_t *w = vp;
    kb_t kb;
    cmd_ln_t *config;
    char part[2][4200], so[32], sc[32];
    char **av = ckd_calloc(w->argc + 16, sizeof(char *));
    int ac = 0, i;

    for (i = 0; i < w->argc; i++) {
        if (i > 0 && (!strcmp(w->argv[i], "-hyp") || !strcmp(w->argv[i], "-hypseg")
                      || !strcmp(w->argv[i], "-ctloffset") || !strcmp(w->argv[i], "-ctlcount"))) { i++; continue; }
        av[ac++] = w->argv[i];
    }
    if (g_hyp[0][0]) { snprintf(part[0], sizeof part[0], "%s.part%03d", g_hyp[0], w->id); av[ac++] = "-hyp"; av[ac++] = part[0]; }
    if (g_hyp[1][0]) { snprintf(part[1], sizeof part[1], "%s.part%03d", g_hyp[1], w->id); av[ac++] = "-hypseg"; av[ac++] = part[1]; }
    snprintf(so, sizeof so, "%d", w->off); snprintf(sc, sizeof sc, "%d", w->cnt);
    av[ac++] = "-ctloffset"; av[ac++] = so; av[ac++] = "-ctlcount"; av[ac++] = sc;

    pthread_mutex_lock(&g_init_lock);           /* model / dictionary / LM loading is not re-entrant */
    g_worker_id = w->id;
    config = cmd_ln_parse_r(NULL, arg, ac, av, TRUE);
    kb_init(&kb, config);
    if (((srch_t *)kb.srch)->op_mode != 4)
        E_FATAL("tst shim: -op_mode 4 (fwdtree) only\n");
    backend_init(&kb, (srch_TST_graph_t *)((srch_t *)kb.srch)->grh->graph_struct);
    install_slots(kb.srch);
    if (g_batch && (g_slot = s3a_batch_attach(g_batch, g_ls, g_sc, g_cs)) < 0) die("batch attach");
    pthread_mutex_unlock(&g_init_lock);
    if (pthread_barrier_wait(&g_start) == PTHREAD_BARRIER_SERIAL_THREAD)
        g_t_start = now_s();        /* every decoder is loaded: the decode clock starts here */

    if (w->cnt > 0)
        kb.stat->tm = ctl_process(cmd_ln_str_r(config, "-ctl"), cmd_ln_str_r(config, "-ctl_lm"),
                                  cmd_ln_str_r(config, "-ctl_mllr"), w->off, w->cnt, utt_decode_adapt, &kb);
    if (kb.matchsegfp) fclose(kb.matchsegfp);
    if (kb.matchfp) fclose(kb.matchfp);
    w->frames = g_frames; w->histframes = g_histframes; w->t_utt = g_t_utt; w->t_search = g_t_search; w->t_word = g_t_word;
    return NULL;
}

static void
concat_parts(const char *dst, int n)
{
    FILE *out = fopen(dst, "w");
    char path[4200], buf[65536];
    int i;
    size_t k;
    if (!out) E_FATAL("cannot write %s\n", dst);
    for (i = 0; i < n; i++) {
        FILE *in;
        snprintf(path, sizeof path, "%s.part%03d", dst, i);
        if ((in = fopen(path, "r")) == NULL) continue;
        while ((k = fread(buf, 1, sizeof buf, in)) > 0) fwrite(buf, 1, k, out);
        fclose(in);
        remove(path);
    }
    fclose(out);
}

#include "s3amd_uttmode.h"

/* ------------------------------------------------------------------ */
/* S3A_LIVE=gpu|cpu: the reference's LIVE API (libAPI/s3_decode.c)     */
/* ------------------------------------------------------------------ */
/*
 * s3_decode_init / _begin_utt / _process (cepstra in blocks) / _end_utt / _hypothesis, as an application embedding
 * the decoder calls them.  s3_decode_init runs kb_init; an integrator installs the replacement slots right after it
 * (two calls), everything else is the reference's: feat_s2mfc2feat_live, utt_decode_block -> srch_utt_decode_blk ->
 * the per-frame slots.  S3A_LIVE=cpu leaves the table alone (the expected output of the test).
 */
static float32 **
read_cep(const char *path, int32 ceplen, int32 *n_out)
{
    FILE *fp = fopen(path, "rb");
    int32 n, i, swap = 0;
    long sz;
    float32 **c;
    if (!fp) E_FATAL("cannot read %s\n", path);
    fseek(fp, 0, SEEK_END); sz = ftell(fp); fseek(fp, 0, SEEK_SET);
    if (fread(&n, 4, 1, fp) != 1) E_FATAL("%s: empty\n", path);
    if ((long)n * 4 + 4 != sz) { SWAP_INT32(&n); swap = 1; }
    if ((long)n * 4 + 4 != sz || n % ceplen) E_FATAL("%s: not a cepstrum file\n", path);
    c = (float32 **)ckd_calloc_2d(n / ceplen + 1, ceplen, sizeof(float32));
    if (fread(c[0], 4, n, fp) != (size_t)n) E_FATAL("%s: short read\n", path);
    if (swap) for (i = 0; i < n; i++) SWAP_FLOAT32(&c[0][i]);
    fclose(fp);
    *n_out = n / ceplen;
    return c;
}

static int
live_mode_main(int argc, char *argv[], int use_gpu)
{
    s3_decode_t d;
    cmd_ln_t *config;
    const char *cepdir, *cepext;
    arg_t *defs;
    int n_live = 0, n_dec = 0, a, b2, n = 0;
    char line[4096], uttid[4096], path[8192];
    FILE *ctl;
    int32 ceplen, block = getenv("S3A_LIVE_BLOCK") ? atoi(getenv("S3A_LIVE_BLOCK")) : 37, n_utt = 0;

    /* the live decoder's own argument table + what sphinx3_decode's table has on top of it (-cepdir, -cepext, ...) */
    while (S3_DECODE_ARG_DEFS[n_live].name) n_live++;
    while (arg[n_dec].name) n_dec++;
    defs = (arg_t *)ckd_calloc(n_live + n_dec + 1, sizeof(arg_t));
    for (a = 0; a < n_live; a++) defs[n++] = S3_DECODE_ARG_DEFS[a];
    for (a = 0; a < n_dec; a++) {
        for (b2 = 0; b2 < n_live && strcmp(arg[a].name, S3_DECODE_ARG_DEFS[b2].name); b2++) ;
        if (b2 == n_live) defs[n++] = arg[a];
    }
    cmd_ln_appl_enter(argc, argv, "default.arg", defs);     /* (parts of the reference read the global table) */
    config = cmd_ln_get();
    if (!config) E_FATAL("bad arguments\n");
    memset(&d, 0, sizeof d);
    if (s3_decode_init(&d, config) != S3_DECODE_SUCCESS) E_FATAL("s3_decode_init failed\n");
    if (use_gpu) {
        srch_t *s = (srch_t *)d.kb.srch;
        if (s->op_mode != 4) E_FATAL("tst shim: -op_mode 4 (fwdtree) only\n");
        backend_init(&d.kb, (srch_TST_graph_t *)s->grh->graph_struct);
        install_slots(s);
    }
    cepdir = cmd_ln_str_r(config, "-cepdir"); cepext = cmd_ln_str_r(config, "-cepext");
    ceplen = feat_cepsize(kbcore_fcb(d.kbcore));
    if ((ctl = fopen(cmd_ln_str_r(config, "-ctl"), "r")) == NULL) E_FATAL("cannot read the control file\n");
    while (fgets(line, sizeof line, ctl)) {
        float32 **cep;
        int32 nfr, f;
        char *hyp = NULL, *id = NULL;
        hyp_t **segs = NULL;
        if (sscanf(line, "%4095s", uttid) != 1 || uttid[0] == '#') continue;
        snprintf(path, sizeof path, "%s/%s%s", cepdir ? cepdir : ".", uttid, cepext ? cepext : ".mfc");
        cep = read_cep(path, ceplen, &nfr);
        if (s3_decode_begin_utt(&d, uttid) != S3_DECODE_SUCCESS) E_FATAL("s3_decode_begin_utt failed\n");
        for (f = 0; f < nfr; f += block)
            if (s3_decode_process(&d, cep + f, nfr - f < block ? nfr - f : block) != S3_DECODE_SUCCESS)
                E_FATAL("s3_decode_process failed\n");
        s3_decode_end_utt(&d);
        if (s3_decode_hypothesis(&d, &id, &hyp, &segs) != S3_DECODE_SUCCESS) E_FATAL("s3_decode_hypothesis failed\n");
        printf("LIVE %s:", id ? id : uttid);
        for (; segs && *segs; segs++)
            printf(" %s(%d,%d,%d,%d)", dict_wordstr(kbcore_dict(d.kbcore), (*segs)->id), (*segs)->sf, (*segs)->ef, (*segs)->ascr, (*segs)->lscr);
        printf(" | %s\n", hyp ? hyp : "");
        ckd_free_2d((void **)cep);
        n_utt++;
    }
    fclose(ctl);
    E_INFO("tst shim live mode: %d utterances through s3_decode_process in blocks of %d frames, %s slots, %ld frames searched by the replacement backend\n",
           n_utt, block, use_gpu ? "replacement" : "reference", g_frames);
    if (d.kb.matchsegfp) { fclose(d.kb.matchsegfp); d.kb.matchsegfp = NULL; }
    if (d.kb.matchfp) { fclose(d.kb.matchfp); d.kb.matchfp = NULL; }
    fflush(stdout);
    return 0;
}

int
main(int argc, char *argv[])
{
    int n_streams = getenv("S3A_STREAMS") ? atoi(getenv("S3A_STREAMS")) : 1, i;
    cmd_ln_t *config;
    worker_t *w;
    pthread_t *th;
    int32 n_utt = 0, base, extra, off;
    long frames = 0, histframes = 0;
    double t_utt = 0, t_search = 0, t_word = 0, wall;
    char line[16384];
    FILE *fp;

    if (getenv("S3A_LIVE"))
        return live_mode_main(argc, argv, strcmp(getenv("S3A_LIVE"), "cpu") != 0);
    /* One process per GPU (torchrun / mpirun set RANK, WORLD_SIZE, LOCAL_RANK): rank r decodes the r-th contiguous
     * share of the control file (inside the caller's -ctloffset / -ctlcount) on GPU LOCAL_RANK and writes
     * <hyp>.part<r> / <hypseg>.part<r>; the parts, concatenated in rank order, are the one-process files.  With S3A_UTT the
     * ranks also exchange their hypothesis records over RCCL (s3a_gather_hyps) and rank 0 writes <hyp> / <hypseg>. */
    if (getenv("WORLD_SIZE") && atoi(getenv("WORLD_SIZE")) > 1 && getenv("RANK")) {
        const int W = atoi(getenv("WORLD_SIZE")), r = atoi(getenv("RANK"));
        const int lr = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : r;
        static char so[32], sc[32], part[2][4300];
        char **av = ckd_calloc(argc + 8, sizeof(char *));
        const char *ctl = NULL;
        int ac = 0, a, uoff = 0, ucnt = -1, n_lines = 0, base, extra, off, cnt;
        if (s3a_set_device(lr) != S3A_OK) E_FATAL("tst shim: rank %d cannot use GPU %d\n", r, lr);
        for (a = 0; a < argc; a++) {
            if (a > 0 && a + 1 < argc && !strcmp(argv[a], "-ctloffset")) { uoff = atoi(argv[++a]); continue; }
            if (a > 0 && a + 1 < argc && !strcmp(argv[a], "-ctlcount")) { ucnt = atoi(argv[++a]); continue; }
            if (a > 0 && a + 1 < argc && !strcmp(argv[a], "-ctl")) ctl = argv[a + 1];
            if (a > 0 && a + 1 < argc && (!strcmp(argv[a], "-hyp") || !strcmp(argv[a], "-hypseg"))) {
                const int k = !strcmp(argv[a], "-hyp") ? 0 : 1;
                snprintf(part[k], sizeof part[k], "%s.part%03d", argv[a + 1], r);
                snprintf(g_final[k], sizeof g_final[k], "%s", argv[a + 1]);
                av[ac++] = argv[a]; av[ac++] = part[k]; a++;
                continue;
            }
            av[ac++] = argv[a];
        }
        if (!ctl || (fp = fopen(ctl, "r")) == NULL) E_FATAL("tst shim: -ctl is required\n");
        while (fgets(line, sizeof line, fp)) if (line[0] != '\n' && line[0] != '#') n_lines++;
        fclose(fp);
        n_lines = n_lines > uoff ? n_lines - uoff : 0;
        if (ucnt >= 0 && ucnt < n_lines) n_lines = ucnt;
        base = n_lines / W; extra = n_lines % W;
        off = uoff + r * base + (r < extra ? r : extra); cnt = base + (r < extra ? 1 : 0);
        g_rank = r; g_world = W; g_rank_first = off - uoff; g_rank_total = n_lines;
        snprintf(so, sizeof so, "%d", off); snprintf(sc, sizeof sc, "%d", cnt);
        av[ac++] = "-ctloffset"; av[ac++] = so; av[ac++] = "-ctlcount"; av[ac++] = sc;
        argc = ac; argv = av;
        E_INFO("tst shim: rank %d of %d on GPU %d: control-file entries %d .. %d\n", r, W, lr, off, off + cnt - 1);
        if (cnt == 0 && !(getenv("S3A_UTT") && atoi(getenv("S3A_UTT")) > 0)) return 0;
    }
    cmd_ln_appl_enter(argc, argv, "default.arg", arg);      /* `arg`: the reference's own table */
    unlimit();
    config = cmd_ln_get();
    if (getenv("S3A_UTT") && atoi(getenv("S3A_UTT")) > 0)
        return utt_mode_main(argc, argv, atoi(getenv("S3A_UTT")));
    if (getenv("S3A_BATCH") && atoi(getenv("S3A_BATCH")) > 0) {
        if (n_streams < 1) n_streams = 1;
        g_n_groups = atoi(getenv("S3A_BATCH"));
        if (g_n_groups > MAX_GROUPS) g_n_groups = MAX_GROUPS;
        if (g_n_groups > n_streams) g_n_groups = n_streams;
        for (i = 0; i < g_n_groups; i++)
            if ((g_batches[i] = s3a_batch_create((n_streams + g_n_groups - 1) / g_n_groups)) == NULL) die("batch create");
    }
    if (!cmd_ln_str_r(config, "-ctl"))
        E_FATAL("-ctl is required\n");
    if (n_streams < 1) n_streams = 1;
    if (cmd_ln_str_r(config, "-hyp")) snprintf(g_hyp[0], sizeof g_hyp[0], "%s", cmd_ln_str_r(config, "-hyp"));
    if (cmd_ln_str_r(config, "-hypseg")) snprintf(g_hyp[1], sizeof g_hyp[1], "%s", cmd_ln_str_r(config, "-hypseg"));
    if ((fp = fopen(cmd_ln_str_r(config, "-ctl"), "r")) == NULL)
        E_FATAL("cannot read the control file\n");
    while (fgets(line, sizeof line, fp))
        if (line[0] != '\n' && line[0] != '#') n_utt++;     /* as ctl_process counts lines */
    fclose(fp);
    if (cmd_ln_int32_r(config, "-ctlcount") < n_utt) n_utt = cmd_ln_int32_r(config, "-ctlcount");

    w = ckd_calloc(n_streams, sizeof(*w));
    th = ckd_calloc(n_streams, sizeof(*th));
    base = n_utt / n_streams; extra = n_utt % n_streams;
    off = cmd_ln_int32_r(config, "-ctloffset");
    wall = now_s();
    pthread_barrier_init(&g_start, NULL, n_streams);
    for (i = 0; i < n_streams; i++) {
        w[i].id = i; w[i].n = n_streams; w[i].argc = argc; w[i].argv = argv;
        w[i].off = off; w[i].cnt = base + (i < extra ? 1 : 0);
        off += w[i].cnt;
        pthread_create(&th[i], NULL, worker_main, &w[i]);
    }
    for (i = 0; i < n_streams; i++) {
        pthread_join(th[i], NULL);
        frames += w[i].frames; histframes += w[i].histframes; t_utt += w[i].t_utt; t_search += w[i].t_search; t_word += w[i].t_word;
    }
    {
        double t_end = now_s();
        E_INFO("tst shim decode-only wall (all decoders loaded -> last one done): %.3f s\n", t_end - g_t_start);
        t_word += 0;
        wall = t_end - wall;
        g_t_start = t_end - g_t_start;
    }
    if (g_hyp[0][0]) concat_parts(g_hyp[0], n_streams);
    if (g_hyp[1][0]) concat_parts(g_hyp[1], n_streams);
    if (frames == 0)
        E_FATAL("tst shim: the replaced slots were never called\n");
    E_INFO("tst shim: %ld frames searched by the replacement backend in %d stream(s)\n", frames, n_streams);
    E_INFO("tst shim: histogram pruning (lextree_hmm_histbin) applied in %ld frames\n", histframes);
    if (g_n_groups) {
        int64_t st = 0, fr = 0, a, b;
        for (i = 0; i < g_n_groups; i++) { s3a_batch_stats(g_batches[i], &a, &b); st += a; fr += b; }
        E_INFO("tst shim: batched engine: %d group(s), %ld steps served %ld decoder-frames (mean batch %.1f)\n",
               g_n_groups, (long)st, (long)fr, st ? (double)fr / st : 0.0);
    }
    E_INFO("tst shim timing: %.1f us/frame inside utterances per stream (%.0f x real time per stream); of which "
           "frame_search (enqueue + the one sync) %.1f us, vithist_prune + word transitions %.1f us\n",
           1e6 * t_utt / frames, 0.01 * frames / (t_utt / n_streams) / n_streams, 1e6 * t_search / frames,
           1e6 * t_word / frames);
    E_INFO("tst shim throughput: %ld frames, decode-only %.3f s = %.0f x real time aggregate "
           "(%.3f s incl. loading %d decoders one after another)\n",
           frames, g_t_start, 0.01 * frames / g_t_start, wall, n_streams);
    return 0;
}
