/*
 * oracle/ref_s3odag_decode.c -- TEST INFRASTRUCTURE: pins oracle/s3o_dag.c.
 *
 * The UNMODIFIED reference decoder (sphinx3_decode, mode 4) whose `bestpath_impl` slot of srch_funcs_t
 * (sphinx3/include/srch.h:528-701; reference implementation srch_TST_bestpath_impl, srch_time_switch_tree.c:1391-1440)
 * is served by the restatement: the first pass, vithist_utt_end, the backtrace and the reference's own
 * vithist_dag_build (gen_dag) run unchanged; then, instead of dag_bypass_filler_nodes + dag_search on that dag_t,
 * s3o_dag_bestpath rebuilds the lattice from the PLAIN ARRAYS of the history table and searches it.  Run with
 * -bestpath 1, its -hyp / -hypseg must be byte-identical to the unmodified reference's; node and link counts of the
 * restated lattice are checked against the reference's dag_t on the way (a mismatch is fatal).
 * Built by oracle/Makefile into oracle/_ref/ (never committed); compiled against the reference where it lies.
 */
#define main sphinx3_decode_reference_main
#include "main_decode.c"        /* the reference's file, in place (for its arg table) */
#undef main
/* srch_TST_graph_t is private to srch_time_switch_tree.c:233-257: included in place, not copied */
#include "srch_time_switch_tree.c"

#include <string.h>
#include "byteorder.h"
#include "srch.h"
#include "lextree.h"
#include "vithist.h"
#include "dag.h"
#include "s3o.h"
#include "s3amd_flatten.h"

static wl_flat_t *g_w;
static int32 *g_basewid;
static long g_utts, g_words;

static glist_t
odag_bestpath_impl(void *srch, dag_t *dag)
{
    srch_t *s = (srch_t *)srch;
    srch_TST_graph_t *tstg = (srch_TST_graph_t *)s->grh->graph_struct;
    vithist_t *vh = tstg->vithist;
    kbcore_t *kbc = s->kbc;
    cmd_ln_t *config = kbcore_config(kbc);
    dict_t *dict = kbcore_dict(kbc);
    const int32 n = vh->n_entry;
    int32 *wid, *sf, *ef, *ascr, *lscr, *score, *out, stats[5] = { 0, 0, 0, 0, 0 }, i, nh = 0, *hw, *hs, nw, nnode = 0;
    uint8 *valid;
    float32 bestpathlw = cmd_ln_float32_r(config, "-bestpathlw");
    s3o_dagcfg_t c;
    s3o_lm3g_t olm;
    glist_t hyp, rhyp = NULL;
    gnode_t *gn;
    dagnode_t *d;

    if (!g_w) {
        g_w = flatten_lm(kbc);
        g_basewid = ckd_calloc(g_w->n_word + 1, 4);
        for (i = 0; i < g_w->n_word; i++) g_basewid[i] = dict_basewid(dict, i);
    }
    memset(&olm, 0, sizeof olm);
    olm.n_ug = g_w->n_ug; olm.n_bg = g_w->n_bg; olm.n_tg = g_w->n_tg;
    olm.ug_prob = g_w->ug_prob; olm.ug_bowt = g_w->ug_bowt; olm.ug_firstbg = g_w->ug_firstbg; olm.bg_wid = g_w->bg_wid;
    olm.bg_prob = g_w->bg_prob; olm.bg_bowt = g_w->bg_bowt; olm.bg_firsttg = g_w->bg_firsttg; olm.tg_wid = g_w->tg_wid;
    olm.tg_prob = g_w->tg_prob; olm.inclass = g_w->inclass;
    memset(&c, 0, sizeof c);
    c.n_word = g_w->n_word; c.basewid = g_basewid; c.is_filler = g_w->is_filler; c.lwid = g_w->lwid; c.fillpen = g_w->fillpen;
    c.startwid = g_w->startwid; c.finishwid = g_w->finishwid; c.start_lwid = g_w->start_lwid; c.finish_lwid = g_w->finish_lwid;
    c.wip = logs3(kbcore_logmath(kbc), kbcore_fillpen(kbc)->wip);
    c.lwf = bestpathlw ? (bestpathlw / cmd_ln_float32_r(config, "-lw")) : 1.0;
    c.min_endfr = cmd_ln_int32_r(config, "-min_endfr"); c.maxedge = cmd_ln_int32_r(config, "-maxedge");
    c.maxlmop = cmd_ln_int32_r(config, "-maxlmop"); c.maxlpf = cmd_ln_int32_r(config, "-maxlpf");

    wid = ckd_calloc(n + 1, 4); sf = ckd_calloc(n + 1, 4); ef = ckd_calloc(n + 1, 4); ascr = ckd_calloc(n + 1, 4);
    lscr = ckd_calloc(n + 1, 4); score = ckd_calloc(n + 1, 4); valid = ckd_calloc(n + 1, 1);
    for (i = 0; i < n; i++) {
        vithist_entry_t *ve = vithist_id2entry(vh, i);
        wid[i] = ve->wid; sf[i] = ve->sf; ef[i] = ve->ef; ascr[i] = ve->ascr; lscr[i] = ve->lscr;
        score[i] = ve->path.score; valid[i] = ve->valid ? 1 : 0;
    }
    /* the first pass's hypothesis (what srch_utt_end handed to gen_dag): vithist_backtrace again */
    hyp = vithist_backtrace(vh, s->exit_id, dict);
    hw = ckd_calloc(glist_count(hyp) + 1, 4); hs = ckd_calloc(glist_count(hyp) + 1, 4);
    for (gn = hyp; gn; gn = gnode_next(gn)) {
        srch_hyp_t *h = (srch_hyp_t *)gnode_ptr(gn);
        hw[nh] = h->id; hs[nh] = h->sf; nh++;
        ckd_free(h);
    }
    glist_free(hyp);
    out = ckd_calloc(5 * (n + 8), 4);
    nw = s3o_dag_bestpath(&c, &olm, n, wid, sf, ef, ascr, lscr, score, valid, vh->n_frm, s->exit_id, nh, hw, hs, out, n + 8, stats);
    for (d = dag->list; d; d = d->alloc_next) nnode++;
    if (getenv("S3O_DAGDUMP")) {
        /* fixture for the device tests: {tag, n, n x int32} records: the table the second pass read, what it produced */
        FILE *fp = fopen(getenv("S3O_DAGDUMP"), "ab");
        int32 hdr[8] = { n, vh->n_frm, s->exit_id, nh, nw, stats[0], stats[1], stats[3] }, tag, cnt;
        const int32 *arrs[8] = { wid, sf, ef, ascr, lscr, score, hw, hs };
        if (!fp) E_FATAL("cannot append to %s\n", getenv("S3O_DAGDUMP"));
        tag = 1; cnt = 8; fwrite(&tag, 4, 1, fp); fwrite(&cnt, 4, 1, fp); fwrite(hdr, 4, 8, fp);
        for (i = 0; i < 8; i++) { tag = 10 + i; cnt = i < 6 ? n : nh; fwrite(&tag, 4, 1, fp); fwrite(&cnt, 4, 1, fp); fwrite(arrs[i], 4, cnt, fp); }
        tag = 20; cnt = 1; fwrite(&tag, 4, 1, fp); fwrite(&cnt, 4, 1, fp); fwrite(&stats[4], 4, 1, fp);
        for (i = 0; i < 5; i++) { tag = 30 + i; cnt = nw > 0 ? nw : 0; fwrite(&tag, 4, 1, fp); fwrite(&cnt, 4, 1, fp); if (cnt) fwrite(out + (size_t)i * (n + 8), 4, cnt, fp); }
        fclose(fp);
    }
    if (stats[0] != nnode || stats[1] != dag->nlink)
        E_FATAL("dag oracle: lattice of %d nodes / %d links, the reference's dag_t has %d / %d\n", stats[0], stats[1], nnode, dag->nlink);
    E_INFO("dag oracle: %s: %d entries -> %d nodes, %d links (+%d bypass), %d LM operations, %d words\n", s->uttid, n, stats[0],
           stats[1], stats[3], stats[4], nw);
    if (nw > 0) {
        for (i = 0; i < nw; i++) {
            srch_hyp_t *h = (srch_hyp_t *)ckd_calloc(1, sizeof(srch_hyp_t));
            h->id = out[i]; h->word = dict_wordstr(dict, h->id); h->sf = out[(n + 8) + i]; h->ef = out[2 * (n + 8) + i];
            h->ascr = out[3 * (n + 8) + i]; h->lscr = out[4 * (n + 8) + i];
            rhyp = glist_add_ptr(rhyp, (void *)h);
        }
        rhyp = glist_reverse(rhyp);
        g_words += nw;
    }
    g_utts++;
    ckd_free(wid); ckd_free(sf); ckd_free(ef); ckd_free(ascr); ckd_free(lscr); ckd_free(score); ckd_free(valid);
    ckd_free(hw); ckd_free(hs); ckd_free(out);
    return rhyp;
}

int
main(int argc, char *argv[])
{
    kb_t kb;
    stat_t *st;
    cmd_ln_t *config;
    srch_t *s;

    print_appl_info(argv[0]);
    cmd_ln_appl_enter(argc, argv, "default.arg", arg);
    unlimit();
    config = cmd_ln_get();
    kb_init(&kb, config);
    st = kb.stat;
    s = (srch_t *)kb.srch;
    if (s->op_mode != 4 || !cmd_ln_boolean_r(config, "-bestpath"))
        E_FATAL("dag oracle: run with -op_mode 4 -bestpath 1\n");
    s->funcs->bestpath_impl = odag_bestpath_impl;
    if (!cmd_ln_str_r(config, "-ctl"))
        E_FATAL("-ctl is required\n");
    st->tm = ctl_process(cmd_ln_str_r(config, "-ctl"), cmd_ln_str_r(config, "-ctl_lm"), cmd_ln_str_r(config, "-ctl_mllr"),
                         cmd_ln_int32_r(config, "-ctloffset"), cmd_ln_int32_r(config, "-ctlcount"), utt_decode, &kb);
    if (kb.matchsegfp) fclose(kb.matchsegfp);
    if (kb.matchfp) fclose(kb.matchfp);
    stat_report_corpus(kb.stat);
    E_INFO("dag oracle: second pass of %ld utterances served by oracle/s3o_dag.c (%ld words)\n", g_utts, g_words);
    if (g_utts == 0)
        E_FATAL("dag oracle: the replaced slot was never called\n");
    return 0;
}
