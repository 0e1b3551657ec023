/*
 * s3amd_uttmode.h -- the whole-utterance driver of integration/sphinx3/s3amd_tst.c (S3A_UTT=lanes): the `decode` slot of
 * srch_funcs_t served by s3a_uttdec_* -- queueing of control-file entries, engines on host threads, the reference finishing
 * every utterance in control-file order (or the device-side hypotheses / second pass), one process per GPU with the
 * end-of-batch exchange over RCCL.  Included by s3amd_tst.c (it uses that file's backend objects); not a translation unit of its own.
 */
/* ------------------------------------------------------------------ */
/* S3A_UTT=L: whole utterances on the device, L at a time              */
/* ------------------------------------------------------------------ */
/*
 * One kb_t (the models, dictionary and LM are loaded ONCE), one s3a_uttdec_t with L lanes.  The
 * control file is walked by the reference's own ctl_process; its per-utterance callback computes the
 * features exactly as utt_decode does (libAPI/utt.c:185-258) and QUEUES the utterance; every L
 * utterances the queue is decoded in one s3a_uttdec_decode call -- srch_TST_begin to the last
 * frame_windup of every utterance on the device, no host work per frame -- and each utterance is then
 * finished in control-file order by the reference's own code: srch_utt_begin, the history table
 * read back into its vithist_t (vithist_fill), srch_utt_end (vithist_utt_end, backtrace, -hyp /
 * -hypseg / lattices / statistics).
 */
typedef struct { char *uttid, *uttfile; float32 *feat; int32 nfr; int32 on_dev; } uq_t;   /* on_dev: feat is device memory (-adcin) */
#define UTT_MAX_ENGINES 8
static s3a_mgau_model_t *g_gms[UTT_MAX_ENGINES];       /* the engines' device models (MLLR: all follow the host model) */
static s3a_uttdec_t *g_ud, *g_uds[UTT_MAX_ENGINES];    /* g_ud = g_uds[0]; S3A_UTT_ENGINES engines of g_lpe lanes each */
static int32 g_n_eng = 1, g_lpe;
static s3a_lm3g_t *g_lm3g;
static s3a_dag_cfg_t g_dag_cfg;         /* what the second pass was enabled with (its word tables stay alive: the library's N-best reads them) */
static uq_t *g_uq;
static int32 g_uq_n, g_uq_cap;
static int g_adcin, g_cmn_current, g_varnorm, g_agc_max;     /* -adcin: features made on the device from the samples */
static s3a_fe_t *g_fe;
static void uq_free(uq_t *q) { ckd_free(q->uttid); ckd_free(q->uttfile); if (q->on_dev) (void)s3a_dev_free(q->feat); else ckd_free(q->feat); }
static kb_t *g_ukb;
static double g_t_dev, g_t_fin, g_t_feat;
static long g_utt_frames, g_max_cand, g_max_new, g_tie_frames, g_frames_lane0;
static long long g_wl_ticks[16];

static int
utt_begin_slot(void *srch)              /* srch_TST_begin without the device work (the lanes did it) */
{
    srch_t *s = srch;
    srch_TST_graph_t *tstg = s->grh->graph_struct;
    vithist_utt_reset(tstg->vithist);
    histprune_zero_histbin(tstg->histprune);
    vithist_utt_begin(tstg->vithist, s->kbc);
    tstg->n_lextrans = 1;
    return SRCH_SUCCESS;
}

static int
utt_end_slot(void *srch)                /* srch_TST_end, :514-560 */
{
    srch_t *s = srch;
    srch_TST_graph_t *tstg = s->grh->graph_struct;
    s->exit_id = vithist_utt_end(tstg->vithist, s->kbc);
    s->stat->utt_wd_exit = vithist_n_entry(tstg->vithist);
    histprune_showhistbin(tstg->histprune, s->stat->nfr, s->uttid);
    lm_cache_stats_dump(kbcore_lm(s->kbc));
    lm_cache_reset(kbcore_lm(s->kbc));
    return (s->exit_id >= 0) ? SRCH_SUCCESS : SRCH_FAILURE;
}

/* -bestpath 1 in utt mode: the SECOND PASS ran on the device behind the utterance's last frame (s3a_uttdec_enable_bestpath:
 * vithist_utt_end, lattice, filler bypass, best path, backtrace); the two slots srch_utt_end calls for it
 * (srch.c:519-530 gen_dag, :609-640 bestpath_impl) hand the result over.  S3A_UTT_HOSTDAG=1: the reference's own
 * vithist_dag_build / dag_search on the table the device produced (the comparison run). */
static int g_dev_dag;
static long g_dag_utts, g_failed_utts;
/* one process per GPU: the rank's hypotheses as (header, words) records for the end-of-batch exchange (s3a_gather_hyps) */
static int g_rank = 0, g_world = 1, g_rank_first = 0, g_rank_total = 0, g_gather = 0;
static char g_final[2][4300];
static s3a_hyp_header_t *g_rec_hdr;
static s3a_hyp_word_t *g_rec_words;
static int32 g_rec_n, g_rec_cap, g_rec_nw, g_rec_wcap;
static void
rec_add(int32 z)
{
    s3a_uttdec_t *ud = g_uds[z / g_lpe];
    s3a_hyp_header_t h;
    int32 need = 0, rc;
    if (g_rec_n == g_rec_cap) { g_rec_cap = g_rec_cap ? 2 * g_rec_cap : 1024; g_rec_hdr = ckd_realloc(g_rec_hdr, (size_t)g_rec_cap * sizeof(*g_rec_hdr)); }
    for (;;) {
        if (g_rec_nw + need > g_rec_wcap) { g_rec_wcap = 2 * (g_rec_nw + need) + 4096; g_rec_words = ckd_realloc(g_rec_words, (size_t)g_rec_wcap * sizeof(*g_rec_words)); }
        rc = g_dev_dag ? s3a_uttdec_bestpath_hyp(ud, z % g_lpe, g_uq[z].uttid, g_rank_first + g_rec_n, &h, g_rec_words + g_rec_nw, g_rec_wcap - g_rec_nw)
                       : s3a_uttdec_hyp_var(ud, z % g_lpe, g_uq[z].uttid, g_rank_first + g_rec_n, &h, g_rec_words + g_rec_nw, g_rec_wcap - g_rec_nw);
        if (rc != S3A_OK) die("hypothesis record");
        if (h.status != -3) break;
        need = h.n_words;
    }
    g_rec_hdr[g_rec_n++] = h;
    if (h.status == 0) g_rec_nw += h.n_words;
}
static __thread int32 g_cur_lane;
static int g_want_lattice;      /* -outlatdir / -nbestdir: the reference's writers and its N-best search get a dag_t made from the device's lattice */
static char **g_wordstr;
/* gen_dag (srch.c:519-530; srch_TST_gen_dag = vithist_dag_build, srch_time_switch_tree.c:1381-1389): the lattice was built on
 * the device behind the utterance's last frame.  Without -outlatdir / -nbestdir srch_utt_end only tests the pointer (an empty
 * dag_t will do); with them the device's lattice is poured into a dag_t -- nodes and links made in the order vithist_dag_build
 * makes them (vithist.c:1243-1281), so that every list reads as the reference's -- and dag_write_htk, nbest_search and their
 * like run on it as they are. */
static dag_t *
utt_gen_dag_slot(void *srch, glist_t hyp)
{
    srch_t *s = srch;
    cmd_ln_t *config = kbcore_config(s->kbc);
    dag_t *dag = ckd_calloc(1, sizeof(*dag));
    s3a_uttdec_t *ud = g_uds[g_cur_lane / g_lpe];
    s3a_lat_info_t info;
    s3a_lat_node_t *nodes;
    s3a_lat_link_t *links;
    dagnode_t **dn;
    int32 i, j, k, *first;
    (void)hyp;
    dag_init(dag, config, kbcore_logmath(s->kbc));
    if (!g_want_lattice) return dag;
    if (s3a_uttdec_lattice(ud, g_cur_lane % g_lpe, &info, NULL, 0, NULL, 0) != S3A_OK) {
        E_ERROR("tst shim: no lattice from the device for %s: %s\n", s->uttid, s3a_last_error());
        dag_destroy(dag);
        return NULL;
    }
    nodes = ckd_calloc(info.n_nodes + 1, sizeof(*nodes));
    links = ckd_calloc(info.n_links + 1, sizeof(*links));
    if (s3a_uttdec_lattice(ud, g_cur_lane % g_lpe, &info, nodes, info.n_nodes, links, info.n_links) != S3A_OK) die("s3a_uttdec_lattice");
    dn = ckd_calloc(info.n_nodes + 1, sizeof(*dn));
    for (k = info.n_nodes - 1; k >= 0; k--) {           /* dag->list is built by head insertion: its first node is made last */
        dagnode_t *d = listelem_malloc(dag->node_alloc);
        d->wid = nodes[k].wid; d->node_ascr = nodes[k].ascr; d->node_lscr = nodes[k].lscr;
        d->sf = nodes[k].sf; d->fef = nodes[k].fef; d->lef = nodes[k].lef;
        d->seqid = info.n_nodes - 1 - k; d->hook = NULL; d->predlist = NULL; d->succlist = NULL; d->reachable = 0;
        d->alloc_next = dag->list; dag->list = d;
        dn[k] = d;
    }
    first = ckd_calloc(info.n_nodes + 2, sizeof(*first));
    for (i = 0; i < info.n_links; i++) first[links[i].from + 1]++;
    for (k = 0; k < info.n_nodes; k++) first[k + 1] += first[k];
    for (k = info.n_nodes - 1; k >= 0; k--)             /* sources in the order they were made; dag_link prepends: a list's last link first */
        for (j = first[k + 1] - 1; j >= first[k]; j--)
            dag_link(dag, dn[k], dn[links[j].to], links[j].ascr, links[j].lscr, links[j].ef, NULL);
    dag->root = dn[info.initial]; dag->end = dn[info.final];
    dag->entry.node = dag->root; dag->entry.ascr = 0; dag->entry.next = NULL; dag->entry.pscr_valid = 0; dag->entry.bypass = NULL;
    dag->final.node = dag->end; dag->final.ascr = info.final_ascr; dag->final.next = NULL; dag->final.pscr_valid = 0; dag->final.bypass = NULL;
    dag->filler_removed = 0; dag->fudged = 0; dag->nfrm = info.n_frames;
    dag->maxedge = cmd_ln_int32_r(config, "-maxedge");
    dag->maxlmop = cmd_ln_int32_r(config, "-maxlmop");
    k = cmd_ln_int32_r(config, "-maxlpf") * dag->nfrm;
    if (k > 0 && dag->maxlmop > k) dag->maxlmop = k;
    dag->lmop = 0;
    if (getenv("S3A_LAT_LIBHTK") && cmd_ln_str_r(config, "-outlatdir")) {
        /* (test hook) dag_write_htk by the library's own formatter beside the reference's file: <file>.libhtk */
        dict_t *dict = kbcore_dict(s->kbc);
        lm_t *lm = kbcore_lm(s->kbc);
        logmath_t *lmath = kbcore_logmath(s->kbc);
        s3a_htk_opts_t ho;
        char str[2048], *hdr = NULL, *buf;
        size_t hl = 0;
        int64_t need;
        int32 nw = dict_size(dict), *base = ckd_calloc(nw + 1, 4), *nalt = ckd_calloc(nw + 1, 4);
        FILE *hf = open_memstream(&hdr, &hl), *fp;
        dag_write_header(hf, config);
        fclose(hf);
        for (i = 0; i < nw; i++) { base[i] = dict_basewid(dict, i); nalt[base[i]]++; }
        memset(&ho, 0, sizeof ho);
        ho.uttid = s->uttid; ho.lmname = lm ? lm->name : NULL; ho.have_lm = lm != NULL; ho.lm_wip = lm ? lm->wip : 0; ho.lm_lw = lm ? lm->lw : 1.0f;
        ho.frate = cmd_ln_exists_r(config, "-frate") ? cmd_ln_int32_r(config, "-frate") : 0;
        ho.log_shift = logmath_get_shift(lmath); ho.log_of_base = log(logmath_get_base(lmath));
        ho.opt_lw = cmd_ln_float32_r(config, "-lw"); ho.opt_wip = cmd_ln_float32_r(config, "-wip");
        ho.basewid = base; ho.n_alt = nalt;
        need = s3a_lattice_format_htk(hdr, &ho, &info, nodes, links, (const char *const *)g_wordstr, NULL, 0);
        buf = ckd_calloc(need + 1, 1);
        s3a_lattice_format_htk(hdr, &ho, &info, nodes, links, (const char *const *)g_wordstr, buf, need + 1);
        ctl_outfile(str, cmd_ln_str_r(config, "-outlatdir"), "libhtk", (s->uttfile ? s->uttfile : s->uttid), s->uttid, cmd_ln_boolean_r(config, "-build_outdirs"));
        if ((fp = fopen(str, "w")) != NULL) { fwrite(buf, 1, need, fp); fclose(fp); }
        ckd_free(buf); ckd_free(base); ckd_free(nalt); free(hdr);
    }
    ckd_free(first); ckd_free(dn); ckd_free(nodes); ckd_free(links);
    return dag;
}
/* dag_dump (srch.c:586-590: the Sphinx-3 format's "custom implementation" slot; NULL in srch_TST_funcs, so that srch_utt_end
 * calls dag_write, dag.c:731-790): the same file by the library's formatter, straight from the device's lattice */
static int
utt_dag_dump_slot(void *srch, dag_t *dag)
{
    srch_t *s = srch;
    cmd_ln_t *config = kbcore_config(s->kbc);
    s3a_uttdec_t *ud = g_uds[g_cur_lane / g_lpe];
    s3a_lat_info_t info;
    s3a_lat_node_t *nodes;
    s3a_lat_link_t *links;
    char str[2048], *hdr = NULL, *buf;
    size_t hl = 0;
    int64_t need;
    int32 ispipe;
    FILE *hf, *fp;
    (void)dag;
    if (s3a_uttdec_lattice(ud, g_cur_lane % g_lpe, &info, NULL, 0, NULL, 0) != S3A_OK) return SRCH_FAILURE;
    nodes = ckd_calloc(info.n_nodes + 1, sizeof(*nodes));
    links = ckd_calloc(info.n_links + 1, sizeof(*links));
    if (s3a_uttdec_lattice(ud, g_cur_lane % g_lpe, &info, nodes, info.n_nodes, links, info.n_links) != S3A_OK) die("s3a_uttdec_lattice");
    hf = open_memstream(&hdr, &hl);
    dag_write_header(hf, config);
    fclose(hf);
    need = s3a_lattice_format_s3(hdr, &info, nodes, links, (const char *const *)g_wordstr, NULL, 0);
    buf = ckd_calloc(need + 1, 1);
    s3a_lattice_format_s3(hdr, &info, nodes, links, (const char *const *)g_wordstr, buf, need + 1);
    ctl_outfile(str, cmd_ln_str_r(config, "-outlatdir"), cmd_ln_str_r(config, "-latext"), (s->uttfile ? s->uttfile : s->uttid), s->uttid,
                cmd_ln_boolean_r(config, "-build_outdirs"));
    E_INFO("Writing lattice file in Sphinx III format: %s\n", str);
    if ((fp = fopen_comp(str, "w", &ispipe)) == NULL) { E_ERROR("fopen_comp (%s,w) failed\n", str); ckd_free(buf); ckd_free(nodes); ckd_free(links); free(hdr); return SRCH_FAILURE; }
    fwrite(buf, 1, need, fp);
    fclose_comp(fp, ispipe);
    ckd_free(buf); ckd_free(nodes); ckd_free(links); free(hdr);
    return SRCH_SUCCESS;
}
/* nbest_impl (srch.c:604-606; srch_TST_nbest_impl, srch_time_switch_tree.c:1442-1492 -> nbest_search, astar.c:656-716): the list by the
 * library's own search (s3a_lattice_nbest: unreachable nodes, filler bypass, heuristic scores, A*) on the device's lattice; the file is
 * written here as nbest_search writes it -- and not left behind when the search found nothing.  S3A_REF_NBEST=1 keeps the reference's
 * own search on the dag_t that utt_gen_dag_slot pours (A/B in tests/test_gpu_dag.py). */
static glist_t
utt_nbest_slot(void *srch, dag_t *dag)
{
    srch_t *s = srch;
    cmd_ln_t *config = kbcore_config(s->kbc);
    lm_t *lm = kbcore_lm(s->kbc);
    s3a_uttdec_t *ud = g_uds[g_cur_lane / g_lpe];
    s3a_lat_info_t info;
    s3a_lat_node_t *nodes;
    s3a_lat_link_t *links;
    s3a_nbest_opts_t o;
    s3a_nbest_t *nb;
    const char *text = NULL;
    int64_t len = 0;
    int32 n_hyp = 0, cnt[4] = { 0, 0, 0, 0 }, ispipe, st;
    char str[2048];
    FILE *fp;
    (void)dag;
    if (!cmd_ln_str_r(config, "-nbestdir")) return NULL;
    if (s3a_uttdec_lattice(ud, g_cur_lane % g_lpe, &info, NULL, 0, NULL, 0) != S3A_OK) { E_ERROR("tst shim: no lattice from the device for %s: %s\n", s->uttid, s3a_last_error()); return NULL; }
    nodes = ckd_calloc(info.n_nodes + 1, sizeof(*nodes));
    links = ckd_calloc(info.n_links + 1, sizeof(*links));
    if (s3a_uttdec_lattice(ud, g_cur_lane % g_lpe, &info, nodes, info.n_nodes, links, info.n_links) != S3A_OK) die("s3a_uttdec_lattice");
    memset(&o, 0, sizeof o);
    o.uttid = s->uttid; o.beam = cmd_ln_float64_r(config, "-beam"); o.beam_logs3 = logs3(kbcore_logmath(s->kbc), o.beam);
    o.nbest = cmd_ln_int32_r(config, "-nbest"); o.maxppath = cmd_ln_int32_r(config, "-maxppath");
    o.lm_wip = lm->wip; o.lm_lw = lm->lw; o.logbase = cmd_ln_float32_r(config, "-logbase"); o.lw = cmd_ln_float32_r(config, "-lw"); o.wip = cmd_ln_float32_r(config, "-wip");
    ctl_outfile(str, cmd_ln_str_r(config, "-nbestdir"), cmd_ln_str_r(config, "-nbestext"), (s->uttfile ? s->uttfile : s->uttid), s->uttid,
                cmd_ln_boolean_r(config, "-build_outdirs"));
    nb = s3a_lattice_nbest(g_lm3g, &g_dag_cfg, &o, &info, nodes, links, (const char *const *)g_wordstr);
    if (nb == NULL) die("s3a_lattice_nbest");
    st = s3a_nbest_result(nb, &text, &len, &n_hyp, cnt);
    if (st != S3A_OK) E_ERROR("maxedge limit (%d) exceeded\n", g_dag_cfg.maxedge);        /* (srch_TST_nbest_impl: no list then) */
    else if (n_hyp <= 0) E_ERROR("%s: A* search failed\n", s->uttid);                       /* (nbest_search unlinks the file) */
    else {
        E_INFO("Writing N-Best list to %s\n", str);
        if ((fp = fopen_comp(str, "w", &ispipe)) == NULL) E_ERROR("fopen_comp (%s,w) failed\n", str);
        else { fwrite(text, 1, (size_t)len, fp); fclose_comp(fp, ispipe); }
    }
    E_INFO("N-Best search(%s) in the library: %5d frm %4d hyp %6d pop %6d exp %8d pp, %d bypass links\n", s->uttid, info.n_frames, n_hyp, cnt[0], cnt[1], cnt[2], cnt[3]);
    s3a_nbest_free(nb);
    ckd_free(nodes); ckd_free(links);
    return NULL;
}
static glist_t
utt_bestpath_slot(void *srch, dag_t *dag)
{
    srch_t *s = srch;
    dict_t *dict = kbcore_dict(s->kbc);
    s3a_dag_result_t r;
    glist_t rhyp = NULL;
    int32 i;
    (void)dag;
    if (s3a_uttdec_bestpath_result(g_uds[g_cur_lane / g_lpe], g_cur_lane % g_lpe, &r) != S3A_OK) die("bestpath result");
    if (r.status == 2) { E_ERROR("Bestpath search failed for %s\n", s->uttid); return NULL; }
    if (r.status == 3 || r.status == 5) {       /* this utterance only: srch_utt_end logs "Bestpath search failed." and goes on */
        E_ERROR("the device's second pass gave up on %s (status %d: link capacity / -maxedge in the filler bypass / positive bypass edge)\n", s->uttid, r.status);
        g_failed_utts++;
        return NULL;
    }
    if (r.status != 0) E_FATAL("tst shim: the device's second pass stopped with status %d: %s\n", r.status, s3a_last_error());
    E_INFO("tst shim: second pass on the device: %s: %d entries -> %d nodes, %d links (+%d bypass), %d LM operations, %d words\n",
           s->uttid, r.n_entry, r.n_node, r.n_link, r.n_bypass, r.lmop, r.n_words);
    for (i = 0; i < r.n_words; i++) {
        srch_hyp_t *h = (srch_hyp_t *)ckd_calloc(1, sizeof(srch_hyp_t));
        h->id = r.wid[i]; h->word = dict_wordstr(dict, h->id); h->sf = r.sf[i]; h->ef = r.ef[i]; h->ascr = r.ascr[i]; h->lscr = r.lscr[i];
        rhyp = glist_add_ptr(rhyp, (void *)h);
    }
    g_dag_utts++;
    return glist_reverse(rhyp);
}

static void
utt_finish(kb_t *kb, int32 z)
{
    srch_t *s = kb->srch;
    srch_TST_graph_t *tstg = s->grh->graph_struct;
    histprune_t *hp = tstg->histprune;
    stat_t *st = kb->stat;
    uq_t *q = &g_uq[z];
    s3a_utt_result_t r;
    int32 f;

    if (s3a_uttdec_result(g_uds[z / g_lpe], z % g_lpe, &r) != S3A_OK) die("uttdec result");
    if (r.err) {
        E_ERROR("tst shim: utterance %s stopped on the device (error bits 0x%x: a capacity of its lane -- S3A_UTT_VHCAP / S3A_UTT_CANDCAP); no hypothesis written\n",
                q->uttid, r.err);
        g_failed_utts++;
        if (g_gather) rec_add(z);               /* (status -1: the exchange carries it, rank 0 writes no line) */
        uq_free(q);
        return;
    }
    if (z == 0 && getenv("S3A_UTT_TICKS")) {
        long long tk[16];
        int i;
        if (s3a_uttdec_wl_ticks(g_ud, 0, tk) == S3A_OK)
            for (i = 0; i < 9; i++) g_wl_ticks[i] += tk[i];
        g_frames_lane0 += r.n_frames;
    }
    kb_set_uttid(q->uttid, q->uttfile, kb);
    s->uttid = kb->uttid;
    s->uttfile = kb->uttfile;
    E_INFO("Processing: %s\n", q->uttid);
    utt_begin(kb);                              /* srch_utt_begin -> utt_begin_slot */
    while (r.n_frames >= s->ascale_sz) {        /* srch.c:697-704 */
        s->ascale = (int32 *)ckd_realloc(s->ascale, (s->ascale_sz + DFLT_UTT_SIZE) * sizeof(int32));
        s->ascale_sz += DFLT_UTT_SIZE;
    }
    for (f = 0; f < r.n_frames; f++) {
        const int32 *fs = r.frame_stat + 8 * f;
        s->ascale[f] = fs[0];                   /* srch.c:752 */
        st->utt_hmm_eval += fs[1]; st->utt_sen_eval += fs[2]; st->utt_gau_eval += fs[3];
        st->utt_cisen_eval += fs[4]; st->utt_cigau_eval += fs[5];
        if (fs[1] / hp->hmm_hist_binsize > hp->hmm_hist_bins - 1) hp->hmm_hist[hp->hmm_hist_bins - 1]++;
        else hp->hmm_hist[fs[1] / hp->hmm_hist_binsize]++;
        if (fs[6]) g_histframes++;
    }
    st->nfr += r.n_frames;                      /* srch.c:839 */
    g_frames += r.n_frames;
    if (r.max_cand > g_max_cand) g_max_cand = r.max_cand;
    if (r.max_new > g_max_new) g_max_new = r.max_new;
    g_tie_frames += r.n_tie_frames;
    vithist_fill(tstg->vithist, r.n_entry, r.n_frm, r.score, r.pred, r.lw0, r.lw1, r.wid, r.sf, r.ef, r.ascr, r.lscr,
                 r.type, r.frame_start, r.bestscore, r.bestvh, kbcore_lm(kb->kbcore));
    g_cur_lane = z;
    if (g_gather) rec_add(z);
    utt_end(kb);                                /* srch_utt_end -> utt_end_slot, gen_hyp, match_write ... */
    st->tot_fr += st->nfr;
    uq_free(q);
}

typedef struct { int32 e, n, veclen, rc, state, queue, dev; const float **feat; const int32 *nfr; pthread_t th; } eng_job_t;
static eng_job_t g_job[UTT_MAX_ENGINES];        /* state: 0 idle, 1 posted, 2 done */
static pthread_mutex_t g_eng_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_eng_cv = PTHREAD_COND_INITIALIZER;
static int g_eng_started;

/* (features in host memory: rows of veclen floats; -adcin: made on the device, rows of the scorer's padded stride) */
static int32 g_feat_dim;
static int32
eng_decode(const eng_job_t *j)
{
    s3a_uttdec_t *ud = g_uds[j->e];
    /* (device features: rows of the OUTPUT dimension -- -ldadim may have cut it -- rounded up to four floats) */
    const int32 stride = j->dev ? 4 * ((g_feat_dim + 3) / 4) : j->veclen;
    if (j->queue) return j->dev ? s3a_uttdec_decode_queue_dev(ud, j->n, j->feat, j->nfr, stride) : s3a_uttdec_decode_queue(ud, j->n, j->feat, j->nfr, stride);
    return j->dev ? s3a_uttdec_decode_dev(ud, j->n, j->feat, j->nfr, stride) : s3a_uttdec_decode(ud, j->n, j->feat, j->nfr, stride);
}

/* one persistent host thread per engine (a thread's first HIP call is expensive: not one per batch) */
static void *
eng_main(void *vp)
{
    eng_job_t *j = vp;
    (void)s3a_dev_sync();               /* (the thread's HIP state, before the decode clock starts) */
    pthread_mutex_lock(&g_eng_lock);
    j->state = 0;
    pthread_cond_broadcast(&g_eng_cv);
    pthread_mutex_unlock(&g_eng_lock);
    for (;;) {
        pthread_mutex_lock(&g_eng_lock);
        while (j->state != 1) pthread_cond_wait(&g_eng_cv, &g_eng_lock);
        pthread_mutex_unlock(&g_eng_lock);
        j->rc = eng_decode(j);
        pthread_mutex_lock(&g_eng_lock);
        j->state = 2;
        pthread_cond_broadcast(&g_eng_cv);
        pthread_mutex_unlock(&g_eng_lock);
    }
    return NULL;
}

/* S3A_UTT_QUEUE=n: lane refill.  n (>= the lanes) control-file entries are collected, dealt to the engines longest first,
 * and every engine decodes its share as ONE queue (s3a_uttdec_decode_queue): a lane takes the next utterance when its own
 * has ended, so a ragged corpus keeps every lane busy (in lock step a batch lasts as long as its longest utterance).
 * The lanes' history tables are reused by their next utterances, so the reference's srch_utt_end cannot finish these
 * utterances: the -hyp / -hypseg lines are written here, in control-file order, from the device's hypothesis records
 * (s3a_uttdec_queue_hyp = vithist_utt_end + backtrace on the device; s3a_hyp_format_var = match_write / matchseg_write);
 * no per-utterance statistics, lattices or second pass in this mode. */
static int g_queue;
static wl_flat_t *g_wflat;
static const int32 *g_sort_nfr;
static int
cmp_longest_first(const void *a, const void *b)
{
    const int32 x = *(const int32 *)a, y = *(const int32 *)b;
    if (g_sort_nfr[x] != g_sort_nfr[y]) return g_sort_nfr[y] - g_sort_nfr[x];
    return x - y;
}

/* -outlatdir / -nbestdir out of a queue (round 6): the utterance's lattice was kept behind its second pass (s3a_uttdec_queue_keep_lattices);
 * the files by the library's formatters and its N-best search, named and written as srch_utt_end would (srch.c:573-606) */
static int g_queue_lat;
static void
queue_write_lattice_files(kb_t *kb, s3a_uttdec_t *ud, int32 q, const uq_t *uq)
{
    kbcore_t *kbc = kb->kbcore;
    cmd_ln_t *config = kbcore_config(kbc);
    lm_t *lm = kbcore_lm(kbc);
    s3a_lat_info_t info;
    s3a_lat_node_t *nodes;
    s3a_lat_link_t *links;
    char str[2048];
    int32 ispipe;
    FILE *fp;
    if (s3a_uttdec_queue_lattice(ud, q, &info, NULL, 0, NULL, 0) != S3A_OK) { E_ERROR("tst shim: no lattice for %s: %s\n", uq->uttid, s3a_last_error()); return; }
    nodes = ckd_calloc(info.n_nodes + 1, sizeof(*nodes));
    links = ckd_calloc(info.n_links + 1, sizeof(*links));
    if (s3a_uttdec_queue_lattice(ud, q, &info, nodes, info.n_nodes, links, info.n_links) != S3A_OK) die("s3a_uttdec_queue_lattice");
    if (cmd_ln_str_r(config, "-outlatdir")) {
        char *hdr = NULL, *buf;
        size_t hl = 0;
        int64_t need;
        FILE *hf = open_memstream(&hdr, &hl);
        const int htk = strcmp(cmd_ln_str_r(config, "-outlatfmt"), "htk") == 0;
        dag_write_header(hf, config);
        fclose(hf);
        if (htk) {
            dict_t *dict = kbcore_dict(kbc);
            logmath_t *lmath = kbcore_logmath(kbc);
            s3a_htk_opts_t ho;
            int32 nw = dict_size(dict), *base = ckd_calloc(nw + 1, 4), *nalt = ckd_calloc(nw + 1, 4), i;
            for (i = 0; i < nw; i++) { base[i] = dict_basewid(dict, i); nalt[base[i]]++; }
            memset(&ho, 0, sizeof ho);
            ho.uttid = uq->uttid; ho.lmname = lm ? lm->name : NULL; ho.have_lm = lm != NULL; ho.lm_wip = lm ? lm->wip : 0; ho.lm_lw = lm ? lm->lw : 1.0f;
            ho.frate = cmd_ln_exists_r(config, "-frate") ? cmd_ln_int32_r(config, "-frate") : 0;
            ho.log_shift = logmath_get_shift(lmath); ho.log_of_base = log(logmath_get_base(lmath));
            ho.opt_lw = cmd_ln_float32_r(config, "-lw"); ho.opt_wip = cmd_ln_float32_r(config, "-wip");
            ho.basewid = base; ho.n_alt = nalt;
            need = s3a_lattice_format_htk(hdr, &ho, &info, nodes, links, (const char *const *)g_wordstr, NULL, 0);
            buf = ckd_calloc(need + 1, 1);
            s3a_lattice_format_htk(hdr, &ho, &info, nodes, links, (const char *const *)g_wordstr, buf, need + 1);
            ckd_free(base); ckd_free(nalt);
        }
        else {
            need = s3a_lattice_format_s3(hdr, &info, nodes, links, (const char *const *)g_wordstr, NULL, 0);
            buf = ckd_calloc(need + 1, 1);
            s3a_lattice_format_s3(hdr, &info, nodes, links, (const char *const *)g_wordstr, buf, need + 1);
        }
        ctl_outfile(str, cmd_ln_str_r(config, "-outlatdir"), cmd_ln_str_r(config, "-latext"), (uq->uttfile ? uq->uttfile : uq->uttid), uq->uttid,
                    cmd_ln_boolean_r(config, "-build_outdirs"));
        if ((fp = fopen_comp(str, "w", &ispipe)) == NULL) E_ERROR("fopen_comp (%s,w) failed\n", str);
        else { fwrite(buf, 1, need, fp); fclose_comp(fp, ispipe); }
        ckd_free(buf); free(hdr);
    }
    if (cmd_ln_str_r(config, "-nbestdir")) {
        s3a_nbest_opts_t o;
        s3a_nbest_t *nb;
        const char *text = NULL;
        int64_t len = 0;
        int32 n_hyp = 0, cnt[4] = { 0, 0, 0, 0 };
        memset(&o, 0, sizeof o);
        o.uttid = uq->uttid; o.beam = cmd_ln_float64_r(config, "-beam"); o.beam_logs3 = logs3(kbcore_logmath(kbc), o.beam);
        o.nbest = cmd_ln_int32_r(config, "-nbest"); o.maxppath = cmd_ln_int32_r(config, "-maxppath");
        o.lm_wip = lm->wip; o.lm_lw = lm->lw; o.logbase = cmd_ln_float32_r(config, "-logbase"); o.lw = cmd_ln_float32_r(config, "-lw"); o.wip = cmd_ln_float32_r(config, "-wip");
        ctl_outfile(str, cmd_ln_str_r(config, "-nbestdir"), cmd_ln_str_r(config, "-nbestext"), (uq->uttfile ? uq->uttfile : uq->uttid), uq->uttid,
                    cmd_ln_boolean_r(config, "-build_outdirs"));
        nb = s3a_lattice_nbest(g_lm3g, &g_dag_cfg, &o, &info, nodes, links, (const char *const *)g_wordstr);
        if (nb == NULL) die("s3a_lattice_nbest");
        if (s3a_nbest_result(nb, &text, &len, &n_hyp, cnt) != S3A_OK) E_ERROR("maxedge limit (%d) exceeded\n", g_dag_cfg.maxedge);
        else if (n_hyp <= 0) E_ERROR("%s: A* search failed\n", uq->uttid);
        else if ((fp = fopen_comp(str, "w", &ispipe)) == NULL) E_ERROR("fopen_comp (%s,w) failed\n", str);
        else { fwrite(text, 1, (size_t)len, fp); fclose_comp(fp, ispipe); }
        s3a_nbest_free(nb);
    }
    ckd_free(nodes); ckd_free(links);
}

static void
utt_flush_queue(kb_t *kb)
{
    static const char **wstr;
    static int32 *base;
    kbcore_t *kbc = kb->kbcore;
    dict_t *dict = kbcore_dict(kbc);
    cmd_ln_t *config = kbcore_config(kbc);
    const int32 n = g_uq_n, veclen = kbcore_fcb(kbc)->stream_len[0];
    int32 *order, *nfr_all, *slot_e, *slot_q, *nfr2, cnt[UTT_MAX_ENGINES], off[UTT_MAX_ENGINES + 1], z, e, n_used = 0;
    const float **feat2;
    eng_job_t *job = g_job;
    double t0;
    if (n == 0) return;
    if (!wstr) {
        const int32 nw = dict_size(dict);
        int32 i;
        wstr = ckd_calloc(nw + 1, sizeof(char *));
        base = ckd_calloc(nw + 1, 4);
        for (i = 0; i < nw; i++) { wstr[i] = dict_wordstr(dict, i); base[i] = dict_basewid(dict, i); }
    }
    order = ckd_calloc(n, 4); nfr_all = ckd_calloc(n, 4); slot_e = ckd_calloc(n, 4); slot_q = ckd_calloc(n, 4);
    nfr2 = ckd_calloc(n, 4); feat2 = ckd_calloc(n, sizeof(*feat2));
    for (z = 0; z < n; z++) { order[z] = z; nfr_all[z] = g_uq[z].nfr; g_utt_frames += g_uq[z].nfr; }
    g_sort_nfr = nfr_all;
    qsort(order, n, sizeof(int32), cmp_longest_first);
    for (e = 0; e < g_n_eng; e++) cnt[e] = 0;
    for (z = 0; z < n; z++) cnt[z % g_n_eng]++;
    off[0] = 0;
    for (e = 0; e < g_n_eng; e++) { off[e + 1] = off[e] + cnt[e]; if (cnt[e] > 0) n_used = e + 1; }
    for (z = 0; z < n; z++) {
        const int32 u = order[z], q = z / g_n_eng;
        e = z % g_n_eng;
        slot_e[u] = e; slot_q[u] = q;
        feat2[off[e] + q] = g_uq[u].feat; nfr2[off[e] + q] = g_uq[u].nfr;
    }
    t0 = now_s();
    for (e = 0; e < n_used; e++) {
        job[e].e = e; job[e].n = cnt[e]; job[e].feat = feat2 + off[e]; job[e].nfr = nfr2 + off[e];
        job[e].veclen = veclen; job[e].rc = S3A_OK; job[e].queue = 1; job[e].dev = g_adcin;
    }
    if (g_n_eng == 1) job[0].rc = eng_decode(&job[0]);
    else {
        pthread_mutex_lock(&g_eng_lock);
        for (e = 0; e < n_used; e++) job[e].state = 1;
        pthread_cond_broadcast(&g_eng_cv);
        for (e = 0; e < n_used; e++) while (job[e].state != 2) pthread_cond_wait(&g_eng_cv, &g_eng_lock);
        for (e = 0; e < n_used; e++) job[e].state = 0;
        pthread_mutex_unlock(&g_eng_lock);
    }
    for (e = 0; e < n_used; e++)
        if (job[e].rc != S3A_OK) {
            int32 bad = 0, q, err;
            for (q = 0; q < job[e].n; q++) if (s3a_uttdec_queue_status(g_uds[e], q, &err, NULL, NULL, NULL) == S3A_OK && err) bad++;
            if (!bad) die("uttdec decode_queue");
            E_ERROR("tst shim: %d utterance(s) of this queue were not decoded: %s\n", bad, s3a_last_error());
        }
    g_t_dev += now_s() - t0;
    t0 = now_s();
    for (z = 0; z < n; z++) {           /* control-file order */
        s3a_uttdec_t *ud = g_uds[slot_e[z]];
        s3a_hyp_header_t h;
        s3a_hyp_word_t *words = NULL;
        int32 need = 0, cap = 0, err = 0, mc = 0, mn = 0;
        for (;;) {
            if (need > cap) { cap = need + 64; words = ckd_realloc(words, (size_t)cap * sizeof(*words)); }
            /* -bestpath 1: the second pass ran on the device at the lane's refill event; its hypothesis is the utterance's */
            /* (-outlatdir / -nbestdir without -bestpath: the pass ran for the lattice, the hypothesis is the first pass's, srch.c:538-552) */
            if (((g_dev_dag && cmd_ln_boolean_r(config, "-bestpath")) ? s3a_uttdec_queue_bestpath_hyp(ud, slot_q[z], g_uq[z].uttid, g_rank_first + g_rec_n, &h, words, cap)
                           : s3a_uttdec_queue_hyp(ud, slot_q[z], g_uq[z].uttid, g_rank_first + g_rec_n, &h, words, cap)) != S3A_OK) die("hypothesis record");
            if (h.status != -3) break;
            need = h.n_words;
        }
        (void)s3a_uttdec_queue_status(ud, slot_q[z], &err, NULL, &mc, &mn);
        if (mc > g_max_cand) g_max_cand = mc;
        if (mn > g_max_new) g_max_new = mn;
        g_frames += g_uq[z].nfr;
        if (g_gather) {
            if (g_rec_n == g_rec_cap) { g_rec_cap = g_rec_cap ? 2 * g_rec_cap : 1024; g_rec_hdr = ckd_realloc(g_rec_hdr, (size_t)g_rec_cap * sizeof(*g_rec_hdr)); }
            if (g_rec_nw + h.n_words > g_rec_wcap) { g_rec_wcap = 2 * (g_rec_nw + h.n_words) + 4096; g_rec_words = ckd_realloc(g_rec_words, (size_t)g_rec_wcap * sizeof(*g_rec_words)); }
            if (h.status == 0 && h.n_words > 0) memcpy(g_rec_words + g_rec_nw, words, (size_t)h.n_words * sizeof(*words));
            g_rec_hdr[g_rec_n++] = h;
            if (h.status == 0) g_rec_nw += h.n_words;
        }
        if (h.status == -1) {
            E_ERROR("tst shim: utterance %s stopped on the device (error bits 0x%x: a capacity of its lane -- S3A_UTT_VHCAP / S3A_UTT_CANDCAP); no hypothesis written\n",
                    g_uq[z].uttid, err);
            g_failed_utts++;
        }
        else if (h.status == -4) E_ERROR("Bestpath search failed for %s\n", g_uq[z].uttid);      /* (dag.c:945: no line) */
        else if (h.status == -5) {
            E_ERROR("the device's second pass gave up on %s: %s\n", g_uq[z].uttid, s3a_last_error());
            g_failed_utts++;
        }
        else if (h.status != 0)
            E_ERROR("s->funcs->utt_end failed\n");     /* (srch.c:495-498: no word exit reached a history entry; no line) */
        else if (g_dev_dag) g_dag_utts++;
        if (h.status == 0) {
            const size_t lcap = 65536 + 64 * (size_t)h.n_words;
            char *m = ckd_calloc(lcap, 1), *sg = ckd_calloc(lcap, 1);
            if (s3a_hyp_format_var(&h, words, wstr, base, g_wflat->is_filler, g_wflat->startwid, g_wflat->finishwid, (float)kbcore_lm(kbc)->lw,
                                   kbcore_lm(kbc)->wip, cmd_ln_int32_r(config, "-hypsegscore_unscale"), m, lcap, sg, lcap) != S3A_OK) die("s3a_hyp_format_var");
            if (kb->matchfp) fputs(m, kb->matchfp);
            if (kb->matchsegfp) fputs(sg, kb->matchsegfp);
            ckd_free(m); ckd_free(sg);
        }
        if (g_queue_lat && h.status != -1) queue_write_lattice_files(kb, ud, slot_q[z], &g_uq[z]);
        ckd_free(words);
        uq_free(&g_uq[z]);
    }
    g_t_fin += now_s() - t0;
    g_uq_n = 0;
    ckd_free(order); ckd_free(nfr_all); ckd_free(slot_e); ckd_free(slot_q); ckd_free(nfr2); ckd_free(feat2);
}

/* the queue, g_lpe utterances per engine; the engines (own stream each) decode side by side, a host thread each --
 * the tails of one engine's launches are another's work */
static void
utt_flush(kb_t *kb)
{
    if (g_queue) { utt_flush_queue(kb); return; }
    const float **feat;
    int32 *nfr, z, e, n_used;
    double t0;
    eng_job_t *job = g_job;
    if (g_uq_n == 0) return;
    feat = ckd_calloc(g_uq_n, sizeof(*feat));
    nfr = ckd_calloc(g_uq_n, sizeof(*nfr));
    for (z = 0; z < g_uq_n; z++) { feat[z] = g_uq[z].feat; nfr[z] = g_uq[z].nfr; g_utt_frames += g_uq[z].nfr; }
    t0 = now_s();
    n_used = (g_uq_n + g_lpe - 1) / g_lpe;
    for (e = 0; e < n_used; e++) {
        job[e].e = e; job[e].n = (e + 1) * g_lpe <= g_uq_n ? g_lpe : g_uq_n - e * g_lpe;
        job[e].feat = feat + e * g_lpe; job[e].nfr = nfr + e * g_lpe;
        job[e].veclen = kbcore_fcb(kb->kbcore)->stream_len[0]; job[e].rc = S3A_OK; job[e].queue = 0; job[e].dev = g_adcin;
    }
    if (g_n_eng == 1) job[0].rc = eng_decode(&job[0]);
    else {
        pthread_mutex_lock(&g_eng_lock);
        for (e = 0; e < n_used; e++) job[e].state = 1;
        pthread_cond_broadcast(&g_eng_cv);
        for (e = 0; e < n_used; e++) while (job[e].state != 2) pthread_cond_wait(&g_eng_cv, &g_eng_lock);
        for (e = 0; e < n_used; e++) job[e].state = 0;
        pthread_mutex_unlock(&g_eng_lock);
    }
    /* a decode call fails when ONE of its utterances could not be decoded (a capacity of that lane: history table,
     * candidate buffers): the other lanes' results are complete -- they are finished as usual, the failed utterance is
     * reported and gets no -hyp / -hypseg line (as an utterance the reference fails on), and the run ends with status 1.
     * Anything else (no lane names an error: the device, the arguments) ends the run here. */
    for (e = 0; e < n_used; e++)
        if (job[e].rc != S3A_OK) {
            s3a_utt_result_t r;
            int32 bad = 0;
            for (z = 0; z < job[e].n; z++) if (s3a_uttdec_result(g_uds[e], z, &r) == S3A_OK && r.err) bad++;
            if (!bad) die("uttdec decode");
            E_ERROR("tst shim: %d utterance(s) of this batch were not decoded: %s\n", bad, s3a_last_error());
        }
    g_t_dev += now_s() - t0;
    t0 = now_s();
    for (z = 0; z < g_uq_n; z++) utt_finish(kb, z);
    g_t_fin += now_s() - t0;
    g_uq_n = 0;
    ckd_free(feat); ckd_free(nfr);
}

/* -adcin: the utterance's samples (what utt.c's static wavfile_read hands utt_decode: <-cepdir>/<file><-cepext>, -adchdr
 * bytes of header skipped, the rest 16-bit samples in the machine's byte order) */
static int16 *
adc_read(const char *uttfile, int32 *nsamps, cmd_ln_t *config)
{
    const char *ext = cmd_ln_str_r(config, "-cepext"), *dir = cmd_ln_str_r(config, "-cepdir");
    const int32 hdr = cmd_ln_int32_r(config, "-adchdr");
    const size_t le = strlen(ext), lf = strlen(uttfile);
    char *path = ckd_calloc((dir ? strlen(dir) : 0) + lf + le + 2, 1);
    FILE *fp;
    long bytes;
    int16 *data = NULL;
    if (le <= lf && strcmp(uttfile + lf - le, ext) == 0) ext = "";
    if (dir) sprintf(path, "%s/%s%s", dir, uttfile, ext); else sprintf(path, "%s%s", uttfile, ext);
    if ((fp = fopen(path, "rb")) == NULL) E_FATAL("fopen(%s,rb) failed\n", path);
    fseek(fp, 0, SEEK_END);
    bytes = ftell(fp) - (hdr > 0 ? hdr : 0);
    if (bytes >= 0 && fseek(fp, hdr > 0 ? hdr : 0, SEEK_SET) == 0) {
        const long n = bytes / (long)sizeof(int16);
        data = ckd_calloc(n > 0 ? n : 1, sizeof(int16));
        if ((long)fread(data, sizeof(int16), n, fp) < n) { E_ERROR("Failed to read %ld samples from %s\n", n, path); ckd_free(data); data = NULL; }
        else *nsamps = (int32)n;
    }
    fclose(fp);
    ckd_free(path);
    return data;
}

static int g_dither, g_swap, g_cmn_prior;
/* the front end fe_init_auto_r (fe_interface.c:212-283) builds from the same options */
static void
adc_frontend_init(cmd_ln_t *config, kbcore_t *kbc)
{
    s3a_fe_params_t p;
    const char *tr = cmd_ln_str_r(config, "-transform"), *cmn = cmd_ln_str_r(config, "-cmn"), *agc = cmd_ln_str_r(config, "-agc");
    if (strcmp(kbcore_fcb(kbc)->name, "1s_c_d_dd") != 0) E_FATAL("tst shim: -adcin with S3A_UTT: feature type 1s_c_d_dd only (is %s)\n", kbcore_fcb(kbc)->name);
    /* -dither / -input_endian act on the samples as they enter the front end's frame buffer (fe_read_frame / fe_shift_frame,
     * fe_sigproc.c:596-640), once per sample and in sample order: done on the samples before they go to the device (utt_collect) */
    g_dither = cmd_ln_boolean_r(config, "-dither") ? 1 : 0;
    g_swap = strcmp(cmd_ln_str_r(config, "-input_endian"), "little") != 0;      /* (this host is little-endian: fe_interface.c:80-84) */
    s3a_fe_default_params(&p);
    p.samprate = cmd_ln_float32_r(config, "-samprate"); p.frate = cmd_ln_int32_r(config, "-frate"); p.wlen = cmd_ln_float32_r(config, "-wlen");
    p.alpha = cmd_ln_float32_r(config, "-alpha"); p.ncep = cmd_ln_int32_r(config, "-ncep"); p.nfft = cmd_ln_int32_r(config, "-nfft");
    p.nfilt = cmd_ln_int32_r(config, "-nfilt"); p.lowerf = cmd_ln_float32_r(config, "-lowerf"); p.upperf = cmd_ln_float32_r(config, "-upperf");
    p.transform = strcmp(tr, "dct") == 0 ? S3A_FE_DCT : strcmp(tr, "htk") == 0 ? S3A_FE_HTK : S3A_FE_LEGACY;
    p.lifter = cmd_ln_int32_r(config, "-lifter"); p.remove_dc = cmd_ln_boolean_r(config, "-remove_dc");
    p.round_filters = cmd_ln_boolean_r(config, "-round_filters"); p.unit_area = cmd_ln_boolean_r(config, "-unit_area");
    p.doublebw = cmd_ln_boolean_r(config, "-doublebw");
    p.logspec = cmd_ln_boolean_r(config, "-smoothspec") ? S3A_FE_SMOOTHSPEC : cmd_ln_boolean_r(config, "-logspec") ? S3A_FE_LOGSPEC : S3A_FE_CEPSTRA;
    if (cmd_ln_str_r(config, "-warp_params") != NULL) {        /* fe_warp_set / fe_warp_set_parameters (fe_warp.c:108-178) */
        const char *wt = cmd_ln_str_r(config, "-warp_type");
        float a = 0.0f, b = 0.0f;
        (void)sscanf(cmd_ln_str_r(config, "-warp_params"), "%f %f", &a, &b);
        p.warp_type = (strcmp(wt, "inverse_linear") == 0 || strcmp(wt, "inverse") == 0) ? S3A_FE_WARP_INVERSE
            : (strcmp(wt, "affine") == 0 || strcmp(wt, "linear") == 0) ? S3A_FE_WARP_AFFINE
            : (strcmp(wt, "piecewise_linear") == 0 || strcmp(wt, "piecewise") == 0) ? S3A_FE_WARP_PIECEWISE : -1;
        if (p.warp_type < 0) E_FATAL("tst shim: -warp_type %s is unknown\n", wt);
        p.warp_params[0] = a; p.warp_params[1] = b;
    }
    if ((g_fe = s3a_fe_init(&p)) == NULL) E_FATAL("tst shim: s3a_fe_init: %s\n", s3a_last_error());
    /* -cmn prior: the mean learnt from the utterances before (cmn_t of the decoder's feat_t, -cmninit) is subtracted on the device
     * and the running sums take the utterance there; the state between utterances stays in the reference's cmn_t (utt_collect) */
    g_cmn_prior = strcmp(cmn, "prior") == 0;
    if (g_cmn_prior && cmd_ln_boolean_r(config, "-varnorm")) E_FATAL("Variance normalization not implemented in live mode decode\n");   /* (cmn_prior.c:150-152) */
    if (strcmp(cmn, "current") != 0 && strcmp(cmn, "none") != 0 && !g_cmn_prior) E_FATAL("tst shim: -adcin with S3A_UTT: -cmn current / prior / none only\n");
    if (strcmp(agc, "max") != 0 && strcmp(agc, "none") != 0) E_FATAL("tst shim: -adcin with S3A_UTT: -agc max / none only\n");
    g_cmn_current = strcmp(cmn, "current") == 0; g_agc_max = strcmp(agc, "max") == 0;
    g_varnorm = cmd_ln_boolean_r(config, "-varnorm");
    g_adcin = 1;
    g_feat_dim = feat_dimension(kbcore_fcb(kbc));
}

/* (the LM contexts: below, with utt_mode_main) */
typedef struct { const char *name; flat_t **flat; s3a_lexsearch_t *ls; s3a_lm3g_t *lm3g; wl_flat_t *w; s3a_uttdec_t *uds[UTT_MAX_ENGINES]; s3a_dag_cfg_t dagc; } lmctx_t;
static lmctx_t *g_ctx;
static int32 g_n_ctx, g_cur_ctx;
static void ctx_switch(kb_t *kb, int32 k);

/* ctl_process callback: utt_decode's feature half (libAPI/utt.c:185-245), then queue */
static void
utt_collect(void *data, utt_res_t *ur, int32 sf, int32 ef, char *uttid)
{
    kb_t *kb = data;
    kbcore_t *kbcore = kb->kbcore;
    cmd_ln_t *config = kbcore_config(kbcore);
    int32 total_frame, veclen = kbcore_fcb(kbcore)->stream_len[0], t;
    uq_t *q;
    double t0 = now_s();

    if (ur->lmname != NULL) {                   /* -ctl_lm (utt.c:240-241) */
        int32 k;
        for (k = 0; k < g_n_ctx; k++) if (strcmp(g_ctx[k].name, ur->lmname) == 0) break;
        if (k == g_n_ctx) E_FATAL("tst shim: -ctl_lm names %s, which is not in the LM set\n", ur->lmname);
        if (k != g_cur_ctx) { utt_flush(kb); ctx_switch(kb, k); }
    }
    if (ur->regmatname != NULL && strcmp(ur->regmatname, kb->adapt_am->prevmllrfn) != 0) {
        /* -ctl_mllr names another regression matrix: what is queued is decoded with the model it was queued for, then the
         * host model is adapted (kb_setmllr, as utt_decode would: utt.c:245-246) and every engine's device model follows */
        utt_flush(kb);
        kb_setmllr(ur->regmatname, ur->cb2mllrname, kb);
        adapt_sync(kb, g_gms, g_n_eng);
    }
    q = &g_uq[g_uq_n++];
    q->uttid = ckd_salloc(uttid);
    q->uttfile = ckd_salloc(ur->uttfile);
    if (g_adcin) {
        /* -adcin: raw audio -> MFCC -> features on the device (utt.c:208-233 does it with fe_process_utt and
         * feat_s2mfc2feat_live on the host); the features stay in HBM and the engines read them there */
        int32 nsamps = 0, stride = 0;
        int16 *adc = adc_read(ur->uttfile, &nsamps, config);
        float *dfeat = NULL;
        (void)sf; (void)ef; (void)t;
        if (adc == NULL) E_FATAL("Cannot read file %s. Forced exit\n", ur->uttfile);
        if (g_swap) { int32 k; for (k = 0; k < nsamps; k++) SWAP_INT16(&adc[k]); }
        if (g_dither) {
            /* every sample that enters a frame gets one draw of the generator kb_init's fe_init_auto_r seeded (-seed), in
             * sample order; the samples behind the last whole frame enter none (utt_decode calls no fe_end_utt).  The
             * generator's state runs on from utterance to utterance, so the control file is walked in order here. */
            const int32 fs = s3a_fe_frame_size(g_fe), sh = s3a_fe_frame_shift(g_fe);
            if (nsamps >= fs) {
                const int32 used = fs + ((nsamps - fs) / sh) * sh;
                int32 k;
                for (k = 0; k < used; k++) adc[k] += (int16)((!(s3_rand_int31() % 4)) ? 1 : 0);
            }
        }
        if (g_cmn_prior) {
            /* feat_s2mfc2feat_live(beginutt, endutt) = feat_s2mfc2feat_block_utt -> feat_cmn: cmn_prior over the padded utterance,
             * then cmn_prior_update (feat.c:1066-1080, cmn_prior.c:95-170) */
            cmn_t *cm = kbcore_fcb(kbcore)->cmn_struct;
            if (s3a_audio_to_feat_dev_prior(g_fe, adc, nsamps, 1, cm->cmn_mean, cm->sum, g_agc_max, &dfeat, &total_frame, &stride) != S3A_OK)
                E_FATAL("tst shim: MFCC / feature computation failed for %s: %s\n", ur->uttfile, s3a_last_error());
            cm->nframe += total_frame + 2 * feat_window_size(kbcore_fcb(kbcore));
            if (cm->nframe > CMN_WIN_HWM) {            /* cmn_prior_shiftwin (static in cmn_prior.c:95-112), restated */
                mfcc_t sf = FLOAT2MFCC(1.0) / cm->nframe;
                int32 i;
                for (i = 0; i < cm->veclen; i++) cm->cmn_mean[i] = cm->sum[i] / cm->nframe;
                if (cm->nframe >= CMN_WIN_HWM) {
                    sf = CMN_WIN * sf;
                    for (i = 0; i < cm->veclen; i++) cm->sum[i] = MFCCMUL(cm->sum[i], sf);
                    cm->nframe = CMN_WIN;
                }
            }
            cmn_prior_update(cm);
        }
        else if (s3a_audio_to_feat_dev(g_fe, adc, nsamps, 1, g_cmn_current, g_varnorm, g_agc_max, &dfeat, &total_frame, &stride) != S3A_OK)
            E_FATAL("tst shim: MFCC / feature computation failed for %s: %s\n", ur->uttfile, s3a_last_error());
        ckd_free(adc);
        if (kbcore_fcb(kbcore)->lda) {         /* -lda / -ldadim: feat_lda_transform behind the feature computation (feat.c:1215-1216) */
            feat_t *fcb = kbcore_fcb(kbcore);
            if (s3a_feat_lda_dev(&dfeat, total_frame, &stride, &fcb->lda[0][0][0], fcb->stream_len[0], feat_dimension(fcb), s3a_fe_stream(g_fe)) != S3A_OK)
                E_FATAL("tst shim: LDA transform failed for %s: %s\n", ur->uttfile, s3a_last_error());
        }
        if (total_frame > S3_MAX_FRAMES) E_FATAL("Maximum number of frames (%d) exceeded\n", S3_MAX_FRAMES);
        if (stride != 4 * ((feat_dimension(kbcore_fcb(kbcore)) + 3) / 4)) E_FATAL("tst shim: -adcin: the front end's features (%d floats per row) do not fit the model's %d-dimensional stream\n", stride, feat_dimension(kbcore_fcb(kbcore)));
        q->nfr = total_frame; q->feat = dfeat; q->on_dev = 1;
    }
    else {
        if ((total_frame = feat_s2mfc2feat(kbcore_fcb(kbcore), ur->uttfile, cmd_ln_str_r(config, "-cepdir"),
                                           cmd_ln_str_r(config, "-cepext"), sf, ef, kb->feat, S3_MAX_FRAMES)) < 0)
            E_FATAL("Cannot read file %s. Forced exit\n", ur->uttfile);
        q->nfr = total_frame; q->on_dev = 0;
        q->feat = ckd_calloc((size_t)total_frame * veclen + 1, sizeof(float32));
        for (t = 0; t < total_frame; t++)
            memcpy(q->feat + (size_t)t * veclen, kb->feat[t][0], veclen * sizeof(float32));
    }
    g_t_feat += now_s() - t0;
    if (g_uq_n == g_uq_cap) utt_flush(kb);
}


#include "s3amd_export.h"

/* Everything that depends on the LANGUAGE MODEL: the flattened trigram, the word level's per-word tables and the engines on the
 * current lextrees (g_ls).  One LM: once.  Several (-lmctlfn): once per LM, each after srch_set_lm has made it the current one
 * (every LM has unigram lextrees of its own: srch_time_switch_tree.c:260-330). */
static wl_flat_t *g_w;
static s3a_wordlevel_cfg_t g_cfg;
static void
lm_context_init(kb_t *kbp, int with_engines)
{
#define kb (*kbp)
    cmd_ln_t *config = cmd_ln_get();
    srch_t *s = kb.srch;
    srch_TST_graph_t *tstg = s->grh->graph_struct;
    kbcore_t *kbc = kb.kbcore;
    mdef_t *mdef = kbcore_mdef(kbc);
    wl_flat_t *w;
    s3a_wordlevel_cfg_t cfg;
    int32 *tree_type, t, e;
    w = flatten_lm(kbc);
    g_lm3g = s3a_lm3g_init(w->n_ug, w->ug_prob, w->ug_bowt, w->ug_firstbg, w->n_bg, w->bg_wid, w->bg_prob, w->bg_bowt,
                           w->bg_firsttg, w->n_tg, w->tg_wid, w->tg_prob, w->inclass, w->n_word);
    if (!g_lm3g) die("s3a_lm3g_init");
    tree_type = ckd_calloc(g_ntree, 4);
    for (t = 0; t < g_ntree; t++) tree_type[t] = g_flat[t]->type;
    memset(&cfg, 0, sizeof cfg);
    cfg.n_word = w->n_word; cfg.n_ci = w->n_ci; cfg.lwid = w->lwid; cfg.is_filler = w->is_filler; cfg.fillpen = w->fillpen;
    cfg.last_ci = w->last_ci; cfg.startwid = w->startwid; cfg.finishwid = w->finishwid; cfg.silwid = w->silwid;
    cfg.start_lwid = w->start_lwid; cfg.finish_lwid = w->finish_lwid; cfg.sil_ci = mdef_silphone(mdef);
    cfg.wbeam_vh = tstg->vithist->wbeam; cfg.bghist = tstg->vithist->bghist;
    cfg.maxwpf = tstg->histprune->maxwpf; cfg.maxhistpf = tstg->histprune->maxhistpf;
    cfg.wordend_beam = s->beam->wordend; cfg.n_lextree = tstg->n_lextree; cfg.epl = tstg->epl;
    cfg.hmmbeam = s->beam->hmm; cfg.pbeam = s->beam->ptrans; cfg.wbeam = s->beam->word;
    cfg.ptranskip = s->beam->ptranskip; cfg.maxhmmpf = tstg->histprune->maxhmmpf; cfg.tree_type = tree_type;
    g_w = w; g_cfg = cfg;
    if (!with_engines) return;
        for (e = 0; e < g_n_eng; e++) {
            /* every further engine gets a model of its own: an engine runs on its model's stream, and engines are
             * to overlap (the lextrees, the trigram and the composite-senone table are shared) */
            s3a_mgau_model_t *gm = g_gm;
            if (e > 0 && g_gms[e]) gm = g_gms[e];          /* (another LM's context: the engines of one index share their model) */
            else if (e > 0) {
                gm = s3a_mgau_init(cmd_ln_str_r(config, "-mean"), cmd_ln_str_r(config, "-var"),
                                   cmd_ln_float32_r(config, "-varfloor"), cmd_ln_str_r(config, "-mixw"),
                                   cmd_ln_float32_r(config, "-mixwfloor"), 1, ".cont.", S3A_MIX_INT_FLOAT_COMP, g_lm);
                if (!gm) die("s3a_mgau_init");
                if (g_mllr_cur[0]) adapt_upload(&kb, gm);       /* -mllr: this engine's model too */
            }
            g_gms[e] = gm;
            s3a_uttdec_opts_t uo;
            s3a_uttdec_opts_from_env(&uo);          /* (this program's tuning switches are environment variables; the library takes arguments) */
            if (getenv("S3A_KF_RELAY_AT") || getenv("S3A_KF_NO_RELAY")) {      /* (the relay of ku_frames' launches: tests run its chain with few lanes) */
                s3a_variants_t va;
                s3a_get_variants(&va);
                va.kf_relay_at = getenv("S3A_KF_RELAY_AT") ? atoi(getenv("S3A_KF_RELAY_AT")) : 0;
                va.kf_no_relay = getenv("S3A_KF_NO_RELAY") != NULL;
                if (s3a_set_variants(&va) != S3A_OK) die("s3a_set_variants");
            }
            g_uds[e] = s3a_uttdec_init_opts(g_ls, gm, mdef->cd2cisen, mdef_n_sen(mdef), mdef->n_ci_sen, cmd_ln_int32_r(config, "-ds"),
                           cmd_ln_int32_r(config, "-cond_ds"), cmd_ln_float64_r(config, "-ci_pbeam"),
                           cmd_ln_float32_r(config, "-tighten_factor"), cmd_ln_int32_r(config, "-maxcdsenpf"), g_cs,
                           g_lm3g, &cfg, g_lpe, S3_MAX_FRAMES, getenv("S3A_UTT_VHCAP") ? atoi(getenv("S3A_UTT_VHCAP")) : 0,
                           getenv("S3A_UTT_CANDCAP") ? atoi(getenv("S3A_UTT_CANDCAP")) : 0, &uo);
            if (!g_uds[e]) die("s3a_uttdec_init");
            if (kb.pl->pheurtype != 0) {        /* -pheurtype 1..3: phoneme look-ahead inside the engine */
                const uint8_t **nci = ckd_calloc(g_ntree, sizeof(*nci));
                int32 t;
                for (t = 0; t < g_ntree; t++) nci[t] = g_flat[t]->ci;
                if (s3a_uttdec_enable_pheur(g_uds[e], kb.pl->pheurtype, kb.pl->pl_beam, cmd_ln_int32_r(config, "-pl_window"), nci,
                                            mdef->sen2cimap, mdef_n_ciphone(mdef)) != S3A_OK) die("s3a_uttdec_enable_pheur");
                ckd_free(nci);
            }
            /* the second pass on the device; lattice files (-outlatdir) are written from the device's lattice, N-best lists
             * (-nbestdir) by the reference's own A* search on a dag_t poured from it (utt_gen_dag_slot) */
            if ((cmd_ln_boolean_r(config, "-bestpath") || cmd_ln_str_r(config, "-outlatdir") || cmd_ln_str_r(config, "-nbestdir"))
                && !getenv("S3A_UTT_HOSTDAG")) {
                s3a_dag_cfg_t dc;
                float32 bplw = cmd_ln_float32_r(config, "-bestpathlw");
                int32 *base = ckd_calloc(w->n_word + 1, 4), i;
                for (i = 0; i < w->n_word; i++) base[i] = dict_basewid(kbcore_dict(kbc), i);
                memset(&dc, 0, sizeof dc);
                dc.n_word = w->n_word; dc.basewid = base; dc.is_filler = w->is_filler; dc.lwid = w->lwid; dc.fillpen = w->fillpen;
                dc.startwid = w->startwid; dc.finishwid = w->finishwid; dc.silwid = w->silwid; dc.start_lwid = w->start_lwid;
                dc.finish_lwid = w->finish_lwid; dc.wip = logs3(kbcore_logmath(kbc), kbcore_fillpen(kbc)->wip);
                dc.lwf = bplw ? (bplw / cmd_ln_float32_r(config, "-lw")) : 1.0;
                dc.min_endfr = cmd_ln_int32_r(config, "-min_endfr"); dc.maxedge = cmd_ln_int32_r(config, "-maxedge");
                dc.maxlmop = cmd_ln_int32_r(config, "-maxlmop"); dc.maxlpf = cmd_ln_int32_r(config, "-maxlpf");
                if (s3a_uttdec_enable_bestpath(g_uds[e], &dc, getenv("S3A_DAG_LINKS") ? atoi(getenv("S3A_DAG_LINKS")) : 0,
                                               getenv("S3A_DAG_PAIRS") ? atoi(getenv("S3A_DAG_PAIRS")) : 0, 1) != S3A_OK) die("s3a_uttdec_enable_bestpath");
                if (e == 0) g_dag_cfg = dc; else ckd_free(base);      /* (the first engine's copy serves utt_nbest_slot: the library's N-best reads the word tables) */
                g_dev_dag = 1;
            }
        }
#undef kb
}

/* -lmctlfn / -ctl_lm / -lmname: a context per LM of the set -- its lextrees flattened, its search space, its trigram, its engines
 * (the acoustic models and their streams are shared by the engines of the same index: contexts never run side by side).
 * utt.c:240-241 switches with srch_set_lm before an utterance; here what is queued is decoded with the LM it was queued for,
 * then the LM's context becomes the current one. */
static void
ctx_save(int32 k, const char *name)
{
    int32 e;
    g_ctx[k].name = name; g_ctx[k].flat = g_flat; g_ctx[k].ls = g_ls; g_ctx[k].lm3g = g_lm3g; g_ctx[k].w = g_w; g_ctx[k].dagc = g_dag_cfg;
    for (e = 0; e < g_n_eng; e++) g_ctx[k].uds[e] = g_uds[e];
}
static void
ctx_switch(kb_t *kb, int32 k)
{
    int32 e;
    if (k == g_cur_ctx) return;
    srch_set_lm((srch_t *)kb->srch, g_ctx[k].name);
    g_flat = g_ctx[k].flat; g_ls = g_ctx[k].ls; g_lm3g = g_ctx[k].lm3g; g_w = g_ctx[k].w; g_wflat = g_ctx[k].w; g_dag_cfg = g_ctx[k].dagc;
    for (e = 0; e < g_n_eng; e++) g_uds[e] = g_ctx[k].uds[e];
    g_ud = g_uds[0];
    g_cur_ctx = k;
}

static int
utt_mode_main(int argc, char *argv[], int n_lanes)
{
    static kb_t kb;
    cmd_ln_t *config = cmd_ln_get();
    srch_t *s;
    srch_TST_graph_t *tstg;
    kbcore_t *kbc;
    mdef_t *mdef;
    wl_flat_t *w;
    s3a_wordlevel_cfg_t cfg;
    int32 *tree_type, t;
    double t_load = now_s(), t_dec;
    (void)argc; (void)argv;

    kb_init(&kb, config);
    s = kb.srch;
    if (s->op_mode != 4) E_FATAL("tst shim: -op_mode 4 (fwdtree) only\n");
    tstg = s->grh->graph_struct;
    kbc = kb.kbcore;
    mdef = kbcore_mdef(kbc);
    /* with -ctl_lm the reference names no current LM until the first utterance does (lmset_init, lmset.c:143-158; the lextrees
     * in use are the first LM's: srch_time_switch_tree.c:357-360): make that LM the current one, as its trees already are */
    if (kbcore_lmset(kbc)->cur_lm == NULL) srch_set_lm(s, kbcore_lmset(kbc)->lmarray[0]->name);
    backend_init(&kb, tstg);
    g_n_eng = getenv("S3A_UTT_ENGINES") ? atoi(getenv("S3A_UTT_ENGINES")) : 1;     /* the lanes are split over the engines */
    if (g_n_eng < 1) g_n_eng = 1;
    if (g_n_eng > UTT_MAX_ENGINES) g_n_eng = UTT_MAX_ENGINES;
    if (g_n_eng > n_lanes) g_n_eng = n_lanes;
    g_lpe = (n_lanes + g_n_eng - 1) / g_n_eng;
    n_lanes = g_lpe * g_n_eng;
    lm_context_init(&kb, getenv("S3A_EXPORT") == NULL);
    w = g_w; cfg = g_cfg;
    {
        lmset_t *ls = kbcore_lmset(kbc);
        g_n_ctx = ls->n_lm;
        g_ctx = ckd_calloc(g_n_ctx > 0 ? g_n_ctx : 1, sizeof(*g_ctx));
        g_cur_ctx = 0;
        for (t = 0; t < g_n_ctx; t++) if (ls->lmarray[t] == ls->cur_lm) g_cur_ctx = t;
        ctx_save(g_cur_ctx, ls->lmarray[g_cur_ctx]->name);
        if (g_n_ctx > 1 && !getenv("S3A_EXPORT")) {
            const int32 first = g_cur_ctx;
            for (t = 0; t < g_n_ctx; t++) {
                if (t == first) continue;
                srch_set_lm(s, ls->lmarray[t]->name);
                flatten_current_trees(tstg);
                if ((g_ls = make_lexsearch(mdef, kbcore_dict2pid(kbc))) == NULL) die("s3a_lexsearch_init");
                lm_context_init(&kb, 1);
                ctx_save(t, ls->lmarray[t]->name);
                E_INFO("tst shim: LM %s: search space and %d engine(s) of its own\n", ls->lmarray[t]->name, g_n_eng);
            }
            g_cur_ctx = -1;
            ctx_switch(&kb, first);
            w = g_w;
        }
    }
    if (!getenv("S3A_EXPORT")) {
        int32 e;
        g_ud = g_uds[0];
        if (g_n_eng > 1) {              /* the engines' host threads */
            for (e = 0; e < g_n_eng; e++) {
                g_job[e].e = e; g_job[e].state = 3;
                if (pthread_create(&g_job[e].th, NULL, eng_main, &g_job[e]) != 0) die("pthread_create");
            }
            pthread_mutex_lock(&g_eng_lock);
            for (e = 0; e < g_n_eng; e++) while (g_job[e].state != 0) pthread_cond_wait(&g_eng_cv, &g_eng_lock);
            pthread_mutex_unlock(&g_eng_lock);
            g_eng_started = 1;
        }
    }
    if (getenv("S3A_EXPORT")) {
        export_bundle(getenv("S3A_EXPORT"), &kb, tstg, w, &cfg);
        return 0;
    }
    if (!g_ud) die("s3a_uttdec_init");
    /* the exchange over RCCL: between the ranks of a multi-GPU run (S3A_NO_RCCL=1: not, e.g. ranks that share one GPU);
     * S3A_GATHER=1 in a single process: the same code with one rank, the files written to <hyp>.gathered (tests) */
    g_gather = (g_world > 1 && !getenv("S3A_NO_RCCL")) || (g_world == 1 && getenv("S3A_GATHER") != NULL);
    if (g_world == 1 && g_gather) {
        FILE *cf = fopen(cmd_ln_str_r(config, "-ctl"), "r");
        char ln[16384];
        int32 nl = 0, k;
        while (cf && fgets(ln, sizeof ln, cf)) if (ln[0] != '\n' && ln[0] != '#') nl++;
        if (cf) fclose(cf);
        nl = nl > cmd_ln_int32_r(config, "-ctloffset") ? nl - cmd_ln_int32_r(config, "-ctloffset") : 0;
        if (cmd_ln_int32_r(config, "-ctlcount") >= 0 && cmd_ln_int32_r(config, "-ctlcount") < nl) nl = cmd_ln_int32_r(config, "-ctlcount");
        g_rank_total = nl;
        for (k = 0; k < 2; k++) {
            const char *nm = cmd_ln_str_r(config, k == 0 ? "-hyp" : "-hypseg");
            if (nm) snprintf(g_final[k], sizeof g_final[k], "%s.gathered", nm);
        }
    }
    s->funcs->utt_begin = utt_begin_slot;
    s->funcs->utt_end = utt_end_slot;
    if (g_dev_dag) {
        s->funcs->gen_dag = utt_gen_dag_slot; s->funcs->bestpath_impl = utt_bestpath_slot;
        if (cmd_ln_str_r(config, "-outlatdir") || cmd_ln_str_r(config, "-nbestdir")) {
            dict_t *dict = kbcore_dict(kbc);
            int32 i;
            g_want_lattice = 1;
            g_wordstr = ckd_calloc(dict_size(dict) + 1, sizeof(*g_wordstr));
            for (i = 0; i < dict_size(dict); i++) g_wordstr[i] = (char *)dict_wordstr(dict, i);
            if (cmd_ln_str_r(config, "-outlatdir") && strcmp(cmd_ln_str_r(config, "-outlatfmt"), "htk") != 0) s->funcs->dag_dump = utt_dag_dump_slot;
            if (cmd_ln_str_r(config, "-nbestdir") && !getenv("S3A_REF_NBEST")) s->funcs->nbest_impl = utt_nbest_slot;
        }
    }
    g_uq_cap = n_lanes;
    g_wflat = w;
    if (cmd_ln_exists_r(config, "-adcin") && cmd_ln_boolean_r(config, "-adcin")) adc_frontend_init(config, kbc);
    if (getenv("S3A_UTT_QUEUE")) {      /* lane refill: this many control-file entries per queue (at least the lanes) */
        if ((cmd_ln_str_r(config, "-outlatdir") || cmd_ln_str_r(config, "-nbestdir") || cmd_ln_boolean_r(config, "-bestpath")) && !g_dev_dag)
            E_FATAL("tst shim: S3A_UTT_QUEUE (lane refill) keeps no history tables: no host second pass / lattices from the host's pass in this mode (the device's second pass serves -bestpath 1, -outlatdir, -nbestdir)\n");
        g_queue = 1;
        if (cmd_ln_str_r(config, "-outlatdir") || cmd_ln_str_r(config, "-nbestdir")) {      /* (round 6: the lattices are kept behind every group's pass) */
            int32 e2;
            g_queue_lat = 1;
            for (e2 = 0; e2 < g_n_eng; e2++) if (s3a_uttdec_queue_keep_lattices(g_uds[e2], 1) != S3A_OK) die("s3a_uttdec_queue_keep_lattices");
        }
        if (atoi(getenv("S3A_UTT_QUEUE")) > g_uq_cap) g_uq_cap = atoi(getenv("S3A_UTT_QUEUE"));
    }
    g_uq = ckd_calloc(g_uq_cap, sizeof(*g_uq));
    g_ukb = &kb;
    t_load = now_s() - t_load;
    t_dec = now_s();
    kb.stat->tm = ctl_process(cmd_ln_str_r(config, "-ctl"), cmd_ln_str_r(config, "-ctl_lm"), cmd_ln_str_r(config, "-ctl_mllr"),
                              cmd_ln_int32_r(config, "-ctloffset"), cmd_ln_int32_r(config, "-ctlcount"), utt_collect, &kb);
    utt_flush(&kb);
    t_dec = now_s() - t_dec;
    if (kb.matchsegfp) fclose(kb.matchsegfp);
    if (kb.matchfp) fclose(kb.matchfp);
    if (g_gather && (g_final[0][0] || g_final[1][0])) {
        /* the end-of-batch exchange: every rank's hypothesis records to every rank over RCCL, in C; rank 0
         * writes the files of the whole control list (match_write / matchseg_write: s3a_hyp_format_var) */
        char rdv[4400];
        s3a_gather_t *gt;
        snprintf(rdv, sizeof rdv, "%s.rccl-id", g_final[0][0] ? g_final[0] : g_final[1]);
        /* (a launcher that hands its ranks a run id of its own -- S3A_RUN_ID, a number per launch; MASTER_PORT only for want of one --
         * gets the rendezvous that compares no clocks; the library removes the file, whatever its name, once every rank holds the id) */
        {
            const char *rid = getenv("S3A_RUN_ID") ? getenv("S3A_RUN_ID") : getenv("MASTER_PORT");
            const unsigned long long run_id = rid ? strtoull(rid, NULL, 10) : 0ull;
            gt = run_id ? s3a_gather_init_run(g_rank, g_world, rdv, run_id) : s3a_gather_init(g_rank, g_world, rdv);
        }
        if (gt == NULL) die("s3a_gather_init");
        if (s3a_gather_hyps(gt, g_rec_n, g_rec_hdr, g_rec_words, g_rank_total) != S3A_OK) die("s3a_gather_hyps");
        if (g_rank == 0) {
            dict_t *dict = kbcore_dict(kbc);
            const int32 nw = dict_size(dict);
            const char **wstr = ckd_calloc(nw + 1, sizeof(char *));
            int32 *base = ckd_calloc(nw + 1, 4), i;
            FILE *fh = g_final[0][0] ? fopen(g_final[0], "w") : NULL, *fs = g_final[1][0] ? fopen(g_final[1], "w") : NULL;
            for (i = 0; i < nw; i++) { wstr[i] = dict_wordstr(dict, i); base[i] = dict_basewid(dict, i); }
            for (i = 0; i < g_rank_total; i++) {
                const s3a_hyp_header_t *h;
                const s3a_hyp_word_t *ww;
                size_t cap;
                char *m, *sg;
                if (s3a_gather_result(gt, i, &h, &ww) != S3A_OK) die("s3a_gather_result");
                if (h->status != 0) continue;               /* the reference writes no line for it (srch.c:495-498) */
                cap = 65536 + 64 * (size_t)h->n_words;
                m = ckd_calloc(cap, 1); sg = ckd_calloc(cap, 1);
                if (s3a_hyp_format_var(h, ww, wstr, base, w->is_filler, w->startwid, w->finishwid, (float)kbcore_lm(kbc)->lw,
                                       kbcore_lm(kbc)->wip, cmd_ln_int32_r(config, "-hypsegscore_unscale"), m, cap, sg, cap) != S3A_OK) die("s3a_hyp_format_var");
                if (fh) fputs(m, fh);
                if (fs) fputs(sg, fs);
                ckd_free(m); ckd_free(sg);
            }
            if (fh) fclose(fh);
            if (fs) fclose(fs);
            E_INFO("tst shim: rank 0 gathered %d utterances from %d ranks over RCCL and wrote the output files\n", g_rank_total, g_world);
        }
        s3a_gather_free(gt);
    }
    if (g_frames == 0) E_FATAL("tst shim: nothing was decoded\n");
    E_INFO("tst shim: %ld frames searched by the replacement backend in %d lane(s), whole utterances on the device%s\n",
           g_frames, n_lanes, g_queue ? " (queues with lane refill)" : "");
    E_INFO("tst shim: histogram pruning (lextree_hmm_histbin) applied in %ld frames\n", g_histframes);
    if (g_dev_dag) E_INFO("tst shim: second pass (lattice + best path) of %ld utterances served by the device\n", g_dag_utts);
    E_INFO("tst shim utt mode: word level: at most %ld candidates and %ld new history entries in a frame; "
           "%ld frames replayed the reference's heap (tied scores)\n", g_max_cand, g_max_new, g_tie_frames);
    if (getenv("S3A_UTT_TICKS")) {
        int i;
        for (i = 0; i < 9; i++)
            E_INFO("tst shim utt mode: word-level phase %d of lane 0: %.2f us per frame\n", i, 0.01 * g_wl_ticks[i] / (g_frames_lane0 ? g_frames_lane0 : 1));
    }
    E_INFO("tst shim utt mode timing: device decode %.3f s (%.1f us/frame-lane, %.0f x real time aggregate), "
           "features %.3f s, hypotheses + output %.3f s\n", g_t_dev, 1e6 * g_t_dev / g_frames,
           0.01 * g_frames / g_t_dev, g_t_feat, g_t_fin);
    E_INFO("tst shim throughput: %ld frames, decode-only %.3f s = %.0f x real time aggregate "
           "(%.3f s incl. loading %d decoders one after another)\n",
           g_frames, t_dec, 0.01 * g_frames / t_dec, t_dec + t_load, 1);
    if (g_failed_utts) {
        E_ERROR("tst shim: %ld utterance(s) were not decoded (see above); every other utterance is in the output\n", g_failed_utts);
        return 1;
    }
    return 0;
}
