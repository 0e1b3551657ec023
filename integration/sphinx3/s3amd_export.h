/*
 * s3amd_export.h -- the decoder bundle writer of integration/sphinx3/s3amd_tst.c (S3A_EXPORT=file).  Included by
 * s3amd_uttmode.h; not a translation unit of its own.
 */
/*
 * S3A_EXPORT=file: everything s3a_uttdec_init takes -- the flattened lextrees, senone sequences, composite
 * senones, transition matrices, the flattened trigram, the dictionary facts, beams and pruning limits, the model
 * file names -- written as tagged records {tag, n, n x int32} (int16 / uint8 arrays widened; strings as bytes in
 * int32 cells).  cmusphinx_amd/bundle.py rebuilds the decoder from it through the C ABI alone, so a measurement
 * or a multi-GPU driver needs this program only ONCE, for loading (kb_init) -- never inside a timed region.
 */
static FILE *g_xfp;
static void
xw(int32 tag, int32 n, const void *d)
{
    fwrite(&tag, 4, 1, g_xfp); fwrite(&n, 4, 1, g_xfp);
    if (n) fwrite(d, 4, n, g_xfp);
}
static void
xw16(int32 tag, int32 n, const int16 *d)
{
    int32 i, *w = ckd_calloc(n + 1, 4);
    for (i = 0; i < n; i++) w[i] = d[i];
    xw(tag, n, w);
    ckd_free(w);
}
static void
xw8(int32 tag, int32 n, const uint8 *d)
{
    int32 i, *w = ckd_calloc(n + 1, 4);
    for (i = 0; i < n; i++) w[i] = d[i];
    xw(tag, n, w);
    ckd_free(w);
}
static void
xwstr(int32 tag, const char *str)
{
    int32 n = (int32)strlen(str) + 1, cells = (n + 3) / 4;
    char *b = ckd_calloc(cells + 1, 4);
    memcpy(b, str, n);
    xw(tag, cells, b);
    ckd_free(b);
}

static void
export_bundle(const char *path, kb_t *kb, srch_TST_graph_t *tstg, wl_flat_t *w, const s3a_wordlevel_cfg_t *cfg)
{
    kbcore_t *kbc = kb->kbcore;
    cmd_ln_t *config = kbcore_config(kbc);
    mdef_t *mdef = kbcore_mdef(kbc);
    dict_t *d = kbcore_dict(kbc);
    dict2pid_t *d2p = kbcore_dict2pid(kbc);
    tmat_t *tmat = kbcore_tmat(kbc);
    int32 ne = mdef_n_emit_state(mdef), t, i;
    if ((g_xfp = fopen(path, "wb")) == NULL) E_FATAL("cannot write %s\n", path);
    {
        int32 h[10] = { g_ntree, ne, tmat->n_tmat, mdef_n_sseq(mdef), d2p->n_comsseq, g_n_comstate, mdef_n_sen(mdef),
                        mdef->n_ci_sen, mdef_n_ciphone(mdef), kbcore_fcb(kbc)->stream_len[0] };
        xw(1, 10, h);
    }
    xw(2, tmat->n_tmat * ne * (ne + 1), g_tp_flat); xw16(3, mdef_n_sseq(mdef) * ne, g_sseq_flat);
    xw16(4, d2p->n_comsseq * ne, g_comsseq_flat); xw(5, g_n_comstate + 1, g_comstate_off);
    xw16(6, g_comstate_off[g_n_comstate], g_comstate); xw(7, g_n_comstate, d2p->comwt);
    xw16(8, mdef_n_sen(mdef), mdef->cd2cisen);
    for (t = 0; t < g_ntree; t++) {
        flat_t *f = g_flat[t];
        int32 h2[4] = { f->n_node, f->n_lc, f->n_root, f->type };
        xw(10, 4, h2); xw(11, f->n_node, f->ssid); xw(12, f->n_node, f->tmatid); xw8(13, f->n_node, f->composite);
        xw(14, f->n_node, f->wid); xw(15, f->n_node, f->prob); xw(16, f->n_node + 1, f->child_off);
        xw(17, f->child_off[f->n_node], f->child);
        if (f->n_lc) { xw16(18, f->n_lc, f->lc); xw(19, f->n_lc + 1, f->lcroot_off); xw(20, f->lcroot_off[f->n_lc], f->lcroot); }
        xw(21, f->n_root, f->root); xw8(22, f->n_node, f->ci);
    }
    {
        int32 h[3] = { w->n_ug, w->n_bg, w->n_tg };
        xw(30, 3, h);
        xw(31, w->n_ug, w->ug_prob); xw(32, w->n_ug, w->ug_bowt); xw(33, w->n_ug + 1, w->ug_firstbg);
        xw(34, w->n_bg, w->bg_wid); xw(35, w->n_bg, w->bg_prob); xw(36, w->n_tg ? w->n_bg : 0, w->bg_bowt);
        xw(37, w->n_tg ? w->n_bg + 1 : 0, w->bg_firsttg); xw(38, w->n_tg, w->tg_wid); xw(39, w->n_tg, w->tg_prob);
    }
    {
        int32 h[7] = { w->n_word, w->startwid, w->finishwid, w->silwid, w->start_lwid, w->finish_lwid, cfg->sil_ci };
        int32 *base = ckd_calloc(w->n_word + 1, 4), *lmraw = ckd_calloc(2, 4);
        size_t tot = 0;
        char *strs, *q;
        xw(40, 7, h); xw(41, w->n_word, w->lwid); xw8(42, w->n_word, w->is_filler); xw(43, w->n_word, w->fillpen);
        xw(44, w->n_word, w->last_ci);
        for (i = 0; i < w->n_word; i++) { base[i] = dict_basewid(d, i); tot += strlen(dict_wordstr(d, i)) + 1; }
        xw(46, w->n_word, base);
        q = strs = ckd_calloc(tot + 8, 1);
        for (i = 0; i < w->n_word; i++) { strcpy(q, dict_wordstr(d, i)); q += strlen(q) + 1; }
        xw(45, (int32)((tot + 3) / 4), strs);
        ckd_free(strs); ckd_free(base); ckd_free(lmraw);
    }
    {
        int32 c[16] = { cfg->wbeam_vh, cfg->bghist, cfg->maxwpf, cfg->maxhistpf, cfg->wordend_beam, cfg->n_lextree, cfg->epl,
                        cfg->hmmbeam, cfg->pbeam, cfg->wbeam, cfg->ptranskip, cfg->maxhmmpf, cmd_ln_int32_r(config, "-ds"),
                        cmd_ln_int32_r(config, "-cond_ds"), cmd_ln_int32_r(config, "-maxcdsenpf"),
                        cmd_ln_int32_r(config, "-hypsegscore_unscale") };
        double dd[8] = { cmd_ln_float64_r(config, "-logbase"), cmd_ln_float32_r(config, "-varfloor"),
                         cmd_ln_float32_r(config, "-mixwfloor"), cmd_ln_float64_r(config, "-ci_pbeam"),
                         cmd_ln_float32_r(config, "-tighten_factor"), (double)kbcore_lm(kbc)->lw,
                         (double)kbcore_lm(kbc)->wip, (double)cmd_ln_float32_r(config, "-bestpathlw") };
        int32 dg[6] = { cmd_ln_int32_r(config, "-min_endfr"), cmd_ln_int32_r(config, "-maxedge"), cmd_ln_int32_r(config, "-maxlmop"),
                        cmd_ln_int32_r(config, "-maxlpf"), logs3(kbcore_logmath(kbc), kbcore_fillpen(kbc)->wip),
                        cmd_ln_boolean_r(config, "-bestpath") ? 1 : 0 };
        xw(50, 16, c); xw(51, 16, dd); xw(55, 6, dg);
        {
            int32 ph[3] = { kb->pl->pheurtype, kb->pl->pl_beam, cmd_ln_int32_r(config, "-pl_window") };
            xw(56, 3, ph); xw16(57, mdef->n_ci_sen + 1, mdef->sen2cimap);
        }
        xwstr(52, cmd_ln_str_r(config, "-mean")); xwstr(53, cmd_ln_str_r(config, "-var")); xwstr(54, cmd_ln_str_r(config, "-mixw"));
    }
    fclose(g_xfp);
    E_INFO("tst shim: decoder bundle written to %s\n", path);
}

