/*
 * integration/sphinx3/s3amd_flatten.h -- what the sphinx3 side of the drop-in hands to libcmusphinx_amd.so: the
 * reference's own structures FLATTENED into plain arrays (no reference type crosses the C ABI).
 *
 *   flatten_tree   lextree_t (libsearch/lextree.h) -> node arrays + CSR child lists in glist order + the root lists per
 *                  left context (what s3a_lexsearch_init / the bundle take)
 *   flatten_lm     lm_t (liblm/lm.h; in-memory or disk-based DMP) -> sorted unigram / bigram / trigram arrays with the
 *                  prob / back-off indirections resolved (s3a_lm3g_init), + the per-word facts the word level and the
 *                  second pass read from dict_t / fillpen_t / mdef_t
 *   vithist_fill   a finished history table (plain arrays) -> the reference's vithist_t, so that its own
 *                  vithist_utt_end / backtrace / DAG / output code run on it unchanged
 *
 * Included by integration/sphinx3/s3amd_tst.c (the drop-in) and by the test infrastructure's pinning programs
 * (oracle/ref_s3odag_decode.c); static functions, reference headers must already be included.
 */
#ifndef S3AMD_FLATTEN_H
#define S3AMD_FLATTEN_H

/* ------------------------------------------------------------------ */
/* flattening lextree_t                                                */
/* ------------------------------------------------------------------ */
typedef struct {
    int32 n_node;
    lextree_node_t **node;      /* index -> reference node (BFS from lextree->root) */
    int32 *ssid, *tmatid, *wid, *prob, *child_off, *child;
    uint8 *composite, *ci;      /* ci: lextree_node_t.ci, the node's CI phone (phoneme look-ahead) */
    int32 n_lc, *lcroot_off, *lcroot, n_root, *root;
    int16 *lc;
    int32 type;
} flat_t;

static int
cmp_ptr(const void *a, const void *b)
{
    const void *x = *(void *const *)a, *y = *(void *const *)b;
    return (x > y) - (x < y);
}

typedef struct { lextree_node_t *p; int32 idx; } pmap_t;
static __thread pmap_t *g_pmap;
static __thread int32 g_npmap;

static int
cmp_pmap(const void *a, const void *b)
{
    const pmap_t *x = a, *y = b;
    return (x->p > y->p) - (x->p < y->p);
}

static int32
node_index(lextree_node_t *p)
{
    int32 lo = 0, hi = g_npmap - 1;
    while (lo <= hi) {
        int32 m = (lo + hi) / 2;
        if (g_pmap[m].p == p) return g_pmap[m].idx;
        if (g_pmap[m].p < p) lo = m + 1; else hi = m - 1;
    }
    E_FATAL("tst shim: lextree node not found while flattening\n");
    return -1;
}

static flat_t *
flatten_tree(lextree_t *lt)
{
    flat_t *f = ckd_calloc(1, sizeof(*f));
    int32 cap = lt->n_node + 16, n = 0, head = 0, i, j, nchild = 0;
    lextree_node_t **q = ckd_calloc(cap, sizeof(*q));
    gnode_t *gn;
    (void)cmp_ptr;

    /* BFS over roots (glist order) then children (glist order).  Only the children of ROOT
     * nodes can have several parents (one per left-context variant of the first phone,
     * lextree.c:626-660); below that every node has one parent, so no dedupe is needed. */
    for (gn = lt->root; gn; gn = gnode_next(gn)) {
        lextree_node_t *ln = gnode_ptr(gn);
        for (j = 0; j < n && q[j] != ln; j++);
        if (j == n) { if (n >= cap) E_FATAL("flatten: node count exceeds n_node\n"); q[n++] = ln; }
    }
    {
        int32 n_rootnodes = n, lvl1_start = n;
        while (head < n) {
            lextree_node_t *ln = q[head];
            int is_root = head < n_rootnodes;
            head++;
            for (gn = ln->children; gn; gn = gnode_next(gn)) {
                lextree_node_t *c = gnode_ptr(gn);
                nchild++;
                if (is_root) {
                    for (j = lvl1_start; j < n && q[j] != c; j++);
                    if (j < n) continue;
                }
                if (n >= cap) { cap *= 2; q = ckd_realloc(q, cap * sizeof(*q)); }
                q[n++] = c;
            }
        }
    }
    f->n_node = n;
    f->node = q;
    g_pmap = ckd_calloc(n, sizeof(pmap_t));
    g_npmap = n;
    for (i = 0; i < n; i++) { g_pmap[i].p = q[i]; g_pmap[i].idx = i; }
    qsort(g_pmap, n, sizeof(pmap_t), cmp_pmap);

    f->ssid = ckd_calloc(n, 4); f->tmatid = ckd_calloc(n, 4); f->wid = ckd_calloc(n, 4);
    f->prob = ckd_calloc(n, 4); f->composite = ckd_calloc(n, 1); f->ci = ckd_calloc(n, 1);
    f->child_off = ckd_calloc(n + 1, 4); f->child = ckd_calloc(nchild + 1, 4);
    for (i = 0, j = 0; i < n; i++) {
        lextree_node_t *ln = q[i];
        f->ssid[i] = ln->ssid;
        f->tmatid[i] = hmm_tmatid(&ln->hmm);
        f->wid[i] = IS_S3WID(ln->wid) ? ln->wid : -1;
        f->prob[i] = ln->prob;
        f->composite[i] = ln->composite ? 1 : 0;
        f->ci[i] = (uint8)ln->ci;
        f->child_off[i] = j;
        for (gn = ln->children; gn; gn = gnode_next(gn))
            f->child[j++] = node_index(gnode_ptr(gn));
    }
    f->child_off[n] = j;
    {
        int32 mx = 0;
        for (i = 0; i < n; i++)
            if (f->child_off[i + 1] - f->child_off[i] > mx) mx = f->child_off[i + 1] - f->child_off[i];
        E_INFO("tst shim: lextree type %d: %d nodes, %d links, %d roots, largest fan-out %d\n", lt->type, n, j,
               glist_count(lt->root), mx);
    }
    f->n_lc = lt->n_lc;
    f->type = lt->type;
    if (lt->n_lc > 0) {
        int32 tot = 0;
        f->lc = ckd_calloc(lt->n_lc, sizeof(int16));
        f->lcroot_off = ckd_calloc(lt->n_lc + 1, 4);
        for (i = 0; i < lt->n_lc; i++)
            tot += glist_count(lt->lcroot[i].root);
        f->lcroot = ckd_calloc(tot + 1, 4);
        for (i = 0, j = 0; i < lt->n_lc; i++) {
            f->lc[i] = lt->lcroot[i].lc;
            f->lcroot_off[i] = j;
            for (gn = lt->lcroot[i].root; gn; gn = gnode_next(gn))
                f->lcroot[j++] = node_index(gnode_ptr(gn));
        }
        f->lcroot_off[lt->n_lc] = j;
    }
    f->n_root = glist_count(lt->root);
    f->root = ckd_calloc(f->n_root + 1, 4);
    for (gn = lt->root, j = 0; gn; gn = gnode_next(gn))
        f->root[j++] = node_index(gnode_ptr(gn));
    ckd_free(g_pmap);
    g_pmap = NULL;
    return f;
}

/* ------------------------------------------------------------------ */
/* flattening lm_t / dict_t for the word level                         */
/* ------------------------------------------------------------------ */
/* The trigram as plain sorted arrays (what lm_3g_dmp.c's DMP layout already is, with the prob /
 * back-off indirections resolved and the segment-relative firsttg made absolute), and the few
 * per-word facts the word level reads from dict_t / fillpen_t / mdef_t.  "No LM word" = -1. */
typedef struct {
    int32 n_ug, n_bg, n_tg;
    int32 *ug_prob, *ug_bowt, *ug_firstbg, *bg_wid, *bg_prob, *bg_bowt, *bg_firsttg, *tg_wid, *tg_prob;
    int32 *inclass;
    int32 n_word, n_ci, *lwid, *fillpen, *last_ci;
    uint8 *is_filler;
    int32 startwid, finishwid, silwid, start_lwid, finish_lwid;
} wl_flat_t;

static wl_flat_t *
flatten_lm(kbcore_t *kbc)
{
    lm_t *lm = kbcore_lm(kbc);
    dict_t *d = kbcore_dict(kbc);
    mdef_t *mdef = kbcore_mdef(kbc);
    wl_flat_t *f = ckd_calloc(1, sizeof(*f));
    int32 i, w;
    bg_t *bg = NULL; bg32_t *bg32 = NULL; tg_t *tg = NULL; tg32_t *tg32 = NULL;
    int own = 0;

    f->n_ug = lm->n_ug;
    f->n_bg = (lm->ugonly) ? 0 : lm->n_bg;
    f->n_tg = (lm->ugonly || lm->bgonly) ? 0 : lm->n_tg;
    f->ug_prob = ckd_calloc(lm->n_ug + 1, 4); f->ug_bowt = ckd_calloc(lm->n_ug + 1, 4);
    f->ug_firstbg = ckd_calloc(lm->n_ug + 1, 4);
    for (i = 0; i < lm->n_ug; i++) { f->ug_prob[i] = lm->ug[i].prob.l; f->ug_bowt[i] = lm->ug[i].bowt.l; }
    for (i = 0; i <= lm->n_ug; i++) f->ug_firstbg[i] = f->n_bg ? lm->ug[i].firstbg : 0;
    if (f->n_bg) {
        /* the bigram / trigram records: in memory, or (disk-based DMP, lm.c:1073-1127, 1476-1520) read here */
        if (lm->is32bits) { bg32 = lm->bg32; tg32 = lm->tg32; } else { bg = lm->bg; tg = lm->tg; }
        if ((lm->is32bits ? (void *)bg32 : (void *)bg) == NULL) {
            size_t sz = lm->is32bits ? sizeof(bg32_t) : sizeof(bg_t);
            void *buf = ckd_calloc(lm->n_bg + 1, sz);
            own = 1;
            if (!lm->fp || fseek(lm->fp, lm->bgoff, SEEK_SET) < 0 || fread(buf, sz, lm->n_bg + 1, lm->fp) != (size_t)(lm->n_bg + 1))
                E_FATAL("tst shim: cannot read the bigrams of a disk-based LM\n");
            if (lm->is32bits) bg32 = buf; else bg = buf;
            if (lm->byteswap)
                for (i = 0; i <= lm->n_bg; i++) {
                    if (lm->is32bits) { SWAP_INT32(&bg32[i].wid); SWAP_INT32(&bg32[i].probid); SWAP_INT32(&bg32[i].bowtid); SWAP_INT32(&bg32[i].firsttg); }
                    else { SWAP_INT16(&bg[i].wid); SWAP_INT16(&bg[i].probid); SWAP_INT16(&bg[i].bowtid); SWAP_INT16(&bg[i].firsttg); }
                }
            if (f->n_tg) {
                sz = lm->is32bits ? sizeof(tg32_t) : sizeof(tg_t);
                buf = ckd_calloc(lm->n_tg + 1, sz);
                if (fseek(lm->fp, lm->tgoff, SEEK_SET) < 0 || fread(buf, sz, lm->n_tg, lm->fp) != (size_t)lm->n_tg)
                    E_FATAL("tst shim: cannot read the trigrams of a disk-based LM\n");
                if (lm->is32bits) tg32 = buf; else tg = buf;
                if (lm->byteswap)
                    for (i = 0; i < lm->n_tg; i++) {
                        if (lm->is32bits) { SWAP_INT32(&tg32[i].wid); SWAP_INT32(&tg32[i].probid); }
                        else { SWAP_INT16(&tg[i].wid); SWAP_INT16(&tg[i].probid); }
                    }
            }
        }
        f->bg_wid = ckd_calloc(lm->n_bg + 1, 4); f->bg_prob = ckd_calloc(lm->n_bg + 1, 4);
        f->bg_bowt = ckd_calloc(lm->n_bg + 1, 4); f->bg_firsttg = ckd_calloc(lm->n_bg + 1, 4);
        for (i = 0; i < lm->n_bg; i++) {
            f->bg_wid[i] = lm->is32bits ? (int32)bg32[i].wid : (int32)bg[i].wid;
            f->bg_prob[i] = lm->bgprob[lm->is32bits ? bg32[i].probid : bg[i].probid].l;
            if (f->n_tg) f->bg_bowt[i] = lm->tgbowt[lm->is32bits ? bg32[i].bowtid : bg[i].bowtid].l;
        }
        if (f->n_tg) {
            /* load_tg, lm.c:1435-1443: absolute first trigram = tg_segbase[b >> log_bg_seg_sz] + firsttg */
            for (i = 0; i <= lm->n_bg; i++)
                f->bg_firsttg[i] = lm->tg_segbase[i >> lm->log_bg_seg_sz]
                    + (lm->is32bits ? (int32)bg32[i].firsttg : (int32)bg[i].firsttg);
            f->tg_wid = ckd_calloc(lm->n_tg + 1, 4); f->tg_prob = ckd_calloc(lm->n_tg + 1, 4);
            for (i = 0; i < lm->n_tg; i++) {
                f->tg_wid[i] = lm->is32bits ? (int32)tg32[i].wid : (int32)tg[i].wid;
                f->tg_prob[i] = lm->tgprob[lm->is32bits ? tg32[i].probid : tg[i].probid].l;
            }
        }
        if (own) { ckd_free(lm->is32bits ? (void *)bg32 : (void *)bg); ckd_free(lm->is32bits ? (void *)tg32 : (void *)tg); }
    }
    f->n_word = dict_size(d);
    f->n_ci = mdef_n_ciphone(mdef);
    f->lwid = ckd_calloc(f->n_word + 1, 4); f->fillpen = ckd_calloc(f->n_word + 1, 4);
    f->last_ci = ckd_calloc(f->n_word + 1, 4); f->is_filler = ckd_calloc(f->n_word + 1, 1);
    if (lm->inclass_ugscore) {
        f->inclass = ckd_calloc(f->n_word + 1, 4);
        for (w = 0; w < f->n_word; w++) f->inclass[w] = lm->inclass_ugscore[w];
    }
    for (w = 0; w < f->n_word; w++) {
        int32 p = dict_last_phone(d, w);
        f->lwid[w] = IS_LMWID(lm, lm->dict2lmwid[w]) && lm->dict2lmwid[w] < (s3lmwid32_t)lm->n_ug ? (int32)lm->dict2lmwid[w] : -1;
        f->is_filler[w] = dict_filler_word(d, w) ? 1 : 0;
        if (f->is_filler[w]) f->fillpen[w] = fillpen(kbcore_fillpen(kbc), w);
        f->last_ci[w] = mdef_is_fillerphone(mdef, p) ? mdef_silphone(mdef) : p;
    }
    f->startwid = dict_startwid(d); f->finishwid = dict_finishwid(d); f->silwid = dict_silwid(d);
    f->start_lwid = IS_LMWID(lm, lm_startwid(lm)) ? (int32)lm_startwid(lm) : -1;
    f->finish_lwid = IS_LMWID(lm, lm_finishwid(lm)) ? (int32)lm_finishwid(lm) : -1;
    E_INFO("tst shim: LM flattened: %d unigrams, %d bigrams, %d trigrams; %d dictionary words\n",
           f->n_ug, f->n_bg, f->n_tg, f->n_word);
    return f;
}

/* a finished history table -> the reference's vithist_t (which srch_TST_begin left holding the dummy
 * <s> entry 0), so that the reference's own vithist_utt_end, backtrace, DAG and output code run on it
 * unchanged.  Blocks are allocated as vithist_entry_alloc (vithist.c:268-294, static there) does. */
static void
vithist_fill(vithist_t *vh, int32 n_entry, int32 n_frm, const int32 *score, const int32 *pred, const int32 *lw0,
             const int32 *lw1, const int32 *wid, const int32 *sf, const int32 *ef, const int32 *ascr,
             const int32 *lscr, const int32 *type, const int32 *frame_start, const int32 *bestscore,
             const int32 *bestvh, lm_t *lm)
{
    int32 id, f;
    if (n_entry > VITHIST_MAXBLKS * VITHIST_BLKSIZE)
        E_FATAL("Viterbi history array exhausted; increase VITHIST_MAXBLKS\n");
    for (id = 0; id < n_entry; id++) {
        vithist_entry_t *ve;
        if (VITHIST_ID2BLKOFFSET(id) == 0 && vh->entry[VITHIST_ID2BLK(id)] == NULL)
            vh->entry[VITHIST_ID2BLK(id)] = ckd_calloc(VITHIST_BLKSIZE, sizeof(vithist_entry_t));
        ve = vithist_id2entry(vh, id);
        ve->wid = wid[id]; ve->sf = sf[id]; ve->ef = ef[id]; ve->ascr = ascr[id]; ve->lscr = lscr[id];
        ve->path.score = score[id]; ve->path.pred = pred[id]; ve->type = type[id]; ve->valid = 1;
        ve->lmstate.lm3g.lwid[0] = lw0[id] < 0 ? BAD_LMWID(lm) : (s3lmwid32_t)lw0[id];
        ve->lmstate.lm3g.lwid[1] = lw1[id] < 0 ? BAD_LMWID(lm) : (s3lmwid32_t)lw1[id];
        ve->rc = NULL; ve->n_rc = 0;
    }
    vh->n_entry = n_entry;
    vh->n_frm = n_frm;
    for (f = 0; f <= n_frm; f++) { vh->frame_start[f] = frame_start[f]; vh->bestscore[f] = bestscore[f]; vh->bestvh[f] = bestvh[f]; }
}


#endif /* S3AMD_FLATTEN_H */
