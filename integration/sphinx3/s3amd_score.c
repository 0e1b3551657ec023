/*
 * oracle/ref_s3amd_decode.c -- TEST INFRASTRUCTURE: the drop-in demonstrated.
 *
 * sphinx3_decode with the MI355X backend plugged in at the reference's own
 * plugin boundary, the srch_funcs_t operator table (sphinx3/include/srch.h:528-701).
 * Everything above the table is the UNMODIFIED reference (oracle/_ref/libs3ref.so:
 * kb/kbcore init, dictionary, LM, lextree/FSG search, vithist, utt_decode,
 * hypothesis output); the three scoring slots are replaced by calls into
 * libcmusphinx_amd.so through its C ABI (include/cmusphinx_amd.h):
 *
 *   gmm_compute_lv1   approx_ci_gmm_compute        (libsearch/gmm_wrap.c:170-211)
 *                       -> s3a_approx_cont_mgau_ci_eval
 *   gmm_compute_lv2   s3_cd_gmm_compute_sen[_comp] (libsearch/gmm_wrap.c:80-167)
 *                       -> s3a_approx_cont_mgau_frame_eval (+ s3a_dict2pid_comsenscr)
 *   utt_begin         srch_{TST,FSG}_begin          + s3a_scorer_utt_begin
 *
 * This file is exactly the shim INTEGRATION.md asks a sphinx3 maintainer to add.
 * The reference's own main_decode.c is #included in place (not copied) so the
 * command-line definition table is the reference's by construction; its main()
 * is renamed and unused.  Built only where /root/reference exists; the binary
 * travels to the GPU box in oracle/_ref/ and tests/test_gpu_dropin.py diffs its
 * hypotheses against the reference's golden results and a live CPU run.
 */
#define main sphinx3_decode_reference_main
#include "main_decode.c"        /* found via -I$(S3)/src/programs: the reference's file, in place */
#undef main

#include <string.h>
#include "srch.h"
#include "gmm_wrap.h"
#include "dict2pid.h"
#include "cmusphinx_amd.h"

static s3a_logmath_t *g_lm;
static s3a_mgau_model_t *g_gm;
static s3a_scorer_t *g_sc;
static s3a_comsen_t *g_cs;
static s3a_ms_mgau_t *g_ms;         /* -senmgau .s3cont. / .semi.: the multi-stream scorer instead */
static int (*g_ref_utt_begin)(void *);
static int32 g_n_sen, g_n_ci_sen;
static long g_lv1_calls, g_lv2_calls;

static void
die(const char *what)
{
    E_FATAL("s3amd shim: %s: %s\n", what, s3a_last_error());
}

/* slot gmm_compute_lv1: int (*)(void *srch, float32 *feat, int32 cache_idx, int32 wav_idx) */
static int
s3amd_gmm_compute_lv1(void *srch, float32 *feat, int32 cache_idx, int32 wav_idx)
{
    srch_t *s = (srch_t *)srch;
    ascr_t *ascr = s->ascr;
    if (s3a_approx_cont_mgau_ci_eval(g_sc, feat, ascr->cache_ci_senscr[cache_idx],
                                     &ascr->cache_best_list[cache_idx], wav_idx) != S3A_OK)
        die("gmm_compute_lv1");
    s->stat->utt_cisen_eval += g_n_ci_sen;
    g_lv1_calls++;
    return SRCH_SUCCESS;
}

/* slot gmm_compute_lv2: int (*)(void *srch, float32 **feat, int32 time) */
static int
s3amd_gmm_compute_lv2(void *srch, float32 **feat, int32 wav_idx)
{
    srch_t *s = (srch_t *)srch;
    ascr_t *ascr = s->ascr;
    int32 best, ns, ng;
    if (s3a_approx_cont_mgau_frame_eval(g_sc, ascr->sen_active, ascr->rec_sen_active,
                                        ascr->senscr, feat[0], wav_idx,
                                        ascr->cache_ci_senscr[s->cache_win_strt], &best, &ns,
                                        &ng) != S3A_OK)
        die("gmm_compute_lv2");
    s->senscale = best;
    s->stat->utt_sen_eval += ns;
    s->stat->utt_gau_eval += ng;
    if (g_cs && s3a_dict2pid_comsenscr(g_cs, ascr->senscr, g_n_sen, ascr->comsen) != S3A_OK)
        die("dict2pid_comsenscr");
    g_lv2_calls++;
    return SRCH_SUCCESS;
}

/* slot gmm_compute_lv2 when kbcore holds an ms_mgau (gmm_wrap.c:136-140 -> ms_cont_mgau_frame_eval) */
static int
s3amd_ms_gmm_compute_lv2(void *srch, float32 **feat, int32 wav_idx)
{
    srch_t *s = (srch_t *)srch;
    ascr_t *ascr = s->ascr;
    feat_t *fcb = kbcore_fcb(s->kbc);
    static float32 *cat;
    float32 *x = feat[0];
    int32 best, f, o;
    if (feat_n_stream(fcb) > 1) {           /* streams are separate rows of feat[][]: concatenate */
        if (!cat) cat = ckd_calloc(s3a_ms_mgau_veclen(g_ms), sizeof(float32));
        for (f = 0, o = 0; f < feat_n_stream(fcb); o += feat_stream_len(fcb, f), f++)
            memcpy(cat + o, feat[f], feat_stream_len(fcb, f) * sizeof(float32));
        x = cat;
    }
    if (s3a_ms_cont_mgau_frame_eval(g_ms, ascr->sen_active, ascr->senscr, x, wav_idx, &best) != S3A_OK)
        die("ms gmm_compute_lv2");
    s->senscale = best;
    if (g_cs && s3a_dict2pid_comsenscr(g_cs, ascr->senscr, g_n_sen, ascr->comsen) != S3A_OK)
        die("dict2pid_comsenscr");
    g_lv2_calls++;
    return SRCH_SUCCESS;
}

static int
s3amd_utt_begin(void *srch)
{
    if (s3a_scorer_utt_begin(g_sc) != S3A_OK)
        die("utt_begin");
    return g_ref_utt_begin(srch);
}

static void
s3amd_install(kb_t *kb)
{
    cmd_ln_t *config = kbcore_config(kb->kbcore);
    mdef_t *mdef = kbcore_mdef(kb->kbcore);
    dict2pid_t *d2p = kbcore_dict2pid(kb->kbcore);
    srch_t *s = (srch_t *)kb->srch;
    int composite_pass;

    if (kbcore_mgau(kb->kbcore) == NULL && kbcore_ms_mgau(kb->kbcore) == NULL)
        E_FATAL("s3amd shim: only -senmgau .cont. / .s3cont. / .semi. models are supported\n");
    if (kbcore_svq(kb->kbcore) || kbcore_gs(kb->kbcore))
        E_FATAL("s3amd shim: sub-VQ / Gaussian selection are not supported (the GPU scores every component)\n");
    if (s3a_device_count() < 1)
        E_FATAL("s3amd shim: no GPU; libcmusphinx_amd has no CPU fallback\n");

    g_lm = s3a_logs3_init(cmd_ln_float64_r(config, "-logbase"), 0, 1);
    g_n_sen = mdef_n_sen(mdef);
    g_n_ci_sen = mdef->n_ci_sen;
    if (kbcore_ms_mgau(kb->kbcore)) {
        /* the same values ms_mgau_init receives in s3_am_init (kbcore.c:342-364) */
        g_ms = s3a_ms_mgau_init(cmd_ln_str_r(config, "-mean"), cmd_ln_str_r(config, "-var"),
                                cmd_ln_float32_r(config, "-varfloor"), cmd_ln_str_r(config, "-mixw"),
                                cmd_ln_float32_r(config, "-mixwfloor"), 1, cmd_ln_str_r(config, "-senmgau"),
                                cmd_ln_exists_r(config, "-lambda") ? cmd_ln_str_r(config, "-lambda") : NULL,
                                cmd_ln_int32_r(config, "-topn"), g_lm);
        if (!g_ms) die("s3a_ms_mgau_init");
        goto composite;
    }
    g_gm = s3a_mgau_init(cmd_ln_str_r(config, "-mean"), cmd_ln_str_r(config, "-var"),
                         cmd_ln_float32_r(config, "-varfloor"), cmd_ln_str_r(config, "-mixw"),
                         cmd_ln_float32_r(config, "-mixwfloor"), 1, ".cont.",
                         S3A_MIX_INT_FLOAT_COMP, g_lm);
    if (!g_gm) die("s3a_mgau_init");
    /* the same values fast_gmm_init receives in kb_init (kb.c:218-231) */
    g_sc = s3a_scorer_init(g_gm, mdef->cd2cisen, g_n_sen, g_n_ci_sen,
                           cmd_ln_int32_r(config, "-ds"), cmd_ln_int32_r(config, "-cond_ds"),
                           cmd_ln_float64_r(config, "-ci_pbeam"),
                           cmd_ln_float32_r(config, "-tighten_factor"),
                           cmd_ln_int32_r(config, "-maxcdsenpf"));
    if (!g_sc) die("s3a_scorer_init");

composite:
    /* composite senones only where the reference's table computes them */
    composite_pass = (s->funcs->gmm_compute_lv2 == s3_cd_gmm_compute_sen_comp);
    if (composite_pass && d2p && d2p->n_comstate > 0) {
        int32 i, j, n = 0, *off, *wt;
        s3senid_t *lst;
        for (i = 0; i < d2p->n_comstate; i++)
            for (j = 0; IS_S3SENID(d2p->comstate[i][j]); j++)
                n++;
        off = ckd_calloc(d2p->n_comstate + 1, sizeof(int32));
        wt = ckd_calloc(d2p->n_comstate, sizeof(int32));
        lst = ckd_calloc(n, sizeof(s3senid_t));
        for (i = 0, n = 0; i < d2p->n_comstate; i++) {
            off[i] = n;
            for (j = 0; IS_S3SENID(d2p->comstate[i][j]); j++)
                lst[n++] = d2p->comstate[i][j];
            wt[i] = d2p->comwt[i];
        }
        off[d2p->n_comstate] = n;
        g_cs = s3a_comsen_init(d2p->n_comstate, off, lst, wt);
        if (!g_cs) die("s3a_comsen_init");
        ckd_free(off); ckd_free(wt); ckd_free(lst);
    }

    /* the three slots */
    if (g_ms) {
        /* gmm_compute_lv1 stays the reference's: it does nothing for multi-stream models (gmm_wrap.c:193-197) */
        s->funcs->gmm_compute_lv2 = s3amd_ms_gmm_compute_lv2;
    }
    else {
        g_ref_utt_begin = s->funcs->utt_begin;
        s->funcs->utt_begin = s3amd_utt_begin;
        s->funcs->gmm_compute_lv1 = s3amd_gmm_compute_lv1;
        s->funcs->gmm_compute_lv2 = s3amd_gmm_compute_lv2;
    }
    E_INFO("s3amd shim installed: %s, %d senones (%d CI), composite pass %s\n", s3a_version(),
           g_n_sen, g_n_ci_sen, g_cs ? "on device" : "none");
}

int
main(int argc, char *argv[])
{
    kb_t kb;
    cmd_ln_t *config;

    cmd_ln_appl_enter(argc, argv, "default.arg", arg);      /* `arg`: the reference's own table */
    unlimit();
    config = cmd_ln_get();
    kb_init(&kb, config);
    s3amd_install(&kb);

    if (!cmd_ln_str_r(config, "-ctl"))
        E_FATAL("-ctl is required\n");
    kb.stat->tm = ctl_process(cmd_ln_str_r(config, "-ctl"), cmd_ln_str_r(config, "-ctl_lm"),
                              cmd_ln_str_r(config, "-ctl_mllr"),
                              cmd_ln_int32_r(config, "-ctloffset"),
                              cmd_ln_int32_r(config, "-ctlcount"), utt_decode, &kb);
    if (kb.matchsegfp) fclose(kb.matchsegfp);
    if (kb.matchfp) fclose(kb.matchfp);
    stat_report_corpus(kb.stat);
    E_INFO("s3amd shim: %ld gmm_compute_lv1 and %ld gmm_compute_lv2 calls served by the GPU\n",
           g_lv1_calls, g_lv2_calls);
    if (g_lv2_calls == 0)
        E_FATAL("s3amd shim: the GPU scoring slots were never called\n");
    s3a_comsen_free(g_cs);
    s3a_ms_mgau_free(g_ms);
    s3a_scorer_free(g_sc);
    s3a_mgau_free(g_gm);
    s3a_logmath_free(g_lm);
    return 0;
}
