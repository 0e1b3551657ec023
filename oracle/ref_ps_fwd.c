/*
 * oracle/ref_ps_fwd.c -- TEST INFRASTRUCTURE (ours): a small batch driver over the unmodified pocketsphinx
 * (oracle/_ref/libpsref.so), in the shape of pocketsphinx_batch (pocketsphinx/src/programs/batch.c), built
 * three ways by oracle/Makefile:
 *
 *   ref_ps_fwd           the unmodified decoder (the truth)
 *   ref_ps_ofwd          -DPS_BACKEND_ORACLE: the first pass served by oracle/s3o_psfwd.c behind
 *                        ps_searchfuncs_t {start, step, finish} (pins the restatement)
 *   ref_ps_amdfwd        -DPS_BACKEND_AMD: the first pass served by libcmusphinx_amd through
 *                        integration/pocketsphinx/ps_search_amd.c (the drop-in, as a test)
 *
 *   ref_ps_fwd [pocketsphinx options] -ctl CTL -cepdir DIR [-cepext .mfc] [-adcin yes] -hyp OUT
 *              [-hypseg OUT] [-bpdump OUT] [-fresh yes] [-batch N]
 *
 * -hyp lines are pocketsphinx_batch's ("<hyp> (<uttid> <score>)", batch.c:741-745); -hypseg lines list the
 * segment iterator's (word sf ef ascr lscr); -bpdump writes, per utterance, the backpointer table as int32
 * records (read by tests/psfwd_dump.py); -fresh yes makes a new ps_decoder_t for every utterance (a decoder's
 * channels keep state from one utterance to the next, see s3o_psfwd_reset); -batch N (AMD only) decodes N
 * utterances at a time as lanes of the device engine (whole utterances on the device, cepstra input only).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sphinxbase/ckd_alloc.h>
#include <sphinxbase/cmd_ln.h>
#include <sphinxbase/err.h>
#include <sphinxbase/feat.h>
#include "pocketsphinx.h"
#include "cmdln_macro.h"
#include "pocketsphinx_internal.h"
#include "ngram_search.h"

#if defined(PS_BACKEND_ORACLE)
int ps_oracle_search_install(ps_decoder_t *ps);
#define INSTALL(ps) ps_oracle_search_install(ps)
#elif defined(PS_BACKEND_AMD)
int ps_amd_search_install(ps_decoder_t *ps, int n_lanes);
int ps_amd_decode_cep_batch(ps_decoder_t *ps, int n_utt, mfcc_t ***cep, const int *n_frames, int fresh,
                            char **out_hyp, int32 *out_score, FILE *segfh, char **uttids, FILE *bpfh);
#define INSTALL(ps) ps_amd_search_install(ps, g_batch > 0 ? g_batch : 1)
#else
#define INSTALL(ps) 0
#endif

/* -trace FILE: after every frame of the first pass, the search's scalars, next active word list and last-phone
 * candidates as text (the unmodified decoder's step slot is wrapped, not changed; the oracle binding writes the
 * same lines from its own state: ps_search_oracle.c) */
FILE *g_trace;
#if !defined(PS_BACKEND_ORACLE) && !defined(PS_BACKEND_AMD)
static ps_searchfuncs_t g_wrapped, *g_orig_vt;
static int
traced_step(ps_search_t *search, int frame_idx)
{
    ngram_search_t *ngs = (ngram_search_t *)search;
    int rv = g_orig_vt->step(search, frame_idx), nf = frame_idx + 1, i;
    fprintf(g_trace, "F %d rv %d best %d lpbest %d dyn %d bpidx %d nacl %d nawl %d ncand %d\n", frame_idx, rv, ngs->best_score,
            ngs->last_phone_best_score, ngs->dynamic_beam, ngs->bpidx, ngs->n_active_chan[nf & 1], ngs->n_active_word[nf & 1],
            ngs->n_lastphn_cand);
    fprintf(g_trace, "W");
    for (i = 0; i < ngs->n_active_word[nf & 1]; i++) fprintf(g_trace, " %d", ngs->active_word_list[nf & 1][i]);
    fprintf(g_trace, "\nC");
    for (i = 0; i < ngs->n_lastphn_cand; i++)
        fprintf(g_trace, " %d:%d:%d", ngs->lastphn_cand[i].wid, ngs->lastphn_cand[i].score, ngs->lastphn_cand[i].bp);
    fprintf(g_trace, "\n");
    return rv;
}
static int
install_trace(ps_decoder_t *ps)
{
    if (!g_trace) return 0;
    g_orig_vt = ps->search->vt;
    g_wrapped = *g_orig_vt;
    g_wrapped.step = traced_step;
    ps->search->vt = &g_wrapped;
    return 0;
}
#undef INSTALL
#define INSTALL(ps) install_trace(ps)
#endif

static const arg_t defn[] = {
    POCKETSPHINX_OPTIONS,
    { "-ctl", ARG_STRING, NULL, "Control file" },
    { "-cepdir", ARG_STRING, NULL, "Input directory" },
    { "-cepext", ARG_STRING, ".mfc", "Input extension" },
    { "-adcin", ARG_BOOLEAN, "no", "Input is raw audio" },
    { "-hyp", ARG_STRING, NULL, "Hypothesis output" },
    { "-hypseg", ARG_STRING, NULL, "Segmentation output" },
    { "-bpdump", ARG_STRING, NULL, "Backpointer table dump" },
    { "-fresh", ARG_BOOLEAN, "no", "A new decoder for every utterance" },
    { "-batch", ARG_INT32, "0", "Utterances per device batch (AMD backend)" },
    { "-trace", ARG_STRING, NULL, "Per-frame trace of the first pass" },
    { "-queue", ARG_BOOLEAN, "no", "AMD backend: the whole control file as ONE queue over the -batch lanes" },
    { "-partial", ARG_INT32, "0", "Ask for a partial hypothesis every N frames (written to <-hyp>.partial)" },
    CMDLN_EMPTY_OPTION
};
static int g_batch;

static mfcc_t **
read_mfc(const char *path, int32 *nfr, int ceplen)
{
    FILE *fp = fopen(path, "rb");
    int32 n, i, swap = 0;
    long sz;
    float32 *buf;
    mfcc_t **cep;
    if (!fp) E_FATAL("cannot open %s\n", path);
    fseek(fp, 0, SEEK_END); sz = ftell(fp); fseek(fp, 0, SEEK_SET);
    if (fread(&n, 4, 1, fp) != 1) E_FATAL("%s: empty\n", path);
    if ((long)n * 4 + 4 != sz) { n = (int32)__builtin_bswap32((uint32)n); swap = 1; }
    if ((long)n * 4 + 4 != sz) E_FATAL("%s: header does not match the file size\n", path);
    buf = ckd_calloc(n, 4);
    if (fread(buf, 4, n, fp) != (size_t)n) E_FATAL("%s: short read\n", path);
    fclose(fp);
    if (swap) for (i = 0; i < n; i++) { uint32 *w = (uint32 *)&buf[i]; *w = __builtin_bswap32(*w); }
    *nfr = n / ceplen;
    cep = (mfcc_t **)ckd_calloc_2d(*nfr, ceplen, sizeof(mfcc_t));
    memcpy(cep[0], buf, (size_t)(*nfr) * ceplen * 4);
    ckd_free(buf);
    return cep;
}

static void
dump_bptable(FILE *fh, ps_decoder_t *ps, const char *uttid)
{
    ngram_search_t *ngs = (ngram_search_t *)ps->search;
    int32 hdr[16], i, len = (int32)strlen(uttid);
    memset(hdr, 0, sizeof hdr);
    hdr[0] = 0x50534250; hdr[1] = len; hdr[2] = ngs->n_frame; hdr[3] = ngs->bpidx; hdr[4] = ngs->bss_head;
    hdr[5] = ngs->best_score; hdr[6] = ngs->last_phone_best_score; hdr[7] = ngs->renormalized;
    hdr[8] = ngs->st.n_root_chan_eval; hdr[9] = ngs->st.n_nonroot_chan_eval; hdr[10] = ngs->st.n_last_chan_eval;
    hdr[11] = ngs->st.n_word_lastchan_eval; hdr[12] = ngs->st.n_lastphn_cand_utt; hdr[13] = ngs->st.n_senone_active_utt;
    hdr[14] = ps->acmod->output_frame;
    fwrite(hdr, 4, 16, fh);
    fwrite(uttid, 1, len, fh);
    for (i = 0; i < ngs->bpidx; i++) {
        bptbl_t *b = &ngs->bp_table[i];
        int32 rec[7] = { b->frame, b->wid, b->bp, b->score, b->s_idx, b->real_wid, b->valid };
        fwrite(rec, 4, 7, fh);
    }
    fwrite(ngs->bscore_stack, 4, ngs->bss_head, fh);
    fwrite(ngs->bp_table_idx, 4, ps->acmod->output_frame + 1, fh);
}

static void
write_seg(FILE *fh, ps_decoder_t *ps, const char *uttid)
{
    int32 score;
    ps_seg_t *seg;
    fprintf(fh, "%s", uttid);
    for (seg = ps_seg_iter(ps, &score); seg; seg = ps_seg_next(seg)) {
        int sf, ef;
        int32 ascr, lscr, lback;
        ps_seg_frames(seg, &sf, &ef);
        ps_seg_prob(seg, &ascr, &lscr, &lback);
        fprintf(fh, " %s %d %d %d %d", ps_seg_word(seg), sf, ef, ascr, lscr);
    }
    fprintf(fh, "\n");
}

int
main(int argc, char **argv)
{
    cmd_ln_t *config;
    ps_decoder_t *ps;
    FILE *ctl, *out, *segfh = NULL, *bpfh = NULL, *partfh = NULL;
    int32 partial = 0;
    char line[4096], path[4096];
    int fresh, adcin;
    if ((config = cmd_ln_parse_r(NULL, defn, argc, argv, TRUE)) == NULL) return 2;
    fresh = cmd_ln_boolean_r(config, "-fresh");
    adcin = cmd_ln_boolean_r(config, "-adcin");
    g_batch = cmd_ln_int32_r(config, "-batch");
    if (cmd_ln_str_r(config, "-trace")) g_trace = fopen(cmd_ln_str_r(config, "-trace"), "w");
    if ((ps = ps_init(config)) == NULL) E_FATAL("ps_init failed\n");
    if (INSTALL(ps) < 0) E_FATAL("backend install failed\n");
    if ((ctl = fopen(cmd_ln_str_r(config, "-ctl"), "r")) == NULL || (out = fopen(cmd_ln_str_r(config, "-hyp"), "w")) == NULL)
        E_FATAL("ctl/hyp\n");
    if (cmd_ln_str_r(config, "-hypseg")) segfh = fopen(cmd_ln_str_r(config, "-hypseg"), "w");
    if (cmd_ln_str_r(config, "-bpdump")) bpfh = fopen(cmd_ln_str_r(config, "-bpdump"), "wb");
    partial = cmd_ln_int32_r(config, "-partial");
    if (partial > 0) { char pp[4200]; snprintf(pp, sizeof pp, "%s.partial", cmd_ln_str_r(config, "-hyp")); partfh = fopen(pp, "w"); }
#if defined(PS_BACKEND_AMD)
    if (g_batch > 0) {
        /* whole utterances on the device, g_batch lanes at a time */
        int ceplen = feat_cepsize(ps->acmod->fcb), n = 0, i, per_call = g_batch;
        if (cmd_ln_boolean_r(config, "-queue")) {       /* every utterance of the control file in one call */
            per_call = 0;
            while (fgets(line, sizeof line, ctl)) per_call++;
            rewind(ctl);
            if (per_call < 1) per_call = 1;
        }
        mfcc_t ***cep = ckd_calloc(per_call, sizeof(*cep));
        int *nfr = ckd_calloc(per_call, sizeof(int));
        char **ids = ckd_calloc(per_call, sizeof(char *)), **hyps = ckd_calloc(per_call, sizeof(char *));
        int32 *scores = ckd_calloc(per_call, sizeof(int32));
        int more = 1;
        if (adcin) E_FATAL("-batch takes cepstrum files\n");
        while (more) {
            char uttid[1024];
            more = fgets(line, sizeof line, ctl) != NULL;
            if (more && sscanf(line, "%1023s", uttid) == 1) {
                int32 k;
                snprintf(path, sizeof path, "%s/%s%s", cmd_ln_str_r(config, "-cepdir"), uttid, cmd_ln_str_r(config, "-cepext"));
                cep[n] = read_mfc(path, &k, ceplen);
                nfr[n] = k;
                ids[n++] = ckd_salloc(uttid);
            }
            if (n == per_call || (!more && n > 0)) {
                if (ps_amd_decode_cep_batch(ps, n, cep, nfr, fresh, hyps, scores, segfh, ids, bpfh) < 0) E_FATAL("batch decode failed\n");
                for (i = 0; i < n; i++) {
                    fprintf(out, "%s (%s %d)\n", hyps[i] ? hyps[i] : "", ids[i], scores[i]);
                    ckd_free(hyps[i]); ckd_free(ids[i]); ckd_free_2d((void **)cep[i]);
                }
                n = 0;
            }
        }
        fclose(out);
        if (segfh) fclose(segfh);
        if (bpfh) fclose(bpfh);
        ps_free(ps);
        return 0;
    }
#endif
    {
    struct timespec ts0, ts1;
    long tot_frames = 0;
    double t_decode = 0.0;
    while (fgets(line, sizeof line, ctl)) {
        char uttid[1024];
        const char *hyp, *id;
        int32 score;
        if (sscanf(line, "%1023s", uttid) != 1) continue;
        if (fresh) {
            cmd_ln_retain(config);
            ps_free(ps);
            if ((ps = ps_init(config)) == NULL) E_FATAL("ps_init failed\n");
            if (INSTALL(ps) < 0) E_FATAL("backend install failed\n");
        }
        snprintf(path, sizeof path, "%s/%s%s", cmd_ln_str_r(config, "-cepdir"), uttid, cmd_ln_str_r(config, "-cepext"));
        if (adcin) {
            FILE *fh = fopen(path, "rb");
            if (!fh) E_FATAL("cannot open %s\n", path);
            clock_gettime(CLOCK_MONOTONIC, &ts0);
            ps_decode_raw(ps, fh, uttid, -1);
            clock_gettime(CLOCK_MONOTONIC, &ts1);
            fclose(fh);
        }
        else {
            int32 nfr;
            mfcc_t **cep = read_mfc(path, &nfr, feat_cepsize(ps->acmod->fcb));
            clock_gettime(CLOCK_MONOTONIC, &ts0);
            ps_start_utt(ps, uttid);
            if (partial > 0 && partfh) {
                /* live use: the cepstra in blocks, a partial result between them (ps_get_hyp while the utterance is open) */
                int32 at;
                for (at = 0; at < nfr; at += partial) {
                    int32 sc;
                    const char *ph, *pid;
                    ps_process_cep(ps, cep + at, nfr - at < partial ? nfr - at : partial, FALSE, FALSE);
                    ph = ps_get_hyp(ps, &sc, &pid);
                    fprintf(partfh, "%s %d: %s (%d)\n", uttid, ps->acmod->output_frame, ph ? ph : "", sc);
                }
            }
            else
            ps_process_cep(ps, cep, nfr, FALSE, TRUE);
            ps_end_utt(ps);
            clock_gettime(CLOCK_MONOTONIC, &ts1);
            ckd_free_2d((void **)cep);
        }
        t_decode += (ts1.tv_sec - ts0.tv_sec) + 1e-9 * (ts1.tv_nsec - ts0.tv_nsec);
        tot_frames += ps->acmod->output_frame;
        if (bpfh) dump_bptable(bpfh, ps, uttid);       /* before ps_get_hyp: a bestpath pass does not touch it either */
        hyp = ps_get_hyp(ps, &score, &id);
        fprintf(out, "%s (%s %d)\n", hyp ? hyp : "", uttid, score);
        if (segfh) write_seg(segfh, ps, uttid);
    }
    E_INFO("decoded %ld frames in %.3f s (%.1f frames/s; features, scoring and search of one utterance at a time, model loading excluded)\n",
           tot_frames, t_decode, t_decode > 0 ? tot_frames / t_decode : 0.0);
    }
    fclose(out);
    if (segfh) fclose(segfh);
    if (bpfh) fclose(bpfh);
    if (partfh) fclose(partfh);
    ps_free(ps);
    return 0;
}
