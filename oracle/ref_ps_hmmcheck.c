/*
 * oracle/ref_ps_hmmcheck.c -- TEST INFRASTRUCTURE (ours): pins oracle/s3o_psfwd.c's hmm_vit_eval restatement on the
 * unmodified pocketsphinx's (oracle/_ref/libpsref.so, hmm.c:789): random 3- and 5-state HMMs, plain and multiplexed,
 * dead states, BAD_SSID states, skip arcs present / absent, through both; every field must come back equal.
 *   ref_ps_hmmcheck N_CASES  -> prints "hmm check: N cases identical" or the first difference, exit code 1
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sphinxbase/ckd_alloc.h>
#include "hmm.h"
#include "s3o_psfwd.h"

static unsigned int g_r = 20260929u;
static unsigned int rnd(void) { g_r = g_r * 1664525u + 1013904223u; return g_r >> 8; }

int
main(int argc, char **argv)
{
    int n_cases = argc > 1 ? atoi(argv[1]) : 100000, c, n_sseq = 40, n_sen = 64, n_tmat = 6;
    for (c = 0; c < n_cases; c++) {
        const int n = (c & 1) ? 5 : 3, mpx = (c >> 1) & 1;
        uint8 ***tp = (uint8 ***)ckd_calloc_3d(n_tmat, n, n + 1, 1);
        uint16 **sseq = (uint16 **)ckd_calloc_2d(n_sseq, n, sizeof(uint16));
        uint16 *sseq_flat = ckd_calloc(n_sseq * n, sizeof(uint16));
        int16 *senscr = ckd_calloc(n_sen, sizeof(int16));
        hmm_context_t *ctx;
        hmm_t h;
        s3o_pshmm_t o;
        int i, j, k, bad = 0;
        int32 r1, r2;
        for (i = 0; i < n_tmat; i++)
            for (j = 0; j < n; j++)
                for (k = 0; k <= n; k++) {
                    int v = 255;
                    if (k == j || k == j + 1) v = rnd() % 200;
                    else if (k == j + 2) v = (rnd() % 3) ? rnd() % 255 : 255;     /* skip arcs: some matrices have none */
                    tp[i][j][k] = (uint8)v;
                }
        for (i = 0; i < n_sseq; i++)
            for (j = 0; j < n; j++) sseq[i][j] = sseq_flat[i * n + j] = rnd() % n_sen;
        for (i = 0; i < n_sen; i++) senscr[i] = (int16)(rnd() % ((c % 7 == 0) ? 30000 : 600));
        ctx = hmm_context_init(n, (uint8 ** const *)tp, senscr, (uint16 * const *)sseq);
        hmm_init(ctx, &h, mpx, rnd() % n_sseq, rnd() % n_tmat);
        for (i = 0; i < n; i++) {
            const unsigned int kind = rnd() % 8;
            h.score[i] = kind == 0 ? WORST_SCORE : kind == 1 ? WORST_SCORE + (int)(rnd() % 500) : -(int32)(rnd() % 2000000);
            if (i == 0 && kind == 0) h.score[0] = -(int32)(rnd() % 100000);
            h.history[i] = (int32)(rnd() % 1000) - 1;
            if (mpx) h.senid[i] = (i > 0 && rnd() % 4 == 0) ? BAD_SSID : rnd() % n_sseq;
        }
        h.out_score = -(int32)(rnd() % 100000); h.out_history = rnd() % 1000; h.bestscore = -7; h.frame = 3;
        memset(&o, 0, sizeof o);
        for (i = 0; i < n; i++) { o.score[i] = h.score[i]; o.hist[i] = h.history[i]; o.senid[i] = h.senid[i]; }
        o.out_score = h.out_score; o.out_hist = h.out_history; o.bestscore = h.bestscore; o.frame = h.frame;
        o.ssid = h.ssid; o.tmatid = h.tmatid; o.mpx = (uint8_t)mpx;
        r1 = hmm_vit_eval(&h);
        r2 = s3o_ps_hmm_vit_eval(&o, n, tp[0][0], sseq_flat, senscr);
        if (r1 != r2 || h.out_score != o.out_score || h.out_history != o.out_hist || h.bestscore != o.bestscore) bad = 1;
        for (i = 0; i < n; i++)
            if (h.score[i] != o.score[i] || h.history[i] != o.hist[i] || h.senid[i] != o.senid[i]) bad = 1;
        if (bad) {
            printf("hmm check: case %d (n=%d mpx=%d) differs: best %d/%d out %d/%d outh %d/%d\n", c, n, mpx, r1, r2, h.out_score,
                   o.out_score, h.out_history, o.out_hist);
            for (i = 0; i < n; i++) printf("  state %d: score %d/%d hist %d/%d senid %d/%d\n", i, h.score[i], o.score[i], h.history[i], o.hist[i], h.senid[i], o.senid[i]);
            return 1;
        }
        hmm_context_free(ctx);
        ckd_free_3d((void ***)tp); ckd_free_2d((void **)sseq); ckd_free(sseq_flat); ckd_free(senscr);
    }
    printf("hmm check: %d cases identical\n", n_cases);
    return 0;
}
