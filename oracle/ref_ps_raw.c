/*
 * oracle/ref_ps_raw.c -- TEST INFRASTRUCTURE: BASELINE.json configs[0], the CPU plumbing case.  The UNMODIFIED
 * pocketsphinx (oracle/_ref/libpsref.so) decoding a raw 16 kHz file through its utterance API -- ps_init,
 * ps_decode_raw (pocketsphinx/src/libpocketsphinx/pocketsphinx.c:575), ps_get_hyp -- with the PTM
 * semi-continuous model, as pocketsphinx/test/unit/test_fsg.c:21-70 does with the bundled model.  No GPU, nothing
 * of ours in the path: it pins what "keeping the ps_decoder_t utterance API" means for the drop-in.
 * usage: ref_ps_raw HMMDIR FSG DICT RAWFILE
 */
#include <stdio.h>
#include <pocketsphinx.h>

int
main(int argc, char *argv[])
{
    ps_decoder_t *ps;
    cmd_ln_t *config;
    FILE *rawfh;
    char const *hyp, *uttid;
    int32 score;
    if (argc != 5) { fprintf(stderr, "usage: ref_ps_raw HMMDIR FSG DICT RAWFILE\n"); return 2; }
    config = cmd_ln_init(NULL, ps_args(), TRUE, "-hmm", argv[1], "-fsg", argv[2], "-dict", argv[3],
                         "-samprate", "16000", NULL);
    if (!config || (ps = ps_init(config)) == NULL) { fprintf(stderr, "ps_init failed\n"); return 1; }
    if ((rawfh = fopen(argv[4], "rb")) == NULL) { perror(argv[4]); return 1; }
    if (ps_decode_raw(ps, rawfh, "goforward", -1) < 0) { fprintf(stderr, "ps_decode_raw failed\n"); return 1; }
    hyp = ps_get_hyp(ps, &score, &uttid);
    printf("HYP: %s (%s %d)\n", hyp ? hyp : "(null)", uttid, score);
    fclose(rawfh);
    ps_free(ps);
    return 0;
}
