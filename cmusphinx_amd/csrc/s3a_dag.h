/*
 * s3a_dag.h -- internals of the second pass on the device (s3a_dag.hip), shared with the whole-utterance engine
 * (s3a_utt.hip), which binds its lanes' history tables to a pass.
 */
#ifndef S3A_DAG_H
#define S3A_DAG_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "s3a_internal.h"

/* io[] words of a lane (device; copied to the host after the pass) */
enum { DG_IO_ACTIVE, DG_IO_STATUS, DG_IO_NENT, DG_IO_ENDID, DG_IO_NHYP, DG_IO_NWORDS, DG_IO_NNODE, DG_IO_NLINK, DG_IO_NBYPASS,
       DG_IO_LMOP, DG_IO_SCORE, DG_IO_FIRSTSCORE, DG_IO_N = 16 };
/* status (s3a_dag_result_t.status) */
#define DG_E_NOEXIT  1      /* no word exit in the whole utterance (vithist_utt_end returns -1) */
#define DG_E_NOPATH  2      /* "Bestpath search failed": no link into the end node / LM operation limit (dag.c:930-947) */
#define DG_E_CAP     3      /* a capacity of the pass (entries, links, bypass pairs, words) or -maxedge */
#define DG_E_TABLE   4      /* inconsistent history table */
#define DG_E_POSEDGE 5      /* a positive bypass edge (the reference drops or keeps those depending on visiting order) */

/* a lane's history table (device pointers; the engine's WLane arrays or the pass's own copy of a host table) */
struct DagTab {
    int32_t *score, *pred, *lw0, *lw1, *wid, *sf, *ef, *ascr, *lscr, *type, *frame_start, *st;
    int32_t cap;
};

struct DagLane {
    DagTab tab;
    int32_t *io, *hyp_wid, *hyp_sf, *out;
    /* per entry */
    int32_t *sfp, *efp, *eslot, *enode, *eapos, *ehk, *knode, *hkent, *aent;
    /* entry / exit hashes */
    unsigned long long *h1key, *h2key, *h2best;
    int32_t *h1first, *h1last, *h1node;
    /* per frame */
    int32_t *ncnt, *nbase, *nfill, *kcnt, *kbase, *acnt, *abase, *afill, *lcnt, *loff;
    /* per node */
    int32_t *nfirst, *nwid, *nsf, *nfef, *nlef, *nkeep, *nfil, *nhk, *hkbase, *nkpos, *phead, *shead, *reach, *bpcnt, *bpbase;
    /* real links: path score, best predecessor link, its LM score */
    int32_t *lpscr, *lhist, *llscr;
    /* bypass pairs (hash) */
    unsigned long long *bkey, *bbest;
    int32_t *bdstar, *bnextp, *bnexts, *bpscr, *bhist, *blscr, *bplist;
    int32_t *task;
};

struct DagShared {
    int32_t n_word, F, E_cap, h1mask, bmask, link_cap, task_cap, hyp_cap;
    int32_t min_endfr, maxedge, maxlmop, maxlpf, startwid, finishwid, silwid, start_lwid, finish_lwid, wip;
    double lwf;
    const int32_t *basewid, *lwid, *fillpen;
    const uint8_t *is_filler;
};

/* engine side (s3a_utt.hip): bind lane z's table, run the pass for the first n lanes on `stream` (utt_end on the device
 * first), fetch the results */
int32_t s3a_dagpass_bind(s3a_dagpass_t *dp, int32_t lane, const DagTab &tab);
int32_t s3a_dagpass_enqueue(s3a_dagpass_t *dp, int32_t n, hipStream_t stream, int32_t do_utt_end);
int32_t s3a_dagpass_fetch(s3a_dagpass_t *dp, int32_t n, hipStream_t stream);
int32_t s3a_dagpass_finish(s3a_dagpass_t *dp, int32_t n, hipStream_t stream);     /* fetch + utterance order */
/* inside a queue with lane refill: descriptors up once, then the pass for the lanes of a refill event (ids on the device) */
int32_t s3a_dagpass_prepare(s3a_dagpass_t *dp, hipStream_t stream);
int32_t s3a_dagpass_enqueue_lanes(s3a_dagpass_t *dp, const int32_t *lane_ids_dev, int32_t n, hipStream_t stream);
const DagLane *s3a_dagpass_dev_lanes(const s3a_dagpass_t *dp);
int32_t s3a_dagpass_lattice_lane(s3a_dagpass_t *dp, int32_t lane, s3a_lat_info_t *info, s3a_lat_node_t *nodes, int32_t node_cap,
                                 s3a_lat_link_t *links, int32_t link_cap);
int32_t s3a_dagpass_hyp_cap(const s3a_dagpass_t *dp);


#endif
