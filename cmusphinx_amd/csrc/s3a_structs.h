/*
 * s3a_structs.h -- the device-object structures shared by the .hip translation units
 * (scorer, composite senones, lexical-tree search, fused decoder frame).
 */
#ifndef S3A_STRUCTS_H
#define S3A_STRUCTS_H

#include <vector>
#include "s3a_device.h"

struct s3a_scorer_s {
    s3a_mgau_model_t *g;
    int32_t n_sen, n_ci_sen;
    int16_t *cd2cisen_h;
    /* fast_gmm_t subset */
    int32_t ds_ratio, cond_ds, ci_pbeam, max_cd, dyn_ci_pbeam, skip_count;
    float tighten_factor;
    /* device */
    int16_t *cd2cisen_d;
    uint8_t *ncomp_d;       /* [S] */
    float *x_d;             /* [D4*4] */
    uint8_t *act_d;         /* [S] */
    int32_t *scr_d;         /* [S] */
    int32_t *ci_d;          /* [n_ci_sen] */
    int32_t *misc_d;        /* [0]=best [1]=ns [2]=ng */
    int32_t *gpart_d;       /* [3][gp_n]: per-workgroup best / #senones / #Gaussians of the fused CD phase */
    int32_t gp_n, gpart_valid;
    int32_t *misc_h;        /* pinned mirror */
    /* mgau_t.bstidx/bstscr/updatetime: the model's own arrays, or private ones when several decoders
     * share one model (s3a_scorer_init_private) */
    int32_t *bstidx_d, *bstscr_d, *updatetime_d;
    int own_state;
    int32_t *ci_occ_h, *idx_h;
};


/*
 * The search state of a lextree node is ONE 64-byte record: the three (or five) state scores, their histories, the exit score and
 * history, the HMM's best score, the frame it is listed for (an HMM evaluation touched 17 cache lines of nine node-indexed arrays for it; now one
 * or two).  The kernels keep their pointers -- sc, hist, outs, outh, bests, frame = the record's fields at node 0 -- and
 * index them with NSI (state st of node v) / NSV (a per-node field of v).  The list stamp (posf) and the other per-node
 * words that sweeps read for ALL nodes stay arrays of their own.
 */
#define NST 16                                  /* int32 words per node record */
#define NSI(st, N, v) ((size_t)(v) * NST + (st))
#define NSV(v) ((size_t)(v) * NST)
/* a node's place in the active list and the frame that place is valid for (pos, posf) are ONE 8-byte pair per node -- they are
 * written together (the emission: one scattered store, one dirty line per listed node instead of two) and read together (is the
 * node on the list, and where): the kernels keep two pointers, posf = pos + 1, and index both with PPX */
#define PPX(v) ((size_t)(v) * 2)
#define PP_SET(pos, v, k, f) (*(int2 *)((pos) + PPX(v)) = make_int2((k), (f)))
/* field offsets for an HMM of ne emitting states (3 or 5): ne scores, ne histories, exit score, exit history, best score,
 * frame tag = 2 ne + 4 <= NST words.  A kernel that holds the derived pointers knows ne as hist - sc. */
#define NS_HIST(ne) (ne)
#define NS_OUTS(ne) (2 * (ne))
#define NS_OUTH(ne) (2 * (ne) + 1)
#define NS_BESTS(ne) (2 * (ne) + 2)
#define NS_FRAME(ne) (2 * (ne) + 3)    /* the frame the HMM is listed for (hmm_frame) */
/* transition-matrix row stride on the device: ne x (ne + 1) words padded to 16-byte multiples (12 / 32) */
#define NS_TPW(ne) ((((ne) * ((ne) + 1)) + 3) & ~3)

struct s3a_comsen_s {
    int32_t n_comstate, n_list;
    int32_t *off_d, *wt_d, *out_d, *scr_d;
    int16_t *list_d;
    size_t scr_cap;
    hipStream_t stream;
};


struct s3a_lexsearch_s {
    int32_t n_tree, N;                  /* N = total nodes */
    int32_t n_emit, n_tmat, n_sen, n_comsen, n_lcmax;
    std::vector<int32_t> node_base;     /* [n_tree+1] */
    std::vector<int32_t> n_lc;          /* per tree */
    std::vector<int32_t> n_root;        /* per tree: distinct root nodes */
    std::vector<std::vector<int16_t>> lc;           /* per tree: lc ids */
    std::vector<std::vector<int32_t>> lcroot_off;   /* per tree: CSR into the tree's root buffer */
    std::vector<int32_t> rootbuf_base;  /* per tree: offset of its root lists in d_rootlist */
    std::vector<int32_t> h_rootlist;    /* host copy of the concatenated root lists */
    /* static (device) */
    int32_t *d_node_base;
    int32_t *d_ssid, *d_tmatid, *d_wid, *d_prob;
    uint8_t *d_comp;
    int32_t *d_child_off, *d_child, *d_par_off, *d_par;
    int32_t *d_rootlist;                /* concatenated root lists (global node ids) */
    int32_t *d_tp;
    int16_t *d_sseq, *d_comsseq, *d_comstate;
    int32_t *d_comstate_off;
    /* state (device) */
    int32_t *d_sc, *d_hist;             /* [3][N] */
    int32_t *d_outs, *d_outh, *d_bests, *d_frame;
    int32_t *d_pos, *d_posf;            /* position in the list of frame posf */
    int32_t *d_act[2];                  /* [N] each; tree t owns [node_base[t], node_base[t+1]) */
    int32_t *d_nact[2];                 /* [n_tree] */
    int cur;                            /* index of the "active" buffer; the other is next_active */
    /* per-frame scratch */
    int32_t *d_cand, *d_ncand, *d_candf;    /* candidate inactive children per tree */
    int32_t *d_turn, *d_selfemit, *d_cnt;   /* [N] */
    int32_t *d_best;                    /* [n_tree][2] best, wbest */
    int32_t *d_exit;                    /* [3][N] wid, score, hist of word exits (tree slices) */
    unsigned long long *d_scan_agg, *d_scan_pre;   /* [T][chunks]: k_dec_scan's chained scan (totals / inclusive prefixes) */
    int32_t *d_scan_flag, scan_epoch, scan_chunks;
    int32_t *d_poswid, *d_posout;       /* [N] by LIST POSITION: word id / exit score of the HMM evaluated there (fused frame) */
    int32_t *d_nexit;                   /* [n_tree] + [n_tree] error flags */
    int32_t *d_calls, *d_ent, *d_eflag, *d_first;   /* enter scratch */
    unsigned long long *d_key;
    int32_t ent_cap;
    int32_t *d_thr;                     /* [8] thresholds + frame statistics */
    int32_t *d_tree_of;                 /* [N] tree index of every node */
    int32_t *d_done;                    /* workgroup completion counter of the fused finishing kernel */
    int32_t *d_hbin;                    /* [1000] lextree_hmm_histbin bins | [1000] = the histogram beam */
    int32_t *d_rootnodes, n_rootnodes;  /* the distinct root nodes of all trees */
    int32_t *d_ps, *d_psof_off, *d_psof, *d_psmem_off, *d_psmem, *d_pstamp, n_pset;    /* parent-set ids: of a node, of a node's children; frame stamps */
    int32_t *d_ctot, *d_n0;             /* per-call root counts [4096], list lengths before the entries [n_tree] */
    int32_t hist_bound, last_nnxt;      /* host upper bound on the coming frame's active HMMs */
    int32_t row_bound;                  /* ... on its LONGEST active list (per tree: sizes the per-position grids) */
    std::vector<int32_t> nnxt_t;        /* per tree: what the last search emitted */
    int32_t *d_pack, *h_pack;           /* per-frame result record (device / pinned host) */
    int32_t pack_max_exits;
    int32_t *h_ring;                    /* pinned staging ring for enter calls */
    int32_t ring_slot;
    int32_t *h_pin;                     /* pinned host mirror for small read-backs */
    hipStream_t stream;
    int own_stream;
    int opt_calls_by_copy, opt_scan_chained;    /* S3A_CALLS_BY_COPY / S3A_SCAN_CHAINED as read when the object was made (tests) */
    int is_clone;                       /* static device arrays borrowed from a prototype (s3a_lexsearch_clone) */
    hipEvent_t ev_pack;                 /* recorded after the frame record's copy (the emission kernel follows it) */
};



/* the frame-synchronous entry points (single calls of the lextree API, the fused frame, the batch of decoders) run 3-state
 * HMMs; 5-state topologies are served by the whole-utterance engine (s3a_uttdec_*) */
#define LS_NEED_3ST(ls, fn) do { if ((ls) && (ls)->n_emit != 3) { \
        s3a_set_error(fn ": %d-state HMMs are decoded by the whole-utterance engine (s3a_uttdec_*) only", (ls)->n_emit); \
        return S3A_EUNSUP; } } while (0)

/* internal cross-TU entry points */
int32_t s3a_scorer_enqueue_raw(s3a_scorer_t *sc, const float *feat, int32_t frame);
int32_t s3a_scorer_reset_frame_state(s3a_scorer_t *sc);
int32_t s3a_reset_state_arrays(hipStream_t stream, int32_t *bstidx, int32_t *bstscr, int32_t *updatetime, int32_t S);
int32_t s3a_dec_stage_calls(s3a_lexsearch_t *ls, int32_t tree_a, int32_t n_a, const int32_t *lc_a,
                            const int32_t *scr_a, const int32_t *hist_a, int32_t tree_b, int32_t n_b,
                            const int32_t *lc_b, const int32_t *scr_b, const int32_t *hist_b,
                            int32_t *groups, int32_t *calls, int32_t max_calls, int32_t *n_calls,
                            int32_t *n_ent, int32_t *n_groups);
int32_t s3a_dec_unpack(s3a_lexsearch_t *ls, const int32_t *p, bool may_hist, int32_t frm,
                       s3a_frame_result_t *res, int32_t *n_exit, int32_t max_exits, int32_t *total);

#endif
