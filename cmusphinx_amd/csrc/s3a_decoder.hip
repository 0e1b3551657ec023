/*
 * s3a_decoder.hip -- the FUSED search frame: the same results as the step-by-step
 * entry points of s3a_scorer.hip / s3a_lextree.hip with a third of the kernel launches.
 *
 * Why: one search frame of sphinx3 mode 4 is ~10 dependent phases on a few thousand
 * HMMs; each phase is microseconds of work, so the frame costs what its kernel
 * BOUNDARIES cost (measured: ~1.8 us per enqueued operation, 31 operations per frame
 * in the step-by-step path, with 4 decoder streams saturating the command processor).
 * The fused frame needs 11:
 *
 *   s3a_decoder_score        H2D feature vector | CI senones | gated CD senones
 *                            (scores stay RAW; the mask is consumed and cleared)
 *   s3a_decoder_search       k_dec_hmm_eval   lextree_hmm_eval with the frame normaliser
 *                                             and the composite-senone max applied on the fly
 *                                             (approx_cont_mgau.c:597-600, dict2pid.c:1029-1048)
 *                            k_dec_resolve    lextree_hmm_propagate_non_leaves, phases mark +
 *                                             resolve in one pass over ALL nodes (no candidate
 *                                             list), beam thresholds recomputed per workgroup
 *                            [k_dec_hist_*    histogram pruning, only when the frame may exceed
 *                                             1.5 x maxhmmpf HMMs]
 *                            k_dec_scan       per tree: turn offsets of the next list, ordered
 *                                             word exits; the last workgroup to finish packs the
 *                                             frame record and resets the per-frame accumulators
 *                            k_dec_emit       ordered emission of the next list, wave-parallel
 *                            D2H record + the frame's ONE synchronisation
 *   s3a_decoder_transition   H2D calls | k_dec_enter1 | k_dec_enter2 | k_dec_enter3_mark
 *                            (lextree_enter for the unigram AND the filler tree of the
 *                            frame, then the active-senone marks of the NEXT frame)
 *
 * Parity: tests/test_gpu_lextree.py runs this path in lock step with the step-by-step one
 * and with the oracle; tests/test_gpu_dropin.py decodes tidigits and RM1 through it and
 * diffs -hyp/-hypseg against the unmodified reference.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <vector>

#include "s3a_device.h"
#include "s3a_structs.h"
#include "s3a_vit.h"
#include "s3a_scan.h"
#include "s3a_decoder_kernels.h"

/* thin kernels over the shared bodies (s3a_decoder_kernels.h) */
template <int EB>
__global__ void __launch_bounds__(EB)
k_dec_hmm_eval(const int32_t *__restrict__ node_base, const int32_t *__restrict__ act,
               const int32_t *__restrict__ nact, int32_t N, int32_t n_tmat,
               const int32_t *__restrict__ ssid, const int32_t *__restrict__ tmatid,
               const int32_t *__restrict__ wid, const uint8_t *__restrict__ comp,
               const int32_t *__restrict__ tp_g, const int16_t *__restrict__ sseq,
               const int16_t *__restrict__ comsseq, const int32_t *__restrict__ cs_off,
               const int16_t *__restrict__ cs_list, const int32_t *__restrict__ cs_wt,
               const int32_t *__restrict__ raw, const int32_t *__restrict__ misc,
               int32_t *sc, int32_t *hist, int32_t *outs, int32_t *outh, int32_t *bests,
               int32_t *best_out, int32_t cf, const int32_t *__restrict__ psof_off,
               const int32_t *__restrict__ psof, int32_t *pstamp,
               const int32_t *__restrict__ gpart, int32_t gpart_n, int32_t *poswid, int32_t *posout)
{
    d_dec_hmm_eval<EB>(node_base, act, nact, N, n_tmat, ssid, tmatid, wid, comp, tp_g, sseq, comsseq, cs_off, cs_list, cs_wt, raw, misc, sc, hist, outs, outh, bests, best_out, cf, psof_off, psof, pstamp, gpart, gpart_n, poswid, posout, blockIdx.x, blockIdx.y);
}

__global__ void __launch_bounds__(DBLOCK)
k_dec_hist_count(const int32_t *__restrict__ node_base, const int32_t *__restrict__ act,
                 const int32_t *__restrict__ nact, int32_t T, FrameBeams bm,
                 const int32_t *__restrict__ best, const int32_t *__restrict__ bests,
                 int32_t *binof, int32_t *hbin, int32_t force_tree, int32_t fbest, int32_t fbw,
                 int32_t nbin)
{
    d_dec_hist_count(node_base, act, nact, T, bm, best, bests, binof, hbin, force_tree, fbest, fbw, nbin, blockIdx.x, blockIdx.y);
}

__global__ void __launch_bounds__(SCAN_THREADS)
k_dec_hist_sort(const int32_t *__restrict__ node_base, int32_t *act, const int32_t *__restrict__ nact,
                int32_t T, FrameBeams bm, const int32_t *__restrict__ binof, int32_t *tmp,
                int32_t *hbin, int32_t *pos, int32_t force_tree, int32_t nbin)
{
    d_dec_hist_sort(node_base, act, nact, T, bm, binof, tmp, hbin, pos, force_tree, nbin, blockIdx.x, blockIdx.y);
}

__global__ void __launch_bounds__(RSBLOCK)
k_dec_resolve(int32_t N, int32_t T, int32_t cf, FrameBeams bm, const int32_t *__restrict__ best,
              const int32_t *__restrict__ nact, const int32_t *__restrict__ node_base,
              const int32_t *__restrict__ tree_of, const int32_t *__restrict__ prob,
              const int32_t *__restrict__ par_off, const int32_t *__restrict__ par,
              const int32_t *__restrict__ pos, const int32_t *__restrict__ posf,
              int32_t *sc, int32_t *hist, int32_t *outs, int32_t *outh, int32_t *bests,
              int32_t *frame, int32_t *turn, int32_t *selfemit, int32_t *cnt,
              unsigned long long *key, int32_t *first, int32_t *hbin,
              const int32_t *__restrict__ ps, const int32_t *__restrict__ pstamp,
              const int32_t *__restrict__ rootnodes, int32_t n_rootnodes,
              const int32_t *__restrict__ propf, int32_t *posout)
{
    d_dec_resolve(N, T, cf, bm, best, nact, node_base, tree_of, prob, par_off, par, pos, posf, sc, hist, outs, outh, bests, frame, turn, selfemit, cnt, key, first, hbin, ps, pstamp, rootnodes, n_rootnodes, propf, posout, blockIdx.x, blockIdx.y);
}

__global__ void __launch_bounds__(SCAN_THREADS)
k_dec_weak(int32_t N, int32_t T, int32_t cf, FrameBeams bm, const int32_t *__restrict__ best,
           const int32_t *__restrict__ nact, const int32_t *__restrict__ node_base,
           const int32_t *__restrict__ act, const int32_t *__restrict__ prob,
           const int32_t *__restrict__ par_off, const int32_t *__restrict__ par,
           const int32_t *__restrict__ pos, const int32_t *__restrict__ posf, const int32_t *__restrict__ sc,
           const int32_t *__restrict__ outs, const int32_t *__restrict__ bests, const int32_t *__restrict__ wid,
           const int32_t *__restrict__ hbin, int32_t *propf, int32_t *weaklist)
{
    d_dec_weak(N, T, cf, bm, best, nact, node_base, act, prob, par_off, par, pos, posf, sc, outs, bests, wid, hbin, propf, weaklist, 0, 0);
}


__global__ void __launch_bounds__(SCAN_THREADS)
k_dec_scan(int32_t N, int32_t T, int32_t cf, FrameBeams bm, const int32_t *__restrict__ node_base,
           const int32_t *__restrict__ act, const int32_t *__restrict__ nact,
           const int32_t *__restrict__ wid, const int32_t *__restrict__ prob,
           const int32_t *__restrict__ outs, const int32_t *__restrict__ outh,
           const int32_t *__restrict__ selfemit, int32_t *cnt, int32_t *base, int32_t *nxt, int32_t *nnxt,
           int32_t *pos, int32_t *posf, int32_t *best, int32_t *exits, int32_t *nexit,
           const int32_t *hbin, int32_t *misc, int32_t *done, int32_t *pack, int32_t max_exits,
           const int32_t *gpart, int32_t gpart_n, const int32_t *poswid, const int32_t *posout, int32_t reordered,
           unsigned long long *st_agg, unsigned long long *st_pre, int32_t *st_flag, int32_t st_stride,
           int32_t epoch, int32_t NC)
{
    d_dec_scan(N, T, cf, bm, node_base, act, nact, wid, prob, outs, outh, selfemit, cnt, base, nxt, nnxt, pos, posf, best, exits, nexit, hbin, misc, done, pack, max_exits, gpart, gpart_n, poswid, posout, reordered, st_agg, st_pre, st_flag, st_stride, epoch, NC, NC, blockIdx.x, blockIdx.y);
}

__global__ void __launch_bounds__(DBLOCK)
k_dec_emit(int32_t cf, const int32_t *__restrict__ node_base, const int32_t *__restrict__ act,
           const int32_t *__restrict__ nact, const int32_t *__restrict__ child_off,
           const int32_t *__restrict__ child, int32_t *turn, int32_t *selfemit,
           const int32_t *__restrict__ base, int32_t *nxt, const int32_t *nnxt, int32_t *pos, int32_t *posf)
{
    d_dec_emit(cf, node_base, act, nact, child_off, child, turn, selfemit, base, nxt, nnxt, pos, posf, blockIdx.x, blockIdx.y);
}

/* the frame's lextree_enter calls as a kernel argument (groups[8] then 4 ints per call) when there are at most
 * CALLS_BY_ARG of them -- the usual case: one call per final phone with an exit -- instead of a host-to-device
 * copy in front of the three kernels */
#define CALLS_BY_ARG 96
struct CallsArg { int32_t v[8 + 4 * CALLS_BY_ARG]; };

__global__ void
k_dec_enter1(Entries ent, int32_t n_ent, const int32_t *__restrict__ calls,
             const int32_t *__restrict__ prob, const int32_t *__restrict__ sc, int32_t thresh,
             unsigned long long *key, int32_t *first, CallsArg ca)
{
    if (calls == NULL) { calls = ca.v + 8; ent.calls = calls; }
    d_dec_enter1(ent, n_ent, calls, prob, sc, thresh, key, first, blockIdx.x, blockIdx.y);
}

__global__ void __launch_bounds__(SCAN_THREADS)
k_dec_enter2(Entries ent, int32_t n_ent, const int32_t *__restrict__ calls,
             const int32_t *__restrict__ prob, const int32_t *__restrict__ sc,
             const int32_t *__restrict__ frame, const int32_t *__restrict__ first, int32_t thresh,
             int32_t nf, int32_t T, const int32_t *__restrict__ nnxt, int32_t *flag, int32_t *ctot, int32_t *n0,
             CallsArg ca)
{
    if (calls == NULL) { calls = ca.v + 8; ent.calls = calls; }
    d_dec_enter2(ent, n_ent, calls, prob, sc, frame, first, thresh, nf, T, nnxt, flag, ctot, n0, blockIdx.x, 0);
}

__global__ void __launch_bounds__(M3BLOCK)
k_dec_enter3_mark(int32_t n_ent_blocks, Entries ent, int32_t n_ent,
                  const int32_t *__restrict__ calls, const int32_t *__restrict__ groups, int32_t n_groups,
                  int32_t nf, const unsigned long long *__restrict__ key, const int32_t *__restrict__ first,
                  const int32_t *__restrict__ flag, const int32_t *__restrict__ ctot,
                  const int32_t *__restrict__ n0, int32_t *sc, int32_t *hist, int32_t *frame,
                  int32_t T, int32_t blocks_per_tree, const int32_t *__restrict__ node_base,
                  int32_t *nxt, int32_t *nnxt, int32_t *pos, int32_t *posf,
                  const int32_t *__restrict__ ssid, const uint8_t *__restrict__ comp,
                  const int16_t *__restrict__ sseq, const int16_t *__restrict__ comsseq,
                  const int32_t *__restrict__ cs_off, const int16_t *__restrict__ cs_list,
                  uint8_t *sen_active, CallsArg ca)
{
    if (calls == NULL) { calls = ca.v + 8; ent.calls = calls; groups = ca.v; }
    d_dec_enter3_mark(n_ent_blocks, ent, n_ent, calls, groups, n_groups, nf, key, first, flag, ctot, n0, sc, hist,
                      frame, T, blocks_per_tree, node_base, nxt, nnxt, pos, posf, ssid, comp, sseq, comsseq, cs_off,
                      cs_list, sen_active, blockIdx.x, 0);
}

/* ------------------------------------------------------------------ */
/* host side                                                           */
/* ------------------------------------------------------------------ */
/* the fused frame orders scorer and search kernels by stream order only */
static int32_t
same_stream(const s3a_lexsearch_t *ls, const s3a_scorer_t *sc)
{
    if (ls->stream == sc->g->dev->stream) return S3A_OK;
    s3a_set_error("s3a_decoder_*: the lexsearch and the scorer must share one stream "
                  "(create the lexsearch with s3a_mgau_stream(g))");
    return S3A_EINVAL;
}

static int32_t
max_tree_nodes(const s3a_lexsearch_t *ls)
{
    int32_t m = 0;
    for (int32_t t = 0; t < ls->n_tree; t++)
        m = max(m, ls->node_base[t + 1] - ls->node_base[t]);
    return m;
}

/* lextree_hmm_histbin (lextree.c:1314-1358) for ONE tree, as a stand-alone operation: adds the
 * tree's HMMs to bin[0..nbin) and reorders its active list exactly as the reference does.
 * (The fused frame runs the same kernels for all trees at once, see s3a_decoder_search.) */
extern "C" int32_t
s3a_lexsearch_hmm_histbin(s3a_lexsearch_t *ls, int32_t tree, int32_t bestscr, int32_t *bin,
                          int32_t nbin, int32_t bw)
{
    if (!ls || !bin || tree < 0 || tree >= ls->n_tree || nbin <= 0 || nbin > NBIN || bw == 0) return S3A_EINVAL;
    const int32_t T = ls->n_tree, maxn = max_tree_nodes(ls);
    FrameBeams bm = { 0, 0, 0, 0, 0 };
    std::vector<int32_t> h(nbin);
    HIPCHK(hipMemsetAsync(ls->d_hbin, 0, NBIN * 4, ls->stream));
    hipLaunchKernelGGL(k_dec_hist_count, dim3((maxn + DBLOCK - 1) / DBLOCK, T), dim3(DBLOCK), 0, ls->stream,
                       ls->d_node_base, ls->d_act[ls->cur], ls->d_nact[ls->cur], T, bm, ls->d_best, ls->d_bests,
                       ls->d_exit + ls->N, ls->d_hbin, tree, bestscr, bw, nbin);
    hipLaunchKernelGGL(k_dec_hist_sort, dim3(T), dim3(SCAN_THREADS), 0, ls->stream, ls->d_node_base,
                       ls->d_act[ls->cur], ls->d_nact[ls->cur], T, bm, ls->d_exit + ls->N, ls->d_exit, ls->d_hbin,
                       ls->d_pos, tree, nbin);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h.data(), ls->d_hbin, (size_t)nbin * 4, hipMemcpyDeviceToHost, ls->stream));
    HIPCHK(hipMemsetAsync(ls->d_hbin, 0, NBIN * 4, ls->stream));
    HIPCHK(hipStreamSynchronize(ls->stream));
    for (int32_t i = 0; i < nbin; i++) bin[i] += h[i];
    return S3A_OK;
}

/* host half of a frame's lextree_enter calls: per call {inscore, inhist, offset of its root list,
 * first entry index}, per tree a group {tree, first entry, last entry + 1, first call}; also the bound on the
 * coming frame's active HMMs (ls->hist_bound).  Shared with the batched engine (s3a_batch.hip). */
int32_t
s3a_dec_stage_calls(s3a_lexsearch_t *ls, int32_t tree_a, int32_t n_a, const int32_t *lc_a, const int32_t *scr_a,
                    const int32_t *hist_a, int32_t tree_b, int32_t n_b, const int32_t *lc_b,
                    const int32_t *scr_b, const int32_t *hist_b, int32_t *groups, int32_t *calls,
                    int32_t max_calls, int32_t *n_calls, int32_t *n_ent_out, int32_t *n_groups_out)
{
    const int32_t T = ls->n_tree;
    int32_t n_ent = 0, n_groups = 0, c = 0, roots = 0;
    std::vector<int32_t> ent_t(T, 0);
    if (n_a < 0 || n_b < 0 || n_a + n_b > max_calls) {
        s3a_set_error("lextree_enter: %d calls in one frame exceed the staging buffer (%d)", n_a + n_b, max_calls);
        return S3A_EINVAL;
    }
    for (int g = 0; g < 2; g++) {
        const int32_t tree = g ? tree_b : tree_a, n = g ? n_b : n_a;
        const int32_t *lc = g ? lc_b : lc_a, *scr = g ? scr_b : scr_a, *hi = g ? hist_b : hist_a;
        if (n == 0) continue;
        if (tree < 0 || tree >= T) return S3A_EINVAL;
        const int32_t lo = n_ent, c_lo = c;
        for (int32_t i = 0; i < n; i++, c++) {
            int32_t k = 0;
            if (ls->n_lc[tree] > 0) {
                for (k = 0; k < ls->n_lc[tree] && ls->lc[tree][k] != lc[i]; k++);
                if (k >= ls->n_lc[tree]) {
                    s3a_set_error("lextree_enter: left context %d is not a root context of tree %d", lc[i], tree);
                    return S3A_EINVAL;
                }
            }
            const int32_t len = ls->lcroot_off[tree][k + 1] - ls->lcroot_off[tree][k];
            if (n_ent + len > ls->ent_cap) { s3a_set_error("lextree_enter: entry staging overflow"); return S3A_EINVAL; }
            calls[4 * c] = scr[i];
            calls[4 * c + 1] = hi[i];
            calls[4 * c + 2] = ls->rootbuf_base[tree] + ls->lcroot_off[tree][k];
            calls[4 * c + 3] = n_ent;
            n_ent += len;
        }
        groups[4 * n_groups] = tree; groups[4 * n_groups + 1] = lo; groups[4 * n_groups + 2] = n_ent;
        groups[4 * n_groups + 3] = c_lo;
        n_groups++;
        roots += ls->n_root[tree];
        ent_t[tree] += min(n_ent - lo, ls->n_root[tree]);
    }
    /* >= the coming frame's active HMMs: what propagation listed + the distinct roots entered */
    ls->hist_bound = ls->last_nnxt + min(n_ent, roots);
    /* per tree: its own share of both (the grids over list positions are per tree) */
    ls->row_bound = 1;
    for (int32_t t = 0; t < T; t++)
        ls->row_bound = max(ls->row_bound, ((size_t)t < ls->nnxt_t.size() ? ls->nnxt_t[t] : 0) + ent_t[t]);
    *n_calls = c; *n_ent_out = n_ent; *n_groups_out = n_groups;
    return S3A_OK;
}

/* the frame record (see k_dec_scan) -> s3a_frame_result_t + per-tree exit counts; shared with the
 * batched engine.  Returns the total number of word exits in *total. */
int32_t
s3a_dec_unpack(s3a_lexsearch_t *ls, const int32_t *p, bool may_hist, int32_t frm, s3a_frame_result_t *res,
               int32_t *n_exit, int32_t max_exits, int32_t *total_out)
{
    const int32_t T = ls->n_tree;
    int32_t total = 0, t;
    res->best_hmm = p[3 * T + 3]; res->best_word = p[3 * T + 4]; res->n_hmm = p[3 * T + 5];
    res->thres = p[3 * T + 0]; res->phone_thres = p[3 * T + 1]; res->word_thres = p[3 * T + 2];
    res->need_histprune = p[3 * T + 6];     /* informational: the histogram beam was applied */
    for (int i = 0; i < 8; i++) res->extra[i] = p[5 * T + 8 + i];
    ls->last_nnxt = 0;
    ls->nnxt_t.assign(T, 0);
    for (t = 0; t < T; t++) { ls->nnxt_t[t] = p[5 * T + 16 + t]; ls->last_nnxt += p[5 * T + 16 + t]; }
    if (res->need_histprune && !may_hist) {
        s3a_set_error("fused frame: internal error, %d active HMMs exceed the host bound %d", res->n_hmm, ls->hist_bound);
        return S3A_EINVAL;
    }
    for (t = 0; t < T; t++) {
        if (p[4 * T + 8 + t] == 2) {
            s3a_set_error("fused frame: the chained scan of tree %d timed out (internal error)", t);
            return S3A_EHIP;
        }
        if (p[4 * T + 8 + t]) {
            s3a_set_error("out.history==-1 at a word exit of tree %d (LEXTREE_OPERATION_FAILURE)", t);
            return S3A_EINVAL;
        }
        n_exit[t] = p[3 * T + 8 + t];
        total += n_exit[t];
    }
    res->n_exit_total = total;
    if (total > max_exits || total > ls->pack_max_exits) {
        s3a_set_error("fused frame: %d word exits in one frame exceed the buffers", total);
        return S3A_EINVAL;
    }
    *total_out = total;
    return S3A_OK;
}

extern "C" int32_t
s3a_decoder_utt_begin(s3a_lexsearch_t *ls, s3a_scorer_t *sc)
{
    LS_NEED_3ST(ls, "s3a_decoder_utt_begin");
    int32_t rc;
    if (!ls || !sc) return S3A_EINVAL;
    if ((rc = same_stream(ls, sc)) != S3A_OK) return rc;
    if ((rc = s3a_scorer_utt_begin(sc)) != S3A_OK) return rc;
    if ((rc = s3a_scorer_reset_frame_state(sc)) != S3A_OK) return rc;
    ls->last_nnxt = 0;
    ls->hist_bound = 0;
    ls->row_bound = 1;
    ls->nnxt_t.assign(ls->n_tree, 0);
    return S3A_OK;      /* d_best / d_done / key / first are left clean by reset, utt_end and k_dec_finish */
}

extern "C" int32_t
s3a_decoder_score(s3a_scorer_t *sc, const float *feat, int32_t frame)
{
    if (!sc || !feat) return S3A_EINVAL;
    return s3a_scorer_enqueue_raw(sc, feat, frame);
}

extern "C" int32_t
s3a_decoder_search(s3a_lexsearch_t *ls, s3a_scorer_t *sc, s3a_comsen_t *cs, int32_t frm,
                   int32_t hmmbeam, int32_t pbeam, int32_t wbeam, int32_t phone_uses_wbeam,
                   int32_t maxhmmpf, s3a_frame_result_t *res, int32_t *n_exit, int32_t *exit_wid,
                   int32_t *exit_score, int32_t *exit_hist, int32_t max_exits)
{
    LS_NEED_3ST(ls, "s3a_decoder_search");
    if (!ls || !sc || !cs || !res || !n_exit || !exit_wid || !exit_score || !exit_hist) return S3A_EINVAL;
    if (same_stream(ls, sc) != S3A_OK) return S3A_EINVAL;
    const int32_t T = ls->n_tree, hdr = 6 * T + 16, maxn = max_tree_nodes(ls);
    const int cur = ls->cur, nxt = cur ^ 1;
    FrameBeams bm = { hmmbeam, pbeam, wbeam, phone_uses_wbeam, maxhmmpf };
    int32_t total = 0;
    /* histogram pruning can only fire when the frame holds more than 1.5 x maxhmmpf HMMs; the host
     * knows an upper bound (last frame's next list + the root entries it sent), so the two
     * histogram kernels are only enqueued when that bound allows it */
    const bool may_hist = ls->hist_bound > maxhmmpf + (maxhmmpf >> 1);
    if (may_hist && -hmmbeam / NBIN == 0) {
        s3a_set_error("s3a_decoder_search: -beam too narrow for histogram pruning (bin width 0)");
        return S3A_EUNSUP;
    }

    /* the fused CD phase left its maxima / counters per workgroup (s3a_scorer_enqueue_raw) */
    const int32_t gpart_n = sc->gpart_valid ? sc->gp_n : 0;
    sc->gpart_valid = 0;
    /* the active lists are at most hist_bound long (host bound): size the per-position grids by it */
    const int32_t rows = min(maxn, max(min(ls->hist_bound, ls->row_bound), 1));
    if (rows >= EVBLOCK_LONG_LIST)
        hipLaunchKernelGGL(k_dec_hmm_eval<256>, dim3((rows + 255) / 256, T), dim3(256),
                       0, ls->stream, ls->d_node_base, ls->d_act[cur],
                       ls->d_nact[cur], ls->N, ls->n_tmat, ls->d_ssid, ls->d_tmatid, ls->d_wid, ls->d_comp,
                       ls->d_tp, ls->d_sseq, ls->d_comsseq, cs->off_d, cs->list_d, cs->wt_d, sc->scr_d,
                       sc->misc_d, ls->d_sc, ls->d_hist, ls->d_outs, ls->d_outh, ls->d_bests, ls->d_best, frm,
                       ls->d_psof_off, ls->d_psof, ls->d_pstamp, sc->gpart_d, gpart_n, ls->d_poswid, ls->d_posout);
    else
        hipLaunchKernelGGL(k_dec_hmm_eval<64>, dim3((rows + 63) / 64, T), dim3(64),
                       0, ls->stream, ls->d_node_base, ls->d_act[cur],
                       ls->d_nact[cur], ls->N, ls->n_tmat, ls->d_ssid, ls->d_tmatid, ls->d_wid, ls->d_comp,
                       ls->d_tp, ls->d_sseq, ls->d_comsseq, cs->off_d, cs->list_d, cs->wt_d, sc->scr_d,
                       sc->misc_d, ls->d_sc, ls->d_hist, ls->d_outs, ls->d_outh, ls->d_bests, ls->d_best, frm,
                       ls->d_psof_off, ls->d_psof, ls->d_pstamp, sc->gpart_d, gpart_n, ls->d_poswid, ls->d_posout);
    if (may_hist) {
        hipLaunchKernelGGL(k_dec_hist_count, dim3((rows + DBLOCK - 1) / DBLOCK, T), dim3(DBLOCK), 0, ls->stream,
                           ls->d_node_base, ls->d_act[cur], ls->d_nact[cur], T, bm, ls->d_best, ls->d_bests,
                           ls->d_exit + ls->N, ls->d_hbin, -1, 0, 1, NBIN);
        hipLaunchKernelGGL(k_dec_hist_sort, dim3(T), dim3(SCAN_THREADS), 0, ls->stream, ls->d_node_base,
                           ls->d_act[cur], ls->d_nact[cur], T, bm, ls->d_exit + ls->N, ls->d_exit, ls->d_hbin,
                           ls->d_pos, -1, NBIN);
    }
    if (phone_uses_wbeam || pbeam < hmmbeam)     /* the phone threshold may fall below the HMM threshold */
        hipLaunchKernelGGL(k_dec_weak, dim3(1), dim3(SCAN_THREADS), 0, ls->stream, ls->N, T, frm, bm, ls->d_best,
                           ls->d_nact[cur], ls->d_node_base, ls->d_act[cur], ls->d_prob, ls->d_par_off, ls->d_par,
                           ls->d_pos, ls->d_posf, ls->d_sc, ls->d_outs, ls->d_bests, ls->d_wid, ls->d_hbin,
                           ls->d_candf, ls->d_exit + 2 * (size_t)ls->N);
    hipLaunchKernelGGL(k_dec_resolve, dim3((ls->N + RSBLOCK - 1) / RSBLOCK), dim3(RSBLOCK), 0, ls->stream,
                       ls->N, T, frm, bm, ls->d_best, ls->d_nact[cur], ls->d_node_base, ls->d_tree_of,
                       ls->d_prob, ls->d_par_off, ls->d_par, ls->d_pos, ls->d_posf, ls->d_sc, ls->d_hist,
                       ls->d_outs, ls->d_outh, ls->d_bests, ls->d_frame, ls->d_turn, ls->d_selfemit,
                       ls->d_cnt, ls->d_key, ls->d_first, ls->d_hbin, ls->d_ps, ls->d_pstamp, ls->d_rootnodes,
                       ls->n_rootnodes, ls->d_candf, ls->d_posout);
    const int32_t scan_nc = scan_workgroups(rows, ls->opt_scan_chained);
    hipLaunchKernelGGL(k_dec_scan, dim3(T * scan_nc), dim3(SCAN_THREADS), 0, ls->stream, ls->N, T, frm, bm,
                       ls->d_node_base, ls->d_act[cur], ls->d_nact[cur], ls->d_wid, ls->d_prob, ls->d_outs,
                       ls->d_outh, ls->d_selfemit, ls->d_cnt, ls->d_cand, ls->d_act[nxt], ls->d_nact[nxt],
                       ls->d_pos, ls->d_posf, ls->d_best, ls->d_exit, ls->d_nexit, ls->d_hbin, sc->misc_d,
                       ls->d_done, ls->h_pack, ls->pack_max_exits, sc->gpart_d, gpart_n, ls->d_poswid, ls->d_posout,
                       may_hist ? 1 : 0, ls->d_scan_agg, ls->d_scan_pre, ls->d_scan_flag, ls->scan_chunks,
                       ++ls->scan_epoch, scan_nc);
    HIPCHK(hipGetLastError());
    /* the last workgroup of k_dec_scan wrote the frame record (header + every exit) straight into pinned host
     * memory -- posted writes, no copy engine in the frame; the host waits for that kernel only, and the emission
     * kernel overlaps its word-level work (the next frame's kernels follow it in stream order) */
    HIPCHK(hipEventRecord(ls->ev_pack, ls->stream));
    hipLaunchKernelGGL(k_dec_emit, dim3(EMIT_BLOCKS, T), dim3(DBLOCK), 0, ls->stream, frm, ls->d_node_base,
                       ls->d_act[cur], ls->d_nact[cur], ls->d_child_off, ls->d_child, ls->d_turn, ls->d_selfemit,
                       ls->d_cand, ls->d_act[nxt], ls->d_nact[nxt], ls->d_pos, ls->d_posf);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventSynchronize(ls->ev_pack));
    const int32_t *p = ls->h_pack;
    {
        const int32_t rc = s3a_dec_unpack(ls, p, may_hist, frm, res, n_exit, max_exits, &total);
        if (rc != S3A_OK) return rc;
    }
    for (int32_t k = 0; k < total; k++) {
        exit_wid[k] = p[hdr + 3 * k];
        exit_score[k] = p[hdr + 3 * k + 1];
        exit_hist[k] = p[hdr + 3 * k + 2];
    }
    return S3A_OK;
}

/*
 * srch_utt_word_trans's lextree_enter calls (tree_a: n_a calls with left contexts; tree_b:
 * one call, the filler tree; either count may be 0), the senone marks of the coming frame
 * and lextree_active_swap.  Also used at utterance begin (cf = -1).
 */
extern "C" int32_t
s3a_decoder_transition(s3a_lexsearch_t *ls, s3a_scorer_t *sc, s3a_comsen_t *cs, int32_t cf,
                       int32_t thresh, int32_t tree_a, int32_t n_a, const int32_t *lc_a,
                       const int32_t *scr_a, const int32_t *hist_a, int32_t tree_b, int32_t n_b,
                       const int32_t *lc_b, const int32_t *scr_b, const int32_t *hist_b)
{
    LS_NEED_3ST(ls, "s3a_decoder_transition");
    if (!ls || !sc || !cs) return S3A_EINVAL;
    if (same_stream(ls, sc) != S3A_OK) return S3A_EINVAL;
    const int32_t T = ls->n_tree, maxn = max_tree_nodes(ls);
    const int nxt = ls->cur ^ 1;
    int32_t *slot = ls->h_ring + (size_t)(ls->ring_slot++ & 7) * ((size_t)2 * 4096 + (size_t)2 * ls->ent_cap);
    int32_t *groups = slot, *calls = slot + 8;          /* pinned staging: [groups 8][calls 4 each] */
    int32_t n_ent = 0, n_groups = 0, c = 0, rc;

    if ((rc = s3a_dec_stage_calls(ls, tree_a, n_a, lc_a, scr_a, hist_a, tree_b, n_b, lc_b, scr_b, hist_b, groups,
                                  calls, 2046, &c, &n_ent, &n_groups)) != S3A_OK)
        return rc;
    const bool by_arg = c <= CALLS_BY_ARG && !ls->opt_calls_by_copy;          /* (the option: tests of the copy path) */
    CallsArg ca;
    memset(&ca, 0, sizeof ca);
    if (by_arg) memcpy(ca.v, slot, (size_t)(8 + 4 * c) * 4);
    const int32_t *d_calls = by_arg ? (const int32_t *)NULL : ls->d_calls + 8;
    const int32_t *d_groups = by_arg ? (const int32_t *)NULL : ls->d_calls;
    const Entries ent = { d_calls, ls->d_rootlist, c };
    if (n_ent > 0) {
        if (!by_arg)
            HIPCHK(hipMemcpyAsync(ls->d_calls, slot, (size_t)(8 + 4 * c) * 4, hipMemcpyHostToDevice, ls->stream));
        hipLaunchKernelGGL(k_dec_enter1, dim3((n_ent + 255) / 256), dim3(256), 0, ls->stream, ent,
                           n_ent, d_calls, ls->d_prob, ls->d_sc, thresh, ls->d_key, ls->d_first, ca);
        hipLaunchKernelGGL(k_dec_enter2, dim3(c), dim3(SCAN_THREADS), 0, ls->stream, ent, n_ent,
                           d_calls, ls->d_prob, ls->d_sc, ls->d_frame, ls->d_first, thresh, cf + 1, T,
                           ls->d_nact[nxt], ls->d_eflag, ls->d_ctot, ls->d_n0, ca);
    }
    {
        const int32_t n_ent_blocks = (n_ent + M3BLOCK - 1) / M3BLOCK;
        /* the lists before the entries hold at most last_nnxt nodes (what the search emitted) */
        const int32_t bpt = (min(maxn, max(ls->last_nnxt, 1)) + M3BLOCK - 1) / M3BLOCK;
        hipLaunchKernelGGL(k_dec_enter3_mark, dim3(n_ent_blocks + bpt * T), dim3(M3BLOCK), 0, ls->stream,
                           n_ent_blocks, ent, n_ent, d_calls, d_groups, n_groups, cf + 1, ls->d_key,
                           ls->d_first, ls->d_eflag, ls->d_ctot, n_ent > 0 ? ls->d_n0 : ls->d_nact[nxt], ls->d_sc,
                           ls->d_hist, ls->d_frame, T, bpt, ls->d_node_base, ls->d_act[nxt], ls->d_nact[nxt],
                           ls->d_pos, ls->d_posf, ls->d_ssid, ls->d_comp, ls->d_sseq, ls->d_comsseq, cs->off_d,
                           cs->list_d, sc->act_d, ca);
    }
    HIPCHK(hipGetLastError());
    ls->cur ^= 1;       /* lextree_active_swap; the new next-list counts are overwritten by k_dec_finish */
    return S3A_OK;
}
