/*
 * s3a_gather.hip -- the end-of-batch exchange of hypotheses in C, over RCCL: when the control file is
 * sharded over the GPUs of a node (-ctloffset / -ctlcount per rank, main_decode.c:164-169) every rank ends with the
 * hypotheses of its own utterances; ONE exchange brings them to every rank in control-file order, so that rank 0 writes
 * -hyp / -hypseg as a single process would.  No per-frame collective exists anywhere in the path.
 *
 * Variable-length records (a header + n_words words per utterance), three all-gathers on device buffers: the ranks'
 * counts, the headers (padded to the largest rank's count), the words (padded to the largest rank's total).  RCCL is
 * loaded at run time (dlopen librccl.so: the library itself does not link it; a host program without RCCL still loads
 * libcmusphinx_amd.so) and the communicator is bootstrapped through a file: rank 0 writes ncclGetUniqueId's 128 bytes to
 * `rendezvous` (write + rename), the other ranks wait for it -- the ranks of one node share a file system.
 */
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/stat.h>
#include <algorithm>
#include <vector>

#include "s3a_device.h"

typedef struct { char internal[128]; } rcclUniqueId;
typedef void *rcclComm_t;
typedef int (*fn_get_id)(rcclUniqueId *);
typedef int (*fn_init_rank)(rcclComm_t *, int, rcclUniqueId, int);
typedef int (*fn_all_gather)(const void *, void *, size_t, int, rcclComm_t, hipStream_t);
typedef int (*fn_destroy)(rcclComm_t);
typedef const char *(*fn_errstr)(int);

struct s3a_gather_s {
    void *lib;
    fn_get_id get_id; fn_init_rank init_rank; fn_all_gather all_gather; fn_destroy destroy; fn_errstr errstr;
    rcclComm_t comm;
    int32_t rank, world;
    int32_t host_staging;           /* the communication library takes host pointers (a test double: see below) */
    hipStream_t stream;
    /* staging, owned by the object and grown as needed: nothing is allocated per call once the sizes have been seen */
    void *d_cnt, *d_all, *d_h, *d_w, *d_ah, *d_aw;
    size_t cap_h, cap_w, cap_ah, cap_aw;
    std::vector<s3a_hyp_header_t> hdr;
    std::vector<s3a_hyp_word_t> words;
    std::vector<int64_t> word_off;
};

#define RCHK(g, expr) do { int r_ = (expr); if (r_ != 0) { s3a_set_error("%s failed: %s", #expr, (g)->errstr ? (g)->errstr(r_) : "?"); return S3A_EHIP; } } while (0)

/* staging memory: device memory for RCCL; plain host memory when the loaded library declares (by exporting the symbol
 * s3a_comm_takes_host_pointers) that its collectives take host pointers -- tests/mock_rccl.c, which lets the padding /
 * ordering arithmetic of this file run with world > 1 on a box without GPUs.  Real RCCL never exports that symbol. */
static int32_t st_alloc(s3a_gather_t *g, void **p, size_t n)
{
    if (g->host_staging) { *p = malloc(n ? n : 1); return *p ? S3A_OK : S3A_ENOMEM; }
    HIPCHK(hipMalloc(p, n ? n : 1));
    return S3A_OK;
}
static void st_free(s3a_gather_t *g, void *p) { if (!p) return; if (g->host_staging) free(p); else (void)hipFree(p); }
static int32_t st_grow(s3a_gather_t *g, void **p, size_t *cap, size_t need)
{
    if (*cap >= need && *p) return S3A_OK;
    st_free(g, *p); *p = NULL; *cap = 0;
    int32_t rc = st_alloc(g, p, need);
    if (rc == S3A_OK) *cap = need;
    return rc;
}
static int32_t st_put(s3a_gather_t *g, void *dst, const void *src, size_t n)
{
    if (!n) return S3A_OK;
    if (g->host_staging) { memcpy(dst, src, n); return S3A_OK; }
    HIPCHK(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, g->stream));
    return S3A_OK;
}
static int32_t st_get(s3a_gather_t *g, void *dst, const void *src, size_t n)
{
    if (!n) return S3A_OK;
    if (g->host_staging) { memcpy(dst, src, n); return S3A_OK; }
    HIPCHK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, g->stream));
    return S3A_OK;
}
static int32_t st_fill(s3a_gather_t *g, void *dst, int v, size_t n)
{
    if (g->host_staging) { memset(dst, v, n); return S3A_OK; }
    HIPCHK(hipMemsetAsync(dst, v, n, g->stream));
    return S3A_OK;
}
static int32_t st_sync(s3a_gather_t *g)
{
    if (g->host_staging) return S3A_OK;
    HIPCHK(hipStreamSynchronize(g->stream));
    return S3A_OK;
}

extern "C" void
s3a_gather_free(s3a_gather_t *g)
{
    if (!g) return;
    if (g->comm && g->destroy) (void)g->destroy(g->comm);
    void *bufs[] = { g->d_cnt, g->d_all, g->d_h, g->d_w, g->d_ah, g->d_aw };
    for (void *p : bufs) st_free(g, p);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    if (g->lib) dlclose(g->lib);
    delete g;
}

/* when this process started, in seconds since the epoch (/proc/self/stat field 22 + the boot time): a rendezvous file
 * older than that was left behind by an earlier run */
static double
process_start_epoch(void)
{
    FILE *fp = fopen("/proc/self/stat", "r");
    char buf[2048];
    unsigned long long ticks = 0, btime = 0;
    if (!fp) return 0.0;
    if (fgets(buf, sizeof buf, fp)) {
        char *p = strrchr(buf, ')');            /* the command name may hold spaces */
        int field = 2;
        for (p = p ? p + 1 : buf; p && *p; ) {
            while (*p == ' ') p++;
            if (++field == 22) { ticks = strtoull(p, NULL, 10); break; }
            while (*p && *p != ' ') p++;
        }
    }
    fclose(fp);
    if ((fp = fopen("/proc/stat", "r")) != NULL) {
        while (fgets(buf, sizeof buf, fp)) if (sscanf(buf, "btime %llu", &btime) == 1) break;
        fclose(fp);
    }
    const long hz = sysconf(_SC_CLK_TCK);
    return btime && hz > 0 ? (double)btime + (double)ticks / (double)hz : 0.0;
}

/* how old (against this process's start) a rendezvous file WITHOUT a run id may be and still count as this run's */
#define S3A_GATHER_FRESH_S 120.0

static s3a_gather_t *gather_init(int32_t rank, int32_t world, const char *rendezvous, unsigned long long run_id);

extern "C" s3a_gather_t *
s3a_gather_init(int32_t rank, int32_t world, const char *rendezvous)
{
    return gather_init(rank, world, rendezvous, 0ull);
}

extern "C" s3a_gather_t *
s3a_gather_init_run(int32_t rank, int32_t world, const char *rendezvous, unsigned long long run_id)
{
    if (run_id == 0ull) { s3a_set_error("s3a_gather_init_run: the run id must not be 0"); return NULL; }
    return gather_init(rank, world, rendezvous, run_id);
}

static s3a_gather_t *
gather_init(int32_t rank, int32_t world, const char *rendezvous_arg, unsigned long long run_id)
{
    /* with a run id the file is <rendezvous>.<run id>: another run's file has another name, and what the file holds is checked
     * against the id; no clock is compared (ranks may be born at any time, on hosts whose clocks differ) */
    char rdv_run[4200];
    const char *rendezvous = rendezvous_arg;
    if (run_id != 0ull && rendezvous_arg && *rendezvous_arg) {
        snprintf(rdv_run, sizeof rdv_run, "%s.%016llx", rendezvous_arg, run_id);
        rendezvous = rdv_run;
    }
    if (rank < 0 || world <= 0 || rank >= world || (world > 1 && (!rendezvous || !*rendezvous))) { s3a_set_error("s3a_gather_init: bad arguments"); return NULL; }
    s3a_gather_t *g = new s3a_gather_s();
    g->lib = NULL; g->comm = NULL; g->stream = NULL; g->rank = rank; g->world = world; g->host_staging = 0;
    g->d_cnt = g->d_all = g->d_h = g->d_w = g->d_ah = g->d_aw = NULL;
    g->cap_h = g->cap_w = g->cap_ah = g->cap_aw = 0;
    const char *names[] = { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so" };
    for (auto n : names) if (!g->lib) g->lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!g->lib) { s3a_set_error("s3a_gather_init: librccl.so not found (%s)", dlerror()); delete g; return NULL; }
    g->get_id = (fn_get_id)dlsym(g->lib, "ncclGetUniqueId"); g->init_rank = (fn_init_rank)dlsym(g->lib, "ncclCommInitRank");
    g->all_gather = (fn_all_gather)dlsym(g->lib, "ncclAllGather"); g->destroy = (fn_destroy)dlsym(g->lib, "ncclCommDestroy");
    g->errstr = (fn_errstr)dlsym(g->lib, "ncclGetErrorString");
    g->host_staging = dlsym(g->lib, "s3a_comm_takes_host_pointers") != NULL;
    if (!g->get_id || !g->init_rank || !g->all_gather || !g->destroy) { s3a_set_error("s3a_gather_init: librccl.so lacks the nccl* entry points"); s3a_gather_free(g); return NULL; }
    rcclUniqueId id;
    memset(&id, 0, sizeof id);
    if (rank == 0) {
        int r = g->get_id(&id);
        if (r != 0) { s3a_set_error("ncclGetUniqueId failed: %s", g->errstr ? g->errstr(r) : "?"); s3a_gather_free(g); return NULL; }
        if (world > 1) {
            char tmp[4300];
            snprintf(tmp, sizeof tmp, "%s.tmp", rendezvous);
            (void)unlink(rendezvous);           /* (whatever an earlier run of the same name left) */
            FILE *fp = fopen(tmp, "wb");
            if (!fp || (run_id != 0ull && fwrite(&run_id, sizeof run_id, 1, fp) != 1) || fwrite(&id, sizeof id, 1, fp) != 1 || fclose(fp) != 0
                || rename(tmp, rendezvous) != 0) {
                s3a_set_error("s3a_gather_init: cannot write the rendezvous file %s", rendezvous); s3a_gather_free(g); return NULL;
            }
        }
    }
    else {
        /* a file left by a crashed earlier run would hold a dead communicator's id.  With a run id: the file's name and first
         * word carry it (the launcher hands every rank of a run the same id -- MASTER_PORT, a job id -- and a new one to the next
         * run).  Without: only a file not older than S3A_GATHER_FRESH_S before this process started is taken -- the ranks of a
         * run must then start within that window of one another, on one host clock (s3a_gather_init's contract) */
        const double born = run_id != 0ull ? 0.0 : process_start_epoch();
        int tries = 0;
        for (;; tries++) {
            struct stat sb;
            if (stat(rendezvous, &sb) == 0 && (run_id != 0ull || born == 0.0 || (double)sb.st_mtime + S3A_GATHER_FRESH_S >= born)) {
                FILE *fp = fopen(rendezvous, "rb");
                if (fp) {
                    unsigned long long got = 0ull;
                    const bool head = run_id == 0ull || (fread(&got, sizeof got, 1, fp) == 1 && got == run_id);
                    const size_t k = head ? fread(&id, sizeof id, 1, fp) : 0;
                    fclose(fp);
                    if (k == 1) break;
                }
            }
            if (tries > 6000) { s3a_set_error("s3a_gather_init: rank %d waited 10 minutes for a fresh %s", rank, rendezvous); s3a_gather_free(g); return NULL; }
            usleep(100000);
        }
    }
    if (!g->host_staging && hipStreamCreate(&g->stream) != hipSuccess) { s3a_set_error("s3a_gather_init: hipStreamCreate failed"); s3a_gather_free(g); return NULL; }
    int r = g->init_rank(&g->comm, world, id, rank);
    if (r != 0) { s3a_set_error("ncclCommInitRank failed: %s", g->errstr ? g->errstr(r) : "?"); g->comm = NULL; s3a_gather_free(g); return NULL; }
    if (st_alloc(g, &g->d_cnt, 2 * sizeof(long long)) != S3A_OK || st_alloc(g, &g->d_all, sizeof(long long) * 2 * world) != S3A_OK) { s3a_gather_free(g); return NULL; }
    if (world > 1) {
        /* the rendezvous file has done its work once EVERY rank holds the id: one small all-gather (each rank's pid) is the proof --
         * nobody returns from it before everybody has entered it, i.e. has read the file --, and rank 0 removes the file right behind
         * it.  A later run under the same name (a launcher that reuses MASTER_PORT as its run id) then never meets this run's file: its
         * ranks wait for their own rank 0's. */
        long long me[2] = { (long long)getpid(), (long long)rank };
        std::vector<long long> all((size_t)2 * world);
        int32_t rc = st_put(g, g->d_cnt, me, sizeof me);
        if (rc == S3A_OK) {
            const int r2 = g->all_gather(g->d_cnt, g->d_all, sizeof me, 0 /* ncclInt8: bytes */, g->comm, g->stream);
            if (r2 != 0) { s3a_set_error("s3a_gather_init: the first all-gather failed: %s", g->errstr ? g->errstr(r2) : "?"); rc = S3A_EHIP; }
        }
        if (rc == S3A_OK) rc = st_get(g, all.data(), g->d_all, sizeof(long long) * 2 * world);
        if (rc == S3A_OK) rc = st_sync(g);
        if (rank == 0) (void)unlink(rendezvous);
        if (rc != S3A_OK) { s3a_gather_free(g); return NULL; }
        for (int r = 0; r < world; r++)
            if (all[2 * r + 1] != r) { s3a_set_error("s3a_gather_init: rank %d answered in rank %d's place (two runs under one rendezvous name?)", (int)all[2 * r + 1], r); s3a_gather_free(g); return NULL; }
    }
    return g;
}

/* this rank's hypotheses in, everybody's out (utterance order; every utterance index 0 .. n_total - 1 exactly once).
 * Every argument is checked BEFORE the first collective: a rank that returns early would leave the others waiting. */
extern "C" int32_t
s3a_gather_hyps(s3a_gather_t *g, int32_t n_local, const s3a_hyp_header_t *hdr, const s3a_hyp_word_t *words, int32_t n_total)
{
    if (!g || n_local < 0 || (n_local > 0 && !hdr) || n_total < 0) { s3a_set_error("s3a_gather_hyps: bad argument"); return S3A_EINVAL; }
    long long nw_local = 0;
    for (int32_t i = 0; i < n_local; i++) {
        if (hdr[i].status == 0 && hdr[i].n_words < 0) { s3a_set_error("s3a_gather_hyps: record %d has a negative word count", i); return S3A_EINVAL; }
        nw_local += hdr[i].status == 0 ? hdr[i].n_words : 0;
    }
    if (nw_local > 0 && !words) { s3a_set_error("s3a_gather_hyps: words missing"); return S3A_EINVAL; }
    const int W = g->world;
    int32_t rc;
    long long cnt[2] = { n_local, nw_local };
    std::vector<long long> all((size_t)2 * W);
    if ((rc = st_put(g, g->d_cnt, cnt, sizeof cnt)) != S3A_OK) return rc;
    RCHK(g, g->all_gather(g->d_cnt, g->d_all, sizeof cnt, 0 /* ncclInt8: bytes */, g->comm, g->stream));
    if ((rc = st_get(g, all.data(), g->d_all, sizeof(long long) * 2 * W)) != S3A_OK || (rc = st_sync(g)) != S3A_OK) return rc;
    long long mh = 1, mw = 1;
    for (int r = 0; r < W; r++) { mh = std::max(mh, all[2 * r]); mw = std::max(mw, all[2 * r + 1]); }
    const size_t hb = (size_t)mh * sizeof(s3a_hyp_header_t), wb = (size_t)mw * sizeof(s3a_hyp_word_t);
    if ((rc = st_grow(g, &g->d_h, &g->cap_h, hb)) != S3A_OK || (rc = st_grow(g, &g->d_w, &g->cap_w, wb)) != S3A_OK
        || (rc = st_grow(g, &g->d_ah, &g->cap_ah, hb * W)) != S3A_OK || (rc = st_grow(g, &g->d_aw, &g->cap_aw, wb * W)) != S3A_OK)
        return rc;          /* (every rank computes the same sizes: an allocation failure is the only way to diverge here) */
    if ((rc = st_fill(g, g->d_h, 0xff, hb)) != S3A_OK || (rc = st_fill(g, g->d_w, 0, wb)) != S3A_OK) return rc;   /* padding headers: utt_index -1 */
    if ((rc = st_put(g, g->d_h, hdr, (size_t)n_local * sizeof(s3a_hyp_header_t))) != S3A_OK) return rc;
    /* the words of the utterances that have any, back to back */
    std::vector<s3a_hyp_word_t> flat((size_t)nw_local);
    {
        size_t pos = 0, src = 0;
        for (int32_t i = 0; i < n_local; i++) {
            const int32_t n = hdr[i].status == 0 ? hdr[i].n_words : 0;
            if (n) memcpy(&flat[pos], words + src, (size_t)n * sizeof(s3a_hyp_word_t));
            pos += n; src += n;
        }
    }
    if ((rc = st_put(g, g->d_w, flat.data(), (size_t)nw_local * sizeof(s3a_hyp_word_t))) != S3A_OK) return rc;
    RCHK(g, g->all_gather(g->d_h, g->d_ah, hb, 0, g->comm, g->stream));          /* the lengths (and all that is fixed-size) */
    RCHK(g, g->all_gather(g->d_w, g->d_aw, wb, 0, g->comm, g->stream));          /* the padded payload */
    std::vector<char> ah(hb * W), aw(wb * W);
    if ((rc = st_get(g, ah.data(), g->d_ah, hb * W)) != S3A_OK || (rc = st_get(g, aw.data(), g->d_aw, wb * W)) != S3A_OK
        || (rc = st_sync(g)) != S3A_OK) return rc;
    struct Rec { int32_t idx; const s3a_hyp_header_t *h; const s3a_hyp_word_t *w; };
    std::vector<Rec> recs;
    for (int r = 0; r < W; r++) {
        const s3a_hyp_header_t *hh = (const s3a_hyp_header_t *)(ah.data() + hb * r);
        const s3a_hyp_word_t *ww = (const s3a_hyp_word_t *)(aw.data() + wb * r);
        size_t pos = 0;
        for (long long i = 0; i < all[2 * r]; i++) {
            if (hh[i].utt_index < 0) continue;
            recs.push_back({ hh[i].utt_index, &hh[i], ww + pos });
            pos += hh[i].status == 0 ? hh[i].n_words : 0;
        }
    }
    std::sort(recs.begin(), recs.end(), [](const Rec &a, const Rec &b) { return a.idx < b.idx; });
    if ((int32_t)recs.size() != n_total) { s3a_set_error("s3a_gather_hyps: %zu utterances arrived, %d expected", recs.size(), n_total); return S3A_EINVAL; }
    for (int32_t i = 0; i < n_total; i++)
        if (recs[i].idx != i) { s3a_set_error("s3a_gather_hyps: utterance %d missing or duplicated across ranks", i); return S3A_EINVAL; }
    g->hdr.resize(n_total); g->word_off.assign((size_t)n_total + 1, 0);
    size_t tot = 0;
    for (int32_t i = 0; i < n_total; i++) { g->hdr[i] = *recs[i].h; g->word_off[i] = (int64_t)tot; tot += recs[i].h->status == 0 ? recs[i].h->n_words : 0; }
    g->word_off[n_total] = (int64_t)tot;
    g->words.resize(tot ? tot : 1);
    for (int32_t i = 0; i < n_total; i++) {
        const int32_t n = recs[i].h->status == 0 ? recs[i].h->n_words : 0;
        if (n) memcpy(&g->words[(size_t)g->word_off[i]], recs[i].w, (size_t)n * sizeof(s3a_hyp_word_t));
    }
    return S3A_OK;
}

extern "C" int32_t
s3a_gather_result(const s3a_gather_t *g, int32_t utt_index, const s3a_hyp_header_t **hdr, const s3a_hyp_word_t **words)
{
    if (!g || !hdr || !words || utt_index < 0 || utt_index >= (int32_t)g->hdr.size()) return S3A_EINVAL;
    *hdr = &g->hdr[utt_index];
    *words = &g->words[(size_t)g->word_off[utt_index]];
    return S3A_OK;
}
