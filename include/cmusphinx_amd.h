/*
 * cmusphinx_amd.h -- C ABI of the MI355X-native acoustic-scoring / Viterbi
 * backend for CMU Sphinx-3 (libcmusphinx_amd.so).
 *
 * Plain C: opaque handles, plain pointers and sizes, int status codes.  No
 * torch / HIP types cross this boundary (streams are passed as void *).
 * Each entry point names the reference interface it replaces; paths are
 * relative to the reference root (cjac/cmusphinx).  INTEGRATION.md shows the
 * srch_funcs_t shim a sphinx3 maintainer adds on top of this header.
 *
 * Conventions
 *   - "host" pointers are ordinary process memory, "dev" pointers are HIP
 *     device memory of the current device.  Functions without a _dev suffix
 *     take and return host memory and synchronise before returning.
 *   - int32 scores are logs3 values (integer log-probabilities in base
 *     -logbase, default 1.0003), exactly the reference's.
 *   - status: S3A_OK (0) or a negative S3A_E* code; handle constructors
 *     return NULL on failure; s3a_last_error() gives the message.  Where the
 *     reference would E_FATAL (abort) on a malformed model file, this library
 *     fails the call instead of killing the host process.
 *   - thread-safety: as the reference (sphinx3/include/srch.h, one kb_t per
 *     thread): a handle must not be used from two threads at once.
 *   - there is NO CPU fallback: every scoring / Viterbi entry point runs HIP
 *     kernels and returns S3A_ENODEV when no gfx950 device is usable.
 */
#ifndef CMUSPHINX_AMD_H
#define CMUSPHINX_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S3A_OK        0
#define S3A_EINVAL   (-1)   /* bad argument */
#define S3A_EIO      (-2)   /* file missing / malformed / checksum error */
#define S3A_ENOMEM   (-3)
#define S3A_ENODEV   (-4)   /* no usable HIP device */
#define S3A_EHIP     (-5)   /* HIP runtime error (see s3a_last_error) */
#define S3A_EUNSUP   (-6)   /* model shape outside what the kernels support */

/* sphinx3/include/s3types.h:192 ; sphinxbase prim_type.h:153 */
#define S3A_LOGPROB_ZERO  ((int32_t)0xc8000000)
#define S3A_MAX_NEG_INT32 ((int32_t)0x80000000)
/* sphinx3/include/cont_mgau.h:121-123 */
#define S3A_NO_BSTIDX   (-1)
#define S3A_NOT_UPDATED (-100)
/* sphinx3/include/cont_mgau.h:127-133 */
#define S3A_FULL_INT_COMP      0
#define S3A_FULL_FLOAT_COMP    1
#define S3A_MIX_INT_FLOAT_COMP 2

const char *s3a_last_error(void);
const char *s3a_version(void);
/* number of usable gfx950 devices (0 when none / no driver); never fails */
int32_t s3a_device_count(void);
/* select the device this thread's subsequent calls use (hipSetDevice) */
int32_t s3a_set_device(int32_t ordinal);

/* ===================================================================== */
/* integer log-domain arithmetic                                          */
/* replaces sphinxbase logmath_t: sphinxbase/include/sphinxbase/logmath.h */
/*   logmath_init  logmath.c:61-161   logmath_add   logmath.c:391-436     */
/*   logmath_log   logmath.c:445-452  log_to_ln / ln_to_log  :466-477     */
/*   logs3_init / logs3  sphinx3/src/libs3decoder/libcommon/logs3.c:101-119 */
/* ===================================================================== */
typedef struct s3a_logmath_s s3a_logmath_t;

s3a_logmath_t *s3a_logmath_init(double base, int32_t shift, int32_t use_table);
s3a_logmath_t *s3a_logs3_init(double base, int32_t breport, int32_t blogtable);
void    s3a_logmath_free(s3a_logmath_t *lm);
int32_t s3a_logmath_add(const s3a_logmath_t *lm, int32_t logb_x, int32_t logb_y);
int32_t s3a_logmath_log(const s3a_logmath_t *lm, double p);
double  s3a_logmath_exp(const s3a_logmath_t *lm, int32_t logb_p);
int32_t s3a_logmath_ln_to_log(const s3a_logmath_t *lm, double log_p);
double  s3a_logmath_log_to_ln(const s3a_logmath_t *lm, int32_t logb_p);
int32_t s3a_logmath_log10_to_log(const s3a_logmath_t *lm, double log_p);
double  s3a_logmath_get_base(const s3a_logmath_t *lm);
int32_t s3a_logmath_get_zero(const s3a_logmath_t *lm);
/* logmath_get_table_shape, logmath.c:357-369 */
int32_t s3a_logmath_get_table_shape(const s3a_logmath_t *lm, uint32_t *out_size,
                                    uint32_t *out_width, uint32_t *out_shift);
/* copy the add-table, widened to uint32, into out[size] */
int32_t s3a_logmath_copy_table(const s3a_logmath_t *lm, uint32_t *out, uint32_t size);
int32_t s3a_logs3(const s3a_logmath_t *lm, double p);

/* ===================================================================== */
/* continuous-density mixture-Gaussian model                              */
/* replaces mgau_model_t: sphinx3/include/cont_mgau.h:170-226             */
/* ===================================================================== */
typedef struct s3a_mgau_model_s s3a_mgau_model_t;

/*
 * mgau_init (sphinx3/src/libs3decoder/libam/cont_mgau.c:901-956), same
 * argument list: reads the S3 means / variances / mixture_weights files,
 * removes uninitialised components, floors variances, precomputes
 * 1/(2 sigma^2) and the log-reciprocal-determinant terms, converts mixture
 * weights to logs3 -- in the reference's order and precision -- then packs the
 * model into its device layout and uploads it.  Only senmgau ".cont." and
 * comp_type S3A_MIX_INT_FLOAT_COMP are supported (the hot path of SURVEY.md
 * section 8); diagonal covariances only.
 */
s3a_mgau_model_t *s3a_mgau_init(const char *meanfile, const char *varfile, double varfloor,
                                const char *mixwfile, double mixwfloor, int32_t precomp,
                                const char *senmgau, int32_t comp_type,
                                s3a_logmath_t *logmath);
/* the same from raw file-domain arrays mean/var [n_mgau][n_density][veclen],
 * mixw [n_mgau][n_density] (what mgau_file_read / mgau_mixw_read hand on) */
s3a_mgau_model_t *s3a_mgau_init_arrays(const float *mean, const float *var, const float *mixw,
                                       int32_t n_mgau, int32_t n_density, int32_t veclen,
                                       double varfloor, double mixwfloor, int32_t precomp,
                                       s3a_logmath_t *logmath);
/*
 * Host-only load: everything mgau_init does up to, but excluding, the upload.
 * For model tools and for checking the loader where no GPU exists; every
 * scoring entry point fails with S3A_ENODEV on such a handle (no CPU scoring).
 */
s3a_mgau_model_t *s3a_mgau_load_host(const char *meanfile, const char *varfile, double varfloor,
                                     const char *mixwfile, double mixwfloor, int32_t precomp,
                                     s3a_logmath_t *logmath);
void    s3a_mgau_free(s3a_mgau_model_t *g);                 /* mgau_free, cont_mgau.c:1210 */
/* An ADAPTED model (-mllr / -ctl_mllr): the parameters mgau_model_t holds after adapt_set_mllr (libam/adaptor.c:106-170:
 * reload, mllr_norm_mgau mllr.c:210-256, variance floor, mgau_precomp cont_mgau.c:857-894 -- the reference's host code,
 * run by kb_setmllr kb.c:335-365) replace the model's in place: mean / prec [n_mgau][max_comp][veclen] (prec = the
 * precomputed 1 / (2 sigma^2)), lrd [n_mgau][max_comp], rows of senone m valid up to s3a_mgau_n_comp(g, m). */
int32_t s3a_mgau_set_params(s3a_mgau_model_t *g, const float *mean, const float *prec, const float *lrd);
int32_t s3a_mgau_n_mgau(const s3a_mgau_model_t *g);         /* mgau_n_mgau   cont_mgau.h:229 */
int32_t s3a_mgau_max_comp(const s3a_mgau_model_t *g);       /* mgau_max_comp cont_mgau.h:230 */
int32_t s3a_mgau_veclen(const s3a_mgau_model_t *g);         /* mgau_veclen   cont_mgau.h:231 */
int32_t s3a_mgau_n_comp(const s3a_mgau_model_t *g, int32_t m); /* mgau_n_comp cont_mgau.h:232 */
double  s3a_mgau_distfloor(const s3a_mgau_model_t *g);
/* host copies of the precomputed parameters, AoS as in the reference
 * (mgau_mean/mgau_var/mgau_lrd/mgau_mixw, cont_mgau.h:233-238); any out
 * pointer may be NULL.  mean/prec: [n_mgau][max_comp][veclen], lrd/mixw:
 * [n_mgau][max_comp]; entries at c >= n_comp[m] are unspecified. */
int32_t s3a_mgau_get_params(const s3a_mgau_model_t *g, float *mean, float *prec,
                            float *lrd, int32_t *mixw, int32_t *n_comp);
/* per-senone mutable state (mgau_t.bstidx/bstscr/updatetime, cont_mgau.h:174-176):
 * reset as srch_TST_begin does (srch_time_switch_tree.c:485-490), read back for tests */
int32_t s3a_mgau_reset_state(s3a_mgau_model_t *g);
int32_t s3a_mgau_get_state(const s3a_mgau_model_t *g, int32_t *bstidx, int32_t *bstscr,
                           int32_t *updatetime);

/* arithmetic of the Gaussian kernel */
#define S3A_GMM_EXACT 0   /* f32 subtract, f64 multiply/accumulate, no FMA: bit-identical to mgau_eval */
#define S3A_GMM_FAST  1   /* f32 FMA accumulation: within +-S3A_GMM_FAST_TOL logs3 units */
int32_t s3a_mgau_set_precision(s3a_mgau_model_t *g, int32_t mode);

/*
 * mgau_eval (cont_mgau.c:1174-1205; declared cont_mgau.h:317-325): score ONE
 * senone m at ONE vector x (host, veclen floats), optionally restricted to the
 * -1-terminated component list active_comp, updating the senone's
 * bstidx/bstscr/updatetime when update_best_id.  Runs on the device; meant for
 * drop-in completeness and parity tests, not throughput.
 */
int32_t s3a_mgau_eval(s3a_mgau_model_t *g, int32_t m, const int32_t *active_comp,
                      const float *x, int32_t fr, int32_t update_best_id);

/*
 * Whole-block senone scoring: senscr[t][s] = mgau_eval(g, s, NULL, feat[t], t, 1)
 * for t < n_frames, s < n_mgau -- the un-normalised all-senones pass that
 * utt_decode's frame loop performs one frame at a time
 * (srch.c:746-751 -> gmm_wrap.c:108-167 with every senone active and the CI
 * beam open).  feat: [n_frames][veclen] float32.  best (optional): per-frame
 * maximum, i.e. the value approx_cont_mgau_frame_eval returns / srch->senscale.
 */
int32_t s3a_mgau_score_frames(s3a_mgau_model_t *g, const float *feat, int32_t n_frames,
                              int32_t *senscr, int32_t *best);
/* device-resident variant: nothing leaves HBM; asynchronous on `stream`
 * (a hipStream_t passed as void *, NULL = default stream) */
int32_t s3a_mgau_score_frames_dev(s3a_mgau_model_t *g, const float *feat_dev, int32_t n_frames,
                                  int32_t *senscr_dev, int32_t *best_dev, void *stream);

/* ===================================================================== */
/* per-frame scoring driver with CI gating                                */
/* replaces fast_gmm_t (sphinx3/include/fast_algo_struct.h:200-262) +     */
/* approx_cont_mgau_{ci,frame}_eval (approx_cont_mgau.c:367-616) + the    */
/* ascr_t buffers they fill (sphinx3/include/ascr.h:99-116)               */
/* ===================================================================== */
typedef struct s3a_scorer_s s3a_scorer_t;

/*
 * Bundles what gmm_wrap.c:108-211 pulls out of srch_t/kbcore_t: the model,
 * mdef's cd2cisen map (mdef.h:206-208; n_sen entries, the first n_ci_sen are
 * identity) and the fast_gmm_init parameters
 * (fast_algo_struct.c:419-461: -ds, -cond_ds, -ci_pbeam (as a probability),
 * -tighten_factor, -maxcdsenpf).  Gaussian selection / sub-VQ are not
 * supported: the GPU evaluates every component.
 */
s3a_scorer_t *s3a_scorer_init(s3a_mgau_model_t *g, const int16_t *cd2cisen, int32_t n_sen,
                              int32_t n_ci_sen, int32_t ds_ratio, int32_t cond_ds,
                              double ci_pbeam, float tighten_factor, int32_t max_cd);
/* the same, with the scorer's own bstidx/bstscr/updatetime state: lets several decoders share one
 * s3a_mgau_model_t (all on that model's stream); s3a_mgau_eval / s3a_mgau_state keep using the model's */
s3a_scorer_t *s3a_scorer_init_private(s3a_mgau_model_t *g, const int16_t *cd2cisen, int32_t n_sen,
                              int32_t n_ci_sen, int32_t ds_ratio, int32_t cond_ds,
                              double ci_pbeam, float tighten_factor, int32_t max_cd);
void    s3a_scorer_free(s3a_scorer_t *sc);
/* srch_TST_begin's per-utterance reset of bstidx/updatetime */
int32_t s3a_scorer_utt_begin(s3a_scorer_t *sc);
/*
 * gmm_compute_lv1 slot == approx_ci_gmm_compute (gmm_wrap.c:170-211) ->
 * approx_cont_mgau_ci_eval: score the CI senones of `feat` for frame `fr`
 * into ci_senscr[n_ci_sen] (ascr->cache_ci_senscr[cache_idx]) and *best_score
 * (ascr->cache_best_list[cache_idx]).
 */
int32_t s3a_approx_cont_mgau_ci_eval(s3a_scorer_t *sc, const float *feat, int32_t *ci_senscr,
                                     int32_t *best_score, int32_t fr);
/*
 * gmm_compute_lv2 slot == s3_cd_gmm_compute_sen (gmm_wrap.c:108-167) ->
 * approx_cont_mgau_frame_eval: sen_active[n_sen] in (CI entries are forced
 * to 1, as the reference does), senscr[n_sen] out (normalised by the frame
 * best for active senones), rec_sen_active[n_sen] out.  Returns the frame best
 * through *best (srch->senscale); *n_sen_eval / *n_gau_eval are
 * mgau_frm_sen_eval / mgau_frm_gau_eval for stat_t (may be NULL).
 */
int32_t s3a_approx_cont_mgau_frame_eval(s3a_scorer_t *sc, uint8_t *sen_active,
                                        uint8_t *rec_sen_active, int32_t *senscr,
                                        const float *feat, int32_t frame,
                                        const int32_t *cache_ci_senscr, int32_t *best,
                                        int32_t *n_sen_eval, int32_t *n_gau_eval);

/* ------------------------------------------------------------------ */
/* The SECONDARY boundary: pocketsphinx's continuous scorer, the object behind */
/* the ps_mgaufuncs_t vtable {name, frame_eval, transform, free} (pocketsphinx/src/libpocketsphinx/     */
/* acmod.h:97-110) that acmod_score calls (acmod.c:1076-1131).  Replaces                                 */
/*   ms_mgau_init             ms_mgau.c:75-138  (arguments = the -mean -var -varfloor -mixw -mixwfloor   */
/*                            -senmgau -topn -aw -logbase values it reads from the config)               */
/*   ms_cont_mgau_frame_eval  ms_mgau.c:163-252 (gauden_dist ms_gauden.c:415-541, senone_eval            */
/*                            ms_senone.c:367-421)                                                       */
/* pocketsphinx conventions: float32 arithmetic, scores are int16, NEGATED (smaller = better), scaled by */
/* >> SENSCR_SHIFT (10) and normalised to best = 0; senone_active is the delta-encoded ascending list of */
/* acmod_flags2list (acmod.c:1220-1271).  A pocketsphinx maintainer wraps the handle in a struct whose   */
/* first member is ps_mgau_t {vt, frame_idx} and whose vt->frame_eval forwards here.    */
/* feat = the frame's streams concatenated.                                                              */
/* ------------------------------------------------------------------ */
typedef struct s3a_ps_mgau_s s3a_ps_mgau_t;
s3a_ps_mgau_t *s3a_ps_ms_mgau_init(const char *meanfile, const char *varfile, double varfloor,
                                   const char *mixwfile, double mixwfloor, const char *senmgau,
                                   int32_t topn, int32_t aw, double logbase);
s3a_ps_mgau_t *s3a_ps_ms_mgau_init_arrays(const float *mean, const float *var, const float *mixw,
                                          int32_t n_mgau, int32_t n_feat, int32_t n_density,
                                          const int32_t *featlen, int32_t n_sen, const int32_t *sen2mgau,
                                          double varfloor, double mixwfloor, int32_t topn, int32_t aw,
                                          double logbase);
void    s3a_ps_ms_mgau_free(s3a_ps_mgau_t *ps);
int32_t s3a_ps_ms_mgau_n_sen(const s3a_ps_mgau_t *ps);
int32_t s3a_ps_ms_mgau_veclen(const s3a_ps_mgau_t *ps);
int32_t s3a_ps_ms_cont_mgau_frame_eval(s3a_ps_mgau_t *ps, int16_t *senscr, const uint8_t *senone_active,
                                       int32_t n_senone_active, const float *feat, int32_t frame,
                                       int32_t compallsen);

/* ------------------------------------------------------------------ */
/* pocketsphinx's senone score dump (-senlogdir): the score interchange format between decoders
 * (SURVEY.md 8(f).3).  Replaces acmod_write_senfh_header (pocketsphinx/src/libpocketsphinx/acmod.c:349-361),
 * acmod_write_scores (:885-923), acmod_read_senfh_header (:805-836), acmod_read_scores_internal (:928-985).
 * Host only (file format); the scores are s3a_ps_ms_cont_mgau_frame_eval's int16 outputs and `active` is the
 * delta-encoded list it takes.  Senones a frame does not list read back as SENSCR_DUMMY (0x7fff). */
/* ------------------------------------------------------------------ */
typedef struct s3a_senlog_s s3a_senlog_t;
s3a_senlog_t *s3a_senlog_open_write(const char *path, const char *mdef_file, int32_t n_sen, double logbase);
int32_t s3a_senlog_write_frame(s3a_senlog_t *s, int32_t n_active, const uint8_t *active, const int16_t *senscr);
s3a_senlog_t *s3a_senlog_open_read(const char *path, int32_t *n_sen, double *logbase);
int32_t s3a_senlog_read_frame(s3a_senlog_t *s, int16_t *senscr, uint8_t *active, int32_t *n_active);
void s3a_senlog_close(s3a_senlog_t *s);

/* ------------------------------------------------------------------ */
/* The MFCC front end: 16-bit samples -> cepstra (the step before feat_s2mfc2feat).
 * Replaces fe_t and its whole-utterance use (sphinxbase/include/sphinxbase/fe.h:300-460):
 *   s3a_fe_default_params   the defaults of waveform_to_cepstral_command_line_macro (fe.h:100-215)
 *   s3a_fe_init             fe_init_auto_r (fe_interface.c:212-283) with the options as a struct
 *   s3a_fe_output_size      fe_get_output_size (:302-306)
 *   s3a_fe_n_frames         the frame count of fe_process_utt + fe_end_utt (:470-502)
 *   s3a_fe_process_utt      fe_start_utt + fe_process_utt + fe_end_utt for one utterance: cep
 *                           [n_frames][output_size] float32, the final partial frame included
 *   s3a_fe_process_utt_dev  the same with samples and cepstra in device memory, enqueued on a stream
 * Float64 signal path in the reference's own operation order (its real-FFT schedule included); the one
 * source of differences from the CPU is log() (device library vs libm, both < 1 ulp): see s3a_fe.hip.
 * Not supported (S3A_EUNSUP / not expressible): -dither, -warp_params, swapped input. */
/* ------------------------------------------------------------------ */
enum { S3A_FE_LEGACY = 0, S3A_FE_DCT = 1, S3A_FE_HTK = 2 };           /* -transform */
enum { S3A_FE_CEPSTRA = 0, S3A_FE_LOGSPEC = 1, S3A_FE_SMOOTHSPEC = 2 }; /* -logspec / -smoothspec */
typedef struct {
    float samprate;         /* -samprate */
    int32_t frate;          /* -frate */
    float wlen;             /* -wlen */
    float alpha;            /* -alpha */
    int32_t ncep, nfft, nfilt;
    float lowerf, upperf;
    int32_t transform;      /* S3A_FE_LEGACY / _DCT / _HTK */
    int32_t lifter, remove_dc, round_filters, unit_area, doublebw;
    int32_t logspec;        /* S3A_FE_CEPSTRA / _LOGSPEC / _SMOOTHSPEC */
    int32_t warp_type;      /* -warp_type: S3A_FE_WARP_NONE / _INVERSE (inverse_linear, the default type) / _AFFINE / _PIECEWISE */
    float warp_params[2];   /* -warp_params: a | a b | a F (fe_warp_*.c); a = 0: no warping */
} s3a_fe_params_t;
enum { S3A_FE_WARP_NONE = 0, S3A_FE_WARP_INVERSE = 1, S3A_FE_WARP_AFFINE = 2, S3A_FE_WARP_PIECEWISE = 3 };
typedef struct s3a_fe_s s3a_fe_t;
void s3a_fe_default_params(s3a_fe_params_t *p);
s3a_fe_t *s3a_fe_init(const s3a_fe_params_t *p);
void s3a_fe_free(s3a_fe_t *fe);
int32_t s3a_fe_output_size(const s3a_fe_t *fe);
int32_t s3a_fe_frame_shift(const s3a_fe_t *fe);
int32_t s3a_fe_frame_size(const s3a_fe_t *fe);
int32_t s3a_fe_n_frames(const s3a_fe_t *fe, int64_t nsamps);
int32_t s3a_fe_process_utt(s3a_fe_t *fe, const int16_t *spch, int64_t nsamps, float *cep, int32_t max_frames,
                           int32_t *n_frames);
int32_t s3a_fe_process_utt_dev(s3a_fe_t *fe, const int16_t *spch_dev, int64_t nsamps, float *cep_dev,
                               int32_t max_frames, int32_t *n_frames, void *stream);
void   *s3a_fe_stream(const s3a_fe_t *fe);      /* the HIP stream the front end enqueues on when none is given */

/* ------------------------------------------------------------------ */
/* Feature computation for the stream type "1s_c_d_dd" (the step before the path).
 * Replaces feat_compute_utt for that type: sphinxbase/src/libsphinxbase/feat/feat.c:1111-1123 over the
 * padded utterance of feat_s2mfc_read (:396-516), cmn() feat/cmn.c:141-208 (-cmn current, -varnorm),
 * agc_max() feat/agc.c:109-126 (-agc max), feat_1s_c_d_dd_cep2feat feat.c:726-769.  Bit-exact float32:
 * the cepstral sums run in frame order.  cep = [n_frames][cepsize] cepstra as read from an .mfc file.
 * _dev leaves the features in device memory (rows of feat_stride floats) for s3a_mgau_score_frames_dev. */
/* ------------------------------------------------------------------ */
int32_t s3a_feat_1s_c_d_dd(const float *cep, int32_t n_frames, int32_t cepsize, int32_t cmn_current,
                           int32_t varnorm, int32_t agc_max, float *feat);
int32_t s3a_feat_1s_c_d_dd_dev(const float *cep, int32_t n_frames, int32_t cepsize, int32_t cmn_current,
                               int32_t varnorm, int32_t agc_max, float *feat_dev, int32_t feat_stride,
                               void *stream);

/* Raw audio to features resident in HBM, one call: what utt_decode does with -adcin
 * (sphinx3/src/libs3decoder/libAPI/utt.c:208-233: fe_start_utt + fe_process_utt -- no fe_end_utt, so with
 * drop_partial_frame the samples behind the last whole frame are dropped as there -- then
 * feat_s2mfc2feat_live(beginutt, endutt) = feat_s2mfc2feat_block_utt, feat.c:1241-1265), the cepstra never leaving the
 * device.  *feat_dev_out: *n_frames rows of *feat_stride floats (zero padded: what s3a_uttdec_decode_dev /
 * s3a_uttdec_decode_queue_dev take); release with s3a_dev_free. */
int32_t s3a_audio_to_feat_dev(s3a_fe_t *fe, const int16_t *spch, int64_t nsamps, int32_t drop_partial_frame,
                              int32_t cmn_current, int32_t varnorm, int32_t agc_max, float **feat_dev_out,
                              int32_t *n_frames, int32_t *feat_stride);
/* the same with -cmn prior (cmn_prior.c:143-170 on the whole padded utterance): every frame loses cmn_mean[cepsize] (cmn_t.cmn_mean:
 * what the decoder learnt from earlier utterances); cmn_sum[cepsize] (cmn_t.sum) comes in, takes the padded frames in frame
 * order (float32) and goes out.  The state between utterances stays the caller's (cmn_t.nframe += n_frames + 6, the window
 * shift, cmn_prior_update: cmn_prior.c:95-141). */
int32_t s3a_audio_to_feat_dev_prior(s3a_fe_t *fe, const int16_t *spch, int64_t nsamps, int32_t drop_partial_frame,
                                    const float *cmn_mean, float *cmn_sum, int32_t agc_max, float **feat_dev_out,
                                    int32_t *n_frames, int32_t *feat_stride);
/* feat_lda_transform (sphinxbase feat/lda.c:137-160; -lda / -ldadim) on feature rows resident in HBM (s3a_audio_to_feat_dev's
 * buffer): row <- the first out_dim entries of LDA x row (float32, terms in order, no FMA) in a NEW buffer whose rows are out_dim
 * rounded up to four floats (zero padded: what the engines take for a model of that dimension); the old buffer is freed, *feat_dev
 * and *feat_stride are updated.  lda = feat_t.lda[0]: [out_dim][in_dim] (the file's eigenvectors as rows), host memory. */
int32_t s3a_feat_lda_dev(float **feat_dev, int32_t n_frames, int32_t *feat_stride, const float *lda, int32_t in_dim, int32_t out_dim,
                         void *stream);

/* ------------------------------------------------------------------ */
/* The multi-stream ("s3.0") senone scorer: -senmgau .s3cont. / .semi. */
/* Replaces ms_mgau_model_t and its functions:                         */
/*   ms_mgau_init             libam/ms_mgau.c:149-227                  */
/*     gauden_init + gauden_dist_precompute  ms_gauden.c:330-476       */
/*     senone_init / senone_mixw_read        ms_senone.c:212-417       */
/*   ms_cont_mgau_frame_eval  ms_mgau.c:242-329 (slot gmm_compute_lv2  */
/*     when kbcore holds an ms_mgau, gmm_wrap.c:136-140)               */
/*     gauden_dist (top-N codewords)         ms_gauden.c:541-644       */
/*     senone_eval                           ms_senone.c:442-490       */
/* Same argument list as ms_mgau_init minus the mdef (only needed for  */
/* senone->codebook mapping FILES, which are not supported, like the   */
/* -lambda interpolation file: the call fails).  Scores are the same   */
/* integers as the reference's (determinant accumulated in float32,    */
/* float64 distance chain, ordered top-N list, ordered log-add).       */
/* ------------------------------------------------------------------ */
typedef struct s3a_ms_mgau_s s3a_ms_mgau_t;
s3a_ms_mgau_t *s3a_ms_mgau_init(const char *meanfile, const char *varfile, double varfloor,
                                const char *mixwfile, double mixwfloor, int32_t precomp,
                                const char *senmgau, const char *lambdafile, int32_t topn,
                                s3a_logmath_t *logmath);
/* raw arrays in file order: mean/var [n_mgau][n_feat][n_density][featlen f], mixw
 * [n_sen][n_feat][n_density]; sen2mgau NULL = one codebook per senone (".s3cont.") */
s3a_ms_mgau_t *s3a_ms_mgau_init_arrays(const float *mean, const float *var, const float *mixw,
                                       int32_t n_mgau, int32_t n_feat, int32_t n_density,
                                       const int32_t *featlen, int32_t n_sen, const int32_t *sen2mgau,
                                       double varfloor, double mixwfloor, int32_t topn,
                                       s3a_logmath_t *logmath);
void    s3a_ms_mgau_free(s3a_ms_mgau_t *msg);
int32_t s3a_ms_mgau_n_sen(const s3a_ms_mgau_t *msg);
int32_t s3a_ms_mgau_topn(const s3a_ms_mgau_t *msg);
int32_t s3a_ms_mgau_veclen(const s3a_ms_mgau_t *msg);
/* ms_cont_mgau_frame_eval: sen_active / senscr are ascr_t's arrays (host); feat = the frame's
 * streams concatenated (feat[0] for one stream); senscr[s] is written (normalised) for the active
 * senones only; *best = the value the reference returns (-> srch->senscale). */
int32_t s3a_ms_cont_mgau_frame_eval(s3a_ms_mgau_t *msg, const uint8_t *sen_active, int32_t *senscr,
                                    const float *feat, int32_t frame, int32_t *best);
/* the top-N lists of the last frame, [n_mgau][n_feat][topn] (test hook; inactive codebooks stale) */
int32_t s3a_ms_mgau_get_dist(s3a_ms_mgau_t *msg, int32_t *dist, int32_t *dist_id);

/*
 * dict2pid_comsenscr (sphinx3/src/libs3decoder/libsearch/dict2pid.c:1029-1048):
 * comsenscr[i] = max_{k in comstate[i]} senscr[k] + comwt[i].  The ragged
 * -1-terminated lists d2p->comstate[i] are passed flattened:
 * comstate[comstate_off[i] .. comstate_off[i+1]).
 */
typedef struct s3a_comsen_s s3a_comsen_t;
s3a_comsen_t *s3a_comsen_init(int32_t n_comstate, const int32_t *comstate_off,
                              const int16_t *comstate, const int32_t *comwt);
void    s3a_comsen_free(s3a_comsen_t *cs);
int32_t s3a_dict2pid_comsenscr(s3a_comsen_t *cs, const int32_t *senscr, int32_t n_sen,
                               int32_t *comsenscr);

/* ===================================================================== */
/* transition matrices + batched HMM Viterbi                              */
/* replaces tmat_t (sphinx3/include/tmat.h), hmm_context_t / hmm_t        */
/* (sphinx3/include/hmm.h:156-197) and hmm_vit_eval (libam/hmm.c:855-873) */
/* ===================================================================== */
typedef struct s3a_tmat_s s3a_tmat_t;
/* tmat_init, libam/tmat.c:155-270 (reads, normalises, floors, converts to logs3) */
s3a_tmat_t *s3a_tmat_init(const char *tmatfile, double tpfloor, int32_t breport,
                          s3a_logmath_t *logmath);
s3a_tmat_t *s3a_tmat_init_arrays(const float *tp, int32_t n_tmat, int32_t n_state,
                                 double tpfloor, s3a_logmath_t *logmath);
/* adopt an already-converted matrix set: tp[n_tmat][n_state][n_state+1] logs3 values
 * (what a host that keeps its own tmat_t passes: tmat_t.tp, sphinx3/include/tmat.h) */
s3a_tmat_t *s3a_tmat_init_logs3(const int32_t *tp, int32_t n_tmat, int32_t n_state);
void    s3a_tmat_free(s3a_tmat_t *t);
int32_t s3a_tmat_n_tmat(const s3a_tmat_t *t);
int32_t s3a_tmat_n_state(const s3a_tmat_t *t);
/* copy tp[n_tmat][n_state][n_state+1] logs3 values */
int32_t s3a_tmat_get_tp(const s3a_tmat_t *t, int32_t *tp);

/*
 * A batch of N HMMs in structure-of-arrays form (the device-side replacement
 * of an array of hmm_t).  Scores/histories per state, exit state, ssid (or
 * per-state ssids for multiplex HMMs), tmatid, bestscore.
 */
typedef struct s3a_hmm_batch_s s3a_hmm_batch_t;
/* hmm_context_init (hmm.c:102-121): n_emit_state 3 or 5 run the hard-wired
 * left-to-right updates (hmm.c:285-776), other sizes the any-topology one
 * (hmm.c:779-852).  sseq: [n_sseq][n_emit_state] senone ids (mdef->sseq). */
s3a_hmm_batch_t *s3a_hmm_batch_init(int32_t n_hmm, int32_t n_emit_state, const s3a_tmat_t *tmat,
                                    const int16_t *sseq, int32_t n_sseq, int32_t n_sen);
void    s3a_hmm_batch_free(s3a_hmm_batch_t *b);
/* hmm_init (hmm.c:130-147) for every HMM i: mpx[i], ssid[i], tmatid[i]; clears all */
int32_t s3a_hmm_batch_setup(s3a_hmm_batch_t *b, const uint8_t *mpx, const int32_t *ssid,
                            const int32_t *tmatid);
/* hmm_clear (hmm.c:225-241) on the listed HMMs (idx NULL = all) */
int32_t s3a_hmm_batch_clear(s3a_hmm_batch_t *b, const int32_t *idx, int32_t n);
/* hmm_enter (hmm.c:244-250) on the listed HMMs */
int32_t s3a_hmm_batch_enter(s3a_hmm_batch_t *b, const int32_t *idx, const int32_t *score,
                            const int64_t *histid, int32_t n, int32_t frame);
/* hmm_vit_eval (hmm.c:855-873) on every HMM of the batch against senscr[n_sen]
 * (host); ret[n_hmm] (optional) receives each HMM's return value (bestscore). */
int32_t s3a_hmm_batch_vit_eval(s3a_hmm_batch_t *b, const int32_t *senscr, int32_t *ret);
/* read back: score[n][5] hist[n][5] out_score[n] out_hist[n] bestscore[n]
 * mpx_ssid[n][5] frame[n]; any pointer may be NULL */
int32_t s3a_hmm_batch_get(const s3a_hmm_batch_t *b, int32_t *score, int64_t *hist,
                          int32_t *out_score, int64_t *out_hist, int32_t *bestscore,
                          int32_t *mpx_ssid, int32_t *frame);

/* ===================================================================== */
/* lexical-tree search: the per-frame operations of sphinx3 mode 4       */
/* replaces lextree_t's frame functions, libsearch/lextree.c:910-1663     */
/* ===================================================================== */
/*
 * A "lexsearch" holds ALL lextrees one decoder searches in lock step (mode 4:
 * -Nlextree unigram trees followed by -Nlextree filler trees,
 * srch_time_switch_tree.c:260-456), flattened by the host from lextree_t after
 * lextree_build: per tree t, nodes 0..n_node[t]-1 in any fixed order with
 *   ssid/tmatid/composite/wid/prob    lextree_node_t fields (lextree.h:187-207); wid < 0 = not a leaf
 *   child_off/child                   CSR of ln->children in glist order
 *   n_lc, lc, lcroot_off, lcroot      lextree->lcroot[i].{lc,root} lists in glist order, or
 *   n_root, root                      lextree->root when n_lc == 0 (filler trees)
 * plus what the two hmm_context_t need: tmat, mdef->sseq, d2p->comsseq and the
 * composite-state member lists (d2p->comstate flattened as for s3a_comsen_init).
 * The trees must be static: composite triphones (dict2pid_is_composite, the only
 * mode kbcore.c:626 builds), -pheurtype 0, 3-state HMMs.  History ids are the
 * int32 vithist entry ids.  `stream` (hipStream_t as void*) orders the search
 * kernels with the scorer's; pass s3a_mgau_stream(g).
 */
typedef struct s3a_lexsearch_s s3a_lexsearch_t;
s3a_lexsearch_t *s3a_lexsearch_init(int32_t n_tree, const int32_t *n_node,
        const int32_t *const *ssid, const int32_t *const *tmatid, const uint8_t *const *composite,
        const int32_t *const *wid, const int32_t *const *prob,
        const int32_t *const *child_off, const int32_t *const *child,
        const int32_t *n_lc, const int16_t *const *lc,
        const int32_t *const *lcroot_off, const int32_t *const *lcroot,
        const int32_t *n_root, const int32_t *const *root,
        const s3a_tmat_t *tmat, const int16_t *sseq, int32_t n_sseq,
        const int16_t *comsseq, int32_t n_comsseq, int32_t n_comstate,
        const int32_t *comstate_off, const int16_t *comstate, void *stream);
/* another decoder over the same lextrees: shares proto's static device arrays, owns its state;
 * proto must outlive its clones (several decoders per GPU: s3a_batch_*) */
s3a_lexsearch_t *s3a_lexsearch_clone(const s3a_lexsearch_t *proto, void *stream);
void    s3a_lexsearch_free(s3a_lexsearch_t *ls);
int32_t s3a_lexsearch_reset(s3a_lexsearch_t *ls);
int32_t s3a_lexsearch_n_node(const s3a_lexsearch_t *ls, int32_t tree);
/* lextree_enter (lextree.c:1093-1236) for ALL calls one frame makes into `tree`
 * (srch_utt_word_trans issues one per word-final CI phone, sequentially): lc[c],
 * inscore[c], inhist[c]; results equal the sequential calls in this order.
 * lc is ignored for trees built without left contexts. */
int32_t s3a_lexsearch_enter(s3a_lexsearch_t *ls, int32_t tree, int32_t n_calls, const int32_t *lc,
                            const int32_t *inscore, const int32_t *inhist, int32_t cf,
                            int32_t thresh);
/* lextree_active_swap (lextree.c:1240-1249) on every tree */
int32_t s3a_lexsearch_active_swap(s3a_lexsearch_t *ls);
/* lextree_hmm_eval (lextree.c:1253-1310) on every tree against DEVICE score arrays
 * (s3a_scorer_senscr_dev / s3a_comsen_dev); per tree: best, wbest (MAX_NEG_INT32 when
 * nothing is active) and the number of active HMMs */
int32_t s3a_lexsearch_hmm_eval(s3a_lexsearch_t *ls, const int32_t *senscr_dev,
                               const int32_t *comsen_dev, int32_t frm, int32_t *best,
                               int32_t *wbest, int32_t *n_active);
/* lextree_hmm_propagate_non_leaves (lextree.c:1365-1597) on every tree */
int32_t s3a_lexsearch_propagate_non_leaves(s3a_lexsearch_t *ls, int32_t cf, int32_t th,
                                           int32_t pth, int32_t wth);
/* lextree_hmm_propagate_leaves (lextree.c:1600-1663) minus the vithist_rescore call:
 * per tree t the word exits in active-list order, n_exit[t] of them at
 * exit_*[t*max_per_tree ...]: wid, out_score - prob, out_history -- exactly the
 * arguments the reference passes to vithist_rescore */
int32_t s3a_lexsearch_propagate_leaves(s3a_lexsearch_t *ls, int32_t wth, int32_t *n_exit,
                                       int32_t *exit_wid, int32_t *exit_score,
                                       int32_t *exit_hist, int32_t max_per_tree);
/*
 * One whole search frame with a single host synchronisation: lextree_hmm_eval on every
 * tree, the beam arithmetic of srch_TST_hmm_compute_lv2 (srch_time_switch_tree.c:826-905;
 * hmmbeam/pbeam/wbeam are beam_t.hmm/ptrans/word, phone_uses_wbeam the -ptranskip frames),
 * lextree_hmm_propagate_non_leaves, lextree_hmm_propagate_leaves.  Word exits of all trees
 * come back concatenated in tree order (n_exit[t] each).  extra_dev (optional, 8 int32 in
 * device memory, e.g. s3a_scorer_misc_dev) is copied into res->extra with the same read-back.
 * need_histprune != 0 reports that the frame held more than 1.5 x maxhmmpf HMMs, i.e. the
 * reference would have histogram-pruned (lextree_hmm_histbin): not done on the device yet.
 */
typedef struct s3a_frame_result_s {
    int32_t best_hmm, best_word, n_hmm;         /* beam_t.bestscore / bestwordscore, #active HMMs */
    int32_t thres, phone_thres, word_thres;     /* beam_t.thres / phone_thres / word_thres */
    int32_t need_histprune;
    int32_t n_exit_total;
    int32_t extra[8];
} s3a_frame_result_t;
int32_t s3a_lexsearch_frame_search(s3a_lexsearch_t *ls, const int32_t *senscr_dev,
                                   const int32_t *comsen_dev, int32_t frm, int32_t hmmbeam,
                                   int32_t pbeam, int32_t wbeam, int32_t phone_uses_wbeam,
                                   int32_t maxhmmpf, const int32_t *extra_dev,
                                   s3a_frame_result_t *res, int32_t *n_exit, int32_t *exit_wid,
                                   int32_t *exit_score, int32_t *exit_hist, int32_t max_exits);
/*
 * The FUSED frame (s3a_decoder.hip): the same results as the calls above with a third of
 * the kernel launches -- a mode-4 frame is bounded by its kernel boundaries, not its work.
 *   s3a_decoder_utt_begin   srch_TST_begin's device side (state reset)
 *   s3a_decoder_score       gmm_compute_lv1 + lv2 (CI + gated CD senones; scores stay raw in
 *                           HBM, the frame normaliser is applied inside the search kernels)
 *   s3a_decoder_search      hmm_compute_lv2 + propagate_graph_ph_lv2 + the word-exit half of
 *                           propagate_graph_wd_lv2; ONE synchronisation; results as for
 *                           s3a_lexsearch_frame_search, res->extra = {-, #CD sen, #CD gau,
 *                           #CI sen, #CI gau, CI best, senscale, -}
 *   s3a_decoder_transition  srch_utt_word_trans's lextree_enter calls into one unigram tree
 *                           (n_a calls) and one filler tree (n_b calls; either may be 0), the
 *                           active-senone marks for the coming frame (select_active_gmm) and
 *                           lextree_active_swap.  Must be called once per frame (and once at
 *                           utterance begin with cf = -1), also when there is nothing to enter.
 */
int32_t s3a_decoder_utt_begin(s3a_lexsearch_t *ls, s3a_scorer_t *sc);
/* lextree_hmm_histbin (lextree.c:1314-1358) for one tree: bin[k] += #HMMs with
 * (bestscr - hmm.bestscore) / bw == k (last bin collects the rest; nbin <= 1000), and the
 * tree's active list is REORDERED bin by bin, within a bin in reverse list order, exactly as
 * the reference's glist rebuild does.  s3a_decoder_search applies the same operation to all
 * trees, followed by the bin scan of srch_time_switch_tree.c:870-892, whenever a frame holds
 * more than 1.5 x maxhmmpf HMMs; res->need_histprune reports that it did. */
int32_t s3a_lexsearch_hmm_histbin(s3a_lexsearch_t *ls, int32_t tree, int32_t bestscr, int32_t *bin,
                                  int32_t nbin, int32_t bw);
int32_t s3a_decoder_score(s3a_scorer_t *sc, const float *feat, int32_t frame);
int32_t s3a_decoder_search(s3a_lexsearch_t *ls, s3a_scorer_t *sc, s3a_comsen_t *cs, int32_t frm,
                           int32_t hmmbeam, int32_t pbeam, int32_t wbeam, int32_t phone_uses_wbeam,
                           int32_t maxhmmpf, s3a_frame_result_t *res, int32_t *n_exit,
                           int32_t *exit_wid, int32_t *exit_score, int32_t *exit_hist,
                           int32_t max_exits);
int32_t s3a_decoder_transition(s3a_lexsearch_t *ls, s3a_scorer_t *sc, s3a_comsen_t *cs, int32_t cf,
                               int32_t thresh, int32_t tree_a, int32_t n_a, const int32_t *lc_a,
                               const int32_t *scr_a, const int32_t *hist_a, int32_t tree_b,
                               int32_t n_b, const int32_t *lc_b, const int32_t *scr_b,
                               const int32_t *hist_b);
/*
 * The BATCHED fused frame (s3a_batch.hip): B decoders share every kernel launch and the one
 * synchronisation of a step.  Utterances are independent, a single decoder's
 * frame is a chain of latency-bound launches that leaves the chip idle, and HIP streams only
 * overlap as far as the hardware queues go; batching the launches is what scales decoders per
 * GPU.  Each decoder keeps its own handles (s3a_lexsearch_t / s3a_scorer_t / s3a_comsen_t), its
 * own host thread and the reference's host code; results are those of s3a_decoder_*.
 *   s3a_batch_attach      register a decoder (all its work moves to the engine's stream) -> slot
 *   s3a_batch_utt_begin   srch_TST_begin's device side; the slot now takes part in the steps
 *   s3a_batch_transition  the frame's lextree_enter calls + lextree_active_swap: RECORDED (host
 *                         only), executed at the head of the slot's next step; required before
 *                         every step, also at utterance begin (cf = -1)
 *   s3a_batch_step        this frame's scoring + search for the slot (arguments as
 *                         s3a_decoder_score + s3a_decoder_search); BLOCKS until every slot that is
 *                         inside an utterance has called it; the last arrival runs the step
 *   s3a_batch_utt_end     lextree_utt_end; the slot leaves the steps (others no longer wait for it)
 *   s3a_batch_submit/run  the same step for single-threaded drivers: submit every active slot, run
 */
typedef struct s3a_batch_s s3a_batch_t;
s3a_batch_t *s3a_batch_create(int32_t max_slots);
void    s3a_batch_free(s3a_batch_t *b);
int32_t s3a_batch_attach(s3a_batch_t *b, s3a_lexsearch_t *ls, s3a_scorer_t *sc, s3a_comsen_t *cs);
int32_t s3a_batch_utt_begin(s3a_batch_t *b, int32_t slot);
int32_t s3a_batch_utt_end(s3a_batch_t *b, int32_t slot);
int32_t s3a_batch_transition(s3a_batch_t *b, int32_t slot, int32_t cf, int32_t thresh, int32_t tree_a,
                             int32_t n_a, const int32_t *lc_a, const int32_t *scr_a, const int32_t *hist_a,
                             int32_t tree_b, int32_t n_b, const int32_t *lc_b, const int32_t *scr_b,
                             const int32_t *hist_b);
int32_t s3a_batch_step(s3a_batch_t *b, int32_t slot, const float *feat, int32_t frame, int32_t frm,
                       int32_t hmmbeam, int32_t pbeam, int32_t wbeam, int32_t phone_uses_wbeam,
                       int32_t maxhmmpf, s3a_frame_result_t *res, int32_t *n_exit, int32_t *exit_wid,
                       int32_t *exit_score, int32_t *exit_hist, int32_t max_exits);
int32_t s3a_batch_submit(s3a_batch_t *b, int32_t slot, const float *feat, int32_t frame, int32_t frm,
                         int32_t hmmbeam, int32_t pbeam, int32_t wbeam, int32_t phone_uses_wbeam,
                         int32_t maxhmmpf, s3a_frame_result_t *res, int32_t *n_exit, int32_t *exit_wid,
                         int32_t *exit_score, int32_t *exit_hist, int32_t max_exits);
int32_t s3a_batch_run(s3a_batch_t *b);
int32_t s3a_batch_stats(s3a_batch_t *b, int64_t *steps, int64_t *slot_frames);
/* srch_TST_select_active_gmm (srch_time_switch_tree.c:1262-1324): clear, then mark the
 * senones of every active HMM (composite ones through their member lists) in a DEVICE
 * flag array of n_sen bytes (s3a_scorer_sen_active_dev) */
int32_t s3a_lexsearch_sen_active(s3a_lexsearch_t *ls, uint8_t *sen_active_dev, int32_t n_sen);
/* lextree_utt_end (lextree.c:936-961) on every tree */
int32_t s3a_lexsearch_utt_end(s3a_lexsearch_t *ls);
/* read-back for tests: which = 0 active list, 1 next_active list (node ids local to the tree) */
int32_t s3a_lexsearch_get_active(const s3a_lexsearch_t *ls, int32_t tree, int32_t which,
                                 int32_t *n_active, int32_t *nodes, int32_t max_nodes);
/* HMM state of one tree: score/hist [3][n_node], the rest [n_node]; any may be NULL */
int32_t s3a_lexsearch_get_hmm(const s3a_lexsearch_t *ls, int32_t tree, int32_t *score,
                              int32_t *hist, int32_t *out_score, int32_t *out_hist,
                              int32_t *bestscore, int32_t *frame);

/* device-resident scoring for a search that lives on the GPU: the scorer's own
 * sen_active / senscr buffers and the composite scores never leave HBM */
void   *s3a_mgau_stream(s3a_mgau_model_t *g);
uint8_t *s3a_scorer_sen_active_dev(s3a_scorer_t *sc);
int32_t *s3a_scorer_senscr_dev(s3a_scorer_t *sc);
int32_t *s3a_comsen_dev(s3a_comsen_t *cs);
/* approx_cont_mgau_frame_eval with sen_active taken from, and senscr left in, the
 * scorer's device buffers; cs (optional) then receives dict2pid_comsenscr on device.
 * -maxcdsenpf's dynamic CI beam needs the host mask and is rejected here. */
/* fully asynchronous variant: also computes the CI senones on the device (no host cache),
 * reads nothing back; misc (8 int32, device): [1] #CD senones evaluated [2] #CD Gaussians
 * [3] #CI senones [4] #CI Gaussians [5] CI best [6] frame normaliser (srch->senscale) */
int32_t s3a_approx_cont_mgau_frame_eval_async(s3a_scorer_t *sc, s3a_comsen_t *cs,
                                              const float *feat, int32_t frame);
int32_t *s3a_scorer_misc_dev(s3a_scorer_t *sc);
int32_t s3a_approx_cont_mgau_frame_eval_dev(s3a_scorer_t *sc, s3a_comsen_t *cs, const float *feat,
                                            int32_t frame, const int32_t *cache_ci_senscr,
                                            int32_t *best, int32_t *n_sen_eval,
                                            int32_t *n_gau_eval);

/* ===================================================================== */
/* the word level + whole utterances on the device (s3a_utt.hip)          */
/* ===================================================================== */
/*
 * with the word level on the device an utterance needs no host between its
 * first and its last frame -- the `decode` slot of srch_funcs_t (sphinx3/include/srch.h:552-555;
 * srch_utt_decode_blk hands the whole block to it, libsearch/srch.c:673-675).
 *
 * s3a_lm3g_t = lm_t (sphinx3/include/lm.h:559-660) flattened by the caller: unigram w has prob
 * ug_prob[w], back-off ug_bowt[w] and the bigrams [ug_firstbg[w], ug_firstbg[w+1]); bigram b has
 * second word bg_wid[b], prob bg_prob[b] (= lm->bgprob[bg.probid].l), back-off bg_bowt[b]
 * (= lm->tgbowt[bg.bowtid].l) and the trigrams [bg_firsttg[b], bg_firsttg[b+1]) (absolute:
 * tg_segbase[b >> log_bg_seg_sz] + bg.firsttg, lm.c:1435-1443); trigram t has third word tg_wid[t]
 * and prob tg_prob[t].  All values as lm_set_param left them (language weight, insertion penalty
 * applied; lm.c:355-388).  Runs must be sorted and duplicate-free (checked).  n_bg = 0 / n_tg = 0
 * select lm->ugonly / lm->bgonly behaviour.  inclass_ugscore (per DICTIONARY word, or NULL) =
 * lm->inclass_ugscore of class-based LMs.  "No LM word" (BAD_LMWID) is any negative id.
 * s3a_lm3g_tg_score is lm_tg_score (lm.c:1661-1833) on the host copy.
 */
typedef struct s3a_lm3g_s s3a_lm3g_t;
s3a_lm3g_t *s3a_lm3g_init(int32_t n_ug, const int32_t *ug_prob, const int32_t *ug_bowt,
                          const int32_t *ug_firstbg, int32_t n_bg, const int32_t *bg_wid,
                          const int32_t *bg_prob, const int32_t *bg_bowt, const int32_t *bg_firsttg,
                          int32_t n_tg, const int32_t *tg_wid, const int32_t *tg_prob,
                          const int32_t *inclass_ugscore, int32_t n_dictword);
/* the same without device arrays: for the library's host-side consumers (s3a_lattice_nbest, s3a_lm3g_tg_score) on a machine without
 * a GPU; the engines refuse such a handle */
s3a_lm3g_t *s3a_lm3g_init_host(int32_t n_ug, const int32_t *ug_prob, const int32_t *ug_bowt,
                               const int32_t *ug_firstbg, int32_t n_bg, const int32_t *bg_wid,
                               const int32_t *bg_prob, const int32_t *bg_bowt, const int32_t *bg_firsttg,
                               int32_t n_tg, const int32_t *tg_wid, const int32_t *tg_prob,
                               const int32_t *inclass_ugscore, int32_t n_dictword);
void s3a_lm3g_free(s3a_lm3g_t *lm);
int32_t s3a_lm3g_tg_score(const s3a_lm3g_t *lm, int32_t lw1, int32_t lw2, int32_t lw3, int32_t wid);

/* What the word level reads of dict_t / fillpen_t / mdef_t / vithist_t / histprune_t / beam_t /
 * srch_TST_graph_t (the caller's arrays are copied). */
typedef struct {
    int32_t n_word, n_ci;           /* dict_size, mdef_n_ciphone */
    const int32_t *lwid;            /* [n_word] lm->dict2lmwid[w], negative = none */
    const uint8_t *is_filler;       /* [n_word] dict_filler_word */
    const int32_t *fillpen;         /* [n_word] fillpen(kbcore_fillpen, w) of filler words */
    const int32_t *last_ci;         /* [n_word] dict_last_phone, filler phones mapped to mdef_silphone */
    int32_t startwid, finishwid, silwid;    /* dict_startwid / _finishwid / _silwid */
    int32_t start_lwid, finish_lwid;        /* lm_startwid / lm_finishwid */
    int32_t sil_ci;                 /* mdef_silphone */
    int32_t wbeam_vh, bghist;       /* vithist_t.wbeam, .bghist (vithist.c:160-190) */
    int32_t maxwpf, maxhistpf;      /* histprune_t */
    int32_t wordend_beam;           /* beam_t.wordend */
    int32_t n_lextree, epl;         /* srch_TST_graph_t.n_lextree, .epl */
    int32_t hmmbeam, pbeam, wbeam;  /* beam_t.hmm, .ptrans, .word */
    int32_t ptranskip, maxhmmpf;    /* beam_t.ptranskip, histprune_t.maxhmmpf */
    const int32_t *tree_type;       /* [2 * n_lextree] lextree_t.type of the lexsearch's trees */
} s3a_wordlevel_cfg_t;

/*
 * s3a_uttdec_t: n_lanes decoder lanes over ONE model (proto's lextrees are shared, every lane gets
 * its own search state, senone state and history table; scorer arguments as s3a_scorer_init).
 * max_frames bounds an utterance; vh_cap (history entries per utterance, 0 = max_frames x
 * maxhistpf) and cand_cap ((word exit, predecessor) pairs per frame, 0 = 1 M) size the buffers --
 * exceeding them ends the utterance with an error, never silently.
 *   s3a_uttdec_decode   n_utt <= n_lanes utterances, feat[z] = n_frames[z] rows of feat_stride
 *                       floats (host): uploads, runs every frame of every utterance on the
 *                       device (srch_TST_begin .. the last frame_windup) and reads the history
 *                       tables back.  Blocks.
 *   s3a_uttdec_result   the finished table of lane z as read-only arrays (valid until the next
 *                       decode): vithist_entry_t fields by entry id, frame_start / bestscore /
 *                       bestvh by frame (n_frm + 1 values), frame_stat[f] = {senscale (->
 *                       srch->ascale[f]), #HMMs, #CD senones, #CD Gaussians, #CI senones, #CI
 *                       Gaussians, histogram pruning applied, #word exits}.
 */
typedef struct s3a_uttdec_s s3a_uttdec_t;
typedef struct {
    int32_t err, n_entry, n_frm, n_frames;
    const int32_t *score, *pred, *lw0, *lw1, *wid, *sf, *ef, *ascr, *lscr, *type;
    const int32_t *frame_start, *bestscore, *bestvh;
    const int32_t *frame_stat;
    int32_t max_cand, max_new, n_tie_frames;
} s3a_utt_result_t;
s3a_uttdec_t *s3a_uttdec_init(const s3a_lexsearch_t *proto, s3a_mgau_model_t *g, const int16_t *cd2cisen,
                              int32_t n_sen, int32_t n_ci_sen, int32_t ds_ratio, int32_t cond_ds,
                              double ci_pbeam, float tighten_factor, int32_t max_cd, s3a_comsen_t *cs,
                              s3a_lm3g_t *lm, const s3a_wordlevel_cfg_t *cfg, int32_t n_lanes,
                              int32_t max_frames, int32_t vh_cap, int32_t cand_cap);
/* The same with the engine's tuning options as arguments (all zero / -1 = the library's choice; every variant gives the
 * same bits): many = lanes from which the grids / kernels for many lanes per launch are used (default 32); big_wl = the word
 * level's candidate phases as launches of their own (-1: by the beams' width); window = frames per look-ahead scoring pass
 * (-1: 8, or up to window_max for ~1024 (lane, frame) slots per pass; 0: per-frame scoring kernels), window_fpc = slots per workgroup chunk of that pass;
 * g_eval / g_res / scan_g / gy / sweep_k = grid sizes of the HMM evaluation, the resolve kernels, the scan, the gated scorer and
 * the nodes per thread of the resolve sweep; no_multi = no shared CD pass of the per-frame scorer; framecheck = run the
 * per-frame invariant kernel; times = print the host-side phases of every decode to stderr; graph = HIP-graph replay of the
 * frames' launches (below).  s3a_uttdec_init is
 * s3a_uttdec_init_opts with opts = NULL (defaults); the library never reads the environment for these:
 * s3a_uttdec_opts_from_env fills the struct from the S3A_UTT_* variables for hosts that want that (the drop-in program,
 * the test harness). */
typedef struct {
    int32_t many, big_wl, window, window_fpc, g_eval, g_res, scan_g, gy, sweep_k, no_multi, framecheck, times;
    int32_t graph;          /* 1: the frames' launches are captured ONCE as a HIP graph (a block of `window` frames per lane count)
                             * and replayed block after block: one graph launch per block instead of ~13 kernel launches per frame */
    int32_t window_max;     /* > 0: the longest look-ahead window the library may choose (lane refill happens at window boundaries) */
    int32_t scan_small_from; /* lanes per launch from which the scan uses 256-thread workgroups (0: 64) */
    int32_t persist;        /* the frames of a look-ahead window as ONE launch (ku_frames: a lane = a persistent 512-thread workgroup, or
                             * a cluster of them, that walks the frame's steps with barriers instead of launch boundaries -- the
                             * reference's srch.c:746-835 loop has none): 0 = whenever the engine's configuration is served
                             * (look-ahead scoring on, no wide-beam word level), -1 = never (the twelve
                             * launches per frame of rounds 2-4), 1 = whatever the lane count (0 keeps the launches for few lanes), 2 = as 1 with
                             * the clusters' general (agent-scope) barrier only, never the XCD-local one (A/B runs).  Same bits either way. */
    int32_t cluster;        /* workgroups per lane of that launch: 0 = the library's choice (1 when the lanes fill the chip, more
                             * -- on one XCD, with a counter barrier between the steps -- when they do not; only an engine that is
                             * alone on its device chooses more than 1: the clusters of one launch must be resident together);
                             * > 0 = that many where they fit, at most 32 (the caller then answers for co-residency).  "Alone" counts
                             * other processes too: the first process with such an engine on a device holds an advisory lock
                             * (/dev/shm/cmusphinx_amd.kf.<PCI bus id>.lock) and the others keep one workgroup per lane */
    int32_t score_rows_max; /* > 0: a cap on the rows (frames) of senone scores ku_frames' calls keep on the device at once (default:
                             * half of the free device memory).  A queue whose frames exceed it goes through in parts (consecutive
                             * utterances whose rows fit; an utterance longer than the cap is an error); s3a_uttdec_decode falls back
                             * to window blocks (K frames per launch from the look-ahead rows).  Same bits either way. */
    int32_t reserved[1];
} s3a_uttdec_opts_t;
void s3a_uttdec_opts_default(s3a_uttdec_opts_t *o);
void s3a_uttdec_opts_from_env(s3a_uttdec_opts_t *o);
s3a_uttdec_t *s3a_uttdec_init_opts(const s3a_lexsearch_t *proto, s3a_mgau_model_t *g, const int16_t *cd2cisen,
                                   int32_t n_sen, int32_t n_ci_sen, int32_t ds_ratio, int32_t cond_ds, double ci_pbeam,
                                   float tighten_factor, int32_t max_cd, s3a_comsen_t *cs, s3a_lm3g_t *lm,
                                   const s3a_wordlevel_cfg_t *cfg, int32_t n_lanes, int32_t max_frames, int32_t vh_cap,
                                   int32_t cand_cap, const s3a_uttdec_opts_t *opts);
void s3a_uttdec_free(s3a_uttdec_t *ud);
int32_t s3a_uttdec_decode(s3a_uttdec_t *ud, int32_t n_utt, const float *const *feat,
                          const int32_t *n_frames, int32_t feat_stride);
/* the same with the features already resident in HBM: rows of feat_stride = 4 * ceil(veclen / 4) floats, zero padded */
int32_t s3a_uttdec_decode_dev(s3a_uttdec_t *ud, int32_t n_utt, const float *const *feat_dev,
                              const int32_t *n_frames, int32_t feat_stride);
int32_t s3a_uttdec_result(s3a_uttdec_t *ud, int32_t lane, s3a_utt_result_t *out);
/* A QUEUE of utterances with lane refill -- what ctl_process (libcommon/corpus.c:538-640) is to the reference: any
 * number of utterances, no coupling between them.  The first n_lanes start together; a lane whose utterance has ended
 * (srch_utt_end, srch.c:482-560) takes the queue's next one (srch_utt_begin, srch.c:453-479: every per-utterance state
 * reset) at the next boundary of the look-ahead scoring window (at the next frame without it) while the other lanes go
 * on.  The lengths are known up front, so the host makes the whole schedule before the first launch and only enqueues;
 * an utterance that stops on an error (a capacity of its lane) is reported, its lane is scrubbed on the device and takes
 * the next utterance.  Results per UTTERANCE of the queue (its index in feat[]): s3a_uttdec_queue_hyp (header + words, as
 * s3a_uttdec_hyp_var), s3a_uttdec_queue_status.  The history tables are reused by the lanes' next utterances:
 * s3a_uttdec_result and the second pass (s3a_uttdec_enable_bestpath: refused) need s3a_uttdec_decode.  Returns the first
 * stopped utterance's error (the others' results are complete). */
int32_t s3a_uttdec_decode_queue(s3a_uttdec_t *ud, int32_t n_utt, const float *const *feat,
                                const int32_t *n_frames, int32_t feat_stride);
int32_t s3a_uttdec_decode_queue_dev(s3a_uttdec_t *ud, int32_t n_utt, const float *const *feat_dev,
                                    const int32_t *n_frames, int32_t feat_stride);
int32_t s3a_uttdec_queue_status(s3a_uttdec_t *ud, int32_t utt, int32_t *err, int32_t *stopped_at,
                                int32_t *max_cand, int32_t *max_new);
/* the schedule s3a_uttdec_decode_queue follows, host arithmetic on the lengths alone (no device): utterance u runs in
 * lane[u] from engine frame f0[u] on; lanes are refilled at multiples of `boundary` frames (the engine's: its look-ahead
 * window, s3a_uttdec_window; 1 without look-ahead scoring).  Returns the engine frames the whole queue takes. */
int32_t s3a_queue_schedule(int32_t n_lanes, int32_t boundary, int32_t n_utt, const int32_t *n_frames,
                           int32_t *lane, int32_t *f0);
/* diagnostics: time lane z's last utterance spent in each phase of the one-workgroup word level, in 100 MHz ticks
 * ([0] frame record + exits, [1] P1, [2] P2 trigram scores, [3] P3 hash insert, [4] P4 entry places, [5] P5 staging,
 * [6] pruning, [7] table + LM contexts, [8] word transitions) */
int32_t s3a_uttdec_wl_ticks(s3a_uttdec_t *ud, int32_t lane, long long *out16);
/* diagnostics: the same for the steps of a frame inside ku_frames (s3a_uttdec_opts_t.persist), of the utterance the lane decoded last:
 * [0] lextree_enter's entry test (lextree.c:1093-1236), [1] its ranking, [2] the entries applied + the senone marks
 * (srch_time_switch_tree.c:1101-1153), [3] composite senones' members, [4] the CI gate (approx_cont_mgau.c:188-284), [5] composite
 * maxima, [6] lextree_hmm_eval, [7] stamps of the propagating HMMs / histogram (lextree.c:1314-1358), [8] -ptranskip's weak HMMs,
 * [9] lextree_hmm_propagate_non_leaves, [10] the ordered scan, [11] emission + word level ([15]: the emission alone when the lane is
 * one workgroup), [12] ticks inside the launches, [13] frames, [14] launches.  *cluster = workgroups per lane of the last launch
 * (0: the engine runs the frame as separate launches). */
/* measurement: where the last decode's device time went when it ran through ku_frames (HIP events on the engine's stream): the
 * up-front scoring (ku_score_window: milliseconds, launches) and ku_frames itself; *cluster = workgroups per lane.  All zero when
 * the call ran the frame as separate launches (fewer lanes than pay for ku_frames, an option it does not serve, persist = -1). */
int32_t s3a_uttdec_last_parts(s3a_uttdec_t *ud, double *score_ms, int32_t *n_score, double *frames_ms, int32_t *n_frames, int32_t *cluster);
int32_t s3a_uttdec_frame_ticks(s3a_uttdec_t *ud, int32_t lane, long long *out16, int32_t *cluster);
/* diagnostics: launches the last call's RELAY had behind its first (0: one launch to the end).  A ku_frames call ends with its slowest
 * lane; when no utterance is left to take and few lanes are still at work, they hand over at a frame boundary to a launch that continues
 * them as clusters of 2, then 4 workgroups (s3a_variants_t.kf_no_relay switches that off; only an engine alone on its device does it) */
int32_t s3a_uttdec_last_relay(const s3a_uttdec_t *ud);
/* diagnostics: where and when the lane's workgroup ran the launch that began at engine frame 512: clock at entry and exit (100 MHz),
 * the hardware's HW_ID and XCC_ID registers (tools/kf_phases.py --placement) */
int32_t s3a_uttdec_frame_dbg(s3a_uttdec_t *ud, int32_t lane, long long *out4);
int32_t s3a_uttdec_n_lanes(const s3a_uttdec_t *ud);
/* diagnostics: lextree_utt_end on every lane, then how many node records of `lane` are NOT an inactive HMM
 * (out[0..5]: state scores 0/1/2, exit score, best score, frame tag; out[6] the first such node, INT_MAX: none;
 * out[7] propagation scratch left set) -- all clean on a healthy lane between utterances */
int32_t s3a_uttdec_selfcheck(s3a_uttdec_t *ud, int32_t lane, int32_t *out8);
/* frames per look-ahead scoring window (one pass over the acoustic model scores every senone of the next K frames of
 * all lanes; approx_cont_mgau_frame_eval's gate is then applied per frame); 0: per-frame scoring kernels */
int32_t s3a_uttdec_window(const s3a_uttdec_t *ud);
/*
 * Phoneme look-ahead (-pheurtype 1..3, -pl_window, -pl_beam; sphinx3/include/cmdln_macro.h:255-267): the CI senones of
 * the next pl_window frames give every CI phone a heuristic score (pl_computePhnHeur, libam/fast_algo_struct.c:219-300:
 * 1 = sum of the phone's best senone, 2 = "sum of averages", 3 = type 1 plus the reference's first-senone terms), and
 * lextree_hmm_propagate_non_leaves (libsearch/lextree.c:1443-1486) enters a child only if score + heuristic of the child's
 * phone reaches the running maximum of that figure over the active list + pl_beam.  pl_beam = logs3(-pl_beam);
 * node_ci[t][i] = lextree_node_t.ci of node i of tree t (the trees of s3a_lexsearch_init, in its order);
 * sen2cimap = mdef_t.sen2cimap[0 .. n_ci_sen] (one entry past the CI senones, as the reference reads it).
 * Together with a phone threshold below the HMM threshold (-ptranskip frames, -pbeam wider than -beam) the engine settles the
 * two dependencies along the list in one kernel (ku_weak_heur; lextree.c:1424-1458).  pheurtype 0: off.
 */
int32_t s3a_uttdec_enable_pheur(s3a_uttdec_t *ud, int32_t pheurtype, int32_t pl_beam, int32_t pl_window,
                                const uint8_t *const *node_ci, const int16_t *sen2cimap, int32_t n_ci);
/*
 * The hypothesis of a finished lane as a fixed-size record -- what one utterance contributes to the end-of-batch
 * gather when the control file is sharded over GPUs ({uttid, words, sf/ef, ascr, lscr, score,
 * n_frames}; one all-gather of these records, no per-frame collective).  s3a_uttdec_hyp = vithist_utt_end
 * (vithist.c:766-860: the final </s> transition, the silence patch when the last frames have no exit) +
 * vithist_backtrace (:1066-1100) on the lane's table; word[].scale = the frame normalisers over [sf, ef)
 * (compute_scale, srch_output.c:52-60) -- computed ON THE DEVICE behind the utterance's last frame, for all lanes in one
 * launch: a decode brings back 32 bytes + 24 bytes per word per utterance; the history tables and per-frame statistics
 * cross to the host only when s3a_uttdec_result asks for them (the first call after a decode fetches all its lanes').  status: 0 ok, -1 decode error, -2 no word exit, -3 more than
 * S3A_HYP_MAXW words.  s3a_hyp_format = match_write / matchseg_write (srch_output.c:74-161): the utterance's
 * -hyp and -hypseg lines (wordstr / basewid / is_filler by dictionary word id; lw, wip = lm_t.lw / .wip for
 * lm_rawscore, lm.c:2171-2178; unscale = -hypsegscore_unscale).
 */
#define S3A_HYP_MAXW 250
typedef struct { int32_t wid, sf, ef, ascr, lscr, scale; } s3a_hyp_word_t;
typedef struct {
    char uttid[96];
    int32_t utt_index, n_words, n_frames, score, total_scale, n_entry, status, exit_id;
    s3a_hyp_word_t word[S3A_HYP_MAXW];
} s3a_hyp_record_t;
int32_t s3a_uttdec_hyp(s3a_uttdec_t *ud, int32_t lane, const char *uttid, int32_t utt_index,
                       s3a_hyp_record_t *rec);
int32_t s3a_hyp_format(const s3a_hyp_record_t *rec, const char *const *wordstr, const int32_t *basewid,
                       const uint8_t *is_filler, int32_t startwid, int32_t finishwid, float lw, int32_t wip,
                       int32_t unscale, char *match_line, size_t match_cap, char *seg_line, size_t seg_cap);
/* The same without a word limit (a ten-minute utterance has thousands of words): a fixed-size header -- what
 * s3a_hyp_record_t begins with -- and as many s3a_hyp_word_t as the hypothesis has.  The end-of-batch exchange is
 * then two collectives: the headers (every rank learns every n_words), then the words, padded per rank to the
 * largest rank's total.  max_words too small (0 to ask): status -3 and hdr->n_words = the count it takes. */
typedef struct {
    char uttid[96];
    int32_t utt_index, n_words, n_frames, score, total_scale, n_entry, status, exit_id;
} s3a_hyp_header_t;
int32_t s3a_uttdec_hyp_var(s3a_uttdec_t *ud, int32_t lane, const char *uttid, int32_t utt_index,
                           s3a_hyp_header_t *hdr, s3a_hyp_word_t *words, int32_t max_words);
int32_t s3a_uttdec_queue_hyp(s3a_uttdec_t *ud, int32_t utt, const char *uttid, int32_t utt_index,
                             s3a_hyp_header_t *hdr, s3a_hyp_word_t *words, int32_t max_words);
int32_t s3a_hyp_format_var(const s3a_hyp_header_t *hdr, const s3a_hyp_word_t *words, const char *const *wordstr,
                           const int32_t *basewid, const uint8_t *is_filler, int32_t startwid, int32_t finishwid,
                           float lw, int32_t wip, int32_t unscale, char *match_line, size_t match_cap,
                           char *seg_line, size_t seg_cap);

/* The word level on its own (one lane, no lextrees): what closes a frame -- vithist_rescore for
 * every word exit, vithist_prune, srch_utt_word_trans, vithist_frame_windup -- on caller-supplied
 * exits; the parity tests drive it frame by frame next to the oracle.  cfg: dictionary / pruning
 * fields of s3a_wordlevel_cfg_t; lcmap_len[t * (n_ci + 1) + p] = length of the root list
 * lextree_enter(tree t, left context p) walks (p == n_ci: no context), negative: not a context. */
typedef struct s3a_wltest_s s3a_wltest_t;
s3a_wltest_t *s3a_wltest_init(s3a_lm3g_t *lm, const s3a_wordlevel_cfg_t *cfg, const int32_t *lcmap_len,
                              int32_t vh_cap, int32_t cand_cap, int32_t max_exits, int32_t max_frames);
void s3a_wltest_free(s3a_wltest_t *wt);
int32_t s3a_wltest_begin(s3a_wltest_t *wt, int32_t startwid, int32_t n_frames);
int32_t s3a_wltest_frame(s3a_wltest_t *wt, const int32_t *n_exit, const int32_t *exits, int32_t best_hmm,
                         int32_t best_word, int32_t word_thres, int32_t *n_calls, int32_t *calls,
                         int32_t *thresh, int32_t *n_ent);
int32_t s3a_wltest_fetch(s3a_wltest_t *wt, int32_t *n_entry, int32_t *n_frm, int32_t *out,
                         int32_t max_entries, int32_t *frames, int32_t max_out_frames,
                         int32_t *n_tie_frames);

/* ===================================================================== */
/* the end-of-batch exchange of hypotheses in C over RCCL  */
/* ===================================================================== */
/*
 * The control file sharded over the ranks (-ctloffset / -ctlcount, main_decode.c:164-169; ctl_process, corpus.c:538): every
 * rank holds the hypotheses of its own utterances (s3a_uttdec_hyp_var / s3a_uttdec_bestpath_hyp: header + words),
 * s3a_gather_hyps brings ALL of them to every rank in utterance order -- three all-gathers over RCCL on device buffers
 * (counts; headers; words, each padded to the largest rank's) -- and s3a_gather_result hands them out (owned by the
 * gather until its next call).  RCCL is loaded at run time; the communicator is bootstrapped through the file
 * `rendezvous` (rank 0 writes ncclGetUniqueId's bytes; one node, one file system).  words = the local utterances' words
 * back to back (status != 0: none).  n_total = utterances of the whole batch: each index exactly once, else an error.
 */
/* (A test double: when the loaded librccl.so exports the symbol `s3a_comm_takes_host_pointers` -- tests/mock_rccl.c does, RCCL does not --
 * the staging buffers handed to ncclAllGather are host memory, so that the exchange's arithmetic with world > 1 can run on a box
 * without GPUs: tests/test_gather_mock.py.  No product path sets it.) */
typedef struct s3a_gather_s s3a_gather_t;
s3a_gather_t *s3a_gather_init(int32_t rank, int32_t world, const char *rendezvous);
/* The same with a RUN ID (any non-zero number the launcher gives every rank of ONE run and not to the next: a job id, a number drawn
 * per launch -- NOT a value that repeats from run to run such as a fixed MASTER_PORT):
 * the file is `rendezvous`.<run id>, its first word is the id, rank 0 removes what an earlier run of the same name left, and no
 * clock is compared -- ranks may start at any time after one another (staggered or containerised launches, a restarted worker,
 * a shared file system whose server keeps another time).  s3a_gather_init without an id recognises a stale file by its age: it
 * takes a file written no longer than 120 s before the calling process started, so the ranks of a run must start within that
 * window of one another and see one clock; prefer s3a_gather_init_run.
 * Either way the file lives only until every rank holds the id: the init ends with one small all-gather (nobody leaves it before
 * everybody has read the file) behind which rank 0 removes the file, so a successful run leaves nothing for the next one to find. */
s3a_gather_t *s3a_gather_init_run(int32_t rank, int32_t world, const char *rendezvous, unsigned long long run_id);
void s3a_gather_free(s3a_gather_t *g);
int32_t s3a_gather_hyps(s3a_gather_t *g, int32_t n_local, const s3a_hyp_header_t *hdr, const s3a_hyp_word_t *words,
                        int32_t n_total);
int32_t s3a_gather_result(const s3a_gather_t *g, int32_t utt_index, const s3a_hyp_header_t **hdr,
                          const s3a_hyp_word_t **words);

/* ===================================================================== */
/* the second pass: lattice from the Viterbi history, best path under the trigram          */
/* ===================================================================== */
/*
 * vithist_dag_build (libsearch/vithist.c:1100-1311), dag_bypass_filler_nodes (dag.c:1037-1075), dag_search /
 * dag_bestpath (dag.c:893-965, 397-484), dag_backtrace (dag.c:590-671) as srch_TST_bestpath_impl drives them
 * (srch_time_switch_tree.c:1391-1440), one workgroup per utterance, all lanes at once.  cfg: per dictionary word
 * basewid (dict_basewid), is_filler (dict_filler_word), lwid (lm->dict2lmwid), fillpen (fillpen() of the filler words);
 * wip = logs3(fillpen_t.wip); lwf = -bestpathlw / -lw (1.0 when -bestpathlw is 0); min_endfr / maxedge / maxlmop /
 * maxlpf = the reference's arguments of those names.  Capacities: max_entries history entries per utterance (the
 * final </s> / silence entries included), link_cap lattice links, pair_cap filler-bypass links.
 */
typedef struct s3a_dagpass_s s3a_dagpass_t;
typedef struct {
    int32_t n_word;
    const int32_t *basewid;
    const uint8_t *is_filler;
    const int32_t *lwid, *fillpen;
    int32_t startwid, finishwid, silwid, start_lwid, finish_lwid, wip;
    double lwf;
    int32_t min_endfr, maxedge, maxlmop, maxlpf;
} s3a_dag_cfg_t;
/* status: 0 ok; 1 no word exit; 2 "Bestpath search failed" (no path / LM operation limit: the reference writes no line);
 * 3 a capacity or -maxedge exceeded; 4 inconsistent table; 5 positive bypass edge (unsupported) */
typedef struct {
    int32_t status, n_words, n_node, n_link, n_bypass, lmop, score, first_pass_score, n_entry, endid;
    const int32_t *wid, *sf, *ef, *ascr, *lscr;     /* [n_words], utterance order; owned by the pass until its next run */
} s3a_dag_result_t;
/* a finished history table handed over from the host (parity tests; the engine binds its device tables itself):
 * n_entry entries INCLUDING vithist_utt_end's final </s> entry (endid), n_frm frames, the first pass's hypothesis */
typedef struct {
    int32_t n_entry, n_frm, endid, n_hyp;
    const int32_t *wid, *sf, *ef, *ascr, *lscr, *score, *hyp_wid, *hyp_sf;
} s3a_dag_table_t;
s3a_dagpass_t *s3a_dagpass_init(s3a_lm3g_t *lm, const s3a_dag_cfg_t *cfg, int32_t n_lanes, int32_t max_entries,
                                int32_t max_frames, int32_t link_cap, int32_t pair_cap);
void s3a_dagpass_free(s3a_dagpass_t *dp);
int32_t s3a_dagpass_run_tables(s3a_dagpass_t *dp, int32_t n_utt, const s3a_dag_table_t *tabs);
int32_t s3a_dagpass_result(const s3a_dagpass_t *dp, int32_t lane, s3a_dag_result_t *out);
/*
 * The lattice itself, for lattice files and for code that wants the reference's dag_t (N-best): the nodes and links
 * vithist_dag_build (vithist.c:1100-1311) makes, in the reference's list orders.  nodes[]: dag->list order = the NODEID
 * order of dag_write (dag.c:731-790); wid / sf / fef / lef / node_ascr / node_lscr of dagnode_t.  links[]: grouped by
 * source node in that order, per source in succlist order (= the order of dag_write's Edges section); ascr / lscr / ef of
 * daglink_t; a node's predlist is its incoming links by source id ascending.  info: dag->nfrm, counts, dag->root and
 * dag->end as node ids, dag->final.ascr, the pass's status.  Call with nodes = links = NULL for the sizes.  Valid until the
 * lane's next pass; available whenever the pass got as far as counting the links (also when the best path then failed).
 */
typedef struct { int32_t wid, sf, fef, lef, ascr, lscr; } s3a_lat_node_t;
typedef struct { int32_t from, to, ascr, lscr, ef; } s3a_lat_link_t;
typedef struct { int32_t status, n_frames, n_nodes, n_links, initial, final, final_ascr; } s3a_lat_info_t;
int32_t s3a_dagpass_lattice(s3a_dagpass_t *dp, int32_t lane, s3a_lat_info_t *info, s3a_lat_node_t *nodes, int32_t node_cap,
                            s3a_lat_link_t *links, int32_t link_cap);
int32_t s3a_uttdec_lattice(s3a_uttdec_t *ud, int32_t lane, s3a_lat_info_t *info, s3a_lat_node_t *nodes, int32_t node_cap,
                           s3a_lat_link_t *links, int32_t link_cap);
/*
 * dag_write (dag.c:731-790: "Frames", "Nodes", "Initial"/"Final", "BestSegAscr 0", "Edges", "End") and dag_write_htk
 * (dag.c:793-897) on such a lattice, into a caller buffer; host code, no device.  `header` = the comment block
 * dag_write_header (dag.c:694-728) prints from the decoder's configuration (the caller has the configuration; may be "").
 * wordstr[wid] = dict_wordstr.  HTK: basewid[wid] = dict_basewid, n_alt[basewid] = pronunciations of the base word
 * (the dict_nextalt chain), logbase / log_shift of the decoder's logmath (a = ascr * ln(base)), lm_lw / lm_wip = lm_t.lw /
 * lm_t.wip for lm_rawscore (lm.c:2172-2178; have_lm = 0: no LM, l = lscr as it is), lmname or NULL, opt_lw / opt_wip = the
 * -lw / -wip arguments as float32, frate = -frate.  Return: the bytes the text needs WITHOUT the terminating NUL; the buffer holds the
 * whole text only when cap > that (a return >= cap: truncated -- call again with a larger buffer); < 0 on bad arguments (a lattice
 * whose initial / final node or link ends lie outside its nodes, negative word ids).
 */
int64_t s3a_lattice_format_s3(const char *header, const s3a_lat_info_t *info, const s3a_lat_node_t *nodes,
                              const s3a_lat_link_t *links, const char *const *wordstr, char *buf, int64_t cap);
typedef struct {
    const char *uttid, *lmname;
    int32_t have_lm, lm_wip, frate, log_shift;
    float lm_lw, opt_lw, opt_wip;
    double log_of_base;
    const int32_t *basewid, *n_alt;
} s3a_htk_opts_t;
int64_t s3a_lattice_format_htk(const char *header, const s3a_htk_opts_t *o, const s3a_lat_info_t *info,
                               const s3a_lat_node_t *nodes, const s3a_lat_link_t *links, const char *const *wordstr,
                               char *buf, int64_t cap);
/*
 * N-best lists on such a lattice (srch_TST_nbest_impl, srch_time_switch_tree.c:1442-1492 -> nbest_search, astar.c:656-716): what the
 * reference does between vithist_dag_build and the list's file -- dag_remove_unreachable, dag_bypass_filler_nodes, dag_compute_hscr,
 * dag_remove_bypass_links (dag.c:303-365, 1037-1112, 521-587), then the A* search over partial paths with the reference's own heap,
 * duplicate table and tie order -- in the library, on the arrays s3a_uttdec_lattice / s3a_dagpass_lattice hand out.  Host code (one
 * best-first chain; the lattice and everything in front of it come from the device).  lm = the engine's trigram; cfg = the second
 * pass's configuration (word tables, -bestpathlw / -lw as lwf, -maxedge, -maxlmop, -maxlpf); the options: uttid, -beam (float64, and
 * logs3 of it), -nbest, -maxppath, and what the file's header prints (-logbase, -lw, -wip) / lm_rawscore needs (lm_t.lw, lm_t.wip).
 * s3a_nbest_result: the file's text as nbest_search writes it (the caller writes the file -- and no file when n_hyp <= 0, as the
 * reference unlinks it), the hypotheses found, counts4 = {pops, expansions, partial paths, bypass links}; returns S3A_OK, or S3A_ENOMEM
 * when -maxedge stopped the bypass (the reference then writes no list: n_hyp = -1).
 */
typedef struct {
    const char *uttid;
    double beam;
    int32_t beam_logs3, nbest, maxppath, lm_wip;
    float logbase, lw, wip, lm_lw;
} s3a_nbest_opts_t;
typedef struct s3a_nbest_s s3a_nbest_t;
s3a_nbest_t *s3a_lattice_nbest(const s3a_lm3g_t *lm, const s3a_dag_cfg_t *cfg, const s3a_nbest_opts_t *o, const s3a_lat_info_t *info,
                               const s3a_lat_node_t *nodes, const s3a_lat_link_t *links, const char *const *wordstr);
int32_t s3a_nbest_result(const s3a_nbest_t *nb, const char **text, int64_t *len, int32_t *n_hyp, int32_t *counts4);
void s3a_nbest_free(s3a_nbest_t *nb);
/* the whole-utterance engine with the second pass: after the frames of every s3a_uttdec_decode* the lanes' tables go
 * through vithist_utt_end and the pass ON THE DEVICE; keep_tables = 0: the history tables are not read back at all
 * (s3a_uttdec_result / s3a_uttdec_hyp then fail; the hypotheses come from s3a_uttdec_bestpath_hyp).
 * s3a_uttdec_bestpath_hyp: the second pass's hypothesis of a lane as header + words (scale = the frame normalisers
 * over [sf, ef), as in s3a_uttdec_hyp); status 0 ok, -1 decode error, -2 no word exit, -4 bestpath failed (the
 * reference writes no line then), -5 the pass gave up on THIS utterance (its link capacity, -maxedge during the filler bypass -- where the
 * reference goes on with a partly bypassed lattice --, or a positive bypass edge): a failed utterance, not a failed batch; -3 max_words too
 * small. */
int32_t s3a_uttdec_enable_bestpath(s3a_uttdec_t *ud, const s3a_dag_cfg_t *cfg, int32_t link_cap, int32_t pair_cap,
                                   int32_t keep_tables);
int32_t s3a_uttdec_bestpath_hyp(s3a_uttdec_t *ud, int32_t lane, const char *uttid, int32_t utt_index,
                                s3a_hyp_header_t *hdr, s3a_hyp_word_t *words, int32_t max_words);
/* With the second pass enabled, s3a_uttdec_decode_queue* (lane refill) runs it at every refill event for the lanes that have
 * ended, before their history tables are reused (from 8 lanes on: per group of n_lanes utterances, each one persistent launch); the
 * result is kept per UTTERANCE of the queue (lattices: s3a_uttdec_queue_keep_lattices below).  Same record and status codes as s3a_uttdec_bestpath_hyp. */
int32_t s3a_uttdec_queue_bestpath_hyp(s3a_uttdec_t *ud, int32_t utt, const char *uttid, int32_t utt_index,
                                      s3a_hyp_header_t *hdr, s3a_hyp_word_t *words, int32_t max_words);
int32_t s3a_uttdec_bestpath_result(s3a_uttdec_t *ud, int32_t lane, s3a_dag_result_t *out);
/* Lattices out of a QUEUE (round 6): switched on before s3a_uttdec_decode_queue*, the lanes' lattices are read back behind every group's
 * (below 8 lanes: every refill event's) second pass -- the host waits for the stream there, the price of it -- and kept per UTTERANCE until
 * the engine's next decode; s3a_uttdec_queue_lattice hands one out as s3a_uttdec_lattice does (nodes = links = NULL: the sizes), for the
 * lattice formatters and s3a_lattice_nbest.  S3A_EUNSUP: that utterance's pass left no lattice. */
int32_t s3a_uttdec_queue_keep_lattices(s3a_uttdec_t *ud, int32_t on);
int32_t s3a_uttdec_queue_lattice(s3a_uttdec_t *ud, int32_t utt, s3a_lat_info_t *info, s3a_lat_node_t *nodes, int32_t node_cap,
                                 s3a_lat_link_t *links, int32_t link_cap);

double s3a_uttdec_last_decode_ms(const s3a_uttdec_t *ud);    /* HIP-event time of the last decode's frames */
/* per-kernel timing for roofline arithmetic: every `every`-th frame of the following decodes is bracketed,
 * launch by launch, by HIP events on the launch stream (0 = off, resets the totals); s3a_uttdec_profile
 * returns per kernel class the summed microseconds / launches (and the class names); returns #classes. */
int32_t s3a_uttdec_set_profile(s3a_uttdec_t *ud, int32_t every);
int32_t s3a_uttdec_profile(const s3a_uttdec_t *ud, double *us, int64_t *launches, const char **names,
                           int32_t max_classes);
int32_t s3a_uttdec_shape(const s3a_uttdec_t *ud, int32_t *n_sen, int32_t *n_ci_sen, int32_t *n_comp_padded,
                         int32_t *veclen, int32_t *n_node, int32_t *n_tree);

/* ===================================================================== */
/* pocketsphinx's first pass on the device (the search half)                                    */
/* replaces, behind ps_searchfuncs_t {start, step, finish, ...} (pocketsphinx/src/libpocketsphinx/                */
/* pocketsphinx_internal.h:68-81), the lexicon-tree Viterbi search of ngram_search_fwdtree.c:                     */
/*   ngram_fwdtree_start :464   ngram_fwdtree_search :1446-1488   ngram_fwdtree_finish :1490                      */
/*   compute_sen_active :513  renormalize_scores :555  evaluate_channels :694 (eval_root_chan :595,               */
/*   eval_nonroot_chan :613, eval_word_chan :634)  prune_channels :1125 (prune_root_chan :714,                    */
/*   prune_nonroot_chan :794, last_phone_transition :877, prune_word_chan :1037)  bptable_maxwpf :1183            */
/*   word_transition :1236  deactivate_channels :1421                                                             */
/* with hmm_vit_eval in pocketsphinx's conventions (hmm.c:532-609 3-state, :249-528 5-state, multiplexed and      */
/* not: int32 path scores, int16 NEGATED senone scores, uint8 negated transition scores), ngram_search_save_bp    */
/* (ngram_search.c:360), ngram_search_exit_score (:601), ngram_search_find_exit (:444) + the backtrace of         */
/* ngram_search_bp_hyp (:486) / ngram_search_bp2itor (:777), and ngram_tg_score >> SENSCR_SHIFT on the            */
/* language model (sphinxbase lm3g_templates.c:73-195 behind ngram_model_set_score, ngram_model_set.c:709).       */
/*                                                                                                                */
/* One LANE = one ps_decoder_t utterance; one workgroup runs a lane's whole frame (no launch per phase, no        */
/* cross-workgroup ordering), lanes run side by side.  The backpointer table that comes back is the               */
/* reference's, entry for entry (order included), so that everything downstream of the first pass -- fwdflat,     */
/* the lattice, bestpath, the segment iterator -- runs unchanged on it (integration/pocketsphinx/ps_search_amd.c).*/
/*                                                                                                                */
/* s3a_psfwd_desc_t is what the search reads of bin_mdef_t / tmat_t / dict_t / dict2pid_t / ngram_search_t /      */
/* ngram_model_t, flattened by the caller (copied at init):                                                       */
/*   channels are numbered roots first: [0, n_root) = ngs->root_chan[], [n_root, n_root + n_nonroot) = the        */
/*   interior channels; ch_child_off/ch_child = a channel's children in sibling (->next, ->alt) order;            */
/*   ch_pen_off/ch_pen_wid = its penult_phn_wid list in homophone_set order; a word's last-phone channels are     */
/*   its right contexts rc_ssid[w_rc_off[w] .. w_rc_off[w+1]) (dict2pid_rssid(last, last2)->ssid[]) with          */
/*   rc_cimap[w_rc_row[w]][ci] = ->cimap[ci]; sp_* = ngs->single_phone_wid[] (the first n_1ph_lm are in the LM)   */
/*   with their permanently allocated root channels; root_lc_ssid[r][lc] / sp_lc_ssid[i][lc] =                    */
/*   dict2pid_ldiph_lc(ciphone, ci2phone, lc).  w_lmwid[w] = the LM's id of dictionary word w                     */
/*   (ngram_model_set_current_wid), negative = NGRAM_INVALID_WID.  LM arrays as s3a_lm3g_init's, values as        */
/*   ngram_model_apply_weights left them (ngram_iter_get).  Beams etc. are ngram_search_calc_beams' values.        */
/* ===================================================================== */
#define S3A_PSW_SINGLE 1    /* dict_is_single_phone */
#define S3A_PSW_FILLER 2    /* dict_filler_word */
#define S3A_PSW_REAL   4    /* dict_real_word */
typedef struct s3a_psfwd_desc_s {
    int32_t n_ci, sil_ci, n_emit, n_sen, n_sseq, n_tmat;
    const uint16_t *sseq;           /* [n_sseq][n_emit]  bin_mdef_t.sseq */
    const uint8_t *tp;              /* [n_tmat][n_emit][n_emit + 1]  tmat_t.tp */
    int32_t n_words, start_wid, finish_wid, silence_wid;
    const int32_t *w_basewid, *w_lmwid;
    const int16_t *w_first_ci, *w_last_ci, *w_last2_ci;     /* last2 = -1 for single-phone words */
    const uint8_t *w_flags;         /* S3A_PSW_* */
    const int32_t *w_rc_off;        /* [n_words + 1] */
    const uint16_t *rc_ssid;
    const int32_t *w_rc_row;        /* [n_words] row of rc_cimap, -1 for single-phone words */
    int32_t n_rc_rows;
    const int16_t *rc_cimap;        /* [n_rc_rows][n_ci] */
    const int16_t *w_rc_tmat;       /* [n_words] bin_mdef_pid2tmatid(last phone) */
    int32_t n_root, n_nonroot;
    const int16_t *root_ci, *root_ci2, *root_tmat;
    const uint16_t *root_ssid0;     /* [n_root] hmm_mpx_ssid(&rhmm->hmm, 0) after create_search_tree */
    const uint16_t *root_lc_ssid;   /* [n_root][n_ci] */
    const int32_t *ch_child_off, *ch_child;     /* [n_root + n_nonroot + 1], channel numbers */
    const int32_t *ch_pen_off, *ch_pen_wid;     /* [n_root + n_nonroot + 1], word ids */
    const uint16_t *nr_ssid;        /* [n_nonroot] */
    const int16_t *nr_tmat, *nr_ci;
    int32_t n_1ph, n_1ph_lm;
    const int32_t *sp_wid;
    const uint16_t *sp_ssid0, *sp_lc_ssid;      /* [n_1ph], [n_1ph][n_ci] */
    const int16_t *sp_tmat, *sp_ci;
    int32_t n_fill;                 /* word_transition's filler loop (:1380-1402): the words it enters, in its order */
    const int32_t *fill_sp;         /* [n_fill] index into sp_* */
    int32_t lm_order, lm_n_ug, lm_n_bg, lm_n_tg, lm_zero;
    const int32_t *ug_prob, *ug_bowt, *ug_firstbg;          /* ug_firstbg [n_ug + 1] */
    const int32_t *bg_wid, *bg_prob, *bg_bowt, *bg_firsttg; /* bg_firsttg [n_bg + 1] */
    const int32_t *tg_wid, *tg_prob;
    int32_t beam, pbeam, wbeam, lpbeam, lponlybeam, fillpen, silpen, nwpen, pip, maxwpf, maxhmmpf;
    /* phone loop look-ahead (-pl_window > 0: phone_loop_search.c; pocketsphinx.c:242-248, :704-712, :823-826): a loop of the CI
     * phones' HMMs runs pl_window frames ahead of the search, and every phone / word transition of the search adds
     * phone_loop_search_score(ci) = the phone's best score - the loop's best score (phone_loop_search.h:103-105).
     * pl_beam / pl_pbeam / pl_pip: phone_loop_search_t.beam / .pbeam / .pip as the decoder holds them; ci_ssid / ci_tmat
     * [n_ci]: bin_mdef_pid2ssid / _pid2tmatid of the CI phones (phone_loop_search_reinit :92-98).  pl_window 0: off. */
    int32_t pl_window, pl_beam, pl_pbeam, pl_pip;
    const uint16_t *ci_ssid;
    const int16_t *ci_tmat;
    /* class-based LMs (sphinxbase ngram_model.c:494-521, ngram_ng_score): a word of a class scores as the class's tag word -- w_lmwid
     * holds the TAG's id for it, as target and as history -- plus its in-class weight w_lmcw[w] = ngram_class_prob (0 for a plain
     * word; 1 = "not in its class": the score is the LM's zero).  NULL: no class words. */
    const int32_t *w_lmcw;          /* [n_words] */
} s3a_psfwd_desc_t;

/* the finished (or running) backpointer table of a lane: bptbl_t by field, ngs->bscore_stack, ngs->bp_table_idx
 * [0 .. n_mark), the search's scalars; pointers are host copies owned by the engine until the lane's next call */
typedef struct {
    int32_t status;                 /* 0 ok; S3A_ENOMEM = a table overflowed (the lane stopped, loudly) */
    int32_t n_frame, n_mark, bpidx, bss_head, best_score, last_phone_best_score, renormalized;
    int32_t st[8];                  /* ngram_search_stats_t: n_phone_eval, n_root_chan_eval, n_nonroot_chan_eval,
                                       n_last_chan_eval, n_word_lastchan_eval, n_lastphn_cand_utt, n_senone_active_utt, 0 */
    const int32_t *frame, *wid, *bp, *score, *s_idx, *real_wid;
    const uint8_t *valid;
    const int32_t *bscore_stack, *bp_table_idx;
} s3a_psfwd_table_t;
typedef struct { int32_t wid, sf, ef, ascr, lscr, bp; } s3a_psfwd_seg_t;

typedef struct s3a_psfwd_s s3a_psfwd_t;
/* max_frames bounds an utterance (int16 frame numbers in the reference: <= 32767); bp_cap / bss_cap = entries of
 * the backpointer table / right-context score stack per lane (0: 64 and 64 * n_ci per frame) -- the reference grows
 * them, exceeding them here stops the lane with status S3A_ENOMEM */
s3a_psfwd_t *s3a_psfwd_init(const s3a_psfwd_desc_t *d, int32_t n_lanes, int32_t max_frames,
                            int32_t bp_cap, int32_t bss_cap);
void    s3a_psfwd_free(s3a_psfwd_t *e);
int32_t s3a_psfwd_n_lanes(const s3a_psfwd_t *e);
/* frame-synchronous use (the three slots; senone scores come from whatever ps_mgau_t the decoder has):
 *   start       ngram_fwdtree_start
 *   sen_active  compute_sen_active for frame_idx: flags[n_sen] (host) = acmod->senone_active_vec as bytes
 *   step        ngram_fwdtree_search for frame_idx given acmod_score's senscr[n_sen] (host; only the flagged
 *               senones are read) and acmod->n_senone_active (a statistic); returns 1, or 0 where the reference
 *               returns 0 (recognition failed), < 0 on error
 *   finish      ngram_fwdtree_finish (n_frames = acmod->output_frame) */
int32_t s3a_psfwd_start(s3a_psfwd_t *e, int32_t lane);
int32_t s3a_psfwd_sen_active(s3a_psfwd_t *e, int32_t lane, int32_t frame_idx, uint8_t *flags);
int32_t s3a_psfwd_step(s3a_psfwd_t *e, int32_t lane, const int16_t *senscr, int32_t frame_idx, int32_t n_senone_active);
int32_t s3a_psfwd_finish(s3a_psfwd_t *e, int32_t lane, int32_t n_frames);
/* frame-synchronous use with -pl_window > 0: the decoder's own phone loop search (an auxiliary ps_search_t that
 * ps_search_forward steps in front of the N-gram search, pocketsphinx.c:704-712) stays where it is; before every step
 * the host hands over what the step adds at its transitions, pl_score[ci] = phone_loop_search_score(pls, ci) for the
 * n_ci CI phones (kept until the next call; NULL: zeros).  S3A_EUNSUP when the engine was built with pl_window 0.
 * In whole-utterance use (s3a_psfwd_decode / _decode_queue) the loop itself runs on the device, inside the lane's
 * launch, pl_window frames ahead of the lane's search (utterances of 1 .. pl_window frames are refused: the reference
 * then steps its search with negative frame numbers, pocketsphinx.c:823-826). */
int32_t s3a_psfwd_set_lookahead(s3a_psfwd_t *e, int32_t lane, const int32_t *pl_score);
/* whole utterances: n_utt <= n_lanes utterances, feat[z] = n_frames[z] rows of the scorer's veclen floats (host);
 * scoring (s3a_ps_ms_cont_mgau_frame_eval's arithmetic, active senones only unless compallsen) and the search of
 * every frame of every lane run on the device between one upload and one download of scalars. */
int32_t s3a_psfwd_decode(s3a_psfwd_t *e, s3a_ps_mgau_t *scorer, int32_t n_utt, const float *const *feat,
                         const int32_t *n_frames, int32_t compallsen, int32_t fresh);
/* A lane is a ps_decoder_t: as in the reference, start does NOT restore a new decoder's channels -- frame numbers,
 * histories and multiplexed senone-sequence ids of pruned channels survive into the next utterance (hmm_clear_scores,
 * hmm.c:176, leaves them), which is observable: a stale frame number equal to f + 1 loses an entry in
 * prune_nonroot_chan (:833-837), stale multiplexed ids activate senones and move the frame's normaliser.  reset (and
 * fresh != 0 in decode) makes the lane a NEW decoder: results then do not depend on what the lane decoded before.
 * get/set_sp_ssid: the multiplexed ids [n_1ph][n_emit] of the single-phone words' channels, which the reference's
 * fwdflat pass shares with the first pass (ngram_search_fwdflat.c:381-387) and leaves changed. */
int32_t s3a_psfwd_reset(s3a_psfwd_t *e, int32_t lane);
int32_t s3a_psfwd_get_sp_ssid(s3a_psfwd_t *e, int32_t lane, uint16_t *ssid);
int32_t s3a_psfwd_set_sp_ssid(s3a_psfwd_t *e, int32_t lane, const uint16_t *ssid);
/* ANY number of utterances through the engine's lanes as a queue (pocketsphinx_batch's loop over a control file, one
 * decoder per lane): every senone score of the batch first (one model-stationary pass over all frames), then ONE launch in
 * which a lane takes the next utterance when its own has ended -- no lane waits for another.  Every utterance starts from
 * a NEW decoder's state (results do not depend on which lane decoded what).  s3a_psfwd_queue_hyp: utterance utt's
 * hypothesis as s3a_psfwd_hyp gives it (at most 256 segments are kept per utterance). */
int32_t s3a_psfwd_decode_queue(s3a_psfwd_t *e, s3a_ps_mgau_t *scorer, int32_t n_utt, const float *const *feat,
                               const int32_t *n_frames, int32_t compallsen);
int32_t s3a_psfwd_queue_hyp(s3a_psfwd_t *e, int32_t utt, int32_t *out_score, s3a_psfwd_seg_t *seg, int32_t max_seg);
int32_t s3a_psfwd_table(s3a_psfwd_t *e, int32_t lane, s3a_psfwd_table_t *out);
/* ngram_search_find_exit(-1) + the backtrace with ngram_search_bp2itor's scores (lwf = 1), made on the device:
 * returns the number of segments (utterance order; 0 = no exit), -3 if max_seg is too small; *out_score = the
 * exit's path score */
int32_t s3a_psfwd_hyp(s3a_psfwd_t *e, int32_t lane, int32_t *out_score, s3a_psfwd_seg_t *seg, int32_t max_seg);
double  s3a_psfwd_last_decode_ms(const s3a_psfwd_t *e);
double  s3a_psfwd_last_score_ms(const s3a_psfwd_t *e);     /* of it, the scoring launches (s3a_psfwd_decode_queue; 0 when they ran beside the search) */

/* ===================================================================== */
/* Kernel variants that are otherwise chosen from the model shape / list sizes.  Every variant gives the same bits;  */
/* the tests force the rarely taken ones, tuning runs sweep the sizes.  The library reads NO environment variable:  */
/* a host that wants environment control reads it itself (as it does for s3a_uttdec_opts_t).  Process-wide; applies */
/* to objects created and calls made after s3a_set_variants.                                                        */
/* ===================================================================== */
typedef struct {
    int32_t scan_chained;           /* the chained multi-workgroup prefix scan for lists of any size (default: from 16 k positions) */
    int32_t calls_by_copy;          /* lextree_enter calls through device memory, not kernel arguments (default: above 96 calls) */
    int32_t batch_no_shared, batch_no_multi;    /* s3a_batch: one scoring launch per decoder instead of the shared-model passes */
    int32_t no_frame_sync_kernel;   /* single-frame scoring through the general kernel */
    int32_t score_nt, score_fpc;    /* whole-utterance scoring: workgroup size (256 / 512 / 1024; 0 = 512), frames per chunk (0 = chosen) */
    int32_t resolve_sweep;          /* whole-utterance engine, many lanes: find the nodes a parent may enter by the sweep over all nodes (round 2/3) instead of from the propagating HMMs' child lists */
    int32_t hist_sort_launch;       /* whole-utterance engine: the histogram sort as a launch of its own in every frame (default: on the count's launch, when a frame needs it) */
    int32_t ps_overlap;             /* s3a_psfwd_decode_queue: score the queue's later utterances BESIDE the search (second stream) instead of before it */
    int32_t ps_score_by_gaussian;   /* pocketsphinx batch scoring: the lane-per-Gaussian kernel (k_ps_cont_slots) instead of lane-per-frame */
    int32_t kf_queue_in_order;      /* ku_frames' queue: the lanes take the utterances in queue order (default: each part's longest first) */
    int32_t kf_no_relay;            /* ku_frames: a call is ONE launch to its end (default: the relay -- when no utterance is left to take and few lanes are
                                     * still at work, they hand over at a frame boundary to a launch that runs them as clusters of 2, then 4 workgroups) */
    int32_t kf_relay_at;            /* > 0: the relay's first hand-over when that many lanes are left, the second at half of it (default: as many as
                                     * fill the chip as clusters of 2 / of 4; tests use it to run the chain with a handful of lanes) */
} s3a_variants_t;
void    s3a_variants_default(s3a_variants_t *v);
void    s3a_get_variants(s3a_variants_t *v);            /* the variants in force (read-modify-write with s3a_set_variants) */
int32_t s3a_set_variants(const s3a_variants_t *v);

/* ===================================================================== */
/* measurement hooks used by bench.py (HIP events on the launch stream)   */
/* ===================================================================== */
/*
 * Run s3a_mgau_score_frames_dev `iters` times back to back on an internal
 * stream and return the average kernel-region time per iteration in
 * microseconds, measured with hipEvent pairs on that stream (not torch's).
 * frames_per_launch: 0 = one launch for the whole block (utterance mode),
 * k > 0 = frame-synchronous mode, ceil(n_frames/k) launches of k frames.
 */
int32_t s3a_bench_score_frames(s3a_mgau_model_t *g, const float *feat_dev, int32_t n_frames,
                               int32_t *senscr_dev, int32_t *best_dev, int32_t frames_per_launch,
                               int32_t iters, double *avg_us, double *avg_kernel_us,
                               int32_t *n_launches);

/* HIP-event stopwatch on the model's launch stream: begin records an event,
 * end records a second one, waits for it and returns the elapsed microseconds
 * of everything enqueued on that stream in between. */
int32_t s3a_stream_timer_begin(s3a_mgau_model_t *g);
int32_t s3a_stream_timer_end(s3a_mgau_model_t *g, double *elapsed_us);

/* raw device-memory helpers so a C host (or ctypes) can stage buffers
 * without linking the HIP runtime itself */
void   *s3a_dev_malloc(size_t nbytes);
int32_t s3a_dev_free(void *p);
int32_t s3a_dev_upload(void *dst_dev, const void *src_host, size_t nbytes);
int32_t s3a_dev_download(void *dst_host, const void *src_dev, size_t nbytes);
int32_t s3a_dev_sync(void);

#ifdef __cplusplus
}
#endif
#endif /* CMUSPHINX_AMD_H */
