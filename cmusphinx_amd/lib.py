"""ctypes binding of libcmusphinx_amd.so (the C ABI in include/cmusphinx_amd.h).

Mirrors the reference's interface names for this path (logmath_*, mgau_init,
mgau_eval, approx_cont_mgau_*_eval, tmat_init, hmm_vit_eval ...) so the parity
tests read like the reference's own unit tests.  Fails loudly when the library
is not built: there is deliberately no fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcmusphinx_amd.so")

S3A_OK = 0
LOGPROB_ZERO = -939524096
GMM_EXACT, GMM_FAST = 0, 1
MIX_INT_FLOAT_COMP = 2

# every symbol include/cmusphinx_amd.h declares (checked by tests/test_abi.py)
_SIGS = {
    "s3a_last_error": (C.c_char_p, []),
    "s3a_version": (C.c_char_p, []),
    "s3a_device_count": (C.c_int32, []),
    "s3a_set_device": (C.c_int32, [C.c_int32]),
    "s3a_logmath_init": (C.c_void_p, [C.c_double, C.c_int32, C.c_int32]),
    "s3a_logs3_init": (C.c_void_p, [C.c_double, C.c_int32, C.c_int32]),
    "s3a_logmath_free": (None, [C.c_void_p]),
    "s3a_logmath_add": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32]),
    "s3a_logmath_log": (C.c_int32, [C.c_void_p, C.c_double]),
    "s3a_logmath_exp": (C.c_double, [C.c_void_p, C.c_int32]),
    "s3a_logmath_ln_to_log": (C.c_int32, [C.c_void_p, C.c_double]),
    "s3a_logmath_log_to_ln": (C.c_double, [C.c_void_p, C.c_int32]),
    "s3a_logmath_log10_to_log": (C.c_int32, [C.c_void_p, C.c_double]),
    "s3a_logmath_get_base": (C.c_double, [C.c_void_p]),
    "s3a_logmath_get_zero": (C.c_int32, [C.c_void_p]),
    "s3a_logmath_get_table_shape": (C.c_int32, [C.c_void_p] + [C.POINTER(C.c_uint32)] * 3),
    "s3a_logmath_copy_table": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "s3a_logs3": (C.c_int32, [C.c_void_p, C.c_double]),
    "s3a_mgau_init": (C.c_void_p, [C.c_char_p, C.c_char_p, C.c_double, C.c_char_p, C.c_double,
                                   C.c_int32, C.c_char_p, C.c_int32, C.c_void_p]),
    "s3a_mgau_init_arrays": (C.c_void_p, [C.c_void_p] * 3 + [C.c_int32] * 3 +
                             [C.c_double, C.c_double, C.c_int32, C.c_void_p]),
    "s3a_mgau_load_host": (C.c_void_p, [C.c_char_p, C.c_char_p, C.c_double, C.c_char_p, C.c_double,
                                        C.c_int32, C.c_void_p]),
    "s3a_mgau_free": (None, [C.c_void_p]),
    "s3a_mgau_set_params": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "s3a_fe_stream": (C.c_void_p, [C.c_void_p]),
    "s3a_audio_to_feat_dev": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "s3a_feat_lda_dev": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "s3a_audio_to_feat_dev_prior": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "s3a_mgau_n_mgau": (C.c_int32, [C.c_void_p]),
    "s3a_mgau_max_comp": (C.c_int32, [C.c_void_p]),
    "s3a_mgau_veclen": (C.c_int32, [C.c_void_p]),
    "s3a_mgau_n_comp": (C.c_int32, [C.c_void_p, C.c_int32]),
    "s3a_mgau_distfloor": (C.c_double, [C.c_void_p]),
    "s3a_mgau_get_params": (C.c_int32, [C.c_void_p] * 6),
    "s3a_mgau_reset_state": (C.c_int32, [C.c_void_p]),
    "s3a_mgau_get_state": (C.c_int32, [C.c_void_p] * 4),
    "s3a_mgau_set_precision": (C.c_int32, [C.c_void_p, C.c_int32]),
    "s3a_mgau_eval": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "s3a_mgau_score_frames": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "s3a_mgau_score_frames_dev": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                              C.c_void_p, C.c_void_p]),
    "s3a_scorer_init": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_double, C.c_float, C.c_int32]),
    "s3a_scorer_init_private": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_double, C.c_float, C.c_int32]),
    "s3a_scorer_free": (None, [C.c_void_p]),
    "s3a_scorer_utt_begin": (C.c_int32, [C.c_void_p]),
    "s3a_approx_cont_mgau_ci_eval": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.POINTER(C.c_int32), C.c_int32]),
    "s3a_approx_cont_mgau_frame_eval": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_int32, C.c_void_p,
                                                    C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                                    C.POINTER(C.c_int32)]),
    "s3a_comsen_init": (C.c_void_p, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "s3a_comsen_free": (None, [C.c_void_p]),
    "s3a_dict2pid_comsenscr": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "s3a_tmat_init": (C.c_void_p, [C.c_char_p, C.c_double, C.c_int32, C.c_void_p]),
    "s3a_tmat_init_arrays": (C.c_void_p, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p]),
    "s3a_tmat_init_logs3": (C.c_void_p, [C.c_void_p, C.c_int32, C.c_int32]),
    "s3a_tmat_free": (None, [C.c_void_p]),
    "s3a_tmat_n_tmat": (C.c_int32, [C.c_void_p]),
    "s3a_tmat_n_state": (C.c_int32, [C.c_void_p]),
    "s3a_tmat_get_tp": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "s3a_hmm_batch_init": (C.c_void_p, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "s3a_hmm_batch_free": (None, [C.c_void_p]),
    "s3a_hmm_batch_setup": (C.c_int32, [C.c_void_p] * 4),
    "s3a_hmm_batch_clear": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32]),
    "s3a_hmm_batch_enter": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "s3a_hmm_batch_vit_eval": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "s3a_hmm_batch_get": (C.c_int32, [C.c_void_p] * 8),
    "s3a_lexsearch_init": (C.c_void_p, [C.c_int32] + [C.c_void_p] * 14 + [C.c_void_p, C.c_void_p, C.c_int32,
                                                                          C.c_void_p, C.c_int32, C.c_int32,
                                                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "s3a_lexsearch_clone": (C.c_void_p, [C.c_void_p, C.c_void_p]),
    "s3a_lexsearch_free": (None, [C.c_void_p]),
    "s3a_lexsearch_reset": (C.c_int32, [C.c_void_p]),
    "s3a_lexsearch_n_node": (C.c_int32, [C.c_void_p, C.c_int32]),
    "s3a_lexsearch_enter": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int32, C.c_int32]),
    "s3a_lexsearch_active_swap": (C.c_int32, [C.c_void_p]),
    "s3a_lexsearch_hmm_eval": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    "s3a_lexsearch_propagate_non_leaves": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "s3a_lexsearch_propagate_leaves": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_int32]),
    "s3a_lexsearch_frame_search": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 6 +
                                   [C.c_void_p] * 6 + [C.c_int32]),
    "s3a_approx_cont_mgau_frame_eval_async": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "s3a_scorer_misc_dev": (C.c_void_p, [C.c_void_p]),
    "s3a_senlog_open_write": (C.c_void_p, [C.c_char_p, C.c_char_p, C.c_int32, C.c_double]),
    "s3a_senlog_write_frame": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "s3a_senlog_open_read": (C.c_void_p, [C.c_char_p, C.c_void_p, C.c_void_p]),
    "s3a_senlog_read_frame": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "s3a_senlog_close": (None, [C.c_void_p]),
    "s3a_fe_default_params": (None, [C.c_void_p]),
    "s3a_fe_init": (C.c_void_p, [C.c_void_p]),
    "s3a_fe_free": (None, [C.c_void_p]),
    "s3a_fe_output_size": (C.c_int32, [C.c_void_p]),
    "s3a_fe_frame_shift": (C.c_int32, [C.c_void_p]),
    "s3a_fe_frame_size": (C.c_int32, [C.c_void_p]),
    "s3a_fe_n_frames": (C.c_int32, [C.c_void_p, C.c_int64]),
    "s3a_fe_process_utt": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    "s3a_fe_process_utt_dev": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p,
                                           C.c_void_p]),
    "s3a_feat_1s_c_d_dd": (C.c_int32, [C.c_void_p] + [C.c_int32] * 5 + [C.c_void_p]),
    "s3a_feat_1s_c_d_dd_dev": (C.c_int32, [C.c_void_p] + [C.c_int32] * 5 + [C.c_void_p, C.c_int32, C.c_void_p]),
    "s3a_ps_ms_mgau_init": (C.c_void_p, [C.c_char_p, C.c_char_p, C.c_double, C.c_char_p, C.c_double, C.c_char_p, C.c_int32,
                                         C.c_int32, C.c_double]),
    "s3a_ps_ms_mgau_init_arrays": (C.c_void_p, [C.c_void_p] * 3 + [C.c_int32] * 3 + [C.c_void_p, C.c_int32, C.c_void_p,
                                                                                 C.c_double, C.c_double, C.c_int32,
                                                                                 C.c_int32, C.c_double]),
    "s3a_ps_ms_mgau_free": (None, [C.c_void_p]),
    "s3a_ps_ms_mgau_n_sen": (C.c_int32, [C.c_void_p]),
    "s3a_ps_ms_mgau_veclen": (C.c_int32, [C.c_void_p]),
    "s3a_ps_ms_cont_mgau_frame_eval": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                                   C.c_int32]),
    "s3a_ms_mgau_init": (C.c_void_p, [C.c_char_p, C.c_char_p, C.c_double, C.c_char_p, C.c_double, C.c_int32,
                                      C.c_char_p, C.c_char_p, C.c_int32, C.c_void_p]),
    "s3a_ms_mgau_init_arrays": (C.c_void_p, [C.c_void_p] * 3 + [C.c_int32] * 3 + [C.c_void_p, C.c_int32, C.c_void_p,
                                                                              C.c_double, C.c_double, C.c_int32,
                                                                              C.c_void_p]),
    "s3a_ms_mgau_free": (None, [C.c_void_p]),
    "s3a_ms_mgau_n_sen": (C.c_int32, [C.c_void_p]),
    "s3a_ms_mgau_topn": (C.c_int32, [C.c_void_p]),
    "s3a_ms_mgau_veclen": (C.c_int32, [C.c_void_p]),
    "s3a_ms_cont_mgau_frame_eval": (C.c_int32, [C.c_void_p] * 4 + [C.c_int32, C.POINTER(C.c_int32)]),
    "s3a_ms_mgau_get_dist": (C.c_int32, [C.c_void_p] * 3),
    "s3a_batch_create": (C.c_void_p, [C.c_int32]),
    "s3a_batch_free": (None, [C.c_void_p]),
    "s3a_batch_attach": (C.c_int32, [C.c_void_p] * 4),
    "s3a_batch_utt_begin": (C.c_int32, [C.c_void_p, C.c_int32]),
    "s3a_batch_utt_end": (C.c_int32, [C.c_void_p, C.c_int32]),
    "s3a_batch_transition": (C.c_int32, [C.c_void_p] + [C.c_int32] * 5 + [C.c_void_p] * 3 + [C.c_int32, C.c_int32] +
                             [C.c_void_p] * 3),
    "s3a_batch_step": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p] + [C.c_int32] * 7 + [C.c_void_p] * 5 + [C.c_int32]),
    "s3a_batch_submit": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p] + [C.c_int32] * 7 + [C.c_void_p] * 5 + [C.c_int32]),
    "s3a_batch_run": (C.c_int32, [C.c_void_p]),
    "s3a_batch_stats": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "s3a_decoder_utt_begin": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "s3a_lexsearch_hmm_histbin": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32]),
    "s3a_decoder_score": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32]),
    "s3a_decoder_search": (C.c_int32, [C.c_void_p] * 3 + [C.c_int32] * 6 + [C.c_void_p] * 5 + [C.c_int32]),
    "s3a_decoder_transition": (C.c_int32, [C.c_void_p] * 3 + [C.c_int32] * 4 + [C.c_void_p] * 3 +
                               [C.c_int32, C.c_int32] + [C.c_void_p] * 3),
    "s3a_lexsearch_sen_active": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32]),
    "s3a_lexsearch_utt_end": (C.c_int32, [C.c_void_p]),
    "s3a_lexsearch_get_active": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                                             C.c_void_p, C.c_int32]),
    "s3a_lexsearch_get_hmm": (C.c_int32, [C.c_void_p, C.c_int32] + [C.c_void_p] * 6),
    "s3a_mgau_stream": (C.c_void_p, [C.c_void_p]),
    "s3a_scorer_sen_active_dev": (C.c_void_p, [C.c_void_p]),
    "s3a_scorer_senscr_dev": (C.c_void_p, [C.c_void_p]),
    "s3a_comsen_dev": (C.c_void_p, [C.c_void_p]),
    "s3a_approx_cont_mgau_frame_eval_dev": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                                        C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                                        C.POINTER(C.c_int32)]),
    "s3a_lm3g_init": (C.c_void_p, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "s3a_lm3g_init_host": (C.c_void_p, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "s3a_lattice_nbest": (C.c_void_p, [C.c_void_p] * 7),
    "s3a_nbest_result": (C.c_int32, [C.c_void_p] * 5),
    "s3a_nbest_free": (None, [C.c_void_p]),
    "s3a_lm3g_free": (None, [C.c_void_p]),
    "s3a_lm3g_tg_score": (C.c_int32, [C.c_void_p] + [C.c_int32] * 4),
    "s3a_uttdec_init": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_double, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_int32]),
    "s3a_uttdec_init_opts": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_double, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                          C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "s3a_uttdec_opts_default": (None, [C.c_void_p]),
    "s3a_uttdec_opts_from_env": (None, [C.c_void_p]),
    "s3a_uttdec_free": (None, [C.c_void_p]),
    "s3a_uttdec_decode": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]),
    "s3a_uttdec_decode_dev": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]),
    "s3a_uttdec_decode_queue": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]),
    "s3a_uttdec_decode_queue_dev": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]),
    "s3a_queue_schedule": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "s3a_uttdec_queue_status": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "s3a_uttdec_queue_hyp": (C.c_int32, [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]),
    "s3a_uttdec_result": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p]),
    "s3a_uttdec_wl_ticks": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p]),
    "s3a_uttdec_frame_ticks": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "s3a_uttdec_frame_dbg": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p]),
    "s3a_uttdec_last_parts": (C.c_int32, [C.c_void_p] + [C.c_void_p] * 5),
    "s3a_uttdec_last_relay": (C.c_int32, [C.c_void_p]),
    "s3a_uttdec_queue_keep_lattices": (C.c_int32, [C.c_void_p, C.c_int32]),
    "s3a_uttdec_queue_lattice": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]),
    "s3a_uttdec_n_lanes": (C.c_int32, [C.c_void_p]),
    "s3a_uttdec_window": (C.c_int32, [C.c_void_p]),
    "s3a_uttdec_enable_pheur": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]),
    "s3a_uttdec_selfcheck": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p]),
    "s3a_gather_init": (C.c_void_p, [C.c_int32, C.c_int32, C.c_char_p]),
    "s3a_get_variants": (None, [C.c_void_p]),
    "s3a_gather_init_run": (C.c_void_p, [C.c_int32, C.c_int32, C.c_char_p, C.c_uint64]),
    "s3a_gather_free": (None, [C.c_void_p]),
    "s3a_gather_hyps": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]),
    "s3a_gather_result": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "s3a_dagpass_init": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "s3a_dagpass_free": (None, [C.c_void_p]),
    "s3a_dagpass_run_tables": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p]),
    "s3a_dagpass_result": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p]),
    "s3a_dagpass_lattice": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]),
    "s3a_uttdec_lattice": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]),
    "s3a_lattice_format_s3": (C.c_int64, [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "s3a_lattice_format_htk": (C.c_int64, [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "s3a_uttdec_enable_bestpath": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "s3a_uttdec_bestpath_hyp": (C.c_int32, [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]),
    "s3a_uttdec_bestpath_result": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p]),
    "s3a_uttdec_queue_bestpath_hyp": (C.c_int32, [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]),
    "s3a_uttdec_set_profile": (C.c_int32, [C.c_void_p, C.c_int32]),
    "s3a_uttdec_profile": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "s3a_variants_default": (None, [C.c_void_p]),
    "s3a_set_variants": (C.c_int32, [C.c_void_p]),
    "s3a_psfwd_init": (C.c_void_p, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "s3a_psfwd_free": (None, [C.c_void_p]),
    "s3a_psfwd_n_lanes": (C.c_int32, [C.c_void_p]),
    "s3a_psfwd_start": (C.c_int32, [C.c_void_p, C.c_int32]),
    "s3a_psfwd_reset": (C.c_int32, [C.c_void_p, C.c_int32]),
    "s3a_psfwd_sen_active": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "s3a_psfwd_step": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32]),
    "s3a_psfwd_finish": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32]),
    "s3a_psfwd_set_lookahead": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p]),
    "s3a_psfwd_decode": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "s3a_psfwd_decode_queue": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]),
    "s3a_psfwd_queue_hyp": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_void_p, C.c_int32]),
    "s3a_psfwd_table": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p]),
    "s3a_psfwd_hyp": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_void_p, C.c_int32]),
    "s3a_psfwd_get_sp_ssid": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p]),
    "s3a_psfwd_set_sp_ssid": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p]),
    "s3a_psfwd_last_decode_ms": (C.c_double, [C.c_void_p]),
    "s3a_psfwd_last_score_ms": (C.c_double, [C.c_void_p]),
    "s3a_uttdec_shape": (C.c_int32, [C.c_void_p] + [C.POINTER(C.c_int32)] * 6),
    "s3a_uttdec_hyp": (C.c_int32, [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.c_void_p]),
    "s3a_uttdec_hyp_var": (C.c_int32, [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]),
    "s3a_hyp_format_var": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float,
                                       C.c_int32, C.c_int32, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
    "s3a_hyp_format": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float,
                                   C.c_int32, C.c_int32, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
    "s3a_uttdec_last_decode_ms": (C.c_double, [C.c_void_p]),
    "s3a_wltest_init": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "s3a_wltest_free": (None, [C.c_void_p]),
    "s3a_wltest_begin": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32]),
    "s3a_wltest_frame": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                     C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "s3a_wltest_fetch": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p, C.c_int32,
                                     C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "s3a_bench_score_frames": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_int32, C.c_int32, C.POINTER(C.c_double),
                                           C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "s3a_stream_timer_begin": (C.c_int32, [C.c_void_p]),
    "s3a_stream_timer_end": (C.c_int32, [C.c_void_p, C.POINTER(C.c_double)]),
    "s3a_dev_malloc": (C.c_void_p, [C.c_size_t]),
    "s3a_dev_free": (C.c_int32, [C.c_void_p]),
    "s3a_dev_upload": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "s3a_dev_download": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "s3a_dev_sync": (C.c_int32, []),
}

_L = None
MISSING: list = []


class S3AError(RuntimeError):
    pass


def load():
    """dlopen libcmusphinx_amd.so and type every entry point.  No fallback."""
    global _L
    if _L is not None:
        return _L
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C cmusphinx_amd/csrc`); cmusphinx_amd has no CPU fallback")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            MISSING.append(name)    # tests/test_abi.py fails on any entry here
            continue
        fn.restype = res
        fn.argtypes = args
    _L = L
    return L


def _err(L):
    m = L.s3a_last_error()
    return m.decode() if m else ""


def check(rc, L=None):
    if rc != S3A_OK:
        L = L or load()
        raise S3AError(f"libcmusphinx_amd error {rc}: {_err(L)}")


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def queue_schedule(n_lanes, boundary, n_frames):
    """the lane-refill schedule of s3a_uttdec_decode_queue (host arithmetic): -> (lane[u], f0[u], engine frames in all)"""
    nf = np.ascontiguousarray(n_frames, np.int32)
    lane, f0 = np.zeros(len(nf), np.int32), np.zeros(len(nf), np.int32)
    total = load().s3a_queue_schedule(int(n_lanes), int(boundary), len(nf), _p(nf), _p(lane), _p(f0))
    if total < 0:
        raise S3AError(f"s3a_queue_schedule: error {total}")
    return lane, f0, int(total)


def device_count() -> int:
    return int(load().s3a_device_count())


class LogMath:
    """logmath_t (sphinxbase logmath.h) / logs3_init."""

    def __init__(self, base=1.0003, shift=0, use_table=1):
        self.L = load()
        self.h = self.L.s3a_logmath_init(base, shift, use_table)
        if not self.h:
            raise S3AError(_err(self.L))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.s3a_logmath_free(self.h)
            self.h = None

    def add(self, x, y): return self.L.s3a_logmath_add(self.h, int(x), int(y))
    def log(self, p): return self.L.s3a_logmath_log(self.h, float(p))
    def exp(self, v): return self.L.s3a_logmath_exp(self.h, int(v))
    def logs3(self, p): return self.L.s3a_logs3(self.h, float(p))
    def ln_to_log(self, v): return self.L.s3a_logmath_ln_to_log(self.h, float(v))
    def log_to_ln(self, v): return self.L.s3a_logmath_log_to_ln(self.h, int(v))
    def log10_to_log(self, v): return self.L.s3a_logmath_log10_to_log(self.h, float(v))

    @property
    def zero(self): return self.L.s3a_logmath_get_zero(self.h)

    @property
    def base(self): return self.L.s3a_logmath_get_base(self.h)

    def table_shape(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(self.L.s3a_logmath_get_table_shape(self.h, C.byref(a), C.byref(b), C.byref(c)), self.L)
        return a.value, b.value, c.value

    @property
    def table(self):
        n = self.table_shape()[0]
        out = np.zeros(n, np.uint32)
        check(self.L.s3a_logmath_copy_table(self.h, _p(out), n), self.L)
        return out


class DevBuf:
    """A raw HIP allocation owned by the library's runtime (no torch needed)."""

    def __init__(self, nbytes):
        self.L = load()
        self.nbytes = int(nbytes)
        self.ptr = self.L.s3a_dev_malloc(self.nbytes)
        if not self.ptr:
            raise S3AError(_err(self.L))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(self.L.s3a_dev_upload(self.ptr, _p(arr), arr.nbytes), self.L)
        return self

    def download(self, dtype, shape):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        check(self.L.s3a_dev_download(_p(out), self.ptr, out.nbytes), self.L)
        return out

    def __del__(self):
        if getattr(self, "ptr", None):
            self.L.s3a_dev_free(self.ptr)
            self.ptr = None


class MgauModel:
    """mgau_model_t (cont_mgau.h) living on the GPU."""

    def __init__(self, h, lm):
        self.L = load()
        self.h = h
        self.lm = lm            # keep the logmath alive: the model borrows it
        self.S = self.L.s3a_mgau_n_mgau(h)
        self.C = self.L.s3a_mgau_max_comp(h)
        self.D = self.L.s3a_mgau_veclen(h)

    @classmethod
    def init(cls, meanfile, varfile, mixwfile, lm: LogMath, varfloor=1e-4, mixwfloor=1e-7,
             precomp=1, senmgau=".cont.", comp_type=MIX_INT_FLOAT_COMP):
        L = load()
        h = L.s3a_mgau_init(meanfile.encode(), varfile.encode(), varfloor, mixwfile.encode(),
                            mixwfloor, precomp, senmgau.encode(), comp_type, lm.h)
        if not h:
            raise S3AError(_err(L))
        return cls(h, lm)

    @classmethod
    def load_host(cls, meanfile, varfile, mixwfile, lm: LogMath, varfloor=1e-4, mixwfloor=1e-7,
                  precomp=1):
        """Loader only (no upload): parameters can be inspected, scoring fails with ENODEV."""
        L = load()
        h = L.s3a_mgau_load_host(meanfile.encode(), varfile.encode(), varfloor, mixwfile.encode(),
                                 mixwfloor, precomp, lm.h)
        if not h:
            raise S3AError(_err(L))
        return cls(h, lm)

    @classmethod
    def init_arrays(cls, mean, var, mixw, lm: LogMath, varfloor=1e-4, mixwfloor=1e-7, precomp=1):
        L = load()
        mean = np.ascontiguousarray(mean, np.float32)
        var = np.ascontiguousarray(var, np.float32)
        S, Cn, D = mean.shape
        mixw = np.ascontiguousarray(mixw, np.float32).reshape(S, Cn)
        h = L.s3a_mgau_init_arrays(_p(mean), _p(var), _p(mixw), S, Cn, D, varfloor, mixwfloor,
                                   precomp, lm.h)
        if not h:
            raise S3AError(_err(L))
        return cls(h, lm)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.s3a_mgau_free(self.h)
            self.h = None

    def params(self):
        S, Cn, D = self.S, self.C, self.D
        mean = np.zeros((S, Cn, D), np.float32)
        prec = np.zeros((S, Cn, D), np.float32)
        lrd = np.zeros((S, Cn), np.float32)
        mixw = np.zeros((S, Cn), np.int32)
        n_comp = np.zeros(S, np.int32)
        check(self.L.s3a_mgau_get_params(self.h, _p(mean), _p(prec), _p(lrd), _p(mixw), _p(n_comp)))
        return dict(mean=mean, prec=prec, lrd=lrd, mixw=mixw, n_comp=n_comp,
                    distfloor=self.L.s3a_mgau_distfloor(self.h))

    def stream(self):
        """The HIP stream this model's kernels run on (hand it to LexSearch for the fused frame)."""
        return self.L.s3a_mgau_stream(self.h)

    def set_precision(self, mode):
        check(self.L.s3a_mgau_set_precision(self.h, mode))

    def reset_state(self):
        check(self.L.s3a_mgau_reset_state(self.h))

    def state(self):
        b = np.zeros(self.S, np.int32); s = np.zeros(self.S, np.int32); u = np.zeros(self.S, np.int32)
        check(self.L.s3a_mgau_get_state(self.h, _p(b), _p(s), _p(u)))
        return b, s, u

    def eval(self, m, x, fr=0, update=1, active=None):
        """mgau_eval(g, m, active, x, fr, update)."""
        x = np.ascontiguousarray(x, np.float32)
        act = None
        if active is not None:
            act = np.ascontiguousarray(list(active) + [-1], np.int32)
        return self.L.s3a_mgau_eval(self.h, int(m), _p(act), _p(x), int(fr), int(update))

    def score_frames(self, feat, want_best=True):
        feat = np.ascontiguousarray(feat, np.float32)
        T = feat.shape[0]
        out = np.empty((T, self.S), np.int32)
        best = np.empty(T, np.int32) if want_best else None
        check(self.L.s3a_mgau_score_frames(self.h, _p(feat), T, _p(out), _p(best)))
        return (out, best) if want_best else out

    def score_frames_dev(self, feat_dev: DevBuf, n_frames, scr_dev: DevBuf, best_dev=None):
        """Asynchronous, device-resident: enqueue on the model's stream and return."""
        check(self.L.s3a_mgau_score_frames_dev(self.h, feat_dev.ptr, n_frames, scr_dev.ptr,
                                               best_dev.ptr if best_dev else None, None))

    def timer_begin(self):
        check(self.L.s3a_stream_timer_begin(self.h))

    def timer_end(self):
        us = C.c_double()
        check(self.L.s3a_stream_timer_end(self.h, C.byref(us)))
        return us.value

    def bench(self, feat_dev: DevBuf, n_frames, scr_dev: DevBuf, best_dev, frames_per_launch, iters):
        us, kus, nl = C.c_double(), C.c_double(), C.c_int32()
        check(self.L.s3a_bench_score_frames(self.h, feat_dev.ptr, n_frames, scr_dev.ptr,
                                            best_dev.ptr if best_dev else None,
                                            frames_per_launch, iters, C.byref(us), C.byref(kus),
                                            C.byref(nl)))
        return us.value, kus.value, nl.value


class Scorer:
    """fast_gmm_t + approx_cont_mgau_{ci,frame}_eval + the ascr_t buffers they fill."""

    def __init__(self, g: MgauModel, cd2cisen, n_ci_sen, ds_ratio=1, cond_ds=0, ci_pbeam=1e-80,
                 tighten_factor=0.5, max_cd=100000, private_state=False):
        self.L = load()
        self.g = g
        self.cd2cisen = np.ascontiguousarray(cd2cisen, np.int16)
        self.n_sen = len(self.cd2cisen)
        self.n_ci_sen = int(n_ci_sen)
        init = self.L.s3a_scorer_init_private if private_state else self.L.s3a_scorer_init
        self.h = init(g.h, _p(self.cd2cisen), self.n_sen, self.n_ci_sen, ds_ratio, cond_ds, ci_pbeam,
                      tighten_factor, max_cd)
        if not self.h:
            raise S3AError(_err(self.L))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.s3a_scorer_free(self.h)
            self.h = None

    def utt_begin(self):
        check(self.L.s3a_scorer_utt_begin(self.h))

    def ci_eval(self, feat, fr):
        feat = np.ascontiguousarray(feat, np.float32)
        ci = np.zeros(max(self.n_ci_sen, 1), np.int32)
        best = C.c_int32(0)
        check(self.L.s3a_approx_cont_mgau_ci_eval(self.h, _p(feat), _p(ci), C.byref(best), int(fr)))
        return ci[:self.n_ci_sen], best.value

    def frame_eval(self, sen_active, senscr, feat, frame, cache_ci_senscr):
        """In-place on sen_active / senscr like the reference; returns (best, ns, ng, rec_active)."""
        feat = np.ascontiguousarray(feat, np.float32)
        ci = np.ascontiguousarray(cache_ci_senscr, np.int32)
        rec = np.zeros(self.n_sen, np.uint8)
        best, ns, ng = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        check(self.L.s3a_approx_cont_mgau_frame_eval(self.h, _p(sen_active), _p(rec), _p(senscr),
                                                     _p(feat), int(frame), _p(ci), C.byref(best),
                                                     C.byref(ns), C.byref(ng)))
        return best.value, ns.value, ng.value, rec

    def frame_eval_seq(self, feats, active=None):
        """ci_eval + frame_eval over a sequence, as srch_utt_decode_blk with -pl_window 1."""
        feats = np.ascontiguousarray(feats, np.float32)
        T, S = feats.shape[0], self.n_sen
        out = dict(senscr=np.zeros((T, S), np.int32), best=np.zeros(T, np.int32),
                   ci_best=np.zeros(T, np.int32), sen_active_out=np.zeros((T, S), np.uint8),
                   bstidx=np.zeros((T, S), np.int32), updatetime=np.zeros((T, S), np.int32),
                   counts=np.zeros((T, 2), np.int32))
        cur = np.zeros(S, np.int32)
        sa = np.zeros(S, np.uint8)
        self.utt_begin()
        for t in range(T):
            ci, cb = self.ci_eval(feats[t], t)
            out["ci_best"][t] = cb
            sa[:] = 1 if active is None else active[t]
            b, ns, ng, _ = self.frame_eval(sa, cur, feats[t], t, ci)
            out["best"][t] = b
            out["counts"][t] = (ns, ng)
            out["senscr"][t] = cur
            out["sen_active_out"][t] = sa
            bi, _, ut = self.g.state()
            out["bstidx"][t] = bi
            out["updatetime"][t] = ut
        return out


def feat_1s_c_d_dd(cep, cmn="current", varnorm=False, agc="none"):
    """feat_compute_utt for "1s_c_d_dd" on the device: cep [n][cepsize] -> feat [n][3 * cepsize]."""
    L = load()
    cep = np.ascontiguousarray(cep, np.float32)
    n, cs = cep.shape
    out = np.zeros((n, 3 * cs), np.float32)
    check(L.s3a_feat_1s_c_d_dd(_p(cep), n, cs, int(cmn == "current"), int(bool(varnorm)), int(agc == "max"), _p(out)))
    return out


class SenLog:
    """pocketsphinx's senone score dump (-senlogdir): SenLog.create(path, mdef, n_sen, logbase) / SenLog.open(path)."""

    def __init__(self, h, n_sen, logbase):
        self.L, self.h, self.n_sen, self.logbase = load(), h, n_sen, logbase

    @classmethod
    def create(cls, path, mdef_file, n_sen, logbase):
        h = load().s3a_senlog_open_write(path.encode(), mdef_file.encode(), n_sen, logbase)
        if not h:
            raise S3AError(_err(load()))
        return cls(h, n_sen, logbase)

    @classmethod
    def open(cls, path):
        n, b = C.c_int32(), C.c_double()
        h = load().s3a_senlog_open_read(path.encode(), C.byref(n), C.byref(b))
        if not h:
            raise S3AError(_err(load()))
        return cls(h, n.value, b.value)

    def write(self, senscr, active=None):
        """active: delta-encoded list (uint8) or None = every senone"""
        senscr = np.ascontiguousarray(senscr, np.int16)
        if active is None:
            check(self.L.s3a_senlog_write_frame(self.h, self.n_sen, None, _p(senscr)))
        else:
            active = np.ascontiguousarray(active, np.uint8)
            check(self.L.s3a_senlog_write_frame(self.h, len(active), _p(active) if len(active) else None, _p(senscr)))

    def read(self):
        """-> (senscr int16[n_sen], active uint8[n_active]) or None at the end of the file"""
        scr = np.zeros(self.n_sen, np.int16)
        act = np.zeros(self.n_sen, np.uint8)
        n = C.c_int32()
        rc = self.L.s3a_senlog_read_frame(self.h, _p(scr), _p(act), C.byref(n))
        if rc == 0:
            return None
        if rc < 0:
            check(rc)
        return scr, act[:n.value].copy()

    def close(self):
        if self.h:
            self.L.s3a_senlog_close(self.h)
            self.h = None

    __del__ = close


class FeParams(C.Structure):
    """s3a_fe_params_t: the front-end options of sphinxbase's fe (fe.h:100-215)."""
    _fields_ = [("samprate", C.c_float), ("frate", C.c_int32), ("wlen", C.c_float), ("alpha", C.c_float),
                ("ncep", C.c_int32), ("nfft", C.c_int32), ("nfilt", C.c_int32), ("lowerf", C.c_float),
                ("upperf", C.c_float), ("transform", C.c_int32), ("lifter", C.c_int32), ("remove_dc", C.c_int32),
                ("round_filters", C.c_int32), ("unit_area", C.c_int32), ("doublebw", C.c_int32),
                ("logspec", C.c_int32), ("warp_type", C.c_int32), ("warp_params", C.c_float * 2)]


FE_TRANSFORMS = {"legacy": 0, "dct": 1, "htk": 2}


class FrontEnd:
    """The MFCC front end on the device (fe_init_auto_r / fe_process_utt / fe_end_utt); options by the
    reference's names without the dash, e.g. FrontEnd(samprate=11025, nfilt=36, transform="dct")."""

    def __init__(self, **opts):
        self.L = load()
        p = FeParams()
        self.L.s3a_fe_default_params(C.byref(p))
        for k, v in opts.items():
            if k == "transform":
                v = FE_TRANSFORMS[v] if isinstance(v, str) else v
            if k == "smoothspec":
                k, v = "logspec", (2 if v else p.logspec)
            if not hasattr(p, k):
                raise TypeError("unknown front-end option " + k)
            setattr(p, k, v)
        self.params = p
        self.h = self.L.s3a_fe_init(C.byref(p))
        if not self.h:
            raise RuntimeError("s3a_fe_init: " + self.L.s3a_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None):
            self.L.s3a_fe_free(self.h)
            self.h = None

    @property
    def output_size(self):
        return self.L.s3a_fe_output_size(self.h)

    @property
    def frame_shift(self):
        return self.L.s3a_fe_frame_shift(self.h)

    @property
    def frame_size(self):
        return self.L.s3a_fe_frame_size(self.h)

    def n_frames(self, nsamps):
        return self.L.s3a_fe_n_frames(self.h, nsamps)

    def process_utt(self, spch):
        """int16 samples -> cepstra [n_frames][output_size] (the final partial frame included)."""
        spch = np.ascontiguousarray(spch, np.int16)
        n = self.n_frames(len(spch))
        out = np.zeros((max(n, 1), self.output_size), np.float32)
        got = C.c_int32(0)
        check(self.L.s3a_fe_process_utt(self.h, _p(spch), len(spch), _p(out), n, C.byref(got)))
        return out[:got.value]


class PsMsMgau:
    """pocketsphinx's continuous scorer (the object behind ps_mgaufuncs_t): int16 negated scores, best = 0."""

    def __init__(self, h):
        self.L = load()
        if not h:
            raise S3AError(_err(self.L))
        self.h = h
        self.n_sen = self.L.s3a_ps_ms_mgau_n_sen(h)
        self.veclen = self.L.s3a_ps_ms_mgau_veclen(h)

    @classmethod
    def init(cls, meanfile, varfile, mixwfile, senmgau=".cont.", topn=4, aw=1, logbase=1.0001, varfloor=1e-4,
             mixwfloor=1e-7):
        L = load()
        e = lambda p: None if p is None else str(p).encode()
        return cls(L.s3a_ps_ms_mgau_init(e(meanfile), e(varfile), varfloor, e(mixwfile), mixwfloor, e(senmgau), int(topn),
                                         int(aw), float(logbase)))

    @classmethod
    def init_arrays(cls, mean, var, mixw, n_mgau, n_density, featlen, topn, aw=1, logbase=1.0001, sen2mgau=None,
                    varfloor=1e-4, mixwfloor=1e-7):
        L = load()
        mean = np.ascontiguousarray(mean, np.float32); var = np.ascontiguousarray(var, np.float32)
        mixw = np.ascontiguousarray(mixw, np.float32)
        fl = np.ascontiguousarray(featlen, np.int32)
        n_sen = mixw.size // (len(fl) * n_density)
        s2m = None if sen2mgau is None else np.ascontiguousarray(sen2mgau, np.int32)
        return cls(L.s3a_ps_ms_mgau_init_arrays(_p(mean), _p(var), _p(mixw), int(n_mgau), len(fl), int(n_density), _p(fl),
                                                n_sen, None if s2m is None else _p(s2m), varfloor, mixwfloor, int(topn),
                                                int(aw), float(logbase)))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.s3a_ps_ms_mgau_free(self.h)
            self.h = None

    def frame_eval(self, senscr, feat, active_list=None, frame=0):
        """ps_mgau_frame_eval: senscr int16[S] in place; active_list = the delta-encoded uint8 list
        (acmod_flags2list) or None for compallsen."""
        x = np.ascontiguousarray(feat, np.float32)
        assert senscr.dtype == np.int16 and senscr.flags.c_contiguous
        if active_list is None:
            check(self.L.s3a_ps_ms_cont_mgau_frame_eval(self.h, _p(senscr), None, 0, _p(x), int(frame), 1))
        else:
            lst = np.ascontiguousarray(active_list, np.uint8)
            check(self.L.s3a_ps_ms_cont_mgau_frame_eval(self.h, _p(senscr), _p(lst) if len(lst) else _p(np.zeros(1, np.uint8)),
                                                        len(lst), _p(x), int(frame), 0))


class MsMgau:
    """ms_mgau_model_t: the multi-stream scorer behind -senmgau .s3cont. / .semi. (s3a_ms_mgau_*)."""

    def __init__(self, h, lm, n_mgau=None, n_feat=None):
        self.L = load()
        if not h:
            raise S3AError(_err(self.L))
        self.h, self.lm = h, lm
        self.n_sen = self.L.s3a_ms_mgau_n_sen(h)
        self.topn = self.L.s3a_ms_mgau_topn(h)
        self.veclen = self.L.s3a_ms_mgau_veclen(h)
        self.n_mgau, self.n_feat = n_mgau, n_feat

    @classmethod
    def init(cls, meanfile, varfile, mixwfile, lm: LogMath, senmgau=".s3cont.", topn=4, varfloor=1e-4,
             mixwfloor=1e-7, lambdafile=None):
        L = load()
        e = lambda p: None if p is None else str(p).encode()
        return cls(L.s3a_ms_mgau_init(e(meanfile), e(varfile), varfloor, e(mixwfile), mixwfloor, 1, e(senmgau),
                                      e(lambdafile), int(topn), lm.h), lm)

    @classmethod
    def init_arrays(cls, mean, var, mixw, n_mgau, n_density, featlen, lm: LogMath, topn, sen2mgau=None,
                    varfloor=1e-4, mixwfloor=1e-7):
        L = load()
        mean = np.ascontiguousarray(mean, np.float32); var = np.ascontiguousarray(var, np.float32)
        mixw = np.ascontiguousarray(mixw, np.float32)
        fl = np.ascontiguousarray(featlen, np.int32)
        n_sen = mixw.size // (len(fl) * n_density)
        s2m = None if sen2mgau is None else np.ascontiguousarray(sen2mgau, np.int32)
        return cls(L.s3a_ms_mgau_init_arrays(_p(mean), _p(var), _p(mixw), int(n_mgau), len(fl), int(n_density), _p(fl),
                                             n_sen, None if s2m is None else _p(s2m), varfloor, mixwfloor, int(topn),
                                             lm.h), lm, n_mgau, len(fl))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.s3a_ms_mgau_free(self.h)
            self.h = None

    def frame_eval(self, sen_active, senscr, feat, frame=0):
        """ms_cont_mgau_frame_eval: senscr (int32[S]) updated in place for the active senones; returns best."""
        sa = np.ascontiguousarray(sen_active, np.uint8)
        x = np.ascontiguousarray(feat, np.float32)
        assert senscr.dtype == np.int32 and senscr.flags.c_contiguous and x.size == self.veclen
        best = C.c_int32(0)
        check(self.L.s3a_ms_cont_mgau_frame_eval(self.h, _p(sa), _p(senscr), _p(x), int(frame), C.byref(best)))
        return best.value

    def last_dist(self):
        n = self.n_mgau * self.n_feat * self.topn
        d = np.zeros(n, np.int32); di = np.zeros(n, np.int32)
        check(self.L.s3a_ms_mgau_get_dist(self.h, _p(d), _p(di)))
        shp = (self.n_mgau, self.n_feat, self.topn)
        return d.reshape(shp), di.reshape(shp)


class Batch:
    """B decoders per kernel launch (s3a_batch_*): attach (LexSearch, Scorer, ComSen) triples, then per
    frame transition() + submit() for every decoder inside an utterance and one run()."""

    def __init__(self, max_slots):
        self.L = load()
        self.h = self.L.s3a_batch_create(int(max_slots))
        if not self.h:
            raise S3AError(_err(self.L))
        self.members, self.pending = [], {}

    def __del__(self):
        if getattr(self, "h", None):
            self.L.s3a_batch_free(self.h)
            self.h = None

    def attach(self, ls: "LexSearch", sc: "Scorer", cs: "ComSen"):
        slot = self.L.s3a_batch_attach(self.h, ls.h, sc.h, cs.h)
        check(slot if slot < 0 else 0)
        self.members.append((ls, sc, cs))
        return slot

    def utt_begin(self, slot):
        check(self.L.s3a_batch_utt_begin(self.h, int(slot)))

    def utt_end(self, slot):
        check(self.L.s3a_batch_utt_end(self.h, int(slot)))

    def transition(self, slot, cf, thresh, a=None, b=None):
        def unpack(g):
            if g is None:
                z = np.zeros(0, np.int32)
                return 0, z, z, z
            return int(g[0]), *(np.ascontiguousarray(x, np.int32) for x in g[1:])
        ta, la, sa, ha = unpack(a)
        tb, lb, sb, hb = unpack(b)
        check(self.L.s3a_batch_transition(self.h, int(slot), int(cf), int(thresh), ta, len(la), _p(la), _p(sa), _p(ha),
                                          tb, len(lb), _p(lb), _p(sb), _p(hb)))

    def _bufs(self, slot):
        ls = self.members[slot][0]
        cap = ls.T * ls.max_node
        return dict(res=FrameResult(), n=np.zeros(ls.T, np.int32), w=np.zeros(cap, np.int32),
                    s=np.zeros(cap, np.int32), h=np.zeros(cap, np.int32), cap=cap, T=ls.T)

    def _call(self, fn, slot, feat, frame, frm, hmmbeam, pbeam, wbeam, phone_uses_wbeam, maxhmmpf):
        o = self._bufs(slot)
        x = np.ascontiguousarray(feat, np.float32)
        o["x"] = x
        check(fn(self.h, int(slot), _p(x), int(frame), int(frm), int(hmmbeam), int(pbeam), int(wbeam),
                 int(phone_uses_wbeam), int(maxhmmpf), C.byref(o["res"]), _p(o["n"]), _p(o["w"]), _p(o["s"]),
                 _p(o["h"]), o["cap"]))
        return o

    @staticmethod
    def _result(o):
        off = np.concatenate([[0], np.cumsum(o["n"])])
        return o["res"], [(o["w"][off[t]:off[t + 1]], o["s"][off[t]:off[t + 1]], o["h"][off[t]:off[t + 1]])
                          for t in range(o["T"])]

    def submit(self, slot, feat, frame, frm, hmmbeam, pbeam, wbeam, phone_uses_wbeam=0, maxhmmpf=20000):
        self.pending[slot] = self._call(self.L.s3a_batch_submit, slot, feat, frame, frm, hmmbeam, pbeam, wbeam,
                                        phone_uses_wbeam, maxhmmpf)

    def run(self):
        """Execute the step for every submitted slot; returns {slot: (FrameResult, exits per tree)}."""
        check(self.L.s3a_batch_run(self.h))
        out = {s: self._result(o) for s, o in self.pending.items()}
        self.pending = {}
        return out

    def step(self, slot, feat, frame, frm, hmmbeam, pbeam, wbeam, phone_uses_wbeam=0, maxhmmpf=20000):
        """Blocking rendezvous (one host thread per decoder)."""
        return self._result(self._call(self.L.s3a_batch_step, slot, feat, frame, frm, hmmbeam, pbeam, wbeam,
                                       phone_uses_wbeam, maxhmmpf))

    def stats(self):
        a, b = C.c_int64(0), C.c_int64(0)
        check(self.L.s3a_batch_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value


class ComSen:
    """dict2pid_comsenscr over flattened comstate lists."""

    def __init__(self, comstate_off, comstate, comwt):
        self.L = load()
        self.off = np.ascontiguousarray(comstate_off, np.int32)
        self.lst = np.ascontiguousarray(comstate, np.int16)
        self.wt = np.ascontiguousarray(comwt, np.int32)
        self.n = len(self.wt)
        self.h = self.L.s3a_comsen_init(self.n, _p(self.off), _p(self.lst), _p(self.wt))
        if not self.h:
            raise S3AError(_err(self.L))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.s3a_comsen_free(self.h)
            self.h = None

    def comsenscr(self, senscr):
        senscr = np.ascontiguousarray(senscr, np.int32)
        out = np.zeros(self.n, np.int32)
        check(self.L.s3a_dict2pid_comsenscr(self.h, _p(senscr), len(senscr), _p(out)))
        return out


class Tmat:
    """tmat_t: transition matrices converted to logs3 (host-side object)."""

    def __init__(self, h, lm):
        self.L = load()
        self.h = h
        self.lm = lm
        self.n_tmat = self.L.s3a_tmat_n_tmat(h)
        self.n_state = self.L.s3a_tmat_n_state(h)

    @classmethod
    def init(cls, path, lm: LogMath, tpfloor=1e-4):
        L = load()
        h = L.s3a_tmat_init(path.encode(), tpfloor, 0, lm.h)
        if not h:
            raise S3AError(_err(L))
        return cls(h, lm)

    @classmethod
    def init_arrays(cls, tp, lm: LogMath, tpfloor=1e-4):
        L = load()
        tp = np.ascontiguousarray(tp, np.float32)
        h = L.s3a_tmat_init_arrays(_p(tp), tp.shape[0], tp.shape[1], tpfloor, lm.h)
        if not h:
            raise S3AError(_err(L))
        return cls(h, lm)

    @classmethod
    def init_logs3(cls, tp_int):
        L = load()
        tp_int = np.ascontiguousarray(tp_int, np.int32)
        h = L.s3a_tmat_init_logs3(_p(tp_int), tp_int.shape[0], tp_int.shape[1])
        if not h:
            raise S3AError(_err(L))
        return cls(h, None)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.s3a_tmat_free(self.h)
            self.h = None

    @property
    def tp(self):
        out = np.zeros((self.n_tmat, self.n_state, self.n_state + 1), np.int32)
        check(self.L.s3a_tmat_get_tp(self.h, _p(out)))
        return out


class HmmBatch:
    """An array of hmm_t on the GPU (structure of arrays) + hmm_vit_eval on all of them."""

    def __init__(self, n_hmm, tmat: Tmat, sseq, n_sen):
        self.L = load()
        self.tmat = tmat
        self.sseq = np.ascontiguousarray(sseq, np.int16)
        self.n = int(n_hmm)
        self.ne = tmat.n_state
        self.h = self.L.s3a_hmm_batch_init(self.n, self.ne, tmat.h, _p(self.sseq),
                                           self.sseq.shape[0], int(n_sen))
        if not self.h:
            raise S3AError(_err(self.L))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.s3a_hmm_batch_free(self.h)
            self.h = None

    def setup(self, mpx, ssid, tmatid):
        check(self.L.s3a_hmm_batch_setup(self.h, _p(np.ascontiguousarray(mpx, np.uint8)),
                                         _p(np.ascontiguousarray(ssid, np.int32)),
                                         _p(np.ascontiguousarray(tmatid, np.int32))))

    def clear(self, idx=None):
        if idx is None:
            check(self.L.s3a_hmm_batch_clear(self.h, None, 0))
        else:
            idx = np.ascontiguousarray(idx, np.int32)
            check(self.L.s3a_hmm_batch_clear(self.h, _p(idx), len(idx)))

    def enter(self, idx, score, histid, frame):
        idx = np.ascontiguousarray(idx, np.int32)
        check(self.L.s3a_hmm_batch_enter(self.h, _p(idx), _p(np.ascontiguousarray(score, np.int32)),
                                         _p(np.ascontiguousarray(histid, np.int64)), len(idx), int(frame)))

    def vit_eval(self, senscr):
        senscr = np.ascontiguousarray(senscr, np.int32)
        ret = np.zeros(self.n, np.int32)
        check(self.L.s3a_hmm_batch_vit_eval(self.h, _p(senscr), _p(ret)))
        return ret

    def get(self):
        n = self.n
        score = np.zeros((n, 5), np.int32); hist = np.zeros((n, 5), np.int64)
        out_score = np.zeros(n, np.int32); out_hist = np.zeros(n, np.int64)
        best = np.zeros(n, np.int32); mpx_ssid = np.zeros((n, 5), np.int32); frame = np.zeros(n, np.int32)
        check(self.L.s3a_hmm_batch_get(self.h, _p(score), _p(hist), _p(out_score), _p(out_hist),
                                       _p(best), _p(mpx_ssid), _p(frame)))
        return dict(score=score, hist=hist, out_score=out_score, out_hist=out_hist, bestscore=best,
                    mpx_ssid=mpx_ssid, frame=frame)


class FrameResult(C.Structure):
    _fields_ = [("best_hmm", C.c_int32), ("best_word", C.c_int32), ("n_hmm", C.c_int32),
                ("thres", C.c_int32), ("phone_thres", C.c_int32), ("word_thres", C.c_int32),
                ("need_histprune", C.c_int32), ("n_exit_total", C.c_int32), ("extra", C.c_int32 * 8)]


class LexSearch:
    """All lextrees of one decoder on the GPU (s3a_lexsearch_*): mode 4's per-frame search ops."""

    def __init__(self, trees, tmat: "Tmat", sseq, comsseq, comstate_off, comstate, n_sen, stream=None):
        self.L = load()
        self.trees = trees
        self.T = len(trees)
        self.n_sen = int(n_sen)
        self.tmat = tmat
        keep = self._keep = []

        def col(key, dt):
            arrs = [np.ascontiguousarray(t[key], dt) for t in trees]
            keep.append(arrs)
            ptrs = (C.c_void_p * self.T)(*[a.ctypes.data_as(C.c_void_p).value if a.size else None for a in arrs])
            keep.append(ptrs)
            return ptrs

        def ints(key):
            a = np.array([int(t[key]) for t in trees], np.int32)
            keep.append(a)
            return _p(a)
        self.sseq = np.ascontiguousarray(sseq, np.int16)
        self.comsseq = np.ascontiguousarray(comsseq, np.int16)
        self.comstate_off = np.ascontiguousarray(comstate_off, np.int32)
        self.comstate = np.ascontiguousarray(comstate, np.int16)
        ne = tmat.n_state
        self.h = self.L.s3a_lexsearch_init(
            self.T, ints("n_node"), col("ssid", np.int32), col("tmatid", np.int32), col("composite", np.uint8),
            col("wid", np.int32), col("prob", np.int32), col("child_off", np.int32), col("child", np.int32),
            ints("n_lc"), col("lc", np.int16), col("lcroot_off", np.int32), col("lcroot", np.int32),
            ints("n_root"), col("root", np.int32), tmat.h, _p(self.sseq), len(self.sseq) // ne,
            _p(self.comsseq), len(self.comsseq) // ne, len(self.comstate_off) - 1, _p(self.comstate_off),
            _p(self.comstate), stream)
        if not self.h:
            raise S3AError(_err(self.L))
        self.max_node = max(int(t["n_node"]) for t in trees)
        self.d_sen = DevBuf(4 * self.n_sen)
        self.d_com = DevBuf(4 * max(len(self.comstate_off) - 1, 1))
        self.d_act = DevBuf(self.n_sen)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.s3a_lexsearch_free(self.h)
            self.h = None

    def clone(self, stream=None):
        """A second decoder over the same trees: static device arrays shared, own state."""
        c = object.__new__(LexSearch)
        c.__dict__.update({k: v for k, v in self.__dict__.items() if k not in ("h", "d_sen", "d_com", "d_act")})
        c._proto = self                 # keeps the prototype (owner of the static arrays) alive
        c.h = self.L.s3a_lexsearch_clone(self.h, stream)
        if not c.h:
            raise S3AError(_err(self.L))
        c.d_sen = DevBuf(4 * self.n_sen)
        c.d_com = DevBuf(4 * max(len(self.comstate_off) - 1, 1))
        c.d_act = DevBuf(self.n_sen)
        return c

    def enter(self, t, lc, scr, hist, cf, thresh):
        lc = np.ascontiguousarray(lc, np.int32); scr = np.ascontiguousarray(scr, np.int32)
        hist = np.ascontiguousarray(hist, np.int32)
        check(self.L.s3a_lexsearch_enter(self.h, int(t), len(lc), _p(lc), _p(scr), _p(hist), int(cf), int(thresh)))

    def swap(self):
        check(self.L.s3a_lexsearch_active_swap(self.h))

    def _upload_scores(self, senscr, comsen):
        self.d_sen.upload(np.ascontiguousarray(senscr, np.int32))
        if len(comsen):
            self.d_com.upload(np.ascontiguousarray(comsen, np.int32))

    def hmm_eval(self, senscr, comsen, frm):
        self._upload_scores(senscr, comsen)
        b = np.zeros(self.T, np.int32); w = np.zeros(self.T, np.int32); n = np.zeros(self.T, np.int32)
        check(self.L.s3a_lexsearch_hmm_eval(self.h, self.d_sen.ptr, self.d_com.ptr, int(frm), _p(b), _p(w), _p(n)))
        return b, w, n

    def propagate(self, cf, th, pth, wth):
        check(self.L.s3a_lexsearch_propagate_non_leaves(self.h, int(cf), int(th), int(pth), int(wth)))

    def leaves(self, wth):
        m = self.max_node
        n = np.zeros(self.T, np.int32)
        w = np.zeros(self.T * m, np.int32); s = np.zeros(self.T * m, np.int32); h = np.zeros(self.T * m, np.int32)
        check(self.L.s3a_lexsearch_propagate_leaves(self.h, int(wth), _p(n), _p(w), _p(s), _p(h), m))
        return [(w[t * m: t * m + n[t]], s[t * m: t * m + n[t]], h[t * m: t * m + n[t]]) for t in range(self.T)]

    def frame_search(self, senscr, comsen, frm, hmmbeam, pbeam, wbeam, phone_uses_wbeam=0, maxhmmpf=20000):
        self._upload_scores(senscr, comsen)
        res = FrameResult()
        cap = self.T * self.max_node
        n = np.zeros(self.T, np.int32)
        w = np.zeros(cap, np.int32); s = np.zeros(cap, np.int32); h = np.zeros(cap, np.int32)
        check(self.L.s3a_lexsearch_frame_search(self.h, self.d_sen.ptr, self.d_com.ptr, int(frm), int(hmmbeam),
                                                int(pbeam), int(wbeam), int(phone_uses_wbeam), int(maxhmmpf), None,
                                                C.byref(res), _p(n), _p(w), _p(s), _p(h), cap))
        off = np.concatenate([[0], np.cumsum(n)])
        return res, [(w[off[t]:off[t + 1]], s[off[t]:off[t + 1]], h[off[t]:off[t + 1]]) for t in range(self.T)]

    def active(self, t, which):
        n = C.c_int32(0)
        nodes = np.zeros(self.max_node, np.int32)
        check(self.L.s3a_lexsearch_get_active(self.h, int(t), int(which), C.byref(n), _p(nodes), self.max_node))
        return nodes[:n.value].copy()

    def state(self, t):
        nn = int(self.trees[t]["n_node"])
        sc = np.zeros((3, nn), np.int32); hi = np.zeros((3, nn), np.int32)
        o = np.zeros(nn, np.int32); oh = np.zeros(nn, np.int32); b = np.zeros(nn, np.int32); f = np.zeros(nn, np.int32)
        check(self.L.s3a_lexsearch_get_hmm(self.h, int(t), _p(sc), _p(hi), _p(o), _p(oh), _p(b), _p(f)))
        return np.stack([sc[0], sc[1], sc[2], hi[0], hi[1], hi[2], o, oh, b, f], 1)

    def sen_active(self):
        check(self.L.s3a_lexsearch_sen_active(self.h, self.d_act.ptr, self.n_sen))
        check(self.L.s3a_dev_sync())
        return self.d_act.download(np.uint8, (self.n_sen,))

    def utt_end(self):
        check(self.L.s3a_lexsearch_utt_end(self.h))

    def histbin(self, t, bestscr, bins, bw):
        """lextree_hmm_histbin on tree t: adds to bins (int32 array) in place, reorders the active list."""
        assert bins.dtype == np.int32 and bins.flags.c_contiguous
        check(self.L.s3a_lexsearch_hmm_histbin(self.h, int(t), int(bestscr), _p(bins), len(bins), int(bw)))

    # ---- the fused frame (s3a_decoder_*): scorer + composite table + lextrees together ----
    def decoder_utt_begin(self, sc: "Scorer"):
        check(self.L.s3a_decoder_utt_begin(self.h, sc.h))

    def decoder_score(self, sc: "Scorer", feat, frame):
        feat = np.ascontiguousarray(feat, np.float32)
        check(self.L.s3a_decoder_score(sc.h, _p(feat), int(frame)))

    def decoder_search(self, sc: "Scorer", cs: "ComSen", frm, hmmbeam, pbeam, wbeam, phone_uses_wbeam=0,
                       maxhmmpf=20000):
        res = FrameResult()
        cap = self.T * self.max_node
        n = np.zeros(self.T, np.int32)
        w = np.zeros(cap, np.int32); s = np.zeros(cap, np.int32); h = np.zeros(cap, np.int32)
        check(self.L.s3a_decoder_search(self.h, sc.h, cs.h, int(frm), int(hmmbeam), int(pbeam), int(wbeam),
                                        int(phone_uses_wbeam), int(maxhmmpf), C.byref(res), _p(n), _p(w),
                                        _p(s), _p(h), cap))
        off = np.concatenate([[0], np.cumsum(n)])
        return res, [(w[off[t]:off[t + 1]], s[off[t]:off[t + 1]], h[off[t]:off[t + 1]]) for t in range(self.T)]

    def decoder_transition(self, sc: "Scorer", cs: "ComSen", cf, thresh, a=None, b=None):
        """a, b = (tree, lc[], scr[], hist[]) or None: this frame's lextree_enter calls; then swap."""
        def unpack(g):
            if g is None:
                z = np.zeros(0, np.int32)
                return 0, z, z, z
            return int(g[0]), *(np.ascontiguousarray(x, np.int32) for x in g[1:])
        ta, la, sa, ha = unpack(a)
        tb, lb, sb, hb = unpack(b)
        check(self.L.s3a_decoder_transition(self.h, sc.h, cs.h, int(cf), int(thresh), ta, len(la), _p(la), _p(sa),
                                            _p(ha), tb, len(lb), _p(lb), _p(sb), _p(hb)))


class WordLevelCfg(C.Structure):
    """s3a_wordlevel_cfg_t"""
    _fields_ = [("n_word", C.c_int32), ("n_ci", C.c_int32), ("lwid", C.c_void_p), ("is_filler", C.c_void_p),
                ("fillpen", C.c_void_p), ("last_ci", C.c_void_p)] + \
               [(k, C.c_int32) for k in ("startwid", "finishwid", "silwid", "start_lwid", "finish_lwid", "sil_ci", "wbeam_vh",
                                         "bghist", "maxwpf", "maxhistpf", "wordend_beam", "n_lextree", "epl", "hmmbeam",
                                         "pbeam", "wbeam", "ptranskip", "maxhmmpf")] + [("tree_type", C.c_void_p)]


class Lm3g:
    """lm_t flattened (s3a_lm3g_init); t = dict with the arrays of a word-level trace"""

    def __init__(self, t, host_only=False):
        """host_only: no device arrays (s3a_lm3g_init_host) -- for the library's host-side consumers on a machine without a GPU"""
        self.L = load()
        g = lambda k: np.ascontiguousarray(t[k], dtype=np.int32) if k in t and len(np.atleast_1d(t[k])) else None
        self.keep = [g(k) for k in ("ug_prob", "ug_bowt", "ug_firstbg", "bg_wid", "bg_prob", "bg_bowt", "bg_firsttg",
                                    "tg_wid", "tg_prob")]
        a = self.keep
        init = self.L.s3a_lm3g_init_host if host_only else self.L.s3a_lm3g_init
        self.h = init(int(t["n_ug"]), _p(a[0]), _p(a[1]), _p(a[2]), int(t["n_bg"]), _p(a[3]), _p(a[4]),
                      _p(a[5]), _p(a[6]), int(t["n_tg"]), _p(a[7]), _p(a[8]), None, int(t["n_word"]))
        if not self.h:
            raise S3AError(_err(self.L))

    def tg_score(self, lw1, lw2, lw3, wid=0):
        return int(self.L.s3a_lm3g_tg_score(self.h, lw1, lw2, lw3, wid))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.s3a_lm3g_free(self.h)
            self.h = None


def wordlevel_cfg(t, tree_type, keep, wordend=None, maxwpf=None, maxhist=None, hmmbeam=-1000000):
    g = lambda k, dt=np.int32: np.ascontiguousarray(t[k], dtype=dt)
    keep += [g("lwid"), g("is_filler", np.uint8), g("fillpen"), g("last_ci"), np.ascontiguousarray(tree_type, dtype=np.int32)]
    a = keep[-5:]
    c = WordLevelCfg()
    c.n_word, c.n_ci = int(t["n_word"]), int(t["n_ci"])
    c.lwid, c.is_filler, c.fillpen, c.last_ci, c.tree_type = (x.ctypes.data for x in a)
    for k in ("startwid", "finishwid", "silwid", "start_lwid", "finish_lwid", "bghist", "n_lextree", "epl"):
        setattr(c, k, int(t[k]))
    c.sil_ci = int(t["n_ci"]) - 1
    c.wbeam_vh = int(t["wbeam"])
    c.maxwpf = int(t["maxwpf"] if maxwpf is None else maxwpf)
    c.maxhistpf = int(t["maxhistpf"] if maxhist is None else maxhist)
    c.wordend_beam = int(t.get("wordend", 0) if wordend is None else wordend)
    c.hmmbeam = hmmbeam
    c.pbeam = c.wbeam = hmmbeam
    c.maxhmmpf = 100000
    return c


class WlTest:
    """The device word level on its own (s3a_wltest_*): one frame per call, next to the oracle."""

    def __init__(self, t, tree_type, cap=1 << 16, cand_cap=1 << 16, max_exits=4096, max_frames=1024, **kw):
        self.L = load()
        self.lm = Lm3g(t)
        self.keep = []
        self.cfg = wordlevel_cfg(t, tree_type, self.keep, **kw)
        self.T, self.n_ci = len(tree_type), int(t["n_ci"])
        self.lcmap_len = np.full((self.T, self.n_ci + 1), 3, np.int32)      # every (tree, context) has a 3-node root list
        self.h = self.L.s3a_wltest_init(self.lm.h, C.byref(self.cfg), _p(self.lcmap_len), cap, cand_cap, max_exits, max_frames)
        if not self.h:
            raise S3AError(_err(self.L))
        self.cap, self.max_frames, self.startwid = cap, max_frames, int(t["startwid"])
        check(self.L.s3a_wltest_begin(self.h, self.startwid, max_frames), self.L)

    def frame(self, trees, best_hmm, best_word, word_thres):
        n_exit = np.array([len(t[1]) for t in trees], np.int32)
        ex = np.zeros((int(n_exit.sum()), 3), np.int32)
        k = 0
        for _, wid, scr, hist in trees:
            n = len(wid)
            ex[k:k + n, 0], ex[k:k + n, 1], ex[k:k + n, 2] = wid, scr, hist
            k += n
        calls = np.zeros(4 * 128, np.int32)
        nc, th, ne = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        check(self.L.s3a_wltest_frame(self.h, _p(n_exit), _p(ex), int(best_hmm), int(best_word), int(word_thres),
                                      C.byref(nc), _p(calls), C.byref(th), C.byref(ne)), self.L)
        return calls[: 4 * nc.value].reshape(-1, 4).copy(), th.value, ne.value

    def table(self):
        n, nf, nt = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        check(self.L.s3a_wltest_fetch(self.h, C.byref(n), C.byref(nf), None, 0, None, 0, C.byref(nt)), self.L)
        out = np.zeros((10, max(n.value, 1)), np.int32)[:, : n.value].copy()
        fr = np.zeros((3, nf.value + 1), np.int32)
        check(self.L.s3a_wltest_fetch(self.h, C.byref(n), C.byref(nf), _p(out), n.value, _p(fr), nf.value + 1, C.byref(nt)), self.L)
        d = {k: out[i] for i, k in enumerate(("score", "pred", "lw0", "lw1", "wid", "sf", "ef", "ascr", "lscr", "type"))}
        d.update(frame_start=fr[0], bestscore=fr[1], bestvh=fr[2], n_tie_frames=nt.value)
        return d

    def __del__(self):
        if getattr(self, "h", None):
            self.L.s3a_wltest_free(self.h)
            self.h = None


HYP_MAXW = 250


class HypWord(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("wid", "sf", "ef", "ascr", "lscr", "scale")]


class HypRecord(C.Structure):
    """s3a_hyp_record_t: one utterance's hypothesis, fixed size (the unit of the end-of-batch gather)"""
    _fields_ = [("uttid", C.c_char * 96)] + \
               [(k, C.c_int32) for k in ("utt_index", "n_words", "n_frames", "score", "total_scale", "n_entry", "status", "exit_id")] + \
               [("word", HypWord * HYP_MAXW)]


class HypHeader(C.Structure):
    """s3a_hyp_header_t: what s3a_hyp_record_t begins with; the words travel beside it (no word limit)"""
    _fields_ = [("uttid", C.c_char * 96)] + \
               [(k, C.c_int32) for k in ("utt_index", "n_words", "n_frames", "score", "total_scale", "n_entry", "status", "exit_id")]


class DagCfg(C.Structure):
    """s3a_dag_cfg_t"""
    _fields_ = [("n_word", C.c_int32), ("basewid", C.c_void_p), ("is_filler", C.c_void_p), ("lwid", C.c_void_p), ("fillpen", C.c_void_p)] + \
               [(k, C.c_int32) for k in ("startwid", "finishwid", "silwid", "start_lwid", "finish_lwid", "wip")] + [("lwf", C.c_double)] + \
               [(k, C.c_int32) for k in ("min_endfr", "maxedge", "maxlmop", "maxlpf")]


class LatInfo(C.Structure):
    """s3a_lat_info_t"""
    _fields_ = [(k, C.c_int32) for k in ("status", "n_frames", "n_nodes", "n_links", "initial", "final", "final_ascr")]


class NbestOpts(C.Structure):
    """s3a_nbest_opts_t"""
    _fields_ = [("uttid", C.c_char_p), ("beam", C.c_double)] + [(k, C.c_int32) for k in ("beam_logs3", "nbest", "maxppath", "lm_wip")] + \
               [(k, C.c_float) for k in ("logbase", "lw", "wip", "lm_lw")]


def lattice_nbest(lm, cfg, opts, info, nodes, links, wordstr):
    """s3a_lattice_nbest: the N-best list of a lattice (nodes [n, 6], links [m, 5] int32 arrays in the reference's list orders, info a
    LatInfo, wordstr a list of bytes) -> (text, n_hyp, counts [pops, expansions, partial paths, bypass links], status)"""
    L = load()
    nodes = np.ascontiguousarray(nodes, np.int32); links = np.ascontiguousarray(links, np.int32)
    wp = (C.c_char_p * len(wordstr))(*wordstr)
    h = L.s3a_lattice_nbest(lm.h, C.byref(cfg), C.byref(opts), C.byref(info), _p(nodes), _p(links) if len(links) else None, wp)
    if not h:
        raise S3AError(_err(L))
    try:
        txt, ln, nh = C.c_char_p(), C.c_int64(), C.c_int32()
        cnt = (C.c_int32 * 4)()
        st = L.s3a_nbest_result(h, C.byref(txt), C.byref(ln), C.byref(nh), cnt)
        return C.string_at(txt, ln.value), int(nh.value), [int(x) for x in cnt], int(st)
    finally:
        L.s3a_nbest_free(h)


class DagResult(C.Structure):
    """s3a_dag_result_t"""
    _fields_ = [(k, C.c_int32) for k in ("status", "n_words", "n_node", "n_link", "n_bypass", "lmop", "score", "first_pass_score", "n_entry", "endid")] + \
               [(k, C.POINTER(C.c_int32)) for k in ("wid", "sf", "ef", "ascr", "lscr")]

    def words(self):
        n = self.n_words
        return np.stack([np.ctypeslib.as_array(getattr(self, k), (n,)).copy() if n else np.zeros(0, np.int32)
                         for k in ("wid", "sf", "ef", "ascr", "lscr")], axis=1) if n else np.zeros((0, 5), np.int32)


class DagTable(C.Structure):
    """s3a_dag_table_t"""
    _fields_ = [(k, C.c_int32) for k in ("n_entry", "n_frm", "endid", "n_hyp")] + \
               [(k, C.c_void_p) for k in ("wid", "sf", "ef", "ascr", "lscr", "score", "hyp_wid", "hyp_sf")]


def dag_cfg(b, keep, bestpathlw=None, min_endfr=None, maxedge=None, maxlmop=None, maxlpf=None):
    """s3a_dag_cfg_t from a bundle dict (cmusphinx_amd/bundle.py); `keep` collects the arrays the struct points into"""
    c = DagCfg()
    arrs = [np.ascontiguousarray(b["basewid"], np.int32), np.ascontiguousarray(b["is_filler"], np.uint8),
            np.ascontiguousarray(b["lwid"], np.int32), np.ascontiguousarray(b["fillpen"], np.int32)]
    keep.extend(arrs)
    c.n_word = b["n_word"]
    c.basewid, c.is_filler, c.lwid, c.fillpen = (a.ctypes.data for a in arrs)
    for k in ("startwid", "finishwid", "silwid", "start_lwid", "finish_lwid"):
        setattr(c, k, b[k])
    c.wip = b.get("wip_logs3", 0)
    lw = bestpathlw if bestpathlw is not None else b.get("bestpathlw", 0.0)
    c.lwf = float(np.float32(lw) / np.float32(b["lw"])) if lw else 1.0
    c.min_endfr = b.get("min_endfr", 3) if min_endfr is None else min_endfr
    c.maxedge = b.get("maxedge", 2000000) if maxedge is None else maxedge
    c.maxlmop = b.get("maxlmop", 100000000) if maxlmop is None else maxlmop
    c.maxlpf = b.get("maxlpf", 40000) if maxlpf is None else maxlpf
    return c


class Variants(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("scan_chained", "calls_by_copy", "batch_no_shared", "batch_no_multi",
                                         "no_frame_sync_kernel", "score_nt", "score_fpc", "resolve_sweep", "hist_sort_launch", "ps_overlap", "ps_score_by_gaussian",
                                         "kf_queue_in_order", "kf_no_relay", "kf_relay_at")]


def set_variants(**kw):
    """s3a_set_variants: force kernel variants (process-wide; objects created afterwards); no arguments = the defaults"""
    v = Variants()
    for k, x in kw.items():
        setattr(v, k, int(x))
    check(load().s3a_set_variants(C.byref(v)), load())


class Gather:
    """s3a_gather_t: the end-of-batch exchange of (header, words) hypotheses over RCCL, in C"""

    def __init__(self, rank, world, rendezvous="", run_id=0):
        """run_id != 0: s3a_gather_init_run (the rendezvous file carries the launcher's run id; no clocks compared)"""
        self.L = load()
        if run_id:
            self.h = self.L.s3a_gather_init_run(int(rank), int(world), rendezvous.encode(), int(run_id))
        else:
            self.h = self.L.s3a_gather_init(int(rank), int(world), rendezvous.encode())
        if not self.h:
            raise S3AError(_err(self.L))

    def gather(self, local, n_total):
        """local: [(HypHeader, words int32 [n, 6])] -> the whole batch's, in utterance order"""
        hdr = (HypHeader * max(len(local), 1))(*[h for h, _ in local])
        words = np.ascontiguousarray(np.concatenate([np.asarray(w, np.int32).reshape(-1, 6) for _, w in local])
                                     if local else np.zeros((0, 6), np.int32))
        check(self.L.s3a_gather_hyps(self.h, len(local), hdr, _p(words) if len(words) else None, int(n_total)), self.L)
        out = []
        for i in range(n_total):
            ph, pw = C.POINTER(HypHeader)(), C.POINTER(C.c_int32)()
            check(self.L.s3a_gather_result(self.h, i, C.byref(ph), C.byref(pw)), self.L)
            h = HypHeader.from_buffer_copy(bytes(ph.contents))
            n = h.n_words if h.status == 0 else 0
            out.append((h, np.ctypeslib.as_array(pw, (n * 6,)).reshape(n, 6).copy() if n else np.zeros((0, 6), np.int32)))
        return out

    def __del__(self):
        if getattr(self, "h", None):
            self.L.s3a_gather_free(self.h)
            self.h = None


class DagPass:
    """s3a_dagpass_t on host-provided history tables (parity tests)"""

    def __init__(self, lm: "Lm3g", cfg: DagCfg, n_lanes, max_entries, max_frames, link_cap=0, pair_cap=0):
        self.L = load()
        self._keep = (lm, cfg)
        self.h = self.L.s3a_dagpass_init(lm.h, C.byref(cfg), int(n_lanes), int(max_entries), int(max_frames), int(link_cap), int(pair_cap))
        if not self.h:
            raise S3AError(_err(self.L))

    def run(self, tables):
        """tables: list of dicts with wid sf ef ascr lscr score (int32 arrays incl. the final </s> entry), n_frm, endid,
        hyp_wid, hyp_sf -> list of DagResult"""
        arr = (DagTable * len(tables))()
        keep = []
        for t, d in zip(arr, tables):
            a = {k: np.ascontiguousarray(d[k], np.int32) for k in ("wid", "sf", "ef", "ascr", "lscr", "score", "hyp_wid", "hyp_sf")}
            keep.append(a)
            t.n_entry, t.n_frm, t.endid, t.n_hyp = len(a["wid"]), int(d["n_frm"]), int(d["endid"]), len(a["hyp_wid"])
            for k, v in a.items():
                setattr(t, k, v.ctypes.data)
        check(self.L.s3a_dagpass_run_tables(self.h, len(tables), arr), self.L)
        out = []
        for z in range(len(tables)):
            r = DagResult()
            check(self.L.s3a_dagpass_result(self.h, z, C.byref(r)), self.L)
            out.append(r)
        return out

    def __del__(self):
        if getattr(self, "h", None):
            self.L.s3a_dagpass_free(self.h)
            self.h = None


class UttResult(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("err", "n_entry", "n_frm", "n_frames")] + \
               [(k, C.POINTER(C.c_int32)) for k in ("score", "pred", "lw0", "lw1", "wid", "sf", "ef", "ascr", "lscr", "type",
                                                    "frame_start", "bestscore", "bestvh", "frame_stat")] + \
               [(k, C.c_int32) for k in ("max_cand", "max_new", "n_tie_frames")]


class UttDecOpts(C.Structure):
    """s3a_uttdec_opts_t: the engine's tuning options (every variant gives the same bits)"""
    _fields_ = [(k, C.c_int32) for k in ("many", "big_wl", "window", "window_fpc", "g_eval", "g_res", "scan_g", "gy", "sweep_k",
                                         "no_multi", "framecheck", "times", "graph", "window_max", "scan_small_from", "persist", "cluster",
                                         "score_rows_max")] + [("reserved", C.c_int32 * 1)]


class UttDec:
    """s3a_uttdec_t: whole utterances on the device, n_lanes at a time (the `decode` slot of srch_funcs_t)"""

    def __init__(self, proto: "LexSearch", g: "MgauModel", cd2cisen, n_ci_sen, comsen: "ComSen", lm: "Lm3g", cfg, n_lanes,
                 ds=1, cond_ds=0, ci_pbeam=1e-80, tighten_factor=0.5, max_cd=100000, max_frames=15000, vh_cap=0, cand_cap=0,
                 opts=None):
        """opts: dict of s3a_uttdec_opts_t fields; what it leaves out comes from the S3A_UTT_* environment variables (this
        harness reads them -- s3a_uttdec_opts_from_env -- the library does not)"""
        self.L = load()
        self._keep = (proto, g, comsen, lm, cfg, np.ascontiguousarray(cd2cisen, np.int16))
        o = UttDecOpts()
        self.L.s3a_uttdec_opts_from_env(C.byref(o))
        for k, v in (opts or {}).items():
            setattr(o, k, int(v))
        self.h = self.L.s3a_uttdec_init_opts(proto.h, g.h, _p(self._keep[5]), len(self._keep[5]), int(n_ci_sen), int(ds), int(cond_ds),
                                             float(ci_pbeam), float(tighten_factor), int(max_cd), comsen.h, lm.h, C.byref(cfg),
                                             int(n_lanes), int(max_frames), int(vh_cap), int(cand_cap), C.byref(o))
        if not self.h:
            raise S3AError(_err(self.L))
        self.n_lanes = int(n_lanes)

    def decode(self, feats):
        """feats: list of float32 [nfr, veclen] arrays (at most n_lanes)"""
        feats = [np.ascontiguousarray(f, np.float32) for f in feats]
        ptrs = (C.c_void_p * len(feats))(*[f.ctypes.data for f in feats])
        nfr = np.array([len(f) for f in feats], np.int32)
        check(self.L.s3a_uttdec_decode(self.h, len(feats), ptrs, _p(nfr), feats[0].shape[1]), self.L)
        return float(self.L.s3a_uttdec_last_decode_ms(self.h))

    def decode_dev(self, bufs, nfr, stride):
        """bufs: DevBuf objects holding [nfr, stride] float32 (stride = 4 * ceil(veclen / 4), zero padded)"""
        ptrs = (C.c_void_p * len(bufs))(*[b.ptr for b in bufs])
        n = np.ascontiguousarray(nfr, np.int32)
        check(self.L.s3a_uttdec_decode_dev(self.h, len(bufs), ptrs, _p(n), int(stride)), self.L)
        return float(self.L.s3a_uttdec_last_decode_ms(self.h))

    def decode_queue(self, feats):
        """feats: ANY number of float32 [nfr, veclen] arrays; a lane takes the queue's next utterance when its own has
        ended (s3a_uttdec_decode_queue).  Raises for the first utterance that stopped; the others' results stand."""
        feats = [np.ascontiguousarray(f, np.float32) for f in feats]
        ptrs = (C.c_void_p * len(feats))(*[f.ctypes.data for f in feats])
        nfr = np.array([len(f) for f in feats], np.int32)
        check(self.L.s3a_uttdec_decode_queue(self.h, len(feats), ptrs, _p(nfr), feats[0].shape[1]), self.L)
        return float(self.L.s3a_uttdec_last_decode_ms(self.h))

    def decode_queue_dev(self, bufs, nfr, stride):
        """the same with the features resident in HBM (bufs: DevBuf objects or raw device addresses)"""
        ptrs = (C.c_void_p * len(bufs))(*[getattr(b, "ptr", b) for b in bufs])
        n = np.ascontiguousarray(nfr, np.int32)
        check(self.L.s3a_uttdec_decode_queue_dev(self.h, len(bufs), ptrs, _p(n), int(stride)), self.L)
        return float(self.L.s3a_uttdec_last_decode_ms(self.h))

    def queue_status(self, utt):
        """-> dict(err, stopped_at, max_cand, max_new) of utterance `utt` of the last queue"""
        v = [C.c_int32() for _ in range(4)]
        check(self.L.s3a_uttdec_queue_status(self.h, int(utt), *[C.byref(x) for x in v]), self.L)
        return dict(zip(("err", "stopped_at", "max_cand", "max_new"), (x.value for x in v)))

    def queue_hyp(self, utt, uttid="", utt_index=0):
        """-> (HypHeader, words int32 [n_words, 6]) of utterance `utt` of the last queue"""
        hdr = HypHeader()
        words = np.zeros((HYP_MAXW, 6), np.int32)
        check(self.L.s3a_uttdec_queue_hyp(self.h, int(utt), uttid.encode(), int(utt_index), C.byref(hdr), _p(words), len(words)), self.L)
        if hdr.status == -3:
            words = np.zeros((hdr.n_words, 6), np.int32)
            check(self.L.s3a_uttdec_queue_hyp(self.h, int(utt), uttid.encode(), int(utt_index), C.byref(hdr), _p(words), len(words)), self.L)
        return hdr, words[:hdr.n_words if hdr.status == 0 else 0].copy()

    def result(self, lane):
        r = UttResult()
        check(self.L.s3a_uttdec_result(self.h, lane, C.byref(r)), self.L)
        n, nf = r.n_entry, r.n_frm
        out = {k: np.ctypeslib.as_array(getattr(r, k), (n,)).copy() for k in
               ("score", "pred", "lw0", "lw1", "wid", "sf", "ef", "ascr", "lscr", "type")}
        for k in ("frame_start", "bestscore", "bestvh"):
            out[k] = np.ctypeslib.as_array(getattr(r, k), (nf + 1,)).copy()
        out["frame_stat"] = np.ctypeslib.as_array(r.frame_stat, (r.n_frames * 8,)).reshape(-1, 8).copy()
        out.update(err=r.err, n_frm=nf, max_cand=r.max_cand, max_new=r.max_new, n_tie_frames=r.n_tie_frames)
        return out

    def set_profile(self, every):
        check(self.L.s3a_uttdec_set_profile(self.h, int(every)), self.L)

    def profile(self):
        """{kernel class: (summed microseconds, timed launches)} since set_profile"""
        us = (C.c_double * 24)(); n = (C.c_int64 * 24)(); names = (C.c_char_p * 24)()
        k = self.L.s3a_uttdec_profile(self.h, us, n, names, 24)
        return {names[i].decode(): (us[i], n[i]) for i in range(k) if n[i]}

    def wl_ticks(self, lane=0):
        """time lane's last utterance spent in the phases of the one-workgroup word level, in microseconds:
        [record + exits, P1, P2, P3, P4, P5, pruning, table + LM contexts, word transitions]"""
        t = (C.c_longlong * 16)()
        check(self.L.s3a_uttdec_wl_ticks(self.h, int(lane), t), self.L)
        return [0.01 * t[i] for i in range(9)]

    def frame_ticks(self, lane=0):
        """ku_frames' clock for the utterance `lane` decoded last -> (dict step -> microseconds, frames, launches, cluster);
        cluster 0: the engine runs the frame as separate launches"""
        t = (C.c_longlong * 16)(); c = C.c_int32()
        check(self.L.s3a_uttdec_frame_ticks(self.h, int(lane), t, C.byref(c)), self.L)
        names = ("enter_test", "enter_rank", "enter_apply_mark", "comsen_mark", "select", "comsen_max", "hmm_eval", "stamp_hist", "weak",
                 "resolve", "scan", "emit_word")
        d = {names[i]: 0.01 * t[i] for i in range(12)}
        d["emit_only"] = 0.01 * t[15]; d["in_launch"] = 0.01 * t[12]
        return d, int(t[13]), int(t[14]), int(c.value)

    def last_parts(self):
        """-> dict(score_ms, n_score, frames_ms, n_frames, cluster) of the last decode (all zero: it ran the frame as launches)"""
        sm, fm = C.c_double(), C.c_double()
        ns, nf, c = C.c_int32(), C.c_int32(), C.c_int32()
        check(self.L.s3a_uttdec_last_parts(self.h, C.byref(sm), C.byref(ns), C.byref(fm), C.byref(nf), C.byref(c)), self.L)
        return dict(score_ms=sm.value, n_score=ns.value, frames_ms=fm.value, n_frames=nf.value, cluster=c.value)

    def last_relay(self):
        """launches the last call's relay had behind its first (s3a_uttdec_last_relay)"""
        return int(self.L.s3a_uttdec_last_relay(self.h))

    def frame_dbg(self, lane=0):
        t = (C.c_longlong * 4)()
        check(self.L.s3a_uttdec_frame_dbg(self.h, int(lane), t), self.L)
        return [int(x) for x in t]

    def hyp(self, lane, uttid="", utt_index=0):
        rec = HypRecord()
        check(self.L.s3a_uttdec_hyp(self.h, lane, uttid.encode(), int(utt_index), C.byref(rec)), self.L)
        return rec

    def window(self):
        return int(self.L.s3a_uttdec_window(self.h))

    def enable_pheur(self, pheurtype, pl_beam, pl_window, node_ci, sen2cimap, n_ci):
        """-pheurtype 1..3: node_ci = per tree the nodes' CI phones (uint8), sen2cimap = int16 [n_ci_sen + 1]"""
        arrs = [np.ascontiguousarray(a, np.uint8) for a in node_ci]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        s2c = np.ascontiguousarray(sen2cimap, np.int16)
        check(self.L.s3a_uttdec_enable_pheur(self.h, int(pheurtype), int(pl_beam), int(pl_window), ptrs, _p(s2c), int(n_ci)), self.L)

    def selfcheck(self, lane):
        out = np.zeros(8, np.int32)
        check(self.L.s3a_uttdec_selfcheck(self.h, int(lane), _p(out)), self.L)
        return out

    def enable_bestpath(self, cfg, link_cap=0, pair_cap=0, keep_tables=True):
        self._dag_cfg = cfg
        check(self.L.s3a_uttdec_enable_bestpath(self.h, C.byref(cfg), int(link_cap), int(pair_cap), 1 if keep_tables else 0), self.L)

    def bestpath_result(self, lane):
        r = DagResult()
        check(self.L.s3a_uttdec_bestpath_result(self.h, int(lane), C.byref(r)), self.L)
        return r

    def bestpath_hyp(self, lane, uttid="", utt_index=0):
        """the second pass's hypothesis: (HypHeader, words int32 [n_words, 6])"""
        hdr = HypHeader()
        words = np.zeros((HYP_MAXW, 6), np.int32)
        check(self.L.s3a_uttdec_bestpath_hyp(self.h, lane, uttid.encode(), int(utt_index), C.byref(hdr), _p(words), len(words)), self.L)
        if hdr.status == -3:
            words = np.zeros((hdr.n_words, 6), np.int32)
            check(self.L.s3a_uttdec_bestpath_hyp(self.h, lane, uttid.encode(), int(utt_index), C.byref(hdr), _p(words), len(words)), self.L)
        return hdr, words[:hdr.n_words if hdr.status == 0 else 0].copy()

    def _lattice(self, fn, k):
        info = LatInfo()
        check(fn(self.h, int(k), C.byref(info), None, 0, None, 0), self.L)
        nodes, links = np.zeros((info.n_nodes, 6), np.int32), np.zeros((max(info.n_links, 1), 5), np.int32)
        check(fn(self.h, int(k), C.byref(info), _p(nodes), info.n_nodes, _p(links), info.n_links), self.L)
        return info, nodes, links[:info.n_links]

    def lattice(self, lane):
        """s3a_uttdec_lattice: (LatInfo, nodes [n, 6] = wid sf fef lef ascr lscr, links [m, 5] = from to ascr lscr ef) of a lock-step decode's lane"""
        return self._lattice(self.L.s3a_uttdec_lattice, lane)

    def queue_keep_lattices(self, on=True):
        check(self.L.s3a_uttdec_queue_keep_lattices(self.h, 1 if on else 0), self.L)

    def queue_lattice(self, utt):
        """s3a_uttdec_queue_lattice: the same for utterance `utt` of the last queue (queue_keep_lattices before the decode)"""
        return self._lattice(self.L.s3a_uttdec_queue_lattice, utt)

    def queue_bestpath_hyp(self, utt, uttid="", utt_index=0):
        """the second pass's hypothesis of utterance `utt` of the last queue: (HypHeader, words int32 [n_words, 6])"""
        hdr = HypHeader()
        words = np.zeros((HYP_MAXW, 6), np.int32)
        check(self.L.s3a_uttdec_queue_bestpath_hyp(self.h, int(utt), uttid.encode(), int(utt_index), C.byref(hdr), _p(words), len(words)), self.L)
        if hdr.status == -3:
            words = np.zeros((hdr.n_words, 6), np.int32)
            check(self.L.s3a_uttdec_queue_bestpath_hyp(self.h, int(utt), uttid.encode(), int(utt_index), C.byref(hdr), _p(words), len(words)), self.L)
        return hdr, words[:hdr.n_words if hdr.status == 0 else 0].copy()

    def hyp_var(self, lane, uttid="", utt_index=0):
        """-> (HypHeader, words int32 [n_words, 6]: wid sf ef ascr lscr scale), however long the hypothesis is"""
        hdr = HypHeader()
        words = np.zeros((HYP_MAXW, 6), np.int32)
        check(self.L.s3a_uttdec_hyp_var(self.h, lane, uttid.encode(), int(utt_index), C.byref(hdr), _p(words), len(words)), self.L)
        if hdr.status == -3:
            words = np.zeros((hdr.n_words, 6), np.int32)
            check(self.L.s3a_uttdec_hyp_var(self.h, lane, uttid.encode(), int(utt_index), C.byref(hdr), _p(words), len(words)), self.L)
        return hdr, words[:hdr.n_words if hdr.status == 0 else 0].copy()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.s3a_uttdec_free(self.h)
            self.h = None
