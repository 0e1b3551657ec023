/* s3a_lm3g.h -- the flattened trigram (s3a_lm3g_init, s3a_utt.hip): device arrays + the host copy */
#ifndef S3A_LM3G_H
#define S3A_LM3G_H
#include <vector>
#include "s3a_wordlevel.h"

struct s3a_lm3g_s {
    WLm d;                      /* device arrays */
    int32_t n_dictword;
    std::vector<int32_t> ug_prob, ug_bowt, ug_firstbg, bg_wid, bg_prob, bg_bowt, bg_firsttg, tg_wid, tg_prob, inclass;
};
#endif
