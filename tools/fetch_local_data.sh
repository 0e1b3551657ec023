#!/bin/bash
# Copy the RM1 decode task (complete CD-continuous 8-Gaussian model + dictionary + trigram
# LM + 20 cepstrum files) out of the read-only reference checkout into tests/_local_data/
# (git-ignored: 8 MB of third-party data is not committed; it travels to the GPU box with
# the gpurun snapshot).  Tests that need it skip when it is absent.
set -e
R=${1:-/root/reference}
D=$(dirname $0)/../tests/_local_data/rm1
mkdir -p $D/feat
RM=$R/sphinx4/models/acoustic/rm1
cp $RM/etc/RM1_clean_13dCep_16k_40mel_130Hz_6800Hz.1800.mdef $D/mdef
cp $RM/dict/fillerdict $RM/dict/RM.dictionary $D/
cp $RM/cd_continuous_8gau/{means,variances,mixture_weights,transition_matrices} $D/
cp $R/sphinx3/src/tests/performance/rm1/RM.2845.trigram.arpa.DMP $D/
F=$R/SphinxTrain/test/res/feat/rm
awk '{print $1}' $F/rm1_train.fileids.100 | head -${2:-20} > $D/rm.ctl
while read u; do mkdir -p $D/feat/$(dirname $u); cp $F/$u.mfc $D/feat/$u.mfc; done < $D/rm.ctl
du -sh $D

# pocketsphinx tasks for the first pass on the device (tests/test_gpu_psfwd.py): the hub4wsj semi-continuous model
# (goforward.raw with model/lm/en/turtle.*: BASELINE configs[0]'s utterance) and the Mandarin 5000-word trigram task
# (tdt_sc_8k + gigatdt.5000.DMP + mandarin_notone.dic: a real LM with unsorted n-gram runs, 97k dictionary words)
P=$(dirname $0)/../tests/_local_data/ps
mkdir -p $P/hub4wsj_sc_8k $P/tdt_sc_8k $P/zh_CN $P/raw
cp $R/pocketsphinx/model/hmm/en_US/hub4wsj_sc_8k/* $P/hub4wsj_sc_8k/
cp $R/pocketsphinx/model/hmm/zh/tdt_sc_8k/* $P/tdt_sc_8k/
cp $R/pocketsphinx/model/lm/zh_CN/gigatdt.5000.DMP $R/pocketsphinx/model/lm/zh_CN/mandarin_notone.dic $P/zh_CN/
cp $R/pocketsphinx/test/data/goforward.raw $R/pocketsphinx/test/data/numbers.raw $R/pocketsphinx/test/data/something.raw $P/raw/
chmod -R u+w $P
du -sh $P
