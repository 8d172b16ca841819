#!/bin/bash
# round-4 GPU session 3: where the fused kernel's phases spend their cycles (stamps of ablation builds), prefetch A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/s3; mkdir -p $O
timeout -k 10 300 python tools/fx_dev.py --quirk 8 > $O/fx_dev_quirk.log 2>&1 || { tail -20 $O/fx_dev_quirk.log; exit 1; }
tail -3 $O/fx_dev_quirk.log
export BLUR_FX_STAMPS=1 FX_DEV_ARGS="--no-check --fused-only 8"
for v in "base|" "xpre|-DFX_XPRE" "nobal|-DFX_NO_BALANCE" "nocommit|-DFX_ABL_NOCOMMIT" "nodswrite|-DFX_ABL_NODSWRITE" "noperm|-DFX_ABL_NOPERM" "noload|-DFX_ABL_NOLOAD" "nostore|-DFX_ABL_NOSTORE" \
         "nosplit|-DFX_ABL_NOSPLIT" "noemit|-DFX_ABL_NOEMIT" "nobarrier|-DFX_ABL_NOBARRIER" "noswap|-DFX_ABL_NOSWAP" "noxread|-DFX_ABL_NOXREAD" "nomem|-DFX_ABL_NOCOMMIT -DFX_ABL_NOLOAD -DFX_ABL_NOSTORE"; do
    name=${v%%|*}; flags=${v#*|}
    tools/fx_variants.sh "st_$name|-DFX_STAMPS $flags" > $O/var_$name.log 2>&1 || { tail -20 $O/var_$name.log; exit 1; }
    echo "$name: $(grep stamps $O/var_$name.log | sort | uniq -c | sort -rn | head -1 | sed 's/.*cycles per step://')  $(grep engine $O/var_$name.log | tail -1 | sed 's/.*ms\/step//')"
done
unset BLUR_FX_STAMPS
export FX_DEV_ARGS="--no-check --fused-only --quirk 8"
tools/fx_variants.sh "q_base|-DFX_DUMMY" "q_xpre|-DFX_XPRE" > $O/var_quirk.log 2>&1; grep -E "variant|engine" $O/var_quirk.log
