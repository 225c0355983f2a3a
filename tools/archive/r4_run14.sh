cd $GRAFT_REPO_ROOT
timeout -k 10 900 python tools/ab_env.py "wlo_pp2:GAVA_PREC=fp16+wlo,GAVA_PP=2" "wlo_pp1:GAVA_PREC=fp16+wlo,GAVA_PP=1" "wlo8_pp2:GAVA_PREC=fp16+wlo8,GAVA_PP=2" "wlo8_pp1:GAVA_PREC=fp16+wlo8,GAVA_PP=1" --rounds 2 2>&1 | grep "==\|FAILED"
timeout -k 10 900 python tools/ab_env.py "direct:" "twopass128:GAVA_PATCH_DIRECT=0" "twopass256:GAVA_PATCH_DIRECT=0,GAVA_PATCH_256=1" --rounds 3 2>&1 | grep "==\|FAILED"
