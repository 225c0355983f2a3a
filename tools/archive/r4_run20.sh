cd $GRAFT_REPO_ROOT
timeout -k 10 900 python tools/ab_env.py "nst2:GAVA_SMALL_NST=2" "nst4:GAVA_SMALL_NST=4" --rounds 5 2>&1 | grep "round\|=="
