cd $GRAFT_REPO_ROOT
for n in 2 3 4; do echo "== NST $n"; GAVA_SMALL_NST=$n timeout -k 10 120 python tools/small_gemm_latency.py 2>&1 | grep "us per"; done
timeout -k 10 900 python tools/ab_env.py "nst2:GAVA_SMALL_NST=2" "nst3:GAVA_SMALL_NST=3" "nst4:GAVA_SMALL_NST=4" --rounds 3 2>&1 | grep "==\|FAILED"
