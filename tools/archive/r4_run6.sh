set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python tools/ab_env.py "pp0:GAVA_PP=0" "pp2:GAVA_PP=2" "pp1:GAVA_PP=1" --rounds 3 > $O/ab_pp_c2.txt 2>&1 || (tail -20 $O/ab_pp_c2.txt; exit 1)
cat $O/ab_pp_c2.txt
timeout -k 10 900 python tools/ab_env.py "pp0:GAVA_PP=0" "pp2:GAVA_PP=2" "pp1:GAVA_PP=1" --rounds 2 --config c5 > $O/ab_pp_c5.txt 2>&1 || (tail -20 $O/ab_pp_c5.txt; exit 1)
cat $O/ab_pp_c5.txt
