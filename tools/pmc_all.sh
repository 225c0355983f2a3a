#!/bin/bash
# PMC passes (tools/pmc.sh) for every big per-layer kernel in the form the forward launches it -> gpurun_out/pmc_final/*.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/pmc_final
for k in qkvfold attn outfold fc2fold ln; do
  bash $R/tools/pmc.sh final_$k $k --iters 3 > $R/gpurun_out/pmc_final/$k.txt 2>&1
  echo "== $k"; grep -E "^void|^\(anon|FETCH_SIZE|WRITE_SIZE|MFMA_BUSY|GRBM_GUI|BANK_CONFLICT|TCC_HIT|TCC_MISS" $R/gpurun_out/pmc_final/$k.txt | cut -c1-110
done
