#!/bin/bash
# GPU box: the round's final measurement set -- the default bench line, the kernel-trace stats and the three PMC passes of the same binary
# (each pass in a run of its own; outputs under gpurun_out/, the summaries are copied to profiles/<round>/ by hand)
mkdir -p gpurun_out/r3
timeout -k 10 600 python bench.py > gpurun_out/r3/bench_r3_final.json 2> gpurun_out/r3/bench_r3_final.err &&
bash tools/prof.sh r3final &&
bash tools/pmc.sh r3fetch FETCH_SIZE &&
bash tools/pmc.sh r3write WRITE_SIZE &&
bash tools/pmc.sh r3mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE &&
echo final_meas_done
