#!/bin/bash
# Round profile on the GPU box (run through gpurun):  bash tools/profile_round.sh <out_dir under gpurun_out>
#   (1) rocprofv3 --kernel-trace --stats of the default bench command (HIP-graph replay) -> per-kernel durations
#   (2) PMC passes, EACH in its own run with --kernel-trace only (never with sys/hip traces), eager launches so that
#       every dispatch is attributed to its kernel symbol:  HBM traffic (FETCH_SIZE | WRITE_SIZE), matrix-core
#       activity (SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_BF16/F32, SQ_BUSY_CU_CYCLES), occupancy
#       (SQ_WAVES, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE), LDS (SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE,
#       SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU), L2 (TCC_HIT_sum, TCC_MISS_sum).
# tools/pmc_report.py folds the CSVs into profiles/rNN/.
set -o pipefail
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-ade"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- $BENCH --steps 10 --warmup 3 > "$OUT/stats.log" 2>&1 || { echo "stats pass failed"; tail -5 "$OUT/stats.log"; exit 1; }
echo "stats pass done"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc$i" -o pmc -- $BENCH --steps 2 --warmup 1 --no-graph > "$OUT/pmc$i.log" 2>&1 \
    || { echo "pmc pass $i ($set) failed"; tail -5 "$OUT/pmc$i.log"; exit 1; }
  echo "pmc pass $i done: $set"
done
# keep only what the report needs (the merge back is capped at 64 MiB)
find "$OUT" -name "*agent_info.csv" -delete
du -sh "$OUT"
