#!/bin/bash
# Clock and issue counters of ONE 16-bit 3x3 layer with random and with zeroed activations (GPU box): bash tools/dvfs_pmc.sh TILE H CIN COUT
# separate --pmc passes (kernel-trace only); summary -> gpurun_out/prof/r04_dvfs_pmc.txt
TILE=${1:-431}; H=${2:-26}; CIN=${3:-256}; COUT=${4:-512}
R=$PWD; OUT=$R/gpurun_out/prof; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
: > $OUT/r04_dvfs_pmc.txt
for Z in "" 1; do
  for PASS in "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA"; do
    rm -rf /tmp/dv; P8_ZERO_X=$Z rocprofv3 --pmc $PASS --kernel-trace -d /tmp/dv -o a -- python $R/tools/conv16_one.py $TILE 30 $H $CIN $COUT > /tmp/dv.log 2>&1
    echo "# activations: ${Z:+zero}${Z:-random}  tile $TILE ${H}x${H} $CIN->$COUT" >> $OUT/r04_dvfs_pmc.txt
    python $R/tools/prof_summary.py /tmp/dv/a_results.db --pmc | grep "conv3x3_p8" | cut -c1-30,90-170 >> $OUT/r04_dvfs_pmc.txt
  done
done
cat $OUT/r04_dvfs_pmc.txt
