#!/bin/bash
# round 4, GPU call 9: the FMA-rate probe built without SLP vectorisation, bare and under PMC counters (what one wave64 v_fma_f32 costs)
O=gpurun_out/r4c9; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize scripts/pk_probe.hip -o $O/pk_probe 2> /dev/null
$O/pk_probe > $O/pk_probe.txt 2>&1; cat $O/pk_probe.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES -d $O/pmc -o p -- $O/pk_probe > $O/pmc.log 2>&1
python scripts/rocpd_summary.py $O/pmc/p_results.db 2>&1 | grep -A100 "PMC counters" | cut -c1-200 > $O/pk_probe_pmc.txt; cat $O/pk_probe_pmc.txt
rm -rf $O/pmc $O/pk_probe
