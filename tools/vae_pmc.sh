# Counters of the VAE step's kernels at HEAD: three rocprofv3 --pmc passes over tools/time_vae_step.py, one over the
# MFMA probe (calibration of the matrix-pipe counter), and a kernel-trace --stats pass; outputs under gpurun_out/vae_pmc.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/vae_pmc; rm -rf $O; mkdir -p $O
COMMIT=${PROFILE_COMMIT:-unknown}
(cd $R/tools/probes && hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip)
cd /tmp && export TMPDIR=/tmp
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $O/p1 -o p1 -- python3 $R/tools/time_vae_step.py > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/p2 -o p2 -- python3 $R/tools/time_vae_step.py > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $O/probe -o probe -- $R/tools/probes/mfma_probe > $O/probe.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o vae -- python3 $R/tools/time_vae_step.py > $O/stats.log 2>&1
cd $R
python3 tools/vae_gemm_pmc.py $O/vae_gemm_pmc.json $COMMIT $O/probe $O/p1 $O/p2 > $O/table.txt
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/vae_kernel_stats.csv
# keep the merge-back small: the raw per-dispatch CSVs of the probe and the trace files are not needed once reduced
find $O -name "*kernel_trace.csv" -delete
cat $O/table.txt
