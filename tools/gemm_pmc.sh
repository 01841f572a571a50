set -e
R=$(pwd)
O=$R/gpurun_out/gemm_pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $O/p1 -o p1 -- python3 $R/tools/time_gemm.py > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --output-format csv -d $O/p2 -o p2 -- python3 $R/tools/time_gemm.py > $O/p2.log 2>&1
cd $R
python3 tools/pmc_kernel_means.py $O/means.json $O/p1 $O/p2 > /dev/null
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/gemm_pmc/means.json'))
for k,v in d.items():
    if 'gemm' in k: print(k, json.dumps({a:round(b,1) for a,b in v.items()}))
PY
