# VALU issue-rate probe: builds valu_probe, counts the VALU instructions of its loop from the
# disassembly, runs it, then collects SQ counters on the same binary.  usage: bash tools/probes/run_valu_probe.sh <out dir>
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$(mkdir -p "$1" && cd "$1" && pwd)
cd $R/tools/probes
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o valu_probe valu_probe.hip --save-temps=obj 2> /dev/null
python3 count_valu.py valu_probe-hip-amdgcn-amd-amdhsa-gfx950.s k_eps_only > $O/valu_loop_instructions.txt
rm -f valu_probe-hip-* valu_probe-host-* valu_probe.hip-hip-*
# static loop count includes 8 rarely-taken tail blocks of ~17 instructions; the PMC pass below gives the executed count
STATIC=$(head -1 $O/valu_loop_instructions.txt | sed 's/.*loop_valu \([0-9]*\).*/\1/')
./valu_probe $STATIC 512 > $O/valu_probe.jsonl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_SALU --output-format csv -d $O/valu_pmc -o valu -- $R/tools/probes/valu_probe $STATIC 512 > $O/valu_probe_under_pmc.jsonl 2> $O/valu_pmc.err
cd $R
python3 tools/probes/valu_report.py $O > $O/valu_report.json
cat $O/valu_report.json
