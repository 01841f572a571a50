# Round evidence for the step kernel (run on the GPU box through gpurun): phase anatomy of the chained launch in both geometries,
# per-wave finishing times, A/B timings, the per-opcode VALU probe and the wave-order probe.  Output: gpurun_out/anatomy/
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/anatomy; rm -rf $O; mkdir -p $O
cd $R
A=$O/chain_anatomy.txt
echo "# k_logreg_chain, phase stamps of one chained step (D3P_DBG=32, tools/time_chained.py 512); us after the LAST arrival of the previous step" > $A
echo "## 16-wave form (128 workgroups per step; production)" >> $A
D3P_DBG=32 python tools/time_chained.py 512 2>&1 >/dev/null | grep -v amdgpu.ids >> $A
echo "## 16-wave form without the gradient atomics (D3P_DBG=36; results wrong, timing only)" >> $A
D3P_DBG=36 python tools/time_chained.py 512 2>&1 >/dev/null | grep -v amdgpu.ids >> $A
echo "## 8-wave form (256 workgroups per step, two resident per CU: round 2's geometry with this round's arithmetic; D3P_CHAIN_W8=1)" >> $A
D3P_CHAIN_W8=1 D3P_DBG=32 python tools/time_chained.py 512 2>&1 >/dev/null | grep -v amdgpu.ids >> $A
echo "## 16-wave form: when each WAVE of a workgroup finished its two examples (D3P_DBG=288), us after the workgroup's first wave" >> $A
D3P_DBG=288 python tools/time_chained.py 512 2>&1 >/dev/null | grep -v amdgpu.ids >> $A
echo "## ... after the logit and the sigmoid (D3P_DBG=800)" >> $A
D3P_DBG=800 python tools/time_chained.py 512 2>&1 >/dev/null | grep -v amdgpu.ids | head -4 >> $A
echo "## ... after the z / dot-product loop (D3P_DBG=1312)" >> $A
D3P_DBG=1312 python tools/time_chained.py 512 2>&1 >/dev/null | grep -v amdgpu.ids | head -4 >> $A
J=$O/chain_ab.jsonl
for i in 1 2; do
TAG=w16 python tools/time_chained.py 2048 >> $J
TAG=w8 D3P_CHAIN_W8=1 python tools/time_chained.py 2048 >> $J
done
TAG=w16_intercept python tools/time_chained.py 2048 4096 512 1 >> $J
TAG=w8_intercept D3P_CHAIN_W8=1 python tools/time_chained.py 2048 4096 512 1 >> $J
TAG=w16_B32768 python tools/time_chained.py 512 32768 >> $J
TAG=generic_B32768 D3P_CHAIN_W8=1 python tools/time_chained.py 512 32768 >> $J
TAG=w16_B8192 python tools/time_chained.py 1024 8192 >> $J
TAG=generic_B8192 D3P_CHAIN_W8=1 python tools/time_chained.py 1024 8192 >> $J
for nw in 64 96 127 136 160 256; do TAG=w16_nw$nw D3P_CHAIN_NW=$nw python tools/time_chained.py 2048 >> $J; done
tools/probes/valu_opcode_probe > $O/valu_opcodes.jsonl
tools/probes/wave_order_probe > $O/wave_order_probe.jsonl
python tools/time_gmm_step.py > $O/gmm_time.txt 2>&1
ls -la $O
