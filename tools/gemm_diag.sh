# Diagnostic instantiations of the bf16x3 product kernel (d3p_vae.hip, D3P_GEMM_DIAG): what does each piece of the K loop cost?
#   in the container:  bash tools/gemm_diag.sh build      (one library per variant under tools/scratch/diag/)
#   on the GPU box:    bash tools/gemm_diag.sh run        (tools/time_gemm.py with every variant; output: gpurun_out/gemm_diag.txt)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/tools/scratch/diag
VARIANTS=${GEMM_DIAG_VARIANTS:-"0 1 2 3 4 6 8 14 16 30"}
case "$1" in
build)
    mkdir -p $D
    python3 -c "import d3p_amd._lib as L; L.build()"
    for v in $VARIANTS; do
        ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DD3P_GEMM_DIAG=$v -c $R/d3p_amd/csrc/d3p_vae.hip -o $D/vae_$v.o &&
          hipcc --offload-arch=gfx950 -fPIC -shared -o $D/libd3p_diag_$v.so $(ls $R/build/d3p_hip/*.o | grep -v d3p_vae) $D/vae_$v.o && rm $D/vae_$v.o ) &
        if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi
    done
    wait
    ls -la $D
    ;;
run)
    mkdir -p $R/gpurun_out
    : > $R/gpurun_out/gemm_diag.txt
    for v in $VARIANTS; do
        echo "== D3P_GEMM_DIAG=$v" >> $R/gpurun_out/gemm_diag.txt
        D3P_HIP_LIBRARY=$D/libd3p_diag_$v.so python3 $R/tools/time_gemm.py 2>&1 | cut -c1-62 >> $R/gpurun_out/gemm_diag.txt
    done
    cat $R/gpurun_out/gemm_diag.txt
    ;;
*) echo "usage: $0 build | run"; exit 2 ;;
esac
