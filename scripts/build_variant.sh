#!/bin/bash
# A/B builds: scripts/build_variant.sh <name> <sed-expr> <file.hip> [<sed-expr> <file.hip> ...]
# compiles patched copies of the named sources (everything else reuses uformer_amd/lib/*.o) into ab/<name>/libuformer_hip.so;
# select it at run time with UFORMER_HIP_LIB=ab/<name>/libuformer_hip.so.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); name=$1; shift
out=$R/ab/$name; mkdir -p $out; objs=""
declare -A patched
while [ $# -ge 2 ]; do
    expr=$1; f=$2; shift 2
    src=$out/$f; [ -f $src ] || cp $R/uformer_amd/csrc/$f $src
    sed -i "$expr" $src; patched[$f]=1
done
for f in uf_core uf_gemm uf_lngemm uf_leff2 uf_attnblk uf_attn uf_elementwise uf_bwd uf_train uf_trainblk uf_pack uf_model; do
    if [ -n "${patched[$f.hip]}" ]; then
        sed -i "s|#include \"uf_internal.h\"|#include \"$R/uformer_amd/csrc/uf_internal.h\"|; s|#include \"uf_common.h\"|#include \"$R/uformer_amd/csrc/uf_common.h\"|" $out/$f.hip
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $out/$f.hip -o $out/$f.o
        objs="$objs $out/$f.o"
    else
        objs="$objs $R/uformer_amd/lib/$f.o"
    fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libuformer_hip.so $objs
echo "built $out/libuformer_hip.so"
