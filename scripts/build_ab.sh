#!/bin/bash
# A/B builds by compile-time switch: scripts/build_ab.sh <name> "<-D flags>" <file.hip> [<file.hip> ...]
# compiles the named sources with the flags (everything else reuses uformer_amd/lib/*.o) into ab/<name>/libuformer_hip.so;
# select it at run time with UFORMER_HIP_LIB=ab/<name>/libuformer_hip.so.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); name=$1; flags=$2; shift 2
out=$R/ab/$name; mkdir -p $out; objs=""
declare -A patched
for f in "$@"; do patched[$f]=1; done
for f in uf_core uf_gemm uf_lngemm uf_leff2 uf_attnblk uf_attn uf_elementwise uf_bwd uf_train uf_trainblk uf_pack uf_model; do
    if [ -n "${patched[$f.hip]}" ]; then
        extra=""; [ "$f" = uf_attnblk ] && [ -z "$AB_SLP" ] && extra="-fno-slp-vectorize"      # as __graft_entry__.EXTRA_FLAGS (AB_SLP=1: without, for A/B runs)
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra $flags -c $R/uformer_amd/csrc/$f.hip -o $out/$f.o
        objs="$objs $out/$f.o"
    else
        objs="$objs $R/uformer_amd/lib/$f.o"
    fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libuformer_hip.so $objs
echo "built $out/libuformer_hip.so ($flags)"
