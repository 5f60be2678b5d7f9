#!/bin/bash
# Dev aid: builds libmot_hip.so variants with extra -D flags into build/variants/<name>.so (only mot_embed.hip differs; the other
# objects are reused), for A/B timing on the GPU box via MOT_DEV_LIB.   usage: tools/variants.sh name1:"-DX -DY" name2:"" ...
set -euo pipefail
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/mixture-of-tokenizers_amd/csrc
out=$root/build/variants
mkdir -p "$out"
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -DMOT_DEV_ABLATION"
for spec in "$@"; do
    name=${spec%%:*}; extra=${spec#*:}
    ( cd "$src" && /opt/rocm/bin/hipcc $flags $extra -c ${VAR_SRC:-mot_embed.hip} -o "$out/$name.o" \
      && objs=$(ls *.o | grep -v "^$(basename ${VAR_SRC:-mot_embed.hip} .hip).o$") \
      && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/$name.so" $objs "$out/$name.o" ) &
done
wait
ls -la "$out"/*.so
