#!/bin/bash
# ablation libraries for the fused HRFormer block kernels: tools/ab/lib_abl<k>.so = both translation units built with -DI2R_ABL=<k>
R=$(cd $(dirname $0)/../.. && pwd)
S=$R/intra-and-inter-human-relation-network-for-mpee_amd/csrc
B=$S/build
mkdir -p /tmp/abv
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form -I $R/include -I $S"
for k in "$@"; do
  ( /opt/rocm/bin/hipcc $F -DI2R_ABL=$k -c $S/i2r_hrformer_lp.hip -o /tmp/abv/abl${k}_a.o && /opt/rocm/bin/hipcc $F -DI2R_ABL=$k -c $S/i2r_hrformer_mlp.hip -o /tmp/abv/abl${k}_m.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $B/*.o | grep -v "/i2r_hrformer_lp.o\|/i2r_hrformer_mlp.o") /tmp/abv/abl${k}_a.o /tmp/abv/abl${k}_m.o -o $R/tools/ab/lib_abl$k.so ) &
done
wait
