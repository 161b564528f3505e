#!/bin/bash
# A/B harness (tuning): build library variants whose 16-bit conv translation units come from /tmp/v/<name>/i2r_conv_lp.inc
# usage: build_variants.sh name:defines ...   (e.g. np0:-DI2R_RING=0); objects of the other sources come from the in-tree build
R=/root/repo
S=$R/intra-and-inter-human-relation-network-for-mpee_amd/csrc
B=$S/build
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I $R/include"
for spec in "$@"; do
  v=${spec%%:*}; defs=${spec#*:}
  for t in bf16 f16; do
    cp $S/i2r_conv_$t.hip /tmp/v/$v/
    /opt/rocm/bin/hipcc $F -I /tmp/v/$v -I $S $defs -c /tmp/v/$v/i2r_conv_$t.hip -o /tmp/v/$v/$t.o &
  done
done
wait
for spec in "$@"; do
  v=${spec%%:*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $B/i2r_api.o $B/i2r_misc.o $B/i2r_hrformer.o $B/i2r_post.o $B/i2r_input.o $B/i2r_encoder.o $B/i2r_conv.o /tmp/v/$v/bf16.o /tmp/v/$v/f16.o -o $R/tools/ab/lib_$v.so
done
ls -la $R/tools/ab/*.so
