#!/bin/bash
# A/B harness: one library variant = ONE translation unit rebuilt with extra flags, linked with the in-tree objects of the others.
# usage: tools/ab/build_variant.sh <name> <source stem, e.g. i2r_hrformer_lp> [hipcc flags, e.g. -DI2R_ATT_RB78=2]  ->  tools/ab/lib_<name>.so
# (run after __graft_entry__.build(); the variants travel to the GPU box with the snapshot; tools select one with I2R_TOOL_LIB=tools/ab/lib_<name>.so)
R=$(cd $(dirname $0)/../.. && pwd)
S=$R/intra-and-inter-human-relation-network-for-mpee_amd/csrc
B=$S/build
name=$1; stem=$2; shift 2
mkdir -p /tmp/abv
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form -I $R/include -I $S "$@" -c $S/$stem.hip -o /tmp/abv/$name.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $B/*.o | grep -v "/$stem.o") /tmp/abv/$name.o -o $R/tools/ab/lib_$name.so
