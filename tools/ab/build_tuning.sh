#!/bin/bash
# tuning build of the fp32 conv translation unit (-DI2R_TUNING: phase stamps, ablation switches, env overrides) linked with the
# in-tree objects of the other sources -> tools/ab/$OUT (default lib_tuning.so; git-ignored; tools/stamp_group.py, tools/one_conv.py load it)
R=${GRAFT_REPO_ROOT:-/root/repo}
S=$R/intra-and-inter-human-relation-network-for-mpee_amd/csrc
B=$S/build
OUT=${OUT:-lib_tuning.so}
O=/tmp/tun/${OUT%.so}.o
mkdir -p /tmp/tun
OW=/tmp/tun/${OUT%.so}_wino.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form -I $R/include -I $S -DI2R_TUNING "$@" -c $S/i2r_conv.hip -o $O || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form -I $R/include -I $S -DI2R_TUNING "$@" -c $S/i2r_conv_wino.hip -o $OW || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $B/i2r_api.o $B/i2r_misc.o $B/i2r_hrformer.o $B/i2r_hrformer_lp.o $B/i2r_hrformer_mlp.o $B/i2r_post.o $B/i2r_input.o $B/i2r_encoder.o \
  $O $OW $B/i2r_conv_bf16.o $B/i2r_conv_f16.o -o $R/tools/ab/$OUT
