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
# every in-tree object except the two conv translation units rebuilt above
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $B/*.o | grep -v "/i2r_conv.o\|/i2r_conv_wino.o") $O $OW -o $R/tools/ab/$OUT
