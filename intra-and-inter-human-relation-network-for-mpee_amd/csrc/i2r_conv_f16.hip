// f16 instantiations of the 16-bit implicit-GEMM convolution (see i2r_conv_lp.inc)
#define I2R_LP_DT 2
#include "i2r_conv_lp.inc"

void* i2r_pick_conv_f16(int nt, int mt, int cap, int pf) { return reinterpret_cast<void*>(pick_lp(nt, mt, cap, pf)); }
