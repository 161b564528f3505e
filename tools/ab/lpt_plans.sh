for i in 1 2; do for v in "" a42 b2121 b1212 a141 b2112 b1221; do echo -n "plan [$v]: "; TOOL_LPT=$v python tools/one_conv.py 32 200 group 2>&1 | grep "per program"; done; done
