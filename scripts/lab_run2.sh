#!/bin/bash
L=eagle-in-llama.cpp_amd/lib/lab_mmx
O=gpurun_out/lab2.log
mkdir -p gpurun_out; : > $O
run() { echo "== $*" >> $O; timeout -k 10 120 $L "$@" >> $O 2>&1; echo "rc $?" >> $O; }
run q4_K 4096 4096 6 plain stamps "mmx d2 pf2 hoist g1 next"
run q4_K 4096 4096 6 norm stamps "mmx d2 pf2 hoist g1 next"
run q4_K 4096 4096 6 qkv stamps "mmx d2 pf2 hoist g1 next"
run q4_K 11008 4096 6 swiglu
run q4_K 4096 11008 6 plain stamps "mmx d2 pf2 hoist g1 next"
run q6_K 4096 11008 6 plain
run q4_K 4096 4096 1 plain stamps "mmx d2 pf2 hoist g1 next"
run q6_K 32000 4096 1 norm
tail -3 $O
