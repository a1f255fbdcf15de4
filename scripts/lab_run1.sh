#!/bin/bash
# first lab sweep: headline shapes at 6 tokens and 1 token
L=eagle-in-llama.cpp_amd/lib/lab_mmx
O=gpurun_out/lab1.log
mkdir -p gpurun_out; : > $O
run() { echo "== $*" >> $O; timeout -k 10 120 $L "$@" >> $O 2>&1; echo "rc $?" >> $O; }
run q4_K 4096 4096 6 plain stamps
run q4_K 4096 4096 6 norm stamps
run q4_K 4096 4096 6 qkv stamps
run q4_K 11008 4096 6 swiglu stamps
run q4_K 4096 11008 6 plain stamps
run q6_K 4096 11008 6 plain stamps
run q6_K 4096 4096 6 norm
run q6_K 32000 4096 6 norm
run q4_K 4096 4096 1 plain stamps
run q4_K 4096 4096 1 qkv
run q4_K 11008 4096 1 swiglu
run q4_K 4096 11008 1 plain
run q6_K 32000 4096 1 norm
run q4_K 4096 8192 1 norm
tail -5 $O
