#!/bin/bash
# Is the library's choice BETWEEN its kernels right, shape by shape?  The benchmark with per-shape clocks under every selection
# tunable of a -DMSG_TUNING build (python -m multi_stylegan_amd.build --variant tuning --flags=-DMSG_TUNING), one process each;
# tools/selection_sweep.py then compares every (direction, shape) against two default runs.  GPU box:  tools/selection_sweep.sh
run() { name=$1; shift; env "$@" MSG_LIB_VARIANT=tuning MSG_CLOCK_SHAPES=1 python bench.py --no-cpu-baseline --no-fp32-leg --no-h2d-leg > gpurun_out/sweep_$name.json 2> gpurun_out/sweep_$name.err; echo "$name exit $?"; }
run s0 MSG_NOP=1
run variant1 MSG_CONV_VARIANT=1          # generic forward kernel: LDS-DMA staging everywhere
run variant2 MSG_CONV_VARIANT=2          # ... register staging everywhere
run pp0 MSG_CONV_PP=0                    # no ping-pong kernel
run shortk0 MSG_CONV_ROW3_SHORTK=0       # short-K layers on the 256 x 256 tile
run narrow0 MSG_CONV_ROW3_NARROW=0       # no 128 x 128 row-sharing tile
run upconv0 MSG_CONV_UPCONV=0            # no activation-stationary up-conv kernel
run w32_0 MSG_CONV_ROW3_W32=0            # 32-wide maps off the row-sharing kernel
run wgrow3_0 MSG_WGRAD_ROW3=0            # generic weight-gradient kernel everywhere
run wgw32_0 MSG_WGRAD_ROW3_W32=0
run lean0 MSG_CONV_LEAN=0
run s0b MSG_NOP=1
