#!/bin/bash
# round 3, call 25: same-box A/B of the staged attention stores: libuvx_prev.so = the build before them (HEAD's attention.hip)
run() {
  UVX_LIB=$2 timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline $3 > gpurun_out/r3c25_$1.json 2> gpurun_out/r3c25_$1.err
  python -c "
import json; d=json.load(open('gpurun_out/r3c25_$1.json')); print('%-16s ms/step %.2f loss %.5f' % ('$1', d['ms_per_step'], d['loss']))"
}
P=$PWD/ultravox_amd/libuvx_prev.so; N=$PWD/ultravox_amd/libuvx.so
run prev_a $P; run new_a $N; run prev_b $P; run new_b $N; run prev_c $P; run new_c $N
run prev_lora $P "--audio-lora-r 8"; run new_lora $N "--audio-lora-r 8"
