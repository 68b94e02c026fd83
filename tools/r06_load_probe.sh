#!/bin/bash
# round 6: tools/global_load_two_process_probe.hip alone, then beside two training processes on the same GPU (the configuration in which
# tools/scatter_determinism_probe.py sees ~1 wrong scatter launch in 150)
S=${1:-40}
B=tools/bin/global_load_two_process_probe
[ -x $B ] || hipcc --offload-arch=gfx950 -O2 -o $B tools/global_load_two_process_probe.hip
echo "== ALONE"; $B 15
echo "== BESIDE TWO TRAINING PROCESSES"
python tools/scatter_determinism_probe.py noise $((S + 25)) > /dev/null 2>&1 &
P1=$!
python tools/scatter_determinism_probe.py noise $((S + 25)) > /dev/null 2>&1 &
P2=$!
sleep 20
$B $S
wait $P1 $P2
