#!/bin/bash
# round 6: tools/lds_atomic_two_process_probe.hip alone, as two processes side by side, and with two queues of one process
# usage (GPU box): bash tools/r06_lds_probe.sh [seconds]
S=${1:-30}
B=tools/bin/lds_atomic_two_process_probe
[ -x $B ] || hipcc --offload-arch=gfx950 -O2 -o $B tools/lds_atomic_two_process_probe.hip
echo "== ALONE (one process, one queue)"; $B $S 1
echo "== ONE PROCESS, TWO QUEUES"; $B $S 2
echo "== TWO PROCESSES side by side"
$B $S 1 > /tmp/lds_probe_b.txt 2>&1 &
PB=$!
$B $S 1
wait $PB
echo "-- the other copy:"; cat /tmp/lds_probe_b.txt
