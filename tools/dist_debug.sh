#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
python - <<'PY'
import re, os
src = open('tests/test_gpu_dist.py').read()
w = re.search(r"WORKER = r'''(.*?)'''", src, re.S).group(1) % os.getcwd()
open('/tmp/w.py', 'w').write(w)
PY
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29731 WORLD_SIZE=2 HSA_ENABLE_IPC_MODE_LEGACY=0
RANK=0 LOCAL_RANK=0 python /tmp/w.py > $O/dist_r0.txt 2>&1 &
RANK=1 LOCAL_RANK=1 python /tmp/w.py > $O/dist_r1.txt 2>&1
wait
tail -12 $O/dist_r0.txt; echo ----; tail -12 $O/dist_r1.txt
