for cfg in "unset unset" "spt1 block=1024" "spt1 block=2048" "unset block=1024" "unset block=2048"; do
  set -- $cfg
  if [ "$1" = unset ]; then unset XRNERF_LIB; else export XRNERF_LIB=/root/repo/xrnerf_amd/libxrnerf_mi355_$1.so; fi
  if [ "$2" = unset ]; then unset XR_SC_TEST; else export XR_SC_TEST=$2; fi
  echo "== lib=$1 XR_SC_TEST=$2"
  TRACE_WIN=0.8 bash tools/gpu_call.sh r06aj trace 2>/dev/null | grep -E "k_scatter|iteration span"
done
