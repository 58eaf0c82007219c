echo "== default"; python tools/gpu_probe.py 2048,16,128 2>&1 | grep -E "streams=1|timing"
echo "== BGP_SYRK4=1"; BGP_SYRK4=1 python tools/gpu_probe.py 2048,16,128 1024,8,32 4096,32,8 2>&1 | grep -E "streams=1|timing"
