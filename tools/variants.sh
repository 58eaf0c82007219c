echo "== default (two-panel, trsm8)"; python tools/gpu_probe.py 2048,16,128 1024,8,32 4096,32,8 2>&1 | grep -E "streams=1|timing"
echo "== BGP_TWO_PANEL=0"; BGP_TWO_PANEL=0 python tools/gpu_probe.py 2048,16,128 2>&1 | grep -E "streams=1|timing"
