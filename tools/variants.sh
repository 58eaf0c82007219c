for a in 0 1 2 4 8 16 6 31; do echo -n "ABL=$a potrf: "; BGP_ABL=$a python tools/gpu_probe.py 2048,16,128 2>&1 | grep timing | sed 's/.*"potrf": {"ms": \([0-9.]*\).*/\1 ms per 16 launches/'; done
