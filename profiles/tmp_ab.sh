mkdir -p gpurun_out
L=build/variants
python profiles/k1_variant_time.py 10000000 $L/k2b_head.so,$L/k2b_cur.so,$L/k2b_parts.so,$L/k2b_head.so,$L/k2b_cur.so,$L/k2b_parts.so 2>&1 | grep k2b_expand | tee gpurun_out/k2b_parts_ab.txt
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
