timeout 300 python -m pytest tests/test_dense_gpu.py tests/test_full_size_gpu.py -x -q 2>&1 | tail -3
tools/ab_run.sh 4000000 w8 w4
for b in 1 3; do echo "w4 BPC=$b"; MEVI_H1_BPC=$b MEVI_PROBE_LIB=tools/probes/ab/libw4.so timeout 300 python tools/probe_dense.py 4000000 2>&1 | tail -1; done
