cd /root/repo; mkdir -p gpurun_out/r05m; O=gpurun_out/r05m
timeout 1200 python -m pytest tests/test_gpu_train16.py -q -m gpu -s > $O/tests16.txt 2>&1; echo "tests rc=$?" >> $O/tests16.txt
grep "^\[\|passed\|failed\|Error" $O/tests16.txt | cut -c1-600
