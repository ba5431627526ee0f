cd /root/repo; mkdir -p gpurun_out/r05n; O=gpurun_out/r05n
timeout 600 python -m pytest tests/test_gpu_train16.py -q -m gpu -s 2>&1 | tail -5 > $O/tests16.txt
timeout 900 python tools/train16_check.py bf16 8 416 > $O/train16_check_bf16.txt 2>&1
timeout 900 python tools/train16_check.py f16 8 416 > $O/train16_check_f16.txt 2>&1
cat $O/tests16.txt; grep -v amdgpu $O/train16_check_bf16.txt $O/train16_check_f16.txt
