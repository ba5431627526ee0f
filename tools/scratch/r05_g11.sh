cd /root/repo; mkdir -p gpurun_out/r05k; O=gpurun_out/r05k
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "tests rc=$?" >> $O/gpu_tests.txt
timeout 3000 bash tools/profile_r05.sh > $O/profile_r05.log 2>&1
cp gpurun_out/prof/* $O/ 2>/dev/null
for k in 1 2 3; do timeout 300 python tools/network_stress.py > $O/network_stress_$k.txt 2>&1; done
tail -3 $O/gpu_tests.txt; ls $O | head -60; cut -c1-300 $O/r05_bench_full_b32.json; cat $O/conv_traffic.json; tail -3 $O/network_stress_1.txt
