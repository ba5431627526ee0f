cd /root/repo; mkdir -p gpurun_out/r05i; O=gpurun_out/r05i
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "tests rc=$?" >> $O/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout 2400 bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1
cp gpurun_out/prof/* $O/ 2>/dev/null
tail -3 $O/gpu_tests.txt; tail -2 $O/smoke.txt; ls $O; cut -c1-400 $O/r05_bench_full_b32.json
