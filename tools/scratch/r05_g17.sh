cd /root/repo; mkdir -p gpurun_out/r05q; O=gpurun_out/r05q
timeout 600 python tools/detector_train_host_profile.py bf16 2>&1 | grep -v amdgpu | cut -c1-160 > $O/host_profile_bf16.txt
timeout 600 python tools/detector_train_host_profile.py f32 2>&1 | grep -v amdgpu | head -12 | cut -c1-160 > $O/host_profile_f32.txt
timeout 600 python bench.py --no-cpu-baseline --no-accuracy --no-batch-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['traffic'], d['roofline']['traffic_source'][:60], d['bf16_storage_mode']['roofline']['traffic'])" > $O/traffic_check.txt 2>&1
cat $O/host_profile_bf16.txt $O/host_profile_f32.txt $O/traffic_check.txt
