cd /root/repo; mkdir -p gpurun_out/r05c; O=gpurun_out/r05c/out.txt
MILLIEYE_WS32_DEBUG=1 timeout 300 python tools/conv_bench.py --batch 32 --tiles 3,50 --splits 1 --reps 3 --prewarm 0.01 --only "1x1 256->128" 2>&1 | grep -v amdgpu | sort | uniq -c | head >> $O
for t in 50 3; do bash tools/pmc_f32.sh r05c $t 52 256 128 1 1 >> $O 2>&1; done
MILLIEYE_WS32_VARIANT=1 bash tools/pmc_f32.sh r05c_v1 50 52 256 128 1 1 >> $O 2>&1
cat $O | cut -c1-400
