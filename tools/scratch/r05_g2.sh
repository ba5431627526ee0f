cd /root/repo; mkdir -p gpurun_out/r05b; O=gpurun_out/r05b/abl.txt
for only in "1x1 256->128" "1x1 512->256"; do
for abl in 0 1 2 3 4 6 7; do echo "ABL=$abl" >> $O; MILLIEYE_WS32_ABL=$abl timeout 300 python tools/conv_bench.py --batch 32 --tiles 50 --splits 1 --only "$only" 2>&1 | grep -v amdgpu >> $O; done
echo "zero x" >> $O; timeout 300 python tools/conv_bench.py --batch 32 --tiles 3,50 --splits 1 --zero --only "$only" 2>&1 | grep -v amdgpu >> $O
done
echo "3x3 52 zero / random" >> $O; timeout 300 python tools/conv_bench.py --batch 32 --tiles 3 --splits 1 --zero --only "3x3 128->256" 2>&1 | grep -v amdgpu >> $O
timeout 300 python tools/conv_bench.py --batch 32 --tiles 3 --splits 1 --only "3x3 128->256" 2>&1 | grep -v amdgpu >> $O
cat $O | cut -c1-160
