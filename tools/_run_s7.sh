L=gym_pomdp_amd/_lib
for v in _base ""; do echo "== variant ${v:-new}"; python tools/gpu_variant_check.py $L/libpomdp_hip$v.so rock rock15 stochrock tag tiger network battleship battleship5 2>&1 | grep -v amdgpu; done > gpurun_out/s7_check.log 2>&1
python tools/gpu_small_shards.py $L/libpomdp_hip_base.so $L/libpomdp_hip.so > gpurun_out/s7_shards.log 2>&1
for v in _base ""; do POMDP_LIB=$L/libpomdp_hip$v.so LT_ENV=rock,tag,tiger,network,battleship,battleship5,stochrock LT_KS=20,64 python tools/gpu_launch_time.py; done 2>&1 | grep -v amdgpu > gpurun_out/s7_launch.log
diff <(sed -n '/base/,/new/p' gpurun_out/s7_check.log | grep -v "==" | cut -c1-40) <(sed -n '/new/,$p' gpurun_out/s7_check.log | grep -v "==" | cut -c1-40) && echo CHECKSUMS EQUAL
grep -v "amdgpu" gpurun_out/s7_shards.log; cat gpurun_out/s7_launch.log
