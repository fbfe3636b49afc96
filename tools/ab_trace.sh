L=$PWD/corsair_amd/csrc
for v in k48 k32; do
lib=$L/libcorsair_hip.so; [ $v = k48 ] && lib=$L/libcorsair_hip_k48.so
CORSAIR_HIP_LIB=$lib CS_PF_EPS_W=3.3e-4 CS_PF_TRACE=32768 CS_PF_TRACE_FILE=gpurun_out/pf_$v.bin python tools/stage_timing.py 32 652 100000 > gpurun_out/st_$v.log 2>&1 && echo == $v && python tools/pf_trace.py gpurun_out/pf_$v.bin | head -12
done
