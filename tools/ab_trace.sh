# per-workgroup trace of one prefilter launch (round starting at iteration 32768) on the bench shape
CS_RANSAC_OVERLAP=0 CS_PF_TRACE=32768 CS_PF_TRACE_FILE=gpurun_out/pf.bin python tools/stage_timing.py 32 652 100000 > gpurun_out/st.log 2>&1 && python tools/pf_trace.py gpurun_out/pf.bin
