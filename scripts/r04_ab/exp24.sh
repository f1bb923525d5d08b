# kernel trace of the hash loop with shared chains of 4 (one rank over RCCL, 2 lanes): is the GPU the bound?
bash scripts/r04_trace_hash.sh 2 12 2>&1 | head -40
rm -f gpurun_out/prof/r04_hash2_kernel_trace.csv
