mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -q -x > gpurun_out/r3/gputest6.log 2>&1; echo "gpu tests rc $?"; tail -4 gpurun_out/r3/gputest6.log
for th in 4 8 16; do echo -n "threads $th: "; FS_COPY_THREADS=$th timeout -k 10 300 python tools/bench_pcie.py; done 2>&1 | tee gpurun_out/r3/pcie_threads.txt
