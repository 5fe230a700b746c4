export TMPDIR=/tmp
for t in 1024 512 256; do
  echo "== PT_GEMM_T64N=$t"
  PT_GEMM_T64N=$t python tools/bench_tomp.py --reps 50 --graph 2>&1 | tail -1 | cut -c1-300
  PT_GEMM_T64N=$t rocprofv3 --kernel-trace --stats -d gpurun_out/tompprof -o k -- python tools/bench_tomp.py --reps 10 > /dev/null 2>&1
  python tools/rocpd_by_grid.py $(find gpurun_out/tompprof -name "*.db" | head -1) 100 | grep "k_gemm" 
  rm -rf gpurun_out/tompprof
done
PT_GEMM_T64N=256 python -m pytest tests -m gpu -x -q -k tomp 2>&1 | tail -2
