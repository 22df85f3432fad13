set -o pipefail
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "grid" > gpurun_out/flat_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/flat_tests.log
for lib in "" profiles/microbench/lib_qg4.so; do
  for extra in "" "--retire"; do
    APT_LIB_PATH=$lib timeout -k 10 100 python profiles/grid_bench.py --s 16 --reps 3 $extra | cut -c1-75 || echo fail
  done
done
