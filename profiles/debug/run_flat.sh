set -o pipefail
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "grid" > gpurun_out/flat_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/flat_tests.log
for extra in "" "--retire"; do timeout -k 10 100 python profiles/grid_bench.py --s 64 --reps 3 $extra | cut -c1-75 || echo fail; done
