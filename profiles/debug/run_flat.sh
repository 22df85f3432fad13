set -o pipefail
for lib in "" profiles/microbench/lib_qg5.so; do
  for extra in "" "--retire"; do
    APT_LIB_PATH=$lib timeout -k 10 100 python profiles/grid_bench.py --s 64 --reps 2 $extra | cut -c1-75 || echo fail
  done
done
APT_GRID_WALK=items timeout -k 10 100 python profiles/grid_bench.py --s 64 --reps 2 | cut -c1-75
APT_GRID_WALK=items timeout -k 10 100 python profiles/grid_bench.py --s 64 --reps 2 --retire | cut -c1-75
