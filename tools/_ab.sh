for x in 1 0 1 0; do
  export PHL_BLUR_PAIRS=$x
  python3 bench.py --no-cpu-baseline --steps 20 > gpurun_out/ab_pairs$x.log 2>&1 || exit 1
  echo "pairs $x" >> gpurun_out/ab_pairs.log
  tail -1 gpurun_out/ab_pairs$x.log | cut -c1-200 >> gpurun_out/ab_pairs.log
done
