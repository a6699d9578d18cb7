#!/bin/bash
# Run ON THE GPU BOX: one GPU timing the shard sizes of a strong-scaling run of BASELINE configs[1] (1e8 queries split over
# P = 1, 2, 4, 8 GPUs -> NQ/P per GPU), with the region sweep forced on (min tiles per CU = 1) and off (huge), to place the
# sweep-vs-stream threshold.  Prints one line per (P, kernel).
for P in 1 2 4 8; do
  for T in 1 1000000; do
    MI_SWEEP_MIN_TILES_PER_CU=$T python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --shard-of $P 2>/dev/null | tail -1 | \
      python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('P=%d shard=%9d queries  min_tiles_per_cu=%-7s kernel=%-12s %.4f ms per step  (%.3e points/s on this GPU)' % ($P, r['config']['queries_per_gpu'], '$T', 'sweep' if '$T'=='1' else 'stream', r['roofline']['kernel_ms'], r['config']['queries_per_gpu']/r['roofline']['kernel_ms']*1e3))"
  done
done
