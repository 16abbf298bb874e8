mkdir -p gpurun_out
for c in 1 2 3 4 5; do
  if [ $c = 5 ]; then python bench.py --config $c --steps 5 --warmup 2 > gpurun_out/r03_bench_c$c.json 2> gpurun_out/r03_bench_c$c.err
  else python bench.py --config $c > gpurun_out/r03_bench_c$c.json 2> gpurun_out/r03_bench_c$c.err; fi
  tail -c 600 gpurun_out/r03_bench_c$c.json
done
