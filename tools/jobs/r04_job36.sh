mkdir -p gpurun_out/r04_g
timeout 900 python bench.py 2> gpurun_out/r04_g/bench.err | grep '^{' > gpurun_out/r04_g/bench.json
for CFG in detector listener joint; do
  timeout 400 python bench.py --config $CFG --no-fp32 --no-cpu-baseline 2> gpurun_out/r04_g/bench_$CFG.err | grep '^{' > gpurun_out/r04_g/bench_$CFG.json
done
